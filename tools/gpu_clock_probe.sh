#!/bin/bash
# effective shader clock under the batch-major kernels: GRBM_GUI_ACTIVE (cycles the GPU was busy) / kernel duration
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/clk -o t -- python $REPO/tools/gpu_tiles_check.py c4 tile_overlap=0 > $OUT/clk.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/clk/*_results.db')[0]
cur = sqlite3.connect(db).cursor()
cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
print(cols)
rows = cur.execute("select kernel_name, dispatch_id, sum(value), max(value), count(*) from counters_collection where counter_name='GRBM_GUI_ACTIVE' group by kernel_name, dispatch_id").fetchall()
kc = [d[0] for d in cur.execute("select * from kernels limit 1").description]
print(kc)
name = "kernel_name" if "kernel_name" in kc else "name"
dur = {}
for r in cur.execute(f"select dispatch_id, {name}, end-start from kernels"):
    dur[r[0]] = (r[1], r[2])
agg = {}
for kname, disp, s, m, n in rows:
    if disp in dur and dur[disp][1] > 0:
        agg.setdefault(kname[:50], []).append((m, s, n, dur[disp][1]))
for k, v in agg.items():
    if len(v) < 3: continue
    mx = sum(x[0] for x in v) / len(v); d = sum(x[3] for x in v) / len(v); n = v[0][2]
    print(f"{k:50s} n_inst={n} avg max-instance GUI_ACTIVE {mx:.0f} cycles, avg duration {d/1e3:.1f} us -> {mx/d:.3f} GHz")
PY
rm -rf $OUT/clk
