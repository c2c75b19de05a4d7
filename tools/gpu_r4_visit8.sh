#!/bin/bash
# Round 4, eighth GPU visit: two 512-thread workgroups per CU for the fused-horizon kernel (options threads = 512, lds_limit_kb = 80)
# against the default (1024 threads, whole LDS) at batches of several workgroups per CU.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
rm -f $OUT/r04h_share_cu.txt
run() {  # workload, B per GPU, extra options...
  wl=$1; b=$2; shift 2
  timeout 300 python bench.py --no-cpu-baseline --no-gradient --workload $wl --candidates-per-gpu $b "$@" > $OUT/tmp_bench.json 2> $OUT/tmp_bench.err
  python - "$wl" "$b" "$*" <<'PY' >> gpurun_out/r04h_share_cu.txt
import json, sys
try:
    d = json.load(open("gpurun_out/tmp_bench.json"))
    print(f"{sys.argv[1]} B={sys.argv[2]} [{sys.argv[3]}]: {d['value']:.0f} rollouts/s, kernel {d['roofline']['kernel_ms']:.4f} ms, parity cov {d['parity'].get('max_rel_cov')}, mean {d['parity'].get('max_abs_dmean')}")
except Exception as e:
    print(sys.argv[1:], "failed", e, open("gpurun_out/tmp_bench.err").read()[-400:])
PY
}
for b in 512 1024 4096; do
  run c2 $b
  run c2 $b --option threads=512 --option lds_limit_kb=80
  run c2 $b --option threads=512 --option lds_limit_kb=52
done
run c3 1024
run c3 1024 --option threads=512 --option lds_limit_kb=80
run c3 4096
run c3 4096 --option threads=512 --option lds_limit_kb=80
run c1 2048
run c1 2048 --option threads=512 --option lds_limit_kb=80
cat $OUT/r04h_share_cu.txt
