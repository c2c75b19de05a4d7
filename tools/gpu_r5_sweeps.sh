line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', '| rollouts/s', round(d['value']), '| ms/step', round(d['ms_per_step'],4), '| kernel ms', round(d['roofline']['kernel_ms'],4))
except Exception as e: print('$1', 'unreadable', e)"; }
for wl in c3 c1; do
for ch in 0 16 24 32 40 48 64; do
  python bench.py --workload $wl --no-cpu-baseline --no-gradient --no-batch-check --steps 20 --warmup 3 --option rows_per_chunk=$ch 2>/dev/null | line "$wl rows_per_chunk=$ch"
done; done
for b in 512 1024 2048; do
  python bench.py --workload c2 --no-cpu-baseline --no-gradient --no-batch-check --steps 10 --warmup 2 --candidates-per-gpu $b 2>/dev/null | line "c2 B=$b default"
  python bench.py --workload c2 --no-cpu-baseline --no-gradient --no-batch-check --steps 10 --warmup 2 --candidates-per-gpu $b --option threads=512 --option lds_limit_kb=80 2>/dev/null | line "c2 B=$b two 512-thread workgroups per CU"
done
