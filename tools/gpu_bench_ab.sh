# A/B of engine options on bench workloads: one line per run (value, ms per step, kernel ms); bench arguments one set per stdin line
while read -r a; do
[ -z "$a" ] && continue
python bench.py $a --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$a', '| rollouts/s', round(d['value']), '| ms/step', round(d['ms_per_step'],4), '| kernel ms', round(d['roofline']['kernel_ms'],4))"
done
