for a in "--workload c1" "--workload c1 --option threads=512" "--workload c1 --option threads=256" "--workload c2 --option threads=512" "--workload c2 --candidates-per-gpu 512" "--workload c2 --candidates-per-gpu 1024" "--workload c1 --candidates-per-gpu 1024" "--workload c1 --candidates-per-gpu 1024 --option threads=256"; do
python bench.py $a --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$a', round(d['value']), round(d['ms_per_step'],4))"
done
