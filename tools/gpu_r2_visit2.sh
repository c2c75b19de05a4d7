#!/bin/bash
# Round-2 second visit: full parity suite on the matrix-core streaming kernel, config-5 bench line (strong scaling mode, 256 candidates).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu_tail.log
timeout 60 tools/microbench/mfma_f64_rate > $OUT/fma_loop_microbench.txt 2>&1
timeout 600 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 --cpu-seconds 6 2>$OUT/bench_c5.err | tee $OUT/bench_c5.json | cut -c1-400
