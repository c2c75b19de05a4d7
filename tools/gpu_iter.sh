#!/bin/bash
# quick iteration: path parity tests, timing of all paths, phase profile (debug lib)
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_iter.log 2>&1; tail -4 gpurun_out/pytest_iter.log
bash tools/gpu_paths_bench.sh 2>&1 | grep -E "force_path=(0|1)" 
GPMPC_LIB=$PWD/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python -u tools/gpu_prof.py 2>&1 | grep -v amdgpu.ids | awk '/^==/ {print} /PROF/ {last=$0} /ms\/launch/ {print last; print}'
