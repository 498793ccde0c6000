#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
set -u
export TMPDIR=/tmp
O=gpurun_out/q
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python tools/bench_train.py 16 f16 > $O/train.txt 2>&1
grep -a "train step" $O/train.txt
python tools/rocprof_kernel_calls.py $O/trace "gemm_kernel<0, 3, 2, 2, 4" 2
python tools/rocprof_kernel_calls.py $O/trace "gemm_kernel<0, 3, 2, 2, 2" 1
rm -rf $O/trace
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -2
