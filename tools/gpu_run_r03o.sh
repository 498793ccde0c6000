#!/bin/bash
python tools/bench_train.py 16 f32 2>&1 | tail -1
FSN_TN_SQUARE_F32=1 python tools/bench_train.py 16 f32 2>&1 | tail -1
FSN_TN_SQUARE_F32=1 timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x -rP 2>&1 | grep -E "passed|failed|margin|^E " | tail -12
