#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
timeout 900 python -m pytest tests/test_gpu_family.py tests/test_gpu_train.py tests/test_gpu_amp.py tests/test_gpu_trainer.py -m gpu -x -q -rP 2>&1 | grep -E "passed|failed|margins|^E " | tail -8
python tools/bench_train.py 16 f32 2>&1 | tail -1
python tools/bench_train.py 16 f16 2>&1 | tail -1
for B in 32; do timeout 120 python tools/bench_family.py improved48 $B 2>&1 | tail -1; done
