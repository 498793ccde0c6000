#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
mkdir -p gpurun_out/q
timeout 300 python -m pytest tests/test_gpu_family.py -m gpu -x -q -k "gru_training" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_family.py tests/test_gpu_train.py tests/test_gpu_amp.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -3
