#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
timeout 600 python -m pytest tests/test_gpu_residency.py -m gpu -x -q -rP -k several_streams 2>&1 | grep -E "gate:|two chain|passed|failed|^E " | tail
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python -m pytest tests/test_gpu_residency.py tests/test_gpu_streaming.py -m gpu -q -k "not hip_graph" 2>&1 | tail -3
