#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_residency.py tests/test_gpu_family.py -m gpu -q -x -rP 2>&1 | grep -E "passed|failed|gate:|two chain|^E " | head -40
