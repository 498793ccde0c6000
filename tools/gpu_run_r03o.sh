#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_family.py -m gpu -q -x -rP -k "config5_full_size" 2>&1 | grep -E "passed|failed|config 5|^E " | tail -8
