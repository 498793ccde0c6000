#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
timeout 600 python -m pytest tests/test_gpu_family.py -m gpu -x -q -k "improved" 2>&1 | tail -2
