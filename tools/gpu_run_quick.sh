#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/gru9
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_family.py -m gpu -q -x -s -k "constructor_variants" 2>&1 | grep -v "^$" | tail -12 | tee $O/tests.txt
