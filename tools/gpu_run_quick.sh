#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
mkdir -p gpurun_out/q
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/q/pytest_full.log 2>&1
grep -n "^E  \|FAILED\|passed\|failed" gpurun_out/q/pytest_full.log | head -20
