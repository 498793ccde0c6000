#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
bash tools/gpu_run_nocache.sh nocache_final 2>&1 | tee gpurun_out/nocache_final.txt
