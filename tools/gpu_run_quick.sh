#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O
rm -f $O/fast_sweep.txt
for B in 72 80 100 144 200 208 240; do timeout 300 python tools/bench_family.py fast $B 2>&1 | tail -1 | tee -a $O/fast_sweep.txt; done
