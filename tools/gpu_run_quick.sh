#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
set -u
export TMPDIR=/tmp
O=gpurun_out/q
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_amp.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python tools/bench_train.py 16 f16 > $O/train.txt 2>&1
grep -a "train step" $O/train.txt
DB=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB $O/ks.md "x"; head -7 $O/ks.md | tail -3 | cut -c1-100
rm -rf $O/trace
