#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/gru5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gru_rows.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -14 > $O/tests.txt
tail -3 $O/tests.txt
timeout 300 python tools/bench_family.py gru 64 2>&1 | tail -1 | tee $O/bench_gru64.txt
timeout 300 python tools/bench_family.py gru 32 2>&1 | tail -1 | tee $O/bench_gru32.txt
timeout 300 python tools/bench_family.py gru 128 2>&1 | tail -1 | tee $O/bench_gru128.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python tools/bench_family.py gru 64 > $O/fam_gru_64.txt 2>&1
DB=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_gru_b64.md "rocprofv3 --kernel-trace --stats -- python tools/bench_family.py gru 64"
python tools/rocprof_gru_timeline.py $O/trace | tee $O/timeline.txt
rm -rf $O/trace
