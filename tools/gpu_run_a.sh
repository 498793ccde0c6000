#!/bin/bash
# GPU session A of round 2: full GPU test suite, the bench line, a rocprofv3 kernel trace and the PMC passes.
# Everything lands under gpurun_out/r02a/.
set -u
O=gpurun_out/${1:-r02a}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
(time timeout 600 python bench.py --steps 10 --warmup 3) > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $B > $O/trace.log 2>&1
DB=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats.md "rocprofv3 --kernel-trace --stats -- $B" && python tools/rocprof_timeline.py $O/trace > $O/timeline.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|GRBM_GUI_ACTIVE|FETCH_SIZE|WRITE_SIZE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES" | head -40 > $O/counters_available.txt
B1="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/pass$i -- $B1 > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i ($C) rc=$?"
done
python tools/rocprof_pmc.py $O/pmc $O/pmc.json > $O/pmc_summary.txt 2>&1
tail -30 $O/pmc_summary.txt
