#!/bin/bash
# Full session: GPU suite, bench line, kernel traces (bench / training f32 + f16 / sibling models), PMC passes.
set -u
TAG=${1:-full}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -q -rP) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^E  |FAILED" $O/pytest.log | tail -8
(time timeout 900 python bench.py --steps 10 --warmup 3) > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json; tail -3 $O/bench.err
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $B > $O/trace.log 2>&1
DB=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_bench_b64.md "rocprofv3 --kernel-trace --stats -- $B" && python tools/rocprof_timeline.py $O/trace > $O/timeline.txt 2>&1
rm -rf $O/trace
for A in f32 f16 "f16 saves=32" "f32 norm=cumulative"; do
  C="python tools/bench_train.py 16 $A"
  N=$(echo $A | tr ' =' '__')
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$N -- $C > $O/train_$N.txt 2>&1
  DB=$(ls $O/trace_$N/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_train_$N.md "rocprofv3 --kernel-trace --stats -- $C"
  grep "train step" $O/train_$N.txt
  rm -rf $O/trace_$N
done
for W in "fast 256" "improved48 32" "improved48 1" "gru 64"; do
  set -- $W
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$1_$2 -- python tools/bench_family.py $1 $2 > $O/fam_$1_$2.txt 2>&1
  DB=$(ls $O/trace_$1_$2/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_$1_b$2.md "rocprofv3 --kernel-trace --stats -- python tools/bench_family.py $1 $2"
  tail -1 $O/fam_$1_$2.txt
  rm -rf $O/trace_$1_$2
done
B1="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/pass$i -- $B1 > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i ($C) rc=$?"
done
python tools/rocprof_pmc.py $O/pmc $O/pmc.json > $O/pmc_summary.txt 2>&1
tail -12 $O/pmc_summary.txt
rm -rf $O/pmc
# HBM traffic of the autocast step per kernel (both save modes): profiles/rNN_pmc_train_f16*.json
bash tools/gpu_run_pmc_train.sh $TAG/pmc_train f16 > $O/pmc_train_f16.log 2>&1; tail -1 $O/pmc_train_f16.log
bash tools/gpu_run_pmc_train.sh $TAG/pmc_train f16 saves=32 > $O/pmc_train_f16_saves32.log 2>&1; tail -1 $O/pmc_train_f16_saves32.log
