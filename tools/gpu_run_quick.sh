#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/gru_pmc
mkdir -p $O
B1="python tools/bench_family.py gru 64"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/pass$i -- $B1 > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i ($C) rc=$?"
done
python tools/rocprof_pmc.py $O/pmc $O/pmc_gru.json "lstm_rec_in_kernel" "lstm_rec_x_kernel" "gru_step1_kernel" "linear_small_out" "fb_chain_kernel" > $O/pmc_summary.txt 2>&1
tail -60 $O/pmc_summary.txt
rm -rf $O/pmc
