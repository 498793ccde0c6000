#!/bin/bash
# Round 4: HBM traffic (PMC) of the AMP training step's kernels: FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only).
set -u
O=gpurun_out/${1:-r04k}
mkdir -p $O
export TMPDIR=/tmp
C="python tools/bench_train.py 16 ${2:-f16}"
i=0
for CT in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CT --kernel-trace --output-format csv -d $O/pmc/pass$i -- $C > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i ($CT) rc=$?"
done
python tools/rocprof_pmc.py $O/pmc $O/pmc_train.json "gemm_tn" "lstm2_g" "fb_chain" "lstm2_group" > $O/pmc_summary.txt 2>&1
python - <<PY
import json
d=json.load(open("$O/pmc_train.json"))["kernels"]
for n,k in sorted(d.items()):
    print(f"{n[:60]:60s} read {k.get('hbm_read_bytes_corrected',0)/1e9:7.3f} GB  write {k.get('hbm_write_bytes',0)/1e9:7.3f} GB  ({k['dispatches']} dispatches)")
PY
rm -rf $O/pmc
