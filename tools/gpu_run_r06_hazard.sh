#!/bin/bash
# Round 6: the store-data hazard - stand-alone probe + the BPTT kernel built without / with weaker forms of the hold.
set -u
O=gpurun_out/${1:-r06hazard}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 600 tools/bin/probe_store_hazard $O/store_hazard.csv) > $O/store_hazard.md 2> $O/store_hazard.err
echo "probe rc=$?"; tail -12 $O/store_hazard.md; tail -3 $O/store_hazard.err
for V in nohold hold2 hold3 shipped; do
  L="--lib tools/bin/$V.so"; [ $V = shipped ] && L=""
  for rep in 1 2; do
    timeout 300 python tools/diag_k32_bwd.py 5 $L > $O/diag_${V}_$rep.txt 2>&1
    echo "== $V run $rep"; grep -E "dx|dw_ih0|dw_hh0|db_ih0" $O/diag_${V}_$rep.txt | cut -c1-400
  done
done
