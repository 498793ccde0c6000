#!/bin/bash
# HBM traffic of one training step per kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate, as the guide
# prescribes) + a kernel trace of tools/bench_train.py.  usage: gpu_run_pmc_train.sh <tag> <arith> [bench_train args...]
set -u
O=gpurun_out/${1:-pmc_train}
A=${2:-f16}
shift; shift
mkdir -p $O
export TMPDIR=/tmp
C="python tools/bench_train.py 16 $A $*"
N=$(echo "$A $*" | tr ' =' '__' | sed 's/_*$//')
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$N/pass$i -- $C > $O/pmc_${N}_pass$i.log 2>&1
  echo "pmc pass $i ($P) rc=$?"; grep "train step" $O/pmc_${N}_pass$i.log
done
python tools/rocprof_pmc.py $O/pmc_$N $O/pmc_train_$N.json "lstm2_g16|gemm_tn|gemm_kernel|gemm_dx|fb_chain|to16|tr_|linear|mse|clip|g16_" > $O/pmc_train_$N.txt 2>&1
rm -rf $O/pmc_$N
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$N -- $C > $O/train_$N.txt 2>&1
DB=$(ls $O/trace_$N/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_train_$N.md "rocprofv3 --kernel-trace --stats -- $C"
grep "train step" $O/train_$N.txt
rm -rf $O/trace_$N
python - <<PY
import json
d = json.load(open("$O/pmc_train_$N.json"))
tot = 0
for n, k in sorted(d["kernels"].items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0) * kv[1]["dispatches"]):
    b = k.get("hbm_bytes_per_launch", 0) * k["dispatches"]
    tot += b
    print(f"{n[:90]:90s} x{k['dispatches']:4d}  {k.get('hbm_bytes_per_launch', 0) / 1e9:8.3f} GB/launch  r {k.get('hbm_read_bytes_corrected', 0) / 1e9:7.3f} w {k.get('hbm_write_bytes', 0) / 1e9:7.3f}")
print("total over the run (5 steps: 2 warm-up + 3 timed)", tot / 1e9, "GB ->", tot / 5e9, "GB per step")
PY
