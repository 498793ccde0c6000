#!/bin/bash
set -u
O=gpurun_out/${1:-r03j}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_family.py tests/test_gpu_parity.py -m gpu -q -rP -k "norm or stft or istft or improved or fast or family or variant or transform or fullband") > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^E  |FAILED|HIP vs fp64" $O/pytest.log | tail -40
for W in "improved48 32" "improved48 1" "fast 256"; do set -- $W; timeout 200 python tools/bench_family.py $1 $2 2>&1 | tail -1; done | tee $O/fam.txt
