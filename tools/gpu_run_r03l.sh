#!/bin/bash
set -u
O=gpurun_out/${1:-r03l}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_family.py -m gpu -q -rP -k "improved or norm") > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^E  |FAILED" $O/pytest.log | tail -10
python - <<'PY' 2>&1 | tee $O/fam.txt
import sys, os
sys.path.insert(0, "tools")
import bench_family as BF
import fullsubnet_amd.improved_fullsubnet as IM
pack = BF.build("improved48")
for B in (2, 4, 8, 16, 32):
    row = []
    for cap in (0, 10**6):
        IM.FUSE_SECTIONS_MAX_TILES = cap
        m = BF.family_step("improved48", B, model_pack=pack)
        row.append(m["ms_per_step"])
    print(f"improved48 B={B}: sections on their own streams {row[0]:.2f} ms, one wavefront {row[1]:.2f} ms")
PY
