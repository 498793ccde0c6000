#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/ovl
mkdir -p $O
timeout 120 tools/bin/probe_overlap_f32 | tee $O/overlap_f32.txt
timeout 120 tools/bin/probe_overlap_f16 | tee $O/overlap_f16.txt
