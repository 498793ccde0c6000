#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_family.py -m gpu -q -x -k "stacked_lstm_on_the_persistent or composed_variants or sequence_model_inference or fast" 2>&1 | grep -v "^$" | tail -8 | tee $O/tests_lstm_rounds.txt
for B in 128 96 80 64; do timeout 300 python tools/bench_composed_lstm.py $B 2>&1 | tail -1 | tee -a $O/composed_lstm3.txt; done
timeout 300 python tools/bench_family.py fast 512 2>&1 | tail -1 | tee -a $O/composed_lstm3.txt
timeout 300 python tools/bench_family.py fast 256 2>&1 | tail -1 | tee -a $O/composed_lstm3.txt
