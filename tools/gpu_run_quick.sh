#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "large_batches or config2_full_size" 2>&1 | tail -1
