#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
(time timeout 1500 python -m pytest tests -m gpu -q -x) 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
