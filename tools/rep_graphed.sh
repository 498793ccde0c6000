#!/bin/bash
for i in $(seq 1 24); do
  s=$(date +%s.%N)
  r=$(python -m pytest tests/test_gpu_streaming.py -q -x -k "graphed_call_replays and improved16" 2>&1 | grep -E "passed|failed" | tail -1)
  e=$(date +%s.%N)
  echo "$i: $r wall $(python -c "print(round($e-$s,1))")"
done
