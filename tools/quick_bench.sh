#!/bin/bash
# A/B timing of kernel build knobs on the GPU box: each knob value gets its own
# code object (the knob is part of the generated source).
for mw in "$@"; do
  echo "== SDP_COL_MIN_WAVES=$mw"
  SDP_COL_MIN_WAVES=$mw python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
