#!/bin/bash
# One extra rocprofv3 PMC pass over bench.py: bash tools/pmc_pass.sh <tag> <counters...>
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused $PMC_BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        if 'sdp_sweep' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('{:28s} n={} mean={:.6g}'.format(k, len(v), sum(v) / len(v)))
PY
tail -3 "$OUT/log.txt"
