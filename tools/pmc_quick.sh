#!/bin/bash
# quick PMC comparison: bash tools/pmc_quick.sh <tag> [K=V switches of the generated kernels ...]   (through gpurun)
# L2 hit / miss, fabric fetch, vector-memory instructions and wave cycles of sdp_sweep_col
set -u
TAG=$1; shift
DEFS=""; for kv in "$@"; do DEFS="$DEFS --debug-define $kv"; done
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-filter-check --no-other-configs $DEFS"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/l2" -- python3 bench.py $ARGS > "$OUT/b1.json" 2> "$OUT/l2.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS > "$OUT/b2.json" 2> "$OUT/fetch.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py $ARGS > "$OUT/b3.json" 2> "$OUT/sq.log"
python3 - "$OUT" <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'sdp_sweep' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('{:22s} n={:3d} mean={:.5g}'.format(k, len(v), sum(v) / len(v)))
if 'TCC_HIT_sum' in acc:
    h, m = sum(acc['TCC_HIT_sum']), sum(acc['TCC_MISS_sum'])
    print('L2 hit rate {:.3f}'.format(h / (h + m)))
P
