#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then HBM counters in
# their own passes (rocprofv3 --pmc must not be combined with trace domains).
# Usage (from the repo root, through gpurun):  bash tools/profile_bench.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-fused $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $ARGS > "$OUT/bench_sq.json" 2> "$OUT/sq.log"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- python3 bench.py $ARGS > "$OUT/bench_l2.json" 2> "$OUT/l2.log"
find "$OUT" -name '*.csv' | head -50
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
