#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then the PMC counters in
# their own passes (rocprofv3 --pmc must not be combined with trace domains).
# Usage (from the repo root, through gpurun):
#     bash tools/profile_bench.sh <tag> <pmc_key> [bench args]
# Writes gpurun_out/prof_<tag>/{summary.txt,kernel_stats.csv,pmc_<pmc_key>.json}; copy the three
# into profiles/ (as <tag>_summary.txt, <tag>_kernel_stats.csv, pmc_<pmc_key>.json) to commit them.
set -u
TAG=${1:-r02}; shift || true
KEY=${1:-synth256_f64_column}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps ${PROF_STEPS:-10} --warmup ${PROF_WARMUP:-2} --no-cpu-baseline --no-filter-check --no-other-configs $*"
run_pmc() {   # name counters...
    local name=$1; shift
    # (a pass that hangs -- the TA_* counters did, for twenty minutes, in round 6 -- must not take the call with it)
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $ARGS > "$OUT/bench_$name.json" 2> "$OUT/$name.log"
}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS
run_pmc l2 TCC_HIT_sum TCC_MISS_sum
run_pmc f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
if [ "${PROF_F32:-0}" = "1" ]; then
    run_pmc f32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
fi
find "$OUT" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
python3 tools/summarize_prof.py "$OUT" "$KEY" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
