#!/bin/bash
# Counter survey of sdp_sweep_col (or any kernel whose name contains $KERNEL) for ONE variant of a bench configuration:
#     bash tools/pmc_survey.sh <tag> [--config C] -- [K=V ..]          (through gpurun; one counter group per pass)
# Prints mean per dispatch of every counter, and a few ratios.  Output under gpurun_out/pmcs_<tag>/.
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmcs_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
KERNEL=${KERNEL:-sdp_sweep}
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/ab_kernel.py --rounds 2 --sweeps 5 "${ARGS[@]}" > "$OUT/$name.txt" 2> "$OUT/$name.log" || echo "pass $name failed" >&2
}
ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run sq3 SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS
run sq4 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp1 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run td TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$OUT" "$KERNEL" <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if sys.argv[2] in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {}
for k, v in sorted(acc.items()):
    v = v[len(v) // 3:]            # (the first dispatches of a process run slower)
    m[k] = sum(v) / len(v)
    print('{:40s} n={:3d} mean={:.6g}'.format(k, len(v), m[k]))
def ratio(a, b, label):
    if a in m and b in m and m[b]:
        print('{:60s} {:.4f}'.format(label, m[a] / m[b]))
ratio('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY / SQ_WAVE_CYCLES')
ratio('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES')
ratio('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES')
ratio('SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES')
ratio('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'LDS bank conflict share')
ratio('SQ_LDS_IDX_ACTIVE', 'SQ_BUSY_CYCLES', 'SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES')
ratio('TA_TA_BUSY_sum', 'GRBM_GUI_ACTIVE', 'TA_TA_BUSY_sum / GRBM_GUI_ACTIVE')
ratio('TCP_TCC_READ_REQ_LATENCY_sum', 'TCP_TCC_READ_REQ_sum', 'mean L1->L2 read latency (clk)')
if 'TCC_HIT_sum' in m: print('L2 hit rate {:.4f}'.format(m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum'])))
P
