#!/bin/bash
# Where the instructions and the time of the resident-chunk kernel go: diagnostic builds that leave one part of a unit out
# (SDP_DIAG_SKIP of csrc/sdp_colres_kernel.h; results wrong), each timed and counted (SQ_INSTS_VALU etc. per dispatch).
#     [CONFIG=noisy256] bash tools/skip_survey.sh [bits ...]       (through gpurun)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/skip_survey
mkdir -p "$OUT"
BITS=${@:-0 1 2 4 8 16 32 64 128}
CFG=${CONFIG:-synth256}
for b in $BITS; do
  D="SDP_EXTRA_DEFINES=SDP_DIAG_SKIP=$b"
  timeout 120 python3 tools/ab_kernel.py --config $CFG --rounds 3 --sweeps 10 -- $D > "$OUT/t_$b.txt" 2>&1
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/p_$b" -- python3 tools/ab_kernel.py --config $CFG --rounds 1 --sweeps 5 -- $D > "$OUT/p_$b.txt" 2> "$OUT/p_$b.log"
  python3 - "$OUT/p_$b" "$b" "$OUT/t_$b.txt" <<'P'
import csv, glob, sys, collections, re
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'sdp_sweep' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v[len(v)//3:]) / len(v[len(v)//3:]) for k, v in acc.items()}
t = re.search(r'median ([0-9.]+)', open(sys.argv[3]).read())
print('skip {:>3s}: median {} ms  VALU {:.4g}  SALU {:.4g}  LDS {:.4g}  VMEM_RD {:.4g}  wait {:.3f}  wait_inst {:.3f}'.format(
    sys.argv[2], t.group(1) if t else '?', m.get('SQ_INSTS_VALU', 0), m.get('SQ_INSTS_SALU', 0), m.get('SQ_INSTS_LDS', 0),
    m.get('SQ_INSTS_VMEM_RD', 0), m.get('SQ_WAIT_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1),
    m.get('SQ_WAIT_INST_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1)))
P
done
