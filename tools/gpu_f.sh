#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/f; mkdir -p $O
rm -f $O/parity_report.txt
SDP_PARITY_REPORT=$PWD/$O/parity_report.txt timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
sort -u $O/parity_report.txt > $O/parity_report_sorted.txt
PROF_STEPS=3 timeout 900 bash tools/profile_bench.sh r02_coupled_staged coupled256_f64_staged --config coupled256 --kernel staged > $O/prof_staged.log 2>&1; tail -5 $O/prof_staged.log
PROF_STEPS=2 timeout 1200 bash tools/profile_bench.sh r02_coupled_direct coupled256_f64_generic --config coupled256 --kernel generic > $O/prof_direct.log 2>&1; tail -5 $O/prof_direct.log
python - <<'PY' > $O/window_time.txt 2>&1
import numpy as np
from stodynprog_amd import models
for kernel in ('auto', 'staged', 'generic'):
    _, s = models.synthetic3d(N=16)
    s.discretize_state(0, 1, 1024, 0, 1, 128, 0, 1, 128)
    s.kernel = kernel
    prob = s._problem()
    prob.set_value(np.zeros(s._state_grid_shape))
    prob.bench_sweeps(1)
    loop, k = prob.bench_sweeps(3)
    print('1024 x 128 x 128 grid, 64 controls, 32 w:', kernel, '->', s.backend_info['kernel'], s.backend_info.get('row_window'), 'ms/sweep {:.2f}'.format(k / 3), flush=True)
    prob.close()
PY
cat $O/window_time.txt
timeout 300 python examples/searev_storage.py > $O/example_searev.txt 2>&1; tail -3 $O/example_searev.txt
