#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_window.py -x -q > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_sweep.py -q -k "random_models or column" > $O/pytest_sweep.log 2>&1; tail -5 $O/pytest_sweep.log
for cu in 2 4; do
  SDP_STG_CU=$cu timeout 600 python bench.py --config coupled256 --kernel staged --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_coupled_cu$cu.json 2> $O/bench_coupled_cu$cu.err; python -c "
import json,sys; d=json.load(open('$O/bench_coupled_cu$cu.json')); print('cu$cu', d['ms_per_step'], d['roofline']['kernel_ms'])"; tail -2 $O/bench_coupled_cu$cu.err
done
PROF_STEPS=3 timeout 900 bash tools/profile_bench.sh r02_coupled_staged coupled256_f64_staged --config coupled256 --kernel staged > $O/prof_staged.log 2>&1; tail -32 $O/prof_staged.log
python - <<'PY' > $O/window_time.txt 2>&1
import numpy as np, sys
sys.path.insert(0, 'tests')
from stodynprog_amd import models
for kernel in ('auto', 'staged', 'generic'):
    _, s = models.synthetic3d(N=16)
    s.discretize_state(0, 1, 1024, 0, 1, 128, 0, 1, 128)
    s.kernel = kernel
    prob = s._problem()
    prob.set_value(np.zeros(s._state_grid_shape))
    prob.bench_sweeps(1)
    loop, k = prob.bench_sweeps(3)
    print(kernel, s.backend_info['kernel'], s.backend_info.get('row_window'), 'ms/sweep', k / 3, flush=True)
    prob.close()
PY
cat $O/window_time.txt
