#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/b; mkdir -p $O
nproc > $O/host.txt; python -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)), os.environ.get('OMP_NUM_THREADS'))" >> $O/host.txt
timeout 900 python -m pytest tests/test_gpu_staged.py -x -q > $O/pytest_staged.log 2>&1; tail -15 $O/pytest_staged.log
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_staged.py > $O/pytest.log 2>&1; tail -8 $O/pytest.log
for k in generic staged; do
  timeout 600 python bench.py --config coupled256 --kernel $k --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_coupled_$k.json 2> $O/bench_coupled_$k.err; cat $O/bench_coupled_$k.json | cut -c1-600; tail -2 $O/bench_coupled_$k.err
done
timeout 600 python bench.py --config synth256 --kernel staged --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_synth_staged.json 2> $O/bench_synth_staged.err; cut -c1-400 $O/bench_synth_staged.json
PROF_STEPS=3 timeout 900 bash tools/profile_bench.sh r02_coupled_staged coupled256_f64_staged --config coupled256 --kernel staged > $O/prof_staged.log 2>&1; tail -32 $O/prof_staged.log
