#!/bin/bash
# repeated multi-rank bench runs under the stand-in: does anything die or stall?
#   bash tools/mock_loop.sh "<ranks ...>" <repeats> ["<exchange lists ...>"] [bench args]
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do
for n in $1; do
for ex in ${3:-rccl,direct,sparse,peer}; do
  f=gpurun_out/mock_loop_${n}_${rep}.txt
  SDP_BENCH_TRACE=1 MOCK_TIMEOUT=${MOCK_TIMEOUT:-400} SDP_COMM_EXCHANGES=$ex PYTHONFAULTHANDLER=1 python tools/mock8_bench.py $n --steps 3 --warmup 1 --no-cpu-baseline --no-filter-check ${4:-} > $f 2>&1
  st=$?
  echo "=== $n ranks, $ex, run $rep: status $st"
  if [ $st -eq 0 ]; then rm -f $f; else grep -m2 "fault\|Fatal\|did not finish" $f | cut -c1-200; fi
done; done; done
