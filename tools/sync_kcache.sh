#!/bin/bash
# Run on the GPU box (through gpurun) AFTER the GPU test-suite: copies every
# model code object the tests compiled there into gpurun_out/kcache/, from
# where `tools/sync_kcache.sh pull` (run locally) moves them into the in-tree
# cache.  Cache keys are content hashes, so stale files are simply never used.
if [ "$1" = "pull" ]; then
  mkdir -p stodynprog_amd/_kcache
  cp -n gpurun_out/kcache/*.hsaco gpurun_out/kcache/*.hip stodynprog_amd/_kcache/ 2>/dev/null
  ls stodynprog_amd/_kcache/*.hsaco | wc -l
else
  mkdir -p gpurun_out/kcache
  cp stodynprog_amd/_kcache/*.hsaco stodynprog_amd/_kcache/*.hip gpurun_out/kcache/
  ls gpurun_out/kcache/*.hsaco | wc -l
fi
