#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/o; mkdir -p $O
timeout 600 python tools/host_rate.py > $O/host_rate.txt 2>&1; cat $O/host_rate.txt
SDP_HOST_NO_OVERLAP=1 timeout 600 python tools/host_rate.py > $O/host_rate_no.txt 2>&1; head -1 $O/host_rate_no.txt
timeout 1200 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_full_size.py -q -k "full_size or 512 or full_256 or value_iterations or column_layout" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
