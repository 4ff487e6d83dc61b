timeout 120 python tools/filter_ab.py 256 2>&1 | tail -3
timeout 120 python tools/filter_ab.py 256 float32 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_filter.py -x -q 2>&1 | tail -3
