timeout -k 10 200 python -u -m pytest tests/test_hip_ops.py tests/test_network_gpu.py -m gpu -q -x --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
timeout -k 10 100 python tools/bench_wgrad.py 2>&1 | tail -6 | cut -c1-110
