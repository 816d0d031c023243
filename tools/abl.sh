timeout -k 10 200 python -u -m pytest tests/test_hip_ops.py tests/test_network_gpu.py -m gpu -q -x --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
DDL_AUTOTUNE_VERBOSE=1 timeout -k 10 100 python tools/bench_conv.py 2>&1 | tail -32
timeout -k 10 120 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-200
