timeout -k 10 300 python -u -m pytest tests/test_hip_ops.py tests/test_network_gpu.py -m gpu -q -x --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo tests_rc=$rc; tail -4 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit 1
DDL_AUTOTUNE_VERBOSE=1 timeout -k 10 200 python tools/bench_wgrad.py 1000 2>&1 | tail -20 || exit 1
timeout -k 10 120 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-200
