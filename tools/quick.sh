timeout -k 10 300 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; echo tests_rc=$?; tail -4 gpurun_out/gpu_tests.log
timeout -k 10 120 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-200
DDL_ONECH=0 timeout -k 10 120 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-200
