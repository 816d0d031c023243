timeout -k 10 300 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; echo tests_rc=$?; tail -2 gpurun_out/gpu_tests.log
for impl in igemm pconv; do DDL_CONV=$impl timeout -k 10 120 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-190; done
