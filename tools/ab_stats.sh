timeout -k 10 400 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo tests_rc=$rc; tail -12 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
for f in 0 1 0 1; do echo "PCONV_STATS=$f"; DDL_PCONV_STATS=$f timeout -k 10 200 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | cut -c12-60,120-175 || exit 1; done
for f in 0 1; do echo "PCONV_STATS=$f dcgan64"; DDL_PCONV_STATS=$f timeout -k 10 200 python bench.py --config dcgan64 --steps 60 --warmup 10 2>&1 | tail -1 | cut -c12-60,120-175 || exit 1; done
