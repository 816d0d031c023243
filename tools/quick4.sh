timeout -k 10 400 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo tests_rc=$rc; tail -3 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
for cfg in dcgan28 dcgan64 classifier; do
timeout -k 10 200 python bench.py --config $cfg --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-190 || exit 1
done
