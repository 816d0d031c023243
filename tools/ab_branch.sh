timeout -k 10 200 python -u -m pytest tests/test_network_gpu.py tests/test_rccl_gpu.py tests/test_classifier_gpu.py -q -x --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo tests_rc=$rc; tail -15 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
for b in 0 1; do for cfg in dcgan28 dcgan64 classifier; do
echo "BRANCH=$b $cfg"; DDL_BRANCH=$b timeout -k 10 200 python bench.py --config $cfg --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-190 || exit 1
done; done
