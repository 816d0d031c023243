timeout -k 10 400 python -u -m pytest tests/test_hip_ops.py tests/test_classifier_gpu.py -m gpu -q -x --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo tests_rc=$rc; tail -3 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --config classifier --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-190 || exit 1
bash tools/prof_cfg.sh classifier
