timeout -k 10 400 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo tests_rc=$rc; tail -15 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
for cfg in dcgan28 dcgan64 classifier; do
timeout -k 10 200 python bench.py --config $cfg --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-190 || exit 1
done
for v in "512 8" "1024 4" "512 4"; do set -- $v; echo "cap=$1 rows=$2"
DDL_CR_SMALL_CAP=$1 DDL_CR_SMALL_ROWS=$2 timeout -k 10 200 python bench.py --config dcgan28 --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-190 || exit 1
done
