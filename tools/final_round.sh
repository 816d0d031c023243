# Full GPU visit: gpu tests, smoke, every bench config, steady-state + sequence profiles of every config.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?
echo "tests_rc=$rc"; tail -3 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
for cfg in dcgan28 dcgan64 classifier localsgd ensemble32; do
  timeout -k 10 240 python bench.py --config $cfg --steps 100 --warmup 10 > gpurun_out/bench_$cfg.log 2>&1 || { echo "bench $cfg failed"; tail -5 gpurun_out/bench_$cfg.log; exit 1; }
  tail -1 gpurun_out/bench_$cfg.log | cut -c1-200
done
bash tools/prof_cfg.sh dcgan28 gan && bash tools/prof_cfg.sh dcgan64 gan && bash tools/prof_cfg.sh classifier || exit 1
echo done
