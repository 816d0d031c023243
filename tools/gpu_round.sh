#!/bin/bash
# One GPU-box visit: gpu tests, every bench config, kernel-trace profiles.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ "$1" != "notest" ]; then
timeout -k 10 400 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
echo "tests_rc=$?"; tail -3 gpurun_out/gpu_tests.log
fi
for cfg in dcgan28 dcgan64 classifier localsgd; do
  timeout -k 10 240 python bench.py --config $cfg --steps 50 --warmup 10 > gpurun_out/bench_$cfg.log 2>&1 || { echo "bench $cfg failed"; tail -5 gpurun_out/bench_$cfg.log; exit 1; }
  tail -1 gpurun_out/bench_$cfg.log | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
for cfg in dcgan64 classifier; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$cfg
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --graph off --steps 10 --warmup 4 > $GRAFT_REPO_ROOT/gpurun_out/prof_$cfg.log 2>&1 || { echo "prof $cfg failed"; exit 1; }
done
echo done
