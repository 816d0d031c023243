# Quick GPU visit: a pytest -k filter ($1, "skip" for none), then tools/ab_env.sh-style A/B if more args follow:
#   tools/gpu_quick.sh "<-k filter>" "<bench args>" "<VAR=val>" ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
filt="$1"; shift
if [ "$filt" != "skip" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread -k "$filt" > gpurun_out/quick_tests.log 2>&1; rc=$?
  echo "tests_rc=$rc"; tail -15 gpurun_out/quick_tests.log; [ $rc -eq 0 ] || exit 1
fi
[ $# -gt 0 ] && bash tools/ab_env.sh "$@"
exit 0
