# Round-3 GPU visit: gpu tests (optionally a -k filter in $1), then the bench records named in $2.. (default: the quick set).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
filt="$1"; shift
if [ "$filt" != "skip" ]; then
  if [ -n "$filt" ] && [ "$filt" != "all" ]; then
    timeout -k 10 900 python -u -m pytest tests -m gpu -v -x --timeout 150 --timeout-method thread -k "$filt" > gpurun_out/gpu_tests.log 2>&1; rc=$?
  else
    timeout -k 10 900 python -u -m pytest tests -m gpu -v -x --timeout 150 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?
  fi
  echo "tests_rc=$rc"; tail -15 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
fi
out=gpurun_out/r3_bench_records.md
rec() {  # name args...
  n=$1; shift
  timeout -k 10 500 python bench.py "$@" > gpurun_out/rec_$n.log 2> gpurun_out/rec_$n.err || { echo "bench $n failed"; tail -5 gpurun_out/rec_$n.err; exit 1; }
  { echo "## $n: \`python bench.py $*\`"; echo '```'; tail -1 gpurun_out/rec_$n.log; echo '```'; } >> $out
  echo "$n: $(tail -1 gpurun_out/rec_$n.log | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"') $(tail -1 gpurun_out/rec_$n.log | grep -o '"ms_per_step": [0-9.]*') $(grep -c 'falling back' gpurun_out/rec_$n.err) fallbacks"
}
for what in "$@"; do
  case $what in
    dcgan28) rec dcgan28 --steps 200 --warmup 20 ;;
    forcedp) rec dcgan28_forcedp --steps 200 --warmup 20 --force-dp ;;
    forcedp_p2p) rec dcgan28_forcedp_p2p --steps 200 --warmup 20 --force-dp --comm p2p ;;
    classifier) rec classifier --config classifier --steps 50 --warmup 10 ;;
    classifier_serial) DDL_CLS_CONCURRENT=0 rec classifier_serial --config classifier --steps 50 --warmup 10 ;;
    classifier_eager) rec classifier_eager --config classifier --steps 50 --warmup 10 --graph off ;;
    localsgd) rec localsgd --config localsgd --steps 50 --warmup 10 ;;
    dcgan64) rec dcgan64 --config dcgan64 --steps 10 --warmup 3 ;;
    ens_ddl_ray) rec ens_ddl_ray --config ensemble32 --preset ddl_ray --steps 20 --warmup 5 ;;
    ens_ddl_normal3) rec ens_ddl_normal3 --config ensemble32 --preset ddl_normal3 --steps 20 --warmup 5 ;;
    ens_ddl) rec ens_ddl --config ensemble32 --preset ddl --steps 10 --warmup 3 ;;
    *) echo "unknown record $what"; exit 1 ;;
  esac
done
echo done
