# One bench.py record per config / mode -> gpurun_out/bench_records.md (copied to profiles/r2_bench_records.md).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/bench_records.md
echo "# Builder-run bench records, one MI355X (gpurun), round 2: one JSON line per \`python bench.py ...\` invocation named in the heading" > $out
rec() {  # name args...
  n=$1; shift
  timeout -k 10 400 python bench.py "$@" > gpurun_out/rec_$n.log 2>&1 || { echo "bench $n failed"; tail -5 gpurun_out/rec_$n.log; exit 1; }
  { echo "## $n: \`python bench.py $*\`"; echo '```'; tail -1 gpurun_out/rec_$n.log; echo '```'; } >> $out
  echo "$n: $(tail -1 gpurun_out/rec_$n.log | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"') $(tail -1 gpurun_out/rec_$n.log | grep -o '"ms_per_step": [0-9.]*')"
}
rec dcgan28 --steps 200 --warmup 20
rec dcgan28_forcedp --steps 200 --warmup 20 --force-dp
rec dcgan28_forcedp_p2p --steps 200 --warmup 20 --force-dp --comm p2p
rec classifier --config classifier --steps 50 --warmup 10
rec localsgd --config localsgd --steps 50 --warmup 10
rec dcgan64 --config dcgan64 --steps 10 --warmup 3
rec ens_ddl_ray --config ensemble32 --preset ddl_ray --steps 20 --warmup 5
rec ens_ddl_normal3 --config ensemble32 --preset ddl_normal3 --steps 20 --warmup 5
rec ens_ddl --config ensemble32 --preset ddl --steps 10 --warmup 3
echo done
