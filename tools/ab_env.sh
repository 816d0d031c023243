# A/B of environment knobs on one box: tools/ab_env.sh "<bench args>" "<VAR=val ...>" "<VAR=val ...>" ... (each variant twice, interleaved)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
args="$1"; shift
for round in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    [ "$v" = "-" ] && v=""
    timeout -k 10 300 env $v python bench.py $args > gpurun_out/ab_$i.log 2> gpurun_out/ab_$i.err || { echo "variant '$v' failed"; tail -5 gpurun_out/ab_$i.err; exit 1; }
    echo "r$round [$v] $(tail -1 gpurun_out/ab_$i.log | grep -o '"ms_per_step": [0-9.]*')"
  done
done
