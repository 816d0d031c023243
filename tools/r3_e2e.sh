# End-to-end ensemble runs on the GPU, full reference schedules, synthetic MNIST-shaped data -> gpurun_out/e2e/*.jsonl
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/e2e
run() {  # name timeout script args...
  n=$1; t=$2; shift 2
  echo "== $n: python $*"
  t0=$(date +%s)
  timeout -k 10 $t python -u "$@" --synthetic --no-plot --log-jsonl gpurun_out/e2e/$n.jsonl > gpurun_out/e2e/$n.out 2> gpurun_out/e2e/$n.err
  rc=$?
  echo "$n wall $(( $(date +%s) - t0 )) s"
  tail -2 gpurun_out/e2e/$n.err | cut -c1-300
  grep -E "model accuracy|ensemble accuracy|done in" gpurun_out/e2e/$n.out | cut -c1-200
  [ $rc -eq 0 ] || { echo "$n failed rc=$rc"; tail -5 gpurun_out/e2e/$n.err; exit 1; }
}
for what in "$@"; do
  case $what in
    ray3) run ddl_ray3 300 entrypoints/DDL_Ray3.py --workers 1 ;;
    ray) run ddl_ray 300 entrypoints/DDL_Ray.py --workers 1 ;;
    ray2) run ddl_ray2 300 entrypoints/DDL_Ray2.py --workers 1 ;;
    normal3) run ddl_normal3 1000 entrypoints/DDL_Normal3.py ;;
    ddl) run ddl 1100 entrypoints/DDL.py ;;
    *) echo "unknown $what"; exit 1 ;;
  esac
done
echo done
