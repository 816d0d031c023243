cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_wg -- python3 $R/tools/bench_wgrad.py > $R/gpurun_out/prof_wg.log 2>&1 || exit 1
echo ok
