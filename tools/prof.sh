cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 180 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_now -- python3 $R/bench.py --graph off --steps 20 --warmup 5 > $R/gpurun_out/prof_now.log 2>&1 || exit 1
echo ok
