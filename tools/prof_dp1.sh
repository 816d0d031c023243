cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_dp1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dp1 -- python3 $R/bench.py --force-dp --steps 20 --warmup 5 > $R/gpurun_out/prof_dp1.log 2>&1 || { tail -5 $R/gpurun_out/prof_dp1.log; exit 1; }
python3 $R/tools/rocprof_summary.py $R/gpurun_out/prof_dp1 --sequence --gan > $R/gpurun_out/prof_dp1_seq.md || exit 1
echo ok
