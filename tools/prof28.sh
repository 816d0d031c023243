# dcgan28: bench (graph replay) + steady-state kernel trace (eager launch so kernels are named per dispatch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 python3 $R/bench.py --steps 100 --warmup 10 > $R/gpurun_out/bench_dcgan28.log 2>&1 || { tail -5 $R/gpurun_out/bench_dcgan28.log; exit 1; }
tail -1 $R/gpurun_out/bench_dcgan28.log | cut -c1-220
rm -rf $R/gpurun_out/prof_dcgan28
timeout -k 10 180 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dcgan28 -- python3 $R/bench.py --graph off --steps 20 --warmup 5 > $R/gpurun_out/prof_dcgan28.log 2>&1 || exit 1
python3 $R/tools/rocprof_summary.py $R/gpurun_out/prof_dcgan28 --steady --gan > $R/gpurun_out/prof_dcgan28_steady.md || exit 1
python3 $R/tools/rocprof_summary.py $R/gpurun_out/prof_dcgan28 --sequence --gan > $R/gpurun_out/prof_dcgan28_seq.md || exit 1
echo ok
