# usage: prof_cfg.sh <config> [gan]  -> gpurun_out/prof_<config>_{steady,seq}.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; C=$1; G=${2:+--gan}
rm -rf $R/gpurun_out/prof_$C
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$C -- python3 $R/bench.py --config $C --graph off --steps 10 --warmup 4 > $R/gpurun_out/prof_$C.log 2>&1 || { tail -5 $R/gpurun_out/prof_$C.log; exit 1; }
python3 $R/tools/rocprof_summary.py $R/gpurun_out/prof_$C --steady $G > $R/gpurun_out/prof_${C}_steady.md || exit 1
python3 $R/tools/rocprof_summary.py $R/gpurun_out/prof_$C --sequence $G > $R/gpurun_out/prof_${C}_seq.md || exit 1
echo ok $C
