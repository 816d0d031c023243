# usage: prof_graph.sh <config> [gan] -> gpurun_out/prof_<config>_graph_steady.md: kernel trace of the GRAPH-replayed step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; C=$1; G=${2:+--gan}
rm -rf $R/gpurun_out/profg_$C
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profg_$C -- python3 $R/bench.py --config $C --steps 20 --warmup 5 > $R/gpurun_out/profg_$C.log 2>&1 || { tail -5 $R/gpurun_out/profg_$C.log; exit 1; }
tail -1 $R/gpurun_out/profg_$C.log | cut -c1-250
python3 $R/tools/rocprof_summary.py $R/gpurun_out/profg_$C --steady $G > $R/gpurun_out/prof_${C}_graph_steady.md || exit 1
python3 $R/tools/rocprof_summary.py $R/gpurun_out/profg_$C --sequence $G > $R/gpurun_out/prof_${C}_graph_seq.md || exit 1
echo ok $C
