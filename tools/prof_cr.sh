cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "256 8" "1024 4" "2048 2"; do set -- $v
export DDL_CR_SMALL_CAP=$1 DDL_CR_SMALL_ROWS=$2
rm -rf $R/gpurun_out/prof_cr
timeout -k 10 180 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cr -- python3 $R/bench.py --graph off --steps 20 --warmup 5 > $R/gpurun_out/prof_cr.log 2>&1 || exit 1
echo "== cap=$1 rows=$2"
python3 $R/tools/rocprof_summary.py $R/gpurun_out/prof_cr --steady --gan | grep -E "colreduce|finalize|steady" || exit 1
done
