# usage: pmc_step.sh <config> [dispatches in 3 steps]  -> gpurun_out/pmc_<config>.md
# Hardware counters over whole eager training steps, one rocprofv3 run per counter group (no trace domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; C=$1
run() {  # name counters...
    n=$1; shift
    rm -rf $R/gpurun_out/pmc_${C}_$n
    timeout -s KILL 280 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${C}_$n -- python3 $R/bench.py --config $C --graph off --steps 4 --warmup 2 > $R/gpurun_out/pmc_${C}_$n.log 2>&1
}
# plans first, without counters: the three passes below then read the same choices from the autotune cache
timeout -k 10 280 python3 $R/bench.py --config $C --graph off --steps 2 --warmup 2 > $R/gpurun_out/pmc_${C}_plans.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${C}_plans.log; exit 1; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES || { tail -5 $R/gpurun_out/pmc_${C}_sq.log; exit 1; }
run rd FETCH_SIZE GRBM_GUI_ACTIVE || { tail -5 $R/gpurun_out/pmc_${C}_rd.log; exit 1; }
run wr WRITE_SIZE || { tail -5 $R/gpurun_out/pmc_${C}_wr.log; exit 1; }
python3 $R/tools/pmc_summary.py --last ${2:-285} --steps 3 $R/gpurun_out/pmc_${C}_sq $R/gpurun_out/pmc_${C}_rd $R/gpurun_out/pmc_${C}_wr > $R/gpurun_out/pmc_$C.md || exit 1
echo ok $C
