cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in G4 G2; do
DDL_AUTOTUNE_VERBOSE=1 timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc3_$c -- python3 $R/tools/one_conv.py $c > $R/gpurun_out/pmc3_$c.log 2>&1 || exit 1
done
echo ok
