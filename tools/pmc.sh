cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in G4; do
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc2_$c -- python3 $R/tools/one_conv.py $c > $R/gpurun_out/pmc2_$c.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc2i_$c -- python3 $R/tools/one_conv.py $c igemm > $R/gpurun_out/pmc2i_$c.log 2>&1 || exit 1
done
echo ok
