cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 100 python3 $R/tools/one_ew.py 784000 32 || exit 1
timeout -k 10 100 python3 $R/tools/one_ew.py 1663488 128 || exit 1
rm -rf $R/gpurun_out/pmc_ew1 $R/gpurun_out/pmc_ew2
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_ew1 -- python3 $R/tools/one_ew.py 784000 32 > $R/gpurun_out/pmc_ew1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_ew2 -- python3 $R/tools/one_ew.py 784000 32 > $R/gpurun_out/pmc_ew2.log 2>&1 || exit 1
echo ok
