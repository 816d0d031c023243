# kernel trace of the batch-32 MLP ensemble step (one member, eager launches) + concurrency A/B without the wgrad side branch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_ens
DDL_CHUNK=1 timeout -k 10 180 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ens -- python3 $R/bench.py --config ensemble32 --preset ddl_ray --members 1 --graph off --steps 40 --warmup 5 > $R/gpurun_out/prof_ens.log 2>&1 || exit 1
cd $R
python3 tools/rocprof_summary.py gpurun_out/prof_ens > gpurun_out/prof_ens_summary.md 2>&1 || true
for knobs in "DDL_BRANCH=0 DDL_SERIAL_MEMBERS=1" "DDL_BRANCH=0"; do echo "== $knobs"; env $knobs timeout -k 10 200 python bench.py --config ensemble32 --preset ddl_ray --steps 10 --warmup 2 2>&1 | grep -o "\"value\": [0-9.]*\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo; done
