# Round-3 evidence on the GPU box: kernel tables (eager kernel trace, steady state), a --force-dp timeline with the
# xGMI all-reduce kernel on its own queue, and hardware-counter tables.  Everything lands in gpurun_out/; copy what
# is to be kept into profiles/.
#   usage: r3_evidence.sh [traces] [dp] [pmc64] [pmccls] [pmc28]
set -o pipefail
R=$GRAFT_REPO_ROOT
for what in "$@"; do
  case $what in
    traces)
      bash $R/tools/prof_cfg.sh dcgan28 gan || exit 1
      bash $R/tools/prof_cfg.sh classifier || exit 1
      bash $R/tools/prof_cfg.sh dcgan64 gan || exit 1 ;;
    dp)
      cd /tmp && export TMPDIR=/tmp
      rm -rf $R/gpurun_out/prof_dp
      timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_dp -- python3 $R/bench.py --force-dp --comm p2p --no-probe --dp-variant default --graph full --steps 10 --warmup 4 > $R/gpurun_out/prof_dp.log 2>&1 || { tail -5 $R/gpurun_out/prof_dp.log; exit 1; }
      python3 $R/tools/rocprof_summary.py $R/gpurun_out/prof_dp --timeline --gan > $R/gpurun_out/prof_dp_timeline.md || exit 1
      tail -1 $R/gpurun_out/prof_dp_timeline.md ;;
    pmc64) bash $R/tools/pmc_step.sh dcgan64 450 || exit 1 ;;
    pmccls) bash $R/tools/pmc_step.sh classifier 639 || exit 1 ;;
    pmc28) bash $R/tools/pmc_step.sh dcgan28 207 || exit 1 ;;
    *) echo "unknown $what"; exit 1 ;;
  esac
done
echo evidence done
