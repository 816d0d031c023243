#!/bin/bash
# GAN learning evidence on one MI355X: cli/gan.py --synthetic for N iterations per mode; losses every 10
# iterations + diagnostics + sample sheets go to gpurun_out/gan_learn/<mode>/ (copied to profiles/ by hand).
#   tools/gan_learn_gpu.sh [iters] [modes...]
set -e
ITERS=${1:-3000}; shift || true
MODES=${@:-"split refexact mixed inference"}
ROOT=$(pwd)
export PYTHONPATH=$ROOT
for m in $MODES; do
  out=$ROOT/gpurun_out/gan_learn/$m
  rm -rf $out; mkdir -p $out; cd $out
  case $m in
    refexact) extra="--reference-exact";;
    *) extra="--bn-policy $m";;
  esac
  echo "== $m"
  timeout -k 10 300 python -u -m distributeddeeplearning_amd.cli.gan --synthetic --no-plot $extra \
    --set iterations=$ITERS --set log_period=250 --log-jsonl $out/log.jsonl --log-every 10 \
    --sample-grid $out/samples.png > $out/stdout.log 2>&1
  tail -n 3 $out/stdout.log
  cd $ROOT
done
