for args in "" "--force-dp" "--force-dp --graph full" "--force-dp --comm rccl" "--force-dp --comm rccl --graph full" "--force-dp --comm-dtype bf16" "--force-dp --zero"; do
echo "args: $args"
timeout -k 10 200 python bench.py --steps 100 --warmup 10 $args 2>&1 | tail -1 | cut -c12-60,120-175 || exit 1
done
for cfg in classifier localsgd; do echo "$cfg --force-dp"; timeout -k 10 200 python bench.py --config $cfg --steps 60 --warmup 10 --force-dp 2>&1 | tail -1 | cut -c12-60,120-175 || exit 1; done
