for f in 0 1 0 1; do echo "COLD=$f"; for cfg in dcgan28 classifier; do DDL_AUTOTUNE_COLD=$f timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 10 2>&1 | tail -1 | cut -c12-60,120-175 || exit 1; done; done
echo "COLD=1 dcgan64"; DDL_AUTOTUNE_COLD=1 timeout -k 10 300 python bench.py --config dcgan64 --steps 60 --warmup 10 2>&1 | tail -1 | cut -c12-60,120-175
