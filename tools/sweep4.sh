for kb in 40 56 76 150; do echo "tapfold lds kb=$kb"
for cfg in dcgan28 dcgan64; do
DDL_TAPFOLD_LDS_KB=$kb timeout -k 10 200 python bench.py --config $cfg --steps 60 --warmup 10 2>&1 | tail -1 | cut -c12-60,120-175 || exit 1
done; done
