for kb in 40 64 80 150; do echo "col2im lds kb=$kb"
for cfg in dcgan28 dcgan64; do
DDL_COL2IM_LDS_KB=$kb timeout -k 10 200 python bench.py --config $cfg --steps 60 --warmup 10 2>&1 | tail -1 | cut -c12-60,120-200 || exit 1
done; done
