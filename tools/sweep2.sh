for v in "256 8" "512 8" "512 4" "1024 4"; do set -- $v; echo "cap=$1 rows=$2"
for cfg in dcgan28 classifier; do
DDL_CR_SMALL_CAP=$1 DDL_CR_SMALL_ROWS=$2 timeout -k 10 200 python bench.py --config $cfg --steps 100 --warmup 10 2>&1 | tail -1 | cut -c12-140 || exit 1
done; done
