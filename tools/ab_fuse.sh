for f in 0 1 0 1; do echo "FUSE=$f"; DDL_FUSE_ACT_BWD=$f timeout -k 10 200 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | cut -c12-60,120-175 || exit 1; done
