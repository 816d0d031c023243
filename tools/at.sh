DDL_AUTOTUNE_VERBOSE=1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 2>&1 | grep -E "autotune\] (fprop|dgrad)|metric" | cut -c1-330
