DDL_AUTOTUNE_VERBOSE=1 timeout -k 10 100 python tools/bench_conv.py 2>&1 | grep -v "^\[autotune\]" | tail -15
DDL_AUTOTUNE_VERBOSE=1 timeout -k 10 100 python tools/bench_conv.py 2>&1 | grep "^\[autotune\]" | cut -c1-260
