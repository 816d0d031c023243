timeout -k 10 120 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-200 || exit 1
timeout -k 10 200 python bench.py --config classifier --steps 30 --warmup 5 2>&1 | tail -1 | cut -c1-200 || exit 1
timeout -k 10 200 python bench.py --config dcgan64 --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-200
