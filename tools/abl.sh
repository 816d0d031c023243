for d in 0 8; do echo "dbg=$d"; DDL_PCONV_DBG=$d timeout -k 10 60 python tools/bench_conv.py 2>&1 | grep -E "D3 dgrad|G4 convT|G6 convT|G4 bwd|G6 bwd|D1" | cut -c1-120; done
timeout -k 10 200 python -u -m pytest tests/test_hip_ops.py tests/test_network_gpu.py -m gpu -q -x --timeout 120 --timeout-method thread 2>&1 | tail -2
