#!/bin/bash
( timeout 600 python -m pytest tests/test_gpu_bigvgan.py -q -s 2>&1 | grep "hip-oracle\|passed\|failed" | tail -6 )
( timeout 300 python tools/bench_configs.py --only c3 2>/dev/null )
