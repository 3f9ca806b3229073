bash tools/gpu_round.sh r3_a tests bench prof pmc sq smoke
export CMD='
echo "## C=64 ring strips (AMP_STRIP_C64=8): bitwise tests"; AMP_STRIP_C64=8 timeout 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_full_size.py tests/test_gpu_generator.py -m gpu -x -q 2>&1 | tail -4
echo "## pair_bench C=64 policy default"; timeout 120 python tools/pair_bench.py --C 64 --k 7 11 --d 1 3 5 --modes -1
echo "## pair_bench C=64 AMP_STRIP_C64=8"; AMP_STRIP_C64=8 timeout 120 python tools/pair_bench.py --C 64 --k 7 11 --d 1 3 5 --modes -1
echo "## pair_bench C=64 AMP_STRIP_C64=9"; AMP_STRIP_C64=9 timeout 120 python tools/pair_bench.py --C 64 --k 7 11 --d 1 3 5 --modes -1
echo "## list API default"; timeout 200 python tools/bench_configs.py --only list
echo "## list API AMP_STRIP_C128=6"; AMP_STRIP_C128=6 timeout 200 python tools/bench_configs.py --only list
echo "## bench line AMP_STRIP_C64=8"; AMP_STRIP_C64=8 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | cut -c1-2500
'
bash tools/gpu_round.sh r3_a_c64 cmd
