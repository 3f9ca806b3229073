timeout 900 python -m pytest tests/test_gpu_resblock.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
export CMD='
timeout 400 python tools/rb_inforward.py --steps 20 --rounds 2 --modes 0 1
'
bash tools/gpu_round.sh r3_r cmd | cut -c1-90
