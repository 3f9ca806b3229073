export CMD='
timeout 300 python tools/sync_probe.py
echo "## HSA_ENABLE_INTERRUPT=0"; HSA_ENABLE_INTERRUPT=0 timeout 300 python tools/sync_probe.py | head -8
timeout 300 python tools/rb_inforward.py --steps 20 --rounds 2 --modes 0 1
'
bash tools/gpu_round.sh r3_h cmd
timeout 600 python -m pytest tests/test_gpu_c1_clips.py -m gpu -x -q -s 2>&1 | tail -8
