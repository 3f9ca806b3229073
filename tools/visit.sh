export CMD='
for n in 0 2 4 8; do echo "## AMP_RB_STAGGER=$n"; AMP_RB_STAGGER=$n timeout 200 python tools/rb_inforward.py --steps 20 --rounds 2 --modes 1 | cut -c1-40 | grep -v "^0,\|^1,[37]\|^1,11\|stage\|mode,"; done
'
bash tools/gpu_round.sh r3_stag cmd
