SQCMD="python $PWD/tools/bench_configs.py --only mel --reps 3" bash tools/gpu_round.sh r3_w_mel sq | tail -4
SQCMD="python $PWD/tools/bench_configs.py --only c3 --reps 2" bash tools/gpu_round.sh r3_w_c3 sq | tail -24
