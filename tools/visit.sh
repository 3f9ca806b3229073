bash tools/gpu_round.sh r3_v tests
