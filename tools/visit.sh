bash tools/gpu_round.sh r3_final2 tests bench smoke
