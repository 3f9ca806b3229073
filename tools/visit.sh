bash tools/gpu_round.sh r3_final tests bench prof pmc sq configs f32 smoke
