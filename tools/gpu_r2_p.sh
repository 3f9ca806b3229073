#!/bin/bash
# Round 2, visit P: measured pair policy as the default -- whole suite, bench with PMC traffic, other configs
bash tools/gpu_round.sh r2_p
