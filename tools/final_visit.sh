#!/bin/bash
# The measurement visit whose outputs become profiles/<tag>_* (one GPU box, ~12 min):  tools/gpu.sh 2400 'bash tools/final_visit.sh r5_final'
#   <tag>        tests, the bench line, rocprofv3 kernel stats + launch manifest, FETCH / WRITE passes, SQ passes, smoke       (config 2)
#   <tag>_c3     kernel stats + manifest, FETCH / WRITE, SQ over BigVGAN-base B = 32                                       (config 3)
#   <tag>_c5     kernel stats + manifest, FETCH / WRITE over the VITS decode path B = 16 with SERIAL resblocks             (config 5)
#   <tag>_vits   kernel stats + manifest over VITS text -> wave, serial resblocks
# then copy with tools/collect_profiles.sh <tag>.
TAG=${1:-final}
bash tools/gpu_round.sh $TAG tests bench prof pmc sq smoke > /dev/null 2>&1
export ROOFTITLE="config 3: BigVGAN-base 24 kHz, B = 32 x 100 x 256 (tools/bench_configs.py --only c3)"
export PROFCMD="python $PWD/tools/bench_configs.py --only c3 --reps 3"
export PROFCMD_PMC="python $PWD/tools/bench_configs.py --only c3 --reps 1"
export SQCMD="python $PWD/tools/bench_configs.py --only c3 --reps 1"
bash tools/gpu_round.sh ${TAG}_c3 prof pmc sq > /dev/null 2>&1
export ROOFTITLE="config 5: VITS enc_q -> flow -> flow^-1 -> HiFi-GAN decoder, B = 16, SERIAL resblocks (AMP_RB_STREAMS=0, tools/bench_configs.py --only c5)"
export PROFCMD="env AMP_RB_STREAMS=0 python $PWD/tools/bench_configs.py --only c5 --reps 3"
export PROFCMD_PMC="env AMP_RB_STREAMS=0 python $PWD/tools/bench_configs.py --only c5 --reps 1"
bash tools/gpu_round.sh ${TAG}_c5 prof pmc > /dev/null 2>&1
export ROOFTITLE="VITS text -> wave at config/vits.json dimensions, SERIAL resblocks (AMP_RB_STREAMS=0, tools/bench_configs.py --only vits)"
export PROFCMD="env AMP_RB_STREAMS=0 python $PWD/tools/bench_configs.py --only vits --reps 3"
bash tools/gpu_round.sh ${TAG}_vits prof > /dev/null 2>&1
tail -3 gpurun_out/$TAG/pytest_gpu.txt
cut -c1-400 gpurun_out/$TAG/bench.json
tail -1 gpurun_out/$TAG/smoke.txt
