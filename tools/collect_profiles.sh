#!/bin/bash
# Copy what tools/final_visit.sh <tag> measured (gpurun_out/<tag>{,_c3,_c5,_vits}/) into the tracked summaries profiles/<name>_*:
#   bash tools/collect_profiles.sh r5_final r5
TAG=${1:-r5_final}; NAME=${2:-r5}
cd "$(dirname "$0")/.."
for c in "" _c3 _c5 _vits; do
  [ -d gpurun_out/$TAG$c/prof ] && python tools/summarize_prof.py gpurun_out/$TAG$c profiles/$NAME$c > /dev/null
  [ -f gpurun_out/$TAG$c/roofline_table.txt ] && cp gpurun_out/$TAG$c/roofline_table.txt profiles/$NAME${c}_roofline_table.txt
  [ -f gpurun_out/$TAG$c/sq_summary.txt ] && cp gpurun_out/$TAG$c/sq_summary.txt profiles/$NAME${c}_sq_counters_in_forward.txt
  [ -f gpurun_out/$TAG$c/commit.txt ] && echo "profiles ${NAME}${c}_*: measured on a snapshot of tree $(head -1 gpurun_out/$TAG$c/commit.txt) $(sed -n 2p gpurun_out/$TAG$c/commit.txt)" > profiles/$NAME${c}_tree.txt
done
# the box's scratch path is not part of the command
sed -i '1s|python [^ ]*/bench.py --steps 5 --warmup 2 --no-cpu-baseline|config 2: HiFi-GAN V1, B = 64 x 80 x 256 (python bench.py --steps 5 --warmup 2 --no-cpu-baseline)|' profiles/${NAME}_roofline_table.txt
sort -u gpurun_out/$TAG/manifest.tsv > profiles/${NAME}_launch_manifest.tsv
{ grep -E "passed|failed" gpurun_out/$TAG/pytest_gpu.txt | tail -1; echo "tree $(head -1 gpurun_out/$TAG/commit.txt)"; } > profiles/${NAME}_pytest_gpu.txt
cp gpurun_out/$TAG/bench.json profiles/${NAME}_bench.json
ls -la profiles/${NAME}_* | head -40
