#!/bin/bash
# Submit one GPU-box visit: stamps the snapshot with the commit it was cut from (.commit_stamp travels with the snapshot; tools/gpu_round.sh
# and the profile tools copy it into what they write), then calls gpurun.     usage: tools/gpu.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.."
{ git rev-parse --short HEAD; git diff --quiet HEAD -- . ':!gpurun_out' || echo "+uncommitted changes: $(git diff --stat HEAD | tail -1)"; } > .commit_stamp
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
