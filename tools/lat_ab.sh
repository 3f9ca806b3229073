mkdir -p gpurun_out/r5_lat; cd /tmp; export TMPDIR=/tmp
for L in old new old new; do
  if [ $L = old ]; then export AMP_LIB_PATH=$GRAFT_REPO_ROOT/amphion_amd/lib/libamphion_hip_old.so; else unset AMP_LIB_PATH; fi
  rm -rf /tmp/tr_$L; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$L -o lt -- python $GRAFT_REPO_ROOT/tools/latency_trace.py --run > /dev/null 2>&1
  echo "== $L"; python $GRAFT_REPO_ROOT/tools/latency_trace.py /tmp/tr_$L | grep -E "conv_small3|forward span"
done
