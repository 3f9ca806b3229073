#!/bin/bash
# Timing-only knock-out builds of ampb_f16x3.hip (-DAMP_AMPB_KO=<bits>: 1 no Activation1d, 2 no tile write, 4 no conv K loop, 8 no halo
# exchange): libamphion_hip_ko<bits>.so = the regular objects + the three ampb units rebuilt with the flag.  Run HERE (no GPU needed), then
# needs the -DAMP_AMPB_KO hooks (profiles/negative_kernels/r5_ampb_knockouts.patch shows them).  On the box:  for k in 0 1 2 4 8 15; do AMP_LIB_PATH=$PWD/amphion_amd/lib/libamphion_hip_ko$k.so python tools/ampb_inforward.py --modes 1 --rounds 1; done
cd "$(dirname "$0")/.."
python -m amphion_amd.build || exit 1
B=amphion_amd/csrc/_build
for KO in ${@:-0 1 2 4 8 15}; do
  D=amphion_amd/csrc/_build_ko$KO; mkdir -p $D
  for KT in 3 5 7 11; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iamphion_amd/csrc -DAMP_KT=$KT -DAMP_AMPB_KO=$KO \
       -c amphion_amd/csrc/ampb_f16x3.hip -o $D/ampb_f16x3_kt$KT.o &
  done
  wait
  OBJS=$(ls $B/*.o | grep -v ampb_f16x3_kt)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o amphion_amd/lib/libamphion_hip_ko$KO.so $OBJS $D/ampb_f16x3_kt*.o && echo built ko$KO
done
