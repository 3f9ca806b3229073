"""Builds libamphion_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m amphion_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to amphion_amd/csrc/_build/, the library to
amphion_amd/lib/libamphion_hip.so (git-ignored, shipped to the GPU box with the snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "lib", "libamphion_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# experiment builds (same-box A/B through AMP_LIB_PATH, see _lib.py): AMP_BUILD_TAG=<tag> AMP_BUILD_FLAGS="-DX ..." write
# lib/libamphion_hip_<tag>.so from objects under csrc/_build_<tag>/
TAG = os.environ.get("AMP_BUILD_TAG", "")
if TAG:
    OBJ = os.path.join(CSRC, "_build_" + TAG)
    LIB = os.path.join(HERE, "lib", "libamphion_hip_" + TAG + ".so")
ARCH = "gfx950"
CONV_TAPS = (1, 2, 3, 5, 7, 11)
PAIR_TAPS = (3, 5, 7, 11)
SMALL_TAPS = (1, 3, 5, 7, 11)
BLK_TAPS = (2, 3, 7, 11)
RB_TAPS = (3, 5, 7, 11)
AMPB_TAPS = (3, 5, 7, 11)

FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC, "-Wall",
         "-Wno-unused-function"] + os.environ.get("AMP_BUILD_FLAGS", "").split()


def _units():
    units = [("generator.hip", "generator.o", []), ("small_kernels.hip", "small_kernels.o", []),
             ("mel.hip", "mel.o", []), ("vits_text.hip", "vits_text.o", []), ("pair3_f16x3.hip", "pair3_f16x3.o", []),
             ("conv_small3_f16x3.hip", "conv_small3_f16x3.o", [])]
    for kt in CONV_TAPS:
        units.append(("conv_mfma.hip", f"conv_mfma_kt{kt}.o", [f"-DAMP_KT={kt}"]))
        units.append(("conv_f16x3.hip", f"conv_f16x3_kt{kt}.o", [f"-DAMP_KT={kt}"]))
    for kt in BLK_TAPS:
        units.append(("conv_blk_f16x3.hip", f"conv_blk_f16x3_kt{kt}.o", [f"-DAMP_KT={kt}"]))
    for kt in SMALL_TAPS:
        units.append(("conv_small_f16x3.hip", f"conv_small_f16x3_kt{kt}.o", [f"-DAMP_KT={kt}"]))
    for kt in PAIR_TAPS:
        units.append(("pair_f16x3.hip", f"pair_f16x3_kt{kt}.o", [f"-DAMP_KT={kt}"]))
        units.append(("pair_strip_f16x3.hip", f"pair_strip_f16x3_kt{kt}.o", [f"-DAMP_KT={kt}"]))
    for kt in RB_TAPS:
        units.append(("rb_f16x3.hip", f"rb_f16x3_kt{kt}.o", [f"-DAMP_KT={kt}"]))
    for kt in AMPB_TAPS:
        units.append(("ampb_f16x3.hip", f"ampb_f16x3_kt{kt}.o", [f"-DAMP_KT={kt}"]))
    return units


def _deps_mtime():
    m = 0.0
    for root in (CSRC, INCLUDE):
        for fn in os.listdir(root):
            if fn.endswith((".h", ".hip", ".cpp")):
                m = max(m, os.path.getmtime(os.path.join(root, fn)))
    return max(m, os.path.getmtime(__file__))


def _compile(unit):
    src, obj, extra = unit
    cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", os.path.join(OBJ, obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    newest = _deps_mtime()
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    units = _units()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, units))
    cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + [os.path.join(OBJ, o) for o in objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
