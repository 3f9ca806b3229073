#!/usr/bin/env python
"""Instruction mix of the kernels in a gfx950 assembly file (hipcc --cuda-device-only -S): VALU / packed / transcendental / MFMA /
LDS / VMEM counts per function (whole function, or the body of the longest loop with --loop) and the VALU issue slots they cost on
this chip (v_pk_*_f32 = 2, v_sin / v_cos / v_exp / v_log / v_rcp / v_rsq / v_sqrt = 4, everything else 1: DESIGN 3.2e).
    python tools/isa_mix.py file.s [--loop] [--serial] [name-substring]
--serial also counts the loads that are followed by `s_waitcnt vmcnt(0)` before the next load is issued: one memory round trip each (how round 5
found sixteen serial reads of the MFMA running sum in ampb_f16x3)."""
import re, sys
from collections import Counter

args = [a for a in sys.argv[1:] if not a.startswith("--")]
loop = "--loop" in sys.argv
txt = open(args[0]).read()
want = args[1] if len(args) > 1 else ""
TRANS = ("v_sin", "v_cos", "v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")
for m in re.finditer(r"^(\w+):\s*;\s*@\1\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    lines = body.splitlines()
    if loop:
        # the longest backward branch: from label .LBBx_y to "s_cbranch* .LBBx_y" / "s_branch .LBBx_y" after it
        pos = {}
        best = (0, 0, 0)
        for i, l in enumerate(lines):
            t = l.strip()
            lm = re.match(r"^(\.LBB\d+_\d+):", t)
            if lm:
                pos[lm.group(1)] = i
            bm = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
            if bm and bm.group(1) in pos and i - pos[bm.group(1)] > best[0]:
                best = (i - pos[bm.group(1)], pos[bm.group(1)], i)
        lines = lines[best[1]:best[2] + 1]
    c = Counter()
    for l in lines:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_pk_") and "_f32" in op: c["pk_f32"] += 1
        elif op.startswith(TRANS): c["trans"] += 1
        elif op.startswith("v_accvgpr"): c["accvgpr"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem" if not op.startswith("scratch_") else "scratch"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_nop"): c["nop"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    serial = ""
    if "--serial" in sys.argv:
        n, last = 0, None
        for i, l in enumerate(lines):
            t = l.strip()
            if t.startswith(("global_load", "buffer_load", "flat_load")):
                last = i
            elif t.startswith("s_waitcnt") and "vmcnt(0)" in t and last is not None:
                n += 1
                last = None
        serial = f"  loads waited for ONE AT A TIME: {n} of {c['vmem']} VMEM instructions"
    slots = c["valu"] + c["accvgpr"] + 2 * c["pk_f32"] + 4 * c["trans"]
    import subprocess
    short = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()[:90]
    print(f"{short}\n   {'loop body' if loop else 'function'}: {dict(c)}  VALU issue slots ~{slots}, MFMA pipe cycles (8-pass) ~{32 * c['mfma']}{serial}")
