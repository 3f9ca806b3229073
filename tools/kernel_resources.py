#!/usr/bin/env python
"""Compile one .hip translation unit for gfx950 and print the per-kernel register / LDS / spill table
(hipcc -Rpass-analysis=kernel-resource-usage).   python tools/kernel_resources.py conv_f16x3.hip -DAMP_KT=3"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
if not os.path.exists(src):
    src = os.path.join(ROOT, "amphion_amd", "csrc", src)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "amphion_amd", "csrc"), "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["/usr/bin/c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print(f"{r['name'][:70]:70s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('TotalSGPRs')} spill={r.get('VGPRs Spill')} "
          f"scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
