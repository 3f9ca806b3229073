#!/usr/bin/env python
"""Per-kernel roofline table of the config-2 forward (HiFi-GAN V1, B = 64 x 80 x 256; --config c3: BigVGAN-base B = 32 x 100 x 256;
--config c5: the VITS decode path, B = 16) from the committed profile set:
profiles/<name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats over bench.py) and <name>_hbm_traffic.csv (--pmc
FETCH_SIZE / WRITE_SIZE passes).  Algorithmic FLOPs per launch come from the kernel's template arguments and the layer
shapes of SURVEY.md Appendix B; peaks from MI355X_MICROARCH.md (f16 MFMA 2516.6 TFLOP/s / 3 MFMAs per term, HBM 8 TB/s).

    python tools/roofline_table.py [profiles/r3] [--config c2|c3|c5] > profiles/r3_roofline_table.txt
"""
import csv
import re
import sys

CONFIG = "c2"
for i, a in enumerate(sys.argv):
    if a == "--config":
        CONFIG = sys.argv[i + 1]
        del sys.argv[i:i + 2]
        break
B, PEAK_TF, PEAK_GBS = {"c2": 64, "c3": 32, "c5": 16}[CONFIG], 2516.6 / 3.0, 8000.0
STAGE = {256: 2048, 128: 16384, 64: 32768, 32: 65536}          # channels -> samples per item at that stage
TENSOR_MB = lambda C: B * C * STAGE[C] * 4 / 1e6
TITLE = {"c2": "config 2 (HiFi-GAN V1, B = 64 x 80 x 256)", "c3": "config 3 (BigVGAN-base 24 kHz, B = 32 x 100 x 256)",
         "c5": "config 5 (VITS enc_q -> flow -> flow^-1 -> HiFi-GAN decoder, B = 16 x 513 x 256)"}[CONFIG]


def shape_c3(name):
    """BigVGAN-base: unfused AMPBlock convs by width (conv_f16x3_kernel<k, WM, WN>: WM 4 -> C = 128, 2 -> C = 64, 1 -> C = 32; the
    row-blocked kernel is the C = 256 stage), anti-aliased activations, whole-AMPBlock launches"""
    m = re.search(r"ampb_f16x3_kernel<(\d+), (\d+), (\d+)", name)
    if m:
        k, wm, wn = int(m.group(1)), int(m.group(2)), int(m.group(3))
        C = 32 * wm
        return f"whole AMPBlock C={C} k={k} ({wm * wn} waves): 6 convs + 6 Activation1d", 6 * 2.0 * C * C * k * STAGE[C] * B / 1e9, 3 * TENSOR_MB(C)
    m = re.search(r"act1d_kernel<(\d+)>", name)
    if m:
        n = int(m.group(1))
        C = 256 if n == 2 else 128
        lab = "Activation1d, stage 0 rows (T = 2048, all edge tiles)" if n == 2 else "Activation1d, 268-MB tensors (stages 1-3, post)"
        return lab, 0.0, 2 * TENSOR_MB(C)
    m = re.search(r"conv_f16x3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:
        k, wm, ni = int(m.group(1)), int(m.group(2)), int(m.group(4))
        if k == 2 or ni != 4:
            return None
        C = {4: 128, 2: 64, 1: 32}[wm]
        return f"conv C={C} k={k} (inside an AMPBlock, unfused; half of them + residual)", 2.0 * C * C * k * STAGE[C] * B / 1e9, 2.5 * TENSOR_MB(C)
    m = re.search(r"conv_blk_kernel<(\d+), ", name)
    if m and int(m.group(1)) in (3, 7, 11):
        k, C = int(m.group(1)), 256
        return f"conv C=256 k={k} (stage 0, row-blocked)", 2.0 * C * C * k * STAGE[C] * B / 1e9, 2.5 * TENSOR_MB(C)
    if "conv_post_stream" in name:
        return "conv_post C=32 -> 1, k=7 + tanh", 2.0 * 32 * 7 * 65536 * B / 1e9, TENSOR_MB(32) + B * 65536 * 4 / 1e6
    return None


def shape(name):
    """-> (label, algorithmic GFLOP per launch, algorithmic MB per launch (read x + write y)) or None"""
    if CONFIG == "c3":
        return shape_c3(name)
    m = re.search(r"pair_(?:strip|f16x3)_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, \d+)*>", name)
    if m:
        k, wm, wn = int(m.group(1)), int(m.group(2)), int(m.group(3))
        mi = int(m.group(6) or 1) if "strip" in name else 1
        C = 32 * wm * mi
        if "strip" in name and wm == 8:
            C = 256
        return f"fused pair C={C} k={k}", 2 * 2.0 * C * C * k * STAGE[C] * B / 1e9, 2 * TENSOR_MB(C)
    m = re.search(r"rb_f16x3_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if m:   # whole ResBlock1 (rb_f16x3.hip): three pairs = six convs per launch, x read once and y written once
        k, wm, wn = int(m.group(1)), int(m.group(2)), int(m.group(3))
        C = 32 * wm
        form = "8 waves" if wm * wn == 8 else "4 waves"
        return f"whole resblock C={C} k={k} ({form})", 6 * 2.0 * C * C * k * STAGE[C] * B / 1e9, 2 * TENSOR_MB(C)
    m = re.search(r"conv_f16x3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:
        k, wm = int(m.group(1)), int(m.group(2))
        if k == 2:
            return None                                              # transposed convs: several shapes share the template
        if wm == 4 and k in (3, 7, 11):
            C = 256                                                  # the unfused C = 256 stage (+ conv_pre in the k = 7 line)
            note = " (+ conv_pre's launches in the average)" if k == 7 else ""
            return f"conv C={C} k={k} (stage 0){note}", 2.0 * C * C * k * STAGE[C] * B / 1e9, 3 * TENSOR_MB(C)
    m = re.search(r"conv_blk_kernel<(\d+), ", name)
    if m:   # row-blocked kernel (conv_blk_f16x3.hip)
        k = int(m.group(1))
        if k in (3, 7, 11):
            C = 256
            return f"conv C=256 k={k} (stage 0, row-blocked)", 2.0 * C * C * k * STAGE[C] * B / 1e9, 3 * TENSOR_MB(C)
        # the two stride-8 transposed convs share the template: 512 -> 256 (T 256 -> 2048) and 256 -> 128 (2048 -> 16384);
        # FLOPs and bytes of the average launch
        gf = (2.0 * 512 * 256 * 16 * 256 * B + 2.0 * 256 * 128 * 16 * 2048 * B) / 2 / 1e9
        mb = (B * 512 * 256 * 4 / 1e6 + TENSOR_MB(256) + TENSOR_MB(256) + TENSOR_MB(128)) / 2
        return "ConvT k=16 s=8, 512->256 / 256->128 (avg launch, row-blocked)", gf, mb
    if "conv_post_stream" in name:
        return "conv_post C=32 -> 1, k=7 + tanh", 2.0 * 32 * 7 * 65536 * B / 1e9, TENSOR_MB(32) + B * 65536 * 4 / 1e6
    return None


def main():
    base = sys.argv[1] if len(sys.argv) > 1 else "profiles/r2"
    stats = {r["Name"]: r for r in csv.DictReader(open(base + "_kernel_stats.csv"))}
    traffic = {}
    import os
    if os.path.exists(base + "_hbm_traffic.csv"):      # optional: a visit without PMC passes lists the algorithmic bytes only
        for r in csv.DictReader(l for l in open(base + "_hbm_traffic.csv") if not l.startswith("#") and l.strip()):
            traffic[r["kernel"]] = float(r["total_MB_corrected"])
    print(f"# per-kernel roofline, {TITLE}, from {base}_kernel_stats.csv / _hbm_traffic.csv")
    print("# peak: f16x3 MFMA %.1f TFLOP/s (2516.6 / 3), HBM %.0f GB/s; 'alg MB' = read x + write y (+ residual for unfused convs)" % (PEAK_TF, PEAK_GBS))
    print("%-64s %6s %9s %9s %8s %6s %9s %9s %8s %6s" % ("kernel", "calls", "avg us", "GFLOP", "TFLOP/s", "frac", "alg MB", "PMC MB", "GB/s", "frac"))
    tot_us = 0.0
    blk7 = any("conv_blk_kernel<7," in n for n in stats)     # then the pipelined k = 7 row is conv_pre alone
    for name, r in stats.items():
        sh = shape(name)
        if blk7 and sh and sh[0].startswith("conv C=256 k=7 (stage 0) "):
            sh = ("conv_pre 80 -> 512, k=7 (T = 256)", 2.0 * 80 * 512 * 7 * 256 * B / 1e9, B * (80 + 512) * 256 * 4 / 1e6)
        us = float(r["AverageNs"]) / 1e3
        if sh is None:
            if CONFIG == "c2" or us * int(r["Calls"]) < 20.0 or name.startswith("__amd"):
                continue
            short = re.sub(r"\(.*", "", name.replace("void amp::", ""))[:62]
            sh = (short, 0.0, 0.0)             # frame-rate / element-wise kernels: time and counter bytes only
        label, gflop, mb = sh
        pmc = traffic.get(name)
        tf = gflop / (us * 1e-6) / 1e3
        gbs = (pmc if pmc else mb) / 1e3 / (us * 1e-6)
        if mb == 0.0 and not pmc:
            gbs = 0.0
        print("%-64s %6s %9.1f %9.1f %8.1f %6.3f %9.0f %9s %8.0f %6.3f" % (label, r["Calls"], us, gflop, tf, tf / PEAK_TF, mb,
                                                                           ("%.0f" % pmc) if pmc else "-", gbs, gbs / PEAK_GBS))
        tot_us += us * int(r["Calls"])
    print("# the launches listed cover %.1f ms of kernel time in the profiled run" % (tot_us / 1e3))


if __name__ == "__main__":
    main()
