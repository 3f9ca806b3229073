"""Where the torch-side copies and small kernels of one VITS text -> wave call come from: every aten op torch itself dispatches
during SynthesizerTrn.infer (the HIP entry points go through ctypes and are not seen), grouped by op and by the first call site
inside amphion_amd/.  Used to hunt the ~200 __amd_rocclr_copyBuffer launches per call the kernel trace shows.

    python tools/vits_copy_trace.py            (needs the GPU)
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        site = "?"
        for fr in reversed(traceback.extract_stack()):
            if "amphion_amd" in fr.filename:
                site = f"{os.path.relpath(fr.filename)}:{fr.lineno}"
                break
        self.sites[(str(func), site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    import bench_configs as bc
    from amphion_amd.models.tts.vits.vits import SynthesizerTrn
    V1 = bc.V1
    full = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3, p_dropout=0.1,
                resblock="1", resblock_kernel_sizes=V1["resblock_kernel_sizes"], resblock_dilation_sizes=V1["resblock_dilation_sizes"],
                upsample_rates=V1["upsample_rates"], upsample_initial_channel=512, upsample_kernel_sizes=V1["upsample_kernel_sizes"],
                n_speakers=0, gin_channels=256, use_sdp=True)
    net = bc.randomize_(SynthesizerTrn(512, 513, 32, **full), 77, g_gain=0.5).to(bc.DEV).eval()
    B, Tx = 16, 100
    g = torch.Generator().manual_seed(13)
    x = torch.randint(0, 512, (B, Tx), generator=g).to(bc.DEV)
    xl = torch.full((B,), Tx)
    n_dp = torch.randn(B, 2, Tx, generator=g).to(bc.DEV)
    kw = dict(noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8, noise_dp=n_dp)
    net.infer(x, xl, **kw)
    torch.cuda.synchronize()
    with Trace() as t:
        net.infer(x, xl, **kw)
    torch.cuda.synchronize()
    by_op = collections.Counter()
    for (op, site), n in t.sites.items():
        by_op[op] += n
    print("== aten ops of one infer call ==")
    for op, n in by_op.most_common():
        print(f"{n:5d}  {op}")
    print("== call sites of the device-side ones ==")
    skip = ("aten.view", "aten.detach", "aten.slice", "aten.unsqueeze", "aten.squeeze", "aten.select", "aten.split", "aten.reshape",
            "aten._unsafe_view", "aten.alias", "aten.t.", "aten.transpose", "aten.expand", "aten.permute", "aten.as_strided")
    for (op, site), n in sorted(t.sites.items(), key=lambda kv: -kv[1]):
        if not op.startswith(skip):
            print(f"{n:5d}  {op:40s} {site}")


if __name__ == "__main__":
    main()
