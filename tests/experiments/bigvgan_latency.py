"""Single-utterance latency of BigVGAN-base (B = 1, 256 frames = 2.7 s at 24 kHz) and its kernel-time breakdown helper:
run under rocprofv3 --kernel-trace --stats for the per-kernel view."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from types import SimpleNamespace as NS
import bench_configs as bc
from oracle import vocoder_oracle as vo
from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN

hp = vo.bigvgan_base_hp()
m = bc.randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).cuda().eval()
with torch.no_grad():
    for T in (256, 860):
        mel = torch.randn(1, 100, T, generator=torch.Generator().manual_seed(0)).cuda()
        ms = bc.timed(lambda: m(mel), 20)
        print(f"BigVGAN-base, ONE utterance of {T} frames ({T * 256 / 24000:.1f} s): {ms:.3f} ms = {T * 256 / 24000 / (ms * 1e-3):.0f} x real time")
