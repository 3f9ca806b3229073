#!/usr/bin/env python
"""Library baseline: the torch restatement of HiFiGAN.forward (oracle/vocoder_oracle.py, the same
op sequence as hifigan.py:203-219 incl. the per-forward weight-norm fold) run ON THE GPU through PyTorch-ROCm, i.e.
MIOpen convolutions + eager element-wise kernels, at the headline shape (HiFi-GAN V1, B=64 x 80 x 256), beside
this repo's HIP path on the same inputs.  Experiment only: prints one JSON line; nothing in the product uses it.

    python tests/experiments/library_baseline.py [--reps 3] [--batch 64]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace as NS
from oracle import synth, vocoder_oracle as vo

V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
          resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    dev = "cuda:0"
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, V1), seed=1234)
    mel = synth.synth_mel(a.batch, 80, 256, seed=0)
    sd_dev = {k: v.to(dev) for k, v in sd.items()}
    mel_dev = mel.to(dev)

    def lib():
        return vo.hifigan_forward(sd_dev, V1, mel_dev)

    with torch.no_grad():
        t0 = time.perf_counter(); y_lib = lib(); torch.cuda.synchronize(); first = time.perf_counter() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            y_lib = lib()
        e1.record(); torch.cuda.synchronize()
        ms_lib = e0.elapsed_time(e1) / a.reps

        from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
        cfg = NS(preprocess=NS(n_mel=80), model=NS(hifigan=NS(**V1)))
        m = HiFiGAN(cfg)
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        y = m(mel_dev); torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            y = m(mel_dev)
        e1.record(); torch.cuda.synchronize()
        ms_hip = e0.elapsed_time(e1) / a.reps
    n = a.batch * 256 * 256
    print(json.dumps({"config": f"HiFi-GAN V1 B={a.batch} x 80 x 256, fp32", "library_miopen_ms": ms_lib,
                      "library_first_call_s": first, "library_samples_per_s": n / ms_lib * 1e3,
                      "hip_path_ms": ms_hip, "hip_samples_per_s": n / ms_hip * 1e3, "speedup": ms_lib / ms_hip,
                      "max_abs_diff_hip_vs_library": (y - y_lib).abs().max().item(),
                      "device": torch.cuda.get_device_name(0), "torch": torch.__version__}))


if __name__ == "__main__":
    main()
