"""Host-issue time and GPU time of each section of SynthesizerTrn.infer (text -> wave, config/vits.json dimensions, B = 16 x 100
tokens): tells a launch-bound section (host >= GPU) from a kernel-bound one.   python tools/vits_sections.py   (needs the GPU)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


SKIP_DEC = "--no-decoder" in sys.argv      # text side only, back to back: its weights stay hot in L2 / the Infinity Cache


def main():
    import bench_configs as bc
    from amphion_amd import _lib
    from amphion_amd.modules import hip_ops
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
    o = net.infer(x, xl, noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8, noise_dp=n_dp)
    t_y0 = int(o["mask"].sum(dim=(1, 2)).max())
    n_z = torch.randn(B, 192, t_y0, generator=g).to(bc.DEV)
    names = ["enc_p", "dp", "durations+sync", "expand+sample", "flow", "mask+dec", "range_check"]
    host = {n: 0.0 for n in names}
    gpu = {n: 0.0 for n in names}
    reps = 10
    for rep in range(reps + 2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        ts = []
        torch.cuda.synchronize()
        ev[0].record(); ts.append(time.perf_counter())
        xe, m_p, logs_p, lens = net.enc_p(x, xl)
        ev[1].record(); ts.append(time.perf_counter())
        logw = net.dp(xe, lens, g=None, reverse=True, noise_scale=0.8, noise=n_dp)
        ev[2].record(); ts.append(time.perf_counter())
        w_ceil, cum, y_lengths = hip_ops.durations(logw, lens, 1.0)
        t_y = int(y_lengths.max().item())
        ev[3].record(); ts.append(time.perf_counter())
        m_e, attn = hip_ops.expand_path(m_p, cum, lens, y_lengths, t_y, want_attn=True)
        logs_e, _ = hip_ops.expand_path(logs_p, cum, lens, y_lengths, t_y)
        z_p = hip_ops.gauss_sample(m_e, logs_e, n_z, 0.667)
        ev[4].record(); ts.append(time.perf_counter())
        z = net.flow(z_p, y_lengths, g=None, reverse=True)
        ev[5].record(); ts.append(time.perf_counter())
        zm = hip_ops.sequence_mask_(z.clone(), y_lengths)
        o = net._dec_exact(zm, None) if not SKIP_DEC else zm
        ev[6].record(); ts.append(time.perf_counter())
        _lib.range_check(x.device)
        ev[7].record(); ts.append(time.perf_counter())
        torch.cuda.synchronize()
        if rep >= 2:
            for i, n in enumerate(names):
                host[n] += (ts[i + 1] - ts[i]) * 1e3 / reps
                gpu[n] += ev[i].elapsed_time(ev[i + 1]) / reps
    print(f"{'section':18s} {'host issue ms':>14s} {'GPU span ms':>12s}")
    for n in names:
        print(f"{n:18s} {host[n]:14.3f} {gpu[n]:12.3f}")
    print(f"{'total':18s} {sum(host.values()):14.3f} {sum(gpu.values()):12.3f}")


if __name__ == "__main__":
    main()
