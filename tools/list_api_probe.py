#!/usr/bin/env python
"""Where the time of one synthesis_audios(ragged=False) call goes, call by call (the steps of
amphion_amd/models/vocoders/gan/gan_vocoder_inference.py re-enacted with a synchronisation after each): host padding, H2D,
generator forward (GPU events), range check, D2H into the pinned staging buffer, per-item crops.  Tuning aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc
from amphion_amd.utils.synthetic import synthetic_mel
from amphion_amd.utils.util import pad_mels_to_tensors
from amphion_amd.models.vocoders.gan import gan_vocoder_inference as gvi


def elim():
    """Leave one step of the sequence out at a time: which one brings the alternating +21 ms?"""
    cfg, m = bc.hifigan()
    lens = torch.randint(60, 400, (64,), generator=torch.Generator().manual_seed(3)).tolist()
    mels = [synthetic_mel(1, 80, L, seed=i)[0] for i, L in enumerate(lens)]
    dev = torch.device("cuda:0")
    with torch.no_grad():
        gvi.synthesis_audios(cfg, m, mels, batch_size=64); torch.cuda.synchronize()
        m.set_profiling(1)
        mb, mf = pad_mels_to_tensors([x.cpu() for x in mels], 64)
        fixed_dev = mb[0].to(dev)
        rf = m.receptive_frames(); T = int(mb[0].shape[-1])
        ext = [min(T, int(f) + rf) for f in mf[0]]
        for skip in ("nothing", "pad", "h2d", "d2h", "crops", "d2h+crops", "pad+h2d", "crops->views"):
            rows = []
            for call in range(10):
                t0 = time.perf_counter()
                if "pad" not in skip:
                    mb, mf = pad_mels_to_tensors([x.cpu() for x in mels], 64)
                mel_dev = fixed_dev if "h2d" in skip else mb[0].to(dev)
                out = m.forward_ragged(mel_dev, ext)
                m.check_range()
                t1 = time.perf_counter()
                o2 = out.squeeze(1)
                if "d2h" not in skip:
                    n = o2.numel(); host = m._amp_host_staging[:n].view(o2.shape); host.copy_(o2, non_blocking=True); torch.cuda.current_stream().synchronize()
                    if skip == "crops->views":
                        crops = [host[i, : int(f) * 256] for i, f in enumerate(mf[0])]
                    elif "crops" not in skip:
                        crops = [host[i, : int(f) * 256].clone() for i, f in enumerate(mf[0])]
                rows.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
            print(f"skip {skip:12s}: fwd+check " + " ".join(f"{a:5.1f}" for a, _ in rows) + "  | total " + " ".join(f"{b:5.1f}" for _, b in rows[:4]), flush=True)
        m.set_profiling(0)


def main():
    if "--elim" in sys.argv:
        return elim()
    cfg, m = bc.hifigan()
    lens = torch.randint(60, 400, (64,), generator=torch.Generator().manual_seed(3)).tolist()
    mels = [synthetic_mel(1, 80, L, seed=i)[0] for i, L in enumerate(lens)]
    dev = torch.device("cuda:0")
    with torch.no_grad():
        gvi.synthesis_audios(cfg, m, mels, batch_size=64); torch.cuda.synchronize()
        m.set_profiling(1)
        print("call,pad_ms,h2d_ms,forward_launch_ms,forward_gpu_ms,forward_sync_ms,check_ms,d2h_ms,crop_ms,total_ms")
        for call in range(12):
            t = [time.perf_counter()]
            mel_batches, mel_frames = pad_mels_to_tensors([x.cpu() for x in mels], 64)
            t.append(time.perf_counter())
            mel_dev = mel_batches[0].to(dev); torch.cuda.synchronize(); t.append(time.perf_counter())
            rf = m.receptive_frames(); T = int(mel_batches[0].shape[-1])
            ext = [min(T, int(f) + rf) for f in mel_frames[0]]
            out = m.forward_ragged(mel_dev, ext); t.append(time.perf_counter())
            torch.cuda.synchronize(); t.append(time.perf_counter())
            m.check_range(); t.append(time.perf_counter())
            n = out.numel(); buf = m._amp_host_staging
            host = buf[:n].view(out.squeeze(1).shape); host.copy_(out.squeeze(1), non_blocking=True); torch.cuda.synchronize(); t.append(time.perf_counter())
            crops = [host[i, : int(f) * 256].clone() for i, f in enumerate(mel_frames[0])]; t.append(time.perf_counter())
            d = [(t[i + 1] - t[i]) * 1e3 for i in range(len(t) - 1)]
            print(f"{call},{d[0]:.2f},{d[1]:.2f},{d[2]:.2f},{m.last_timing_ms(0):.2f},{d[3]:.2f},{d[4]:.2f},{d[5]:.2f},{d[6]:.2f},{(t[-1] - t[0]) * 1e3:.2f}")
        m.set_profiling(0)
        # and the product function itself, call by call
        for call in range(8):
            t0 = time.perf_counter(); gvi.synthesis_audios(cfg, m, mels, batch_size=64); torch.cuda.synchronize()
            print(f"synthesis_audios call {call}: {(time.perf_counter() - t0) * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
