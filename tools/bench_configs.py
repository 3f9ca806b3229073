#!/usr/bin/env python
"""Throughput of the other BASELINE.json configs on one MI355X (reported in DESIGN.md; bench.py stays the
configs[1] headline):  c3 BigVGAN-base 24 kHz B=32,  c5 VITS enc_q -> flow -> flow^-1 -> dec B=16,  mel = the
front end at B=64 x 65 536 samples (mel_large: 1 024 x 65 536),  list = the list API on ragged utterances,  lat = single-utterance
latency,  c1 = the 16 real clips of configs[0] end to end.

    python tools/bench_configs.py [--reps 5] [--only c1|c3|c5|vits|mel|mel_large|pcm|list|lat]
Product path only (amphion_amd modules + seeded random-init weights); one JSON line per config."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace as NS
from amphion_amd.utils.synthetic import randomize_, synthetic_mel

V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
          resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)
DEV = "cuda:0"


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def timed_sync(fn, reps):
    """median WALL time of one call followed by a device synchronisation: what a caller who needs the result waits (the back-to-back
    figure of timed() lets the host run ahead of the GPU, which hides a graph launch's own set-up cost)"""
    import statistics
    fn(); torch.cuda.synchronize()
    w = []
    for _ in range(max(reps, 15)):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); w.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(w)


def hifigan():
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
    cfg = NS(preprocess=NS(n_mel=80, hop_size=256, sample_rate=22050, extract_amplitude_phase=False),
             model=NS(hifigan=NS(**V1), generator="hifigan"))
    return cfg, randomize_(HiFiGAN(cfg), 1234).to(DEV).eval()


def c3(reps):
    from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
    hp = dict(V1, activation="snakebeta", snake_logscale=True)
    m = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).to(DEV).eval()
    mel = torch.randn(32, 100, 256, generator=torch.Generator().manual_seed(0)).to(DEV)
    ms = timed(lambda: m(mel), reps)
    n = 32 * 256 * 256
    return [{"config": "C3 BigVGAN-base 24 kHz, B=32 x 100 mel x 256 frames", "ms_per_step": ms, "samples_per_s": n / ms * 1e3, "x_realtime": n / ms * 1e3 / 24000}]


def c3_launch_counts():
    """kernel launches of ONE C3 forward by family, from the handle's own launch log (amp_gen_kernel_name over every resblock)"""
    from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
    hp = dict(V1, activation="snakebeta", snake_logscale=True)
    m = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).to(DEV).eval()
    mel = torch.randn(32, 100, 256, generator=torch.Generator().manual_seed(0)).to(DEV)
    m.set_profiling(1)
    m(mel); torch.cuda.synchronize()
    fused = sum(1 for i in range(4) for j in range(3) if any("ampb_f16x3" in n for n in m.kernel_names(100 + 16 * i + j, 0)))
    m.set_profiling(0)
    unfused = 12 - fused
    return {"whole_ampblock_launches": fused, "act1d_launches": 6 * unfused + 1, "conv_launches_in_ampblocks": 6 * unfused,
            "act1d_launches_round3": 73}


def c5(reps):
    from amphion_amd.models.tts.vits.vits import SynthesizerTrnDecodePath
    net = SynthesizerTrnDecodePath(513, 192, 192, "1", V1["resblock_kernel_sizes"], V1["resblock_dilation_sizes"],
                                   V1["upsample_rates"], V1["upsample_initial_channel"], V1["upsample_kernel_sizes"])
    net = randomize_(net, 4321, g_gain=0.5).to(DEV).eval()
    g = torch.Generator().manual_seed(7)
    y = torch.rand(16, 513, 256, generator=g).to(DEV); lens = torch.full((16,), 256); noise = torch.randn(16, 192, 256, generator=g).to(DEV)
    ms = timed(lambda: net.reconstruct(y, lens, noise=noise), reps)
    n = 16 * 256 * 256
    return [{"config": "C5 VITS enc_q -> flow -> flow(reverse) -> HiFi-GAN decoder, B=16, T=256", "ms_per_step": ms, "samples_per_s": n / ms * 1e3, "x_realtime": n / ms * 1e3 / 22050}]


def vits(reps):
    """Full VITS inference, text -> wave (SynthesizerTrn.infer, vits.py:320-369) at config/vits.json dimensions: B = 16 token sequences
    of 100 -> durations -> ~400 frames each -> HiFi-GAN decoder.  Host to host (infer synchronises once for the frame count and once
    for the operand-range check); the decoder's share comes from its own HIP events."""
    from amphion_amd.models.tts.vits.vits import SynthesizerTrn
    full = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3, p_dropout=0.1,
                resblock="1", resblock_kernel_sizes=V1["resblock_kernel_sizes"], resblock_dilation_sizes=V1["resblock_dilation_sizes"],
                upsample_rates=V1["upsample_rates"], upsample_initial_channel=512, upsample_kernel_sizes=V1["upsample_kernel_sizes"],
                n_speakers=0, gin_channels=256, use_sdp=True)
    net = randomize_(SynthesizerTrn(512, 513, 32, **full), 77, g_gain=0.5).to(DEV).eval()
    B, Tx = 16, 100
    g = torch.Generator().manual_seed(13)
    x = torch.randint(0, 512, (B, Tx), generator=g).to(DEV)
    xl = torch.full((B,), Tx)
    n_dp = torch.randn(B, 2, Tx, generator=g).to(DEV)
    kw = dict(noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8, noise_dp=n_dp)
    o = net.infer(x, xl, **kw)
    frames = o["mask"].sum(dim=(1, 2))
    n_z = torch.randn(B, 192, int(frames.max()), generator=g).to(DEV)
    fn = lambda: net.infer(x, xl, noise_z=n_z, **kw)
    ms = timed(fn, reps)
    net.dec.set_profiling(1)
    fn(); torch.cuda.synchronize()
    dec_ms = net.dec.last_timing_ms(0)
    net.dec.set_profiling(0)
    n = int(frames.sum().item()) * 256
    return [{"config": f"VITS text -> wave (SynthesizerTrn.infer, config/vits.json dimensions), B={B} x {Tx} tokens -> {int(frames.sum().item())} frames",
             "ms_per_step": ms, "samples_per_s": n / ms * 1e3, "x_realtime": n / ms * 1e3 / 22050, "decoder_gpu_ms": dec_ms,
             "share_outside_decoder": 1.0 - dec_ms / ms, "frames_per_item": [int(f) for f in frames.tolist()]}]


def mel(reps):
    from amphion_amd.utils.mel import mel_spectrogram_torch
    pp = NS(sample_rate=22050, n_fft=1024, win_size=1024, hop_size=256, n_mel=80, fmin=0, fmax=8000)
    wav = (torch.rand(64, 65536, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    ms = timed(lambda: mel_spectrogram_torch(wav, pp), reps)
    byts = 64 * 65536 * 4 + 64 * 80 * 256 * 4
    out = [{"config": "mel front end (reflect pad + STFT 1024/256 + mel 80 + log), B=64 x 65536 samples", "ms_per_step": ms,
            "samples_per_s": 64 * 65536 / ms * 1e3, "algorithmic_GBps": byts / ms / 1e6}]
    # the other transform lengths (one workgroup per frame: radix-2 for powers of two, mixed radix otherwise -- fallbacks, not fast paths)
    for sr, n_fft, hop, n_mel in ((44100, 2048, 512, 128), (24000, 1920, 480, 128), (16000, 512, 128, 80)):
        pq = NS(sample_rate=sr, n_fft=n_fft, win_size=n_fft, hop_size=hop, n_mel=n_mel, fmin=0, fmax=None)
        w = wav[:, : (65536 // hop) * hop]
        ms2 = timed(lambda: mel_spectrogram_torch(w, pq), max(2, reps // 2))
        b2 = 64 * w.shape[1] * 4 + 64 * n_mel * (w.shape[1] // hop) * 4
        out.append({"config": f"mel front end, n_fft {n_fft} / hop {hop} / {n_mel} mel, B=64 x {w.shape[1]} samples", "ms_per_step": ms2,
                    "samples_per_s": 64 * w.shape[1] / ms2 * 1e3, "algorithmic_GBps": b2 / ms2 / 1e6})
    # the inverse direction (STFT.inverse, utils/stft.py:183-222: irfft frames + overlap-add) at the same lengths
    from amphion_amd.utils.stft import STFT
    for n_fft, hop in ((1024, 256), (2048, 512), (1920, 480), (512, 128)):
        st = STFT(n_fft, hop, n_fft)
        F = 65536 // hop + 1
        g = torch.Generator().manual_seed(2)
        magn = (torch.rand(64, n_fft // 2 + 1, F, generator=g) * 2).to(DEV)
        ph = ((torch.rand(64, n_fft // 2 + 1, F, generator=g) * 2 - 1) * 3.14159).to(DEV)
        ms3 = timed(lambda: st.inverse(magn, ph), max(2, reps // 2))
        out.append({"config": f"inverse STFT, n_fft {n_fft} / hop {hop}, B=64 x {F} frames", "ms_per_step": ms3,
                    "samples_per_s": 64 * (F - 1) * hop / ms3 * 1e3, "algorithmic_GBps": (2 * magn.numel() * 4 + 64 * (F - 1) * hop * 4) / ms3 / 1e6})
    return out


def mel_large(reps):
    """The front end in the regime feature extraction runs it in (processors/acoustic_extractor.py:376-449: a dataset's worth of audio):
    1 024 x 65 536 samples = 268 MB in, 84 MB out -- 16 rounds of 512 workgroups instead of the one round of `mel`, so the figure is the
    kernel's sustained rate, not its start-up."""
    from amphion_amd.utils.mel import mel_spectrogram_torch
    pp = NS(sample_rate=22050, n_fft=1024, win_size=1024, hop_size=256, n_mel=80, fmin=0, fmax=8000)
    B = 1024
    wav = (torch.rand(B, 65536, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    ms = timed(lambda: mel_spectrogram_torch(wav, pp), reps)
    byts = B * 65536 * 4 + B * 80 * 256 * 4
    return [{"config": f"mel front end at dataset scale, B={B} x 65536 samples ({byts / 1e6:.0f} MB algorithmic)", "ms_per_step": ms,
             "samples_per_s": B * 65536 / ms * 1e3, "x_realtime": B * 65536 / ms * 1e3 / 22050, "algorithmic_bytes": byts,
             "algorithmic_GBps": byts / ms / 1e6, "frac_of_hbm_peak": byts / ms / 1e6 / 8000.0, "ns_per_frame": ms * 1e6 / (B * 256),
             "bound": "VALU (about 800 vector instructions per 1024-point frame, DESIGN.md §3.5), not HBM"}]


def c1(reps):
    """BASELINE.json configs[0] on the GPU: the 16 real clips (3.0-12.9 s, 103.6 s of 22.05 kHz audio; int16 PCM kept as a test
    fixture, tests/golden/golden_c1.npz) through wav -> mel front end -> HiFi-GAN V1 -> crop -> 16-bit PCM, ONE utterance per forward
    (inference.batch_size = 1, as bins/vocoder/inference.py runs it), host memory to host memory; median of `reps` passes."""
    import statistics as st
    import numpy as np
    from amphion_amd.utils import mel as M
    from amphion_amd.utils.io import wav_to_pcm16
    G = np.load(os.path.join(ROOT, "tests", "golden", "golden_c1.npz"))
    clips = [torch.from_numpy(G[f"pcm_{i}"].astype(np.float32) / 32768.0).pin_memory() for i in range(16)]
    cfg, m = hifigan()
    pp = NS(sample_rate=22050, n_fft=1024, win_size=1024, hop_size=256, n_mel=80, fmin=0, fmax=8000)

    def one_pass():
        outs = []
        for w in clips:
            wd = w.to(DEV, non_blocking=True).unsqueeze(0)
            mel = M.extract_mel_features(wd, pp)                  # [n_mel, F]
            wav = m._forward_graphed_view(mel.unsqueeze(0))   # [1, 1, F * hop]: what vocoder_inference / synthesis_audios call (cached graph per 32-frame bucket from the 2nd pass on)
            outs.append(wav_to_pcm16(wav[0, :, : mel.shape[-1] * 256]).cpu())
        return outs

    one_pass(); one_pass(); torch.cuda.synchronize()     # the second pass is where forward_graphed captures its buckets (~2 ms per bucket, once)
    wall = []
    for _ in range(max(3, reps)):
        t0 = time.perf_counter(); outs = one_pass(); torch.cuda.synchronize(); wall.append((time.perf_counter() - t0) * 1e3)
    secs = sum(int(c.numel()) for c in clips) / 22050.0
    med = st.median(wall)
    return [{"config": "C1: 16 LJSpeech-style clips, wav -> mel -> HiFi-GAN V1 -> PCM16, batch_size = 1, host to host", "ms_total": med,
             "ms_all": [round(w, 1) for w in wall], "audio_s": secs, "x_realtime": secs / (med * 1e-3), "ms_per_clip": med / 16,
             "samples_out": int(sum(o.numel() for o in outs))}]


def pcm(reps):
    from amphion_amd.utils.io import wav_to_pcm16
    wav = (torch.rand(64, 65536, generator=torch.Generator().manual_seed(2)) * 2.2 - 1.1).to(DEV)
    ms = timed(lambda: wav_to_pcm16(wav), reps)
    byts = 64 * 65536 * (4 + 2)
    return [{"config": "fp32 -> PCM16 (amp_wav_to_pcm16), B=64 x 65536 samples", "ms_per_step": ms,
             "samples_per_s": 64 * 65536 / ms * 1e3, "algorithmic_GBps": byts / ms / 1e6}]


def lst(reps):
    """The list API, host to host.  Every figure is the MEDIAN of n >= 7 calls after one warm-up, with the spread, and the
    time is split three ways: GPU time of the generator forward alone (the handle's own HIP events), GPU-stream time of the
    whole call (H2D + forward + D2H, torch events around it) and host wall time -- round 2's line was ONE call and the
    driver's box disagreed with the builder's by 28 ms without saying where."""
    import statistics as st
    from amphion_amd.models.vocoders.gan.gan_vocoder_inference import synthesis_audios
    cfg, m = hifigan()
    lens = torch.randint(60, 400, (64,), generator=torch.Generator().manual_seed(3)).tolist()
    mels = [synthetic_mel(1, 80, L, seed=i)[0] for i, L in enumerate(lens)]
    n_calls = max(7, reps)
    out = []
    for ragged in (True, False):
        synthesis_audios(cfg, m, mels, batch_size=64, ragged=ragged); torch.cuda.synchronize()
        m.set_profiling(1)
        wall, stream, fwd = [], [], []
        for _ in range(n_calls):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            synthesis_audios(cfg, m, mels, batch_size=64, ragged=ragged)
            e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
            wall.append((t1 - t0) * 1e3); stream.append(e0.elapsed_time(e1)); fwd.append(m.last_timing_ms(0))
        m.set_profiling(0)
        n = sum(lens) * 256
        med = st.median(wall)
        out.append({"config": f"synthesis_audios(ragged={ragged}): 64 utterances of 60..400 frames, HiFi-GAN V1, host list API incl. H2D/D2H",
                    "ms_total": med, "n": n_calls, "ms_min": min(wall), "ms_max": max(wall), "ms_all": [round(w, 1) for w in wall], "gpu_stream_ms_all": [round(w, 1) for w in stream],
                    "gpu_forward_ms": st.median(fwd), "gpu_stream_ms": st.median(stream), "host_only_ms": med - st.median(fwd),
                    "samples_per_s": n / med * 1e3, "x_realtime": n / med * 1e3 / 22050})
    return out


def lat(reps):
    """One utterance through HiFi-GAN V1, eager and as a hipGraph replay, with the resblocks of a stage one after the other
    (amp_set_resblock_streams(0): rounds 1-3) and on concurrent streams (the default for launches this small)."""
    from amphion_amd import _lib
    out = []
    for streams in (0, -1):
        _lib.check(_lib.lib().amp_set_resblock_streams(streams))
        cfg, m = hifigan()
        tag = "sequential resblocks" if streams == 0 else "concurrent resblocks (default)"
        for T in (256, 860):
            mel1 = synthetic_mel(1, 80, T, seed=5).to(DEV)
            ms = timed(lambda: m(mel1), 20)
            out.append({"config": f"latency: HiFi-GAN V1, ONE utterance of {T} frames ({T * 256 / 22050:.1f} s of audio), {tag}", "ms": ms,
                        "ms_call_to_sync": timed_sync(lambda: m(mel1), 20), "x_realtime": T * 256 / 22050 / (ms * 1e-3)})
            replay, static_in, _ = m.capture(1, T)
            static_in.copy_(mel1)
            ms = timed(replay, 20)
            out.append({"config": f"latency, hipGraph replay: ONE utterance of {T} frames, {tag}", "ms": ms,
                        "ms_call_to_sync": timed_sync(replay, 20), "x_realtime": T * 256 / 22050 / (ms * 1e-3)})
            m.forward_graphed(mel1); m.forward_graphed(mel1)     # eager, then captured: what vocoder_inference / synthesis_audios call
            ms = timed(lambda: m._forward_graphed_view(mel1), 20)
            out.append({"config": f"latency, public API path (forward_graphed as vocoder_inference calls it: bucketed graph cache, copy in + replay): ONE utterance of {T} frames, {tag}",
                        "ms": ms, "ms_call_to_sync": timed_sync(lambda: m._forward_graphed_view(mel1), 20), "x_realtime": T * 256 / 22050 / (ms * 1e-3)})
    # BigVGAN-base, one 3-s utterance: its resblocks end in convs / AMPBlock kernels, so the accumulating launches are chained
    from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
    hp = dict(V1, activation="snakebeta", snake_logscale=True)
    for streams in (0, -1):
        _lib.check(_lib.lib().amp_set_resblock_streams(streams))
        m = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).to(DEV).eval()
        mel1 = torch.randn(1, 100, 256, generator=torch.Generator().manual_seed(0)).to(DEV)
        m(mel1)
        replay, static_in, _ = m.capture(1, 256)
        static_in.copy_(mel1)
        ms = timed(replay, 20)
        out.append({"config": "latency, hipGraph replay: BigVGAN-base, ONE utterance of 256 frames, " + ("sequential resblocks" if streams == 0 else "concurrent resblocks (default)"),
                    "ms": ms, "x_realtime": 256 * 256 / 24000 / (ms * 1e-3)})
    _lib.check(_lib.lib().amp_set_resblock_streams(-1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="", choices=["", "c1", "c3", "c5", "vits", "mel", "mel_large", "pcm", "list", "lat"])
    a = ap.parse_args()
    runs = {"c1": c1, "c3": c3, "c5": c5, "vits": vits, "mel": mel, "mel_large": mel_large, "pcm": pcm, "list": lst, "lat": lat}
    with torch.no_grad():
        for name, fn in runs.items():
            if a.only in ("", name):
                for o in fn(a.reps):
                    print(json.dumps(o), flush=True)
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
