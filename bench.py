#!/usr/bin/env python
"""bench.py -- vocoder-inference throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torch.distributed environment launches its own N ranks (it re-executes
itself through torch.distributed.run on 127.0.0.1); under an external launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.

A "step" is one forward of the hot path (HiFi-GAN V1 generator, mel -> waveform) over one batch of
synthetic mels already resident in HBM: BASELINE.json configs[1] = B=64 x 80 mel x 256 frames per GPU
(weak scaling: every rank runs its own B=64 shard; for N>1 the step includes the RCCL gather of the
audio to rank 0; before the timed region rank 0 checks that the gathered [64 N, L] tensor equals N
single-GPU runs bit for bit).  Prints ONE JSON line on rank 0; at N = 1 the line also carries the CPU baseline, the
PyTorch-ROCm library baseline and the other BASELINE.json configs (C3 BigVGAN, C5 VITS, mel front end, latency).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from types import SimpleNamespace as NS  # noqa: E402

SAMPLE_RATE = 22050
B_PER_GPU, N_MEL, T_FRAMES = 64, 80, 256
# Algorithmic cost per OUTPUT SAMPLE of HiFi-GAN V1 (SURVEY.md §8d / BASELINE.md §3):
FLOP_PER_SAMPLE_ALL = 2_398_848
FLOP_PER_SAMPLE_MRF = 2_322_432      # the 72 ResBlock convs (96.8 %)
BYTES_PER_SAMPLE_MRF = 14_976        # layer-wise minimum fp32 HBM traffic of those convs
PEAK_FP32_TFLOPS = 157.3             # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
PEAK_F16_TFLOPS = 2516.6             # dense f16 MFMA (256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0
# Sustained f16 MFMA rate of a register-resident v_mfma_f32_32x32x16_f16 loop on RANDOM operands, measured
# on MI355X (profiles/r1_mfma_peak_microbench.txt): the chip clocks down to ~1.6 GHz under MFMA load
# (2192 TF on zeros, 1580-1630 TF on random data), so this -- not 2516.6 -- is what a perfect kernel gets.
SUSTAINED_F16_TFLOPS = 1600.0
# written by tools/summarize_prof.py from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_round.sh (newest round first)
PROFILE_TRAFFIC_CSVS = [os.path.join(ROOT, "profiles", n) for n in ("r6_hbm_traffic.csv", "r5_hbm_traffic.csv", "r4_hbm_traffic.csv", "r3_hbm_traffic.csv", "r2_hbm_traffic.csv")]
# arithmetic of the conv contractions -> (dtype string, peak for ALGORITHMIC flops, note)
PRECISIONS = {
    "f16x3": ("f32 (split-f16 MFMA: 3 x v_mfma_f32_32x32x16_f16 per term, f32 accumulate)", PEAK_F16_TFLOPS / 3.0,
              "dense f16 MFMA peak / 3 MFMAs per algorithmic product term"),
    "f32": ("f32", PEAK_FP32_TFLOPS, "fp32 MFMA (= fp32 vector) peak"),
}


# HiFi-GAN V1 hyper-parameters (reference config/vits.json:36-71)
HIFIGAN_V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
                  upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
                  resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])


def build_model(device):
    """The product path only: amphion_amd module + seeded random-init weights (no checkpoints offline)."""
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
    from amphion_amd.utils.synthetic import randomize_

    hp = dict(HIFIGAN_V1)
    cfg = NS(preprocess=NS(n_mel=N_MEL, hop_size=256, sample_rate=SAMPLE_RATE), model=NS(hifigan=NS(**hp)))
    model = randomize_(HiFiGAN(cfg), 1234)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}   # CPU copy for the cpu_baseline leg
    return model.to(device).eval(), sd, hp


def cpu_baseline(sd, hp, budget_s=20.0):
    """Reference CPU path (oracle = the reference's own torch ops) on the host cores, bounded sample.
    The ONLY place bench.py touches oracle/."""
    from amphion_amd.utils.synthetic import synthetic_mel
    from oracle import vocoder_oracle as vo

    # torch's CPU convs collapse when oversubscribed on the 2 x EPYC 9575F GPU host (measured with
    # tests/experiments/cpu_threads_sweep.py at B=4,T=256: 8 thr x5.3 RT, 16 thr x6.7, 32 thr x4.3, 64 thr x2.4,
    # 128 thr x1.3, 256 thr x0.17); 16 threads is the best, so that is the baseline we report.
    cores = min(os.cpu_count() or 1, int(os.environ.get("AMP_CPU_BASELINE_THREADS", "16")))
    torch.set_num_threads(cores)
    import statistics
    with torch.no_grad():
        vo.hifigan_forward(sd, hp, synthetic_mel(1, N_MEL, 32, seed=1))  # warm-up
        # leg b1: B = 1, T = 256, median of 3
        mel1 = synthetic_mel(1, N_MEL, T_FRAMES, seed=3)
        t_b1 = []
        for _ in range(3):
            t0 = time.perf_counter()
            vo.hifigan_forward(sd, hp, mel1)
            t_b1.append(time.perf_counter() - t0)
        b1_s = statistics.median(t_b1)
        # leg b4: B = 4, T = 256, as many forwards as fit the budget (the figure the line's top-level fields carry)
        B, T = 4, T_FRAMES
        mel = synthetic_mel(B, N_MEL, T, seed=2)
        t0 = time.perf_counter()
        reps = 0
        while True:
            vo.hifigan_forward(sd, hp, mel)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s * 0.35 or reps >= 16:
                break
        # leg b64: the GPU line's OWN workload (BASELINE configs[1]: B = 64, T = 256), one forward -- about 30 s on 16 threads (the same per-item work as
        # b4 sixteen times; skipped, with the reason, when b4 predicts more than 60 s)
        b64 = None
        per_b4 = el / reps
        if per_b4 * (B_PER_GPU / B) <= 60.0:
            try:
                mel64 = synthetic_mel(B_PER_GPU, N_MEL, T_FRAMES, seed=0)
                t64 = time.perf_counter()
                vo.hifigan_forward(sd, hp, mel64)
                s64 = time.perf_counter() - t64
                n64 = B_PER_GPU * T_FRAMES * 256
                b64 = {"value": n64 / s64, "unit": "samples/s", "x_realtime": n64 / s64 / SAMPLE_RATE, "s_per_forward": s64,
                       "sample": f"B={B_PER_GPU}, T={T_FRAMES}: ONE forward of the GPU line's workload, {s64:.1f} s"}
                del mel64
            except Exception as e:  # noqa: BLE001
                b64 = {"error": f"{type(e).__name__}: {e}"[:300]}
        else:
            b64 = {"skipped": f"b4 predicts {per_b4 * B_PER_GPU / B:.0f} s for one B={B_PER_GPU} forward"}
        # leg c1_clips: the CPU twin of other_configs.c1_clips = bins/vocoder/inference.py over the 16 clips at inference.batch_size = 1
        # (reference :97-111): wav -> mel front end -> HiFi-GAN V1 -> crop -> PCM16, one utterance per forward, once
        c1 = None
        try:
            import numpy as np
            from oracle import pcm16 as opcm
            G = np.load(os.path.join(ROOT, "tests", "golden", "golden_c1.npz"))
            clips = [torch.from_numpy(G[f"pcm_{i}"].astype(np.float32) / 32768.0) for i in range(16)]
            pp = NS(sample_rate=SAMPLE_RATE, n_fft=1024, win_size=1024, hop_size=256, n_mel=N_MEL, fmin=0, fmax=8000)
            t1 = time.perf_counter()
            n_out = 0
            for w in clips:
                m = vo.extract_mel_features(w.unsqueeze(0), pp)
                m = m if m.dim() == 3 else m.unsqueeze(0)
                y = vo.hifigan_forward(sd, hp, m)
                y = y.reshape(-1)[: m.shape[-1] * 256]
                n_out += int(opcm.float_to_pcm16(y.numpy()).size) if hasattr(opcm, "float_to_pcm16") else int(y.numel())
            c1_s = time.perf_counter() - t1
            secs = sum(int(c.numel()) for c in clips) / float(SAMPLE_RATE)
            c1 = {"s_total": c1_s, "audio_s": secs, "x_realtime": secs / c1_s, "samples_out": n_out,
                  "sample": "the 16 clips of tests/golden/golden_c1.npz, one utterance per forward, oracle front end + generator + PCM16, once"}
        except Exception as e:  # noqa: BLE001  (a reported baseline: its failure must not cost the line)
            c1 = {"error": f"{type(e).__name__}: {e}"[:300]}
    samples = reps * B * T * 256
    n1 = T_FRAMES * 256
    head = b64 if (b64 and "value" in b64) else {"value": samples / el, "x_realtime": samples / el / SAMPLE_RATE}
    return {
        "value": head["value"],
        "unit": "samples/s",
        "x_realtime": head["x_realtime"],
        "cores": torch.get_num_threads(),
        "threads": torch.get_num_threads(),
        "host_cores": os.cpu_count(),
        "kind": "port",
        "sample": (f"1 x HiFi-GAN V1 forward at B={B_PER_GPU}, T={T_FRAMES} -- the GPU line's workload -- (oracle/vocoder_oracle.py, torch CPU fp32, "
                   f"{torch.get_num_threads()} threads), {b64['s_per_forward']:.1f} s") if (b64 and "value" in b64) else
                  (f"{reps} x HiFi-GAN V1 forward at B={B}, T={T} (oracle/vocoder_oracle.py, torch CPU fp32, "
                   f"{torch.get_num_threads()} threads), {el:.1f} s"),
        "sample_note": "top-level value = leg b64 (the GPU line's own B = 64 batch, one forward) when it ran, else leg b4; "
                       "threads = the measured optimum of tests/experiments/cpu_threads_sweep.py, not the host's core count",
        "legs": {
            "b1": {"value": n1 / b1_s, "unit": "samples/s", "x_realtime": n1 / b1_s / SAMPLE_RATE, "s_per_forward": b1_s,
                   "sample": "B=1, T=256, median of 3 forwards"},
            "b4": {"value": samples / el, "unit": "samples/s", "x_realtime": samples / el / SAMPLE_RATE, "sample": f"B=4, T=256, {reps} forwards in {el:.1f} s"},
            "b64": b64,
            "c1_clips": c1,
        },
    }


def library_baseline(sd, hp, device, reps=3):
    """The reference's op sequence (oracle/vocoder_oracle.py = hifigan.py:203-219 incl. the per-forward weight-norm
    fold) run ON THE GPU through PyTorch-ROCm -- MIOpen convolutions + eager element-wise kernels -- at the bench
    shape, i.e. what the unmodified reference gets on this MI355X.  A reported baseline beside
    cpu_baseline, same checker-only use of oracle/."""
    from amphion_amd.utils.synthetic import synthetic_mel
    from oracle import vocoder_oracle as vo

    sd_dev = {k: v.to(device) for k, v in sd.items()}
    mel = synthetic_mel(B_PER_GPU, N_MEL, T_FRAMES, seed=0).to(device)
    with torch.no_grad():
        t0 = time.perf_counter()
        vo.hifigan_forward(sd_dev, hp, mel)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            vo.hifigan_forward(sd_dev, hp, mel)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    n = B_PER_GPU * T_FRAMES * 256
    del sd_dev
    torch.cuda.empty_cache()
    return {"value": n / ms * 1e3, "unit": "samples/s", "x_realtime": n / ms * 1e3 / SAMPLE_RATE, "ms_per_step": ms,
            "first_call_s": first, "kind": "reference ops on PyTorch-ROCm (MIOpen convs, eager element-wise, fp32)",
            "sample": f"{reps} x HiFi-GAN V1 forward at B={B_PER_GPU}, T={T_FRAMES} on the same GPU, after one warm-up call "
                      f"({first:.1f} s: MIOpen solver search)", "torch": torch.__version__}


def strict_fp32(device, steps=5):
    """The same workload with the conv contractions in exact fp32 (v_mfma_f32_32x32x2_f32, amp_set_precision(AMP_PRECISION_F32)):
    the no-emulation number, in the driver's own line.  Its peak is the fp32 MFMA peak (157.3 TFLOP/s)."""
    from amphion_amd import _lib
    from amphion_amd.utils.synthetic import synthetic_mel

    _lib.set_precision("f32")
    try:
        model, _, hp = build_model(device)
        mel = synthetic_mel(B_PER_GPU, N_MEL, T_FRAMES, seed=0).to(device)
        with torch.no_grad():
            for _ in range(2):
                model(mel)
            torch.cuda.synchronize()
            model.set_profiling(steps)
            t0 = time.perf_counter()
            for _ in range(steps):
                model(mel)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        dom = sum(model.last_timing_ms(100 + 16 * 1 + 2, b) for b in range(steps)) / steps / 6.0     # six unfused k = 11 convs at C = 128
        names = sorted({n for b in range(steps) for n in model.kernel_names(100 + 16 * 1 + 2, b)})
        model.set_profiling(0)
        n = B_PER_GPU * T_FRAMES * model.hop_factor
        C1, T1 = hp["upsample_initial_channel"] // 4, T_FRAMES * hp["upsample_rates"][0] * hp["upsample_rates"][1]
        dom_tf = 2.0 * C1 * C1 * 11 * B_PER_GPU * T1 / (dom * 1e-3) / 1e12
        ms = el / steps * 1e3
        return {"ms_per_step": ms, "steps": steps, "samples_per_s": n / ms * 1e3, "x_realtime": n / ms * 1e3 / SAMPLE_RATE,
                "whole_forward_tflops": FLOP_PER_SAMPLE_ALL * n / (ms * 1e-3) / 1e12,
                "frac_of_fp32_peak": FLOP_PER_SAMPLE_ALL * n / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                "dominant_conv": {"kernel": " | ".join(names), "launch_us": dom * 1e3, "tflops": dom_tf, "frac_of_fp32_peak": dom_tf / PEAK_FP32_TFLOPS},
                "dtype": "f32 operands, f32 MFMA, f32 accumulate (no split-f16 emulation)"}
    finally:
        _lib.set_precision("f16x3")
        torch.cuda.empty_cache()


def conv_flop_per_frame(convs):
    """sum of 2 * Cin * Cout * k * (outputs per mel frame) over (cin, cout, k, outputs_per_frame) tuples"""
    return sum(2.0 * ci * co * k * m for ci, co, k, m in convs)


def other_configs(reps=5):
    """The other BASELINE.json configs on this GPU, driver-timed in the same run (product path only; runners in
    tools/bench_configs.py), each with its own algorithmic cost and roofline fraction."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as bc

    out = {}
    with torch.no_grad():
        # C3 BigVGAN-base 24 kHz, B=32: convs = HiFi-GAN V1's with a 100-bin conv_pre; the 73 anti-aliased Snake
        # activations are VALU work (per element: 2x up 12 taps, 2 Snake, 12-tap down), reported as their HBM bytes
        r = bc.c3(reps)[0]
        n = 32 * 256 * 256
        flop = (FLOP_PER_SAMPLE_ALL + 2.0 * (100 - N_MEL) * 512 * 7 / 256) * n
        # byte side: the convs' layer-wise minimum 15 830 B/sample + the 73 anti-aliased activations at their fused
        # minimum of one read and one write each, 15 232 B/sample = 31 062 B/sample; the time is the sum of a conv part that is
        # MFMA-bound and an activation part that is HBM-bound, so both fractions are of the SAME step time
        act_bytes, conv_bytes = 15_232.0 * n, 15_830.0 * n
        r.update({"algorithmic_conv_tflop": flop / 1e12, "conv_tflops": flop / r["ms_per_step"] / 1e9,
                  "frac_of_f16x3_mfma_peak": flop / r["ms_per_step"] / 1e9 / (PEAK_F16_TFLOPS / 3.0),
                  "activation_algorithmic_GB": act_bytes / 1e9, "conv_layerwise_min_GB": conv_bytes / 1e9,
                  "algorithmic_GBps": (act_bytes + conv_bytes) / r["ms_per_step"] / 1e6,
                  "frac_of_hbm_peak": (act_bytes + conv_bytes) / r["ms_per_step"] / 1e6 / PEAK_HBM_GBS,
                  "roofline_floor_ms": {"convs_at_f16x3_mfma_peak": flop / (PEAK_F16_TFLOPS / 3.0) / 1e9,
                                        "activations_at_hbm_peak": act_bytes / PEAK_HBM_GBS / 1e6},
                  "launches_per_forward": bc.c3_launch_counts(),
                  "note": "round 4: the AMPBlocks of the C = 32 stage (and k = 3 at C = 64) run as ONE launch each (csrc/ampb_f16x3.hip); the "
                          "C >= 128 stages keep separate conv / act1d launches"})
        out["c3_bigvgan"] = r
        torch.cuda.empty_cache()
        # C5 VITS decode path B=16: enc_q (513 -> 192, WN 16 x k5) + flow both ways (4 couplings x WN 4 x k5, twice) + decoder
        r = bc.c5(reps)[0]
        H = 192
        wn = lambda layers, k: [(H, 2 * H, k, 1)] * layers + [(H, 2 * H, 1, 1)] * (layers - 1) + [(H, H, 1, 1)]
        enc_q = [(513, H, 1, 1)] + wn(16, 5) + [(H, 2 * H, 1, 1)]
        coupling = [(H // 2, H, 1, 1)] + wn(4, 5) + [(H, H // 2, 1, 1)]
        frame_flop = conv_flop_per_frame(enc_q) + 2 * 4 * conv_flop_per_frame(coupling)
        n = 16 * 256 * 256
        flop = (FLOP_PER_SAMPLE_ALL + 2.0 * (H - N_MEL) * 512 * 7 / 256) * n + frame_flop * 16 * 256
        r.update({"algorithmic_conv_tflop": flop / 1e12, "conv_tflops": flop / r["ms_per_step"] / 1e9,
                  "frac_of_f16x3_mfma_peak": flop / r["ms_per_step"] / 1e9 / (PEAK_F16_TFLOPS / 3.0)})
        out["c5_vits_decode"] = r
        torch.cuda.empty_cache()
        out["vits_text_to_wave"] = bc.vits(reps)[0]       # full SynthesizerTrn.infer at config/vits.json dimensions
        torch.cuda.empty_cache()
        mels = bc.mel(max(reps, 10))
        r = mels[0]
        r.update({"algorithmic_bytes": 64 * 65536 * 4 + 64 * 80 * 256 * 4, "frac_of_hbm_peak": r["algorithmic_GBps"] / PEAK_HBM_GBS})
        out["mel_front_end"] = r
        out["mel_front_end_other_nfft"] = mels[1:]               # n_fft 2048 / 1920 / 512: the one-workgroup-per-frame kernels
        out["mel_front_end_large"] = bc.mel_large(reps)[0]     # 1 024 x 65 536 samples: the dataset-extraction regime
        torch.cuda.empty_cache()
        out["c1_clips"] = bc.c1(reps)[0]                        # BASELINE configs[0]: the 16 real clips end to end, batch_size = 1
        out["latency"] = bc.lat(reps)                           # eager and hipGraph replay (what the drop-in entry points use for repeats)
        out["list_api"] = bc.lst(reps)          # synthesis_audios on 64 utterances of 60..400 frames, host to host
        torch.cuda.empty_cache()
    return out


_DROP_KEYS = {"note", "sample_note", "peak_note", "traffic_source", "kernel_source", "parity_note", "ms_all", "gpu_stream_ms_all",
              "frames_per_item", "per_item_ms", "launches_per_forward", "roofline_floor_ms", "algorithmic_bytes", "samples_out"}


def _compact(o, strlen=90):
    """numbers to 5 significant digits, long strings cut, numeric arrays and commentary dropped"""
    if isinstance(o, bool) or o is None or isinstance(o, int):
        return o
    if isinstance(o, float):
        return float(f"{o:.5g}")
    if isinstance(o, str):
        return o if len(o) <= strlen else o[: strlen - 3] + "..."
    if isinstance(o, (list, tuple)):
        if o and all(isinstance(x, (int, float)) for x in o):
            return [float(f"{x:.4g}") for x in o] if len(o) <= 8 else None      # (per-rank figures of an 8-GPU run stay)
        return [c for c in (_compact(x, strlen) for x in o) if c is not None]
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if k in _DROP_KEYS:
                continue
            c = _compact(v, strlen)
            if c is not None:
                out[k] = c
        return out
    return str(o)[:strlen]


def compact_line(result, detail_path):
    """The printed JSON line: the contract's keys untouched, `roofline` and `cpu_baseline` with their numbers, one short object per side leg,
    `summary_ms` first AND last."""
    keep_whole = ("metric", "value", "unit", "x_realtime", "x_realtime_per_gpu", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                  "scaling", "vs_baseline", "dtype", "data", "config", "gather_check")
    line = {"summary_ms": result.get("summary_ms")}
    for k in keep_whole:
        if k in result:
            line[k] = result[k]
    roof = dict(result.get("roofline", {}))
    if isinstance(roof.get("kernel"), str):
        roof["kernel"] = roof["kernel"].split(" (")[0]           # the instantiation's name; the description is in the detail file
    line["roofline"] = _compact(roof, 120)
    for k, v in result.items():
        if k in line or k in ("roofline", "summary_ms", "parity_note"):
            continue
        line[k] = _compact(v, 48 if k == "other_configs" else 100)
    def strip(o):   # the side legs keep ms / x_realtime / fractions; what follows from them is in the detail file
        if isinstance(o, dict):
            return {k: strip(v) for k, v in o.items() if k not in ("samples_per_s", "algorithmic_conv_tflop", "activation_algorithmic_GB",
                                                                   "conv_layerwise_min_GB", "n", "ms_min", "ms_max", "audio_s")}
        if isinstance(o, list):
            return [strip(x) for x in o]
        return o
    if isinstance(line.get("other_configs"), dict):
        line["other_configs"] = strip(line["other_configs"])
    lat = line.get("other_configs", {}).get("latency") if isinstance(line.get("other_configs"), dict) else None
    if isinstance(lat, list):                                   # twelve rows -> {frames / mode: ms}
        short = {}
        for row in result["other_configs"]["latency"]:
            cfg = row.get("config", "")
            mode = "graph" if "hipGraph" in cfg else "api" if "public API" in cfg else "eager"
            frames = "860" if "860 frames" in cfg else "256"
            rb = "seq" if "sequential" in cfg else "conc"
            model = "bigvgan_" if "BigVGAN" in cfg else ""
            short[f"{model}{frames}f_{mode}_{rb}"] = float(f"{row.get('ms', 0.0):.4g}")
        line["other_configs"]["latency"] = short
    oth = line.get("other_configs")
    if isinstance(oth, dict) and isinstance(result.get("other_configs", {}).get("mel_front_end_other_nfft"), list):
        short = {}                                              # seven rows -> {direction_nfft: ms}
        for row in result["other_configs"]["mel_front_end_other_nfft"]:
            cfg = row.get("config", "")
            n = cfg.split("n_fft ")[1].split(" ")[0] if "n_fft " in cfg else "?"
            short[("inverse_" if cfg.startswith("inverse") else "forward_") + n + "_ms"] = float(f"{row.get('ms_per_step', 0.0):.4g}")
        oth["mel_front_end_other_nfft"] = short
    line["detail_file"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    line["summary_ms_tail"] = result.get("summary_ms")
    return line


class AgreedFailure(RuntimeError):
    """a failure every rank has been told about (multi_gpu_diagnostics): safe to report in the line, nobody is left inside a collective"""


def multi_gpu_diagnostics(model, mel, total_items, device, per_rank_ms, reps=5):
    """N > 1 only, after the timed region (collective calls: every rank runs this).  What a first 8-GPU run needs to be read:
    per-rank step times, the generator alone (no gather), the fp32 gather alone (nothing to overlap with) and the same gather with
    16-bit PCM rows (amp_wav_to_pcm16 on the device first: half the bytes on every xGMI link)."""
    from amphion_amd.distributed import gather_audio
    from amphion_amd.utils.io import wav_to_pcm16

    def agree(err):
        """collective: every rank learns whether ANY rank failed its local part of the phase; then all of them raise the same exception
        (an exception swallowed on one rank left the others blocked in the next collective)"""
        flag = torch.tensor([1.0 if err else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() > 0:
            raise AgreedFailure(err or "another rank failed in this phase")

    def local(fn):
        """a rank-local step (allocations, kernels): its failure is agreed on before anybody enters the next collective"""
        err, res = None, None
        try:
            res = fn()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
        agree(err)
        return res

    def timed_ms(fn, collective):
        dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        err = None
        e0.record()
        try:
            for _ in range(reps):
                fn()
        except Exception as e:  # noqa: BLE001
            if collective:
                raise      # inside a collective nothing can be agreed any more: let it propagate, the process group's timeout ends every rank
            err = f"{type(e).__name__}: {e}"[:300]
        e1.record()
        torch.cuda.synchronize()
        if not collective:
            agree(err)
        t = torch.tensor([e0.elapsed_time(e1) / reps], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    with torch.no_grad():
        wav = local(lambda: model(mel).squeeze(1))
        pcm = local(lambda: wav_to_pcm16(wav))
        compute_ms = timed_ms(lambda: model(mel), collective=False)
        # the receive buffer of the gather is the one allocation inside the collective: try its size first, on every rank, and agree
        local(lambda: torch.empty((total_items,) + tuple(wav.shape[1:]), dtype=wav.dtype, device=device) if dist.get_rank() == 0 else None)
        gather_ms = timed_ms(lambda: gather_audio(wav, total_items, dst=0), collective=True)
        gather_pcm_ms = timed_ms(lambda: gather_audio(pcm, total_items, dst=0), collective=True)
        pcm_convert_ms = timed_ms(lambda: wav_to_pcm16(wav), collective=False)
    step_ms = max(per_rank_ms)
    nbytes = wav.numel() * 4
    return {"per_rank_ms": [round(v, 3) for v in per_rank_ms], "per_rank_ms_min": min(per_rank_ms), "per_rank_ms_max": step_ms,
            "compute_only_ms": compute_ms, "gather_ms": gather_ms, "gather_ms_overlapped": max(0.0, step_ms - compute_ms),
            "gather_ms_pcm16": gather_pcm_ms, "pcm16_convert_ms": pcm_convert_ms,
            "gather_bytes_per_rank": nbytes, "gather_GBps_per_link": nbytes / gather_ms / 1e6,
            "gather_note": "gather_ms: the fp32 gather alone, stream idle otherwise (max over ranks); gather_ms_overlapped: what it adds to a step "
                           "when issued asynchronously behind the next batch's forward (step - compute_only); gather_ms_pcm16: int16 rows"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_command(argv, n):
    """`python bench.py --gpus N` outside a launcher: the command that starts N ranks of this script on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU, library and other-config legs (profiling runs)")
    ap.add_argument("--precision", choices=sorted(PRECISIONS), default="f16x3",
                    help="arithmetic of the conv contractions (include/amphion_hip.h: amp_precision)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start our own N ranks (one process per GPU, RCCL over xGMI) and pass their line through
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: this node exposes {torch.cuda.device_count()} GPU(s)")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(relaunch_command(sys.argv[1:], args.gpus), env=env))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from amphion_amd import _lib
    from amphion_amd.distributed import gather_audio
    from amphion_amd.utils.synthetic import synthetic_mel

    _lib.set_precision(args.precision)
    dtype_str, peak_tflops, peak_note = PRECISIONS[args.precision]

    model, sd, hp = build_model(device)
    mel = synthetic_mel(B_PER_GPU, N_MEL, T_FRAMES, seed=rank).to(device)  # resident in HBM before timing
    L = T_FRAMES * model.hop_factor
    total_items = B_PER_GPU * world

    pending = []   # N > 1: (gathered audio, work) of the steps in flight

    def step():
        """One batch per rank through the generator; for N > 1 the audio is then gathered on rank 0 WITHOUT
        stalling the compute stream (the transfer overlaps the next batch, as in a serving loop); every gather
        is waited for inside the timed region (fence())."""
        with torch.no_grad():
            wav = model(mel)
            if world > 1:
                pending.append(gather_audio(wav.squeeze(1), total_items, dst=0, async_op=True) + (wav,))
                return pending[-1][0]
            return wav

    def fence():
        for _, work, _ in pending:
            work.wait()
        pending.clear()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    gather_check = None
    if world > 1:
        # the gathered [64 N, L] tensor must equal N single-GPU runs bit for bit: rank 0 recomputes
        # every rank's shard (same seeded mel, same weights) on its own GPU and compares -- untimed
        got = step()
        fence()
        if rank == 0:
            with torch.no_grad():
                for r in range(world):
                    ref = model(synthetic_mel(B_PER_GPU, N_MEL, T_FRAMES, seed=r).to(device)).squeeze(1)
                    if not torch.equal(got[r * B_PER_GPU:(r + 1) * B_PER_GPU], ref):
                        raise SystemExit(f"gather check FAILED: rows of rank {r} differ from a single-GPU run of its shard")
            gather_check = f"gathered [{total_items}, {L}] == {world} single-GPU runs, bitwise"
        dist.barrier()
    # HIP events around the kernel groups of EVERY timed forward, recorded on the launch stream without
    # synchronising (a ring of `steps` event sets inside the handle); read back after the closing fence.
    # dominant kernel: the fused ResBlock pairs of stage 1 (C=128), kernel 11 -- resblock j=2 of stage i=1 is
    # three back-to-back launches of the fused pair kernel the policy picks (k = 11 pairs with dilation 1, 3, 5; the name comes from the library)
    DOM_STAGE, DOM_RB, DOM_LAUNCHES = 1, 2, 3
    model.set_profiling(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    multi = None
    if world > 1:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        elapsed = max(float(t.item()) for t in every)
    fwd_ms, mrf_ms, stage_ms, dom_ms = [], [], [], []
    for back in range(args.steps):
        fwd_ms.append(model.last_timing_ms(0, back))
        mrf_ms.append(model.last_timing_ms(1, back))
        stage_ms.append([model.last_timing_ms(2 + i, back) for i in range(len(hp["upsample_rates"]))])
        dom_ms.append(model.last_timing_ms(100 + 16 * DOM_STAGE + DOM_RB, back))
    # which kernel the launches of that resblock ran: reported by the library for the profiled forwards themselves
    # (amp_gen_kernel_name: the launch policy's actual pick for this shape, rocprofv3 spelling), not assumed here
    knames = sorted({n for back in range(args.steps) for n in model.kernel_names(100 + 16 * DOM_STAGE + DOM_RB, back)})
    model.set_profiling(0)
    if world > 1:
        # after the timed region and after its event ring has been read (the diagnostics run forwards of their own); collective calls,
        # so every rank makes them -- and a failure is reported in the line, it does not cost the line
        try:
            multi = multi_gpu_diagnostics(model, mel, total_items, device, per_rank_ms)
        except AgreedFailure as e:     # raised on EVERY rank after an all-reduce of the error flag; anything else propagates
            multi = {"multi_gpu_diagnostics_error": f"{type(e).__name__}: {e}"[:400], "per_rank_ms": [round(v, 3) for v in per_rank_ms]}

    if rank == 0:
        samples_per_step = total_items * L
        value = samples_per_step * args.steps / elapsed
        n_local = B_PER_GPU * L
        mrf_s = (sum(mrf_ms) / len(mrf_ms)) * 1e-3
        fwd_s = (sum(fwd_ms) / len(fwd_ms)) * 1e-3
        mrf_tflops = FLOP_PER_SAMPLE_MRF * n_local / mrf_s / 1e12
        mrf_gbs = BYTES_PER_SAMPLE_MRF * n_local / mrf_s / 1e9
        # ---- roofline of the dominant kernel (per launch), plus the whole MRF stack beside it ----
        C1 = hp["upsample_initial_channel"] // 4                      # channels of stage 1
        T1 = T_FRAMES * hp["upsample_rates"][0] * hp["upsample_rates"][1]
        k_dom = hp["resblock_kernel_sizes"][DOM_RB]
        fused = args.precision == "f16x3"
        dom_launches = DOM_LAUNCHES if fused else 2 * DOM_LAUNCHES     # unfused: conv1 and conv2 are separate launches
        dom_flop = 2 * (2.0 * C1 * C1 * k_dom * B_PER_GPU * T1) / (1 if fused else 2)   # per launch
        dom_bytes = (2 if fused else 2.5) * 4.0 * B_PER_GPU * C1 * T1  # read x + write y (+ residual when unfused)
        dom_s = (sum(dom_ms) / len(dom_ms)) * 1e-3 / dom_launches
        dom_tflops = dom_flop / dom_s / 1e12
        traffic = None
        kname = " | ".join(knames)
        traffic_csv = None
        if len(knames) == 1:
            # PMC bytes of exactly this instantiation from the newest profile set that has it (a PMC pass cannot run inside
            # the timed process: "static").  FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, bytes per launch.
            for path in PROFILE_TRAFFIC_CSVS:
                if not os.path.exists(path):
                    continue
                rows = [line for line in open(path) if ("amp::" + knames[0] + "(") in line]
                if rows:
                    n = sum(float(r.rsplit(",", 6)[1]) for r in rows)
                    traffic = sum(float(r.rsplit(",", 1)[1]) * float(r.rsplit(",", 6)[1]) for r in rows) / n * 1e6
                    traffic_csv = os.path.relpath(path, ROOT)
                    break
        roofline = {
            "kernel": kname + (" (fused ResBlock pair, C=128, k=11: conv1 -> LDS -> conv2 + residual; the three launches of "
                               "resblock j=2 of stage 1, dilations 1 / 3 / 5)" if fused else ""),
            "bound": "mfma",
            "achieved": dom_tflops,
            "peak": peak_tflops,
            "peak_note": peak_note,
            "unit": "TFLOP/s",
            "frac": dom_tflops / peak_tflops,
            "kernel_source": "amp_gen_kernel_name (library-reported for the timed forwards)",
            "traffic": traffic,
            "traffic_over_algorithmic": (traffic / dom_bytes) if traffic else None,
            "traffic_source": ("static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command "
                               "(FETCH x2 gfx950 correction), " + traffic_csv) if traffic else
                              "no PMC pass of this kernel instantiation is on file under profiles/",
            "algorithmic_flop_per_launch": dom_flop,
            "algorithmic_bytes_per_launch": dom_bytes,
            "launch_us": dom_s * 1e6,
            "hbm_term": {"achieved": dom_bytes / dom_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": dom_bytes / dom_s / 1e9 / PEAK_HBM_GBS},
            "sustained_peak": (SUSTAINED_F16_TFLOPS / 3.0) if fused else None,
            "frac_of_sustained": dom_tflops / (SUSTAINED_F16_TFLOPS / 3.0) if fused else None,
            "sustained_note": "register-resident f16 MFMA loop on random data sustains 1 617 TFLOP/s at 1.70 GHz and 1.27-1.30 kW (clock-limited "
                              "below the package cap), profiles/r5_power_per_kernel.txt" if fused else None,
            "mrf_stack": {"tflops": mrf_tflops, "frac": mrf_tflops / peak_tflops, "ms": mrf_s * 1e3,
                          "ms_per_stage": [sum(v[i] for v in stage_ms) / len(stage_ms) for i in range(len(stage_ms[0]))],
                          "hbm_gbs_layerwise_min": mrf_gbs},
            "forward_ms": fwd_s * 1e3,
            "whole_forward_tflops": FLOP_PER_SAMPLE_ALL * n_local / fwd_s / 1e12,
        }
        result = {
            "metric": "audio samples/sec (HiFi-GAN V1 22.05 kHz generator, mel->wav)",
            "value": value,
            "unit": "samples/s",
            "x_realtime": value / SAMPLE_RATE,
            "x_realtime_per_gpu": value / SAMPLE_RATE / world,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype_str,
            "data": "synthetic",
            "config": {
                "workload": f"HiFi-GAN V1 22.05 kHz, batch={B_PER_GPU}/GPU synthetic {N_MEL}-ch x {T_FRAMES}-frame mels "
                            f"(BASELINE.json configs[1]), random-init weights, fp32 in/out, conv arithmetic {args.precision}",
                "global_batch": total_items,
                "frames": T_FRAMES,
                "samples_per_step": samples_per_step,
                "parallelism": f"batch-sharded x{world}, result gather on rank 0" if world > 1 else "single GPU",
            },
            "roofline": roofline,
            "parity_note": "this exact shape is under tests/test_gpu_full_size.py (-m gpu): items 0 / 31 / 63 of the batch equal "
                           "that item vocoded alone bit for bit, a batch equals its halves, the CPU oracle agrees on items 0, 21, 42 and 63 "
                           "within 1e-4 (measured 3e-6); the oracle is not run on all 64 items (30 s per batch on the host)",
        }
        if gather_check:
            result["gather_check"] = gather_check
        if multi:
            result.update(multi)
        if not args.no_cpu_baseline and world == 1:
            del out
            # the side legs must never cost the headline line: a failure is reported in place of the figure
            legs = [("other_configs", other_configs), ("library_baseline", lambda: library_baseline(sd, hp, device)),
                    ("cpu_baseline", lambda: cpu_baseline(sd, hp))]
            if args.precision == "f16x3":
                legs.insert(0, ("strict_fp32", lambda: strict_fp32(device)))
            for key, leg in legs:
                try:
                    result[key] = leg()
                except Exception as e:  # noqa: BLE001
                    result[key] = {"error": f"{type(e).__name__}: {e}"[:500]}
                    torch.cuda.empty_cache()
        # a one-glance summary at the HEAD of the line (a record that keeps only the start or the end of stdout still has every leg's figure)
        def _ms(path):
            d = result
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return round(d, 3) if isinstance(d, (int, float)) else None
        summary = {"c2_f16x3_ms": round(result["ms_per_step"], 3), "c2_strict_fp32_ms": _ms(("strict_fp32", "ms_per_step")),
                   "c3_bigvgan_ms": _ms(("other_configs", "c3_bigvgan", "ms_per_step")), "c5_vits_decode_ms": _ms(("other_configs", "c5_vits_decode", "ms_per_step")),
                   "c1_clips_ms": _ms(("other_configs", "c1_clips", "ms_total")), "vits_text_to_wave_ms": _ms(("other_configs", "vits_text_to_wave", "ms_per_step")),
                   "mel_1024_ms": _ms(("other_configs", "mel_front_end", "ms_per_step")),
                   "cpu_b64_x_realtime": _ms(("cpu_baseline", "legs", "b64", "x_realtime")), "cpu_b4_x_realtime": _ms(("cpu_baseline", "legs", "b4", "x_realtime")),
                   "cpu_b1_x_realtime": _ms(("cpu_baseline", "legs", "b1", "x_realtime")),
                   "cpu_c1_clips_x_realtime": _ms(("cpu_baseline", "legs", "c1_clips", "x_realtime")),
                   "miopen_ms": _ms(("library_baseline", "ms_per_step"))}
        try:
            for row in result["other_configs"]["mel_front_end_other_nfft"]:
                for n in (2048, 1920, 512):
                    if f"n_fft {n} " in row["config"] and "inverse" not in row["config"]:
                        summary[f"mel_{n}_ms"] = round(row["ms_per_step"], 4)
            summary["list_api_ragged_ms"] = round(result["other_configs"]["list_api"][0]["ms_total"], 3)
        except (KeyError, TypeError, IndexError):
            pass
        result = {"summary_ms": summary, **result}
        # The full record (per-repeat arrays, notes, every sub-leg) goes to a side file; the printed line is the compact form: every required key,
        # `roofline`, `cpu_baseline`, one figure per leg, <= 6 KB, with the summary repeated as the LAST key -- a record that keeps only the head or
        # only the last ~2 KB of the line still carries every BASELINE config's number.
        detail_path = os.path.join(ROOT, "gpurun_out", "bench_detail.json" if "other_configs" in result else "bench_detail_headline_only.json")
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(result, f)
        except OSError:
            detail_path = None
        line = compact_line(result, detail_path)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
