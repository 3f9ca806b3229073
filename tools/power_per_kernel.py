#!/usr/bin/env python
"""Package power and shader clock of each heavy kernel RUNNING ALONE: a >= 3 s loop of one kernel through the C ABI's
op-level entry points at its BASELINE shape, `rocm-smi` power + sclk sampled underneath (first second discarded), next to
tests/experiments/mfma_peak (register-resident random-operand MFMA loop) on the same box in the same run.

    python tools/power_per_kernel.py [--seconds 3.5] > profiles/r5_power_per_kernel.txt

Columns: W, MHz (median of the samples), algorithmic TFLOP/s, raw f16 MFMA TFLOP/s (= 3 x algorithmic x the recomputed-halo factor the
launch manifest implies is NOT applied: 3 x algorithmic, a lower bound), and that as a fraction of the microbenchmark's sustained rate.
Tuning aid; not part of the product."""
import argparse, ctypes, os, re, statistics, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amphion_amd import _lib


class Sampler:
    """rocm-smi in a thread (the main thread sits in hipDeviceSynchronize, which releases the GIL)"""

    def __init__(self):
        self.rows, self.stop, self.t = [], False, None

    def _once(self):
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        except Exception:  # noqa: BLE001
            return None
        w = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        return (time.perf_counter(), float(w.group(1)) if w else None, float(c.group(1)) if c else None)

    def __enter__(self):
        def run():
            while not self.stop:
                r = self._once()
                if r:
                    self.rows.append(r)
        self.t0 = time.perf_counter()
        self.t = threading.Thread(target=run, daemon=True)
        self.t.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.t.join(timeout=15)

    def summary(self, skip_s=1.0):
        rows = [r for r in self.rows if r[0] - self.t0 >= skip_s] or self.rows
        w = [r[1] for r in rows if r[1] is not None]
        c = [r[2] for r in rows if r[2] is not None]
        return (statistics.median(w) if w else float("nan"), statistics.median(c) if c else float("nan"), len(rows))


def conv_handles(C, k, dils, seed=1):
    L = _lib.lib()
    g = torch.Generator().manual_seed(seed)
    hs = []
    for d in dils:
        w = (torch.randn(C, C, k, generator=g) * (C * k) ** -0.5).contiguous()
        b = torch.randn(C, generator=g) * 0.1
        h = ctypes.c_void_p()
        _lib.check(L.amp_conv_create(0, C, C, k, 1, d, (k * d - d) // 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
        hs.append(h)
    return hs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.5)
    ap.add_argument("--jobs", type=str, default="", help="comma-separated indices into the job list (default: all), e.g. 0,1 = the two fused-pair jobs")
    a = ap.parse_args()
    _lib.set_precision("f16x3")
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    st = _lib.current_stream_ptr(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    B = 64
    jobs = []

    def pair_job(C, k, T):
        h1, h2 = conv_handles(C, k, (1, 3, 5)), conv_handles(C, k, (1, 1, 1), seed=2)
        bufs = [torch.randn(B, C, T, device=dev) * 0.5 for _ in range(2)]
        bufs.append(torch.empty_like(bufs[0]))

        def fn():
            for q in range(3):
                _lib.check(L.amp_pair_forward(h1[q], h2[q], p(bufs[q % 2]), B, T, 0.1, p(bufs[2]), st))
        return fn, 3 * 2 * 2.0 * C * C * k * B * T, f"fused pairs C={C} k={k} (3 launches, d = 1, 3, 5; policy kernel), B={B} T={T}"

    def rb_job(C, k, T):
        h1, h2 = conv_handles(C, k, (1, 3, 5)), conv_handles(C, k, (1, 1, 1), seed=2)
        a1, a2 = (ctypes.c_void_p * 3)(*[h.value for h in h1]), (ctypes.c_void_p * 3)(*[h.value for h in h2])
        x = torch.randn(B, C, T, device=dev) * 0.5
        y = torch.empty_like(x)
        keep = (h1, h2)

        def fn():
            _lib.check(L.amp_resblock_forward(a1, a2, 3, p(x), B, T, 0.1, p(y), st))
        fn.keep = keep
        return fn, 6 * 2.0 * C * C * k * B * T, f"whole ResBlock C={C} k={k}, B={B} T={T}"

    def conv_job(C, k, T):
        (h,) = conv_handles(C, k, (1,))
        x = torch.randn(B, C, T, device=dev) * 0.5
        y = torch.empty_like(x)

        def fn():
            _lib.check(L.amp_conv_forward(h, p(x), B, T, 0.1, None, 1.0, p(y), st))
        return fn, 2.0 * C * C * k * B * T, f"conv C={C} k={k} (stage 0), B={B} T={T}"

    def ampb_job(C, k, T, Bb=32):
        h1, h2 = conv_handles(C, k, (1, 3, 5)), conv_handles(C, k, (1, 1, 1), seed=2)
        a1, a2 = (ctypes.c_void_p * 3)(*[h.value for h in h1]), (ctypes.c_void_p * 3)(*[h.value for h in h2])
        x = torch.randn(Bb, C, T, device=dev) * 0.5
        y = torch.empty_like(x)
        al = (torch.randn(6, C) * 0.3).to(dev)
        be = (torch.randn(6, C) * 0.3).to(dev)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle import vocoder_oracle as vo
        f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12).contiguous().float()
        keep = (h1, h2, f)

        def fn():
            _lib.check(L.amp_ampblock_forward(a1, a2, 3, p(al), p(be), 1, p(f), p(f), p(x), Bb, T, p(y), 0, 1.0, st))
        fn.keep = keep
        return fn, 6 * 2.0 * C * C * k * Bb * T, f"whole AMPBlock C={C} k={k}, B={Bb} T={T}"

    jobs.append(lambda: pair_job(128, 11, 16384))
    jobs.append(lambda: pair_job(128, 7, 16384))
    jobs.append(lambda: rb_job(64, 11, 32768))
    jobs.append(lambda: rb_job(64, 7, 32768))
    jobs.append(lambda: rb_job(128, 3, 16384))
    jobs.append(lambda: rb_job(32, 11, 65536))
    jobs.append(lambda: rb_job(32, 3, 65536))
    jobs.append(lambda: conv_job(256, 11, 2048))
    jobs.append(lambda: ampb_job(32, 11, 65536))
    jobs.append(lambda: ampb_job(32, 3, 65536))

    if a.jobs:
        jobs = [jobs[int(i)] for i in a.jobs.split(",")]
    stamp = ""
    if os.path.exists(os.path.join(ROOT, ".commit_stamp")):
        stamp = " ".join(open(os.path.join(ROOT, ".commit_stamp")).read().split())
    print(f"# power / clock per kernel, each looped alone for {a.seconds} s (tools/power_per_kernel.py), {torch.cuda.get_device_name(0)}; tree {stamp}")
    # microbenchmark first AND last (thermal drift)
    def micro():
        exe = os.path.join(ROOT, "tests", "experiments", "mfma_peak")
        with Sampler() as s:
            out = subprocess.run([exe, str(a.seconds)], capture_output=True, text=True).stdout
        m = re.search(r"([0-9.]+) TFLOP/s", out)
        w, c, n = s.summary()
        return (float(m.group(1)) if m else float("nan")), w, c, n
    tf0, w0, c0, n0 = micro()
    print(f"{'mfma_peak microbenchmark (random f16 operands, register-resident)':78s} {w0:7.0f} W {c0:6.0f} MHz   raw {tf0:7.1f} TFLOP/s   [{n0} samples]")
    print(f"{'kernel':78s} {'W':>7s}   {'MHz':>6s}       {'alg TF':>7s} {'raw f16 TF':>10s} {'/ micro':>8s} {'ms/launch-set':>13s}")
    with torch.no_grad():
        for mk in jobs:
            try:
                fn, flop, name = mk()
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 2
                n = max(4, int(a.seconds * 1e3 / ms))
                with Sampler() as s:
                    e0.record()
                    for _ in range(n):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
                w, c, ns = s.summary()
                tf = flop / (ms * 1e-3) / 1e12
                print(f"{name:78s} {w:7.0f} W {c:6.0f} MHz   {tf:7.1f} {3 * tf:10.1f} {3 * tf / tf0:8.2f} {ms:13.3f}   [{ns} samples]", flush=True)
                del fn
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                print(f"{'(job failed)':78s} {type(e).__name__}: {e}"[:200], flush=True)
    tf1, w1, c1, n1 = micro()
    print(f"{'mfma_peak microbenchmark, again at the end':78s} {w1:7.0f} W {c1:6.0f} MHz   raw {tf1:7.1f} TFLOP/s   [{n1} samples]")


if __name__ == "__main__":
    main()
