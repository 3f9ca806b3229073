"""GPU time of HiFi-GAN V1 on a batch of 64 utterances of 60..400 frames: padded forward vs forward_ragged (tiles beyond
an utterance's valid length exit at once), and the host-side pieces of synthesis_audios around it."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import bench_configs as bc

cfg, m = bc.hifigan()
lens = torch.randint(60, 400, (64,), generator=torch.Generator().manual_seed(3))
order = torch.argsort(lens, descending=True)
lens_s = lens[order]
T = int(lens_s[0])
mel = torch.zeros(64, 80, T)
for r, L in enumerate(lens_s.tolist()):
    mel[r, :, :L] = bc.synthetic_mel(1, 80, L, seed=r)[0]
mel_d = mel.cuda()
lens_d = lens_s.to(torch.int32).cuda()


def gpu_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    print("frames: padded", 64 * T, "valid", int(lens.sum()))
    print("padded forward        %.2f ms" % gpu_ms(lambda: m(mel_d)))
    print("forward_ragged        %.2f ms" % gpu_ms(lambda: m.forward_ragged(mel_d, lens_d)))
    out = m.forward_ragged(mel_d, lens_d)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); o = out.squeeze(1).cpu(); t1 = time.perf_counter()
    print("D2H of [64, %d] fp32  %.2f ms" % (out.shape[-1], (t1 - t0) * 1e3))
    t0 = time.perf_counter(); crops = [o[r, : int(L) * 256].clone() for r, L in enumerate(lens_s.tolist())]; t1 = time.perf_counter()
    print("64 cropped clones     %.2f ms" % ((t1 - t0) * 1e3))
    t0 = time.perf_counter(); md = mel.to("cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("H2D of the mel batch  %.2f ms" % ((t1 - t0) * 1e3))
