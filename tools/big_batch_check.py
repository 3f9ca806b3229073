"""Robustness probes beyond the BASELINE sizes (tuning aid; not part of the product): HiFi-GAN V1 at B = 512 / 1024 and BigVGAN-base at B = 256 against the same
batch in 64- / 32-item chunks, bitwise (byte offsets beyond 4 GB, element counts beyond 2^31); one 20 000-frame utterance (232 s) in split-f16 against exact-fp32 arithmetic."""
import sys, torch
sys.path.insert(0, '.')
from types import SimpleNamespace as NS
from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
from amphion_amd.utils.synthetic import randomize_, synthetic_mel
V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
          resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)
m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**V1)))), 1234).cuda().eval()
for B in (512, 1024):
    mel = synthetic_mel(B, 80, 256, seed=5).cuda()
    with torch.no_grad():
        big = m(mel)
        torch.cuda.synchronize()
        ok = True
        for i in range(0, B, 64):
            part = m(mel[i:i + 64].contiguous())
            if not torch.equal(part, big[i:i + 64]):
                ok = False; print("MISMATCH at items", i, (part - big[i:i+64]).abs().max().item())
        print(f"B={B}: [{tuple(big.shape)}] finite={bool(torch.isfinite(big).all())} equal_to_64-item_chunks={ok} mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del big, mel
    torch.cuda.empty_cache()

# BigVGAN-base, B = 256 against 32-item chunks; one 4-minute utterance in both precisions
from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
from amphion_amd import _lib
hp = dict(V1, activation="snakebeta", snake_logscale=True)
bg = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).cuda().eval()
mel = torch.randn(256, 100, 256, generator=torch.Generator().manual_seed(0)).cuda()
with torch.no_grad():
    big = bg(mel); ok = True
    for i in range(0, 256, 32):
        if not torch.equal(bg(mel[i:i + 32].contiguous()), big[i:i + 32]):
            ok = False; print("BigVGAN MISMATCH at", i)
    print(f"BigVGAN B=256: finite={bool(torch.isfinite(big).all())} equal_to_32-item_chunks={ok}", flush=True)
    del big, mel
    torch.cuda.empty_cache()
    _lib.set_precision("f32")     # the precision is a property of the handle: a second instance of each model, built (lazily, at its first forward) in exact fp32
    m32 = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**V1)))), 1234).cuda().eval()
    bg32 = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).cuda().eval()
    for name, net, net32, nm in (("HiFi-GAN", m, m32, 80), ("BigVGAN", bg, bg32, 100)):
        long_mel = (synthetic_mel(1, nm, 20000, seed=9) if nm == 80 else torch.randn(1, 100, 20000, generator=torch.Generator().manual_seed(2)) * 0.5).cuda()
        b = net32(long_mel)
        a = net(long_mel)
        print(f"{name} one utterance of 20 000 frames ({a.shape[-1] / 22050:.0f} s): finite={bool(torch.isfinite(a).all())} max |f16x3 - f32| = {(a - b).abs().max().item():.2e} (|y| max {a.abs().max().item():.2f})", flush=True)
