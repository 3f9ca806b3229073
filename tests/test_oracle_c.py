"""Cross-check the torch-functional oracle against the independent plain-C restatement of the same
formulas (oracle/c/vocoder_ref.c, double accumulation).  CPU-only, small sizes."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import c_ref
from oracle import vocoder_oracle as vo


def _r(*s, seed=0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed))


def test_conv1d_formula():
    for (cin, cout, k, d, T) in [(5, 7, 3, 1, 20), (4, 4, 7, 3, 31), (3, 2, 11, 5, 64), (6, 3, 1, 1, 9)]:
        x, w, b = _r(2, cin, T, seed=1), _r(cout, cin, k, seed=2), _r(cout, seed=3)
        pad = vo.get_padding(k, d)
        ref = F.conv1d(x, w, b, dilation=d, padding=pad).numpy()
        assert np.abs(c_ref.conv1d(x.numpy(), w.numpy(), b.numpy(), d, pad) - ref).max() <= 1e-5


def test_conv_transpose1d_formula():
    for (cin, cout, k, u, T) in [(6, 4, 16, 8, 5), (4, 3, 4, 2, 17), (3, 5, 8, 4, 1), (2, 2, 6, 2, 9)]:
        x, w, b = _r(2, cin, T, seed=4), _r(cin, cout, k, seed=5), _r(cout, seed=6)
        p = (k - u) // 2
        ref = F.conv_transpose1d(x, w, b, stride=u, padding=p).numpy()
        out = c_ref.conv_transpose1d(x.numpy(), w.numpy(), b.numpy(), u, p)
        assert out.shape == ref.shape and np.abs(out - ref).max() <= 1e-5


def test_weight_norm_fold():
    g, v = _r(6, 1, 1, seed=7).abs() + 0.1, _r(6, 5, 3, seed=8)
    ref = vo.fold_weight_norm(g, v).numpy()
    assert np.abs(c_ref.fold_weight_norm(g.numpy(), v.numpy()) - ref).max() <= 1e-6
    # ConvTranspose1d: dim 0 is C_in -- same routine, different leading dim
    m = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(4, 3, 4, 2))
    sd = m.state_dict()
    w = c_ref.fold_weight_norm(sd["weight_g"].numpy(), sd["weight_v"].numpy())
    assert np.abs(w - m.weight.detach().numpy()).max() <= 1e-6


def test_activation1d_formula():
    x = _r(2, 3, 29, seed=9) * 1.5
    al, be = _r(3, seed=10) * 0.3, _r(3, seed=11) * 0.3
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    ref = vo.activation1d(x, al, be, True).numpy()
    out = c_ref.activation1d(x.numpy(), torch.exp(al).numpy(), torch.exp(be).numpy(), f.numpy(), f.numpy())
    assert np.abs(out - ref).max() <= 2e-6
    x1 = _r(1, 3, 1, seed=12)
    ref = vo.activation1d(x1, al, None, False).numpy()
    out = c_ref.activation1d(x1.numpy(), al.numpy(), al.numpy(), f.numpy(), f.numpy())
    assert np.abs(out - ref).max() <= 2e-6


def test_stft_mel_formula():
    from types import SimpleNamespace as NS

    pp = NS(sample_rate=8000, n_fft=128, win_size=128, hop_size=32, n_mel=20, fmin=0, fmax=None)
    y = (torch.rand(2, 640, generator=torch.Generator().manual_seed(13)) * 2 - 1) * 0.7
    win = vo.hann_periodic(128).numpy()
    re, im = c_ref.stft(y.numpy(), 128, 32, (128 - 32) // 2, win)
    _, _, rre, rim = vo.amplitude_phase_spectrum(y, pp)
    assert np.abs(re - rre.numpy()).max() <= 2e-5 and np.abs(im - rim.numpy()).max() <= 2e-5
    basis = vo.mel_filterbank(8000, 128, 20, 0, None)
    mel = c_ref.logmel(re, im, basis, 1e-6, 1e-5)
    assert np.abs(mel - vo.mel_spectrogram_torch(y, pp).numpy()).max() <= 2e-4
    # TacotronSTFT padding (n_fft/2) and frame count L/hop + 1
    re, im = c_ref.stft(y.numpy(), 128, 32, 64, win)
    mag, _ = vo.taco_stft_transform(y, 128, 32, 128)
    assert re.shape == tuple(mag.shape)
    assert np.abs(np.sqrt(re**2 + im**2) - mag.numpy()).max() <= 2e-5


def test_resblock_composed_from_c_ops():
    """ResBlock1 (hifigan.py:93-100) composed from the C primitives equals the oracle's."""
    from oracle import synth

    C, k = 8, 3
    shapes = {}
    for p in range(3):
        for nm in ("convs1", "convs2"):
            shapes[f"rb.{nm}.{p}.bias"] = (C,)
            shapes[f"rb.{nm}.{p}.weight_g"] = (C, 1, 1)
            shapes[f"rb.{nm}.{p}.weight_v"] = (C, C, k)
    sd = synth.synth_state_dict(shapes, 5)
    x = _r(1, C, 40, seed=14)
    ref = vo.resblock1(sd, "rb", x, k, [1, 3, 5], torch.float32).numpy()
    cur = x.numpy()
    for p, d in enumerate([1, 3, 5]):
        w1 = c_ref.fold_weight_norm(sd[f"rb.convs1.{p}.weight_g"].numpy(), sd[f"rb.convs1.{p}.weight_v"].numpy())
        w2 = c_ref.fold_weight_norm(sd[f"rb.convs2.{p}.weight_g"].numpy(), sd[f"rb.convs2.{p}.weight_v"].numpy())
        xt = np.where(cur > 0, cur, cur * np.float32(0.1))
        xt = c_ref.conv1d(xt, w1, sd[f"rb.convs1.{p}.bias"].numpy(), d, vo.get_padding(k, d))
        xt = np.where(xt > 0, xt, xt * np.float32(0.1))
        xt = c_ref.conv1d(xt, w2, sd[f"rb.convs2.{p}.bias"].numpy(), 1, vo.get_padding(k, 1))
        cur = xt + cur
    assert np.abs(cur - ref).max() <= 1e-5
