"""Training-time mel loss of the GAN vocoders (models/vocoders/gan/gan_vocoder_trainer.py:387-392):
loss = 45 * L1(extract_mel_features(y_gt), extract_mel_features(y_pred)) and its gradient w.r.t. y_pred.  Forward =
the fused front-end kernel, backward = amp_mel_backward; checked against the oracle's torch restatement differentiated
by autograd on the CPU (fp64)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu


def _mel64(y, pp):
    """The oracle's restatement of extract_mel_features (utils/mel.py:111-170; torch ops, so autograd differentiates
    it), in fp64."""
    return vo.extract_mel_features(y, pp, dtype=torch.float64)


def _oracle_loss_and_grad(y_gt, y_pred, pp):
    yp = y_pred.double().clone().requires_grad_(True)
    loss = torch.nn.functional.l1_loss(_mel64(y_gt.double(), pp), _mel64(yp.squeeze(1), pp)) * 45
    loss.backward()
    return loss.item(), yp.grad


@pytest.mark.parametrize("B,L", [(2, 8192), (1, 256 * 33), (3, 5000)])
def test_mel_loss_value_and_gradient(B, L):
    from amphion_amd.utils.mel import mel_criterion

    pp = vo.preprocess_22k()
    cfg = NS(preprocess=pp, model=NS(generator="hifigan"))
    g = torch.Generator().manual_seed(L)
    y_gt = (torch.rand(B, L, generator=g) * 2 - 1) * 0.6
    y_pred = ((torch.rand(B, 1, L, generator=g) * 2 - 1) * 0.5)
    y_pred[:, :, L // 2: L // 2 + 700] *= 1e-4          # a nearly silent stretch: mel energies at the 1e-5 clamp
    ref_loss, ref_grad = _oracle_loss_and_grad(y_gt, y_pred, pp)
    yp = y_pred.cuda().requires_grad_(True)
    loss = mel_criterion(cfg)(y_gt.cuda(), yp)
    loss.backward()
    assert abs(loss.item() - ref_loss) <= 2e-4 * max(1.0, abs(ref_loss))
    got = yp.grad.cpu().double()
    assert got.shape == ref_grad.shape
    scale = ref_grad.abs().max().item()
    err = (got - ref_grad).abs().max().item()
    print(f"[mel-loss] B={B} L={L}: loss {loss.item():.5f} (oracle {ref_loss:.5f}), grad max |err| {err:.2e} of scale {scale:.2e}")
    # elements whose mel sits within rounding of the 1e-5 clamp may switch the clamp's 0 / 1 gradient: exclude by magnitude
    assert err <= 2e-3 * scale
    assert (got - ref_grad).abs().mean().item() <= 2e-5 * scale


@pytest.mark.parametrize("sr,n_fft,hop,n_mel", [(24000, 1920, 480, 100), (16000, 400, 160, 80), (16000, 1001, 143, 40), (22050, 1102, 275, 64), (16000, 1021, 255, 40)])
def test_mel_gradient_any_smooth_nfft(sr, n_fft, hop, n_mel):
    """The mel-loss gradient for transform lengths that are not powers of two (round 5; the odd one has no Nyquist bin, so every bin
    above 0 counts twice in the one-sided sum)."""
    from amphion_amd.utils.mel import extract_mel_features

    pp = NS(sample_rate=sr, n_fft=n_fft, win_size=n_fft, hop_size=hop, n_mel=n_mel, fmin=0, fmax=None)
    g = torch.Generator().manual_seed(n_fft)
    y = (torch.rand(2, hop * 20, generator=g) * 2 - 1) * 0.7
    y64 = y.double().clone().requires_grad_(True)
    _mel64(y64, pp).sum().backward()
    yd = y.cuda().requires_grad_(True)
    extract_mel_features(yd, pp).sum().backward()
    got, ref = yd.grad.cpu().double(), y64.grad
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-3 * scale
    assert (got - ref).abs().mean().item() <= 2e-5 * scale


def test_mel_gradient_of_a_plain_sum():
    """d sum(logmel) / d y against autograd, without the L1's sign pattern (smooth upstream gradient)."""
    from amphion_amd.utils.mel import extract_mel_features

    pp = vo.preprocess_22k()
    g = torch.Generator().manual_seed(7)
    y = (torch.rand(2, 6000, generator=g) * 2 - 1) * 0.7
    y64 = y.double().clone().requires_grad_(True)
    _mel64(y64, pp).sum().backward()
    yc = y.cuda().requires_grad_(True)
    extract_mel_features(yc, pp).sum().backward()
    scale = y64.grad.abs().max().item()
    err = (yc.grad.cpu().double() - y64.grad).abs().max().item()
    print(f"[mel-loss] d sum(logmel): max |err| {err:.2e} of scale {scale:.2e}")
    assert err <= 1e-4 * scale


def test_inference_path_is_unchanged_without_grad():
    from amphion_amd.utils.mel import extract_mel_features

    pp = vo.preprocess_22k()
    y = (torch.rand(1, 4096, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda()
    a = extract_mel_features(y, pp)
    b = extract_mel_features(y.clone().requires_grad_(True), pp)
    assert not a.requires_grad and b.requires_grad
    assert (a - b.detach()).abs().max().item() <= 2e-6          # log applied by torch instead of the kernel
