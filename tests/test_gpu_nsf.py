"""GPU parity: NSF-HiFiGAN drop-in vs golden vectors of the reference class (f0 accepted and, as in the reference,
without effect).  Tolerance 1e-4 max-abs."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_precision")]
HERE = os.path.dirname(os.path.abspath(__file__))
HP = dict(vo.hifigan_v1_hp(), harmonic_num=8, upsample_initial_channel=128)


@pytest.mark.parametrize("tag", ["b1_t9", "b2_t17"])
def test_nsfhifigan_golden(tag):
    from amphion_amd.models.vocoders.gan.gan_vocoder_inference import vocoder_inference
    from amphion_amd.models.vocoders.gan.generator.nsfhifigan import NSFHiFiGAN

    g = np.load(os.path.join(HERE, "golden", "golden_nsf.npz"))
    cfg = NS(preprocess=NS(n_mel=80, sample_rate=22050, hop_size=256, extract_amplitude_phase=False),
             model=NS(nsfhifigan=NS(**HP)))
    m = NSFHiFiGAN(cfg)
    m.load_state_dict(synth.synth_state_dict(synth.nsfhifigan_param_shapes(80, HP), 99, g_gain=0.6))
    m = m.cuda().eval()
    mel = torch.from_numpy(g[f"nsf_{tag}_mel"])
    f0 = torch.from_numpy(g[f"nsf_{tag}_f0"])
    ref = g[f"nsf_{tag}_wav"]
    with torch.no_grad():
        y = m(mel.cuda(), f0.cuda()).cpu().numpy()
        y0 = m(mel.cuda()).cpu().numpy()
    assert y.shape == ref.shape and np.abs(y - ref).max() <= 1e-4
    assert np.array_equal(y, y0)
    out = vocoder_inference(cfg, m, mel, f0s=f0, device="cuda")     # the reference wrapper's f0 branch (:30-36)
    assert np.abs(out.numpy() - ref[:, 0]).max() <= 1e-4


def test_short_f0_is_refused():
    """The reference crops the waveform to the f0 length when f0 has fewer frames than the mel (nsfhifigan.py:267-269);
    that crop is not built: the drop-in raises instead of returning audio of a different length."""
    from amphion_amd.models.vocoders.gan.generator.nsfhifigan import NSFHiFiGAN

    cfg = NS(preprocess=NS(n_mel=80, sample_rate=22050, hop_size=256, extract_amplitude_phase=False),
             model=NS(nsfhifigan=NS(**HP)))
    m = NSFHiFiGAN(cfg)
    m.load_state_dict(synth.synth_state_dict(synth.nsfhifigan_param_shapes(80, HP), 99, g_gain=0.6))
    m = m.cuda().eval()
    mel = torch.randn(1, cfg.preprocess.n_mel, 12).cuda()
    with pytest.raises(ValueError, match="f0"):
        m(mel, torch.zeros(1, 9).cuda())
    m(mel, torch.zeros(1, 12).cuda())
    m(mel, torch.zeros(1, 15).cuda())       # longer f0: the reference crops x_source, not x -- same output
