"""GPU parity: APNet drop-in (frame-rate ResBlock branches on the conv kernels, polar head, "same"-padded ISTFT)
vs golden vectors of the reference class.  Tolerances: logamp / audio 1e-4 max-abs; rea / imag 1e-4 relative to
the largest magnitude; phase compared modulo 2*pi away from vanishing (R, I) (atan2's branch cut)."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_precision")]
HERE = os.path.dirname(os.path.abspath(__file__))
HP = dict(vo.apnet_recipe_hp(), ASP_channel=96, PSP_channel=64)
PP = dict(n_mel=80, n_fft=1024, hop_size=256, win_size=1024, extract_amplitude_phase=True)


@pytest.mark.parametrize("tag", ["b1_t10", "b2_t27"])
def test_apnet_golden(tag):
    from amphion_amd.models.vocoders.gan.gan_vocoder_inference import vocoder_inference
    from amphion_amd.models.vocoders.gan.generator.apnet import APNet

    g = np.load(os.path.join(HERE, "golden", "golden_apnet.npz"))
    cfg = NS(preprocess=NS(**PP), model=NS(apnet=NS(**HP)))
    m = APNet(cfg)
    m.load_state_dict(synth.synth_state_dict(synth.apnet_param_shapes(80, 1024, HP), 321, g_gain=0.45))
    m = m.cuda().eval()
    mel = torch.from_numpy(g[f"apnet_{tag}_mel"])
    with torch.no_grad():
        logamp, pha, rea, imag, audio = [t.cpu().numpy() for t in m(mel.cuda())]
    ref = {n: g[f"apnet_{tag}_{n}"] for n in ("logamp", "pha", "rea", "imag", "audio")}
    assert audio.shape == ref["audio"].shape
    assert np.abs(logamp - ref["logamp"]).max() <= 1e-4
    scale = max(np.abs(ref["rea"]).max(), np.abs(ref["imag"]).max())
    assert np.abs(rea - ref["rea"]).max() <= 1e-4 * scale
    assert np.abs(imag - ref["imag"]).max() <= 1e-4 * scale
    dp = np.angle(np.exp(1j * (pha - ref["pha"])))
    assert np.abs(dp).max() <= 1e-3          # modulo 2*pi; the synthetic net keeps |R + iI| away from 0
    assert np.abs(audio - ref["audio"]).max() <= 1e-4
    out = vocoder_inference(cfg, m, mel, device="cuda")            # extract_amplitude_phase branch (:26-27)
    assert np.abs(out.numpy() - ref["audio"][:, 0]).max() <= 1e-4
