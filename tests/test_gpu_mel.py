"""GPU parity: the fused STFT/mel front-end kernel vs golden vectors of the reference's utils/mel.py
and utils/stft.py, and vs the CPU oracle on seeded audio.

Tolerances (fp32): linear spectra 2e-5 relative to the spectrum's max (FFT rounding differs from
pocketfft's); log-mel 1e-3 absolute (the log amplifies rounding of small mel energies)."""
import numpy as np
import pytest
import torch

from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu


def _wav(golden):
    y = torch.from_numpy(golden["wav_pcm16"].astype(np.float32) / 32768.0).unsqueeze(0)
    return y, torch.stack([y[0], torch.roll(y[0], 777) * 0.5])


@pytest.mark.parametrize("tag,pp", [("22k", vo.preprocess_22k()), ("24k", vo.preprocess_24k())])
def test_mel_front_end_golden(golden, tag, pp):
    from amphion_amd.utils import mel as M

    y, y2 = _wav(golden)
    out = M.extract_mel_features(y.cuda(), pp).cpu().numpy()
    ref = golden[f"mel_{tag}_extract"]
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 1e-3
    out = M.mel_spectrogram_torch(y2.cuda(), pp).cpu().numpy()
    assert out.shape == golden[f"mel_{tag}_melspec_b2"].shape
    assert np.abs(out - golden[f"mel_{tag}_melspec_b2"]).max() <= 1e-3
    lin = M.extract_linear_features(y.cuda(), pp).cpu().numpy()
    ref = golden[f"mel_{tag}_linear"]
    assert lin.shape == ref.shape
    assert np.abs(lin - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    la, ph, re, im = (t.cpu().numpy() for t in M.amplitude_phase_spectrum(y2.cuda(), pp))
    scale = max(1.0, float(np.abs(golden[f"mel_{tag}_re"]).max()))
    assert np.abs(re - golden[f"mel_{tag}_re"]).max() <= 2e-5 * scale
    assert np.abs(im - golden[f"mel_{tag}_im"]).max() <= 2e-5 * scale
    # log-amplitude: compare where the bin is not rounding noise
    big = np.exp(golden[f"mel_{tag}_logamp"]) > 1e-3
    assert np.abs(la - golden[f"mel_{tag}_logamp"])[big].max() <= 5e-2


@pytest.mark.parametrize("tag,pp", [("22k", vo.preprocess_22k()), ("24k", vo.preprocess_24k())])
def test_tacotron_stft_golden(golden, tag, pp):
    from amphion_amd.utils.stft import TacotronSTFT

    _, y2 = _wav(golden)
    taco = TacotronSTFT(pp.n_fft, pp.hop_size, pp.win_size, pp.n_mel, pp.sample_rate, pp.fmin, pp.fmax).cuda()
    mel, energy = taco.mel_spectrogram(y2.cuda())
    assert np.abs(mel.cpu().numpy() - golden[f"taco_{tag}_mel"]).max() <= 1e-3
    ref_e = golden[f"taco_{tag}_energy"]
    assert np.abs(energy.cpu().numpy() - ref_e).max() <= 2e-5 * max(1.0, ref_e.max())
    mag, phase = taco.stft_fn.transform(y2.cuda())
    ref = golden[f"taco_{tag}_mag"]
    assert mag.shape == ref.shape
    assert np.abs(mag.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, ref.max())


@pytest.mark.parametrize("B,L", [(1, 256 * 3), (3, 256 * 17), (2, 22016), (1, 1000)])
def test_mel_vs_oracle_seeded(B, L):
    from amphion_amd.utils import mel as M

    pp = vo.preprocess_22k()
    g = torch.Generator().manual_seed(L)
    y = (torch.rand(B, L, generator=g) * 2 - 1) * 0.8
    ref = vo.mel_spectrogram_torch(y, pp)
    out = M.mel_spectrogram_torch(y.cuda(), pp).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 1e-3


def test_mel_small_nfft_and_window_shorter_than_fft():
    from types import SimpleNamespace as NS

    from amphion_amd.utils import mel as M

    pp = NS(sample_rate=16000, n_fft=512, win_size=400, hop_size=160, n_mel=40, fmin=20, fmax=None)
    g = torch.Generator().manual_seed(1)
    y = (torch.rand(2, 4000, generator=g) * 2 - 1) * 0.5
    ref = vo.mel_spectrogram_torch(y, pp)
    out = M.mel_spectrogram_torch(y.cuda(), pp).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 1e-3


def test_mel_errors():
    from amphion_amd._lib import AmpError
    from amphion_amd.utils import mel as M

    pp = vo.preprocess_22k()
    with pytest.raises(RuntimeError):
        M.extract_mel_features(torch.zeros(1, 4096), pp)          # CPU tensor
    with pytest.raises(AmpError):
        M.extract_mel_features(torch.zeros(1, 100).cuda(), pp)    # shorter than the reflect padding


# ---- inverse STFT / Griffin-Lim (utils/stft.py:78-95,183-222) ---------------------------------------------
@pytest.mark.parametrize("tag", ["n1024", "n512w400"])
def test_stft_inverse_golden(tag):
    import os

    from amphion_amd.utils.stft import STFT

    gi = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_istft.npz"))
    nfft, hop, win = [int(v) for v in gi[tag + "_cfg"]]
    st = STFT(nfft, hop, win)
    wav = st.inverse(torch.from_numpy(gi[tag + "_mag"]).cuda(), torch.from_numpy(gi[tag + "_phase"]).cuda()).cpu().numpy()
    ref = gi[tag + "_wav"]
    assert wav.shape == ref.shape
    assert np.abs(wav - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_stft_forward_round_trip_and_griffin_lim():
    """transform -> inverse reconstructs the interior of the signal (hann, hop = n_fft/4: perfect
    reconstruction up to fp32), and griffin_lim keeps the reference's contract."""
    from amphion_amd.utils.stft import STFT, griffin_lim

    g = torch.Generator().manual_seed(5)
    y = (torch.rand(2, 4096, generator=g) * 2 - 1).cuda()
    st = STFT(1024, 256, 1024)
    rec = st(y)
    assert tuple(rec.shape) == (2, 1, 4096)
    assert (rec[:, 0] - y).abs().max().item() <= 1e-4
    mag, _ = st.transform(y)
    np.random.seed(0)
    sig = griffin_lim(mag, st, n_iters=3)
    assert tuple(sig.shape) == (2, 4096) and torch.isfinite(sig).all()
    m2, _ = st.transform(sig)
    # a few iterations already bring the magnitude error well below the random-phase start
    assert (m2 - mag).abs().mean().item() < 0.5 * mag.abs().mean().item()
    with pytest.raises(RuntimeError):
        st.inverse(mag.cpu(), mag.cpu())


# ---- feature extraction pipeline (processors/acoustic_extractor.py:376-449, utils/io.py:12-30) ---------------
def test_ragged_mel_batch_equals_per_utterance(tmp_path):
    """One ragged kernel launch == the reference's one-file-at-a-time extraction, bit for bit; .npy layout."""
    from types import SimpleNamespace as NS

    from amphion_amd.processors.acoustic_extractor import extract_mel_features_dataset, extract_utt_acoustic_features_vocoder
    from amphion_amd.utils.mel import extract_mel_features, extract_mel_features_batch

    pp = vo.preprocess_22k()
    g = torch.Generator().manual_seed(3)
    lens = [22016, 5000, 700, 12345]
    wavs = [(torch.rand(L, generator=g) * 2 - 1) * 0.8 for L in lens]
    mels = extract_mel_features_batch(wavs, pp)
    for w, m in zip(wavs, mels):
        solo = extract_mel_features(w.cuda().unsqueeze(0), pp)
        assert m.shape == solo.shape and torch.equal(m, solo)
        ref = vo.extract_mel_features(w.unsqueeze(0), pp)
        assert (m.cpu() - ref).abs().max().item() <= 1e-3

    pp2 = NS(**vars(pp), extract_mel=True, extract_audio=True, extract_energy=True, energy_extract_mode="from_mel",
             extract_amplitude_phase=False, mel_dir="mels", audio_dir="audios", energy_dir="energys")
    cfg = NS(preprocess=pp2)
    utts = [{"Uid": f"u{i}", "Path": ""} for i in range(len(wavs))]
    extract_mel_features_dataset(str(tmp_path / "a"), cfg, utts, wavs=wavs, batch_size=3)
    for u, w in zip(utts, wavs):
        extract_utt_acoustic_features_vocoder(str(tmp_path / "b"), cfg, u, wav_torch=w)
        ma = np.load(tmp_path / "a" / "mels" / (u["Uid"] + ".npy"))
        mb = np.load(tmp_path / "b" / "mels" / (u["Uid"] + ".npy"))
        assert ma.dtype == np.float32 and ma.ndim == 2 and ma.shape[0] == 80 and np.array_equal(ma, mb)
        assert np.array_equal(np.load(tmp_path / "a" / "audios" / (u["Uid"] + ".npy")), w.numpy())
        assert np.load(tmp_path / "b" / "energys" / (u["Uid"] + ".npy")).shape == (ma.shape[1],)
