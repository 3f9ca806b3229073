"""GPU parity: the fused STFT/mel front-end kernel vs golden vectors of the reference's utils/mel.py
and utils/stft.py, and vs the CPU oracle on seeded audio.

Tolerances (fp32): linear spectra 2e-5 relative to the spectrum's max (FFT rounding differs from
pocketfft's); log-mel 1e-4 absolute wherever the mel energy exceeds 1e-3, 1e-3 below that (the log amplifies the
rounding of small energies: d log m = dm / m); the measured errors are printed (run with -s)."""
import numpy as np
import pytest
import torch

from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu


def _check_logmel(out, ref, what):
    """log-mel parity at the north-star tolerance: <= 1e-4 where mel > 1e-3, <= 1e-3 elsewhere; prints what it measured."""
    out, ref = np.asarray(out, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert out.shape == ref.shape
    d = np.abs(out - ref)
    big = np.exp(ref) > 1e-3
    eb = float(d[big].max()) if big.any() else 0.0
    print(f"[mel] {what}: max |err| {eb:.2e} where mel > 1e-3 ({100.0 * big.mean():.0f} % of the bins), {float(d.max()):.2e} everywhere")
    assert eb <= 1e-4, what
    assert float(d.max()) <= 1e-3, what


def _wav(golden):
    y = torch.from_numpy(golden["wav_pcm16"].astype(np.float32) / 32768.0).unsqueeze(0)
    return y, torch.stack([y[0], torch.roll(y[0], 777) * 0.5])


@pytest.mark.parametrize("tag,pp", [("22k", vo.preprocess_22k()), ("24k", vo.preprocess_24k())])
def test_mel_front_end_golden(golden, tag, pp):
    from amphion_amd.utils import mel as M

    y, y2 = _wav(golden)
    out = M.extract_mel_features(y.cuda(), pp).cpu().numpy()
    _check_logmel(out, golden[f"mel_{tag}_extract"], f"extract_mel_features {tag} vs reference")
    out = M.mel_spectrogram_torch(y2.cuda(), pp).cpu().numpy()
    _check_logmel(out, golden[f"mel_{tag}_melspec_b2"], f"mel_spectrogram_torch {tag} vs reference")
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
    _check_logmel(mel.cpu().numpy(), golden[f"taco_{tag}_mel"], f"TacotronSTFT {tag} vs reference")
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
    _check_logmel(out.numpy(), ref.numpy(), f"seeded B={B} L={L} vs oracle")


@pytest.mark.parametrize("F", [2, 31, 32, 33, 65, 257])
def test_mel_frame_counts_around_the_32_frame_tile(F):
    """The n_fft = 1024 kernel works on tiles of 32 frames (8 waves x 4): frame counts on both sides of the tile
    and wave boundaries, every output (mel, magnitude, real, imaginary) against the fp64 DFT of the oracle."""
    from amphion_amd.utils import mel as M

    pp = vo.preprocess_22k()
    L = F * 256
    g = torch.Generator().manual_seed(F)
    y = (torch.rand(2, L, generator=g) * 2 - 1) * 0.9
    _check_logmel(M.mel_spectrogram_torch(y.cuda(), pp).cpu().numpy(), vo.mel_spectrogram_torch(y, pp).numpy(), f"F={F}")
    la, ph, re, im = (t.cpu().double() for t in M.amplitude_phase_spectrum(y.cuda(), pp))
    # fp64 reference of the framed, windowed real DFT
    pad = (pp.n_fft - pp.hop_size) // 2
    yp = torch.nn.functional.pad(y.double().unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    X = torch.stft(yp, pp.n_fft, hop_length=pp.hop_size, win_length=pp.win_size, window=torch.hann_window(pp.win_size, dtype=torch.float64),
                   center=False, return_complex=True)
    scale = float(X.abs().max())
    assert re.shape == X.real.shape
    er, ei = float((re - X.real).abs().max()), float((im - X.imag).abs().max())
    print(f"[mel] F={F}: spectrum max |err| re {er:.2e} im {ei:.2e} (scale {scale:.1f})")
    assert er <= 2e-6 * scale and ei <= 2e-6 * scale


def test_mel_bands_equal_the_dense_basis_sum():
    """amp_mel_desc.mel_bands_dev only skips exact zeros of the filterbank: with and without it the log-mel agrees to
    summation-order rounding, and a dense (non-triangular) basis without bands is summed over all 513 bins."""
    import ctypes

    from amphion_amd import _lib
    from amphion_amd.utils import mel as M

    pp = vo.preprocess_22k()
    g = torch.Generator().manual_seed(9)
    y = ((torch.rand(2, 256 * 40, generator=g) * 2 - 1) * 0.7).cuda()
    basis, window = M._basis_and_window(pp, y.device)
    ref = M.mel_spectrogram_torch(y, pp)
    out = M._run(y, pp, n_mel=pp.n_mel, pad_mode=0, mag_eps=1e-6, log_clip=1e-5, basis=basis.clone(), window=window)["mel"]   # a copy: no band table
    assert (out - ref).abs().max().item() <= 2e-6
    dense = (torch.rand(7, 513, generator=g) * 0.01).cuda()
    lin = M._run(y, pp, n_mel=0, pad_mode=0, mag_eps=1e-9, log_clip=0.0, want=("mag",), window=window)["mag"]
    got = M._run(y, pp, n_mel=7, pad_mode=0, mag_eps=1e-9, log_clip=0.0, basis=dense, window=window)["mel"]
    want = torch.einsum("mk,bkf->bmf", dense.double(), lin.double())
    assert (got.double() - want).abs().max().item() <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("sr,n_mel,fmax", [(16000, 40, 7600), (24000, 100, 12000), (44100, 128, None), (48000, 136, None), (48000, 160, None), (48000, 256, None)])
def test_mel_filter_counts_on_every_projection_path(sr, n_mel, fmax):
    """The n_fft = 1024 kernel keeps the filters' non-zero bands packed in LDS (1 664 floats, bands padded to 8): up to 128 filters a
    lane runs filters m and m + 64 as one stream, 136 (1 640 floats) add a second round read from the tables per frame, and 160 / 256
    filters on 513 bins no longer fit (1 792 / 2 184 floats) and take the loop over the global rows.  All against the oracle, and against the dense sum without a band
    table."""
    from types import SimpleNamespace as NS

    from amphion_amd.utils import mel as M

    pp = NS(sample_rate=sr, n_fft=1024, win_size=1024, hop_size=256, n_mel=n_mel, fmin=0, fmax=fmax)
    g = torch.Generator().manual_seed(n_mel)
    y = (torch.rand(2, 256 * 37 + 11, generator=g) * 2 - 1) * 0.8
    out = M.mel_spectrogram_torch(y.cuda(), pp)
    _check_logmel(out.cpu().numpy(), vo.mel_spectrogram_torch(y, pp).numpy(), f"{sr} Hz, {n_mel} filters vs oracle")
    basis, window = M._basis_and_window(pp, out.device)
    bands = M.basis_bands(basis.cpu().numpy()).numpy()
    padded = int((((bands[:, 1] - bands[:, 0]) + 7) // 8 * 8).sum())
    print(f"[mel] {n_mel} filters: {padded} packed floats ({'LDS' if padded <= 1664 else 'global rows'})")
    dense = M._run(y.cuda(), pp, n_mel=n_mel, pad_mode=0, mag_eps=1e-6, log_clip=1e-5, basis=basis.clone(), window=window)["mel"]   # a copy: no band table
    ref = M._run(y.cuda(), pp, n_mel=n_mel, pad_mode=0, mag_eps=1e-6, log_clip=1e-5, basis=basis, window=window)["mel"]
    assert (dense - ref).abs().max().item() <= 5e-6


def test_mel_small_nfft_and_window_shorter_than_fft():
    from types import SimpleNamespace as NS

    from amphion_amd.utils import mel as M

    pp = NS(sample_rate=16000, n_fft=512, win_size=400, hop_size=160, n_mel=40, fmin=20, fmax=None)
    g = torch.Generator().manual_seed(1)
    y = (torch.rand(2, 4000, generator=g) * 2 - 1) * 0.5
    ref = vo.mel_spectrogram_torch(y, pp)
    out = M.mel_spectrogram_torch(y.cuda(), pp).cpu()
    _check_logmel(out.numpy(), ref.numpy(), "n_fft 512 / win 400 (generic radix-2 kernel) vs oracle")


@pytest.mark.parametrize("sr,n_fft,win,hop,n_mel,L", [
    (24000, 1920, 1920, 480, 128, 480 * 21),      # egs/vocoder/vocos/emilia_singnet.json:15 (2^7 * 3 * 5)
    (44100, 2048, 2048, 512, 128, 512 * 9),       # power of two beyond the wave-per-frame kernel
    (48000, 4096, 4096, 1024, 128, 1024 * 7),     # the largest length: 90 KB of LDS per frame (needs the dynamic-LDS attribute)
    (16000, 400, 400, 160, 80, 160 * 33),         # 2^4 * 5^2
    (22050, 1000, 800, 250, 64, 250 * 12),        # 2^3 * 5^3, window shorter than the transform
    (16000, 882, 882, 147, 40, 147 * 20),         # 2 * 3^2 * 7^2
    (16000, 1001, 1001, 143, 40, 143 * 25),       # odd: 7 * 11 * 13
    (22050, 1102, 1102, 275, 64, 275 * 14),       # 2 * 19 * 29: two run-time radix passes (round 5, last session)
    (16000, 646, 646, 160, 40, 160 * 20),         # 2 * 17 * 19
    (16000, 1021, 1021, 255, 40, 255 * 12),       # prime: ONE pass, a direct DFT
    (16000, 4093, 4093, 1023, 80, 1023 * 6),      # the largest prime below the limit
    (16000, 68, 68, 17, 20, 17 * 90),             # 2^2 * 17 near the lower limit
])
def test_mel_any_smooth_nfft(sr, n_fft, win, hop, n_mel, L):
    """torch.stft accepts every n_fft (utils/mel.py:145-169); amp_mel_forward does for every length in [64, 4096] (mixed-radix kernel: compile-time
    butterflies for the primes 2 .. 13, a run-time radix pass for larger ones): log-mel and the linear spectrum against the oracle."""
    from types import SimpleNamespace as NS

    from amphion_amd.utils import mel as M

    pp = NS(sample_rate=sr, n_fft=n_fft, win_size=win, hop_size=hop, n_mel=n_mel, fmin=0, fmax=None)
    g = torch.Generator().manual_seed(n_fft)
    y = (torch.rand(2, L, generator=g) * 2 - 1) * 0.7
    ref = vo.mel_spectrogram_torch(y, pp)
    out = M.mel_spectrogram_torch(y.cuda(), pp).cpu()
    _check_logmel(out.numpy(), ref.numpy(), f"n_fft {n_fft} (mixed radix) vs oracle")
    lin_ref = vo.extract_linear_features(y[:1], pp)
    lin = M.extract_linear_features(y[:1].cuda(), pp).cpu()
    assert lin.shape == lin_ref.shape
    assert (lin - lin_ref).abs().max().item() <= 2e-5 * max(1.0, lin_ref.abs().max().item())


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AMP_FUZZ_OFFSET", "0")), int(__import__("os").environ.get("AMP_FUZZ_OFFSET", "0")) + 10))
def test_mel_random_smooth_nfft(seed):
    """Seeded random transform lengths built from the primes 2 .. 13 and, in a third of the cases, one larger prime (every radix pass and their
    orders), random hop and window: the linear spectrum against the oracle (torch.stft semantics)."""
    import random

    from types import SimpleNamespace as NS

    from amphion_amd.utils import mel as M

    rng = random.Random(9000 + seed)
    while True:
        n = 1
        for p in (2, 3, 5, 7, 11, 13):
            n *= p ** rng.choice([0, 0, 1, 1, 2, 3] if p == 2 else [0, 0, 1, 1, 2] if p <= 5 else [0, 0, 0, 1])
        n *= rng.choice([1, 2, 4, 8])
        n *= rng.choice([1, 1, 1, 17, 19, 23, 31, 37, 61, 127])      # a prime factor above 13: the run-time radix pass
        if 64 <= n <= 4096:
            break
    hop = max(1, n // rng.choice([2, 3, 4, 5, 8]))
    win = n if rng.random() < 0.7 else max(8, n - 2 * rng.randrange(1, n // 4))
    pp = NS(sample_rate=16000, n_fft=n, win_size=win, hop_size=hop, n_mel=20, fmin=0, fmax=None)
    g = torch.Generator().manual_seed(seed)
    L = n + hop * rng.choice([3, 8, 17])
    y = (torch.rand(2, L, generator=g) * 2 - 1) * 0.7
    ref = vo.extract_linear_features(y[:1], pp)
    out = M.extract_linear_features(y[:1].cuda(), pp).cpu()
    assert out.shape == ref.shape, (n, hop, win)
    err = (out - ref).abs().max().item()
    assert err <= 3e-5 * max(1.0, ref.abs().max().item()), f"n_fft={n} hop={hop} win={win}: {err:.2e}"


def test_mel_nfft_outside_the_range_is_refused():
    from types import SimpleNamespace as NS

    from amphion_amd._lib import AmpError
    from amphion_amd.utils import mel as M

    for n in (32, 4100):
        pp = NS(sample_rate=16000, n_fft=n, win_size=n, hop_size=n // 4, n_mel=8, fmin=0, fmax=None)
        with pytest.raises(AmpError, match=r"\[64, 4096\]"):
            M.mel_spectrogram_torch(torch.zeros(1, 3 * 4100).cuda(), pp)


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


@pytest.mark.parametrize("nfft,hop,win", [(1920, 480, 1920), (400, 100, 400), (1000, 250, 800), (1001, 143, 1001), (4096, 1024, 4096),
                                          (1102, 275, 1102), (1021, 255, 1021), (646, 160, 600)])
def test_stft_inverse_any_smooth_nfft(nfft, hop, win):
    """STFT.transform / STFT.inverse (utils/stft.py:152-222) for lengths that are not powers of two (round 5: mixed-radix kernels in both
    directions; an odd length has no Nyquist bin): against the oracle's restatement of the conv-basis formulation."""
    from amphion_amd.utils.stft import STFT

    g = torch.Generator().manual_seed(nfft)
    y = (torch.rand(2, hop * 24, generator=g) * 2 - 1) * 0.8
    st = STFT(nfft, hop, win)
    mag, phase = st.transform(y.cuda())
    rmag, rphase = vo.taco_stft_transform(y, nfft, hop, win)
    assert mag.shape == rmag.shape
    assert (mag.cpu() - rmag).abs().max().item() <= 2e-5 * max(1.0, rmag.abs().max().item())
    wav = st.inverse(rmag.cuda(), rphase.cuda()).cpu()
    ref = vo.taco_stft_inverse(rmag, rphase, nfft, hop, win)
    assert wav.shape == ref.shape
    assert (wav - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


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
        _check_logmel(m.cpu().numpy(), ref.numpy(), f"ragged batch item of {w.shape[0]} samples vs oracle")

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


def test_out_of_range_audio_is_reported_without_stalling(capsys):
    """utils/mel.py:21-24 prints when the audio leaves [-1, 1]; here the notice comes from a later call / an explicit
    flush (the min / max are read back asynchronously)."""
    from amphion_amd.utils import mel as M

    pp = vo.preprocess_22k()
    y = torch.zeros(1, 4096).cuda()
    y[0, 100] = 1.5
    y[0, 200] = -2.0
    M.flush_range_warnings()
    capsys.readouterr()
    M.extract_mel_features(y, pp)
    M.flush_range_warnings()
    out = capsys.readouterr().out
    assert "max value is" in out and "1.5" in out and "min value is" in out and "-2.0" in out
    M.extract_mel_features(y * 0.1, pp)
    M.flush_range_warnings()
    assert capsys.readouterr().out == ""


# ---- the transform lengths round 5 added, against outputs of the REAL reference (tests/golden/make_golden_nfft.py) -------------
NFFT_TAGS = ["n1920", "n2048", "n1001", "n1021", "n400w320"]


@pytest.fixture(scope="module")
def gn():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_nfft.npz"))


@pytest.mark.parametrize("tag", NFFT_TAGS)
def test_mel_front_end_new_lengths_golden(gn, tag):
    """extract_mel_features / extract_linear_features / amplitude_phase_spectrum (utils/mel.py:20-52,111-170) at n_fft 1920 / 2048 / 1001 / 1021 / 400:
    the HIP front end against the reference's own outputs."""
    from test_oracle_nfft import nfft_case

    from amphion_amd.utils import mel as M

    pp, y = nfft_case(gn, tag)
    _check_logmel(M.extract_mel_features(y.cuda(), pp).cpu().numpy(), gn[f"{tag}_mel"], f"extract_mel_features {tag} vs reference")
    lin = M.extract_linear_features(y[:1].cuda(), pp).cpu().numpy()
    ref = gn[f"{tag}_linear"]
    assert lin.shape == ref.shape
    assert np.abs(lin - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    la, ph, re, im = (t.cpu().numpy() for t in M.amplitude_phase_spectrum(y.cuda(), pp))
    scale = max(1.0, float(np.abs(gn[f"{tag}_re"]).max()))
    assert np.abs(re - gn[f"{tag}_re"]).max() <= 2e-5 * scale
    assert np.abs(im - gn[f"{tag}_im"]).max() <= 2e-5 * scale
    big = np.exp(gn[f"{tag}_logamp"]) > 1e-3
    assert np.abs(la - gn[f"{tag}_logamp"])[big].max() <= 5e-2


@pytest.mark.parametrize("tag", NFFT_TAGS)
def test_stft_new_lengths_golden(gn, tag):
    """STFT.transform / STFT.inverse (utils/stft.py:152-222) at the same lengths against the reference's own outputs."""
    from test_oracle_nfft import nfft_case

    from amphion_amd.utils.stft import STFT

    pp, y = nfft_case(gn, tag)
    st = STFT(pp.n_fft, pp.hop_size, pp.win_size)
    mag, phase = st.transform(y.cuda())
    ref = gn[f"{tag}_stft_mag"]
    assert tuple(mag.shape) == ref.shape
    assert np.abs(mag.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    big = ref > 1e-2 * ref.max()
    d = np.angle(np.exp(1j * (phase.cpu().numpy() - gn[f"{tag}_stft_phase"])))
    assert np.abs(d[big]).max() <= 1e-3
    wav = st.inverse(torch.from_numpy(gn[f"{tag}_inv_mag"]).cuda(), torch.from_numpy(gn[f"{tag}_inv_phase"]).cuda()).cpu().numpy()
    ref = gn[f"{tag}_inv_wav"]
    assert wav.shape == ref.shape
    assert np.abs(wav - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


# ---- round 6: the wave-per-frame kernels off n_fft = 1024 (mel_wave_kernel<16 | 15 | 4>) on the paths the goldens do not walk ----------------
@pytest.mark.parametrize("sr,n_fft,hop,n_mel", [(44100, 2048, 512, 128), (24000, 1920, 480, 100), (16000, 512, 128, 80)])
def test_mel_wave_kernel_ragged_batch_equals_per_utterance(sr, n_fft, hop, n_mel):
    """A ragged batch through the wave-per-frame kernel (each utterance's reflection padding mirrors at ITS end, later frames of a row are left untouched)
    == every utterance alone, bit for bit; and the oracle on each."""
    from types import SimpleNamespace as NS

    from amphion_amd.utils import mel as M

    pp = NS(sample_rate=sr, n_fft=n_fft, win_size=n_fft, hop_size=hop, n_mel=n_mel, fmin=0, fmax=None)
    g = torch.Generator().manual_seed(n_fft + 1)
    lens = [hop * 43, hop * 9 + 77, n_fft + 5, hop * 21 - 3]
    wavs = [(torch.rand(L, generator=g) * 2 - 1) * 0.8 for L in lens]
    mels = M.extract_mel_features_batch(wavs, pp, device="cuda")
    for w, m in zip(wavs, mels):
        alone = M.extract_mel_features(w[None].cuda(), pp)        # (a batch of one comes back without its batch axis, as the reference's does)
        assert m.shape == alone.shape and torch.equal(m, alone)
        ref = vo.extract_mel_features(w[None], pp)
        assert m.shape == ref.shape
        _check_logmel(m.cpu().numpy(), ref.numpy(), f"n_fft {n_fft}: ragged item of {w.shape[0]} samples vs oracle")


@pytest.mark.parametrize("n_fft,hop,n_mel", [(2048, 512, 256), (1920, 480, 200), (2048, 512, 160), (2048, 511, 80), (1920, 479, 64)])
def test_mel_wave_kernel_limits_fall_back_without_changing_the_answer(n_fft, hop, n_mel):
    """What the wave-per-frame kernel does not take -- more mel channels than its LDS holds beside eight exchange buffers, an odd hop (sample pairs must be 8-byte
    pairs of ONE frame offset) -- runs on the one-workgroup-per-frame kernels: same semantics, checked against the oracle."""
    from types import SimpleNamespace as NS

    from amphion_amd.utils import mel as M

    pp = NS(sample_rate=44100, n_fft=n_fft, win_size=n_fft, hop_size=hop, n_mel=n_mel, fmin=0, fmax=None)
    g = torch.Generator().manual_seed(n_mel)
    y = (torch.rand(2, hop * 14, generator=g) * 2 - 1) * 0.7
    out = M.mel_spectrogram_torch(y.cuda(), pp).cpu()
    ref = vo.mel_spectrogram_torch(y, pp)
    _check_logmel(out.numpy(), ref.numpy(), f"n_fft {n_fft} hop {hop} n_mel {n_mel} vs oracle")


def test_mel_wave_kernel_many_frames_and_shortest_signal():
    """64 utterances x 128 frames (several workgroups per item, all eight waves busy) against the oracle on two probed items; and the shortest signal the
    reflection padding admits (every frame an edge frame)."""
    from types import SimpleNamespace as NS

    from amphion_amd.utils import mel as M

    pp = NS(sample_rate=24000, n_fft=1920, win_size=1920, hop_size=480, n_mel=128, fmin=0, fmax=12000)
    g = torch.Generator().manual_seed(77)
    y = (torch.rand(64, 480 * 128, generator=g) * 2 - 1) * 0.9
    out = M.extract_mel_features(y.cuda(), pp).cpu()
    for i in (0, 37, 63):
        ref = vo.extract_mel_features(y[i: i + 1], pp)           # (a batch of one: no batch axis)
        _check_logmel(out[i].numpy(), ref.numpy(), f"n_fft 1920, item {i} of 64")
    pad = (1920 - 480) // 2
    ys = (torch.rand(1, pad + 1 + 480, generator=g) * 2 - 1) * 0.5
    ys = ys[:, : (ys.shape[1] // 480) * 480]
    out = M.extract_mel_features(ys.cuda(), pp).cpu()
    ref = vo.extract_mel_features(ys, pp)
    _check_logmel(out.numpy(), ref.numpy(), "n_fft 1920, shortest signal")
