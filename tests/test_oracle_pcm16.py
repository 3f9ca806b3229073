"""oracle/pcm16.py against hand-derived known answers of the torchaudio 2.0.2 / libsox 14.4.2 conversion
(parity unpinned: neither library is available offline; see the oracle header)."""
import numpy as np

from oracle import pcm16


def test_known_answers():
    lsb = 2.0 ** -15          # one 16-bit step in float units
    cases = [
        (0.0, 0), (-0.0, 0),
        (1.0, 32767), (2.0, 32767), (np.inf, 32767),
        (-1.0, -32768), (-2.0, -32768), (-np.inf, -32768),
        (0.5, 16384), (-0.5, -16384),
        (lsb, 1), (-lsb, -1),
        (0.5 * lsb, 1),            # exactly half a step rounds UP (sox adds 1 << 15, then shifts)
        (-0.5 * lsb, 0),           # ... also for negatives: -0.5 -> 0
        (0.49 * lsb, 0), (-0.51 * lsb, -1),
        (1.5 * lsb, 2), (-1.5 * lsb, -1),
        (32767 * lsb, 32767), (32766.5 * lsb, 32767), (32766.49 * lsb, 32766),
        (1.0 - 2.0 ** -24, 32767),  # largest float below 1: saturates instead of wrapping
        (-32767.5 * lsb, -32767), (-32767.51 * lsb, -32768),
    ]
    x = np.array([c[0] for c in cases], dtype=np.float32)
    want = np.array([c[1] for c in cases], dtype=np.int16)
    got = pcm16.float_to_pcm16(x)
    assert got.dtype == np.int16
    assert got.tolist() == want.tolist()


def test_truncation_before_rounding():
    # sox_sample_t truncates x * 2^31 toward zero BEFORE the 16-bit rounding: x = -(0.5 + 2^-17) steps has
    # x * 2^31 = -32768.5 -> -32768 -> (-32768 + 32768) >> 16 = 0, not floor(-0.5000076 + 0.5) = -1
    x = np.float32(-(0.5 + 2.0 ** -17) * 2.0 ** -15)
    assert float(x) * 2.0 ** 31 == -32768.5
    assert pcm16.float_to_pcm16(np.array([x]))[0] == 0
    assert pcm16.float_to_sox_sample(np.array([x]))[0] == -32768


def test_matches_round_half_up_in_range():
    rng = np.random.default_rng(0)
    x = (rng.random(200000, dtype=np.float32) * 2.2 - 1.1).astype(np.float32)
    got = pcm16.float_to_pcm16(x).astype(np.int64)
    d = np.trunc(np.clip(x.astype(np.float64) * 2.0 ** 31, -2.0 ** 31, 2.0 ** 31 - 1))
    want = np.clip(np.floor((d + 32768.0) / 65536.0), -32768, 32767).astype(np.int64)
    assert (got == want).all()


def test_lens_zero_the_tail_and_nan():
    x = np.full((2, 6), 0.25, dtype=np.float32)
    y = pcm16.float_to_pcm16(x, lens=[4, 0])
    assert y.tolist() == [[8192, 8192, 8192, 8192, 0, 0], [0] * 6]
    assert pcm16.float_to_pcm16(np.array([np.nan], dtype=np.float32))[0] == -32768


def test_c_restatement_agrees_bit_for_bit():
    # oracle/c/vocoder_ref.c: ref_pcm16 -- the same conversion written with C integer types
    from oracle import c_ref

    rng = np.random.default_rng(1)
    x = (rng.random(300000, dtype=np.float32) * 2.4 - 1.2).astype(np.float32)
    lsb = 2.0 ** -15
    edge = np.array([0.0, 1.0, -1.0, 2.0, -2.0, np.inf, -np.inf, np.nan, 0.5 * lsb, -0.5 * lsb, 32766.5 * lsb, -32767.5 * lsb,
                     -(0.5 + 2.0 ** -17) * lsb, 1.0 - 2.0 ** -24, 3e38, -3e38, 1e-30], dtype=np.float32)
    for v in (x, edge, x.reshape(300, 1000)):
        assert np.array_equal(c_ref.pcm16(v), pcm16.float_to_pcm16(v))
