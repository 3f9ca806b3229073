"""amp_wav_to_pcm16 / amphion_amd.utils.io against oracle/pcm16.py: integer output, so the bar is BIT-EXACT."""
import wave

import numpy as np
import pytest
import torch

from oracle import pcm16 as opcm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _edge_values():
    lsb = 2.0 ** -15
    v = [0.0, -0.0, 1.0, -1.0, 2.0, -2.0, np.inf, -np.inf, np.nan, 0.5, -0.5, lsb, -lsb, 0.5 * lsb, -0.5 * lsb,
         0.49 * lsb, -0.51 * lsb, 1.5 * lsb, -1.5 * lsb, 32766.5 * lsb, 32766.49 * lsb, 1.0 - 2.0 ** -24,
         -32767.5 * lsb, -32767.51 * lsb, -(0.5 + 2.0 ** -17) * lsb, 1e-30, -1e-30, 3e38, -3e38]
    return np.array(v, dtype=np.float32)


def test_edge_values_bit_exact():
    from amphion_amd.utils.io import wav_to_pcm16
    x = _edge_values()
    got = wav_to_pcm16(torch.from_numpy(x).to(DEV)).cpu().numpy()
    assert got.dtype == np.int16
    assert got.tolist() == opcm.float_to_pcm16(x).tolist()


@pytest.mark.parametrize("B,L", [(1, 1), (1, 7), (3, 8), (2, 2049), (5, 4096), (64, 65536)])
def test_random_batches_bit_exact(B, L):
    from amphion_amd.utils.io import wav_to_pcm16
    g = torch.Generator().manual_seed(B * 100003 + L)
    x = torch.rand(B, L, generator=g) * 2.4 - 1.2          # 8 % of the samples clip
    x[:, ::97] = torch.round(x[:, ::97] * 32768.0 + 0.5) / 32768.0 - 0.5 / 32768.0   # exact half steps
    got = wav_to_pcm16(x.to(DEV)).cpu().numpy()
    assert np.array_equal(got, opcm.float_to_pcm16(x.numpy()))


def test_lengths_and_strided_rows():
    from amphion_amd.utils.io import wav_to_pcm16
    g = torch.Generator().manual_seed(5)
    big = (torch.rand(4, 5000, generator=g) * 2 - 1).to(DEV)
    view = big[:, :4099]                                    # row stride 5000, unaligned row length
    lens = [4099, 0, 17, 2048]
    got = wav_to_pcm16(view, lens).cpu().numpy()
    assert np.array_equal(got, opcm.float_to_pcm16(view.cpu().numpy(), lens))
    one = wav_to_pcm16(big[2, 3:1003])                      # 1-D, unaligned base pointer
    assert np.array_equal(one.cpu().numpy(), opcm.float_to_pcm16(big[2, 3:1003].cpu().numpy()))


def test_full_size_idempotent_on_grid():
    # size-independent property at the headline shape: values already on the 16-bit grid map to themselves
    from amphion_amd.utils.io import wav_to_pcm16
    g = torch.Generator().manual_seed(9)
    q = torch.randint(-32768, 32768, (64, 65536), generator=g, dtype=torch.int32)
    got = wav_to_pcm16((q.to(torch.float32) / 32768.0).to(DEV)).cpu()
    assert torch.equal(got.to(torch.int32), q)


def test_save_audio_files(tmp_path):
    from amphion_amd.utils.io import save_audio, save_audios
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(3, 3000, generator=g) * 1.6 - 0.8)
    lens = [3000, 1234, 2999]
    paths = [tmp_path / f"u{i}.wav" for i in range(3)]
    save_audios(paths, x.to(DEV), lens, 22050)
    for p, row, l in zip(paths, x.numpy(), lens):
        with wave.open(str(p), "rb") as w:
            assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 22050, l)
            data = np.frombuffer(w.readframes(l), dtype="<i2")
        assert np.array_equal(data, opcm.float_to_pcm16(row[:l]))
    # save_audio: turn_up scales the peak to volume_peak (utils/io.py:60-63), add_silence pads fs // 20 zeros
    p = tmp_path / "t.wav"
    save_audio(p, x[0].to(DEV), 16000, add_silence=True, turn_up=True, volume_peak=0.9)
    w32 = x[0] * (0.9 / torch.maximum(x[0].max(), x[0].min().abs()))
    with wave.open(str(p), "rb") as w:
        assert w.getnframes() == 3000 + 2 * 800
        data = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    assert not data[:800].any() and not data[-800:].any()
    assert np.array_equal(data[800:-800], opcm.float_to_pcm16(w32.numpy()))
    assert abs(int(np.abs(data).max()) - round(0.9 * 32768)) <= 1


def test_rejects_cpu_tensor():
    from amphion_amd.utils.io import wav_to_pcm16
    with pytest.raises(RuntimeError):
        wav_to_pcm16(torch.zeros(4))
