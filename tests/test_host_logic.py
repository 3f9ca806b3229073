"""CPU-only: host-side logic and the C-ABI surface (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
from types import SimpleNamespace as NS

import pytest
import torch

from amphion_amd import _lib
from oracle import synth
from oracle import vocoder_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "amphion_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(amp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 19
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in amphion_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert _lib.lib().amp_version() >= 100


def _desc(**over):
    hp = vo.hifigan_v1_hp()
    from amphion_amd.models.vocoders.gan.generator._engine import HipGenerator

    d = HipGenerator._fill_desc(_lib.AMP_ARCH_HIFIGAN, 80, 512, hp["upsample_rates"], hp["upsample_kernel_sizes"],
                                hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"], "1")
    for k, v in over.items():
        setattr(d, k, v)
    return d


def test_gen_create_validation_and_weight_keys():
    L = _lib.lib()
    h = ctypes.c_void_p()
    with pytest.raises(_lib.AmpError):
        _lib.check(L.amp_gen_create(ctypes.byref(_desc(arch=7)), ctypes.byref(h)))
    with pytest.raises(_lib.AmpError) as ei:  # bigvgan.py:132-135 message
        _lib.check(L.amp_gen_create(ctypes.byref(_desc(arch=_lib.AMP_ARCH_BIGVGAN, activation=0)), ctypes.byref(h)))
    assert "activation incorrectly specified" in str(ei.value)
    with pytest.raises(_lib.AmpError):
        _lib.check(L.amp_gen_create(ctypes.byref(_desc(n_stages=0)), ctypes.byref(h)))

    _lib.check(L.amp_gen_create(ctypes.byref(_desc()), ctypes.byref(h)))
    try:
        assert L.amp_gen_hop(h) == 256
        w = torch.zeros(512, 80, 7)
        shp = (ctypes.c_int64 * 3)(512, 80, 7)
        _lib.check(L.amp_gen_set_weight(h, b"conv_pre.weight_v", ctypes.c_void_p(w.data_ptr()), shp, 3))
        _lib.check(L.amp_gen_set_weight(h, b"module.conv_pre.weight_v", ctypes.c_void_p(w.data_ptr()), shp, 3))
        with pytest.raises(_lib.AmpError) as ei:
            _lib.check(L.amp_gen_set_weight(h, b"conv_pre.nonsense", ctypes.c_void_p(w.data_ptr()), shp, 3))
        assert "unexpected key" in str(ei.value)
        bad = (ctypes.c_int64 * 3)(512, 81, 7)
        with pytest.raises(_lib.AmpError) as ei:
            _lib.check(L.amp_gen_set_weight(h, b"conv_pre.weight_v", ctypes.c_void_p(w.data_ptr()), bad, 3))
        assert "shape" in str(ei.value)
        if L.amp_device_count() == 0:
            # no CPU fallback: finalize/forward refuse loudly without a device
            with pytest.raises(_lib.AmpError):
                _lib.check(L.amp_gen_finalize(h))
            with pytest.raises(_lib.AmpError):
                _lib.check(L.amp_gen_forward(h, ctypes.c_void_p(8), None, 1, 4, ctypes.c_void_p(8), ctypes.c_void_p(8), 1 << 30, None))
    finally:
        L.amp_gen_destroy(h)


def test_conv_create_refuses_without_device():
    L = _lib.lib()
    if L.amp_device_count() != 0:
        pytest.skip("GPU present")
    w = torch.zeros(4, 4, 3)
    h = ctypes.c_void_p()
    with pytest.raises(_lib.AmpError) as ei:
        _lib.check(L.amp_conv_create(0, 4, 4, 3, 1, 1, 1, ctypes.c_void_p(w.data_ptr()), None, ctypes.byref(h)))
    assert "no CPU fallback" in str(ei.value)


def test_fused_wn_entry_points_validate_without_device():
    """amp_conv_create_gated / amp_wn_forward (the fused WN layer): argument checks and the no-GPU refusal."""
    L = _lib.lib()
    w = torch.zeros(2 * 64, 64, 5)
    b = torch.zeros(2 * 64)
    h = ctypes.c_void_p()
    if L.amp_device_count() == 0:
        with pytest.raises(_lib.AmpError) as ei:
            _lib.check(L.amp_conv_create_gated(64, 5, 1, 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
        assert "no CPU fallback" in str(ei.value)
    with pytest.raises(_lib.AmpError):
        _lib.check(L.amp_conv_create_gated(64, 5, 1, 2, None, None, ctypes.byref(h)))          # null weight
    with pytest.raises(_lib.AmpError):
        _lib.check(L.amp_wn_forward(None, None, 0, None, None, 0, None, 1, 8, None, None, None))  # nothing to run
    assert L.amp_set_small_conv(1) == 0


def test_kernel_policy_switches_validate_without_device():
    """The A/B switches of round 2's third session are plain host state: callable without a GPU, -1 restores the default,
    out-of-range modes are refused with a message."""
    L = _lib.lib()
    assert L.amp_version() >= 122
    for mode in (0, 1, 2, 3, -1):
        assert L.amp_set_conv_blk(mode) == 0
    with pytest.raises(_lib.AmpError) as ei:
        _lib.check(L.amp_set_conv_blk(4))
    assert "amp_set_conv_blk" in str(ei.value)
    for on in (0, 1, -1):
        assert L.amp_set_conv_rg_fast(on) == 0
        assert L.amp_set_pingpong(on) == 0


def test_round4_switches_validate_without_device():
    """amp_version 141; the concurrent-resblock and attention switches are plain host state with range checks; a handle-less
    amp_gen_prepare_streams is refused with a message instead of touching a device."""
    L = _lib.lib()
    assert L.amp_version() >= 142
    for mode in (0, 1, -1):
        assert L.amp_set_resblock_streams(mode) == 0
    assert L.amp_set_pair_strips(0) == 0 and L.amp_set_pair_strips(-1) == 0
    with pytest.raises(_lib.AmpError) as ei:           # the four-wave strips left with ABI 142
        _lib.check(L.amp_set_pair_strips(1))
    assert "amp_set_pair_strips" in str(ei.value)
    for gone in ("amp_conv_act_forward", "amp_set_fuse_act", "amp_set_wn_layer_fusion"):     # removed with their kernels
        assert not hasattr(L, gone)
    with pytest.raises(_lib.AmpError) as ei:
        _lib.check(L.amp_set_resblock_streams(2))
    assert "amp_set_resblock_streams" in str(ei.value)
    for on in (0, 1):
        assert L.amp_set_rel_attention_tiled(on) == 0
    with pytest.raises(_lib.AmpError) as ei:
        _lib.check(L.amp_gen_prepare_streams(None))
    assert "amp_gen_prepare_streams" in str(ei.value)


def test_forward_graphed_cache_is_not_module_state():
    """The graph cache and its lock live in a weak-keyed side table of _engine: a generator pickles / deep-copies without them
    (a lock or a CUDAGraph in __dict__ would make torch.save(model) raise)."""
    import copy
    import pickle
    from types import SimpleNamespace as NS

    from amphion_amd.models.vocoders.gan.generator import _engine
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = dict(resblock="1", upsample_rates=[4, 4], upsample_kernel_sizes=[8, 8], upsample_initial_channel=32,
              resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 3], [1, 3]])
    m = HiFiGAN(NS(preprocess=NS(n_mel=8, hop_size=16), model=NS(hifigan=NS(**hp))))
    _engine._graph_caches.setdefault(m, {})["epoch"] = 0
    m2 = copy.deepcopy(m)
    assert m2 not in _engine._graph_caches and len(pickle.dumps(m)) > 0
    assert _engine.HipGenerator.GRAPH_MAX_FRAMES <= 4096 and _engine.HipGenerator.GRAPH_BUCKET_FRAMES >= 8


def test_c_abi_from_plain_c(tmp_path):
    """include/amphion_hip.h compiles as C99 and a plain-C program links against the library and exercises the
    no-GPU paths (tests/c/abi_smoke.c) -- the boundary is a C ABI, not a Python extension."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = tmp_path / "abi_smoke"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe), "-L", libdir, "-lamphion_hip",
                    "-Wl,-rpath," + libdir], check=True)
    env = dict(os.environ)
    try:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    except Exception:
        pass
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi ok" in r.stdout


def test_mel_num_frames():
    L = _lib.lib()
    d = _lib.amp_mel_desc(1024, 1024, 256, 80, 0, 1e-9, 1e-5)
    assert L.amp_mel_num_frames(ctypes.byref(d), 22016) == 86
    assert L.amp_mel_num_frames(ctypes.byref(d), 256 * 40) == 40
    d.pad_mode = 1
    assert L.amp_mel_num_frames(ctypes.byref(d), 256 * 40) == 41    # TacotronSTFT: L/hop + 1


def _model(seed=1234):
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = vo.hifigan_v1_hp()
    cfg = NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp)))
    m = HiFiGAN(cfg)
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), seed)
    m.load_state_dict(sd)
    return m, sd


def test_state_dict_round_trip_both_forms():
    m, sd = _model()
    out = m.state_dict()
    assert list(out.keys()) == list(sd.keys())
    assert all(torch.equal(out[k], sd[k]) for k in sd)
    # folded form: keys collapse to .weight exactly like torch.nn.utils.remove_weight_norm
    m.remove_weight_norm()
    folded = m.state_dict()
    assert "conv_pre.weight" in folded and "conv_pre.weight_g" not in folded
    w = vo.fold_weight_norm(sd["ups.0.weight_g"], sd["ups.0.weight_v"])
    assert (folded["ups.0.weight"] - w).abs().max().item() <= 1e-6
    with pytest.raises(ValueError):
        m.conv_pre.remove_weight_norm()
    # a weight-normed module accepts a folded checkpoint and vice versa
    m2, _ = _model(seed=7)
    m2.load_state_dict(folded)
    assert torch.equal(m2.state_dict()["conv_post.weight"], folded["conv_post.weight"])
    m.load_state_dict(sd)
    assert "conv_pre.weight_g" in m.state_dict()
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in sd.items() if k != "conv_post.bias"})


def test_signature_tracks_parameter_changes():
    m, sd = _model()
    s0 = m._amp_signature()
    assert m._amp_signature() == s0
    with torch.no_grad():
        m.conv_post.bias.add_(1.0)
    assert m._amp_signature() != s0
    s1 = m._amp_signature()
    m.load_state_dict(sd)
    assert m._amp_signature() != s1
    assert m.hop_factor == 256


def test_per_forward_change_detection_without_the_tree_walk():
    """_amp_unchanged() (what every forward checks instead of the full signature) sees in-place updates, re-allocated
    storage and re-registered Parameter objects."""
    m, sd = _model()
    m._amp_snap = m._amp_snapshot()
    assert m._amp_unchanged()
    with torch.no_grad():
        m.conv_post.bias.add_(1.0)                      # optimizer step / copy_: version bump
    assert not m._amp_unchanged()
    m._amp_snap = m._amp_snapshot()
    m.load_state_dict(sd)                               # in-place copy_ into every tensor
    assert not m._amp_unchanged()
    m._amp_snap = m._amp_snapshot()
    m.double()                                          # _apply: new storage behind the same Parameter objects
    assert not m._amp_unchanged()
    m.float()
    m._amp_snap = m._amp_snapshot()
    m.conv_post.bias = torch.nn.Parameter(m.conv_post.bias.detach().clone())   # a new Parameter object
    assert not m._amp_unchanged()
    m._amp_snap = m._amp_snapshot()
    assert m._amp_unchanged()


def test_pad_mels_to_tensors_matches_reference_semantics():
    from amphion_amd.utils.util import pad_mels_to_tensors

    mels = [torch.full((3, T), float(i + 1)) for i, T in enumerate((4, 7, 2, 5, 6))]
    t, f = pad_mels_to_tensors(mels, None)
    assert len(t) == 1 and tuple(t[0].shape) == (5, 3, 7) and f[0].tolist() == [4, 7, 2, 5, 6]
    assert t[0][2, :, 2:].abs().sum() == 0 and t[0][2, :, :2].eq(3).all()
    t, f = pad_mels_to_tensors(mels, 2)
    assert [tuple(x.shape) for x in t] == [(2, 3, 7), (2, 3, 5), (1, 3, 6)]
    assert [x.tolist() for x in f] == [[4, 7], [2, 5], [6]] and f[0].dtype == torch.int32
    t, f = pad_mels_to_tensors(mels[:4], 2)
    assert [tuple(x.shape) for x in t] == [(2, 3, 7), (2, 3, 5)]


def test_registry_surface():
    from amphion_amd.models.vocoders import vocoder_inference as vi
    from amphion_amd.models.vocoders.gan import gan_vocoder_inference as gi

    assert set(vi._vocoders) == {"hifigan", "bigvgan", "melgan", "nsfhifigan", "apnet"}
    assert vi._vocoder_forward_funcs["hifigan"] is gi.vocoder_inference
    assert vi._vocoder_infer_funcs["bigvgan"] is gi.synthesis_audios
    import inspect

    assert list(inspect.signature(gi.vocoder_inference).parameters)[:6] == ["cfg", "model", "mels", "f0s", "device", "fast_inference"]
    assert list(inspect.signature(vi.synthesis).parameters) == ["cfg", "vocoder_weight_file", "n_samples", "pred", "f0s", "batch_size", "fast_inference"]


def test_load_audio_torch_and_save_feature(tmp_path):
    """processors drop-in, host side only: PCM16 wav -> float32 [-1, 1] at the config rate; .npy layout of utils/io.py."""
    import wave

    import numpy as np

    from amphion_amd.processors.acoustic_extractor import load_audio_torch, save_feature

    sr = 24000
    t = np.arange(sr // 2) / sr
    pcm = (0.5 * np.sin(2 * np.pi * 440 * t) * 32767).astype(np.int16)
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())
    x, fs = load_audio_torch(str(p), 24000)
    assert fs == 24000 and x.dtype.is_floating_point and x.shape[0] == pcm.shape[0]
    assert abs(float(x.abs().max()) - 0.5) < 1e-3
    y, fs2 = load_audio_torch(str(p), 22050)
    assert fs2 == 22050 and abs(y.shape[0] - round(pcm.shape[0] * 22050 / 24000)) <= 1
    save_feature(str(tmp_path / "out"), "mels", "uid0", np.zeros((80, 7), np.float32))
    assert np.load(tmp_path / "out" / "mels" / "uid0.npy").shape == (80, 7)
    save_feature(str(tmp_path / "out"), "mels", "uid0", np.ones((80, 7), np.float32), overrides=False)
    assert np.load(tmp_path / "out" / "mels" / "uid0.npy").sum() == 0


def test_generator_is_inference_only():
    # no backward through the HIP kernels.  The forward runs whenever the reference's would (a CPU tensor is refused either way: no
    # fallback); what fails is a BACKWARD through the generator -- the output of a call autograd would have recorded (input requires
    # grad; training mode + trainable parameters: a GAN trainer through the integration patch) carries a grad_fn that raises
    # (tests/test_gpu_generator.py::test_backward_through_the_generator_raises runs that on the GPU)
    import warnings
    from types import SimpleNamespace as NS

    import pytest
    import torch

    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = dict(resblock="1", upsample_rates=[2, 2], upsample_kernel_sizes=[4, 4], upsample_initial_channel=32,
              resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1, 3, 5]])
    m = HiFiGAN(NS(preprocess=NS(n_mel=8), model=NS(hifigan=NS(**hp))))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 8, 4, requires_grad=True))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 8, 4))                             # default training mode, autograd on: an ordinary (CPU: refused) call
    from amphion_amd.models.vocoders.gan.generator._engine import _InferenceOnly
    w = torch.ones(3, requires_grad=True)
    y = _InferenceOnly.apply(torch.zeros(2, 3), w)
    assert y.requires_grad
    with pytest.raises(RuntimeError, match="inference-only"):
        y.sum().backward()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # under no_grad, frozen, or in eval: nothing to say
        with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
            m(torch.zeros(1, 8, 4))
        m.requires_grad_(False)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(torch.zeros(1, 8, 4))
        m2 = HiFiGAN(NS(preprocess=NS(n_mel=8), model=NS(hifigan=NS(**hp)))).eval()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m2(torch.zeros(1, 8, 4))


def test_bench_relaunch_command_and_cli_guards():
    """`python bench.py --gpus N` outside a launcher re-executes itself through torch.distributed.run on 127.0.0.1 with
    N processes (the driver's own launch line); without a GPU it exits with a message, not a traceback."""
    import importlib.util
    import subprocess
    import sys

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.relaunch_command(["--gpus", "4", "--steps", "3"], 4)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "ROCm device" in (r.stderr + r.stdout) and "Traceback" not in r.stderr


def test_few_host_threads_scope_and_cpu_budget():
    """The list / batch entry points cap torch's intra-op threads while they pad and crop on the host (a 128-thread OpenMP pool
    under a 16-CPU cgroup quota got the whole process throttled: profiles/r3_o_list_api_cgroup_throttle.txt) and restore the
    setting afterwards, also when the body raises or scopes nest.  The setting is the calling thread's own (omp_set_num_threads):
    a scope open in one thread is not visible in another (code health 12)."""
    import threading

    import torch

    from amphion_amd.utils.util import _cpu_budget, few_host_threads

    assert 1 <= _cpu_budget() <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    with few_host_threads(4) as scope:
        assert torch.get_num_threads() <= max(1, min(4, before))
        assert 1 <= scope.n <= 4
        inner_seen = torch.get_num_threads()
        with few_host_threads(2):
            assert torch.get_num_threads() <= inner_seen
        assert torch.get_num_threads() == inner_seen
        seen = []
        t = threading.Thread(target=lambda: seen.append(torch.get_num_threads()))
        t.start()
        t.join(10)
    assert torch.get_num_threads() == before
    assert seen and seen[0] >= inner_seen                 # the other thread kept its own (default) pool size
    try:
        with few_host_threads(2):
            raise KeyError("boom")
    except KeyError:
        pass
    assert torch.get_num_threads() == before
    torch.set_num_threads(1)                  # already below the cap: left alone
    try:
        with few_host_threads(4):
            assert torch.get_num_threads() == 1
        assert torch.get_num_threads() == 1
    finally:
        torch.set_num_threads(before)


def test_mel_range_ring_reports_and_skips_calls_that_never_launched():
    """amphion_amd.utils.mel._RangeRing bookkeeping without a GPU: a slot whose copy landed is decoded (min below -1 / max above 1
    printed, in-range values silent); a call that raised before its launch (sequence word never written) is skipped after the
    blocking wait instead of holding every later notice back."""
    import numpy as np

    from amphion_amd.utils import mel as M

    ring = M._RangeRing.__new__(M._RangeRing)
    ring.host_np = np.zeros((M._RANGE_SLOTS, 4), np.int32)
    ring.seq, ring.done, ring.stream_dev = 0, 0, None
    ring.dev_ptr, ring.host_ptr = 1 << 20, 1 << 21
    synced = []
    real = M.torch.cuda.synchronize
    M.torch.cuda.synchronize = lambda dev=None: synced.append(dev)
    try:
        d0, h0, r0, s0 = ring.next()
        assert (d0, h0, r0, s0) == ((1 << 20) + 16, (1 << 21) + 16, (1 << 20) + 32, 1)
        ring.next(); ring.next()
        # call 1 landed out of range, call 2 never launched, call 3 landed in range
        ring.host_np[1] = [np.float32(-2.5).view(np.int32), np.float32(1.0).view(np.int32), 1, 0]
        ring.host_np[3] = [np.float32(-1.0).view(np.int32), np.float32(1.0).view(np.int32), 3, 0]
        import contextlib, io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ring.flush(block=False)
        assert "min value is  -2.5" in buf.getvalue() and "max value" not in buf.getvalue()
        assert ring.done == 1 and not synced                       # stuck on call 2 without blocking
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ring.flush(block=True)
        assert ring.done == 3 and len(synced) == 1 and buf.getvalue() == ""
    finally:
        M.torch.cuda.synchronize = real
