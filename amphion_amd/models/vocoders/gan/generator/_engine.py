"""Shared host logic of the MI355X generators: parameter holders that keep the reference's
state_dict key names/shapes, and the bridge from an nn.Module to an ``amp_gen`` handle.

The nn.Parameters stay the source of truth (``load_state_dict``, ``.to()``, ``.cuda()``,
``remove_weight_norm()`` all work as in the reference); the folded + MFMA-packed device copy
inside the handle is rebuilt lazily whenever a parameter changes.
"""
from __future__ import annotations

import ctypes
import os
import threading
import warnings
import weakref

import torch
import torch.nn as nn

from amphion_amd import _lib


# forward_graphed's caches: generator -> {(B, T bucket, device): capture() triple}.  A weak-keyed side table, not attributes: graphs and
# locks must not travel with torch.save(model) / copy.deepcopy(model).
_graph_caches = weakref.WeakKeyDictionary()
_graph_lock = threading.Lock()
_GRAPH_CACHE_ON = os.environ.get("AMP_GRAPH_CACHE", "1") != "0"      # AMP_GRAPH_CACHE=0: forward_graphed always runs eagerly (A/B switch)


def _norm_except_dim0(v: torch.Tensor) -> torch.Tensor:
    return v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))


class ConvParams(nn.Module):
    """Parameters of ``weight_norm(Conv1d(...))`` / ``weight_norm(ConvTranspose1d(...))`` or of a
    plain conv, under the reference's key names: ``bias``, ``weight_g``, ``weight_v`` (or
    ``weight`` once weight-norm is removed / never applied).

    Initialisation reproduces the reference's RNG consumption: torch's default Conv init, then
    (``init_normal=True``) the ``init_weights`` draw of gan_utils.py:25-28 -- which, applied after
    ``weight_norm``, only touches the derived ``weight`` attribute and leaves g/v unchanged
    (hifigan.py:55,90,200-201).
    """

    def __init__(self, cin, cout, k, *, transposed=False, stride=1, dilation=1, padding=0, weight_norm=True,
                 bias=True, init_normal=False):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.transposed, self.stride, self.dilation, self.padding = transposed, stride, dilation, padding
        if transposed:
            ref = nn.ConvTranspose1d(cin, cout, k, stride, padding=padding, bias=bias)
        else:
            ref = nn.Conv1d(cin, cout, k, 1, dilation=dilation, padding=padding, bias=bias)
        w = ref.weight.data
        b = ref.bias.data.clone() if bias else None
        if weight_norm:
            # weight_norm() deletes `weight` and registers g, v AFTER the existing bias
            if bias:
                self.bias = nn.Parameter(b)
            else:
                self.register_parameter("bias", None)
            self.weight_g = nn.Parameter(_norm_except_dim0(w).clone())
            self.weight_v = nn.Parameter(w.clone())
            if init_normal:
                torch.empty_like(w).normal_(0.0, 0.01)  # consumed, no effect on g/v (see docstring)
        else:
            if init_normal:
                w = w.normal_(0.0, 0.01)
            self.weight = nn.Parameter(w.clone())
            if bias:
                self.bias = nn.Parameter(b)
            else:
                self.register_parameter("bias", None)

    @property
    def has_weight_norm(self):
        return "weight_g" in self._parameters

    def folded_weight(self) -> torch.Tensor:
        """w = g * v / ||v|| (torch.nn.utils.weight_norm, dim=0)."""
        if not self.has_weight_norm:
            return self.weight
        return self.weight_g * (self.weight_v / _norm_except_dim0(self.weight_v))

    def remove_weight_norm(self):
        if not self.has_weight_norm:
            raise ValueError("weight_norm of 'weight' not found")  # as torch.nn.utils.remove_weight_norm
        w = self.folded_weight().detach()
        del self._parameters["weight_g"]
        del self._parameters["weight_v"]
        self.weight = nn.Parameter(w)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # accept the other form (folded <-> weight-normed) by converting this holder first
        has_w = prefix + "weight" in state_dict
        has_gv = prefix + "weight_g" in state_dict and prefix + "weight_v" in state_dict
        if self.has_weight_norm and has_w and not has_gv:
            self.remove_weight_norm()
        elif not self.has_weight_norm and has_gv and not has_w:
            w = self._parameters.pop("weight")
            self.weight_g = nn.Parameter(_norm_except_dim0(w.data).clone())
            self.weight_v = nn.Parameter(w.data.clone())
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def extra_repr(self):
        kind = "ConvTranspose1d" if self.transposed else "Conv1d"
        return f"{kind}({self.cin}, {self.cout}, k={self.k}, stride={self.stride}, dilation={self.dilation}, " \
               f"padding={self.padding}, weight_norm={self.has_weight_norm})"


def _destroy_handle(ptr):
    try:
        _lib.lib().amp_gen_destroy(ctypes.c_void_p(ptr))
    except Exception:  # interpreter shutdown
        pass


class _InferenceOnly(torch.autograd.Function):
    """Identity on the generator's output whose backward raises: see ``HipGenerator._amp_forward``."""

    @staticmethod
    def forward(ctx, out, witness):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        raise RuntimeError("amphion_amd generators are inference-only (no backward through the HIP kernels): a gradient was asked for "
                           "THROUGH the generator -- a training step, or an input that requires grad.  Call under torch.no_grad() / "
                           ".eval() for inference; build the reference class (module._reference_<Name>, kept by "
                           "amphion_amd.integration) for training")


class HipGenerator(nn.Module):
    """Base of HiFiGAN / HiFiGAN_vits / BigVGAN: owns the ``amp_gen`` handle and the workspace."""

    def __init__(self):
        super().__init__()
        self._amp_handle = None
        self._amp_finalizer = None
        self._amp_sig = None
        self._amp_snap = None
        self._amp_epoch = 0
        self._amp_device = None
        self._amp_ws = None
        self._amp_profiling = 0

    # subclasses provide the architecture descriptor
    def _amp_desc(self) -> _lib.amp_gen_desc:  # pragma: no cover - abstract
        raise NotImplementedError

    @staticmethod
    def _fill_desc(arch, n_in, c0, rates, ksizes, rb_kernels, rb_dils, resblock, activation=_lib.AMP_ACT_LRELU,
                   logscale=False, gin=0):
        d = _lib.amp_gen_desc()
        if len(rates) > _lib.AMP_MAX_STAGES or len(rb_kernels) > _lib.AMP_MAX_KERNELS:
            raise ValueError("too many upsample stages / resblock kernels")
        d.arch, d.n_in, d.upsample_initial_channel, d.n_stages = arch, int(n_in), int(c0), len(rates)
        for i, (u, k) in enumerate(zip(rates, ksizes)):
            d.upsample_rates[i], d.upsample_kernel_sizes[i] = int(u), int(k)
        d.n_kernels = len(rb_kernels)
        for j, (k, dl) in enumerate(zip(rb_kernels, rb_dils)):
            if len(dl) > _lib.AMP_MAX_DILATIONS:
                raise ValueError("too many dilations")
            d.resblock_kernel_sizes[j] = int(k)
            d.n_dilations[j] = len(dl)
            for p, v in enumerate(dl):
                d.resblock_dilation_sizes[j][p] = int(v)
        d.resblock_type = 1 if str(resblock) == "1" else 2
        d.activation, d.snake_logscale, d.gin_channels = int(activation), int(bool(logscale)), int(gin)
        return d

    @property
    def hop_factor(self) -> int:
        """Output samples per input frame = prod(upsample_rates)."""
        d = self._amp_desc()
        h = 1
        for i in range(d.n_stages):
            h *= d.upsample_rates[i]
        return h

    def _amp_weights(self):
        """(reference key, tensor) pairs handed to ``amp_gen_set_weight``; subclasses may filter / transform."""
        return self.state_dict().items()

    def _amp_signature(self):
        """Full identity of the weights: (key, storage address, in-place version, shape) of every state tensor.  A walk
        over the module tree (~0.25 ms for HiFi-GAN V1's 234 tensors): taken when the handle is built and whenever the
        per-forward check below fails."""
        return tuple((k, v.data_ptr(), v._version, tuple(v.shape)) for k, v in self.state_dict(keep_vars=True).items())

    def _amp_snapshot(self):
        """Per-forward change detection without the tree walk (~0.05 ms): the Parameter / buffer OBJECTS registered in
        every sub-module, plus (address, version) of each -- catches load_state_dict and optimizer steps (in-place:
        version), .to() / .cuda() / .float() (new storage), remove_weight_norm and any re-registered Parameter
        (object identity).  A sub-MODULE swapped for another after the first forward is only seen through
        ``invalidate()`` -- the reference never does that to a built generator."""
        mods = list(self.modules())
        tens = [t for m in mods for d in (m._parameters, m._buffers) for t in d.values()]
        return mods, tens, [(t.data_ptr(), t._version) if t is not None else None for t in tens]

    def _amp_unchanged(self):
        mods, tens, marks = self._amp_snap
        i = 0
        for m in mods:
            for d in (m._parameters, m._buffers):
                for t in d.values():
                    if i >= len(tens) or t is not tens[i]:
                        return False
                    if t is not None and marks[i] != (t.data_ptr(), t._version):
                        return False
                    i += 1
        return i == len(tens)

    def invalidate(self):
        """Forget the packed weights: the next forward re-reads every parameter."""
        self._amp_release()

    def _amp_release(self):
        if self._amp_finalizer is not None:
            self._amp_finalizer()  # destroys the handle once
        self._amp_handle = self._amp_finalizer = self._amp_sig = self._amp_device = self._amp_snap = None
        self._amp_epoch = getattr(self, "_amp_epoch", 0) + 1          # captured graphs of the old handle are dead

    def _amp_ensure(self, device):
        if self._amp_handle is not None and device == self._amp_device and self._amp_unchanged():
            return self._amp_handle
        sig = self._amp_signature()
        if self._amp_handle is not None and sig == self._amp_sig and device == self._amp_device:
            self._amp_snap = self._amp_snapshot()      # same weights behind re-registered objects
            return self._amp_handle
        self._amp_release()
        L = _lib.lib()
        desc = self._amp_desc()
        h = ctypes.c_void_p()
        _lib.check(L.amp_gen_create(ctypes.byref(desc), ctypes.byref(h)))
        fin = weakref.finalize(self, _destroy_handle, h.value)
        prev_prec = L.amp_get_precision()
        try:
            staged = []
            for key, t in self._amp_weights():
                c = t.detach().to(torch.float32).contiguous()      # host or device: amp_gen_set_weight takes either
                staged.append((key, c, c.is_cuda and c.data_ptr() != t.data_ptr()))
            # amp_gen_set_weight copies with a blocking hipMemcpy on the NULL stream; torch's side streams do not order with
            # it, so a temporary made by a conversion kernel (fp16 / bf16 or non-contiguous parameters) on a non-default
            # current stream -- capture()'s warm-up always runs on one -- must be complete before it is read
            for dev in {c.device for _, c, converted in staged if converted}:
                torch.cuda.current_stream(dev).synchronize()
            for key, c, _ in staged:
                shape = (ctypes.c_int64 * c.dim())(*c.shape)
                _lib.check(L.amp_gen_set_weight(h, key.encode(), ctypes.c_void_p(c.data_ptr()), shape, c.dim()))
            del staged
            with torch.cuda.device(device):
                if getattr(self, "_amp_force_f32", False):      # forward_exact_range fell back: pack for the fp32 kernels
                    _lib.check(L.amp_set_precision(_lib.AMP_PRECISION_F32))
                _lib.check(L.amp_gen_finalize(h))
                if self._amp_profiling:
                    _lib.check(L.amp_gen_set_profiling(h, int(self._amp_profiling)))
        except Exception:
            fin()
            raise
        finally:
            L.amp_set_precision(prev_prec)
        self._amp_handle, self._amp_finalizer, self._amp_sig, self._amp_device = h, fin, sig, device
        self._amp_snap = self._amp_snapshot()
        return h

    def _amp_forward(self, x, g=None, lengths=None, workspace=None, lens_dev=None):
        # Inference only: the HIP kernels have no backward.  The forward itself runs whenever the reference's would -- a module left in
        # its default training mode and called without torch.no_grad() is an ordinary inference call (round 3 refused it) --
        # but where autograd would have recorded a graph through the reference generator (an input that requires grad; training mode
        # with trainable parameters: what a GAN trainer does through the registry / class patch of amphion_amd.integration) the output
        # carries a grad_fn that RAISES when a backward pass reaches it: loss_g.backward() fails loudly instead of succeeding through
        # the discriminator only with a generator that silently never trains.
        guard = None
        if torch.is_grad_enabled() and isinstance(x, torch.Tensor):
            if x.requires_grad:
                guard = x
            elif self.training:
                guard = next((p for p in self.parameters() if p.requires_grad), None)
        x = _lib.require_device_tensor(x, "generator input")
        if x.dim() != 3:
            raise ValueError(f"expected [B, C, T] input, got {tuple(x.shape)}")
        dev = x.device
        p0 = next(self.parameters())
        if p0.device != dev:
            raise RuntimeError(f"generator parameters are on {p0.device} but the input is on {dev}")
        L = _lib.lib()
        h = self._amp_ensure(dev)
        B, C, T = x.shape
        if C != self._amp_n_in:
            raise ValueError(f"expected {self._amp_n_in} input channels, got {C}")
        cond_ptr = None
        if g is not None:
            g = _lib.require_device_tensor(g, "g")
            if g.dim() != 3 or g.shape[0] != B or g.shape[2] != 1:
                raise ValueError(f"g must be [B, gin_channels, 1], got {tuple(g.shape)}")
            cond_ptr = ctypes.c_void_p(g.data_ptr())
        lens_ptr = None
        if lens_dev is not None:          # int32 [B] on the device, validated by the caller (graph capture: no host read of it here)
            lens_ptr = ctypes.c_void_p(lens_dev.data_ptr())
        elif lengths is not None:
            lengths = torch.as_tensor(lengths)
            if lengths.numel() != B or int(lengths.max()) > T or int(lengths.min()) < 1:
                raise ValueError(f"lengths must hold B={B} values in [1, {T}]")
            lengths = lengths.to(device=dev, dtype=torch.int32).contiguous()
            lens_ptr = ctypes.c_void_p(lengths.data_ptr())
        hop = L.amp_gen_hop(h)
        need = L.amp_gen_workspace_bytes(h, B, T)
        ws = workspace
        if ws is None:
            if self._amp_ws is None or self._amp_ws.numel() < need or self._amp_ws.device != dev:
                self._amp_ws = None
                self._amp_ws = torch.empty(need, dtype=torch.uint8, device=dev)
            ws = self._amp_ws
        elif ws.numel() < need or ws.device != dev:
            raise ValueError("workspace too small for this (B, T)")
        out = torch.empty((B, 1, T * hop), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.amp_gen_forward_ragged(h, ctypes.c_void_p(x.data_ptr()), cond_ptr, lens_ptr, B, T,
                                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                                ws.numel(), _lib.current_stream_ptr(dev)))
        return out if guard is None else _InferenceOnly.apply(out, guard)

    def check_range(self):
        """Synchronise and raise ``AmpError`` (``status == _lib.AMP_ERR_RANGE``) if a forward since the last check
        staged an activation outside the split-f16 operand range (|x| > 4094 or non-finite; the fp32 reference has no
        such limit).  A later forward reports the same without synchronising.  ``forward_exact_range`` is the
        self-healing form."""
        if self._amp_handle is None:
            return
        dev = next(self.parameters()).device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().amp_gen_range_check(self._amp_handle, _lib.current_stream_ptr(dev)))

    def forward_exact_range(self, x, g=None, lengths=None):
        """Forward with the fp32 reference's operand range: runs the f16x3 kernels, checks the range flag (one
        synchronisation) and, if an activation did not fit, repeats the call on the exact-fp32 MFMA kernels (handle
        rebuilt once; it stays in fp32 from then on).  This is what the drop-in entry points (``vocoder_inference``,
        ``synthesis_audios``, ``inference_batches``) call: they copy the audio to the host anyway, so the check's
        synchronisation is free and no batch -- the last or only one included -- returns inf / NaN audio silently."""
        return self._amp_exact_range(lambda: self._amp_forward(x, g, lengths=lengths))

    def _amp_exact_range(self, run):
        """``run()`` (one or more forwards of this generator), then the range check; on ``AMP_ERR_RANGE`` -- from the check
        or from ``run`` itself, which reports a flag an EARLIER unchecked forward left behind -- switch this generator to
        the exact-fp32 kernels and run again."""
        try:
            out = run()
            self.check_range()
            return out
        except _lib.AmpError as e:
            if e.status != _lib.AMP_ERR_RANGE or getattr(self, "_amp_force_f32", False):
                raise
        warnings.warn("amphion_amd: an activation left the split-f16 operand range; this generator now runs the "
                      "exact-fp32 kernels", RuntimeWarning, stacklevel=3)
        self._amp_force_f32 = True
        self._amp_release()
        out = run()
        self.check_range()
        return out

    def receptive_frames(self):
        """One-sided receptive field of the generator in INPUT frames, rounded up with a margin: output sample n of a
        forward depends on input frames within this distance of n / hop only.  conv_pre (k - 1) / 2 frames; per stage the
        transposed conv (one input sample either side), the widest resblock (sum over its dilations of
        (k - 1) / 2 * d [+ (k - 1) / 2 for ResBlock1's second conv] samples, + 6 per anti-aliased activation for BigVGAN)
        at that stage's rate; the final activation (BigVGAN) and conv_post at the output rate.  HiFi-GAN V1: 13.3 -> 17,
        BigVGAN-base 18.9 -> 22.  Lets ``synthesis_audios`` stop a padded item's work a receptive field beyond its own
        length without changing one bit of the samples it keeps."""
        d = self._amp_desc()
        big = d.arch == _lib.AMP_ARCH_BIGVGAN
        rf = 3.0                                   # conv_pre, k = 7
        rate = 1
        for i in range(d.n_stages):
            rf += 1.0 / rate                       # ConvTranspose1d with k = 2u: two taps, one input sample either side
            rate *= d.upsample_rates[i]
            widest = 0
            for j in range(d.n_kernels):
                k = d.resblock_kernel_sizes[j]
                w = 0
                for p in range(d.n_dilations[j]):
                    w += (k - 1) // 2 * d.resblock_dilation_sizes[j][p]
                    if d.resblock_type == 1:
                        w += (k - 1) // 2
                    if big:
                        w += 6 * (2 if d.resblock_type == 1 else 1)
                widest = max(widest, w)
            rf += widest / rate
        rf += (3 + (6 if big else 0)) / rate       # conv_post k = 7 (+ activation_post)
        return int(rf * 1.15) + 2

    def forward_ragged(self, x, lengths, g=None):
        """A zero-padded batch of utterances of different lengths in ONE forward: item b holds
        ``lengths[b]`` valid frames; ``out[b, 0, : lengths[b] * hop]`` is bit-identical to running that
        utterance alone (every layer pads at the utterance's own end), the tail beyond it is unspecified.
        This is what lets ``synthesis_audios`` replace the reference's B=1 loop
        (gan_vocoder_inference.py:74-96) by true batches without changing results."""
        return self._amp_forward(x, g, lengths=lengths)

    # ---- hipGraph capture (launch-bound small batches) ----
    def capture(self, B, T, g_shape=None, ragged=False, workspace=None):
        """Capture one forward at a fixed shape into a hipGraph and return ``(replay, static_in, static_out)``.

        A forward is 51 (HiFi-GAN V1) dependent kernel launches; for a single utterance each of them runs a few
        tens of microseconds, so host launch cost and inter-launch gaps are a visible share of the latency.
        ``replay()`` re-runs the captured launches with one ``hipGraphLaunch``: copy the new mel into
        ``static_in`` (same shape), call ``replay()``, read ``static_out``.  The kernels launch on the caller's
        stream and neither allocate nor synchronise, so plain stream capture works; one eager warm-up forward
        runs first (it builds the handle and sets the >64 KiB dynamic-LDS attributes outside the capture).

        The graph records raw device pointers: it owns its OWN workspace (kept alive by ``replay``; eager forwards of
        other shapes may re-allocate the module's), and it is tied to the packed weights of the current handle --
        ``replay()`` raises once the parameters changed (``load_state_dict``, ``.to()``, an in-place update ...) instead
        of reading freed memory; capture again after such a change.
        """
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("capture() needs the generator on a ROCm device")
        static_in = torch.zeros((B, self._amp_n_in, T), dtype=torch.float32, device=dev)
        static_g = torch.zeros(g_shape, dtype=torch.float32, device=dev) if g_shape is not None else None
        # ragged=True: the graph runs forward_ragged with the valid lengths read from ``replay.static_lens`` (int32 [B] on the
        # device) at replay time -- one graph then serves every batch of <= T frames per item (forward_graphed's buckets)
        static_lens = torch.full((B,), T, dtype=torch.int32, device=dev) if ragged else None
        was_profiling = self._amp_profiling
        self.set_profiling(False)                       # event records are not part of the product graph
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            self._amp_forward(static_in, static_g, lens_dev=static_lens)      # warm-up: handle, function attributes, side streams
            need = _lib.lib().amp_gen_workspace_bytes(self._amp_handle, B, T)
            # the graph's own scratch -- or the caller's (forward_graphed: one scratch shared by all of a generator's graphs, which
            # replay one at a time on one stream)
            ws = workspace if workspace is not None else torch.empty(need, dtype=torch.uint8, device=dev)
            if ws.numel() < need or ws.device != dev:
                raise ValueError("capture(): workspace too small for this (B, T)")
            self._amp_forward(static_in, static_g, workspace=ws, lens_dev=static_lens)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            static_out = self._amp_forward(static_in, static_g, workspace=ws, lens_dev=static_lens)
        self.set_profiling(was_profiling)
        epoch, handle = self._amp_epoch, self._amp_handle.value

        def replay(check=True):
            # (check=False: the caller has just run _amp_ensure, whose walk over every parameter is the expensive part of this test)
            if self._amp_epoch != epoch or self._amp_handle is None or self._amp_handle.value != handle or (check and not self._amp_unchanged()):
                raise RuntimeError("captured graph is stale: the generator's parameters / device / precision changed "
                                   "after capture(); capture again")
            graph.replay()
            return static_out

        replay.graph = graph
        replay.workspace = ws
        replay.static_g = static_g
        replay.static_lens = static_lens
        return replay, static_in, static_out

    # ---- cached graphs for repeated small shapes (single utterances) ----
    GRAPH_BUCKET_FRAMES = 32        # shapes are rounded up to a multiple of this many frames
    GRAPH_MAX_FRAMES = 1024         # larger batches run eagerly: the gain shrinks with the batch
    GRAPH_MAX_ENTRIES = 48          # graphs share ONE scratch (sized for GRAPH_MAX_FRAMES); each owns only its input / output tensors (~1.3 MB)

    def forward_graphed(self, x, lengths=None):
        """``forward`` / ``forward_ragged`` of a small batch through a cached hipGraph; returns a tensor of its own (see
        ``_forward_graphed_view`` for what happens underneath)."""
        return self._forward_graphed_view(x, lengths, clone=True)

    def _forward_graphed_view(self, x, lengths=None, clone=False):
        """``forward`` / ``forward_ragged`` of a small batch through a cached hipGraph: the (B, T) shape is rounded up to a bucket of
        ``GRAPH_BUCKET_FRAMES`` frames, the bucket's graph -- captured the SECOND time the bucket is seen -- runs the ragged forward
        with the valid lengths in a device buffer, and ``out[b, 0, : lengths[b] * hop]`` (``lengths`` defaults to T for every item) is
        bit-identical to the eager forward of that utterance alone (the ragged contract: every layer pads at the utterance's own end;
        tests/test_gpu_inference_api.py).  What it buys: one ``hipGraphLaunch`` instead of ~85 launch / event calls, and the
        concurrent-resblock launch order without host gaps -- a 3-s utterance 0.95 -> 0.83 ms.  Falls back to the eager call for
        batches beyond ``GRAPH_MAX_FRAMES`` frames, profiling runs and inputs that require grad (a generator called WITH a conditioning
        input never comes here: this entry point takes none, callers with ``g`` use ``forward``).  The graphs
        die with the packed weights (``load_state_dict`` / ``.to()`` / precision switch: the cache is dropped and rebuilt).
        PRIVATE because of ``clone=False``: the result is then a VIEW of the graph's own output buffer, overwritten by the next call for the
        same bucket -- only for callers that copy it away at once (``vocoder_inference`` / ``synthesis_audios``: straight to the host).
        The public ``forward_graphed`` always clones."""
        x = _lib.require_device_tensor(x, "generator input")
        B, C, T = x.shape
        eager = lambda: self._amp_forward(x, lengths=lengths)
        if (not _GRAPH_CACHE_ON or B * T > self.GRAPH_MAX_FRAMES or self._amp_profiling or torch.is_grad_enabled() and (x.requires_grad or self.training)
                or torch.cuda.is_current_stream_capturing()):
            return eager()
        if lengths is not None:
            lt = torch.as_tensor(lengths).reshape(-1).to(torch.int32).cpu()
            if lt.numel() != B or int(lt.max()) > T or int(lt.min()) < 1:
                raise ValueError(f"lengths must hold B={B} values in [1, {T}]")
        else:
            lt = torch.full((B,), T, dtype=torch.int32)
        Tb = -(-T // self.GRAPH_BUCKET_FRAMES) * self.GRAPH_BUCKET_FRAMES
        with _graph_lock:
            self._amp_ensure(x.device)              # a stale handle bumps the epoch here, before the cache is consulted
            cache = _graph_caches.setdefault(self, {})
            if cache.get("epoch") != self._amp_epoch:
                cache.clear()
                cache["epoch"] = self._amp_epoch
            key = (B, Tb, str(x.device))
            ent = cache.get(key)
            if ent is None:                         # first sight of this bucket: run it eagerly, capture if it comes again
                cache[key] = "seen"
                while sum(1 for k in cache if isinstance(k, tuple)) > self.GRAPH_MAX_ENTRIES:
                    cache.pop(next(k for k in cache if isinstance(k, tuple)))
                return eager()
            if ent == "seen":
                ws = cache.get("ws")
                need = _lib.lib().amp_gen_workspace_bytes(self._amp_handle, B, Tb)
                if ws is None or ws.device != x.device:
                    cap = _lib.lib().amp_gen_workspace_bytes(self._amp_handle, 1, self.GRAPH_MAX_FRAMES)
                    ws = cache["ws"] = torch.empty(max(need, cap + cap // 8), dtype=torch.uint8, device=x.device)
                if ws.numel() < need:                   # (a shape whose scratch outgrows the shared one: leave it eager)
                    return eager()
                # capture()'s warm-up forwards and the capture itself run on the SHARED scratch: order them behind the previous replay
                # if that went to another stream (capture() itself only orders them behind the current stream)
                last = cache.get("last")
                if last is not None and last[0] != torch.cuda.current_stream(x.device):
                    torch.cuda.current_stream(x.device).wait_event(last[1])
                ent = cache[key] = self.capture(B, Tb, ragged=True, workspace=ws)
            replay, static_in, static_out = ent
            # the graphs share one scratch: a replay on ANOTHER stream than the previous one first waits for that one to finish
            st = torch.cuda.current_stream(x.device)
            last = cache.get("last")
            if last is not None and last[0] != st:
                st.wait_event(last[1])
            static_in[:, :, :T].copy_(x)            # frames beyond an item's length are never read (the kernels select on the lengths)
            lens_key = tuple(lt.tolist())
            if getattr(replay, "lens_key", None) != lens_key:      # the lengths the buffer already holds need no second upload
                if B == 1:
                    replay.static_lens.fill_(int(lt[0]))           # a one-element fill kernel on the stream: no staging at all
                else:
                    # pinned staging, four slots used in turn: a slot is rewritten only after the copy that read it has completed
                    ring = getattr(replay, "lens_ring", None)
                    if ring is None:
                        ring = replay.lens_ring = [[torch.empty(B, dtype=torch.int32).pin_memory(), None] for _ in range(4)]
                        replay.lens_slot = 0
                    slot = ring[replay.lens_slot]
                    replay.lens_slot = (replay.lens_slot + 1) % len(ring)
                    if slot[1] is not None:
                        slot[1].synchronize()
                    slot[0].copy_(lt)
                    replay.static_lens.copy_(slot[0], non_blocking=True)   # stream-ordered before the replay
                    slot[1] = torch.cuda.Event()
                    slot[1].record(st)
                replay.lens_key = lens_key
            replay(check=False)
            if last is None or last[0] != st:
                ev = torch.cuda.Event()
                cache["last"] = (st, ev)
            else:
                ev = last[1]
            hop = static_out.shape[-1] // Tb
            out = static_out[:, :, : T * hop]
            out = out.clone() if clone else out
            ev.record(st)
            return out

    # ---- profiling hooks used by bench.py ----
    def set_profiling(self, slots=1):
        """Record HIP events around the kernels of every forward into a ring of ``slots`` event sets
        (True = 1, False / 0 = off); see ``amp_gen_set_profiling``."""
        self._amp_profiling = int(slots)
        if self._amp_handle is not None:
            _lib.check(_lib.lib().amp_gen_set_profiling(self._amp_handle, int(slots)))

    def last_timing_ms(self, which=0, back=0):
        """HIP-event time of a recorded forward on its launch stream (``back`` = 0: the latest).  which: 0 whole
        forward, 1 MRF conv stack, 2 + i stage i, 100 + 16*i + j resblock j of stage i."""
        ms = ctypes.c_float()
        _lib.check(_lib.lib().amp_gen_timing_ms(self._amp_handle, back, which, ctypes.byref(ms)))
        return ms.value

    def kernel_names(self, which, back=0):
        """The kernels resblock j of stage i (``which`` = 100 + 16*i + j) launched in a recorded forward, as the library
        reports them (``amp_gen_kernel_name``: rocprofv3 spelling incl. template arguments), a list of names."""
        buf = ctypes.create_string_buffer(1024)
        _lib.check(_lib.lib().amp_gen_kernel_name(self._amp_handle, back, which, buf, len(buf)))
        return [n for n in buf.value.decode().split(" | ") if n]
