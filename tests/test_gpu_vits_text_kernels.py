"""Round-4 launch savers of the VITS text side: every fused / ragged entry point against the launches it
replaces -- bit for bit on the valid columns -- and with the memory beyond each utterance's end POISONED (NaN), which is what the
`* x_mask` launches they remove used to hide.  Op-level parity with oracle/vits_infer_oracle.py stays in test_gpu_vits_infer.py."""
import pytest
import torch

from oracle import vits_infer_oracle as vio

pytestmark = pytest.mark.gpu


def _mask(lens, T):
    return (torch.arange(T).view(1, 1, T) < lens.view(-1, 1, 1))


def _poison(x, lens):
    """NaN in the columns t >= lens[b]"""
    x = x.clone()
    x.masked_fill_(~_mask(lens, x.shape[-1]).expand_as(x), float("nan"))
    return x


@pytest.mark.parametrize("B,H,dk,T,window,lens", [
    (3, 2, 96, 100, 4, [100, 61, 1]),          # config/vits.json heads; one tile of keys
    (2, 2, 16, 3, 4, [3, 2]),                  # shorter than the window
    (2, 1, 32, 130, 4, [130, 129]),            # two key tiles, a ragged tail of 2 keys
    (1, 4, 64, 300, 7, [257]),                 # three key tiles, the widest window of the tiled form
    (2, 2, 8, 17, 0, [17, 5]),                 # no relative window at all
    (1, 2, 128, 33, 2, None),                  # dk = 128 (every thread of pass 3 live), no lengths
])
def test_rel_attention_tiled_bitwise(B, H, dk, T, window, lens):
    from amphion_amd import _lib
    from amphion_amd.modules import hip_ops

    g = torch.Generator().manual_seed(B * 1000 + T)
    C = H * dk
    q, k, v = (torch.randn(B, C, T, generator=g).cuda() for _ in range(3))
    ek, ev = (torch.randn(2 * window + 1, dk, generator=g).cuda() * 0.3 for _ in range(2))
    ld = torch.tensor(lens, dtype=torch.int32).cuda() if lens is not None else None
    L = _lib.lib()
    try:
        _lib.check(L.amp_set_rel_attention_tiled(0))
        ref = hip_ops.rel_attention(q, k, v, ek, ev, ld, H, window)
    finally:
        _lib.check(L.amp_set_rel_attention_tiled(1))
    got = hip_ops.rel_attention(q, k, v, ek, ev, ld, H, window)
    assert torch.equal(got, ref)
    # the merged-projection form reads the three slices of one tensor in place
    qkv = torch.cat([q, k, v], dim=1).contiguous()
    assert torch.equal(hip_ops.rel_attention_qkv(qkv, ek, ev, ld, H, window), ref)
    # and against the oracle's attention (identity projections), valid queries only
    if window == 4 and lens is not None:
        sd = {}
        for n in ("q", "k", "v", "o"):
            sd[f"a.conv_{n}.weight"], sd[f"a.conv_{n}.bias"] = torch.eye(C).unsqueeze(-1), torch.zeros(C)
        sd["a.emb_rel_k"], sd["a.emb_rel_v"] = ek.cpu().unsqueeze(0), ev.cpu().unsqueeze(0)
        x = q.cpu()
        ms = _mask(torch.tensor(lens), T).float()
        want = vio.relative_self_attention(sd, "a", x, ms, H, window)
        same = hip_ops.rel_attention(q, q, q, ek, ev, ld, H, window).cpu()
        assert ((same - want) * ms).abs().max().item() <= 5e-5


def test_rel_attention_falls_back_outside_the_tile_kernel():
    from amphion_amd.modules import hip_ops

    g = torch.Generator().manual_seed(3)
    B, H, dk, T, window = 1, 1, 6, 20, 9        # dk no multiple of 4, 19 relative positions: the one-query kernel
    q, k, v = (torch.randn(B, H * dk, T, generator=g).cuda() for _ in range(3))
    ek, ev = (torch.randn(2 * window + 1, dk, generator=g).cuda() for _ in range(2))
    out = hip_ops.rel_attention(q, k, v, ek, ev, None, H, window)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("C,T,lens", [(192, 100, [100, 37, 1]), (24, 33, [33, 32, 31]), (300, 70, [64, 70, 5]), (7, 1, [1, 1, 0])])
def test_layer_norm_ragged(C, T, lens):
    from amphion_amd.modules import hip_ops

    g = torch.Generator().manual_seed(C + T)
    B = len(lens)
    lt = torch.tensor(lens)
    ld = lt.to(torch.int32).cuda()
    x, r, p = (torch.randn(B, C, T, generator=g) for _ in range(3))
    gm, bt = (1 + 0.2 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    valid = _mask(lt, T).expand(B, C, T).cuda()
    for gelu in (False, True):
        ref = hip_ops.layer_norm_c(x.cuda(), gm, bt, res=r.cuda(), post=p.cuda(), gelu=gelu)
        got = hip_ops.layer_norm_c(_poison(x, lt).cuda(), gm, bt, res=_poison(r, lt).cuda(), post=_poison(p, lt).cuda(), gelu=gelu, lens=ld)
        assert torch.equal(got[valid], ref[valid])
        assert (got[~valid] == 0).all()


@pytest.mark.parametrize("C,T,dil,lens", [(192, 100, 1, [100, 37, 1]), (192, 100, 9, [100, 60, 8]), (24, 70, 3, [33, 70, 64]), (300, 40, 3, [40, 17, 2])])
def test_dwconv_layer_norm_bitwise(C, T, dil, lens):
    from amphion_amd.modules import hip_ops

    g = torch.Generator().manual_seed(C + T + dil)
    B = len(lens)
    lt = torch.tensor(lens)
    ld = lt.to(torch.int32).cuda()
    x = torch.randn(B, C, T, generator=g)
    w, b = torch.randn(C, 1, 3, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    gm, bt = (1 + 0.2 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    valid = _mask(lt, T).expand(B, C, T).cuda()
    ref = hip_ops.layer_norm_c(hip_ops.dwconv(x.cuda(), w, b, ld, dil), gm, bt, gelu=True)
    got = hip_ops.dwconv_layer_norm_c(_poison(x, lt).cuda(), w, b, dil, gm, bt, lens=ld, gelu=True)
    assert torch.equal(got[valid], ref[valid])
    assert (got[~valid] == 0).all()
    # no lengths, no bias
    ref = hip_ops.layer_norm_c(hip_ops.dwconv(x.cuda(), w, None, None, dil), gm, bt)
    assert torch.equal(hip_ops.dwconv_layer_norm_c(x.cuda(), w, None, dil, gm, bt), ref)


@pytest.mark.parametrize("cin,cout,k,T,lens", [(192, 768, 3, 100, [100, 37, 1]), (768, 192, 3, 100, [100, 64, 33]), (192, 576, 1, 150, [150, 1, 97]),
                                                (96, 192, 1, 400, [400, 130, 31]), (192, 384, 5, 256, [256, 200, 9])])
def test_conv_forward_ragged(cin, cout, k, T, lens):
    from amphion_amd.modules.hip_ops import HipConv1d

    torch.manual_seed(cin + k)
    B = len(lens)
    lt = torch.tensor(lens)
    ld = lt.to(torch.int32).cuda()
    conv = HipConv1d(cin, cout, k, padding=(k - 1) // 2, weight_norm=False).cuda()
    x = torch.randn(B, cin, T)
    m = _mask(lt, T)
    ref = conv((x * m).cuda(), slope_out=0.0)
    got = conv(_poison(x, lt).cuda(), slope_out=0.0, lens=ld)
    valid = m.expand(B, cout, T).cuda()
    assert torch.equal(got[valid], ref[valid])


@pytest.mark.parametrize("H", [128, 64])          # the fused WN kernels / the unfused ops (H <= 64 is outside the fused ones)
def test_coupling_and_wn_without_mask_launches(H):
    """ResidualCouplingLayer (reverse and forward) on a ragged batch: the round-4 form (kernels take the lengths) against the
    reference composition written with dense masks, on inputs whose padding is garbage."""
    from amphion_amd.modules import hip_ops
    from amphion_amd.modules.flow.modules import ResidualCouplingLayer

    torch.manual_seed(11)
    B, C, T = 3, 32, 90
    lt = torch.tensor([90, 33, 64])
    ld = lt.to(torch.int32).cuda()
    layer = ResidualCouplingLayer(C, H, 5, 1, 4, mean_only=True).cuda().eval()
    with torch.no_grad():
        layer.post.weight.normal_(0, 0.05)
        layer.post.bias.normal_(0, 0.05)
    x = torch.randn(B, C, T)
    m = _mask(lt, T)
    valid = m.expand(B, C, T).cuda()
    xm = (x * m).cuda()
    for reverse in (True, False):
        with torch.no_grad():
            # the composition of modules/flow/modules.py:378-397 with explicit masks
            h = hip_ops.sequence_mask_(layer.pre(xm[:, :C // 2].contiguous()), ld)
            h = layer.enc(h, ld)
            st = hip_ops.sequence_mask_(layer.post(h), ld)
            x1 = xm[:, C // 2:]
            x1 = (x1 - st) * m.cuda() if reverse else st + x1 * m.cuda()
            want = torch.cat([xm[:, :C // 2], x1], 1)
            got = layer(xm, ld, reverse=reverse)
            got = got if reverse else got[0]
        assert torch.equal(got[valid], want[valid])
        assert (got[:, C // 2:][~valid[:, C // 2:]] == 0).all()


def test_expand_path_strided_and_zero_durations():
    from amphion_amd.modules import hip_ops

    g = torch.Generator().manual_seed(9)
    B, D, Tx = 3, 40, 75
    lt = torch.tensor([75, 40, 1])
    ld = lt.to(torch.int32).cuda()
    mask = _mask(lt, Tx).float()
    logw = torch.randn(B, 1, Tx, generator=g) * 1.5 - 0.5
    logw[:, :, ::7] = -30.0                      # exp -> 0: ceil(0) = 0 frames: tokens that own nothing
    w_ceil, cum, ylen = hip_ops.durations(logw.cuda(), ld, 1.0)
    rw = torch.ceil(torch.exp(logw) * mask)
    assert torch.equal(w_ceil.cpu(), rw)
    ry = torch.clamp_min(rw.sum(dim=(1, 2)), 1).long()
    ty = int(ry.max())
    ymask = (torch.arange(ty).view(1, 1, ty) < ry.view(B, 1, 1)).float()
    path = vio.generate_path(rw, mask.unsqueeze(2) * ymask.unsqueeze(-1))
    stats = torch.randn(B, 2 * D, Tx, generator=g).cuda()
    for half in (0, 1):
        src = stats[:, half * D:(half + 1) * D]                 # a view: read through its batch stride
        out, attn = hip_ops.expand_path(src, cum, ld, ylen, ty, want_attn=(half == 0))
        want = torch.matmul(path.squeeze(1), src.cpu().transpose(1, 2)).transpose(1, 2)
        assert torch.equal(out.cpu(), want)
        if attn is not None:
            assert torch.equal(attn.cpu(), path)


def test_encoder_ragged_batch_equals_single_items():
    """The text encoder on a ragged batch gives, item by item, what it gives on each item alone (nothing beyond an utterance's end
    leaks into it) and zeros beyond."""
    from amphion_amd.models.tts.vits.vits import TextEncoder

    torch.manual_seed(5)
    enc = TextEncoder(50, 16, 64, 128, 2, 3, 3, 0.1).cuda().eval()
    lens = [41, 17, 1, 33]
    T = max(lens)
    tok = torch.randint(0, 50, (len(lens), T))
    with torch.no_grad():
        h, m, logs, _ = enc(tok.cuda(), torch.tensor(lens))
        for b, n in enumerate(lens):
            h1, m1, logs1, _ = enc(tok[b:b + 1, :n].cuda(), torch.tensor([n]))
            for got, want in ((h, h1), (m, m1), (logs, logs1)):
                assert (got[b, :, :n] - want[0]).abs().max().item() <= 2e-5, (b, n)
                assert (got[b, :, n:] == 0).all()


@pytest.mark.parametrize("inverse", [True, False])
@pytest.mark.parametrize("flip", [False, True])
def test_spline_flow_with_fused_projection(inverse, flip):
    """amp_spline_flow_proj == the 1 x 1 projection (here in fp64 on the host) followed by the spline step, and == the oracle's
    transform; the conditioning tensor is poisoned beyond the lengths."""
    from amphion_amd.modules import hip_ops

    g = torch.Generator().manual_seed(17)
    B, C, T, K = 3, 192, 70, 10
    lt = torch.tensor([70, 33, 1])
    ld = lt.to(torch.int32).cuda()
    mask = _mask(lt, T).float()
    z = torch.randn(B, 2, T, generator=g) * 3
    hc = torch.randn(B, C, T, generator=g)
    w = torch.randn(3 * K - 1, C, generator=g) / C**0.5 * 3
    bias = torch.randn(3 * K - 1, generator=g)
    h = (torch.einsum("rc,bct->brt", w.double(), hc.double()) + bias.double().view(1, -1, 1)).float().contiguous()
    zin = torch.flip(z, [1]) if flip else z
    want = hip_ops.spline_flow(zin.cuda(), h.cuda(), ld, K, 64, 5.0, inverse, flip_in=flip, flip_out=flip).cpu()
    got = hip_ops.spline_flow_proj(zin.cuda(), _poison(hc, lt).cuda(), w.cuda(), bias.cuda(), ld, K, 64, 5.0, inverse, flip_in=flip,
                                   flip_out=flip).cpu()
    assert torch.isfinite(got).all()
    # (the projection is summed in another order than the host's: ~1e-6 on h, which the steepest bins of the inverse magnify)
    assert (got - want).abs().max().item() <= 5e-5, (got - want).abs().max().item()
    hm = (h * mask).reshape(B, 1, 3 * K - 1, T).permute(0, 1, 3, 2)
    y1 = vio.rq_spline(z[:, 1:], hm[..., :K] / 8.0, hm[..., K:2 * K] / 8.0, hm[..., 2 * K:], inverse, 5.0)
    ref = torch.cat([z[:, :1], y1], 1) * mask
    ref = torch.flip(ref, [1]) if flip else ref
    assert (got - ref).abs().max().item() <= 2e-4, (got - ref).abs().max().item()


@pytest.mark.parametrize("C,T,dil,lens", [(192, 100, 3, [100, 37, 1]), (192, 100, 9, [100, 60, 8]), (24, 70, 3, [33, 70, 64]), (7, 5, 1, [5, 2, 1])])
def test_dds_seam_bitwise(C, T, dil, lens):
    """amp_dds_seam == amp_layer_norm_c (gelu, post) followed by amp_dwconv_layer_norm_c, bit for bit."""
    from amphion_amd.modules import hip_ops
    from amphion_amd.modules.base import LayerNorm
    from amphion_amd.modules.flow.modules import DepthwiseConv1d

    torch.manual_seed(C + T + dil)
    B = len(lens)
    ld = torch.tensor(lens, dtype=torch.int32).cuda()
    n2, n1 = LayerNorm(C).cuda(), LayerNorm(C).cuda()
    sep = DepthwiseConv1d(C, 3, dil).cuda()
    with torch.no_grad():
        for n in (n2, n1):
            n.gamma.normal_(1.0, 0.2)
            n.beta.normal_(0.0, 0.1)
    y, x = torch.randn(B, C, T).cuda(), torch.randn(B, C, T).cuda()
    with torch.no_grad():
        x_ref = n2(y, gelu=True, post=x)
        z_ref = hip_ops.dwconv_layer_norm_c(x_ref, sep.weight.detach().contiguous(), sep.bias.detach().contiguous(), dil, n1.gamma.detach(),
                                            n1.beta.detach(), lens=ld, eps=n1.eps, gelu=True)
        x_new, z = hip_ops.dds_seam(y, x, n2, sep, n1, ld)
    assert torch.equal(x_new, x_ref)
    assert torch.equal(z, z_ref)
    with torch.no_grad():                                  # no lengths
        z_ref = hip_ops.dwconv_layer_norm_c(x_ref, sep.weight.detach().contiguous(), sep.bias.detach().contiguous(), dil, n1.gamma.detach(),
                                            n1.beta.detach(), eps=n1.eps, gelu=True)
        assert torch.equal(hip_ops.dds_seam(y, x, n2, sep, n1, None)[1], z_ref)


def test_ddsconv_module_equals_unfused_layers():
    """DDSConv.forward (seam kernel between the layers) == its layers run one launch at a time."""
    from amphion_amd.modules import hip_ops
    from amphion_amd.modules.flow.modules import DDSConv

    torch.manual_seed(3)
    m = DDSConv(192, 3, 3).cuda().eval()
    with torch.no_grad():
        for n in list(m.norms_1) + list(m.norms_2):
            n.gamma.normal_(1.0, 0.2)
            n.beta.normal_(0.0, 0.1)
    x = torch.randn(3, 192, 90).cuda()
    ld = torch.tensor([90, 41, 7], dtype=torch.int32).cuda()
    with torch.no_grad():
        got = m(x, ld)
        r = x
        for i in range(3):
            yy = hip_ops.dwconv(r, m.convs_sep[i].weight.detach().contiguous(), m.convs_sep[i].bias.detach().contiguous(), ld, m.convs_sep[i].dilation)
            yy = m.norms_1[i](yy, gelu=True)
            yy = m.convs_1x1[i](yy)
            r = m.norms_2[i](yy, gelu=True, post=r)
        want = hip_ops.sequence_mask_(r, ld)
    assert torch.equal(got, want)
