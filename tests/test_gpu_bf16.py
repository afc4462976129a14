"""GPU parity tests of the mixed-precision path (BASELINE config 4, "mixed bf16"; ``VV_PRECISION=bf16`` /
``[mi355x] precision = bf16``): the convolutions round their operands to bf16 and accumulate in fp32 on
v_mfma_f32_32x32x16_bf16, everything else stays fp32.

The reference has no bf16 mode, so the oracle for these tests is the restatement in ``oracle/unet_oracle.py`` with
``MIXED = MIXED_BF16`` (operand rounding per operation, fp32 accumulation), checked operation by operation and through a
train step; the bar against the fp32 reference arithmetic itself is the one SURVEY.md App. B.14 sets for this config:
AUROC, not per-cube 1e-3.  Tolerances are written at each assert."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _r(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _pack(lib, L, w, G, mode, K, N, st):
    ent = (L.PackEntry * 1)(L.PackEntry(0, 0, mode | L.PACK_BF16, K, K, N))
    tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
    out = torch.zeros(G, 9 * K * N, device='cuda')
    L.check(lib.vv_pack_weights(tab.data_ptr(), 1, G, w.data_ptr(), w[0].numel(), out.data_ptr(), out.stride(0), 9 * K * N, st), 'pack')
    return out


@pytest.mark.parametrize('H,Cin,Cout,B', [(32, 16, 32, 3), (32, 32, 32, 2), (32, 64, 32, 2), (16, 64, 64, 5), (8, 256, 128, 9),
                                           (4, 128, 256, 33), (16, 32, 64, 1)])
def test_bf16_conv3x3_matches_rounded_operand_reference(H, Cin, Cout, B):
    """vv_conv_mfma with VV_CONV_BF16: forward (BatchNorm+ReLU on load, bias, BatchNorm partial sums) and data gradient
    against fp64 convolutions of the bf16-rounded operands.  What is left is fp32 accumulation order (~1e-6) plus the rare
    activation whose fp32 value sits on a bf16 rounding boundary: 2e-4 of the tensor maximum.  The same comparison against
    UNROUNDED operands must fail that bar -- the flag really changes the arithmetic."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 1000 + Cin)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    bias = torch.randn(G, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for dgrad in (False, True):
        K, N = (Cout, Cin) if dgrad else (Cin, Cout)
        if N % 32:
            continue
        src = torch.randn(G, B * H * H, K, generator=g).cuda() if dgrad else x
        pk = _pack(lib, L, w, G, 1 if dgrad else 0, K, N, st)
        nt = lib.vv_conv_ntiles2(B, H, H, L.CONV3, L.CONV_BF16)
        y = torch.full((G, B * H * H, N), 3.0, device='cuda')
        s_ = torch.zeros(G, nt, 2, N, device='cuda')
        cp = L.ConvParams(L.CONV3, L.IN_PLAIN if dgrad else L.IN_ACT, G, B, H, H, K, K, N, L.view(src, K, 0, src.stride(0)),
                          None if dgrad else a.data_ptr(), None if dgrad else b.data_ptr(), K, L.NULL_VIEW, 0, L.CONV_BF16, None,
                          pk.data_ptr(), pk.stride(0), None if dgrad else bias.data_ptr(), N, L.view(y, N, 0, y.stride(0)),
                          None if dgrad else s_.data_ptr())
        L.check(lib.vv_conv_mfma(C.byref(cp), st), 'conv')
        for gi in range(G):
            xin = src[gi].view(B, H, H, K).permute(0, 3, 1, 2)
            if not dgrad:
                xin = torch.relu(torch.addcmul(b[gi].view(1, -1, 1, 1), xin, a[gi].view(1, -1, 1, 1)))
            got = y[gi].view(B, H, H, N).permute(0, 3, 1, 2).double()
            refs = []
            for rnd in (_r, lambda t: t):
                xi, wi = rnd(xin).double(), rnd(w[gi]).double()
                if dgrad:
                    refs.append(torch.nn.grad.conv2d_input((B, N, H, H), wi, xi, padding=1))
                else:
                    refs.append(F.conv2d(xi, wi, bias[gi].double(), padding=1))
            scale = refs[0].abs().max().item()
            err = (got - refs[0]).abs().max().item()
            assert err <= 2e-4 * scale, (dgrad, gi, err, scale)
            assert (got - refs[1]).abs().max().item() > 2e-4 * scale          # not the fp32 arithmetic
            if not dgrad:
                tot = s_[gi].sum(0).double()
                torch.testing.assert_close(tot[0], refs[0].sum((0, 2, 3)), rtol=1e-3, atol=2e-3 * scale * B)
                torch.testing.assert_close(tot[1], (refs[0] ** 2).sum((0, 2, 3)), rtol=2e-3, atol=1e-3)


@pytest.mark.parametrize('H,Cin,Cout,B', [(16, 64, 32, 3), (8, 128, 64, 5), (4, 256, 128, 17)])
def test_bf16_transposed_conv_forward(H, Cin, Cout, B):
    """ConvTranspose2d(k3, s2, p1, op1) forward (model/unet.py:54) with bf16 operands, all four output phases."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 77 + Cin)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    wt = (torch.randn(G, Cin, Cout, 3, 3, generator=g) * 0.1).cuda()          # nn.ConvTranspose2d layout per group
    bias = torch.randn(G, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    st = torch.cuda.current_stream().cuda_stream
    pk = _pack(lib, L, wt, G, 2, Cin, Cout, st)
    y = torch.full((G, B * 4 * H * H, Cout), 3.0, device='cuda')
    cp = L.ConvParams(L.CONVT_FWD, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, L.view(x, Cin, 0, x.stride(0)), a.data_ptr(), b.data_ptr(),
                      Cin, L.NULL_VIEW, 0, L.CONV_BF16, None, pk.data_ptr(), pk.stride(0), bias.data_ptr(), Cout,
                      L.view(y, Cout, 0, y.stride(0)), None)
    L.check(lib.vv_conv_mfma(C.byref(cp), st), 'convT')
    for gi in range(G):
        xin = torch.relu(torch.addcmul(b[gi].view(1, -1, 1, 1), x[gi].view(B, H, H, Cin).permute(0, 3, 1, 2), a[gi].view(1, -1, 1, 1)))
        ref = F.conv_transpose2d(_r(xin).double(), _r(wt[gi]).double(), bias[gi].double(), stride=2, padding=1, output_padding=1)
        got = y[gi].view(B, 2 * H, 2 * H, Cout).permute(0, 3, 1, 2).double()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-4 * scale


@pytest.mark.parametrize('H,Cin,Cout,B', [(16, 64, 32, 3), (8, 128, 64, 5), (4, 256, 128, 17)])
def test_bf16_transposed_conv_data_gradient(H, Cin, Cout, B):
    """Data gradient of ConvTranspose2d(k3, s2, p1, op1) (a stride-2 gather over the 2H x 2W output gradient) with bf16 operands:
    d(in)[ci] = conv2d(bf16(dy), bf16(Wt), stride 2, pad 1) in fp64."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 55 + Cin)
    dy = torch.randn(G, B * 4 * H * H, Cout, generator=g).cuda()
    wt = (torch.randn(G, Cin, Cout, 3, 3, generator=g) * 0.1).cuda()          # nn.ConvTranspose2d layout per group
    st = torch.cuda.current_stream().cuda_stream
    pk = _pack(lib, L, wt, G, 3, Cout, Cin, st)
    out = torch.full((G, B * H * H, Cin), 3.0, device='cuda')
    cp = L.ConvParams(L.CONVT_DGRAD, L.IN_PLAIN, G, B, H, H, Cout, Cout, Cin, L.view(dy, Cout, 0, dy.stride(0)), None, None, 0,
                      L.NULL_VIEW, 0, L.CONV_BF16, None, pk.data_ptr(), pk.stride(0), None, 0, L.view(out, Cin, 0, out.stride(0)), None)
    L.check(lib.vv_conv_mfma(C.byref(cp), st), 'dgradT')
    for gi in range(G):
        dyn = dy[gi].view(B, 2 * H, 2 * H, Cout).permute(0, 3, 1, 2)
        ref = F.conv2d(_r(dyn).double(), _r(wt[gi]).double(), None, stride=2, padding=1)
        ref32 = F.conv2d(dyn.double(), wt[gi].double(), None, stride=2, padding=1)
        got = out[gi].view(B, H, H, Cin).permute(0, 3, 1, 2).double()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-4 * scale
        assert (got - ref32).abs().max().item() > 2e-4 * scale


@pytest.mark.parametrize('H,Cin,CinP,Cout,B,ks', [(32, 12, 16, 32, 2, 3), (32, 32, 32, 32, 3, 1), (32, 64, 64, 32, 2, 5), (32, 32, 32, 64, 1, 2),
                                                 (16, 64, 64, 64, 5, 4), (16, 128, 128, 64, 2, 1), (8, 128, 128, 128, 9, 2),
                                                 (8, 256, 256, 128, 3, 1), (4, 128, 128, 256, 33, 2), (4, 256, 256, 256, 16, 1), (4, 32, 32, 32, 5, 1)])
@pytest.mark.parametrize('stored16', [False, True])
def test_bf16_weight_gradient_matches_rounded_operand_reference(H, Cin, CinP, Cout, B, ks, stored16):
    """vv_wgrad_bf16 (+ vv_wgrad_reduce into the nn.Conv2d weight layout) against the float64 weight gradient of the
    bf16-rounded operands -- every (ci-blocks, co-blocks) workgroup shape (1x1, 2x1, 1x2, 2x2 blocks of 32 channels), ragged
    batch (images per tile do not divide B), zero-padded input channels, k-split > 1: <= 2e-4 of the tensor maximum; the
    unrounded operands are measurably somewhere else; two runs are bitwise identical.
    stored16: the layer input and dy are bf16 TENSORS (what the mixed-precision bank stores, BASELINE config 4) -- the LDS-ring kernel
    (global -> LDS DMA three tiles deep, BatchNorm+ReLU applied in place in LDS); the same bars."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 100 + Cin)
    x = torch.randn(G, B * H * H, CinP, generator=g)
    x[:, :, Cin:] = 0
    x = x.cuda()
    dy = torch.randn(G, B * H * H, Cout, generator=g).cuda()
    flags = 0
    xs, dys, esz = x, dy, 1
    if stored16:
        flags = L.WGRAD_DY_BF16 | L.WGRAD_X_BF16
        xs, dys = x.to(torch.bfloat16), dy.to(torch.bfloat16)
        x, dy = xs.float(), dys.float()          # the values the kernel sees
        esz = 2                                  # strides are given in fp32 units of the same buffer layout
    a = (torch.rand(G, CinP, generator=g) + 0.5).cuda()
    b = (torch.randn(G, CinP, generator=g) * 0.2).cuda()
    if CinP != Cin:
        b[:, Cin:] = 0
    st = torch.cuda.current_stream().cuda_stream
    nt, nblk, kw = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.vv_wgrad_bf16_plan(L.CONV3 | (flags << 8), B, H, H, CinP, Cout, C.byref(nt), C.byref(nblk), C.byref(kw)) == 1
    ks = min(ks, nt.value)
    nci, nco = (CinP + 31) // 32, Cout // 32
    outs = []
    for rep in range(2):
        part = torch.full((G, nci * nco * ks * kw.value * 9 * 1024), 7.0, device='cuda')
        grad = torch.zeros(G, Cout * Cin * 9, device='cuda')
        wp = L.WgradParams(L.CONV3, L.IN_ACT, G, B, H, H, Cin, CinP, Cout, ks, L.view(xs, CinP, 0, xs.stride(0) // esz), a.data_ptr(), b.data_ptr(),
                           CinP, L.NULL_VIEW, 0, flags, None, L.View(dys.data_ptr(), dys.stride(0) // esz, Cout, 0), part.data_ptr(), part.stride(0))
        L.check(lib.vv_wgrad_bf16(C.byref(wp), st), 'wgrad_bf16')
        L.check(lib.vv_wgrad_reduce(L.CONV3, G, Cin, CinP, Cout, ks * kw.value, part.data_ptr(), part.stride(0), grad.data_ptr(),
                                    grad.stride(0), st), 'reduce')
        outs.append(grad.view(G, Cout, Cin, 3, 3).clone())
    assert torch.equal(outs[0], outs[1])
    for gi in range(G):
        act = torch.relu(torch.addcmul(b[gi].view(1, -1, 1, 1), x[gi].view(B, H, H, CinP).permute(0, 3, 1, 2), a[gi].view(1, -1, 1, 1)))
        act = act[:, :Cin]
        dyn = dy[gi].view(B, H, H, Cout).permute(0, 3, 1, 2)
        ref = torch.nn.grad.conv2d_weight(_r(act).double(), (Cout, Cin, 3, 3), _r(dyn).double(), padding=1)
        ref32 = torch.nn.grad.conv2d_weight(act.double(), (Cout, Cin, 3, 3), dyn.double(), padding=1)
        scale = ref.abs().max().item()
        err = (outs[0][gi].double() - ref).abs().max().item()
        assert err <= 2e-4 * scale, (gi, err, scale)
        assert (outs[0][gi].double() - ref32).abs().max().item() > 2e-4 * scale


@pytest.mark.parametrize('H,Cin,Cout,B,ks', [(16, 64, 32, 3, 2), (8, 128, 64, 5, 1), (4, 256, 128, 17, 3), (8, 32, 32, 2, 1), (16, 32, 64, 1, 2)])
def test_bf16_transposed_conv_weight_gradient(H, Cin, Cout, B, ks):
    """vv_wgrad_bf16 with kind = VV_CONVT_FWD (weight gradient of ConvTranspose2d(k3,s2,p1,op1), H x W = its input): against the
    float64 weight gradient of the bf16-rounded operands in the nn.ConvTranspose2d layout [Cin][Cout][3][3]."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 31 + Cin)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    dy = torch.randn(G, B * 4 * H * H, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    st = torch.cuda.current_stream().cuda_stream
    nt, nblk, kw = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.vv_wgrad_bf16_plan(L.CONVT_FWD, B, H, H, Cin, Cout, C.byref(nt), C.byref(nblk), C.byref(kw)) == 1
    ks = min(ks, nt.value)
    nci, nco = Cin // 32, Cout // 32
    part = torch.full((G, nci * nco * ks * kw.value * 9 * 1024), 7.0, device='cuda')
    grad = torch.zeros(G, Cin * Cout * 9, device='cuda')
    wp = L.WgradParams(L.CONVT_FWD, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, ks, L.view(x, Cin, 0, x.stride(0)), a.data_ptr(), b.data_ptr(),
                       Cin, L.NULL_VIEW, 0, 0, None, L.View(dy.data_ptr(), dy.stride(0), Cout, 0), part.data_ptr(), part.stride(0))
    L.check(lib.vv_wgrad_bf16(C.byref(wp), st), 'wgradT_bf16')
    L.check(lib.vv_wgrad_reduce(L.CONVT_FWD, G, Cin, Cin, Cout, ks * kw.value, part.data_ptr(), part.stride(0), grad.data_ptr(),
                                grad.stride(0), st), 'reduce')
    got = grad.view(G, Cin, Cout, 3, 3)
    for gi in range(G):
        act = torch.relu(torch.addcmul(b[gi].view(1, -1, 1, 1), x[gi].view(B, H, H, Cin).permute(0, 3, 1, 2), a[gi].view(1, -1, 1, 1)))
        dyn = dy[gi].view(B, 2 * H, 2 * H, Cout).permute(0, 3, 1, 2)
        # conv_transpose2d(x, Wt) == the data gradient of conv2d(., Wt, stride 2, pad 1): dWt = conv2d_weight(dy, Wt.shape, x)
        ref = torch.nn.grad.conv2d_weight(_r(dyn).double(), (Cin, Cout, 3, 3), _r(act).double(), stride=2, padding=1)
        ref32 = torch.nn.grad.conv2d_weight(dyn.double(), (Cin, Cout, 3, 3), act.double(), stride=2, padding=1)
        scale = ref.abs().max().item()
        assert (got[gi].double() - ref).abs().max().item() <= 2e-4 * scale
        assert (got[gi].double() - ref32).abs().max().item() > 2e-4 * scale


def _as_bf16_storage(t):
    """the tensor's values rounded to bf16 and stored as bf16 at the start of an fp32-sized buffer (what the kernels write / read
    with their *_BF16 storage flags: same element indexing, group stride still counted in floats)"""
    buf = torch.zeros_like(t)
    flat = buf.view(t.shape[0], -1).view(torch.bfloat16)
    flat[:, :t[0].numel()] = t.reshape(t.shape[0], -1).to(torch.bfloat16)
    return buf


@pytest.mark.parametrize('H,Cin,Cout,B', [(32, 32, 64, 2), (8, 128, 128, 5), (4, 256, 128, 9)])
def test_bf16_stored_activation_gradients(H, Cin, Cout, B):
    """VV_CONV_OUT_BF16 / VV_BNBWD_DA_BF16: (a) a data-gradient launch that stores its output as bf16 writes exactly the
    rounded values of the fp32 launch, and its per-tile column sums are those of the stored values; (b) BatchNorm backward reading
    bf16 dA / dpool gives bit-identical dy, dgamma, dbeta to reading an fp32 tensor that holds the same (rounded) values."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H + Cin)
    st = torch.cuda.current_stream().cuda_stream
    # (a)
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    dy = torch.randn(G, B * H * H, Cout, generator=g).cuda()
    pk = _pack(lib, L, w, G, 1, Cout, Cin, st)
    nt = lib.vv_conv_ntiles2(B, H, H, L.CONV3, L.CONV_BF16)
    outs, stats = [], []
    for flag in (L.CONV_BF16, L.CONV_BF16 | L.CONV_OUT_BF16):
        o = torch.zeros(G, B * H * H, Cin, device='cuda')
        s_ = torch.zeros(G, nt, 2, Cin, device='cuda')
        cp = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cout, Cout, Cin, L.view(dy, Cout, 0, dy.stride(0)), None, None, 0, L.NULL_VIEW,
                          0, flag, None, pk.data_ptr(), pk.stride(0), None, 0, L.view(o, Cin, 0, o.stride(0)), s_.data_ptr())
        L.check(lib.vv_conv_mfma(C.byref(cp), st), 'dgrad')
        outs.append(o)
        stats.append(s_.sum(1)[:, 0])
    stored = outs[1].view(G, -1).view(torch.bfloat16)[:, :B * H * H * Cin].float().view(G, B * H * H, Cin)
    assert torch.equal(stored, _r(outs[0]))
    torch.testing.assert_close(stats[1], stored.double().sum(1).float(), rtol=1e-4, atol=1e-3)
    # (b)
    Cc = Cin
    y = torch.randn(G, B * H * H, Cc, generator=g).cuda()
    dA = _r(torch.randn(G, B * H * H, Cc, generator=g)).cuda()
    dP = _r(torch.randn(G, B * (H // 2) * (H // 2), Cc, generator=g)).cuda()
    a = (torch.rand(G, Cc, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cc, generator=g) * 0.2).cuda()
    mean = torch.randn(G, Cc, generator=g).cuda() * 0.1
    invstd = (torch.rand(G, Cc, generator=g) + 0.5).cuda()
    gamma = (torch.rand(G, Cc, generator=g) + 0.5).cuda()
    nblk = lib.vv_bn_bwd_nblk(B, H, H, Cc)
    for pool in (False, True):
        res = []
        for da16 in (False, True):
            dAx, dPx = (_as_bf16_storage(dA), _as_bf16_storage(dP)) if da16 else (dA, dP)
            dz = torch.zeros(G, B * H * H, Cc, device='cuda')
            part = torch.zeros(G, nblk * 2 * Cc, device='cuda')
            dgm, dbt, scr = torch.zeros(G, Cc, device='cuda'), torch.zeros(G, Cc, device='cuda'), torch.zeros(G, 2 * Cc, device='cuda')
            bp = L.BnBwdParams(G, B, H, H, Cc, L.BNBWD_DA_BF16 if da16 else 0, y.data_ptr(), y.stride(0), a.data_ptr(), b.data_ptr(),
                               mean.data_ptr(), invstd.data_ptr(), Cc, L.view(dAx, Cc, 0, dAx.stride(0)),
                               dPx.data_ptr() if pool else None, dPx.stride(0) if pool else 0, dz.data_ptr(), dz.stride(0), part.data_ptr())
            L.check(lib.vv_bn_bwd_reduce(C.byref(bp), st), 'reduce')
            L.check(lib.vv_bn_bwd_apply(C.byref(bp), gamma.data_ptr(), Cc, dgm.data_ptr(), dbt.data_ptr(), Cc, scr.data_ptr(), st), 'apply')
            res.append((dz, dgm, dbt))
        for x, z in zip(res[0], res[1]):
            assert torch.equal(x, z)
        assert res[0][0].abs().max() > 0
        # (c) every tensor bf16 (y, dA, dpool in; dy out): the 8-channel-per-lane kernel against the path above fed with fp32 tensors
        # that hold the same rounded values -- another summation order of the partial sums, so fp32 round-off on dgamma / dbeta and
        # at most a bf16 ulp on dy
        yr = _r(y)
        dz0 = torch.zeros(G, B * H * H, Cc, device='cuda')
        part = torch.zeros(G, nblk * 2 * Cc, device='cuda')
        g0, b0, scr = torch.zeros(G, Cc, device='cuda'), torch.zeros(G, Cc, device='cuda'), torch.zeros(G, 2 * Cc, device='cuda')
        bp = L.BnBwdParams(G, B, H, H, Cc, 0, yr.data_ptr(), yr.stride(0), a.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(), Cc,
                           L.view(dA, Cc, 0, dA.stride(0)), dP.data_ptr() if pool else None, dP.stride(0) if pool else 0,
                           dz0.data_ptr(), dz0.stride(0), part.data_ptr())
        L.check(lib.vv_bn_bwd_reduce(C.byref(bp), st), 'reduce')
        L.check(lib.vv_bn_bwd_apply(C.byref(bp), gamma.data_ptr(), Cc, g0.data_ptr(), b0.data_ptr(), Cc, scr.data_ptr(), st), 'apply')
        y16, dA16, dP16 = _as_bf16_storage(y), _as_bf16_storage(dA), _as_bf16_storage(dP)
        dz1 = torch.zeros(G, B * H * H, Cc, device='cuda')
        g1, b1 = torch.zeros(G, Cc, device='cuda'), torch.zeros(G, Cc, device='cuda')
        bp = L.BnBwdParams(G, B, H, H, Cc, L.BNBWD_DA_BF16 | L.BNBWD_Y_BF16 | L.BNBWD_DZ_BF16, y16.data_ptr(), y16.stride(0), a.data_ptr(),
                           b.data_ptr(), mean.data_ptr(), invstd.data_ptr(), Cc, L.view(dA16, Cc, 0, dA16.stride(0)),
                           dP16.data_ptr() if pool else None, dP16.stride(0) if pool else 0, dz1.data_ptr(), dz1.stride(0), part.data_ptr())
        L.check(lib.vv_bn_bwd_reduce(C.byref(bp), st), 'reduce16')
        L.check(lib.vv_bn_bwd_apply(C.byref(bp), gamma.data_ptr(), Cc, g1.data_ptr(), b1.data_ptr(), Cc, scr.data_ptr(), st), 'apply16')
        got = dz1.view(G, -1).view(torch.bfloat16)[:, :B * H * H * Cc].float().view(G, B * H * H, Cc)
        torch.testing.assert_close(g1, g0, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(b1, b0, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(got, dz0, rtol=2 ** -7, atol=1e-4)


def _build_bf16(monkeypatch, kind='net4'):
    monkeypatch.setenv('VV_PRECISION', 'bf16')
    from test_gpu_unet import _build
    net, sd, tot_of = _build(kind, False)
    assert net.bank().precision == 'bf16'
    return net, sd, tot_of


def test_bf16_forward_matches_mixed_oracle(monkeypatch):
    """Whole Net4 bank, eval mode, vs the oracle restating the same operand roundings.  A value that sits on a bf16 rounding
    boundary can round differently in the two implementations (their fp32 inputs differ by ~1e-7) and that 0.4 % step then
    travels down the network, so whole-network agreement is statistical: rms <= 1e-3 of the output rms (observed 2.4e-4), at
    least 3x closer to the mixed oracle than to the fp32 oracle (observed 4.8x), per-cube scores rel <= 1e-2."""
    from oracle import unet_oracle as O
    net, sd, tot_of = _build_bf16(monkeypatch)
    raw, flow = O.seeded_cubes(6, tot_of, 0)
    x, x_of = O.cubes_to_inputs(raw, flow)
    spec = O.bank_spec('net4')
    net.eval()
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = net(x.cuda(), x_of.cuda())
    err = {}
    for tag, mixed in (('mixed', O.MIXED_BF16), ('fp32', None)):
        monkeypatch.setattr(O, 'MIXED', mixed)
        with torch.no_grad():
            oo, ro, ot, rt = O.bank_forward({k: v.clone() for k, v in sd.items()}, spec, x, x_of, False, False)
        err[tag] = max(float((raw_o.cpu() - ro).pow(2).mean().sqrt() / ro.pow(2).mean().sqrt()),
                       float((of_o.cpu() - oo).pow(2).mean().sqrt() / oo.pow(2).mean().sqrt()))
        if mixed is not None:
            rs = ((raw_t - raw_o) ** 2).sum(dim=(1, 2, 3)).cpu().numpy()
            np.testing.assert_allclose(rs, O.cube_scores(ro, rt).numpy(), rtol=1e-2)
    assert err['mixed'] <= 1e-3, err
    assert err['fp32'] >= 3 * err['mixed'], err


def test_bf16_train_steps_match_mixed_oracle(monkeypatch):
    """Net4 in the mixed mode vs the mixed oracle through the reference's loop shape (train.py:383-402: model(x, x_of) ->
    MSELoss -> backward -> Adam).  Train-mode BatchNorm over 6 cubes amplifies the rounding-boundary steps described above
    (forward rms 4e-3), and the gradient of an untrained net amplifies them again, so this is a sanity bar on the direction
    and size of the whole gradient -- cosine >= 0.98, norm ratio within 5 % -- and rel 2e-3 on the 3-step loss trajectory;
    the operation-level tests above are the parity tests proper."""
    from oracle import unet_oracle as O
    net, sd, tot_of = _build_bf16(monkeypatch)
    raw, flow = O.seeded_cubes(6, tot_of, 0)
    x, x_of = O.cubes_to_inputs(raw, flow)
    spec = O.bank_spec('net4')
    monkeypatch.setattr(O, 'MIXED', O.MIXED_BF16)
    sdo = {k: v.clone() for k, v in sd.items()}
    opt = O.AdamState(O.param_names(sdo))
    out = [O.train_step(sdo, spec, x, x_of, opt) for _ in range(3)]
    ref_losses, ref_grads = np.array([[o[0], o[1]] for o in out]), out[0][2]
    monkeypatch.setattr(O, 'MIXED', None)
    net.train()
    opt = torch.optim.Adam(net.parameters(), eps=1e-7, weight_decay=0.0)
    lf = torch.nn.MSELoss()
    losses = []
    for step in range(3):
        of_o, raw_o, of_t, raw_t = net(x.cuda(), x_of.cuda())
        l_raw, l_of = lf(raw_t.detach(), raw_o), lf(of_t.detach(), of_o)
        losses.append([l_raw.item(), l_of.item()])
        opt.zero_grad()
        (l_raw + l_of).backward()
        if step == 0:
            dot = n1 = n2 = 0.0
            for k, p_ in net.named_parameters():
                gref = ref_grads.get(k)
                if gref is None or k.endswith('.0.bias') or k.endswith('.3.bias'):
                    continue                      # conv bias in front of BN: mathematically zero
                gg = p_.grad.cpu().double()
                dot += float((gg * gref.double()).sum())
                n1 += float((gg ** 2).sum())
                n2 += float((gref.double() ** 2).sum())
            assert dot / (n1 * n2) ** 0.5 >= 0.98, dot / (n1 * n2) ** 0.5
            assert abs((n1 / n2) ** 0.5 - 1.0) <= 5e-2, (n1, n2)
        opt.step()
    np.testing.assert_allclose(np.array(losses), ref_losses, rtol=2e-3)


def test_bf16_dy_storage_is_bitwise_neutral(monkeypatch):
    """In the mixed mode BatchNorm backward stores dy as bf16 because both of its consumers (data gradient, weight gradient)
    round it to bf16 on load anyway: the gradients of a whole backward pass must be bit-identical with and without that
    (VV_BF16_DZ=0 keeps the fp32 tensor), Full bank, ragged batch."""
    from oracle import unet_oracle as O
    from test_gpu_unet import _build
    raw, flow = O.seeded_cubes(37, 5, 9, smooth=False)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    grads = []
    monkeypatch.setenv('VV_PRECISION', 'bf16')
    monkeypatch.setenv('VV_BF16_DA', '0')          # the other (non-neutral) storage choices off: this test isolates dy
    monkeypatch.setenv('VV_BF16_Y', '0')
    for dz in ('1', '0'):
        monkeypatch.setenv('VV_BF16_DZ', dz)
        net, sd, _ = _build('full', False)
        net.train()
        bank = net.bank()
        assert bank.dz16 == (dz == '1')
        ws = bank.set_input_cubes(rawd, flowd, None, 37)
        bank.forward(ws, True)
        bank.backward(ws)
        grads.append(bank.grads.clone())
    assert torch.isfinite(grads[0]).all() and grads[0].abs().max() > 0
    assert torch.equal(grads[0], grads[1])


def test_bf16_odd_batch_sizes(monkeypatch):
    """ragged batches in the mixed mode (images per tile do not divide B on any level; B = 1 in eval mode as test.py scores a
    single cube): losses / scores against the mixed oracle, rel 5e-3 (2e-2 for B = 2: BatchNorm over two cubes at 4x4 amplifies
    the rounding-boundary steps)."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    spec = O.bank_spec('net4')
    for B in (2, 7, 17):
        net, sd, tot_of = _build_bf16(monkeypatch)
        raw, flow = O.seeded_cubes(B, 1, 11)
        x, x_of = O.cubes_to_inputs(raw, flow)
        net.train()
        tr = FusedTrainer(net)
        ws = tr.step_nchw(x.cuda(), x_of.cuda())
        l_raw, l_of = (float(v) for v in tr.losses(ws))
        monkeypatch.setattr(O, 'MIXED', O.MIXED_BF16)
        lr, lo, _ = O.train_step(sd, spec, x, x_of, O.AdamState(O.param_names(sd)))
        monkeypatch.setattr(O, 'MIXED', None)
        tol = 2e-2 if B == 2 else 5e-3
        assert abs(l_raw - lr) <= tol * lr and abs(l_of - lo) <= tol * lo, (B, l_raw, lr, l_of, lo)
        assert torch.isfinite(tr.bank.params).all()
    net, sd, tot_of = _build_bf16(monkeypatch)
    net.eval()
    raw, flow = O.seeded_cubes(1, 1, 12)
    x, x_of = O.cubes_to_inputs(raw, flow)
    r, o = FusedTrainer(net).score_nchw(x.cuda(), x_of.cuda())
    monkeypatch.setattr(O, 'MIXED', O.MIXED_BF16)
    rs, os_ = O.score_pass(sd, spec, x, x_of, 1)
    monkeypatch.setattr(O, 'MIXED', None)
    np.testing.assert_allclose(r.cpu().numpy(), rs, rtol=1e-2)
    np.testing.assert_allclose(o.cpu().numpy(), os_, rtol=1e-2)


def test_bf16_scores_and_auc_close_to_fp32(monkeypatch):
    """Config 4's bar: the bf16 path is judged on AUROC.  Same weights, same cubes, eval mode: per-cube
    scores of the two precisions within 10 % of each other after 40 training steps each (observed: 1 cube of 96 beyond 5 %), and the AUROC of a labelled cube set
    within 1e-2."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    from test_gpu_unet import _build
    n = 96
    raw, flow = O.seeded_cubes(n, 1, 5)
    raw = raw.copy()
    raw[n // 2:, :, 8:24, 8:24] = 255 - raw[n // 2:, :, 8:24, 8:24]           # "anomalies": a patch inverted
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    labels = np.zeros(n, dtype=np.int64)
    labels[n // 2:] = 1
    res = {}
    for prec in ('fp32', 'bf16'):
        monkeypatch.setenv('VV_PRECISION', prec)
        net, sd, _ = _build('net4', False)
        assert net.bank().precision == prec
        net.train()
        tr = FusedTrainer(net)
        for _ in range(40):
            tr.step_cubes(rawd, flowd, torch.arange(n // 2, device='cuda'))
        net.eval()
        r, o = FusedTrainer(net, reset_optimizer=False).score_cubes(rawd, flowd)
        res[prec] = (r.cpu().numpy(), o.cpu().numpy())
    for i in range(2):
        np.testing.assert_allclose(res['bf16'][i], res['fp32'][i], rtol=1e-1)
    aucs = [O.roc_auc(res[p][0] + res[p][1], labels) for p in ('fp32', 'bf16')]
    assert abs(aucs[0] - aucs[1]) <= 1e-2, aucs
    assert aucs[0] > 0.6                                                        # the labelled set is separable at all


@pytest.mark.parametrize('H,Cin,Cout,B,mode', [
    (32, 16, 32, 3, 'plain'), (32, 32, 32, 2, 'act'), (32, 64, 32, 2, 'cat'), (32, 32, 64, 1, 'plain'),
    (16, 32, 64, 3, 'plain'), (16, 64, 64, 5, 'act'), (16, 128, 64, 2, 'cat'), (16, 64, 128, 1, 'plain'), (16, 64, 32, 2, 'plain'),
    (8, 64, 128, 9, 'plain'), (8, 128, 128, 6, 'act'), (8, 256, 128, 5, 'cat'), (8, 128, 256, 3, 'plain'), (8, 128, 64, 7, 'plain'),
    (4, 128, 256, 33, 'plain'), (4, 256, 256, 17, 'act'), (4, 256, 128, 40, 'plain'), (4, 32, 32, 5, 'act')])
def test_bf16_gemm_conv_all_bf16_tensors(H, Cin, Cout, B, mode):
    """vv_conv_bf16.hip (round 4): the GEMM-shaped 3x3 kernel that all-bf16 launches run (VV_CONV_BF16 | VV_CONV_OUT_BF16 |
    VV_CONV_ALLSRC_BF16) -- every tile shape (TN = 32 / 64 / 128 on all four pyramid levels), every input mode it serves (plain =
    data gradient / pooled / frame-erased input, BatchNorm+ReLU on load, skip concat), ragged batches (tiles of 4 / 16 images).
    (a) against an fp64 convolution of the bf16-rounded operands: 2e-4 of the tensor maximum before the output rounding, i.e.
        the stored bf16 value is within one bf16 ulp (2^-8 relative) of the rounded reference;
    (b) BIT-EQUAL to the round-3 kernel (VV_CONV_NO_GEMM16) on the same inputs: both accumulate chunk -> tap -> one 16-channel
        MFMA in fp32 in the same order, so the new data path (filter fragments from L2, permuted two-plane LDS tile) must not move
        a single bit;
    (c) per-tile column sums / sums of squares of the stored values add up to the tensor's."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 1000 + Cin + Cout)
    st = torch.cuda.current_stream().cuda_stream
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    bias = torch.randn(G, Cout, generator=g).cuda()
    csplit = Cin // 2 if mode == 'cat' else Cin
    x0 = _r(torch.randn(G, B * H * H, csplit, generator=g)).cuda()
    x1 = _r(torch.randn(G, B * H * H, Cin - csplit, generator=g)).cuda() if mode == 'cat' else None
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    x0s = _as_bf16_storage(x0)
    x1s = _as_bf16_storage(x1) if x1 is not None else None
    pk = _pack(lib, L, w, G, 0, Cin, Cout, st)
    in_mode = {'plain': L.IN_PLAIN, 'act': L.IN_ACT, 'cat': L.IN_CAT}[mode]
    res = []
    for extra in (0, L.CONV_NO_GEMM16):
        flags = L.CONV_BF16 | L.CONV_OUT_BF16 | L.CONV_ALLSRC_BF16 | extra
        nt = lib.vv_conv_ntiles2(B, H, H, L.CONV3, flags)
        y = torch.full((G, B * H * H * Cout // 2 + 8,), 3.0, device='cuda')
        s_ = torch.full((G, nt, 2, Cout), -7.0, device='cuda')
        cp = L.ConvParams(L.CONV3, in_mode, G, B, H, H, Cin, Cin, Cout, L.View(x0s.data_ptr(), x0s.stride(0), csplit, 0),
                          a.data_ptr() if mode != 'plain' else None, b.data_ptr() if mode != 'plain' else None, Cin,
                          L.View(x1s.data_ptr(), x1s.stride(0), Cin - csplit, 0) if x1s is not None else L.NULL_VIEW, csplit, flags, None,
                          pk.data_ptr(), pk.stride(0), bias.data_ptr(), Cout, L.View(y.data_ptr(), y.stride(0), Cout, 0), s_.data_ptr())
        L.check(lib.vv_conv_mfma(C.byref(cp), st), 'conv')
        stored = y.view(torch.bfloat16)[:, :B * H * H * Cout].float().view(G, B * H * H, Cout)
        assert float(y[0, -1]) == 3.0                                   # nothing written past the tensor
        res.append((stored, s_))
    assert lib.vv_conv_ntiles2(B, H, H, L.CONV3, L.CONV_BF16 | L.CONV_OUT_BF16 | L.CONV_ALLSRC_BF16) == lib.vv_conv_ntiles(B, H, H)
    new, old = res[0][0], res[1][0]
    assert torch.equal(new, old), (new - old).abs().max().item()       # (b)
    for gi in range(G):
        xin = x0[gi].view(B, H, H, csplit).permute(0, 3, 1, 2)
        if mode != 'plain':
            xin = torch.relu(torch.addcmul(b[gi, :csplit].view(1, -1, 1, 1), xin, a[gi, :csplit].view(1, -1, 1, 1)))
        if mode == 'cat':
            xin = torch.cat([xin, x1[gi].view(B, H, H, Cin - csplit).permute(0, 3, 1, 2)], 1)
        ref = F.conv2d(_r(xin).double(), _r(w[gi]).double(), bias[gi].double(), padding=1)
        got = new[gi].view(B, H, H, Cout).permute(0, 3, 1, 2).double()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= (2e-4 + 2 ** -8) * scale, ((got - ref).abs().max().item(), scale)     # (a)
        tot = res[0][1][gi].sum(0).double()                              # (c)
        torch.testing.assert_close(tot[0], new[gi].double().sum(0), rtol=1e-4, atol=1e-3 * scale * B)
        torch.testing.assert_close(tot[1], (new[gi].double() ** 2).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('H,Cin,Cout,B', [(32, 32, 64, 3), (16, 64, 128, 2), (16, 64, 64, 3), (8, 128, 256, 5), (8, 64, 128, 6),
                                          (4, 256, 256, 19), (4, 64, 64, 33)])
@pytest.mark.parametrize('legacy', [False, True])
def test_bf16_conv_second_output_view(H, Cin, Cout, B, legacy):
    """vv_conv_params.out1 / osplit (round 4): a concat layer's data gradient leaves as two dense tensors, channels [0, Cout/2) and
    [Cout/2, Cout).  Both bf16-output kernels (the GEMM-shaped one: N tile inside a half or spanning both; the round-3 kernel, which
    keeps the 32x32 level) must store exactly the values of the single-view launch, write nothing else, and leave the same per-tile
    column sums (the transposed conv's bias gradient reads them over ALL channels)."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G, half = 2, Cout // 2
    g = torch.Generator(device='cpu').manual_seed(H * 77 + Cin + Cout)
    st = torch.cuda.current_stream().cuda_stream
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    x0s = _as_bf16_storage(_r(torch.randn(G, B * H * H, Cin, generator=g)).cuda())
    pk = _pack(lib, L, w, G, 0, Cin, Cout, st)
    flags = L.CONV_BF16 | L.CONV_OUT_BF16 | L.CONV_ALLSRC_BF16 | (L.CONV_NO_GEMM16 if legacy else 0)
    nt = lib.vv_conv_ntiles2(B, H, H, L.CONV3, flags)
    M = B * H * H

    def run(split):
        y = torch.full((G, M * Cout // 2 + 8,), 3.0, device='cuda')
        s_ = torch.full((G, nt, 2, Cout), -7.0, device='cuda')
        cp = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cin, Cin, Cout, L.View(x0s.data_ptr(), x0s.stride(0), Cin, 0), None, None, Cin,
                          L.NULL_VIEW, Cin, flags, None, pk.data_ptr(), pk.stride(0), None, 0, L.View(y.data_ptr(), y.stride(0), Cout, 0),
                          s_.data_ptr())
        if split:
            cp.out = L.View(y.data_ptr(), y.stride(0), half, 0)
            cp.out1 = L.View(y.data_ptr() + M * half * 2, y.stride(0), half, 0)
            cp.osplit = half
        L.check(lib.vv_conv_mfma(C.byref(cp), st), 'conv')
        torch.cuda.synchronize()
        assert float(y[0, -1]) == 3.0 and float(y[1, -1]) == 3.0
        return y.view(torch.bfloat16)[:, :M * Cout], s_

    one, s_one = run(False)
    two, s_two = run(True)
    one = one.view(G, M, Cout)
    assert torch.equal(two[:, :M * half].view(G, M, half), one[:, :, :half])
    assert torch.equal(two[:, M * half:].view(G, M, half), one[:, :, half:])
    assert torch.equal(s_one, s_two)
    # the fp32 / non-bf16-output paths refuse the second view instead of ignoring it
    y = torch.zeros(G, M * Cout + 8, device='cuda')
    cp = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cin, Cin, Cout, L.View(x0s.data_ptr(), x0s.stride(0), Cin, 0), None, None, Cin,
                      L.NULL_VIEW, Cin, L.CONV_BF16, None, pk.data_ptr(), pk.stride(0), None, 0, L.View(y.data_ptr(), y.stride(0), half, 0), None)
    cp.out1, cp.osplit = L.View(y.data_ptr() + M * half * 4, y.stride(0), half, 0), half
    assert lib.vv_conv_mfma(C.byref(cp), st) != 0


@pytest.mark.parametrize('B,Cin,Cout,mode,variant', [(90, 32, 32, 'act', 'stats'), (90, 32, 32, 'plain', 'bnf'), (90, 16, 32, 'plain', 'stats'),
                                                     (47, 32, 64, 'plain', 'split'), (90, 16, 32, 'act', 'relu'), (128, 32, 32, 'plain', 'none')])
def test_bf16_ring_conv_bitwise_equal(B, Cin, Cout, mode, variant):
    """conv_ring16_kernel (round 5, csrc/vv_conv_ring16.hip: persistent workgroups, halo of the next tile by LDS-DMA, filter in
    registers) against conv_mfma_kernel<.., BF> (VV_CONV_NO_RING) on the same all-bf16 tensors of the 32x32 level: output, BatchNorm
    column sums, the fused BatchNorm-backward sums and the two-plane output of a concat layer's data gradient agree BIT FOR BIT, for
    runs of tiles that cross UNet / N-tile boundaries inside a workgroup."""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G, H = 6, 32
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cpu').manual_seed(B * 100 + Cin + Cout)
    M = B * H * H
    xs = _as_bf16_storage(_r(torch.randn(G, M, Cin, generator=g)).cuda())
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    bias = torch.randn(G, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    zs = _as_bf16_storage(_r(torch.randn(G, M, Cout, generator=g)).cuda())
    bn = [(torch.rand(G, Cout, generator=g) + 0.5).cuda() for _ in range(4)]
    pk = _pack(lib, L, w, G, 0, Cin, Cout, st)
    base = L.CONV_BF16 | L.CONV_OUT_BF16 | L.CONV_ALLSRC_BF16 | (L.CONV_RELU if variant == 'relu' else 0)
    nt = lib.vv_conv_ntiles2(B, H, H, L.CONV3, base)
    assert G * (Cout // 32) * nt >= 4 * 512
    outs = []
    for flag in (0, L.CONV_NO_RING):
        y = torch.full((G, M * Cout // 2 + 8), float('nan'), device='cuda')
        s_ = torch.full((G, nt, 2, Cout), float('nan'), device='cuda')
        use_stats = variant in ('stats', 'split', 'relu')
        cp = L.ConvParams(L.CONV3, L.IN_ACT if mode == 'act' else L.IN_PLAIN, G, B, H, H, Cin, Cin, Cout,
                          L.View(xs.data_ptr(), xs.stride(0), Cin, 0), a.data_ptr(), b.data_ptr(), Cin, L.NULL_VIEW, 0, base | flag, None,
                          pk.data_ptr(), pk.stride(0), bias.data_ptr(), Cout, L.View(y.data_ptr(), y.stride(0), Cout, 0),
                          s_.data_ptr() if use_stats else None)
        if variant == 'split':
            half = Cout // 2
            cp.out = L.View(y.data_ptr(), y.stride(0), half, 0)
            cp.out1 = L.View(y.data_ptr() + M * half * 2, y.stride(0), half, 0)
            cp.osplit = half
        if variant == 'bnf':
            cp.bn_z, cp.bn_z_gstride = zs.data_ptr(), zs.stride(0)
            cp.bn_a, cp.bn_b, cp.bn_mean, cp.bn_invstd = (t.data_ptr() for t in bn)
            cp.bn_gstride, cp.bn_partial = Cout, s_.data_ptr()
        L.check(lib.vv_conv_mfma(C.byref(cp), st), 'conv')
        torch.cuda.synchronize()
        assert float(y[0, -1]) != float(y[0, -1])            # nothing written behind the tensor (still NaN)
        outs.append((y.view(torch.int16)[:, :M * Cout].clone(), s_.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    if variant != 'none':
        assert not torch.isnan(outs[0][1]).any()
        assert torch.equal(outs[0][1], outs[1][1])
