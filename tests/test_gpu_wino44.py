"""vv_conv_wino44 (round 5): the Winograd F(4x4,3x3) form of the UNet bank's 3x3 convolution (model/unet.py:10,13; forward and
data gradient) at kernel level, through the C ABI, against the direct implicit-GEMM kernel (vv_conv_mfma) and a float64 torch
convolution of the same tensors.  Tolerance: F(4x4)'s transform constants (up to 8) leave a few 1e-6 of the tensor's maximum
(profiles/r05_wino44_numerics.txt); the bar here is 5e-5 of the maximum."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

BAR = 5e-5


def _pack(lib, L, fn, w, G, mode, K, N, taps, st):
    ent = (L.PackEntry * 1)(L.PackEntry(0, 0, mode, K, K, N))
    tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
    out = torch.zeros(G, taps * K * N, device='cuda')
    L.check(fn(tab.data_ptr(), 1, G, w.data_ptr(), w[0].numel(), out.data_ptr(), out.stride(0), (9 if taps == 9 else 1) * K * N, st), 'pack')
    return out


@pytest.mark.parametrize('H,Cin,Cout,B', [(32, 16, 32, 3), (32, 64, 32, 2), (32, 32, 64, 1), (16, 64, 64, 5), (16, 128, 64, 2), (8, 256, 128, 9),
                                           (8, 128, 128, 8), (4, 128, 256, 33), (4, 256, 256, 70), (16, 32, 64, 1),
                                           # two pixel tiles per workgroup with an odd tile count (the second group idles) / with three N tiles
                                           (16, 64, 32, 5), (16, 64, 96, 3), (8, 32, 96, 7)])
def test_wino44_matches_direct_conv_and_float64(H, Cin, Cout, B):
    """forward panel (BatchNorm+ReLU-on-load input, bias, BatchNorm partial sums) and data-gradient panel (plain input), ragged last
    workgroup (B not a multiple of the images per workgroup), every level"""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 1000 + Cin + Cout)
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
        pd = _pack(lib, L, lib.vv_pack_weights, w, G, 1 if dgrad else 0, K, N, 9, st)
        pw = _pack(lib, L, lib.vv_pack_wino44, w, G, 1 if dgrad else 0, K, N, 36, st)
        outs, stats = [], []
        for fn, pk, nt in ((lib.vv_conv_mfma, pd, lib.vv_conv_ntiles(B, H, H)), (lib.vv_conv_wino44, pw, lib.vv_wino44_ntiles(B, H))):
            y = torch.full((G, B * H * H, N), 3.0, device='cuda')
            s_ = torch.zeros(G, nt, 2, N, device='cuda')
            mode = L.IN_PLAIN if dgrad else L.IN_ACT
            cp = L.ConvParams(L.CONV3, mode, G, B, H, H, K, K, N, L.view(src, K, 0, src.stride(0)),
                              None if dgrad else a.data_ptr(), None if dgrad else b.data_ptr(), K, L.NULL_VIEW, 0, 0, None,
                              pk.data_ptr(), pk.stride(0), None if dgrad else bias.data_ptr(), N, L.view(y, N, 0, y.stride(0)),
                              None if dgrad else s_.data_ptr())
            L.check(fn(C.byref(cp), st), 'conv')
            outs.append(y)
            stats.append(s_.sum(1))
        scale = outs[0].abs().max().item()
        err = (outs[0] - outs[1]).abs().max().item()
        assert err <= BAR * scale, (dgrad, err, scale)
        # float64 reference of group 0
        xin = (torch.relu(src[0] * a[0] + b[0]) if not dgrad else src[0]).double().view(B, H, H, K).permute(0, 3, 1, 2)
        wd = w[0].double()
        if dgrad:
            wd = wd.flip(2, 3).transpose(0, 1)
        yr = torch.nn.functional.conv2d(xin, wd, None if dgrad else bias[0].double(), padding=1).permute(0, 2, 3, 1).reshape(B * H * H, N)
        assert (yr - outs[1][0].double()).abs().max().item() <= BAR * scale
        if not dgrad:
            torch.testing.assert_close(stats[0], stats[1], rtol=2e-4, atol=2e-3 * max(1.0, scale))


@pytest.mark.parametrize('H,C0,C1,Cout,B', [(32, 32, 32, 32, 3), (16, 64, 64, 64, 3), (8, 128, 128, 128, 5), (4, 128, 128, 256, 9)])
def test_wino44_concat_input_fused_bn_sums_and_relu(H, C0, C1, Cout, B):
    """(1) concat input (VV_IN_CAT, the decoder's first conv, model/unet.py:57-60) against float64; (2) a data-gradient launch that
    also leaves the BatchNorm-backward partial sums of its consumer (bn_partial) -- the output is untouched by the fusion and the sums
    match a float64 evaluation on the launch's own output; (3) VV_CONV_RELU (the folded eval path)"""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 77 + Cout)
    st = torch.cuda.current_stream().cuda_stream
    Cin = C0 + C1
    x0 = torch.randn(G, B * H * H, C0, generator=g).cuda()
    x1 = torch.randn(G, B * H * H, C1, generator=g).cuda()
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    bias = torch.randn(G, Cout, generator=g).cuda()
    pk = _pack(lib, L, lib.vv_pack_wino44, w, G, 0, Cin, Cout, 36, st)
    ref = torch.cat([torch.relu(x0 * a[:, None, :C0] + b[:, None, :C0]), x1], 2)
    for relu in (False, True):
        y = torch.full((G, B * H * H, Cout), 3.0, device='cuda')
        cp = L.ConvParams(L.CONV3, L.IN_CAT, G, B, H, H, Cin, Cin, Cout, L.view(x0, C0, 0, x0.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                          L.view(x1, C1, 0, x1.stride(0)), C0, L.CONV_RELU if relu else 0, None, pk.data_ptr(), pk.stride(0), bias.data_ptr(), Cout,
                          L.view(y, Cout, 0, y.stride(0)), None)
        L.check(lib.vv_conv_wino44(C.byref(cp), st), 'conv cat')
        for gi in range(G):
            yr = torch.nn.functional.conv2d(ref[gi].view(B, H, H, Cin).permute(0, 3, 1, 2).double(), w[gi].double(), bias[gi].double(), padding=1)
            yr = yr.permute(0, 2, 3, 1).reshape(B * H * H, Cout)
            scale = yr.abs().max().item()
            if relu:
                yr = torch.relu(yr)
            assert (yr - y[gi].double()).abs().max().item() <= BAR * scale

    dy = torch.randn(G, B * H * H, Cout, generator=g).cuda()
    z = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    mean = (torch.randn(G, Cin, generator=g) * 0.1).cuda()
    invstd = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    pk = _pack(lib, L, lib.vv_pack_wino44, w, G, 1, Cout, Cin, 36, st)
    nt = lib.vv_wino44_ntiles(B, H)
    dA = torch.zeros(G, B * H * H, Cin, device='cuda')
    part = torch.full((G, nt, 2, Cin), 7.0, device='cuda')
    cp = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cout, Cout, Cin, L.view(dy, Cout, 0, dy.stride(0)), None, None, 0, L.NULL_VIEW, 0, 0,
                      None, pk.data_ptr(), pk.stride(0), None, 0, L.view(dA, Cin, 0, dA.stride(0)), None,
                      z.data_ptr(), z.stride(0), a.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(), Cin, part.data_ptr())
    L.check(lib.vv_conv_wino44(C.byref(cp), st), 'dgrad + bn sums')
    bare = torch.zeros_like(dA)
    cp2 = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cout, Cout, Cin, L.view(dy, Cout, 0, dy.stride(0)), None, None, 0, L.NULL_VIEW, 0, 0,
                       None, pk.data_ptr(), pk.stride(0), None, 0, L.view(bare, Cin, 0, bare.stride(0)), None)
    L.check(lib.vv_conv_wino44(C.byref(cp2), st), 'dgrad')
    assert torch.equal(dA, bare)                                   # the fused sums do not touch the output
    d = dA.double() * ((a[:, None] * z + b[:, None]) > 0)
    xh = (z.double() - mean[:, None].double()) * invstd[:, None].double()
    s = part.double().sum(1)
    torch.testing.assert_close(s[:, 0], d.sum(1), rtol=1e-4, atol=1e-3 * d.abs().sum(1).max().item() / (B * H * H) ** 0.5)
    torch.testing.assert_close(s[:, 1], (d * xh).sum(1), rtol=1e-4, atol=1e-3 * (d * xh).abs().sum(1).max().item() / (B * H * H) ** 0.5)
    cp.stats = part.data_ptr()                                     # a stats pointer and bn_partial together are refused
    assert lib.vv_conv_wino44(C.byref(cp), st) != 0


def test_wino44_output_into_channel_slice_and_determinism():
    """the output view's channel stride / offset (a conv writing into a wider buffer) and run-to-run bit equality"""
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G, B, H, Cin, Cout, CS, CO = 3, 4, 16, 64, 32, 96, 32
    g = torch.Generator(device='cpu').manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    pk = _pack(lib, L, lib.vv_pack_wino44, w, G, 0, Cin, Cout, 36, st)
    res = []
    for rep in range(2):
        y = torch.full((G, B * H * H, CS), 3.0, device='cuda')
        cp = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cin, Cin, Cout, L.view(x, Cin, 0, x.stride(0)), None, None, 0, L.NULL_VIEW, 0, 0, None,
                          pk.data_ptr(), pk.stride(0), None, 0, L.view(y, CS, CO, y.stride(0)), None)
        L.check(lib.vv_conv_wino44(C.byref(cp), st), 'conv')
        res.append(y)
    assert torch.equal(res[0], res[1])
    y = res[0]
    assert (y[:, :, :CO] == 3.0).all() and (y[:, :, CO + Cout:] == 3.0).all()
    yr = torch.nn.functional.conv2d(x[1].view(B, H, H, Cin).permute(0, 3, 1, 2).double(), w[1].double(), padding=1).permute(0, 2, 3, 1).reshape(B * H * H, Cout)
    assert (yr - y[1, :, CO:CO + Cout].double()).abs().max().item() <= BAR * yr.abs().max().item()


def test_bank_routes_to_wino44_on_request(monkeypatch):
    """VV_WINO44=all: every 3x3 forward / data-gradient launch of a train step through vv_conv_wino44 (panels, BatchNorm partial rows,
    the fused BatchNorm-backward sums and the concat layers' bias partials all follow the kernel's tile count) -- three Adam steps
    against the oracle at the default path's bars, and the default path itself stays on F(2x2)."""
    import numpy as np
    from oracle import unet_oracle as O
    from test_gpu_unet import _build
    from vec_vad_amd.trainer import FusedTrainer
    monkeypatch.setenv('VV_WINO44', 'all')
    net, sd, tot_of = _build('net4', False)
    raw, flow = O.seeded_cubes(6, tot_of, 3)
    x, x_of = O.cubes_to_inputs(raw, flow)
    net.train()
    tr = FusedTrainer(net)
    labels = None
    opt = O.AdamState(O.param_names(sd))
    for step in range(3):
        ws = tr.step_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), torch.arange(6, device='cuda'))
        l_raw, l_of = [float(v) for v in tr.losses(ws)]
        lr_, lo_, _ = O.train_step(sd, O.bank_spec('net4'), x, x_of, opt)
        assert abs(l_raw - lr_) <= 1e-3 * lr_ and abs(l_of - lo_) <= 1e-3 * lo_, (step, l_raw, lr_, l_of, lo_)
    bank = tr.bank if hasattr(tr, 'bank') else None
    if bank is not None:
        ws = bank.workspace(6)
        labels = [c[2] for c in ws.fwd[True].calls]
        assert 'pack_wino44' in labels
        assert all(c[0] is bank.lib.vv_conv_wino44 for c in ws.fwd[True].calls if c[2].startswith('conv') and c[2][4:].isdigit())
    net.eval()
    r, o = tr.score_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda())
    rs, os_ = O.score_pass(sd, O.bank_spec('net4'), x, x_of, 6)
    np.testing.assert_allclose(r.cpu().numpy(), rs, rtol=5e-3)
    np.testing.assert_allclose(o.cpu().numpy(), os_, rtol=5e-3)


@pytest.mark.parametrize('mode', ['all', '1'])
def test_eval_scoring_through_wino44(monkeypatch, mode):
    """VV_WINO44_EVAL: the eval-mode forward on the folded model with its 3x3 launches routed to vv_conv_wino44 ('all': every one, at a
    batch the policy would not take; '1' = the default policy at a batch where it takes the 16x16 / 8x8-level launches) against the
    oracle's scores at the default path's bar and against the F(2x2) path."""
    import numpy as np
    from oracle import unet_oracle as O
    from test_gpu_unet import _build
    from vec_vad_amd.trainer import FusedTrainer
    n = 6 if mode == 'all' else 512
    res = {}
    for m in (mode, '0'):
        monkeypatch.setenv('VV_WINO44_EVAL', m)
        net, sd, tot_of = _build('net4', False)
        raw, flow = O.seeded_cubes(n, tot_of, 1)
        net.eval()
        tr = FusedTrainer(net)
        r, o = tr.score_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda())
        res[m] = (r.cpu().numpy(), o.cpu().numpy())
        routed = [l.idx for l in tr.bank.lay.convs if tr.bank._w44(n, l, False, evalm=True)]
        assert (len(routed) == 14 if m == 'all' else routed == [] if m == '0' else routed == [3, 4, 5, 8, 9, 10, 11, 12]), routed
    np.testing.assert_allclose(res[mode][0], res['0'][0], rtol=2e-4)
    np.testing.assert_allclose(res[mode][1], res['0'][1], rtol=2e-4)
    if mode == 'all':
        x, x_of = O.cubes_to_inputs(raw, flow)
        rs, os_ = O.score_pass(sd, O.bank_spec('net4'), x, x_of, n)
        np.testing.assert_allclose(res[mode][0], rs, rtol=1e-3)
        np.testing.assert_allclose(res[mode][1], os_, rtol=1e-3)


def test_wino44_first_layers_wait_for_their_side_stream_pack(monkeypatch):
    """with VV_WINO44=all the forward launches of conv0 / conv1 read F(4x4) panels that vv_pack_wino44 writes on the SIDE
    stream of the captured two-stream forward; only the third conv joined that stream.  Captured steps with the side stream stalled in
    front of each of its launches (FusedTrainer.debug_delay) must give the parameters of the one-stream order, bit for bit."""
    from oracle import unet_oracle as O
    from test_gpu_unet import _build
    from vec_vad_amd.trainer import FusedTrainer
    monkeypatch.setenv('VV_WINO44', 'all')
    raw, flow = O.seeded_cubes(12, 1, 5)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    outs = []
    for overlap, delay in (('0', None), ('free', (1, 400000))):
        monkeypatch.setenv('VV_GRAPH_OVERLAP', overlap)
        net, _, _ = _build('net4', False)
        net.train()
        tr = FusedTrainer(net)
        tr.debug_delay = delay
        for s in range(4):          # eager, capturing, two replays: every step packs the panels of freshly updated weights
            tr.step_cubes(rawd, flowd, torch.arange(12, device='cuda'))
        torch.cuda.synchronize()
        if delay:
            cap = [c for k, c in tr._graphs.items() if k[0] == 'train'][0]
            assert cap.schedule == 'free'
            labels = [(c[2], m) for c, m in zip(cap.ws.fwdq[True].calls, cap.ws.fwdq[True].meta)]
            assert dict(labels)['conv0'][1] == ('pack_wino44',) and dict(labels)['pack_wino44'][0] == 1
        outs.append(tr.bank.params.clone())
    assert torch.equal(outs[0], outs[1])
