"""FlowNet2 forward: oracle vs the golden run of the real reference graph (CPU); HIP conv stack vs torch-CPU ops and the
whole HIP FlowNet2 vs the oracle (GPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import digest, digest_close, load_golden


def _inputs(H=128, W=192):
    rng = np.random.default_rng(42)
    base = rng.uniform(0, 255, (1, 3, 1, H, W)).astype(np.float32)
    second = np.roll(base, (2, 3), axis=(3, 4)) + rng.normal(0, 2, base.shape).astype(np.float32)
    return torch.from_numpy(np.clip(np.concatenate([base, second], 2), 0, 255).astype(np.float32))


def _seeded_sd():
    from oracle import flownet2_oracle as FO
    g = load_golden('flownet2_128x192')
    from vec_vad_amd.flownet2 import FlowNet2
    net = FlowNet2()
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert [s[0] for s in shapes] == [str(s) for s in g['param_names']]          # same state_dict keys, same order
    assert [int(np.prod(s[1])) for s in shapes] == list(g['param_numel'])
    assert sum(int(np.prod(s[1])) for s in shapes) == 162518834
    return net, FO.seeded_state_dict(shapes, seed=0), g


def test_flownet2_oracle_matches_reference_graph():
    from oracle import flownet2_oracle as FO
    torch.set_num_threads(min(8, torch.get_num_threads()))
    net, sd, g = _seeded_sd()
    out = FO.flownet2_forward(sd, _inputs())
    assert list(out.shape) == list(g['out_shape'])
    np.testing.assert_allclose(out.numpy()[0, :, ::16, ::16], g['out_samples'], rtol=1e-3, atol=1e-4)
    assert digest_close(digest(out), g['out_digest'], 1e-3)


CONV_CASES = [  # (kind, R, stride, Cin, Cout, H, W, relu)
    ('conv', 7, 2, 3, 64, 64, 96, True), ('conv', 7, 2, 12, 64, 40, 72, True), ('conv', 5, 2, 64, 128, 32, 48, True),
    ('conv', 3, 1, 473, 256, 16, 24, True), ('conv', 3, 2, 256, 512, 16, 24, True), ('conv', 1, 1, 256, 32, 16, 24, True),
    ('conv', 3, 1, 1026, 2, 4, 6, False), ('conv', 3, 1, 194, 64, 32, 48, False), ('conv', 3, 1, 11, 64, 24, 40, True),
    ('conv', 3, 1, 82, 16, 20, 36, False), ('conv', 3, 1, 6, 64, 9, 33, True),
    # the few-channel first layers take the row-K form of the kernel (vv_conv2d_mfma kind 2): ragged sizes, several tiles per row
    ('conv', 7, 2, 3, 64, 37, 139, True), ('conv', 3, 1, 6, 64, 17, 70, False), ('conv', 7, 2, 3, 64, 16, 66, False),
    # predict_flow heads at the small pyramid levels: one wave per filter tap (conv3x3_n2_tap_kernel with 32 / 16 / 64 lanes per pixel)
    ('conv', 3, 1, 770, 2, 28, 20, False), ('conv', 3, 1, 386, 2, 40, 56, False), ('conv', 3, 1, 128, 2, 7, 9, False),
    ('conv', 3, 1, 194, 2, 104, 100, False), ('conv', 3, 1, 16, 2, 120, 100, False), ('conv', 3, 1, 32, 2, 101, 103, False),
    # the H/64 level runs 8 x 16 pixel tiles (maps at most 16 wide), with and without split-K
    ('conv', 3, 1, 1024, 1024, 7, 16, True), ('conv', 3, 2, 512, 1024, 14, 32, True), ('conv', 3, 1, 96, 64, 5, 13, True),
    ('deconv', 4, 2, 1024, 512, 7, 16, True),
    ('deconv', 4, 2, 1024, 512, 2, 3, True), ('deconv', 4, 2, 1026, 256, 4, 6, True), ('deconv', 4, 2, 2, 2, 4, 6, False),
    ('deconv', 4, 2, 162, 16, 12, 20, True),
]


@pytest.mark.gpu
@pytest.mark.parametrize('kind,R,stride,Cin,Cout,H,W,relu', CONV_CASES)
def test_conv2d_hip_vs_torch_cpu(kind, R, stride, Cin, Cout, H, W, relu):
    import torch.nn as nn
    from vec_vad_amd.flownet2 import _Runner, _Buf, _to_buf
    g = torch.Generator().manual_seed(R * 1000 + Cin)
    x = torch.randn(2, Cin, H, W, generator=g)
    if kind == 'conv':
        m = nn.Conv2d(Cin, Cout, R, stride=stride, padding=(R - 1) // 2)
        ref = F.conv2d(x, m.weight, m.bias, stride=stride, padding=(R - 1) // 2)
    else:
        m = nn.ConvTranspose2d(Cin, Cout, 4, 2, 1, bias=Cin != 2)
        ref = F.conv_transpose2d(x, m.weight, m.bias, stride=2, padding=1)
    if relu:
        ref = F.leaky_relu(ref, 0.1)
    ref = ref.detach()
    layer = nn.Sequential(m, nn.LeakyReLU(0.1)) if relu else m
    layer = layer.cuda()
    src = _to_buf(x.cuda())
    # write into a channel slice of a wider (concat-like) buffer to exercise coff / cstride
    dst = _Buf(2, ref.shape[2], ref.shape[3], Cout + 6, 'cuda')
    _Runner()(layer, src, dst, 4)
    out = dst.t[..., 4:4 + Cout].permute(0, 3, 1, 2).cpu()
    scale = float(ref.abs().max())
    assert torch.allclose(out, ref.detach(), rtol=0, atol=2e-5 * scale + 1e-6), float((out - ref).abs().max())
    assert float(dst.t[..., :4].abs().max()) == 0 and float(dst.t[..., 4 + Cout:].abs().max()) == 0   # neighbours untouched


# (Cin, Cout, H, W, relu): sizes that pass the runner's Winograd gate (H % 4 == 0, W % 32 == 0, >= VV_FN2_WINO_MIN_WGS workgroups at
# batch 2; H even) -- odd channel counts (K padded to 8 inside the panel; 473 = FlowNetC's conv3_1), 32 / 64 / 256 output channels, a
# one-block-wide and a one-block-high image, plain and LeakyReLU epilogues
WINO_CASES = [(64, 128, 32, 64, True), (473, 256, 8, 64, True), (162, 32, 56, 64, True), (11, 64, 28, 64, True),
              (194, 64, 16, 32, False), (128, 128, 4, 512, True), (24, 32, 60, 32, True),
              (512, 512, 14, 32, True), (40, 64, 6, 64, False)]       # H % 4 == 2: the last block's second tile row is masked


@pytest.mark.gpu
@pytest.mark.parametrize('Cin,Cout,H,W,relu', WINO_CASES)
def test_conv2d_winograd_vs_torch_cpu(Cin, Cout, H, W, relu, monkeypatch):
    """vv_conv2d_wino (round 4): FlowNet2's large stride-1 3x3 layers in Winograd F(2x2,3x3) form, against the fp32 torch CPU
    convolution -- the same 2e-5-of-the-maximum bar as the direct kernel (the transforms add a few ulp; the UNet path's Winograd
    kernel is held to the same bar against its direct form) -- and that the runner really took the Winograd path."""
    import torch.nn as nn
    from vec_vad_amd import flownet2 as FN
    monkeypatch.setattr(FN, '_WINO_MIN_WGS', 1)
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(2, Cin, H, W, generator=g)
    m = nn.Conv2d(Cin, Cout, 3, stride=1, padding=1)
    ref = F.conv2d(x, m.weight, m.bias, padding=1)
    if relu:
        ref = F.leaky_relu(ref, 0.1)
    ref = ref.detach()
    layer = (nn.Sequential(m, nn.LeakyReLU(0.1)) if relu else m).cuda()
    src = FN._to_buf(x.cuda())
    dst = FN._Buf(2, H, W, Cout + 6, 'cuda')
    run = FN._Runner()
    run(layer, src, dst, 4)
    assert ('wino', id(m)) in run.cache
    out = dst.t[..., 4:4 + Cout].permute(0, 3, 1, 2).cpu()
    scale = float(ref.abs().max())
    assert torch.allclose(out, ref, rtol=0, atol=2e-5 * scale + 1e-6), float((out - ref).abs().max())
    assert float(dst.t[..., :4].abs().max()) == 0 and float(dst.t[..., 4 + Cout:].abs().max()) == 0   # neighbours untouched


@pytest.mark.gpu
def test_upsample4_vs_torch():
    from vec_vad_amd.flownet2 import _upsample4
    x = torch.randn(2, 2, 9, 13)
    for bil, ac in ((True, False), (True, True), (False, False)):
        ref = F.interpolate(x * 1.0, scale_factor=4, mode='bilinear' if bil else 'nearest', **({'align_corners': ac} if bil else {})) * 20.0
        out = _upsample4(x.cuda(), bil, 20.0, ac).cpu()
        assert torch.allclose(out, ref, rtol=1e-5, atol=2e-5), (bil, ac, float((out - ref).abs().max()))


@pytest.mark.gpu
def test_flownet2_hip_vs_oracle_and_golden(monkeypatch):
    from oracle import flownet2_oracle as FO
    torch.set_num_threads(min(8, torch.get_num_threads()))
    net, sd, g = _seeded_sd()
    net.load_state_dict(sd)
    net = net.cuda().eval()
    inp = _inputs()
    out = net(inp.cuda()).cpu()
    ref = FO.flownet2_forward(sd, inp)
    assert list(out.shape) == list(g['out_shape'])
    scale = float(ref.abs().max())
    err = float((out - ref).abs().max())
    assert err <= 1e-3 * scale, (err, scale)
    np.testing.assert_allclose(out.numpy()[0, :, ::16, ::16], g['out_samples'], rtol=5e-3, atol=1e-3 * scale)
    out_g = net.forward_graphed(inp.cuda()).cpu()          # hipGraph replay gives the same bits
    assert torch.equal(out_g, out)
    out_g2 = net.forward_graphed(inp.cuda()).cpu()
    assert torch.equal(out_g2, out)
    # FlowNetSD runs on a second stream beside the FlowNetC -> S1 -> S2 chain: the serial schedule gives the same bits
    monkeypatch.setenv('VV_FN2_OVERLAP', '0')
    assert torch.equal(net(inp.cuda()).cpu(), out)
    monkeypatch.delenv('VV_FN2_OVERLAP')
    for at in ('0', '2'):          # ... and so does every fork point of the FlowNetSD branch (default: after FlowNetC)
        monkeypatch.setenv('VV_FN2_SD_AT', at)
        assert torch.equal(net(inp.cuda()).cpu(), out)
    monkeypatch.delenv('VV_FN2_SD_AT')
    with pytest.raises(Exception):
        net(inp)            # CPU tensor: no fallback


@pytest.mark.gpu
def test_flownet2_align_corners_true_matches_oracle():
    """nn.Upsample(bilinear) as the authors' PyTorch 0.3 computed it (align_corners=True) behind
    FlowNet2(upsample_align_corners=True): same bar against the oracle run with the same switch, and the switch must matter."""
    from oracle import flownet2_oracle as FO
    from vec_vad_amd.flownet2 import FlowNet2
    torch.set_num_threads(min(8, torch.get_num_threads()))
    _, sd, g = _seeded_sd()
    inp = _inputs()
    outs = {}
    for ac in (False, True):
        net = FlowNet2(upsample_align_corners=ac)
        net.load_state_dict(sd)
        net = net.cuda().eval()
        outs[ac] = net(inp.cuda()).cpu()
    ref = FO.flownet2_forward(sd, inp, align_corners=True)
    scale = float(ref.abs().max())
    assert float((outs[True] - ref).abs().max()) <= 1e-3 * scale
    assert float((outs[False] - ref).abs().max()) > 1e-2 * scale          # the two conventions give different flow


@pytest.mark.gpu
def test_flownet2_glue_kernels_vs_oracle_ops():
    """vv_flownet_prep / vv_warp_pack12 / vv_fusion_pack11 (flownet2.py:66-136) against the same steps written with the numpy
    op restatements: input normalisation, x4 up-sampling in all three modes, Resample2d, ChannelNorm, concat order."""
    from oracle import flow_ops_oracle as ops
    from vec_vad_amd import _lib as L
    lib = L.lib()
    B, H, W = 2, 64, 96
    rng = np.random.default_rng(3)
    inp = torch.from_numpy(rng.uniform(0, 255, (B, 3, 2, H, W)).astype(np.float32))
    st = torch.cuda.current_stream().cuda_stream
    x6 = torch.full((B, H, W, 8), 7.0, device='cuda')
    i0 = torch.full((B, H, W, 4), 7.0, device='cuda')
    i1 = torch.full((B, H, W, 4), 7.0, device='cuda')
    ws = torch.empty(int(lib.vv_flownet_prep_workspace_bytes(B)) // 4, device='cuda')
    L.check(lib.vv_flownet_prep(inp.cuda().data_ptr(), B, H, W, 255.0, ws.data_ptr(), ws.numel() * 4, x6.data_ptr(), i0.data_ptr(),
                                i1.data_ptr(), st), 'prep')
    mean = inp.contiguous().view(B, 3, -1).mean(dim=-1).view(B, 3, 1, 1, 1)
    x = (inp - mean) / 255.0
    x1, x2 = x[:, :, 0], x[:, :, 1]
    xc = torch.cat((x1, x2), 1)
    assert torch.allclose(x6.cpu()[..., :6].permute(0, 3, 1, 2), xc, rtol=0, atol=2e-6)
    assert float(x6[..., 6:].abs().max()) == 0 and float(i0[..., 3].abs().max()) == 0 and float(i1[..., 3].abs().max()) == 0
    assert torch.equal(i0[..., :3], x6[..., :3]) and torch.equal(i1[..., :3], x6[..., 3:6])
    f2 = torch.from_numpy(rng.normal(0, 0.15, (B, 2, H // 4, W // 4)).astype(np.float32))
    f2buf = torch.zeros(B, H // 4, W // 4, 4, device='cuda')
    f2buf[..., :2] = f2.permute(0, 2, 3, 1).cuda()
    x6c = x6.cpu()[..., :6].permute(0, 3, 1, 2).contiguous()          # compare against what the kernel itself read
    for mode in (0, 1, 2):
        out = torch.full((B, H, W, 12), 9.0, device='cuda')
        L.check(lib.vv_warp_pack12(x6.data_ptr(), i1.data_ptr(), f2buf.data_ptr(), 4, B, H, W, mode, 20.0, 20.0, out.data_ptr(), st), 'warp')
        kw = {} if mode == 0 else {'align_corners': mode == 2}
        flow = F.interpolate(f2 * 20.0, scale_factor=4, mode='nearest' if mode == 0 else 'bilinear', **kw)
        got = out.cpu().permute(0, 3, 1, 2)
        assert torch.allclose(got[:, 9:11] * 20.0, flow, rtol=1e-5, atol=2e-5), mode
        fl_k = (got[:, 9:11] * 20.0).contiguous()                       # warp checked on the kernel's own flow (rounding of /20*20 aside)
        warped = torch.from_numpy(ops.resample2d_fwd(x6c[:, 3:].contiguous().numpy(), flow.contiguous().numpy()))
        assert torch.allclose(got[:, 6:9], warped, rtol=0, atol=5e-4), (mode, float((got[:, 6:9] - warped).abs().max()))
        nrm = torch.from_numpy(ops.channelnorm_fwd((x6c[:, :3] - got[:, 6:9]).contiguous().numpy()))
        assert torch.allclose(got[:, 11:12], nrm, rtol=1e-5, atol=1e-6)
        assert torch.equal(got[:, :6], x6c)
    sdf = torch.from_numpy(rng.normal(0, 40.0, (B, 2, H // 4, W // 4)).astype(np.float32))
    sdbuf = torch.zeros(B, H // 4, W // 4, 4, device='cuda')
    sdbuf[..., :2] = sdf.permute(0, 2, 3, 1).cuda()
    out = torch.full((B, H, W, 12), 9.0, device='cuda')
    L.check(lib.vv_fusion_pack11(x6.data_ptr(), i1.data_ptr(), f2buf.data_ptr(), 4, sdbuf.data_ptr(), 4, B, H, W, 20.0, out.data_ptr(), st), 'fusion')
    got = out.cpu().permute(0, 3, 1, 2)
    s2 = F.interpolate(f2 * 20.0, scale_factor=4, mode='nearest')
    sdl = F.interpolate(sdf / 20.0, scale_factor=4, mode='nearest')
    assert torch.allclose(got[:, 3:5], sdl, rtol=1e-6, atol=1e-7) and torch.allclose(got[:, 5:7], s2, rtol=1e-6, atol=1e-7)
    img = x6c[:, 3:].contiguous().numpy()
    ref = torch.cat([x6c[:, :3], got[:, 3:5], got[:, 5:7],
                     torch.from_numpy(ops.channelnorm_fwd(got[:, 3:5].contiguous().numpy())),
                     torch.from_numpy(ops.channelnorm_fwd(got[:, 5:7].contiguous().numpy())),
                     torch.from_numpy(ops.channelnorm_fwd((x6c[:, :3] - torch.from_numpy(ops.resample2d_fwd(img, got[:, 3:5].contiguous().numpy()))).numpy())),
                     torch.from_numpy(ops.channelnorm_fwd((x6c[:, :3] - torch.from_numpy(ops.resample2d_fwd(img, got[:, 5:7].contiguous().numpy()))).numpy()))], 1)
    assert torch.allclose(got[:, :11], ref, rtol=1e-5, atol=2e-6), float((got[:, :11] - ref).abs().max())
    assert float(got[:, 11].abs().max()) == 0
