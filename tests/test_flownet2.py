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
    torch.set_num_threads(8)
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


@pytest.mark.gpu
def test_upsample4_vs_torch():
    from vec_vad_amd.flownet2 import _upsample4
    x = torch.randn(2, 2, 9, 13)
    for bil in (True, False):
        ref = F.interpolate(x * 1.0, scale_factor=4, mode='bilinear' if bil else 'nearest', **({'align_corners': False} if bil else {})) * 20.0
        out = _upsample4(x.cuda(), bil, 20.0).cpu()
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_flownet2_hip_vs_oracle_and_golden():
    from oracle import flownet2_oracle as FO
    torch.set_num_threads(8)
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
    with pytest.raises(Exception):
        net(inp)            # CPU tensor: no fallback
