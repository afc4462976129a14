"""FlowNet2 native ops: the numpy oracle against closed-form known answers (CPU), and the HIP kernels against the oracle
(GPU).  The oracle for these ops is "parity unpinned" (see oracle/flow_ops_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import flow_ops_oracle as F


def test_oracle_known_answers():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((2, 8, 6, 7)).astype(np.float32)
    # correlation: zero displacement channel of corr(a, a) is mean_c a^2 ; FlowNetC geometry keeps H, W
    out = F.correlation_fwd(a, a, 4, 1, 4, 1, 2)
    assert out.shape == (2, 25, 6, 7)
    np.testing.assert_allclose(out[:, 12], (a.astype(np.float64) ** 2).mean(1), rtol=1e-6)
    # displacement (tj=+1 -> dy=+2, ti=-1 -> dx=-2): out[y,x] = mean_c a[y,x] * b[y+2, x-2] with zero padding
    b = rng.standard_normal((2, 8, 6, 7)).astype(np.float32)
    out = F.correlation_fwd(a, b, 4, 1, 4, 1, 2)
    ref = np.zeros((2, 6, 7))
    ref[:, :4, 2:] = (a[:, :, :4, 2:].astype(np.float64) * b[:, :, 2:, :5]).mean(1)
    np.testing.assert_allclose(out[:, 3 * 5 + 1], ref, rtol=1e-6, atol=1e-7)
    assert F.correlation_out_shape(256, 56, 128, 20, 1, 20, 1, 2) == (441, 56, 128)
    # resample: zero flow is the identity; integer flow shifts with edge clamping; half-pixel flow averages neighbours
    img = rng.standard_normal((2, 3, 6, 7)).astype(np.float32)
    z = np.zeros((2, 2, 6, 7), np.float32)
    np.testing.assert_array_equal(F.resample2d_fwd(img, z), img)
    fl = z.copy(); fl[:, 0] = 1.0
    sh = F.resample2d_fwd(img, fl)
    np.testing.assert_array_equal(sh[..., :-1], img[..., 1:])
    np.testing.assert_array_equal(sh[..., -1], img[..., -1])
    fl = z.copy(); fl[:, 1] = 0.5
    hv = F.resample2d_fwd(img, fl)
    np.testing.assert_allclose(hv[:, :, :-1], 0.5 * (img[:, :, :-1] + img[:, :, 1:]), rtol=1e-6, atol=1e-7)
    # channel norm
    np.testing.assert_allclose(F.channelnorm_fwd(img)[:, 0], np.sqrt((img.astype(np.float64) ** 2).sum(1)), rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,H,W,pad,md,s1,s2', [(1, 256, 12, 40, 20, 20, 1, 2), (2, 64, 9, 33, 20, 20, 1, 2),
                                                   (1, 32, 7, 16, 4, 4, 1, 2), (1, 48, 10, 70, 8, 8, 1, 1)])
def test_correlation_hip_vs_oracle(B, C, H, W, pad, md, s1, s2):
    from vec_vad_amd.flow_ops import Correlation
    rng = np.random.default_rng(1)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    ref = F.correlation_fwd(a, b, pad, 1, md, s1, s2)
    out = Correlation(pad_size=pad, kernel_size=1, max_displacement=md, stride1=s1, stride2=s2, corr_multiply=1)(
        torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,H,W,pad,k,md,s1,s2', [(1, 16, 10, 14, 5, 3, 4, 1, 2), (2, 8, 9, 11, 4, 3, 2, 1, 1), (1, 5, 12, 12, 8, 5, 4, 2, 2)])
def test_correlation_kernel_size_gt1_vs_oracle(B, C, H, W, pad, k, md, s1, s2):
    """The reference's public signature in full (correlation.py:6-27, correlation_cuda_kernel.cu:79-88): kernel_size > 1 sums the
    k x k patch around both positions.  FlowNet2 only instantiates kernel_size 1; the general form is the plain kernel."""
    from vec_vad_amd.flow_ops import Correlation
    rng = np.random.default_rng(k * 10 + C)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    ref = F.correlation_fwd(a, b, pad, k, md, s1, s2)
    out = Correlation(pad_size=pad, kernel_size=k, max_displacement=md, stride1=s1, stride2=s2, corr_multiply=1)(
        torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=2e-6)


@pytest.mark.gpu
def test_resample_and_channelnorm_hip_vs_oracle():
    from vec_vad_amd.flow_ops import Resample2d, ChannelNorm
    rng = np.random.default_rng(2)
    img = rng.uniform(0, 255, (2, 3, 64, 96)).astype(np.float32)
    flow = (rng.standard_normal((2, 2, 64, 96)) * 6).astype(np.float32)
    flow[0, :, :4] = 1000.0        # far outside: clamps to the edge pixel
    flow[1, :, -3:] = -1000.0
    out = Resample2d()(torch.from_numpy(img).cuda(), torch.from_numpy(flow).cuda()).cpu().numpy()
    np.testing.assert_array_equal(out, F.resample2d_fwd(img, flow))        # bit exact: same operation order
    for C in (2, 3, 11):
        x = rng.standard_normal((2, C, 33, 50)).astype(np.float32)
        out = ChannelNorm()(torch.from_numpy(x).cuda()).cpu().numpy()
        np.testing.assert_allclose(out, F.channelnorm_fwd(x), rtol=2e-7, atol=0)


@pytest.mark.gpu
def test_correlation_flownetc_size_properties():
    """BASELINE config 5 geometry (1024x448 pair -> conv3 features [1,256,56,128]): size-independent properties --
    corr(a, a)[centre] = mean_c a^2 and corr(a,b)[tj,ti](y,x) == corr(b,a)[-tj,-ti](y+2tj, x+2ti)."""
    from vec_vad_amd.flow_ops import correlation
    g = torch.Generator(device='cpu').manual_seed(3)
    a = torch.randn(1, 256, 56, 128, generator=g).cuda()
    b = torch.randn(1, 256, 56, 128, generator=g).cuda()
    caa = correlation(a, a, 20, 1, 20, 1, 2, 1)
    assert caa.shape == (1, 441, 56, 128)
    torch.testing.assert_close(caa[:, 220], (a * a).mean(1), rtol=1e-4, atol=1e-6)
    cab = correlation(a, b, 20, 1, 20, 1, 2, 1)
    cba = correlation(b, a, 20, 1, 20, 1, 2, 1)
    tj, ti = 3, -4
    c1 = cab[0, (tj + 10) * 21 + (ti + 10)]
    c2 = cba[0, (-tj + 10) * 21 + (-ti + 10)]
    torch.testing.assert_close(c1[:56 - 2 * tj, -2 * ti:], c2[2 * tj:, :128 + 2 * ti], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,H,W', [(1, 32, 5, 64), (2, 64, 7, 128), (1, 256, 56, 128), (1, 256, 48, 64)])
def test_correlation_nhwc_fused_vs_oracle(B, C, H, W):
    """vv_correlation_nhwc (FlowNetC's geometry, NHWC in, 1/C + LeakyReLU(0.1) fused, written into a channel slice) vs
    the numpy oracle at small sizes and vs the generic NCHW kernel at BASELINE config-5 size; untouched channels of the
    destination buffer keep their contents."""
    import ctypes as C_
    from vec_vad_amd import _lib as L
    from vec_vad_amd.flow_ops import correlation
    g = torch.Generator(device='cpu').manual_seed(5)
    a = torch.randn(B, H, W, C, generator=g).cuda()
    b = torch.randn(B, H, W, C, generator=g).cuda()
    out = torch.full((B, H, W, 476), 7.0, device='cuda')
    L.check(L.lib().vv_correlation_nhwc(a.data_ptr(), b.data_ptr(), C, B, C, H, W, out.data_ptr(), 476, 32, 0.1,
                                       torch.cuda.current_stream().cuda_stream), 'correlation_nhwc')
    assert torch.all(out[..., :32] == 7.0) and torch.all(out[..., 473:] == 7.0)
    got = out[..., 32:473].permute(0, 3, 1, 2)
    an, bn = a.permute(0, 3, 1, 2).contiguous(), b.permute(0, 3, 1, 2).contiguous()
    if H * W <= 1024:
        ref = torch.from_numpy(F.correlation_fwd(an.cpu().numpy(), bn.cpu().numpy(), 20, 1, 20, 1, 2)).cuda()
    else:
        ref = correlation(an, bn, 20, 1, 20, 1, 2, 1)
    ref = torch.where(ref > 0, ref, ref * 0.1)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-6)
    # widths outside {64, 128} are refused (the caller falls back to the generic op)
    assert L.lib().vv_correlation_nhwc(a.data_ptr(), b.data_ptr(), C, B, C, H, 96, out.data_ptr(), 476, 32, 0.1,
                                       torch.cuda.current_stream().cuda_stream) == 3
