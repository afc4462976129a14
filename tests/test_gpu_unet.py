"""GPU parity tests (run with `pytest -m gpu` on an MI355X): HIP UNet bank vs the oracle restatement and vs the golden
vectors produced by the real reference.  Tolerances: per-cube scores rel <= 1e-3 (BASELINE.json north_star; observed
~1e-6), loss trajectories rel <= 1e-3."""
import numpy as np
import pytest
import torch

from _util import digest, digest_close, load_golden

pytestmark = pytest.mark.gpu

CASES = [('net4_nf32_nopad', 'net4', False, 6, None), ('net4_nf32_pad', 'net4', True, 3, None),
         ('full_nf32_nopad', 'full', False, 4, None), ('net4_nf32_rawrange4', 'net4', False, 3, 4),
         ('1raw1of_nf32_nopad', '1raw1of', False, 3, None),
         ('1raw1of_nf64_nopad', '1raw1of', False, 3, None)]      # features_root = 64: the class's default (model/unet.py:563)


def _build(kind, padding, rawRange=None, seed=0, nf=32):
    from oracle import unet_oracle as O
    from model.unet import SelfCompleteNet4, SelfCompleteNetFull, SelfCompleteNet1raw1of
    cls = {'net4': SelfCompleteNet4, 'full': SelfCompleteNetFull, '1raw1of': SelfCompleteNet1raw1of}[kind]
    tot_of = {'net4': 1, 'full': 5, '1raw1of': 1}[kind]
    net = cls(features_root=nf, tot_raw_num=5, tot_of_num=tot_of, border_mode='predict', rawRange=rawRange, useFlow=True,
              padding=padding)
    sd = O.seeded_state_dict(kind, nf=nf, padding=padding, seed=seed)
    net.load_state_dict(sd)
    return net.cuda(), sd, tot_of


def test_library_loaded_and_arch():
    from vec_vad_amd import _lib
    l = _lib.lib()
    assert l.vv_device_arch_ok() == 1, 'these kernels are built for gfx950 only'


@pytest.mark.parametrize('name,kind,padding,n,rawRange', CASES)
def test_eval_scores_and_outputs(name, kind, padding, n, rawRange):
    from oracle import unet_oracle as O
    g = load_golden(name)
    net, sd, tot_of = _build(kind, padding, rawRange, nf=64 if 'nf64' in name else 32)
    raw, flow = O.seeded_cubes(n, tot_of, 0)
    x, x_of = O.cubes_to_inputs(raw, flow)
    net.eval()
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = net(x.cuda(), x_of.cuda())
    assert list(raw_o.shape) == list(g['eval_raw_out_shape']) and list(of_o.shape) == list(g['eval_of_out_shape'])
    rs = ((raw_t - raw_o) ** 2).sum(dim=(1, 2, 3)).cpu().numpy()
    os_ = ((of_t - of_o) ** 2).sum(dim=(1, 2, 3)).cpu().numpy()
    np.testing.assert_allclose(rs, g['eval_raw_scores'], rtol=1e-3)
    np.testing.assert_allclose(os_, g['eval_of_scores'], rtol=1e-3)
    assert digest_close(digest(raw_o), g['eval_raw_out_digest'], 1e-4)
    assert digest_close(digest(of_o), g['eval_of_out_digest'], 1e-4)
    # fused score kernel == scores recomputed from the returned reconstructions
    bank = net.bank()
    r2, o2 = bank.cube_scores(bank.workspace(n))
    np.testing.assert_allclose(r2.cpu().numpy(), rs, rtol=1e-5)
    np.testing.assert_allclose(o2.cpu().numpy(), os_, rtol=1e-5)
    # oracle agrees too (same tolerance)
    spec = O.bank_spec(kind, 5, tot_of, 'predict', rawRange, True)
    with torch.no_grad():
        oo, ro, ot, rt = O.bank_forward(sd, spec, x, x_of, False, padding)
    np.testing.assert_allclose(rs, O.cube_scores(ro, rt).numpy(), rtol=1e-3)
    assert torch.allclose(raw_o.cpu(), ro, rtol=0, atol=2e-5 * float(ro.abs().max()))


@pytest.mark.parametrize('name,kind,padding,n,rawRange', CASES)
def test_three_train_steps_fused(name, kind, padding, n, rawRange):
    """train.py:376-402 with the fused path (HIP fwd/bwd + fused Adam) vs the reference's 3-step trajectory."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    g = load_golden(name)
    net, sd, tot_of = _build(kind, padding, rawRange, nf=64 if 'nf64' in name else 32)
    raw, flow = O.seeded_cubes(n, tot_of, 0)
    x, x_of = O.cubes_to_inputs(raw, flow)
    net.train()
    tr = FusedTrainer(net)
    tr.keep_outputs = True          # the reconstructions of step 0 are compared below
    xs, xo = x.cuda(), x_of.cuda()
    losses = []
    for step in range(3):
        ws = tr.step_nchw(xs, xo)
        l_raw, l_of = tr.losses(ws)
        losses.append([float(l_raw), float(l_of)])
        if step == 0:
            with torch.no_grad():
                of_o, raw_o = tr.bank.outputs_nchw(ws)
            assert digest_close(digest(raw_o), g['train_raw_out_digest'], 1e-4)
            assert digest_close(digest(of_o), g['train_of_out_digest'], 1e-4)
    np.testing.assert_allclose(np.array(losses), g['losses'], rtol=1e-3)
    sdn = net.state_dict()
    names = [str(s) for s in g['final_names']]
    bad = []
    for i, k in enumerate(names):
        if k.endswith('num_batches_tracked'):
            assert float(sdn[k]) == g['final_digests'][i][0], k
        elif not digest_close(digest(sdn[k]), g['final_digests'][i], 2e-2):
            bad.append(k)
    assert not bad, bad[:8]
    net.eval()
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = net(xs, xo)
    # post-training scores at the north-star bar (1e-3).  The reference arithmetic itself (oracle fp32 vs fp64 through the same 3
    # Adam steps) spreads ~2e-5 here and the HIP path sits at the same distance from fp64 (tools/diag_post_train.py,
    # test_post_training_distance_from_fp64_is_the_reference_arithmetics_own)
    np.testing.assert_allclose(((raw_t - raw_o) ** 2).sum(dim=(1, 2, 3)).cpu().numpy(), g['post_raw_scores'], rtol=1e-3)
    np.testing.assert_allclose(((of_t - of_o) ** 2).sum(dim=(1, 2, 3)).cpu().numpy(), g['post_of_scores'], rtol=1e-3)


def _oracle_trajectory(dt, nthr, kind, tot_of, n, steps, seed=0):
    """The oracle through `steps` train steps (train.py:383-402) + the eval-mode score pass, in dtype dt on nthr threads."""
    from oracle import unet_oracle as O
    torch.set_num_threads(nthr)
    sd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in O.seeded_state_dict(kind, nf=32, padding=False, seed=0).items()}
    raw, flow = O.seeded_cubes(n, tot_of, seed)
    x, xo = O.cubes_to_inputs(raw, flow)
    x, xo = x.to(dt), xo.to(dt)
    spec = O.bank_spec(kind)
    opt = O.AdamState(O.param_names(sd))
    losses = np.array([O.train_step(sd, spec, x, xo, opt)[:2] for _ in range(steps)])
    rs, os_ = O.score_pass(sd, spec, x, xo, n)
    return losses, rs.astype(np.float64), os_.astype(np.float64), sd


def _param_rel_l2(a, ref):
    from oracle import unet_oracle as O
    num = den = 0.0
    for k in O.param_names(ref):
        if k.endswith('.0.bias') or k.endswith('.3.bias'):
            continue            # conv bias in front of BatchNorm: zero gradient, never moves
        d = a[k].double().cpu() - ref[k].double()
        num += float((d ** 2).sum())
        den += float((ref[k].double() ** 2).sum())
    return (num / den) ** 0.5


@pytest.mark.parametrize('kind,tot_of,n,steps', [('net4', 1, 6, 3), ('net4', 1, 6, 6), ('full', 5, 4, 3), ('net4', 1, 64, 6)])
def test_post_training_distance_from_fp64_is_the_reference_arithmetics_own(kind, tot_of, n, steps):
    """the post-training tolerances, demonstrated instead of loosened.  After 3 / 6 fused train steps the
    HIP path's losses, eval-mode scores and parameters are compared with the oracle run in FLOAT64 (the exact trajectory), next
    to the distance of the oracle's own fp32 run (the reference's arithmetic) from that fp64 trajectory on 32 and on 4 threads.
    Bars: scores / losses abs-rel <= 1e-3 (north star) AND <= 4 x the reference arithmetic's own spread (+1e-5 floor); updated
    parameters (Adam's first steps are lr*sign(g): a gradient whose sign is decided by round-off flips a 2e-3 move) relative L2
    <= 2 x the fp32 oracle's.  Observed on MI355X: scores 2e-5 (3 steps) / 2e-4 (6 steps) for both, parameters 2.6e-3 vs
    4.3e-3 (3 steps), 7e-3 vs 1e-2 (6 steps)."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    net, sd, _ = _build(kind, False)
    raw, flow = O.seeded_cubes(n, tot_of, 0)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    net.train()
    tr = FusedTrainer(net)
    ls = []
    for s in range(steps):
        ws = tr.step_cubes(rawd, flowd, torch.arange(n, device='cuda'))
        ls.append([float(v) for v in tr.losses(ws)])
    net.eval()
    r, o = [t.cpu().numpy().astype(np.float64) for t in tr.score_cubes(rawd, flowd)]
    f64 = _oracle_trajectory(torch.float64, 32, kind, tot_of, n, steps)
    f32a = _oracle_trajectory(torch.float32, 32, kind, tot_of, n, steps)
    f32b = _oracle_trajectory(torch.float32, 4, kind, tot_of, n, steps)
    rel = lambda p, q: float((np.abs(p - q) / np.abs(q)).max())
    for name, got, i in (('loss', np.array(ls), 0), ('raw score', r, 1), ('of score', o, 2)):
        e_hip = rel(got, f64[i])
        e_ref = max(rel(f32a[i], f64[i]), rel(f32b[i], f64[i]))
        assert e_hip <= 1e-3, (name, e_hip)
        assert e_hip <= 4 * e_ref + 1e-5, (name, e_hip, e_ref)
    p_hip = _param_rel_l2(net.state_dict(), f64[3])
    p_ref = max(_param_rel_l2(f32a[3], f64[3]), _param_rel_l2(f32b[3], f64[3]))
    assert p_hip <= 2 * p_ref, (p_hip, p_ref)


def test_autograd_dropin_matches_oracle_grads():
    """The reference's own loop shape: model(x, x_of) -> nn.MSELoss -> backward -> torch.optim.Adam (train.py:383-402)."""
    from oracle import unet_oracle as O
    g = load_golden('net4_nf32_nopad')
    net, sd, tot_of = _build('net4', False)
    raw, flow = O.seeded_cubes(6, 1, 0)
    x, x_of = O.cubes_to_inputs(raw, flow)
    net.train()
    opt = torch.optim.Adam(net.parameters(), eps=1e-7, weight_decay=0.0)
    lf = torch.nn.MSELoss()
    losses = []
    for step in range(3):
        of_o, raw_o, of_t, raw_t = net(x.cuda(), x_of.cuda())
        l_raw, l_of = lf(raw_t.detach(), raw_o), lf(of_t.detach(), of_o)
        losses.append([l_raw.item(), l_of.item()])
        opt.zero_grad()
        (1.0 * l_raw + 1.0 * l_of).backward()
        if step == 0:
            names = [str(s) for s in g['grad_names']]
            params = dict(net.named_parameters())
            num = den = 0.0
            for i, k in enumerate(names):
                gd = g['grad_digests'][i]
                d = digest(params[k].grad)
                if k.endswith('.0.bias') or k.endswith('.3.bias'):
                    continue                      # conv bias in front of BN: mathematically zero
                num += float(np.sum((d[4:] - gd[4:]) ** 2))
                den += float(np.sum(gd[4:] ** 2))
                assert abs(d[2] - gd[2]) <= 5e-2 * gd[2] + 1e-12, (k, d[2], gd[2])   # sum of squares within 5%
            assert num <= (2e-2 ** 2) * den
        opt.step()
    np.testing.assert_allclose(np.array(losses), g['losses'], rtol=1e-3)


def test_batch_independence_and_determinism_large():
    """BASELINE-sized batch (256): eval-mode scores do not depend on batch composition, and the kernels are
    run-to-run bitwise deterministic (fixed-order reductions, no atomics)."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    net, sd, tot_of = _build('net4', False)
    net.eval()
    raw, flow = O.seeded_cubes(256, 1, 5, smooth=False)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    tr = FusedTrainer(net)
    r_all, o_all = [t.clone() for t in tr.score_cubes(rawd, flowd)]
    r_again, o_again = [t.clone() for t in tr.score_cubes(rawd, flowd)]
    assert torch.equal(r_all, r_again) and torch.equal(o_all, o_again)
    idx = torch.arange(100, 137, device='cuda')
    r_sub, o_sub = tr.score_cubes(rawd, flowd, idx)
    assert torch.allclose(r_sub, r_all[100:137], rtol=1e-5) and torch.allclose(o_sub, o_all[100:137], rtol=1e-5)
    # oracle spot check on a few cubes of the large batch
    x, x_of = O.cubes_to_inputs(raw[:4], flow[:4])
    rs, os_ = O.score_pass(sd, O.bank_spec('net4'), x, x_of, 4)
    np.testing.assert_allclose(r_all[:4].cpu().numpy(), rs, rtol=1e-3)
    np.testing.assert_allclose(o_all[:4].cpu().numpy(), os_, rtol=1e-3)
    # training determinism at B=256: two identical runs give identical parameters
    outs = []
    for rep in range(2):
        n2, _, _ = _build('net4', False)
        n2.train()
        t2 = FusedTrainer(n2)
        for s in range(2):
            t2.step_cubes(rawd, flowd, torch.arange(256, device='cuda'))
        outs.append(t2.bank.params.clone())
    assert torch.equal(outs[0], outs[1])


def test_side_stream_weight_grad_is_bitwise_identical():
    """overlap=True / 'paired' (weight gradients on a side stream, double-buffered dy) must not change a single bit."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    raw, flow = O.seeded_cubes(40, 1, 9)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    outs = []
    for overlap in (False, True, 'paired'):
        net, _, _ = _build('net4', False)
        net.train()
        tr = FusedTrainer(net, overlap=overlap)
        for s in range(3):
            tr.step_cubes(rawd, flowd, torch.arange(40, device='cuda'))
        outs.append(tr.bank.params.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize('precision,kind,B,sched', [('fp32', 'net4', 32, None), ('fp32', 'net4', 16, '0'), ('bf16', 'full', 24, None),
                                                   ('fp32', 'net4', 24, 'paired'), ('bf16', 'net4', 40, 'free')])
def test_hipgraph_step_and_scoring_bitwise_equal_to_eager(monkeypatch, precision, kind, B, sched):
    """The captured train step (cube gather + forward + backward + Adam with device-side step scalars) and the captured scoring
    pass replay bit for bit what the eager launch loop computes -- with different cubes every step (static index buffer), over
    enough steps that Adam's bias corrections matter, at the per-rank batch sizes of the reference's DataParallel split
    (train.py:375: 256 / 8 and config.cfg's 128 / 8)."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    monkeypatch.setenv('VV_PRECISION', precision)
    # the captured step's backward pass runs the weight gradients as a parallel graph branch (default 'free'; '0': one stream)
    if sched is not None:
        monkeypatch.setenv('VV_GRAPH_OVERLAP', sched)
    tot_of = 1 if kind == 'net4' else 5
    raw, flow = O.seeded_cubes(3 * B, tot_of, 21)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    g = torch.Generator().manual_seed(5)
    perms = [torch.randperm(3 * B, generator=g)[:B].cuda() for _ in range(7)]
    outs = []
    for graph in (False, True):
        net, _, _ = _build(kind, False)
        net.train()
        tr = FusedTrainer(net)
        tr.use_graph = graph
        losses = []
        for p in perms:
            ws = tr.step_cubes(rawd, flowd, p)
            losses.append(torch.stack([x for x in tr.losses(ws) if x is not None]).clone())
        cap = [c for k, c in tr._graphs.items() if k[0] == 'train' and c != 'warm']
        assert (len(cap) == 1 and len(cap[0].segments) == 1 and cap[0].launches >= 95) if graph else not cap
        assert tr.bank.adam_t == 7 and int(tr.bank._adam_t_dev.item()) == 7
        net.eval()
        sc = []
        for k in range(4):          # eager, capturing, two replays -- on different cubes, then the default "first B cubes" form
            r, o = tr.score_cubes(rawd, flowd, perms[k])
            sc.append(torch.cat([r, o]).clone())
        r, o = tr.score_cubes(rawd, flowd, None, B)
        sc.append(torch.cat([r, o]).clone())
        outs.append((tr.bank.params.clone(), tr.bank.bufs.clone(), tr.bank.nbt.clone(), torch.stack(losses), torch.stack(sc)))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[0][3]).all() and int(outs[0][2].min()) == 7


def test_bn_backward_sums_fused_into_data_gradient(monkeypatch):
    """The first reduction pass of a layer's BatchNorm backward (sum dz, sum dz * xhat) runs in the epilogue of the Winograd
    data-gradient launch that produces dA (vv_conv_params.bn_partial, VV_BNBWD_PARTIALS_PER_TILE) where that launch is the only
    producer; VV_FUSE_BN_SUMS=0 keeps the separate vv_bn_bwd_reduce pass.  Same sums in another order: one step's gradients agree
    to fp32 round-off (measured 2.2e-6 of a tensor's norm at worst, bar 1e-5); ragged batch (partial pixel tiles on the 8x8 / 4x4 levels)."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    raw, flow = O.seeded_cubes(37, 1, 11)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    grads = []
    for fuse in ('0', '1'):
        monkeypatch.setenv('VV_FUSE_BN_SUMS', fuse)
        net, _, _ = _build('net4', False)
        net.train()
        tr = FusedTrainer(net)
        assert tr.bank.fuse_bn_sums == (fuse == '1')
        tr.step_cubes(rawd, flowd, torch.arange(37, device='cuda'))
        torch.cuda.synchronize()
        grads.append(tr.bank.grads_gu())
    lay = tr.bank.lay
    worst = 0.0
    for key, (off, shape) in lay.p.items():
        n = int(torch.tensor(shape).prod())
        a, b = grads[0][:, off:off + n].double(), grads[1][:, off:off + n].double()
        worst = max(worst, ((a - b).norm() / a.norm().clamp_min(1e-30)).item())
    assert 0.0 < worst <= 1e-5, worst          # > 0: the two paths really are different code


def test_bn_backward_sums_fused_into_transposed_conv_data_gradient(monkeypatch):
    """Round 6: layers 7 / 9 / 11 feed only a transposed conv, so their dA is that transposed conv's data gradient (vv_conv_mfma,
    VV_CONVT_DGRAD) -- its epilogue leaves sum dz, sum dz * xhat per 128-pixel tile (VV_BNBWD_PARTIALS_PER_TTILE) and the three
    vv_bn_bwd_reduce passes go.  VV_FUSE_BN_SUMS_T=0 keeps them.  Same sums in another order: gradients agree to fp32 round-off
    (bar 1e-5 of a tensor's norm, the Winograd fusion's bar); bit-identical upstream of the first fused launch; ragged batches (B = 37:
    partial tiles on the 8x8 / 4x4 levels, B = 3: a single partial tile)."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    for B in (37, 3):
        raw, flow = O.seeded_cubes(B, 1, 11)
        rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
        grads, nred = [], []
        for fuse in ('0', '1'):
            monkeypatch.setenv('VV_FUSE_BN_SUMS_T', fuse)
            net, _, _ = _build('net4', False)
            net.train()
            tr = FusedTrainer(net)
            tr.step_cubes(rawd, flowd, torch.arange(B, device='cuda'))
            torch.cuda.synchronize()
            grads.append(tr.bank.grads_gu())
            ws = tr.bank.workspace(B)
            nred.append([c[2] for c in tr.bank.backward_plan(ws, fused=tr.bank.fuse_outconv).calls if c[2].startswith('bn_bwd_reduce')])
        assert nred[0] == ['bn_bwd_reduce11', 'bn_bwd_reduce9', 'bn_bwd_reduce7', 'bn_bwd_reduce5', 'bn_bwd_reduce3', 'bn_bwd_reduce1']
        assert nred[1] == ['bn_bwd_reduce5', 'bn_bwd_reduce3', 'bn_bwd_reduce1']
        lay = tr.bank.lay
        worst = 0.0
        for key, (off, shape) in lay.p.items():
            n = int(torch.tensor(shape).prod())
            a, b = grads[0][:, off:off + n].double(), grads[1][:, off:off + n].double()
            d = ((a - b).norm() / a.norm().clamp_min(1e-30)).item()
            if key.split('.')[0] in ('c12', 'c13', 'o', 't2'):
                assert d == 0.0, (key, d)            # computed before the first fused launch (dgradT2) has any effect
            worst = max(worst, d)
        assert 0.0 < worst <= 1e-5, (B, worst)



@pytest.mark.parametrize('kind', ['net4', 'full'])
def test_bn_backward_sums_fused_into_bf16_data_gradient(monkeypatch, kind):
    """Round 4: the same fusion for the all-bf16 bank on the 32x32 level (vv_conv_mfma's 32-wide data-gradient launches leave
    sum dz, sum dz * xhat per 256-pixel tile: VV_BNBWD_PARTIALS_PER_CTILE) -- layers 0 and 12 lose their vv_bn_bwd_reduce pass.
    The sums are over the stored (bf16) gradient in both paths, in another order: layer 12's own gradients (the first consumer of
    fused sums in the backward order) move by fp32 round-off (measured 1.4e-6 of the tensor's norm, bar 1e-5).  From there on a
    different c1 / c2 flips the bf16 rounding of a few stored dy elements, and every further layer of the backward chain rounds
    its dy and data gradient to bf16 again: the difference grows to a few 1e-3 of a tensor's norm at layer 0 (measured 7.5e-3 at
    worst) -- the step-to-step noise floor of bf16 gradients, the same effect as the 2e-3 script bar of test_gpu_scripts.py.
    Bar 2e-2 for those; the training-level bf16 bars (losses over steps against the mixed oracle, test_gpu_bf16.py) are unaffected."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    monkeypatch.setenv('VV_PRECISION', 'bf16')
    _, _, tot_of = _build(kind, False)
    raw, flow = O.seeded_cubes(21, tot_of, 11)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    grads, nred = [], []
    for fuse in ('0', '1'):
        monkeypatch.setenv('VV_FUSE_BN_SUMS', fuse)
        net, _, _ = _build(kind, False)
        net.train()
        tr = FusedTrainer(net)
        tr.step_cubes(rawd, flowd, torch.arange(21, device='cuda'))
        torch.cuda.synchronize()
        grads.append(tr.bank.grads_gu())
        ws = tr.bank.workspace(21)
        nred.append(sum(1 for c in tr.bank.backward_plan(ws, fused=tr.bank.fuse_outconv).calls if c[2].startswith('bn_bwd_reduce')))
    assert nred[0] - nred[1] == 2, nred
    lay = tr.bank.lay
    worst = first = 0.0
    for key, (off, shape) in lay.p.items():
        n = int(torch.tensor(shape).prod())
        a, b = grads[0][:, off:off + n].double(), grads[1][:, off:off + n].double()
        d = ((a - b).norm() / a.norm().clamp_min(1e-30)).item()
        if key.startswith('c13.') or key.startswith('o.'):
            assert d == 0.0, (key, d)              # upstream of the first fused launch
        elif key.startswith('c12.'):
            first = max(first, d)
        else:
            worst = max(worst, d)
    assert 0.0 < first <= 1e-5, first
    assert worst <= 2e-2, worst


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_two_stream_schedules_have_every_dependency(monkeypatch, precision):
    """The two-stream schedules again, with one stream stalled before each of its launches (FusedTrainer.debug_delay): the other
    stream runs far ahead, so a cross-stream dependency that is only satisfied by luck (e.g. a side-stream weight gradient reading
    a data gradient the main stream has not written yet) changes the result.  Both directions, both schedules."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    monkeypatch.setenv('VV_PRECISION', precision)
    raw, flow = O.seeded_cubes(24, 1, 4)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    outs = []
    for overlap, delay in ((False, None), (True, (0, 200000)), (True, (1, 200000)), ('paired', (0, 200000)), ('paired', (1, 200000))):
        net, _, _ = _build('net4', False)
        net.train()
        tr = FusedTrainer(net, overlap=overlap)
        tr.debug_delay = delay
        for s in range(2):
            tr.step_cubes(rawd, flowd, torch.arange(24, device='cuda'))
        outs.append(tr.bank.params.clone())
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


def test_odd_batch_sizes_train():
    """ragged batches (last partial batch is kept, train.py:373 drop_last=False): B = 1, 2, 7, 17."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    for B in (1, 2, 7, 17):
        net, sd, tot_of = _build('net4', False)
        raw, flow = O.seeded_cubes(B, 1, 11)
        x, x_of = O.cubes_to_inputs(raw, flow)
        net.train()
        tr = FusedTrainer(net)
        ws = tr.step_nchw(x.cuda(), x_of.cuda())
        l_raw, l_of = tr.losses(ws)
        opt = O.AdamState(O.param_names(sd))
        lr, lo, _ = O.train_step(sd, O.bank_spec('net4'), x, x_of, opt)
        assert abs(float(l_raw) - lr) <= 1e-3 * lr and abs(float(l_of) - lo) <= 1e-3 * lo
    # B = 1 in eval mode (test.py scores frames with a single cube)
    net, sd, tot_of = _build('net4', False)
    net.eval()
    raw, flow = O.seeded_cubes(1, 1, 12)
    x, x_of = O.cubes_to_inputs(raw, flow)
    tr = FusedTrainer(net)
    r, o = tr.score_nchw(x.cuda(), x_of.cuda())
    rs, os_ = O.score_pass(sd, O.bank_spec('net4'), x, x_of, 1)
    np.testing.assert_allclose(r.cpu().numpy(), rs, rtol=1e-3)
    np.testing.assert_allclose(o.cpu().numpy(), os_, rtol=1e-3)


def test_state_dict_roundtrip_and_move():
    net, sd, _ = _build('net4', False)
    sd2 = net.state_dict()
    for k in sd:
        assert torch.equal(sd2[k].cpu(), sd[k]), k
    net.cpu()
    assert all(torch.equal(net.state_dict()[k], sd[k]) for k in sd)
    with pytest.raises(Exception):
        net(torch.zeros(1, 15, 32, 32), torch.zeros(1, 2, 32, 32))      # no CPU fallback


@pytest.mark.parametrize('H,Cin,Cout,B', [(32, 16, 32, 3), (32, 32, 32, 2), (16, 64, 64, 5), (8, 256, 128, 9), (4, 128, 256, 33),
                                           (16, 32, 64, 1)])
def test_winograd_conv_matches_direct_conv(H, Cin, Cout, B):
    """vv_conv_wino (Winograd F(2x2,3x3) on the matrix cores) against vv_conv_mfma (direct implicit GEMM) on the same
    tensors, forward panel and data-gradient panel, BatchNorm+ReLU-on-load input, bias and BatchNorm partial sums:
    agreement to fp32 round-off of the transforms (ragged last workgroup: B is not a multiple of the images per tile)."""
    import ctypes as C
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 1000 + Cin)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()          # nn.Conv2d layout per group
    bias = torch.randn(G, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    st = torch.cuda.current_stream().cuda_stream
    U = Cout * Cin * 9

    def pack(fn, mode, K, N, taps):
        ent = (L.PackEntry * 1)(L.PackEntry(0, 0, mode, K, K, N))
        tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
        out = torch.zeros(G, taps * K * N, device='cuda')
        L.check(fn(tab.data_ptr(), 1, G, w.data_ptr(), U, out.data_ptr(), out.stride(0), (9 if taps == 9 else 1) * K * N, st), 'pack')
        return out

    for dgrad in (False, True):
        K, N = (Cout, Cin) if dgrad else (Cin, Cout)
        if N % 32:
            continue
        src = torch.randn(G, B * H * H, K, generator=g).cuda() if dgrad else x
        pd, pw = pack(lib.vv_pack_weights, 1 if dgrad else 0, K, N, 9), pack(lib.vv_pack_wino, 1 if dgrad else 0, K, N, 16)
        outs, stats = [], []
        for fn, pk, nt in ((lib.vv_conv_mfma, pd, lib.vv_conv_ntiles(B, H, H)), (lib.vv_conv_wino, pw, lib.vv_wino_ntiles(B, H))):
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
        assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * scale, (dgrad, (outs[0] - outs[1]).abs().max().item(), scale)
        if not dgrad:
            torch.testing.assert_close(stats[0], stats[1], rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize('H,C0,C1,Cout,B', [(32, 32, 32, 32, 3), (8, 128, 128, 128, 5), (4, 128, 128, 256, 9)])
def test_winograd_conv_concat_input_and_fused_bn_sums(H, C0, C1, Cout, B):
    """Two more launch forms of vv_conv_wino at kernel level: (1) a concat input (VV_IN_CAT: channels [0, C0) = BatchNorm+ReLU of
    src0, [C0, C0+C1) = src1 as it is -- the decoder's first conv, model/unet.py:57-60) against the direct kernel; (2) a data-gradient
    launch that also leaves the BatchNorm-backward partial sums of its consumer (vv_conv_params.bn_partial): sum dz and sum dz * xhat
    with dz = dA [a z + b > 0] against a float64 evaluation on the launch's own output."""
    import ctypes as C
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
    U = Cout * Cin * 9

    def pack(fn, mode, K, N, taps):
        ent = (L.PackEntry * 1)(L.PackEntry(0, 0, mode, K, K, N))
        tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
        out = torch.zeros(G, taps * K * N, device='cuda')
        L.check(fn(tab.data_ptr(), 1, G, w.data_ptr(), U, out.data_ptr(), out.stride(0), (9 if taps == 9 else 1) * K * N, st), 'pack')
        return out

    # (1) concat input, forward panel
    outs = []
    for fn, pk in ((lib.vv_conv_mfma, pack(lib.vv_pack_weights, 0, Cin, Cout, 9)), (lib.vv_conv_wino, pack(lib.vv_pack_wino, 0, Cin, Cout, 16))):
        y = torch.full((G, B * H * H, Cout), 3.0, device='cuda')
        cp = L.ConvParams(L.CONV3, L.IN_CAT, G, B, H, H, Cin, Cin, Cout, L.view(x0, C0, 0, x0.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                          L.view(x1, C1, 0, x1.stride(0)), C0, 0, None, pk.data_ptr(), pk.stride(0), None, 0, L.view(y, Cout, 0, y.stride(0)), None)
        L.check(fn(C.byref(cp), st), 'conv cat')
        outs.append(y)
    scale = outs[0].abs().max().item()
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * scale
    ref = torch.cat([torch.relu(x0 * a[:, None, :C0] + b[:, None, :C0]), x1], 2)      # what the kernel must have convolved
    yr = torch.nn.functional.conv2d(ref[0].view(B, H, H, Cin).permute(0, 3, 1, 2).double(), w[0].double(), padding=1)
    assert (yr.permute(0, 2, 3, 1).reshape(B * H * H, Cout) - outs[1][0].double()).abs().max().item() <= 1e-4 * scale

    # (2) data gradient Cout -> Cin of the same filter with the fused BatchNorm-backward sums of the Cin-channel producer
    dy = torch.randn(G, B * H * H, Cout, generator=g).cuda()
    z = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    mean = (torch.randn(G, Cin, generator=g) * 0.1).cuda()
    invstd = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    pk = pack(lib.vv_pack_wino, 1, Cout, Cin, 16)
    nt = lib.vv_wino_ntiles(B, H)
    dA = torch.zeros(G, B * H * H, Cin, device='cuda')
    part = torch.full((G, nt, 2, Cin), 7.0, device='cuda')
    cp = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cout, Cout, Cin, L.view(dy, Cout, 0, dy.stride(0)), None, None, 0, L.NULL_VIEW, 0, 0,
                      None, pk.data_ptr(), pk.stride(0), None, 0, L.view(dA, Cin, 0, dA.stride(0)), None,
                      z.data_ptr(), z.stride(0), a.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(), Cin, part.data_ptr())
    L.check(lib.vv_conv_wino(C.byref(cp), st), 'dgrad + bn sums')
    bare = torch.zeros_like(dA)
    cp2 = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cout, Cout, Cin, L.view(dy, Cout, 0, dy.stride(0)), None, None, 0, L.NULL_VIEW, 0, 0,
                       None, pk.data_ptr(), pk.stride(0), None, 0, L.view(bare, Cin, 0, bare.stride(0)), None)
    L.check(lib.vv_conv_wino(C.byref(cp2), st), 'dgrad')
    assert torch.equal(dA, bare)                                   # the fused sums do not touch the output
    d = dA.double() * ((a[:, None] * z + b[:, None]) > 0)
    xh = (z.double() - mean[:, None].double()) * invstd[:, None].double()
    s = part.double().sum(1)
    torch.testing.assert_close(s[:, 0], d.sum(1), rtol=1e-4, atol=1e-3 * d.abs().sum(1).max().item() / (B * H * H) ** 0.5)
    torch.testing.assert_close(s[:, 1], (d * xh).sum(1), rtol=1e-4, atol=1e-3 * (d * xh).abs().sum(1).max().item() / (B * H * H) ** 0.5)
    # a stats pointer and bn_partial together are refused
    cp.stats = part.data_ptr()
    assert lib.vv_conv_wino(C.byref(cp), st) != 0


def test_winograd_kernels_random_sweep():
    """Seeded sweep over launch shapes the parametrised tests above do not enumerate: vv_conv_wino (forward / data-gradient panel,
    plain / BatchNorm+ReLU / concat input, every level, ragged batches) against vv_conv_mfma, and vv_wgrad_mfma's Winograd form
    against the direct form for k-splits from 1 to more workgroups than pixel tiles (workgroups without a tile write a zero slab)."""
    import ctypes as C
    import random
    from vec_vad_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    rnd = random.Random(1234)
    G = 2

    def pack(fn, w, mode, K, N, taps):
        ent = (L.PackEntry * 1)(L.PackEntry(0, 0, mode, K, K, N))
        tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
        out = torch.zeros(G, taps * K * N, device='cuda')
        L.check(fn(tab.data_ptr(), 1, G, w.data_ptr(), w[0].numel(), out.data_ptr(), out.stride(0), (9 if taps == 9 else 1) * K * N, st), 'pack')
        return out

    for case in range(20):
        H = rnd.choice([32, 16, 8, 4])
        Cin, Cout = rnd.choice([16, 32, 64, 96, 128]), rnd.choice([32, 64, 128])
        B = rnd.randint(1, 11 if H >= 16 else 37)
        mode = rnd.choice(['plain', 'act', 'cat'])
        dgrad = mode == 'plain' and rnd.random() < 0.5
        if mode == 'cat' and Cin < 32:
            Cin = 64
        gen = torch.Generator(device='cpu').manual_seed(case)
        K, N = (Cout, Cin) if dgrad else (Cin, Cout)
        if N % 32:
            N = 32
        w = (torch.randn(G, (K if dgrad else N), (N if dgrad else K), 3, 3, generator=gen) * 0.1).cuda()   # [Cout][Cin] of the forward conv
        a = (torch.rand(G, K, generator=gen) + 0.5).cuda()
        b = (torch.randn(G, K, generator=gen) * 0.2).cuda()
        c0 = rnd.choice([16, 32]) if mode == 'cat' else K
        c0 = min(c0, K - 16) // 16 * 16 if mode == 'cat' else K
        x0 = torch.randn(G, B * H * H, c0, generator=gen).cuda()
        x1 = torch.randn(G, B * H * H, max(K - c0, 1), generator=gen).cuda()
        outs = []
        for fn, pk, nt in ((lib.vv_conv_mfma, pack(lib.vv_pack_weights, w, 1 if dgrad else 0, K, N, 9), lib.vv_conv_ntiles(B, H, H)),
                           (lib.vv_conv_wino, pack(lib.vv_pack_wino, w, 1 if dgrad else 0, K, N, 16), lib.vv_wino_ntiles(B, H))):
            y = torch.full((G, B * H * H, N), 5.0, device='cuda')
            s_ = torch.zeros(G, nt, 2, N, device='cuda')
            im = {'plain': L.IN_PLAIN, 'act': L.IN_ACT, 'cat': L.IN_CAT}[mode]
            cp = L.ConvParams(L.CONV3, im, G, B, H, H, K, K, N, L.view(x0, c0, 0, x0.stride(0)),
                              None if mode == 'plain' else a.data_ptr(), None if mode == 'plain' else b.data_ptr(), K,
                              L.view(x1, K - c0, 0, x1.stride(0)) if mode == 'cat' else L.NULL_VIEW, c0 if mode == 'cat' else 0, 0, None,
                              pk.data_ptr(), pk.stride(0), None, 0, L.view(y, N, 0, y.stride(0)), s_.data_ptr())
            L.check(fn(C.byref(cp), st), 'conv')
            outs.append((y, s_.sum(1)))
        scale = outs[0][0].abs().max().item()
        assert (outs[0][0] - outs[1][0]).abs().max().item() <= 2e-5 * scale, (case, H, K, N, B, mode, dgrad)
        torch.testing.assert_close(outs[0][1], outs[1][1], rtol=3e-4, atol=3e-3 * max(1.0, scale))

    for case in range(12):
        H = rnd.choice([32, 16, 8, 4])
        Cin, Cout = rnd.choice([16, 32, 64]), rnd.choice([32, 64])
        B = rnd.randint(1, 9 if H >= 16 else 21)
        gen = torch.Generator(device='cpu').manual_seed(100 + case)
        x = torch.randn(G, B * H * H, Cin, generator=gen).cuda()
        dy = torch.randn(G, B * H * H, Cout, generator=gen).cuda()
        a = (torch.rand(G, Cin, generator=gen) + 0.5).cuda()
        b = (torch.randn(G, Cin, generator=gen) * 0.2).cuda()
        nci, nco = (Cin + 31) // 32, Cout // 32
        nt = lib.vv_wgrad_ntiles(L.CONV3, B, H, H)
        ks = rnd.choice([1, 2, 3, 7, nt, nt + 3])
        res = []
        for flag in (0, 256):
            part = torch.full((G, nci * nco * ks * 9 * 1024), 9.0, device='cuda')
            grad = torch.zeros(G, Cout * Cin * 9, device='cuda')
            wp = L.WgradParams(L.CONV3, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, ks, L.view(x, Cin, 0, x.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                               L.NULL_VIEW, 0, flag, None, L.View(dy.data_ptr(), dy.stride(0), Cout, 0), part.data_ptr(), part.stride(0))
            L.check(lib.vv_wgrad_mfma(C.byref(wp), st), 'wgrad')
            L.check(lib.vv_wgrad_reduce(L.CONV3, G, Cin, Cin, Cout, ks, part.data_ptr(), part.stride(0), grad.data_ptr(), grad.stride(0), st), 'reduce')
            res.append(grad)
        scale = res[0].abs().max().item()
        assert (res[0] - res[1]).abs().max().item() <= 5e-5 * scale, (case, H, Cin, Cout, B, ks)


def test_full_bank_b512_linearity_and_determinism():
    """BASELINE config 4 size (SelfCompleteNetFull = 10 UNets, B = 512, fp32): size-independent properties of the whole
    backward pass -- it is linear in d(loss)/d(out) (doubling dout doubles every gradient bit for bit: scaling by 2 is exact
    in fp32 and the reductions run in a fixed order), it is run-to-run deterministic, and the loss / per-cube scores of the
    forward agree with the oracle on a few cubes of the batch."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    net, sd, tot_of = _build('full', False)
    raw, flow = O.seeded_cubes(512, tot_of, 3, smooth=False)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    # eval-mode scores of the first cubes vs the oracle (batch independent in eval mode; before any train-mode forward moves
    # the BatchNorm running statistics)
    net.eval()
    r, o = FusedTrainer(net, reset_optimizer=False).score_cubes(rawd, flowd)
    x, x_of = O.cubes_to_inputs(raw[:3], flow[:3])
    rs, os_ = O.score_pass(sd, O.bank_spec('full'), x, x_of, 3)
    np.testing.assert_allclose(r[:3].cpu().numpy(), rs, rtol=1e-3)
    np.testing.assert_allclose(o[:3].cpu().numpy(), os_, rtol=1e-3)
    net.train()
    bank = net.bank()
    ws = bank.set_input_cubes(rawd, flowd, None, 512)
    bank.forward(ws, True)
    bank.backward(ws)
    g1 = bank.grads.clone()
    bank.backward(ws)
    assert torch.equal(bank.grads, g1)                      # deterministic
    ws.dout4.mul_(2.0)
    bank.backward(ws)
    g2 = bank.grads.clone()
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert torch.equal(g2, 2.0 * g1)                        # linear, bit for bit


@pytest.mark.parametrize('H,Cin,Cout,B', [(32, 32, 32, 3), (16, 64, 64, 5), (8, 256, 128, 9), (4, 128, 256, 33), (32, 16, 32, 2)])
def test_winograd_weight_gradient_matches_direct(H, Cin, Cout, B):
    """vv_wgrad_mfma with pad0 bit 8 (Winograd F(2x2,3x3) weight gradient: dU = sum_tiles V^T dM, dg = G^T dU G) against the
    direct tap-by-tap kernel on the same activation / gradient tensors (BatchNorm+ReLU-on-load input, ragged batch), through
    the same slab reduction into the nn.Conv2d weight layout, and against a float64 evaluation on a sample of entries."""
    import ctypes as C
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G = 2
    g = torch.Generator(device='cpu').manual_seed(H * 100 + Cin)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    dy = torch.randn(G, B * H * H, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    st = torch.cuda.current_stream().cuda_stream
    nci, nco = (Cin + 31) // 32, Cout // 32
    nt = lib.vv_wgrad_ntiles(L.CONV3, B, H, H)
    ks = max(1, min(nt, 3))
    outs = []
    for flag in (0, 256):
        part = torch.zeros(G, nci * nco * ks * 9 * 1024, device='cuda')
        grad = torch.zeros(G, Cout * Cin * 9, device='cuda')
        wp = L.WgradParams(L.CONV3, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, ks, L.view(x, Cin, 0, x.stride(0)), a.data_ptr(), b.data_ptr(),
                           Cin, L.NULL_VIEW, 0, flag, None, L.View(dy.data_ptr(), dy.stride(0), Cout, 0), part.data_ptr(), part.stride(0))
        L.check(lib.vv_wgrad_mfma(C.byref(wp), st), 'wgrad')
        L.check(lib.vv_wgrad_reduce(L.CONV3, G, Cin, Cin, Cout, ks, part.data_ptr(), part.stride(0), grad.data_ptr(), grad.stride(0),
                                    st), 'reduce')
        outs.append(grad.view(G, Cout, Cin, 3, 3).clone())
    scale = outs[0].abs().max().item()
    for o in outs[1:]:
        assert (outs[0] - o).abs().max().item() <= 5e-5 * scale, ((outs[0] - o).abs().max().item(), scale)
    # float64 spot check: dW[co,ci,ky,kx] = sum_{b,y,x} act[b,y+ky-1,x+kx-1,ci] * dy[b,y,x,co]
    act = torch.relu(x.double() * a.double()[:, None, :] + b.double()[:, None, :]).view(G, B, H, H, Cin)
    dyd = dy.double().view(G, B, H, H, Cout)
    pad = torch.nn.functional.pad(act, (0, 0, 1, 1, 1, 1))
    for (gi, co, ci, ky, kx) in ((0, 0, 0, 0, 0), (1, Cout - 1, Cin - 1, 2, 1), (0, 5, 7, 1, 1), (1, 17, 3, 0, 2)):
        ref = (pad[gi, :, ky:ky + H, kx:kx + H, ci] * dyd[gi, :, :, :, co]).sum().item()
        assert abs(outs[1][gi, co, ci, ky, kx].item() - ref) <= 2e-4 * scale + 1e-4 * abs(ref), (gi, co, ci, ky, kx)


def test_eval_fold_matches_unfolded_path_and_tracks_updates(monkeypatch):
    """Eval mode (test.py:255-257,312-345) runs on the folded model -- BatchNorm folded into filter + bias once per model state
    (vv_fold_bn), ReLU in the conv epilogue (VV_CONV_RELU), plain loads -- instead of the train-mode kernel family with running
    statistics (VV_EVAL_FOLD=0).  Both paths against each other (fp32 re-association of a*(W x): rel <= 2e-5 on per-cube scores) and
    the folded model must follow every state change: a train step (raw-pointer writes by Adam / BatchNorm), load_state_dict, and
    an in-place edit of a running statistic through the module's buffer."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    raw, flow = O.seeded_cubes(37, 1, 4)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    x, x_of = O.cubes_to_inputs(raw, flow)

    def scores(fold, mutate):
        monkeypatch.setenv('VV_EVAL_FOLD', '1' if fold else '0')
        net, sd, _ = _build('net4', False)
        assert net.bank().eval_fold == fold
        tr = FusedTrainer(net)
        out = []
        net.eval()
        out.append([t.clone() for t in tr.score_cubes(rawd, flowd)])
        net.train()
        tr.step_cubes(rawd, flowd, torch.arange(37, device='cuda'))            # Adam + running statistics move
        net.eval()
        out.append([t.clone() for t in tr.score_cubes(rawd, flowd)])
        with torch.no_grad():
            net.inc0.conv.conv[1].running_var.mul_(1.5)                             # in-place edit through the module
        out.append([t.clone() for t in tr.score_cubes(rawd, flowd)])
        net.load_state_dict(sd)                                                     # back to the seeded model
        out.append([t.clone() for t in tr.score_cubes(rawd, flowd)])
        return out, sd

    a, sd = scores(True, True)
    b, _ = scores(False, True)
    for sa, sb in zip(a, b):
        for ta, tb in zip(sa, sb):
            torch.testing.assert_close(ta, tb, rtol=2e-5, atol=0)
    assert not torch.allclose(a[0][0], a[1][0], rtol=1e-4)      # the train step changed the scores
    assert not torch.allclose(a[1][0], a[2][0], rtol=1e-6)      # ... so did the edited running variance
    torch.testing.assert_close(a[3][0], a[0][0], rtol=0, atol=0)  # ... and load_state_dict restores them bit for bit
    rs, os_ = O.score_pass(sd, O.bank_spec('net4'), x, x_of, 37)
    np.testing.assert_allclose(a[0][0].cpu().numpy(), rs, rtol=1e-3)
    np.testing.assert_allclose(a[0][1].cpu().numpy(), os_, rtol=1e-3)


def test_useflow_false_matches_oracle():
    """useFlow=False (config.cfg `useFlow`, model/unet.py:74,161,244-267; train.py:389-392: loss = loss_raw alone, no lambda):
    the bank has the five raw UNets only, forward returns empty flow lists like the reference, a fused train step and the
    eval-mode scores agree with the oracle."""
    from oracle import unet_oracle as O
    from model.unet import SelfCompleteNet4
    from vec_vad_amd.trainer import FusedTrainer
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=False,
                           padding=False)
    sd = O.seeded_state_dict('net4', nf=32, useFlow=False, padding=False, seed=0)
    net.load_state_dict(sd)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())          # no *_of modules are registered
    net = net.cuda()
    raw, flow = O.seeded_cubes(5, 1, 2)
    x, x_of = O.cubes_to_inputs(raw, flow)
    spec = O.bank_spec('net4', 5, 1, 'predict', None, False)
    net.eval()
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = net(x.cuda(), x_of.cuda())
    assert of_o == [] and of_t == [] and list(raw_o.shape) == [5, 15, 32, 32]
    rs, _ = O.score_pass(sd, spec, x, x_of, 5, useFlow=False)
    np.testing.assert_allclose(((raw_t - raw_o) ** 2).sum(dim=(1, 2, 3)).cpu().numpy(), rs, rtol=1e-3)
    net.train()
    tr = FusedTrainer(net, lambda_raw=0.5)          # lambda_raw must NOT enter the loss without flow (train.py:391-392)
    ws = tr.step_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), torch.arange(5, device='cuda'))
    l_raw, l_of = tr.losses(ws)
    assert l_of is None
    sdo = {k: v.clone() for k, v in sd.items()}
    opt = O.AdamState(O.param_names(sdo))
    lr_, _, gref = O.train_step(sdo, spec, x, x_of, opt, useFlow=False)
    assert abs(float(l_raw) - lr_) <= 1e-3 * lr_
    net.eval()
    r, o = tr.score_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda())
    assert o is None
    rs2, _ = O.score_pass(sdo, spec, x, x_of, 5, useFlow=False)
    np.testing.assert_allclose(r.cpu().numpy(), rs2, rtol=1e-3)


@pytest.mark.parametrize('precision,kind,nf,B', [('fp32', 'net4', 32, 6), ('bf16', 'full', 32, 5), ('fp32', '1raw1of', 64, 3)])
def test_fused_outconv_forward_backward_bitwise_equal_to_two_launches(monkeypatch, precision, kind, nf, B):
    """vv_outconv_fwdbwd (round 4): the 1x1 output conv's forward and backward in one pass over y -- what the fused train step runs --
    leaves exactly the bits of vv_outconv_fwd + vv_outconv_bwd: per-cube scores, every parameter gradient of the bank (the
    output conv's through its partials, all others through dA and the BatchNorm-backward partial sums)."""
    monkeypatch.setenv('VV_PRECISION', precision)
    from oracle import unet_oracle as O
    res = []
    for fuse in ('1', '0'):
        monkeypatch.setenv('VV_FUSE_OUTCONV', fuse)
        net, sd, tot_of = _build(kind, False, nf=nf)
        net.train()
        bank = net.bank()
        assert bank.fuse_outconv == (fuse == '1')
        raw, flow = O.seeded_cubes(B, tot_of, 5)
        ws = bank.set_input_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), torch.arange(B, device='cuda'))
        bank.forward(ws, True, outputs=False)
        assert getattr(ws.fwdq[True], 'fused_outconv', False) == (fuse == '1')
        bank.backward(ws, fused=True)
        labels = [c[2] for c in bank.backward_plan(ws, fused=bank.fuse_outconv).calls]
        assert ('outconv_bwd' in labels) == (fuse == '0')
        torch.cuda.synchronize()
        res.append((ws.score.clone(), bank.grads.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    assert float(res[0][1].abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('kind,nf,B', [('net4', 32, 5), ('full', 32, 3)])
def test_bf16_concat_gradient_in_two_planes_bitwise_equal(monkeypatch, kind, nf, B):
    """vv_conv_params.out1 in the train step (round 4): with all-bf16 tensors the data gradient of the three concat layers leaves as
    two dense planes (skip half -> BatchNorm backward of the encoder layer, upsampled half -> transposed-conv backward).  A pure
    layout change: every parameter gradient of the bank keeps its bits."""
    monkeypatch.setenv('VV_PRECISION', 'bf16')
    from oracle import unet_oracle as O
    res = []
    for split in ('1', '0'):
        monkeypatch.setenv('VV_SPLIT_DCAT', split)
        net, sd, tot_of = _build(kind, False, nf=nf)
        net.train()
        bank = net.bank()
        assert bank.split_dcat == (split == '1')
        raw, flow = O.seeded_cubes(B, tot_of, 5)
        ws = bank.set_input_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), torch.arange(B, device='cuda'))
        bank.forward(ws, True, outputs=False)
        bank.backward(ws, fused=True)
        torch.cuda.synchronize()
        res.append((ws.score.clone(), bank.grads.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    assert float(res[0][1].abs().max()) > 0



# ---- features_root other than 32 / 64: the model runs EMBEDDED in the next engine width (vec_vad_amd/unet.py engine_width) ----
@pytest.mark.parametrize('kind,nf,n', [('net4', 4, 6), ('net4', 16, 6), ('full', 8, 4), ('1raw1of', 48, 3), ('net4', 20, 5)])
def test_embedded_width_matches_oracle_eval_train_and_state(kind, nf, n):
    """model/unet.py:74,271,563 take any features_root.  Zero-padded into the 32- / 64-wide engine, the extra channels must hold
    exact zeros forward, receive exact-zero gradients and stay at zero through Adam, so that the module IS the reference's model:
    eval scores and outputs vs the oracle at the model's own width (1e-3 / 2e-5 of max), three fused train steps (losses 1e-3,
    parameters like test_three_train_steps_fused), the autograd drop-in's gradients, and the zero block checked bit for bit."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    from vec_vad_amd.unet import engine_width, _embed_pieces
    net, sd, tot_of = _build(kind, False, nf=nf)
    assert net._embedded and net._engine_nf == engine_width(nf) and net._engine_nf in (32, 64)
    raw, flow = O.seeded_cubes(n, tot_of, 0)
    x, x_of = O.cubes_to_inputs(raw, flow)
    xs, xo = x.cuda(), x_of.cuda()
    spec = O.bank_spec(kind)
    net.eval()
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = net(xs, xo)
        oo, ro, ot, rt = O.bank_forward(sd, spec, x, x_of, False, False)
    np.testing.assert_allclose(((raw_t - raw_o) ** 2).sum(dim=(1, 2, 3)).cpu().numpy(), O.cube_scores(ro, rt).numpy(), rtol=1e-3)
    np.testing.assert_allclose(((of_t - of_o) ** 2).sum(dim=(1, 2, 3)).cpu().numpy(), O.cube_scores(oo, ot).numpy(), rtol=1e-3)
    assert torch.allclose(raw_o.cpu(), ro, rtol=0, atol=2e-5 * float(ro.abs().max()))
    # autograd drop-in: gradients in the module's own shapes vs the oracle's
    net.train()
    sd_g = {k: v.clone() for k, v in sd.items()}
    _, _, g_ref = O.train_step(sd_g, spec, x, x_of, O.AdamState(O.param_names(sd_g)))
    of_o, raw_o, of_t, raw_t = net(xs, xo)
    lf = torch.nn.MSELoss()
    (lf(raw_t.detach(), raw_o) + lf(of_t.detach(), of_o)).backward()
    num = den = 0.0
    for k, p in net.named_parameters():
        if k.endswith('.0.bias') or k.endswith('.3.bias'):
            continue
        assert p.grad.shape == g_ref[k].shape
        num += float(((p.grad.cpu().double() - g_ref[k].double()) ** 2).sum())
        den += float((g_ref[k].double() ** 2).sum())
    # (4 - 6 cubes in train-mode BatchNorm + max-pool: a tie decided by round-off moves a gradient entry; observed 1e-4 .. 2.2e-3,
    #  the 2- and 3-cube gradients of tests/test_gpu_ddp.py sit at 2e-2)
    assert (num / den) ** 0.5 < 1e-2, (num / den) ** 0.5
    # fused trainer, three steps from the seeded state (the module was not stepped: reload it, which also exercises the push path)
    net.load_state_dict(sd)
    net.zero_grad()
    tr = FusedTrainer(net)
    sd_o = {k: v.clone() for k, v in sd.items()}
    opt = O.AdamState(O.param_names(sd_o))
    for step in range(3):
        ws = tr.step_nchw(xs, xo)
        l_hip = [float(v) for v in tr.losses(ws)]
        l_ref = O.train_step(sd_o, spec, x, x_of, opt)[:2]
        np.testing.assert_allclose(l_hip, l_ref, rtol=1e-3)
    # a direct read through parameters() must see the trained values too, not the copies from before the steps
    k0, p0 = next(iter(net.named_parameters()))
    first = p0.detach().cpu().clone()
    assert not torch.equal(first, sd[k0]), 'parameters() returned the stale pre-training copy'
    got = net.state_dict()                  # pulls the trained blocks out of the engine's tensors
    assert torch.equal(first, got[k0].cpu())
    assert _param_rel_l2(got, sd_o) < 2e-2
    for k, v in got.items():
        assert v.shape == sd_o[k].shape, k
        if k.endswith('num_batches_tracked'):
            assert int(v) == int(sd_o[k]) == 3          # load_state_dict reset the counter; three fused steps since
        elif k.endswith('running_mean') or k.endswith('running_var'):
            # (three Adam steps of lr * sign(g) apart in a few round-off-decided weights, seen through the batch statistics of 6
            #  cubes -- 96 samples on the 4x4 level: a norm-wise bar like the golden test's digests of the final state)
            d = (v.cpu().double() - sd_o[k].double()).norm() / sd_o[k].double().norm()
            assert float(d) < 5e-2, (k, float(d))
    # the padding of the engine's tensors is still exactly zero (parameters, Adam moments, running statistics)
    bank = net.bank()
    mask = torch.ones_like(bank.params, dtype=torch.bool)
    for (g, key, p) in net._param_index:
        off, shape = bank.lay.p[key]
        m = mask[g, off:off + int(np.prod(shape))].view(shape)
        for mi, bi in _embed_pieces(bank.lay, key, tuple(p.shape)):
            m[bi] = False
    assert int(mask.sum()) > 0
    assert float(bank.params[mask].abs().max()) == 0.0
    assert float(bank.adam_m[mask].abs().max()) == 0.0 and float(bank.adam_v[mask].abs().max()) == 0.0
    # ... and the eval-mode scores after training agree with the oracle's on ITS trained weights
    net.eval()
    r, o = [t.cpu().numpy() for t in tr.score_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda())]
    rs, os_ = O.score_pass(sd_o, spec, x, x_of, n)
    np.testing.assert_allclose(r, rs, rtol=1e-3)
    np.testing.assert_allclose(o, os_, rtol=1e-3)


@pytest.mark.parametrize('B,Cin,Cout,mode,bnf,relu', [(64, 32, 32, 'act', False, False), (43, 32, 32, 'plain', True, False),
                                                      (43, 16, 32, 'plain', False, False), (64, 16, 32, 'act', False, True),
                                                      (37, 32, 64, 'plain', False, False), (64, 32, 64, 'act', True, False)])
def test_wino_ring_bitwise_equal(B, Cin, Cout, mode, bnf, relu):
    """vv_conv_wino's persistent LDS-DMA ring kernel (32x32 level, K <= 32: csrc/vv_wino.hip wino_ring_kernel) against the per-tile
    kernel (VV_CONV_NO_RING) on the same tensors: output, BatchNorm partial sums and the fused BatchNorm-backward sums must agree
    BIT FOR BIT (same chunk -> MFMA order, same epilogue arithmetic), including runs of tiles that cross a UNet / N-tile boundary
    in the middle of a workgroup's run (B = 43 / 37: 5 tiles per workgroup) and the folded eval path's ReLU epilogue."""
    import ctypes as C
    from vec_vad_amd import _lib as L
    lib = L.lib()
    G, H = 6, 32
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cpu').manual_seed(B * 100 + Cin + Cout)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    bias = torch.randn(G, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    z = torch.randn(G, B * H * H, Cout, generator=g).cuda()
    bn = [(torch.rand(G, Cout, generator=g) + 0.5).cuda() for _ in range(4)]
    ent = (L.PackEntry * 1)(L.PackEntry(0, 0, 0, Cin, Cin, Cout))
    tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
    pk = torch.zeros(G, 16 * Cin * Cout, device='cuda')
    L.check(lib.vv_pack_wino(tab.data_ptr(), 1, G, w.data_ptr(), w[0].numel(), pk.data_ptr(), pk.stride(0), Cin * Cout, st), 'pack')
    nt = lib.vv_wino_ntiles(B, H)
    assert G * (Cout // 32) * nt >= 4 * 512                # enough tiles for the ring kernel to take the launch
    outs = []
    for flag in (0, L.CONV_NO_RING):
        y = torch.full((G, B * H * H, Cout), float('nan'), device='cuda')
        s_ = torch.full((G, nt, 2, Cout), float('nan'), device='cuda')
        cp = L.ConvParams(L.CONV3, L.IN_ACT if mode == 'act' else L.IN_PLAIN, G, B, H, H, Cin, Cin, Cout, L.view(x, Cin, 0, x.stride(0)),
                          a.data_ptr(), b.data_ptr(), Cin, L.NULL_VIEW, 0, flag | (L.CONV_RELU if relu else 0), None, pk.data_ptr(),
                          pk.stride(0), bias.data_ptr(), Cout, L.view(y, Cout, 0, y.stride(0)), None if bnf else s_.data_ptr())
        if bnf:
            cp.bn_z, cp.bn_z_gstride = z.data_ptr(), z.stride(0)
            cp.bn_a, cp.bn_b, cp.bn_mean, cp.bn_invstd = (t.data_ptr() for t in bn)
            cp.bn_gstride, cp.bn_partial = Cout, s_.data_ptr()
        L.check(lib.vv_conv_wino(C.byref(cp), st), 'conv')
        torch.cuda.synchronize()
        outs.append((y, s_))
    assert not torch.isnan(outs[0][0]).any() and not torch.isnan(outs[0][1]).any()
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    if relu:
        assert float(outs[0][0].min()) == 0.0


@pytest.mark.parametrize('precision,kind,nf,B', [('fp32', 'net4', 32, 5), ('fp32', '1raw1of', 64, 3), ('bf16', 'full', 32, 3)])
def test_grouped_slab_reduction_bitwise_equal_to_per_layer_kernel(monkeypatch, precision, kind, nf, B):
    """vv_wgrad_reduce_grouped (round 5: one workgroup per 32 x 32 filter tile, nine taps transposed through LDS, full-line stores)
    against the per-layer, one-element-per-thread vv_wgrad_reduce (VV_GROUP_REDUCE=0): the same slabs summed in the same order --
    every weight gradient of the bank (3x3 convs incl. the 12-channel first layer, transposed convs) keeps its bits."""
    monkeypatch.setenv('VV_PRECISION', precision)
    from oracle import unet_oracle as O
    res = []
    for grouped in ('1', '0'):
        monkeypatch.setenv('VV_GROUP_REDUCE', grouped)
        net, sd, tot_of = _build(kind, False, nf=nf)
        net.train()
        bank = net.bank()
        raw, flow = O.seeded_cubes(B, tot_of, 7)
        ws = bank.set_input_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), torch.arange(B, device='cuda'))
        bank.forward(ws, True, outputs=False)
        bank.backward(ws)
        labels = [c[2] for c in bank.backward_plan(ws, fused=bank.fuse_outconv).calls]
        assert any(l.startswith('wgradT_reduce0') for l in labels)
        assert (sum(1 for l in labels if 'reduce' in l and l.startswith('wgrad')) <= 3) == (grouped == '1')
        torch.cuda.synchronize()
        res.append(bank.grads.clone())
    assert torch.equal(res[0], res[1])
    assert float(res[0].abs().max()) > 0


def test_wino_two_n_tiles_per_workgroup_bitwise_equal():
    """wino_conv_kernel<H, 2> (round 5: 64 output channels per workgroup where the grid stays in whole rounds) against the one-N-tile form
    (VV_WINO_NB=1, read once per process: a child process computes the reference digest) on launch shapes the policy takes -- forward
    with BatchNorm+ReLU on load, bias, BatchNorm partial sums; data gradient with the fused BatchNorm-backward sums: bit for bit."""
    import os
    import subprocess
    import sys
    code = r'''
import ctypes as C, hashlib, sys, torch
sys.path.insert(0, %r)
from vec_vad_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
h = hashlib.sha256()
for (H, Cin, Cout, B, G, bnf) in ((8, 64, 128, 512, 2, False), (16, 64, 64, 256, 2, False), (8, 128, 128, 512, 2, True)):
    g = torch.Generator(device='cpu').manual_seed(H * 131 + Cout)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    bias = torch.randn(G, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    ent = (L.PackEntry * 1)(L.PackEntry(0, 0, 1 if bnf else 0, Cin, Cin, Cout))
    tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
    wsrc = w.transpose(1, 2).contiguous() if bnf else w          # mode 1 reads W[co = k][ci = n]
    pk = torch.zeros(G, 16 * Cin * Cout, device='cuda')
    L.check(lib.vv_pack_wino(tab.data_ptr(), 1, G, wsrc.data_ptr(), wsrc[0].numel(), pk.data_ptr(), pk.stride(0), Cin * Cout, st), 'pack')
    nt = lib.vv_wino_ntiles(B, H)
    y = torch.zeros(G, B * H * H, Cout, device='cuda')
    s_ = torch.zeros(G, nt, 2, Cout, device='cuda')
    if bnf:
        z = torch.randn(G, B * H * H, Cout, generator=g).cuda()
        a2 = (torch.rand(G, Cout, generator=g) + 0.5).cuda(); b2 = (torch.randn(G, Cout, generator=g) * 0.2).cuda()
        mean = (torch.randn(G, Cout, generator=g) * 0.1).cuda(); inv = (torch.rand(G, Cout, generator=g) + 0.5).cuda()
        cp = L.ConvParams(L.CONV3, L.IN_PLAIN, G, B, H, H, Cin, Cin, Cout, L.view(x, Cin, 0, x.stride(0)), None, None, 0, L.NULL_VIEW, 0, 0,
                          None, pk.data_ptr(), pk.stride(0), None, 0, L.view(y, Cout, 0, y.stride(0)), None,
                          z.data_ptr(), z.stride(0), a2.data_ptr(), b2.data_ptr(), mean.data_ptr(), inv.data_ptr(), Cout, s_.data_ptr())
    else:
        cp = L.ConvParams(L.CONV3, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, L.view(x, Cin, 0, x.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                          L.NULL_VIEW, 0, 0, None, pk.data_ptr(), pk.stride(0), bias.data_ptr(), Cout, L.view(y, Cout, 0, y.stride(0)), s_.data_ptr())
    L.check(lib.vv_conv_wino(C.byref(cp), st), 'conv')
    torch.cuda.synchronize()
    h.update(y.cpu().numpy().tobytes()); h.update(s_.cpu().numpy().tobytes())
    assert float(y.abs().max()) > 0
print('DIGEST', h.hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for nb in ('0', '1'):
        env = dict(os.environ, VV_WINO_NB=nb)
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([l for l in out.stdout.splitlines() if l.startswith('DIGEST')][-1])
    assert digests[0] == digests[1]
