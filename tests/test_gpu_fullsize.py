"""GPU parity at BASELINE.json's full sizes ("what's missing" #1): the tilings the bench times are checked
against the oracle, not only against themselves.

  * config 2: SelfCompleteNet4, B = 256, ONE TRAIN STEP -- losses and every parameter gradient vs ``oracle.train_step`` on the same
    256 cubes (train.py:383-402).  This is the only place the B=256 weight-gradient k-split, the XCD remap and the multi-round
    conv tilings meet an independent implementation.
  * config 4: SelfCompleteNetFull, mixed bf16, B = 512 -- train-mode loss of the whole batch and eval-mode scores vs the mixed oracle
    (model/unet.py:410-556).
  * config 5: FlowNet2 on a 1024x448 pair (1024x436 zero-padded, flownet2.py:65-149) vs the oracle; hipGraph replay bit-equal.

Tolerances are written at each assert; the oracle runs on the host cores in a few seconds per case."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grad_table(net):
    """{state_dict name: gradient tensor (view into bank.grads)} after a fused step."""
    bank = net.bank()
    by_id = {id(p): (g, key) for (g, key, p) in net._param_index}
    out = {}
    for name, p in net.named_parameters():
        g, key = by_id[id(p)]
        out[name] = bank.grad_view(g, key, shape=p.shape)
    return out


@pytest.mark.parametrize('seed', [17, 5])
def test_net4_b256_train_step_gradients_vs_oracle(seed):
    """BASELINE config 2 at full size: losses rel <= 1e-3 (observed ~1e-6) and EVERY parameter gradient against the oracle.
    Gradients of a 256-cube train-mode step are sums with heavy cancellation (BatchNorm backward removes mean and projection) over
    ReLU / max-pool gates, a few of which sit within round-off of a tie: the reference's own fp32 arithmetic (oracle fp32) is
    1.4e-3 (median) / 5.7e-3 (max) away from the oracle run in FLOAT64, per tensor, norm-wise -- so "<= 1e-3 of the fp32
    oracle" is a bar the reference itself does not meet.  The test therefore measures against the fp64 gradient and calibrates
    on the reference arithmetic in the same run: per tensor ||g_hip - g_f64|| / ||g_f64|| <= 3 x the fp32 oracle's distance
    (+2e-4 floor), median over tensors <= 1.25 x the fp32 oracle's median, and the whole gradient (all tensors as one vector)
    <= 1.5 x.  Observed on MI355X: median 1.2e-3 (HIP) vs 1.35e-3 (oracle fp32), max 4.1e-3 vs 5.7e-3.  Conv biases in front of
    BatchNorm have a mathematically zero gradient (HIP writes exact zeros, torch round-off noise)."""
    from oracle import unet_oracle as O
    from test_gpu_unet import _build
    from vec_vad_amd.trainer import FusedTrainer
    torch.set_num_threads(min(32, torch.get_num_threads()))
    net, sd, tot_of = _build('net4', False)
    B = 256
    raw, flow = O.seeded_cubes(B, tot_of, seed)           # two independent batches (round 4: one seed was one sample per tiling)
    x, x_of = O.cubes_to_inputs(raw, flow)
    net.train()
    tr = FusedTrainer(net)
    ws = tr.step_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), torch.arange(B, device='cuda'))
    l_raw, l_of = [float(v) for v in tr.losses(ws)]
    grads = {k: v.detach().cpu().double() for k, v in _grad_table(net).items()}
    ref = {}
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        sdo = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        opt = O.AdamState(O.param_names(sdo))
        ref[tag] = O.train_step(sdo, O.bank_spec('net4'), x.to(dt), x_of.to(dt), opt)
    lr_, lo_, g32 = ref['f32']
    g64 = ref['f64'][2]
    assert abs(l_raw - lr_) <= 1e-3 * lr_ and abs(l_of - lo_) <= 1e-3 * lo_, (l_raw, lr_, l_of, lo_)
    e_hip, e_ref = [], []
    n_hip = n_ref = den = 0.0
    for k, g in g64.items():
        gh = grads[k]
        if k.endswith('.0.bias') or k.endswith('.3.bias'):
            wk = k[:-4] + 'weight'
            assert float(gh.abs().max()) <= 1e-6 * float(g64[wk].abs().max()) + 1e-12, k
            continue
        nrm = float(g.norm())
        eh, er = float((gh - g).norm()) / nrm, float((g32[k].double() - g).norm()) / nrm
        assert eh <= 3 * er + 2e-4, (k, eh, er)
        e_hip.append(eh)
        e_ref.append(er)
        n_hip += float(((gh - g) ** 2).sum())
        n_ref += float(((g32[k].double() - g) ** 2).sum())
        den += float((g ** 2).sum())
    print('per-tensor gradient error vs fp64: HIP median %.2e max %.2e ; oracle fp32 median %.2e max %.2e ; whole gradient %.2e vs %.2e'
          % (np.median(e_hip), max(e_hip), np.median(e_ref), max(e_ref), (n_hip / den) ** 0.5, (n_ref / den) ** 0.5))
    assert np.median(e_hip) <= 1.25 * np.median(e_ref), (np.median(e_hip), np.median(e_ref))
    assert (n_hip / den) ** 0.5 <= 1.5 * (n_ref / den) ** 0.5, (n_hip, n_ref, den)
    # per-cube scores of the train-mode forward (batch statistics) vs the oracle's train-mode forward
    with torch.no_grad():
        sd2 = {k: v.clone() for k, v in sd.items()}
        oo, ro, ot, rt = O.bank_forward(sd2, O.bank_spec('net4'), x, x_of, True, False)
    r, o = tr.bank.cube_scores(ws)
    np.testing.assert_allclose(r.cpu().numpy(), O.cube_scores(ro, rt).numpy(), rtol=1e-3)
    np.testing.assert_allclose(o.cpu().numpy(), O.cube_scores(oo, ot).numpy(), rtol=1e-3)
    # running statistics after the step
    sdn = net.state_dict()
    for k in sd2:
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert torch.allclose(sdn[k].cpu(), sd2[k], rtol=1e-4, atol=1e-6), k


def test_full_bank_bf16_b512_vs_mixed_oracle(monkeypatch):
    """BASELINE config 4 at full size: SelfCompleteNetFull (10 UNets), mixed bf16, B = 512.
    Train-mode forward of the whole batch: loss_raw / loss_of rel <= 5e-5 of the mixed oracle on the same 512 cubes (observed
    5.5e-7 / 2.4e-6: rounding-boundary flips average out over 512 cubes); per-cube train-mode scores rel <= 2e-2 for every cube and
    rms <= 3e-3.  Eval-mode scores of a 4-cube subset rel <= 1e-4 (observed 1.7e-6).  One train step then leaves finite parameters that moved by <= 2*lr."""
    from oracle import unet_oracle as O
    from test_gpu_bf16 import _build_bf16
    from vec_vad_amd.trainer import FusedTrainer
    torch.set_num_threads(min(32, torch.get_num_threads()))
    net, sd, tot_of = _build_bf16(monkeypatch, 'full')
    assert tot_of == 5
    B = 512
    raw, flow = O.seeded_cubes(B, tot_of, 23)
    x, x_of = O.cubes_to_inputs(raw, flow)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    spec = O.bank_spec('full')
    monkeypatch.setattr(O, 'MIXED', O.MIXED_BF16)
    # eval: 4-cube subset
    net.eval()
    tr = FusedTrainer(net)
    r, o = tr.score_cubes(rawd, flowd, torch.tensor([0, 100, 301, 511], device='cuda'))
    sel = [0, 100, 301, 511]
    rs, os_ = O.score_pass({k: v.clone() for k, v in sd.items()}, spec, x[sel], x_of[sel], 4)
    from _util import observe
    observe('full_bf16_b512 eval scores vs mixed oracle', raw=float(np.max(np.abs(r.cpu().numpy() - rs) / np.abs(rs))),
            of=float(np.max(np.abs(o.cpu().numpy() - os_) / np.abs(os_))))
    np.testing.assert_allclose(r.cpu().numpy(), rs, rtol=1e-4)          # observed 1.7e-6 / 6.5e-7 (round 3); round 2's bar was 1e-2
    np.testing.assert_allclose(o.cpu().numpy(), os_, rtol=1e-4)
    # train-mode forward of the full batch
    net.train()
    p0 = tr.bank.params.clone()
    ws = tr.step_cubes(rawd, flowd, torch.arange(B, device='cuda'))
    l_raw, l_of = [float(v) for v in tr.losses(ws)]
    with torch.no_grad():
        oo, ro, ot, rt = O.bank_forward({k: v.clone() for k, v in sd.items()}, spec, x, x_of, True, False)
        _, lr_, lo_ = O.train_loss(oo, ro, ot, rt)
    lr_, lo_ = float(lr_), float(lo_)
    observe('full_bf16_b512 train losses vs mixed oracle', raw=abs(l_raw - lr_) / lr_, of=abs(l_of - lo_) / lo_)
    # observed 5.5e-7 / 2.4e-6 (round 3; rounding-boundary flips average out over 512 cubes); round 2's bar was 2e-3
    assert abs(l_raw - lr_) <= 5e-5 * lr_ and abs(l_of - lo_) <= 5e-5 * lo_, (l_raw, lr_, l_of, lo_)
    r, o = tr.bank.cube_scores(ws)
    for got, ref in ((r.cpu().numpy(), O.cube_scores(ro, rt).numpy()), (o.cpu().numpy(), O.cube_scores(oo, ot).numpy())):
        rel = np.abs(got - ref) / ref
        assert rel.max() <= 2e-2, rel.max()
        assert np.sqrt(np.mean(rel ** 2)) <= 3e-3, np.sqrt(np.mean(rel ** 2))
    p1 = tr.bank.params
    assert torch.isfinite(p1).all()
    moved = float((p1 - p0).abs().max())
    assert 5e-4 < moved <= 2.0e-3 * 1.01, moved         # Adam's first step is lr * sign(g) (|step| <= lr up to the eps term)


def _pair_1024x448():
    """one [1,3,2,448,1024] pair of uniform(0,255) float32 -- rows 436..447 are the zero padding of 1024x436."""
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 255, (1, 3, 1, 448, 1024)).astype(np.float32)
    # smooth a little and shift so that the pair carries a real displacement field (pure noise gives a degenerate correlation)
    base = (base + np.roll(base, 1, 3) + np.roll(base, 1, 4) + np.roll(np.roll(base, 1, 3), 1, 4)) / 4.0
    second = np.roll(base, (3, -5), axis=(3, 4)) + rng.normal(0, 2, base.shape).astype(np.float32)
    pair = np.clip(np.concatenate([base, second], 2), 0, 255).astype(np.float32)
    pair[:, :, :, 436:, :] = 0.0
    return torch.from_numpy(pair)


def test_flownet2_1024x448_vs_oracle():
    """BASELINE config 5 at full size.  Bar: max |flow_hip - flow_oracle| <= 1e-3 * max|flow_oracle| (same bar as the 128x192
    test); hipGraph replay (the path bench times) bit-equal to the eager launch sequence."""
    from oracle import flownet2_oracle as FO
    from test_flownet2 import _seeded_sd
    torch.set_num_threads(min(32, torch.get_num_threads()))
    net, sd, g = _seeded_sd()
    net.load_state_dict(sd)
    net = net.cuda().eval()
    inp = _pair_1024x448()
    out = net(inp.cuda()).cpu()
    assert list(out.shape) == [1, 2, 448, 1024]
    ref = FO.flownet2_forward(sd, inp)
    scale = float(ref.abs().max())
    err = float((out - ref).abs().max())
    print('flownet2 1024x448: max err %.3e, scale %.3e' % (err, scale))
    assert np.isfinite(err) and err <= 1e-3 * scale, (err, scale)
    out_g = net.forward_graphed(inp.cuda()).cpu()
    assert torch.equal(out_g, out)
    assert torch.equal(net.forward_graphed(inp.cuda()).cpu(), out)
