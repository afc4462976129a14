"""GPU: 120 fused optimisation steps (train.py:379-402's loop body, many times over) -- the only test that would see a slow drift in
the Winograd / fused-BatchNorm-sum / folded-eval / hipGraph paths, which the 3- and 6-step oracle comparisons cannot.
(Promoted from tools/longtrain_check.py)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 120


def _net(kind='net4'):
    from oracle import unet_oracle as O
    from model.unet import SelfCompleteNet4
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True,
                           padding=False)
    net.load_state_dict(O.seeded_state_dict('net4', nf=32, padding=False, seed=0))
    return net.cuda().train()


def _batches(pool, B, seed=1):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return [torch.randperm(pool, generator=g)[:B] for _ in range(STEPS)]


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_120_steps_losses_fall_and_stay_finite(monkeypatch, precision):
    """256 seeded cubes, 64 per step, through the fused engine (hipGraph replay from the third step on): every loss finite, the
    raw loss falls to less than half and the flow loss to less than 0.8 of the first step's, the trained model scores finite."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    monkeypatch.setenv('VV_PRECISION', precision)
    net = _net()
    tr = FusedTrainer(net)
    raw, flow = O.seeded_cubes(256, 1, 3)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    hist = []
    for s, idx in enumerate(_batches(256, 64)):
        ws = tr.step_cubes(rawd, flowd, idx.cuda())
        l_raw, l_of = tr.losses(ws)
        hist.append(torch.stack([l_raw, l_of]).clone())
    hist = torch.stack(hist).cpu().numpy()
    assert np.isfinite(hist).all()
    assert hist[-5:, 0].mean() < 0.5 * hist[0, 0] and hist[-5:, 1].mean() < 0.8 * hist[0, 1], (hist[0], hist[-5:].mean(0))
    assert [c for k, c in tr._graphs.items() if k[0] == 'train' and c != 'warm']          # the steps really were graph replays
    net.eval()
    r, o = tr.score_cubes(rawd, flowd, torch.arange(64, device='cuda'))
    assert torch.isfinite(r).all() and torch.isfinite(o).all() and float(r.min()) > 0.0


def _oracle_run(nthr, raw, flow, batches):
    from oracle import unet_oracle as O
    torch.set_num_threads(nthr)
    sd = O.seeded_state_dict('net4', nf=32, padding=False, seed=0)
    spec = O.bank_spec('net4')
    opt = O.AdamState(O.param_names(sd))
    losses = []
    for idx in batches:
        x, xo = O.cubes_to_inputs(raw[idx.numpy()], flow[idx.numpy()])
        losses.append(O.train_step(sd, spec, x, xo, opt)[:2])
    x, xo = O.cubes_to_inputs(raw, flow)
    rs, os_ = O.score_pass(sd, spec, x, xo, raw.shape[0])
    return np.array(losses), rs.astype(np.float64), os_.astype(np.float64)


def test_120_steps_trained_model_vs_oracle_trained_the_same_way():
    """The same 120 steps (16 cubes of 32 per step) through the fp32 oracle = the reference's arithmetic -- twice, on 32 and on 4
    threads, whose difference is what summation order alone does to a 120-step Adam trajectory -- and through the HIP path.  The
    eval-mode scores of the three trained models and the loss histories are compared: the HIP path must sit no further from the
    oracle than 4 x the oracle's own two runs sit from each other (+1e-4).  That calibration IS the bar: 120 Adam steps on small
    train-mode BatchNorm batches are chaotic -- measured on MI355X with 8 cubes per step, the oracle's two runs end 6 % apart on the
    flow scores (HIP: 13 % from the nearer one), so a fixed tolerance would be either meaningless or flaky; a drifting kernel shows as
    a ratio far above 4 (and as losses that do not fall, previous test).  With 16 cubes per step (this test), same box, three
    repeats identical: flow scores HIP 4.7e-2 vs oracle spread 7.5e-2, raw scores 5.3e-3 vs 4.3e-3, loss history 2.4e-3 vs 2.9e-3."""
    from oracle import unet_oracle as O
    from vec_vad_amd.trainer import FusedTrainer
    raw, flow = O.seeded_cubes(32, 1, 5)
    batches = _batches(32, 16, seed=2)
    net = _net()
    tr = FusedTrainer(net)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    ls = []
    for idx in batches:
        ws = tr.step_cubes(rawd, flowd, idx.cuda())
        ls.append(torch.stack(list(tr.losses(ws))).clone())
    ls = torch.stack(ls).cpu().numpy().astype(np.float64)
    net.eval()
    r, o = [t.cpu().numpy().astype(np.float64) for t in tr.score_cubes(rawd, flowd)]
    a = _oracle_run(32, raw, flow, batches)
    b = _oracle_run(4, raw, flow, batches)
    rel = lambda p, q: float(np.sqrt(((p - q) ** 2).sum() / (q ** 2).sum()))
    obs = {}
    for name, got, i in (('loss history', ls, 0), ('raw score', r, 1), ('of score', o, 2)):
        e_hip = min(rel(got, a[i]), rel(got, b[i]))
        e_ref = rel(b[i], a[i])
        obs[name] = (e_hip, e_ref)
        assert e_hip <= 0.5, (name, e_hip, e_ref)
        assert e_hip <= 4 * e_ref + 1e-4, (name, e_hip, e_ref)
    print('OBSERVED longtrain (hip vs oracle, oracle 4thr vs 32thr):', obs)
