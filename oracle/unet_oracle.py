"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of VEC_VAD's cube-completion path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``vec_vad_amd``) never imports it and fails loudly when the HIP library is
missing.

What is restated (citations are into the reference tree, ``/root/reference``):

* ``double_conv / inconv / down / up / outconv``                 model/unet.py:4-70
* ``SelfCompleteNet4.forward``                                   model/unet.py:172-267
* ``SelfCompleteNetFull.forward``                                model/unet.py:410-556
* ``SelfCompleteNet1raw1of.forward``                             model/unet.py:619-652
* training loss  ``lambda_raw*MSE(raw)+lambda_of*MSE(of)``       train.py:241,385-392
* per-cube score ``MSELoss(reduce=False)`` summed over (C,H,W)   train.py:414,421-426 ; test.py:330-335
* ``Adam(eps=1e-7, weight_decay=0)``, torch defaults lr=1e-3     train.py:376,400-402
* z-normalisation with training statistics, w_raw/w_of mix       test.py:260-266,338-345
* frame score = max over painted bbox mask, empty frame = -1e5   test.py:276,350-357,387-389
* frame-level ROC-AUC                                            utils.py:29-39

The arithmetic (conv / batch-norm / pooling) lives in PyTorch in the reference as well (torch 1.1 + cuDNN
there, torch CPU / oneDNN here); this restatement calls the *functional* torch CPU ops in fp32 (or fp64 on
request) on tensors taken from a reference-named ``state_dict``, so it shares no module code with either
the reference or the product.

Pinning: ``tests/golden/make_goldens.py`` imports the real reference modules in the authoring container
and stores their outputs on formula-seeded weights/inputs as small fixtures; ``tests/test_oracle_golden.py``
checks this restatement against those fixtures.  The reference has no tests of its own.

The ``MIXED`` switch below (bf16 operand rounding, BASELINE config 4) has no counterpart in the reference -- it computes in
fp32 only -- so that mode is PARITY UNPINNED by construction: it restates torch.autocast(bfloat16)'s treatment of the
convolutions and is held in place by ``tests/test_oracle_golden.py`` (with every rounding switched off the custom autograd
function must reproduce the plain functional ops and their autograd gradients exactly).  With ``MIXED = None`` nothing of it runs.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

RAW_C = 3   # model/unet.py:92
OF_C = 2    # model/unet.py:93


# ----------------------------------------------------------------------------------------------------
# Bank description: which UNets exist, which frame each erases, what it predicts.
# ----------------------------------------------------------------------------------------------------
def bank_spec(kind, tot_raw_num=5, tot_of_num=None, border_mode='predict', rawRange=None, useFlow=True):
    """List of UNet descriptors in the order the reference appends outputs.

    kind: 'net4' (model/unet.py:73), 'full' (:270), '1raw1of' (:559).
    Each entry: dict(role='raw'|'of', stems={inc,down1..3,up1..3,outc}, erase=frame index removed from the
    input, tgt=index of the predicted raw frame / flow map, out_c=3|2).
    """
    if tot_of_num is None:
        tot_of_num = {'net4': 1, 'full': 5, '1raw1of': 1}[kind]
    assert tot_of_num <= tot_raw_num
    if border_mode == 'predict' or (kind == 'full' and border_mode == 'elasticPredict'):
        raw_center, of_center = tot_raw_num - 1, tot_of_num - 1
    else:
        raw_center, of_center = (tot_raw_num - 1) // 2, (tot_of_num - 1) // 2
    offset = raw_center - of_center
    assert offset >= 0
    if rawRange is None:
        rr = list(range(tot_raw_num))
    else:
        if rawRange < 0:
            rawRange += tot_raw_num
        assert rawRange < tot_raw_num
        rr = [rawRange]

    def stems(tag):
        if kind == '1raw1of':
            if tag == 'raw':
                return dict(inc='inc', down=['down1', 'down2', 'down3'], up=['up1', 'up2', 'up3'], outc='outc')
            return dict(inc='inc_of', down=['down_of1', 'down_of2', 'down_of3'],
                        up=['up_of1', 'up_of2', 'up_of3'], outc='outc_of')
        role, i = tag
        if role == 'raw':
            return dict(inc='inc%d' % i, down=['down%d%d' % (i, k) for k in (1, 2, 3)],
                        up=['up%d%d' % (i, k) for k in (1, 2, 3)], outc='outc%d' % i)
        if kind == 'net4':
            return dict(inc='inc_of', down=['down_of%d' % k for k in (1, 2, 3)],
                        up=['up_of%d' % k for k in (1, 2, 3)], outc='outc_of')
        return dict(inc='inc_of%d' % i, down=['down_of%d%d' % (i, k) for k in (1, 2, 3)],
                    up=['up_of%d%d' % (i, k) for k in (1, 2, 3)], outc='outc_of%d' % i)

    out = []
    if kind == '1raw1of':
        e = tot_raw_num - 1
        out.append(dict(role='raw', stems=stems('raw'), erase=e, tgt=e, out_c=RAW_C))
        if useFlow:
            out.append(dict(role='of', stems=stems('of'), erase=e, tgt=tot_raw_num - 1 - offset, out_c=OF_C))
        return out
    for raw_i in rr:
        out.append(dict(role='raw', stems=stems(('raw', raw_i)), erase=raw_i, tgt=raw_i, out_c=RAW_C))
        of_i = raw_i - offset
        if useFlow and 0 <= of_i < tot_of_num:
            out.append(dict(role='of', stems=stems(('of', of_i)), erase=raw_i, tgt=of_i, out_c=OF_C))
    return out


def all_unet_stems(kind, tot_raw_num=5, tot_of_num=None, useFlow=True):
    """Every UNet that owns parameters (independent of rawRange), in module registration order is not needed
    here -- only the set of stems."""
    spec = bank_spec(kind, tot_raw_num, tot_of_num, 'predict', None, useFlow)
    seen, out = set(), []
    for u in spec:
        if u['stems']['inc'] not in seen:
            seen.add(u['stems']['inc'])
            out.append(u)
    return out


# ----------------------------------------------------------------------------------------------------
# Formula-seeded parameters (numpy PCG64 keyed by parameter name -> identical on every box).
# ----------------------------------------------------------------------------------------------------
def unet_param_shapes(stems, in_ch, nf, out_c):
    """(name, shape, kind) for one UNet in state_dict naming (model/unet.py:4-70)."""
    ents = []

    def dc(prefix, cin, cout):
        for (ci, bi, a, b) in ((0, 1, cin, cout), (3, 4, cout, cout)):
            ents.append(('%s.%d.weight' % (prefix, ci), (b, a, 3, 3), 'conv_w'))
            ents.append(('%s.%d.bias' % (prefix, ci), (b,), 'conv_b'))
            ents.append(('%s.%d.weight' % (prefix, bi), (b,), 'bn_w'))
            ents.append(('%s.%d.bias' % (prefix, bi), (b,), 'bn_b'))
            ents.append(('%s.%d.running_mean' % (prefix, bi), (b,), 'bn_rm'))
            ents.append(('%s.%d.running_var' % (prefix, bi), (b,), 'bn_rv'))
            ents.append(('%s.%d.num_batches_tracked' % (prefix, bi), (), 'bn_nbt'))

    dc(stems['inc'] + '.conv.conv', in_ch, nf)
    c = nf
    for d in stems['down']:
        dc(d + '.mpconv.1.conv', c, 2 * c)
        c *= 2
    for u in stems['up']:
        ents.append((u + '.up.weight', (c, c // 2, 3, 3), 'convT_w'))
        ents.append((u + '.up.bias', (c // 2,), 'convT_b'))
        dc(u + '.conv.conv', c, c // 2)
        c //= 2
    ents.append((stems['outc'] + '.conv.weight', (out_c, nf, 1, 1), 'conv_w'))
    ents.append((stems['outc'] + '.conv.bias', (out_c,), 'conv_b'))
    return ents


def seeded_state_dict(kind, nf=32, tot_raw_num=5, tot_of_num=None, useFlow=True, padding=False, seed=0,
                      dtype=torch.float32):
    """Deterministic, box-independent state_dict with non-trivial BN parameters and running statistics."""
    in_ch = RAW_C * (tot_raw_num if padding else tot_raw_num - 1)
    sd = OrderedDict()
    for u in all_unet_stems(kind, tot_raw_num, tot_of_num, useFlow):
        for name, shape, k in unet_param_shapes(u['stems'], in_ch, nf, u['out_c']):
            rng = np.random.default_rng(zlib.crc32(name.encode()) + 7919 * seed)
            if k == 'conv_w':
                bound = 1.0 / np.sqrt(shape[1] * shape[2] * shape[3])
                v = rng.uniform(-bound, bound, shape)
            elif k == 'convT_w':
                bound = 1.0 / np.sqrt(shape[1] * shape[2] * shape[3])
                v = rng.uniform(-bound, bound, shape)
            elif k in ('conv_b', 'convT_b'):
                v = rng.uniform(-0.05, 0.05, shape)
            elif k == 'bn_w':
                v = rng.uniform(0.5, 1.5, shape)
            elif k == 'bn_b':
                v = rng.uniform(-0.2, 0.2, shape)
            elif k == 'bn_rm':
                v = rng.uniform(-0.1, 0.1, shape)
            elif k == 'bn_rv':
                v = rng.uniform(0.5, 1.5, shape)
            else:
                sd[name] = torch.zeros((), dtype=torch.long)
                continue
            sd[name] = torch.from_numpy(np.asarray(v, dtype=np.float64)).to(dtype)
    return sd


def seeded_cubes(n, tot_of_num=1, seed=0, smooth=True):
    """Synthetic cubes in the reference's on-disk layout (train.py:218-222): raw uint8 [N,5,32,32,3],
    flow float32 [N,T_of,32,32,2]."""
    rng = np.random.default_rng(1000 + seed)
    raw = rng.integers(0, 256, (n, 5, 32, 32, 3)).astype(np.float32)
    if smooth:  # cheap box blur so that BN statistics look image-like
        raw = (raw + np.roll(raw, 1, 2) + np.roll(raw, 1, 3) + np.roll(np.roll(raw, 1, 2), 1, 3)) / 4.0
    raw = np.clip(np.rint(raw), 0, 255).astype(np.uint8)
    flow = (rng.standard_normal((n, tot_of_num, 32, 32, 2)) * 2.0).astype(np.float32)
    return raw, flow


def cubes_to_inputs(raw, flow):
    """vad_datasets.py:130-168 (cube_to_train_dataset.__getitem__ + ToTensor + default collate):
    [N,T,H,W,C] -> transpose(1,2,0,3) -> [H,W,T*C] -> CHW, uint8 -> float/255, float32 unchanged."""
    n, t, h, w, c = raw.shape
    x = np.transpose(raw, (0, 2, 3, 1, 4)).reshape(n, h, w, t * c)
    x = np.transpose(x, (0, 3, 1, 2))
    if raw.dtype == np.uint8:
        x = x.astype(np.float32) / 255.0
    if flow.ndim == 4:
        flow = flow[:, None]
    n, tf, h, w, cf = flow.shape
    f = np.transpose(flow, (0, 2, 3, 1, 4)).reshape(n, h, w, tf * cf)
    f = np.transpose(f, (0, 3, 1, 2)).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(f))


# ----------------------------------------------------------------------------------------------------
# Mixed precision (BASELINE config 4, "mixed bf16").  The reference itself is fp32 only; config 4 asks for the
# torch.autocast(bfloat16) treatment of the convolutions: operands rounded to bf16 (nearest even), products accumulated
# in fp32.  MIXED = None is the reference arithmetic.  MIXED = {...} restates what the HIP path computes per operation,
# so that the bf16 path can be checked operation by operation instead of only through AUROC:
#   'fwd'      3x3 conv and transposed conv forward:  conv(bf16(x), bf16(w)) + b, fp32 result
#   'dgrad'    3x3 conv data gradient:                conv_input_grad(bf16(dy), bf16(w))
#   'dgradT'   transposed-conv data gradient          (fp32 operands while False)
#   'wgrad'    3x3 conv weight gradient:              corr(bf16(x), bf16(dy)) for maps of at least 'wgrad_min_hw' pixels a side,
#              fp32 operands otherwise; 'wgradT' the same for the transposed conv
#   'dx16'     activation gradients are STORED as bf16: the data gradients returned by the 3x3 conv / transposed conv and the
#              gradient the 1x1 output conv hands to the last double_conv are rounded (torch.autocast's dtype for them)
#   'y16'      the outputs of the 3x3 conv / transposed conv (bias added) are STORED as bf16 (torch.autocast's output dtype);
#              BatchNorm statistics are those of the stored tensor
# Bias, BatchNorm, max-pool, the 1x1 output conv, the loss and Adam compute in fp32.
# ----------------------------------------------------------------------------------------------------
MIXED = None
MIXED_BF16 = {'fwd': True, 'dgrad': True, 'dgradT': True, 'wgrad': True, 'wgrad_min_hw': 0, 'wgradT': True, 'dx16': True, 'y16': True}


def _r(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _MixedConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, transposed, cfg):
        ctx.save_for_backward(x, w)
        ctx.transposed, ctx.cfg = transposed, cfg
        xr, wr = (_r(x), _r(w)) if cfg['fwd'] else (x, w)
        if transposed:
            out = F.conv_transpose2d(xr, wr, b, stride=2, padding=1, output_padding=1)
        else:
            out = F.conv2d(xr, wr, b, padding=1)
        return _r(out) if cfg.get('y16', False) else out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        cfg, tr = ctx.cfg, ctx.transposed
        dx = dw = db = None
        dq = cfg['dgradT'] if tr else cfg['dgrad']
        dyd, wd = (_r(dy), _r(w)) if dq else (dy, w)
        wq = cfg.get('wgradT', False) if tr else (cfg['wgrad'] and x.shape[-1] >= cfg.get('wgrad_min_hw', 0))
        dyw, xw = (_r(dy), _r(x)) if wq else (dy, x)
        if tr:
            if ctx.needs_input_grad[0]:
                dx = F.conv2d(dyd, wd, None, stride=2, padding=1)
            if ctx.needs_input_grad[1]:
                dw = torch.nn.grad.conv2d_weight(dyw, w.shape, xw, stride=2, padding=1)
        else:
            if ctx.needs_input_grad[0]:
                dx = torch.nn.grad.conv2d_input(x.shape, wd, dyd, padding=1)
            if ctx.needs_input_grad[1]:
                dw = torch.nn.grad.conv2d_weight(xw, w.shape, dyw, padding=1)
        if ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        if dx is not None and cfg.get('dx16', False):
            dx = _r(dx)
        return dx, dw, db, None, None


class _RoundGrad(torch.autograd.Function):
    """identity whose gradient is rounded to bf16 (the activation gradient the output conv hands back, 'dx16')"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


def _conv3(x, w, b):
    if MIXED is None:
        return F.conv2d(x, w, b, padding=1)
    return _MixedConv.apply(x, w, b, False, MIXED)


def _convT(x, w, b):
    if MIXED is None:
        return F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    return _MixedConv.apply(x, w, b, True, MIXED)


# ----------------------------------------------------------------------------------------------------
# Functional network
# ----------------------------------------------------------------------------------------------------
def _double_conv(sd, prefix, x, train):
    # model/unet.py:9-16 : (conv3x3 p1 -> BN(eps 1e-5, momentum 0.1) -> ReLU) x 2
    for ci, bi in ((0, 1), (3, 4)):
        x = _conv3(x, sd['%s.%d.weight' % (prefix, ci)], sd['%s.%d.bias' % (prefix, ci)])
        rm, rv = sd['%s.%d.running_mean' % (prefix, bi)], sd['%s.%d.running_var' % (prefix, bi)]
        x = F.batch_norm(x, rm, rv, sd['%s.%d.weight' % (prefix, bi)], sd['%s.%d.bias' % (prefix, bi)],
                         training=train, momentum=0.1, eps=1e-5)
        if train:
            sd['%s.%d.num_batches_tracked' % (prefix, bi)] += 1
        x = F.relu(x)
    return x


def unet_forward(sd, stems, x, train):
    """One UNet (model/unet.py:187-241): inc, 3x down (maxpool2 + double_conv), 3x up
    (ConvTranspose2d k3 s2 p1 op1, cat([skip, up]), double_conv), 1x1 out conv."""
    x1 = _double_conv(sd, stems['inc'] + '.conv.conv', x, train)
    skips = [x1]
    h = x1
    for d in stems['down']:
        h = _double_conv(sd, d + '.mpconv.1.conv', F.max_pool2d(h, 2), train)
        skips.append(h)
    h = skips.pop()
    for u in stems['up']:
        up = _convT(h, sd[u + '.up.weight'], sd[u + '.up.bias'])
        h = _double_conv(sd, u + '.conv.conv', torch.cat([skips.pop(), up], dim=1), train)
    if MIXED is not None and MIXED.get('dx16', False):
        h = _RoundGrad.apply(h)
    return F.conv2d(h, sd[stems['outc'] + '.conv.weight'], sd[stems['outc'] + '.conv.bias'])


def bank_forward(sd, spec, x, x_of, train, padding=False):
    """SelfCompleteNet*.forward (model/unet.py:172-267, 410-556, 619-652).
    Returns (of_out, raw_out, of_tgt, raw_tgt) exactly like the reference (empty lists when no flow)."""
    raw_o, raw_t, of_o, of_t = [], [], [], []
    for u in spec:
        e = u['erase']
        if padding:
            inc = x.clone()
            inc[:, e * RAW_C:(e + 1) * RAW_C] = 0
        else:
            inc = torch.cat([x[:, :e * RAW_C], x[:, (e + 1) * RAW_C:]], dim=1)
        out = unet_forward(sd, u['stems'], inc, train)
        if u['role'] == 'raw':
            raw_o.append(out)
            raw_t.append(x[:, u['tgt'] * RAW_C:(u['tgt'] + 1) * RAW_C])
        else:
            of_o.append(out)
            of_t.append(x_of[:, u['tgt'] * OF_C:(u['tgt'] + 1) * OF_C])
    raw_o, raw_t = torch.cat(raw_o, 1), torch.cat(raw_t, 1)
    if of_o:
        of_o, of_t = torch.cat(of_o, 1), torch.cat(of_t, 1)
    return of_o, raw_o, of_t, raw_t


def cube_scores(out, tgt):
    """train.py:421-426 / test.py:330-335: squared error summed over (C,H,W) per cube."""
    return ((tgt - out) ** 2).sum(dim=(1, 2, 3))


def train_loss(of_o, raw_o, of_t, raw_t, lambda_raw=1.0, lambda_of=1.0, useFlow=True):
    """train.py:385-392."""
    l_raw = F.mse_loss(raw_o, raw_t.detach())
    if useFlow:
        l_of = F.mse_loss(of_o, of_t.detach())
        return lambda_raw * l_raw + lambda_of * l_of, l_raw, l_of
    return l_raw, l_raw, None


def param_names(sd):
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked'))]


class AdamState:
    """torch.optim.Adam(lr=1e-3, betas=(0.9,0.999), eps=1e-7, weight_decay=0) restated (train.py:376)."""

    def __init__(self, names, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m = {n: None for n in names}
        self.v = {n: None for n in names}

    def step(self, sd, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        for n, g in grads.items():
            if g is None:
                continue
            if self.m[n] is None:
                self.m[n] = torch.zeros_like(g)
                self.v[n] = torch.zeros_like(g)
            self.m[n].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[n].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[n].sqrt() / (bc2 ** 0.5)).add_(self.eps)
            sd[n].data.addcdiv_(self.m[n], denom, value=-self.lr / bc1)


def train_step(sd, spec, x, x_of, opt, lambda_raw=1.0, lambda_of=1.0, padding=False, useFlow=True):
    """One iteration of train.py:379-402.  Returns (loss_raw, loss_of, grads)."""
    names = param_names(sd)
    for n in names:
        sd[n].requires_grad_(True)
        sd[n].grad = None
    of_o, raw_o, of_t, raw_t = bank_forward(sd, spec, x, x_of, True, padding)
    loss, l_raw, l_of = train_loss(of_o, raw_o, of_t, raw_t, lambda_raw, lambda_of, useFlow)
    loss.backward()
    grads = {n: (sd[n].grad.detach().clone() if sd[n].grad is not None else None) for n in names}
    with torch.no_grad():
        for n in names:
            sd[n].requires_grad_(False)
        opt.step(sd, grads)
    return float(l_raw.detach()), (float(l_of.detach()) if l_of is not None else 0.0), grads


@torch.no_grad()
def score_pass(sd, spec, x, x_of, batch_size, padding=False, useFlow=True):
    """train.py:413-427 (eval-mode pass, shuffle=False) -> (raw_scores[N], of_scores[N])."""
    rs, os_ = [], []
    for s in range(0, x.shape[0], batch_size):
        of_o, raw_o, of_t, raw_t = bank_forward(sd, spec, x[s:s + batch_size], x_of[s:s + batch_size], False, padding)
        rs.append(cube_scores(raw_o, raw_t).numpy())
        if useFlow:
            os_.append(cube_scores(of_o, of_t).numpy())
    return np.concatenate(rs), (np.concatenate(os_) if useFlow else None)


def normalised_scores(raw_s, of_s, raw_train, of_train, w_raw=1.0, w_of=1.0):
    """test.py:260-266,338-345: population std (np.std, ddof=0)."""
    r = (raw_s - np.mean(raw_train)) / np.std(raw_train)
    if of_s is None:
        return w_raw * r
    o = (of_s - np.mean(of_train)) / np.std(of_train)
    return w_raw * r + w_of * o


def frame_score_map(scores, bboxes, h, w, big_number=100000):
    """test.py:276,350-357: paint each cube score into its bbox, max-combine, empty -> -1e5."""
    res = -1.0 * np.ones((h, w)) * big_number
    for m in range(len(scores)):
        x_min, x_max = int(np.ceil(bboxes[m][0])), int(np.ceil(bboxes[m][2]))
        y_min, y_max = int(np.ceil(bboxes[m][1])), int(np.ceil(bboxes[m][3]))
        mask = -1.0 * np.ones((h, w)) * big_number
        mask[y_min:y_max, x_min:x_max] = scores[m]
        res = np.maximum(res, mask)
    return res


def roc_auc(scores, labels):
    """utils.py:29-39 (sklearn roc_curve + auc) restated as the tie-aware rank statistic."""
    scores = np.asarray(scores, dtype=np.float64).ravel()
    labels = np.asarray(labels).ravel().astype(bool)
    pos, neg = scores[labels], scores[~labels]
    if len(pos) == 0 or len(neg) == 0:
        return float('nan')
    order = np.argsort(np.concatenate([neg, pos]), kind='mergesort')
    allv = np.concatenate([neg, pos])[order]
    ranks = np.empty(len(allv), dtype=np.float64)
    i = 0
    while i < len(allv):
        j = i
        while j + 1 < len(allv) and allv[j + 1] == allv[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty(len(allv))
    r[order] = ranks
    rp = r[len(neg):].sum()
    return float((rp - len(pos) * (len(pos) + 1) / 2.0) / (len(pos) * len(neg)))
