"""GPU bring-up tool (not a pytest test): runs the HIP UNet bank one launch at a time and compares EVERY intermediate
(pre-BN conv outputs, transposed-conv outputs, BN scale/shift, reconstructions, scores, per-layer dy, dA, and all
parameter gradients) against a plain torch-CPU evaluation of the same graph.

    gpurun -- python tests/gpu_debug_layers.py [net4|full] [B]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402
from vec_vad_amd.bank import conv_key_to_state_name  # noqa: E402
from model.unet import SelfCompleteNet4, SelfCompleteNetFull  # noqa: E402


def err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = (a - b).abs().max().item()
    s = b.abs().max().item() + 1e-30
    return d, d / s


def cpu_graph(sd, stems, x_inc, train):
    """returns dict of intermediates with autograd: y[l] (pre-BN incl. bias), t[u], out."""
    ys, ts = [], []

    def dc(prefix, x):
        for ci, bi in ((0, 1), (3, 4)):
            y = F.conv2d(x, sd['%s.%d.weight' % (prefix, ci)], sd['%s.%d.bias' % (prefix, ci)], padding=1)
            y.retain_grad()
            ys.append(y)
            x = F.relu(F.batch_norm(y, sd['%s.%d.running_mean' % (prefix, bi)].clone(), sd['%s.%d.running_var' % (prefix, bi)].clone(),
                                    sd['%s.%d.weight' % (prefix, bi)], sd['%s.%d.bias' % (prefix, bi)], training=train,
                                    momentum=0.1, eps=1e-5))
        return x

    x1 = dc(stems['inc'] + '.conv.conv', x_inc)
    skips = [x1]
    h = x1
    for d in stems['down']:
        h = dc(d + '.mpconv.1.conv', F.max_pool2d(h, 2))
        skips.append(h)
    h = skips.pop()
    for u in stems['up']:
        t = F.conv_transpose2d(h, sd[u + '.up.weight'], sd[u + '.up.bias'], stride=2, padding=1, output_padding=1)
        t.retain_grad()
        ts.append(t)
        h = dc(u + '.conv.conv', torch.cat([skips.pop(), t], 1))
    out = F.conv2d(h, sd[stems['outc'] + '.conv.weight'], sd[stems['outc'] + '.conv.bias'])
    return ys, ts, out


def nhwc(t):   # [B,C,H,W] -> [B*H*W, C]
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'net4'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    padding = len(sys.argv) > 3 and sys.argv[3] == 'pad'
    torch.manual_seed(0)
    tot_of = 1 if kind == 'net4' else 5
    cls = SelfCompleteNet4 if kind == 'net4' else SelfCompleteNetFull
    model = cls(features_root=32, tot_raw_num=5, tot_of_num=tot_of, border_mode='predict', rawRange=None, useFlow=True,
                padding=padding)
    sd = O.seeded_state_dict(kind, nf=32, padding=padding, seed=0)
    model.load_state_dict(sd)
    model = model.cuda()
    model.train()
    raw, flow = O.seeded_cubes(B, tot_of, 0)
    x, x_of = O.cubes_to_inputs(raw, flow)
    bank = model.bank()
    ws = bank.set_input_nchw(x.cuda(), x_of.cuda())
    torch.cuda.synchronize()
    # cube adapter kernel vs NCHW path
    ws_c = ws.cube.clone()
    bank.set_input_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), None, B)
    torch.cuda.synchronize()
    print('cube_gather vs nchw_to_nhwc: max diff', (ws.cube - ws_c).abs().max().item(),
          ' vs oracle', (ws.cube.cpu() - nhwc(x).reshape(ws.cube.shape)).abs().max().item())
    stream = torch.cuda.current_stream().cuda_stream
    # ---- forward, one launch at a time
    for fn, args, label in ws.fwd[True].calls:
        rc = fn(*args, stream)
        torch.cuda.synchronize()
        if rc:
            print('LAUNCH FAILED', label, rc)
            return 1
    units = bank.units
    spec = O.bank_spec(kind)
    # map bank unit order -> oracle stems
    table = model._unit_table()
    for p in [k for k in sd if sd[k].dtype.is_floating_point]:
        sd[p] = sd[p].clone()
    pn = O.param_names(sd)
    for n in pn:
        sd[n].requires_grad_(True)
    worst = 0.0
    total_loss = 0
    cpu_inter = []
    for g, (u, stems) in enumerate(table):
        e = u.erase
        if padding:
            inc = x.clone()
            inc[:, e * 3:(e + 1) * 3] = 0
        else:
            inc = torch.cat([x[:, :e * 3], x[:, (e + 1) * 3:]], 1)
        ys, ts, out = cpu_graph(sd, stems, inc, True)
        out.retain_grad()
        cpu_inter.append((ys, ts, out))
        for l, y in enumerate(ys):
            d, r = err(ws.y[l][g], nhwc(y))
            worst = max(worst, r)
            if r > 1e-4:
                print('  FWD g%d y%d  abs %.3e rel %.3e' % (g, l, d, r))
        for ui, t in enumerate(ts):
            d, r = err(ws.t[ui][g], nhwc(t))
            worst = max(worst, r)
            if r > 1e-4:
                print('  FWD g%d t%d  abs %.3e rel %.3e' % (g, ui, d, r))
        d, r = err(ws.out4[g][:, :u.out_c], nhwc(out))
        worst = max(worst, r)
        print('g%d (%s erase %d) out  abs %.3e rel %.3e' % (g, u.role, e, d, r))
        tgt = x[:, u.tgt * 3:(u.tgt + 1) * 3] if u.role == 'raw' else x_of[:, u.tgt * 2:(u.tgt + 1) * 2]
        sc = ((out - tgt) ** 2).sum(dim=(1, 2, 3))
        d, r = err(ws.score[g], sc)
        print('    score rel %.3e' % r)
        n_same = sum(1 for v in units if v.role == u.role)
        total_loss = total_loss + ((out - tgt) ** 2).sum() / (B * n_same * u.out_c * 1024)
    print('forward worst rel err %.3e' % worst)
    total_loss.backward()
    # dout
    for g, (u, stems) in enumerate(table):
        ys, ts, out = cpu_inter[g]
        d, r = err(ws.dout4[g][:, :u.out_c], nhwc(out.grad))
        if r > 1e-4:
            print('  dout g%d rel %.3e' % (g, r))
    # running stats
    # ---- backward, one launch at a time
    bank.backward  # noqa
    if ws.bwd is None:
        ws.bwd = bank._plan_backward(ws, B)
    lay = bank.lay
    for fn, args, label in ws.bwd.calls:
        rc = fn(*args, stream)
        torch.cuda.synchronize()
        if rc:
            print('LAUNCH FAILED', label, rc)
            return 1
        if label.startswith('bn_bwd_apply'):
            l = int(label[len('bn_bwd_apply'):])
            L_ = lay.convs[l]
            M = B * L_.H * L_.H
            pos = len(lay.convs) - 1 - l          # backward visiting order 13, 12, ...; dy buffers alternate
            for g in range(bank.G):
                dz = ws.dz2[pos % 2][g][:M * L_.cout].view(M, L_.cout)
                d, r = err(dz, nhwc(cpu_inter[g][0][l].grad))
                if r > 2e-4:
                    print('  BWD g%d dy%d abs %.3e rel %.3e' % (g, l, d, r))
        if label.startswith('dgradT'):
            pass
    # parameter grads
    gw = 0.0
    bad = 0
    for g, (u, stems) in enumerate(table):
        for key, (off, shape) in lay.p.items():
            name = conv_key_to_state_name(stems, key)
            ref = sd[name].grad
            n = ref.numel()
            mine = bank.grad_view(g, key, shape=ref.shape)
            d, r = err(mine, ref)
            is_bn_bias_conv = key.startswith('c') and key.endswith('.b')
            if is_bn_bias_conv:
                continue
            gw = max(gw, r)
            if r > 5e-4:
                bad += 1
                if bad < 40:
                    print('  GRAD g%d %-40s abs %.3e rel %.3e  |ref| %.3e' % (g, name, d, r, ref.abs().max().item()))
    print('param-grad worst rel err %.3e  (bad: %d)' % (gw, bad))
    return 0


if __name__ == '__main__':
    sys.exit(main())
