"""Generate golden vectors by running the REAL reference modules (authoring container only).

    python tests/golden/make_goldens.py        # needs /root/reference ; writes tests/golden/*.npz

The reference has no tests or fixtures of its own, so parity is pinned by the outputs of
the reference's ``model/unet.py`` classes (imported, never copied) on formula-seeded weights and cubes.  The
weights/cubes are regenerated on any box from numpy PCG64 streams (``oracle.unet_oracle.seeded_state_dict`` /
``seeded_cubes``), so the fixtures only hold *outputs*: per-cube scores, losses, and digests of outputs, gradients
and updated parameters.  Nothing under /root/reference is read at test time.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402

np.int = int  # reference uses the removed alias (utils.py:22, vad_datasets.py:74-83)
sys.path.insert(0, '/root/reference')
# the reference's model/ has no __init__.py (namespace package) and would lose against this repo's regular package of the
# same name: load the reference file explicitly (imported from where it lies, never copied)
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location('ref_model_unet', '/root/reference/model/unet.py')
_ref_unet = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_ref_unet)
SelfCompleteNet4, SelfCompleteNetFull, SelfCompleteNet1raw1of = (_ref_unet.SelfCompleteNet4, _ref_unet.SelfCompleteNetFull,
                                                                  _ref_unet.SelfCompleteNet1raw1of)
assert SelfCompleteNet4.__module__ == 'ref_model_unet'

# stubs so that vad_datasets imports
sys.modules['cv2'] = types.ModuleType('cv2')
tv = types.ModuleType('torchvision')
tvt = types.ModuleType('torchvision.transforms')


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _ToTensor:
    def __call__(self, pic):
        img = torch.from_numpy(np.ascontiguousarray(pic.transpose((2, 0, 1))))
        if isinstance(img, torch.ByteTensor):
            return img.float().div(255)
        return img


tvt.Compose, tvt.ToTensor = _Compose, _ToTensor
tv.transforms = tvt
sys.modules['torchvision'] = tv
sys.modules['torchvision.transforms'] = tvt
import vad_datasets as ref_ds  # noqa: E402
from utils import save_roc_pr_curve_data  # noqa: E402


def digest(t, k=8):
    """Small, order-sensitive digest of a tensor: [sum, sum|x|, sum x^2, k strided samples]."""
    a = t.detach().double().reshape(-1)
    idx = torch.linspace(0, a.numel() - 1, k).long()
    w = torch.cos(torch.arange(a.numel(), dtype=torch.float64) * 0.37)  # position-sensitive
    return np.concatenate([[a.sum().item(), a.abs().sum().item(), (a * a).sum().item(), (a * w).sum().item()],
                           a[idx].numpy()])


def build_ref(kind, nf, padding, rawRange=None):
    cls = {'net4': SelfCompleteNet4, 'full': SelfCompleteNetFull, '1raw1of': SelfCompleteNet1raw1of}[kind]
    tot_of = {'net4': 1, 'full': 5, '1raw1of': 1}[kind]
    net = cls(features_root=nf, tot_raw_num=5, tot_of_num=tot_of, border_mode='predict', rawRange=rawRange,
              useFlow=True, padding=padding)
    sd = O.seeded_state_dict(kind, nf=nf, padding=padding, seed=0)
    missing = net.load_state_dict(sd, strict=True)
    return net, tot_of


def ref_inputs(n, tot_of, seed):
    raw, flow = O.seeded_cubes(n, tot_of, seed)
    if tot_of == 1:
        flow_in = flow[:, 0]  # train.py saves [N,32,32,2] for context_of_num=0; cube_to_train_dataset adds the axis
    else:
        flow_in = flow
    ds = ref_ds.cube_to_train_dataset(raw, target=flow_in)
    items = [ds[i] for i in range(n)]
    x = torch.stack([it[0] for it in items]).float()
    x_of = torch.stack([it[1] for it in items]).float()
    return raw, flow, x, x_of


def case_model(kind, nf, padding, n, rawRange=None):
    torch.manual_seed(0)
    net, tot_of = build_ref(kind, nf, padding, rawRange)
    raw, flow, x, x_of = ref_inputs(n, tot_of, seed=0)
    out = {}
    # adapter parity (vad_datasets.py:130-168)
    out['x_digest'] = digest(x)
    out['xof_digest'] = digest(x_of)
    # ---- eval forward + per-cube scores (test.py:319-335)
    net.eval()
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = net(x, x_of)
        sf = torch.nn.MSELoss(reduce=False)
        out['eval_raw_scores'] = sf(raw_t, raw_o).numpy().sum(3).sum(2).sum(1)
        out['eval_of_scores'] = sf(of_t, of_o).numpy().sum(3).sum(2).sum(1)
        out['eval_raw_out_digest'] = digest(raw_o)
        out['eval_of_out_digest'] = digest(of_o)
        out['eval_raw_out_shape'] = np.array(raw_o.shape)
        out['eval_of_out_shape'] = np.array(of_o.shape)
    # ---- 3 train steps (train.py:376-402)
    net.train()
    opt = torch.optim.Adam(net.parameters(), eps=1e-7, weight_decay=0.0)
    lf = torch.nn.MSELoss()
    names = [k for k, _ in net.named_parameters()]
    losses = []
    for step in range(3):
        of_o, raw_o, of_t, raw_t = net(x, x_of)
        l_raw = lf(raw_t.detach(), raw_o)
        l_of = lf(of_t.detach(), of_o)
        loss = 1.0 * l_raw + 1.0 * l_of
        losses.append([l_raw.item(), l_of.item()])
        opt.zero_grad()
        loss.backward()
        if step == 0:
            out['train_raw_out_digest'] = digest(raw_o)
            out['train_of_out_digest'] = digest(of_o)
            g = dict(net.named_parameters())
            out['grad_names'] = np.array(names)
            out['grad_digests'] = np.stack([digest(g[k].grad) if g[k].grad is not None else np.zeros(12) for k in names])
        opt.step()
    out['losses'] = np.array(losses)
    sd = net.state_dict()
    keys = list(sd.keys())
    out['final_names'] = np.array(keys)
    out['final_digests'] = np.stack([digest(sd[k].double()) if sd[k].numel() > 1 else
                                     np.full(12, float(sd[k])) for k in keys])
    # ---- eval scores after training (train.py:413-427)
    net.eval()
    with torch.no_grad():
        of_o, raw_o, of_t, raw_t = net(x, x_of)
        out['post_raw_scores'] = sf(raw_t, raw_o).numpy().sum(3).sum(2).sum(1)
        out['post_of_scores'] = sf(of_t, of_o).numpy().sum(3).sum(2).sum(1)
    return out


def case_script(kind='net4', nf=32, n_train=24, n_test_frames=10, batch=8, epochs=2, graded=False):
    """train.py:365-433 + test.py:251-357 + utils.py:29-39 driven with a fixed (shuffle=False) order.
    graded=True (the 240-frame case): an "anomalous" frame shifts its last raw frame by 1..3 pixels instead of inverting it, so
    normal and anomalous scores overlap and the AUROC is sensitive to rank swaps (AUROC within 1e-3 is then a real statement:
    one swapped pair among ~90 x 90 moves it by 1.2e-4)."""
    torch.manual_seed(0)
    net, tot_of = build_ref(kind, nf, False)
    raw, flow, x, x_of = ref_inputs(n_train, tot_of, seed=1)
    opt = torch.optim.Adam(net.parameters(), eps=1e-7, weight_decay=0.0)
    lf = torch.nn.MSELoss()
    net.train()
    losses = []
    for ep in range(epochs):
        for s in range(0, n_train, batch):
            of_o, raw_o, of_t, raw_t = net(x[s:s + batch], x_of[s:s + batch])
            l_raw, l_of = lf(raw_t.detach(), raw_o), lf(of_t.detach(), of_o)
            losses.append([l_raw.item(), l_of.item()])
            opt.zero_grad()
            (l_raw + l_of).backward()
            opt.step()
    net.eval()
    sf = torch.nn.MSELoss(reduce=False)
    rs, os_ = [], []
    with torch.no_grad():
        for s in range(0, n_train, batch):
            of_o, raw_o, of_t, raw_t = net(x[s:s + batch], x_of[s:s + batch])
            rs.append(sf(raw_t, raw_o).numpy().sum(3).sum(2).sum(1))
            os_.append(sf(of_t, of_o).numpy().sum(3).sum(2).sum(1))
    raw_train, of_train = np.concatenate(rs), np.concatenate(os_)
    # test frames: frame f has (f % 4) cubes (0 => empty frame); labels alternate
    rng = np.random.default_rng(77)
    h, w = 240, 360
    frame_scores, labels, all_cube_scores = [], [], []
    cube_seed = 100
    for f in range(n_test_frames):
        nc = f % 4
        res = -1.0 * np.ones((h, w)) * 100000
        if nc > 0:
            _, _, xt, xt_of = ref_inputs(nc, tot_of, seed=cube_seed + f)
            if f % 2 == 1:  # "anomalous": perturb the last frame so reconstruction error grows
                xt = xt.clone()
                if graded:
                    xt[:, 12:15] = torch.roll(xt[:, 12:15], 1 + f % 3, dims=3)
                else:
                    xt[:, 12:15] = 1.0 - xt[:, 12:15]
            with torch.no_grad():
                of_o, raw_o, of_t, raw_t = net(xt, xt_of)
            r = sf(raw_t, raw_o).numpy().sum(3).sum(2).sum(1)
            o = sf(of_t, of_o).numpy().sum(3).sum(2).sum(1)
            r = (r - np.mean(raw_train)) / np.std(raw_train)
            o = (o - np.mean(of_train)) / np.std(of_train)
            sc = 1.0 * r + 1.0 * o
            all_cube_scores.append(sc)
            bbs = []
            for m in range(nc):
                x0, y0 = rng.uniform(0, w - 60), rng.uniform(0, h - 60)
                bbs.append([x0, y0, x0 + rng.uniform(10, 50), y0 + rng.uniform(10, 50)])
            for m in range(nc):
                mask = -1.0 * np.ones((h, w)) * 100000
                bb = bbs[m]
                mask[int(np.ceil(bb[1])):int(np.ceil(bb[3])), int(np.ceil(bb[0])):int(np.ceil(bb[2]))] = sc[m]
                res = np.max(np.concatenate([res[:, :, None], mask[:, :, None]], axis=2), axis=2)
        frame_scores.append(res.max())
        labels.append(f % 2 == 1)
    auc = save_roc_pr_curve_data(np.array(frame_scores), np.array(labels), '/tmp/_golden_roc.npz', verbose=False)
    return dict(losses=np.array(losses), raw_train=raw_train, of_train=of_train,
                frame_scores=np.array(frame_scores), labels=np.array(labels),
                cube_scores=np.concatenate(all_cube_scores), auc=np.array(auc))


def main():
    cases = {
        'net4_nf32_nopad': lambda: case_model('net4', 32, False, 6),
        'net4_nf32_pad': lambda: case_model('net4', 32, True, 3),
        'full_nf32_nopad': lambda: case_model('full', 32, False, 4),
        'net4_nf32_rawrange4': lambda: case_model('net4', 32, False, 3, rawRange=4),
        '1raw1of_nf32_nopad': lambda: case_model('1raw1of', 32, False, 3),
        '1raw1of_nf64_nopad': lambda: case_model('1raw1of', 64, False, 3),      # the class's default features_root (model/unet.py:563)
        'script_net4': lambda: case_script(),
        'script_net4_f240': lambda: case_script(n_test_frames=240, graded=True),
    }
    only = sys.argv[1:]
    for name, fn in cases.items():
        if only and name not in only:
            continue
        out = fn()
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        print(name, '->', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
