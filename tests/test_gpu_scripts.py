"""GPU: train.py / test.py loop bodies on a tiny synthetic dataset laid out like UCSDped2, against the golden run of the
real reference modules (tests/golden/script_net4.npz: train.py:365-433 + test.py:251-357 + utils.py:29-39)."""
import os

import numpy as np
import pytest
import torch

from _util import load_golden

pytestmark = pytest.mark.gpu


def test_train_then_test_scripts_match_reference(tmp_path, monkeypatch):
    from oracle import unet_oracle as O
    import train as T
    import test as S
    from model.unet import SelfCompleteNet4
    g = load_golden('script_net4')
    torch.manual_seed(0)
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None,
                           useFlow=True, padding=False)
    net.load_state_dict(O.seeded_state_dict('net4', nf=32, padding=False, seed=0))
    raw, flow = O.seeded_cubes(24, 1, 1)
    lines = []
    sd, raw_train, of_train = T.train_block(net, [lambda: (raw, flow[:, 0])], epochs=2, batch_size=8, shuffle_seed=-1,
                                            device='cuda', log=lines.append)
    assert all(k.startswith('module.') for k in sd)
    np.testing.assert_allclose(raw_train, g['raw_train'], rtol=5e-3)
    np.testing.assert_allclose(of_train, g['of_train'], rtol=5e-3)
    # running-average loss lines (printed every 5 batches): first line is batch 0 of epoch 0
    first = lines[0]
    assert 'raw loss: ' in first
    l_raw0 = float(first.split('raw loss: ')[1].split(',')[0])
    assert abs(l_raw0 - g['losses'][0][0]) <= 1e-3 * g['losses'][0][0]

    # ---- test stage on synthetic frames (same construction as the golden script)
    rng = np.random.default_rng(77)
    h, w = 240, 360
    fset, fset2, bset, labels = [], [], [], []
    for f in range(10):
        nc = f % 4
        if nc == 0:
            fset.append([[np.zeros((0, 5, 32, 32, 3), np.uint8)]])
            fset2.append([[np.zeros((0, 32, 32, 2), np.float32)]])
            bset.append([[np.zeros((0, 4))]])
        else:
            rw, fl = O.seeded_cubes(nc, 1, 100 + f)
            if f % 2 == 1:
                rw = rw.copy()
                rw[:, 4] = 255 - rw[:, 4]           # same perturbation as 1 - x on the [0,1] tensor
            bbs = []
            for m in range(nc):
                x0, y0 = rng.uniform(0, w - 60), rng.uniform(0, h - 60)
                bbs.append([x0, y0, x0 + rng.uniform(10, 50), y0 + rng.uniform(10, 50)])
            fset.append([[rw]])
            fset2.append([[fl[:, 0]]])
            bset.append([[np.array(bbs)]])
        labels.append(f % 2 == 1)
    net2 = S.load_model(SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict',
                                         rawRange=None, useFlow=True, padding=False), sd, 'cuda')
    stats_r = [[(np.mean(raw_train), np.std(raw_train))]]
    stats_o = [[(np.mean(of_train), np.std(of_train))]]
    out_dir = str(tmp_path / 'score_mask')
    fs = S.score_frames([[[net2]]], stats_r, stats_o, fset, fset2, bset, h, w, 1.0, 1.0, True, 'cuda', score_batch=5,
                        result_dir=out_dir)
    np.testing.assert_allclose(fs, g['frame_scores'], rtol=1e-2, atol=1e-2)
    # the saved maps have the reference's format (torch-pickled float64 [h,w]) and the same maxima
    m3 = torch.load(os.path.join(out_dir, '3'), weights_only=False)
    assert m3.shape == (h, w) and m3.dtype == np.float64 and abs(m3.max() - fs[3]) < 1e-12
    from utils import save_roc_pr_curve_data, frame_roc_auc
    auc = save_roc_pr_curve_data(fs, np.array(labels), str(tmp_path / 'roc.npz'), verbose=False)
    assert abs(auc - float(g['auc'])) <= 1e-3
    assert abs(frame_roc_auc(fs, np.array(labels)) - auc) < 1e-12


@pytest.mark.gpu
def test_calc_optical_flow_driver(tmp_path, monkeypatch):
    """calc_optical_flow.py end to end on a synthetic UCSD-style tree: file layout, border pair selection, and
    saved flow == resize_back(FlowNet2(resize(pair))) with both resizes checked against the cv2-arithmetic oracle."""
    from PIL import Image
    import calc_optical_flow as COF
    from oracle import resize_oracle as R
    from vad_datasets import unified_dataset_interface
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    H, W = 48, 72
    frames = {}
    for v, n in (('Train001', 3), ('Train002', 2)):
        os.makedirs(os.path.join('raw_datasets', 'UCSDped2', 'Train', v))
        base = rng.integers(0, 256, (H, W + 8), dtype=np.uint8)
        for k in range(n):
            g = np.ascontiguousarray(base[:, k * 2:k * 2 + W])          # a pattern sliding 2 px per frame
            frames[(v, k)] = g
            Image.fromarray(g).save(os.path.join('raw_datasets', 'UCSDped2', 'Train', v, '%03d.tif' % (k + 1)))
    ds = unified_dataset_interface('UCSDped2', os.path.join('raw_datasets', 'UCSDped2'), context_frame_num=1, mode='train',
                                   border_mode='hard')
    torch.manual_seed(0)
    net = COF.FlowNet2().cuda().eval()
    COF.calc_optical_flow(ds, flownet2=net, log=lambda *a: None)
    order = [('Train001', 0), ('Train001', 1), ('Train001', 2), ('Train002', 0), ('Train002', 1)]
    # (first, second) frame matched for each index: border frames use the first two of the clipped context
    pairs = [(0, 0), (1, 2), (1, 2), (3, 3), (3, 4)]
    for idx, (v, k) in enumerate(order):
        path = os.path.join('optical_flow', 'UCSDped2', 'Train', v, '%03d.npy' % (k + 1))
        got = np.load(path)
        assert got.shape == (H, W, 2) and got.dtype == np.float32
        a, b = (frames[order[j]] for j in pairs[idx])
        im = [np.repeat(R.resize_linear(x, (512, 384))[:, :, None], 3, 2) for x in (a, b)]
        ims = np.array([im]).transpose((0, 4, 1, 2, 3)).astype(np.float32)
        flow = net(torch.from_numpy(ims).cuda())[0].cpu().numpy().transpose((1, 2, 0))
        ref = R.resize_linear(np.ascontiguousarray(flow), (W, H))
        assert np.array_equal(got, ref), idx
    assert np.isfinite(got).all()
