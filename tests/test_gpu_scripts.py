"""GPU: train.py / test.py loop bodies on a tiny synthetic dataset laid out like UCSDped2, against the golden run of the
real reference modules (tests/golden/script_net4.npz: train.py:365-433 + test.py:251-357 + utils.py:29-39)."""
import os

import numpy as np
import pytest
import torch

from _util import load_golden

pytestmark = pytest.mark.gpu


from _util import observe as _observe  # noqa: E402


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


@pytest.mark.parametrize('precision,golden', [('fp32', 'script_net4'), ('fp32', 'script_net4_f240'), ('bf16', 'script_net4'),
                                              ('bf16', 'script_net4_f240')])
def test_train_then_test_scripts_match_reference(tmp_path, monkeypatch, precision, golden):
    """train.py's training block + test.py's scoring / frame maps / AUROC against the reference's own outputs (golden).
    fp32: the north-star tolerances -- per-cube training scores rel <= 1e-3, AUROC abs <= 1e-3; the oracle's
    own fp32-vs-fp64 spread through the same 6 Adam steps is ~3e-4 on the scores
    (tests/test_oracle_golden.py::test_oracle_fp32_fp64_spread_after_training).  z-normalised frame scores abs <= 5e-3: the
    normalisation (s - mu) / sigma amplifies a relative per-cube error by mu / sigma = 106 (raw) + 35 (flow) on this training
    set (sigma is 1 % of mu), so 5e-3 on z is a per-cube agreement of 3.5e-5 -- 30x tighter than the per-cube bar.
    OBSERVED (round 6, one box, gpurun_out/observed.jsonl = profiles/r06_parity_switches.jsonl; worst of the 10- and 240-frame goldens):

        switch                               train_raw   train_of   frame (f240)   AUROC
        default (F(2x2) Winograd fwd+dgrad)   3.19e-4     3.58e-5    3.84e-3        6.9e-5
        VV_FUSE_BN_SUMS=0                     2.40e-4     3.16e-5    3.51e-3        6.9e-5
        VV_WINO44=0 | dgrad                   (no effect: at 8-cube batches the policy routes no launch to F(4x4))
        VV_WINOGRAD_WGRAD=0                   3.05e-4     3.49e-5    3.85e-3        6.9e-5
        VV_WINOGRAD=0 (direct 3x3 kernels)    1.82e-4     4.08e-5    9.39e-4        6.9e-5

    i.e. the 3.8e-3 on the z-normalised frame scores is the Winograd FORWARD / data-gradient form (a few ulp per layer, tested <= 2e-5
    of the tensor maximum against the direct kernel) carried through six Adam steps and multiplied by mu / sigma ~ 141: a per-cube
    agreement of 2.7e-5.  It has been there since the Winograd path became the default (round 2; the "1.3e-3" this docstring quoted
    until round 5 was measured on the direct kernels of round 1 and never updated); neither the fused BatchNorm-backward sums nor
    F(4x4) moved it.  The frame bar stays at 5e-3 = 1.3 x observed (tighter than 2 x observed); the per-cube bar is the north star's.
    The 240-frame golden has graded anomalies (normal and anomalous scores overlap, AUROC 0.768): one swapped pair moves its AUROC
    by 1.2e-4, so the AUROC bar is a real statement there; on the 10-frame golden it only says the ranking is identical.
    bf16 (`[mi355x] precision = bf16`, BASELINE config 4), against the REFERENCE's fp32 golden.  OBSERVED (round 6, same run): training
    scores 1.32e-3 / 1.45e-4, first loss 5.9e-5, z-normalised frame scores 6.55e-3 (7.25e-3 with VV_FUSE_BN_SUMS=0), AUROC 6.9e-5.  Bars =
    2 x observed, rounded: training scores 2e-3 (unchanged since round 4), loss 1.2e-4 (was 2e-4), frame scores 1.4e-2 (was 2e-2), AUROC 1e-3
    (the config's own criterion).  Why the bf16 training-score bar is 2e-3 and not the fp32 path's 1e-3 (round 4):
    after 6 Adam steps the bf16 path sits 6.4e-4 from the reference's fp32 scores with the round-3 conv kernel and 1.26e-3 with the
    round-4 one (VV_CONV_GEMM16=0 / 1), although the two kernels' convolution outputs are BIT-EQUAL (tests/test_gpu_bf16.py): they sum
    the BatchNorm partial sums in a different order, scale / shift move by 1e-7 relative, a handful of activations on a bf16 rounding
    boundary flip by one ulp (2^-8), and six training steps on 8-cube batches carry that to a coherent ~1e-3 shift of all 24 scores.
    That spread between two correct implementations IS the resolution of this comparison."""
    monkeypatch.setenv('VV_PRECISION', precision)
    tol = {'fp32': dict(train=1e-3, loss=1e-3, frame=5e-3, auc=1e-3), 'bf16': dict(train=2e-3, loss=1.2e-4, frame=1.4e-2, auc=1e-3)}[precision]
    from oracle import unet_oracle as O
    import train as T
    import test as S
    from model.unet import SelfCompleteNet4
    g = load_golden(golden)
    n_frames, graded = len(g['frame_scores']), golden.endswith('f240')
    torch.manual_seed(0)
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None,
                           useFlow=True, padding=False)
    net.load_state_dict(O.seeded_state_dict('net4', nf=32, padding=False, seed=0))
    raw, flow = O.seeded_cubes(24, 1, 1)
    lines = []
    sd, raw_train, of_train = T.train_block(net, [lambda: (raw, flow[:, 0])], epochs=2, batch_size=8, shuffle_seed=-1,
                                            device='cuda', log=lines.append)
    assert all(k.startswith('module.') for k in sd)
    obs = dict(train_raw=_rel(raw_train, g['raw_train']), train_of=_rel(of_train, g['of_train']))
    np.testing.assert_allclose(raw_train, g['raw_train'], rtol=tol['train'])
    np.testing.assert_allclose(of_train, g['of_train'], rtol=tol['train'])
    # running-average loss lines (printed every 5 batches): first line is batch 0 of epoch 0
    first = lines[0]
    assert 'raw loss: ' in first
    l_raw0 = float(first.split('raw loss: ')[1].split(',')[0])
    obs['loss0'] = abs(l_raw0 - g['losses'][0][0]) / g['losses'][0][0]
    assert abs(l_raw0 - g['losses'][0][0]) <= tol['loss'] * g['losses'][0][0]

    # ---- test stage on synthetic frames (same construction as the golden script)
    rng = np.random.default_rng(77)
    h, w = 240, 360
    fset, fset2, bset, labels = [], [], [], []
    for f in range(n_frames):
        nc = f % 4
        if nc == 0:
            fset.append([[np.zeros((0, 5, 32, 32, 3), np.uint8)]])
            fset2.append([[np.zeros((0, 32, 32, 2), np.float32)]])
            bset.append([[np.zeros((0, 4))]])
        else:
            rw, fl = O.seeded_cubes(nc, 1, 100 + f)
            if f % 2 == 1:
                rw = rw.copy()
                if graded:
                    rw[:, 4] = np.roll(rw[:, 4], 1 + f % 3, axis=2)        # torch.roll(x[:, 12:15], k, dims=3) on the NCHW tensor
                else:
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
    obs['frame'] = float(np.max(np.abs(np.asarray(fs) - g['frame_scores']) / (1.0 + np.abs(g['frame_scores']))))
    from utils import save_roc_pr_curve_data, frame_roc_auc
    auc = save_roc_pr_curve_data(fs, np.array(labels), str(tmp_path / 'roc.npz'), verbose=False)
    obs['auc'] = abs(auc - float(g['auc']))
    _observe('scripts[%s-%s]' % (precision, golden), **obs)
    np.testing.assert_allclose(fs, g['frame_scores'], rtol=tol['frame'], atol=tol['frame'])
    # the saved maps have the reference's format (torch-pickled float64 [h,w]) and the same maxima
    m3 = torch.load(os.path.join(out_dir, '3'), weights_only=False)
    assert m3.shape == (h, w) and m3.dtype == np.float64 and abs(m3.max() - fs[3]) < 1e-12
    assert abs(auc - float(g['auc'])) <= tol['auc']
    assert abs(frame_roc_auc(fs, np.array(labels)) - auc) < 1e-12


def test_score_cubes_device_bounded_chunks_equal_one_upload():
    """test.py's scoring helper uploads the test set in bounded super-chunks through one staging buffer: chunks smaller
    than a frame, frames smaller than a launch, empty frames -- same scores, bit for bit, as one chunk holding everything."""
    from oracle import unet_oracle as O
    import test as S
    from model.unet import SelfCompleteNet4
    from vec_vad_amd.trainer import FusedTrainer
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True,
                           padding=False)
    net.load_state_dict(O.seeded_state_dict('net4', nf=32, padding=False, seed=0))
    net = net.cuda().eval()
    tr = FusedTrainer(net)
    counts = [3, 0, 9, 1, 0, 4, 7]
    cubes, flows = [], []
    for k, c in enumerate(counts):
        rw, fl = O.seeded_cubes(max(c, 1), 1, 50 + k)
        cubes.append(rw[:c])
        flows.append(fl[:c, 0])
    ref_r, ref_o = S.score_cubes_device(tr, cubes, flows, 4, chunk_cubes=10 ** 6)
    assert ref_r.shape[0] == sum(counts)
    for chunk in (5, 4, 11):
        r, o = S.score_cubes_device(tr, cubes, flows, 4, chunk_cubes=chunk)
        assert torch.equal(r, ref_r) and torch.equal(o, ref_o), chunk
    x, x_of = O.cubes_to_inputs(np.concatenate(cubes), np.concatenate(flows)[:, None])
    rs, os_ = O.score_pass(O.seeded_state_dict('net4', nf=32, padding=False, seed=0), O.bank_spec('net4'), x, x_of, sum(counts))
    np.testing.assert_allclose(ref_r.cpu().numpy(), rs, rtol=1e-3)
    np.testing.assert_allclose(ref_o.cpu().numpy(), os_, rtol=1e-3)


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
    COF.calc_optical_flow(ds, flownet2=net, log=lambda *a: None, pairs_per_launch=1)
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
    # several frames per launch (the default): same files, same flow up to fp32 round-off (split-K choices depend on the batch)
    one = {idx: np.load(os.path.join('optical_flow', 'UCSDped2', 'Train', v, '%03d.npy' % (k + 1))) for idx, (v, k) in enumerate(order)}
    COF.calc_optical_flow(ds, flownet2=net, log=lambda *a: None, of_root_dir='./optical_flow_b', pairs_per_launch=3)
    for idx, (v, k) in enumerate(order):
        got = np.load(os.path.join('optical_flow_b', 'UCSDped2', 'Train', v, '%03d.npy' % (k + 1)))
        scale = max(1e-6, float(np.abs(one[idx]).max()))
        assert got.shape == one[idx].shape and float(np.abs(got - one[idx]).max()) <= 1e-4 * scale, idx


def _synthetic_ped2_tree(rng, n_train=(4, 3), n_test=(4,)):
    """raw_datasets/ + optical_flow/ + bbox files laid out like UCSDped2 (240x360 grey .tif frames, [h,w,2] flow .npy)."""
    from PIL import Image
    H, W = 240, 360
    frames, flows, boxes = {}, {}, {}
    for mode, sub, counts in (('train', 'Train', n_train), ('test', 'Test', n_test)):
        all_boxes = []
        for v, n in enumerate(counts, start=1):
            name = '%s%03d' % (sub, v)
            os.makedirs(os.path.join('raw_datasets', 'UCSDped2', sub, name))
            os.makedirs(os.path.join('optical_flow', 'UCSDped2', sub, name))
            if mode == 'test':
                os.makedirs(os.path.join('raw_datasets', 'UCSDped2', sub, name + '_gt'))
            for k in range(n):
                g = rng.integers(0, 256, (H, W), dtype=np.uint8)
                fl = (rng.standard_normal((H, W, 2)) * 2).astype(np.float32)
                fl[:60, :90] = 0                                   # a still corner: boxes there fail the motion test
                frames.setdefault(mode, []).append(g)
                flows.setdefault(mode, []).append(fl)
                Image.fromarray(g).save(os.path.join('raw_datasets', 'UCSDped2', sub, name, '%03d.tif' % (k + 1)))
                np.save(os.path.join('optical_flow', 'UCSDped2', sub, name, '%03d.npy' % (k + 1)), fl)
                if mode == 'test':
                    gt = np.zeros((H, W), np.uint8)
                    if k % 2:
                        gt[100:120, 100:130] = 255
                    Image.fromarray(gt).save(os.path.join('raw_datasets', 'UCSDped2', sub, name + '_gt', '%03d.bmp' % (k + 1)))
                nb = int(rng.integers(0, 4)) if k else 3
                bb = []
                for m in range(nb):
                    x0, y0 = rng.uniform(95, W - 70), rng.uniform(65, H - 70)
                    bb.append([x0, y0, x0 + rng.uniform(8, 64), y0 + rng.uniform(8, 64), rng.random()])
                if k == 0:
                    bb[0] = [5.0, 4.0, 40.0, 50.0, 0.9]            # inside the still corner -> dropped (energy 0)
                    bb[1] = [200.0, 130.0, 264.0, 194.0, 0.9]      # 64x64 -> exact 2x area path
                all_boxes.append(np.array(bb).reshape(-1, 5))
        arr = np.empty(len(all_boxes), dtype=object)
        for i, b in enumerate(all_boxes):
            arr[i] = b
        np.save(os.path.join('raw_datasets', 'UCSDped2', 'bboxes_%s_obj_det_with_motion.npy' % mode), arr, allow_pickle=True)
        boxes[mode] = all_boxes
    return frames, flows, boxes


def test_extraction_stage_and_full_scripts(tmp_path, monkeypatch):
    """foreground.extract_train / extract_test (train.py:102-226, test.py:98-176) against the oracle's get_foreground,
    then ``train.main`` and ``test.main`` end to end from frames on disk to a frame-level AUC."""
    import shutil
    import foreground as FG
    import train as T
    import test as S
    import vad_datasets as V
    from oracle import resize_oracle as R
    from utils import calc_block_idx
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(11)
    frames, flows, boxes = _synthetic_ped2_tree(rng)
    cfg = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'config.cfg')).read()
    cfg = cfg.replace('epochs = 10', 'epochs = 1').replace('batch_size = 128', 'batch_size = 4')
    cfg = cfg.replace('[UCSDped2]\n', '[UCSDped2]\nh_block = 2\nw_block = 2\ntrain_block_mode = 9\n')      # dataset-level overrides
    open('config.cfg', 'w').write(cfg)
    c = T.read_config('config.cfg')
    assert c['h_block'] == 2 and c['cp'].getint('UCSDped2', 'train_block_mode') == 9 and c['batch_size'] == 4

    def expected(mode, vid_lens, block_mode):
        fvi = [v for v, n in enumerate(vid_lens) for _ in range(n)]
        cells = {}
        for idx in range(len(fvi)):
            rng_raw = V.context_range(idx, 'predict', 4, len(fvi), fvi)
            raw_stack = np.array([np.repeat(frames[mode][i][None], 3, 0) for i in rng_raw])         # [5,3,H,W]
            fl = np.array([np.transpose(flows[mode][i], [2, 0, 1]) for i in rng_raw])                # [5,2,H,W]
            bb = boxes[mode][idx]
            if len(bb) == 0:
                continue
            pr = np.transpose(R.get_foreground(raw_stack, bb, 32), [0, 1, 3, 4, 2])                  # [n,5,32,32,3]
            pf = np.transpose(R.get_foreground(fl, bb, 32), [0, 1, 3, 4, 2])                         # [n,5,32,32,2]
            mag = (pf.astype(np.float64) ** 2).sum(axis=(2, 3, 4)).mean(axis=1)
            for m in range(len(bb)):
                if mag[m] > 0:
                    for (hi, wi) in calc_block_idx(bb[m, 0], bb[m, 2], bb[m, 1], bb[m, 3], 120.0, 180.0, mode=block_mode):
                        cells.setdefault((idx, hi, wi), []).append((pr[m], pf[m], bb[m]))
        return cells

    # ---- train extraction
    FG.extract_train(c, 'cuda', log=lambda *a: None)
    fr = np.load('data/raw2flow/UCSDped2_foreground_train_obj_det_with_motion-raw.npy', allow_pickle=True)
    ff = np.load('data/raw2flow/UCSDped2_foreground_train_obj_det_with_motion-flow.npy', allow_pickle=True)
    assert fr.shape == (2, 2) and ff.shape == (2, 2)
    exp = expected('train', (4, 3), 9)
    n_tot = 0
    for hi in range(2):
        for wi in range(2):
            want = [e for (idx, h2, w2), lst in sorted(exp.items()) if (h2, w2) == (hi, wi) for e in lst]
            n_tot += len(want)
            assert len(fr[hi][wi]) == len(want) == len(ff[hi][wi])
            if want:
                assert fr[hi][wi].dtype == np.uint8 and fr[hi][wi].shape[1:] == (5, 32, 32, 3)
                assert ff[hi][wi].dtype == np.float32 and ff[hi][wi].shape[1:] == (5, 32, 32, 2)
                assert np.array_equal(fr[hi][wi], np.array([e[0] for e in want]))
                assert np.array_equal(ff[hi][wi], np.array([e[1] for e in want]))
    assert n_tot >= 8
    # the two hand-placed boxes of frame 0: one dropped by the motion test, the other present
    assert not any(np.array_equal(e[2][:4], [5.0, 4.0, 40.0, 50.0]) for lst in exp.values() for e in lst)

    # ---- test extraction (block mode 1) incl. boxes and frame labels
    FG.extract_test(c, 'cuda', log=lambda *a: None)
    tr = np.load('data/raw2flow/UCSDped2_foreground_test_obj_det_with_motion-raw.npy', allow_pickle=True)
    tf = np.load('data/raw2flow/UCSDped2_foreground_test_obj_det_with_motion-flow.npy', allow_pickle=True)
    tb = np.load('data/raw2flow/UCSDped2_foreground_bbox_test_obj_det_with_motion.npy', allow_pickle=True)
    assert tr.shape == (4, 2, 2) and tb.shape == (4, 2, 2)
    exp = expected('test', (4,), 1)
    for idx in range(4):
        for hi in range(2):
            for wi in range(2):
                want = exp.get((idx, hi, wi), [])
                assert len(tr[idx][hi][wi]) == len(want)
                if want:
                    assert np.array_equal(tr[idx][hi][wi], np.array([e[0] for e in want]))
                    assert np.array_equal(tf[idx][hi][wi], np.array([e[1] for e in want]))
                    assert np.array_equal(tb[idx][hi][wi], np.array([e[2] for e in want]))
    assert np.load('data/raw2flow/UCSDped2_frame_labels_test.npy').tolist() == [False, True, False, True]

    # ---- the scripts themselves, from frames on disk: extraction -> training -> scoring -> AUC
    shutil.rmtree('data')
    T.main('config.cfg')
    assert os.path.exists('data/raw2flow/UCSDped2_model_obj_det_with_motion_SelfComplete.npy')
    auc = S.main('config.cfg')
    assert auc is not None and 0.0 <= auc <= 1.0
    fs = np.load('results/UCSDped2/frame_scores_obj_det_with_motion_SelfComplete.npy')
    assert fs.shape == (4,) and np.isfinite(fs).all()
