#!/usr/bin/env python
"""``python test.py`` -- abnormal-event detection stage of VEC_VAD on the MI355X UNet-bank engine.

Drop-in for the reference's test.py:193-399: loads ``<ds>_model_<mode>_SelfComplete.npy`` and the training-score files
written by train.py (or by the reference: same torch-pickle layout and ``module.``-prefixed keys), scores every test
cube in eval mode, z-normalises with the training-score mean / population std (test.py:260-266,338-345), paints the
scores into the bbox rectangles and max-combines them into the per-frame map ``results/<ds>/score_mask/<frame>``
(test.py:350-358), then evaluates frame-level ROC-AUC (test.py:362-399, utils.py:29-65).

Differences on purpose: the reference runs one tiny batch per frame (1-30 cubes); eval-mode BatchNorm makes scores
batch independent, so many frames are scored per launch (``[mi355x] score_batch``).  Ground-truth frame labels come from
``<data_root>/<modality>/<ds>_frame_labels_test.npy`` (bool per frame; write it once from
``unified_dataset_interface(..., mode='test')`` targets, test.py:366-392); without that file the evaluation step is
skipped.  Cube errors never leave HBM between the UNet bank and the frame score (``vv_frame_scores``); the h x w masks are
only painted when ``[mi355x] save_score_masks`` asks for the reference's ``score_mask/<frame>`` files.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from train import read_config, build_network  # noqa: E402
from utils import save_roc_pr_curve_data  # noqa: E402
from vad_datasets import frame_size  # noqa: E402
from vec_vad_amd.trainer import FusedTrainer  # noqa: E402
from vec_vad_amd import scoring  # noqa: E402

BIG = 100000


def load_model(net, state_dict, device):
    """test.py:255-257: the saved keys carry DataParallel's 'module.' prefix."""
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}
    net.load_state_dict(sd)
    net.to(device)
    net.eval()
    return net


def load_artifacts(base, fg, method, shanghai, build_net, device, h_block=1, w_block=1):
    """test.py:230-266: the three files train.py wrote (torch pickles despite the .npy suffix, train.py:432-436) -> (net_set,
    raw statistics, flow statistics) with the reference's nesting -- [h][w] for UCSDped2 / avenue, [scene][h][w] for ShanghaiTech;
    a block without a trained model is an empty list (train.py:370 skips blocks with <= 1 cube).  Works on files written by the
    reference itself (tests/test_ref_written_files.py): 'module.'-prefixed keys, aliased storages, int64 num_batches_tracked."""
    weights = torch.load(base + 'model_{}_{}.npy'.format(fg, method), map_location='cpu', weights_only=False)
    raw_tr = torch.load(base + 'raw_training_scores_{}_{}.npy'.format(fg, method), weights_only=False)
    of_tr = torch.load(base + 'of_training_scores_{}_{}.npy'.format(fg, method), weights_only=False)

    def build(wl):
        return [load_model(build_net(), wl[0], device)] if len(wl) > 0 else []

    def stat(a):
        a = np.asarray(a)
        return (np.mean(a), np.std(a)) if a.size else (0.0, 1.0)      # population std, test.py:264-266

    if shanghai:
        net_set = [[[build(weights[s][hh][ww]) for ww in range(len(weights[s][hh]))] for hh in range(len(weights[s]))]
                   for s in range(len(weights))]
        # (the statistics' nesting follows the FILE like the networks' does: h_block / w_block are accepted for compatibility only)
        st_r = [[[stat(raw_tr[s][hh][ww]) for ww in range(len(weights[s][hh]))] for hh in range(len(weights[s]))]
                for s in range(len(weights))]
        st_o = [[[stat(of_tr[s][hh][ww]) for ww in range(len(weights[s][hh]))] for hh in range(len(weights[s]))]
                for s in range(len(weights))]
    else:
        net_set = [[build(weights[hh][ww]) for ww in range(len(weights[hh]))] for hh in range(len(weights))]
        st_r = [[stat(raw_tr[hh][ww]) for ww in range(len(weights[hh]))] for hh in range(len(weights))]
        st_o = [[stat(of_tr[hh][ww]) for ww in range(len(weights[hh]))] for hh in range(len(weights))]
    return net_set, st_r, st_o


def score_cubes_device(trainer, cube_list, flow_list, score_batch, chunk_cubes=None):
    """cube_list / flow_list: per-frame arrays [n_i,5,32,32,3] uint8 / [n_i,(Tf,)32,32,2] fp32 (n_i may be 0).
    The cubes go to the GPU in bounded super-chunks (``chunk_cubes``, default 32 launches' worth, >= 4096: ~55 KB per cube for the
    5raw+5of bank, so host and device staging stay at a few hundred MB whatever the size of the test set) through ONE fixed device
    staging buffer, and are scored in launches of exactly ``score_batch`` cubes (the tail launch re-scores the chunk's last cube
    as padding: eval-mode scores do not depend on the batch) -- one workspace, one launch plan and one captured hipGraph serve
    the whole test set.  Returns the DEVICE tensors (raw [n], of [n] | None) of all cubes in frame order -- they feed
    vv_frame_scores without visiting the host."""
    dev = trainer.bank.device
    keep = [k for k in range(len(cube_list)) if len(cube_list[k])]
    if not keep:
        return torch.zeros(0, device=dev), None
    counts = [len(cube_list[k]) for k in keep]
    n = int(sum(counts))
    B = int(min(score_batch, n)) if n < score_batch else int(score_batch)
    cap = int(chunk_cubes) if chunk_cubes else max(4096, 32 * B)
    cap = max(B, min(cap, n))

    def prep(k):
        raw = np.asarray(cube_list[k])
        flow = np.asarray(flow_list[k], dtype=np.float32)
        if raw.dtype != np.uint8:
            raise TypeError('foreground cubes must be uint8 (the -raw.npy files of train.py:218-222), got %s' % raw.dtype)
        return (raw[:, None] if raw.ndim == 4 else raw), (flow[:, None] if flow.ndim == 4 else flow)

    r0, f0 = prep(keep[0])
    rawd = torch.empty((cap,) + r0.shape[1:], dtype=torch.uint8, device=dev)
    flowd = torch.empty((cap,) + f0.shape[1:], dtype=torch.float32, device=dev)
    r_all = torch.empty(n, device=dev)
    o_all = None
    done = 0
    pend_r, pend_f, pend_n = [], [], 0

    def flush():
        nonlocal done, pend_r, pend_f, pend_n, o_all
        m_tot = pend_n
        if not m_tot:
            return
        rawd[:m_tot].copy_(torch.from_numpy(np.ascontiguousarray(np.concatenate(pend_r))))
        flowd[:m_tot].copy_(torch.from_numpy(np.ascontiguousarray(np.concatenate(pend_f))))
        for s0 in range(0, m_tot, B):
            idx = torch.arange(s0, s0 + B, device=dev).clamp_(max=m_tot - 1)
            r, o = trainer.score_cubes(rawd, flowd, idx)
            m = min(B, m_tot - s0)
            r_all[done + s0:done + s0 + m] = r[:m]
            if o is not None:
                if o_all is None:
                    o_all = torch.empty(n, device=dev)
                o_all[done + s0:done + s0 + m] = o[:m]
        done += m_tot
        pend_r, pend_f, pend_n = [], [], 0

    for k in keep:
        raw, flow = prep(k)
        p = 0
        while p < len(raw):              # a single frame may hold more cubes than a chunk
            take = min(len(raw) - p, cap - pend_n)
            pend_r.append(raw[p:p + take])
            pend_f.append(flow[p:p + take])
            pend_n += take
            p += take
            if pend_n == cap:
                flush()
    flush()
    return r_all, o_all


def score_cubes_batched(trainer, cube_list, flow_list, score_batch):
    """Host view of ``score_cubes_device``: per-frame (raw_scores [n_i], of_scores [n_i] | None) numpy arrays."""
    r, o = score_cubes_device(trainer, cube_list, flow_list, score_batch)
    r = r.cpu().numpy()
    o = o.cpu().numpy() if o is not None else None
    out, p = [], 0
    for c in cube_list:
        n = len(c)
        out.append((r[p:p + n], o[p:p + n] if o is not None else None))
        p += n
    return out


def paint_frame(scores, bboxes, h, w):
    """test.py:350-357: each cube's score fills its (ceil'ed) bbox; maps are max-combined; untouched pixels = -1e5."""
    res = -1.0 * np.ones((h, w)) * BIG
    for m in range(len(scores)):
        bb = bboxes[m]
        x_min, x_max = int(np.ceil(bb[0])), int(np.ceil(bb[2]))
        y_min, y_max = int(np.ceil(bb[1])), int(np.ceil(bb[3]))
        region = res[y_min:y_max, x_min:x_max]
        np.maximum(region, scores[m], out=region)
    return res


def score_frames(net_set, stats_raw, stats_of, foreground_set, foreground_set2, bbox_set, h, w, w_raw, w_of, useFlow,
                 device, score_batch=2048, scene_idx=None, result_dir=None, log=print, return_device=False):
    """Per-frame anomaly scores.  ``net_set[(s,)hh][ww]`` is a list with 0 or 1 eval-mode networks;
    ``stats_*[(s,)hh][ww]`` = (mean, std) of the training scores.

    The per-cube errors stay in HBM: ``vv_frame_scores`` z-normalises, weights and max-reduces them per frame
    (= the maximum of the reference's painted h x w mask, test.py:350-357,391).  Only when ``result_dir`` is given are the
    masks themselves painted (on the host) and saved as ``<result_dir>/<frame>`` like the reference does."""
    n_frames = len(foreground_set)
    mask_groups = [] if result_dir else None      # per scored group: (frame -> slice, host cube scores, boxes); masks are painted one frame at a time
    fs_dev = torch.full((n_frames,), -float(BIG), dtype=torch.float64, device=device)
    hb, wb = len(foreground_set[0]), len(foreground_set[0][0])
    trainers = {}
    keys = sorted(set(scene_idx[f] - 1 for f in range(n_frames))) if scene_idx is not None else [None]
    for hh in range(hb):
        for ww in range(wb):
            for key in keys:      # frames are grouped by the model that scores them (one per scene for ShanghaiTech)
                frames = [f for f in range(n_frames) if scene_idx is None or scene_idx[f] - 1 == key]
                counts = np.zeros(n_frames, np.int64)
                for f in frames:
                    counts[f] = len(foreground_set[f][hh][ww])
                if counts.sum() == 0:
                    continue
                frames = [f for f in frames if counts[f]]
                models = net_set[key][hh][ww] if key is not None else net_set[hh][ww]
                boxes = np.concatenate([np.asarray(bbox_set[f][hh][ww], dtype=np.float64)[:, :4] for f in frames])
                off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
                n = int(off[-1])
                if len(models) > 0:
                    net = models[0]
                    if id(net) not in trainers:
                        trainers[id(net)] = FusedTrainer(net)
                    st_r = stats_raw[key][hh][ww] if key is not None else stats_raw[hh][ww]
                    st_o = (stats_of[key][hh][ww] if key is not None else stats_of[hh][ww]) if useFlow else (0.0, 1.0)
                    r, o = score_cubes_device(trainers[id(net)], [foreground_set[f][hh][ww] for f in frames],
                                              [foreground_set2[f][hh][ww] for f in frames], score_batch)
                    o = o if useFlow else None
                    stats = np.array([[st_r[0], st_r[1], st_o[0], st_o[1]]], np.float64)
                    cube_stat = np.zeros(n, np.int32)
                else:        # anomaly: no object in the training set in this block (test.py:346-348)
                    r, o = torch.zeros(n, device=device), None
                    stats = np.array([[0.0, 1.0, 0.0, 1.0]])
                    cube_stat = np.full(n, -1, np.int32)
                scoring.frame_scores(r, o, off, cube_stat, stats, scoring.box_paints(boxes, h, w), w_raw, w_of, out=fs_dev)
                if mask_groups is not None:
                    if len(models) > 0:
                        sc = w_raw * ((r.cpu().numpy().astype(np.float32) - stats[0, 0]) / stats[0, 1])
                        if o is not None:
                            sc = sc + w_of * ((o.cpu().numpy().astype(np.float32) - stats[0, 2]) / stats[0, 3])
                    else:
                        sc = np.ones(n) * BIG
                    mask_groups.append((off, sc, boxes))           # off is indexed by frame: cubes of frame f = [off[f], off[f+1])
    if result_dir:
        os.makedirs(result_dir, exist_ok=True)
        for f in range(n_frames):       # one h x w float64 mask alive at a time, like the reference (test.py:350-358)
            fmap = -1.0 * np.ones((h, w)) * BIG
            for off, sc, boxes in mask_groups:
                if off[f + 1] > off[f]:
                    sl = slice(off[f], off[f + 1])
                    np.maximum(fmap, paint_frame(sc[sl], boxes[sl], h, w), out=fmap)
            torch.save(fmap, os.path.join(result_dir, '{}'.format(f)))
    return fs_dev if return_device else fs_dev.cpu().numpy()


def main(config_path='config.cfg'):
    c = read_config(config_path)
    cp, ds, fg, root, mod, method = c['cp'], c['dataset_name'], c['mode_fg'], c['data_root_dir'], c['modality'], c['method']
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(device)
    if not cp.getboolean(ds, 'test_foreground_saved') and not cp.getboolean(ds, 'scores_saved'):   # test.py:98-176
        from foreground import extract_test
        extract_test(c, device)
    base = os.path.join(root, mod, ds + '_')
    h, w, _, _ = frame_size[ds]
    results_dir = 'results'
    shanghai = ds == 'ShanghaiTech'
    frame_scores_path = os.path.join(results_dir, ds, 'frame_scores_{}_{}.npy'.format(fg, method))
    if not cp.getboolean(ds, 'scores_saved'):
        fset = np.load(base + 'foreground_test_{}-raw.npy'.format(fg), allow_pickle=True)
        fset2 = np.load(base + 'foreground_test_{}-flow.npy'.format(fg), allow_pickle=True)
        bset = np.load(base + 'foreground_bbox_test_{}.npy'.format(fg), allow_pickle=True)
        scene_idx = np.load(base + 'scene_idx.npy') if shanghai else None
        net_set, st_r, st_o = load_artifacts(base, fg, method, shanghai, lambda: build_network(c), device, c['h_block'], c['w_block'])
        mask_dir = os.path.join(results_dir, ds, 'score_mask') if c['save_score_masks'] else None
        fs = score_frames(net_set, st_r, st_o, fset, fset2, bset, h, w, c['w_raw'], c['w_of'], c['useFlow'], device,
                          c['score_batch'], scene_idx, mask_dir)
        os.makedirs(os.path.join(results_dir, ds), exist_ok=True)
        np.save(frame_scores_path, fs)
    else:
        fs = np.load(frame_scores_path)

    # ---- evaluation (test.py:362-399), criterion = 'frame'
    lab_path = base + 'frame_labels_test.npy'
    if not os.path.exists(lab_path):
        print('no {} -> frame-level evaluation skipped (ground-truth masks need the dataset frame indexers)'.format(lab_path))
        return None
    labels = np.load(lab_path).astype(bool)
    print('Evaluating {} by frame-criterion:'.format(ds))
    if shanghai:
        scene_idx = np.load(base + 'scene_idx.npy')
        aucs = []
        for si in sorted(set(scene_idx)):
            sel = scene_idx == si
            aucs.append(save_roc_pr_curve_data(fs[sel], labels[sel], os.path.join(
                results_dir, ds, '{}_{}_{}_frame_results_scene_{}.npz'.format(mod, fg, method, si))))
        auc = float(np.mean(aucs))
        print('Average frame-level AUC is {}'.format(auc))
    else:
        path = os.path.join(results_dir, ds, '{}_{}_{}_frame_results.npz'.format(mod, fg, method))
        print('Results written to {}:'.format(path))
        auc = save_roc_pr_curve_data(fs, labels, path)
        # the same number from the device-side pair count (vv_roc_auc_counts); the .npz above keeps the reference's layout
        auc_dev = scoring.roc_auc(torch.from_numpy(np.asarray(fs, np.float64)).to(device), labels)
        print('AUC@ROC (device pair count) is {}'.format(auc_dev))
    return auc


if __name__ == '__main__':
    main()
