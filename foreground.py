"""Foreground (cube) extraction stage shared by train.py and test.py -- reference train.py:102-226 and test.py:98-180.

For every frame: decode its temporal context (raw frames and the pre-computed optical flow ``.npy`` fields), cut every
detected box out of every context frame and resize it to ``patch_size`` (ONE ``vv_crop_resize`` launch per modality per
frame instead of one ``cv2.resize`` call per box per frame), drop boxes whose flow energy is below ``motionThr``, and file
the cubes under the grid block(s) the box falls in.  Outputs use the reference's file names and nesting so that either
implementation can consume the other's files.

The bounding boxes themselves come from ``raw_datasets/<ds>/bboxes_{train,test}_<mode>.npy`` (written by the reference's
mmdet / motion detector stage, train.py:44-99, which is outside the hot path); the trivially computable 'frame' mode is
produced here.
"""
import os

import numpy as np
import torch

from utils import calc_block_idx
from vad_datasets import frame_size, unified_dataset_interface


def save_nested(path, nested, depth):
    """np.save of a ``depth``-level nested list of arrays as an object array of exactly that nesting (what the
    reference's ``np.save(path, nested_list)`` produced under NumPy < 1.24 whenever the leaves were ragged)."""
    def shape_of(x, d):
        return () if d == 0 else (len(x),) + shape_of(x[0], d - 1)
    arr = np.empty(shape_of(nested, depth), dtype=object)
    for idx in np.ndindex(arr.shape):
        leaf = nested
        for i in idx:
            leaf = leaf[i]
        arr[idx] = np.asarray(leaf)
    # atomic: a reader (another rank, a later run) sees the previous complete file or the new complete file, never a torn one
    final = path if path.endswith('.npy') else path + '.npy'
    tmp = '%s.tmp%d' % (final, os.getpid())
    with open(tmp, 'wb') as f:
        np.save(f, arr, allow_pickle=True)
    os.replace(tmp, final)


def load_bboxes(c, mode, dataset=None):
    """train.py:96-99 / test.py:92-96.  ``<raw_dataset_dir>/<ds>/bboxes_<mode>_<fg mode>.npy``; 'frame' mode needs no
    detector and is generated when the file is absent."""
    cp, ds, fg = c['cp'], c['dataset_name'], c['mode_fg']
    path = os.path.join(c['raw_dataset_dir'], ds, 'bboxes_{}_{}.npy'.format(mode, fg))
    if cp.getboolean(ds, '{}_bbox_saved'.format(mode)) or os.path.exists(path):
        return np.load(path, allow_pickle=True)
    if fg == 'frame' and dataset is not None:
        h, w = frame_size[ds][0], frame_size[ds][1]
        boxes = [np.array([[0, 0, w, h]]) for _ in range(len(dataset))]
        np.save(path, boxes)
        return boxes
    raise NotImplementedError(
        '{}_bbox_saved = False: the object-detector / motion foreground localisation (reference train.py:44-95: mmdet '
        'cascade R-CNN + fore_det) is outside the hot path built here; produce {} with the reference once.'.format(mode, path))


def _datasets(c, mode, all_bboxes):
    cp, ds, method = c['cp'], c['dataset_name'], c['method']
    kw = dict(dataset_name=ds, mode=mode, border_mode=cp.get(method, 'border_mode'), all_bboxes=all_bboxes,
              patch_size=cp.getint(ds, 'patch_size'))
    raw = unified_dataset_interface(dir=os.path.join('raw_datasets', ds), file_format=frame_size[ds][2],
                                    context_frame_num=cp.getint(method, 'context_frame_num'), **kw)
    flow = unified_dataset_interface(dir=os.path.join('optical_flow', ds), file_format='.npy',
                                     context_frame_num=cp.getint(method, 'context_of_num'), **kw)
    return raw, flow


def frame_cubes(raw_ds, flow_ds, idx, motion_thr, device='cuda'):
    """Cubes of frame ``idx`` that pass the motion test: (raw ``[m,(T,)P,P,3]`` uint8, flow ``[m,(Tf,)P,P,2]`` float32,
    kept box indices).  Flow energy per box = sum of squares over the patch (mean over the context frames when there is a
    context), train.py:159-170."""
    raw = raw_ds.cubes_device(idx, device)                  # [n,T,P,P,C]
    flow = flow_ds.cubes_device(idx, device)                # [n,Tf,P,P,2]
    mag = (flow.double() ** 2).sum(dim=(2, 3, 4)).mean(dim=1)          # one context frame: the mean is the sum itself
    keep = torch.nonzero(mag.cpu() > motion_thr).flatten().numpy()
    raw, flow = raw.cpu().numpy(), flow.cpu().numpy()
    if raw_ds.context_frame_num == 0:
        raw = raw[:, 0]
    if flow_ds.context_frame_num == 0:
        flow = flow[:, 0]
    return raw[keep], flow[keep], keep


def extract_train(c, device='cuda', log=print):
    """train.py:102-226 for modality raw2flow.  Writes ``<root>/raw2flow/<ds>_foreground_train_<fg>-{raw,flow}.npy``
    (ShanghaiTech: ``..._seg_<k>-{raw,flow}.npy`` every ``saveSegNum`` frames, frames visited in a random order)."""
    cp, ds, fg, root, mod = c['cp'], c['dataset_name'], c['mode_fg'], c['data_root_dir'], c['modality']
    hb, wb = c['h_block'], c['w_block']
    all_bboxes = load_bboxes(c, 'train')
    raw_ds, flow_ds = _datasets(c, 'train', all_bboxes)
    h_step, w_step = frame_size[ds][0] / hb, frame_size[ds][1] / wb
    motion_thr, block_mode = cp.getfloat(ds, 'motionThr'), cp.getint(ds, 'train_block_mode')
    shanghai = ds == 'ShanghaiTech'
    os.makedirs(os.path.join(root, mod), exist_ok=True)
    base = os.path.join(root, mod, ds + '_foreground_train_{}'.format(fg))

    def empty():
        def grid():
            return [[([], []) for _ in range(wb)] for _ in range(hb)]
        return [grid() for _ in range(raw_ds.scene_num)] if shanghai else grid()

    def dump(sets, suffix):
        pick = (lambda k: [[[np.array(cell[k]) for cell in row] for row in scene] for scene in sets]) if shanghai else \
            (lambda k: [[np.array(cell[k]) for cell in row] for row in sets])
        save_nested(base + suffix + '-raw.npy', pick(0), 3 if shanghai else 2)
        save_nested(base + suffix + '-flow.npy', pick(1), 3 if shanghai else 2)

    order = np.random.default_rng(c['shuffle_seed']).permutation(len(raw_ds)) if shanghai else np.arange(len(raw_ds))
    seg_num = cp.getint(ds, 'saveSegNum') if shanghai else 0
    sets, count, seg = empty(), 0, 0
    for it, idx in enumerate(order):
        idx = int(idx)
        log('Extracting foreground in {}-th batch, {} in total'.format(it + 1, len(raw_ds)))
        boxes = all_bboxes[idx]
        if len(boxes) > 0:
            raw, flow, keep = frame_cubes(raw_ds, flow_ds, idx, motion_thr, device)
            grid = sets[raw_ds.scene_idx[idx] - 1] if shanghai else sets
            for k, b in enumerate(keep):
                bb = boxes[b]
                for (hi, wi) in calc_block_idx(bb[0], bb[2], bb[1], bb[3], h_step, w_step, mode=block_mode):
                    grid[hi][wi][0].append(raw[k])
                    grid[hi][wi][1].append(flow[k])
        count += 1
        if shanghai and count == seg_num:
            dump(sets, '_seg_{}'.format(seg))
            sets, count, seg = empty(), 0, seg + 1
    if shanghai:
        if len(raw_ds) % seg_num != 0:
            dump(sets, '_seg_{}'.format(seg))
    else:
        dump(sets, '')
    log('foreground for training data saved!')


def extract_test(c, device='cuda', log=print):
    """test.py:98-176: per-frame, per-block cubes + their boxes ->
    ``<ds>_foreground_test_<fg>-{raw,flow}.npy``, ``<ds>_foreground_bbox_test_<fg>.npy`` (+ ``<ds>_scene_idx.npy``)."""
    cp, ds, fg, root, mod = c['cp'], c['dataset_name'], c['mode_fg'], c['data_root_dir'], c['modality']
    hb, wb = c['h_block'], c['w_block']
    all_bboxes = load_bboxes(c, 'test')
    raw_ds, flow_ds = _datasets(c, 'test', all_bboxes)
    os.makedirs(os.path.join(root, mod), exist_ok=True)
    base = os.path.join(root, mod, ds + '_')
    if ds == 'ShanghaiTech':
        np.save(base + 'scene_idx.npy', raw_ds.scene_idx)
    h_step, w_step = frame_size[ds][0] / hb, frame_size[ds][1] / wb
    motion_thr, block_mode = cp.getfloat(ds, 'motionThr'), cp.getint(ds, 'test_block_mode')
    n = len(raw_ds)
    sets = [[[([], [], []) for _ in range(wb)] for _ in range(hb)] for _ in range(n)]
    for idx in range(n):
        log('Extracting foreground in {}-th batch, {} in total'.format(idx + 1, n))
        boxes = all_bboxes[idx]
        if len(boxes) > 0:
            raw, flow, keep = frame_cubes(raw_ds, flow_ds, idx, motion_thr, device)
            for k, b in enumerate(keep):
                bb = boxes[b]
                for (hi, wi) in calc_block_idx(bb[0], bb[2], bb[1], bb[3], h_step, w_step, mode=block_mode):
                    cell = sets[idx][hi][wi]
                    cell[0].append(raw[k])
                    cell[1].append(flow[k])
                    cell[2].append(bb)
    for k, name in ((0, 'foreground_test_{}-raw.npy'), (1, 'foreground_test_{}-flow.npy'), (2, 'foreground_bbox_test_{}.npy')):
        save_nested(base + name.format(fg), [[[np.array(cell[k]) for cell in row] for row in fr] for fr in sets], 3)
    # frame-level ground truth for the evaluation step (test.py:376-392 reads it through the dataset at evaluation time)
    if raw_ds.return_gt:
        labels = np.array([bool(np.asarray(raw_ds._gt(i)).max() > 0) for i in range(n)])
        np.save(base + 'frame_labels_test.npy', labels)
    log('foreground for testing data saved!')
