"""Block localisation and ROC/PR evaluation (reference: utils.py:5-26 calc_block_idx, utils.py:29-65
save_roc_pr_curve_data).  CPU-side, O(#frames); kept so that train.py / test.py are drop-in."""
import numpy as np


def calc_block_idx(x_min, x_max, y_min, y_max, h_step, w_step, mode):
    """Cells of the h_block x w_block grid touched by a bbox: its centre (mode 1), plus edge mid-points (mode > 1),
    plus corners (mode >= 9).  Each probe point is averaged with the centre before binning (reference utils.py:8-22)."""
    cy, cx = (y_min + y_max) / 2.0, (x_min + x_max) / 2.0
    pts = [(cy, cx)]
    if mode > 1:
        pts += [(y_min, cx), (y_max, cx), (cy, x_min), (cy, x_max)]
    if mode >= 9:
        pts += [(y_min, x_min), (y_max, x_max), (y_max, x_min), (y_min, x_max)]
    cells = set()
    for (py, px) in pts:
        cells.add((int(((py + cy) / 2.0) / h_step), int(((px + cx) / 2.0) / w_step)))
    return list(cells)


def frame_roc_auc(scores, labels):
    """Tie-aware ROC-AUC (what sklearn.metrics.roc_curve + auc return), without the sklearn dependency."""
    scores = np.asarray(scores, dtype=np.float64).ravel()
    labels = np.asarray(labels).ravel().astype(bool)
    pos, neg = scores[labels], scores[~labels]
    if len(pos) == 0 or len(neg) == 0:
        return float('nan')
    allv = np.concatenate([neg, pos])
    order = np.argsort(allv, kind='mergesort')
    sv = allv[order]
    # average ranks over ties
    _, inv, cnt = np.unique(sv, return_inverse=True, return_counts=True)
    ends = np.cumsum(cnt)
    avg = ends - (cnt - 1) / 2.0
    ranks = np.empty(len(sv))
    ranks[order] = avg[inv]
    rp = ranks[len(neg):].sum()
    return float((rp - len(pos) * (len(pos) + 1) / 2.0) / (len(pos) * len(neg)))


def save_roc_pr_curve_data(scores, labels, file_path, verbose=True):
    """Same outputs / file layout as the reference when scikit-learn is importable (utils.py:29-65); without it only the
    ROC-AUC (rank statistic) is computed and stored."""
    s = np.asarray(scores).ravel()
    y = np.asarray(labels).ravel()
    pos, neg = s[y == 1], s[y != 1]
    out = {'preds': np.concatenate((neg, pos)),                       # negatives first, like the reference's file
           'truth': np.concatenate((np.zeros_like(neg), np.ones_like(pos)))}
    try:
        from sklearn import metrics
    except ImportError:
        out['roc_auc'] = frame_roc_auc(out['preds'], out['truth'] > 0)
        if verbose:
            print('AUC@ROC is {}'.format(out['roc_auc']))
        np.savez_compressed(file_path, **out)
        return out['roc_auc']
    out['fpr'], out['tpr'], out['roc_thresholds'] = metrics.roc_curve(out['truth'], out['preds'])
    out['roc_auc'] = metrics.auc(out['fpr'], out['tpr'])
    miss = 1 - out['tpr']
    k = np.nanargmin(np.absolute(miss - out['fpr']))                  # equal-error point
    # precision/recall with "normal" as the positive class, then with "anomaly" as the positive class
    for tag, sc, kw in (('norm', out['preds'], {}), ('anom', -out['preds'], {'pos_label': 0})):
        pr, rc, th = metrics.precision_recall_curve(out['truth'], sc, **kw)
        out['precision_' + tag], out['recall_' + tag], out['pr_thresholds_' + tag] = pr, rc, th
        out['pr_auc_' + tag] = metrics.auc(rc, pr)
    if verbose:
        print('AUC@ROC is {}'.format(out['roc_auc']), 'EER1 is {}'.format(out['fpr'][k]), 'EER2 is {}'.format(miss[k]))
    np.savez_compressed(file_path, **out)
    return out['roc_auc']
