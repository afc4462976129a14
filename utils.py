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
    scores = np.asarray(scores).flatten()
    labels = np.asarray(labels).flatten()
    scores_pos, scores_neg = scores[labels == 1], scores[labels != 1]
    truth = np.concatenate((np.zeros_like(scores_neg), np.ones_like(scores_pos)))
    preds = np.concatenate((scores_neg, scores_pos))
    try:
        from sklearn.metrics import roc_curve, precision_recall_curve, auc
    except ImportError:
        roc_auc = frame_roc_auc(preds, truth > 0)
        if verbose:
            print('AUC@ROC is {}'.format(roc_auc))
        np.savez_compressed(file_path, preds=preds, truth=truth, roc_auc=roc_auc)
        return roc_auc
    fpr, tpr, roc_thresholds = roc_curve(truth, preds)
    roc_auc = auc(fpr, tpr)
    fnr = 1 - tpr
    k = np.nanargmin(np.absolute(fnr - fpr))
    eer1, eer2 = fpr[k], fnr[k]
    precision_norm, recall_norm, pr_thresholds_norm = precision_recall_curve(truth, preds)
    pr_auc_norm = auc(recall_norm, precision_norm)
    precision_anom, recall_anom, pr_thresholds_anom = precision_recall_curve(truth, -preds, pos_label=0)
    pr_auc_anom = auc(recall_anom, precision_anom)
    if verbose:
        print('AUC@ROC is {}'.format(roc_auc), 'EER1 is {}'.format(eer1), 'EER2 is {}'.format(eer2))
    np.savez_compressed(file_path, preds=preds, truth=truth, fpr=fpr, tpr=tpr, roc_thresholds=roc_thresholds,
                        roc_auc=roc_auc, precision_norm=precision_norm, recall_norm=recall_norm,
                        pr_thresholds_norm=pr_thresholds_norm, pr_auc_norm=pr_auc_norm, precision_anom=precision_anom,
                        recall_anom=recall_anom, pr_thresholds_anom=pr_thresholds_anom, pr_auc_anom=pr_auc_anom)
    return roc_auc
