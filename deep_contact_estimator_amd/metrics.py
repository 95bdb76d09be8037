"""Everything the reference's src/test.py:19-70 computes with scikit-learn, as closed forms of ONE
16x16 integer matrix C[gt][pred] (accumulated on the GPU by dce_confusion_counts, added across
GPUs with one all-reduce).  Host-side numpy on 256 integers; no per-window data leaves the device.

Reference formulas restated (scikit-learn semantics):
  confusion_matrix(gt, pred, labels=[0,1])[i][j]   = #(gt = i, pred = j)
  precision_score binary                            = TP / (TP + FP)          (0 when nothing predicted)
  precision_score(average='weighted')               = sum_c support_c * precision_c / sum_c support_c
  jaccard_score binary                              = TP / (TP + FP + FN)     (0 when empty)
  jaccard_score(average='weighted')                 = support-weighted mean of the per-class scores
Leg l of class id c is bit (3 - l) of c: legs [RF, LF, RH, LH], MSB first (src/test.py:23-26).
"""
from __future__ import annotations

import numpy as np

LEG_NAMES = ("leg_rf", "leg_lf", "leg_rh", "leg_lh")


def confusion16(pred, labels) -> np.ndarray:
    """numpy fallback-free *definition* of the matrix (used by tests): C[gt][pred]."""
    p = np.asarray(pred).reshape(-1).astype(np.int64)
    g = np.asarray(labels).reshape(-1).astype(np.int64)
    return np.bincount(g * 16 + p, minlength=256).reshape(16, 16)


def _div(a, b):
    """scikit-learn's zero_division default for precision / Jaccard: 0 when nothing to divide by."""
    return float(a) / float(b) if b else 0.0


def _npdiv(a, b):
    """numpy's int / int as the reference writes it (src/test.py:28,35-45): 0/0 -> nan, x/0 -> inf."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.true_divide(a, b)


def leg_confusion(C: np.ndarray, leg: int) -> np.ndarray:
    """2x2 matrix M[gt_bit][pred_bit] of one leg."""
    bit = ((np.arange(16) >> (3 - leg)) & 1)
    M = np.zeros((2, 2), np.int64)
    for a in (0, 1):
        for b in (0, 1):
            M[a, b] = C[np.ix_(bit == a, bit == b)].sum()
    return M


def metrics_from_confusion16(C) -> dict:
    C = np.asarray(C, dtype=np.int64).reshape(16, 16)
    n = int(C.sum())
    out = {"num_data": n, "acc": float(_npdiv(np.trace(C), n))}
    # ---- compute_confusion_mat (src/test.py:19-47), including its FN/FP naming as written there
    cm, fn, fp = {}, {}, {}
    for l, name in enumerate(LEG_NAMES):
        cm[name] = leg_confusion(C, l)
    cm["total"] = sum(cm[name] for name in LEG_NAMES)
    cm["total_ratio"] = _npdiv(cm["total"], np.sum(cm["total"]))
    for name in LEG_NAMES + ("total",):
        M = cm[name]
        fn[name] = float(_npdiv(M[0, 1], M[0, 0] + M[0, 1]))     # nan on an empty denominator, like the reference
        fp[name] = float(_npdiv(M[1, 0], M[1, 0] + M[1, 1]))
    out.update(confusion_mat=cm, fn_rate=fn, fp_rate=fp)
    out["acc_per_leg"] = np.array([float(_npdiv(cm[name][0, 0] + cm[name][1, 1], n)) for name in LEG_NAMES])
    # ---- compute_precision / compute_jaccard (src/test.py:50-70)
    tp = np.diag(C).astype(np.float64)
    support = C.sum(axis=1).astype(np.float64)          # gt counts
    predicted = C.sum(axis=0).astype(np.float64)        # pred counts
    present = (support + predicted) > 0                 # sklearn's label set: union of gt and pred
    prec_c = np.array([_div(tp[c], predicted[c]) for c in range(16)])
    jac_c = np.array([_div(tp[c], support[c] + predicted[c] - tp[c]) for c in range(16)])
    wsum = support[present].sum()
    out["precision_of_class"] = _div((support * prec_c)[present].sum(), wsum)
    out["jaccard_of_class"] = _div((support * jac_c)[present].sum(), wsum)
    out["precision_of_legs"] = [_div(cm[nm][1, 1], cm[nm][1, 1] + cm[nm][0, 1]) for nm in LEG_NAMES]
    out["jaccard_of_legs"] = [_div(cm[nm][1, 1], cm[nm][1, 1] + cm[nm][0, 1] + cm[nm][1, 0]) for nm in LEG_NAMES]
    T = cm["total"]
    out["precision_of_all_legs"] = _div(T[1, 1], T[1, 1] + T[0, 1])
    out["jaccard_of_all_legs"] = _div(T[1, 1], T[1, 1] + T[0, 1] + T[1, 0])
    return out


def report_lines(mt: dict) -> list[str]:
    """The report of the reference's src/test.py:142-220, line for line (tools that parse the
    reference's stdout keep working): the labelled block, then the raw-value block."""
    cm, fn, fp = mt["confusion_mat"], mt["fn_rate"], mt["fp_rate"]
    acc_leg = mt["acc_per_leg"]
    L = ["Test accuracy in terms of class is: %.4f" % mt["acc"]]
    L += ["Accuracy of leg %d is: %.4f" % (i, acc_leg[i]) for i in range(4)]
    L += ["Accuracy is: %.4f" % (np.sum(acc_leg) / 4.0), "---------------",
          "Precision of class is: %.4f" % mt["precision_of_class"]]
    L += ["Precision of leg %d is: %.4f" % (i, mt["precision_of_legs"][i]) for i in range(4)]
    L += ["Precision of all legs is: %.4f" % mt["precision_of_all_legs"], "---------------",
          "jaccard of class is: %.4f" % mt["jaccard_of_class"]]
    L += ["jaccard of leg %d is: %.4f" % (i, mt["jaccard_of_legs"][i]) for i in range(4)]
    L += ["jaccard of all legs is: %.4f" % mt["jaccard_of_all_legs"], "---------------"]
    for nm in ("rf", "lf", "rh", "lh"):
        L += ["confusion matrix of leg %s is: " % nm, str(cm["leg_" + nm])]
    L += ["confusion matrix sum is: ", str(cm["total"]), "confusion matrix ratio: ", str(cm["total_ratio"]), "---------------"]
    L += ["false negative rate of leg %s is: %.4f" % (nm, fn["leg_" + nm]) for nm in ("rf", "lf", "rh", "lh")]
    L += ["AVG false negative rate is: %.4f" % fn["total"], "---------------"]
    L += ["false positive rate of leg %s is: %.4f" % (nm, fp["leg_" + nm]) for nm in ("rf", "lf", "rh", "lh")]
    L += ["AVG false positive rate is: %.4f" % fp["total"], "---------------"]
    raw = [mt["acc"], *acc_leg, np.sum(acc_leg) / 4.0, None, mt["precision_of_class"], *mt["precision_of_legs"],
           mt["precision_of_all_legs"], None, mt["jaccard_of_class"], *mt["jaccard_of_legs"], mt["jaccard_of_all_legs"], None,
           *[fn["leg_" + nm] for nm in ("rf", "lf", "rh", "lh")], fn["total"], None,
           *[fp["leg_" + nm] for nm in ("rf", "lf", "rh", "lh")], fp["total"]]
    L += ["---------------" if v is None else str(v) for v in raw]
    return L
