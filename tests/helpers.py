"""Shared helpers for the parity tests."""
import os

import numpy as np

import cases as C

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def rel_err(a, b):
    """max |a-b| / max(|b|) -- 'relative to the largest reference entry', the measure the
    1e-4 parity target of BASELINE.json is read in (logits cross zero, so an
    element-wise relative error is meaningless there)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)


def check_grads(prefix, got, gold, case, tol, names=None):
    """Compare a dict of parameter gradients with the golden entries '<prefix><name>'
    (full arrays, or '@sample' + '@norm' for store='sampled' cases)."""
    worst = 0.0
    for name in (names if names is not None else got.keys()):
        g = np.asarray(got[name], np.float32)
        key = prefix + name
        if key in gold:
            e = rel_err(g, gold[key])
        else:
            idx = C.sample_idx(g.size)
            e = rel_err(g.reshape(-1)[idx], gold[key + "@sample"])
            nrm = float(np.linalg.norm(g.astype(np.float64)))
            e = max(e, abs(nrm - float(gold[key + "@norm"])) / max(float(gold[key + "@norm"]), 1e-30))
        assert e < tol, "%s: grad %s rel err %.3e >= %.1e" % (case["name"], name, e, tol)
        worst = max(worst, e)
    return worst
