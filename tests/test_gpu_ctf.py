"""svae_ctf_filter (device CTF filter bank) against the filters the reference's spatial_vae/ctf.py:33-56 produced
(tests/golden/ctf_golden.npz, written by gen_golden.py --ctf) and against the numpy host form on a larger table."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR

pytestmark = pytest.mark.gpu
TOL = 2e-6       # doubles inside, one fp32 rounding at the end; relative to the largest filter tap


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


def test_device_filters_match_reference_fixture():
    from spatial_vae_amd import ops
    with np.load(os.path.join(GOLDEN_DIR, "ctf_golden.npz")) as f:
        gold = {k: f[k] for k in f.files}
    for key, (n, m, s) in {"filt_9x9_s1": (9, 9, 1), "filt_15x13_s2": (15, 13, 2), "filt_39x39_s1": (39, 39, 1)}.items():
        got = ops.ctf_filter(gold["table"], n, m, scale=s).cpu().numpy()
        assert got.shape == gold[key].shape
        assert _rel(got, gold[key]) < TOL, key


@pytest.mark.parametrize("n,m", [(39, 39), (40, 24), (79, 79)])
def test_device_filters_match_host_form(n, m):
    from spatial_vae_amd import ctf as C, ops
    rs = np.random.RandomState(n)
    P = 200
    tab = np.stack([rs.uniform(0.8, 3.5, P), np.full(P, 2.7), rs.choice([200.0, 300.0], P), rs.uniform(1.0, 2.5, P),
                    rs.uniform(0, 200, P), rs.uniform(5, 15, P), np.zeros(P), rs.uniform(0, 180, P)], 1)
    want = C.ctf_filter({k: tab[:, i] for i, k in enumerate(C.COLUMNS)}, n, m, scale=1.5)
    got = ops.ctf_filter(tab, n, m, scale=1.5).cpu().numpy()
    assert _rel(got, want) < TOL


def test_oversized_filters_are_refused():
    from spatial_vae_amd import ops
    with pytest.raises(RuntimeError, match="do not fit"):
        ops.ctf_filter(np.ones((2, 8)), 128, 128)
    with pytest.raises(RuntimeError):
        ops.ctf_filter(np.ones((2, 7)), 9, 9)
