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
    # above ~80 x 80 the transform no longer fits the LDS and runs from a global scratch area (the reference has no limit)
    for key, (n, m, s) in {"big_129x129_s1": (129, 129, 1), "big_100x96_s2": (100, 96, 2)}.items():
        got = ops.ctf_filter(gold["table"][:2], n, m, scale=s).cpu().numpy()
        assert got.shape == gold[key].shape
        assert _rel(got, gold[key]) < TOL, key


@pytest.mark.parametrize("n,m,P", [(39, 39, 200), (40, 24, 200), (79, 79, 200), (81, 81, 9), (127, 127, 700), (255, 200, 3)])
def test_device_filters_match_host_form(n, m, P):
    """Against the numpy oracle: in-LDS sizes, the first size past the LDS limit, more particles than scratch workgroups
    (700 > 512: the grid strides), and a large non-square box."""
    from oracle import ctf_oracle as C
    from spatial_vae_amd import ops
    rs = np.random.RandomState(n)
    tab = np.stack([rs.uniform(0.8, 3.5, P), np.full(P, 2.7), rs.choice([200.0, 300.0], P), rs.uniform(1.0, 2.5, P),
                    rs.uniform(0, 200, P), rs.uniform(5, 15, P), np.zeros(P), rs.uniform(0, 180, P)], 1)
    want = C.ctf_filter({k: tab[:, i] for i, k in enumerate(C.COLUMNS)}, n, m, scale=1.5)
    got = ops.ctf_filter(tab, n, m, scale=1.5).cpu().numpy()
    assert _rel(got, want) < TOL


def test_bad_filter_arguments_are_refused():
    from spatial_vae_amd import _lib, ops
    with pytest.raises(RuntimeError):
        ops.ctf_filter(np.ones((2, 7)), 9, 9)
    L = _lib.lib()
    assert L.svae_ctf_filter_workspace_bytes(4, 39, 39) == 0
    need = L.svae_ctf_filter_workspace_bytes(4, 128, 128)
    assert need == 4 * 128 * 128 * 3 * 8
    tab = torch.ones(4, 8, dtype=torch.float64, device="cuda")
    out = torch.empty(4, 128, 128, device="cuda")
    rc = L.svae_ctf_filter(tab.data_ptr(), out.data_ptr(), 4, 128, 128, 1.0, None, 0, None)     # large filters need scratch
    assert rc == -2 and b"scratch" in L.svae_last_error()
