"""Seeded shape sweep over the kernel alternatives: every case runs the decoder forward + backward five times in one
process -- the r01/r02 kernels (SVAE_DENSE4=0, SVAE_WGRAD2=0), the defaults (dense4_kernel / dense4_dual_kernel where the
plan takes them, wgrad2_kernel, the merged tail launch), the defaults with the tail launch split, and dense4_kernel forced at
either block width -- and compares all outputs.  The ISA
check of the hand-issued loads (tools/check_asm_loads.py) is static; this is its dynamic counterpart: shapes nobody picked by
hand (ragged tile counts, 1-3 octet row ranges per split, every activation, 1-4 layers, 1-3 channels, posed / explicit
coordinates, z_dim 0) must give the same numbers whichever kernels run."""
import os

import numpy as np
import pytest
import torch.nn as nn

from helpers import rel_err
from test_gpu_dense4 import _run

pytestmark = pytest.mark.gpu


def _cases(count, seed, big=False):
    rs = np.random.RandomState(seed)
    acts = [nn.Tanh, nn.Tanh, nn.Sigmoid, nn.ReLU, nn.LeakyReLU]
    out = []
    for i in range(count):
        n = int(rs.choice([5, 8, 11, 12, 16, 20, 28]))
        B = int(rs.randint(1, 24))
        zd = int(rs.choice([0, 1, 2, 5]))
        H = int(rs.choice([24, 32, 64, 96, 100, 128, 200, 500]))
        if H >= 200:
            B = min(B, 6)
        L = int(rs.randint(1, 5))
        if big:      # one-off soak at sizes where the wide / dual launches and long register-ring trips are what runs
            n = int(rs.choice([28, 32, 40]))
            B = int(rs.randint(24, 200))
            H = int(rs.choice([200, 256, 500, 512]))
            L = int(rs.randint(2, 4))
        C = int(rs.randint(1, 4))
        act = acts[int(rs.randint(len(acts)))]
        posed = bool(rs.randint(2))
        out.append(("fuzz%02d_n%d_B%d_z%d_H%d_L%d_C%d_%s_%s" % (i, n, B, zd, H, L, C, act.__name__, "posed" if posed else "coords"),
                    n, B, zd, H, L, C, act, posed))
    return out


# SVAE_FUZZ_COUNT widens the sweep for a one-off soak run (profiles/r03_kernel_fuzz_soak.txt: 400 geometries)
CASES = _cases(int(os.environ.get("SVAE_FUZZ_COUNT", "28")), 20261005, big=os.environ.get("SVAE_FUZZ_BIG") == "1")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_every_kernel_choice_gives_the_same_numbers(case, monkeypatch):
    monkeypatch.setenv("SVAE_WGRAD2", "0")
    monkeypatch.setenv("SVAE_TAIL_MERGE", "0")
    old, c_old = _run(case, "0", monkeypatch)
    assert c_old.get("dense4", 0) == 0 and c_old.get("wgrad2", 0) == 0
    monkeypatch.delenv("SVAE_WGRAD2")
    monkeypatch.delenv("SVAE_TAIL_MERGE")
    monkeypatch.delenv("SVAE_DENSE4", raising=False)
    runs = {}
    # (SVAE_DENSE4, SVAE_TAIL_MERGE): the default dispatch with and without the merged tail launch, then dense4_kernel forced
    # at either block width wherever it is legal (small shapes take dense_kernel by default)
    for d4, merge in (("", "1"), ("", "0"), ("1", "1"), ("2", "1")):
        monkeypatch.setenv("SVAE_TAIL_MERGE", merge)
        os.environ.pop("SVAE_DENSE4", None)
        new, c_new = _run_default(case, monkeypatch) if d4 == "" else _run(case, d4, monkeypatch)
        runs[(d4, merge)] = new
        L = case[5]
        if L >= 2:
            assert c_new.get("wgrad2", 0) == L - 1, c_new
        tol = 1e-5      # blocked fp32 sums over up to 12 k rows in different orders (the goldens allow 2e-5)
        for k in old:
            assert rel_err(new[k], old[k]) < tol, (case[0], d4, merge, k, rel_err(new[k], old[k]))
    for k in runs[("", "1")]:   # where the split-K reduction runs does not change a bit
        assert np.array_equal(runs[("", "1")][k], runs[("", "0")][k]), (case[0], k)


def _run_default(case, monkeypatch):
    """_run sets SVAE_DENSE4 to a mode; the default dispatch is the unset variable."""
    import test_gpu_dense4 as t

    class _NoDense4(object):
        def setenv(self, k, v):
            if k != "SVAE_DENSE4":
                monkeypatch.setenv(k, v)

    return t._run(case, "", _NoDense4())
