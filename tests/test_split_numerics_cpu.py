"""The fp16x3 split-operand representation (spatial_vae_amd/csrc/split.h) emulated in numpy: its GEMM error against fp64
must be that of an fp32 GEMM (tools/split_numerics.py).  CPU only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import split_numerics as SN  # noqa: E402


def test_split_gemm_is_as_accurate_as_fp32():
    res = SN.study(M=1024, K=500, N=256, seed=1)
    for kind in ("forward", "dgrad", "wgrad"):
        e32, e16 = res[kind + " fp32"], res[kind + " fp16x3"]
        assert e16[1] <= 1.5 * e32[1], (kind, e16, e32)            # rms error
        assert e16[0] <= 3e-6, (kind, e16)                          # max error relative to the largest entry


def test_split_reconstructs_operands_to_22_bits():
    rs = np.random.RandomState(2)
    x = np.tanh(rs.normal(size=100000)).astype(np.float32)
    hi, lo = SN.split(x, 1024.0)
    back = (hi.astype(np.float64) + lo.astype(np.float64)) / 1024.0
    assert np.abs(back - x).max() <= 2.0 ** -22                     # activations in (-1, 1): absolute 2^-22
    w = (rs.uniform(-1, 1, size=100000) / np.sqrt(500)).astype(np.float32)
    s = SN.pow2_floor(8192 / np.abs(w).max())
    hi, lo = SN.split(w, s)
    back = (hi.astype(np.float64) + lo.astype(np.float64)) / s
    assert np.abs(back - w).max() <= np.abs(w).max() * 2.0 ** -21
