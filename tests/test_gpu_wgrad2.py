"""wgrad2_kernel (csrc/wgrad2.h: two waves per SIMD, operands straight into a register ring) against wgrad_kernel on the same
inputs, in one process: SVAE_WGRAD2=0 keeps every hidden-layer weight gradient on wgrad_kernel.  Both kernels run the same
m-ordered fma chain per element of dW within the same row-range splits and hand the same partial layouts to the same reduce
kernels, so every gradient must agree bit for bit -- except, in the rank-1 forms, the output layer's dW_o / db_o that the
weight-gradient kernel accumulates on the side in plain VALU code (sums of four products per octet, which hipcc contracts into
fma chains differently in the two kernels: last-ulp differences, bounded here at 1e-6 of the largest entry).  Cases: plain form (C = 2, 3; rectifiers), rank-1 tanh and
sigmoid, one to three weight gradients per step, widths that leave the last tile pair / quad partly out of range (H = 96:
3 tiles; 100: 4; 500: 16), row counts that leave a remainder of 0, 1 and 2 octets after the 3-octet ring trips."""
import numpy as np
import pytest
import torch.nn as nn

from helpers import rel_err
from test_gpu_dense4 import _run

pytestmark = pytest.mark.gpu

CASES = [
    # name, n (image side), B, z, H, L, C, act, posed
    ("tanh_c1_rank1", 8, 6, 2, 64, 2, 1, nn.Tanh, True),
    ("sigmoid_c1_rank1", 8, 6, 2, 64, 2, 1, nn.Sigmoid, True),
    ("tanh_c2_plain", 8, 6, 2, 100, 2, 2, nn.Tanh, True),
    ("tanh_c3_L3", 8, 4, 3, 64, 3, 3, nn.Tanh, True),
    ("tanh_c1_L4_rank1", 8, 5, 3, 64, 4, 1, nn.Tanh, True),          # rank-1 on the last layer, plain below it
    ("relu_c1_coords", 8, 7, 2, 64, 2, 1, nn.ReLU, False),
    ("leaky_c2_z0", 12, 8, 0, 64, 2, 2, nn.LeakyReLU, False),        # 144 pixels: Npad 160, padded octets skipped
    ("tanh_h96_three_tiles", 8, 6, 2, 96, 3, 2, nn.Tanh, True),
    ("tanh_h500", 28, 8, 2, 500, 2, 1, nn.Tanh, True),               # BASELINE width
    ("tanh_h500_b3", 28, 3, 2, 500, 2, 1, nn.Tanh, True),            # few rows: splits of 1-2 octets (no full ring trip)
    ("tanh_big_rows", 40, 9, 2, 64, 2, 2, nn.Tanh, True),            # 1600 pixels, no padding
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_wgrad2_is_bit_identical_to_wgrad_kernel(case, monkeypatch):
    monkeypatch.setenv("SVAE_WGRAD2", "0")
    old, c_old = _run(case, "1", monkeypatch)
    monkeypatch.setenv("SVAE_WGRAD2", "1")
    new, c_new = _run(case, "1", monkeypatch)
    L = case[5]
    assert c_old.get("wgrad2", 0) == 0 and c_old["wgrad_fp32"] == L - 1
    assert c_new["wgrad2"] == L - 1 and c_new["wgrad_fp32"] == L - 1, c_new
    rank1 = case[6] == 1 and case[7] in (nn.Tanh, nn.Sigmoid)
    out_layer = "g.layers.%d." % (2 * L - 1)
    for k in old:
        if rank1 and k.startswith(out_layer):
            assert rel_err(new[k], old[k]) < 1e-6, (case[0], k)
        else:
            assert np.array_equal(old[k], new[k]), (case[0], k, float(np.abs(old[k] - new[k]).max()))


def test_generic_fused_output_form_keeps_wgrad_kernel(monkeypatch):
    """SVAE_FUSE_OUT=1 with two output channels: dh is formed from all channels inside wgrad_kernel<2>; wgrad2_kernel has no
    such form and must not be dispatched for that layer."""
    monkeypatch.setenv("SVAE_FUSE_OUT", "1")
    monkeypatch.setenv("SVAE_WGRAD2", "1")
    _, c = _run(("fused_c2", 8, 6, 2, 64, 3, 2, nn.Tanh, True), "1", monkeypatch)
    assert c["wgrad_fp32"] == 2 and c["wgrad2"] == 1, c
