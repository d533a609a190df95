"""dense4_kernel (csrc/dense4.h: four row tiles per wave, one 16-byte row-operand load per k-step) against dense_kernel on the
same inputs, in one process: SVAE_DENSE4=0 keeps the hidden-layer GEMMs on dense_kernel, =1 takes dense4_kernel wherever it is
legal (row space in whole 128-row groups, width in whole 64-column blocks, no residual).  Every variant the decoder
dispatches is covered -- plain forward, forward with the output layer's logits in the epilogue (1, 2, 3 channels), plain data
gradient (L = 3), data gradient into the coordinate layer (FIRST) with and without the rank-1 output-layer form for tanh
(LASTD 2) and sigmoid (LASTD 3), explicit coordinates and posed grids -- the dispatch is read back (svae_path_counts), the
forward activations must agree BIT FOR BIT (same k-ordered fma chain per element), everything that passes through a
differently-blocked fixed-order sum (logits, gradients) to 2e-6 of the largest entry.  The golden / oracle / full-size suites
run the default dispatch, which takes dense4_kernel at the BASELINE sizes of configs 2-5."""
import contextlib
import io

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # name, n (image side), B, z, H, L, C, act, posed
    ("tanh_c1_rank1", 8, 6, 2, 64, 2, 1, nn.Tanh, True),          # CF C=1, FIRST + LASTD 2
    ("sigmoid_c1_rank1", 8, 6, 2, 64, 2, 1, nn.Sigmoid, True),    # FIRST + LASTD 3
    ("tanh_c2_stream", 8, 6, 2, 100, 2, 2, nn.Tanh, True),        # CF C=2, FIRST without LASTD (streaming out_bwd), Hp 128
    ("tanh_c3_L3", 8, 4, 3, 64, 3, 3, nn.Tanh, True),             # plain fwd, CF C=3, plain dgrad, FIRST
    ("tanh_c1_L3_rank1", 8, 4, 3, 64, 3, 1, nn.Tanh, True),       # LASTD 2 on a non-first layer
    ("relu_c1_coords", 8, 6, 2, 64, 2, 1, nn.ReLU, False),        # explicit coordinates (forward(x, z)), rectifier
    ("leaky_c2_coords", 12, 8, 0, 64, 2, 2, nn.LeakyReLU, False), # z_dim 0, 144 pixels (Npad 160: 5 tiles per image)
    ("tanh_h500", 28, 8, 2, 500, 2, 1, nn.Tanh, True),            # BASELINE width, 25 tiles per image
    ("tanh_h96_odd_tiles", 8, 6, 2, 96, 3, 2, nn.Tanh, True),     # three column tiles: NT = 2 is not legal, NT = 1 is
]


def _run(case, mode, monkeypatch):
    import spatial_vae.models as models
    from spatial_vae_amd import _lib
    name, n, B, zd, H, L, C, act, posed = case
    monkeypatch.setenv("SVAE_DENSE4", mode)
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(zd, H, n_out=C, num_layers=L, activation=act).to(dev)
    N = n * n
    g = torch.Generator().manual_seed(11)
    z = torch.randn(B, max(zd, 1), generator=g)[:, :zd].to(dev).requires_grad_(zd > 0)
    dy = (torch.randn(B, N, C, generator=g) / N).to(dev)
    _lib.path_counts(reset=True)
    if posed:
        x0, x1 = np.meshgrid(np.linspace(-1, 1, n), np.linspace(1, -1, n))
        grid = torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)).to(dev)
        theta = torch.randn(B, generator=g).to(dev).requires_grad_(True)
        dx = (0.1 * torch.randn(B, 2, generator=g)).to(dev).requires_grad_(True)
        y, logits = p.forward_posed(grid, B, theta=theta, dx=dx, z=z if zd > 0 else None, return_logits=True)
        extra = [theta, dx]
    else:
        x = (torch.rand(B, N, 2, generator=g) * 2 - 1).to(dev).requires_grad_(True)
        y = p(x, z if zd > 0 else torch.zeros(B, 0, device=dev))
        logits = y
        extra = [x]
    y.backward(dy)
    torch.cuda.synchronize()
    counts = _lib.path_counts()
    out = {"y": y.detach().cpu().numpy(), "logits": logits.detach().cpu().numpy()}
    for k, q in p.named_parameters():
        out["g." + k] = q.grad.detach().cpu().numpy()
    for i, t in enumerate(extra + ([z] if zd > 0 else [])):
        out["gin.%d" % i] = t.grad.detach().cpu().numpy()
    return out, counts


@pytest.mark.parametrize("mode", ["1", "2"], ids=["nt1", "nt2"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_dense4_matches_dense_kernel(case, mode, monkeypatch):
    """SVAE_DENSE4=1: 32-column blocks, three waves per SIMD (the form small launches take); =2: 64-column blocks, two waves
    per SIMD (the form the BASELINE sizes of configs 2-5 take; an odd number of column tiles falls back to 1)."""
    old, c_old = _run(case, "0", monkeypatch)
    new, c_new = _run(case, mode, monkeypatch)
    L = case[5]
    assert c_old.get("dense4", 0) == 0
    assert c_new["dense4"] == 2 * (L - 1), c_new                     # every hidden-layer GEMM, forward and data gradient
    for k in old:
        e = rel_err(new[k], old[k])
        assert e < (1e-5 if case[4] >= 500 else 2e-6), (case[0], k, e)   # H = 500: sums over 6272 rows of differently-rounded logits


def test_dense4_forward_activations_are_bit_identical(monkeypatch):
    """The plain forward stores the same k-ordered fma chain per element: a 3-layer decoder's y can differ between the two
    kernels only through the output layer's differently-blocked partial sums, so compare an L = 2, C = 3 decoder whose logits
    come from out_fwd_kernel (SVAE_FUSE_LOGITS=0: one pass over the stored activations, identical in both runs)."""
    monkeypatch.setenv("SVAE_FUSE_LOGITS", "0")
    case = ("bits", 8, 6, 2, 64, 2, 3, nn.Tanh, True)
    old, _ = _run(case, "0", monkeypatch)
    for mode in ("1", "2"):
        new, c_new = _run(case, mode, monkeypatch)
        assert c_new["dense4"] == 2
        assert np.array_equal(old["y"], new["y"]) and np.array_equal(old["logits"], new["logits"]), mode


TAIL_CASES = [
    # 20 images of 64 padded rows = 1280 rows = 2.5 sets of 512 rows: the forced split leaves 1 set wide, 1.5 sets to the tail
    ("tail_tanh_c1_rank1", 8, 20, 2, 64, 2, 1, nn.Tanh, True),
    ("tail_tanh_c2_stream", 8, 20, 2, 100, 2, 2, nn.Tanh, True),
    ("tail_tanh_c3_L3", 8, 20, 3, 64, 3, 3, nn.Tanh, True),
    ("tail_relu_coords", 8, 20, 2, 64, 2, 1, nn.ReLU, False),
    ("tail_tanh_h500", 28, 8, 2, 500, 2, 1, nn.Tanh, True),           # 6400 rows, the split falls inside an image (800 rows)
]


@pytest.mark.parametrize("case", TAIL_CASES, ids=[c[0] for c in TAIL_CASES])
def test_split_layer_launch_matches_single_launch(case, monkeypatch):
    """A dense4_kernel<2> launch whose last round of workgroups is less than half full hands the sets of that round to a
    second launch at half the block width (launch_dense: BASELINE cfg 2 is such a launch).  The rows of the two launches then
    carry different numbers of per-column-block partials (logits, d(coords)), which the finish kernels must pick per row.
    SVAE_DENSE4_TAIL==<k> forces the split after k sets of 512 rows at test sizes; results must equal the single launch's."""
    monkeypatch.setenv("SVAE_DENSE4_TAIL", "0")
    one, c_one = _run(case, "2", monkeypatch)
    monkeypatch.setenv("SVAE_DENSE4_TAIL", "=1")
    two, c_two = _run(case, "2", monkeypatch)
    B, n = case[2], case[1]
    rows = B * ((n * n + 31) // 32 * 32)
    assert rows > 512, "the case must span more than one set for the split to exist"
    assert c_one.get("dense4_tail", 0) == 0
    expect = 2 * (case[5] - 1) if case[6] <= 2 else 2 * (case[5] - 1) - 1      # C = 3: the logits partials have no room for a tail
    assert c_two["dense4_tail"] == expect, c_two
    for k in one:
        e = rel_err(two[k], one[k])
        assert e < (1e-5 if case[4] >= 500 else 2e-6), (case[0], k, e)
