"""Numerics of the fp16x3 split-operand GEMM (spatial_vae_amd/csrc/split.h) emulated in numpy, against fp64 and against an
fp32 GEMM.  Operands: s*x = hi + lo with hi = half(s x), lo = half(s x - hi), s a power of two per tensor; product =
hi_a hi_w + hi_a lo_w + lo_a hi_w accumulated in fp32 (numpy's float32 matmul of half-valued float32 arrays: products of
two halfs are exact in fp32, the accumulation is fp32 like the MFMA's).

    python tools/split_numerics.py          # prints the table quoted in DESIGN.md
"""
import numpy as np


def split(x, s):
    xs = x.astype(np.float64) * s
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def pow2_floor(v):
    return 2.0 ** np.floor(np.log2(v))


def gemm_fp16x3(A, sa, W, sw):
    Ah, Al = split(A, sa)
    Wh, Wl = split(W, sw)
    return ((Ah @ Wh) + (Ah @ Wl) + (Al @ Wh)).astype(np.float64) / (sa * sw)


def errors(C, ref):
    e = np.abs(C - ref)
    return float(e.max() / np.abs(ref).max()), float(np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean()))


def study(M=4096, K=500, N=500, seed=0):
    rs = np.random.RandomState(seed)
    A = np.tanh(rs.normal(size=(M, K))).astype(np.float32)                       # activations in (-1, 1)
    W = (rs.uniform(-1, 1, size=(K, N)) / np.sqrt(K)).astype(np.float32)         # nn.Linear-like weights
    G = (rs.normal(size=(M, N)) * 1e-6 * np.exp(rs.normal(size=(M, N)) * 2)).astype(np.float32)   # wide-range gradients
    sw = pow2_floor(8192 / np.abs(W).max())
    sg = pow2_floor(16384 / (np.abs(G).max() * 8))                               # a bound 8x above the true maximum
    out = {}
    ref = A.astype(np.float64) @ W.astype(np.float64)
    out["forward fp32"] = errors((A @ W).astype(np.float64), ref)
    out["forward fp16x3"] = errors(gemm_fp16x3(A, 1024.0, W, sw), ref)
    ref = G.astype(np.float64) @ W.T.astype(np.float64)
    out["dgrad fp32"] = errors((G @ W.T).astype(np.float64), ref)
    out["dgrad fp16x3"] = errors(gemm_fp16x3(G, sg, np.ascontiguousarray(W.T), sw), ref)
    ref = G.T.astype(np.float64) @ A.astype(np.float64)
    out["wgrad fp32"] = errors((G.T @ A).astype(np.float64), ref)
    out["wgrad fp16x3"] = errors(gemm_fp16x3(np.ascontiguousarray(G.T), sg, A, 1024.0), ref)
    return out


if __name__ == "__main__":
    for name, (emax, erms) in study().items():
        print("%-16s max|err|/max|ref| %.2e   rms err / rms ref %.2e" % (name, emax, erms))
