// HBM-bound kernels around the MFMA GEMMs: weight packing, the per-image first-layer tables,
// the coordinate layer (A1 + A2), the output layer (A4), the likelihoods (A5, A6) and the small
// gradient reductions of A8.  All reductions are fixed-order (no atomics): results are bitwise
// reproducible run to run.
#pragma once
#include <type_traits>
#include "common.h"

namespace svae {

// ---------------------------------------------------------------- weight packing
// W (H x H, row-major [n][k]) -> the two LDS images of dense_kernel, zero-padded to Hp.  An image is
// an array of 1 KiB slabs [contraction octet g][output tile t]; inside a slab [quad hh][lane 32][4]:
//   wf (forward, contract over k, lane = n):        k = 8g + 4hh + e, n = 32t + lane
//   wb (data gradient, contract over n, lane = k):  n = 8g + 4hh + e, k = 32t + lane
// so that MFMA lane l = lane + 32*hh finds its four k-steps at bytes 16*l of the slab.
__device__ __forceinline__ long slab_index(int contr, int out, int ntile) {
    return (((long)(contr >> 3) * ntile + (out >> 5)) * 2 + ((contr >> 2) & 1)) * 128 + (out & 31) * 4 + (contr & 3);
}
// ---------------------------------------------------------------- per-image tables
// tab[b][k][0..4] = W_c[k][p] + sum_q W_bi[k][p][q] z[b][q]      (models.py:104, 114-121)
// tab[b][k][5]    = b_c[k]   + sum_q W_z[k][q]  z[b][q]          (models.py:104, 111-112)
// posebuf[b]      = (cos t_b, sin t_b, dx0, dx1)                  (train_mnist.py:54-58, 70-71)
// wb_row_scale[l] (nullable): the data-gradient image of layer l holds W[n][k] * wb_row_scale[n] -- the rank-1 output-layer
// backward (dense_kernel LASTD >= 2) folds w_o[n] into the weights it contracts over n.
// Both in ONE launch (r01: one kernel for the tables and one per layer for the packing): what a forward call prepares from
// the parameters, z and the pose.  Blocks [0, nb_tab) build the tables, the rest pack layer (blk - nb_tab) / nb_pack.
struct PrepareArgs {
    const float* coord_w; const float* coord_b; const float* latent_w; const float* bil_w; const float* z;
    float* tab; float4* posebuf; PoseArgs pose;
    int B, H, Hp, Zd, in_dim;
    int nb_tab, nb_pack, nlayers;
    const float* W[SVAE_MAX_HIDDEN]; float* wf[SVAE_MAX_HIDDEN]; float* wb[SVAE_MAX_HIDDEN];
    const float* wb_row_scale[SVAE_MAX_HIDDEN];
};
__global__ void __launch_bounds__(256) prepare_kernel(PrepareArgs a) {
    if ((int)blockIdx.x < a.nb_tab) {
        const long idx = (long)blockIdx.x * 256 + threadIdx.x;
        if (idx < a.B) {
            float c, s, d0, d1;
            image_pose(a.pose, (int)idx, c, s, d0, d1);
            a.posebuf[idx] = make_float4(c, s, d0, d1);
        }
        if (idx >= (long)a.B * a.Hp) return;
        const int b = idx / a.Hp, k = idx % a.Hp;
        float out[kSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k < a.H) {
            for (int p = 0; p < a.in_dim; ++p) {
                float w = a.coord_w[k * a.in_dim + p];
                if (a.bil_w)
                    for (int q = 0; q < a.Zd; ++q) w += a.bil_w[((long)k * a.in_dim + p) * a.Zd + q] * a.z[(long)b * a.Zd + q];
                out[p] = w;
            }
            float bias = a.coord_b[k];
            if (a.latent_w)
                for (int q = 0; q < a.Zd; ++q) bias += a.latent_w[(long)k * a.Zd + q] * a.z[(long)b * a.Zd + q];
            out[kBiasSlot] = bias;
        }
        float4* dst = reinterpret_cast<float4*>(a.tab + idx * kSlots);
        dst[0] = make_float4(out[0], out[1], out[2], out[3]);
        dst[1] = make_float4(out[4], out[5], out[6], out[7]);
        return;
    }
    const int rel = blockIdx.x - a.nb_tab;
    const int l = rel / a.nb_pack;
    const long idx = (long)(rel - l * a.nb_pack) * 256 + threadIdx.x;
    if (l >= a.nlayers || idx >= (long)a.Hp * a.Hp) return;
    const int n = idx / a.Hp, k = idx % a.Hp;
    const float v = (n < a.H && k < a.H) ? a.W[l][(long)n * a.H + k] : 0.0f;
    a.wf[l][slab_index(k, n, a.Hp / 32)] = v;
    a.wb[l][slab_index(n, k, a.Hp / 32)] = (a.wb_row_scale[l] && n < a.H) ? v * a.wb_row_scale[l][n] : v;
}

struct RowGeo {
    int N, Npad, Hp, in_dim, act, B;
};

// ---------------------------------------------------------------- coordinate layer, forward
// a0[m][k] = act( sum_p feat_p(x''_m) tab[b][k][p] + tab[b][k][5] ).  A block owns 128 features x 2 row-halves
// (thread = (k, half)) and a chunk of up to kL0Chunk row octets of ONE image: the posed coordinates of the chunk's rows
// are computed once per block into LDS ([octet][half] -> x0 x 4 | x1 x 4, read back as two broadcast ds_read_b128: the
// 32 lanes of a half-wave share an address), the thread's table entry is loaded once, and the loop does nothing but
// 2 LDS reads, 8 FMAs, 4 activations and ONE 16-byte store per octet (1 KiB contiguous per wave instruction).  r01's
// form re-read 16 coordinates and the table entry for every 4 stores (19 loads per 4 stores: 4.2 TB/s); this one is
// bound by the store stream.
constexpr int kL0Chunk = 32;  // row octets per block (LDS: 32 x 2 x 32 B = 2 KiB)
template <int ACT>
__global__ void __launch_bounds__(256) layer0_fwd_kernel(PoseArgs pose, const float4* __restrict__ posebuf,
                                                         const float* __restrict__ tab, float* __restrict__ a0, RowGeo g,
                                                         int oct_per_chunk, int chunks_per_image, long nchunks) {
    __shared__ float4 xs[kL0Chunk * 2 * 2];
    const long chunk = (long)blockIdx.z * gridDim.y + blockIdx.y;
    if (chunk >= nchunks) return;  // block-uniform
    const int b = (int)(chunk / chunks_per_image), ci = (int)(chunk - (long)b * chunks_per_image);
    const int oimg = g.Npad >> 3;
    const int oc0 = ci * oct_per_chunk;
    const int noc = (oc0 + oct_per_chunk <= oimg) ? oct_per_chunk : oimg - oc0;
    const float4 pb = posebuf[b];  // identity (1, 0, 0, 0) when the coordinates are explicit
    const float* cbase = pose.coords ? pose.coords + (long)b * g.N * 2 : pose.grid;
    float* xf = reinterpret_cast<float*>(xs);
    for (int j = threadIdx.x; j < noc * 8; j += 256) {  // row j of the chunk; pad rows (i >= N) are (0, 0)
        const int i = oc0 * 8 + j;
        float x0 = 0.0f, x1 = 0.0f;
        if (i < g.N) {
            const float2 raw = *reinterpret_cast<const float2*>(cbase + (long)i * 2);
            x0 = pb.x * raw.x - pb.y * raw.y + pb.z;
            x1 = pb.y * raw.x + pb.x * raw.y + pb.w;
        }
        const int slot = (j >> 2) * 8 + (j & 3);  // (octet, half) pair j >> 2: [x0 x 4][x1 x 4]
        xf[slot] = x0;
        xf[slot + 4] = x1;
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Hp * 2) return;
    const int h = t & 1, k = t >> 1;
    const float4* tp = reinterpret_cast<const float4*>(tab + ((long)b * g.Hp + k) * kSlots);
    const float4 t0 = tp[0], t1 = tp[1];
    const long ostride = (long)g.Hp * 8;
    float* dst = a0 + (((long)b * oimg + oc0) * g.Hp * 2 + t) * 4;
    const bool expand = g.in_dim == 5;
#pragma unroll 4
    for (int jo = 0; jo < noc; ++jo) {
        const float4 X0 = xs[(jo * 2 + h) * 2], X1 = xs[(jo * 2 + h) * 2 + 1];
        const float x0[4] = {X0.x, X0.y, X0.z, X0.w}, x1[4] = {X1.x, X1.y, X1.z, X1.w};
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = t1.y + x0[e] * t0.x + x1[e] * t0.y;
            if (expand) v += (x0[e] * x0[e]) * t0.z + (x1[e] * x1[e]) * t0.w + (x0[e] * x1[e]) * t1.x;
            out[e] = act_fwd<ACT>(v);
        }
        *reinterpret_cast<float4*>(dst + jo * ostride) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// ---------------------------------------------------------------- output layer, forward (A4)
// logits[m][c] = sum_n a[m][n] W_o[c][n] + b_o[c]; y = sigmoid(logits) (+ softplus on channel 0).
// One wave per row octet: lane (nl, h) walks the feature tiles with 16-byte loads (4 rows each)
// and the 32 lanes of a half-wave are summed at the end.
template <int C>
__global__ void out_fwd_kernel(const float* __restrict__ a, const float* __restrict__ out_w,
                               const float* __restrict__ out_b, float* __restrict__ y, float* __restrict__ logits,
                               RowGeo g, int H, int softplus, long noct) {
    const int lane = threadIdx.x & 63;
    const int nl = lane & 31, h = lane >> 5;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    for (long o = wave0; o < noct; o += nwaves) {
        float p[4][C];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < C; ++c) p[e][c] = 0.0f;
        for (int t = 0; t < g.Hp / 32; ++t) {
            const int n = t * 32 + nl;
            const float4 v = *reinterpret_cast<const float4*>(a + ((o * g.Hp + n) * 8 + 4 * h));
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float w = (n < H) ? out_w[c * H + n] : 0.0f;
                p[0][c] += v.x * w; p[1][c] += v.y * w; p[2][c] += v.z * w; p[3][c] += v.w * w;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < C; ++c) p[e][c] = wave_sum32(p[e][c]);
        if (nl < 4) {
            const long m = 8 * o + 4 * h + nl;
            const int b = m / g.Npad, i = m % g.Npad;
            if (i < g.N) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float s0 = nl == 0 ? p[0][c] : nl == 1 ? p[1][c] : nl == 2 ? p[2][c] : p[3][c];
                    const float lg = s0 + out_b[c];
                    float s = 1.0f / (1.0f + expf(-lg));
                    if (softplus && c == 0) s = log1pf(expf(s));
                    const long oi = ((long)b * g.N + i) * C + c;
                    y[oi] = s;
                    if (logits) logits[oi] = lg;
                }
            }
        }
    }
}

// Output layer when the last dense_kernel already contracted a_{L-1} with W_o (CF epilogue): sum the per-column-block
// partial logits [nblk][C][Mp], add the bias, apply Sigmoid (+softplus).  One thread per (image, pixel).
// Rows from m_split on carry nblk_tail partials per channel instead of nblk: the GEMM that wrote them ran its last partial round
// of workgroups at half the block width (launch_dense).  m_split = Mp when the layer was one launch.
__global__ void logits_finish_kernel(const float* __restrict__ lpart, const float* __restrict__ out_b, float* __restrict__ y,
                                     float* __restrict__ logits, RowGeo g, int C, int nblk, int softplus, long Mp, long m_split,
                                     int nblk_tail) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)g.B * g.N) return;
    const int b = (int)(t / g.N), i = (int)(t - (long)b * g.N);
    const long m = (long)b * g.Npad + i;
    if (m >= m_split) nblk = nblk_tail;
    for (int c = 0; c < C; ++c) {
        float lg = 0.0f;
        for (int k = 0; k < nblk; ++k) lg += lpart[((long)k * C + c) * Mp + m];
        lg += out_b[c];
        float s = 1.0f / (1.0f + expf(-lg));
        if (softplus && c == 0) s = log1pf(expf(s));
        y[t * C + c] = s;
        if (logits) logits[t * C + c] = lg;
    }
}

// The same finish with the Bernoulli log-likelihood of train_mnist.py:78-81 / train_galaxy.py:116-119 folded in (A5 inside
// A4's last kernel): one block per image sums the partial logits, applies bias + Sigmoid, writes y (and the logits) and, from
// the SAME fp32 sigmoid value torch would see, the clamped log terms of F.binary_cross_entropy -- loglik[b] --
// and d(loglik_b)/d(y) -- dll -- exactly as bce_kernel computes them from y in a second pass.
__global__ void __launch_bounds__(1024) logits_finish_bce_kernel(const float* __restrict__ lpart, const float* __restrict__ out_b,
                                                                const float* __restrict__ target, float* __restrict__ y,
                                                                float* __restrict__ logits, float* __restrict__ loglik,
                                                                float* __restrict__ dll, RowGeo g, int C, int nblk, long Mp,
                                                                long m_split, int nblk_tail) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    float acc = 0.0f;
    // gridDim.y chunks of the image's pixels (1 at 28 x 28 and 40 x 40: one pixel per thread, a single round trip; 16 at
    // 128 x 128); with several chunks `loglik` is the [B][gridDim.y] partial array that loglik_reduce_kernel sums
    const int per = (g.N + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = (i0 + per < g.N) ? i0 + per : g.N;
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const long m = (long)b * g.Npad + i;
        const long t = (long)b * g.N + i;
        const int nbk = m >= m_split ? nblk_tail : nblk;
        for (int c = 0; c < C; ++c) {
            float lg = 0.0f;
            for (int k = 0; k < nbk; ++k) lg += lpart[((long)k * C + c) * Mp + m];
            lg += out_b[c];
            const float s = 1.0f / (1.0f + expf(-lg));
            const float tg = target[t * C + c];
            y[t * C + c] = s;
            if (logits) logits[t * C + c] = lg;
            const float ls = fmaxf(logf(s), -100.0f);
            const float l1s = fmaxf(log1pf(-s), -100.0f);
            acc += tg * ls + (1.0f - tg) * l1s;
            if (dll) dll[t * C + c] = -(s - tg) / fmaxf((1.0f - s) * s, 1e-12f);
        }
    }
    acc = block_sum_waves(acc, red);
    if (threadIdx.x == 0) loglik[(long)b * gridDim.y + blockIdx.y] = acc;
}

// loglik[b] = sum over the pixel chunks of part[b][chunk], in chunk order
__global__ void loglik_reduce_kernel(const float* __restrict__ part, float* __restrict__ loglik, int B, int chunks) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float s = 0.0f;
    for (int k = 0; k < chunks; ++k) s += part[(long)b * chunks + k];
    loglik[b] = s;
}

// ---------------------------------------------------------------- output layer, backward, step 1
// do_p[c][mp] = dy * dy_scale[b] * (softplus') * s(1-s), recomputing s from the logits exactly as the forward did.  The
// grid runs over the PADDED row space (c, mp) and writes the exact zeros of the pad rows itself (no memset launch).
__global__ void dlogits_kernel(const float* __restrict__ logits, const float* __restrict__ dy,
                               const float* __restrict__ dy_scale, float* __restrict__ do_p, int B, int N, int Npad,
                               int C, int softplus, long Mp, unsigned* __restrict__ amax) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over C * Mp
    float m = 0.0f;
    if (idx < (long)C * Mp) {
        const int c = (int)(idx / Mp);
        const long mp = idx - (long)c * Mp;
        const int b = (int)(mp / Npad), i = (int)(mp - (long)b * Npad);
        float v = 0.0f;
        if (i < N) {
            const long src = ((long)b * N + i) * C + c;
            const float lg = logits[src];
            const float s = 1.0f / (1.0f + expf(-lg));
            float g = dy[src];
            if (dy_scale) g *= dy_scale[b];
            if (softplus && c == 0) g *= 1.0f / (1.0f + expf(-s));
            v = g * s * (1.0f - s);
        }
        do_p[idx] = v;
        m = fabsf(v);
    }
    if (amax) {  // fp16x3 mode: max |do| (non-negative floats order like their bit patterns; max is order-independent)
        __shared__ float red[4];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (m > 0.0f) atomicMax(amax, __float_as_uint(m));  // one atomic per block, none for all-zero blocks
        }
    }
}

// ---------------------------------------------------------------- output layer, backward, step 2
// dh[m][n] = (sum_c do[m][c] W_o[c][n]) * act'(a[m][n])  -> octet-major;  partial dW_o[c][n] per
// row chunk.  Thread = fixed (n, half); it streams its column through the chunk's octets.
template <int ACT, int C>
__global__ void out_bwd_kernel(const float* __restrict__ a, const float* __restrict__ do_p,
                               const float* __restrict__ out_w, float* __restrict__ dh, float* __restrict__ wpart,
                               float* __restrict__ bpart, int H, int Hp, long Mp, long noct, long oct_per_chunk) {
    const int t = blockIdx.x * 256 + threadIdx.x;  // over Hp*2
    if (t >= Hp * 2) return;
    const int h = t & 1, n = t >> 1;
    float w[C], pw[C], pb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        w[c] = (n < H) ? out_w[c * H + n] : 0.0f;
        pw[c] = 0.0f;
        pb[c] = 0.0f;
    }
    const long o0 = blockIdx.y * oct_per_chunk;
    const long o1 = (o0 + oct_per_chunk < noct) ? o0 + oct_per_chunk : noct;
    for (long o = o0; o < o1; ++o) {
        const long off = (o * Hp + n) * 8 + 4 * h;
        const float4 av = *reinterpret_cast<const float4*>(a + off);
        float4 da = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 d = *reinterpret_cast<const float4*>(do_p + (long)c * Mp + 8 * o + 4 * h);
            da.x += d.x * w[c]; da.y += d.y * w[c]; da.z += d.z * w[c]; da.w += d.w * w[c];
            pw[c] += (d.x * av.x + d.y * av.y) + (d.z * av.z + d.w * av.w);
            pb[c] += (d.x + d.y) + (d.z + d.w);
        }
        da.x *= act_grad<ACT>(av.x); da.y *= act_grad<ACT>(av.y);
        da.z *= act_grad<ACT>(av.z); da.w *= act_grad<ACT>(av.w);
        *reinterpret_cast<float4*>(dh + off) = da;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) wpart[(((long)blockIdx.y * 2 + h) * C + c) * Hp + n] = pw[c];
    if (n == 0) {  // the two half-wave threads of column 0 also carry this chunk's share of db_o
#pragma unroll
        for (int c = 0; c < C; ++c) bpart[((long)blockIdx.y * 2 + h) * C + c] = pb[c];
    }
}

// dW_o[c][n] = sum over (chunk, half) of wpart; db_o[c] = sum over (chunk, half) of bpart, in a fixed order.  The
// partials are ~1 MB in L2; the kernel is bound by round trips, so it is built for loads in flight: a block of 1024 threads
// owns 64 consecutive columns of one channel (a 256-byte coalesced sweep per part), thread = (column, part lane of 16), each
// thread sums its parts in 8 interleaved chains, and the 16 lanes of a column are combined through LDS in a fixed order
// (r01: 32 columns x 8 lanes with one chain per thread -- 64 dependent loads -- took 33 us for 512 parts).
constexpr int kObrLanes = 16;
__global__ void __launch_bounds__(1024) out_bwd_reduce_kernel(const float* __restrict__ wpart, const float* __restrict__ bpart,
                                                              float* __restrict__ dWo, float* __restrict__ dbo, int C, int H,
                                                              int Hp, int nparts) {
    __shared__ float red[kObrLanes][65];
    const int col = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int blocks_per_c = (Hp + 63) / 64;
    if (blockIdx.x == (unsigned)(C * blocks_per_c)) {  // last block: the C bias gradients, same lane scheme over bpart
        float s = 0.0f;
        if (col < C)
            for (int i = pl; i < nparts; i += kObrLanes) s += bpart[(long)i * C + col];
        red[pl][col] = s;
        __syncthreads();
        if (pl == 0 && col < C && dbo) {
            float t = red[0][col];
#pragma unroll
            for (int j = 1; j < kObrLanes; ++j) t += red[j][col];
            dbo[col] = t;
        }
        return;
    }
    const int c = blockIdx.x / blocks_per_c, n = (blockIdx.x % blocks_per_c) * 64 + col;
    float s = 0.0f;
    if (n < Hp) {
        const float* p = wpart + (long)c * Hp + n;
        const long st = (long)C * Hp;
        float ch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = pl;
        for (; i + 7 * kObrLanes < nparts; i += 8 * kObrLanes) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ch[j] += p[(long)(i + j * kObrLanes) * st];
        }
        for (; i < nparts; i += kObrLanes) ch[0] += p[(long)i * st];
        s = ((ch[0] + ch[1]) + (ch[2] + ch[3])) + ((ch[4] + ch[5]) + (ch[6] + ch[7]));
    }
    red[pl][col] = s;
    __syncthreads();
    if (pl == 0 && n < H && dWo) {
        float t = red[0][col];
#pragma unroll
        for (int j = 1; j < kObrLanes; ++j) t += red[j][col];
        dWo[(long)c * H + n] = t;
    }
}

// ---------------------------------------------------------------- coordinate layer, backward
// (a) parameter side: per (image, chunk) partial sums over rows of dh0 and dh0 * feat_p:
//     sg[part][k][half][0..4] = G[k][p], [5] = S[k].  Thread = fixed (k, half).
__global__ void layer0_bwd_params_kernel(PoseArgs pose, const float4* __restrict__ posebuf,
                                         const float* __restrict__ dh0, float* __restrict__ sgpart, RowGeo g,
                                         int oct_per_chunk, int chunks_per_image) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Hp * 2) return;
    const int h = t & 1, k = t >> 1;
    const int b = blockIdx.y / chunks_per_image, ci = blockIdx.y % chunks_per_image;
    const int oimg = g.Npad / 8;
    const int oc0 = ci * oct_per_chunk;
    const int oc1 = (oc0 + oct_per_chunk < oimg) ? oc0 + oct_per_chunk : oimg;
    const float4 pb = posebuf[b];
    float G[5] = {0, 0, 0, 0, 0};
    float S = 0.0f;
    for (int oc = oc0; oc < oc1; ++oc) {
        const long o = (long)b * oimg + oc;
        const float4 d = *reinterpret_cast<const float4*>(dh0 + (o * g.Hp + k) * 8 + 4 * h);
        const float dv[4] = {d.x, d.y, d.z, d.w};
        const int i0 = oc * 8 + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float2 x = row_coord(pose, pb, b, i0 + e, g.N);
            S += dv[e];
            G[0] += dv[e] * x.x;
            G[1] += dv[e] * x.y;
            if (g.in_dim == 5) {
                G[2] += dv[e] * x.x * x.x;
                G[3] += dv[e] * x.y * x.y;
                G[4] += dv[e] * x.x * x.y;
            }
        }
    }
    float4* dst = reinterpret_cast<float4*>(sgpart + (((long)blockIdx.y * g.Hp + k) * 2 + h) * kSlots);
    dst[0] = make_float4(G[0], G[1], G[2], G[3]);
    dst[1] = make_float4(G[4], S, 0.0f, 0.0f);
}

// sgimg[b][k][slot] = sum over (chunk, half) of sgpart
__global__ void sg_reduce_kernel(const float* __restrict__ sgpart, float* __restrict__ sgimg, int B, int Hp,
                                 int chunks_per_image) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over B*Hp*8
    if (idx >= (long)B * Hp * kSlots) return;
    const int slot = idx % kSlots;
    const long bk = idx / kSlots;
    const int k = bk % Hp, b = bk / Hp;
    float s = 0.0f;
    for (int ci = 0; ci < chunks_per_image; ++ci)
        for (int h = 0; h < 2; ++h)
            s += sgpart[((((long)b * chunks_per_image + ci) * Hp + k) * 2 + h) * kSlots + slot];
    sgimg[idx] = s;
}

// (b) coordinate side: dfeat[m][p] = sum_k dh0[m][k] tab[b][k][p], chained through the feature
//     expansion to d(coords).  One wave per row octet: lane = (row-in-octet, k mod 8).
__global__ void layer0_bwd_coords_kernel(PoseArgs pose, const float4* __restrict__ posebuf,
                                         const float* __restrict__ dh0, const float* __restrict__ tab,
                                         float* __restrict__ dcoords, RowGeo g, long noct) {
    const int lane = threadIdx.x & 63;
    const int ml = lane & 7, kq = lane >> 3;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    for (long o = wave0; o < noct; o += nwaves) {
        const long m = 8 * o + ml;
        const int b = m / g.Npad, i = m % g.Npad;
        float acc[5] = {0, 0, 0, 0, 0};
        for (int k = kq; k < g.Hp; k += 8) {
            const float d = dh0[(o * g.Hp + k) * 8 + ml];
            const float4* tp = reinterpret_cast<const float4*>(tab + ((long)b * g.Hp + k) * kSlots);
            const float4 t0 = tp[0];
            acc[0] += d * t0.x;
            acc[1] += d * t0.y;
            if (g.in_dim == 5) {
                const float4 t1 = tp[1];
                acc[2] += d * t0.z;
                acc[3] += d * t0.w;
                acc[4] += d * t1.x;
            }
        }
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            acc[p] += __shfl_xor(acc[p], 8);
            acc[p] += __shfl_xor(acc[p], 16);
            acc[p] += __shfl_xor(acc[p], 32);
        }
        if (kq == 0 && i < g.N) {
            float d0 = acc[0], d1 = acc[1];
            if (g.in_dim == 5) {
                const float2 x = row_coord(pose, posebuf[b], b, i, g.N);
                d0 += 2.0f * x.x * acc[2] + x.y * acc[4];
                d1 += 2.0f * x.y * acc[3] + x.x * acc[4];
            }
            *reinterpret_cast<float2*>(dcoords + ((long)b * g.N + i) * 2) = make_float2(d0, d1);
        }
    }
}

// (a') + (b') + (d) + (e), fused path: the data-gradient GEMM of the first hidden layer already reduced dh0 over each
//      32-row tile (dense_kernel<.., FIRST>: sgtile[tile][k] = (G0, G1, S, -), both row halves summed; the fp16x3 kernel
//      keeps the halves apart: nhalf = 2) and over each column block (dfpart[block][mp] = d(coords) partial).  What is left
//      are per-image, fixed-order sums, done by ONE launch of B x 2 blocks instead of four kernels:
//        role 0 (blockIdx.y == 0): sgimg[b][k] = sum over the image's tiles (and halves); then dz[b][q] from it (d);
//        role 1 (blockIdx.y == 1): d(coords)[b][i] = sum over column blocks; then dtheta[b], ddx[b] from it (e).
struct FirstLayerImageArgs {
    const float* sgtile;
    int nhalf, Timg, H, Hp;
    float* sgimg;
    const float* latent_w;
    const float* bil_w;
    float* dz;
    int Zd, in_dim;
    const float* dfpart;
    int nblocks, N, Npad;
    long Mp;
    float* dcoords;
    const float* grid;
    const float4* posebuf;
    float* dtheta;
    float* ddx;
    long m_split;
    int nblocks_tail;
};
// one block's share: image b, role 0 (sgimg, dz) or 1 (d(coords), dtheta, ddx); `red` is 4 floats of LDS
__device__ __forceinline__ void first_layer_image_block(const FirstLayerImageArgs& a, int b, int role, float* red) {
    const float* __restrict__ sgtile = a.sgtile;
    const int nhalf = a.nhalf, Timg = a.Timg, H = a.H, Hp = a.Hp;
    float* __restrict__ sgimg = a.sgimg;
    const float* __restrict__ latent_w = a.latent_w;
    const float* __restrict__ bil_w = a.bil_w;
    float* __restrict__ dz = a.dz;
    const int Zd = a.Zd, in_dim = a.in_dim;
    const float* __restrict__ dfpart = a.dfpart;
    const int nblocks = a.nblocks, N = a.N, Npad = a.Npad;
    const long Mp = a.Mp;
    float* __restrict__ dcoords = a.dcoords;
    const float* __restrict__ grid = a.grid;
    const float4* __restrict__ posebuf = a.posebuf;
    float* __restrict__ dtheta = a.dtheta;
    float* __restrict__ ddx = a.ddx;
    const long m_split = a.m_split;
    const int nblocks_tail = a.nblocks_tail;

    if (role == 0) {
        for (int k = threadIdx.x; k < Hp; k += 256) {
            // the image's Timg * nhalf partial rows, 8 loads in flight (fixed order: chain j takes rows j, j+8, ...)
            const float* src = sgtile + ((long)b * Timg * nhalf * Hp + k) * 4;
            const long rst = (long)Hp * 4;
            const int rows = Timg * nhalf;
            float c0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int r = 0;
            for (; r + 7 < rows; r += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 v = *reinterpret_cast<const float4*>(src + (r + j) * rst);
                    c0[j] += v.x; c1[j] += v.y; c2[j] += v.z;
                }
            }
            for (int j = 0; r < rows; ++r, ++j) {
                const float4 v = *reinterpret_cast<const float4*>(src + r * rst);
                c0[j] += v.x; c1[j] += v.y; c2[j] += v.z;
            }
            const float g0 = ((c0[0] + c0[1]) + (c0[2] + c0[3])) + ((c0[4] + c0[5]) + (c0[6] + c0[7]));
            const float g1 = ((c1[0] + c1[1]) + (c1[2] + c1[3])) + ((c1[4] + c1[5]) + (c1[6] + c1[7]));
            const float sv = ((c2[0] + c2[1]) + (c2[2] + c2[3])) + ((c2[4] + c2[5]) + (c2[6] + c2[7]));
            float4* dst = reinterpret_cast<float4*>(sgimg + ((long)b * Hp + k) * kSlots);
            dst[0] = make_float4(g0, g1, 0.0f, 0.0f);
            dst[1] = make_float4(0.0f, sv, 0.0f, 0.0f);
        }
        if (!dz || Zd <= 0) return;
        // dz[b][q] = sum_k S_b[k] W_z[k][q] + sum_{k,p} G_b[k][p] W_bi[k][p][q]; every thread re-reads what it wrote itself
        for (int q = 0; q < Zd; ++q) {
            float s = 0.0f;
            for (int k = threadIdx.x; k < H; k += 256) {
                const float* e = sgimg + ((long)b * Hp + k) * kSlots;
                s += e[kBiasSlot] * latent_w[(long)k * Zd + q];
                if (bil_w)
                    for (int p = 0; p < in_dim; ++p) s += e[p] * bil_w[((long)k * in_dim + p) * Zd + q];
            }
            s = block_sum256(s, red);
            if (threadIdx.x == 0) dz[(long)b * Zd + q] = s;
        }
        return;
    }
    if (!dcoords) return;
    const bool want_pose = dtheta || ddx;
    float4 pb = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    if (want_pose) pb = posebuf[b];
    float st = 0.0f, s0 = 0.0f, s1 = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const long mp = (long)b * Npad + i;
        float d0 = 0.0f, d1 = 0.0f;
        const int nbk = mp >= m_split ? nblocks_tail : nblocks;   // rows of a half-width tail launch carry twice the partials
        for (int nb = 0; nb < nbk; ++nb) {
            const float2 v = *reinterpret_cast<const float2*>(dfpart + ((long)nb * Mp + mp) * 2);
            d0 += v.x; d1 += v.y;
        }
        *reinterpret_cast<float2*>(dcoords + ((long)b * N + i) * 2) = make_float2(d0, d1);
        if (want_pose) {  // dtheta[b] = sum_i dx0 (-s g0 - c g1) + dx1 (c g0 - s g1);  ddx[b] = sum_i dcoords[b,i]
            const float2 gr = *reinterpret_cast<const float2*>(grid + (long)i * 2);
            st += d0 * (-pb.y * gr.x - pb.x * gr.y) + d1 * (pb.x * gr.x - pb.y * gr.y);
            s0 += d0;
            s1 += d1;
        }
    }
    if (!want_pose) return;
    st = block_sum256(st, red);
    s0 = block_sum256(s0, red);
    s1 = block_sum256(s1, red);
    if (threadIdx.x == 0) {
        if (dtheta) dtheta[b] = st;
        if (ddx) {
            ddx[2 * b] = s0;
            ddx[2 * b + 1] = s1;
        }
    }
}

__global__ void __launch_bounds__(256) first_layer_image_kernel(FirstLayerImageArgs a) {
    __shared__ float red[4];
    first_layer_image_block(a, blockIdx.x, blockIdx.y, red);
}
// The same per-image blocks and the split-K reduction of the first hidden layer's weight gradient in ONE launch: both are
// latency-bound (25.5 + 20.6 us at BASELINE cfg 2 as two launches) and independent of each other, so their blocks fill the
// chip together.  Blocks [0, nimg) are (image, role) pairs, the rest wgrad_reduce blocks.
__global__ void __launch_bounds__(256) backward_tail_kernel(FirstLayerImageArgs f, WgradReduceArgs r, unsigned nimg, int B) {
    __shared__ float red[4][64];
    if (blockIdx.x < nimg) first_layer_image_block(f, (int)(blockIdx.x % (unsigned)B), (int)(blockIdx.x / (unsigned)B), &red[0][0]);
    else wgrad_reduce_block(r, blockIdx.x - nimg, red);
}

// (c) first-layer parameter gradients from the per-image sums.  A block owns 16 consecutive (k, slot) columns of sgimg;
//     thread = (column, image lane of 16): each lane walks its images ONCE (4 loads in flight), accumulating the plain sum
//     and the z-weighted sums of one chunk of latent dimensions together; the 16 lanes are combined in LDS in a fixed order.
//     dW_c[k][p] = sum_b G_b[k][p];  db_c[k] = sum_b S_b[k];  dW_z[k][q] = sum_b S_b[k] z[b][q];
//     dW_bi[k][p][q] = sum_b G_b[k][p] z[b][q]
constexpr int kZChunk = 8;
constexpr int kPgCols = 16, kPgLanes = 16;
__global__ void __launch_bounds__(256) layer0_param_grads_kernel(const float* __restrict__ sgimg, const float* __restrict__ z,
                                                                 float* __restrict__ dWc, float* __restrict__ dbc,
                                                                 float* __restrict__ dWz, float* __restrict__ dWbi, int B, int H,
                                                                 int Hp, int Zd, int in_dim) {
    __shared__ float red[kPgLanes][kPgCols + 1];
    const int col = threadIdx.x & (kPgCols - 1), bl = threadIdx.x / kPgCols;
    const int t = blockIdx.x * kPgCols + col;  // over H * kSlots
    const bool in_range = t < H * kSlots;
    const int k = in_range ? t / kSlots : 0, slot = in_range ? t % kSlots : 0;
    const bool is_g = in_range && slot < in_dim, is_s = in_range && slot == kBiasSlot;
    const float* src = sgimg + (long)k * kSlots + slot;
    const long bstride = (long)Hp * kSlots;
    auto combine = [&](float v) {  // fixed-order sum of the image lanes of this column; valid where bl == 0
        __syncthreads();
        red[bl][col] = v;
        __syncthreads();
        float r = red[0][col];
#pragma unroll
        for (int j = 1; j < kPgLanes; ++j) r += red[j][col];
        return r;
    };
    const bool want_z = (is_s && dWz) || (is_g && dWbi);
    const int nq0 = Zd < kZChunk ? Zd : kZChunk;
    float plain = 0.0f;
    float acc[kZChunk];
#pragma unroll
    for (int q = 0; q < kZChunk; ++q) acc[q] = 0.0f;
    if (is_g || is_s) {
        int b = bl;
        for (; b + 3 * kPgLanes < B; b += 4 * kPgLanes) {   // 4 images in flight
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = src[(long)(b + j * kPgLanes) * bstride];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                plain += v[j];
                if (want_z) {
#pragma unroll
                    for (int q = 0; q < kZChunk; ++q)
                        if (q < nq0) acc[q] += v[j] * z[(long)(b + j * kPgLanes) * Zd + q];
                }
            }
        }
        for (; b < B; b += kPgLanes) {
            const float v = src[(long)b * bstride];
            plain += v;
            if (want_z) {
#pragma unroll
                for (int q = 0; q < kZChunk; ++q)
                    if (q < nq0) acc[q] += v * z[(long)b * Zd + q];
            }
        }
    }
    plain = combine(plain);
    if (bl == 0) {
        if (is_g && dWc) dWc[k * in_dim + slot] = plain;
        if (is_s && dbc) dbc[k] = plain;
    }
    for (int q0 = 0; q0 < Zd; q0 += kZChunk) {  // uniform trip count: combine() holds barriers
        if (q0 > 0) {  // latent dimensions beyond the first chunk: another pass over the images
#pragma unroll
            for (int q = 0; q < kZChunk; ++q) acc[q] = 0.0f;
            if (want_z)
                for (int b = bl; b < B; b += kPgLanes) {
                    const float v = src[(long)b * bstride];
#pragma unroll
                    for (int q = 0; q < kZChunk; ++q)
                        if (q0 + q < Zd) acc[q] += v * z[(long)b * Zd + q0 + q];
                }
        }
#pragma unroll
        for (int q = 0; q < kZChunk; ++q) {
            if (q0 + q >= Zd) break;  // block-uniform
            const float r = combine(acc[q]);
            if (bl == 0 && want_z) {
                if (is_s) dWz[(long)k * Zd + q0 + q] = r;
                else dWbi[((long)k * in_dim + slot) * Zd + q0 + q] = r;
            }
        }
    }
}

// (d) dz[b][q] = sum_k S_b[k] W_z[k][q] + sum_{k,p} G_b[k][p] W_bi[k][p][q]   (one block per image)
__global__ void dz_kernel(const float* __restrict__ sgimg, const float* __restrict__ latent_w,
                          const float* __restrict__ bil_w, float* __restrict__ dz, int H, int Hp, int Zd, int in_dim) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    for (int q = 0; q < Zd; ++q) {
        float s = 0.0f;
        for (int k = threadIdx.x; k < H; k += 256) {
            const float* e = sgimg + ((long)b * Hp + k) * kSlots;
            s += e[kBiasSlot] * latent_w[(long)k * Zd + q];
            if (bil_w)
                for (int p = 0; p < in_dim; ++p) s += e[p] * bil_w[((long)k * in_dim + p) * Zd + q];
        }
        s = block_sum256(s, red);
        if (threadIdx.x == 0) dz[(long)b * Zd + q] = s;
    }
}

// (e) pose gradients from d(coords) (one block per image):
//     dtheta[b] = sum_i dx0 (-s g0 - c g1) + dx1 (c g0 - s g1);  ddx[b] = sum_i dcoords[b,i]
__global__ void pose_bwd_kernel(const float* __restrict__ dcoords, const float* __restrict__ grid,
                                const float4* __restrict__ posebuf, float* __restrict__ dtheta,
                                float* __restrict__ ddx, int N) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float4 pb = posebuf[b];
    float st = 0.0f, s0 = 0.0f, s1 = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float2 d = *reinterpret_cast<const float2*>(dcoords + ((long)b * N + i) * 2);
        const float2 gr = *reinterpret_cast<const float2*>(grid + (long)i * 2);
        st += d.x * (-pb.y * gr.x - pb.x * gr.y) + d.y * (pb.x * gr.x - pb.y * gr.y);
        s0 += d.x;
        s1 += d.y;
    }
    st = block_sum256(st, red);
    s0 = block_sum256(s0, red);
    s1 = block_sum256(s1, red);
    if (threadIdx.x == 0) {
        if (dtheta) dtheta[b] = st;
        if (ddx) {
            ddx[2 * b] = s0;
            ddx[2 * b + 1] = s1;
        }
    }
}

// ---------------------------------------------------------------- A5: Bernoulli log-likelihood
// loglik[b] = sum_j t log(s) + (1-t) log(1-s), both logs clamped at -100 (torch's
// binary_cross_entropy), dll = -(s-t)/max((1-s)s, 1e-12).  One block per image.
__global__ void bce_kernel(const float* __restrict__ y_hat, const float* __restrict__ target,
                           float* __restrict__ loglik, float* __restrict__ dll, int n) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float acc = 0.0f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const long idx = (long)b * n + j;
        const float s = y_hat[idx], t = target[idx];
        const float ls = fmaxf(logf(s), -100.0f);
        const float l1s = fmaxf(log1pf(-s), -100.0f);
        acc += t * ls + (1.0f - t) * l1s;
        if (dll) dll[idx] = -(s - t) / fmaxf((1.0f - s) * s, 1e-12f);
    }
    acc = block_sum256(acc, red);
    if (threadIdx.x == 0) loglik[b] = acc;
}

// ---------------------------------------------------------------- A5/A6: Gaussian log-likelihood
// One block per image.  Follows train_particles.py:102-139 including the channel-interleave quirk
// (mean = first N entries of the (N*C)-long row, log-variance = last N), the CTF cross-correlation
// of the mean image (zero padding k/2) and the pixel mask.  filt/dflt: (B, N) scratch for the
// filtered mean and the gradient flowing back into the filter's output.
__global__ void gaussian_kernel(const float* __restrict__ yp, const float* __restrict__ target,
                                const uint8_t* __restrict__ mask, const float* __restrict__ ctf, int k,
                                float* __restrict__ loglik, float* __restrict__ dll, float* __restrict__ filt,
                                float* __restrict__ dflt, int N, int C) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int n = (int)(sqrtf((float)N) + 0.5f);
    const float* row = yp + (long)b * N * C;
    const bool noise = C > 1;
    const int pad = k / 2;
    const float* f = ctf ? ctf + (long)b * k * k : nullptr;
    if (ctf) {
        for (int j = threadIdx.x; j < N; j += 256) {
            const int r = j / n, cidx = j % n;
            float s = 0.0f;
            for (int u = 0; u < k; ++u) {
                const int rr = r + u - pad;
                if (rr < 0 || rr >= n) continue;
                for (int v = 0; v < k; ++v) {
                    const int cc = cidx + v - pad;
                    if (cc < 0 || cc >= n) continue;
                    s += row[rr * n + cc] * f[u * k + v];
                }
            }
            filt[(long)b * N + j] = s;
        }
        __syncthreads();
    }
    float acc = 0.0f;
    for (int j = threadIdx.x; j < N; j += 256) {
        const bool on = mask ? mask[j] != 0 : true;
        const float mu = ctf ? filt[(long)b * N + j] : row[j];
        const float diff = mu - target[(long)b * N + j];
        float dmu = 0.0f, dlv = 0.0f;
        if (on) {
            if (noise) {
                const float lv = row[N + j];
                const float iv = expf(-lv);
                acc += -0.5f * (diff * diff / expf(lv) + lv);
                dmu = -diff / expf(lv);
                dlv = -0.5f * (1.0f - diff * diff * iv);
            } else {
                acc += -0.5f * diff * diff;
                dmu = -diff;
            }
        }
        if (dll) {
            if (ctf) dflt[(long)b * N + j] = dmu;
            else dll[(long)b * N * C + j] = dmu;
            if (noise) dll[(long)b * N * C + N + j] = dlv;
        }
    }
    acc = block_sum256(acc, red);
    if (threadIdx.x == 0) loglik[b] = acc;
    if (ctf && dll) {
        __syncthreads();
        // adjoint of the cross-correlation: dimg[r'][c'] = sum_{u,v} d[r'-u+pad][c'-v+pad] f[u][v]
        for (int j = threadIdx.x; j < N; j += 256) {
            const int r = j / n, cidx = j % n;
            float s = 0.0f;
            for (int u = 0; u < k; ++u) {
                const int rr = r - u + pad;
                if (rr < 0 || rr >= n) continue;
                for (int v = 0; v < k; ++v) {
                    const int cc = cidx - v + pad;
                    if (cc < 0 || cc >= n) continue;
                    s += dflt[(long)b * N + rr * n + cc] * f[u * k + v];
                }
            }
            dll[(long)b * N * C + j] = s;
        }
    }
}

// The same with the CTF cross-correlation and its adjoint running out of LDS: the zero-padded image
// ((n + k - 1)^2 floats; 78 x 78 for 40 x 40 particles with a 39 x 39 filter) and the filter (rows zero-padded to a
// multiple of 4 taps) sit in LDS; a thread produces 4 consecutive outputs of a row, 4 taps at a time, from two 16-byte
// reads of the image row and one (broadcast) of the filter row per 16 multiply-adds, no bounds tests.  Every output
// still accumulates its taps in the reference's (u, v) order and the padding only adds exact zeros, so results equal
// gaussian_kernel's bit for bit.  512 threads: the 400 (row, column group) tasks of a 40 x 40 image run in one round.
constexpr int kCtfThreads = 512;
struct CtfLds {
    int W, Wp, kp, M;  // padded side, its row stride, padded filter row, left margin of the adjoint's image
    __host__ __device__ static CtfLds make(int n, int k) {
        CtfLds g;
        const int pad = k / 2;
        g.W = n + 2 * pad;
        g.kp = (k + 3) & ~3;
        g.M = 3 + ((4 - ((2 * pad) & 3)) & 3);     // >= 3 and (2 pad - 3 + M) % 4 == 0
        g.Wp = ((g.W + g.M + 3) & ~3) + 8;          // room for the margin and for reads past the last tap
        return g;
    }
};
__global__ void __launch_bounds__(kCtfThreads) gaussian_ctf_lds_kernel(const float* __restrict__ yp,
                                                                        const float* __restrict__ target,
                                                                        const uint8_t* __restrict__ mask,
                                                                        const float* __restrict__ ctf, int k,
                                                                        float* __restrict__ loglik, float* __restrict__ dll,
                                                                        float* __restrict__ filt, float* __restrict__ dflt,
                                                                        int N) {
    extern __shared__ __attribute__((aligned(16))) float ctf_lds[];
    __shared__ float red[kCtfThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = (int)(sqrtf((float)N) + 0.5f);
    const CtfLds g = CtfLds::make(n, k);
    const int pad = k / 2, W = g.W, Wp = g.Wp, kp = g.kp;
    float* P = ctf_lds;            // W rows x Wp
    float* F = ctf_lds + W * Wp;   // k rows x kp
    const float* row = yp + (long)b * N;  // C == 1 with a CTF (checked by the caller)
    const float* f = ctf + (long)b * k * k;
    // image at column offset 0 (forward) or M (adjoint); everything else zero
    auto stage = [&](const float* src, int margin) {
        for (int i = tid; i < W * Wp; i += kCtfThreads) {
            const int r = i / Wp - pad, c = i % Wp - margin - pad;
            P[i] = (r >= 0 && r < n && c >= 0 && c < n) ? src[r * n + c] : 0.0f;
        }
    };
    stage(row, 0);
    for (int i = tid; i < k * kp; i += kCtfThreads) {
        const int u = i / kp, v = i % kp;
        F[i] = v < k ? f[u * k + v] : 0.0f;
    }
    __syncthreads();
    const int groups = (n + 3) / 4;
    auto pass = [&](auto flip_tag, float* out) {
        constexpr bool flip = decltype(flip_tag)::value;
        for (int t = tid; t < n * groups; t += kCtfThreads) {
            const int r = t / groups, c0 = (t % groups) * 4;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int u = 0; u < k; ++u) {
                const float* prow = P + (r + (flip ? 2 * pad - u : u)) * Wp;
                const float* frow = F + u * kp;
                for (int v0 = 0; v0 < kp; v0 += 4) {
                    // forward: output i, tap v0+j reads column c0 + i + v0 + j           = s + i + j,     s = c0 + v0
                    // adjoint: output i, tap v0+j reads column c0 + i + 2 pad - v0 - j + M = s + 3 + i - j, s = c0 + 2 pad - v0 - 3 + M
                    const int s0 = flip ? c0 + 2 * pad - v0 - 3 + g.M : c0 + v0;
                    const float4 pa = *reinterpret_cast<const float4*>(prow + s0);
                    const float4 pb = *reinterpret_cast<const float4*>(prow + s0 + 4);
                    const float4 fv = *reinterpret_cast<const float4*>(frow + v0);
                    const float pw[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
                    const float ff[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] += (flip ? pw[3 + i - j] : pw[i + j]) * ff[j];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c0 + i < n) out[r * n + c0 + i] = acc[i];
        }
    };
    pass(std::false_type(), filt + (long)b * N);
    __syncthreads();
    float acc = 0.0f;
    for (int j = tid; j < N; j += kCtfThreads) {
        const bool on = mask ? mask[j] != 0 : true;
        const float diff = filt[(long)b * N + j] - target[(long)b * N + j];
        float dmu = 0.0f;
        if (on) {
            acc += -0.5f * diff * diff;
            dmu = -diff;
        }
        if (dll) dflt[(long)b * N + j] = dmu;
    }
    acc = wave_sum32(acc);
    acc += __shfl_xor(acc, 32);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int i = 0; i < kCtfThreads / 64; ++i) t += red[i];
        loglik[b] = t;
    }
    if (dll) {
        __syncthreads();
        stage(dflt + (long)b * N, g.M);
        __syncthreads();
        pass(std::true_type(), dll + (long)b * N);
    }
}

// ---------------------------------------------------------------- A7: the three scalars of a minibatch and their backward
// out = {elbo, log_p, kl} = {mean(loglik) - mean(kl), mean(loglik), mean(kl)}  (train_mnist.py:81, 86-88); one block,
// fixed summation order.  Replaces four tiny torch kernels forward and five backward.
__global__ void elbo_head_fwd_kernel(const float* __restrict__ loglik, const float* __restrict__ kl, int B, float* __restrict__ out) {
    __shared__ float red[4];
    float a = 0.0f, b = 0.0f;
    for (int i = threadIdx.x; i < B; i += 256) {
        a += loglik[i];
        b += kl[i];
    }
    a = block_sum256(a, red);
    b = block_sum256(b, red);
    if (threadIdx.x == 0) {
        const float lp = a / (float)B, k = b / (float)B;
        out[0] = lp - k;
        out[1] = lp;
        out[2] = k;
    }
}
// d/d loglik[b] = (g_elbo + g_logp) / B, d/d kl[b] = (g_kl - g_elbo) / B; absent upstream gradients are null
__global__ void elbo_head_bwd_kernel(const float* __restrict__ g_elbo, const float* __restrict__ g_logp,
                                     const float* __restrict__ g_kl, int B, float* __restrict__ dloglik, float* __restrict__ dkl) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float ge = g_elbo ? g_elbo[0] : 0.0f, gl = g_logp ? g_logp[0] : 0.0f, gk = g_kl ? g_kl[0] : 0.0f;
    dloglik[i] = (ge + gl) / (float)B;
    dkl[i] = (gk - ge) / (float)B;
}

// ---------------------------------------------------------------- A7: latent head (reparameterise, split, KL)
struct LatentGeo {
    int B, inf, rotate, translate, mu_penalty;
    float dx_scale, z_scale, theta_prior;
};

// one thread per image: z = exp(logstd)*r + mu; theta / dx / content; kl[b]
__global__ void latent_fwd_kernel(const float* __restrict__ q, const float* __restrict__ r, float* __restrict__ theta,
                                  float* __restrict__ dx, float* __restrict__ zc, float* __restrict__ kl, LatentGeo g) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= g.B) return;
    const float* mu = q + (long)b * 2 * g.inf;
    const float* ls = mu + g.inf;
    const int off = g.rotate ? 1 : 0, c0 = off + (g.translate ? 2 : 0), zd = g.inf - c0;
    float k = 0.0f;
    for (int j = 0; j < g.inf; ++j) {
        const float sd = expf(ls[j]);
        const float z = sd * r[(long)b * g.inf + j] + mu[j];
        if (j < off) {
            theta[b] = z;
            const float s = g.theta_prior;
            k += -ls[j] + logf(s) + (sd * sd + (g.mu_penalty ? mu[j] * mu[j] : 0.0f)) / 2.0f / (s * s) - 0.5f;
        } else {
            k += -ls[j] + 0.5f * sd * sd + 0.5f * mu[j] * mu[j] - 0.5f;
            if (j < c0) dx[2 * b + (j - off)] = z * g.dx_scale;
            else zc[(long)b * zd + (j - c0)] = z * g.z_scale;
        }
    }
    kl[b] = k;
}

// one thread per (image, latent): d(loss)/d(mu), d(loss)/d(logstd)
__global__ void latent_bwd_kernel(const float* __restrict__ q, const float* __restrict__ r,
                                  const float* __restrict__ g_theta, const float* __restrict__ g_dx,
                                  const float* __restrict__ g_zc, const float* __restrict__ g_kl, float* __restrict__ gq,
                                  LatentGeo g) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)g.B * g.inf) return;
    const int j = idx % g.inf, b = idx / g.inf;
    const float* mu = q + (long)b * 2 * g.inf;
    const float* ls = mu + g.inf;
    const int off = g.rotate ? 1 : 0, c0 = off + (g.translate ? 2 : 0), zd = g.inf - c0;
    const float sd = expf(ls[j]);
    const float gk = g_kl ? g_kl[b] : 0.0f;
    float dz, dmu, dls;
    if (j < off) {
        const float s2 = g.theta_prior * g.theta_prior;
        dz = g_theta ? g_theta[b] : 0.0f;
        dmu = g.mu_penalty ? gk * mu[j] / s2 : 0.0f;
        dls = gk * (-1.0f + sd * sd / s2);
    } else {
        if (j < c0) dz = g_dx ? g_dx[2 * b + (j - off)] * g.dx_scale : 0.0f;
        else dz = g_zc ? g_zc[(long)b * zd + (j - c0)] * g.z_scale : 0.0f;
        dmu = gk * mu[j];
        dls = gk * (-1.0f + sd * sd);
    }
    gq[(long)b * 2 * g.inf + j] = dz + dmu;
    gq[(long)b * 2 * g.inf + g.inf + j] = dz * r[idx] * sd + dls;
}

// ---------------------------------------------------------------- rotation augmentation (train_galaxy.py:41-54, train_particles.py:31-43)
// Pillow's Image.rotate(angle, resample=BICUBIC) for a batch of square-or-not images resident in HBM.  The host supplies,
// per image, the six inverse-affine coefficients exactly as PIL/Image.py computes them (doubles) or an exact quarter-turn
// code; the resampler below follows libImaging/Geometry.c (affine_transform + bicubic_filter8/32RGB/32F) operation by
// operation in doubles with contraction off, so results are bit-identical to Pillow's (oracle/pil_rotate.py is the CPU
// restatement it is tested against).  U8 = the galaxy/MNIST path: samples are (uint8)(y*255), the result is clamped to
// [0,255], truncated and returned as (double)q/255 rounded to float, as the reference's astype chain does.
struct RotGeo {
    int B, rows, cols, C;
};

template <bool U8>
__global__ void rotate_bicubic_kernel(const float* __restrict__ y, float* __restrict__ out, const double* __restrict__ mat,
                                      const int* __restrict__ quarter, RotGeo g) {
#pragma clang fp contract(off)
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)g.rows * g.cols * g.C;
    if (t >= per * g.B) return;
    const int b = (int)(t / per);
    const int rem = (int)(t - (long)b * per);
    const int c = rem % g.C;
    const int px = (rem / g.C) % g.cols, py = rem / (g.C * g.cols);
    const float* img = y + (long)b * per;
    const int w = g.cols, h = g.rows;
    auto sample = [&](int yy, int xx) -> float { return img[((long)yy * w + xx) * g.C + c]; };
    const int q = quarter[b];
    if (q >= 0) {  // exact multiples of 90 degrees: copy / transpose (Image.rotate fast paths), counter-clockwise
        int sy = py, sx = px;
        if (q == 1) { sy = px; sx = w - 1 - py; }
        if (q == 2) { sy = h - 1 - py; sx = w - 1 - px; }
        if (q == 3) { sy = h - 1 - px; sx = py; }
        const float v = sample(sy, sx);
        out[t] = U8 ? (float)((double)(unsigned char)(v * 255.0f) / 255.0) : v;
        return;
    }
    const double* a = mat + (long)b * 6;
    double xin = (double)px + 0.5, yin = (double)py + 0.5;
    const double xo = a[0] * xin + a[1] * yin + a[2];
    const double yo = a[3] * xin + a[4] * yin + a[5];
    if (xo < 0.0 || xo >= (double)w || yo < 0.0 || yo >= (double)h) {
        out[t] = 0.0f;
        return;
    }
    xin = xo - 0.5;
    yin = yo - 0.5;
    int x = xin < 0.0 ? (int)floor(xin) : (int)xin;
    int yy = yin < 0.0 ? (int)floor(yin) : (int)yin;
    const double dx = xin - x, dy = yin - yy;
    --x;
    --yy;
    int xc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xc[k] = min(max(x + k, 0), w - 1);
    auto horner = [](double p1, double p2, double p3, double p4, double d) { return p1 + d * (p2 + d * (p3 + d * p4)); };
    auto row = [&](int r) -> double {
        if (U8) {  // integer coefficient arithmetic, as C promotes UINT8 operands
            int s[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = (int)(unsigned char)(sample(r, xc[k]) * 255.0f);
            return horner((double)s[1], (double)(-s[0] + s[2]), (double)(2 * (s[0] - s[1]) + s[2] - s[3]),
                          (double)(-s[0] + s[1] - s[2] + s[3]), dx);
        }
        float s[4];  // FLOAT32 operands: the coefficient expressions are float arithmetic
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = sample(r, xc[k]);
        const float p2 = -s[0] + s[2], p3 = 2.0f * (s[0] - s[1]) + s[2] - s[3], p4 = -s[0] + s[1] - s[2] + s[3];
        return horner((double)s[1], (double)p2, (double)p3, (double)p4, dx);
    };
    const double v1 = row(min(max(yy, 0), h - 1));
    const double v2 = (yy + 1 >= 0 && yy + 1 < h) ? row(yy + 1) : v1;
    const double v3 = (yy + 2 >= 0 && yy + 2 < h) ? row(yy + 2) : v2;
    const double v4 = (yy + 3 >= 0 && yy + 3 < h) ? row(yy + 3) : v3;
    const double v = horner(v2, -v1 + v3, 2 * (v1 - v2) + v3 - v4, -v1 + v2 - v3 + v4, dy);
    if (U8) {
        const double qv = v <= 0.0 ? 0.0 : (v >= 255.0 ? 255.0 : (double)(unsigned char)v);
        out[t] = (float)(qv / 255.0);
    } else {
        out[t] = (float)v;
    }
}

// ---------------------------------------------------------------- CTF filter bank (spatial_vae/ctf.py:7-56)
// One workgroup per particle: evaluate the closed-form 2-D CTF on the FFT frequency grid (n x m), bring it to real space
// with a separable inverse DFT (rows, then columns; doubles throughout, the sizes are ~39 x 39), fftshift, negate, store
// as float.  params row = [defocus um, cs mm, voltage kV, apix A, bfactor, ampcont %, dfdiff, dfang deg]
// (ctf.py:29); as in the reference dfdiff is unused: both defoci are defocus*10000 (ctf.py:47-48).
// LDS: c (n*m doubles) | T (n*m double2) | twiddles of length m and n (double2 each).  SCRATCH = true (filters above
// ~80 x 80, whose 24 n m bytes exceed the 160 KiB of LDS): c and T live in a per-workgroup slice of a caller-provided
// global scratch area instead (only the twiddles stay in LDS) and the grid strides over the particles; same arithmetic
// in the same order, so both forms give identical bits.
template <bool SCRATCH>
__global__ void __launch_bounds__(256) ctf_filter_kernel(const double* __restrict__ params, float* __restrict__ out, int count,
                                                          int n, int m, double scale, double* __restrict__ scratch) {
#pragma clang fp contract(off)
    extern __shared__ double lds_ctf[];
    double* c = SCRATCH ? scratch + (long)blockIdx.x * 3 * n * m : lds_ctf;
    double2* T = reinterpret_cast<double2*>(c + (long)n * m);
    double2* wm = SCRATCH ? reinterpret_cast<double2*>(lds_ctf) : T + (long)n * m;
    double2* wn = wm + m;
    for (int particle = blockIdx.x; particle < count; particle += gridDim.x) {
    __syncthreads();   // the previous particle's last pass has finished reading T and the twiddles
    const double* p = params + (long)particle * 8;
    const double apix = p[3] * scale;
    const double dfu = p[0] * 10000.0, dfv = p[0] * 10000.0;
    const double dfang = 2.0 * M_PI * p[7] / 360.0;
    const double volt = p[2] * 1000.0, cs = p[1] * 1e7, w = p[5] / 100.0, bfactor = p[4];
    const double lam = 12.2639 / sqrt(volt + 0.97845e-6 * volt * volt);
    const double amp = sqrt(1.0 - w * w);
    for (int k = threadIdx.x; k < m; k += 256) {
        double sn, cn;
        sincospi(2.0 * k / m, &sn, &cn);
        wm[k] = make_double2(cn, sn);
    }
    for (int k = threadIdx.x; k < n; k += 256) {
        double sn, cn;
        sincospi(2.0 * k / n, &sn, &cn);
        wn[k] = make_double2(cn, sn);
    }
    for (int e = threadIdx.x; e < n * m; e += 256) {
        const int a = e / m, b = e - a * m;
        // np.fft.fftfreq: 0, 1, ..., (len-1)/2, -(len/2), ..., -1, all over len; then / apix
        const double fx = (double)(a <= (n - 1) / 2 ? a : a - n) / n / apix;   // ctf.py: x = first (row) frequency
        const double fy = (double)(b <= (m - 1) / 2 ? b : b - m) / m / apix;
        const double s2 = fx * fx + fy * fy;
        const double df = 0.5 * (dfu + dfv + (dfu - dfv) * cos(2.0 * (atan2(fy, fx) - dfang)));
        const double gamma = 2.0 * M_PI * (-0.5 * df * lam * s2 + 0.25 * cs * lam * lam * lam * s2 * s2);
        c[e] = (amp * sin(gamma) - w * cos(gamma)) * exp(-bfactor / 4.0 * s2);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n * m; e += 256) {  // T[a][v] = sum_b c[a][b] e^{+2 pi i b v / m}
        const int a = e / m, v = e - a * m;
        double re = 0.0, im = 0.0;
        int k = 0;
        for (int b = 0; b < m; ++b) {
            const double cv = c[a * m + b];
            re += cv * wm[k].x;
            im += cv * wm[k].y;
            k += v;
            if (k >= m) k -= m;
        }
        T[e] = make_double2(re, im);
    }
    __syncthreads();
    const double inv = 1.0 / ((double)n * m);
    float* o = out + (long)particle * n * m;
    for (int e = threadIdx.x; e < n * m; e += 256) {  // out[i][j] = -Re X[(i + (n+1)/2) % n][(j + (m+1)/2) % m]
        const int i = e / m, j = e - i * m;
        const int u = (i + (n + 1) / 2) % n, v = (j + (m + 1) / 2) % m;
        double re = 0.0;
        int k = 0;
        for (int a = 0; a < n; ++a) {
            const double2 t = T[a * m + v];
            re += t.x * wn[k].x - t.y * wn[k].y;
            k += u;
            if (k >= n) k -= n;
        }
        o[e] = (float)(-(re * inv));
    }
    }
}

// ---------------------------------------------------------------- Adam over a flat buffer (A7: optim.step())
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float step_size, float sqrt_bc2, float b1, float b2, float eps, int zero_grad) {
    // operation order of ATen's Adam: denom = sqrt(v) / sqrt(bc2) + eps;  p += (-step_size) * (m / denom)
    // zero_grad: the gradient is cleared behind the update (optim.zero_grad(), train_mnist.py:150) -- no separate pass
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 g4 = *reinterpret_cast<const float4*>(g + i);
        float4 m4 = *reinterpret_cast<float4*>(m + i), v4 = *reinterpret_cast<float4*>(v + i), p4 = *reinterpret_cast<float4*>(p + i);
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mm[e] = b1 * mm[e] + (1.0f - b1) * gg[e];
            vv[e] = b2 * vv[e] + (1.0f - b2) * gg[e] * gg[e];
            pp[e] += -step_size * (mm[e] / (sqrtf(vv[e]) / sqrt_bc2 + eps));
        }
        *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        for (long j = i; j < n; ++j) {
            const float gj = g[j];
            const float mj = b1 * m[j] + (1.0f - b1) * gj, vj = b2 * v[j] + (1.0f - b2) * gj * gj;
            m[j] = mj;
            v[j] = vj;
            p[j] += -step_size * (mj / (sqrtf(vj) / sqrt_bc2 + eps));
            if (zero_grad) g[j] = 0.0f;
        }
    }
}

}  // namespace svae
