// fp16x3 split-operand GEMM path (opt-in: SVAE_GEMM=fp16x3) -- fp32-accurate products on the f16 matrix pipe.
//
// Every fp32 operand x is carried as two halfs  x ~= hi + lo / 2048,  hi = half(x), lo = half((x - hi) * 2048), and a
// product a*w is formed as  hi_a*hi_w + (hi_a*lo_w + lo_a*hi_w) / 2048  with fp32 accumulation in the MFMA: the dropped
// lo*lo term is 2^-22 relative, so the result is as accurate as an fp32 GEMM (measured: same 5e-7 relative error against
// fp64 as the fp32 MFMA path, tools/split_numerics.py), while v_mfma_f32_32x32x16_f16 runs at 16x the rate of
// v_mfma_f32_32x32x2_f32: three of them per K=16 step cost 96 cycles where the fp32 form needs 512.
// Operands must be bounded (|x| < 65504): the path is used after tanh / sigmoid activations only.
//
// Layouts (all 16-byte = 8-half fragments, addressed in uint4 units):
//   rows  As[T][kc][part][lane]   T = 32-row tile, kc = 16-feature step, part 0 = hi / 1 = lo, lane = MFMA lane
//                                 (r = lane & 31 the row, h = lane >> 5): features 16 kc + 8 h + (0..7) of row 32 T + r.
//                                 One (T, kc, part) block is 1 KiB: a wave loads it with one global_load_dwordx4.
//   weights Ws[kc][nt][part][lane] nt = 32-column tile: contraction indices 16 kc + 8 h + (0..7) of column 32 nt + r.
//                                 One block is 1 KiB and goes to LDS verbatim: ds_read_b128 is lane-linear, conflict-free.
#pragma once
#include "common.h"

namespace svae {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kLoScale = 2048.0f;
constexpr float kLoInv = 1.0f / 2048.0f;

union Frag {  // 8 halfs <-> 16 bytes
    f16x8 h;
    uint4 u;
};

__device__ __forceinline__ void split8(const float (&x)[8], uint4& hi, uint4& lo) {
    Frag fh, fl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 hj = (_Float16)x[j];
        fh.h[j] = hj;
        fl.h[j] = (_Float16)((x[j] - (float)hj) * kLoScale);
    }
    hi = fh.u;
    lo = fl.u;
}

// W (H x H, row-major [out n][in k]) -> Ws for the forward contraction over k: B[k][n] = W[n][k].
// transpose != 0 packs the data-gradient form instead: contraction over n, B[n][k] = W[n][k].
__global__ void split_weights_kernel(const float* __restrict__ W, uint4* __restrict__ ws, int H, int Hp, int transpose) {
    const int ntile = Hp / 32, KC = Hp / 16;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (kc, nt, lane)
    if (idx >= (long)KC * ntile * 64) return;
    const int lane = idx & 63, nt = (int)((idx >> 6) % ntile), kc = (int)((idx >> 6) / ntile);
    const int col = nt * 32 + (lane & 31), c0 = kc * 16 + 8 * (lane >> 5);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int con = c0 + j;
        const int n = transpose ? con : col, k = transpose ? col : con;
        x[j] = (n < H && k < H) ? W[(long)n * H + k] : 0.0f;
    }
    uint4 hi, lo;
    split8(x, hi, lo);
    const long blk = ((long)kc * ntile + nt) * 2;
    ws[blk * 64 + lane] = hi;
    ws[(blk + 1) * 64 + lane] = lo;
}

// fp32 octet-major plane (element (m, k) at ((m>>3)*Hp + k)*8 + (m&7)) -> As
__global__ void split_rows_kernel(const float* __restrict__ in, uint4* __restrict__ as, long tiles, int Hp) {
    const int KC = Hp / 16;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (T, kc, lane)
    if (idx >= tiles * KC * 64) return;
    const int lane = idx & 63;
    const int kc = (int)((idx >> 6) % KC);
    const long T = (idx >> 6) / KC;
    const long m = T * 32 + (lane & 31);
    const int k0 = kc * 16 + 8 * (lane >> 5);
    const float* p = in + ((m >> 3) * Hp + k0) * 8 + (m & 7);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p[j * 8];
    uint4 hi, lo;
    split8(x, hi, lo);
    const long blk = (T * KC + kc) * 2;
    as[blk * 64 + lane] = hi;
    as[(blk + 1) * 64 + lane] = lo;
}

struct SplitArgs {
    const uint4* as;     // row operand, split
    const uint4* ws;     // weights, split
    float* out;          // fp32 octet-major result
    const float* bias;   // (H)
    const float* resid;  // fp32 octet-major a_{l-1} (RESID)
    long tiles;
    int Hp, H, act;
    // CF: partial logits of the output layer, as in dense_kernel
    const float* out_w;
    float* lpart;
    long Mp;
};

constexpr int kSplitG = 2;  // K=16 steps per LDS chunk

template <int NT>
struct SplitCfg {
    static constexpr int BLOCKS = kSplitG * NT * 2;        // 1 KiB blocks per chunk
    static constexpr int LDS_BYTES = 2 * BLOCKS * 1024;    // double-buffered
    static constexpr int PER_THREAD = BLOCKS * 64 / 256;   // uint4 per thread per chunk
};

// out(32 rows x NT*32 cols per wave) = act( As * Ws + bias [+ resid] ); 4 waves (4 row tiles) share the weight chunks.
template <int NT, bool RESID, int CF>
__global__ __launch_bounds__(256, 2) void dense_split_fwd_kernel(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    using Cfg = SplitCfg<NT>;
    constexpr int G = kSplitG, BLOCKS = Cfg::BLOCKS, PT = Cfg::PER_THREAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 31, h = lane >> 5;
    const long tile = (long)blockIdx.x * 4 + wave;
    const bool live = tile < a.tiles;
    const long tl = live ? tile : a.tiles - 1;
    const int Hp = a.Hp, KC = Hp / 16, ntile = Hp / 32;
    const int nb = blockIdx.y;  // column block of NT tiles
    const int nchunk = KC / G;

    const uint4* ap = a.as + (tl * KC) * 2 * 64 + lane;
    // chunk c, block (g, t, part) lives at ws block ((c*G + g)*ntile + nb*NT + t)*2 + part; thread moves uint4 i*256+tid
    auto wsrc = [&](int c, int i) -> const uint4* {
        const int e = i * 256 + tid, blk = e >> 6, within = e & 63;
        const int part = blk & 1, t = (blk >> 1) % NT, g = (blk >> 1) / NT;
        return a.ws + ((((long)(c * G + g) * ntile + nb * NT + t) * 2 + part) * 64 + within);
    };

    f32x16 acc[NT], accx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; accx[t][r] = 0.0f; }

    uint4 stage[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) stage[i] = *wsrc(0, i);
#pragma unroll
    for (int i = 0; i < PT; ++i) smem4[i * 256 + tid] = stage[i];
    Frag ah, al;
    ah.u = ap[0];
    al.u = ap[64];
    __syncthreads();

    for (int c = 0; c < nchunk; ++c) {
        const uint4* buf = smem4 + (c & 1) * (BLOCKS * 64);
        const int cn = c + 1 < nchunk ? c + 1 : c;  // the last iteration re-fetches its own chunk (never used)
#pragma unroll
        for (int i = 0; i < PT; ++i) stage[i] = *wsrc(cn, i);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int kc = c * G + g;
            const int kn = kc + 1 < KC ? kc + 1 : kc;
            Frag nh, nlo;
            nh.u = ap[(long)kn * 128];
            nlo.u = ap[(long)kn * 128 + 64];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                Frag bh, bl;
                bh.u = buf[((g * NT + t) * 2) * 64 + lane];
                bl.u = buf[((g * NT + t) * 2 + 1) * 64 + lane];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bh.h, acc[t], 0, 0, 0);
                accx[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bl.h, accx[t], 0, 0, 0);
                accx[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, bh.h, accx[t], 0, 0, 0);
            }
            ah = nh;
            al = nlo;
        }
        uint4* nbuf = smem4 + ((c + 1) & 1) * (BLOCKS * 64);
#pragma unroll
        for (int i = 0; i < PT; ++i) nbuf[i * 256 + tid] = stage[i];
        __syncthreads();
    }

    if (!live) return;
    auto epi = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr int NB = NT * 32;
        constexpr int CFN = CF > 0 ? CF : 1;
        float wo[CFN][NT], lp[CFN][16], bias[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = nb * NB + t * 32 + nl;
            const float bv = a.bias[n < a.H ? n : a.H - 1];
            bias[t] = (n < a.H) ? bv : 0.0f;
#pragma unroll
            for (int c = 0; c < CFN; ++c) {
                const float wv = CF > 0 ? a.out_w[c * a.H + (n < a.H ? n : a.H - 1)] : 0.0f;
                wo[c][t] = (n < a.H) ? wv : 0.0f;
            }
        }
#pragma unroll
        for (int c = 0; c < CFN; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) lp[c][r] = 0.0f;
        const long off0 = (tl * 4 * (long)Hp + nb * NB + nl) * 8 + 4 * h;
        const long qstride = (long)Hp * 8;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long off = off0 + q * qstride + (long)t * 32 * 8;
                float4 v = make_float4(acc[t][4 * q] + accx[t][4 * q] * kLoInv, acc[t][4 * q + 1] + accx[t][4 * q + 1] * kLoInv,
                                       acc[t][4 * q + 2] + accx[t][4 * q + 2] * kLoInv, acc[t][4 * q + 3] + accx[t][4 * q + 3] * kLoInv);
                if (RESID) {
                    const float4 fr = *reinterpret_cast<const float4*>(a.resid + off);
                    v.x += fr.x; v.y += fr.y; v.z += fr.z; v.w += fr.w;
                }
                v.x = act_fwd<ACT>(v.x + bias[t]); v.y = act_fwd<ACT>(v.y + bias[t]);
                v.z = act_fwd<ACT>(v.z + bias[t]); v.w = act_fwd<ACT>(v.w + bias[t]);
                *reinterpret_cast<float4*>(a.out + off) = v;
                if (CF > 0) {
#pragma unroll
                    for (int c = 0; c < CFN; ++c) {
                        lp[c][4 * q] += v.x * wo[c][t]; lp[c][4 * q + 1] += v.y * wo[c][t];
                        lp[c][4 * q + 2] += v.z * wo[c][t]; lp[c][4 * q + 3] += v.w * wo[c][t];
                    }
                }
            }
        }
        if (CF > 0) {
#pragma unroll
            for (int c = 0; c < CFN; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) lp[c][r] = half_sum_dpp_hi(lp[c][r]);
            if (nl == 31) {
#pragma unroll
                for (int c = 0; c < CFN; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(a.lpart + ((long)nb * CFN + c) * a.Mp + tl * 32 + 8 * q + 4 * h) =
                            make_float4(lp[c][4 * q], lp[c][4 * q + 1], lp[c][4 * q + 2], lp[c][4 * q + 3]);
            }
        }
    };
    if (a.act == SVAE_ACT_TANH) epi(std::integral_constant<int, SVAE_ACT_TANH>());
    else epi(std::integral_constant<int, SVAE_ACT_SIGMOID>());
}

}  // namespace svae
