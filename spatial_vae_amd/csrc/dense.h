// fp32-MFMA GEMM kernels of the decoder's hidden H x H layers (SURVEY.md 8a rows A3 and A8;
// reference: nn.Linear + activation in SpatialGenerator.layers, spatial_vae/models.py:77-83, 126,
// ResidLinear models.py:13-21, and their autograd backward).
//
// All three GEMMs of a hidden layer run on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain,
// 64 FLOP/clk/SIMD, peak 157.3 TFLOP/s):
//   forward        a_l[m][n]      = act( sum_k a_{l-1}[m][k] W[n][k] + b[n] (+ a_{l-1}[m][n]) )
//   data gradient  dh_{l-1}[m][k] = ( sum_n dh_l[m][n] W[n][k] (+ dh_l[m][k]) ) * act'(a_{l-1}[m][k])
//   weight grad.   dW[n][k]       = sum_m dh_l[m][n] a_{l-1}[m][k],   db[n] = sum_m dh_l[m][n]
// Activations and gradients live in HBM in the octet-major layout of common.h.
#pragma once
#include "common.h"

namespace svae {

// ------------------------------------------------------------------------------------------
// dense_kernel: OUT(Mp x Hp) = epilogue( IN(Mp x Hp) * Wp ), one 32-row tile per wave.
//
// MFMA roles (32x32x2: A lane l -> A[i=l&31][k=l>>5], B lane l -> B[k=l>>5][j=l&31]):
//   A = row operand: lane (m = l&31, h = l>>5) supplies IN[m][8g + 4h + e] at step (g, e);
//       one global dword per step, prefetched one octet (4 steps) ahead.
//   B = packed weights from LDS: lane (n = l&31, h) reads the 16 bytes
//       Wp[g][n][4h .. 4h+3] with one ds_read_b128 per 4 steps and column tile: the wave's 64
//       lanes read 1 KiB contiguous, conflict-free.
//   D: lane (n, h) holds rows m = 8q + 4h + r (q = reg>>2, r = reg&3) of column n, i.e. four
//       consecutive rows per register quad = one 16-byte octet-major store.
// A workgroup is 4 waves = 4 consecutive row tiles sharing the weight chunks, which stream
// L2 -> LDS with global_load_lds (no registers), double-buffered, one barrier per chunk.
// NT column tiles (NB = 32*NT columns) are accumulated at a time: NT = 16 keeps a full
// 512-wide layer in 256 accumulator registers at one wave per SIMD.
// ------------------------------------------------------------------------------------------
struct DenseArgs {
    const float* in;    // row operand, octet-major (Mp x Hp)
    const float* wp;    // packed weights [Hp/8][Hp][8] (contraction index in the octet)
    float* out;         // octet-major (Mp x Hp)
    const float* bias;  // forward: (H) bias; data gradient: unused
    const float* aux;   // data gradient: a_{l-1} octet-major (its act' multiplies the result)
    long tiles;         // Mp/32
    int Hp;
    int H;
    int act;
    int resid;
};

template <int ACT, bool DGRAD>
__device__ __forceinline__ float4 dense_epilogue(float4 v, float bias, const float* resid_ptr, const float* aux_ptr) {
    if (resid_ptr) {
        const float4 r = *reinterpret_cast<const float4*>(resid_ptr);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (DGRAD) {
        const float4 a = *reinterpret_cast<const float4*>(aux_ptr);
        v.x *= act_grad<ACT>(a.x); v.y *= act_grad<ACT>(a.y);
        v.z *= act_grad<ACT>(a.z); v.w *= act_grad<ACT>(a.w);
    } else {
        v.x = act_fwd<ACT>(v.x + bias); v.y = act_fwd<ACT>(v.y + bias);
        v.z = act_fwd<ACT>(v.z + bias); v.w = act_fwd<ACT>(v.w + bias);
    }
    return v;
}

template <int NT, bool DGRAD>
__global__ __launch_bounds__(256, (NT == 16 ? 1 : 2)) void dense_kernel(DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NB = NT * 32;            // columns per accumulation block
    constexpr int G = (NT == 16) ? 2 : 4;  // contraction octets per LDS chunk
    constexpr int CHUNK = G * NB * 8;      // floats per LDS buffer
    constexpr int NINSTR = G * NT;         // 1 KiB global_load_lds wave-instructions per chunk

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const long tile = (long)blockIdx.x * 4 + wave;
    const bool live = tile < a.tiles;
    const long tl = live ? tile : a.tiles - 1;  // dead waves recompute the last tile and store nothing
    const int Hp = a.Hp;
    const int noct = Hp / 8;
    const int nchunk = noct / G;

    // this lane's row (m = nl) of the row operand: element (m, k) at ((m>>3)*Hp + k)*8 + (m&7)
    const float* arow = a.in + ((tl * 4 + (nl >> 3)) * (long)Hp + 4 * h) * 8 + (nl & 7);

    for (int nb = 0; nb < Hp / NB; ++nb) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        auto stage = [&](int c, int buf) {
#pragma unroll
            for (int j = 0; j < (NINSTR + 3) / 4; ++j) {
                const int idx = wave + 4 * j;
                if (idx < NINSTR) {
                    const int gl = idx / NT, tt = idx % NT;
                    const float* src = a.wp + (((long)(c * G + gl) * Hp + nb * NB + tt * 32) * 8) + lane * 4;
                    float* dst = smem + buf * CHUNK + (gl * NB + tt * 32) * 8;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
            }
        };

        stage(0, 0);
        float av[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = arow[e * 8];

        for (int c = 0; c < nchunk; ++c) {
            __syncthreads();  // chunk c has landed (vmcnt drained) and buffer (c+1)&1 is free again
            if (c + 1 < nchunk) stage(c + 1, (c + 1) & 1);
            const float* bbuf = smem + (c & 1) * CHUNK + nl * 8 + 4 * h;
#pragma unroll
            for (int gl = 0; gl < G; ++gl) {
                const int gnext = c * G + gl + 1;
                float an[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) an[e] = (gnext < noct) ? arow[(long)gnext * 64 + e * 8] : 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 b = *reinterpret_cast<const float4*>(bbuf + (gl * NB + t * 32) * 8);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], b.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], b.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], b.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], b.w, acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = an[e];
            }
        }

        // ---- epilogue: bias/activation (forward) or act' of the previous layer (data gradient)
        auto epi = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = nb * NB + t * 32 + nl;
                float bias = 0.0f;
                if (!DGRAD) bias = (n < a.H) ? a.bias[n] : 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long off = ((tl * 4 + q) * (long)Hp + n) * 8 + 4 * h;
                    float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
                    v = dense_epilogue<ACT, DGRAD>(v, bias, a.resid ? a.in + off : nullptr, DGRAD ? a.aux + off : nullptr);
                    if (live) *reinterpret_cast<float4*>(a.out + off) = v;
                }
            }
        };
        switch (a.act) {
            case SVAE_ACT_TANH: epi(std::integral_constant<int, SVAE_ACT_TANH>()); break;
            case SVAE_ACT_LEAKYRELU: epi(std::integral_constant<int, SVAE_ACT_LEAKYRELU>()); break;
            case SVAE_ACT_RELU: epi(std::integral_constant<int, SVAE_ACT_RELU>()); break;
            default: epi(std::integral_constant<int, SVAE_ACT_SIGMOID>()); break;
        }
        __syncthreads();  // every wave is done with the LDS buffers before the next block restages buffer 0
    }
}

// ------------------------------------------------------------------------------------------
// wgrad_kernel: partial dW over a range of row octets.
//   D[i = n][j = k] += sum_m A[n][m] B[m][k],  A = dh_l^T, B = a_{l-1}; both operands are read as
//   16-byte octet-major vectors (lane = feature, 4 consecutive rows = 4 k-steps), 1 KiB
//   contiguous per wave instruction, straight into registers and double-buffered: no LDS.
// A workgroup owns a 256 x 256 block of dW (4 waves x (4 x 4) tiles of 32 x 32 = 256
// accumulator registers per lane) and the row range of split blockIdx.y; partial blocks go to
// slab[split] and are summed in fixed order by wgrad_reduce_kernel (deterministic, no atomics).
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* dh;     // octet-major (Mp x Hp): dh_l
    const float* aprev;  // octet-major (Mp x Hp): a_{l-1}
    float* slab;         // [S][Hp][Hp]
    float* bslab;        // [S][2][Hp] partial bias gradients (two half-waves)
    long noct;
    int Hp;
    int nblk1;           // blocks per side = ceil(ntile / 8)
};

__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int Hp = a.Hp, ntile = Hp / 32;
    const int bi = blockIdx.x / a.nblk1, bj = blockIdx.x % a.nblk1;
    const int ibase = bi * 8 + (wave >> 1) * 4, jbase = bj * 8 + (wave & 1) * 4;
    const long S = gridDim.y;
    const long per = (a.noct + S - 1) / S;
    const long o0 = blockIdx.y * per;
    const long o1 = (o0 + per < a.noct) ? o0 + per : a.noct;

    const float* pa[4];
    const float* pb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int it = (ibase + t < ntile) ? ibase + t : ntile - 1;
        const int jt = (jbase + t < ntile) ? jbase + t : ntile - 1;
        pa[t] = a.dh + ((long)it * 32 + nl) * 8 + 4 * h;
        pb[t] = a.aprev + ((long)jt * 32 + nl) * 8 + 4 * h;
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    float4 ac[4], bc[4];
    const long ostride = (long)Hp * 8;
    if (o0 < o1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ac[t] = *reinterpret_cast<const float4*>(pa[t] + o0 * ostride);
            bc[t] = *reinterpret_cast<const float4*>(pb[t] + o0 * ostride);
        }
    }
    for (long o = o0; o < o1; ++o) {
        float4 an[4], bn[4];
        const long on = (o + 1 < o1) ? o + 1 : o;  // last iteration reloads the same octet (harmless)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            an[t] = *reinterpret_cast<const float4*>(pa[t] + on * ostride);
            bn[t] = *reinterpret_cast<const float4*>(pb[t] + on * ostride);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i].x, bc[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i].y, bc[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i].z, bc[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i].w, bc[j].w, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bs[t] += (ac[t].x + ac[t].y) + (ac[t].z + ac[t].w);
            ac[t] = an[t];
            bc[t] = bn[t];
        }
    }

    float* slab = a.slab + (long)blockIdx.y * Hp * Hp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (ibase + i >= ntile) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (jbase + j >= ntile) continue;
            const int k = (jbase + j) * 32 + nl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (ibase + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(long)n * Hp + k] = acc[i][j][r];
            }
        }
    }
    if (bj == 0 && (wave & 1) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ibase + i < ntile) a.bslab[((long)blockIdx.y * 2 + h) * Hp + (ibase + i) * 32 + nl] = bs[i];
    }
}

// dW[n][k] = sum_s slab[s][n][k] (n, k < H), db[n] = sum_s sum_half bslab[s][half][n]
__global__ void wgrad_reduce_kernel(const float* slab, const float* bslab, float* dW, float* db, int H, int Hp, int S) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < (long)H * H) {
        const int n = idx / H, k = idx % H;
        float s = 0.0f;
        for (int i = 0; i < S; ++i) s += slab[((long)i * Hp + n) * Hp + k];
        if (dW) dW[idx] = s;
    }
    if (idx < H && db) {
        float s = 0.0f;
        for (int i = 0; i < 2 * S; ++i) s += bslab[(long)i * Hp + idx];
        db[idx] = s;
    }
}

}  // namespace svae
