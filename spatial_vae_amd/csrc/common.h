// Shared device/host definitions for the gfx950 spatial-VAE decoder kernels.
//
// ROW SPACE.  Every image is padded to whole 32-row tiles: Npad = 32*ceil(N/32) and the
// padded row index of pixel i of image b is  mp = b*Npad + i.  A 32-row MFMA tile therefore
// never straddles two images, so the per-image quantities (pose, latent projection) are
// wave-uniform.  Pad rows carry finite garbage forward and exact zeros backward (their
// upstream gradient is zero), so they never reach a result.
//
// ACTIVATION LAYOUT ("octet-major").  An (Mp x F) activation / gradient tensor is stored as
// [Mp/8][F][8]: element (m, f) lives at ((m>>3)*F + f)*8 + (m&7).  F = Hp = H rounded up
// to 32.  Why: in the 32x32x2 fp32 MFMA a lane supplies ONE value per k-step, and a 32x32
// accumulator holds, per lane, four consecutive rows of one column.  With rows grouped by 8
//   * the GEMM epilogues store their accumulators as 16-byte vectors, 1 KiB contiguous per
//     wave instruction (lane = column, the 4 consecutive rows are the vector);
//   * the weight-gradient GEMM, which contracts over rows, loads both operands the same way
//     (16-byte vectors, 1 KiB contiguous), four k-steps per load;
//   * the forward / data-gradient GEMMs, which contract over features, read the row operand
//     with one dword per lane and k-step: 4 x 64-byte segments per wave instruction, at a
//     rate (256 B per 1024 MFMA cycles) that is nowhere near a limit.
// Packed weights use the same format with the contraction index in the role of m.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svae.h"

namespace svae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kSlots = 8;     // floats per (image, feature) entry of the first-layer table
constexpr int kBiasSlot = 5;  // slot holding b_c[k] + (W_z z_b)[k]; slots 0..4 = effective coord weights

struct Geo {
    int B, N, H, L, Zd, C, in_dim, act, flags;
    int Timg;   // 32-row tiles per image
    int Npad;   // 32*Timg
    int Mp;     // B*Npad padded rows
    int Hp;     // H rounded up to 32
    int ntile;  // Hp/32 feature tiles
    long noct;  // Mp/8 row octets
    long tiles; // Mp/32 row tiles
};

inline Geo make_geo(const svae_desc& d) {
    Geo g;
    g.B = d.B; g.N = d.N; g.H = d.H; g.L = d.L; g.Zd = d.Zd; g.C = d.C;
    g.in_dim = d.in_dim; g.act = d.act; g.flags = d.flags;
    g.Timg = (d.N + 31) / 32;
    g.Npad = g.Timg * 32;
    g.Mp = d.B * g.Npad;
    g.Hp = (d.H + 31) / 32 * 32;
    g.ntile = g.Hp / 32;
    g.noct = (long)g.Mp / 8;
    g.tiles = (long)g.Mp / 32;
    return g;
}

// Per-image pose handed to kernels by value (pointers may be null).
struct PoseArgs {
    const float* coords;  // (B,N,2) or null
    const float* grid;    // (N,2)
    const float* theta;   // (B) or null
    const float* dx;      // (B,2) or null
};

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// tanh(x) = 1 - 2/(exp(2x)+1); v_exp_f32 + v_rcp_f32 (1 ulp each).  Absolute error ~1e-7 over the
// whole range, saturates cleanly (exp -> inf gives 1, exp -> 0 gives -1).
__device__ __forceinline__ float fast_tanh(float x) {
    float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);  // 2*log2(e)
    return 1.0f - 2.0f * fast_rcp(t + 1.0f);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
    float t = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return fast_rcp(1.0f + t);
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float h) {
    if (ACT == SVAE_ACT_TANH) return fast_tanh(h);
    if (ACT == SVAE_ACT_LEAKYRELU) return h > 0.0f ? h : 0.01f * h;
    if (ACT == SVAE_ACT_RELU) return h > 0.0f ? h : 0.0f;
    return fast_sigmoid(h);
}
// derivative of the activation expressed through its OUTPUT a (what is kept in HBM)
template <int ACT>
__device__ __forceinline__ float act_grad(float a) {
    if (ACT == SVAE_ACT_TANH) return 1.0f - a * a;
    if (ACT == SVAE_ACT_LEAKYRELU) return a > 0.0f ? 1.0f : 0.01f;
    if (ACT == SVAE_ACT_RELU) return a > 0.0f ? 1.0f : 0.0f;
    return a * (1.0f - a);
}

// act' through the activation output a with the activation chosen at run time, WITHOUT branches (it is used per
// element inside MFMA loops): smooth activations are c0 + a*(c1 + c2*a) (tanh: 1 - a^2, sigmoid: a - a^2), the
// rectifiers a > 0 ? 1 : slope; the coefficients are wave-uniform.
struct ActCoef {
    float c0, c1, c2, slope, rect;  // rect = 1 for the rectifiers, 0 otherwise
};
__device__ __forceinline__ ActCoef act_coef(int act) {
    ActCoef k;
    k.c0 = act == SVAE_ACT_TANH ? 1.0f : 0.0f;
    k.c1 = act == SVAE_ACT_SIGMOID ? 1.0f : 0.0f;
    k.c2 = (act == SVAE_ACT_TANH || act == SVAE_ACT_SIGMOID) ? -1.0f : 0.0f;
    k.slope = act == SVAE_ACT_LEAKYRELU ? 0.01f : 0.0f;
    k.rect = (act == SVAE_ACT_LEAKYRELU || act == SVAE_ACT_RELU) ? 1.0f : 0.0f;
    return k;
}
__device__ __forceinline__ float act_grad_rt(const ActCoef& k, float a) {
    const float poly = k.c0 + a * (k.c1 + k.c2 * a);
    const float rl = a > 0.0f ? 1.0f : k.slope;
    return poly + k.rect * (rl - poly);
}

// coordinates of pixel i of image b (i < N), from explicit coords or grid + pose
__device__ __forceinline__ float2 pixel_coord(const PoseArgs& p, int b, int i, int N, float c, float s, float dx0,
                                              float dx1) {
    if (p.coords) {
        const float2 v = *reinterpret_cast<const float2*>(p.coords + ((long)b * N + i) * 2);
        return v;
    }
    const float2 g = *reinterpret_cast<const float2*>(p.grid + (long)i * 2);
    // x' = x @ [[c, s], [-s, c]]  (train_mnist.py:54-59), then + dx (train_mnist.py:70-74)
    return make_float2(c * g.x - s * g.y + dx0, s * g.x + c * g.y + dx1);
}

__device__ __forceinline__ void image_pose(const PoseArgs& p, int b, float& c, float& s, float& dx0, float& dx1) {
    c = 1.0f; s = 0.0f; dx0 = 0.0f; dx1 = 0.0f;
    if (!p.coords) {
        if (p.theta) {
            const float t = p.theta[b];
            c = cosf(t);
            s = sinf(t);
        }
        if (p.dx) {
            dx0 = p.dx[2 * b];
            dx1 = p.dx[2 * b + 1];
        }
    }
}

// coordinates of padded row i of image b; pad rows (i >= N) read as (0, 0)
__device__ __forceinline__ float2 row_coord(const PoseArgs& pose, const float4 pb, int b, int i, int N) {
    if (i >= N) return make_float2(0.0f, 0.0f);
    return pixel_coord(pose, b, i, N, pb.x, pb.y, pb.z, pb.w);
}

// [x0, x1, x0^2, x1^2, x0*x1] (models.py:99-102); only the first in_dim entries are used
__device__ __forceinline__ void coord_feats(float2 x, float f[5]) {
    f[0] = x.x; f[1] = x.y; f[2] = x.x * x.x; f[3] = x.y * x.y; f[4] = x.x * x.y;
}

__device__ __forceinline__ float wave_sum32(float v) {  // sum over the 32 lanes of one half-wave
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

// Sum over the 32 lanes of one half-wave with DPP only (no LDS crossbar traffic, unlike __shfl_xor's ds_bpermute, which
// competes with the GEMM's ds_read_b128 stream): quad butterflies, half-mirror, mirror, then row_bcast15 folds row 0
// into row 1 (row 2 into row 3).  The result is valid in lanes 16..31 of each half-wave ONLY.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    // old = 0 with bound_ctrl lets LLVM's DPP combiner fold the move into v_add_f32_dpp where every row is enabled
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float half_sum_dpp_hi(float v) {
    v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    v = dpp_add<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    return v;
}

// Sixteen values per lane, each to be summed over the 32 lanes of its half-wave: instead of 16 full butterflies (160
// instructions) the lanes SPLIT the work -- after exchanging with lane^1 a lane carries only 8 of the 16 sums, after
// lane^2 only 4; those 4 are then summed over the row's four quads (row_ror 4, 8) and over the two rows (lane^16).
// 72 instructions.  On return out[0..3] are the complete sums for value indices 4*q + (0..3) with
//   q = ((lane & 1) << 1) | ((lane >> 1) & 1)
// i.e. the lanes 0..3 of a half-wave together hold all 16 results (every other lane holds a copy of its class).
__device__ __forceinline__ void half_reduce16(const float (&v)[16], float (&out)[4]) {
    const bool b0 = (threadIdx.x & 1) != 0, b1 = (threadIdx.x & 2) != 0;
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float keep = b0 ? v[j + 8] : v[j], send = b0 ? v[j] : v[j + 8];
        w[j] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xf, 0xf, true));  // lane ^ 1
    }
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = b1 ? w[j + 4] : w[j], send = b1 ? w[j] : w[j + 4];
        x[j] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xf, 0xf, true));  // lane ^ 2
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[j]), 0x124, 0xf, 0xf, true));  // row_ror 4
        x[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[j]), 0x128, 0xf, 0xf, true));  // row_ror 8
        out[j] = x[j] + __shfl_xor(x[j], 16);                                                                   // the other row
    }
}

// sum over a 256-thread block; result valid in every thread.  red must hold 4 floats.
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum32(v);
    v += __shfl_xor(v, 32);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// sum over a block of up to 1024 threads (a whole number of waves); result valid in every thread.  red: 16 floats.
__device__ __forceinline__ float block_sum_waves(float v, float* red) {
    v = wave_sum32(v);
    v += __shfl_xor(v, 32);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
    for (unsigned w = 0; w < (blockDim.x >> 6); ++w) s += red[w];
    return s;
}

}  // namespace svae
