// Small-batch Linear layers of the inference network on the fp32 MFMA (reference: InferenceNetwork,
// spatial_vae/models.py:24-54 -- nn.Linear + activation per layer -- and their autograd backward).
//
// The encoder works on B rows (one per image: 64..512), not on B*N pixel rows, so its GEMMs are tiny: at BASELINE cfg 2 the
// three layers are 256 x 784 x 500, 256 x 500 x 500, 256 x 500 x 10.  What bounds them is not bytes or flops but the SERIAL
// chain of one output tile's MFMAs: with the 32x32x2 instruction a K = 784 tile is 392 dependent 64-cycle instructions =
// 12 us, which is what the vendor GEMM takes (8 launches of ~11 us per step, plus separate tanh, tanh-backward and column-sum
// kernels: 15 launches, 0.12 ms).  Here a 16 x 16 output tile belongs to a WORKGROUP: its four waves split the contraction
// in four contiguous quarters (v_mfma_f32_16x16x4_f32: 4 k per 32-cycle instruction) and are summed through LDS in a fixed
// order.  Operands come straight from L2 with ordinary loads, all of a wave's loads for up to 16 k-steps in flight at once
// (a first version with one wave per tile and the whole contraction in it spent 25 us per launch waiting: 12 round trips
// of ~2 us); the forward reads both operands as 16-byte vectors along k.  Bias + activation sit in the forward epilogue,
// and ONE backward launch per layer produces dW, db and dx with act' applied to the upstream gradient as it is loaded.
// 6 launches per step instead of 15.
//
// 16x16x4 operand layout: A lane l -> A[i = l & 15][k = l >> 4], B lane l -> B[k = l >> 4][j = l & 15],
// D lane l -> D[i = 4 (l >> 4) + r][j = l & 15], r = 0..3.  Everything is exact fp32 fma arithmetic in a fixed order.
#pragma once
#include "common.h"

namespace svae {

constexpr int SVAE_ACT_NONE = -1;

__device__ __forceinline__ float enc_act(int act, float v) {
    switch (act) {
        case SVAE_ACT_TANH: return fast_tanh(v);
        case SVAE_ACT_LEAKYRELU: return v > 0.0f ? v : 0.01f * v;
        case SVAE_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case SVAE_ACT_SIGMOID: return fast_sigmoid(v);
        default: return v;
    }
}
// derivative through the layer's OUTPUT a (what the forward kept)
__device__ __forceinline__ float enc_act_grad(int act, float a) {
    switch (act) {
        case SVAE_ACT_TANH: return 1.0f - a * a;
        case SVAE_ACT_LEAKYRELU: return a > 0.0f ? 1.0f : 0.01f;
        case SVAE_ACT_RELU: return a > 0.0f ? 1.0f : 0.0f;
        case SVAE_ACT_SIGMOID: return a * (1.0f - a);
        default: return 1.0f;
    }
}

struct LinFwdArgs {
    const float* x;    // (M, K) row-major
    const float* w;    // (N, K) row-major: nn.Linear.weight
    const float* b;    // (N) or null
    float* out;        // (M, N)
    int M, K, N, act;
};

constexpr int kEncUnroll = 16;  // k-steps (of 4) per loop trip and operand: 32-48 loads in flight per wave

// sum of the four waves' accumulators (and of one extra scalar per lane) through LDS, fixed order (0 + 1) + (2 + 3); the result
// is valid in wave 0
__device__ __forceinline__ void enc_reduce4(v4f& acc, float& extra, float (*red)[5][64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
    red[wave][4][lane] = extra;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
        extra = (red[0][4][lane] + red[1][4][lane]) + (red[2][4][lane] + red[3][4][lane]);
    }
}

// out[m][n] = act( sum_k x[m][k] w[n][k] + b[n] ).  One 16 x 16 tile per workgroup; wave s contracts the s-th quarter of k.
// VEC: K is a multiple of 4 and both operands are read as float4 along k -- lane (i, g) of k-group u holds
// k = 16 u + 4 g + (0..3) of its row, and MFMA t of the group consumes component t of BOTH operands (any bijection between
// k and (instruction, lane group) is a valid contraction order as long as A and B agree).
template <bool VEC>
__global__ void __launch_bounds__(256) enc_linear_fwd_kernel(LinFwdArgs a) {
    __shared__ float red[4][5][64];
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int tiles_n = (a.N + 15) >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int row = tm * 16 + i, col = tn * 16 + i;
    const bool va = row < a.M, vb = col < a.N;
    const float* ap = a.x + (long)(va ? row : a.M - 1) * a.K;
    const float* bp = a.w + (long)(vb ? col : a.N - 1) * a.K;
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (VEC) {
        const int groups = (a.K + 15) >> 4;            // k-groups of 16 (the last one may be partial: K % 16 in {0, 4, 8, 12})
        const int per = (groups + 3) >> 2;
        const int u0 = wave * per, u1 = (u0 + per < groups) ? u0 + per : groups;
        constexpr int U = kEncUnroll / 2;              // 8 groups = 32 instructions per trip (16 groups per trip was measured: slower)
        for (int ub = u0; ub < u1; ub += U) {
            float4 av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = (ub + u) * 16 + 4 * g;
                const bool in = ub + u < u1 && k < a.K;
                const int kc = in ? k : 0;
                av[u] = *reinterpret_cast<const float4*>(ap + kc);
                bv[u] = *reinterpret_cast<const float4*>(bp + kc);
                if (!in || !va) av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!in || !vb) bv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, acc, 0, 0, 0);
            }
        }
    } else {
        const int steps = (a.K + 3) >> 2;              // k-steps of 4: lane group g takes k = 4 s + g
        const int per = (steps + 3) >> 2;
        const int s0 = wave * per, s1 = (s0 + per < steps) ? s0 + per : steps;
        for (int sb = s0; sb < s1; sb += kEncUnroll) {
            float av[kEncUnroll], bv[kEncUnroll];
#pragma unroll
            for (int u = 0; u < kEncUnroll; ++u) {
                const int k = (sb + u) * 4 + g;
                const bool in = sb + u < s1 && k < a.K;
                const int kc = in ? k : 0;
                av[u] = (in && va) ? ap[kc] : 0.0f;
                bv[u] = (in && vb) ? bp[kc] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kEncUnroll; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    float none = 0.0f;
    enc_reduce4(acc, none, red);
    if (wave != 0) return;
    const int n = tn * 16 + i;  // D: column j = lane & 15, rows 4 g + r
    if (n < a.N) {
        const float bias = a.b ? a.b[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = tm * 16 + 4 * g + r;
            if (m < a.M) a.out[(long)m * a.N + n] = enc_act(a.act, acc[r] + bias);
        }
    }
}

struct LinBwdArgs {
    const float* x;     // (M, K) the layer's input
    const float* w;     // (N, K)
    const float* out;   // (M, N) the layer's output (after the activation); unused when act == NONE
    const float* dout;  // (M, N) d(loss)/d(out)
    float* dw;          // (N, K) or null
    float* db;          // (N) or null
    float* dx;          // (M, K) or null
    int M, K, N, act;
    int tiles_w;        // workgroups of role W: ceil(N/16) * ceil(K/16) (0 when neither dw nor db is wanted)
};

// One launch, two roles, chosen per workgroup; the four waves split the contraction as in the forward kernel.  With
// dpre[m][n] = dout[m][n] * act'(out[m][n]) formed as it is loaded:
//   role W (tile of dW: 16 outputs n x 16 inputs k, contraction over the M rows):  dW[n][k] = sum_m dpre[m][n] x[m][k];
//           the tiles of input-column block 0 also carry db[n] = sum_m dpre[m][n] (their own A operands, summed in a fixed order)
//   role X (tile of dx: 16 rows m x 16 inputs k, contraction over the N outputs):  dx[m][k] = sum_n dpre[m][n] w[n][k]
// VECX: N is a multiple of 4 (and dout / out are 16-byte aligned): role X reads the upstream gradient and the layer output as
// float4 along n, lane (row, g) of n-group u holding n = 16 u + 4 g + (0..3); instruction t of the group takes component t and
// the weight row n = 16 u + 4 g + t (see the forward kernel for why any such order is a valid contraction).
template <bool VECX>
__global__ void __launch_bounds__(256) enc_linear_bwd_kernel(LinBwdArgs a) {
    __shared__ float red[4][5][64];
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int tiles_k = (a.K + 15) >> 4;
    const bool has_act = a.act != SVAE_ACT_NONE;
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    float bsum = 0.0f;
    if ((int)blockIdx.x < a.tiles_w) {
        const int tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
        const int n = tn * 16 + i, k = tk * 16 + i;
        const bool vn = n < a.N, vk = k < a.K;
        const int nc = vn ? n : a.N - 1, kc = vk ? k : a.K - 1;
        const int steps = (a.M + 3) >> 2;              // contraction steps of 4 rows: lane group g takes row 4 s + g
        const int per = (steps + 3) >> 2;
        const int s0 = wave * per, s1 = (s0 + per < steps) ? s0 + per : steps;
        for (int sb = s0; sb < s1; sb += kEncUnroll) {
            float dv[kEncUnroll], ov[kEncUnroll], xv[kEncUnroll];
#pragma unroll
            for (int u = 0; u < kEncUnroll; ++u) {
                const long m = (long)(sb + u) * 4 + g;
                const bool in = sb + u < s1 && m < a.M;
                const long mc = in ? m : 0;
                dv[u] = (in && vn) ? a.dout[mc * a.N + nc] : 0.0f;
                ov[u] = has_act ? a.out[mc * a.N + nc] : 0.0f;
                xv[u] = (in && vk) ? a.x[mc * a.K + kc] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kEncUnroll; ++u) {
                const float dp = dv[u] * enc_act_grad(a.act, ov[u]);   // dv is 0 outside the tile / the rows
                bsum += dp;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dp, xv[u], acc, 0, 0, 0);
            }
        }
        enc_reduce4(acc, bsum, red);
        if (wave != 0) return;
        if (a.dw && vk) {  // D: column j = lane & 15 -> k, rows 4 g + r -> n
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = tn * 16 + 4 * g + r;
                if (nn < a.N) a.dw[(long)nn * a.K + k] = acc[r];
            }
        }
        if (tk == 0 && a.db) {  // the four row groups g of output n: fixed order (0 + 2) + (1 + 3)
            bsum += __shfl_xor(bsum, 32);
            bsum += __shfl_xor(bsum, 16);
            if (g == 0 && vn) a.db[n] = bsum;
        }
        return;
    }
    if (!a.dx) return;
    const int t = blockIdx.x - a.tiles_w;
    const int tm = t / tiles_k, tk = t - tm * tiles_k;
    const int row = tm * 16 + i, k = tk * 16 + i;
    const bool vr = row < a.M, vk = k < a.K;
    const long rc = vr ? row : a.M - 1;
    const int kc = vk ? k : a.K - 1;
    if (VECX) {
        const int groups = (a.N + 15) >> 4;
        const int per = (groups + 3) >> 2;
        const int u0 = wave * per, u1 = (u0 + per < groups) ? u0 + per : groups;
        constexpr int U = kEncUnroll / 4;               // 4 groups = 16 instructions per trip: 8 float4 + 16 dword loads in flight
        for (int ub = u0; ub < u1; ub += U) {
            float4 dv[U], ov[U];
            float wv[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = (ub + u) * 16 + 4 * g;
                const bool in = ub + u < u1 && n < a.N;
                const int ncl = in ? n : 0;
                dv[u] = *reinterpret_cast<const float4*>(a.dout + rc * a.N + ncl);
                ov[u] = has_act ? *reinterpret_cast<const float4*>(a.out + rc * a.N + ncl) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (!in || !vr) dv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < 4; ++t) wv[u][t] = (in && vk) ? a.w[(long)(ncl + t) * a.K + kc] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u].x * enc_act_grad(a.act, ov[u].x), wv[u][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u].y * enc_act_grad(a.act, ov[u].y), wv[u][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u].z * enc_act_grad(a.act, ov[u].z), wv[u][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u].w * enc_act_grad(a.act, ov[u].w), wv[u][3], acc, 0, 0, 0);
            }
        }
    }
    const int steps = VECX ? 0 : (a.N + 3) >> 2;       // contraction steps of 4 outputs: lane group g takes n = 4 s + g
    const int per = (steps + 3) >> 2;
    const int s0 = wave * per, s1 = (s0 + per < steps) ? s0 + per : steps;
    for (int sb = s0; sb < s1; sb += kEncUnroll) {
        float dv[kEncUnroll], ov[kEncUnroll], wv[kEncUnroll];
#pragma unroll
        for (int u = 0; u < kEncUnroll; ++u) {
            const int n = (sb + u) * 4 + g;
            const bool in = sb + u < s1 && n < a.N;
            const int ncl = in ? n : 0;
            dv[u] = (in && vr) ? a.dout[rc * a.N + ncl] : 0.0f;
            ov[u] = has_act ? a.out[rc * a.N + ncl] : 0.0f;
            wv[u] = (in && vk) ? a.w[(long)ncl * a.K + kc] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kEncUnroll; ++u)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u] * enc_act_grad(a.act, ov[u]), wv[u], acc, 0, 0, 0);
    }
    enc_reduce4(acc, bsum, red);
    if (wave != 0) return;
    if (vk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = tm * 16 + 4 * g + r;
            if (m < a.M) a.dx[(long)m * a.K + k] = acc[r];
        }
    }
}

}  // namespace svae
