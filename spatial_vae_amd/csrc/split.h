// fp16x3 split-operand GEMM path (opt-in: SVAE_GEMM=fp16x3) -- fp32-accurate products on the f16 matrix pipe.
//
// Every fp32 operand tensor X is carried as two half tensors of s*X,  hi = half(s x), lo = half(s x - hi), with a
// power-of-two scale s per tensor that puts its largest entries near 2^10..2^14 (so that lo stays a NORMAL half for every
// entry that matters); a product is formed as  hi_a*hi_w + hi_a*lo_w + lo_a*hi_w  in ONE fp32 MFMA accumulator and the
// epilogue multiplies by 1/(s_a s_w) (exact).  The dropped lo*lo term is 2^-22 relative: the result is as accurate as an
// fp32 GEMM (tools/split_numerics.py: 2.8e-7 rms against fp64, the fp32 GEMM's own figure), while
// v_mfma_f32_32x32x16_f16 runs at 16x the rate of v_mfma_f32_32x32x2_f32: three of them per K=16 step cost 96 cycles
// where the fp32 form needs 512.  Scales: activations after tanh / sigmoid (|a| <= 1) 2^10; weights 2^floor(log2(8192 /
// max|W|)) per matrix (split_wamax_kernel + weight_scale); the gradient entering the last hidden layer from a bound
// (split_scale_kernel), the gradients further down from the tracked maximum of the plane (split_scale_amax_kernel).
//
// Layouts (all 16-byte = 8-half fragments, addressed in uint4 units):
//   rows  As[T][kc][part][lane]   T = 32-row tile, kc = 16-feature step, part 0 = hi / 1 = lo, lane = MFMA lane
//                                 (r = lane & 31 the row, h = lane >> 5): features 16 kc + 8 h + (0..7) of row 32 T + r.
//                                 One (T, kc, part) block is 1 KiB: a wave loads it with one global_load_dwordx4.
//   weights Ws[kc][nt][part][lane] nt = 32-column tile: contraction indices 16 kc + 8 h + (0..7) of column 32 nt + r.
//                                 One block is 1 KiB and goes to LDS verbatim: ds_read_b128 is lane-linear, conflict-free.
//   columns Cs[ms][ft][part][lane] ms = 16-row step, ft = 32-feature tile, lane = h*32 + f: rows 16 ms + 8 h + (0..7) of
//                                 feature 32 ft + f (both operands of the weight gradient, whose contraction runs over rows).
#pragma once
#include "common.h"
#include "dense.h"        // glds16
#include "elementwise.h"  // RowGeo

namespace svae {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kActScale = 1024.0f;  // activations in [-1, 1]
constexpr float kActInv = 1.0f / 1024.0f;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
union Frag {  // 8 halfs <-> 16 bytes
    f16x8 h;
    uint4 u;
    u32x4 v;  // the form inline asm takes ("v" constraint on a 128-bit register tuple)
};

// 16 bytes per lane into the registers the variable already lives in; invisible to hipcc's waitcnt pass like every
// vector-memory operation of the GEMM loop (see dense.h): the loop's own s_waitcnt are hand-counted.
__device__ __forceinline__ void load_frag(const uint4* p, u32x4& v) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(v) : "v"(p) : "memory");
}
// the same with a wave-uniform base pointer in SGPRs, a 32-bit per-lane byte offset and an immediate: no 64-bit VALU
// address arithmetic inside the GEMM loop
template <int OFF>
__device__ __forceinline__ void load_frag_s(const void* sbase, unsigned voff, u32x4& v) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(v) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_frag(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}

__device__ __forceinline__ void split8(const float (&x)[8], float sc, uint4& hi, uint4& lo) {
    Frag fh, fl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = x[j] * sc;  // power of two: exact
        const _Float16 hj = (_Float16)xs;
        fh.h[j] = hj;
        fl.h[j] = (_Float16)(xs - (float)hj);
    }
    hi = fh.u;
    lo = fl.u;
}

// max |W| of one weight matrix as float bits (non-negative floats order like their bit patterns): 64 blocks, one atomic each
__global__ void split_wamax_kernel(const float* __restrict__ W, long count, unsigned* __restrict__ amax) {
    __shared__ float red[4];
    float m = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(W[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
// s = 2^floor(log2(8192 / max|W|)) from those bits
__device__ __forceinline__ float weight_scale(unsigned amax_bits) {
    const float wmax = __uint_as_float(amax_bits);
    float sc = 1.0f;
    if (wmax > 0.0f && wmax < 3.0e38f) {
        int e;
        frexpf(8192.0f / wmax, &e);
        e = e - 1 < -60 ? -60 : (e - 1 > 60 ? 60 : e - 1);
        sc = ldexpf(1.0f, e);
    }
    return sc;
}

// W (H x H, row-major [out n][in k]) -> Ws for the forward contraction over k: B[k][n] = W[n][k].
// transpose != 0 packs the data-gradient form instead: contraction over n, B[n][k] = W[n][k].
// (every thread derives the matrix scale from the amax bits; thread 0 publishes {s, 1/s} for the GEMM epilogue)
__global__ void split_weights_kernel(const float* __restrict__ W, uint4* __restrict__ ws, int H, int Hp, int transpose,
                                     const unsigned* __restrict__ amax, float* __restrict__ wscale) {
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
    const float sc = weight_scale(amax[0]);
    if (idx == 0) {
        wscale[0] = sc;
        wscale[1] = 1.0f / sc;
    }
    uint4 hi, lo;
    split8(x, sc, hi, lo);
    const long blk = ((long)kc * ntile + nt) * 2;
    ws[blk * 64 + lane] = hi;
    ws[(blk + 1) * 64 + lane] = lo;
}

// fp32 octet-major plane (element (m, k) at ((m>>3)*Hp + k)*8 + (m&7)) -> As, multiplied by its power-of-two scale
// (*scale for a gradient, else the activation scale)
__global__ void split_rows_kernel(const float* __restrict__ in, uint4* __restrict__ as, long tiles, int Hp,
                                  const float* __restrict__ scale) {
    const int KC = Hp / 16;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (T, kc, lane)
    if (idx >= tiles * KC * 64) return;
    const int lane = idx & 63;
    const int kc = (int)((idx >> 6) % KC);
    const long T = (idx >> 6) / KC;
    const long m = T * 32 + (lane & 31);
    const int k0 = kc * 16 + 8 * (lane >> 5);
    const float* p = in + ((m >> 3) * Hp + k0) * 8 + (m & 7);
    const float sc = scale ? scale[0] : kActScale;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p[j * 8];
    uint4 hi, lo;
    split8(x, sc, hi, lo);
    const long blk = (T * KC + kc) * 2;
    as[blk * 64 + lane] = hi;
    as[(blk + 1) * 64 + lane] = lo;
}

// Coordinate layer in fp16x3 mode: a0 goes out TWICE from one pass -- fp32 octet-major (the backward kernels read it) and
// split row fragments As (the forward GEMM's operand) -- instead of a conversion pass over the fp32 plane.  The two
// layouts want different thread mappings (4 rows x 1 feature per 16-byte store vs 1 row x 8 features), so a workgroup
// computes a 32-row x 64-feature tile with the fp32 mapping (stores coalesced as in layer0_fwd_kernel), parks it in LDS
// and re-reads it with the fragment mapping: both streams leave as contiguous 1 KiB per wave instruction.
template <int ACT>
__global__ void __launch_bounds__(256) layer0_fwd_split_kernel(PoseArgs pose, const float4* __restrict__ posebuf,
                                                                const float* __restrict__ tab, float* __restrict__ a0,
                                                                uint4* __restrict__ as, uint4* __restrict__ cs, RowGeo g) {
    __shared__ float tile[32][65];  // [row][feature], padded: both access patterns are bank-conflict-free
    const long T = blockIdx.x;      // 32-row tile
    const int f0 = blockIdx.y * 64; // first feature of this block
    const int Timg = g.Npad >> 5;
    const int b = (int)(T / Timg);
    const int irow0 = (int)(T - (long)b * Timg) * 32;  // first pixel row of the tile inside its image
    const float4 pb = posebuf[b];
    const float* cbase = pose.coords ? pose.coords + (long)b * g.N * 2 : pose.grid;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int task = threadIdx.x + 256 * it;     // (octet oo, feature k, half h): 4 rows of one feature
        const int oo = task >> 7, k = (task & 127) >> 1, h = task & 1;
        const float4* tp = reinterpret_cast<const float4*>(tab + ((long)b * g.Hp + f0 + k) * kSlots);
        const float4 t0 = tp[0], t1 = tp[1];
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = irow0 + oo * 8 + 4 * h + e;
            const float2 raw = *reinterpret_cast<const float2*>(cbase + (long)(i < g.N ? i : g.N - 1) * 2);
            const bool in = i < g.N;
            const float x0 = in ? pb.x * raw.x - pb.y * raw.y + pb.z : 0.0f;
            const float x1 = in ? pb.y * raw.x + pb.x * raw.y + pb.w : 0.0f;
            float v = t1.y + x0 * t0.x + x1 * t0.y;
            if (g.in_dim == 5) v += (x0 * x0) * t0.z + (x1 * x1) * t0.w + (x0 * x1) * t1.x;
            out[e] = act_fwd<ACT>(v);
            tile[oo * 8 + 4 * h + e][k] = out[e];
        }
        const long o = T * 4 + oo;
        if (a0) *reinterpret_cast<float4*>(a0 + ((o * g.Hp + f0 + k) * 8 + 4 * h)) = make_float4(out[0], out[1], out[2], out[3]);
    }
    __syncthreads();
    {   // fragments: thread = (K-step kq of this block's four, lane): row lane & 31, features 16 kq + 8 (lane >> 5) + j
        const int kq = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int r = lane & 31, c0 = kq * 16 + 8 * (lane >> 5);
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = tile[r][c0 + j];
        uint4 hi, lo;
        split8(x, kActScale, hi, lo);
        const int KC = g.Hp / 16;
        uint4* dst = as + ((T * KC + (f0 >> 4) + kq) * 2) * 64 + lane;
        dst[0] = hi;
        dst[64] = lo;
    }
    if (cs) {  // column fragments for the weight gradient (kept forward -> backward): thread = (16-row step mq, tile fq, lane)
        const int mq = threadIdx.x >> 7, fq = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
        const int r0 = 16 * mq + 8 * (lane >> 5), c = 32 * fq + (lane & 31);
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = tile[r0 + j][c];
        uint4 hi, lo;
        split8(x, kActScale, hi, lo);
        const int FT = g.Hp / 32;
        uint4* dst = cs + (((T * 2 + mq) * FT + (f0 >> 5) + fq) * 2) * 64 + lane;
        dst[0] = hi;
        dst[64] = lo;
    }
}

// Power-of-two scale for the gradient entering the last hidden layer: |dh[m][n]| = |sum_c do[m][c] W_o[c][n]| act' is at
// most  amax|do| * C * amax|W_o|  (act' <= 1 for tanh and sigmoid); the scale maps that bound to 2^14, inside the half
// range with room for the GEMM's partial sums.  scale[0] = s, scale[1] = 1/s (both exact).
__global__ void split_scale_kernel(const unsigned* __restrict__ amax_do_bits, const float* __restrict__ out_w, int C, int H,
                                   float* __restrict__ scale) {
    __shared__ float red[4];
    float m = 0.0f;
    for (int i = threadIdx.x; i < C * H; i += 256) m = fmaxf(m, fabsf(out_w[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float wmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float bound = __uint_as_float(amax_do_bits[0]) * (float)C * wmax;
        float sc = 1.0f;
        if (bound > 0.0f && bound < 3.0e38f) {
            int e;
            frexpf(16384.0f / bound, &e);  // 16384 / bound = f * 2^e, f in [0.5, 1)
            e = e - 1 < -60 ? -60 : (e - 1 > 60 ? 60 : e - 1);
            sc = ldexpf(1.0f, e);
        }
        scale[0] = sc;
        scale[1] = 1.0f / sc;
    }
}

// {s, 1/s} for a gradient plane whose largest entry is known: s = 2^floor(log2(16384 / max))
__global__ void split_scale_amax_kernel(const unsigned* __restrict__ amax_bits, float* __restrict__ scale) {
    const float m = __uint_as_float(amax_bits[0]);
    float sc = 1.0f;
    if (m > 0.0f && m < 3.0e38f) {
        int e;
        frexpf(16384.0f / m, &e);
        e = e - 1 < -60 ? -60 : (e - 1 > 60 ? 60 : e - 1);
        sc = ldexpf(1.0f, e);
    }
    scale[0] = sc;
    scale[1] = 1.0f / sc;
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
    // data gradient (MODE 1 / 2): act' of aux multiplies the result; scale[1] undoes the operand scale
    const float* aux;    // a_{l-1}, fp32 octet-major
    const uint4* auxc;   // ... or its column fragments (MODE 2 when the coordinate layer wrote no fp32 plane)
    const float* scale;  // device {s, 1/s} of the row operand when it is a gradient; null: an activation (kActScale)
    const float* wscale; // device {s, 1/s} of the weights
    unsigned* amax_out;  // MODE 1: max |result| as float bits (the next layer's gradient scale derives from it), or null
    // MODE 2 (into the coordinate layer): reduced on the spot, see dense_kernel FIRST
    PoseArgs pose;
    const float4* posebuf;
    const float* tab;
    float* sgtile;
    float* dfpart;
    int N, Timg;
};

// timing-only ablation switches (never set in the product build): 1 = every tile reads the rows of tile 0 (L2-hot row
// operand), 2 = no result stores, 4 = no weight DMA / barrier in the loop, 8 = no weight-fragment LDS reads in the loop, 128 = no data-gradient epilogue, 256 = weight gradient without DMA in the loop, 512 = without barriers
#ifndef SVAE_SPLIT_ABLATE
#define SVAE_SPLIT_ABLATE 0
#endif
#ifndef SVAE_SPLIT_G
#define SVAE_SPLIT_G 2
#endif
constexpr int kSplitG = SVAE_SPLIT_G;  // K=16 steps per LDS chunk: the contraction length must be a multiple of 16 G

template <int NT>
struct SplitCfg {
    static constexpr int BLOCKS = kSplitG * NT * 2;        // 1 KiB blocks per chunk
    static constexpr int LDS_BYTES = 2 * BLOCKS * 1024;    // double-buffered
};

// MODE 0: out(32 rows x NT*32 cols per wave) = act( As * Ws + bias [+ resid] )            (forward; CF: + partial logits)
// MODE 1: out = ((As * Ws) / s [+ resid]) * act'(aux)                                      (data gradient, stored)
// MODE 2: the same value reduced on the spot into the coordinate layer's partial sums      (dense_kernel's FIRST)
// kSplitWaves waves (row tiles) share the weight chunks: at f16-MFMA speed the vector-memory path (64 B/clk/CU) is the
// scarce resource -- every wave streams 2 KiB of row fragments per K-step and its share of the 2 KiB x NT weight blocks;
// eight waves per workgroup instead of four halve the weight share (ablation: the LDS-DMA alone cost 0.09 of 0.32 ms).
#ifndef SVAE_SPLIT_BD2
#define SVAE_SPLIT_BD2 2  // weight-fragment read-ahead (column tiles) of the fused first-layer variant, which has registers to spare
#endif
#ifndef SVAE_SPLIT_WAVES
#define SVAE_SPLIT_WAVES 8
#endif
constexpr int kSplitWaves = SVAE_SPLIT_WAVES;
// waves per workgroup of a variant: the fused first-layer data gradient keeps ~230 registers, i.e. two waves per SIMD; as ONE
// 8-wave workgroup per CU all of them are in the same phase and nothing overlaps its long epilogue (27 % of the kernel)
// -- as two independent 4-wave workgroups the epilogue of one runs under the MFMA loop of the other.
#ifndef SVAE_SPLIT_WAVES_M2
#define SVAE_SPLIT_WAVES_M2 4
#endif
template <int MODE>
struct SplitWaves {
    static constexpr int value = MODE == 2 ? SVAE_SPLIT_WAVES_M2 : kSplitWaves;
};
template <int NT, int MODE, bool RESID, int CF>
__global__ __launch_bounds__(SplitWaves<MODE>::value * 64, (SplitWaves<MODE>::value == 8 ? 1 : (MODE == 2 ? 2 : 3))) void dense_split_kernel(SplitArgs a) {
    constexpr int WV = SplitWaves<MODE>::value;
    static_assert(CF == 0 || MODE == 0, "CF is a forward epilogue");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    using Cfg = SplitCfg<NT>;
    constexpr int G = kSplitG, BLOCKS = Cfg::BLOCKS;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform values must be SCALAR computations from here on: an SGPR written by a VALU instruction
    // (v_readfirstlane) needs 5 wait states before a vector-memory instruction may read it, and hipcc's hazard
    // recognizer does not see SGPR operands of inline asm -- a readfirstlane right in front of an asm load sent the
    // load to a stale base (memory fault at address 0).  One readfirstlane here, fenced by s_nop, and scalar math after.
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    asm volatile("s_nop 4" : "+s"(wave));
    const int nl = lane & 31, h = lane >> 5;
    const int Hp = a.Hp, KC = Hp / 16, ntile = Hp / 32;
    // XCD-aware 1-D grid: workgroups are dealt round-robin over the 8 XCDs, so ids congruent mod 8 share an L2.  The
    // nblk column blocks of one 128-row group get consecutive ids ON ONE XCD: the group's row operand is fetched from
    // HBM once and re-read from that L2 (a 2-D grid re-reads the whole plane per column block: 4x the HBM traffic,
    // which at f16-MFMA speed is what bounds the kernel).
    const int nblk = ntile / NT;
    const long local = blockIdx.x >> 3;
    const int nb = (int)(local % nblk);  // column block of NT tiles
    const long group = (local / nblk) * 8 + (blockIdx.x & 7);
    const long tile = group * WV + wave;
    const bool live = tile < a.tiles;
    const long tl = live ? tile : a.tiles - 1;
    const int nchunk = KC / G;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // Weight chunks reach LDS by LDS-DMA issued from asm (glds16), ONE chunk ahead: at the top of chunk c every wave has
    // waited for its own pieces of chunk c, the barrier publishes them and proves that nobody reads the other buffer
    // any more, then the pieces of chunk c+1 go out into that buffer.  Row-operand fragments (hi, lo per K-step) are
    // re-issued for the next chunk right behind the MFMAs that consumed them.  The loop body is ONE straight-line
    // block (no phases, no early exits): in-flight asm-load destinations cross only the loop's own back-edge in the
    // registers they were loaded into (tools/check_asm_loads.py).  In-order vmcnt bookkeeping per wave:
    //   issue order per chunk:  DMA x P, then per step g: A(g).hi, A(g).lo
    //   step g needs A(g) of this chunk (issued in the previous one): behind it are 2(G-1-g) A loads of that chunk,
    //   this chunk's P pieces and 2g A loads -> vmcnt(2(G-1) + P), one constant for every step;
    //   the barrier needs this wave's pieces of chunk c (issued at the top of chunk c-1): behind them 2G A loads.
    constexpr int P = BLOCKS / WV;  // 1 KiB pieces per wave and chunk
    static_assert(BLOCKS % WV == 0, "the chunk must divide over the waves");
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)smem4;
    // Addresses are prepared once: piece j of this wave is block idx = wave + WV j of a chunk; its source offset inside
    // the chunk and its LDS address are loop-invariant, the chunk base advances by a scalar add.  (Computed per piece
    // inside the loop this was ~130 instructions per 48 MFMAs: a third of the kernel.)
    unsigned poff[P], pm0[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int idx = wave + WV * j;
        const int part = idx & 1, t = (idx >> 1) % NT, g = (idx >> 1) / NT;
        poff[j] = (unsigned)((((g * ntile + nb * NT + t) * 2 + part) * 64 + lane) * 16);
        pm0[j] = lds_base + (unsigned)idx * 1024u;  // scalar: wave and the LDS base are SGPR values
    }
    const unsigned chunk_bytes = (unsigned)(G * ntile * 2 * 1024);  // weight bytes per chunk
    const char* wchunk = reinterpret_cast<const char*>(a.ws);       // wave-uniform: base of the chunk being staged
    auto stage_chunk = [&](const char* base, unsigned bufoff) {
#pragma unroll
        for (int j = 0; j < P; ++j) glds16_s(base, poff[j], pm0[j] + bufoff);
    };
    // row operand: wave-uniform tile base in SGPRs + per-lane byte offset; K-step g at +2048 g, lo at +1024
    const char* abase = reinterpret_cast<const char*>(a.as + (((SVAE_SPLIT_ABLATE & 1) ? 0 : tl) * KC) * 2 * 64);
    const unsigned aoff = (unsigned)lane * 16u;  // this lane inside a 1 KiB block; chunk c starts at + c * G * 2048
    auto load_rows = [&](unsigned off, int g, Frag& fh, Frag& fl) {
        if (g == 0) { load_frag_s<0>(abase, off, fh.v); load_frag_s<1024>(abase, off, fl.v); }
        if (g == 1) { load_frag_s<2048>(abase, off, fh.v); load_frag_s<3072>(abase, off, fl.v); }
        if (g == 2) { load_frag_s<0>(abase, off + 4096u, fh.v); load_frag_s<1024>(abase, off + 4096u, fl.v); }
        if (g == 3) { load_frag_s<2048>(abase, off + 4096u, fh.v); load_frag_s<3072>(abase, off + 4096u, fl.v); }
    };
    static_assert(G == 2 || G == 4, "load_rows addresses at most four K-steps per chunk");
    // Row fragments are re-issued AD chunks ahead (ring slots stay compile-time constants: the chunk loop is unrolled by
    // AD, its body one straight-line block).  AD = 2 was measured for the fused first-layer variant, which runs at two
    // waves per SIMD and waits on memory 42 % of its wave time: no change (0.356 ms either way), so one chunk it stays.
    constexpr int AD = 1;
    Frag rh[AD][G], rl[AD][G];
#pragma unroll
    for (int u = 0; u < AD; ++u)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            rh[u][g].v = (u32x4){0u, 0u, 0u, 0u};
            rl[u][g].v = (u32x4){0u, 0u, 0u, 0u};
        }
    stage_chunk(wchunk, 0u);
#pragma unroll
    for (int u = 0; u < AD; ++u) {
        const int cu = u < nchunk ? u : nchunk - 1;
#pragma unroll
        for (int g = 0; g < G; ++g) load_rows(aoff + (unsigned)(cu * G * 2048), g, rh[u][g], rl[u][g]);
    }
#pragma unroll
    for (int u = 0; u < AD; ++u)
#pragma unroll
        for (int g = 0; g < G; ++g) wait_frag<0>(rh[u][g].v, rl[u][g].v);

    for (int c0 = 0; c0 < nchunk; c0 += AD) {  // nchunk = ntile is even whenever this path is taken
#pragma unroll
        for (int u = 0; u < AD; ++u) {
            const int c = c0 + u;
            const bool more = c + 1 < nchunk;  // the last chunk re-stages itself into the idle buffer (never read)
            const uint4* buf = smem4 + (c & 1) * (BLOCKS * 64);
            if (SVAE_SPLIT_ABLATE & 16) wait_frag<2 * G + P>(rh[u][0].v, rl[u][0].v);
            else wait_frag<2 * G>(rh[u][0].v, rl[u][0].v);
            if (!(SVAE_SPLIT_ABLATE & (4 | 32)) || c == 0) __syncthreads();
            wchunk += more ? chunk_bytes : 0u;
            const int cn = c + AD < nchunk ? c + AD : nchunk - 1;  // the chunk whose row fragments are fetched now
            const unsigned anext = aoff + (unsigned)(cn * G * 2048);
            if (!(SVAE_SPLIT_ABLATE & (4 | 64))) stage_chunk(wchunk, (unsigned)(((c + 1) & 1) * BLOCKS * 1024));
            else {
#pragma unroll
                for (int j = 0; j < P; ++j) asm volatile("s_nop 0" ::: "memory");
            }
            // weight fragments are read kBD column tiles ahead of the MFMAs that use them (a ring of kBD + 1 register pairs)
            constexpr int kBD = (MODE == 2) ? SVAE_SPLIT_BD2 : 2, kBR = kBD + 1;
            Frag bq[kBR][2];
#pragma unroll
            for (int i = 0; i < kBD; ++i) {
                if (i < G * NT) {
                    bq[i][0].u = buf[(i * 2) * 64 + lane];
                    bq[i][1].u = buf[(i * 2 + 1) * 64 + lane];
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                // behind A(g) of this chunk: the rest of its issue chunk, AD-1 whole chunks, this chunk's pieces and loads
                wait_frag<2 * (G - 1) + (AD - 1) * (P + 2 * G) + P>(rh[u][g].v, rl[u][g].v);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int it = g * NT + t;
                    const int cur = (SVAE_SPLIT_ABLATE & 8) ? 0 : it % kBR, nxt = (it + kBD) % kBR;
                    const int ni = it + kBD;  // the (step, tile) read now, used kBD tiles later
                    if (ni < G * NT && !(SVAE_SPLIT_ABLATE & 8)) {
                        bq[nxt][0].u = buf[(ni * 2) * 64 + lane];
                        bq[nxt][1].u = buf[(ni * 2 + 1) * 64 + lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[u][g].h, bq[cur][0].h, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[u][g].h, bq[cur][1].h, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl[u][g].h, bq[cur][0].h, acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                load_rows(anext, g, rh[u][g], rl[u][g]);
            }
        }
    }
    {   // drain: the last fragments and pieces are in flight and never used
        unsigned sink = 0;
#pragma unroll
        for (int u = 0; u < AD; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                wait_frag<0>(rh[u][g].v, rl[u][g].v);
                sink += rh[u][g].v[0] + rl[u][g].v[0];
            }
        if (a.tiles < 0) a.out[0] = __uint_as_float(sink);  // never true: ties the registers
    }
    __syncthreads();

    if (!live) return;
    auto epi = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        if constexpr (MODE == 0) {
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
        const float inv = kActInv * a.wscale[1];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long off = off0 + q * qstride + (long)t * 32 * 8;
                float4 v = make_float4(acc[t][4 * q] * inv, acc[t][4 * q + 1] * inv, acc[t][4 * q + 2] * inv, acc[t][4 * q + 3] * inv);
                if (RESID) {
                    const float4 fr = *reinterpret_cast<const float4*>(a.resid + off);
                    v.x += fr.x; v.y += fr.y; v.z += fr.z; v.w += fr.w;
                }
                v.x = act_fwd<ACT>(v.x + bias[t]); v.y = act_fwd<ACT>(v.y + bias[t]);
                v.z = act_fwd<ACT>(v.z + bias[t]); v.w = act_fwd<ACT>(v.w + bias[t]);
                if (!(SVAE_SPLIT_ABLATE & 2) || a.tiles < 0) *reinterpret_cast<float4*>(a.out + off) = v;
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
            for (int c = 0; c < CFN; ++c) {
                float s4[4];
                half_reduce16(lp[c], s4);
                if (nl < 4) {
                    const int q = ((nl & 1) << 1) | (nl >> 1);
                    *reinterpret_cast<float4*>(a.lpart + ((long)nb * CFN + c) * a.Mp + tl * 32 + 8 * q + 4 * h) =
                        make_float4(s4[0], s4[1], s4[2], s4[3]);
                }
            }
        }
        } else if ((SVAE_SPLIT_ABLATE & 128) && a.tiles >= 0) {
            // ablation: no data-gradient epilogue (a.tiles >= 0 is always true; the else branch keeps the code alive)
        } else {
        constexpr int NB = NT * 32;
        const float inv = (a.scale ? a.scale[1] : kActInv) * a.wscale[1];
        const long off0 = (tl * 4 * (long)Hp + nb * NB + nl) * 8 + 4 * h;
        const long qstride = (long)Hp * 8;
        // MODE 2 without a residual defers the 1/(s_a s_w) factor to its (few) sums instead of paying it per element
        constexpr bool kLateInv = (MODE == 2) && !RESID;
        auto value = [&](int t, int q) {  // this lane's 4 rows (8q + 4h + 0..3) of column tile t, times act'(a_{l-1})
            const long off = off0 + q * qstride + (long)t * 32 * 8;
            const float iv = kLateInv ? 1.0f : inv;
            float4 v = make_float4(acc[t][4 * q] * iv, acc[t][4 * q + 1] * iv, acc[t][4 * q + 2] * iv, acc[t][4 * q + 3] * iv);
            if (RESID) {
                const float4 fr = *reinterpret_cast<const float4*>(a.resid + off);
                v.x += fr.x; v.y += fr.y; v.z += fr.z; v.w += fr.w;
            }
            float4 ax;
            if (MODE == 2 && a.auxc) {
                // rows 8q + 4h + (0..3) of column (tile t, nl): halfs 4h..4h+3 of the fragment of 16-row step 2 tl + (q >> 1),
                // octet q & 1 -- a = (hi + lo) / 2^10
                const uint4* fr = a.auxc + ((((tl * 2 + (q >> 1)) * (Hp / 32) + nb * NT + t) * 2) * 64 + (q & 1) * 32 + nl);
                union { uint2 u; _Float16 f[4]; } ph, pl;
                ph.u = reinterpret_cast<const uint2*>(fr)[h];
                pl.u = reinterpret_cast<const uint2*>(fr + 64)[h];
                ax = make_float4(((float)ph.f[0] + (float)pl.f[0]) * kActInv, ((float)ph.f[1] + (float)pl.f[1]) * kActInv,
                                 ((float)ph.f[2] + (float)pl.f[2]) * kActInv, ((float)ph.f[3] + (float)pl.f[3]) * kActInv);
            } else {
                ax = *reinterpret_cast<const float4*>(a.aux + off);
            }
            v.x *= act_grad<ACT>(ax.x); v.y *= act_grad<ACT>(ax.y); v.z *= act_grad<ACT>(ax.z); v.w *= act_grad<ACT>(ax.w);
            return v;
        };
        if constexpr (MODE == 1) {
            float vmax = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = value(t, q);
                    *reinterpret_cast<float4*>(a.out + off0 + q * qstride + (long)t * 32 * 8) = v;
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            if (a.amax_out) {  // one atomic per wave; non-negative floats order like their bit patterns
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
                if (lane == 0 && vmax > 0.0f) atomicMax(a.amax_out, __float_as_uint(vmax));
            }
        } else {
            // coordinates of this lane's 16 rows (row 8q + 4h + r of the tile), wave-uniform image
            const int b = (int)(tl / a.Timg);
            const int i0 = (int)(tl % a.Timg) * 32 + 4 * h;
            const float4 pb = a.posebuf[b];
            const float* cbase = a.pose.coords ? a.pose.coords + (long)b * a.N * 2 : a.pose.grid;
            // (x0, x1) and the two per-row partial sums are kept as float pairs: v * pair is one packed FMA (v_pk_fma_f32)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 x01[16], pd01[16];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + 8 * q + r;
                    const float2 g2 = *reinterpret_cast<const float2*>(cbase + (long)(i < a.N ? i : a.N - 1) * 2);
                    const bool in = i < a.N;
                    x01[4 * q + r] = (f32x2){in ? pb.x * g2.x - pb.y * g2.y + pb.z : 0.0f, in ? pb.y * g2.x + pb.x * g2.y + pb.w : 0.0f};
                    pd01[4 * q + r] = (f32x2){0.0f, 0.0f};
                }
            const float late = kLateInv ? inv : 1.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int k = nb * NB + t * 32 + nl;
                const float2 wt = *reinterpret_cast<const float2*>(a.tab + ((long)b * Hp + k) * kSlots);
                const f32x2 w = (f32x2){wt.x, wt.y};
                float sv = 0.0f;
                f32x2 g01 = (f32x2){0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = value(t, q);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const f32x2 vb = (f32x2){vv[r], vv[r]};
                        sv += vv[r];
                        g01 = __builtin_elementwise_fma(vb, x01[4 * q + r], g01);
                        pd01[4 * q + r] = __builtin_elementwise_fma(vb, w, pd01[4 * q + r]);
                    }
                }
                *reinterpret_cast<float4*>(a.sgtile + (((tl * 2 + h) * (long)Hp) + k) * 4) =
                    make_float4(g01.x * late, g01.y * late, sv * late, 0.0f);
            }
            {
                float pd0[16], pd1[16], s0[4], s1[4];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    pd0[i] = pd01[i].x;
                    pd1[i] = pd01[i].y;
                }
                half_reduce16(pd0, s0);
                half_reduce16(pd1, s1);
                if (nl < 4) {
                    const int q = ((nl & 1) << 1) | (nl >> 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long m = tl * 32 + 8 * q + 4 * h + r;
                        *reinterpret_cast<float2*>(a.dfpart + ((long)nb * a.Mp + m) * 2) = make_float2(s0[r] * late, s1[r] * late);
                    }
                }
            }
        }
        }
    };
    if (a.act == SVAE_ACT_TANH) epi(std::integral_constant<int, SVAE_ACT_TANH>());
    else epi(std::integral_constant<int, SVAE_ACT_SIGMOID>());
}

// Output layer backward in fp16x3 mode: dh[m][n] = (sum_c do[m][c] W_o[c][n]) act'(a[m][n]) leaves the pass ALREADY in the
// two operand forms of the GEMMs that consume it (row fragments for the data gradient, column fragments for the weight
// gradient, both scaled by gscale[0]) instead of an fp32 plane plus two conversion passes; fp32 is written only when a
// residual epilogue needs it (KEEP32).  Same LDS-tile scheme as layer0_fwd_split_kernel; a workgroup walks a range of
// 32-row tiles of one 64-feature block and keeps out_bwd_kernel's partial sums for dW_o / db_o:
// wpart[(chunk*4 + sub)][c][n], bpart[(chunk*4 + sub)][c], sub = (thread >> 7)*2 + half.
template <int ACT, int C, bool KEEP32>
__global__ void __launch_bounds__(256) out_bwd_split_kernel(const float* __restrict__ a, const float* __restrict__ do_p,
                                                             const float* __restrict__ out_w, float* __restrict__ dh32,
                                                             uint4* __restrict__ as, uint4* __restrict__ cs,
                                                             float* __restrict__ wpart, float* __restrict__ bpart,
                                                             float* __restrict__ hbpart, const float* __restrict__ gscale,
                                                             int H, int Hp, long Mp, long tiles, long tiles_per_chunk) {
    __shared__ float tile[32][65];
    const int f0 = blockIdx.x * 64;
    const int k = (threadIdx.x & 127) >> 1, h = threadIdx.x & 1, osub = threadIdx.x >> 7;
    const int n = f0 + k;
    const float sc = gscale[0];
    float w[C], pw[C], pb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        w[c] = (n < H) ? out_w[c * H + n] : 0.0f;
        pw[c] = 0.0f;
        pb[c] = 0.0f;
    }
    float hb = 0.0f;  // this thread's share of db_{L-1}[n] = sum_m dh[m][n] (the weight-gradient kernel no longer sums it)
    const int KC = Hp / 16, FT = Hp / 32;
    const long t0 = (long)blockIdx.y * tiles_per_chunk;
    const long t1 = (t0 + tiles_per_chunk < tiles) ? t0 + tiles_per_chunk : tiles;
    for (long T = t0; T < t1; ++T) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int oo = osub + 2 * it;
            const long o = T * 4 + oo;
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
            if (KEEP32) *reinterpret_cast<float4*>(dh32 + off) = da;
            hb += (da.x + da.y) + (da.z + da.w);
            const int r = oo * 8 + 4 * h;
            tile[r][k] = da.x; tile[r + 1][k] = da.y; tile[r + 2][k] = da.z; tile[r + 3][k] = da.w;
        }
        __syncthreads();
        {   // row fragments: thread = (K-step kq of this block's four, lane)
            const int kq = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const int r = lane & 31, c0 = kq * 16 + 8 * (lane >> 5);
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = tile[r][c0 + j];
            uint4 hi, lo;
            split8(x, sc, hi, lo);
            uint4* dst = as + ((T * KC + (f0 >> 4) + kq) * 2) * 64 + lane;
            dst[0] = hi;
            dst[64] = lo;
        }
        {   // column fragments: thread = (16-row step mq, tile fq, lane)
            const int mq = threadIdx.x >> 7, fq = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
            const int r0 = 16 * mq + 8 * (lane >> 5), c = 32 * fq + (lane & 31);
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = tile[r0 + j][c];
            uint4 hi, lo;
            split8(x, sc, hi, lo);
            uint4* dst = cs + (((T * 2 + mq) * FT + (f0 >> 5) + fq) * 2) * 64 + lane;
            dst[0] = hi;
            dst[64] = lo;
        }
        __syncthreads();
    }
    const long part = (long)blockIdx.y * 4 + osub * 2 + h;
#pragma unroll
    for (int c = 0; c < C; ++c) wpart[(part * C + c) * Hp + n] = pw[c];
    hbpart[part * Hp + n] = hb;
    if (n == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) bpart[part * C + c] = pb[c];
    }
}

// db[n] = sum over parts of hbpart[part][n], fixed order (8 interleaved chains per column, combined in LDS)
__global__ void colsum_reduce_kernel(const float* __restrict__ hbpart, float* __restrict__ db, int H, int Hp, int nparts) {
    __shared__ float red[8][33];
    const int col = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + col;
    float s = 0.0f;
    if (n < Hp) {  // Hp is the row stride; callers with unpadded rows (svae_colsum) have columns past it in the last block
        float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;  // four chains in a fixed order: several loads in flight per thread
        int i = pl;
        for (; i + 24 < nparts; i += 32) {
            s += hbpart[(long)i * Hp + n];
            s1 += hbpart[(long)(i + 8) * Hp + n];
            s2 += hbpart[(long)(i + 16) * Hp + n];
            s3 += hbpart[(long)(i + 24) * Hp + n];
        }
        for (; i < nparts; i += 8) s += hbpart[(long)i * Hp + n];
        s = (s + s1) + (s2 + s3);
    }
    red[pl][col] = s;
    __syncthreads();
    if (pl == 0 && n < H) {
        float t = red[0][col];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += red[j][col];
        db[n] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient in fp16x3 mode: dW[n][k] = sum_m dh[m][n] a[m][k], both operands as COLUMN fragments
//   Cs[ms][ft][part][lane]   ms = 16-row step, ft = 32-feature tile, lane = h*32 + f: rows 16 ms + 8 h + (0..7) of
//                            feature 32 ft + f -- the contraction (rows) runs inside the 16-byte fragment.
// fp32 octet-major stores exactly those 8 rows contiguously, so the conversion is a straight copy-and-split.
__global__ void split_cols_kernel(const float* __restrict__ in, uint4* __restrict__ cs, long noct, int Hp,
                                  const float* __restrict__ scale) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (octet, feature)
    if (idx >= noct * Hp) return;
    const int f = (int)(idx % Hp);
    const long o = idx / Hp;
    const float4* p = reinterpret_cast<const float4*>(in + idx * 8);
    const float4 v0 = p[0], v1 = p[1];
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint4 hi, lo;
    split8(x, scale ? scale[0] : kActScale, hi, lo);
    const int FT = Hp / 32;
    const long blk = (((o >> 1) * FT + (f >> 5)) * 2) * 64 + (o & 1) * 32 + (f & 31);
    cs[blk] = hi;
    cs[blk + 64] = lo;
}

struct SplitWgradArgs {
    const uint4* dh;     // Cs of the gradient (scaled by *gscale)
    const uint4* ap;     // Cs of a_{l-1}
    float* slab;         // [S][Hp][Hp] partial dW
    float* bslab;        // [S*2][Hp] partial db
    const float* gscale; // {s, 1/s} of the gradient
    long nsteps;         // Mp / 16
    int Hp, nblk1, S;    // nblk1 = 256-wide blocks per side, S = row-range splits
};
constexpr int kSplitWgradLds = 4 * 32 * 1024;

// One 256 x 256 block of dW per workgroup (blockIdx.x), one range of 16-row steps per blockIdx.y.  Per step the
// workgroup stages 8 + 8 feature tiles x (hi, lo) = 32 KiB by LDS-DMA (two steps ahead, four buffers, one barrier per two
// steps); wave (wi, wj)
// reads its 4 + 4 tiles from LDS and issues 4 x 4 x 3 MFMAs into 256 accumulator registers.
// PRIV: every wave stages its OWN 4 + 4 tiles into a private two-slot ring (16 KiB per slot) and the kernel has no barrier at
// all -- twice the L2 -> LDS traffic, but with one wave per SIMD a workgroup barrier stalls the matrix pipe of all four
// SIMDs (ablation: 0.31 ms with barriers, 0.22 ms without); !PRIV is the shared four-buffer scheme.
#ifndef SVAE_SPLIT_WGRAD_PRIV
#define SVAE_SPLIT_WGRAD_PRIV 1
#endif
template <bool BIAS, bool PRIV = (SVAE_SPLIT_WGRAD_PRIV != 0)>  // BIAS: also sum the gradient's columns (db)
__global__ __launch_bounds__(256, 1) void split_wgrad_kernel(SplitWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smw[];
    const int tid = threadIdx.x, lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    asm volatile("s_nop 4" : "+s"(wave));
    const int wi = wave >> 1, wj = wave & 1;
    const int nl = lane & 31, h = lane >> 5;
    const int Hp = a.Hp, FT = Hp / 32;
    // 1-D grid over (block of dW, row-range split).  The nblk^2 blocks of one split read the same rows of both operands:
    // when the split count is a multiple of 8 they get workgroup ids congruent mod 8, i.e. ONE XCD and its L2, so the
    // operands come from HBM once instead of nblk times (the kernel is HBM-bound at f16-MFMA speed).
    const int S = a.S, nb2 = a.nblk1 * a.nblk1;
    int blk, split;
    if (S % 8 == 0) {
        const int local = blockIdx.x >> 3;
        blk = local % nb2;
        split = (local / nb2) * 8 + (blockIdx.x & 7);
    } else {
        blk = blockIdx.x % nb2;
        split = blockIdx.x / nb2;
    }
    const int bn = blk / a.nblk1, bk = blk % a.nblk1;
    const long per = (a.nsteps + S - 1) / S;
    const long s0 = (long)split * per;
    long s1 = s0 + per;
    if (s1 > a.nsteps) s1 = a.nsteps;
    const int nst = s1 > s0 ? (int)(s1 - s0) : 0;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    // DMA pieces: block idx = wave + 4 j of a step's 32 (j < 4: gradient tiles, j >= 4: activation tiles)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)smw;
    unsigned poff[8], pm0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = wave + 4 * j;
        const int part = idx & 1, tile = (idx >> 1) & 7;
        int ft = (j < 4 ? bn : bk) * 8 + tile;
        ft = ft < FT ? ft : FT - 1;  // past the matrix edge: a valid tile, its products are never stored
        poff[j] = (unsigned)(((ft * 2 + part) * 64 + lane) * 16);
        pm0[j] = lds_base + (unsigned)idx * 1024u;
    }
    const unsigned step_bytes = (unsigned)FT * 2048u;
    const char* gbase = reinterpret_cast<const char*>(a.dh) + s0 * (long)step_bytes;
    const char* abase = reinterpret_cast<const char*>(a.ap) + s0 * (long)step_bytes;
    auto stage = [&](const char* gb, const char* ab, unsigned bufoff) {
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16_s(j < 4 ? gb : ab, poff[j], pm0[j] + bufoff);
    };
    // PRIV: this wave's 16 pieces of a step: block p = (which * 4 + i) * 2 + part of its slot (which 0 = gradient tile
    // wi*4 + i, 1 = activation tile wj*4 + i)
    unsigned qoff[16], qm0[16];
    if (PRIV) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int part = p & 1, i = (p >> 1) & 3, which = p >> 3;
            int ft = which ? bk * 8 + wj * 4 + i : bn * 8 + wi * 4 + i;
            ft = ft < FT ? ft : FT - 1;
            qoff[p] = (unsigned)(((ft * 2 + part) * 64 + lane) * 16);
            qm0[p] = lds_base + (unsigned)(wave * 32 + p) * 1024u;  // wave's ring: 2 slots x 16 KiB
        }
    }
    auto stage_priv = [&](const char* gb, const char* ab, unsigned slotoff) {
#pragma unroll
        for (int p = 0; p < 16; ++p) glds16_s(p < 8 ? gb : ab, qoff[p], qm0[p] + slotoff);
    };
    // fragment addresses inside a step's LDS image: shared (32 blocks: gradient tiles 0..7, activation tiles 8..15) or private
    const int gtile0 = PRIV ? 0 : wi * 4, atile0 = PRIV ? 4 : 8 + wj * 4;
    auto compute_step = [&](const uint4* buf) {
        // one wave per SIMD: nothing else hides the LDS latency, so the activation fragments of column tile j+1 are
        // read while the 12 MFMAs of tile j run (sched_barrier pins the order the source states)
        // reads in the order the first MFMAs need them (LDS returns in order): the first product can issue after two reads
        Frag gh[4], gl[4], ah[2], al[2];
        ah[0].u = buf[(atile0 * 2) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) gh[i].u = buf[((gtile0 + i) * 2) * 64 + lane];
        al[0].u = buf[(atile0 * 2 + 1) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) gl[i].u = buf[((gtile0 + i) * 2 + 1) * 64 + lane];
        if (BIAS && wj == 0) {  // bias gradient: column sums of the gradient tiles
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) t += (float)gh[i].h[e] + (float)gl[i].h[e];
                bs[i] += t;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j + 1 < 4) {
                ah[(j + 1) & 1].u = buf[((atile0 + j + 1) * 2) * 64 + lane];
                al[(j + 1) & 1].u = buf[((atile0 + j + 1) * 2 + 1) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            // the three products of one accumulator are three MFMAs apart (four independent accumulators in between)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[i].h, ah[j & 1].h, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[i].h, al[j & 1].h, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[i].h, ah[j & 1].h, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (PRIV && nst > 0) {
        const uint4* ring = smw + wave * 2048;  // 32 KiB per wave
        stage_priv(gbase, abase, 0u);
        for (int c = 0; c < nst; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of step c; nobody else reads them: no barrier
            const long sn = c + 1 < nst ? c + 1 : c;          // past the end: re-stage the last step (never read)
            if (!(SVAE_SPLIT_ABLATE & 256))
                stage_priv(gbase + sn * (long)step_bytes, abase + sn * (long)step_bytes, (unsigned)((c + 1) & 1) * 16u * 1024u);
            compute_step(ring + (c & 1) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant last stage must have landed before the wave ends
    } else if (nst > 0) {
        // four 32 KiB buffers, one barrier per TWO steps: at the barrier before steps (c, c+1) every wave has waited for its
        // pieces of both, nobody reads buffers (c+2) % 4 and (c+3) % 4 any more (steps c-2, c-1), and steps c+2, c+3 go out
        // into them -- two steps (~3000 MFMA cycles) ahead of their use
        int snext = 0;  // next step to stage; past the end the last step is re-staged (never read)
        auto stage_next = [&]() {
            const long sidx = snext < nst ? snext : nst - 1;
            stage(gbase + sidx * (long)step_bytes, abase + sidx * (long)step_bytes, (unsigned)(snext & 3) * 32u * 1024u);
            ++snext;
        };
        stage_next();
        stage_next();
        for (int c = 0; c < nst; c += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(SVAE_SPLIT_ABLATE & 512) || c == 0) __syncthreads();
            if (!(SVAE_SPLIT_ABLATE & 256)) {
                stage_next();
                stage_next();
            }
            compute_step(smw + (c & 3) * 2048);
            if (c + 1 < nst) compute_step(smw + ((c + 1) & 3) * 2048);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // slab[split][n][k]: lane holds column k, registers rows n = (r & 3) + 8 (r >> 2) + 4 h
    const float ginv = a.gscale[1];
    const float inv = ginv * kActInv;
    float* slab = a.slab + (long)split * Hp * Hp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nt = bn * 8 + wi * 4 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kt = bk * 8 + wj * 4 + j;
            if (nt < FT && kt < FT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    slab[(long)n * Hp + kt * 32 + nl] = acc[i][j][r] * inv;
                }
            }
        }
    }
    if (BIAS && bk == 0 && wj == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nt = bn * 8 + wi * 4 + i;
            if (nt < FT) a.bslab[((long)split * 2 + h) * Hp + nt * 32 + nl] = bs[i] * ginv;
        }
    }
}

}  // namespace svae
