// dense4_kernel: the hidden-layer GEMM with FOUR row tiles per wave fed by ONE 16-byte load per k-step.
// (nn.Linear + activation of SpatialGenerator.layers, spatial_vae/models.py:77-83, 126, and its
// autograd backward -- the same three roles as dense_kernel: forward, data gradient, their fused epilogues.)
//
// Why.  In dense_kernel a lane supplies one row-operand dword per k-step, so a wave issues one global_load_dword per NT
// MFMAs.  Timing that kernel with parts compiled out (tools/dense_ablate.hip, r03) shows those loads cost 8.7 % of the launch
// (issued but never awaited: no change, so it is their ISSUE, not their latency), the B-fragment ds_reads 6 %, the epilogue
// 2.6 %, the weight DMA + barriers 1 %: on gfx950 every vector-memory or VALU instruction between fp32 MFMAs takes
// matrix-pipe time (tools/mfma_valu_probe.hip: the fp32 MFMA shares the vector ALUs; VALU phases of one wave do not run
// under another wave's MFMAs).  So the lever is instructions per MFMA.
// The octet-major layout stores FOUR CONSECUTIVE ROWS of one feature in 16 bytes.  Lane i' (0..31) of a wave owns rows
// 4i' .. 4i'+3 of a 128-row group and loads them with one global_load_dwordx4: component c is row 4i'+c = row i' of the
// "virtual" tile V_c = {rows = c mod 4} of the group, and the four components are the A operands of four MFMAs (tiles
// V_0..V_3) that share one B value.  Per k-step 1 vector load and NT B values feed 4 NT MFMAs: with NT = 2 (128 accumulator
// registers, two waves per SIMD) that is 6 vector-memory instructions per 32 MFMAs where dense_kernel<4> issues 8 per 16.
// Output: accumulator register 4q+r of tile V_c holds virtual row 8q+4h+r = row 32q + 16h + 4r + c of the group, so the four
// tiles' registers of one (q, r) are four consecutive rows -- one 16-byte octet-major store, as before -- and q is the
// ACTUAL 32-row tile of the group (tiles never straddle images: the per-tile reductions of the FIRST epilogue carry over).
// Numerics: per output element the same k-ordered fma chain as dense_kernel; the plain forward is bit-identical to it
// (tools/dense4_proto.hip).  Measured on random data (same tool, MI355X): H = 500, 204 800 rows 0.856 -> 0.802 ms; H = 1024
// 0.536 -> 0.483 ms; at 51 200 rows (BASELINE cfg 1) the coarser work quantum loses (0.240 -> 0.247 ms), so small launches
// stay with dense_kernel (use_dense4 in api.hip).
#pragma once
#include "dense.h"

namespace svae {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ void load_a4(const float* p, f32x4v& v) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "+v"(v) : "v"(p), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm4(f32x4v& v) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "i"(N) : "memory");
}
// act'(a) of the rank-1 output-layer forms for four row operands in two packed instructions (one fma per element costs ~5 %
// of the launch here, the packed form ~4 %: tools/dense4_proto.hip): tanh 1 - a^2, sigmoid a - a^2
template <int LASTD>
__device__ __forceinline__ f32x4v rank1_actgrad(f32x4v x) {
    f32x2v lo = {x[0], x[1]}, hi = {x[2], x[3]};
    if (LASTD == 2) {
        const f32x2v one = {1.0f, 1.0f};
        asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(lo) : "v"(lo), "v"(one));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(hi) : "v"(hi), "v"(one));
    } else {
        asm volatile("v_pk_fma_f32 %0, %1, %1, %1 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(lo) : "v"(lo));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %1 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(hi) : "v"(hi));
    }
    return f32x4v{lo[0], lo[1], hi[0], hi[1]};
}

// Same arguments as dense_kernel (DenseArgs; a.resid must be 0); `groups` = Mp / 128 (the caller guarantees Mp % 128 == 0).
//   DGRAD  data gradient: the epilogue multiplies by act'(aux)
//   FIRST  (DGRAD) data gradient into the coordinate layer: reduce instead of store (dense_kernel's FIRST epilogue)
//   LASTD  0, or 2 (tanh) / 3 (sigmoid): the rank-1 output-layer form (DGRAD): `in` is a_{L-1}, rows scaled by do[m] in the epilogue
//   CF     forward of the last hidden layer: the epilogue also contracts with W_o (a.C channels) into a.lpart
template <int NT, bool DGRAD, bool FIRST, int LASTD, int CF>
__global__ __launch_bounds__(256, 2) void dense4_kernel(DenseArgs a, long groups) {
    static_assert(NT == 1 || NT == 2, "4 NT accumulator tiles: two waves per SIMD up to NT = 2");
    static_assert(!FIRST || DGRAD, "FIRST is a data-gradient epilogue");
    static_assert(LASTD == 0 || ((LASTD == 2 || LASTD == 3) && DGRAD), "LASTD: the rank-1 data-gradient forms only");
    static_assert(CF == 0 || (!DGRAD && CF == 1), "CF is a forward epilogue (a flag: the channel count is a.C)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using Cfg = DenseCfg<NT>;
    constexpr int NB = Cfg::NB, G = Cfg::G, CHUNK = Cfg::CHUNK, NINSTR = Cfg::NINSTR;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int Hp = a.Hp;
    const int noct = Hp / 8;
    const int nchunk = noct / G;
    const int ntile = Hp / 32;
    // 1-D XCD-aware grid: the column blocks of one set of 4 row groups get consecutive ids on ONE XCD (one L2)
    const int nblk = ntile / NT;
    const long local = blockIdx.x >> 3;
    const int nb = (int)(local % nblk);
    const long set = (local / nblk) * 8 + (blockIdx.x & 7);
    const long rg = set * 4 + wave;
    const bool live = rg < groups;
    const long rgl = live ? rg : groups - 1;  // dead waves recompute the last group and store nothing

    // this lane's 16 bytes (rows 4 nl .. 4 nl + 3 of the group) of feature k: ((16 rgl + nl/2) Hp + k) 8 + 4 (nl & 1)
    const float* arow = a.in + ((rgl * 16 + (nl >> 1)) * (long)Hp + 4 * h) * 8 + (nl & 1) * 4;
    const float* bfrag = smem + lane * 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;

    f32x16 acc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.0f;

    // weight chunks: dense_kernel's staging (LDS-DMA from inline asm, double-buffered, one barrier per chunk)
    constexpr int PW = (NINSTR + 3) / 4;
    unsigned poff[PW], pm0[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int idx = wave + 4 * j;
        const int gl = (idx < NINSTR ? idx : NINSTR - 1) / NT, tt = (idx < NINSTR ? idx : NINSTR - 1) % NT;
        poff[j] = (unsigned)(((gl * ntile + nb * NT + tt) * 256 + lane * 4) * 4);
        pm0[j] = lds_base + (unsigned)(idx < NINSTR ? idx : NINSTR - 1) * 1024u;
    }
    const unsigned chunk_bytes = (unsigned)(G * ntile) * 1024u;
    auto stage_piece = [&](int c, int buf, int j) {
        glds16_s(reinterpret_cast<const char*>(a.wp) + (long)c * chunk_bytes, poff[j], pm0[j] + (unsigned)(buf * CHUNK) * 4u);
    };
    auto stage = [&](int c, int buf) {
#pragma unroll
        for (int j = 0; j < PW; ++j) stage_piece(c, buf, j);
    };
    auto read_b = [&](int o, float4 (&b)[NT]) {
        const float* base = bfrag + ((o / G) & 1) * CHUNK + (o % G) * NT * 256;
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const float4*>(base + t * 256);
    };
    // Queue proof, per wave (as in dense_kernel): a slot is [wait][4 NT MFMAs][A load][DMA pieces, last octet only].  Between
    // the issue of an A load and its use exactly one chunk (16 slots) later lie 15 A loads and P DMA pieces: vmcnt(15 + P).
    // Before the barrier of chunk c the last DMA piece of chunk c+1 (issued at the end of chunk c-1) has the 12 A loads of
    // slots 0..11 behind it: vmcnt(12).
    constexpr int P = PW;
    constexpr int PE = (P + 3) / 4;
    static_assert(G == 4, "the slot offsets assume 4 octets per chunk");

    stage(0, 0);
    stage(nchunk > 1 ? 1 : 0, 1);
    f32x4v av[G][4];
    float4 b0[NT], b1[NT];
#pragma unroll
    for (int gl = 0; gl < G; ++gl)
#pragma unroll
        for (int e = 0; e < 4; ++e) av[gl][e] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    load_a4<0>(arow, av[0][0]); load_a4<32>(arow, av[0][1]); load_a4<64>(arow, av[0][2]); load_a4<96>(arow, av[0][3]);
    load_a4<256>(arow, av[1][0]); load_a4<288>(arow, av[1][1]); load_a4<320>(arow, av[1][2]); load_a4<352>(arow, av[1][3]);
    load_a4<512>(arow, av[2][0]); load_a4<544>(arow, av[2][1]); load_a4<576>(arow, av[2][2]); load_a4<608>(arow, av[2][3]);
    load_a4<768>(arow, av[3][0]); load_a4<800>(arow, av[3][1]); load_a4<832>(arow, av[3][2]); load_a4<864>(arow, av[3][3]);
#pragma unroll
    for (int gl = 0; gl < G; ++gl)
#pragma unroll
        for (int e = 0; e < 4; ++e) wait_vm4<0>(av[gl][e]);
    __syncthreads();  // chunks 0 and 1 have landed in LDS
    read_b(0, b0);
    const int spare = nchunk & 1;  // buffer that held chunk nchunk-2: the sink of redundant re-stages
    for (int c = 0; c < nchunk; ++c) {
        const float* anext = arow + (long)(c + 1 < nchunk ? c + 1 : nchunk - 1) * (G * 64);
        const bool more = c + 2 < nchunk;
        const int cstage = more ? c + 2 : nchunk - 1, bstage = more ? (c & 1) : spare;
#pragma unroll
        for (int gl = 0; gl < G; ++gl) {
            const int o = c * G + gl;
            if (gl == G - 1) {
                wait_vm4<12>(av[gl][0]);
                __syncthreads();
            }
            const int onext = (o + 1 < noct) ? o + 1 : noct - 1;
            if (gl & 1) read_b(onext, b0); else read_b(onext, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wait_vm4<15 + P>(av[gl][e]);
                const f32x4v x = LASTD != 0 ? rank1_actgrad<LASTD>(av[gl][e]) : av[gl][e];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 bb = (gl & 1) ? b1[t] : b0[t];
                    const float bv = e == 0 ? bb.x : e == 1 ? bb.y : e == 2 ? bb.z : bb.w;
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[0], bv, acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[1], bv, acc[1][t], 0, 0, 0);
                    acc[2][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[2], bv, acc[2][t], 0, 0, 0);
                    acc[3][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[3], bv, acc[3][t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // re-issue this register quad's load for the next chunk (byte offset 256*gl + 32*e)
                if (gl == 0 && e == 0) load_a4<0>(anext, av[0][0]);
                if (gl == 0 && e == 1) load_a4<32>(anext, av[0][1]);
                if (gl == 0 && e == 2) load_a4<64>(anext, av[0][2]);
                if (gl == 0 && e == 3) load_a4<96>(anext, av[0][3]);
                if (gl == 1 && e == 0) load_a4<256>(anext, av[1][0]);
                if (gl == 1 && e == 1) load_a4<288>(anext, av[1][1]);
                if (gl == 1 && e == 2) load_a4<320>(anext, av[1][2]);
                if (gl == 1 && e == 3) load_a4<352>(anext, av[1][3]);
                if (gl == 2 && e == 0) load_a4<512>(anext, av[2][0]);
                if (gl == 2 && e == 1) load_a4<544>(anext, av[2][1]);
                if (gl == 2 && e == 2) load_a4<576>(anext, av[2][2]);
                if (gl == 2 && e == 3) load_a4<608>(anext, av[2][3]);
                if (gl == 3 && e == 0) load_a4<768>(anext, av[3][0]);
                if (gl == 3 && e == 1) load_a4<800>(anext, av[3][1]);
                if (gl == 3 && e == 2) load_a4<832>(anext, av[3][2]);
                if (gl == 3 && e == 3) load_a4<864>(anext, av[3][3]);
                if (gl == G - 1) {  // this wave's DMA pieces of chunk c+2, PE per k-step
#pragma unroll
                    for (int j = 0; j < PE; ++j)
                        if (e * PE + j < P) stage_piece(cstage, bstage, e * PE + j);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    {   // drain: the last A loads and DMA pieces are in flight and never used; keep their registers tied up
        float sink = 0.0f;
#pragma unroll
        for (int gl = 0; gl < G; ++gl)
#pragma unroll
            for (int e = 0; e < 4; ++e) { wait_vm4<0>(av[gl][e]); sink += av[gl][e][0]; }
        if (a.tiles < 0) a.out[0] = sink;  // never true
    }
    if (!live) return;

    // ---- epilogue.  A "slab" s = q * NT + t is the 16 rows 32q + 16h + 4r + c (r, c = 0..3) of column n of tile t: four
    // 16-byte vectors (one per r, the four tiles' registers 4q + r).  Loads of slab s+1 are issued before the stores of slab s
    // (vmcnt counts stores too, in issue order).
    auto epi = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr int NS = 4 * NT;
        float bias[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = nb * NB + t * 32 + nl;
            const float bv = DGRAD ? 0.0f : a.bias[n < a.H ? n : a.H - 1];
            bias[t] = (n < a.H) ? bv : 0.0f;
        }
        // float offset of vector (q, r) of column tile t in an octet-major plane
        auto voff = [&](int q, int r, int t) -> long {
            return ((rgl * 16 + 4 * q + 2 * h + (r >> 1)) * (long)Hp + nb * NB + t * 32 + nl) * 8 + 4 * (r & 1);
        };
        float4 xa[2][4];   // aux (a_{l-1}) of slab s / s+1
        float4 dqv[2][4];  // rank-1 forms: d(loss)/d(logit) of the 16 rows of tile q / q+1 (zero on pad rows)
        auto fetch_aux = [&](int s, float4 (&fa)[4]) {
            const int q = s / NT, t = s % NT;
#pragma unroll
            for (int r = 0; r < 4; ++r) fa[r] = *reinterpret_cast<const float4*>(a.aux + voff(q, r, t));
        };
        auto fetch_dq = [&](int q, float4 (&d)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = *reinterpret_cast<const float4*>(a.do_p + rgl * 128 + 32 * q + 16 * h + 4 * r);
        };
        // vector (q, r) of tile t after the elementwise part of the epilogue
        auto value = [&](int q, int r, int t, const float4& aux, const float4& dq) -> float4 {
            float4 v = make_float4(acc[0][t][4 * q + r], acc[1][t][4 * q + r], acc[2][t][4 * q + r], acc[3][t][4 * q + r]);
            v = dense_epilogue<ACT, DGRAD>(v, bias[t], aux);
            if (LASTD != 0) { v.x *= dq.x; v.y *= dq.y; v.z *= dq.z; v.w *= dq.w; }
            return v;
        };
        if (!FIRST) {
            if (DGRAD) fetch_aux(0, xa[0]);
            if (LASTD != 0) fetch_dq(0, dqv[0]);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int q = s / NT, t = s % NT;
                if (s + 1 < NS) {
                    if (DGRAD) fetch_aux(s + 1, xa[(s + 1) & 1]);
                    if (LASTD != 0 && (s + 1) % NT == 0) fetch_dq((s + 1) / NT, dqv[((s + 1) / NT) & 1]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 v = value(q, r, t, xa[s & 1][r], dqv[q & 1][r]);
                    *reinterpret_cast<float4*>(a.out + voff(q, r, t)) = v;
                    if (CF > 0) {  // the activations stay in the accumulator registers for the contraction with W_o below
                        acc[0][t][4 * q + r] = v.x; acc[1][t][4 * q + r] = v.y; acc[2][t][4 * q + r] = v.z; acc[3][t][4 * q + r] = v.w;
                    }
                }
            }
            if (CF > 0) {
                // partial logits of this block's columns: per output channel (a.C is wave-uniform) and 32-row tile q, the 16
                // rows' products are summed over the column tiles in the order t = 0 .. NT-1, then over the 32 lanes of each
                // half-wave (half_reduce16: lane class cls ends up with rows 16h + 4 cls .. +3 of the tile)
                for (int c = 0; c < a.C; ++c) {
                    float wo[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int n = nb * NB + t * 32 + nl;
                        const float wv = a.out_w[c * a.H + (n < a.H ? n : a.H - 1)];
                        wo[t] = (n < a.H) ? wv : 0.0f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float lp[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) lp[j] = 0.0f;
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc) lp[4 * r + cc] += acc[cc][t][4 * q + r] * wo[t];
                        float s4[4];
                        half_reduce16(lp, s4);
                        if (nl < 4) {
                            const int cls = ((nl & 1) << 1) | (nl >> 1);
                            *reinterpret_cast<float4*>(a.lpart + ((long)nb * a.C + c) * a.Mp + rgl * 128 + 32 * q + 16 * h + 4 * cls) =
                                make_float4(s4[0], s4[1], s4[2], s4[3]);
                        }
                    }
                }
            }
        } else {
            // FIRST: dh0 is reduced on the spot (dense_kernel's FIRST epilogue, per actual tile q of the group): over the tile's
            // rows into (G0, G1, S) per feature, over this block's features into d(coords) per row.
            fetch_aux(0, xa[0]);
            if (LASTD != 0) fetch_dq(0, dqv[0]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long tile = rgl * 4 + q;
                const int b = (int)(tile / a.Timg);
                const int i0 = (int)(tile % a.Timg) * 32 + 16 * h;  // this lane's rows of the image: i0 + 4r + c
                const float4 pb = a.posebuf[b];  // identity (1, 0, 0, 0) when the coordinates are explicit
                const float* cbase = a.pose.coords ? a.pose.coords + (long)b * a.N * 2 : a.pose.grid;
                float2 raw[16];
                float x0[16], x1[16], pd0[16], pd1[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {  // 16 independent loads, no branches: pad rows re-read row N-1
                    const int i = i0 + j;
                    raw[j] = *reinterpret_cast<const float2*>(cbase + (long)(i < a.N ? i : a.N - 1) * 2);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool in = i0 + j < a.N;
                    x0[j] = in ? pb.x * raw[j].x - pb.y * raw[j].y + pb.z : 0.0f;
                    x1[j] = in ? pb.y * raw[j].x + pb.x * raw[j].y + pb.w : 0.0f;
                    pd0[j] = 0.0f; pd1[j] = 0.0f;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int s = q * NT + t;
                    const int k = nb * NB + t * 32 + nl;
                    const float2 w = *reinterpret_cast<const float2*>(a.tab + ((long)b * Hp + k) * kSlots);
                    if (s + 1 < NS) {
                        fetch_aux(s + 1, xa[(s + 1) & 1]);
                        if (LASTD != 0 && (s + 1) % NT == 0) fetch_dq((s + 1) / NT, dqv[((s + 1) / NT) & 1]);
                    }
                    float sv = 0.0f, g0 = 0.0f, g1 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float4 v = value(q, r, t, xa[s & 1][r], dqv[q & 1][r]);
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            const int j = 4 * r + cc;
                            sv += vv[cc];
                            g0 += vv[cc] * x0[j];
                            g1 += vv[cc] * x1[j];
                            pd0[j] += vv[cc] * w.x;
                            pd1[j] += vv[cc] * w.y;
                        }
                    }
                    // the two half-waves hold the two row halves of the same (tile, feature): one sum
                    g0 += __shfl_xor(g0, 32);
                    g1 += __shfl_xor(g1, 32);
                    sv += __shfl_xor(sv, 32);
                    if (h == 0) *reinterpret_cast<float4*>(a.sgtile + (tile * (long)Hp + k) * 4) = make_float4(g0, g1, sv, 0.0f);
                }
                // d(coords) of each row: this block's NB features = over the tiles (done) and the 32 lanes
                float s0[4], s1[4];
                half_reduce16(pd0, s0);
                half_reduce16(pd1, s1);
                if (nl < 4) {
                    const int cls = ((nl & 1) << 1) | (nl >> 1);
                    const long m = tile * 32 + 16 * h + 4 * cls;
                    float4* dst = reinterpret_cast<float4*>(a.dfpart + ((long)nb * a.Mp + m) * 2);
                    dst[0] = make_float4(s0[0], s1[0], s0[1], s1[1]);
                    dst[1] = make_float4(s0[2], s1[2], s0[3], s1[3]);
                }
            }
        }
    };
    switch (a.act) {
        case SVAE_ACT_TANH: epi(std::integral_constant<int, SVAE_ACT_TANH>()); break;
        case SVAE_ACT_LEAKYRELU: epi(std::integral_constant<int, SVAE_ACT_LEAKYRELU>()); break;
        case SVAE_ACT_RELU: epi(std::integral_constant<int, SVAE_ACT_RELU>()); break;
        default: epi(std::integral_constant<int, SVAE_ACT_SIGMOID>()); break;
    }
}

}  // namespace svae
