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
// 0.536 -> 0.483 ms; at 51 200 rows (BASELINE cfg 1) the NT = 2 form's coarser work quantum loses (0.240 -> 0.247 ms) and the
// half-width form NT = 1 (three waves per SIMD) is taken instead (use_dense4 in api.hip).
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
// act'(a) of the rank-1 output-layer forms for the four row operands of a k-step: tanh 1 - a^2, sigmoid a - a^2.  Plain fma's
// that the compiler sees (it may pack them itself): r03 tried the packed form from inline asm (v_pk_fma_f32 x 2).  It is ~0.7 %
// faster in isolation (tools/dense4_proto.hip) but an MFMA may read a VGPR written by a packed-fp32 instruction only two wait
// states later and hipcc's hazard recognizer does not look inside inline asm: the NT = 1 instantiation came out with ONE wait
// state and read stale operands (gradients off by 3-30 %), and both repairs -- an s_nop behind the pair, or this slot's
// row-operand load moved between the pair and the MFMAs -- cost 2.5 % of the launch (0.840 -> 0.862 ms): an idle issue slot in
// front of an MFMA batch is paid in full.
template <int LASTD>
__device__ __forceinline__ f32x4v rank1_actgrad(f32x4v x) {
    f32x4v r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(-x[i], x[i], LASTD == 2 ? 1.0f : x[i]);
    return r;
}

// Same arguments as dense_kernel (DenseArgs; a.resid must be 0); `groups` = Mp / 128 (the caller guarantees Mp % 128 == 0);
// the launch covers the sets (4 row groups each) from set0 on: a layer may be split into a wide launch over the sets that fill
// whole rounds of resident workgroups and a half-width launch over the rest (launch_dense in api.hip).
//   DGRAD  data gradient: the epilogue multiplies by act'(aux)
//   FIRST  (DGRAD) data gradient into the coordinate layer: reduce instead of store (dense_kernel's FIRST epilogue)
//   LASTD  0, or 2 (tanh) / 3 (sigmoid): the rank-1 output-layer form (DGRAD): `in` is a_{L-1}, rows scaled by do[m] in the epilogue
//   CF     forward of the last hidden layer: the epilogue also contracts with W_o (CF = a.C channels) into a.lpart
// waves per SIMD the register allocation is sized for: 4 NT accumulator tiles of 16 registers + 64 row-operand registers
template <int NT>
struct Dense4Occ {
    static constexpr int value = NT == 1 ? 3 : 2;
};

// The workgroup's program, shared by dense4_kernel (one block width per launch) and dense4_dual_kernel (both widths in one
// launch); `bid` is the workgroup's index among those of its width.
template <int NT, bool DGRAD, bool FIRST, int LASTD, int CF>
__device__ __forceinline__ void dense4_body(const DenseArgs& a, long groups, long set0, unsigned bid) {
    static_assert(NT == 1 || NT == 2, "4 NT accumulator tiles: two waves per SIMD up to NT = 2");
    static_assert(!FIRST || DGRAD, "FIRST is a data-gradient epilogue");
    static_assert(LASTD == 0 || ((LASTD == 2 || LASTD == 3) && DGRAD), "LASTD: the rank-1 data-gradient forms only");
    static_assert(CF == 0 || (!DGRAD && CF <= SVAE_MAX_OUT), "CF = output channels of the forward epilogue's W_o contraction");
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
    const long local = bid >> 3;
    const int nb = (int)(local % nblk);
    const long set = set0 + (local / nblk) * 8 + (bid & 7);   // set0: the first set of 4 row groups of this width's range
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

    // FIRST: the posed coordinates (x0, x1) of this wave's 128 rows (0 on pad rows) and, for the rank-1 forms, their
    // d(loss)/d(logit) go into a wave-private LDS area behind the weight buffers now, two rows per lane; the epilogue reads them
    // back four rows at a time (wave-private: no barrier, the compiler's lgkmcnt waits order the write and the reads)
    float* rowx = smem + 2 * CHUNK + wave * (3 * 128);  // [x0 | x1 | do] x 128 rows
    if (FIRST) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = lane + 64 * rr;
            const long tile = rgl * 4 + (row >> 5);
            const int b = (int)(tile / a.Timg);
            const int i = (int)(tile % a.Timg) * 32 + (row & 31);
            const float4 pb = a.posebuf[b];  // identity (1, 0, 0, 0) when the coordinates are explicit
            const float* cbase = a.pose.coords ? a.pose.coords + (long)b * a.N * 2 : a.pose.grid;
            const float2 g2 = *reinterpret_cast<const float2*>(cbase + (long)(i < a.N ? i : a.N - 1) * 2);
            const bool in = i < a.N;
            rowx[row] = in ? pb.x * g2.x - pb.y * g2.y + pb.z : 0.0f;
            rowx[128 + row] = in ? pb.y * g2.x + pb.x * g2.y + pb.w : 0.0f;
            if (LASTD != 0) rowx[256 + row] = a.do_p[rgl * 128 + row];
        }
    }
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
    // one k-step: the four row tiles x NT column tiles that share this A vector and these B values
    auto kstep = [&](const f32x4v& x, const float4 (&bf)[NT], int e) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float bv = e == 0 ? bf[t].x : e == 1 ? bf[t].y : e == 2 ? bf[t].z : bf[t].w;
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[0], bv, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[1], bv, acc[1][t], 0, 0, 0);
            acc[2][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[2], bv, acc[2][t], 0, 0, 0);
            acc[3][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[3], bv, acc[3][t], 0, 0, 0);
        }
    };
    // re-issue the load of register quad (gl, e) for the next chunk (byte offset 256 gl + 32 e)
    auto reload = [&](int gl, int e, const float* anext) {
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
    };
    // all chunks but the last: MFMAs of chunk c, A loads of chunk c+1, DMA of chunk c+2
    for (int c = 0; c + 1 < nchunk; ++c) {
        const float* anext = arow + (long)(c + 1) * (G * 64);
        const bool more = c + 2 < nchunk;
        const int cstage = more ? c + 2 : nchunk - 1, bstage = more ? (c & 1) : spare;
#pragma unroll
        for (int gl = 0; gl < G; ++gl) {
            const int o = c * G + gl;
            if (gl == G - 1) {
                // the next octet opens chunk c+1: its DMA must have landed in every wave's view, and every wave must hold its
                // last fragments of chunk c before that buffer is reused
                wait_vm4<12>(av[gl][0]);
                __syncthreads();
            }
            if (gl & 1) read_b(o + 1, b0); else read_b(o + 1, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wait_vm4<15 + P>(av[gl][e]);
                const f32x4v x = LASTD != 0 ? rank1_actgrad<LASTD>(av[gl][e]) : av[gl][e];
                // all four transforms first, then the MFMAs: the first MFMA then reads a value written three instructions
                // earlier (left alone, hipcc alternates fma / s_nop 1 / MFMA at NT = 1: two idle wait states per MFMA)
                if (LASTD != 0) __builtin_amdgcn_sched_barrier(0);
                if (gl & 1) kstep(x, b1, e); else kstep(x, b0, e);
                __builtin_amdgcn_sched_barrier(0);
                reload(gl, e, anext);
                if (gl == G - 1) {  // this wave's DMA pieces of chunk c+2, PE per k-step
#pragma unroll
                    for (int j = 0; j < PE; ++j)
                        if (e * PE + j < P) stage_piece(cstage, bstage, e * PE + j);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    {
        // The last chunk: nothing left to fetch (the loop above used to re-load and re-stage the last chunk redundantly), so
        // everything in flight is drained once and the octets run as bare MFMAs -- and only the octets that hold at least one
        // real contraction index: with H = 500 the 64th octet (k = 504..511) is zero padding in both operands, 1.6 % of the
        // MFMAs.  Its weights were published by the barrier of chunk nchunk-2 (or the prologue's): no barrier here.
#pragma unroll
        for (int gl = 0; gl < G; ++gl)
#pragma unroll
            for (int e = 0; e < 4; ++e) wait_vm4<0>(av[gl][e]);
        const int o0 = (nchunk - 1) * G;
        const int gtail = (a.H + 7) / 8 - o0;  // 1 .. 4 octets with real data (Hp - H < 32)
#pragma unroll
        for (int gl = 0; gl < G; ++gl) {
            if (gl < gtail) {
                if (gl + 1 < G) { if (gl & 1) read_b(o0 + gl + 1, b0); else read_b(o0 + gl + 1, b1); }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4v x = LASTD != 0 ? rank1_actgrad<LASTD>(av[gl][e]) : av[gl][e];
                    if (LASTD != 0) __builtin_amdgcn_sched_barrier(0);
                    if (gl & 1) kstep(x, b1, e); else kstep(x, b0, e);
                    if (LASTD != 0) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (!live) return;

    // ---- epilogue.  A "slab" s = q * NT + t is the 16 rows 32q + 16h + 4r + c (r, c = 0..3) of column n of tile t: four
    // 16-byte vectors (one per r, the four tiles' registers 4q + r).  Everything is ordered tile by tile (q outermost), so the
    // accumulator registers of a finished tile are dead and the per-tile state of the fused forms fits beside the rest:
    // no spills (a spilled value costs a scratch round trip per use here, and the wave sits in its epilogue meanwhile).
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
        // vector (q, r) of tile t after the elementwise part of the epilogue
        auto value = [&](int q, int r, int t, const float4& aux, const float4& dq) -> float4 {
            float4 v = make_float4(acc[0][t][4 * q + r], acc[1][t][4 * q + r], acc[2][t][4 * q + r], acc[3][t][4 * q + r]);
            v = dense_epilogue<ACT, DGRAD>(v, bias[t], aux);
            if (LASTD != 0) { v.x *= dq.x; v.y *= dq.y; v.z *= dq.z; v.w *= dq.w; }
            return v;
        };
        if (!FIRST) {
            float4 xa[2][4];   // aux (a_{l-1}) of slab s / s+1: loads of slab s+1 are issued before the stores of slab s
            float4 dqv[2][4];  // rank-1 forms: d(loss)/d(logit) of the 16 rows of tile q / q+1 (zero on pad rows)
            auto fetch_aux = [&](int s_, float4 (&fa)[4]) {
                const int q = s_ / NT, t = s_ % NT;
#pragma unroll
                for (int r = 0; r < 4; ++r) fa[r] = *reinterpret_cast<const float4*>(a.aux + voff(q, r, t));
            };
            auto fetch_dq = [&](int q, float4 (&d)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = *reinterpret_cast<const float4*>(a.do_p + rgl * 128 + 32 * q + 16 * h + 4 * r);
            };
            // CF: this lane's W_o entries, and per channel the partial logits of the current tile's 16 rows
            constexpr int CFN = CF > 0 ? CF : 1;
            float wo[CFN][NT], lp[CFN][16];
            if (CF > 0) {
#pragma unroll
                for (int c = 0; c < CFN; ++c)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int n = nb * NB + t * 32 + nl;
                        const float wv = a.out_w[c * a.H + (n < a.H ? n : a.H - 1)];
                        wo[c][t] = (n < a.H) ? wv : 0.0f;
                    }
            }
            if (DGRAD) fetch_aux(0, xa[0]);
            if (LASTD != 0) fetch_dq(0, dqv[0]);
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                const int q = s_ / NT, t = s_ % NT;
                if (s_ + 1 < NS) {
                    if (DGRAD) fetch_aux(s_ + 1, xa[(s_ + 1) & 1]);
                    if (LASTD != 0 && (s_ + 1) % NT == 0) fetch_dq((s_ + 1) / NT, dqv[((s_ + 1) / NT) & 1]);
                }
                if (CF > 0 && t == 0) {
#pragma unroll
                    for (int c = 0; c < CFN; ++c)
#pragma unroll
                        for (int j = 0; j < 16; ++j) lp[c][j] = 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 v = value(q, r, t, xa[s_ & 1][r], dqv[q & 1][r]);
                    *reinterpret_cast<float4*>(a.out + voff(q, r, t)) = v;
                    if (CF > 0) {
#pragma unroll
                        for (int c = 0; c < CFN; ++c) {
                            lp[c][4 * r] += v.x * wo[c][t]; lp[c][4 * r + 1] += v.y * wo[c][t];
                            lp[c][4 * r + 2] += v.z * wo[c][t]; lp[c][4 * r + 3] += v.w * wo[c][t];
                        }
                    }
                }
                if (CF > 0 && t == NT - 1) {
                    // partial logits of tile q over this block's columns: summed over the column tiles in the order t = 0 .. NT-1
                    // (above), then over the 32 lanes of each half-wave (half_reduce16: lane class cls ends up with rows
                    // 16h + 4 cls .. +3 of the tile)
#pragma unroll
                    for (int c = 0; c < CFN; ++c) {
                        float s4[4];
                        half_reduce16(lp[c], s4);
                        if (nl < 4) {
                            const int cls = ((nl & 1) << 1) | (nl >> 1);
                            *reinterpret_cast<float4*>(a.lpart + ((long)nb * CFN + c) * a.Mp + rgl * 128 + 32 * q + 16 * h + 4 * cls) =
                                make_float4(s4[0], s4[1], s4[2], s4[3]);
                        }
                    }
                }
            }
        } else {
            // FIRST: dh0 is reduced on the spot (dense_kernel's FIRST epilogue, per actual tile q of the group): over the tile's
            // rows into (G0, G1, S) per feature, over this block's features into d(coords) per row.  Each tile goes in two
            // half passes of 8 rows per lane (rows 16h + 8 rh + j, j = 0..7 <-> r = 2 rh + j/4, c = j % 4), which halves the
            // per-tile state and lets ONE half_reduce16 serve both coordinate components.  Per-row operands come from the
            // LDS area staged in the prologue; the a_0 vectors of the next half pass are loaded before this one is computed.
            float4 xa[2][NT][2];
            auto fetch_aux = [&](int hp, float4 (&fa)[NT][2]) {
                const int q = hp >> 1, rh = hp & 1;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) fa[t][r2] = *reinterpret_cast<const float4*>(a.aux + voff(q, 2 * rh + r2, t));
            };
            fetch_aux(0, xa[0]);
            float2 w[NT];
            float sv[NT], g0[NT], g1[NT];
#pragma unroll
            for (int hp = 0; hp < 8; ++hp) {
                const int q = hp >> 1, rh = hp & 1;
                const long tile = rgl * 4 + q;
                if (rh == 0) {
                    const int b = (int)(tile / a.Timg);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        w[t] = *reinterpret_cast<const float2*>(a.tab + ((long)b * Hp + nb * NB + t * 32 + nl) * kSlots);
                        sv[t] = 0.0f; g0[t] = 0.0f; g1[t] = 0.0f;
                    }
                }
                if (hp + 1 < 8) fetch_aux(hp + 1, xa[(hp + 1) & 1]);
                const float* rx = rowx + 32 * q + 16 * h + 8 * rh;  // this lane's 8 rows of the group
                float4 x0v[2], x1v[2], dq[2];
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    x0v[r2] = *reinterpret_cast<const float4*>(rx + 4 * r2);
                    x1v[r2] = *reinterpret_cast<const float4*>(rx + 128 + 4 * r2);
                    if (LASTD != 0) dq[r2] = *reinterpret_cast<const float4*>(rx + 256 + 4 * r2);
                }
                float pd[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) pd[j] = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const float4 v = value(q, 2 * rh + r2, t, xa[hp & 1][t][r2], dq[r2]);
                        const float vv[4] = {v.x, v.y, v.z, v.w};
                        const float xx0[4] = {x0v[r2].x, x0v[r2].y, x0v[r2].z, x0v[r2].w};
                        const float xx1[4] = {x1v[r2].x, x1v[r2].y, x1v[r2].z, x1v[r2].w};
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            const int j = 4 * r2 + cc;
                            sv[t] += vv[cc];
                            g0[t] += vv[cc] * xx0[cc];
                            g1[t] += vv[cc] * xx1[cc];
                            pd[j] += vv[cc] * w[t].x;
                            pd[8 + j] += vv[cc] * w[t].y;
                        }
                    }
                // d(coords) of the 8 rows: this block's NB features = over the tiles (done) and the 32 lanes of the half-wave;
                // lane class cls ends up with values 4 cls .. 4 cls + 3: component cls / 2 of rows 4 (cls % 2) .. + 3
                float s4[4];
                half_reduce16(pd, s4);
                if (nl < 4) {
                    const int cls = ((nl & 1) << 1) | (nl >> 1);
                    float* dst = a.dfpart + ((long)nb * a.Mp + tile * 32 + 16 * h + 8 * rh + 4 * (cls & 1)) * 2 + (cls >> 1);
                    dst[0] = s4[0]; dst[2] = s4[1]; dst[4] = s4[2]; dst[6] = s4[3];
                }
                if (rh == 1) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        // the two half-waves hold the two row halves of the same (tile, feature): one sum
                        const float G0 = g0[t] + __shfl_xor(g0[t], 32), G1 = g1[t] + __shfl_xor(g1[t], 32);
                        const float S = sv[t] + __shfl_xor(sv[t], 32);
                        const int k = nb * NB + t * 32 + nl;
                        if (h == 0) *reinterpret_cast<float4*>(a.sgtile + (tile * (long)Hp + k) * 4) = make_float4(G0, G1, S, 0.0f);
                    }
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

template <int NT, bool DGRAD, bool FIRST, int LASTD, int CF>
__global__ __launch_bounds__(256, Dense4Occ<NT>::value) void dense4_kernel(DenseArgs a, long groups, long set0) {
    dense4_body<NT, DGRAD, FIRST, LASTD, CF>(a, groups, set0, blockIdx.x);
}

// Both block widths in ONE launch: workgroups [0, grid_main) run 64-column blocks over the sets [0, sets_main), the rest
// 32-column blocks over the sets from sets_main on -- the half-width units of a layer's last partial round (launch_dense in
// api.hip) then need no launch of their own: no kernel boundary, and they backfill the CUs as the wide workgroups drain.
// grid_main is a multiple of 8, so a workgroup's XCD label (id mod 8) is the same in both numberings.  Registers and LDS are
// the wide form's (two workgroups per CU for either width); sets_main = all sets and no further workgroups is the plain wide
// launch.
template <bool DGRAD, bool FIRST, int LASTD, int CF>
__global__ __launch_bounds__(256, 2) void dense4_dual_kernel(DenseArgs a, long groups, long sets_main, unsigned grid_main) {
    if (blockIdx.x < grid_main) {
        const long gmain = sets_main * 4 < groups ? sets_main * 4 : groups;
        dense4_body<2, DGRAD, FIRST, LASTD, CF>(a, gmain, 0, blockIdx.x);
    } else {
        dense4_body<1, DGRAD, FIRST, LASTD, CF>(a, groups, sets_main, blockIdx.x - grid_main);
    }
}

}  // namespace svae
