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
// dense_kernel: OUT(Mp x Hp) = epilogue( IN(Mp x Hp) * Wp ), one 32-row tile x (NT*32)-column block per wave.
//
// MFMA roles (32x32x2: A lane l -> A[i=l&31][k=l>>5], B lane l -> B[k=l>>5][j=l&31]):
//   A = row operand: lane (m = l&31, h = l>>5) supplies IN[m][8g + 4h + e] at step (g, e): one global
//       dword per step, fetched exactly one chunk (16 steps) before its use by an asm load that is
//       re-issued right behind the MFMA group that consumed the register's previous value.
//   B = packed weights from LDS: lane (n = l&31, h) reads the 16 bytes W[n][8g + 4h .. +3] with
//       one ds_read_b128 per 4 steps and column tile; slabs are stored [h][n][4] so that lane l
//       reads bytes 16*l .. 16*l+15 of a 1 KiB slab: lane-linear, bank-conflict-free.  The fragments of
//       octet g+1 are read into registers before the MFMAs of octet g are issued.
//   D: lane (n, h) holds rows m = 8q + 4h + r (q = reg>>2, r = reg&3) of column n, i.e. four
//       consecutive rows per register quad = one 16-byte octet-major store.
// A workgroup is 4 waves = 4 consecutive row tiles of ONE column block (blockIdx.y) sharing the weight
// chunks (4 contraction octets), which stream L2 -> LDS by LDS-DMA issued from inline asm, double-buffered,
// one barrier per chunk.  NT = 4 (128 columns, 64 accumulator registers, 3-4 waves per SIMD) is the default:
// it was fastest once the memory instructions were spread (see the loop), and its work quantum divides
// BASELINE cfg 2 evenly over the chip (25 passes per SIMD).
// ------------------------------------------------------------------------------------------
struct DenseArgs {
    const float* in;    // row operand, octet-major (Mp x Hp)
    const float* wp;    // packed weights, 1 KiB slabs [Hp/8][Hp/32] of [k-quad 2][column 32][k 4] (prepare_kernel)
    float* out;         // octet-major (Mp x Hp)
    const float* bias;  // forward: (H) bias; data gradient: unused
    const float* aux;   // data gradient: a_{l-1} octet-major (its act' multiplies the result)
    long tiles;         // Mp/32
    int Hp;
    int H;
    int act;
    int resid;
    // FIRST (data gradient into the coordinate layer, in_dim == 2): instead of storing dh0 the epilogue
    // reduces it on the spot -- over the tile's rows into (G0, G1, S) per feature, over this block's
    // features into d(coords) per row (dW_c, db_c, dW_z, dz, dtheta, ddx all derive from these)
    PoseArgs pose;
    const float4* posebuf;  // (B) cos, sin, dx0, dx1
    const float* tab;       // (B, Hp, 8) effective first-layer weights, slots 0,1 used here
    float* sgtile;          // [tiles][Hp][4] = (G0, G1, S, -), both row halves of the tile summed
    float* dfpart;          // [Hp/NB][Mp][2]
    long Mp;
    int N, Timg;
    // LASTD (data gradient out of the LAST hidden layer, no residual): `in` is a_{L-1} itself and the row
    // operand dh_{L-1}[m][n] = (sum_c do[m][c] W_o[c][n]) * act'(a_{L-1}[m][n]) is formed in registers.
    //   LASTD == 1: any C, exactly that expression per element (LDS table of W_o; ~10 VALU/LDS ops per element: +21 %).
    //   LASTD == 2 (tanh) / 3 (sigmoid): ONE output channel -- dh_{L-1} = diag(do) P diag(w_o) with P = act'(a_{L-1}), a
    //               rank-1 scaling of P, so the GEMM runs on P itself (one fma per element) against weights whose rows
    //               were scaled by w_o when they were packed, and the epilogue multiplies each row by do[m].
    const float* do_p;   // [C][Mp] d(loss)/d(logits), zero on pad rows
    const float* out_w;  // (C, H)
    int C;
    // CF (forward of the LAST hidden layer, any C <= SVAE_MAX_OUT): the epilogue also contracts its activations with W_o over
    // this workgroup's columns -- the output layer's logits, one partial per column block, so a_{L-1} is not re-read
    float* lpart;        // [Hp/NB][C][Mp]
};

template <int ACT, bool DGRAD>
__device__ __forceinline__ float4 dense_epilogue(float4 v, float bias, float4 aux) {
    if (DGRAD) {
        v.x *= act_grad<ACT>(aux.x); v.y *= act_grad<ACT>(aux.y);
        v.z *= act_grad<ACT>(aux.z); v.w *= act_grad<ACT>(aux.w);
    } else {
        v.x = act_fwd<ACT>(v.x + bias); v.y = act_fwd<ACT>(v.y + bias);
        v.z = act_fwd<ACT>(v.z + bias); v.w = act_fwd<ACT>(v.w + bias);
    }
    return v;
}

// One 1 KiB LDS-DMA piece: 64 lanes x 16 bytes, global (per-lane address) -> LDS (wave-uniform base in
// M0 + 16*lane).  Issued through inline asm on purpose: for the builtin, hipcc assumes the DMA may alias
// any later ds_read and puts s_waitcnt vmcnt(0) in front of it, which serialises the whole L2->LDS
// latency into every chunk.  Hidden from the compiler, the pieces only ever make its own vmcnt waits
// conservative; their completion is awaited by dma_wait_all() before the barrier that publishes them.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_byte_addr)
        : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The same piece with a wave-uniform base pointer in SGPRs and a 32-bit per-lane byte offset: no 64-bit VALU address
// arithmetic per piece.  (An SGPR operand of inline asm is invisible to hipcc's hazard recognizer: the base must not be
// written by a VALU instruction -- v_readfirstlane -- within 5 instructions of the DMA; tools/check_asm_loads.py checks.)
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
        : "memory");
}

// Row operand of one 4-octet chunk: 16 dwords per lane at byte offsets 256*octet + 32*kstep from p.
// Also asm, so that EVERY vector-memory operation of the main loop is invisible to hipcc's waitcnt
// pass and the loop's two hand-counted s_waitcnt are the only ones (see the queue proof at the loop).
// hipcc regards the destinations as written the moment the statement ends: it must never get a reason
// to copy, spill or re-home them before wait_vm_chunk (a loop-carried phi copy of in-flight registers
// once sent the late data into the DMA address register: memory fault).  Hence (1) the operands are
// "+v" -- the load lands in the registers the variable already lives in -- and (2) load and wait sit in
// the SAME loop iteration, so no in-flight value crosses a back-edge.  tools/check_asm_loads.py
// verifies on the ISA that nothing touches a destination between its load and its wait.
__device__ __forceinline__ void load_a_chunk(const float* p, float (&v)[4][4]) {
    asm volatile(
        "global_load_dword %0, %16, off\n\t"
        "global_load_dword %1, %16, off offset:32\n\t"
        "global_load_dword %2, %16, off offset:64\n\t"
        "global_load_dword %3, %16, off offset:96\n\t"
        "global_load_dword %4, %16, off offset:256\n\t"
        "global_load_dword %5, %16, off offset:288\n\t"
        "global_load_dword %6, %16, off offset:320\n\t"
        "global_load_dword %7, %16, off offset:352\n\t"
        "global_load_dword %8, %16, off offset:512\n\t"
        "global_load_dword %9, %16, off offset:544\n\t"
        "global_load_dword %10, %16, off offset:576\n\t"
        "global_load_dword %11, %16, off offset:608\n\t"
        "global_load_dword %12, %16, off offset:768\n\t"
        "global_load_dword %13, %16, off offset:800\n\t"
        "global_load_dword %14, %16, off offset:832\n\t"
        "global_load_dword %15, %16, off offset:864"
        : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[1][0]), "+v"(v[1][1]),
          "+v"(v[1][2]), "+v"(v[1][3]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[2][2]), "+v"(v[2][3]),
          "+v"(v[3][0]), "+v"(v[3][1]), "+v"(v[3][2]), "+v"(v[3][3])
        : "v"(p)
        : "memory");
}
// one dword of the row operand at an immediate byte offset, into the register the variable lives in
template <int OFF>
__device__ __forceinline__ void load_a1(const float* p, float& v) {
    asm volatile("global_load_dword %0, %1, off offset:%2" : "+v"(v) : "v"(p), "i"(OFF) : "memory");
}
// wait until at most N vector-memory operations are outstanding, tied to ONE register
template <int N>
__device__ __forceinline__ void wait_vm1(float& v) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "i"(N) : "memory");
}

// wait until at most N vector-memory operations of this wave are outstanding; the chunk registers are
// passed through so that no use of them can be scheduled above the wait
template <int N>
__device__ __forceinline__ void wait_vm_chunk(float (&v)[4][4]) {
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[1][0]), "+v"(v[1][1]),
                   "+v"(v[1][2]), "+v"(v[1][3]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[2][2]), "+v"(v[2][3]),
                   "+v"(v[3][0]), "+v"(v[3][1]), "+v"(v[3][2]), "+v"(v[3][3])
                 : "i"(N)
                 : "memory");
}

// Timing-only ablation switches for tools/dense_ablate.hip (never defined in the product build):
//   1 = no epilogue   2 = no barrier / LDS-DMA in the loop   4 = no B-fragment reads in the loop
//   8 = no A-operand loads in the loop   16 = half of the MFMAs   32 = every tile reads the rows of tile 0
//   64 = LDS-DMA issued but never awaited and no barrier   128 = A-operand loads issued but never awaited in the loop
#ifndef SVAE_ABLATE
#define SVAE_ABLATE 0
#endif

template <int NT>
struct DenseCfg {
    static constexpr int NB = NT * 32;                 // columns per accumulation block
    static constexpr int G = 4;                        // contraction octets per LDS chunk
    static constexpr int CHUNK = G * NB * 8;           // floats per LDS buffer
    static constexpr int LDS_BYTES = 2 * CHUNK * 4;    // two buffers
    static constexpr int NINSTR = G * NT;              // 1 KiB global_load_lds wave-instructions per chunk
};

#ifndef SVAE_NT4_WAVES
#define SVAE_NT4_WAVES 3
#endif
// the forward variant that also contracts with W_o (CF) keeps 168 registers in its epilogue.  Sizing it for 4 waves per
// SIMD (the loop itself needs 124; the overflow then spills in the epilogue only) was measured: 0.938 vs 0.92 ms, so 3.
#ifndef SVAE_CF_WAVES
#define SVAE_CF_WAVES 3
#endif
// waves per SIMD the register allocation is sized for: 1 at NT = 16 (256 accumulators), 2 at NT = 8, and at
// NT <= 4 three for the plain kernels (the fused FIRST / LASTD variants keep more values live: two)
template <int NT, bool FUSED>
struct DenseOcc {
    static constexpr int value = NT == 16 ? 1 : (NT == 8 || FUSED) ? 2 : SVAE_NT4_WAVES;
};

template <int NT, bool DGRAD, bool RESID, bool FIRST = false, int LASTD = 0, int CF = 0>
__global__ __launch_bounds__(256, ((CF > 0 && NT <= 4) ? SVAE_CF_WAVES : DenseOcc<NT, (FIRST || LASTD != 0)>::value)) void dense_kernel(DenseArgs a) {
    static_assert(!FIRST || DGRAD, "FIRST is a data-gradient epilogue");
    static_assert(CF == 0 || (!DGRAD && CF == 1), "CF is a forward epilogue (a flag: the channel count is a.C)");
    static_assert(LASTD == 0 || (DGRAD && !RESID), "LASTD is a data-gradient prologue without residual");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using Cfg = DenseCfg<NT>;
    constexpr int NB = Cfg::NB, G = Cfg::G, CHUNK = Cfg::CHUNK, NINSTR = Cfg::NINSTR;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int Hp = a.Hp;
    const int noct = Hp / 8;       // multiple of 4
    const int nchunk = noct / G;
    // XCD-aware 1-D grid (gridDim.y == 1): workgroup ids congruent mod 8 share an XCD and its L2; the column blocks of one
    // 128-row group get consecutive ids on ONE XCD, so the group's row operand comes from HBM once instead of once per
    // column block.  (The kernel is MFMA-bound either way; this trims HBM traffic and power.)  Legacy 2-D grid otherwise.
    long grp;
    int nb_;
    if (gridDim.y == 1) {
        const int nblk = (Hp / 32) / NT;
        const long local = blockIdx.x >> 3;
        nb_ = (int)(local % nblk);
        grp = (local / nblk) * 8 + (blockIdx.x & 7);
    } else {
        nb_ = blockIdx.y;
        grp = blockIdx.x;
    }
    const long tile = grp * 4 + wave;
    const bool live = tile < a.tiles;
    const long tl = live ? tile : a.tiles - 1;  // dead waves recompute the last tile and store nothing

    // this lane's row (m = nl) of the row operand: element (m, k) at ((m>>3)*Hp + k)*8 + (m&7)
    const float* arow = a.in + ((((SVAE_ABLATE & 32) ? 0 : tl) * 4 + (nl >> 3)) * (long)Hp + 4 * h) * 8 + (nl & 7);
    const float* bfrag = smem + lane * 4;  // this lane's 16 bytes of a 1 KiB (32-column x 8-k) slab
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const int ntile = Hp / 32;

    // one column block (NB columns) per workgroup: blockIdx.y.  Keeping the work quantum at
    // 32 rows x NB columns (not x Hp) gives the dispatcher twice as many, half as long workgroups to
    // balance over the 256 CUs x 2 resident workgroups (6.25 rounds instead of 3.1 at BASELINE cfg 2).
    {
        const int nb = nb_;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        // weight chunk c (G contraction octets x NT column tiles, 1 KiB slabs) -> LDS buffer buf.  Piece j of this wave
        // is slab idx = wave + 4 j of the chunk: its byte offset inside the chunk and its LDS address are loop-invariant
        // (prepared here), the chunk base is one scalar multiply-add on the kernel argument.
        constexpr int PW = NINSTR / 4;
        unsigned poff[PW], pm0[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int idx = wave + 4 * j;
            const int gl = idx / NT, tt = idx % NT;
            poff[j] = (unsigned)(((gl * ntile + nb * NT + tt) * 256 + lane * 4) * 4);
            pm0[j] = lds_base + (unsigned)idx * 1024u;
        }
        const unsigned chunk_bytes = (unsigned)(G * ntile) * 1024u;
        auto stage_piece = [&](int c, int buf, int j) {
            glds16_s(reinterpret_cast<const char*>(a.wp) + (long)c * chunk_bytes, poff[j], pm0[j] + (unsigned)(buf * CHUNK) * 4u);
        };
        auto stage = [&](int c, int buf) {
#pragma unroll
            for (int j = 0; j < PW; ++j) stage_piece(c, buf, j);
        };
        // B fragments of contraction octet o: NT x 16 bytes per lane (4 k-steps each), conflict-free
        auto read_b = [&](int o, float4 (&b)[NT]) {
            const float* base = bfrag + ((o / G) & 1) * CHUNK + (o % G) * NT * 256;
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const float4*>(base + t * 256);
        };
        // Main loop.  No branches (indices past the end are clamped: a harmless re-read / re-stage of valid
        // data) and no compiler-visible vector-memory operation.  The memory instructions are SPREAD: after
        // every group of NT MFMAs (one k-step over the column tiles) the wave re-issues the single A-operand
        // dword that this same group will need one chunk later, and -- in the chunk's last octet, behind the
        // barrier -- its share of the next-but-one chunk's DMA pieces.  (Clumps of 16 loads / P DMA pieces left
        // the MFMA pipe idle for hundreds of cycles per chunk.)
        // Queue proof, per wave: a slot is [wait][NT MFMAs][A load][DMA pieces, last octet only].  Between the
        // issue of an A dword and its use exactly one chunk (16 slots) later lie 15 A loads and P DMA pieces,
        // for every slot, so ONE constant serves them all: s_waitcnt vmcnt(15 + P); in the first chunk fewer
        // operations are outstanding and its operand was drained in the prologue.  Before the barrier in
        // chunk c, the last DMA piece of chunk c+1 (issued at the end of chunk c-1) has the 12 A loads of
        // slots 0..11 behind it: vmcnt(12).
        constexpr int P = NINSTR / 4;       // DMA pieces per wave and chunk
        constexpr int PE = (P + 3) / 4;     // ... per k-step of the last octet
        static_assert(G == 4, "load_a_chunk / the slot offsets assume 4 octets per chunk");
        // LASTD: W_o (zero rows for unused channels, zero-padded to Hp) into LDS behind the weight buffers, this
        // lane's row of d(logits) into registers.  These compiler-visible loads precede every asm memory op.
        float* wo_lds = smem + 2 * CHUNK;
        const ActCoef acoef = act_coef(a.act);
        float dlog[SVAE_MAX_OUT] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (LASTD == 1) {
            for (int i = threadIdx.x; i < SVAE_MAX_OUT * Hp; i += 256) {
                const int c = i / Hp, n = i - c * Hp;
                wo_lds[i] = (c < a.C && n < a.H) ? a.out_w[c * a.H + n] : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < SVAE_MAX_OUT; ++c) {
                const float v = a.do_p[(long)(c < a.C ? c : 0) * a.Mp + tl * 32 + nl];
                dlog[c] = c < a.C ? v : 0.0f;
            }
        }
        stage(0, 0);
        stage(nchunk > 1 ? 1 : 0, 1);
        float av[G][4];
        float4 b0[NT], b1[NT];
#pragma unroll
        for (int gl = 0; gl < G; ++gl)
#pragma unroll
            for (int e = 0; e < 4; ++e) av[gl][e] = 0.0f;
        load_a_chunk(arow, av);
        wait_vm_chunk<0>(av);
        __syncthreads();  // chunks 0 and 1 (and the W_o table) have landed in LDS
        read_b(0, b0);
        const int spare = nchunk & 1;  // buffer that held chunk nchunk-2: the sink of redundant re-stages
        for (int c = 0; c < nchunk; ++c) {
            const float* anext = arow + (long)(c + 1 < nchunk ? c + 1 : nchunk - 1) * (G * 64);
            const bool more = c + 2 < nchunk;
            const int cstage = more ? c + 2 : nchunk - 1, bstage = more ? (c & 1) : spare;
#pragma unroll
            for (int gl = 0; gl < G; ++gl) {
                const int o = c * G + gl;
                if (gl == G - 1 && !(SVAE_ABLATE & 2)) {
                    // the next octet opens chunk c+1: its DMA must have landed in every wave's view, and
                    // every wave must hold its last fragments of chunk c before that buffer is reused
                    wait_vm1<12>(av[gl][0]);
                    __syncthreads();
                }
                const int onext = (o + 1 < noct) ? o + 1 : noct - 1;
                if (!(SVAE_ABLATE & 4)) {
                    if (gl & 1) read_b(onext, b0); else read_b(onext, b1);
                }
                float sdo[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (LASTD == 1) {  // sum_c dlog_c * W_o[c][k] for this octet's four k-steps (k = 8o + 4h + e)
                    sdo[0] = sdo[1] = sdo[2] = sdo[3] = 0.0f;
#pragma unroll
                    for (int cc = 0; cc < SVAE_MAX_OUT; ++cc) {
                        const float4 w = *reinterpret_cast<const float4*>(wo_lds + cc * Hp + o * 8 + 4 * h);
                        sdo[0] += dlog[cc] * w.x; sdo[1] += dlog[cc] * w.y; sdo[2] += dlog[cc] * w.z; sdo[3] += dlog[cc] * w.w;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!(SVAE_ABLATE & 128)) wait_vm1<15 + P>(av[gl][e]);
                    // rank-1 forms: act'(a) in ONE fma -- tanh 1 - a^2 (LASTD == 2), sigmoid a - a^2 (LASTD == 3).  (Forming it
                    // one slot ahead, behind the previous MFMA group, was measured: no change -- the cost is the VALU
                    // instruction's share of the matrix pipe, ~13 cycles per slot, not the dependency.)
                    const float x = LASTD == 1 ? sdo[e] * act_grad_rt(acoef, av[gl][e])
                                    : LASTD == 2 ? __builtin_fmaf(-av[gl][e], av[gl][e], 1.0f)
                                    : LASTD == 3 ? __builtin_fmaf(-av[gl][e], av[gl][e], av[gl][e])
                                                 : av[gl][e];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float4 bb = (gl & 1) ? b1[t] : b0[t];
                        const float bv = e == 0 ? bb.x : e == 1 ? bb.y : e == 2 ? bb.z : bb.w;
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bv, acc[t], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // re-issue this register's load for the next chunk (byte offset 256*gl + 32*e)
                    if (!(SVAE_ABLATE & 8)) {
                        if (gl == 0 && e == 0) load_a1<0>(anext, av[0][0]);
                        if (gl == 0 && e == 1) load_a1<32>(anext, av[0][1]);
                        if (gl == 0 && e == 2) load_a1<64>(anext, av[0][2]);
                        if (gl == 0 && e == 3) load_a1<96>(anext, av[0][3]);
                        if (gl == 1 && e == 0) load_a1<256>(anext, av[1][0]);
                        if (gl == 1 && e == 1) load_a1<288>(anext, av[1][1]);
                        if (gl == 1 && e == 2) load_a1<320>(anext, av[1][2]);
                        if (gl == 1 && e == 3) load_a1<352>(anext, av[1][3]);
                        if (gl == 2 && e == 0) load_a1<512>(anext, av[2][0]);
                        if (gl == 2 && e == 1) load_a1<544>(anext, av[2][1]);
                        if (gl == 2 && e == 2) load_a1<576>(anext, av[2][2]);
                        if (gl == 2 && e == 3) load_a1<608>(anext, av[2][3]);
                        if (gl == 3 && e == 0) load_a1<768>(anext, av[3][0]);
                        if (gl == 3 && e == 1) load_a1<800>(anext, av[3][1]);
                        if (gl == 3 && e == 2) load_a1<832>(anext, av[3][2]);
                        if (gl == 3 && e == 3) load_a1<864>(anext, av[3][3]);
                    }
                    if (gl == G - 1 && !(SVAE_ABLATE & 2)) {  // this wave's DMA pieces of chunk c+2, PE per k-step
#pragma unroll
                        for (int j = 0; j < PE; ++j)
                            if (e * PE + j < P) stage_piece(cstage, bstage, e * PE + j);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        {   // drain: the last A dwords and DMA pieces are in flight and never used; keep their registers tied up
            float sink = 0.0f;
#pragma unroll
            for (int gl = 0; gl < G; ++gl)
#pragma unroll
                for (int e = 0; e < 4; ++e) { wait_vm1<0>(av[gl][e]); sink += av[gl][e]; }
            if (a.tiles < 0) a.out[0] = sink;  // never true
        }

        // ---- epilogue: bias/activation (forward) or act' of the previous layer (data gradient).
        // vmcnt counts stores as well as loads, in issue order: a load issued after a store cannot be
        // waited for without also waiting for that store.  So all loads of column tile t+1 (bias is
        // batched up front) are issued BEFORE the stores of tile t, and nothing in here branches.
        if (SVAE_ABLATE & 1) {
            if (a.tiles < 0) {  // never true: keeps the accumulators alive without an epilogue
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a.out[t * 16 + r] = acc[t][r];
            }
        } else if (live) {
            auto epi = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
                float bias[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int n = nb * NB + t * 32 + nl;
                    const float bv = DGRAD ? 0.0f : a.bias[n < a.H ? n : a.H - 1];
                    bias[t] = (n < a.H) ? bv : 0.0f;
                }
                float4 dq[4];  // rank-1 forms: d(loss)/d(logit) of this lane's rows 8q + 4h .. +3 (zero on pad rows)
                if (LASTD >= 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dq[q] = *reinterpret_cast<const float4*>(a.do_p + tl * 32 + 8 * q + 4 * h);
                }
                const long off0 = (tl * 4 * (long)Hp + nb * NB + nl) * 8 + 4 * h;  // (q = 0, t = 0)
                const long qstride = (long)Hp * 8;                                  // next row octet
                float4 xa[2][4], xr[2][4];  // aux (data gradient) and residual operands of tile t / t+1
                auto fetch = [&](int t, float4 (&fa)[4], float4 (&fr)[4]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long off = off0 + q * qstride + (long)t * 32 * 8;
                        if (DGRAD) fa[q] = *reinterpret_cast<const float4*>(a.aux + off);
                        if (RESID) fr[q] = *reinterpret_cast<const float4*>(a.in + off);
                    }
                };
                auto finish = [&](int t, const float4 (&fa)[4], const float4 (&fr)[4]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
                        if (RESID) {
                            v.x += fr[q].x; v.y += fr[q].y; v.z += fr[q].z; v.w += fr[q].w;
                        }
                        v = dense_epilogue<ACT, DGRAD>(v, bias[t], fa[q]);
                        if (LASTD >= 2) { v.x *= dq[q].x; v.y *= dq[q].y; v.z *= dq[q].z; v.w *= dq[q].w; }
                        *reinterpret_cast<float4*>(a.out + off0 + q * qstride + (long)t * 32 * 8) = v;
                        if (CF > 0) {  // the activations stay in the accumulator registers for the contraction with W_o below
                            acc[t][4 * q] = v.x; acc[t][4 * q + 1] = v.y; acc[t][4 * q + 2] = v.z; acc[t][4 * q + 3] = v.w;
                        }
                    }
                };
                if (!FIRST) {
                    if (DGRAD || RESID) fetch(0, xa[0], xr[0]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if ((DGRAD || RESID) && t + 1 < NT) fetch(t + 1, xa[(t + 1) & 1], xr[(t + 1) & 1]);
                        finish(t, xa[t & 1], xr[t & 1]);
                    }
                    if (CF > 0) {
                        // partial logits of this block's columns, one output channel at a time (a.C is wave-uniform): 16 + NT
                        // registers whatever the channel count.  Per row the products are summed over the column tiles in
                        // the order t = 0 .. NT-1, then over the 32 lanes of each half-wave; lanes nl < 4 store.
                        for (int c = 0; c < a.C; ++c) {
                            float wo[NT], lp[16];
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                const int n = nb * NB + t * 32 + nl;
                                const float wv = a.out_w[c * a.H + (n < a.H ? n : a.H - 1)];
                                wo[t] = (n < a.H) ? wv : 0.0f;
                            }
#pragma unroll
                            for (int r = 0; r < 16; ++r) lp[r] = 0.0f;
#pragma unroll
                            for (int t = 0; t < NT; ++t)
#pragma unroll
                                for (int r = 0; r < 16; ++r) lp[r] += acc[t][r] * wo[t];
                            float s4[4];
                            half_reduce16(lp, s4);
                            if (nl < 4) {  // lane class -> row quad q; rows 8q + 4h .. +3 of the tile are consecutive
                                const int q = ((nl & 1) << 1) | (nl >> 1);
                                *reinterpret_cast<float4*>(a.lpart + ((long)nb * a.C + c) * a.Mp + tl * 32 + 8 * q + 4 * h) =
                                    make_float4(s4[0], s4[1], s4[2], s4[3]);
                            }
                        }
                    }
                } else {
                    // coordinates of this lane's 16 rows (row 8q + 4h + r of the tile), wave-uniform image
                    const int b = (int)(tl / a.Timg);
                    const int i0 = (int)(tl % a.Timg) * 32 + 4 * h;
                    const float4 pb = a.posebuf[b];  // identity (1, 0, 0, 0) when the coordinates are explicit
                    const float* cbase = a.pose.coords ? a.pose.coords + (long)b * a.N * 2 : a.pose.grid;
                    float2 raw[16];
                    float x0[16], x1[16], pd0[16], pd1[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q)  // 16 independent loads, no branches: pad rows re-read row N-1
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = i0 + 8 * q + r;
                            raw[4 * q + r] = *reinterpret_cast<const float2*>(cbase + (long)(i < a.N ? i : a.N - 1) * 2);
                        }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = 4 * q + r;
                            const bool in = i0 + 8 * q + r < a.N;
                            const float2 g2 = raw[idx];
                            x0[idx] = in ? pb.x * g2.x - pb.y * g2.y + pb.z : 0.0f;
                            x1[idx] = in ? pb.y * g2.x + pb.x * g2.y + pb.w : 0.0f;
                            pd0[idx] = 0.0f; pd1[idx] = 0.0f;
                        }
                    float2 tk[2];
                    auto fetch_tab = [&](int t) {
                        const int k = nb * NB + t * 32 + nl;
                        return *reinterpret_cast<const float2*>(a.tab + ((long)b * Hp + k) * kSlots);
                    };
                    fetch(0, xa[0], xr[0]);
                    tk[0] = fetch_tab(0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (t + 1 < NT) {
                            fetch(t + 1, xa[(t + 1) & 1], xr[(t + 1) & 1]);
                            tk[(t + 1) & 1] = fetch_tab(t + 1);
                        }
                        float sv = 0.0f, g0 = 0.0f, g1 = 0.0f;
                        const float2 w = tk[t & 1];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
                            if (RESID) {
                                const float4 fr = xr[t & 1][q];
                                v.x += fr.x; v.y += fr.y; v.z += fr.z; v.w += fr.w;
                            }
                            v = dense_epilogue<ACT, true>(v, 0.0f, xa[t & 1][q]);
                            if (LASTD >= 2) { v.x *= dq[q].x; v.y *= dq[q].y; v.z *= dq[q].z; v.w *= dq[q].w; }
                            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                sv += vv[r];
                                g0 += vv[r] * x0[4 * q + r];
                                g1 += vv[r] * x1[4 * q + r];
                                pd0[4 * q + r] += vv[r] * w.x;
                                pd1[4 * q + r] += vv[r] * w.y;
                            }
                        }
                        // the two half-waves hold the two row halves of the same feature: one sum per (tile, feature)
                        g0 += __shfl_xor(g0, 32);
                        g1 += __shfl_xor(g1, 32);
                        sv += __shfl_xor(sv, 32);
                        const int k = nb * NB + t * 32 + nl;
                        if (h == 0) *reinterpret_cast<float4*>(a.sgtile + (tl * (long)Hp + k) * 4) = make_float4(g0, g1, sv, 0.0f);
                    }
                    // d(coords) of each row: sum this block's NB features = over the tiles (done) and the 32 lanes
                    {
                        float s0[4], s1[4];
                        half_reduce16(pd0, s0);
                        half_reduce16(pd1, s1);
                        if (nl < 4) {
                            const int q = ((nl & 1) << 1) | (nl >> 1);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const long m = tl * 32 + 8 * q + 4 * h + r;
                                *reinterpret_cast<float2*>(a.dfpart + ((long)nb * a.Mp + m) * 2) = make_float2(s0[r], s1[r]);
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
    // LASTW (weight gradient of the LAST hidden layer): `dh` is a_{L-1} and dh is formed from it in registers
    const float* do_p;   // [C][Mp]
    const float* out_w;  // (C, H)
    float* wpart;        // [S][2][C][Hp] partial dW_o (layout of out_bwd_kernel's wpart)
    float* bpart;        // [S][2][C] partial db_o
    long Mp;
    int H, act;
    const float* dh;     // octet-major (Mp x Hp): dh_l
    const float* aprev;  // octet-major (Mp x Hp): a_{l-1}
    float* slab;         // [S][Hp][Hp]
    float* bslab;        // [S][2][Hp] partial bias gradients (two half-waves)
    long noct;
    int vo, po;          // row octets per image that hold at least one real row (ceil(N / 8)), and per padded image (Npad / 8)
    int Hp;
    int nblk1;           // blocks per side = ceil(ntile / 8)
    int S;               // > 0: 1-D XCD-aware grid of nblk1^2 * S workgroups; 0: 2-D grid (blocks, splits)
};

constexpr int kWgradLdsBytes = 4 * 4 * 9 * 1024;  // 4 waves x 4 ring slots x (8 operand pieces + 1 d(logits) piece) KiB

// Measured (r01): on gfx950 the fp32 MFMA does not overlap with VALU work -- weaving the transform's VALU
// instructions between the MFMAs (sched_group_barrier 1:3) made this kernel SLOWER (1.03 vs 0.96 ms; 0.82 ms
// without the transform).  A per-element transform inside an fp32 GEMM loop costs MFMA-pipe time 1:1, so the
// fused form below is opt-in (SVAE_FUSE_OUT=1); the default keeps out_bwd_kernel's streaming pass.
// CL = 0: dh comes from HBM.  CL = C (1..4): LASTW, dh_{L-1} = (sum_c do_c W_o[c]) * act'(a_{L-1}) formed from the
// a_{L-1} fragments, with the rows of d(logits) arriving as a 9th DMA piece per octet; the waves of output
// column-block 0 also accumulate dW_o[c][n] = sum_m do[m][c] a_{L-1}[m][n] and db_o.
// R1 = 1 (tanh) / 2 (sigmoid), with CL == 1: the rank-1 form of LASTW -- dh_{L-1}[m][n] = do[m] act'(a[m][n]) w_o[n], so the
// fragments only take act'(a) * do[m] (2 VALU ops per element: t = d a, then d - t a or t - t a) and the factor w_o[n],
// constant along the contraction, is applied to row n of dW and to db[n] by wgrad_reduce_kernel (row_scale).
template <int CL, int R1 = 0>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradArgs a) {
    static_assert(R1 == 0 || CL == 1, "the rank-1 form needs exactly one output channel");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int Hp = a.Hp, ntile = Hp / 32;
    // grid: (blocks of dW, row-range splits), or 1-D and XCD-aware when a.S > 0 (then S % 8 == 0): the blocks of one split
    // read the same rows of both operands and get workgroup ids congruent mod 8 -- one XCD, one L2
    int blk, split;
    long S;
    if (a.S > 0) {
        const int nb2 = a.nblk1 * a.nblk1, local = blockIdx.x >> 3;
        S = a.S;
        blk = local % nb2;
        split = (local / nb2) * 8 + (blockIdx.x & 7);
    } else {
        S = gridDim.y;
        blk = blockIdx.x;
        split = blockIdx.y;
    }
    const int bi = blk / a.nblk1, bj = blk % a.nblk1;
    const int ibase = bi * 8 + (wave >> 1) * 4, jbase = bj * 8 + (wave & 1) * 4;
    // The contraction runs over the row octets that hold real rows: an image padded from N to Npad rows ends in (Npad - N) / 8
    // all-padding octets (28 x 28: 2 of 100) whose gradient rows are exactly zero, so they are not fetched or multiplied at all.
    // o counts those octets compactly; octet_of(o) = (o / vo) po + o % vo is kept incrementally by the DMA issue below.
    const long nco = (a.noct / a.po) * a.vo;
    const long per = (nco + S - 1) / S;
    const long o0 = split * per;
    const long o1 = (o0 + per < nco) ? o0 + per : nco;

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
    const bool own_bias = bj == 0 && (wave & 1) == 0;  // the waves whose bias partials are stored (see the end of the kernel)

    // Operands stream HBM/L2 -> LDS by LDS-DMA into a PRIVATE ring of this wave (4 slots x 8 pieces of
    // 1 KiB: dh tiles 0-3, a_prev tiles 0-3; lane l's 16 bytes at 16*l of a piece = conflict-free
    // ds_read_b128), three octets ahead of the MFMAs; fragments move LDS -> registers one octet ahead
    // with ordinary (compiler-visible) ds_reads.  An octet is 64 MFMAs = 4096 cycles, a loaded-chip HBM
    // round trip is about as long, hence the depth.  No asm load has a register destination here: hipcc
    // re-homes in-flight registers of such loads across the loop back-edge (see load_a_chunk), and a ring
    // cannot avoid having loads in flight there.  No barrier either: the ring belongs to one wave.
    // Queue per wave, oldest first, when the fragments of octet o+1 are about to be read:
    //     DMA(o+1)[8] DMA(o+2)[8] DMA(o+3)[8]   ->  s_waitcnt vmcnt(16).
    // The loop is branch-free: DMA and fragment reads past the end are clamped re-loads that are never
    // multiplied.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kSlotFloats = 9 * 256;  // 8 operand pieces + 1 piece for the d(logits) rows (used when CL > 0)
    constexpr int kPieces = CL > 0 ? 9 : 8;
    float* ring = smem + wave * (4 * kSlotFloats);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;
    const long ostride = (long)Hp * 8;
    // The row octet is wave-uniform, so each piece is (scalar base of the octet) + (this lane's constant 32-bit byte offset):
    // the 64-bit address of a piece costs two scalar adds instead of two VALU adds per lane (r03: 16-18 VALU instructions per
    // octet gone from a loop in which every one of them takes matrix-pipe time).
    unsigned voa[4], vob[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        voa[t] = (unsigned)((pa[t] - a.dh) * 4);
        vob[t] = (unsigned)((pb[t] - a.aprev) * 4);
    }
    const unsigned vod = (unsigned)(((lane >> 1) < 1 ? (lane & 1) : 0) * 16);  // CL == 1: lanes 0, 1 fetch do_p[8 oc + 4 hh .. +3]
    long dma_j = o0, dma_oct = 0;   // next compact octet to fetch and its position in the row space (all wave-uniform)
    int dma_r = 0;
    if (o0 < o1) {
        const long img = o0 / a.vo;
        dma_r = (int)(o0 - img * a.vo);
        dma_oct = img * a.po + dma_r;
    }
    auto dma = [&](long o) {   // called with o = o0, o0 + 1, ...: beyond the range the last octet is re-loaded (never multiplied)
        const long oc = dma_oct;
        const long off = oc * ostride;
        const unsigned slot = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)((o - o0) & 3) * (kSlotFloats * 4u));
        ++dma_j;
        if (dma_j < o1) {
            ++dma_r; ++dma_oct;
            if (dma_r == a.vo) { dma_r = 0; dma_oct += a.po - a.vo; }
        }
        const float* ba = a.dh + off;
        const float* bb = a.aprev + off;
#pragma unroll
        for (int t = 0; t < 4; ++t) glds16_s(ba, voa[t], slot + t * 1024u);
#pragma unroll
        for (int t = 0; t < 4; ++t) glds16_s(bb, vob[t], slot + (4 + t) * 1024u);
        if (CL == 1) {
            glds16_s(a.do_p + oc * 8, vod, slot + 8 * 1024u);
        } else if (CL > 1) {
            // lane (2c + hh) fetches do_p[c][8*oc + 4*hh .. +3]; the other lanes re-fetch lane 0's 16 bytes
            const int c = (lane >> 1) < CL ? (lane >> 1) : 0, hh = (lane >> 1) < CL ? (lane & 1) : 0;
            glds16(a.do_p + (long)c * a.Mp + oc * 8 + 4 * hh, slot + 8 * 1024u);
        }
    };
    // LASTW state: W_o for this lane's four dh columns, partial dW_o / db_o
    const ActCoef acoef = act_coef(a.act);
    float wo[CL > 0 ? CL : 1][4], pw[CL > 0 ? CL : 1][4], pbias[CL > 0 ? CL : 1];
    const bool own_out = (CL > 0) && bj == 0 && (wave & 1) == 0;  // one wave per dh column tile
    if (CL > 0) {
#pragma unroll
        for (int c = 0; c < CL; ++c) {
            pbias[c] = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n = (ibase + t) * 32 + nl;
                wo[c][t] = (ibase + t < ntile && n < a.H) ? a.out_w[c * a.H + n] : 0.0f;
                pw[c][t] = 0.0f;
            }
        }
    }
    auto frag = [&](long o, float4 (&xa)[4], float4 (&xb)[4]) {
        const float* sl = ring + ((o - o0) & 3) * kSlotFloats + lane * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xa[t] = *reinterpret_cast<const float4*>(sl + t * 256);
            xb[t] = *reinterpret_cast<const float4*>(sl + (4 + t) * 256);
        }
        if (CL > 0) {
            const float* dl = ring + ((o - o0) & 3) * kSlotFloats + 8 * 256 + h * 4;
            float4 d[CL > 0 ? CL : 1];
#pragma unroll
            for (int c = 0; c < CL; ++c) d[c] = *reinterpret_cast<const float4*>(dl + c * 8);
            // dW_o / db_o partials, in the waves that own an output column tile only (wave-uniform branch around pure VALU
            // work: 3 waves in 4 skip it); a clamped re-load (o >= o1) must not be accumulated twice: weight 0
            if (own_out) {
                const float wgt = (o < o1) ? 1.0f : 0.0f;
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    pbias[c] += wgt * ((d[c].x + d[c].y) + (d[c].z + d[c].w));
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        pw[c][t] += wgt * ((d[c].x * xa[t].x + d[c].y * xa[t].y) + (d[c].z * xa[t].z + d[c].w * xa[t].w));
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (R1 != 0) {  // act'(a) d: t = d a, then tanh d - t a = d (1 - a^2), sigmoid t - t a = d a (1 - a)
                    const float4 v = xa[t];
                    const float4 tt = make_float4(d[0].x * v.x, d[0].y * v.y, d[0].z * v.z, d[0].w * v.w);
                    xa[t] = make_float4(__builtin_fmaf(-tt.x, v.x, R1 == 1 ? d[0].x : tt.x), __builtin_fmaf(-tt.y, v.y, R1 == 1 ? d[0].y : tt.y),
                                        __builtin_fmaf(-tt.z, v.z, R1 == 1 ? d[0].z : tt.z), __builtin_fmaf(-tt.w, v.w, R1 == 1 ? d[0].w : tt.w));
                    continue;
                }
                float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    s4.x += d[c].x * wo[c][t]; s4.y += d[c].y * wo[c][t]; s4.z += d[c].z * wo[c][t]; s4.w += d[c].w * wo[c][t];
                }
                xa[t] = make_float4(s4.x * act_grad_rt(acoef, xa[t].x), s4.y * act_grad_rt(acoef, xa[t].y),
                                    s4.z * act_grad_rt(acoef, xa[t].z), s4.w * act_grad_rt(acoef, xa[t].w));
            }
        }
    };
    // R1: the same work in two phases, so that the LDS round trip of the next octet's fragments hides under this octet's
    // 64 MFMAs (this kernel runs ONE wave per SIMD: whatever this wave waits for, the matrix pipe waits for too) and only the
    // 32 VALU ops of the transform stand between two MFMA batches.
    auto frag_raw = [&](long o, float4 (&xa)[4], float4 (&xb)[4], float4& d) {
        const float* sl = ring + ((o - o0) & 3) * kSlotFloats + lane * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xa[t] = *reinterpret_cast<const float4*>(sl + t * 256);
            xb[t] = *reinterpret_cast<const float4*>(sl + (4 + t) * 256);
        }
        d = *reinterpret_cast<const float4*>(ring + ((o - o0) & 3) * kSlotFloats + 8 * 256 + h * 4);
    };
    auto frag_post = [&](long o, float4 (&xa)[4], const float4& d) {
        if (own_out) {  // dW_o / db_o partials from the raw a_{L-1} fragments (see frag)
            const float wgt = (o < o1) ? 1.0f : 0.0f;
            pbias[0] += wgt * ((d.x + d.y) + (d.z + d.w));
#pragma unroll
            for (int t = 0; t < 4; ++t) pw[0][t] += wgt * ((d.x * xa[t].x + d.y * xa[t].y) + (d.z * xa[t].z + d.w * xa[t].w));
        }
        // (r03: the same transform in packed form -- v_pk_mul_f32 / v_pk_fma_f32, 16 instead of 32 instructions per octet -- was
        // measured SLOWER here, 0.854 -> 0.867 ms: at one wave per SIMD a packed fp32 instruction costs the matrix pipe more than
        // the two scalar ones it replaces.  In dense4_kernel, two waves per SIMD, the packed form is the faster one.)
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // act'(a) d: t = d a, then tanh d - t a = d (1 - a^2), sigmoid t - t a = d a (1 - a)
            const float4 v = xa[t];
            const float4 tt = make_float4(d.x * v.x, d.y * v.y, d.z * v.z, d.w * v.w);
            xa[t] = make_float4(__builtin_fmaf(-tt.x, v.x, R1 == 1 ? d.x : tt.x), __builtin_fmaf(-tt.y, v.y, R1 == 1 ? d.y : tt.y),
                                __builtin_fmaf(-tt.z, v.z, R1 == 1 ? d.z : tt.z), __builtin_fmaf(-tt.w, v.w, R1 == 1 ? d.w : tt.w));
        }
    };
    auto mul = [&](const float4 (&xa)[4], const float4 (&xb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].x, xb[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].y, xb[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].z, xb[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].w, xb[j].w, acc[i][j], 0, 0, 0);
        if (own_bias) {  // wave-uniform: only the waves that store bslab sum the columns of dh (12 VALU instructions per octet)
#pragma unroll
            for (int t = 0; t < 4; ++t) bs[t] += (xa[t].x + xa[t].y) + (xa[t].z + xa[t].w);
        }
    };
    if (o0 < o1) {
        float4 ca[4], cb[4], na[4], nb[4];
        dma(o0);
        dma(o0 + 1);
        dma(o0 + 2);
        if (CL > 0) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        float4 nd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (R1 != 0) {
            frag_raw(o0, ca, cb, nd);
            frag_post(o0, ca, nd);
        } else {
            frag(o0, ca, cb);
        }
        for (long o = o0; o < o1; ++o) {
            dma(o + 3);
            if (CL > 0) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  // DMA(o+1) has landed (9 pieces per octet)
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (R1 != 0) frag_raw(o + 1, na, nb, nd);
            else frag(o + 1, na, nb);
            if (CL == 0 || R1 != 0) __builtin_amdgcn_sched_barrier(0);
            mul(ca, cb);
            if (CL == 0 || R1 != 0) __builtin_amdgcn_sched_barrier(0);
            if (R1 != 0) frag_post(o + 1, na, nd);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ca[t] = na[t];
                cb[t] = nb[t];
            }
        }
        dma_wait_all();  // the clamped pieces still in flight (they only ever target this wave's ring)
    }

    float* slab = a.slab + (long)split * Hp * Hp;
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
    if (own_out) {
#pragma unroll
        for (int c = 0; c < CL; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (ibase + i < ntile) a.wpart[(((long)split * 2 + h) * CL + c) * Hp + (ibase + i) * 32 + nl] = pw[c][i];
            if (bi == 0 && wave == 0 && nl == 0) a.bpart[((long)split * 2 + h) * CL + c] = pbias[c];
        }
    }
    if (own_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ibase + i < ntile) a.bslab[((long)split * 2 + h) * Hp + (ibase + i) * 32 + nl] = bs[i];
    }
}

// dW[n][k] = sum_s slab[s][n][k] (n, k < H), db[n] = sum_s sum_half bslab[s][half][n].  The S (typically 64) slabs of
// 4 Hp^2 bytes sit in L2 / MALL; what bounds this kernel is loads in flight, not bytes.  A block owns 64 consecutive
// elements of dW; thread = (element, slab group g of 4): it sums slabs g, g+4, g+8, ... in four interleaved chains, and the
// four groups are combined through LDS in a fixed order (deterministic).  16 independent loads per thread are in flight
// (r01: one thread per element walked all S slabs, 4 loads in flight: 0.042 ms for 64 MB).
// row_scale (nullable): dW row n and db[n] are multiplied by row_scale[n] (the rank-1 output-layer form: w_o[n]).
struct WgradReduceArgs {
    const float* slab;
    const float* bslab;
    float* dW;
    float* db;
    int H, Hp, S;
    const float* row_scale;
};
// one block's share (bx = block index among ceil(H^2 / 64)); `red` is 4 x 64 floats of LDS
__device__ __forceinline__ void wgrad_reduce_block(const WgradReduceArgs& a, unsigned bx, float (*red)[64]) {
    const float* __restrict__ slab = a.slab;
    const float* __restrict__ bslab = a.bslab;
    float* __restrict__ dW = a.dW;
    float* __restrict__ db = a.db;
    const float* __restrict__ row_scale = a.row_scale;
    const int H = a.H, Hp = a.Hp, S = a.S;
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long idx = (long)bx * 64 + col;
    const bool in = idx < (long)H * H;
    float part = 0.0f;
    if (in) {
        const int n = (int)(idx / H), k = (int)(idx - (long)n * H);
        const float* p = slab + (long)n * Hp + k;
        const long st = (long)Hp * Hp;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        int i = grp;
        for (; i + 12 < S; i += 16) {
            s0 += p[(long)i * st];
            s1 += p[(long)(i + 4) * st];
            s2 += p[(long)(i + 8) * st];
            s3 += p[(long)(i + 12) * st];
        }
        for (; i < S; i += 4) s0 += p[(long)i * st];
        part = (s0 + s1) + (s2 + s3);
    }
    red[grp][col] = part;
    __syncthreads();
    if (grp == 0 && in && dW) {
        const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        dW[idx] = row_scale ? t * row_scale[idx / H] : t;
    }
    if (db && bx * 64 < (unsigned)H) {  // the first ceil(H / 64) blocks also carry 64 bias entries each
        __syncthreads();
        const int n = bx * 64 + col;
        float s = 0.0f;
        if (n < H)
            for (int i = grp; i < 2 * S; i += 4) s += bslab[(long)i * Hp + n];
        red[grp][col] = s;
        __syncthreads();
        if (grp == 0 && n < H) {
            const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
            db[n] = row_scale ? t * row_scale[n] : t;
        }
    }
}
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WgradReduceArgs a) {
    __shared__ float red[4][64];
    wgrad_reduce_block(a, blockIdx.x, red);
}

}  // namespace svae
