// wgrad2_kernel: the hidden layers' weight gradient with both operands loaded straight into registers, TWO waves per SIMD.
// (autograd's dW = dh^T a_prev of nn.Linear in SpatialGenerator.layers, spatial_vae/models.py:77-83.)
//
// Why a second form.  wgrad_kernel (dense.h) gives a wave 4 x 4 tiles of dW (256 accumulator registers, ONE wave per SIMD)
// and moves its operands HBM -> LDS ring -> registers (8-9 DMA pieces + 8-9 ds_read_b128 per 64 MFMAs).  r03's probes
// (tools/mfma_valu_probe.hip) show what that costs on gfx950: every vector instruction between fp32 MFMAs takes matrix-pipe
// time, and about twice as much at one wave per SIMD as at two.  The rank-1 form's transform (32 VALU instructions per 64
// MFMAs) is why the headline config's weight gradient sat at 78 % MFMA-busy where the plain form reaches 85 %.
// Here a wave owns 2 x 4 tiles (128 accumulator registers, two waves per SIMD), its six operand vectors per row octet come
// by one global_load_dwordx4 each (lane = feature, 4 consecutive rows = 4 k-steps, as before: 1 KiB contiguous per wave
// instruction) into a ring of three octets held in registers -- no LDS, no ds_read, no barrier -- and the transform's
// instructions (16 per 32 MFMAs: the same ratio) are shared between two waves' MFMA streams.
// Same partial-sum layout as wgrad_kernel (slab / bslab / wpart / bpart): wgrad_reduce_kernel and out_bwd_reduce_kernel
// do not change.  Numerics: per element of dW the same m-ordered fma chain per split as wgrad_kernel (k-steps in row order
// within an octet, octets in order, splits summed in order by the reduce): bit-identical to it.
#pragma once
#include "dense4.h"

namespace svae {

// 16 bytes per lane from (wave-uniform base in SGPRs) + (32-bit per-lane byte offset); as for every asm load here the
// destination is an in/out operand so that the compiler keeps the register allocated across the load's flight, and the base
// must not come from a VALU instruction (v_readfirstlane) within 5 instructions (tools/check_asm_loads.py checks both)
__device__ __forceinline__ void load_s4(const void* sbase, unsigned voff, f32x4v& v) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(v) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_slot6(f32x4v& a0, f32x4v& a1, f32x4v& b0, f32x4v& b1, f32x4v& b2, f32x4v& b3) {
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_slot7(f32x4v& a0, f32x4v& a1, f32x4v& b0, f32x4v& b1, f32x4v& b2, f32x4v& b3, f32x4v& d) {
    asm volatile("s_waitcnt vmcnt(%7)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(d) : "i"(N) : "memory");
}

// R1 = 0: dh comes from HBM (a.dh).  R1 = 1 (tanh) / 2 (sigmoid): the rank-1 LASTW form of wgrad_kernel<1, R1> -- a.dh is
// a_{L-1}, the fragments take act'(a) do[m], w_o[n] is applied by the reduce, and the waves of column block 0 also accumulate
// dW_o / db_o partials.
// A workgroup owns a 128 x 256 block of dW: wave w -> dh tiles 2 (w >> 1) .., a_prev tiles 4 (w & 1) ..
template <int R1>
__global__ __launch_bounds__(256, 2) void wgrad2_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int Hp = a.Hp, ntile = Hp / 32;
    const int nbA = (ntile + 3) / 4, nbB = (ntile + 7) / 8, nb2 = nbA * nbB;
    int blk, split;
    long S;
    if (a.S > 0) {   // 1-D XCD-aware grid (S % 8 == 0): the blocks of one split share an XCD and its L2
        const int local = blockIdx.x >> 3;
        S = a.S;
        blk = local % nb2;
        split = (local / nb2) * 8 + (blockIdx.x & 7);
    } else {
        S = gridDim.y;
        blk = blockIdx.x;
        split = blockIdx.y;
    }
    const int bi = blk / nbB, bj = blk % nbB;
    const int ibase = bi * 4 + (wave >> 1) * 2, jbase = bj * 8 + (wave & 1) * 4;
    // compact row octets (all-padding octets at the end of an image are skipped), as in wgrad_kernel
    const long nco = (a.noct / a.po) * a.vo;
    const long per = (nco + S - 1) / S;
    const long o0 = split * per;
    const long o1 = (o0 + per < nco) ? o0 + per : nco;
    const long nmine = o1 > o0 ? o1 - o0 : 0;

    unsigned voa[2], vob[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int it = (ibase + t < ntile) ? ibase + t : ntile - 1;
        voa[t] = (unsigned)((((long)it * 32 + nl) * 8 + 4 * h) * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int jt = (jbase + t < ntile) ? jbase + t : ntile - 1;
        vob[t] = (unsigned)((((long)jt * 32 + nl) * 8 + 4 * h) * 4);
    }
    const unsigned vod = (unsigned)(16 * h);   // R1: do_p[8 oc + 4 h .. + 3], the rows of this lane's k-steps

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bs[2] = {0.0f, 0.0f};
    const bool own_bias = bj == 0 && (wave & 1) == 0;   // one wave per dh tile pair stores the bias partials
    const bool own_out = R1 != 0 && own_bias;
    float pw[2] = {0.0f, 0.0f}, pbias = 0.0f;

    // Ring of three octets in registers.  Queue per wave, oldest first, when octet o is about to be multiplied:
    //     loads(o)[L] loads(o+1)[L] loads(o+2)[L]   ->   s_waitcnt vmcnt(2 L),   L = 6 (+1 with the d(logits) rows)
    // and the loads of o+3 are issued behind the MFMAs of o, into the registers those MFMAs just read.
    constexpr int L = R1 != 0 ? 7 : 6;
    f32x4v ra[3][2], rb[3][4], rd[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        rd[s] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < 2; ++t) ra[s][t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < 4; ++t) rb[s][t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const long ostride = (long)Hp * 8;
    long nxt_j = o0, nxt_oct = 0;   // next compact octet to fetch and its position in the row space (wave-uniform)
    int nxt_r = 0;
    if (nmine > 0) {
        const long img = o0 / a.vo;
        nxt_r = (int)(o0 - img * a.vo);
        nxt_oct = img * a.po + nxt_r;
    }
    auto issue = [&](auto slot) {   // beyond the range the last octet is fetched again (never multiplied)
        constexpr int s = decltype(slot)::value;
        const long oc = nxt_oct;
        ++nxt_j;
        {   // scalar selects, no branch: the loop body stays one basic block
            const bool more = nxt_j < o1, wrap = nxt_r + 1 == a.vo;
            nxt_oct += more ? (wrap ? 1 + a.po - a.vo : 1) : 0;
            nxt_r = more ? (wrap ? 0 : nxt_r + 1) : nxt_r;
        }
        const float* ba = a.dh + oc * ostride;
        const float* bb = a.aprev + oc * ostride;
        load_s4(ba, voa[0], ra[s][0]);
        load_s4(ba, voa[1], ra[s][1]);
        load_s4(bb, vob[0], rb[s][0]);
        load_s4(bb, vob[1], rb[s][1]);
        load_s4(bb, vob[2], rb[s][2]);
        load_s4(bb, vob[3], rb[s][3]);
        if (R1 != 0) load_s4(a.do_p + oc * 8, vod, rd[s]);
    };
    // own: this wave also keeps the bias (and dW_o / db_o) partials -- a wave-uniform branch around pure VALU work.  (A compile-time
    // copy of the loop per role was tried: it spills, 384 bytes, and hipcc then re-homes in-flight ring registers.)
    auto multiply = [&](auto slot, bool own) {
        constexpr int s = decltype(slot)::value;
        const bool OWN = own;
        f32x4v x[2];
        if (R1 != 0) {
            const f32x4v d = rd[s];
            if (OWN) {   // dW_o / db_o partials from the raw a_{L-1} fragments
                pbias += (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    pw[t] += (d[0] * ra[s][t][0] + d[1] * ra[s][t][1]) + (d[2] * ra[s][t][2] + d[3] * ra[s][t][3]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {   // act'(a) d: t = d a, then tanh d - t a = d (1 - a^2), sigmoid t - t a = d a (1 - a)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ra[s][t][e], tt = d[e] * v;
                    x[t][e] = __builtin_fmaf(-tt, v, R1 == 1 ? d[e] : tt);
                }
            }
        } else {
            x[0] = ra[s][0];
            x[1] = ra[s][1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i][e], rb[s][j][e], acc[i][j], 0, 0, 0);
        if (OWN) {   // column sums of dh
#pragma unroll
            for (int t = 0; t < 2; ++t) bs[t] += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto wait = [&](auto slot, auto n) {
        constexpr int s = decltype(slot)::value;
        constexpr int N = decltype(n)::value;
        if (R1 != 0) wait_slot7<N>(ra[s][0], ra[s][1], rb[s][0], rb[s][1], rb[s][2], rb[s][3], rd[s]);
        else wait_slot6<N>(ra[s][0], ra[s][1], rb[s][0], rb[s][1], rb[s][2], rb[s][3]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using W2L = std::integral_constant<int, 2 * L>;
    using W0 = std::integral_constant<int, 0>;
    auto run = [&](bool own) {
        issue(S0());
        issue(S1());
        issue(S2());
        const long n3 = nmine / 3;
        for (long it = 0; it < n3; ++it) {
            wait(S0(), W2L()); multiply(S0(), own); issue(S0());
            wait(S1(), W2L()); multiply(S1(), own); issue(S1());
            wait(S2(), W2L()); multiply(S2(), own); issue(S2());
        }
        // the ring now holds the octets 3 n3 .. 3 n3 + 2 (clamped to the last): drain once, multiply the 0-2 that are left
        wait(S0(), W0()); wait(S1(), W0()); wait(S2(), W0());
        const int rem = (int)(nmine - 3 * n3);
        if (rem > 0) multiply(S0(), own);
        if (rem > 1) multiply(S1(), own);
    };
    if (nmine > 0) {
        run(own_bias);
    }

    float* slab = a.slab + (long)split * Hp * Hp;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
        for (int i = 0; i < 2; ++i)
            if (ibase + i < ntile) a.wpart[((long)split * 2 + h) * Hp + (ibase + i) * 32 + nl] = pw[i];
        if (bi == 0 && wave == 0 && nl == 0) a.bpart[(long)split * 2 + h] = pbias;
    }
    if (own_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (ibase + i < ntile) a.bslab[((long)split * 2 + h) * Hp + (ibase + i) * 32 + nl] = bs[i];
    }
}

}  // namespace svae
