// How do fp32 MFMAs and VALU instructions share a gfx950 SIMD?  Two probes, no memory traffic:
//   A  interleave: every group of 4 independent v_mfma_f32_32x32x2_f32 is followed by K VALU fmas (the shape of
//      dense_kernel's main loop: K = 1 in the rank-1 data gradient), at 1 / 2 / 3 waves per SIMD
//   B  phases: every wave alternates P MFMAs with Q VALU fmas (main loop / epilogue of a tile), co-resident waves
//      staggered by a fraction of a period -- do one wave's VALU phases run under the other waves' MFMA phases?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
__global__ __launch_bounds__(256) void probe_a(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = a + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < K; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 0.999f, 0.001f);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// K VALU whose result feeds the NEXT group's MFMA A operand (the rank-1 form's act'(a))
template <int K>
__global__ __launch_bounds__(256) void probe_a_dep(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        float x = a;
#pragma unroll
        for (int j = 0; j < K; ++j) x = __builtin_fmaf(-x, x, 1.0f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void probe_b(float* out, int periods, int P, int Q, int stagger_mfma, float a0, float b0) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = a + j;
    // stagger: workgroup w of the CU (blockIdx.x / 256 under round-robin placement) starts with w * stagger_mfma extra MFMAs
    const int lead = (blockIdx.x / 256) * stagger_mfma;
    for (int i = 0; i < lead / 4; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    for (int p = 0; p < periods; ++p) {
        for (int i = 0; i < P / 4; ++i) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
        for (int i = 0; i < Q / 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* g_out;
template <class F>
float timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

template <int K>
void run_a(int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 6000 / waves_per_simd;
    const float ms = timeit([&] { hipLaunchKernelGGL(probe_a<K>, dim3(blocks), dim3(256), 0, 0, g_out, iters, 1.0f, 0.5f); });
    const float msd = timeit([&] { hipLaunchKernelGGL(probe_a_dep<K>, dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.3f, 0.5f); });
    // cycles per group of 4 MFMAs per SIMD, assuming 2.4 GHz is NOT valid: report time per group in ns and relative
    const double groups = (double)iters * waves_per_simd;  // per SIMD
    printf("A  waves/SIMD %d  K=%2d  %7.3f ms  %6.1f ns per 4-MFMA group (indep VALU) | %7.3f ms %6.1f ns (VALU feeds the MFMA)\n",
           waves_per_simd, K, ms, ms * 1e6 / groups, msd, msd * 1e6 / groups);
}

void run_b(int waves_per_simd, int P, int Q, int stagger) {
    const int blocks = 256 * waves_per_simd, periods = 48 / waves_per_simd;
    const float ms = timeit([&] { hipLaunchKernelGGL(probe_b, dim3(blocks), dim3(256), 0, 0, g_out, periods, P, Q, stagger, 1.0f, 0.5f); });
    const float ms0 = timeit([&] { hipLaunchKernelGGL(probe_b, dim3(blocks), dim3(256), 0, 0, g_out, periods, P, 0, stagger, 1.0f, 0.5f); });
    const float msq = timeit([&] { hipLaunchKernelGGL(probe_b, dim3(blocks), dim3(256), 0, 0, g_out, periods, 0, Q, 0, 1.0f, 0.5f); });
    printf("B  waves/SIMD %d  P=%4d MFMA  Q=%4d VALU  stagger %4d : both %7.3f ms | MFMA only %7.3f | VALU only %7.3f | sum %7.3f  -> hidden %.0f %%\n",
           waves_per_simd, P, Q, stagger, ms, ms0, msq, ms0 + msq, 100.0 * (ms0 + msq - ms) / (msq > 0 ? msq : 1));
}

int main() {
    hipMalloc(&g_out, (size_t)256 * 8 * 256 * 4);
    for (int w = 1; w <= 3; ++w) {
        run_a<0>(w); run_a<1>(w); run_a<2>(w); run_a<4>(w); run_a<8>(w); run_a<16>(w);
    }
    for (int w = 1; w <= 3; ++w) {
        run_b(w, 1024, 512, 0);
        run_b(w, 1024, 512, 1024 / (w > 1 ? w : 1));
        run_b(w, 1024, 1024, 1024 / (w > 1 ? w : 1));
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
