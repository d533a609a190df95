// Bare fp32 MFMA issue-rate check: what does v_mfma_f32_32x32x2_f32 sustain on THIS device with
// 1 or 2 waves per SIMD and no memory traffic?   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, const char* what) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("%-40s %8.3f ms  %7.1f TFLOP/s\n", what, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<8>(256, 4000, "256 WG x 4 waves (1/SIMD), 8 acc");
    run<8>(512, 4000, "512 WG x 4 waves (2/SIMD), 8 acc");
    run<4>(512, 8000, "512 WG (2/SIMD), 4 acc");
    run<8>(1600, 1024, "1600 WG (6.25 rounds at 1/CU... 2/CU), 8 acc");
    run<16>(256, 2000, "256 WG (1/SIMD), 16 acc");
    return 0;
}
