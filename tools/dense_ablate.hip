// Standalone timing of the product's dense_kernel with parts compiled out (-DSVAE_ABLATE=mask, see dense.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSVAE_ABLATE=N tools/dense_ablate.hip -o tools/dense_ablate_N
#include "../spatial_vae_amd/csrc/api.hip"
#ifndef NTV
#define NTV 4
#endif
int main() {
    const int Hp = 512, H = 500;
    const long Mp = 204800, tiles = Mp / 32;
    float *in, *out, *wp, *bias;
    (void)hipMalloc(&in, Mp * Hp * 4); (void)hipMalloc(&out, Mp * Hp * 4); (void)hipMalloc(&wp, Hp * Hp * 4); (void)hipMalloc(&bias, Hp * 4);
    (void)hipMemset(in, 0, Mp * Hp * 4); (void)hipMemset(wp, 0, Hp * Hp * 4); (void)hipMemset(bias, 0, Hp * 4);
    DenseArgs a; a.in = in; a.wp = wp; a.out = out; a.bias = bias; a.aux = nullptr; a.tiles = tiles; a.Hp = Hp; a.H = H;
    a.act = SVAE_ACT_TANH; a.resid = 0;
    dim3 grid((unsigned)((tiles + 3) / 4), Hp / 32 / NTV);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_dense_nt<NTV, false>(a, grid, 0);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) launch_dense_nt<NTV, false>(a, grid, 0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("ABLATE=%2d NT=%d  %.4f ms  (%.1f TFLOP/s executed if all MFMAs ran)  %s\n", SVAE_ABLATE, NTV, ms,
           2.0 * Mp * Hp * Hp / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
