// Back-to-back v_mfma_f32_32x32x16_f16 accumulating into the SAME 16-register block versus
// rotating over 2 / 4 blocks: does a dependent accumulate chain stall the matrix pipe?
//   mode 0: c0 c0 c0 c0 | c1 c1 c1 c1 | ...  (what gemm_h_lds does: 4 terms of one output block)
//   mode 1: c0 c1 c0 c1 c0 c1 c0 c1 | c2 c3 ...
//   mode 2: c0 c1 c2 c3 x4
// 256 workgroups x 512 threads (2 waves per SIMD) and x 256 threads (1 wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void k(int iters, float *out) {
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 1e-3f + i); hb[i] = (_Float16)(1.f + i); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
#define M(c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c, 0, 0, 0)
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { M(c0); M(c0); M(c0); M(c0); M(c1); M(c1); M(c1); M(c1); M(c2); M(c2); M(c2); M(c2); M(c3); M(c3); M(c3); M(c3); }
        if (MODE == 1) { M(c0); M(c1); M(c0); M(c1); M(c0); M(c1); M(c0); M(c1); M(c2); M(c3); M(c2); M(c3); M(c2); M(c3); M(c2); M(c3); }
        if (MODE == 2) { M(c0); M(c1); M(c2); M(c3); M(c0); M(c1); M(c2); M(c3); M(c0); M(c1); M(c2); M(c3); M(c0); M(c1); M(c2); M(c3); }
        __builtin_amdgcn_sched_barrier(0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <int MODE>
void run(int threads, float *out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(100, out);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d, %d waves/SIMD: %.2f ns per MFMA per wave (%.2f ns per SIMD slot)\n", MODE, threads / 256,
           ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) / (threads / 256));
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    for (int threads : {256, 512}) { run<0>(threads, out); run<1>(threads, out); run<2>(threads, out); }
    return 0;
}
