// As phases.hip, but the VALU phase is the real thing: GELU (packed Horner, v_exp) + hi/lo fp16 split of a
// 128 x 32 tile (tile_gelu / split from codlad_amd/csrc/common.h), the matrix phase 96 MFMAs on those
// fragments.  KIND 0: plain v_fma_f32 filler of equal instruction count; 1: GELU + split.
#include "../../codlad_amd/csrc/common.h"
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(int iters, float *out) {
    const float a = threadIdx.x * 1e-3f + 0.5f;
    Tile t;
    for (int bo = 0; bo < 4; ++bo)
        for (int r = 0; r < 16; ++r) t.b[bo][r] = a + 0.01f * (bo * 16 + r);
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    SplitFrag f[8];
    for (int it = 0; it < iters; ++it) {
        if (KIND == 1) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int p = 0; p < 4; ++p) split_pair<true>(f[ks], t, ks, p);
        } else {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int q = 0; q < 14; ++q) t.b[bo][r] = __builtin_fmaf(t.b[bo][r], 0.999f, a);
                }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) { f[ks].hi[j] = (_Float16)t.b[ks >> 1][(ks & 1) * 8 + j]; f[ks].lo[j] = f[ks].hi[j]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            mfma_f16<3>(c[g & 3], f[g >> 2].hi, f[g >> 2].lo, f[g >> 2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // feed the result back so that nothing is loop-invariant
#pragma unroll
        for (int bo = 0; bo < 4; ++bo)
#pragma unroll
            for (int r = 0; r < 16; ++r) t.b[bo][r] = c[bo][r] * 1e-9f + a;
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = 0.f;
    for (int bo = 0; bo < 4; ++bo) r += t.b[bo][3] + c[bo][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
float run(int threads, float *out) {
    const int iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND><<<256, threads>>>(20, out);
    (void)hipEventRecord(e0);
    k<KIND><<<256, threads>>>(iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    printf("us per {VALU phase + 96 MFMAs} per wave:   1 wave/SIMD | 2 waves/SIMD (both units)\n");
    printf("  v_fma filler (14 per element)        %6.2f | %6.2f\n", run<0>(256, out), run<0>(512, out));
    printf("  GELU + split (common.h)              %6.2f | %6.2f\n", run<1>(256, out), run<1>(512, out));
    return 0;
}
