// Would 16-column tiles (v_mfma_f32_16x16x32_f16, 32 accumulator registers per 128 x 16 tile instead of 64
// per 128 x 32) be fed fast enough from LDS?  Per group: two 16-byte weight-fragment reads per lane + 3 MFMAs,
// weights resident in LDS (128 KB), 8 waves per workgroup, one workgroup per CU - the edge kernels' shape.
//   32x32x16: 96 MFMAs per 128x128 block and 32 columns;  16x16x32: 96 MFMAs per block and 16 columns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WIDE>
__global__ __launch_bounds__(512, 2) void k(int iters, const u32x4 *g, float *out) {
    extern __shared__ __align__(16) u32x4 wl[];
    for (int i = threadIdx.x; i < 8192; i += 512) wl[i] = g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(lane * 0.01f + i);
    f32x16 c32[4] = {{0}, {0}, {0}, {0}};
    f32x4 c16[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    const u32x4 *w = wl + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int gidx = 0; gidx < 32; ++gidx) {           // one 128x128 block: 32 groups x 3 MFMAs
            const f16x8 hi = __builtin_bit_cast(f16x8, w[(gidx * 2 + 0) * 64]);
            const f16x8 lo = __builtin_bit_cast(f16x8, w[(gidx * 2 + 1) * 64]);
            if (WIDE) {
                c32[gidx & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, b, c32[gidx & 3], 0, 0, 0);
                c32[gidx & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, b, c32[gidx & 3], 0, 0, 0);
                c32[gidx & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, b, c32[gidx & 3], 0, 0, 0);
            } else {
                c16[gidx & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, b, c16[gidx & 7], 0, 0, 0);
                c16[gidx & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, b, c16[gidx & 7], 0, 0, 0);
                c16[gidx & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, b, c16[gidx & 7], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += c32[i][0];
    for (int i = 0; i < 8; ++i) r += c16[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int WIDE>
void run(const u32x4 *g, float *out) {
    const int iters = 400;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<WIDE><<<256, 512, 131072>>>(20, g, out);
    (void)hipEventRecord(e0);
    k<WIDE><<<256, 512, 131072>>>(iters, g, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_block_us = ms * 1e3 / iters;              // one 128x128 block application per wave
    const int cols = WIDE ? 32 : 16;
    printf("%s: %.2f us per 128x128 block and wave (2 waves per SIMD) = %.3f us per 32 columns and SIMD\n",
           WIDE ? "32x32x16" : "16x16x32", per_block_us, per_block_us * 2 * 32 / cols / 2);
}

int main() {
    u32x4 *g;
    float *out;
    (void)hipMalloc(&g, 131072);
    (void)hipMemset(g, 0x3c, 131072);
    (void)hipMalloc(&out, 256 * 512 * 4);
    run<1>(g, out);
    run<0>(g, out);
    return 0;
}
