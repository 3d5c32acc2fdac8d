// The real contraction primitive (csrc/common.h gemm128_h_lds) in isolation: one or two waves per SIMD, weights
// resident in LDS, no global traffic inside the timed loop.  Reports cycles per 128x128x32 contraction (96 MFMAs in
// the f16x3 mode = 3072 matrix-pipe cycles) for
//   mfma   : the contraction with the operand conversion removed (fragments prepared once)
//   plain  : operand split only (layer 1)
//   gelu   : GELU + split on the way in (layers 2, 3)
//   valu   : the GELU + split instruction stream alone (no MFMA)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-honor-nans -I codlad_amd/csrc tools/ubench/chain.hip -o tools/ubench/chain
#include "common.h"
#include <cstdio>

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void k(int iters, const u32x4 *wg, float *out, long long *cyc) {
    extern __shared__ __align__(16) u32x4 wl[];
    for (int i = threadIdx.x; i < 4096; i += NW * 64) wl[i] = wg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const GeluK gk = gelu_consts(0);
    Tile in, acc;
    for (int b = 0; b < 4; ++b)
        for (int r = 0; r < 16; ++r) { in.b[b][r] = 0.01f * (lane + r + 16 * b) - 0.7f; acc.b[b][r] = 0.f; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // MFMAs only: one fragment pair, reused
            SplitFrag x;
            for (int p = 0; p < 4; ++p) split_pair<false>(x, in, 0, p, gk);
            const u32x4 *w = wl + lane;
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                mfma_f16<3>(acc.b[g & 3], as_f16x8(w[(g * 2 + 0) * 64]), as_f16x8(w[(g * 2 + 1) * 64]), x);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 1) {
            gemm128_h_lds<3, false>(acc, in, wl, lane, gk);
        } else if (MODE == 2) {
            gemm128_h_lds<3, true>(acc, in, wl, lane, gk);
        } else {                    // the VALU stream of MODE 2 alone
            SplitFrag x;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    split_pair<true>(x, in, ks, p, gk);
                    acc.b[p][ks] += (float)x.hi[2 * p] + (float)x.lo[2 * p + 1];
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        // keep the dependency chain of a real MLP: the next contraction reads this one's output
#pragma unroll
        for (int b = 0; b < 4; ++b) in.b[b][it & 15] += acc.b[b][(it + 3) & 15] * 1e-6f;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < 16; ++i) r += acc.b[b][i] + in.b[b][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NW>
void run(const char *name, const u32x4 *wg, float *out, long long *cyc) {
    const int iters = 400;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<MODE, NW><<<256, NW * 64, 65536>>>(20, wg, out, cyc);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<MODE, NW><<<256, NW * 64, 65536>>>(iters, wg, out, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256.0 * iters;
    printf("%-6s %d wave(s)/SIMD: %8.0f shader cycles per contraction per wave, %7.3f us (clock %.2f GHz)\n", name, NW / 4, c,
           ms * 1e3 / iters, c / (ms * 1e3 / iters) / 1e3);
}

int main() {
    u32x4 *wg; float *out; long long *cyc;
    (void)hipMalloc(&wg, 65536); (void)hipMemset(wg, 0x3c, 65536);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    run<0, 4>("mfma", wg, out, cyc);  run<0, 8>("mfma", wg, out, cyc);
    run<1, 4>("plain", wg, out, cyc); run<1, 8>("plain", wg, out, cyc);
    run<2, 4>("gelu", wg, out, cyc);  run<2, 8>("gelu", wg, out, cyc);
    run<3, 4>("valu", wg, out, cyc);  run<3, 8>("valu", wg, out, cyc);
    return 0;
}
