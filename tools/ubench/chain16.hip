// v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16 for THE contraction of the edge kernels (128 x 128 block, 32 columns
// per wave, f16x3: three matrix instructions per fragment pair, GELU + hi / lo split of the 64 input elements per lane on the
// way in, weight fragments from LDS two groups ahead), on RANDOM operands, two waves per SIMD on every CU - the regime in which
// the chip's clock, not the cycle count, decides.  MI355X_MICROARCH.md reports the 16x16x32 shape at
// 1.12-1.15 x the FLOP/s of 32x32x16 in bare loops at the power limit; is that so with this kernel's vector work beside it?
//   32: common.h gemm128_h_lds<3, true> (the production primitive): 32 groups x {2 ds_read_b128, 3 MFMA 32x32x16, 1 pair GELU + split}
//   16: the same work as 32 groups x {2 ds_read_b128, 6 MFMA 16x16x32 (two 16-column halves share a weight fragment), 1 pair}
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-honor-nans -I codlad_amd/csrc tools/ubench/chain16.hip -o tools/ubench/chain16
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>

struct Tile16 {
    f32x4 a[8][2];      // [row block of 16 features][column half]
};

DEV void mfma16_f16x3(f32x4 &acc, f16x8 whi, f16x8 wlo, const SplitFrag &x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, x.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, x.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, x.hi, acc, 0, 0, 0);
}

// the eight inputs of k-step s, column half cg: registers of row blocks 2 s and 2 s + 1 (an accumulator tile chains into the
// next layer's B operand exactly as in the 32 x 32 layout)
DEV void split_pair16(SplitFrag &f, const Tile16 &in, int s, int cg, int p, const GeluK &gk) {
    const f32x4 &v = in.a[2 * s + (p >> 1)][cg];
    f32x2 t[1] = {f32x2{v[2 * (p & 1)], v[2 * (p & 1) + 1]}};
    gelu_pairs<1>(t, gk);
    const f16x2 hh = __builtin_convertvector(t[0], f16x2);
    const f16x2 ll = split_lo_pair(hh, t[0]);
    f.hi[2 * p] = hh.x; f.hi[2 * p + 1] = hh.y;
    f.lo[2 * p] = ll.x; f.lo[2 * p + 1] = ll.y;
}

DEV void gemm128_h16_lds(Tile16 &acc, const Tile16 &in, const u32x4 *wl, int lane, const GeluK &gk) {
    const u32x4 *w = wl + lane;
    u32x4 ring[3][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        ring[g][0] = w[(g * 2 + 0) * 64];
        ring[g][1] = w[(g * 2 + 1) * 64];
    }
    SplitFrag x[2], xn[2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int p = 0; p < 4; ++p) split_pair16(x[cg], in, 0, cg, p, gk);
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        const int s = g >> 3, rb = g & 7;
        if (g + 2 < 32) {
            ring[(g + 2) % 3][0] = w[((g + 2) * 2 + 0) * 64];
            ring[(g + 2) % 3][1] = w[((g + 2) * 2 + 1) * 64];
        }
        if (s + 1 < 4) split_pair16(xn[rb >> 2], in, s + 1, rb >> 2, rb & 3, gk);      // one pair per group, as in the 32 x 32 chain
        mfma16_f16x3(acc.a[rb][0], as_f16x8(ring[g % 3][0]), as_f16x8(ring[g % 3][1]), x[0]);
        mfma16_f16x3(acc.a[rb][1], as_f16x8(ring[g % 3][0]), as_f16x8(ring[g % 3][1]), x[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (rb == 7) { x[0] = xn[0]; x[1] = xn[1]; }
    }
}

__device__ inline float rnd(unsigned &s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) * (1.0f / 32768.0f) - 1.0f;
}

// energy shares of the production contraction (SHAPE 32 only): MODE 1 = the matrix instructions and their LDS fragment reads alone
// (operand fragments split once, outside the loop), MODE 2 = the GELU + split vector stream alone (fragments folded into a sum)
template <int MODE>
DEV void part32(Tile &acc, const Tile &in, const u32x4 *wl, int lane, const GeluK &gk, SplitFrag (&pre)[8]) {
    const u32x4 *w = wl + lane;
    if (MODE == 1) {
        u32x4 ring[3][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) { ring[g][0] = w[(g * 2 + 0) * 64]; ring[g][1] = w[(g * 2 + 1) * 64]; }
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (g + 2 < 32) { ring[(g + 2) % 3][0] = w[((g + 2) * 2 + 0) * 64]; ring[(g + 2) % 3][1] = w[((g + 2) * 2 + 1) * 64]; }
            mfma_f16<3>(acc.b[g & 3], as_f16x8(ring[g % 3][0]), as_f16x8(ring[g % 3][1]), pre[g >> 2]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            SplitFrag x;
#pragma unroll
            for (int p = 0; p < 4; ++p) split_pair<true>(x, in, ks, p, gk);
            const u32x4 a = __builtin_bit_cast(u32x4, x.hi), b = __builtin_bit_cast(u32x4, x.lo);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc.b[ks & 3][4 * (ks >> 2) + i] += __builtin_bit_cast(float, (a[i] & 0x007fffffu) | 0x3f800000u) * 1e-3f
                                                                          + __builtin_bit_cast(float, (b[i] & 0x007fffffu) | 0x3f800000u) * 1e-3f;
        }
    }
}

template <int SHAPE, int MODE = 0>
__global__ __launch_bounds__(512, 2) void k(int iters, const u32x4 *wg, float *out, long long *cyc) {
    extern __shared__ __align__(16) u32x4 wl[];
    for (int i = threadIdx.x; i < 4096; i += 512) wl[i] = wg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const GeluK gk = gelu_consts(0);
    unsigned seed = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    float r = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (SHAPE == 32) {
        Tile in, acc;
        for (int b = 0; b < 4; ++b)
            for (int q = 0; q < 16; ++q) { in.b[b][q] = 1.5f * rnd(seed); acc.b[b][q] = 0.f; }
        SplitFrag pre[8];
        for (int ks = 0; ks < 8; ++ks)
            for (int p = 0; p < 4; ++p) split_pair<true>(pre[ks], in, ks, p, gk);
        for (int it = 0; it < iters; ++it) {
            if (MODE == 0) gemm128_h_lds<3, true>(acc, in, wl, lane, gk);
            else part32<MODE>(acc, in, wl, lane, gk, pre);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) { in.b[b][q] = 1e-4f * acc.b[b][q] + in.b[(b + 1) & 3][(q + 5) & 15]; acc.b[b][q] = 0.f; }
        }
        for (int b = 0; b < 4; ++b)
            for (int q = 0; q < 16; ++q) r += in.b[b][q];
    } else {
        Tile16 in, acc;
        for (int b = 0; b < 8; ++b)
            for (int c = 0; c < 2; ++c)
                for (int q = 0; q < 4; ++q) { in.a[b][c][q] = 1.5f * rnd(seed); acc.a[b][c][q] = 0.f; }
        for (int it = 0; it < iters; ++it) {
            gemm128_h16_lds(acc, in, wl, lane, gk);
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { in.a[b][c][q] = 1e-4f * acc.a[b][c][q] + in.a[(b + 1) & 7][c][(q + 1) & 3]; acc.a[b][c][q] = 0.f; }
        }
        for (int b = 0; b < 8; ++b)
            for (int c = 0; c < 2; ++c)
                for (int q = 0; q < 4; ++q) r += in.a[b][c][q];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int MODE = 0>
void run(const u32x4 *wg, float *out, long long *cyc) {
    const int iters = 2000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<SHAPE, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<SHAPE, MODE><<<256, 512, 65536>>>(200, wg, out, cyc);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<SHAPE, MODE><<<256, 512, 65536>>>(iters, wg, out, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256.0 * iters;
    float probe;
    (void)hipMemcpy(&probe, out, 4, hipMemcpyDeviceToHost);
    printf("%2dx%2d %s 2 waves/SIMD: %7.0f shader cycles per contraction per wave, %6.3f us per contraction per wave, clock %.2f GHz  (check %g)\n",
           SHAPE, SHAPE, MODE == 0 ? "           " : (MODE == 1 ? "matrix only" : "vector only"), c, ms * 1e3 / iters, c / (ms * 1e3 / iters) / 1e3, probe);
}

int main() {
    u32x4 *wg; float *out; long long *cyc;
    (void)hipMalloc(&wg, 65536);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    // random weights: fp16 "hi" halves ~ N(0, 0.09), "lo" halves 2^-11 of that scale (64 KB = 32 768 halves, alternating 1 KB fragments)
    _Float16 *hw = (_Float16 *)malloc(65536);
    srand(5);
    for (int f = 0; f < 64; ++f)
        for (int i = 0; i < 512; ++i) {
            float u = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
            hw[f * 512 + i] = (_Float16)((f & 1) ? u * 4.8828125e-4f : u);
        }
    (void)hipMemcpy(wg, hw, 65536, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        run<32>(wg, out, cyc);
        run<16>(wg, out, cyc);
        run<32, 1>(wg, out, cyc);
        run<32, 2>(wg, out, cyc);
    }
    return 0;
}
