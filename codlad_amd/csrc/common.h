// Shared device helpers for the gfx950 kernels of libcodlad_hip.so.
//
// "Chain layout": a [128 features] x [32 columns] fp32 tile held by one 64-lane wave in
// 64 VGPRs per lane, exactly as the accumulators of four v_mfma_f32_32x32x2_f32 blocks:
//   lane = 32*h + c  (c = column: an edge or a node, h = lane half)
//   t.b[bo][r]       = value[feature 32*bo + (r&3) + 8*(r>>2) + 4*h][column c]
// An accumulator tile is directly the B operand (k = feature) of the next layer's MFMAs, so
// a GELU MLP chains through registers with no LDS round trip and no cross-lane traffic:
// MFMA step (b, r) of the next layer consumes register t.b[b][r] of every lane, and the
// weight operand for that step is W[32*bo + (lane&31)][32*b + (r&3) + 8*(r>>2) + 4*(lane>>5)],
// which is how codlad_pack_block_host lays a 128x128 block out (one float4 per lane per step,
// covering the four output blocks bo).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Tile {
    f32x16 b[4];
};

#define DEV __device__ __forceinline__

DEV void tile_zero(Tile &t) {
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; ++r) t.b[bo][r] = 0.f;
}

// 128 contiguous floats (16-byte aligned) -> this lane's 64 features.
DEV void tile_load_row(Tile &t, const float *row, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = p[8 * bo + 2 * q + h];
            t.b[bo][4 * q + 0] = v.x;
            t.b[bo][4 * q + 1] = v.y;
            t.b[bo][4 * q + 2] = v.z;
            t.b[bo][4 * q + 3] = v.w;
        }
}

DEV void tile_add_row(Tile &t, const float *row, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = p[8 * bo + 2 * q + h];
            t.b[bo][4 * q + 0] += v.x;
            t.b[bo][4 * q + 1] += v.y;
            t.b[bo][4 * q + 2] += v.z;
            t.b[bo][4 * q + 3] += v.w;
        }
}

DEV void tile_store_row(const Tile &t, float *row, int h) {
    float4 *p = reinterpret_cast<float4 *>(row);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            p[8 * bo + 2 * q + h] = make_float4(t.b[bo][4 * q + 0], t.b[bo][4 * q + 1],
                                                t.b[bo][4 * q + 2], t.b[bo][4 * q + 3]);
}

struct WeightQuad {
    float x, y, z, w;
};

// One 16-byte weight fetch per lane through a buffer descriptor: the 64 KB block base lives in
// SGPRs, the lane offset is one 32-bit VGPR and the step offset a scalar, so the 64 loads of a
// block need no per-load 64-bit VGPR address (with flat addressing hipcc precomputed and spilled
// one address pair per load).  The result is bit-cast as a whole: indexing the builtin's vector
// result element-wise makes ROCm 7.2's clang emit a 4-byte load and splat it.
DEV WeightQuad weight_load(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(WeightQuad, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// acc += W(128x128, packed) @ in, for the 32 columns of this wave.  256 MFMAs.
// Weights stream from global memory (L2-resident, 64 KB per block) through a 4-deep
// register ring: one coalesced 1 KiB wave load feeds four MFMAs (256 matrix-pipe cycles).
DEV void gemm128(Tile &acc, const Tile &in, const float *__restrict__ Wpacked, int lane) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wpacked), 0, 65536, 0x00020000);
    const int voff = lane * 16;
    WeightQuad ring[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ring[s] = weight_load(rsrc, voff, s * 1024);
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const WeightQuad w = ring[s & 3];
        if (s + 4 < 64) ring[s & 3] = weight_load(rsrc, voff, (s + 4) * 1024);
        const float x = in.b[s >> 4][s & 15];
        acc.b[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, x, acc.b[0], 0, 0, 0);
        acc.b[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, x, acc.b[1], 0, 0, 0);
        acc.b[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, x, acc.b[2], 0, 0, 0);
        acc.b[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, x, acc.b[3], 0, 0, 0);
        // pin the ring order: load for step s+4, then the four MFMAs of step s
        __builtin_amdgcn_sched_barrier(0);
    }
}

// exact-erf GELU, same association as ATen's CPU kernel: (0.5*x) * (1 + erf(x / sqrt(2)))
DEV float gelu_erf(float x) { return (0.5f * x) * (1.0f + erff(x * 0.70710678118654752440f)); }

DEV void tile_gelu(Tile &t) {
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; ++r) t.b[bo][r] = gelu_erf(t.b[bo][r]);
}

// sum over the lane's own 64 features plus the partner half's 64 -> all 128 features of a column
DEV float column_sum128(float own) { return own + __shfl_xor(own, 32, 64); }

DEV float tile_own_sum(const Tile &t) {
    float s = 0.f;
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += t.b[bo][r];
    return s;
}

// LayerNorm over the 128 features of each column, no affine (two-pass moments like ATen).
DEV void tile_layernorm(Tile &t, float eps) {
    const float mean = column_sum128(tile_own_sum(t)) * (1.0f / 128.0f);
    float v = 0.f;
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = t.b[bo][r] - mean;
            t.b[bo][r] = d;
            v += d * d;
        }
    const float rstd = 1.0f / sqrtf(column_sum128(v) * (1.0f / 128.0f) + eps);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; ++r) t.b[bo][r] *= rstd;
}

// t = gate * (t * (1 + scale) + shift), per-feature vectors of 128 floats (adaLN "modulate")
DEV void tile_modulate(Tile &t, const float *shift, const float *scale, const float *gate, int h) {
    const float4 *ps = reinterpret_cast<const float4 *>(shift);
    const float4 *pc = reinterpret_cast<const float4 *>(scale);
    const float4 *pg = reinterpret_cast<const float4 *>(gate);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = 8 * bo + 2 * q + h;
            const float4 s = ps[o], c = pc[o], g = pg[o];
            t.b[bo][4 * q + 0] = g.x * (t.b[bo][4 * q + 0] * (1.0f + c.x) + s.x);
            t.b[bo][4 * q + 1] = g.y * (t.b[bo][4 * q + 1] * (1.0f + c.y) + s.y);
            t.b[bo][4 * q + 2] = g.z * (t.b[bo][4 * q + 2] * (1.0f + c.z) + s.z);
            t.b[bo][4 * q + 3] = g.w * (t.b[bo][4 * q + 3] * (1.0f + c.w) + s.w);
        }
}

// One reverse-diffusion update of a scalar (gaussian_diffusion.py:313-318, 364-367, 246-249, 446),
// every product and sum rounded separately like the reference's elementwise tensor ops.
DEV float ddpm_step(float xt, float eps, float v, const float *cf, float noise) {
#pragma clang fp contract(off)
    const float frac = (v + 1.0f) / 2.0f;
    const float logvar = frac * cf[5] + (1.0f - frac) * cf[4];
    const float x0 = cf[0] * xt - cf[1] * eps;
    const float mean = cf[2] * x0 + cf[3] * xt;
    return mean + (cf[6] * expf(0.5f * logvar)) * noise;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void codlad_set_error(const char *fmt, ...);
int codlad_check_launch(const char *what);

#define CODLAD_REQUIRE(cond, msg)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            codlad_set_error("%s: %s", __func__, msg);              \
            return -1;                                              \
        }                                                           \
    } while (0)
