// Helpers shared by the "wide" small-job kernels (node_wide_kernels.hip, edge_wide_kernels.hip): a 32-column tile is
// owned by FOUR waves, wave bo computing output block bo (features 32 bo .. 32 bo + 31) of every contraction with its
// quarter of the weight block held in registers, and the operand tile travelling between the waves as split-fp16
// fragments in LDS.
#pragma once
#include "common.h"

constexpr int FRAG_U4 = 1024;                        // one tile as fragments: [8 k-steps][hi, lo][64 lanes] x 16 B

struct BlockQuarter {                                // the weight fragments of one output block of one 128x128 block
    u32x4 w[8][2];
    DEV void start(const void *Wpacked, int bo, int lane) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(Wpacked), 0, 65536, 0x00020000);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            w[ks][0] = weight_frag_load(rsrc, lane, (ks * 4 + bo) * 2 + 0);
            w[ks][1] = weight_frag_load(rsrc, lane, (ks * 4 + bo) * 2 + 1);
        }
    }
    // acc += W[32 bo .. 32 bo + 31][:] @ tile, the tile read as fragments from LDS
    // TRANSPOSED: operands swapped, the block arrives with lane = output feature, registers = the tile's columns
    template <int TERMS, bool TRANSPOSED = false>
    DEV void run(f32x16 &acc, const u32x4 *frag, int lane) const {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            SplitFrag x;
            x.hi = as_f16x8(frag[(ks * 2 + 0) * 64 + lane]);
            x.lo = as_f16x8(frag[(ks * 2 + 1) * 64 + lane]);
            mfma_f16<TERMS, TRANSPOSED>(acc, as_f16x8(w[ks][0]), as_f16x8(w[ks][1]), x);
        }
    }
};

// The 16 registers of output block `bo` ARE the elements of the fragments of k-steps 2 bo and 2 bo + 1 (register
// 8 s + j = element j of k-step 2 bo + s): activate (optionally), split, write both fragments.
template <bool GELU>
DEV void publish_quarter(u32x4 *frag, f32x16 q, int bo, int lane, const GeluK &gk) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        SplitFrag f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f32x2 x = {q[8 * s + 2 * p], q[8 * s + 2 * p + 1]};
            if (GELU) {
                f32x2 t[1] = {x};
                gelu_pairs<1>(t, gk);
                x = t[0];
            }
            const f16x2 hh = __builtin_convertvector(x, f16x2);
            const f16x2 ll = split_lo_pair(hh, x);
            f.hi[2 * p] = hh.x; f.hi[2 * p + 1] = hh.y;
            f.lo[2 * p] = ll.x; f.lo[2 * p + 1] = ll.y;
        }
        frag[((2 * bo + s) * 2 + 0) * 64 + lane] = __builtin_bit_cast(u32x4, f.hi);
        frag[((2 * bo + s) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, f.lo);
    }
}

DEV void quarter_load(f32x16 &a, const float *row, int bo, int h) {     // this wave's block of a 128-float vector
    const float4 *p = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = p[8 * bo + 2 * q + h];
        a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
}
DEV void quarter_store(const f32x16 &a, float *row, int bo, int h) {
    float4 *p = reinterpret_cast<float4 *>(row);
#pragma unroll
    for (int q = 0; q < 4; ++q) p[8 * bo + 2 * q + h] = make_float4(a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}
DEV f32x16 quarter_of(const Tile &t, int bo) {       // bo is wave-uniform
    return bo == 0 ? t.b[0] : (bo == 1 ? t.b[1] : (bo == 2 ? t.b[2] : t.b[3]));
}
DEV void xch_write(float4 *buf, const f32x16 &a, int bo, int h, int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) buf[(8 * bo + 2 * q + h) * 32 + c] = make_float4(a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}
DEV void xch_read(Tile &t, const float4 *buf, int h, int c) {
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = buf[(8 * bo + 2 * q + h) * 32 + c];
            t.b[bo][4 * q + 0] = v.x; t.b[bo][4 * q + 1] = v.y; t.b[bo][4 * q + 2] = v.z; t.b[bo][4 * q + 3] = v.w;
        }
}
