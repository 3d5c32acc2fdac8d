// Helpers shared by the "wide" small-job kernels (node_wide_kernels.hip, edge_wide_kernels.hip): a 32-column tile is
// owned by FOUR waves, wave bo computing output block bo (features 32 bo .. 32 bo + 31) of every contraction with its
// quarter of the weight block held in registers, and the operand tile travelling between the waves as split-fp16
// fragments in LDS.
#pragma once
#include "common.h"
#include <type_traits>

// A wave-uniform look-up (tile list, node table) as a SCALAR load, written out: its counter (lgkmcnt) is not the vector
// loads', so a table entry can be waited for while the weight quarters requested before it are still in flight (the
// compiler's own choice for these reads is a vector load followed by s_waitcnt vmcnt(0)).  The wait is part of the
// statement.  Only for data no kernel in flight writes.
typedef int i32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
DEV int2 scalar_load(const int2 *p) {
    i32x2_t r;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return make_int2(r.x, r.y);
}
DEV int4 scalar_load(const int4 *p) {
    i32x4_t r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return make_int4(r.x, r.y, r.z, r.w);
}

// Kernel arguments are read where they are first used, each read a round trip of ~1 us to the cold argument segment in
// front of whatever needs it, and re-read later rather than kept.  PIN_ARGS forces the listed ones into scalar registers
// in one batch at the top of a kernel (as in / out operands: the compiler can no longer see that they could be re-read).
#define PIN1(x) asm volatile("" : "+s"(x))
// ... a pointer keeps its address space through the pin (an opaque generic pointer would turn every access into a flat one)
#define PIN_PTR(p)                                                                                              \
    do {                                                                                                        \
        auto pin_g_ = (__attribute__((address_space(1))) typename std::remove_pointer<decltype(p)>::type *)(p);  \
        asm volatile("" : "+s"(pin_g_));                                                                        \
        p = (decltype(p))pin_g_;                                                                                \
    } while (0)
DEV void pin_gelu(GeluK &k) {
#pragma unroll
    for (int i = 0; i <= CODLAD_GELU_DEGREE; ++i) PIN1(k.c[i]);
    PIN1(k.clamp);
}

constexpr int FRAG_U4 = 1024;                        // one tile as fragments: [8 k-steps][hi, lo][64 lanes] x 16 B

struct BlockQuarter {                                // the weight fragments of one output block of one 128x128 block
    u32x4 w[8][2];
    template <int KS0, int KS1>
    DEV void start_part(const void *Wpacked, int bo, int lane) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(Wpacked), 0, 65536, 0x00020000);
#pragma unroll
        for (int ks = KS0; ks < KS1; ++ks) {
            w[ks][0] = weight_frag_load(rsrc, lane, (ks * 4 + bo) * 2 + 0);
            w[ks][1] = weight_frag_load(rsrc, lane, (ks * 4 + bo) * 2 + 1);
        }
    }
    DEV void start(const void *Wpacked, int bo, int lane) { start_part<0, 8>(Wpacked, bo, lane); }
    // A block requested in two parts: the first HEAD_KS k-steps two blocks ahead of use (while the buffer's previous
    // block is still the current one's neighbour), the rest one block ahead, when the other buffer has been consumed -
    // 24 registers fewer in flight than two whole quarters, which is what node_kernel_w was short of (it parked
    // arriving fragments in scratch, and every scratch access waits for ALL outstanding loads).
    static constexpr int HEAD_KS = 5;
    DEV void start_head(const void *Wpacked, int bo, int lane) { start_part<0, HEAD_KS>(Wpacked, bo, lane); }
    DEV void start_tail(const void *Wpacked, int bo, int lane) { start_part<HEAD_KS, 8>(Wpacked, bo, lane); }
    // acc += W[32 bo .. 32 bo + 31][:] @ tile, the tile read as fragments from LDS
    // TRANSPOSED: operands swapped, the block arrives with lane = output feature, registers = the tile's columns
    template <int TERMS, bool TRANSPOSED = false>
    DEV void run(f32x16 &acc, const u32x4 *frag, int lane) const {
        // operand fragments one k-step ahead, and no further: left to itself the scheduler hoists all sixteen LDS reads
        // (64 registers) in front of the MFMAs and spills the weight quarters that are waiting their turn
        SplitFrag x, xn;
        x.hi = as_f16x8(frag[0 * 64 + lane]);
        x.lo = as_f16x8(frag[1 * 64 + lane]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 1 < 8) {
                xn.hi = as_f16x8(frag[((ks + 1) * 2 + 0) * 64 + lane]);
                xn.lo = as_f16x8(frag[((ks + 1) * 2 + 1) * 64 + lane]);
            }
            mfma_f16<TERMS, TRANSPOSED>(acc, as_f16x8(w[ks][0]), as_f16x8(w[ks][1]), x);
            __builtin_amdgcn_sched_barrier(0);
            x = xn;
        }
    }
};

// The 16 registers of output block `bo` ARE the elements of the fragments of k-steps 2 bo and 2 bo + 1 (register
// 8 s + j = element j of k-step 2 bo + s): activate (optionally), split, write both fragments.
template <bool GELU>
DEV void publish_quarter(u32x4 *frag, f32x16 q, int bo, int lane, const GeluK &gk) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // the four pairs of a k-step side by side (gelu_pairs<4>: the dependent Horner steps of one pair are separated by
        // those of the others; one pair at a time a publish took ~2 800 cycles for 16 values, tools/node_wide_stamps.py)
        f32x2 x[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) x[p] = f32x2{q[8 * s + 2 * p], q[8 * s + 2 * p + 1]};
        if (GELU) gelu_pairs<4>(x, gk);
        SplitFrag f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f16x2 hh = __builtin_convertvector(x[p], f16x2);
            const f16x2 ll = split_lo_pair(hh, x[p]);
            f.hi[2 * p] = hh.x; f.hi[2 * p + 1] = hh.y;
            f.lo[2 * p] = ll.x; f.lo[2 * p + 1] = ll.y;
        }
        frag[((2 * bo + s) * 2 + 0) * 64 + lane] = __builtin_bit_cast(u32x4, f.hi);
        frag[((2 * bo + s) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, f.lo);
        // one k-step at a time: interleaving both costs registers these kernels (two or three weight quarters resident)
        // do not have
        __builtin_amdgcn_sched_barrier(0);
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

// tile_layernorm_affine on a tile that lies in LDS as [32 chunks][32 columns] float4 (xch_write): the column's values of
// this lane half are streamed through the two moment sums in the order the one-wave kernel adds its registers (block
// after block, pairs of registers on two interleaved partial sums).  `own` = this wave's block bo of the same tile (the
// registers it wrote to the buffer), normalised and modulated in place.
DEV void xch_layernorm_affine(f32x16 &own, const float4 *buf, float eps, const float *A, const float *B, int bo, int h, int c) {
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = buf[(8 * b + 2 * q + h) * 32 + c];
            s2 += f32x2{v.x, v.y};
            s2 += f32x2{v.z, v.w};
        }
    const float mean = column_sum128(s2.x + s2.y) * (1.0f / 128.0f);
    const f32x2 m2 = {mean, mean};
    f32x2 v2 = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = buf[(8 * b + 2 * q + h) * 32 + c];
            const f32x2 d0 = f32x2{v.x, v.y} - m2;
            v2 = d0 * d0 + v2;
            const f32x2 d1 = f32x2{v.z, v.w} - m2;
            v2 = d1 * d1 + v2;
        }
    const float rstd = 1.0f / sqrtf(column_sum128(v2.x + v2.y) * (1.0f / 128.0f) + eps);
    const f32x2 r2 = {rstd, rstd};
    const float4 *pa = reinterpret_cast<const float4 *>(A);
    const float4 *pb = reinterpret_cast<const float4 *>(B);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = 8 * bo + 2 * q + h;
        const float4 ka = pa[o], kb = pb[o];
        tile_set_pair(own, 4 * q, (tile_pair(own, 4 * q) - m2) * (r2 * f32x2{ka.x, ka.y}) + f32x2{kb.x, kb.y});
        tile_set_pair(own, 4 * q + 2, (tile_pair(own, 4 * q + 2) - m2) * (r2 * f32x2{ka.z, ka.w}) + f32x2{kb.z, kb.w});
    }
}
