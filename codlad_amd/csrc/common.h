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
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// pair (r, r+1) of an accumulator block as a packed fp32 operand / result (adjacent, even-aligned
// registers: v_pk_add / v_pk_mul / v_pk_fma work on them in place, two elements per issue slot)
DEV f32x2 tile_pair(const f32x16 &v, int r) { return f32x2{v[r], v[r + 1]}; }
DEV void tile_set_pair(f32x16 &v, int r, f32x2 x) {
    v[r] = x.x;
    v[r + 1] = x.y;
}

DEV void tile_add_row(Tile &t, const float *row, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = p[8 * bo + 2 * q + h];
            tile_set_pair(t.b[bo], 4 * q, tile_pair(t.b[bo], 4 * q) + f32x2{v.x, v.y});
            tile_set_pair(t.b[bo], 4 * q + 2, tile_pair(t.b[bo], 4 * q + 2) + f32x2{v.z, v.w});
        }
}

// t = t * scale + row (the residual entering a block-exponent-scaled accumulator, with its pre-scaled bias)
DEV void tile_scale_add_row(Tile &t, float scale, const float *row, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(row);
    const f32x2 sc = {scale, scale};
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = p[8 * bo + 2 * q + h];
            tile_set_pair(t.b[bo], 4 * q, tile_pair(t.b[bo], 4 * q) * sc + f32x2{v.x, v.y});
            tile_set_pair(t.b[bo], 4 * q + 2, tile_pair(t.b[bo], 4 * q + 2) * sc + f32x2{v.z, v.w});
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

// Edge state (h_E0, E1, h_E) in HBM, per node one block of 64 edges x 128 features stored half by half and
// "chunk-major" within a half: [2 halves of 32 edges][32 chunks of 4 features][32 edges][4 floats].  A lane owns an
// edge (column), so with one row per edge every wave instruction touched 64 separate lines; in this order the 16
// bytes of neighbouring lanes are neighbours in memory, an instruction covers one contiguous kilobyte (lanes 0-31:
// chunk o, lanes 32-63: chunk o+1) and the 16 instructions of a 32-edge tile cover 16 contiguous KB (round 2; with
// all 64 edges of a chunk together a tile was 32 runs of 512 B at a 1 KB stride: 3 % slower when an edge kernel is
// launched back to back, no measurable difference inside the job).
// In the split-fp16 modes the 16-byte slots of h_E0 and h_E hold fp16 halves instead of fp32 numbers (same bytes,
// "pre-split edge state" below); E1 and the fp32-MFMA mode keep fp32.
#define EDGE_BLOCK (64 * 128)
#define EDGE_F4(e) (((e) & 31) + ((e) >> 5) * 1024)   // float4 index of edge e's first chunk; chunk stride 32
// STREAM: the streaming (non-temporal) hint for a wave's own edge tile in the persistent kernels of large jobs - a tile
// of the per-sample edge state is read once per launch and what is written is read next by another kernel, 1.1 GB
// later, so neither should push the gathered Q rows and the streamed weight fragments out of the 4 MB L2s (edge
// update -4 % in the job).  Blocks shared by the ensemble members of a structure (h_E0, E1) and the tile kernels of
// small jobs, whose whole state stays in cache, keep the default policy.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool STREAM = false>
DEV void tile_load_edge(Tile &t, const float *block, int e, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(block) + EDGE_F4(e);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p + (8 * bo + 2 * q + h) * 32);
#ifdef CODLAD_NO_STREAM_HINT     // A/B builds: default cache policy everywhere (does the edge state of a small job stay in the Infinity Cache?)
            const f32x4 v = *src;
#else
            const f32x4 v = STREAM ? __builtin_nontemporal_load(src) : *src;
#endif
            t.b[bo][4 * q + 0] = v.x;
            t.b[bo][4 * q + 1] = v.y;
            t.b[bo][4 * q + 2] = v.z;
            t.b[bo][4 * q + 3] = v.w;
        }
}

DEV void tile_add_edge(Tile &t, const float *block, int e, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(block) + EDGE_F4(e);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = p[(8 * bo + 2 * q + h) * 32];
            t.b[bo][4 * q + 0] += v.x;
            t.b[bo][4 * q + 1] += v.y;
            t.b[bo][4 * q + 2] += v.z;
            t.b[bo][4 * q + 3] += v.w;
        }
}

template <bool STREAM = false>
DEV void tile_store_edge(const Tile &t, float *block, int e, int h) {
    float4 *p = reinterpret_cast<float4 *>(block) + EDGE_F4(e);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {t.b[bo][4 * q + 0], t.b[bo][4 * q + 1], t.b[bo][4 * q + 2], t.b[bo][4 * q + 3]};
            f32x4 *dst = reinterpret_cast<f32x4 *>(p + (8 * bo + 2 * q + h) * 32);
#ifdef CODLAD_NO_STREAM_HINT
            *dst = v;
#else
            if (STREAM) __builtin_nontemporal_store(v, dst);
            else *dst = v;
#endif
        }
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

// ---------------------------------------------------------------------------------------------
// fp32-equivalent contraction on the fp16 matrix pipe ("f16x4" / "f16x3").
//
// v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 vector lanes: measured on MI355X
// (tools/ubench/coexec.hip) a wave issuing fp32 MFMAs and a partner wave issuing v_fma_f32 take
// exactly the SUM of their separate times, so bias/GELU/LayerNorm work is never hidden behind an
// fp32 contraction.  The f16 MFMA (32x32x16, 32 cycles for 16 k) has its own pipe and 16x the rate.
// Both operands are split into two fp16 halves by round-to-nearest,
//     x = hi + lo (+ eps),  hi = f16(x),  lo = f16(x - hi),  |eps| <= max(2^-22 |x|, 2^-25)
// (below |x| ~ 2^-3 the `lo` half is a subnormal fp16 and the error is an ABSOLUTE 2^-25; beyond 65504 there
// is no split at all: hi = inf), and the partial products hi*hi + hi*lo + lo*hi (+ lo*lo in the four-term
// mode, see mfma_f16) are accumulated in fp32 by the MFMA (fp16 x fp16 products are exact in fp32).  Per 16 k
// this costs 3-4 x 32 = 96-128 matrix-pipe cycles against 8 x 64 = 512 ALU cycles for the fp32 MFMA.
// To keep both operands in the range where the bound is the relative one, every weight block is stored with a
// power-of-two BLOCK EXPONENT chosen at pack time (weights.py) and the activations between the layers of an MLP
// carry the accumulated exponent (gelu_consts below); the scale is taken out where it is free (a LayerNorm, a
// constant multiply that exists anyway, a pre-scaled bias).  Range violations are caught by the status word.
//
// Layout: same chain as above.  k-step ks = 2*b + s consumes registers 8s..8s+7 of input block b;
// element j of lane half h is feature 32*b + 16*s + 8*(j>>2) + 4*h + (j&3).  A packed weight block
// (64 KB) is [ks 0..7][bo 0..3][split hi,lo][lane 0..63][8 halves]: one 16-byte read per lane per
// fragment, lane-linear (conflict-free from LDS, coalesced from L2).
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct SplitFrag {
    f16x8 hi, lo;
};

DEV f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

// One fp32 product as TERMS fp16 products accumulated in fp32, smallest first:
//   TERMS = 4: w_lo x_lo + w_lo x_hi + w_hi x_lo + w_hi x_hi            ("f16x4")
//   TERMS = 3: the w_lo x_lo term (<= 2^-22 |w x|, the size of the split's own representation
//              error) is dropped: a quarter fewer matrix instructions   ("f16x3", the default)
// TRANSPOSED swaps the MFMA operands.  The A and B register layouts are mirror images (lane&31 =
// the m / n index, lane>>5 and the 8 halves = k), so the same two fragments then produce the
// transposed output block: lane&31 = output feature, registers = the tile's 32 columns (edges).
template <int TERMS, bool TRANSPOSED = false>
DEV void mfma_f16(f32x16 &acc, f16x8 whi, f16x8 wlo, const SplitFrag &x) {
    static_assert(TERMS == 3 || TERMS == 4, "f16x3 or f16x4");
    if (TRANSPOSED) {
        if (TERMS == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.lo, wlo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.hi, wlo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.lo, whi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.hi, whi, acc, 0, 0, 0);
    } else {
        if (TERMS == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, x.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, x.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, x.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, x.hi, acc, 0, 0, 0);
    }
}

// acc += W(128x128, split-fp16 packed block in LDS) @ in: 32 groups (k-step, out block) of TERMS f16
// MFMAs; weight fragments are read from LDS two groups ahead through a register ring.
//
// Measured on MI355X (tools/ubench/coexec2.hip, phases.hip, phases2.hip): about six fp32 VALU
// instructions ride free under one f16 MFMA, the rest of the vector work adds to the matrix time,
// whether it is interleaved in the same wave, run as a separate phase or issued by the SIMD
// partner; so the goal is simply the fewest VALU instructions.
struct GeluK;
template <int N>
DEV void gelu_pairs(f32x2 (&x)[N], const GeluK &gk);

// lo = f16(x - f32(hi)) for a pair, in two instructions: v_fma_mix{lo,hi}_f16 read the fp16 half of
// `hi` directly, form hi * -1 + x in fp32 (exact: hi is x rounded to 11 bits) and write the rounded
// fp16 result into one half of the destination.  Same bits as convert / subtract / convert (checked
// on 2^21 values), three instructions per pair for the whole split instead of five.
DEV f16x2 split_lo_pair(f16x2 hi, f32x2 x) {
    const unsigned hb = __builtin_bit_cast(unsigned, hi);
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hb), "v"(x.x));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hb), "v"(x.y));
    return __builtin_bit_cast(f16x2, lo);
}

// hi/lo fp16 split of register pair P (0..3) of k-step KS into elements 2P, 2P+1 of a fragment.
// GELU_IN: the tile holds pre-activations and GELU is applied here, on the way into the
// contraction - the activation of k-step ks+1 then sits between the MFMAs of k-step ks of the
// SAME tile (no second tile, no extra registers), where the matrix pipe hides it.
template <bool GELU_IN>
DEV void split_pair(SplitFrag &f, const Tile &in, int ks, int p, const GeluK &gk) {
    const f32x16 &v = in.b[ks >> 1];
    const int r = (ks & 1) * 8 + 2 * p;
    f32x2 x = {v[r], v[r + 1]};
    if (GELU_IN) {
        f32x2 t[1] = {x};
        gelu_pairs<1>(t, gk);
        x = t[0];
    }
    const f16x2 hh = __builtin_convertvector(x, f16x2);
    const f16x2 ll = split_lo_pair(hh, x);
    f.hi[2 * p] = hh.x; f.hi[2 * p + 1] = hh.y;
    f.lo[2 * p] = ll.x; f.lo[2 * p + 1] = ll.y;
}

// Pre-split edge state (round 3).  The edge state h_E (and h_E0) of the split-fp16 modes is STORED as the fp16 halves the
// contractions consume: the 16-byte slot that held four fp32 features now holds eight fp16 halves, laid out so that a
// k-step's B-operand fragments are whole slots - for the eight features a lane half owns in k-step (b, s),
//   f0 + {0,1,2,3} and f0 + 8 + {0,1,2,3},  f0 = 32 b + 16 s + 4 h,
// the slot at chunk 8 b + 4 s + h holds their eight `hi` halves and the slot at chunk 8 b + 4 s + 2 + h their eight `lo`
// halves.  Loaded with the same 16 instructions as before, registers 8 s .. 8 s + 3 of block b ARE the hi fragment and
// 8 s + 4 .. 8 s + 7 the lo fragment: the layer-1 operand costs no vector instruction at all (before: convert + two
// v_fma_mix per pair, 96 per tile and lane in the message kernel).  hi / lo are exactly what the on-the-fly split of the
// fp32 value produces, so every contraction is bit-identical to round 2's; only a consumer that needs the VALUE (the edge
// update's residual) sees hi + lo instead of the fp32 number, a difference of at most 2^-22 relative (2^-25 absolute
// below 2^-3) - the representation error the contractions always had.
DEV void presplit_frag(SplitFrag &f, const Tile &in, int ks) {
    // four-register sub-vectors taken with constant shuffles and bit-cast as a whole: building the fragment element by
    // element (u32x4{bits(v[r]), ...}) makes ROCm 7.2's clang narrow the tile's 16-byte loads to 4 bytes and splat
    // element 0 (wrong code; the same pitfall as weight_load above)
    const f32x16 &v = in.b[ks >> 1];
    f32x4 a, b;
    if (ks & 1) {
        a = __builtin_shufflevector(v, v, 8, 9, 10, 11);
        b = __builtin_shufflevector(v, v, 12, 13, 14, 15);
    } else {
        a = __builtin_shufflevector(v, v, 0, 1, 2, 3);
        b = __builtin_shufflevector(v, v, 4, 5, 6, 7);
    }
    f.hi = __builtin_bit_cast(f16x8, a);
    f.lo = __builtin_bit_cast(f16x8, b);
}

// fp32 chain tile -> stored form, in place (eight features of a k-step at a time)
DEV void tile_presplit(Tile &t) {
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x2 hi[4], lo[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x2 x = {t.b[bo][8 * s + 2 * p], t.b[bo][8 * s + 2 * p + 1]};
                hi[p] = __builtin_convertvector(x, f16x2);
                lo[p] = split_lo_pair(hi[p], x);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                t.b[bo][8 * s + p] = __builtin_bit_cast(float, hi[p]);
                t.b[bo][8 * s + 4 + p] = __builtin_bit_cast(float, lo[p]);
            }
        }
}

// stored form -> values (hi + lo in fp32), in place, fused with the edge update's residual set-up:
// t = (hi + lo) * scale + row  (scale = 2^E of the block exponents, row = the pre-scaled bias b13)
DEV void tile_unsplit_scale_add_row(Tile &t, float scale, const float *row, int h) {
    const float4 *pr = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the halves are taken out of whole four-register sub-vectors (see presplit_frag: bit-casting single
            // elements of the tile to fp16 pairs is miscompiled by ROCm 7.2's clang)
            SplitFrag f;
            presplit_frag(f, t, 2 * bo + s);
            const u32x4 hb = __builtin_bit_cast(u32x4, f.hi), lb = __builtin_bit_cast(u32x4, f.lo);
            float val[8];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // hi + lo in fp32 (exact: 22 bits), one v_fma_mix_f32 per element: f16 * 1.0 + f16
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(val[2 * p]) : "v"(hb[p]), "v"(lb[p]));
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(val[2 * p + 1]) : "v"(hb[p]), "v"(lb[p]));
            }
            const float4 ra = pr[8 * bo + 2 * (2 * s) + h], rb = pr[8 * bo + 2 * (2 * s + 1) + h];
            t.b[bo][8 * s + 0] = fmaf(val[0], scale, ra.x); t.b[bo][8 * s + 1] = fmaf(val[1], scale, ra.y);
            t.b[bo][8 * s + 2] = fmaf(val[2], scale, ra.z); t.b[bo][8 * s + 3] = fmaf(val[3], scale, ra.w);
            t.b[bo][8 * s + 4] = fmaf(val[4], scale, rb.x); t.b[bo][8 * s + 5] = fmaf(val[5], scale, rb.y);
            t.b[bo][8 * s + 6] = fmaf(val[6], scale, rb.z); t.b[bo][8 * s + 7] = fmaf(val[7], scale, rb.w);
        }
}

// PRESPLIT: `in` holds the stored (pre-split) form of an edge tile; its fragments are taken as they are.
// AHEAD = groups the fragment reads run ahead of their use (ring of AHEAD + 1 pairs): 2 for the kernels with two waves per
// SIMD; the one-wave-per-SIMD edge update takes 1 (8 registers fewer; a group is then ~130 cycles against ~64 of LDS latency).
template <int TERMS, int KS0, int NKS, bool GELU_IN, bool TRANSPOSED = false, bool PRESPLIT = false, int AHEAD = 2>
DEV void gemm_h_lds(Tile &acc, const Tile &in, const u32x4 *wl, int lane, const GeluK &gk) {
    static_assert(!(GELU_IN && PRESPLIT), "a stored tile is never an activation input");
    const u32x4 *w = wl + lane;
    constexpr int G0 = KS0 * 4, NG = NKS * 4, R = AHEAD + 1;
    u32x4 ring[R][2];
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) {
        ring[g][0] = w[((G0 + g) * 2 + 0) * 64];
        ring[g][1] = w[((G0 + g) * 2 + 1) * 64];
    }
    SplitFrag x, xn;
    if (PRESPLIT) presplit_frag(x, in, KS0);
    else {
#pragma unroll
        for (int p = 0; p < 4; ++p) split_pair<GELU_IN>(x, in, KS0, p, gk);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int ks = KS0 + (g >> 2), bo = g & 3;
        if (g + AHEAD < NG) {
            ring[(g + AHEAD) % R][0] = w[((G0 + g + AHEAD) * 2 + 0) * 64];
            ring[(g + AHEAD) % R][1] = w[((G0 + g + AHEAD) * 2 + 1) * 64];
        }
        if (!PRESPLIT && ks + 1 < KS0 + NKS) split_pair<GELU_IN>(xn, in, ks + 1, bo, gk);
        mfma_f16<TERMS, TRANSPOSED>(acc.b[bo], as_f16x8(ring[g % R][0]), as_f16x8(ring[g % R][1]), x);
        __builtin_amdgcn_sched_barrier(0);
        if (bo == 3) {
            if (PRESPLIT) { if (ks + 1 < KS0 + NKS) presplit_frag(x, in, ks + 1); }
            else x = xn;
        }
    }
}

// acc += W @ act(in): GELU_IN applies GELU to `in` on the fly (see split_pair)
template <int TERMS, bool GELU_IN, bool TRANSPOSED = false, bool PRESPLIT = false, int AHEAD = 2>
DEV void gemm128_h_lds(Tile &acc, const Tile &in, const u32x4 *wl, int lane, const GeluK &gk) {
    gemm_h_lds<TERMS, 0, 8, GELU_IN, TRANSPOSED, PRESPLIT, AHEAD>(acc, in, wl, lane, gk);
}

// The same k-step with the weight fragments fetched from global memory (L2-resident) through a
// buffer descriptor (8 coalesced 1 KiB wave loads per k-step).
DEV u32x4 weight_frag_load(__amdgpu_buffer_rsrc_t rsrc, int lane, int frag_index) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, frag_index * 1024, 0));
}

// DEPTH = groups in flight.  start() issues the first DEPTH-1 groups' fragment loads and can be
// called long before run(), so that the L2 latency of a streamed block is paid while the wave does
// something else (other loads, the LDS-resident part of the same contraction).
template <int TERMS, int KS0, int NKS, bool GELU_IN, int DEPTH = 4, bool PRESPLIT = false>
struct StreamedGemm {
    static constexpr int G0 = KS0 * 4, NG = NKS * 4;
    static_assert((DEPTH & (DEPTH - 1)) == 0 && DEPTH <= NG, "ring depth: power of two, at most the group count");
    __amdgpu_buffer_rsrc_t rsrc;
    u32x4 ring[DEPTH][2];

    DEV void start(const void *Wpacked, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(Wpacked), 0, 65536, 0x00020000);
#pragma unroll
        for (int g = 0; g < DEPTH - 1; ++g) {
            ring[g][0] = weight_frag_load(rsrc, lane, (G0 + g) * 2 + 0);
            ring[g][1] = weight_frag_load(rsrc, lane, (G0 + g) * 2 + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    DEV void run(Tile &acc, const Tile &in, int lane, const GeluK &gk) {
        SplitFrag x, xn;
        if (PRESPLIT) presplit_frag(x, in, KS0);
        else {
#pragma unroll
            for (int p = 0; p < 4; ++p) split_pair<GELU_IN>(x, in, KS0, p, gk);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ks = KS0 + (g >> 2), bo = g & 3;
            if (g + DEPTH - 1 < NG) {
                ring[(g + DEPTH - 1) & (DEPTH - 1)][0] = weight_frag_load(rsrc, lane, (G0 + g + DEPTH - 1) * 2 + 0);
                ring[(g + DEPTH - 1) & (DEPTH - 1)][1] = weight_frag_load(rsrc, lane, (G0 + g + DEPTH - 1) * 2 + 1);
            }
            if (!PRESPLIT && ks + 1 < KS0 + NKS) split_pair<GELU_IN>(xn, in, ks + 1, bo, gk);
            mfma_f16<TERMS>(acc.b[bo], as_f16x8(ring[g & (DEPTH - 1)][0]), as_f16x8(ring[g & (DEPTH - 1)][1]), x);
            __builtin_amdgcn_sched_barrier(0);
            if (bo == 3) {
                if (PRESPLIT) { if (ks + 1 < KS0 + NKS) presplit_frag(x, in, ks + 1); }
                else x = xn;
            }
        }
    }
};

template <int TERMS, int KS0, int NKS, bool GELU_IN, bool PRESPLIT = false>
DEV void gemm_h_glb(Tile &acc, const Tile &in, const void *Wpacked, int lane, const GeluK &gk) {
    StreamedGemm<TERMS, KS0, NKS, GELU_IN, 4, PRESPLIT> g;
    g.start(Wpacked, lane);
    g.run(acc, in, lane, gk);
}

// GELU(x) = x Phi(x) = max(x, 0) - |x| * (erfc(|x|/sqrt 2) / 2), branch-free:
//   s = min(|x|, 4 sqrt 2),   erfc(s/sqrt 2)/2 = exp2(s g(s) - 1),
// g a polynomial fit of log2(erfc(s/sqrt 2))/s on [0, 4 sqrt 2] (so neither the 1/sqrt 2 nor the 1/2
// costs an instruction; one v_exp_f32 finishes it).  The fit minimises the error of GELU itself,
// |x| (erfc/2) ln2 s |dg|: where erfc is tiny, g may be far off, and that is what lets a low degree
// reach fp32 level.  Degree CODLAD_GELU_DEGREE = 5 (default; 4, 6 and 8 selectable for A/B runs):
//   max |error| over |x| < 1 / 2 / 4 / 8:   deg 5  1.4e-7 / 1.6e-7 / 2.2e-7 / 3.1e-7
//                                           deg 6  0.9e-7 / 1.2e-7 / 1.6e-7 / 2.8e-7
//                                           deg 8  0.6e-7 / 0.9e-7 / 1.4e-7 / 2.4e-7
//   x/2 (1 + erf(x/sqrt 2)) in fp32 with a correctly rounded erf (what the reference computes):
//                                                  0.9e-7 / 1.6e-7 / 3.0e-7 / 4.5e-7
// (beyond |x| ~ 2 every variant is at the rounding of its own fp32 output).  x * (tiny) keeps full
// relative accuracy in the negative tail.  10 VALU instructions at degree 5 (min, 6 fma, exp, max,
// fma with |.| and - as source modifiers) against ~45 plus divergent branches for the library erff.
// -DCODLAD_EXACT_ERF selects erff() for A/B validation.
#ifndef CODLAD_GELU_DEGREE
#define CODLAD_GELU_DEGREE 5
#endif
#define CODLAD_GELU_CLAMP 5.656854249492381f
#if CODLAD_GELU_DEGREE == 4     // A/B only: one instruction per element less, max |error| 6e-7 .. 9e-7 (2-4 x degree 5)
#define CODLAD_GELU_COEFFS {-0.0004881024651695043f, 0.007198719307780266f, -0.052146632224321365f, \
                            -0.45959585905075073f, -1.1510004997253418f}
#elif CODLAD_GELU_DEGREE == 5
#define CODLAD_GELU_COEFFS {2.992418740177527e-05f, -0.0007398742018267512f, 0.007977462373673916f, \
                            -0.05323818698525429f, -0.45891568064689636f, -1.1511471271514893f}
#elif CODLAD_GELU_DEGREE == 6
#define CODLAD_GELU_COEFFS {4.27874283559504e-06f, -1.2798205716535449e-05f, -0.0005757861654274166f, \
                            0.0076707834377884865f, -0.052948564291000366f, -0.4590439200401306f, -1.1511269807815552f}
#elif CODLAD_GELU_DEGREE == 8
#define CODLAD_GELU_COEFFS {5.136640197633824e-07f, -9.57114389166236e-06f, 7.503479719161987e-05f,          \
                            -0.00028452760307118297f, 1.5291185263777152e-05f, 0.006930838339030743f,       \
                            -0.05243462696671486f, -0.4592214822769165f, -1.1511043310165405f}
#else
#error "CODLAD_GELU_DEGREE must be 4, 5, 6 or 8"
#endif
// GELU on a value that carries a power-of-two scale (block exponents of the split-fp16 modes, see
// weights.py / DESIGN.md): for x' = 2^E x the functions below return 2^E GELU(x), bit for bit the scaled
// result of the unscaled evaluation, at the same instruction count: with s' = 2^E s the Horner steps become
// p' <- p' s' + c_k 2^(-E (D+1-k)) (every intermediate an exact power-of-two multiple of the unscaled one),
// the exponent argument s' p' - 1 is the unscaled one, and max(x', 0) - |x'| e = 2^E (max(x, 0) - |x| e).
// The constants are per-launch values (kernel arguments -> SGPRs); E = 0 gives the plain coefficients.
struct GeluK {
    float c[CODLAD_GELU_DEGREE + 1];
    float clamp;       // CODLAD_GELU_CLAMP * 2^E
    float up, down;    // 2^E, 2^-E (exact-erf A/B build only)
};

__host__ __device__ inline float pow2i(int e) {     // 2^e, |e| <= 126
    union { unsigned u; float f; } v;
    v.u = (unsigned)(127 + e) << 23;
    return v.f;
}

__host__ __device__ inline GeluK gelu_consts(int E) {
    constexpr float cf[CODLAD_GELU_DEGREE + 1] = CODLAD_GELU_COEFFS;
    GeluK k;
    for (int i = 0; i <= CODLAD_GELU_DEGREE; ++i) k.c[i] = cf[i] * pow2i(-E * (CODLAD_GELU_DEGREE + 1 - i));
    k.clamp = CODLAD_GELU_CLAMP * pow2i(E);
    k.up = pow2i(E);
    k.down = pow2i(-E);
    return k;
}
constexpr int CODLAD_MAX_CHAIN_EXP = 16;   // |E| bound of a GELU input: c_0 2^(-6E) stays a normal fp32

DEV float gelu_erf(float x, const GeluK &gk) {
#ifdef CODLAD_EXACT_ERF
    const float u = x * gk.down;
    return ((0.5f * u) * (1.0f + erff(u * 0.70710678118654752440f))) * gk.up;
#else
    const float s = fminf(fabsf(x), gk.clamp);
    float p = gk.c[0];
#pragma unroll
    for (int k = 1; k <= CODLAD_GELU_DEGREE; ++k) p = fmaf(p, s, gk.c[k]);
    const float e = __builtin_amdgcn_exp2f(fmaf(p, s, -1.0f));
    return fmaf(-e, fabsf(x), fmaxf(x, 0.0f));
#endif
}

// v + (v of another lane) through the DPP crossbar: one VALU instruction (v_add_f32_dpp), no LDS.
template <int CTRL>
DEV float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Sum over the 32 lanes of each wave half.  The totals land in lanes 16-31 (lower half) and
// 48-63 (upper half); other lanes hold partial sums.
DEV float half_wave_sum(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    // row_bcast:15 adds lane 15 of the previous row; rows 1 and 3 then hold their half's total
    // (rows 0 and 2 receive nothing useful and are not read)
    v = dpp_add<0x142>(v);
    return v;
}

// The same on packed fp32 math (v_pk_fma_f32 for the Horner steps), N pairs evaluated side by
// side so that the dependent steps of one pair are separated by those of the others (a
// v_pk_fma_f32 that consumes the previous one's result otherwise costs an s_nop).
template <int N>
DEV void gelu_pairs(f32x2 (&x)[N], const GeluK &gk) {
#ifdef CODLAD_EXACT_ERF
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x[i].x = gelu_erf(x[i].x, gk);
        x[i].y = gelu_erf(x[i].y, gk);
    }
#else
    f32x2 t[N], p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        t[i].x = fminf(fabsf(x[i].x), gk.clamp);
        t[i].y = fminf(fabsf(x[i].y), gk.clamp);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = t[i] * gk.c[0] + gk.c[1];
#pragma unroll
    for (int k = 2; k <= CODLAD_GELU_DEGREE; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = p[i] * t[i] + gk.c[k];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = p[i] * t[i] + -1.0f;     // s g(s) - 1
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float ex = __builtin_amdgcn_exp2f(p[i].x), ey = __builtin_amdgcn_exp2f(p[i].y);
        x[i].x = fmaf(-ex, fabsf(x[i].x), fmaxf(x[i].x, 0.0f));
        x[i].y = fmaf(-ey, fabsf(x[i].y), fmaxf(x[i].y, 0.0f));
    }
#endif
}

DEV void tile_gelu(Tile &t, const GeluK &gk) {
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; r += 8) {
            f32x2 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = f32x2{t.b[bo][r + 2 * i], t.b[bo][r + 2 * i + 1]};
            gelu_pairs<4>(v, gk);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t.b[bo][r + 2 * i] = v[i].x;
                t.b[bo][r + 2 * i + 1] = v[i].y;
            }
        }
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

// LayerNorm followed by modulate with the affine form folded: t = (t - mean) * (rstd * A) + B,
// A = gate (1 + scale), B = gate shift (two vectors instead of three, two VALU per element
// instead of four; the moments are computed exactly as in tile_layernorm).
DEV void tile_layernorm_affine(Tile &t, float eps, const float *A, const float *B, int h) {
    // packed fp32 throughout (pairs of adjacent accumulator registers): the sums run over two
    // interleaved partial sums, which is also what halves their dependency chains
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; r += 2) s2 += tile_pair(t.b[bo], r);
    const float mean = column_sum128(s2.x + s2.y) * (1.0f / 128.0f);
    const f32x2 m2 = {mean, mean};
    f32x2 v2 = {0.f, 0.f};
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 d = tile_pair(t.b[bo], r) - m2;
            tile_set_pair(t.b[bo], r, d);
            v2 = d * d + v2;
        }
    const float rstd = 1.0f / sqrtf(column_sum128(v2.x + v2.y) * (1.0f / 128.0f) + eps);
    const f32x2 r2 = {rstd, rstd};
    const float4 *pa = reinterpret_cast<const float4 *>(A);
    const float4 *pb = reinterpret_cast<const float4 *>(B);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = 8 * bo + 2 * q + h;
            const float4 ka = pa[o], kb = pb[o];
            tile_set_pair(t.b[bo], 4 * q, tile_pair(t.b[bo], 4 * q) * (r2 * f32x2{ka.x, ka.y}) + f32x2{kb.x, kb.y});
            tile_set_pair(t.b[bo], 4 * q + 2, tile_pair(t.b[bo], 4 * q + 2) * (r2 * f32x2{ka.z, ka.w}) + f32x2{kb.z, kb.w});
        }
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

// One reverse-diffusion update of a scalar (gaussian_diffusion.py:303-349, 364-367, 246-249, 446),
// every product and sum rounded separately like the reference's elementwise tensor ops.
// cf = one row of Tables.step_coefficients(): {sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1,
// posterior_mean_coef2, log variance (minimum, or THE log variance of the fixed-variance samplers), log beta, nonzero mask,
// mode}; mode bits (p_mean_variance's branches): 1 = the model predicts x_0 (ModelMeanType.START_X) instead of the noise,
// 2 = fixed variance (FIXED_SMALL / FIXED_LARGE: cf[4] is the step's log variance, `v` is not read), 4 = clip_denoised
// (pred_xstart clamped into [-1, 1]).  mode 0 = epsilon prediction with the learned-range variance, what test.py samples with.
// *x0_out (optional) receives pred_xstart, the self-conditioning input of the next step.
#define CODLAD_DDPM_START_X 1
#define CODLAD_DDPM_FIXED_VAR 2
#define CODLAD_DDPM_CLIP 4
DEV float ddpm_step(float xt, float out, float v, const float *cf, float noise, float *x0_out = nullptr) {
#pragma clang fp contract(off)
    const int mode = (int)cf[7];
    float logvar = cf[4];
    if (!(mode & CODLAD_DDPM_FIXED_VAR)) {
        const float frac = (v + 1.0f) / 2.0f;
        logvar = frac * cf[5] + (1.0f - frac) * cf[4];
    }
    float x0 = (mode & CODLAD_DDPM_START_X) ? out : cf[0] * xt - cf[1] * out;
    if (mode & CODLAD_DDPM_CLIP) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    const float mean = cf[2] * x0 + cf[3] * xt;
    if (x0_out) *x0_out = x0;
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
