// "Wide" split-fp16 edge kernels for SMALL jobs, their own translation unit (see edge_args.h).
//
// The tile kernels (edge_tile_kernels.hip) give a 32-edge tile to ONE wave, which then runs two or three dependent
// contractions of 96 MFMAs behind a 128-152 KB fill of its workgroup's LDS: 18 / 26 us per launch however few tiles the
// job has (one protein of 87 residues: 174 tiles on a chip with 1 024 SIMDs).  Here a workgroup of FOUR waves owns the
// tile, like the node update of small jobs (node_wide_kernels.hip, helpers in wide_common.h):
//   * wave bo computes output block bo of every contraction - 8 k-steps x TERMS MFMAs instead of 32 - with its 16 KB
//     quarter of each weight block loaded straight from L2 into registers (all of them requested before anything is
//     waited for; no LDS fill, and a persistent workgroup keeps them across its tiles);
//   * the operand tile travels between the waves as split-fp16 fragments in LDS: the wave that owns block b holds
//     exactly the registers that make up the fragments of k-steps 2 b and 2 b + 1, so each element is activated and
//     split once, by its producer; the stored (pre-split) edge state is published as it is;
//   * a wave loads and stores only its own quarter of the tile, of the Q row and of the P row;
//   * the LayerNorm of the edge update needs whole columns: the fp32 tile is exchanged through LDS and every wave
//     streams the column's 64 values of its lane half through the moment sums in the one-wave kernel's order (no
//     64-register tile), keeping only its own block.
// Same arithmetic in the same order per element and per accumulator => bit-identical to upd_kernel_h / msg_kernel_h
// and the tile kernels (tests/test_hip_parity.py).  The message kernel writes the per-half, per-lane-half partial
// sums of the tile kernels (planes of S), which the node kernel adds up in the per-node kernel's order.
#include "edge_args.h"
#include "wide_common.h"

namespace {

constexpr int EW_WAVES = 4;
// LDS map in 16-byte words
constexpr int EW_FRAG_X = 0;                          // the stored tile as fragments (layer-1 operand)
constexpr int EW_FRAG_A = EW_FRAG_X + FRAG_U4;        // GELU(layer 1)
constexpr int EW_FRAG_B = EW_FRAG_A + FRAG_U4;        // GELU(layer 2)          (edge update)
constexpr int EW_XCH = EW_FRAG_B + FRAG_U4;           // fp32 tile [32 chunks][32 columns] float4 (edge update)
constexpr int EW_VEC = EW_XCH + 1024;                 // b2, b3, modulate A, modulate B (128 floats each)
constexpr int EW_END = EW_VEC + 4 * 32;
constexpr int EW_LDS_BYTES = EW_END * 16;

// this wave's quarter (block bo) of edge column e of a stored edge block
DEV void edge_quarter_load(f32x16 &a, const float *block, int e, int bo, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(block) + EDGE_F4(e);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p + (8 * bo + 2 * q + h) * 32);
        a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
}
DEV void edge_quarter_store(const f32x16 &a, float *block, int e, int bo, int h) {
    float4 *p = reinterpret_cast<float4 *>(block) + EDGE_F4(e);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = {a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
        *reinterpret_cast<f32x4 *>(p + (8 * bo + 2 * q + h) * 32) = v;
    }
}

// The stored form of block bo IS the fragments of k-steps 2 bo, 2 bo + 1 (common.h, "pre-split edge state"): registers
// 8 s .. 8 s + 3 the hi fragment, 8 s + 4 .. 8 s + 7 the lo fragment.  Whole four-register sub-vectors (presplit_frag).
DEV void publish_stored_quarter(u32x4 *frag, const f32x16 &v, int bo, int lane) {
    const f32x4 h0 = __builtin_shufflevector(v, v, 0, 1, 2, 3), l0 = __builtin_shufflevector(v, v, 4, 5, 6, 7);
    const f32x4 h1 = __builtin_shufflevector(v, v, 8, 9, 10, 11), l1 = __builtin_shufflevector(v, v, 12, 13, 14, 15);
    frag[((2 * bo + 0) * 2 + 0) * 64 + lane] = __builtin_bit_cast(u32x4, h0);
    frag[((2 * bo + 0) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, l0);
    frag[((2 * bo + 1) * 2 + 0) * 64 + lane] = __builtin_bit_cast(u32x4, h1);
    frag[((2 * bo + 1) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, l1);
}

// tile_unsplit_scale_add_row for one block: t = (hi + lo) * scale + row
DEV void quarter_unsplit_scale_add_row(f32x16 &t, float scale, const float *row, int bo, int h) {
    const float4 *pr = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x4 a, b;
        if (s) {
            a = __builtin_shufflevector(t, t, 8, 9, 10, 11);
            b = __builtin_shufflevector(t, t, 12, 13, 14, 15);
        } else {
            a = __builtin_shufflevector(t, t, 0, 1, 2, 3);
            b = __builtin_shufflevector(t, t, 4, 5, 6, 7);
        }
        const u32x4 hb = __builtin_bit_cast(u32x4, a), lb = __builtin_bit_cast(u32x4, b);
        float val[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(val[2 * p]) : "v"(hb[p]), "v"(lb[p]));
            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(val[2 * p + 1]) : "v"(hb[p]), "v"(lb[p]));
        }
        const float4 ra = pr[8 * bo + 2 * (2 * s) + h], rb = pr[8 * bo + 2 * (2 * s + 1) + h];
        t[8 * s + 0] = fmaf(val[0], scale, ra.x); t[8 * s + 1] = fmaf(val[1], scale, ra.y);
        t[8 * s + 2] = fmaf(val[2], scale, ra.z); t[8 * s + 3] = fmaf(val[3], scale, ra.w);
        t[8 * s + 4] = fmaf(val[4], scale, rb.x); t[8 * s + 5] = fmaf(val[5], scale, rb.y);
        t[8 * s + 6] = fmaf(val[6], scale, rb.z); t[8 * s + 7] = fmaf(val[7], scale, rb.w);
    }
}

// tile_presplit for one block
DEV void quarter_presplit(f32x16 &t) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f16x2 hi[4], lo[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x2 x = {t[8 * s + 2 * p], t[8 * s + 2 * p + 1]};
            hi[p] = __builtin_convertvector(x, f16x2);
            lo[p] = split_lo_pair(hi[p], x);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            t[8 * s + p] = __builtin_bit_cast(float, hi[p]);
            t[8 * s + 4 + p] = __builtin_bit_cast(float, lo[p]);
        }
    }
}

// Edge update, per tile:
//   h_E[n,j] <- mod3(LN(h_E[n,j] + W13 GELU(W12 GELU(P_i + Q_j + W11e h_E[n,j]) + b12) + b13))
// HOISTED: encoder layer 0 with a.E1 given - layer 1's edge contraction comes precomputed.
// (A variant that read a tile's addresses from a per-job table - one look-up instead of tile list -> node table ->
// neighbour list - and requested the next tile's rows a tile ahead was measured and dropped: no faster where a workgroup
// walks several tiles, where the waves' own instructions are the limit, and slower where it has one.)
template <bool HOISTED, int TERMS>
__global__ __launch_bounds__(EW_WAVES * 64, 2) void upd_wide_kernel(EdgeTileArgs a) {
    extern __shared__ __align__(16) u32x4 wl[];
    // every argument in one batch, pinned (wide_common.h)
    const int2 *tile_list = a.tile_list;
    const int4 *node_info = a.node_info;
    const int32_t *E_idx = a.E_idx;
    const float *hE_in = a.hE_in, *E1 = a.E1, *P = a.P, *Q = a.Q;
    float *hE_out = a.hE_out;
    const void *W1h = a.W1h, *W2h = a.W2h, *W3h = a.W3h;
    int n_tiles = a.n_tiles, in_by_src = a.in_by_src;
    float res_scale = a.res_scale, ln_eps = a.ln_eps;
    GeluK gelu_a = a.gelu_a, gelu_b = a.gelu_b;
    PIN_PTR(tile_list); PIN_PTR(node_info); PIN_PTR(E_idx); PIN_PTR(hE_in); PIN_PTR(E1); PIN_PTR(P); PIN_PTR(Q); PIN_PTR(hE_out);
    PIN_PTR(W1h); PIN_PTR(W2h); PIN_PTR(W3h); PIN1(n_tiles); PIN1(in_by_src); PIN1(res_scale); PIN1(ln_eps);
    pin_gelu(gelu_a); pin_gelu(gelu_b);
    asm volatile("" :: "s"(a.b2), "s"(a.b3), "s"(a.mods3));
    u32x4 *fragX = wl + EW_FRAG_X, *fragA = wl + EW_FRAG_A, *fragB = wl + EW_FRAG_B;
    float4 *xch = reinterpret_cast<float4 *>(wl + EW_XCH);
    const float *c_b2 = reinterpret_cast<const float *>(wl + EW_VEC), *c_b3 = c_b2 + HD;
    const float *c_modA = c_b2 + 2 * HD, *c_modB = c_b2 + 3 * HD;
    const int tid = threadIdx.x, lane = tid & 63, bo = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    // W12's quarter stays in registers for all of the workgroup's tiles; W11e's and W13's are requested per tile (three
    // resident quarters leave too few registers for a tile at two waves per SIMD: the compiler then parks arriving
    // fragments in scratch, and every scratch access waits for ALL outstanding loads)
    BlockQuarter w1, w2, w3;
    w2.start(W2h, bo, lane);
    {   // per-launch vectors -> LDS (the folded modulation exactly as the tile kernels compute it); first read after the
        // first barrier of the tile loop
        const int i = tid & 31;
        if (tid < 32) wl[EW_VEC + i] = reinterpret_cast<const u32x4 *>(a.b2)[i];
        else if (tid >= 64 && tid < 96) wl[EW_VEC + 32 + i] = reinterpret_cast<const u32x4 *>(a.b3)[i];
        else if (tid >= 128 && tid < 160) {
            const float4 *m = reinterpret_cast<const float4 *>(a.mods3);
            const float4 s = m[i], cc = m[32 + i], g = m[64 + i];
            float4 *cf = reinterpret_cast<float4 *>(wl + EW_VEC);
            cf[64 + i] = make_float4(g.x * (1.0f + cc.x), g.y * (1.0f + cc.y), g.z * (1.0f + cc.z), g.w * (1.0f + cc.w));
            cf[96 + i] = make_float4(g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w);
        }
    }
    int grid = gridDim.x;
    PIN1(grid);
    for (int t = blockIdx.x; t < n_tiles; t += grid) {
        const int2 tn = scalar_load(tile_list + t);
        const int4 info = scalar_load(node_info + tn.x);
        const int n = tn.x, src = info.x, base = info.y, K = info.z;
        const int col = 32 * tn.y + c;
        const bool valid = col < K;
        const int colc = valid ? col : 0;
        const int j = E_idx[(size_t)src * 64 + colc];
        if (!HOISTED) w1.start(W1h, bo, lane);
        else w3.start(W3h, bo, lane);
        f32x16 xq, acc, pq;
        edge_quarter_load(xq, hE_in + (size_t)(in_by_src ? src : n) * EDGE_BLOCK, colc, bo, h);   // operand and residual
        quarter_load(acc, Q + (size_t)(base + j) * HD, bo, h);
        quarter_load(pq, P + (size_t)n * HD, bo, h);
        if (HOISTED) {
            f32x16 e1;
            edge_quarter_load(e1, E1 + (size_t)src * EDGE_BLOCK, colc, bo, h);
            acc += pq;
            acc += e1;
        } else {
            publish_stored_quarter(fragX, xq, bo, lane);
            acc += pq;
            __syncthreads();
            w1.run<TERMS>(acc, fragX, lane);                                   // layer 1
            w3.start(W3h, bo, lane);
        }
        publish_quarter<true>(fragA, acc, bo, lane, gelu_a);
        __syncthreads();
        f32x16 t2;
        quarter_load(t2, c_b2, bo, h);
        w2.run<TERMS>(t2, fragA, lane);                                        // layer 2 on GELU(layer 1)
        publish_quarter<true>(fragB, t2, bo, lane, gelu_b);
        // layer 3 accumulates onto (h_E + b13) * 2^E (c_b3 holds b13 * 2^E)
        quarter_unsplit_scale_add_row(xq, res_scale, c_b3, bo, h);
        __syncthreads();
        w3.run<TERMS>(xq, fragB, lane);                                        // layer 3 on GELU(layer 2)
        xch_write(xch, xq, bo, h, c);
        __syncthreads();
        xch_layernorm_affine(xq, xch, ln_eps, c_modA, c_modB, bo, h, c);
        quarter_presplit(xq);
        if (valid) edge_quarter_store(xq, hE_out + (size_t)n * EDGE_BLOCK, col, bo, h);
    }
}

// Message kernel, per tile: S[half + 2 h][n] = sum over the tile's valid edges of GELU(W2 GELU(P_i + Q_j + W1e h_E[i,j]) + b2),
// the last contraction with swapped operands (lane = feature, registers = edges) as in msg_kernel_h.
template <bool HOISTED, int TERMS>
__global__ __launch_bounds__(EW_WAVES * 64, 2) void msg_wide_kernel(EdgeTileArgs a) {
    extern __shared__ __align__(16) u32x4 wl[];
    const int2 *tile_list = a.tile_list;
    const int4 *node_info = a.node_info;
    const int32_t *E_idx = a.E_idx;
    const float *xsrc = HOISTED ? a.E1 : a.hE_in;    // layer-1 edge operand: hoisted term or h_E
    const float *P = a.P, *Q = a.Q, *b2 = a.b2;
    float *S = a.S;
    const void *W1h = a.W1h, *W2h = a.W2h;
    int n_tiles = a.n_tiles, n_nodes = a.n_nodes, in_by_src = a.in_by_src;
    GeluK gelu_a = a.gelu_a, gelu_b = a.gelu_b;
    PIN_PTR(tile_list); PIN_PTR(node_info); PIN_PTR(E_idx); PIN_PTR(xsrc); PIN_PTR(P); PIN_PTR(Q); PIN_PTR(b2); PIN_PTR(S);
    PIN_PTR(W1h); PIN_PTR(W2h); PIN1(n_tiles); PIN1(n_nodes); PIN1(in_by_src);
    pin_gelu(gelu_a); pin_gelu(gelu_b);
    u32x4 *fragX = wl + EW_FRAG_X, *fragA = wl + EW_FRAG_A;
    const int tid = threadIdx.x, lane = tid & 63, bo = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    BlockQuarter w1, w2;
    if (!HOISTED) w1.start(W1h, bo, lane);
    w2.start(W2h, bo, lane);
    const float bias = b2[32 * bo + c];
    int grid = gridDim.x;
    PIN1(grid);
    for (int t = blockIdx.x; t < n_tiles; t += grid) {
        const int2 tn = scalar_load(tile_list + t);
        const int4 info = scalar_load(node_info + tn.x);
        const int n = tn.x, half = tn.y, src = info.x, base = info.y, K = info.z;
        const int colc = (32 * half + c < K) ? 32 * half + c : 0;
        const int j = E_idx[(size_t)src * 64 + colc];
        f32x16 xq, acc, pq;
        edge_quarter_load(xq, xsrc + (size_t)((HOISTED || in_by_src) ? src : n) * EDGE_BLOCK, colc, bo, h);
        quarter_load(acc, Q + (size_t)(base + j) * HD, bo, h);
        quarter_load(pq, P + (size_t)n * HD, bo, h);
        if (HOISTED) {
            acc += pq;
            acc += xq;
        } else {
            publish_stored_quarter(fragX, xq, bo, lane);
            acc += pq;
            __syncthreads();
            w1.run<TERMS>(acc, fragX, lane);                                   // layer 1 (pre-split h_E tile)
        }
        publish_quarter<true>(fragA, acc, bo, lane, gelu_a);
        f32x16 t2;
        {
            float bv = bias;
            asm volatile("" : "+v"(bv));
#pragma unroll
            for (int r = 0; r < 16; ++r) t2[r] = bv;
        }
        __syncthreads();
        w2.run<TERMS, true>(t2, fragA, lane);                                  // layer 2 on GELU(layer 1), transposed
#pragma unroll
        for (int r = 0; r < 16; r += 8) {
            f32x2 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = f32x2{t2[r + 2 * i], t2[r + 2 * i + 1]};
            gelu_pairs<4>(v, gelu_b);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t2[r + 2 * i] = v[i].x;
                t2[r + 2 * i + 1] = v[i].y;
            }
        }
        const int cnt = K - 32 * half;
        float s0 = 0.f;
        if (cnt >= 32) {
            f32x2 s2 = tile_pair(t2, 0);
#pragma unroll
            for (int r = 2; r < 16; r += 2) s2 += tile_pair(t2, r);
            s0 = s2.x + s2.y;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s0 += ((r & 3) + 8 * (r >> 2) + 4 * h < cnt) ? t2[r] : 0.f;
        }
        S[((size_t)(half + 2 * h) * n_nodes + n) * HD + 32 * bo + c] = s0;
        // HOISTED has no barrier between the fragment write above and the next tile's: the slowest wave may still be
        // reading this tile's fragments
        if (HOISTED) __syncthreads();
    }
}

template <int TERMS>
void launch_wide_h(bool update, const EdgeArgs &ea, const int2 *tile_list, int n_tiles, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        set_max_lds(reinterpret_cast<const void *>(upd_wide_kernel<false, TERMS>), EW_LDS_BYTES);
        set_max_lds(reinterpret_cast<const void *>(upd_wide_kernel<true, TERMS>), EW_LDS_BYTES);
        attr_set = true;
    }
    const bool hoisted = ea.E1 != nullptr;
    EdgeTileArgs ta;
    static_cast<EdgeArgs &>(ta) = ea;
    ta.tile_list = tile_list; ta.n_tiles = n_tiles;
    const int cap = 2 * num_cu();                    // two workgroups (one wave per SIMD each) per CU
    dim3 grid(n_tiles < cap ? n_tiles : cap), block(EW_WAVES * 64);
    const size_t lds_msg = 16 * (size_t)EW_FRAG_B;
    if (update && hoisted) hipLaunchKernelGGL((upd_wide_kernel<true, TERMS>), grid, block, EW_LDS_BYTES, st, ta);
    else if (update) hipLaunchKernelGGL((upd_wide_kernel<false, TERMS>), grid, block, EW_LDS_BYTES, st, ta);
    else if (hoisted) hipLaunchKernelGGL((msg_wide_kernel<true, TERMS>), grid, block, lds_msg, st, ta);
    else hipLaunchKernelGGL((msg_wide_kernel<false, TERMS>), grid, block, lds_msg, st, ta);
}

}  // namespace

void launch_edge_wide(int terms, bool update, const EdgeArgs &ea, const int2 *tile_list, int n_tiles, hipStream_t st) {
    if (terms == 3) launch_wide_h<3>(update, ea, tile_list, n_tiles, st);
    else launch_wide_h<4>(update, ea, tile_list, n_tiles, st);
}
