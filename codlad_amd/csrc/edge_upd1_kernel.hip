// Edge update of the split-fp16 modes, ONE wave per SIMD (round 4).  Own translation unit (edge_args.h says why).
//
// upd_kernel_h (edge_upd_kernel.hip) runs two waves per SIMD at the 256 registers that allows: three register tiles
// are live, nothing is left to prefetch into, and five of the eight k-steps of W11e stream from L2 for every tile -
// 88 vector-memory instructions per 32-edge tile whose latency is exposed (profiles/r02_final_ablation.txt: the kernel
// takes exactly as long with its arithmetic removed).  Here a workgroup is four waves, each alone on its SIMD with the
// whole 512-register file:
//   * the k-steps of W11e that do not fit the LDS (5 of 8) are RESIDENT IN REGISTERS (160 per lane, loaded once per
//     launch, used as the MFMA A operand directly): no streamed fragment at all, 48 instead of 88 vector-memory
//     instructions per tile;
//   * the next tile's edge rows are requested right after layer 1 into a second register tile, its gathered Q rows
//     right after layer 2 into the (then free) accumulator: both travel under the remaining contractions, and the
//     dependent look-ups that open a node (table entry -> neighbour list -> rows) run a whole node ahead;
//   * same arithmetic in the same order as upd_kernel_h and the tile kernels: bit-identical results.
#include "edge_args.h"

#ifndef U1_ABLATE
#define U1_ABLATE 0     // timing experiments (results wrong on purpose): 1 no arithmetic, 2 no tile loads, 4 no Q gather, 8 no store, 16 every lane the same Q row
#endif
// -DU1_STAMP: diagnostic build - per wave, shader cycles spent between the marks below, summed over its tiles, written to
// a.S (the bench launch helper passes the workspace's S for this): [wave][16] floats.  Not for production.
#ifdef U1_STAMP
#define U1_MARK(k) do { const long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define U1_MARK(k) do { } while (0)
#endif
#ifndef U1_FUSED_RESIDUAL
#define U1_FUSED_RESIDUAL 1
#endif
#ifndef U1_AHEAD
#define U1_AHEAD 1        // LDS fragment reads run this many groups ahead in layers 2 and 3 (common.h gemm_h_lds)
#endif
constexpr int U1_NW = 4;                                   // waves per workgroup = SIMDs per CU
constexpr int U1_REG_KS = 8 - UPD_W1_KS;                   // k-steps of W11e held in registers

// acc += W[ks KS0 .. KS0+NKS) @ in, weight fragments from registers (wr[ks - KS0][bo][hi, lo]); `in` is a stored
// (pre-split) tile.  Same group order as gemm_h_lds / StreamedGemm::run.
template <int TERMS, int KS0, int NKS>
DEV void gemm_h_reg(Tile &acc, const Tile &in, const u32x4 (&wr)[NKS][4][2]) {
#pragma unroll
    for (int k = 0; k < NKS; ++k) {
        SplitFrag x;
        presplit_frag(x, in, KS0 + k);
#pragma unroll
        for (int bo = 0; bo < 4; ++bo)
            mfma_f16<TERMS>(acc.b[bo], as_f16x8(wr[k][bo][0]), as_f16x8(wr[k][bo][1]), x);
    }
}

// A stored (pre-split) edge tile as it arrives from memory, kept in the ACCUMULATOR half of the register file: sixteen
// 16-byte quads, q[b][2 s] / q[b][2 s + 1] = the hi / lo B-operand fragments of k-step 2 b + s (common.h, "pre-split edge
// state").  The quads are loaded by inline asm with the "a" constraint: the only consumers are matrix instructions (which
// read AGPRs directly) and one pass of v_accvgpr_read when the residual is formed, and the 256 architectural registers are
// needed for the three tiles the vector pipe works on.  The compiler does not know these loads (no s_waitcnt is generated
// for them): see `wait_rows` at the point of use.
struct QTile {
    f32x4 q[4][4];
};

template <bool STREAM>
DEV void qtile_request(QTile &t, const float *block, int e, int h) {
    const char *p = reinterpret_cast<const char *>(reinterpret_cast<const float4 *>(block) + EDGE_F4(e) + h * 32);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) {
        const char *pb = p + bo * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (STREAM) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=a"(t.q[bo][q]) : "v"(pb), "n"(q * 1024) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(t.q[bo][q]) : "v"(pb), "n"(q * 1024) : "memory");
        }
    }
}

DEV void qtile_frag(SplitFrag &f, const QTile &in, int ks) {
    f.hi = __builtin_bit_cast(f16x8, in.q[ks >> 1][2 * (ks & 1)]);
    f.lo = __builtin_bit_cast(f16x8, in.q[ks >> 1][2 * (ks & 1) + 1]);
}

// gemm_h_lds<TERMS, KS0, NKS, false, false, true> with the stored tile in a QTile (same group order, same ring)
template <int TERMS, int KS0, int NKS>
DEV void gemm_q_lds(Tile &acc, const QTile &in, const u32x4 *wl, int lane) {
    const u32x4 *w = wl + lane;
    constexpr int G0 = KS0 * 4, NG = NKS * 4, AHEAD = U1_AHEAD, R = AHEAD + 1;
    u32x4 ring[R][2];
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) {
        ring[g][0] = w[((G0 + g) * 2 + 0) * 64];
        ring[g][1] = w[((G0 + g) * 2 + 1) * 64];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int ks = KS0 + (g >> 2), bo = g & 3;
        if (g + AHEAD < NG) {
            ring[(g + AHEAD) % R][0] = w[((G0 + g + AHEAD) * 2 + 0) * 64];
            ring[(g + AHEAD) % R][1] = w[((G0 + g + AHEAD) * 2 + 1) * 64];
        }
        SplitFrag x;
        qtile_frag(x, in, ks);
        mfma_f16<TERMS>(acc.b[bo], as_f16x8(ring[g % R][0]), as_f16x8(ring[g % R][1]), x);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc += W[ks KS0 .. KS0+NKS) @ in, weight fragments from registers (wr[ks - KS0][bo][hi, lo])
template <int TERMS, int KS0, int NKS>
DEV void gemm_q_reg(Tile &acc, const QTile &in, const u32x4 (&wr)[NKS][4][2]) {
#pragma unroll
    for (int k = 0; k < NKS; ++k) {
        SplitFrag x;
        qtile_frag(x, in, KS0 + k);
#pragma unroll
        for (int bo = 0; bo < 4; ++bo)
            mfma_f16<TERMS>(acc.b[bo], as_f16x8(wr[k][bo][0]), as_f16x8(wr[k][bo][1]), x);
    }
}

// tile_unsplit_scale_add_row with the stored tile in a QTile: t = (hi + lo) * scale + row
DEV void qtile_unsplit_scale_add_row(Tile &t, const QTile &in, float scale, const float *row, int h) {
    const float4 *pr = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int bo = 0; bo < 4; ++bo)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 hb = __builtin_bit_cast(u32x4, in.q[bo][2 * s]), lb = __builtin_bit_cast(u32x4, in.q[bo][2 * s + 1]);
            float val[8];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
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

// Layer 1 of a tile (acc += W11e @ xn: k-steps 0 .. UPD_W1_KS - 1 from LDS, the rest from registers, in the group order of
// gemm_h_lds / StreamedGemm::run) fused with the residual of layer 3 (x = (hi + lo) * scale + row, the arithmetic of
// tile_unsplit_scale_add_row): every group of matrix instructions is followed by the two residual elements that come out of
// the SAME quads the group's B fragments are (two v_accvgpr_read, two v_fma_mix, two v_fma), so that the residual rides in
// the shadow of the matrix instructions and the stored tile is never copied to architectural registers as a whole.
template <int TERMS>
DEV void layer1_residual(Tile &acc, Tile &x, const QTile &xn, const u32x4 *wl, const u32x4 (&wr)[U1_REG_KS][4][2], int lane,
                         float scale, const float *row, int h) {
    const u32x4 *w = wl + lane;
    constexpr int NG = UPD_W1_KS * 4, AHEAD = U1_AHEAD, R = AHEAD + 1;
    const float4 *pr = reinterpret_cast<const float4 *>(row);
    u32x4 ring[R][2];
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) {
        ring[g][0] = w[(g * 2 + 0) * 64];
        ring[g][1] = w[(g * 2 + 1) * 64];
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int b = ks >> 1, s = ks & 1;
        // this k-step's two quads, taken out of the accumulator file HERE (four v_accvgpr_read each) and used from
        // architectural registers by the matrix instructions and by the residual alike
        f32x4 hq = xn.q[b][2 * s], lq = xn.q[b][2 * s + 1];
        asm volatile("" : "+v"(hq), "+v"(lq));
        SplitFrag f;
        f.hi = __builtin_bit_cast(f16x8, hq);
        f.lo = __builtin_bit_cast(f16x8, lq);
        const float4 ra = pr[8 * b + 4 * s + h], rb = pr[8 * b + 4 * s + 2 + h];
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) {
            const int g = ks * 4 + bo;
            if (ks < UPD_W1_KS) {
                if (g + AHEAD < NG) {
                    ring[(g + AHEAD) % R][0] = w[((g + AHEAD) * 2 + 0) * 64];
                    ring[(g + AHEAD) % R][1] = w[((g + AHEAD) * 2 + 1) * 64];
                }
                mfma_f16<TERMS>(acc.b[bo], as_f16x8(ring[g % R][0]), as_f16x8(ring[g % R][1]), f);
            } else {
                mfma_f16<TERMS>(acc.b[bo], as_f16x8(wr[ks - UPD_W1_KS][bo][0]), as_f16x8(wr[ks - UPD_W1_KS][bo][1]), f);
            }
            {   // residual elements 2 bo, 2 bo + 1 of this k-step's eight
                const int p = bo;
                const unsigned hb = __builtin_bit_cast(u32x4, hq)[p], lb = __builtin_bit_cast(u32x4, lq)[p];
                float v0, v1;
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(v0) : "v"(hb), "v"(lb));
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(v1) : "v"(hb), "v"(lb));
                const float4 r = p < 2 ? ra : rb;
                x.b[b][8 * s + 2 * p] = fmaf(v0, scale, (p & 1) ? r.z : r.x);
                x.b[b][8 * s + 2 * p + 1] = fmaf(v1, scale, (p & 1) ? r.w : r.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <bool HOISTED, int TERMS>
__global__ __launch_bounds__(U1_NW * 64, 1) void upd1_kernel_h(EdgeArgs a) {
    extern __shared__ __align__(16) u32x4 wl[];
    constexpr int NT = U1_NW * 64;
    constexpr int W1_U4 = UPD_W1_KS * 512;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 *consts = wl + 2 * LDS_BLOCK_U4 + W1_U4;
    {
        const u32x4 *g1 = reinterpret_cast<const u32x4 *>(a.W1h);
        const u32x4 *g2 = reinterpret_cast<const u32x4 *>(a.W2h);
        const u32x4 *g3 = reinterpret_cast<const u32x4 *>(a.W3h);
        for (int i = tid; i < LDS_BLOCK_U4; i += NT) {
            wl[i] = g2[i];
            wl[LDS_BLOCK_U4 + i] = g3[i];
        }
        if (!HOISTED)
            for (int i = tid; i < W1_U4; i += NT) wl[2 * LDS_BLOCK_U4 + i] = g1[i];
        if (tid < 32) consts[tid] = reinterpret_cast<const u32x4 *>(a.b2)[tid];
        if (tid >= 64 && tid < 96) consts[32 + (tid & 31)] = reinterpret_cast<const u32x4 *>(a.b3)[tid & 31];
        if (tid >= 128 && tid < 160) {
            // modulate folded to one multiply-add: A = gate (1 + scale), B = gate shift
            const float4 *m = reinterpret_cast<const float4 *>(a.mods3);
            const int i = tid & 31;
            const float4 s = m[i], c = m[32 + i], g = m[64 + i];
            float4 *cf = reinterpret_cast<float4 *>(consts);
            cf[64 + i] = make_float4(g.x * (1.0f + c.x), g.y * (1.0f + c.y), g.z * (1.0f + c.z), g.w * (1.0f + c.w));
            cf[96 + i] = make_float4(g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w);
        }
    }
    // the register-resident part of W11e: k-steps UPD_W1_KS .. 7, this lane's 16 bytes of every fragment
    u32x4 wr[U1_REG_KS][4][2];
    if (!HOISTED) {
        const u32x4 *g1 = reinterpret_cast<const u32x4 *>(a.W1h) + lane;
#pragma unroll
        for (int k = 0; k < U1_REG_KS; ++k)
#pragma unroll
            for (int bo = 0; bo < 4; ++bo)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // loaded straight into the accumulator half of the register file ("a" constraint): these 160 registers
                    // are only ever read by matrix instructions, and the 256 architectural registers are all needed for
                    // the tiles the vector pipe works on.  (Left to itself hipcc keeps them in VGPR class and copies
                    // every fragment back with four v_accvgpr_read per use.)
                    const u32x4 *p = g1 + ((((UPD_W1_KS + k) * 4 + bo) * 2) + s) * 64;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wr[k][bo][s]) : "v"(p) : "memory");
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const u32x4 *w2 = wl, *w3 = wl + LDS_BLOCK_U4, *w1 = wl + 2 * LDS_BLOCK_U4;
    const float *c_base = reinterpret_cast<const float *>(consts);
    float2 *Pslots = reinterpret_cast<float2 *>(consts + EDGE_CONST_U4 + wave * 64);       // two slots of 512 bytes per wave
    const int h = lane >> 5, c = lane & 31;
    // (wave-uniform, and told so: everything the walk derives from these then lives in scalar registers)
    const NodeSpan span0 = wave_node_span(a.n_nodes, U1_NW, wave);
    const NodeSpan span = {__builtin_amdgcn_readfirstlane(span0.first), __builtin_amdgcn_readfirstlane(span0.end),
                           __builtin_amdgcn_readfirstlane(span0.stride)};
    const int stride = span.stride, n_end = span.end;
    if (span.first >= n_end) return;
    const float2 *Prows = reinterpret_cast<const float2 *>(a.P);
    const float *E1 = a.E1;
    auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto rows_of = [&](int node, int s) { return a.hE_in + (size_t)(a.in_by_src ? s : node) * EDGE_BLOCK; };

    // ---- the tile sequence of this wave, looked up ahead of the arithmetic -----------------------------------------
    // A wave walks nodes first, first + stride, ... and gives EVERY node two tiles (columns 0-31 and 32-63).  A node with
    // K <= 32 has nothing in its second tile: its lanes then redo column 0 and store into the padding of the node's block
    // (columns >= K, which nothing reads) - wasted work, which is why the launcher sends jobs with many such nodes to
    // upd_kernel_h.  What that buys: one fixed sequence of vector-memory instructions per tile and no branch around any
    // of them.  Per tile (the loop is rotated: its header sits between layers 2 and 3):
    //     layer 3 | 16 Q-row requests for the next tile -> acc | LayerNorm | 16 stores |
    //     4 small loads of the walk | P + Q | layer 1, residual | 16 row requests for the tile after -> xn | layer 2
    // Everything the compiler counts is requested and used inside one iteration (the Q rows behind the 16 stores and the 4
    // small loads: s_waitcnt vmcnt(20 + ...), never a wait for the stores themselves; the small loads are used right after
    // layer 1), so its s_waitcnt counts are exact.  (With a conditional load or store in the loop, or requests in flight
    // across the back edge, hipcc fell back to s_waitcnt vmcnt(0) at tile boundaries: no prefetch at all.)  The edge rows
    // (xn) are requested by inline asm the compiler does not count; they are older than the Q rows of the same tile, so
    // the wait for those covers them.
    const int n_last = n_end - 1 - ((n_end - 1 - span.first) % stride);        // last node of the walk
    auto next_node = [&](int n) { return n + stride <= n_last ? n + stride : n_last; };
    // `look` = (node ln, half lh): the tile whose rows are requested next; R1 = the node after ln
    int ln = span.first, lsrc, lbase, lK, lh = 0;
    int r1n = next_node(ln), r1src, r1base, r1K;
    {
        const int4 i0 = a.node_info[ln], i1 = a.node_info[r1n];
        lsrc = rfl(i0.x); lbase = rfl(i0.y); lK = rfl(i0.z);
        r1src = rfl(i1.x); r1base = rfl(i1.y); r1K = rfl(i1.z);
    }
    int ljA = a.E_idx[(size_t)lsrc * 64 + (c < lK ? c : 0)];                  // look's neighbours for columns c and 32 + c
    int ljB = a.E_idx[(size_t)lsrc * 64 + (32 + c < lK ? 32 + c : 0)];
    float2 lpv = Prows[(size_t)ln * 64 + lane];                               // its P row, as loaded (two floats per lane)
    int n2 = next_node(r1n);
    int look_par = 0;                                                         // parity of look's node (its P slot)
    int tiles_left = 2 * ((n_last - span.first) / stride + 1);                // tiles whose TAIL is still to run

#ifdef U1_STAMP
    long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    const long long tstart = tprev;
#endif
    Tile x, acc, t2, e1n;
    QTile xn;
    // request the stored rows of `look` -> xn (hoisted term -> e1n)
    auto issue_rows = [&]() {
        const int e = 32 * lh + c;
        const int ec = e < lK ? e : 0;
        if (HOISTED) {
            qtile_request<false>(xn, rows_of(ln, lsrc), ec, h);
            tile_load_edge<false>(e1n, E1 + (size_t)lsrc * EDGE_BLOCK, ec, h);
        } else if (!(U1_ABLATE & 2)) {
            qtile_request<true>(xn, rows_of(ln, lsrc), ec, h);
        }
    };
    // ... its gathered Q rows -> acc, and its P row into the slot of its node's parity
    auto issue_q = [&]() {
        const int pq = lbase + (lh ? ljB : ljA);
        if (!(U1_ABLATE & 4))
            tile_load_row(acc, a.Q + (size_t)((U1_ABLATE & 16) ? rfl(pq) : pq) * HD, h);
        Pslots[look_par * 64 + lane] = lpv;
    };
    // the walk's own loads: R1's neighbour-list entries and P row, the table entry of the node after R1
    struct WalkLoads { int jA, jB; float2 pv; int4 info; };
    auto walk_issue = [&]() {
        WalkLoads w;
        w.jA = a.E_idx[(size_t)r1src * 64 + (c < r1K ? c : 0)];
        w.jB = a.E_idx[(size_t)r1src * 64 + (32 + c < r1K ? 32 + c : 0)];
        w.pv = Prows[(size_t)r1n * 64 + lane];
        w.info = a.node_info[n2];
        return w;
    };
    auto walk_pin = [&](WalkLoads &w) {     // the loaded values are taken here, used or not: one s_waitcnt in straight-line code
        asm volatile("" : "+v"(w.jA), "+v"(w.jB), "+v"(w.pv.x), "+v"(w.pv.y), "+v"(w.info.x), "+v"(w.info.y), "+v"(w.info.z),
                     "+v"(w.info.w));      // .w too: a dead element's register is reused at once, behind an s_waitcnt vmcnt(0)
    };
    auto walk_commit = [&](const WalkLoads &w) {          // move `look` one tile on (selects only, no memory access)
        const bool nn = lh == 1;                          // second tile done: on to the next node
        const int s2 = rfl(w.info.x), b2 = rfl(w.info.y), k2 = rfl(w.info.z);
        ljA = nn ? w.jA : ljA; ljB = nn ? w.jB : ljB;
        lpv.x = nn ? w.pv.x : lpv.x; lpv.y = nn ? w.pv.y : lpv.y;
        ln = nn ? r1n : ln; lsrc = nn ? r1src : lsrc; lbase = nn ? r1base : lbase; lK = nn ? r1K : lK;
        r1n = nn ? n2 : r1n; r1src = nn ? s2 : r1src; r1base = nn ? b2 : r1base; r1K = nn ? k2 : r1K;
        n2 = nn ? next_node(n2) : n2;
        look_par ^= nn ? 1 : 0;
        lh ^= 1;
    };

    // HEAD of a tile whose node has parity `parity` (acc holds its Q rows, xn its stored rows, e1n its hoisted term, its
    // P row is in slot `parity`): P + Q, layer 1, the residual -> x, layer 2 -> t2.  After layer 1 the stored rows of
    // `look` are requested (xn is free then) and the walk's loads, issued at the start, are taken.
    auto head = [&](int parity, WalkLoads &w) {
        int lds_off = 0;
        asm volatile("" : "+v"(lds_off));       // keeps the constant reads inside the loop (see upd_kernel_h)
        const float *c_b2 = c_base + lds_off, *c_b3 = c_b2 + HD;
        U1_MARK(0);                 // advance / loop overhead
        w = walk_issue();
        tile_add_row(acc, reinterpret_cast<const float *>(Pslots + parity * 64) + lds_off, h);
        // the tile's Q rows have been used, so its stored rows (requested before them) are in xn; nothing that reads xn
        // may move above this point
        __builtin_amdgcn_sched_barrier(0);
        U1_MARK(1);                 // P + Q (the wait for the tile's rows)
        // layer 1; the residual enters layer 3's accumulator as (hi + lo) * 2^E + b13 * 2^E
        if (HOISTED) {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) acc.b[bo] += e1n.b[bo];
            qtile_unsplit_scale_add_row(x, xn, a.res_scale, c_b3, h);
        } else if (!(U1_ABLATE & 1)) {
#if U1_FUSED_RESIDUAL
            layer1_residual<TERMS>(acc, x, xn, w1, wr, lane, a.res_scale, c_b3, h);
#else
            gemm_q_lds<TERMS, 0, UPD_W1_KS>(acc, xn, w1, lane);                                   // layer 1, k-steps in LDS
            gemm_q_reg<TERMS, UPD_W1_KS, U1_REG_KS>(acc, xn, wr);                                 // layer 1, k-steps in registers
            __builtin_amdgcn_sched_barrier(0);
            qtile_unsplit_scale_add_row(x, xn, a.res_scale, c_b3, h);
#endif
        } else {
            qtile_unsplit_scale_add_row(x, xn, a.res_scale, c_b3, h);
        }
        __builtin_amdgcn_sched_barrier(0);
        U1_MARK(2);                 // layer 1 + residual
        walk_pin(w);
        issue_rows();
        __builtin_amdgcn_sched_barrier(0);
        U1_MARK(3);                 // walk loads taken, rows requested
        tile_load_row(t2, c_b2, h);
        if (!(U1_ABLATE & 1)) gemm128_h_lds<TERMS, true, false, false, U1_AHEAD>(t2, acc, w2, lane, a.gelu_a);   // layer 2 on GELU(layer 1)
        else { for (int bo = 0; bo < 4; ++bo) t2.b[bo] += acc.b[bo]; }
        __builtin_amdgcn_sched_barrier(0);
        U1_MARK(4);                 // layer 2
    };

    // prologue: tile 0's rows, then its HEAD with `look` = tile 1
    int dn = ln, dh = 0;                 // the tile whose TAIL runs next
    int node_par = 0;                    // parity of its node
    WalkLoads w = walk_issue();
    issue_rows();
    issue_q();
    walk_pin(w);
    walk_commit(w);                      // look = (first, 1)
    head(0, w);                          // requests look's stored rows; w = what the next commit needs
    for (;;) {
        // TAIL of tile (dn, dh): layer 3, LayerNorm, store; the next tile's Q rows are requested after layer 3 (acc is free)
        {
            int lds_off = 0;
            asm volatile("" : "+v"(lds_off));
            const float *c_modA = c_base + lds_off + 2 * HD, *c_modB = c_modA + HD;
            if (!(U1_ABLATE & 1)) gemm128_h_lds<TERMS, true, false, false, U1_AHEAD>(x, t2, w3, lane, a.gelu_b);   // layer 3 on GELU(layer 2)
            else { for (int bo = 0; bo < 4; ++bo) x.b[bo] += t2.b[bo]; }
            __builtin_amdgcn_sched_barrier(0);
            U1_MARK(5);             // layer 3
            issue_q();
            __builtin_amdgcn_sched_barrier(0);
            U1_MARK(6);             // Q rows requested
            if (!(U1_ABLATE & 1)) {
                tile_layernorm_affine(x, a.ln_eps, c_modA, c_modB, h);
                tile_presplit(x);
            }
            // every lane stores: a column beyond K (a copy of column 0's arithmetic) lands in the padding of this node's block
            if (!(U1_ABLATE & 8) || x.b[0][0] == 12345.f)
                tile_store_edge<true>(x, a.hE_out + (size_t)dn * EDGE_BLOCK, 32 * dh + c, h);
            U1_MARK(7);             // LayerNorm, split, stores issued
        }
        if (--tiles_left == 0) break;
        walk_commit(w);                  // look = the tile after the one entering HEAD
        const bool new_node = dh == 1;
        dn = new_node ? next_node(dn) : dn;
        dh ^= 1;
        node_par ^= new_node ? 1 : 0;
        head(node_par, w);
    }
#ifdef U1_STAMP
    if (a.S != nullptr && lane == 0) {
        float *o = a.S + ((size_t)blockIdx.x * U1_NW + wave) * 16;
        for (int k = 0; k < 8; ++k) o[k] = (float)ph[k];
        o[8] = (float)(__builtin_amdgcn_s_memtime() - tstart);
        o[9] = (float)(2 * ((n_last - span.first) / stride + 1));
    }
#endif
}

template <int TERMS>
static void launch_upd1_t(const EdgeArgs &ea, hipStream_t st) {
    static bool attr_set = false;     // one flag per TERMS instantiation
    constexpr size_t lds = 16 * (edge_lds_u4<true, U1_NW>() + U1_NW * 32);      // two P slots per wave
    static_assert(lds <= 160 * 1024, "kernel exceeds the CU's LDS");
    if (!attr_set) {
        set_max_lds(reinterpret_cast<const void *>(upd1_kernel_h<false, TERMS>), lds);
        set_max_lds(reinterpret_cast<const void *>(upd1_kernel_h<true, TERMS>), lds);
        attr_set = true;
    }
    const int groups = (ea.n_nodes + U1_NW - 1) / U1_NW;
    dim3 grid(groups < edge_cus() ? groups : edge_cus()), block(U1_NW * 64);
    if (ea.E1 != nullptr) hipLaunchKernelGGL((upd1_kernel_h<true, TERMS>), grid, block, lds, st, ea);
    else hipLaunchKernelGGL((upd1_kernel_h<false, TERMS>), grid, block, lds, st, ea);
}

void launch_edge_upd1(int terms, const EdgeArgs &ea, hipStream_t st) {
    if (terms == 3) launch_upd1_t<3>(ea, st);
    else launch_upd1_t<4>(ea, st);
}
