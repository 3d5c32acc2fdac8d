// Small-job variants of the two split-fp16 edge kernels (their own translation unit, see edge_args.h).
#include "edge_args.h"

// ---------------------------------------------------------------------------------------------
// The same pipelines as upd_kernel_h / msg_kernel_h (denoiser_kernels.hip), with the work dealt out per
// non-empty 32-edge tile {node, half} of the job's tile list instead of per node (twice the busy waves when the
// job has fewer nodes than the grid has waves); the message kernel then writes one partial neighbour sum per
// half, S[half][n_nodes][128], which the node kernel adds up.  Kept as separate kernels with their own argument
// struct: the per-node kernels above sit exactly at the 256-register limit and any change to their signature
// or body shape makes hipcc spill in their inner loops (-9 % measured).
// ---------------------------------------------------------------------------------------------

// Edge update, per tile:
//   h_E[n,j] <- mod3(LN(h_E[n,j] + W13 GELU(W12 GELU(P_i + Q_j + W11e h_E[n,j]) + b12) + b13))
// HOISTED: encoder layer 0 with a.E1 given - layer 1's edge contraction comes precomputed.
template <int NWAVES, bool HOISTED, int TERMS>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void upd_tile_kernel_h(EdgeTileArgs a) {
    constexpr bool TILEWISE = true;
    extern __shared__ __align__(16) u32x4 wl[];
    constexpr int NT = NWAVES * 64;
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
    __syncthreads();
    const u32x4 *w2 = wl, *w3 = wl + LDS_BLOCK_U4, *w1 = wl + 2 * LDS_BLOCK_U4;
    const float *c_base = reinterpret_cast<const float *>(consts);
    float *Pslot = reinterpret_cast<float *>(consts + EDGE_CONST_U4 + wave * 32);
    const int h = lane >> 5, c = lane & 31;
    // one 32-edge tile: half `half` of node n
    auto one_tile = [&](int n, int src, int base, int K, int half) {
        const float *rows = a.hE_in + (size_t)(a.in_by_src ? src : n) * EDGE_BLOCK;
        float *out_rows = a.hE_out + (size_t)n * EDGE_BLOCK;
        const int col = 32 * half + c;
        const bool valid = col < K;
        const int colc = valid ? col : 0;
        const int j = a.E_idx[(size_t)src * 64 + colc];
        Tile x, acc, t2;
        StreamedGemm<TERMS, UPD_W1_KS, 8 - UPD_W1_KS, false, 8, true> tail1;
        if (!HOISTED) tail1.start(a.W1h, lane);
        // the constants never change, so the compiler would read them once, before the node
        // loop, into ~300 registers and spill those; an opaque zero offset keeps the reads here
        int lds_off = 0;
        asm volatile("" : "+v"(lds_off));
        const float *c_b2 = c_base + lds_off, *c_b3 = c_b2 + HD;
        const float *c_modA = c_b2 + 2 * HD, *c_modB = c_b2 + 3 * HD;
        tile_load_row(acc, a.Q + (size_t)(base + j) * HD, h);
        tile_add_row(acc, Pslot, h);
        tile_load_edge(x, rows, colc, h);                          // layer-1 operand and residual
        if (HOISTED) {
            tile_add_edge(acc, a.E1 + (size_t)src * EDGE_BLOCK, colc, h);
        } else {
            gemm_h_lds<TERMS, 0, UPD_W1_KS, false, false, true>(acc, x, w1, lane, a.gelu_a);   // layer 1, resident k-steps
            tail1.run(acc, x, lane, a.gelu_a);                                    // layer 1, streamed k-steps
        }
        tile_load_row(t2, c_b2, h);
        gemm128_h_lds<TERMS, true>(t2, acc, w2, lane, a.gelu_a);   // layer 2 on GELU(layer 1)
        // layer 3 accumulates onto (h_E + b13) * 2^E: the input tile stays in registers for the
        // residual instead of being fetched from HBM a second time (c_b3 holds b13 * 2^E)
        tile_unsplit_scale_add_row(x, a.res_scale, c_b3, h);       // x arrives pre-split: its value is hi + lo
        gemm128_h_lds<TERMS, true>(x, t2, w3, lane, a.gelu_b);     // layer 3 on GELU(layer 2)
        tile_layernorm_affine(x, a.ln_eps, c_modA, c_modB, h);
        tile_presplit(x);
        if (valid) tile_store_edge(x, out_rows, col, h);
    };
    // this node's P row: one coalesced 512-byte read, staged in the wave's own LDS slot
    // (same wave writes and reads: program order + lgkmcnt, no barrier)
    auto stage_P = [&](int n) {
        reinterpret_cast<float2 *>(Pslot)[lane] = reinterpret_cast<const float2 *>(a.P + (size_t)n * HD)[lane];
    };
    if constexpr (TILEWISE) {
        for (int t = blockIdx.x * NWAVES + wave; t < a.n_tiles; t += gridDim.x * NWAVES) {
            const int2 tn = a.tile_list[t];
            const int4 info = a.node_info[tn.x];
            stage_P(tn.x);
            one_tile(tn.x, info.x, info.y, info.z, tn.y);
        }
    } else {
        const NodeSpan span = wave_node_span(a.n_nodes, NWAVES, wave);
        for (int n = span.first; n < span.end; n += span.stride) {
            const int4 info = a.node_info[n];
            stage_P(n);
            for (int half = 0; half < 2; ++half) {
                if (32 * half >= info.z) break;
                one_tile(n, info.x, info.y, info.z, half);
            }
        }
    }
}

// Message kernel, per tile: S[n] = sum_j GELU(W2 GELU(P_i + Q_j + W1e h_E[i,j]) + b2) over the K
// neighbours.  Same LDS residency scheme as upd_kernel_h; on top of that
//   * the LAST contraction is issued with swapped MFMA operands, so its output block arrives
//     transposed (lane = feature, registers = the tile's 32 edges): the sum over neighbours is
//     then 15 register adds per block instead of a 5-step cross-lane reduction per register;
//   * with the 64-register running sum gone, the next tile's edge rows are fetched while layer 2
//     runs and its Q rows while the epilogue runs (a wave walks its (node, half) tiles in order).
template <int NWAVES, bool HOISTED, int TERMS>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void msg_tile_kernel_h(EdgeTileArgs a) {
    constexpr bool TILEWISE = true;
    extern __shared__ __align__(16) u32x4 wl[];
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 *consts = wl + 2 * LDS_BLOCK_U4;
    {
        const u32x4 *g1 = reinterpret_cast<const u32x4 *>(a.W1h);
        const u32x4 *g2 = reinterpret_cast<const u32x4 *>(a.W2h);
        for (int i = tid; i < LDS_BLOCK_U4; i += NT) {
            if (!HOISTED) wl[i] = g1[i];
            wl[LDS_BLOCK_U4 + i] = g2[i];
        }
        if (tid < 32) consts[tid] = reinterpret_cast<const u32x4 *>(a.b2)[tid];
    }
    __syncthreads();
    const u32x4 *w1 = wl, *w2 = wl + LDS_BLOCK_U4;
    const float *c_b2 = reinterpret_cast<const float *>(consts);
    float2 *Pslot = reinterpret_cast<float2 *>(consts + EDGE_CONST_U4 + wave * 32);
    const int h = lane >> 5, c = lane & 31;
    const float *xsrc = HOISTED ? a.E1 : a.hE_in;    // layer-1 edge operand: hoisted term or h_E
    auto block_of = [&](int node, int s) {
        return xsrc + (size_t)((HOISTED || a.in_by_src) ? s : node) * EDGE_BLOCK;
    };
    const float2 *Prows = reinterpret_cast<const float2 *>(a.P);

    if constexpr (TILEWISE) {
        // small jobs: a wave walks non-empty (node, half) tiles of the tile list; each tile writes its own
        // partial sum S[half][node], which the node kernel adds up.  Same pipeline as below: the next tile's
        // rows are requested while this tile's layer 2 / epilogue run.
        const int tstride = gridDim.x * NWAVES;
        int t = blockIdx.x * NWAVES + wave;
        if (t >= a.n_tiles) return;
        float bias[4];
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) bias[bo] = c_b2[32 * bo + c];
        int2 tn = a.tile_list[t];
        int4 info = a.node_info[tn.x];
        int n = tn.x, half = tn.y;
        int src = __builtin_amdgcn_readfirstlane(info.x), base = __builtin_amdgcn_readfirstlane(info.y);
        int K = __builtin_amdgcn_readfirstlane(info.z);
        int colc = (32 * half + c < K) ? 32 * half + c : 0;
        int j = a.E_idx[(size_t)src * 64 + colc];
        Pslot[lane] = Prows[(size_t)n * 64 + lane];
        Tile x, acc, t2;
        tile_load_edge(x, block_of(n, src), colc, h);
        tile_load_row(acc, a.Q + (size_t)(base + j) * HD, h);
        for (;;) {
            const int tnext = t + tstride;
            const bool more = tnext < a.n_tiles;
            const int2 tn2 = more ? a.tile_list[tnext] : tn;
            const int4 ninfo = a.node_info[tn2.x];
            tile_add_row(acc, reinterpret_cast<const float *>(Pslot), h);
            if (HOISTED) {
#pragma unroll
                for (int bo = 0; bo < 4; ++bo) acc.b[bo] += x.b[bo];
            } else {
                gemm128_h_lds<TERMS, false, false, true>(acc, x, w1, lane, a.gelu_a);      // layer 1 (pre-split h_E tile)
            }
            const int nsrc = __builtin_amdgcn_readfirstlane(ninfo.x), nbase = __builtin_amdgcn_readfirstlane(ninfo.y);
            const int nK = __builtin_amdgcn_readfirstlane(ninfo.z);
            const int ncolc = (32 * tn2.y + c < nK) ? 32 * tn2.y + c : 0;
            const int nj = a.E_idx[(size_t)nsrc * 64 + ncolc];
            const float2 npv = Prows[(size_t)tn2.x * 64 + lane];
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) {
                float bv = bias[bo];
                asm volatile("" : "+v"(bv));
#pragma unroll
                for (int r = 0; r < 16; ++r) t2.b[bo][r] = bv;
            }
            gemm128_h_lds<TERMS, true, true>(t2, acc, w2, lane, a.gelu_a);    // layer 2 on GELU(layer 1), transposed
            tile_load_edge(x, block_of(tn2.x, nsrc), ncolc, h);
            tile_load_row(acc, a.Q + (size_t)(nbase + nj) * HD, h);
            __builtin_amdgcn_sched_barrier(0);
            tile_gelu(t2, a.gelu_b);
            const int cnt = K - 32 * half;
            // one partial per half AND lane half (plane half + 2 h), not reduced across the lane halves: the node
            // kernel adds them as (a0 + a1) + (b0 + b1), the order of msg_kernel_h, so results do not depend on
            // which of the two kernels ran
            float *Srow = a.S + ((size_t)(half + 2 * h) * a.n_nodes + n) * HD;
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) {
                float s0 = 0.f;
                if (cnt >= 32) {
                    f32x2 s2 = tile_pair(t2.b[bo], 0);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) s2 += tile_pair(t2.b[bo], r);
                    s0 = s2.x + s2.y;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s0 += ((r & 3) + 8 * (r >> 2) + 4 * h < cnt) ? t2.b[bo][r] : 0.f;
                }
                Srow[32 * bo + c] = s0;
            }
            if (!more) break;
            t = tnext; tn = tn2; n = tn2.x; half = tn2.y; src = nsrc; base = nbase; K = nK;
            Pslot[lane] = npv;
        }
        return;
    }

    const NodeSpan span = wave_node_span(a.n_nodes, NWAVES, wave);
    const int stride = span.stride, n_end = span.end;
    int n = span.first;                              // wave-uniform
    if (n >= n_end) return;

    int4 info = a.node_info[n];
    int src = __builtin_amdgcn_readfirstlane(info.x), base = __builtin_amdgcn_readfirstlane(info.y);
    int K = __builtin_amdgcn_readfirstlane(info.z);
    int jA = a.E_idx[(size_t)src * 64 + (c < K ? c : 0)];
    int jB = a.E_idx[(size_t)src * 64 + (32 + c < K ? 32 + c : 0)];
    Pslot[lane] = Prows[(size_t)n * 64 + lane];
    int half = 0;
    Tile x, acc, t2;
    tile_load_edge(x, block_of(n, src), c < K ? c : 0, h);
    tile_load_row(acc, a.Q + (size_t)(base + jA) * HD, h);
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    float bias[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) bias[bo] = c_b2[32 * bo + c];

    // the node after this one (kept equal to the current node when there is none, so that the
    // prefetch below always has a valid address and needs no branch)
    int4 ninfo = info;
    int nsrc = src, nbase = base, nK = K, njA = jA, njB = jB;
    float2 npv = {0.f, 0.f};
    for (;;) {
        const int n2 = n + stride;
        const bool next_node = n2 < n_end;
        const bool first_half = half == 0;
        if (first_half && next_node) ninfo = a.node_info[n2];
        tile_add_row(acc, reinterpret_cast<const float *>(Pslot), h);
        if (HOISTED) {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) acc.b[bo] += x.b[bo];
        } else {
            gemm128_h_lds<TERMS, false, false, true>(acc, x, w1, lane, a.gelu_a);      // layer 1 (pre-split h_E tile)
        }
        if (first_half && next_node) {                   // next node's neighbour list and P row
            nsrc = __builtin_amdgcn_readfirstlane(ninfo.x);
            nbase = __builtin_amdgcn_readfirstlane(ninfo.y);
            nK = __builtin_amdgcn_readfirstlane(ninfo.z);
            njA = a.E_idx[(size_t)nsrc * 64 + (c < nK ? c : 0)];
            njB = a.E_idx[(size_t)nsrc * 64 + (32 + c < nK ? 32 + c : 0)];
            npv = Prows[(size_t)n2 * 64 + lane];
        }
        const bool next_half = first_half && K > 32;
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) {
            float bv = bias[bo];
            asm volatile("" : "+v"(bv));   // or the 64 copies are built once, outside the loop, and spilled
#pragma unroll
            for (int r = 0; r < 16; ++r) t2.b[bo][r] = bv;
        }
        gemm128_h_lds<TERMS, true, true>(t2, acc, w2, lane, a.gelu_a);    // layer 2 on GELU(layer 1), output transposed
        {   // edge rows and Q rows of the next tile, in flight during the epilogue
            const int pn = next_half ? n : (next_node ? n2 : n), ps = next_half ? src : nsrc;
            const int pe = next_half ? (32 + c < K ? 32 + c : 0) : (c < nK ? c : 0);
            const int pq = next_half ? base + jB : nbase + njA;
            tile_load_edge(x, block_of(pn, ps), pe, h);
            tile_load_row(acc, a.Q + (size_t)pq * HD, h);
            __builtin_amdgcn_sched_barrier(0);
        }
        tile_gelu(t2, a.gelu_b);
        const int cnt = K - 32 * half;                   // valid edges of this tile (wave-uniform)
        if (cnt >= 32) {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) {
                f32x2 s2 = tile_pair(t2.b[bo], 0);      // packed adds: two partial sums per block
#pragma unroll
                for (int r = 2; r < 16; r += 2) s2 += tile_pair(t2.b[bo], r);
                sum[bo] += s2.x + s2.y;
            }
        } else {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) {
                float s0 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s0 += ((r & 3) + 8 * (r >> 2) + 4 * h < cnt) ? t2.b[bo][r] : 0.f;
                sum[bo] += s0;
            }
        }
        if (next_half) {
            half = 1;
            continue;
        }
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) {
            const float tot = sum[bo] + __shfl_xor(sum[bo], 32);
            if (h == 0) a.S[(size_t)n * HD + 32 * bo + c] = tot;
            sum[bo] = 0.f;
        }
        if (!next_node) break;
        n = n2; src = nsrc; base = nbase; K = nK; jA = njA; jB = njB;
        Pslot[lane] = npv;
        half = 0;
    }
}


template <int TERMS>
static void launch_tile_h(bool update, const EdgeArgs &ea, const int2 *tile_list, int n_tiles, hipStream_t st) {
    static bool attr_set = false;
    constexpr int MSG_WAVES = 8, UPD_WAVES = 8;
    const size_t lds_msg = 16 * edge_lds_u4<false, MSG_WAVES>(), lds_upd = 16 * edge_lds_u4<true, UPD_WAVES>();
    if (!attr_set) {
        set_max_lds(reinterpret_cast<const void *>(msg_tile_kernel_h<MSG_WAVES, false, TERMS>), lds_msg);
        set_max_lds(reinterpret_cast<const void *>(msg_tile_kernel_h<MSG_WAVES, true, TERMS>), lds_msg);
        set_max_lds(reinterpret_cast<const void *>(upd_tile_kernel_h<UPD_WAVES, false, TERMS>), lds_upd);
        set_max_lds(reinterpret_cast<const void *>(upd_tile_kernel_h<UPD_WAVES, true, TERMS>), lds_upd);
        attr_set = true;
    }
    const int nw = update ? UPD_WAVES : MSG_WAVES;
    const bool hoisted = ea.E1 != nullptr;
    EdgeTileArgs ta;
    static_cast<EdgeArgs &>(ta) = ea;
    ta.tile_list = tile_list; ta.n_tiles = n_tiles;
    const int groups = (n_tiles + nw - 1) / nw;
    dim3 grid(groups < num_cu() ? groups : num_cu()), block(nw * 64);
    if (update && hoisted) hipLaunchKernelGGL((upd_tile_kernel_h<UPD_WAVES, true, TERMS>), grid, block, lds_upd, st, ta);
    else if (update) hipLaunchKernelGGL((upd_tile_kernel_h<UPD_WAVES, false, TERMS>), grid, block, lds_upd, st, ta);
    else if (hoisted) hipLaunchKernelGGL((msg_tile_kernel_h<MSG_WAVES, true, TERMS>), grid, block, lds_msg, st, ta);
    else hipLaunchKernelGGL((msg_tile_kernel_h<MSG_WAVES, false, TERMS>), grid, block, lds_msg, st, ta);
}

void launch_edge_tile(int terms, bool update, const EdgeArgs &ea, const int2 *tile_list, int n_tiles, hipStream_t st) {
    if (terms == 3) launch_tile_h<3>(update, ea, tile_list, n_tiles, st);
    else launch_tile_h<4>(update, ea, tile_list, n_tiles, st);
}
