// Split-fp16 node update for SMALL and medium jobs ("wide" kernel), its own translation unit (see edge_args.h).
//
// node_kernel_h (denoiser_kernels.hip) gives a 32-node tile to ONE wave, which walks the up to 13 weight blocks of
// the update one after the other: right when thousands of tiles share every block through LDS, but a job of a
// few hundred nodes then keeps a handful of waves busy for 13 dependent block fetches (46 us per launch at 87
// nodes: half of a DDPM step).  Here a workgroup of EIGHT waves (g, bo), g in {0, 1}, owns one tile:
//   * every contraction is cut by output block: a wave computes the 32 output features 32 bo .. 32 bo + 31 - 8
//     k-steps x TERMS MFMAs - with its 16 KB of weight fragments loaded straight from L2 into registers, two
//     blocks ahead of use;
//   * independent contractions run side by side: the four W_in chunks of the FFN and the up to four projections
//     two at a time (g and g + 2); the four W_out chunks accumulate onto one accumulator and stay a chain on the
//     waves g = 0 (keeping the summation order of the one-wave kernel), but a short one: their GELU-ed, split
//     inputs are already waiting in LDS;
//   * operands travel between waves as split-fp16 FRAGMENTS (16 KB per tile in LDS): the wave that owns output
//     block b holds exactly the registers that make up the fragments of k-steps 2b and 2b + 1, so every element is
//     activated and split once, by its producer; LayerNorm / modulation need whole columns and run on the waves
//     g = 0, each on the full tile (exchanged as fp32 through LDS), with the code of the one-wave kernel.
// Same arithmetic in the same order per element and per accumulator => results are bit-identical to
// node_kernel_h (tests/test_hip_parity.py).  Critical path: W3 | LN | 2 x W_in | 4 x W_out | LN | 2 projections = 9
// short contractions of 24 MFMAs instead of 13 of 96.
#include "node_args.h"
#include "wide_common.h"

namespace {

constexpr int WIDE_WAVES = 8;
// LDS map in 16-byte words
constexpr int L_FRAG_A = 0;                          // input fragments of the current phase (S, v, projection input)
constexpr int L_FRAG_B = L_FRAG_A + FRAG_U4;         // projection input h_V + h_Venc
constexpr int L_HID = L_FRAG_B + FRAG_U4;            // 4 hidden tiles of the FFN, GELU-ed, as fragments
constexpr int L_XCH = L_HID + 4 * FRAG_U4;           // one fp32 tile [32 chunks][32 columns] float4
constexpr int L_MOD = L_XCH + 1024;                  // folded modulation vectors A1, B1, A2, B2
constexpr int L_VEC = L_MOD + 128;                   // b3, b_in[4], b_out, projection biases [4] (128 floats each)
constexpr int L_END = L_VEC + 10 * 32;
constexpr int WIDE_LDS_BYTES = L_END * 16;

#ifdef NW_STAMP      // diagnostic build (tools/edge_variants.py): phase stamps of workgroup 0's first wave, left in h_V[0][0..]
#define NW_MARK(i) do { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NW_MARK(i) do {} while (0)
#endif

template <bool MODE_UPD, int TERMS>
__global__ __launch_bounds__(WIDE_WAVES * 64, 2) void node_kernel_w(NodeArgs a) {
    extern __shared__ __align__(16) u32x4 wl[];
#ifdef NW_STAMP
    unsigned long long stamp[16] = {};
#ifdef NW_REPEAT     // the whole body twice, the stamps of the second pass: what a warm instruction cache / TLB / L2 would give
    for (int rep = 0; rep < 2; ++rep) {
    __syncthreads();
#endif
#endif
    NW_MARK(0);
    u32x4 *fragA = wl + L_FRAG_A, *fragB = wl + L_FRAG_B;
    float4 *xch = reinterpret_cast<float4 *>(wl + L_XCH);
    const float *modAB = reinterpret_cast<const float *>(wl + L_MOD);
    const float *lv = reinterpret_cast<const float *>(wl + L_VEC);
    // Every kernel argument the prologue uses, fetched in ONE batch and pinned in scalar registers: left to itself the
    // compiler loads each where it is first used, waits for it there, and re-reads some later rather than keep them -
    // seven dependent round trips to the (cold) argument segment, ~1 us each, before the tile's rows were even requested
    // (tools/node_wide_stamps.py).
    const int4 *p_node_info = a.node_info;
    const float *p_S = a.S;
    float *p_hV = a.hV;
    const void *p_w3 = a.blk_h[0], *p_win0 = a.blk_h[1];
    int n_nodes = a.n_nodes, s_partials = a.s_partials;
    PIN_PTR(p_node_info); PIN_PTR(p_S); PIN_PTR(p_hV); PIN_PTR(p_w3); PIN_PTR(p_win0);
    asm volatile("" : "+s"(n_nodes), "+s"(s_partials)
                 : "s"(a.n_proj), "s"(a.proj_flags[0]), "s"(a.proj_flags[1]), "s"(a.proj_flags[2]), "s"(a.proj_flags[3]),
                   "s"(a.b3), "s"(a.b_in), "s"(a.b_out), "s"(a.proj_b[0]), "s"(a.proj_b[1]), "s"(a.proj_b[2]), "s"(a.proj_b[3]),
                   "s"(a.mods), "s"(a.s_scale), "s"(a.blk_h[9]), "s"(a.blk_h[10]));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, bo = wave & 3;                  // g in {0, 1}: which contraction of a round; output block
    const int h = lane >> 5, c = lane & 31;
    const int node = blockIdx.x * 32 + c;
    const bool valid = node < n_nodes;
    const int nc = valid ? node : n_nodes - 1;
    const bool lead = g == 0;                                // the four waves that own the tile's columns
    const GeluK plain = gelu_consts(0);
    const bool any_sum = ((a.n_proj > 0 && (a.proj_flags[0] & 1)) | (a.n_proj > 1 && (a.proj_flags[1] & 1)) |
                          (a.n_proj > 2 && (a.proj_flags[2] & 1)) | (a.n_proj > 3 && (a.proj_flags[3] & 1))) != 0;

    // Every wave works through a fixed list of blocks with two fragment buffers: the block after next is requested as
    // soon as a buffer has been consumed (a third buffer costs more in spilled registers than its earlier requests return).
    //   lead, update : W3, W_in 0, W_in 2, W_out 0, W_out 1, W_out 2, W_out 3, projection 0, projection 2
    //   other, update: W_in 1, W_in 3, projection 1, projection 3  (requested after the first barrier: until then the
    //                  lead waves' rows and W3 have the CU's path to L2 to themselves)
    //   input kernel : projection g, projection g + 2
    BlockQuarter w0, w1;
    auto proj_blk = [&](int p) { return a.blk_h[(MODE_UPD ? 9 : 0) + p]; };
    // the tile's own rows (lead waves: this wave's block of them), requested before anything else
    f32x16 vq, sq, evq;
    int4 info;                                               // requested after the rows: nothing in front of them waits for it
    if (MODE_UPD) {
        if (lead) {
            quarter_load(sq, p_S + (size_t)nc * HD, bo, h);
            if (s_partials) {       // tile kernels: one partial per half and lane half, planes half + 2 h; same order as msg_kernel_h
                // all four planes are read whatever K (no look-up to wait for); planes 1 and 3 of a node with K <= 32 hold
                // stale bits, which are selected away, never added
                const size_t plane = (size_t)n_nodes * HD;
                f32x16 s1, s2, s3;
                quarter_load(s1, p_S + plane + (size_t)nc * HD, bo, h);
                quarter_load(s2, p_S + 2 * plane + (size_t)nc * HD, bo, h);
                quarter_load(s3, p_S + 3 * plane + (size_t)nc * HD, bo, h);
                quarter_load(vq, p_hV + (size_t)nc * HD, bo, h);
                w0.start(p_w3, bo, lane);              // W3
                info = p_node_info[nc];
                __builtin_amdgcn_sched_barrier(0);           // everything requested before the first wait
#ifdef NW_STAMP
                NW_MARK(11);
                { int z = info.z; asm volatile("" : "+v"(z)); }
                NW_MARK(12);
                asm volatile("" : "+v"(s3));
                NW_MARK(13);
                asm volatile("" : "+v"(vq));
                NW_MARK(14);
                NW_MARK(15);
#endif
                const f32x16 two = (sq + s1) + (s2 + s3), one = sq + s2;
                const bool both = info.z > 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) sq[r] = both ? two[r] : one[r];
                // W_in 0 only now: with the four planes of S, h_V and two weight quarters in flight the compiler parks
                // arriving fragments in scratch, and every such store waits for ALL outstanding loads (5 round trips)
                __builtin_amdgcn_sched_barrier(0);
                w1.start_head(p_win0, bo, lane);         // W_in 0
            } else {
                quarter_load(vq, p_hV + (size_t)nc * HD, bo, h);
                w0.start(p_w3, bo, lane);              // W3
                w1.start_head(p_win0, bo, lane);         // W_in 0
                info = p_node_info[nc];
            }
        } else {
            info = p_node_info[nc];
        }
    } else {
        if (g < a.n_proj) w0.start(proj_blk(g), bo, lane);
        if (g + 2 < a.n_proj) w1.start(proj_blk(g + 2), bo, lane);
        info = p_node_info[nc];
    }
    {   // small vectors -> LDS: slot 0 b3, 1-4 b_in, 5 b_out, 6-9 projection biases (zeros where absent)
        const int i = tid & 31;
        for (int sl = tid >> 5; sl < 10; sl += WIDE_WAVES * 2) {
            const float *src = nullptr;
            if (MODE_UPD && sl == 0) src = a.b3;
            else if (MODE_UPD && sl >= 1 && sl <= 4) src = a.b_in + (sl - 1) * HD;
            else if (MODE_UPD && sl == 5) src = a.b_out;
            else if (sl >= 6 && sl - 6 < a.n_proj) src = a.proj_b[sl - 6];
            wl[L_VEC + sl * 32 + i] = src ? reinterpret_cast<const u32x4 *>(src)[i] : u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (MODE_UPD && tid >= 256 && tid < 320) {               // a wave that is not waiting for the tile's rows
        const int k = (tid >> 5) & 1;
        const float4 *m = reinterpret_cast<const float4 *>(a.mods) + 96 * k;
        const int i = tid & 31;
        const float4 sv = m[i], cc = m[32 + i], gg = m[64 + i];
        float4 *cf = reinterpret_cast<float4 *>(wl + L_MOD) + 64 * k;
        cf[i] = make_float4(gg.x * (1.0f + cc.x), gg.y * (1.0f + cc.y), gg.z * (1.0f + cc.z), gg.w * (1.0f + cc.w));
        cf[32 + i] = make_float4(gg.x * sv.x, gg.y * sv.y, gg.z * sv.z, gg.w * sv.w);
    }
    NW_MARK(10);

    if (MODE_UPD) {
        // ---- phase A: t = W3 @ (S / 64) + K b3 / 64, v = LN1(h_V + 64 t / 30) ------------------------------
        if (lead) {
            sq *= a.s_scale;
            publish_quarter<false>(fragA, sq, bo, lane, plain);
        }
        __syncthreads();                                     // S fragments, staged vectors, modAB
        NW_MARK(1);
        if (lead) {
            f32x16 q;
            quarter_load(q, lv, bo, h);
            q *= (float)info.z * 0.015625f;
            w0.run<TERMS>(q, fragA, lane);                   // W3
            w1.start_tail(a.blk_h[1], bo, lane);             // W_in 0
            w0.start_head(a.blk_h[5], bo, lane);             // W_in 2
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                vq[r] += (q[r] * a.t_scale) / 30.0f;
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // four divisions side by side, not sixteen (registers)
            }
            xch_write(xch, vq, bo, h, c);
        } else {
            w0.start(a.blk_h[3], bo, lane);                  // W_in 1
            w1.start_head(a.blk_h[7], bo, lane);             // W_in 3
        }
        __syncthreads();
        NW_MARK(2);
        if (lead) {
            // the LayerNorm needs whole columns: every lead wave streams the column's values through the moment sums
            // (the one-wave kernel's order) and normalises its own block
            xch_layernorm_affine(vq, xch, 1e-6f, modAB, modAB + HD, bo, h, c);
            publish_quarter<false>(fragA, vq, bo, lane, plain);
        }
        __syncthreads();
        NW_MARK(3);
        // ---- phase B1: hidden chunk = GELU(W_in[chunk] @ v + b_in[chunk]); chunks g and g + 2 ----------------
        {
            f32x16 q;
            quarter_load(q, lv + (1 + g) * HD, bo, h);
            if (lead) {
                w1.run<TERMS>(q, fragA, lane);               // W_in 0
                w0.start_tail(a.blk_h[5], bo, lane);         // W_in 2
                w1.start_head(a.blk_h[2], bo, lane);         // W_out 0
            } else {
                w0.run<TERMS>(q, fragA, lane);               // W_in 1
                w1.start_tail(a.blk_h[7], bo, lane);         // W_in 3
                if (1 < a.n_proj) w0.start_head(proj_blk(1), bo, lane);
            }
            publish_quarter<true>(wl + L_HID + g * FRAG_U4, q, bo, lane, a.gelu_ffn);
            quarter_load(q, lv + (3 + g) * HD, bo, h);
            if (lead) {
                w0.run<TERMS>(q, fragA, lane);               // W_in 2
                w1.start_tail(a.blk_h[2], bo, lane);         // W_out 0
                w0.start_head(a.blk_h[4], bo, lane);         // W_out 1
            } else {
                w1.run<TERMS>(q, fragA, lane);               // W_in 3
                if (1 < a.n_proj) w0.start_tail(proj_blk(1), bo, lane);
                if (3 < a.n_proj) w1.start_head(proj_blk(3), bo, lane);
            }
            publish_quarter<true>(wl + L_HID + (2 + g) * FRAG_U4, q, bo, lane, a.gelu_ffn);
        }
        __syncthreads();
        NW_MARK(4);
        // ---- phase B2: t = b_out + sum_ch W_out[ch] @ hidden[ch] (one accumulator, chunk after chunk) -----
        if (lead) {
            f32x16 q;
            quarter_load(q, lv + 5 * HD, bo, h);
            w1.run<TERMS>(q, wl + L_HID + 0 * FRAG_U4, lane);
            w0.start_tail(a.blk_h[4], bo, lane);             // W_out 1
            w1.start_head(a.blk_h[6], bo, lane);             // W_out 2
            w0.run<TERMS>(q, wl + L_HID + 1 * FRAG_U4, lane);
            w1.start_tail(a.blk_h[6], bo, lane);
            w0.start_head(a.blk_h[8], bo, lane);             // W_out 3
            w1.run<TERMS>(q, wl + L_HID + 2 * FRAG_U4, lane);
            w0.start_tail(a.blk_h[8], bo, lane);
            if (0 < a.n_proj) w1.start_head(proj_blk(0), bo, lane);
            w0.run<TERMS>(q, wl + L_HID + 3 * FRAG_U4, lane);
            if (0 < a.n_proj) w1.start_tail(proj_blk(0), bo, lane);
            if (2 < a.n_proj) w0.start_head(proj_blk(2), bo, lane);
            if (any_sum && !a.venc_is_self) quarter_load(evq, a.hVenc_in + (size_t)nc * HD, bo, h);   // travels under the LayerNorm
            vq += q * a.ffn_scale;
            xch_write(xch, vq, bo, h, c);
        }
        __syncthreads();
        NW_MARK(5);
        if (lead) xch_layernorm_affine(vq, xch, 1e-6f, modAB + 2 * HD, modAB + 3 * HD, bo, h, c);
    } else {
        __syncthreads();                                     // staged vectors
    }
    NW_MARK(6);
    // ---- new h_V: store, publish the projection inputs ----------------------------------------------------
    if (lead) {
        f32x16 mine;
        if (MODE_UPD) {
            mine = vq;
        } else {
            // h_V = x_in(x) is element-wise per feature: every lead wave computes its own block
            const float x0 = a.x[nc * 3 + 0], x1 = a.x[nc * 3 + 1], x2 = a.x[nc * 3 + 2];
            const bool sc = a.in_dim == 6;
            const bool have_sc = sc && a.x_sc != nullptr;
            const float s0 = have_sc ? a.x_sc[nc * 3 + 0] : 0.f, s1 = have_sc ? a.x_sc[nc * 3 + 1] : 0.f,
                        s2 = have_sc ? a.x_sc[nc * 3 + 2] : 0.f;
            quarter_load(mine, a.x_in_b, bo, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * bo + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float *wr = a.x_in_w + f * a.in_dim;
                float acc = 0.f;
                if (sc) {
                    acc = fmaf(s2, wr[2], fmaf(s1, wr[1], s0 * wr[0]));
                    wr += 3;
                }
                mine[r] += fmaf(x2, wr[2], fmaf(x1, wr[1], fmaf(x0, wr[0], acc)));
            }
        }
        if (valid) {
            quarter_store(mine, a.hV + (size_t)node * HD, bo, h);
            if (a.hVenc_out) quarter_store(mine, a.hVenc_out + (size_t)node * HD, bo, h);
        }
        publish_quarter<false>(fragA, mine, bo, lane, plain);
        if (any_sum) {
            f32x16 sum = mine;
            if (a.venc_is_self) {
                sum += mine;
            } else if (MODE_UPD) {
                sum += evq;
            } else {
                f32x16 e;
                quarter_load(e, a.hVenc_in + (size_t)nc * HD, bo, h);
                sum += e;
            }
            publish_quarter<false>(fragB, sum, bo, lane, plain);
        }
    }
    __syncthreads();
    NW_MARK(7);
    // ---- phase C: projections g and g + 2 on the waves (g, bo) ----------------------------------------------
    auto project = [&](int p, const BlockQuarter &wp) {
        const int fl = a.proj_flags[p];
        f32x16 out;
        quarter_load(out, lv + (6 + p) * HD, bo, h);
        if (fl & 2) {
            f32x16 ts;
            quarter_load(ts, a.TS + (size_t)info.w * HD, bo, h);
            out += ts;
        }
        wp.run<TERMS>(out, (fl & 1) ? fragB : fragA, lane);
        if (valid) quarter_store(out, a.proj_out[p] + (size_t)node * HD, bo, h);
    };
    // buffers: input kernel w0 / w1; update kernel: lead w1 (projection 0) / w0 (projection 2), others w0 / w1
    // (branches, not `lead ? w1 : w0`: a select between two buffers is 64 v_cndmask and a third buffer's registers)
    if (MODE_UPD && lead) {
        if (0 < a.n_proj) project(0, w1);
        if (2 < a.n_proj) {
            w0.start_tail(proj_blk(2), bo, lane);
            project(2, w0);
        }
    } else if (MODE_UPD) {
        if (1 < a.n_proj) project(1, w0);
        if (3 < a.n_proj) {
            w1.start_tail(proj_blk(3), bo, lane);
            project(3, w1);
        }
    } else {
        if (g < a.n_proj) project(g, w0);
        if (g + 2 < a.n_proj) project(g + 2, w1);
    }
#ifdef NW_STAMP
    NW_MARK(8);
    __builtin_amdgcn_s_waitcnt(0);
    NW_MARK(9);
#ifdef NW_REPEAT
    }
#endif
    if (MODE_UPD && blockIdx.x == 0 && threadIdx.x == 0) {
        for (int i = 0; i < 9; ++i) a.hV[i] = (float)(stamp[i + 1] - stamp[i]);
        a.hV[0] = (float)(stamp[1] - stamp[0]);
        a.hV[9] = (float)(stamp[10] - stamp[0]); a.hV[10] = (float)(stamp[1] - stamp[10]);
        a.hV[11] = (float)(stamp[11] - stamp[0]); a.hV[12] = (float)(stamp[12] - stamp[11]); a.hV[13] = (float)(stamp[13] - stamp[12]);
        a.hV[14] = (float)(stamp[14] - stamp[13]); a.hV[15] = (float)(stamp[15] - stamp[14]);
    }
#endif
}

template <int TERMS>
void launch_w(bool upd, const NodeArgs &na, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        set_max_lds(reinterpret_cast<const void *>(node_kernel_w<true, TERMS>), WIDE_LDS_BYTES);
        set_max_lds(reinterpret_cast<const void *>(node_kernel_w<false, TERMS>), WIDE_LDS_BYTES);
        attr_set = true;
    }
    dim3 grid((na.n_nodes + 31) / 32), block(WIDE_WAVES * 64);
    if (upd) hipLaunchKernelGGL((node_kernel_w<true, TERMS>), grid, block, WIDE_LDS_BYTES, st, na);
    else hipLaunchKernelGGL((node_kernel_w<false, TERMS>), grid, block, WIDE_LDS_BYTES, st, na);
}

}  // namespace

void launch_node_wide(int terms, bool upd, const NodeArgs &na, hipStream_t st) {
    if (terms == 3) launch_w<3>(upd, na, st);
    else launch_w<4>(upd, na, st);
}
