// Split-fp16 node update for SMALL jobs, four waves per 32-node tile with the whole register file ("quad" kernel), its own
// translation unit (see edge_args.h).
//
// node_kernel_w (node_wide_kernels.hip) cuts every contraction of the update by output block over EIGHT waves: two per
// SIMD, 256 registers each, of which two 64-register buffers hold the weight quarters in flight.  On an idle chip a
// request to the weights takes ~3 us and a contraction of 24 MFMAs 0.4 us: with two buffers a wave cannot ask early
// enough, and the update of a small job was four dependent round trips to memory around ~3.5 us of arithmetic (21.6 us
// per launch, two thirds of a DDPM step at 87 residues; tools/node_wide_stamps.py).  Here FOUR waves own the tile - wave
// bo computes output block bo of all 13 contractions, one after the other - at one wave per SIMD, i.e. with 512 registers
// per lane: a ring of QUAD_RING (4) weight quarters, every one refilled with the block QUAD_RING places ahead as soon as its
// own has been consumed (what does not fit the 256 architectural registers lives in accumulator registers: copies, not
// scratch - a scratch access waits for every load in flight; rings of 4 / 5 / 6: 182 / 185 / 188 us per step at 87 residues).  Same phases, same LDS exchange (fragments; fp32 tile for
// the LayerNorms, moments streamed in the one-wave order), same arithmetic per element and accumulator as node_kernel_w
// and node_kernel_h => bit-identical (tests/test_hip_parity.py).
#include "node_args.h"
#include "wide_common.h"

namespace {

constexpr int QUAD_WAVES = 4;
#ifndef QUAD_RING
#define QUAD_RING 4
#endif
// LDS map in 16-byte words (as node_kernel_w's)
constexpr int Q_FRAG_A = 0;                          // input fragments of the current phase (S, v, projection input)
constexpr int Q_FRAG_B = Q_FRAG_A + FRAG_U4;         // projection input h_V + h_Venc
constexpr int Q_HID = Q_FRAG_B + FRAG_U4;            // 4 hidden tiles of the FFN, GELU-ed, as fragments
constexpr int Q_XCH = Q_HID + 4 * FRAG_U4;           // one fp32 tile [32 chunks][32 columns] float4
constexpr int Q_MOD = Q_XCH + 1024;                  // folded modulation vectors A1, B1, A2, B2
constexpr int Q_VEC = Q_MOD + 128;                   // b3, b_in[4], b_out, projection biases [4] (128 floats each)
constexpr int Q_END = Q_VEC + 10 * 32;
constexpr int QUAD_LDS_BYTES = Q_END * 16;

// the 13 blocks in execution order as indices into NodeArgs::blk_h ([W3, Win0, Wout0, Win1, Wout1, .., proj0..3])
__device__ constexpr int quad_block(int k) {
    constexpr int order[13] = {0, 1, 3, 5, 7, 2, 4, 6, 8, 9, 10, 11, 12};
    return order[k];
}

#ifdef NQ_STAMP      // diagnostic build (tools/node_quad_stamps.py): phase stamps of workgroup 0's first wave, left in PQ[0][0][0..]
#define NQ_MARK(i) do { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NQ_MARK(i) do {} while (0)
#endif

template <int TERMS>
__global__ __launch_bounds__(QUAD_WAVES * 64, 1) void node_kernel_q(NodeArgs a) {
    extern __shared__ __align__(16) u32x4 wl[];
#ifdef NQ_STAMP
    unsigned long long stamp[12] = {};
#endif
    NQ_MARK(0);
    u32x4 *fragA = wl + Q_FRAG_A, *fragB = wl + Q_FRAG_B;
    float4 *xch = reinterpret_cast<float4 *>(wl + Q_XCH);
    const float *modAB = reinterpret_cast<const float *>(wl + Q_MOD);
    const float *lv = reinterpret_cast<const float *>(wl + Q_VEC);
    // every argument the prologue uses in one batch, pinned (wide_common.h)
    const int4 *p_node_info = a.node_info;
    const float *p_S = a.S;
    float *p_hV = a.hV;
    int n_nodes = a.n_nodes, s_partials = a.s_partials, n_proj = a.n_proj;
    PIN_PTR(p_node_info); PIN_PTR(p_S); PIN_PTR(p_hV);
    asm volatile("" : "+s"(n_nodes), "+s"(s_partials), "+s"(n_proj)
                 : "s"(a.proj_flags[0]), "s"(a.proj_flags[1]), "s"(a.proj_flags[2]), "s"(a.proj_flags[3]), "s"(a.b3), "s"(a.b_in),
                   "s"(a.b_out), "s"(a.proj_b[0]), "s"(a.proj_b[1]), "s"(a.proj_b[2]), "s"(a.proj_b[3]), "s"(a.mods), "s"(a.s_scale),
                   "s"(a.blk_h[0]), "s"(a.blk_h[1]), "s"(a.blk_h[3]), "s"(a.blk_h[5]), "s"(a.blk_h[7]), "s"(a.blk_h[2]));
    NQ_MARK(10);
    const int tid = threadIdx.x, lane = tid & 63, bo = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int node = blockIdx.x * 32 + c;
    const bool valid = node < n_nodes;
    const int nc = valid ? node : n_nodes - 1;
    const GeluK plain = gelu_consts(0);
    const bool any_sum = ((n_proj > 0 && (a.proj_flags[0] & 1)) | (n_proj > 1 && (a.proj_flags[1] & 1)) |
                          (n_proj > 2 && (a.proj_flags[2] & 1)) | (n_proj > 3 && (a.proj_flags[3] & 1))) != 0;
    const int n_blocks = 9 + n_proj;

    BlockQuarter ring[QUAD_RING];
    // block k lives in ring[k % QUAD_RING]; request(k) is called when that buffer's previous block has been consumed
    auto request = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (k < 9 || k < n_blocks) ring[k % QUAD_RING].start(a.blk_h[quad_block(k)], bo, lane);
    };
    auto contract = [&](auto kc, f32x16 &acc, const u32x4 *frag) {
        constexpr int k = decltype(kc)::value;
        ring[k % QUAD_RING].template run<TERMS>(acc, frag, lane);
        if (k + QUAD_RING < 13) request(std::integral_constant<int, (k + QUAD_RING < 13 ? k + QUAD_RING : 0)>{});
    };
#define QK(k) std::integral_constant<int, k>{}

    // the tile's own rows (this wave's block of them) first, then the first blocks
    f32x16 vq, sq, evq;
    quarter_load(sq, p_S + (size_t)nc * HD, bo, h);
    f32x16 s1, s2, s3;
    if (s_partials) {       // tile-wise message kernels: one partial per half and lane half, planes half + 2 h
        // all four planes are read whatever K (no look-up to wait for); planes 1 and 3 of a node with K <= 32 hold stale
        // bits, which are selected away, never added
        const size_t plane = (size_t)n_nodes * HD;
        quarter_load(s1, p_S + plane + (size_t)nc * HD, bo, h);
        quarter_load(s2, p_S + 2 * plane + (size_t)nc * HD, bo, h);
        quarter_load(s3, p_S + 3 * plane + (size_t)nc * HD, bo, h);
    }
    quarter_load(vq, p_hV + (size_t)nc * HD, bo, h);
    const int4 info = p_node_info[nc];
    request(QK(0));
    request(QK(1));
    request(QK(2));
    NQ_MARK(11);
    {   // small vectors -> LDS: slot 0 b3, 1-4 b_in, 5 b_out, 6-9 projection biases (zeros where absent)
        const int i = tid & 31;
        for (int sl = tid >> 5; sl < 10; sl += QUAD_WAVES * 2) {
            const float *src = nullptr;
            if (sl == 0) src = a.b3;
            else if (sl >= 1 && sl <= 4) src = a.b_in + (sl - 1) * HD;
            else if (sl == 5) src = a.b_out;
            else if (sl - 6 < n_proj) src = a.proj_b[sl - 6];
            wl[Q_VEC + sl * 32 + i] = src ? reinterpret_cast<const u32x4 *>(src)[i] : u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (tid >= 192) {                                        // folded modulation: A = gate (1 + scale), B = gate shift
        const int k = (tid >> 5) & 1;
        const float4 *m = reinterpret_cast<const float4 *>(a.mods) + 96 * k;
        const int i = tid & 31;
        const float4 sv = m[i], cc = m[32 + i], gg = m[64 + i];
        float4 *cf = reinterpret_cast<float4 *>(wl + Q_MOD) + 64 * k;
        cf[i] = make_float4(gg.x * (1.0f + cc.x), gg.y * (1.0f + cc.y), gg.z * (1.0f + cc.z), gg.w * (1.0f + cc.w));
        cf[32 + i] = make_float4(gg.x * sv.x, gg.y * sv.y, gg.z * sv.z, gg.w * sv.w);
    }
    if (s_partials) {
        const f32x16 two = (sq + s1) + (s2 + s3), one = sq + s2;
        const bool both = info.z > 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) sq[r] = both ? two[r] : one[r];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (QUAD_RING > 3) request(QK(3));
    if (QUAD_RING > 4) request(QK(4));
    if (QUAD_RING > 5) request(QK(5));

    // ---- phase A: t = W3 @ (S / 64) + K b3 / 64, v = LN1(h_V + 64 t / 30) ----------------------------------
    sq *= a.s_scale;
    publish_quarter<false>(fragA, sq, bo, lane, plain);
    __syncthreads();                                         // S fragments, staged vectors, modAB
    NQ_MARK(1);
    {
        f32x16 q;
        quarter_load(q, lv, bo, h);
        q *= (float)info.z * 0.015625f;
        contract(QK(0), q, fragA);                           // W3
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            vq[r] += (q[r] * a.t_scale) / 30.0f;
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        xch_write(xch, vq, bo, h, c);
    }
    __syncthreads();
    NQ_MARK(2);
    xch_layernorm_affine(vq, xch, 1e-6f, modAB, modAB + HD, bo, h, c);
    publish_quarter<false>(fragA, vq, bo, lane, plain);
    __syncthreads();
    NQ_MARK(3);
    // ---- phase B1: hidden chunk = GELU(W_in[chunk] @ v + b_in[chunk]) ---------------------------------------
    {
        f32x16 q;
        quarter_load(q, lv + 1 * HD, bo, h);
        contract(QK(1), q, fragA);
        publish_quarter<true>(wl + Q_HID + 0 * FRAG_U4, q, bo, lane, a.gelu_ffn);
        quarter_load(q, lv + 2 * HD, bo, h);
        contract(QK(2), q, fragA);
        publish_quarter<true>(wl + Q_HID + 1 * FRAG_U4, q, bo, lane, a.gelu_ffn);
        NQ_MARK(4);
        quarter_load(q, lv + 3 * HD, bo, h);
        contract(QK(3), q, fragA);
        publish_quarter<true>(wl + Q_HID + 2 * FRAG_U4, q, bo, lane, a.gelu_ffn);
        quarter_load(q, lv + 4 * HD, bo, h);
        contract(QK(4), q, fragA);
        publish_quarter<true>(wl + Q_HID + 3 * FRAG_U4, q, bo, lane, a.gelu_ffn);
    }
    __syncthreads();
    NQ_MARK(5);
    // ---- phase B2: t = b_out + sum_ch W_out[ch] @ hidden[ch] (one accumulator, chunk after chunk) -----------
    {
        f32x16 q;
        quarter_load(q, lv + 5 * HD, bo, h);
        contract(QK(5), q, wl + Q_HID + 0 * FRAG_U4);
        contract(QK(6), q, wl + Q_HID + 1 * FRAG_U4);
        contract(QK(7), q, wl + Q_HID + 2 * FRAG_U4);
        contract(QK(8), q, wl + Q_HID + 3 * FRAG_U4);
        if (any_sum && !a.venc_is_self) quarter_load(evq, a.hVenc_in + (size_t)nc * HD, bo, h);   // travels under the LayerNorm
        vq += q * a.ffn_scale;
        xch_write(xch, vq, bo, h, c);
    }
    __syncthreads();
    NQ_MARK(6);
    xch_layernorm_affine(vq, xch, 1e-6f, modAB + 2 * HD, modAB + 3 * HD, bo, h, c);
    // ---- new h_V: store, publish the projection inputs -------------------------------------------------------
    if (valid) {
        quarter_store(vq, a.hV + (size_t)node * HD, bo, h);
        if (a.hVenc_out) quarter_store(vq, a.hVenc_out + (size_t)node * HD, bo, h);
    }
    publish_quarter<false>(fragA, vq, bo, lane, plain);
    if (any_sum) {
        f32x16 sum = vq;
        if (a.venc_is_self) sum += vq;
        else sum += evq;
        publish_quarter<false>(fragB, sum, bo, lane, plain);
    }
    __syncthreads();
    NQ_MARK(7);
    // ---- phase C: the projections, one after the other ------------------------------------------------------
    auto project = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        const int fl = a.proj_flags[p];
        f32x16 out;
        quarter_load(out, lv + (6 + p) * HD, bo, h);
        if (fl & 2) {
            f32x16 ts;
            quarter_load(ts, a.TS + (size_t)info.w * HD, bo, h);
            out += ts;
        }
        ring[(9 + p) % QUAD_RING].template run<TERMS>(out, (fl & 1) ? fragB : fragA, lane);
        if (valid) quarter_store(out, a.proj_out[p] + (size_t)node * HD, bo, h);
    };
    if (0 < n_proj) project(QK(0));
    if (1 < n_proj) project(QK(1));
    if (2 < n_proj) project(QK(2));
    if (3 < n_proj) project(QK(3));
#ifdef NQ_STAMP
    NQ_MARK(8);
    __builtin_amdgcn_s_waitcnt(0);
    NQ_MARK(9);
    // (of the launches WITH projections: the decoder's last update has none and would hide that phase)
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_proj > 0) {
        for (int i = 0; i < 9; ++i) a.proj_out[0][i] = (float)(stamp[i + 1] - stamp[i]);
        a.proj_out[0][9] = (float)(stamp[10] - stamp[0]); a.proj_out[0][10] = (float)(stamp[11] - stamp[10]);
        a.proj_out[0][11] = (float)(stamp[1] - stamp[11]);
    }
#endif
#undef QK
}

template <int TERMS>
void launch_q(const NodeArgs &na, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        set_max_lds(reinterpret_cast<const void *>(node_kernel_q<TERMS>), QUAD_LDS_BYTES);
        attr_set = true;
    }
    dim3 grid((na.n_nodes + 31) / 32), block(QUAD_WAVES * 64);
    hipLaunchKernelGGL((node_kernel_q<TERMS>), grid, block, QUAD_LDS_BYTES, st, na);
}

}  // namespace

// the update kernel (S -> new h_V -> projections) of small jobs; the input kernel stays node_kernel_w<false>
void launch_node_quad(int terms, const NodeArgs &na, hipStream_t st) {
    if (terms == 3) launch_q<3>(na, st);
    else launch_q<4>(na, st);
}
