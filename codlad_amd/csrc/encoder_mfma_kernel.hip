// the TensorProductConvLayer (encoder_kernels.hip says what it computes) with its two dense layers on the
// matrix pipe - the variant `codlad_tp_conv` runs for group = 64 (the intra-level graphs: ~90 neighbours per atom, the bulk
// of the encoder's time).
//
// Per edge the layer evaluates fc = Linear(36, 36) -> ReLU -> Linear(36, weight_numel) and contracts the weight_numel
// (192 / 288 / 384) per-edge weights with the sender's features and the spherical harmonics.  The second Linear is 90 % of
// the arithmetic (384 x 36 multiply-adds per edge) and a plain matrix product over the edges: weights^T [idx][edge] =
// fc3_w [idx][m] hid^T [m][edge] + b.  Here:
//   * a wave takes one receiving node at a time and 32 of its edges per step; lanes l and l + 32 both stand for edge
//     l & 31 (the geometry, the 12-wide edge embedding and the sender's features are evaluated in both halves - 300 of
//     the 15 000 multiply-adds of an edge - so that no operand ever has to cross lanes);
//   * fc.0 and fc.3 run on v_mfma_f32_32x32x16_f16 with every fp32 operand split into fp16 hi + lo halves (22 bits, three
//     matrix instructions per product, fp32 accumulation - the denoiser's f16x3 contraction), biases riding as one more
//     k-slot.  The edges are the B operand's columns, so a result block arrives with lane = edge and 16 weight rows in
//     registers - and the rows of fc.3 are ORDERED (when the workgroup packs them into LDS, once) so that the 16 rows a
//     lane half receives are exactly the weights of one output channel's paths, next to the lane's own inputs: the
//     tensor-product contraction is 16 in-lane multiply-adds per block, nothing is shuffled;
//   * fc.0's result block is already distributed as the next product's B operand wants it (a lane half owns 20 / 16 of the 36
//     hidden units; fc.3's k-slots are assigned accordingly), ReLU and the fp16 split happen in registers;
//   * per node the 48 output channels are accumulated per lane over the steps and summed over the 32 lanes of a half
//     by shuffles at the end: a fixed order per node, whatever shares the launch.
// The packed weights (13 blocks x 3 k-steps x (hi, lo) x 1 KB + fc.0's 12 KB = 90 KB for depth 2) live in LDS for the
// workgroup's lifetime; workgroups are persistent (one per CU, 12 waves).
//
// Powers of two keep the fp16 halves in range whatever the magnitudes are: every operand is scaled before the split so
// that its largest element lands in a fixed binade - the two weight matrices once per workgroup (largest |w| -> [256, 512)),
// an edge's 37 fc inputs and its 36 hidden activations per edge (largest -> [8, 16); a column of the B operand is one
// edge, so the scale is a per-lane number) - and the products are descaled exactly; the descale of fc.3 rides on the
// factor that zeroes the inputs of a lane without an edge, so it costs nothing.
#include "encoder_common.h"

// Measurement-only build switch (tools/ablate_encoder.sh; results are wrong with it): 1 = no fc.3 blocks, 2 = every lane
// reads the receiving node's own row instead of its sender's (no gather), 3 = both.
#ifndef CODLAD_TP_ABLATE
#define CODLAD_TP_ABLATE 0
#endif

int num_cu();                                       // denoiser_kernels.hip
void set_max_lds(const void *fn, size_t bytes);     // denoiser_kernels.hip

namespace {

constexpr int MF_WAVES = 8;                  // waves of the packing kernel and of the conv kernel at depth 1, 2
// 12 waves per workgroup = three per SIMD: without the SLP vectoriser the kernel needs 137 / 152 / 191 registers per lane at
// depth 0 / 1 / 2, and held to the 168 of three waves depth 2 spills three of them (303 -> 272 us on a 1.1 M-edge graph)
__host__ __device__ constexpr int conv_waves(int) { return 12; }

// 2^(target - floor(log2 m)): the power of two that moves m (> 0) into [2^target, 2^(target + 1))
DEV float pow2_scale(float m, int target) {
    const int e = (int)((__float_as_uint(fmaxf(m, 1e-30f)) >> 23) & 0xff) - 127;
    return __uint_as_float((unsigned)(127 + target - e) << 23);
}

struct RowSpec {
    int idx;          // row of fc.3 (weight index of the tensor product), -1: a zero row
    float factor;     // the path's Wigner normalisation, folded into the row
};

// Which fc.3 row sits in slot i (0..15) of lane half `half` of block `tile`: the instruction order of
// FullyConnectedTensorProduct (i_in1, i_in2, i_out) gives the offsets O1..O10 of the weight blocks (encoder_kernels.hip).
//   blocks 0-5   12x0e, channel w = 2 tile + half:        slots 0-11 O1[u][w] . x0[u],  12-15 O4[u][w] . (v1[u] . Y1) / sqrt 3
//   blocks 6-7   4x1o,  w = 2 (tile - 6) + half:          slots 0-11 O2[u][w] . x0[u] Y1 / sqrt 3,  12-15 O3[u][w] . v1[u] / sqrt 3
//   block  8     4x1o,  w = half + 2 (slot / 8):          slots 0-3 O6[u][w] . (v1[u] x Y2),  4-7 O8[u][w] . (v2[u] x Y1) / sqrt 6
//   blocks 9-10  4x1e,  w = 2 (tile - 9) + half:          slots 0-3 O5 . (v1 x Y1) / sqrt 6, 4-7 O7 . v2 / sqrt 3, 8-11 O10 . (v2 x Y2)
//   blocks 11-12 12x0o, w = 8 (tile - 11) + 4 half + slot / 4:   O9[u][w] . (v2[u] . Y1) / sqrt 3   (block 12: half 0 only)
DEV RowSpec row_spec(int depth, int tile, int half, int i) {
    constexpr int O1 = 0, O2 = 144, O3 = 192, O4 = 208, O5 = 256, O6 = 272, O7 = 288, O8 = 304, O9 = 320, O10 = 368;
    const RowSpec zero = {-1, 0.f};
    if (tile < 6) {
        const int w = 2 * tile + half;
        if (i < 12) return {O1 + i * NS + w, 1.0f};
        return depth >= 1 ? RowSpec{O4 + (i - 12) * NS + w, INV_SQRT3} : zero;
    }
    if (tile < 8) {
        const int w = 2 * (tile - 6) + half;
        if (i < 12) return {O2 + i * NV + w, INV_SQRT3};
        return depth >= 1 ? RowSpec{O3 + (i - 12) * NV + w, INV_SQRT3} : zero;
    }
    if (depth < 1) return zero;
    if (tile == 8) {
        const int w = half + 2 * (i >> 3), j = i & 7;
        if (j < 4) return {O6 + j * NV + w, 1.0f};
        return depth >= 2 ? RowSpec{O8 + (j - 4) * NV + w, INV_SQRT6} : zero;
    }
    if (tile < 11) {
        const int w = 2 * (tile - 9) + half;
        if (i < 4) return {O5 + i * NV + w, INV_SQRT6};
        if (depth < 2 || i >= 12) return zero;
        if (i < 8) return {O7 + (i - 4) * NV + w, INV_SQRT3};
        return {O10 + (i - 8) * NV + w, 1.0f};
    }
    if (depth < 2 || (tile == 12 && half == 1)) return zero;
    const int w = 8 * (tile - 11) + 4 * half + (i >> 2);
    return {O9 + (i & 3) * NS + w, INV_SQRT3};
}

// Hidden unit behind k-slot q (0..23) of lane half hb of fc.3's contraction: the half's own rows of fc.0's result, in
// the order they sit in its registers.  -1: zero column, -2: the bias column (the lane supplies its scale there).
DEV int hidden_of_slot(int hb, int q) {
    if (q < 16) return 8 * (q >> 2) + (q & 3) + 4 * hb;      // block 0 of fc.0: rows (r & 3) + 8 (r >> 2) + 4 h
    if (hb == 0) return q < 20 ? 32 + (q - 16) : -1;          // block 1, rows 0-3: units 32-35
    return q == 16 ? -2 : -1;
}

DEV void store_frag(u32x4 *dst, const float (&v)[8]) {     // dst[0] = the hi fragment, dst[64] the lo fragment of this lane
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 x = {v[j], v[j + 1]};
        const f16x2 hh = __builtin_convertvector(x, f16x2);
        const f16x2 ll = split_lo_pair(hh, x);
        hi[j] = hh.x; hi[j + 1] = hh.y;
        lo[j] = ll.x; lo[j + 1] = ll.y;
    }
    dst[0] = __builtin_bit_cast(u32x4, hi);
    dst[64] = __builtin_bit_cast(u32x4, lo);
}

DEV void split8(const float (&v)[8], f16x8 &hi, f16x8 &lo) {
    u32x4 H, L;                                   // whole 32-bit words: element-wise inserts into f16x8 cost a v_mov / v_perm each
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 x = {v[j], v[j + 1]};
        const f16x2 hh = __builtin_convertvector(x, f16x2);
        const f16x2 ll = split_lo_pair(hh, x);
        H[j >> 1] = __builtin_bit_cast(unsigned, hh);
        L[j >> 1] = __builtin_bit_cast(unsigned, ll);
    }
    hi = as_f16x8(H);
    lo = as_f16x8(L);
}

DEV f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// D += A(block in LDS: hi at frag[0], lo at frag[64]) . B(hi, lo), smallest terms first
DEV void mfma3(f32x16 &D, const u32x4 *frag, f16x8 bhi, f16x8 blo) {
    const f16x8 ahi = as_f16x8(frag[0]), alo = as_f16x8(frag[64]);
    D = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, D, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, D, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, D, 0, 0, 0);
}

template <int N>
DEV float half_sum(float v) {                         // over the 32 lanes of a lane half
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// The kernel's read-only LDS image: [fc.3: NT blocks x 3 k-steps][hi | lo][64 lanes] | [fc.0: 2 blocks x 3 k-steps][hi | lo][64] |
// scales (fc.3, fc.0, 0, 0) | edge-embedding rows (emb0: 12 x 12, emb3: 12 x 16).
__host__ __device__ constexpr int tiles_of(int depth) { return depth == 0 ? 8 : (depth == 1 ? 11 : 13); }
__host__ __device__ constexpr int image_bytes(int depth) { return (tiles_of(depth) + 2) * 3 * 128 * 16 + (4 + 12 * 12 + 12 * 16) * 4; }

// Built by one workgroup of 64 MF_WAVES threads into `img` (LDS or global memory); `red`: 2 MF_WAVES floats of LDS.
template <int DEPTH, int NW>
DEV void build_image(const codlad_tp_conv_args &a, u32x4 *img, float *red) {
    constexpr int NT = tiles_of(DEPTH);
    u32x4 *A3 = img, *A0 = img + NT * 3 * 128;
    float *scales = reinterpret_cast<float *>(img + (NT + 2) * 3 * 128);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the weight matrices' scales (largest |w| or |b| of fc.3 over the rows this depth uses, of fc.0)
    float S_W3, S_W0;
    {
        constexpr int N3 = DEPTH == 0 ? 192 : (DEPTH == 1 ? 288 : 384);
        float m3 = 0.f, m0 = 0.f;
        for (int i = threadIdx.x; i < N3 * 36; i += 64 * NW) m3 = fmaxf(m3, fabsf(a.fc3_w[i]));
        for (int i = threadIdx.x; i < N3; i += 64 * NW) m3 = fmaxf(m3, fabsf(a.fc3_b[i]));
        for (int i = threadIdx.x; i < 36 * 36; i += 64 * NW) m0 = fmaxf(m0, fabsf(a.fc0_w[i]));
        for (int i = threadIdx.x; i < 36; i += 64 * NW) m0 = fmaxf(m0, fabsf(a.fc0_b[i]));
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            m3 = fmaxf(m3, __shfl_xor(m3, m, 64));
            m0 = fmaxf(m0, __shfl_xor(m0, m, 64));
        }
        if (lane == 0) { red[2 * wave] = m3; red[2 * wave + 1] = m0; }
        __syncthreads();
        for (int w = 0; w < NW; ++w) { m3 = fmaxf(m3, red[2 * w]); m0 = fmaxf(m0, red[2 * w + 1]); }
        S_W3 = pow2_scale(m3, 8);
        S_W0 = pow2_scale(m0, 8);
    }
    if (threadIdx.x < 4) scales[threadIdx.x] = threadIdx.x == 0 ? S_W3 : (threadIdx.x == 1 ? S_W0 : 0.f);
    {
        // emb0 rows: [w(type recv), w(type snd), w(smearing 0..7), bias, 0]  (12 floats), emb3 rows: [w 0..11, bias, 0 0 0] (16)
        float *e0 = scales + 4, *e3 = e0 + 12 * 12;
        for (int i = threadIdx.x; i < 12 * 12; i += 64 * NW) {
            const int o = i / 12, k = i % 12;
            float v = 0.f;
            if (k < 2) v = a.emb_in == 14 ? a.emb0_w[o * 14 + k] : 0.f;
            else if (k < 10) v = a.emb_in == 14 ? a.emb0_w[o * 14 + 4 + k] : a.emb0_w[o * 8 + (k - 2)];
            else if (k == 10) v = a.emb0_b[o];
            e0[i] = v;
        }
        for (int i = threadIdx.x; i < 12 * 16; i += 64 * NW) {
            const int o = i / 16, k = i % 16;
            e3[i] = k < 12 ? a.emb3_w[o * NS + k] : (k == 12 ? a.emb3_b[o] : 0.f);
        }
    }
    // fragment (block, k-step) of lane l = row (l & 31), k-slots 8 (l >> 5) + 0..7
    for (int f = threadIdx.x; f < (NT + 2) * 3 * 64; f += 64 * NW) {
        const int l = f & 63, blk = (f >> 6) / 3, ks = (f >> 6) % 3;
        const int row = l & 31, hb = l >> 5;
        float v[8];
        if (blk < NT) {                                             // fc.3
            const RowSpec rs = row_spec(DEPTH, blk, (row >> 2) & 1, 4 * (row >> 3) + (row & 3));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = hidden_of_slot(hb, 8 * ks + j);
                float w = 0.f;
                if (rs.idx >= 0 && m != -1) w = m >= 0 ? a.fc3_w[rs.idx * 36 + m] : a.fc3_b[rs.idx];
                v[j] = (S_W3 * rs.factor) * w;
            }
            store_frag(A3 + (blk * 3 + ks) * 128 + l, v);
        } else {                                                    // fc.0: rows = hidden units, k-slot = input index, 36 = bias
            const int m = 32 * (blk - NT) + row;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * hb + j;
                float w = 0.f;
                if (m < 36 && k <= 36) w = k < 36 ? a.fc0_w[m * 36 + k] : a.fc0_b[m];
                v[j] = S_W0 * w;
            }
            store_frag(A0 + ((blk - NT) * 3 + ks) * 128 + l, v);
        }
    }
}

template <int DEPTH>
__global__ __launch_bounds__(64 * MF_WAVES) void tp_conv_pack_kernel(codlad_tp_conv_args a, u32x4 *img) {
    __shared__ float red[2 * MF_WAVES];
    build_image<DEPTH, MF_WAVES>(a, img, red);
}

template <int DEPTH>
__global__ __launch_bounds__(64 * conv_waves(DEPTH)) void tp_conv_mfma_kernel(codlad_tp_conv_args a) {
    constexpr int NW = conv_waves(DEPTH);
    constexpr int NT = tiles_of(DEPTH);
    constexpr int D_OUT = width_of(DEPTH + 1);
    // the read-only image (build_image), then the waves' output rows (48 floats each) and 2 NW floats for build_image
    extern __shared__ __align__(16) u32x4 lds[];
    const u32x4 *A3 = lds, *A0 = lds + NT * 3 * 128;
    float *out_sh = reinterpret_cast<float *>(lds + image_bytes(DEPTH) / 16);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;

    // ---- the read-only image: packed here by every workgroup, or copied when the caller packed it once (a.packed)
    if (a.packed) {
        const u32x4 *src = static_cast<const u32x4 *>(a.packed);
        constexpr int N16 = image_bytes(DEPTH) / 16, PER = (N16 + 64 * NW - 1) / (64 * NW);
        u32x4 v[PER];                                       // all of a thread's loads in flight, then the stores
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 64 * NW;
            if (i < N16) v[k] = src[i];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 64 * NW;
            if (i < N16) lds[i] = v[k];
        }
    } else {
        build_image<DEPTH, NW>(a, lds, out_sh + NW * 48);
    }
    __syncthreads();
    const float *scales = reinterpret_cast<const float *>(lds + (NT + 2) * 3 * 128);

    // the edge-embedding MLP's weights, read back as wave-uniform (broadcast) 16-byte LDS reads: through the scalar cache
    // every output row cost a round trip per step with nothing to overlap it
    //   emb0 rows: [w(type recv), w(type snd), w(smearing 0..7), bias, 0]  (12 floats), emb3 rows: [w 0..11, bias, 0 0 0] (16)
    const float S_W3 = scales[0], S_W0 = scales[1];
    const float *e0w = scales + 4, *e3w = e0w + 12 * 12;     // 16-byte aligned
    const float step = a.smear_stop / 7.0f, coeff = -0.5f / (step * step);
    float *my_out = out_sh + wave * 48;
    // path coefficients sqrt((2 l_out + 1) / sum of mul_in1 over the paths into the same output block)
    constexpr float C0E = DEPTH == 0 ? 0.28867513459481287f : 0.25f;
    constexpr float C1O = DEPTH == 0 ? 0.5f : (DEPTH == 1 ? 0.38729833462074170f : 0.35355339059327379f);
    constexpr float C1E = DEPTH == 1 ? 0.86602540378443865f : 0.5f;
    constexpr float C0O = 0.5f;

    for (int n = blockIdx.x * NW + wave; n < a.n_recv; n += gridDim.x * NW) {      // wave-uniform
        const int e0 = __builtin_amdgcn_readfirstlane(a.ptr[n]), e1 = __builtin_amdgcn_readfirstlane(a.ptr[n + 1]);
        const float xr = a.xyz_recv[3 * n], yr = a.xyz_recv[3 * n + 1], zr = a.xyz_recv[3 * n + 2];
        float hr[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) hr[k] = a.h_recv[(size_t)n * a.d_recv + k];
        const float typ_r = a.typ_recv ? a.typ_recv[n] : 0.f;
        // per-lane partial sums of this half's output channels
        float s0e[6], s0o[8];
        Vec3 s1o[2], s1e[2];
#pragma unroll
        for (int t = 0; t < 6; ++t) s0e[t] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) s0o[t] = 0.f;
        s1o[0] = s1o[1] = s1e[0] = s1e[1] = {0.f, 0.f, 0.f};

        for (int base = e0; base < e1; base += 32) {
            const bool live = base + c < e1;
            const int s = (CODLAD_TP_ABLATE & 2) ? 0 : (live ? a.snd[base + c] : a.snd[e0]);
            // geometry: r = sign (x_snd - x_recv), |r|, Y(r / |r|)
            const float rx = a.r_sign * (a.xyz_snd[3 * s] - xr), ry = a.r_sign * (a.xyz_snd[3 * s + 1] - yr),
                        rz = a.r_sign * (a.xyz_snd[3 * s + 2] - zr);
            const float d = sqrtf(rx * rx + ry * ry + rz * rz);
            const float inv = 1.0f / fmaxf(d, 1e-12f);                    // F.normalize
            const float ux = rx * inv, uy = ry * inv, uz = rz * inv;
            const Vec3 y1 = {1.7320508075688772f * ux, 1.7320508075688772f * uy, 1.7320508075688772f * uz};
            float y2[5];
            {
                const float s5 = 2.2360679774997896f, s15 = 3.8729833462074170f;          // sqrt 5, sqrt 15
                y2[0] = s15 * ux * uz;
                y2[1] = s15 * ux * uy;
                y2[2] = s5 * (uy * uy - 0.5f * (ux * ux + uz * uz));
                y2[3] = s15 * uy * uz;
                y2[4] = 0.5f * s15 * (uz * uz - ux * ux);
            }
            // fc input [edge embedding (12) | scalars | scalars | 1], scaled
            const float *hs = a.h_snd + (size_t)s * a.d_snd;
            float x0[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) x0[k] = hs[k];
            float in48[48];
            {
                float sm[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float t = d - step * (float)k;
                    sm[k] = expf(coeff * t * t);
                }
                float h1[NS];
                const float typ_s = a.emb_in == 14 ? a.typ_snd[s] : 0.f;   // [z_recv, z_snd, 0 0 0 0, smearing] or the smearing alone
#pragma unroll
                for (int o = 0; o < NS; ++o) {
                    const float *w = e0w + o * 12;
                    float acc = w[10];
                    acc = fmaf(typ_r, w[0], acc);
                    acc = fmaf(typ_s, w[1], acc);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = fmaf(sm[k], w[2 + k], acc);
                    h1[o] = fmaxf(acc, 0.f);
                }
#pragma unroll
                for (int o = 0; o < NS; ++o) {
                    const float *w = e3w + o * 16;
                    float acc = w[12];
#pragma unroll
                    for (int k = 0; k < NS; ++k) acc = fmaf(h1[k], w[k], acc);
                    in48[o] = acc;
                }
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                in48[12 + k] = a.attr_recv_first ? hr[k] : x0[k];
                in48[24 + k] = a.attr_recv_first ? x0[k] : hr[k];
            }
            float s_in;                                                     // this edge's input scale (the same in both halves)
            {
                float m = 1.0f;                                             // the bias slot's 1
#pragma unroll
                for (int k = 0; k < 36; ++k) m = fmaxf(m, fabsf(in48[k]));
                s_in = pow2_scale(m, 3);
#pragma unroll
                for (int k = 0; k < 36; ++k) in48[k] *= s_in;
            }
            in48[36] = s_in;
#pragma unroll
            for (int k = 37; k < 48; ++k) in48[k] = 0.f;

            // ---- fc.0 on the matrix pipe: hidden^T [unit][edge]; this half supplies k-slots 16 ks + 8 h + 0..7
            f32x16 H0 = zero16(), H1 = zero16();
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = h ? in48[16 * ks + 8 + j] : in48[16 * ks + j];
                f16x8 bhi, blo;
                split8(v, bhi, blo);
                mfma3(H0, A0 + ks * 128 + lane, bhi, blo);
                mfma3(H1, A0 + (3 + ks) * 128 + lane, bhi, blo);
            }
            // ReLU, rescale (H = s_in S_W0 pre-activation), the edge's hidden scale, split: the B operand of fc.3, k-slot
            // q = 8 ks + j
            f16x8 hhi[3], hlo[3];
            float s_hid;
            {
                const float rs = 1.0f / (s_in * S_W0);                      // a power of two
                float m = 1.0f;                                             // the bias slot's 1
#pragma unroll
                for (int j = 0; j < 16; ++j) { H0[j] = rs * fmaxf(H0[j], 0.f); m = fmaxf(m, H0[j]); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { H1[j] = h ? 0.f : rs * fmaxf(H1[j], 0.f); m = fmaxf(m, H1[j]); }
                m = fmaxf(m, __shfl_xor(m, 32, 64));                        // the other half holds the rest of this edge's units
                s_hid = pow2_scale(m, 3);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = s_hid * H0[j];
                split8(v, hhi[0], hlo[0]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = s_hid * H0[8 + j];
                split8(v, hhi[1], hlo[1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = h ? (j == 0 ? s_hid : 0.f) : s_hid * H1[j];
#pragma unroll
                for (int j = 4; j < 8; ++j) v[j] = 0.f;
                split8(v, hhi[2], hlo[2]);
            }

            // ---- inputs of the paths, times fc.3's descale - or zero for a lane without an edge: its weights then
            // multiply nothing
            const float lv = live ? 1.0f / (s_hid * S_W3) : 0.f;
            float p0[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) p0[u] = lv * x0[u];
            // (the derived inputs - dot, cross and Wigner products with Y1 / Y2 - are formed right before the block that
            // uses them: all of them at once do not fit the register file next to the operands)
            Vec3 v1[NV], v2[NV];
            if (DEPTH >= 1) {
#pragma unroll
                for (int u = 0; u < NV; ++u) v1[u] = {lv * hs[12 + 3 * u], lv * hs[13 + 3 * u], lv * hs[14 + 3 * u]};
            }

            // ---- fc.3 block by block, each contracted at once with the inputs next to it
            // (one block's fragments and results in flight at a time: the scheduling barrier keeps the compiler from pulling
            // more into flight than the register file holds.  Loading block t + 1's fragments during block t was tried: no gain -
            // the step is not waiting on LDS - and, with the loads landing in the registers the preceding matrix instruction
            // still reads, results that changed from run to run; tools/ablate_encoder.sh has the measurements' recipe.)
            auto block = [&](int t) {
                __builtin_amdgcn_sched_barrier(0);
                f32x16 D = zero16();
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) mfma3(D, A3 + (t * 3 + ks) * 128 + lane, hhi[ks], hlo[ks]);
                return D;
            };
#pragma unroll
            for (int t = 0; t < ((CODLAD_TP_ABLATE & 1) ? 0 : 6); ++t) {     // 12x0e
                const f32x16 D = block(t);
                float r = 0.f;
#pragma unroll
                for (int u = 0; u < NS; ++u) r = fmaf(D[u], p0[u], r);
                if (DEPTH >= 1) {
#pragma unroll
                    for (int u = 0; u < NV; ++u) r = fmaf(D[12 + u], dot3(v1[u], y1), r);
                }
                s0e[t] += r;
            }
#pragma unroll
            for (int t = 0; t < ((CODLAD_TP_ABLATE & 1) ? 0 : 2); ++t) {     // 4x1o: scalars x Y1, vectors
                const f32x16 D = block(6 + t);
                float sc = 0.f;
#pragma unroll
                for (int u = 0; u < NS; ++u) sc = fmaf(D[u], p0[u], sc);
                Vec3 r = {sc * y1.x, sc * y1.y, sc * y1.z};
                if (DEPTH >= 1) {
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        r.x = fmaf(D[12 + u], v1[u].x, r.x);
                        r.y = fmaf(D[12 + u], v1[u].y, r.y);
                        r.z = fmaf(D[12 + u], v1[u].z, r.z);
                    }
                }
                s1o[t].x += r.x; s1o[t].y += r.y; s1o[t].z += r.z;
            }
            if (DEPTH >= 1 && !(CODLAD_TP_ABLATE & 1)) {
                if (DEPTH >= 2) {                                           // the second vector block is first needed here
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < NV; ++u) v2[u] = {lv * hs[24 + 3 * u], lv * hs[25 + 3 * u], lv * hs[26 + 3 * u]};
                }
                {
                    const f32x16 D = block(8);                              // 4x1o: (v1 x Y2), (v2 x Y1)
                    Vec3 q6[NV], c8[NV];
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        q6[u] = w121(v1[u], y2);
                        if (DEPTH >= 2) c8[u] = cross3(v2[u], y1);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        Vec3 r = {0.f, 0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < NV; ++u) {
                            r.x = fmaf(D[8 * q + u], q6[u].x, r.x);
                            r.y = fmaf(D[8 * q + u], q6[u].y, r.y);
                            r.z = fmaf(D[8 * q + u], q6[u].z, r.z);
                        }
                        if (DEPTH >= 2) {
#pragma unroll
                            for (int u = 0; u < NV; ++u) {
                                r.x = fmaf(D[8 * q + 4 + u], c8[u].x, r.x);
                                r.y = fmaf(D[8 * q + 4 + u], c8[u].y, r.y);
                                r.z = fmaf(D[8 * q + 4 + u], c8[u].z, r.z);
                            }
                        }
                        s1o[q].x += r.x; s1o[q].y += r.y; s1o[q].z += r.z;
                    }
                }
                Vec3 c5[NV], q10[NV];
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    c5[u] = cross3(v1[u], y1);
                    if (DEPTH >= 2) q10[u] = w121(v2[u], y2);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {                               // 4x1e
                    const f32x16 D = block(9 + t);
                    Vec3 r = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        r.x = fmaf(D[u], c5[u].x, r.x);
                        r.y = fmaf(D[u], c5[u].y, r.y);
                        r.z = fmaf(D[u], c5[u].z, r.z);
                    }
                    if (DEPTH >= 2) {
#pragma unroll
                        for (int u = 0; u < NV; ++u) {
                            r.x = fmaf(D[4 + u], v2[u].x, fmaf(D[8 + u], q10[u].x, r.x));
                            r.y = fmaf(D[4 + u], v2[u].y, fmaf(D[8 + u], q10[u].y, r.y));
                            r.z = fmaf(D[4 + u], v2[u].z, fmaf(D[8 + u], q10[u].z, r.z));
                        }
                    }
                    s1e[t].x += r.x; s1e[t].y += r.y; s1e[t].z += r.z;
                }
            }
            if (DEPTH >= 2 && !(CODLAD_TP_ABLATE & 1)) {
                float d9[NV];
#pragma unroll
                for (int u = 0; u < NV; ++u) d9[u] = dot3(v2[u], y1);
#pragma unroll
                for (int t = 0; t < 2; ++t) {                               // 12x0o
                    const f32x16 D = block(11 + t);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float r = 0.f;
#pragma unroll
                        for (int u = 0; u < NV; ++u) r = fmaf(D[4 * q + u], d9[u], r);
                        s0o[4 * t + q] += r;
                    }
                }
            }
        }

        // ---- sums over the half's 32 edges-in-flight, channel by channel; lane 0 of each half files its channels
        // half h owns: 0e channels 2 t + h; 1o / 1e channels 2 t + h (block 8 adds to 1o channel h + 2 q = the same two);
        // 0o channels 4 h + q (t = 0) and 8 + q (t = 1, half 0 only)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const float v = half_sum<32>(s0e[t]);
            if (c == 0) my_out[2 * t + h] = C0E * v;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float vx = half_sum<32>(s1o[t].x), vy = half_sum<32>(s1o[t].y), vz = half_sum<32>(s1o[t].z);
            if (c == 0) {
                float *o = my_out + 12 + 3 * (2 * t + h);
                o[0] = C1O * vx; o[1] = C1O * vy; o[2] = C1O * vz;
            }
        }
        if (DEPTH >= 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float vx = half_sum<32>(s1e[t].x), vy = half_sum<32>(s1e[t].y), vz = half_sum<32>(s1e[t].z);
                if (c == 0) {
                    float *o = my_out + 24 + 3 * (2 * t + h);
                    o[0] = C1E * vx; o[1] = C1E * vy; o[2] = C1E * vz;
                }
            }
        }
        if (DEPTH >= 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = half_sum<32>(s0o[q]);
                if (c == 0 && (q < 4 || h == 0)) my_out[36 + (q < 4 ? 4 * h + q : 8 + (q - 4))] = C0O * v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // out[n] = (accumulate ? out[n] : pad(h_recv[n])) + sum / degree
        {
            const int deg = e1 - e0;
            const float scale = deg > 0 ? 1.0f / (float)deg : 0.f;
            float *o = a.out + (size_t)n * D_OUT;
            if (lane < D_OUT) {
                const float base = a.accumulate ? o[lane] : (lane < a.d_recv ? a.h_recv[(size_t)n * a.d_recv + lane] : 0.f);
                o[lane] = base + (deg > 0 ? my_out[lane] : 0.f) * scale;
            }
        }
        __builtin_amdgcn_wave_barrier();                                    // my_out is rewritten by the next node
    }
}

template <int DEPTH>
void launch(const codlad_tp_conv_args &a, hipStream_t st) {
    constexpr int NW = conv_waves(DEPTH);
    const size_t lds = (size_t)image_bytes(DEPTH) + (NW * 48 + 2 * NW) * sizeof(float);
    static bool raised = false;
    if (!raised) {
        set_max_lds(reinterpret_cast<const void *>(tp_conv_mfma_kernel<DEPTH>), lds);
        raised = true;
    }
    const int wanted = (a.n_recv + NW - 1) / NW;
    const int grid = wanted < num_cu() ? wanted : num_cu();
    hipLaunchKernelGGL((tp_conv_mfma_kernel<DEPTH>), dim3(grid), dim3(64 * NW), lds, st, a);
}

}  // namespace

void launch_tp_conv_mfma(const codlad_tp_conv_args &a, hipStream_t st) {
    if (a.depth == 0) launch<0>(a, st);
    else if (a.depth == 1) launch<1>(a, st);
    else launch<2>(a, st);
}

int tp_conv_image_bytes(int depth) { return image_bytes(depth); }

void launch_tp_conv_pack(const codlad_tp_conv_args &a, void *image, hipStream_t st) {
    u32x4 *img = static_cast<u32x4 *>(image);
    if (a.depth == 0) hipLaunchKernelGGL((tp_conv_pack_kernel<0>), dim3(1), dim3(64 * MF_WAVES), 0, st, a, img);
    else if (a.depth == 1) hipLaunchKernelGGL((tp_conv_pack_kernel<1>), dim3(1), dim3(64 * MF_WAVES), 0, st, a, img);
    else hipLaunchKernelGGL((tp_conv_pack_kernel<2>), dim3(1), dim3(64 * MF_WAVES), 0, st, a, img);
}
