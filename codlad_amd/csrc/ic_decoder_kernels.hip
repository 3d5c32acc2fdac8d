// Row 9: the IC decoder, VAE.decoder = map_out + IC_Decoder[_angle].forward
// (reference models/vae_model.py:759-764, 375-412, 467-503; Dense / radial basis / envelope: models/gcn_nn.py).
//
// Two kinds of kernel, by what the work is made of:
//  * per-residue dense layers (40..53-wide Linear + swish chains): one LANE per residue, 64 residues per
//    workgroup, the outputs of a layer dealt to its 4 waves.  A lane keeps the layer's input vector in registers,
//    the weight row of the output being produced is wave-uniform and comes through the scalar cache (constant
//    address space -> s_load), so a product is one v_fmac with an SGPR operand and nothing is shuffled between
//    lanes.  Vectors pass from layer to layer through LDS columns, activated once by the wave that produced them.
//  * the distance-conditioned messages (per directed CG edge: 15 sines, a 15 -> 40 Linear, an envelope, and a
//    40-wide multiply-accumulate into the receiving residue): one WAVE (= one workgroup) per receiving residue.  Per chunk of 64
//    incoming edges, lane = edge evaluates the radial basis and the Linear (again with scalar weight rows) and
//    leaves the 40 filter values in LDS; then lane = feature walks the edges in CSR order, multiplies by the
//    sender's row (one coalesced 160-byte read) and accumulates.
// Every sum keeps the order of the reference's ops (fma chain over k, then + bias; edges in scatter order), so
// results do not depend on which residues share a launch.
//
// Scratch (caller's float [M][200], opaque): S^T [40][M] | V [M][40] | phi_a [M][40] | phi_b [M][40].
// (V node-major since round 3: a receiving residue's 40 sums leave as one coalesced 160-byte store; feature-major, a wave
// wrote 40 dwords a column apart - 101 MB of write traffic per launch for 5.7 MB of sums.)
#include "common.h"
#include "../../include/codlad_hip.h"

int num_cu();             // denoiser_kernels.hip
int dec_edge_variant();   // denoiser_kernels.hip: CODLAD_OPT_DEC_EDGE_VARIANT

#define DF 40
#define PI_F 3.14159265358979323846f
#define CG_CUTOFF 21.0f

typedef const __attribute__((address_space(4))) float *kfloat_p;
DEV kfloat_p as_uniform(const float *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (kfloat_p)p;
#pragma clang diagnostic pop
}

DEV float swishf(float v) { return v * (1.0f / (1.0f + expf(-v))); }

// One Linear layer for the 64 residues of a workgroup, its outputs dealt round-robin to the 4 waves: output c = wave + 4 i
// (i = 0, 1, ...) is emit(i, c, b[c] + sum_k W[c][k] * in[k]).  `in`: this lane's LDS column (stride 64 floats), already
// activated if the layer's input is.
//
// Where the weights come from (round 3): the layer's [OUT][IN] matrix and bias are STAGED in LDS - fetched by the whole
// workgroup as one coalesced read into registers while the previous layer computes (`fetch`), written to the staging
// buffer between two barriers (`commit`), and read back as wave-uniform (broadcast) ds_read_b128.  Before, each output row
// came through the scalar cache, one s_load + s_waitcnt per row with nothing to overlap it (scalar loads return out of
// order, so a wave cannot keep a second row in flight): 10 exposed L2 round trips per layer per wave, 27 us for a kernel
// with 3 us of arithmetic.  The arithmetic is unchanged: one fma chain over k in index order, then + bias.
template <int IN> struct RowStride { static constexpr int v = (IN + 3) & ~3; };      // rows start 16-byte aligned
template <int IN, int OUT> struct Staged {
    static constexpr int STRIDE = RowStride<IN>::v, FLOATS = STRIDE * OUT + OUT, PER_THREAD = (FLOATS + 255) / 256;
    float r[PER_THREAD];
    DEV void fetch(const float *W, const float *b) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int idx = (int)threadIdx.x + 256 * i, row = idx / STRIDE, k = idx - row * STRIDE;
            float v = 0.f;
            if (idx < STRIDE * OUT) { if (k < IN) v = W[row * IN + k]; }
            else if (idx < FLOATS) v = b[idx - STRIDE * OUT];
            r[i] = v;
        }
    }
    // the caller puts a barrier before (nobody still reads the buffer) and after (everybody sees it)
    DEV void commit(float *wbuf) const {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int idx = (int)threadIdx.x + 256 * i;
            if (idx < FLOATS) wbuf[idx] = r[i];
        }
    }
};

template <int IN, int OUT, typename Emit>
DEV void wg_dense(const float *wbuf, const float *in, int wave, Emit emit) {
    constexpr int STRIDE = RowStride<IN>::v;
    float x[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) x[k] = in[k * 64];
#pragma unroll
    for (int i = 0; i < (OUT + 3) / 4; ++i) {
        const int c = wave + 4 * i;
        if (c < OUT) {
            const float *wr = wbuf + c * STRIDE;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < IN; ++k) acc = fmaf(x[k], wr[k], acc);
            emit(i, c, acc + wbuf[STRIDE * OUT + c]);
        }
    }
}
#define STAGE_FLOATS(IN, OUT) (((IN + 3) & ~3) * (OUT) + (OUT))

struct Scratch {
    float *S, *V, *phi0;
    size_t plane;
    DEV float *phi(int which) const { return phi0 + (size_t)(which & 1) * plane; }   // (an array member indexed at run time lives in scratch memory)
};
DEV Scratch scratch_of(float *scr, int M) {
    Scratch s;
    s.plane = (size_t)DF * M;
    s.S = scr;
    s.V = scr + s.plane;
    s.phi0 = scr + 2 * s.plane;
    return s;
}

// phi of a message block: inv_dense = Dense -> swish -> Dense on the raw state  (vae_model.py:360-363).
// `first` holds inv0's weights, fetched by the caller; the caller's last barrier is behind every reader of wbuf.
DEV void write_phi(const codlad_decoder_weights &w, int blk, Staged<DF, DF> &first, float *wbuf, const float *s_col,
                   float *t_col, float *phi, int n, bool active, int wave) {
    first.commit(wbuf);
    Staged<DF, DF> second;
    second.fetch(w.inv1_w[blk], w.inv1_b[blk]);
    __syncthreads();
    wg_dense<DF, DF>(wbuf, s_col, wave, [&](int, int c, float v) { t_col[c * 64] = swishf(v); });
    __syncthreads();
    second.commit(wbuf);
    __syncthreads();
    wg_dense<DF, DF>(wbuf, t_col, wave, [&](int, int c, float v) {
        if (active) phi[(size_t)n * DF + c] = v;
    });
}

// S = cat(map_out(z_q), res_embed[z])  (vae_model.py:759-764, 380-383), phi for block 0
__global__ __launch_bounds__(256) void dec_init_kernel(codlad_decoder_weights w, const float *z_q,
                                                      const int32_t *cg_z, int M, float *scr) {
    __shared__ float col[2 * DF][64];
    __shared__ __attribute__((aligned(16))) float wbuf[STAGE_FLOATS(DF, DF)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = blockIdx.x * 64 + lane;
    const bool active = n < M;
    const int nn = active ? n : M - 1;
    const Scratch sc = scratch_of(scr, M);
    float *s_col = &col[0][lane], *t_col = &col[DF][lane];
    Staged<DF, DF> inv0;
    inv0.fetch(w.inv0_w[0], w.inv0_b[0]);
    const int z = cg_z[nn];
    const bool mapped = w.map_out_w != nullptr;          // no map_out (the C2 model): z_q is the 36-wide latent itself
    const float q0 = mapped ? z_q[3 * nn] : 0.f, q1 = mapped ? z_q[3 * nn + 1] : 0.f, q2 = mapped ? z_q[3 * nn + 2] : 0.f;
    if (mapped) {
        Staged<3, 36> mo;
        mo.fetch(w.map_out_w, w.map_out_b);
        mo.commit(wbuf);
    }
    __syncthreads();
    for (int c = wave; c < DF; c += 4) {
        float s;
        if (c < 36) s = mapped ? fmaf(q2, wbuf[4 * c + 2], fmaf(q1, wbuf[4 * c + 1], q0 * wbuf[4 * c])) + wbuf[4 * 36 + c]   // F.linear
                               : z_q[(size_t)36 * nn + c];
        else s = w.res_embed[z * 4 + (c - 36)];
        s_col[c * 64] = s;
        if (active) sc.S[(size_t)c * M + n] = s;
    }
    __syncthreads();
    write_phi(w, 0, inv0, wbuf, s_col, t_col, sc.phi(0), n, active, wave);
}

// a / d the way the compiler's IEEE expansion computes it (v_rcp, one refinement of the reciprocal, two of the
// quotient) with the reciprocal shared by the 15 numerators; d is a distance >= 1.7e-4, |a| <= 1: no range scaling.
DEV float refined_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
DEV float div_by(float a, float d, float r) {
    float q = a * r;
    q = fmaf(fmaf(-d, q, a), r, q);
    return fmaf(fmaf(-d, q, a), r, q);
}

// V[n] = sum over incoming edges (j -> n) of phi_j * (dist_embed(rbf(d)) * envelope(d))   (vae_model.py:483-488)
#define WS_STRIDE 41
#ifndef EDGE_WAVES
#define EDGE_WAVES 1   // receivers per workgroup: 1 lets 15 waves share a CU's LDS (cfg 5: 1.04 ms against 1.15 with 4)
#endif
__global__ __launch_bounds__(64 * EDGE_WAVES) void dec_edge_exact_kernel(codlad_decoder_weights w, int blk, const float *cg_xyz,
                                                      const int32_t *csr_ptr, const int32_t *csr_src, int M,
                                                      float *scr) {
    __shared__ float filt[EDGE_WAVES][64 * WS_STRIDE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = __builtin_amdgcn_readfirstlane(blockIdx.x * EDGE_WAVES + wave);
    if (n >= M) return;                                // waves are independent: no workgroup barrier below
    const Scratch sc = scratch_of(scr, M);
    const float *phi_in = sc.phi(blk);
    float *my = filt[wave];
    const float xi = cg_xyz[3 * n], yi = cg_xyz[3 * n + 1], zi = cg_xyz[3 * n + 2];
    const int c = lane < DF ? lane : 0;
    kfloat_p Wd = as_uniform(w.dist_w[blk]), bd = as_uniform(w.dist_b[blk]);
    float v = 0.f;
    const int e0 = __builtin_amdgcn_readfirstlane(csr_ptr[n]), e1 = __builtin_amdgcn_readfirstlane(csr_ptr[n + 1]);
    for (int base = e0; base < e1; base += 64) {
        const int cnt = e1 - base < 64 ? e1 - base : 64;
        int j = 0;
        if (lane < cnt) {
            j = csr_src[base + lane];
            // preprocess_r (gcn_nn.py:66-70): eps added per component
            const float rx = cg_xyz[3 * j] - xi, ry = cg_xyz[3 * j + 1] - yi, rz = cg_xyz[3 * j + 2] - zi;
            const float d = sqrtf(((rx * rx + 1e-8f) + (ry * ry + 1e-8f)) + (rz * rz + 1e-8f));
            // CosineEnvelope and PainnRadialBasis (gcn_nn.py:231-255): sin(n pi d / cutoff) / d, 0 beyond the cutoff
            float env = 0.5f * (cosf(PI_F * d / CG_CUTOFF) + 1.0f);
            if (d >= CG_CUTOFF) env = 0.f;
            const float rd = refined_rcp(d);
            float rbf[15];
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const float coef = ((float)(k + 1) * PI_F) / CG_CUTOFF;
                rbf[k] = div_by(sinf(coef * d), d, rd);
                if (d >= CG_CUTOFF) rbf[k] = 0.f;
            }
            for (int f = 0; f < DF; ++f) {
                kfloat_p wr = Wd + f * 15;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 15; ++k) acc = fmaf(rbf[k], wr[k], acc);
                my[lane * WS_STRIDE + f] = (acc + bd[f]) * env;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int e = 0;
        for (; e + 8 <= cnt; e += 8) {                  // eight sender rows in flight, summed in edge order
            float ph[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float *row = phi_in + (size_t)__builtin_amdgcn_readlane(j, e + u) * DF;
                ph[u] = row[c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v += ph[u] * my[(e + u) * WS_STRIDE + c];
        }
        for (; e < cnt; ++e) {
            const float *row = phi_in + (size_t)__builtin_amdgcn_readlane(j, e) * DF;
            v += row[c] * my[e * WS_STRIDE + c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane < DF) sc.V[(size_t)n * DF + lane] = v;
}

// The same message sum with the 15 -> 40 filter on the matrix pipe (default since round 3; the kernel above stays
// selectable, CODLAD_OPT_DEC_EDGE_VARIANT = 1).  Per chunk of 64 incoming edges, lane = edge:
//   * d, the envelope, ONE sine / cosine pair and the three-term recurrence sin((k+1) x) = 2 cos x sin(k x) - sin((k-1) x)
//     for the 15 radial basis values (x = pi d / cutoff in (0, pi): the recurrence's error grows like k / sin x, which only
//     matters towards x = pi, where the envelope has already taken the edge's weight to zero) - 15 library sines before;
//   * u[k] = 16 env rbf[k] (k < 15), u[15] = 16 env: the envelope and the bias ride inside the contraction,
//     filter[f] env = sum_k W'[f][k] u[k] / 256 with W' = 16 [dist_w | dist_b] (the powers of two keep the fp16 `lo` halves
//     of these small numbers out of the subnormal range and leave exactly at the end);
//   * u split into fp16 hi + lo halves (22 bits, as in the denoiser's contractions) and contracted with the equally
//     split W' by v_mfma_f32_32x32x16_f16, three products per fp32 product, fp32 accumulation: 12 matrix instructions per
//     64 edges where 600 v_fmac per edge and 40 LDS stores per edge stood before.  The edges are the A operand's rows,
//     so the result arrives with lane = feature and the 32 edges of a group in registers - the layout the accumulation
//     wants: per edge one coalesced 128-byte read of the sender's row per lane half and one multiply-add.
// A lane half sums its own 16 edges of a group in register order, the halves are added at the end: a fixed order per
// receiving residue, whatever shares the launch (the reference's scatter order is not reproduced bit for bit; nor
// was it before, to the last ulp of sinf).
DEV void split_u(f32x2 x, f16x2 &hi, f16x2 &lo) {
    hi = __builtin_convertvector(x, f16x2);
    lo = split_lo_pair(hi, x);
}

__global__ __launch_bounds__(64, 4) void dec_edge_kernel(codlad_decoder_weights w, int blk, const float *cg_xyz,
                                                       const int32_t *csr_ptr, const int32_t *csr_src, int M,
                                                       float *scr) {
    __shared__ int jsh[64];
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const Scratch sc = scratch_of(scr, M);
    const float *phi_in = sc.phi(blk);
    // B operand: lane (feature 32 b + c, k group h) holds W'[feature][8 h .. 8 h + 7], split; features >= 40 are zero
    f16x8 whi[2], wlo[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int f = 32 * b + c;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 v = {0.f, 0.f};
            if (f < DF) {
                const int k0 = 8 * h + i;
                v.x = 16.0f * w.dist_w[blk][f * 15 + k0];                                  // k0 <= 14
                v.y = 16.0f * (k0 + 1 < 15 ? w.dist_w[blk][f * 15 + k0 + 1] : w.dist_b[blk][f]);
            }
            f16x2 hh, ll;
            split_u(v, hh, ll);
            whi[b][i] = hh.x; whi[b][i + 1] = hh.y;
            wlo[b][i] = ll.x; wlo[b][i + 1] = ll.y;
        }
    }
    for (int n = blockIdx.x; n < M; n += gridDim.x) {          // wave-uniform
        const float xi = cg_xyz[3 * n], yi = cg_xyz[3 * n + 1], zi = cg_xyz[3 * n + 2];
        const int e0 = __builtin_amdgcn_readfirstlane(csr_ptr[n]), e1 = __builtin_amdgcn_readfirstlane(csr_ptr[n + 1]);
        float v0 = 0.f, v1 = 0.f;                              // features c (block 0) and 32 + c (block 1, c < 8)
        for (int base = e0; base < e1; base += 64) {
            const int cnt = e1 - base < 64 ? e1 - base : 64;
            const bool live = lane < cnt;
            const int j = live ? csr_src[base + lane] : n;
            // preprocess_r (gcn_nn.py:66-70): eps added per component
            const float rx = cg_xyz[3 * j] - xi, ry = cg_xyz[3 * j + 1] - yi, rz = cg_xyz[3 * j + 2] - zi;
            const float d = sqrtf(((rx * rx + 1e-8f) + (ry * ry + 1e-8f)) + (rz * rz + 1e-8f));
            // CosineEnvelope and PainnRadialBasis (gcn_nn.py:231-271): sin(n pi d / cutoff) / d, both 0 beyond the cutoff
            const float x = (PI_F / CG_CUTOFF) * d;
            float s1, c1;
            sincosf(x, &s1, &c1);
            const bool inside = live && d < CG_CUTOFF;
            const float env16 = inside ? 8.0f * (c1 + 1.0f) : 0.f;            // 16 * 0.5 (cos + 1)
            const float g = env16 * refined_rcp(d);                          // 16 env / d
            const float two_c = c1 + c1;
            float sk = s1, skm = 0.f;
            f16x8 ahi[2], alo[2];                                            // k 0-7, k 8-15 of this lane's edge
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                f32x2 u;
                u.x = sk * g;                                                // sin((k+1) x) / d, enveloped
                float sn = fmaf(two_c, sk, -skm);
                skm = sk; sk = sn;
                if (k + 1 < 15) {
                    u.y = sk * g;
                    sn = fmaf(two_c, sk, -skm);
                    skm = sk; sk = sn;
                } else {
                    u.y = env16;                                             // the bias column
                }
                f16x2 hh, ll;
                split_u(u, hh, ll);
                ahi[k >> 3][k & 7] = hh.x; ahi[k >> 3][(k & 7) + 1] = hh.y;
                alo[k >> 3][k & 7] = ll.x; alo[k >> 3][(k & 7) + 1] = ll.y;
            }
            // A operand of edge group g (edges 32 g .. 32 g + 31): lanes 0-31 its k 0-7, lanes 32-63 its k 8-15.  Lanes
            // 0-31 hold group 0's edges, lanes 32-63 group 1's: swap the upper half of the k 0-7 registers with the lower
            // half of the k 8-15 registers (v_permlane32_swap).
            u32x4 a0h = __builtin_bit_cast(u32x4, ahi[0]), a1h = __builtin_bit_cast(u32x4, ahi[1]);
            u32x4 a0l = __builtin_bit_cast(u32x4, alo[0]), a1l = __builtin_bit_cast(u32x4, alo[1]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                auto r = __builtin_amdgcn_permlane32_swap(a0h[q], a1h[q], false, false);
                a0h[q] = r[0]; a1h[q] = r[1];
                auto t = __builtin_amdgcn_permlane32_swap(a0l[q], a1l[q], false, false);
                a0l[q] = t[0]; a1l[q] = t[1];
            }
            const f16x8 Ahi[2] = {as_f16x8(a0h), as_f16x8(a1h)}, Alo[2] = {as_f16x8(a0l), as_f16x8(a1l)};
            jsh[lane] = j;
            const int groups = cnt > 32 ? 2 : 1;                             // wave-uniform
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                if (gq >= groups) break;
                f32x16 D0, D1;
#pragma unroll
                for (int r = 0; r < 16; ++r) D0[r] = D1[r] = 0.f;
                D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[gq], wlo[0], D0, 0, 0, 0);
                D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[gq], whi[0], D0, 0, 0, 0);
                D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[gq], whi[0], D0, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[gq], wlo[1], D1, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[gq], whi[1], D1, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[gq], whi[1], D1, 0, 0, 0);
                // register r of lane half h = edge 32 gq + (r & 3) + 8 (r >> 2) + 4 h; edges beyond cnt carry u = 0
                // eight sender rows in flight per lane half (two batches per group: the registers this saves buy a fifth
                // wave per SIMD, which hides more of the gather's latency than sixteen rows per wave did)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    int jj[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) jj[r] = jsh[32 * gq + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * h];
                    float ph0[8], ph1[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float *row = phi_in + (size_t)jj[r] * DF;
                        ph0[r] = row[c];
                        ph1[r] = c < 8 ? row[32 + c] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        v0 = fmaf(ph0[r], D0[r0 + r], v0);
                        v1 = fmaf(ph1[r], D1[r0 + r], v1);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();                                 // jsh is rewritten by the next chunk
        }
        v0 += __shfl_xor(v0, 32, 64);
        v1 += __shfl_xor(v1, 32, 64);
        if (h == 0) {
            float *out = sc.V + (size_t)n * DF;
            out[c] = v0 * 0.00390625f;                                       // 2^-8: the two factors of 16
            if (c < 8) out[32 + c] = v1 * 0.00390625f;
        }
    }
}

// S += dense_blocks[blk](V)  (Sequential(swish, Linear, swish, Linear), vae_model.py:365-369, 489), phi for blk + 1
__global__ __launch_bounds__(256) void dec_dense_kernel(codlad_decoder_weights w, int blk, int M, float *scr) {
    __shared__ float col[3 * DF][64];
    __shared__ __attribute__((aligned(16))) float wbuf[STAGE_FLOATS(DF, DF)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = blockIdx.x * 64 + lane;
    const bool active = n < M;
    const int nn = active ? n : M - 1;
    const Scratch sc = scratch_of(scr, M);
    float *a_col = &col[0][lane], *t_col = &col[DF][lane], *s_col = &col[2 * DF][lane];
    Staged<DF, DF> st;
    st.fetch(w.dense1_w[blk], w.dense1_b[blk]);
    float s_old[DF / 4];                                  // this wave's rows of S, in flight under the two layers
#pragma unroll
    for (int i = 0; i < DF / 4; ++i) s_old[i] = sc.S[(size_t)(wave + 4 * i) * M + nn];
    {   // the workgroup's 64 x 40 sums are one contiguous block of V: coalesced dword reads, transposed into the columns
        const size_t first = (size_t)blockIdx.x * 64 * DF, total = (size_t)M * DF;
        for (int i = threadIdx.x; i < 64 * DF; i += 256) {
            const float v = first + i < total ? sc.V[first + i] : 0.f;
            col[i % DF][i / DF] = swishf(v);
        }
    }
    st.commit(wbuf);
    st.fetch(w.dense3_w[blk], w.dense3_b[blk]);
    __syncthreads();
    wg_dense<DF, DF>(wbuf, a_col, wave, [&](int, int c, float v) { t_col[c * 64] = swishf(v); });
    __syncthreads();
    st.commit(wbuf);
    if (blk < 3) st.fetch(w.inv0_w[blk + 1], w.inv0_b[blk + 1]);
    __syncthreads();
    wg_dense<DF, DF>(wbuf, t_col, wave, [&](int i, int c, float v) {
        const float s = s_old[i] + v;
        s_col[c * 64] = s;
        if (active) sc.S[(size_t)c * M + n] = s;
    });
    if (blk == 3) return;
    __syncthreads();
    write_phi(w, blk + 1, st, wbuf, s_col, a_col, sc.phi(blk + 1), n, active, wave);
}

// One staged layer: barrier (the previous layer's columns are written, nobody reads the staging buffer any more), this
// layer's weights into the buffer, the next layer's on their way into registers, barrier, compute.
template <int IN, int OUT, typename Next, typename Emit>
DEV void staged_layer(const Staged<IN, OUT> &cur, float *wbuf, const float *in, int wave, Next fetch_next, Emit emit) {
    __syncthreads();
    cur.commit(wbuf);
    fetch_next();
    __syncthreads();
    wg_dense<IN, OUT>(wbuf, in, wave, emit);
}

// The output heads (vae_model.py:393-412 / 490-503): backbone angles and torsions, side-chain angles
// (embedding or head), the four residual torsion blocks and the final torsion head; bond lengths from tables.
// Every head is Sequential(swish, Linear, swish, Linear): `xs` holds swish of the running state, refreshed by the
// wave that updates a row; the running state itself (row c belongs to wave c mod 4 throughout) stays in that wave's
// registers.
template <bool ANGLE>
__global__ __launch_bounds__(256) void dec_heads_kernel(codlad_decoder_weights w, const int32_t *cg_z, int M,
                                                       const float *S, float *ic) {
    constexpr int F = ANGLE ? DF + 10 : DF;
    // rows: xs 0..51 | u 52..103 | small 104..135
    __shared__ float col[136][64];
    __shared__ __attribute__((aligned(16))) float wbuf[STAGE_FLOATS(F, F)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = blockIdx.x * 64 + lane;
    const bool active = n < M;
    const int nn = active ? n : M - 1;
    float *xs = &col[0][lane], *u_col = &col[52][lane], *sm = &col[104][lane];
    float *bb_angle = sm + 3 * 64, *bb_tors = sm + 9 * 64, *sc_angle = sm + 12 * 64, *sc_tors = sm + 22 * 64;
    Staged<DF, 3> a1;
    a1.fetch(w.bb_ang1_w, w.bb_ang1_b);
    const int z = cg_z[nn];
    float t[(F + 3) / 4];                                                           // rows wave, wave + 4, ... of the state
#pragma unroll
    for (int i = 0; i < DF / 4; ++i) {
        const int c = wave + 4 * i;
        t[i] = S[(size_t)c * M + nn];
        xs[c * 64] = swishf(t[i]);
    }
    Staged<3, 3> a3;
    staged_layer(a1, wbuf, xs, wave, [&] { a3.fetch(w.bb_ang3_w, w.bb_ang3_b); },
                 [&](int, int c, float v) { sm[c * 64] = swishf(v); });
    Staged<DF + 3, 3> t1;
    staged_layer(a3, wbuf, sm, wave, [&] { t1.fetch(w.bb_tor1_w, w.bb_tor1_b); }, [&](int, int c, float v) {
        bb_angle[c * 64] = v;
        xs[(DF + c) * 64] = swishf(v);                                             // cat([S, bb_angle])
    });
    Staged<3, 3> t3;
    staged_layer(t1, wbuf, xs, wave, [&] { t3.fetch(w.bb_tor3_w, w.bb_tor3_b); },
                 [&](int, int c, float v) { sm[(6 + c) * 64] = swishf(v); });
    Staged<F, F> r1;
    Staged<DF, 10> s1;
    staged_layer(t3, wbuf, sm + 6 * 64, wave, [&] {
        if (ANGLE) s1.fetch(w.sc_ang1_w, w.sc_ang1_b);
        else r1.fetch(w.tor1_w[0], w.tor1_b[0]);
    }, [&](int, int c, float v) { bb_tors[c * 64] = v; });
    if (ANGLE) {
        Staged<10, 10> s3;
        staged_layer(s1, wbuf, xs, wave, [&] { s3.fetch(w.sc_ang3_w, w.sc_ang3_b); },
                     [&](int, int c, float v) { u_col[c * 64] = swishf(v); });
        staged_layer(s3, wbuf, u_col, wave, [&] { r1.fetch(w.tor1_w[0], w.tor1_b[0]); }, [&](int i, int c, float v) {
            sc_angle[c * 64] = v;
            t[(F + 3) / 4 > DF / 4 ? DF / 4 + i : 0] = v;                           // cat([S, sc_angle]): row DF + c
            xs[(DF + c) * 64] = swishf(v);
        });
    } else {
        for (int k = wave; k < 10; k += 4) sc_angle[k * 64] = w.sc_angle_emb[z * 10 + k];
    }
    Staged<F, 10> f1;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        Staged<F, F> r3;
        staged_layer(r1, wbuf, xs, wave, [&] { r3.fetch(w.tor3_w[b], w.tor3_b[b]); },
                     [&](int, int c, float v) { u_col[c * 64] = swishf(v); });
        staged_layer(r3, wbuf, u_col, wave, [&] {
            if (b < 3) r1.fetch(w.tor1_w[b + 1], w.tor1_b[b + 1]);
            else f1.fetch(w.fin1_w, w.fin1_b);
        }, [&](int i, int c, float v) {
            t[i] += v;
            xs[c * 64] = swishf(t[i]);
        });
    }
    Staged<10, 10> f3;
    staged_layer(f1, wbuf, xs, wave, [&] { f3.fetch(w.fin3_w, w.fin3_b); },
                 [&](int, int c, float v) { u_col[c * 64] = swishf(v); });
    staged_layer(f3, wbuf, u_col, wave, [] {}, [&](int, int c, float v) { sc_tors[c * 64] = v; });
    __syncthreads();
    if (!active) return;
    float *o = ic + (size_t)n * 39;
    for (int k = wave; k < 13; k += 4) {
        if (k < 3) {
            o[k * 3 + 0] = w.bb_dist[z * 3 + k];
            o[k * 3 + 1] = bb_angle[k * 64];
            o[k * 3 + 2] = bb_tors[k * 64];
        } else {
            o[k * 3 + 0] = w.sc_dist[z * 10 + (k - 3)];
            o[k * 3 + 1] = sc_angle[(k - 3) * 64];
            o[k * 3 + 2] = sc_tors[(k - 3) * 64];
        }
    }
}

extern "C" int codlad_ic_decode(const codlad_decoder_weights *w, const float *z_q,
                                const int32_t *cg_z, const float *cg_xyz, const int32_t *csr_ptr,
                                const int32_t *csr_src, int M, float *scratch, float *ic_out,
                                void *stream) {
    CODLAD_REQUIRE(w && z_q && cg_z && cg_xyz && csr_ptr && csr_src && scratch && ic_out, "null pointer");
    CODLAD_REQUIRE(M > 0, "M must be positive");
    hipStream_t st = (hipStream_t)stream;
    const dim3 per_lane((M + 63) / 64), per_wave((M + EDGE_WAVES - 1) / EDGE_WAVES), block(256);
    const int edge_grid = M < 16 * num_cu() ? M : 16 * num_cu();    // persistent waves, four per SIMD
    const bool exact = dec_edge_variant() == 1;
    hipLaunchKernelGGL(dec_init_kernel, per_lane, block, 0, st, *w, z_q, cg_z, M, scratch);
    for (int blk = 0; blk < 4; ++blk) {
        if (exact) hipLaunchKernelGGL(dec_edge_exact_kernel, per_wave, dim3(64 * EDGE_WAVES), 0, st, *w, blk, cg_xyz, csr_ptr, csr_src, M, scratch);
        else hipLaunchKernelGGL(dec_edge_kernel, dim3(edge_grid), dim3(64), 0, st, *w, blk, cg_xyz, csr_ptr, csr_src, M, scratch);
        hipLaunchKernelGGL(dec_dense_kernel, per_lane, block, 0, st, *w, blk, M, scratch);
    }
    if (w->angle) hipLaunchKernelGGL(dec_heads_kernel<true>, per_lane, block, 0, st, *w, cg_z, M, scratch, ic_out);
    else hipLaunchKernelGGL(dec_heads_kernel<false>, per_lane, block, 0, st, *w, cg_z, M, scratch, ic_out);
    return codlad_check_launch("codlad_ic_decode");
}
