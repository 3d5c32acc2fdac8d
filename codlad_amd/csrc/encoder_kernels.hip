// the e3nn encoder / CG prior of the VAE (reference models/vae_model.py:21-311, models/gcn_nn.py:163-219),
// i.e. what `--experiment recon` (VAE.get_latent_wovq) and the conditional prior (get_latent_cg) run before the decoder.
//
// One kernel does a whole TensorProductConvLayer (gcn_nn.py:176-219) for one receiving-node set:
//   out[n] (+)= mean over the edges (n <- s) of  tp( h_snd[s], Y(r), fc( [edge_embed(r, types) | h[:12] | h[:12]] ) )
// with tp = e3nn's o3.FullyConnectedTensorProduct(in, 1x0e+1x1o+1x2e, out, shared_weights=False) restated from its
// published definition (oracle/e3nn_lite.py says exactly what was restated and how far it is pinned: the non-trivial
// Wigner symbols equal the buffers e3nn itself left in the reference's shipped checkpoint).
//
// Mapping: GROUP lanes (1, 16 or 64) share a receiving node and each takes one of its edges; everything per edge - the
// distance, Gaussian smearing, the edge-embedding MLP, the real spherical harmonics up to l = 2, the hidden layer of fc
// and the pre-contracted inputs of every path - lives in the lane's registers.  The second layer of fc (36 -> 192 / 288 /
// 384 per-edge weights, the bulk of the arithmetic) is never materialised: the row of fc.3 that yields weight (path, u, w)
// is wave-uniform and comes through the scalar cache, the weight is formed as a 36-term dot product with SGPR operands and
// consumed at once.  Outputs are produced one (irrep, w) at a time, summed over the group's lanes by shuffles and
// accumulated in an LDS slot per node, so a node of any degree needs no atomics and the sum has a fixed order.
#include "encoder_common.h"

int tp_conv_variant();                                                       // denoiser_kernels.hip: CODLAD_OPT_TP_CONV_VARIANT
void launch_tp_conv_mfma(const codlad_tp_conv_args &a, hipStream_t st);      // encoder_mfma_kernel.hip
void launch_tp_conv_pack(const codlad_tp_conv_args &a, void *image, hipStream_t st);
int tp_conv_image_bytes(int depth);

namespace {

template <int GROUP>
DEV float group_sum(float v) {
#pragma unroll
    for (int m = GROUP / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// one per-edge weight: fc.3 row `idx` (wave-uniform) . hid + bias
DEV float edge_weight(kfloat_p fc3_w, kfloat_p fc3_b, int idx, const float (&hid)[36]) {
    kfloat_p row = fc3_w + idx * 36;
    float acc = fc3_b[idx];
#pragma unroll
    for (int m = 0; m < 36; ++m) acc = fmaf(hid[m], row[m], acc);
    return acc;
}

template <int DEPTH, int GROUP>
__global__ __launch_bounds__(64) void tp_conv_kernel(codlad_tp_conv_args a) {
    constexpr int NODES = 64 / GROUP;                    // receiving nodes per wave
    constexpr int D_IN = width_of(DEPTH), D_OUT = width_of(DEPTH + 1);
    __shared__ float acc_sh[NODES][48];
    const int lane = threadIdx.x, g = lane / GROUP, gl = lane % GROUP;
    const int n = blockIdx.x * NODES + g;
    const bool node_ok = n < a.n_recv;
    const int nn = node_ok ? n : a.n_recv - 1;
    const int e0 = a.ptr[nn], e1 = node_ok ? a.ptr[nn + 1] : e0;
    for (int i = lane; i < NODES * 48; i += 64) (&acc_sh[0][0])[i] = 0.f;
    // longest edge list of the wave's nodes (wave-uniform trip count)
    int deg = e1 - e0;
    int max_deg = deg;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int o = __shfl_xor(max_deg, m, 64);
        max_deg = o > max_deg ? o : max_deg;
    }
    max_deg = __builtin_amdgcn_readfirstlane(max_deg);
    const float xr = a.xyz_recv[3 * nn], yr = a.xyz_recv[3 * nn + 1], zr = a.xyz_recv[3 * nn + 2];
    float hr[NS];                                        // the receiving node's scalars
#pragma unroll
    for (int k = 0; k < NS; ++k) hr[k] = a.h_recv[(size_t)nn * a.d_recv + k];
    const float typ_r = a.typ_recv ? a.typ_recv[nn] : 0.f;
    kfloat_p fc0_w = uni(a.fc0_w), fc0_b = uni(a.fc0_b), fc3_w = uni(a.fc3_w), fc3_b = uni(a.fc3_b);
    kfloat_p emb0_w = uni(a.emb0_w), emb0_b = uni(a.emb0_b), emb3_w = uni(a.emb3_w), emb3_b = uni(a.emb3_b);
    const float step = a.smear_stop / 7.0f, coeff = -0.5f / (step * step);

    for (int c0 = 0; c0 < max_deg; c0 += GROUP) {
        const bool live = c0 + gl < deg;
        const int s = live ? a.snd[e0 + c0 + gl] : 0;
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
        // fc input: [edge embedding (12) | scalars | scalars]
        float in36[36];
        {
            float sm[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t = d - step * (float)k;
                sm[k] = expf(coeff * t * t);
            }
            float h1[NS];
            if (a.emb_in == 14) {                                       // [z_recv, z_snd, 0 0 0 0, smearing]
                const float typ_s = a.typ_snd[s];
#pragma unroll
                for (int o = 0; o < NS; ++o) {
                    float acc = emb0_b[o];
                    acc = fmaf(typ_r, emb0_w[o * 14], acc);
                    acc = fmaf(typ_s, emb0_w[o * 14 + 1], acc);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = fmaf(sm[k], emb0_w[o * 14 + 6 + k], acc);
                    h1[o] = fmaxf(acc, 0.f);
                }
            } else {
#pragma unroll
                for (int o = 0; o < NS; ++o) {
                    float acc = emb0_b[o];
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = fmaf(sm[k], emb0_w[o * 8 + k], acc);
                    h1[o] = fmaxf(acc, 0.f);
                }
            }
#pragma unroll
            for (int o = 0; o < NS; ++o) {
                float acc = emb3_b[o];
#pragma unroll
                for (int k = 0; k < NS; ++k) acc = fmaf(h1[k], emb3_w[o * NS + k], acc);
                in36[o] = acc;
            }
        }
        // sender features
        const float *hs = a.h_snd + (size_t)s * a.d_snd;
        float x0[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) x0[k] = hs[k];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            in36[12 + k] = a.attr_recv_first ? hr[k] : x0[k];
            in36[24 + k] = a.attr_recv_first ? x0[k] : hr[k];
        }
        float hid[36];
#pragma unroll
        for (int o = 0; o < 36; ++o) {
            float acc = fc0_b[o];
#pragma unroll
            for (int k = 0; k < 36; ++k) acc = fmaf(in36[k], fc0_w[o * 36 + k], acc);
            hid[o] = fmaxf(acc, 0.f);
        }
        // pre-contracted inputs of the paths that start from a vector block
        Vec3 v1[NV], v2[NV];
        float d4[NV], d9[NV];
        Vec3 c5[NV], q6[NV], c8[NV], q10[NV];
        if (DEPTH >= 1) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                v1[u] = {hs[12 + 3 * u], hs[13 + 3 * u], hs[14 + 3 * u]};
                d4[u] = dot3(v1[u], y1) * INV_SQRT3;
                c5[u] = cross3(v1[u], y1);
                q6[u] = w121(v1[u], y2);
            }
        }
        if (DEPTH >= 2) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                v2[u] = {hs[24 + 3 * u], hs[25 + 3 * u], hs[26 + 3 * u]};
                d9[u] = dot3(v2[u], y1) * INV_SQRT3;
                c8[u] = cross3(v2[u], y1);
                q10[u] = w121(v2[u], y2);
            }
        }
        // instruction order of FullyConnectedTensorProduct (i_in1, i_in2, i_out) and the offsets of their weight blocks
        constexpr int O1 = 0, O2 = 144, O3 = 192, O4 = 208, O5 = 256, O6 = 272, O7 = 288, O8 = 304, O9 = 320, O10 = 368;
        // path coefficients sqrt((2 l_out + 1) / sum of mul_in1 over the paths into the same output block)
        constexpr float C0E = DEPTH == 0 ? 0.28867513459481287f : 0.25f;                                  // 1/sqrt 12, 1/4
        constexpr float C1O = DEPTH == 0 ? 0.5f : (DEPTH == 1 ? 0.38729833462074170f : 0.35355339059327379f);   // sqrt(3/12, 3/20, 3/24)
        constexpr float C1E = DEPTH == 1 ? 0.86602540378443865f : 0.5f;                                   // sqrt(3/4), sqrt(3/12)
        constexpr float C0O = 0.5f;                                                                        // sqrt(1/4)
        float *acc = acc_sh[g];
        // ---- 12x0e
        for (int w = 0; w < NS; ++w) {
            float r = 0.f;
#pragma unroll
            for (int u = 0; u < NS; ++u) r = fmaf(edge_weight(fc3_w, fc3_b, O1 + u * NS + w, hid), x0[u], r);
            if (DEPTH >= 1) {
#pragma unroll
                for (int u = 0; u < NV; ++u) r = fmaf(edge_weight(fc3_w, fc3_b, O4 + u * NS + w, hid), d4[u], r);
            }
            r = group_sum<GROUP>(live ? C0E * r : 0.f);
            if (gl == 0) acc[w] += r;
        }
        // ---- 4x1o
        for (int w = 0; w < NV; ++w) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < NS; ++u) t = fmaf(edge_weight(fc3_w, fc3_b, O2 + u * NV + w, hid), x0[u], t);
            t *= INV_SQRT3;
            Vec3 r = {t * y1.x, t * y1.y, t * y1.z};
            if (DEPTH >= 1) {
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const float w3 = edge_weight(fc3_w, fc3_b, O3 + u * NV + w, hid) * INV_SQRT3;
                    const float w6 = edge_weight(fc3_w, fc3_b, O6 + u * NV + w, hid);
                    r.x = fmaf(w3, v1[u].x, fmaf(w6, q6[u].x, r.x));
                    r.y = fmaf(w3, v1[u].y, fmaf(w6, q6[u].y, r.y));
                    r.z = fmaf(w3, v1[u].z, fmaf(w6, q6[u].z, r.z));
                }
            }
            if (DEPTH >= 2) {
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const float w8 = edge_weight(fc3_w, fc3_b, O8 + u * NV + w, hid) * INV_SQRT6;
                    r.x = fmaf(w8, c8[u].x, r.x);
                    r.y = fmaf(w8, c8[u].y, r.y);
                    r.z = fmaf(w8, c8[u].z, r.z);
                }
            }
            const float sx = group_sum<GROUP>(live ? C1O * r.x : 0.f), sy = group_sum<GROUP>(live ? C1O * r.y : 0.f),
                        sz = group_sum<GROUP>(live ? C1O * r.z : 0.f);
            if (gl == 0) {
                acc[12 + 3 * w] += sx; acc[13 + 3 * w] += sy; acc[14 + 3 * w] += sz;
            }
        }
        // ---- 4x1e
        if (DEPTH >= 1) {
            for (int w = 0; w < NV; ++w) {
                Vec3 r = {0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const float w5 = edge_weight(fc3_w, fc3_b, O5 + u * NV + w, hid) * INV_SQRT6;
                    r.x = fmaf(w5, c5[u].x, r.x);
                    r.y = fmaf(w5, c5[u].y, r.y);
                    r.z = fmaf(w5, c5[u].z, r.z);
                }
                if (DEPTH >= 2) {
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        const float w7 = edge_weight(fc3_w, fc3_b, O7 + u * NV + w, hid) * INV_SQRT3;
                        const float w10 = edge_weight(fc3_w, fc3_b, O10 + u * NV + w, hid);
                        r.x = fmaf(w7, v2[u].x, fmaf(w10, q10[u].x, r.x));
                        r.y = fmaf(w7, v2[u].y, fmaf(w10, q10[u].y, r.y));
                        r.z = fmaf(w7, v2[u].z, fmaf(w10, q10[u].z, r.z));
                    }
                }
                const float sx = group_sum<GROUP>(live ? C1E * r.x : 0.f), sy = group_sum<GROUP>(live ? C1E * r.y : 0.f),
                            sz = group_sum<GROUP>(live ? C1E * r.z : 0.f);
                if (gl == 0) {
                    acc[24 + 3 * w] += sx; acc[25 + 3 * w] += sy; acc[26 + 3 * w] += sz;
                }
            }
        }
        // ---- 12x0o
        if (DEPTH >= 2) {
            for (int w = 0; w < NS; ++w) {
                float r = 0.f;
#pragma unroll
                for (int u = 0; u < NV; ++u) r = fmaf(edge_weight(fc3_w, fc3_b, O9 + u * NS + w, hid), d9[u], r);
                r = group_sum<GROUP>(live ? C0O * r : 0.f);
                if (gl == 0) acc[36 + w] += r;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // out[n] = (accumulate ? out[n] : pad(h_recv[n])) + sum / degree
    if (node_ok) {
        const float scale = deg > 0 ? 1.0f / (float)deg : 0.f;
        float *o = a.out + (size_t)n * D_OUT;
        for (int k = gl; k < D_OUT; k += GROUP) {
            const float base = a.accumulate ? o[k] : (k < a.d_recv ? a.h_recv[(size_t)n * a.d_recv + k] : 0.f);
            o[k] = base + acc_sh[g][k] * scale;
        }
    }
    (void)D_IN;
}

// y = W2 act(W1 x + b1) + b2 per row: 64 rows per workgroup (lane = row), the outputs of a layer dealt round-robin to its
// 4 waves (a single wave walking all 72 output rows one scalar-cache round trip after the other took 70-100 us for a few
// hundred rows); weights through the scalar cache, the hidden vector through LDS columns.  act: 0 tanh, 1 relu.  Used for
// the encoder's dense head, map_in and the prior's mu / sigma heads.
template <int IN>
__global__ __launch_bounds__(256) void mlp_rows_kernel(const float *x, int n, const float *w1, const float *b1, int hidden,
                                                      const float *w2, const float *b2, int out_dim, int act, int mode,
                                                      float *y) {
    __shared__ float hcol[36][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * 64 + lane;
    const int ii = i < n ? i : n - 1;
    float in[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) in[k] = x[(size_t)ii * IN + k];
    kfloat_p W1 = uni(w1), B1 = uni(b1), W2 = uni(w2), B2 = uni(b2);
    if (hidden > 0) {
        for (int o = wave; o < hidden; o += 4) {
            float acc = B1[o];
#pragma unroll
            for (int k = 0; k < IN; ++k) acc = fmaf(in[k], W1[o * IN + k], acc);
            hcol[o][lane] = act == 0 ? tanhf(acc) : fmaxf(acc, 0.f);
        }
        __syncthreads();
    }
    for (int o = wave; o < out_dim; o += 4) {
        float acc = B2[o];
        if (hidden > 0) {
            for (int k = 0; k < hidden; ++k) acc = fmaf(hcol[k][lane], W2[o * hidden + k], acc);
        } else {
#pragma unroll
            for (int k = 0; k < IN; ++k) acc = fmaf(in[k], W2[o * IN + k], acc);
        }
        if (mode == 1) acc = 1e-9f + expf(acc / 2.0f);             // prior: H_sigma = 1e-9 + exp(logvar / 2)
        if (i < n) y[(size_t)i * out_dim + o] = acc;
    }
}

// node[I] = [ mean over the atoms of bead I of h_atom (48) | h_cg[I] (36) ]  (vae_model.py:159-160: scatter_mean of
// cat([atom, cg[mapping]]) over mapping; the CG half is constant within a bead).  One thread per (bead, column).
__global__ void bead_mean_kernel(const float *h_atom, const float *h_cg, const int *bead_ptr, const int *bead_atoms,
                                 int n_cg, float *node) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cg * 84) return;
    const int I = i / 84, k = i % 84;
    const int e0 = bead_ptr[I], e1 = bead_ptr[I + 1];
    float v = 0.f;
    if (k < 48) {
        for (int e = e0; e < e1; ++e) v += h_atom[(size_t)bead_atoms[e] * 48 + k];
        v = e1 > e0 ? v / (float)(e1 - e0) : 0.f;
    } else {
        v = e1 > e0 ? h_cg[(size_t)I * 36 + (k - 48)] : 0.f;
    }
    node[i] = v;
}

__global__ void embed_rows_kernel(const float *table, const int *idx, int n, int width, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * width) return;
    out[i] = table[(size_t)idx[i / width] * width + i % width];
}

// ---------------------------------------------------------------------------------------------------------------------
// Receivers' CSR of a pair list (the reference's make_directed + the grouping its scatter does, models/gcn_nn.py:54-64):
// histogram -> scan -> fill -> per-receiver sort by sender, so that the edge order inside a receiver (and with it the
// rounding of its mean) is a function of the graph alone, not of the atomics' arrival order.
//   work: degA [n] | degB [n] | flags [2] | tmp [2 E] | slot [2 E]
// (the counting pass keeps what its atomics return - an edge's slot inside its receiver's group - so the fill pass needs
// no atomics of its own)
__global__ __launch_bounds__(256) void csr_hist_kernel(const int64_t *pairs, int E, int n, int *work) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool in = e < E;
    const int a = in ? (int)pairs[2 * (size_t)e] : -1, b = in ? (int)pairs[2 * (size_t)e + 1] : -1;
    int *slot = work + 2 * n + 2 + 2 * (size_t)E;
    // first index: a pair list in torch.nonzero order has long runs of equal receivers - one atomic per run of a wave
    // instead of one per pair on the same counter (the run's first lane adds the run's length and hands out the slots)
    const int prev = __shfl_up(a, 1, 64);
    const bool head = lane == 0 || prev != a;
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);          // run heads at or below this lane
    const int my_head = 63 - __builtin_clzll(below);
    const unsigned long long above = heads & ~((2ull << my_head) - 1ull);      // the next run's head, if any
    const int run_end = above ? __builtin_ctzll(above) : 64;
    int base = 0;
    if (head && in) base = atomicAdd(work + a, run_end - lane);
    base = __shfl(base, my_head, 64);
    if (in) {
        slot[e] = base + (lane - my_head);
        slot[E + e] = atomicAdd(work + n + b, 1);
        if (a > b) work[2 * n] = 1;            // benign race: every writer stores 1
        if (b > a) work[2 * n + 1] = 1;
    }
}

// one workgroup: ptr = exclusive scan of degA (+ degB when the list holds one direction only); degA keeps every receiver's
// number of edges of the first kind (the second kind's slots start behind them)
__global__ __launch_bounds__(1024) void csr_scan_kernel(int n, int mode, int *work, int *ptr) {
    __shared__ int part[1024];
    const int both = mode == 0 && !(work[2 * n] && work[2 * n + 1]);
    const int per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += work[i] + (both ? work[n + i] : 0);
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                       // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        ptr[i] = run;
        run += work[i] + (both ? work[n + i] : 0);
    }
    if (threadIdx.x == 1023) ptr[n] = part[1023];
    __syncthreads();                                            // every thread has read the flags
    if (threadIdx.x == 0) work[2 * n] = both;                   // the fill reads the decision from here
}

__global__ void csr_fill_kernel(const int64_t *pairs, int E, int n, const int *ptr, int *work) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int a = (int)pairs[2 * (size_t)e], b = (int)pairs[2 * (size_t)e + 1];
    int *tmp = work + 2 * n + 2;
    const int *slot = tmp + 2 * (size_t)E;
    tmp[ptr[a] + slot[e]] = b;
    if (work[2 * n]) tmp[ptr[b] + work[b] + slot[E + e]] = a;
}

// one wave per receiver: senders ascending (rank sort; equal senders keep their slots' order).  Up to 256 senders sit in four
// registers per lane and are compared through v_readlane (a loop of uniform-address loads was one L1 round trip per sender);
// longer lists fall back to that loop.
__global__ __launch_bounds__(256) void csr_sort_kernel(int n, const int *ptr, const int *work, int *snd) {
    const int node = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (node >= n) return;
    const int *tmp = work + 2 * n + 2;
    const int e0 = __builtin_amdgcn_readfirstlane(ptr[node]), deg = __builtin_amdgcn_readfirstlane(ptr[node + 1]) - e0;
    if (deg <= 256) {
        int k[4], r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            k[q] = 64 * q + lane < deg ? tmp[e0 + 64 * q + lane] : 0x7fffffff;
            r[q] = 0;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {                                       // senders 64 p .. 64 p + 63, one by one
            const int lim = deg - 64 * p < 64 ? deg - 64 * p : 64;
            for (int j = 0; j < lim; ++j) {                                  // j is wave-uniform
                const int kj = __builtin_amdgcn_readlane(k[p], j);
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] += (kj < k[q]) || (kj == k[q] && 64 * p + j < 64 * q + lane);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (64 * q + lane < deg) snd[e0 + r[q]] = k[q];
        return;
    }
    for (int i = lane; i < deg; i += 64) {
        const int key = tmp[e0 + i];
        int rank = 0;
        for (int j = 0; j < deg; ++j) {
            const int k = tmp[e0 + j];
            rank += (k < key) || (k == key && j < i);
        }
        snd[e0 + rank] = key;
    }
}

template <int DEPTH>
void launch_tp(const codlad_tp_conv_args &a, hipStream_t st) {
    if (tp_conv_variant() == 0 || (tp_conv_variant() == 2 && a.group == 64)) return launch_tp_conv_mfma(a, st);
    if (a.group == 1) hipLaunchKernelGGL((tp_conv_kernel<DEPTH, 1>), dim3((a.n_recv + 63) / 64), dim3(64), 0, st, a);
    else if (a.group == 16) hipLaunchKernelGGL((tp_conv_kernel<DEPTH, 16>), dim3((a.n_recv + 3) / 4), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((tp_conv_kernel<DEPTH, 64>), dim3(a.n_recv), dim3(64), 0, st, a);
}

}  // namespace

extern "C" int codlad_tp_conv(const codlad_tp_conv_args *a, void *stream) {
    CODLAD_REQUIRE(a && a->ptr && a->snd && a->xyz_recv && a->xyz_snd && a->h_recv && a->h_snd && a->out, "null pointer");
    CODLAD_REQUIRE(a->fc0_w && a->fc0_b && a->fc3_w && a->fc3_b && a->emb0_w && a->emb0_b && a->emb3_w && a->emb3_b,
                   "null weight pointer");
    CODLAD_REQUIRE(a->n_recv > 0, "n_recv must be positive");
    CODLAD_REQUIRE(a->depth >= 0 && a->depth <= 2, "depth must be 0, 1 or 2");
    CODLAD_REQUIRE(a->group == 1 || a->group == 16 || a->group == 64, "group must be 1, 16 or 64");
    CODLAD_REQUIRE(a->emb_in == 14 || a->emb_in == 8, "edge embedding input must be 14 (types + smearing) or 8 (smearing)");
    CODLAD_REQUIRE(a->emb_in == 8 || (a->typ_recv && a->typ_snd), "node types are needed for a 14-wide edge embedding");
    CODLAD_REQUIRE(a->d_snd == width_of(a->depth) && a->d_recv >= 12 && a->d_recv <= width_of(a->depth + 1),
                   "feature widths do not match the depth");
    hipStream_t st = (hipStream_t)stream;
    if (a->depth == 0) launch_tp<0>(*a, st);
    else if (a->depth == 1) launch_tp<1>(*a, st);
    else launch_tp<2>(*a, st);
    return codlad_check_launch("codlad_tp_conv");
}

extern "C" int codlad_tp_conv_args_size(void) { return (int)sizeof(codlad_tp_conv_args); }

extern "C" int codlad_tp_conv_image_bytes(int depth) { return depth >= 0 && depth <= 2 ? tp_conv_image_bytes(depth) : -1; }

extern "C" int codlad_tp_conv_pack(const codlad_tp_conv_args *a, void *image, void *stream) {
    CODLAD_REQUIRE(a && image, "null pointer");
    CODLAD_REQUIRE(a->fc0_w && a->fc0_b && a->fc3_w && a->fc3_b && a->emb0_w && a->emb0_b && a->emb3_w && a->emb3_b,
                   "null weight pointer");
    CODLAD_REQUIRE(a->depth >= 0 && a->depth <= 2 && (a->emb_in == 14 || a->emb_in == 8), "bad depth or emb_in");
    launch_tp_conv_pack(*a, image, (hipStream_t)stream);
    return codlad_check_launch("codlad_tp_conv_pack");
}

extern "C" int codlad_receiver_csr(const int64_t *pairs, int n_pairs, int n_nodes, int mode, int32_t *ptr, int32_t *snd,
                                   int32_t *work, void *stream) {
    CODLAD_REQUIRE(pairs && ptr && snd && work, "null pointer");
    CODLAD_REQUIRE(n_pairs > 0 && n_nodes > 0 && (mode == 0 || mode == 1), "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const hipError_t e = hipMemsetAsync(work, 0, sizeof(int32_t) * (2 * (size_t)n_nodes + 2), st);
    if (e != hipSuccess) { codlad_set_error("codlad_receiver_csr: %s", hipGetErrorString(e)); return (int)e; }
    const dim3 per_pair((n_pairs + 255) / 256), block(256);
    hipLaunchKernelGGL(csr_hist_kernel, per_pair, block, 0, st, pairs, n_pairs, n_nodes, work);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, st, n_nodes, mode, work, ptr);
    hipLaunchKernelGGL(csr_fill_kernel, per_pair, block, 0, st, pairs, n_pairs, n_nodes, ptr, work);
    hipLaunchKernelGGL(csr_sort_kernel, dim3((n_nodes + 3) / 4), block, 0, st, n_nodes, ptr, work, snd);
    return codlad_check_launch("codlad_receiver_csr");
}

extern "C" int codlad_mlp_rows(const float *x, int n, int in_dim, const float *w1, const float *b1, int hidden,
                               const float *w2, const float *b2, int out_dim, int act, int mode, float *y, void *stream) {
    CODLAD_REQUIRE(x && w2 && b2 && y && n > 0, "bad arguments");
    CODLAD_REQUIRE(hidden == 0 || (w1 && b1 && hidden <= 36), "hidden layer of at most 36 units");
    CODLAD_REQUIRE(out_dim > 0 && out_dim <= 36 && (act == 0 || act == 1) && (mode == 0 || mode == 1), "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((n + 63) / 64), block(256);
    if (in_dim == 84) hipLaunchKernelGGL(mlp_rows_kernel<84>, grid, block, 0, st, x, n, w1, b1, hidden, w2, b2, out_dim, act, mode, y);
    else if (in_dim == 48) hipLaunchKernelGGL(mlp_rows_kernel<48>, grid, block, 0, st, x, n, w1, b1, hidden, w2, b2, out_dim, act, mode, y);
    else if (in_dim == 36) hipLaunchKernelGGL(mlp_rows_kernel<36>, grid, block, 0, st, x, n, w1, b1, hidden, w2, b2, out_dim, act, mode, y);
    else { codlad_set_error("codlad_mlp_rows: in_dim must be 84, 48 or 36"); return -1; }
    return codlad_check_launch("codlad_mlp_rows");
}

extern "C" int codlad_bead_mean(const float *h_atom, const float *h_cg, const int32_t *bead_ptr, const int32_t *bead_atoms,
                                int n_cg, float *node, void *stream) {
    CODLAD_REQUIRE(h_atom && h_cg && bead_ptr && bead_atoms && node && n_cg > 0, "bad arguments");
    hipLaunchKernelGGL(bead_mean_kernel, dim3((n_cg * 84 + 255) / 256), dim3(256), 0, (hipStream_t)stream, h_atom, h_cg,
                       bead_ptr, bead_atoms, n_cg, node);
    return codlad_check_launch("codlad_bead_mean");
}

extern "C" int codlad_embed_rows(const float *table, const int32_t *idx, int n, int width, float *out, void *stream) {
    CODLAD_REQUIRE(table && idx && out && n > 0 && width > 0, "bad arguments");
    hipLaunchKernelGGL(embed_rows_kernel, dim3((n * width + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, idx, n,
                       width, out);
    return codlad_check_launch("codlad_embed_rows");
}
