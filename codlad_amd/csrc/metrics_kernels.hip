// Evaluation metrics that directly follow the sampling path in the reference's loop
// (reference test.py:97-166, called at test.py:589-593): reconstruction losses on internal
// coordinates, coordinate MSE, bond-graph distance error, steric-clash ratios and interaction
// losses.  All of them are sums over short index lists, so they are computed in ONE pass: the
// lists are laid end to end as one item space, every thread strides over it, ten partial sums are
// kept in double precision and reduced per block; a one-block finish kernel adds the block
// partials in a fixed order (deterministic) and applies the reference's normalisations.
// Compiled with -ffp-contract=off: each distance is rounded like the reference's unfused fp32 ops,
// which matters for the 1.2 A clash threshold.
#include "common.h"
#include "../../include/codlad_hip.h"

namespace {
constexpr int N_ACC = 10;
constexpr int MAX_BLOCKS = 256;
constexpr float EPS = 1e-7f;   // reference test.py:27

__device__ inline float dist(const float *xyz, int64_t i, int64_t j) {
    const float dx = xyz[3 * i] - xyz[3 * j], dy = xyz[3 * i + 1] - xyz[3 * j + 1], dz = xyz[3 * i + 2] - xyz[3 * j + 2];
    return sqrtf(((dx * dx + dy * dy) + dz * dz) + EPS);
}

__global__ __launch_bounds__(256) void metrics_partial_kernel(codlad_metric_inputs in, double *partials) {
    double acc[N_ACC];
#pragma unroll
    for (int k = 0; k < N_ACC; ++k) acc[k] = 0.0;
    const int64_t o1 = in.n_atoms, o2 = o1 + in.n_edges, o3 = o2 + in.n_clash, o4 = o3 + in.n_bb,
                  o5 = o4 + in.n_inter, o6 = o5 + in.n_pipi, o7 = o6 + in.n_ic;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < o7; t += (int64_t)gridDim.x * blockDim.x) {
        if (t < o1) {                                   // xyz_result, test.py:148-151
            const float dx = in.xyz_recon[3 * t] - in.xyz[3 * t], dy = in.xyz_recon[3 * t + 1] - in.xyz[3 * t + 1],
                        dz = in.xyz_recon[3 * t + 2] - in.xyz[3 * t + 2];
            acc[0] += (dx * dx + dy * dy) + dz * dz;
        } else if (t < o2) {                            // ged_result, test.py:141-146
            const int64_t *e = in.edge_list + 2 * (t - o1);
            const float d = dist(in.xyz_recon, e[0], e[1]) - dist(in.xyz, e[0], e[1]);
            acc[1] += d * d;
        } else if (t < o3) {                            // clash_result, test.py:124-131
            const int64_t *e = in.clash_list + 2 * (t - o2);
            acc[2] += dist(in.xyz_recon, e[0], e[1]) < 1.2f ? 1.0 : 0.0;
        } else if (t < o4) {                            // clash_result, test.py:133-138
            const int64_t *e = in.bb_NO_list + 2 * (t - o3);
            acc[3] += dist(in.xyz_recon, e[0], e[1]) < 1.2f ? 1.0 : 0.0;
        } else if (t < o5) {                            // inter_result, test.py:103-106
            const int64_t *e = in.interaction_list + 2 * (t - o4);
            acc[4] += fmaxf(dist(in.xyz_recon, e[0], e[1]) - 4.0f, 0.0f);
        } else if (t < o6) {                            // inter_result, test.py:109-113
            const int64_t *q = in.pi_pi_list + 4 * (t - o5);
            float c0[3], c1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                c0[k] = (in.xyz_recon[3 * q[0] + k] + in.xyz_recon[3 * q[1] + k]) / 2.0f;
                c1[k] = (in.xyz_recon[3 * q[2] + k] + in.xyz_recon[3 * q[3] + k]) / 2.0f;
            }
            const float dx = c0[0] - c1[0], dy = c0[1] - c1[1], dz = c0[2] - c1[2];
            acc[5] += fmaxf(sqrtf(((dx * dx + dy * dy) + dz * dz) + EPS) - 6.0f, 0.0f);
        } else {                                        // recon_result, test.py:153-166
            const int64_t s = t - o6;
            const float m = in.ic_mask[s];
            const float *a = in.ic + 3 * s, *b = in.ic_recon + 3 * s;
            const float lb = (b[0] - a[0]) * m;
            acc[6] += lb * lb;
            acc[7] += sqrtf(2.0f * (1.0f - cosf(a[1] - b[1])) + EPS) * m;
            acc[8] += sqrtf(2.0f * (1.0f - cosf(a[2] - b[2])) + EPS) * m;
            acc[9] += m;
        }
    }
    __shared__ double red[256];
    for (int k = 0; k < N_ACC; ++k) {
        red[threadIdx.x] = acc[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partials[blockIdx.x * N_ACC + k] = red[0];
        __syncthreads();
    }
}

__global__ void metrics_finish_kernel(codlad_metric_inputs in, const double *partials, int n_blocks, float *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[N_ACC];
    for (int k = 0; k < N_ACC; ++k) {
        s[k] = 0.0;
        for (int b = 0; b < n_blocks; ++b) s[k] += partials[b * N_ACC + k];
    }
    const double natom = s[9];
    out[0] = (float)(s[6] / natom);                                         // loss_bond
    out[1] = (float)(s[7] / natom);                                         // loss_angle
    out[2] = (float)(s[8] / natom);                                         // loss_torsion
    out[3] = in.n_atoms ? (float)(s[0] / (double)in.n_atoms) : 0.0f;        // loss_xyz
    out[4] = in.n_edges ? (float)(s[1] / (double)in.n_edges) : 0.0f;        // loss_graph
    const float nbr = in.n_clash ? (float)(s[2] / (double)in.n_clash) : 0.0f;
    const float bb = in.n_bb ? (float)(s[3] / (double)in.n_bb) : 0.0f;
    out[5] = nbr + bb;                                                      // loss_nbr
    const double n_total = (double)in.n_inter + (double)in.n_pipi;
    float inter = 0.0f, pipi = 0.0f;
    if (in.n_inter) inter = (float)(s[4] / (double)in.n_inter) * (float)((double)in.n_inter / n_total);
    if (in.n_pipi) {
        pipi = (float)(s[5] / (double)in.n_pipi);
        inter += pipi * (float)((double)in.n_pipi / n_total);
    }
    out[6] = inter;                                                         // loss_inter
    out[7] = pipi;                                                          // loss_pi_pi
}
// Bond-graph validity (reference test.py:168-188 -> utils/protein_module.py:251-318): atoms i != j of one structure
// are bonded when |x_i - x_j| < (r_i + r_j) * scale, r = the covalent cut-off radius of the element.  Per structure
// the kernel counts the bonded unordered pairs of the reference coordinates, of the reconstructed ones and the
// pairs on which the two graphs differ - over all atoms and over the heavy atoms (Z != 1) only.  Arithmetic as the
// reference's unfused fp32 tensor ops (sum of squares over x, y, z in that order, sqrt, (r_i + r_j) * scale).
__global__ __launch_bounds__(256) void bond_graph_kernel(const float *xyz, const float *xyz_recon, const float *radius,
                                                         const int32_t *heavy, const int32_t *struct_ptr, float scale,
                                                         int32_t *counts) {
    const int s = blockIdx.x;
    const int a0 = struct_ptr[s], n = struct_ptr[s + 1] - a0;
    int c[6] = {0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.y; i < n; i += gridDim.y) {
        const float xi = xyz[3 * (a0 + i)], yi = xyz[3 * (a0 + i) + 1], zi = xyz[3 * (a0 + i) + 2];
        const float ui = xyz_recon[3 * (a0 + i)], vi = xyz_recon[3 * (a0 + i) + 1], wi = xyz_recon[3 * (a0 + i) + 2];
        const float ri = radius[a0 + i];
        const int hi = heavy[a0 + i];
        for (int j = i + 1 + threadIdx.x; j < n; j += blockDim.x) {
            const float dx = xi - xyz[3 * (a0 + j)], dy = yi - xyz[3 * (a0 + j) + 1], dz = zi - xyz[3 * (a0 + j) + 2];
            const float du = ui - xyz_recon[3 * (a0 + j)], dv = vi - xyz_recon[3 * (a0 + j) + 1],
                        dw = wi - xyz_recon[3 * (a0 + j) + 2];
            const float cut = (ri + radius[a0 + j]) * scale;
            const bool ref = sqrtf((dx * dx + dy * dy) + dz * dz) < cut;
            const bool gen = sqrtf((du * du + dv * dv) + dw * dw) < cut;
            const int hv = hi & heavy[a0 + j];
            c[0] += ref; c[1] += gen; c[2] += ref != gen;
            c[3] += ref & hv; c[4] += gen & hv; c[5] += (ref != gen) & hv;
        }
    }
    __shared__ int red[256];
    for (int k = 0; k < 6; ++k) {
        red[threadIdx.x] = c[k];
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0 && red[0]) atomicAdd(counts + 6 * s + k, red[0]);   // integers: order-independent
        __syncthreads();
    }
}
}  // namespace

extern "C" int codlad_bond_graph_counts(const float *xyz, const float *xyz_recon, const float *radius,
                                        const int32_t *heavy, const int32_t *struct_ptr, int n_struct, int max_atoms,
                                        float scale, int32_t *counts, void *stream) {
    CODLAD_REQUIRE(xyz && xyz_recon && radius && heavy && struct_ptr && counts, "null pointer");
    CODLAD_REQUIRE(n_struct > 0 && max_atoms > 0 && scale > 0.f, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * 6 * (size_t)n_struct, st);
    if (e != hipSuccess) {
        codlad_set_error("codlad_bond_graph_counts: %s", hipGetErrorString(e));
        return (int)e;
    }
    int by = (max_atoms + 7) / 8;
    by = by < 1 ? 1 : (by > 512 ? 512 : by);
    hipLaunchKernelGGL(bond_graph_kernel, dim3(n_struct, by), dim3(256), 0, st, xyz, xyz_recon, radius, heavy, struct_ptr,
                       scale, counts);
    return codlad_check_launch("codlad_bond_graph_counts");
}

extern "C" int codlad_metrics_scratch_bytes(void) { return MAX_BLOCKS * N_ACC * (int)sizeof(double); }

extern "C" int codlad_eval_metrics(const codlad_metric_inputs *in, float *out8, void *scratch, void *stream) {
    CODLAD_REQUIRE(in && out8 && scratch, "null pointer");
    CODLAD_REQUIRE(in->n_atoms >= 0 && in->n_edges >= 0 && in->n_clash >= 0 && in->n_bb >= 0 && in->n_inter >= 0 &&
                       in->n_pipi >= 0 && in->n_ic >= 0, "negative count");
    CODLAD_REQUIRE((!in->n_atoms || (in->xyz && in->xyz_recon)) && (!in->n_edges || in->edge_list) &&
                       (!in->n_clash || in->clash_list) && (!in->n_bb || in->bb_NO_list) &&
                       (!in->n_inter || in->interaction_list) && (!in->n_pipi || in->pi_pi_list) &&
                       (!in->n_ic || (in->ic && in->ic_recon && in->ic_mask)), "a non-empty list has a null pointer");
    CODLAD_REQUIRE(in->n_atoms > 0 || (in->n_edges + in->n_clash + in->n_bb + in->n_inter + in->n_pipi) == 0,
                   "index lists without coordinates");
    const int64_t items = in->n_atoms + in->n_edges + in->n_clash + in->n_bb + in->n_inter + in->n_pipi + in->n_ic;
    int blocks = (int)((items + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > MAX_BLOCKS ? MAX_BLOCKS : blocks);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(metrics_partial_kernel, dim3(blocks), dim3(256), 0, st, *in, (double *)scratch);
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(1), dim3(64), 0, st, *in, (const double *)scratch, blocks, out8);
    return codlad_check_launch("codlad_eval_metrics");
}
