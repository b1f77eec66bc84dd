// gfx950 kernels: batch-norm statistics / backward, and the fused loss forward+backward.
#include <cstdlib>
#include "kernels.h"
#include "device_utils.h"

namespace cunvsm {

// =============================================================================================
// Batch normalisation, per-activation, γ ≡ 1, β = projection bias, batch statistics only
// (replaces cudnnBatchNormalizationForwardTraining/Backward, cpp/cudnn_utils.cu:107-124,158-177).
// The forward column sums Σx, Σx² come from the projection GEMM's epilogue (gather_gemm.hip): fp32 over a
// 128-row tile, merged with native fp64 atomics, so var = E[x²] − E[x]² is formed in double.
// =============================================================================================
// Batch statistics of V columns from the fp64 column sums Σx, Σx² the projection GEMM left (cudnn_utils.cu:107-124:
// biased variance, 1/sqrt(σ² + ε)). Every wave of the loss kernel evaluates this for its own columns — a few dozen
// fp64 operations — instead of a separate one-block launch on the critical stream between the GEMM and the loss.
template <int V>
__device__ __forceinline__ void bn_stats_from_sums(const double* __restrict__ sums, int dim, int c, double n, float eps,
                                                   float (&mean)[V], float (&inv_std)[V]) {
#pragma clang fp contract(on)       // by the language rule: the same values in every kernel this is inlined into
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const double m = sums[c + i] / n;
        double var = sums[dim + c + i] / n - m * m;            // biased variance
        if (var < 0.0) var = 0.0;
        mean[i] = static_cast<float>(m);
        inv_std[i] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    }
}

__global__ void colsum_finalize_kernel(const double* __restrict__ sums, int dim, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < dim) out[c] = static_cast<float>(sums[c]);
}

void launch_colsum_finalize(const double* sums, int dim, float* out, hipStream_t s) {
    NVSM_LAUNCH(colsum_finalize_kernel, dim3(ceil_div(dim, 256)), dim3(256), 0, s, sums, dim, out);
}

// dx = invσ · (dy − (dβ + x̂·dγ) / N)       (= invσ/N · (N·dy − dβ − x̂·dγ), cuDNN per-activation backward)
template <int V>
__global__ __launch_bounds__(256) void bn_dx_kernel(float* __restrict__ dy, const float* __restrict__ pre,
                                                    const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                    const double* __restrict__ sums, float* __restrict__ dbeta,
                                                    float* __restrict__ dgamma, float* __restrict__ grad_bias,
                                                    float inv_n, uint32_t total, uint32_t nvec, int dim) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        const uint32_t b = q / nvec;
        const uint32_t c = (q - b * nvec) * V;
        const size_t off = static_cast<size_t>(b) * dim + c;
        float g[V], x[V], mu[V], is[V], db[V], dg[V];
        ldv<V>(dy + off, g);
        ldv<V>(pre + off, x);
        ldv<V>(mean + c, mu);
        ldv<V>(inv_std + c, is);
#pragma unroll
        for (int i = 0; i < V; ++i) {              // dβ = Σdy, dγ = Σdy·x̂ as floats (cudnn_utils.cu:158-173)
            db[i] = static_cast<float>(sums[c + i]);
            dg[i] = static_cast<float>(sums[dim + c + i]);
        }
        if (b == 0) {                              // ∂β = Σdy is the bias gradient; ∂γ is computed and dropped (:173)
            stv<V>(dbeta + c, db);
            stv<V>(dgamma + c, dg);
            stv<V>(grad_bias + c, db);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float xhat = (x[i] - mu[i]) * is[i];
            g[i] = is[i] * (g[i] - (db[i] + xhat * dg[i]) * inv_n);
        }
        stv<V>(dy + off, g);
    }
}

void launch_bn_dx(float* dy, const float* pre, const float* mean, const float* inv_std, const double* sums, float* dbeta,
                  float* dgamma, float* grad_bias, double n_global, int64_t rows, int dim, hipStream_t s) {
    if (rows <= 0) return;
    const float inv_n = static_cast<float>(1.0 / n_global);
    if (dim % 4 == 0) {
        const uint32_t nvec = dim / 4, total = static_cast<uint32_t>(rows * nvec);
        NVSM_LAUNCH(bn_dx_kernel<4>, dim3(stream_grid(total, 256)), dim3(256), 0, s, dy, pre, mean, inv_std,
                           sums, dbeta, dgamma, grad_bias, inv_n, total, nvec, dim);
    } else {
        const uint32_t nvec = dim, total = static_cast<uint32_t>(rows * nvec);
        NVSM_LAUNCH(bn_dx_kernel<1>, dim3(stream_grid(total, 256)), dim3(256), 0, s, dy, pre, mean, inv_std,
                           sums, dbeta, dgamma, grad_bias, inv_n, total, nvec, dim);
    }
}

// =============================================================================================
// Fused loss: one wave per example. The example's projection row (entity_dim floats) stays in
// registers (float4 per lane at de = 256), the R = k+1 document rows are streamed with one
// coalesced 1 KB load each, the dot products are wave-shuffle reductions, and the kernel emits
//   proj[b] = act(BN(pre[b]))                      (cpp/params.cu:425-446)
//   p_j = clamp(σ(±E[id_j]·proj[b]))               (cpp/objective.cu:184-246)
//   Σ ω_j·log p_j                                  (:250-305)
//   coef_j = ±ω_j·dlogσ(p_j)/B                     (:354-401)   signed multiplier
//   dy[b] = (Σ_r coef_j·E[id_j]) · act'(proj[b])   (:420-425, cpp/params.cu:474-491)
//   column sums Σdy, Σdy·x̂                        (grad_bias / BN backward statistics)
// replacing F8–F16 and B1–B4 and their four (N x de) temporaries with a single pass over E.
// A block of 4 waves walks kExamplesPerWave examples per wave so that the column statistics need one
// fp64 atomic per column per 64 examples.
// =============================================================================================
constexpr int kExamplesPerWave = 8;
constexpr int kRowGroup = 4;

// The workgroup's share of [loss | Σdy | Σdy·x̂] (four waves' column sums in LDS) goes into the ordered grid-wide sum
// (device_utils.h grid_sum_ordered) whose result lands in a.loss_acc[0 .. 2 de]: value 0 is the loss word, 1 .. de the
// column sums of dy, de + 1 .. 2 de those of dy·x̂ (batch-norm only).
__device__ __forceinline__ void loss_block_sums(const LossArgs& a, const float* s_dy, const float* s_dyx, const float* s_loss, int* flag) {
    const int de = a.de;
    const int n = 1 + (a.bn ? 2 : 1) * de;
    auto val = [&](int i) -> float {
        if (i == 0) return (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        const float* sp = (i <= de) ? s_dy + (i - 1) : s_dyx + (i - 1 - de);
        return (sp[0] + sp[de]) + (sp[2 * de] + sp[3 * de]);
    };
    double* dst = a.loss_acc;
    grid_sum_ordered<256>(a.sums.part, a.sums.part2, a.sums.arrive, a.sums.fan, n, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x),
                          val, [&](int i, double v) { dst[i] = v; }, flag);
}

template <int V, int NITER, bool L2E = false>
__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
    extern __shared__ float lds[];          // [2][4][de] column stats + [4] loss + the grid sum's flag
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int de = a.de, R = a.R;
    const int64_t e0 = (static_cast<int64_t>(blockIdx.x) * 4 + wid) * kExamplesPerWave;
    const int64_t e1 = min(a.B, e0 + kExamplesPerWave);

    float sdy[NITER][V], sdyx[NITER][V];
    float mu[NITER][V], is[NITER][V], beta[NITER][V];
    bool valid[NITER];
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
        const int c = (it * 64 + lane) * V;
        valid[it] = (c < de);
#pragma unroll
        for (int i = 0; i < V; ++i) { sdy[it][i] = 0.f; sdyx[it][i] = 0.f; mu[it][i] = 0.f; is[it][i] = 1.f; beta[it][i] = 0.f; }
        if (a.bn && valid[it]) {
            bn_stats_from_sums<V>(a.bn_sums, de, c, a.bn_n, a.bn_eps, mu[it], is[it]);
            if (blockIdx.x == 0 && wid == 0) { stv<V>(a.bn_mean + c, mu[it]); stv<V>(a.bn_inv_std + c, is[it]); }   // for bn_dx
            ldv<V>(a.bias + c, beta[it]);
        }
    }
    float wave_loss = 0.f;

    for (int64_t b = e0; b < e1; ++b) {
        float out[NITER][V], xhat[NITER][V], gp[NITER][V];
        float ssq = 0.f;
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int c = (it * 64 + lane) * V;
#pragma unroll
            for (int i = 0; i < V; ++i) { out[it][i] = 0.f; xhat[it][i] = 0.f; gp[it][i] = 0.f; }
            if (valid[it]) {
                float x[V];
                ldv<V>(a.pre + b * de + c, x);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float y = x[i];
                    if (a.bn) { xhat[it][i] = (x[i] - mu[it][i]) * is[it][i]; y = xhat[it][i] + beta[it][i]; }
                    y = (a.nonlinearity == 0) ? tanhf(y) : fminf(fmaxf(y, a.clip_min), a.clip_max);
                    out[it][i] = y;
                    ssq += y * y;
                }
                stv<V>(a.proj + b * de + c, out[it]);
            }
        }
        ssq = wave_sum(ssq);
        if (lane == 0) a.pp[b] = ssq * a.inv_de;

        float w = a.inst_w ? a.inst_w[b] : 1.f;
        if (a.rebalance) w = w * a.neg_scale;                       // objective.cu:268-274
        const float w_pos = a.rebalance ? w * static_cast<float>(a.k) : w;   // :282-290

        // Stream the R document rows in groups of kRowGroup: while one group is being reduced the next
        // group's kRowGroup independent 1 KB loads are already in flight.
        float e_next[kRowGroup][NITER][V];
#pragma unroll
        for (int u = 0; u < kRowGroup; ++u) {
#pragma unroll
            for (int it = 0; it < NITER; ++it)
#pragma unroll
                for (int i = 0; i < V; ++i) e_next[u][it][i] = 0.f;
            if (u < R) {
                const size_t id = static_cast<size_t>(a.ids[b * R + u]);
#pragma unroll
                for (int it = 0; it < NITER; ++it)
                    if (valid[it]) ldv<V>(a.E + id * de + (it * 64 + lane) * V, e_next[u][it]);
            }
        }
        for (int r0 = 0; r0 < R; r0 += kRowGroup) {
            float e[kRowGroup][NITER][V];
#pragma unroll
            for (int u = 0; u < kRowGroup; ++u)
#pragma unroll
                for (int it = 0; it < NITER; ++it)
#pragma unroll
                    for (int i = 0; i < V; ++i) e[u][it][i] = e_next[u][it][i];
#pragma unroll
            for (int u = 0; u < kRowGroup; ++u) {
                const int rn = r0 + kRowGroup + u;
                if (rn < R) {
                    const size_t id = static_cast<size_t>(a.ids[b * R + rn]);
#pragma unroll
                    for (int it = 0; it < NITER; ++it)
                        if (valid[it]) ldv<V>(a.E + id * de + (it * 64 + lane) * V, e_next[u][it]);
                }
            }
            if (L2E) {
                // l2_normalize_entity_reprs (objective.cu:168-174, Normalizer::forward cuda_utils.cu:12-46): every gathered
                // row is divided by its norm before it is negated / multiplied; the loss and the projection gradient
                // see the normalised rows (the gradient w.r.t. the table is taken in materialize_grad_entity_l2_kernel)
                float nsq[kRowGroup];
#pragma unroll
                for (int u = 0; u < kRowGroup; ++u) {
                    float t = 0.f;
#pragma unroll
                    for (int it = 0; it < NITER; ++it)
#pragma unroll
                        for (int i = 0; i < V; ++i) t += e[u][it][i] * e[u][it][i];
                    nsq[u] = t;
                }
#pragma unroll
                for (int u = 0; u < kRowGroup; ++u) nsq[u] = wave_sum(nsq[u]);
#pragma unroll
                for (int u = 0; u < kRowGroup; ++u) {
                    const float nrm = sqrtf(nsq[u]);
#pragma unroll
                    for (int it = 0; it < NITER; ++it)
#pragma unroll
                        for (int i = 0; i < V; ++i) e[u][it][i] = (r0 + u < R) ? e[u][it][i] / nrm : 0.f;
                }
            }
            float dot[kRowGroup];
#pragma unroll
            for (int u = 0; u < kRowGroup; ++u) {
                float d = 0.f;
#pragma unroll
                for (int it = 0; it < NITER; ++it)
#pragma unroll
                    for (int i = 0; i < V; ++i) d += out[it][i] * e[u][it][i];
                dot[u] = d;
            }
#pragma unroll
            for (int u = 0; u < kRowGroup; ++u) dot[u] = wave_sum(dot[u]);
#pragma unroll
            for (int u = 0; u < kRowGroup; ++u) {
                const int r = r0 + u;
                if (r >= R) break;
                const float sign = (r == 0) ? 1.f : -1.f;                // objective.cu:184-187
                const float sx = sign * dot[u];
                float p = (sx >= 0.f) ? 1.f / (1.f + expf(-sx)) : expf(sx) / (1.f + expf(sx));   // cuda_utils.h:205-207
                p = fminf(fmaxf(p, a.sig_eps), a.sig_hi);                // :209
                const float wj = (r == 0) ? w_pos : w;
                wave_loss += logf(p) * wj;                               // objective.cu:250-305
                const float d = (static_cast<double>(p) >= a.d_hi || p <= a.d_eps) ? 0.f : 1.f - p;   // cuda_utils.h:229-231
                const float m = wj * (d * a.inv_batch);                  // objective.cu:357-371
                const float cf = sign * m;
                if (lane == 0) {
                    a.coef[b * R + r] = cf;
                    a.probs[b * R + r] = p;
                }
#pragma unroll
                for (int it = 0; it < NITER; ++it)
#pragma unroll
                    for (int i = 0; i < V; ++i) gp[it][i] += cf * e[u][it][i];   // fold_columns, :420-425
            }
        }

        // nonlinearity' on the OUTPUT (params.cu:474-491) and column statistics
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            if (!valid[it]) continue;
            float g[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float y = out[it][i];
                const float dd = (a.nonlinearity == 0) ? (1.f - y * y) : ((y > a.clip_min && y < a.clip_max) ? 1.f : 0.f);
                g[i] = dd * gp[it][i];
                sdy[it][i] += g[i];
                sdyx[it][i] += g[i] * xhat[it][i];
            }
            stv<V>(a.dy + b * de + (it * 64 + lane) * V, g);
        }
    }

    // ---- block reduction of the column statistics and the loss ----
    float* s_dy = lds;                       // [4][de]
    float* s_dyx = lds + 4 * de;             // [4][de]
    float* s_loss = lds + 8 * de;            // [4]
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
        const int c = (it * 64 + lane) * V;
        if (!valid[it]) continue;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            s_dy[wid * de + c + i] = sdy[it][i];
            s_dyx[wid * de + c + i] = sdyx[it][i];
        }
    }
    if (lane == 0) s_loss[wid] = wave_loss;
    __syncthreads();
    loss_block_sums(a, s_dy, s_dyx, s_loss, reinterpret_cast<int*>(s_loss + 4));
}

// ---------------------------------------------------------------------------------------------
// Fast path (entity_dim % 4 == 0, entity_dim <= 256, R <= 64): one float4 column slice per lane, the ids of an
// example held one per lane, RB document rows in flight per wave (scalar row base + 32-bit lane offset, no
// per-lane bounds branches: out-of-range columns / rows are clamped to a harmless re-read and masked in the
// arithmetic), the NEXT example's projection row and ids prefetched behind them, and the sigmoid / log /
// multiplier arithmetic done once per row in lane r instead of redundantly in all 64 lanes. Measured alone at
// the NVSM config: 151 us vs 234 us for the generic kernel below; a gather-only kernel with the same access
// pattern (no arithmetic, no outputs) takes 132 us.
// ---------------------------------------------------------------------------------------------
// LAZY: E decays lazily (kernels.h LazyView): a gathered row gets the factors of the updates it sat out, one by one. The
// stamps of the example's rows are fetched by the lanes that hold their ids (one vector load, in front of the row loads);
// the factor history lives in two registers (lane l: factors l and l + 64) and is read with v_readlane — the update
// counter is wave-uniform, so the loop costs a handful of scalar / VALU cycles per pending update and no memory access.
template <bool LAZY>
__device__ __forceinline__ void loss_refresh_row(int from, int now, float dlo, float dhi, float (&e)[4]) {
    if (LAZY) {
        for (int u = from; u < now; ++u) {
            const int src = __builtin_bit_cast(int, (u & 64) ? dhi : dlo);
            const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(src, u & 63));
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] *= d;
        }
    }
}

// PIPE (R <= RB, the whole example's rows in one set): two register sets — the rows of example b + 1 are requested BEFORE the
// arithmetic of example b, so a wave always has a set in flight (up to 2 RB rows) instead of alternating between "RB rows
// requested" and "none while it computes"; two waves per SIMD instead of three. The arithmetic and its order are the same
// statements (finish()): results are bit-identical to the single-set form.
template <int RB, bool LAZY = false, bool PIPE = false>
__global__ __launch_bounds__(256) void loss_rows_kernel(LossArgs a, int ex_per_wave) {
    // Floating-point contraction by the language rule (a product and a sum in ONE expression fuse, nothing else does),
    // not by the optimiser's choice: the LAZY and the eager instantiation — different loop structures around the same
    // arithmetic — were contracted differently under the default (Σdy came out one ulp apart) and have to agree bit for bit.
    // (Spelling the fusions out with fmaf under contract(off) pinned them too but kept the compiler from pairing them into
    // v_pk_fma_f32: 155 -> 166 us.)
#pragma clang fp contract(on)
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int de = a.de, R = a.R;
    const int c = lane * 4;
    const bool valid = c < de;
    const uint32_t coff = valid ? static_cast<uint32_t>(c) * 4u : 0u;      // byte offset inside a row (clamped: harmless re-read)
    const int64_t e0 = (static_cast<int64_t>(blockIdx.x) * 4 + wid) * ex_per_wave;
    const int64_t e1 = min(a.B, e0 + ex_per_wave);

    float sdy[4] = {0, 0, 0, 0}, sdyx[4] = {0, 0, 0, 0};
    float mu[4] = {0, 0, 0, 0}, is[4] = {1, 1, 1, 1}, beta[4] = {0, 0, 0, 0};
    float lane_loss = 0.f;
    const int lane_r = lane < R ? lane : R - 1;
    float dlo = 1.f, dhi = 1.f;
    if (LAZY) { dlo = a.lazyE.decay[lane]; dhi = a.lazyE.decay[lane + 64]; }
    // The batch statistics μ, 1/sqrt(σ² + ε) of the columns: a few dozen fp64 operations per column (two divisions, a square
    // root, a reciprocal). Every wave used to evaluate its own four columns — the same 256 values four times per workgroup, a
    // microsecond and a half of each wave's life in front of its first load, which shows at per-rank batch sizes where a wave
    // lives for two or three examples. Now wave w evaluates columns 64 w .. 64 w + 63, one per lane, and the workgroup shares them
    // through LDS (the same function on the same inputs: the same bits); the first example's inputs are requested before that.
    auto bn_setup = [&]() __attribute__((always_inline)) {
        if (!a.bn) return;
        float* s_mu = lds + 8 * de + 8;       // [de] | [de] behind the column-sum scratch of the epilogue
        float* s_is = s_mu + de;
        for (int col = threadIdx.x; col < de; col += 256) {
            float m1[1], i1[1];
            bn_stats_from_sums<1>(a.bn_sums, de, col, a.bn_n, a.bn_eps, m1, i1);
            s_mu[col] = m1[0]; s_is[col] = i1[0];
            if (blockIdx.x == 0) { a.bn_mean[col] = m1[0]; a.bn_inv_std[col] = i1[0]; }       // for bn_dx
        }
        ldv<4>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.bias) + coff), beta);
        __syncthreads();
        if (valid) { ldv<4>(s_mu + c, mu); ldv<4>(s_is + c, is); }
    };

    // rows r0 .. r0 + RB - 1 of the example whose ids the lanes hold in `ids_lane`, one 16 B column slice per lane
    auto issue_rows = [&](float (&e)[RB][4], int ids_lane, int r0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = min(r0 + u, R - 1);
            const size_t id = static_cast<size_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(ids_lane, r)));
            const char* rowp = reinterpret_cast<const char*>(a.E + id * de);
            ldv<4>(reinterpret_cast<const float*>(rowp + coff), e[u]);
        }
    };
    // everything of example b behind its loads: projection row, the R dot products, loss, multipliers, dy and the column sums
    // (`e` holds rows 0 .. RB - 1 on entry; further sets of RB rows are fetched here)
    auto finish = [&](int64_t b, const float (&x)[4], int myid, float w, int mystamp, float (&e)[RB][4]) __attribute__((always_inline)) {
        float out[4], xhat[4] = {0, 0, 0, 0}, gp[4] = {0, 0, 0, 0};
        float ssq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y = x[i];
            if (a.bn) { xhat[i] = (x[i] - mu[i]) * is[i]; y = xhat[i] + beta[i]; }
            y = (a.nonlinearity == 0) ? tanhf(y) : fminf(fmaxf(y, a.clip_min), a.clip_max);
            y = valid ? y : 0.f;
            out[i] = y;
            ssq += y * y;
        }
        if (valid) stv<4>(a.proj + b * de + c, out);
        ssq = wave_sum(ssq);
        if (lane == 0) a.pp[b] = ssq * a.inv_de;

        if (a.rebalance) w = w * a.neg_scale;
        const float w_pos = a.rebalance ? w * static_cast<float>(a.k) : w;

        for (int r0 = 0; r0 < R; r0 += RB) {
            if (r0 > 0) issue_rows(e, myid, r0);
            if (LAZY) {
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int r = min(r0 + u, R - 1);
                    loss_refresh_row<LAZY>(__builtin_amdgcn_readlane(mystamp, r), a.lazyE.now, dlo, dhi, e[u]);
                }
            }
            float dot[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) d += out[i] * e[u][i];
                dot[u] = d;
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) dot[u] = wave_sum(dot[u]);
            float dv = 0.f;
#pragma unroll
            for (int u = 0; u < RB; ++u) dv = (lane == u) ? dot[u] : dv;
            const int r = r0 + lane;
            const bool rv = (lane < RB) && (r < R);
            const float sign = (r == 0) ? 1.f : -1.f;
            const float sx = sign * dv;
            float p = (sx >= 0.f) ? 1.f / (1.f + expf(-sx)) : expf(sx) / (1.f + expf(sx));
            p = fminf(fmaxf(p, a.sig_eps), a.sig_hi);
            const float wj = (r == 0) ? w_pos : w;
            lane_loss += rv ? logf(p) * wj : 0.f;
            const float d = (static_cast<double>(p) >= a.d_hi || p <= a.d_eps) ? 0.f : 1.f - p;
            const float m = wj * (d * a.inv_batch);
            const float cf = rv ? sign * m : 0.f;
            if (rv) { a.coef[b * R + r] = cf; a.probs[b * R + r] = p; }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const float cu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cf), u));
#pragma unroll
                for (int i = 0; i < 4; ++i) gp[i] += cu * e[u][i];
            }
        }
        float g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float y = out[i];
            const float dd = (a.nonlinearity == 0) ? (1.f - y * y) : ((y > a.clip_min && y < a.clip_max) ? 1.f : 0.f);
            g[i] = valid ? dd * gp[i] : 0.f;
            sdy[i] += g[i];
            sdyx[i] += g[i] * xhat[i];
        }
        if (valid) stv<4>(a.dy + b * de + c, g);
    };
    auto load_inputs = [&](int64_t b, float (&x)[4], int& id, float& w) __attribute__((always_inline)) {
        ldv<4>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.pre + b * de) + coff), x);
        id = a.ids[b * R + lane_r];
        if (a.inst_w) w = a.inst_w[b];
    };

    if constexpr (!PIPE) {
        float xn[4] = {0, 0, 0, 0};
        int idn = 0, stampn = 0;
        float wn = 1.f;
        if (e0 < e1) load_inputs(e0, xn, idn, wn);
        bn_setup();
        if (e0 < e1) {
            if (LAZY) stampn = a.lazyE.stamp[static_cast<uint32_t>(idn)];
        }
        for (int64_t b = e0; b < e1; ++b) {
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = xn[i];
            const int myid = idn;
            const float w = wn;
            // (LAZY: the stamps of this example's rows were requested at the end of the turn before — a dependent load behind the
            //  ids, which it used to follow right here, in front of the row loads that need the ids too)
            const int mystamp = stampn;
            float e[RB][4];
            issue_rows(e, myid, 0);
            if (b + 1 < e1) load_inputs(b + 1, xn, idn, wn);
            finish(b, x, myid, w, mystamp, e);
            if (LAZY && b + 1 < e1) stampn = a.lazyE.stamp[static_cast<uint32_t>(idn)];      // (idn arrived long ago)
        }
    } else {
        // inputs run two examples ahead (the ids of b + 1 must have arrived when its rows are requested, at the start of turn b)
        float xc[4] = {0, 0, 0, 0}, x1[4] = {0, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
        int idc = 0, id1 = 0, id2 = 0, stampc = 0, stamp1 = 0;
        float wc = 1.f, w1 = 1.f, w2 = 1.f;
        float eA[RB][4], eB[RB][4];
        if (e0 < e1) load_inputs(e0, xc, idc, wc);
        bn_setup();
        if (e0 < e1) {
            if (LAZY) stampc = a.lazyE.stamp[static_cast<uint32_t>(idc)];
            issue_rows(eA, idc, 0);
            if (e0 + 1 < e1) load_inputs(e0 + 1, x1, id1, w1);
        }
        auto turn = [&](int64_t b, float (&cur)[RB][4], float (&nxt)[RB][4]) __attribute__((always_inline)) {
            if (b + 1 < e1) {
                issue_rows(nxt, id1, 0);
                if (LAZY) stamp1 = a.lazyE.stamp[static_cast<uint32_t>(id1)];
            }
            if (b + 2 < e1) load_inputs(b + 2, x2, id2, w2);
            finish(b, xc, idc, wc, stampc, cur);
#pragma unroll
            for (int i = 0; i < 4; ++i) { xc[i] = x1[i]; x1[i] = x2[i]; }
            idc = id1; id1 = id2; wc = w1; w1 = w2; stampc = stamp1;
        };
        for (int64_t b = e0; b < e1; b += 2) {
            turn(b, eA, eB);
            if (b + 1 < e1) turn(b + 1, eB, eA);
        }
    }
    const float wave_loss = wave_sum(lane_loss);
    float* s_dy = lds; float* s_dyx = lds + 4 * de; float* s_loss = lds + 8 * de;
    if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s_dy[wid * de + c + i] = sdy[i]; s_dyx[wid * de + c + i] = sdyx[i]; }
    }
    if (lane == 0) s_loss[wid] = wave_loss;
    __syncthreads();
    loss_block_sums(a, s_dy, s_dyx, s_loss, reinterpret_cast<int*>(s_loss + 4));
}

// (Round 3 tried the document rows landing in LDS instead of registers — entity_dim = 256 makes a row exactly the 1 KB one
//  global_load_lds_dwordx4 moves per wave; two 17 KB slots per wave, the next example's rows in flight by LDS-DMA under this
//  example's arithmetic, bit-identical outputs. One workgroup of four waves then owns a CU's LDS: 68 rows in flight per CU
//  where twelve register-landing waves hold up to 204, and the kernel took 222 instead of 180 us in-step at |D| = 100 k, 383
//  instead of 279 us at |D| = 2 M. The register file (512 KB per CU) is the larger landing buffer; removed.)

template <int V, int NITER>
static void launch_loss_t(const LossArgs& a_in, hipStream_t s) {
    const int grid = ceil_div(a_in.B, 4 * kExamplesPerWave);
    LossArgs a = a_in;
    a.sums.fan = grid_sum_fan(grid);
    const size_t shmem = (8 * static_cast<size_t>(a.de) + 8) * sizeof(float);
    if (a.l2_entity) NVSM_LAUNCH((loss_kernel<V, NITER, true>), dim3(grid), dim3(256), shmem, s, a);
    else NVSM_LAUNCH((loss_kernel<V, NITER, false>), dim3(grid), dim3(256), shmem, s, a);
}

template <int RB>
static void launch_loss_rows(const LossArgs& a_in, hipStream_t s) {
    LossArgs a = a_in;
    // ~5 blocks per CU; at most 16 examples per wave ...
    int epw = static_cast<int>((a.B + 4 * 1280 - 1) / (4 * 1280));
    epw = epw < 1 ? 1 : (epw > 16 ? 16 : epw);
    // ... and twenty from 40 k examples: 640 workgroups, fewer than the 768 the chip holds at once (three per CU at 161
    // registers). A workgroup ends by handing its column sums over (device_utils.h grid_sum_ordered: stores, a wait for them,
    // a counter) — 3-4 us of idle waves that a second round of workgroups queues behind; in a single round only the last
    // hand-over shows. 51 200 examples: 180 -> 176 us in-step (1 280 / 800 / 753 workgroups: 180 / 196 / 183 us).
    if (a.B >= 40960) epw = 20;
    // (batch 4096: two examples per wave, 512 workgroups — half as many fp64 atomics per column: 44 -> 33 us)
    if (epw < 2 && a.B >= 2048) epw = 2;
    // per-rank batches (round 5): about 400-500 workgroups — ONE round of workgroups with room to spare (768 fit) instead of a
    // round and a bit (batch 6 400: 800 workgroups of two examples per wave; the stragglers of the second round cost a whole
    // workgroup's latency): alone 47.9 -> 42.8 us, step 0.2637 -> 0.2569 ms with four examples per wave (3: 0.2604; 5, 6: slower)
    if (a.B > 4096 && a.B <= 8192) epw = static_cast<int>((a.B + 4 * 512 - 1) / (4 * 512));
    // Two row sets per wave (PIPE; two workgroups fit a CU) where the documents table does not fit the 256 MB Infinity Cache and
    // every row comes out of HBM at 1 800 cycles: |D| = 2 M, batch 51 200: 295 -> 244 us in the step (0.44 -> 0.54 of peak) with
    // one round of 512 workgroups. But two such workgroups leave a CU no registers for anything else, and what runs next to this
    // kernel is the documents CSR build, which the documents update — the longest kernel of that step — waits for: with ~380
    // workgroups (one and a half per CU; 34 examples per wave at batch 51 200) the kernel takes 270 us and the STEP 1.611 -> 1.548
    // ms (30 examples per wave: 1.567, 32: 1.572, 36: 1.557, 38: 1.567, 40: 1.60, 50: 1.63). Where the table is cache-resident
    // (|D| = 100 k) the two-set kernel gains 6 us (179 -> 173) and the STEP loses 8 (0.894 -> 0.902) for the same reason: single
    // set. Interleaved A/B, tools/ab_roof.sh; NVSM_LOSS_PIPE=0/1, NVSM_LOSS_EPW (experiments build) override.
    const bool pipe = a.R <= RB && loss_two_row_sets(a.E_rows, a.de);
    if (pipe && a.B >= 40960) epw = static_cast<int>((a.B + 4 * 380 - 1) / (4 * 380));
    const int epw_env = tuning().loss_epw;      // experiments
    if (epw_env > 0) epw = epw_env;
    const int grid = ceil_div(a.B, 4 * epw);
    a.sums.fan = grid_sum_fan(grid);
    const size_t shmem = (10 * static_cast<size_t>(a.de) + 8) * sizeof(float);      // column-sum scratch [2][4][de] | loss [4] | flag | μ [de] | 1/σ [de]
    if (pipe) {
        if (a.lazyE.stamp) NVSM_LAUNCH((loss_rows_kernel<RB, true, true>), dim3(grid), dim3(256), shmem, s, a, epw);
        else NVSM_LAUNCH((loss_rows_kernel<RB, false, true>), dim3(grid), dim3(256), shmem, s, a, epw);
        return;
    }
    if (a.lazyE.stamp) NVSM_LAUNCH((loss_rows_kernel<RB, true>), dim3(grid), dim3(256), shmem, s, a, epw);
    else NVSM_LAUNCH((loss_rows_kernel<RB, false>), dim3(grid), dim3(256), shmem, s, a, epw);
}

bool loss_two_row_sets(int64_t table_rows, int de) {
    const int pipe_env = tuning().loss_pipe;
    return pipe_env >= 0 ? pipe_env != 0 : table_rows * static_cast<int64_t>(de) * 4 > (static_cast<int64_t>(256) << 20);
}

bool loss_reads_lazily(int de, int R, bool l2_entity) { return de % 4 == 0 && de <= 256 && R <= 64 && !l2_entity; }

void launch_loss(const LossArgs& a, hipStream_t s) {
    if (a.B <= 0) return;
    const int de = a.de;
    if (de % 4 == 0 && de <= 256 && a.R <= 64 && !a.l2_entity) {      // (the optional entity normaliser: generic kernel)
        if (a.R <= 6) launch_loss_rows<6>(a, s);
        else if (a.R <= 11) launch_loss_rows<11>(a, s);
        else launch_loss_rows<17>(a, s);
        return;
    }
    if (de % 4 == 0) {
        if (de <= 256) launch_loss_t<4, 1>(a, s);
        else if (de <= 512) launch_loss_t<4, 2>(a, s);
        else launch_loss_t<4, 4>(a, s);                 // de ≤ 1024, the reference's own limit (block = dim)
    } else {
        if (de <= 64) launch_loss_t<1, 1>(a, s);
        else if (de <= 128) launch_loss_t<1, 2>(a, s);
        else launch_loss_t<1, 4>(a, s);                 // odd dims up to 256
    }
}

// out[b] = Σ_t G[b][t]² · inv_dim — one wave per row (cpp/updates_adam.cu:232-240, updates_adagrad.cu:136-143)
__global__ __launch_bounds__(256) void row_meansq_kernel(const float* __restrict__ G, int64_t rows, int dim, float inv_dim,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t b = blockIdx.x * 4 + (threadIdx.x >> 6); b < rows; b += static_cast<int64_t>(gridDim.x) * 4) {
        float s = 0.f;
        for (int t = lane; t < dim; t += 64) {
            const float v = G[b * dim + t];
            s += v * v;
        }
        s = wave_sum(s);
        if (lane == 0) out[b] = s * inv_dim;
    }
}

void launch_row_meansq(const float* G, int64_t rows, int dim, float inv_dim, float* out, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(row_meansq_kernel, dim3(stream_grid(rows * 64, 256)), dim3(256), 0, s, G, rows, dim, inv_dim, out);
}

__global__ void materialize_grad_entity_kernel(const float* __restrict__ coef, const float* __restrict__ proj, int64_t N,
                                               int R, int de, float* __restrict__ out) {
    const int64_t total = N * de;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t j = i / de;
        const int t = static_cast<int>(i - j * de);
        out[i] = proj[(j / R) * de + t] * coef[j];
    }
}

void launch_materialize_grad_entity(const float* coef, const float* proj, int64_t N, int R, int de, float* out, hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(materialize_grad_entity_kernel, dim3(stream_grid(N * de, 256)), dim3(256), 0, s, coef, proj, N, R, de, out);
}

// ---------------------------------------------------------------------------------------------
// L2 row normaliser (optional: --l2_phrase_normalization / --l2_entity_normalization, off in both recipes).
// Normalizer::forward / backward, cpp/cuda_utils.cu:12-130: y = x / ‖x‖;  grad_in = (g·‖x‖² − x·(x·g)) / ‖x‖³.
// One wave per row; not tuned — these are small streaming passes next to the gathers.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_rows_forward_kernel(const float* __restrict__ x, int64_t rows, int dim,
                                                              float* __restrict__ y, float* __restrict__ norms) {
    const int lane = threadIdx.x & 63;
    for (int64_t b = blockIdx.x * 4 + (threadIdx.x >> 6); b < rows; b += static_cast<int64_t>(gridDim.x) * 4) {
        float s = 0.f;
        for (int t = lane; t < dim; t += 64) { const float v = x[b * dim + t]; s += v * v; }
        const float n = sqrtf(wave_sum(s));
        if (lane == 0) norms[b] = n;
        for (int t = lane; t < dim; t += 64) y[b * dim + t] = x[b * dim + t] / n;
    }
}
void launch_l2_rows_forward(const float* x, int64_t rows, int dim, float* y, float* norms, hipStream_t s) {
    if (rows > 0) hipLaunchKernelGGL(l2_rows_forward_kernel, dim3(stream_grid(rows * 64, 256)), dim3(256), 0, s, x, rows, dim, y, norms);
}

// gin = scale · (g·n² − x·(x·g)) / n³ (in place over g allowed), msq[b] = mean_t(gin²) when msq != null
__global__ __launch_bounds__(256) void l2_rows_backward_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                               const float* __restrict__ norms, int64_t rows, int dim,
                                                               float scale, float inv_dim, float* __restrict__ gin,
                                                               float* __restrict__ msq) {
    const int lane = threadIdx.x & 63;
    for (int64_t b = blockIdx.x * 4 + (threadIdx.x >> 6); b < rows; b += static_cast<int64_t>(gridDim.x) * 4) {
        float cr = 0.f;
        for (int t = lane; t < dim; t += 64) cr += x[b * dim + t] * g[b * dim + t];
        cr = wave_sum(cr);
        const float n = norms[b], n2 = n * n, n3 = n2 * n;
        float sq = 0.f;
        for (int t = lane; t < dim; t += 64) {
            const float v = ((g[b * dim + t] * n2 - x[b * dim + t] * cr) / n3) * scale;
            gin[b * dim + t] = v;
            sq += v * v;
        }
        if (msq) { sq = wave_sum(sq); if (lane == 0) msq[b] = sq * inv_dim; }
    }
}
void launch_l2_rows_backward(const float* g, const float* x, const float* norms, int64_t rows, int dim, float scale,
                             float* gin, float* msq, hipStream_t s) {
    if (rows <= 0) return;
    const float inv_dim = static_cast<float>(std::exp(-std::log(static_cast<double>(dim))));
    hipLaunchKernelGGL(l2_rows_backward_kernel, dim3(stream_grid(rows * 64, 256)), dim3(256), 0, s, g, x, norms, rows, dim, scale,
                       inv_dim, gin, msq);
}

// l2_normalize_entity_reprs: the gradient of every gathered document row, materialised as the reference does
// (objective.cu:354-412): g_j = coef_j · proj[j / R] is the gradient w.r.t. the normalised row (the sign of the
// negatives is already in coef_j), sent back through Normalizer::backward with the raw row E[id_j] as cached input.
// out[j] = (g_j·n² − e·(e·g_j)) / n³, msq[j] = mean_t(out[j]²) (the per-entry mean of squares Adagrad / Adam accumulate).
__global__ __launch_bounds__(256) void materialize_grad_entity_l2_kernel(const float* __restrict__ coef, const float* __restrict__ proj,
                                                                         const float* __restrict__ E, const int* __restrict__ ids,
                                                                         int64_t N, int R, int de, float inv_de,
                                                                         float* __restrict__ out, float* __restrict__ msq) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = blockIdx.x * 4 + (threadIdx.x >> 6); j < N; j += static_cast<int64_t>(gridDim.x) * 4) {
        const float* e = E + static_cast<size_t>(ids[j]) * de;
        const float* pr = proj + (j / R) * de;
        const float cf = coef[j];
        float nsq = 0.f, cr = 0.f;
        for (int t = lane; t < de; t += 64) { const float ev = e[t]; nsq += ev * ev; cr += ev * (pr[t] * cf); }
        nsq = wave_sum(nsq); cr = wave_sum(cr);
        const float n = sqrtf(nsq), n2 = n * n, n3 = n2 * n;
        float sq = 0.f;
        for (int t = lane; t < de; t += 64) {
            const float v = ((pr[t] * cf) * n2 - e[t] * cr) / n3;
            out[j * de + t] = v;
            sq += v * v;
        }
        sq = wave_sum(sq);
        if (lane == 0 && msq) msq[j] = sq * inv_de;
    }
}
void launch_materialize_grad_entity_l2(const float* coef, const float* proj, const float* E, const int* ids, int64_t N, int R,
                                       int de, float* out, float* msq, hipStream_t s) {
    if (N <= 0) return;
    const float inv_de = static_cast<float>(std::exp(-std::log(static_cast<double>(de))));
    hipLaunchKernelGGL(materialize_grad_entity_l2_kernel, dim3(stream_grid(N * 64, 256)), dim3(256), 0, s, coef, proj, E, ids, N, R,
                       de, inv_de, out, msq);
}

}  // namespace cunvsm
