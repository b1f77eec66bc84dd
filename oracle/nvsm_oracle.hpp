// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU restatement of the cuNVSM TextEntity (LSE / NVSM) training hot path, used only as the
// parity checker for the HIP kernels (tests/, __graft_entry__.smoke(), bench.py's
// cpu_baseline leg). Nothing under cunvsm_amd/ may include, link or call this.
//
// Parity status: PINNED. The reference cannot be built here (CUDA + cuDNN + the un-vendored
// cvangysel/device_matrix@master), so this restatement is pinned against every fp64 golden
// vector the reference's own tests hold for the path (tests/golden/*.json, transcribed from
// cpp/model_tests.cu, cpp/updates_tests.cu, cpp/cudnn_utils_tests.cu, cpp/cuda_utils_tests.cu)
// — see tests/test_oracle_golden.py.
//
// Every function cites the reference file:line (relative to the cuNVSM tree) it restates.
// Memory layouts are the reference's raw buffers: device_matrix is column-major, so an
// embedding table (dim x n) is row-major [n][dim]; the projection is (entity_dim x word_dim)
// column-major, T[r + entity_dim * c].
#pragma once

#include <algorithm>
#include <cmath>
#include <functional>
#include <cstdint>
#include <cstring>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace nvsm_oracle {

typedef long idx_t;                       // include/cuNVSM/base.h:28  (typedef long int32)
typedef std::minstd_rand0 RNG;            // include/cuNVSM/base.h:36

enum Nonlinearity { TANH = 0, HARD_TANH = 1 };                  // proto/nvsm.proto:11-14
enum UpdateMethod { SGD = 0, ADAGRAD = 1, ADAM = 2 };           // proto/nvsm.proto:40-44
enum AdamMode { ADAM_NONE = 0, ADAM_SPARSE = 1, ADAM_DENSE_UPDATE = 2,
                ADAM_DENSE_UPDATE_DENSE_VARIANCE = 3 };         // proto/nvsm.proto:50-55

struct Config {
    int64_t num_words = 0, num_entities = 0;
    int word_dim = 0, entity_dim = 0;
    int window = 1;
    int num_random = 1;              // TrainConfig.num_random_entities
    int batch_norm = 0;
    int nonlinearity = TANH;
    int clip_sigmoid = 0;
    int bias_negative_samples = 0;
    int l2_phrase = 0, l2_entity = 0;
    double lambda = 0.0;             // TrainConfig.regularization_lambda
    int update_method = SGD;
    int adam_mode = ADAM_NONE;
    double bn_epsilon = 1e-4;        // cpp/objective.cu:114
    double beta1 = 0.9, beta2 = 0.999, opt_epsilon = 1e-6;   // include/cuNVSM/updates.h:21,183-185
};

// ---------------------------------------------------------------------------------------------
// Scalar functors — include/cuNVSM/cuda_utils.h
// ---------------------------------------------------------------------------------------------

// cuda_utils.h:192-214
template <typename F>
inline F truncated_sigmoid(const F x, const F eps) {
    const F prob = (x >= F(0)) ? F(1) / (F(1) + std::exp(-x))
                               : std::exp(x) / (F(1) + std::exp(x));
    const F hi = static_cast<F>(1.0 - static_cast<double>(eps));
    return std::min(std::max(prob, eps), hi);
}

// cuda_utils.h:217-235
template <typename F>
inline F sigmoid_to_log_sigmoid_deriv(const F p, const F eps) {
    return (static_cast<double>(p) >= (1.0 - static_cast<double>(eps)) || p <= eps) ? F(0) : F(1) - p;
}

// cuda_utils.h:86-147 — clip bounds widened by one ulp; derivative tests the OUTPUT.
template <typename F>
struct Clip {
    F min_, max_;
    Clip(F lo = F(-1), F hi = F(1), F eps = F(1e-5))
        : min_(std::nextafter(lo, lo - eps)), max_(std::nextafter(hi, hi + eps)) {}
    F fwd(F x) const { return std::min(std::max(x, min_), max_); }
    F deriv_from_output(F y) const { return (y > min_ && y < max_) ? F(1) : F(0); }
};

// ---------------------------------------------------------------------------------------------
// RNG consumers — include/cuNVSM/cuda_utils.h:24-56, cpp/labels.cu:4-22
// ---------------------------------------------------------------------------------------------

// cuda_utils.h:24-33: a fresh uniform_int_distribution<long>(0, max-1) per draw.
inline void generate_random_indexes(idx_t max, size_t num, RNG* rng, idx_t* out) {
    for (size_t i = 0; i < num; ++i) out[i] = std::uniform_int_distribution<idx_t>(0, max - 1)(*rng);
}

// labels.cu:4-22: [label, neg_1 .. neg_k] per instance, negatives uniform over ALL entities.
inline void generate_labels(const idx_t* labels, idx_t num_entities, size_t num_labels,
                            size_t num_negative, RNG* rng, idx_t* out) {
    const size_t R = num_negative + 1;
    for (size_t i = 0; i < num_labels; ++i) {
        out[i * R] = labels[i];
        generate_random_indexes(num_entities, num_negative, rng, out + i * R + 1);
    }
}

// cuda_utils.h:35-56: Glorot-uniform over the raw (column-major) buffer, in buffer order.
template <typename F>
inline void init_matrix_glorot(F* data, size_t rows, size_t cols, RNG* rng) {
    const F max = std::sqrt(6.0 / static_cast<double>(rows + cols));
    const int n = static_cast<int>(rows * cols);
    for (int i = 0; i < n; ++i) data[i] = 2 * max * (std::generate_canonical<F, 1>(*rng) - 0.5);
}

// ---------------------------------------------------------------------------------------------
// Batch normalisation — cpp/cudnn_utils.cu:82-183 (cuDNN PER_ACTIVATION training mode, γ≡1).
// x, y: [n][dim] row-major (= reference dim x n column-major).
// ---------------------------------------------------------------------------------------------
// Column sums over the batch in double, accumulated per thread over contiguous row blocks and merged in
// thread order (deterministic for a given thread count).
template <typename Fn>
inline void column_sums(size_t n, size_t dim, std::vector<double>* a, std::vector<double>* b2, Fn fn) {
#ifdef _OPENMP
    const int nt = (n * dim > (size_t(1) << 16)) ? omp_get_max_threads() : 1;
#else
    const int nt = 1;
#endif
    std::vector<std::vector<double>> la(nt, std::vector<double>(dim, 0.0)), lb(nt, std::vector<double>(dim, 0.0));
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        const size_t lo = n * tid / nt, hi = n * (tid + 1) / nt;
        double* pa = la[tid].data();
        double* pb = lb[tid].data();
        for (size_t r0 = lo; r0 < hi; ++r0) fn(r0, pa, pb);
    }
    a->assign(dim, 0.0); b2->assign(dim, 0.0);
    for (int t = 0; t < nt; ++t)
        for (size_t r = 0; r < dim; ++r) { (*a)[r] += la[t][r]; (*b2)[r] += lb[t][r]; }
}

typedef std::function<void(double*, size_t)> AllReduce;   // in-place sum across data-parallel ranks (tests only)

// n_total = number of rows the statistics span (= n, or the global batch under sync batch-norm).
template <typename F>
inline void bn_forward(const F* x, size_t n, size_t dim, const F* bias, F eps,
                       F* y, F* mean, F* inv_std, const AllReduce* ar = nullptr, size_t n_total = 0) {
    const double nn = static_cast<double>(n_total ? n_total : n);
    std::vector<double> s, s2, unused;
    column_sums(n, dim, &s, &unused, [&](size_t b, double* pa, double*) {
        for (size_t r = 0; r < dim; ++r) pa[r] += x[b * dim + r];
    });
    if (ar) (*ar)(s.data(), dim);
    for (size_t r = 0; r < dim; ++r) mean[r] = static_cast<F>(s[r] / nn);
    column_sums(n, dim, &s2, &unused, [&](size_t b, double* pa, double*) {
        for (size_t r = 0; r < dim; ++r) {
            const double d = static_cast<double>(x[b * dim + r]) - mean[r];
            pa[r] += d * d;
        }
    });
    if (ar) (*ar)(s2.data(), dim);
    for (size_t r = 0; r < dim; ++r)
        inv_std[r] = static_cast<F>(1.0 / std::sqrt(s2[r] / nn + static_cast<double>(eps)));   // biased variance
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < static_cast<int64_t>(n); ++b)
        for (size_t r = 0; r < dim; ++r)
            y[b * dim + r] = (x[b * dim + r] - mean[r]) * inv_std[r] + bias[r];
}

// cudnn_utils.cu:143-183: dβ = Σdy; dγ = Σ dy·x̂ (dropped by the caller); dx = invσ/N·(N·dy − dβ − x̂·dγ).
// dx may alias dy.
template <typename F>
inline void bn_backward(const F* dy, const F* x, size_t n, size_t dim, const F* mean,
                        const F* inv_std, F* dx, F* grad_bias, const AllReduce* ar = nullptr, size_t n_total = 0) {
    const double nn = static_cast<double>(n_total ? n_total : n);
    std::vector<double> dbeta, dgamma;
    column_sums(n, dim, &dbeta, &dgamma, [&](size_t b, double* pa, double* pb) {
        for (size_t r = 0; r < dim; ++r) {
            const double xhat = (static_cast<double>(x[b * dim + r]) - mean[r]) * inv_std[r];
            pa[r] += dy[b * dim + r];
            pb[r] += static_cast<double>(dy[b * dim + r]) * xhat;
        }
    });
    if (ar) { (*ar)(dbeta.data(), dim); (*ar)(dgamma.data(), dim); }
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < static_cast<int64_t>(n); ++b)
        for (size_t r = 0; r < dim; ++r) {
            const double xhat = (static_cast<double>(x[b * dim + r]) - mean[r]) * inv_std[r];
            dx[b * dim + r] = static_cast<F>(
                (static_cast<double>(inv_std[r]) / nn) * (nn * static_cast<double>(dy[b * dim + r]) - dbeta[r] - xhat * dgamma[r]));
        }
    for (size_t r = 0; r < dim; ++r) grad_bias[r] = static_cast<F>(dbeta[r]);
}

// ---------------------------------------------------------------------------------------------
// L2 row normaliser — cpp/cuda_utils.cu:12-130 (optional stage; off in both LSE and NVSM recipes).
// ---------------------------------------------------------------------------------------------
template <typename F>
inline void normalizer_forward(const F* x, size_t n, size_t dim, F* y, F* norms) {
    for (size_t b = 0; b < n; ++b) {
        F s = 0;
        for (size_t t = 0; t < dim; ++t) s += x[b * dim + t] * x[b * dim + t];
        norms[b] = std::sqrt(s);
        for (size_t t = 0; t < dim; ++t) y[b * dim + t] = x[b * dim + t] / norms[b];
    }
}

// grad_in = (g·‖x‖² − x·(x·g)) / ‖x‖³  — cuda_utils.cu:70-130
template <typename F>
inline void normalizer_backward(const F* g, const F* x_cache, const F* norms, size_t n, size_t dim, F* gin) {
    for (size_t b = 0; b < n; ++b) {
        F cross = 0;
        for (size_t t = 0; t < dim; ++t) cross += x_cache[b * dim + t] * g[b * dim + t];
        const F n2 = norms[b] * norms[b], n3 = std::pow(norms[b], F(3));
        for (size_t t = 0; t < dim; ++t)
            gin[b * dim + t] = (g[b * dim + t] * n2 - x_cache[b * dim + t] * cross) / n3;
    }
}

// ---------------------------------------------------------------------------------------------
// Gather-mean — cpp/params.cu:75-95 (average_repr_kernel). Divides by window even when weighted.
// ---------------------------------------------------------------------------------------------
template <typename F>
inline void average_repr(const F* repr, size_t dim, const idx_t* indices, const F* weights,
                         size_t num_out, size_t window, F* out) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < static_cast<int64_t>(num_out); ++b) {
        F* o = out + b * dim;
        for (size_t t = 0; t < dim; ++t) o[t] = 0;
        for (size_t w = 0; w < window; ++w) {
            const F* row = repr + static_cast<size_t>(indices[b * window + w]) * dim;
            const F wt = weights ? weights[b * window + w] : F(1);
            for (size_t t = 0; t < dim; ++t) o[t] += wt * row[t];
        }
        for (size_t t = 0; t < dim; ++t) o[t] = o[t] / static_cast<F>(window);
    }
}

// ---------------------------------------------------------------------------------------------
// Sparse gradient group — include/cuNVSM/storage.h SingleGradientType:
// (grad [num_grads][dim], indices [num_grads*window], window, weights-or-null).
// ---------------------------------------------------------------------------------------------
template <typename F>
struct SparseGrad {
    F* grad;                 // modified in place by Adagrad / Adam-SPARSE (as in the reference)
    size_t num_grads, dim;
    const idx_t* indices;
    size_t window;
    const F* weights;        // may be null
};

// Embedding-table storage + SGD application — cpp/storage.cu:37-102, include/cuNVSM/storage_inl.h.
template <typename F>
struct RepresentationsStorage {
    size_t n = 0, dim = 0;
    std::vector<F> data;     // [n][dim]

    RepresentationsStorage() {}
    RepresentationsStorage(size_t n_, size_t dim_) : n(n_), dim(dim_), data(n_ * dim_, F(0)) {}

    // storage.cu:51-102: optional dense decay, then scatter-add lr·wt·g (update_repr_kernel :37-49).
    void update(const std::vector<SparseGrad<F>>& descs, F lr, F scaled_lambda) {
        if (scaled_lambda > F(0)) {
            const F s = static_cast<F>(1.0 - static_cast<double>(scaled_lambda) * static_cast<double>(lr));
            F* d = data.data();
            const int64_t total = static_cast<int64_t>(data.size());
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < total; ++i) d[i] *= s;
        }
        for (const SparseGrad<F>& g : descs) {
            // Each thread owns the rows with (row % nthreads == tid) and walks ITS entries in batch order, so
            // every row sees its contributions in exactly the serial order (deterministic, race-free).
            const size_t total = g.num_grads * g.window;
#ifdef _OPENMP
            const size_t nth = (total * dim > (size_t(1) << 16)) ? static_cast<size_t>(omp_get_max_threads()) : 1;
#else
            const size_t nth = 1;
#endif
            std::vector<size_t> start(nth + 1, 0);
            std::vector<uint32_t> order(total);
            for (size_t e = 0; e < total; ++e) start[static_cast<size_t>(g.indices[e]) % nth + 1]++;
            for (size_t t = 0; t < nth; ++t) start[t + 1] += start[t];
            {
                std::vector<size_t> cur(start.begin(), start.end() - 1);
                for (size_t e = 0; e < total; ++e) order[cur[static_cast<size_t>(g.indices[e]) % nth]++] = static_cast<uint32_t>(e);
            }
#pragma omp parallel for schedule(static, 1) num_threads(nth)
            for (int64_t tid = 0; tid < static_cast<int64_t>(nth); ++tid) {
                for (size_t q = start[tid]; q < start[tid + 1]; ++q) {
                    const size_t e = order[q];
                    const F wt = g.weights ? g.weights[e] : F(1);
                    F* row = data.data() + static_cast<size_t>(g.indices[e]) * dim;
                    const F* src = g.grad + (e / g.window) * g.dim;
                    for (size_t t = 0; t < dim; ++t) row[t] += lr * wt * src[t];
                }
            }
        }
    }

    // storage_inl.h:4-32: θ = (1−λ·lr)·θ + lr·g   (g supplied per element)
    template <typename GradFn>
    void update_dense(GradFn grad_at, F lr, F scaled_lambda) {
        const F s = static_cast<F>(1.0 - static_cast<double>(scaled_lambda) * static_cast<double>(lr));
        F* d = data.data();
        const int64_t total = static_cast<int64_t>(data.size());
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < total; ++i) d[i] = d[i] * s + grad_at(static_cast<size_t>(i)) * lr;
    }
};

// Embedding-table optimisers — cpp/updates.cu:36-48, updates_adagrad.cu:72-179, updates_adam.cu:111-385.
template <typename F>
struct RepresentationsUpdater {
    int method = SGD, mode = ADAM_NONE;
    F beta1 = F(0.9), beta2 = F(0.999), epsilon = F(1e-6);
    uint64_t t = 1;                               // updates_adam.cu:130
    RepresentationsStorage<F> s0, s1;             // Adagrad: s0 = a (n x 1). Adam: s0 = m, s1 = v.

    void init(int method_, int mode_, size_t n, size_t dim, double b1, double b2, double eps) {
        method = method_; mode = mode_;
        beta1 = static_cast<F>(b1); beta2 = static_cast<F>(b2); epsilon = static_cast<F>(eps); t = 1;
        if (method == ADAGRAD) {
            s0 = RepresentationsStorage<F>(n, 1);                                  // updates_adagrad.cu:79-81
        } else if (method == ADAM) {
            s0 = RepresentationsStorage<F>(n, dim);                                // updates_adam.cu:122-124
            s1 = RepresentationsStorage<F>(n, mode < ADAM_DENSE_UPDATE_DENSE_VARIANCE ? 1 : dim);  // :125-127
        }
    }

    // mean over dims of g² per gradient column (reduce_axis<square> then scale by exp(-log(dim)))
    // updates_adagrad.cu:136-143, updates_adam.cu:232-240
    static void mean_squares(const SparseGrad<F>& g, std::vector<F>* out) {
        out->assign(g.num_grads, F(0));
        const F inv = static_cast<F>(std::exp(-std::log(static_cast<double>(g.dim))));
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < static_cast<int64_t>(g.num_grads); ++b) {
            F s = 0;
            for (size_t t = 0; t < g.dim; ++t) s += g.grad[b * g.dim + t] * g.grad[b * g.dim + t];
            (*out)[b] = s * inv;
        }
    }

    void update(RepresentationsStorage<F>* storage, std::vector<SparseGrad<F>>* descs, F lr, F scaled_lambda) {
        if (method == SGD) {                       // updates.cu:36-48
            storage->update(*descs, lr, scaled_lambda);
        } else if (method == ADAGRAD) {
            update_adagrad(storage, descs, lr, scaled_lambda);
        } else {
            update_adam(storage, descs, lr, scaled_lambda);
        }
    }

    // updates_adagrad.cu:99-179
    void update_adagrad(RepresentationsStorage<F>* storage, std::vector<SparseGrad<F>>* descs, F lr, F scaled_lambda) {
        if (descs->size() != 1) throw std::runtime_error("Adagrad currently does not implement multiple gradients.");
        SparseGrad<F>& g = descs->front();
        std::vector<F> avg;
        mean_squares(g, &avg);
        std::vector<SparseGrad<F>> avg_desc{{avg.data(), g.num_grads, 1, g.indices, g.window, g.weights}};
        s0.update(avg_desc, F(1), F(0));                                             // :153-158
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < static_cast<int64_t>(g.num_grads); ++b) {            // adagrad_update_kernel :83-97
            F agg = 0;
            for (size_t w = 0; w < g.window; ++w) agg += s0.data[static_cast<size_t>(g.indices[b * g.window + w])];
            agg /= static_cast<F>(g.window);
            const F d = std::sqrt(agg + epsilon);
            for (size_t t = 0; t < g.dim; ++t) g.grad[b * g.dim + t] /= d;
        }
        storage->update(*descs, lr, scaled_lambda);                                  // :177-178
    }

    // updates_adam.cu:153-385
    void update_adam(RepresentationsStorage<F>* storage, std::vector<SparseGrad<F>>* descs, F lr, F scaled_lambda) {
        const bool use_sgd_regularization = (mode < ADAM_DENSE_UPDATE_DENSE_VARIANCE);   // :162
        const F one_m_b1 = static_cast<F>(1.0 - static_cast<double>(beta1));
        const F one_m_b2 = static_cast<F>(1.0 - static_cast<double>(beta2));

        s0.update(*descs, one_m_b1, F(1));                                           // m_t  :196-200
        if (!use_sgd_regularization) {                                               // :203-213
            const F c = static_cast<F>((1.0 - static_cast<double>(beta1)) * static_cast<double>(scaled_lambda));
            const int64_t tot = static_cast<int64_t>(s0.data.size());
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < tot; ++i) s0.data[i] += (-c) * storage->data[i];
        }
        if (mode < ADAM_DENSE_UPDATE_DENSE_VARIANCE) {                               // v_t  :216-252
            std::vector<std::vector<F>> keep(descs->size());
            std::vector<SparseGrad<F>> sq;
            for (size_t i = 0; i < descs->size(); ++i) {
                const SparseGrad<F>& g = (*descs)[i];
                mean_squares(g, &keep[i]);
                sq.push_back({keep[i].data(), g.num_grads, 1, g.indices, g.window, g.weights});
            }
            s1.update(sq, one_m_b2, F(1));
        } else {                                                                     // :253-282
            RepresentationsStorage<F> agg(s1.n, s1.dim);
            agg.update(*descs, F(1), F(0));
            const int64_t tot = static_cast<int64_t>(agg.data.size());
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < tot; ++i) {
                agg.data[i] += (-scaled_lambda) * storage->data[i];
                agg.data[i] = agg.data[i] * agg.data[i];
            }
            s1.update_dense([&](size_t i) { return agg.data[i]; }, one_m_b2, F(1));
        }

        const F bc = static_cast<F>(std::sqrt(1.0 - std::pow(static_cast<double>(beta2), static_cast<double>(t))) /
                                    (1.0 - std::pow(static_cast<double>(beta1), static_cast<double>(t))));   // :285
        t += 1;

        if (mode == ADAM_DENSE_UPDATE) {                                             // :293-311
            const size_t dim = s0.dim;
            storage->update_dense(
                [&](size_t i) { return (s0.data[i] / (std::sqrt(s1.data[i / dim]) + epsilon)) * bc; }, lr, scaled_lambda);
        } else if (mode == ADAM_DENSE_UPDATE_DENSE_VARIANCE) {                       // :312-328
            storage->update_dense(
                [&](size_t i) { return (s0.data[i] / (std::sqrt(s1.data[i]) + epsilon)) * bc; }, lr, F(0));
        } else {                                                                     // SPARSE  :332-384
            if (descs->size() != 1) throw std::runtime_error("Sparse Adam currently does not implement multiple gradients.");
            SparseGrad<F>& g = descs->front();
            const size_t dim = g.dim;
#pragma omp parallel for schedule(static)
            for (int64_t b = 0; b < static_cast<int64_t>(g.num_grads); ++b) {        // adam_sparse_update_kernel :132-151
                F agg_v = 0;
                for (size_t w = 0; w < g.window; ++w) agg_v += s1.data[static_cast<size_t>(g.indices[b * g.window + w])];
                agg_v /= static_cast<F>(g.window);
                const F denom = std::sqrt(agg_v) + epsilon;
                for (size_t tt = 0; tt < dim; ++tt) {
                    F agg_m = 0;
                    for (size_t w = 0; w < g.window; ++w)
                        agg_m += s0.data[static_cast<size_t>(g.indices[b * g.window + w]) * dim + tt];
                    agg_m /= static_cast<F>(g.window);
                    g.grad[b * dim + tt] = bc * agg_m / denom;
                }
            }
            storage->update(*descs, lr, use_sgd_regularization ? scaled_lambda : F(0));
        }
    }
};

// Projection storage + optimisers — cpp/storage.cu:185-228, updates.cu:24-34,
// updates_adagrad.cu:33-70, updates_adam.cu:46-105, include/cuNVSM/updates.h:23-62.
template <typename F>
struct TransformStorage {
    std::vector<F> transform, bias;
    TransformStorage() {}
    TransformStorage(size_t word_dim, size_t entity_dim) : transform(word_dim * entity_dim, F(0)), bias(entity_dim, F(0)) {}

    static void update_dense(std::vector<F>* p, const F* g, F lr, F lambda, bool square) {
        const F s = static_cast<F>(1.0 - static_cast<double>(lambda) * static_cast<double>(lr));
        for (size_t i = 0; i < p->size(); ++i) {
            const F gi = square ? g[i] * g[i] : g[i];
            (*p)[i] = (*p)[i] * s + gi * lr;
        }
    }
    // storage.cu:198-228 — the bias slot hard-codes λ = 0.
    void update(const F* g_transform, const F* g_bias, F lr, F scaled_lambda, bool square = false) {
        update_dense(&transform, g_transform, lr, scaled_lambda, square);
        update_dense(&bias, g_bias, lr, F(0), square);
    }
};

template <typename F>
struct TransformUpdater {
    int method = SGD;
    F beta1 = F(0.9), beta2 = F(0.999), epsilon = F(1e-6);
    uint64_t t = 1;
    TransformStorage<F> s0, s1;

    void init(int method_, size_t word_dim, size_t entity_dim, double b1, double b2, double eps) {
        method = method_; beta1 = static_cast<F>(b1); beta2 = static_cast<F>(b2); epsilon = static_cast<F>(eps); t = 1;
        if (method == ADAGRAD) s0 = TransformStorage<F>(word_dim, entity_dim);
        if (method == ADAM) { s0 = TransformStorage<F>(word_dim, entity_dim); s1 = TransformStorage<F>(word_dim, entity_dim); }
    }

    // g_transform / g_bias are modified in place (as in the reference).
    void update(TransformStorage<F>* storage, F* g_transform, F* g_bias, F lr, F scaled_lambda) {
        const size_t nt = storage->transform.size(), nb = storage->bias.size();
        if (method == SGD) {                                                         // updates.cu:24-34
            storage->update(g_transform, g_bias, lr, scaled_lambda);
        } else if (method == ADAGRAD) {                                              // updates_adagrad.cu:33-70
            s0.update(g_transform, g_bias, F(1), F(0), /*square=*/true);
            for (size_t i = 0; i < nt; ++i) g_transform[i] = g_transform[i] / std::sqrt(s0.transform[i] + epsilon);
            for (size_t i = 0; i < nb; ++i) g_bias[i] = g_bias[i] / std::sqrt(s0.bias[i] + epsilon);
            storage->update(g_transform, g_bias, lr, scaled_lambda);
        } else {                                                                     // updates_adam.cu:46-105
            for (size_t i = 0; i < nt; ++i) g_transform[i] += (-scaled_lambda) * storage->transform[i];   // updates.h:23-62 (T only)
            s0.update(g_transform, g_bias, static_cast<F>(1.0 - static_cast<double>(beta1)), F(1));       // bias moments never decay
            s1.update(g_transform, g_bias, static_cast<F>(1.0 - static_cast<double>(beta2)), F(1), true);
            const F bc = static_cast<F>(std::sqrt(1.0 - std::pow(static_cast<double>(beta2), static_cast<double>(t))) /
                                        (1.0 - std::pow(static_cast<double>(beta1), static_cast<double>(t))));
            for (size_t i = 0; i < nt; ++i) g_transform[i] = (s0.transform[i] * bc) / (std::sqrt(s1.transform[i]) + epsilon);
            for (size_t i = 0; i < nb; ++i) g_bias[i] = (s0.bias[i] * bc) / (std::sqrt(s1.bias[i]) + epsilon);
            t += 1;
            storage->update(g_transform, g_bias, lr, F(0));
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Forward / backward intermediates — include/cuNVSM/intermediate_results.h:232-307
// ---------------------------------------------------------------------------------------------
template <typename F>
struct ForwardResult {
    size_t B = 0, window = 0, R = 0;
    std::vector<idx_t> words, entity_ids;
    std::vector<F> word_weights;
    std::vector<F> phrase, phrase_raw, phrase_norms;      // [B][dw]  (raw = before optional l2 norm)
    std::vector<F> pre, proj;                             // [B][de]  pre = T·x (+b); proj = act(BN(pre))
    std::vector<F> bn_mean, bn_inv_std;
    std::vector<F> ent, ent_raw, ent_norms;               // [N][de]  signed (negated for negatives)
    std::vector<F> probs, mass, bweights;                 // [N]
    double cost = NAN;
};

template <typename F>
struct Gradients {
    std::vector<F> grad_entity;       // [N][de]
    std::vector<F> grad_phrase;       // [B][dw]
    std::vector<F> grad_transform;    // de x dw column-major
    std::vector<F> grad_bias;         // [de]
    std::vector<F> multipliers;       // [N]
    std::vector<F> grad_proj;         // [B][de]  d cost / d (T·x + b) after nonlinearity' (and BN bwd)
};

// ---------------------------------------------------------------------------------------------
// Model — include/cuNVSM/model.h:75-131, cpp/model.cu; objective cpp/objective.cu:30-481.
// ---------------------------------------------------------------------------------------------
template <typename F>
struct Model {
    Config cfg;
    RepresentationsStorage<F> words, entities;
    TransformStorage<F> transform;
    RepresentationsUpdater<F> words_upd, entities_upd;
    TransformUpdater<F> transform_upd;
    ForwardResult<F> fwd;
    Gradients<F> grads;
    // data-parallel test hooks: world size and an in-place cross-rank sum
    size_t world = 1;
    AllReduce allreduce;
    // exact data-parallel tables (test hook): every rank applies the sparse gradients of ALL ranks' windows, in rank order —
    // the update of the single process on the global batch. -1 = off (tables are updated from the rank's own windows).
    int exact_rank = -1;
    // owner-partitioned documents table (test hook, with exact_rank >= 0; SURVEY.md §8e last bullet): row r of E
    // and of its optimiser state belongs to rank r mod world. A rank applies, of the GLOBAL batch's sparse gradients, only the
    // entries of its own rows — an eighth of the update's work at eight ranks — and its dense decay is only valid for those
    // rows: the caller then replaces every rank's rows by their owners' (owned_row_exchange below is that all-gather, written
    // as a sum with one non-zero term per row), which leaves all replicas equal to the single process's table. The optimiser
    // state is never exchanged: a row's state is read by its own update only (the documents table has window 1). Not the
    // words table: with a window the reference's Adagrad / sparse-Adam direction of a row averages the state of the OTHER
    // words of the window (cpp/updates_adagrad.cu:83-97, updates_adam.cu:132-151), which belong to other ranks.
    bool owner_rows_entities = false;
    void owned_row_exchange() {
        if (!(owner_rows_entities && world > 1 && allreduce && exact_rank >= 0)) return;
        const size_t de = static_cast<size_t>(cfg.entity_dim);
        std::vector<double> t(entities.data.size(), 0.0);
        for (size_t r = 0; r < entities.n; ++r)
            if (static_cast<int>(r % world) == exact_rank)
                for (size_t k = 0; k < de; ++k) t[r * de + k] = static_cast<double>(entities.data[r * de + k]);
        allreduce(t.data(), t.size());
        for (size_t i = 0; i < t.size(); ++i) entities.data[i] = static_cast<F>(t[i]);
    }

    // all-gather through the cross-rank sum: every rank contributes its slice at its own offset of a zeroed buffer
    // (float, double and ids below 2^53 pass through a double unchanged)
    template <typename T>
    std::vector<T> all_gather(const T* p, size_t n) const {
        std::vector<double> t(n * world, 0.0);
        for (size_t i = 0; i < n; ++i) t[static_cast<size_t>(exact_rank) * n + i] = static_cast<double>(p[i]);
        allreduce(t.data(), t.size());
        std::vector<T> out(t.size());
        for (size_t i = 0; i < t.size(); ++i) out[i] = static_cast<T>(t[i]);
        return out;
    }

    explicit Model(const Config& c) : cfg(c),
        words(c.num_words, c.word_dim), entities(c.num_entities, c.entity_dim),
        transform(c.word_dim, c.entity_dim) {
        words_upd.init(c.update_method, c.adam_mode, c.num_words, c.word_dim, c.beta1, c.beta2, c.opt_epsilon);
        entities_upd.init(c.update_method, c.adam_mode, c.num_entities, c.entity_dim, c.beta1, c.beta2, c.opt_epsilon);
        transform_upd.init(c.update_method, c.word_dim, c.entity_dim, c.beta1, c.beta2, c.opt_epsilon);
    }

    // model.cu:37-43: words → entities → transform; bias = 0 (params.cu:361-372).
    void initialize(RNG* rng) {
        init_matrix_glorot(words.data.data(), cfg.word_dim, cfg.num_words, rng);
        init_matrix_glorot(entities.data.data(), cfg.entity_dim, cfg.num_entities, rng);
        init_matrix_glorot(transform.transform.data(), cfg.entity_dim, cfg.word_dim, rng);
        std::fill(transform.bias.begin(), transform.bias.end(), F(0));
    }

    // params.cu:377-451 — Transform::transform. x [B][dw] → pre, out [B][de].
    void transform_forward(const F* x, size_t B, bool bn, F bn_eps, F* pre, F* out, F* mean, F* inv_std) const {
        const size_t dw = cfg.word_dim, de = cfg.entity_dim;
        const F* T = transform.transform.data();
        const F* bias = transform.bias.data();
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < static_cast<int64_t>(B); ++b) {
            F* p = pre + b * de;
            for (size_t r = 0; r < de; ++r) p[r] = bn ? F(0) : bias[r];          // dst_contains_bias (:396-405,421)
            for (size_t c = 0; c < dw; ++c) {
                const F xc = x[b * dw + c];
                const F* col = T + c * de;
                for (size_t r = 0; r < de; ++r) p[r] += col[r] * xc;
            }
        }
        if (bn) bn_forward(pre, B, de, bias, bn_eps, out, mean, inv_std,          // :425-428
                           (world > 1 && allreduce) ? &allreduce : nullptr, B * world);
        else std::memcpy(out, pre, sizeof(F) * B * de);
        const Clip<F> clip;
        const int64_t total = static_cast<int64_t>(B * de);
        if (cfg.nonlinearity == TANH) {
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < total; ++i) out[i] = std::tanh(out[i]);       // :431-436
        } else {
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < total; ++i) out[i] = clip.fwd(out[i]);        // :437-443
        }
    }

    // objective.cu:30-313 — compute_cost. entity_ids [B*R] come from generate_labels (F2).
    void forward(const idx_t* w_idx, const F* w_wt, const idx_t* entity_ids, const F* inst_w, size_t B) {
        const size_t dw = cfg.word_dim, de = cfg.entity_dim, win = cfg.window;
        const size_t k = cfg.num_random, R = k + 1, N = B * R;
        ForwardResult<F>& f = fwd;          // buffers are reused across steps (no 2 GB of fresh pages per step)
        f.B = B; f.window = win; f.R = R;
        f.phrase_raw.clear(); f.phrase_norms.clear(); f.ent_raw.clear(); f.ent_norms.clear();
        f.words.assign(w_idx, w_idx + B * win);
        f.word_weights.assign(w_wt, w_wt + B * win);
        f.entity_ids.assign(entity_ids, entity_ids + N);

        f.phrase.resize(B * dw);
        average_repr(words.data.data(), dw, f.words.data(), f.word_weights.data(), B, win, f.phrase.data());   // :126-130
        if (cfg.l2_phrase) {                                                                                   // :136-142
            f.phrase_raw = f.phrase; f.phrase_norms.resize(B);
            normalizer_forward(f.phrase_raw.data(), B, dw, f.phrase.data(), f.phrase_norms.data());
        }
        f.pre.resize(B * de); f.proj.resize(B * de); f.bn_mean.resize(de); f.bn_inv_std.resize(de);
        transform_forward(f.phrase.data(), B, cfg.batch_norm, static_cast<F>(cfg.bn_epsilon),
                          f.pre.data(), f.proj.data(), f.bn_mean.data(), f.bn_inv_std.data());               // :145-148

        f.ent.resize(N * de);
        average_repr(entities.data.data(), de, f.entity_ids.data(), static_cast<const F*>(nullptr), N, 1, f.ent.data());   // :164-166
        if (cfg.l2_entity) {                                                                                   // :168-174
            f.ent_raw = f.ent; f.ent_norms.resize(N);
            normalizer_forward(f.ent_raw.data(), N, de, f.ent.data(), f.ent_norms.data());
        }
        const F sig_eps = cfg.clip_sigmoid ? F(1e-7) : F(0);                                                  // :245-246
        f.probs.resize(N); f.mass.resize(N); f.bweights.resize(N);
        // instance-weight fix-up — :268-290
        const bool rebalance = (!cfg.bias_negative_samples && k > 1);
        const F neg_scale = static_cast<F>((static_cast<double>(static_cast<F>(k)) + 1.0) / (2.0 * static_cast<double>(static_cast<F>(k))));
#pragma omp parallel for schedule(static)
        for (int64_t j = 0; j < static_cast<int64_t>(N); ++j) {
            const size_t b = j / R;
            const bool positive = (j % R == 0);
            F* e = f.ent.data() + j * de;
            if (!positive) for (size_t t = 0; t < de; ++t) e[t] = -e[t];                                      // :184-187
            const F* p = f.proj.data() + b * de;
            F s = 0;
            for (size_t t = 0; t < de; ++t) s += p[t] * e[t];                                                 // :196-239
            f.probs[j] = truncated_sigmoid(s, sig_eps);                                                       // :242-246
            F w = inst_w[b];
            if (rebalance) { w = w * neg_scale; if (positive) w = w * static_cast<F>(k); }
            f.bweights[j] = w;
            f.mass[j] = std::log(f.probs[j]) * w;                                                             // :250-305
        }
        f.cost = NAN;
    }

    // intermediate_results.cu:80-124
    double get_cost() {
        if (std::isnan(fwd.cost)) {
            F s = 0;
            for (size_t j = 0; j < fwd.mass.size(); ++j) s += fwd.mass[j];
            F log_data_prob = s;
            if (world > 1 && allreduce) { double d = static_cast<double>(s); allreduce(&d, 1); log_data_prob = static_cast<F>(d); }
            log_data_prob /= static_cast<F>(fwd.B * world);
            fwd.cost = -static_cast<double>(log_data_prob);
        }
        return fwd.cost;
    }

    // intermediate_results.cu:126-129
    F scaled_regularization_lambda() const { return static_cast<F>(cfg.lambda) / static_cast<F>(fwd.B * world); }

    // objective.cu:315-481 — compute_gradients; params.cu:453-535 — Transform::backward.
    void backward() {
        const size_t dw = cfg.word_dim, de = cfg.entity_dim;
        const ForwardResult<F>& f = fwd;
        const size_t B = f.B, R = f.R, N = B * R;
        Gradients<F>& g = grads;
        g.multipliers.resize(N); g.grad_entity.resize(N * de); g.grad_proj.assign(B * de, F(0));
        const F bsn = static_cast<F>(std::exp(-std::log(static_cast<double>(B * world))));                    // :354 (global batch)
        const AllReduce* ar = (world > 1 && allreduce) ? &allreduce : nullptr;
        const F d_eps = cfg.clip_sigmoid ? F(1e-6) : F(0);                                                    // :367-368
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < static_cast<int64_t>(B); ++b) {
            F* gp = g.grad_proj.data() + b * de;
            const F* p = f.proj.data() + b * de;
            for (size_t r = 0; r < R; ++r) {
                const size_t j = b * R + r;
                const F m = f.bweights[j] * (sigmoid_to_log_sigmoid_deriv(f.probs[j], d_eps) * bsn);         // :357-371
                g.multipliers[j] = m;
                F* ge = g.grad_entity.data() + j * de;
                const F* e = f.ent.data() + j * de;
                for (size_t t = 0; t < de; ++t) {
                    F v = p[t] * m;                                                                           // :381-395
                    if (r != 0) v = -v;                                                                       // :398-401
                    ge[t] = v;
                    gp[t] += m * e[t];                                                                        // fold_columns :420-425
                }
            }
        }
        if (cfg.l2_entity) {                                                                                  // :405-412
            std::vector<F> tmp(N * de);
            // the cached input of the entity normaliser is the un-negated gathered rows
            normalizer_backward(g.grad_entity.data(), f.ent_raw.data(), f.ent_norms.data(), N, de, tmp.data());
            g.grad_entity.swap(tmp);
        }
        // Transform::backward — params.cu:473-491 nonlinearity' on the OUTPUT
        const Clip<F> clip;
        const int64_t total = static_cast<int64_t>(B * de);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < total; ++i) {
            const F y = f.proj[i];
            const F d = (cfg.nonlinearity == TANH) ? static_cast<F>(1.0 - static_cast<double>(y * y))   // cuda_utils.h:74-82
                                                   : clip.deriv_from_output(y);
            g.grad_proj[i] = d * g.grad_proj[i];
        }
        g.grad_bias.assign(de, F(0));
        if (!cfg.batch_norm) {                                                                                // :509-514
            for (size_t b = 0; b < B; ++b)
                for (size_t r = 0; r < de; ++r) g.grad_bias[r] += g.grad_proj[b * de + r];
            if (ar) {
                std::vector<double> t(g.grad_bias.begin(), g.grad_bias.end());
                (*ar)(t.data(), t.size());
                for (size_t r = 0; r < de; ++r) g.grad_bias[r] = static_cast<F>(t[r]);
            }
        } else {                                                                                              // :515-521
            bn_backward(g.grad_proj.data(), f.pre.data(), B, de, f.bn_mean.data(), f.bn_inv_std.data(),
                        g.grad_proj.data(), g.grad_bias.data(), ar, B * world);
        }
        // ∂T = gproj · phraseᵀ — params.cu:526-531
        g.grad_transform.assign(de * dw, F(0));
        {
#ifdef _OPENMP
            const int nt = omp_get_max_threads();
#else
            const int nt = 1;
#endif
            std::vector<std::vector<F>> local(nt, std::vector<F>(de * dw, F(0)));
#pragma omp parallel
            {
#ifdef _OPENMP
                std::vector<F>& acc = local[omp_get_thread_num()];
#else
                std::vector<F>& acc = local[0];
#endif
#pragma omp for schedule(static)
                for (int64_t b = 0; b < static_cast<int64_t>(B); ++b) {
                    const F* gp = g.grad_proj.data() + b * de;
                    const F* x = f.phrase.data() + b * dw;
                    for (size_t c = 0; c < dw; ++c) {
                        const F xc = x[c];
                        F* col = acc.data() + c * de;
                        for (size_t r = 0; r < de; ++r) col[r] += gp[r] * xc;
                    }
                }
            }
            for (int t = 0; t < nt; ++t)
                for (size_t i = 0; i < de * dw; ++i) g.grad_transform[i] += local[t][i];
            if (ar) {                                  // the one dense all-reduce of the data-parallel step
                std::vector<double> t(g.grad_transform.begin(), g.grad_transform.end());
                (*ar)(t.data(), t.size());
                for (size_t i = 0; i < de * dw; ++i) g.grad_transform[i] = static_cast<F>(t[i]);
            }
        }
        // gphrase = Tᵀ·gproj — objective.cu:447-456
        g.grad_phrase.assign(B * dw, F(0));
        const F* T = transform.transform.data();
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < static_cast<int64_t>(B); ++b) {
            const F* gp = g.grad_proj.data() + b * de;
            F* gx = g.grad_phrase.data() + b * dw;
            for (size_t c = 0; c < dw; ++c) {
                const F* col = T + c * de;
                F s = 0;
                for (size_t r = 0; r < de; ++r) s += col[r] * gp[r];
                gx[c] = s;
            }
        }
        if (cfg.l2_phrase) {                                                                                  // :461-468
            std::vector<F> tmp(B * dw);
            normalizer_backward(g.grad_phrase.data(), f.phrase_raw.data(), f.phrase_norms.data(), B, dw, tmp.data());
            g.grad_phrase.swap(tmp);
        }
        const F inv_w = static_cast<F>(std::exp(-std::log(static_cast<double>(f.window))));                   // :471-476
        {
            F* gp = g.grad_phrase.data();
            const int64_t tot = static_cast<int64_t>(g.grad_phrase.size());
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < tot; ++i) gp[i] = gp[i] * inv_w;
        }
    }

    // model.cu:187-220: entities → words → transform. Gradients are consumed (modified in place).
    void update(F lr, F scaled_lambda) {
        const ForwardResult<F>& f = fwd;
        const size_t N = f.B * f.R;
        if (world > 1 && allreduce && exact_rank >= 0) {
            // the global batch's sparse gradients on every rank (the multipliers already carry 1 / B_global)
            std::vector<F> ge_all = all_gather(grads.grad_entity.data(), grads.grad_entity.size());
            std::vector<idx_t> ids_all = all_gather(f.entity_ids.data(), N);
            std::vector<F> gp_all = all_gather(grads.grad_phrase.data(), grads.grad_phrase.size());
            std::vector<idx_t> words_all = all_gather(f.words.data(), f.words.size());
            std::vector<F> ww_all = all_gather(f.word_weights.data(), f.word_weights.size());
            if (owner_rows_entities) {
                // only the entries of this rank's rows, in the global batch's order (what a rank of the partitioned engine
                // would keep after filtering the gathered ids)
                const size_t de = static_cast<size_t>(cfg.entity_dim);
                std::vector<F> ge_own; std::vector<idx_t> ids_own;
                for (size_t j = 0; j < ids_all.size(); ++j)
                    if (static_cast<int>(static_cast<size_t>(ids_all[j]) % world) == exact_rank) {
                        ids_own.push_back(ids_all[j]);
                        ge_own.insert(ge_own.end(), ge_all.begin() + j * de, ge_all.begin() + (j + 1) * de);
                    }
                std::vector<SparseGrad<F>> ge{{ge_own.data(), ids_own.size(), de, ids_own.data(), 1, static_cast<const F*>(nullptr)}};
                entities_upd.update(&entities, &ge, lr, scaled_lambda);
                owned_row_exchange();
            } else {
            std::vector<SparseGrad<F>> ge{{ge_all.data(), N * world, static_cast<size_t>(cfg.entity_dim), ids_all.data(), 1,
                                           static_cast<const F*>(nullptr)}};
            entities_upd.update(&entities, &ge, lr, scaled_lambda);
            }
            std::vector<SparseGrad<F>> gw{{gp_all.data(), f.B * world, static_cast<size_t>(cfg.word_dim), words_all.data(), f.window,
                                           ww_all.data()}};
            words_upd.update(&words, &gw, lr, scaled_lambda);
            transform_upd.update(&transform, grads.grad_transform.data(), grads.grad_bias.data(), lr, scaled_lambda);
            return;
        }
        std::vector<SparseGrad<F>> ge{{grads.grad_entity.data(), N, static_cast<size_t>(cfg.entity_dim),
                                       f.entity_ids.data(), 1, static_cast<const F*>(nullptr)}};             // intermediate_results.cu:300-308
        entities_upd.update(&entities, &ge, lr, scaled_lambda);
        std::vector<SparseGrad<F>> gw{{grads.grad_phrase.data(), f.B, static_cast<size_t>(cfg.word_dim),
                                       f.words.data(), f.window, f.word_weights.data()}};                    // :286-296
        words_upd.update(&words, &gw, lr, scaled_lambda);
        transform_upd.update(&transform, grads.grad_transform.data(), grads.grad_bias.data(), lr, scaled_lambda);
    }

    // --- gradient checker: cpp/gradient_check.cu:5-140 + storage.cu:133-183,264-283 ---------------
    size_t num_parameters() const {
        return words.data.size() + entities.data.size() + transform.transform.size() + transform.bias.size();
    }
    F* parameter_ptr(size_t i) {
        if (i < words.data.size()) return &words.data[i];
        i -= words.data.size();
        if (i < entities.data.size()) return &entities.data[i];
        i -= entities.data.size();
        if (i < transform.transform.size()) return &transform.transform[i];
        i -= transform.transform.size();
        return &transform.bias[i];
    }
    // Dense gradient of parameter i assembled from the sparse descriptors (storage.cu:133-183).
    F parameter_gradient(size_t i) const {
        const ForwardResult<F>& f = fwd;
        if (i < words.data.size()) {
            const size_t row = i / cfg.word_dim, t = i % cfg.word_dim;
            F s = 0;
            for (size_t gi = 0; gi < f.words.size(); ++gi)
                if (static_cast<size_t>(f.words[gi]) == row)
                    s += f.word_weights[gi] * grads.grad_phrase[(gi / f.window) * cfg.word_dim + t];
            return s;
        }
        i -= words.data.size();
        if (i < entities.data.size()) {
            const size_t row = i / cfg.entity_dim, t = i % cfg.entity_dim;
            F s = 0;
            for (size_t gi = 0; gi < f.entity_ids.size(); ++gi)
                if (static_cast<size_t>(f.entity_ids[gi]) == row) s += grads.grad_entity[gi * cfg.entity_dim + t];
            return s;
        }
        i -= entities.data.size();
        if (i < transform.transform.size()) return grads.grad_transform[i];
        i -= transform.transform.size();
        return grads.grad_bias[i];
    }
};

}  // namespace nvsm_oracle
