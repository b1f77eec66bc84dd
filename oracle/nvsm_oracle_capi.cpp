// TEST INFRASTRUCTURE — NOT PRODUCT CODE. C ABI over nvsm_oracle.hpp for ctypes (tests/, smoke(),
// bench.py's cpu_baseline leg). See the header for the parity statement and reference citations.
//
// All array arguments are double at this boundary (converted to the model's dtype internally)
// except the *_native entry points used by the CPU-baseline timing loop.
#include "nvsm_oracle.hpp"

#include <map>
#include <memory>

using namespace nvsm_oracle;

extern "C" {

struct orc_config {
    int64_t num_words, num_entities;
    int32_t word_dim, entity_dim, window, num_random;
    int32_t batch_norm, nonlinearity, clip_sigmoid, bias_negative_samples, l2_phrase, l2_entity;
    int32_t update_method, adam_mode;
    double lambda, bn_epsilon, beta1, beta2, opt_epsilon;
};

}  // extern "C"

namespace {

Config to_config(const orc_config& c) {
    Config k;
    k.num_words = c.num_words; k.num_entities = c.num_entities;
    k.word_dim = c.word_dim; k.entity_dim = c.entity_dim; k.window = c.window; k.num_random = c.num_random;
    k.batch_norm = c.batch_norm; k.nonlinearity = c.nonlinearity; k.clip_sigmoid = c.clip_sigmoid;
    k.bias_negative_samples = c.bias_negative_samples; k.l2_phrase = c.l2_phrase; k.l2_entity = c.l2_entity;
    k.update_method = c.update_method; k.adam_mode = c.adam_mode;
    k.lambda = c.lambda; k.bn_epsilon = c.bn_epsilon; k.beta1 = c.beta1; k.beta2 = c.beta2; k.opt_epsilon = c.opt_epsilon;
    return k;
}

template <typename F>
std::vector<F> from_d(const double* p, size_t n) {
    std::vector<F> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = static_cast<F>(p[i]);
    return v;
}
template <typename F>
void to_d(const std::vector<F>& v, double* out) {
    for (size_t i = 0; i < v.size(); ++i) out[i] = static_cast<double>(v[i]);
}
template <typename F>
void to_d(const F* v, size_t n, double* out) {
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<double>(v[i]);
}

struct ModelBase {
    virtual ~ModelBase() {}
    virtual void initialize(RNG* rng) = 0;
    virtual std::map<std::string, std::pair<void*, size_t>> tensors() = 0;   // name → (ptr to F, count)
    virtual int dtype() const = 0;
    virtual void forward(const idx_t*, const double*, const idx_t*, const double*, size_t) = 0;
    virtual void forward_native(const idx_t*, const void*, const idx_t*, const void*, size_t) = 0;
    virtual double get_cost() = 0;
    virtual void backward() = 0;
    virtual void update(double lr, double scaled_lambda) = 0;
    virtual double scaled_lambda() = 0;
    virtual int gradcheck(const idx_t*, const double*, const idx_t*, const double*, size_t, double, double, double*, int*) = 0;
    virtual void set_allreduce(int (*fn)(double*, int64_t, void*), void* user, int world) = 0;
    virtual void set_exact_tables(int rank) = 0;
    virtual void set_owner_rows(int on) = 0;
};

template <typename F>
struct ModelImpl : ModelBase {
    Model<F> m;
    explicit ModelImpl(const Config& c) : m(c) {}
    int dtype() const override { return sizeof(F) == 8 ? 0 : 1; }
    void initialize(RNG* rng) override { m.initialize(rng); }
    std::map<std::string, std::pair<void*, size_t>> tensors() override {
        std::map<std::string, std::pair<void*, size_t>> t;
        auto add = [&](const char* name, std::vector<F>& v) { t[name] = std::make_pair(static_cast<void*>(v.data()), v.size()); };
        // parameter names follow cpp/params.cu:29-33 + storage.cu:118-120,246-249 (joined at model.cu:78-80)
        add("word_representations-representations", m.words.data);
        add("entity_representations-representations", m.entities.data);
        add("word_entity_mapping-transform", m.transform.transform);
        add("word_entity_mapping-bias", m.transform.bias);
        add("words.s0", m.words_upd.s0.data); add("words.s1", m.words_upd.s1.data);
        add("entities.s0", m.entities_upd.s0.data); add("entities.s1", m.entities_upd.s1.data);
        add("transform.s0.transform", m.transform_upd.s0.transform); add("transform.s0.bias", m.transform_upd.s0.bias);
        add("transform.s1.transform", m.transform_upd.s1.transform); add("transform.s1.bias", m.transform_upd.s1.bias);
        add("phrase", m.fwd.phrase); add("pre", m.fwd.pre); add("proj", m.fwd.proj);
        add("bn_mean", m.fwd.bn_mean); add("bn_inv_std", m.fwd.bn_inv_std);
        add("ent", m.fwd.ent); add("probs", m.fwd.probs); add("mass", m.fwd.mass); add("bweights", m.fwd.bweights);
        add("grad_entity", m.grads.grad_entity); add("grad_phrase", m.grads.grad_phrase);
        add("grad_transform", m.grads.grad_transform); add("grad_bias", m.grads.grad_bias);
        add("multipliers", m.grads.multipliers); add("grad_proj", m.grads.grad_proj);
        return t;
    }
    void forward(const idx_t* w, const double* ww, const idx_t* ids, const double* iw, size_t B) override {
        std::vector<F> ww_ = from_d<F>(ww, B * m.cfg.window), iw_ = from_d<F>(iw, B);
        m.forward(w, ww_.data(), ids, iw_.data(), B);
    }
    void forward_native(const idx_t* w, const void* ww, const idx_t* ids, const void* iw, size_t B) override {
        m.forward(w, static_cast<const F*>(ww), ids, static_cast<const F*>(iw), B);
    }
    double get_cost() override { return m.get_cost(); }
    void set_allreduce(int (*fn)(double*, int64_t, void*), void* user, int world_size) override {
        m.world = world_size > 1 ? world_size : 1;
        if (fn) m.allreduce = [fn, user](double* p, size_t n) { if (fn(p, static_cast<int64_t>(n), user) != 0) throw std::runtime_error("allreduce failed"); };
        else m.allreduce = nullptr;
    }
    void set_exact_tables(int rank) override { m.exact_rank = rank; }
    void set_owner_rows(int on) override { m.owner_rows_entities = on != 0; }
    void backward() override { m.backward(); }
    void update(double lr, double sl) override { m.update(static_cast<F>(lr), static_cast<F>(sl)); }
    double scaled_lambda() override { return static_cast<double>(m.scaled_regularization_lambda()); }

    // cpp/gradient_check.cu:5-140. Returns #failed; *max_rel = worst relative error among checked.
    int gradcheck(const idx_t* w, const double* ww, const idx_t* ids, const double* iw, size_t B,
                  double eps, double thresh, double* max_rel, int* num_checked) override {
        forward(w, ww, ids, iw, B);
        m.get_cost();
        m.backward();
        const Gradients<F> saved = m.grads;
        const size_t np = m.num_parameters();
        std::vector<double> predict(np);
        for (size_t i = 0; i < np; ++i) predict[i] = -static_cast<double>(m.parameter_gradient(i));
        int failed = 0, checked = 0;
        double worst = 0.0;
        for (size_t i = 0; i < np; ++i) {
            F* p = m.parameter_ptr(i);
            const F orig = *p;
            *p = orig + static_cast<F>(eps);
            forward(w, ww, ids, iw, B);
            const double cp = m.get_cost();
            *p = orig - static_cast<F>(eps);
            forward(w, ww, ids, iw, B);
            const double cm = m.get_cost();
            *p = orig;
            const double approx = (cp - cm) / (2.0 * eps);
            const double pred = predict[i];
            const double denom = std::max(std::abs(pred), std::abs(approx));
            const double rel = denom > 0 ? std::abs(pred - approx) / denom : 0.0;
            const double ratio = approx != 0.0 ? pred / approx : NAN;
            ++checked;
            if (pred * approx < 0.0) {
                ++failed;
            } else if (rel >= thresh) {
                if (!std::isnan(ratio)) ++failed;
            }
            if (!std::isnan(ratio)) worst = std::max(worst, rel);
        }
        m.grads = saved;
        forward(w, ww, ids, iw, B);
        m.get_cost();
        m.grads = saved;
        if (max_rel) *max_rel = worst;
        if (num_checked) *num_checked = checked;
        return failed;
    }
};

struct RepsBase {
    virtual ~RepsBase() {}
    virtual void fill(double v) = 0;
    virtual void set(const double*) = 0;
    virtual size_t get(int which, double* out) = 0;
    virtual void update(int ngroups, double** grads, const int64_t* num_grads, const idx_t** idx,
                        const int64_t* window, const double** weights, double lr, double lambda) = 0;
    virtual void update_dense_const(double g, double lr, double lambda) = 0;
};
template <typename F>
struct RepsImpl : RepsBase {
    RepresentationsStorage<F> st;
    RepresentationsUpdater<F> up;
    RepsImpl(size_t n, size_t dim, int method, int mode, double b1, double b2, double eps) : st(n, dim) {
        up.init(method, mode, n, dim, b1, b2, eps);
    }
    void fill(double v) override { std::fill(st.data.begin(), st.data.end(), static_cast<F>(v)); }
    void set(const double* p) override { st.data = from_d<F>(p, st.data.size()); }
    size_t get(int which, double* out) override {
        const std::vector<F>& v = which == 0 ? st.data : (which == 1 ? up.s0.data : up.s1.data);
        if (out) to_d(v, out);
        return v.size();
    }
    void update(int ngroups, double** grads, const int64_t* num_grads, const idx_t** idx,
                const int64_t* window, const double** weights, double lr, double lambda) override {
        std::vector<std::vector<F>> g(ngroups), w(ngroups);
        std::vector<SparseGrad<F>> descs;
        for (int i = 0; i < ngroups; ++i) {
            g[i] = from_d<F>(grads[i], num_grads[i] * st.dim);
            if (weights && weights[i]) w[i] = from_d<F>(weights[i], num_grads[i] * window[i]);
            descs.push_back({g[i].data(), static_cast<size_t>(num_grads[i]), st.dim, idx[i],
                             static_cast<size_t>(window[i]), (weights && weights[i]) ? w[i].data() : nullptr});
        }
        up.update(&st, &descs, static_cast<F>(lr), static_cast<F>(lambda));
        for (int i = 0; i < ngroups; ++i) to_d(g[i], grads[i]);
    }
    void update_dense_const(double g, double lr, double lambda) override {
        st.update_dense([&](size_t) { return static_cast<F>(g); }, static_cast<F>(lr), static_cast<F>(lambda));
    }
};

struct TrBase {
    virtual ~TrBase() {}
    virtual void fill(double v) = 0;
    virtual size_t get(int which, double* out) = 0;
    virtual void update(double* gt, double* gb, double lr, double lambda) = 0;
};
template <typename F>
struct TrImpl : TrBase {
    TransformStorage<F> st;
    TransformUpdater<F> up;
    TrImpl(size_t wd, size_t ed, int method, double b1, double b2, double eps) : st(wd, ed) { up.init(method, wd, ed, b1, b2, eps); }
    void fill(double v) override {
        std::fill(st.transform.begin(), st.transform.end(), static_cast<F>(v));
        std::fill(st.bias.begin(), st.bias.end(), static_cast<F>(v));
    }
    size_t get(int which, double* out) override {
        const std::vector<F>* v = nullptr;
        switch (which) {
            case 0: v = &st.transform; break; case 1: v = &st.bias; break;
            case 2: v = &up.s0.transform; break; case 3: v = &up.s0.bias; break;
            case 4: v = &up.s1.transform; break; default: v = &up.s1.bias; break;
        }
        if (out) to_d(*v, out);
        return v->size();
    }
    void update(double* gt, double* gb, double lr, double lambda) override {
        std::vector<F> a = from_d<F>(gt, st.transform.size()), b = from_d<F>(gb, st.bias.size());
        up.update(&st, a.data(), b.data(), static_cast<F>(lr), static_cast<F>(lambda));
        to_d(a, gt); to_d(b, gb);
    }
};

}  // namespace

extern "C" {

// ---- RNG (std::minstd_rand0, include/cuNVSM/base.h:36) ----
void* orc_rng_create(uint64_t seed) { return new RNG(static_cast<RNG::result_type>(seed)); }
void orc_rng_free(void* r) { delete static_cast<RNG*>(r); }
void orc_rng_seed(void* r, uint64_t seed) { static_cast<RNG*>(r)->seed(static_cast<RNG::result_type>(seed)); }
uint64_t orc_rng_get_state(void* r) { std::stringstream ss; ss << *static_cast<RNG*>(r); uint64_t s; ss >> s; return s; }
void orc_rng_set_state(void* r, uint64_t s) { std::stringstream ss; ss << s; ss >> *static_cast<RNG*>(r); }
void orc_generate_labels(void* r, const idx_t* labels, int64_t num_entities, int64_t n, int64_t k, idx_t* out) {
    generate_labels(labels, num_entities, n, k, static_cast<RNG*>(r), out);
}
void orc_glorot(void* r, int dtype, int64_t rows, int64_t cols, double* out) {
    if (dtype == 0) { init_matrix_glorot(out, rows, cols, static_cast<RNG*>(r)); }
    else { std::vector<float> t(rows * cols); init_matrix_glorot(t.data(), rows, cols, static_cast<RNG*>(r)); to_d(t, out); }
}

// ---- model ----
void* orc_model_create(const orc_config* c, int dtype) {
    try {
        if (dtype == 0) return static_cast<ModelBase*>(new ModelImpl<double>(to_config(*c)));
        return static_cast<ModelBase*>(new ModelImpl<float>(to_config(*c)));
    } catch (...) { return nullptr; }
}
void orc_model_free(void* h) { delete static_cast<ModelBase*>(h); }
void orc_model_initialize(void* h, void* rng) { static_cast<ModelBase*>(h)->initialize(static_cast<RNG*>(rng)); }
int64_t orc_model_tensor_size(void* h, const char* name) {
    auto t = static_cast<ModelBase*>(h)->tensors();
    auto it = t.find(name);
    return it == t.end() ? -1 : static_cast<int64_t>(it->second.second);
}
int orc_model_get(void* h, const char* name, double* out) {
    ModelBase* m = static_cast<ModelBase*>(h);
    auto t = m->tensors();
    auto it = t.find(name);
    if (it == t.end()) return -1;
    if (m->dtype() == 0) to_d(static_cast<double*>(it->second.first), it->second.second, out);
    else to_d(static_cast<float*>(it->second.first), it->second.second, out);
    return 0;
}
int orc_model_set(void* h, const char* name, const double* in) {
    ModelBase* m = static_cast<ModelBase*>(h);
    auto t = m->tensors();
    auto it = t.find(name);
    if (it == t.end()) return -1;
    const size_t n = it->second.second;
    if (m->dtype() == 0) { double* p = static_cast<double*>(it->second.first); for (size_t i = 0; i < n; ++i) p[i] = in[i]; }
    else { float* p = static_cast<float*>(it->second.first); for (size_t i = 0; i < n; ++i) p[i] = static_cast<float>(in[i]); }
    return 0;
}
// raw pointer to the tensor in the model's dtype (for bulk float32 transfer in the CPU baseline)
void* orc_model_tensor_ptr(void* h, const char* name) {
    auto t = static_cast<ModelBase*>(h)->tensors();
    auto it = t.find(name);
    return it == t.end() ? nullptr : it->second.first;
}
void orc_model_forward(void* h, const idx_t* words, const double* ww, const idx_t* ids, const double* iw, int64_t B) {
    static_cast<ModelBase*>(h)->forward(words, ww, ids, iw, B);
}
void orc_model_forward_native(void* h, const idx_t* words, const void* ww, const idx_t* ids, const void* iw, int64_t B) {
    static_cast<ModelBase*>(h)->forward_native(words, ww, ids, iw, B);
}
double orc_model_get_cost(void* h) { return static_cast<ModelBase*>(h)->get_cost(); }
void orc_model_backward(void* h) { static_cast<ModelBase*>(h)->backward(); }
int orc_model_update(void* h, double lr, double scaled_lambda) {
    try { static_cast<ModelBase*>(h)->update(lr, scaled_lambda); return 0; } catch (...) { return -1; }
}
void orc_model_set_allreduce(void* h, int (*fn)(double*, int64_t, void*), void* user, int world) {
    static_cast<ModelBase*>(h)->set_allreduce(fn, user, world);
}
void orc_model_set_exact_tables(void* h, int rank) { static_cast<ModelBase*>(h)->set_exact_tables(rank); }
void orc_model_set_owner_rows(void* h, int on) { static_cast<ModelBase*>(h)->set_owner_rows(on); }
double orc_model_scaled_lambda(void* h) { return static_cast<ModelBase*>(h)->scaled_lambda(); }
int orc_model_gradcheck(void* h, const idx_t* words, const double* ww, const idx_t* ids, const double* iw, int64_t B,
                        double eps, double thresh, double* max_rel, int* num_checked) {
    return static_cast<ModelBase*>(h)->gradcheck(words, ww, ids, iw, B, eps, thresh, max_rel, num_checked);
}
// Small problems on a many-core host: an OpenMP region over 256 hardware threads costs far more than the loop it splits
// (2.5 s instead of 5 ms per step for a 64-window batch on the 128-core GPU box). n <= 0 restores the default.
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    static const int default_threads = omp_get_max_threads();
    omp_set_num_threads(n > 0 ? (n < default_threads ? n : default_threads) : default_threads);
#else
    (void)n;
#endif
}
int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// ---- standalone embedding-table updater (cpp/updates_tests.cu, model_tests.cu:153-275) ----
void* orc_reps_create(int64_t n, int64_t dim, int method, int mode, double b1, double b2, double eps, int dtype) {
    if (dtype == 0) return static_cast<RepsBase*>(new RepsImpl<double>(n, dim, method, mode, b1, b2, eps));
    return static_cast<RepsBase*>(new RepsImpl<float>(n, dim, method, mode, b1, b2, eps));
}
void orc_reps_free(void* h) { delete static_cast<RepsBase*>(h); }
void orc_reps_fill(void* h, double v) { static_cast<RepsBase*>(h)->fill(v); }
void orc_reps_set(void* h, const double* p) { static_cast<RepsBase*>(h)->set(p); }
int64_t orc_reps_get(void* h, int which, double* out) { return static_cast<RepsBase*>(h)->get(which, out); }
int orc_reps_update(void* h, int ngroups, double** grads, const int64_t* num_grads, const idx_t** idx,
                    const int64_t* window, const double** weights, double lr, double lambda) {
    try { static_cast<RepsBase*>(h)->update(ngroups, grads, num_grads, idx, window, weights, lr, lambda); return 0; }
    catch (...) { return -1; }
}
void orc_reps_update_dense_const(void* h, double g, double lr, double lambda) {
    static_cast<RepsBase*>(h)->update_dense_const(g, lr, lambda);
}

// ---- standalone projection updater ----
void* orc_tr_create(int64_t word_dim, int64_t entity_dim, int method, double b1, double b2, double eps, int dtype) {
    if (dtype == 0) return static_cast<TrBase*>(new TrImpl<double>(word_dim, entity_dim, method, b1, b2, eps));
    return static_cast<TrBase*>(new TrImpl<float>(word_dim, entity_dim, method, b1, b2, eps));
}
void orc_tr_free(void* h) { delete static_cast<TrBase*>(h); }
void orc_tr_fill(void* h, double v) { static_cast<TrBase*>(h)->fill(v); }
int64_t orc_tr_get(void* h, int which, double* out) { return static_cast<TrBase*>(h)->get(which, out); }
void orc_tr_update(void* h, double* gt, double* gb, double lr, double lambda) { static_cast<TrBase*>(h)->update(gt, gb, lr, lambda); }

// ---- stateless pieces (fp64) ----
void orc_average_repr(const double* repr, int64_t dim, const idx_t* idx, const double* weights,
                      int64_t num_out, int64_t window, double* out) {
    average_repr(repr, dim, idx, weights, num_out, window, out);
}
void orc_bn_forward(const double* x, int64_t n, int64_t dim, const double* bias, double eps,
                    double* y, double* mean, double* inv_std) {
    bn_forward(x, n, dim, bias, eps, y, mean, inv_std);
}
void orc_bn_backward(const double* dy, const double* x, int64_t n, int64_t dim, const double* mean,
                     const double* inv_std, double* dx, double* grad_bias) {
    bn_backward(dy, x, n, dim, mean, inv_std, dx, grad_bias);
}
void orc_normalizer_forward(const double* x, int64_t n, int64_t dim, double* y, double* norms) {
    normalizer_forward(x, n, dim, y, norms);
}
void orc_normalizer_backward(const double* g, const double* x, const double* norms, int64_t n, int64_t dim, double* gin) {
    normalizer_backward(g, x, norms, n, dim, gin);
}
double orc_truncated_sigmoid(double x, double eps) { return truncated_sigmoid(x, eps); }
double orc_truncated_sigmoid_f32(double x, double eps) { return truncated_sigmoid(static_cast<float>(x), static_cast<float>(eps)); }
double orc_sigmoid_deriv(double p, double eps) { return sigmoid_to_log_sigmoid_deriv(p, eps); }
double orc_clip(double x, int dtype) { return dtype == 0 ? Clip<double>().fwd(x) : Clip<float>().fwd(static_cast<float>(x)); }
double orc_clip_deriv(double y, int dtype) {
    return dtype == 0 ? Clip<double>().deriv_from_output(y) : Clip<float>().deriv_from_output(static_cast<float>(y));
}

}  // extern "C"
