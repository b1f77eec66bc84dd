// C++ host-side wrapper over the C ABI (include/cunvsm_amd.h) with the shape of cuNVSM's
// Model<TextEntity::Objective> (include/cuNVSM/model.h:75-131): what a maintainer of the reference would
// instantiate in cpp/main.cu:515-520 instead of `Model<ObjectiveT>`. Header-only; link -lcunvsm_amd.
//
//   reference                                         here
//   Model(num_words, num_entities, desc, train_cfg)   cunvsm_amd::Model(nvsm_config)
//   model.initialize(&rng)                            model.initialize(seed)   |   model.initialize(rng) with the caller's
//                                                     std::minstd_rand0, whose state is handed over and taken back
//   ForwardResult* r = model.compute_cost(batch,&rng) model.compute_cost(batch)            (result lives in the handle)
//   Gradients* g = model.compute_gradients(*r)        model.compute_gradients()
//   model.update(*g, lr, r->scaled_regularization_lambda())   model.update(lr, model.scaled_regularization_lambda())
//   r->get_cost()                                     model.get_cost()
//   model.get_data()                                  model.get_data()
//
// Error behaviour: the reference CHECK()s and aborts; here every failure throws cunvsm_amd::Error carrying the
// nvsm_status (never aborts the process).
#pragma once

#include <map>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../cunvsm_amd.h"

namespace cunvsm_amd {

struct Error : std::runtime_error {
    int status;
    Error(int st, const char* what) : std::runtime_error(what), status(st) {}
};

inline void check(int st) {
    if (st != NVSM_OK) throw Error(st, nvsm_last_error());
}

// TextEntity::Batch (include/cuNVSM/data.h:114-177) — a view over the caller's four arrays.
struct Batch {
    nvsm_batch raw{};
    Batch(const int64_t* features, const float* feature_weights, const int64_t* labels, const float* weights,
          int64_t num_instances, bool on_device = false) {
        raw.features = features; raw.feature_weights = feature_weights; raw.labels = labels; raw.weights = weights;
        raw.num_instances = num_instances; raw.on_device = on_device ? 1 : 0;
    }
    int64_t num_instances() const { return raw.num_instances; }
};

class Model {
 public:
    explicit Model(const nvsm_config& cfg) : cfg_(cfg) { check(nvsm_create(&cfg, &h_)); }
    ~Model() { nvsm_destroy(h_); }
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;

    void initialize(uint64_t seed) { check(nvsm_initialize(h_, seed)); }
    // model.initialize(&rng) (cpp/model.cu:37-43) with the reference's own generator object: the Glorot draws continue
    // from *rng's state and *rng is advanced past them, exactly as if the reference had consumed it (cpp/main.cu:520)
    void initialize(std::minstd_rand0* rng) {
        check(nvsm_rng_set_state(h_, state_of(*rng)));
        check(nvsm_initialize_from_rng_state(h_));
        sync_rng(rng);
    }
    // after compute_cost with the host sampler: advance the caller's generator past the negatives that were drawn
    void sync_rng(std::minstd_rand0* rng) {
        uint64_t s = 0;
        check(nvsm_rng_get_state(h_, &s));
        std::stringstream ss; ss << s; ss >> *rng;
    }
    void push_rng(const std::minstd_rand0& rng) { check(nvsm_rng_set_state(h_, state_of(rng))); }

    // entity_ids: optional output of the caller's own label generator; nullptr = sample as configured
    void compute_cost(const Batch& batch, const int64_t* entity_ids = nullptr) { check(nvsm_compute_cost(h_, &batch.raw, entity_ids)); }
    void compute_gradients() { check(nvsm_compute_gradients(h_)); }
    void update(float learning_rate, float scaled_regularization_lambda) { check(nvsm_update(h_, learning_rate, scaled_regularization_lambda)); }
    float get_cost() { float c = 0.f; check(nvsm_get_cost(h_, &c)); return c; }
    float scaled_regularization_lambda() { return nvsm_scaled_regularization_lambda(h_); }
    // backprop(result, lr) (cpp/model.cu:176-185)
    void backprop(float learning_rate) { compute_gradients(); update(learning_rate, scaled_regularization_lambda()); }
    // one iterate_data loop body (cpp/main.cu:400-444)
    float step(const Batch& batch, float learning_rate, bool want_cost = true) {
        float c = 0.f;
        check(nvsm_step(h_, &batch.raw, nullptr, learning_rate, want_cost ? &c : nullptr));
        return c;
    }

    // the same loop body with the loss read back one step late (cpp/main.cu:427-444 without the per-step stall):
    //   wait_inputs(); refill the host batch; t = step_deferred(batch, lr); cost of the PREVIOUS step = deferred_cost(t_prev)
    int64_t step_deferred(const Batch& batch, float learning_rate) {
        int64_t t = 0;
        check(nvsm_step_deferred(h_, &batch.raw, nullptr, learning_rate, &t));
        return t;
    }
    float deferred_cost(int64_t ticket) { float c = 0.f; check(nvsm_deferred_cost(h_, ticket, &c)); return c; }
    void wait_inputs() { check(nvsm_wait_inputs(h_)); }

    // ModelBase::get_data() (cpp/model.cu:64-93): name → host copy, in the layout write_to_hdf5 expects
    std::map<std::string, std::vector<float>> get_data() {
        static const char* names[] = {"word_representations-representations", "entity_representations-representations",
                                      "word_entity_mapping-transform", "word_entity_mapping-bias"};
        std::map<std::string, std::vector<float>> out;
        for (const char* n : names) {
            int64_t cnt = 0;
            check(nvsm_param_size(h_, n, &cnt));
            std::vector<float> v(static_cast<size_t>(cnt));
            check(nvsm_get_param(h_, n, v.data(), cnt));
            out.emplace(n, std::move(v));
        }
        return out;
    }
    void set_param(const std::string& name, const std::vector<float>& v) { check(nvsm_set_param(h_, name.c_str(), v.data(), static_cast<int64_t>(v.size()))); }

    // Storage::increment_parameter (cpp/storage.cu:123-131) — the gradient checker's poke
    void increment_parameter(const std::string& name, int64_t index, float epsilon) { check(nvsm_increment_parameter(h_, name.c_str(), index, epsilon)); }
    double get_cost_f64() { double c = 0.0; check(nvsm_get_cost_f64(h_, &c)); return c; }

    void synchronize() { check(nvsm_synchronize(h_)); }
    void comm_init(const char id[128]) { check(nvsm_comm_init(h_, id)); }
    nvsm_model* handle() { return h_; }
    const nvsm_config& config() const { return cfg_; }

 private:
    static uint64_t state_of(const std::minstd_rand0& rng) { std::stringstream ss; ss << rng; uint64_t s = 0; ss >> s; return s; }
    nvsm_config cfg_;
    nvsm_model* h_ = nullptr;
};

}  // namespace cunvsm_amd
