// cuNVSMTrainModel — the reference's trainer (cpp/main.cu) on top of libcunvsm_amd.so's C ABI: same option names,
// defaults and validation (cpp/main.cu:15-76,623-721), same training loop (iterate_data :366-469, train :492-621),
// same log lines and the same outputs ("<output>_meta" protobuf, "<output>_<epoch>[_<batch>].hdf5").
//
// Differences, all forced by what exists on this platform:
//   * <path to Indri index> is either an Indri 5.x repository directory, read without libindri (host/indri_index.hpp;
//     --document_list resolves docnos through the repository's docno look-up files), or the TREC-text collection file itself, indexed
//     in memory on start-up (host/trectext_index.hpp).
//   * only TextEntity::Objective (LSE / NVSM) is accelerated: non-zero --entity_similarity_weight /
//     --term_similarity_weight are refused with a clear message (the optional l2 normalisers are supported: untuned).
//   * extensions: --stopwords, --device, --sampler {host,device}, --allow_ragged_batches, and data parallelism over RCCL
//     (--gpus N spawns one process per GPU; or --world_size / --rank [or WORLD_SIZE / RANK / LOCAL_RANK] under any launcher).
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <thread>

#include <chrono>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <memory>
#include <set>

#include "../../include/cunvsm_amd.h"
#include "data.hpp"
#include "flags.hpp"
#include "hdf5_writer.hpp"
#include "index_source.hpp"
#include "indri_index.hpp"
#include "rendezvous.hpp"
#include "trectext_index.hpp"

using namespace nvsm_host;

namespace {

uint64_t FLAGS_num_epochs, FLAGS_document_cutoff, FLAGS_word_repr_size, FLAGS_entity_repr_size, FLAGS_batch_size, FLAGS_window_size,
    FLAGS_num_random_entities, FLAGS_seed, FLAGS_max_vocabulary_size, FLAGS_min_document_frequency;
std::string FLAGS_document_list, FLAGS_term_blacklist, FLAGS_update_method, FLAGS_weighting, FLAGS_feature_weighting, FLAGS_nonlinearity,
    FLAGS_output, FLAGS_stopwords, FLAGS_sampler;
double FLAGS_regularization_lambda, FLAGS_learning_rate, FLAGS_max_document_frequency, FLAGS_entity_similarity_weight,
    FLAGS_term_similarity_weight;
bool FLAGS_bias_negative_samples, FLAGS_l2_phrase_normalization, FLAGS_l2_entity_normalization, FLAGS_batch_normalization,
    FLAGS_include_oov, FLAGS_compute_initial_cost, FLAGS_check_gradients, FLAGS_no_shuffle, FLAGS_dump_initial_model,
    FLAGS_allow_ragged_batches, FLAGS_logtostderr, FLAGS_alsologtostderr, FLAGS_dp_exact_tables;
int64_t FLAGS_dump_every, FLAGS_v, FLAGS_device, FLAGS_minloglevel, FLAGS_gpus, FLAGS_world_size, FLAGS_rank;
std::string FLAGS_comm_id_file, FLAGS_comm_nonce;

void define_flags(Flags* f) {      // names, defaults and help strings of cpp/main.cu:15-76
    f->define_uint64("num_epochs", &FLAGS_num_epochs, 100000, "Number of training iterations.");
    f->define_uint64("document_cutoff", &FLAGS_document_cutoff, 0, "Number of documents per epoch (default: all).");
    f->define_string("document_list", &FLAGS_document_list, "", "Path to document list (default: all).");
    f->define_string("term_blacklist", &FLAGS_term_blacklist, "", "Path to term blacklist (default: none).");
    f->define_uint64("word_repr_size", &FLAGS_word_repr_size, 4, "Dimensionality of word representations.");
    f->define_uint64("entity_repr_size", &FLAGS_entity_repr_size, 4, "Dimensionality of entity representations.");
    f->define_uint64("batch_size", &FLAGS_batch_size, 1024, "Size of training batches.");
    f->define_uint64("window_size", &FLAGS_window_size, 8, "Size of training word windows.");
    f->define_uint64("num_random_entities", &FLAGS_num_random_entities, 1, "Number of random negative examples sampled for each positive example.");
    f->define_uint64("seed", &FLAGS_seed, 0, "Pseudo-random number generator seed.");
    f->define_double("regularization_lambda", &FLAGS_regularization_lambda, 0.01, "Regularization lambda.");
    f->define_double("learning_rate", &FLAGS_learning_rate, 0.0, "Learning rate.");
    f->define_string("update_method", &FLAGS_update_method, "", "Update method (sgd, adagrad, sparse_adam, dense_adam or full_adam).");
    f->define_string("weighting", &FLAGS_weighting, "auto", "Instance weighting strategy (auto, uniform or inv_doc_frequency).");
    f->define_string("feature_weighting", &FLAGS_feature_weighting, "uniform", "Feature weighting strategy (uniform or self_information).");
    f->define_bool("bias_negative_samples", &FLAGS_bias_negative_samples, false, "Introduces a bias towards negative samples. This is considered a bug in the CIKM model.");
    f->define_string("nonlinearity", &FLAGS_nonlinearity, "", "Nonlinearity (tanh or hard_tanh).");
    f->define_bool("l2_phrase_normalization", &FLAGS_l2_phrase_normalization, false, "Enables l2 normalization of phrase representations.");
    f->define_bool("l2_entity_normalization", &FLAGS_l2_entity_normalization, false, "Enables l2 normalization of entity representations.");
    f->define_bool("batch_normalization", &FLAGS_batch_normalization, false, "Enables batch normalization.");
    f->define_uint64("max_vocabulary_size", &FLAGS_max_vocabulary_size, 60000, "Maximum vocabulary size.");
    f->define_uint64("min_document_frequency", &FLAGS_min_document_frequency, 2, "Minimum document frequency of term in order to be retained by vocabulary filtering.");
    f->define_double("max_document_frequency", &FLAGS_max_document_frequency, 0.5, "Maximum document frequency of term in order to be retained by vocabulary filtering. If smaller than 1.0, then max_document_frequency is interpreted as relative to the index size; otherwise, it is considered an absolute threshold.");
    f->define_bool("include_oov", &FLAGS_include_oov, false, "Whether to include a special-purpose OoV token for term positions with a filtered dictionary term.");
    f->define_bool("compute_initial_cost", &FLAGS_compute_initial_cost, false, "Compute the cost before any learning is performed.");
    f->define_bool("check_gradients", &FLAGS_check_gradients, false, "Enable gradient checking. CAUTION: this will lead to insanely slow learning. "
                   "(Central differences with epsilon 1e-2 rather than the reference's 1e-4: the forward pass runs in fp32 — costs are read from the fp64 "
                   "device accumulator — and a disagreement within twice the fp32 resolution of the difference quotient is not held against a gradient.)");
    f->define_bool("no_shuffle", &FLAGS_no_shuffle, false, "Do not shuffle the training set.");
    f->define_bool("dump_initial_model", &FLAGS_dump_initial_model, false, "Dump the model after random initialization, but before training.");
    f->define_int64("dump_every", &FLAGS_dump_every, 0, "Number of batches that should be processed before the model is dumped during a single epoch. The model is always dumped at the end of every epoch.");
    f->define_double("entity_similarity_weight", &FLAGS_entity_similarity_weight, 0.0, "Mixture weight of the entity-entity objective.");
    f->define_double("term_similarity_weight", &FLAGS_term_similarity_weight, 0.0, "Mixture weight of the term-term objective.");
    f->define_string("output", &FLAGS_output, "", "Path to output model.");
    // extensions
    f->define_string("stopwords", &FLAGS_stopwords, "", "Stop list applied while indexing the collection (Indri <word> parameter file or plain words).");
    f->define_int64("device", &FLAGS_device, -1, "HIP device ordinal (default: LOCAL_RANK, else the rank, else 0).");
    f->define_int64("gpus", &FLAGS_gpus, 1, "Data parallel over this many GPUs of the node: the trainer starts one process per GPU (ranks 0..N-1 on devices "
                    "0..N-1). Every global batch of --batch_size windows is split into N contiguous slices; the projection / bias / batch-norm "
                    "gradients are all-reduced over RCCL each step, the embedding tables are updated rank-locally and averaged over the ranks at the "
                    "end of every epoch and before every model dump (rank 0 writes the outputs).");
    f->define_bool("dp_exact_tables", &FLAGS_dp_exact_tables, false, "Data parallel: every rank applies the embedding updates of ALL ranks' windows "
                   "(an all-gather of the update's inputs per step): the replicas stay identical and follow the single-GPU run on the "
                   "whole batch, at the table-update cost of the whole batch on every rank. Default: rank-local updates, averaged at "
                   "epoch ends and before dumps.");
    f->define_int64("world_size", &FLAGS_world_size, 0, "Data-parallel ranks when an external launcher starts them (default: WORLD_SIZE, else 1).");
    f->define_int64("rank", &FLAGS_rank, -1, "This process's rank (default: RANK, else 0).");
    f->define_string("comm_id_file", &FLAGS_comm_id_file, "", "File through which rank 0 hands the RCCL unique id to the other ranks "
                     "(default: comm_<hash of the run's nonce> in $XDG_RUNTIME_DIR or /tmp/cunvsm-<uid>, a directory private to the user).");
    f->define_string("comm_nonce", &FLAGS_comm_nonce, "", "What names this run to all of its ranks (default: the launcher's run id, else "
                     "parent pid + MASTER_PORT; --gpus N hands its children a random one). A rendezvous file with another nonce is ignored.");
    f->define_string("sampler", &FLAGS_sampler, "host", "Negative sampler: host (minstd_rand0, draw-for-draw the reference) or device.");
    f->define_bool("allow_ragged_batches", &FLAGS_allow_ragged_batches, false, "Train on batches whose size is not a multiple of 1024 instead of skipping them as the reference does.");
    // glog's own options that the reference's scripts pass
    f->define_bool("logtostderr", &FLAGS_logtostderr, true, "Log to stderr (there is no log-file sink).");
    f->define_bool("alsologtostderr", &FLAGS_alsologtostderr, true, "Accepted for compatibility.");
    f->define_int64("v", &FLAGS_v, 0, "Verbosity of VLOG messages.");
    f->define_int64("minloglevel", &FLAGS_minloglevel, 0, "Accepted for compatibility.");
}

void check_status(int status, const char* what) {
    if (status != NVSM_OK) NVSM_LOG(FATAL) << what << ": " << nvsm_last_error();
}
#define NVSM_CALL(expr) check_status((expr), #expr)

uint64_t rng_state(const RNG& rng) { std::stringstream ss; ss << rng; uint64_t s; ss >> s; return s; }
void rng_set_state(RNG* rng, uint64_t s) { std::stringstream ss; ss << s; ss >> *rng; }

template <typename T>
std::string vec_to_string(const std::vector<T>& v) {          // include/cuNVSM/base.h:121-128
    std::ostringstream os;
    os << std::setprecision(20) << "[";
    for (const T& x : v) os << x << ", ";
    os << "]";
    return os.str();
}

template <typename ContainerT>
ContainerT* read_strings(const std::string& path) {           // cpp/main.cu:148-163
    std::ifstream file(path);
    NVSM_CHECK(file.good()) << "cannot read " << path;
    ContainerT* strings = new ContainerT;
    std::string str;
    while (std::getline(file, str)) if (!str.empty()) strings->insert(strings->end(), str);
    return strings;
}

bool is_directory(const std::string& path) { struct stat st; return stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
bool is_file(const std::string& path) { struct stat st; return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

struct TrainConfig {
    uint64_t num_epochs, batch_size, window_size, num_random_entities;
    float regularization_lambda, learning_rate;
    int update_method, adam_mode;
    bool no_shuffle;
};

class Trainer {
 public:
    Trainer(nvsm_model* model, const TrainConfig& tc, int64_t num_words, int64_t num_entities, int dw, int de, int world_size, int rank)
        : model_(model), tc_(tc), num_words_(num_words), num_entities_(num_entities), dw_(dw), de_(de), world_size_(world_size), rank_(rank) {}

    // DumpModelFn (cpp/main.cu:335-364) + write_to_hdf5 (include/cuNVSM/lse_hdf5_inl.h)
    void dump_model(size_t epoch, const std::string& identifier) {
        // data parallel: the replicas' tables have drifted apart (rank-local sparse updates): what is written is their mean,
        // and every replica continues from it. A collective — every rank gets here at the same batch.
        if (world_size_ > 1) NVSM_CALL(nvsm_dp_average_tables(model_));
        if (FLAGS_output.empty() || rank_ != 0) return;
        std::stringstream ss;
        ss << FLAGS_output << "_" << epoch;
        if (!identifier.empty()) ss << "_" << identifier;
        ss << ".hdf5";
        const std::string filename = ss.str();
        std::vector<float> E(static_cast<size_t>(num_entities_) * de_), b(de_), T(static_cast<size_t>(de_) * dw_), W(static_cast<size_t>(num_words_) * dw_);
        NVSM_CALL(nvsm_get_param(model_, "entity_representations-representations", E.data(), static_cast<int64_t>(E.size())));
        NVSM_CALL(nvsm_get_param(model_, "word_entity_mapping-bias", b.data(), static_cast<int64_t>(b.size())));
        NVSM_CALL(nvsm_get_param(model_, "word_entity_mapping-transform", T.data(), static_cast<int64_t>(T.size())));
        NVSM_CALL(nvsm_get_param(model_, "word_representations-representations", W.data(), static_cast<int64_t>(W.size())));
        // ModelBase::get_data() is a std::map: datasets are created in name order; dims {cols, rows} (cpp/hdf5.cu:33-35)
        write_hdf5(filename, {
            {"entity_representations-representations", static_cast<unsigned long long>(num_entities_), static_cast<unsigned long long>(de_), E.data()},
            {"word_entity_mapping-bias", 1ull, static_cast<unsigned long long>(de_), b.data()},
            {"word_entity_mapping-transform", static_cast<unsigned long long>(dw_), static_cast<unsigned long long>(de_), T.data()},
            {"word_representations-representations", static_cast<unsigned long long>(num_words_), static_cast<unsigned long long>(dw_), W.data()}});
        NVSM_LOG(INFO) << "Saved model to " << filename << ".";
    }

    // iterate_data (cpp/main.cu:366-469)
    std::pair<size_t, float> iterate_data(bool backpropagate, DataSourceInterface* data_source, Batch* batch, size_t dump_epoch, bool may_dump) {
        size_t epoch_num_batches = 0;
        float agg_cost = 0.0;
        const auto iteration_start = std::chrono::steady_clock::now();
        // The reference reads the loss of every batch right after it (cpp/main.cu:427-444), which idles the GPU while the
        // host queues the next step's launches. Here the loss of batch k is read one step later, after batch k+1 has been
        // queued (nvsm_step_deferred), so that a whole step is always queued ahead of the GPU; the aggregate, the
        // per-batch log line and the order of everything else are unchanged.
        struct Pending { bool valid = false; int64_t ticket = 0; size_t index = 0; std::chrono::steady_clock::time_point start; };
        Pending pending;
        auto finish = [&](Pending& p) {
            if (!p.valid) return;
            float cost = 0.f;
            NVSM_CALL(nvsm_deferred_cost(model_, p.ticket, &cost));
            agg_cost += cost;
            log_batch(p.index, cost, data_source, iteration_start, p.start);
            p.valid = false;
        };
        struct Range { explicit Range(const char* n) { nvsm_range_push(n); } ~Range() { nvsm_range_pop(); } };
        while (data_source->has_next()) {
            Range batch_range("Batch");                                                   // nvtxRangePush("Batch"), cpp/main.cu:386
            const auto batch_start = std::chrono::steady_clock::now();
            NVSM_CALL(nvsm_wait_inputs(model_));        // the previous step has copied this host batch to the device
            batch->clear();
            { Range fetch_range("FetchData"); data_source->next(batch); }                 // :388-390
            const size_t n_global = batch->num_instances();
            const size_t G = static_cast<size_t>(world_size_);
            if (n_global % 1024 != 0 && !FLAGS_allow_ragged_batches) {                        // maxThreadsPerBlock, :392-398
                NVSM_LOG(ERROR) << "Skipping Batch #" << epoch_num_batches << " as it is not a multiple of " << 1024 << " (" << n_global << " instances).";
            } else if (n_global % G != 0) {
                NVSM_LOG(ERROR) << "Skipping Batch #" << epoch_num_batches << " as it does not split evenly over " << G << " ranks (" << n_global << " instances).";
            } else if (n_global > 0) {
                // data parallel: every rank reads the same global batch (same seed, same stream) and trains on its contiguous
                // slice [rank·n/G, (rank+1)·n/G)
                const size_t n = n_global / G, lo = n * static_cast<size_t>(rank_), w = static_cast<size_t>(tc_.window_size);
                nvsm_batch b;
                b.features = batch->features() + lo * w; b.feature_weights = batch->feature_weights() + lo * w;
                b.labels = batch->labels() + lo; b.weights = batch->weights() + lo;
                b.num_instances = static_cast<int64_t>(n); b.on_device = 0;
                float cost = 0.f;
                if (FLAGS_check_gradients) {
                    NVSM_CALL(nvsm_compute_cost(model_, &b, nullptr));
                    NVSM_CALL(nvsm_compute_gradients(model_));
                    NVSM_CALL(nvsm_get_cost(model_, &cost));
                    // cpp/main.cu:414-425 passes epsilon 1e-4 (and is validated in the fp64 test build). The forward pass here is
                    // fp32: per-term rounding puts ~4e-9 of noise on the cost, as large as the 2·ε·g signal of the small
                    // gradients at ε = 1e-4; ε = 1e-2 keeps the same 10 % threshold meaningful for every parameter.
                    NVSM_CHECK(check_gradients(b, cost, 1e-2f, 1e-1f)) << "Gradient check failed.";
                    if (backpropagate) NVSM_CALL(nvsm_update(model_, tc_.learning_rate, nvsm_scaled_regularization_lambda(model_)));
                } else if (backpropagate) {
                    // compute_cost → compute_gradients → update(lr, scaled λ) → get_cost (cpp/main.cu:405-444) as ONE call:
                    // same arithmetic, the independent halves of the backward pass overlapped on two streams
                    int64_t ticket = 0;
                    NVSM_CALL(nvsm_step_deferred(model_, &b, nullptr, tc_.learning_rate, &ticket));
                    finish(pending);
                    pending.valid = true; pending.ticket = ticket; pending.index = epoch_num_batches; pending.start = batch_start;
                    windows_ += n_global;
                    if (may_dump && FLAGS_dump_every > 0 && epoch_num_batches > 0 && epoch_num_batches % static_cast<size_t>(FLAGS_dump_every) == 0)
                        dump_model(dump_epoch, std::to_string(epoch_num_batches));
                    ++epoch_num_batches;
                    continue;
                } else {
                    NVSM_CALL(nvsm_compute_cost(model_, &b, nullptr));
                    NVSM_CALL(nvsm_compute_gradients(model_));
                    NVSM_CALL(nvsm_get_cost(model_, &cost));
                }
                agg_cost += cost;
                windows_ += n_global;
                log_batch(epoch_num_batches, cost, data_source, iteration_start, batch_start);
            }
            if (may_dump && FLAGS_dump_every > 0 && epoch_num_batches > 0 && epoch_num_batches % static_cast<size_t>(FLAGS_dump_every) == 0)
                dump_model(dump_epoch, std::to_string(epoch_num_batches));
            ++epoch_num_batches;
        }
        finish(pending);
        NVSM_CHECK(epoch_num_batches > 0) << "No batches to train during epoch";
        return std::make_pair(epoch_num_batches, agg_cost);
    }

    // the per-batch log line of cpp/main.cu:445-451
    void log_batch(size_t index, float cost, DataSourceInterface* data_source, std::chrono::steady_clock::time_point iteration_start,
                   std::chrono::steady_clock::time_point batch_start) {
        if (verbosity() < 1) return;
        const double epoch_duration = std::chrono::duration<double>(std::chrono::steady_clock::now() - iteration_start).count();
        const double batch_duration = std::chrono::duration<double>(std::chrono::steady_clock::now() - batch_start).count();
        const double progress = data_source->progress();
        std::string remaining = "unknown time";
        if (progress > 0.0 && std::isfinite(progress)) remaining = seconds_to_humanreadable_time((1.0 - progress) * (epoch_duration / progress));
        NVSM_LOG(INFO) << "Batch #" << index << " (" << std::setprecision(8) << progress * 100.0 << "%; " << remaining
                       << " remaining): cost=" << cost << ", duration=" << batch_duration;
    }

    uint64_t windows() const { return windows_; }

    // GradientCheckFn (cpp/gradient_check.cu:3-140) as main.cu calls it (:414-425: epsilon 1e-4, relative error
    // threshold 1e-1): central differences of the cost for EVERY scalar parameter, negatives replayed (the sampled
    // document ids of the batch are read back once and handed to every re-evaluation).
    bool check_gradients(const nvsm_batch& b, float cost, float epsilon, float relative_error_threshold) {
        const int64_t B = b.num_instances, w = static_cast<int64_t>(tc_.window_size), R = static_cast<int64_t>(tc_.num_random_entities) + 1;
        const int64_t N = B * R;
        std::vector<float> idsf(N), gT(static_cast<size_t>(de_) * dw_), gb(de_), gphrase(static_cast<size_t>(B) * dw_), gent(static_cast<size_t>(N) * de_);
        NVSM_CALL(nvsm_get_tensor(model_, "entity_ids", idsf.data(), N));
        NVSM_CALL(nvsm_get_tensor(model_, "grad_transform", gT.data(), static_cast<int64_t>(gT.size())));
        NVSM_CALL(nvsm_get_tensor(model_, "grad_bias", gb.data(), static_cast<int64_t>(gb.size())));
        NVSM_CALL(nvsm_get_tensor(model_, "grad_phrase", gphrase.data(), static_cast<int64_t>(gphrase.size())));
        NVSM_CALL(nvsm_get_tensor(model_, "grad_entity", gent.data(), static_cast<int64_t>(gent.size())));
        std::vector<int64_t> ids(N);
        for (int64_t i = 0; i < N; ++i) ids[i] = static_cast<int64_t>(idsf[i]);
        // RepresentationsStorage::get_parameter_gradient (cpp/storage.cu:133-183), densified once on the host
        std::vector<double> gW(static_cast<size_t>(num_words_) * dw_, 0.0), gE(static_cast<size_t>(num_entities_) * de_, 0.0);
        for (int64_t e = 0; e < B * w; ++e) {
            const int64_t r = b.features[e];
            const double wt = b.feature_weights ? b.feature_weights[e] : 1.0;
            for (int t = 0; t < dw_; ++t) gW[r * dw_ + t] += wt * gphrase[(e / w) * dw_ + t];
        }
        for (int64_t j = 0; j < N; ++j)
            for (int t = 0; t < de_; ++t) gE[ids[j] * de_ + t] += gent[static_cast<size_t>(j) * de_ + t];

        auto eval = [&]() {       // the cost in fp64 (the device accumulator's precision): float costs differ in the 7th digit only
            double c = 0.0;
            NVSM_CALL(nvsm_compute_cost(model_, &b, ids.data()));
            NVSM_CALL(nvsm_get_cost_f64(model_, &c));
            return c;
        };
        // "Sanity check to make sure we're getting the right RNG state" (:28-29)
        NVSM_CHECK(std::fabs(eval() - cost) <= 1e-5 * std::max(1.0, std::fabs(static_cast<double>(cost)))) << "cost is not reproducible with the recorded document ids";

        bool checked = true;
        size_t num_checked = 0;
        auto check_param = [&](const char* what, const char* name, int64_t idx, double gradient) {
            const float gradient_predict = static_cast<float>(-gradient);            // :42-43
            NVSM_CALL(nvsm_increment_parameter(model_, name, idx, epsilon));
            const double cost_added = eval();
            NVSM_CALL(nvsm_increment_parameter(model_, name, idx, -2.0f * epsilon));
            const double cost_removed = eval();
            NVSM_CALL(nvsm_increment_parameter(model_, name, idx, epsilon));
            const float gradient_approx = static_cast<float>((cost_added - cost_removed) / (2.0 * epsilon));
            const float relative_error = std::fabs(gradient_predict - gradient_approx) / std::max(std::fabs(gradient_predict), std::fabs(gradient_approx));
            const float ratio = gradient_approx != 0.f ? gradient_predict / gradient_approx : NAN;
            ++num_checked;
            // The forward pass is fp32 (the reference's checker runs in its double-precision test build): the two costs
            // carry ≈3e-8 relative noise each, so the difference quotient resolves a derivative only down to
            // 3e-8·|cost| / (2ε). A disagreement inside twice that resolution says nothing about the gradient (it shows up
            // with --l2_entity_normalization, whose gradients are of that size) and is not held against it.
            const double resolution = 3e-8 * std::max(1.0, std::fabs(static_cast<double>(cost))) / (2.0 * epsilon);
            if (std::fabs(static_cast<double>(gradient_predict) - gradient_approx) <= 2.0 * resolution) {
                NVSM_VLOG(2) << "Parameter " << idx << " of " << what << " agrees within the fp32 resolution of the difference quotient (approx="
                             << gradient_approx << ", predict=" << gradient_predict << ").";
                return;
            }
            if (gradient_predict * gradient_approx < 0.f) {
                NVSM_LOG(ERROR) << "Parameter " << idx << " of " << what << " has gradient with incorrect direction (approx=" << gradient_approx
                                << ", predict=" << gradient_predict << ", ratio=" << ratio << ", relative error=" << relative_error << ").";
                checked = false;
            } else if (relative_error >= relative_error_threshold) {
                NVSM_VLOG(1) << "Parameter " << idx << " of " << what << " most likely has incorrect gradient (approx=" << gradient_approx
                             << ", predict=" << gradient_predict << ", ratio=" << ratio << ", relative error=" << relative_error << ").";
                if (!std::isnan(ratio)) checked = false;
            } else if (gradient_approx != 0.f || gradient_predict != 0.f) {
                NVSM_VLOG(2) << "Parameter " << idx << " of " << what << " has correct gradient (approx=" << gradient_approx << ", predict="
                             << gradient_predict << ", ratio=" << ratio << ", relative error=" << relative_error << ").";
            }
        };
        // model->params_ in ParamIdentifier order: WORD_REPRS, TRANSFORM (projection then bias), ENTITY_REPRS
        for (size_t i = 0; i < gW.size(); ++i) check_param("word representations", "word_representations-representations", static_cast<int64_t>(i), gW[i]);
        for (size_t i = 0; i < gT.size(); ++i) check_param("transform", "word_entity_mapping-transform", static_cast<int64_t>(i), gT[i]);
        for (size_t i = 0; i < gb.size(); ++i) check_param("transform (bias)", "word_entity_mapping-bias", static_cast<int64_t>(i), gb[i]);
        for (size_t i = 0; i < gE.size(); ++i) check_param("entity representations", "entity_representations-representations", static_cast<int64_t>(i), gE[i]);
        NVSM_CHECK(std::fabs(eval() - cost) <= 1e-5 * std::max(1.0, std::fabs(static_cast<double>(cost)))) << "parameters were not restored";
        // leave the model as the caller had it: forward result and gradients of this batch
        NVSM_CALL(nvsm_compute_gradients(model_));
        NVSM_VLOG(1) << "Gradient check: " << num_checked << " parameters, " << (checked ? "passed" : "FAILED") << ".";
        return checked;
    }

 private:
    nvsm_model* model_;
    TrainConfig tc_;
    int64_t num_words_, num_entities_;
    int dw_, de_;
    int world_size_, rank_;
    uint64_t windows_ = 0;
};

void* pinned_alloc(size_t bytes) { void* p = nullptr; check_status(nvsm_host_alloc(bytes, &p), "nvsm_host_alloc"); return p; }
void pinned_free(void* p) { (void)nvsm_host_free(p); }

int run(int argc, char** argv) {
    Flags flags;
    define_flags(&flags);
    const std::vector<std::string> args = flags.parse(argc, argv);
    verbosity() = static_cast<int>(FLAGS_v);
    log_to_stderr() = FLAGS_logtostderr || FLAGS_alsologtostderr;

    if (args.size() < 2) {
        std::cerr << "Usage: " << args[0] << " [OPTIONS] <path to Indri index | TREC-text collection>\n" << flags.usage();
        NVSM_LOG(FATAL) << "Check failed: argc >= 2 Usage: " << args[0] << " [OPTIONS] <path to Indri index>";
    }
    static const std::map<std::string, std::pair<int, int>> UPDATE_METHODS = {                         // cpp/main.cu:479-485
        {"sgd", {NVSM_SGD, NVSM_ADAM_NONE}}, {"adagrad", {NVSM_ADAGRAD, NVSM_ADAM_NONE}}, {"sparse_adam", {NVSM_ADAM, NVSM_ADAM_SPARSE}},
        {"dense_adam", {NVSM_ADAM, NVSM_ADAM_DENSE_UPDATE}}, {"full_adam", {NVSM_ADAM, NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE}}};
    static const std::map<std::string, WeightingStrategy> WEIGHTING_STRATEGIES = {
        {"auto", AUTOMATIC_WEIGHTING}, {"uniform", UNIFORM}, {"inv_doc_frequency", INV_DOC_FREQUENCY}};  // :136-140
    static const std::map<std::string, TermWeightingStrategy> FEATURE_WEIGHTING_STRATEGIES = {
        {"uniform", UNIFORM_TERM_WEIGHTING}, {"self_information", SELF_INFORMATION_TERM_WEIGHTING}};    // :142-145
    static const std::map<std::string, int> NONLINEARITIES = {{"tanh", NVSM_TANH}, {"hard_tanh", NVSM_HARD_TANH}};   // :487-490
    NVSM_CHECK(UPDATE_METHODS.count(FLAGS_update_method)) << "Please specify a valid --update_method.";
    NVSM_CHECK(WEIGHTING_STRATEGIES.count(FLAGS_weighting)) << "Please specify a valid --weighting.";
    NVSM_CHECK(FEATURE_WEIGHTING_STRATEGIES.count(FLAGS_feature_weighting)) << "Please specify a valid --feature_weighting.";
    NVSM_CHECK(NONLINEARITIES.count(FLAGS_nonlinearity)) << "Please specify a valid --nonlinearity.";
    NVSM_CHECK(FLAGS_sampler == "host" || FLAGS_sampler == "device") << "--sampler must be host or device.";

    // ---- data parallelism: --gpus N spawns the ranks; otherwise world size / rank come from the flags or the launcher's environment
    auto env_int = [](const char* name, int64_t dflt) { const char* v = std::getenv(name); return (v && *v) ? static_cast<int64_t>(std::atoll(v)) : dflt; };
    if (FLAGS_gpus > 1 && FLAGS_world_size == 0 && !std::getenv("WORLD_SIZE")) {
        if (nvsm_device_count() < FLAGS_gpus) NVSM_LOG(FATAL) << "--gpus " << FLAGS_gpus << " but only " << nvsm_device_count() << " HIP device(s) are visible.";
        // re-execute this binary once per GPU with explicit --world_size / --rank / --device / --comm_id_file (the children
        // initialise HIP themselves; the parent only waits)
        // a nonce of this launch: a rendezvous file any earlier run left behind cannot carry it
        const std::string nonce = FLAGS_comm_nonce.empty()
            ? "gpus:" + std::to_string(static_cast<long long>(getpid())) + ":" + std::to_string(static_cast<long long>(wall_clock_ns())) : FLAGS_comm_nonce;
        const std::string id_file = FLAGS_comm_id_file.empty() ? default_comm_id_path(nonce) : FLAGS_comm_id_file;
        rendezvous_clear(id_file);
        std::vector<pid_t> kids;
        for (int64_t r = 0; r < FLAGS_gpus; ++r) {
            const pid_t pid = fork();
            NVSM_CHECK(pid >= 0) << "fork failed";
            if (pid == 0) {
                std::vector<std::string> av(argv, argv + argc);
                av.push_back("--world_size=" + std::to_string(FLAGS_gpus)); av.push_back("--rank=" + std::to_string(r));
                av.push_back("--device=" + std::to_string(r)); av.push_back("--comm_id_file=" + id_file);
                av.push_back("--comm_nonce=" + nonce);
                std::vector<char*> cav;
                for (std::string& a : av) cav.push_back(&a[0]);
                cav.push_back(nullptr);
                execv("/proc/self/exe", cav.data());
                std::perror("execv");
                _exit(127);
            }
            kids.push_back(pid);
        }
        int rc = 0;
        for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1; }
        std::remove(id_file.c_str());
        return rc;
    }
    const int world_size = static_cast<int>(FLAGS_world_size > 0 ? FLAGS_world_size : env_int("WORLD_SIZE", 1));
    const int rank = static_cast<int>(FLAGS_rank >= 0 ? FLAGS_rank : env_int("RANK", 0));
    NVSM_CHECK(world_size >= 1 && rank >= 0 && rank < world_size) << "bad --world_size / --rank";
    if (FLAGS_device < 0) FLAGS_device = world_size > 1 ? env_int("LOCAL_RANK", rank) : 0;
    {
        // this thread — it queues every launch of every step — and the prefetch thread it starts later, onto the CPUs of the GPU's
        // NUMA node (include/cunvsm_amd.h nvsm_bind_host_thread; NVSM_BIND_HOST=0: no)
        int node = -1;
        if (nvsm_bind_host_thread(static_cast<int>(FLAGS_device), &node) == NVSM_OK) NVSM_VLOG(1) << "GPU " << FLAGS_device << " hangs off NUMA node " << node << ".";
    }
    // RCCL bootstrap file (rendezvous.hpp): rank 0 removes whatever an earlier run left under the name FIRST THING — before
    // the minutes it spends indexing the collection, and long before any other rank of this run looks for the file
    const int64_t process_start_ns = wall_clock_ns();
    const std::string comm_nonce = world_size > 1 ? comm_run_nonce(FLAGS_comm_nonce) : std::string();
    const std::string comm_id_file = world_size > 1 ? (FLAGS_comm_id_file.empty() ? default_comm_id_path(comm_nonce) : FLAGS_comm_id_file) : std::string();
    if (world_size > 1 && rank == 0) rendezvous_clear(comm_id_file);
    if (world_size > 1) {
        NVSM_CHECK(FLAGS_batch_size % static_cast<uint64_t>(world_size) == 0) << "--batch_size must be a multiple of the number of ranks.";
        NVSM_CHECK(!FLAGS_check_gradients) << "--check_gradients is a single-GPU diagnostic.";
        if (rank != 0 && FLAGS_v < 2) { FLAGS_v = 0; verbosity() = 0; }       // rank 0 narrates
    }

    const std::string repository_path = args[1];
    std::unique_ptr<IndexInterface> index;
    if (is_directory(repository_path)) {
        // an Indri repository (cpp/data_indri.cpp:18-66), read without libindri (host/indri_index.hpp)
        if (!IndriDiskIndex::looks_like_repository(repository_path))
            NVSM_LOG(FATAL) << "Unable to open Indri parameters: " << repository_path << " holds no manifest / index.";
        NVSM_LOG(INFO) << "Opening Indri repository " << repository_path << ".";
        IndriDiskIndex* disk = IndriDiskIndex::open(repository_path);
        NVSM_LOG(INFO) << "Index holds " << disk->documentCount() << " documents, " << disk->termCount() << " term occurrences, "
                       << disk->uniqueTermCount() << " unique terms.";
        index.reset(disk);
    } else {
        NVSM_CHECK(is_file(repository_path)) << "cannot read collection " << repository_path;
        NVSM_LOG(INFO) << "Indexing " << repository_path << ".";
        TrectextIndex* built = TrectextIndex::from_file(repository_path, FLAGS_stopwords);
        NVSM_LOG(INFO) << "Indexed " << built->documentCount() << " documents, " << built->termCount() << " term occurrences, "
                       << built->uniqueTermCount() << " unique terms.";
        index.reset(built);
    }

    NVSM_CHECK(FLAGS_max_vocabulary_size > 0);
    uint64_t max_document_frequency = 0;                                               // cpp/main.cu:662-671
    if (FLAGS_max_document_frequency <= 1.0) {
        max_document_frequency = static_cast<uint64_t>(std::ceil(index->documentCount() * FLAGS_max_document_frequency));
        NVSM_LOG(INFO) << "Setting max_document_frequency to " << max_document_frequency << ".";
    } else {
        max_document_frequency = static_cast<uint64_t>(FLAGS_max_document_frequency);
    }

    TrainConfig tc;
    tc.num_epochs = FLAGS_num_epochs; tc.batch_size = FLAGS_batch_size; tc.window_size = FLAGS_window_size;
    tc.num_random_entities = FLAGS_num_random_entities;
    tc.regularization_lambda = static_cast<float>(FLAGS_regularization_lambda);
    tc.learning_rate = static_cast<float>(FLAGS_learning_rate);
    tc.update_method = UPDATE_METHODS.at(FLAGS_update_method).first;
    tc.adam_mode = UPDATE_METHODS.at(FLAGS_update_method).second;
    tc.no_shuffle = FLAGS_no_shuffle;
    NVSM_CHECK(FLAGS_entity_similarity_weight >= 0.0 && FLAGS_entity_similarity_weight <= 1.0);
    NVSM_CHECK(FLAGS_term_similarity_weight >= 0.0 && FLAGS_term_similarity_weight <= 1.0);
    NVSM_CHECK(FLAGS_seed > 0) << "Please specify a --seed value.";
    if (tc.learning_rate == 0.0f) tc.learning_rate = (tc.update_method == NVSM_ADAM) ? 0.001f : 0.01f;   // :710-721
    if (FLAGS_entity_similarity_weight != 0.0 || FLAGS_term_similarity_weight != 0.0)
        NVSM_LOG(FATAL) << "only the text-entity objective (LSE / NVSM) is implemented on this platform; "
                           "--entity_similarity_weight and --term_similarity_weight must be 0.";

    NVSM_LOG(INFO) << "Model descriptor: word_repr_size: " << FLAGS_word_repr_size << " entity_repr_size: " << FLAGS_entity_repr_size
                   << " transform_desc { batch_normalization: " << (FLAGS_batch_normalization ? "true" : "false") << " nonlinearity: "
                   << (NONLINEARITIES.at(FLAGS_nonlinearity) == NVSM_TANH ? "TANH" : "HARD_TANH") << " } clip_sigmoid: true bias_negative_samples: "
                   << (FLAGS_bias_negative_samples ? "true" : "false");
    NVSM_LOG(INFO) << "Data configuration: repository_path: \"" << repository_path << "\" max_vocabulary_size: " << FLAGS_max_vocabulary_size
                   << " min_document_frequency: " << FLAGS_min_document_frequency << " max_document_frequency: " << max_document_frequency
                   << " include_oov: " << (FLAGS_include_oov ? "true" : "false");
    NVSM_LOG(INFO) << "Training configuration: num_epochs: " << tc.num_epochs << " batch_size: " << tc.batch_size << " window_size: "
                   << tc.window_size << " num_random_entities: " << tc.num_random_entities << " regularization_lambda: " << tc.regularization_lambda
                   << " learning_rate: " << tc.learning_rate << " update_method: " << FLAGS_update_method << " no_shuffle: " << (tc.no_shuffle ? "true" : "false");
    NVSM_LOG(INFO) << "FLOATING_POINT_TYPE=float32";

    RNG rng;
    rng.seed(static_cast<RNG::result_type>(FLAGS_seed));                                 // cpp/main.cu:729-730

    if (nvsm_device_count() < 1) NVSM_LOG(FATAL) << "no HIP device visible: cuNVSMTrainModel has no CPU path.";
    set_batch_allocator(pinned_alloc, pinned_free);

    // construct_data_source<TextEntity::Objective> (:228-238): IndriSource wrapped in AsyncSource(10 batches)
    std::unique_ptr<std::vector<std::string>> document_list;
    if (!FLAGS_document_list.empty()) {
        NVSM_LOG(INFO) << "Reading document list from " << FLAGS_document_list << ".";
        document_list.reset(read_strings<std::vector<std::string>>(FLAGS_document_list));
    }
    std::unique_ptr<IndexSource::TermBlacklist> term_blacklist;
    if (!FLAGS_term_blacklist.empty()) {
        NVSM_LOG(INFO) << "Reading term blacklist from " << FLAGS_term_blacklist << ".";
        term_blacklist.reset(read_strings<IndexSource::TermBlacklist>(FLAGS_term_blacklist));
    }
    IndexSource* index_source = new IndexSource(
        index.release(), tc.window_size, &rng, FLAGS_max_vocabulary_size, FLAGS_min_document_frequency, max_document_frequency,
        FLAGS_document_cutoff, FLAGS_include_oov, false /* include_digits */, document_list.get(), term_blacklist.get(),
        !tc.no_shuffle, AUTOMATIC_SAMPLING, WEIGHTING_STRATEGIES.at(FLAGS_weighting), FEATURE_WEIGHTING_STRATEGIES.at(FLAGS_feature_weighting));
    std::unique_ptr<DataSourceInterface> data_source(new AsyncSource(10, tc.batch_size, tc.window_size, index_source));

    Metadata meta;
    data_source->extract_metadata(&meta);
    const size_t vocabulary_size = meta.term_size(), corpus_size = meta.object_size();
    NVSM_CHECK(vocabulary_size > 0);
    NVSM_CHECK(corpus_size > 0);
    NVSM_LOG(INFO) << "Training statistics: vocabulary size=" << vocabulary_size << ", corpus size=" << corpus_size;

    nvsm_config cfg;
    nvsm_config_default(&cfg);
    cfg.num_words = static_cast<int64_t>(vocabulary_size); cfg.num_entities = static_cast<int64_t>(corpus_size);
    cfg.word_repr_size = static_cast<int32_t>(FLAGS_word_repr_size); cfg.entity_repr_size = static_cast<int32_t>(FLAGS_entity_repr_size);
    cfg.batch_normalization = FLAGS_batch_normalization; cfg.nonlinearity = NONLINEARITIES.at(FLAGS_nonlinearity);
    cfg.clip_sigmoid = 1;                                                                // :645
    cfg.bias_negative_samples = FLAGS_bias_negative_samples;
    cfg.l2_normalize_phrase_reprs = FLAGS_l2_phrase_normalization; cfg.l2_normalize_entity_reprs = FLAGS_l2_entity_normalization;
    cfg.window_size = static_cast<int32_t>(tc.window_size); cfg.num_random_entities = static_cast<int32_t>(tc.num_random_entities);
    cfg.regularization_lambda = tc.regularization_lambda;
    cfg.update_method = tc.update_method; cfg.adam_mode = tc.adam_mode;
    cfg.max_batch_size = static_cast<int32_t>(tc.batch_size / static_cast<uint64_t>(world_size));      // this rank's slice
    cfg.world_size = world_size; cfg.rank = rank; cfg.sync_batch_norm = 1;
    cfg.dp_exact_tables = (world_size > 1 && FLAGS_dp_exact_tables) ? 1 : 0;
    cfg.device = static_cast<int32_t>(FLAGS_device);
    cfg.sampler = FLAGS_sampler == "host" ? NVSM_SAMPLER_HOST_MINSTD : NVSM_SAMPLER_DEVICE;
    nvsm_model* model = nullptr;
    NVSM_CALL(nvsm_create(&cfg, &model));
    if (world_size > 1) {
        // RCCL bootstrap without a rendezvous service: rank 0 publishes the 128-byte ncclUniqueId through a file (rendezvous.hpp:
        // private directory, exclusive create, no links followed, written under a temporary name and renamed so that a reader
        // never sees half of it); the readers only accept a file of their own user that carries this run's nonce and was
        // created after they themselves started (minus the skew between the ranks' start-ups). ncclCommInitRank is itself a
        // barrier, so once it returns on rank 0 every rank has read the file and it can go.
        const std::string& id_file = comm_id_file;
        char id[kCommIdBytes];
        if (rank == 0) {
            NVSM_CALL(nvsm_comm_unique_id(id));
            rendezvous_publish(id_file, comm_nonce, id);
        } else {
            bool got = false;
            std::string why;
            const int64_t not_before = process_start_ns - int64_t(120) * 1000000000;      // ranks of one launch start within two minutes
            for (int tries = 0; tries < 6000 && !got; ++tries) {       // up to 10 minutes: rank 0 may still be indexing the collection
                got = rendezvous_read(id_file, comm_nonce, not_before, id, &why);
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            NVSM_CHECK(got) << "rank " << rank << ": no RCCL id for this run appeared in " << id_file << " (the file " << why << ")";
        }
        NVSM_CALL(nvsm_comm_init(model, id));
        if (rank == 0) rendezvous_clear(id_file);
        int ranks = 0;
        NVSM_CALL(nvsm_comm_size(model, &ranks));
        NVSM_LOG(INFO) << "Data parallel: rank " << rank << " of " << world_size << " on device " << FLAGS_device << " (RCCL communicator of " << ranks
                       << " ranks); " << cfg.max_batch_size << " of every " << tc.batch_size << " windows.";
    }
    // model.initialize(rng): the SAME generator the data source has just drawn from (cpp/main.cu:497-520)
    NVSM_CALL(nvsm_rng_set_state(model, rng_state(rng)));
    NVSM_CALL(nvsm_initialize_from_rng_state(model));
    NVSM_CALL(nvsm_synchronize(model));
    const uint64_t num_parameters = vocabulary_size * FLAGS_word_repr_size + corpus_size * FLAGS_entity_repr_size +
                                    FLAGS_entity_repr_size * FLAGS_word_repr_size + FLAGS_entity_repr_size;
    NVSM_LOG(INFO) << "Initialized cuNVSM with " << num_parameters << " parameters for training on " << vocabulary_size << " words and "
                   << corpus_size << " objects.";

    if (!FLAGS_output.empty() && rank == 0) {                                            // :527-537
        std::ofstream meta_file(FLAGS_output + "_meta", std::ios::binary);
        const std::string wire = meta.SerializeAsString();
        meta_file.write(wire.data(), static_cast<std::streamsize>(wire.size()));
        NVSM_CHECK(meta_file.good()) << "cannot write " << FLAGS_output << "_meta";
    }

    Trainer trainer(model, tc, static_cast<int64_t>(vocabulary_size), static_cast<int64_t>(corpus_size),
                    static_cast<int>(FLAGS_word_repr_size), static_cast<int>(FLAGS_entity_repr_size), world_size, rank);
    Batch batch(tc.batch_size, tc.window_size);
    std::vector<float> epoch_costs;

    // data_source->reset() re-shuffles with the shared generator (cpp/data_indri.cpp:404): hand the state over and back
    auto reset_data_source = [&] {
        uint64_t s = 0;
        NVSM_CALL(nvsm_rng_get_state(model, &s));
        rng_set_state(&rng, s);
        data_source->reset();
        NVSM_CALL(nvsm_rng_set_state(model, rng_state(rng)));
    };

    if (FLAGS_compute_initial_cost) {                                                    // :543-561
        const auto r = trainer.iterate_data(false, data_source.get(), &batch, 0, false);
        reset_data_source();
        epoch_costs.push_back(r.second / r.first);
        NVSM_LOG(INFO) << "Epoch #0 (initial): cost=" << vec_to_string(epoch_costs);
    }
    if (FLAGS_dump_initial_model) trainer.dump_model(0, "");
    // (libhdf5 is resolved with dlopen: loaded and initialised here, where the reference's loader has long done it, not inside
    //  the first dump at the end of epoch 1 — on a cold box that was 0.9 s of the second epoch's cumulative batches-per-second)
    if (!FLAGS_output.empty() && rank == 0) hdf5_preload();

    const auto start = std::chrono::steady_clock::now();
    size_t num_batches = 0;
    for (size_t epoch = 1; epoch <= tc.num_epochs; ++epoch) {                            // :575-620
        nvsm_range_push("Epoch");                                                        // :577
        const auto epoch_start = std::chrono::steady_clock::now();
        const uint64_t windows_before = trainer.windows();
        const auto r = trainer.iterate_data(true, data_source.get(), &batch, epoch, true);
        num_batches += r.first;
        const double epoch_duration = std::chrono::duration<double>(std::chrono::steady_clock::now() - epoch_start).count();
        const double total_duration = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        epoch_costs.push_back(r.second / r.first);
        NVSM_LOG(INFO) << "Epoch #" << epoch << ": duration=" << seconds_to_humanreadable_time(epoch_duration) << " ("
                       << num_batches / total_duration << " batches/second) cost=" << vec_to_string(epoch_costs);
        NVSM_VLOG(1) << "Epoch #" << epoch << ": " << (trainer.windows() - windows_before) / epoch_duration << " n-gram windows/second";
        trainer.dump_model(epoch, "");
        reset_data_source();
        nvsm_range_pop();
    }
    NVSM_CALL(nvsm_synchronize(model));
    NVSM_VLOG(1) << "Training loop: " << num_batches << " batches in "
                 << std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() << " seconds (incl. the model dumps)";
    data_source.reset();
    nvsm_destroy(model);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        return run(argc, argv);
    } catch (const FatalError&) {
        return 1;                       // the message has been logged (cpp/main.cu:113-134 exits 1 from its terminate handler)
    } catch (const std::exception& e) {
        std::cerr << "Exception: " << e.what() << std::endl;
        return 1;
    }
}
