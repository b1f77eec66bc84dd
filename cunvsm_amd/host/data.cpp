#include "data.hpp"

#include <algorithm>
#include <cstring>

namespace nvsm_host {

namespace {
void* default_alloc(size_t bytes) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, bytes ? bytes : 64) != 0) throw std::bad_alloc();
    return p;
}
void default_free(void* p) { std::free(p); }
BatchAllocFn g_alloc = default_alloc;
BatchFreeFn g_free = default_free;
}  // namespace

void set_batch_allocator(BatchAllocFn alloc, BatchFreeFn free_fn) {
    g_alloc = alloc ? alloc : default_alloc;
    g_free = free_fn ? free_fn : default_free;
}

// ---- Batch: cpp/data.cu:8-92 ----
Batch::Batch(size_t batch_size, size_t window_size)
    : batch_size_(batch_size), window_size_(window_size), features_(nullptr), feature_weights_(nullptr), labels_(nullptr),
      weights_(nullptr), num_instances_(0) {
    NVSM_CHECK(batch_size_ > 0);
    NVSM_CHECK(window_size_ > 0);
    features_ = static_cast<WordIdxType*>(g_alloc(batch_size_ * window_size_ * sizeof(WordIdxType)));
    feature_weights_ = static_cast<WeightType*>(g_alloc(batch_size_ * window_size_ * sizeof(WeightType)));
    labels_ = static_cast<ObjectIdxType*>(g_alloc(batch_size_ * sizeof(ObjectIdxType)));
    weights_ = static_cast<WeightType*>(g_alloc(batch_size_ * sizeof(WeightType)));
    clear();
}

Batch::~Batch() {
    g_free(features_); g_free(feature_weights_); g_free(labels_); g_free(weights_);
}

void Batch::swap(Batch* other) {
    NVSM_CHECK(other != nullptr);
    NVSM_CHECK(batch_size_ == other->batch_size_);
    NVSM_CHECK(window_size_ == other->window_size_);
    std::swap(features_, other->features_);
    std::swap(feature_weights_, other->feature_weights_);
    std::swap(labels_, other->labels_);
    std::swap(weights_, other->weights_);
    std::swap(num_instances_, other->num_instances_);
}

// ---- DataSource ----
void DataSource::next(Batch* batch) {
    NVSM_CHECK(batch->empty());
    while (!batch->full() && !overflow_buffer_.empty()) {
        const InstanceT& inst = overflow_buffer_.front();
        push_instance(std::get<0>(inst), std::get<1>(inst), std::get<2>(inst), std::get<3>(inst), batch);
        overflow_buffer_.pop_front();
    }
}

void DataSource::push_instance(const std::vector<WordIdxType>& features, const std::vector<WeightType>& feature_weights,
                               ObjectIdxType object_id, WeightType weight, Batch* batch) {
    if (batch->full()) {
        overflow_buffer_.push_back(std::make_tuple(features, feature_weights, object_id, weight));
        return;
    }
    const size_t w = batch->window_size();
    NVSM_CHECK(features.size() == w);
    std::copy(features.begin(), features.end(), &batch->features_[batch->num_instances_ * w]);
    if (!feature_weights.empty()) {
        NVSM_CHECK(feature_weights.size() == features.size());
        std::copy(feature_weights.begin(), feature_weights.end(), &batch->feature_weights_[batch->num_instances_ * w]);
    } else {
        std::fill(&batch->feature_weights_[batch->num_instances_ * w], &batch->feature_weights_[(batch->num_instances_ + 1) * w],
                  static_cast<WeightType>(1.0));
    }
    batch->labels_[batch->num_instances_] = object_id;
    batch->weights_[batch->num_instances_] = weight;
    ++batch->num_instances_;
}

// push_instance without the vectors: `features` / `feature_weights` (may be null: all 1) point at window_size values.
// The batch must have room (the callers check full() first).
void DataSource::push_window(const WordIdxType* features, const WeightType* feature_weights, ObjectIdxType object_id,
                             WeightType weight, Batch* batch) {
    const size_t w = batch->window_size(), at = batch->num_instances_;
    std::copy(features, features + w, &batch->features_[at * w]);
    if (feature_weights) std::copy(feature_weights, feature_weights + w, &batch->feature_weights_[at * w]);
    else std::fill(&batch->feature_weights_[at * w], &batch->feature_weights_[(at + 1) * w], static_cast<WeightType>(1.0));
    batch->labels_[at] = object_id;
    batch->weights_[at] = weight;
    ++batch->num_instances_;
}

VocabularyT construct_vocabulary(const std::vector<std::string>& words) {
    VocabularyT vocabulary;
    vocabulary["<UNK>"] = 0;
    for (const std::string& word : words)
        if (vocabulary.find(word) == vocabulary.end()) vocabulary.insert({word, static_cast<WordIdxType>(vocabulary.size())});
    return vocabulary;
}

void InMemoryDocumentSource::next(Batch* batch) {
    DataSource::next(batch);
    // at least one pass over the documents, more when pad_batch_ (include/cuNVSM/data.h:318-346)
    while (batch->num_instances() == 0 || (pad_batch_ && batch->num_instances() < batch->maximum_size())) {
        for (const auto& document : documents_) {
            std::vector<WordIdxType> tokens;
            for (const std::string& word : split(document.second)) {
                const auto it = vocabulary_.find(word);
                if (it != vocabulary_.end()) tokens.push_back(it->second);
            }
            const WeightType weight = static_cast<WeightType>(std::exp(-std::log(static_cast<double>(tokens.size()))));
            create_instances(tokens, document.first, weight, 1 /* stride */, batch);
        }
    }
    ++num_batches_emitted_;
}

// ---- AsyncSource ----
AsyncSource::AsyncSource(size_t num_concurrent_batches, size_t batch_size, size_t window_size, DataSourceInterface* source)
    : source_(source), buffers_(num_concurrent_batches) {
    NVSM_CHECK(num_concurrent_batches > 0);
    for (auto& buffer : buffers_) {
        buffer.reset(new Batch(batch_size, window_size));
        empty_.push_back(buffer.get());
    }
    start_worker();
}

AsyncSource::~AsyncSource() { stop_worker(); }

void AsyncSource::worker() {
    for (;;) {
        Batch* batch = nullptr;
        {
            std::unique_lock<std::mutex> lock(mu_);
            cv_.wait(lock, [&] { return stop_ || !empty_.empty(); });
            if (stop_) break;
            batch = empty_.front();
            empty_.pop_front();
        }
        // only this thread touches the wrapped source between start_worker() and stop_worker()
        if (!source_->has_next()) {
            std::lock_guard<std::mutex> lock(mu_);
            empty_.push_front(batch);
            break;
        }
        source_->next(batch);
        {
            std::lock_guard<std::mutex> lock(mu_);
            full_.push_back(batch);
        }
        cv_.notify_all();
    }
    {
        std::lock_guard<std::mutex> lock(mu_);
        worker_done_ = true;
    }
    cv_.notify_all();
}

void AsyncSource::start_worker() {
    NVSM_CHECK(!thread_.joinable());
    stop_ = false;
    worker_done_ = false;
    thread_ = std::thread(&AsyncSource::worker, this);
}

void AsyncSource::stop_worker() {
    {
        std::lock_guard<std::mutex> lock(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    if (thread_.joinable()) thread_.join();
}

void AsyncSource::reset() {
    // cpp/data_async.cpp:66-71 — batches the worker had already prepared are discarded with the old epoch
    stop_worker();
    {
        std::lock_guard<std::mutex> lock(mu_);
        while (!full_.empty()) { full_.front()->clear(); empty_.push_back(full_.front()); full_.pop_front(); }
    }
    source_->reset();
    start_worker();
}

bool AsyncSource::has_next() const {
    std::unique_lock<std::mutex> lock(mu_);
    cv_.wait(lock, [&] { return !full_.empty() || worker_done_; });
    return !full_.empty();
}

void AsyncSource::next(Batch* batch) {
    NVSM_CHECK(batch->empty());
    NVSM_CHECK(has_next());
    Batch* buffer_batch;
    {
        std::lock_guard<std::mutex> lock(mu_);
        buffer_batch = full_.front();
        full_.pop_front();
    }
    batch->swap(buffer_batch);
    buffer_batch->clear();
    {
        std::lock_guard<std::mutex> lock(mu_);
        empty_.push_back(buffer_batch);
    }
    cv_.notify_all();
}

// ---- RepeatingSource: cpp/data_repeating.cpp ----
void RepeatingSource::next(Batch* batch) {
    if (!source_->has_next()) {
        source_->reset();
        ++current_iteration_;
        NVSM_CHECK(current_iteration_ < num_repeats_);
    }
    source_->next(batch);
}

bool RepeatingSource::has_next() const {
    if (current_iteration_ + 1 < num_repeats_) return true;
    if (current_iteration_ + 1 == num_repeats_) return source_->has_next();
    NVSM_LOG(FATAL) << "This should not happen.";
    return false;
}

}  // namespace nvsm_host
