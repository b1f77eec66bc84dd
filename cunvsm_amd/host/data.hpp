// Host data pipeline feeding the training step: the batch container handed to the C ABI (nvsm_batch) and the data
// sources that fill it. Mirrors include/cuNVSM/data.h — BatchInterface / DataSourceInterface (:50-84),
// TextEntity::Batch (:114-177) + cpp/data.cu:8-124, TextEntity::DataSource (:181-282), InMemoryDocumentSource
// (:299-365), AsyncSource (cpp/data_async.cpp), RepeatingSource (cpp/data_repeating.cpp). The pair objectives'
// sources (RepresentationSimilarity, MultiSource) are outside the hot-path scope.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>

#include "base.hpp"
#include "metadata.hpp"

namespace nvsm_host {

// ---- batch memory: the reference pins its batches with cudaHostAlloc (cpp/data.cu:16-27); the CLI installs the
// library's pinned allocator (nvsm_host_alloc), the CPU-only unit tests keep plain aligned memory -------------
typedef void* (*BatchAllocFn)(size_t bytes);
typedef void (*BatchFreeFn)(void* p);
void set_batch_allocator(BatchAllocFn alloc, BatchFreeFn free_fn);

typedef std::tuple<std::vector<WordIdxType>, std::vector<WeightType>, ObjectIdxType, WeightType> InstanceT;
typedef std::deque<InstanceT> InstancesT;

class Batch {
 public:
    Batch(size_t batch_size, size_t window_size);
    ~Batch();
    Batch(const Batch&) = delete;
    Batch& operator=(const Batch&) = delete;

    void clear() { num_instances_ = 0; }
    bool full() const { return num_instances_ == batch_size_; }
    bool empty() const { return num_instances_ == 0; }
    void swap(Batch* other);                       // pointer swap (cpp/data.cu:77-92)

    size_t num_instances() const { return num_instances_; }
    size_t maximum_size() const { return batch_size_; }
    size_t window_size() const { return window_size_; }

    // the four flat arrays of nvsm_batch
    const WordIdxType* features() const { return features_; }
    const WeightType* feature_weights() const { return feature_weights_; }
    const ObjectIdxType* labels() const { return labels_; }
    const WeightType* weights() const { return weights_; }

 private:
    friend class DataSource;
    const size_t batch_size_, window_size_;
    WordIdxType* features_;
    WeightType* feature_weights_;
    ObjectIdxType* labels_;
    WeightType* weights_;
    size_t num_instances_;
};

class DataSourceInterface {
 public:
    virtual ~DataSourceInterface() {}
    virtual void reset() = 0;
    virtual bool has_next() const = 0;
    virtual float progress() const = 0;
    virtual void extract_metadata(Metadata* metadata) const = 0;
    virtual void next(Batch* batch) = 0;
};

// TextEntity::DataSource (include/cuNVSM/data.h:181-282): overflow buffer + windowing helper.
class DataSource : public DataSourceInterface {
 public:
    DataSource(size_t vocabulary_size, size_t corpus_size) : vocabulary_size_(vocabulary_size), corpus_size_(corpus_size) {}

    void next(Batch* batch) override;              // drains the overflow buffer first (:193-205)
    bool has_next() const override { return !overflow_buffer_.empty(); }
    size_t vocabulary_size() const { return vocabulary_size_; }
    size_t corpus_size() const { return corpus_size_; }
    float progress() const override { return NAN; }
    void extract_metadata(Metadata*) const override {}

    // cpp/data.cu:94-124: copy into the next free slot, or park in the overflow buffer when the batch is full
    void push_instance(const std::vector<WordIdxType>& features, const std::vector<WeightType>& feature_weights,
                       ObjectIdxType object_id, WeightType weight, Batch* batch);
    void push_window(const WordIdxType* features, const WeightType* feature_weights, ObjectIdxType object_id, WeightType weight,
                     Batch* batch);

    // include/cuNVSM/data.h:236-273: sliding windows of batch->window_size() tokens, `stride` apart
    template <typename Iterable>
    void create_instances(const Iterable& tokens, ObjectIdxType object_id, WeightType weight, size_t stride, Batch* batch) {
        std::deque<WordIdxType> buffer;
        for (const auto token : tokens) {
            buffer.push_back(static_cast<WordIdxType>(token));
            if (buffer.size() == batch->window_size()) {
                push_instance(std::vector<WordIdxType>(buffer.begin(), buffer.end()), std::vector<WeightType>(), object_id, weight, batch);
                for (size_t i = 0; i < stride; ++i) buffer.pop_front();
            }
        }
        if (buffer.size() == batch->window_size())
            push_instance(std::vector<WordIdxType>(buffer.begin(), buffer.end()), std::vector<WeightType>(), object_id, weight, batch);
    }

 protected:
    size_t vocabulary_size_, corpus_size_;
    bool overflow_empty() const { return overflow_buffer_.empty(); }
    InstancesT overflow_buffer_;
};

typedef std::map<std::string, WordIdxType> VocabularyT;
typedef std::vector<std::pair<ObjectIdxType, std::string>> CorpusT;
VocabularyT construct_vocabulary(const std::vector<std::string>& words);       // include/cuNVSM/data.h:284-297

class InMemoryDocumentSource : public DataSource {                             // include/cuNVSM/data.h:299-365
 public:
    InMemoryDocumentSource(const VocabularyT& vocabulary, const CorpusT& documents, bool pad_batch = false)
        : DataSource(vocabulary.size(), documents.size()), vocabulary_(vocabulary), documents_(documents), pad_batch_(pad_batch) { reset(); }
    void reset() override { num_batches_emitted_ = 0; }
    void next(Batch* batch) override;
    bool has_next() const override { return DataSource::has_next() || num_batches_emitted_ < 2; }
 private:
    size_t num_batches_emitted_ = 0;
    const VocabularyT vocabulary_;
    const CorpusT documents_;
    const bool pad_batch_;
};

// AsyncSource (cpp/data_async.cpp): one worker thread keeps `num_concurrent_batches` pre-allocated batches filled;
// the consumer swaps pointers. The reference spins on two boost::lockfree queues; here the same hand-over runs on a
// mutex + condition variable (no busy-waiting next to the thread that feeds the GPU). Takes ownership of `source`.
class AsyncSource : public DataSourceInterface {
 public:
    AsyncSource(size_t num_concurrent_batches, size_t batch_size, size_t window_size, DataSourceInterface* source);
    ~AsyncSource() override;
    void reset() override;
    void next(Batch* batch) override;
    bool has_next() const override;
    float progress() const override { return source_->progress(); }
    void extract_metadata(Metadata* metadata) const override { source_->extract_metadata(metadata); }
 private:
    void start_worker();
    void stop_worker();
    void worker();
    std::unique_ptr<DataSourceInterface> source_;
    std::vector<std::unique_ptr<Batch>> buffers_;
    mutable std::mutex mu_;
    mutable std::condition_variable cv_;
    std::deque<Batch*> empty_, full_;
    std::thread thread_;
    bool stop_ = false, worker_done_ = true;
};

// RepeatingSource (cpp/data_repeating.cpp): replays a finite source `num_repeats` times (size_t(-1) = forever).
class RepeatingSource : public DataSourceInterface {
 public:
    RepeatingSource(size_t num_repeats, DataSourceInterface* source) : num_repeats_(num_repeats), source_(source) {}
    void reset() override { current_iteration_ = 0; source_->reset(); }
    void next(Batch* batch) override;
    bool has_next() const override;
    float progress() const override { return source_->progress() + static_cast<float>(current_iteration_) / num_repeats_; }
    void extract_metadata(Metadata* metadata) const override { source_->extract_metadata(metadata); }
 private:
    const size_t num_repeats_;
    std::unique_ptr<DataSourceInterface> source_;
    size_t current_iteration_ = 0;
};

}  // namespace nvsm_host
