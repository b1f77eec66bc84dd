#include "index_source.hpp"

#include <algorithm>
#include <cmath>
#include <queue>

namespace nvsm_host {

// std::shuffle as libstdc++ implemented it up to GCC 6 — one uniform_int_distribution draw per element — which is the
// algorithm behind the seed-pinned expectation of the reference's own test (cpp/data_tests.cpp:571-575, "relies on
// seed == 1"; reproduced in tests/cpp/host_tests.cpp). GCC 7+ draws two swap positions per generator call, which
// permutes differently, so std::shuffle itself would not replay the reference's pinned order.
template <typename RandomIt, typename URNG>
static void shuffle_one_draw_per_element(RandomIt first, RandomIt last, URNG&& g) {
    if (first == last) return;
    typedef typename std::make_unsigned<typename std::iterator_traits<RandomIt>::difference_type>::type udiff_t;
    typedef std::uniform_int_distribution<udiff_t> distr_t;
    typedef typename distr_t::param_type param_t;
    distr_t d;
    for (RandomIt i = first + 1; i != last; ++i) std::iter_swap(i, first + d(g, param_t(0, static_cast<udiff_t>(i - first))));
}

// ---------------------------------------------------------------------------------------------
// instance generators — cpp/data_indri.cpp:107-410
// ---------------------------------------------------------------------------------------------
class InstanceGeneratorBase {
 public:
    explicit InstanceGeneratorBase(IndexSource* source) : source_(source) { NVSM_CHECK(source_ != nullptr); }
    virtual ~InstanceGeneratorBase() {}
    virtual void generate(InstancesT* instances) = 0;
    // Optional fast path of IndexSource::next: write the next instances straight into the batch until it is full or the
    // generator runs dry — the same instances in the same order generate() + push_instance would deliver, without the
    // two heap allocations per instance and the detour through the overflow deque. false = not implemented.
    virtual bool fill(Batch* /*batch*/) { return false; }
    virtual bool has_next() const = 0;
    virtual void reset() = 0;

 protected:
    // :112-136 — index term ids → model term ids; unknown terms become the OoV token 0 or are dropped
    void generate_terms(const std::vector<TERMID_T>& term_list, std::vector<WordIdxType>* terms) const {
        NVSM_CHECK(terms->empty());
        for (const TERMID_T term_id : term_list) {
            const auto it = source_->term_id_mapping_.find(term_id);
            if (it != source_->term_id_mapping_.end()) terms->push_back(static_cast<WordIdxType>(it->second));
            else if (source_->include_oov_) terms->push_back(0);
        }
    }
    IndexSource* source_;
};

class SequentialInstanceGenerator : public InstanceGeneratorBase {            // :138-222
 public:
    explicit SequentialInstanceGenerator(IndexSource* source) : InstanceGeneratorBase(source) { reset(); }

    void generate(InstancesT* instances) override {
        const ObjectIdxType label = static_cast<ObjectIdxType>(current_->first);
        std::vector<WordIdxType> terms;
        generate_terms(source_->index_->termList(current_->second), &terms);
        const int64_t object_length = source_->document_lengths_[label];
        // exp(log(avg_document_length_) - log(object_length)), avg in WeightType (:161-163)
        const WeightType weight = static_cast<WeightType>(
            std::exp(static_cast<double>(std::log(source_->avg_document_length_)) - std::log(static_cast<double>(object_length))));
        create_instances(terms, label, weight, 1 /* stride */, instances);
        ++current_;
    }
    bool has_next() const override { return current_ != source_->document_id_mapping_.end(); }
    void reset() override { current_ = source_->document_id_mapping_.begin(); }

 private:
    void create_instances(const std::vector<WordIdxType>& tokens, ObjectIdxType object_id, WeightType weight, size_t stride,
                          InstancesT* instances) {
        std::deque<WordIdxType> buffer;
        const size_t w = source_->window_size_;
        for (const WordIdxType token : tokens) {
            buffer.push_back(token);
            if (buffer.size() == w) {
                const std::vector<WordIdxType> window(buffer.begin(), buffer.end());
                instances->emplace_back(window, source_->compute_term_weights(window), object_id, weight);
                for (size_t i = 0; i < stride; ++i) buffer.pop_front();
            }
        }
        if (buffer.size() == w) {
            const std::vector<WordIdxType> window(buffer.begin(), buffer.end());
            instances->emplace_back(window, source_->compute_term_weights(window), object_id, weight);
        }
    }
    IndexSource::DocumentIdMapping::iterator current_;
};

class StochasticInstanceGenerator : public InstanceGeneratorBase {            // :224-410
 public:
    StochasticInstanceGenerator(SamplingStrategy sampling_strategy, WeightingStrategy weighting_strategy, IndexSource* source, RNG* rng)
        : InstanceGeneratorBase(source), sampling_strategy_(sampling_strategy), weighting_strategy_(weighting_strategy), rng_(rng) {
        NVSM_CHECK(sampling_strategy != AUTOMATIC_SAMPLING);
        NVSM_CHECK(weighting_strategy != AUTOMATIC_WEIGHTING);
        size_t num_terms = 0, num_document_too_short = 0;
        NVSM_LOG(INFO) << "Loading documents into memory.";
        for (const auto& pair : source->document_id_mapping_) {
            const ObjectIdxType label = static_cast<ObjectIdxType>(pair.first);
            std::vector<WordIdxType>& terms = term_lists_[label];
            generate_terms(source->index_->termList(pair.second), &terms);
            if (terms.size() < source->window_size_) {
                NVSM_LOG(WARNING) << "Document " << pair.second << " only has " << terms.size() << " in-vocabulary tokens.";
                term_lists_.erase(label);
                ++num_document_too_short;
                continue;
            }
            num_terms += terms.size();
        }
        NVSM_LOG(INFO) << "Unable to generate n-grams for " << num_document_too_short << " documents as they were too short.";
        avg_document_length_ = num_terms / static_cast<double>(term_lists_.size());
        reset();
    }

    void generate(InstancesT* instances) override {
        const size_t num = std::min<size_t>(instance_order_.size(), 102400ul);
        const size_t w = source_->window_size_;
        std::vector<WordIdxType> buffer(w, 0);
        for (size_t i = 0; i < num; ++i) {
            const ObjectIdxType label = std::get<0>(instance_order_.front());
            const ObjectIdxType term_source_label = std::get<1>(instance_order_.front());
            const uint16_t position = std::get<2>(instance_order_.front());
            const std::vector<WordIdxType>& terms = term_lists_.at(term_source_label);
            std::copy(terms.begin() + position, terms.begin() + position + w, buffer.begin());
            const int64_t object_length = static_cast<int64_t>(term_lists_.at(label).size());
            WeightType weight = 1.0;
            if (weighting_strategy_ == INV_DOC_FREQUENCY)
                weight = static_cast<WeightType>(std::exp(std::log(avg_document_length_) - std::log(static_cast<double>(object_length))));
            instances->emplace_back(buffer, source_->compute_term_weights(buffer), label, weight);
            instance_order_.pop_front();
        }
    }

    bool fill(Batch* batch) override {
        const size_t w = source_->window_size_;
        if (term_ptr_.empty()) {             // label -> its term list / its length, indexed instead of looked up in the map
            ObjectIdxType max_label = 0;
            for (const auto& pair : term_lists_) max_label = std::max(max_label, pair.first);
            term_ptr_.assign(static_cast<size_t>(max_label) + 1, nullptr);
            for (const auto& pair : term_lists_) term_ptr_[static_cast<size_t>(pair.first)] = &pair.second;
        }
        const std::vector<WeightType>& table = source_->term_weight_table();
        std::vector<WeightType> fw(table.empty() ? 0 : w);
        while (!batch->full() && !instance_order_.empty()) {
            const ObjectIdxType label = std::get<0>(instance_order_.front());
            const ObjectIdxType term_source_label = std::get<1>(instance_order_.front());
            const uint16_t position = std::get<2>(instance_order_.front());
            const std::vector<WordIdxType>* terms = term_ptr_.at(static_cast<size_t>(term_source_label));
            const std::vector<WordIdxType>* own = term_ptr_.at(static_cast<size_t>(label));
            NVSM_CHECK(terms != nullptr && own != nullptr);
            const WordIdxType* window = terms->data() + position;
            WeightType weight = 1.0;
            if (weighting_strategy_ == INV_DOC_FREQUENCY)
                weight = static_cast<WeightType>(std::exp(std::log(avg_document_length_) - std::log(static_cast<double>(static_cast<int64_t>(own->size())))));
            if (!table.empty())
                for (size_t j = 0; j < w; ++j) fw[j] = table[static_cast<size_t>(window[j])];
            source_->push_window(window, table.empty() ? nullptr : fw.data(), label, weight, batch);
            instance_order_.pop_front();
        }
        return true;
    }

    bool has_next() const override { return !instance_order_.empty(); }

    void reset() override {
        if (!instance_order_.empty()) {
            NVSM_LOG(WARNING) << "Resetting instance generator while there are still instances to consume.";
            instance_order_.clear();
        }
        // For NGRAM_FREQUENCY resampling (:305-309)
        const long num_samples = std::max<long>(
            static_cast<long>(std::ceil(avg_document_length_ - static_cast<double>(source_->window_size_) + 1)), 1l);
        if (sampling_strategy_ == NONE) NVSM_LOG(INFO) << "Generating instance pointers.";
        else NVSM_LOG(INFO) << "Generating instance pointers (" << num_samples << " samples per document).";

        for (const auto& pair : term_lists_) {
            const ObjectIdxType label = pair.first;
            const int document_length = static_cast<int>(pair.second.size());
            // the index document length includes stopped / out-of-vocabulary positions (:337-339)
            NVSM_CHECK(source_->document_lengths_.at(label) >= document_length);
            const long max_position = document_length - static_cast<long>(source_->window_size_) + 1;
            if (sampling_strategy_ == NONE) {
                if (document_length >= (1 << 16)) {
                    NVSM_LOG(WARNING) << "Skipping instance generation from object " << label << " as it exceeds 2^16 terms ("
                                      << document_length << ").";
                    continue;
                }
                for (long position = 0; position < max_position; position += 1 /* stride */)
                    instance_order_.emplace_back(label, label, static_cast<uint16_t>(position));
            } else if (sampling_strategy_ == NGRAM_FREQUENCY) {
                std::uniform_int_distribution<int> term_position_distribution(0, static_cast<int>(max_position - 1));
                for (long i = 0; i < num_samples; ++i)
                    instance_order_.emplace_back(label, label, static_cast<uint16_t>(term_position_distribution(*rng_)));
            } else {
                NVSM_LOG(FATAL) << "Invalid sampling strategy: " << sampling_strategy_;
            }
        }
        NVSM_LOG(INFO) << "Shuffling " << instance_order_.size() << " instance pointers.";
        shuffle_one_draw_per_element(instance_order_.begin(), instance_order_.end(), *rng_);   // std::shuffle, :404
    }

 private:
    const SamplingStrategy sampling_strategy_;
    const WeightingStrategy weighting_strategy_;
    double avg_document_length_ = 0.0;
    std::map<ObjectIdxType, std::vector<WordIdxType>> term_lists_;
    std::vector<const std::vector<WordIdxType>*> term_ptr_;
    std::deque<std::tuple<ObjectIdxType, ObjectIdxType, uint16_t>> instance_order_;
    RNG* const rng_;
};

// ---------------------------------------------------------------------------------------------
// IndexSource — cpp/data_indri.cpp:412-915
// ---------------------------------------------------------------------------------------------
IndexSource::IndexSource(IndexInterface* index, size_t window_size, RNG* rng, size_t max_vocabulary_size,
                         size_t min_document_frequency, size_t max_document_frequency, size_t documents_cutoff, bool include_oov,
                         bool include_digits, const std::vector<std::string>* document_list, const TermBlacklist* term_blacklist,
                         bool shuffle, SamplingStrategy sampling_strategy, WeightingStrategy weighting_strategy,
                         TermWeightingStrategy term_weighting_strategy)
    : DataSource(0, 0), index_(index), window_size_(window_size), include_oov_(include_oov),
      term_weighting_strategy_(term_weighting_strategy) {
    initialize(max_vocabulary_size, min_document_frequency, max_document_frequency, include_digits, documents_cutoff, shuffle,
               sampling_strategy, weighting_strategy, document_list, term_blacklist, rng);
}

IndexSource::~IndexSource() {}

void IndexSource::reset() {
    instance_generator_->reset();
    num_terms_emitted_ = 0;
}

void IndexSource::next(Batch* batch) {                                          // :499-523
    NVSM_CHECK(!term_id_mapping_.empty());
    NVSM_CHECK(batch->window_size() == window_size_);
    DataSource::next(batch);
    if (overflow_empty() && instance_generator_->fill(batch)) return;
    InstancesT instances;
    while (!batch->full() && has_next()) {
        instance_generator_->generate(&instances);
        while (!instances.empty()) {
            const InstanceT& inst = instances.front();
            push_instance(std::get<0>(inst), std::get<1>(inst), std::get<2>(inst), std::get<3>(inst), batch);
            instances.pop_front();
        }
    }
}

bool IndexSource::has_next() const { return DataSource::has_next() || instance_generator_->has_next(); }

void IndexSource::extract_metadata(Metadata* metadata) const {                  // :530-551
    for (const auto& pair : term_id_mapping_) {
        Metadata::TermInfo t;
        t.index_term_id = static_cast<int32_t>(pair.first);
        t.model_term_id = static_cast<int32_t>(pair.second);
        t.term_frequency = static_cast<int32_t>(inv_term_id_to_term_freq_.at(pair.second));
        metadata->term.push_back(t);
    }
    metadata->total_terms = static_cast<int32_t>(total_num_terms_);
    for (const auto& pair : document_id_mapping_) {
        Metadata::ObjectInfo o;
        o.model_object_id = static_cast<int32_t>(pair.first);
        o.index_object_id = static_cast<int32_t>(pair.second);
        metadata->object.push_back(o);
    }
}

std::map<std::string, int64_t> IndexSource::build_term_identifiers_map() const {       // :553-569
    std::map<std::string, int64_t> m;
    for (const auto& pair : term_id_mapping_) {
        const bool inserted = m.insert({index_->term(pair.first), static_cast<int64_t>(pair.second)}).second;
        NVSM_CHECK(inserted);
    }
    return m;
}

std::map<std::string, int64_t> IndexSource::build_document_identifiers_map() const {   // :571-589
    std::map<std::string, int64_t> m;
    for (const auto& pair : document_id_mapping_) {
        const bool inserted = m.insert({index_->docno(pair.second), static_cast<int64_t>(pair.first)}).second;
        NVSM_CHECK(inserted);
    }
    return m;
}

// :591-620. The reference tests `document_id_mapping_` membership with the INDEX document id although the map is
// keyed by MODEL ids; kept as is (it only matters under --document_cutoff / --document_list), counted here from the
// term lists instead of the inverted file.
size_t IndexSource::compute_term_frequency(TERMID_T term_id) {
    if (!restricted_built_) {
        const DOCID_T lo = index_->documentBase(), hi = index_->documentMaximum();
        for (DOCID_T d = lo; d < hi; ++d) {
            if (document_id_mapping_.find(static_cast<size_t>(d)) == document_id_mapping_.end()) continue;
            for (const TERMID_T t : index_->termList(d)) restricted_term_frequency_[t] += 1;
        }
        restricted_built_ = true;
    }
    const auto it = restricted_term_frequency_.find(term_id);
    return it == restricted_term_frequency_.end() ? 0 : it->second;
}

void IndexSource::initialize(size_t max_vocabulary_size, size_t min_document_frequency, size_t max_document_frequency,
                             bool include_digits, size_t documents_cutoff, bool shuffle, SamplingStrategy sampling_strategy,
                             WeightingStrategy weighting_strategy, const std::vector<std::string>* document_list,
                             const TermBlacklist* term_blacklist, RNG* rng) {
    NVSM_CHECK(index_.get() != nullptr);
    if (sampling_strategy == AUTOMATIC_SAMPLING) sampling_strategy = shuffle ? NGRAM_FREQUENCY : NONE;                 // :657-659
    if (weighting_strategy == AUTOMATIC_WEIGHTING) weighting_strategy = sampling_strategy == NONE ? INV_DOC_FREQUENCY : UNIFORM;

    // ---- documents (:665-743) ----
    {
        NVSM_LOG(INFO) << "Building document-id mapping for Indri.";
        const size_t document_count = index_->documentCount();
        const size_t document_list_size = (document_list == nullptr) ? document_count : document_list->size();
        const size_t num_documents = std::min(std::min(documents_cutoff > 0 ? documents_cutoff : document_count, document_count),
                                              document_list_size);
        document_lengths_.assign(num_documents, 0);
        size_t document_length_agg = 0, model_doc_id = 0, discarded_documents = 0;
        auto consider = [&](DOCID_T index_doc_id) {
            const int64_t document_length = index_->documentLength(index_doc_id);
            if (document_length >= static_cast<int64_t>(window_size_)) {
                document_id_mapping_.insert(std::make_pair(model_doc_id, index_doc_id));
                document_lengths_[model_doc_id] = document_length;
                document_length_agg += document_length;
                ++model_doc_id;
            } else {
                ++discarded_documents;
            }
        };
        if (document_list == nullptr) {
            DOCID_T index_doc_id = index_->documentBase();
            const DOCID_T max_doc_id = index_->documentMaximum();
            while (document_id_mapping_.size() < num_documents && index_doc_id < max_doc_id) {
                consider(index_doc_id);
                ++index_doc_id;
            }
        } else {
            const std::vector<DOCID_T> int_doc_ids = index_->documentIDsFromDocno(*document_list);
            NVSM_CHECK(int_doc_ids.size() == document_list->size());
            for (const DOCID_T index_doc_id : int_doc_ids) {
                if (document_id_mapping_.size() >= num_documents) break;
                consider(index_doc_id);
            }
        }
        NVSM_LOG(INFO) << "Discarded " << discarded_documents << " documents which were too short.";
        NVSM_CHECK(document_id_mapping_.size() <= num_documents);
        corpus_size_ = document_id_mapping_.size();
        NVSM_CHECK(corpus_size_ > 0);
        avg_document_length_ = document_length_agg / static_cast<WeightType>(document_id_mapping_.size());
        NVSM_CHECK(avg_document_length_ > 0.0);
    }

    // ---- vocabulary (:749-869) ----
    {
        NVSM_LOG(INFO) << "Building term-id mapping for Indri.";
        size_t num_terms = 0;
        const size_t corpus_unique_term = index_->uniqueTermCount() + 1;
        typedef std::pair<int64_t, size_t> TermInfo;                           // (collection frequency, index term id)
        std::priority_queue<TermInfo, std::vector<TermInfo>, std::greater<TermInfo>> pq;
        size_t discarded_zero = 0, discarded_blacklist = 0, discarded_digits = 0, discarded_df_high = 0, discarded_df_low = 0;
        for (const VocabularyEntry& entry : index_->vocabulary()) {
            if (entry.term_id == 0) { ++discarded_zero; continue; }
            if (!include_digits && is_number(entry.term)) { ++discarded_digits; continue; }
            if (min_document_frequency > 0 && entry.document_count < min_document_frequency) { ++discarded_df_low; continue; }
            if (max_document_frequency > 0 && entry.document_count > max_document_frequency) { ++discarded_df_high; continue; }
            if (term_blacklist != nullptr && term_blacklist->count(entry.term)) { ++discarded_blacklist; continue; }
            const int64_t frequency = static_cast<int64_t>(entry.total_count);
            NVSM_CHECK(frequency > 0);
            if (max_vocabulary_size && (corpus_unique_term > max_vocabulary_size)) {
                if (pq.size() >= max_vocabulary_size && pq.top().first < frequency) pq.pop();
                if (pq.size() < max_vocabulary_size) pq.push(std::make_pair(frequency, static_cast<size_t>(entry.term_id)));
            } else {
                pq.push(std::make_pair(frequency, static_cast<size_t>(entry.term_id)));
            }
        }
        if (max_vocabulary_size) NVSM_CHECK(pq.size() <= max_vocabulary_size);
        if (include_oov_) {
            term_id_mapping_.insert(std::make_pair(0, 0));
            inv_term_id_mapping_.insert(std::make_pair(0, 0));
            inv_term_id_to_term_freq_.insert(std::make_pair(0, 1));
        }
        while (!pq.empty()) {                                                   // ascending (frequency, term id): rare terms get low ids
            const size_t index_term_id = pq.top().second;
            const size_t our_term_id = term_id_mapping_.size();
            size_t frequency = 0;
            if (corpus_size() == index_->documentCount()) frequency = static_cast<size_t>(pq.top().first);
            else frequency = compute_term_frequency(static_cast<TERMID_T>(index_term_id));
            pq.pop();
            if (frequency == 0) continue;
            num_terms += frequency;
            term_id_mapping_.insert(std::make_pair(static_cast<TERMID_T>(index_term_id), our_term_id));
            inv_term_id_mapping_.insert(std::make_pair(our_term_id, static_cast<TERMID_T>(index_term_id)));
            inv_term_id_to_term_freq_.insert(std::make_pair(our_term_id, static_cast<int64_t>(frequency)));
        }
        NVSM_LOG(INFO) << "Vocabulary filtering discarded " << discarded_zero << " meta-terms, " << discarded_blacklist
                       << " blacklisted terms, " << discarded_digits << " terms that contained a digit, " << discarded_df_high
                       << " terms that had too high document frequency, " << discarded_df_low
                       << " terms that had too low document frequency.";
        vocabulary_size_ = term_id_mapping_.size();
        NVSM_CHECK(num_terms > 0);
        total_num_terms_ = num_terms;
    }
    const double log_ratio = std::log10(static_cast<double>(total_num_terms_)) - std::log10(static_cast<double>(vocabulary_size_));
    NVSM_LOG(INFO) << "Index contains " << vocabulary_size_ << " unique terms and " << total_num_terms_
                   << " term occurrences (log-ratio=" << log_ratio << ").";

    if (!shuffle) {
        NVSM_CHECK(sampling_strategy == NONE);
        instance_generator_.reset(new SequentialInstanceGenerator(this));
    } else {
        instance_generator_.reset(new StochasticInstanceGenerator(sampling_strategy, weighting_strategy, this, rng));
    }
}

// compute_term_weights per model term id, evaluated once with that very function (empty under uniform term weighting)
const std::vector<WeightType>& IndexSource::term_weight_table() const {
    if (term_weighting_strategy_ != UNIFORM_TERM_WEIGHTING && term_weight_table_.empty()) {
        size_t n = 0;
        for (const auto& pair : inv_term_id_to_term_freq_) n = std::max(n, static_cast<size_t>(pair.first) + 1);
        term_weight_table_.assign(n, static_cast<WeightType>(0));
        for (const auto& pair : inv_term_id_to_term_freq_)
            term_weight_table_[pair.first] = compute_term_weights({static_cast<WordIdxType>(pair.first)})[0];
    }
    return term_weight_table_;
}

std::vector<WeightType> IndexSource::compute_term_weights(const std::vector<WordIdxType>& terms) const {
    if (term_weighting_strategy_ == UNIFORM_TERM_WEIGHTING) return std::vector<WeightType>();
    std::vector<WeightType> weights;
    for (const WordIdxType term_id : terms) {
        const WeightType self_information = -std::log(
            static_cast<WeightType>(inv_term_id_to_term_freq_.at(static_cast<size_t>(term_id))) / total_num_terms_);
        weights.push_back(self_information);
    }
    return weights;
}

int64_t IndexSource::term_id(const std::string& term) const {
    const TERMID_T index_term_id = index_->term(term);
    const auto it = term_id_mapping_.find(index_term_id);
    return it == term_id_mapping_.end() ? -1 : static_cast<int64_t>(it->second);
}

std::string IndexSource::term(int64_t model_term_id) const {
    return index_->term(inv_term_id_mapping_.at(static_cast<size_t>(model_term_id)));
}

}  // namespace nvsm_host
