// IndexSource — the reference's TextEntity::IndriSource (include/cuNVSM/data.h:383-529, cpp/data_indri.cpp:107-915)
// over IndexInterface: document selection, top-K vocabulary by collection frequency, and the two instance
// generators (sequential windows; stochastic = all term lists in memory, positions sampled per document, shuffled).
// Every std:: random facility is the one the reference calls (minstd_rand0, uniform_int_distribution<int>,
// std::shuffle on a deque), so with libstdc++ the instance order replays the reference's for a given seed — the
// seed-1 expectations of cpp/data_tests.cpp:476-590 are checked in tests/cpp/host_tests.cpp.
#pragma once

#include <map>
#include <memory>
#include <set>

#include "data.hpp"
#include "index.hpp"

namespace nvsm_host {

enum SamplingStrategy { AUTOMATIC_SAMPLING, NONE, NGRAM_FREQUENCY };                 // data.h:371-373
enum WeightingStrategy { AUTOMATIC_WEIGHTING, UNIFORM, INV_DOC_FREQUENCY };          // :375-377
enum TermWeightingStrategy { UNIFORM_TERM_WEIGHTING, SELF_INFORMATION_TERM_WEIGHTING };  // :379-381

class InstanceGeneratorBase;

class IndexSource : public DataSource {
 public:
    typedef std::map<TERMID_T, size_t> TermIdMapping;
    typedef std::map<size_t, DOCID_T> DocumentIdMapping;
    typedef std::set<std::string> TermBlacklist;

    // Takes ownership of `index` (as IndriSource(DiskIndex*, ...) does, data.h:405-418).
    IndexSource(IndexInterface* index, size_t window_size, RNG* rng, size_t max_vocabulary_size = 0,
                size_t min_document_frequency = 0, size_t max_document_frequency = 0, size_t documents_cutoff = 0,
                bool include_oov = false, bool include_digits = false,
                const std::vector<std::string>* document_list = nullptr, const TermBlacklist* term_blacklist = nullptr,
                bool shuffle = false, SamplingStrategy sampling_strategy = AUTOMATIC_SAMPLING,
                WeightingStrategy weighting_strategy = AUTOMATIC_WEIGHTING,
                TermWeightingStrategy term_weighting_strategy = UNIFORM_TERM_WEIGHTING);
    ~IndexSource() override;

    void reset() override;
    void next(Batch* batch) override;
    bool has_next() const override;
    float progress() const override { return static_cast<float>(num_terms_emitted_ / static_cast<double>(total_num_terms_)); }
    void extract_metadata(Metadata* metadata) const override;

    int64_t term_id(const std::string& term) const;            // model term id or -1
    std::string term(int64_t model_term_id) const;
    const TermIdMapping& term_id_mapping() const { return term_id_mapping_; }
    const DocumentIdMapping& document_id_mapping() const { return document_id_mapping_; }
    const std::map<size_t, int64_t>& term_frequencies() const { return inv_term_id_to_term_freq_; }
    size_t window_size() const { return window_size_; }
    size_t total_num_terms() const { return total_num_terms_; }
    std::map<std::string, int64_t> build_term_identifiers_map() const;
    std::map<std::string, int64_t> build_document_identifiers_map() const;

    std::vector<WeightType> compute_term_weights(const std::vector<WordIdxType>& terms) const;   // data.h:465-490
    const std::vector<WeightType>& term_weight_table() const;

 private:
    friend class InstanceGeneratorBase;
    friend class SequentialInstanceGenerator;
    friend class StochasticInstanceGenerator;
    void initialize(size_t max_vocabulary_size, size_t min_document_frequency, size_t max_document_frequency,
                    bool include_digits, size_t documents_cutoff, bool shuffle, SamplingStrategy sampling_strategy,
                    WeightingStrategy weighting_strategy, const std::vector<std::string>* document_list,
                    const TermBlacklist* term_blacklist, RNG* rng);
    size_t compute_term_frequency(TERMID_T term_id);

    std::unique_ptr<IndexInterface> index_;
    const size_t window_size_;
    const bool include_oov_;
    size_t num_terms_emitted_ = 0;
    size_t total_num_terms_ = 0;
    WeightType avg_document_length_ = 0;
    TermIdMapping term_id_mapping_;
    std::map<size_t, TERMID_T> inv_term_id_mapping_;
    std::map<size_t, int64_t> inv_term_id_to_term_freq_;
    mutable std::vector<WeightType> term_weight_table_;
    std::vector<int64_t> document_lengths_;
    DocumentIdMapping document_id_mapping_;
    std::map<TERMID_T, size_t> restricted_term_frequency_;     // lazily built when the corpus is a subset of the index
    bool restricted_built_ = false;
    std::unique_ptr<InstanceGeneratorBase> instance_generator_;
    const TermWeightingStrategy term_weighting_strategy_;
};

}  // namespace nvsm_host
