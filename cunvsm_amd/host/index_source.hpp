// IndexSource — the data source of cuNVSMTrainModel: n-gram windows of the documents of an index, as
// TextEntity::Batch rows. Same constructor arguments, vocabulary / document selection and — given the same seed — the
// same instances in the same order as the reference's TextEntity::IndriSource (include/cuNVSM/data.h:383-529,
// cpp/data_indri.cpp:412-915), which its own tests pin (cpp/data_tests.cpp:365-683, restated in
// tests/cpp/host_tests.cpp). Built differently:
//   * the collection is mapped to model term ids ONCE into a flat token arena (32-bit ids + one offset per document)
//     through a dense index-term → model-term table;
//   * an epoch is a flat array of (document, position) references — all windows in document order, all windows
//     shuffled, or `samples per document` sampled positions shuffled — consumed by a cursor;
//   * next() copies windows from the arena straight into the batch's rows. There is no per-instance object, no
//     intermediate queue and no generator class hierarchy.
// What has to be the reference's is the random stream: one uniform_int_distribution<int> draw per sampled position, in
// document order, then the one-draw-per-element shuffle libstdc++ shipped before GCC 7 (the permutation the reference's
// seed-pinned test expects), all on the caller's minstd_rand0.
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

    void reset() override;                 // next epoch: re-draws / re-shuffles with the shared generator
    void next(Batch* batch) override;
    bool has_next() const override;
    float progress() const override;       // windows handed out / windows of this epoch
    void extract_metadata(Metadata* metadata) const override;

    int64_t term_id(const std::string& term) const;            // model term id or -1
    std::string term(int64_t model_term_id) const;
    const TermIdMapping& term_id_mapping() const { return model_term_of_; }
    const DocumentIdMapping& document_id_mapping() const { return index_doc_of_; }
    const std::map<size_t, int64_t>& term_frequencies() const { return frequency_of_model_term_; }
    size_t window_size() const { return window_; }
    size_t total_num_terms() const { return corpus_term_occurrences_; }
    std::map<std::string, int64_t> build_term_identifiers_map() const;
    std::map<std::string, int64_t> build_document_identifiers_map() const;

    // −log(term frequency / total term occurrences) per term; empty under uniform term weighting (data.h:465-490)
    std::vector<WeightType> compute_term_weights(const std::vector<WordIdxType>& terms) const;

 private:
    class WindowFeeder;                    // index_source.cpp
    void choose_documents(size_t documents_cutoff, const std::vector<std::string>* document_list);
    void choose_vocabulary(size_t max_vocabulary_size, size_t min_document_frequency, size_t max_document_frequency,
                           bool include_digits, const TermBlacklist* term_blacklist);
    size_t occurrences_in_chosen_documents(TERMID_T term_id);

    std::unique_ptr<IndexInterface> index_;
    const size_t window_;
    const bool oov_token_;                 // out-of-vocabulary positions become model term 0 instead of being dropped
    const TermWeightingStrategy term_weighting_;
    size_t corpus_term_occurrences_ = 0;
    WeightType mean_index_document_length_ = 0;
    TermIdMapping model_term_of_;                          // index term id → model term id
    std::map<size_t, TERMID_T> index_term_of_;             // model term id → index term id
    std::map<size_t, int64_t> frequency_of_model_term_;
    DocumentIdMapping index_doc_of_;                       // model document id → index document id
    std::vector<int64_t> index_document_length_;           // by model document id
    std::map<TERMID_T, size_t> occurrences_;               // term occurrences inside the chosen documents (built on demand)
    bool occurrences_counted_ = false;
    std::unique_ptr<WindowFeeder> feeder_;
};

}  // namespace nvsm_host
