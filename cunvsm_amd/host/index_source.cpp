#include "index_source.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace nvsm_host {

namespace {

// One window of one document: where an instance's features start in the token arena.
struct WindowRef {
    uint32_t doc;        // model document id (= the instance's label)
    uint32_t pos;        // first token of the window inside the document
};

// The permutation std::shuffle produced in libstdc++ up to GCC 6: element i (from the second on) is swapped with the
// element at a position drawn uniformly from [0, i], one generator-backed draw per element. The reference's seed-pinned
// expectation (cpp/data_tests.cpp:571-575, "relies on seed == 1") was recorded with that library; GCC 7 and later pair
// up two positions per generator call and permute differently, so std::shuffle of today cannot be used.
void shuffle_pre_gcc7(std::vector<WindowRef>* v, RNG* rng) {
    typedef std::uniform_int_distribution<uint64_t> Draw;
    Draw draw;
    for (size_t i = 1; i < v->size(); ++i) std::swap((*v)[i], (*v)[draw(*rng, Draw::param_type(0, i))]);
}

// Keeps the `capacity` most frequent terms seen so far under the reference's replacement rule (cpp/data_indri.cpp:
// 821-841): a newcomer displaces the current minimum — minimum by (frequency, index term id) — only if it is strictly
// more frequent, so among equally frequent terms at the cut the earlier arrivals stay, and among the survivors the one
// with the smallest id goes first. capacity 0 = unbounded.
class FrequentTerms {
 public:
    explicit FrequentTerms(size_t capacity) : capacity_(capacity) {}
    void offer(int64_t frequency, TERMID_T term) {
        const Entry e(frequency, term);
        if (capacity_ == 0) { kept_.insert(e); return; }
        if (kept_.size() >= capacity_ && kept_.begin()->first < frequency) kept_.erase(kept_.begin());
        if (kept_.size() < capacity_) kept_.insert(e);
    }
    // ascending (frequency, term id): rare terms first — the order model term ids are handed out in
    const std::set<std::pair<int64_t, TERMID_T>>& ascending() const { return kept_; }
 private:
    typedef std::pair<int64_t, TERMID_T> Entry;
    const size_t capacity_;
    std::set<Entry> kept_;
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// WindowFeeder: the collection as model term ids + the order in which this epoch's windows are handed out
// ---------------------------------------------------------------------------------------------------------------------
class IndexSource::WindowFeeder {
 public:
    enum Order { DOCUMENT_ORDER, ALL_WINDOWS_SHUFFLED, SAMPLED_POSITIONS_SHUFFLED };

    WindowFeeder(IndexSource* owner, Order order, bool inverse_length_weights, RNG* rng)
        : owner_(owner), order_(order), rng_(rng), window_(owner->window_) {
        load_documents();
        assign_weights(inverse_length_weights);
        if (owner->term_weighting_ != UNIFORM_TERM_WEIGHTING) {
            // the weight of a term depends on the term alone: evaluated once per model term with the public function
            size_t terms = 0;
            for (const auto& tf : owner->frequency_of_model_term_) terms = std::max(terms, tf.first + 1);
            term_weight_.assign(terms, static_cast<WeightType>(0));
            for (const auto& tf : owner->frequency_of_model_term_)
                term_weight_[tf.first] = owner->compute_term_weights({static_cast<WordIdxType>(tf.first)})[0];
        }
        start_epoch();
    }

    bool pending() const { return cursor_ < epoch_.size(); }
    double fraction_done() const { return epoch_.empty() ? 1.0 : static_cast<double>(cursor_) / static_cast<double>(epoch_.size()); }

    void start_epoch() {
        if (pending()) NVSM_LOG(WARNING) << "Resetting instance generator while there are still instances to consume.";
        cursor_ = 0;
        if (order_ == DOCUMENT_ORDER) {
            if (epoch_.empty()) plan_every_window(false);          // the same plan every epoch
            return;
        }
        epoch_.clear();
        if (order_ == ALL_WINDOWS_SHUFFLED) {
            NVSM_LOG(INFO) << "Generating instance pointers.";
            plan_every_window(true);
        } else {
            plan_sampled_windows();
        }
        NVSM_LOG(INFO) << "Shuffling " << epoch_.size() << " instance pointers.";
        shuffle_pre_gcc7(&epoch_, rng_);                           // cpp/data_indri.cpp:404
    }

    // copies windows into the batch's free rows until it is full or the epoch is over
    void feed(Batch* batch) {
        std::vector<WordIdxType> ids(window_);
        std::vector<WeightType> per_term(term_weight_.empty() ? 0 : window_);
        while (!batch->full() && pending()) {
            const WindowRef ref = epoch_[cursor_++];
            const int32_t* src = arena_.data() + first_token_[ref.doc] + ref.pos;
            for (size_t j = 0; j < window_; ++j) ids[j] = static_cast<WordIdxType>(src[j]);
            if (!term_weight_.empty())
                for (size_t j = 0; j < window_; ++j) per_term[j] = term_weight_[static_cast<size_t>(src[j])];
            owner_->push_window(ids.data(), term_weight_.empty() ? nullptr : per_term.data(), static_cast<ObjectIdxType>(ref.doc),
                                instance_weight_[ref.doc], batch);
        }
    }

 private:
    size_t tokens_of(size_t doc) const { return static_cast<size_t>(first_token_[doc + 1] - first_token_[doc]); }

    // every chosen document's term list → model term ids (cpp/data_indri.cpp:112-136: unknown terms become the OoV token 0
    // or are dropped). The shuffled orders leave out documents with fewer in-vocabulary tokens than a window (:258-276).
    void load_documents() {
        TERMID_T max_index_term = 0;
        for (const auto& m : owner_->model_term_of_) max_index_term = std::max(max_index_term, m.first);
        std::vector<int32_t> model_term(static_cast<size_t>(max_index_term) + 1, -1);
        for (const auto& m : owner_->model_term_of_) model_term[static_cast<size_t>(m.first)] = static_cast<int32_t>(m.second);

        const size_t docs = owner_->index_doc_of_.empty() ? 0 : owner_->index_doc_of_.rbegin()->first + 1;
        first_token_.assign(docs + 1, 0);
        usable_.assign(docs, 0);
        if (order_ != DOCUMENT_ORDER) NVSM_LOG(INFO) << "Loading documents into memory.";
        size_t too_short = 0, usable_tokens = 0, usable_docs = 0;
        size_t next_doc = 0;
        for (const auto& d : owner_->index_doc_of_) {              // ascending model document id
            for (; next_doc < d.first; ++next_doc) first_token_[next_doc + 1] = arena_.size();
            const size_t before = arena_.size();
            for (const TERMID_T t : owner_->index_->termList(d.second)) {
                const int32_t m = (t >= 0 && static_cast<size_t>(t) < model_term.size()) ? model_term[static_cast<size_t>(t)] : -1;
                if (m >= 0) arena_.push_back(m);
                else if (owner_->oov_token_) arena_.push_back(0);
            }
            size_t n = arena_.size() - before;
            if (order_ != DOCUMENT_ORDER && n < window_) {
                NVSM_LOG(WARNING) << "Document " << d.second << " only has " << n << " in-vocabulary tokens.";
                arena_.resize(before);
                n = 0;
                ++too_short;
            } else {
                // the index's document length counts stopped / out-of-vocabulary positions too (:337-339)
                NVSM_CHECK(owner_->index_document_length_.at(d.first) >= static_cast<int64_t>(n));
                usable_[d.first] = 1;
                usable_tokens += n;
                ++usable_docs;
            }
            first_token_[d.first + 1] = arena_.size();
            next_doc = d.first + 1;
        }
        if (order_ != DOCUMENT_ORDER) {
            NVSM_LOG(INFO) << "Unable to generate n-grams for " << too_short << " documents as they were too short.";
            mean_tokens_ = static_cast<double>(usable_tokens) / static_cast<double>(usable_docs);
        }
    }

    void assign_weights(bool inverse_length_weights) {
        instance_weight_.assign(usable_.size(), static_cast<WeightType>(1));
        for (size_t doc = 0; doc < usable_.size(); ++doc) {
            if (!usable_[doc]) continue;
            if (order_ == DOCUMENT_ORDER) {
                // exp(log(mean index length, in WeightType) − log(index length)) (:161-163), whatever the weighting strategy
                instance_weight_[doc] = static_cast<WeightType>(std::exp(static_cast<double>(std::log(owner_->mean_index_document_length_)) -
                                                                         std::log(static_cast<double>(owner_->index_document_length_[doc]))));
            } else if (inverse_length_weights) {
                // the same ratio over in-vocabulary tokens, in double (:357-361)
                instance_weight_[doc] = static_cast<WeightType>(std::exp(std::log(mean_tokens_) - std::log(static_cast<double>(static_cast<int64_t>(tokens_of(doc))))));
            }
        }
    }

    void plan_every_window(bool positions_are_16_bit) {
        for (size_t doc = 0; doc < usable_.size(); ++doc) {
            if (!usable_[doc]) continue;
            const size_t n = tokens_of(doc);
            if (positions_are_16_bit && n >= (size_t(1) << 16)) {   // the reference stores positions in a uint16_t (:344-351)
                NVSM_LOG(WARNING) << "Skipping instance generation from object " << doc << " as it exceeds 2^16 terms (" << n << ").";
                continue;
            }
            if (n < window_) continue;
            for (size_t pos = 0; pos + window_ <= n; ++pos) epoch_.push_back(WindowRef{static_cast<uint32_t>(doc), static_cast<uint32_t>(pos)});
        }
    }

    // NGRAM_FREQUENCY (:305-309,364-378): the same number of windows from every document — as many as an average document
    // has — at positions drawn uniformly, documents in ascending id order, one int distribution per document
    void plan_sampled_windows() {
        const long samples = std::max<long>(static_cast<long>(std::ceil(mean_tokens_ - static_cast<double>(window_) + 1)), 1l);
        NVSM_LOG(INFO) << "Generating instance pointers (" << samples << " samples per document).";
        for (size_t doc = 0; doc < usable_.size(); ++doc) {
            if (!usable_[doc]) continue;
            const long last = static_cast<long>(tokens_of(doc)) - static_cast<long>(window_);
            std::uniform_int_distribution<int> position(0, static_cast<int>(last));
            for (long s = 0; s < samples; ++s)                       // narrowed to 16 bits as the reference's tuple does
                epoch_.push_back(WindowRef{static_cast<uint32_t>(doc), static_cast<uint32_t>(static_cast<uint16_t>(position(*rng_)))});
        }
    }

    IndexSource* const owner_;
    const Order order_;
    RNG* const rng_;
    const size_t window_;
    std::vector<int32_t> arena_;               // model term ids of all usable documents, back to back
    std::vector<uint64_t> first_token_;        // [documents + 1] offsets into arena_
    std::vector<uint8_t> usable_;              // document takes part in this source's epochs
    std::vector<WeightType> instance_weight_;  // per document
    std::vector<WeightType> term_weight_;      // per model term; empty = all 1
    double mean_tokens_ = 0.0;                 // in-vocabulary tokens per usable document (shuffled orders)
    std::vector<WindowRef> epoch_;
    size_t cursor_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------------
// IndexSource
// ---------------------------------------------------------------------------------------------------------------------
IndexSource::IndexSource(IndexInterface* index, size_t window_size, RNG* rng, size_t max_vocabulary_size,
                         size_t min_document_frequency, size_t max_document_frequency, size_t documents_cutoff, bool include_oov,
                         bool include_digits, const std::vector<std::string>* document_list, const TermBlacklist* term_blacklist,
                         bool shuffle, SamplingStrategy sampling_strategy, WeightingStrategy weighting_strategy,
                         TermWeightingStrategy term_weighting_strategy)
    : DataSource(0, 0), index_(index), window_(window_size), oov_token_(include_oov), term_weighting_(term_weighting_strategy) {
    NVSM_CHECK(index_.get() != nullptr);
    if (sampling_strategy == AUTOMATIC_SAMPLING) sampling_strategy = shuffle ? NGRAM_FREQUENCY : NONE;                 // :657-659
    if (weighting_strategy == AUTOMATIC_WEIGHTING) weighting_strategy = sampling_strategy == NONE ? INV_DOC_FREQUENCY : UNIFORM;
    choose_documents(documents_cutoff, document_list);
    choose_vocabulary(max_vocabulary_size, min_document_frequency, max_document_frequency, include_digits, term_blacklist);
    WindowFeeder::Order order = WindowFeeder::DOCUMENT_ORDER;
    if (shuffle) order = sampling_strategy == NONE ? WindowFeeder::ALL_WINDOWS_SHUFFLED : WindowFeeder::SAMPLED_POSITIONS_SHUFFLED;
    else NVSM_CHECK(sampling_strategy == NONE);                    // sampling positions without shuffling is refused (:873-875)
    feeder_.reset(new WindowFeeder(this, order, weighting_strategy == INV_DOC_FREQUENCY, rng));
}

IndexSource::~IndexSource() {}

void IndexSource::reset() { feeder_->start_epoch(); }

void IndexSource::next(Batch* batch) {
    NVSM_CHECK(!model_term_of_.empty());
    NVSM_CHECK(batch->window_size() == window_);
    DataSource::next(batch);               // nothing is ever parked in the base class's overflow queue by this source
    feeder_->feed(batch);
}

bool IndexSource::has_next() const { return DataSource::has_next() || feeder_->pending(); }

float IndexSource::progress() const { return static_cast<float>(feeder_->fraction_done()); }

// which documents: the first `documents_cutoff` of the index (or of --document_list) that are at least a window long;
// model document ids are handed out in that order (:665-743)
void IndexSource::choose_documents(size_t documents_cutoff, const std::vector<std::string>* document_list) {
    NVSM_LOG(INFO) << "Building document-id mapping for Indri.";
    const size_t in_index = index_->documentCount();
    size_t budget = documents_cutoff > 0 ? std::min(documents_cutoff, in_index) : in_index;
    std::vector<DOCID_T> candidates;
    if (document_list != nullptr) {
        budget = std::min(budget, document_list->size());
        candidates = index_->documentIDsFromDocno(*document_list);
        NVSM_CHECK(candidates.size() == document_list->size());
    }
    index_document_length_.assign(budget, 0);
    size_t length_sum = 0, too_short = 0;
    auto take = [&](DOCID_T index_doc) {
        const int64_t length = index_->documentLength(index_doc);
        if (length < static_cast<int64_t>(window_)) { ++too_short; return; }
        const size_t model_doc = index_doc_of_.size();
        index_doc_of_.emplace(model_doc, index_doc);
        index_document_length_[model_doc] = length;
        length_sum += static_cast<size_t>(length);
    };
    if (document_list == nullptr) {
        for (DOCID_T d = index_->documentBase(), end = index_->documentMaximum(); d < end && index_doc_of_.size() < budget; ++d) take(d);
    } else {
        for (size_t i = 0; i < candidates.size() && index_doc_of_.size() < budget; ++i) take(candidates[i]);
    }
    NVSM_LOG(INFO) << "Discarded " << too_short << " documents which were too short.";
    corpus_size_ = index_doc_of_.size();
    NVSM_CHECK(corpus_size_ > 0);
    mean_index_document_length_ = length_sum / static_cast<WeightType>(index_doc_of_.size());
    NVSM_CHECK(mean_index_document_length_ > 0.0);
}

// :591-620. The reference tests `document_id_mapping_` membership with the INDEX document id although that map is keyed
// by MODEL ids; kept as is (it only matters under --document_cutoff / --document_list). Counted from the term lists.
size_t IndexSource::occurrences_in_chosen_documents(TERMID_T term_id) {
    if (!occurrences_counted_) {
        for (DOCID_T d = index_->documentBase(), end = index_->documentMaximum(); d < end; ++d) {
            if (index_doc_of_.count(static_cast<size_t>(d)) == 0) continue;
            for (const TERMID_T t : index_->termList(d)) occurrences_[t] += 1;
        }
        occurrences_counted_ = true;
    }
    const auto hit = occurrences_.find(term_id);
    return hit == occurrences_.end() ? 0 : hit->second;
}

// which terms: document-frequency band, no numbers, no blacklisted terms, then the max_vocabulary_size most frequent;
// model term ids ascend with (collection frequency, index term id) (:749-869)
void IndexSource::choose_vocabulary(size_t max_vocabulary_size, size_t min_document_frequency, size_t max_document_frequency,
                                    bool include_digits, const TermBlacklist* term_blacklist) {
    NVSM_LOG(INFO) << "Building term-id mapping for Indri.";
    // the cap only applies when the index has more terms than it allows (:819); otherwise everything that passes is kept
    const bool capped = max_vocabulary_size > 0 && index_->uniqueTermCount() + 1 > max_vocabulary_size;
    FrequentTerms frequent(capped ? max_vocabulary_size : 0);
    size_t meta = 0, blacklisted = 0, numeric = 0, too_common = 0, too_rare = 0;
    for (const VocabularyEntry& v : index_->vocabulary()) {
        if (v.term_id == 0) ++meta;
        else if (!include_digits && is_number(v.term)) ++numeric;
        else if (min_document_frequency > 0 && v.document_count < min_document_frequency) ++too_rare;
        else if (max_document_frequency > 0 && v.document_count > max_document_frequency) ++too_common;
        else if (term_blacklist != nullptr && term_blacklist->count(v.term)) ++blacklisted;
        else {
            NVSM_CHECK(static_cast<int64_t>(v.total_count) > 0);
            frequent.offer(static_cast<int64_t>(v.total_count), v.term_id);
        }
    }
    if (max_vocabulary_size) NVSM_CHECK(frequent.ascending().size() <= max_vocabulary_size);
    auto admit = [&](TERMID_T index_term, int64_t frequency) {
        const size_t model_term = model_term_of_.size();
        model_term_of_.emplace(index_term, model_term);
        index_term_of_.emplace(model_term, index_term);
        frequency_of_model_term_.emplace(model_term, frequency);
    };
    if (oov_token_) admit(0, 1);
    const bool whole_index = corpus_size() == index_->documentCount();
    for (const auto& entry : frequent.ascending()) {
        const size_t frequency = whole_index ? static_cast<size_t>(entry.first) : occurrences_in_chosen_documents(entry.second);
        if (frequency == 0) continue;                              // never occurs in the chosen documents
        corpus_term_occurrences_ += frequency;
        admit(entry.second, static_cast<int64_t>(frequency));
    }
    NVSM_LOG(INFO) << "Vocabulary filtering discarded " << meta << " meta-terms, " << blacklisted << " blacklisted terms, " << numeric
                   << " terms that contained a digit, " << too_common << " terms that had too high document frequency, " << too_rare
                   << " terms that had too low document frequency.";
    vocabulary_size_ = model_term_of_.size();
    NVSM_CHECK(corpus_term_occurrences_ > 0);
    const double log_ratio = std::log10(static_cast<double>(corpus_term_occurrences_)) - std::log10(static_cast<double>(vocabulary_size_));
    NVSM_LOG(INFO) << "Index contains " << vocabulary_size_ << " unique terms and " << corpus_term_occurrences_
                   << " term occurrences (log-ratio=" << log_ratio << ").";
}

void IndexSource::extract_metadata(Metadata* metadata) const {                  // :530-551
    for (const auto& m : model_term_of_) {
        Metadata::TermInfo t;
        t.index_term_id = static_cast<int32_t>(m.first);
        t.model_term_id = static_cast<int32_t>(m.second);
        t.term_frequency = static_cast<int32_t>(frequency_of_model_term_.at(m.second));
        metadata->term.push_back(t);
    }
    metadata->total_terms = static_cast<int32_t>(corpus_term_occurrences_);
    for (const auto& d : index_doc_of_) {
        Metadata::ObjectInfo o;
        o.model_object_id = static_cast<int32_t>(d.first);
        o.index_object_id = static_cast<int32_t>(d.second);
        metadata->object.push_back(o);
    }
}

std::map<std::string, int64_t> IndexSource::build_term_identifiers_map() const {       // :553-569
    std::map<std::string, int64_t> by_string;
    for (const auto& m : model_term_of_) NVSM_CHECK(by_string.emplace(index_->term(m.first), static_cast<int64_t>(m.second)).second);
    return by_string;
}

std::map<std::string, int64_t> IndexSource::build_document_identifiers_map() const {   // :571-589
    std::map<std::string, int64_t> by_docno;
    for (const auto& d : index_doc_of_) NVSM_CHECK(by_docno.emplace(index_->docno(d.second), static_cast<int64_t>(d.first)).second);
    return by_docno;
}

std::vector<WeightType> IndexSource::compute_term_weights(const std::vector<WordIdxType>& terms) const {
    std::vector<WeightType> weights;
    if (term_weighting_ == UNIFORM_TERM_WEIGHTING) return weights;
    weights.reserve(terms.size());
    for (const WordIdxType t : terms)      // self-information of the term: −log p(term), p in WeightType
        weights.push_back(-std::log(static_cast<WeightType>(frequency_of_model_term_.at(static_cast<size_t>(t))) / corpus_term_occurrences_));
    return weights;
}

int64_t IndexSource::term_id(const std::string& term) const {
    const auto hit = model_term_of_.find(index_->term(term));
    return hit == model_term_of_.end() ? -1 : static_cast<int64_t>(hit->second);
}

std::string IndexSource::term(int64_t model_term_id) const { return index_->term(index_term_of_.at(static_cast<size_t>(model_term_id))); }

}  // namespace nvsm_host
