// An in-memory IndexInterface built straight from a TREC-text collection (<DOC><DOCNO>..</DOCNO><TEXT>..</TEXT></DOC>),
// i.e. what `IndriBuildIndex` + libindri provide to the reference (scripts/functions.sh:330-367: class trectext, no
// stemmer, an optional stop list). libindri and its on-disk format are not available here, so the
// index is rebuilt from the text on start-up: Cranfield (1400 documents) takes ~0.1 s.
//   * tokens  = maximal runs of ASCII letters / digits, lower-cased. Indri's TextTokenizer additionally folds
//               acronyms ("n.y." → "ny") and apostrophes; on test_data/cranfield_collection this tokenizer yields
//               261 065 tokens / 1400 documents / longest 698 vs Indri's 260 760 / 1400 / 698 (TUTORIAL.md:34-43).
//   * stopped words keep their position and get term id 0, as Indri's term lists do (the reference treats id 0 as
//     out-of-vocabulary: cpp/data_indri.cpp:120-131, cpp/data_tests.cpp:278-283).
//   * term ids are assigned from 1 in order of first occurrence; documents are numbered from 1 (Indri's documentBase).
#pragma once

#include <istream>
#include <map>
#include <set>
#include <unordered_map>

#include "index.hpp"

namespace nvsm_host {

class TrectextIndex : public IndexInterface {
 public:
    TrectextIndex() {}
    // stopwords: lower-case words (an Indri <stopper><word>..</word></stopper> parameter file or one word per line)
    void load(std::istream& in, const std::set<std::string>& stopwords = std::set<std::string>());
    static TrectextIndex* from_file(const std::string& path, const std::string& stopword_path = "");
    static std::set<std::string> read_stopwords(const std::string& path);
    static std::vector<std::string> tokenize(const std::string& text);

    void add_document(const std::string& docno, const std::string& text, const std::set<std::string>& stopwords);

    DOCID_T documentBase() override { return 1; }
    DOCID_T documentMaximum() override { return static_cast<DOCID_T>(term_lists_.size()) + 1; }
    uint64_t documentCount() override { return term_lists_.size(); }
    int64_t documentLength(DOCID_T doc) override { return static_cast<int64_t>(term_lists_.at(doc - 1).size()); }
    uint64_t uniqueTermCount() override { return terms_.size(); }
    uint64_t termCount() const { return total_terms_; }
    std::vector<VocabularyEntry> vocabulary() override;
    std::vector<TERMID_T> termList(DOCID_T doc) override { return term_lists_.at(doc - 1); }
    std::string term(TERMID_T id) override { return id >= 1 && id <= static_cast<TERMID_T>(terms_.size()) ? terms_[id - 1] : std::string("[OOV]"); }
    TERMID_T term(const std::string& t) override { const auto it = ids_.find(t); return it == ids_.end() ? 0 : it->second; }
    std::vector<DOCID_T> documentIDsFromDocno(const std::vector<std::string>& docnos) override;
    std::string docno(DOCID_T doc) override { return docnos_.at(doc - 1); }

 private:
    std::vector<std::string> terms_;                       // id - 1 → term
    std::unordered_map<std::string, TERMID_T> ids_;
    std::vector<uint64_t> total_count_, document_count_;   // per term id - 1
    std::vector<std::vector<TERMID_T>> term_lists_;        // doc - 1 → term ids (0 = stopped)
    std::vector<std::string> docnos_;
    std::unordered_map<std::string, DOCID_T> docno_to_id_;
    uint64_t total_terms_ = 0;
};

}  // namespace nvsm_host
