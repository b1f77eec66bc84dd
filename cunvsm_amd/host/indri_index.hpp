// Reader of an Indri 5.x repository's disk index (index/<n>/ of the repository) — the part of
// indri::index::DiskIndex that IndriSource uses (cpp/data_indri.cpp:18-105,120-131,620-869): document lengths, the
// per-document term lists of the direct file and the vocabulary with its collection statistics. libindri is not
// available here and the on-disk format is documented only by Indri's (un-vendored) source, so the layout below was
// established on the repository the reference ships for its own test (test_data/Brown_index, Indri 5.8) and is pinned
// by that test's expectations (cpp/data_tests.cpp:623-683, restated in tests/cpp/host_tests.cpp):
//   manifest            XML: corpus/{document-base, maximum-document, total-documents, unique-terms, frequent-terms}
//   documentLengths     uint32 per document
//   documentStatistics  24 B per document: uint64 offset, uint32 byteLength, indexedLength, totalLength, uniqueTerms
//   directFile          at `offset`: RVL(termCount) RVL(fieldCount) termCount x RVL(termID) [fields]; 0 = stopped word
//   frequentTerms       per term: RVL totalCount, documentCount, maxDocLen, minDocLen, termID, strlen, bytes,
//                       RVL invertedOffset, invertedLength
//   infrequentString    "BulkTree" of 8 KiB blocks; leaf blocks (uint16 header with bit 15 set, low bits = entry count,
//                       (keyEnd, valueEnd) uint16 pairs growing down from the block end) map term string → RVL
//                       totalCount, documentCount, maxDocLen, minDocLen, localID, invertedOffset, invertedLength;
//                       termID = localID + number of frequent terms; leaves are stored in key order
//   RVL                 little-endian base-128, the LAST byte of a number has bit 7 set
//   collection/forwardLookup0, reverseLookup0   docno ↔ document id (Lemur Keyfile B-trees; layout in indri_index.cpp),
//                       read on the first docno look-up (--document_list, build_document_identifiers_map)
#pragma once

#include <unordered_map>

#include "index.hpp"

namespace nvsm_host {

class IndriDiskIndex : public IndexInterface {
 public:
    // `repository_path`: the directory that holds "manifest", "index/", "collection/"
    static IndriDiskIndex* open(const std::string& repository_path);
    static bool looks_like_repository(const std::string& path);

    DOCID_T documentBase() override { return document_base_; }
    DOCID_T documentMaximum() override { return document_maximum_; }
    uint64_t documentCount() override { return total_documents_; }
    int64_t documentLength(DOCID_T doc) override;
    uint64_t uniqueTermCount() override { return unique_terms_; }
    uint64_t termCount() const { return total_terms_; }
    std::vector<VocabularyEntry> vocabulary() override { return vocabulary_; }
    std::vector<TERMID_T> termList(DOCID_T doc) override;
    std::string term(TERMID_T id) override;
    TERMID_T term(const std::string& t) override;
    std::vector<DOCID_T> documentIDsFromDocno(const std::vector<std::string>& docnos) override;
    std::string docno(DOCID_T doc) override;

 private:
    IndriDiskIndex() {}
    void load_docno_lookups();
    std::string repository_path_;
    bool docnos_loaded_ = false;
    std::unordered_map<DOCID_T, std::string> docno_of_;
    std::unordered_map<std::string, DOCID_T> id_of_docno_;
    DOCID_T document_base_ = 1, document_maximum_ = 1;
    uint64_t total_documents_ = 0, unique_terms_ = 0, total_terms_ = 0;
    std::vector<uint32_t> document_lengths_;
    struct DocStat { uint64_t offset; uint32_t byte_length; };
    std::vector<DocStat> doc_stats_;
    std::string direct_file_;                                  // the whole direct file (1.7 MB for Brown)
    std::vector<VocabularyEntry> vocabulary_;                  // iteration order: frequent terms, then infrequent by string
    std::unordered_map<TERMID_T, size_t> by_id_;
    std::unordered_map<std::string, TERMID_T> by_string_;
};

}  // namespace nvsm_host
