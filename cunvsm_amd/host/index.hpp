// The slice of indri::index::DiskIndex / QueryEnvironment / CompressedCollection that the reference's IndriSource
// actually calls (cpp/data_indri.cpp:120-131,560-575,620-869,871-887), as an abstract interface. The reference's
// own tests drive IndriSource through exactly this surface with a gmock MockDiskIndex (cpp/data_tests.cpp:192-330);
// here the same seam carries the index back-ends that exist without libindri (trectext_index.hpp) and the test fake.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace nvsm_host {

typedef int64_t TERMID_T;   // lemur::api::TERMID_T (the reference widens it to its 64-bit int32, data.h:385-386)
typedef int64_t DOCID_T;

struct VocabularyEntry {             // indri::index::DiskTermData / TermData as read at data_indri.cpp:747-795
    TERMID_T term_id = 0;
    std::string term;
    uint64_t document_count = 0;     // termData->corpus.documentCount
    uint64_t total_count = 0;        // termData->corpus.totalCount
};

class IndexInterface {
 public:
    virtual ~IndexInterface() {}
    virtual DOCID_T documentBase() = 0;                        // first internal document id
    virtual DOCID_T documentMaximum() = 0;                     // one past the last internal document id
    virtual uint64_t documentCount() = 0;
    virtual int64_t documentLength(DOCID_T doc) = 0;           // indexed length, stopped / OoV positions included
    virtual uint64_t uniqueTermCount() = 0;
    virtual std::vector<VocabularyEntry> vocabulary() = 0;     // vocabularyIterator(), in the index's iteration order
    virtual std::vector<TERMID_T> termList(DOCID_T doc) = 0;   // termList(doc)->terms(): 0 = stopped / out of vocabulary
    virtual std::string term(TERMID_T id) = 0;
    virtual TERMID_T term(const std::string& t) = 0;           // 0 when unknown
    // QueryEnvironment::documentIDsFromMetadata("docno", ...) and CompressedCollection::retrieveMetadatum(doc, "docno")
    virtual std::vector<DOCID_T> documentIDsFromDocno(const std::vector<std::string>& docnos) = 0;
    virtual std::string docno(DOCID_T doc) = 0;
};

}  // namespace nvsm_host
