#include "indri_index.hpp"

#include <sys/stat.h>

#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

#include "base.hpp"

namespace nvsm_host {
namespace {

std::string read_file(const std::string& path, bool required = true) {
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) {
        if (required) NVSM_LOG(FATAL) << "Unable to open Indri index: cannot read " << path;
        return std::string();
    }
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// <tag>value</tag> (first occurrence after `from`)
std::string xml_value(const std::string& xml, const std::string& tag, size_t from = 0) {
    const std::string open = "<" + tag + ">", close = "</" + tag + ">";
    const size_t a = xml.find(open, from);
    if (a == std::string::npos) return std::string();
    const size_t b = xml.find(close, a);
    if (b == std::string::npos) return std::string();
    return xml.substr(a + open.size(), b - a - open.size());
}

struct RvlReader {
    const unsigned char* p; const unsigned char* end;
    bool ok = true;
    uint64_t next() {
        uint64_t v = 0; int shift = 0;
        while (p < end && shift < 64) {
            const unsigned char c = *p++;
            if (c & 0x80) return v | (static_cast<uint64_t>(c & 0x7f) << shift);
            v |= static_cast<uint64_t>(c) << shift;
            shift += 7;
        }
        ok = false;
        return 0;
    }
};

bool exists(const std::string& path) { struct stat st; return stat(path.c_str(), &st) == 0; }

}  // namespace

bool IndriDiskIndex::looks_like_repository(const std::string& path) {
    return exists(path + "/manifest") && exists(path + "/index");
}

IndriDiskIndex* IndriDiskIndex::open(const std::string& repository_path) {
    // LoadIndex (cpp/data_indri.cpp:35-66): the repository manifest must name exactly one index
    const std::string repo_manifest = read_file(repository_path + "/manifest");
    const std::string indexes = xml_value(repo_manifest, "indexes");
    if (indexes.empty()) NVSM_LOG(FATAL) << "Indri repository does not contain an index.";
    size_t count = 0;
    for (size_t p = indexes.find("<index>"); p != std::string::npos; p = indexes.find("<index>", p + 1)) ++count;
    if (count != 1) NVSM_LOG(FATAL) << "Indri repository contain more than one index.";
    std::string index_name = xml_value(indexes, "index");
    const std::string dir = repository_path + "/index/" + index_name + "/";

    std::unique_ptr<IndriDiskIndex> idx(new IndriDiskIndex);
    idx->repository_path_ = repository_path;
    const std::string manifest = read_file(dir + "manifest");
    const std::string corpus = xml_value(manifest, "corpus");
    if (corpus.empty() || xml_value(manifest, "type") != "DiskIndex") NVSM_LOG(FATAL) << "Unable to open Indri index: " << dir << "manifest is not a DiskIndex manifest";
    idx->document_base_ = std::stoll(xml_value(corpus, "document-base"));
    idx->document_maximum_ = std::stoll(xml_value(corpus, "maximum-document"));
    idx->total_documents_ = std::stoull(xml_value(corpus, "total-documents"));
    idx->unique_terms_ = std::stoull(xml_value(corpus, "unique-terms"));
    idx->total_terms_ = std::stoull(xml_value(corpus, "total-terms"));
    const uint64_t frequent = std::stoull(xml_value(corpus, "frequent-terms"));

    const std::string lengths = read_file(dir + "documentLengths");
    const std::string stats = read_file(dir + "documentStatistics");
    NVSM_CHECK(lengths.size() == idx->total_documents_ * 4) << "documentLengths has an unexpected size";
    NVSM_CHECK(stats.size() == idx->total_documents_ * 24) << "documentStatistics has an unexpected size";
    idx->document_lengths_.resize(idx->total_documents_);
    std::memcpy(idx->document_lengths_.data(), lengths.data(), lengths.size());
    idx->doc_stats_.resize(idx->total_documents_);
    for (uint64_t d = 0; d < idx->total_documents_; ++d) {
        std::memcpy(&idx->doc_stats_[d].offset, stats.data() + 24 * d, 8);
        std::memcpy(&idx->doc_stats_[d].byte_length, stats.data() + 24 * d + 8, 4);
    }
    idx->direct_file_ = read_file(dir + "directFile");

    // vocabulary: frequent terms first, then the infrequent ones in string order (DiskIndex::vocabularyIterator)
    auto add = [&](TERMID_T id, const std::string& term, uint64_t total, uint64_t docs) {
        VocabularyEntry e;
        e.term_id = id; e.term = term; e.total_count = total; e.document_count = docs;
        idx->by_id_[id] = idx->vocabulary_.size();
        idx->by_string_[term] = id;
        idx->vocabulary_.push_back(e);
    };
    {
        const std::string ft = read_file(dir + "frequentTerms", false);
        RvlReader r{reinterpret_cast<const unsigned char*>(ft.data()), reinterpret_cast<const unsigned char*>(ft.data()) + ft.size()};
        while (r.p < r.end) {
            const uint64_t total = r.next(), docs = r.next();
            r.next(); r.next();                                    // max / min document length
            const uint64_t id = r.next(), len = r.next();
            if (!r.ok || static_cast<uint64_t>(r.end - r.p) < len) NVSM_LOG(FATAL) << "frequentTerms is corrupt";
            const std::string term(reinterpret_cast<const char*>(r.p), len);
            r.p += len;
            r.next(); r.next();                                    // inverted-file offset / length
            if (!r.ok) NVSM_LOG(FATAL) << "frequentTerms is corrupt";
            add(static_cast<TERMID_T>(id), term, total, docs);
        }
        NVSM_CHECK(idx->vocabulary_.size() == frequent) << "frequent-terms of the manifest disagrees with frequentTerms";
    }
    {
        const std::string tree = read_file(dir + "infrequentString", false);
        constexpr size_t kBlock = 8192;
        for (size_t b = 0; b + kBlock <= tree.size(); b += kBlock) {
            const unsigned char* blk = reinterpret_cast<const unsigned char*>(tree.data()) + b;
            const unsigned header = blk[0] | (blk[1] << 8);
            if (!(header & 0x8000)) continue;                      // interior node
            const unsigned count = header & 0x7fff;
            unsigned prev_end = 2;
            for (unsigned i = 0; i < count; ++i) {
                const unsigned char* pair = blk + kBlock - 4 * (i + 1);
                const unsigned key_end = pair[0] | (pair[1] << 8), value_end = pair[2] | (pair[3] << 8);
                if (key_end < prev_end || value_end < key_end || value_end > kBlock) NVSM_LOG(FATAL) << "infrequentString is corrupt";
                const std::string term(reinterpret_cast<const char*>(blk) + prev_end, key_end - prev_end);
                RvlReader r{blk + key_end, blk + value_end};
                const uint64_t total = r.next(), docs = r.next();
                r.next(); r.next();
                const uint64_t local_id = r.next();
                if (!r.ok) NVSM_LOG(FATAL) << "infrequentString is corrupt";
                add(static_cast<TERMID_T>(local_id + frequent), term, total, docs);
                prev_end = value_end;
            }
        }
    }
    NVSM_CHECK(idx->vocabulary_.size() == idx->unique_terms_) << "read " << idx->vocabulary_.size() << " terms, the manifest says " << idx->unique_terms_;
    return idx.release();
}

int64_t IndriDiskIndex::documentLength(DOCID_T doc) {
    NVSM_CHECK(doc >= document_base_ && doc < document_maximum_) << "document id out of range";
    return document_lengths_[static_cast<size_t>(doc - document_base_)];
}

std::vector<TERMID_T> IndriDiskIndex::termList(DOCID_T doc) {
    NVSM_CHECK(doc >= document_base_ && doc < document_maximum_) << "document id out of range";
    const DocStat& st = doc_stats_[static_cast<size_t>(doc - document_base_)];
    NVSM_CHECK(st.offset + st.byte_length <= direct_file_.size()) << "directFile is truncated";
    const unsigned char* p = reinterpret_cast<const unsigned char*>(direct_file_.data()) + st.offset;
    RvlReader r{p, p + st.byte_length};
    const uint64_t term_count = r.next();
    r.next();                                                      // field count
    std::vector<TERMID_T> terms;
    terms.reserve(term_count);
    for (uint64_t i = 0; i < term_count; ++i) terms.push_back(static_cast<TERMID_T>(r.next()));
    NVSM_CHECK(r.ok) << "directFile is corrupt";
    return terms;
}

std::string IndriDiskIndex::term(TERMID_T id) {
    const auto it = by_id_.find(id);
    return it == by_id_.end() ? std::string("[OOV]") : vocabulary_[it->second].term;
}

TERMID_T IndriDiskIndex::term(const std::string& t) {
    const auto it = by_string_.find(t);
    return it == by_string_.end() ? 0 : it->second;
}

// ---------------------------------------------------------------------------------------------------------------------
// docno look-ups: collection/forwardLookup0 (document id → docno) and collection/reverseLookup0 (docno → document id),
// what QueryEnvironment::documentIDsFromMetadata("docno", …) and CompressedCollection::retrieveMetadatum(doc, "docno")
// read (cpp/data_indri.cpp:695,571-589). Both are Lemur "Keyfile" B-trees of 4 KiB pages. Layout as established on the
// reference's own repository (test_data/Brown_index/collection, 500 documents ca01 … cr09 ↔ ids 1 … 500):
//   leaf page   u16be record count | u16be bytes used | u16be prefix length L | u16be level (0 = leaf) | … 28 bytes of
//               header in all; then `count` u16be record offsets (relative to byte 28 of the page, records grow down from
//               the page end, table order = key order); the last L bytes of the page are the key prefix all records of
//               the page share and do not repeat
//   record      u8 key-suffix length, suffix, u8 value length, value
//   forward     key = the id as six base-64 digits, most significant first, each OR 0x40 ("@@@@Eq" = 369); value =
//               the docno, NUL-terminated
//   reverse     key = the docno; value = the id as little-endian int32
// Index (level > 0) pages are not needed: every leaf of the file is visited once and the pairs go into hash maps.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr size_t kKeyfilePage = 4096, kKeyfileHeader = 28;

template <typename Fn>
void for_each_keyfile_record(const std::string& file, const std::string& path, Fn&& fn) {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(file.data());
    auto be16 = [](const unsigned char* p) { return static_cast<size_t>(p[0]) << 8 | p[1]; };
    for (size_t page = kKeyfilePage; page + kKeyfilePage <= file.size(); page += kKeyfilePage) {     // page 0 is the file header
        const unsigned char* pg = base + page;
        const size_t count = be16(pg), prefix_len = be16(pg + 4), level = be16(pg + 6);
        if (count == 0 || level != 0) continue;
        if (kKeyfileHeader + 2 * count + prefix_len > kKeyfilePage) continue;                        // not a leaf page
        const std::string prefix(reinterpret_cast<const char*>(pg + kKeyfilePage - prefix_len), prefix_len);
        for (size_t i = 0; i < count; ++i) {
            const size_t at = kKeyfileHeader + be16(pg + kKeyfileHeader + 2 * i);
            NVSM_CHECK(at + 2 <= kKeyfilePage) << path << ": record offset outside its page";
            const size_t klen = pg[at];
            NVSM_CHECK(at + 1 + klen + 1 <= kKeyfilePage) << path << ": key runs over the page end";
            const size_t vlen = pg[at + 1 + klen];
            NVSM_CHECK(vlen < 128) << path << ": values of 128 bytes or more are not supported by this reader";
            NVSM_CHECK(at + 1 + klen + 1 + vlen <= kKeyfilePage) << path << ": value runs over the page end";
            fn(prefix + std::string(reinterpret_cast<const char*>(pg + at + 1), klen), pg + at + 1 + klen + 1, vlen);
        }
    }
}
}  // namespace

void IndriDiskIndex::load_docno_lookups() {
    if (docnos_loaded_) return;
    docnos_loaded_ = true;
    const std::string fwd_path = repository_path_ + "/collection/forwardLookup0", rev_path = repository_path_ + "/collection/reverseLookup0";
    const std::string fwd = read_file(fwd_path, false), rev = read_file(rev_path, false);
    if (fwd.empty() || rev.empty())
        NVSM_LOG(FATAL) << "the repository has no docno look-up files (" << fwd_path << ", " << rev_path
                        << "): it was built without <metadata><forward>docno</forward><backward>docno</backward></metadata>";
    for_each_keyfile_record(fwd, fwd_path, [&](const std::string& key, const unsigned char* value, size_t vlen) {
        int64_t id = 0;
        for (const char c : key) {
            NVSM_CHECK((static_cast<unsigned char>(c) & 0xc0) == 0x40) << fwd_path << ": unexpected key byte";
            id = id * 64 + (static_cast<unsigned char>(c) & 0x3f);
        }
        size_t n = vlen;
        while (n > 0 && value[n - 1] == 0) --n;                    // stored NUL-terminated
        docno_of_[static_cast<DOCID_T>(id)] = std::string(reinterpret_cast<const char*>(value), n);
    });
    for_each_keyfile_record(rev, rev_path, [&](const std::string& key, const unsigned char* value, size_t vlen) {
        NVSM_CHECK(vlen == 4) << rev_path << ": document ids are expected as 4-byte integers";
        uint32_t id = 0;
        std::memcpy(&id, value, 4);
        id_of_docno_[key] = static_cast<DOCID_T>(id);
    });
    NVSM_CHECK(docno_of_.size() == total_documents_ && id_of_docno_.size() == total_documents_)
        << "docno look-up files hold " << docno_of_.size() << " / " << id_of_docno_.size() << " entries for " << total_documents_ << " documents";
}

// QueryEnvironment::documentIDsFromMetadata("docno", list) (cpp/data_indri.cpp:695): one id per listed docno, in list order
std::vector<DOCID_T> IndriDiskIndex::documentIDsFromDocno(const std::vector<std::string>& docnos) {
    load_docno_lookups();
    std::vector<DOCID_T> ids;
    ids.reserve(docnos.size());
    for (const std::string& d : docnos) {
        const auto hit = id_of_docno_.find(d);
        if (hit == id_of_docno_.end()) NVSM_LOG(FATAL) << "document list names " << d << ", which the index does not hold";
        ids.push_back(hit->second);
    }
    return ids;
}

std::string IndriDiskIndex::docno(DOCID_T doc) {
    load_docno_lookups();
    const auto hit = docno_of_.find(doc);
    NVSM_CHECK(hit != docno_of_.end()) << "no docno for document " << doc;
    return hit->second;
}

}  // namespace nvsm_host
