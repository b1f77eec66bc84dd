#include "trectext_index.hpp"

#include <algorithm>
#include <fstream>
#include <sstream>

#include "base.hpp"

namespace nvsm_host {

std::vector<std::string> TrectextIndex::tokenize(const std::string& text) {
    std::vector<std::string> tokens;
    std::string cur;
    for (const unsigned char c : text) {
        if ((c >= 'a' && c <= 'z') || (c >= '0' && c <= '9')) cur.push_back(static_cast<char>(c));
        else if (c >= 'A' && c <= 'Z') cur.push_back(static_cast<char>(c - 'A' + 'a'));
        else if (!cur.empty()) { tokens.push_back(cur); cur.clear(); }
    }
    if (!cur.empty()) tokens.push_back(cur);
    return tokens;
}

void TrectextIndex::add_document(const std::string& docno, const std::string& text, const std::set<std::string>& stopwords) {
    std::vector<TERMID_T> list;
    std::set<TERMID_T> seen;
    for (const std::string& tok : tokenize(text)) {
        if (stopwords.count(tok)) { list.push_back(0); continue; }
        auto it = ids_.find(tok);
        TERMID_T id;
        if (it == ids_.end()) {
            terms_.push_back(tok);
            total_count_.push_back(0);
            document_count_.push_back(0);
            id = static_cast<TERMID_T>(terms_.size());
            ids_.emplace(tok, id);
        } else {
            id = it->second;
        }
        total_count_[id - 1] += 1;
        if (seen.insert(id).second) document_count_[id - 1] += 1;
        list.push_back(id);
        ++total_terms_;
    }
    docno_to_id_[docno] = static_cast<DOCID_T>(term_lists_.size()) + 1;
    term_lists_.push_back(std::move(list));
    docnos_.push_back(docno);
}

namespace {
// case-insensitive search of an SGML tag name starting at `from`
size_t find_tag(const std::string& s, const std::string& tag, size_t from) {
    const size_t n = tag.size();
    for (size_t i = s.find('<', from); i != std::string::npos; i = s.find('<', i + 1)) {
        if (i + n > s.size()) return std::string::npos;
        bool ok = true;
        for (size_t j = 0; j < n && ok; ++j) ok = std::toupper(static_cast<unsigned char>(s[i + j])) == tag[j];
        if (ok) return i;
    }
    return std::string::npos;
}
std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace(static_cast<unsigned char>(s[a]))) ++a;
    while (b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) --b;
    return s.substr(a, b - a);
}
}  // namespace

void TrectextIndex::load(std::istream& in, const std::set<std::string>& stopwords) {
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string all = ss.str();
    // the text-bearing fields Indri's trectext class indexes
    static const char* kFields[] = {"<TEXT>", "<HEADLINE>", "<HEAD>", "<TTL>", "<HL>", "<LP>", "<LEADPARA>"};
    size_t pos = 0;
    for (;;) {
        const size_t d0 = find_tag(all, "<DOC>", pos);
        if (d0 == std::string::npos) break;
        size_t d1 = find_tag(all, "</DOC>", d0);
        if (d1 == std::string::npos) d1 = all.size();
        const std::string doc = all.substr(d0, d1 - d0);
        pos = d1 + 1;
        const size_t n0 = find_tag(doc, "<DOCNO>", 0), n1 = find_tag(doc, "</DOCNO>", 0);
        if (n0 == std::string::npos || n1 == std::string::npos || n1 < n0) continue;
        const std::string docno = trim(doc.substr(n0 + 7, n1 - n0 - 7));
        std::string text;
        for (const char* field : kFields) {
            const std::string open(field), close = "</" + open.substr(1);
            size_t p = 0;
            for (;;) {
                const size_t t0 = find_tag(doc, open, p);
                if (t0 == std::string::npos) break;
                size_t t1 = find_tag(doc, close, t0);
                if (t1 == std::string::npos) t1 = doc.size();
                text.append(doc, t0 + open.size(), t1 - t0 - open.size());
                text.push_back('\n');
                p = t1 + 1;
            }
        }
        add_document(docno, text, stopwords);
    }
}

std::set<std::string> TrectextIndex::read_stopwords(const std::string& path) {
    std::set<std::string> words;
    if (path.empty()) return words;
    std::ifstream f(path);
    NVSM_CHECK(f.good()) << "cannot read stop list " << path;
    std::stringstream ss;
    ss << f.rdbuf();
    std::string all = ss.str();
    // Indri parameter file: <word>the</word>; otherwise whitespace-separated words
    if (all.find("<word>") != std::string::npos) {
        size_t p = 0;
        for (;;) {
            const size_t a = all.find("<word>", p);
            if (a == std::string::npos) break;
            const size_t b = all.find("</word>", a);
            if (b == std::string::npos) break;
            for (const std::string& t : tokenize(all.substr(a + 6, b - a - 6))) words.insert(t);
            p = b + 7;
        }
    } else {
        for (const std::string& t : tokenize(all)) words.insert(t);
    }
    return words;
}

TrectextIndex* TrectextIndex::from_file(const std::string& path, const std::string& stopword_path) {
    std::ifstream f(path);
    NVSM_CHECK(f.good()) << "cannot read collection " << path;
    TrectextIndex* index = new TrectextIndex;
    index->load(f, read_stopwords(stopword_path));
    return index;
}

std::vector<VocabularyEntry> TrectextIndex::vocabulary() {
    std::vector<VocabularyEntry> v(terms_.size());
    for (size_t i = 0; i < terms_.size(); ++i) {
        v[i].term_id = static_cast<TERMID_T>(i + 1);
        v[i].term = terms_[i];
        v[i].total_count = total_count_[i];
        v[i].document_count = document_count_[i];
    }
    return v;
}

std::vector<DOCID_T> TrectextIndex::documentIDsFromDocno(const std::vector<std::string>& docnos) {
    std::vector<DOCID_T> ids;
    for (const std::string& d : docnos) {
        const auto it = docno_to_id_.find(d);
        if (it != docno_to_id_.end()) ids.push_back(it->second);
    }
    return ids;
}

}  // namespace nvsm_host
