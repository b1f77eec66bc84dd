// CPU unit tests of the host layer (cunvsm_amd/host): every expectation below restates a googletest case of the
// reference — cpp/data_tests.cpp (data sources, IndriSource over a mock index, Async / Repeating sources) and
// cpp/utils_tests.cpp — with the same inputs and the same expected values (numbers only). Driven by
// tests/test_host_layer.py, which builds this file with g++ and checks the per-test verdicts.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <set>
#include <sstream>

#include "../../cunvsm_amd/host/data.hpp"
#include "../../cunvsm_amd/host/index_source.hpp"
#include "../../cunvsm_amd/host/indri_index.hpp"
#include "../../cunvsm_amd/host/rendezvous.hpp"
#include "../../cunvsm_amd/host/trectext_index.hpp"

using namespace nvsm_host;

static int g_failures = 0;
#define EXPECT_TRUE(c) do { if (!(c)) { std::printf("    EXPECT failed %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_failures; } } while (0)
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) { std::ostringstream os_; os_ << (a) << " vs " << (b); \
    std::printf("    EXPECT_EQ failed %s:%d: %s == %s (%s)\n", __FILE__, __LINE__, #a, #b, os_.str().c_str()); ++g_failures; } } while (0)
#define EXPECT_NEAR(a, b, tol) do { if (!(std::fabs((a) - (b)) <= (tol))) { std::printf("    EXPECT_NEAR failed %s:%d: %s=%g %s=%g\n", \
    __FILE__, __LINE__, #a, double(a), #b, double(b)); ++g_failures; } } while (0)

template <typename T>
static std::string str(const std::vector<T>& v) { std::ostringstream os; for (const auto& x : v) os << x << " "; return os.str(); }
#define EXPECT_VEC(a, ...) do { const std::vector<long> exp_ = {__VA_ARGS__}; const auto tmp_ = (a); std::vector<long> got_(tmp_.begin(), tmp_.end()); \
    if (got_ != exp_) { std::printf("    EXPECT_VEC failed %s:%d: got [%s] want [%s]\n", __FILE__, __LINE__, str(got_).c_str(), str(exp_).c_str()); ++g_failures; } } while (0)

static std::vector<WordIdxType> features(const Batch& b) { return std::vector<WordIdxType>(b.features(), b.features() + b.num_instances() * b.window_size()); }
static std::vector<WeightType> feature_weights(const Batch& b) { return std::vector<WeightType>(b.feature_weights(), b.feature_weights() + b.num_instances() * b.window_size()); }
static std::vector<ObjectIdxType> labels(const Batch& b) { return std::vector<ObjectIdxType>(b.labels(), b.labels() + b.num_instances()); }
static std::vector<WeightType> weights(const Batch& b) { return std::vector<WeightType>(b.weights(), b.weights() + b.num_instances()); }

// ---- cpp/data_tests.cpp:70-110 ----
static const std::vector<std::string> kWords = {"hello", "world", "freedom", "of", "speech", "does", "not", "exist", "or", "to", "that", "is", "question"};
static const CorpusT kDocs = {{1, "hello world freedom of speech"}, {2, "speech freedom does not exist"}, {3, "exist or not to exist that is question"}};

static void test_InMemoryDocumentSource() {
    InMemoryDocumentSource source(construct_vocabulary(kWords), kDocs);
    Batch batch(1024, 3);
    source.next(&batch);
    EXPECT_EQ(batch.num_instances(), 12u);
}
static void test_InMemoryDocumentSource_pad_batch() {
    InMemoryDocumentSource source(construct_vocabulary(kWords), kDocs, true);
    Batch batch(1024, 3);
    source.next(&batch);
    EXPECT_EQ(batch.num_instances(), 1024u);
}

class NullSource : public DataSource {
 public:
    NullSource() : DataSource(11, 11) {}
    void reset() override {}
};

// ---- :127-190 ----
static void test_create_instances() {
    NullSource source;
    Batch batch(6, 3);
    source.create_instances(std::vector<size_t>{1, 2, 3, 4, 5, 6, 7, 8}, 1337, 1.0, 1, &batch);
    EXPECT_VEC(features(batch), 1, 2, 3, 2, 3, 4, 3, 4, 5, 4, 5, 6, 5, 6, 7, 6, 7, 8);
    EXPECT_VEC(labels(batch), 1337, 1337, 1337, 1337, 1337, 1337);
}
static void test_create_instances_overflow() {
    NullSource source;
    Batch batch(2, 3);
    source.create_instances(std::vector<size_t>{1, 2, 3, 4, 5, 6, 7, 8}, 1337, 1.0, 1, &batch);
    EXPECT_TRUE(source.has_next());
    EXPECT_VEC(features(batch), 1, 2, 3, 2, 3, 4);
    batch.clear();
    EXPECT_TRUE(source.has_next());
    source.next(&batch);
    EXPECT_VEC(features(batch), 3, 4, 5, 4, 5, 6);
    batch.clear();
    EXPECT_TRUE(source.has_next());
    source.next(&batch);
    EXPECT_VEC(features(batch), 5, 6, 7, 6, 7, 8);
    batch.clear();
    EXPECT_TRUE(!source.has_next());
}

// ---- the MockDiskIndex of :192-330 as a plain IndexInterface ----
class FakeIndex : public IndexInterface {
 public:
    FakeIndex(bool add_oov, bool actual_tf) {
        const int tf[7] = {2, 3, 2, 1, 1, 1, 1};
        const TERMID_T ids[7] = {1, 2, 3, 4, 5, 10, 111};
        for (int i = 0; i < 7; ++i) { VocabularyEntry e; e.term_id = ids[i]; e.term = "test"; e.total_count = actual_tf ? tf[i] : 5; vocab_.push_back(e); }
        d0_ = {1, 2, 3, 4};
        if (add_oov) d0_.insert(d0_.end(), {0, 0, 0});
        d0_.insert(d0_.end(), {3, 2, 1});
        d1_ = {10, 2};
        if (add_oov) d1_.insert(d1_.end(), {0, 0, 0, 0, 0});
        d1_.insert(d1_.end(), {111, 5});
        len0_ = 7 + (add_oov ? 3 : 0);
        len1_ = 4 + (add_oov ? 5 : 0);
    }
    DOCID_T documentBase() override { return 0; }
    DOCID_T documentMaximum() override { return 2; }
    uint64_t documentCount() override { return 2; }
    int64_t documentLength(DOCID_T d) override { return d == 0 ? len0_ : len1_; }
    uint64_t uniqueTermCount() override { return 7; }
    std::vector<VocabularyEntry> vocabulary() override { return vocab_; }
    std::vector<TERMID_T> termList(DOCID_T d) override { return d == 0 ? d0_ : d1_; }
    std::string term(TERMID_T) override { return "test"; }
    TERMID_T term(const std::string&) override { return 0; }
    std::vector<DOCID_T> documentIDsFromDocno(const std::vector<std::string>&) override { return {}; }
    std::string docno(DOCID_T d) override { return std::to_string(d); }
 private:
    std::vector<VocabularyEntry> vocab_;
    std::vector<TERMID_T> d0_, d1_;
    int64_t len0_, len1_;
};

typedef std::pair<ObjectIdxType, std::vector<WordIdxType>> Inst;
static std::vector<Inst> get_instances(IndexSource* source) {             // :332-360
    std::vector<Inst> out;
    Batch batch(4, source->window_size());
    while (source->has_next()) {
        source->next(&batch);
        const auto f = features(batch);
        const auto l = labels(batch);
        for (size_t i = 0; i < l.size(); ++i)
            out.push_back({l[i], std::vector<WordIdxType>(f.begin() + i * source->window_size(), f.begin() + (i + 1) * source->window_size())});
        batch.clear();
    }
    return out;
}
static std::multiset<Inst> as_set(const std::vector<Inst>& v) { return std::multiset<Inst>(v.begin(), v.end()); }

// ---- :365-462 ----
static void test_IndriSource() {
    RNG rng;
    IndexSource source(new FakeIndex(true, false), 3, &rng, 0, 0, 0, 0, false, false, nullptr, nullptr, false, NONE);
    EXPECT_EQ(source.vocabulary_size(), 7u);
    EXPECT_EQ(source.corpus_size(), 2u);
    const IndexSource::TermIdMapping want_terms = {{1, 0}, {2, 1}, {3, 2}, {4, 3}, {5, 4}, {10, 5}, {111, 6}};
    EXPECT_TRUE(source.term_id_mapping() == want_terms);
    const IndexSource::DocumentIdMapping want_docs = {{0, 0}, {1, 1}};
    EXPECT_TRUE(source.document_id_mapping() == want_docs);

    Batch batch(4, 3);
    EXPECT_TRUE(source.has_next());
    source.next(&batch);
    EXPECT_VEC(features(batch), 0, 1, 2, 1, 2, 3, 2, 3, 2, 3, 2, 1);
    for (const WeightType w : feature_weights(batch)) EXPECT_EQ(w, 1.0f);
    EXPECT_VEC(labels(batch), 0, 0, 0, 0);
    const float avg_doc_length = 9.5;
    for (const WeightType w : weights(batch)) EXPECT_NEAR(w, avg_doc_length / 10.0, 1e-6);
    batch.clear();

    EXPECT_TRUE(source.has_next());
    source.next(&batch);
    EXPECT_VEC(features(batch), 2, 1, 0, 5, 1, 6, 1, 6, 4);
    for (const WeightType w : feature_weights(batch)) EXPECT_EQ(w, 1.0f);
    EXPECT_VEC(labels(batch), 0, 1, 1);
    const auto w = weights(batch);
    EXPECT_EQ(w.size(), 3u);
    if (w.size() == 3) { EXPECT_NEAR(w[0], avg_doc_length / 10.0, 1e-6); EXPECT_NEAR(w[1], avg_doc_length / 9.0, 1e-6); EXPECT_NEAR(w[2], avg_doc_length / 9.0, 1e-6); }
    EXPECT_TRUE(!source.has_next());
    source.reset();
    EXPECT_TRUE(source.has_next());
}

// ---- :464-478 ----
static void test_IndriSource_UnsupportedSampling_Death() {
    RNG rng;
    bool died = false;
    try { IndexSource source(new FakeIndex(false, false), 3, &rng, 0, 0, 0, 0, false, false, nullptr, nullptr, false, NGRAM_FREQUENCY); }
    catch (const FatalError&) { died = true; }
    EXPECT_TRUE(died);
}

// ---- :480-506 ----
static void test_StochasticIndriSource() {
    RNG rng;
    IndexSource source(new FakeIndex(true, false), 3, &rng, 0, 0, 0, 0, false, false, nullptr, nullptr, true, NONE);
    const std::vector<Inst> want = {{0, {0, 1, 2}}, {0, {1, 2, 3}}, {0, {2, 3, 2}}, {0, {3, 2, 1}}, {0, {2, 1, 0}}, {1, {5, 1, 6}}, {1, {1, 6, 4}}};
    EXPECT_TRUE(as_set(get_instances(&source)) == as_set(want));
}

// ---- :508-538 ("Relies on seed == 1": a default-constructed minstd_rand0) ----
static void test_StochasticIndriSource_Resampling() {
    RNG rng;
    IndexSource source(new FakeIndex(true, false), 3, &rng, 0, 0, 0, 0, false, false, nullptr, nullptr, true, NGRAM_FREQUENCY);
    const std::vector<Inst> want = {{0, {0, 1, 2}}, {0, {0, 1, 2}}, {0, {2, 3, 2}}, {0, {3, 2, 1}},
                                    {1, {5, 1, 6}}, {1, {5, 1, 6}}, {1, {1, 6, 4}}, {1, {1, 6, 4}}};
    const auto got = get_instances(&source);
    EXPECT_TRUE(as_set(got) == as_set(want));
    if (!(as_set(got) == as_set(want))) for (const auto& g : got) std::printf("      got (%ld: %s)\n", long(g.first), str(g.second).c_str());
}

// ---- :540-590 ----
static void test_StochasticIndriSource_SelfInformation() {
    RNG rng;
    IndexSource source(new FakeIndex(true, true), 3, &rng, 0, 0, 0, 0, false, false, nullptr, nullptr, true, NGRAM_FREQUENCY, UNIFORM,
                       SELF_INFORMATION_TERM_WEIGHTING);
    const std::map<size_t, int64_t> want_tf = {{0, 1}, {1, 1}, {2, 1}, {3, 1}, {4, 2}, {5, 2}, {6, 3}};
    EXPECT_TRUE(source.term_frequencies() == want_tf);
    Batch batch(4, 3);
    EXPECT_TRUE(source.has_next());
    source.next(&batch);
    EXPECT_VEC(features(batch), 6, 3, 1, 5, 0, 5, 6, 3, 1, 4, 6, 5);
    const double tf[12] = {3, 1, 1, 2, 1, 2, 3, 1, 1, 2, 3, 2};
    const auto fw = feature_weights(batch);
    EXPECT_EQ(fw.size(), 12u);
    for (size_t i = 0; i < fw.size() && i < 12; ++i) EXPECT_NEAR(fw[i], -std::log(tf[i] / 11.0), 1e-6);
}

// ---- :738-772 ----
class CountingSource : public DataSource {
 public:
    explicit CountingSource(size_t num_batches) : DataSource(num_batches, num_batches), num_batches_(num_batches) {}
    void reset() override { batch_idx_ = 0; }
    void next(Batch* batch) override {
        DataSource::next(batch);
        for (size_t i = 0; i < batch->maximum_size(); ++i)
            push_instance(std::vector<WordIdxType>(batch->window_size(), batch_idx_), std::vector<WeightType>(), batch_idx_, 1.0, batch);
        ++batch_idx_;
    }
    bool has_next() const override { return batch_idx_ < num_batches_; }
 private:
    const size_t num_batches_;
    size_t batch_idx_ = 0;
};

// ---- :782-812 (the reference repeats it 11 times to shake out races) ----
static void test_AsyncSource() {
    for (int round = 0; round < 11; ++round) {
        AsyncSource source(3, 128, 3, new CountingSource(8));
        Batch batch(128, 3);
        size_t idx = 0;
        while (source.has_next()) {
            source.next(&batch);
            EXPECT_TRUE(features(batch) == std::vector<WordIdxType>(128 * 3, idx));
            EXPECT_TRUE(labels(batch) == std::vector<ObjectIdxType>(128, idx));
            EXPECT_TRUE(weights(batch) == std::vector<WeightType>(128, 1.0));
            batch.clear();
            ++idx;
        }
        EXPECT_EQ(idx, 8u);
        source.reset();
        // a second epoch after reset() delivers the same stream
        idx = 0;
        while (source.has_next()) { source.next(&batch); EXPECT_TRUE(labels(batch) == std::vector<ObjectIdxType>(128, idx)); batch.clear(); ++idx; }
        EXPECT_EQ(idx, 8u);
    }
}

// ---- :868-896 ----
static void test_RepeatingSource() {
    RepeatingSource source(3, new CountingSource(2));
    Batch batch(128, 3);
    size_t idx = 0;
    while (source.has_next()) {
        source.next(&batch);
        EXPECT_TRUE(features(batch) == std::vector<WordIdxType>(128 * 3, idx % 2));
        EXPECT_TRUE(labels(batch) == std::vector<ObjectIdxType>(128, idx % 2));
        EXPECT_TRUE(weights(batch) == std::vector<WeightType>(128, 1.0));
        batch.clear();
        ++idx;
    }
    EXPECT_EQ(idx, 6u);
    source.reset();
}

// ---- cpp/utils_tests.cpp ----
static void test_utils() {
    EXPECT_TRUE((range<float>(1, 5, 2) == std::vector<float>{1, 1, 2, 2, 3, 3, 4, 4}));
    std::vector<size_t> flattened;
    flatten<size_t>({{8, 9, 10}, {5, 7, 2}, {3}}, &flattened);
    EXPECT_TRUE((flattened == std::vector<size_t>{8, 9, 10, 5, 7, 2, 3}));
    EXPECT_TRUE(is_number("123"));
    EXPECT_TRUE(is_number("aaa1bbb2ccc3d"));
    EXPECT_TRUE(!is_number("hello"));
    EXPECT_EQ(seconds_to_humanreadable_time(3725.9), std::string("1 hours, 2 minutes and 5 seconds"));
}

static void test_Batch_swap() {                                   // cpp/data.cu:77-92
    NullSource source;
    Batch a(4, 2), b(4, 2);
    source.push_instance({7, 8}, {0.5f, 0.25f}, 3, 2.0f, &a);
    const WordIdxType* pa = a.features();
    a.swap(&b);
    EXPECT_TRUE(a.empty());
    EXPECT_EQ(b.num_instances(), 1u);
    EXPECT_TRUE(b.features() == pa);
    EXPECT_VEC(features(b), 7, 8);
    EXPECT_EQ(feature_weights(b)[1], 0.25f);
    EXPECT_EQ(weights(b)[0], 2.0f);
}

// ---- Metadata wire format (proto/nvsm.proto:88-103) ----
static void test_Metadata_roundtrip() {
    Metadata m;
    m.term.push_back({5, 0, 7}); m.term.push_back({300, 1, 123456}); m.term.push_back({0, 0, 0});
    m.object.push_back({1, 0}); m.object.push_back({70000, 69999});
    m.total_terms = 260760;
    const std::string wire = m.SerializeAsString();
    // field 1 (term), length 4: 08 05 18 07  (model_term_id 0 is a proto3 default and stays off the wire)
    const unsigned char head[6] = {0x0a, 0x04, 0x08, 0x05, 0x18, 0x07};
    EXPECT_TRUE(wire.size() > 6 && std::equal(head, head + 6, reinterpret_cast<const unsigned char*>(wire.data())));
    Metadata p;
    EXPECT_TRUE(p.ParseFromString(wire));
    EXPECT_EQ(p.term.size(), 3u);
    EXPECT_EQ(p.object.size(), 2u);
    EXPECT_EQ(p.total_terms, 260760);
    if (p.term.size() == 3) { EXPECT_EQ(p.term[1].index_term_id, 300); EXPECT_EQ(p.term[1].term_frequency, 123456); EXPECT_EQ(p.term[2].index_term_id, 0); }
    if (p.object.size() == 2) { EXPECT_EQ(p.object[1].index_object_id, 70000); EXPECT_EQ(p.object[1].model_object_id, 69999); }
    EXPECT_TRUE(!p.ParseFromString(std::string("\x0a\x7f", 2)));     // truncated length-delimited field
}

// ---- TrectextIndex + IndexSource end to end on an inline collection ----
static void test_TrectextIndex() {
    std::istringstream in(
        "<DOC>\n<DOCNO> d-1 </DOCNO>\n<TEXT>\nThe quick brown fox, the LAZY dog. n.y. 1958\n</TEXT>\n</DOC>\n"
        "<DOC>\n<DOCNO>d-2</DOCNO>\n<HEADLINE>Fox news</HEADLINE>\n<TEXT>\nquick quick fox\n</TEXT>\n</DOC>\n");
    TrectextIndex* index = new TrectextIndex;
    index->load(in, {"the"});
    EXPECT_EQ(index->documentCount(), 2u);
    EXPECT_EQ(index->documentBase(), 1);
    EXPECT_EQ(index->documentMaximum(), 3);
    // the quick brown fox the lazy dog n y 1958
    EXPECT_EQ(index->documentLength(1), 10);
    EXPECT_VEC(index->termList(1), 0, 1, 2, 3, 0, 4, 5, 6, 7, 8);
    EXPECT_EQ(index->term(TERMID_T(3)), std::string("fox"));
    EXPECT_EQ(index->term(std::string("lazy")), 4);
    EXPECT_EQ(index->term(std::string("the")), 0);
    EXPECT_EQ(index->docno(2), std::string("d-2"));
    // headline text is indexed after the body fields in field order: TEXT first, then HEADLINE
    EXPECT_VEC(index->termList(2), 1, 1, 3, 3, 9);
    std::vector<VocabularyEntry> v = index->vocabulary();
    EXPECT_EQ(v.size(), 9u);
    EXPECT_EQ(v[0].term, std::string("quick")); EXPECT_EQ(v[0].total_count, 3u); EXPECT_EQ(v[0].document_count, 2u);
    EXPECT_EQ(v[2].term, std::string("fox")); EXPECT_EQ(v[2].total_count, 3u);
    EXPECT_VEC(index->documentIDsFromDocno({"d-2", "d-1"}), 2, 1);

    RNG rng;
    IndexSource source(index, 3, &rng, 0, 0, 0, 0, false, false /* digits dropped */, nullptr, nullptr, false, NONE);
    EXPECT_EQ(source.corpus_size(), 2u);
    EXPECT_EQ(source.vocabulary_size(), 8u);                     // "1958" contains a digit
    EXPECT_EQ(source.term_id("1958"), -1);
    EXPECT_EQ(source.term(source.term_id("dog")), std::string("dog"));
    Metadata meta;
    source.extract_metadata(&meta);
    EXPECT_EQ(meta.term_size(), 8u);
    EXPECT_EQ(meta.object_size(), 2u);
    EXPECT_EQ(meta.total_terms, 3 + 1 + 3 + 1 + 1 + 1 + 1 + 1);
    EXPECT_EQ(meta.object[1].index_object_id, 2);
}

// ---- probes used by tests/test_host_layer.py to look at the files / streams the host layer produces ----
#include "../../cunvsm_amd/host/hdf5_writer.hpp"
#include <fstream>
static int probe(int argc, char** argv) {
    const std::string what = argv[1];
    if (what == "--write-meta" && argc >= 3) {
        Metadata m;
        for (int i = 0; i < 5; ++i) m.term.push_back({100 + i, i, 7 * i + 1});
        m.term.push_back({-3, 5, 2147483647});
        for (int i = 0; i < 3; ++i) m.object.push_back({i + 1, i});
        m.total_terms = 260760;
        std::ofstream f(argv[2], std::ios::binary);
        const std::string w = m.SerializeAsString();
        f.write(w.data(), w.size());
        return f.good() ? 0 : 1;
    }
    if (what == "--write-hdf5" && argc >= 3) {
        std::vector<float> W(5 * 3), E(4 * 2), T(3 * 2), b(2);
        for (size_t i = 0; i < W.size(); ++i) W[i] = 0.5f + i;
        for (size_t i = 0; i < E.size(); ++i) E[i] = -1.f * i;
        for (size_t i = 0; i < T.size(); ++i) T[i] = 10.f + i;
        b = {0.25f, -0.75f};
        write_hdf5(argv[2], {{"entity_representations-representations", 4, 2, E.data()}, {"word_entity_mapping-bias", 1, 2, b.data()},
                             {"word_entity_mapping-transform", 3, 2, T.data()}, {"word_representations-representations", 5, 3, W.data()}});
        return 0;
    }
    if (what == "--index-stats" && argc >= 3) {     // JSON: collection statistics + the first windows of a sequential and a shuffled pass
        TrectextIndex* index = TrectextIndex::from_file(argv[2]);
        int64_t min_len = 1 << 30, max_len = 0;
        for (DOCID_T d = index->documentBase(); d < index->documentMaximum(); ++d) { min_len = std::min(min_len, index->documentLength(d)); max_len = std::max(max_len, index->documentLength(d)); }
        std::printf("{\"documents\": %lu, \"tokens\": %lu, \"unique\": %lu, \"min_len\": %ld, \"max_len\": %ld, ",
                    (unsigned long)index->documentCount(), (unsigned long)index->termCount(), (unsigned long)index->uniqueTermCount(), (long)min_len, (long)max_len);
        RNG rng; rng.seed(1);
        const uint64_t max_df = (index->documentCount() + 1) / 2;
        IndexSource source(index, 10, &rng, 60000, 2, max_df, 0, false, false, nullptr, nullptr, true, AUTOMATIC_SAMPLING, UNIFORM);
        Metadata meta; source.extract_metadata(&meta);
        std::printf("\"vocabulary\": %lu, \"corpus\": %lu, \"total_terms\": %d, ", (unsigned long)source.vocabulary_size(), (unsigned long)source.corpus_size(), meta.total_terms);
        Batch batch(4096, 10);
        size_t instances = 0, batches = 0; long checksum = 0;
        while (source.has_next()) { source.next(&batch); instances += batch.num_instances(); ++batches;
            for (size_t i = 0; i < batch.num_instances() * 10; ++i) checksum = (checksum * 31 + batch.features()[i]) % 1000000007L;
            batch.clear(); }
        std::printf("\"instances\": %lu, \"batches\": %lu, \"feature_checksum\": %ld}\n", (unsigned long)instances, (unsigned long)batches, checksum);
        return 0;
    }
    if (what == "--time-source" && argc >= 6) {
        // --time-source <trectext> <window> <batch> <epochs>: instances per second of the training data source alone
        // (shuffled epochs incl. the per-epoch reset, uniform weighting), single thread
        TrectextIndex* index = TrectextIndex::from_file(argv[2]);
        const size_t window = std::stoul(argv[3]), batch_size = std::stoul(argv[4]), epochs = std::stoul(argv[5]);
        RNG rng; rng.seed(1);
        const uint64_t max_df = static_cast<uint64_t>(std::ceil(index->documentCount() * 0.5));
        IndexSource source(index, window, &rng, 60000, 2, max_df, 0, false, false, nullptr, nullptr, true, AUTOMATIC_SAMPLING, UNIFORM);
        Batch batch(batch_size, window);
        size_t instances = 0; int64_t checksum = 0;
        double t_next = 0.0, t_reset = 0.0;
        for (size_t e = 0; e < epochs; ++e) {
            auto t0 = std::chrono::steady_clock::now();
            while (source.has_next()) {
                source.next(&batch);
                instances += batch.num_instances();
                checksum += batch.features()[0] + batch.labels()[batch.num_instances() - 1];
                batch.clear();
            }
            auto t1 = std::chrono::steady_clock::now();
            source.reset();
            auto t2 = std::chrono::steady_clock::now();
            t_next += std::chrono::duration<double>(t1 - t0).count();
            t_reset += std::chrono::duration<double>(t2 - t1).count();
        }
        std::printf("{\"instances\": %lu, \"next_seconds\": %.4f, \"reset_seconds\": %.4f, \"instances_per_second\": %.0f, \"checksum\": %ld}\n",
                    (unsigned long)instances, t_next, t_reset, instances / (t_next + t_reset), (long)checksum);
        return 0;
    }
    if (what == "--dump-epoch" && argc >= 10) {
        // --dump-epoch <trectext> <out.bin> <window> <batch> <seed> <max_vocab> <min_df> <cutoff>: the batches of the first
        // epoch exactly as cuNVSMTrainModel's data source produces them (shuffled, --weighting uniform), + the state of
        // the shared generator right after the source was built (= what the trainer initialises the model from)
        TrectextIndex* index = TrectextIndex::from_file(argv[2]);
        const size_t window = std::stoul(argv[4]), batch_size = std::stoul(argv[5]);
        RNG rng; rng.seed(std::stoul(argv[6]));
        const uint64_t max_df = static_cast<uint64_t>(std::ceil(index->documentCount() * 0.5));
        IndexSource source(index, window, &rng, std::stoul(argv[7]), std::stoul(argv[8]), max_df, std::stoul(argv[9]), false, false,
                           nullptr, nullptr, true, AUTOMATIC_SAMPLING, UNIFORM);
        std::stringstream st; st << rng;
        std::ofstream f(argv[3], std::ios::binary);
        Batch batch(batch_size, window);
        size_t nb = 0;
        while (source.has_next()) {
            source.next(&batch);
            const int64_t n = static_cast<int64_t>(batch.num_instances());
            f.write(reinterpret_cast<const char*>(&n), 8);
            f.write(reinterpret_cast<const char*>(batch.features()), n * window * 8);
            f.write(reinterpret_cast<const char*>(batch.feature_weights()), n * window * 4);
            f.write(reinterpret_cast<const char*>(batch.labels()), n * 8);
            f.write(reinterpret_cast<const char*>(batch.weights()), n * 4);
            batch.clear();
            ++nb;
        }
        std::printf("{\"rng_state\": %s, \"vocabulary\": %lu, \"corpus\": %lu, \"batches\": %lu}\n", st.str().c_str(),
                    (unsigned long)source.vocabulary_size(), (unsigned long)source.corpus_size(), (unsigned long)nb);
        return f.good() ? 0 : 1;
    }
    return 2;
}

// ---- cpp/data_tests.cpp:623-683: the Indri 5.8 repository the reference ships (test_data/Brown_index) ----
static std::string g_brown_path;
static void test_IndriSource_Brown() {
    if (g_brown_path.empty()) { std::printf("    (skipped: NVSM_BROWN_INDEX not set)\n"); return; }
    IndriDiskIndex* index = IndriDiskIndex::open(g_brown_path);
    EXPECT_EQ(index->documentCount(), 500u);
    EXPECT_EQ(index->uniqueTermCount(), 29980u);
    EXPECT_EQ(index->termCount(), 1032531u);
    EXPECT_EQ(index->documentLength(1), 2032);
    EXPECT_EQ(index->term(TERMID_T(1)), std::string("time"));
    EXPECT_EQ(index->term(std::string("jury")) > 10, true);
    RNG rng;
    IndexSource source(index, 16, &rng, 0, 0, 0, 0, false, false, nullptr, nullptr, true, AUTOMATIC_SAMPLING, UNIFORM);
    EXPECT_EQ(source.corpus_size(), 500u);
    size_t i = 0;
    for (const auto& pair : source.document_id_mapping()) { EXPECT_EQ(pair.first, i); EXPECT_EQ(pair.second, static_cast<DOCID_T>(i + 1)); ++i; }
    Batch batch(4, 16);
    EXPECT_TRUE(source.has_next());
    source.next(&batch);
    std::map<size_t, std::string> str_instances;
    for (size_t b = 0; b < batch.num_instances(); ++b) {
        std::string tmp;
        for (size_t j = 0; j < 16; ++j) tmp += source.term(batch.features()[b * 16 + j]) + " ";
        str_instances[static_cast<size_t>(source.document_id_mapping().at(static_cast<size_t>(batch.labels()[b])))] = tmp;
    }
    const std::map<size_t, std::string> want = {
        {405, "kept signal dowl car coming steady clear start back hamburger shut device want hang eat dont "},
        {215, "conspire uncle make secret gift money mother story end child illness delirium brought feverish compulsion ride "},
        {434, "morgan im usually strong woman im awfully tired hungry start meal eat meal girl cry morgan "},
        {392, "write pleasant note beautiful last life part ways goodbye forever word fifty dollar add postscript beg "}};
    EXPECT_TRUE(str_instances == want);
    if (!(str_instances == want)) for (const auto& kv : str_instances) std::printf("      got %zu: %s\n", kv.first, kv.second.c_str());
}

// ---- --document_list on an Indri repository: QueryEnvironment::documentIDsFromMetadata("docno", …) (cpp/data_indri.cpp:695)
// and retrieveMetadatum(doc, "docno") (:571-589) through the repository's own docno key files. The Brown repository holds
// the 500 Brown-corpus files ca01 … cr09 in that order as documents 1 … 500. ----
static void test_IndriRepository_docno_lookups() {
    if (g_brown_path.empty()) { std::printf("    (skipped: NVSM_BROWN_INDEX not set)\n"); return; }
    std::unique_ptr<IndriDiskIndex> index(IndriDiskIndex::open(g_brown_path));
    const std::vector<DOCID_T> ids = index->documentIDsFromDocno({"ca01", "cj75", "cr09", "ca44", "cb01", "cj76"});
    EXPECT_VEC(ids, 1, 369, 500, 44, 45, 370);
    EXPECT_EQ(index->docno(1), std::string("ca01"));
    EXPECT_EQ(index->docno(369), std::string("cj75"));
    EXPECT_EQ(index->docno(500), std::string("cr09"));
    // every document: forward and reverse files agree
    for (DOCID_T d = index->documentBase(); d < index->documentMaximum(); ++d) {
        const std::vector<DOCID_T> back = index->documentIDsFromDocno({index->docno(d)});
        EXPECT_EQ(back.size(), 1u);
        if (back.size() == 1) EXPECT_EQ(back[0], d);
    }
    bool died = false;
    try { index->documentIDsFromDocno({"zz99"}); } catch (const FatalError&) { died = true; }
    EXPECT_TRUE(died);

    // a document list selects and ORDERS the corpus (model ids follow the list), cpp/data_indri.cpp:693-717
    RNG rng;
    const std::vector<std::string> list = {"cr09", "ca01", "cj75"};
    IndexSource source(IndriDiskIndex::open(g_brown_path), 16, &rng, 0, 0, 0, 0, false, false, &list, nullptr, false, NONE);
    EXPECT_EQ(source.corpus_size(), 3u);
    const IndexSource::DocumentIdMapping want_docs = {{0, 500}, {1, 1}, {2, 369}};
    EXPECT_TRUE(source.document_id_mapping() == want_docs);
    const std::map<std::string, int64_t> by_docno = source.build_document_identifiers_map();
    const std::map<std::string, int64_t> want_by_docno = {{"cr09", 0}, {"ca01", 1}, {"cj75", 2}};
    EXPECT_TRUE(by_docno == want_by_docno);
}

// ---- the RCCL id rendezvous of a data-parallel run (no reference counterpart: host/rendezvous.hpp) ----
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
static void test_Rendezvous() {
    char dir_t[] = "/tmp/nvsm_rdv_XXXXXX";
    const char* dir = mkdtemp(dir_t);
    EXPECT_TRUE(dir != nullptr);
    if (!dir) return;
    const std::string path = std::string(dir) + "/comm", nonce = "run:abc";
    char id[kCommIdBytes], got[kCommIdBytes];
    for (int i = 0; i < kCommIdBytes; ++i) id[i] = static_cast<char>(i * 7 + 3);
    std::string why;
    const int64_t t0 = wall_clock_ns();
    EXPECT_TRUE(!rendezvous_read(path, nonce, t0, got, &why));                    // nothing there yet
    rendezvous_publish(path, nonce, id);
    EXPECT_TRUE(rendezvous_read(path, nonce, t0, got, &why));
    EXPECT_TRUE(std::memcmp(id, got, kCommIdBytes) == 0);
    struct stat st;
    EXPECT_TRUE(::stat(path.c_str(), &st) == 0 && (st.st_mode & 0777) == 0600);
    // a leftover of an earlier run: older than this reader, or under another run's nonce
    EXPECT_TRUE(!rendezvous_read(path, nonce, wall_clock_ns() + 1000000000, got, &why));
    EXPECT_EQ(why, std::string("is older than this run"));
    EXPECT_TRUE(!rendezvous_read(path, "run:other", t0, got, &why));
    EXPECT_EQ(why, std::string("belongs to another run (nonce)"));
    // a file whose writer has died (a crashed run relaunched from the same shell: same fall-back nonce): written by a child that
    // has exited and been reaped — its pid names no process
    rendezvous_clear(path);
    {
        const pid_t child = ::fork();
        if (child == 0) { rendezvous_publish(path, nonce, id); ::_exit(0); }
        int status = 0;
        EXPECT_TRUE(child > 0 && ::waitpid(child, &status, 0) == child);
        EXPECT_TRUE(!rendezvous_read(path, nonce, t0, got, &why));
        EXPECT_EQ(why, std::string("was written by a process that no longer exists (stale)"));
    }
    // what the old protocol wrote (the bare 128 bytes) is not taken for an id
    rendezvous_clear(path);
    { const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT, 0600); EXPECT_TRUE(fd >= 0 && ::write(fd, id, kCommIdBytes) == kCommIdBytes); ::close(fd); }
    EXPECT_TRUE(!rendezvous_read(path, nonce, t0, got, &why));
    // readable by others: somebody else could have written it
    rendezvous_clear(path);
    rendezvous_publish(path, nonce, id);
    EXPECT_TRUE(::chmod(path.c_str(), 0644) == 0);
    EXPECT_TRUE(!rendezvous_read(path, nonce, t0, got, &why));
    EXPECT_EQ(why, std::string("is accessible to group / others"));
    // a symbolic link in its place is not followed, by the reader or by the writer's temporary file
    rendezvous_clear(path);
    const std::string target = std::string(dir) + "/elsewhere";
    rendezvous_publish(target, nonce, id);
    EXPECT_TRUE(::symlink(target.c_str(), path.c_str()) == 0);
    EXPECT_TRUE(!rendezvous_read(path, nonce, t0, got, &why));
    EXPECT_EQ(why, std::string("is a symbolic link"));
    rendezvous_publish(path, nonce, id);                                            // rename replaces the link itself
    EXPECT_TRUE(::lstat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode));
    EXPECT_TRUE(rendezvous_read(path, nonce, t0, got, &why));
    // the default location is private to the user, and two runs (nonces) do not share a name
    const std::string a = default_comm_id_path("ppid:1:port:29500"), b = default_comm_id_path("ppid:2:port:29500");
    EXPECT_TRUE(a != b);
    const std::string adir = a.substr(0, a.rfind('/'));
    EXPECT_TRUE(::stat(adir.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 077) == 0);
    EXPECT_EQ(comm_run_nonce("given"), std::string("given"));
    rendezvous_clear(path); rendezvous_clear(target);
    ::rmdir(dir);
}

int main(int argc, char** argv) {
    log_to_stderr() = false;
    if (const char* e = std::getenv("NVSM_BROWN_INDEX")) g_brown_path = e;
    if (argc > 1 && std::string(argv[1]).compare(0, 2, "--") == 0) {
        try { return probe(argc, argv); } catch (const std::exception& e) { std::printf("probe failed: %s\n", e.what()); return 1; }
    }
    const std::vector<std::pair<const char*, std::function<void()>>> tests = {
        {"InMemoryDocumentSource.InMemoryDocumentSource", test_InMemoryDocumentSource},
        {"InMemoryDocumentSource.pad_batch", test_InMemoryDocumentSource_pad_batch},
        {"DataSourceTest.create_instances", test_create_instances},
        {"DataSourceTest.create_instances_overflow", test_create_instances_overflow},
        {"IndriSourceTest.IndriSource", test_IndriSource},
        {"IndriSourceTest.IndriSource_UnsupportedSampling_Death", test_IndriSource_UnsupportedSampling_Death},
        {"IndriSourceTest.StochasticIndriSource", test_StochasticIndriSource},
        {"IndriSourceTest.StochasticIndriSource_Resampling", test_StochasticIndriSource_Resampling},
        {"IndriSourceTest.StochasticIndriSource_SelfInformation", test_StochasticIndriSource_SelfInformation},
        {"MetaSourceTest.AsyncSource", test_AsyncSource},
        {"MetaSourceTest.RepeatingSource", test_RepeatingSource},
        {"Base.utils", test_utils},
        {"Batch.swap", test_Batch_swap},
        {"Metadata.roundtrip", test_Metadata_roundtrip},
        {"TrectextIndex.end_to_end", test_TrectextIndex},
        {"IndriSourceTest.Brown", test_IndriSource_Brown},
        {"IndriRepository.docno_lookups", test_IndriRepository_docno_lookups},
        {"DataParallel.rendezvous_file", test_Rendezvous},
    };
    int failed_tests = 0;
    for (const auto& t : tests) {
        if (argc > 1 && std::string(argv[1]) != t.first) continue;
        const int before = g_failures;
        try { t.second(); }
        catch (const std::exception& e) { std::printf("    exception: %s\n", e.what()); ++g_failures; }
        const bool ok = g_failures == before;
        std::printf("[%s] %s\n", ok ? "PASS" : "FAIL", t.first);
        failed_tests += !ok;
    }
    std::printf("%d failed\n", failed_tests);
    return failed_tests ? 1 : 0;
}
