// Host orchestration of the NVSM / LSE step on MI355X. See model.h.
#include "model.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace cunvsm {

// ---------------------------------------------------------------------------------------------
// RCCL through dlopen (rccl.h types restated minimally; ABI of RCCL 2.x / NCCL 2.x)
// ---------------------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    typedef struct { char internal[128]; } UniqueId;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;      // (send, recv, count per rank, type, comm, stream)
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    static constexpr int kInt8 = 0, kFloat32 = 7, kFloat64 = 8, kSum = 0;   // ncclInt8, ncclFloat32, ncclFloat64, ncclSum

    static RcclApi* load() {
        static RcclApi api;
        if (api.handle) return &api;
        // a copy the process already holds (PyTorch's, loaded as "librccl.so") is shared; otherwise ROCm's
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (api.handle) break;
        }
        for (const char* n : names) {
            if (api.handle) break;
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.handle) throw Error(NVSM_ERR_DEVICE, std::string("cannot load librccl: ") + dlerror());
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.handle, "ncclCommCount"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy)
            throw Error(NVSM_ERR_DEVICE, "librccl lacks the expected nccl* symbols");
        return &api;
    }
};

// 1-rank communicator + the two all-reduce flavours of the step, on a private stream (nvsm_comm_selftest)
void rccl_selftest(int device) {
    RcclApi* api = RcclApi::load();
    NVSM_HIP_CHECK(hipSetDevice(device));
    RcclApi::UniqueId u;
    if (api->GetUniqueId(&u) != 0) throw Error(NVSM_ERR_DEVICE, "ncclGetUniqueId failed");
    void* comm = nullptr;
    if (api->CommInitRank(&comm, 1, u, 0) != 0) throw Error(NVSM_ERR_DEVICE, "ncclCommInitRank(1 rank) failed");
    hipStream_t s;
    NVSM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 1024;
    DevBuf<float> f; DevBuf<double> d;
    f.alloc(n); d.alloc(n);
    std::vector<float> hf(n); std::vector<double> hd(n);
    for (int i = 0; i < n; ++i) { hf[i] = 0.5f * i; hd[i] = 0.25 * i; }
    NVSM_HIP_CHECK(hipMemcpy(f.p, hf.data(), n * sizeof(float), hipMemcpyHostToDevice));
    NVSM_HIP_CHECK(hipMemcpy(d.p, hd.data(), n * sizeof(double), hipMemcpyHostToDevice));
    int rc = api->AllReduce(f.p, f.p, n, RcclApi::kFloat32, RcclApi::kSum, comm, s);
    if (rc == 0) rc = api->AllReduce(d.p, d.p, n, RcclApi::kFloat64, RcclApi::kSum, comm, s);
    // as nvsm_step does with world_size > 1: the next collective of the same communicator on a second, lower-priority
    // stream ordered behind the first by an event, then back on the first stream
    hipStream_t s2; hipEvent_t e1, e2;
    int lo = 0, hi = 0;
    NVSM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    NVSM_HIP_CHECK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
    NVSM_HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    NVSM_HIP_CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    NVSM_HIP_CHECK(hipEventRecord(e1, s));
    NVSM_HIP_CHECK(hipStreamWaitEvent(s2, e1, 0));
    if (rc == 0) rc = api->AllReduce(f.p, f.p, n, RcclApi::kFloat32, RcclApi::kSum, comm, s2);
    NVSM_HIP_CHECK(hipEventRecord(e2, s2));
    NVSM_HIP_CHECK(hipStreamWaitEvent(s, e2, 0));
    if (rc == 0) rc = api->AllReduce(d.p, d.p, n, RcclApi::kFloat64, RcclApi::kSum, comm, s);
    // the byte all-gather of the exact-tables mode: one rank's gather is a copy
    DevBuf<float> g;
    g.alloc(n);
    if (rc == 0 && api->AllGather) rc = api->AllGather(f.p, g.p, n * sizeof(float), RcclApi::kInt8, comm, s);
    NVSM_HIP_CHECK(hipStreamSynchronize(s));
    NVSM_HIP_CHECK(hipStreamSynchronize(s2));
    (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipStreamDestroy(s2);
    if (rc == 0 && api->AllGather) {
        std::vector<float> rg(n);
        NVSM_HIP_CHECK(hipMemcpy(rg.data(), g.p, n * sizeof(float), hipMemcpyDeviceToHost));
        if (rg != hf) { api->CommDestroy(comm); (void)hipStreamDestroy(s); throw Error(NVSM_ERR_DEVICE, "1-rank all-gather did not return its input"); }
    }
    std::vector<float> rf(n); std::vector<double> rd(n);
    NVSM_HIP_CHECK(hipMemcpy(rf.data(), f.p, n * sizeof(float), hipMemcpyDeviceToHost));
    NVSM_HIP_CHECK(hipMemcpy(rd.data(), d.p, n * sizeof(double), hipMemcpyDeviceToHost));
    api->CommDestroy(comm);
    (void)hipStreamDestroy(s);
    if (rc != 0) throw Error(NVSM_ERR_DEVICE, "ncclAllReduce failed in the self-test");
    if (rf != hf || rd != hd) throw Error(NVSM_ERR_DEVICE, "1-rank all-reduce did not return its input (wrong dtype / op enum?)");
}

// nvsm_comm_latency: the step's three collectives on a 1-rank communicator, timed one kind at a time (HIP events around
// `repeats` back-to-back calls on a stream of the handle's kind): what a call costs before any wire time
void rccl_latency(int device, int de, int dw, int repeats, float us[3], int64_t bytes[3]) {
    if (de <= 0 || dw <= 0 || repeats <= 0) throw Error(NVSM_ERR_INVALID_ARGUMENT, "dimensions and repeats must be positive");
    RcclApi* api = RcclApi::load();
    NVSM_HIP_CHECK(hipSetDevice(device));
    RcclApi::UniqueId u;
    if (api->GetUniqueId(&u) != 0) throw Error(NVSM_ERR_DEVICE, "ncclGetUniqueId failed");
    void* comm = nullptr;
    if (api->CommInitRank(&comm, 1, u, 0) != 0) throw Error(NVSM_ERR_DEVICE, "ncclCommInitRank(1 rank) failed");
    hipStream_t s;
    NVSM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // (out of place: an in-place all-reduce over one rank is a no-op that RCCL returns from without launching anything — 0.02 us;
    //  out of place it launches its copy kernel, which is the launch + kernel floor a real collective starts from)
    DevBuf<double> d, d2; DevBuf<float> f, f2;
    d.alloc(static_cast<size_t>(1 + 2 * de), true); f.alloc(static_cast<size_t>(de) * dw, true);
    d2.alloc(static_cast<size_t>(1 + 2 * de), true); f2.alloc(static_cast<size_t>(de) * dw, true);
    hipEvent_t e0, e1;
    NVSM_HIP_CHECK(hipEventCreate(&e0)); NVSM_HIP_CHECK(hipEventCreate(&e1));
    const size_t counts[3] = {static_cast<size_t>(2 * de), static_cast<size_t>(1 + 2 * de), static_cast<size_t>(de) * dw};
    int rc = 0;
    for (int k = 0; k < 3; ++k) {
        auto call = [&] {
            return k < 2 ? api->AllReduce(d.p, d2.p, counts[k], RcclApi::kFloat64, RcclApi::kSum, comm, s)
                         : api->AllReduce(f.p, f2.p, counts[k], RcclApi::kFloat32, RcclApi::kSum, comm, s);
        };
        for (int i = 0; i < 3 && rc == 0; ++i) rc = call();          // (first-use set-up out of the way)
        NVSM_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < repeats && rc == 0; ++i) rc = call();
        NVSM_HIP_CHECK(hipEventRecord(e1, s));
        NVSM_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        NVSM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        us[k] = ms * 1e3f / static_cast<float>(repeats);
        bytes[k] = static_cast<int64_t>(counts[k] * (k < 2 ? sizeof(double) : sizeof(float)));
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    api->CommDestroy(comm);
    (void)hipStreamDestroy(s);
    if (rc != 0) throw Error(NVSM_ERR_DEVICE, "ncclAllReduce failed in the latency probe");
}

void rccl_unique_id(char id[128]) {
    RcclApi* api = RcclApi::load();
    RcclApi::UniqueId u;
    const int rc = api->GetUniqueId(&u);
    if (rc != 0) throw Error(NVSM_ERR_DEVICE, "ncclGetUniqueId failed");
    std::memcpy(id, u.internal, 128);
}

// ---------------------------------------------------------------------------------------------
// roctx ranges (the reference brackets Epoch / Batch / FetchData / ComputeCost / ComputeGradients / UpdateParameters with
// nvtxRangePush/Pop, cpp/main.cu:386-431, and every function with PROFILE_FUNCTION): resolved at run time from
// librocprofiler-sdk-roctx so that a rocprofv3 --marker-trace run shows the same ranges plus one per kernel group.
// Without the library (or with NVSM_ROCTX=0) the calls are no-ops.
// ---------------------------------------------------------------------------------------------
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    static Roctx& get() {
        static Roctx r = [] {
            Roctx x;
            if (!tuning().roctx) return x;      // (process-wide: decided by the first caller's switches)
            void* h = nullptr;
            for (const char* n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1",
                                  "libroctx64.so", "libroctx64.so.4"}) {
                h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
            if (!h) return x;
            x.push = reinterpret_cast<decltype(x.push)>(dlsym(h, "roctxRangePushA"));
            x.pop = reinterpret_cast<decltype(x.pop)>(dlsym(h, "roctxRangePop"));
            if (!x.push || !x.pop) x.push = nullptr, x.pop = nullptr;
            return x;
        }();
        return r;
    }
};
void range_push(const char* name) { Roctx& r = Roctx::get(); if (r.push) r.push(name); }
void range_pop() { Roctx& r = Roctx::get(); if (r.pop) r.pop(); }
struct RangeScope {
    explicit RangeScope(const char* name) { range_push(name); }
    ~RangeScope() { range_pop(); }
};

// ---------------------------------------------------------------------------------------------
// Profiler: HIP events on the handle's stream around each kernel group
// ---------------------------------------------------------------------------------------------
void Profiler::begin(const char* name, hipStream_t s) {
    if (!enabled) return;
    if (!only.empty()) {                       // comma-separated list of the kernel groups to time
        const size_t len = std::strlen(name);
        bool hit = false;
        for (size_t pos = 0; pos <= only.size() && !hit;) {
            size_t end = only.find(',', pos);
            if (end == std::string::npos) end = only.size();
            hit = (end - pos == len) && only.compare(pos, len, name) == 0;
            pos = end + 1;
        }
        if (!hit) return;
    }
    Slot& sl = slots_[name];
    if (sl.used == sl.ev.size()) {
        hipEvent_t a, b;
        NVSM_HIP_CHECK(hipEventCreate(&a));
        NVSM_HIP_CHECK(hipEventCreate(&b));
        sl.ev.emplace_back(a, b);
    }
    cur_ = &sl;
    NVSM_HIP_CHECK(hipEventRecord(sl.ev[sl.used].first, s));
}
bool Profiler::selected(const char* name) const {
    if (!enabled) return false;
    if (only.empty()) return true;
    const size_t len = std::strlen(name);
    for (size_t pos = 0; pos <= only.size();) {
        size_t end = only.find(',', pos);
        if (end == std::string::npos) end = only.size();
        if ((end - pos == len) && only.compare(pos, len, name) == 0) return true;
        pos = end + 1;
    }
    return false;
}
bool Profiler::bind(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    if (!selected(name)) return false;
    Slot& sl = slots_[name];
    if (sl.used == sl.ev.size()) {
        hipEvent_t a, b;
        NVSM_HIP_CHECK(hipEventCreate(&a));
        NVSM_HIP_CHECK(hipEventCreate(&b));
        sl.ev.emplace_back(a, b);
    }
    *start = sl.ev[sl.used].first; *stop = sl.ev[sl.used].second;
    sl.used++;
    return true;
}
void Profiler::end(hipStream_t s) {
    if (!enabled || !cur_) return;
    NVSM_HIP_CHECK(hipEventRecord(cur_->ev[cur_->used].second, s));
    cur_->used++;
    cur_ = nullptr;
}
void Profiler::reset() {
    for (auto& kv : slots_) kv.second.used = 0;
    notes_.clear();
}
std::vector<std::string> Profiler::names() const {
    std::vector<std::string> v;
    for (auto& kv : slots_) v.push_back(kv.first);
    for (auto& kv : notes_) v.push_back(kv.first);
    return v;
}
bool Profiler::get(const std::string& name, double* ms, int64_t* launches) {
    auto nt = notes_.find(name);
    if (nt != notes_.end()) { *ms = 0.0; *launches = nt->second; return true; }
    auto it = slots_.find(name);
    if (it == slots_.end()) return false;
    double total = 0.0;
    for (size_t i = 0; i < it->second.used; ++i) {
        NVSM_HIP_CHECK(hipEventSynchronize(it->second.ev[i].second));
        float t = 0.f;
        NVSM_HIP_CHECK(hipEventElapsedTime(&t, it->second.ev[i].first, it->second.ev[i].second));
        total += t;
    }
    *ms = total;
    *launches = static_cast<int64_t>(it->second.used);
    return true;
}
Profiler::~Profiler() {
    for (auto& kv : slots_)
        for (auto& e : kv.second.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
}

struct ProfScope {
    Profiler& p; hipStream_t s;
    ProfScope(Profiler& p_, const char* name, hipStream_t s_) : p(p_), s(s_) { range_push(name); p.begin(name, s); }
    ~ProfScope() { p.end(s); range_pop(); }
};
#define PROF(name) ProfScope _prof_scope(prof, name, stream_)
#define PROF_ON(name, strm) ProfScope _prof_scope(prof, name, strm)

// kernels.h: the event the next NVSM_LAUNCH of this host thread carries as its completion event
static thread_local hipEvent_t tl_start_event = nullptr, tl_stop_event = nullptr;
void set_launch_events(hipEvent_t start, hipEvent_t stop) { tl_start_event = start; tl_stop_event = stop; }
hipEvent_t take_launch_events(hipEvent_t* start) {
    hipEvent_t e = tl_stop_event;
    if (start) *start = tl_start_event;
    tl_start_event = tl_stop_event = nullptr;
    return e;
}
static bool stop_events_enabled() {
    return tuning().stop_events;
}
// `launch` enqueues ONE kernel on `s` through NVSM_LAUNCH; `ev` then stands for everything queued on `s` up to and including
// it, exactly as a hipEventRecord behind it would (NVSM_STOP_EVENTS=0: that plain record, for A/B runs)
template <class F>
static void launch_and_record(hipEvent_t ev, hipStream_t s, F&& launch) {
    const bool bound = stop_events_enabled();
    if (bound) set_launch_events(nullptr, ev);
    launch();
    if (!bound) { NVSM_HIP_CHECK(hipEventRecord(ev, s)); return; }
    if (hipEvent_t left = take_launch_events(nullptr)) NVSM_HIP_CHECK(hipEventRecord(left, s));      // nothing was launched
}

// A kernel group that is ONE launch through NVSM_LAUNCH, timed by the profiler: its event pair rides on the launch as start /
// stop events (the kernel's own execution time, nothing queued around it); `single` false: the plain records around the group.
template <class F>
static void timed_launch(Profiler& prof, const char* name, hipStream_t s, bool single, F&& launch) {
    hipEvent_t a = nullptr, b = nullptr;
    if (single && stop_events_enabled() && prof.bind(name, &a, &b)) {
        RangeScope r(name);
        set_launch_events(a, b);
        launch();
        hipEvent_t ls = nullptr;
        if (hipEvent_t le = take_launch_events(&ls)) {      // nothing was launched: an empty interval
            NVSM_HIP_CHECK(hipEventRecord(ls, s));
            NVSM_HIP_CHECK(hipEventRecord(le, s));
        }
        return;
    }
    ProfScope scope(prof, name, s);
    launch();
}

// ---------------------------------------------------------------------------------------------
// NVSM_CHUNK_ORDER=0 (A/B runs): level-1 chunks as numbered instead of in batch order
static bool chunk_order_enabled() {
    return tuning().chunk_order;
}
static int bits_for(int64_t n) {
    int b = 1;
    while ((int64_t(1) << b) < n) ++b;
    return b;
}

// whether a table of this size, updated from at most `max_entries` entries per step, decays lazily (see alloc_table)
bool Model::table_decays_lazily(bool documents, int64_t rows, int dim, int64_t max_entries) const {
    const int method = cfg_.update_method, mode = cfg_.adam_mode;
    const bool sparse_adam = method == NVSM_ADAM && mode <= NVSM_ADAM_SPARSE;
    const bool decays = sparse_adam || (method != NVSM_ADAM && cfg_.regularization_lambda > 0.f);
    const double lazy_min_mb = tune_.lazy_min_mb >= 0.0 ? tune_.lazy_min_mb : (sparse_adam ? 96.0 : 384.0);
    const double state_mb = static_cast<double>(rows) * dim * sizeof(float) * (method == NVSM_ADAM ? 2.0 : 1.0) / 1048576.0;
    const int tab_mask = tune_.lazy_tables;          // experiments: bit 0 = words, bit 1 = documents
    return tune_.lazy_decay && decays && state_mb >= lazy_min_mb && ((tab_mask >> (documents ? 1 : 0)) & 1) &&
           static_cast<double>(rows) * table_split_ratio() >= static_cast<double>(max_entries);
}

void Model::alloc_table(TableState& t, int64_t rows, int dim, int64_t max_entries) {
    t.rows = rows; t.dim = dim; t.max_entries = max_entries;
    t.P.alloc(rows * dim, true);
    const int method = cfg_.update_method, mode = cfg_.adam_mode;
    if (method == NVSM_ADAGRAD) {
        t.sc[0].alloc(rows, true); t.sc[1].alloc(rows, true);
    } else if (method == NVSM_ADAM) {
        t.m.alloc(rows * dim, true);
        if (mode == NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE) t.vfull.alloc(rows * dim, true);
        else { t.sc[0].alloc(rows, true); t.sc[1].alloc(rows, true); }
    }
    // (a long row of c entries has at most c / chunk + 1 chunks and there are fewer than n / chunk long rows: at most 2 n / chunk
    //  chunks; SGD / Adagrad handles cut the long rows of their small batches into shorter chunks — chunk_entries())
    {
        int64_t c1 = 2 * max_entries / kChunk;
        if (method != NVSM_ADAM && dim <= 128) c1 = std::max<int64_t>(c1, 2 * std::min<int64_t>(max_entries, kChunkSmallMaxEntries) / kChunkSmall);
        t.max_chunks = static_cast<int>(c1 + 2);
        t.max_chunks2 = static_cast<int>(c1 / kFan + 2);
    }
    for (int k = 0; k < t.idx_sets; ++k) {
        TableState::CsrIndex& x = t.idx[k];
        x.sorted_key.alloc(max_entries); x.sorted_entry.alloc(max_entries);
        x.csr_zeroed.alloc(csr_counter_ints(rows), true); x.chunk_base.alloc(rows, true);
        x.touched.alloc(std::min<int64_t>(rows, max_entries));
        x.chunk_desc.alloc(static_cast<size_t>(t.max_chunks) * 3, true);
        x.chunk2_base.alloc(rows, true);
        x.chunk2_desc.alloc(static_cast<size_t>(t.max_chunks2) * 2, true);
        // (batches of a few thousand windows are launch-latency chains: no extra launches there)
        // (from 60 x 4096 entries: the 2-GPU share of the metric's batch, 25 600 windows of ten words, 0.5428 -> 0.5363 ms with the order;
        //  16 384 windows: 0.425 -> 0.429 the other way; 12 800: nothing)
        if (chunk_order_enabled() && max_entries >= 60 * 4096) x.chunk_order.alloc(t.max_chunks, true);
    }
    if (t.idx[0].chunk_order.p) { t.chunk_key.alloc(t.max_chunks, true); t.chunk_key_sorted.alloc(t.max_chunks, true); }
    t.partial.alloc(static_cast<size_t>(t.max_chunks) * dim);
    t.partial_q.alloc(t.max_chunks, true);
    t.partial2.alloc(static_cast<size_t>(t.max_chunks2) * dim);
    t.partial2_q.alloc(t.max_chunks2, true);
    t.arrive_row.alloc(rows, true); t.arrive2.alloc(t.max_chunks2, true);      // zero once: the last arriver resets its counter
    {
        const bool sparse_adam = method == NVSM_ADAM && mode <= NVSM_ADAM_SPARSE;
        // When lazy decay pays. The dense passes it saves must cost more than what it adds (a snapshot and a stamp launch per
        // update on the table's stream, a stamp load per gathered row): tables of hundreds of MB (configs[4]) always; with
        // sparse Adam — two dense arrays per table, P and m — from about a hundred MB of state, which is what makes the
        // per-rank share of the 8-GPU metric (6 400 windows against 50 k / 100 k rows) lazy for both tables: 0.335 -> 0.292 ms
        // per step (words alone 0.299, documents alone 0.322; interleaved A/B). Not the LSE shape (Adagrad, batch 4096, a
        // 100 MB words table): 0.182 -> 0.193 ms with a lazy words table. NVSM_LAZY_MIN_MB overrides (tests use small tables).
        // (tune_.lazy_decay is per handle: tests build an eager twin)
        t.lazy = table_decays_lazily(&t == &ents_, rows, dim, max_entries);
        t.lazy_scalar = sparse_adam || (method == NVSM_ADAGRAD && &t == &ents_);
        if (t.lazy) t.stamp.alloc(rows, true);
        for (float& d : t.decay_hist) d = 1.f;
    }
    t.sort_bits = bits_for(rows);
    t.sort_temp_bytes = sort_pairs_temp_bytes(max_entries, t.sort_bits);
    t.sort_temp.alloc(t.sort_temp_bytes, true);      // zero once: the arrival counter only ever grows
}

void Model::alloc_sums(SumsBufs& b, int colgroups, int contrib_cap, int width_cap) {
    GridSumWs& w = b.ws;
    w.colgroups = colgroups; w.contrib_cap = contrib_cap; w.width_cap = width_cap;
    w.groups_cap = contrib_cap / 16 + 1;      // (grid_sum_fan() is at least 16)
    w.fan = 16;
    b.part.alloc(static_cast<size_t>(colgroups) * contrib_cap * width_cap);
    b.part2.alloc(static_cast<size_t>(colgroups) * w.groups_cap * width_cap);
    b.arrive.alloc(static_cast<size_t>(colgroups) * (w.groups_cap + 1), true);      // zero once: the last arriver resets its counter
    w.part = b.part.p; w.part2 = b.part2.p; w.arrive = b.arrive.p;
}

Model::Model(const nvsm_config& cfg) : tune_(Tuning::from_env()), cfg_(cfg), R_(cfg.num_random_entities + 1), rng_(1) {
    TuningScope tuning_scope(&tune_);      // (the only place the environment is read: the kernels' launchers see this handle's switches)
    auto bad = [](const std::string& m) { throw Error(NVSM_ERR_INVALID_ARGUMENT, m); };
    if (cfg.num_words <= 0 || cfg.num_entities <= 0) bad("num_words and num_entities must be positive");
    if (cfg.word_repr_size <= 0 || cfg.entity_repr_size <= 0) bad("representation sizes must be positive");
    if (cfg.window_size <= 0 || cfg.num_random_entities < 0) bad("window_size must be > 0 and num_random_entities >= 0");
    if (cfg.max_batch_size <= 0) bad("max_batch_size must be positive");
    if (cfg.regularization_lambda < 0) bad("regularization_lambda must be >= 0");          // cpp/storage.cu:62-63
    if (cfg.num_words >= (int64_t(1) << 31) || cfg.num_entities >= (int64_t(1) << 31)) bad("tables are limited to 2^31 rows");
    if (cfg.nonlinearity != NVSM_TANH && cfg.nonlinearity != NVSM_HARD_TANH) bad("nonlinearity not implemented");  // params.cu:444-445
    if (cfg.update_method < NVSM_SGD || cfg.update_method > NVSM_ADAM) bad("unknown update_method");
    if (cfg.adam_mode < NVSM_ADAM_NONE || cfg.adam_mode > NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE) bad("unknown adam_mode");
    if (cfg.entity_repr_size > 1024 || (cfg.entity_repr_size % 4 != 0 && cfg.entity_repr_size > 256))
        throw Error(NVSM_ERR_UNSUPPORTED, "entity_repr_size must be <= 1024 (multiple of 4) or <= 256");
    if (cfg.word_repr_size > 4096) throw Error(NVSM_ERR_UNSUPPORTED, "word_repr_size must be <= 4096");
    if (cfg.world_size < 1 || cfg.rank < 0 || cfg.rank >= cfg.world_size) bad("bad world_size / rank");
    const int64_t B = cfg.max_batch_size;
    if (cfg.window_size > 2048 || R_ > 2048 || B * std::max<int64_t>(cfg.window_size, R_) >= (int64_t(1) << 26))
        bad("window_size and num_random_entities + 1 must be <= 2048 and max_batch_size x max(window, negatives + 1) < 2^26");
    if (B * std::max<int64_t>(cfg.word_repr_size, cfg.entity_repr_size) >= (int64_t(1) << 32) ||
        B * R_ * cfg.entity_repr_size >= (int64_t(1) << 40))
        bad("max_batch_size too large for 32-bit work indexing");
    exact_ = cfg.dp_exact_tables != 0 && cfg.world_size > 1;
    const int64_t Bu = exact_ ? B * cfg.world_size : B;      // windows per table update
    if (exact_ && (Bu * std::max<int64_t>(cfg.window_size, R_) >= (int64_t(1) << 26) ||
                   Bu * std::max<int64_t>(cfg.word_repr_size, cfg.entity_repr_size) >= (int64_t(1) << 32)))
        bad("dp_exact_tables: world_size x max_batch_size exceeds the limits of a single batch");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(NVSM_ERR_NO_DEVICE, "no HIP device visible — cunvsm_amd has no CPU fallback");
    if (cfg.device < 0 || cfg.device >= ndev) bad("device ordinal out of range");
    NVSM_HIP_CHECK(hipSetDevice(cfg.device));
    {
        // the main stream carries the step's critical chain: highest priority; the side streams (sorts, documents update,
        // dT GEMM) lowest. Interleaved A/B: 1.248 vs 1.261 ms per step.
        int lo = 0, hi = 0;
        NVSM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        NVSM_HIP_CHECK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));
        own_stream_ = true;
        NVSM_HIP_CHECK(hipStreamCreateWithPriority(&aux_stream_, hipStreamNonBlocking, lo));
        // Side stream 2 (CSR builds, projection update, the dT product except for large batches of eagerly decayed tables): lowest
        // priority, like side stream 1. (Highest priority where it carries the dT product of a large batch next to lazily decayed
        // tables was worth 2 % there — |V| = 500 k, |D| = 2 M 1.703 -> 1.670 ms — as the only handle of its process, and cost 25 %
        // — 1.64 -> 2.03 — as the second handle of a process whose first was still alive: the runtime has few hardware queues per
        // priority level, and a handle's main stream and side stream 2 then shared one. Not done. NVSM_AUX2_PRIO, experiments build.)
        // The four streams are created HERE, in THIS order: with side stream 2 created later (behind the tables) every shape ran
        // 1.2 to 3.5 times as long — streams are mapped onto hardware queues in order of creation.
        const int aux2_prio = tune_.aux2_prio;   // 0 lowest, 1 middle, 2 highest
        NVSM_HIP_CHECK(hipStreamCreateWithPriority(&aux2_stream_, hipStreamNonBlocking, aux2_prio == 0 ? lo : (aux2_prio == 2 ? hi : (lo + hi) / 2)));
        // Four streams, not five: the runtime multiplexes streams onto four hardware queues, and with a fifth stream the
        // host-batch copies shared a queue with compute and stopped overlapping it (1.22 -> 1.7 ms per step with host
        // batches). The inputs' copies therefore ride on side stream 3 in front of the documents sort that needs them.
        const int aux3_prio = tune_.aux3_prio;   // 0 lowest, 1 middle, 2 highest
        NVSM_HIP_CHECK(hipStreamCreateWithPriority(&aux3_stream_, hipStreamNonBlocking, aux3_prio == 0 ? lo : (aux3_prio == 2 ? hi : (lo + hi) / 2)));
        copy_stream_ = aux3_stream_;
    }
    // Events that only order this device's streams among themselves are made without the system-scope fence of a default
    // event (hipEventDisableSystemFence: what it gives up — visibility to the host and to other devices — nobody asks of
    // them; the host-visible results travel behind stream synchronisations and the events of their copies): LSE batch 4096
    // 0.209 -> 0.203 ms per step, no effect at the NVSM shape. NVSM_EVENT_FENCE=0: default events, 2: device-scope release.
    // (Data parallel handles keep the default events: memory that peers write over xGMI is in play there, the gain is
    //  confined to launch-latency-bound shapes, and a multi-GPU node has not been available to measure on.)
    const int ev_fence_env = tune_.event_fence;
    const int ev_fence = ev_fence_env >= 0 ? ev_fence_env : (cfg.world_size > 1 ? 0 : 1);
    const unsigned dev_flags = hipEventDisableTiming | (ev_fence == 1 ? hipEventDisableSystemFence : 0u) | (ev_fence == 2 ? hipEventReleaseToDevice : 0u);
    NVSM_HIP_CHECK(hipEventCreateWithFlags(&ev_csr_ents_, dev_flags));
    NVSM_HIP_CHECK(hipEventCreateWithFlags(&ev_inputs_, dev_flags));
    NVSM_HIP_CHECK(hipEventCreateWithFlags(&ev_csr_, dev_flags));
    NVSM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&cost_host_), sizeof(double), hipHostMallocDefault));
    for (hipEvent_t* e : {&ev_loss_, &ev_dx_, &ev_bwdx_, &ev_E_done_, &ev_T_done_, &ev_step_begin_[0], &ev_step_begin_[1], &ev_gathered_, &ev_words_late_, &ev_cost_ready_, &ev_untouched_})
        NVSM_HIP_CHECK(hipEventCreateWithFlags(e, dev_flags));
    for (hipEvent_t* e : {&ev_copied_, &ev_host_ids_[0], &ev_host_ids_[1], &ev_cost_copied_})      // (the host waits on these)
        NVSM_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    NVSM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&err_host_), sizeof(int), hipHostMallocDefault));
    *err_host_ = 0;
    debug_ = tune_.debug;
    dp_single_stream_ = tune_.dp_t_on_main;
    if (exact_) dp_single_stream_ = true;      // (every collective of that mode is issued on the main stream)

    const int dw = cfg.word_repr_size, de = cfg.entity_repr_size, w = cfg.window_size;
    const int64_t N = B * R_;
    alloc_table(words_, cfg.num_words, dw, Bu * w);
    ents_.idx_sets = 2;
    alloc_table(ents_, cfg.num_entities, de, Bu * R_);
    T_.alloc(static_cast<size_t>(de) * dw, true); b_.alloc(de, true);
    if (cfg.update_method != NVSM_SGD) { s0T_.alloc(static_cast<size_t>(de) * dw, true); s0b_.alloc(de, true); }
    if (cfg.update_method == NVSM_ADAM) { s1T_.alloc(static_cast<size_t>(de) * dw, true); s1b_.alloc(de, true); }

    for (int p = 0; p < 2; ++p) { in_words_[p].alloc(B * w); in_labels_[p].alloc(B); in_wwts_[p].alloc(B * w); in_instw_[p].alloc(B); }
    in_ids64_.alloc(N);
    if (cfg.sampler == NVSM_SAMPLER_HOST_MINSTD)
        for (int p = 0; p < 2; ++p) NVSM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&host_ids_pin_[p]), N * sizeof(int64_t), hipHostMallocDefault));
    if (cfg.world_size > 1) { loss_tmp_.alloc(1, true); loss_red_.alloc(1, true); }
    widx_.alloc(B * w); ids_buf_[0].alloc(N); ids_buf_[1].alloc(N); ids_p_ = ids_buf_[0].p;
    phrase_.alloc(B * dw); phrase_alt_.alloc(B * dw); phrase_p_ = phrase_.p; pre_.alloc(B * de); proj_.alloc(B * de); dy_.alloc(B * de); gphrase_.alloc(B * dw);
    if (cfg.l2_normalize_phrase_reprs) { phrase_raw_.alloc(B * dw); phrase_norms_.alloc(B); }
    if (cfg.l2_normalize_entity_reprs) { grad_entity_.alloc(Bu * R_ * de); ge_msq_.alloc(Bu * R_); }
    coef_.alloc(N); probs_.alloc(N); pp_.alloc(B); msq_w_.alloc(B); msq_parts_.alloc(B * gemm_rowsq_parts(dw));
    if (cfg.update_method == NVSM_ADAM && cfg.adam_mode <= NVSM_ADAM_SPARSE) U_.alloc(Bu * dw);
    if (cfg.update_method == NVSM_ADAGRAD) scale_w_.alloc(Bu);
    if (exact_) {
        xg_ids_[0].alloc(Bu * R_); xg_ids_[1].alloc(Bu * R_); xg_ids_p_ = xg_ids_[0].p;
        xg_widx_.alloc(Bu * w); xg_wwts_.alloc(Bu * w);
        xg_proj_.alloc(Bu * de); xg_coef_.alloc(Bu * R_); xg_pp_.alloc(Bu); xg_gphrase_.alloc(Bu * dw); xg_msq_w_.alloc(Bu);
    }
    stats_.alloc(4 * de + 1, true); stats_fwd_ = stats_.p; stats_bwd_ = stats_.p + 2 * de;
    {
        // projection GEMM: one column group per column part (LDS-stationary kernel: <= 4, a workgroup per CU) or per 128-column
        // tile (tiled kernel: a contribution per 128-row tile); loss kernel: one workgroup per 4 .. 64 examples
        hipDeviceProp_t prop{};
        NVSM_HIP_CHECK(hipGetDeviceProperties(&prop, cfg.device));
        const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        num_cus_ = cus;
        // or per 32 rows and all columns at once (row-panel kernel, per-rank batch sizes)
        alloc_sums(sums_fwd_, std::max(4, (de + 127) / 128), std::max<int>(cus, static_cast<int>((B + 31) / 32)), std::max(2 * 160, 2 * de));
        alloc_sums(sums_bwd_, 1, static_cast<int>((B + 3) / 4), 2 * de + 1);
    }
    bn_mean_.alloc(de, true); bn_inv_std_.alloc(de, true); dbeta_.alloc(de, true); dgamma_.alloc(de, true);
    planes_fwd_.alloc(gemm_split_planes_bytes(de, dw), true); planes_bwd_.alloc(gemm_split_planes_bytes(dw, de), true);
    rplanes_fwd_.alloc(gemm_rsplit_planes_bytes(de, dw), true); rplanes_bwd_.alloc(gemm_rsplit_planes_bytes(dw, de), true);
    split_fwd_ = GemmSplitWs{planes_fwd_.p, planes_fwd_.n, false, rplanes_fwd_.p, rplanes_fwd_.n, false};
    split_bwd_ = GemmSplitWs{planes_bwd_.p, planes_bwd_.n, false, rplanes_bwd_.p, rplanes_bwd_.n, false};
    // (+ the tail of the data-parallel step's folded collective behind the projection gradient: db | loss hi | loss lo — backward_T)
    gT_.alloc(static_cast<size_t>(de) * dw + de + 2, true); gb_.alloc(de, true);
    // split-K slabs of the dT product: 128 at the 51 200-window batch (400 rows each); a per-rank batch of a few thousand
    // windows cut 128 ways is 600 workgroups of two 32-deep K tiles each — all prologue, epilogue and 39 MB of partials
    // (158 us next to the updates at batch 6 400: step 0.360 ms; 50 slabs 0.291, 25 slabs 0.295, 12 slabs 0.296, interleaved
    // A/B). 128 rows per slab at least. Large batches: 16 slabs — since round 3 the dT product starts behind the fused
    // batch-norm-backward / dx product instead of next to it and must still be through before the next projection product:
    // 16 slabs of 3 200 rows write 5 MB of partials instead of 39 (NVSM shape 0.976 ms with 128 slabs, 0.938 with 64, 0.929
    // with 32-48, 0.926 with 16, 1.000 with 8; batch 12 800: 0.442 with 100 slabs, 0.417 with 50, 0.405 with 8-24; interleaved
    // A/B). NVSM_DT_SLABS overrides.
    {
        // Round 5: per-rank batches in slabs of ~400 rows (16 at batch 6 400; was 50): the projection update now adds the slabs up
        // itself (TransformUpdateArgs::partial — every thread walks them), and the product sits on the chain the next forward
        // product waits for: batch 6 400 0.2746 ms with 50 slabs, 0.2655 with 25, 0.2635 with 16, 0.2706 with 8 (interleaved A/B).
        gemm_slabs_want_ = tune_.dt_slabs > 0 ? tune_.dt_slabs : (B > gemm_rows_max_m() ? 16 : static_cast<int>(std::min<int64_t>(24, std::max<int64_t>(8, B / 400))));
    }
    // the split-K dT kernel (gemm_dt.hip): at most a slab per two CUs (two workgroups per slab)
    dt_ok_ = gemm_dt_covers(dw, de, static_cast<int>(B));
    // A batch BELOW max_batch_size can need MORE slabs than the full one (slab lengths round up to whole K tiles: 6 400 rows cut
    // 64 ways are 50 slabs of 128, 6 144 rows are 64 slabs of 96), but never more than asked for: size for that bound.
    // (the split-K dT kernel's slabs — up to a slab per two CUs — only where a batch of this handle can take that kernel: a per-rank
    //  or small-batch handle needs gemm_slabs_want_ slabs, 2-7 MB instead of 39)
    const bool dt_possible = dt_ok_ && gemm_split_products() != 0 && B >= std::min<int64_t>(tune_.dt_min_batch, kDtMainMinBatch);
    const int slabs = std::max({gemm_slabs_want_, tune_.dtw_slabs, kDtwSlabs, dt_possible ? std::max(tune_.dt_slabs, num_cus_ / 2) : 0, 1});
    gT_partial_.alloc(static_cast<size_t>(slabs) * de * dw);
    NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
}

Model::~Model() {
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (aux_stream_) { (void)hipStreamSynchronize(aux_stream_); (void)hipStreamDestroy(aux_stream_); }
    if (aux2_stream_) { (void)hipStreamSynchronize(aux2_stream_); (void)hipStreamDestroy(aux2_stream_); }
    if (aux3_stream_) { (void)hipStreamSynchronize(aux3_stream_); (void)hipStreamDestroy(aux3_stream_); }
    if (ev_csr_ents_) (void)hipEventDestroy(ev_csr_ents_);
    if (ev_inputs_) (void)hipEventDestroy(ev_inputs_);
    if (ev_csr_) (void)hipEventDestroy(ev_csr_);
    for (hipEvent_t e : {ev_loss_, ev_dx_, ev_bwdx_, ev_E_done_, ev_T_done_, ev_copied_, ev_step_begin_[0], ev_step_begin_[1],
                         ev_host_ids_[0], ev_host_ids_[1], ev_gathered_, ev_words_late_, ev_cost_ready_, ev_cost_copied_, ev_untouched_}) if (e) (void)hipEventDestroy(e);
    if (cost_host_) (void)hipHostFree(cost_host_);
    for (int p = 0; p < 2; ++p) if (host_ids_pin_[p]) (void)hipHostFree(host_ids_pin_[p]);
    if (err_host_) (void)hipHostFree(err_host_);
    // (copy_stream_ is side stream 3)
    for (DeferredCost& d : deferred_) { if (d.ev) (void)hipEventDestroy(d.ev); if (d.host) (void)hipHostFree(d.host); }
    if (comm_ && rccl_) rccl_->CommDestroy(comm_);
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

void Model::set_stream(hipStream_t s) {
    synchronize();
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
    if (s) { stream_ = s; own_stream_ = false; }
    else {
        // back to a stream of the handle's own, at the highest priority like the one the constructor made
        int lo = 0, hi = 0;
        NVSM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        NVSM_HIP_CHECK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));
        own_stream_ = true;
    }
}

void Model::synchronize() {
    NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
    NVSM_HIP_CHECK(hipStreamSynchronize(aux_stream_));
    NVSM_HIP_CHECK(hipStreamSynchronize(aux2_stream_));
    NVSM_HIP_CHECK(hipStreamSynchronize(aux3_stream_));      // (= copy_stream_)
    E_pending_ = T_pending_ = false;
    raise_device_error();
}

// Kernels report bad input ids (and, with NVSM_DEBUG=1, non-finite intermediates) by storing a code into a page-locked
// host word; every host-side wait of the engine looks at it afterwards. The offending ids were replaced by row 0 on the
// device, so nothing was written out of bounds; the step's numbers are meaningless and the caller is told so.
void Model::raise_device_error() {
    const int code = *static_cast<volatile int*>(err_host_);
    if (code == 0) return;
    *err_host_ = 0;
    // The arrival counters of the ordered sums and of the one-launch table passes return to zero at the end of every launch
    // that runs to completion; a launch that flagged an error may have left some behind, and every later sum would silently
    // add the wrong partials: clear them before the caller goes on. The callers have waited for much less than the whole
    // step (step(cost): the copy of the loss word; get_cost(): the main stream; deferred_cost(): one event) — the table passes
    // and ordered sums of the same step may still be running on the side streams, and a counter zeroed under a running
    // kernel loses its last arriver: every stream of the handle is quiet BEFORE the counters are touched.
    for (hipStream_t s : {stream_, aux_stream_, aux2_stream_, aux3_stream_})
        if (s) (void)hipStreamSynchronize(s);
    E_pending_ = T_pending_ = false;
    words_tail_pending_ = false;
    // (a failed step may have queued the hoisted decay of the words rows without entries with no update behind it — the handle is
    //  then PARTIALLY updated, which the caller is told by the exception; what must not survive is the bookkeeping that would make
    //  the next stand-alone update() skip that decay or refuse to run)
    words_untouched_hoisted_ = false; words_untouched_pending_ = false; words_snapshot_early_ = false;
    settle_words_stamp();
    for (DevBuf<int>* b : {&sums_fwd_.arrive, &sums_bwd_.arrive, &words_.arrive_row, &words_.arrive2, &ents_.arrive_row, &ents_.arrive2})
        if (b->p) (void)hipMemset(b->p, 0, b->n * sizeof(int));
    (void)hipDeviceSynchronize();      // (the fills are queued on the null stream, which the handle's streams do not follow)
    if (code == NVSM_BAD_WORD_ID) throw Error(NVSM_ERR_INVALID_ARGUMENT, "a word id of the batch is outside [0, num_words)");
    if (code == NVSM_BAD_ENTITY_ID) throw Error(NVSM_ERR_INVALID_ARGUMENT, "a document id of the batch is outside [0, num_entities)");
    if (code == NVSM_SORT_TIMEOUT) throw Error(NVSM_ERR_DEVICE, "the row sort's grid-wide wait timed out (workgroups not co-resident?)");
    static const char* const names[] = {"phrase", "pre (projection)", "proj", "probs", "grad_proj", "grad_phrase", "grad_transform",
                                        "word_representations", "entity_representations", "transform"};
    const int which = code - NVSM_NONFINITE_BASE;
    throw Error(NVSM_ERR_DEVICE, std::string("NVSM_DEBUG: non-finite values in ") +
                                 ((which >= 0 && which < 10) ? names[which] : "an intermediate"));
}

// NVSM_DEBUG=1 — CHECK_MATRIX of the reference's debug build (cpp/objective.cu:134,141,152 …)
void Model::debug_check(const float* x, int64_t n, int which) {
    if (debug_) launch_check_finite(x, n, err_host_, NVSM_NONFINITE_BASE + which, stream_);
}

// nvsm_step leaves the side streams' tails (documents update; dT GEMM + projection update) running when it returns, so
// that they overlap the start of the next step; whoever next touches what a tail reads or writes joins it first.
void Model::join_T() {
    if (!T_pending_) return;
    NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_T_done_, 0));
    T_pending_ = false;
}
void Model::join_E() {
    if (!E_pending_) return;
    NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_E_done_, 0));
    E_pending_ = false;
}

void Model::debug_delay(int microseconds) {
    NVSM_HIP_CHECK(hipSetDevice(cfg_.device));
    launch_delay(microseconds, stream_);
}

// ModelBase::initialize (cpp/model.cu:37-43) with init_matrix_glorot (include/cuNVSM/cuda_utils.h:35-56):
// same generator, same draw order (words → entities → transform), same float expression.
void Model::initialize(uint64_t seed) {
    if (seed == 0) throw Error(NVSM_ERR_INVALID_ARGUMENT, "Please specify a seed value > 0");   // cpp/main.cu:708
    rng_.seed(static_cast<std::minstd_rand0::result_type>(seed));
    device_seed_ = seed;
    initialize_from_rng_state();
}

void Model::initialize_from_rng_state() {
    NVSM_HIP_CHECK(hipSetDevice(cfg_.device));
    synchronize();
    lazy_flush_all();          // (fresh parameters carry no pending decay: every row is current as of updates_done)
    if (device_seed_ == 0) device_seed_ = 1;
    auto glorot = [&](DevBuf<float>& dst, size_t rows, size_t cols) {
        std::vector<float> h(rows * cols);
        const float max = std::sqrt(6.0 / static_cast<double>(rows + cols));
        for (size_t i = 0; i < h.size(); ++i) h[i] = 2 * max * (std::generate_canonical<float, 1>(rng_) - 0.5);
        NVSM_HIP_CHECK(hipMemcpy(dst.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    };
    glorot(words_.P, cfg_.word_repr_size, cfg_.num_words);
    glorot(ents_.P, cfg_.entity_repr_size, cfg_.num_entities);
    glorot(T_, cfg_.entity_repr_size, cfg_.word_repr_size);
    planes_stale();      // (T rewritten: its bf16 planes are stale)
    NVSM_HIP_CHECK(hipMemset(b_.p, 0, b_.n * sizeof(float)));                                   // params.cu:368-369
    NVSM_HIP_CHECK(hipStreamSynchronize(nullptr));      // (queued on the null stream, which the handle's streams do not wait for)
}

uint64_t Model::rng_get_state() { std::stringstream ss; ss << rng_; uint64_t s; ss >> s; return s; }
void Model::rng_set_state(uint64_t s) { std::stringstream ss; ss << s; ss >> rng_; }

// ---------------------------------------------------------------------------------------------
// collectives
// ---------------------------------------------------------------------------------------------
void Model::comm_init(const char id[128]) {
    if (cfg_.world_size <= 1) return;
    rccl_ = RcclApi::load();
    RcclApi::UniqueId u;
    std::memcpy(u.internal, id, 128);
    NVSM_HIP_CHECK(hipSetDevice(cfg_.device));
    const int rc = rccl_->CommInitRank(&comm_, cfg_.world_size, u, cfg_.rank);
    if (rc != 0) throw Error(NVSM_ERR_DEVICE, std::string("ncclCommInitRank: ") + (rccl_->GetErrorString ? rccl_->GetErrorString(rc) : "error"));
    if (exact_ && !rccl_->AllGather) throw Error(NVSM_ERR_DEVICE, "librccl lacks ncclAllGather (dp_exact_tables)");
    comm_ranks_ = cfg_.world_size;
    if (rccl_->CommCount) { int n = 0; if (rccl_->CommCount(comm_, &n) == 0) comm_ranks_ = n; }
    // The step issues its collectives on ONE communicator from two streams ordered by events (never two in flight). Before
    // the first step relies on that, the same pattern runs once with known values; a communicator that does not deliver the
    // right sums that way gets every collective on the main stream instead (and must pass the check there).
    if (!dp_single_stream_ && !comm_order_check(true)) {
        std::fprintf(stderr, "cunvsm_amd: rank %d: collectives on two event-ordered streams failed their check; using the main stream for all "
                             "of them (NVSM_DP_T_ON_MAIN=1)\n", cfg_.rank);
        dp_single_stream_ = true;
    }
    if (dp_single_stream_ && !comm_order_check(false))
        throw Error(NVSM_ERR_DEVICE, "the RCCL communicator does not all-reduce correctly (checked with known values)");
}

// The step's collective pattern with known values: f64 on the main stream (batch-norm sums), f64 again (backward sums), f32 on
// side stream 2 behind an event (projection gradient), f64 on the main stream behind an event (the next step's sums). Rank r
// contributes r + 1, so every element must come back as G (G + 1) / 2 times its index weight. The verdict is itself summed
// over the ranks (on the main stream, which both modes use), so that all ranks choose the same mode.
bool Model::comm_order_check(bool two_streams) {
    const int n = 2 * cfg_.entity_repr_size + 1, nf = 4096, G = cfg_.world_size;
    DevBuf<double> d; DevBuf<float> f; DevBuf<double> verdict;
    d.alloc(n); f.alloc(nf); verdict.alloc(1);
    std::vector<double> hd(n); std::vector<float> hf(nf);
    bool ok = true;
    hipStream_t side = two_streams ? aux2_stream_ : stream_;
    hipEvent_t e1 = nullptr, e2 = nullptr;
    NVSM_HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    NVSM_HIP_CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    for (int round = 0; round < 3 && ok; ++round) {
        for (int i = 0; i < n; ++i) hd[i] = (cfg_.rank + 1) * (1.0 + i);
        for (int i = 0; i < nf; ++i) hf[i] = static_cast<float>((cfg_.rank + 1) * (1 + i % 7));
        NVSM_HIP_CHECK(hipMemcpyAsync(d.p, hd.data(), n * sizeof(double), hipMemcpyHostToDevice, stream_));
        NVSM_HIP_CHECK(hipMemcpyAsync(f.p, hf.data(), nf * sizeof(float), hipMemcpyHostToDevice, stream_));
        int rc = rccl_->AllReduce(d.p, d.p, n, RcclApi::kFloat64, RcclApi::kSum, comm_, stream_);
        NVSM_HIP_CHECK(hipEventRecord(e1, stream_));
        NVSM_HIP_CHECK(hipStreamWaitEvent(side, e1, 0));
        if (rc == 0) rc = rccl_->AllReduce(f.p, f.p, nf, RcclApi::kFloat32, RcclApi::kSum, comm_, side);
        NVSM_HIP_CHECK(hipEventRecord(e2, side));
        NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, e2, 0));
        if (rc == 0) rc = rccl_->AllReduce(d.p, d.p, n, RcclApi::kFloat64, RcclApi::kSum, comm_, stream_);
        NVSM_HIP_CHECK(hipMemcpyAsync(hd.data(), d.p, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
        NVSM_HIP_CHECK(hipMemcpyAsync(hf.data(), f.p, nf * sizeof(float), hipMemcpyDeviceToHost, stream_));
        NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
        NVSM_HIP_CHECK(hipStreamSynchronize(side));
        const double s1 = 0.5 * G * (G + 1);
        ok = rc == 0;
        for (int i = 0; i < n && ok; ++i) ok = hd[i] == s1 * G * (1.0 + i);             // summed twice: the second time G equal copies
        for (int i = 0; i < nf && ok; ++i) ok = hf[i] == static_cast<float>(s1 * (1 + i % 7));
    }
    (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    // every rank must reach the same verdict: sum the failures (on the main stream, with the plainest possible call)
    double bad = ok ? 0.0 : 1.0;
    NVSM_HIP_CHECK(hipMemcpyAsync(verdict.p, &bad, sizeof(double), hipMemcpyHostToDevice, stream_));
    const int rc = rccl_->AllReduce(verdict.p, verdict.p, 1, RcclApi::kFloat64, RcclApi::kSum, comm_, stream_);
    NVSM_HIP_CHECK(hipMemcpyAsync(&bad, verdict.p, sizeof(double), hipMemcpyDeviceToHost, stream_));
    NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
    return rc == 0 && bad == 0.0;
}

// Data parallel: the embedding tables are updated rank-locally (north_star "sparse embedding rows stay
// GPU-local"), so the replicas drift apart. This replaces every replica's W and E by the mean over ranks (parameter
// averaging; the optimiser state stays rank-local) — a collective every rank must call; the trainer calls it before each
// model dump and at the end of every epoch so that what rank 0 writes carries all ranks' updates.
void Model::average_tables() {
    if (cfg_.world_size <= 1 || exact_) return;      // (exact tables: the replicas are bit-identical already)
    NVSM_HIP_CHECK(hipSetDevice(cfg_.device));
    synchronize();
    lazy_flush_all();
    const float inv = 1.0f / static_cast<float>(cfg_.world_size);
    for (TableState* t : {&words_, &ents_}) {
        const int64_t n = static_cast<int64_t>(t->P.n);
        // chunks keep the callback transport's host staging bounded; RCCL takes them as they come
        const int64_t chunk = int64_t(1) << 24;
        for (int64_t off = 0; off < n; off += chunk)
            allreduce_f32(t->P.p + off, std::min(chunk, n - off), stream_);
        launch_scale(t->P.p, n, inv, stream_);
    }
    NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Model::allreduce_f64(double* dev, int64_t n) {
    if (cfg_.world_size <= 1) return;
    if (comm_ && !ar_fn_) {            // an installed callback takes precedence (every rank must use the same transport)
        const int rc = rccl_->AllReduce(dev, dev, n, RcclApi::kFloat64, RcclApi::kSum, comm_, stream_);
        if (rc != 0) throw Error(NVSM_ERR_DEVICE, "ncclAllReduce(f64) failed");
    } else if (ar_fn_) {
        ar_host_.resize(n);
        NVSM_HIP_CHECK(hipMemcpyAsync(ar_host_.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
        NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
        if (ar_fn_(ar_host_.data(), n, ar_user_) != 0) throw Error(NVSM_ERR_DEVICE, "all-reduce callback failed");
        NVSM_HIP_CHECK(hipMemcpyAsync(dev, ar_host_.data(), n * sizeof(double), hipMemcpyHostToDevice, stream_));
        NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
    } else {
        throw Error(NVSM_ERR_STATE, "world_size > 1 but neither nvsm_comm_init nor an all-reduce callback was set");
    }
}

void Model::allreduce_f32(float* dev, int64_t n, hipStream_t strm) {
    if (cfg_.world_size <= 1) return;
    if (comm_ && !ar_fn_) {
        const int rc = rccl_->AllReduce(dev, dev, n, RcclApi::kFloat32, RcclApi::kSum, comm_, strm);
        if (rc != 0) throw Error(NVSM_ERR_DEVICE, "ncclAllReduce(f32) failed");
    } else if (ar_fn_) {
        std::vector<float> h(n);
        NVSM_HIP_CHECK(hipMemcpyAsync(h.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost, strm));
        NVSM_HIP_CHECK(hipStreamSynchronize(strm));
        ar_host_.assign(h.begin(), h.end());
        if (ar_fn_(ar_host_.data(), n, ar_user_) != 0) throw Error(NVSM_ERR_DEVICE, "all-reduce callback failed");
        for (int64_t i = 0; i < n; ++i) h[i] = static_cast<float>(ar_host_[i]);
        NVSM_HIP_CHECK(hipMemcpyAsync(dev, h.data(), n * sizeof(float), hipMemcpyHostToDevice, strm));
        NVSM_HIP_CHECK(hipStreamSynchronize(strm));
    } else {
        throw Error(NVSM_ERR_STATE, "world_size > 1 but neither nvsm_comm_init nor an all-reduce callback was set");
    }
}

// `bytes` from every rank, rank-major (dp_exact_tables). RCCL: one ncclAllGather of bytes. Callback transport (tests): the
// callback can only sum doubles, so every rank lays its 32-bit words into its own slot of a zeroed buffer — a sum with one
// non-zero term per element, exact in a double.
void Model::allgather(const void* send, void* recv, size_t bytes, hipStream_t strm) {
    const size_t G = static_cast<size_t>(cfg_.world_size);
    if (comm_ && !ar_fn_) {
        const int rc = rccl_->AllGather(send, recv, bytes, RcclApi::kInt8, comm_, strm);
        if (rc != 0) throw Error(NVSM_ERR_DEVICE, "ncclAllGather failed");
    } else if (ar_fn_) {
        if (bytes % 4 != 0) throw Error(NVSM_ERR_INVALID_ARGUMENT, "all-gather of a size that is not a multiple of 4 bytes");
        const size_t n = bytes / 4;
        std::vector<uint32_t> h(n), all(G * n);
        NVSM_HIP_CHECK(hipMemcpyAsync(h.data(), send, bytes, hipMemcpyDeviceToHost, strm));
        NVSM_HIP_CHECK(hipStreamSynchronize(strm));
        ar_host_.assign(G * n, 0.0);
        for (size_t i = 0; i < n; ++i) ar_host_[static_cast<size_t>(cfg_.rank) * n + i] = static_cast<double>(h[i]);
        if (ar_fn_(ar_host_.data(), static_cast<int64_t>(G * n), ar_user_) != 0) throw Error(NVSM_ERR_DEVICE, "all-reduce callback failed");
        for (size_t i = 0; i < G * n; ++i) all[i] = static_cast<uint32_t>(ar_host_[i]);
        NVSM_HIP_CHECK(hipMemcpyAsync(recv, all.data(), G * bytes, hipMemcpyHostToDevice, strm));
        NVSM_HIP_CHECK(hipStreamSynchronize(strm));
    } else {
        throw Error(NVSM_ERR_STATE, "world_size > 1 but neither nvsm_comm_init nor an all-reduce callback was set");
    }
}

// exact tables: what the loss kernel and the backward product left for the table passes, from every rank (main stream, in
// front of the passes). The ids were gathered by compute_cost, for the CSR builds.
void Model::gather_update_inputs() {
    if (!exact_) return;
    PROF("allgather_update_inputs");
    const int64_t B = B_;
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size;
    allgather(proj_.p, xg_proj_.p, B * de * sizeof(float), stream_);
    allgather(coef_.p, xg_coef_.p, B * R_ * sizeof(float), stream_);
    allgather(pp_.p, xg_pp_.p, B * sizeof(float), stream_);
    allgather(gphrase_.p, xg_gphrase_.p, B * dw * sizeof(float), stream_);
    allgather(msq_w_.p, xg_msq_w_.p, B * sizeof(float), stream_);
}

Model::UpdateInputs Model::update_inputs() const {
    if (!exact_) return UpdateInputs{proj_.p, coef_.p, pp_.p, ids_p_, gphrase_.p, wwts_, msq_w_.p, widx_.p, B_};
    return UpdateInputs{xg_proj_.p, xg_coef_.p, xg_pp_.p, xg_ids_p_, xg_gphrase_.p, wwts_ ? xg_wwts_.p : nullptr, xg_msq_w_.p,
                        xg_widx_.p, B_ * cfg_.world_size};
}

// UniformLabelGenerator::generate (cpp/labels.cu:4-22) → generate_random_indexes (include/cuNVSM/cuda_utils.h:24-33): slot 0 of
// every example is its label, slots 1..k are `std::uniform_int_distribution<long>(0, |D|-1)(rng)`, a fresh distribution per
// draw, on the caller's std::minstd_rand0. This is that loop draw for draw — the same generator steps, the same rejections,
// the same quotients — written out so that it costs ≈2 ns instead of ≈8 ns per draw (819 200 draws per step at the NVSM
// shape): libstdc++'s downscaling branch (bits/uniform_int_dist.h; the generator's range 2^31-3 is not a power of two, so
// no other branch applies) is   scaling = urng_range / n;  past = n · scaling;  do r = g() − g.min(); while (r ≥ past);
// return r / scaling;   and minstd_rand0 is x ← 16807·x mod (2^31 − 1). tests/test_gpu_parity.py compares it with the std:: calls.
void Model::draw_reference_negatives(const int64_t* labels, int64_t B, int64_t* ids) {
    // Data parallel: the ranks hold copies of ONE generator state (the trainer hands every rank the same stream), and rank r
    // trains on instances [r·B, (r+1)·B) of the global batch. Every rank therefore replays the draws of the WHOLE global batch
    // — world_size · B · k of them, in instance order — and keeps the ones of its own slice: each instance gets the negatives
    // the single-GPU run would have given it, no two ranks share a negative set, and the generators stay in step across the
    // ranks (they also shuffle the data source). Costs world_size x the draws per rank: this is the parity sampler.
    const int64_t before = (cfg_.world_size > 1) ? static_cast<int64_t>(cfg_.rank) * B : 0;
    const int64_t after = (cfg_.world_size > 1) ? static_cast<int64_t>(cfg_.world_size - 1 - cfg_.rank) * B : 0;
    const int k = R_ - 1;
    const uint64_t n = static_cast<uint64_t>(cfg_.num_entities);
    constexpr uint64_t kMod = 2147483647ull, kRange = kMod - 2;          // max() − min() = 2147483646 − 1
    if (n > kRange) {                                                     // not the downscaling branch: leave it to libstdc++
        auto draw = [&] { return std::uniform_int_distribution<long>(0, cfg_.num_entities - 1)(rng_); };
        for (int64_t i = 0; i < before * k; ++i) (void)draw();
        for (int64_t i = 0; i < B; ++i) {
            ids[i * R_] = labels[i];
            for (int r = 1; r < R_; ++r) ids[i * R_ + r] = draw();
        }
        for (int64_t i = 0; i < after * k; ++i) (void)draw();
        return;
    }
    std::stringstream ss; ss << rng_;
    uint64_t x = 0; ss >> x;
    const uint64_t scaling = kRange / n, past = n * scaling;
    const double inv = 1.0 / static_cast<double>(scaling);
    auto draw = [&]() -> uint64_t {
        uint64_t ret;
        do {
            const uint64_t p = x * 16807ull;                          // < 2^46
            x = (p & kMod) + (p >> 31);                               // mod 2^31 − 1
            if (x >= kMod) x -= kMod;
            ret = x - 1;
        } while (ret >= past);
        return ret;
    };
    for (int64_t i = 0; i < before * k; ++i) (void)draw();              // (the quotient is not needed to advance the state)
    for (int64_t i = 0; i < B; ++i) {
        ids[i * R_] = labels[i];
        for (int r = 1; r < R_; ++r) {
            const uint64_t ret = draw();
            uint64_t q = static_cast<uint64_t>(static_cast<double>(ret) * inv);      // ret / scaling, fixed up to be exact
            if ((q + 1) * scaling <= ret) ++q; else if (q * scaling > ret) --q;
            ids[i * R_ + r] = static_cast<int64_t>(q);
        }
    }
    for (int64_t i = 0; i < after * k; ++i) (void)draw();
    rng_.seed(static_cast<std::minstd_rand0::result_type>(x));          // 1 ≤ x < 2^31 − 1: seed() stores it unchanged
}

// ---------------------------------------------------------------------------------------------
// compute_cost — cpp/objective.cu:30-313
// ---------------------------------------------------------------------------------------------
void Model::compute_cost(const nvsm_batch& batch, const int64_t* entity_ids) {
    const int64_t B = batch.num_instances;
    if (B <= 0 || B > cfg_.max_batch_size) throw Error(NVSM_ERR_INVALID_ARGUMENT, "num_instances must be in (0, max_batch_size]");
    if (!batch.features || !batch.labels) throw Error(NVSM_ERR_INVALID_ARGUMENT, "features and labels are required");
    NVSM_HIP_CHECK(hipSetDevice(cfg_.device));
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size, w = cfg_.window_size, k = cfg_.num_random_entities;
    const int64_t N = B * R_;
    B_ = B;
    have_forward_ = have_grads_ = false;
    cost_valid_ = false;
    loss_reduced_ = false;
    loss_folded_ = false;
    RangeScope range_cc("ComputeCost");                 // cpp/main.cu:409
    if (cfg_.l2_normalize_entity_reprs) join_E();      // that documents update still reads ids_, which the prologue rewrites
    // The previous step's CSR builds (radix sorts on the side streams) read ids_ / widx_, which this step's prologue is
    // about to rewrite: the main stream must be behind them. In steady state both are long finished (update() / step()
    // joined them before the row passes), so the waits cost nothing; they matter for compute_cost; compute_cost without
    // an update in between and for the documents build, which step() never joins on the main stream.
    // (A wait is a packet the main stream stops at for several microseconds even when the event has long fired: the fused
    //  step joins both builds in one place, in front of the words update, and nothing is waited for here then.)
    if (inputs_recorded_) {
        if (!csr_joined_ents_) NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_csr_ents_, 0));
        if (!csr_joined_words_) NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_csr_, 0));
        csr_joined_ents_ = csr_joined_words_ = true;
    }

    ids_p_ = (ids_p_ == ids_buf_[0].p) ? ids_buf_[1].p : ids_buf_[0].p;
    // device-sampler mode: zeroing the statistics, narrowing the word ids and drawing the document ids are one launch
    const bool fused_prologue = !entity_ids && cfg_.sampler != NVSM_SAMPLER_HOST_MINSTD;
    if (!fused_prologue) {
        settle_words_stamp();      // (the fused prologue sets the previous words update's stamps itself)
        NVSM_HIP_CHECK(hipMemsetAsync(stats_.p, 0, stats_.n * sizeof(double), stream_));   // Σx Σx² | loss Σdy Σdy·x̂
    }

    // F1: batch → HBM (objective.cu:36-61)
    const int64_t* words_dev;
    {
        PROF("h2d_batch");
        if (batch.on_device) {
            words_dev = batch.features;
            labels_dev_ = batch.labels;
            wwts_ = batch.feature_weights;
            instw_ = batch.weights;
        } else {
            // Host batch: the four arrays go to the device on a copy stream of their own, into the staging set the step
            // before last used — so the copies of step k run while step k-1 is still computing (the caller, running
            // ahead, has usually queued them by then) instead of sitting in front of step k on the main stream.
            // ev_step_begin_[p] was recorded on the main stream when the previous step's compute_cost started, i.e. behind
            // everything the step before that one — the last reader of this staging set — had queued there.
            const int p = in_parity_ ^= 1;
            // When the copy stream is also the stream of the documents CSR build (the default layout), that order is there
            // already: this copy is queued behind the previous step's documents sort, which waited for that step's prologue,
            // i.e. for everything the step before it had on the main stream — no event, and no record packet in front of
            // the prologue on the critical stream.
            const bool csr_at_start = tune_.csr_after == 0;
            // (the layout follows the batch size: the argument needs the previous step's documents sort on the copy stream)
            const bool copies_behind_sort = csr_at_start && csr_stream_layout() == 4 && last_csr_layout_ == 4;
            if (!(copies_behind_sort && copy_stream_ == aux3_stream_ && aux3_stream_)) {
                NVSM_HIP_CHECK(hipStreamWaitEvent(copy_stream_, ev_step_begin_[p ^ 1], 0));
                NVSM_HIP_CHECK(hipEventRecord(ev_step_begin_[p], stream_));
            }
            // Page-locked arrays (nvsm_host_alloc / hipHostMalloc: the trainer's and the reference's batches) are pulled by ONE
            // kernel that reads them over PCIe; anything else (pageable memory) goes through hipMemcpyAsync, which stages it.
            // The runtime's copy call occupied this thread for most of a step even for page-locked sources (kernels.h HostPull).
            HostPull pull{};
            auto bring = [&](void* dst, const void* src, size_t bytes) {
                const bool use_pull = tune_.host_pull;
                void* dev_view = nullptr;
                hipPointerAttribute_t at{};
                // (the pull kernel reads 16 bytes per lane: a source that is only element-aligned — a slice of a page-locked
                //  batch at an odd instance offset — takes the copy engine instead)
                if (use_pull && bytes % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 &&
                    hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost &&
                    hipHostGetDevicePointer(&dev_view, const_cast<void*>(src), 0) == hipSuccess && dev_view) {
                    pull.dst[pull.count] = dst; pull.src[pull.count] = dev_view; pull.bytes[pull.count] = bytes; ++pull.count;
                } else {
                    (void)hipGetLastError();      // (a failed attribute query of a pageable pointer is not an error of ours)
                    NVSM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, copy_stream_));
                }
            };
            bring(in_words_[p].p, batch.features, B * w * sizeof(int64_t));
            bring(in_labels_[p].p, batch.labels, B * sizeof(int64_t));
            words_dev = in_words_[p].p;
            labels_dev_ = in_labels_[p].p;
            wwts_ = nullptr; instw_ = nullptr;
            if (batch.feature_weights) { bring(in_wwts_[p].p, batch.feature_weights, B * w * sizeof(float)); wwts_ = in_wwts_[p].p; }
            if (batch.weights) { bring(in_instw_[p].p, batch.weights, B * sizeof(float)); instw_ = in_instw_[p].p; }
            launch_host_pull(pull, copy_stream_);
            NVSM_HIP_CHECK(hipEventRecord(ev_copied_, copy_stream_));
            NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_copied_, 0));
            copied_recorded_ = true;
        }
        last_batch_on_host_ = !batch.on_device;
        if (!fused_prologue) launch_narrow_i64(words_dev, widx_.p, B * w, cfg_.num_words, err_host_, NVSM_BAD_WORD_ID, stream_);
    }

    // F2: target + negative document ids (objective.cu:63-89 → labels.cu:4-22)
    {
        PROF("sample_entities");
        if (entity_ids) {
            NVSM_HIP_CHECK(hipMemcpyAsync(in_ids64_.p, entity_ids, N * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
            launch_narrow_i64(in_ids64_.p, ids_p_, N, cfg_.num_entities, err_host_, NVSM_BAD_ENTITY_ID, stream_);
        } else if (cfg_.sampler == NVSM_SAMPLER_HOST_MINSTD) {
            host_labels_.resize(B);
            if (batch.on_device) {
                // device-resident labels: read back on the copy stream — the host waits for these B words only, not for
                // whatever the main stream still has queued
                NVSM_HIP_CHECK(hipMemcpyAsync(host_labels_.data(), batch.labels, B * sizeof(int64_t), hipMemcpyDeviceToHost, copy_stream_));
                NVSM_HIP_CHECK(hipStreamSynchronize(copy_stream_));
            } else {
                std::memcpy(host_labels_.data(), batch.labels, B * sizeof(int64_t));
            }
            // page-locked, two of them: this one was last read by the copy of the step before last (its event has long
            // fired), so the host never waits for the stream here and a deferred-loss loop stays one step ahead of the GPU
            const int hp = host_ids_parity_ ^= 1;
            if (host_ids_used_[hp]) NVSM_HIP_CHECK(hipEventSynchronize(ev_host_ids_[hp]));
            draw_reference_negatives(host_labels_.data(), B, host_ids_pin_[hp]);
            NVSM_HIP_CHECK(hipMemcpyAsync(in_ids64_.p, host_ids_pin_[hp], N * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
            NVSM_HIP_CHECK(hipEventRecord(ev_host_ids_[hp], stream_));
            host_ids_used_[hp] = true;
            launch_narrow_i64(in_ids64_.p, ids_p_, N, cfg_.num_entities, err_host_, NVSM_BAD_ENTITY_ID, stream_);
        } else {
            // (the prologue carries "inputs consumed, ids final" as its completion event: no packet between it and the gather)
            StampJob sj{};
            if (words_stamp_pending_) {
                words_stamp_pending_ = false;
                const Csr cw = csr_of(words_, words_stamp_n_);
                sj.list = cw.touched; sj.count = cw.num_touched; sj.stamp = words_.stamp.p; sj.value = words_.updates_done;
            }
            launch_and_record(ev_inputs_, stream_, [&] {
                launch_step_prologue(words_dev, widx_.p, B * w, labels_dev_, B, R_, cfg_.num_words, cfg_.num_entities,
                                     device_seed_ + 0x9E37u * cfg_.rank, step_count_, ids_p_, stats_.p, static_cast<int>(stats_.n),
                                     err_host_, stream_, sj);
            });
        }
    }
    ++step_count_;
    if (exact_) {
        // exact tables: the CSRs are those of the global batch — every rank's ids, rank-major (the gathered document ids
        // alternate between two buffers for the reason the rank's own do)
        PROF("allgather_ids");
        xg_ids_p_ = (xg_ids_p_ == xg_ids_[0].p) ? xg_ids_[1].p : xg_ids_[0].p;
        allgather(ids_p_, xg_ids_p_, N * sizeof(int), stream_);
        allgather(widx_.p, xg_widx_.p, B * w * sizeof(int), stream_);
        if (wwts_) allgather(wwts_, xg_wwts_.p, B * w * sizeof(float), stream_);
    }
    const int64_t Bu = exact_ ? B * cfg_.world_size : B;          // windows of the table updates
    const int* csr_ids = exact_ ? xg_ids_p_ : ids_p_;
    const int* csr_widx = exact_ ? xg_widx_.p : widx_.p;

    // Row-order (CSR) of both tables for the update, on the side streams: needs only the indices.
    const int csr_after = tune_.csr_after;
    if (!fused_prologue || exact_) NVSM_HIP_CHECK(hipEventRecord(ev_inputs_, stream_));
    inputs_recorded_ = true;
    // two side streams: the sorts are latency-bound chains of small launches, so the two tables' builds run next to
    // each other (at batch 4096 one behind the other they were the longest chain of the whole step)
    // Which side stream builds which table's CSR. Side stream 1 still carries the PREVIOUS step's documents update when this
    // step begins (it runs ~150 us into it), so a sort queued there starts late and lands on the loss kernel; side stream 2
    // (dT GEMM + projection update of the previous step) is free by then. NVSM_SORT_LAYOUT: 0 = documents on side stream 1,
    // words on 2 (default); 1 = both on 2, words first; 2 = both on 2, documents first; 3 = documents on 2, words on 1.
    // The documents CSR arrays are still being read by the previous step's documents update, so a build on another stream
    // would have to wait for it (ev_E_done_) all the same; hence the second set of CSR arrays (TableState::CsrIndex).
    // 4 = documents on side stream 3, words on 2: neither queues behind the previous step's tails (default).
    // Where the dT product runs on the main stream (dt_on_main(): large batches of eager tables) side stream 2 is idle for the
    // whole step, and both builds go there one behind the other, documents first (2): the loss kernel then runs next to
    // one sort at a time instead of two, and the words CSR is still early (NVSM shape 0.933 -> 0.911 ms, loss kernel
    // 185 -> 176 us in-step; everywhere else 2 is 4-25 % slower than 4: interleaved A/B).
    const int sort_layout = csr_stream_layout();
    last_csr_layout_ = sort_layout;
    // which: 1 = the documents table, 2 = the words table, 3 = both
    auto launch_csr_builds = [&](hipEvent_t after, int which = 3) {
        const int layout = aux3_stream_ ? sort_layout : (sort_layout == 4 ? 0 : sort_layout);
        hipStream_t se = (layout == 0) ? aux_stream_ : (layout == 4 ? aux3_stream_ : aux2_stream_);
        hipStream_t sw = (layout == 3) ? aux_stream_ : aux2_stream_;
        if (which & 1) NVSM_HIP_CHECK(hipStreamWaitEvent(se, after, 0));
        if (which & 2) NVSM_HIP_CHECK(hipStreamWaitEvent(sw, after, 0));
        // (the documents table has two sets of CSR arrays: its build does not wait for the previous documents update)
        if ((which & 1) && se != aux_stream_ && E_pending_ && ents_.idx_sets < 2) NVSM_HIP_CHECK(hipStreamWaitEvent(se, ev_E_done_, 0));
        if (which & 1) csr_joined_ents_ = false;
        if (which & 2) { csr_joined_words_ = false; words_csr_stream_ = sw; }
        auto ents = [&] { { PROF_ON("csr_entities", se); build_csr(ents_, csr_ids, Bu * R_, se); } NVSM_HIP_CHECK(hipEventRecord(ev_csr_ents_, se)); };
        // (lazily decayed words table with a per-row scalar: the scalars of the rows this batch touches are brought up to date
        //  into the snapshot the moments pass reads right here, behind the build that lists those rows — it needs nothing the
        //  step computes, and in front of the words update it was a launch of 7 us on the critical stream)
        auto wrds = [&] {
            { PROF_ON("csr_words", sw); build_csr(words_, csr_widx, Bu * w, sw); }
            if (words_.lazy && words_.lazy_scalar && tune_.early_snapshot) { lazy_scalar_snapshot(words_, csr_of(words_, Bu * w), sw); words_snapshot_early_ = true; }
            // The fused step knows lr and λ already: the decay of the words rows WITHOUT entries (SGD / Adagrad, λ > 0, a table
            // much larger than the batch: launch_untouched_rows) goes here, under the forward pass, instead of into the update's
            // tail where the next step's word gather waited for it (LSE batch 4096: 16-22 us per step). Nothing of this step
            // reads those rows; the NEXT word gather may. NVSM_HOIST_UNTOUCHED=2 (the default, tuning.h): behind the event the words
            // update waits for, with an event of its own that the next word gather follows (the build is not lengthened by the
            // pass's 11 us); 1: in front of that event — the main stream is then behind the pass without a wait of its own (a wait is
            // a packet the stream stops at for 6-10 us even when the event has long fired). Mode 2 leans on `sw` being the SAME
            // stream in consecutive steps (the next build's sort on it clears the row bounds this pass reads): the layout is a
            // function of the batch size and the handle's switches only — checked below.
            words_untouched_hoisted_ = false;
            const bool hoist = hoist_untouched_ && !words_.lazy && (cfg_.update_method == NVSM_SGD || cfg_.update_method == NVSM_ADAGRAD);
            auto hoisted_pass = [&] {
                RowPassArgs ua = final_words_pass_args(hoist_lr_, hoist_sl_);
                if (!launch_untouched_rows(csr_of(words_, Bu * w), ua, sw)) return false;
                words_untouched_hoisted_ = true;
                prof.note("untouched_words_hoisted");
                return true;
            };
            if (hoist && tune_.hoist_untouched == 1) (void)hoisted_pass();
            NVSM_HIP_CHECK(hipEventRecord(ev_csr_, sw));
            if (hoist && tune_.hoist_untouched == 2 && words_untouched_stream_prev_ && words_untouched_stream_prev_ != sw)
                throw Error(NVSM_ERR_STATE, "the words CSR stream changed between steps while a hoisted decay leans on it");
            if (hoist && tune_.hoist_untouched == 2) words_untouched_stream_prev_ = sw;
            if (hoist && tune_.hoist_untouched == 2 && hoisted_pass()) {
                NVSM_HIP_CHECK(hipEventRecord(ev_untouched_, sw));
                words_untouched_pending_ = true;
            }
        };
        if (layout == 1) { if (which & 2) wrds(); if (which & 1) ents(); } else { if (which & 1) ents(); if (which & 2) wrds(); }
        if ((which & 1) && se != aux_stream_) NVSM_HIP_CHECK(hipStreamWaitEvent(aux_stream_, ev_csr_ents_, 0));     // the documents update follows its CSR
    };
    const bool any_lazy = words_.lazy || ents_.lazy;
    // (a lazily decayed documents table read by the generic loss kernel is refreshed by list first: that needs the CSR now)
    const bool csr_first = csr_after == 0 || (ents_.lazy && !loss_reads_lazily(de, static_cast<int>(R_), cfg_.l2_normalize_entity_reprs != 0));
    // NVSM_WORDS_CSR_LATE=1 (experiment): the words table's build behind the loss kernel instead of at the step's start — it is
    // needed only behind the dx and dT products, whose MFMA-bound 0.18 ms it then runs next to, instead of next to the
    // HBM-bound loss kernel
    const bool words_csr_late_env = tune_.words_csr_late;
    const bool words_csr_late = words_csr_late_env && csr_after == 0 && !any_lazy;
    // (the previous step's hoisted decay of the words rows without entries — see `wrds` above — wrote rows this step's word gather
    //  may read: the main stream follows its event; issued here, in front of the builds that record the event anew)
    if (words_untouched_pending_) { NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_untouched_, 0)); words_untouched_pending_ = false; }
    if (csr_first) launch_csr_builds(ev_inputs_, words_csr_late ? 1 : 3);
    // (lazy dense decay: the gathers below bring the rows they read up to date on the fly — LazyView — and the row passes of
    //  the update do it for real; nothing waits for the sorts here)

    // F3: phrase representations (objective.cu:126-130). The previous step's dT GEMM may still be reading its phrase
    // matrix on the side stream: write the other one.
    if (T_pending_) phrase_p_ = (phrase_p_ == phrase_.p) ? phrase_alt_.p : phrase_.p;
    // NVSM_JOIN_E (experiments): where the main stream waits for the previous step's documents update — 0 = right before
    // the loss kernel (its first reader), 1 = before the projection GEMM, 2 = before the word gather
    const int join_e_at = tune_.join_e;
    if (join_e_at == 2) join_E();
    if (words_tail_pending_) { join_T(); words_tail_pending_ = false; }      // the previous step's streaming decay of the words table (step())

    // F3 inside F5 (round 6): where the forward product's kernel forms the phrase rows itself as it stages them — the gather
    // kernel's arithmetic in its order, `phrase` written on the way for the dT product — the gather is not a launch of its own:
    // one launch, one gap and a write + read of the phrase matrix less at the head of the critical stream. Not with the phrase
    // normaliser (it sits between the two) and not in the experiments that queue the CSR builds behind the gather.
    const LazyView words_view = lazy_view(words_);
    const bool l2p_fwd = cfg_.l2_normalize_phrase_reprs != 0;
    const bool gather_in_product = gather_fused_at(B) && !l2p_fwd && csr_after != 1;
    if (!gather_in_product) {
        const bool l2p = l2p_fwd;
        const LazyView& lv = words_view;
        timed_launch(prof, "gather_mean_words", stream_, /*single=*/!l2p, [&] {
            launch_gather_mean(words_.P.p, dw, widx_.p, wwts_, w, B, l2p ? phrase_raw_.p : phrase_p_, stream_, &lv);
            // optional phrase normaliser (objective.cu:136-142): the raw means stay cached for its backward pass
            if (l2p) launch_l2_rows_forward(phrase_raw_.p, B, dw, phrase_p_, phrase_norms_.p, stream_);
        });
    }
    if (csr_after == 1 && !csr_first) { NVSM_HIP_CHECK(hipEventRecord(ev_gathered_, stream_)); launch_csr_builds(ev_gathered_); }
    if (!gather_in_product) debug_check(phrase_p_, B * dw, 0);                  // CHECK_MATRIX(*result->phrase_reprs_), objective.cu:134,141

    // F5: projection GEMM  pre[B][de] = phrase[B][dw] · Tt[dw][de] (+ b when no BN)   (params.cu:417-421)
    // F6 (first half): with batch-norm the column sums Σx, Σx² of the projection ride in the GEMM epilogue
    join_T();        // the previous step's projection update (after its dT GEMM, the last reader of dy)
    if (join_e_at == 1) join_E();
    if (gather_in_product) {
        const GatherFused gf{words_.P.p, widx_.p, wwts_, w, &words_view};
        bool launched = false;
        {
            PROF("gemm_fwd");
            if (B <= gemm_rows_max_m())
                launched = launch_gemm_rsplit(0, phrase_p_, T_.p, pre_.p, static_cast<int>(B), de, dw, dw, de, de, 1.f,
                                              cfg_.batch_normalization ? nullptr : b_.p, stream_, cfg_.batch_normalization ? stats_fwd_ : nullptr,
                                              &sums_fwd_.ws, nullptr, 0.f, &split_fwd_, nullptr, &gf);
        }
        if (!launched) throw Error(NVSM_ERR_UNSUPPORTED, "forward product with the gather inside refused a shape its caller had checked");
        prof.note("gather_in_product");
        debug_check(phrase_p_, B * dw, 0);
    } else {
        PROF("gemm_fwd");
        launch_gemm(0, 0, phrase_p_, T_.p, pre_.p, static_cast<int>(B), de, dw, dw, de, de, 1.f,
                    cfg_.batch_normalization ? nullptr : b_.p, 1, 0, stream_,
                    cfg_.batch_normalization ? stats_fwd_ : nullptr, nullptr, 0.f, nullptr,
                    /*busy_chip=*/words_.lazy || ents_.lazy,       // long sorts and a long documents-update tail next to it
                    &sums_fwd_.ws, &split_fwd_);
    }

    const double B_global = static_cast<double>(B) * ((cfg_.world_size > 1) ? cfg_.world_size : 1);
    const double bn_n = (cfg_.world_size > 1 && cfg_.sync_batch_norm) ? B_global : static_cast<double>(B);
    if (csr_after == 2 && !csr_first) { NVSM_HIP_CHECK(hipEventRecord(ev_gathered_, stream_)); launch_csr_builds(ev_gathered_); }
    debug_check(pre_.p, B * de, 1);                     // CHECK_MATRIX(*result->word_projections_), objective.cu:152
    // F6: batch statistics (cudnn_utils.cu:107-124), ε = 1e-4 (objective.cu:114)
    if (cfg_.batch_normalization && cfg_.world_size > 1 && cfg_.sync_batch_norm) {
        PROF("allreduce_bn_stats");
        allreduce_f64(stats_fwd_, 2 * de);
    }
    // (μ and 1/sqrt(σ²+ε) themselves are evaluated by the loss kernel from these sums)

    // F7–F16 + B1–B4: fused loss
    join_E();        // the previous step's documents update: reads proj / coef, writes E
    {
        // Timed (bench.py's roofline kernel, inside its timed region): the event pair rides on the loss kernel itself as its
        // start / stop events — the kernel's own execution time, and no record packets around it on the critical stream
        // (two plain records cost the step ~15 us). Not when the kernel already carries ev_loss_ (small batches).
        hipEvent_t prof_start = nullptr, prof_stop = nullptr;
        const bool prof_bound = !loss_stop_event_ && stop_events_enabled() && prof.bind("loss_fused", &prof_start, &prof_stop);
        RangeScope loss_range("loss_fused");
        if (!prof_bound) prof.begin("loss_fused", stream_);
        LossArgs a{};
        a.pre = pre_.p; a.bn_mean = bn_mean_.p; a.bn_inv_std = bn_inv_std_.p; a.bias = b_.p;
        a.bn_sums = stats_fwd_; a.bn_n = bn_n; a.bn_eps = 1e-4f;
        a.E = ents_.P.p; a.E_rows = ents_.rows; a.ids = ids_p_; a.inst_w = instw_;
        a.proj = proj_.p; a.dy = dy_.p; a.coef = coef_.p; a.probs = probs_.p; a.pp = pp_.p;
        a.loss_acc = stats_bwd_; a.colstats = stats_bwd_ + 1; a.sums = sums_bwd_.ws;
        a.B = B; a.de = de; a.R = R_; a.k = k;
        a.bn = cfg_.batch_normalization; a.nonlinearity = cfg_.nonlinearity;
        a.l2_entity = cfg_.l2_normalize_entity_reprs;
        a.rebalance = (!cfg_.bias_negative_samples && k > 1);                                 // objective.cu:268
        a.sig_eps = cfg_.clip_sigmoid ? 1e-7f : 0.f;                                            // :245-246
        a.sig_hi = static_cast<float>(1.0 - static_cast<double>(a.sig_eps));
        a.d_eps = cfg_.clip_sigmoid ? 1e-6f : 0.f;                                              // :367-368
        a.d_hi = 1.0 - static_cast<double>(a.d_eps);
        a.inv_batch = static_cast<float>(std::exp(-std::log(B_global)));                        // :354
        a.neg_scale = static_cast<float>((static_cast<double>(static_cast<float>(k)) + 1.0) /
                                         (2.0 * static_cast<double>(static_cast<float>(k))));   // :270-273
        a.clip_min = std::nextafter(-1.0f, -1.0f - 1e-5f);                                      // cuda_utils.h:91-96
        a.clip_max = std::nextafter(1.0f, 1.0f + 1e-5f);
        a.inv_de = static_cast<float>(std::exp(-std::log(static_cast<double>(de))));
        if (ents_.lazy) {
            if (loss_reads_lazily(a.de, a.R, a.l2_entity != 0)) a.lazyE = lazy_view(ents_);
            else {
                // the generic loss kernel (odd dimensions, entity normaliser) reads the rows as they are: the documents of
                // this batch (the touched list of their CSR) are brought up to date first, behind the sort
                NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_csr_ents_, 0));
                Csr ce = csr_of(ents_, Bu * R_);
                lazy_refresh(ents_, &ce, stream_);
            }
        }
        if (loss_reads_lazily(a.de, a.R, a.l2_entity != 0) && a.R <= 17 && loss_two_row_sets(a.E_rows, a.de)) prof.note("loss_two_row_sets");
        // (the fused step at small batches starts the documents update behind the loss kernel: ev_loss_ rides on the kernel)
        if (loss_stop_event_) { launch_and_record(loss_stop_event_, stream_, [&] { launch_loss(a, stream_); }); loss_stop_event_ = nullptr; }
        else if (prof_bound) {
            set_launch_events(prof_start, prof_stop);
            launch_loss(a, stream_);
            hipEvent_t left_start = nullptr;
            if (hipEvent_t left = take_launch_events(&left_start)) {      // (nothing was launched: an empty interval)
                NVSM_HIP_CHECK(hipEventRecord(left_start, stream_));
                NVSM_HIP_CHECK(hipEventRecord(left, stream_));
            }
        } else launch_loss(a, stream_);
        if (!prof_bound) prof.end(stream_);
    }
    if (csr_after == 3 && !csr_first) { NVSM_HIP_CHECK(hipEventRecord(ev_gathered_, stream_)); launch_csr_builds(ev_gathered_); }
    if (words_csr_late) { NVSM_HIP_CHECK(hipEventRecord(ev_words_late_, stream_)); launch_csr_builds(ev_words_late_, 2); }
    NVSM_HIP_CHECK(hipGetLastError());      // a failed launch of any kernel above surfaces here, not at the next sync
    have_forward_ = true;
    if (debug_) {
        if (any_lazy) { NVSM_HIP_CHECK(hipStreamSynchronize(stream_)); lazy_flush_all(); }
        debug_check(proj_.p, B * de, 2); debug_check(probs_.p, N, 3); debug_check(dy_.p, B * de, 4);
        debug_check(words_.P.p, static_cast<int64_t>(words_.P.n), 7); debug_check(ents_.P.p, static_cast<int64_t>(ents_.P.n), 8);
        debug_check(T_.p, static_cast<int64_t>(T_.n), 9);
        NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
        raise_device_error();                // ids out of range / non-finite values are reported by the call that saw them
    }
}

// ---------------------------------------------------------------------------------------------
// compute_gradients — cpp/objective.cu:315-481, cpp/params.cu:453-535
// ---------------------------------------------------------------------------------------------
void Model::compute_gradients() {
    if (!have_forward_) throw Error(NVSM_ERR_STATE, "compute_gradients requires compute_cost");
    NVSM_HIP_CHECK(hipSetDevice(cfg_.device));
    RangeScope range_cg("ComputeGradients");            // cpp/main.cu:414
    backward_dx();
    backward_T(stream_);
    NVSM_HIP_CHECK(hipGetLastError());
    have_grads_ = true;
    if (debug_) {
        debug_check(gphrase_.p, B_ * cfg_.word_repr_size, 5);
        debug_check(gT_.p, static_cast<int64_t>(gT_.n), 6);
        NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
        raise_device_error();
    }
}

void Model::backward_dx() {
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size, w = cfg_.window_size;
    const int64_t B = B_;
    const bool dp = cfg_.world_size > 1;
    const double B_global = static_cast<double>(B) * (dp ? cfg_.world_size : 1);

    const bool need_msq = cfg_.update_method == NVSM_ADAGRAD ||
                          (cfg_.update_method == NVSM_ADAM && cfg_.adam_mode != NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE);
    const bool l2p = cfg_.l2_normalize_phrase_reprs != 0;
    const float inv_w = static_cast<float>(std::exp(-std::log(static_cast<double>(w))));
    const float inv_dw = static_cast<float>(std::exp(-std::log(static_cast<double>(dw))));
    // Per-rank batch sizes (gemm_rows.hip): batch-norm backward, the dx·T product and the rows' mean of squares are ONE launch
    // — the kernel owns whole rows of dy, applies dx = invσ·(dy − (dβ + x̂·dγ)/N) as it loads them (writing dx back for the
    // dT product) and finishes each row's sum of squares itself. Three launches and two gaps less on the critical stream.
    const bool sync_bn_order = !dp || cfg_.sync_batch_norm;       // (per-shard batch-norm under DP reduces AFTER bn_dx: separate launches)
    // Large batches: the same fusion in the split-bf16 kernel (gemm_split.hip). Its launch carries ev_bwdx_, so the planes of T
    // must not be cut by a launch of their own between the two: cut here if they are stale.
    const bool split_fuse = tune_.split_fuse;      // A/B runs
    const bool big = B > gemm_rows_max_m();
    auto split_ready = [&] {
        if (!split_bwd_.ready) { launch_gemm_split_planes(1, T_.p, dw, de, de, split_bwd_.planes, stream_); split_bwd_.ready = true; }
    };
    // (the same for the row-panel kernel of the per-rank batch sizes, gemm_rsplit.hip)
    auto rsplit_ready = [&] {
        if (!split_bwd_.rready && B >= 512 && gemm_rsplit_covers(1, static_cast<int>(B), dw, de, false, need_msq, false)) {
            launch_gemm_rsplit_planes(1, T_.p, dw, de, de, split_bwd_.rplanes, stream_);
            split_bwd_.rready = true;
        }
    };
    auto dx_product = [&](const BnDxFused* fused) {
        if (big) return launch_gemm_split(1, dy_.p, T_.p, gphrase_.p, static_cast<int>(B), dw, de, de, de, dw, inv_w, nullptr, stream_, nullptr,
                                          nullptr, need_msq ? msq_w_.p : nullptr, inv_dw, &split_bwd_, fused);
        if (launch_gemm_rsplit(1, dy_.p, T_.p, gphrase_.p, static_cast<int>(B), dw, de, de, de, dw, inv_w, nullptr, stream_, nullptr,
                               nullptr, need_msq ? msq_w_.p : nullptr, inv_dw, &split_bwd_, fused)) return true;
        return launch_gemm_rows(1, dy_.p, T_.p, gphrase_.p, static_cast<int>(B), dw, de, de, de, dw, inv_w, nullptr, stream_, nullptr,
                                nullptr, need_msq ? msq_w_.p : nullptr, inv_dw, fused);
    };
    auto fused_covers = [&](bool bn) {
        if (B < 512 || (big && !split_fuse)) return false;
        return big ? gemm_split_covers(1, static_cast<int>(B), dw, de, bn) : gemm_rows_covers(1, static_cast<int>(B), dw, de, false, need_msq, bn);
    };
    if (cfg_.batch_normalization && sync_bn_order && !l2p && fused_covers(true)) {
        if (dp) allreduce_f64(stats_bwd_, 1 + 2 * de);
        if (big) split_ready(); else rsplit_ready();
        BnDxFused bn{dy_.p, pre_.p, bn_mean_.p, bn_inv_std_.p, stats_bwd_ + 1, dbeta_.p, dgamma_.p, gb_.p, dp ? B_global : static_cast<double>(B)};
        bool launched = false;
        {
            PROF("gemm_bwd_x");
            launch_and_record(ev_bwdx_, stream_, [&] { launched = dx_product(&bn); });
        }
        if (launched) {
            // dx is final when this kernel is through: the dT GEMM of the fused step follows it (a wait on a kernel-borne event
            // must be issued right behind the launch, see below)
            if (dx_follower_) NVSM_HIP_CHECK(hipStreamWaitEvent(dx_follower_, ev_bwdx_, 0));
            if (dp) loss_reduced_ = true;
            return;
        }
        // (shape not covered: the event was recorded on an empty launch; fall through to the separate kernels. The statistics
        //  have been all-reduced already under DP, which the code below must not repeat.)
        throw Error(NVSM_ERR_UNSUPPORTED, "fused backward GEMM refused a shape its caller had checked");
    }

    // ... and without batch-norm (the LSE recipe) the same kernel finalises the bias gradient Σdy and the rows' mean of squares:
    // colsum_finalize + GEMM + sum_parts as one launch
    if (!cfg_.batch_normalization && !l2p && fused_covers(false)) {
        // (dp_fold(): Σdy only feeds the bias gradient here — it and the loss word ride on the dT all-reduce, backward_T)
        if (dp && !dp_fold()) allreduce_f64(stats_bwd_, 1 + de);
        if (big) split_ready(); else rsplit_ready();
        BnDxFused bias_only{nullptr, nullptr, nullptr, nullptr, stats_bwd_ + 1, nullptr, nullptr, gb_.p, 1.0};
        bool launched = false;
        {
            PROF("gemm_bwd_x");
            launch_and_record(ev_bwdx_, stream_, [&] { launched = dx_product(&bias_only); });
        }
        if (!launched) throw Error(NVSM_ERR_UNSUPPORTED, "fused backward GEMM refused a shape its caller had checked");
        if (dx_follower_) NVSM_HIP_CHECK(hipStreamWaitEvent(dx_follower_, ev_bwdx_, 0));
        if (dp && !dp_fold()) loss_reduced_ = true;
        return;
    }

    // B5: bias gradient / BN backward (params.cu:509-521)
    {
        PROF("bn_backward");
        // dx is final behind the last kernel of this block: the dT GEMM of the fused step may start (ev_dx_ rides on that
        // kernel as its completion event)
        // (A wait on a kernel-borne event is resolved when it is ISSUED: issued after more kernels have been queued behind
        //  the one that carries the event, it ends up behind those too — the dT GEMM started after the dx GEMM's row
        //  statistics instead of next to the dx GEMM. The stream that is to follow dx waits right here.)
        auto dx_final = [&](auto&& launch) {
            launch_and_record(ev_dx_, stream_, launch);
            if (dx_follower_) NVSM_HIP_CHECK(hipStreamWaitEvent(dx_follower_, ev_dx_, 0));
        };
        auto bn_dx = [&](double n) {
            launch_bn_dx(dy_.p, pre_.p, bn_mean_.p, bn_inv_std_.p, stats_bwd_ + 1, dbeta_.p, dgamma_.p, gb_.p, n, B, de, stream_);
        };
        auto colsum = [&] { launch_colsum_finalize(stats_bwd_ + 1, de, gb_.p, stream_); };
        if (cfg_.batch_normalization) {
            if (dp && cfg_.sync_batch_norm) {
                allreduce_f64(stats_bwd_, 1 + 2 * de);
                dx_final([&] { bn_dx(B_global); });
            } else if (dp && dp_fold()) {
                // per-shard statistics: dx needs nothing from the other ranks; the bias gradient (this shard's dβ, written by bn_dx)
                // and the loss word are summed over the ranks together with dT (backward_T): ONE collective per step
                dx_final([&] { bn_dx(static_cast<double>(B)); });
            } else if (dp) {
                bn_dx(static_cast<double>(B));
                allreduce_f64(stats_bwd_, 1 + 2 * de);
                dx_final(colsum);
            } else {
                dx_final([&] { bn_dx(static_cast<double>(B)); });
            }
        } else {
            if (dp && !dp_fold()) allreduce_f64(stats_bwd_, 1 + de);
            dx_final(colsum);
        }
    }
    // B7 + B9: gphrase[B][dw] = dx[B][de] · T (stored [dw][de]) / w   (objective.cu:447-476)
    {
        PROF("gemm_bwd_x");
        // the words update of Adam (sparse / dense_update) and Adagrad needs mean_t(gphrase[b][t]²) per window
        // (cpp/updates_adam.cu:232-240, updates_adagrad.cu:136-143): emitted by the GEMM epilogue, per column tile
        // (A/B, interleaved: 1.235 ms per step with the epilogue fusion vs 1.262 ms with a separate row-mean-of-squares pass)
        int msq_parts = 0;
        launch_gemm(0, 1, dy_.p, T_.p, gphrase_.p, static_cast<int>(B), dw, de, de, de, dw, l2p ? 1.f : inv_w, nullptr, 1, 0, stream_,
                    nullptr, (need_msq && !l2p) ? msq_parts_.p : nullptr, inv_dw, &msq_parts, false, nullptr, &split_bwd_);
        // ev_bwdx_: the dx GEMM, the last reader of T before its update, is through (and gphrase / its row statistics final)
        if (l2p) {      // Normalizer::backward, then the division by the window (objective.cu:461-476); mean of squares of the result
            launch_l2_rows_backward(gphrase_.p, phrase_raw_.p, phrase_norms_.p, B, dw, inv_w, gphrase_.p,
                                    need_msq ? msq_w_.p : nullptr, stream_);
            NVSM_HIP_CHECK(hipEventRecord(ev_bwdx_, stream_));
        } else if (need_msq) {
            launch_and_record(ev_bwdx_, stream_, [&] { launch_sum_parts(msq_parts_.p, msq_parts, B, msq_w_.p, B, stream_); });
        } else {
            NVSM_HIP_CHECK(hipEventRecord(ev_bwdx_, stream_));
        }
    }
    if (dp && !dp_fold()) loss_reduced_ = true;
}

// Data parallel: may the step's small f64 all-reduce ride on the dT all-reduce? Only with synchronised batch-norm statistics does
// anything in front of the dT product need another rank's sums (dx = f(Σdy, Σdy·x̂ over the GLOBAL batch)); with per-shard
// statistics, and without batch-norm, [loss | Σdy] feed nothing but the bias gradient and the reported loss: one collective per step.
bool Model::dp_fold() const {
    return cfg_.world_size > 1 && !(cfg_.batch_normalization && cfg_.sync_batch_norm) && tune_.dp_fold;
}

// B6: ∂T (stored [dw][de]) = phraseᵀ[dw x B] · dx[B x de], split-K over the batch   (params.cu:526-531)
void Model::backward_T(hipStream_t strm) {
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size;
    const int64_t B = B_;
    const bool dp = cfg_.world_size > 1;
    {
        // (timed by events that ride on the launches themselves: the product's and the slab reduce's own execution times, which
        //  is what a kernel trace reports — a record pair around the group on a side stream timed the wait for CUs as well)
        const int slabs = gemm_split_k_slabs(static_cast<int>(B), gemm_slabs_want_);
        const size_t stride = static_cast<size_t>(de) * dw;
        auto reduce = [&](int n) {
            // (the fused step without collectives: the projection update that follows on this stream adds the slabs up itself)
            if (fuse_slab_sum_) { pending_slabs_ = n; prof.note("slab_sum_in_update"); return; }
            timed_launch(prof, "gemm_bwd_T_reduce", strm, true, [&] { launch_splitk_reduce(gT_partial_.p, n, stride, gT_.p, static_cast<int64_t>(stride), strm); });
        };
        pending_slabs_ = 0;
        if (tune_.skip_dt) {
            prof.note("dt_skipped_timing_only");      // (experiments build: gT keeps the previous step's values)
        } else if (use_dt()) {
            // the split-K product on the bf16 matrix pipe (gemm_dt.hip): two workgroups per slab.
            // Slabs: the kernel ALONE is fastest with a workgroup on every CU (128 slabs: 55 us at batch 51 200), but in a step it
            // runs next to the documents pass, which wants the CUs it leaves free and the bandwidth its partials do not take: on
            // the main stream 48 slabs (96 workgroups, 15 MB of partials) — 0.891 ms per step against 0.90 with 128 and 0.899 with
            // 32; full_adam 0.764 / 0.79 / 0.763 —; on side stream 2 next to both table passes of lazily decayed tables the same 48
            // (|V| = 500 k, |D| = 2 M: 1.696 ms against 1.739 with 16 slabs, 1.704 with 32 / 64, 1.717 with 128, 1.86 with 8 — with few
            // slabs the product, trickling onto CUs as they fall empty, is what the next step waits for; |V| = 50 k, |D| = 2 M:
            // 1.497 against 1.486 with 16). Interleaved A/B, tools/ab_shapes.sh; NVSM_DT_SLABS (experiments build) overrides.
            const int want = std::min(tune_.dt_slabs > 0 ? tune_.dt_slabs : 48, gemm_dt_default_slabs(static_cast<int>(B), num_cus_));
            const int dslabs = gemm_dt_slabs(static_cast<int>(B), want);
            bool ok = true;
            timed_launch(prof, "gemm_bwd_T", strm, true, [&] {
                ok = launch_gemm_dt(phrase_p_, dy_.p, dslabs == 1 ? gT_.p : gT_partial_.p, dw, de, static_cast<int>(B), dw, de, want, strm);
            });
            if (!ok) throw Error(NVSM_ERR_UNSUPPORTED, "dT product refused a shape its caller had checked");
            prof.note("dt_split_bf16");
            if (dslabs > 1) reduce(dslabs);
        } else if (use_dtw_at(B)) {
            // per-rank batches: the split-bf16 product in workgroups of one wave (gemm_dtw.hip), which go wherever one wave of the
            // table passes next to it has left
            // Slabs: twelve (480 waves at the metric's dimensions: 40 tiles x 12) — in the step 8 and 12 tie, 4 / 6 lose 3-15 % at the
            // LSE shape and at batch 12 800 (too few waves: 78 us alone at batch 6 400 against 25 with 16), 16 / 32 lose 0.5-2 % (more
            // partials for the projection update to add up on the chain the next forward product waits for); alone the kernel likes
            // 32-64 (20 us at batch 6 400; the tiled fp32 kernel 37, gemm_dt 18): profiles/r06_exp_dtw_*.txt. NVSM_DTW_SLABS overrides.
            const int want = tune_.dtw_slabs > 0 ? tune_.dtw_slabs : kDtwSlabs;
            const int wslabs = gemm_dtw_slabs(static_cast<int>(B), want);
            bool ok = true;
            timed_launch(prof, "gemm_bwd_T", strm, true, [&] {
                ok = launch_gemm_dtw(phrase_p_, dy_.p, wslabs == 1 ? gT_.p : gT_partial_.p, dw, de, static_cast<int>(B), dw, de, want, strm);
            });
            if (!ok) throw Error(NVSM_ERR_UNSUPPORTED, "dT product (wave-sized) refused a shape its caller had checked");
            prof.note("dt_wave_sized");
            if (wslabs > 1) reduce(wslabs);
        } else if (slabs == 1) {
            timed_launch(prof, "gemm_bwd_T", strm, true, [&] {
                launch_gemm(1, 0, phrase_p_, dy_.p, gT_.p, dw, de, static_cast<int>(B), dw, de, de, 1.f, nullptr, 1, 0, strm);
            });
        } else {
            timed_launch(prof, "gemm_bwd_T", strm, true, [&] {
                launch_gemm(1, 0, phrase_p_, dy_.p, gT_partial_.p, dw, de, static_cast<int>(B), dw, de, de, 1.f, nullptr,
                            gemm_slabs_want_, stride, strm, nullptr, nullptr, 0.f, nullptr,
                            /*busy_chip=*/strm != stream_);      // fused step: next to the words / documents update
            });
            reduce(slabs);
        }
    }
    // data parallel: one all-reduce of the dense projection gradient over xGMI
    if (dp) {
        PROF_ON("allreduce_grad", strm);
        const int64_t nT = static_cast<int64_t>(de) * dw;
        if (dp_fold()) {
            // [dT | db | loss hi | loss lo] in one f32 all-reduce (gb_ holds this rank's Σdy — written by bn_dx / the backward
            // product's prologue / colsum_finalize, all in front of this stream's dT product; the loss word is final since the loss kernel)
            launch_dp_pack_tail(gb_.p, stats_bwd_, gT_.p + nT, de, strm);
            allreduce_f32(gT_.p, nT + de + 2, strm);
            // (the summed loss goes to a word of its own, read on THIS stream — get_cost / step_deferred —: the loss kernel's word
            //  belongs to the main stream, which may be a step ahead of side stream 2 by the time this runs)
            launch_dp_unpack_tail(gT_.p + nT, gb_.p, loss_red_.p, de, strm);
            loss_reduced_ = true; loss_folded_ = true; loss_stream_ = strm;
            prof.note("dp_one_collective");
        } else {
            allreduce_f32(gT_.p, nT, strm);
        }
        cost_valid_ = false;
    }
}

float Model::scaled_regularization_lambda() const {
    const double Bg = static_cast<double>(B_ > 0 ? B_ : cfg_.max_batch_size) * (cfg_.world_size > 1 ? cfg_.world_size : 1);
    return cfg_.regularization_lambda / static_cast<float>(Bg);      // intermediate_results.cu:126-129
}

// ForwardResult::get_cost (intermediate_results.cu:80-124): −(Σ mass)/B, one D2H + stream sync.
float Model::get_cost() {
    if (!have_forward_) throw Error(NVSM_ERR_STATE, "get_cost requires compute_cost");
    if (!cost_valid_) {
        double s = 0.0;
        const double* src = stats_bwd_;
        if (cfg_.world_size > 1 && !loss_reduced_) {
            // Data parallel, before the backward pass has all-reduced [loss | Σdy | Σdy·x̂]: the loss word still holds this
            // rank's share only. All-reduce a copy (a collective: every rank must make the same call, as every rank makes the
            // same compute_cost / compute_gradients calls) and leave the word itself to the backward pass.
            NVSM_HIP_CHECK(hipMemcpyAsync(loss_tmp_.p, stats_bwd_, sizeof(double), hipMemcpyDeviceToDevice, stream_));
            allreduce_f64(loss_tmp_.p, 1);
            src = loss_tmp_.p;
        }
        hipStream_t ls = stream_;
        if (loss_folded_) { src = loss_red_.p; ls = loss_stream_; }      // (summed behind the dT all-reduce, on that stream: backward_T)
        NVSM_HIP_CHECK(hipMemcpyAsync(&s, src, sizeof(double), hipMemcpyDeviceToHost, ls));
        NVSM_HIP_CHECK(hipStreamSynchronize(ls));
        raise_device_error();
        const double Bg = static_cast<double>(B_) * (cfg_.world_size > 1 ? cfg_.world_size : 1);
        cost_ = -(s / Bg);
        cost_valid_ = true;
    }
    return static_cast<float>(cost_);
}

// ---------------------------------------------------------------------------------------------
// update — cpp/model.cu:187-220: entities → words → transform
// ---------------------------------------------------------------------------------------------
float Model::adam_bc(uint64_t t) const {
    const double b1 = static_cast<double>(0.9f), b2 = static_cast<double>(0.999f);
    return static_cast<float>(std::sqrt(1.0 - std::pow(b2, static_cast<double>(t))) / (1.0 - std::pow(b1, static_cast<double>(t))));
}

// Entries per level-1 chunk of a long row. The small-batch steps of SGD and Adagrad on NARROW rows (the LSE recipe: batch 4 096,
// 128-wide word rows, eight thread groups to a workgroup) end in the hottest row's chain of dependent round trips — a 64-entry chunk
// walked five entries at a time is thirteen of them —: chunks of 32 there, LSE 0.1562 -> 0.1504 ms. Not where a workgroup holds three
// rows' groups (300-wide rows: twice the chunks are nearly twice the chunk workgroups next to the other pass — Adagrad at batch
// 6 400 0.2384 -> 0.2457, the Adam modes +1...4 %) and not for large batches (the passes are bytes there).
int Model::chunk_entries(const TableState& t, int64_t n) const {
    return (cfg_.update_method != NVSM_ADAM && n <= kChunkSmallMaxEntries && t.dim <= 128) ? kChunkSmall : kChunk;
}

Csr Model::csr_of(TableState& t, int64_t n) {
    Csr c{};
    TableState::CsrIndex& x = t.idx[t.idx_cur];
    c.sorted_key = x.sorted_key.p; c.sorted_entry = x.sorted_entry.p;
    c.row_begin = x.csr_zeroed.p; c.row_end = x.csr_zeroed.p + t.rows; c.chunk_base = x.chunk_base.p;
    c.chunk_desc = x.chunk_desc.p; c.num_chunks = x.csr_zeroed.p + 2 * t.rows;
    c.num_touched = c.num_chunks + 2; c.touched = x.touched.p;
    c.partial = t.partial.p; c.partial_q = t.partial_q.p;
    c.chunk2_base = x.chunk2_base.p; c.chunk2_desc = x.chunk2_desc.p; c.chunk_order = x.chunk_order.p;
    c.partial2 = t.partial2.p; c.partial2_q = t.partial2_q.p;
    c.arrive_row = t.arrive_row.p; c.arrive2 = t.arrive2.p;
    c.n = n; c.rows = t.rows; c.max_chunks = t.max_chunks; c.max_chunks2 = t.max_chunks2;
    c.chunk = chunk_entries(t, n);
    return c;
}

void Model::build_csr(TableState& t, const int* keys, int64_t n, hipStream_t s) {
    if (t.idx_sets > 1) t.idx_cur ^= 1;      // the other set may still be read by the previous step's update
    TableState::CsrIndex& x = t.idx[t.idx_cur];
    // (the sort's first launch also clears the CSR's per-step counters: no memset launch)
    sort_pairs(t.sort_temp.p, t.sort_temp_bytes, &t.sort_epoch, keys, x.sorted_key.p, nullptr, x.sorted_entry.p, n, t.sort_bits, err_host_, s,
               x.csr_zeroed.p, csr_counter_ints(t.rows));
    launch_csr_build(csr_of(t, n), s, n > 0, x.chunk_order.p ? t.chunk_key.p : nullptr);
    if (x.chunk_order.p) launch_chunk_order(csr_of(t, n), t.chunk_key.p, t.chunk_key_sorted.p, t.sort_temp.p, t.sort_temp_bytes, s, /*keys_written=*/n > 0);
}

static void fill_adam_consts(RowPassArgs& a, float bc, float sl);
// Streaming (nt) loads / stores for state the row passes touch once per step, so that it does not displace the gradient
// rows they gather (each read 10-17 times) from the caches: 1 = documents moments, 2 = documents rows, 4 = word moments,
// 8 = word rows. Interleaved A/B at the bench shape: 0: 1.134, 1: 1.128, 3: 1.125, 5: 1.136, 9: 1.134, 15: 1.144 ms.
static int nt_mask() {
    return tuning().nt_mask;
}

// ---- lazy dense decay (kernels.h) ---------------------------------------------------------------------------------
void Model::lazy_refresh(TableState& t, const Csr* touched, hipStream_t s) {
    if (!t.lazy) return;
    LazyRefreshArgs r{};
    r.P = t.P.p; r.m = t.m.p;
    r.sc = t.lazy_scalar ? t.sc[0].p : nullptr;
    r.sc_snapshot = t.lazy_scalar ? t.sc[1].p : nullptr;
    r.stamp = t.stamp.p; r.rows = t.rows; r.dim = t.dim; r.now = t.updates_done;
    r.s_m = 1.f; r.s_v = 1.f;
    if (cfg_.update_method == NVSM_ADAM) { RowPassArgs c{}; fill_adam_consts(c, 1.f, 0.f); r.s_m = c.s_m; r.s_v = c.s_v; }
    std::memcpy(r.decay, t.decay_hist, sizeof(r.decay));
    int64_t max_rows = t.rows;
    if (touched) { r.list = touched->touched; r.list_count = touched->num_touched; max_rows = std::min<int64_t>(touched->n, t.rows); }
    PROF_ON(&t == &words_ ? "lazy_refresh_words" : "lazy_refresh_entities", s);
    launch_lazy_refresh(r, max_rows, s);
}

LazyView Model::lazy_view(const TableState& t) const {
    LazyView v{};
    if (!t.lazy) return v;
    v.stamp = t.stamp.p; v.now = t.updates_done;
    std::memcpy(v.decay, t.decay_hist, sizeof(v.decay));
    return v;
}

// before the passes of an update: the per-row scalar of the rows the batch touches, brought up to date into the
// snapshot the row pass reads (a few KB; P and m are refreshed by the row passes themselves)
void Model::lazy_scalar_snapshot(TableState& t, const Csr& c, hipStream_t s) {
    if (!t.lazy || !t.lazy_scalar) return;
    LazyRefreshArgs r{};
    r.scalars_only = 1;
    r.sc = t.sc[0].p; r.sc_snapshot = t.sc[1].p;
    r.stamp = t.stamp.p; r.rows = t.rows; r.dim = t.dim; r.now = t.updates_done;
    r.s_m = 1.f; r.s_v = 1.f;
    if (cfg_.update_method == NVSM_ADAM) { RowPassArgs k{}; fill_adam_consts(k, 1.f, 0.f); r.s_v = k.s_v; }
    r.list = c.touched; r.list_count = c.num_touched;
    PROF_ON(&t == &words_ ? "lazy_scalars_words" : "lazy_scalars_entities", s);
    launch_lazy_refresh(r, std::min<int64_t>(c.n, t.rows), s);
}

void Model::lazy_flush_all() {
    if (!words_.lazy && !ents_.lazy) return;
    settle_words_stamp();
    synchronize();
    lazy_refresh(words_, nullptr, stream_);
    lazy_refresh(ents_, nullptr, stream_);
    NVSM_HIP_CHECK(hipStreamSynchronize(stream_));
}

// the passes of one update of a lazy table: rows without entries are skipped (their decay stays pending); the per-row
// scalar is read from the snapshot the refresh left in sc[1] and written to sc[0] (no ping-pong flip: rows that are not
// visited must keep their value where it is)
void Model::lazy_begin_update(TableState& t, RowPassArgs& a, bool scalar_pingpong) {
    if (!t.lazy) return;
    a.lazy = 1;
    if (scalar_pingpong) { a.sc_in = t.sc[1].p; a.sc_out = t.sc[0].p; }
    a.pending = lazy_view(t);                                        // what the rows this pass visits sat out
    t.decay_hist[t.updates_done % kLazyHistory] = a.decay;           // factor of update number updates_done + 1 on P
}
// The words update ends the step's critical stream, and its stamps are first looked at by the next step's word gather: they are
// set by that step's prologue kernel (one launch less at the end of the chain; the touched-row list lives until the next CSR
// build, which follows the prologue), or by settle_words_stamp() when something else comes first.
void Model::settle_words_stamp() {
    if (!words_stamp_pending_) return;
    words_stamp_pending_ = false;
    TableState& t = words_;
    Csr c = csr_of(t, words_stamp_n_);
    PROF("lazy_stamp_words");
    launch_stamp_rows(c, t.stamp.p, t.updates_done, std::min<int64_t>(c.n, t.rows), stream_);
}

void Model::lazy_end_update(TableState& t, const Csr& c, hipStream_t s) {
    if (!t.lazy) return;
    t.updates_done += 1;
    if (&t == &words_ && s == stream_ && tune_.stamp_in_prologue && t.updates_done % kLazyHistory != 0) {
        words_stamp_pending_ = true;
        words_stamp_n_ = c.n;
        prof.note("lazy_stamp_words");
    } else {   // the touched rows carry this update
        PROF_ON(&t == &words_ ? "lazy_stamp_words" : "lazy_stamp_entities", s);
        launch_stamp_rows(c, t.stamp.p, t.updates_done, std::min<int64_t>(c.n, t.rows), s);
    }
    // the kernel arguments carry the factors of the last kLazyHistory updates: nobody may fall further behind
    if (t.updates_done % kLazyHistory == 0) lazy_refresh(t, nullptr, s);
}

static void fill_adam_consts(RowPassArgs& a, float bc, float sl) {
    const double b1 = static_cast<double>(0.9f), b2 = static_cast<double>(0.999f);
    a.one_m_b1 = static_cast<float>(1.0 - b1);
    a.one_m_b2 = static_cast<float>(1.0 - b2);
    a.s_m = static_cast<float>(1.0 - 1.0 * static_cast<double>(a.one_m_b1));     // storage.cu:65-67 with λ = 1, lr = 1−β
    a.s_v = static_cast<float>(1.0 - 1.0 * static_cast<double>(a.one_m_b2));
    a.bc = bc;
    a.eps = 1e-6f;                                                                 // updates.h:21
    a.c_reg = static_cast<float>((1.0 - b1) * static_cast<double>(sl));            // updates_adam.cu:208-212
}

void Model::update_entities(float lr, float sl, hipStream_t strm, hipEvent_t row_pass_after) {
    const UpdateInputs u = update_inputs();
    const int64_t N = u.B * R_;
    const int de = cfg_.entity_repr_size;
    TableState& t = ents_;
    Csr c = csr_of(t, N);
    RowPassArgs a{};
    a.table = 1; a.X = u.proj; a.coefs = u.coef; a.sq_src = u.pp; a.div = static_cast<uint32_t>(R_);
    a.div_magic = (uint64_t(1) << 37) / a.div + 1;
    a.P = t.P.p; a.m = t.m.p; a.v = t.vfull.p; a.dim = de;
    a.lr = lr; a.lambda = sl; a.eps = 1e-6f;
    a.decay = sl > 0.f ? static_cast<float>(1.0 - static_cast<double>(sl) * static_cast<double>(lr)) : 1.f;
    a.sc_in = t.sc[t.sc_cur].p; a.sc_out = t.sc[t.sc_cur ^ 1].p;
    bool swap_sc = false;
    switch (cfg_.update_method) {
        case NVSM_SGD: a.kind = ROW_SGD; a.sq_src = nullptr; a.dense = sl > 0.f; break;
        case NVSM_ADAGRAD: a.kind = ROW_ADAGRAD_ENT; a.dense = 1; swap_sc = true; break;
        default: {
            fill_adam_consts(a, adam_bc(t.t), sl);
            t.t += 1;
            a.dense = 1;
            if (cfg_.adam_mode == NVSM_ADAM_DENSE_UPDATE) { a.kind = ROW_ADAM_DENSE; swap_sc = true; }
            else if (cfg_.adam_mode == NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE) { a.kind = ROW_ADAM_FULL; a.sq_src = nullptr; a.decay = 1.f; }
            else { a.kind = ROW_ADAM_SPARSE_ENT; swap_sc = true; }
        }
    }
    if (cfg_.l2_normalize_entity_reprs) {
        // optional entity normaliser: the per-entry gradient rows are materialised (as the reference does) and scattered as
        // they are: source row = entry, coefficient 1, per-entry mean of squares
        launch_materialize_grad_entity_l2(u.coef, u.proj, t.P.p, u.ids, N, R_, de, grad_entity_.p, ge_msq_.p, strm);
        a.X = grad_entity_.p; a.coefs = nullptr; a.div = 1; a.div_magic = (uint64_t(1) << 37) + 1;
        if (a.sq_src) a.sq_src = ge_msq_.p;
    }
    // (Capping this grid so that GEMM workgroups of the other stream find free registers on every CU was measured in
    // the fused step: 1.30 -> 1.31-1.33 ms, no gain; RowPassArgs::max_blocks stays 0.)
    if (t.lazy) {
        lazy_begin_update(t, a, swap_sc);
        swap_sc = false;
        lazy_scalar_snapshot(t, c, strm);
    }
    a.nt_m = nt_mask() & 1; a.nt_p = (nt_mask() >> 1) & 1;
    if (row_pass_after) NVSM_HIP_CHECK(hipStreamWaitEvent(strm, row_pass_after, 0));
    int path;
    // (timed by an event pair that rides on the pass's own launch: the kernel's execution time in the step — bench.py
    //  roofline_update —, not the side stream's wait for it)
    timed_launch(prof, "row_pass_entities", strm, /*single=*/true, [&] { path = launch_table_pass(c, a, strm); });
    if (path == TABLE_PASS_ENTRY_WALK) prof.note("entry_walk_entities");
    if (swap_sc) t.sc_cur ^= 1;
    lazy_end_update(t, c, strm);
}

// what the LAST pass of an SGD / Adagrad words update looks like to the rows without entries (launch_untouched_rows): P *= decay
RowPassArgs Model::final_words_pass_args(float lr, float sl) {
    RowPassArgs a{};
    a.table = 0; a.kind = ROW_SGD; a.div = static_cast<uint32_t>(cfg_.window_size);
    a.div_magic = (uint64_t(1) << 37) / a.div + 1;
    a.P = words_.P.p; a.m = words_.m.p; a.v = words_.vfull.p; a.dim = cfg_.word_repr_size;
    a.lr = lr; a.lambda = sl; a.eps = 1e-6f;
    a.decay = sl > 0.f ? static_cast<float>(1.0 - static_cast<double>(sl) * static_cast<double>(lr)) : 1.f;
    a.dense = sl > 0.f;
    return a;
}

void Model::update_words(float lr, float sl) {
    const int dw = cfg_.word_repr_size, w = cfg_.window_size;
    const UpdateInputs u = update_inputs();
    const int64_t n = u.B * w;
    TableState& t = words_;
    Csr c = csr_of(t, n);
    RowPassArgs a{};
    a.table = 0; a.X = u.gphrase; a.wts = u.wwts; a.div = static_cast<uint32_t>(w);
    a.div_magic = (uint64_t(1) << 37) / a.div + 1;
    a.P = t.P.p; a.m = t.m.p; a.v = t.vfull.p; a.dim = dw;
    a.lr = lr; a.lambda = sl; a.eps = 1e-6f;
    a.decay = sl > 0.f ? static_cast<float>(1.0 - static_cast<double>(sl) * static_cast<double>(lr)) : 1.f;
    const int method = cfg_.update_method, mode = cfg_.adam_mode;
    // (the fused step queued the decay of the rows without entries behind the CSR build already — with these very lr and λ)
    const bool hoisted = words_untouched_hoisted_ && lr == hoist_lr_ && sl == hoist_sl_;
    if (words_untouched_hoisted_ && !hoisted) throw Error(NVSM_ERR_STATE, "the hoisted words decay was queued with another learning rate / lambda");
    words_untouched_hoisted_ = false;

    if (method == NVSM_SGD) {
        a.kind = ROW_SGD; a.dense = sl > 0.f; a.untouched_done = hoisted;
        lazy_begin_update(t, a, false);
        { PROF("row_pass_words"); launch_table_pass(c, a, stream_, words_untouched_stream_); }
        lazy_end_update(t, c, stream_);
        return;
    }
    if (method == NVSM_ADAGRAD) {
        RowPassArgs s = a;                                             // accumulator pass (updates_adagrad.cu:136-158)
        // the row's accumulator is read and written by one lane only (no other lane of the row's group looks at it), so
        // it is updated in place: rows without entries keep their value without being visited at all
        s.kind = ROW_SCALAR_ACC; s.sq_src = u.msq_w; s.dense = 0;
        s.sc_in = t.sc[t.sc_cur].p; s.sc_out = t.sc[t.sc_cur].p;
        { PROF("adagrad_acc_words"); launch_table_pass(c, s, stream_, words_untouched_stream_); }
        { PROF("adagrad_scale_words"); launch_adagrad_scale(t.sc[t.sc_cur].p, u.widx, w, u.B, 1e-6f, scale_w_.p, stream_); }
        a.kind = ROW_SGD; a.src_scale = scale_w_.p; a.dense = sl > 0.f; a.untouched_done = hoisted;
        lazy_begin_update(t, a, false);
        { PROF("row_pass_words"); launch_table_pass(c, a, stream_, words_untouched_stream_); }
        lazy_end_update(t, c, stream_);
        return;
    }
    // Adam
    fill_adam_consts(a, adam_bc(t.t), sl);
    t.t += 1;
    if (mode == NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE) {
        a.kind = ROW_ADAM_FULL; a.dense = 1; a.decay = 1.f;
        { PROF("row_pass_words"); launch_table_pass(c, a, stream_, words_untouched_stream_); }
        return;
    }
    a.sq_src = u.msq_w; a.dense = 1;     // from the dx GEMM's epilogue
    a.sc_in = t.sc[t.sc_cur].p; a.sc_out = t.sc[t.sc_cur ^ 1].p;
    if (mode == NVSM_ADAM_DENSE_UPDATE) {
        a.kind = ROW_ADAM_DENSE;
        { PROF("row_pass_words"); launch_table_pass(c, a, stream_, words_untouched_stream_); }
        t.sc_cur ^= 1;
        return;
    }
    // SPARSE: moments, then the window-averaged direction, then the scatter (updates_adam.cu:332-384)
    a.kind = ROW_ADAM_MV;
    a.nt_m = (nt_mask() >> 2) & 1;
    lazy_begin_update(t, a, true);
    if (!words_snapshot_early_) lazy_scalar_snapshot(t, c, stream_);      // (else: done behind the CSR build, compute_cost)
    words_snapshot_early_ = false;
    int path;
    { PROF("row_pass_words_mv"); path = launch_table_pass(c, a, stream_, words_untouched_stream_); }
    if (path == TABLE_PASS_ENTRY_WALK) prof.note("entry_walk_words");
    if (!t.lazy) t.sc_cur ^= 1;
    { PROF("adam_u_words"); launch_adam_u(t.m.p, t.sc[t.sc_cur].p, dw, u.widx, w, u.B, a.bc, a.eps, U_.p, stream_); }
    RowPassArgs r = a;
    r.kind = ROW_SGD; r.X = U_.p; r.sq_src = nullptr; r.dense = sl > 0.f; r.wide = 1;
    r.nt_m = 0; r.nt_p = (nt_mask() >> 3) & 1;
    { PROF("row_pass_words_u"); launch_table_pass(c, r, stream_, words_untouched_stream_); }
    lazy_end_update(t, c, stream_);
}

void Model::update_transform(float lr, float sl, hipStream_t strm) {
    PROF_ON("transform_update", strm);
    TransformUpdateArgs a{};
    a.T = T_.p; a.b = b_.p; a.gT = gT_.p; a.gb = gb_.p;
    a.s0T = s0T_.p; a.s0b = s0b_.p; a.s1T = s1T_.p; a.s1b = s1b_.p;
    a.nT = cfg_.entity_repr_size * cfg_.word_repr_size; a.nb = cfg_.entity_repr_size;
    a.method = cfg_.update_method;
    a.lr = lr; a.lambda = sl; a.eps = 1e-6f;
    const double b1 = static_cast<double>(0.9f), b2 = static_cast<double>(0.999f);
    a.one_m_b1 = static_cast<float>(1.0 - b1); a.one_m_b2 = static_cast<float>(1.0 - b2);
    a.s_m = static_cast<float>(1.0 - 1.0 * static_cast<double>(a.one_m_b1));
    a.s_v = static_cast<float>(1.0 - 1.0 * static_cast<double>(a.one_m_b2));
    a.bc = adam_bc(t_transform_);
    if (cfg_.update_method == NVSM_ADAM) t_transform_ += 1;
    // T changes: its bf16 planes for the next step's two projection products are written by the update kernel itself (off the
    // critical path in the fused step: side stream 2, which the next projection GEMM joins anyway). Which layout: the kernels
    // this step's batch size runs on (a step of another size finds its planes stale and cuts them itself).
    planes_stale();
    a.de = cfg_.entity_repr_size;
    if (pending_slabs_ > 0) { a.partial = gT_partial_.p; a.slabs = pending_slabs_; a.slab_stride = static_cast<size_t>(a.nT); pending_slabs_ = 0; }
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size, B = static_cast<int>(B_);
    if (gemm_split_products() && tune_.planes_in_update) {
        if (B_ > gemm_rows_max_m()) {
            if (gemm_split_covers(0, B, de, dw, false)) { a.pt[0] = gemm_split_plane_target(de, dw, split_fwd_.planes, 0); split_fwd_.ready = true; }
            if (gemm_split_covers(1, B, dw, de, false)) { a.pt[1] = gemm_split_plane_target(dw, de, split_bwd_.planes, 1); split_bwd_.ready = true; }
        } else if (B >= 512) {
            if (gemm_rsplit_covers(0, B, de, dw, cfg_.batch_normalization != 0, false, false)) {
                a.pt[0] = gemm_rsplit_plane_target(de, dw, split_fwd_.rplanes, 0); split_fwd_.rready = true;
            }
            if (gemm_rsplit_covers(1, B, dw, de, false, false, false)) {
                a.pt[1] = gemm_rsplit_plane_target(dw, de, split_bwd_.rplanes, 1); split_bwd_.rready = true;
            }
        }
    }
    launch_transform_update(a, strm);
    if (!tune_.planes_in_update) cut_transform_planes(strm);
}

// The dispatch of a step at `batch` windows in one place, as text (nvsm_describe): the three projection products' kernels, where
// the dT product and the CSR builds run, the tables' decay mode, and every switch that is off its default. Mirrors launch_gemm's
// order of preference (gather_gemm.hip) and the rules of step(); tests/test_gpu_switches.py holds it against the profiler's notes.
std::string Model::describe(int64_t batch) const {
    TuningScope scope(&tune_);
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size;
    const int B = static_cast<int>(std::min<int64_t>(std::max<int64_t>(batch, 1), cfg_.max_batch_size));
    const bool need_msq = cfg_.update_method == NVSM_ADAGRAD || (cfg_.update_method == NVSM_ADAM && cfg_.adam_mode != NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE);
    const bool bn = cfg_.batch_normalization != 0, l2p = cfg_.l2_normalize_phrase_reprs != 0;
    auto product = [&](int b_layout, int N, int K, bool stats, bool rowsq, bool fused_bn) -> std::string {
        const int rows_max = gemm_rows_max_m();
        if (B >= 512 && B <= rows_max && gemm_rsplit_covers(b_layout, B, N, K, stats, rowsq, fused_bn))
            return "gemm_rsplit (3 bf16 planes, " + std::to_string(gemm_split_products()) + " of 9 products, 32-row panels)";
        if (B >= 512 && B <= rows_max && gemm_rows_covers(b_layout, B, N, K, stats, rowsq, fused_bn)) return "gemm_rows (exact fp32 MFMA, 32-row panels)";
        if (B > rows_max && gemm_split_covers(b_layout, B, N, K, fused_bn))
            return "gemm_split (3 bf16 planes, " + std::to_string(gemm_split_products()) + " of 9 products)";
        if (B >= 1024 && K % 4 == 0 && N % 4 == 0) return "gemm_tstat (exact fp32 MFMA, projection stationary in LDS) or tiled";
        return "gemm_f32_mfma (exact fp32 MFMA, 128 x 128 tiles)";
    };
    std::string out = "batch " + std::to_string(B) + ": forward " + product(0, de, dw, bn, false, false) +
                      (gather_fused_at(B) && !l2p ? " with the word gather inside" : "");
    const bool fuse = !l2p && B >= 512 && (B <= gemm_rows_max_m() || tune_.split_fuse);
    out += " | backward " + product(1, dw, de, false, need_msq && !l2p, fuse) + (fuse ? " with the batch-norm backward / bias gradient inside" : "");
    // (B_ decides use_dt() / dt_on_main() at run time: evaluated here for `B`)
    const bool dt = use_dt_at(B);
    const bool dt_main = dt_on_main_at(B);
    out += std::string(" | dT ") + (dt ? "gemm_dt (3 bf16 planes, split-K)" : (use_dtw_at(B) ? "gemm_dtw (3 bf16 planes, split-K, one-wave workgroups)" : "gemm_f32_mfma / gemm_panel split-K (exact fp32 MFMA)")) +
           (dt_main ? " on the main stream" : " on side stream 2");
    if (loss_reads_lazily(de, static_cast<int>(R_), cfg_.l2_normalize_entity_reprs != 0))      // (= the row-gathering kernel covers the shape)
        out += std::string(" | loss loss_rows (") + (R_ <= 17 && loss_two_row_sets(ents_.rows, de) ? "two row sets per wave: documents table beyond the Infinity Cache" : "one row set per wave") + ")";
    else out += " | loss loss_kernel (generic)";
    out += std::string(" | tables: words ") + (words_.lazy ? "lazy" : "eager") + " decay, documents " + (ents_.lazy ? "lazy" : "eager") + " decay";
    out += " | CSR stream layout " + std::to_string(tune_.sort_layout >= 0 ? tune_.sort_layout : (dt_main ? 2 : 4));
    char buf[2048];
    const char* sw = tuning_describe(tune_, buf, sizeof(buf));
    out += std::string(" | switches: ") + (sw[0] ? sw : "defaults");
    return out;
}

// the forward product of a batch of B windows forms the phrase rows itself (compute_cost): the kernel launch_gemm would take
// covers the shape with the gather inside
bool Model::gather_fused_at(int64_t B) const {
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size, w = cfg_.window_size;
    if (!gemm_split_products() || B < 512 || !split_fwd_.rplanes) return false;
    if (B <= gemm_rows_max_m())
        return (tune_.gather_fuse & 1) && gemm_rsplit_gather_covers(static_cast<int>(B), de, dw, cfg_.batch_normalization != 0, w);
    return false;
}

// The projection update may add up the dT product's slabs itself (TransformUpdateArgs::partial) where its sum is launch_splitk_reduce's
// sum: the update kernel reproduces the VECTORISED reduce's order (16 interleaved groups, then the groups), which that launcher only
// takes for d_e * d_w % 4 == 0 and 16 B aligned buffers — otherwise it adds the slabs one after the other, and the fused step would
// differ from compute_cost; compute_gradients; update in the last bits.
bool Model::slab_sum_fusable() const {
    const size_t nT = static_cast<size_t>(cfg_.entity_repr_size) * cfg_.word_repr_size;
    return tune_.slab_sum_in_update && cfg_.world_size <= 1 && nT % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(gT_partial_.p) | reinterpret_cast<uintptr_t>(gT_.p)) % 16 == 0;
}

// which side streams build the two tables' CSRs this step (compute_cost: NVSM_SORT_LAYOUT)
int Model::csr_stream_layout() const {
    const int sort_layout_env = tune_.sort_layout;
    return sort_layout_env >= 0 ? sort_layout_env : (dt_on_main() ? 2 : 4);
}

// This step's dT product runs on the split-bf16 split-K kernel (gemm_dt.hip): large batches of the shapes it covers, unless the
// exact-fp32 kernels are asked for. Not per-rank batches: there the product rides on side stream 2 next to the two table passes,
// and a kernel whose workgroups each need a whole CU's registers waits for CUs to fall empty — 152 us in-step at batch 6 400
// for 18 us alone, exactly as long as the 128 x 128-tiled fp32 kernel takes there, whose small workgroups slip in between the
// passes' and leave the step shorter: batch 6 400 / 12 800 / 25 600: 0.291 / 0.403 / 0.584 ms tiled against 0.312 / 0.426 /
// 0.610 (0.340 / 0.438 / 0.593 with this kernel on the main stream). Interleaved A/B, tools/ab_shapes.sh.
bool Model::use_dt() const { return use_dt_at(B_); }

// the fused step's dT product on the main stream (see step()): large batches of eager tables
bool Model::dt_on_main() const { return dt_on_main_at(B_); }
// (round 5, after the table passes went to five waves per SIMD: with eagerly decayed tables and one rank the main-stream placement
//  — below — pays from 16 384 windows: 16 384: 0.436 -> 0.428 ms, 20 480: 0.492 -> 0.481, 25 600: 0.564 -> 0.547, 32 768: 0.659 ->
//  0.640, 40 000: 0.753 -> 0.731; 12 800: 0.381 -> 0.422 the other way. Lazily decayed tables and data-parallel ranks keep dt_min_batch.)
bool Model::use_dt_at(int64_t B) const {
    const bool early = B >= kDtMainMinBatch && !words_.lazy && !ents_.lazy && cfg_.world_size <= 1 && tune_.dt_on_main != 0;
    return dt_ok_ && gemm_split_products() != 0 && (B >= tune_.dt_min_batch || early);
}
// ... or on the wave-sized kernel (gemm_dtw.hip): the batches gemm_dt does not take, up to NVSM_DTW_MAX_B
bool Model::use_dtw_at(int64_t B) const {
    return !use_dt_at(B) && gemm_split_products() != 0 && tune_.dtw_max_batch > 0 && B >= 64 && B <= tune_.dtw_max_batch &&
           gemm_dtw_covers(cfg_.word_repr_size, cfg_.entity_repr_size, static_cast<int>(B));
}
bool Model::dt_on_main_at(int64_t B) const {      // (one rule for step() and describe())
    const int dt_main_env = tune_.dt_on_main;
    return (dt_main_env >= 0 ? dt_main_env != 0 : (B >= kDtMainMinBatch && !words_.lazy && !ents_.lazy)) && use_dt_at(B) && cfg_.world_size <= 1;
}

// T changed: its bf16 planes for the next two projection products, behind the writer on the writer's stream (off the critical
// path in the fused step: side stream 2, which the next projection GEMM joins anyway)
void Model::planes_stale() { split_fwd_.ready = split_bwd_.ready = split_fwd_.rready = split_bwd_.rready = false; }

void Model::cut_transform_planes(hipStream_t strm) {
    planes_stale();
    if (!gemm_split_products()) return;                                   // (the exact-fp32 kernels are in use)
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size;
    if (B_ <= gemm_rows_max_m()) {
        // per-rank batch sizes: the planes in the row-panel kernel's fragment order (gemm_rsplit.hip), where it covers the shapes
        const int B = static_cast<int>(B_);
        if (B >= 512 && gemm_rsplit_covers(0, B, de, dw, cfg_.batch_normalization != 0, false, false)) {
            launch_gemm_rsplit_planes(0, T_.p, de, dw, de, split_fwd_.rplanes, strm);
            split_fwd_.rready = true;
        }
        if (B >= 512 && gemm_rsplit_covers(1, B, dw, de, false, false, false)) {
            launch_gemm_rsplit_planes(1, T_.p, dw, de, de, split_bwd_.rplanes, strm);
            split_bwd_.rready = true;
        }
        return;
    }
    launch_gemm_split_planes(0, T_.p, de, dw, de, split_fwd_.planes, strm);      // forward: B = T as [K = dw][N = de]
    launch_gemm_split_planes(1, T_.p, dw, de, de, split_bwd_.planes, strm);      // backward: B stored [N = dw][K = de]
    split_fwd_.ready = split_bwd_.ready = true;
}

void Model::update(float lr, float scaled_lambda) {
    if (!have_grads_) throw Error(NVSM_ERR_STATE, "update requires compute_gradients");
    if (lr < 0.f || scaled_lambda < 0.f) throw Error(NVSM_ERR_INVALID_ARGUMENT, "learning_rate and lambda must be >= 0");   // storage.cu:62-63
    NVSM_HIP_CHECK(hipSetDevice(cfg_.device));
    RangeScope range_up("UpdateParameters");            // cpp/main.cu:429
    NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_csr_ents_, 0));     // join the side-stream CSR builds
    NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_csr_, 0));
    csr_joined_ents_ = csr_joined_words_ = true;
    gather_update_inputs();
    update_entities(lr, scaled_lambda, stream_);
    update_words(lr, scaled_lambda);
    update_transform(lr, scaled_lambda, stream_);
    NVSM_HIP_CHECK(hipGetLastError());
    have_grads_ = false;      // gradients are consumed (the reference's optimisers overwrite them too)
}

// One loop body of iterate_data (cpp/main.cu:400-444). With everything known up front the independent chains of the
// backward half run concurrently instead of back to back:
//   main stream   : batch-norm backward → dx GEMM (MFMA bound) → words update (L2 / HBM bound)
//   side stream 1 : [documents CSR build] → documents update (needs only the loss kernel's outputs; HBM bound)
//   side stream 2 : [words CSR build] → dT GEMM + reduce (MFMA bound) → (all-reduce) → projection update (after the dx
//                   GEMM has read T)
// and the two side-stream tails are joined by the NEXT compute_cost where it needs T and E (join_T / join_E), not here.
// Results are identical to compute_cost; compute_gradients; update — only the interleaving differs.
void Model::step(const nvsm_batch& batch, const int64_t* entity_ids, float lr, float* cost) {
    if (exact_) {
        // exact data-parallel tables: the table passes need every rank's loss-kernel and backward results, gathered on the main
        // stream — nothing to run next to anything: the three calls in a row
        compute_cost(batch, entity_ids);
        const float sl_exact = scaled_regularization_lambda();
        if (lr < 0.f || sl_exact < 0.f) throw Error(NVSM_ERR_INVALID_ARGUMENT, "learning_rate and lambda must be >= 0");
        compute_gradients();
        update(lr, sl_exact);
        if (cost) *cost = get_cost();
        return;
    }
    const bool fewer_events = tune_.fewer_events;
    const int docs_after_dx_env = tune_.docs_after_dx;
    const bool docs_after_dx = docs_after_dx_env >= 0 ? docs_after_dx_env != 0 : batch.num_instances >= 16384;
    const bool loss_event = !(docs_after_dx && fewer_events);      // (see below)
    loss_stop_event_ = loss_event ? ev_loss_ : nullptr;
    {
        // lr and λ are known before the forward pass: parts of the update that depend on nothing else can be queued early (compute_cost)
        const double Bg = static_cast<double>(batch.num_instances > 0 ? batch.num_instances : 1) * (cfg_.world_size > 1 ? cfg_.world_size : 1);
        hoist_lr_ = lr; hoist_sl_ = cfg_.regularization_lambda / static_cast<float>(Bg);      // = scaled_regularization_lambda() behind compute_cost
        hoist_untouched_ = tune_.hoist_untouched != 0 && lr >= 0.f && hoist_sl_ > 0.f;
    }
    try { compute_cost(batch, entity_ids); } catch (...) { loss_stop_event_ = nullptr; hoist_untouched_ = false; words_untouched_hoisted_ = false; throw; }
    loss_stop_event_ = nullptr;
    hoist_untouched_ = false;
    // The caller wants this step's loss: the loss word is final behind the loss kernel, 0.3 ms into a 0.9 ms step. A copy on side
    // stream 3 behind an event recorded here lets the host read it while the backward pass and the updates still run, and
    // queue the next step in the meantime (waiting for the whole step instead — get_cost() — left the GPU idle while the host
    // queued: 5 % of the NVSM step). Not under data parallelism, where the word is summed over the ranks in the backward pass.
    const bool early_cost = cost && cfg_.world_size <= 1 && aux3_stream_ && cost_host_;
    if (early_cost) {
        NVSM_HIP_CHECK(hipEventRecord(ev_cost_ready_, stream_));
        NVSM_HIP_CHECK(hipStreamWaitEvent(aux3_stream_, ev_cost_ready_, 0));
        NVSM_HIP_CHECK(hipMemcpyAsync(cost_host_, stats_bwd_, sizeof(double), hipMemcpyDeviceToHost, aux3_stream_));
        NVSM_HIP_CHECK(hipEventRecord(ev_cost_copied_, aux3_stream_));
    }
    const float sl = scaled_regularization_lambda();
    if (lr < 0.f || sl < 0.f) throw Error(NVSM_ERR_INVALID_ARGUMENT, "learning_rate and lambda must be >= 0");
    // Data parallel: the dT GEMM, the all-reduce of the projection gradient and the projection update ride on side stream 2
    // exactly as the single-GPU tail does. The three collectives of a step stay totally ordered on every rank by the
    // events below (BN forward sums → BN backward sums on the main stream → [ev_dx] gradient on side stream 2 →
    // [join_T] next step's BN forward sums), so one communicator serves both streams. NVSM_DP_T_ON_MAIN=1 keeps all
    // collectives on the main stream (dT GEMM + all-reduce + update no longer overlap the words update).
    const bool dp = cfg_.world_size > 1 && dp_single_stream_;
    RangeScope range_bu("ComputeGradients+UpdateParameters");      // cpp/main.cu:414,429 — one interleaved region here
    // ... but only once the dx GEMM is through at large batches: next to the MFMA-bound GEMM the row pass (100 k short-lived
    // waves) keeps the GEMM's workgroups from becoming resident — measured at B = 51 200: dx GEMM 132 → 203 us, dT GEMM
    // 195 → 402 us, step 1.099 → 1.126 ms. (NVSM_DOCS_AFTER_DX=0/1 overrides.)
    // side stream 1 (behind the documents CSR build): the documents update, HBM-bound — free to run on next to the next
    // step's projection GEMM; the next loss kernel joins it. It needs the loss kernel's outputs; when it is held behind the
    // dx GEMM anyway, that GEMM's event stands for the loss kernel's too (an event recorded between two kernels of the main
    // stream costs the stream a bubble of several microseconds).
    if (loss_event) NVSM_HIP_CHECK(hipStreamWaitEvent(aux_stream_, ev_loss_, 0));      // (recorded with the loss kernel)
    // NVSM_DOCS_ON_MAIN (experiments): 1 = the documents update on the main stream in front of the words update, 2 = behind
    // it (two HBM-bound passes one after the other instead of next to each other)
    const int docs_on_main = tune_.docs_on_main;
    dx_follower_ = dp ? nullptr : aux2_stream_;      // side stream 2 runs the dT GEMM as soon as dx is final
    if (docs_after_dx || docs_on_main) backward_dx();
    if (docs_on_main == 1) {
        NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_csr_ents_, 0));
        update_entities(lr, sl, stream_, nullptr);
    } else if (docs_on_main == 0) {
        if (!loss_event) NVSM_HIP_CHECK(hipStreamWaitEvent(aux_stream_, ev_bwdx_, 0));
        if (tune_.docs_delay_us > 0) launch_delay(tune_.docs_delay_us, aux_stream_);      // (experiments)
        update_entities(lr, sl, aux_stream_, (docs_after_dx && loss_event) ? ev_bwdx_ : nullptr);
        NVSM_HIP_CHECK(hipEventRecord(ev_E_done_, aux_stream_));
        E_pending_ = true;
    }
    if (!docs_after_dx && !docs_on_main) backward_dx();
    dx_follower_ = nullptr;
    if (dp) {
        backward_T(stream_);
    } else {
        // side stream 2 (behind the words CSR build): the MFMA-bound dT GEMM next to the HBM-bound words update, then the
        // projection update; the next projection GEMM joins it
        // (started as soon as dx is final rather than after the dx GEMM: 1.084 vs 1.100 ms per step, interleaved A/B)
        // (the wait for ev_dx_ was issued by backward_dx, right behind the kernel that carries the event)
        // Large batches of the eager tables: the dT product (split-bf16, gemm_dt.hip: workgroups of a whole CU's LDS and most
        // of its registers) on the MAIN stream in front of the words update. On side stream 2 such workgroups trickle in
        // behind the thousands of small workgroups of the two table passes (0.55 ms next to them, 0.05 ms alone) and the
        // projection update behind them came too late for the next step; the 128 x 128-tiled fp32 kernel fits into the gaps
        // but costs the passes 70 us of the step (a timing run without the product). In front of the passes it has the chip
        // for 45-70 us while the documents pass starts up, and the projection update still runs on side stream 2.
        // Interleaved A/B: NVSM shape 0.931 -> 0.921 ms, full_adam 0.850 -> 0.787; batch 25 600 0.588 -> 0.597 and
        // |D| = 2 M 1.71 -> 1.72 the other way (the main stream is their longer chain): hence the rule. NVSM_DT_ON_MAIN=0 / 1.
        if (dt_on_main()) {
            // (the slab sum leaves the main stream too: the projection update on side stream 2 adds the slabs up)
            fuse_slab_sum_ = slab_sum_fusable();
            try { backward_T(stream_); } catch (...) { fuse_slab_sum_ = false; throw; }
            fuse_slab_sum_ = false;
            NVSM_HIP_CHECK(hipEventRecord(ev_gathered_, stream_));
            NVSM_HIP_CHECK(hipStreamWaitEvent(aux2_stream_, ev_gathered_, 0));
        } else {
            // (the slabs of the split-K product are added up by the projection update behind it on the same stream)
            fuse_slab_sum_ = slab_sum_fusable();      // (data parallel: the summed gradient is all-reduced first)
            try { backward_T(aux2_stream_); } catch (...) { fuse_slab_sum_ = false; throw; }
            fuse_slab_sum_ = false;
        }
        NVSM_HIP_CHECK(hipStreamWaitEvent(aux2_stream_, ev_bwdx_, 0));      // the dx GEMM is the last reader of T
        update_transform(lr, sl, aux2_stream_);
        NVSM_HIP_CHECK(hipEventRecord(ev_T_done_, aux2_stream_));
        T_pending_ = true;
    }
    // The words update follows the words CSR build only. The documents build is not waited for on this stream at all: its
    // ids live in the buffer this step's prologue wrote, the next prologue writes the other one, and the step after that is
    // behind the next loss kernel, which joins this step's documents update, which followed its CSR build. (One event for
    // both builds, made on side stream 2 behind a wait for the documents build, was the previous form: two cross-stream hops
    // of 10 us each, and at batch 4096 the documents build — not the backward GEMM — then decided when the words update
    // started: 24 us of idle main stream.)
    NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_csr_, 0));
    csr_joined_ents_ = csr_joined_words_ = true;
    // A words table split into rows with and without entries (tables larger than the batch, not lazily decayed): the
    // streaming decay of the rows WITHOUT entries does not belong on the critical stream between the passes over the rows
    // with entries — disjoint rows — but behind the projection update on side stream 2, which built this CSR and is idle by
    // then; the next step's word gather joins that stream (LSE batch 4096: 14-17 us of a 0.2 ms step). NVSM_UNTOUCHED_ASIDE=0: off.
    const bool untouched_aside = tune_.untouched_aside;
    // (only when side stream 2 is the stream that built the words CSR: the untouched pass reads its row bounds, and the next
    //  step's sort on that stream clears them — a pass queued on any other stream would have neither order)
    const bool words_aside = untouched_aside && !dp && !words_.lazy && words_csr_stream_ == aux2_stream_ && !words_untouched_hoisted_ &&
                             row_pass_split(csr_of(words_, B_ * cfg_.window_size));
    words_untouched_stream_ = words_aside ? aux2_stream_ : nullptr;
    update_words(lr, sl);
    words_untouched_stream_ = nullptr;
    if (words_aside) {
        NVSM_HIP_CHECK(hipEventRecord(ev_T_done_, aux2_stream_));      // (again: now it stands for the streaming decay too)
        T_pending_ = true;
        words_tail_pending_ = true;
    }
    if (docs_on_main == 2) {
        NVSM_HIP_CHECK(hipStreamWaitEvent(stream_, ev_csr_ents_, 0));
        update_entities(lr, sl, stream_, nullptr);
    }
    if (dp) update_transform(lr, sl, stream_);
    NVSM_HIP_CHECK(hipGetLastError());
    have_grads_ = false;
    if (early_cost) {
        NVSM_HIP_CHECK(hipEventSynchronize(ev_cost_copied_));
        raise_device_error();
        cost_ = -(*cost_host_ / static_cast<double>(B_));
        cost_valid_ = true;
        *cost = static_cast<float>(cost_);
    } else if (cost) {
        *cost = get_cost();
    }
}

int64_t Model::step_deferred(const nvsm_batch& batch, const int64_t* entity_ids, float lr) {
    step(batch, entity_ids, lr, nullptr);
    DeferredCost& d = deferred_[next_ticket_ % NVSM_MAX_DEFERRED];
    if (!d.host) {
        NVSM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&d.host), sizeof(double), hipHostMallocDefault));
        NVSM_HIP_CHECK(hipEventCreateWithFlags(&d.ev, hipEventDisableTiming));
    }
    // the loss word is final after the loss kernel (and, data parallel, after the all-reduce in the backward pass): both
    // are behind us on the main stream
    // (data parallel with the folded collective: the summed word lives in loss_red_, final on the stream that ran the dT all-reduce)
    hipStream_t ls = loss_folded_ ? loss_stream_ : stream_;
    NVSM_HIP_CHECK(hipMemcpyAsync(d.host, loss_folded_ ? loss_red_.p : stats_bwd_, sizeof(double), hipMemcpyDeviceToHost, ls));
    NVSM_HIP_CHECK(hipEventRecord(d.ev, ls));
    d.batch = static_cast<double>(B_) * (cfg_.world_size > 1 ? cfg_.world_size : 1);
    d.ticket = next_ticket_;
    return next_ticket_++;
}

float Model::deferred_cost(int64_t ticket) {
    if (ticket < 0 || ticket >= next_ticket_) throw Error(NVSM_ERR_INVALID_ARGUMENT, "unknown ticket");
    DeferredCost& d = deferred_[ticket % NVSM_MAX_DEFERRED];
    if (d.ticket != ticket) throw Error(NVSM_ERR_STATE, "ticket is older than the NVSM_MAX_DEFERRED most recent steps");
    NVSM_HIP_CHECK(hipEventSynchronize(d.ev));
    raise_device_error();
    return static_cast<float>(-(*d.host / d.batch));
}

void Model::wait_inputs() {
    // a host batch has been consumed once the copy stream is through with it; a device-resident one once the step's
    // prologue and gather have read it (ev_inputs_ sits behind the prologue; the feature weights are read later still,
    // but device-resident batches are the caller's to keep alive until the step has run)
    if (last_batch_on_host_) { if (copied_recorded_) NVSM_HIP_CHECK(hipEventSynchronize(ev_copied_)); }
    else if (inputs_recorded_) NVSM_HIP_CHECK(hipEventSynchronize(ev_inputs_));
}

// ---------------------------------------------------------------------------------------------
// parameter / tensor access
// ---------------------------------------------------------------------------------------------
Model::ParamRef Model::find_param(const std::string& name) {
    auto tab = [&](TableState& t, const std::string& sub) -> ParamRef {
        if (sub == "representations") return {t.P.p, static_cast<int64_t>(t.P.n)};
        if (sub == "m") return {t.m.p, static_cast<int64_t>(t.m.n)};
        if (sub == "v") {
            if (t.vfull.p) return {t.vfull.p, static_cast<int64_t>(t.vfull.n)};
            if (cfg_.update_method == NVSM_ADAM) return {t.sc[t.sc_cur].p, static_cast<int64_t>(t.sc[t.sc_cur].n)};
        }
        if (sub == "a" && cfg_.update_method == NVSM_ADAGRAD) return {t.sc[t.sc_cur].p, static_cast<int64_t>(t.sc[t.sc_cur].n)};
        return {nullptr, 0};
    };
    if (name == "word_representations-representations") return tab(words_, "representations");
    if (name == "entity_representations-representations") return tab(ents_, "representations");
    if (name == "word_entity_mapping-transform") return {T_.p, static_cast<int64_t>(T_.n)};
    if (name == "word_entity_mapping-bias") return {b_.p, static_cast<int64_t>(b_.n)};
    if (name.rfind("word_representations/", 0) == 0) return tab(words_, name.substr(21));
    if (name.rfind("entity_representations/", 0) == 0) return tab(ents_, name.substr(23));
    if (name == "word_entity_mapping/s0_transform") return {s0T_.p, static_cast<int64_t>(s0T_.n)};
    if (name == "word_entity_mapping/s0_bias") return {s0b_.p, static_cast<int64_t>(s0b_.n)};
    if (name == "word_entity_mapping/s1_transform") return {s1T_.p, static_cast<int64_t>(s1T_.n)};
    if (name == "word_entity_mapping/s1_bias") return {s1b_.p, static_cast<int64_t>(s1b_.n)};
    return {nullptr, 0};
}

int64_t Model::param_size(const std::string& name) {
    ParamRef r = find_param(name);
    if (!r.p) throw Error(NVSM_ERR_INVALID_ARGUMENT, "unknown parameter: " + name);
    return r.n;
}
void Model::get_param(const std::string& name, float* dst, int64_t count) {
    ParamRef r = find_param(name);
    if (!r.p) throw Error(NVSM_ERR_INVALID_ARGUMENT, "unknown parameter: " + name);
    if (count != r.n) throw Error(NVSM_ERR_INVALID_ARGUMENT, "size mismatch for " + name);
    synchronize();
    lazy_flush_all();
    NVSM_HIP_CHECK(hipMemcpy(dst, r.p, count * sizeof(float), hipMemcpyDeviceToHost));
}
void Model::set_param(const std::string& name, const float* src, int64_t count) {
    ParamRef r = find_param(name);
    if (!r.p) throw Error(NVSM_ERR_INVALID_ARGUMENT, "unknown parameter: " + name);
    if (count != r.n) throw Error(NVSM_ERR_INVALID_ARGUMENT, "size mismatch for " + name);
    synchronize();
    lazy_flush_all();
    NVSM_HIP_CHECK(hipMemcpy(r.p, src, count * sizeof(float), hipMemcpyHostToDevice));
    planes_stale();
}

void Model::increment_param(const std::string& name, int64_t index, float delta) {
    ParamRef r = find_param(name);
    if (!r.p) throw Error(NVSM_ERR_INVALID_ARGUMENT, "unknown parameter: " + name);
    if (index < 0 || index >= r.n) throw Error(NVSM_ERR_INVALID_ARGUMENT, "parameter index out of range for " + name);
    synchronize();
    lazy_flush_all();
    float v = 0.f;
    NVSM_HIP_CHECK(hipMemcpy(&v, r.p + index, sizeof(float), hipMemcpyDeviceToHost));
    v += delta;
    NVSM_HIP_CHECK(hipMemcpy(r.p + index, &v, sizeof(float), hipMemcpyHostToDevice));
    planes_stale();
}

int64_t Model::tensor_size(const std::string& name) {
    const int64_t B = B_, N = B_ * R_;
    const int dw = cfg_.word_repr_size, de = cfg_.entity_repr_size;
    if (name == "phrase" || name == "grad_phrase") return B * dw;
    if (name == "pre" || name == "proj" || name == "grad_proj") return B * de;
    if (name == "probs" || name == "multipliers" || name == "entity_ids") return N;
    if (name == "bn_mean" || name == "bn_inv_std" || name == "grad_bias") return de;
    if (name == "grad_transform") return static_cast<int64_t>(de) * dw;
    if (name == "grad_entity") return N * de;
    if (name == "arrival_counters") return 6;
    throw Error(NVSM_ERR_INVALID_ARGUMENT, "unknown tensor: " + name);
}

void Model::get_tensor(const std::string& name, float* dst, int64_t count) {
    if (!have_forward_) throw Error(NVSM_ERR_STATE, "no forward result");
    if (count != tensor_size(name)) throw Error(NVSM_ERR_INVALID_ARGUMENT, "size mismatch for " + name);
    const bool needs_grads = name.rfind("grad_", 0) == 0;
    if (needs_grads && !have_grads_) throw Error(NVSM_ERR_STATE, name + " requires compute_gradients (and no update since)");
    const float* src = nullptr;
    std::vector<float> tmp;
    if (name == "phrase") src = phrase_p_;
    else if (name == "pre") src = pre_.p;
    else if (name == "proj") src = proj_.p;
    else if (name == "probs") src = probs_.p;
    else if (name == "multipliers") src = coef_.p;
    else if (name == "bn_mean") src = bn_mean_.p;
    else if (name == "bn_inv_std") src = bn_inv_std_.p;
    else if (name == "grad_phrase") src = gphrase_.p;
    else if (name == "grad_proj") src = dy_.p;
    else if (name == "grad_bias") src = gb_.p;
    else if (name == "grad_transform") src = gT_.p;
    else if (name == "grad_entity") {
        if (grad_entity_.n < static_cast<size_t>(count)) grad_entity_.alloc(count);
        if (cfg_.l2_normalize_entity_reprs)
            launch_materialize_grad_entity_l2(coef_.p, proj_.p, ents_.P.p, ids_p_, B_ * R_, R_, cfg_.entity_repr_size, grad_entity_.p, nullptr, stream_);
        else
            launch_materialize_grad_entity(coef_.p, proj_.p, B_ * R_, R_, cfg_.entity_repr_size, grad_entity_.p, stream_);
        src = grad_entity_.p;
    } else if (name == "arrival_counters") {
        // How many arrival counters of the last-arriver hand-overs (device_utils.h grid_sum_ordered, update.hip table passes) are NOT
        // zero with every stream of the handle quiet: [forward sums, loss sums, words rows, words level 2, documents rows, documents
        // level 2]. Every launch that runs to completion returns its counters to zero; a counter left behind would make a later sum
        // hand over early (tests/test_gpu_soak.py).
        synchronize();
        int k = 0;
        for (DevBuf<int>* b : {&sums_fwd_.arrive, &sums_bwd_.arrive, &words_.arrive_row, &words_.arrive2, &ents_.arrive_row, &ents_.arrive2}) {
            std::vector<int> h(b->n);
            if (b->n) NVSM_HIP_CHECK(hipMemcpy(h.data(), b->p, b->n * sizeof(int), hipMemcpyDeviceToHost));
            int64_t nz = 0;
            for (int v : h) nz += v != 0;
            dst[k++] = static_cast<float>(nz);
        }
        return;
    } else if (name == "entity_ids") {
        std::vector<int> h(count);
        synchronize();
        NVSM_HIP_CHECK(hipMemcpy(h.data(), ids_p_, count * sizeof(int), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < count; ++i) dst[i] = static_cast<float>(h[i]);
        return;
    }
    synchronize();
    NVSM_HIP_CHECK(hipMemcpy(dst, src, count * sizeof(float), hipMemcpyDeviceToHost));
}

}  // namespace cunvsm
