// extern "C" surface of libcunvsm_amd.so — see include/cunvsm_amd.h. No C++ exception and no abort
// crosses this boundary: every entry point returns an nvsm_status and records nvsm_last_error().
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fcntl.h>
#include <limits.h>
#include <sched.h>
#include <unistd.h>
#include <vector>
#include <string>

#include "model.h"

namespace cunvsm {
void rccl_unique_id(char id[128]);
void rccl_selftest(int device);
void rccl_latency(int device, int de, int dw, int repeats, float us[3], int64_t bytes[3]);
void range_push(const char* name);
void range_pop();
}

using cunvsm::Error;
using cunvsm::Model;

static thread_local std::string g_last_error;

struct nvsm_model {
    Model impl;
    explicit nvsm_model(const nvsm_config& c) : impl(c) {}
};

template <typename Fn>
static int guarded(Fn&& fn) {
    try {
        fn();
        return NVSM_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.status;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return NVSM_ERR_DEVICE;
    } catch (...) {
        g_last_error = "unknown error";
        return NVSM_ERR_DEVICE;
    }
}

// a call on a handle runs under the handle's switches (tuning.h): read once, by nvsm_create
template <typename Fn>
static int guarded_on(nvsm_model* m, Fn&& fn) {
    cunvsm::TuningScope scope(&m->impl.tune());
    return guarded(fn);
}
// a debug hook (no handle) reads the environment itself, per call: tests switch variables between calls
template <typename Fn>
static int guarded_hook(Fn&& fn) {
    const cunvsm::Tuning t = cunvsm::Tuning::from_env();
    cunvsm::TuningScope scope(&t);
    return guarded(fn);
}

#define NVSM_REQUIRE(ptr)                                          \
    if (!(ptr)) {                                                  \
        g_last_error = "null argument: " #ptr;                     \
        return NVSM_ERR_INVALID_ARGUMENT;                          \
    }

extern "C" {

const char* nvsm_last_error(void) { return g_last_error.c_str(); }
const char* nvsm_version(void) { return "cunvsm_amd 0.1 (gfx950)"; }

int nvsm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// "0-63,128-191" -> CPU set (sysfs cpulist format)
static bool parse_cpulist(const std::string& text, cpu_set_t* set) {
    CPU_ZERO(set);
    bool any = false;
    size_t at = 0;
    while (at < text.size()) {
        size_t end = text.find(',', at);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(at, end - at);
        at = end + 1;
        if (item.empty() || item[0] < '0' || item[0] > '9') continue;
        char* rest = nullptr;
        const long lo = std::strtol(item.c_str(), &rest, 10);
        long hi = lo;
        if (rest && *rest == '-') hi = std::strtol(rest + 1, nullptr, 10);
        for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) { if (c >= 0) { CPU_SET(static_cast<int>(c), set); any = true; } }
    }
    return any;
}
static std::string read_first_line(const std::string& path) {
    std::string out;
    if (FILE* f = std::fopen(path.c_str(), "r")) {
        char buf[4096];
        if (std::fgets(buf, sizeof(buf), f)) out = buf;
        std::fclose(f);
    }
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
    return out;
}

// The PCI device directory (/sys/bus/pci/devices/<id>/) of HIP device `device`, found WITHOUT touching the HIP runtime: the
// render nodes of AMD PCI devices this process may open (a container's device cgroup admits only its own GPUs), in ascending
// minor order, the device-th of them — after HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are plain lists of ordinals
// that select among MORE than the admitted nodes, which is how bare-metal hosts use them. A guess, good on every box seen;
// nvsm_create re-checks it against hipDeviceGetPCIBusId once the runtime is up (rebind_if_guessed_wrong).
static std::string sysfs_device_dir_without_hip(int device) {
    std::vector<std::pair<int, std::string>> nodes;      // (render minor, pci dir)
    if (DIR* d = opendir("/sys/class/drm")) {
        while (dirent* e = readdir(d)) {
            int minor = -1;
            if (std::sscanf(e->d_name, "renderD%d", &minor) != 1) continue;
            char real[PATH_MAX];
            const std::string link = std::string("/sys/class/drm/") + e->d_name + "/device";
            if (!realpath(link.c_str(), real)) continue;
            const std::string dir = std::string(real) + "/";
            if (dir.find("/pci") == std::string::npos || read_first_line(dir + "vendor") != "0x1002") continue;      // (partitions of a GPU are platform devices)
            const int fd = open((std::string("/dev/dri/") + e->d_name).c_str(), O_RDWR | O_CLOEXEC);
            if (fd < 0) continue;
            close(fd);
            nodes.emplace_back(minor, dir);
        }
        closedir(d);
    }
    std::sort(nodes.begin(), nodes.end());
    int index = device;
    for (const char* name : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"}) {
        const char* v = cunvsm::env_raw(name);
        if (!v || !v[0]) continue;
        std::vector<int> list;
        bool plain = true;
        for (const char* p = v; *p && plain;) {
            char* end = nullptr;
            const long x = std::strtol(p, &end, 10);
            if (end == p || x < 0) { plain = false; break; }
            list.push_back(static_cast<int>(x));
            p = (*end == ',') ? end + 1 : end;
            if (*end && *end != ',') plain = false;
        }
        if (!plain) return std::string();                                   // (UUIDs: leave it to the runtime)
        const int most = list.empty() ? -1 : *std::max_element(list.begin(), list.end());
        if (most < static_cast<int>(nodes.size()) && static_cast<int>(list.size()) < static_cast<int>(nodes.size())) {
            if (index < 0 || index >= static_cast<int>(list.size())) return std::string();
            index = list[static_cast<size_t>(index)];
        }
        break;
    }
    if (index < 0 || index >= static_cast<int>(nodes.size())) return std::string();
    return nodes[static_cast<size_t>(index)].second;
}

static thread_local bool tl_bound_by_guess = false;      // this thread's mask was set from the sysfs guess ...
static thread_local cpu_set_t tl_mask_before;            // ... and this is what it was before
static thread_local std::string tl_guessed_dir;

// mask := the device's local CPUs ∩ `within`; false: left alone (no topology / empty intersection)
static bool bind_to_device_dir(const std::string& dir, const cpu_set_t& within, int* numa_node) {
    const std::string node = read_first_line(dir + "numa_node");
    if (numa_node && !node.empty()) *numa_node = std::atoi(node.c_str());
    cpu_set_t local, both;
    if (!parse_cpulist(read_first_line(dir + "local_cpulist"), &local)) return false;
    CPU_AND(&both, &local, &within);
    if (CPU_COUNT(&both) == 0) return false;                                // (the caller's mask excludes the node: theirs wins)
    return sched_setaffinity(0, sizeof(both), &both) == 0;
}
static std::string hip_device_dir(int device) {
    char bus[64] = {0};
    NVSM_HIP_CHECK(hipDeviceGetPCIBusId(bus, static_cast<int>(sizeof(bus)), device));
    std::string id(bus);
    for (char& ch : id) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
    return "/sys/bus/pci/devices/" + id + "/";
}
// nvsm_create: the runtime is up now — was the guess right?
static void rebind_if_guessed_wrong(int device) noexcept {
    if (!tl_bound_by_guess) return;
    tl_bound_by_guess = false;
    try {      // (best effort: a handle that has just been made is not given up over the thread's affinity)
        char a[PATH_MAX], b[PATH_MAX];
        const std::string exact = hip_device_dir(device);
        if (realpath(exact.c_str(), a) && realpath(tl_guessed_dir.c_str(), b) && std::string(a) == std::string(b)) return;
        (void)bind_to_device_dir(exact, tl_mask_before, nullptr);
    } catch (...) {
        (void)hipGetLastError();
    }
}

int nvsm_bind_host_thread(int device, int* numa_node) {
    if (numa_node) *numa_node = -1;
    return guarded([&] {
        if (device < 0) throw Error(NVSM_ERR_INVALID_ARGUMENT, "device ordinal out of range");
        cpu_set_t now;
        const bool have_mask = sched_getaffinity(0, sizeof(now), &now) == 0;
        // Before the HIP runtime exists in this process, if the device can be found without it: the runtime's own threads and
        // host allocations are then made on the device's node too (LSE batch 4 096, GPU on node 0: 0.1495 ms in every process
        // bound from the start against 0.150-0.162 bound after initialisation, 0.158-0.18 unbound: tools/numa_exp.sh)
        const std::string guess = sysfs_device_dir_without_hip(device);
        if (!guess.empty()) {
            const std::string node = read_first_line(guess + "numa_node");
            if (numa_node && !node.empty()) *numa_node = std::atoi(node.c_str());
            if (!cunvsm::env_bind_host() || !have_mask) return;
            if (bind_to_device_dir(guess, now, nullptr)) { tl_bound_by_guess = true; tl_mask_before = now; tl_guessed_dir = guess; }
            return;
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error(NVSM_ERR_NO_DEVICE, "no HIP device visible");
        if (device >= ndev) throw Error(NVSM_ERR_INVALID_ARGUMENT, "device ordinal out of range");
        const std::string dir = hip_device_dir(device);
        const std::string node = read_first_line(dir + "numa_node");
        if (numa_node && !node.empty()) *numa_node = std::atoi(node.c_str());
        if (!cunvsm::env_bind_host() || !have_mask) return;
        (void)bind_to_device_dir(dir, now, nullptr);
    });
}

// Defaults of the reference CLI: cpp/main.cu:15-76 + the NVSM recipe of scripts/functions.sh:266,380-399.
void nvsm_config_default(nvsm_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->word_repr_size = 300; c->entity_repr_size = 256;
    c->batch_normalization = 1; c->nonlinearity = NVSM_HARD_TANH;
    c->clip_sigmoid = 1; c->bias_negative_samples = 0;
    c->window_size = 10; c->num_random_entities = 10;
    c->regularization_lambda = 1e-2f;
    c->update_method = NVSM_ADAM; c->adam_mode = NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE;
    c->max_batch_size = 51200;
    c->device = 0; c->sampler = NVSM_SAMPLER_HOST_MINSTD;
    c->world_size = 1; c->rank = 0; c->sync_batch_norm = 1;
}

int nvsm_create(const nvsm_config* cfg, nvsm_model** out) {
    NVSM_REQUIRE(cfg); NVSM_REQUIRE(out);
    *out = nullptr;
    return guarded([&] { *out = new nvsm_model(*cfg); rebind_if_guessed_wrong(cfg->device); });
}

void nvsm_destroy(nvsm_model* m) {
    try { delete m; } catch (...) {}
}

int nvsm_initialize(nvsm_model* m, uint64_t seed) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.initialize(seed); }); }
int nvsm_initialize_from_rng_state(nvsm_model* m) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.initialize_from_rng_state(); }); }
int nvsm_host_alloc(size_t bytes, void** out) {
    NVSM_REQUIRE(out);
    *out = nullptr;
    return guarded([&] { NVSM_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault)); });
}
int nvsm_host_free(void* p) { return guarded([&] { if (p) NVSM_HIP_CHECK(hipHostFree(p)); }); }
int nvsm_rng_get_state(nvsm_model* m, uint64_t* state) { NVSM_REQUIRE(m); NVSM_REQUIRE(state); return guarded_on(m, [&] { *state = m->impl.rng_get_state(); }); }
int nvsm_rng_set_state(nvsm_model* m, uint64_t state) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.rng_set_state(state); }); }

int nvsm_param_size(nvsm_model* m, const char* name, int64_t* count) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(name); NVSM_REQUIRE(count);
    return guarded_on(m, [&] { *count = m->impl.param_size(name); });
}
int nvsm_get_param(nvsm_model* m, const char* name, float* dst, int64_t count) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(name); NVSM_REQUIRE(dst);
    return guarded_on(m, [&] { m->impl.get_param(name, dst, count); });
}
int nvsm_set_param(nvsm_model* m, const char* name, const float* src, int64_t count) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(name); NVSM_REQUIRE(src);
    return guarded_on(m, [&] { m->impl.set_param(name, src, count); });
}

int nvsm_increment_parameter(nvsm_model* m, const char* name, int64_t index, float delta) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(name);
    return guarded_on(m, [&] { m->impl.increment_param(name, index, delta); });
}

int nvsm_compute_cost(nvsm_model* m, const nvsm_batch* batch, const int64_t* entity_ids) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(batch);
    return guarded_on(m, [&] { m->impl.compute_cost(*batch, entity_ids); });
}
int nvsm_compute_gradients(nvsm_model* m) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.compute_gradients(); }); }
int nvsm_update(nvsm_model* m, float lr, float scaled_lambda) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.update(lr, scaled_lambda); }); }
int nvsm_get_cost(nvsm_model* m, float* cost) { NVSM_REQUIRE(m); NVSM_REQUIRE(cost); return guarded_on(m, [&] { *cost = m->impl.get_cost(); }); }
int nvsm_get_cost_f64(nvsm_model* m, double* cost) { NVSM_REQUIRE(m); NVSM_REQUIRE(cost); return guarded_on(m, [&] { (void)m->impl.get_cost(); *cost = m->impl.cost_f64(); }); }
float nvsm_scaled_regularization_lambda(nvsm_model* m) { return m ? m->impl.scaled_regularization_lambda() : 0.f; }
int nvsm_step(nvsm_model* m, const nvsm_batch* batch, const int64_t* entity_ids, float lr, float* cost) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(batch);
    return guarded_on(m, [&] { m->impl.step(*batch, entity_ids, lr, cost); });
}

int nvsm_step_deferred(nvsm_model* m, const nvsm_batch* batch, const int64_t* entity_ids, float lr, int64_t* ticket) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(batch); NVSM_REQUIRE(ticket);
    return guarded_on(m, [&] { *ticket = m->impl.step_deferred(*batch, entity_ids, lr); });
}
int nvsm_deferred_cost(nvsm_model* m, int64_t ticket, float* cost) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(cost);
    return guarded_on(m, [&] { *cost = m->impl.deferred_cost(ticket); });
}
int nvsm_wait_inputs(nvsm_model* m) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.wait_inputs(); }); }

int nvsm_tensor_size(nvsm_model* m, const char* name, int64_t* count) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(name); NVSM_REQUIRE(count);
    return guarded_on(m, [&] { *count = m->impl.tensor_size(name); });
}
int nvsm_get_tensor(nvsm_model* m, const char* name, float* dst, int64_t count) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(name); NVSM_REQUIRE(dst);
    return guarded_on(m, [&] { m->impl.get_tensor(name, dst, count); });
}

int nvsm_set_stream(nvsm_model* m, void* s) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.set_stream(static_cast<hipStream_t>(s)); }); }
int nvsm_describe(nvsm_model* m, int64_t batch, char* buf, int64_t buf_bytes) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(buf);
    return guarded_on(m, [&] {
        const std::string d = m->impl.describe(batch);
        if (static_cast<int64_t>(d.size()) + 1 > buf_bytes) throw Error(NVSM_ERR_INVALID_ARGUMENT, "buffer too small");
        std::memcpy(buf, d.c_str(), d.size() + 1);
    });
}
int nvsm_synchronize(nvsm_model* m) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.synchronize(); }); }

int nvsm_comm_unique_id(char id[128]) { NVSM_REQUIRE(id); return guarded([&] { cunvsm::rccl_unique_id(id); }); }
int nvsm_comm_init(nvsm_model* m, const char id[128]) { NVSM_REQUIRE(m); NVSM_REQUIRE(id); return guarded_on(m, [&] { m->impl.comm_init(id); }); }
int nvsm_comm_size(nvsm_model* m, int* ranks) { NVSM_REQUIRE(m); NVSM_REQUIRE(ranks); return guarded_on(m, [&] { *ranks = m->impl.comm_ranks(); }); }
int nvsm_dp_average_tables(nvsm_model* m) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.average_tables(); }); }
void nvsm_range_push(const char* name) { if (name) cunvsm::range_push(name); }
void nvsm_range_pop(void) { cunvsm::range_pop(); }
int nvsm_comm_selftest(int device) { return guarded_hook([&] { cunvsm::rccl_selftest(device); }); }
int nvsm_comm_latency(int device, int entity_dim, int word_dim, int repeats, float us[3], int64_t bytes[3]) {
    NVSM_REQUIRE(us); NVSM_REQUIRE(bytes);
    return guarded_hook([&] { cunvsm::rccl_latency(device, entity_dim, word_dim, repeats, us, bytes); });
}
int nvsm_set_allreduce_callback(nvsm_model* m, nvsm_allreduce_fn fn, void* user) {
    NVSM_REQUIRE(m);
    return guarded_on(m, [&] { m->impl.set_allreduce_callback(fn, user); });
}

int nvsm_debug_set_table_pass_form(int one_launch) { cunvsm::set_table_pass_one_launch(one_launch != 0); return NVSM_OK; }
int nvsm_debug_delay(nvsm_model* m, int microseconds) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.debug_delay(microseconds); }); }
int nvsm_profile_enable(nvsm_model* m, int enable) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.synchronize(); m->impl.prof.enabled = enable != 0; }); }
int nvsm_profile_select(nvsm_model* m, const char* kernel) {
    NVSM_REQUIRE(m);
    return guarded_on(m, [&] { m->impl.synchronize(); m->impl.prof.only = kernel ? kernel : ""; });
}
int nvsm_profile_reset(nvsm_model* m) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.synchronize(); m->impl.prof.reset(); }); }
int nvsm_profile_names(nvsm_model* m, char* buf, int64_t buf_bytes) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(buf);
    return guarded([&] {
        int64_t off = 0;
        for (const std::string& n : m->impl.prof.names()) {
            if (off + static_cast<int64_t>(n.size()) + 2 > buf_bytes) throw Error(NVSM_ERR_INVALID_ARGUMENT, "buffer too small");
            std::memcpy(buf + off, n.c_str(), n.size() + 1);
            off += n.size() + 1;
        }
        if (off + 1 > buf_bytes) throw Error(NVSM_ERR_INVALID_ARGUMENT, "buffer too small");
        buf[off] = '\0';
    });
}
int nvsm_profile_get(nvsm_model* m, const char* kernel, double* total_ms, int64_t* launches) {
    NVSM_REQUIRE(m); NVSM_REQUIRE(kernel); NVSM_REQUIRE(total_ms); NVSM_REQUIRE(launches);
    return guarded([&] {
        if (!m->impl.prof.get(kernel, total_ms, launches)) throw Error(NVSM_ERR_INVALID_ARGUMENT, std::string("no such kernel: ") + kernel);
    });
}

// ---- debug hooks (tests only) ----
int nvsm_debug_gemm(int variant, int M, int N, int K, const float* hostA, const float* hostB, float* hostC) {
    NVSM_REQUIRE(hostA); NVSM_REQUIRE(hostB); NVSM_REQUIRE(hostC);
    return guarded_hook([&] {
        const int al = (variant >> 1) & 1, bl = variant & 1;
        const bool exact = (variant >> 30) & 1;               // bit 30: the exact-fp32 tiled kernel even where the planes kernel covers the shape
        const int split = (variant & 0x3fffffff) >> 2;        // variant bits: [exact << 30 | split_k want << 2 | a_layout << 1 | b_layout]
        cunvsm::DevBuf<float> A, B, C, P;
        A.alloc(static_cast<size_t>(M) * K); B.alloc(static_cast<size_t>(K) * N); C.alloc(static_cast<size_t>(M) * N);
        NVSM_HIP_CHECK(hipMemcpy(A.p, hostA, A.n * sizeof(float), hipMemcpyHostToDevice));
        NVSM_HIP_CHECK(hipMemcpy(B.p, hostB, B.n * sizeof(float), hipMemcpyHostToDevice));
        const int lda = al ? M : K, ldb = bl ? K : N;
        if (split >= 1 && al == 1 && bl == 0 && !exact && cunvsm::gemm_dt_covers(M, N, K)) {
            // the projection-gradient shape: the split-bf16 split-K kernel the model uses for it (gemm_dt.hip)
            const int slabs = cunvsm::gemm_dt_slabs(K, split);
            P.alloc(static_cast<size_t>(slabs) * M * N);
            if (!cunvsm::launch_gemm_dt(A.p, B.p, P.p, M, N, K, lda, ldb, split, nullptr)) throw Error(NVSM_ERR_UNSUPPORTED, "gemm_dt refused a covered shape");
            cunvsm::launch_splitk_reduce(P.p, slabs, static_cast<size_t>(M) * N, C.p, static_cast<int64_t>(M) * N, nullptr);
            NVSM_HIP_CHECK(hipDeviceSynchronize());
        } else if (split > 1) {
            const int slabs = cunvsm::gemm_split_k_slabs(K, split);
            P.alloc(static_cast<size_t>(slabs) * M * N);
            cunvsm::launch_gemm(al, bl, A.p, B.p, P.p, M, N, K, lda, ldb, N, 1.f, nullptr, split, static_cast<size_t>(M) * N, nullptr);
            cunvsm::launch_splitk_reduce(P.p, slabs, static_cast<size_t>(M) * N, C.p, static_cast<int64_t>(M) * N, nullptr);
        } else {
            cunvsm::DevBuf<char> planes, rplanes;
            planes.alloc(cunvsm::gemm_split_planes_bytes(N, K));
            rplanes.alloc(cunvsm::gemm_rsplit_planes_bytes(N, K));
            cunvsm::GemmSplitWs sws{planes.p, planes.n, false, rplanes.p, rplanes.n, false};
            cunvsm::launch_gemm(al, bl, A.p, B.p, C.p, M, N, K, lda, ldb, N, 1.f, nullptr, 1, 0, nullptr, nullptr, nullptr, 0.f, nullptr, false,
                                nullptr, &sws);
            NVSM_HIP_CHECK(hipDeviceSynchronize());
        }
        NVSM_HIP_CHECK(hipDeviceSynchronize());
        NVSM_HIP_CHECK(hipMemcpy(hostC, C.p, C.n * sizeof(float), hipMemcpyDeviceToHost));
    });
}

// average milliseconds of one launch_gemm of the batch-sized products on device-resident operands (A [M][K], B per b_layout):
// extras bit 0 = ordered column statistics (the forward product), bit 1 = row sums of squares (the backward one)
int nvsm_debug_gemm_time(int b_layout, int M, int N, int K, int extras, int repeats, float* avg_ms) {
    NVSM_REQUIRE(avg_ms);
    return guarded_hook([&] {
        cunvsm::DevBuf<float> A, B, C, rowsq, part;
        cunvsm::DevBuf<double> stats, part2;
        cunvsm::DevBuf<int> arrive;
        A.alloc(static_cast<size_t>(M) * K); B.alloc(static_cast<size_t>(K) * N); C.alloc(static_cast<size_t>(M) * N);
        {   // operands with all 24 significant bits in use (zeros would flatter a kernel: less switching, higher clocks)
            std::vector<float> h(std::max(A.n, B.n));
            uint32_t x = 12345u;
            auto fill = [&](cunvsm::DevBuf<float>& d, float scale) {
                for (size_t i = 0; i < d.n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (static_cast<float>(x >> 8) * (1.f / 8388608.f) - 1.f) * scale; }
                NVSM_HIP_CHECK(hipMemcpy(d.p, h.data(), d.n * sizeof(float), hipMemcpyHostToDevice));
            };
            fill(A, 0.5f); fill(B, 0.1f);
        }
        cunvsm::GridSumWs ws{};
        if (extras & 1) {
            ws.colgroups = 8; ws.contrib_cap = M / 32 + 512; ws.width_cap = 2 * (N > 160 ? N : 160); ws.groups_cap = ws.contrib_cap / 16 + 1; ws.fan = 16;
            part.alloc(static_cast<size_t>(ws.colgroups) * ws.contrib_cap * ws.width_cap);
            part2.alloc(static_cast<size_t>(ws.colgroups) * ws.groups_cap * ws.width_cap);
            arrive.alloc(static_cast<size_t>(ws.colgroups) * (ws.groups_cap + 1), true);
            stats.alloc(2 * static_cast<size_t>(N), true);
            ws.part = part.p; ws.part2 = part2.p; ws.arrive = arrive.p;
        }
        if (extras & 2) rowsq.alloc(static_cast<size_t>(M) * cunvsm::gemm_rowsq_parts(N));
        hipStream_t s;
        NVSM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipEvent_t e0, e1;
        NVSM_HIP_CHECK(hipEventCreate(&e0)); NVSM_HIP_CHECK(hipEventCreate(&e1));
        const int ldb = b_layout ? K : N;
        int parts = 0;
        cunvsm::DevBuf<char> planes, rplanes;
        planes.alloc(cunvsm::gemm_split_planes_bytes(N, K));
        rplanes.alloc(cunvsm::gemm_rsplit_planes_bytes(N, K));
        cunvsm::GemmSplitWs sws{planes.p, planes.n, false, rplanes.p, rplanes.n, false};
        auto go = [&] {
            sws.ready = (extras & 4) != 0 && sws.ready;          // extras bit 2: the planes of B stay valid between launches
            sws.rready = (extras & 4) != 0 && sws.rready;
            cunvsm::launch_gemm(0, b_layout, A.p, B.p, C.p, M, N, K, K, ldb, N, 1.f, nullptr, 1, 0, s, (extras & 1) ? stats.p : nullptr,
                                (extras & 2) ? rowsq.p : nullptr, 1.f, &parts, false, (extras & 1) ? &ws : nullptr, &sws);
        };
        for (int i = 0; i < 3; ++i) go();
        NVSM_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < repeats; ++i) go();
        NVSM_HIP_CHECK(hipEventRecord(e1, s));
        NVSM_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        NVSM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / static_cast<float>(repeats > 0 ? repeats : 1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    });
}

// the dT product alone on device-resident operands (phrase [rows][M], dx [rows][N]): average ms of the split-K kernel and of
// the reduce behind it. which 0 = gemm_dt (split-bf16), 2 = the tiled exact-fp32 kernel
int nvsm_debug_dt_time(int M, int N, int rows, int slabs, int repeats, int which, float* kernel_ms, float* reduce_ms) {
    NVSM_REQUIRE(kernel_ms); NVSM_REQUIRE(reduce_ms);
    return guarded_hook([&] {
        cunvsm::DevBuf<float> A, B, C, P;
        A.alloc(static_cast<size_t>(rows) * M); B.alloc(static_cast<size_t>(rows) * N); C.alloc(static_cast<size_t>(M) * N);
        {
            std::vector<float> h(std::max(A.n, B.n));
            uint32_t x = 4242u;
            auto fill = [&](cunvsm::DevBuf<float>& d, float scale) {
                for (size_t i = 0; i < d.n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (static_cast<float>(x >> 8) * (1.f / 8388608.f) - 1.f) * scale; }
                NVSM_HIP_CHECK(hipMemcpy(d.p, h.data(), d.n * sizeof(float), hipMemcpyHostToDevice));
            };
            fill(A, 0.5f); fill(B, 0.1f);
        }
        hipStream_t s;
        NVSM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        if (which != 0 && which != 2) throw Error(NVSM_ERR_INVALID_ARGUMENT, "which: 0 = split-bf16 kernel, 2 = tiled fp32 kernel");
        const int ns = which == 0 ? cunvsm::gemm_dt_slabs(rows, slabs) : cunvsm::gemm_split_k_slabs(rows, slabs);
        const size_t stride = static_cast<size_t>(M) * N;
        P.alloc(static_cast<size_t>(std::max(ns, 1)) * stride);
        auto product = [&] {
            bool ok = true;
            if (which == 0) ok = cunvsm::launch_gemm_dt(A.p, B.p, P.p, M, N, rows, M, N, slabs, s);
            else cunvsm::launch_gemm(1, 0, A.p, B.p, P.p, M, N, rows, M, N, N, 1.f, nullptr, slabs, stride, s);
            if (!ok) throw Error(NVSM_ERR_UNSUPPORTED, "the dT kernel refused the shape");
        };
        auto reduce = [&] { cunvsm::launch_splitk_reduce(P.p, ns, stride, C.p, static_cast<int64_t>(stride), s); };
        hipEvent_t e0, e1, e2;
        NVSM_HIP_CHECK(hipEventCreate(&e0)); NVSM_HIP_CHECK(hipEventCreate(&e1)); NVSM_HIP_CHECK(hipEventCreate(&e2));
        for (int i = 0; i < 3; ++i) { product(); reduce(); }
        const int reps = repeats > 0 ? repeats : 1;
        NVSM_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) product();
        NVSM_HIP_CHECK(hipEventRecord(e1, s));
        for (int i = 0; i < reps; ++i) reduce();
        NVSM_HIP_CHECK(hipEventRecord(e2, s));
        NVSM_HIP_CHECK(hipEventSynchronize(e2));
        float a = 0.f, b = 0.f;
        NVSM_HIP_CHECK(hipEventElapsedTime(&a, e0, e1)); NVSM_HIP_CHECK(hipEventElapsedTime(&b, e1, e2));
        *kernel_ms = a / reps; *reduce_ms = b / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipStreamDestroy(s);
    });
}

int nvsm_debug_sort(int64_t n, int bits, const int32_t* keys, int32_t* keys_out, int32_t* vals_out, int repeats, float* avg_ms) {
    NVSM_REQUIRE(keys); NVSM_REQUIRE(keys_out); NVSM_REQUIRE(vals_out);
    return guarded_hook([&] {
        if (n <= 0) return;
        cunvsm::DevBuf<int> K, KO, VO;
        cunvsm::DevBuf<char> tmp;
        K.alloc(n); KO.alloc(n); VO.alloc(n);
        const size_t tb = cunvsm::sort_pairs_temp_bytes(n, bits);
        tmp.alloc(tb, true);
        int* err = nullptr;
        NVSM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&err), sizeof(int), hipHostMallocDefault));
        *err = 0;
        uint64_t epoch = 0;
        NVSM_HIP_CHECK(hipMemcpy(K.p, keys, n * sizeof(int), hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        NVSM_HIP_CHECK(hipEventCreate(&e0)); NVSM_HIP_CHECK(hipEventCreate(&e1));
        cunvsm::sort_pairs(tmp.p, tb, &epoch, K.p, KO.p, nullptr, VO.p, n, bits, err, nullptr);      // warm-up
        NVSM_HIP_CHECK(hipEventRecord(e0, nullptr));
        const int reps = repeats > 0 ? repeats : 1;
        for (int r = 0; r < reps; ++r)      // repeats: the arrival counter keeps growing across calls
            cunvsm::sort_pairs(tmp.p, tb, &epoch, K.p, KO.p, nullptr, VO.p, n, bits, err, nullptr);
        NVSM_HIP_CHECK(hipEventRecord(e1, nullptr));
        NVSM_HIP_CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        NVSM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (avg_ms) *avg_ms = ms / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        const int code = *err;
        (void)hipHostFree(err);
        if (code) throw Error(NVSM_ERR_DEVICE, "sort reported an error");
        NVSM_HIP_CHECK(hipMemcpy(keys_out, KO.p, n * sizeof(int), hipMemcpyDeviceToHost));
        NVSM_HIP_CHECK(hipMemcpy(vals_out, VO.p, n * sizeof(int), hipMemcpyDeviceToHost));
    });
}

int nvsm_debug_gather_mean(int64_t num_rows, int dim, const float* table, const int64_t* idx, const float* wts,
                           int window, int64_t num_out, float* out) {
    NVSM_REQUIRE(table); NVSM_REQUIRE(idx); NVSM_REQUIRE(out);
    return guarded_hook([&] {
        cunvsm::DevBuf<float> T, W, O;
        cunvsm::DevBuf<int64_t> I64;
        cunvsm::DevBuf<int> I;
        T.alloc(num_rows * dim); O.alloc(num_out * dim); I64.alloc(num_out * window); I.alloc(num_out * window);
        NVSM_HIP_CHECK(hipMemcpy(T.p, table, T.n * sizeof(float), hipMemcpyHostToDevice));
        NVSM_HIP_CHECK(hipMemcpy(I64.p, idx, I64.n * sizeof(int64_t), hipMemcpyHostToDevice));
        if (wts) { W.alloc(num_out * window); NVSM_HIP_CHECK(hipMemcpy(W.p, wts, W.n * sizeof(float), hipMemcpyHostToDevice)); }
        cunvsm::launch_narrow_i64(I64.p, I.p, num_out * window, num_rows, nullptr, 0, nullptr);
        cunvsm::launch_gather_mean(T.p, dim, I.p, wts ? W.p : nullptr, window, num_out, O.p, nullptr);
        NVSM_HIP_CHECK(hipDeviceSynchronize());
        NVSM_HIP_CHECK(hipMemcpy(out, O.p, O.n * sizeof(float), hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
