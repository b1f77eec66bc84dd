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

#include "c_api_internal.h"

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
namespace cunvsm { void set_last_error(const std::string& what) { g_last_error = what; } }
using cunvsm::guarded;
using cunvsm::guarded_on;
using cunvsm::guarded_hook;

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

}  // extern "C"
