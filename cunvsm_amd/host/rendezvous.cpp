#include "rendezvous.hpp"

#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "base.hpp"

namespace nvsm_host {

namespace {

struct Header {
    char magic[8];            // "NVSMRCC2"
    uint64_t nonce_hash;
    int64_t created_ns;
    uint32_t pid;
    uint32_t id_bytes;
    // where `pid` means something: the writer's host and pid namespace. A reader elsewhere (another container or node of an external
    // launcher that shares the file over a volume) cannot ask whether that pid is alive and does not try.
    uint64_t host_hash;       // fnv1a(hostname | boot id)
    uint64_t pidns_inode;     // st_ino of /proc/self/ns/pid (0: unknown)
};
const char kMagic[8] = {'N', 'V', 'S', 'M', 'R', 'C', 'C', '2'};

uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

uint64_t this_host_hash() {
    char host[256] = {0};
    (void)::gethostname(host, sizeof(host) - 1);
    std::string key = host;
    if (FILE* f = std::fopen("/proc/sys/kernel/random/boot_id", "r")) {
        char b[64] = {0};
        if (std::fgets(b, sizeof(b), f)) key += std::string("|") + b;
        std::fclose(f);
    }
    return fnv1a(key);
}
uint64_t this_pidns_inode() {
    struct stat st;
    return ::stat("/proc/self/ns/pid", &st) == 0 ? static_cast<uint64_t>(st.st_ino) : 0;
}

bool write_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n > 0) {
        const ssize_t k = ::write(fd, c, n);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        c += k; n -= static_cast<size_t>(k);
    }
    return true;
}

}  // namespace

int64_t wall_clock_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

std::string comm_run_nonce(const std::string& explicit_nonce) {
    if (!explicit_nonce.empty()) return explicit_nonce;
    const char* run_id = std::getenv("TORCHELASTIC_RUN_ID");
    if (run_id && *run_id && std::strcmp(run_id, "none") != 0) return std::string("run:") + run_id;
    const char* port = std::getenv("MASTER_PORT");
    return "ppid:" + std::to_string(static_cast<long long>(getppid())) + ":port:" + (port ? port : "");
}

std::string default_comm_id_path(const std::string& nonce) {
    std::string dir;
    const char* xdg = std::getenv("XDG_RUNTIME_DIR");
    struct stat st;
    if (xdg && *xdg && ::stat(xdg, &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 077) == 0) {
        dir = xdg;
    } else {
        dir = "/tmp/cunvsm-" + std::to_string(static_cast<long long>(geteuid()));
        if (::mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) NVSM_LOG(FATAL) << "cannot create " << dir << ": " << std::strerror(errno);
        // (lstat: a symbolic link somebody else put there is not our directory)
        if (::lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & 077) != 0)
            NVSM_LOG(FATAL) << dir << " is not a private directory of this user; pass --comm_id_file.";
    }
    char name[64];
    std::snprintf(name, sizeof(name), "/comm_%016llx", static_cast<unsigned long long>(fnv1a(nonce)));
    return dir + name;
}

void rendezvous_clear(const std::string& path) {
    if (::unlink(path.c_str()) != 0 && errno != ENOENT) NVSM_LOG(FATAL) << "cannot remove the stale rendezvous file " << path << ": " << std::strerror(errno);
}

void rendezvous_publish(const std::string& path, const std::string& nonce, const char id[kCommIdBytes]) {
    const std::string tmp = path + ".tmp." + std::to_string(static_cast<long long>(getpid()));
    ::unlink(tmp.c_str());
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) NVSM_LOG(FATAL) << "cannot create " << tmp << ": " << std::strerror(errno);
    Header h{};
    std::memcpy(h.magic, kMagic, 8);
    h.nonce_hash = fnv1a(nonce); h.created_ns = wall_clock_ns(); h.pid = static_cast<uint32_t>(getpid()); h.id_bytes = kCommIdBytes;
    h.host_hash = this_host_hash(); h.pidns_inode = this_pidns_inode();
    const bool ok = write_all(fd, &h, sizeof(h)) && write_all(fd, id, kCommIdBytes) && ::fsync(fd) == 0;
    ::close(fd);
    if (!ok || ::rename(tmp.c_str(), path.c_str()) != 0) {
        const int e = errno;
        ::unlink(tmp.c_str());
        NVSM_LOG(FATAL) << "cannot publish " << path << ": " << std::strerror(e);
    }
}

bool rendezvous_read(const std::string& path, const std::string& nonce, int64_t not_before_ns, char id[kCommIdBytes], std::string* why) {
    auto no = [&](const char* reason) { if (why) *why = reason; return false; };
    const int fd = ::open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return no(errno == ELOOP ? "is a symbolic link" : "not there yet");
    struct stat st;
    Header h{};
    char buf[kCommIdBytes];
    bool ok = ::fstat(fd, &st) == 0;
    const char* reason = "cannot be examined";
    if (ok && !S_ISREG(st.st_mode)) { ok = false; reason = "is not a regular file"; }
    if (ok && st.st_uid != geteuid()) { ok = false; reason = "belongs to another user"; }
    if (ok && (st.st_mode & 077) != 0) { ok = false; reason = "is accessible to group / others"; }
    if (ok && (::read(fd, &h, sizeof(h)) != static_cast<ssize_t>(sizeof(h)) || ::read(fd, buf, kCommIdBytes) != kCommIdBytes)) { ok = false; reason = "is incomplete"; }
    ::close(fd);
    if (!ok) return no(reason);
    if (std::memcmp(h.magic, kMagic, 8) != 0 || h.id_bytes != kCommIdBytes) return no("is not a rendezvous file");
    if (h.nonce_hash != fnv1a(nonce)) return no("belongs to another run (nonce)");
    if (h.created_ns < not_before_ns) return no("is older than this run");
    // The writer (rank 0 of this launch) is alive for as long as anybody may read the file: a file whose writer is gone was left
    // behind by a run that crashed — a relaunch from the same shell within the clock-skew window carries the same fall-back nonce
    // (parent pid + port), and its ranks would otherwise join a communicator id nobody is waiting on. The question can only be
    // asked where the writer's pid means something: on its host, in its pid namespace (the header says which). A reader in
    // another container / on another node — an external launcher with --comm_id_file on a shared volume — relies on nonce and
    // creation time alone.
    // (/proc/self/ns/pid unreadable on BOTH sides — inode 0 in the header and here — on the same host: the check is made as if the
    //  namespaces agreed, as it was before the header carried one; a reader that knows its namespace never trusts a writer that did not)
    const unsigned long long my_ns = this_pidns_inode();
    const bool same_ns = h.pidns_inode != 0 ? h.pidns_inode == my_ns : my_ns == 0;
    const bool same_place = h.host_hash == this_host_hash() && same_ns;
    if (h.pid == 0) return no("carries no writer pid");
    if (same_place && ::kill(static_cast<pid_t>(h.pid), 0) != 0 && errno == ESRCH) return no("was written by a process that no longer exists (stale)");
    std::memcpy(id, buf, kCommIdBytes);
    return true;
}

}  // namespace nvsm_host
