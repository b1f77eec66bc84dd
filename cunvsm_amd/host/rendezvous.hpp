// File rendezvous through which rank 0 of a data-parallel cuNVSMTrainModel run hands the 128-byte RCCL id
// (ncclUniqueId) to the other ranks of the node. No counterpart in the reference, which is single-GPU
// (cpp/model.cu:13-14). The file lives in a directory only the user can write, is created exclusively without
// following links, and carries a nonce that names the run, so that neither a leftover of a crashed run nor a file
// planted by another local user is ever taken for this run's id.
#pragma once

#include <cstdint>
#include <string>

namespace nvsm_host {

constexpr int kCommIdBytes = 128;

// What names this run to every one of its ranks: `explicit_nonce` (--comm_nonce, what --gpus N hands its children) when
// given, else the launcher's run id (TORCHELASTIC_RUN_ID unless "none"), else parent pid + MASTER_PORT — the ranks of one
// torch.distributed.run agent share both.
std::string comm_run_nonce(const std::string& explicit_nonce);
// $XDG_RUNTIME_DIR, else /tmp/cunvsm-<uid> (created 0700; refused unless it is a real directory owned by the user and
// closed to everybody else), + "/comm_" + a hash of the nonce
std::string default_comm_id_path(const std::string& nonce);
int64_t wall_clock_ns();

// rank 0, first thing: a leftover of an earlier run under the same name goes before any reader can look at it
void rendezvous_clear(const std::string& path);
// rank 0: header (magic, nonce hash, creation time) + id into `path`.tmp.<pid> opened O_CREAT | O_EXCL | O_NOFOLLOW, mode 0600,
// then renamed over `path` — a reader sees all of it or nothing. Throws FatalError on failure.
void rendezvous_publish(const std::string& path, const std::string& nonce, const char id[kCommIdBytes]);
// other ranks: true when `path` holds this run's id. A file that is a symbolic link, belongs to another user, is open to
// group / others, carries another nonce or was created before `not_before_ns` is NOT this run's (why says which) — the
// caller keeps polling until rank 0 has replaced it or its patience runs out.
bool rendezvous_read(const std::string& path, const std::string& nonce, int64_t not_before_ns, char id[kCommIdBytes], std::string* why);

}  // namespace nvsm_host
