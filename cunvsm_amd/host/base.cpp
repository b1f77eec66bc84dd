#include "base.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <mutex>

namespace nvsm_host {

int& verbosity() { static int v = 0; return v; }
bool& log_to_stderr() { static bool b = true; return b; }

LogMessage::LogMessage(const char* file, int line, LogSeverity severity) : severity_(severity) {
    const char* base = std::strrchr(file, '/');
    base = base ? base + 1 : file;
    const auto now = std::chrono::system_clock::now();
    const std::time_t t = std::chrono::system_clock::to_time_t(now);
    const long usec = static_cast<long>(std::chrono::duration_cast<std::chrono::microseconds>(now.time_since_epoch()).count() % 1000000);
    std::tm tm_buf;
    localtime_r(&t, &tm_buf);
    char head[64];
    std::snprintf(head, sizeof(head), "%c%02d%02d %02d:%02d:%02d.%06ld ", "IWEF"[severity], tm_buf.tm_mon + 1, tm_buf.tm_mday,
                  tm_buf.tm_hour, tm_buf.tm_min, tm_buf.tm_sec, usec);
    stream_ << head << base << ":" << line << "] ";
}

LogMessage::~LogMessage() noexcept(false) {
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (log_to_stderr() || severity_ >= LOG_ERROR) std::cerr << stream_.str() << std::endl;
    }
    // glog aborts on LOG(FATAL) / failed CHECKs; the host layer raises instead so that the CLI can exit(1) cleanly
    // and the tests can observe the failure (cpp/main.cu:113-134 installs a terminate handler for the same purpose).
    if (severity_ == LOG_FATAL) throw FatalError(stream_.str());
}

}  // namespace nvsm_host
