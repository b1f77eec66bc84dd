// Host-side foundations of the cuNVSMTrainModel replacement (everything above the C ABI of libcunvsm_amd.so):
// integer / float typedefs, the shared generator type, small string helpers and a glog-shaped logger.
// Mirrors include/cuNVSM/base.h (typedefs :25-36, split :104-110, seconds_to_humanreadable_time :241-254,
// is_number :256-259) — restated, glog / protobuf free.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <iterator>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace nvsm_host {

// include/cuNVSM/base.h:28 — `typedef long int32;` : indices are 64-bit on LP64.
typedef int64_t WordIdxType;
typedef int64_t ObjectIdxType;
typedef float WeightType;              // release build: FLOATING_POINT_TYPE=float32 (cpp/CMakeLists.txt:17)
typedef std::minstd_rand0 RNG;         // include/cuNVSM/base.h:36

struct FatalError : std::runtime_error {
    explicit FatalError(const std::string& what) : std::runtime_error(what) {}
};

// ---- logging: "I0929 12:34:56.123456 file:line] message" on stderr, as glog prints it ----
enum LogSeverity { LOG_INFO = 0, LOG_WARNING = 1, LOG_ERROR = 2, LOG_FATAL = 3 };
int& verbosity();                      // --v
bool& log_to_stderr();                 // --logtostderr (default true; there is no log-file sink)

class LogMessage {
 public:
    LogMessage(const char* file, int line, LogSeverity severity);
    ~LogMessage() noexcept(false);
    std::ostream& stream() { return stream_; }
 private:
    std::ostringstream stream_;
    LogSeverity severity_;
};

#define NVSM_LOG(severity) ::nvsm_host::LogMessage(__FILE__, __LINE__, ::nvsm_host::LOG_##severity).stream()
#define NVSM_VLOG(level) if (::nvsm_host::verbosity() >= (level)) NVSM_LOG(INFO)
#define NVSM_CHECK(cond) if (!(cond)) NVSM_LOG(FATAL) << "Check failed: " #cond " "

inline std::vector<std::string> split(const std::string& str) {
    std::istringstream iss(str);
    return std::vector<std::string>(std::istream_iterator<std::string>{iss}, std::istream_iterator<std::string>{});
}

// include/cuNVSM/base.h:256-259: "is a number" = contains at least one digit.
inline bool is_number(const std::string& s) {
    for (unsigned char c : s) if (std::isdigit(c)) return true;
    return false;
}

inline std::string seconds_to_humanreadable_time(double seconds) {
    const long hours = static_cast<long>(std::floor(seconds / 3600.0));
    seconds -= hours * 3600.0;
    const long minutes = static_cast<long>(std::floor(seconds / 60.0));
    seconds -= minutes * 60.0;
    std::stringstream stream;
    stream << hours << " hours, " << minutes << " minutes and " << static_cast<long>(std::floor(seconds)) << " seconds";
    return stream.str();
}

template <typename T>
inline std::vector<T> range(size_t start, size_t end, size_t repeat = 1) {     // include/cuNVSM/base.h:158-168
    std::vector<T> v;
    for (size_t i = start; i < end; ++i)
        for (size_t j = 0; j < repeat; ++j) v.push_back(static_cast<T>(i));
    return v;
}

template <typename T>
inline void flatten(const std::vector<std::vector<T>>& iterable, std::vector<T>* const flattened) {   // :147-156
    for (const auto& instance : iterable) flattened->insert(flattened->end(), instance.begin(), instance.end());
}

}  // namespace nvsm_host
