#include "flags.hpp"

#include <cerrno>
#include <cstdlib>
#include <sstream>

#include "base.hpp"

namespace nvsm_host {

void Flags::assign(const std::string& name, const Entry& e, const std::string& value) {
    errno = 0;
    char* end = nullptr;
    switch (e.type) {
        case 'u': {
            if (!value.empty() && value[0] == '-') NVSM_LOG(FATAL) << "illegal value '" << value << "' specified for uint64 flag '" << name << "'";
            const unsigned long long v = std::strtoull(value.c_str(), &end, 10);
            if (errno || end == value.c_str() || *end) NVSM_LOG(FATAL) << "illegal value '" << value << "' specified for uint64 flag '" << name << "'";
            *static_cast<uint64_t*>(e.var) = v;
            break;
        }
        case 'i': {
            const long long v = std::strtoll(value.c_str(), &end, 10);
            if (errno || end == value.c_str() || *end) NVSM_LOG(FATAL) << "illegal value '" << value << "' specified for int64 flag '" << name << "'";
            *static_cast<int64_t*>(e.var) = v;
            break;
        }
        case 'd': {
            const double v = std::strtod(value.c_str(), &end);
            if (errno || end == value.c_str() || *end) NVSM_LOG(FATAL) << "illegal value '" << value << "' specified for double flag '" << name << "'";
            *static_cast<double*>(e.var) = v;
            break;
        }
        case 'b': {
            if (value == "true" || value == "1" || value == "t" || value == "yes" || value == "y") *static_cast<bool*>(e.var) = true;
            else if (value == "false" || value == "0" || value == "f" || value == "no" || value == "n") *static_cast<bool*>(e.var) = false;
            else NVSM_LOG(FATAL) << "illegal value '" << value << "' specified for bool flag '" << name << "'";
            break;
        }
        default: *static_cast<std::string*>(e.var) = value;
    }
}

std::vector<std::string> Flags::parse(int argc, char** argv) {
    std::vector<std::string> positional;
    if (argc > 0) positional.push_back(argv[0]);
    bool options_done = false;
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (options_done || arg.size() < 2 || arg[0] != '-') { positional.push_back(arg); continue; }
        if (arg == "--") { options_done = true; continue; }
        std::string body = arg.substr(arg[1] == '-' ? 2 : 1);
        std::string name = body, value;
        bool has_value = false;
        const size_t eq = body.find('=');
        if (eq != std::string::npos) { name = body.substr(0, eq); value = body.substr(eq + 1); has_value = true; }
        auto it = flags_.find(name);
        if (it == flags_.end() && !has_value && name.compare(0, 2, "no") == 0) {         // --noflag
            auto neg = flags_.find(name.substr(2));
            if (neg != flags_.end() && neg->second.type == 'b') { *static_cast<bool*>(neg->second.var) = false; continue; }
        }
        if (it == flags_.end()) NVSM_LOG(FATAL) << "unknown command line flag '" << name << "'";
        if (!has_value) {
            if (it->second.type == 'b') { *static_cast<bool*>(it->second.var) = true; continue; }
            if (i + 1 >= argc) NVSM_LOG(FATAL) << "flag '" << name << "' is missing its argument";
            value = argv[++i];
        }
        assign(name, it->second, value);
    }
    return positional;
}

std::string Flags::usage() const {
    std::ostringstream os;
    for (const auto& kv : flags_) {
        const char* type = kv.second.type == 'u' ? "uint64" : kv.second.type == 'i' ? "int64" : kv.second.type == 'd' ? "double"
                         : kv.second.type == 'b' ? "bool" : "string";
        os << "    -" << kv.first << " (" << kv.second.help << ") type: " << type << " default: " << kv.second.def << "\n";
    }
    return os.str();
}

}  // namespace nvsm_host
