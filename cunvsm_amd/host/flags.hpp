// A gflags-shaped command-line parser (the reference defines its options with DEFINE_* and calls
// gflags::ParseCommandLineFlags(&argc, &argv, true), cpp/main.cu:15-76,625): --name=value, --name value, -name,
// --boolflag, --noboolflag, "--" ends the options; parsed flags are removed from argv. Unknown flags are an error.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace nvsm_host {

class Flags {
 public:
    void define_uint64(const std::string& name, uint64_t* var, uint64_t def, const std::string& help) { *var = def; add(name, 'u', var, help, std::to_string(def)); }
    void define_int64(const std::string& name, int64_t* var, int64_t def, const std::string& help) { *var = def; add(name, 'i', var, help, std::to_string(def)); }
    void define_double(const std::string& name, double* var, double def, const std::string& help) { *var = def; add(name, 'd', var, help, std::to_string(def)); }
    void define_bool(const std::string& name, bool* var, bool def, const std::string& help) { *var = def; add(name, 'b', var, help, def ? "true" : "false"); }
    void define_string(const std::string& name, std::string* var, const std::string& def, const std::string& help) { *var = def; add(name, 's', var, help, "\"" + def + "\""); }

    // Returns the positional arguments (argv[0] included). Throws FatalError on a malformed / unknown flag.
    std::vector<std::string> parse(int argc, char** argv);
    std::string usage() const;

 private:
    struct Entry { char type; void* var; std::string help, def; };
    void add(const std::string& name, char type, void* var, const std::string& help, const std::string& def) { flags_[name] = Entry{type, var, help, def}; }
    void assign(const std::string& name, const Entry& e, const std::string& value);
    std::map<std::string, Entry> flags_;
};

}  // namespace nvsm_host
