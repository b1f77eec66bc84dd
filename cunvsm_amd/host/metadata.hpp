// lse::Metadata (proto/nvsm.proto:88-103) without libprotobuf: a plain struct plus an encoder / decoder of the
// protobuf wire format, so that "<output>_meta" (cpp/main.cu:527-537) stays readable by the reference's Python tools
// (py/nvsm/base.py parses it with the generated nvsm_pb2).
//   message Metadata { repeated TermInfo term = 1; repeated ObjectInfo object = 2; int32 total_terms = 3; }
//   message TermInfo { int32 index_term_id = 1; int32 model_term_id = 2; int32 term_frequency = 3; }
//   message ObjectInfo { int32 index_object_id = 1; int32 model_object_id = 2; }
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace nvsm_host {

struct Metadata {
    struct TermInfo { int32_t index_term_id = 0, model_term_id = 0, term_frequency = 0; };
    struct ObjectInfo { int32_t index_object_id = 0, model_object_id = 0; };
    std::vector<TermInfo> term;
    std::vector<ObjectInfo> object;
    int32_t total_terms = 0;

    size_t term_size() const { return term.size(); }
    size_t object_size() const { return object.size(); }

    std::string SerializeAsString() const;          // proto3: zero-valued scalars are omitted, as protoc's code does
    bool ParseFromString(const std::string& data);  // accepts any field order, skips unknown fields
};

}  // namespace nvsm_host
