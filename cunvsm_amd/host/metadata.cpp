#include "metadata.hpp"

namespace nvsm_host {
namespace {

void put_varint(std::string* out, uint64_t v) {
    while (v >= 0x80) { out->push_back(static_cast<char>((v & 0x7f) | 0x80)); v >>= 7; }
    out->push_back(static_cast<char>(v));
}
// int32 fields are encoded as the sign-extended 64-bit varint (10 bytes when negative)
void put_int32(std::string* out, int field, int32_t v) {
    if (v == 0) return;                                  // proto3 default: not on the wire
    put_varint(out, static_cast<uint64_t>(field) << 3);  // wire type 0
    put_varint(out, static_cast<uint64_t>(static_cast<int64_t>(v)));
}
void put_message(std::string* out, int field, const std::string& body) {
    put_varint(out, (static_cast<uint64_t>(field) << 3) | 2);
    put_varint(out, body.size());
    out->append(body);
}

struct Reader {
    const unsigned char* p; const unsigned char* end; bool ok = true;
    uint64_t varint() {
        uint64_t v = 0; int shift = 0;
        while (p < end && shift < 64) {
            const unsigned char b = *p++;
            v |= static_cast<uint64_t>(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
        }
        ok = false; return 0;
    }
    bool skip(int wire) {
        switch (wire) {
            case 0: varint(); return ok;
            case 1: if (end - p < 8) return ok = false; p += 8; return true;
            case 2: { const uint64_t n = varint(); if (!ok || static_cast<uint64_t>(end - p) < n) return ok = false; p += n; return true; }
            case 5: if (end - p < 4) return ok = false; p += 4; return true;
            default: return ok = false;
        }
    }
};

template <typename Fn>
bool parse_fields(const unsigned char* p, const unsigned char* end, Fn&& on_field) {
    Reader r{p, end};
    while (r.p < r.end) {
        const uint64_t key = r.varint();
        if (!r.ok) return false;
        const int field = static_cast<int>(key >> 3), wire = static_cast<int>(key & 7);
        if (!on_field(field, wire, r)) return false;
    }
    return r.ok;
}

}  // namespace

std::string Metadata::SerializeAsString() const {
    std::string out;
    for (const TermInfo& t : term) {
        std::string body;
        put_int32(&body, 1, t.index_term_id); put_int32(&body, 2, t.model_term_id); put_int32(&body, 3, t.term_frequency);
        put_message(&out, 1, body);
    }
    for (const ObjectInfo& o : object) {
        std::string body;
        put_int32(&body, 1, o.index_object_id); put_int32(&body, 2, o.model_object_id);
        put_message(&out, 2, body);
    }
    put_int32(&out, 3, total_terms);
    return out;
}

bool Metadata::ParseFromString(const std::string& data) {
    term.clear(); object.clear(); total_terms = 0;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(data.data());
    return parse_fields(b, b + data.size(), [&](int field, int wire, Reader& r) {
        if ((field == 1 || field == 2) && wire == 2) {
            const uint64_t n = r.varint();
            if (!r.ok || static_cast<uint64_t>(r.end - r.p) < n) return false;
            const unsigned char* sub = r.p; r.p += n;
            if (field == 1) {
                TermInfo t;
                if (!parse_fields(sub, sub + n, [&](int f, int w, Reader& rr) {
                        if (w != 0) return rr.skip(w);
                        const int32_t v = static_cast<int32_t>(rr.varint());
                        if (f == 1) t.index_term_id = v; else if (f == 2) t.model_term_id = v; else if (f == 3) t.term_frequency = v;
                        return rr.ok; })) return false;
                term.push_back(t);
            } else {
                ObjectInfo o;
                if (!parse_fields(sub, sub + n, [&](int f, int w, Reader& rr) {
                        if (w != 0) return rr.skip(w);
                        const int32_t v = static_cast<int32_t>(rr.varint());
                        if (f == 1) o.index_object_id = v; else if (f == 2) o.model_object_id = v;
                        return rr.ok; })) return false;
                object.push_back(o);
            }
            return true;
        }
        if (field == 3 && wire == 0) { total_terms = static_cast<int32_t>(r.varint()); return r.ok; }
        return r.skip(wire);
    });
}

}  // namespace nvsm_host
