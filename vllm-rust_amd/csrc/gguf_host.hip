// GGUF container: header / metadata / tensor-table parser and tensor byte reader (host code only), the C form of
// GgufHeader::read and read_tensor_bytes (crates/core/src/quantization/gguf/header.rs:221-321) plus the HF -> llama.cpp
// tensor-name map of the GGUF weight loader (gguf/mod.rs:1416-1479).  Versions 1-3 (v1: u32 counts and lengths), on-disk
// dims reversed into the logical [out, in] orientation, tensor data aligned to general.alignment (default 32), every
// GGML type id the reference classifies (candle-native, I-quants, plain integers); anything else is an error that names
// the tensor.
#include "vmx_internal.h"

#include <stdlib.h>

#include <map>
#include <exception>
#include <new>
#include <string>
#include <vector>

namespace {

struct MetaValue {
    uint32_t type = 0;          // GGUF value-type id 0..12 (9 = array)
    uint32_t elem_type = 0;     // arrays: element type
    uint64_t u = 0;             // integers / bool (sign-extended into i for the signed types)
    int64_t i = 0;
    double f = 0.0;
    std::string s;
    std::vector<MetaValue> arr;
};

struct TensorInfo {
    std::string name;
    std::vector<uint64_t> shape;   // logical orientation (on-disk dims reversed)
    uint32_t ggml_type = 0;
    uint64_t offset = 0;           // relative to tensor_data_offset
};

// bytes per block / elements per block of a GGML type id, or 0 / 0 when the id is not one the reference accepts
// (header.rs:325-362; iq/mod.rs:55-93)
bool type_geometry(uint32_t id, uint32_t *type_size, uint32_t *block) {
    struct G { uint32_t id, ts, bs; };
    static const G tab[] = {{0, 4, 1},     {1, 2, 1},     {2, 18, 32},   {3, 20, 32},   {6, 22, 32},   {7, 24, 32},  {8, 34, 32},
                            {9, 36, 32},   {10, 84, 256}, {11, 110, 256}, {12, 144, 256}, {13, 176, 256}, {14, 210, 256}, {15, 292, 256},
                            {30, 2, 1},    {24, 1, 1},    {25, 2, 1},    {26, 4, 1},    {27, 8, 1},    {17, 74, 256}, {18, 98, 256},
                            {21, 110, 256}, {22, 82, 256}, {23, 136, 256}};
    for (const G &g : tab)
        if (g.id == id) {
            *type_size = g.ts;
            *block = g.bs;
            return true;
        }
    return false;
}

struct Reader {
    const uint8_t *p = nullptr;
    size_t n = 0, pos = 0;
    FILE *f = nullptr;
    uint64_t fsize = 0;   // file mode: total bytes
    bool fail = false;
    // bytes that can still be read: every length field of the header is checked against it before anything is resized
    uint64_t remaining() const { return f ? (fsize > pos ? fsize - pos : 0) : (uint64_t)(n - pos); }
    bool take(void *dst, size_t k) {
        if (fail) return false;
        if (f) {
            if (fread(dst, 1, k, f) != k) { fail = true; return false; }
            pos += k;
            return true;
        }
        if (k > n - pos) { fail = true; return false; }
        memcpy(dst, p + pos, k);
        pos += k;
        return true;
    }
    template <typename T> T le() {
        uint8_t b[sizeof(T)] = {0};
        take(b, sizeof(T));
        uint64_t v = 0;
        for (size_t i = 0; i < sizeof(T); i++) v |= (uint64_t)b[i] << (8 * i);
        T out;
        memcpy(&out, &v, sizeof(T));
        return out;
    }
    uint64_t count(bool is64) { return is64 ? le<uint64_t>() : (uint64_t)le<uint32_t>(); }
    bool str(bool is64, std::string &out) {
        const uint64_t len = count(is64);
        if (fail || len > ((uint64_t)1 << 31) || len > remaining()) { fail = true; return false; }
        out.resize((size_t)len);
        return len == 0 || take(&out[0], (size_t)len);
    }
};

// 1 ok, 0 truncated, -1 bad value
int read_value(Reader &r, uint32_t type, bool is64, MetaValue &v, int depth) {
    v.type = type;
    switch (type) {
        case 0: v.u = r.le<uint8_t>(); v.i = (int64_t)v.u; break;
        case 1: v.i = (int8_t)r.le<uint8_t>(); v.u = (uint64_t)v.i; break;
        case 2: v.u = r.le<uint16_t>(); v.i = (int64_t)v.u; break;
        case 3: v.i = (int16_t)r.le<uint16_t>(); v.u = (uint64_t)v.i; break;
        case 4: v.u = r.le<uint32_t>(); v.i = (int64_t)v.u; break;
        case 5: v.i = (int32_t)r.le<uint32_t>(); v.u = (uint64_t)v.i; break;
        case 6: { const uint32_t b = r.le<uint32_t>(); float f; memcpy(&f, &b, 4); v.f = f; break; }
        case 7: {
            const uint8_t b = r.le<uint8_t>();
            if (!r.fail && b > 1) { vmx_set_error("GGUF: unexpected bool value %u", b); return -1; }
            v.u = b; v.i = b;
            break;
        }
        case 8: r.str(is64, v.s); break;
        case 9: {
            if (depth > 4) { vmx_set_error("GGUF: metadata arrays nested too deep"); return -1; }
            v.elem_type = r.le<uint32_t>();
            const uint64_t len = r.count(is64);
            if (r.fail) return 0;
            if (len > ((uint64_t)1 << 28)) { vmx_set_error("GGUF: metadata array of %llu elements", (unsigned long long)len); return -1; }
            if (len > r.remaining()) { r.fail = true; return 0; }   // an element takes at least one byte
            v.arr.resize((size_t)len);
            for (size_t i = 0; i < v.arr.size(); i++) {
                const int rc = read_value(r, v.elem_type, is64, v.arr[i], depth + 1);
                if (rc != 1) return rc;
            }
            break;
        }
        case 10: v.u = r.le<uint64_t>(); v.i = (int64_t)v.u; break;
        case 11: v.i = (int64_t)r.le<uint64_t>(); v.u = (uint64_t)v.i; break;
        case 12: { const uint64_t b = r.le<uint64_t>(); memcpy(&v.f, &b, 8); break; }
        default: vmx_set_error("GGUF: unrecognized metadata value-type %u", type); return -1;
    }
    return r.fail ? 0 : 1;
}

}  // namespace

struct vmx_gguf {
    uint32_t version = 0;
    uint64_t alignment = 32, tensor_data_offset = 0;
    std::map<std::string, MetaValue> meta;
    std::vector<std::string> meta_keys;   // file order
    std::vector<TensorInfo> tensors;      // file order
    std::map<std::string, int> by_name;
    std::string path;
};

namespace {
// product of the dims with overflow check
bool element_count(const TensorInfo &t, uint64_t *out) {
    uint64_t n = 1;
    for (uint64_t d : t.shape) {
        if (d != 0 && n > UINT64_MAX / d) return false;
        n *= d;
    }
    *out = n;
    return true;
}

int parse_unguarded(Reader &r, vmx_gguf *g);

// The parser allocates from lengths it reads; an allocation failure or any other C++ exception must not cross the C ABI.
int parse(Reader &r, vmx_gguf *g) {
    try {
        return parse_unguarded(r, g);
    } catch (const std::bad_alloc &) {
        vmx_set_error("GGUF: out of host memory while reading the header");
        return VMX_ENOMEM;
    } catch (const std::exception &e) {
        vmx_set_error("GGUF: %s", e.what());
        return VMX_EINVAL;
    }
}

int parse_unguarded(Reader &r, vmx_gguf *g) {
    char magic[4] = {0, 0, 0, 0};
    r.take(magic, 4);
    if (r.fail || memcmp(magic, "GGUF", 4) != 0) {
        vmx_set_error("not a GGUF file (bad magic [%u, %u, %u, %u])", (uint8_t)magic[0], (uint8_t)magic[1], (uint8_t)magic[2], (uint8_t)magic[3]);
        return VMX_EINVAL;
    }
    g->version = r.le<uint32_t>();
    const bool is64 = g->version >= 2;   // v1 used u32 counts, string and array lengths
    const uint64_t tensor_count = r.count(is64), kv_count = r.count(is64);
    if (r.fail || tensor_count > (1u << 24) || kv_count > (1u << 24)) {
        vmx_set_error("GGUF: truncated or implausible header (version %u)", g->version);
        return VMX_EINVAL;
    }
    for (uint64_t k = 0; k < kv_count; k++) {
        std::string key;
        r.str(is64, key);
        const uint32_t vt = r.le<uint32_t>();
        if (r.fail) break;
        MetaValue v;
        const int rc = read_value(r, vt, is64, v, 0);
        if (rc < 0) return VMX_EINVAL;
        if (rc == 0) break;
        if (!g->meta.count(key)) g->meta_keys.push_back(key);
        g->meta[key] = std::move(v);
    }
    for (uint64_t t = 0; t < tensor_count && !r.fail; t++) {
        TensorInfo ti;
        r.str(is64, ti.name);
        const uint32_t n_dims = r.le<uint32_t>();
        if (r.fail || n_dims > 8) { r.fail = true; break; }
        ti.shape.resize(n_dims);
        for (uint32_t d = 0; d < n_dims; d++) ti.shape[n_dims - 1 - d] = r.count(is64);   // reversed: 2-D weights read [out, in]
        ti.ggml_type = r.le<uint32_t>();
        ti.offset = r.le<uint64_t>();
        if (r.fail) break;
        uint32_t ts, bs;
        if (!type_geometry(ti.ggml_type, &ts, &bs)) {
            vmx_set_error("GGUF tensor '%s': unsupported GGML type id %u", ti.name.c_str(), ti.ggml_type);
            return VMX_EUNSUPPORTED;
        }
        uint64_t elems = 0;
        if (!element_count(ti, &elems) || elems % bs != 0 || elems / bs > UINT64_MAX / ts) {
            vmx_set_error("GGUF tensor '%s': element count overflows or is not a multiple of the %u-element block of type %u", ti.name.c_str(),
                          bs, ti.ggml_type);
            return VMX_EINVAL;
        }
        g->by_name[ti.name] = (int)g->tensors.size();
        g->tensors.push_back(std::move(ti));
    }
    if (r.fail) {
        vmx_set_error("GGUF: unexpected end of data at byte %zu", r.pos);
        return VMX_EINVAL;
    }
    g->alignment = 32;
    auto it = g->meta.find("general.alignment");
    if (it != g->meta.end()) {
        const MetaValue &v = it->second;
        if (v.type == 0 || v.type == 2 || v.type == 4) g->alignment = v.u;
        else if ((v.type == 1 || v.type == 3 || v.type == 5) && v.i >= 0) g->alignment = (uint64_t)v.i;
    }
    if (g->alignment == 0) {
        vmx_set_error("GGUF: general.alignment is 0");
        return VMX_EINVAL;
    }
    g->tensor_data_offset = (r.pos + g->alignment - 1) / g->alignment * g->alignment;
    return VMX_OK;
}

uint64_t tensor_bytes(const TensorInfo &t) {   // cannot overflow: checked when the header was parsed
    uint32_t ts = 0, bs = 1;
    type_geometry(t.ggml_type, &ts, &bs);
    uint64_t n = 1;
    element_count(t, &n);
    return n / bs * ts;
}
}  // namespace

extern "C" int vmx_gguf_parse(const void *data, size_t bytes, vmx_gguf **out) {
    VMX_REQUIRE(data && out, "gguf_parse: null pointer");
    vmx_gguf *g = new (std::nothrow) vmx_gguf();
    VMX_REQUIRE(g, "gguf_parse: out of host memory");
    Reader r;
    r.p = (const uint8_t *)data;
    r.n = bytes;
    const int rc = parse(r, g);
    if (rc) { delete g; return rc; }
    *out = g;
    return VMX_OK;
}

extern "C" int vmx_gguf_open(const char *path, vmx_gguf **out) {
    VMX_REQUIRE(path && out, "gguf_open: null pointer");
    FILE *f = fopen(path, "rb");
    VMX_REQUIRE(f, "gguf_open: cannot open '%s'", path);
    vmx_gguf *g = new (std::nothrow) vmx_gguf();
    if (!g) { fclose(f); vmx_set_error("gguf_open: out of host memory"); return VMX_ENOMEM; }
    Reader r;
    r.f = f;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long end = ftell(f);
        r.fsize = end > 0 ? (uint64_t)end : 0;
    }
    rewind(f);
    const int rc = parse(r, g);
    fclose(f);
    if (rc) { delete g; return rc; }
    g->path = path;
    *out = g;
    return VMX_OK;
}

extern "C" void vmx_gguf_close(vmx_gguf *g) { delete g; }
extern "C" int vmx_gguf_version(const vmx_gguf *g) { return g ? (int)g->version : -1; }
extern "C" uint64_t vmx_gguf_tensor_data_offset(const vmx_gguf *g) { return g ? g->tensor_data_offset : 0; }
extern "C" int vmx_gguf_tensor_count(const vmx_gguf *g) { return g ? (int)g->tensors.size() : -1; }
extern "C" int vmx_gguf_metadata_count(const vmx_gguf *g) { return g ? (int)g->meta_keys.size() : -1; }

extern "C" int vmx_gguf_find_tensor(const vmx_gguf *g, const char *name) {
    if (!g || !name) return -1;
    auto it = g->by_name.find(name);
    return it == g->by_name.end() ? -1 : it->second;
}

extern "C" int vmx_gguf_tensor_info(const vmx_gguf *g, int index, vmx_gguf_tensor *out) {
    VMX_REQUIRE(g && out, "gguf_tensor_info: null pointer");
    VMX_REQUIRE(index >= 0 && index < (int)g->tensors.size(), "gguf_tensor_info: index %d out of range", index);
    const TensorInfo &t = g->tensors[index];
    out->name = t.name.c_str();
    out->n_dims = (int)t.shape.size();
    for (int d = 0; d < 8; d++) out->shape[d] = d < out->n_dims ? t.shape[d] : 1;
    out->ggml_type = (int)t.ggml_type;
    out->offset = t.offset;
    out->n_bytes = tensor_bytes(t);
    return VMX_OK;
}

extern "C" int vmx_gguf_read_tensor(const vmx_gguf *g, int index, void *dst, size_t dst_bytes) {
    VMX_REQUIRE(g && dst, "gguf_read_tensor: null pointer");
    VMX_REQUIRE(index >= 0 && index < (int)g->tensors.size(), "gguf_read_tensor: index %d out of range", index);
    VMX_REQUIRE(!g->path.empty(), "gguf_read_tensor: the header was parsed from memory, not opened from a file");
    const uint64_t nb = tensor_bytes(g->tensors[index]);
    VMX_REQUIRE(dst_bytes >= nb, "gguf_read_tensor: '%s' needs %llu bytes, buffer has %zu", g->tensors[index].name.c_str(),
                (unsigned long long)nb, dst_bytes);
    FILE *f = fopen(g->path.c_str(), "rb");
    VMX_REQUIRE(f, "gguf_read_tensor: cannot reopen '%s'", g->path.c_str());
    int rc = VMX_OK;
    if (fseeko(f, (off_t)(g->tensor_data_offset + g->tensors[index].offset), SEEK_SET) != 0 || fread(dst, 1, nb, f) != nb) {
        vmx_set_error("gguf_read_tensor: short read of '%s'", g->tensors[index].name.c_str());
        rc = VMX_EINVAL;
    }
    fclose(f);
    return rc;
}

extern "C" const char *vmx_gguf_metadata_key(const vmx_gguf *g, int index) {
    if (!g || index < 0 || index >= (int)g->meta_keys.size()) return nullptr;
    return g->meta_keys[index].c_str();
}

namespace {
const MetaValue *find_meta(const vmx_gguf *g, const char *key, int array_index) {
    if (!g || !key) return nullptr;
    auto it = g->meta.find(key);
    if (it == g->meta.end()) return nullptr;
    const MetaValue *v = &it->second;
    if (array_index >= 0) {
        if (v->type != 9 || array_index >= (int)v->arr.size()) return nullptr;
        v = &v->arr[array_index];
    }
    return v;
}
}  // namespace

// value-type id of a key (9 = array; then *array_len / *elem_type are filled), or -1 when absent
extern "C" int vmx_gguf_metadata_type(const vmx_gguf *g, const char *key, uint64_t *array_len, int *elem_type) {
    const MetaValue *v = find_meta(g, key, -1);
    if (!v) return -1;
    if (array_len) *array_len = v->type == 9 ? v->arr.size() : 0;
    if (elem_type) *elem_type = v->type == 9 ? (int)v->elem_type : -1;
    return (int)v->type;
}

// array_index < 0: the value itself; >= 0: that element of an array value
extern "C" int vmx_gguf_metadata_int(const vmx_gguf *g, const char *key, int array_index, int64_t *out) {
    const MetaValue *v = find_meta(g, key, array_index);
    VMX_REQUIRE(v && out, "gguf_metadata_int: key '%s' not found", key ? key : "(null)");
    VMX_REQUIRE(v->type <= 5 || v->type == 7 || v->type == 10 || v->type == 11, "gguf_metadata_int: '%s' is not an integer (type %u)", key,
                v->type);
    *out = v->i;
    return VMX_OK;
}

extern "C" int vmx_gguf_metadata_float(const vmx_gguf *g, const char *key, int array_index, double *out) {
    const MetaValue *v = find_meta(g, key, array_index);
    VMX_REQUIRE(v && out, "gguf_metadata_float: key '%s' not found", key ? key : "(null)");
    if (v->type == 6 || v->type == 12) *out = v->f;
    else if (v->type <= 5 || v->type == 10 || v->type == 11) *out = (v->type == 10) ? (double)v->u : (double)v->i;
    else { vmx_set_error("gguf_metadata_float: '%s' is not numeric (type %u)", key, v->type); return VMX_EINVAL; }
    return VMX_OK;
}

extern "C" const char *vmx_gguf_metadata_string(const vmx_gguf *g, const char *key, int array_index) {
    const MetaValue *v = find_meta(g, key, array_index);
    return (v && v->type == 8) ? v->s.c_str() : nullptr;
}

// HF-style module prefix ("model.layers.3.self_attn.q_proj", "model.norm", ...) -> llama.cpp tensor-name prefix
// ("blk.3.attn_q", "output_norm", ...), candidate `which` in priority order (post_attention_layernorm has two).
// Returns the length written (without the terminator), 0 when there is no such candidate.
extern "C" int vmx_gguf_llama_cpp_name(const char *hf_prefix, int which, char *out, size_t out_bytes) {
    if (!hf_prefix || !out || out_bytes == 0 || which < 0) return 0;
    const std::string p(hf_prefix);
    static const char *top[][2] = {{"model.embed_tokens", "token_embd"},
                                   {"model.norm", "output_norm"},
                                   {"lm_head", "output"},
                                   {"model.rope_freqs", "rope_freqs"},
                                   {"model.embed_tokens_per_layer", "per_layer_token_embd"},
                                   {"model.per_layer_model_projection", "per_layer_model_proj"},
                                   {"model.per_layer_projection_norm", "per_layer_proj_norm"}};
    std::string res;
    for (auto &t : top)
        if (p == t[0]) { if (which == 0) res = t[1]; goto done; }
    {
        const std::string pre = "model.layers.";
        if (p.compare(0, pre.size(), pre) != 0) return 0;
        const size_t dot = p.find('.', pre.size());
        if (dot == std::string::npos || dot == pre.size()) return 0;
        const std::string idx = p.substr(pre.size(), dot - pre.size()), sub = p.substr(dot + 1);
        for (char c : idx)
            if (c < '0' || c > '9') return 0;
        static const char *subs[][3] = {{"self_attn.q_proj", "attn_q", nullptr},
                                        {"self_attn.k_proj", "attn_k", nullptr},
                                        {"self_attn.v_proj", "attn_v", nullptr},
                                        {"self_attn.o_proj", "attn_output", nullptr},
                                        {"self_attn.q_norm", "attn_q_norm", nullptr},
                                        {"self_attn.k_norm", "attn_k_norm", nullptr},
                                        {"mlp.gate_proj", "ffn_gate", nullptr},
                                        {"mlp.up_proj", "ffn_up", nullptr},
                                        {"mlp.down_proj", "ffn_down", nullptr},
                                        {"per_layer_input_gate", "inp_gate", nullptr},
                                        {"per_layer_projection", "proj", nullptr},
                                        {"post_per_layer_input_norm", "post_norm", nullptr},
                                        {"layer_scalar", "layer_output_scale", nullptr},
                                        {"input_layernorm", "attn_norm", nullptr},
                                        {"post_attention_layernorm", "post_attention_norm", "ffn_norm"},
                                        {"pre_feedforward_layernorm", "ffn_norm", nullptr},
                                        {"post_feedforward_layernorm", "post_ffw_norm", nullptr}};
        for (auto &s : subs)
            if (sub == s[0]) {
                const char *c = which == 0 ? s[1] : (which == 1 ? s[2] : nullptr);
                if (c) res = "blk." + idx + "." + c;
                break;
            }
    }
done:
    if (res.empty() || res.size() + 1 > out_bytes) return 0;
    memcpy(out, res.c_str(), res.size() + 1);
    return (int)res.size();
}
