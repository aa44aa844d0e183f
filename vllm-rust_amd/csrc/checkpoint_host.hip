// Host-side checkpoint plumbing (no device code): the safetensors container and the detection of a checkpoint's
// quantization method, so that real AWQ / GPTQ / EXL3 checkpoints can feed the executors instead of synthetic tensors.
//   * quantization detection follows crates/core/src/quantization/detection.rs:82-314: config.json's
//     "quantization_config" first (quant_method -> method, bits, group_size, desc_act, activation_scheme), then
//     quantize_config.json (GPTQ form: bits + group_size, "gptq_marlin" routed to Marlin; AWQ form: w_bit / q_group_size),
//     then the presence of *.gguf files;
//   * the safetensors reader is the container the reference's weight loader memory-maps
//     (crates/core/src/quantization/weight_loader.rs:597-714 reads "<prefix>.qweight / .qzeros / .scales" from it):
//     u64 little-endian header length, a JSON object {name: {dtype, shape, data_offsets}, "__metadata__": {...}}, then
//     the tensor bytes.  Every length and offset is checked against the file before it is used.
#include "vmx_internal.h"

#include <dirent.h>
#include <stdlib.h>

#include <exception>
#include <map>
#include <memory>
#include <new>
#include <ctype.h>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------- a small JSON reader (RFC 8259 values, UTF-8 kept as is)
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    bool num_is_int = false;
    int64_t inum = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;   // file order
    const Json *get(const char *key) const {
        if (kind != Obj) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JsonParser {
    const char *p, *end;
    std::string err;
    int depth = 0;
    bool fail(const char *what) {
        if (err.empty()) err = std::string(what) + " at byte " + std::to_string((long long)(p - start));
        return false;
    }
    const char *start;
    JsonParser(const char *s, size_t n) : p(s), end(s + n), start(s) {}
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool lit(const char *s) {
        const size_t n = strlen(s);
        if ((size_t)(end - p) < n || memcmp(p, s, n) != 0) return fail("unexpected token");
        p += n;
        return true;
    }
    static void utf8(std::string &o, uint32_t c) {
        if (c < 0x80) o += (char)c;
        else if (c < 0x800) { o += (char)(0xC0 | (c >> 6)); o += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { o += (char)(0xE0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
        else { o += (char)(0xF0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3F)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t &v) {
        if (end - p < 4) return fail("truncated \\u escape");
        v = 0;
        for (int i = 0; i < 4; i++) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string &out) {
        if (p >= end || *p != '"') return fail("expected a string");
        p++;
        while (p < end && *p != '"') {
            if ((unsigned char)*p < 0x20) return fail("control character in a string");
            if (*p != '\\') { out += *p++; continue; }
            if (++p >= end) return fail("truncated escape");
            const char c = *p++;
            switch (c) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    uint32_t v;
                    if (!hex4(v)) return false;
                    if (v >= 0xD800 && v < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {   // surrogate pair
                        p += 2;
                        uint32_t lo;
                        if (!hex4(lo)) return false;
                        if (lo >= 0xDC00 && lo < 0xE000) v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(out, v);
                    break;
                }
                default: return fail("unknown escape");
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool value(Json &v) {
        if (++depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end of JSON");
        bool ok = true;
        const char c = *p;
        if (c == '{') {
            v.kind = Json::Obj;
            p++;
            ws();
            if (p < end && *p == '}') { p++; }
            else {
                for (;;) {
                    ws();
                    std::string key;
                    if (!string(key)) { ok = false; break; }
                    ws();
                    if (p >= end || *p != ':') { ok = fail("expected ':'"); break; }
                    p++;
                    v.obj.emplace_back(std::move(key), Json());
                    if (!value(v.obj.back().second)) { ok = false; break; }
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == '}') { p++; break; }
                    ok = fail("expected ',' or '}'");
                    break;
                }
            }
        } else if (c == '[') {
            v.kind = Json::Arr;
            p++;
            ws();
            if (p < end && *p == ']') { p++; }
            else {
                for (;;) {
                    v.arr.emplace_back();
                    if (!value(v.arr.back())) { ok = false; break; }
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == ']') { p++; break; }
                    ok = fail("expected ',' or ']'");
                    break;
                }
            }
        } else if (c == '"') {
            v.kind = Json::Str;
            ok = string(v.str);
        } else if (c == 't') { v.kind = Json::Bool; v.b = true; ok = lit("true"); }
        else if (c == 'f') { v.kind = Json::Bool; v.b = false; ok = lit("false"); }
        else if (c == 'n') { v.kind = Json::Null; ok = lit("null"); }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            const char *s = p;
            if (*p == '-') p++;
            bool is_int = true;
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
                if (*p == '.' || *p == 'e' || *p == 'E') is_int = false;
                p++;
            }
            const std::string t(s, p);
            char *e = nullptr;
            v.kind = Json::Num;
            v.num = strtod(t.c_str(), &e);
            if (e == t.c_str() || *e) ok = fail("malformed number");
            if (ok && is_int && t.size() <= 19) {
                v.num_is_int = true;
                v.inum = strtoll(t.c_str(), nullptr, 10);
            }
        } else {
            ok = fail("unexpected character");
        }
        depth--;
        return ok;
    }
    bool parse(Json &root) {
        if (!value(root)) return false;
        ws();
        if (p != end) return fail("trailing characters after the JSON value");
        return true;
    }
};

bool json_int(const Json *j, int64_t *out) {
    if (!j || j->kind != Json::Num) return false;
    if (j->num_is_int) { *out = j->inum; return true; }
    if (j->num != (double)(int64_t)j->num) return false;
    *out = (int64_t)j->num;
    return true;
}

// ---------------------------------------------------------------- quantization detection
int method_from_name(const std::string &m, const Json *cfg) {
    struct E { const char *name; int id; };
    static const E tab[] = {{"fp8", VMX_QM_FP8},          {"gptq", VMX_QM_GPTQ},           {"awq", VMX_QM_AWQ},
                            {"awq_triton", VMX_QM_AWQ},   {"awq_marlin", VMX_QM_AWQ_MARLIN}, {"gptq_marlin", VMX_QM_MARLIN},
                            {"exl3", VMX_QM_EXL3}};
    for (const E &e : tab)
        if (m == e.name) return e.id;
    // the methods the reference names but this library has no kernels for (bitsandbytes, compressed-tensors, modelopt, ...)
    static const char *known[] = {"bitsandbytes", "squeezellm", "compressed-tensors", "torchao", "experts_int8", "moe_wna16", "fbgemm_fp8",
                                  "ptpc_fp8", "mxfp4", "cpu_awq", "cpu_wna16", "inc", "auto-round", "fp_quant", "quark"};
    for (const char *k : known)
        if (m == k) return VMX_QM_OTHER;
    if (m == "modelopt") {
        // detection.rs:143-160: recognised only with quant_algo MXFP8 / FP8 / FP8_PER_CHANNEL_PER_TOKEN / FP8_PB_WO / NVFP4
        // (case-insensitive), nested under "quantization" or flat; anything else is "not detected" and the search goes on
        const Json *algo = nullptr;
        if (cfg) {
            const Json *nested = cfg->get("quantization");
            if (nested && nested->kind == Json::Obj) algo = nested->get("quant_algo");
            if (!algo) algo = cfg->get("quant_algo");
        }
        if (!algo || algo->kind != Json::Str) return VMX_QM_NONE;
        std::string a = algo->str;
        for (char &ch : a) ch = (char)toupper((unsigned char)ch);
        for (const char *ok : {"MXFP8", "FP8", "FP8_PER_CHANNEL_PER_TOKEN", "FP8_PB_WO", "NVFP4"})
            if (a == ok) return VMX_QM_OTHER;
        return VMX_QM_NONE;
    }
    return VMX_QM_NONE;
}

void fill_common(const Json &q, vmx_quant_info *out) {
    int64_t v;
    if (json_int(q.get("bits"), &v)) out->bits = (int)v;
    if (json_int(q.get("group_size"), &v)) out->group_size = (int)v;
    const Json *d = q.get("desc_act");
    if (d && d->kind == Json::Bool) out->desc_act = d->b ? 1 : 0;
    const Json *a = q.get("activation_scheme");
    if (a && a->kind == Json::Str) snprintf(out->activation_scheme, sizeof(out->activation_scheme), "%s", a->str.c_str());
}

// config.json -> "quantization_config" (detection.rs:101-176): 1 found, 0 not there / method not recognised
int detect_config_json(const Json &root, vmx_quant_info *out) {
    const Json *q = root.get("quantization_config");
    if (!q || q->kind != Json::Obj) return 0;
    const Json *m = q->get("quant_method");
    if (!m || m->kind != Json::Str) return 0;
    const int id = method_from_name(m->str, q);
    if (id == VMX_QM_NONE) return 0;
    out->method = id;
    snprintf(out->method_name, sizeof(out->method_name), "%s", m->str.c_str());
    fill_common(*q, out);
    return 1;
}

// quantize_config.json (detection.rs:178-223): GPTQ form first (bits and group_size), then AWQ form (w_bit)
int detect_quantize_config(const Json &root, vmx_quant_info *out) {
    if (root.kind != Json::Obj) return 0;
    int64_t bits, gs;
    if (json_int(root.get("bits"), &bits) && json_int(root.get("group_size"), &gs)) {
        const Json *m = root.get("quant_method");
        const bool marlin = m && m->kind == Json::Str && m->str == "gptq_marlin";
        out->method = marlin ? VMX_QM_MARLIN : VMX_QM_GPTQ;
        snprintf(out->method_name, sizeof(out->method_name), "%s", marlin ? "gptq_marlin" : "gptq");
        fill_common(root, out);
        return 1;
    }
    int64_t wbit;
    if (json_int(root.get("w_bit"), &wbit)) {
        out->method = VMX_QM_AWQ;
        snprintf(out->method_name, sizeof(out->method_name), "awq");
        out->bits = (int)wbit;
        int64_t g;
        if (json_int(root.get("q_group_size"), &g)) out->group_size = (int)g;
        return 1;
    }
    return 0;
}

bool read_file(const std::string &path, std::string &out, size_t limit) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long n = ftell(f);
        if (n >= 0 && (size_t)n <= limit) {
            rewind(f);
            out.resize((size_t)n);
            ok = n == 0 || fread(&out[0], 1, (size_t)n, f) == (size_t)n;
        }
    }
    fclose(f);
    return ok;
}

void quant_info_reset(vmx_quant_info *out) {
    memset(out, 0, sizeof(*out));
    out->method = VMX_QM_NONE;
    out->bits = -1;
    out->group_size = -1;
    out->desc_act = -1;
}

int detect_json_texts(const char *config_json, const char *quantize_config_json, vmx_quant_info *out) {
    quant_info_reset(out);
    if (config_json) {
        Json root;
        JsonParser jp(config_json, strlen(config_json));
        if (jp.parse(root) && detect_config_json(root, out)) return VMX_OK;   // an unreadable file is "no answer", as in the reference
        quant_info_reset(out);
    }
    if (quantize_config_json) {
        Json root;
        JsonParser jp(quantize_config_json, strlen(quantize_config_json));
        if (jp.parse(root) && detect_quantize_config(root, out)) return VMX_OK;
        quant_info_reset(out);
    }
    return VMX_OK;
}

// ---------------------------------------------------------------- safetensors
struct StTensor {
    std::string name, dtype;
    std::vector<int64_t> shape;
    uint64_t begin = 0, end = 0;   // relative to the data section
};

int dtype_bytes(const std::string &d) {
    struct E { const char *n; int b; };
    static const E tab[] = {{"BOOL", 1}, {"U8", 1},  {"I8", 1},  {"F8_E4M3", 1}, {"F8_E5M2", 1}, {"I16", 2}, {"U16", 2}, {"F16", 2},
                            {"BF16", 2}, {"I32", 4}, {"U32", 4}, {"F32", 4},     {"F64", 8},     {"I64", 8}, {"U64", 8}};
    for (const E &e : tab)
        if (d == e.n) return e.b;
    return 0;
}

}  // namespace

struct vmx_safetensors {
    std::string path;
    uint64_t data_offset = 0, data_bytes = 0;
    std::vector<StTensor> tensors;            // file (header) order
    std::map<std::string, int> by_name;
    std::vector<std::pair<std::string, std::string>> metadata;
};

namespace {
int st_open(const char *path, vmx_safetensors *h) {
    FILE *f = fopen(path, "rb");
    VMX_REQUIRE(f, "safetensors_open: cannot open '%s'", path);
    std::unique_ptr<FILE, int (*)(FILE *)> guard(f, fclose);
    VMX_REQUIRE(fseek(f, 0, SEEK_END) == 0, "safetensors_open: cannot seek in '%s'", path);
    const long fsz = ftell(f);
    rewind(f);
    uint8_t lenb[8];
    VMX_REQUIRE(fsz >= 8 && fread(lenb, 1, 8, f) == 8, "safetensors: '%s' is shorter than the 8-byte header length", path);
    uint64_t hlen = 0;
    for (int i = 0; i < 8; i++) hlen |= (uint64_t)lenb[i] << (8 * i);
    VMX_REQUIRE(hlen >= 2 && hlen <= (uint64_t)100 * 1024 * 1024 && hlen <= (uint64_t)fsz - 8,
                "safetensors: header length %llu does not fit the file (%ld bytes)", (unsigned long long)hlen, fsz);
    std::string text((size_t)hlen, '\0');
    VMX_REQUIRE(fread(&text[0], 1, (size_t)hlen, f) == (size_t)hlen, "safetensors: truncated header in '%s'", path);
    Json root;
    JsonParser jp(text.data(), text.size());
    if (!jp.parse(root) || root.kind != Json::Obj) {
        vmx_set_error("safetensors: header of '%s' is not a JSON object (%s)", path, jp.err.c_str());
        return VMX_EINVAL;
    }
    h->path = path;
    h->data_offset = 8 + hlen;
    h->data_bytes = (uint64_t)fsz - h->data_offset;
    for (const auto &kv : root.obj) {
        if (kv.first == "__metadata__") {
            if (kv.second.kind == Json::Obj)
                for (const auto &m : kv.second.obj)
                    if (m.second.kind == Json::Str) h->metadata.emplace_back(m.first, m.second.str);
            continue;
        }
        const Json &t = kv.second;
        const Json *dt = t.get("dtype"), *sh = t.get("shape"), *off = t.get("data_offsets");
        VMX_REQUIRE(dt && dt->kind == Json::Str && sh && sh->kind == Json::Arr && off && off->kind == Json::Arr && off->arr.size() == 2,
                    "safetensors: tensor '%s' lacks dtype / shape / data_offsets", kv.first.c_str());
        StTensor st;
        st.name = kv.first;
        st.dtype = dt->str;
        const int eb = dtype_bytes(st.dtype);
        VMX_REQUIRE(eb > 0, "safetensors: tensor '%s' has unknown dtype '%s'", st.name.c_str(), st.dtype.c_str());
        VMX_REQUIRE(sh->arr.size() <= 8, "safetensors: tensor '%s' has %zu dimensions", st.name.c_str(), sh->arr.size());
        uint64_t n = 1;
        for (const Json &d : sh->arr) {
            int64_t v;
            VMX_REQUIRE(json_int(&d, &v) && v >= 0, "safetensors: tensor '%s' has a non-integer dimension", st.name.c_str());
            VMX_REQUIRE(v == 0 || n <= UINT64_MAX / (uint64_t)v, "safetensors: tensor '%s': element count overflows", st.name.c_str());
            n *= (uint64_t)v;
            st.shape.push_back(v);
        }
        int64_t b, e;
        VMX_REQUIRE(json_int(&off->arr[0], &b) && json_int(&off->arr[1], &e) && b >= 0 && e >= b && (uint64_t)e <= h->data_bytes,
                    "safetensors: tensor '%s' has data_offsets outside the file", st.name.c_str());
        VMX_REQUIRE(n <= UINT64_MAX / (uint64_t)eb && (uint64_t)(e - b) == n * (uint64_t)eb,
                    "safetensors: tensor '%s': %lld bytes of data for shape x dtype = %llu bytes", st.name.c_str(), (long long)(e - b),
                    (unsigned long long)(n * (uint64_t)eb));
        st.begin = (uint64_t)b;
        st.end = (uint64_t)e;
        VMX_REQUIRE(!h->by_name.count(st.name), "safetensors: tensor '%s' appears twice", st.name.c_str());
        h->by_name[st.name] = (int)h->tensors.size();
        h->tensors.push_back(std::move(st));
    }
    return VMX_OK;
}
}  // namespace

extern "C" int vmx_quant_detect_json(const char *config_json, const char *quantize_config_json, vmx_quant_info *out) {
    VMX_REQUIRE(out, "quant_detect_json: null output");
    try {
        return detect_json_texts(config_json, quantize_config_json, out);
    } catch (const std::bad_alloc &) {
        vmx_set_error("quant_detect_json: out of host memory");
        return VMX_ENOMEM;
    }
}

extern "C" int vmx_quant_detect_dir(const char *model_dir, vmx_quant_info *out) {
    VMX_REQUIRE(model_dir && out, "quant_detect_dir: null pointer");
    try {
        const std::string dir(model_dir);
        std::string cfg, qcfg;
        const bool has_cfg = read_file(dir + "/config.json", cfg, (size_t)64 << 20);
        const bool has_q = read_file(dir + "/quantize_config.json", qcfg, (size_t)64 << 20);
        if (int rc = detect_json_texts(has_cfg ? cfg.c_str() : nullptr, has_q ? qcfg.c_str() : nullptr, out)) return rc;
        if (out->method != VMX_QM_NONE) return VMX_OK;
        if (DIR *d = opendir(model_dir)) {   // detection.rs:97-104,225-236
            while (const dirent *e = readdir(d)) {
                const size_t n = strlen(e->d_name);
                if (n > 5 && strcmp(e->d_name + n - 5, ".gguf") == 0) {
                    out->method = VMX_QM_GGUF;
                    snprintf(out->method_name, sizeof(out->method_name), "gguf");
                    break;
                }
            }
            closedir(d);
        }
        return VMX_OK;
    } catch (const std::bad_alloc &) {
        vmx_set_error("quant_detect_dir: out of host memory");
        return VMX_ENOMEM;
    }
}

extern "C" int vmx_safetensors_open(const char *path, vmx_safetensors **out) {
    VMX_REQUIRE(path && out, "safetensors_open: null pointer");
    vmx_safetensors *h = new (std::nothrow) vmx_safetensors();
    VMX_REQUIRE(h, "safetensors_open: out of host memory");
    int rc;
    try {
        rc = st_open(path, h);
    } catch (const std::bad_alloc &) {
        vmx_set_error("safetensors_open: out of host memory");
        rc = VMX_ENOMEM;
    } catch (const std::exception &e) {
        vmx_set_error("safetensors_open: %s", e.what());
        rc = VMX_EINVAL;
    }
    if (rc) { delete h; return rc; }
    *out = h;
    return VMX_OK;
}

extern "C" void vmx_safetensors_close(vmx_safetensors *h) { delete h; }
extern "C" int vmx_safetensors_tensor_count(const vmx_safetensors *h) { return h ? (int)h->tensors.size() : -1; }

extern "C" int vmx_safetensors_find(const vmx_safetensors *h, const char *name) {
    if (!h || !name) return -1;
    const auto it = h->by_name.find(name);
    return it == h->by_name.end() ? -1 : it->second;
}

extern "C" int vmx_safetensors_tensor_info(const vmx_safetensors *h, int index, vmx_safetensors_tensor *out) {
    VMX_REQUIRE(h && out, "safetensors_tensor_info: null pointer");
    VMX_REQUIRE(index >= 0 && index < (int)h->tensors.size(), "safetensors_tensor_info: index %d out of range", index);
    const StTensor &t = h->tensors[index];
    memset(out, 0, sizeof(*out));
    out->name = t.name.c_str();
    out->dtype = t.dtype.c_str();
    out->element_bytes = dtype_bytes(t.dtype);
    out->n_dims = (int)t.shape.size();
    for (size_t i = 0; i < t.shape.size(); i++) out->shape[i] = t.shape[i];
    out->file_offset = h->data_offset + t.begin;
    out->n_bytes = t.end - t.begin;
    return VMX_OK;
}

extern "C" int vmx_safetensors_read_tensor(const vmx_safetensors *h, int index, void *dst_host, size_t dst_bytes) {
    VMX_REQUIRE(h && dst_host, "safetensors_read_tensor: null pointer");
    VMX_REQUIRE(index >= 0 && index < (int)h->tensors.size(), "safetensors_read_tensor: index %d out of range", index);
    const StTensor &t = h->tensors[index];
    const uint64_t nb = t.end - t.begin;
    VMX_REQUIRE(dst_bytes >= nb, "safetensors_read_tensor: '%s' needs %llu bytes, buffer has %zu", t.name.c_str(), (unsigned long long)nb, dst_bytes);
    FILE *f = fopen(h->path.c_str(), "rb");
    VMX_REQUIRE(f, "safetensors_read_tensor: cannot reopen '%s'", h->path.c_str());
    bool ok = fseek(f, (long)(h->data_offset + t.begin), SEEK_SET) == 0 && (nb == 0 || fread(dst_host, 1, (size_t)nb, f) == (size_t)nb);
    fclose(f);
    VMX_REQUIRE(ok, "safetensors_read_tensor: short read of '%s'", t.name.c_str());
    return VMX_OK;
}

extern "C" const char *vmx_safetensors_metadata(const vmx_safetensors *h, const char *key) {
    if (!h || !key) return nullptr;
    for (const auto &kv : h->metadata)
        if (kv.first == key) return kv.second.c_str();
    return nullptr;
}
