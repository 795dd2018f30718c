#include "model_onnx.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <set>

#include "../../include/bnhip.h"

namespace bnhip {

namespace {

// ---------------------------------------------------------------------------------------------- protobuf wire reader
struct PB {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    PB(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    bool done() const { return !ok || p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) { ok = false; return 0; }
            uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    // reads one field header + payload; for wire type 2 sets data/len, for 0 sets val, for 1 / 5 sets data/len (8 / 4 bytes)
    bool next(uint32_t* field, int* wire, uint64_t* val, const uint8_t** data, size_t* len) {
        if (done()) return false;
        uint64_t key = varint();
        if (!ok) return false;
        *field = (uint32_t)(key >> 3); *wire = (int)(key & 7);
        *val = 0; *data = nullptr; *len = 0;
        switch (*wire) {
            case 0: *val = varint(); return ok;
            case 1: if ((size_t)(end - p) < 8) { ok = false; return false; } *data = p; *len = 8; p += 8; return true;
            case 5: if ((size_t)(end - p) < 4) { ok = false; return false; } *data = p; *len = 4; p += 4; return true;
            case 2: {
                uint64_t l = varint();
                if (!ok || l > (uint64_t)(end - p)) { ok = false; return false; }
                *data = p; *len = (size_t)l; p += l;
                return true;
            }
            default: ok = false; return false;         // groups: not used by ONNX
        }
    }
};

struct OTensor {
    std::string name;
    std::vector<int64_t> dims;
    int dtype = 0;
    const uint8_t* raw = nullptr; size_t raw_len = 0;
    std::vector<float> f; std::vector<int64_t> i64; std::vector<double> f64; std::vector<int32_t> i32;
    bool external = false;
};
struct OAttr {
    std::string name;
    float f = 0; int64_t i = 0; std::string s;
    std::vector<float> floats; std::vector<int64_t> ints;
    OTensor t; bool has_t = false;
};
struct ONode {
    std::string op, name, domain;
    std::vector<std::string> in, out;
    std::vector<OAttr> attrs;
    const OAttr* attr(const char* n) const { for (auto& a : attrs) if (a.name == n) return &a; return nullptr; }
    int64_t ai(const char* n, int64_t def) const { auto a = attr(n); return a ? a->i : def; }
    float af(const char* n, float def) const { auto a = attr(n); return a ? a->f : def; }
};
struct OValue { std::string name; std::vector<int64_t> dims; int elem = 0; bool has_shape = false; };

float f32_at(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }

bool parse_tensor(const uint8_t* b, size_t n, OTensor* t) {
    PB pb(b, n);
    uint32_t f; int w; uint64_t v; const uint8_t* d; size_t l;
    while (pb.next(&f, &w, &v, &d, &l)) {
        switch (f) {
            case 1: if (w == 0) t->dims.push_back((int64_t)v); else if (w == 2) { PB q(d, l); while (!q.done()) t->dims.push_back((int64_t)q.varint()); if (!q.ok) return false; } break;
            case 2: t->dtype = (int)v; break;
            case 4: if (w == 5) t->f.push_back(f32_at(d)); else if (w == 2) { if (l % 4) return false; for (size_t k = 0; k < l; k += 4) t->f.push_back(f32_at(d + k)); } break;
            case 5: if (w == 0) t->i32.push_back((int32_t)v); else if (w == 2) { PB q(d, l); while (!q.done()) t->i32.push_back((int32_t)q.varint()); if (!q.ok) return false; } break;
            case 7: if (w == 0) t->i64.push_back((int64_t)v); else if (w == 2) { PB q(d, l); while (!q.done()) t->i64.push_back((int64_t)q.varint()); if (!q.ok) return false; } break;
            case 8: if (w == 2) t->name.assign((const char*)d, l); break;
            case 9: if (w == 2) { t->raw = d; t->raw_len = l; } break;
            case 10: if (w == 1) { double x; memcpy(&x, d, 8); t->f64.push_back(x); } else if (w == 2) { if (l % 8) return false; for (size_t k = 0; k < l; k += 8) { double x; memcpy(&x, d + k, 8); t->f64.push_back(x); } } break;
            case 13: t->external = true; break;
            case 14: if (v == 1) t->external = true; break;
            default: break;
        }
    }
    if (!pb.ok) return false;
    // Dimensions are indexed with int arithmetic downstream: every one must be in [0, INT_MAX] and the element count must
    // not overflow (a negative or wrapped dim would let a tensor whose payload is empty claim any shape).
    if (t->dims.size() > 8) return false;
    uint64_t cnt = 1;
    for (int64_t dv : t->dims) {
        if (dv < 0 || dv > 0x7fffffffLL) return false;
        cnt *= (uint64_t)dv;
        if (cnt > (1ull << 31)) return false;
    }
    return true;
}

bool parse_attr(const uint8_t* b, size_t n, OAttr* a) {
    PB pb(b, n);
    uint32_t f; int w; uint64_t v; const uint8_t* d; size_t l;
    while (pb.next(&f, &w, &v, &d, &l)) {
        switch (f) {
            case 1: if (w == 2) a->name.assign((const char*)d, l); break;
            case 2: if (w == 5) a->f = f32_at(d); break;
            case 3: if (w == 0) a->i = (int64_t)v; break;
            case 4: if (w == 2) a->s.assign((const char*)d, l); break;
            case 5: if (w == 2) { if (!parse_tensor(d, l, &a->t)) return false; a->has_t = true; } break;
            case 7: if (w == 5) a->floats.push_back(f32_at(d)); else if (w == 2) { if (l % 4) return false; for (size_t k = 0; k < l; k += 4) a->floats.push_back(f32_at(d + k)); } break;
            case 8: if (w == 0) a->ints.push_back((int64_t)v); else if (w == 2) { PB q(d, l); while (!q.done()) a->ints.push_back((int64_t)q.varint()); if (!q.ok) return false; } break;
            default: break;
        }
    }
    return pb.ok;
}

bool parse_node(const uint8_t* b, size_t n, ONode* nd) {
    PB pb(b, n);
    uint32_t f; int w; uint64_t v; const uint8_t* d; size_t l;
    while (pb.next(&f, &w, &v, &d, &l)) {
        if (w != 2) continue;
        switch (f) {
            case 1: nd->in.emplace_back((const char*)d, l); break;
            case 2: nd->out.emplace_back((const char*)d, l); break;
            case 3: nd->name.assign((const char*)d, l); break;
            case 4: nd->op.assign((const char*)d, l); break;
            case 5: { OAttr a; if (!parse_attr(d, l, &a)) return false; nd->attrs.push_back(std::move(a)); break; }
            case 7: nd->domain.assign((const char*)d, l); break;
            default: break;
        }
    }
    return pb.ok;
}

// ValueInfoProto -> name + tensor shape (dim_param / missing dims become -1)
bool parse_value_info(const uint8_t* b, size_t n, OValue* vi) {
    PB pb(b, n);
    uint32_t f; int w; uint64_t v; const uint8_t* d; size_t l;
    while (pb.next(&f, &w, &v, &d, &l)) {
        if (f == 1 && w == 2) vi->name.assign((const char*)d, l);
        else if (f == 2 && w == 2) {                       // TypeProto
            PB tp(d, l);
            uint32_t f2; int w2; uint64_t v2; const uint8_t* d2; size_t l2;
            while (tp.next(&f2, &w2, &v2, &d2, &l2)) {
                if (f2 != 1 || w2 != 2) continue;          // tensor_type
                PB tt(d2, l2);
                uint32_t f3; int w3; uint64_t v3; const uint8_t* d3; size_t l3;
                while (tt.next(&f3, &w3, &v3, &d3, &l3)) {
                    if (f3 == 1 && w3 == 0) vi->elem = (int)v3;
                    else if (f3 == 2 && w3 == 2) {         // TensorShapeProto
                        vi->has_shape = true;
                        PB sp(d3, l3);
                        uint32_t f4; int w4; uint64_t v4; const uint8_t* d4; size_t l4;
                        while (sp.next(&f4, &w4, &v4, &d4, &l4)) {
                            if (f4 != 1 || w4 != 2) continue;      // Dimension
                            int64_t dim = -1;
                            PB dp(d4, l4);
                            uint32_t f5; int w5; uint64_t v5; const uint8_t* d5; size_t l5;
                            while (dp.next(&f5, &w5, &v5, &d5, &l5)) if (f5 == 1 && w5 == 0) dim = (int64_t)v5;
                            if (!dp.ok) return false;
                            vi->dims.push_back(dim);
                        }
                        if (!sp.ok) return false;
                    }
                }
                if (!tt.ok) return false;
            }
            if (!tp.ok) return false;
        }
    }
    return pb.ok;
}

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; do { man <<= 1; e++; } while (!(man & 0x400)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13); }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

// element count of validated dims (parse_tensor bounds them); anything out of range maps to a count no payload can match
size_t numel(const std::vector<int64_t>& d) {
    uint64_t n = 1;
    for (auto v : d) {
        if (v < 0 || v > 0x7fffffffLL) return SIZE_MAX / 16;
        n *= (uint64_t)v;
        if (n > (1ull << 31)) return SIZE_MAX / 16;
    }
    return (size_t)n;
}

// initializer -> float vector (FLOAT, FLOAT16, DOUBLE) ; false when the dtype is not a float type or sizes disagree
bool tensor_floats(const OTensor& t, std::vector<float>* out) {
    const size_t n = numel(t.dims);
    out->clear();
    if (t.dtype == 1) {
        if (t.raw) { if (t.raw_len != n * 4) return false; out->resize(n); if (n) memcpy(out->data(), t.raw, n * 4); }
        else { if (t.f.size() != n) return false; *out = t.f; }
        return true;
    }
    if (t.dtype == 10) {
        if (t.raw) { if (t.raw_len != n * 2) return false; out->resize(n); for (size_t k = 0; k < n; k++) { uint16_t h; memcpy(&h, t.raw + 2 * k, 2); (*out)[k] = half_to_float(h); } }
        else { if (t.i32.size() != n) return false; out->resize(n); for (size_t k = 0; k < n; k++) (*out)[k] = half_to_float((uint16_t)t.i32[k]); }
        return true;
    }
    if (t.dtype == 11) {
        if (t.raw) { if (t.raw_len != n * 8) return false; out->resize(n); for (size_t k = 0; k < n; k++) { double x; memcpy(&x, t.raw + 8 * k, 8); (*out)[k] = (float)x; } }
        else { if (t.f64.size() != n) return false; out->resize(n); for (size_t k = 0; k < n; k++) (*out)[k] = (float)t.f64[k]; }
        return true;
    }
    return false;
}
bool tensor_ints(const OTensor& t, std::vector<int64_t>* out) {
    const size_t n = numel(t.dims);
    out->clear();
    if (t.dtype == 7) {
        if (t.raw) { if (t.raw_len != n * 8) return false; out->resize(n); if (n) memcpy(out->data(), t.raw, n * 8); }
        else { if (t.i64.size() != n) return false; *out = t.i64; }
        return true;
    }
    if (t.dtype == 6) {
        if (t.raw) { if (t.raw_len != n * 4) return false; out->resize(n); for (size_t k = 0; k < n; k++) { int32_t x; memcpy(&x, t.raw + 4 * k, 4); (*out)[k] = x; } }
        else { if (t.i32.size() != n) return false; out->assign(t.i32.begin(), t.i32.end()); }
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------- constant folding
// Exporters leave small all-constant subgraphs behind - torch writes the pads of F.pad as ConstantOfShape -> Concat -> Reshape ->
// Slice -> Transpose -> Reshape -> Cast on int64 constants - which an ONNX runtime folds at load time.  So does this reader:
// a node whose inputs are all constants (initialisers, Constant nodes, earlier folds) and whose op is in the small vocabulary
// below is evaluated here and its output becomes an initialiser.  Tensors are int64 or float32, at most 65 536 elements.
struct CT {
    std::vector<int64_t> dims;
    bool is_int = true;
    std::vector<int64_t> iv;
    std::vector<float> fv;
    size_t size() const { return is_int ? iv.size() : fv.size(); }
};
bool ct_from(const OTensor& t, CT* c) {
    c->dims = t.dims;
    if (numel(t.dims) > 65536) return false;
    if (t.dtype == 7 || t.dtype == 6) { c->is_int = true; return tensor_ints(t, &c->iv); }
    if (t.dtype == 1) { c->is_int = false; return tensor_floats(t, &c->fv); }
    return false;
}
OTensor ct_to(const CT& c, const std::string& name) {
    OTensor t; t.name = name; t.dims = c.dims;
    if (c.is_int) { t.dtype = 7; t.i64 = c.iv; } else { t.dtype = 1; t.f = c.fv; }
    return t;
}
// gather elements of `c` by flat source index
CT ct_take(const CT& c, const std::vector<int64_t>& dims, const std::vector<size_t>& src) {
    CT o; o.dims = dims; o.is_int = c.is_int;
    if (c.is_int) { o.iv.resize(src.size()); for (size_t k = 0; k < src.size(); k++) o.iv[k] = c.iv[src[k]]; }
    else { o.fv.resize(src.size()); for (size_t k = 0; k < src.size(); k++) o.fv[k] = c.fv[src[k]]; }
    return o;
}
std::vector<size_t> ct_strides(const std::vector<int64_t>& d) {
    std::vector<size_t> st(d.size(), 1);
    for (int k = (int)d.size() - 2; k >= 0; k--) st[k] = st[k + 1] * (size_t)d[k + 1];
    return st;
}
// returns false when the node cannot (or need not) be folded
bool fold_node(const ONode& nd, const std::vector<CT>& in, const std::vector<bool>& present, CT* out) {
    auto has = [&](size_t k) { return k < in.size() && present[k]; };
    const std::string& op = nd.op;
    if (op == "Identity" && has(0)) { *out = in[0]; return true; }
    if (op == "Cast" && has(0)) {
        const int64_t to = nd.ai("to", 0);
        *out = in[0];
        if (to == 7 || to == 6) {
            if (!in[0].is_int) {
                // (float -> integer of a NaN, an infinity or a value beyond the target's range is undefined behaviour in C++ and
                // implementation-defined in ONNX: such a node is not folded - the graph then fails planning instead of silently folding garbage)
                const float lim = to == 6 ? 2147483520.f : 9.2233715e18f;
                for (float v : in[0].fv) if (!(v > -lim && v < lim)) return false;
                out->is_int = true; out->iv.resize(in[0].fv.size());
                for (size_t k = 0; k < in[0].fv.size(); k++) out->iv[k] = (int64_t)in[0].fv[k];
                out->fv.clear();
            }
            return true;
        }
        if (to == 1) { if (in[0].is_int) { out->is_int = false; out->fv.resize(in[0].iv.size()); for (size_t k = 0; k < in[0].iv.size(); k++) out->fv[k] = (float)in[0].iv[k]; out->iv.clear(); } return true; }
        return false;
    }
    if (op == "ConstantOfShape" && has(0) && in[0].is_int && in[0].dims.size() == 1) {
        size_t n = 1;
        for (int64_t d : in[0].iv) { if (d < 0 || d > 65536) return false; n *= (size_t)d; if (n > 65536) return false; }
        out->dims = in[0].iv;
        const OAttr* a = nd.attr("value");
        CT v; v.is_int = false; v.fv = {0.f};
        if (a && a->has_t && !ct_from(a->t, &v)) return false;
        if (v.size() != 1) return false;
        out->is_int = v.is_int;
        if (v.is_int) out->iv.assign(n, v.iv[0]); else out->fv.assign(n, v.fv[0]);
        return true;
    }
    if (op == "Shape" && has(0)) {
        if (nd.attr("start") || nd.attr("end")) return false;          // (opset 15's sliced Shape: not folded - the full shape would be wrong)
        out->is_int = true; out->dims = {(int64_t)in[0].dims.size()}; out->iv = in[0].dims; return true;
    }
    if (op == "Concat") {
        if (in.empty()) return false;
        for (size_t k = 0; k < in.size(); k++) if (!present[k] || in[k].is_int != in[0].is_int || in[k].dims.size() != in[0].dims.size()) return false;
        const int r = (int)in[0].dims.size();
        int64_t ax = nd.ai("axis", 0); if (ax < 0) ax += r;
        if (r < 1 || ax < 0 || ax >= r) return false;
        size_t outer = 1, inner = 1;
        for (int k = 0; k < ax; k++) outer *= (size_t)in[0].dims[k];
        for (int k = (int)ax + 1; k < r; k++) inner *= (size_t)in[0].dims[k];
        out->dims = in[0].dims; out->dims[ax] = 0; out->is_int = in[0].is_int;
        for (auto& c : in) {
            for (int k = 0; k < r; k++) if (k != ax && c.dims[k] != in[0].dims[k]) return false;
            out->dims[ax] += c.dims[ax];
        }
        if (numel(out->dims) > 65536) return false;
        for (size_t o = 0; o < outer; o++)
            for (auto& c : in) {
                const size_t n = (size_t)c.dims[ax] * inner;
                if (c.is_int) out->iv.insert(out->iv.end(), c.iv.begin() + o * n, c.iv.begin() + (o + 1) * n);
                else out->fv.insert(out->fv.end(), c.fv.begin() + o * n, c.fv.begin() + (o + 1) * n);
            }
        return true;
    }
    if (op == "Reshape" && has(0) && has(1) && in[1].is_int) {
        const size_t total = in[0].size();
        std::vector<int64_t> d = in[1].iv;
        size_t known = 1; int neg = -1;
        for (size_t k = 0; k < d.size(); k++) {
            if (d[k] == 0 && nd.ai("allowzero", 0) == 0) { if (k >= in[0].dims.size()) return false; d[k] = in[0].dims[k]; }
            if (d[k] == -1) { if (neg >= 0) return false; neg = (int)k; } else { if (d[k] < 0) return false; known *= (size_t)d[k]; }
        }
        if (neg >= 0) { if (!known || total % known) return false; d[neg] = (int64_t)(total / known); known *= (size_t)d[neg]; }
        if (known != total) return false;
        *out = in[0]; out->dims = d;
        return true;
    }
    if ((op == "Unsqueeze" || op == "Squeeze") && has(0)) {
        std::vector<int64_t> axes;
        if (const OAttr* a = nd.attr("axes")) axes = a->ints; else if (has(1) && in[1].is_int) axes = in[1].iv; else if (op == "Unsqueeze") return false;
        std::vector<int64_t> d = in[0].dims;
        if (op == "Unsqueeze") {
            const int r = (int)(d.size() + axes.size());
            std::vector<char> mark(r, 0);
            for (auto a : axes) { if (a < 0) a += r; if (a < 0 || a >= r || mark[a]) return false; mark[a] = 1; }
            std::vector<int64_t> o; size_t q = 0;
            for (int k = 0; k < r; k++) o.push_back(mark[k] ? 1 : d[q++]);
            d = o;
        } else {
            std::vector<int64_t> o;
            const int r = (int)d.size();
            for (int k = 0; k < r; k++) {
                bool drop = axes.empty() ? d[k] == 1 : false;
                for (auto a : axes) { if (a < 0) a += r; if (a == k) { if (d[k] != 1) return false; drop = true; } }
                if (!drop) o.push_back(d[k]);
            }
            d = o;
        }
        *out = in[0]; out->dims = d;
        return true;
    }
    if (op == "Transpose" && has(0)) {
        const int r = (int)in[0].dims.size();
        std::vector<int64_t> perm;
        if (const OAttr* a = nd.attr("perm")) perm = a->ints; else for (int k = r - 1; k >= 0; k--) perm.push_back(k);
        if ((int)perm.size() != r) return false;
        std::vector<char> seen(r, 0);
        for (auto q : perm) { if (q < 0 || q >= r || seen[q]) return false; seen[q] = 1; }
        std::vector<int64_t> od(r);
        for (int k = 0; k < r; k++) od[k] = in[0].dims[perm[k]];
        const auto ist = ct_strides(in[0].dims), ost = ct_strides(od);
        std::vector<size_t> src(in[0].size());
        for (size_t o = 0; o < src.size(); o++) { size_t rem = o, s = 0; for (int k = 0; k < r; k++) { const size_t c = rem / ost[k]; rem %= ost[k]; s += c * ist[perm[k]]; } src[o] = s; }
        *out = ct_take(in[0], od, src);
        return true;
    }
    if (op == "Slice" && has(0) && has(1) && has(2) && in[1].is_int && in[2].is_int) {
        const int r = (int)in[0].dims.size();
        const size_t ns = in[1].iv.size();
        std::vector<int64_t> axes, steps(ns, 1);
        if (has(3)) { if (!in[3].is_int) return false; axes = in[3].iv; } else for (size_t k = 0; k < ns; k++) axes.push_back((int64_t)k);
        if (has(4)) { if (!in[4].is_int) return false; steps = in[4].iv; }
        if (in[2].iv.size() != ns || axes.size() != ns || steps.size() != ns) return false;
        std::vector<int64_t> st(r, 0), sp(r, 1), cnt(in[0].dims);
        for (size_t k = 0; k < ns; k++) {
            int64_t ax = axes[k]; if (ax < 0) ax += r;
            if (ax < 0 || ax >= r || steps[k] == 0) return false;
            const int64_t n = in[0].dims[ax];
            // (a step beyond the axis length selects what a step of n + 1 selects; clamping keeps the count arithmetic below away from
            // INT64 overflow - steps come from the model file)
            steps[k] = std::min<int64_t>(std::max<int64_t>(steps[k], -(n + 1)), n + 1);
            int64_t a = in[1].iv[k], b = in[2].iv[k];
            if (steps[k] > 0) {
                if (a < 0) a += n; if (b < 0) b += n;
                a = std::min(std::max<int64_t>(a, 0), n); b = std::min(std::max<int64_t>(b, 0), n);
                cnt[ax] = b > a ? (b - a + steps[k] - 1) / steps[k] : 0;
            } else {
                if (a < 0) a += n; if (b < 0) b = b < -n ? -1 : b + n;        // (ends far below -n mean "through element 0")
                a = std::min(std::max<int64_t>(a, -1), n - 1); b = std::min(std::max<int64_t>(b, -1), n - 1);
                cnt[ax] = a > b ? (a - b + (-steps[k]) - 1) / (-steps[k]) : 0;
            }
            st[ax] = a; sp[ax] = steps[k];
        }
        const auto ist = ct_strides(in[0].dims), ost = ct_strides(cnt);
        std::vector<size_t> src(numel(cnt));
        for (size_t o = 0; o < src.size(); o++) { size_t rem = o; int64_t s = 0; for (int k = 0; k < r; k++) { const int64_t c = (int64_t)(rem / ost[k]); rem %= ost[k]; s += (st[k] + c * sp[k]) * (int64_t)ist[k]; } src[o] = (size_t)s; }
        *out = ct_take(in[0], cnt, src);
        return true;
    }
    if (op == "Gather" && has(0) && has(1) && in[1].is_int) {
        const int r = (int)in[0].dims.size();
        int64_t ax = nd.ai("axis", 0); if (ax < 0) ax += r;
        if (r < 1 || ax < 0 || ax >= r) return false;
        std::vector<int64_t> od(in[0].dims.begin(), in[0].dims.begin() + ax);
        od.insert(od.end(), in[1].dims.begin(), in[1].dims.end());
        od.insert(od.end(), in[0].dims.begin() + ax + 1, in[0].dims.end());
        if (numel(od) > 65536) return false;
        size_t outer = 1, inner = 1;
        for (int k = 0; k < ax; k++) outer *= (size_t)in[0].dims[k];
        for (int k = (int)ax + 1; k < r; k++) inner *= (size_t)in[0].dims[k];
        const int64_t n = in[0].dims[ax];
        std::vector<size_t> src;
        for (size_t o = 0; o < outer; o++)
            for (int64_t idx : in[1].iv) {
                if (idx < 0) idx += n;
                if (idx < 0 || idx >= n) return false;
                for (size_t q = 0; q < inner; q++) src.push_back((o * (size_t)n + (size_t)idx) * inner + q);
            }
        *out = ct_take(in[0], od, src);
        return true;
    }
    if ((op == "Add" || op == "Sub" || op == "Mul" || op == "Div") && has(0) && has(1) && in[0].is_int && in[1].is_int) {
        // integer shape arithmetic: equal shapes, or one side a single element
        const CT& a = in[0]; const CT& b = in[1];
        if (!(a.dims == b.dims || a.size() == 1 || b.size() == 1)) return false;
        const size_t n = std::max(a.size(), b.size());
        out->is_int = true; out->dims = a.size() >= b.size() ? a.dims : b.dims; out->iv.resize(n);
        for (size_t k = 0; k < n; k++) {
            const int64_t x = a.iv[a.size() == 1 ? 0 : k], y = b.iv[b.size() == 1 ? 0 : k];
            if (op == "Div" && (y == 0 || (y == -1 && x == INT64_MIN))) return false;
            // (values come from the model file: wrap like two's-complement hardware instead of signed-overflow UB)
            const uint64_t ux = (uint64_t)x, uy = (uint64_t)y;
            out->iv[k] = op == "Add" ? (int64_t)(ux + uy) : op == "Sub" ? (int64_t)(ux - uy) : op == "Mul" ? (int64_t)(ux * uy) : x / y;
        }
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------- in-graph audio front-ends
// Every ONNX classifier the reference ships computes its spectrogram inside the graph (internal/classifier/model_catalog.go:
// 412-426 BirdNET v2.4 "dfttrunc", :490-501 the BattyBirdNET backbone, :273-311 Perch v2 "with in-graph DFT",
// internal/classifier/birdnet_v3_onnx.go:44-48 "its mel front-end is a Conv1d").  The transform arrives in one of four
// forms - a MatMul with a constant (truncated) DFT basis, a strided Conv1d whose filters are window x DFT rows, the opset-17
// STFT operator, the opset-17 DFT operator - and the reader keeps such a value SYMBOLIC (SpecH) until it sees how it is
// used: real part or magnitude into a constant mel MatMul followed by the canonical tail becomes the TFLite-shaped chain
// (GATHER framing, window MUL, RFFT2D, CAST / COMPLEX_ABS, FULLY_CONNECTED) that the engine's front-end recogniser turns into
// the fp64 FFT + banded-mel kernels; any other use materialises the value as the dense fp32 GEMM the file literally asks for.
struct SpecH {
    int frames = -1;            // IR tensor [1, F, Lfft]: windowed, zero-padded frames
    int Lfft = 0, F = 0;
    std::vector<int> bins;      // DFT bin of every column (row)
    // 0 real, 1 imaginary, 2 real^2, 3 imaginary^2, 4 power, 5 magnitude, 6 [re rows | im rows] (Conv1d), 7 [.., K, 2] (STFT / DFT),
    // 8 = 7 squared elementwise
    int part = 0;
    bool bins_major = false;    // the ONNX value is [N, K, F] (Conv1d) rather than [N, F, K]
    bool last1 = false;         // a trailing axis of 1 is still attached (Slice [0:1] of a [.., K, 2] value before its Squeeze)
};

// B [rows n = 0..N-1][K columns]: every column j equal to cos(2 pi n k_j / N) (kind 0) or -sin(2 pi n k_j / N) (kind 1)?
bool dft_columns(const std::vector<float>& B, int N, int K, int* kind, std::vector<int>* bins) {
    if (N < 4 || K < 1 || (size_t)N * K != B.size()) return false;
    std::vector<double> ct(N), st(N);
    for (int i = 0; i < N; i++) { ct[i] = std::cos(2.0 * M_PI * i / N); st[i] = std::sin(2.0 * M_PI * i / N); }
    for (int kd = 0; kd < 2; kd++) {
        bins->assign(K, 0);
        bool ok = true;
        for (int j = 0; j < K && ok; j++) {
            int k;
            if (kd == 0) {
                double c1 = std::min(1.0, std::max(-1.0, (double)B[(size_t)1 * K + j]));
                k = (int)std::lround(std::acos(c1) * N / (2.0 * M_PI));
            } else {
                double s1 = -(double)B[(size_t)1 * K + j], s2 = -(double)B[(size_t)2 * K + j];
                double c = std::fabs(s1) > 1e-9 ? s2 / (2.0 * s1) : 1.0;
                k = (int)std::lround(std::atan2(s1, c) * N / (2.0 * M_PI));
            }
            if (k < 0 || k > N / 2) { ok = false; break; }
            for (int n = 0; n < N; n++) {
                const long ph = ((long)n * k) % N;
                const double want = kd == 0 ? ct[ph] : -st[ph];
                if (std::fabs((double)B[(size_t)n * K + j] - want) > 1e-5) { ok = false; break; }
            }
            (*bins)[j] = k;
        }
        if (ok) { *kind = kd; return true; }
    }
    return false;
}

// Conv1d filters W [2K rows][L taps]: rows 0..K-1 = w[n] cos(2 pi k_j n / N), rows K..2K-1 = -w[n] sin(2 pi k_j n / N)?
// Recovers the window, the transform length N >= L and the bins.
bool dft_conv_rows(const std::vector<float>& W, int R, int L, std::vector<float>* window, int* Nfft, std::vector<int>* bins) {
    if (R < 2 || (R & 1) || L < 8 || (size_t)R * L != W.size()) return false;
    const int K = R / 2;
    window->assign(L, 0.f);
    // the window from the row pair with the largest energy (every pair carries the same one)
    for (int n = 0; n < L; n++) (*window)[n] = std::hypot(W[(size_t)0 * L + n], W[(size_t)K * L + n]);
    int n0 = 0; float wmax = 0.f;
    for (int n = 0; n + 1 < L; n++) { float v = std::min((*window)[n], (*window)[n + 1]); if (v > wmax) { wmax = v; n0 = n; } }
    if (!(wmax > 0.f)) return false;
    std::vector<double> om(K);
    for (int j = 0; j < K; j++) {
        const double a0 = std::atan2(-(double)W[(size_t)(K + j) * L + n0], (double)W[(size_t)j * L + n0]);
        const double a1 = std::atan2(-(double)W[(size_t)(K + j) * L + n0 + 1], (double)W[(size_t)j * L + n0 + 1]);
        double d = a1 - a0;
        while (d < 0) d += 2.0 * M_PI;
        while (d >= 2.0 * M_PI) d -= 2.0 * M_PI;
        if (d > M_PI + 1e-6) return false;
        om[j] = d;
    }
    int N = L;
    if (K >= 2) {
        double dmin = 1e30;
        for (int j = 0; j + 1 < K; j++) { double d = std::fabs(om[j + 1] - om[j]); if (d > 1e-9) dmin = std::min(dmin, d); }
        if (dmin > 1e29) return false;
        N = (int)std::lround(2.0 * M_PI / dmin);
    }
    if (N < L || N > (1 << 16) || (N & 1)) return false;
    bins->assign(K, 0);
    double werr = 0;
    for (int j = 0; j < K; j++) {
        const int k = (int)std::lround(om[j] * N / (2.0 * M_PI));
        if (k < 0 || k > N / 2) return false;
        (*bins)[j] = k;
        for (int n = 0; n < L; n++) {
            const double ph = 2.0 * M_PI * (double)(((long)n * k) % N) / N, w = (*window)[n];
            werr = std::max(werr, std::fabs((double)W[(size_t)j * L + n] - w * std::cos(ph)));
            werr = std::max(werr, std::fabs((double)W[(size_t)(K + j) * L + n] + w * std::sin(ph)));
        }
    }
    *Nfft = N;
    return werr <= 2e-5 * std::max(1.0f, wmax);
}

int igcd_(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

}  // namespace

bool parse_onnx(const void* blob, size_t nbytes, TflModel* out, std::string* err, int* code) {
    *code = BNHIP_E_MODEL;
    auto fail = [&](const std::string& s) { *err = s; return false; };
    if (!blob || nbytes < 4) return fail("model is neither a TFLite flatbuffer nor an ONNX protobuf");
    // ---- ModelProto
    const uint8_t* gp = nullptr; size_t gl = 0;
    int64_t ir_version = 0;
    {
        PB pb((const uint8_t*)blob, nbytes);
        uint32_t f; int w; uint64_t v; const uint8_t* d; size_t l;
        while (pb.next(&f, &w, &v, &d, &l)) {
            if (f == 1 && w == 0) ir_version = (int64_t)v;
            else if (f == 7 && w == 2) { gp = d; gl = l; }
        }
        if (!pb.ok || !gp || ir_version <= 0 || ir_version > 64)
            return fail("model is neither a TFLite flatbuffer (TFL3) nor a readable ONNX ModelProto");
    }
    // ---- GraphProto
    std::vector<ONode> nodes;
    std::map<std::string, OTensor> inits;
    std::vector<OValue> g_in, g_out;
    {
        PB pb(gp, gl);
        uint32_t f; int w; uint64_t v; const uint8_t* d; size_t l;
        while (pb.next(&f, &w, &v, &d, &l)) {
            if (w != 2) continue;
            if (f == 1) { ONode nd; if (!parse_node(d, l, &nd)) return fail("ONNX: malformed NodeProto"); nodes.push_back(std::move(nd)); }
            else if (f == 5) { OTensor t; if (!parse_tensor(d, l, &t)) return fail("ONNX: malformed initializer"); std::string nm = t.name; inits[nm] = std::move(t); }
            else if (f == 11) { OValue vi; if (!parse_value_info(d, l, &vi)) return fail("ONNX: malformed graph input"); g_in.push_back(std::move(vi)); }
            else if (f == 12) { OValue vi; if (!parse_value_info(d, l, &vi)) return fail("ONNX: malformed graph output"); g_out.push_back(std::move(vi)); }
        }
        if (!pb.ok) return fail("ONNX: malformed GraphProto");
    }
    if (nodes.size() > 100000) return fail("ONNX: too many nodes");
    for (auto& kv : inits) if (kv.second.external) { *code = BNHIP_E_UNSUPPORTED; return fail("ONNX: external tensor data is not supported (the model must be self-contained): " + kv.first); }
    // Constant nodes behave like initializers
    for (auto& nd : nodes)
        if (nd.op == "Constant" && nd.out.size() == 1) {
            const OAttr* a = nd.attr("value");
            if (a && a->has_t) { OTensor t = a->t; t.name = nd.out[0]; inits[nd.out[0]] = std::move(t); }
            else if ((a = nd.attr("value_float"))) { OTensor t; t.name = nd.out[0]; t.dtype = 1; t.f.push_back(a->f); inits[nd.out[0]] = std::move(t); }
            else { *code = BNHIP_E_UNSUPPORTED; return fail("ONNX: Constant node without a tensor value: " + nd.name); }
        }
    // ... and so do small all-constant subgraphs (fold_node): evaluated in graph order, their outputs become initialisers
    std::set<int> folded;
    for (size_t ni = 0; ni < nodes.size(); ni++) {
        const ONode& nd = nodes[ni];
        if (nd.op == "Constant" || nd.out.size() != 1 || nd.in.empty() || (!nd.domain.empty() && nd.domain != "ai.onnx")) continue;
        std::vector<CT> cin(nd.in.size()); std::vector<bool> present(nd.in.size(), false);
        bool all_const = true;
        for (size_t k = 0; k < nd.in.size() && all_const; k++) {
            if (nd.in[k].empty()) continue;
            auto it = inits.find(nd.in[k]);
            if (it == inits.end() || !ct_from(it->second, &cin[k])) all_const = false; else present[k] = true;
        }
        CT res;
        if (!all_const || !fold_node(nd, cin, present, &res)) continue;
        inits[nd.out[0]] = ct_to(res, nd.out[0]);
        folded.insert((int)ni);
    }
    // the runtime input: the graph input that is not an initializer
    const OValue* gin = nullptr;
    for (auto& vi : g_in) if (!inits.count(vi.name)) { if (gin) { *code = BNHIP_E_UNSUPPORTED; return fail("ONNX: more than one runtime input"); } gin = &vi; }
    if (!gin) return fail("ONNX: graph has no runtime input");
    if (g_out.empty()) return fail("ONNX: graph has no outputs");
    if (gin->elem != 1) { *code = BNHIP_E_UNSUPPORTED; return fail("ONNX: runtime input must be float32"); }
    if (!gin->has_shape || gin->dims.size() < 2) { *code = BNHIP_E_UNSUPPORTED; return fail("ONNX: runtime input needs a static shape [batch, ...]"); }

    // ---- lowering
    *code = BNHIP_E_UNSUPPORTED;
    TflModel& m = *out;
    m.description = "onnx ir_version " + std::to_string(ir_version);
    std::map<std::string, int> tid;                          // ONNX value name -> tensor index
    auto new_act = [&](const std::string& name, const std::vector<int>& shape) {
        TflTensor t; t.name = name; t.shape = shape; t.type = TT_FLOAT32;
        m.tensors.push_back(std::move(t));
        tid[name] = (int)m.tensors.size() - 1;
        return (int)m.tensors.size() - 1;
    };
    {
        std::vector<int> sh;
        for (size_t k = 0; k < gin->dims.size(); k++) {
            int64_t dv = gin->dims[k];
            if (k == 0) dv = 1;                              // batch (symbolic or not): the engine batches itself
            if (dv <= 0 || dv > (1 << 28)) return fail("ONNX: runtime input has a dynamic non-batch dimension");
            sh.push_back((int)dv);
        }
        m.inputs.push_back(new_act(gin->name, sh));
    }
    auto const_f = [&](const std::string& name, std::vector<float>* v, std::vector<int64_t>* dims) -> bool {
        auto it = inits.find(name);
        if (it == inits.end()) return false;
        if (!tensor_floats(it->second, v)) return false;
        if (dims) *dims = it->second.dims;
        return true;
    };
    auto const_i = [&](const std::string& name, std::vector<int64_t>* v) -> bool {
        auto it = inits.find(name);
        return it != inits.end() && tensor_ints(it->second, v);
    };
    // tensor index of an operand: an existing activation, or a float initializer materialised as a constant
    auto operand = [&](const std::string& name) -> int {
        auto it = tid.find(name);
        if (it != tid.end()) return it->second;
        std::vector<float> v; std::vector<int64_t> dims;
        if (!const_f(name, &v, &dims)) return -1;
        std::vector<int> sh; for (auto d : dims) sh.push_back((int)d);
        int t = m.add_const_f32(name, sh, v);
        tid[name] = t;
        return t;
    };
    auto bshape = [&](const std::vector<int>& a, const std::vector<int>& b, std::vector<int>* z) -> bool {
        size_t r = std::max(a.size(), b.size());
        z->assign(r, 1);
        for (size_t k = 0; k < r; k++) {
            int da = k < r - a.size() ? 1 : a[k - (r - a.size())], db = k < r - b.size() ? 1 : b[k - (r - b.size())];
            if (da != db && da != 1 && db != 1) return false;
            (*z)[k] = std::max(da, db);
        }
        return true;
    };
    auto add_op = [&](int opc, std::vector<int> ins, int outt) -> TflOp& {
        TflOp o; o.code = opc; o.inputs = std::move(ins); o.outputs = {outt};
        m.ops.push_back(std::move(o));
        return m.ops.back();
    };
    // ---- image tensors.  ONNX convolutions are NCHW; the engine's kernels are NHWC.  A rank-4 value produced by a Conv / pool
    // (and everything elementwise downstream of it) is kept channels-last: its IR tensor has shape [N, H, W, C] while the
    // ONNX value it stands for is [N, C, H, W] (`chl`).  Axis attributes and per-channel constants are permuted on the way in;
    // a value only changes memory layout (TRANSPOSE) where the graph really depends on the element order: a Flatten /
    // Reshape of an image with H*W > 1 and C > 1, an unusual Transpose, or a graph output.
    std::set<int> chl;
    auto onnx_shape = [&](int t) -> std::vector<int> {
        const auto& sh = m.tensors[t].shape;
        if (!chl.count(t)) return sh;
        return {sh[0], sh[3], sh[1], sh[2]};
    };
    int tmp_id = 0;
    // NCHW-ordered copy of a channels-last value (no-op when only one of C, H*W exceeds 1: same element order)
    auto to_nchw = [&](int t) -> int {
        if (!chl.count(t)) return t;
        const auto sh = m.tensors[t].shape;           // [N, H, W, C]
        const std::vector<int> osh = {sh[0], sh[3], sh[1], sh[2]};
        const std::string nm = m.tensors[t].name + "/nchw" + std::to_string(tmp_id++);
        TflTensor nt; nt.name = nm; nt.shape = osh; nt.type = TT_FLOAT32;
        m.tensors.push_back(std::move(nt));
        const int o = (int)m.tensors.size() - 1;
        if (sh[3] == 1 || sh[1] * sh[2] == 1) add_op(OP_RESHAPE, {t}, o).new_shape = osh;
        else add_op(OP_TRANSPOSE, {t, m.add_const_i32(nm + "/perm", {4}, {0, 3, 1, 2})}, o);
        return o;
    };
    // channels-last copy of a plain NCHW-ordered rank-4 value
    auto to_chl = [&](int t) -> int {
        if (chl.count(t)) return t;
        const auto sh = m.tensors[t].shape;           // [N, C, H, W]
        const std::vector<int> osh = {sh[0], sh[2], sh[3], sh[1]};
        const std::string nm = m.tensors[t].name + "/nhwc" + std::to_string(tmp_id++);
        TflTensor nt; nt.name = nm; nt.shape = osh; nt.type = TT_FLOAT32;
        m.tensors.push_back(std::move(nt));
        const int o = (int)m.tensors.size() - 1;
        if (sh[1] == 1 || sh[2] * sh[3] == 1) add_op(OP_RESHAPE, {t}, o).new_shape = osh;
        else add_op(OP_TRANSPOSE, {t, m.add_const_i32(nm + "/perm", {4}, {0, 2, 3, 1})}, o);
        chl.insert(o);
        return o;
    };
    static const int kAxisToChl[4] = {0, 3, 1, 2};       // ONNX axis (N, C, H, W) -> axis of the channels-last tensor
    // conv / pool geometry: ONNX pads [top, left, bottom, right] (or auto_pad) -> output size and explicit top / left pads
    auto window_geom = [&](const ONode& nd, int H, int W, int kh, int kw, int sh_, int sw_, int dh, int dw, int* Ho, int* Wo, int* pt, int* pl,
                           std::string* why) -> bool {
        std::string ap = "NOTSET";
        if (const OAttr* a = nd.attr("auto_pad")) ap = a->s;
        int pb = 0, pr = 0; *pt = 0; *pl = 0;
        const int eh = dh * (kh - 1) + 1, ew = dw * (kw - 1) + 1;
        if (ap == "NOTSET" || ap.empty()) {
            if (const OAttr* a = nd.attr("pads")) {
                if (a->ints.size() != 4) { *why = "pads must have 4 entries"; return false; }
                *pt = (int)a->ints[0]; *pl = (int)a->ints[1]; pb = (int)a->ints[2]; pr = (int)a->ints[3];
            }
        } else if (ap == "SAME_UPPER" || ap == "SAME_LOWER") {
            const int oh = (H + sh_ - 1) / sh_, ow = (W + sw_ - 1) / sw_;
            const int th = std::max((oh - 1) * sh_ + eh - H, 0), tw = std::max((ow - 1) * sw_ + ew - W, 0);
            *pt = ap == "SAME_UPPER" ? th / 2 : th - th / 2; pb = th - *pt;
            *pl = ap == "SAME_UPPER" ? tw / 2 : tw - tw / 2; pr = tw - *pl;
        } else if (ap != "VALID") { *why = "auto_pad " + ap; return false; }
        if (*pt < 0 || *pl < 0 || pb < 0 || pr < 0 || *pt >= eh || *pl >= ew || pb >= eh || pr >= ew) { *why = "padding out of range"; return false; }
        if (nd.ai("ceil_mode", 0) != 0) { *why = "ceil_mode"; return false; }
        *Ho = (H + *pt + pb - eh) / sh_ + 1; *Wo = (W + *pl + pr - ew) / sw_ + 1;
        if (H + *pt + pb < eh || W + *pl + pr < ew || *Ho < 1 || *Wo < 1) { *why = "window larger than the padded image"; return false; }
        return true;
    };
    // ---- symbolic spectra (see SpecH) and the lazy [N, K, F] <-> [1, F, K] transposition of Conv1d-style tensors
    std::map<std::string, std::vector<int>> users;            // ONNX value -> consuming node indices
    for (size_t ni = 0; ni < nodes.size(); ni++) for (auto& nm : nodes[ni].in) if (!nm.empty()) users[nm].push_back((int)ni);
    std::set<std::string> graph_outs;
    for (auto& vi : g_out) graph_outs.insert(vi.name);
    std::set<int> tr3;                                        // IR tensor [1, F, K] standing for the ONNX value [N, K, F]
    std::map<std::string, SpecH> spec;                        // symbolic values (no IR tensor yet)
    std::map<std::string, std::vector<int>> spec_pending;     // ... and the nodes that must be lowered literally to materialise them
    std::set<int> lowered;                                    // node indices already lowered literally
    auto new_t = [&](const std::string& name, const std::vector<int>& shape, int type) {
        TflTensor t; t.name = name; t.shape = shape; t.type = type;
        m.tensors.push_back(std::move(t));
        return (int)m.tensors.size() - 1;
    };
    // plain [1, K, F] copy of a lazily transposed value
    auto untr3 = [&](int t) -> int {
        if (!tr3.count(t)) return t;
        const auto sh = m.tensors[t].shape;               // [1, F, K]
        const int o = new_t(m.tensors[t].name + "/kf" + std::to_string(tmp_id++), {sh[0], sh[2], sh[1]}, TT_FLOAT32);
        add_op(OP_TRANSPOSE, {t, m.add_const_i32(m.tensors[o].name + "/perm", {3}, {0, 2, 1})}, o);
        return o;
    };
    auto as_tr3 = [&](int t) -> int {                      // [1, F, K] form of a rank-3 value
        if (tr3.count(t)) return t;
        const auto sh = m.tensors[t].shape;               // [1, K, F]
        const int o = new_t(m.tensors[t].name + "/fk" + std::to_string(tmp_id++), {sh[0], sh[2], sh[1]}, TT_FLOAT32);
        if (sh[1] == 1 || sh[2] == 1) add_op(OP_RESHAPE, {t}, o).new_shape = m.tensors[o].shape;
        else add_op(OP_TRANSPOSE, {t, m.add_const_i32(m.tensors[o].name + "/perm", {3}, {0, 2, 1})}, o);
        tr3.insert(o);
        return o;
    };
    // the one consumer of an ONNX value that is not a graph output, or nullptr
    auto only_user = [&](const std::string& nm) -> const ONode* {
        auto it = users.find(nm);
        if (it == users.end() || it->second.size() != 1 || graph_outs.count(nm)) return nullptr;
        return &nodes[it->second[0]];
    };
    auto scalar_const = [&](const std::string& nm, float* v) -> bool {
        std::vector<float> c;
        if (!const_f(nm, &c, nullptr) || c.size() != 1) return false;
        *v = c[0];
        return true;
    };
    // Does the chain behind the mel MatMul output `nm` (mel axis = ONNX axis `mel_ax` of a rank-3 value) have the shape the
    // engine's recogniser fuses?  (Pow, Pow | Mul self, Pow | Max, Log, Mul) -> [reverse Slice over the mel axis] -> [Transpose 0 2 1]
    // -> Unsqueeze / Reshape to a rank-4 image
    // (the compressed spectrogram may ALSO be a graph output - Perch v2 exports its in-graph spectrogram as output 2,
    // internal/inference/onnx/classifier.go:495-505 - which the engine drops unless asked for; the tail is judged by its one consumer)
    auto sole_consumer = [&](const std::string& nm) -> const ONode* {
        auto it = users.find(nm);
        return it == users.end() || it->second.size() != 1 ? nullptr : &nodes[it->second[0]];
    };
    auto tail_ok = [&](std::string nm, int mel_ax) -> bool {
        int npow = 0; bool logc = false;
        for (int guard = 0; guard < 12; guard++) {
            const ONode* u = sole_consumer(nm);
            float v;
            if (!u || u->out.empty()) return false;
            if (u->op == "Pow" && npow < 2 && !logc && u->in.size() == 2 && u->in[0] == nm && scalar_const(u->in[1], &v)) { npow++; nm = u->out[0]; continue; }
            if (u->op == "Mul" && npow == 0 && !logc && u->in.size() == 2 && u->in[0] == nm && u->in[1] == nm) { npow = 1; nm = u->out[0]; continue; }
            const bool is_max = u->op == "Max" && u->in.size() == 2 && u->in[0] == nm && scalar_const(u->in[1], &v) && v > 0.f;
            // torch.clamp(min = floor): Clip with a constant min and no max
            const bool is_clip = u->op == "Clip" && !u->attr("max") && !u->attr("min") && u->in.size() >= 2 && u->in[0] == nm && scalar_const(u->in[1], &v) && v > 0.f &&
                                 (u->in.size() < 3 || u->in[2].empty());
            if ((is_max || is_clip) && npow == 0 && !logc) {
                const ONode* lg = sole_consumer(u->out[0]);
                if (!lg || lg->op != "Log") return false;
                logc = true; nm = lg->out[0];
                const ONode* mu = sole_consumer(nm);
                if (mu && mu->op == "Mul" && mu->in.size() == 2 && ((mu->in[0] == nm && scalar_const(mu->in[1], &v)) || (mu->in[1] == nm && scalar_const(mu->in[0], &v)))) nm = mu->out[0];
                continue;
            }
            if (u->op == "Slice" && u->in.size() >= 5) {
                std::vector<int64_t> st, en, ax, sp;
                if (!const_i(u->in[1], &st) || !const_i(u->in[2], &en) || !const_i(u->in[3], &ax) || !const_i(u->in[4], &sp)) return false;
                if (st.size() != 1 || en.size() != 1 || ax.size() != 1 || sp.size() != 1) return false;      // (an empty initializer has no data() to index)
                if (sp[0] != -1 || st[0] != -1 || (ax[0] != mel_ax && ax[0] != mel_ax - 3)) return false;
                nm = u->out[0];
                u = sole_consumer(nm);
                if (!u) return false;
            }
            if (u->op == "Transpose") {
                const OAttr* pa = u->attr("perm");
                if (!pa || pa->ints != std::vector<int64_t>{0, 2, 1}) return false;
                nm = u->out[0];
                u = sole_consumer(nm);
                if (!u) return false;
            }
            return u->op == "Unsqueeze" || u->op == "Reshape";
        }
        return false;
    };
    // IR side: is `t` ([1, F, Lfft]) produced by [PAD of the last axis] <- MUL(constant window) <- RESHAPE <- GATHER(constant
    // sliding-window selector), the framing the engine's recogniser walks?
    auto frames_canonical = [&](int t) -> bool {
        std::vector<int> prod(m.tensors.size(), -1);
        for (size_t oi = 0; oi < m.ops.size(); oi++) for (int o : m.ops[oi].outputs) prod[o] = (int)oi;
        auto skip = [&](int x) { while (prod[x] >= 0 && m.ops[prod[x]].code == OP_RESHAPE) x = m.ops[prod[x]].inputs[0]; return x; };
        t = skip(t);
        if (prod[t] >= 0 && m.ops[prod[t]].code == OP_PAD) t = skip(m.ops[prod[t]].inputs[0]);
        if (prod[t] >= 0 && m.ops[prod[t]].code == OP_MUL) {
            const TflOp& mu = m.ops[prod[t]];
            const int wc = m.tensors[mu.inputs[1]].data ? 1 : (m.tensors[mu.inputs[0]].data ? 0 : -1);
            if (wc < 0) return false;
            t = skip(mu.inputs[1 - wc]);
        }
        return prod[t] >= 0 && m.ops[prod[t]].code == OP_GATHER && m.tensors[m.ops[prod[t]].inputs[1]].data != nullptr;
    };
    // framing of a [1, T] signal into windowed, zero-padded frames [1, F, Lfft], written the way tf.signal.frame converts
    auto emit_frames = [&](int sig, int T, int L, int hop, int Lfft, const std::vector<float>& window, const std::string& base, int* F_out) -> int {
        if (hop < 1 || L < 1 || L > T || Lfft < L) { *F_out = 0; return -1; }     // callers range-check; a zero hop must never reach the division
        const int F = (T - L) / hop + 1;
        int sub = igcd_(igcd_(L, hop), T);
        const int nsub = T / sub, Q = L / sub, step = hop / sub;
        const int r1 = new_t(base + "/subframes", {1, nsub, sub}, TT_FLOAT32);
        add_op(OP_RESHAPE, {sig}, r1).new_shape = {1, nsub, sub};
        std::vector<int32_t> sel((size_t)F * Q);
        for (int f = 0; f < F; f++) for (int q = 0; q < Q; q++) sel[(size_t)f * Q + q] = f * step + q;
        const int ga = new_t(base + "/gather", {1, F, Q, sub}, TT_FLOAT32);
        { TflOp& g = add_op(OP_GATHER, {r1, m.add_const_i32(base + "/frame_selector", {F, Q}, sel)}, ga); g.axis = 1; g.batch_dims = 0; }
        const int fr = new_t(base + "/frames", {1, F, L}, TT_FLOAT32);
        add_op(OP_RESHAPE, {ga}, fr).new_shape = {1, F, L};
        int wn = new_t(base + "/windowed", {1, F, L}, TT_FLOAT32);
        add_op(OP_MUL, {fr, m.add_const_f32(base + "/window", {L}, window)}, wn);
        if (Lfft != L) {
            const int pd = new_t(base + "/padded", {1, F, Lfft}, TT_FLOAT32);
            add_op(OP_PAD, {wn, m.add_const_i32(base + "/frame_pad", {3, 2}, {0, 0, 0, 0, 0, Lfft - L})}, pd);
            wn = pd;
        }
        *F_out = F;
        return wn;
    };
    std::function<bool(const ONode&, bool)> lower_node;
    // materialise symbolic inputs: lower, literally, every node that was skipped to keep them symbolic
    auto materialise = [&](const std::string& nm) -> bool {
        auto it = spec_pending.find(nm);
        if (it == spec_pending.end()) return true;
        std::vector<int> todo = it->second;
        std::sort(todo.begin(), todo.end());
        todo.erase(std::unique(todo.begin(), todo.end()), todo.end());
        for (int ni : todo) {
            if (lowered.count(ni)) continue;
            lowered.insert(ni);
            for (auto& o : nodes[ni].out) { spec.erase(o); }
            if (!lower_node(nodes[ni], false)) return false;
        }
        for (int ni : todo) for (auto& o : nodes[ni].out) spec_pending.erase(o);
        return true;
    };
    auto make_spec = [&](const ONode& nd, int ni, const SpecH& h, std::initializer_list<std::string> from) {
        spec[nd.out[0]] = h;
        std::vector<int> pend;
        for (auto& f : from) { auto it = spec_pending.find(f); if (it != spec_pending.end()) pend.insert(pend.end(), it->second.begin(), it->second.end()); }
        pend.push_back(ni);
        spec_pending[nd.out[0]] = pend;
    };
    // RFFT2D chain + mel projection for a real-part / magnitude spectrum; returns the [1, F, n_mels] tensor
    auto emit_fused = [&](const SpecH& h, const std::vector<float>& mel_km /*[K][n_mels]*/, int n_mels, const std::string& base) -> int {
        const int nb = h.Lfft / 2 + 1;
        const int e1 = new_t(base + "/fft_in", {1, h.F, 1, h.Lfft}, TT_FLOAT32);
        add_op(OP_RESHAPE, {h.frames}, e1).new_shape = {1, h.F, 1, h.Lfft};
        const int ft = new_t(base + "/rfft", {1, h.F, 1, nb}, TT_COMPLEX64);
        add_op(OP_RFFT2D, {e1, m.add_const_i32(base + "/fft_length", {2}, {1, h.Lfft})}, ft);
        const int sq = new_t(base + "/bins_c", {1, h.F, nb}, TT_COMPLEX64);
        add_op(OP_RESHAPE, {ft}, sq).new_shape = {1, h.F, nb};
        const int re = new_t(base + (h.part == 5 ? "/magnitude" : "/real"), {1, h.F, nb}, TT_FLOAT32);
        if (h.part == 5) add_op(OP_COMPLEX_ABS, {sq}, re);
        else { TflOp& c = add_op(OP_CAST, {sq}, re); c.in_type = TT_COMPLEX64; c.out_type = TT_FLOAT32; }
        const int r2 = new_t(base + "/bins2d", {h.F, nb}, TT_FLOAT32);
        add_op(OP_RESHAPE, {re}, r2).new_shape = {h.F, nb};
        std::vector<float> wfull((size_t)n_mels * nb, 0.f);               // [n_mels][nb]: the file's columns scattered to their bins
        for (size_t j = 0; j < h.bins.size(); j++)
            for (int mm = 0; mm < n_mels; mm++) wfull[(size_t)mm * nb + h.bins[j]] += mel_km[j * (size_t)n_mels + mm];
        const int mm2 = new_t(base + "/mel2d", {h.F, n_mels}, TT_FLOAT32);
        add_op(OP_FULLY_CONNECTED, {r2, m.add_const_f32(base + "/mel", {n_mels, nb}, wfull), -1}, mm2);
        const int r3 = new_t(base + "/mel", {1, h.F, n_mels}, TT_FLOAT32);
        add_op(OP_RESHAPE, {mm2}, r3).new_shape = {1, h.F, n_mels};
        return r3;
    };
    int node_index = -1;
    // Recognised producers / consumers of symbolic spectra.  Returns 1: the node is handled (its output is symbolic or was
    // emitted in fused form); 0: lower it literally (symbolic inputs have been materialised); -1: error (err set).
    auto try_symbolic = [&](const ONode& nd, int ni) -> int {
        if (nd.out.empty()) return 0;
        const std::string& oname = nd.out[0];
        auto act_of = [&](const std::string& nm) -> int { auto it = tid.find(nm); return it == tid.end() || m.tensors[it->second].data ? -1 : it->second; };
        auto H = [&](size_t k) -> const SpecH* { if (k >= nd.in.size()) return nullptr; auto it = spec.find(nd.in[k]); return it == spec.end() ? nullptr : &it->second; };
        bool any = false;
        for (auto& nm : nd.in) if (spec.count(nm)) any = true;
        // ---- producers
        if (!any && nd.op == "MatMul" && nd.in.size() == 2) {
            const int a = act_of(nd.in[0]);
            std::vector<float> B; std::vector<int64_t> bd;
            if (a >= 0 && !chl.count(a) && !tr3.count(a) && m.tensors[a].shape.size() == 3 && const_f(nd.in[1], &B, &bd) && bd.size() == 2 &&
                bd[0] == m.tensors[a].shape[2] && bd[0] >= 64 && frames_canonical(a)) {
                SpecH h; int kind = 0;
                if (dft_columns(B, (int)bd[0], (int)bd[1], &kind, &h.bins)) {
                    h.frames = a; h.Lfft = (int)bd[0]; h.F = m.tensors[a].shape[1]; h.part = kind;
                    make_spec(nd, ni, h, {});
                    return 1;
                }
            }
            return 0;
        }
        if (!any && nd.op == "Conv" && nd.in.size() == 2) {
            const int a = act_of(nd.in[0]);
            std::vector<float> Wv; std::vector<int64_t> wd;
            if (a >= 0 && !chl.count(a) && m.tensors[a].shape.size() == 3 && !tr3.count(a) && m.tensors[a].shape[1] == 1 && const_f(nd.in[1], &Wv, &wd) &&
                wd.size() == 3 && wd[1] == 1 && nd.ai("group", 1) == 1) {
                const OAttr* st = nd.attr("strides"); const OAttr* pd = nd.attr("pads"); const OAttr* dl = nd.attr("dilations"); const OAttr* ap = nd.attr("auto_pad");
                const int64_t hop64 = st && st->ints.size() == 1 ? st->ints[0] : 1;
                const int hop = hop64 >= 1 && hop64 <= m.tensors[a].shape[2] ? (int)hop64 : 0;       // (0 = not a framing stride: lowered literally)
                bool plain = (!pd || (pd->ints.size() == 2 && pd->ints[0] == 0 && pd->ints[1] == 0)) && (!dl || (dl->ints.size() == 1 && dl->ints[0] == 1)) &&
                             (!ap || ap->s.empty() || ap->s == "NOTSET" || ap->s == "VALID");
                SpecH h; std::vector<float> window; int Nfft = 0;
                const int T = m.tensors[a].shape[2], L = (int)wd[2];
                if (plain && hop >= 1 && L <= T && dft_conv_rows(Wv, (int)wd[0], L, &window, &Nfft, &h.bins)) {
                    const int sig = new_t(oname + "/signal", {1, T}, TT_FLOAT32);
                    add_op(OP_RESHAPE, {a}, sig).new_shape = {1, T};
                    h.frames = emit_frames(sig, T, L, hop, Nfft, window, oname, &h.F);
                    h.Lfft = Nfft; h.part = 6; h.bins_major = true;
                    make_spec(nd, ni, h, {});
                    return 1;
                }
            }
            return 0;
        }
        if (!any && (nd.op == "STFT" || nd.op == "DFT")) {
            const int a = act_of(nd.in[0]);
            if (a < 0 || chl.count(a) || tr3.count(a)) return 0;
            const auto ash = m.tensors[a].shape;
            SpecH h; h.part = 7;
            if (nd.op == "STFT") {
                std::vector<int64_t> step, flen; std::vector<float> win;
                if (nd.ai("onesided", 1) != 1 || nd.in.size() < 2 || !const_i(nd.in[1], &step) || step.size() != 1 || step[0] < 1) return 0;
                const bool has_w = nd.in.size() > 2 && !nd.in[2].empty();
                if (has_w && !const_f(nd.in[2], &win, nullptr)) return 0;
                const bool sig3 = ash.size() == 3 && ash[2] == 1;
                if (!(sig3 || ash.size() == 2)) return 0;
                const int T = ash[1];
                int64_t L64 = has_w ? (int64_t)win.size() : 0;
                if (nd.in.size() > 3 && !nd.in[3].empty()) { if (!const_i(nd.in[3], &flen) || flen.size() != 1) return 0; if (has_w && flen[0] != L64) return 0; L64 = flen[0]; }
                // range checks on the int64 values, before any narrowing (2^32 would become hop 0, a negative length a huge allocation)
                if (L64 < 8 || L64 > T || (L64 & 1) || step[0] > T) return 0;
                const int L = (int)L64;
                if (!has_w) win.assign(L, 1.0f);
                const int sig = new_t(oname + "/signal", {1, T}, TT_FLOAT32);
                add_op(OP_RESHAPE, {a}, sig).new_shape = {1, T};
                h.frames = emit_frames(sig, T, L, (int)step[0], L, win, oname, &h.F);
                h.Lfft = L;
            } else {
                if (nd.ai("onesided", 0) != 1 || nd.ai("inverse", 0) != 0 || ash.size() != 4 || ash[3] != 1) return 0;
                int64_t ax = nd.ai("axis", 1); if (ax < 0) ax += 4;
                if (ax != 2 || !frames_canonical(a)) return 0;
                int n = ash[2], N = n;
                if (nd.in.size() > 1 && !nd.in[1].empty()) { std::vector<int64_t> dl; if (!const_i(nd.in[1], &dl) || dl.size() != 1 || dl[0] < 0 || dl[0] > (1 << 20)) return 0; N = (int)dl[0]; }
                if (N < n || N < 8 || (N & 1)) return 0;
                int fr = new_t(oname + "/frames", {1, ash[1], n}, TT_FLOAT32);
                add_op(OP_RESHAPE, {a}, fr).new_shape = {1, ash[1], n};
                if (N != n) {
                    const int pd = new_t(oname + "/padded", {1, ash[1], N}, TT_FLOAT32);
                    add_op(OP_PAD, {fr, m.add_const_i32(oname + "/frame_pad", {3, 2}, {0, 0, 0, 0, 0, N - n})}, pd);
                    fr = pd;
                }
                h.frames = fr; h.Lfft = N; h.F = ash[1];
            }
            for (int k = 0; k <= h.Lfft / 2; k++) h.bins.push_back(k);
            make_spec(nd, ni, h, {});
            return 1;
        }
        if (!any) return 0;
        // ---- consumers
        const SpecH* h0 = H(0);
        const SpecH* h1 = H(1);
        auto derive = [&](SpecH h, int part, std::initializer_list<std::string> from) { h.part = part; make_spec(nd, ni, h, from); return 1; };
        auto same_src = [&](const SpecH& x, const SpecH& y) { return x.frames == y.frames && x.bins == y.bins && x.bins_major == y.bins_major && x.last1 == y.last1; };
        if (nd.op == "Slice" && h0 && (h0->part == 6 || h0->part == 7) && nd.in.size() >= 4) {
            std::vector<int64_t> st, en, ax, sp;
            if (const_i(nd.in[1], &st) && const_i(nd.in[2], &en) && const_i(nd.in[3], &ax) && st.size() == 1 && en.size() == 1 && ax.size() == 1 &&
                (nd.in.size() < 5 || nd.in[4].empty() || (const_i(nd.in[4], &sp) && sp.size() == 1 && sp[0] == 1))) {
                const int K = (int)h0->bins.size();
                if (h0->part == 6 && ax[0] == 1 && ((st[0] == 0 && en[0] == K) || (st[0] == K && en[0] >= 2 * K)))
                    return derive(*h0, st[0] == 0 ? 0 : 1, {nd.in[0]});
                if (h0->part == 7 && (ax[0] == 3 || ax[0] == -1) && ((st[0] == 0 && en[0] == 1) || (st[0] == 1 && en[0] >= 2))) {
                    SpecH h = *h0; h.last1 = true;
                    return derive(h, st[0] == 0 ? 0 : 1, {nd.in[0]});
                }
            }
        } else if (nd.op == "Squeeze" && h0 && h0->last1 && (h0->part == 0 || h0->part == 1)) {
            std::vector<int64_t> axes;
            if (const OAttr* pa = nd.attr("axes")) axes = pa->ints; else if (nd.in.size() > 1) const_i(nd.in[1], &axes);
            if (axes.size() == 1 && (axes[0] == 3 || axes[0] == -1)) { SpecH h = *h0; h.last1 = false; return derive(h, h0->part, {nd.in[0]}); }
        } else if (nd.op == "Gather" && h0 && h0->part == 7 && nd.in.size() == 2) {
            std::vector<int64_t> idx; auto it = inits.find(nd.in[1]);
            const int64_t ax = nd.ai("axis", 0);
            if (it != inits.end() && it->second.dims.empty() && tensor_ints(it->second, &idx) && idx.size() == 1 && (ax == 3 || ax == -1) && (idx[0] == 0 || idx[0] == 1))
                return derive(*h0, (int)idx[0], {nd.in[0]});
        } else if (nd.op == "Transpose" && h0 && nd.in.size() == 1 && h0->part != 6) {
            // torch.stft's layout change [N, F, K, 2] -> [N, K, F, 2] and `.transpose(1, 2)` of a rank-3 spectrum: only the tag flips
            const OAttr* pa = nd.attr("perm");
            const bool r4 = h0->part == 7 || h0->part == 8 || h0->last1;
            if (pa && ((r4 && pa->ints == std::vector<int64_t>{0, 2, 1, 3}) || (!r4 && pa->ints == std::vector<int64_t>{0, 2, 1}))) {
                SpecH h = *h0; h.bins_major = !h.bins_major;
                return derive(h, h0->part, {nd.in[0]});
            }
        } else if (nd.op == "Mul" && h0 && h1 && nd.in[0] == nd.in[1] && !h0->last1) {
            if (h0->part == 0 || h0->part == 1) return derive(*h0, h0->part + 2, {nd.in[0]});
            if (h0->part == 7) return derive(*h0, 8, {nd.in[0]});
        } else if (nd.op == "Pow" && h0 && !h1 && !h0->last1 && nd.in.size() == 2) {
            float e;
            if (scalar_const(nd.in[1], &e) && e == 2.0f) {
                if (h0->part == 0 || h0->part == 1) return derive(*h0, h0->part + 2, {nd.in[0]});
                if (h0->part == 7) return derive(*h0, 8, {nd.in[0]});
            }
        } else if (nd.op == "Add" && h0 && h1 && same_src(*h0, *h1) && ((h0->part == 2 && h1->part == 3) || (h0->part == 3 && h1->part == 2))) {
            return derive(*h0, 4, {nd.in[0], nd.in[1]});
        } else if (nd.op == "ReduceSum" && h0 && h0->part == 8) {
            std::vector<int64_t> axes;
            if (const OAttr* pa = nd.attr("axes")) axes = pa->ints; else if (nd.in.size() > 1 && !nd.in[1].empty()) const_i(nd.in[1], &axes);
            if (axes.size() == 1 && (axes[0] == 3 || axes[0] == -1) && nd.ai("keepdims", 1) == 0) return derive(*h0, 4, {nd.in[0]});
        } else if (nd.op == "Sqrt" && h0 && h0->part == 4) {
            return derive(*h0, 5, {nd.in[0]});
        } else if (nd.op == "MatMul" && nd.in.size() == 2) {
            // the mel projection: [N, F, K] x [K, M]  or  [M, K] x [N, K, F]
            const SpecH* h = h0 && !h0->bins_major ? h0 : (h1 && h1->bins_major ? h1 : nullptr);
            const std::string& cname = h == h0 ? nd.in[1] : nd.in[0];
            std::vector<float> Mv; std::vector<int64_t> md;
            if (h && !h->last1 && (h->part == 0 || h->part == 5) && const_f(cname, &Mv, &md) && md.size() == 2) {
                const int K = (int)h->bins.size();
                const bool left = h != h0;                            // constant on the left: [M, K]
                const int n_mels = (int)(left ? md[0] : md[1]);
                if ((left ? md[1] : md[0]) == K && tail_ok(oname, left ? 1 : 2)) {
                    std::vector<float> km((size_t)K * n_mels);
                    for (int k = 0; k < K; k++) for (int mm = 0; mm < n_mels; mm++) km[(size_t)k * n_mels + mm] = left ? Mv[(size_t)mm * K + k] : Mv[(size_t)k * n_mels + mm];
                    const SpecH hc = *h;
                    const int r3 = emit_fused(hc, km, n_mels, oname);
                    tid[oname] = r3;
                    if (hc.bins_major) tr3.insert(r3);
                    return 1;
                }
            }
        }
        // unrecognised use: the literal graph
        for (auto& nm : nd.in) if (spec.count(nm) && !materialise(nm)) return -1;
        return 0;
    };
    lower_node = [&](const ONode& nd, bool sym) -> bool {
        const std::string where = nd.op + " (" + (nd.name.empty() ? (nd.out.empty() ? "?" : nd.out[0]) : nd.name) + ")";
        if (!nd.domain.empty() && nd.domain != "ai.onnx") return fail("ONNX: operator from unsupported domain " + nd.domain + ": " + where);
        if (nd.op == "Constant" || folded.count((int)(&nd - nodes.data()))) return true;
        if (sym) {
            const int r = try_symbolic(nd, node_index);
            if (r < 0) return false;
            if (r == 1) return true;
        }
        if (nd.out.empty() || nd.in.empty()) { *code = BNHIP_E_MODEL; return fail("ONNX: node without inputs/outputs: " + where); }
        {
            // ops that understand the lazy [N, K, F] <-> [1, F, K] tag; everything else gets the plain tensor
            static const std::set<std::string> tr3_aware = {"Add", "Sub", "Mul", "Div", "Pow", "Max", "Min", "Relu", "Sigmoid", "Tanh", "Exp", "Log", "Sqrt",
                "Abs", "Neg", "Floor", "Ceil", "HardSwish", "LeakyRelu", "Elu", "Gelu", "Sin", "Cos", "Identity", "Dropout", "Cast", "Slice", "Transpose", "MatMul", "Conv"};
            if (!tr3_aware.count(nd.op))
                for (auto& nm : nd.in) { auto it = tid.find(nm); if (it != tid.end() && tr3.count(it->second)) it->second = untr3(it->second); }
        }
        const std::string& oname = nd.out[0];
        auto in_act = [&](size_t k) -> int { if (k >= nd.in.size()) return -1; auto it = tid.find(nd.in[k]); return it == tid.end() || m.tensors[it->second].data ? -1 : it->second; };
        if (nd.op == "MatMul" && nd.in.size() == 2 && in_act(0) < 0 && in_act(1) >= 0 && m.tensors[in_act(1)].shape.size() == 3 && !chl.count(in_act(1))) {
            // [M, K] x [N, K, F] (the mel projection of a bins-major spectrogram): a dense layer over the K axis of the [1, F, K] form
            std::vector<float> A; std::vector<int64_t> ad;
            if (!const_f(nd.in[0], &A, &ad) || ad.size() != 2) return fail("ONNX: " + where + ": first operand must be a constant matrix or an activation");
            const int x = as_tr3(in_act(1));
            const auto xs = m.tensors[x].shape;               // [1, F, K]
            if (xs[2] != (int)ad[1]) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": inner dimensions disagree"); }
            const int o = new_act(oname, {1, xs[1], (int)ad[0]});
            add_op(OP_FULLY_CONNECTED, {x, m.add_const_f32(nd.in[0] + "/w", {(int)ad[0], (int)ad[1]}, A)}, o).keep_num_dims = true;
            tr3.insert(o);
        } else if (nd.op == "Gemm" || nd.op == "MatMul") {
            int a = in_act(0);
            if (a >= 0) a = untr3(a);
            if (a < 0 || nd.in.size() < 2) return fail("ONNX: " + where + ": first operand must be an activation");
            a = to_nchw(a);
            std::vector<float> B; std::vector<int64_t> bd;
            if (!const_f(nd.in[1], &B, &bd) || bd.size() != 2) return fail("ONNX: " + where + ": second operand must be a constant float matrix");
            const bool gemm = nd.op == "Gemm";
            const float alpha = gemm ? nd.af("alpha", 1.0f) : 1.0f, beta = gemm ? nd.af("beta", 1.0f) : 1.0f;
            const bool tb = gemm && nd.ai("transB", 0) != 0;
            if (gemm && nd.ai("transA", 0) != 0) return fail("ONNX: " + where + ": transA is not supported");
            const int K = (int)(tb ? bd[1] : bd[0]), N = (int)(tb ? bd[0] : bd[1]);
            const std::vector<int> ash = m.tensors[a].shape;      // by value: add_const_f32 below reallocates m.tensors
            if (ash.empty() || ash.back() != K) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": inner dimensions disagree"); }
            std::vector<float> W((size_t)N * K);
            for (int n = 0; n < N; n++)
                for (int k = 0; k < K; k++) W[(size_t)n * K + k] = alpha * (tb ? B[(size_t)n * K + k] : B[(size_t)k * N + n]);
            std::vector<int> ins = {a, m.add_const_f32(nd.in[1] + "/w", {N, K}, W)};
            if (gemm && nd.in.size() > 2 && !nd.in[2].empty()) {
                std::vector<float> C; std::vector<int64_t> cd;
                if (!const_f(nd.in[2], &C, &cd)) return fail("ONNX: " + where + ": C must be a constant");
                std::vector<float> bias(N);
                if (C.size() == 1) for (int n = 0; n < N; n++) bias[n] = beta * C[0];
                else if ((int)C.size() == N) for (int n = 0; n < N; n++) bias[n] = beta * C[n];
                else return fail("ONNX: " + where + ": C must broadcast along the output columns");
                ins.push_back(m.add_const_f32(nd.in[2] + "/b", {N}, bias));
            }
            std::vector<int> osh = ash; osh.back() = N;
            add_op(OP_FULLY_CONNECTED, ins, new_act(oname, osh)).keep_num_dims = true;
        } else if (nd.op == "Add" || nd.op == "Sub" || nd.op == "Mul" || nd.op == "Div" || nd.op == "Pow" || nd.op == "Max" || nd.op == "Min") {
            if (nd.in.size() != 2) return fail("ONNX: " + where + ": exactly two operands are supported");
            int a = -1, b = -1;
            const int ia = in_act(0), ib = in_act(1);
            const bool img = (ia >= 0 && chl.count(ia)) || (ib >= 0 && chl.count(ib));
            if (img) {
                // image arithmetic: the other side is an image too, or a constant that broadcasts over [N, C, H, W]
                auto side = [&](size_t k, int act) -> int {
                    if (act >= 0) return m.tensors[act].shape.size() == 4 ? to_chl(act) : -2;
                    std::vector<float> v; std::vector<int64_t> dims;
                    if (!const_f(nd.in[k], &v, &dims) || dims.size() > 4) return -1;
                    std::vector<int64_t> d4(4, 1);
                    for (size_t q = 0; q < dims.size(); q++) d4[4 - dims.size() + q] = dims[q];
                    std::vector<float> w(v.size());                  // [n][c][h][w] -> [n][h][w][c]
                    size_t at = 0;
                    for (int64_t n = 0; n < d4[0]; n++) for (int64_t c = 0; c < d4[1]; c++) for (int64_t h = 0; h < d4[2]; h++) for (int64_t x = 0; x < d4[3]; x++)
                        w[(((size_t)n * d4[2] + h) * d4[3] + x) * d4[1] + c] = v[at++];
                    return m.add_const_f32(nd.in[k] + "/nhwc" + std::to_string(tmp_id++), {(int)d4[0], (int)d4[2], (int)d4[3], (int)d4[1]}, w);
                };
                a = side(0, ia); b = side(1, ib);
                if (a == -2 || b == -2) return fail("ONNX: " + where + ": an image and a non-image activation cannot be combined");
            } else { a = operand(nd.in[0]); b = operand(nd.in[1]); }
            if (a < 0 || b < 0) return fail("ONNX: " + where + ": operand is neither an activation nor a float constant");
            bool out_tr3 = false;
            if (tr3.count(a) || tr3.count(b)) {
                // the tag survives arithmetic with a scalar or with another tagged value of the same shape
                auto scalar = [&](int t) { return m.tensors[t].data != nullptr && m.tensors[t].numel() == 1; };
                if ((tr3.count(a) && tr3.count(b) && m.tensors[a].shape == m.tensors[b].shape) || (tr3.count(a) && scalar(b)) || (tr3.count(b) && scalar(a))) out_tr3 = true;
                else { a = untr3(a); b = untr3(b); }
            }
            std::vector<int> z;
            if (!bshape(m.tensors[a].shape, m.tensors[b].shape, &z)) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": shapes do not broadcast"); }
            const int opc = nd.op == "Add" ? OP_ADD : nd.op == "Sub" ? OP_SUB : nd.op == "Mul" ? OP_MUL : nd.op == "Div" ? OP_DIV :
                            nd.op == "Pow" ? OP_POW : nd.op == "Max" ? OP_MAXIMUM : OP_MINIMUM;
            const int zo = new_act(oname, z);
            add_op(opc, {a, b}, zo);
            if (img) chl.insert(zo);
            if (out_tr3) tr3.insert(zo);
        } else if (nd.op == "Relu" || nd.op == "Sigmoid" || nd.op == "Tanh" || nd.op == "Exp" || nd.op == "Log" || nd.op == "Sqrt" ||
                   nd.op == "Abs" || nd.op == "Neg" || nd.op == "Floor" || nd.op == "Ceil" || nd.op == "HardSwish" || nd.op == "LeakyRelu" ||
                   nd.op == "Elu" || nd.op == "Gelu" || nd.op == "Sin" || nd.op == "Cos") {
            const int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            const int opc = nd.op == "Relu" ? OP_RELU : nd.op == "Sigmoid" ? OP_LOGISTIC : nd.op == "Tanh" ? OP_TANH : nd.op == "Exp" ? OP_EXP :
                            nd.op == "Log" ? OP_LOG : nd.op == "Sqrt" ? OP_SQRT : nd.op == "Abs" ? OP_ABS : nd.op == "Neg" ? OP_NEG :
                            nd.op == "Floor" ? OP_FLOOR : nd.op == "Ceil" ? OP_CEIL : nd.op == "HardSwish" ? OP_HARD_SWISH :
                            nd.op == "LeakyRelu" ? OP_LEAKY_RELU : nd.op == "Elu" ? OP_ELU : nd.op == "Gelu" ? OP_GELU : nd.op == "Sin" ? OP_SIN : OP_COS;
            if (nd.op == "Elu" && nd.af("alpha", 1.0f) != 1.0f) return fail("ONNX: " + where + ": alpha != 1 is not supported");
            const int uo = new_act(oname, m.tensors[a].shape);
            if (chl.count(a)) chl.insert(uo);
            if (tr3.count(a)) tr3.insert(uo);
            TflOp& o = add_op(opc, {a}, uo);
            if (nd.op == "LeakyRelu") o.alpha = nd.af("alpha", 0.01f);
            if (nd.op == "Gelu") { const OAttr* ap = nd.attr("approximate"); o.approximate = ap && ap->s == "tanh"; }
        } else if (nd.op == "Clip") {
            const int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            float lo = -INFINITY, hi = INFINITY;
            if (const OAttr* p = nd.attr("min")) lo = p->f;
            if (const OAttr* p = nd.attr("max")) hi = p->f;
            std::vector<float> cv;
            if (nd.in.size() > 1 && !nd.in[1].empty()) { if (!const_f(nd.in[1], &cv, nullptr) || cv.size() != 1) return fail("ONNX: " + where + ": min must be a constant scalar"); lo = cv[0]; }
            if (nd.in.size() > 2 && !nd.in[2].empty()) { if (!const_f(nd.in[2], &cv, nullptr) || cv.size() != 1) return fail("ONNX: " + where + ": max must be a constant scalar"); hi = cv[0]; }
            const auto sh = m.tensors[a].shape;
            const bool cimg = chl.count(a) != 0;
            struct TagOnExit { std::set<int>& c; std::map<std::string, int>& t; const std::string& n; bool on; ~TagOnExit() { auto it = t.find(n); if (on && it != t.end()) c.insert(it->second); } } tag_clip{chl, tid, oname, cimg};
            if (lo == 0.0f && hi == 6.0f) add_op(OP_RELU6, {a}, new_act(oname, sh));
            else if (lo == 0.0f && std::isinf(hi)) add_op(OP_RELU, {a}, new_act(oname, sh));
            else if (lo == -1.0f && hi == 1.0f) add_op(OP_RELU_N1_TO_1, {a}, new_act(oname, sh));
            else {
                int cur = a;
                if (!std::isinf(lo)) { int c = m.add_const_f32(oname + "/min", {1}, {lo}); int t = std::isinf(hi) ? new_act(oname, sh) : new_act(oname + "/lo", sh); add_op(OP_MAXIMUM, {cur, c}, t); cur = t; if (cimg) chl.insert(t); }
                if (!std::isinf(hi)) { int c = m.add_const_f32(oname + "/max", {1}, {hi}); add_op(OP_MINIMUM, {cur, c}, new_act(oname, sh)); }
                if (std::isinf(lo) && std::isinf(hi)) tid[oname] = a;
            }
        } else if (nd.op == "Softmax") {
            int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            a = to_nchw(a);
            const int rank = (int)m.tensors[a].shape.size();
            int64_t ax = nd.ai("axis", -1);
            if (ax < 0) ax += rank;
            if (ax != rank - 1) return fail("ONNX: " + where + ": only softmax over the last axis is supported");
            add_op(OP_SOFTMAX, {a}, new_act(oname, m.tensors[a].shape)).beta = 1.0f;
        } else if (nd.op == "Identity" || nd.op == "Dropout" || nd.op == "Cast") {
            const int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            if (nd.op == "Cast" && nd.ai("to", 1) != 1) return fail("ONNX: " + where + ": only casts to float32 are supported");
            tid[oname] = a;                                   // inference-time no-op
        } else if (nd.op == "Flatten" || nd.op == "Reshape" || nd.op == "Squeeze" || nd.op == "Unsqueeze") {
            int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            a = to_nchw(a);                                   // element order as the graph sees it (free for 1 x 1 images)
            const auto& ish = m.tensors[a].shape;
            const size_t total = m.tensors[a].numel();
            std::vector<int> osh;
            if (nd.op == "Flatten") {
                int64_t ax = nd.ai("axis", 1); if (ax < 0) ax += (int64_t)ish.size();
                if (ax < 1 || ax > (int64_t)ish.size()) return fail("ONNX: " + where + ": axis out of range");
                size_t lead = 1; for (int64_t k = 0; k < ax; k++) lead *= (size_t)ish[k];
                osh = {(int)lead, (int)(total / std::max<size_t>(lead, 1))};
            } else if (nd.op == "Reshape") {
                std::vector<int64_t> sv;
                if (nd.in.size() < 2 || !const_i(nd.in[1], &sv)) return fail("ONNX: " + where + ": shape must be a constant");
                size_t known = 1; int neg = -1;
                for (size_t k = 0; k < sv.size(); k++) {
                    int64_t dv = sv[k];
                    if (dv == 0) dv = k < ish.size() ? ish[k] : 1;
                    if (k == 0 && dv != -1) dv = 1;            // batch dimension
                    if (dv == -1) { if (neg >= 0) return fail("ONNX: " + where + ": more than one -1"); neg = (int)k; osh.push_back(1); }
                    else { osh.push_back((int)dv); known *= (size_t)dv; }
                }
                if (neg >= 0) osh[neg] = (int)(total / std::max<size_t>(known, 1));
            } else {
                std::vector<int64_t> axes;
                if (const OAttr* p = nd.attr("axes")) axes = p->ints;
                else if (nd.in.size() > 1 && !const_i(nd.in[1], &axes)) return fail("ONNX: " + where + ": axes must be constant");
                const int rank_out = nd.op == "Unsqueeze" ? (int)(ish.size() + axes.size()) : (int)ish.size();
                std::vector<char> mark(std::max(rank_out, 1), 0);
                for (auto ax : axes) {
                    if (ax < 0) ax += rank_out;
                    if (ax < 0 || ax >= rank_out) return fail("ONNX: " + where + ": axis out of range");
                    if (mark[ax]) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": duplicate axis"); }
                    mark[ax] = 1;
                }
                if (nd.op == "Unsqueeze") {
                    size_t q = 0;
                    for (int k = 0; k < rank_out; k++) {
                        if (mark[k]) { osh.push_back(1); continue; }
                        if (q >= ish.size()) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": axes do not fit the input rank"); }
                        osh.push_back(ish[q++]);
                    }
                }
                else for (size_t k = 0; k < ish.size(); k++) { if (axes.empty() ? (ish[k] == 1 && k > 0) : mark[k]) { if (ish[k] != 1) return fail("ONNX: " + where + ": squeezed dimension is not 1"); } else osh.push_back(ish[k]); }
            }
            size_t chk = 1; for (int dv : osh) chk *= (size_t)dv;
            if (chk != total || osh.empty() || osh[0] != 1) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": reshape changes the element count or the batch dimension"); }
            add_op(OP_RESHAPE, {a}, new_act(oname, osh)).new_shape = osh;
        } else if (nd.op == "BatchNormalization") {
            int a = in_act(0);
            if (a >= 0 && m.tensors[a].shape.size() == 4) a = to_chl(a);      // channels = axis 1 of the ONNX value = last axis here
            std::vector<float> sc, bi, mu, va;
            if (a < 0 || nd.in.size() < 5 || !const_f(nd.in[1], &sc, nullptr) || !const_f(nd.in[2], &bi, nullptr) || !const_f(nd.in[3], &mu, nullptr) || !const_f(nd.in[4], &va, nullptr))
                return fail("ONNX: " + where + ": scale / bias / mean / var must be constants");
            const auto sh = m.tensors[a].shape;
            const bool bimg = sh.size() == 4 && chl.count(a);     // channels are the last axis either way: [C] constants broadcast
            if (!(sh.size() == 2 || bimg) || (int)sc.size() != sh.back() || bi.size() != sc.size() || mu.size() != sc.size() || va.size() != sc.size())
                return fail("ONNX: " + where + ": only [batch, channels] inputs and convolution outputs are supported");
            const float eps = nd.af("epsilon", 1e-5f);
            std::vector<float> A(sc.size()), Bv(sc.size());
            for (size_t k = 0; k < sc.size(); k++) { A[k] = sc[k] / std::sqrt(va[k] + eps); Bv[k] = bi[k] - mu[k] * A[k]; }
            const int t1 = new_act(oname + "/scaled", sh);
            add_op(OP_MUL, {a, m.add_const_f32(oname + "/a", {(int)sc.size()}, A)}, t1);
            const int t2 = new_act(oname, sh);
            add_op(OP_ADD, {t1, m.add_const_f32(oname + "/b", {(int)sc.size()}, Bv)}, t2);
            if (bimg) { chl.insert(t1); chl.insert(t2); }
        } else if (nd.op == "Concat") {
            std::vector<int> ins;
            bool cimg = false;
            for (size_t k = 0; k < nd.in.size(); k++) { int t = in_act(k); if (t >= 0 && chl.count(t)) cimg = true; }
            if (!cimg) {
                // one-channel planes [N, 1, H, W] stacked along the channel axis (the spectrogram branches of an audio front-end):
                // each is its own channels-last form already, so the stack is built channels-last
                bool planes = nd.ai("axis", 1) == 1 && nd.in.size() >= 2;
                for (size_t k = 0; k < nd.in.size() && planes; k++) { int t = in_act(k); planes = t >= 0 && m.tensors[t].shape.size() == 4 && m.tensors[t].shape[1] == 1; }
                cimg = planes;
            }
            for (size_t k = 0; k < nd.in.size(); k++) {
                int t = cimg ? in_act(k) : operand(nd.in[k]);
                if (t < 0) return fail("ONNX: " + where + (cimg ? ": every operand of an image concatenation must be an activation" : ": operand has no value"));
                if (cimg) { if (m.tensors[t].shape.size() != 4) return fail("ONNX: " + where + ": rank mismatch"); t = to_chl(t); }
                ins.push_back(t);
            }
            std::vector<int> osh = m.tensors[ins[0]].shape;
            int64_t ax = nd.ai("axis", 1); if (ax < 0) ax += (int64_t)osh.size();
            if (ax < 1 || ax >= (int64_t)osh.size()) return fail("ONNX: " + where + ": axis out of range");
            if (cimg) ax = kAxisToChl[ax];
            osh[ax] = 0;
            for (int t : ins) { if (m.tensors[t].shape.size() != osh.size()) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": rank mismatch"); } osh[ax] += m.tensors[t].shape[ax]; }
            const int co = new_act(oname, osh);
            add_op(OP_CONCATENATION, ins, co).axis = (int)ax;
            if (cimg) chl.insert(co);
        } else if (nd.op == "Conv") {
            int a = in_act(0);
            if (a < 0 || nd.in.size() < 2 || (m.tensors[a].shape.size() != 4 && m.tensors[a].shape.size() != 3))
                return fail("ONNX: " + where + ": input must be a rank-3 or rank-4 activation (1-D / 2-D convolution)");
            // Conv1d ([N, C, T], weights [M, C/group, k]): the 2-D operator with a unit height - the clip becomes the channels-last
            // image [1, 1, T, C] (the same memory as the [1, T, C] form), the attributes gain a leading 1 / 0
            const bool c1d = m.tensors[a].shape.size() == 3;
            ONode nd1;
            if (c1d) {
                if (tr3.count(a) == 0 && chl.count(a)) return fail("ONNX: " + where + ": unexpected image operand");
                const int x3 = as_tr3(a);                     // [1, T, C]
                const auto xs = m.tensors[x3].shape;
                const int x4 = new_t(m.tensors[x3].name + "/h1_" + std::to_string(tmp_id++), {1, 1, xs[1], xs[2]}, TT_FLOAT32);
                add_op(OP_RESHAPE, {x3}, x4).new_shape = {1, 1, xs[1], xs[2]};
                chl.insert(x4);
                a = x4;
                nd1 = nd;
                for (auto& at : nd1.attrs) {
                    if ((at.name == "strides" || at.name == "dilations" || at.name == "kernel_shape") && at.ints.size() == 1) at.ints.insert(at.ints.begin(), 1);
                    else if (at.name == "pads" && at.ints.size() == 2) at.ints = {0, at.ints[0], 0, at.ints[1]};
                }
            } else a = to_chl(a);
            const ONode& cn = c1d ? nd1 : nd;
            const auto ish = m.tensors[a].shape;              // [1, H, W, C]
            std::vector<float> Wv; std::vector<int64_t> wd;
            if (!const_f(nd.in[1], &Wv, &wd) || wd.size() != (c1d ? 3u : 4u)) return fail("ONNX: " + where + ": weights must be a constant [M, C/group, kh, kw] tensor");
            if (c1d) wd.insert(wd.begin() + 2, 1);
            const int M = (int)wd[0], Cg = (int)wd[1], kh = (int)wd[2], kw = (int)wd[3], C = ish[3];
            const int64_t group = nd.ai("group", 1);
            auto two = [&](const char* key, int dflt, int* x, int* y) -> bool {
                *x = *y = dflt;
                if (const OAttr* p = cn.attr(key)) { if (p->ints.size() != 2) return false; if (p->ints[0] < 1 || p->ints[0] > (1 << 20) || p->ints[1] < 1 || p->ints[1] > (1 << 20)) return false; *x = (int)p->ints[0]; *y = (int)p->ints[1]; }
                return *x >= 1 && *y >= 1;
            };
            int sh_, sw_, dh, dw;
            if (!two("strides", 1, &sh_, &sw_) || !two("dilations", 1, &dh, &dw)) return fail("ONNX: " + where + ": strides / dilations must have two positive entries");
            if (const OAttr* p = cn.attr("kernel_shape")) if (p->ints.size() != 2 || p->ints[0] != kh || p->ints[1] != kw) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": kernel_shape disagrees with the weights"); }
            if (M < 1 || kh < 1 || kw < 1 || group < 1 || (int64_t)Cg * group != C || M % group) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": weight shape disagrees with the input channels / group"); }
            int Ho, Wo, pt, pl; std::string why;
            if (!window_geom(cn, ish[1], ish[2], kh, kw, sh_, sw_, dh, dw, &Ho, &Wo, &pt, &pl, &why)) return fail("ONNX: " + where + ": " + why);
            std::vector<int> ins = {a};
            int opc;
            int mult = 1;
            if (group == 1) {                                 // OIHW -> OHWI
                std::vector<float> w((size_t)M * kh * kw * C);
                for (int o = 0; o < M; o++) for (int c = 0; c < C; c++) for (int i = 0; i < kh; i++) for (int j = 0; j < kw; j++)
                    w[(((size_t)o * kh + i) * kw + j) * C + c] = Wv[(((size_t)o * C + c) * kh + i) * kw + j];
                ins.push_back(m.add_const_f32(nd.in[1] + "/ohwi", {M, kh, kw, C}, w));
                opc = OP_CONV_2D;
            } else if (group == C && Cg == 1) {               // depthwise (channel multiplier M / C): [M,1,kh,kw] -> [1,kh,kw,M]
                std::vector<float> w((size_t)kh * kw * M);
                for (int o = 0; o < M; o++) for (int i = 0; i < kh; i++) for (int j = 0; j < kw; j++)
                    w[((size_t)i * kw + j) * M + o] = Wv[((size_t)o * kh + i) * kw + j];
                ins.push_back(m.add_const_f32(nd.in[1] + "/1hwm", {1, kh, kw, M}, w));
                opc = OP_DEPTHWISE_CONV_2D; mult = M / C;
            } else return fail("ONNX: " + where + ": grouped convolutions other than depthwise are not supported");
            if (nd.in.size() > 2 && !nd.in[2].empty()) {
                std::vector<float> bv;
                if (!const_f(nd.in[2], &bv, nullptr) || (int)bv.size() != M) return fail("ONNX: " + where + ": bias must be a constant [M] tensor");
                ins.push_back(m.add_const_f32(nd.in[2] + "/b", {M}, bv));
            }
            const int co = c1d ? new_t(oname + "/h1", {1, Ho, Wo, M}, TT_FLOAT32) : new_act(oname, {1, Ho, Wo, M});
            TflOp& o = add_op(opc, ins, co);
            o.stride_h = sh_; o.stride_w = sw_; o.dil_h = dh; o.dil_w = dw; o.depth_multiplier = mult;
            // padding: VALID / TF-SAME where the pads say exactly that (the planner's fused patterns key on them), explicit otherwise
            const int same_h = (ish[1] + sh_ - 1) / sh_, same_w = (ish[2] + sw_ - 1) / sw_;
            const int spt = std::max((same_h - 1) * sh_ + dh * (kh - 1) + 1 - ish[1], 0) / 2, spl = std::max((same_w - 1) * sw_ + dw * (kw - 1) + 1 - ish[2], 0) / 2;
            if (pt == 0 && pl == 0 && Ho == (ish[1] - (dh * (kh - 1) + 1)) / sh_ + 1 && Wo == (ish[2] - (dw * (kw - 1) + 1)) / sw_ + 1) o.padding = 1;
            else if (Ho == same_h && Wo == same_w && pt == spt && pl == spl) o.padding = 0;
            else {
                o.padding = 1; o.explicit_pad = true; o.pad_t = pt; o.pad_l = pl;      // bottom / right follow from the output size
                o.pad_b = std::max((Ho - 1) * sh_ + dh * (kh - 1) + 1 - ish[1] - pt, 0); o.pad_r = std::max((Wo - 1) * sw_ + dw * (kw - 1) + 1 - ish[2] - pl, 0);
            }
            chl.insert(co);
            if (c1d) {                                        // [1, 1, Wo, M] is the [1, Wo, M] form of the ONNX value [N, M, Wo]
                const int r = new_act(oname, {1, Wo, M});
                add_op(OP_RESHAPE, {co}, r).new_shape = {1, Wo, M};
                tr3.insert(r);
            }
        } else if (nd.op == "GlobalAveragePool" || nd.op == "ReduceMean") {
            int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            const bool img = chl.count(a) != 0;
            const int rank = (int)m.tensors[a].shape.size();
            std::vector<int64_t> axes; bool keep = true;
            if (nd.op == "GlobalAveragePool") { if (rank != 4) return fail("ONNX: " + where + ": input must be rank 4"); if (!img) a = to_chl(a); axes = {2, 3}; }
            else {
                keep = nd.ai("keepdims", 1) != 0;
                if (const OAttr* p = nd.attr("axes")) axes = p->ints;
                else if (nd.in.size() > 1 && !nd.in[1].empty()) { if (!const_i(nd.in[1], &axes)) return fail("ONNX: " + where + ": axes must be constant"); }
                else return fail("ONNX: " + where + ": reduction over all axes is not supported");
            }
            const bool cl = chl.count(a) != 0;
            std::vector<int32_t> ax32; std::vector<char> red(rank, 0);
            for (auto ax : axes) { if (ax < 0) ax += rank; if (ax < 1 || ax >= rank || red[ax]) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": bad reduction axis"); } red[ax] = 1; ax32.push_back((int32_t)(cl ? kAxisToChl[ax] : ax)); }
            std::sort(ax32.begin(), ax32.end());
            std::vector<int> osh; const auto ish = m.tensors[a].shape;
            for (int k = 0; k < rank; k++) { const bool r = std::find(ax32.begin(), ax32.end(), k) != ax32.end(); if (r) { if (keep) osh.push_back(1); } else osh.push_back(ish[k]); }
            // without keepdims the result has lower rank: it stays in the graph's element order only when the kept image axes are
            // not mixed, i.e. (for an image) both spatial axes are reduced -> [N, C]
            if (cl && !keep && !(red[2] && red[3] && !red[1])) return fail("ONNX: " + where + ": this reduction of an image without keepdims is not supported");
            const int ro = new_act(oname, osh);
            add_op(OP_MEAN, {a, m.add_const_i32(oname + "/axes", {(int)ax32.size()}, ax32)}, ro).keep_dims = keep;
            if (cl && keep) chl.insert(ro);
        } else if (nd.op == "MaxPool" || nd.op == "AveragePool") {
            int a = in_act(0);
            if (a < 0 || m.tensors[a].shape.size() != 4) return fail("ONNX: " + where + ": input must be a rank-4 activation");
            a = to_chl(a);
            const auto ish = m.tensors[a].shape;
            const OAttr* ks = nd.attr("kernel_shape");
            if (!ks || ks->ints.size() != 2) return fail("ONNX: " + where + ": kernel_shape must have two entries");
            int sh_ = 1, sw_ = 1;
            if (const OAttr* p = nd.attr("strides")) { if (p->ints.size() != 2 || p->ints[0] < 1 || p->ints[0] > (1 << 20) || p->ints[1] < 1 || p->ints[1] > (1 << 20)) return fail("ONNX: " + where + ": strides"); sh_ = (int)p->ints[0]; sw_ = (int)p->ints[1]; }
            if (const OAttr* p = nd.attr("dilations")) for (auto dv : p->ints) if (dv != 1) return fail("ONNX: " + where + ": dilated pooling is not supported");
            if (nd.op == "AveragePool" && nd.ai("count_include_pad", 0) != 0) return fail("ONNX: " + where + ": count_include_pad is not supported");
            if (ks->ints[0] < 1 || ks->ints[0] > (1 << 20) || ks->ints[1] < 1 || ks->ints[1] > (1 << 20)) return fail("ONNX: " + where + ": kernel_shape out of range");
            const int kh = (int)ks->ints[0], kw = (int)ks->ints[1];
            int Ho, Wo, pt, pl; std::string why;
            if (kh < 1 || kw < 1 || sh_ < 1 || sw_ < 1 || !window_geom(nd, ish[1], ish[2], kh, kw, sh_, sw_, 1, 1, &Ho, &Wo, &pt, &pl, &why)) return fail("ONNX: " + where + ": " + why);
            // the pooling kernels know VALID and TF-SAME padding (excluded from the average, as ONNX does by default)
            const int same_h = (ish[1] + sh_ - 1) / sh_, same_w = (ish[2] + sw_ - 1) / sw_;
            const int spt = std::max((same_h - 1) * sh_ + kh - ish[1], 0) / 2, spl = std::max((same_w - 1) * sw_ + kw - ish[2], 0) / 2;
            int padding;
            if (pt == 0 && pl == 0 && Ho == (ish[1] - kh) / sh_ + 1 && Wo == (ish[2] - kw) / sw_ + 1) padding = 1;
            else if (Ho == same_h && Wo == same_w && pt == spt && pl == spl) padding = 0;
            else return fail("ONNX: " + where + ": only VALID and SAME_UPPER-equivalent pooling padding is supported");
            const int po = new_act(oname, {1, Ho, Wo, ish[3]});
            TflOp& o = add_op(nd.op == "MaxPool" ? OP_MAX_POOL_2D : OP_AVERAGE_POOL_2D, {a}, po);
            o.filter_h = kh; o.filter_w = kw; o.stride_h = sh_; o.stride_w = sw_; o.padding = padding;
            chl.insert(po);
        } else if (nd.op == "Transpose") {
            const int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            const int rank = (int)m.tensors[a].shape.size();
            std::vector<int64_t> perm;
            if (const OAttr* p = nd.attr("perm")) perm = p->ints; else for (int k = rank - 1; k >= 0; k--) perm.push_back(k);
            if ((int)perm.size() != rank) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": perm length != rank"); }
            {
                std::vector<char> seen(rank, 0);
                for (auto pk : perm) {
                    if (pk < 0 || pk >= rank || seen[pk]) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": perm is not a permutation"); }
                    seen[pk] = 1;
                }
            }
            if (tr3.count(a) && perm == std::vector<int64_t>{0, 2, 1}) {
                // [N, K, F] -> [N, F, K]: exactly the tensor the tag stands on
                const int t = new_act(oname, m.tensors[a].shape);
                add_op(OP_RESHAPE, {a}, t).new_shape = m.tensors[a].shape;
                return true;
            }
            const bool to_cf = rank == 4 && perm == std::vector<int64_t>{0, 3, 1, 2};     // NHWC data -> NCHW value
            const bool to_cl = rank == 4 && perm == std::vector<int64_t>{0, 2, 3, 1};     // NCHW value -> NHWC data
            if (to_cf && !chl.count(a)) {
                // the result is an NCHW value whose channels-last form is exactly this tensor: no data movement, an alias tensor
                // carries the tag (the source may still be used as a plain tensor elsewhere)
                const int t = new_act(oname, m.tensors[a].shape);
                add_op(OP_RESHAPE, {a}, t).new_shape = m.tensors[a].shape;
                chl.insert(t);
            } else if (to_cl && chl.count(a)) {
                const int t = new_act(oname, m.tensors[a].shape);
                add_op(OP_RESHAPE, {a}, t).new_shape = m.tensors[a].shape;
            } else {
                const int src = to_nchw(untr3(a));
                std::vector<int> osh; std::vector<int32_t> p32;
                for (auto pk : perm) { if (pk < 0 || pk >= rank) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": bad permutation"); } osh.push_back(m.tensors[src].shape[pk]); p32.push_back((int32_t)pk); }
                if (perm[0] != 0) return fail("ONNX: " + where + ": the batch axis cannot move");
                add_op(OP_TRANSPOSE, {src, m.add_const_i32(oname + "/perm", {rank}, p32)}, new_act(oname, osh));
            }
        } else if (nd.op == "HardSigmoid") {
            const int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            const float al = nd.af("alpha", 0.2f), be = nd.af("beta", 0.5f);
            const auto sh = m.tensors[a].shape; const bool img = chl.count(a) != 0;
            const int t1 = new_act(oname + "/ax", sh), t2 = new_act(oname + "/axb", sh), t3 = new_act(oname + "/lo", sh), t4 = new_act(oname, sh);
            add_op(OP_MUL, {a, m.add_const_f32(oname + "/alpha", {1}, {al})}, t1);
            add_op(OP_ADD, {t1, m.add_const_f32(oname + "/beta", {1}, {be})}, t2);
            add_op(OP_MAXIMUM, {t2, m.add_const_f32(oname + "/zero", {1}, {0.0f})}, t3);
            add_op(OP_MINIMUM, {t3, m.add_const_f32(oname + "/one", {1}, {1.0f})}, t4);
            if (img) for (int t : {t1, t2, t3, t4}) chl.insert(t);
        } else if (nd.op == "Pad") {
            int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            const OAttr* md = nd.attr("mode");
            if (md && !md->s.empty() && md->s != "constant") return fail("ONNX: " + where + ": only constant padding is supported");
            const int rank = (int)m.tensors[a].shape.size();
            std::vector<int64_t> pads;
            if (const OAttr* p = nd.attr("pads")) pads = p->ints;
            else if (nd.in.size() < 2 || !const_i(nd.in[1], &pads)) return fail("ONNX: " + where + ": pads must be constant");
            if ((int)pads.size() != 2 * rank) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": pads length != 2 * rank"); }
            if (nd.in.size() > 2 && !nd.in[2].empty()) { std::vector<float> cv; if (!const_f(nd.in[2], &cv, nullptr) || cv.size() != 1 || cv[0] != 0.0f) return fail("ONNX: " + where + ": only zero padding is supported"); }
            if (nd.af("value", 0.0f) != 0.0f) return fail("ONNX: " + where + ": only zero padding is supported");
            const bool img = chl.count(a) != 0;
            std::vector<int32_t> pv(2 * rank, 0); std::vector<int> osh = m.tensors[a].shape;
            for (int k = 0; k < rank; k++) {
                const int64_t lo = pads[k], hi = pads[rank + k];
                if (lo < 0 || hi < 0 || lo > (1 << 24) || hi > (1 << 24) || (k == 0 && (lo || hi))) return fail("ONNX: " + where + ": negative or batch padding is not supported");
                const int q = img ? kAxisToChl[k] : k;
                pv[2 * q] = (int32_t)lo; pv[2 * q + 1] = (int32_t)hi; osh[q] += (int)(lo + hi);
            }
            const int po = new_act(oname, osh);
            add_op(OP_PAD, {a, m.add_const_i32(oname + "/pads", {rank, 2}, pv)}, po);
            if (img) chl.insert(po);
        } else if (nd.op == "ReduceMin" || nd.op == "ReduceMax" || nd.op == "ReduceSum") {
            int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
            a = to_nchw(a);
            const int rank = (int)m.tensors[a].shape.size();
            std::vector<int64_t> axes;
            if (const OAttr* pa = nd.attr("axes")) axes = pa->ints;
            else if (nd.in.size() > 1 && !nd.in[1].empty()) { if (!const_i(nd.in[1], &axes)) return fail("ONNX: " + where + ": axes must be constant"); }
            else return fail("ONNX: " + where + ": reduction over all axes is not supported");
            const bool keep = nd.ai("keepdims", 1) != 0;
            std::vector<int32_t> ax32; std::vector<char> red(rank, 0);
            for (auto ax : axes) { if (ax < 0) ax += rank; if (ax < 1 || ax >= rank || red[ax]) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": bad reduction axis"); } red[ax] = 1; ax32.push_back((int32_t)ax); }
            std::sort(ax32.begin(), ax32.end());
            std::vector<int> osh;
            for (int k = 0; k < rank; k++) { if (red[k]) { if (keep) osh.push_back(1); } else osh.push_back(m.tensors[a].shape[k]); }
            const int ro = new_act(oname, osh);
            add_op(nd.op == "ReduceMin" ? OP_REDUCE_MIN : nd.op == "ReduceMax" ? OP_REDUCE_MAX : OP_SUM, {a, m.add_const_i32(oname + "/axes", {(int)ax32.size()}, ax32)}, ro).keep_dims = keep;
        } else if (nd.op == "Gather") {
            int a = in_act(0);
            if (a < 0 || nd.in.size() < 2) return fail("ONNX: " + where + ": data must be an activation");
            a = to_nchw(a);
            auto it = inits.find(nd.in[1]);
            std::vector<int64_t> idx;
            if (it == inits.end() || !tensor_ints(it->second, &idx)) return fail("ONNX: " + where + ": indices must be a constant integer tensor");
            const auto ish = m.tensors[a].shape;
            const int rank = (int)ish.size();
            int64_t ax = nd.ai("axis", 0); if (ax < 0) ax += rank;
            if (ax < 1 || ax >= rank) return fail("ONNX: " + where + ": axis out of range (the batch axis cannot be gathered)");
            std::vector<int32_t> i32v;
            for (auto v : idx) { if (v < 0) v += ish[ax]; if (v < 0 || v >= ish[ax]) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": index out of range"); } i32v.push_back((int32_t)v); }
            std::vector<int> idims; for (auto dv : it->second.dims) idims.push_back((int)dv);
            std::vector<int> osh(ish.begin(), ish.begin() + ax);
            osh.insert(osh.end(), idims.begin(), idims.end());
            osh.insert(osh.end(), ish.begin() + ax + 1, ish.end());
            if (idims.empty()) {                              // scalar index: the axis disappears
                std::vector<int> gsh(ish.begin(), ish.begin() + ax); gsh.push_back(1); gsh.insert(gsh.end(), ish.begin() + ax + 1, ish.end());
                const int g1 = new_act(oname + "/g", gsh);
                { TflOp& g = add_op(OP_GATHER, {a, m.add_const_i32(oname + "/idx", {1}, i32v)}, g1); g.axis = (int)ax; g.batch_dims = 0; }
                add_op(OP_RESHAPE, {g1}, new_act(oname, osh)).new_shape = osh;
            } else {
                TflOp& g = add_op(OP_GATHER, {a, m.add_const_i32(oname + "/idx", idims, i32v)}, new_act(oname, osh)); g.axis = (int)ax; g.batch_dims = 0;
            }
        } else if (nd.op == "Slice") {
            int a = in_act(0);
            if (a < 0 || nd.in.size() < 3) return fail("ONNX: " + where + ": data must be an activation (opset >= 10 form)");
            const bool t3 = tr3.count(a) != 0;
            if (!t3) a = to_nchw(a);
            const auto ish = m.tensors[a].shape;
            const int rank = (int)ish.size();
            std::vector<int64_t> st, en, axs, sp;
            if (!const_i(nd.in[1], &st) || !const_i(nd.in[2], &en)) return fail("ONNX: " + where + ": starts / ends must be constant");
            if (nd.in.size() > 3 && !nd.in[3].empty()) { if (!const_i(nd.in[3], &axs)) return fail("ONNX: " + where + ": axes must be constant"); }
            else for (size_t k = 0; k < st.size(); k++) axs.push_back((int64_t)k);
            if (nd.in.size() > 4 && !nd.in[4].empty()) { if (!const_i(nd.in[4], &sp)) return fail("ONNX: " + where + ": steps must be constant"); }
            else sp.assign(st.size(), 1);
            if (en.size() != st.size() || axs.size() != st.size() || sp.size() != st.size()) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": starts / ends / axes / steps disagree"); }
            std::vector<int32_t> b(rank, 0), e(rank), s1(rank, 1);
            std::vector<int> osh = ish;
            for (int k = 0; k < rank; k++) e[k] = ish[k];
            int rev_axis = -1; bool plain_rev = st.size() == 1;
            for (size_t q = 0; q < st.size(); q++) {
                int64_t ax = axs[q]; if (ax < 0) ax += rank;
                if (ax < 1 || ax >= rank) return fail("ONNX: " + where + ": axis out of range (the batch axis cannot be sliced)");
                if (t3) ax = ax == 1 ? 2 : 1;                 // ONNX [N, K, F] axis -> axis of the [1, F, K] tensor
                const int64_t n = ish[ax], step = sp[q];
                if (step == 0) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": step 0"); }
                int64_t s0 = st[q], e0 = en[q];
                if (step > 0) {
                    s0 = std::min<int64_t>(std::max<int64_t>(s0 < 0 ? s0 + n : s0, 0), n); e0 = std::min<int64_t>(std::max<int64_t>(e0 < 0 ? e0 + n : e0, 0), n);
                    osh[ax] = (int)std::max<int64_t>(0, (e0 - s0 + step - 1) / step);
                    plain_rev = false;
                } else {
                    s0 = std::min<int64_t>(std::max<int64_t>(s0 < 0 ? s0 + n : s0, -1), n - 1); e0 = std::min<int64_t>(std::max<int64_t>(e0 < 0 ? e0 + n : e0, -1), n - 1);
                    osh[ax] = (int)std::max<int64_t>(0, (s0 - e0 + (-step) - 1) / (-step));
                    if (!(step == -1 && s0 == n - 1 && e0 == -1)) plain_rev = false; else rev_axis = (int)ax;
                }
                if (osh[ax] < 1) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": empty slice"); }
                b[ax] = (int32_t)s0; e[ax] = (int32_t)e0; s1[ax] = (int32_t)step;
            }
            const int so = new_act(oname, osh);
            if (plain_rev && rev_axis >= 0) {
                add_op(OP_REVERSE_V2, {a, m.add_const_i32(oname + "/axis", {1}, {rev_axis})}, so);      // tf ReverseV2 arrives as Slice(step -1)
            } else {
                // negative-step ends of -1 mean "through element 0": expressed with the end mask
                TflOp& o = add_op(OP_STRIDED_SLICE, {a, m.add_const_i32(oname + "/begin", {rank}, b), m.add_const_i32(oname + "/end", {rank}, e),
                                                     m.add_const_i32(oname + "/strides", {rank}, s1)}, so);
                for (int k = 0; k < rank; k++) if (s1[k] < 0 && e[k] < 0) o.end_mask |= 1 << k;
            }
            if (t3) tr3.insert(so);
        } else if (nd.op == "STFT" || nd.op == "DFT") {
            // literal form (no recognised use): framing + one dense layer against the [2K, L] basis, rows interleaved (re_k, im_k)
            int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": signal must be an activation");
            const auto ash = m.tensors[a].shape;
            int frames = -1, L = 0, F = 0;
            if (nd.op == "STFT") {
                std::vector<int64_t> step, flen; std::vector<float> win;
                if (nd.ai("onesided", 1) != 1) return fail("ONNX: " + where + ": only the one-sided transform is supported");
                if (nd.in.size() < 2 || !const_i(nd.in[1], &step) || step.size() != 1 || step[0] < 1) return fail("ONNX: " + where + ": frame_step must be a constant");
                const bool has_w = nd.in.size() > 2 && !nd.in[2].empty();
                if (has_w && !const_f(nd.in[2], &win, nullptr)) return fail("ONNX: " + where + ": window must be a constant");
                int64_t L64 = has_w ? (int64_t)win.size() : 0;
                if (nd.in.size() > 3 && !nd.in[3].empty()) { if (!const_i(nd.in[3], &flen) || flen.size() != 1 || (has_w && flen[0] != L64)) return fail("ONNX: " + where + ": frame_length must be a constant equal to the window length"); L64 = flen[0]; }
                if (!((ash.size() == 3 && ash[2] == 1) || ash.size() == 2) || L64 < 2 || L64 > ash[1]) return fail("ONNX: " + where + ": signal must be [N, T, 1] with T >= frame_length");
                if (step[0] > ash[1]) return fail("ONNX: " + where + ": frame_step exceeds the signal length");      // (int64 check before narrowing: 2^32 would become hop 0)
                L = (int)L64;
                if (!has_w) win.assign(L, 1.0f);
                const int sig = new_t(oname + "/signal", {1, ash[1]}, TT_FLOAT32);
                add_op(OP_RESHAPE, {a}, sig).new_shape = {1, ash[1]};
                frames = emit_frames(sig, ash[1], L, (int)step[0], L, win, oname, &F);
            } else {
                int64_t ax = nd.ai("axis", 1); if (ax < 0) ax += (int64_t)ash.size();
                if (nd.ai("onesided", 0) != 1 || nd.ai("inverse", 0) != 0 || ash.size() != 4 || ash[3] != 1 || ax != 2) return fail("ONNX: " + where + ": only the forward one-sided transform of [N, F, n, 1] along axis 2 is supported");
                L = ash[2]; F = ash[1];
                if (nd.in.size() > 1 && !nd.in[1].empty()) { std::vector<int64_t> dl; if (!const_i(nd.in[1], &dl) || dl.size() != 1 || dl[0] < L || dl[0] > (1 << 20)) return fail("ONNX: " + where + ": dft_length must be a constant >= the axis length"); L = (int)dl[0]; }
                frames = new_t(oname + "/frames", {1, F, ash[2]}, TT_FLOAT32);
                add_op(OP_RESHAPE, {a}, frames).new_shape = {1, F, ash[2]};
            }
            const int K = L / 2 + 1, n_in = m.tensors[frames].shape[2];
            if ((size_t)2 * K * n_in > ((size_t)1 << 27)) return fail("ONNX: " + where + ": transform too large for the dense form");
            std::vector<float> W((size_t)2 * K * n_in);
            for (int k = 0; k < K; k++)
                for (int n = 0; n < n_in; n++) {
                    const double ph = 2.0 * M_PI * (double)(((long)k * n) % L) / L;
                    W[((size_t)2 * k) * n_in + n] = (float)std::cos(ph);
                    W[((size_t)2 * k + 1) * n_in + n] = (float)-std::sin(ph);
                }
            const int y = new_t(oname + "/ri", {1, F, 2 * K}, TT_FLOAT32);
            add_op(OP_FULLY_CONNECTED, {frames, m.add_const_f32(oname + "/basis", {2 * K, n_in}, W)}, y).keep_num_dims = true;
            add_op(OP_RESHAPE, {y}, new_act(oname, {1, F, K, 2})).new_shape = {1, F, K, 2};
        } else {
            return fail("ONNX: unsupported operator " + where);
        }
        return true;
    };
    for (size_t ni = 0; ni < nodes.size(); ni++) {
        node_index = (int)ni;
        if (lowered.count((int)ni)) continue;
        if (!lower_node(nodes[ni], true)) return false;
    }
    for (auto& vi : g_out) if (spec.count(vi.name) && !materialise(vi.name)) return false;
    for (auto& vi : g_out) {
        auto it = tid.find(vi.name);
        if (it == tid.end()) { *code = BNHIP_E_MODEL; return fail("ONNX: graph output is not produced by any node: " + vi.name); }
        m.outputs.push_back(to_nchw(untr3(it->second)));      // an image output leaves in the graph's own (NCHW) order
    }
    // operators nothing reads (the framing emitted for a spectrum that ended up in its literal form, Constant-folded leftovers)
    for (bool again = true; again;) {
        again = false;
        std::vector<int> uses(m.tensors.size(), 0);
        for (auto& o : m.ops) if (o.code != OP_NOP) for (int t : o.inputs) if (t >= 0) uses[t]++;
        for (int t : m.outputs) uses[t]++;
        for (auto& o : m.ops) {
            if (o.code == OP_NOP) continue;
            bool used = false;
            for (int t : o.outputs) if (uses[t]) used = true;
            if (!used) { o.code = OP_NOP; again = true; }
        }
    }
    {
        std::vector<TflOp> live;
        for (auto& o : m.ops) if (o.code != OP_NOP) live.push_back(std::move(o));
        m.ops.swap(live);
    }
    *code = BNHIP_OK;
    return true;
}

}  // namespace bnhip
