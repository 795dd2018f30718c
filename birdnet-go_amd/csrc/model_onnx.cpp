#include "model_onnx.h"

#include <cmath>
#include <cstring>
#include <map>

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
    return pb.ok;
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

size_t numel(const std::vector<int64_t>& d) { size_t n = 1; for (auto v : d) n *= (size_t)std::max<int64_t>(v, 0); return n; }

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
    for (const ONode& nd : nodes) {
        const std::string where = nd.op + " (" + (nd.name.empty() ? (nd.out.empty() ? "?" : nd.out[0]) : nd.name) + ")";
        if (!nd.domain.empty() && nd.domain != "ai.onnx") return fail("ONNX: operator from unsupported domain " + nd.domain + ": " + where);
        if (nd.op == "Constant") continue;
        if (nd.out.empty() || nd.in.empty()) { *code = BNHIP_E_MODEL; return fail("ONNX: node without inputs/outputs: " + where); }
        const std::string& oname = nd.out[0];
        auto in_act = [&](size_t k) -> int { if (k >= nd.in.size()) return -1; auto it = tid.find(nd.in[k]); return it == tid.end() || m.tensors[it->second].data ? -1 : it->second; };
        if (nd.op == "Gemm" || nd.op == "MatMul") {
            const int a = in_act(0);
            if (a < 0 || nd.in.size() < 2) return fail("ONNX: " + where + ": first operand must be an activation");
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
            const int a = operand(nd.in[0]), b = operand(nd.in[1]);
            if (a < 0 || b < 0) return fail("ONNX: " + where + ": operand is neither an activation nor a float constant");
            std::vector<int> z;
            if (!bshape(m.tensors[a].shape, m.tensors[b].shape, &z)) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": shapes do not broadcast"); }
            const int opc = nd.op == "Add" ? OP_ADD : nd.op == "Sub" ? OP_SUB : nd.op == "Mul" ? OP_MUL : nd.op == "Div" ? OP_DIV :
                            nd.op == "Pow" ? OP_POW : nd.op == "Max" ? OP_MAXIMUM : OP_MINIMUM;
            add_op(opc, {a, b}, new_act(oname, z));
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
            TflOp& o = add_op(opc, {a}, new_act(oname, m.tensors[a].shape));
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
            if (lo == 0.0f && hi == 6.0f) add_op(OP_RELU6, {a}, new_act(oname, sh));
            else if (lo == 0.0f && std::isinf(hi)) add_op(OP_RELU, {a}, new_act(oname, sh));
            else if (lo == -1.0f && hi == 1.0f) add_op(OP_RELU_N1_TO_1, {a}, new_act(oname, sh));
            else {
                int cur = a;
                if (!std::isinf(lo)) { int c = m.add_const_f32(oname + "/min", {1}, {lo}); int t = std::isinf(hi) ? new_act(oname, sh) : new_act(oname + "/lo", sh); add_op(OP_MAXIMUM, {cur, c}, t); cur = t; }
                if (!std::isinf(hi)) { int c = m.add_const_f32(oname + "/max", {1}, {hi}); add_op(OP_MINIMUM, {cur, c}, new_act(oname, sh)); }
                if (std::isinf(lo) && std::isinf(hi)) tid[oname] = a;
            }
        } else if (nd.op == "Softmax") {
            const int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
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
            const int a = in_act(0);
            if (a < 0) return fail("ONNX: " + where + ": operand must be an activation");
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
                for (auto ax : axes) { if (ax < 0) ax += rank_out; if (ax < 0 || ax >= rank_out) return fail("ONNX: " + where + ": axis out of range"); mark[ax] = 1; }
                if (nd.op == "Unsqueeze") { size_t q = 0; for (int k = 0; k < rank_out; k++) osh.push_back(mark[k] ? 1 : ish[q++]); }
                else for (size_t k = 0; k < ish.size(); k++) { if (axes.empty() ? (ish[k] == 1 && k > 0) : mark[k]) { if (ish[k] != 1) return fail("ONNX: " + where + ": squeezed dimension is not 1"); } else osh.push_back(ish[k]); }
            }
            size_t chk = 1; for (int dv : osh) chk *= (size_t)dv;
            if (chk != total || osh.empty() || osh[0] != 1) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": reshape changes the element count or the batch dimension"); }
            add_op(OP_RESHAPE, {a}, new_act(oname, osh)).new_shape = osh;
        } else if (nd.op == "BatchNormalization") {
            const int a = in_act(0);
            std::vector<float> sc, bi, mu, va;
            if (a < 0 || nd.in.size() < 5 || !const_f(nd.in[1], &sc, nullptr) || !const_f(nd.in[2], &bi, nullptr) || !const_f(nd.in[3], &mu, nullptr) || !const_f(nd.in[4], &va, nullptr))
                return fail("ONNX: " + where + ": scale / bias / mean / var must be constants");
            const auto sh = m.tensors[a].shape;
            if (sh.size() != 2 || (int)sc.size() != sh[1] || bi.size() != sc.size() || mu.size() != sc.size() || va.size() != sc.size())
                return fail("ONNX: " + where + ": only [batch, channels] inputs are supported");
            const float eps = nd.af("epsilon", 1e-5f);
            std::vector<float> A(sc.size()), Bv(sc.size());
            for (size_t k = 0; k < sc.size(); k++) { A[k] = sc[k] / std::sqrt(va[k] + eps); Bv[k] = bi[k] - mu[k] * A[k]; }
            const int t1 = new_act(oname + "/scaled", sh);
            add_op(OP_MUL, {a, m.add_const_f32(oname + "/a", {(int)sc.size()}, A)}, t1);
            add_op(OP_ADD, {t1, m.add_const_f32(oname + "/b", {(int)sc.size()}, Bv)}, new_act(oname, sh));
        } else if (nd.op == "Concat") {
            std::vector<int> ins;
            for (auto& nm : nd.in) { int t = operand(nm); if (t < 0) return fail("ONNX: " + where + ": operand has no value"); ins.push_back(t); }
            std::vector<int> osh = m.tensors[ins[0]].shape;
            int64_t ax = nd.ai("axis", 1); if (ax < 0) ax += (int64_t)osh.size();
            if (ax < 1 || ax >= (int64_t)osh.size()) return fail("ONNX: " + where + ": axis out of range");
            osh[ax] = 0;
            for (int t : ins) { if (m.tensors[t].shape.size() != osh.size()) { *code = BNHIP_E_MODEL; return fail("ONNX: " + where + ": rank mismatch"); } osh[ax] += m.tensors[t].shape[ax]; }
            add_op(OP_CONCATENATION, ins, new_act(oname, osh)).axis = (int)ax;
        } else {
            return fail("ONNX: unsupported operator " + where);
        }
    }
    for (auto& vi : g_out) {
        auto it = tid.find(vi.name);
        if (it == tid.end()) { *code = BNHIP_E_MODEL; return fail("ONNX: graph output is not produced by any node: " + vi.name); }
        m.outputs.push_back(it->second);
    }
    *code = BNHIP_OK;
    return true;
}

}  // namespace bnhip
