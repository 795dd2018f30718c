#include "tflite_model.h"

#include <cstring>

namespace bnhip {

const char* op_name(int code) {
    switch (code) {
        case OP_ADD: return "ADD"; case OP_AVERAGE_POOL_2D: return "AVERAGE_POOL_2D";
        case OP_CONCATENATION: return "CONCATENATION"; case OP_CONV_2D: return "CONV_2D";
        case OP_DEPTHWISE_CONV_2D: return "DEPTHWISE_CONV_2D"; case OP_DEQUANTIZE: return "DEQUANTIZE";
        case OP_FULLY_CONNECTED: return "FULLY_CONNECTED";
        case OP_LOGISTIC: return "LOGISTIC"; case OP_MAX_POOL_2D: return "MAX_POOL_2D"; case OP_MUL: return "MUL";
        case OP_RELU: return "RELU"; case OP_RELU6: return "RELU6"; case OP_RESHAPE: return "RESHAPE";
        case OP_SOFTMAX: return "SOFTMAX"; case OP_PAD: return "PAD"; case OP_GATHER: return "GATHER";
        case OP_TRANSPOSE: return "TRANSPOSE"; case OP_MEAN: return "MEAN"; case OP_SUB: return "SUB";
        case OP_DIV: return "DIV"; case OP_SQUEEZE: return "SQUEEZE"; case OP_STRIDED_SLICE: return "STRIDED_SLICE";
        case OP_CAST: return "CAST"; case OP_EXPAND_DIMS: return "EXPAND_DIMS"; case OP_SUM: return "SUM";
        case OP_POW: return "POW"; case OP_REDUCE_MAX: return "REDUCE_MAX"; case OP_REDUCE_MIN: return "REDUCE_MIN";
        case OP_REVERSE_V2: return "REVERSE_V2"; case OP_HARD_SWISH: return "HARD_SWISH";
        case OP_BATCH_MATMUL: return "BATCH_MATMUL"; case OP_RFFT2D: return "RFFT2D"; case OP_IMAG: return "IMAG";
        case OP_REAL: return "REAL"; case OP_COMPLEX_ABS: return "COMPLEX_ABS";
        default: return "UNKNOWN";
    }
}

namespace {

// Bounds-checked flatbuffer cursor. Any violation sets ok=false; accessors then return zeros.
struct FB {
    const uint8_t* b;
    size_t n;
    bool ok = true;
    std::string why;

    bool in(size_t p, size_t len) {
        if (p > n || len > n - p) { if (ok) { ok = false; why = "offset out of bounds"; } return false; }
        return true;
    }
    uint32_t u32(size_t p) { uint32_t v = 0; if (in(p, 4)) memcpy(&v, b + p, 4); return v; }
    int32_t i32(size_t p) { int32_t v = 0; if (in(p, 4)) memcpy(&v, b + p, 4); return v; }
    uint16_t u16(size_t p) { uint16_t v = 0; if (in(p, 2)) memcpy(&v, b + p, 2); return v; }
    int8_t i8(size_t p) { int8_t v = 0; if (in(p, 1)) memcpy(&v, b + p, 1); return v; }
    float f32(size_t p) { float v = 0; if (in(p, 4)) memcpy(&v, b + p, 4); return v; }

    // absolute position of field `slot` in table at `t`, 0 if absent
    size_t field(size_t t, int slot) {
        int32_t so = i32(t);
        int64_t vt = (int64_t)t - so;
        if (vt < 0 || !in((size_t)vt, 4)) { ok = false; why = "bad vtable"; return 0; }
        uint16_t vsz = u16((size_t)vt);
        size_t off_pos = 4 + 2 * (size_t)slot;
        if (off_pos + 2 > vsz) return 0;
        uint16_t off = u16((size_t)vt + off_pos);
        return off ? t + off : 0;
    }
    size_t indirect(size_t t, int slot) {
        size_t p = field(t, slot);
        if (!p) return 0;
        size_t q = p + u32(p);
        return in(q, 4) ? q : 0;
    }
    // vector: returns data pos, sets len
    size_t vec(size_t t, int slot, size_t elem, size_t* len) {
        *len = 0;
        size_t p = indirect(t, slot);
        if (!p) return 0;
        size_t l = u32(p);
        if (elem && l > (n - p) / elem) { ok = false; why = "vector length out of bounds"; return 0; }
        if (!in(p + 4, l * elem)) return 0;
        *len = l;
        return p + 4;
    }
    std::vector<int> vec_i32(size_t t, int slot) {
        size_t l; size_t p = vec(t, slot, 4, &l);
        std::vector<int> v(l);
        for (size_t i = 0; i < l; i++) v[i] = i32(p + 4 * i);
        return v;
    }
    std::vector<size_t> vec_tables(size_t t, int slot) {
        size_t l; size_t p = vec(t, slot, 4, &l);
        std::vector<size_t> v;
        for (size_t i = 0; i < l; i++) { size_t e = p + 4 * i; size_t q = e + u32(e); if (in(q, 4)) v.push_back(q); }
        return v;
    }
    std::string str(size_t t, int slot) {
        size_t l; size_t p = vec(t, slot, 1, &l);
        return p ? std::string(reinterpret_cast<const char*>(b + p), l) : std::string();
    }
    int scalar_i32(size_t t, int slot, int def) { size_t p = field(t, slot); return p ? i32(p) : def; }
    int scalar_i8(size_t t, int slot, int def) { size_t p = field(t, slot); return p ? i8(p) : def; }
    bool scalar_bool(size_t t, int slot) { size_t p = field(t, slot); return p ? i8(p) != 0 : false; }
    float scalar_f32(size_t t, int slot, float def) { size_t p = field(t, slot); return p ? f32(p) : def; }
};

size_t type_size(int t) {
    switch (t) {
        case TT_FLOAT32: case TT_INT32: return 4;
        case TT_FLOAT16: return 2;
        case TT_INT64: case TT_COMPLEX64: return 8;
        case TT_UINT8: case TT_INT8: return 1;
        default: return 0;
    }
}

}  // namespace

bool parse_tflite(const void* blob, size_t n, TflModel* out, std::string* err) {
    FB fb{reinterpret_cast<const uint8_t*>(blob), n};
    if (!blob || n < 16 || memcmp(fb.b + 4, "TFL3", 4) != 0) {
        *err = "not a TFLite flatbuffer (missing TFL3 identifier)";
        return false;
    }
    size_t root = fb.u32(0);
    if (!fb.in(root, 4)) { *err = "bad root offset"; return false; }
    out->description = fb.str(root, 3);

    std::vector<int> codes;
    for (size_t ct : fb.vec_tables(root, 1)) {
        int dep = fb.scalar_i8(ct, 0, 0);
        int neu = fb.scalar_i32(ct, 3, 0);
        codes.push_back(dep > neu ? dep : neu);
    }
    struct Buf { const uint8_t* p; size_t n; };
    std::vector<Buf> bufs;
    for (size_t bt : fb.vec_tables(root, 4)) {
        size_t l; size_t p = fb.vec(bt, 0, 1, &l);
        bufs.push_back(Buf{p ? fb.b + p : nullptr, l});
    }
    auto sgs = fb.vec_tables(root, 2);
    if (sgs.size() != 1) { *err = "expected exactly one subgraph"; return false; }
    size_t sg = sgs[0];

    for (size_t tt : fb.vec_tables(sg, 0)) {
        TflTensor t;
        t.shape = fb.vec_i32(tt, 0);
        t.type = fb.scalar_i8(tt, 1, 0);
        t.name = fb.str(tt, 3);
        uint32_t bi = (uint32_t)fb.scalar_i32(tt, 2, 0);
        for (int d : t.shape) if (d < 0) { *err = "dynamic tensor shape unsupported: " + t.name; return false; }
        if (bi != 0) {
            if (bi >= bufs.size()) { *err = "tensor buffer index out of range: " + t.name; return false; }
            if (bufs[bi].p && bufs[bi].n) {
                size_t ts = type_size(t.type);
                if (!ts || bufs[bi].n != t.numel() * ts) {
                    *err = "constant tensor size/type mismatch: " + t.name;
                    return false;
                }
                t.data = bufs[bi].p;
                t.nbytes = bufs[bi].n;
            }
        }
        out->tensors.push_back(std::move(t));
    }
    out->inputs = fb.vec_i32(sg, 1);
    out->outputs = fb.vec_i32(sg, 2);
    const int nt = (int)out->tensors.size();

    for (size_t ot : fb.vec_tables(sg, 3)) {
        TflOp o;
        uint32_t ci = (uint32_t)fb.scalar_i32(ot, 0, 0);
        if (ci >= codes.size()) { *err = "opcode index out of range"; return false; }
        o.code = codes[ci];
        o.inputs = fb.vec_i32(ot, 1);
        o.outputs = fb.vec_i32(ot, 2);
        for (int i : o.inputs) if (i < -1 || i >= nt) { *err = "operator input index out of range"; return false; }
        for (int i : o.outputs) if (i < 0 || i >= nt) { *err = "operator output index out of range"; return false; }
        size_t op = fb.indirect(ot, 4);
        if (op) {
            switch (o.code) {
                case OP_CONV_2D:
                    o.padding = fb.scalar_i8(op, 0, 0); o.stride_w = fb.scalar_i32(op, 1, 1);
                    o.stride_h = fb.scalar_i32(op, 2, 1); o.act = fb.scalar_i8(op, 3, 0);
                    o.dil_w = fb.scalar_i32(op, 4, 1); o.dil_h = fb.scalar_i32(op, 5, 1);
                    break;
                case OP_DEPTHWISE_CONV_2D:
                    o.padding = fb.scalar_i8(op, 0, 0); o.stride_w = fb.scalar_i32(op, 1, 1);
                    o.stride_h = fb.scalar_i32(op, 2, 1); o.depth_multiplier = fb.scalar_i32(op, 3, 1);
                    o.act = fb.scalar_i8(op, 4, 0); o.dil_w = fb.scalar_i32(op, 5, 1);
                    o.dil_h = fb.scalar_i32(op, 6, 1);
                    break;
                case OP_AVERAGE_POOL_2D: case OP_MAX_POOL_2D:
                    o.padding = fb.scalar_i8(op, 0, 0); o.stride_w = fb.scalar_i32(op, 1, 1);
                    o.stride_h = fb.scalar_i32(op, 2, 1); o.filter_w = fb.scalar_i32(op, 3, 0);
                    o.filter_h = fb.scalar_i32(op, 4, 0); o.act = fb.scalar_i8(op, 5, 0);
                    break;
                case OP_FULLY_CONNECTED:
                    o.act = fb.scalar_i8(op, 0, 0); o.keep_num_dims = fb.scalar_bool(op, 2);
                    break;
                case OP_ADD: case OP_MUL: case OP_SUB: case OP_DIV:
                    o.act = fb.scalar_i8(op, 0, 0);
                    break;
                case OP_CONCATENATION:
                    o.axis = fb.scalar_i32(op, 0, 0); o.act = fb.scalar_i8(op, 1, 0);
                    break;
                case OP_SOFTMAX: o.beta = fb.scalar_f32(op, 0, 1.0f); break;
                case OP_MEAN: case OP_SUM: case OP_REDUCE_MAX: case OP_REDUCE_MIN:
                    o.keep_dims = fb.scalar_bool(op, 0);
                    break;
                case OP_GATHER: o.axis = fb.scalar_i32(op, 0, 0); o.batch_dims = fb.scalar_i32(op, 1, 0); break;
                case OP_RESHAPE: o.new_shape = fb.vec_i32(op, 0); break;
                case OP_SQUEEZE: o.squeeze_dims = fb.vec_i32(op, 0); break;
                case OP_CAST: o.in_type = fb.scalar_i8(op, 0, 0); o.out_type = fb.scalar_i8(op, 1, 0); break;
                case OP_BATCH_MATMUL: o.adj_x = fb.scalar_bool(op, 0); o.adj_y = fb.scalar_bool(op, 1); break;
                default: break;
            }
        }
        if (o.stride_w <= 0) o.stride_w = 1;
        if (o.stride_h <= 0) o.stride_h = 1;
        if (o.dil_w <= 0) o.dil_w = 1;
        if (o.dil_h <= 0) o.dil_h = 1;
        out->ops.push_back(std::move(o));
    }
    if (!fb.ok) { *err = "malformed flatbuffer: " + fb.why; return false; }
    if (out->inputs.size() != 1) { *err = "expected exactly one graph input"; return false; }
    if (out->outputs.empty()) { *err = "graph has no outputs"; return false; }
    for (int i : out->inputs) if (i < 0 || i >= nt) { *err = "graph input index out of range"; return false; }
    for (int i : out->outputs) if (i < 0 || i >= nt) { *err = "graph output index out of range"; return false; }
    return true;
}

}  // namespace bnhip
