#include "tflite_model.h"

#include <algorithm>
#include <cstring>

namespace bnhip {

const char* op_name(int code) {
    switch (code) {
        case OP_NOP: return "NOP";
        case OP_ADD: return "ADD"; case OP_AVERAGE_POOL_2D: return "AVERAGE_POOL_2D";
        case OP_CONCATENATION: return "CONCATENATION"; case OP_CONV_2D: return "CONV_2D";
        case OP_DEPTHWISE_CONV_2D: return "DEPTHWISE_CONV_2D"; case OP_DEQUANTIZE: return "DEQUANTIZE";
        case OP_FLOOR: return "FLOOR"; case OP_FULLY_CONNECTED: return "FULLY_CONNECTED";
        case OP_LOGISTIC: return "LOGISTIC"; case OP_MAX_POOL_2D: return "MAX_POOL_2D"; case OP_MUL: return "MUL";
        case OP_RELU: return "RELU"; case OP_RELU_N1_TO_1: return "RELU_N1_TO_1"; case OP_RELU6: return "RELU6";
        case OP_RESHAPE: return "RESHAPE"; case OP_SOFTMAX: return "SOFTMAX"; case OP_TANH: return "TANH";
        case OP_PAD: return "PAD"; case OP_GATHER: return "GATHER";
        case OP_TRANSPOSE: return "TRANSPOSE"; case OP_MEAN: return "MEAN"; case OP_SUB: return "SUB";
        case OP_DIV: return "DIV"; case OP_SQUEEZE: return "SQUEEZE"; case OP_STRIDED_SLICE: return "STRIDED_SLICE";
        case OP_EXP: return "EXP"; case OP_SPLIT: return "SPLIT";
        case OP_CAST: return "CAST"; case OP_MAXIMUM: return "MAXIMUM"; case OP_MINIMUM: return "MINIMUM";
        case OP_NEG: return "NEG"; case OP_PADV2: return "PADV2"; case OP_SLICE: return "SLICE"; case OP_SIN: return "SIN";
        case OP_EXPAND_DIMS: return "EXPAND_DIMS"; case OP_LOG: return "LOG"; case OP_SUM: return "SUM";
        case OP_SQRT: return "SQRT"; case OP_RSQRT: return "RSQRT";
        case OP_POW: return "POW"; case OP_REDUCE_PROD: return "REDUCE_PROD"; case OP_REDUCE_MAX: return "REDUCE_MAX";
        case OP_REDUCE_MIN: return "REDUCE_MIN"; case OP_SQUARE: return "SQUARE"; case OP_LEAKY_RELU: return "LEAKY_RELU";
        case OP_SQUARED_DIFFERENCE: return "SQUARED_DIFFERENCE"; case OP_ABS: return "ABS"; case OP_CEIL: return "CEIL";
        case OP_REVERSE_V2: return "REVERSE_V2"; case OP_COS: return "COS"; case OP_ELU: return "ELU";
        case OP_ROUND: return "ROUND"; case OP_HARD_SWISH: return "HARD_SWISH";
        case OP_BATCH_MATMUL: return "BATCH_MATMUL"; case OP_RFFT2D: return "RFFT2D"; case OP_IMAG: return "IMAG";
        case OP_REAL: return "REAL"; case OP_COMPLEX_ABS: return "COMPLEX_ABS"; case OP_GELU: return "GELU";
        default: return "UNKNOWN";
    }
}

int TflModel::add_const_f32(const std::string& name, const std::vector<int>& shape, const std::vector<float>& v) {
    auto buf = std::make_shared<std::vector<uint8_t>>(v.size() * sizeof(float));
    if (!v.empty()) memcpy(buf->data(), v.data(), buf->size());
    owned.push_back(buf);
    TflTensor t;
    t.name = name; t.shape = shape; t.type = TT_FLOAT32;
    t.data = buf->data(); t.nbytes = buf->size();
    tensors.push_back(std::move(t));
    return (int)tensors.size() - 1;
}

int TflModel::add_const_i32(const std::string& name, const std::vector<int>& shape, const std::vector<int32_t>& v) {
    auto buf = std::make_shared<std::vector<uint8_t>>(v.size() * sizeof(int32_t));
    if (!v.empty()) memcpy(buf->data(), v.data(), buf->size());
    owned.push_back(buf);
    TflTensor t;
    t.name = name; t.shape = shape; t.type = TT_INT32;
    t.data = buf->data(); t.nbytes = buf->size();
    tensors.push_back(std::move(t));
    return (int)tensors.size() - 1;
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
                case OP_STRIDED_SLICE:
                    o.begin_mask = fb.scalar_i32(op, 0, 0); o.end_mask = fb.scalar_i32(op, 1, 0);
                    o.ellipsis_mask = fb.scalar_i32(op, 2, 0); o.new_axis_mask = fb.scalar_i32(op, 3, 0);
                    o.shrink_axis_mask = fb.scalar_i32(op, 4, 0);
                    break;
                case OP_LEAKY_RELU: o.alpha = fb.scalar_f32(op, 0, 0.0f); break;
                case OP_SPLIT: o.num_splits = fb.scalar_i32(op, 0, 0); break;
                case OP_GELU: o.approximate = fb.scalar_bool(op, 0); break;
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

// ------------------------------------------------------------------------------------------------ validation
namespace {
struct OpSpec {
    int code;
    int min_in, max_in;      // operand count bounds (max_in < 0: unbounded)
    int required;            // the first `required` operands must be present (index >= 0)
    unsigned i32_const_mask; // operands that, when constant, must be INT32 (axes, shapes, paddings, selectors)
    unsigned f32_const_mask; // operands that, when constant, must be FLOAT32 (weights, biases, scalars)
};
const OpSpec kSpecs[] = {
    {OP_CONV_2D, 2, 3, 2, 0, 0x6}, {OP_DEPTHWISE_CONV_2D, 2, 3, 2, 0, 0x6}, {OP_FULLY_CONNECTED, 2, 3, 2, 0, 0x6},
    {OP_ADD, 2, 2, 2, 0, 0x3}, {OP_SUB, 2, 2, 2, 0, 0x3}, {OP_MUL, 2, 2, 2, 0, 0x3}, {OP_DIV, 2, 2, 2, 0, 0x3},
    {OP_POW, 2, 2, 2, 0, 0x3}, {OP_MAXIMUM, 2, 2, 2, 0, 0x3}, {OP_MINIMUM, 2, 2, 2, 0, 0x3},
    {OP_SQUARED_DIFFERENCE, 2, 2, 2, 0, 0x3},
    {OP_MEAN, 2, 2, 2, 0x2, 0}, {OP_SUM, 2, 2, 2, 0x2, 0}, {OP_REDUCE_MAX, 2, 2, 2, 0x2, 0}, {OP_REDUCE_MIN, 2, 2, 2, 0x2, 0},
    {OP_REDUCE_PROD, 2, 2, 2, 0x2, 0},
    {OP_RESHAPE, 1, 2, 1, 0x2, 0}, {OP_SQUEEZE, 1, 1, 1, 0, 0}, {OP_EXPAND_DIMS, 2, 2, 2, 0x2, 0},
    {OP_TRANSPOSE, 2, 2, 2, 0x2, 0}, {OP_PAD, 2, 2, 2, 0x2, 0}, {OP_PADV2, 3, 3, 3, 0x2, 0x4},
    {OP_GATHER, 2, 2, 2, 0x2, 0}, {OP_REVERSE_V2, 2, 2, 2, 0x2, 0}, {OP_STRIDED_SLICE, 4, 4, 4, 0xe, 0},
    {OP_SLICE, 3, 3, 3, 0x6, 0}, {OP_SPLIT, 2, 2, 2, 0x1, 0},
    {OP_CONCATENATION, 1, -1, 1, 0, 0}, {OP_RFFT2D, 2, 2, 2, 0x2, 0}, {OP_BATCH_MATMUL, 2, 2, 2, 0, 0x3},
    {OP_AVERAGE_POOL_2D, 1, 1, 1, 0, 0}, {OP_MAX_POOL_2D, 1, 1, 1, 0, 0}, {OP_SOFTMAX, 1, 1, 1, 0, 0},
    {OP_DEQUANTIZE, 1, 1, 1, 0, 0}, {OP_CAST, 1, 1, 1, 0, 0}, {OP_REAL, 1, 1, 1, 0, 0}, {OP_IMAG, 1, 1, 1, 0, 0},
    {OP_COMPLEX_ABS, 1, 1, 1, 0, 0},
    {OP_LOGISTIC, 1, 1, 1, 0, 0x1}, {OP_RELU, 1, 1, 1, 0, 0x1}, {OP_RELU6, 1, 1, 1, 0, 0x1}, {OP_RELU_N1_TO_1, 1, 1, 1, 0, 0x1},
    {OP_TANH, 1, 1, 1, 0, 0x1}, {OP_HARD_SWISH, 1, 1, 1, 0, 0x1}, {OP_EXP, 1, 1, 1, 0, 0x1}, {OP_LOG, 1, 1, 1, 0, 0x1},
    {OP_SQRT, 1, 1, 1, 0, 0x1}, {OP_RSQRT, 1, 1, 1, 0, 0x1}, {OP_ABS, 1, 1, 1, 0, 0x1}, {OP_NEG, 1, 1, 1, 0, 0x1},
    {OP_SQUARE, 1, 1, 1, 0, 0x1}, {OP_LEAKY_RELU, 1, 1, 1, 0, 0x1}, {OP_ELU, 1, 1, 1, 0, 0x1}, {OP_SIN, 1, 1, 1, 0, 0x1},
    {OP_COS, 1, 1, 1, 0, 0x1}, {OP_FLOOR, 1, 1, 1, 0, 0x1}, {OP_CEIL, 1, 1, 1, 0, 0x1}, {OP_ROUND, 1, 1, 1, 0, 0x1},
    {OP_GELU, 1, 1, 1, 0, 0x1},
};
}  // namespace

bool validate_graph(const TflModel& m, std::string* err) {
    const int nt = (int)m.tensors.size();
    auto bad = [&](const TflOp& o, const std::string& what) {
        *err = std::string("malformed graph: ") + op_name(o.code) + " (code " + std::to_string(o.code) + "): " + what;
        return false;
    };
    for (const TflTensor& t : m.tensors) {
        if (t.shape.size() > 6) { *err = "malformed graph: tensor rank > 6: " + t.name; return false; }
        size_t n = 1;
        for (int d : t.shape) {
            if (d < 0) { *err = "malformed graph: negative dimension in " + t.name; return false; }
            if (d && n > ((size_t)1 << 40) / (size_t)d) { *err = "malformed graph: tensor too large: " + t.name; return false; }
            n *= (size_t)d;
        }
    }
    for (int t : m.inputs) if (t < 0 || t >= nt) { *err = "malformed graph: input index out of range"; return false; }
    for (int t : m.outputs) if (t < 0 || t >= nt) { *err = "malformed graph: output index out of range"; return false; }
    for (const TflOp& o : m.ops) {
        if (o.code == OP_NOP) continue;
        if (o.outputs.empty()) return bad(o, "operator has no outputs");
        for (int t : o.outputs) if (t < 0 || t >= nt) return bad(o, "output index out of range");
        for (int t : o.inputs) if (t < -1 || t >= nt) return bad(o, "input index out of range");
        for (int t : o.outputs) if (m.tensors[t].data) return bad(o, "operator writes a constant tensor");
        const OpSpec* sp = nullptr;
        for (const OpSpec& k : kSpecs) if (k.code == o.code) { sp = &k; break; }
        if (!sp) continue;                 // unknown to the table: the planner reports it as unsupported by name
        const int ni = (int)o.inputs.size();
        if (ni < sp->min_in || (sp->max_in >= 0 && ni > sp->max_in))
            return bad(o, "expected " + std::to_string(sp->min_in) + (sp->max_in == sp->min_in ? "" : "+") + " operands, got " + std::to_string(ni));
        for (int i = 0; i < sp->required && i < ni; i++) if (o.inputs[i] < 0) return bad(o, "required operand " + std::to_string(i) + " is absent");
        if (o.code == OP_CONCATENATION) for (int t : o.inputs) if (t < 0) return bad(o, "absent operand");
        for (int i = 0; i < ni && i < 32; i++) {
            const int t = o.inputs[i];
            if (t < 0 || !m.tensors[t].data) continue;
            const TflTensor& c = m.tensors[t];
            if (((sp->i32_const_mask >> i) & 1) && c.type != TT_INT32)
                return bad(o, "constant operand " + std::to_string(i) + " must be int32: " + c.name);
            if (((sp->f32_const_mask >> i) & 1) && c.type != TT_FLOAT32)
                return bad(o, "constant operand " + std::to_string(i) + " must be float32: " + c.name);
        }
        // the dimensions the planner indexes weights by
        if (o.code == OP_CONV_2D || o.code == OP_DEPTHWISE_CONV_2D) {
            const TflTensor& w = m.tensors[o.inputs[1]];
            const TflTensor& y = m.tensors[o.outputs[0]];
            const TflTensor& x = m.tensors[o.inputs[0]];
            if (w.shape.size() != 4) return bad(o, "filter must have rank 4: " + w.name);
            if (x.shape.size() != 4 || y.shape.size() != 4) return bad(o, "input and output must have rank 4");
            const int co = y.shape[3];
            if (o.code == OP_CONV_2D && (w.shape[0] != co || w.shape[3] != x.shape[3]))
                return bad(o, "filter dimensions disagree with the input/output channels: " + w.name);
            if (o.code == OP_DEPTHWISE_CONV_2D && (w.shape[0] != 1 || w.shape[3] != co))
                return bad(o, "depthwise filter dimensions disagree with the output channels: " + w.name);
            if (ni > 2 && o.inputs[2] >= 0 && m.tensors[o.inputs[2]].numel() != (size_t)co)
                return bad(o, "bias length != output channels");
        }
        if (o.code == OP_FULLY_CONNECTED) {
            const TflTensor& w = m.tensors[o.inputs[1]];
            const TflTensor& x = m.tensors[o.inputs[0]];
            if (w.shape.size() != 2) return bad(o, "weights must have rank 2: " + w.name);
            if (x.shape.empty() || x.shape.back() != w.shape[1]) return bad(o, "weights' inner dimension != input width");
            if (m.tensors[o.outputs[0]].numel() % (size_t)std::max(w.shape[0], 1) != 0 || w.shape[0] <= 0)
                return bad(o, "output size is not a multiple of the weight rows");
            if (ni > 2 && o.inputs[2] >= 0 && m.tensors[o.inputs[2]].numel() != (size_t)w.shape[0])
                return bad(o, "bias length != output width");
        }
    }
    return true;
}

}  // namespace bnhip
