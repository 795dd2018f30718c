// Graph rewrites applied to the parsed model before planning.  The reference hands its TFLite backend any float graph
// (internal/inference/tflite/classifier.go:38-92); a TF -> TFLite export of an EfficientNet-style network routinely
// carries patterns a hand-written planner would otherwise have to special-case everywhere:
//   * float16 constants behind DEQUANTIZE (the reference's own range-filter model ships that way);
//   * unfolded batch norm: per-channel constant MUL / ADD / SUB after a convolution or dense layer;
//   * explicit ZeroPadding2D (PAD) in front of a stride-2 VALID convolution.
// Each pass rewrites the model in place (new constants are owned by the model, removed ops become OP_NOP), so the
// planner's fusion patterns see the same canonical graph whatever the exporter emitted.
#include <cmath>
#include <cstring>

#include "tflite_model.h"

namespace bnhip {

namespace {

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; do { man <<= 1; e++; } while (!(man & 0x400)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13); }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

struct Uses {
    std::vector<std::vector<int>> consumers;   // tensor -> op indices
    std::vector<int> uses;                     // consumer count + graph outputs
    explicit Uses(const TflModel& m) {
        consumers.assign(m.tensors.size(), {});
        uses.assign(m.tensors.size(), 0);
        for (int i = 0; i < (int)m.ops.size(); i++) {
            if (m.ops[i].code == OP_NOP) continue;
            for (int t : m.ops[i].inputs) if (t >= 0) { consumers[t].push_back(i); uses[t]++; }
        }
        for (int t : m.outputs) uses[t]++;
    }
};

// DEQUANTIZE(const float16) -> float32 constant
bool pass_dequantize(TflModel* m, std::string* err) {
    for (auto& o : m->ops) {
        if (o.code != OP_DEQUANTIZE) continue;
        const TflTensor& src = m->tensors[o.inputs[0]];
        if (!src.data || src.type != TT_FLOAT16) { *err = "DEQUANTIZE: only constant float16 inputs are supported"; return false; }
        std::vector<float> v(src.numel());
        const uint16_t* h = reinterpret_cast<const uint16_t*>(src.data);
        for (size_t k = 0; k < v.size(); k++) v[k] = half_to_float(h[k]);
        auto buf = std::make_shared<std::vector<uint8_t>>(v.size() * 4);
        if (!v.empty()) memcpy(buf->data(), v.data(), buf->size());
        m->owned.push_back(buf);
        TflTensor& dst = m->tensors[o.outputs[0]];
        dst.data = buf->data(); dst.nbytes = buf->size(); dst.type = TT_FLOAT32;
        o.code = OP_NOP;
    }
    return true;
}

// per-output-channel constant?  (numel == Co with the channel as the innermost dimension, or a scalar)
bool per_channel_const(const TflModel& m, int t, int Co, std::vector<float>* out) {
    if (t < 0) return false;
    const TflTensor& c = m.tensors[t];
    if (!c.data || c.type != TT_FLOAT32) return false;
    const size_t n = c.numel();
    if (n == 1) { out->assign(Co, c.f32()[0]); return true; }
    if (n != (size_t)Co || c.shape.empty() || c.shape.back() != Co) return false;
    out->assign(c.f32(), c.f32() + Co);
    return true;
}

// y = conv(x) [no fused activation], y used once by  y*c | y+c | y-c | c-y  with a per-channel constant c:
//   fold into the weights / bias, repeat; the last folded op's fused activation moves onto the convolution.
void pass_fold_affine(TflModel* m) {
    for (int oi = 0; oi < (int)m->ops.size(); oi++) {
        const int code = m->ops[oi].code;
        if (code != OP_CONV_2D && code != OP_DEPTHWISE_CONV_2D && code != OP_FULLY_CONNECTED) continue;
        {
            const TflOp& o = m->ops[oi];
            const TflTensor& w0 = m->tensors[o.inputs[1]];
            if (!w0.data || w0.type != TT_FLOAT32) continue;
            if (o.inputs.size() > 2 && o.inputs[2] >= 0 && (!m->tensors[o.inputs[2]].data || m->tensors[o.inputs[2]].type != TT_FLOAT32)) continue;
        }
        const int Co = code == OP_FULLY_CONNECTED ? m->tensors[m->ops[oi].inputs[1]].shape[0] : m->tensors[m->ops[oi].inputs[1]].shape[code == OP_CONV_2D ? 0 : 3];
        std::vector<float> scale(Co, 1.0f), shift(Co, 0.0f);
        bool folded = false, have_scale = false;
        for (;;) {
            if (m->ops[oi].act != 0) break;
            Uses U(*m);
            const int y = m->ops[oi].outputs[0];
            if (U.uses[y] != 1 || U.consumers[y].size() != 1) break;
            const int ci = U.consumers[y][0];
            TflOp& c = m->ops[ci];
            if ((c.code != OP_MUL && c.code != OP_ADD && c.code != OP_SUB) || c.inputs.size() != 2) break;
            const bool y_first = c.inputs[0] == y;
            const int other = y_first ? c.inputs[1] : c.inputs[0];
            if (other == y) break;
            std::vector<float> cv;
            if (!per_channel_const(*m, other, Co, &cv)) break;
            if (m->tensors[c.outputs[0]].numel() != m->tensors[y].numel()) break;      // the constant must not broadcast y up
            if (c.code == OP_MUL) { for (int k = 0; k < Co; k++) { scale[k] *= cv[k]; shift[k] *= cv[k]; } have_scale = true; }
            else if (c.code == OP_ADD) { for (int k = 0; k < Co; k++) shift[k] += cv[k]; }
            else if (y_first) { for (int k = 0; k < Co; k++) shift[k] -= cv[k]; }
            else { for (int k = 0; k < Co; k++) { scale[k] = -scale[k]; shift[k] = cv[k] - shift[k]; } have_scale = true; }
            m->ops[oi].outputs[0] = c.outputs[0];
            m->ops[oi].act = c.act;
            c.code = OP_NOP;
            folded = true;
        }
        if (!folded) continue;
        TflOp& o = m->ops[oi];
        const TflTensor w = m->tensors[o.inputs[1]];
        std::vector<float> bias(Co, 0.0f);
        if (o.inputs.size() > 2 && o.inputs[2] >= 0) memcpy(bias.data(), m->tensors[o.inputs[2]].f32(), (size_t)Co * 4);
        for (int k = 0; k < Co; k++) bias[k] = bias[k] * scale[k] + shift[k];
        if (have_scale) {
            std::vector<float> wn(w.f32(), w.f32() + w.numel());
            if (code == OP_DEPTHWISE_CONV_2D) {               // [1, kh, kw, Co]: channel innermost
                for (size_t i = 0; i < wn.size(); i++) wn[i] *= scale[i % (size_t)Co];
            } else {                                          // [Co, ...]: channel outermost
                const size_t per = wn.size() / (size_t)Co;
                for (size_t i = 0; i < wn.size(); i++) wn[i] *= scale[i / per];
            }
            o.inputs[1] = m->add_const_f32(w.name + "/folded", w.shape, wn);
        }
        const int bt = m->add_const_f32(w.name + "/folded_bias", {Co}, bias);
        if (o.inputs.size() > 2) o.inputs[2] = bt; else { while (o.inputs.size() < 2) o.inputs.push_back(-1); o.inputs.push_back(bt); }
    }
}

// PAD(x, const paddings on H/W only, zeros) -> VALID conv / depthwise conv: the conv takes the padding itself
void pass_fold_pad(TflModel* m) {
    Uses U(*m);
    for (int pi = 0; pi < (int)m->ops.size(); pi++) {
        TflOp& p = m->ops[pi];
        if (p.code != OP_PAD || p.inputs.size() != 2) continue;
        const TflTensor& pt = m->tensors[p.inputs[1]];
        const TflTensor& x = m->tensors[p.inputs[0]];
        if (!pt.data || pt.type != TT_INT32 || x.shape.size() != 4 || pt.numel() != 8) continue;
        const int32_t* pv = pt.i32();
        if (pv[0] || pv[1] || pv[6] || pv[7]) continue;                       // batch / channel padding: not foldable
        bool neg = false;
        for (int k = 0; k < 8; k++) neg |= pv[k] < 0;
        if (neg) continue;
        const int y = p.outputs[0];
        if (U.uses[y] != 1 || U.consumers[y].size() != 1) continue;
        TflOp& c = m->ops[U.consumers[y][0]];
        if ((c.code != OP_CONV_2D && c.code != OP_DEPTHWISE_CONV_2D) || c.padding != 1 /*VALID*/ || c.inputs[0] != y || c.explicit_pad)
            continue;
        c.explicit_pad = true;
        c.pad_t = pv[2]; c.pad_b = pv[3]; c.pad_l = pv[4]; c.pad_r = pv[5];
        c.inputs[0] = p.inputs[0];
        p.code = OP_NOP;
    }
}

// AVERAGE_POOL_2D whose window is the whole image (VALID, any stride) is a global average pool: rewritten to the MEAN over
// (H, W) with keep_dims that Keras' GlobalAveragePooling2D converts to, so that the squeeze-excite and pooling-head patterns
// of the planner see one spelling
void pass_global_pool(TflModel* m) {
    for (auto& o : m->ops) {
        if (o.code != OP_AVERAGE_POOL_2D || o.inputs.size() != 1 || o.outputs.size() != 1 || o.act != 0) continue;
        const auto& ish = m->tensors[o.inputs[0]].shape;
        const auto& osh = m->tensors[o.outputs[0]].shape;
        if (ish.size() != 4 || osh.size() != 4 || o.filter_h != ish[1] || o.filter_w != ish[2] || osh[1] != 1 || osh[2] != 1) continue;
        if (o.padding != 1 /*VALID*/ && !(ish[1] == 1 && ish[2] == 1)) {
            // SAME with a full-size window also yields 1x1 only for stride >= size; partial windows would average fewer pixels
            if (o.stride_h < ish[1] || o.stride_w < ish[2]) continue;
        }
        const int ax = m->add_const_i32(m->tensors[o.outputs[0]].name + "/axes", {2}, {1, 2});
        o.code = OP_MEAN;
        o.inputs.push_back(ax);
        o.keep_dims = true;
    }
}

}  // namespace

bool run_graph_passes(TflModel* m, std::string* err) {
    if (!pass_dequantize(m, err)) return false;
    pass_fold_affine(m);
    pass_fold_pad(m);
    pass_global_pool(m);
    return true;
}

}  // namespace bnhip
