#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include <mutex>

#include "../../include/bnhip.h"
#include "hostpipe.h"

namespace bnhip {

#define HIPCHK(call)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            *err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return false;                                                                      \
        }                                                                                      \
    } while (0)

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Planner {
    const TflModel& m;
    std::vector<int> producer;                 // tensor -> op index (-1: const / input)
    std::vector<std::vector<int>> consumers;   // tensor -> op indices
    std::vector<int> uses;                     // consumer count + graph outputs
    std::vector<char> absorbed;                // op consumed by a fusion
    std::string err;
    int code = BNHIP_E_UNSUPPORTED;

    explicit Planner(const TflModel& mm) : m(mm) {
        int nt = (int)m.tensors.size();
        producer.assign(nt, -1);
        consumers.assign(nt, {});
        uses.assign(nt, 0);
        absorbed.assign(m.ops.size(), 0);
        for (int i = 0; i < (int)m.ops.size(); i++) {
            if (m.ops[i].code == OP_NOP) { absorbed[i] = 1; continue; }      // removed by a graph pass
            for (int o : m.ops[i].outputs) producer[o] = i;
            for (int t : m.ops[i].inputs)
                if (t >= 0) { consumers[t].push_back(i); uses[t]++; }
        }
        for (int t : m.outputs) uses[t]++;
    }
    bool is_const(int t) const { return t >= 0 && m.tensors[t].data != nullptr; }
    const TflTensor& T(int t) const { return m.tensors[t]; }
    bool fail(const std::string& s) { if (err.empty()) err = s; return false; }

    static bool shape_op(int code) { return code == OP_RESHAPE || code == OP_EXPAND_DIMS || code == OP_SQUEEZE; }
    // tf.signal.frame slices the signal to a whole number of sub-frames first; when nothing is cut the
    // STRIDED_SLICE is an identity and is treated like a reshape
    bool identity_op(int oi) const {
        const TflOp& o = m.ops[oi];
        if (shape_op(o.code)) return true;
        return o.code == OP_STRIDED_SLICE && !o.inputs.empty() && !o.outputs.empty() &&
               m.tensors[o.inputs[0]].numel() == m.tensors[o.outputs[0]].numel();
    }

    int skip_up(int t) {          // walk producers through pure shape ops
        while (t >= 0 && producer[t] >= 0 && identity_op(producer[t])) {
            absorbed[producer[t]] = 1;
            t = m.ops[producer[t]].inputs[0];
        }
        return t;
    }
    int skip_down(int t) {        // walk single consumers through pure shape ops
        while (consumers[t].size() == 1 && identity_op(consumers[t][0])) {
            absorbed[consumers[t][0]] = 1;
            t = m.ops[consumers[t][0]].outputs[0];
        }
        return t;
    }
    int only_consumer(int t) const { return consumers[t].size() == 1 ? consumers[t][0] : -1; }
    // scalar float const?
    bool const_scalar(int t, float* v) const {
        if (!is_const(t) || T(t).type != TT_FLOAT32 || T(t).numel() != 1) return false;
        *v = T(t).f32()[0];
        return true;
    }
    // binary op with one const-scalar operand: returns the other operand
    int bin_const(const TflOp& o, float* c, bool* const_is_rhs = nullptr) const {
        if (o.inputs.size() != 2) return -1;
        if (const_scalar(o.inputs[1], c)) { if (const_is_rhs) *const_is_rhs = true; return o.inputs[0]; }
        if (const_scalar(o.inputs[0], c)) { if (const_is_rhs) *const_is_rhs = false; return o.inputs[1]; }
        return -1;
    }
};

struct FrontendMatch {
    int L = 0, Lfft = 0, hop = 0, F = 0, n_mels = 0;
    std::vector<float> window;
    int mel_tensor = -1;     // const [n_mels, nbins]
    float p1 = 1.f, p2 = 1.f, eps = 0.f, norm_sub = 0.f, norm_mul = 1.f;
    bool reverse = false;
    bool magnitude = false;  // COMPLEX_ABS instead of the real part
    bool normalize = true;   // per-clip min/max normalisation in front of the framing (false: raw samples)
    int pad_left = 0, pad_right = 0;   // zero samples the graph PADs around the clip before framing
    bool log_compress = false;         // scale * log(max(x, floor)) instead of the two POWs
    float log_floor = 0.f, log_scale = 1.f;
    bool time_major = false; // image is [1, F, n_mels, 1] (no TRANSPOSE) instead of [1, n_mels, F, 1]
    int out_tensor = -1;     // [1, n_mels, F, 1] or [1, F, n_mels, 1]
};

// Recognise one MelSpec branch around RFFT2D op `ri` (see header comment in synth_model.py for the graph).
bool match_frontend(Planner& P, int ri, FrontendMatch* fm) {
    const TflModel& m = P.m;
    const TflOp& R = m.ops[ri];
    if (R.inputs.size() != 2 || !P.is_const(R.inputs[1]) || P.T(R.inputs[1]).numel() != 2)
        return P.fail("RFFT2D: fft_length must be a constant [2]");
    const int32_t* fl = P.T(R.inputs[1]).i32();
    if (fl[0] != 1) return P.fail("RFFT2D: only 1-D transforms (fft_length[0]==1) are supported");
    fm->Lfft = fl[1];
    P.absorbed[ri] = 1;

    // ---- upstream: window MUL <- framing GATHER <- normalisation chain <- graph input
    int t = P.skip_up(R.inputs[0]);
    int pi = P.producer[t];
    // frames shorter than the transform: tf.signal.stft zero-pads them at the end (PAD on the last axis only)
    if (pi >= 0 && m.ops[pi].code == OP_PAD) {
        const TflOp& pd = m.ops[pi];
        const TflTensor& pv = P.T(pd.inputs[1]);
        const int rank = (int)P.T(pd.inputs[0]).shape.size();
        if (!pv.data || (int)pv.numel() != 2 * rank) return P.fail("front-end: frame PAD needs constant paddings");
        for (int d = 0; d < 2 * rank - 1; d++)
            if (pv.i32()[d] != 0) return P.fail("front-end: frames may only be zero-padded at the end of the last axis");
        P.absorbed[pi] = 1;
        t = P.skip_up(pd.inputs[0]);
        pi = P.producer[t];
    }
    int frames_t = t;
    if (pi >= 0 && m.ops[pi].code == OP_MUL) {
        const TflOp& mu = m.ops[pi];
        int wc = P.is_const(mu.inputs[1]) ? 1 : (P.is_const(mu.inputs[0]) ? 0 : -1);
        if (wc < 0) return P.fail("front-end: window MUL without a constant operand");
        const TflTensor& wt = P.T(mu.inputs[wc]);
        if (wt.type != TT_FLOAT32) return P.fail("front-end: window must be float32");
        fm->window.assign(wt.f32(), wt.f32() + wt.numel());
        frames_t = mu.inputs[1 - wc];
        P.absorbed[pi] = 1;
    }
    t = P.skip_up(frames_t);
    pi = P.producer[t];
    if (pi < 0 || m.ops[pi].code != OP_GATHER) return P.fail("front-end: framing pattern (GATHER) not found");
    {
        const TflOp& ga = m.ops[pi];
        if (ga.axis != 1 || ga.batch_dims != 0 || !P.is_const(ga.inputs[1]) || P.T(ga.inputs[1]).shape.size() != 2)
            return P.fail("front-end: unsupported GATHER framing");
        const TflTensor& sel = P.T(ga.inputs[1]);
        const TflTensor& par = P.T(ga.inputs[0]);
        if (par.shape.size() != 3) return P.fail("front-end: GATHER params must be [1, n_sub, sub]");
        int sub = par.shape[2], F = sel.shape[0], Q = sel.shape[1];
        const int32_t* sv = sel.i32();
        int step = F > 1 ? sv[Q] - sv[0] : 1;
        for (int f = 0; f < F; f++)
            for (int q = 0; q < Q; q++)
                if (sv[f * Q + q] != f * step + q) return P.fail("front-end: GATHER selector is not a sliding window");
        fm->F = F; fm->hop = step * sub; fm->L = Q * sub;
        P.absorbed[pi] = 1;
        t = P.skip_up(ga.inputs[0]);
    }
    if (fm->window.empty()) fm->window.assign(fm->L, 1.0f);
    if ((int)fm->window.size() != fm->L) return P.fail("front-end: window length != frame length");
    if (fm->L > fm->Lfft) return P.fail("front-end: frame length > fft length");

    // optional zero padding of the whole clip ([1, n] -> [1, pad_left + n + pad_right])
    if (P.producer[t] >= 0 && m.ops[P.producer[t]].code == OP_PAD) {
        const TflOp& pd = m.ops[P.producer[t]];
        const TflTensor& pv = P.T(pd.inputs[1]);
        const auto& ish = P.T(pd.inputs[0]).shape;
        if (!pv.data || pv.numel() != 4 || ish.size() != 2 || pv.i32()[0] != 0 || pv.i32()[1] != 0 || pv.i32()[2] < 0 || pv.i32()[3] < 0)
            return P.fail("front-end: clip PAD must be constant [[0,0],[left,right]] on a [1, n] tensor");
        fm->pad_left = pv.i32()[2]; fm->pad_right = pv.i32()[3];
        P.absorbed[P.producer[t]] = 1;
        t = P.skip_up(pd.inputs[0]);
    }
    // normalisation: MUL(SUB(DIV(SUB(x, REDUCE_MIN x), ADD(REDUCE_MAX(.), eps)), c_sub), c_mul) - or none at all
    if (t == m.inputs[0]) {
        fm->normalize = false;
    } else {
        int p_mul = P.producer[t];
        if (p_mul < 0 || m.ops[p_mul].code != OP_MUL) return P.fail("front-end: normalisation (MUL) not found");
        int n2 = P.bin_const(m.ops[p_mul], &fm->norm_mul);
        if (n2 < 0) return P.fail("front-end: normalisation MUL needs a scalar constant");
        int p_sub = P.producer[n2];
        bool rhs = false;
        // (x - c), or the (x + (-c)) a converter may rewrite it to
        if (p_sub < 0 || (m.ops[p_sub].code != OP_SUB && m.ops[p_sub].code != OP_ADD)) return P.fail("front-end: normalisation (SUB c) not found");
        int nm = P.bin_const(m.ops[p_sub], &fm->norm_sub, &rhs);
        if (nm < 0 || (m.ops[p_sub].code == OP_SUB && !rhs)) return P.fail("front-end: normalisation SUB needs a scalar constant rhs");
        if (m.ops[p_sub].code == OP_ADD) fm->norm_sub = -fm->norm_sub;
        int p_div = P.producer[nm];
        if (p_div < 0 || m.ops[p_div].code != OP_DIV) return P.fail("front-end: normalisation (DIV) not found");
        int s1 = m.ops[p_div].inputs[0], dn = m.ops[p_div].inputs[1];
        int p_add = P.producer[dn];
        if (p_add < 0 || m.ops[p_add].code != OP_ADD) return P.fail("front-end: normalisation (ADD eps) not found");
        int mx = P.bin_const(m.ops[p_add], &fm->eps);
        if (mx < 0) return P.fail("front-end: normalisation ADD needs a scalar constant");
        int p_max = P.producer[mx];
        if (p_max < 0 || m.ops[p_max].code != OP_REDUCE_MAX || m.ops[p_max].inputs[0] != s1)
            return P.fail("front-end: normalisation (REDUCE_MAX of shifted signal) not found");
        int p_s1 = P.producer[s1];
        if (p_s1 < 0 || m.ops[p_s1].code != OP_SUB) return P.fail("front-end: normalisation (SUB min) not found");
        int x = m.ops[p_s1].inputs[0], mn = m.ops[p_s1].inputs[1];
        int p_min = P.producer[mn];
        if (p_min < 0 || m.ops[p_min].code != OP_REDUCE_MIN || m.ops[p_min].inputs[0] != x)
            return P.fail("front-end: normalisation (REDUCE_MIN) not found");
        if (x != m.inputs[0]) return P.fail("front-end: normalisation does not start at the graph input");
        for (int op : {p_mul, p_sub, p_div, p_add, p_max, p_s1, p_min}) P.absorbed[op] = 1;
    }

    // ---- downstream: real part -> mel matmul -> POW(s) -> REVERSE -> TRANSPOSE -> [1, n_mels, F, 1]
    t = P.skip_down(R.outputs[0]);
    int ci = P.only_consumer(t);
    if (ci < 0) return P.fail("front-end: STFT output must have one consumer");
    if (m.ops[ci].code == OP_COMPLEX_ABS) fm->magnitude = true;
    else if (!(m.ops[ci].code == OP_CAST || m.ops[ci].code == OP_REAL)) return P.fail("front-end: expected CAST/REAL/COMPLEX_ABS after RFFT2D");
    P.absorbed[ci] = 1;
    t = P.skip_down(m.ops[ci].outputs[0]);
    ci = P.only_consumer(t);
    if (ci < 0 || (m.ops[ci].code != OP_FULLY_CONNECTED && m.ops[ci].code != OP_BATCH_MATMUL))
        return P.fail("front-end: mel projection (FULLY_CONNECTED / BATCH_MATMUL) not found");
    {
        const TflOp& fc = m.ops[ci];
        if (!P.is_const(fc.inputs[1]) || (fc.inputs.size() > 2 && fc.inputs[2] >= 0))
            return P.fail("front-end: mel projection must have constant weights and no bias");
        const TflTensor& w = P.T(fc.inputs[1]);
        const int nbins = fm->Lfft / 2 + 1;
        // FULLY_CONNECTED keeps [n_mels, bins]; tf.tensordot may also arrive as BATCH_MATMUL with the [bins, n_mels] matrix
        // (or its transpose with adj_y) on the right
        const bool bmm = fc.code == OP_BATCH_MATMUL;
        if (bmm && fc.adj_x) return P.fail("front-end: mel BATCH_MATMUL with adj_x is not supported");
        const bool rows_are_mels = !bmm || fc.adj_y;
        if (w.shape.size() != 2 || w.shape[rows_are_mels ? 1 : 0] != nbins || w.type != TT_FLOAT32)
            return P.fail("front-end: mel matrix shape mismatch");
        fm->n_mels = w.shape[rows_are_mels ? 0 : 1];
        if (rows_are_mels) fm->mel_tensor = fc.inputs[1];
        else {                                              // re-lay [bins, n_mels] -> [n_mels, bins]
            std::vector<float> wt((size_t)fm->n_mels * nbins);
            for (int k = 0; k < nbins; k++)
                for (int mm = 0; mm < fm->n_mels; mm++) wt[(size_t)mm * nbins + k] = w.f32()[(size_t)k * fm->n_mels + mm];
            fm->mel_tensor = const_cast<TflModel&>(m).add_const_f32(w.name + "/T", {fm->n_mels, nbins}, wt);
        }
        P.absorbed[ci] = 1;
        t = P.skip_down(fc.outputs[0]);
    }
    // log compression: MUL(LOG(MAXIMUM(x, floor)), scale)
    ci = P.only_consumer(t);
    if (ci >= 0 && m.ops[ci].code == OP_MAXIMUM) {
        int src = P.bin_const(m.ops[ci], &fm->log_floor);
        if (src < 0 || !(fm->log_floor > 0.f)) return P.fail("front-end: log compression needs MAXIMUM with a positive scalar floor");
        P.absorbed[ci] = 1;
        t = P.skip_down(m.ops[ci].outputs[0]);
        ci = P.only_consumer(t);
        if (ci < 0 || m.ops[ci].code != OP_LOG) return P.fail("front-end: LOG after MAXIMUM not found");
        P.absorbed[ci] = 1;
        t = P.skip_down(m.ops[ci].outputs[0]);
        fm->log_compress = true;
        ci = P.only_consumer(t);
        if (ci >= 0 && m.ops[ci].code == OP_MUL && P.bin_const(m.ops[ci], &fm->log_scale) >= 0) {
            P.absorbed[ci] = 1;
            t = P.skip_down(m.ops[ci].outputs[0]);
        } else fm->log_scale = 1.f;
    }
    int npow = 0;
    // the square may arrive as POW(x, 2), SQUARE(x) or MUL(x, x)
    if (!fm->log_compress) {
        ci = P.only_consumer(t);
        const bool mul_self = ci < 0 && P.consumers[t].size() == 2 && P.consumers[t][0] == P.consumers[t][1] &&
                              m.ops[P.consumers[t][0]].code == OP_MUL && m.ops[P.consumers[t][0]].inputs[0] == t && m.ops[P.consumers[t][0]].inputs[1] == t;
        if (mul_self) ci = P.consumers[t][0];
        if (ci >= 0 && (m.ops[ci].code == OP_SQUARE || mul_self)) {
            fm->p1 = 2.0f; npow = 1;
            P.absorbed[ci] = 1;
            t = P.skip_down(m.ops[ci].outputs[0]);
        }
    }
    while (!fm->log_compress && (ci = P.only_consumer(t)) >= 0 && m.ops[ci].code == OP_POW && npow < 2) {
        float e;
        if (!P.const_scalar(m.ops[ci].inputs[1], &e)) return P.fail("front-end: POW exponent must be a scalar constant");
        (npow == 0 ? fm->p1 : fm->p2) = e;
        npow++;
        P.absorbed[ci] = 1;
        t = P.skip_down(m.ops[ci].outputs[0]);
    }
    ci = P.only_consumer(t);
    if (ci >= 0 && m.ops[ci].code == OP_REVERSE_V2) {
        const TflTensor& ax = P.T(m.ops[ci].inputs[1]);
        int rank = (int)P.T(m.ops[ci].inputs[0]).shape.size();
        if (!ax.data || ax.numel() != 1 || ((ax.i32()[0] + rank) % rank) != rank - 1)
            return P.fail("front-end: REVERSE_V2 must flip the mel axis");
        fm->reverse = true;
        P.absorbed[ci] = 1;
        t = P.skip_down(m.ops[ci].outputs[0]);
        ci = P.only_consumer(t);
    }
    {
        // time-major image: the [1, F, n_mels] tensor is only reshaped to [1, F, n_mels, 1]
        const auto& ts = P.T(t).shape;
        if (!fm->reverse && ts.size() == 4 && ts[0] == 1 && ts[1] == fm->F && ts[2] == fm->n_mels && ts[3] == 1) {
            fm->time_major = true;
            fm->out_tensor = t;
            return true;
        }
    }
    if (ci < 0 || m.ops[ci].code != OP_TRANSPOSE) return P.fail("front-end: TRANSPOSE to [mel, time] not found");
    {
        const TflTensor& pm = P.T(m.ops[ci].inputs[1]);
        if (!pm.data || pm.numel() != 3 || pm.i32()[0] != 0 || pm.i32()[1] != 2 || pm.i32()[2] != 1)
            return P.fail("front-end: unsupported TRANSPOSE permutation");
        P.absorbed[ci] = 1;
        t = P.skip_down(m.ops[ci].outputs[0]);
    }
    const auto& os = P.T(t).shape;
    if (os.size() != 4 || os[0] != 1 || os[1] != fm->n_mels || os[2] != fm->F || os[3] != 1)
        return P.fail("front-end: unexpected spectrogram tensor shape");
    fm->out_tensor = t;
    return true;
}

// G[n][m'] = sum_k cos(2 pi k n / Lfft) * Mel[k][m]   (fp64; the window is applied separately in fp32)
std::vector<double> build_G(const Planner& P, const FrontendMatch& fm, int Kp, int NTP) {
    const int nb = fm.Lfft / 2 + 1, nm = fm.n_mels, N = fm.Lfft;
    const float* melT = P.T(fm.mel_tensor).f32();      // [n_mels][nbins]
    std::vector<double> ctab(N);
    for (int i = 0; i < N; i++) ctab[i] = std::cos(2.0 * M_PI * (double)i / (double)N);
    std::vector<int> krows;                             // bins with any non-zero mel weight (DFT truncation)
    for (int k = 0; k < nb; k++) {
        bool nz = false;
        for (int mm = 0; mm < nm && !nz; mm++) nz = melT[(size_t)mm * nb + k] != 0.0f;
        if (nz) krows.push_back(k);
    }
    std::vector<double> melk(krows.size() * (size_t)nm);
    for (size_t r = 0; r < krows.size(); r++)
        for (int mm = 0; mm < nm; mm++) melk[r * nm + mm] = (double)melT[(size_t)mm * nb + krows[r]];
    std::vector<double> G((size_t)Kp * NTP, 0.0);     // rows n' = 0..Lfft/2 (cos symmetry folds the rest)
    std::vector<double> row(nm);
    for (int n = 0; n <= fm.Lfft / 2; n++) {
        std::fill(row.begin(), row.end(), 0.0);
        for (size_t r = 0; r < krows.size(); r++) {
            double c = ctab[(size_t)(((long long)krows[r] * n) % N)];
            const double* mk = &melk[r * nm];
            for (int mm = 0; mm < nm; mm++) row[mm] += c * mk[mm];
        }
        for (int mo = 0; mo < nm; mo++) {
            int mm = fm.reverse ? nm - 1 - mo : mo;
            G[(size_t)n * NTP + mo] = row[mm];
        }
    }
    return G;
}

}  // namespace

// ================================================================================================ build
Engine::~Engine() {
    if (device >= 0) hipSetDevice(device);
    drop_graphs();
    for (auto& e : prof) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto e : ev_pool) hipEventDestroy(e);
    for (auto& p : step_ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (mm_scratch) hipFree(mm_scratch);
    if (d_hand) hipFree(d_hand);
    for (void* p : {(void*)act_arena, (void*)w_arena, (void*)d_stage_in, (void*)d_stage_logits, (void*)d_stage_emb,
                    (void*)d_stage_pcm, (void*)d_post_conf, (void*)d_topk_conf, (void*)d_topk_idx})
        if (p) hipFree(p);
    hostpipe_free(hostpipe);
    for (int i = 0; i < kMaxKStreams; i++) if (kstream[i]) hipStreamSynchronize(kstream[i]);
    for (int c = 0; c < kMaxDepth; c++) {
        if (c > 0 && ctx_arena[c]) hipFree(ctx_arena[c]);
        if (ev_ctx_done[c]) hipEventDestroy(ev_ctx_done[c]);
    }
    if (ev_ctx_fork) hipEventDestroy(ev_ctx_fork);
    for (int i = 0; i < kMaxLanes - 1; i++) if (ev_join[i]) hipEventDestroy(ev_join[i]);
    if (ev_fork) hipEventDestroy(ev_fork);
    release_streams();
}

int pw_switches_from_env() {
    auto sw = [](const char* name, int off, int force) {
        const char* e = getenv(name);
        return !e ? 0 : e[0] == '0' ? off : e[0] == '2' ? force : 0;
    };
    return sw("BNHIP_PW_B16", PW_SW_B16_OFF, PW_SW_B16_FORCE) | sw("BNHIP_PW_B16S", PW_SW_B16S_OFF, PW_SW_B16S_FORCE) |
           sw("BNHIP_PW_WS", PW_SW_WS_OFF, PW_SW_WS_FORCE) | sw("BNHIP_PW_LAT", PW_SW_LAT_OFF, 0);
}

bool Engine::build(TflModel m, int dev, int maxb, bool plan_only, std::string* err, int* code) {
    device = dev;
    max_batch = maxb;
    pw_sw = pw_switches_from_env();      // (tests flip these between engines of one process; nothing reads them after this line)
    *code = BNHIP_E_UNSUPPORTED;
    // graph rewrites first (float16 constants behind DEQUANTIZE, unfolded batch norm, PAD + VALID convolutions): the
    // patterns below then see one canonical form whatever the exporter emitted
    if (!run_graph_passes(&m, err)) { *code = BNHIP_E_UNSUPPORTED; return false; }
    // Which graph outputs are the logits and the embedding: the reference decides by model family from the input length and
    // the number / size of the outputs (internal/inference/onnx/detection.go:24-112: v2.4 logits 0 [+ embedding 1]; BirdNET v3.0
    // (160000 samples, 2 outputs) the 1280-wide port is the embedding, the other the predictions; Perch v2 (160000 samples, 4
    // outputs: embedding, spatial embedding, spectrogram, logits) logits 3, embedding 0).  Explicit options win.  The boundary
    // returns logits + embedding only, so every other output is dropped here and never computed.
    if (!m.inputs.empty() && !m.outputs.empty()) {
        const size_t n_in = m.tensors[m.inputs[0]].numel();
        const int n_out = (int)m.outputs.size();
        auto last_dim = [&](int oi) { const auto& sh = m.tensors[m.outputs[oi]].shape; return sh.empty() ? 0 : sh.back(); };
        int li = 0, ei = n_out > 1 ? 1 : -1;
        if (n_in == 160000 && n_out == 4) { li = 3; ei = 0; }
        else if (n_in == 160000 && n_out == 2) { if (last_dim(0) == 1280) { ei = 0; li = 1; } else { ei = 1; li = 0; } }
        if (logits_output >= 0) li = logits_output;
        if (embedding_output != -2) ei = embedding_output;
        if (li < 0 || li >= n_out || ei >= n_out || ei == li) { *code = BNHIP_E_INVALID; *err = "logits_output / embedding_output do not name two distinct graph outputs"; return false; }
        std::vector<int> keep = {m.outputs[li]};
        if (ei >= 0) keep.push_back(m.outputs[ei]);
        logits_output = li; embedding_output = ei;
        m.outputs = keep;
    }
    Planner P(m);

    const TflTensor& tin = m.tensors[m.inputs[0]];
    if (tin.type != TT_FLOAT32 || tin.shape.size() < 2 || tin.shape[0] != 1 || tin.numel() == 0 || tin.numel() > ((size_t)1 << 30)) {
        *err = "graph input must be float32 [1, ...] (one clip / one feature row per batch entry)";
        return false;
    }
    n_samples = (int)tin.numel();      // [1, n_samples] for the audio models; any [1, ...] block for generic graphs

    // ---------------------------------------------------------------- front-end
    std::vector<FrontendMatch> fms;
    for (int i = 0; i < (int)m.ops.size(); i++)
        if (m.ops[i].code == OP_RFFT2D) {
            FrontendMatch fm;
            if (!match_frontend(P, i, &fm)) { *err = P.err; *code = P.code; return false; }
            fms.push_back(fm);
        }
    // Graphs without an STFT are accepted when they are plain dense stacks on a [1,D] input: the bat heads
    // (CustomClassifier, internal/inference/backend.go:31-52) and the range-filter meta-model (RangeFilter, :55-76).
    const bool dense_only = fms.empty();
    int spec_tensor = -1;
    std::vector<int> chan_of(fms.size(), 0);
    if (!dense_only) {
        int cc = P.only_consumer(fms[0].out_tensor);
        if (fms.size() == 1 && (cc < 0 || m.ops[cc].code != OP_CONCATENATION)) {
            spec_tensor = fms[0].out_tensor;
        } else {
            if (cc < 0 || m.ops[cc].code != OP_CONCATENATION || m.ops[cc].axis != 3) {
                *err = "front-end: spectrogram channels must be concatenated on axis 3";
                return false;
            }
            const TflOp& cat = m.ops[cc];
            if (cat.inputs.size() != fms.size()) { *err = "front-end: concat arity != number of STFT branches"; return false; }
            for (size_t i = 0; i < fms.size(); i++) {
                auto it = std::find(cat.inputs.begin(), cat.inputs.end(), fms[i].out_tensor);
                if (it == cat.inputs.end()) { *err = "front-end: STFT branch does not feed the concat"; return false; }
                chan_of[i] = (int)(it - cat.inputs.begin());
            }
            P.absorbed[cc] = 1;
            spec_tensor = cat.outputs[0];
        }
    }
    C_spec = (int)fms.size();
    for (auto& fm : fms)
        if (fm.n_mels != fms[0].n_mels || fm.F != fms[0].F) { *err = "front-end: branches disagree on [mel, time] shape"; return false; }

    // ---------------------------------------------------------------- weights arena (host image first)
    std::vector<float> wimg;
    auto wpush = [&](const float* src, size_t n) -> size_t {
        size_t off = align_up(wimg.size(), 64);       // 256-byte alignment
        wimg.resize(off + n);
        memcpy(&wimg[off], src, n * sizeof(float));
        return off;
    };
    std::vector<std::pair<const float**, size_t>> wfix;   // pointer slots to patch after upload
    std::map<int, size_t> wcache;                      // tflite tensor -> offset (plain upload)
    auto wconst = [&](int t) -> size_t {
        auto it = wcache.find(t);
        if (it != wcache.end()) return it->second;
        size_t off = wpush(m.tensors[t].f32(), m.tensors[t].numel());
        wcache[t] = off;
        return off;
    };

    // values
    std::map<int, int> tv;    // tflite tensor -> value id
    auto new_val = [&](int tfl, size_t elems) {
        Value v; v.tfl = tfl; v.elems = elems;
        vals.push_back(v);
        return (int)vals.size() - 1;
    };
    v_input = new_val(m.inputs[0], n_samples);
    vals[v_input].external = true;
    tv[m.inputs[0]] = v_input;
    v_mm = new_val(-1, 2);

    std::vector<size_t> step_w[4];   // per-step weight offsets (SIZE_MAX = none)
    std::vector<size_t> step_bx;     // per-step offset of the split-bf16 weight image (S_PW steps of a bf16x3 engine)
    auto add_step = [&](Step s, size_t o0 = SIZE_MAX, size_t o1 = SIZE_MAX, size_t o2 = SIZE_MAX, size_t o3 = SIZE_MAX) {
        steps.push_back(s);
        step_w[0].push_back(o0); step_w[1].push_back(o1); step_w[2].push_back(o2); step_w[3].push_back(o3);
        size_t obx = SIZE_MAX;
        if (bf16x3 && s.kind == S_PW && o0 != SIZE_MAX && pw_bx3_ok(s.C)) {
            // three exact bf16 pieces of every weight, in the kernel's slab / lane order (1.5x the fp32 bytes)
            std::vector<float> wsrc(wimg.begin() + o0, wimg.begin() + o0 + (size_t)s.Co * s.C);
            std::vector<uint16_t> img = pw_bx3_image(wsrc.data(), s.Co, s.C);
            std::vector<float> asf((img.size() + 1) / 2);
            memcpy(asf.data(), img.data(), img.size() * 2);
            obx = wpush(asf.data(), asf.size());
            // WHICH arithmetic a layer runs is decided here, by shape, never by a timing (with a create-time race
            // between the two kernels the same clip could get different logits on different devices of one handle, and
            // from process to process): the split-bf16 kernel where the layer is compute-bound - algorithmic intensity at
            // max_batch >= 12 flop/B; measured at batch 256 the split wins every layer above that line by 19-32 % and ties
            // (+-4 %) the HBM-bound early projections below it (b1-b3: 5-10 flop/B) - everywhere with bf16x3 = 2 or
            // "precision":"bf16".  The autotuner then only picks the TILE inside that kernel family.
            const double Mr = (double)maxb * s.H * s.W, Nr = s.Co, Kr = s.C;
            const double intensity = 2.0 * Mr * Nr * Kr / (4.0 * (Mr * Kr + Mr * Nr + Nr * Kr));
            if (bf16x3 >= 2 || precision == 1 || intensity >= 12.0) { steps.back().bx = 1; steps.back().wm = steps.back().wm_full = 6; }
        }
        step_bx.push_back(obx);
    };

    // front-end steps
    if (!dense_only) {
        for (auto& fm : fms)
            if (fm.normalize != fms[0].normalize || fm.pad_left != fms[0].pad_left || fm.pad_right != fms[0].pad_right) {
                *err = "front-end: branches disagree on normalisation / clip padding";
                return false;
            }
        if (fms[0].normalize) {
            Step s; s.kind = S_MINMAX; s.name = "clip_minmax"; s.kclass = "clip_minmax"; s.in0 = v_input; s.out = v_mm;
            s.bytes = (double)n_samples * 4;
            add_step(s);
        }
        int v_spec = new_val(spec_tensor, (size_t)fms[0].n_mels * fms[0].F * C_spec);
        int v_xn = -1;                      // normalised clip (FFT front-end only)
        struct FftFin { int v_bins, spec; size_t o_mel, o_span; bool banded; bool fused; };
        std::vector<FftFin> fft_fin;                 // FFT-path channels awaiting mel + pow + NHWC store
        tv[spec_tensor] = v_spec;
        for (size_t i = 0; i < fms.size(); i++) {
            const FrontendMatch& fm = fms[i];
            FrontSpec fs;
            fs.L = fm.L; fs.Lfft = fm.Lfft; fs.hop = fm.hop; fs.F = fm.F; fs.n_mels = fm.n_mels; fs.c = chan_of[i];
            fs.NTP = (int)align_up(fm.n_mels, 16);
            fs.Kp = (int)align_up(fm.Lfft / 2 + 1, frontend_kc(fm.Lfft, fm.hop, fs.NTP));
            if (fs.NTP > 128) { *err = "front-end: more than 128 mel bins unsupported"; return false; }
            if (fm.eps != fms[0].eps || fm.norm_sub != fms[0].norm_sub || fm.norm_mul != fms[0].norm_mul) {
                *err = "front-end: branches use different normalisation constants";
                return false;
            }
            fs.p1 = fm.p1; fs.p2 = fm.p2; fs.eps = fm.eps; fs.norm_sub = fm.norm_sub; fs.norm_mul = fm.norm_mul;
            fs.normalize = fm.normalize; fs.log_compress = fm.log_compress; fs.time_major = fm.time_major;
            fs.pad_left = fm.pad_left; fs.log_floor = fm.log_floor; fs.log_scale = fm.log_scale;
            if ((long)(fm.F - 1) * fm.hop + fm.L > (long)n_samples + fm.pad_left + fm.pad_right) {
                *code = BNHIP_E_MODEL;
                *err = "front-end: the framing selector reaches beyond the (padded) clip";
                return false;
            }
            if (fm.time_major != fms[0].time_major) { *err = "front-end: branches disagree on the image layout"; return false; }
            // bins the mel matrix actually uses (DFT truncation)
            std::vector<int> bins;
            {
                const int nbins = fm.Lfft / 2 + 1;
                const float* melT = P.T(fm.mel_tensor).f32();
                for (int k = 0; k < nbins; k++) {
                    bool nz = false;
                    for (int mm = 0; mm < fm.n_mels && !nz; mm++) nz = melT[(size_t)mm * nbins + k] != 0.0f;
                    if (nz) bins.push_back(k);
                }
            }
            const bool can_fft = stft_supported(fm.Lfft, (int)bins.size()) && !bins.empty();
            if (fm.magnitude && !can_fft) {
                *code = BNHIP_E_UNSUPPORTED;
                *err = "front-end: magnitude STFT (COMPLEX_ABS) is only implemented for fft_length 512 / 1024 / 2048";
                return false;
            }
            // the folded-GEMM kernel only knows the v2.4 layer (normalised clip, power compression, [mel, time] image)
            const bool variant = !fm.normalize || fm.log_compress || fm.time_major || fm.pad_left || fm.pad_right;
            if (variant && !can_fft) {
                *code = BNHIP_E_UNSUPPORTED;
                *err = "front-end: log-mel / unnormalised / padded front-ends are only implemented for fft_length 512 / 1024 / 2048";
                return false;
            }
            fs.fft = fm.magnitude || variant || (frontend_fft != 0 && can_fft);     // measured faster than the folded GEMM (0.95 vs 1.05 ms)
            if (fs.fft) {
                // normalise (once) -> STFT bins -> mel GEMM -> pow + NHWC store
                if (!fm.normalize) v_xn = v_input;
                if (v_xn < 0) {
                    v_xn = new_val(-1, n_samples);
                    Step nz; nz.kind = S_NORMALIZE; nz.name = "normalize"; nz.kclass = "frontend"; nz.in0 = v_input; nz.in1 = v_mm;
                    nz.out = v_xn; nz.bytes = (double)n_samples * 8;
                    add_step(nz);
                }
                fs.nb = (int)bins.size(); fs.nbp = (int)align_up(fs.nb, 4); fs.mode = fm.magnitude ? 1 : 0;
                std::vector<float> wfull(fm.window.begin(), fm.window.end());
                std::vector<float> binsf(bins.size());
                memcpy(binsf.data(), bins.data(), bins.size() * sizeof(int));            // int32 image in the float arena
                const int nbins = fm.Lfft / 2 + 1;
                const float* melT = P.T(fm.mel_tensor).f32();
                std::vector<float> melw((size_t)fm.n_mels * fs.nbp, 0.f);                // [n_mels][nbp], rows in output order
                for (int mo = 0; mo < fm.n_mels; mo++) {
                    int mm = fm.reverse ? fm.n_mels - 1 - mo : mo;
                    for (int r = 0; r < fs.nb; r++) melw[(size_t)mo * fs.nbp + r] = melT[(size_t)mm * nbins + bins[r]];
                }
                size_t o_win = wpush(wfull.data(), wfull.size()), o_bins = wpush(binsf.data(), binsf.size());
                size_t o_mel = wpush(melw.data(), melw.size());
                std::vector<double> twt = stft_build_tables(fm.Lfft, bins.data(), fs.nb);
                fs.stft_zmask = stft_zmask(fm.Lfft, bins.data(), fs.nb);
                size_t o_tw = wpush(reinterpret_cast<const float*>(twt.data()), twt.size() * 2);   // fp64 image, 256-B aligned
                specs.push_back(fs);
                const int si = (int)specs.size() - 1;
                // band structure of the mel rows: [lo, hi) of the nonzero columns.  A filterbank (2 nonzeros per bin) takes the
                // banded kernel; anything denser than a quarter of the matrix stays a GEMM.
                std::vector<int> span((size_t)2 * fm.n_mels, 0);
                long span_sum = 0;
                for (int mo = 0; mo < fm.n_mels; mo++) {
                    int lo = fs.nb, hi = 0;
                    for (int r = 0; r < fs.nb; r++)
                        if (melw[(size_t)mo * fs.nbp + r] != 0.0f) { lo = std::min(lo, r); hi = r + 1; }
                    if (hi == 0) lo = 0;
                    span[2 * mo] = lo; span[2 * mo + 1] = hi;
                    span_sum += hi - lo;
                }
                std::vector<float> spanf(span.size());
                memcpy(spanf.data(), span.data(), span.size() * sizeof(int));            // int32 image in the float arena
                size_t o_span = wpush(spanf.data(), spanf.size());
                const bool banded = !getenv("BNHIP_NO_MEL_BANDED") && span_sum * 4 <= (long)fm.n_mels * fs.nb;
                // a banded mel matrix is applied by the wave that transformed the frame (k_stft_bins<.., MEL>): no bins tensor,
                // no separate mel kernel; the image value is written by the STFT step itself
                int mel_quads = 0;
                std::vector<float> meltab;
                // OPT-IN (BNHIP_FUSE_MEL=1): measured slower than the separate banded kernel in every form tried - the STFT kernels
                // are fp64-VALU / LDS-latency bound at two to three waves per SIMD, and the epilogue's extra LDS round trips land on
                // their critical path: v2.4 batch 256, stft0 440 -> 512-531 us, stft1 209 -> 293-320 us against the 116 us
                // k_mel_banded launch they replace (70.2 k -> 67.6-68.4 k clips/s).  It does remove 0.9 MB per clip of HBM traffic
                // and a launch, and stays parity-tested.
                if (banded && getenv("BNHIP_FUSE_MEL")) meltab = stft_mel_table(fm.Lfft, melw.data(), span.data(), fm.n_mels, fs.nb, fs.nbp, &mel_quads);
                const bool fused = !meltab.empty();
                int v_bins = fused ? -1 : new_val(-1, (size_t)fm.F * fs.nbp);
                Step st; st.kind = S_STFT; st.name = "stft" + std::to_string(i); st.kclass = "stft"; st.in0 = v_xn; st.out = fused ? v_spec : v_bins;
                st.spec = si;
                st.flops = (double)fm.F * 2.5 * fm.Lfft * std::log2((double)fm.Lfft / 2);      // ~5 N/2 log2(N/2) per frame
                st.bytes = (double)n_samples * 4 + (double)fm.F * (fused ? fm.n_mels : fs.nbp) * 4;
                if (fused) { st.mode = 1; st.S = mel_quads; st.name += "+mel"; st.flops += 4.0 * fm.F * fs.nb; }
                add_step(st, o_win, o_bins, o_tw, fused ? wpush(meltab.data(), meltab.size()) : SIZE_MAX);
                fft_fin.push_back({v_bins, si, o_mel, o_span, banded, fused});
                continue;
            }
            if (frontend_lds_bytes(fs.Lfft, fs.Kp, fs.hop, fs.NTP) > 160 * 1024) {
                *err = "front-end: frame tile does not fit in LDS";
                return false;
            }
            std::vector<double> G = build_G(P, fm, fs.Kp, fs.NTP);
            size_t goff = wpush(reinterpret_cast<const float*>(G.data()), G.size() * 2);   // fp64 image, 256-B aligned
            // w[n'] for n' = 0..Lfft/2, then the mirror weights w[Lfft-n'] (0 when n' = 0, n' = Lfft/2, or beyond the frame)
            std::vector<float> wpad((size_t)2 * fs.Kp, 0.0f);
            for (int n = 0; n <= fm.Lfft / 2; n++) {
                if (n < fm.L) wpad[n] = fm.window[n];
                int mi = fm.Lfft - n;
                if (n > 0 && mi != n && mi < fm.L) wpad[fs.Kp + n] = fm.window[mi];
            }
            size_t woff = wpush(wpad.data(), wpad.size());
            specs.push_back(fs);
            Step f; f.kind = S_FRONTEND; f.name = "melspec" + std::to_string(i); f.kclass = "frontend";
            f.in0 = v_input; f.in1 = v_mm; f.out = v_spec; f.spec = (int)specs.size() - 1;
            f.flops = 2.0 * fm.F * (fm.Lfft / 2 + 1) * fm.n_mels;
            f.bytes = (double)n_samples * 4 + (double)fm.F * fm.n_mels * 4;
            add_step(f, goff, woff);
        }
        // mel + pow + NHWC store of the FFT-path channels: two adjacent channels go out as one float2 per pixel.  Banded mel
        // matrices take one fused kernel; otherwise the mel projection is a k_pw_gemm per channel followed by k_mel_finish.
        for (size_t k = 0; k < fft_fin.size();) {
            if (fft_fin[k].fused) { k++; continue; }        // mel + compression + store happened in the STFT kernel
            const FrontSpec& a = specs[fft_fin[k].spec];
            bool pair = k + 1 < fft_fin.size() && !fft_fin[k + 1].fused && specs[fft_fin[k + 1].spec].c == a.c + 1 && (a.c & 1) == 0 &&
                        specs[fft_fin[k + 1].spec].F == a.F && specs[fft_fin[k + 1].spec].n_mels == a.n_mels &&
                        specs[fft_fin[k + 1].spec].log_compress == a.log_compress && specs[fft_fin[k + 1].spec].log_floor == a.log_floor &&
                        specs[fft_fin[k + 1].spec].log_scale == a.log_scale;
            const int nch = pair ? 2 : 1;
            bool banded = fft_fin[k].banded && (!pair || fft_fin[k + 1].banded) &&
                          mel_banded_supported(a.n_mels, a.nbp, pair ? specs[fft_fin[k + 1].spec].nbp : 0);
            if (banded) {
                Step mb; mb.kind = S_MELBAND; mb.name = "melband" + std::to_string(a.c); mb.kclass = "frontend";
                mb.in0 = fft_fin[k].v_bins; mb.spec = fft_fin[k].spec; mb.out = v_spec; mb.S = nch;
                double bytes = 4.0 * a.F * (a.nbp + a.n_mels), flops = 0;
                if (pair) {
                    mb.in1 = fft_fin[k + 1].v_bins; mb.op = fft_fin[k + 1].spec; mb.name += "+" + std::to_string(a.c + 1);
                    bytes += 4.0 * a.F * (specs[mb.op].nbp + a.n_mels);
                }
                for (int c = 0; c < nch; c++) flops += 4.0 * a.F * specs[fft_fin[k + c].spec].nb;     // ~2 nonzeros per bin
                mb.bytes = bytes; mb.flops = flops;
                add_step(mb, fft_fin[k].o_mel, fft_fin[k].o_span, pair ? fft_fin[k + 1].o_mel : SIZE_MAX, pair ? fft_fin[k + 1].o_span : SIZE_MAX);
                k += nch;
                continue;
            }
            int v_T[2] = {-1, -1};
            for (int c = 0; c < nch; c++) {
                const FrontSpec& fc = specs[fft_fin[k + c].spec];
                v_T[c] = new_val(-1, (size_t)fc.F * fc.n_mels);
                Step g; g.kind = S_PW; g.name = "mel" + std::to_string(fc.c); g.kclass = "pw_gemm"; g.in0 = fft_fin[k + c].v_bins; g.out = v_T[c];
                g.H = fc.F; g.W = 1; g.C = fc.nbp; g.Co = fc.n_mels; g.Ho = fc.F; g.Wo = 1; g.act = ACT_NONE;
                g.flops = 2.0 * fc.F * fc.nbp * fc.n_mels;
                g.bytes = 4.0 * ((double)fc.F * fc.nbp + (double)fc.F * fc.n_mels);
                g.wbytes = 4.0 * fc.nbp * fc.n_mels;
                add_step(g, fft_fin[k + c].o_mel);
            }
            Step mf; mf.kind = S_MELFIN; mf.name = "melspec" + std::to_string(a.c); mf.kclass = "frontend";
            mf.in0 = v_T[0]; mf.spec = fft_fin[k].spec; mf.out = v_spec;
            mf.S = nch;
            if (pair) { mf.in1 = v_T[1]; mf.op = fft_fin[k + 1].spec; mf.name += "+" + std::to_string(a.c + 1); }
            mf.bytes = 8.0 * a.F * a.n_mels * mf.S;
            add_step(mf);
            k += mf.S;
        }
    }

    // ---------------------------------------------------------------- CNN ops
    auto hwc = [&](int t, int* H, int* W, int* C) -> bool {
        const auto& s = m.tensors[t].shape;
        if (s.size() == 4 && s[0] == 1) { *H = s[1]; *W = s[2]; *C = s[3]; return true; }
        if (s.size() == 2 && s[0] == 1) { *H = 1; *W = 1; *C = s[1]; return true; }
        return false;
    };
    // fused activation of a conv / dense op -> activation the kernels apply; the two the fused kernels do not implement
    // (RELU_N1_TO_1, TANH) become a separate elementwise step (*post)
    auto map_act = [&](int fused, int* post = nullptr) -> int {
        if (post) *post = 0;
        switch (fused) {
            case 0: return ACT_NONE; case 1: return ACT_RELU; case 3: return ACT_RELU6;
            case 2: if (post) { *post = U_RELU_N1_TO_1; return ACT_NONE; } return -1;
            case 4: if (post) { *post = U_TANH; return ACT_NONE; } return -1;
            default: return -1;
        }
    };
    // top / left zero padding of a convolution-like op (SAME: TF's rule; VALID: none; explicit: folded-in PAD)
    auto conv_pads = [&](const TflOp& o, int H, int W, int Ho, int Wo, int kh, int kw, int* pt, int* pl) {
        *pt = 0; *pl = 0;
        if (o.explicit_pad) { *pt = o.pad_t; *pl = o.pad_l; return; }
        if (o.padding == 0) {
            *pt = std::max((Ho - 1) * o.stride_h + (kh - 1) * o.dil_h + 1 - H, 0) / 2;
            *pl = std::max((Wo - 1) * o.stride_w + (kw - 1) * o.dil_w + 1 - W, 0) / 2;
        }
    };
    // trailing swish / sigmoid detection on tensor y produced by op `oi`; returns final tensor and act
    auto trailing_act = [&](int y, int* act) -> int {
        if (P.consumers[y].size() == 2 && P.uses[y] == 2) {
            int a = P.consumers[y][0], b = P.consumers[y][1];
            int lg = m.ops[a].code == OP_LOGISTIC ? a : (m.ops[b].code == OP_LOGISTIC ? b : -1);
            int mu = lg == a ? b : a;
            if (lg >= 0 && m.ops[mu].code == OP_MUL && m.ops[mu].act == 0) {
                int sgt = m.ops[lg].outputs[0];
                const auto& mi = m.ops[mu].inputs;
                bool ok = P.uses[sgt] == 1 && ((mi[0] == y && mi[1] == sgt) || (mi[1] == y && mi[0] == sgt));
                if (ok) { P.absorbed[lg] = 1; P.absorbed[mu] = 1; *act = ACT_SWISH; return m.ops[mu].outputs[0]; }
            }
        } else if (P.uses[y] == 1 && P.consumers[y].size() == 1 && !P.absorbed[P.consumers[y][0]]) {
            // a stand-alone activation op right behind the producer (ONNX lowerings, unfused exports)
            const int ci = P.consumers[y][0];
            const int cc = m.ops[ci].code;
            const int a = cc == OP_LOGISTIC ? ACT_SIGMOID : cc == OP_RELU ? ACT_RELU : cc == OP_RELU6 ? ACT_RELU6 :
                          cc == OP_HARD_SWISH ? ACT_HARD_SWISH : -1;
            if (a >= 0) { P.absorbed[ci] = 1; *act = a; return m.ops[ci].outputs[0]; }
        }
        return y;
    };
    const bool fuse_expdw = !(getenv("BNHIP_NO_FUSE_EXPDW") && atoi(getenv("BNHIP_NO_FUSE_EXPDW")) != 0);
    struct ScaledAlias { int v_data; int v_scale; };
    std::map<int, ScaledAlias> scaled;   // tflite tensor (SE MUL output) -> (data value, scale value)

    for (int oi = 0; oi < (int)m.ops.size(); oi++) {
        if (P.absorbed[oi]) continue;
        const TflOp& o = m.ops[oi];
        const std::string oname = m.tensors[o.outputs[0]].name;
        auto need_val = [&](int t) -> int {
            auto it = tv.find(t);
            return it == tv.end() ? -1 : it->second;
        };
        // binds the step's output to TFLite tensor `outt` and appends it; a fused activation the kernels do not implement
        // (post != 0) runs as a separate elementwise step on an internal value
        auto emit = [&](Step st, int outt, size_t elems, int post, size_t o0 = SIZE_MAX, size_t o1 = SIZE_MAX, size_t o2 = SIZE_MAX,
                        size_t o3 = SIZE_MAX) {
            if (!post) { st.out = new_val(outt, elems); tv[outt] = st.out; add_step(st, o0, o1, o2, o3); return; }
            st.out = new_val(-1, elems);
            add_step(st, o0, o1, o2, o3);
            Step u; u.kind = S_EW_UNARY; u.kclass = "elementwise"; u.name = st.name + "/act"; u.in0 = st.out; u.op = post;
            u.out = new_val(outt, elems); tv[outt] = u.out; u.bytes = 8.0 * elems;
            add_step(u);
        };
        // an operand of a generic op: activation value, or constant uploaded to the weight arena
        struct Operand { int val = -1; size_t woff = SIZE_MAX; bool ok = false; };
        auto operand = [&](int t) -> Operand {
            Operand r;
            if (t < 0) return r;
            r.val = need_val(t);
            if (r.val >= 0) { r.ok = true; return r; }
            if (P.is_const(t) && m.tensors[t].type == TT_FLOAT32) { r.woff = wconst(t); r.ok = true; }
            return r;
        };
        // shape -> rank-5 (batch + 4) by left-padding with 1s; false when the rank is larger or the leading dim is not 1
        auto shape5 = [&](const std::vector<int>& sh, int out[5]) -> bool {
            std::vector<int> v = sh;
            while (v.size() > 5 && v[0] == 1) v.erase(v.begin());
            if (v.size() > 5) return false;
            for (int k = 0; k < 5; k++) out[k] = 1;
            for (size_t k = 0; k < v.size(); k++) out[5 - v.size() + k] = v[k];
            return out[0] == 1;
        };
        auto dense_strides = [&](const int d5[5], long st[5]) { st[4] = 1; for (int k = 3; k >= 0; k--) st[k] = st[k + 1] * d5[k + 1]; };
        switch (o.code) {
            case OP_CONV_2D: {
                int in_t = o.inputs[0];
                if (!P.is_const(o.inputs[1])) { *err = "CONV_2D with non-constant filter"; return false; }
                const TflTensor& w = m.tensors[o.inputs[1]];
                int H, W, C, Ho, Wo, Co;
                if (!hwc(in_t, &H, &W, &C) || !hwc(o.outputs[0], &Ho, &Wo, &Co) || w.shape.size() != 4 || w.shape[3] != C) {
                    *err = "CONV_2D: unsupported shapes at " + oname; return false;
                }
                int kh = w.shape[1], kw = w.shape[2];
                int post = 0;
                int act = map_act(o.act, &post);
                if (act < 0) { *err = "CONV_2D: unsupported fused activation"; return false; }
                int outt = o.outputs[0];
                if (act == ACT_NONE && !post) outt = trailing_act(outt, &act);
                size_t boff = (o.inputs.size() > 2 && o.inputs[2] >= 0) ? wconst(o.inputs[2]) : SIZE_MAX;
                bool pw = kh == 1 && kw == 1 && o.stride_h == 1 && o.stride_w == 1 && !o.explicit_pad && Ho == H && Wo == W;
                Step s; s.name = oname; s.H = H; s.W = W; s.C = C; s.Ho = Ho; s.Wo = Wo; s.Co = Co; s.act = act;
                // ---- MBConv front half: 1x1 expand whose only consumer is a depthwise conv -> one fused kernel
                if (pw && !post && fuse_expdw && scaled.find(in_t) == scaled.end() && P.uses[outt] == 1 && P.consumers[outt].size() == 1) {
                    int di = P.consumers[outt][0];
                    const TflOp& d = m.ops[di];
                    int dH, dW, dC, dHo, dWo, dCo;
                    if (d.code == OP_DEPTHWISE_CONV_2D && !P.absorbed[di] && d.inputs[0] == outt && P.is_const(d.inputs[1]) &&
                        hwc(outt, &dH, &dW, &dC) && hwc(d.outputs[0], &dHo, &dWo, &dCo) && dCo == dC && d.depth_multiplier == 1 &&
                        d.dil_h == 1 && d.dil_w == 1 && d.stride_h == d.stride_w) {
                        const TflTensor& wd = m.tensors[d.inputs[1]];
                        int kd = wd.shape.size() == 4 ? wd.shape[1] : 0;
                        int act_d = map_act(d.act);
                        int fpt = 0, fpl = 0;
                        conv_pads(d, dH, dW, dHo, dWo, kd, kd, &fpt, &fpl);
                        if (kd == wd.shape[2] && wd.shape[3] == dC && act_d >= 0 && expdw_supported(kd, d.stride_h, C, Co, act, bf16x3 ? precision : 0) &&
                            expdw_sum_slabs(ExpDwGeo{kd, d.stride_h, dH, dW, dHo, dWo, fpt, fpl}) > 0 && need_val(in_t) >= 0) {
                            int dout = d.outputs[0];
                            if (act_d == ACT_NONE) dout = trailing_act(dout, &act_d);
                            P.absorbed[di] = 1;
                            Step f; f.kind = S_EXPAND_DW; f.kclass = "expand_dw"; f.name = oname + "+dw";
                            f.in0 = need_val(in_t);
                            f.H = H; f.W = W; f.C = C; f.Co = Co; f.Ho = dHo; f.Wo = dWo; f.kh = kd; f.kw = kd;
                            f.sh = d.stride_h; f.sw = d.stride_w; f.act = act; f.act2 = act_d;
                            f.pt = fpt; f.pl = fpl;
                            f.flops = 2.0 * H * W * C * Co + 2.0 * dHo * dWo * Co * kd * kd;
                            f.bytes = 4.0 * ((double)H * W * C + (double)dHo * dWo * Co);
                            f.wbytes = 4.0 * (C * Co + kd * kd * Co);
                            f.out = new_val(dout, (size_t)dHo * dWo * Co);
                            tv[dout] = f.out;
                            // padded parameter copies: every load in k_expand_dw is unconditional (see kernels.hip)
                            const int Kw = expdw_kw(C), Cp = expdw_cp(Co);
                            std::vector<float> wep((size_t)Cp * Kw, 0.f), bep(Cp, 0.f), wdp((size_t)kd * kd * Cp, 0.f), bdp(Cp, 0.f);
                            const float* wsrc = w.f32();
                            for (int n = 0; n < Co; n++) memcpy(&wep[(size_t)n * Kw], wsrc + (size_t)n * C, (size_t)C * sizeof(float));
                            if (o.inputs.size() > 2 && o.inputs[2] >= 0) memcpy(bep.data(), m.tensors[o.inputs[2]].f32(), (size_t)Co * sizeof(float));
                            const float* dsrc = wd.f32();
                            for (int t = 0; t < kd * kd; t++) memcpy(&wdp[(size_t)t * Cp], dsrc + (size_t)t * Co, (size_t)Co * sizeof(float));
                            if (d.inputs.size() > 2 && d.inputs[2] >= 0) memcpy(bdp.data(), m.tensors[d.inputs[2]].f32(), (size_t)Co * sizeof(float));
                            size_t o_we = wpush(wep.data(), wep.size()), o_be = wpush(bep.data(), bep.size());
                            size_t o_wd = wpush(wdp.data(), wdp.size()), o_bd = wpush(bdp.data(), bdp.size());
                            add_step(f, o_we, o_be, o_wd, o_bd);
                            if (bf16x3 && expdw_bx_ok(C)) {          // split-bf16 image of the expand weights (autotuned per layer)
                                std::vector<uint16_t> img = expdw_bx_image(wsrc, Co, C);
                                std::vector<float> asf((img.size() + 1) / 2);
                                memcpy(asf.data(), img.data(), img.size() * 2);
                                step_bx.back() = wpush(asf.data(), asf.size());
                                // (arithmetic by rule, not by timing: the split-bf16 phase 1 measured +0-0.5 % at best, so it is
                                // used only where asked for)
                                if (bf16x3 >= 2 || precision == 1) steps.back().bx = 1;
                            }
                            break;
                        }
                    }
                }
                if (pw) {
                    s.kind = S_PW; s.kclass = "pw_gemm";
                    auto sc = scaled.find(in_t);
                    if (sc != scaled.end()) { s.in0 = sc->second.v_data; s.in1 = sc->second.v_scale; }
                    else s.in0 = need_val(in_t);
                    if (s.in0 < 0) { *err = "CONV_2D: input has no value: " + oname; return false; }
                    // residual ADD fusion
                    if (!post && P.uses[outt] == 1 && P.consumers[outt].size() == 1) {
                        int ai = P.consumers[outt][0];
                        const TflOp& ad = m.ops[ai];
                        if (ad.code == OP_ADD && ad.act == 0 && !P.absorbed[ai]) {
                            int other = ad.inputs[0] == outt ? ad.inputs[1] : ad.inputs[0];
                            int vo = need_val(other);
                            if (vo >= 0 && m.tensors[other].shape == m.tensors[outt].shape) {
                                s.in2 = vo; P.absorbed[ai] = 1; outt = ad.outputs[0];
                            }
                        }
                    }
                    s.flops = 2.0 * H * W * C * Co;
                    s.bytes = 4.0 * ((double)H * W * C + (double)H * W * Co * (s.in2 >= 0 ? 2 : 1));
                    s.wbytes = 4.0 * C * Co;
                    emit(s, outt, (size_t)Ho * Wo * Co, post, wconst(o.inputs[1]), boff);
                } else if (conv_igemm_supported(C, Co, kh, kw) && scaled.find(in_t) == scaled.end() && !getenv("BNHIP_NO_CONV_IGEMM") &&
                           (double)max_batch * Ho * Wo * std::max(Ho * Wo, C) < 1.0e12) {
                    // a real convolution (kh x kw over >= 4 input channels): implicit GEMM on the f32 MFMA, weights as in the file
                    s.kind = S_CONV_IGEMM; s.kclass = "conv_igemm";
                    s.in0 = need_val(in_t);
                    if (s.in0 < 0) { *err = "CONV_2D: input has no value: " + oname; return false; }
                    s.kh = kh; s.kw = kw; s.sh = o.stride_h; s.sw = o.stride_w; s.g.dh = o.dil_h; s.g.dw = o.dil_w;
                    conv_pads(o, H, W, Ho, Wo, kh, kw, &s.pt, &s.pl);
                    s.flops = 2.0 * Ho * Wo * Co * kh * kw * C;
                    s.bytes = 4.0 * ((double)H * W * C + (double)Ho * Wo * Co);
                    s.wbytes = 4.0 * kh * kw * C * Co;
                    emit(s, outt, (size_t)Ho * Wo * Co, post, wconst(o.inputs[1]), boff);
                } else if (o.dil_h != 1 || o.dil_w != 1 || (Co & 3)) {
                    // dilated convolutions and channel counts the vectorised kernels do not cover: generic kernel, weights as
                    // in the file (OHWI)
                    s.kind = S_CONV_GENERIC; s.kclass = "conv_generic";
                    auto sc = scaled.find(in_t);
                    if (sc != scaled.end()) { *err = "CONV_2D: squeeze-excite scale in front of an unsupported convolution at " + oname; return false; }
                    s.in0 = need_val(in_t);
                    if (s.in0 < 0) { *err = "CONV_2D: input has no value: " + oname; return false; }
                    s.kh = kh; s.kw = kw; s.sh = o.stride_h; s.sw = o.stride_w; s.g.dh = o.dil_h; s.g.dw = o.dil_w;
                    conv_pads(o, H, W, Ho, Wo, kh, kw, &s.pt, &s.pl);
                    s.flops = 2.0 * Ho * Wo * Co * kh * kw * C;
                    s.bytes = 4.0 * ((double)H * W * C + (double)Ho * Wo * Co);
                    emit(s, outt, (size_t)Ho * Wo * Co, post, wconst(o.inputs[1]), boff);
                } else {
                    s.kind = S_CONV_DIRECT; s.kclass = "conv_direct";
                    if (scaled.find(in_t) != scaled.end()) { *err = "CONV_2D: squeeze-excite scale in front of an unsupported convolution at " + oname; return false; }
                    s.in0 = need_val(in_t);
                    if (s.in0 < 0) { *err = "CONV_2D: input has no value: " + oname; return false; }
                    s.kh = kh; s.kw = kw; s.sh = o.stride_h; s.sw = o.stride_w;
                    conv_pads(o, H, W, Ho, Wo, kh, kw, &s.pt, &s.pl);
                    // re-lay OHWI -> [kh][kw][Cin][Cout]
                    std::vector<float> wt((size_t)kh * kw * C * Co);
                    const float* ws = w.f32();
                    for (int oc = 0; oc < Co; oc++)
                        for (int i = 0; i < kh; i++)
                            for (int j = 0; j < kw; j++)
                                for (int ic = 0; ic < C; ic++)
                                    wt[(((size_t)i * kw + j) * C + ic) * Co + oc] = ws[(((size_t)oc * kh + i) * kw + j) * C + ic];
                    s.flops = 2.0 * Ho * Wo * Co * kh * kw * C;
                    s.bytes = 4.0 * ((double)H * W * C + (double)Ho * Wo * Co);
                    {
                        // second image for the MFMA stem (used when stem_mfma_supported): [Cout][32], kk = i*8 + j*2 + ic
                        ConvParams cp{nullptr, nullptr, nullptr, nullptr, 1, H, W, C, Ho, Wo, Co, kh, kw, s.sh, s.sw, s.pt, s.pl, act};
                        if (stem_mfma_supported(cp) && !getenv("BNHIP_NO_STEM_MFMA")) {
                            std::vector<float> wm((size_t)Co * 32, 0.f), bp(Co, 0.f);
                            for (int oc = 0; oc < Co; oc++)
                                for (int i = 0; i < 3; i++)
                                    for (int j = 0; j < 3; j++)
                                        for (int ic = 0; ic < 2; ic++)
                                            wm[(size_t)oc * 32 + i * 8 + j * 2 + ic] = ws[(((size_t)oc * kh + i) * kw + j) * C + ic];
                            if (o.inputs.size() > 2 && o.inputs[2] >= 0) memcpy(bp.data(), m.tensors[o.inputs[2]].f32(), (size_t)Co * sizeof(float));
                            // ---- stem whose only consumer is a 3x3 stride-1 depthwise conv: one k_expand_dw<STEM> launch; the
                            // stem output (the largest tensor of the network) never reaches HBM
                            if (!post && fuse_expdw && !getenv("BNHIP_NO_FUSE_STEM") && P.uses[outt] == 1 && P.consumers[outt].size() == 1 && (Co % 32) == 0) {
                                int di = P.consumers[outt][0];
                                const TflOp& d = m.ops[di];
                                int dH, dW, dC, dHo, dWo, dCo;
                                if (d.code == OP_DEPTHWISE_CONV_2D && !P.absorbed[di] && d.inputs[0] == outt && P.is_const(d.inputs[1]) &&
                                    hwc(outt, &dH, &dW, &dC) && hwc(d.outputs[0], &dHo, &dWo, &dCo) && dCo == dC && d.depth_multiplier == 1 &&
                                    d.dil_h == 1 && d.dil_w == 1 && d.stride_h == 1 && d.stride_w == 1) {
                                    const TflTensor& wd = m.tensors[d.inputs[1]];
                                    int kd = wd.shape.size() == 4 ? wd.shape[1] : 0;
                                    int act_d = map_act(d.act);
                                    int fpt = 0, fpl = 0;
                                    conv_pads(d, dH, dW, dHo, dWo, kd, kd, &fpt, &fpl);
                                    if (kd == 3 && wd.shape[2] == 3 && wd.shape[3] == dC && act_d >= 0 && s.in0 >= 0 &&
                                        expdw_sum_slabs(ExpDwGeo{kd, 1, dH, dW, dHo, dWo, fpt, fpl, true}) > 0) {
                                        int dout = d.outputs[0];
                                        if (act_d == ACT_NONE) dout = trailing_act(dout, &act_d);
                                        P.absorbed[di] = 1;
                                        Step f; f.kind = S_EXPAND_DW; f.kclass = "expand_dw"; f.name = oname + "+dw"; f.mode = 1;
                                        f.in0 = s.in0;
                                        f.H = Ho; f.W = Wo; f.C = 32; f.Co = Co; f.Ho = dHo; f.Wo = dWo; f.kh = kd; f.kw = kd; f.sh = 1; f.sw = 1;
                                        f.pt = fpt; f.pl = fpl; f.act = act; f.act2 = act_d;
                                        f.H2 = H; f.W2 = W; f.pt2 = s.pt; f.pl2 = s.pl;
                                        f.flops = 2.0 * Ho * Wo * Co * kh * kw * C + 2.0 * dHo * dWo * Co * kd * kd;
                                        f.bytes = 4.0 * ((double)H * W * C + (double)dHo * dWo * Co);
                                        f.wbytes = 4.0 * (Co * 32 + kd * kd * Co);
                                        f.out = new_val(dout, (size_t)dHo * dWo * Co);
                                        tv[dout] = f.out;
                                        const int Cp = expdw_cp(Co);
                                        // expand-side weights: the first 24 of the 32 columns of the MFMA stem image (row 2 is the
                                        // 8-wide half slab of the fused kernel; columns 24..31 are the zero padding of k_stem_mfma)
                                        std::vector<float> wep((size_t)Cp * 24, 0.f), bep(Cp, 0.f), wdp((size_t)kd * kd * Cp, 0.f), bdp(Cp, 0.f);
                                        for (int n = 0; n < Co; n++) memcpy(&wep[(size_t)n * 24], &wm[(size_t)n * 32], 24 * sizeof(float));
                                        memcpy(bep.data(), bp.data(), bp.size() * sizeof(float));
                                        const float* dsrc = wd.f32();
                                        for (int t = 0; t < kd * kd; t++) memcpy(&wdp[(size_t)t * Cp], dsrc + (size_t)t * Co, (size_t)Co * sizeof(float));
                                        if (d.inputs.size() > 2 && d.inputs[2] >= 0) memcpy(bdp.data(), m.tensors[d.inputs[2]].f32(), (size_t)Co * sizeof(float));
                                        size_t o_we = wpush(wep.data(), wep.size()), o_be = wpush(bep.data(), bep.size());
                                        size_t o_wd = wpush(wdp.data(), wdp.size()), o_bd = wpush(bdp.data(), bdp.size());
                                        add_step(f, o_we, o_be, o_wd, o_bd);
                                        break;
                                    }
                                }
                            }
                            emit(s, outt, (size_t)Ho * Wo * Co, post, wpush(wt.data(), wt.size()), boff, wpush(wm.data(), wm.size()), wpush(bp.data(), bp.size()));
                            break;
                        }
                    }
                    emit(s, outt, (size_t)Ho * Wo * Co, post, wpush(wt.data(), wt.size()), boff);
                }
                break;
            }
            case OP_DEPTHWISE_CONV_2D: {
                int in_t = o.inputs[0];
                const TflTensor& w = m.tensors[o.inputs[1]];
                int H, W, C, Ho, Wo, Co;
                if (!P.is_const(o.inputs[1]) || !hwc(in_t, &H, &W, &C) || !hwc(o.outputs[0], &Ho, &Wo, &Co) || w.shape.size() != 4 ||
                    w.shape[3] != Co || o.depth_multiplier < 1 || Co != C * o.depth_multiplier) {
                    *err = "DEPTHWISE_CONV_2D: unsupported configuration at " + oname + " (filter must be constant [1,kh,kw,C*depth_multiplier])";
                    return false;
                }
                int post = 0;
                int act = map_act(o.act, &post);
                if (act < 0) { *err = "DEPTHWISE_CONV_2D: unsupported fused activation"; return false; }
                int outt = o.outputs[0];
                if (act == ACT_NONE && !post) outt = trailing_act(outt, &act);
                Step s; s.name = oname;
                s.in0 = need_val(in_t);
                if (s.in0 < 0) { *err = "DEPTHWISE_CONV_2D: input has no value"; return false; }
                s.H = H; s.W = W; s.C = C; s.Ho = Ho; s.Wo = Wo; s.Co = Co; s.act = act;
                s.kh = w.shape[1]; s.kw = w.shape[2]; s.sh = o.stride_h; s.sw = o.stride_w;
                conv_pads(o, H, W, Ho, Wo, s.kh, s.kw, &s.pt, &s.pl);
                s.flops = 2.0 * Ho * Wo * Co * s.kh * s.kw;
                s.bytes = 4.0 * ((double)H * W * C + (double)Ho * Wo * Co);
                size_t boff = (o.inputs.size() > 2 && o.inputs[2] >= 0) ? wconst(o.inputs[2]) : SIZE_MAX;
                if (o.depth_multiplier != 1 || o.dil_h != 1 || o.dil_w != 1) {
                    // channel multipliers and dilation: generic kernel (out channel oc reads input channel oc / multiplier)
                    s.kind = S_CONV_GENERIC; s.kclass = "conv_generic";
                    s.g.dh = o.dil_h; s.g.dw = o.dil_w; s.g.depthwise = 1; s.g.mult = o.depth_multiplier;
                } else {
                    s.kind = S_DW; s.kclass = "dwconv";
                }
                emit(s, outt, (size_t)Ho * Wo * Co, post, wconst(o.inputs[1]), boff);
                break;
            }
            case OP_MEAN: {
                int in_t = o.inputs[0];
                int H, W, C;
                const TflTensor& ax = m.tensors[o.inputs[1]];
                bool ok = hwc(in_t, &H, &W, &C) && ax.data && ax.numel() == 2 &&
                          ((ax.i32()[0] == 1 && ax.i32()[1] == 2) || (ax.i32()[0] == 2 && ax.i32()[1] == 1));
                if (!ok) goto generic_reduce;
                int vin = need_val(in_t);
                if (vin < 0) { *err = "MEAN: input has no value"; return false; }
                int S = mean_splits(H * W);
                bool fused_sum = false;
                Step mp; mp.kind = S_MEAN_PARTIAL; mp.kclass = "mean"; mp.name = oname + "/partial";
                mp.in0 = vin; mp.H = H; mp.W = W; mp.C = C;
                if (!steps.empty() && steps.back().kind == S_EXPAND_DW && steps.back().out == vin) {
                    const Step& d = steps.back();
                    const ExpDwGeo dg{d.kh, d.sh, d.H, d.W, d.Ho, d.Wo, d.pt, d.pl, d.mode == 1};
                    int slabs = expdw_sum_slabs(dg);
                    if (slabs > 0) {
                        S = slabs;
                        fused_sum = true;
                        // sized for the candidate shape with the most tiles: the autotuner may pick another one
                        mp.out = new_val(-1, (size_t)expdw_max_slabs(dg) * C);
                        steps.back().out2 = mp.out;
                        steps.back().S = S;
                    }
                } else if (!steps.empty() && steps.back().kind == S_DW && steps.back().out == vin) {
                    // the producer is a depthwise conv: let it emit the per-slab channel sums itself
                    const Step& d = steps.back();
                    DwParams dp{nullptr, nullptr, nullptr, nullptr, 1, d.H, d.W, d.C, d.Ho, d.Wo, d.kh, d.kw, d.sh, d.sw, d.pt, d.pl, d.act};
                    int slabs = dwconv_sum_slabs(dp);
                    if (slabs > 0) {
                        S = slabs;
                        fused_sum = true;
                        // sized for whichever kernel the autotuner ends up with (the LDS-staged form has its own tile counts)
                        int cap = S;
                        if (dwconv_lds_supported(dp)) cap = std::max(cap, expdw_max_slabs(ExpDwGeo{d.kh, d.sh, d.H, d.W, d.Ho, d.Wo, d.pt, d.pl}));
                        mp.out = new_val(-1, (size_t)cap * C);
                        steps.back().out2 = mp.out;
                        steps.back().S = S;
                    }
                }
                mp.S = S;
                if (!fused_sum) mp.out = new_val(-1, (size_t)S * C);
                mp.bytes = 4.0 * H * W * C;
                // squeeze-excite pattern?
                bool se = false;
                int mt = o.outputs[0];
                do {
                    // MEAN -> FC(Cr) + act -> FC(C) + act -> MUL with the MEAN's input, each FC a 1x1 CONV_2D or a FULLY_CONNECTED,
                    // with any pure reshapes in between (Keras: GlobalAveragePooling2D -> Reshape(1,1,C) -> Conv2D ... -> Multiply)
                    if (P.uses[mt] != 1) break;
                    std::vector<char> save = P.absorbed;
                    auto fc_like = [&](int oi, int K, int* N) -> bool {      // op oi: [*, K] -> [*, N] with constant weights
                        if (oi < 0 || P.absorbed[oi]) return false;
                        const TflOp& f = m.ops[oi];
                        if ((f.code != OP_CONV_2D && f.code != OP_FULLY_CONNECTED) || f.inputs.size() < 2 || !P.is_const(f.inputs[1])) return false;
                        const TflTensor& w = m.tensors[f.inputs[1]];
                        if (w.type != TT_FLOAT32) return false;
                        if (f.code == OP_CONV_2D) {
                            if (w.shape.size() != 4 || w.shape[1] != 1 || w.shape[2] != 1 || w.shape[3] != K || f.stride_h != 1 || f.stride_w != 1) return false;
                        } else if (w.shape.size() != 2 || w.shape[1] != K) return false;
                        if (f.inputs.size() > 2 && f.inputs[2] >= 0 && (!P.is_const(f.inputs[2]) || (int)m.tensors[f.inputs[2]].numel() != w.shape[0])) return false;
                        *N = w.shape[0];
                        return true;
                    };
                    auto fused_or_trailing = [&](int oi, int* act) -> int {  // activation of FC op oi; returns the tensor after it
                        *act = map_act(m.ops[oi].act);
                        if (*act < 0) return -1;
                        int y = m.ops[oi].outputs[0];
                        if (*act == ACT_NONE) y = trailing_act(y, act);
                        return y;
                    };
                    int c1 = P.only_consumer(P.skip_down(mt));
                    int Cr = 0, Cb = 0;
                    if (!fc_like(c1, C, &Cr)) { P.absorbed = save; break; }
                    int act1 = ACT_NONE, act2 = ACT_NONE;
                    int r = fused_or_trailing(c1, &act1);
                    if (r < 0) { P.absorbed = save; break; }
                    r = P.skip_down(r);
                    int c2 = P.uses[r] == 1 ? P.only_consumer(r) : -1;
                    if (!fc_like(c2, Cr, &Cb) || Cb != C) { P.absorbed = save; break; }
                    const TflTensor& w1 = m.tensors[m.ops[c1].inputs[1]];
                    const TflTensor& w2 = m.tensors[m.ops[c2].inputs[1]];
                    (void)w1;
                    int e = fused_or_trailing(c2, &act2);
                    if (e < 0) { P.absorbed = save; break; }
                    e = P.skip_down(e);
                    int mu = P.uses[e] == 1 ? P.only_consumer(e) : -1;
                    if (mu < 0 || m.ops[mu].code != OP_MUL || m.ops[mu].act != 0) { P.absorbed = save; break; }
                    int other = m.ops[mu].inputs[0] == e ? m.ops[mu].inputs[1] : m.ops[mu].inputs[0];
                    if (other != in_t) { P.absorbed = save; break; }
                    // matched
                    P.absorbed[c1] = 1; P.absorbed[c2] = 1; P.absorbed[mu] = 1;
                    if (!fused_sum) add_step(mp);
                    Step ss; ss.kind = S_SE; ss.kclass = "se"; ss.name = oname + "/se";
                    ss.in0 = mp.out; ss.H = H; ss.W = W; ss.C = C; ss.Cr = Cr; ss.S = S; ss.act = act1; ss.act2 = act2;
                    ss.out = new_val(e, (size_t)C);
                    tv[e] = ss.out;
                    ss.flops = 4.0 * C * Cr;
                    size_t b1 = (m.ops[c1].inputs.size() > 2 && m.ops[c1].inputs[2] >= 0) ? wconst(m.ops[c1].inputs[2]) : SIZE_MAX;
                    size_t b2 = (m.ops[c2].inputs.size() > 2 && m.ops[c2].inputs[2] >= 0) ? wconst(m.ops[c2].inputs[2]) : SIZE_MAX;
                    // second FC transposed [C][Cr] -> [Cr][C] so the kernel reads it coalesced
                    std::vector<float> w2t((size_t)C * Cr);
                    {
                        const float* w2s = w2.f32();
                        for (int cc = 0; cc < C; cc++)
                            for (int jj = 0; jj < Cr; jj++) w2t[(size_t)jj * C + cc] = w2s[(size_t)cc * Cr + jj];
                    }
                    add_step(ss, wconst(m.ops[c1].inputs[1]), b1, wpush(w2t.data(), w2t.size()), b2);
                    int u = m.ops[mu].outputs[0];
                    int pc = P.uses[u] == 1 ? P.only_consumer(u) : -1;
                    bool fold = false;
                    if (pc >= 0 && m.ops[pc].code == OP_CONV_2D && P.is_const(m.ops[pc].inputs[1]) && m.ops[pc].inputs[0] == u) {
                        const TflTensor& wp = m.tensors[m.ops[pc].inputs[1]];
                        fold = wp.shape.size() == 4 && wp.shape[1] == 1 && wp.shape[2] == 1 && m.ops[pc].stride_h == 1 &&
                               m.ops[pc].stride_w == 1 && (C % 4) == 0;
                    }
                    if (fold) {
                        scaled[u] = ScaledAlias{vin, ss.out};
                    } else {
                        Step bs; bs.kind = S_BINARY; bs.kclass = "elementwise"; bs.name = m.tensors[u].name;
                        bs.in0 = vin; bs.in1 = ss.out; bs.op = 1; bs.mode = 1; bs.H = H; bs.W = W; bs.C = C;
                        bs.out = new_val(u, (size_t)H * W * C); tv[u] = bs.out;
                        bs.bytes = 8.0 * H * W * C;
                        add_step(bs);
                    }
                    se = true;
                } while (0);
                if (!se) {
                    if (!fused_sum) add_step(mp);
                    Step mf; mf.kind = S_MEAN_FINISH; mf.kclass = "mean"; mf.name = oname;
                    mf.in0 = mp.out; mf.H = H; mf.W = W; mf.C = C; mf.S = S;
                    mf.out = new_val(mt, (size_t)C); tv[mt] = mf.out;
                    add_step(mf);
                }
                break;
            }
            case OP_FULLY_CONNECTED: {
                int in_t = o.inputs[0];
                if (!P.is_const(o.inputs[1])) { *err = "FULLY_CONNECTED: weights must be constant at " + oname; return false; }
                const TflTensor& w = m.tensors[o.inputs[1]];
                const int C = w.shape[1], N = w.shape[0];
                const size_t in_elems = m.tensors[in_t].numel();
                if (C <= 0 || in_elems % (size_t)C) { *err = "FULLY_CONNECTED: input size is not a multiple of the weight width at " + oname; return false; }
                const int rows = (int)(in_elems / (size_t)C);          // leading dimensions flatten into GEMM rows
                int post = 0;
                int act = map_act(o.act, &post);
                if (act < 0) { *err = "FULLY_CONNECTED: unsupported fused activation"; return false; }
                int outt = o.outputs[0];
                if (act == ACT_NONE && !post && std::find(m.outputs.begin(), m.outputs.end(), outt) == m.outputs.end())
                    outt = trailing_act(outt, &act);
                Step s; s.kind = S_PW; s.kclass = "pw_gemm"; s.name = oname;
                s.in0 = need_val(in_t);
                if (s.in0 < 0) { *err = "FULLY_CONNECTED: input has no value"; return false; }
                s.H = rows; s.W = 1; s.C = C; s.Ho = rows; s.Wo = 1; s.Co = N; s.act = act;
                s.flops = 2.0 * rows * C * N;
                s.bytes = 4.0 * rows * (C + N);
                s.wbytes = 4.0 * C * N;
                size_t boff = (o.inputs.size() > 2 && o.inputs[2] >= 0) ? wconst(o.inputs[2]) : SIZE_MAX;
                emit(s, outt, (size_t)rows * N, post, wconst(o.inputs[1]), boff);
                break;
            }
            case OP_BATCH_MATMUL: {
                // activation x constant matrix only (a dense layer written as a matmul): W[N][K] = rhs^T at plan time
                int in_t = o.inputs[0], r_t = o.inputs[1];
                if (!P.is_const(r_t) || m.tensors[r_t].shape.size() != 2 || o.adj_x) { *err = "BATCH_MATMUL: only activation x constant [K,N] is supported at " + oname; return false; }
                const TflTensor& r = m.tensors[r_t];
                const int K = o.adj_y ? r.shape[1] : r.shape[0], N = o.adj_y ? r.shape[0] : r.shape[1];
                const size_t in_elems = m.tensors[in_t].numel();
                if (K <= 0 || in_elems % (size_t)K) { *err = "BATCH_MATMUL: inner dimensions disagree at " + oname; return false; }
                std::vector<float> wt((size_t)N * K);
                for (int n = 0; n < N; n++)
                    for (int k = 0; k < K; k++) wt[(size_t)n * K + k] = o.adj_y ? r.f32()[(size_t)n * K + k] : r.f32()[(size_t)k * N + n];
                const int rows = (int)(in_elems / (size_t)K);
                Step s; s.kind = S_PW; s.kclass = "pw_gemm"; s.name = oname; s.in0 = need_val(in_t);
                if (s.in0 < 0) { *err = "BATCH_MATMUL: input has no value"; return false; }
                s.H = rows; s.W = 1; s.C = K; s.Ho = rows; s.Wo = 1; s.Co = N; s.act = ACT_NONE;
                s.flops = 2.0 * rows * K * N; s.bytes = 4.0 * rows * (K + N); s.wbytes = 4.0 * K * N;
                emit(s, o.outputs[0], (size_t)rows * N, 0, wpush(wt.data(), wt.size()));
                break;
            }
            case OP_LOGISTIC: case OP_RELU: case OP_RELU6: case OP_HARD_SWISH: {
                int vin = need_val(o.inputs[0]);
                if (vin < 0) { *err = std::string(op_name(o.code)) + ": input has no value"; return false; }
                Step s; s.kind = S_UNARY; s.kclass = "elementwise"; s.name = oname; s.in0 = vin;
                s.act = o.code == OP_LOGISTIC ? ACT_SIGMOID : o.code == OP_RELU ? ACT_RELU : o.code == OP_RELU6 ? ACT_RELU6 : ACT_HARD_SWISH;
                s.out = new_val(o.outputs[0], vals[vin].elems); tv[o.outputs[0]] = s.out;
                s.bytes = 8.0 * vals[vin].elems;
                add_step(s);
                break;
            }
            case OP_TANH: case OP_EXP: case OP_LOG: case OP_SQRT: case OP_RSQRT: case OP_ABS: case OP_NEG: case OP_SQUARE:
            case OP_LEAKY_RELU: case OP_ELU: case OP_SIN: case OP_COS: case OP_FLOOR: case OP_CEIL: case OP_ROUND:
            case OP_RELU_N1_TO_1: case OP_GELU: {
                int vin = need_val(o.inputs[0]);
                if (vin < 0) { *err = std::string(op_name(o.code)) + ": input has no value"; return false; }
                Step s; s.kind = S_EW_UNARY; s.kclass = "elementwise"; s.name = oname; s.in0 = vin;
                switch (o.code) {
                    case OP_TANH: s.op = U_TANH; break; case OP_EXP: s.op = U_EXP; break; case OP_LOG: s.op = U_LOG; break;
                    case OP_SQRT: s.op = U_SQRT; break; case OP_RSQRT: s.op = U_RSQRT; break; case OP_ABS: s.op = U_ABS; break;
                    case OP_NEG: s.op = U_NEG; break; case OP_SQUARE: s.op = U_SQUARE; break; case OP_LEAKY_RELU: s.op = U_LEAKY_RELU; break;
                    case OP_ELU: s.op = U_ELU; break; case OP_SIN: s.op = U_SIN; break; case OP_COS: s.op = U_COS; break;
                    case OP_FLOOR: s.op = U_FLOOR; break; case OP_CEIL: s.op = U_CEIL; break; case OP_ROUND: s.op = U_ROUND; break;
                    case OP_RELU_N1_TO_1: s.op = U_RELU_N1_TO_1; break;
                    default: s.op = o.approximate ? U_GELU_TANH : U_GELU; break;
                }
                s.g.alpha = o.alpha;
                s.out = new_val(o.outputs[0], vals[vin].elems); tv[o.outputs[0]] = s.out;
                s.bytes = 8.0 * vals[vin].elems;
                add_step(s);
                break;
            }
            case OP_ADD: case OP_MUL: case OP_SUB: case OP_DIV: case OP_POW: case OP_MAXIMUM: case OP_MINIMUM:
            case OP_SQUARED_DIFFERENCE: {
                // numpy-style broadcasting over right-aligned shapes; either operand may be a constant
                if (o.act < 0 || o.act > 4) { *err = "binary op: unsupported fused activation"; return false; }
                Operand A = operand(o.inputs[0]), B = operand(o.inputs[1]);
                if (!A.ok || !B.ok) { *err = std::string(op_name(o.code)) + ": operand has no value at " + oname; return false; }
                if (A.val < 0 && B.val < 0) { *err = std::string(op_name(o.code)) + ": constant expression (no activation operand) at " + oname; return false; }
                int da[5], db[5], dz[5];
                if (!shape5(m.tensors[o.inputs[0]].shape, da) || !shape5(m.tensors[o.inputs[1]].shape, db) ||
                    !shape5(m.tensors[o.outputs[0]].shape, dz)) { *err = "binary op: rank > 4 (+batch) unsupported at " + oname; return false; }
                for (int k = 0; k < 5; k++)
                    if ((da[k] != dz[k] && da[k] != 1) || (db[k] != dz[k] && db[k] != 1)) { *err = "binary op: shapes do not broadcast at " + oname; return false; }
                long sta[5], stb[5];
                dense_strides(da, sta); dense_strides(db, stb);
                Step s; s.kind = S_EW_BINARY; s.kclass = "elementwise"; s.name = oname; s.act = o.act;
                s.op = o.code == OP_ADD ? B_ADD : o.code == OP_MUL ? B_MUL : o.code == OP_SUB ? B_SUB : o.code == OP_DIV ? B_DIV :
                       o.code == OP_POW ? B_POW : o.code == OP_MAXIMUM ? B_MAX : o.code == OP_MINIMUM ? B_MIN : B_SQDIFF;
                for (int k = 0; k < 4; k++) {
                    s.g.d[k] = dz[k + 1];
                    s.g.sa[k] = da[k + 1] == 1 ? 0 : sta[k + 1];
                    s.g.sb[k] = db[k + 1] == 1 ? 0 : stb[k + 1];
                }
                s.in0 = A.val; s.in1 = B.val; s.g.a_const = A.val < 0; s.g.b_const = B.val < 0;
                const size_t elems = m.tensors[o.outputs[0]].numel();
                s.out = new_val(o.outputs[0], elems); tv[o.outputs[0]] = s.out;
                s.bytes = 12.0 * elems;
                add_step(s, A.woff, B.woff);
                break;
            }
            case OP_AVERAGE_POOL_2D: case OP_MAX_POOL_2D: {
                int H, W, C, Ho, Wo, Co;
                if (!hwc(o.inputs[0], &H, &W, &C) || !hwc(o.outputs[0], &Ho, &Wo, &Co) || Co != C || o.filter_h < 1 || o.filter_w < 1) {
                    *err = std::string(op_name(o.code)) + ": unsupported shapes at " + oname; return false;
                }
                if (o.act < 0 || o.act > 4) { *err = "pool: unsupported fused activation"; return false; }
                int vin = need_val(o.inputs[0]);
                if (vin < 0) { *err = std::string(op_name(o.code)) + ": input has no value"; return false; }
                Step s; s.kind = S_POOL; s.kclass = "pool"; s.name = oname; s.in0 = vin; s.act = o.act;
                s.mode = o.code == OP_AVERAGE_POOL_2D ? 0 : 1;
                s.H = H; s.W = W; s.C = C; s.Ho = Ho; s.Wo = Wo; s.Co = C; s.kh = o.filter_h; s.kw = o.filter_w; s.sh = o.stride_h; s.sw = o.stride_w;
                if (o.padding == 0) {
                    s.pt = std::max((Ho - 1) * s.sh + s.kh - H, 0) / 2;
                    s.pl = std::max((Wo - 1) * s.sw + s.kw - W, 0) / 2;
                }
                s.out = new_val(o.outputs[0], (size_t)Ho * Wo * C); tv[o.outputs[0]] = s.out;
                s.bytes = 4.0 * ((double)H * W * C + (double)Ho * Wo * C);
                add_step(s);
                break;
            }
            case OP_SOFTMAX: {
                int vin = need_val(o.inputs[0]);
                const auto& sh = m.tensors[o.inputs[0]].shape;
                if (vin < 0 || sh.empty()) { *err = "SOFTMAX: input has no value"; return false; }
                Step s; s.kind = S_SOFTMAX; s.kclass = "softmax"; s.name = oname; s.in0 = vin;
                s.C = sh.back(); s.H = (int)(vals[vin].elems / (size_t)std::max(s.C, 1)); s.g.alpha = o.beta;
                s.out = new_val(o.outputs[0], vals[vin].elems); tv[o.outputs[0]] = s.out;
                s.bytes = 8.0 * vals[vin].elems;
                add_step(s);
                break;
            }
            case OP_CONCATENATION: {
                int dz[5];
                const int rank = (int)m.tensors[o.outputs[0]].shape.size();
                if (!shape5(m.tensors[o.outputs[0]].shape, dz) || rank < 1) { *err = "CONCATENATION: rank unsupported at " + oname; return false; }
                int ax = o.axis < 0 ? o.axis + rank : o.axis;
                if (ax < 0 || ax >= rank) { *err = "CONCATENATION: axis out of range at " + oname; return false; }
                const int ax5 = ax + (5 - std::min(rank, 5));
                if (ax5 < 1) { *err = "CONCATENATION: cannot concatenate along the batch dimension at " + oname; return false; }
                if (o.act != 0) { *err = "CONCATENATION: fused activation unsupported"; return false; }
                long sto[5]; dense_strides(dz, sto);
                const size_t elems = m.tensors[o.outputs[0]].numel();
                const int vout = new_val(o.outputs[0], elems);
                tv[o.outputs[0]] = vout;
                long off = 0; int total_ax = 0;
                for (size_t ii = 0; ii < o.inputs.size(); ii++) {
                    Operand A = operand(o.inputs[ii]);
                    int di[5];
                    if (!A.ok || !shape5(m.tensors[o.inputs[ii]].shape, di)) { *err = "CONCATENATION: operand has no value at " + oname; return false; }
                    for (int k = 0; k < 5; k++) if (k != ax5 && di[k] != dz[k]) { *err = "CONCATENATION: operand shapes disagree at " + oname; return false; }
                    long sti[5]; dense_strides(di, sti);
                    Step s; s.kind = S_COPY; s.kclass = "copy"; s.name = oname + "/" + std::to_string(ii); s.in0 = A.val; s.out = vout;
                    s.g.a_const = A.val < 0;
                    for (int k = 0; k < 4; k++) { s.g.d[k] = di[k + 1]; s.g.sa[k] = sti[k + 1]; s.g.so[k] = sto[k + 1]; }
                    s.g.offo = off * sto[ax5];
                    s.bytes = 8.0 * m.tensors[o.inputs[ii]].numel();
                    add_step(s, A.woff);
                    off += di[ax5]; total_ax += di[ax5];
                }
                if (total_ax != dz[ax5]) { *err = "CONCATENATION: operand sizes do not add up at " + oname; return false; }
                break;
            }
            case OP_STRIDED_SLICE: case OP_SLICE: {
                int vin = need_val(o.inputs[0]);
                const auto& ish = m.tensors[o.inputs[0]].shape;
                const int rank = (int)ish.size();
                if (vin < 0) { *err = std::string(op_name(o.code)) + ": input has no value at " + oname; return false; }
                for (size_t k = 1; k < o.inputs.size(); k++)
                    if (!P.is_const(o.inputs[k]) || (int)m.tensors[o.inputs[k]].numel() != rank) { *err = std::string(op_name(o.code)) + ": begin/end/strides must be constant vectors of the input rank at " + oname; return false; }
                if (o.code == OP_STRIDED_SLICE && (o.ellipsis_mask || o.new_axis_mask)) { *err = "STRIDED_SLICE: ellipsis / new-axis masks unsupported at " + oname; return false; }
                int di[5];
                if (!shape5(ish, di) || rank > 5) { *err = std::string(op_name(o.code)) + ": rank unsupported at " + oname; return false; }
                const int lead = 5 - rank;
                long sti[5]; dense_strides(di, sti);
                int cnt5[5] = {1, 1, 1, 1, 1}; long st5[5] = {0, 0, 0, 0, 0}; long off = 0;
                const int32_t* bg = m.tensors[o.inputs[1]].i32();
                const int32_t* en = m.tensors[o.inputs[2]].i32();
                const int32_t* sr = o.code == OP_STRIDED_SLICE ? m.tensors[o.inputs[3]].i32() : nullptr;
                size_t total = 1;
                for (int k = 0; k < rank; k++) {
                    const int dim = ish[k];
                    long b, e, st;
                    if (o.code == OP_SLICE) {                              // begin / size; size -1 = to the end
                        b = bg[k]; st = 1; e = en[k] < 0 ? dim : b + en[k];
                    } else {
                        st = sr[k];
                        if (st == 0) { *err = "STRIDED_SLICE: zero stride at " + oname; return false; }
                        b = bg[k]; e = en[k];
                        if (b < 0) b += dim;
                        if (e < 0) e += dim;
                        if ((o.begin_mask >> k) & 1) b = st > 0 ? 0 : dim - 1;
                        if ((o.end_mask >> k) & 1) e = st > 0 ? dim : -1;
                        if ((o.shrink_axis_mask >> k) & 1) { e = b + 1; st = 1; }
                        if (st > 0) { b = std::min<long>(std::max<long>(b, 0), dim); e = std::min<long>(std::max<long>(e, 0), dim); }
                        else { b = std::min<long>(std::max<long>(b, -1), dim - 1); e = std::min<long>(std::max<long>(e, -1), dim - 1); }
                    }
                    long n = st > 0 ? (e > b ? (e - b + st - 1) / st : 0) : (b > e ? (b - e - st - 1) / (-st) : 0);
                    if (n <= 0 || b < 0 || b >= dim || b + (n - 1) * st < 0 || b + (n - 1) * st >= dim) { *err = std::string(op_name(o.code)) + ": empty or out-of-range slice at " + oname; return false; }
                    cnt5[lead + k] = (int)n; st5[lead + k] = st * sti[lead + k]; off += b * sti[lead + k];
                    total *= (size_t)n;
                }
                if (cnt5[0] != 1 || total != m.tensors[o.outputs[0]].numel()) { *err = std::string(op_name(o.code)) + ": slice does not match the output shape at " + oname; return false; }
                Step s; s.kind = S_COPY; s.kclass = "copy"; s.name = oname; s.in0 = vin;
                long so = 1;
                for (int k = 3; k >= 0; k--) { s.g.d[k] = cnt5[k + 1]; s.g.sa[k] = st5[k + 1]; s.g.so[k] = so; so *= cnt5[k + 1]; }
                s.g.offa = off;
                s.out = new_val(o.outputs[0], total); tv[o.outputs[0]] = s.out;
                s.bytes = 8.0 * total;
                add_step(s);
                break;
            }
            case OP_TRANSPOSE: case OP_REVERSE_V2: {
                int vin = need_val(o.inputs[0]);
                const auto& ish = m.tensors[o.inputs[0]].shape;
                const int rank = (int)ish.size();
                int di[5];
                if (vin < 0 || !P.is_const(o.inputs[1]) || !shape5(ish, di) || rank > 5) { *err = std::string(op_name(o.code)) + ": unsupported operands at " + oname; return false; }
                const int lead = 5 - rank;
                long sti[5]; dense_strides(di, sti);
                int dz[5]; long st5[5]; long off = 0;
                const TflTensor& pt = m.tensors[o.inputs[1]];
                if (o.code == OP_TRANSPOSE) {
                    if ((int)pt.numel() != rank) { *err = "TRANSPOSE: permutation length != rank at " + oname; return false; }
                    int perm5[5]; std::vector<char> seen(5, 0);
                    for (int k = 0; k < lead; k++) perm5[k] = k;
                    for (int k = 0; k < rank; k++) { int pk = pt.i32()[k]; if (pk < 0 || pk >= rank) { *err = "TRANSPOSE: bad permutation at " + oname; return false; } perm5[lead + k] = lead + pk; }
                    for (int k = 0; k < 5; k++) { if (seen[perm5[k]]) { *err = "TRANSPOSE: bad permutation at " + oname; return false; } seen[perm5[k]] = 1; }
                    if (perm5[0] != 0) { *err = "TRANSPOSE: the batch dimension cannot move at " + oname; return false; }
                    for (int k = 0; k < 5; k++) { dz[k] = di[perm5[k]]; st5[k] = sti[perm5[k]]; }
                } else {
                    for (int k = 0; k < 5; k++) { dz[k] = di[k]; st5[k] = sti[k]; }
                    for (size_t q = 0; q < pt.numel(); q++) {
                        int ax = pt.i32()[q]; if (ax < 0) ax += rank;
                        if (ax < 0 || ax >= rank || lead + ax == 0) { *err = "REVERSE_V2: bad axis at " + oname; return false; }
                        off += (long)(di[lead + ax] - 1) * sti[lead + ax]; st5[lead + ax] = -sti[lead + ax];
                    }
                }
                Step s; s.kind = S_COPY; s.kclass = "copy"; s.name = oname; s.in0 = vin;
                long so = 1;
                for (int k = 3; k >= 0; k--) { s.g.d[k] = dz[k + 1]; s.g.sa[k] = st5[k + 1]; s.g.so[k] = so; so *= dz[k + 1]; }
                s.g.offa = off;
                s.out = new_val(o.outputs[0], vals[vin].elems); tv[o.outputs[0]] = s.out;
                s.bytes = 8.0 * vals[vin].elems;
                add_step(s);
                break;
            }
            case OP_GATHER: {
                // GATHER of an activation with a CONSTANT selector that is affine in its (<= 2) indices - tf.signal.frame's
                // sliding window idx[f][q] = f * step + q, a strided pick idx[i] = i0 + i * a, a single element - is a strided
                // view of the input: one copy kernel.  (Inside a recognised audio front-end the framing never runs: the STFT
                // kernel fetches its frames from the clip; this is the literal path for graphs the recogniser does not take.)
                int vin = need_val(o.inputs[0]);
                const auto& ish = m.tensors[o.inputs[0]].shape;
                const int rank = (int)ish.size();
                if (vin < 0 || o.inputs.size() != 2 || !P.is_const(o.inputs[1]) || m.tensors[o.inputs[1]].type != TT_INT32 || o.batch_dims != 0) { *err = "GATHER: needs an activation and a constant int32 selector at " + oname; return false; }
                const TflTensor& sel = m.tensors[o.inputs[1]];
                int ax = o.axis < 0 ? o.axis + rank : o.axis;
                if (ax < 1 || ax >= rank || sel.shape.size() > 2 || sel.numel() < 1) { *err = "GATHER: unsupported axis / selector rank at " + oname; return false; }
                const int n0 = sel.shape.empty() ? 1 : sel.shape[0], n1 = sel.shape.size() == 2 ? sel.shape[1] : 1;
                const int32_t* sv = sel.i32();
                const long i0 = sv[0], a0 = n0 > 1 ? (long)sv[n1] - sv[0] : 0, a1 = n1 > 1 ? (long)sv[1] - sv[0] : 0;
                for (int f = 0; f < n0; f++)
                    for (int q = 0; q < n1; q++) {
                        const long want = i0 + f * a0 + q * a1;
                        if (sv[(size_t)f * n1 + q] != want || want < 0 || want >= ish[ax]) { *err = "GATHER: the constant selector is not affine (or out of range) at " + oname; return false; }
                    }
                long pre = 1, post = 1;
                for (int k = 1; k < ax; k++) pre *= ish[k];
                for (int k = ax + 1; k < rank; k++) post *= ish[k];
                if (ish[0] != 1) { *err = "GATHER: batch dimension must be 1 at " + oname; return false; }
                Step s; s.kind = S_COPY; s.kclass = "copy"; s.name = oname; s.in0 = vin;
                // view [pre, n0, n1, post] of the input [pre, ish[ax], post]
                const long din[4] = {pre, n0, n1, post};
                const long sin_[4] = {(long)ish[ax] * post, a0 * post, a1 * post, 1};
                long so = 1;
                for (int k = 3; k >= 0; k--) { s.g.d[k] = (int)din[k]; s.g.sa[k] = sin_[k]; s.g.so[k] = so; so *= din[k]; }
                s.g.offa = i0 * post;
                const size_t elems = m.tensors[o.outputs[0]].numel();
                if ((size_t)(pre * n0 * n1 * post) != elems) { *err = "GATHER: output shape mismatch at " + oname; return false; }
                s.out = new_val(o.outputs[0], elems); tv[o.outputs[0]] = s.out;
                s.bytes = 8.0 * elems;
                add_step(s);
                break;
            }
            case OP_PAD: case OP_PADV2: {
                int vin = need_val(o.inputs[0]);
                const auto& ish = m.tensors[o.inputs[0]].shape;
                const int rank = (int)ish.size();
                int di[5], dz[5];
                if (vin < 0 || !P.is_const(o.inputs[1]) || (int)m.tensors[o.inputs[1]].numel() != 2 * rank || !shape5(ish, di) ||
                    !shape5(m.tensors[o.outputs[0]].shape, dz) || rank > 5) { *err = std::string(op_name(o.code)) + ": unsupported operands at " + oname; return false; }
                float fillv = 0.0f;
                if (o.code == OP_PADV2 && !P.const_scalar(o.inputs[2], &fillv)) { *err = "PADV2: pad value must be a scalar constant at " + oname; return false; }
                const int lead = 5 - rank;
                const int32_t* pv = m.tensors[o.inputs[1]].i32();
                long sto[5]; dense_strides(dz, sto);
                long off = 0;
                for (int k = 0; k < rank; k++) {
                    if (pv[2 * k] < 0 || pv[2 * k + 1] < 0 || di[lead + k] + pv[2 * k] + pv[2 * k + 1] != dz[lead + k]) { *err = std::string(op_name(o.code)) + ": paddings disagree with the output shape at " + oname; return false; }
                    off += (long)pv[2 * k] * sto[lead + k];
                }
                if (dz[0] != 1) { *err = std::string(op_name(o.code)) + ": batch padding unsupported at " + oname; return false; }
                Step s; s.kind = S_COPY; s.kclass = "copy"; s.name = oname; s.in0 = vin;
                long si = 1;
                for (int k = 3; k >= 0; k--) { s.g.d[k] = di[k + 1]; s.g.sa[k] = si; si *= di[k + 1]; s.g.so[k] = sto[k + 1]; }
                s.g.offo = off; s.g.fill = true; s.g.alpha = fillv;
                const size_t elems = m.tensors[o.outputs[0]].numel();
                s.out = new_val(o.outputs[0], elems); tv[o.outputs[0]] = s.out;
                s.bytes = 4.0 * (elems + vals[vin].elems);
                add_step(s);
                break;
            }
            case OP_SPLIT: {
                // inputs: axis (constant scalar), value; outputs: num_splits equal slices along the axis
                int vin = need_val(o.inputs[1]);
                const auto& ish = m.tensors[o.inputs[1]].shape;
                const int rank = (int)ish.size();
                int di[5];
                if (vin < 0 || !P.is_const(o.inputs[0]) || m.tensors[o.inputs[0]].numel() != 1 || !shape5(ish, di) || rank > 5) { *err = "SPLIT: unsupported operands at " + oname; return false; }
                int ax = m.tensors[o.inputs[0]].i32()[0]; if (ax < 0) ax += rank;
                const int lead = 5 - rank, nsp = (int)o.outputs.size();
                if (ax < 0 || ax >= rank || lead + ax == 0 || nsp < 1 || di[lead + ax] % nsp) { *err = "SPLIT: bad axis or split count at " + oname; return false; }
                long sti[5]; dense_strides(di, sti);
                int dz[5]; for (int k = 0; k < 5; k++) dz[k] = di[k];
                dz[lead + ax] = di[lead + ax] / nsp;
                size_t elems = 1; for (int k = 0; k < 5; k++) elems *= (size_t)dz[k];
                for (int q = 0; q < nsp; q++) {
                    if (m.tensors[o.outputs[q]].numel() != elems) { *err = "SPLIT: output shape mismatch at " + oname; return false; }
                    Step s; s.kind = S_COPY; s.kclass = "copy"; s.name = m.tensors[o.outputs[q]].name; s.in0 = vin;
                    long so = 1;
                    for (int k = 3; k >= 0; k--) { s.g.d[k] = dz[k + 1]; s.g.sa[k] = sti[k + 1]; s.g.so[k] = so; so *= dz[k + 1]; }
                    s.g.offa = (long)q * dz[lead + ax] * sti[lead + ax];
                    s.out = new_val(o.outputs[q], elems); tv[o.outputs[q]] = s.out;
                    s.bytes = 8.0 * elems;
                    add_step(s);
                }
                break;
            }
            case OP_SUM: case OP_REDUCE_MAX: case OP_REDUCE_MIN: case OP_REDUCE_PROD:
            generic_reduce: {
                int vin = need_val(o.inputs[0]);
                const auto& ish = m.tensors[o.inputs[0]].shape;
                const int rank = (int)ish.size();
                int di[5];
                if (vin < 0 || !P.is_const(o.inputs[1]) || !shape5(ish, di) || rank > 5) { *err = std::string(op_name(o.code)) + ": unsupported operands at " + oname; return false; }
                const int lead = 5 - rank;
                const TflTensor& ax = m.tensors[o.inputs[1]];
                Step s; s.kind = S_REDUCE; s.kclass = "reduce"; s.name = oname; s.in0 = vin;
                for (size_t q = 0; q < ax.numel(); q++) {
                    int a = ax.i32()[q]; if (a < 0) a += rank;
                    if (a < 0 || a >= rank || lead + a == 0) { *err = std::string(op_name(o.code)) + ": bad axis at " + oname; return false; }
                    s.g.mask |= 1 << (lead + a - 1);
                }
                size_t elems = 1;
                for (int k = 0; k < 4; k++) { s.g.d[k] = di[k + 1]; if (!((s.g.mask >> k) & 1)) elems *= (size_t)di[k + 1]; }
                if (elems != m.tensors[o.outputs[0]].numel()) { *err = std::string(op_name(o.code)) + ": output shape mismatch at " + oname; return false; }
                s.op = o.code == OP_MEAN ? R_MEAN : o.code == OP_SUM ? R_SUM : o.code == OP_REDUCE_MAX ? R_MAX : o.code == OP_REDUCE_MIN ? R_MIN : R_PROD;
                s.out = new_val(o.outputs[0], elems); tv[o.outputs[0]] = s.out;
                s.bytes = 4.0 * (vals[vin].elems + elems);
                add_step(s);
                break;
            }
            case OP_RESHAPE: case OP_SQUEEZE: case OP_EXPAND_DIMS: case OP_CAST: {
                int vin = need_val(o.inputs[0]);
                if (vin < 0) { *err = std::string(op_name(o.code)) + ": input has no value"; return false; }
                if (o.code == OP_CAST && (m.tensors[o.inputs[0]].type != TT_FLOAT32 || m.tensors[o.outputs[0]].type != TT_FLOAT32)) {
                    *err = "CAST: only float32 -> float32 outside the recognised front-end at " + oname; return false;
                }
                if (m.tensors[o.outputs[0]].numel() != vals[vin].elems) { *err = "reshape changes element count"; return false; }
                tv[o.outputs[0]] = vin;   // contiguous alias
                break;
            }
            default:
                *err = std::string("unsupported operator ") + op_name(o.code) + " (code " + std::to_string(o.code) + ") at " + oname;
                return false;
        }
    }

    // ---------------------------------------------------------------- outputs
    {
        int lt = m.outputs[0];
        auto it = tv.find(lt);
        if (it == tv.end()) { *err = "graph output 0 was not produced by a supported op"; return false; }
        v_logits = it->second;
        n_classes = (int)vals[v_logits].elems;
        vals[v_logits].external = true;
        if (m.outputs.size() > 1) {
            auto ie = tv.find(m.outputs[1]);
            if (ie == tv.end()) { *err = "graph output 1 (embedding) was not produced by a supported op"; return false; }
            v_emb = ie->second;
            emb_dim = (int)vals[v_emb].elems;
        }
    }

    tensor_value = tv;
    // ---------------------------------------------------------------- liveness + arena
    for (int si = 0; si < (int)steps.size(); si++) {
        for (int v : {steps[si].in0, steps[si].in1, steps[si].in2, steps[si].out, steps[si].out2}) {
            if (v < 0) continue;
            if (vals[v].first < 0) vals[v].first = si;
            vals[v].last = si;
        }
    }
    if (v_emb >= 0) vals[v_emb].last = (int)steps.size();      // keep until copy-out
    // The plan is laid out twice: for max_batch clips (unsplit calls) and for one lane's share.  Lanes drift apart in
    // time, and values that reuse each other's memory have different per-clip sizes, so addressing lanes as clip offsets
    // into ONE liveness-reused layout lets lane 0's step-k output overwrite lane 1's still-live step-j input: each lane
    // gets its own copy of the (smaller) lane layout instead.
    // the kernels index activations with 32-bit element offsets (batch x per-clip elements): refuse a max_batch that a
    // tensor of this model would overflow rather than compute with wrapped addresses
    for (const Value& v : vals)
        if (v.elems * (size_t)max_batch >= ((size_t)1 << 31)) {
            *err = "max_batch too large for this model: a " + std::to_string(v.elems) + "-element activation times " +
                   std::to_string(max_batch) + " clips exceeds 32-bit indexing";
            *code = BNHIP_E_INVALID;
            return false;
        }
    auto plan_arena = [&](size_t cap, bool lane_plan) -> size_t {
        struct Block { size_t off, size; };
        std::vector<Block> free_list;
        size_t top = 0;
        auto off_of = [&](int v) -> size_t& { return lane_plan ? vals[v].offset_lane : vals[v].offset; };
        std::vector<std::vector<int>> born(steps.size() + 1), dies(steps.size() + 2);
        for (int v = 0; v < (int)vals.size(); v++) {
            if (vals[v].external || vals[v].first < 0) continue;
            born[vals[v].first].push_back(v);
            dies[std::min<size_t>(vals[v].last + 1, steps.size() + 1)].push_back(v);
        }
        for (size_t si = 0; si <= steps.size(); si++) {
            for (int v : dies[si]) {
                free_list.push_back(Block{off_of(v), align_up(vals[v].elems * 4 * cap, 256)});
                // coalesce
                std::sort(free_list.begin(), free_list.end(), [](const Block& a, const Block& b) { return a.off < b.off; });
                std::vector<Block> merged;
                for (auto& b : free_list) {
                    if (!merged.empty() && merged.back().off + merged.back().size == b.off) merged.back().size += b.size;
                    else merged.push_back(b);
                }
                free_list.swap(merged);
            }
            if (si == steps.size()) break;
            for (int v : born[si]) {
                size_t need = align_up(vals[v].elems * 4 * cap, 256);
                int best = -1;
                if (no_reuse) free_list.clear();
                for (int i = 0; i < (int)free_list.size(); i++)
                    if (free_list[i].size >= need && (best < 0 || free_list[i].size < free_list[best].size)) best = i;
                if (best >= 0) {
                    off_of(v) = free_list[best].off;
                    free_list[best].off += need; free_list[best].size -= need;
                    if (free_list[best].size == 0) free_list.erase(free_list.begin() + best);
                } else {
                    // extend the arena top (absorb a trailing free block if adjacent)
                    if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {
                        off_of(v) = free_list.back().off;
                        top = off_of(v) + need;
                        free_list.pop_back();
                    } else {
                        off_of(v) = top; top += need;
                    }
                }
            }
        }
        return top;
    };
    n_lanes = std::max(1, std::min(n_lanes, kMaxLanes));
    lane_cap = (max_batch + n_lanes - 1) / n_lanes;
    act_bytes = plan_arena((size_t)max_batch, false);
    lane_bytes = n_lanes > 1 ? align_up(plan_arena((size_t)lane_cap, true), 256) : 0;
    act_bytes = std::max(act_bytes, lane_bytes * (size_t)n_lanes);
    pick_split();

    // ---------------------------------------------------------------- device allocation
    w_bytes = wimg.size() * sizeof(float);
    if (plan_only) { device = -1; *code = BNHIP_OK; return true; }   // CPU-side planning only (tests, describe)
    *code = BNHIP_E_RUNTIME;
    HIPCHK(hipSetDevice(device));
    stream = kernel_stream(0);
    if (!stream) { *err = "hipStreamCreate failed"; return false; }
    n_lanes = std::max(1, std::min(n_lanes, kMaxLanes));
    if (n_lanes > 1) {
        HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        for (int i = 0; i < n_lanes - 1; i++) {
            lane_stream[i] = kernel_stream(1 + i);
            if (!lane_stream[i]) { *err = "hipStreamCreate failed"; return false; }
            HIPCHK(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
        }
    }
    own_stream = true;
    HIPCHK(hipMalloc((void**)&w_arena, std::max<size_t>(w_bytes, 256)));
    if (!defer_weights) HIPCHK(hipMemcpy(w_arena, wimg.data(), w_bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&act_arena, std::max<size_t>(act_bytes, 256)));
    HIPCHK(hipMalloc((void**)&mm_scratch, (size_t)kMaxDepth * kMaxLanes * kMinMaxScratch * sizeof(float)));
    HIPCHK(hipMemset(mm_scratch, 0, (size_t)kMaxDepth * kMaxLanes * kMinMaxScratch * sizeof(float)));      // (arrival counters start at zero and reset themselves)
    depth = std::max(1, std::min(depth, kMaxDepth));
    host_depth = std::max(1, std::min(host_depth, kMaxDepth));
    if (depth > 1 && !ensure_contexts(depth, err)) return false;
    for (size_t si = 0; si < steps.size(); si++) {
        const float** slots[4] = {&steps[si].w0, &steps[si].w1, &steps[si].w2, &steps[si].w3};
        for (int k = 0; k < 4; k++)
            if (step_w[k][si] != SIZE_MAX) *slots[k] = reinterpret_cast<const float*>(w_arena) + step_w[k][si];
        if (step_bx[si] != SIZE_MAX) {
            steps[si].wbx = reinterpret_cast<const uint16_t*>(reinterpret_cast<const float*>(w_arena) + step_bx[si]);
        }
        if (steps[si].kind == S_FRONTEND) {
            specs[steps[si].spec].G = reinterpret_cast<const double*>(steps[si].w0);
            specs[steps[si].spec].window = steps[si].w1;
        }
        if (steps[si].kind == S_STFT) {
            specs[steps[si].spec].window_full = steps[si].w0;
            specs[steps[si].spec].bins = reinterpret_cast<const int*>(steps[si].w1);
            specs[steps[si].spec].stft_tw = reinterpret_cast<const double*>(steps[si].w2);
        }
    }
    HIPCHK(hipMalloc((void**)&d_stage_in, (size_t)max_batch * n_samples * 4));
    HIPCHK(hipMalloc((void**)&d_stage_logits, (size_t)max_batch * n_classes * 4));
    HIPCHK(hipMalloc((void**)&d_post_conf, (size_t)max_batch * n_classes * 4));
    if (emb_dim) HIPCHK(hipMalloc((void**)&d_stage_emb, (size_t)max_batch * emb_dim * 4));
    if (!defer_weights) mark_bf16_storage();
    if (autotune && !defer_weights) tune_or_load();
    *code = BNHIP_OK;
    return true;
}

// Streams are pooled per DEVICE, not per engine: HIP maps all streams of a process on a device onto at most four hardware
// queues, so a second engine on the same GPU (another model, or the second shard of a {"devices":[0,0]} handle) that created
// streams of its own would push everybody onto shared queues - measured 0.61-0.69x the single-engine rate for two engines
// on one GPU.  Engines on one device therefore share its (up to four) kernel streams and its one copy stream; their work
// interleaves in stream order, which is all the ordering independent clips need.
namespace {
struct DevStreams { hipStream_t k[Engine::kMaxKStreams] = {nullptr, nullptr, nullptr, nullptr}; hipStream_t xfer = nullptr; int refs = 0; };
std::mutex g_ds_mu;
std::map<int, DevStreams> g_ds;
}  // namespace

hipStream_t Engine::kernel_stream(int i) {
    if (i < 0 || i >= kMaxKStreams || device < 0) return nullptr;
    if (kstream[i]) return kstream[i];
    std::lock_guard<std::mutex> lk(g_ds_mu);
    DevStreams& ds = g_ds[device];
    if (!ds_ref) { ds.refs++; ds_ref = true; }
    if (!ds.k[i] && hipStreamCreateWithFlags(&ds.k[i], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ds.k[i] = nullptr; }
    kstream[i] = ds.k[i];
    return kstream[i];
}
hipStream_t Engine::copy_stream() {
    if (device < 0) return nullptr;
    if (xfer_stream) return xfer_stream;
    std::lock_guard<std::mutex> lk(g_ds_mu);
    DevStreams& ds = g_ds[device];
    if (!ds_ref) { ds.refs++; ds_ref = true; }
    if (!ds.xfer) {
        // highest priority: copies are what should go first (and a queue pooled by priority is less likely to be a kernel queue)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&ds.xfer, hipStreamNonBlocking, greatest) != hipSuccess) { (void)hipGetLastError(); ds.xfer = nullptr; }
    }
    xfer_stream = ds.xfer;
    return xfer_stream;
}
void Engine::release_streams() {
    for (int i = 0; i < kMaxKStreams; i++) if (kstream[i]) hipStreamSynchronize(kstream[i]);
    if (xfer_stream) hipStreamSynchronize(xfer_stream);
    std::lock_guard<std::mutex> lk(g_ds_mu);
    if (ds_ref) {
        auto it = g_ds.find(device);
        if (it != g_ds.end() && --it->second.refs == 0) {
            for (auto st : it->second.k) if (st) hipStreamDestroy(st);
            if (it->second.xfer) hipStreamDestroy(it->second.xfer);
            g_ds.erase(it);
        }
        ds_ref = false;
    }
    for (int i = 0; i < kMaxKStreams; i++) kstream[i] = nullptr;
    xfer_stream = nullptr;
}

// Contexts 0..d-1 (stream, completion event, activation arena; context 0 shares the engine's own arena).  Idempotent.
bool Engine::ensure_contexts(int d, std::string* err, bool with_streams) {
    d = std::max(1, std::min(d, kMaxDepth));
    if (device < 0) { *err = "plan-only model has no contexts"; return false; }
    hipSetDevice(device);
    if (with_streams && !ev_ctx_fork) HIPCHK(hipEventCreateWithFlags(&ev_ctx_fork, hipEventDisableTiming));
    for (int c = 0; c < d; c++) {
        if (with_streams) {
            if (!ctx_stream[c]) { ctx_stream[c] = kernel_stream(1 + c); if (!ctx_stream[c]) { *err = "hipStreamCreate failed"; return false; } }
            if (!ev_ctx_done[c]) HIPCHK(hipEventCreateWithFlags(&ev_ctx_done[c], hipEventDisableTiming));
        }
        if (!ctx_arena[c]) {
            if (c == 0) ctx_arena[c] = act_arena;
            else HIPCHK(hipMalloc((void**)&ctx_arena[c], std::max<size_t>(act_bytes, 256)));
        }
    }
    return true;
}

void Engine::finish_deferred() {
    if (device < 0) return;
    hipSetDevice(device);
    mark_bf16_storage();
    if (autotune) tune_or_load();
    defer_weights = false;
}

// Where a plan's tuning comes from, in this order (round 6: "one plan per library"):
//   1. BNHIP_TUNE_FILE          an experiment's own file (tools/profile_round.sh: bench, kernel trace and every PMC pass of one round)
//   2. the process cache        an engine of the same plan (model geometry, batch, depth, precision, switches, architecture) was
//                               already tuned in this process - the other shards of a multi-device handle, a second handle on the same
//                               model: they adopt its decisions, so a clip's bits do not depend on which engine it lands on
//   3. "tune_dir" / BNHIP_TUNE_DIR   a directory of recorded tunings named by that key (birdnet-go_amd/tune/ holds the ones the
//                               committed PMC passes ran on): the counters under profiles/ then describe the plan that is timed
//   4. the three create-time tuners (timing races: no two runs agree on every tile); recorded into 1 / 3 on request
// Switches that change what the tuners may pick (BNHIP_EXPDW_FORCE, BNHIP_DW_LDS, BNHIP_NO_DW_LDS, BNHIP_TUNE_BY_TIME) bypass 2 and 3;
// BNHIP_TUNE_CACHE=0 bypasses 2 only.
namespace {
std::mutex g_tune_mu;
std::map<std::string, std::string> g_tune_cache;
bool tune_experiment_env() {
    for (const char* n : {"BNHIP_EXPDW_FORCE", "BNHIP_DW_LDS", "BNHIP_NO_DW_LDS", "BNHIP_TUNE_BY_TIME"}) if (getenv(n)) return true;
    return false;
}
bool tune_cache_off() {                                     // BNHIP_TUNE_CACHE=0: tests of the directory path, A/B runs of the tuners themselves
    const char* e = getenv("BNHIP_TUNE_CACHE");
    return e && atoi(e) == 0;
}
bool read_text_file(const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096]; size_t n;
    out->clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) { out->append(buf, n); if (out->size() > (1u << 22)) break; }
    fclose(f);
    return true;
}
}  // namespace

void Engine::tune_or_load() {
    const char* tf = getenv("BNHIP_TUNE_FILE");
    const bool experiment = tune_experiment_env();
    const std::string key = tune_key();
    std::string text;
    auto adopted = [&](const std::string& src) {
        tune_source = src;
        if (getenv("BNHIP_DEBUG")) fprintf(stderr, "[bnhip] tuning %s: %s\n", key.c_str(), src.c_str());
    };
    if (tf && *tf && read_text_file(tf, &text) && apply_tuning_text(text)) { adopted(std::string("file:") + tf); }
    else {
        bool done = false;
        if (!experiment) {
            if (!tune_cache_off()) {
                std::lock_guard<std::mutex> lk(g_tune_mu);
                auto it = g_tune_cache.find(key);
                if (it != g_tune_cache.end()) text = it->second; else text.clear();
            }
            else text.clear();
            if (!text.empty() && apply_tuning_text(text)) { adopted("process-cache"); done = true; }
            if (!done && !tune_dir.empty() && read_text_file(tune_dir + "/" + key + ".tune", &text) && apply_tuning_text(text)) {
                adopted("dir:" + key + ".tune"); done = true;
            }
        }
        if (!done) {
            autotune_pw(); autotune_expdw(); autotune_dw();
            adopted("self-tuned");
            if (!experiment && !tune_dir.empty() && getenv("BNHIP_TUNE_RECORD")) {
                const std::string path = tune_dir + "/" + key + ".tune";
                FILE* ex = fopen(path.c_str(), "r");
                if (ex) fclose(ex); else save_tuning(path.c_str());
            }
        }
    }
    // BNHIP_TUNE_FILE records whatever this engine ended up with (timed here, or adopted from the cache / the directory) when the
    // file does not exist yet; never overwritten: another engine of the process may own it
    if (tf && *tf && tune_source.rfind("file:", 0) != 0) { FILE* ex = fopen(tf, "r"); if (ex) fclose(ex); else save_tuning(tf); }
    if (!experiment && !tune_cache_off()) {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        g_tune_cache.emplace(key, tuning_text());          // (first writer wins: every later engine of this plan adopts it)
    }
}

// One line per step: what the three create-time tuners decide (tile shapes, kernel flavour, LDS-staged depthwise, slab counts).
static const char* kTuneMagic = "bnhip-tuning-2";
// what a tuning was made FOR, beyond the step names: every step's geometry, the clip length and the device architecture
// (the header of version 1 carried none of them), the device's CU count (the fill rules depend on it)
static unsigned long long tune_plan_hash(const Engine& e) {
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](long long v) { for (int b = 0; b < 8; b++) { h ^= (unsigned long long)(v >> (8 * b)) & 0xff; h *= 1099511628211ull; } };
    mix(e.n_samples); mix(e.n_classes);
    for (const Step& s : e.steps) { mix((int)s.kind); mix(s.H); mix(s.W); mix(s.C); mix(s.Co); mix(s.Ho); mix(s.Wo); mix(s.kh); mix(s.kw); mix(s.sh); mix(s.sw); mix(s.act); }
    hipDeviceProp_t pr{};
    if (e.device >= 0 && hipGetDeviceProperties(&pr, e.device) == hipSuccess) {
        for (const char* c = pr.gcnArchName; *c && *c != ':'; c++) mix(*c);
        mix(pr.multiProcessorCount);
    } else (void)hipGetLastError();
    return h;
}
// file name / cache key of a plan's tuning: everything the header line checks
std::string Engine::tune_key() const {
    char b[160];
    snprintf(b, sizeof b, "%016llx_b%d_d%d_h%d_p%d_x%d_l%d_s%x", tune_plan_hash(*this), max_batch, depth, host_depth, precision, bf16x3, n_lanes, (unsigned)pw_sw);
    return b;
}
std::string Engine::tuning_text() const {
    std::ostringstream os;
    char b[1024];
    snprintf(b, sizeof b, "%s %zu %d %d %d %d %d %llx\n", kTuneMagic, steps.size(), max_batch, depth, host_depth, precision, bf16x3, tune_plan_hash(*this));
    os << b;
    for (size_t i = 0; i < steps.size(); i++) {
        const Step& s = steps[i];
        snprintf(b, sizeof b, "%zu %d %d %d %d %d %d %d %d %d ", i, (int)s.kind, s.nt, s.wm, s.nt_full, s.wm_full, s.shape, s.dwl, s.bx, s.S);
        os << b << s.name << "\n";
    }
    return os.str();
}
bool Engine::save_tuning(const char* path) const {
    FILE* f = fopen(path, "w");
    if (!f) return false;
    const std::string t = tuning_text();
    fwrite(t.data(), 1, t.size(), f);
    fclose(f);
    return true;
}
bool Engine::load_tuning(const char* path) {
    std::string text;
    if (!read_text_file(path, &text) || !apply_tuning_text(text)) return false;
    if (getenv("BNHIP_DEBUG")) fprintf(stderr, "[bnhip] tuning read from %s\n", path);
    return true;
}
bool Engine::apply_tuning_text(const std::string& text) {
    FILE* f = fmemopen(const_cast<char*>(text.data()), text.size(), "r");
    if (!f) return false;
    char magic[32] = {0}; size_t n = 0; int mb = 0, dp = 0, hd = 0, pr = 0, bx = 0; unsigned long long ph = 0;
    bool ok = fscanf(f, "%31s %zu %d %d %d %d %d %llx", magic, &n, &mb, &dp, &hd, &pr, &bx, &ph) == 8 && !strcmp(magic, kTuneMagic) && n == steps.size() &&
              mb == max_batch && dp == depth && hd == host_depth && pr == precision && bx == bf16x3 && ph == tune_plan_hash(*this);
    struct Row { int kind, nt, wm, ntf, wmf, shape, dwl, bx, S; };
    std::vector<Row> rows(ok ? n : 0);
    for (size_t i = 0; ok && i < n; i++) {
        size_t idx = 0; Row& r = rows[i]; char name[512] = {0};
        ok = fscanf(f, "%zu %d %d %d %d %d %d %d %d %d %511[^\n]", &idx, &r.kind, &r.nt, &r.wm, &r.ntf, &r.wmf, &r.shape, &r.dwl, &r.bx, &r.S, name) == 11 &&
             idx == i && r.kind == (int)steps[i].kind && steps[i].name == name;
        if (ok && steps[i].kind == S_PW) ok = r.nt >= 0 && r.nt <= 8 && r.ntf >= 0 && r.ntf <= 8 && r.wm >= 0 && r.wm <= 12 && r.wmf >= 0 && r.wmf <= 12 &&
                                              (((r.wm >= 5) == (steps[i].wm >= 5) && (r.wmf >= 5) == (steps[i].wm_full >= 5)) || !steps[i].wbx);       // (never switches the arithmetic family: bf16 storage was decided from it)
        if (ok && steps[i].kind == S_DW) {                    // the staged form only where its tuner would have timed it
            const Step& t = steps[i];
            DwParams dp{nullptr, nullptr, nullptr, nullptr, 1, t.H, t.W, t.C, t.Ho, t.Wo, t.kh, t.kw, t.sh, t.sw, t.pt, t.pl, t.act};
            ok = r.dwl == 0 || (r.dwl == 1 && dwconv_lds_supported(dp) && (t.out2 < 0 || dwconv_sum_slabs(dp) > 0));
        }
        if (ok && (steps[i].kind == S_EXPAND_DW || (steps[i].kind == S_DW && r.dwl))) {
            const bool st_ = steps[i].kind == S_EXPAND_DW && steps[i].mode == 1;
            // (the same geometry the tuner and the launcher use: a layer whose phase 1 runs on the bf16 pipe has no eight-wave shapes)
            const bool pipe16 = steps[i].kind == S_EXPAND_DW &&
                                expdw_sk_pipe16(steps[i].C, steps[i].act, st_, precision, steps[i].bx && steps[i].wbx != nullptr && bf16x3);
            const ExpDwGeo g{steps[i].kh, steps[i].sh, steps[i].H, steps[i].W, steps[i].Ho, steps[i].Wo, steps[i].pt, steps[i].pl, st_,
                             (steps[i].kind == S_EXPAND_DW && !pipe16) ? expdw_skw(steps[i].C, steps[i].act, st_) : 0};
            ok = r.shape >= 0 && r.shape < expdw_num_shapes() && expdw_shape_fits(r.shape, g) &&
                 (steps[i].kind != S_EXPAND_DW || r.bx == steps[i].bx);            // (never switches the arithmetic)
        }
    }
    fclose(f);
    if (!ok) return false;
    // A row changes what the tuners decide and nothing else; what follows from a decision (the tile count the consumers of the
    // per-tile sums index by) is recomputed from the plan, not read: a stale or edited file can pick a slower kernel, not a wrong one.
    for (size_t i = 0; i < n; i++) {
        Step& s = steps[i]; const Row& r = rows[i];
        if (s.kind == S_PW) { s.nt = r.nt; s.wm = r.wm; s.nt_full = r.ntf; s.wm_full = r.wmf; }
        const bool staged = s.kind == S_DW && r.dwl;
        if (s.kind == S_DW) s.dwl = r.dwl;
        if (s.kind != S_EXPAND_DW && !staged) continue;
        s.shape = r.shape;
        if (s.out2 < 0) continue;
        const bool st_ = s.kind == S_EXPAND_DW && s.mode == 1;
        const bool pipe16 = s.kind == S_EXPAND_DW && expdw_sk_pipe16(s.C, s.act, st_, precision, s.bx && s.wbx != nullptr && bf16x3);
        const ExpDwGeo g{s.kh, s.sh, s.H, s.W, s.Ho, s.Wo, s.pt, s.pl, st_, (s.kind == S_EXPAND_DW && !pipe16) ? expdw_skw(s.C, s.act, st_) : 0};
        s.S = expdw_shape_slabs(s.shape, g);
        for (auto& c : steps) if (&c != &s && c.in0 == s.out2) c.S = s.S;
    }
    return true;
}

// bf16 activation storage ("precision":"bf16" only; BNHIP_BF16_ACT=0 keeps fp32 storage for A/B runs).  A value is kept as
// bf16 in HBM when its one producer can round on the way out and every consumer can widen on the way in: the outputs of the
// split-bf16 GEMMs (no residual), of the tiled / LDS-staged depthwise kernels and of the fused expand + depthwise kernel,
// consumed only as the A operand of split-bf16 GEMMs (which round that operand to bf16 anyway - storing it rounded changes
// nothing for them, except that a fused squeeze-excite scale is applied to the rounded value) or as the input of a tiled /
// LDS-staged depthwise convolution (fp32 taps on bf16-rounded inputs: the only place the rounding is new).  That is exactly
// the set of 6x-expanded tensors, the ones the HBM-bound layers of an MBConv stack move.  Graph inputs / outputs, residual
// and scale operands, squeeze-excite sums and everything touched by any other kernel stay fp32.  Arena offsets are
// unchanged (a bf16 value uses the first half of its block).
void Engine::mark_bf16_storage() {
    for (auto& v : vals) v.half = false;
    if (precision != 1 || device < 0) return;
    if (const char* e = getenv("BNHIP_BF16_ACT")) if (atoi(e) == 0) return;
    auto bx_pw = [&](const Step& s) { return s.kind == S_PW && s.bx && s.wbx && s.wm >= 5 && s.wm_full >= 5; };
    // bf16 residual stream (BNHIP_BF16_RESID=0: block outputs stay fp32): the narrow tensors between the blocks - projection
    // outputs, read by the next expand, by the next projection's residual add and, in ratio-1 blocks, by a depthwise kernel -
    // are what the HBM-bound early projections of a bf16 engine mostly move
    bool resid = true;
    if (const char* e = getenv("BNHIP_BF16_RESID")) resid = atoi(e) != 0;
    auto dw_ok = [&](const Step& s) {
        if (s.kind != S_DW || (s.C & 3)) return false;
        DwParams p{nullptr, nullptr, nullptr, nullptr, 1, s.H, s.W, s.C, s.Ho, s.Wo, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.act};
        return dwconv_sum_slabs(p) > 0;
    };
    for (size_t vi = 0; vi < vals.size(); vi++) {
        Value& v = vals[vi];
        const int id = (int)vi;
        if (v.external || id == v_input || id == v_logits || id == v_emb) continue;
        int producers = 0, consumers = 0;
        bool ok = true;
        for (const Step& s : steps) {
            if (s.out == id) {
                producers++;
                bool stem_ok = false;
                if (s.kind == S_CONV_DIRECT && !s.w2) {        // the direct stem (not the MFMA one): 4-pixel kernels only
                    ConvParams cp{nullptr, nullptr, nullptr, nullptr, 1, s.H, s.W, s.C, s.Ho, s.Wo, s.Co, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.act};
                    stem_ok = conv_direct_bf16_ok(cp);
                }
                // (a projection that adds a residual may write bf16 too - the bf16 residual stream, `resid` below)
                const bool p_ok = (bx_pw(s) && (s.Co & 3) == 0 && (s.in2 < 0 || resid)) || dw_ok(s) || (s.kind == S_EXPAND_DW && (s.Co & 3) == 0) || stem_ok;
                if (!p_ok) ok = false;
            }
            if (s.out2 == id) ok = false;
            if (s.in1 == id) ok = false;
            if (s.in2 == id) {                                  // residual operand of a projection: its epilogue can widen bf16
                consumers++;
                if (!(resid && bx_pw(s) && (s.Co & 3) == 0)) ok = false;
            }
            if (s.in0 == id) {
                consumers++;
                // (the GEMM fetches 8 channels per load; the fused kernel takes bf16 input in its bf16-pipe chunk-loop form only)
                const bool c_ok = (bx_pw(s) && (s.C & 7) == 0) || dw_ok(s) ||
                                  (resid && s.kind == S_EXPAND_DW && s.mode != 1 && expdw_sk_pipe16(s.C, s.act, false, precision, s.bx && s.wbx != nullptr && bf16x3));
                if (!c_ok) ok = false;
            }
        }
        if (ok && producers == 1 && consumers >= 1) v.half = true;
    }
    // accounting: the algorithmic bytes of the steps that touch a bf16 value shrink with it
    for (Step& s : steps) {
        if (s.in0 >= 0 && vals[s.in0].half) s.bytes -= 2.0 * (double)vals[s.in0].elems;
        if (s.kind == S_PW && s.in2 >= 0 && vals[s.in2].half) s.bytes -= 2.0 * (double)vals[s.in2].elems;
        if (s.out >= 0 && vals[s.out].half) s.bytes -= 2.0 * (double)vals[s.out].elems;
    }
}

// Per-layer choice of the pw_gemm N-tile width: the best width depends on (M, N, K) through occupancy, grid size and
// padding in ways no closed-form rule captured (late layers at batch 256 have as few as 288 blocks), so each
// pointwise/FC step is timed once at create time on its real shapes and buffers (contents are irrelevant to timing).
// Tile shape of every fused expand+depthwise layer: time each shape of the kernel's table that fits the layer (the
// pixel-count cost model picks wrongly when a shape's LDS/register footprint costs more than its smaller halo saves -
// also for a pipelined engine: choosing by fewest expanded pixels there measured -1.5 %, unlike k_pw_gemm's tiles).
void Engine::autotune_expdw() {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int n = (max_batch + n_lanes - 1) / n_lanes;
    for (size_t si = 0; si < steps.size(); si++) {
        Step& s = steps[si];
        if (s.kind != S_EXPAND_DW) continue;
        float* in0 = vptr(s.in0, d_stage_in, d_stage_logits, nullptr);
        float* out = vptr(s.out, d_stage_in, d_stage_logits, nullptr);
        float* out2 = vptr(s.out2, d_stage_in, d_stage_logits, nullptr);
        float best = 1e30f; int best_idx = -1, best_bx = 0;
        const bool can_bx = s.wbx != nullptr && s.mode != 1 && bf16x3;
        const int bx_fixed = (can_bx && s.bx) ? 1 : 0;      // arithmetic is the planner's decision (shape rule); only the tile is timed
        const bool pipe16 = expdw_sk_pipe16(s.C, s.act, s.mode == 1, precision, s.bx && s.wbx != nullptr && bf16x3);
        const ExpDwGeo sg0{s.kh, s.sh, s.H, s.W, s.Ho, s.Wo, s.pt, s.pl, s.mode == 1, pipe16 ? 0 : expdw_skw(s.C, s.act, s.mode == 1)};
        // Two clocks per candidate: three launches back to back (how the layer runs inside a step: the next kernel's head
        // fills this one's tail) and the best of three isolated launches (what it costs when nothing covers its tail).  The
        // back-to-back time decides; the isolated one breaks near-ties (within 8 %), because a shape with few, long blocks can
        // look 7 % better back to back and be 60 % worse alone (b3 of the v2.4 stack: 8x32 tiles 201 vs 215 us back to back,
        // 329 vs 198 us isolated) while the pipelined throughput cannot tell the two apart.
        struct Cand { int idx; float b2b, iso; };
        std::vector<Cand> cands;
        for (int idx = 0; idx < expdw_num_shapes(); idx++) {
            if (!expdw_shape_fits(idx, sg0)) continue;
            const int bx = bx_fixed;
            auto go = [&]() {
                StemGeom sg{s.H2, s.W2, s.pt2, s.pl2};
                launch_expand_dw(in0, s.w0, s.w1, s.w2, s.w3, out, out2, n, s.H, s.W, s.C, s.Co, s.Ho, s.Wo, s.kh, s.sh, s.pt, s.pl,
                                 s.act, s.act2, idx, s.mode == 1 ? &sg : nullptr, stream, bx ? s.wbx : nullptr, precision,
                                 vals[s.out].half ? 1 : 0, vals[s.in0].half ? 1 : 0);          // the storage flavour the calls will run
            };
            go();
            hipEventRecord(a, stream);
            for (int r = 0; r < 3; r++) go();
            hipEventRecord(b, stream);
            hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            float iso = 1e30f;
            for (int r = 0; r < 3; r++) {
                hipEventRecord(a, stream); go(); hipEventRecord(b, stream);
                hipEventSynchronize(b);
                float t = 0; hipEventElapsedTime(&t, a, b);
                iso = std::min(iso, t);
            }
            if (getenv("BNHIP_DEBUG")) fprintf(stderr, "[bnhip] tune %-16s expand_dw shape=%d bx=%d: %.1f us back to back, %.1f us isolated\n", s.name.c_str(), idx, bx, ms / 3 * 1e3, iso * 1e3);
            cands.push_back({idx, ms / 3, iso});
            if (ms / 3 < best * 0.98f) { best = ms / 3; best_idx = idx; best_bx = bx; }
        }
        if (best_idx >= 0) {
            float best_iso = 0.f;
            for (const Cand& c : cands) if (c.idx == best_idx) best_iso = c.iso;
            for (const Cand& c : cands)
                if (c.b2b <= best * 1.08f && c.iso < best_iso * 0.8f) { best_idx = c.idx; best_iso = c.iso; }
        }
        if (best_idx < 0) continue;
        if (const char* f = getenv("BNHIP_EXPDW_FORCE")) {          // debug / A-B: "b3/expand+dw=0,b2/expand+dw=10"
            const std::string key = s.name + "=";
            const char* q = strstr(f, key.c_str());
            if (q) { const int idx = atoi(q + key.size()); if (expdw_shape_fits(idx, sg0)) best_idx = idx; }
        }
        s.shape = best_idx;
        s.bx = best_bx;
        if (s.out2 >= 0) {                       // the consumers of the per-tile sums index them by tile count
            s.S = expdw_shape_slabs(best_idx, sg0);
            for (auto& c : steps) if (&c != &s && c.in0 == s.out2) c.S = s.S;
        }
    }
    hipStreamSynchronize(stream);
    hipEventDestroy(a); hipEventDestroy(b);
    (void)hipGetLastError();
}

// Plain depthwise layers: the register-tiled kernel reads its taps through L1/L2, the LDS-staged form (the fused kernel's
// second phase on a copied footprint) pays a staging pass instead; which one wins depends on the filter size, the channel
// count and the image shape, so both are timed per layer - with every tile shape / orientation of the staged form.
void Engine::autotune_dw() {
    if (getenv("BNHIP_NO_DW_LDS")) return;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int n = (max_batch + n_lanes - 1) / n_lanes;
    for (size_t si = 0; si < steps.size(); si++) {
        Step& s = steps[si];
        if (s.kind != S_DW) continue;
        float* in0 = vptr(s.in0, d_stage_in, d_stage_logits, nullptr);
        float* out = vptr(s.out, d_stage_in, d_stage_logits, nullptr);
        float* out2 = vptr(s.out2, d_stage_in, d_stage_logits, nullptr);
        DwParams p{in0, s.w0, s.w1, out, n, s.H, s.W, s.C, s.Ho, s.Wo, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.act};
        if (!dwconv_lds_supported(p)) continue;
        if (s.out2 >= 0 && dwconv_sum_slabs(p) == 0) continue;      // (sums planned for the tiled kernel only)
        const ExpDwGeo g{s.kh, s.sh, s.H, s.W, s.Ho, s.Wo, s.pt, s.pl};
        auto timeit = [&](auto&& go) {
            go();
            hipEventRecord(a, stream);
            for (int r = 0; r < 3; r++) go();
            hipEventRecord(b, stream);
            hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            return ms / 3;
        };
        float best = timeit([&]() { launch_dwconv(p, out2, stream); });
        if (getenv("BNHIP_DEBUG")) fprintf(stderr, "[bnhip] tune %-16s dwconv register-tiled: %.1f us\n", s.name.c_str(), best * 1e3);
        if (getenv("BNHIP_DW_LDS")) best = 1e30f;                   // tests: the staged form wherever it exists
        int best_idx = -1;
        for (int idx = 0; idx < expdw_num_shapes(); idx++) {
            if (!expdw_shape_fits(idx, g)) continue;
            const float ms = timeit([&]() { launch_dwconv_lds(p, out2, idx, stream); });
            if (getenv("BNHIP_DEBUG")) fprintf(stderr, "[bnhip] tune %-16s dwconv LDS shape=%d: %.1f us\n", s.name.c_str(), idx, ms * 1e3);
            if (ms < best * 0.97f) { best = ms; best_idx = idx; }
        }
        if (best_idx < 0) continue;
        s.dwl = 1; s.shape = best_idx;
        if (s.out2 >= 0) {
            s.S = expdw_shape_slabs(best_idx, g);
            for (auto& c : steps) if (&c != &s && c.in0 == s.out2) c.S = s.S;
        }
    }
    hipStreamSynchronize(stream);
    hipEventDestroy(a); hipEventDestroy(b);
    (void)hipGetLastError();
}

void Engine::autotune_pw() {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    // two tunings: the batch one lane launches with (+1 % over tuning at max_batch), and max_batch for calls that run unsplit
    const int n_lane = (max_batch + n_lanes - 1) / n_lanes;
    for (int pass = 0; pass < 2; pass++) {
        const int n = pass == 0 ? n_lane : max_batch;
        if (pass == 1 && n == n_lane) { for (auto& s : steps) { s.nt_full = s.nt; s.wm_full = s.wm; } break; }
        for (auto& s : steps) {
            if (s.kind != S_PW || (s.C & 3)) continue;
            float* in0 = vptr(s.in0, d_stage_in, d_stage_logits, nullptr);
            float* in1 = vptr(s.in1, d_stage_in, d_stage_logits, nullptr);
            float* in2 = vptr(s.in2, d_stage_in, d_stage_logits, nullptr);
            float* out = vptr(s.out, d_stage_in, d_stage_logits, nullptr);
            float best = 1e30f; int best_nt = 0, best_wm = 0;
            double best_work = 1e300;
            // wm 2, 1: k_pw_gemm with 128- / 64-row tiles; 4, 3: the same tiles on the software-pipelined k_pw_pipe.
            // Selection: a serial engine takes the fastest candidate.  A pipelined one (depth > 1) is issue-bound - the other
            // context fills every stall - so there the candidate with the least padded MFMA work wins and time only breaks
            // ties: N = 80 / 112 take exact 80- / 112-column tiles although 48- / 64-column ones are 10-45 % faster alone
            // (measured: +1.2 % on the pipelined bench).
            // The full-batch tuning (pass 1) is what a context runs, and the host-pointer pipeline (host_depth > 1) runs its
            // chunks on contexts too; the lane tuning (pass 0) serves serial calls of a depth-1 engine and stays by time.
            const bool by_work = (depth > 1 || (pass == 1 && host_depth > 1)) && !getenv("BNHIP_TUNE_BY_TIME");
            const bool arith_bx = s.bx && s.wbx && bf16x3;        // the planner's shape rule: which kernel family computes this layer
            for (int wm = arith_bx ? 0 : 4; wm >= 1; wm--) {
                for (int nt = 1; nt <= 8; nt++) {
                    const long M_ = (long)n * s.H * s.W, bm = (wm == 1 || wm == 3) ? 64 : 128;
                    long cols = (long)((s.Co + nt * 16 - 1) / (nt * 16)) * nt * 16;
                    if (cols * 100 > (long)((s.Co + 15) / 16 * 16) * 130) continue;         // skip absurd padding
                    if (wm > 2 && !pw_pipe_ok(nt, wm - 2, s.C)) continue;
                    if (nt > 4 && !by_work) continue;                                       // wide tiles never win alone (2 waves/SIMD)
                    const double work = (double)((M_ + bm - 1) / bm * bm) * (double)cols;
                    if (by_work && work > best_work * 1.01) continue;                       // cannot win: skip the timing
                    PwParams p{in0, s.w0, s.w1, in1, in2, out, n * s.H * s.W, s.Co, s.C, s.H * s.W, s.act, nt, wm};
                    launch_pw_gemm(p, stream);                                             // warm-up
                    hipEventRecord(a, stream);
                    for (int r = 0; r < 3; r++) launch_pw_gemm(p, stream);
                    hipEventRecord(b, stream);
                    hipEventSynchronize(b);
                    float ms = 0; hipEventElapsedTime(&ms, a, b);
                    if (getenv("BNHIP_DEBUG")) fprintf(stderr, "[bnhip] tune %-16s n=%d M=%d N=%d K=%d nt=%d wm=%d: %.1f us (%.1f TF)\n", s.name.c_str(), n, n * s.H * s.W, s.Co, s.C, nt, wm, ms / 3 * 1e3, 2.0 * n * s.H * s.W * s.Co * s.C / (ms / 3 * 1e-3) / 1e12);
                    const bool less_work = by_work && work < best_work * 0.99;
                    if (less_work || ms < best * 0.98f) { best = ms; best_nt = nt; best_wm = wm; best_work = std::min(best_work, work); }
                }
            }
            if (arith_bx) {
                // split-bf16 candidates (wm 5 / 6 = 64- / 128-row tiles, 7 / 8 the software-pipelined form): the fastest tile
                float bbest = 1e30f; int bnt = 0, bwm = 0;
                for (int wm = 12; wm >= 5; wm--)
                    for (int nt = 1; nt <= 8; nt++) {
                        if (wm == 11 && precision != 1) continue;
                        long cols = (long)((s.Co + nt * 16 - 1) / (nt * 16)) * nt * 16;
                        if (wm != 12 && cols * 100 > (long)((s.Co + 15) / 16 * 16) * 130) continue;
                        if (wm >= 7 && wm <= 8 && !pw_bx3p_ok(nt, wm - 6, s.C)) continue;
                        if ((wm == 9 || wm == 10) && !pw_b16_ok(precision, s.C, pw_sw)) continue;
                        if (wm == 10 && precision != 0) continue;              // (64-row tiles of k_pw_b16: six-product form only)
                        PwParams p{in0, s.w0, s.w1, in1, in2, out, n * s.H * s.W, s.Co, s.C, s.H * s.W, s.act, nt, wm};
                        p.prec = precision; p.sw = pw_sw;
                        p.a_bf16 = vals[s.in0].half ? 1 : 0; p.out_bf16 = vals[s.out].half ? 1 : 0;      // the flavour the calls will run
                        if (wm == 11) {                                        // weights-stationary form: no tile to choose
                            if (nt != 1 || !pw_b16s_ok(p)) continue;
                        }
                        if (wm == 12) {                                        // weight columns in LDS: 64-, 96- or 128-column blocks
                            // (round 6, VERDICT r5: only where a call owns the GPU - a serial engine's lane tuning.  In a pipelined
                            // engine a 512-thread block that holds a CU's LDS and registers shuts the other context out for its
                            // duration: its launches stretched 39 -> 113 us in the timed trace for +-0 on the bench.)
                            if (by_work) continue;
                            if ((nt != 4 && nt != 6 && nt != 8) || !pw_ws_ok(p) || !pw_ws_fills(p)) continue;
                            if (precision == 1 && nt != 4) continue;
                        }
                        p.res_bf16 = (s.in2 >= 0 && vals[s.in2].half) ? 1 : 0;
                        launch_pw_bx3(p, s.wbx, stream);
                        hipEventRecord(a, stream);
                        for (int r = 0; r < 3; r++) launch_pw_bx3(p, s.wbx, stream);
                        hipEventRecord(b, stream);
                        hipEventSynchronize(b);
                        float ms = 0; hipEventElapsedTime(&ms, a, b);
                        if (getenv("BNHIP_DEBUG")) fprintf(stderr, "[bnhip] tune %-16s n=%d M=%d N=%d K=%d nt=%d wm=%d (bf16x3): %.1f us (%.1f TF fp32-equivalent)\n", s.name.c_str(), n, n * s.H * s.W, s.Co, s.C, nt, wm, ms / 3 * 1e3, 2.0 * n * s.H * s.W * s.Co * s.C / (ms / 3 * 1e-3) / 1e12);
                        if (ms < bbest * 0.98f) { bbest = ms; bnt = nt; bwm = wm; }
                    }
                if (bnt) { best_nt = bnt; best_wm = bwm; } else { best_nt = 0; best_wm = 6; }
            }
            if (pass == 0) { s.nt = best_nt; s.wm = best_wm; } else { s.nt_full = best_nt; s.wm_full = best_wm; }
        }
    }
    hipStreamSynchronize(stream);
    hipEventDestroy(a); hipEventDestroy(b);
    (void)hipGetLastError();
}

// ================================================================================================ run
float* Engine::vptr(int v, const float* d_in, float* d_logits, float* d_emb, int lane) const {
    if (v < 0) return nullptr;
    if (v == v_input) return const_cast<float*>(d_in);
    if (v == v_logits) return d_logits;
    if (part_hand && v == v_hand) return part_hand;
    (void)d_emb;
    char* base = cur_arena ? cur_arena : act_arena;
    // lane >= 0: that lane's own copy of the lane layout (values are [clip][elems] from the lane's first clip)
    if (lane >= 0) return reinterpret_cast<float*>(base + (size_t)lane * lane_bytes + vals[v].offset_lane);
    return reinterpret_cast<float*>(base + vals[v].offset);
}

hipEvent_t Engine::get_event() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}

void Engine::drop_graphs() {
    for (auto& g : graphs) if (g.exec) hipGraphExecDestroy(g.exec);
    graphs.clear();
}

// The plan is a fixed sequence of ~60 launches; for the product's call pattern (one clip per Predict) the forward pass
// is launch-bound, so once the same (input, output, n) combination shows up a second time it is captured into a
// hipGraph and replayed from then on (the host-pointer entry points always use the same staging buffers).  Per-launch
// profiling needs real launches and bypasses the graph.
// Call i of a pipelined sequence: wait (on the GPU) for whatever the caller's stream has queued so far, then run the whole
// plan on context i % depth.  Calls on different contexts overlap; calls on the same context are ordered by its stream.
bool Engine::run_pipelined(const float* d_in, int n, float* d_logits, float* d_emb, std::string* err) {
    if (n <= 0 || n > max_batch) { *err = "batch size out of range"; return false; }
    if (depth <= 1 || (profiling && profile_filter.empty())) return run(d_in, n, d_logits, d_emb, err);
    const int c = (int)(call_idx++ % (unsigned)depth);
    if (ctx_stream[c] != stream) {              // (context 0 IS the main stream until the caller hands in its own)
        hipEventRecord(ev_ctx_fork, stream);
        hipStreamWaitEvent(ctx_stream[c], ev_ctx_fork, 0);
    }
    cur_arena = ctx_arena[c];
    cur_stream = ctx_stream[c];
    cur_ctx = c;
    bool ok = run_eager(d_in, n, d_logits, d_emb, err);
    cur_arena = nullptr;
    cur_stream = nullptr;
    cur_ctx = -1;
    return ok;
}
// One chunk of a host-pointer call in context c's arena on stream st (hostpipe.cpp orders st behind the chunk's copy and
// records its completion): the whole plan, unsplit.
bool Engine::run_on_context(int c, hipStream_t st, const float* d_in, int n, float* d_logits, float* d_emb, std::string* err) {
    if (n <= 0 || n > max_batch) { *err = "batch size out of range"; return false; }
    if (c < 0 || c >= kMaxDepth || !st || !ctx_arena[c]) { *err = "context does not exist"; return false; }
    call_idx++;                                  // a later unsplit run() orders itself behind the contexts
    cur_arena = ctx_arena[c];
    cur_stream = st;
    cur_ctx = c;                                 // (the context's own min/max scratch)
    bool ok = run_eager(d_in, n, d_logits, d_emb, err);
    cur_ctx = -1;
    cur_arena = nullptr;
    cur_stream = nullptr;
    return ok;
}
// Steps [s0, s1) for n clips in context c's arena on stream st (hostpipe.cpp's two-phase calls); the one value that crosses the cut
// is read / written at `hand` instead of its arena slot.
bool Engine::run_part(int c, hipStream_t st, int s0, int s1, const float* d_in, int n, float* hand, float* d_logits, float* d_emb, std::string* err) {
    if (n <= 0 || n > max_batch) { *err = "batch size out of range"; return false; }
    if (c < 0 || c >= kMaxDepth || !st || !ctx_arena[c]) { *err = "context does not exist"; return false; }
    if (v_hand < 0 || !hand || s0 < 0 || s1 > (int)steps.size() || s0 >= s1 || (s0 != 0 && s0 != split_step) || (s1 != (int)steps.size() && s1 != split_step)) {
        *err = "not a cut of this plan"; return false;
    }
    call_idx++;
    cur_arena = ctx_arena[c];
    cur_stream = st;
    cur_ctx = c;
    part_s0 = s0; part_s1 = s1; part_hand = hand;
    bool ok = run_eager(d_in, n, d_logits, d_emb, err);
    part_s0 = 0; part_s1 = -1; part_hand = nullptr;
    cur_ctx = -1;
    cur_arena = nullptr;
    cur_stream = nullptr;
    return ok;
}
bool Engine::ensure_hand(std::string* err) {
    if (d_hand || v_hand < 0) return v_hand >= 0;
    if (hipMalloc((void**)&d_hand, std::max<size_t>((size_t)max_batch * hand_clip_bytes(), 256)) != hipSuccess) {
        (void)hipGetLastError(); d_hand = nullptr; *err = "hand-off allocation failed"; return false;
    }
    return true;
}
// Where a call that fits one batch is cut in two (run_part): the launch boundaries a single activation crosses are the block
// boundaries of the stack; of those, the first one whose crossing value has fewer than `rows_min` spatial positions per clip - from
// there on a quarter-batch chunk has too few rows to fill 256 CUs and the layers are worth running over a larger group (v2.4 stack:
// the input of b5; cuts from b4 to b7 measured within 2 % of each other, earlier and later ones lose).
// BNHIP_HOST_SPLIT=<step index> picks a boundary by hand (snapped to the next candidate), -1 disables; read once here, at plan time.
void Engine::pick_split() {
    split_step = -1; v_hand = -1; split_candidates.clear();
    const int ns = (int)steps.size();
    std::vector<int> crossing(ns, -1);
    for (int k = 1; k < ns; k++) {
        int cnt = 0, who = -1; bool bad = false;
        for (int v = 0; v < (int)vals.size(); v++) {
            const Value& V = vals[v];
            if (V.first < 0 || V.first >= k || V.last < k) continue;      // not alive across the boundary before step k
            if (v == v_input || v == v_logits || v == v_emb || V.external) { bad = true; break; }
            cnt++; who = v;
        }
        if (!bad && cnt == 1) { split_candidates.push_back(k); crossing[k] = who; }
    }
    int want = -2;
    if (const char* e = getenv("BNHIP_HOST_SPLIT")) want = atoi(e);
    if (want == -1 || split_candidates.empty()) return;
    int pick = -1;
    if (want >= 0) {
        for (int k : split_candidates) if (k >= want) { pick = k; break; }
    } else {
        const int rows_min = (int)((long)(device >= 0 ? device_cus() : 256) * 1000 / 256);      // (1000 on the 256 CUs of an MI355X)
        for (int k : split_candidates) {
            const Step& s = steps[k];                                    // the consumer's input geometry = the crossing value's
            if ((s.kind == S_PW || s.kind == S_EXPAND_DW || s.kind == S_DW) && s.H * s.W < rows_min && s.H * s.W > 0) { pick = k; break; }
        }
    }
    if (pick < 0) return;
    split_step = pick; v_hand = crossing[pick];
    if (getenv("BNHIP_DEBUG")) {
        fprintf(stderr, "[bnhip] host split: step %d (%s), value %d (%zu elems/clip%s); candidates:", split_step, steps[split_step].name.c_str(), v_hand,
                vals[v_hand].elems, vals[v_hand].half ? ", bf16" : "");
        for (int k : split_candidates) fprintf(stderr, " %d", k);
        fprintf(stderr, "\n");
    }
}
void Engine::sync_contexts() {
    for (int i = 0; i < kMaxKStreams; i++) if (kstream[i]) hipStreamSynchronize(kstream[i]);
}

bool Engine::run(const float* d_in, int n, float* d_logits, float* d_emb, std::string* err) {
    if (n <= 0 || n > max_batch) { *err = "batch size out of range"; return false; }
    if (call_idx) {
        // an unsplit call after pipelined ones: order it (on the GPU) behind whatever the contexts still have queued -
        // context 0 shares this call's arena
        for (int c = 0; c < kMaxDepth; c++)
            if (ctx_stream[c] && ctx_stream[c] != stream) { hipEventRecord(ev_ctx_done[c], ctx_stream[c]); hipStreamWaitEvent(stream, ev_ctx_done[c], 0); }
        if (kstream[0] && kstream[0] != stream) hipStreamSynchronize(kstream[0]);      // (caller-owned main stream: the engine's own may still hold a chunk)
    }
    if (!use_graphs || profiling) return run_eager(d_in, n, d_logits, d_emb, err);
    GraphEntry* ge = nullptr;
    for (auto& g : graphs)
        if (g.in == d_in && g.logits == d_logits && g.emb == d_emb && g.n == n) { ge = &g; break; }
    if (ge && ge->exec) {
        hipError_t e = hipGraphLaunch(ge->exec, stream);
        if (e != hipSuccess) { *err = std::string("hipGraphLaunch: ") + hipGetErrorString(e); return false; }
        return true;
    }
    if (!ge) {                                   // first sighting: run eagerly (also primes one-time function attributes)
        if (graphs.size() >= 8) { if (graphs.front().exec) hipGraphExecDestroy(graphs.front().exec); graphs.erase(graphs.begin()); }
        graphs.push_back(GraphEntry{d_in, d_logits, d_emb, n, 1, nullptr});
        return run_eager(d_in, n, d_logits, d_emb, err);
    }
    // second sighting: capture, instantiate, replay
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { (void)hipGetLastError(); use_graphs = false; return run_eager(d_in, n, d_logits, d_emb, err); }
    bool ok = run_eager(d_in, n, d_logits, d_emb, err);
    e = hipStreamEndCapture(stream, &graph);
    if (!ok || e != hipSuccess || !graph) {
        if (graph) hipGraphDestroy(graph);
        (void)hipGetLastError();
        use_graphs = false;                      // capture is not available on this stream: stay eager
        return run_eager(d_in, n, d_logits, d_emb, err);
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess || !exec) { (void)hipGetLastError(); use_graphs = false; return run_eager(d_in, n, d_logits, d_emb, err); }
    ge->exec = exec;
    e = hipGraphLaunch(exec, stream);
    if (e != hipSuccess) { *err = std::string("hipGraphLaunch: ") + hipGetErrorString(e); return false; }
    return true;
}

// Two lanes: the batch is split into two clip ranges that run the same plan on two streams (forked/joined with
// events around the call).  Every value is laid out [clip][elems], so a lane is just a clip offset into the same
// buffers.  The lanes fill each other's gaps: the short latency-bound launches (squeeze-excite, late layers with few
// workgroups) and every kernel's tail wave overlap the other lane's work.  Launches are interleaved step by step so
// neither lane waits for the other's enqueue.
bool Engine::run_eager(const float* d_in_all, int n_all, float* d_logits_all, float* d_emb_all, std::string* err) {
    struct Lane { const float* d_in; int n; float* d_logits; float* d_emb; hipStream_t st; int clip0; };
    Lane lanes[kMaxLanes];
    int nl = 1;
    hipStream_t main_stream = cur_stream ? cur_stream : stream;
    if (mm_dirty) {
        //  a call on this engine failed: a launch of k_clip_minmax_parts that did not run to completion leaves its arrival
        // counter non-zero, and every later small call on that (context, lane) would normalise with stale min / max without any
        // error.  Error path only: drain the engine's streams and zero the scratch before anything else is queued.
        for (int i = 0; i < kMaxKStreams; i++) if (kstream[i]) hipStreamSynchronize(kstream[i]);
        if (stream) hipStreamSynchronize(stream);
        if (mm_scratch) hipMemset(mm_scratch, 0, (size_t)kMaxDepth * kMaxLanes * kMinMaxScratch * sizeof(float));
        (void)hipGetLastError();
        mm_dirty = false;
    }
    // (an unfiltered profile wants clean per-kernel times and stays single-lane; a class-filtered one measures the
    // kernels as they run in production, overlapped)
    if (n_lanes > 1 && !cur_stream && (!profiling || !profile_filter.empty()) && n_all >= dual_lane_min) nl = std::min(n_lanes, n_all);
    for (int li = 0, c0 = 0; li < nl; li++) {
        int cnt = n_all / nl + (li < n_all % nl ? 1 : 0);
        lanes[li] = Lane{d_in_all + (size_t)c0 * n_samples, cnt, d_logits_all + (size_t)c0 * n_classes,
                         d_emb_all ? d_emb_all + (size_t)c0 * emb_dim : nullptr, li == 0 ? main_stream : lane_stream[li - 1], c0};
        c0 += cnt;
    }
    hipEvent_t st_a = nullptr, st_b = nullptr;
    if (step_timing) { hipEventCreate(&st_a); hipEventCreate(&st_b); hipEventRecord(st_a, main_stream); }
    if (nl > 1) {
        hipEventRecord(ev_fork, main_stream);
        for (int li = 1; li < nl; li++) hipStreamWaitEvent(lanes[li].st, ev_fork, 0);
    }
    const int s_end = part_s1 < 0 ? (int)steps.size() : part_s1;
    for (int si = part_s0; si < s_end; si++) {
      for (int li = 0; li < nl; li++) {
        const Step& s = steps[si];
        const float* d_in = lanes[li].d_in; float* d_logits = lanes[li].d_logits; float* d_emb = lanes[li].d_emb;
        const int n = lanes[li].n, clip0 = nl > 1 ? li : -1;      // arena addressing: lane index, or -1 for the unsplit layout
        hipStream_t stream = lanes[li].st;
        float* in0 = vptr(s.in0, d_in, d_logits, d_emb, clip0);
        float* in1 = vptr(s.in1, d_in, d_logits, d_emb, clip0);
        float* in2 = vptr(s.in2, d_in, d_logits, d_emb, clip0);
        float* out = vptr(s.out, d_in, d_logits, d_emb, clip0);
        float* out2 = vptr(s.out2, d_in, d_logits, d_emb, clip0);
        ProfEntry pe{};
        // diagnostics (tools/debug): a host-side synchronize before / after every launch of one kernel class
        static const char* dbg_sync_before = getenv("BNHIP_DEBUG_SYNC_BEFORE");
        static const char* dbg_sync_after = getenv("BNHIP_DEBUG_SYNC_AFTER");
        // ... and a co-runner filter: context 1 launches only the listed kernel classes (its outputs are then garbage)
        static const char* dbg_ctx1_only = getenv("BNHIP_DEBUG_CTX1_ONLY");
        if (dbg_ctx1_only && cur_ctx == 1 && !strstr(dbg_ctx1_only, s.kclass)) continue;
        static const char* dbg_ctx1_skip = getenv("BNHIP_DEBUG_CTX1_SKIP");
        if (dbg_ctx1_skip && cur_ctx == 1 && strstr(dbg_ctx1_skip, s.kclass)) continue;
        if (dbg_sync_before && (!strcmp(dbg_sync_before, s.kclass) || s.name.rfind(dbg_sync_before, 0) == 0)) hipStreamSynchronize(stream);
        const bool prof_this = profiling && (profile_filter.empty() || profile_filter == s.kclass);
        if (prof_this) { pe.a = get_event(); pe.b = get_event(); pe.step = si; pe.n = n; hipEventRecord(pe.a, stream); }
        switch (s.kind) {
            case S_MINMAX:
                // (scratch of the small-call form: a buffer of its own per context and lane - the arena's full and lane layouts
                // share memory, a counter kept there would be overwritten by the other layout's activations between calls)
                launch_clip_minmax(in0, n, n_samples, specs[0].eps, reinterpret_cast<float2*>(out),
                                   mm_scratch ? mm_scratch + ((size_t)std::max(cur_ctx, 0) * kMaxLanes + (size_t)li) * kMinMaxScratch : nullptr, stream);
                break;
            case S_FRONTEND: {
                const FrontSpec& fs = specs[s.spec];
                FrontendParams p;
                p.x = in0; p.mm = reinterpret_cast<const float2*>(in1); p.G = fs.G; p.window = fs.window; p.out = out;
                p.n_samples = n_samples; p.L = fs.L; p.Lfft = fs.Lfft; p.Kp = fs.Kp; p.hop = fs.hop; p.F = fs.F; p.n_mels = fs.n_mels;
                p.NTP = fs.NTP; p.C = C_spec; p.c = fs.c; p.norm_sub = fs.norm_sub; p.norm_mul = fs.norm_mul;
                p.p1 = fs.p1; p.p2 = fs.p2; p.n_clips = n;
                launch_frontend(p, stream);
                break;
            }
            case S_NORMALIZE:
                launch_normalize(in0, reinterpret_cast<const float2*>(in1), out, n, n_samples, specs[0].norm_sub, specs[0].norm_mul, stream);
                break;
            case S_STFT: {
                const FrontSpec& fs = specs[s.spec];
                StftParams p{in0, fs.window_full, fs.bins, out, n_samples, fs.Lfft, fs.L, fs.hop, fs.F, fs.nb, fs.nbp, fs.mode, n, fs.stft_tw, fs.pad_left};
                p.zmask = fs.stft_zmask;
                if (s.mode == 1) {                           // fused mel epilogue: `out` is the spectrogram image
                    p.out = nullptr; p.img = out; p.mel = s.w3; p.mel_quads = s.S; p.n_mels = fs.n_mels; p.Ctot = C_spec; p.c0 = fs.c;
                    p.logc = fs.log_compress ? 1 : 0; p.time_major = fs.time_major ? 1 : 0; p.p1 = fs.p1; p.p2 = fs.p2;
                    p.lfloor = fs.log_floor; p.lscale = fs.log_scale;
                }
                {
                    static const bool f32_off = getenv("BNHIP_STFT_F32") && atoi(getenv("BNHIP_STFT_F32")) == 0;
                    // fp32 transform: bf16 engines whose front-end compresses with a floored log (Perch-style) - a power-law
                    // compression (v2.4: x^0.45) amplifies fp32 transform noise in near-empty bins by orders of magnitude
                    if (precision == 1 && s.mode != 1 && fs.log_compress && !f32_off) p.f32 = 1;
                }
                launch_stft_bins(p, stream);
                break;
            }
            case S_MELBAND: {
                const FrontSpec& fa = specs[s.spec];
                const FrontSpec& fb = specs[s.S == 2 ? s.op : s.spec];
                MelBandParams p{{in0, s.S == 2 ? in1 : in0}, {s.w0, s.S == 2 ? s.w2 : s.w0},
                                {reinterpret_cast<const int*>(s.w1), reinterpret_cast<const int*>(s.S == 2 ? s.w3 : s.w1)},
                                {fa.nbp, fb.nbp}, {fa.p1, fb.p1}, {fa.p2, fb.p2}, out, fa.F, fa.n_mels, C_spec, fa.c,
                                fa.log_compress ? 1 : 0, fa.log_floor, fa.log_scale, fa.time_major ? 1 : 0};
                launch_mel_banded(p, s.S, n, stream);
                break;
            }
            case S_MELFIN: {
                const FrontSpec& fa = specs[s.spec];
                const FrontSpec& fb = specs[s.S == 2 ? s.op : s.spec];
                MelFinParams p{{in0, s.S == 2 ? in1 : in0}, {fa.p1, fb.p1}, {fa.p2, fb.p2}, out, fa.F, fa.n_mels, fa.n_mels, C_spec, fa.c,
                               fa.log_compress ? 1 : 0, fa.log_floor, fa.log_scale, fa.time_major ? 1 : 0};
                launch_mel_finish(p, s.S, n, stream);
                break;
            }
            case S_CONV_DIRECT: {
                ConvParams p{in0, s.w0, s.w1, out, n, s.H, s.W, s.C, s.Ho, s.Wo, s.Co, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.act};
                p.out_bf16 = vals[s.out].half ? 1 : 0;
                if (s.w2) launch_stem_mfma(p, s.w2, s.w3, stream);
                else launch_conv_direct(p, stream);
                break;
            }
            case S_CONV_IGEMM:
                launch_conv_igemm(in0, s.w0, s.w1, out, n, s.H, s.W, s.C, s.Ho, s.Wo, s.Co, s.kh, s.kw, s.sh, s.sw, s.g.dh, s.g.dw, s.pt, s.pl,
                                  s.act, 0, 0, stream);
                break;
            case S_PW: {
                PwParams p{in0, s.w0, s.w1, in1, in2, out, n * s.H * s.W, s.Co, s.C, s.H * s.W, s.act, nl > 1 ? s.nt : s.nt_full,
                           nl > 1 ? s.wm : s.wm_full};
                p.prec = precision;
                p.a_bf16 = vals[s.in0].half ? 1 : 0; p.out_bf16 = vals[s.out].half ? 1 : 0;
                p.res_bf16 = (s.in2 >= 0 && vals[s.in2].half) ? 1 : 0;
                p.sw = pw_sw;
                if (p.wm >= 5 && s.wbx) launch_pw_bx3(p, s.wbx, stream);
                else launch_pw_gemm(p, stream);
                break;
            }
            case S_DW: {
                DwParams p{in0, s.w0, s.w1, out, n, s.H, s.W, s.C, s.Ho, s.Wo, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.act};
                p.in_bf16 = vals[s.in0].half ? 1 : 0; p.out_bf16 = vals[s.out].half ? 1 : 0;
                if (s.dwl) launch_dwconv_lds(p, out2, s.shape, stream);
                else launch_dwconv(p, out2, stream);
                break;
            }
            case S_EXPAND_DW:
                {
                    StemGeom sg{s.H2, s.W2, s.pt2, s.pl2};
                    launch_expand_dw(in0, s.w0, s.w1, s.w2, s.w3, out, out2, n, s.H, s.W, s.C, s.Co,
                                     s.Ho, s.Wo, s.kh, s.sh, s.pt, s.pl, s.act, s.act2, s.shape, s.mode == 1 ? &sg : nullptr, stream,
                                     s.bx ? s.wbx : nullptr, precision, vals[s.out].half ? 1 : 0, vals[s.in0].half ? 1 : 0);
                }
                break;
            case S_MEAN_PARTIAL:
                launch_mean_partial(in0, out, n, s.H * s.W, s.C, s.S, stream);
                break;
            case S_MEAN_FINISH:
                launch_mean_finish(in0, out, n, s.H * s.W, s.C, s.S, stream);
                break;
            case S_SE: {
                SeParams p{in0, s.S, s.H * s.W, s.w0, s.w1, s.w2, s.w3, out, n, s.C, s.Cr, s.act, s.act2};
                if (cur_stream) p.threads = 256;      // pipelined call: 4-wave workgroups slot in beside the other context's kernels
                launch_se(p, stream);
                break;
            }
            case S_UNARY:
                launch_unary(in0, out, vals[s.in0].elems * (size_t)n, s.act, stream);
                break;
            case S_BINARY:
                launch_binary(in0, s.mode == 2 ? s.w0 : in1, out, vals[s.in0].elems * (size_t)n, s.op, s.mode, s.H * s.W, s.C,
                              s.act, stream);
                break;
            case S_EW_UNARY:
                launch_unary_op(in0, out, vals[s.in0].elems * (size_t)n, s.op, s.g.alpha, stream);
                break;
            case S_EW_BINARY: {
                BcastParams p;
                p.a = s.g.a_const ? s.w0 : in0; p.b = s.g.b_const ? s.w1 : in1; p.out = out;
                for (int k = 0; k < 4; k++) { p.d[k] = s.g.d[k]; p.sa[k] = s.g.sa[k]; p.sb[k] = s.g.sb[k]; }
                p.bsa = s.g.a_const ? 0 : (long)vals[s.in0].elems; p.bsb = s.g.b_const ? 0 : (long)vals[s.in1].elems;
                p.op = s.op; p.act = s.act;
                launch_binary_bcast(p, n, stream);
                break;
            }
            case S_POOL: {
                PoolParams p{in0, out, n, s.H, s.W, s.C, s.Ho, s.Wo, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.mode, s.act};
                launch_pool2d(p, stream);
                break;
            }
            case S_COPY: {
                if (s.g.fill) launch_fill(out, vals[s.out].elems * (size_t)n, s.g.alpha, stream);
                CopyParams p;
                p.in = s.g.a_const ? s.w0 : in0; p.out = out;
                for (int k = 0; k < 4; k++) { p.d[k] = s.g.d[k]; p.si[k] = s.g.sa[k]; p.so[k] = s.g.so[k]; }
                p.offi = s.g.offa; p.offo = s.g.offo;
                p.bsi = s.g.a_const ? 0 : (long)vals[s.in0].elems; p.bso = (long)vals[s.out].elems;
                launch_copy_view(p, n, stream);
                break;
            }
            case S_SOFTMAX:
                launch_softmax_rows(in0, out, (size_t)s.H * (size_t)n, s.C, s.g.alpha, stream);
                break;
            case S_REDUCE: {
                ReduceParams p{in0, out, {s.g.d[0], s.g.d[1], s.g.d[2], s.g.d[3]}, s.g.mask, s.op};
                launch_reduce(p, n, stream);
                break;
            }
            case S_CONV_GENERIC: {
                GenConvParams p{in0, s.w0, s.w1, out, n, s.H, s.W, s.C, s.Ho, s.Wo, s.Co, s.kh, s.kw, s.sh, s.sw, s.g.dh, s.g.dw,
                                s.pt, s.pl, s.act, s.g.depthwise, s.g.mult};
                launch_conv_generic(p, stream);
                break;
            }
        }
        if (prof_this) { hipEventRecord(pe.b, stream); prof.push_back(pe); }
        if (dbg_sync_after && (!strcmp(dbg_sync_after, s.kclass) || s.name.rfind(dbg_sync_after, 0) == 0)) hipStreamSynchronize(stream);
      }
    }
    for (int li = 0; li < nl; li++) {
        const Lane& L = lanes[li];
        if (L.d_emb && v_emb >= 0 && s_end == (int)steps.size()) {
            hipError_t e = hipMemcpyAsync(L.d_emb, vptr(v_emb, L.d_in, L.d_logits, L.d_emb, nl > 1 ? li : -1), (size_t)L.n * emb_dim * 4,
                                          hipMemcpyDeviceToDevice, L.st);
            if (e != hipSuccess) { *err = std::string("emb copy: ") + hipGetErrorString(e); mm_dirty = true; return false; }
        }
    }
    for (int li = 1; li < nl; li++) {
        hipEventRecord(ev_join[li - 1], lanes[li].st);
        hipStreamWaitEvent(main_stream, ev_join[li - 1], 0);
    }
    if (step_timing) { hipEventRecord(st_b, main_stream); step_ev.emplace_back(st_a, st_b); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *err = std::string("kernel launch: ") + hipGetErrorString(e); mm_dirty = true; return false; }
    return true;
}

int Engine::steps_read(double* start_ms, double* end_ms, int cap) {
    sync_contexts();
    if (stream) hipStreamSynchronize(stream);
    const int n = (int)step_ev.size();
    for (int i = 0; i < n; i++) {
        float a = 0, b = 0;
        hipEventElapsedTime(&a, step_ev[0].first, step_ev[i].first);
        hipEventElapsedTime(&b, step_ev[0].first, step_ev[i].second);
        if (i < cap) { if (start_ms) start_ms[i] = a; if (end_ms) end_ms[i] = b; }
    }
    for (auto& p : step_ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    step_ev.clear();
    (void)hipGetLastError();
    return n;
}

// ================================================================================================ describe / profile
static void jesc(std::ostringstream& os, const std::string& s) {
    for (char c : s) { if (c == '"' || c == '\\') os << '\\'; os << c; }
}

std::string Engine::describe() const {
    std::ostringstream os;
    os << "{\"tune_source\":\"";
    jesc(os, tune_source);
    os << "\",\"tune_key\":\"" << (device >= 0 ? tune_key() : std::string()) << "\",\"split_step\":" << split_step << ",\"n_samples\":" << n_samples << ",\"n_classes\":" << n_classes << ",\"emb_dim\":" << emb_dim
       << ",\"max_batch\":" << max_batch << ",\"logits_output\":" << logits_output << ",\"embedding_output\":" << embedding_output << ",\"precision\":\"" << (precision ? "bf16" : "f32") << "\",\"lanes\":" << n_lanes << ",\"lane_min_batch\":" << dual_lane_min << ",\"act_arena_bytes\":" << act_bytes << ",\"weight_bytes\":" << w_bytes
       << ",\"specs\":[";
    for (size_t i = 0; i < specs.size(); i++) {
        const FrontSpec& f = specs[i];
        os << (i ? "," : "") << "{\"frame_length\":" << f.L << ",\"fft_length\":" << f.Lfft << ",\"hop\":" << f.hop
           << ",\"frames\":" << f.F << ",\"n_mels\":" << f.n_mels << ",\"channel\":" << f.c << ",\"p1\":" << f.p1
           << ",\"p2\":" << f.p2 << "}";
    }
    os << "],\"steps\":[";
    for (size_t i = 0; i < steps.size(); i++) {
        const Step& s = steps[i];
        os << (i ? "," : "") << "{\"i\":" << i << ",\"kernel\":\"" << s.kclass << "\",\"name\":\"";
        jesc(os, s.name);
        os << "\",\"H\":" << s.H << ",\"W\":" << s.W << ",\"C\":" << s.C << ",\"Co\":" << s.Co << ",\"k\":" << s.kh
           << ",\"stride\":" << s.sh << ",\"act\":" << s.act << ",\"fused_scale\":" << (s.kind == S_PW && s.in1 >= 0 ? 1 : 0)
           << ",\"shape\":" << s.shape << ",\"dw_lds\":" << s.dwl << ",\"bx\":" << s.bx << ",\"nt\":" << s.nt << ",\"wm\":" << s.wm << ",\"nt_full\":" << s.nt_full << ",\"wm_full\":" << s.wm_full << ",\"fused_res\":" << (s.kind == S_PW && s.in2 >= 0 ? 1 : 0) << ",\"fused_sum\":" << (s.out2 >= 0 ? 1 : 0) << ",\"flops\":" << s.flops << ",\"bytes\":" << s.bytes << ",\"wbytes\":" << s.wbytes
           << ",\"out_v\":" << s.out << ",\"out2_v\":" << s.out2
           << ",\"in_bf16\":" << ((s.in0 >= 0 && vals[s.in0].half) ? 1 : 0) << ",\"out_bf16\":" << ((s.out >= 0 && vals[s.out].half) ? 1 : 0) << "}";
    }
    os << "]}";
    return os.str();
}

std::string Engine::profile_read() {
    hipStreamSynchronize(stream);
    struct Agg { double ms = 0, flops = 0, bytes = 0; long launches = 0; };
    std::map<std::string, Agg> agg;
    std::vector<std::string> order;
    std::vector<Agg> per_step(steps.size());
    for (auto& e : prof) {
        float ms = 0;
        hipEventElapsedTime(&ms, e.a, e.b);
        const Step& s = steps[e.step];
        if (!agg.count(s.kclass)) order.push_back(s.kclass);
        Agg& a = agg[s.kclass];
        a.ms += ms; a.launches++; a.flops += s.flops * e.n; a.bytes += s.bytes * e.n + s.wbytes;
        Agg& ps = per_step[e.step];
        ps.ms += ms; ps.launches++; ps.flops += s.flops * e.n; ps.bytes += s.bytes * e.n + s.wbytes;
        ev_pool.push_back(e.a); ev_pool.push_back(e.b);
    }
    prof.clear();
    std::ostringstream os;
    os << "[";
    for (size_t i = 0; i < order.size(); i++) {
        const Agg& a = agg[order[i]];
        os << (i ? "," : "") << "{\"kernel\":\"" << order[i] << "\",\"launches\":" << a.launches << ",\"ms\":" << a.ms
           << ",\"flops\":" << a.flops << ",\"bytes\":" << a.bytes << "}";
    }
    for (size_t i = 0; i < steps.size(); i++) {
        const Agg& a = per_step[i];
        if (!a.launches) continue;
        os << ",{\"step\":" << i << ",\"kernel\":\"" << steps[i].kclass << "\",\"name\":\"";
        jesc(os, steps[i].name);
        os << "\",\"launches\":" << a.launches << ",\"ms\":" << a.ms << ",\"flops\":" << a.flops << ",\"bytes\":" << a.bytes << "}";
    }
    os << "]";
    return os.str();
}

}  // namespace bnhip
