// Launch wrappers for the gfx950 kernels in kernels.hip.  All tensors are fp32, activations NHWC.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <vector>
#include <cstdint>

namespace bnhip {

// The dynamic-LDS limit of a kernel is an attribute of the function ON THE CURRENT DEVICE (round 5: process-wide once-flags left every
// further device of a multi-device handle at the 64 KB default; setting it per launch cost a runtime call on the one-clip path,
// ADVICE r5).  Once per (function, device): one bit per device ordinal in a flag that belongs to this instantiation.
// Compute units of the CURRENT device (cached per ordinal; 256 when no device answers - plan-only engines, the MI355X figure).
// The grid-fill rules (pw_fill_grid, pw_ws_fills, the parted chunk loops of k_expand_dw_sk, Engine::pick_split) are stated in
// CUs, not as MI355X literals.
inline int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    if (dev < 0 || dev >= 64) return 256;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    hipDeviceProp_t pr{};
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) { (void)hipGetLastError(); return 256; }
    cus[dev].store(pr.multiProcessorCount, std::memory_order_relaxed);
    return pr.multiProcessorCount;
}

template <auto Kern>
inline void lds_limit_once(int bytes) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev < 0 || dev >= 64) { hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); return; }
    const unsigned long long bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess)
        done.fetch_or(bit, std::memory_order_release);
    else (void)hipGetLastError();
}

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU_N1_TO_1 = 2, ACT_RELU6 = 3, ACT_TANH = 4, ACT_SWISH = 100, ACT_SIGMOID = 101, ACT_HARD_SWISH = 102 };

// ---- ingest
void launch_pcm_to_f32(const void* pcm, int bits /*16, 24, 32*/, float* out, size_t n, hipStream_t s);

// ---- front-end
// per-clip (min, max(x-min)+eps) as TFLite's REDUCE_MIN/SUB/REDUCE_MAX/ADD chain produces them
constexpr int kMinMaxParts = 16;      // blocks per clip of the small-call form; scratch = [clip][2 * kMinMaxParts + 2] floats, zero before first use
void launch_clip_minmax(const float* x, int n_clips, int n_samples, float eps, float2* mm, float* scratch /*nullable*/, hipStream_t s);

struct FrontendParams {
    const float* x;        // [B, n_samples] raw clip
    const float2* mm;      // [B] (min, range+eps)
    const double* G;       // [Kp, NTP] folded DFT-real*mel matrix (fp64), rows n' = 0..Lfft/2 (then zero padding)
    const float* window;   // [2*Kp]: w[n'] then w[Lfft-n'] (0 where the mirror sample does not exist)
    float* out;            // [B, n_mels, F, C]
    int n_samples, L, Lfft, Kp, hop, F, n_mels, NTP, C, c;   // c = channel index written; Kp = roundup(Lfft/2+1, 32)
    float norm_sub, norm_mul;   // (x-min)/range - norm_sub) * norm_mul
    float p1, p2;          // y = pow(pow(v, p1), p2); p2 == 1 => single pow
    int n_clips;
};
void launch_frontend(const FrontendParams& p, hipStream_t s);
size_t frontend_lds_bytes(int Lfft, int Kp, int hop, int NTP);

// FFT-based front-end (stft.hip): normalise once, STFT -> needed bins (fp32), then k_pw_gemm with the mel matrix and
// k_mel_finish.  Serves both the real-part (CAST) and the magnitude (COMPLEX_ABS) graph.
struct StftParams {
    const float* xn;        // [B, n_samples] normalised clip
    const float* window;    // [L]
    const int* bins;        // [nb] ascending DFT bins the mel matrix uses
    float* out;             // [B, F, nbp]; columns nb..nbp-1 are written as zeros
    int n_samples, Lfft, L, hop, F, nb, nbp, mode /*0 real part, 1 magnitude*/, n_clips;
    const double* tw = nullptr;   // plan-time twiddle image (stft_build_tables)
    int pad_left = 0;             // frame f starts at sample f * hop - pad_left; samples outside [0, n_samples) are zero
    int nb_cap = 0, fpw = 0;    // set by the launcher
    int f32 = 0;                // 1: the transform in fp32 ("precision":"bf16" engines; not with the fused mel epilogue)
    unsigned zmask = 0xffffffffu; // output pruning of the last FFT stage (stft_zmask): which of the Q outputs of the closing Q-point transforms any
                                // needed bin reads, four bits per pass of that stage (wave-uniform); all ones = compute everything
    // fused mel epilogue (mel != nullptr): the wave that transformed a frame also applies the banded mel matrix and the
    // compression and writes the frame's n_mels values straight into the spectrogram image - the DFT bins never reach HBM.
    // mel = stft_mel_table(): [64 lanes][8] int32 {band A, first quad, quads, weight offset; band B ...} then the band
    // weights, each band's aligned quads back to back (mel_quads float4s).
    const float* mel = nullptr;
    int mel_quads = 0, n_mels = 0, Ctot = 1, c0 = 0, logc = 0, time_major = 0;
    float p1 = 1.f, p2 = 1.f, lfloor = 0.f, lscale = 1.f;
    float* img = nullptr;       // [B, n_mels, F, Ctot] (or [B, F, n_mels, Ctot] when time_major)
};
std::vector<double> stft_build_tables(int Lfft, const int* bins, int nb);
unsigned stft_zmask(int Lfft, const int* bins, int nb);     // plan time: StftParams::zmask for a needed-bin list
// Fused-mel table for launch_stft_bins: melw [n_mels][nbp] (zero outside each row's band), span [2 n_mels] = [lo, hi) per row.
// Empty when the front-end cannot take the fused form (more than 512 bins, more than 128 bands, table too large for LDS).
std::vector<float> stft_mel_table(int Lfft, const float* melw, const int* span, int n_mels, int nb, int nbp, int* quads);
void launch_normalize(const float* x, const float2* mm, float* out, int n_clips, int n_samples, float norm_sub,
                      float norm_mul, hipStream_t s);
bool stft_supported(int Lfft, int nb);
void launch_stft_bins(const StftParams& p, hipStream_t s);
struct MelFinParams {
    const float* T[2];      // [B, F, ldt] per channel handled by this launch
    float p1[2], p2[2];
    float* out;             // [B, n_mels, F, Ctot]
    int F, n_mels, ldt, Ctot, c0;   // c0 = first channel written (two channels are written as one float2 when c0 is even)
    int log = 0;            // 1: lscale * log(max(v, lfloor)) instead of the two powers (all channels of the launch)
    float lfloor = 0.f, lscale = 1.f;
    int time_major = 0;     // 1: out is [B, F, n_mels, Ctot]
};
void launch_mel_finish(const MelFinParams& p, int nch, int n_clips, hipStream_t s);
struct MelBandParams {      // banded mel filterbank + pow + NHWC store, one or two channels per launch
    const float* bins[2];   // [B, F, nbp_c] STFT values on the needed bins
    const float* w[2];      // [n_mels, nbp_c] mel rows in output order (zero outside each row's band)
    const int* span[2];     // [n_mels][2]: first / one-past-last nonzero column of each row
    int nbp[2];
    float p1[2], p2[2];
    float* out;             // [B, n_mels, F, Ctot]
    int F, n_mels, Ctot, c0;
    int log = 0;            // as in MelFinParams
    float lfloor = 0.f, lscale = 1.f;
    int time_major = 0;
};
bool mel_banded_supported(int n_mels, int nbp0, int nbp1);
void launch_mel_banded(const MelBandParams& p, int nch, int n_clips, hipStream_t s);
int frontend_kc(int Lfft, int hop, int NTP);   // K-chunk of the front-end GEMM (G rows per LDS stage): 16 or 32; Kp is padded to it

// ---- CNN
struct ConvParams {       // direct conv, small Cin (stem)
    const float* in; const float* w /*[kh][kw][Cin][Cout]*/; const float* bias; float* out;
    int B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, pt, pl, act;
    int out_bf16 = 0;     // bf16 activation storage (conv_direct_bf16_ok shapes only)
};
void launch_conv_direct(const ConvParams& p, hipStream_t s);
bool conv_direct_bf16_ok(const ConvParams& p);
// general convolution as an implicit GEMM on the f32 MFMA (weights OHWI as in the file): Cin % 4 == 0, kh * kw * Cin >= 32
bool conv_igemm_supported(int Cin, int Cout, int kh, int kw);
void launch_conv_igemm(const float* in, const float* w_ohwi, const float* bias, float* out, int B, int H, int W, int Cin, int Ho, int Wo,
                       int Cout, int kh, int kw, int sh, int sw, int dh, int dw, int pt, int pl, int act, int nt /*0 = heuristic*/,
                       int wm /*1 = 64-row tiles, else 128*/, hipStream_t s);
// MFMA stem: wm = [Cout][32] weights in (row, 4 columns, channel) order with zero pads, bias_p = [Cout] (zeros if absent)
bool stem_mfma_supported(const ConvParams& p);
void launch_stem_mfma(const ConvParams& p, const float* wm, const float* bias_p, hipStream_t s);

struct PwParams {         // pointwise conv / fully-connected as GEMM: out[M,N] = act(A[M,K] W[N,K]^T + b) (+res)
    const float* A; const float* W; const float* bias; const float* ascale; const float* res; float* out;
    int M, N, K, HW, act;
    int nt = 0;           // N-tile width in 16-column units (1..8); 0 = built-in heuristic
    int wm = 0;           // row tile: 1 = 64 rows per block, otherwise 128; 3 / 4 = the same tiles on k_pw_pipe
    int prec = 0;         // k_pw_bx3 only: 0 = six bf16 products per fp32 product (fp32-equivalent), 1 = one (plain bf16 operands,
                          // fp32 accumulate: the "precision":"bf16" engines)
    int a_bf16 = 0;       // k_pw_bx3 / k_pw_bx3p only: A holds bf16 values (bf16 activation storage, K % 4 == 0)
    int out_bf16 = 0;     // any pw kernel: out is written as bf16 (N % 4 == 0)
    int res_bf16 = 0;     // any pw kernel: res holds bf16 values (N % 4 == 0) - the residual stream of a "precision":"bf16" engine
    int sw = 0;           // PW_SW_* bits: experiment / test switches of the split-bf16 kernel family, read from the environment ONCE per
                          // engine (Engine::build) and carried here - no getenv on the launch path
};
enum { PW_SW_B16_OFF = 1, PW_SW_B16_FORCE = 2, PW_SW_B16S_OFF = 4, PW_SW_B16S_FORCE = 8, PW_SW_WS_OFF = 16, PW_SW_WS_FORCE = 32, PW_SW_LAT_OFF = 64 };
int pw_switches_from_env();       // BNHIP_PW_B16 / _B16S / _WS / _LAT (0 only): "0" = never, "2" = wherever the kernel accepts the layer (parity tests)
void launch_pw_gemm(const PwParams& p, hipStream_t s);
bool pw_pipe_ok(int nt, int wm, int K);   // PwParams::wm = 2 + wm selects the software-pipelined kernel (k_pw_pipe)
int pw_default_nt(int M, int N);
// split-bf16 variant (k_pw_bx3): the same GEMM on v_mfma_f32_16x16x32_bf16 with fp32-equivalent products (three exact bf16
// pieces per operand, six products).  Needs whole 32-wide K slabs and the plan-time weight image; PwParams::wm 5 / 6 select
// its 64- / 128-row tiles.
bool pw_bx3_ok(int K);
bool pw_bx3p_ok(int nt, int wm /*1 | 2*/, int K);     // PwParams::wm 7 / 8: software-pipelined form (k_pw_bx3p)
int pw_bx3_npad(int N);
std::vector<uint16_t> pw_bx3_image(const float* W, int N, int K);
void launch_pw_bx3(const PwParams& p, const uint16_t* Wimg, hipStream_t s);
// The same GEMM with A streamed straight from global memory two slabs ahead (pw_b16.hip; PwParams::wm = 9 / 10: 128- / 64-row
// tiles; one product per operand pair for "precision":"bf16" engines - 128-row tiles only - or the six-product fp32-equivalent
// form): called by launch_pw_bx3, which has resolved the tile (nt = 16-column units) and the grid.
bool pw_b16_ok(int prec, int K, int sw);
bool pw_b16s_ok(const PwParams& p);   // weights-stationary form for skinny layers (N <= 32, K <= 192) of one-product engines: PwParams::wm = 11
void launch_pw_b16s(const PwParams& p, const uint16_t* Wimg, int Npad, hipStream_t s);
void launch_pw_b16(const PwParams& p, const uint16_t* Wimg, int nt, int wm /*1 | 2*/, int Npad, int nblk_n, unsigned nblk, hipStream_t s);
// pw_ws.hip - k_pw_ws (PwParams::wm = 12, nt = 4 | 6 | 8: 16-column units of a column block; same image, same K order, same
// product order per accumulator: bit-identical to k_pw_bx3): short K, wide N without a squeeze-excite scale (the 6x expands, the
// layer in front of the pooling): a block's weight columns stay in LDS, its waves walk 32-row tile pairs with A streamed from
// global memory through a register ring several slabs ahead, epilogue straight from the accumulators.  A call too small for it
// (pw_ws_ok) takes a tiled kernel - same bits.
// pw_ws.hip - k_pw_lat: the long-K layers (K >= 256) of SMALL calls (one to a few clips: <= 2048 16 x 16 output tiles): one wave per
// (row tile, column group), operands from global memory into a register ring several slabs ahead, no LDS, no barrier; bit-identical to
// k_pw_bx3.  Taken by launch_pw_bx3 for every call it accepts, whatever the tuned tile.
bool pw_lat_ok(const PwParams& p);
void launch_pw_lat(const PwParams& p, const uint16_t* Wimg, int Npad, hipStream_t s);
bool pw_ws_ok(const PwParams& p);
bool pw_ws_fills(const PwParams& p);      // the call is large enough to put blocks on half of the chip
void launch_pw_ws(const PwParams& p, const uint16_t* Wimg, int Npad, hipStream_t s);

struct DwParams {
    const float* in; const float* w /*[kh][kw][C]*/; const float* bias; float* out;
    int B, H, W, C, Ho, Wo, kh, kw, sh, sw, pt, pl, act;
    int in_bf16 = 0, out_bf16 = 0;   // bf16 activation storage (tiled and LDS-staged kernels only: dwconv_sum_slabs(p) > 0 or dwl)
};
// partial (nullable): [B, dwconv_sum_slabs(p), C] per-slab channel sums of the OUTPUT (fused squeeze-excite mean);
// dwconv_sum_slabs returns 0 when the shape has no tiled kernel (then no fused sums are available).
int dwconv_sum_slabs(const DwParams& p);
void launch_dwconv(const DwParams& p, float* partial, hipStream_t s);

// fused MBConv front half: y = act_d(dwconv(act_e(x We^T + be)) + bd); partial (nullable) [B, slabs, Cmid]
bool expdw_supported(int k, int s, int Cin, int Cmid, int act_e = 0, int prec = 0);   // (Cin > 128: "precision":"bf16" engines only)
// Layer geometry for the tile-shape helpers.  Shape indices 0 .. n-1 are the instantiated tile shapes in image orientation;
// n .. 2n-1 the same shapes with the roles of rows and columns swapped (tall, narrow images - a time-major spectrogram -
// tile badly with 16- / 32-column tiles): the kernel then walks the image through pixel strides and reads the depthwise taps
// transposed.  `stem` layers (raw-image variant) only exist in image orientation.
struct ExpDwGeo { int k, s, H, W, Ho, Wo, pt, pl; bool stem = false; int skw = 0; };   // skw: expdw_skw() of the layer (0: not the small-K chunk-loop form)
int expdw_skw(int Cin, int act_e, bool stem);
bool expdw_sk_pipe16(int Cin, int act_e, bool stem, int prec, bool have_image);   // phase 1 on the bf16 pipe (then no eight-wave shapes: pass skw = 0)
int expdw_sum_slabs(const ExpDwGeo& g);   // slabs of the cost-model shape; 0 = no tile shape fits (do not fuse)
int expdw_num_shapes();                   // 2n
bool expdw_shape_fits(int idx, const ExpDwGeo& g, bool planning = true);   // planning: also honour BNHIP_EXPDW_ORIENT (the launchers pass false: no getenv per launch)
int expdw_shape_slabs(int idx, const ExpDwGeo& g);
int expdw_default_shape(const ExpDwGeo& g);
int expdw_max_slabs(const ExpDwGeo& g);
struct StemGeom { int Hin, Win, pt, pl; };   // raw image size and the stem conv's top/left padding
// parameters are the planner's padded copies: we [expdw_cp(Cmid)][expdw_kw(Cin)], be/bd [Cp], wd [k*k][Cp] (zeros beyond)
int expdw_kw(int Cin);
int expdw_cp(int Cmid);
void launch_expand_dw(const float* x, const float* we, const float* be, const float* wd, const float* bd, float* y,
                      float* partial, int B, int H, int W, int Cin, int Cmid, int Ho, int Wo, int k, int s, int pt,
                      int pl, int act_e, int act_d, int shape /* index into the shape table; -1 = cost model */,
                      const StemGeom* stem /* non-null: x is the raw image and the expand is the 3x3/2 stem (see kernels.hip) */,
                      hipStream_t st, const uint16_t* wep = nullptr /* non-null: phase 1 on the split-bf16 MFMA (expdw_bx_image) */,
                      int prec = 0 /* with wep: 1 = plain bf16 operands (one product) */,
                      int out_bf16 = 0 /* y is written as bf16 (bf16 activation storage; Cmid % 4 == 0) */,
                      int in_bf16 = 0 /* x holds bf16 values (bf16 residual stream): only where expdw_sk_pipe16 holds */);
// plain depthwise convolution through the same kernel (COPY mode: LDS-staged taps); shape as for launch_expand_dw, partial
// (nullable) [B, expdw_shape_slabs(shape, geo), C]
bool dwconv_lds_supported(const DwParams& p);
void launch_dwconv_lds(const DwParams& p, float* partial, int shape, hipStream_t st);
bool expdw_bx_ok(int Cin);
int expdw_kp(int Cin);
std::vector<uint16_t> expdw_bx_image(const float* We /*[Cmid][Cin]*/, int Cmid, int Cin);

// mean over H*W: in [B,HW,C] -> partial [B,S,C] (sums), S = number of pixel splits
int mean_splits(int HW);
void launch_mean_partial(const float* in, float* partial, int B, int HW, int C, int S, hipStream_t s);
// finish: out[b][c] = sum_s partial[b][s][c] / HW
void launch_mean_finish(const float* partial, float* out, int B, int HW, int C, int S, hipStream_t s);

struct SeParams {         // squeeze-excite FCs on pooled sums
    const float* partial; int S; int HW;         // pooled sums [B,S,C]
    const float* w1; const float* b1;            // [Cr, C], [Cr]
    const float* w2; const float* b2;            // w2 TRANSPOSED to [Cr, C] at plan time; [C]
    float* scale;                                // [B, C]
    int B, C, Cr, act1, act2;
    int threads = 0;                             // workgroup size (0 = 1024); see launch_se
};
void launch_se(const SeParams& p, hipStream_t s);

// ---- generic fallbacks (unfused graphs)
void launch_unary(const float* in, float* out, size_t n, int act, hipStream_t s);
// mode 0: same shape; mode 1: b is [B,1,1,C] broadcast over HW (a is [B,HW,C]); mode 2: b scalar
void launch_binary(const float* a, const float* b, float* out, size_t n, int op /*0 add 1 mul 2 sub*/, int mode,
                   int HW, int C, int act, hipStream_t s);

// ---- generic fallbacks, second tier (generic.hip): any float op the fused plan does not absorb
enum UnaryOp : int { U_ABS = 200, U_SQRT, U_RSQRT, U_LOG, U_EXP, U_NEG, U_SQUARE, U_TANH, U_LEAKY_RELU, U_ELU, U_SIN, U_COS,
                     U_FLOOR, U_CEIL, U_ROUND, U_RELU_N1_TO_1, U_LOGISTIC, U_RELU, U_RELU6, U_HARD_SWISH, U_GELU, U_GELU_TANH };
enum BinaryOp : int { B_ADD = 0, B_MUL = 1, B_SUB = 2, B_DIV = 3, B_POW = 4, B_MAX = 5, B_MIN = 6, B_SQDIFF = 7 };
enum ReduceOp : int { R_SUM = 0, R_MEAN = 1, R_MAX = 2, R_MIN = 3, R_PROD = 4 };
void launch_unary_op(const float* in, float* out, size_t n, int op, float alpha, hipStream_t s);
struct BcastParams {      // out[clip][i0..i3] = act(a (op) b); per-clip 4-D output dims, operand strides 0 on broadcast dims
    const float* a; const float* b; float* out;
    int d[4]; long sa[4], sb[4]; long bsa, bsb;      // bs*: clip stride (0 for a constant operand)
    int op, act;
};
void launch_binary_bcast(const BcastParams& p, int n_clips, hipStream_t s);
struct CopyParams {       // strided view copy, see generic.hip
    const float* in; float* out;
    int d[4]; long si[4], so[4]; long offi, offo, bsi, bso;
};
void launch_copy_view(const CopyParams& p, int n_clips, hipStream_t s);
void launch_fill(float* out, size_t n, float v, hipStream_t s);
struct PoolParams { const float* in; float* out; int B, H, W, C, Ho, Wo, kh, kw, sh, sw, pt, pl, mode /*0 avg, 1 max*/, act; };
void launch_pool2d(const PoolParams& p, hipStream_t s);
void launch_softmax_rows(const float* in, float* out, size_t rows, int n, float beta, hipStream_t s);
struct ReduceParams { const float* in; float* out; int d[4]; int mask /*bit k: dim k is reduced*/; int op; };
void launch_reduce(const ReduceParams& p, int n_clips, hipStream_t s);
struct GenConvParams {    // weights as in the file: OHWI (conv) or [1][kh][kw][Cout] (depthwise, Cout = Cin * mult)
    const float* in; const float* w; const float* bias; float* out;
    int B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, dh, dw, pt, pl, act, depthwise, mult;
};
void launch_conv_generic(const GenConvParams& p, hipStream_t s);

// ---- post-processing
// activation 0: float32(1/(1+exp(-sens*float64(x)))); 1: softmax (f32 max-sub, f64 exp, f32 sum); 2: f32-div sigmoid
void launch_activation(const float* logits, float* conf, int n_clips, int n_classes, int activation, double sens,
                       hipStream_t s);
void launch_topk(const float* conf, int n_clips, int n_classes, int k, float* out_conf, int32_t* out_idx,
                 hipStream_t s);

// ---- ultrasonic frame-CV (float64)
std::vector<double> us_twiddle_table(int fft_size);      // fft/2 (cos, -sin)(2 pi j / fft) pairs + the symmetric Hann window [fft]; uploaded once per (device, size)
void launch_us_frame_power(const void* samples /* float64, or int16 PCM when pcm16 */, int pcm16, int n_clips, int n, int fft_size,
                           int hop, int frames, int split_bin, const double* d_tw, double* powers /*[n_clips, frames]*/, hipStream_t s);
void launch_us_cv(const double* powers, int n_clips, int frames, double* cv, hipStream_t s);

}  // namespace bnhip
