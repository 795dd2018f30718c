/*
 * ORACLE — test infrastructure only.  Plain-C restatement of the reference's in-tree Go
 * arithmetic on either side of the native Invoke() call.  Never linked into libbnhip.so and
 * never called by the product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load liboracle.so.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 */
#define _USE_MATH_DEFINES
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- PCM -> float32: internal/audiocore/convert/pcm.go:226-268,
 *      internal/analysis/process.go:491-495 */
void orc_pcm16_to_f32(const uint8_t *b, int n, float *out) {
    for (int i = 0; i < n; i++) {
        int16_t s = (int16_t)((uint16_t)b[2 * i] | ((uint16_t)b[2 * i + 1] << 8));
        out[i] = (float)s / 32768.0f;
    }
}
void orc_pcm24_to_f32(const uint8_t *b, int n, float *out) {
    for (int i = 0; i < n; i++) {
        int32_t s = (int32_t)b[3 * i] | ((int32_t)b[3 * i + 1] << 8) | ((int32_t)b[3 * i + 2] << 16);
        if (s & 0x00800000) s |= ~0x00FFFFFF;
        out[i] = (float)s / 8388608.0f;
    }
}
void orc_pcm32_to_f32(const uint8_t *b, int n, float *out) {
    for (int i = 0; i < n; i++) {
        int32_t s = (int32_t)((uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) |
                              ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24));
        out[i] = (float)s / 2147483648.0f;
    }
}

/* ---- v2.4 confidence: internal/classifier/analyze.go:113-115,197-208
 *      conf = float32(1/(1+exp(-sensitivity*float64(logit)))) */
void orc_sigmoid_sens(const float *logits, int n, double sensitivity, float *out) {
    for (int i = 0; i < n; i++)
        out[i] = (float)(1.0 / (1.0 + exp(-sensitivity * (double)logits[i])));
}

/* ---- ONNX-side sigmoid: internal/inference/onnx/postprocess.go:8-10
 *      1.0 / (1.0 + float32(exp(float64(-x)))), division in float32 */
void orc_sigmoid_f32div(const float *logits, int n, float *out) {
    for (int i = 0; i < n; i++) {
        float e = (float)exp((double)(-logits[i]));
        out[i] = 1.0f / (1.0f + e);
    }
}

/* ---- softmax: internal/classifier/perch_onnx.go:315-335 == onnx/postprocess.go:20-43
 *      max-subtract in f32, exp in f64 narrowed to f32, f32 running sum in index order */
void orc_softmax(const float *x, int n, float *out) {
    if (n <= 0) return;
    float m = x[0];
    for (int i = 1; i < n; i++) if (x[i] > m) m = x[i];
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        out[i] = (float)exp((double)(x[i] - m));
        sum += out[i];
    }
    for (int i = 0; i < n; i++) out[i] /= sum;
}

/* ---- top-K: internal/classifier/analyze.go:220-301 (Lomuto quickselect, pivot = rightmost,
 *      strict '>' compare, then sort the first k descending).  idx[] carries label indices.
 *      Go's sort.Slice is unstable, so tie order is implementation-defined; callers compare
 *      tied scores as sets. */
typedef struct { float conf; int idx; } orc_res;

static int orc_partition(orc_res *r, int left, int right) {
    orc_res pivot = r[right];
    int i = left - 1;
    for (int j = left; j < right; j++) {
        if (r[j].conf > pivot.conf) {
            i++;
            orc_res t = r[i]; r[i] = r[j]; r[j] = t;
        }
    }
    orc_res t = r[i + 1]; r[i + 1] = r[right]; r[right] = t;
    return i + 1;
}
static int orc_cmp_desc(const void *a, const void *b) {
    float ca = ((const orc_res *)a)->conf, cb = ((const orc_res *)b)->conf;
    return (ca < cb) - (ca > cb);
}
/* returns number of results written (min(k,n)); conf_out/idx_out sized >= that */
int orc_topk(const float *conf, int n, int k, float *conf_out, int *idx_out) {
    if (n <= 0 || k <= 0) return 0;
    orc_res *r = (orc_res *)malloc(sizeof(orc_res) * (size_t)n);
    for (int i = 0; i < n; i++) { r[i].conf = conf[i]; r[i].idx = i; }
    int m = k < n ? k : n;
    if (k >= n) {
        qsort(r, (size_t)n, sizeof(orc_res), orc_cmp_desc);
    } else {
        int left = 0, right = n - 1;
        while (left < right) {
            int p = orc_partition(r, left, right);
            if (p == k - 1) break;
            if (p < k - 1) left = p + 1; else right = p - 1;
        }
        qsort(r, (size_t)k, sizeof(orc_res), orc_cmp_desc);
    }
    for (int i = 0; i < m; i++) { conf_out[i] = r[i].conf; idx_out[i] = r[i].idx; }
    free(r);
    return m;
}

/* ---- ultrasonic frame-CV: internal/audiocore/ultrasonic/filter.go:20-145 (all float64) */
typedef struct { double re, im; } orc_c128;

/* filter.go:101-136: bit reversal + butterflies with the w *= wn recurrence */
void orc_fft(orc_c128 *d, int n) {
    if (n <= 1) return;
    int j = 0;
    for (int i = 1; i < n; i++) {
        int bit = n >> 1;
        while (j & bit) { j ^= bit; bit >>= 1; }
        j ^= bit;
        if (i < j) { orc_c128 t = d[i]; d[i] = d[j]; d[j] = t; }
    }
    for (int size = 2; size <= n; size <<= 1) {
        int half = size >> 1;
        double th = -2.0 * M_PI / (double)size;
        orc_c128 wn = { cos(th), sin(th) };
        for (int start = 0; start < n; start += size) {
            orc_c128 w = { 1.0, 0.0 };
            for (int k = 0; k < half; k++) {
                orc_c128 u = d[start + k];
                orc_c128 x = d[start + k + half];
                orc_c128 v = { w.re * x.re - w.im * x.im, w.re * x.im + w.im * x.re };
                d[start + k].re = u.re + v.re; d[start + k].im = u.im + v.im;
                d[start + k + half].re = u.re - v.re; d[start + k + half].im = u.im - v.im;
                orc_c128 w2 = { w.re * wn.re - w.im * wn.im, w.re * wn.im + w.im * wn.re };
                w = w2;
            }
        }
    }
}

/* filter.go:139-145 symmetric Hann */
void orc_hanning(double *w, int n) {
    for (int i = 0; i < n; i++) w[i] = 0.5 * (1.0 - cos(2.0 * M_PI * (double)i / (double)(n - 1)));
}

/* filter.go:76-97 population std / mean */
double orc_cv(const double *v, int n) {
    if (n < 2) return 0.0;
    double sum = 0.0;
    for (int i = 0; i < n; i++) sum += v[i];
    double mean = sum / (double)n;
    if (mean <= 0.0) return 0.0;
    double sq = 0.0;
    for (int i = 0; i < n; i++) { double d = v[i] - mean; sq += d * d; }
    return sqrt(sq / (double)n) / mean;
}

/* filter.go:20-66.  Returns ok (1/0); *cv_out gets the CV; frame_powers (nullable) gets the
 * per-frame powers (caller sizes it 1+(n-fft)/hop). */
int orc_us_frame_cv(const double *samples, int n, int sample_rate, int fft_size, int hop,
                    int split_hz, double *cv_out, double *frame_powers) {
    *cv_out = 0.0;
    if (n < fft_size || sample_rate <= 0 || fft_size < 2 || hop <= 0) return 0;
    if (fft_size & (fft_size - 1)) return 0;
    if (split_hz < 0 || split_hz >= sample_rate / 2) return 0;
    int frames = 1 + (n - fft_size) / hop;
    if (frames < 2) return 0;
    double bin_width = (double)sample_rate / (double)fft_size;
    int split_bin = (int)((double)split_hz / bin_width);
    int nyq = fft_size / 2;
    double *win = (double *)malloc(sizeof(double) * (size_t)fft_size);
    double *pw = (double *)malloc(sizeof(double) * (size_t)frames);
    orc_c128 *buf = (orc_c128 *)malloc(sizeof(orc_c128) * (size_t)fft_size);
    orc_hanning(win, fft_size);
    for (int f = 0; f < frames; f++) {
        int off = f * hop;
        for (int i = 0; i < fft_size; i++) { buf[i].re = samples[off + i] * win[i]; buf[i].im = 0.0; }
        orc_fft(buf, fft_size);
        double power = 0.0;
        for (int b = split_bin; b <= nyq; b++) {
            double p = buf[b].re * buf[b].re + buf[b].im * buf[b].im;
            if (b > 0 && b < nyq) p *= 2.0;
            power += p;
        }
        pw[f] = power;
    }
    *cv_out = orc_cv(pw, frames);
    if (frame_powers) memcpy(frame_powers, pw, sizeof(double) * (size_t)frames);
    free(win); free(pw); free(buf);
    return 1;
}

/* ---- resampler edges: internal/audiocore/resample/resample.go:120-124,161-169
 *      in: float32(int16)/32768; out: clamp(f,-1,1), int16(f*32767) truncating toward zero */
void orc_resample_edge_out(const float *f, int n, int16_t *out) {
    for (int i = 0; i < n; i++) {
        float v = f[i];
        if (v > 1.0f) v = 1.0f;
        if (v < -1.0f) v = -1.0f;
        out[i] = (int16_t)(v * 32767.0f);
    }
}
