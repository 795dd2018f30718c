"""Synthetic-weights BirdNET v2.4-topology model, emitted as a real ``.tflite`` flatbuffer.

Why this exists: the reference embeds `BirdNET_GLOBAL_6K_V2.4_Model_FP32.tflite`
(`internal/classifier/birdnet.go:1195-1246`) but the blob is absent from the snapshot
(`.MISSING_LARGE_BLOBS`), and there is no network.  BASELINE.json's bench configs are quoted on
"synthetic" data and random-init weights of the architecture, so this module authors a model with
the published v2.4 I/O contract (`internal/classifier/model_registry.go:138-153`: 48 kHz, 3 s,
144000 samples in, 6522 logits out; 1024-d embedding `internal/inference/onnx/detection.go:18`)
using the op vocabulary a TF->TFLite conversion of the v2.4 graph produces:

  front-end  [EXTERNAL: BirdNET-Analyzer MelSpecLayerSimple]: per-clip min/max normalise to [-1,1];
             tf.signal.stft (periodic Hann, no padding) via GATHER-framing + RFFT2D; complex->float
             CAST (keeps the real part); mel matmul (96 bins); x^2; x^(1/(1+exp(mag_scale)));
             reverse mel axis; transpose -> [1,96,511,1]; two such channels
             (n_fft 2048/hop 278/0-3 kHz and n_fft 1024/hop 280/0.5-15 kHz) concatenated.
  backbone   EfficientNet-B0-style MBConv stack (CONV_2D / DEPTHWISE_CONV_2D with folded BN,
             swish as LOGISTIC+MUL, squeeze-excite as MEAN->CONV->swish->CONV->LOGISTIC->MUL,
             residual ADD), 1x1 top conv to 1024, global MEAN, FULLY_CONNECTED head.

Everything here is deterministic in `seed`.  It is a model *authoring* tool: the engine never
special-cases models produced by it; they go through the same TFLite reader as a real file would.
"""
from dataclasses import dataclass, field
from math import gcd

import numpy as np

from . import tflite_schema as S
from .tflite_build import GraphBuilder


@dataclass
class SpecConfig:
    frame_length: int
    frame_step: int
    fmin: float
    fmax: float
    fft_length: int = 0              # 0: = frame_length; larger: frames are zero-padded (PAD) before RFFT2D, as tf.signal.stft does


@dataclass
class SynthConfig:
    n_samples: int = 144000
    sample_rate: int = 48000
    specs: tuple = (SpecConfig(2048, 278, 0.0, 3000.0), SpecConfig(1024, 280, 500.0, 15000.0))
    n_mels: int = 96
    mag_scale: float = 1.23
    complex_mode: str = "real"       # "real" (CAST complex->float) or "abs" (COMPLEX_ABS)
    normalize: bool = True           # per-clip min/max normalisation in front of the STFT (v2.4); False: raw samples (Perch-style)
    pad: tuple = (0, 0)              # zero samples added left / right of the clip before framing (PAD)
    compress: str = "pow"            # "pow": x^2 then x^(1/(1+exp(mag_scale))) (v2.4);  "log": log_scale * log(max(x, log_floor))
    log_floor: float = 1e-2
    log_scale: float = 0.1
    time_major: bool = False         # True: spectrogram image is [1, frames, mel, 1] (no REVERSE / TRANSPOSE), Perch-style
    se_form: str = "conv"            # squeeze-excite spelling: "conv" (MEAN keep_dims -> 1x1 CONV_2Ds), "avgpool" (the same with a
                                     # whole-image AVERAGE_POOL_2D for the MEAN, also in front of the head), "keras" (MEAN -> RESHAPE [1,1,1,C] ->
                                     # CONV_2Ds, what GlobalAveragePooling2D + Reshape converts to), "dense" (MEAN -> FULLY_CONNECTED with fused
                                     # RELU -> FULLY_CONNECTED -> LOGISTIC -> RESHAPE [1,1,1,C])
    fe_forms: tuple = ()             # alternative op forms a converter may emit for the same front-end arithmetic:
                                     # "add_neg" (x + (-0.5) for x - 0.5), "bmm" (BATCH_MATMUL with the [bins, mel] matrix),
                                     # "bmm_adj" (BATCH_MATMUL, [mel, bins] matrix with adj_y), "square" (SQUARE for POW 2), "mul_self" (MUL(x, x))
    stem: int = 32
    # (expand_ratio, kernel, stride, out_channels, repeats)
    blocks: tuple = ((1, 3, 1, 16, 1), (6, 3, 2, 24, 2), (6, 5, 2, 40, 2), (6, 3, 2, 80, 3),
                     (6, 5, 1, 112, 3), (6, 5, 2, 192, 4), (6, 3, 1, 320, 1))
    se_ratio: float = 0.25
    top: int = 1024
    n_classes: int = 6522
    emit_embeddings: bool = False    # second graph output = 1024-d embedding (bat / Perch-style)
    perch_outputs: bool = False      # four outputs in Perch v2's order: embedding, spatial embedding, spectrogram, logits
    emb_first: bool = False          # two outputs with the embedding FIRST (one of the two orders BirdNET v3.0 exports come in)
    seed: int = 2024
    head_bias: float = -4.0
    name: str = "birdnet_v24_synth"


def tiny_config(**kw):
    """Small geometry for fast CPU-oracle tests (same op vocabulary, same code paths)."""
    base = dict(n_samples=12000, sample_rate=48000,
                specs=(SpecConfig(512, 94, 0.0, 3000.0), SpecConfig(256, 96, 500.0, 15000.0)),
                n_mels=32, stem=8,
                blocks=((1, 3, 1, 8, 1), (6, 3, 2, 12, 2), (6, 5, 2, 20, 1), (6, 3, 1, 24, 1)),
                top=64, n_classes=50, name="birdnet_tiny_synth")
    base.update(kw)
    return SynthConfig(**base)


def perch_config(**kw):
    """Stand-in for Google Perch v2 at its published dimensions (internal/classifier/model_registry.go:168-178: 32 kHz, 5 s,
    14795 classes; output shapes internal/inference/onnx/classifier.go:495-505: embedding [B,1536], spatial embedding
    [B,16,4,1536], spectrogram [B,500,128], logits [B,14795]): log-mel front-end (20 ms window / 10 ms hop, 128 bands,
    60 Hz - 16 kHz, 0.1 * log(max(mag, 1e-2))) producing a 500 x 128 time-major image, EfficientNet-B3-shaped MBConv stack
    (which turns 500 x 128 into exactly the 16 x 4 x 1536 spatial embedding the reference lists), 1536-d embedding, 14795
    logits.  The real artefact ships as ONNX and is absent from the snapshot: topology and front-end constants are
    [EXTERNAL - unverified]; this pins the workload size of BASELINE configs[4], not Perch's numerics."""
    base = dict(n_samples=160000, sample_rate=32000, specs=(SpecConfig(640, 320, 60.0, 16000.0, 1024),), n_mels=128,
                complex_mode="abs", normalize=False, pad=(160, 160), compress="log", time_major=True, stem=40,
                blocks=((1, 3, 1, 24, 2), (6, 3, 2, 32, 3), (6, 5, 2, 48, 3), (6, 3, 2, 96, 5), (6, 5, 1, 136, 5),
                        (6, 5, 2, 232, 6), (6, 3, 1, 384, 2)),
                top=1536, n_classes=14795, emit_embeddings=True, perch_outputs=True, head_bias=0.0, seed=2025,
                name="perch_v2_like_synth")
    base.update(kw)
    return SynthConfig(**base)


def tiny_perch_config(**kw):
    """Small geometry with the Perch-style front-end (log-mel, time-major, padded clip, frames shorter than the FFT)."""
    base = dict(n_samples=8000, sample_rate=32000, specs=(SpecConfig(320, 160, 60.0, 16000.0, 512),), n_mels=32,
                complex_mode="abs", normalize=False, pad=(80, 80), compress="log", time_major=True, stem=8,
                blocks=((1, 3, 1, 8, 2), (6, 3, 2, 12, 2), (6, 5, 2, 20, 1), (6, 3, 1, 24, 1)),
                top=64, n_classes=50, emit_embeddings=True, head_bias=0.0, name="perch_tiny_synth")
    base.update(kw)
    return SynthConfig(**base)


def n_frames(n_samples, frame_length, frame_step):
    return 1 + (n_samples - frame_length) // frame_step


def hann_periodic(n):
    """tf.signal.hann_window(periodic=True)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def mel_weight_matrix(n_mels, n_bins, sample_rate, fmin, fmax):
    """tf.signal.linear_to_mel_weight_matrix (HTK mel scale, unnormalised triangles)."""
    def hz2mel(f):
        return 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)
    nyq = sample_rate / 2.0
    lin = np.linspace(0.0, nyq, n_bins)[1:]
    bins_mel = hz2mel(lin)[:, None]
    edges = np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2)
    lo, ce, up = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower = (bins_mel - lo) / (ce - lo)
    upper = (up - bins_mel) / (up - ce)
    w = np.maximum(0.0, np.minimum(lower, upper))
    return np.pad(w, [[1, 0], [0, 0]]).astype(np.float32)   # [n_bins, n_mels]


def _same_out(n, s):
    return (n + s - 1) // s


def build_model(cfg: SynthConfig = None, container="tflite", dft="matmul", trunc=True) -> bytes:
    """container="tflite": the flatbuffer the reference's default backend loads; "onnx": the SAME graph and weights as an ONNX
    file (NCHW body) with the audio front-end in one of the forms exporters emit (onnx_audio.py: dft = "matmul" - the
    reference's "dfttrunc" files with trunc=True - "conv1d", "stft", "dft")."""
    cfg = cfg or SynthConfig()
    rng = np.random.default_rng(cfg.seed)
    g = GraphBuilder(description=f"{cfg.name} seed={cfg.seed} (synthetic weights)")
    i32 = lambda v: np.asarray(v, np.int32)
    f32 = lambda v: np.asarray(v, np.float32)

    x = g.tensor([1, cfg.n_samples], name="INPUT")

    # ------------------------------------------------------------------ front-end
    chans = []
    n_pad = cfg.n_samples + cfg.pad[0] + cfg.pad[1]
    for ci, sp in enumerate(cfg.specs):
        L, hop = sp.frame_length, sp.frame_step
        Lfft = sp.fft_length or L
        F = n_frames(n_pad, L, hop)
        nb = Lfft // 2 + 1
        pre = f"MEL{ci}/"
        if cfg.normalize:
            ax1 = g.const(i32([1]), pre + "axis")
            mn = g.op("REDUCE_MIN", [x, ax1], [1, 1], dict(keep_dims=1))
            s1 = g.op("SUB", [x, mn], [1, cfg.n_samples], {})
            mx = g.op("REDUCE_MAX", [s1, ax1], [1, 1], dict(keep_dims=1))
            dn = g.op("ADD", [mx, g.const(f32(1e-6))], [1, 1], {})
            nm = g.op("DIV", [s1, dn], [1, cfg.n_samples], {})
            if "add_neg" in cfg.fe_forms:
                n2 = g.op("ADD", [nm, g.const(f32(-0.5))], [1, cfg.n_samples], {})
            else:
                n2 = g.op("SUB", [nm, g.const(f32(0.5))], [1, cfg.n_samples], {})
            xn = g.op("MUL", [n2, g.const(f32(2.0))], [1, cfg.n_samples], {})
        else:
            xn = x
        if cfg.pad != (0, 0):
            xn = g.op("PAD", [xn, g.const(i32([[0, 0], list(cfg.pad)]), pre + "clip_pad")], [1, n_pad], {})
        # tf.signal.frame: sub-frames of gcd(L,hop) samples, gathered
        sub = gcd(L, hop)
        nsub = n_pad // sub
        r1 = g.op("RESHAPE", [xn, g.const(i32([1, nsub, sub]))], [1, nsub, sub],
                  dict(new_shape=[1, nsub, sub]))
        sel = (np.arange(F)[:, None] * (hop // sub) + np.arange(L // sub)[None, :]).astype(np.int32)
        ga = g.op("GATHER", [r1, g.const(sel, pre + "frame_selector")], [1, F, L // sub, sub],
                  dict(axis=1, batch_dims=0))
        fr = g.op("RESHAPE", [ga, g.const(i32([1, F, L]))], [1, F, L], dict(new_shape=[1, F, L]))
        wn = g.op("MUL", [fr, g.const(hann_periodic(L), pre + "hann")], [1, F, L], {})
        if Lfft != L:
            wn = g.op("PAD", [wn, g.const(i32([[0, 0], [0, 0], [0, Lfft - L]]), pre + "frame_pad")], [1, F, Lfft], {})
        e1 = g.op("RESHAPE", [wn, g.const(i32([1, F, 1, Lfft]))], [1, F, 1, Lfft],
                  dict(new_shape=[1, F, 1, Lfft]))
        ft = g.op("RFFT2D", [e1, g.const(i32([1, Lfft]), pre + "fft_length")], [1, F, 1, nb], {},
                  out_dtype=S.COMPLEX64)
        sq = g.op("RESHAPE", [ft, g.const(i32([1, F, nb]))], [1, F, nb],
                  dict(new_shape=[1, F, nb]), out_dtype=S.COMPLEX64)
        if cfg.complex_mode == "real":
            re = g.op("CAST", [sq], [1, F, nb], dict(in_data_type=S.COMPLEX64, out_data_type=S.FLOAT32))
        else:
            re = g.op("COMPLEX_ABS", [sq], [1, F, nb], {})
        r2 = g.op("RESHAPE", [re, g.const(i32([F, nb]))], [F, nb], dict(new_shape=[F, nb]))
        mel = mel_weight_matrix(cfg.n_mels, nb, cfg.sample_rate, sp.fmin, sp.fmax)
        if "bmm" in cfg.fe_forms:
            mm = g.op("BATCH_MATMUL", [r2, g.const(np.ascontiguousarray(mel), pre + "mel")], [F, cfg.n_mels], dict(adj_x=0, adj_y=0))
        elif "bmm_adj" in cfg.fe_forms:
            mm = g.op("BATCH_MATMUL", [r2, g.const(np.ascontiguousarray(mel.T), pre + "mel")], [F, cfg.n_mels], dict(adj_x=0, adj_y=1))
        else:
            mm = g.op("FULLY_CONNECTED", [r2, g.const(np.ascontiguousarray(mel.T), pre + "mel"), -1],
                      [F, cfg.n_mels], dict(fused_activation_function=S.ACT_NONE))
        r3 = g.op("RESHAPE", [mm, g.const(i32([1, F, cfg.n_mels]))], [1, F, cfg.n_mels],
                  dict(new_shape=[1, F, cfg.n_mels]))
        if cfg.compress == "pow":
            if "square" in cfg.fe_forms:
                p1 = g.op("SQUARE", [r3], [1, F, cfg.n_mels], {})
            elif "mul_self" in cfg.fe_forms:
                p1 = g.op("MUL", [r3, r3], [1, F, cfg.n_mels], {})
            else:
                p1 = g.op("POW", [r3, g.const(f32(2.0))], [1, F, cfg.n_mels], {})
            expo = 1.0 / (1.0 + np.exp(cfg.mag_scale))
            cz = g.op("POW", [p1, g.const(f32(expo), pre + "mag_exponent")], [1, F, cfg.n_mels], {})
        else:
            fl = g.op("MAXIMUM", [r3, g.const(f32(cfg.log_floor), pre + "log_floor")], [1, F, cfg.n_mels], {})
            lg = g.op("LOG", [fl], [1, F, cfg.n_mels], {})
            cz = g.op("MUL", [lg, g.const(f32(cfg.log_scale), pre + "log_scale")], [1, F, cfg.n_mels], {})
        if cfg.time_major:
            ex = g.op("RESHAPE", [cz, g.const(i32([1, F, cfg.n_mels, 1]))], [1, F, cfg.n_mels, 1],
                      dict(new_shape=[1, F, cfg.n_mels, 1]))
        else:
            rv = g.op("REVERSE_V2", [cz, g.const(i32([2]))], [1, F, cfg.n_mels], {})
            tr = g.op("TRANSPOSE", [rv, g.const(i32([0, 2, 1]))], [1, cfg.n_mels, F], {})
            ex = g.op("RESHAPE", [tr, g.const(i32([1, cfg.n_mels, F, 1]))], [1, cfg.n_mels, F, 1],
                      dict(new_shape=[1, cfg.n_mels, F, 1]))
        chans.append(ex)
    F0 = n_frames(n_pad, cfg.specs[0].frame_length, cfg.specs[0].frame_step)
    H, W = (F0, cfg.n_mels) if cfg.time_major else (cfg.n_mels, F0)
    for sp in cfg.specs[1:]:
        assert n_frames(n_pad, sp.frame_length, sp.frame_step) == F0
    C = len(chans)
    t = g.op("CONCATENATION", chans, [1, H, W, C], dict(axis=3)) if C > 1 else chans[0]

    # ------------------------------------------------------------------ helpers
    def conv(t, cin, cout, k, s, act, gain, H, W, name):
        w = rng.standard_normal((cout, k, k, cin)).astype(np.float32) * np.float32(gain / np.sqrt(k * k * cin))
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        Ho, Wo = _same_out(H, s), _same_out(W, s)
        y = g.op("CONV_2D", [t, g.const(w, name + "/w"), g.const(b, name + "/b")], [1, Ho, Wo, cout],
                 dict(padding=S.PAD_SAME, stride_w=s, stride_h=s, fused_activation_function=S.ACT_NONE,
                      dilation_w_factor=1, dilation_h_factor=1), name=name)
        if act == "swish":
            sg = g.op("LOGISTIC", [y], [1, Ho, Wo, cout])
            y = g.op("MUL", [y, sg], [1, Ho, Wo, cout], {})
        elif act == "sigmoid":
            y = g.op("LOGISTIC", [y], [1, Ho, Wo, cout])
        return y, Ho, Wo

    def dwconv(t, c, k, s, gain, H, W, name):
        w = rng.standard_normal((1, k, k, c)).astype(np.float32) * np.float32(gain / np.sqrt(k * k))
        b = (rng.standard_normal(c) * 0.1).astype(np.float32)
        Ho, Wo = _same_out(H, s), _same_out(W, s)
        y = g.op("DEPTHWISE_CONV_2D", [t, g.const(w, name + "/w"), g.const(b, name + "/b")],
                 [1, Ho, Wo, c],
                 dict(padding=S.PAD_SAME, stride_w=s, stride_h=s, depth_multiplier=1,
                      fused_activation_function=S.ACT_NONE, dilation_w_factor=1, dilation_h_factor=1),
                 name=name)
        sg = g.op("LOGISTIC", [y], [1, Ho, Wo, c])
        y = g.op("MUL", [y, sg], [1, Ho, Wo, c], {})
        return y, Ho, Wo

    # ------------------------------------------------------------------ backbone
    t, H, W = conv(t, C, cfg.stem, 3, 2, "swish", 1.6, H, W, "stem")
    cin = cfg.stem
    bi = 0
    for (er, k, s, cout, reps) in cfg.blocks:
        for r in range(reps):
            bi += 1
            stride = s if r == 0 else 1
            name = f"b{bi}"
            inp = t
            mid = cin * er
            if er != 1:
                t, H, W = conv(t, cin, mid, 1, 1, "swish", 1.6, H, W, name + "/expand")
            t, H, W = dwconv(t, mid, k, stride, 1.6, H, W, name + "/dw")
            # squeeze-excite (se_form "none": a plain inverted-residual block, MobileNetV2-style - no per-tile sums in the fused kernels)
            cse = max(1, int(cin * cfg.se_ratio))
            if cfg.se_form == "none":
                t, H, W = conv(t, mid, cout, 1, 1, None, 1.4, H, W, name + "/project")
                if stride == 1 and cin == cout:
                    t = g.op("ADD", [t, inp], [1, H, W, cout], dict(fused_activation_function=S.ACT_NONE))
                cin = cout
                continue
            if cfg.se_form in ("conv", "avgpool"):
                if cfg.se_form == "avgpool":
                    m = g.op("AVERAGE_POOL_2D", [t], [1, 1, 1, mid], dict(padding=S.PAD_VALID, stride_w=W, stride_h=H, filter_width=W,
                                                                           filter_height=H, fused_activation_function=S.ACT_NONE))
                else:
                    m = g.op("MEAN", [t, g.const(i32([1, 2]))], [1, 1, 1, mid], dict(keep_dims=1))
                m, _, _ = conv(m, mid, cse, 1, 1, "swish", 1.0, 1, 1, name + "/se_reduce")
                m, _, _ = conv(m, cse, mid, 1, 1, "sigmoid", 1.0, 1, 1, name + "/se_expand")
            elif cfg.se_form == "keras":
                m = g.op("MEAN", [t, g.const(i32([1, 2]))], [1, mid], dict(keep_dims=0))
                m = g.op("RESHAPE", [m, g.const(i32([1, 1, 1, mid]))], [1, 1, 1, mid], dict(new_shape=[1, 1, 1, mid]))
                m, _, _ = conv(m, mid, cse, 1, 1, "swish", 1.0, 1, 1, name + "/se_reduce")
                m, _, _ = conv(m, cse, mid, 1, 1, "sigmoid", 1.0, 1, 1, name + "/se_expand")
            else:
                m = g.op("MEAN", [t, g.const(i32([1, 2]))], [1, mid], dict(keep_dims=0))
                w1 = (rng.standard_normal((cse, mid)) / np.sqrt(mid)).astype(np.float32)
                w2 = (rng.standard_normal((mid, cse)) / np.sqrt(cse)).astype(np.float32)
                m = g.op("FULLY_CONNECTED", [m, g.const(w1, name + "/se_fc1/w"), g.const((rng.standard_normal(cse) * 0.1).astype(np.float32))],
                         [1, cse], dict(fused_activation_function=S.ACT_RELU))
                m = g.op("FULLY_CONNECTED", [m, g.const(w2, name + "/se_fc2/w"), g.const((rng.standard_normal(mid) * 0.1).astype(np.float32))],
                         [1, mid], dict(fused_activation_function=S.ACT_NONE))
                m = g.op("LOGISTIC", [m], [1, mid])
                m = g.op("RESHAPE", [m, g.const(i32([1, 1, 1, mid]))], [1, 1, 1, mid], dict(new_shape=[1, 1, 1, mid]))
            t = g.op("MUL", [t, m], [1, H, W, mid], {})
            t, H, W = conv(t, mid, cout, 1, 1, None, 1.4, H, W, name + "/project")
            if stride == 1 and cin == cout:
                t = g.op("ADD", [t, inp], [1, H, W, cout], dict(fused_activation_function=S.ACT_NONE))
            cin = cout
    t, H, W = conv(t, cin, cfg.top, 1, 1, "swish", 1.6, H, W, "top")
    if cfg.se_form == "avgpool":
        gp = g.op("AVERAGE_POOL_2D", [t], [1, 1, 1, cfg.top], dict(padding=S.PAD_VALID, stride_w=W, stride_h=H, filter_width=W,
                                                                    filter_height=H, fused_activation_function=S.ACT_NONE))
        emb4 = g.op("RESHAPE", [gp, g.const(i32([1, cfg.top]))], [1, cfg.top], dict(new_shape=[1, cfg.top]), name="GLOBAL_AVG_POOL")
    else:
        emb4 = g.op("MEAN", [t, g.const(i32([1, 2]))], [1, cfg.top], dict(keep_dims=0), name="GLOBAL_AVG_POOL")
    wh = (rng.standard_normal((cfg.n_classes, cfg.top)) * (2.0 / np.sqrt(cfg.top))).astype(np.float32)
    bh = (cfg.head_bias + rng.standard_normal(cfg.n_classes) * 0.5).astype(np.float32)
    logits = g.op("FULLY_CONNECTED", [emb4, g.const(wh, "head/w"), g.const(bh, "head/b")],
                  [1, cfg.n_classes], dict(fused_activation_function=S.ACT_NONE), name="CLASS_DENSE_LAYER")
    outs = [logits, emb4] if cfg.emit_embeddings else [logits]
    if cfg.emit_embeddings and cfg.emb_first:
        outs = [emb4, logits]
    if cfg.perch_outputs:            # internal/inference/onnx/classifier.go:495-505: [B,1536], [B,16,4,1536], [B,500,128], [B,14795]
        assert len(chans) == 1
        spec3 = g.op("RESHAPE", [chans[0], g.const(i32([1, F0, cfg.n_mels]))], [1, F0, cfg.n_mels], dict(new_shape=[1, F0, cfg.n_mels]))
        outs = [emb4, t, spec3, logits]
    if container == "onnx":
        from . import onnx_audio
        return onnx_audio.transcribe(g, [x], outs, dft=dft, trunc=trunc)
    return g.finish([x], outs)


def build_dense_model(dims, hidden_act="relu", final_sigmoid=False, fp16_weights=False, seed=7, name="dense_synth",
                      input_scale=None):
    """Dense-only graph on a [1, dims[0]] input: the shape of the reference's secondary models -
    a BattyBirdNET regional head (1024 -> C logits; `internal/classifier/bat_onnx.go:282`,
    `internal/inference/backend.go:31-52` CustomClassifier) and the range-filter meta-model
    ([lat, lon, week] -> per-species occurrence, in-graph sigmoid; `backend.go:55-76`).  With
    fp16_weights the constants are stored as float16 behind DEQUANTIZE ops, like the reference's
    `BirdNET_GLOBAL_6K_V2.4_MData_Model_V2_FP16.tflite`."""
    rng = np.random.default_rng(seed)
    g = GraphBuilder(description=f"{name} seed={seed} (synthetic weights)")
    x = g.tensor([1, dims[0]], name="INPUT")
    t = x
    for li in range(len(dims) - 1):
        cin, cout = dims[li], dims[li + 1]
        last = li == len(dims) - 2
        w = (rng.standard_normal((cout, cin)) * (1.0 / np.sqrt(cin))).astype(np.float32)
        if li == 0 and input_scale is not None:      # fold an input normalisation (e.g. lat/90, lon/180, week/48) into layer 0
            w = (w / np.asarray(input_scale, np.float32)[None, :]).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        if fp16_weights:
            w16 = g.const(w.astype(np.float16), f"fc{li}/w_f16")
            wt = g.op("DEQUANTIZE", [w16], [cout, cin], name=f"fc{li}/w")
            b16 = g.const(b.astype(np.float16), f"fc{li}/b_f16")
            bt = g.op("DEQUANTIZE", [b16], [cout], name=f"fc{li}/b")
        else:
            wt, bt = g.const(w, f"fc{li}/w"), g.const(b, f"fc{li}/b")
        act = S.ACT_NONE if (last or hidden_act != "relu") else S.ACT_RELU
        t = g.op("FULLY_CONNECTED", [t, wt, bt], [1, cout], dict(fused_activation_function=act), name=f"fc{li}")
        if not last and hidden_act == "swish":
            sg = g.op("LOGISTIC", [t], [1, cout])
            t = g.op("MUL", [t, sg], [1, cout], {})
    if final_sigmoid:
        t = g.op("LOGISTIC", [t], [1, dims[-1]], name="OUTPUT")
    return g.finish([x], [t])


def synth_clips(n, n_samples=144000, sample_rate=48000, first=0):
    """BASELINE.json config-2 input: clip i = 0.5*sin(2*pi*f_i*t), f_i = 500+37*i Hz, plus
    N(0,0.05^2) noise from numpy default_rng(1234+i), clamped to [-1,1]."""
    t = np.arange(n_samples, dtype=np.float64) / sample_rate
    out = np.empty((n, n_samples), np.float32)
    for j in range(n):
        i = first + j
        f = 500.0 + 37.0 * i
        sig = 0.5 * np.sin(2.0 * np.pi * f * t) + np.random.default_rng(1234 + i).normal(0.0, 0.05, n_samples)
        out[j] = np.clip(sig, -1.0, 1.0).astype(np.float32)
    return out
