"""WAV/PCM ingest + clip framing feeding the batch API (SURVEY.md section 8f "next" row 1).

Formats follow the reference's converters (`internal/audiocore/convert/pcm.go:206-268`: 16/24/32-bit
little-endian integer PCM -> float32 by /32768, /8388608, /2147483648; WAVE_FORMAT_EXTENSIBLE with the
PCM sub-format is what `tawnyowl.wav` uses).  Framing follows the documented file-analysis behaviour
(`doc/wiki/file-analysis.md:1-44`, flags `cmd/root.go:93-95`): consecutive `clip_len` windows advanced by
`clip_len - overlap`; the final partial window is zero-padded to full length (kept if it holds at
least `min_tail` seconds of audio).  Host-side numpy; the conversion arithmetic is restated here
independently of the oracle (product code must not import oracle/).
"""
import struct

import numpy as np


class WavError(ValueError):
    pass


def read_wav(path_or_bytes):
    """-> (samples float32 [n] (first channel), sample_rate, bit_depth)."""
    raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if len(raw) < 12 or raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        raise WavError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack_from("<I", raw, pos + 4)[0]
        body = raw[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise WavError("missing fmt/data chunk")
    tag, channels, rate, _, block_align, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if tag == 0xFFFE:                               # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word is the real tag
        if len(fmt) < 40:
            raise WavError("truncated extensible fmt chunk")
        tag = struct.unpack_from("<H", fmt, 24)[0]
    if tag != 1:
        raise WavError(f"unsupported WAV format tag {tag} (integer PCM only)")
    if bits not in (16, 24, 32) or channels < 1:
        raise WavError(f"unsupported audio bit depth: {bits}")
    bps = bits // 8
    n = len(data) // (bps * channels)
    b = np.frombuffer(data, np.uint8, n * bps * channels).reshape(n, channels, bps)[:, 0, :]
    if bits == 16:
        s = b.copy().view("<i2").reshape(n).astype(np.float32) / np.float32(32768.0)
    elif bits == 24:
        v = (b[:, 0].astype(np.int32) | (b[:, 1].astype(np.int32) << 8) | (b[:, 2].astype(np.int32) << 16))
        v = np.where(v & 0x800000, v | ~0xFFFFFF, v).astype(np.int32)
        s = v.astype(np.float32) / np.float32(8388608.0)
    else:
        s = b.copy().view("<i4").reshape(n).astype(np.float32) / np.float32(2147483648.0)
    return s, int(rate), int(bits)


def frame_clips(samples, sample_rate, clip_seconds=3.0, overlap_seconds=0.0, min_tail_seconds=1.0):
    """-> (clips float32 [n_clips, clip_len], start_times_seconds [n_clips])."""
    clip_len = int(round(clip_seconds * sample_rate))
    hop = clip_len - int(round(overlap_seconds * sample_rate))
    if clip_len <= 0 or hop <= 0:
        raise ValueError("overlap must be smaller than the clip length")
    x = np.asarray(samples, np.float32).reshape(-1)
    starts = []
    p = 0
    while p < x.size:
        remain = x.size - p
        if remain >= clip_len or remain >= int(min_tail_seconds * sample_rate) or not starts:
            starts.append(p)
        if remain <= clip_len:
            break
        p += hop
    clips = np.zeros((len(starts), clip_len), np.float32)
    for i, s0 in enumerate(starts):
        seg = x[s0:s0 + clip_len]
        clips[i, :seg.size] = seg
    return clips, np.asarray(starts, np.float64) / sample_rate


def analyze_file(path, classifier, labels=None, sensitivity=1.0, overlap_seconds=0.0, top_k=10, threshold=0.0,
                 model_rate=48000, device=0):
    """File analysis = read -> (resample to the model's rate) -> frame -> batched predict_topk.
    Returns rows (start_s, end_s, label|index, confidence).

    `model_rate` is the sample rate the classifier's clip length is expressed in (48 kHz for BirdNET v2.4,
    `internal/classifier/model_registry.go:138-153`).  A file at another rate is resampled on the GPU first, as the
    reference resamples file input to the model rate (`internal/audiocore/resample/resample.go`, call sites
    `analysis/buffer_consumer.go:118,192`); window times are in seconds of audio, whatever the file's rate."""
    s, rate, _ = read_wav(path)
    if model_rate <= 0:
        raise WavError(f"invalid model sample rate {model_rate}")
    if rate != model_rate:
        from . import host                               # GPU resampler (bnhip_resample_f32); fails loudly without a device
        s = host.Resampler(rate, model_rate, device=device).resample_f32(s)
    clip_seconds = classifier.n_samples / float(model_rate)
    clips, starts = frame_clips(s, model_rate, clip_seconds, overlap_seconds)
    if clips.shape[1] != classifier.n_samples:
        raise WavError(f"framed clip length {clips.shape[1]} != model input {classifier.n_samples}")
    conf, idx = classifier.predict_topk(clips.reshape(-1), clips.shape[0], k=top_k, sensitivity=sensitivity)
    rows = []
    for i in range(clips.shape[0]):
        for c, j in zip(conf[i], idx[i]):
            if c >= threshold:
                rows.append((float(starts[i]), float(starts[i] + clip_seconds), labels[j] if labels else int(j), float(c)))
    return rows
