"""PyTorch re-implementations of the model families the reference runs, used as an INDEPENDENT WRITER (torch's own ONNX
exporter serialises them) and an INDEPENDENT EXECUTOR (torch-CPU fp32 / fp64 forward passes) for the parity fixtures.

Nothing here shares code with the repo's own model writers (`birdnet-go_amd/synth_model.py`, `onnx_build.py`,
`onnx_audio.py`), its readers or its oracles: the graphs below are written the way a PyTorch user would write them
(`torch.stft`, `nn.Conv2d` + `nn.BatchNorm2d` + `nn.SiLU`, `nn.AdaptiveAvgPool2d`, `torch.flip`, `nn.Linear`), and the
ONNX files are whatever `torch.onnx.export` makes of that.  The reference's real artefacts (BirdNET v2.4 `.tflite` /
`.onnx`, Perch v2 `.onnx`, BattyBirdNET heads: `internal/classifier/model_catalog.go:296-311,466-501`) are absent from the
snapshot, so the weights are random (seeded) and only the I/O contracts are the reference's
(`internal/inference/onnx/detection.go:9-115`, `classifier.go:470-508`):

  * `BirdNetLike`   [N, 144000] @ 48 kHz -> logits [N, 6522] (+ embeddings [N, 1024]); in-graph mel front-end after
                    BirdNET-Analyzer's MelSpecLayerSimple [EXTERNAL]: min/max normalise, two STFTs (2048/278, 1024/280), real
                    part, 96-band mel, x^2, x^(1/(1+e^1.23)), frequency flip, [N, 2, 96, 511]; EfficientNet-style MBConv/SE body.
                    Two spellings of the DFT: `torch.stft` (ONNX `STFT`) and a strided `nn.Conv1d` with cos / -sin kernels.
  * `PerchLike`     [N, 160000] @ 32 kHz -> embedding [N, 1536], spatial embedding [N, 16, 4, 1536], spectrogram
                    [N, 500, 128], logits [N, 14795] (the reference's output order, `perch_onnx.go:28`), log-mel front-end,
                    NCHW body.
  * `BatHead`       [N, 1024] embedding -> [N, C <= 38] regional head (`bat_onnx.go:252-282`).

The ONNX exporter in this image needs a one-line patch: `torch.onnx.export(dynamo=False)` ends by importing the `onnx`
package (to splice onnxscript functions into the proto), which is not installed; no such functions exist in these graphs,
so that final step is replaced by the identity.  Everything else is torch's stock TorchScript exporter.
"""
import io
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, np.float64) / 700.0)


def mel_filterbank(n_mels, n_fft, sample_rate, fmin, fmax):
    """HTK-scale triangular filters on the one-sided DFT bins, [n_fft // 2 + 1, n_mels] (unnormalised; DC row zero)."""
    n_bins = n_fft // 2 + 1
    freqs = np.arange(n_bins, dtype=np.float64) * sample_rate / n_fft
    pts = np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2)
    fm = hz_to_mel(freqs)
    fb = np.zeros((n_bins, n_mels), np.float64)
    for m in range(n_mels):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        up = (fm - lo) / (ce - lo)
        down = (hi - fm) / (hi - ce)
        fb[:, m] = np.maximum(0.0, np.minimum(up, down))
    fb[0, :] = 0.0
    return torch.tensor(fb, dtype=torch.float32)


class MelSpec(nn.Module):
    """One spectrogram channel.  dft = "stft": torch.stft;  "conv": strided Conv1d whose kernels are the windowed DFT rows."""

    def __init__(self, sample_rate, n_fft, hop, n_mels, fmin, fmax, mag_scale=1.23, dft="stft"):
        super().__init__()
        self.n_fft, self.hop, self.dft = n_fft, hop, dft
        self.register_buffer("window", torch.hann_window(n_fft, periodic=True))
        self.register_buffer("mel", mel_filterbank(n_mels, n_fft, sample_rate, fmin, fmax))
        self.exponent = 1.0 / (1.0 + math.exp(mag_scale))
        if dft == "conv":
            n = torch.arange(n_fft, dtype=torch.float64)
            k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)
            ang = 2.0 * math.pi * k[:, None] * n[None, :] / n_fft
            w = self.window.double()[None, :]
            kern = torch.cat([torch.cos(ang) * w, -torch.sin(ang) * w], 0).float()      # [2K, n_fft]
            self.register_buffer("dft_kernel", kern[:, None, :])

    def forward(self, x):                                      # x: [N, T] normalised samples
        if self.dft == "conv":
            ri = F.conv1d(x[:, None, :], self.dft_kernel, stride=self.hop)             # [N, 2K, frames]
            re = ri[:, : self.n_fft // 2 + 1, :]
        else:
            st = torch.stft(x, self.n_fft, hop_length=self.hop, window=self.window, center=False, return_complex=False)
            re = st[..., 0]                                    # [N, K, frames]: tf.cast(complex -> float) keeps the real part
        spec = torch.matmul(re.transpose(1, 2), self.mel)      # [N, frames, mels]
        spec = spec.pow(2.0)
        spec = spec.pow(self.exponent)
        spec = torch.flip(spec, dims=[2])
        return spec.transpose(1, 2)                            # [N, mels, frames]


class ConvBNAct(nn.Sequential):
    def __init__(self, cin, cout, k, stride=1, groups=1, act=True):
        layers = [nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False), nn.BatchNorm2d(cout)]
        if act:
            layers.append(nn.SiLU())
        super().__init__(*layers)


class SqueezeExcite(nn.Module):
    def __init__(self, c, r):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(c, r, 1)
        self.fc2 = nn.Conv2d(r, c, 1)

    def forward(self, x):
        s = self.pool(x)
        s = F.silu(self.fc1(s))
        return x * torch.sigmoid(self.fc2(s))


class MBConv(nn.Module):
    def __init__(self, cin, cout, expand, k, stride, se_ratio=0.25):
        super().__init__()
        mid = cin * expand
        self.expand = ConvBNAct(cin, mid, 1) if expand != 1 else nn.Identity()
        self.dw = ConvBNAct(mid, mid, k, stride, groups=mid)
        self.se = SqueezeExcite(mid, max(1, int(cin * se_ratio)))
        self.project = ConvBNAct(mid, cout, 1, act=False)
        self.residual = stride == 1 and cin == cout

    def forward(self, x):
        y = self.project(self.se(self.dw(self.expand(x))))
        return x + y if self.residual else y


def _body(cin, stem, blocks, top):
    layers = [ConvBNAct(cin, stem, 3, 2)]
    c = stem
    for expand, k, stride, cout, reps in blocks:
        for r in range(reps):
            layers.append(MBConv(c, cout, expand, k, stride if r == 0 else 1))
            c = cout
    layers.append(ConvBNAct(c, top, 1))
    return nn.Sequential(*layers)


def _randomise_bn(model, gen):
    """Give every BatchNorm non-trivial statistics and affine terms (default init would fold to the identity), and the
    convolutions / dense layers enough gain that the logits spread over several units (torch's default init lets the signal
    die out over 20 layers: every clip would get the same logits to 1e-2 and a top-1 comparison would mean nothing)."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            fan_in = m.in_channels // m.groups * m.kernel_size[0] * m.kernel_size[1]
            m.weight.data.copy_(torch.randn(m.weight.shape, generator=gen) * math.sqrt(2.2 / fan_in))
            if m.bias is not None:
                m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
        elif isinstance(m, nn.Linear):
            m.weight.data.copy_(torch.randn(m.weight.shape, generator=gen) * math.sqrt(6.0 / m.in_features))
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            n = m.num_features
            m.running_mean.copy_(torch.randn(n, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(n, generator=gen) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(n, generator=gen) * 0.6 + 0.9)
            m.bias.data.copy_(torch.randn(n, generator=gen) * 0.1)


def _calibrate(model, example, head_std=2.5):
    """One data-dependent rescaling pass (LSUV-style): every convolution / dense layer is scaled so that its output has unit
    standard deviation on `example` (the logits: `head_std`).  Deterministic; keeps 30-layer random stacks in a numerically
    sane range, so fp32-vs-fp64 differences stay at rounding level and logits spread over a few units."""
    hooks = []

    def hook(mod, inp, out):
        target = head_std if isinstance(mod, nn.Linear) and mod.out_features > 64 else 1.0
        k = target / max(float(out.std()), 1e-6)
        mod.weight.data.mul_(k)
        if mod.bias is not None:
            mod.bias.data.mul_(k)
        return out * k

    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            hooks.append(m.register_forward_hook(hook))
    model.eval()
    with torch.no_grad():
        model(example)
    for h in hooks:
        h.remove()


def _calibration_clips(n_samples, sample_rate, n=2):
    t = torch.arange(n_samples, dtype=torch.float64) / sample_rate
    g = torch.Generator().manual_seed(99)
    clips = [(0.4 * torch.sin(2 * math.pi * (700.0 + 450.0 * i) * t) + 0.1 * torch.randn(n_samples, generator=g, dtype=torch.float64)).float() for i in range(n)]
    clips.append((torch.rand(n_samples, generator=g, dtype=torch.float64) * 2.0 - 1.0).float())          # full-scale noise (the reference's Perch benchmark input)
    clips.append((0.01 * torch.randn(n_samples, generator=g, dtype=torch.float64)).float())             # near silence
    return torch.stack(clips)


BIRDNET_BLOCKS = ((1, 3, 1, 16, 1), (6, 3, 2, 24, 2), (6, 5, 2, 40, 2), (6, 3, 2, 80, 3), (6, 5, 1, 112, 3), (6, 5, 2, 192, 4), (6, 3, 1, 320, 1))
BIRDNET_TINY_BLOCKS = ((1, 3, 1, 8, 1), (6, 3, 2, 12, 2), (6, 5, 2, 20, 1), (6, 3, 1, 24, 1))
PERCH_BLOCKS = ((1, 3, 1, 24, 2), (6, 3, 2, 32, 3), (6, 5, 2, 48, 3), (6, 3, 2, 96, 5), (6, 5, 1, 136, 5), (6, 5, 2, 232, 6), (6, 3, 1, 384, 2))
PERCH_TINY_BLOCKS = ((1, 3, 1, 8, 2), (6, 3, 2, 12, 2), (6, 5, 2, 20, 1), (6, 3, 2, 24, 1), (6, 3, 2, 24, 1))


class BirdNetLike(nn.Module):
    def __init__(self, size="full", dft="stft", embeddings=False, seed=2401):
        super().__init__()
        torch.manual_seed(seed)
        if size == "full":
            self.n_samples, sr, n_mels, stem, blocks, top, n_classes = 144000, 48000, 96, 32, BIRDNET_BLOCKS, 1024, 6522
            specs = ((2048, 278, 0.0, 3000.0), (1024, 280, 500.0, 15000.0))
        else:
            self.n_samples, sr, n_mels, stem, blocks, top, n_classes = 12000, 48000, 32, 8, BIRDNET_TINY_BLOCKS, 64, 50
            specs = ((512, 94, 0.0, 3000.0), (256, 96, 500.0, 15000.0))
        self.embeddings = embeddings
        self.spec1 = MelSpec(sr, specs[0][0], specs[0][1], n_mels, specs[0][2], specs[0][3], dft=dft)
        self.spec2 = MelSpec(sr, specs[1][0], specs[1][1], n_mels, specs[1][2], specs[1][3], dft=dft)
        self.body = _body(2, stem, blocks, top)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.drop = nn.Dropout(0.25)
        self.head = nn.Linear(top, n_classes)
        gen = torch.Generator().manual_seed(seed + 1)
        _randomise_bn(self, gen)
        _calibrate(self, _calibration_clips(self.n_samples, sr))
        with torch.no_grad():
            self.head.bias.fill_(-4.0)

    def forward(self, x):
        lo = x.min(dim=1, keepdim=True)[0]
        x = x - lo
        x = x / (x.max(dim=1, keepdim=True)[0] + 1e-6)
        x = (x - 0.5) * 2.0
        img = torch.stack([self.spec1(x), self.spec2(x)], dim=1)           # [N, 2, mels, frames]
        emb = self.pool(self.body(img)).flatten(1)
        logits = self.head(self.drop(emb))
        return (logits, emb) if self.embeddings else logits


class PerchLike(nn.Module):
    def __init__(self, size="full", seed=2502):
        super().__init__()
        torch.manual_seed(seed)
        if size == "full":
            self.n_samples, sr, self.n_fft, self.win, self.hop, n_mels = 160000, 32000, 1024, 640, 320, 128
            stem, blocks, top, n_classes, self.frames = 40, PERCH_BLOCKS, 1536, 14795, 500
        else:
            self.n_samples, sr, self.n_fft, self.win, self.hop, n_mels = 8000, 32000, 512, 320, 160, 32
            stem, blocks, top, n_classes, self.frames = 8, PERCH_TINY_BLOCKS, 64, 50, 50
        self.pad = (self.n_fft + (self.frames - 1) * self.hop - self.n_samples) // 2
        assert 2 * self.pad + self.n_samples == self.n_fft + (self.frames - 1) * self.hop
        self.register_buffer("window", torch.hann_window(self.win, periodic=True))
        self.register_buffer("mel", mel_filterbank(n_mels, self.n_fft, sr, 60.0, 16000.0))
        self.body = _body(1, stem, blocks, top)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(top, n_classes)
        _randomise_bn(self, torch.Generator().manual_seed(seed + 1))
        _calibrate(self, _calibration_clips(self.n_samples, sr))
        with torch.no_grad():
            self.head.bias.zero_()

    def forward(self, x):
        xp = F.pad(x, (self.pad, self.pad))
        st = torch.stft(xp, self.n_fft, hop_length=self.hop, win_length=self.win, window=self.window, center=False, return_complex=False)
        mag = torch.sqrt(st[..., 0] ** 2 + st[..., 1] ** 2)                  # [N, K, frames]
        mel = torch.matmul(mag.transpose(1, 2), self.mel)                   # [N, frames, mels]
        spec = 0.1 * torch.log(torch.clamp(mel, min=1e-2))
        feat = self.body(spec[:, None, :, :])                               # [N, C, H, W]
        emb = self.pool(feat).flatten(1)
        return emb, feat.permute(0, 2, 3, 1), spec, self.head(emb)


class BatHead(nn.Module):
    def __init__(self, dim=1024, hidden=128, n_classes=38, seed=2603):
        super().__init__()
        torch.manual_seed(seed)
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.ReLU(), nn.Dropout(0.2), nn.Linear(hidden, n_classes))
        _randomise_bn(self, torch.Generator().manual_seed(seed + 1))
        _calibrate(self, torch.randn(8, dim, generator=torch.Generator().manual_seed(seed + 2)).abs(), head_std=2.5)

    def forward(self, e):
        return self.net(e)


def export_onnx(model, example, input_names, output_names, opset=17):
    """torch.onnx.export (TorchScript exporter) -> bytes; batch dimension dynamic on every input / output."""
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    keep = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto      # (needs the absent `onnx` package; no-op for these graphs)
    try:
        import warnings
        f = io.BytesIO()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(model.eval(), (example,), f, dynamo=False, opset_version=opset, input_names=input_names, output_names=output_names,
                              dynamic_axes={n: {0: "N"} for n in list(input_names) + list(output_names)}, do_constant_folding=True)
        return f.getvalue()
    finally:
        onnx_proto_utils._add_onnxscript_fn = keep


def forward_np(model, x, dtype=torch.float32):
    """torch-CPU forward pass in `dtype`; returns a tuple of numpy arrays."""
    import copy
    m = copy.deepcopy(model).eval().to(dtype)
    with torch.no_grad():
        out = m(torch.from_numpy(np.ascontiguousarray(x)).to(dtype))
    out = out if isinstance(out, tuple) else (out,)
    return tuple(o.numpy() for o in out)


# ---------------------------------------------------------------------------------------------- fixture inputs and model registry
def config2_clips(n, n_samples, sample_rate, first=0):
    """clip i = 0.5 sin(2 pi f_i t), f_i = 500 + 37 i Hz, + N(0, 0.05^2) noise, rng 1234 + i, clamp +-1."""
    t = np.arange(n_samples, dtype=np.float64) / sample_rate
    out = []
    for i in range(first, first + n):
        x = 0.5 * np.sin(2.0 * np.pi * (500.0 + 37.0 * i) * t) + np.random.default_rng(1234 + i).normal(0.0, 0.05, n_samples)
        out.append(np.clip(x, -1.0, 1.0))
    return np.asarray(out, np.float32)


def tawnyowl_clips(n_samples, golden_dir):
    """The reference's own audio fixture (tawnyowl.wav, 32-bit PCM, committed delta-coded): five 3 s windows, cut to n_samples."""
    z = np.load(golden_dir + "/tawnyowl_pcm32.npz")
    pcm = np.cumsum(z["delta"].astype(np.int64))
    pcm = ((pcm + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int32)
    x = (pcm.astype(np.float32) / np.float32(2147483648.0)).reshape(5, 144000)        # convert/pcm.go:226-268: float32(int32) / 2^31
    return np.ascontiguousarray(x[:, :n_samples])


def fixture_inputs(family, n_samples, golden_dir):
    if family == "birdnet":
        return np.concatenate([config2_clips(4, n_samples, 48000), np.zeros((1, n_samples), np.float32), tawnyowl_clips(n_samples, golden_dir)])
    if family == "perch":                                     # cmd/perch-benchmark/main.go:246-252: U[-1, 1]; plus config-2 tones at 32 kHz, silence
        u = np.random.default_rng(777).uniform(-1.0, 1.0, (2, n_samples)).astype(np.float32)
        return np.concatenate([u, config2_clips(2, n_samples, 32000), np.zeros((1, n_samples), np.float32)])
    if family == "bat":
        return np.abs(np.random.default_rng(4321).standard_normal((8, n_samples))).astype(np.float32)
    raise ValueError(family)


# name -> (family, constructor, input names, output names, index of the logits output)
MODELS = {
    "bn_tiny_stft": ("birdnet", lambda: BirdNetLike("tiny", "stft"), ["samples"], ["logits"], 0),
    "bn_tiny_conv": ("birdnet", lambda: BirdNetLike("tiny", "conv"), ["samples"], ["logits"], 0),
    "bn_tiny_emb": ("birdnet", lambda: BirdNetLike("tiny", "stft", embeddings=True), ["samples"], ["logits", "embeddings"], 0),
    "perch_tiny": ("perch", lambda: PerchLike("tiny"), ["inputs"], ["embedding", "spatial_embedding", "spectrogram", "label"], 3),
    "bat_head": ("bat", lambda: BatHead(), ["embedding"], ["scores"], 0),
    "bn_full_stft": ("birdnet", lambda: BirdNetLike("full", "stft"), ["samples"], ["logits"], 0),
    "bn_full_conv": ("birdnet", lambda: BirdNetLike("full", "conv"), ["samples"], ["logits"], 0),
    "perch_full": ("perch", lambda: PerchLike("full"), ["inputs"], ["embedding", "spatial_embedding", "spectrogram", "label"], 3),
}
COMMITTED_ONNX = ("bn_tiny_stft", "bn_tiny_emb", "perch_tiny", "bat_head")     # small enough to live in the tree; the others are re-exported on demand


def build(name):
    """-> (model, onnx bytes, example input shape): deterministic in the torch version."""
    family, ctor, ins, outs, _ = MODELS[name]
    model = ctor().eval()
    n = model.n_samples if hasattr(model, "n_samples") else 1024
    blob = export_onnx(model, torch.zeros(1, n) + 0.01, ins, outs)
    return model, blob, n
