"""C-ABI surface, CPU only: the library loads, exports every symbol include/bnhip.h declares, and
fails loudly (never falls back) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from birdnet_go_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "bnhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bnhip_[a-z0-9_]+)\s*\(", src)))


def test_header_and_library_agree(built_lib):
    lib = C.CDLL(built_lib)
    fns = header_functions()
    assert len(fns) >= 18
    for f in fns:
        assert hasattr(lib, f), f"libbnhip.so does not export {f}"
    assert sorted(host.SYMBOLS) == fns, "host.py SYMBOLS out of sync with include/bnhip.h"


def test_no_torch_or_cxx_types_in_header():
    src = open(os.path.join(ROOT, "include", "bnhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)     # declarations only
    assert "std::" not in src and "torch" not in src and "at::" not in src and "hip" not in src.replace("bnhip", "").replace("hip_stream", "")


def test_version_and_error_strings(built_lib):
    lib = host.load_library()
    assert b"gfx950" in lib.bnhip_version()
    assert isinstance(lib.bnhip_last_error(), bytes)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_gpu_is_a_loud_sentinel_error(built_lib, tiny_blob):
    with pytest.raises(host.ErrHIPUnavailable) as e:
        host.HipClassifier(tiny_blob)
    assert e.value.code == host.E_NO_DEVICE
    with pytest.raises(host.ErrHIPUnavailable):
        host.init()
    with pytest.raises(host.HipError):
        host.us_frame_cv(np.zeros((1, 144000)), 256000)


def test_invalid_arguments(built_lib):
    lib = host.load_library()
    out = C.c_void_p()
    assert lib.bnhip_model_create(None, 0, None, C.byref(out)) == host.E_INVALID
    assert lib.bnhip_model_info(None, None, None, None) == host.E_INVALID
    assert lib.bnhip_predict(None, None, 1, None, None) == host.E_INVALID
    lib.bnhip_model_destroy(None)   # idempotent / NULL-safe like Close()


def test_window_entries_reject_null_and_nonsense(built_lib, tiny_blob):
    """Every bnhip_windows_* entry with NULL handles / outputs, negative or unknown sources, a negative cap; the tick entry with a
    plan-only model, a bit depth it does not know and a window size that is not the model's clip - all BNHIP_E_INVALID with a
    message, nothing crashes, nothing is consumed."""
    from birdnet_go_amd import stream as S
    lib = host.load_library()
    w = S.NativeWindows(4, 8, max_batch=2)                  # (sets the prototypes)
    h = w._h
    vp, ci = C.c_void_p, C.c_int
    n, src, p = ci(7), (ci * 4)(), vp()
    assert lib.bnhip_windows_create(4, 8, 2, None) == host.E_INVALID
    assert lib.bnhip_windows_info(None, None, None, None, None) == host.E_INVALID
    assert lib.bnhip_windows_info(h, None, None, None, None) == host.BNHIP_OK       # every output is optional
    assert lib.bnhip_windows_add_source(None, b"mic", 64, C.byref(n)) == host.E_INVALID
    assert lib.bnhip_windows_add_source(h, b"mic", 64, None) == host.E_INVALID
    assert lib.bnhip_windows_add_source(h, None, 64, C.byref(n)) == host.E_INVALID and n.value == -1
    assert lib.bnhip_windows_write(None, 0, b"ab", 2) == host.E_INVALID
    assert lib.bnhip_windows_write(h, -1, b"ab", 2) == host.E_INVALID
    assert lib.bnhip_windows_write(h, 0, None, 2) == host.E_INVALID
    assert lib.bnhip_windows_remove_source(h, -3) == host.E_INVALID and lib.bnhip_windows_reset(h, 9) == host.E_INVALID
    assert lib.bnhip_windows_stats(h, 0, None, None, None) == host.E_INVALID           # no such source yet
    assert lib.bnhip_windows_collect(h, -1, src, C.byref(n), C.byref(p)) == host.E_INVALID and n.value == 0
    assert lib.bnhip_windows_collect(h, 2, None, C.byref(n), None) == host.E_INVALID
    assert lib.bnhip_windows_collect(None, 2, src, C.byref(n), None) == host.E_INVALID
    assert lib.bnhip_windows_ready(h, None) == host.E_INVALID
    i = w.add_source("mic", 32)
    w.write(i, bytes(range(8)))
    assert lib.bnhip_windows_write(h, i, None, 0) == host.BNHIP_OK                   # an empty write is a write (counted)
    assert w.stats(i) == (2, 0, 8)
    # the tick entry: arguments are checked before anything is read from the rings
    conf, idx = (C.c_float * 20)(), (C.c_int32 * 20)()
    m = host.HipClassifier(tiny_blob, plan_only=True)
    args = lambda bits, k=10: (h, m._h, bits, 0, 1.0, k, src, C.byref(n), conf, idx, None)
    assert lib.bnhip_windows_predict_topk(None, m._h, 16, 0, 1.0, 10, src, C.byref(n), conf, idx, None) == host.E_INVALID
    assert lib.bnhip_windows_predict_topk(h, None, 16, 0, 1.0, 10, src, C.byref(n), conf, idx, None) == host.E_INVALID
    assert lib.bnhip_windows_predict_topk(*args(12)) == host.E_INVALID and b"bit depth" in lib.bnhip_last_error()
    assert lib.bnhip_windows_predict_topk(*args(16, 0)) == host.E_INVALID
    assert lib.bnhip_windows_predict_topk(*args(16)) == host.E_INVALID and b"plan-only" in lib.bnhip_last_error()
    assert w.stats(i) == (2, 0, 8) and w.ready() == 1                               # still there
    m.close()
    lib.bnhip_windows_destroy(None)
    w.close()


def test_us_guards_need_no_gpu(built_lib):
    # the reference's guard clauses (filter.go:21-37) answer before any device work
    cv, ok = host.us_frame_cv(np.zeros((2, 100)), 256000)
    assert not ok.any() and (cv == 0).all()
    cv, ok = host.us_frame_cv(np.zeros((1, 20000)), 48000, split_hz=30000)
    assert not ok.any()
    cv, ok = host.us_frame_cv(np.zeros((1, 20000)), 256000, fft_size=6000)
    assert not ok.any()
    cv, ok = host.us_frame_cv(np.zeros((1, 8192)), 256000)
    assert not ok.any()


# ------------------------------------------------------------------------------------------------ the cgo preamble, compiled
GO_SHIM = os.path.join(ROOT, "birdnet-go_amd", "go", "internal", "inference", "hip", "backend_hip.go")


def extract_preamble():
    """The C preamble of the cgo file, verbatim (everything between the `/*` that follows `package hip` and `*/ import "C"`)."""
    src = open(GO_SHIM).read()
    m = re.search(r"package hip\s*/\*(.*?)\*/\s*import \"C\"", src, flags=re.S)
    assert m, "cgo preamble not found"
    return m.group(1)


@pytest.fixture(scope="module")
def cabi_driver(tmp_path_factory, built_lib):
    import subprocess
    d = tmp_path_factory.mktemp("cabi")
    pre = "\n".join(l for l in extract_preamble().splitlines() if not l.startswith("#cgo"))
    (d / "preamble_extracted.h").write_text(pre)
    exe = str(d / "cabi_driver")
    # gcc, C11, warnings are errors: `snprintf` before <stdio.h> would stop the build here
    subprocess.check_call(["gcc", "-std=c11", "-D_DEFAULT_SOURCE", "-Wall", "-Wextra", "-Werror", "-O1", "-I", str(d),
                           os.path.join(ROOT, "tests", "native", "cabi_driver.c"), "-o", exe, "-ldl", "-lm"])
    return exe


def test_cgo_preamble_compiles_and_go_call_sequence_cpu(cabi_driver, built_lib, tmp_path):
    """The Go shim's C half under -Wall -Wextra -Werror, and its call sequence's error paths on the CPU: missing library,
    garbage / truncated model, invalid option, plan-only handle, predict rejected with a message, unload + retry."""
    import subprocess
    from birdnet_go_amd import synth_model as sm
    model = tmp_path / "dense.tflite"
    model.write_bytes(sm.build_dense_model([64, 16, 5]))
    r = subprocess.run([cabi_driver, built_lib, str(model), "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "cpu sequence ok" in r.stdout


def test_go_shim_pins_the_os_thread_around_error_fetch():
    src = open(GO_SHIM).read()
    # every exported entry that can fail fetches the thread-local error text: each must hold the OS thread
    for fn in ("func Init(", "func NewClassifierWithOptions(", "func (c *Classifier) predict(", "func (c *Classifier) PredictBatch(",
               "func (c *Classifier) predictTopK(", "func (c *Classifier) PredictPCM16(", "func ComputeUSFrameCV(", "func NewResampler(",
               "func (r *Resampler) ResampleTo(", "func (r *Resampler) Flush(", "func NewWindowAssembler(", "func (w *WindowAssembler) AddSource(",
               "func (w *WindowAssembler) Write(", "func (w *WindowAssembler) collectLocked(", "func (c *Classifier) PredictWindows(", "func (c *Classifier) PredictWindowsTopK("):
        body = src[src.index(fn):]
        body = body[:body.index("\n}\n")]
        assert "runtime.LockOSThread()" in body and "defer runtime.UnlockOSThread()" in body, fn
    assert src.index("#include <stdio.h>") < src.index("snprintf")
    # a tick cannot race Close (ticks exclude each other, Close waits for calls in flight), and a failed device call
    # still tells the host which sources gave up a window
    close_body = src[src.index("func (w *WindowAssembler) Close("):]
    close_body = close_body[:close_body.index("\n}\n")]
    assert "w.tick.Lock()" in close_body and "w.life.Lock()" in close_body
    for fn in ("func (w *WindowAssembler) Collect(", "func (c *Classifier) PredictWindows(", "func (c *Classifier) PredictWindowsTopK("):
        body = src[src.index(fn):]
        body = body[:body.index("\n}\n")]
        assert "w.tick.Lock()" in body and "w.life.RLock()" in body, fn
    topk = src[src.index("func (c *Classifier) PredictWindowsTopK("):]
    assert "return sources, nil, nil, nil, err" in topk[:topk.index("\n}\n")]


@pytest.mark.gpu
def test_go_call_sequence_on_gpu(cabi_driver, built_lib, tmp_path, tiny_blob, tiny_cfg, gpu):
    """Init -> NewClassifier -> Predict -> PredictBatch -> PredictTopK -> Close from a plain C host, results vs the oracle."""
    import subprocess
    from birdnet_go_amd import synth_model as sm
    from oracle.interp import Interpreter
    model = tmp_path / "tiny.tflite"
    model.write_bytes(tiny_blob)
    x = sm.synth_clips(5, tiny_cfg.n_samples, tiny_cfg.sample_rate)
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    r = subprocess.run([cabi_driver, built_lib, str(model), "gpu", str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), "5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(5, -1)
    ref = Interpreter(tiny_blob).invoke(x)[0]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
    assert (got.argmax(1) == ref.argmax(1)).all() and np.abs(sig(got) - sig(ref)).max() <= 1e-4


def test_go_shim_implements_every_backend_interface():
    """internal/inference/backend.go declares five interfaces; the shim must carry every method of each (names and arity as in
    the reference), plus the ultrasonic gate and the resampler the audio path calls."""
    src = open(GO_SHIM).read()
    want = {
        "Classifier": ["Predict(samples []float32) ([]float32, error)", "NumSpecies() int", "Close()"],
        "EmbeddingExtractor": ["PredictWithEmbeddings(samples []float32) (logits, embeddings []float32, err error)"],
        "CustomClassifier": ["PredictEmbedding(embeddings []float32) ([]float32, error)", "NumClasses() int", "InputDim() int", "Labels() []string", "Close()"],
        "RangeFilter": ["Predict(latitude, longitude, week float32) ([]float32, error)", "NumSpecies() int", "Close()",
                        "PredictBatch(inputs []float32, batchSize int) ([]float32, error)"],
    }
    recv = {"Classifier": "c *Classifier", "EmbeddingExtractor": "c *Classifier", "CustomClassifier": "cc *CustomClassifier", "RangeFilter": "r *RangeFilter"}
    for iface, methods in want.items():
        for m in methods:
            assert f"func ({recv[iface]}) {m}" in src, (iface, m)
    for m in ("EstimateOutputBytes(inputBytes int) int", "ResampleTo(input, dst []byte) (int, error)", "ResampleInto(input []byte) ([]byte, error)",
              "FromRate() int", "ToRate() int", "Close() error", "String() string"):
        assert f"func (r *Resampler) {m}" in src, m
    assert "func NewResampler(fromRate, toRate int" in src and "func ResampleBytes(pcm []byte, fromRate, toRate int" in src
    assert "func ComputeUSFrameCV(samples []float64, sampleRate int, cfg USFilterConfig" in src
    assert "StrictF32 bool" in src and '`,"bf16x3":0`' in src
    # the real-time window path (rows a3 / a4): the assembler's methods, and the same surface in the no-tag stub
    stub = open(os.path.join(os.path.dirname(GO_SHIM), "stub_nohip.go")).read()
    for m in ("AddSource(", "RemoveSource(", "Write(", "Collect(", "OverwriteStats(", "Reset(", "WindowBytes(", "Pinned(", "Close("):
        assert f"func (w *WindowAssembler) {m}" in src and f"func (*WindowAssembler) {m}" in stub, m
    assert "func (c *Classifier) PredictWindows(w *WindowAssembler) (sources []int, windows []byte, logits []float32, err error)" in src
    assert "func (*Classifier) PredictWindows(*WindowAssembler) ([]int, []byte, []float32, error)" in stub
    assert "func (c *Classifier) PredictWindowsTopK(w *WindowAssembler, k int, sensitivity float64)" in src
    assert "func (*Classifier) PredictWindowsTopK(*WindowAssembler, int, float64) ([]int, []byte, []float32, []int32, error)" in stub


@pytest.mark.gpu
def test_go_call_sequence_with_a_range_filter_model(cabi_driver, built_lib, tmp_path, gpu):
    """RangeFilter.PredictBatch / CustomClassifier.PredictEmbedding are the batch entry on a dense model: the same C sequence
    with the [lat, lon, week] meta-model (fp16 weights behind DEQUANTIZE, like the reference's MData file), vs the oracle."""
    import subprocess
    from birdnet_go_amd import synth_model as sm
    from oracle.interp import Interpreter
    blob = sm.build_dense_model([3, 32, 16, 9], final_sigmoid=True, fp16_weights=True, input_scale=[90.0, 180.0, 48.0])
    model = tmp_path / "rf.tflite"
    model.write_bytes(blob)
    rng = np.random.default_rng(4)
    x = np.stack([rng.uniform(-90, 90, 7), rng.uniform(-180, 180, 7), rng.integers(1, 49, 7)], 1).astype(np.float32)
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    r = subprocess.run([cabi_driver, built_lib, str(model), "gpu", str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), "7"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(7, -1)
    ref = Interpreter(blob).invoke(x)[0]
    assert np.abs(got - ref).max() <= 2e-5
