"""The same MelSpec arithmetic in the other op forms a TF -> TFLite conversion may emit (the real v2.4 file is absent, so the
recogniser should not depend on one spelling): `x + (-0.5)` for `x - 0.5`, tf.tensordot as BATCH_MATMUL (either operand
orientation) instead of FULLY_CONNECTED, the square as SQUARE or MUL(x, x) instead of POW(x, 2).  Each form must plan onto
the fused front-end kernels and match the oracle, and all forms must agree with the canonical one bit for bit on the GPU
(same kernels, same constants)."""
import numpy as np
import pytest

from birdnet_go_amd import host, synth_model as sm
from oracle.interp import Interpreter

FORMS = [("add_neg",), ("bmm",), ("bmm_adj", "square"), ("mul_self", "add_neg"), ("square", "bmm")]


@pytest.mark.parametrize("forms", FORMS)
def test_front_end_op_forms_plan_fused(built_lib, forms):
    for mode in ("real", "abs"):
        cfg = sm.tiny_config(fe_forms=forms, complex_mode=mode, specs=(sm.SpecConfig(512, 94, 0.0, 3000.0), sm.SpecConfig(512, 94, 500.0, 15000.0)))
        c = host.HipClassifier(sm.build_model(cfg), plan_only=True)
        try:
            kinds = [s["kernel"] for s in c.describe()["steps"]]
            assert kinds[:5] == ["clip_minmax", "frontend", "stft", "stft", "frontend"], kinds[:6]
            assert not any(k.startswith("generic") or k == "elementwise" for k in kinds)
        finally:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("forms", FORMS)
def test_front_end_op_forms_vs_oracle(gpu, forms):
    specs = (sm.SpecConfig(512, 94, 0.0, 3000.0), sm.SpecConfig(512, 94, 500.0, 15000.0))
    x = sm.synth_clips(4, 12000, 48000)
    x[2] = 0.0
    outs = {}
    for f in ((), forms):
        blob = sm.build_model(sm.tiny_config(fe_forms=f, specs=specs))
        ref = Interpreter(blob).invoke(x)[0]
        c = host.HipClassifier(blob, max_batch=8, autotune=False)
        try:
            outs[f] = c.predict_batch(x.reshape(-1), 4)
        finally:
            c.close()
        assert (outs[f].argmax(1) == ref.argmax(1)).all() and np.abs(outs[f] - ref).max() < 1e-3
    assert np.array_equal(outs[()], outs[forms])


SE_FORMS = ["keras", "dense", "avgpool"]


@pytest.mark.parametrize("form", SE_FORMS)
def test_squeeze_excite_spellings_fuse(built_lib, form):
    """GlobalAveragePooling2D + Reshape(1,1,C) in front of the 1x1 convolutions (Keras), or Dense layers with a fused RELU and a
    RESHAPE in front of the MUL (MobileNet-style): both must land on the fused squeeze-excite kernel, scale folded into the
    projection's operand load, exactly like the canonical MEAN(keep_dims) -> CONV_2D form."""
    ref_kinds = None
    for f in ("conv", form):
        c = host.HipClassifier(sm.build_model(sm.tiny_config(se_form=f)), plan_only=True)
        try:
            st = c.describe()["steps"]
            kinds = [s["kernel"] for s in st]
            assert kinds.count("se") == 5 and "elementwise" not in kinds and not any(k.startswith("generic") for k in kinds)
            assert sum(s["fused_scale"] for s in st if s["kernel"] == "pw_gemm") == 5
            ref_kinds = ref_kinds or kinds
            assert kinds == ref_kinds
        finally:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", SE_FORMS)
def test_squeeze_excite_spellings_vs_oracle(gpu, form):
    cfg = sm.tiny_config(se_form=form)
    blob = sm.build_model(cfg)
    x = sm.synth_clips(5, cfg.n_samples, cfg.sample_rate)
    ref = Interpreter(blob).invoke(x)[0]
    for kw in (dict(), dict(autotune=False, lanes=1)):
        c = host.HipClassifier(blob, max_batch=8, **kw)
        try:
            got = c.predict_batch(x.reshape(-1), 5)
        finally:
            c.close()
        assert (got.argmax(1) == ref.argmax(1)).all() and np.abs(got - ref).max() < 1e-3


@pytest.mark.gpu
def test_fused_mel_epilogue_vs_the_separate_banded_kernel(gpu, full_blob):
    """k_stft_bins<.., MEL> (mel projection + compression in the wave that transformed the frame; no bins tensor, no
    k_mel_banded launch) against the separate kernel: the same products, summed per aligned quad and then per band instead of
    in one running chain, so the logits agree to fp32 rounding - on the v2.4 layer (two channels, real part, power
    compression), on silence (where the power-law compression amplifies everything), and on the log-mel / magnitude /
    time-major variant.  Both forms are also held to the oracle."""
    import os
    from birdnet_go_amd import host, synth_model as sm
    for blob, n, rate in ((full_blob, 144000, 48000), (sm.build_model(sm.tiny_perch_config()), 8000, 32000)):
        x = sm.synth_clips(4, n, rate)
        x[1] = 0.0
        ref = Interpreter(blob).invoke(x)
        ref = ref[3] if len(ref) == 4 else ref[0]
        plain = host.HipClassifier(blob, max_batch=4, autotune=False)
        os.environ["BNHIP_FUSE_MEL"] = "1"                     # opt-in: measured slower than the separate kernel
        try:
            fused = host.HipClassifier(blob, max_batch=4, autotune=False)
        finally:
            del os.environ["BNHIP_FUSE_MEL"]
        try:
            kf = [s["name"] for s in fused.describe()["steps"]]
            kp = [s["name"] for s in plain.describe()["steps"]]
            assert any(k.endswith("+mel") for k in kf) and not any(k.startswith("melband") for k in kf)
            assert any(k.startswith("melband") for k in kp)
            a, b = fused.predict_batch(x.reshape(-1), 4), plain.predict_batch(x.reshape(-1), 4)
            assert np.isfinite(a).all() and np.abs(a - b).max() < 1e-4
            for got in (a, b):
                assert (got.argmax(1) == ref.argmax(1)).all() and np.abs(got - ref).max() < 1e-3
        finally:
            fused.close(); plain.close()


@pytest.mark.gpu
def test_output_pruned_stft_is_bit_identical_to_the_full_transform(gpu, full_blob, monkeypatch):
    """VERDICT r4 #3, the output-pruned STFT: the mel bank of the v2.4 front-end reads 129 of the 2048-point transform's 1 025 bins,
    so the closing 4-point stage of k_stft_bins<16> computes and stores only the outputs a needed bin reads (StftParams::zmask, from
    the plan).  Every kept output is the same chain of additions as in the full stage: the logits must not move by one bit - on
    material, on silence (the case the fp64 transform exists for) and with the fused mel epilogue."""
    from birdnet_go_amd import host, synth_model as sm
    x = sm.synth_clips(6, 144000, 48000, first=31)
    x[2] = 0.0
    for fuse in ("0", "1"):
        monkeypatch.setenv("BNHIP_FUSE_MEL", fuse)
        out = {}
        for prune in ("1", "0"):
            monkeypatch.setenv("BNHIP_STFT_PRUNE", prune)
            c = host.HipClassifier(full_blob, max_batch=8, autotune=False)
            try:
                out[prune] = c.predict_batch(x.reshape(-1), 6).copy()
            finally:
                c.close()
        assert np.isfinite(out["1"]).all() and np.array_equal(out["1"], out["0"]), np.abs(out["1"] - out["0"]).max()
