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
