"""The blocking host-pointer entries at the size the product boundary is measured on (VERDICT r2 "Next" #1).

bnhip_predict / _pcm16 / _pcm / _topk cut a call of >= 128 clips into chunks that run on alternating contexts fed from
pinned staging (csrc/hostpipe.cpp).  The contract under test is the reference's: the caller's memory is only read before
the call returns, every output is complete on return (internal/analysis/process.go:280-295,
internal/inference/onnx/classifier.go:372-430), and the values are the ones the serial path produces.
"""
import numpy as np
import pytest

from birdnet_go_amd import host, synth_model as sm
from oracle import gofuncs as G
from oracle.interp import Interpreter

from test_parity_gpu import PROB_TOL, assert_parity


def test_hostpipe_symbols_and_option_parse():
    """CPU: the option is accepted by a plan-only model (no device touched)."""
    blob = sm.build_model(sm.tiny_config())
    clf = host.HipClassifier(blob, plan_only=True, host_depth=2)
    assert clf.num_species() > 0
    clf.close()


@pytest.mark.gpu
def test_pcm16_2048_clips_sampled_rows_vs_oracle(gpu, full_blob):
    """2048 int16 clips in ONE blocking call (8 chunks over two contexts): sampled rows vs the oracle, every chunk's rows vs
    the same clips sent in small serial calls, and a repeat of the call is bit-identical."""
    B = 2048
    x256 = sm.synth_clips(256, 144000, 48000)
    pcm256 = (np.clip(x256, -1, 1) * 32767).astype(np.int16)
    # 8 chunks with distinct content: chunk c = the 256 clips rotated by 31 * c, so a chunk landing in the wrong place shows
    pcm = np.concatenate([np.roll(pcm256, 31 * c, axis=0) for c in range(8)], axis=0)
    rows = [0, 255, 256, 777, 1023, 1024, 1800, 2047]
    xr = G.pcm_to_f32(pcm[rows].tobytes(), 16).reshape(len(rows), 144000)
    ref = Interpreter(full_blob).invoke(xr)[0]
    clf = host.HipClassifier(full_blob, max_batch=256)
    try:
        out = np.zeros((B, clf.num_species()), np.float32)
        got = clf.predict_pcm16(pcm.reshape(-1), B, out=out).copy()
        assert np.isfinite(got).all()
        assert_parity(got[rows], ref)
        assert np.abs(got[rows] - ref).max() < 1e-3
        # chunk placement: row r of chunk c equals row (r - 31 c) mod 256 of chunk 0, exactly (same kernels, same tiles)
        for c in range(1, 8):
            assert np.array_equal(got[c * 256:(c + 1) * 256], np.roll(got[:256], 31 * c, axis=0)), f"chunk {c}"
        # the serial path (calls below the pipelining threshold) agrees to the stated tolerance
        small = np.concatenate([clf.predict_pcm16(pcm[i:i + 64].reshape(-1), 64) for i in range(0, 256, 64)], axis=0)
        sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
        assert np.abs(sig(small) - sig(got[:256])).max() <= PROB_TOL
        assert (small.argmax(1) == got[:256].argmax(1)).all()
        again = clf.predict_pcm16(pcm.reshape(-1), B)
        assert np.array_equal(again, got)
    finally:
        clf.close()


@pytest.mark.gpu
def test_ragged_calls_embeddings_and_topk_through_the_pipeline(gpu, tiny_blob):
    """Chunk arithmetic on a small model: clip counts that do not divide, embeddings, float32 / 24-bit input and the fused
    top-k all through the pipelined path, each against the serial path of a second handle (host_depth 1, small calls)."""
    cfg = sm.tiny_config()
    ref_clf = host.HipClassifier(tiny_blob, max_batch=16)
    clf = host.HipClassifier(tiny_blob, max_batch=64)
    try:
        for n in (128, 129, 191, 200, 333):
            x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate)
            want = np.concatenate([ref_clf.predict_batch(x[i:i + 16].reshape(-1), min(16, n - i)) for i in range(0, n, 16)], axis=0)
            got = clf.predict_batch(x.reshape(-1), n)
            assert got.shape == want.shape and np.abs(got - want).max() < 1e-4, n
            if clf.emb_dim:
                lg, em = clf.predict_batch(x.reshape(-1), n, want_embeddings=True)
                assert np.array_equal(lg, got) and em.shape == (n, clf.emb_dim) and np.isfinite(em).all()
            conf, idx = clf.predict_topk(x.reshape(-1), n, k=5, activation=0, sensitivity=1.0)
            wc = G.sigmoid_sensitivity(got, 1.0)
            assert (idx[:, 0] == wc.argmax(1)).all() and np.abs(conf[:, 0] - wc.max(1)).max() <= 1e-6
        # 24-bit PCM through the pipeline
        n = 150
        x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate)
        i24 = np.clip(np.round(x * 8388607), -8388608, 8388607).astype(np.int32)
        raw = np.zeros((i24.size, 3), np.uint8)
        u = i24.reshape(-1).astype(np.uint32)
        raw[:, 0] = u & 0xff; raw[:, 1] = (u >> 8) & 0xff; raw[:, 2] = (u >> 16) & 0xff
        xf = G.pcm_to_f32(raw.tobytes(), 24).reshape(n, cfg.n_samples)
        want = np.concatenate([ref_clf.predict_batch(xf[i:i + 16].reshape(-1), min(16, n - i)) for i in range(0, n, 16)], axis=0)
        got = clf.predict_pcm(raw.tobytes(), 24, n)
        assert np.abs(got - want).max() < 1e-4
    finally:
        clf.close(); ref_clf.close()


@pytest.mark.gpu
def test_two_engine_handle_shards_through_the_pipeline(gpu, tiny_blob):
    """"devices":[0,0]: two engines, two worker threads, each running its shard through its own staging ring and the shared
    copy pool; results equal the single-engine handle's."""
    cfg = sm.tiny_config()
    n = 512
    x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate)
    one = host.HipClassifier(tiny_blob, max_batch=64)
    two = host.HipClassifier(tiny_blob, max_batch=64, devices=[0, 0], replicate="peer")
    try:
        a = one.predict_batch(x.reshape(-1), n)
        b = two.predict_batch(x.reshape(-1), n)
        assert np.abs(a - b).max() < 1e-4 and (a.argmax(1) == b.argmax(1)).all()
    finally:
        one.close(); two.close()


@pytest.mark.gpu
def test_two_contexts_with_different_data_equal_the_serial_results(gpu, full_blob):
    """Regression for the round-3 finding, root-caused in round 4 (profiles/r04_pk_hazard.txt): with two contexts in flight, 2-4 % of
    the clips came out slightly wrong.  gfx950 executes a packed-fp32 VALU instruction whose src1 op_sel bit is set (the SLP
    vectoriser had turned k_mel_banded's band sums into such v_pk_fma_f32 chains) wrongly - low half, lanes 48-63 - while another
    wave on the CU runs v_mfma_f32_16x16x32_bf16, i.e. beside the other context's split-bf16 GEMMs; never when run alone, and
    independent of the data (tools/ubench/pkf32_vs_bf16mfma.hip reproduces it with registers only).  Device-pointer entry, depth 2,
    eight distinct batches queued back to back, against the same engine run one call at a time: bit-identical."""
    from test_parity_gpu import _DevBuf
    B, NB = 256, 8
    x256 = sm.synth_clips(B, 144000, 48000)
    xs = [np.roll(x256, 31 * c, axis=0).copy() for c in range(NB)]
    clf = host.HipClassifier(full_blob, max_batch=B, depth=2, lanes=1)
    xd = [_DevBuf(x.nbytes) for x in xs]
    o = _DevBuf(NB * B * 6522 * 4)
    try:
        for d, x in zip(xd, xs):
            d.upload(x)
        for c in range(NB):                                        # one call at a time: nothing overlaps
            clf.predict_device(xd[c].at(0), B, o.at(c * B * 6522 * 4))
            clf.synchronize()
        ref = o.download((NB, B, 6522)).copy()
        for c in range(1, NB):                                     # (and the clips do not care where in a batch they sit)
            assert np.array_equal(ref[c], np.roll(ref[0], 31 * c, axis=0))
        for rep in range(4):
            for c in range(NB):
                clf.predict_device(xd[c].at(0), B, o.at(c * B * 6522 * 4))
            clf.synchronize()
            got = o.download((NB, B, 6522))
            bad = int((np.abs(got - ref).max(2) > 0).sum())
            assert bad == 0, f"{bad} clips differ between overlapped and serial execution (rep {rep})"
    finally:
        clf.close(); o.free()
        for d in xd:
            d.free()


@pytest.mark.gpu
def test_bf16_storage_engine_through_the_pipeline_equals_its_serial_path(gpu, full_blob):
    """A "precision":"bf16" engine keeps the expanded tensors as bf16 in HBM (engine.cpp mark_bf16_storage); every context of
    the host pipeline has its own activation arena with the same value layout, so 512 distinct clips through two contexts
    must equal the same engine's serial path (host_depth 1) bit for bit, and a repeat of the call must equal itself."""
    x256 = sm.synth_clips(256, 144000, 48000)
    pcm = np.concatenate([(np.clip(x256, -1, 1) * 32767).astype(np.int16), np.roll((np.clip(x256, -1, 1) * 32767).astype(np.int16), 17, axis=0)], axis=0)
    outs = {}
    for hd in (2, 1):
        clf = host.HipClassifier(full_blob, max_batch=256, precision="bf16", host_depth=hd, autotune=False)
        try:
            assert any(s["out_bf16"] for s in clf.describe()["steps"])
            outs[hd] = clf.predict_pcm16(pcm.reshape(-1), 512).copy()
            if hd == 2:
                assert np.array_equal(clf.predict_pcm16(pcm.reshape(-1), 512), outs[hd])
        finally:
            clf.close()
    assert np.isfinite(outs[2]).all()
    assert np.array_equal(outs[2][256:], np.roll(outs[2][:256], 17, axis=0))
    assert np.array_equal(outs[2], outs[1])


@pytest.mark.gpu
def test_pinned_caller_buffers_are_bit_identical_and_skip_the_staging(gpu, full_blob):
    """memory from bnhip_host_alloc is recognised per call; input read and logits / embeddings written by DMA
    straight from / to the caller's buffers.  Same bits as the pageable path: one clip, a serial-path batch, a pipelined call with
    ragged chunking, int16 input, pinned input with pageable output and the reverse."""
    clf = host.HipClassifier(full_blob, max_batch=256)
    try:
        n = 300                                               # pipelined (>= 128), not a multiple of the chunk unit
        x = sm.synth_clips(n, 144000, 48000)
        pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
        ref = clf.predict_batch(x.reshape(-1), n)
        ref_pcm = clf.predict_pcm16(pcm.reshape(-1), n)
        with host.PinnedArray((n, 144000), np.float32) as pi, host.PinnedArray((n, 144000), np.int16) as pp, \
                host.PinnedArray((n, clf.num_species()), np.float32) as po:
            pi.array[:] = x; pp.array[:] = pcm
            po.array[:] = np.nan
            import ctypes
            lib = host.load_library()
            lib.bnhip_debug_pinned_inputs.restype = ctypes.c_long
            before = lib.bnhip_debug_pinned_inputs()
            got = clf.predict_batch(pi.array.reshape(-1), n, out=po.array)
            assert got.ctypes.data == po.array.ctypes.data and np.array_equal(got, ref)
            assert lib.bnhip_debug_pinned_inputs() == before + 1          # recognised (whole range inside one allocation), not staged
            # pageable input is not counted (it goes through the pinned staging slots)
            assert np.array_equal(clf.predict_batch(x.reshape(-1), n), ref) and lib.bnhip_debug_pinned_inputs() == before + 1
            assert np.array_equal(clf.predict_batch(pi.array.reshape(-1), n), ref)             # pinned in, pageable out
            po.array[:] = np.nan
            assert np.array_equal(clf.predict_batch(x.reshape(-1), n, out=po.array), ref)      # pageable in, pinned out
            po.array[:] = np.nan
            assert np.array_equal(clf.predict_pcm16(pp.array.reshape(-1), n, out=po.array), ref_pcm)
            for k in (1, 7, 64):                              # the serial path (below the pipelining threshold)
                assert np.array_equal(clf.predict_batch(pi.array[:k].reshape(-1), k, out=po.array[:k]), clf.predict_batch(x[:k].reshape(-1), k))
    finally:
        clf.close()
    with pytest.raises(host.HipError):
        host.PinnedArray((0,), np.float32)


def test_plan_cut_of_the_two_phase_call_is_a_block_boundary():
    """CPU: Engine::pick_split cuts the v2.4 plan where exactly one activation crosses (a block boundary) - the input of b5, the first
    block whose input has fewer than 1000 positions per clip; BNHIP_HOST_SPLIT=-1 (read at plan time) disables it."""
    import os
    blob = sm.build_model()
    clf = host.HipClassifier(blob, plan_only=True)
    d = clf.describe()
    clf.close()
    k = d["split_step"]
    assert k > 0 and d["steps"][k]["name"] == "b5/expand+dw" and d["steps"][k - 1]["name"] == "b4/project"
    os.environ["BNHIP_HOST_SPLIT"] = "-1"
    try:
        clf = host.HipClassifier(blob, plan_only=True)
        assert clf.describe()["split_step"] == -1
        clf.close()
    finally:
        os.environ.pop("BNHIP_HOST_SPLIT")


@pytest.mark.gpu
def test_two_phase_call_on_the_tiny_model_embeddings_and_24_bit(gpu, tiny_blob):
    """The same contract on the small model (cut right behind the stem, embeddings out of the back phase, 24-bit PCM through the fronts):
    two-phase == whole-plan chunks, bit for bit, at clip counts that leave ragged chunks."""
    import ctypes
    import os
    lib = host.load_library()
    lib.bnhip_debug_split_calls.restype = ctypes.c_long
    cfg = sm.tiny_config()
    clf = host.HipClassifier(tiny_blob, max_batch=256)
    try:
        assert clf.describe()["split_step"] > 0
        #  the per-call diagnostic switches only exist in a process that started with BNHIP_HOST_DIAG (conftest sets it):
        # without it the NOSPLIT leg below would silently measure the split path twice
        assert os.environ.get("BNHIP_HOST_DIAG"), "tests/conftest.py must export BNHIP_HOST_DIAG before the library is loaded"
        for n in (129, 191, 256):
            x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate)
            i24 = np.clip(np.round(x * 8388607), -8388608, 8388607).astype(np.int32)
            raw = np.zeros((i24.size, 3), np.uint8)
            u = i24.reshape(-1).astype(np.uint32)
            raw[:, 0] = u & 0xff; raw[:, 1] = (u >> 8) & 0xff; raw[:, 2] = (u >> 16) & 0xff
            res = {}
            for env in (None, "1"):
                if env:
                    os.environ["BNHIP_HOST_NOSPLIT"] = env
                try:
                    before = lib.bnhip_debug_split_calls()
                    a = clf.predict_batch(x.reshape(-1), n, want_embeddings=True) if clf.emb_dim else (clf.predict_batch(x.reshape(-1), n).copy(),)
                    b = clf.predict_pcm(raw.tobytes(), 24, n).copy()
                    assert lib.bnhip_debug_split_calls() - before == (0 if env else 2), (n, env)
                    res[env] = tuple(np.array(v) for v in a) + (b,)
                finally:
                    os.environ.pop("BNHIP_HOST_NOSPLIT", None)
            for u_, v_ in zip(res[None], res["1"]):
                assert np.isfinite(u_).all() and np.array_equal(u_, v_), n
    finally:
        clf.close()


@pytest.mark.gpu
def test_two_phase_call_equals_whole_plan_chunks_bit_for_bit(gpu, full_blob):
    """A blocking call that fits one batch cuts the PLAN as well as the batch (hostpipe.cpp host_run_split): fronts per chunk, backs over
    groups of chunks, the crossing activation through hand-off memory.  Same kernels, same per-clip arithmetic: logits, embeddings and
    top-k equal the whole-plan chunk schedule (BNHIP_HOST_NOSPLIT, read per call) bit for bit - float32 and int16 input, pageable and
    page-locked, clip counts that do not divide, fp32 and bf16-storage engines - and the path is the one taken (counter)."""
    import ctypes
    import os
    lib = host.load_library()
    lib.bnhip_debug_split_calls.restype = ctypes.c_long
    x = sm.synth_clips(256, 144000, 48000)
    pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
    for kw in ({}, {"precision": "bf16", "autotune": False}):
        clf = host.HipClassifier(full_blob, max_batch=256, **kw)
        try:
            assert clf.describe()["split_step"] > 0
            for n in (256, 200, 129):
                got, ref = {}, {}
                for dst, env in ((got, None), (ref, "1")):
                    if env:
                        os.environ["BNHIP_HOST_NOSPLIT"] = env
                    try:
                        before = lib.bnhip_debug_split_calls()
                        if clf.emb_dim:
                            dst["f32"], dst["emb"] = clf.predict_batch(x[:n].reshape(-1), n, want_embeddings=True)
                        else:
                            dst["f32"] = clf.predict_batch(x[:n].reshape(-1), n).copy()
                        dst["pcm"] = clf.predict_pcm16(pcm[:n].reshape(-1), n).copy()
                        dst["topk"] = clf.predict_topk(x[:n].reshape(-1), n, k=10, activation=0, sensitivity=1.0)
                        assert lib.bnhip_debug_split_calls() - before == (0 if env else 3), (n, env)
                    finally:
                        os.environ.pop("BNHIP_HOST_NOSPLIT", None)
                assert np.isfinite(got["f32"]).all()
                assert np.array_equal(got["f32"], ref["f32"]) and np.array_equal(got["pcm"], ref["pcm"]), (n, kw)
                if "emb" in got:
                    assert np.array_equal(got["emb"], ref["emb"]), (n, kw)
                assert np.array_equal(got["topk"][0], ref["topk"][0]) and np.array_equal(got["topk"][1], ref["topk"][1]), (n, kw)
            if not kw:
                with host.PinnedArray((256, 144000), np.float32) as pi, host.PinnedArray((256, clf.num_species()), np.float32) as po:
                    pi.array[:] = x
                    assert np.array_equal(clf.predict_batch(pi.array.reshape(-1), 256, out=po.array), got["f32"] if n == 256 else clf.predict_batch(x.reshape(-1), 256))
                # the oracle on sampled rows of the two-phase call
                rows = [0, 63, 64, 191, 192, 255]
                full = clf.predict_batch(x.reshape(-1), 256)
                assert_parity(full[rows], Interpreter(full_blob).invoke(x[rows])[0])
        finally:
            clf.close()
