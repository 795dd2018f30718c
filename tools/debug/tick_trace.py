"""Where one tick of the real-time path goes: 256 sources -> bnhip_windows_predict_topk, beside the two-step
collect + bnhip_predict_pcm_topk and the plain pinned int16 calls, p50 / p95 over 30 repetitions each; BNHIP_HOST_TRACE=1 in the
environment adds the per-chunk GPU timeline of every call to stderr.  Run on the GPU box: python tools/debug/tick_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, stream as S, synth_model as sm

blob = sm.build_model()
x = sm.synth_clips(256, 144000, 48000)
pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
clf = host.HipClassifier(blob, max_batch=256)
ncls = clf.num_species()
clip_b = 144000 * 2
ovb, rdb = clip_b // 2, clip_b - clip_b // 2
win = S.NativeWindows(ovb, rdb, 256)
ids = [win.add_source(f"s{i}", 2 * clip_b) for i in range(256)]
fresh = pcm.view(np.uint8).reshape(256, -1)[:, :rdb]
R = int(os.environ.get("REPS", "30"))


def stats(name, ts):
    a = np.asarray(ts) * 1e3
    print(f"{name:34s} p50 {np.percentile(a, 50):7.3f}  p95 {np.percentile(a, 95):7.3f}  min {a.min():7.3f}  max {a.max():7.3f} ms", flush=True)


def timed(fn, pre=None):
    ts = []
    for _ in range(R + 2):
        if pre:
            pre()
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return ts[2:]


def write_all():
    for i in ids:
        win.write(i, fresh[i])


with host.PinnedArray((256, 144000), np.int16) as pp, host.PinnedArray((256, ncls), np.float32) as po:
    pp.array[:] = pcm
    stats("predict_pcm16 pinned -> logits", timed(lambda: clf.predict_pcm16(pp.array.reshape(-1), 256, out=po.array)))
    stats("predict_pcm_topk pinned", timed(lambda: clf.predict_pcm_topk(pp.array.reshape(-1), 16, 256, 10, 0, 1.0)))
    stats("predict_pcm_topk pageable", timed(lambda: clf.predict_pcm_topk(pcm.reshape(-1), 16, 256, 10, 0, 1.0)))
stats("capture-side writes (256)", timed(write_all, pre=lambda: win.collect()))
win.collect()
stats("collect alone", timed(lambda: win.collect(), pre=write_all))
rows = [None]
def two_step():
    idxs, r = win.collect()
    clf.predict_pcm_topk(r.reshape(-1), 16, len(idxs), 10, 0, 1.0)
stats("two-step: collect + predict", timed(two_step, pre=write_all))
stats("one call: windows_predict_topk", timed(lambda: win.predict_topk(clf, 16, 10, 0, 1.0), pre=write_all))
win.close(); clf.close()
