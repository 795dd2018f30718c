#!/usr/bin/env python
"""Throughput bench for the BirdNET v2.4 hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one rank
per GPU).  A *step* = one pass of the whole hot path (per-clip normalise -> fused mel front-end ->
CNN -> species head, raw logits out) over one batch of synthetic 3 s / 48 kHz clips per GPU: at N = 1 the
256 clips of BASELINE.json configs[1] ("BirdNET v2.4 fp32, batch 256x3s@48kHz synthetic sine+noise"), at
N > 1 the 1024 clips per GPU of configs[2] ("batch 8192 clips sharded 8xMI355X"; a `weak_256` leg keeps
the 256-per-GPU figure comparable with the N = 1 line, `host_pointer_per_rank` measures the PCIe- and
host-staging-inclusive rate with every rank ingesting at once).  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0 with `roofline` (dominant
kernel, measured with HIP events on the launch stream inside the timed region) and `cpu_baseline`
(the numpy/BLAS oracle restatement timed on this box's host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF headline is 2:1 sparse)
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_F64_MFMA_TFLOPS = 78.6       # f64 MFMA: half the f32-input rate (v_mfma_f64_16x16x4_f64, 64 cycles)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU per step (0: the workload's BASELINE value: 256 at N = 1 / 1024 at N > 1 (configs[1] / [2]); Perch 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--input-sets", type=int, default=4, help="distinct resident input batches rotated over the steps")
    ap.add_argument("--bf16x3", type=int, default=None, help="split-bf16 MFMA path for the pointwise layers: 0 off, 1 where the "
                    "autotuner measures it faster, 2 everywhere eligible (default: the library's)")
    ap.add_argument("--precision", default=None, choices=["f32", "bf16"], help="engine option \"precision\": bf16 = MFMA operands rounded "
                    "to bf16, fp32 accumulate/storage (BASELINE configs[4] quotes Perch on bf16); default f32")
    ap.add_argument("--workload", default="birdnet", choices=["birdnet", "bat", "perch"],
                    help="birdnet = BASELINE configs[1] (the contract's line); bat = configs[3]: BattyBirdNET pipeline on 256 kHz "
                         "material (ultrasonic frame-CV gate + backbone embeddings + ONNX regional head), an extra line")
    ap.add_argument("--no-fp32-run", action="store_true", help="skip the secondary (untimed-by-contract) run with bf16x3 = 0")
    ap.add_argument("--no-oracle-check", action="store_true", help="skip the max-abs probability diff vs the oracle (3 rows, outside the timed region)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="duration of the CPU baseline's throughput loop")
    ap.add_argument("--no-profile", action="store_true", help="disable per-kernel HIP-event timing")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bat / Perch legs attached to the headline line as `secondary`")
    ap.add_argument("--no-distribution", action="store_true", help="skip the per-step distribution run (>= 40 extra steps)")
    ap.add_argument("--no-host-pointer", action="store_true", help="skip the host_pointer leg (rates through the blocking host-pointer entries)")
    ap.add_argument("--detail", action="store_true", help="print a per-launch table to stderr")
    ap.add_argument("--depth", type=int, default=2, help="engine pipeline depth: successive batches run on alternating "
                    "contexts (own stream + activation arena) so one batch's tail overlaps the next one's head; 1 = off "
                    "(then each batch is split over two concurrent lanes instead)")
    return ap.parse_args()


def lib_digest():
    """Content digest of the loaded library's sources + flags (birdnet-go_amd/build.py writes it beside the .so)."""
    try:
        from birdnet_go_amd import host
        return open(os.path.join(os.path.dirname(host.LIB_PATH), ".build_digest")).read().strip()
    except Exception:
        return None


def plan_signature(desc):
    """What the create-time tuners decided, as one hash: per step (name, tiles, kernel flavour, fused-kernel shape, staged depthwise,
    split-bf16 phase 1) - the same decisions a BNHIP_TUNE_FILE records."""
    import hashlib
    h = hashlib.sha256()
    for s_ in desc["steps"]:
        h.update(repr((s_["name"], s_["nt"], s_["wm"], s_["nt_full"], s_["wm_full"], s_["shape"], s_["dw_lds"], s_["bx"])).encode())
    return h.hexdigest()[:16]


def _bound_evidence(pattern, workload, cur_digest, cur_plan):
    """Newest committed counter file of a kind, IF it was collected on the library that is loaded now (tools/profile_round.sh stamps
    every counter file with the library digest, the tune file's hash and the plan signature of its passes).  Counters cannot be
    collected inside the timed run, so the line quotes them from a file - a file from another build would put stale counters under a
    fresh headline, so it is dropped with a note instead."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", pattern))
                   if ("perch" in os.path.basename(f)) == (workload == "perch"))
    if not files:
        return None, None, {"dropped": "no committed counter file"}
    path = files[-1]
    data = json.load(open(path))
    bind = data.get("_binding")
    rel = os.path.relpath(path, ROOT)
    if not bind or not bind.get("lib_digest"):
        return None, rel, {"dropped": f"{rel} carries no library digest (collected before round 5): not quoted"}
    if not cur_digest or bind["lib_digest"] != cur_digest:
        return None, rel, {"dropped": f"{rel} was collected on library {str(bind['lib_digest'])[:12]}, this run loaded {str(cur_digest)[:12]}: not quoted"}
    note = {"lib_digest": cur_digest[:16], "tune_sha256": (bind.get("tune_sha256") or "")[:16], "plan_signature_of_the_passes": bind.get("plan_signature"),
            "plan_matches_this_run": bind.get("plan_signature") == cur_plan}
    return data, rel, note


def pmc_traffic(kclass, workload="birdnet", cur_digest=None, cur_plan=None):
    """HBM bytes per launch of a kernel class from the newest committed PMC pass (profiles/rNN_traffic.json, produced by
    tools/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench; FETCH_SIZE
    doubled per the gfx950 correction in MI355X_MICROARCH.md), bound to the loaded library (see _bound_evidence)."""
    names = {"expand_dw": "k_expand_dw", "pw_gemm": "k_pw_", "frontend": "k_frontend", "dwconv": "k_dwconv",
             "conv_direct": "k_conv_direct", "se": "k_se", "stft": "k_stft"}
    pref = names.get(kclass)
    if not pref:
        return None
    data, rel, note = _bound_evidence("r*_traffic.json", workload, cur_digest, cur_plan)
    if data is None:
        return {"bytes_per_launch": None, "source": rel, "binding": note}
    tot = n = 0
    for k, v in data.items():
        if k.startswith(pref):
            tot += v["hbm_bytes_per_launch"] * v["dispatches"]
            n += v["dispatches"]
    if not n:
        return None
    return {"bytes_per_launch": tot / n, "source": rel, "binding": note}


def pmc_mfma_util(workload="birdnet", cur_digest=None, cur_plan=None):
    """BASELINE.md section 4's figure: MFMA-pipe utilisation of the pointwise-conv + dense kernels over THEIR OWN time, from the
    newest committed PMC pass (profiles/rNN_mfma_util.json, tools/pmc_summary.py): SQ_VALU_MFMA_BUSY_CYCLES summed over the
    class's dispatches / (their summed durations x 2.4 GHz x 1024 SIMDs), bound to the loaded library (see _bound_evidence)."""
    d, rel, note = _bound_evidence("r*_mfma_util.json", workload, cur_digest, cur_plan)
    if d is None:
        return {"source": rel, "binding": note, "pointwise_and_dense": None, "expand_dw": None,
                "target": ">= 0.40 over the pointwise + dense kernels' own time (BASELINE.md section 4)"}
    cls = d.get("classes", {})
    out = {"source": rel, "binding": note, "numerator": d.get("numerator"), "denominator": d.get("denominator"),
           "pointwise_and_dense": cls.get("pw_gemm", {}).get("mfma_util"), "expand_dw": cls.get("expand_dw", {}).get("mfma_util"),
           "valu_per_mfma": {k: v.get("valu_per_mfma") for k, v in cls.items() if v.get("valu_per_mfma")},
           "clock_note": d.get("clock_note"),
           "target": ">= 0.40 over the pointwise + dense kernels' own time (BASELINE.md section 4)"}
    return out


def cpu_worker_main(argv):
    """`bench.py --cpu-worker <blob> <n_samples> <rate> <seconds> <procs>`: the CPU baseline, run in a process of its own
    (no GPU runtime in it, so it may fork).  Throughput mode = what the reference's own benchmark loop does
    (cmd/benchmark/benchmark.go:99-133: one clip per Predict for a fixed duration), once per host core: `procs` forked
    single-thread workers, each pinned to its core, each running whole-model batch-1 forward passes of the torch-CPU
    restatement (oracle/torch_cpu.py: oneDNN convolutions, channels-last, fused SiLU) for `seconds`; the rate is clips
    completed / seconds, summed.  Latency mode = one process, batch 1, with 1 thread and with nproc - 1 threads (the
    reference's rule, internal/inference/tflite/classifier.go:48-58, threads.go:13-30)."""
    blob_path, n_samples, sample_rate, seconds, procs = argv[0], int(argv[1]), int(argv[2]), float(argv[3]), int(argv[4])
    os.environ["OMP_NUM_THREADS"] = "1"
    import birdnet_go_amd  # noqa: F401
    from birdnet_go_amd import synth_model as sm
    import torch
    torch.set_num_threads(1)
    from oracle.torch_cpu import TorchCPU
    model = TorchCPU(open(blob_path, "rb").read())
    x = sm.synth_clips(4, n_samples, sample_rate, first=100000)
    cores = sorted(os.sched_getaffinity(0))
    # two passes: one worker per logical CPU, then one per two (a 256-thread host is 128 cores x SMT, and 256 workers of ~20 MB
    # working set each are DRAM-bound); the better total is the baseline, both are printed
    for procs_now in ([procs, max(1, procs // 2)] if procs >= 16 else [procs]):
        _cpu_throughput_pass(model, x, cores, procs_now, seconds / (2 if procs >= 16 else 1))
    _cpu_latency(model, x, cores, torch)


def _cpu_throughput_pass(model, x, cores, procs, seconds):
    step = max(1, len(cores) // procs)
    rd, wr = os.pipe()
    pids = []
    for w in range(procs):
        pid = os.fork()
        if pid == 0:
            try:
                os.close(rd)
                try:
                    os.sched_setaffinity(0, {cores[(w * step) % len(cores)]})
                except OSError:
                    pass
                model.invoke(x[:1])                                  # warm-up (weight re-layout, oneDNN primitive cache)
                n, t0 = 0, time.time()
                while time.time() - t0 < seconds:
                    model.invoke(x[n % 4:n % 4 + 1])
                    n += 1
                os.write(wr, f"{n} {time.time() - t0:.6f}\n".encode())
            finally:
                os._exit(0)
        pids.append(pid)
    os.close(wr)
    buf = b""
    while True:
        chunk = os.read(rd, 65536)
        if not chunk:
            break
        buf += chunk
    for pid in pids:
        os.waitpid(pid, 0)
    rows = [ln.split() for ln in buf.decode().splitlines() if ln.strip()]
    rate = sum(int(r[0]) / float(r[1]) for r in rows)
    print(f"CPU_THROUGHPUT {len(rows)} {sum(int(r[0]) for r in rows)} {rate:.3f}", flush=True)


def _cpu_latency(model, x, cores, torch):
    # latency mode (after the children are gone: the whole box is free)
    def lat(threads, reps):
        torch.set_num_threads(threads)
        model.invoke(x[:1])
        ts = []
        for i in range(reps):
            t0 = time.time(); model.invoke(x[i % 4:i % 4 + 1]); ts.append(time.time() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3
    print(f"CPU_LATENCY_1T {lat(1, 10):.3f}")
    # the reference would hand XNNPACK nproc - 1 threads; torch's OpenMP pool on a 256-core host turns that into seconds per
    # clip (measured: 22 s), which says nothing about the reference - 8 threads is the figure reported beside the 1-thread one
    nthr = max(1, min(8, len(cores) - 1))
    print(f"CPU_LATENCY_ALLT {nthr} {lat(nthr, 5):.3f}")


def cpu_baseline(blob, n_samples, sample_rate, seconds):
    """CPU restatement ("port") timed on the host cores beside the GPU run: see cpu_worker_main."""
    import subprocess
    import tempfile

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    seconds = seconds or 12.0
    tmp = tempfile.NamedTemporaryFile(suffix=".tflite", delete=False)
    tmp.write(blob)
    tmp.close()
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    res = {"value": None, "unit": "clips/s", "cores": cores, "kind": "port", "sample": "cpu baseline worker failed"}
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", tmp.name, str(n_samples), str(sample_rate),
                              str(seconds), str(cores)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True,
                             timeout=seconds * 4 + 240).stdout
        thr = lat1 = latn = None
        passes = []
        for line in out.splitlines():
            f = line.split()
            if f and f[0] == "CPU_THROUGHPUT":
                passes.append((int(f[1]), int(f[2]), float(f[3])))
                if thr is None or float(f[3]) > thr[2]:
                    thr = passes[-1]
            elif f and f[0] == "CPU_LATENCY_1T":
                lat1 = float(f[1])
            elif f and f[0] == "CPU_LATENCY_ALLT":
                latn = (int(f[1]), float(f[2]))
        if thr:
            res = {"value": thr[2], "unit": "clips/s", "cores": cores, "kind": "port",
                   "clips_per_s_per_core": thr[2] / cores, "worker_processes": thr[0],
                   "passes": [{"processes": q[0], "clips": q[1], "clips_per_s": q[2]} for q in passes],
                   "clips_per_s_one_core_alone": 1e3 / lat1 if lat1 else None,
                   "latency_ms_batch1": {"threads_1": lat1, f"threads_{latn[0]}" if latn else "threads_all": latn[1] if latn else None},
                   "sample": f"{thr[1]} clips of the config-2 generator in {seconds / (2 if cores >= 16 else 1):.0f} s: {thr[0]} single-thread processes (pinned, of {cores} host CPUs; the better of the passes), "
                             "each looping whole-model batch-1 forward passes like the reference's own benchmark loop "
                             "(cmd/benchmark/benchmark.go:99-133) through the torch-CPU restatement of the graph (oracle/torch_cpu.py: oneDNN "
                             f"convolutions, channels-last, fused SiLU; pinned to the parity oracle in tests/test_torch_cpu.py); {time.time() - t0:.0f} s incl. "
                             "start-up; latency = one process, batch 1, median.  RESTATEMENT baseline - NOT TFLite/XNNPACK (no TFLite "
                             "runtime or real weights exist in this environment); the reference documents ~11 clips/s on a 4-core "
                             "Raspberry Pi 5 (doc/wiki/faq.md:192)"}
    except Exception as ex:                                   # noqa: BLE001  (the baseline must never take the bench line down)
        res["sample"] = f"cpu baseline failed: {type(ex).__name__}"
    finally:
        os.unlink(tmp.name)
    return res


def host_pointer_rates(clf, x, reps_small=100, reps_mid=12, reps_big=5):
    """PCIe-inclusive rates through the BLOCKING host-pointer entries - what the reference's callers use
    (internal/analysis/process.go:280-295 one clip per Predict; internal/inference/onnx/classifier.go:372-430 PredictBatch):
    pageable caller memory in, logits in caller memory on return.  Median wall time per call.  x: [256, n_samples] float32."""
    def t(fn, reps):
        fn(); fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]
    res = {}
    ncls = clf.num_species()
    pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
    out256 = np.zeros((256, ncls), np.float32)
    out2048 = np.zeros((2048, ncls), np.float32)
    dt = t(lambda: clf.predict_batch(x[:1].reshape(-1), 1), reps_small)
    res["f32_1"] = {"ms": dt * 1e3, "clips_per_s": 1 / dt}
    dt = t(lambda: clf.predict_topk(x[:1].reshape(-1), 1, 10, 0, 1.0), reps_small)
    res["f32_1_topk"] = {"ms": dt * 1e3, "clips_per_s": 1 / dt}
    dt = t(lambda: clf.predict_batch(x[:8].reshape(-1), 8), reps_small)
    res["f32_8"] = {"ms": dt * 1e3, "clips_per_s": 8 / dt}
    dt = t(lambda: clf.predict_batch(x.reshape(-1), 256, out=out256), reps_mid)
    res["f32_256"] = {"ms": dt * 1e3, "clips_per_s": 256 / dt}
    dt = t(lambda: clf.predict_pcm16(pcm.reshape(-1), 256, out=out256), reps_mid)
    res["pcm16_256"] = {"ms": dt * 1e3, "clips_per_s": 256 / dt}
    bigp = np.tile(pcm, (8, 1))
    dt = t(lambda: clf.predict_pcm16(bigp.reshape(-1), 2048, out=out2048), reps_big)
    res["pcm16_2048"] = {"ms": dt * 1e3, "clips_per_s": 2048 / dt}
    del bigp
    big = np.tile(x, (8, 1))
    dt = t(lambda: clf.predict_batch(big.reshape(-1), 2048, out=out2048), reps_big)
    res["f32_2048"] = {"ms": dt * 1e3, "clips_per_s": 2048 / dt}
    del big
    # the same calls from page-locked caller memory (bnhip_host_alloc: what the Go shim keeps per classifier, as the OpenVINO shim
    # keeps its C-allocated input, backend_openvino.go:673-680): the copy engines read / write the caller's buffers directly
    from birdnet_go_amd import host as _host
    with _host.PinnedArray((256, x.shape[1]), np.float32) as pi, _host.PinnedArray((256, x.shape[1]), np.int16) as pp, \
            _host.PinnedArray((256, ncls), np.float32) as po:
        pi.array[:] = x; pp.array[:] = pcm
        same = bool(np.array_equal(clf.predict_batch(pi.array.reshape(-1), 256, out=po.array), clf.predict_batch(x.reshape(-1), 256, out=out256)))
        dt = t(lambda: clf.predict_batch(pi.array[:1].reshape(-1), 1, out=po.array[:1]), reps_small)
        res["f32_1_pinned"] = {"ms": dt * 1e3, "clips_per_s": 1 / dt}
        dt = t(lambda: clf.predict_batch(pi.array.reshape(-1), 256, out=po.array), reps_mid)
        res["f32_256_pinned"] = {"ms": dt * 1e3, "clips_per_s": 256 / dt}
        dt = t(lambda: clf.predict_pcm16(pp.array.reshape(-1), 256, out=po.array), reps_mid)
        res["pcm16_256_pinned"] = {"ms": dt * 1e3, "clips_per_s": 256 / dt}
        res["pinned_bit_identical_to_pageable"] = same
    # one tick of the real-time window path at 256 sources: every source's capture bytes written into the
    # library's window assembler (bnhip_windows_write), one collect into its page-locked batch buffer, one bnhip_predict_pcm_topk
    # over the rows - where the reference runs 256 batch-1 Predict calls behind inferenceMu (orchestrator.go:531)
    from birdnet_go_amd import stream as _stream
    clip_b = x.shape[1] * 2
    ovb, rdb = clip_b // 2, clip_b - clip_b // 2
    win = _stream.NativeWindows(ovb, rdb, 256)
    try:
        ids = [win.add_source(f"src{i}", 2 * clip_b) for i in range(256)]
        fresh = pcm.view(np.uint8).reshape(256, -1)[:, :rdb]
        parts = {"write": [], "collect": [], "predict": [], "one_call": []}

        def tick(one_call):
            t0 = time.perf_counter()
            for i in ids:
                win.write(i, fresh[i])
            t1 = time.perf_counter()
            parts["write"].append(t1 - t0)
            if one_call:                                      # bnhip_windows_predict_topk: rows assembled under the device's work
                idxs = win.predict_topk(clf, 16, 10, 0, 1.0)[0]
                parts["one_call"].append(time.perf_counter() - t1)
            else:
                idxs, rows = win.collect()
                t2 = time.perf_counter()
                clf.predict_pcm_topk(rows.reshape(-1), 16, len(idxs), 10, 0, 1.0)
                parts["collect"].append(t2 - t1); parts["predict"].append(time.perf_counter() - t2)
            return len(idxs)
        assert tick(False) == 256 and tick(True) == 256
        n_ticks = 30                                          # (a distribution, not one median of 12)
        for _ in range(n_ticks):
            tick(False); tick(True)
        pct = lambda k, q: float(np.percentile(np.asarray(parts[k][-n_ticks:]) * 1e3, q))
        med = {k: pct(k, 50) for k in parts}
        res["realtime_tick_256"] = {"ms": med["one_call"], "ms_p95": pct("one_call", 95), "ms_min": float(min(parts["one_call"][-n_ticks:]) * 1e3),
                                    "ticks": n_ticks, "windows_per_s": 256 / (med["one_call"] * 1e-3),
                                    "two_step_collect_ms": med["collect"], "two_step_predict_pcm_topk_ms": med["predict"],
                                    "two_step_sum_ms_p50": float(np.percentile((np.asarray(parts["collect"][-n_ticks:]) + np.asarray(parts["predict"][-n_ticks:])) * 1e3, 50)),
                                    "capture_side_write_ms": med["write"], "batch_buffer_pinned": win.pinned}
    finally:
        win.close()
    res["note"] = ("blocking C-ABI entries bnhip_predict / bnhip_predict_pcm16, outputs complete on return; calls of >= 128 clips run as chunks "
                   "on two contexts - a call that fits one batch with the plan cut in two as well: fronts per chunk, backs over groups of chunks (csrc/hostpipe.cpp): pageable caller memory is staged through the library's pinned slots by copy threads, "
                   "`_pinned` legs pass bnhip_host_alloc memory, which the copy engines read and write directly; median of the calls; "
                   "realtime_tick_256: 256 sources written into the window assembler, then ONE bnhip_windows_predict_topk (ms = that call; the two-step "
                   "collect + bnhip_predict_pcm_topk beside it)")
    return res


def bat_chirps(n, n_samples=144000, rate=256000, first=0):
    """SURVEY 8d cfg 4 (ii): FM chirp 80 -> 25 kHz, 5 ms, 10 Hz repetition, + N(0, 0.01^2) noise, seed 4321 + i; int16 PCM."""
    t = np.arange(n_samples, dtype=np.float64) / rate
    out = np.empty((n, n_samples), np.int16)
    for j in range(n):
        rng = np.random.default_rng(4321 + first + j)
        ph = t % 0.1
        on = ph < 0.005
        f0, f1 = 80000.0, 25000.0
        phase = 2 * np.pi * (f0 * ph + 0.5 * (f1 - f0) / 0.005 * ph * ph)
        sig = np.where(on, 0.3 * np.sin(phase), 0.0) + rng.normal(0.0, 0.01, n_samples)
        out[j] = np.clip(np.round(sig * 32767.0), -32768, 32767).astype(np.int16)
    return out


def main_bat(args):
    """BASELINE configs[3]: batch of 0.5625 s @ 256 kHz clips, everything resident in HBM: frame-CV gate (8192-point complex128 FFT
    per frame, LDS-resident) -> v2.4-topology backbone -> 1024-d embedding -> regional head (ONNX container) -> scores."""
    import torch
    import birdnet_go_amd  # noqa: F401
    from birdnet_go_amd import host, onnx_build as ob, synth_model as sm
    import ctypes as C
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, n_cls = args.batch or 256, 38
    cfg = sm.SynthConfig(emit_embeddings=True)
    backbone_blob = sm.build_model(cfg)
    head_blob, _ = ob.build_dense_head([cfg.top, n_cls], style="gemm", seed=23)
    backbone = host.HipClassifier(backbone_blob, max_batch=B, depth=1, bf16x3=args.bf16x3)
    head = host.HipClassifier(head_blob, max_batch=B, bf16x3=args.bf16x3)
    pcm_host = bat_chirps(B)
    pcm = torch.from_numpy(pcm_host).to(dev)
    x = (pcm.to(torch.float32) / 32768.0).contiguous()                       # float32(int16) / 32768 (process.go:491-495)
    frames = 1 + (cfg.n_samples - 8192) // 4096
    scratch = torch.empty((B, frames), dtype=torch.float64, device=dev)
    cv = torch.empty(B, dtype=torch.float64, device=dev)
    logits = torch.empty((B, cfg.n_classes), dtype=torch.float32, device=dev)
    emb = torch.empty((B, cfg.top), dtype=torch.float32, device=dev)
    scores = torch.empty((B, n_cls), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    backbone.set_stream(stream.cuda_stream)
    head.set_stream(stream.cuda_stream)
    lib = host.load_library()
    lib.bnhip_us_frame_cv_device.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p]

    def step():
        rc = lib.bnhip_us_frame_cv_device(0, pcm.data_ptr(), 1, B, cfg.n_samples, 256000, 8192, 4096, 20000, scratch.data_ptr(),
                                          cv.data_ptr(), stream.cuda_stream)
        assert rc == frames, rc
        backbone.predict_device(x.data_ptr(), B, logits.data_ptr(), emb.data_ptr())
        head.predict_device(emb.data_ptr(), B, scores.data_ptr())

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(stream)
    lib.bnhip_us_frame_cv_device(0, pcm.data_ptr(), 1, B, cfg.n_samples, 256000, 8192, 4096, 20000, scratch.data_ptr(), cv.data_ptr(),
                                 stream.cuda_stream)
    ev[1].record(stream)
    backbone.predict_device(x.data_ptr(), B, logits.data_ptr(), emb.data_ptr())
    ev[2].record(stream)
    head.predict_device(emb.data_ptr(), B, scores.data_ptr())
    ev[3].record(stream)
    torch.cuda.synchronize(dev)
    stage_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # parity of the timed configuration's own output against the CPU restatements (two clips, outside the timed region)
    from oracle import gofuncs as G, onnx_interp
    from oracle.interp import Interpreter
    rows = [0, B - 1]
    cv_ref = np.array([G.us_frame_cv(pcm_host[r].astype(np.float64) / 32768.0, 256000)[0] for r in rows])
    xr = (pcm_host[rows].astype(np.float32) / np.float32(32768.0))
    _, emb_ref = Interpreter(backbone_blob, conv_backend="torch").invoke(xr)
    sc_ref = onnx_interp.run(head_blob, emb_ref)[0]
    cv_err = float(np.abs(cv[rows].cpu().numpy() - cv_ref).max() / max(np.abs(cv_ref).max(), 1e-30))
    sg = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
    sc_err = float(np.abs(sg(scores[rows].cpu().numpy()) - sg(sc_ref)).max())
    # algorithmic work of the gate: frames x 5 N log2 N flops on complex128, 2 bytes in per sample
    fft_flops = B * frames * 5.0 * 8192 * 13
    out = {"metric": "0.5625s-256kHz bat clips/sec (1 GPU): frame-CV gate + backbone embedding + regional head", "value": B * args.steps / dt,
           "unit": "clips/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (gate: f64)", "data": "synthetic FM chirps "
           "(SURVEY 8d cfg 4), random-init weights",
           "config": {"workload": "BASELINE configs[3]: BattyBirdNET, 256 kHz input, batch %d x 144000 samples, int16 + float32 resident in HBM" % B,
                      "batch_per_gpu": B, "head_classes": n_cls, "head_container": "onnx"},
           "stages_ms": {"us_frame_cv": stage_ms[0], "backbone": stage_ms[1], "head": stage_ms[2]},
           "us_frame_cv": {"frames_per_clip": frames, "gflops_f64": fft_flops / 1e9,
                           "tflops_f64": fft_flops / (stage_ms[0] * 1e-3) / 1e12, "input_gbs": B * cfg.n_samples * 2 / (stage_ms[0] * 1e-3) / 1e9},
           "max_rel_cv_diff_vs_go_restatement": cv_err, "max_abs_score_diff_vs_oracle": sc_err}
    backbone.close(); head.close()
    return out


def main():
    args = parse()
    if args.workload == "bat":
        print(json.dumps(main_bat(args)))
        return
    out = run_model(args)
    if out is not None:
        # The contract is ONE JSON line from rank 0, and a driver may read it as the last line of stdout: RCCL prints a version
        # banner through C stdio when a communicator is created (N > 1), which would otherwise surface after this line when
        # libc flushes at exit.  Flush C stdio first so that the JSON line is the last thing on stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def dist_env():
    """(world, rank, local_rank) from the launcher's environment (torch.distributed.run sets them; plain `python bench.py` = 1, 0, 0)."""
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def dist_begin(dev, backend=None):
    """Process-group set-up of the N-rank run: one process per GPU, backend nccl (= RCCL over xGMI); the CPU test of this very
    code path substitutes gloo (tests/test_dist_cpu.py).  BENCH_FORCE_DIST=1 exercises it on one GPU; BENCH_DIST_BACKEND=gloo lets
    several ranks share ONE GPU (RCCL refuses two ranks on a device) - the record of the per-rank ingest leg on a one-GPU box.
    Returns use_dist."""
    import torch.distributed as dist
    backend = backend or os.environ.get("BENCH_DIST_BACKEND", "nccl")
    world, rank, _ = dist_env()
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
            os.environ["NCCL_DEBUG"] = "WARN"          # no version banner on stdout (see the note where the JSON line is printed)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return use_dist


def dist_model_bytes(cfg, use_dist, dev):
    """Frozen weights: built once on rank 0, broadcast to the other ranks (the only collective on this path)."""
    import birdnet_go_amd  # noqa: F401
    from birdnet_go_amd import shard, synth_model as sm
    _, rank, _ = dist_env()
    if not use_dist:
        return sm.build_model(cfg)
    blob = sm.build_model(cfg) if rank == 0 else None
    return shard.broadcast_model_bytes(blob, src=0, device=dev)


def dist_timing(dt, clips_per_rank_step, steps, use_dist, dev):
    """Contract: the job's time is the MAX over ranks.  Also returns every rank's own clips/s (all-gathered) and the group size the
    collective layer reports, so a scaling run shows at a glance whether all ranks were really there and which one was slow."""
    import torch
    import torch.distributed as dist
    if not use_dist:
        return dt, [clips_per_rank_step * steps / dt], 1
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(every, t)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), [clips_per_rank_step * steps / float(e.item()) for e in every], dist.get_world_size()


def dist_gather(vals, use_dist, dev):
    """Every rank's list of floats on every rank: [[rank 0's], [rank 1's], ...] (one all_gather; [vals] without a group)."""
    import torch
    import torch.distributed as dist
    if not use_dist:
        return [[float(v) for v in vals]]
    t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(every, t)
    return [[float(q) for q in e.cpu().tolist()] for e in every]


def ranks_ingest_leg(calls, reps, use_dist, dev):
    """The N-rank ingest measurement: `calls` = {name: (fn, n_clips)}, fn one BLOCKING host-pointer call on this
    rank.  Every repetition starts behind a barrier, so all ranks stage, copy and compute AT THE SAME TIME - what a node-wide
    deployment does to the host's memory system and PCIe roots, and what a resident-input bench cannot show.  Per call: each rank's
    median time; the job's figure is n_clips x ranks / the slowest rank's median (the contract's max-over-ranks rule)."""
    import torch.distributed as dist
    out = {}
    for name, (fn, n_clips) in calls.items():
        fn()                                                         # warm-up (first use allocates the pinned slots)
        ts = []
        for _ in range(reps):
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        per_rank = [r[0] for r in dist_gather([med], use_dist, dev)]
        out[name] = {"n_clips_per_rank": n_clips, "ms_per_rank": [t * 1e3 for t in per_rank], "ms_max": max(per_rank) * 1e3,
                     "clips_per_s_whole_job": n_clips * len(per_rank) / max(per_rank),
                     "clips_per_s_per_rank": [n_clips / t for t in per_rank]}
    return out


def weak_leg(blob, gpu, depth, args, xs, n, use_dist, dev, stream):
    """`n` clips per GPU (BASELINE configs[1]'s batch) through an engine planned for n, on the first n clips of each resident input
    set: the figure that compares with the N = 1 line when the N-rank workload is configs[2]'s 1 024 per GPU.  Same bracket as the
    contract's: warm-up, barrier + synchronize, K steps, synchronize + barrier, max over ranks."""
    import torch
    import torch.distributed as dist
    from birdnet_go_amd import host
    clf = host.HipClassifier(blob, device=gpu, max_batch=n, depth=depth, lanes=1 if depth > 1 else None, bf16x3=args.bf16x3, precision=args.precision)
    clf.set_stream(stream.cuda_stream)
    outs = [torch.empty((n, clf.num_species()), dtype=torch.float32, device=dev) for _ in xs]
    k = [0]

    def step():
        i = k[0] % len(xs); k[0] += 1
        clf.predict_device(xs[i].data_ptr(), n, outs[i].data_ptr())
    for _ in range(max(args.warmup, 1)):
        step()
    clf.synchronize(); torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    clf.synchronize(); torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    clf.close()
    world = dist.get_world_size() if use_dist else 1
    dt, rates, _ = dist_timing(dt, n, args.steps, use_dist, dev)
    return {"value": n * world * args.steps / dt, "unit": "clips/s", "batch_per_gpu": n, "ms_per_step": dt / args.steps * 1e3,
            "per_rank_clips_per_s": rates, "workload": "BASELINE configs[1]'s batch on every GPU (weak scaling of the N = 1 line)"}


def ranks_host_pointer(blob, gpu, args, x256, use_dist, dev):
    """Every rank at once through the blocking host-pointer entries (bnhip_predict_pcm16: the int16 capture format, 74 MB per 256
    clips): 256 and 2 048 clips, from pageable caller memory (staged by the rank's NUMA-bound copy pool through pinned slots) and from
    page-locked memory (bnhip_host_alloc, placed on the GPU's node)."""
    from birdnet_go_amd import host
    clf = host.HipClassifier(blob, device=gpu, max_batch=256, bf16x3=args.bf16x3, precision=args.precision)
    ncls = clf.num_species()
    pcm = (np.clip(x256, -1, 1) * 32767).astype(np.int16)
    big = np.tile(pcm, (8, 1))
    out256, out2048 = np.zeros((256, ncls), np.float32), np.zeros((2048, ncls), np.float32)
    import ctypes as C
    with host.PinnedArray((2048, x256.shape[1]), np.int16) as pp, host.PinnedArray((2048, ncls), np.float32) as po:
        pp.array[:] = big
        calls = {
            "pcm16_256": (lambda: clf.predict_pcm16(pcm.reshape(-1), 256, out=out256), 256),
            "pcm16_256_pinned": (lambda: clf.predict_pcm16(pp.array[:256].reshape(-1), 256, out=po.array[:256]), 256),
            "pcm16_2048": (lambda: clf.predict_pcm16(big.reshape(-1), 2048, out=out2048), 2048),
            "pcm16_2048_pinned": (lambda: clf.predict_pcm16(pp.array.reshape(-1), 2048, out=po.array), 2048),
        }
        res = ranks_ingest_leg(calls, 7, use_dist, dev)
    pool = [C.c_int(-9) for _ in range(4)]
    host.load_library().bnhip_debug_copy_pool(gpu, *[C.byref(q) for q in pool])          # (after the calls: the pool's threads have started and bound themselves)
    clf.close()
    placement = dist_gather([q.value for q in pool], use_dist, dev)
    res["numa"] = {"per_rank": [{"node": int(q[0]), "copy_threads": int(q[1]), "bound": int(q[2]), "node_cpus_usable": int(q[3])} for q in placement],
                   "note": "copy threads and pinned staging slots of a rank live on its GPU's NUMA node (csrc/numa.cpp; BNHIP_NUMA=0 turns it off)"}
    res["note"] = ("all ranks enter each call behind one barrier; ms = each rank's median of 7; whole-job rate = clips x ranks / slowest rank "
                   "(PCIe + host staging inclusive; bnhip_predict_pcm16, outputs complete on return)")
    return res


def baseline_batch(workload, world):
    """Clips per GPU and step of the BASELINE configuration a run stands for: configs[1] = 256 clips on one GPU; configs[2] = 8 192 clips
    over 8 GPUs = 1 024 per GPU (any N > 1 takes the per-GPU shard of the 8-GPU configuration: weak scaling); configs[4] (Perch) = 4 096
    over 8 = 512 per GPU at every N."""
    if workload == "perch":
        return 512
    return 1024 if world > 1 else 256


def run_model(args):
    """One model workload (birdnet = the contract's line, perch = BASELINE configs[4]); returns the JSON object on rank 0."""
    import torch
    import torch.distributed as dist
    import birdnet_go_amd  # noqa: F401
    from birdnet_go_amd import host, shard, synth_model as sm

    world, rank, local_rank = dist_env()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists by design)")
    gpu = local_rank % max(1, torch.cuda.device_count())       # (one rank per GPU; ranks > GPUs only in the one-GPU record, BENCH_DIST_BACKEND=gloo)
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    use_dist = dist_begin(dev)

    perch = args.workload == "perch"
    cfg = sm.perch_config() if perch else sm.SynthConfig()
    default_batch = not args.batch
    if not args.batch:
        args.batch = baseline_batch(args.workload, world)
    blob = dist_model_bytes(cfg, use_dist, dev)

    B = args.batch
    depth = max(1, args.depth)
    clf = host.HipClassifier(blob, device=gpu, max_batch=B, depth=depth, lanes=1 if depth > 1 else None, bf16x3=args.bf16x3,
                             precision=args.precision)
    local_rank = gpu
    lo, _ = shard.shard_range(B * world, rank, world)       # weak scaling: B clips per rank, distinct seeds
    # NSETS distinct input batches, rotated step by step (round 1 re-ran the same 256 clips every step: 147 MB, small enough
    # to come back from the 256 MiB Infinity Cache; four sets = 590 MB of distinct input do not).  Set 0 is the config-2
    # generator at this rank's clip indices; sets 1.. continue the same generator at later indices.
    NSETS = max(1, args.input_sets)
    x_host = sm.synth_clips(B, cfg.n_samples, cfg.sample_rate, first=lo)
    xs = [torch.from_numpy(x_host).to(dev)]
    for k in range(1, NSETS):
        xs.append(torch.from_numpy(sm.synth_clips(B, cfg.n_samples, cfg.sample_rate, first=lo + k * B * world)).to(dev))
    outs = [torch.empty((B, clf.num_species()), dtype=torch.float32, device=dev) for _ in range(NSETS)]
    x, logits = xs[0], outs[0]
    step_no = [0]
    stream = torch.cuda.current_stream(dev)
    clf.set_stream(stream.cuda_stream)

    def step():
        k = step_no[0] % NSETS
        step_no[0] += 1
        clf.predict_device(xs[k].data_ptr(), B, outs[k].data_ptr())

    # warm-up: W untimed steps.  The last one is bracketed launch-by-launch to find the dominant kernel class and
    # the per-class breakdown; the TIMED region then brackets only that class (bracketing every launch costs ~7 %
    # of a step because each event is a kernel boundary; one class costs ~1 %).
    for _ in range(max(args.warmup - 1, 0)):
        step()
    torch.cuda.synchronize(dev)
    warm_prof, prof_steps = [], []
    if not args.no_profile:
        clf.profile_filter(None)
        clf.profile_enable(True)
    if args.warmup > 0:
        step()
    torch.cuda.synchronize(dev)
    if not args.no_profile:
        warm_prof, prof_steps = clf.profile_read(per_step=True)
        if warm_prof:
            clf.profile_filter(max(warm_prof, key=lambda r: r["ms"])["kernel"])
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = clf.profile_read() if not args.no_profile else []
    if args.detail and rank == 0:
        for r in prof_steps:
            ms = r["ms"] / r["launches"]
            print(f"{r['step']:3d} {r['kernel']:12s} {r['name']:28s} {ms * 1e3:8.1f} us  "
                  f"{r['flops'] / r['launches'] / (ms * 1e-3) / 1e12:6.1f} TF  {r['bytes'] / r['launches'] / (ms * 1e-3) / 1e9:7.0f} GB/s",
                  file=sys.stderr)
    clf.profile_enable(False)
    # per-step distribution (>= 30 batches, median + p95, like cmd/perch-benchmark/main.go:31-32,354-391): a second,
    # untimed-by-contract run of >= 40 steps with ONE event pair per step on the stream the step runs on.  With two contexts
    # in flight the steps overlap, so two figures: the interval between consecutive completions (what throughput is made of)
    # and each step's own start-to-end latency.
    dist_stats = None
    if rank == 0 and not args.no_distribution:
        nd = max(40, args.steps)
        clf.profile_steps(True)
        for _ in range(nd):
            step()
        clf.synchronize()
        torch.cuda.synchronize(dev)
        st_ms, en_ms = clf.profile_steps_read()
        clf.profile_steps(False)
        if len(en_ms) >= 8:
            done = np.sort(en_ms)
            # with `depth` contexts the steps complete in bursts of `depth`; the time per step is the spacing of completions
            # `depth` apart, divided by `depth` (the first `depth` steps start on an empty GPU and are left out)
            iv = ((done[depth:] - done[:-depth]) / depth)[depth:]
            lat = (en_ms - st_ms)[depth:]
            pct = lambda v, q: float(np.percentile(v, q))
            dist_stats = {"steps": int(len(en_ms)), "completion_interval_ms": {"p50": pct(iv, 50), "p95": pct(iv, 95), "max": float(iv.max())},
                          "step_latency_ms": {"p50": pct(lat, 50), "p95": pct(lat, 95)},
                          "clips_per_s_at_p50_interval": B / (pct(iv, 50) * 1e-3),
                          "method": "one HIP event pair per step on the step's own stream; interval = spacing of step completions "
                                    "`pipeline_depth` apart / pipeline_depth (steps on alternating contexts overlap and finish in bursts)"}
    dt, rank_rates, rccl_ranks = dist_timing(dt, B, args.steps, use_dist, dev)
    # N > 1 (every rank takes part; rank 0 reports): the 256-per-GPU figure of the N = 1 line, and the ingest legs
    weak_256 = ingest = None
    if world > 1 and not perch and default_batch and B != 256:
        weak_256 = weak_leg(blob, gpu, depth, args, xs, 256, use_dist, dev, stream)
    if world > 1 and not perch and not args.no_host_pointer:
        ingest = ranks_host_pointer(blob, gpu, args, x_host[:256], use_dist, dev)

    ok = all(bool(torch.isfinite(o).all().item()) for o in outs[:min(NSETS, args.steps + args.warmup)])
    # set 0 was last computed inside (or, for short runs, before) the timed region by the same engine: checks below use it
    # cheap end-to-end consistency check outside the timed region: the first rows of the full batch must match what a
    # small call (single lane, single context, other tile shapes) computes for the same clips
    clf.synchronize()
    small = torch.empty((8, clf.num_species()), dtype=torch.float32, device=dev)
    clf.predict_device(x.data_ptr(), 8, small.data_ptr())
    clf.synchronize()
    torch.cuda.synchronize(dev)
    consist = float((logits[:8] - small).abs().max().item())
    # different tiles only reorder fp32 sums (1e-5 on logits of magnitude ~10); anything larger is a wrong kernel, and a
    # throughput measured with a wrong kernel is not a result (this check caught an LDS staging overflow in round 1)
    if not (consist <= 1e-3):
        print(f"[bench] CONSISTENCY CHECK FAILED: full-batch logits differ from the small-batch path by {consist}", file=sys.stderr)
    # second half of BASELINE's metric: max-abs probability diff of the timed configuration's own output (full batch, pipelined
    # engine) against the CPU restatement, on a few rows, outside the timed region.  The oracle is the checker here, nothing
    # measured runs through it.
    prob_diff = top1_same = None
    oracle_rows = sorted({0, B // 2 - 1 if B > 1 else 0, B - 1})
    if rank == 0 and not args.no_oracle_check:
        from oracle.interp import Interpreter
        ref = Interpreter(blob, conv_backend="torch").invoke(x_host[oracle_rows])[3 if perch else 0]     # Perch v2: logits are output 3
        got = logits[oracle_rows].float().cpu().numpy()
        sg = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
        if perch:                                   # Perch scores are a softmax over the 14795 logits (perch_onnx.go:315-335)
            def sg(v):
                e = np.exp(v.astype(np.float64) - v.astype(np.float64).max(axis=1, keepdims=True))
                return e / e.sum(axis=1, keepdims=True)
        prob_diff = float(np.abs(sg(got) - sg(ref)).max())
        top1_same = bool((got.argmax(1) == ref.argmax(1)).all())
        if not (prob_diff <= (2e-2 if args.precision == "bf16" else 1e-4) and top1_same):
            print(f"[bench] PARITY CHECK FAILED: max-abs prob diff vs oracle {prob_diff}, top-1 identical {top1_same}", file=sys.stderr)
    if rank == 0:
        total_clips = B * world * args.steps
        out = {
            "metric": ("5s-32kHz clips/sec (whole node) + max-abs softmax diff vs the fp32 oracle, Perch-v2-dimension stand-in" if perch else
                       "3s-48kHz clips/sec (whole node) + max-abs prob diff vs TFLite, BirdNET v2.4"), "value": total_clips / dt, "unit": "clips/s",
            "max_abs_prob_diff_vs_oracle": prob_diff, "top1_identical_vs_oracle": top1_same,
            "prob_diff_reference": f"oracle restatement of the TFLite float op semantics on rows {oracle_rows} of the timed batch "
                                   "(tolerance 1e-4; no TFLite runtime or real weights exist in this environment)",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "per_rank_clips_per_s": rank_rates, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 MFMA operands, f32 accumulate and storage" if args.precision == "bf16" else "f32",
            "data": ("synthetic sine+noise clips at 32 kHz; random-init weights of a Perch-v2-DIMENSION stand-in (synth_model.perch_config: "
                     "the real artefact is ONNX, absent from the snapshot, and its graph is unknown here)") if perch else
                    "synthetic sine+noise clips (SURVEY 8d cfg 2); random-init BirdNET-v2.4-topology weights "
                    "(real .tflite absent from the reference snapshot)",
            "config": {"workload": ("BASELINE configs[4]: Google Perch v2 dimensions (14,795-class head), batch 512 x 5 s @ 32 kHz per GPU "
                                    "(4096 over 8), log-mel front-end + EfficientNet-B3-shaped CNN + head on device, raw logits out")
                                   if perch else
                                   (f"BASELINE configs[2]: BirdNET v2.4 fp32, batch {B * world} clips sharded over {world} x MI355X ({B} per GPU), "
                                    "weights broadcast once over RCCL, mel front-end + CNN + head on device, raw logits out"
                                    if world > 1 and B == 1024 else
                                    f"BASELINE configs[1]: BirdNET v2.4 fp32, batch {B} x 3 s @ 48 kHz per GPU, "
                                    "mel front-end + CNN + head on device, raw logits out"),
                       "inputs": "device-resident (device-only): the clips are in HBM when the timed region starts; rates through the "
                                 "blocking host-pointer entries (PCIe-inclusive) are in `host_pointer`",
                       "batch_per_gpu": B, "n_samples": cfg.n_samples, "n_classes": clf.num_species(),
                       "sharding": f"clips index-contiguous over {world} rank(s); weights broadcast once",
                       "pipeline_depth": depth, "input_sets": NSETS},
            "finite_outputs": ok, "max_abs_logit_diff_vs_small_batch": consist, "consistent": bool(consist <= 1e-3),
        }
        if dist_stats:
            out["step_distribution"] = dist_stats
        if weak_256:
            out["weak_256"] = weak_256
        if ingest:
            out["host_pointer_per_rank"] = ingest
        if prof:
            prof = sorted(prof, key=lambda r: -r["ms"])
            dom = prof[0]
            per_launch_ms = dom["ms"] / dom["launches"]
            # roofline side of the dominant kernel class: algorithmic intensity vs the machine balance of the pipe it
            # computes on (the mel front-end runs on the f64 MFMA, everything else on the f32-input MFMA)
            peak_tf = PEAK_F64_MFMA_TFLOPS if dom["kernel"] == "frontend" else PEAK_F32_MFMA_TFLOPS
            pipe_note = None
            if args.precision == "bf16" and dom["kernel"] in ("pw_gemm", "expand_dw"):
                # a bf16 engine issues only SOME layers of a class on the bf16 pipe (one bf16 product per MAC): pointwise layers on
                # the split-bf16 kernel (wm >= 5) and fused layers with bx = 1; the small-K fused layers and the HBM-bound early
                # projections stay on the f32-input MFMA.  The class is priced against the work-weighted (harmonic) peak of the
                # pipes its layers actually issue on - 2 500 TF across the board would mislabel it.
                f_bf16 = f_f32 = 0.0
                for s_ in clf.describe()["steps"]:
                    if s_["kernel"] != dom["kernel"]:
                        continue
                    on_bf16 = (s_["wm_full"] >= 5) if dom["kernel"] == "pw_gemm" else bool(s_["bx"])
                    if on_bf16:
                        f_bf16 += s_["flops"]
                    else:
                        f_f32 += s_["flops"]
                if f_bf16 + f_f32 > 0:
                    peak_tf = (f_bf16 + f_f32) / (f_bf16 / PEAK_BF16_MFMA_TFLOPS + f_f32 / PEAK_F32_MFMA_TFLOPS)
                    pipe_note = {"flops_share_on_bf16_mfma": f_bf16 / (f_bf16 + f_f32), "peak": "work-weighted harmonic mean of 2500 (bf16) and 157.3 (f32) TF"}
                else:
                    peak_tf = PEAK_BF16_MFMA_TFLOPS
            intensity = dom["flops"] / max(dom["bytes"], 1.0)
            # a contraction class whose intensity is above the f32 machine balance is priced against the matrix pipe it issues
            # on even when that pipe (bf16: 312 flop/B balance) would in theory leave it HBM-bound: its measured HBM traffic
            # is the algorithmic minimum at a fraction of the HBM rate, so "hbm" would misname what limits it
            mfma_class = dom["kernel"] in ("pw_gemm", "expand_dw", "frontend", "conv_igemm")
            if intensity > peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9) or (mfma_class and intensity > PEAK_F32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)):
                ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
                roof = {"kernel": dom["kernel"], "bound": "mfma", "achieved": ach, "peak": peak_tf,
                        "unit": "TFLOP/s", "frac": ach / peak_tf, "traffic": None}
            else:
                ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
                roof = {"kernel": dom["kernel"], "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None}
            roof["flop_per_byte"] = intensity
            if dom["kernel"] == "expand_dw" and args.precision != "bf16":
                roof["bound_note"] = ("priced against the f32-input MFMA peak as the contract asks; on gfx950 floating-point VALU work does not co-execute with "
                                      "v_mfma_f32_16x16x4_f32 (tools/ubench/mfma_f32_valu.hip), so this class - an MFMA GEMM plus the swish of the 6x-expanded tensor and "
                                      "the depthwise taps in one kernel - is bound by matrix time + fp-VALU time, of which it reaches ~0.77 (DESIGN.md 5.3)")
            if pipe_note:
                roof["pipes"] = pipe_note
            cur_dig, cur_plan = lib_digest(), plan_signature(clf.describe())
            tr = pmc_traffic(dom["kernel"], args.workload, cur_dig, cur_plan)
            if tr:
                roof["traffic"] = tr["bytes_per_launch"]
                roof["traffic_source"] = tr["source"]
                roof["traffic_binding"] = tr["binding"]
            roof["launches"] = dom["launches"]
            roof["avg_launch_ms"] = per_launch_ms
            desc = clf.describe()
            lanes = desc.get("lanes", 1) if B >= desc.get("lane_min_batch", 1 << 30) else 1
            if depth > 1:
                # successive batches overlap (pipeline depth): a launch of this class shares the GPU with the other
                # context's kernels, so its per-launch rate (the contract's definition; also what rocprofv3's per-kernel
                # average shows) is below what the class sustains when it owns the GPU.  `exclusive` is that figure: the
                # same class bracketed in the last warm-up step, which runs alone.
                roof["pipeline_depth"] = depth
            if lanes > 1 or depth > 1:
                # the engine splits the batch over `lanes` streams that run concurrently: every launch above covers
                # B/lanes clips and shares the GPU with the other lane's kernels, so the per-launch rate (the contract's
                # definition, and what rocprofv3's per-kernel average shows) is ~1/lanes of what the class sustains
                # when it owns the GPU.  `exclusive` is that figure: the same class bracketed in the last warm-up step,
                # which runs single-lane.
                roof["lanes"] = lanes
                w = next((r for r in warm_prof if r["kernel"] == dom["kernel"]), None)
                if w and w["ms"]:
                    if roof["bound"] == "mfma":
                        ex = w["flops"] / (w["ms"] * 1e-3) / 1e12
                    else:
                        ex = w["bytes"] / (w["ms"] * 1e-3) / 1e9
                    roof["exclusive"] = {"achieved": ex, "frac": ex / roof["peak"], "avg_launch_ms": w["ms"] / w["launches"],
                                         "launches": w["launches"]}
            out["roofline"] = roof
            mu = pmc_mfma_util(args.workload, cur_dig, cur_plan)
            if mu:
                out["mfma_util"] = mu
            out["evidence"] = {"lib_digest": cur_dig, "plan_signature": cur_plan, "tune_file": os.environ.get("BNHIP_TUNE_FILE") or None,
                               "tune_source": clf.describe().get("tune_source"), "tune_key": clf.describe().get("tune_key"),
                               "note": "counter files under profiles/ are quoted only when they were collected on this library digest"}
            wsum = sum(r["ms"] for r in warm_prof) or 1.0
            roof["share_of_kernel_time"] = next((r["ms"] for r in warm_prof if r["kernel"] == dom["kernel"]), 0.0) / wsum
            out["kernels_warmup_pass"] = [{"kernel": r["kernel"], "ms_per_step": r["ms"], "launches_per_step": r["launches"],
                                           "tflops": r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] else 0.0,
                                           "gbs": r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] else 0.0}
                                          for r in sorted(warm_prof, key=lambda r: -r["ms"])]
        dsc = clf.describe()
        desc_steps = dsc["steps"]
        out["weights"] = {"model_file_bytes": len(blob), "device_weight_image_bytes": dsc.get("weight_bytes"), "note": ("stand-in; the real perch_v2_no_dft_fp32.onnx is 413.4 MB "
                          "(internal/classifier/model_catalog.go:311)") if perch else
                          "stand-in; the real BirdNET_v2.4_fp32_dfttrunc.onnx is 54.07 MB (model_catalog.go:426)"}
        out["arithmetic"] = {
            "bf16x3": args.bf16x3 if args.bf16x3 is not None else 1,
            "layers_on_split_bf16_mfma": sum(1 for s_ in desc_steps if s_["kernel"] == "pw_gemm" and s_["wm_full"] >= 5),
            "note": "fp32 storage and accumulation throughout; pointwise / dense layers the autotuner moved to the split-bf16 "
                    "kernel form every fp32 product from three exact bf16 pieces per operand (six bf16 MFMA products, error "
                    "<= 2^-23 per product; DESIGN.md section 5, tests/test_bf16x3.py); everything else on the f32-input MFMA"}
        if world == 1 and not args.no_fp32_run and (args.bf16x3 is None or args.bf16x3 != 0) and args.precision != "bf16":
            # the same measurement with every contraction on the f32-input MFMA (outside the contract's timed region): what the
            # split-bf16 path buys, and the number to quote if one insists on f32 MFMA arithmetic only
            clf.close()
            clf = host.HipClassifier(blob, device=local_rank, max_batch=B, depth=depth, lanes=1 if depth > 1 else None, bf16x3=0)
            clf.set_stream(stream.cuda_stream)
            for _ in range(max(args.warmup, 1)):
                step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize(dev)
            dt0 = time.perf_counter() - t1
            out["fp32_mfma_only"] = {"value": B * args.steps / dt0, "unit": "clips/s", "ms_per_step": dt0 / args.steps * 1e3}
        if world == 1 and not perch and not args.no_host_pointer and B == 256:
            # the product boundary: the same model through the blocking host-pointer entries (own engine: default options)
            clf.close()
            clf = host.HipClassifier(blob, device=local_rank, max_batch=B, bf16x3=args.bf16x3, precision=args.precision)
            out["host_pointer"] = host_pointer_rates(clf, x_host)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(blob, cfg.n_samples, cfg.sample_rate, args.cpu_seconds)
    clf.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and world == 1 and not use_dist and not perch and not args.no_secondary and args.precision != "bf16":
        # BASELINE configs[3] / configs[4] next to the headline, at reduced step counts, so that they pass through the same
        # driver-observed run; each leg is the same code `--workload bat|perch` runs on its own
        import copy
        sec = {}
        for name, kw in (("bat", {"workload": "bat", "steps": 10, "warmup": 2}),
                         ("perch_f32", {"workload": "perch", "steps": 8, "warmup": 2, "precision": None}),
                         ("perch_bf16", {"workload": "perch", "steps": 8, "warmup": 2, "precision": "bf16"})):
            a2 = copy.copy(args)
            a2.batch = 0; a2.no_cpu_baseline = True; a2.no_fp32_run = True; a2.no_host_pointer = True; a2.no_secondary = True
            a2.no_distribution = True; a2.detail = False
            for k, v in kw.items():
                setattr(a2, k, v)
            try:
                leg = main_bat(a2) if name == "bat" else run_model(a2)
                keep = ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "data", "config", "max_abs_prob_diff_vs_oracle",
                        "top1_identical_vs_oracle", "stages_ms", "max_rel_cv_diff_vs_go_restatement", "max_abs_score_diff_vs_oracle",
                        "roofline", "consistent", "weights")
                sec[name] = {k: leg[k] for k in keep if k in leg}
            except Exception as ex:                         # noqa: BLE001  (a secondary leg must not take the headline down)
                sec[name] = {"error": f"{type(ex).__name__}: {ex}"}
        out["secondary"] = sec
    return out if rank == 0 else None


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker_main(sys.argv[2:])
    else:
        main()
