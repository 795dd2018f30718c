"""NUMA placement of the host ingest path (csrc/numa.cpp, csrc/hostpipe.cpp; VERDICT r5 item 3, SURVEY 8e "scaling limiter is host PCM
staging bandwidth"): per GPU, the copy threads that stage its chunks run on the CPUs of the GPU's own NUMA node and its pinned slots
are allocated there.  CPU: the sysfs parsing on a fake tree.  GPU: what the box reports, and that placement never changes a bit."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from birdnet_go_amd import host


def _cpulist(lib, text):
    buf = (C.c_int * 4096)()
    n = lib.bnhip_debug_parse_cpulist(text.encode(), buf, 4096)
    return list(buf[:n])


def test_cpulist_parsing(built_lib):
    lib = host.load_library()
    assert _cpulist(lib, "0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert _cpulist(lib, " 5 ") == [5]
    assert _cpulist(lib, "7,3,3-4") == [3, 4, 7]                      # sorted, duplicates folded
    assert _cpulist(lib, "0-127,256-383") == list(range(128)) + list(range(256, 384))
    for bad in ("", "3-1", "a", "1,,2", "1-", "-3", "1;2", "99999999999"):
        assert _cpulist(lib, bad) == [], bad


def test_pci_numa_node_and_cpus_from_a_fake_sysfs(built_lib, tmp_path):
    lib = host.load_library()
    dev = tmp_path / "bus" / "pci" / "devices"
    for bdf, node in (("0000:c1:00.0", "1\n"), ("0000:05:00.0", "0\n"), ("0000:85:00.0", "-1\n"), ("0000:99:00.0", "junk\n")):
        (dev / bdf).mkdir(parents=True)
        (dev / bdf / "numa_node").write_text(node)
    for node, cpus in ((0, "0-63,128-191\n"), (1, "64-127,192-255\n")):
        d = tmp_path / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus)

    def probe(bdf):
        node, buf = C.c_int(-7), (C.c_int * 1024)()
        n = lib.bnhip_debug_numa_probe(str(tmp_path).encode(), bdf.encode(), C.byref(node), buf, 1024)
        return node.value, list(buf[:n])

    assert probe("0000:C1:00.0") == (1, list(range(64, 128)) + list(range(192, 256)))      # hipDeviceGetPCIBusId prints upper case
    assert probe("0000:05:00.0") == (0, list(range(0, 64)) + list(range(128, 192)))
    assert probe("0000:85:00.0") == (-1, [])                                                # the kernel's "no NUMA information"
    assert probe("0000:99:00.0") == (-1, [])
    assert probe("0000:aa:00.0") == (-1, [])                                                # no such device
    assert probe("") == (-1, [])


@pytest.mark.gpu
def test_copy_pool_of_the_device_and_placement_changes_no_bit(gpu, tiny_blob, tiny_cfg, tmp_path):
    """The GPU's copy pool exists and, where the box exposes NUMA information, runs on the GPU's node; a 160-clip pageable call (staged by
    the pool through the pinned slots) gives the same bits with BNHIP_NUMA=0 (one unbound pool, default page placement)."""
    blob_path = tmp_path / "m.tflite"
    blob_path.write_bytes(tiny_blob)
    code = (
        "import ctypes as C, json, sys, numpy as np\n"
        "import birdnet_go_amd\n"
        "from birdnet_go_amd import host, synth_model as sm\n"
        "cfg = sm.tiny_config()\n"
        "c = host.HipClassifier(open(sys.argv[1], 'rb').read(), max_batch=64, autotune=False)\n"      # (two processes: no timing race may pick the tiles)
        "x = sm.synth_clips(160, cfg.n_samples, cfg.sample_rate)\n"
        "big = np.tile(x, (1, 1))\n"
        "y = c.predict_batch(big.reshape(-1), 160)\n"
        "lib = host.load_library()\n"
        "v = [C.c_int(-9) for _ in range(4)]\n"
        "lib.bnhip_debug_copy_pool(0, *[C.byref(q) for q in v])\n"
        "print(json.dumps({'pool': [q.value for q in v], 'y': y.tobytes().hex()}))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(numa):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("BNHIP_NUMA", None)
        if not numa:
            env["BNHIP_NUMA"] = "0"
        out = subprocess.run([sys.executable, "-c", code, str(blob_path)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    on, off = run(True), run(False)
    node, threads, bound, cpus = on["pool"]
    print("copy pool of device 0: node", node, "threads", threads, "bound", bound, "usable cpus of the node", cpus)
    assert threads >= 1
    if node >= 0 and cpus > 0:
        assert bound == threads <= cpus
    else:
        assert bound == 0
    assert off["pool"][0] == -1 and off["pool"][2] == 0
    assert on["y"] == off["y"]


def test_numa_preference_scope_restores_the_threads_own_policy(tmp_path):
    """csrc/numa.cpp NumaPrefer (g++, no GPU): inside the scope the thread prefers the node, afterwards its own policy is back -
    MPOL_DEFAULT here, an interleave policy set by the host in the second half - and a refused syscall changes nothing."""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "numa_policy")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "birdnet-go_amd", "csrc"),
                           os.path.join(root, "tests", "native", "numa_policy.cpp"), os.path.join(root, "birdnet-go_amd", "csrc", "numa.cpp"), "-o", exe, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout.strip()
    assert out.startswith("OK") or out.startswith("SKIP"), out
