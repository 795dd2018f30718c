"""N>1 path on CPU: world_size-2 gloo.  Checks shard arithmetic, the model-bytes broadcast (the only
collective on the hot path) and that sharded oracle runs reassemble to the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest

from birdnet_go_amd import shard, synth_model as sm


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 8192, 8193):
        for w in (1, 2, 3, 8):
            rs = [shard.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import birdnet_go_amd  # noqa: F401
    from oracle.interp import Interpreter
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = sm.tiny_config()
    blob = sm.build_model(cfg) if rank == 0 else None
    blob = shard.broadcast_model_bytes(blob, src=0)
    n = 5
    lo, hi = shard.shard_range(n, rank, world)
    x = sm.synth_clips(hi - lo, cfg.n_samples, cfg.sample_rate, first=lo)
    out = Interpreter(blob).invoke(x)[0]
    # the ENGINE on every rank, as far as a CPU goes: the broadcast bytes go through the C ABI's reader and planner
    # (plan_only), and every rank must arrive at the same plan as rank 0's local build
    from birdnet_go_amd import host
    clf = host.HipClassifier(blob, plan_only=True, max_batch=hi - lo)
    d = clf.describe()
    plan = (clf.n_samples, clf.num_species(), d["weight_bytes"], tuple((s["kernel"], s["name"]) for s in d["steps"]))
    clf.close()
    dist.barrier()
    q.put((rank, lo, hi, len(blob), out, plan))
    dist.destroy_process_group()


def test_gloo_world2_broadcast_and_shard():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    cfg = sm.tiny_config()
    blob = sm.build_model(cfg)
    assert all(r[3] == len(blob) for r in res)
    from oracle.interp import Interpreter
    whole = Interpreter(blob).invoke(sm.synth_clips(5, cfg.n_samples, cfg.sample_rate))[0]
    got = np.concatenate([r[4] for r in res])
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 3, 3, 5)
    from birdnet_go_amd import host
    c = host.HipClassifier(blob, plan_only=True, max_batch=3)
    d = c.describe()
    want = (c.n_samples, c.num_species(), d["weight_bytes"], tuple((s["kernel"], s["name"]) for s in d["steps"]))
    assert res[0][5] == want and res[1][5] == want           # both ranks planned the broadcast model identically
    assert np.abs(got - whole).max() < 1e-5


def _bench_worker(rank, world, port, q):
    """bench.py's own N-rank code path, gloo substituted for RCCL, up to the point where the engine is created."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist
    import birdnet_go_amd  # noqa: F401
    import bench
    from birdnet_go_amd import host
    assert bench.dist_env() == (world, rank, rank)
    dev = torch.device("cpu")
    use_dist = bench.dist_begin(dev, backend="gloo")
    cfg = sm.tiny_config()
    blob = bench.dist_model_bytes(cfg, use_dist, dev)
    lo, hi = shard.shard_range(4 * world, rank, world)                 # weak scaling: B clips per rank, distinct seeds
    clf = host.HipClassifier(blob, plan_only=True, max_batch=hi - lo)   # (on the GPU box: device=local_rank)
    plan = tuple(s["kernel"] for s in clf.describe()["steps"])
    clf.close()
    dist.barrier()
    dt, rates, ranks = bench.dist_timing(0.5 + 0.25 * rank, 4, 10, use_dist, dev)     # rank 1 pretends to be slower
    dist.barrier()
    # the per-rank ingest leg (round 6): every repetition behind one barrier, each rank's median gathered, the job's rate from the
    # slowest rank - here with a stand-in for the blocking device call (rank 1 twice as slow)
    import time
    gathered = bench.dist_gather([rank, 10.0 * rank + 1], use_dist, dev)
    leg = bench.ranks_ingest_leg({"fake_256": (lambda: time.sleep(0.02 * (rank + 1)), 256)}, 3, use_dist, dev)
    q.put((rank, use_dist, len(blob), (lo, hi), plan, dt, rates, ranks, gathered, leg))
    dist.destroy_process_group()


def test_bench_n_rank_path_under_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    blob = sm.build_model(sm.tiny_config())
    for rank, use_dist, n, rng, plan, dt, rates, ranks, gathered, leg in res:
        assert gathered == [[0.0, 1.0], [1.0, 11.0]]
        f = leg["fake_256"]
        assert f["n_clips_per_rank"] == 256 and len(f["ms_per_rank"]) == 2 and f["ms_per_rank"] == res[0][9]["fake_256"]["ms_per_rank"]
        assert 19 <= f["ms_per_rank"][0] <= 35 and 39 <= f["ms_per_rank"][1] <= 60 and f["ms_max"] == max(f["ms_per_rank"])
        assert abs(f["clips_per_s_whole_job"] - 512 / (f["ms_max"] * 1e-3)) < 1e-6
        assert use_dist and n == len(blob) and ranks == 2
        assert rng == (4 * rank, 4 * rank + 4)
        assert plan == res[0][4]
        assert abs(dt - 0.75) < 1e-9                                   # the job's time is the slowest rank's
        assert np.allclose(rates, [80.0, 40.0 / 0.75])                 # every rank's own rate, in rank order, on every rank
