"""world_size-2, -3 and -8 gloo tests (CPU) of the multi-GPU plumbing: video sharding, the single gather of
the entropy series, and frame sharding with a halo in transition mode.  The per-rank compute is
the CPU oracle here (the product's compute is the HIP engine; on the GPU box bench.py drives it)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, q):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "viewport-entropy-toolkit_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from viewport_entropy_toolkit import _dist, _synthetic
    from oracle import vet_oracle as vo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case == "videos":
            videos = [_synthetic.random_walk_video(6, 20 + 3 * v, base_seed=9, video_id=v) for v in range(5)]
            compute = lambda mv_: vo.spatial_series(mv_[0], mv_[1], 100, 200, [20])[0]  # noqa: E731
            got = _dist.analyze_videos(videos, compute)
            if rank == 0:
                ref = {v: compute(videos[v]) for v in range(5)}
                ok = set(got) == set(ref) and all(np.array_equal(got[v], ref[v]) for v in ref)
                q.put(("videos", ok))
        elif case == "frames":
            mu, mv = _synthetic.random_walk_video(10, 41, base_seed=4)
            compute = lambda a, b: vo.transition_series(a, b, 100, 200, [20])[0]  # noqa: E731
            got = _dist.transition_frame_sharded(mu, mv, compute)
            if rank == 0:
                q.put(("frames", bool(np.array_equal(got, compute(mu, mv)))))
        elif case == "batched":
            videos = [_synthetic.random_walk_video(5 + v, 12 + 2 * v, base_seed=3, video_id=v) for v in range(7)]
            one = lambda mv_: vo.spatial_series(mv_[0], mv_[1], 100, 200, [20])[0]  # noqa: E731
            got = _dist.analyze_videos_batched(videos, lambda vs: [one(v) for v in vs])
            if rank == 0:
                ok = set(got) == set(range(7)) and all(np.array_equal(got[v], one(videos[v])) for v in got)
                q.put(("batched", ok))
        elif case == "sframes":
            mu, mv = _synthetic.random_walk_video(10, 41, base_seed=4, p_absent=0.1)
            compute = lambda a, b: vo.spatial_series(a, b, 100, 200, [20, 50])[0]  # noqa: E731
            got = _dist.spatial_frame_sharded(mu, mv, compute)
            if rank == 0:
                q.put(("sframes", bool(np.array_equal(got, compute(mu, mv)))))
        elif case == "fixed":
            s = np.arange(7, dtype=np.float64) + 100 * rank
            got = _dist.gather_series(s, max_len=7)
            if rank == 0:
                q.put(("fixed", len(got) == world and all(np.array_equal(got[r], np.arange(7) + 100.0 * r) for r in range(world))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("videos", 2), ("frames", 2), ("fixed", 2),
                                        # uneven shards: 5 videos / 40 rows over 3 ranks, 8 ranks with more ranks than videos
                                        ("videos", 3), ("frames", 3), ("videos", 8), ("frames", 8),
                                        ("sframes", 2), ("sframes", 3), ("sframes", 8),
                                        ("batched", 2), ("batched", 3), ("batched", 8)])
def test_ranks_gloo(case, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    name, ok = q.get(timeout=10)
    assert name == case and ok


def test_shard_arithmetic():
    sys.path.insert(0, str(ROOT / "viewport-entropy-toolkit_amd"))
    from viewport_entropy_toolkit import _dist
    assert _dist.video_shard(8, 3, 8) == [3] and _dist.video_shard(10, 1, 4) == [1, 5, 9]
    for rows, world in ((9999, 8), (5, 8), (0, 2), (16, 4)):
        blocks = [_dist.frame_shard(rows, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1
    r0, r1, f0, f1 = _dist.transition_frame_block(10000, 3, 8)
    assert (f0, f1) == (r0, r1 + 1) and r1 - r0 in (1249, 1250)
    # single process: gather_series is the identity
    assert np.array_equal(_dist.gather_series(np.arange(3.0))[0], np.arange(3.0))


def test_placement_check():
    """Two ranks of a one-process-per-GPU job on one device: refused when distinct devices are required (RCCL jobs),
    reported otherwise (a gloo rehearsal on one device)."""
    sys.path.insert(0, str(ROOT / "viewport-entropy-toolkit_amd"))
    from viewport_entropy_toolkit import _dist
    ok = [{"rank": r, "pci_bus_id": f"0000:{5 + r:02x}:00.0", "host": "n0"} for r in range(8)]
    s = _dist.check_placement(ok, require_distinct=True)
    assert s == {"n_ranks": 8, "n_devices": 8, "distinct": True, "shared": {}}
    stacked = [{"rank": r, "pci_bus_id": "0000:05:00.0", "host": "n0"} for r in range(3)] + [{"rank": 3, "pci_bus_id": "0000:06:00.0", "host": "n0"}]
    s = _dist.check_placement(stacked, require_distinct=False)
    assert s["n_devices"] == 2 and not s["distinct"] and s["shared"] == {"n0:0000:05:00.0": [0, 1, 2]}
    with pytest.raises(_dist.PlacementError, match=r"ranks \[0, 1, 2\]"):
        _dist.check_placement(stacked, require_distinct=True)
    # the same bus id on two hosts is two devices
    two_hosts = [{"rank": 0, "pci_bus_id": "0000:05:00.0", "host": "a"}, {"rank": 1, "pci_bus_id": "0000:05:00.0", "host": "b"}]
    assert _dist.check_placement(two_hosts, require_distinct=True)["distinct"]


def test_default_engine_device_never_falls_back_to_device_0(monkeypatch):
    """an out-of-range LOCAL_RANK / VET_DEVICE must raise, not stack the ranks on GPU 0."""
    sys.path.insert(0, str(ROOT / "viewport-entropy-toolkit_amd"))
    from viewport_entropy_toolkit import _native
    pick = _native.Engine.default_device_id
    monkeypatch.delenv("VET_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert pick(1) == 0 and pick(8) == 0
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert pick(8) == 3
    with pytest.raises(_native.NativeUnavailable, match="LOCAL_RANK=3 names no visible device"):
        pick(1)
    monkeypatch.setenv("VET_DEVICE", "0")           # VET_DEVICE wins over LOCAL_RANK (a rehearsal of N ranks on one GPU)
    assert pick(1) == 0
    monkeypatch.setenv("VET_DEVICE", "-1")
    with pytest.raises(_native.NativeUnavailable, match="VET_DEVICE=-1"):
        pick(8)
    monkeypatch.setenv("VET_DEVICE", "gpu0")
    with pytest.raises(_native.NativeUnavailable, match="not a device index"):
        pick(8)
