"""Multi-GPU scale-out: one process per GPU, videos (or frame blocks) sharded across ranks,
one gather of the entropy time-series.

The reference has no distributed code; its README suggests one process per video
(README.md:108-120).  The path shards naturally:

* videos are independent              -> rank r takes videos r, r+W, r+2W, ...
* spatial frames are independent, and a transition row needs only the previous frame
                                        -> contiguous frame blocks with a 1-frame halo
* the only exchange is the result: ONE gather of the per-rank series to rank 0
  (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).

No collective touches the data path.  ``compute`` callbacks receive numpy arrays and return the
per-frame series; the analyzers' engine call is the default in bench.py, tests pass the oracle.
"""

from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def video_shard(n_videos: int, rank: int, world: int) -> List[int]:
    """Round-robin video indices of ``rank``."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("need 0 <= rank < world")
    return list(range(rank, n_videos, world))


def frame_shard(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of output rows for ``rank`` (sizes differ by at most 1)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("need 0 <= rank < world")
    base, extra = divmod(n_rows, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def transition_frame_block(n_frames: int, rank: int, world: int) -> Tuple[int, int, int, int]:
    """Transition rows [r0, r1) of ``rank`` and the frames [f0, f1) they need (1-frame halo:
    row r compares frames r and r+1)."""
    r0, r1 = frame_shard(max(n_frames - 1, 0), rank, world)
    return r0, r1, r0, (r1 + 1 if r1 > r0 else r0)


class PlacementError(RuntimeError):
    """Two ranks of a one-process-per-GPU job compute on the same device."""


def check_placement(records: Sequence[dict], require_distinct: bool) -> dict:
    """``records``: one dict per rank with at least ``rank`` and ``pci_bus_id`` (+ ``host``).  Returns the summary a job
    prints (ranks per device, whether all devices are distinct); raises ``PlacementError`` naming the ranks that share
    a device when ``require_distinct`` (RCCL jobs: one rank per GPU is the whole point of the scale-out,
    README.md:108-120 of the reference runs one process per video)."""
    by_dev = {}
    for r in records:
        by_dev.setdefault((r.get("host", ""), r["pci_bus_id"]), []).append(int(r["rank"]))
    shared = {f"{h}:{d}" if h else d: ranks for (h, d), ranks in by_dev.items() if len(ranks) > 1}
    if shared and require_distinct:
        raise PlacementError(
            "ranks share a GPU: " + "; ".join(f"device {d} <- ranks {ranks}" for d, ranks in sorted(shared.items()))
            + " (one process per GPU expected: check LOCAL_RANK / HIP_VISIBLE_DEVICES of the launcher)")
    return {"n_ranks": len(records), "n_devices": len(by_dev), "distinct": not shared, "shared": shared}


def placement(engine, require_distinct: Optional[bool] = None) -> dict:
    """Which device every rank's engine context computes on, gathered on ALL ranks (one ``all_gather_object`` outside
    any timed region): ``{"ranks": [{rank, device_index, pci_bus_id, host, visible_devices}], "backend", "rccl_version",
    "n_devices", "distinct"}``.  Under the ``nccl`` (= RCCL) backend two ranks on one device raise ``PlacementError`` on
    every rank (``require_distinct`` overrides; a gloo rehearsal on one device may share it)."""
    import socket

    import torch
    import torch.distributed as dist

    lib = engine.lib
    mine = {"rank": 0, "device_index": int(engine.device_id), "pci_bus_id": engine.pci_bus_id(),
            "host": socket.gethostname(), "visible_devices": int(lib.vet_device_count())}
    backend, version = None, None
    if dist.is_available() and dist.is_initialized():
        mine["rank"] = dist.get_rank()
        backend = dist.get_backend()
        records = [None] * dist.get_world_size()
        dist.all_gather_object(records, mine)
    else:
        records = [mine]
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            version = "unknown"
    if require_distinct is None:
        require_distinct = backend == "nccl"
    summary = check_placement(records, require_distinct)
    return dict(summary, ranks=sorted(records, key=lambda r: r["rank"]), backend=backend, rccl_version=version)


def gather_series(series: np.ndarray, dst: int = 0, max_len: Optional[int] = None, device=None):
    """ONE gather of every rank's 1-D float64 series to ``dst`` (through the process group's backend
    whenever one is initialised, also for a world of one rank).

    The payload is ``[len, values..., padding]`` of a common size; ``max_len`` (known when all
    videos have the same number of frames) avoids the extra MAX all-reduce.  Returns the list
    of per-rank arrays on ``dst`` and ``None`` elsewhere.  Works without an initialised
    process group (single process)."""
    import torch
    import torch.distributed as dist

    series = np.ascontiguousarray(series, dtype=np.float64).reshape(-1)
    if not (dist.is_available() and dist.is_initialized()):
        return [series.copy()]
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is not None:
        dev = device
    elif dist.get_backend() == "nccl":           # RCCL moves device buffers
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    if max_len is None:
        m = torch.tensor([len(series)], dtype=torch.int64, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        max_len = int(m.item())
    if len(series) > max_len:
        raise ValueError("series longer than max_len")
    payload = torch.zeros(max_len + 1, dtype=torch.float64, device=dev)
    payload[0] = float(len(series))
    payload[1:1 + len(series)] = torch.from_numpy(series).to(dev)
    out = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, out, dst=dst)
    if rank != dst:
        return None
    res = []
    for t in out:
        t = t.cpu().numpy()
        res.append(t[1:1 + int(t[0])].copy())
    return res


def analyze_videos(videos: Sequence, compute: Callable[[object], np.ndarray], dst: int = 0,
                   max_len: Optional[int] = None, device=None):
    """One video per rank at a time: rank r computes videos r, r+W, ...; after each round ONE
    gather brings that round's series to ``dst``.  Returns {video index: series} on ``dst``."""
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    results = {}
    rounds = (len(videos) + world - 1) // world
    for rd in range(rounds):
        vid = rd * world + rank
        series = compute(videos[vid]) if vid < len(videos) else np.empty(0)
        got = gather_series(series, dst=dst, max_len=max_len, device=device)
        if got is not None:
            for r, s in enumerate(got):
                if rd * world + r < len(videos):
                    results[rd * world + r] = s
    return results if rank == dst else None


def analyze_videos_batched(videos: Sequence, compute_batch: Callable[[list], List[np.ndarray]], dst: int = 0, device=None):
    """Many SHORT videos (the reference's scale-out unit, README.md:108-120): rank r takes videos r, r+W, r+2W, ... and
    runs its whole share through ONE batched engine call (``compute_batch(list of videos) -> list of series``, e.g.
    ``lambda vs: [r["entropy"] for r in plan.spatial_batch(vs)]`` — one launch instead of one per video), then ONE
    gather brings every rank's series to ``dst``.  Returns {video index: series} on ``dst``."""
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = video_shard(len(videos), rank, world)
    series = [np.ascontiguousarray(s, dtype=np.float64).reshape(-1) for s in compute_batch([videos[v] for v in mine])] if mine else []
    if len(series) != len(mine):
        raise ValueError("compute_batch must return one series per video")
    # one payload per rank: [number of videos, their lengths..., the series back to back]
    payload = np.concatenate([[float(len(series))], [float(len(s)) for s in series]] + series) if series else np.zeros(1)
    got = gather_series(payload, dst=dst, device=device)
    if got is None:
        return None
    results = {}
    for r, pay in enumerate(got):
        n = int(pay[0])
        lens = pay[1:1 + n].astype(np.int64)
        off = 1 + n
        for j, v in enumerate(video_shard(len(videos), r, world)[:n]):
            results[v] = pay[off:off + lens[j]].copy()
            off += int(lens[j])
    return results


def transition_frame_sharded(mu: np.ndarray, mv: np.ndarray, compute: Callable[[np.ndarray, np.ndarray], np.ndarray],
                             dst: int = 0, device=None):
    """One video across all ranks in transition mode: each rank computes a contiguous block of
    rows from its frames plus a 1-frame halo; ONE gather concatenates them on ``dst``."""
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    T = mu.shape[0]
    r0, r1, f0, f1 = transition_frame_block(T, rank, world)
    part = compute(mu[f0:f1], mv[f0:f1]) if r1 > r0 else np.empty(0)
    base, extra = divmod(max(T - 1, 0), world)
    got = gather_series(part, dst=dst, max_len=base + (1 if extra else 0), device=device)
    return np.concatenate(got) if got is not None else None


def spatial_frame_sharded(mu: np.ndarray, mv: np.ndarray, compute: Callable[[np.ndarray, np.ndarray], np.ndarray],
                          dst: int = 0, device=None):
    """One video across all ranks in spatial mode (strong scaling): frames are independent
    (entropy_utils.py:147-211 touches one row), so each rank computes a contiguous block of frames
    (``frame_shard``, no halo) and ONE gather concatenates the blocks on ``dst``."""
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    T = mu.shape[0]
    f0, f1 = frame_shard(T, rank, world)
    part = compute(mu[f0:f1], mv[f0:f1]) if f1 > f0 else np.empty(0)
    base, extra = divmod(T, world)
    got = gather_series(part, dst=dst, max_len=base + (1 if extra else 0), device=device)
    return np.concatenate(got) if got is not None else None


def analyze_directories(directories: Sequence, config=None, mode: str = "spatial", dst: int = 0, device=None):
    """One video directory per GPU at a time (README.md:108-120 of the reference suggests a process
    pool; here it is one process per GPU under ``torch.distributed.run``).  Every rank builds its own
    analyzer, runs ``process_directory`` + ``compute_entropy`` on its share and contributes the
    entropy series to ONE gather per round.  Returns ``{directory index: entropy[T]}`` on ``dst``."""
    from . import SpatialEntropyAnalyzer, TransitionEntropyAnalyzer

    import torch.distributed as dist

    cls = {"spatial": SpatialEntropyAnalyzer, "transition": TransitionEntropyAnalyzer}[mode]
    analyzer = cls(config)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        from . import _native
        placement(_native.Engine.default())       # raises PlacementError under RCCL if two ranks share a GPU

    def compute(directory):
        analyzer.process_directory(directory)
        return analyzer.compute_entropy()["entropy"].to_numpy(dtype=np.float64)

    return analyze_videos(list(directories), compute, dst=dst, device=device)
