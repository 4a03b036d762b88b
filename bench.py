#!/usr/bin/env python3
"""bench.py — whole-job throughput of the viewport -> tile -> entropy hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload config3|config2|config4|config5]

A "step" is one pass of the hot path (one C-ABI call: samples resident in HBM -> per-frame
entropy + nearest-tile assignments in HBM) over one synthetic video per GPU.  N > 1 is launched
by torch.distributed.run, one rank per GPU, one video per rank (weak scaling, no data-path
collective); the per-video entropy series are collected on rank 0 with ONE RCCL gather per
step.  Rank 0 prints one JSON line (contract in the task statement) carrying `roofline`
(dominant kernel, hipEvent-timed on the launch stream inside the timed region) and, at N = 1,
`cpu_baseline` (the C port of the reference path timed on this box's host cores on a bounded
sample of the same workload), and `parity`: after the timed region every rank compares the series
its timed steps produced with the C port's on the first frames of the same video (indices bit-exact,
entropy within 1e-6) and runs one golden of the real reference through the HIP path; a violation
prints no metric line and exits with status 3.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "viewport-entropy-toolkit_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable

WORKLOADS = {
    # name: (users, frames, tile_counts, mode, weighted)      BASELINE.json configs[...]
    "config2": (64, 3000, [50, 100, 200], "spatial", True),
    "config3": (1024, 30000, [500], "spatial", True),
    "config3u": (1024, 30000, [500], "spatial", False),     # nearest-tile (unweighted) histogram
    "config4": (256, 10000, [50, 100, 200], "spatial", True),
    "config5": (512, 10000, [200], "transition", True),
    # 64 config-2-sized videos per GPU in one launch (vet_spatial_entropy_batch); --loop runs them one by one
    "config2x64": (64, 3000, [50, 100, 200], "spatial", True),
    "config2x64u": (64, 3000, [50, 100, 200], "spatial", False),      # the same batch, nearest-tile counts (k_spatial_u_lds)
    "config2x64t": (64, 3000, [50, 100, 200], "transition", True),    # the same batch in transition mode (k_transition_run)
    # 64 videos of config 5's audience (512 users, tile_counts=[200]) x 1000 frames, transition mode, one launch
    "config5x64": (512, 1000, [200], "transition", True),
    # the reference's default AnalyzerConfig.tile_counts (config.py:27) on a config-4-sized video
    "defaults": (256, 10000, [20, 50, 100, 250, 1000], "spatial", True),
    "defaults_u": (256, 10000, [20, 50, 100, 250, 1000], "spatial", False),
}
BATCH = {"config2x64": 64, "config2x64u": 64, "config2x64t": 64, "config5x64": 64}


def synth_video(U, T, seed, video_id, kind="random_walk"):
    """SURVEY.md §8d generators, vectorised over users (one PCG64 stream per video).
    random_walk: independent smooth walks (the default workload); uniform: uniform on the sphere,
    no locality at all; clustered: everybody within a few degrees of one moving attention point
    (what real audiences do: same-row / same-tile contention)."""
    rng = np.random.default_rng(seed + video_id * 10**6)
    if kind == "uniform":
        mu = rng.random((T, U))
        mv = np.clip(np.arccos(1.0 - 2.0 * rng.random((T, U))) / np.pi, 0.0, 1.0)
    elif kind == "clustered":
        cu = np.mod(0.5 + np.cumsum(rng.normal(0.0, 0.004, (T, 1)), axis=0), 1.0)
        cv = np.clip(0.5 + np.cumsum(rng.normal(0.0, 0.002, (T, 1)), axis=0), 0.2, 0.8)
        mu = np.mod(cu + rng.normal(0.0, 0.02, (T, U)), 1.0)
        mv = np.clip(cv + rng.normal(0.0, 0.02, (T, U)), 0.0, 1.0)
    elif kind in ("uniform_half", "uniform_quarter"):   # diagnostic: uniform over a part of the sphere (table footprint)
        frac = 0.5 if kind == "uniform_half" else 0.25
        mu = rng.random((T, U))
        mv = np.clip(np.arccos(1.0 - 2.0 * frac * rng.random((T, U))) / np.pi, 0.0, 1.0)
    elif kind == "single":          # diagnostic: everybody in one direction (pure per-frame overhead)
        mu, mv = np.full((T, U), 0.3), np.full((T, U), 0.4)
    elif kind == "fixed":           # diagnostic: U distinct directions, the same in every frame (cache-hot rows)
        mu = np.broadcast_to(rng.random((1, U)), (T, U)).copy()
        mv = np.broadcast_to(np.clip(np.arccos(1.0 - 2.0 * rng.random((1, U))) / np.pi, 0.0, 1.0), (T, U)).copy()
    else:
        mu = np.mod(0.5 + np.cumsum(rng.normal(0.0, 0.01, (T, U)), axis=0), 1.0)
        mv = np.clip(0.5 + np.cumsum(rng.normal(0.0, 0.005, (T, U)), axis=0), 0.0, 1.0)
    return np.ascontiguousarray(mu), np.ascontiguousarray(mv)


def kernel_src_sha():
    """Hash of the device code: PMC records are only valid for the sources they were measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = ROOT / "viewport-entropy-toolkit_amd" / "csrc"
    for f in sorted(csrc.glob("*.hpp")) + sorted(csrc.glob("*.hip")):
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(mu, mv, tcs, mode, weighted, budget_s, W=100, H=200):
    """C port of the reference path (oracle/vet_oracle.c) on a bounded sample of the same workload:
    one thread (the reference itself is single-threaded Python) and OpenMP over the box's core share
    (spatial: frames; transition: the nearest-tile sweep of the samples, then the rows).  The real
    reference's own figures (measured in the build container) are quoted next to them.
    Returns (record, (frames, entropy, indices)): the outputs of the single-thread run are not thrown
    away, the parity gate compares the engine's series with them."""
    from oracle import c_port
    c_port.load()
    T, U = mu.shape
    fn = (lambda a, b: c_port.spatial_series(a, b, W, H, tcs, use_weight_distribution=weighted)[:2]) \
        if mode == "spatial" else (lambda a, b: c_port.transition_series(a, b, W, H, tcs))

    def timed(threads, budget):
        c_port.set_threads(threads)
        probe = max(2, min(T, 4096 * threads // max(U, 1) + 2))
        t0 = time.perf_counter()
        fn(mu[:probe], mv[:probe])
        dt = time.perf_counter() - t0
        frames = int(max(probe, min(T, probe * budget / max(dt, 1e-6))))
        t0 = time.perf_counter()
        res = fn(mu[:frames], mv[:frames])
        dt = time.perf_counter() - t0
        return frames, dt, res

    frames, dt, res = timed(1, budget_s * 0.6)
    out = {"value": frames * U / dt, "unit": "samples/s", "frames_per_s": frames / dt, "cores": 1, "kind": "port",
           "sample": f"first {frames} of {T} frames x {U} users of the same workload, "
                     f"oracle/vet_oracle.c (gcc -O2, scalar FP64), {dt:.1f} s",
           "host_cpus": os.cpu_count(), "cpu_model": _cpu_model(),
           "reference_python": "viewport-entropy-toolkit itself, 1 core of a Xeon 2.1 GHz (BASELINE.md): "
                               "753 samples/s spatial at 51 tiles, 144 pair-samples/s transition at 201 tiles"}
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # every core this process may use; a GPU box hands a one-GPU job a CPU share (16 cores) of a 256-core host through
    # its cgroup, where 256 threads only fight each other — both thread counts are timed, the faster one is reported
    tried = []
    for share in sorted({min(16, aff), aff}):
        if share > 1:
            f2, d2, _ = timed(share, budget_s * 0.2)
            tried.append({"value": f2 * U / d2, "unit": "samples/s", "frames_per_s": f2 / d2, "cores": share,
                          "sample": f"first {f2} frames, OpenMP over "
                                    f"{'frames' if mode == 'spatial' else 'samples (nearest-tile sweep) and rows'}, {d2:.1f} s"})
    c_port.set_threads(1)
    if tried:
        best = max(tried, key=lambda r: r["value"])
        out["all_cores"] = dict(best, affinity_cpus=aff, host_cpus=os.cpu_count(), thread_counts_tried=tried)
    return out, (frames, res[0], res[1])


GOLDEN = {"spatial": ("g4_spatial.npz", "w_tc50", [50]), "transition": ("g5_transition.npz", "tc200", [200])}


def _max_rel(got, ref):
    """max |got - ref| / |ref| with nan == nan (a nan on one side only is inf); both exactly 0 count as 0."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if got.shape != ref.shape:
        return float("inf")
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    if (nan_g != nan_r).any():
        return float("inf")
    ok = ~nan_r
    if not ok.any():
        return 0.0
    den = np.where(ref[ok] == 0.0, 1.0, np.abs(ref[ok]))
    return float(np.max(np.abs(got[ok] - ref[ok]) / den))


def golden_through_hip(eng, mode):
    """One committed golden of the real reference (tests/golden/, written by oracle/gen_golden.py from the live
    reference) through the product path in THIS process: the golden's raw tracks -> the product's ingest
    (_ingest.build_dense) -> the HIP engine -> compared with the reference's own outputs.  Spatial: G4 w_tc50 under both
    weighted formulations (sweep and table); transition: G5 tc200.  Returns (ok, text, max_rel)."""
    from viewport_entropy_toolkit import _ingest, _native, _quantiser
    fname, tag, tcs = GOLDEN[mode]
    g = np.load(ROOT / "tests" / "golden" / fname, allow_pickle=False)
    order = [int(str(c)[4:]) for c in g[f"{tag}__columns"]]
    _, mu, mv = _ingest.build_dense([(g["time_in"][u], g["mu_in"][u], g["mv_in"][u]) for u in order])
    plan = _native.Plan(eng, [_quantiser.lattice_xyz(tc) for tc in tcs], 120.0, 2.0, True, 100, 200)
    worst, bad = 0.0, 0
    try:
        if mode == "spatial":
            for policy in (-1, 1):
                plan.set_table_policy(policy)
                res = plan.spatial(mu=mu, mv=mv, want_assign=True)
                bad += int(np.count_nonzero(res["assign"] != g[f"{tag}__assign"]))
                worst = max(worst, _max_rel(res["entropy"], g[f"{tag}__entropy"]))
        else:
            res = plan.transition(mu=mu, mv=mv, want_pairs=True)
            bad += int(np.count_nonzero(res["pairs"] != g[f"{tag}__pairs"]))
            worst = max(worst, _max_rel(res["entropy"], g[f"{tag}__entropy"]))
    finally:
        plan.close()
    ok = bad == 0 and worst <= 1e-6
    name = f"{fname.split('_')[0]}:{tag}"
    return ok, (f"{name} ok" if ok else f"{name} FAILED ({bad} index mismatches, entropy max rel {worst:.3g})"), worst


def parity_gate(eng, mode, tcs, weighted, W, H, mu_h, mv_h, ent_dev, idx_dev, port_out, n_frames):
    """SURVEY.md §8d 'parity gates in the same run': the series the timed steps produced (still in device memory)
    against the C port of the reference path on the first frames of the same video — tile indices bit-exact, entropy
    within 1e-6 relative, nan == nan — plus one golden of the real reference through the HIP path.  `port_out` = the
    outputs the cpu_baseline leg already computed (frames, entropy, indices); without it the port runs on `n_frames`
    frames here.  Replaces the per-frame loops of analyzers/spatial_entropy.py:107-164 / transition_entropy.py:107-175."""
    T = mu_h.shape[0]
    if n_frames <= 0:                    # --parity-frames 0: EVERY frame of the timed video, on the box's CPU share
        n_frames, port_out = T, None
    if port_out is None:
        from oracle import c_port
        c_port.load()
        aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        c_port.set_threads(1 if n_frames <= 64 else min(16, aff))
        take = min(T, n_frames + (1 if mode == "transition" else 0))
        if mode == "spatial":
            p_ent, p_idx = c_port.spatial_series(mu_h[:take], mv_h[:take], W, H, tcs, use_weight_distribution=weighted)[:2]
        else:
            p_ent, p_idx = c_port.transition_series(mu_h[:take], mv_h[:take], W, H, tcs)
        c_port.set_threads(1)
    else:
        _, p_ent, p_idx = port_out
    rows = len(p_ent)
    e = ent_dev[:rows].cpu().numpy()
    i = idx_dev[:rows].cpu().numpy()
    mism = int(np.count_nonzero(i != p_idx))
    rel = _max_rel(e, p_ent)
    g_ok, g_text, g_rel = golden_through_hip(eng, mode)
    ok = mism == 0 and rel <= 1e-6 and g_ok
    return {"ok": bool(ok), "frames": int(rows), "assign_mismatches": mism, "entropy_max_rel": rel, "golden": g_text,
            "golden_entropy_max_rel": g_rel, "tolerance": {"indices": "bit-exact", "entropy_rel": 1e-6},
            "against": "oracle/vet_oracle.c (C port of the reference path, pinned by tests/test_oracle_c.py against the "
                       "reference's goldens) on the first frames of the timed video; the engine's series are the ones the "
                       "timed steps left in device memory",
            "indices": "nearest tile of tile_counts[0] per sample" if mode == "spatial" else "(prior, current) tile pairs of tile_counts[0]"}


def expected_step_ms(workload, mode, strong, world, U, T_total, kernel_ms_max, gather_ms_max, pipelined):
    """The model of an N-GPU step, printed beside the measurement.
    Weak scaling (one video per rank, BASELINE config 4): no data-path collective, every rank runs the 1-GPU kernel on its
    own video; the gather of the entropy series (80 KB per rank at config 4) is enqueued behind the kernel and overlaps
    the next step's kernel: step = max(kernel + enqueue, gather) -> efficiency 1 as long as the gather is shorter than
    the kernel (0.142 ms at config 4).
    Strong scaling of transition mode (one video cut into N frame blocks, BASELINE config 5): k_transition_run takes
    9 us + 6.8 us per row of a persistent workgroup (tools/transition_scaling.py) on 8 x 256 workgroup slots, so the
    kernel of a rank with R/N rows takes 9 + 6.8 * ceil(R / N / 2048) us — 43 us at N = 1, 29 at 2, 23 at 4, 16 at 8
    (one row per workgroup is the floor) — and the step can never be shorter than the gather: config 5 is too small to
    scale (48.7 us on ONE GPU); it is reported, not tuned for."""
    enqueue_ms = 0.006                                   # host enqueue + stream gaps of a step (1-GPU: step - kernel)
    if strong and mode == "transition":
        rows = T_total - 1
        per_rank = -(-rows // world)
        slots = 8 * 256
        kernel = (9.0 + 6.8 * -(-per_rank // slots)) * 1e-3 if U <= 512 else kernel_ms_max
        model = "9 us + 6.8 us x ceil(rows per rank / 2048 workgroup slots), + enqueue; step >= gather"
    else:
        kernel = kernel_ms_max                           # same shape per rank as at N = 1
        model = "1-GPU kernel per rank (this run's slowest rank) + enqueue; the gather overlaps the next step's kernel"
    body = kernel + enqueue_ms
    step = max(body, gather_ms_max) if pipelined else body + gather_ms_max
    return {"step_ms": step, "kernel_ms": kernel, "gather_ms": gather_ms_max, "enqueue_ms": enqueue_ms, "model": model,
            "limiter": "gather" if (pipelined and gather_ms_max > body) else "kernel"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of ONE node, one rank per GPU (N > 1 without an outer launcher starts its own ranks). The two "
                         "BASELINE.json scaling configs: `--gpus N --workload config4` (one video per GPU, one RCCL gather: weak "
                         "scaling) and `--gpus N --workload config5 --shard frames` (one video cut along the frame axis with a "
                         "one-frame halo: strong scaling); the N > 1 line carries per_rank step / kernel / gather times")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS),
                    help="config2..config5 = BASELINE.json configs[1..4] (config3 = the single-GPU roofline config, the default); "
                         "config3u = config3 with use_weight_distribution=False; *x64 = 64 videos per GPU in one launch")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the drop-in API (host arrays -> DataFrame) timing")
    ap.add_argument("--data", default="random_walk", choices=["random_walk", "uniform", "clustered", "single", "fixed", "uniform_half", "uniform_quarter"],
                    help="synthetic sample distribution (default: SURVEY §8d random walks)")
    ap.add_argument("--loop", action="store_true", help="batched workloads: one call per video instead of one launch")
    ap.add_argument("--policy", type=int, default=0, choices=[-1, 0, 1],
                    help="table policy of the plan (include/vet.h): 0 by call size (default), 1 table, -1 sweep")
    ap.add_argument("--grid", default="100x200", metavar="WxH",
                    help="pixel grid of the quantiser (AnalyzerConfig.video_width x video_height; default: the reference's "
                         "100x200 = 20 301 directions; its README example is 200x400)")
    ap.add_argument("--parity-frames", type=int, default=64,
                    help="frames of the timed video the parity gate checks against the C port when the cpu_baseline leg "
                         "(whose larger sample it otherwise reuses) does not run; 0 = EVERY frame of the timed video (OpenMP "
                         "port on the box's CPU share: ~20 s at config 3)")
    ap.add_argument("--inject-fault", default="none", choices=["none", "assign", "entropy", "table"],
                    help="TEST HOOK of the parity gate: corrupt one nearest-tile word / one entropy value of the engine's "
                         "output after the timed region, or build the engine's plan on a wrong lattice; bench.py must exit non-zero")
    ap.add_argument("--no-variants", action="store_true", help="skip the nearest-tile / FP64-weights variants (N = 1)")
    ap.add_argument("--shard", default="videos", choices=["videos", "frames"],
                    help="N > 1: one video per GPU (weak scaling, default) or ONE video cut along the frame "
                         "axis with a 1-frame halo in transition mode (strong scaling, BASELINE config 5)")
    args = ap.parse_args()

    # --gpus N without an outer launcher: start the N ranks ourselves, as a CHILD process and before
    # anything here touches torch or HIP (a process that has initialised the GPU must not exec).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would report the wrong n_gpus")
    multi = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # under torch.distributed.run: RCCL path
    # VET_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks share devices, the gather goes through host memory); the real runs use RCCL.
    backend = os.environ.get("VET_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from viewport_entropy_toolkit import _dist, _native, _quantiser

    U, T, tcs, mode, weighted = WORKLOADS[args.workload]
    try:
        W, H = (int(v) for v in args.grid.lower().split("x"))
    except ValueError:
        raise SystemExit(f"--grid {args.grid!r}: expected WxH")
    # the lattices the ENGINE is built on: the workload's, unless the parity gate's test hook asks for a wrong table
    plan_tcs = list(tcs)
    if args.inject_fault == "table":
        plan_tcs[0] += 2
    T_total = T
    strong = args.shard == "frames" and world > 1
    if strong:
        # every rank synthesises the same video and keeps its block of rows (+ halo frame)
        mu_h, mv_h = synth_video(U, T, args.seed, 0, args.data)
        if mode == "transition":
            r0, r1, f0, f1 = _dist.transition_frame_block(T, rank, world)
        else:
            f0, f1 = _dist.frame_shard(T, rank, world)
        mu_h, mv_h = np.ascontiguousarray(mu_h[f0:f1]), np.ascontiguousarray(mv_h[f0:f1])
        T = f1 - f0
    else:
        mu_h, mv_h = synth_video(U, T, args.seed, rank, args.data)
    mu = torch.from_numpy(mu_h).to(dev)
    mv = torch.from_numpy(mv_h).to(dev)
    R = T if mode == "spatial" else T - 1
    ent = torch.empty(R, dtype=torch.float64, device=dev)
    # nearest-tile output of lattice 0: [T,U] int32 (spatial) / [(T-1),U,2] int32 (transition)
    idx = torch.empty((T, U) if mode == "spatial" else (R, U, 2), dtype=torch.int32, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    # frame shards differ by at most one row: gather buffers of the largest shard
    Rg = R if not strong else (T_total - (1 if mode == "transition" else 0) + world - 1) // world
    send = ent if Rg == R else torch.zeros(Rg, dtype=torch.float64, device=dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    gathered = [torch.empty(Rg, dtype=torch.float64, device=cdev) for _ in range(world)] if (multi and rank == 0) else None

    eng = _native.Engine(dev_index)
    t0 = time.perf_counter()
    plan = _native.Plan(eng, [_quantiser.lattice_xyz(tc) for tc in plan_tcs], 120.0, 2.0, weighted, W, H)
    eng.synchronize()
    plan_ms = (time.perf_counter() - t0) * 1e3
    # which device every rank's engine computes on (PCI bus id through the C-ABI), gathered on all ranks; under RCCL two
    # ranks on one device abort the run here, before anything is timed
    place = _dist.placement(eng)
    if args.policy:
        plan.set_table_policy(args.policy)
    # everything of a step — the engine's kernels, the copy into the gather buffer, the RCCL gather — is
    # enqueued on ONE explicit stream, so the gather is ordered after the kernel that produces its input
    run_stream = torch.cuda.Stream(device=dev)
    stream = run_stream.cuda_stream
    torch.cuda.synchronize()

    n_batch = BATCH.get(args.workload, 1)
    if n_batch > 1:
        # n_batch videos of the workload's shape per GPU (different seeds), resident in HBM
        import ctypes as C
        mus = [torch.from_numpy(synth_video(U, T, args.seed, rank * n_batch + v, args.data)[0]).to(dev) for v in range(n_batch)]
        mvs = [torch.from_numpy(synth_video(U, T, args.seed, rank * n_batch + v, args.data)[1]).to(dev) for v in range(n_batch)]
        ents = [torch.empty(R, dtype=torch.float64, device=dev) for _ in range(n_batch)]
        idxs = [torch.empty((T, U) if mode == "spatial" else (R, U, 2), dtype=torch.int32, device=dev) for _ in range(n_batch)]
        vids = (_native.Video * n_batch)(*[_native.Video(mus[v].data_ptr(), mvs[v].data_ptr(), U, T, ents[v].data_ptr(),
                                                         idxs[v].data_ptr(), None) for v in range(n_batch)])

    pipelined = multi and backend == "nccl" and n_batch == 1 and not os.environ.get("VET_BENCH_SYNC_GATHER")
    ent_bufs = [ent, torch.empty_like(ent)] if pipelined else [ent]
    send_bufs = [send, (torch.zeros_like(send) if send is not ent else ent_bufs[1])] if pipelined else [send]
    in_flight = [None, None]
    step_no = [0]

    def step():
        if n_batch > 1:
            if args.loop:
                for v in range(n_batch):
                    if mode == "spatial":
                        plan.spatial_device(mus[v].data_ptr(), mvs[v].data_ptr(), U, T, ents[v].data_ptr(),
                                            d_assign=idxs[v].data_ptr(), d_status=status.data_ptr(), stream=stream)
                    else:
                        plan.transition_device(mus[v].data_ptr(), mvs[v].data_ptr(), U, T, ents[v].data_ptr(),
                                               d_pairs=idxs[v].data_ptr(), d_status=status.data_ptr(), stream=stream)
            elif mode == "spatial":
                plan.spatial_batch_device(vids, d_status=status.data_ptr(), stream=stream)
            else:
                plan.transition_batch_device(vids, d_status=status.data_ptr(), stream=stream)
            if multi:
                dist.gather(ents[0], gathered, dst=0)
            return
        # RCCL: the gather of step i runs on the process group's stream while the kernel of step i + 1 computes
        # into the other entropy buffer; a buffer is handed to a kernel again only after the gather that read it
        # two steps earlier has completed (a stream-level wait, the host does not block)
        b = step_no[0] & 1 if pipelined else 0
        step_no[0] += 1
        e_buf, s_buf = ent_bufs[b], send_bufs[b]
        if in_flight[b] is not None:
            in_flight[b].wait()
            in_flight[b] = None
        if mode == "spatial":
            plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, e_buf.data_ptr(), d_assign=idx.data_ptr(),
                                d_status=status.data_ptr(), stream=stream)
        else:
            plan.transition_device(mu.data_ptr(), mv.data_ptr(), U, T, e_buf.data_ptr(), d_pairs=idx.data_ptr(),
                                   d_status=status.data_ptr(), stream=stream)
        if multi:
            if s_buf is not e_buf:
                s_buf[:R].copy_(e_buf)
            if pipelined:
                in_flight[b] = dist.gather(s_buf, gathered, dst=0, async_op=True)
            else:
                dist.gather(s_buf if backend == "nccl" else s_buf.cpu(), gathered, dst=0)

    def fence():
        for b in range(2):
            if in_flight[b] is not None:
                in_flight[b].wait()
                in_flight[b] = None
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    table_build_ms = None
    with torch.cuda.stream(run_stream):
        # the first call of a plan builds what it needs on the device (row statistics, weight table, per-direction
        # records: k_row_stats + k_wtab + k_dirrec, hipEvent-timed by the engine); reported beside plan_build_ms
        eng.profile_enable(True)
        eng.profile_reset()
        t_first = time.perf_counter()
        step()
        fence()
        first_step_ms = (time.perf_counter() - t_first) * 1e3
        table_build_ms = eng.profile_get("k_wtab")[0]
        for _ in range(max(args.warmup - 1, 0)):
            step()
        fence()
        eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    # ---- outside the timed region: what a step's time is made of on THIS rank (kernel time from the engine's
    # hipEvents above; the RCCL gather alone, hipEvents around an in-order gather of the step's series)
    gather_ms = None
    if multi:
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        with torch.cuda.stream(run_stream):
            fence()
            g0.record(run_stream)
            for _ in range(reps):
                buf = send_bufs[0] if n_batch == 1 else ents[0]
                dist.gather(buf if backend == "nccl" else buf.cpu(), gathered, dst=0)
            g1.record(run_stream)
            fence()
        gather_ms = g0.elapsed_time(g1) / reps
    # the practical ceiling beside the 8 TB/s nominal: a device-to-device copy that moves the workload's algorithmic
    # bytes (read + write), timed once with hipEvents
    copy_gbps = None
    if rank == 0:
        alg_probe = int((16 + (4 if mode == "spatial" else 8)) * U * T + 8 * R)
        half = max(alg_probe // 2 // 8, 1)
        src_c = torch.empty(half, dtype=torch.float64, device=dev).fill_(1.0)
        dst_c = torch.empty_like(src_c)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(4):
            c0.record(); dst_c.copy_(src_c); c1.record()
            torch.cuda.synchronize()
            t = c0.elapsed_time(c1)
            best = t if best is None else min(best, t)
        copy_gbps = 2.0 * half * 8 / (best * 1e-3) / 1e9
        del src_c, dst_c
    kname = "k_spatial" if mode == "spatial" else "k_transition"
    if mode == "transition":
        formulation = "k_transition: nearest-tile LUT gather + LDS bucket statistics"
    elif not weighted:
        formulation = "k_spatial_u(_lds): nearest-tile LUT + integer histogram stream"
    elif plan.last_formulation(0) == "table":
        formulation = (f"k_spatial_lut: direction weight table gather over the distinct directions of a frame (rows of "
                       f"{plan.table_stride(0)} entries, {len(tcs)} lattice(s) fused in one launch)")
    else:
        formulation = f"k_spatial_w: brute-force FP64 sweep ({plan.last_formulation(0)})"
    k_ms, k_n = eng.profile_get(kname)
    fin_ms, fin_n = eng.profile_get("k_finalize")
    eng.profile_enable(False)

    per_rank = None
    if multi:
        # one small all_gather after the timed region: [step time, kernel time per step, gather alone] of every rank
        mine = torch.tensor([elapsed / args.steps * 1e3, (k_ms + fin_ms) / args.steps, gather_ms or 0.0], dtype=torch.float64, device=cdev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu().numpy()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert int(status.sum().item()) == 0, "engine flagged out-of-range samples or empty frames"
    e_host = (ents[-1] if n_batch > 1 else ent).cpu().numpy()
    assert np.isfinite(e_host).all()

    # ---- parity gate, outside the timed region, on EVERY rank: the series of the timed steps against the
    # C port on the first frames of this rank's video, and one reference golden through the HIP path
    chk_ent, chk_idx = (ents[0], idxs[0]) if n_batch > 1 else (ent, idx)
    chk_mu, chk_mv = (mu_h, mv_h) if (n_batch == 1 or rank == 0) else synth_video(U, T, args.seed, rank * n_batch, args.data)
    if args.inject_fault == "assign":
        chk_idx.view(-1)[U * 3 + 5] ^= 1
    elif args.inject_fault == "entropy":
        chk_ent[min(7, R - 1)] *= 1.0 + 3e-6
    cpu_rec, port_out = None, None
    if world == 1 and not args.no_cpu_baseline:
        cpu_rec, port_out = cpu_baseline(mu_h, mv_h, tcs, mode, weighted, args.cpu_seconds, W, H)
    parity = parity_gate(eng, mode, tcs, weighted, W, H, chk_mu, chk_mv, chk_ent, chk_idx, port_out, args.parity_frames)
    if multi:
        # one all_reduce joins the ranks' verdicts: [failed, index mismatches, max relative entropy error, frames]
        flag = torch.tensor([0.0 if parity["ok"] else 1.0, float(parity["assign_mismatches"])], dtype=torch.float64, device=cdev)
        worst = torch.tensor([parity["entropy_max_rel"], parity["golden_entropy_max_rel"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        parity.update(ok=bool(flag[0].item() == 0), ranks_failed=int(flag[0].item()), assign_mismatches=int(flag[1].item()),
                      entropy_max_rel=float(worst[0].item()), golden_entropy_max_rel=float(worst[1].item()),
                      frames_per_rank=parity["frames"], ranks_checked=world)
    if not parity["ok"]:
        # a wrong result has no throughput: no metric line, the verdict on stderr, non-zero exit on every rank
        if rank == 0:
            print("PARITY GATE FAILED: " + json.dumps(parity), file=sys.stderr, flush=True)
        plan.close()
        if multi:
            dist.destroy_process_group()
        raise SystemExit(3)
    if pipelined:       # both buffers hold the same series (same input every step)
        assert np.array_equal(ent_bufs[1].cpu().numpy(), e_host, equal_nan=True)
        if rank == 0:
            assert np.array_equal(gathered[0][:R].cpu().numpy(), e_host, equal_nan=True), "gathered series differs from the local one"

    if rank == 0:
        samples_per_step = (U * T_total if strong else U * T * world) * n_batch
        ms_per_step = elapsed / args.steps * 1e3
        # algorithmic bytes: 16 B in + 4 B (8 B transition) out per sample and 8 B of
        # entropy per frame, the samples counted once whatever the number of lattices; per launch of
        # the dominant kernel = per step / launches per step
        form = plan.last_formulation(0) if (mode == "spatial" and weighted) else None
        # the arithmetic the path computes in (not a precision claim): entropies are FP64 everywhere
        dtype = {"table": "f64 (u32 block-floating-point table weights, u64 fixed-point histogram)",
                 "ftable": "f64 (f32 table weights scaled per row, f64 histogram)",
                 "sweep": "f64 (weights evaluated in f64, 2^-52 fixed-point u64 histogram)",
                 "precise": "f64"}.get(form, "f64 (int32 tile counts, f64 entropy)")
        # what the dominant kernel is bound by; `frac` stays the fraction of the HBM roofline (BASELINE.json's metric)
        bound = {"table": "l2_gather+lds_atomic", "ftable": "l2_gather+lds_atomic", "sweep": "fp64_valu", "precise": "fp64_valu"}.get(
            form, "hbm" if mode == "spatial" else "lds_atomic+issue")
        per_sample_out = 4 if mode == "spatial" else 8
        alg_bytes_step = ((16 + per_sample_out) * U * T + 8 * R) * n_batch
        launches_per_step = max(k_n / args.steps, 1.0)
        alg_bytes_launch = alg_bytes_step / launches_per_step
        avg_kernel_ms = k_ms / max(k_n, 1)
        achieved = alg_bytes_launch / (avg_kernel_ms * 1e-3) / 1e9 if k_n else None
        # PMC traffic is collected in separate rocprofv3 --pmc passes (tools/pmc.sh) and recorded with the
        # hash of the kernel sources it was measured on; a stale record is dropped, not replayed
        traffic, traffic_src = None, None
        tf = ROOT / "profiles" / "pmc_traffic.json"
        if tf.exists() and args.data == "random_walk":
            try:
                rec = json.loads(tf.read_text()).get(args.workload, {})
                if rec.get("kernel_src_sha") == kernel_src_sha():
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_src = {"file": "profiles/pmc_traffic.json", "kernel_ms_at_collection": rec.get("kernel_ms"),
                                   "kernel_src_sha": rec.get("kernel_src_sha")}
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "viewport samples/sec", "value": samples_per_step / (ms_per_step * 1e-3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic" if args.data == "random_walk" else f"synthetic ({args.data})",
            "config": {"workload": f"{args.workload}: {1 if strong else world} video(s) x {U} users x "
                                   f"{T_total if strong else T} frames, "
                                   f"tile_counts={tcs}, {mode}, "
                                   f"use_weight_distribution={weighted}, fov=120, W={W}, H={H}",
                       "users": U, "frames": T_total if strong else T, "tile_counts": tcs, "mode": mode, "grid": [W, H],
                       "videos_per_gpu": n_batch, "batched_launch": bool(n_batch > 1 and not args.loop), "parallelism": (f"one video cut into {world} frame blocks" if strong
                                       else f"one video per GPU x{world}")},
            "frames_per_s": (T_total if strong else R * world) * n_batch / (ms_per_step * 1e-3),
            "roofline": {"bound": bound, "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                         "frac_is": "algorithmic bytes / kernel time / HBM peak (the HBM roofline fraction), whatever `bound` says",
                         "measured_copy_ceiling": {"GBps": copy_gbps, "what": "device-to-device copy moving the workload's "
                                                   "algorithmic bytes (read + write), hipEvent-timed outside the timed region",
                                                   "frac_of_copy": (achieved / copy_gbps) if (achieved and copy_gbps) else None},
                         "traffic": traffic, "traffic_provenance": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "avg_kernel_ms": avg_kernel_ms, "launches": k_n,
                         "note": ("weighted FoV histogram: on top of the algorithmic bytes the gather moves 6 B per in-FoV "
                                  "tile of every distinct direction of a frame from L2 / Infinity Cache and issues one "
                                  "LDS atomic per entry; see `secondary` for that formulation's own speed of light "
                                  "(DESIGN.md §5)") if (mode == "spatial" and weighted) else
                                 ("transition mode: bytes are not the limiter (traffic = 1.14 x algorithmic); the per-row LDS "
                                  "work of the five barrier-separated steps is: see `secondary`") if mode == "transition" else
                                 "integer histogram stream: HBM-bound"},
            "kernel_ms_per_step": {kname: k_ms / args.steps, "k_finalize": fin_ms / args.steps},
            "formulation": formulation,
            "plan_build_ms": plan_ms,
            # built inside the plan's first call (before the timed region): k_row_stats + k_wtab + k_dirrec, hipEvents
            "table_build_ms": table_build_ms, "first_call_ms": first_step_ms,
            "parity": parity,
        }
        if mode == "spatial" and weighted:
            # the arithmetic choice, auditable from this line: the proven worst-case |dH|/H of the integer formulations for
            # EVERY possible frame of this plan (vet_plan_error_bounds, k_row_stats), per lattice; the contract is 1e-6
            bounds = [plan.error_bounds(k) for k in range(len(tcs))]
            pick = 0 if form in ("table", "ftable") else 1
            out["formulation_bound"] = {
                "formulation": form, "proven_rel_entropy_error": (max(b[pick] for b in bounds) if form in ("table", "sweep") else None),
                "per_lattice": {"table": [b[0] for b in bounds], "sweep": [b[1] for b in bounds]},
                "contract": 1e-6, "engine_threshold": 1e-7,
                "what": "bound on |H_formulation - H_exact| / H_exact over every frame the plan can be given (<= 1024 users for the "
                        "sweep), computed on the device from the plan's own weight rows; ftable / precise sum FP64 weights instead "
                        "(ftable: 1.2e-7 by construction)"}
            if plan.table_rows():
                out["table"] = {"rows": plan.table_rows(), "stride": plan.table_stride(0), "directions": plan.n_dirs,
                                "bytes": (plan.table_rows() + 1) * plan.table_stride(0) * 6}
        if per_rank is not None:
            # attribution of a scaling point: per-rank step / kernel / gather times (ms) and their spread
            out["per_rank"] = {"step_ms": per_rank[:, 0].tolist(), "kernel_ms": per_rank[:, 1].tolist(),
                               "gather_ms": per_rank[:, 2].tolist(),
                               "step_ms_max_over_min": float(per_rank[:, 0].max() / max(per_rank[:, 0].min(), 1e-9)),
                               "kernel_ms_max_over_min": float(per_rank[:, 1].max() / max(per_rank[:, 1].min(), 1e-9)),
                               "gather": f"one {'RCCL' if backend == 'nccl' else backend} gather of {Rg} FP64 per rank and step"
                                         + (", overlapped with the next step's kernel inside the timed region" if pipelined else ""),
                               "note": "gather_ms is the gather alone (in order, after the timed region); step_ms is each rank's own clock"}
            # what the curve should look like, so that a SCALE point can be judged the moment it exists
            # placement, attested by every rank's engine context (vet_device_pci_bus_id) and checked before the timed region
            out["per_rank"].update({
                "device_pci_bus_id": [r["pci_bus_id"] for r in place["ranks"]],
                "device_index": [r["device_index"] for r in place["ranks"]],
                "host": [r["host"] for r in place["ranks"]],
                "visible_devices": [r["visible_devices"] for r in place["ranks"]],
                "distinct_devices": place["distinct"], "n_devices": place["n_devices"],
                "backend": place["backend"], "rccl_version": place["rccl_version"],
                "torch_device_count": torch.cuda.device_count()})
            out["per_rank"]["expected"] = expected_step_ms(args.workload, mode, strong, world, U, T_total,
                                                            float(per_rank[:, 1].max()), float(per_rank[:, 2].max()), pipelined)
        # besides the HBM figure, say what the run formulation is really bound by
        if mode == "spatial" and weighted and k_n:
            n_lat = [2 * (tc // 2) + 1 for tc in tcs]
            cap = (1.0 - np.cos(np.radians(120.0 / 2.0))) / 2.0          # share of tiles inside the FoV cap
            if plan.last_formulation(0) == "table":
                # speed of light of THIS formulation: a frame gathers one 6-byte entry per in-FoV tile of each
                # DISTINCT direction among its users and issues one 64-bit LDS atomic per entry
                dirs = plan.read_dirs()
                _, alias = np.unique(dirs + 0.0, axis=0, return_inverse=True)
                ids = (mv_h * H).astype(np.int64) * (W + 1) + (mu_h * W).astype(np.int64)
                rows = np.sort(alias.reshape(-1)[ids], axis=1)
                distinct = 1 + (np.diff(rows, axis=1) != 0).sum(1)
                # how many row loads a walk over GROUPS of F consecutive frames could share (distinct rows of a group /
                # sum of its frames' distinct rows; first 2048 frames): the reuse the round-4 joint-walk experiment had
                # to live on (profiles/r04/v4_two_kernel_joint_walk_experiment.log)
                reuse = {}
                head = rows[:2048]
                for F in (2, 4, 8, 16):
                    g = head[: len(head) // F * F].reshape(-1, F * head.shape[1])
                    g = np.sort(g, axis=1)
                    union = 1 + (np.diff(g, axis=1) != 0).sum(1)
                    reuse[str(F)] = float(union.sum() / max(distinct[: len(head) // F * F].sum(), 1))
                entries = float(distinct.sum()) * cap * sum(n_lat) * n_batch / launches_per_step
                cu, clk = 256, 2.4e9
                lds_rate = 64 / 6.6 * clk * cu                      # conflict-free ds_add_u64: 6.6 clk per wave (probed)
                t_l2, t_mall = 6 * entries / (70e9 * cu), 6 * entries / (33.5e9 * cu)
                t_lds = entries / lds_rate
                out["roofline"]["secondary"] = {
                    "bound": "row gather (6 B per entry through L2 / Infinity Cache) and one ds_add_u64 per entry",
                    "distinct_directions_per_frame": float(distinct.mean()), "users_per_frame": U,
                    "row_reuse_over_frame_groups": reuse,
                    "entries_per_launch": entries, "gathered_bytes_per_launch": 6 * entries,
                    "achieved": 6 * entries / (avg_kernel_ms * 1e-3) / 1e9 / cu, "unit": "GB/s per CU",
                    "guide_rates_GBps_per_cu": {"xcd_l2": [66, 73], "infinity_cache": 33.5, "hbm": [23, 24]},
                    "sol_ms": {"gather_all_l2_hits": t_l2 * 1e3, "gather_all_infinity_cache": t_mall * 1e3,
                               "lds_atomics_conflict_free": t_lds * 1e3},
                    "frac_of_sol": max(t_l2, t_lds) / (avg_kernel_ms * 1e-3)}
                # The path the kernel actually queues on (profiles/r06/config4_workgroup_timeline.txt): 128-byte
                # line requests of a CU's L1 to the L2 — a row's weight and tile lines (6 B per entry + one partial line at the
                # end of either array), one record line per sample, the sample stream and the assignment stores — against the
                # L1's outstanding-miss capacity, 64 slots / 296 cycles measured miss latency (profiles/r02/config3_pmc_stalls.json)
                present = float(np.count_nonzero(~np.isnan(mu_h) & ~np.isnan(mv_h))) * n_batch / launches_per_step
                row_lines = entries * 6.0 / 128.0 + float(distinct.sum()) * n_batch / launches_per_step
                lines = row_lines + present + present * 16.0 / 128.0 + present * 4.0 / 128.0
                ceiling = 64.0 / 296.0
                # measured L2 requests of the same kernel (TCC_REQ, tools/pmc3.sh), if the record is of these kernel sources
                measured, m_src = None, None
                qf = ROOT / "profiles" / "pmc_sq.json"
                if qf.exists() and args.data == "random_walk" and (W, H) == (100, 200):
                    try:
                        rec = json.loads(qf.read_text()).get(args.workload, {})
                        if rec.get("kernel_src_sha") == kernel_src_sha() and rec.get("TCC_REQ_sum"):
                            measured = float(rec["TCC_REQ_sum"])
                            m_src = {"file": "profiles/pmc_sq.json", "counter": "TCC_REQ_sum", "kernel_src_sha": rec["kernel_src_sha"]}
                    except Exception:  # noqa: BLE001
                        measured = None
                used = measured if measured else lines
                rate = used / (avg_kernel_ms * 1e-3 * clk * cu)
                out["roofline"]["secondary"]["l1_miss_path"] = {
                    "what": "128-byte line requests of the CUs' L1s per launch / kernel time, per clock and CU, against the L1's "
                            "outstanding-miss capacity (64 slots / 296 cycles per miss); measured = TCC_REQ of a rocprofv3 --pmc pass "
                            "on these kernel sources, modelled = row lines + one record line per sample + sample stream + stores",
                    "line_requests_per_launch": {"measured": measured, "modelled": lines, "provenance": m_src},
                    "row_lines_modelled": row_lines, "record_lines_modelled": present,
                    "lines_per_clock_per_cu": rate, "ceiling_lines_per_clock_per_cu": ceiling, "frac_of_ceiling": rate / ceiling,
                    "clock_assumed_hz": clk}
            else:
                flop = 13.5 * sum(n_lat) * U * T * n_batch / launches_per_step
                out["roofline"]["secondary"] = {
                    "bound": "fp64 valu (13.5 flop per tile and sample, SURVEY.md 8d)", "flop_per_launch": flop,
                    "achieved": flop / (avg_kernel_ms * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s"}
        if mode == "transition" and k_n:
            # Transition mode's own limiter: the per-row LDS work of k_transition_run (atomics on the per-tile words and the
            # bucket hash, five barrier-separated steps) — not bytes.  LDS wave instructions and LDS-busy cycles per launch
            # come from the SQ counter passes of tools/pmc_sq.sh (profiles/pmc_sq.json, valid for the kernel sources they
            # were collected on); the latency model is tools/transition_scaling.py's fit.
            rows_launch = R * n_batch / launches_per_step
            cu, clk, slots = 256, 2.4e9, 8 * 256
            sq, sq_src = None, None
            qf = ROOT / "profiles" / "pmc_sq.json"
            if qf.exists() and args.data == "random_walk":
                try:
                    rec = json.loads(qf.read_text()).get(args.workload, {})
                    if rec.get("kernel_src_sha") == kernel_src_sha():
                        sq, sq_src = rec, {"file": "profiles/pmc_sq.json", "kernel_src_sha": rec.get("kernel_src_sha")}
                except Exception:  # noqa: BLE001
                    sq = None
            sec = {"bound": "LDS wave instructions of the row's five steps and their round trips (16 waves per CU)",
                   "rows_per_launch": rows_launch,
                   "latency_model": {"what": "9 us + 6.8 us per row of a persistent workgroup, 2048 workgroup slots "
                                             "(tools/transition_scaling.py, 512 users)",
                                     "ms": (9.0 + 6.8 * rows_launch / slots) * 1e-3 if (U <= 512 and n_batch == 1) else None}}
            if sq:
                insts, active, conflict = sq["SQ_INSTS_LDS"], sq["SQ_LDS_IDX_ACTIVE"], sq["SQ_LDS_BANK_CONFLICT"]
                sec.update({
                    "lds_wave_instructions_per_row": insts / rows_launch,
                    "lds_cycles_per_instruction": {"measured": active / insts, "without_bank_conflicts": (active - conflict) / insts},
                    "sol_ms": {"lds_busy_as_measured": active / (cu * clk) * 1e3,
                               "lds_busy_conflict_free": (active - conflict) / (cu * clk) * 1e3},
                    "frac_of_sol": (active - conflict) / (cu * clk) / (avg_kernel_ms * 1e-3),
                    "lds_busy_share_of_kernel": active / (cu * clk) / (avg_kernel_ms * 1e-3),
                    "provenance": sq_src,
                    "reading": "the LDS is busy a third of the kernel: the row's dependent LDS round trips (a wave waits in "
                               "s_waitcnt for half of its life) bound it, not LDS throughput and not bytes"})
            out["roofline"]["secondary"] = sec
        if world == 1 and mode == "spatial" and weighted and n_batch == 1 and not args.no_variants:
            # the same video with use_weight_distribution=False (every user counts 1 on its nearest
            # tile): the HBM-streaming formulation of the path, reported beside the headline
            plan_u = _native.Plan(eng, [_quantiser.lattice_xyz(tc) for tc in tcs], 120.0, 2.0, False, W, H)
            torch.cuda.set_stream(run_stream)
            for _ in range(args.warmup):
                plan_u.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr(),
                                      d_status=status.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            eng.profile_enable(True)
            eng.profile_reset()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                plan_u.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr(),
                                      d_status=status.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            ku_ms, ku_n = eng.profile_get("k_spatial")
            eng.profile_enable(False)
            plan_u.close()
            per_launch = alg_bytes_step / max(ku_n / args.steps, 1.0)
            out["nearest_tile_variant"] = {
                "config": "same video, use_weight_distribution=False", "ms_per_step": dt * 1e3,
                "value": U * T / dt, "unit": "samples/s",
                "roofline": {"bound": "hbm", "kernel": "k_spatial (k_spatial_u_lds)",
                             "achieved": per_launch / (ku_ms / max(ku_n, 1) * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": per_launch / (ku_ms / max(ku_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "avg_kernel_ms": ku_ms / max(ku_n, 1), "launches": ku_n}}
        if world == 1 and mode == "spatial" and weighted and n_batch == 1 and not args.no_variants:
            # ---- what the same video costs with FP64 weights (the arithmetic the u32 table replaces), outside the timed region:
            # (a) exact FP64 weight rows summed per tile (the d_weights output: k_weights, calculate_tile_weights at the
            #     reference's precision, entropy_utils.py:131-136) on top of the entropy pass;
            # (b) policy -1: every sample sweeps every tile, weights evaluated in FP64 (acos / pow), no table at all
            reps = max(1, min(args.steps, 5))
            n0 = 2 * (tcs[0] // 2) + 1
            wts = torch.empty((T, n0), dtype=torch.float64, device=dev)
            ent_w = torch.empty_like(ent)

            def run_w(p):
                p.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent_w.data_ptr(), d_assign=idx.data_ptr(),
                                 d_weights=wts.data_ptr(), d_status=status.data_ptr(), stream=stream)

            def run_e(p):
                p.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent_w.data_ptr(), d_assign=idx.data_ptr(),
                                 d_status=status.data_ptr(), stream=stream)

            def clock(fn, p):
                fn(p)                                   # first call: builds what it needs (exact rows: k_wexact)
                torch.cuda.synchronize()
                eng.profile_enable(True)
                eng.profile_reset()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn(p)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                km = {k: eng.profile_get(k)[0] / reps for k in ("k_spatial", "k_weights", "k_finalize")}
                eng.profile_enable(False)
                return dt, km

            torch.cuda.set_stream(run_stream)
            dt_w, km_w = clock(run_w, plan)
            fp64 = {"exact_weight_rows": {
                "what": "entropy pass as timed + the weights pass: exact FP64 weights (ocml acos / pow) of every in-FoV tile, "
                        "summed per tile in FP64 (k_weights_gather; d_weights output [frames x tiles] written to HBM)",
                "ms_per_step": dt_w * 1e3, "value": U * T / dt_w, "unit": "samples/s",
                "kernel_ms": km_w, "weights_bytes_written": int(T) * n0 * 8}}
            if len(tcs) == 1:
                # audit of the arithmetic choice inside this run: the entropy recomputed on the host, in FP64, from those
                # exact weight sums (first frames) against the series the u32 table produced in the timed steps
                m = min(T, 256)
                w = np.abs(wts[:m].cpu().numpy())
                pr = w / w.sum(axis=1, keepdims=True)
                with np.errstate(divide="ignore", invalid="ignore"):
                    h = -np.where(pr > 0, pr * np.log2(pr), 0.0).sum(axis=1) / np.log2(n0)
                fp64["exact_weight_rows"]["entropy_from_fp64_weights_vs_timed_series_max_rel"] = _max_rel(e_host[:m], h)
                fp64["exact_weight_rows"]["frames_audited"] = m
            del wts
            if form != "sweep":
                plan.set_table_policy(-1)
                dt_s, km_s = clock(run_e, plan)
                form_s = plan.last_formulation(0)
                e_timed = torch.from_numpy(e_host).to(dev)       # (`ent` has been reused by the nearest-tile variant above)
                d = (ent_w - e_timed).abs() / e_timed.abs().clamp_min(1e-300)
                fp64["sweep"] = {"what": "policy -1: brute-force sweep, FP64 weights per (sample, tile), 2^-52 fixed-point histogram",
                                 "formulation": form_s, "ms_per_step": dt_s * 1e3, "value": U * T / dt_s, "unit": "samples/s",
                                 "kernel_ms": km_s, "entropy_vs_timed_series_max_rel": float(d.max().item())}
                plan.set_table_policy(args.policy)
            out["fp64_weights_variant"] = fp64
        if world == 1 and n_batch == 1 and not args.no_api:
            # the drop-in API itself, from host arrays to the result DataFrame (PCIe inclusive; never `value`)
            import tempfile
            from viewport_entropy_toolkit import AnalyzerConfig, SpatialEntropyAnalyzer, TransitionEntropyAnalyzer
            from viewport_entropy_toolkit.config import EntropyConfig
            with tempfile.TemporaryDirectory() as tmp:
                cls = SpatialEntropyAnalyzer if mode == "spatial" else TransitionEntropyAnalyzer
                an = cls(AnalyzerConfig(video_width=W, video_height=H, tile_counts=list(tcs), output_dir=Path(tmp),
                                        entropy_config=EntropyConfig(use_weight_distribution=weighted)))
                an.load_arrays(np.arange(T, dtype=np.float64) * 0.1, mu_h, mv_h)
                best = None
                for _ in range(4):
                    t0 = time.perf_counter()
                    df = an.compute_entropy()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                assert np.allclose(df["entropy"].to_numpy(), e_host, rtol=1e-6)
                t0 = time.perf_counter()
                cell = dict(df["tile_weights"][len(df) // 2])
                fetch_ms = (time.perf_counter() - t0) * 1e3
                t0 = time.perf_counter()
                for i in range(0, len(df), max(len(df) // 16, 1)):       # 16 more blocks of 256 frames
                    dict(df["tile_assignments"][i])
                next_ms = (time.perf_counter() - t0) * 1e3 / 16
                out["drop_in_api"] = {
                    "call": f"{cls.__name__}.compute_entropy(): host arrays -> DataFrame[time, entropy, tile_weights, "
                            "tile_assignments]; the dict columns stay in device memory until a cell is read",
                    "ms": best * 1e3, "samples_per_s": U * T / best, "first_cell_fetch_ms": fetch_ms, "later_cell_fetch_ms": next_ms,
                    "cell_len": len(cell), "note": "PCIe-inclusive (pageable numpy arrays); not the headline value"}
        if cpu_rec is not None:
            out["cpu_baseline"] = cpu_rec
        print(json.dumps(out), flush=True)
    plan.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
