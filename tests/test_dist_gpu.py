"""GPU: the multi-GPU plumbing with the HIP engine as the per-rank compute (SURVEY.md §4 v):
(i)  one rank under the `nccl` backend (= RCCL): `_dist.analyze_videos` / `transition_frame_sharded` go
     through the RCCL branch of `gather_series` and equal the plain loop bit for bit;
(ii) two ranks under `gloo`, both on device 0, each with its own engine context: videos sharded over the
     ranks and one video cut along the frame axis equal the one-process result bit for bit;
(iii) `bench.py --gpus 2` starts its own ranks (here rehearsed with the gloo backend on one device) and
     reports n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
TCS = [50, 100]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _paths():
    for p in (str(ROOT), str(ROOT / "viewport-entropy-toolkit_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _videos():
    from viewport_entropy_toolkit import _synthetic
    return [_synthetic.random_walk_video(24 + 8 * v, 90 + 11 * v, base_seed=21, video_id=v, p_absent=0.05) for v in range(5)]


def _make_plans():
    from viewport_entropy_toolkit import _native, _quantiser
    eng = _native.Engine(0)
    tiles = [_quantiser.lattice_xyz(tc) for tc in TCS]
    spatial = _native.Plan(eng, tiles, 120.0, 2.0, True, 100, 200)
    spatial.set_table_policy(1)
    return eng, spatial


def _worker(rank, world, port, q):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from viewport_entropy_toolkit import _dist, _synthetic
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng, plan = _make_plans()
        videos = _videos()
        spatial = lambda v: plan.spatial(mu=v[0], mv=v[1], want_assign=False)["entropy"]          # noqa: E731
        got = _dist.analyze_videos(videos, spatial)
        mu, mv = _synthetic.random_walk_video(40, 203, base_seed=8)
        trans = lambda a, b: plan.transition(mu=a, mv=b, want_pairs=False)["entropy"]             # noqa: E731
        cut = _dist.transition_frame_sharded(mu, mv, trans)
        cut_s = _dist.spatial_frame_sharded(mu, mv, lambda a, b: plan.spatial(mu=a, mv=b, want_assign=False)["entropy"])
        # every rank's share of the videos in ONE batched launch, one gather
        got_b = _dist.analyze_videos_batched(videos, lambda vs: [r["entropy"] for r in plan.spatial_batch(vs)])
        # placement: both ranks report device 0's PCI id; a job that requires one rank per GPU refuses it on every rank
        pl = _dist.placement(eng)
        ok_p = (pl["backend"] == "gloo" and pl["n_devices"] == 1 and not pl["distinct"] and len(pl["ranks"]) == world
                and pl["ranks"][0]["pci_bus_id"] == pl["ranks"][1]["pci_bus_id"] == eng.pci_bus_id())
        try:
            _dist.placement(eng, require_distinct=True)
            ok_p = False
        except _dist.PlacementError as e:
            ok_p = ok_p and "ranks [0, 1]" in str(e)
        if rank == 1:
            q.put(("placement", ok_p))
        if rank == 0:
            ok_t0 = ok_p
            ok_v = set(got) == set(range(len(videos))) and all(np.array_equal(got[v], spatial(videos[v])) for v in got)
            ok_t = ok_t0 and bool(np.array_equal(cut, trans(mu, mv)))
            ok_s = bool(np.array_equal(cut_s, plan.spatial(mu=mu, mv=mv, want_assign=False)["entropy"]))
            ok_b = set(got_b) == set(range(len(videos))) and all(np.array_equal(got_b[v], spatial(videos[v])) for v in got_b)
            q.put((ok_v, ok_t and ok_s and ok_b))
        plan.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_with_hip_compute():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = [q.get(timeout=10), q.get(timeout=10)]
    assert ("placement", True) in got, "rank 1: placement record wrong, or two ranks on one device were not refused"
    ok_v, ok_t = [g for g in got if g[0] != "placement"][0]
    assert ok_v, "videos sharded over two ranks differ from the one-process result"
    assert ok_t, "frame-sharded transition / spatial entropy or the batched video shares differ from the one-process result"


def _rccl_worker(port, q):
    _paths()
    import torch
    import torch.distributed as dist
    from viewport_entropy_toolkit import _dist, _synthetic
    torch.cuda.set_device(0)                       # torch's HIP runtime first, as in bench.py
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        eng, plan = _make_plans()
        videos = _videos()
        spatial = lambda v: plan.spatial(mu=v[0], mv=v[1], want_assign=False)["entropy"]          # noqa: E731
        got = _dist.analyze_videos(videos, spatial)
        ok_v = set(got) == set(range(len(videos))) and all(np.array_equal(got[v], spatial(videos[v])) for v in got)
        mu, mv = _synthetic.random_walk_video(40, 203, base_seed=8)
        trans = lambda a, b: plan.transition(mu=a, mv=b, want_pairs=False)["entropy"]             # noqa: E731
        ok_t = bool(np.array_equal(_dist.transition_frame_sharded(mu, mv, trans), trans(mu, mv)))
        fixed = _dist.gather_series(np.arange(9.0), max_len=9)
        ok_f = len(fixed) == 1 and bool(np.array_equal(fixed[0], np.arange(9.0)))
        pl = _dist.placement(eng)                    # world 1 under RCCL: a PCI id, the RCCL version, distinct by construction
        ok_f = ok_f and pl["backend"] == "nccl" and pl["distinct"] and bool(pl["rccl_version"]) \
            and ":" in pl["ranks"][0]["pci_bus_id"] and pl["ranks"][0]["device_index"] == 0
        q.put((ok_v, ok_t, ok_f))
        plan.close()
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_gather():
    """world_size 1 under the nccl backend (its own process, torch's runtime initialised first as in bench.py):
    the gather of gather_series runs through RCCL on device buffers."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert q.get(timeout=10) == (True, True, True)


def _stream_worker(q):
    _paths()
    import torch
    from viewport_entropy_toolkit import _synthetic
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng, plan = _make_plans()
    mu_h, mv_h = _synthetic.random_walk_video(64, 4000, base_seed=2)
    want = plan.spatial(mu=mu_h, mv=mv_h, want_assign=False)["entropy"]
    mu, mv = torch.from_numpy(mu_h).to(dev), torch.from_numpy(mv_h).to(dev)
    ok = []
    # (a) torch's default stream (handle 0 -> VET_STREAM_LEGACY): the clone is enqueued behind the kernel on the same stream
    ent = torch.zeros(4000, dtype=torch.float64, device=dev)
    plan.spatial_device(mu.data_ptr(), mv.data_ptr(), 64, 4000, ent.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    got = ent.clone()
    torch.cuda.synchronize()
    ok.append(bool(np.array_equal(got.cpu().numpy(), want)))
    # (b) an explicit non-default stream: kernel and consumer on that stream
    st = torch.cuda.Stream(device=dev)
    ent2 = torch.zeros(4000, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        plan.spatial_device(mu.data_ptr(), mv.data_ptr(), 64, 4000, ent2.data_ptr(), stream=st.cuda_stream)
        got2 = ent2.clone()
    st.synchronize()
    ok.append(bool(np.array_equal(got2.cpu().numpy(), want)))
    # (c) stream=None: the engine's own stream, made visible by Engine.synchronize()
    ent3 = torch.zeros(4000, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    plan.spatial_device(mu.data_ptr(), mv.data_ptr(), 64, 4000, ent3.data_ptr())
    eng.synchronize()
    ok.append(bool(np.array_equal(ent3.cpu().numpy(), want)))
    q.put(tuple(ok))
    plan.close()


def test_device_pointer_calls_on_torch_streams():
    """`stream` of the device-pointer entry points (include/vet.h): torch's default stream (handle 0) is the legacy null
    stream, an explicit stream is used as is, None is the engine's own stream."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_stream_worker, args=(q,))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert q.get(timeout=10) == (True, True, True)


@pytest.mark.parametrize("workload,shard", [("config2", "videos"), ("config5", "frames"), ("config4", "videos"), ("config2", "frames")])
def test_bench_launches_its_own_ranks(workload, shard):
    """`python bench.py --gpus 2` with no outer launcher: two ranks, n_gpus = 2 in the line (gloo rehearsal on
    one device; the driver's multi-GPU runs use RCCL, one rank per GPU)."""
    env = dict(os.environ, VET_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--workload", workload, "--shard", shard, "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["value"] > 0
    assert rec["scaling"] == ("strong" if shard == "frames" else "weak")
    pr = rec["per_rank"]            # attribution of a scaling point: every rank's step / kernel / gather time
    assert len(pr["step_ms"]) == len(pr["kernel_ms"]) == len(pr["gather_ms"]) == 2
    assert all(k > 0 for k in pr["kernel_ms"]) and all(g > 0 for g in pr["gather_ms"])
    assert pr["step_ms_max_over_min"] >= 1.0
    # the line attests its own placement: the device every rank's engine context ran on, the backend, and whether the
    # ranks held distinct devices (a gloo rehearsal on a one-GPU box shares device 0 and says so)
    assert len(pr["device_pci_bus_id"]) == 2 and all(len(d) >= 7 and ":" in d for d in pr["device_pci_bus_id"])
    assert pr["backend"] == "gloo" and pr["device_index"] == [0, 0]
    assert pr["distinct_devices"] is False and pr["n_devices"] == 1 and pr["torch_device_count"] >= 1
    assert rec["parity"]["ok"] and rec["parity"]["ranks_checked"] == 2 and rec["parity"]["assign_mismatches"] == 0


@pytest.mark.parametrize("workload", ["config2", "config5"])
def test_bench_rccl_pipelined_gather(workload):
    """The driver's launch line with one rank: bench.py under torch.distributed.run takes the RCCL branch, where the
    gather of a step overlaps the next step's kernel (two entropy buffers); the line must come out and the series
    gathered on rank 0 must equal the local one (asserted inside bench.py)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "VET_BENCH_BACKEND"):
        env.pop(k, None)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
                          "--gpus", "1", "--steps", "5", "--warmup", "2", "--workload", workload, "--no-cpu-baseline",
                          "--no-api"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    pr = rec["per_rank"]
    assert len(pr["step_ms"]) == 1 and pr["kernel_ms"][0] > 0 and pr["gather_ms"][0] > 0
    assert pr["kernel_ms"][0] <= rec["ms_per_step"] * 1.001          # the kernel is inside the step
    assert rec["roofline"]["measured_copy_ceiling"]["GBps"] > 100       # 3.8 MB at config 2: a launch-bound copy
    assert pr["backend"] == "nccl" and pr["rccl_version"] and pr["distinct_devices"] is True
    assert len(pr["device_pci_bus_id"]) == 1 and ":" in pr["device_pci_bus_id"][0]
    assert rec["parity"]["ok"] and rec["parity"]["ranks_checked"] == 1


def _engine_then_torch_worker(q):
    _paths()
    from viewport_entropy_toolkit import _native, _synthetic
    eng, plan = _make_plans()                      # the engine's library (and a HIP runtime) first ...
    mu, mv = _synthetic.random_walk_video(32, 64, base_seed=5)
    ent = plan.spatial(mu=mu, mv=mv, want_assign=False)["entropy"]
    import torch                                   # ... torch.cuda afterwards, in the same process
    torch.cuda.set_device(0)
    t = torch.arange(16, device="cuda", dtype=torch.float64)
    ok_torch = float((t * 2).sum().item()) == 240.0
    ent2 = plan.spatial(mu=mu, mv=mv, want_assign=False)["entropy"]
    import re
    runtimes = sorted(set(re.findall(r"\S*libamdhip64\S*", open("/proc/self/maps").read())))
    q.put((ok_torch, bool(np.array_equal(ent, ent2)), runtimes, _native.HIP_RUNTIME_PRELOADED))
    plan.close()


def test_engine_first_then_torch_cuda():
    """One HIP runtime per process whatever the import order: _native.load_library() preloads the runtime the
    installed torch ships, so creating the engine first and touching torch.cuda afterwards works
    (round 2: 'No HIP GPUs are available' in torch._C._cuda_init)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_engine_then_torch_worker, args=(q,))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    ok_torch, same, runtimes, preloaded = q.get(timeout=10)
    assert ok_torch and same
    assert len(runtimes) == 1, runtimes
    assert preloaded and runtimes[0] == preloaded


def _run_bench(*extra, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "VET_BENCH_BACKEND"):
        env.pop(k, None)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--no-api", "--no-variants",
                           *extra], env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("workload", ["config2", "config5", "config2x64"])
def test_bench_line_carries_the_parity_gate(workload):
    """SURVEY.md §8d 'parity gates in the same run': the line bench.py prints holds the comparison of the timed series with
    the C port of the reference path (indices bit-exact, entropy <= 1e-6) and one reference golden through the HIP path."""
    out = _run_bench("--workload", workload, "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    par = rec["parity"]
    assert par["ok"] is True and par["assign_mismatches"] == 0 and par["frames"] >= 63
    assert 0.0 <= par["entropy_max_rel"] <= 1e-6 and par["golden"].endswith(" ok")
    assert par["golden"].startswith("g5:tc200" if rec["config"]["mode"] == "transition" else "g4:w_tc50")


@pytest.mark.parametrize("fault,workload", [("assign", "config2"), ("entropy", "config2"), ("table", "config2"),
                                            ("assign", "config5"), ("entropy", "config5"), ("table", "config5")])
def test_bench_fails_on_a_parity_violation(fault, workload):
    """One flipped nearest-tile word, one entropy value 3e-6 off, or an engine plan built on the wrong lattice: bench.py prints
    no metric line and exits non-zero."""
    out = _run_bench("--workload", workload, "--no-cpu-baseline", "--inject-fault", fault)
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert "PARITY GATE FAILED" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], "a metric line was printed for a wrong result"


def test_bench_parity_reuses_the_cpu_baseline_sample():
    """With the cpu_baseline leg on, the gate compares the engine with the outputs that leg computed (its whole sample)."""
    out = _run_bench("--workload", "config2", "--cpu-seconds", "3")
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["parity"]["ok"] and rec["parity"]["frames"] > 64
    assert str(rec["parity"]["frames"]) in rec["cpu_baseline"]["sample"]
    assert rec["cpu_baseline"]["all_cores"]["cores"] > 1


def test_bench_parity_violation_at_two_ranks():
    """N > 1: every rank checks its own series and one all_reduce joins the verdicts; a violation on any rank fails the job
    (no metric line, non-zero exit from the launcher)."""
    env = dict(os.environ, VET_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload",
                          "config2", "--no-cpu-baseline", "--no-api", "--no-variants", "--inject-fault", "entropy"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "PARITY GATE FAILED" in out.stderr and '"ranks_failed": 2' in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
