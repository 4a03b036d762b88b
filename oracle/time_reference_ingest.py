#!/usr/bin/env python3
"""Times the reference's ingest (``process_directory``: data_utils.py:289-410) beside this
repo's vectorised ingest on the same CSV directory and checks that both produce the same frames.

TEST INFRASTRUCTURE ONLY, build container only (the reference is mounted at /root/reference and
never travels).  Both packages import as ``viewport_entropy_toolkit``, so each runs in its own
child process; a child dumps ``time[T]`` and the per-cell unit vectors ``xyz[T,U,3]`` (NaN = None
cell) and the parent compares them bit for bit.  SURVEY.md §8f row 1.

    python oracle/time_reference_ingest.py [--users 64] [--rows 3000] [--absent 0.05]
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent

CHILD = r"""
import json, sys, time, types
from pathlib import Path
import numpy as np
which, src, directory, out = sys.argv[1:5]
if which == "reference":
    sys.modules.setdefault("pyvista", types.ModuleType("pyvista"))   # off-path plotting module, not installed
sys.path.insert(0, src)
import viewport_entropy_toolkit as vt
from viewport_entropy_toolkit.config import AnalyzerConfig
an = vt.SpatialEntropyAnalyzer(AnalyzerConfig(tile_counts=[50], output_dir=Path(out) / ("out_" + which)))
t0 = time.perf_counter()
an.process_directory(Path(directory))
dt = time.perf_counter() - t0
if which == "reference":
    df = an._data_cache["vectors"]
    names = [c for c in df.columns if c != "time"]
    xyz = np.full((len(df), len(names), 3), np.nan)
    for j, name in enumerate(names):
        for i, v in enumerate(df[name]):
            if v is not None:
                xyz[i, j] = (v.x, v.y, v.z)
    times = df["time"].to_numpy(dtype=np.float64)
else:
    from viewport_entropy_toolkit import _quantiser
    times, mu, mv, names = an._dense
    lon, lat = _quantiser.axis_angles(an.config.video_width, an.config.video_height)
    ok = ~np.isnan(mu)
    px = np.where(ok, mu * an.config.video_width, 0).astype(np.int64)
    py = np.where(ok, mv * an.config.video_height, 0).astype(np.int64)
    xyz = _quantiser.vector_xyz(lon[px], lat[py])
    xyz[~ok] = np.nan
np.savez(Path(out) / (which + ".npz"), time=times, xyz=xyz, names=np.array(names))
print(json.dumps({"which": which, "seconds": dt}))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=64)
    ap.add_argument("--rows", type=int, default=3000)
    ap.add_argument("--absent", type=float, default=0.05, help="share of rows dropped per user (ragged tracks)")
    args = ap.parse_args()
    sys.path.insert(0, str(REPO / "viewport-entropy-toolkit_amd"))
    from viewport_entropy_toolkit import _synthetic

    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        video = tmp / "video"
        video.mkdir()
        rng = np.random.default_rng(7)
        for u in range(args.users):
            t, mu, mv = _synthetic.random_walk_user(args.rows, 1234 + u)
            keep = rng.random(args.rows) >= args.absent
            keep[0] = True
            np.savetxt(video / f"user{u:03d}.csv", np.c_[t, mu, mv][keep], delimiter=",",
                       header="time,2dmu,2dmv", comments="", fmt="%.12f")
        env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
        res = {}
        for which, src in (("ours", REPO / "viewport-entropy-toolkit_amd"), ("reference", "/root/reference/src")):
            p = subprocess.run([sys.executable, "-c", CHILD, which, str(src), str(video), str(tmp)], env=env,
                               capture_output=True, text=True, cwd=tmp)
            if p.returncode:
                sys.exit(f"{which} failed:\n{p.stderr}")
            res[which] = json.loads(p.stdout.strip().splitlines()[-1])["seconds"]
        a, b = np.load(tmp / "ours.npz"), np.load(tmp / "reference.npz")
        same = (list(a["names"]) == list(b["names"]) and np.array_equal(a["time"], b["time"])
                and np.array_equal(a["xyz"], b["xyz"], equal_nan=True))
        print(json.dumps({"users": args.users, "rows": args.rows, "frames": int(len(a["time"])),
                          "reference_s": res["reference"], "ours_s": res["ours"],
                          "speedup": res["reference"] / res["ours"], "identical_frames": bool(same)}))
        if not same:
            sys.exit("ingest mismatch")


if __name__ == "__main__":
    main()
