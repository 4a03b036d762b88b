"""End-to-end run_analysis() on a synthetic directory of user CSVs (default: BASELINE config 2 shape):
ingest (native CSV loader + dense frames), entropy on the GPU, CSV + graph outputs."""
import argparse, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "viewport-entropy-toolkit_amd"))
import numpy as np
import viewport_entropy_toolkit as vt
from viewport_entropy_toolkit import _synthetic
from viewport_entropy_toolkit.config import AnalyzerConfig

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=64)
ap.add_argument("--rows", type=int, default=3000)
ap.add_argument("--tile-counts", default="50,100,200")
args = ap.parse_args()
with tempfile.TemporaryDirectory() as tmp:
    video = Path(tmp) / "video"; video.mkdir()
    for u in range(args.users):
        t, mu, mv = _synthetic.random_walk_user(args.rows, 1234 + u)
        np.savetxt(video / f"user{u:03d}.csv", np.c_[t, mu, mv], delimiter=",", header="time,2dmu,2dmv", comments="", fmt="%.12f")
    an = vt.SpatialEntropyAnalyzer(AnalyzerConfig(tile_counts=[int(x) for x in args.tile_counts.split(",")], output_dir=Path(tmp) / "out"))
    for rep in range(3):
        t0 = time.perf_counter(); an.process_directory(video); t1 = time.perf_counter()
        an.compute_entropy(); t2 = time.perf_counter()
        an.create_visualization(f"run{rep}"); t3 = time.perf_counter()
        print(f"{args.users} users x {args.rows} rows, tile_counts={args.tile_counts}: ingest {1e3*(t1-t0):.1f} ms, "
              f"compute_entropy {1e3*(t2-t1):.1f} ms, csv+graph {1e3*(t3-t2):.1f} ms, total {1e3*(t3-t0):.1f} ms", flush=True)
