"""LDS bank-conflict model of the table gather's ds_add_u64 stream.

One wave instruction = 4 gather groups (rows of 4 random directions) x 16 lanes, each lane adding one
8-byte histogram slot; 64 banks x 4 B -> 32 slot classes (tile mod 32); cost = the deepest class.
Compares the plain row layout (lane l holds sorted entries 4l..4l+3 of a 64-entry block, so an
instruction adds every fourth entry) with the interleaved one (lane l holds entries l, l+16, l+32,
l+48: an instruction adds 16 consecutive entries).  Host-only numpy; no GPU."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "viewport-entropy-toolkit_amd"))
from viewport_entropy_toolkit import _quantiser


def simulate(tile_count=500, fov=120.0, waves=300, seed=0):
    tiles = _quantiser.lattice_xyz(tile_count)
    tiles = tiles / np.linalg.norm(tiles, axis=1, keepdims=True)
    lon, lat = _quantiser.axis_angles(100, 200)
    max_ang = np.radians(fov / 2)

    def row(px, py):
        v = _quantiser.vector_xyz(lon[px], lat[py])
        v = v / np.linalg.norm(v)
        return np.nonzero(np.arccos(np.clip(tiles @ v, -1, 1)) < max_ang)[0]

    out = {}
    for layout in ("plain", "interleaved"):
        rng = np.random.default_rng(seed)
        depth = instr = 0
        for _ in range(waves):
            rows = [row(rng.integers(0, 101), rng.integers(0, 201)) for _ in range(4)]
            for blk in range(0, max(len(r) for r in rows), 64):
                for k in range(4):
                    hit = []
                    for r in rows:
                        seg = r[blk:blk + 64]
                        full = 4 * len(seg) >= 3 * 64
                        idx = np.arange(16) + 16 * k if (layout == "interleaved" and full) else np.arange(16) * 4 + k
                        hit.append(seg[idx[idx < len(seg)]])
                    hit = np.concatenate(hit)
                    if len(hit):
                        depth += np.bincount(hit % 32, minlength=32).max()
                        instr += 1
        out[layout] = depth / instr
    return out


if __name__ == "__main__":
    for tc in (200, 500, 1000):
        print(tc, {k: round(float(v), 2) for k, v in simulate(tc).items()})
