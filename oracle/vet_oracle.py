"""CPU oracle: a numpy restatement of the reference's viewport -> tile -> entropy path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of bench.py as the CHECKER.  The product package never
imports this module; the product path is HIP-only and fails loudly without its
extension.

Parity status: PINNED.  Every function here is checked by
tests/test_oracle_golden.py against tests/golden/*.npz, which were produced by
running the real reference (oracle/gen_golden.py) in the build container.

Each function cites the reference lines (relative to /root/reference/src/
viewport_entropy_toolkit/) whose behaviour it restates.  Quirks are reproduced,
not fixed: lattice size 2*floor(n/2)+1, the -180/-90 -> 0.0 remap, first
appearance frame order, last-duplicate-wins, nan on degenerate frames, and the
transition-entropy bucket/stale-variable behaviour.
"""

from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

GOLDEN_RATIO = (1 + np.sqrt(5)) / 2


# --------------------------------------------------------------------------- #
# value quantisers                                                            #
# --------------------------------------------------------------------------- #
def vector_from_spherical(lon, lat) -> np.ndarray:
    """data_types.py:183-216 (Vector.from_spherical): xyz rounded to 6 decimals
    with numpy-scalar ``round`` (= rint(v*1e6)/1e6)."""
    lon, lat = np.broadcast_arrays(np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64))
    theta = np.radians(lon)
    phi = np.radians(90 - lat)
    x = np.sin(phi) * np.cos(theta)
    y = np.sin(phi) * np.sin(theta)
    z = np.cos(phi)
    return np.stack([np.round(x, 6), np.round(y, 6), np.round(z, 6)], axis=-1) + 0.0


def fibonacci_lattice(tile_count: int) -> np.ndarray:
    """utilities/data_utils.py:25-56: returns 2*floor(n/2)+1 centres, [n,3]."""
    if tile_count <= 0:
        raise ValueError("Number of points must be positive")
    N = int(tile_count / 2)
    lons, lats = [], []
    for i in range(-N, N + 1):
        lat = np.arcsin(2 * i / (2 * N + 1)) * 180 / np.pi
        lon = (i % GOLDEN_RATIO) * 360 / GOLDEN_RATIO
        lon = ((lon + 180) % 360) - 180
        lons.append(lon)
        lats.append(lat)
    return vector_from_spherical(np.array(lons), np.array(lats))


def axis_tables(W: int, H: int) -> Tuple[np.ndarray, np.ndarray]:
    """px -> lon (W+1 entries) and py -> lat (H+1 entries) after
    pixel_to_spherical (data_utils.py:264-286) and the rounding + remap of
    format_trajectory_data (data_utils.py:390-397).  ``round`` there acts on
    Python floats, so it is done with Python's ``round`` here too."""
    if W <= 0 or H <= 0 or W % 2 or H % 2:
        raise ValueError("Video dimensions must be positive even numbers")
    lon_axis = np.empty(W + 1)
    for px in range(W + 1):
        lon = round(float((px / W) * 360 - 180), 1)
        if lon <= -180:
            lon = (lon + 360) % 360 - 180
        lon_axis[px] = lon
    lat_axis = np.empty(H + 1)
    for py in range(H + 1):
        lat = round(float(90 - (py / H) * 180), 1)
        if lat <= -90:
            lat = (lat + 180) % 180 - 90
        lat_axis[py] = lat
    return lon_axis, lat_axis


def direction_grid(W: int, H: int) -> np.ndarray:
    """[H+1][W+1][3] Vector grid for every pixel (py, px)."""
    lon_axis, lat_axis = axis_tables(W, H)
    return vector_from_spherical(lon_axis[None, :], lat_axis[:, None])


def normalize_to_pixel(v: np.ndarray, dim: int) -> np.ndarray:
    """data_utils.py:243-261: truncation toward zero of v*dim; v must be in [0,1]."""
    v = np.asarray(v, dtype=np.float64)
    ok = ~np.isnan(v)
    if np.any((v[ok] < 0) | (v[ok] > 1)):
        raise ValueError("Normalized coordinates must be between 0 and 1")
    out = np.full(v.shape, -1, dtype=np.int64)
    out[ok] = (v[ok] * dim).astype(np.int64)
    return out


# --------------------------------------------------------------------------- #
# per-direction geometry                                                      #
# --------------------------------------------------------------------------- #
def _unit(v: np.ndarray) -> np.ndarray:
    return v / np.sqrt((v * v).sum(axis=-1, keepdims=True))


def angular_distances(dirs: np.ndarray, tiles: np.ndarray) -> np.ndarray:
    """entropy_utils.py:41-87: arccos(clip(dot(v1/|v1|, v2/|v2|), -1, 1)),
    for every (direction, tile) -> [D, n]."""
    c = _unit(np.atleast_2d(dirs)) @ _unit(tiles).T
    return np.arccos(np.clip(c, -1.0, 1.0))


def nearest_tile(dirs: np.ndarray, tiles: np.ndarray, chunk: int = 8192) -> np.ndarray:
    """entropy_utils.py:89-106: argmin distance, first minimum on ties."""
    dirs = np.atleast_2d(dirs)
    out = np.empty(len(dirs), dtype=np.int32)
    for s in range(0, len(dirs), chunk):
        out[s:s + chunk] = np.argmin(angular_distances(dirs[s:s + chunk], tiles), axis=1)
    return out


def tile_weight_rows(dirs: np.ndarray, tiles: np.ndarray, fov_angle: float = 120.0,
                     power_factor: float = 2.0, use_weight_distribution: bool = True,
                     return_keys: bool = False):
    """entropy_utils.py:108-144: [D, n] weight rows (0 where the reference has
    no dict entry).  ``return_keys``: also the [D, n] mask of the tiles that ARE
    in the reference's dict — every tile with distance < max (:131-135), also
    when ``(..) ** power_factor`` underflows to exactly 0.0."""
    d = angular_distances(dirs, tiles)
    if use_weight_distribution:
        mx = np.radians(fov_angle / 2.0)
        keys = d < mx
        with np.errstate(invalid="ignore", under="ignore"):
            w = np.where(keys, ((mx - d) / mx) ** power_factor, 0.0)
    else:
        w = np.zeros_like(d)
        w[np.arange(len(d)), np.argmin(d, axis=1)] = 1.0
        keys = w > 0
    return (w, keys) if return_keys else w


# --------------------------------------------------------------------------- #
# entropy                                                                     #
# --------------------------------------------------------------------------- #
def _max_entropy(count) -> float:
    p = 1.0 / count
    return -count * p * np.log2(p)


def spatial_entropy_from_hist(hist: np.ndarray, touched: np.ndarray, n_tiles: int,
                              use_weight_distribution: bool) -> float:
    """entropy_utils.py:194-211 given the per-tile weight sums."""
    h = hist[touched]
    total = float(hist.sum())
    with np.errstate(all="ignore"):
        p = h / total
        ent = float(-(p * np.log2(p)).sum()) if len(h) else 0.0
        if use_weight_distribution or total > n_tiles:
            mx = _max_entropy(n_tiles)
        else:
            mx = _max_entropy(total)
        return float(np.float64(ent) / np.float64(mx))


def spatial_entropy_frame(dirs: np.ndarray, tiles: np.ndarray, fov_angle: float = 120.0,
                          power_factor: float = 2.0, use_weight_distribution: bool = True):
    """entropy_utils.py:147-211 for one frame: ``dirs`` are the present users'
    Vectors in column order.  Returns (entropy, hist[n], nearest[U])."""
    if len(dirs) == 0:
        raise ValueError("Empty vector dictionary")
    rows, keys = tile_weight_rows(dirs, tiles, fov_angle, power_factor, use_weight_distribution, return_keys=True)
    hist = np.zeros(len(tiles))
    for r in rows:                      # user order, as the reference accumulates
        hist += r
    # the tiles in the reference's weight_per_tile dict: in some user's FoV, whatever the weight — a tile whose
    # summed weight is 0.0 (or underflows against the total) makes the frame NaN (0 * log2 0, :195-198)
    touched = keys.any(axis=0)
    near = nearest_tile(dirs, tiles)
    return spatial_entropy_from_hist(hist, touched, len(tiles), use_weight_distribution), hist, near


def transition_entropy_pairs(prev_tile: Sequence[int], cur_tile: Sequence[int], n_tiles: int) -> float:
    """entropy_utils.py:213-332 as a literal dict walk over the common users'
    (source tile, destination tile) pairs in column order.  Reproduces:
    * :280-287 — membership is tested in ``weight_per_tile`` so the first user of
      a source tile lands in an int-keyed bucket that later users never join;
    * :307-315 — the entropy loop re-uses the ``transition_weight`` left behind
      by the summing loop (the last bucket's weight) for every bucket."""
    weight_per_tile: Dict[int, int] = {}
    trans: Dict[int, Dict[object, int]] = {}
    total = 0
    for p, c in zip(prev_tile, cur_tile):
        p, c = int(p), int(c)
        if p not in weight_per_tile:
            trans[p] = {("first", c): 1}
        else:
            key = ("tile", c)
            trans[p][key] = trans[p].get(key, 0) + 1
        weight_per_tile[p] = weight_per_tile.get(p, 0) + 1
        total += 1
    if total == 0:
        raise ZeroDivisionError("float division by zero")
    ent = 0
    with np.errstate(all="ignore"):
        for p, tw in weight_per_tile.items():
            prop = float(tw) / float(total)
            tsum = 0
            last = None
            for k in trans[p]:
                last = trans[p][k]
                tsum += last
            cell = 0
            for _ in trans[p]:
                q = float(last) / float(tsum)
                cell += q * np.log2(q)
            ent += -prop * cell
        mx = _max_entropy(n_tiles) if total > n_tiles else total * -(1 / total) * np.log2(1 / total)
        return float(np.float64(ent) / np.float64(mx))


def transition_entropy_closed_form(prev_tile: np.ndarray, cur_tile: np.ndarray, n_tiles: int) -> float:
    """The same quantity in the form the HIP kernel evaluates:
    per source tile p with m users (column order a_1<...<a_m): K = 1 + number of
    distinct destinations among a_2..a_m; w = 1 if m == 1 else the count (among
    a_2..a_m) of the destination whose first appearance is latest;
    cell = -(m/N) * K * (w/m) * log2(w/m)."""
    prev_tile = np.asarray(prev_tile)
    cur_tile = np.asarray(cur_tile)
    N = len(prev_tile)
    if N == 0:
        raise ZeroDivisionError("float division by zero")
    ent = 0.0
    with np.errstate(all="ignore"):
        seen = []
        for p in prev_tile:
            if p not in seen:
                seen.append(p)
        for p in seen:
            dst = cur_tile[prev_tile == p]
            m = len(dst)
            rest = dst[1:]
            if m == 1:
                K, w = 1, 1
            else:
                first_pos = {}
                for i, c in enumerate(rest):
                    first_pos.setdefault(int(c), i)
                last_c = max(first_pos, key=lambda c: first_pos[c])
                K = 1 + len(first_pos)
                w = int((rest == last_c).sum())
            q = w / m
            ent += -(m / N) * (K * (q * np.log2(q)))
        mx = _max_entropy(n_tiles) if N > n_tiles else N * -(1 / N) * np.log2(1 / N)
        return float(np.float64(ent) / np.float64(mx))


# --------------------------------------------------------------------------- #
# whole-series drivers (what the analyzers compute)                           #
# --------------------------------------------------------------------------- #
def sample_directions(mu: np.ndarray, mv: np.ndarray, W: int, H: int):
    """mu/mv [T][U] (NaN = absent) -> (px, py, present, grid)."""
    present = ~(np.isnan(mu) | np.isnan(mv))
    px = normalize_to_pixel(mu, W)
    py = normalize_to_pixel(mv, H)
    return px, py, present, direction_grid(W, H)


def spatial_series(mu: np.ndarray, mv: np.ndarray, W: int, H: int, tile_counts: Sequence[int],
                   fov_angle: float = 120.0, power_factor: float = 2.0,
                   use_weight_distribution: bool = True, want_weights: bool = False):
    """analyzers/spatial_entropy.py:107-164 on dense frame-major arrays.
    Returns (entropy[T], assign[T][U] (-1 absent), weights[T][n0] or None)."""
    px, py, present, grid = sample_directions(mu, mv, W, H)
    T, U = mu.shape
    lattices = [fibonacci_lattice(tc) for tc in tile_counts]
    flat = grid.reshape(-1, 3)
    D = len(flat)
    near_tabs = [nearest_tile(flat, L) for L in lattices]
    did = np.where(present, py * (W + 1) + px, 0)
    assign = np.where(present, near_tabs[0][did], -1).astype(np.int32)
    ent = np.zeros(T)
    weights = np.zeros((T, len(lattices[0]))) if want_weights else None
    # weight rows per *distinct* direction actually used (the reference recomputes
    # them per sample; the values are identical)
    used = np.unique(did[present])
    remap = np.full(D, -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    for k, L in enumerate(lattices):
        rows, keys = tile_weight_rows(flat[used], L, fov_angle, power_factor, use_weight_distribution, return_keys=True)
        for t in range(T):
            ids = remap[did[t][present[t]]]
            if len(ids) == 0:
                raise ValueError("Empty vector dictionary")
            r = rows[ids]
            hist = np.zeros(len(L))
            for row in r:
                hist += row
            touched = keys[ids].any(axis=0)
            ent[t] += spatial_entropy_from_hist(hist, touched, len(L), use_weight_distribution)
            if want_weights and k == 0:
                # dense convention of the C-ABI (include/vet.h): a tile that is a key of the reference's dict with
                # the value 0.0 carries -0.0, a tile that is no key +0.0
                weights[t] = np.where(touched & (hist == 0), -0.0, hist)
    return ent / len(lattices), assign, weights


def transition_series(mu: np.ndarray, mv: np.ndarray, W: int, H: int, tile_counts: Sequence[int],
                      closed_form: bool = True):
    """analyzers/transition_entropy.py:107-175.  Returns (entropy[T-1],
    pairs[T-1][U][2] (-1 where the user is not in both frames))."""
    px, py, present, grid = sample_directions(mu, mv, W, H)
    T, U = mu.shape
    flat = grid.reshape(-1, 3)
    did = np.where(present, py * (W + 1) + px, 0)
    ent = np.zeros(T - 1)
    pairs = np.full((T - 1, U, 2), -1, dtype=np.int32)
    fn = transition_entropy_closed_form if closed_form else transition_entropy_pairs
    for k, tc in enumerate(tile_counts):
        L = fibonacci_lattice(tc)
        near = nearest_tile(flat, L)[did]
        for t in range(1, T):
            both = present[t] & present[t - 1]
            p, c = near[t - 1][both], near[t][both]
            ent[t - 1] += fn(p, c, len(L))
            if k == 0:
                pairs[t - 1, both, 0] = p
                pairs[t - 1, both, 1] = c
    return ent / len(tile_counts), pairs


# --------------------------------------------------------------------------- #
# naive lat/lon-grid tiling                            #
# --------------------------------------------------------------------------- #
def naive_tile_indices(lon: np.ndarray, lat: np.ndarray, tile_height, tile_width):
    """entropy_utils.py:378-381: int((lon+180)/w), int((lat+90)/h) (truncation)."""
    return ((np.asarray(lon) + 180) / tile_width).astype(np.int64), ((np.asarray(lat) + 90) / tile_height).astype(np.int64)


def naive_entropy_from_counts(counts: np.ndarray, tile_height, tile_width, use_weight_distribution: bool) -> float:
    """entropy_utils.py:425-452 given the number of users per occupied tile."""
    counts = np.asarray(counts, dtype=np.float64)
    counts = counts[counts > 0]
    total = float(counts.sum())
    num_tiles = int(180.0 / tile_height) * int(360.0 / tile_width)
    with np.errstate(all="ignore"):
        p = counts / total
        ent = float(-(p * np.log2(p)).sum()) if len(p) else 0.0
        mx = _max_entropy(num_tiles) if (use_weight_distribution or total > num_tiles) else _max_entropy(total)
        return float(np.float64(ent) / np.float64(mx))


def naive_series(mu: np.ndarray, mv: np.ndarray, W: int, H: int, tile_height, tile_width,
                 use_weight_distribution: bool = True):
    """analyzers/naive_spatial_entropy.py:102-152 on dense arrays.  Returns (entropy[T],
    lon_idx[T][U], lat_idx[T][U]) with -1 where absent."""
    px, py, present, _ = sample_directions(mu, mv, W, H)
    lon_axis, lat_axis = axis_tables(W, H)
    li_axis, lj_axis = naive_tile_indices(lon_axis, lat_axis, tile_height, tile_width)
    T, U = mu.shape
    li = np.where(present, li_axis[np.where(present, px, 0)], -1)
    lj = np.where(present, lj_axis[np.where(present, py, 0)], -1)
    ent = np.zeros(T)
    for t in range(T):
        if not present[t].any():
            raise ValueError("Empty radial points dictionary")
        keys = li[t][present[t]] * 100000 + lj[t][present[t]]
        _, counts = np.unique(keys, return_counts=True)
        ent[t] = naive_entropy_from_counts(counts, tile_height, tile_width, use_weight_distribution)
    return ent, li, lj


# --------------------------------------------------------------------------- #
# ingest (frame index construction)                                           #
# --------------------------------------------------------------------------- #
def format_trajectories(tracks: List[Tuple[np.ndarray, np.ndarray, np.ndarray]]):
    """data_utils.py:289-410 on raw per-user (time, mu, mv) arrays, in the given
    user order: drop NaN rows, time -= min, round(time, 1), frame index in
    first-appearance order across users, last duplicate wins.
    Returns (frame_times[T], mu[T][U], mv[T][U]) with NaN = absent."""
    frame_of: Dict[float, int] = {}
    times: List[float] = []
    cells = []
    for u, (t, a, b) in enumerate(tracks):
        t, a, b = (np.asarray(x, dtype=np.float64) for x in (t, a, b))
        keep = ~(np.isnan(t) | np.isnan(a) | np.isnan(b))
        t, a, b = t[keep], a[keep], b[keep]
        if len(t) == 0:
            raise ValueError("No valid data")
        t = np.round(t - t.min(), 1)
        for ti, ai, bi in zip(t, a, b):
            ti = float(ti)
            if ti not in frame_of:
                frame_of[ti] = len(times)
                times.append(ti)
            cells.append((frame_of[ti], u, ai, bi))
    T, U = len(times), len(tracks)
    mu = np.full((T, U), np.nan)
    mv = np.full((T, U), np.nan)
    for f, u, ai, bi in cells:
        mu[f, u], mv[f, u] = ai, bi
    return np.array(times), mu, mv


# --------------------------------------------------------------------------- #
# tile boundary / area geometry of the Fibonacci tiling      #
# --------------------------------------------------------------------------- #
def _norm3(v: np.ndarray) -> np.ndarray:
    return np.sqrt((v * v).sum(axis=-1))


def _gc_point_near(n_a: np.ndarray, n_b: np.ndarray, centre: np.ndarray):
    """data_utils.py:445-468 + 483-503: the two great circles with normals n_a, n_b meet in +-p,
    p = normalise(cross(normalise(n_a), normalise(n_b))); the one whose chord to ``centre`` is shorter after
    rounding to 4 decimals wins, ties go to -p.  Returns (point, rounded chord length)."""
    p = np.cross(n_a / _norm3(n_a), n_b / _norm3(n_b))
    p = p / _norm3(p)
    l1 = np.round(_norm3(centre - p), 4)
    l2 = np.round(_norm3(centre + p), 4)
    return (p, l1) if l1 < l2 else (-p, l2)


def fb_tile_boundaries(tile_count: int) -> List[List[Tuple[np.ndarray, np.ndarray]]]:
    """data_utils.py:58-189 (get_fb_tile_boundaries): per tile the list of boundary edges (two points each)
    in the order the reference appends them."""
    if tile_count <= 0:
        raise ValueError("Tile counts cannot be less than 1 for to visualize tiling!")
    C = fibonacci_lattice(tile_count)
    n = len(C)
    result = []
    for i in range(n):
        seg = C[i] - C                                   # get_line_segment(c_i, c_j) for every j
        length = _norm3(seg)
        others = [j for j in range(n) if j != i]
        order = sorted(others, key=lambda j: length[j])  # stable: ties keep index order
        edges = []
        if order:
            limit = length[order[0]] * 1.7
            near = []
            for j in order:
                if length[j] >= limit:
                    break
                near.append(j)
            for a, j in enumerate(near):
                hits = []
                for b, k in enumerate(near):
                    if b == a:
                        continue
                    pt, chord = _gc_point_near(seg[j], seg[k], C[i])
                    hits.append((pt, chord, k))
                if len(hits) < 2:
                    continue
                hits.sort(key=lambda h: h[1])
                first, second = hits[0], hits[1]
                _, corner_chord = _gc_point_near(seg[first[2]], seg[second[2]], C[i])
                mid = (C[i] + C[j]) / 2
                mid = mid / _norm3(mid)
                if corner_chord > np.round(_norm3(C[i] - mid), 4):
                    edges.append((first[0], second[0]))
        result.append(edges)
    return result


def tile_corners(edges: Sequence[Tuple[np.ndarray, np.ndarray]]) -> np.ndarray:
    """data_utils.py:530-575 (get_tile_corners): corner walk over the edges rounded to 4 decimals."""
    key = lambda p: tuple((np.round(np.asarray(p, dtype=np.float64), 4) + 0.0).tolist())   # noqa: E731
    start = [key(edges[0][0]), key(edges[0][1])]
    walk = list(start)
    seen = {start[0]: True, start[1]: True}
    adj: Dict[tuple, list] = {}
    for p1, p2 in edges:
        k1, k2 = key(p1), key(p2)
        adj.setdefault(k1, []).append(k2)
        adj.setdefault(k2, []).append(k1)
        seen.setdefault(k1, False)
        seen.setdefault(k2, False)
    cur = walk[1]
    while not seen[adj[cur][0]] or not seen[adj[cur][1]]:
        cur = adj[cur][0] if not seen[adj[cur][0]] else adj[cur][1]
        walk.append(cur)
        seen[cur] = True
    return np.array(walk, dtype=np.float64)


def spherical_triangle_area(p1, p2, p3) -> float:
    """data_utils.py:600-655: spherical excess of the triangle of the three normalised points."""
    v = [np.asarray(p, dtype=np.float64) / _norm3(np.asarray(p, dtype=np.float64)) for p in (p1, p2, p3)]

    def angle(a, b, c):
        t1 = b - np.dot(b, a) * a
        t2 = c - np.dot(c, a) * a
        t1 = t1 / _norm3(t1)
        t2 = t2 / _norm3(t2)
        return np.arccos(np.clip(np.dot(t1, t2), -1.0, 1.0))

    return float(angle(v[0], v[1], v[2]) + angle(v[1], v[2], v[0]) + angle(v[2], v[0], v[1]) - np.pi)


def polygon_area(edges) -> float:
    """data_utils.py:657-678: fan triangulation from the first corner of the corner walk."""
    c = tile_corners(edges)
    if len(c) < 3:
        raise ValueError("At least 3 boundary points are needed for a polygon.")
    total = 0.0
    for i in range(1, len(c) - 1):
        total += spherical_triangle_area(c[0], c[i], c[i + 1])
    return total


def fb_tile_areas(tile_count: int):
    """data_utils.py:680-710: (areas[n], fractions of the sphere[n])."""
    areas = np.array([polygon_area(e) for e in fb_tile_boundaries(tile_count)])
    return areas, areas / (4 * np.pi)
