"""Seeded synthetic viewport-centre trajectories (SURVEY.md §8d input generator).

Stand-alone (numpy only) so that the golden-vector generator under ``oracle/``
can load this file by path without importing the rest of the package.

Per user ``u`` of video ``video_id`` the generator is seeded with
``base_seed + video_id * 10**6 + u`` and draws a smooth random walk:

* ``time[t] = t * 0.1``
* ``mu = (0.5 + cumsum(N(0, 0.01))) mod 1``        (longitude axis, wraps)
* ``mv = clip(0.5 + cumsum(N(0, 0.005)), 0, 1)``   (latitude axis, clamps)

``uniform_sphere`` draws the histogram worst case instead (no locality):
``mu ~ U[0,1)``, ``mv = arccos(1 - 2 U) / pi``.
"""

from __future__ import annotations

import numpy as np

__all__ = ["random_walk_user", "random_walk_video", "uniform_sphere_video"]


def random_walk_user(n_frames: int, seed: int):
    """One user's ``(time, mu, mv)`` float64 arrays of length ``n_frames``."""
    rng = np.random.default_rng(seed)
    time = np.arange(n_frames, dtype=np.float64) * 0.1
    mu = np.mod(0.5 + np.cumsum(rng.normal(0.0, 0.01, n_frames)), 1.0)
    mv = np.clip(0.5 + np.cumsum(rng.normal(0.0, 0.005, n_frames)), 0.0, 1.0)
    return time, mu, mv


def random_walk_video(n_users: int, n_frames: int, base_seed: int = 1234,
                      video_id: int = 0, p_absent: float = 0.0):
    """Frame-major dense arrays ``mu[T][U]``, ``mv[T][U]`` (NaN = absent)."""
    mu = np.empty((n_frames, n_users), dtype=np.float64)
    mv = np.empty((n_frames, n_users), dtype=np.float64)
    for u in range(n_users):
        _, a, b = random_walk_user(n_frames, base_seed + video_id * 10**6 + u)
        mu[:, u] = a
        mv[:, u] = b
    if p_absent > 0.0:
        rng = np.random.default_rng(base_seed + video_id * 10**6 + 999_983)
        gone = rng.random((n_frames, n_users)) < p_absent
        # keep at least one user per frame, as the reference's frame index does
        gone[np.arange(n_frames), rng.integers(0, n_users, n_frames)] = False
        mu[gone] = np.nan
        mv[gone] = np.nan
    return mu, mv


def uniform_sphere_video(n_users: int, n_frames: int, base_seed: int = 1234,
                         video_id: int = 0):
    """Uniform-on-sphere samples: no temporal or spatial locality."""
    rng = np.random.default_rng(base_seed + video_id * 10**6 + 7)
    mu = rng.random((n_frames, n_users))
    mv = np.arccos(1.0 - 2.0 * rng.random((n_frames, n_users))) / np.pi
    return mu, np.clip(mv, 0.0, 1.0)
