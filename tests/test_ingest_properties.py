"""Property tests (hypothesis) of the dense ingest against the oracle's literal restatement of
format_trajectory_data: arbitrary unsorted, duplicated, disjoint and NaN-laden tracks."""
import numpy as np
from hypothesis import given, settings, strategies as st

from viewport_entropy_toolkit import _ingest
from oracle import vet_oracle as vo


def _clean(t, a, b):
    keep = ~(np.isnan(t) | np.isnan(a) | np.isnan(b))
    t, a, b = t[keep], a[keep], b[keep]
    return t - t.min(), a, b


track = st.lists(
    st.tuples(st.integers(0, 400), st.floats(0, 1, allow_nan=False), st.floats(0, 1, allow_nan=False),
              st.booleans()),
    min_size=1, max_size=40)


@settings(max_examples=150, deadline=None, derandomize=True)
@given(st.lists(track, min_size=1, max_size=6), st.floats(0, 1000, allow_nan=False))
def test_build_dense_matches_the_reference_semantics(tracks, t0):
    raw = []
    for rows in tracks:
        t = np.array([t0 + r[0] * 0.05 for r in rows])            # 0.05 steps: rounding merges neighbours
        a = np.array([np.nan if r[3] and i % 7 == 3 else r[1] for i, r in enumerate(rows)])
        b = np.array([r[2] for r in rows])
        if np.all(np.isnan(a)):
            a[0] = 0.5
        raw.append((t, a, b))
    got = _ingest.build_dense([_clean(*r) for r in raw])
    ref = vo.format_trajectories(raw)
    for x, y in zip(got, ref):
        assert np.array_equal(x, y, equal_nan=True)
    times, mu, mv = got
    assert len(set(times.tolist())) == len(times)                 # one frame per rounded time
    assert (~np.isnan(mu)).any(axis=1).all()                      # every frame has at least one user
    assert mu.flags["C_CONTIGUOUS"] and mu.shape == mv.shape == (len(times), len(raw))
