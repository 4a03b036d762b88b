"""GPU: hypothesis-driven parity of the HIP path against the CPU oracle on small random problems
(SURVEY.md §4 iii): random user / frame counts, absent samples, lattice sets, FoV and power,
both weighted formulations, unweighted mode and transition mode."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import vet_oracle as vo
from tests._tol import W_RTOL, w_atol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from viewport_entropy_toolkit import _native
    return _native


@pytest.fixture(scope="module")
def engine(native):
    return native.Engine.default()


problem = st.fixed_dictionaries(dict(
    U=st.one_of(st.integers(1, 70), st.integers(71, 300)), T=st.integers(1, 12), seed=st.integers(0, 2 ** 31 - 1),
    p_absent=st.sampled_from([0.0, 0.1, 0.5]),
    tcs=st.lists(st.sampled_from([1, 2, 3, 20, 50, 64, 65, 100, 129, 250, 500, 1000]), min_size=1, max_size=3),
    fov=st.sampled_from([30.0, 90.0, 120.0, 150.0, 360.0]), power=st.sampled_from([0.5, 1.0, 2.0, 3.0]),
    mode=st.sampled_from(["sweep", "table", "unweighted", "transition"]),
    edge=st.booleans()))


def make_samples(pr):
    rng = np.random.default_rng(pr["seed"])
    U, T = pr["U"], pr["T"]
    if pr["edge"]:          # exact pixel boundaries, 0.0 and 1.0: the remap and clamp corners
        mu = rng.integers(0, 101, (T, U)) / 100.0
        mv = rng.integers(0, 201, (T, U)) / 200.0
    else:
        mu, mv = rng.random((T, U)), rng.random((T, U))
    gone = rng.random((T, U)) < pr["p_absent"]
    gone[:, 0] = False
    mu[gone] = np.nan
    mv[gone] = np.nan
    return mu, mv


@settings(max_examples=120, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(problem)
def test_hip_matches_oracle(native, engine, pr):
    mu, mv = make_samples(pr)
    tcs, mode = pr["tcs"], pr["mode"]
    weighted = mode in ("sweep", "table", "transition")
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], pr["fov"], pr["power"], weighted, 100, 200)
    plan.set_table_policy({"sweep": -1, "table": 1}.get(mode, 0))
    try:
        if mode == "transition":
            if pr["T"] < 2:
                return
            res = plan.transition(mu=mu, mv=mv)
            ent, pairs = vo.transition_series(mu, mv, 100, 200, tcs, closed_form=False)
            assert np.array_equal(res["pairs"], pairs)
            np.testing.assert_allclose(res["entropy"], ent, rtol=1e-9, equal_nan=True)
        else:
            res = plan.spatial(mu=mu, mv=mv, want_weights=True)
            ent, assign, weights = vo.spatial_series(mu, mv, 100, 200, tcs, fov_angle=pr["fov"],
                                                     power_factor=pr["power"], use_weight_distribution=weighted,
                                                     want_weights=True)
            assert np.array_equal(res["assign"], assign)
            if weighted:      # the reference's values under every formulation (tests/_tol.py); unweighted: integer counts
                np.testing.assert_allclose(res["weights"], weights, rtol=W_RTOL, atol=w_atol(pr["U"], pr["power"]))
            else:
                assert np.array_equal(res["weights"], weights)
            ok = np.isfinite(ent)
            assert np.array_equal(np.isnan(res["entropy"]), np.isnan(ent))
            fp = any(plan.last_formulation(k) == "ftable" for k in range(len(tcs)))
            np.testing.assert_allclose(res["entropy"][ok], ent[ok], rtol=2e-7 if fp else 1e-8, atol=1e-15)
    finally:
        plan.close()


big_problem = st.fixed_dictionaries(dict(
    U=st.integers(4097, 7000), T=st.integers(2, 4), seed=st.integers(0, 2 ** 31 - 1),
    p_absent=st.sampled_from([0.0, 0.05, 0.6]), tc=st.sampled_from([2, 20, 50, 200, 1000]),
    spread=st.sampled_from(["uniform", "few", "pixel"])))


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(big_problem)
def test_transition_beyond_4096_users_matches_the_dict_walk(native, engine, pr):
    """k_transition_big (the bucket hash in LDS, the row cut into ranges of source tiles) on random problems: users
    scattered over the sphere (many buckets: several ranges), packed into a few directions (buckets of thousands of
    users) or on exact pixel corners; against the reference's literal dict walk."""
    rng = np.random.default_rng(pr["seed"])
    U, T = pr["U"], pr["T"]
    if pr["spread"] == "uniform":
        mu, mv = rng.random((T, U)), rng.random((T, U))
    elif pr["spread"] == "few":
        mu = rng.choice(rng.random(7), (T, U))
        mv = rng.choice(rng.random(5), (T, U))
    else:
        mu = rng.integers(0, 101, (T, U)) / 100.0
        mv = rng.integers(0, 201, (T, U)) / 200.0
    gone = rng.random((T, U)) < pr["p_absent"]
    gone[:, 0] = False
    mu[gone] = np.nan
    mv[gone] = np.nan
    plan = native.Plan(engine, [vo.fibonacci_lattice(pr["tc"])], 120.0, 2.0, True, 100, 200)
    try:
        res = plan.transition(mu=mu, mv=mv, want_srccount=True)
        ent, pairs = vo.transition_series(mu, mv, 100, 200, [pr["tc"]], closed_form=False)
        assert np.array_equal(res["pairs"], pairs)
        assert np.array_equal(res["common"], (~gone[1:] & ~gone[:-1]).sum(1))
        np.testing.assert_allclose(res["entropy"], ent, rtol=1e-9, equal_nan=True)
    finally:
        plan.close()
