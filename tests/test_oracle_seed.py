"""H.oracle_from_state (the seeding the in-motion GPU tests rest on) pinned on the CPU: an oracle seeded with the state another oracle has
reached must continue bit for bit like it -- same order, same arithmetic.  Catches the two seeding mistakes that are easy to make (a fluid
particle's `is_dynamic` flag, base_solver.py:139 / :555; the density a particle is ADDED with, from which the reference derives its mass)."""
import numpy as np
import pytest

from tests import helpers as H


@pytest.mark.parametrize("method,fixed", [("wcsph", 0), ("dfsph", 2), ("pcisph", 2)])
def test_seeded_oracle_continues_like_the_one_it_was_seeded_from(method, fixed):
    cfg = H.dam_break_scene(method=method, end=(0.24, 0.3, 0.2), velocity=(0.3, -1.0, 0.2))
    a = H.build_oracle(cfg, jitter=0.002, seed=3, fixed_iterations=fixed)
    a.prepare()
    a.step(6)
    ids = H.oracle_ids(a)
    x, v = a.field("particle_positions").copy(), a.field("particle_velocities").copy()
    b = H.oracle_from_state(cfg, x, v, ids, fixed_iterations=fixed)
    assert np.array_equal(b.field("particle_masses"), a.field("particle_masses"))
    b.prepare()
    a.step(4)
    b.step(4)
    assert a.last_pairs == b.last_pairs and a.last_pairs > 0
    assert np.array_equal(H.oracle_ids(a), H.oracle_ids(b))
    for f in ("particle_positions", "particle_velocities", "particle_densities"):
        np.testing.assert_array_equal(a.field(f), b.field(f), err_msg=f)
    a.close(); b.close()
