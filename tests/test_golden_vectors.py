"""Committed regression vectors (tests/golden/oracle_vectors.npz, made by tests/golden/make_oracle_vectors.py).

CPU: the oracle still reproduces them bit for bit (guards the checker itself against drift between rounds).
GPU: the HIP path matches them to 1e-9 through the C ABI — the comparison needs no oracle code at run time.
"""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = os.path.join(HERE, "golden", "oracle_vectors.npz")


def _maker():
    spec = importlib.util.spec_from_file_location("make_oracle_vectors", os.path.join(HERE, "golden", "make_oracle_vectors.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_reproduces_committed_vectors(oracle_mod):
    want = np.load(VEC)
    got = _maker().build()
    assert sorted(want.files) == sorted(got)
    for k in want.files:
        assert np.array_equal(want[k], got[k], equal_nan=True), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["panda", "ref", "ref30"])
def test_device_matches_committed_vectors(name):
    import longtermplanner_amd as amd
    g = np.load(VEC)
    D, lim = amd.limit_set(name)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    ins = [g[f"{name}/{k}"] for k in ("q_goal", "q_0", "v_0", "a_0")]
    r = ltp.planBatchHost(*ins, sample=True)
    assert np.array_equal(r["traj_len"], g[f"{name}/traj_len"])
    assert np.array_equal(r["status"] == 0, g[f"{name}/status"] == 1)
    assert np.array_equal(r["slowest"], g[f"{name}/slowest"]) and np.array_equal(r["mod"], g[f"{name}/mod"])
    for k in ("t_opt", "t_scaled", "v_drive", "t_required", "dir"):
        assert np.nanmax(np.abs(r[k] - g[f"{name}/{k}"])) <= 1e-9, k
    maker = _maker()
    for p in range(ins[0].shape[0]):
        L = int(r["traj_len"][p])
        if L <= 0:
            continue
        idx = maker.sample_indices(L)
        rows = amd.unpack_trajectory(r["packed"], int(r["offsets"][p]), D, L)
        for k, arr in enumerate(rows):
            assert np.max(np.abs(arr[:, idx] - g[f"{name}/samples"][k, p])) <= 1e-9, (p, k)
