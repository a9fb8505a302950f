"""Synthetic query batches and the named limit sets of SURVEY.md §8(d).

The distribution is the reference's tests/randomConfiguration.m:14-34 generalised to per-joint limits;
the generator is counter-based (splitmix64 of (seed, query, joint, field)) so the HIP kernel
``k_generate`` (csrc/ltp_aux_kernels.hip) and this numpy mirror produce bit-identical batches and any
shard can be generated independently on any rank.
"""
import numpy as np

PANDA = dict(
    q_min=[-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973],
    q_max=[2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973],
    v_max=[2.175, 2.175, 2.175, 2.175, 2.61, 2.61, 2.61],
    a_max=[15.0, 7.5, 10.0, 12.5, 15.0, 20.0, 20.0],
    j_max=[7500.0, 3750.0, 5000.0, 6250.0, 7500.0, 10000.0, 10000.0],
)


def limit_set(name, dof=None):
    """'panda' (7-DoF arm), 'ref' (reference README limits v1/a2/j15, q +-3.14), 'ref30'."""
    if name == "panda":
        return 7, {k: list(v) for k, v in PANDA.items()}
    if name in ("ref", "ref30"):
        d = dof or (30 if name == "ref30" else 7)
        return d, dict(q_min=[-3.14] * d, q_max=[3.14] * d, v_max=[1.0] * d, a_max=[2.0] * d, j_max=[15.0] * d)
    raise ValueError(name)


_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def unit_random(seed, query, joint, field):
    """u in [0,1): same integer arithmetic as unit_random() in csrc/ltp_aux_kernels.hip."""
    with np.errstate(over="ignore"):
        query = np.asarray(query, dtype=np.uint64)
        joint = np.asarray(joint, dtype=np.uint64)
        ctr = (query * np.uint64(64) + joint) * np.uint64(4) + np.uint64(field) + np.uint64(1)
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * ctr
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def generate_queries(n, limits, seed=12345, first_query=0):
    """Row-major [n][dof] float64 arrays (q_goal, q_0, v_0, a_0); every query passes checkInputs."""
    q_min, q_max, v_max, a_max, j_max = (np.asarray(limits[k], dtype=np.float64) for k in ("q_min", "q_max", "v_max", "a_max", "j_max"))
    dof = q_min.size
    q = (np.arange(n, dtype=np.uint64) + np.uint64(first_query))[:, None]
    j = np.arange(dof, dtype=np.uint64)[None, :]
    u0, u1, u2, u3 = (unit_random(seed, q, j, f) for f in range(4))
    eps = 1e-6
    q0 = q_min + u0 * (q_max - q_min)
    qg = q_min + u1 * (q_max - q_min)
    vm = v_max - eps
    v0 = -vm + u2 * (2.0 * vm)
    pos = v0 >= 0.0
    with np.errstate(invalid="ignore"):
        a_lb = np.where(pos, -(a_max - eps), np.maximum(-(a_max - eps), -np.sqrt(2.0 * j_max * (v_max - np.abs(v0)))))
        a_ub = np.where(pos, np.minimum(a_max - eps, np.sqrt(2.0 * j_max * (v_max - v0))), a_max + 0.0 * v0)
    a0 = a_lb + u3 * (a_ub - a_lb)
    return qg, q0, v0, a0
