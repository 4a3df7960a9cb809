"""Additive component regressors (friction, spring): oracle checks on CPU, HIP parity on GPU."""
import ctypes as C

import numpy as np
import pytest

from rosdyn_amd.samples import uniform_pm1

COMPS = [dict(type=0, joint=0, min_velocity=0.05, max_velocity=0.6, parameters=[3.0, 1.5]),
         dict(type=1, joint=2, min_velocity=0.1, max_velocity=0.0, parameters=[2.0, 0.7, 0.05]),     # max <= 0 -> 1e6
         dict(type=2, joint=1, parameters=[40.0, -3.0]),
         dict(type=0, joint=5, min_velocity=0.0, max_velocity=10.0, parameters=[0.3, 0.1]),          # min < 1e-6 -> 1e-6
         dict(type=1, joint=5, min_velocity=0.2, max_velocity=0.5, parameters=[0.2, 0.1, 0.3])]


class _OC(C.Structure):
    _fields_ = [("type", C.c_int), ("joint", C.c_int), ("min_velocity", C.c_double), ("max_velocity", C.c_double),
                ("parameters", C.c_double * 3)]


def _oracle(q, dq, n, with_tau=True):
    from oracle.oracle import lib
    arr = (_OC * len(COMPS))()
    for a, c in zip(arr, COMPS):
        a.type, a.joint = c["type"], c["joint"]
        a.min_velocity, a.max_velocity = c.get("min_velocity", 0.0), c.get("max_velocity", 0.0)
        a.parameters[:] = (list(c["parameters"]) + [0, 0, 0])[:3]
    K = sum(3 if c["type"] == 1 else 2 for c in COMPS)
    N = len(q)
    Cm = np.empty((N, n, K))
    tau = np.zeros((N, n))
    f = lib().orc_components_batch
    f.restype = None
    dp = C.POINTER(C.c_double)
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, dp, dp, dp, dp]
    f(C.cast(arr, C.c_void_p), len(COMPS), n, N, q.ctypes.data_as(dp), dq.ctypes.data_as(dp), Cm.ctypes.data_as(dp),
      tau.ctypes.data_as(dp) if with_tau else None)
    return Cm, tau


def _inputs(N, n):
    q = uniform_pm1(41, (N, n))
    dq = uniform_pm1(42, (N, n))
    dq[0] = 0.0                      # omega == 0 branch (friction_polynomial2.h:45-46)
    dq[1] = 0.01                     # inside the threshold band
    dq[2] = -0.99                    # beyond max_velocity
    return q, dq


def test_oracle_component_semantics():
    q, dq = _inputs(64, 6)
    Cm, tau = _oracle(q, dq, 6)
    # friction1 on joint 0: clamp to +-0.6, sign saturates at |omega| >= 0.05
    om = np.clip(dq[:, 0], -0.6, 0.6)
    assert np.array_equal(Cm[:, 0, 1], om) and np.array_equal(Cm[:, 0, 0], np.clip(om / 0.05, -1, 1))
    # friction2 on joint 2: max_velocity <= 0 -> 1e6 (no clamp in range); omega^2 * sign
    sg = np.where(dq[:, 2] == 0, 0.0, np.where(dq[:, 2] > 0.1, 1.0, np.where(dq[:, 2] < -0.1, -1.0, dq[:, 2] / 0.1)))
    assert np.array_equal(Cm[:, 2, 2], sg) and np.array_equal(Cm[:, 2, 3], dq[:, 2]) and np.allclose(Cm[:, 2, 4], dq[:, 2]**2 * sg, rtol=1e-15)
    # spring on joint 1
    assert np.array_equal(Cm[:, 1, 5], q[:, 1]) and np.all(Cm[:, 1, 6] == 1.0)
    # min_velocity < 1e-6 -> 1e-6
    assert np.array_equal(Cm[:, 5, 7], np.clip(dq[:, 5] / 1e-6, -1, 1))
    # rows of other joints are zero; torque = regressor * parameters
    mask = np.ones_like(Cm, dtype=bool)
    k0 = 0
    for c in COMPS:
        cols = 3 if c["type"] == 1 else 2
        mask[:, c["joint"], k0:k0 + cols] = False
        k0 += cols
    assert np.all(Cm[mask] == 0.0)
    pars = np.concatenate([c["parameters"] for c in COMPS])
    assert np.allclose(tau, Cm @ pars, rtol=1e-14, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["sample", "element"])
def test_hip_components_match_oracle(layout):
    torch = pytest.importorskip("torch")
    from rosdyn_amd.components import ComponentSet
    N, n = 3001, 6
    q, dq = _inputs(N, n)
    Cr, tr = _oracle(q, dq, n)
    cs = ComponentSet(COMPS, n)
    assert cs.columns == Cr.shape[2]
    if layout == "element":
        tq, tdq = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq))
        tau = torch.zeros((n, N), dtype=torch.float64, device="cuda")
        Cg = cs.getRegressor(tq, tdq, layout="element", tau_add=tau).cpu().numpy().transpose(2, 1, 0)
        tg = tau.cpu().numpy().T
    else:
        tq, tdq = (torch.from_numpy(x).cuda() for x in (q, dq))
        tau = torch.zeros((N, n), dtype=torch.float64, device="cuda")
        Cg = cs.getRegressor(tq, tdq, layout="sample", tau_add=tau).cpu().numpy().transpose(0, 2, 1)
        tg = tau.cpu().numpy()
    assert np.array_equal(Cg, Cr)                      # pure selects / clamps / one product: bit exact
    assert np.abs(tg - tr).max() <= 1e-13 * max(1.0, np.abs(tr).max())
