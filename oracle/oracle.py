"""ctypes driver of the C oracle (oracle/rosdyn_oracle.c) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED at the reference level (see rosdyn_oracle.c header): the reference is unbuildable in
this image and its tests carry no golden vectors.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import urdf_model

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


class _UJoint(C.Structure):
    _fields_ = [("urdf_type", C.c_int), ("xyz", C.c_double * 3), ("quat", C.c_double * 4), ("axis", C.c_double * 3)]


class _ULink(C.Structure):
    _fields_ = [("has_inertial", C.c_int), ("mass", C.c_double), ("xyz", C.c_double * 3), ("quat", C.c_double * 4),
                ("ixx", C.c_double), ("ixy", C.c_double), ("ixz", C.c_double),
                ("iyy", C.c_double), ("iyz", C.c_double), ("izz", C.c_double)]


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("rosdyn_oracle.c", "components_oracle.c", "ik_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        l.orc_chain_create.restype = C.c_void_p
        l.orc_chain_create.argtypes = [C.c_int, C.POINTER(_UJoint), C.POINTER(_ULink), dp, C.c_int, C.POINTER(C.c_int)]
        l.orc_chain_destroy.argtypes = [C.c_void_p]
        for name, nargs in (("orc_fk", 2), ("orc_jacobian", 2), ("orc_jacobian_link", 3),("orc_twist", 3), ("orc_dtwist", 6), ("orc_ddtwist", 5), ("orc_ddtwist_parts", 6),
                            ("orc_joint_torque", 6), ("orc_regressor", 4), ("orc_joint_inertia", 2),
                            ("orc_nominal_parameters", 1)):
            f = getattr(l, name)
            f.restype = None
            f.argtypes = [C.c_void_p] + [dp] * nargs
        l.orc_batch_torque_regressor.restype = C.c_int
        l.orc_batch_torque_regressor.argtypes = [C.c_void_p, C.c_long, dp, dp, dp, dp, dp, C.c_int]
        l.orc_has_openmp.restype = C.c_int
        l.orc_frame_distance.restype = None
        l.orc_frame_distance.argtypes = [dp, dp, dp]
        l.orc_frame_distance_quat.restype = None
        l.orc_frame_distance_quat.argtypes = [dp, dp, dp, dp, dp]
        l.orc_solve_quadprog.restype = C.c_int
        l.orc_solve_quadprog.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp]
        l.orc_local_ik.restype = C.c_int
        l.orc_local_ik.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp, dp, dp, C.c_double, C.c_double, C.c_int, dp,
                                   C.POINTER(C.c_int)]
        _lib = l
    return _lib


def frame_distance(T_wa, T_wb):
    """getFrameDistance (frame_distance.h:44-49) for one pair of 3x4 [R|p] frames."""
    a, b, d = _c(T_wa).reshape(12), _c(T_wb).reshape(12), np.empty(6)
    lib().orc_frame_distance(_p(a), _p(b), _p(d))
    return d


def frame_distance_quat(T_wa, T_wb, jac=False):
    """getFrameDistanceQuat (jac=False) / getFrameDistanceQuatJac (jac=True: returns (distance, 6x6 jacobian))."""
    a, b, d = _c(T_wa).reshape(12), _c(T_wb).reshape(12), np.empty(6)
    J = np.empty((6, 6)) if jac else None
    lib().orc_frame_distance_quat(_p(a), _p(b), _p(np.array([1.0 if jac else 0.0])), _p(d), _p(J))
    return (d, J) if jac else d


class _OracleComponent(C.Structure):
    _fields_ = [("type", C.c_int), ("joint", C.c_int), ("min_velocity", C.c_double), ("max_velocity", C.c_double),
                ("parameters", C.c_double * 3)]


def components_regressor(specs, n, q, dq):
    """Per-joint additive components (components_oracle.c: friction_polynomial1.h:45-52, friction_polynomial2.h:42-58,
    ideal_spring.h:64-70).  specs = [(type, joint, min_velocity, max_velocity, (p0, p1, p2)), ...]; q, dq (N, n).
    Returns (C (N, n, K), tau (N, n))."""
    arr = (_OracleComponent * len(specs))()
    for a, (ty, joint, vmin, vmax, par) in zip(arr, specs):
        a.type, a.joint, a.min_velocity, a.max_velocity = ty, joint, vmin, vmax
        a.parameters[:] = (list(par) + [0.0, 0.0, 0.0])[:3]
    K = sum(3 if sp[0] == 1 else 2 for sp in specs)
    q, dq = _c(q), _c(dq)
    N = len(q)
    Cm, tau = np.empty((N, n, K)), np.zeros((N, n))
    f = lib().orc_components_batch
    f.restype = None
    dp = C.POINTER(C.c_double)
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, dp, dp, dp, dp]
    f(C.cast(arr, C.c_void_p), len(specs), n, N, _p(q), _p(dq), _p(Cm), _p(tau))
    return Cm, tau


def solve_quadprog(G, g0, CI, ci0):
    """min 1/2 x'Gx + g0'x  s.t.  CI'x + ci0 >= 0  (Goldfarb-Idnani); returns (status, x)."""
    G, g0, CI, ci0 = _c(G), _c(g0), _c(CI), _c(ci0)
    x = np.empty(len(g0))
    st = lib().orc_solve_quadprog(len(g0), len(ci0), _p(G), _p(g0), _p(CI), _p(ci0), _p(x))
    return st, x


def _p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleChain(object):
    """Per-sample evaluation of the restated reference, looped over the batch on the host.

    Shapes mirror the reference's Eigen return values: Y[s] is the (n x P) matrix of getRegressor,
    J[s] the (6 x n) Jacobian, M[s] (n x n), T[s, l] the 3x4 [R|p] of T_bl[l], twists[s, l] = [lin; ang].
    """

    def __init__(self, urdf, base, tool, gravity=(0.0, 0.0, 0.0), input_joint_names=None):
        self.spec = spec = urdf_model.load(urdf, base, tool, gravity, input_joint_names)
        nj = spec.n_joints
        uj = (_UJoint * max(nj, 1))()
        for i, j in enumerate(spec.joints):
            uj[i].urdf_type = j.urdf_type
            uj[i].xyz[:] = j.xyz
            uj[i].quat[:] = j.quat
            uj[i].axis[:] = j.axis
        ul = (_ULink * (nj + 1))()
        for i, l in enumerate(spec.links):
            ul[i].has_inertial = int(l.has_inertial)
            ul[i].mass = l.mass
            ul[i].xyz[:] = l.xyz
            ul[i].quat[:] = l.quat
            ul[i].ixx, ul[i].ixy, ul[i].ixz, ul[i].iyy, ul[i].iyz, ul[i].izz = l.inertia
        g = (C.c_double * 3)(*spec.gravity)
        idx = (C.c_int * max(spec.n_active, 1))(*spec.input_chain_index)
        self._h = lib().orc_chain_create(nj, uj, ul, g, spec.n_active, idx)
        if not self._h:
            raise RuntimeError("oracle: chain too long")
        self.n, self.nJ, self.L, self.P = spec.n_active, nj, nj + 1, 10 * nj

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_chain_destroy(self._h)
            self._h = None

    def _in(self, *arrs):
        out = [np.atleast_2d(_c(a)) for a in arrs]
        for a in out:
            assert a.shape == out[0].shape and a.shape[1] == self.n, "inputs must be (N, n_active)"
        return out

    def fk(self, q):
        (q,) = self._in(q)
        T = np.empty((len(q), self.L, 3, 4))
        for s in range(len(q)):
            lib().orc_fk(self._h, _p(q[s]), _p(T[s]))
        return T

    def jacobian(self, q):
        (q,) = self._in(q)
        J = np.empty((len(q), self.n, 6))
        for s in range(len(q)):
            lib().orc_jacobian(self._h, _p(q[s]), _p(J[s]))
        return np.transpose(J, (0, 2, 1))  # (N, 6, n)

    def jacobian_link(self, q, link_index):
        (q,) = self._in(q)
        J = np.empty((len(q), self.n, 6))
        li = np.array([float(link_index)])
        for s in range(len(q)):
            lib().orc_jacobian_link(self._h, _p(q[s]), _p(li), _p(J[s]))
        return np.transpose(J, (0, 2, 1))  # (N, 6, n)

    def local_ik(self, T_target, seed, weight=None, toll=1e-4, max_iter=100, damping=0.0):
        """computeLocalIk / computeWeigthedLocalIk per sample.  T_target (N, 3, 4); returns (sol (N, n), status (N,), iterations (N,))."""
        (seed,) = self._in(seed)
        T = _c(T_target).reshape(len(seed), 12)
        w = None if weight is None else _c(weight)
        lo, hi = _c(self.spec.q_min), _c(self.spec.q_max)
        sol = np.empty_like(seed)
        status = np.empty(len(seed), dtype=np.int32)
        iters = np.empty(len(seed), dtype=np.int32)
        it = C.c_int(0)
        for s in range(len(seed)):
            status[s] = lib().orc_local_ik(self._h, self.n, self.L, _p(T[s]), _p(seed[s]), _p(w), _p(lo), _p(hi), float(toll),
                                           float(damping), int(max_iter), _p(sol[s]), C.byref(it))
            iters[s] = it.value
        return sol, status, iters

    def twist(self, q, dq):
        q, dq = self._in(q, dq)
        tw = np.empty((len(q), self.L, 6))
        for s in range(len(q)):
            lib().orc_twist(self._h, _p(q[s]), _p(dq[s]), _p(tw[s]))
        return tw

    def dtwist(self, q, dq, ddq, parts=False):
        q, dq, ddq = self._in(q, dq, ddq)
        a = np.empty((len(q), self.L, 6))
        al = np.empty_like(a) if parts else None
        an = np.empty_like(a) if parts else None
        for s in range(len(q)):
            lib().orc_dtwist(self._h, _p(q[s]), _p(dq[s]), _p(ddq[s]), _p(a[s]),
                             _p(al[s]) if parts else None, _p(an[s]) if parts else None)
        return (a, al, an) if parts else a

    def ddtwist(self, q, dq, ddq, dddq):
        q, dq, ddq, dddq = self._in(q, dq, ddq, dddq)
        j = np.empty((len(q), self.L, 6))
        for s in range(len(q)):
            lib().orc_ddtwist(self._h, _p(q[s]), _p(dq[s]), _p(ddq[s]), _p(dddq[s]), _p(j[s]))
        return j

    def ddtwist_parts(self, q, dq, ddq, dddq):
        """(getDDTwistLinearPart, getDDTwistNonLinearPart), each (N, L, 6)."""
        q, dq, ddq, dddq = self._in(q, dq, ddq, dddq)
        jl, jn = np.empty((len(q), self.L, 6)), np.empty((len(q), self.L, 6))
        for s in range(len(q)):
            lib().orc_ddtwist_parts(self._h, _p(q[s]), _p(dq[s]), _p(ddq[s]), _p(dddq[s]), _p(jl[s]), _p(jn[s]))
        return jl, jn

    def joint_torque(self, q, dq, ddq, ext=None, wrenches=False):
        q, dq, ddq = self._in(q, dq, ddq)
        tau = np.empty((len(q), self.n))
        w = np.empty((len(q), self.L, 6)) if wrenches else None
        if ext is not None:
            ext = _c(ext).reshape(len(q), self.L, 6)
        for s in range(len(q)):
            lib().orc_joint_torque(self._h, _p(q[s]), _p(dq[s]), _p(ddq[s]), _p(ext[s]) if ext is not None else None,
                                   _p(tau[s]), _p(w[s]) if wrenches else None)
        return (tau, w) if wrenches else tau

    def regressor(self, q, dq, ddq):
        q, dq, ddq = self._in(q, dq, ddq)
        Y = np.empty((len(q), self.P, self.n))  # column-major n x P per sample
        for s in range(len(q)):
            lib().orc_regressor(self._h, _p(q[s]), _p(dq[s]), _p(ddq[s]), _p(Y[s]))
        return np.transpose(Y, (0, 2, 1))  # (N, n, P)

    def joint_inertia(self, q):
        (q,) = self._in(q)
        M = np.empty((len(q), self.n, self.n))
        for s in range(len(q)):
            lib().orc_joint_inertia(self._h, _p(q[s]), _p(M[s]))
        return np.transpose(M, (0, 2, 1))

    def nominal_parameters(self):
        pi = np.zeros(self.P)
        lib().orc_nominal_parameters(self._h, _p(pi))
        return pi

    def batch_torque_regressor(self, q, dq, ddq, threads=1, want_tau=True, want_Y=True, bufs=None):
        """C-side batch loop (optionally OpenMP) -- the cpu_baseline leg.  Returns (tau, Y (N, n, P), threads).
        `bufs` = (tau (N, n), Y (N, P, n)) pre-touched output buffers to keep page faults out of a timing."""
        q, dq, ddq = self._in(q, dq, ddq)
        N = len(q)
        if bufs is not None:
            tau, Y = bufs
            assert tau.shape == (N, self.n) and Y.shape == (N, self.P, self.n) and tau.flags.c_contiguous and Y.flags.c_contiguous
        else:
            tau = np.empty((N, self.n)) if want_tau else None
            Y = np.empty((N, self.P, self.n)) if want_Y else None
        used = lib().orc_batch_torque_regressor(self._h, N, _p(q), _p(dq), _p(ddq), _p(tau), _p(Y), int(threads))
        return tau, (np.transpose(Y, (0, 2, 1)) if Y is not None else None), used
