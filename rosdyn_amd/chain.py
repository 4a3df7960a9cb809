"""Batched mirror of ``rosdyn::Chain`` (rosdyn_core/include/rosdyn_core/primitives.h:235-555) over the
C-ABI of librdyn_hip.so.  Method names and argument meaning are the reference's; every ``q/Dq/DDq`` is a
batch: a ``torch.float64`` CUDA tensor of shape (N, n_active) (sample-major) or (n_active, N) with
``layout="element"``.  Outputs are CUDA tensors that alias nothing (fresh allocations, or the caller's
``out=``).  The arithmetic happens in the HIP kernels; torch only owns the memory and the stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import LAYOUT_ELEMENT_MAJOR, LAYOUT_SAMPLE_MAJOR, Batch, RegressorLayout, check, lib


def _torch():
    import torch
    return torch


class Chain(object):
    """rosdyn::createChain(urdf, base_frame, tool_frame, gravity)  (primitives.h:566)."""

    def __init__(self, urdf_xml, base_frame, tool_frame, gravity=(0.0, 0.0, 0.0), _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        if "<robot" not in urdf_xml:
            with open(urdf_xml) as f:
                urdf_xml = f.read()
        h = C.c_void_p()
        g = (C.c_double * 3)(*[float(x) for x in gravity])
        check(lib().rdyn_chain_from_urdf(urdf_xml.encode(), base_frame.encode(), tool_frame.encode(), g, C.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib._lib is not None:
            _lib._lib.rdyn_chain_destroy(h)
            self._h = None

    def clone(self):                                  # primitives.h:554
        h = C.c_void_p()
        check(lib().rdyn_chain_clone(self._h, C.byref(h)))
        return Chain(None, None, None, _handle=h)

    # ---- getters, primitives.h:364-447
    def getLinksNumber(self):
        return lib().rdyn_chain_links_number(self._h)

    def getJointsNumber(self):
        return lib().rdyn_chain_joints_number(self._h)

    def getActiveJointsNumber(self):
        return lib().rdyn_chain_active_joints_number(self._h)

    def getLinksName(self):
        return [lib().rdyn_chain_link_name(self._h, i).decode() for i in range(self.getLinksNumber())]

    def getJointsName(self):
        return [lib().rdyn_chain_joint_name(self._h, i).decode() for i in range(self.getJointsNumber())]

    def getMoveableJointNames(self):
        return [lib().rdyn_chain_moveable_joint_name(self._h, i).decode()
                for i in range(lib().rdyn_chain_moveable_joints_number(self._h))]

    def getActiveJointsName(self):
        return [lib().rdyn_chain_active_joint_name(self._h, i).decode() for i in range(self.getActiveJointsNumber())]

    def getJointTypes(self):
        return [lib().rdyn_chain_joint_type(self._h, i) for i in range(self.getJointsNumber())]

    def getGravity(self):
        g = (C.c_double * 3)()
        check(lib().rdyn_chain_gravity(self._h, g))
        return np.array(g[:])

    def setInputJointsName(self, names):              # primitives.h:362; returns bool like the reference
        arr = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
        st = lib().rdyn_chain_set_input_joints(self._h, arr, len(names))
        if st == 6:
            return False
        check(st)
        return True

    def _limits(self, which):
        n = self.getActiveJointsNumber()
        bufs = [np.zeros(n) for _ in range(5)]
        check(lib().rdyn_chain_limits(self._h, *[b.ctypes.data_as(C.POINTER(C.c_double)) for b in bufs]))
        return bufs[which]

    def getQMax(self):
        return self._limits(0)

    def getQMin(self):
        return self._limits(1)

    def getDQMax(self):
        return self._limits(2)

    def getDDQMax(self):
        return self._limits(3)

    def getTauMax(self):
        return self._limits(4)

    def getNominalParameters(self):                   # primitives.h:548
        pi = np.zeros(10 * self.getJointsNumber())
        check(lib().rdyn_nominal_parameters(self._h, pi.ctypes.data_as(C.POINTER(C.c_double))))
        return pi

    def getBodyReduction(self):
        """Rigid-body reduction (include/rdyn.h: rdyn_chain_reduction; no reference counterpart): None when every chain joint is an
        input joint, else (body_joint (nJ,), X (nJ, 10, 10), pi_body (bodies, 10)) with
        Y[:, 10 f:10 f + 10] = Y[:, 10 b:10 b + 10] @ X[f], b = body_joint[f] (zero columns where b < 0)."""
        nj = self.getJointsNumber()
        body = np.zeros(nj, dtype=np.int32)
        X = np.zeros((nj, 10, 10))
        nb = lib().rdyn_chain_reduction(self._h, None, None, None)
        if nb <= 0:
            return None
        pi = np.zeros((nb, 10))
        lib().rdyn_chain_reduction(self._h, body.ctypes.data, X.ctypes.data, pi.ctypes.data)
        return body, X, pi

    # ---- batch plumbing
    def _batch(self, layout, q, dq=None, ddq=None):
        torch = _torch()
        n = self.getActiveJointsNumber()
        lay = LAYOUT_ELEMENT_MAJOR if layout == "element" else LAYOUT_SAMPLE_MAJOR
        ts = [t for t in (q, dq, ddq) if t is not None]
        for t in ts:
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or t.dim() != 2:
                raise ValueError("inputs must be contiguous 2-D float64 CUDA tensors")
            if t.shape != q.shape or t.device != q.device:
                raise ValueError("Input data dimensions mismatch")   # primitives_impl.h:1302
        N = q.shape[1] if lay == LAYOUT_ELEMENT_MAJOR else q.shape[0]
        nin = q.shape[0] if lay == LAYOUT_ELEMENT_MAJOR else q.shape[1]
        if nin != n:
            raise ValueError("Input data dimensions mismatch")
        b = Batch()
        b.n_samples = N
        b.q = q.data_ptr()
        b.dq = dq.data_ptr() if dq is not None else None
        b.ddq = ddq.data_ptr() if ddq is not None else None
        b.layout = lay
        b.device = q.device.index if q.device.index is not None else -1
        b.stream = torch.cuda.current_stream(q.device).cuda_stream
        return b, N, lay

    def _out(self, q, N, lay, rec_shape, out=None):
        torch = _torch()
        shape = (N,) + tuple(rec_shape) if lay == LAYOUT_SAMPLE_MAJOR else tuple(rec_shape) + (N,)
        if out is None:
            return torch.empty(shape, dtype=torch.float64, device=q.device)
        if tuple(out.shape) != shape or out.dtype != torch.float64 or not out.is_contiguous() or out.device != q.device:
            raise ValueError("out must be a contiguous float64 tensor of shape %s" % (shape,))
        return out

    def evaluateAll(self, q, Dq, DDq, layout="sample"):
        """include/rdyn.h: rdyn_evaluate_all -- every getter of the samples in ONE launch (chains of up to 10 joints; longer chains: the
        single-purpose launches).  Returns a dict with the records the single-purpose getters return: T_links, J, twists, dtwists, tau,
        tau_nonlinear, M, Y (per-sample Eigen images (N, P, n), or element-major (P, n, N))."""
        from ._lib import RegressorLayout
        b, N, lay = self._batch(layout, q, Dq, DDq)
        n, L, P = self.getActiveJointsNumber(), self.getLinksNumber(), 10 * self.getJointsNumber()
        o = dict(T_links=self._out(q, N, lay, (L, 4, 3)), J=self._out(q, N, lay, (n, 6)), twists=self._out(q, N, lay, (L, 6)),
                 dtwists=self._out(q, N, lay, (L, 6)), tau=self._out(q, N, lay, (n,)), tau_nonlinear=self._out(q, N, lay, (n,)),
                 M=self._out(q, N, lay, (n, n)), Y=self._out(q, N, lay, (P, n)))
        yl = RegressorLayout(n * P, 1, n) if lay == LAYOUT_SAMPLE_MAJOR else RegressorLayout(1, N, n * N)

        class _Out(C.Structure):
            _fields_ = [(k, C.c_void_p) for k in ("T_links", "J", "twists", "dtwists", "tau", "tau_nonlinear", "M", "Y", "y_layout")]
        rec = _Out(*([o[k].data_ptr() for k in ("T_links", "J", "twists", "dtwists", "tau", "tau_nonlinear", "M", "Y")] + [C.addressof(yl)]))
        check(lib().rdyn_evaluate_all(self._h, C.byref(b), C.byref(rec)))
        return o

    # ---- kinematics, primitives.h:452-463.  Record shapes are the transposes of the Eigen (column-major) objects:
    #      sample-major T[s] is (4, 3) = columns of the 3x4 [R | p]; use .transpose(-1, -2) for the matrix.
    def getTransformation(self, q, layout="sample", out=None):
        b, N, lay = self._batch(layout, q)
        T = self._out(q, N, lay, (4, 3), out)
        check(lib().rdyn_transformation(self._h, C.byref(b), T.data_ptr(), None))
        return T

    def getTransformations(self, q, layout="sample", out=None):
        b, N, lay = self._batch(layout, q)
        T = self._out(q, N, lay, (self.getLinksNumber(), 4, 3), out)
        check(lib().rdyn_transformation(self._h, C.byref(b), None, T.data_ptr()))
        return T

    def getJacobian(self, q, layout="sample", out=None):
        b, N, lay = self._batch(layout, q)
        J = self._out(q, N, lay, (self.getActiveJointsNumber(), 6), out)
        check(lib().rdyn_jacobian(self._h, C.byref(b), J.data_ptr()))
        return J

    def _link_index(self, link_name):
        names = self.getLinksName()
        if link_name not in names:
            raise ValueError("link " + link_name + " is not member of the chain")     # primitives_impl.h:920, 960, 1022
        return names.index(link_name)

    def getJacobianLink(self, q, link_name, layout="sample", out=None):              # primitives.h:456
        b, N, lay = self._batch(layout, q)
        J = self._out(q, N, lay, (self.getActiveJointsNumber(), 6), out)
        check(lib().rdyn_jacobian_link(self._h, C.byref(b), self._link_index(link_name), J.data_ptr()))
        return J

    def getTransformationLink(self, q, link_name, layout="sample"):                  # primitives.h:453
        T = self.getTransformations(q, layout=layout)
        i = self._link_index(link_name)
        return T[:, i] if layout != "element" else T[i]

    def getTwistLink(self, q, Dq, link_name, layout="sample"):                       # primitives.h:458
        tw = self.getTwist(q, Dq, layout=layout)
        i = self._link_index(link_name)
        return tw[:, i] if layout != "element" else tw[i]

    # ---- local inverse kinematics, primitives.h:510, 526 (batched: one pose per batch entry)
    def computeLocalIk(self, T_b_t, seed, toll=1e-4, max_iterations=100, weight=None, layout="sample", out=None, damping=0.0):
        """Returns (sol, status, iterations); status 1 = the reference's `true`, 0 = `false` (iteration cap instead of
        the reference's max_time), < 0 = the QP of that pose failed (see include/rdyn.h).  T_b_t: the record
        getTransformation returns ((N, 4, 3) sample-major / (4, 3, N) element-major)."""
        torch = _torch()
        b, N, lay = self._batch(layout, seed)
        shape = (N, 4, 3) if lay == LAYOUT_SAMPLE_MAJOR else (4, 3, N)
        if (tuple(T_b_t.shape) != shape or T_b_t.dtype != torch.float64 or not T_b_t.is_contiguous()
                or T_b_t.device != seed.device):
            raise ValueError("T_b_t must be a contiguous float64 tensor of shape %s" % (shape,))
        sol = self._out(seed, N, lay, (self.getActiveJointsNumber(),), out)
        status = torch.empty(N, dtype=torch.int32, device=seed.device)
        iters = torch.empty(N, dtype=torch.int32, device=seed.device)
        w = None
        if weight is not None:
            w = (C.c_double * 6)(*[float(v) for v in weight])
        # damping > 0: Levenberg term (no reference counterpart; what 7-DOF arms need, see include/rdyn.h)
        check(lib().rdyn_local_ik_damped(self._h, C.byref(b), T_b_t.data_ptr(), w, float(toll), float(damping), int(max_iterations),
                                         sol.data_ptr(), status.data_ptr(), iters.data_ptr()))
        return sol, status, iters

    def computeWeigthedLocalIk(self, T_b_t, weight, seed, toll=1e-4, max_iterations=100, layout="sample", out=None, damping=0.0):
        return self.computeLocalIk(T_b_t, seed, toll, max_iterations, weight=weight, layout=layout, out=out, damping=damping)

    def getMultiplicity(self, q):                                                    # primitives_impl.h:1470-1516
        """Host-side: every joint vector equal to q up to whole turns of the revolute input joints within the limits."""
        q = np.asarray(q, dtype=np.float64)
        qmax, qmin = self.getQMax(), self.getQMin()
        names, types = self.getJointsName(), self.getJointTypes()
        axes = []
        for idx, name in enumerate(self.getActiveJointsName()):
            vals = [q[idx]]
            if types[names.index(name)] == 0:   # REVOLUTE (continuous included, primitives_impl.h:74-77)
                if qmax[idx] - qmin[idx] > 2 * np.pi * 1e4:
                    # the reference enumerates every turn up to the 1e10 default limit; refuse instead of exhausting memory
                    raise ValueError("getMultiplicity: joint %s has no finite position limits" % name)
                tmp = q[idx]
                while True:
                    tmp += 2 * np.pi
                    if tmp > qmax[idx]:
                        break
                    vals.append(tmp)
                tmp = q[idx]
                while True:
                    tmp -= 2 * np.pi
                    if tmp < qmin[idx]:
                        break
                    vals.append(tmp)
            axes.append(vals)
        multiturn = [q.copy()]
        for idx, vals in enumerate(axes):
            size = len(multiturn)
            for v in vals[1:]:
                for im in range(size):
                    new_q = multiturn[im].copy()
                    new_q[idx] = v
                    multiturn.append(new_q)
        return multiturn

    def getTwist(self, q, Dq, layout="sample", out=None):
        b, N, lay = self._batch(layout, q, Dq)
        tw = self._out(q, N, lay, (self.getLinksNumber(), 6), out)
        check(lib().rdyn_twist(self._h, C.byref(b), tw.data_ptr(), None))
        return tw

    def getDTwist(self, q, Dq, DDq, layout="sample", out=None):
        b, N, lay = self._batch(layout, q, Dq, DDq)
        tw = self._out(q, N, lay, (self.getLinksNumber(), 6), out)
        check(lib().rdyn_twist(self._h, C.byref(b), None, tw.data_ptr()))
        return tw

    def _parts(self, q, Dq, DDq, DDDq, layout, which):
        b, N, lay = self._batch(layout, q, Dq, DDq)
        out = self._out(q, N, lay, (self.getLinksNumber(), 6))
        ptrs = [None, None, None]
        ptrs[which] = out.data_ptr()
        check(lib().rdyn_twist_parts(self._h, C.byref(b), DDDq.data_ptr() if DDDq is not None else None, *ptrs))
        return out

    def getDTwistLinearPart(self, q, DDq, layout="sample"):            # primitives.h:468
        return self._parts(q, None, DDq, None, layout, 0)

    def getDTwistNonLinearPart(self, q, Dq, layout="sample"):          # primitives.h:473
        return self._parts(q, Dq, None, None, layout, 1)

    def getDDTwist(self, q, Dq, DDq, DDDq, layout="sample"):           # primitives.h:488
        return self._parts(q, Dq, DDq, DDDq, layout, 2)

    def getDDTwistLinearPart(self, q, DDDq, layout="sample"):          # primitives.h:476
        b, N, lay = self._batch(layout, q)
        out = self._out(q, N, lay, (self.getLinksNumber(), 6))
        check(lib().rdyn_jerk_parts(self._h, C.byref(b), DDDq.data_ptr(), out.data_ptr(), None))
        return out

    def getDDTwistNonLinearPart(self, q, Dq, DDq, layout="sample"):    # primitives.h:480
        b, N, lay = self._batch(layout, q, Dq, DDq)
        out = self._out(q, N, lay, (self.getLinksNumber(), 6))
        check(lib().rdyn_jerk_parts(self._h, C.byref(b), None, None, out.data_ptr()))
        return out

    def getWrench(self, q, Dq, DDq, ext_wrenches_in_link_frame=None, layout="sample", out=None):      # primitives.h:530
        """Link wrenches (N, L, 6) / (L, 6, N), base-frame coordinates at the link origins; [:, -1] is getWrenchTool."""
        b, N, lay = self._batch(layout, q, Dq, DDq)
        w = self._out(q, N, lay, (self.getLinksNumber(), 6), out)
        ext = ext_wrenches_in_link_frame.data_ptr() if ext_wrenches_in_link_frame is not None else None
        check(lib().rdyn_wrench(self._h, C.byref(b), ext, w.data_ptr()))
        return w

    def jointIndex(self, name):                                         # primitives.h:447: -1 when not an input joint
        names = self.getActiveJointsName()
        return names.index(name) if name in names else -1

    def getJointTorqueExt(self, q, Dq, DDq, ext_wrenches_in_link_frame, layout="sample", out=None):   # primitives.h:539
        """ext: (N, L, 6) for layout="sample", (L, 6, N) for "element"."""
        b, N, lay = self._batch(layout, q, Dq, DDq)
        tau = self._out(q, N, lay, (self.getActiveJointsNumber(),), out)
        check(lib().rdyn_joint_torque_ext(self._h, C.byref(b), ext_wrenches_in_link_frame.data_ptr(), tau.data_ptr()))
        return tau

    # ---- dynamics, primitives.h:539-547
    def getJointTorque(self, q, Dq, DDq, layout="sample", out=None):
        b, N, lay = self._batch(layout, q, Dq, DDq)
        tau = self._out(q, N, lay, (self.getActiveJointsNumber(),), out)
        check(lib().rdyn_joint_torque(self._h, C.byref(b), tau.data_ptr()))
        return tau

    def getJointTorqueNonLinearPart(self, q, Dq, layout="sample", out=None):
        b, N, lay = self._batch(layout, q, Dq)
        tau = self._out(q, N, lay, (self.getActiveJointsNumber(),), out)
        check(lib().rdyn_joint_torque_nonlinear(self._h, C.byref(b), tau.data_ptr()))
        return tau

    def getJointInertia(self, q, layout="sample", out=None):
        b, N, lay = self._batch(layout, q)
        n = self.getActiveJointsNumber()
        M = self._out(q, N, lay, (n, n), out)
        check(lib().rdyn_joint_inertia(self._h, C.byref(b), M.data_ptr()))
        return M

    def getRegressor(self, q, Dq, DDq, layout="sample", y_layout=None, out=None, tau_out=None, with_torque=False):
        """Regressor (and optionally the fused joint torque).

        y_layout: "per_sample" -> (N, P, n): Y[s] is the column-major n x P Eigen image (default for layout="sample");
                  "stacked"    -> (P, N*n):  column-major (N*n) x P, row = s*n + j;
                  "element"    -> (P, n, N): column-major (n*N) x P, row = j*N + s (default for layout="element").
        """
        torch = _torch()
        b, N, lay = self._batch(layout, q, Dq, DDq)
        n, P = self.getActiveJointsNumber(), 10 * self.getJointsNumber()
        if y_layout is None:
            y_layout = "element" if lay == LAYOUT_ELEMENT_MAJOR else "per_sample"
        if y_layout == "per_sample":
            shape, yl = (N, P, n), RegressorLayout(n * P, 1, n)
        elif y_layout == "stacked":
            shape, yl = (P, N * n), RegressorLayout(n, 1, N * n)
        elif y_layout == "element":
            shape, yl = (P, n, N), RegressorLayout(1, N, n * N)
        else:
            raise ValueError("unknown y_layout %r" % (y_layout,))
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=q.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float64 tensor of shape %s" % (shape,))
        tau = None
        if with_torque or tau_out is not None:
            tau = self._out(q, N, lay, (n,), tau_out)
        check(lib().rdyn_regressor(self._h, C.byref(b), tau.data_ptr() if tau is not None else None, out.data_ptr(), C.byref(yl)))
        return (out, tau) if tau is not None else out


    def getRegressorGram(self, q, Dq, DDq, tau_meas=None, layout="sample", chunk_samples=0, out=None, accumulate=False,
                         workspace=None, packed_out=None):
        """Normal equations of the stacked regressor of this batch without leaving Y in HBM:
        returns (G = A^T A (P, P), c = A^T tau_meas (P,), bb = tau_meas^T tau_meas (1,)).  include/rdyn.h: rdyn_regressor_gram.
        packed_out: a (P*P + P + 2,) float64 buffer [G | c | bb | count] -- the all-reduce payload of the multi-GPU path; G, c, bb are
        written straight into it (the returned tensors are views), the count slot is the caller's (rosdyn_amd.gram.packed_buffer)."""
        torch = _torch()
        b, N, lay = self._batch(layout, q, Dq, DDq)
        P = 10 * self.getJointsNumber()
        if tau_meas is not None and (tau_meas.shape != q.shape or tau_meas.dtype != torch.float64 or not tau_meas.is_contiguous()):
            raise ValueError("Input data dimensions mismatch")
        if packed_out is not None:
            if out is not None:
                raise ValueError("packed_out and out are exclusive")
            if packed_out.dtype != torch.float64 or packed_out.numel() != P * P + P + 2 or not packed_out.is_contiguous() or packed_out.device != q.device:
                raise ValueError("packed_out must be a contiguous float64 tensor of P*P + P + 2 elements on the batch's device")
            flat = packed_out.view(-1)
            out = (flat[:P * P].view(P, P), flat[P * P:P * P + P], flat[P * P + P:P * P + P + 1])
        if out is None:
            out = (torch.empty((P, P), dtype=torch.float64, device=q.device), torch.empty((P,), dtype=torch.float64, device=q.device),
                   torch.empty((1,), dtype=torch.float64, device=q.device))
            if accumulate:
                raise ValueError("accumulate needs out=")
        G, c, bb = out
        nbytes = lib().rdyn_regressor_gram_workspace_bytes(self._h, chunk_samples)
        if workspace is None:
            workspace = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        check(lib().rdyn_regressor_gram(self._h, C.byref(b), tau_meas.data_ptr() if tau_meas is not None else None,
                                        G.data_ptr(), c.data_ptr(), bb.data_ptr(), 1 if accumulate else 0, chunk_samples,
                                        workspace.data_ptr(), workspace.numel()))
        return G, c, bb

    def getRegressorTsqr(self, q, Dq, DDq, tau_meas=None, layout="sample", out=None, accumulate=False, workspace=None):
        """R factor of the stacked [regressor | tau_meas] of this batch WITHOUT forming A'A (include/rdyn.h: rdyn_regressor_tsqr,
        BASELINE.json configs[2]).  Returns R1 = [R d; 0 rho] as a (P + 1, P + 1) tensor in math layout (upper triangular)."""
        torch = _torch()
        b, N, lay = self._batch(layout, q, Dq, DDq)
        P1 = 10 * self.getJointsNumber() + 1
        if tau_meas is not None and (tau_meas.shape != q.shape or tau_meas.dtype != torch.float64 or not tau_meas.is_contiguous()):
            raise ValueError("Input data dimensions mismatch")
        buf = torch.zeros((P1, P1), dtype=torch.float64, device=q.device) if out is None else out.t().contiguous()
        nbytes = lib().rdyn_regressor_tsqr_workspace_bytes(self._h)
        if nbytes == 0:
            raise ValueError("rdyn_regressor_tsqr: chains of 1..8 input joints in chain order, at most 112 columns after the reduction")
        if workspace is None:
            workspace = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        check(lib().rdyn_regressor_tsqr(self._h, C.byref(b), tau_meas.data_ptr() if tau_meas is not None else None, buf.data_ptr(),
                                        1 if accumulate else 0, workspace.data_ptr(), workspace.numel()))
        if out is not None:           # the C side is column-major: `buf` was a transposed copy of `out`
            out.copy_(buf.t())
            return out
        return buf.t()

    def getIdentificationGram(self, components, q, Dq, DDq, tau_meas, layout="sample", out=None, accumulate=False, workspace=None):
        """The identification step in one call (include/rdyn.h: rdyn_identification_gram): normal equations of [Y | C] with
        the measured torque; `components` is a rosdyn_amd.components.ComponentSet or None.  Returns (G (P+K, P+K), c (P+K,),
        bb (1,)); the unknowns are [the 10-per-link inertial parameters ; the components' parameters]."""
        torch = _torch()
        b, N, lay = self._batch(layout, q, Dq, DDq)
        if tau_meas.shape != q.shape or tau_meas.dtype != torch.float64 or not tau_meas.is_contiguous():
            raise ValueError("Input data dimensions mismatch")
        arr, n_comps = (C.cast(components._arr, C.c_void_p), components.n_comps) if components is not None else (None, 0)
        cols = 10 * self.getJointsNumber() + (components.columns if components is not None else 0)
        if out is None:
            out = (torch.empty((cols, cols), dtype=torch.float64, device=q.device), torch.empty((cols,), dtype=torch.float64, device=q.device),
                   torch.empty((1,), dtype=torch.float64, device=q.device))
            if accumulate:
                raise ValueError("accumulate needs out=")
        G, c, bb = out
        nbytes = lib().rdyn_identification_gram_workspace_bytes(self._h, arr, n_comps)
        if nbytes == 0:
            raise ValueError("at most 111 columns (regressor + components) are supported")
        if workspace is None:
            workspace = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        check(lib().rdyn_identification_gram(self._h, arr, n_comps, C.byref(b), tau_meas.data_ptr(), G.data_ptr(), c.data_ptr(),
                                             bb.data_ptr(), 1 if accumulate else 0, workspace.data_ptr(), workspace.numel()))
        return G, c, bb


    def getIdentificationTsqr(self, components, q, Dq, DDq, tau_meas, layout="sample", out=None, accumulate=False, workspace=None):
        """R factor of the identification step's stacked [Y | C | tau_meas] WITHOUT forming the normal equations (include/rdyn.h:
        rdyn_identification_tsqr).  Returns R1 = [R d; 0 rho] as a (P + K + 1, P + K + 1) tensor in math layout (upper triangular);
        solve with rosdyn_amd.gram.solve_r_factor."""
        torch = _torch()
        b, N, lay = self._batch(layout, q, Dq, DDq)
        if tau_meas.shape != q.shape or tau_meas.dtype != torch.float64 or not tau_meas.is_contiguous():
            raise ValueError("Input data dimensions mismatch")
        arr, n_comps = (C.cast(components._arr, C.c_void_p), components.n_comps) if components is not None else (None, 0)
        n1 = 10 * self.getJointsNumber() + (components.columns if components is not None else 0) + 1
        buf = torch.zeros((n1, n1), dtype=torch.float64, device=q.device) if out is None else out.t().contiguous()
        nbytes = lib().rdyn_identification_tsqr_workspace_bytes(self._h, arr, n_comps)
        if nbytes == 0:
            raise ValueError("rdyn_identification_tsqr: chains of 1..8 input joints in chain order, at most 112 columns after the reduction")
        if workspace is None:
            workspace = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        check(lib().rdyn_identification_tsqr(self._h, arr, n_comps, C.byref(b), tau_meas.data_ptr(), buf.data_ptr(), 1 if accumulate else 0,
                                             workspace.data_ptr(), workspace.numel()))
        R1 = buf.t()   # the library writes column-major
        if out is not None:
            out.copy_(R1)
            return out
        return R1


    def lastTsqrReport(self, n_samples, workspace, components=None):
        """What the last getRegressorTsqr / getIdentificationTsqr call that used `workspace` did (include/rdyn.h:
        rdyn_tsqr_last_report): dict(route = 0 Householder folds / 1 preconditioned CholeskyQR; stage = 0 accepted after round 0,
        1 after round 1, 2 the stand-by Householder factorisation ran; n_deferred; gamma, rho of the two rounds).  Synchronises."""
        from ._lib import RdynTsqrReport
        torch = _torch()
        arr, n_comps = (C.cast(components._arr, C.c_void_p), components.n_comps) if components is not None else (None, 0)
        rep = RdynTsqrReport()
        stream = torch.cuda.current_stream(workspace.device).cuda_stream
        check(lib().rdyn_tsqr_last_report(self._h, arr, n_comps, int(n_samples), workspace.data_ptr(), workspace.device.index or 0, stream,
                                          C.byref(rep)))
        return dict(route=rep.route, stage=rep.stage, n_deferred=rep.n_deferred, gamma=tuple(rep.gamma), rho=tuple(rep.rho))


def createChain(urdf_xml, base_frame, tool_frame, gravity=(0.0, 0.0, 0.0)):
    return Chain(urdf_xml, base_frame, tool_frame, gravity)
