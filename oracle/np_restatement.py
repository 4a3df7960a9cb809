"""Independent numpy restatement of the hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Second, differently-structured statement of the same functions as oracle/rosdyn_oracle.c, used to
(i) cross-check the literal C oracle and (ii) generate the golden fixtures under tests/golden/
(tests/golden/make_golden.py).  It deliberately takes other routes than the C file:

* rotations from rpy as R = Rz(yaw) Ry(pitch) Rx(roll) (no quaternion detour);
* vectorised over the batch;
* the regressor in the closed form  W' = [ d | [alpha]x + [w]x[w]x | 0 ; 0 | -[d]x | L(alpha) + [w]x L(w) ]
  instead of the ten dense 6x6 basis matrices (primitives_impl.h:1324-1333), and the O(n^2) row
  assembly by composing wrench shifts link by link instead of the all-pairs translation (1343-1347);
* joint torques by a textbook RNEA with 3-vector Newton/Euler equations about the centre of mass
  (NOT the spatial-inertia form of primitives_impl.h:1240-1250);
* the joint inertia matrix from unit-acceleration RNEA calls (M e_k = tau(q,0,e_k) - tau(q,0,0)).

PARITY UNPINNED at the reference level -- see oracle/rosdyn_oracle.c.
Conventions (reference): 6-vectors are [lin; ang] (spacevect_algebra.h:44-52); base-frame coordinates,
reference point = link origin; Dtwists are spatial accelerations with zero base acceleration.
"""
import numpy as np

from . import urdf_model


def _rot_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _L(x):
    """(..., 3) -> (..., 3, 6) with  I x = L(x) [Ixx Ixy Ixz Iyy Iyz Izz]^T."""
    z = np.zeros_like(x[..., 0])
    r0 = np.stack([x[..., 0], x[..., 1], x[..., 2], z, z, z], -1)
    r1 = np.stack([z, x[..., 0], z, x[..., 1], x[..., 2], z], -1)
    r2 = np.stack([z, z, x[..., 0], z, x[..., 1], x[..., 2]], -1)
    return np.stack([r0, r1, r2], -2)


def _skew_b(x):
    z = np.zeros_like(x[..., 0])
    return np.stack([np.stack([z, -x[..., 2], x[..., 1]], -1),
                     np.stack([x[..., 2], z, -x[..., 0]], -1),
                     np.stack([-x[..., 1], x[..., 0], z], -1)], -2)


class NpChain(object):
    def __init__(self, urdf, base, tool, gravity=(0.0, 0.0, 0.0), input_joint_names=None):
        spec = urdf_model.load(urdf, base, tool, gravity, input_joint_names)
        self.spec = spec
        self.nJ, self.L, self.n, self.P = spec.n_joints, spec.n_links, spec.n_active, 10 * spec.n_joints
        self.g = np.array(spec.gravity, dtype=float)
        self.jtype, self.R_pj, self.t_pj, self.u = [], [], [], []
        for j in spec.joints:
            self.jtype.append("R" if j.urdf_type in (0, 1) else ("P" if j.urdf_type == 2 else "F"))
            self.R_pj.append(_rot_rpy(*j.rpy))
            self.t_pj.append(np.array(j.xyz, dtype=float))
            a = np.array(j.axis, dtype=float)
            nrm = np.linalg.norm(a)
            self.u.append(a / nrm if nrm > 0 else a)
        self.mass, self.com, self.Icom, self.Iorigin = [], [], [], []
        for l in spec.links:
            m = l.mass if l.has_inertial else 0.0
            c = np.array(l.xyz if l.has_inertial else [0, 0, 0], dtype=float)
            ixx, ixy, ixz, iyy, iyz, izz = l.inertia if l.has_inertial else [0.0] * 6
            I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
            Rc = _rot_rpy(*l.rpy) if l.has_inertial else np.eye(3)
            Ic = Rc @ I @ Rc.T
            self.mass.append(m)
            self.com.append(c)
            self.Icom.append(Ic)
            self.Iorigin.append(Ic + m * (_skew(c) @ _skew(c).T))
        self.in_of = [-1] * self.nJ
        for k, c in enumerate(spec.input_chain_index):
            self.in_of[c] = k

    # -------------------------------------------------------------- helpers
    def _sorted(self, x):
        x = np.atleast_2d(np.asarray(x, dtype=float))
        out = np.zeros((x.shape[0], self.nJ))
        for c in range(self.nJ):
            if self.in_of[c] >= 0:
                out[:, c] = x[:, self.in_of[c]]
        return out

    def _forward(self, q, dq=None, ddq=None):
        sq = self._sorted(q)
        N = sq.shape[0]
        sdq = self._sorted(dq) if dq is not None else np.zeros_like(sq)
        sddq = self._sorted(ddq) if ddq is not None else np.zeros_like(sq)
        R = [np.broadcast_to(np.eye(3), (N, 3, 3)).copy()]
        p = [np.zeros((N, 3))]
        z = [np.zeros((N, 3))]
        vl, va = [np.zeros((N, 3))], [np.zeros((N, 3))]
        al, aa = [np.zeros((N, 3))], [np.zeros((N, 3))]
        for l in range(1, self.L):
            j = l - 1
            u, K = self.u[j], _skew(self.u[j])
            if self.jtype[j] == "R":
                s, c = np.sin(sq[:, j])[:, None, None], np.cos(sq[:, j])[:, None, None]
                Rjc = np.eye(3) + s * K + (1 - c) * (K @ K)
                Rpc = self.R_pj[j] @ Rjc
                tpc = np.broadcast_to(self.t_pj[j], (N, 3))
            elif self.jtype[j] == "P":
                Rpc = np.broadcast_to(self.R_pj[j], (N, 3, 3))
                tpc = self.t_pj[j] + sq[:, j:j + 1] * (self.R_pj[j] @ u)
            else:
                Rpc = np.broadcast_to(self.R_pj[j], (N, 3, 3))
                tpc = np.broadcast_to(self.t_pj[j], (N, 3))
            d = np.einsum("nij,nj->ni", R[l - 1], tpc)
            p.append(p[l - 1] + d)
            R.append(R[l - 1] @ Rpc)
            zl = np.einsum("nij,j->ni", R[l - 1], self.R_pj[j] @ u)   # rotated by the PARENT frame
            z.append(zl)
            Sl = zl if self.jtype[j] == "P" else np.zeros((N, 3))
            Sa = zl if self.jtype[j] == "R" else np.zeros((N, 3))
            # twist
            nvl = vl[l - 1] + np.cross(va[l - 1], d) + Sl * sdq[:, j:j + 1]
            nva = va[l - 1] + Sa * sdq[:, j:j + 1]
            # v x S  (spatialCrossProduct)
            cl = np.cross(nva, Sl) + np.cross(nvl, Sa)
            ca = np.cross(nva, Sa)
            nal = al[l - 1] + np.cross(aa[l - 1], d) + cl * sdq[:, j:j + 1] + Sl * sddq[:, j:j + 1]
            naa = aa[l - 1] + ca * sdq[:, j:j + 1] + Sa * sddq[:, j:j + 1]
            vl.append(nvl); va.append(nva); al.append(nal); aa.append(naa)
        return dict(N=N, R=R, p=p, z=z, vl=vl, va=va, al=al, aa=aa, sdq=sdq, sddq=sddq)

    # -------------------------------------------------------------- API
    def fk(self, q):
        f = self._forward(q)
        return np.stack([np.concatenate([f["R"][l], f["p"][l][:, :, None]], 2) for l in range(self.L)], 1)

    def jacobian(self, q):
        f = self._forward(q)
        J = np.zeros((f["N"], 6, self.n))
        pt = f["p"][self.L - 1]
        for k, c in enumerate(self.spec.input_chain_index):
            l = c + 1
            if self.jtype[c] == "R":
                J[:, :3, k] = np.cross(f["z"][l], pt - f["p"][l])
                J[:, 3:, k] = f["z"][l]
            elif self.jtype[c] == "P":
                J[:, :3, k] = f["z"][l]
        return J

    def twist(self, q, dq):
        f = self._forward(q, dq)
        return np.stack([np.concatenate([f["vl"][l], f["va"][l]], 1) for l in range(self.L)], 1)

    def dtwist(self, q, dq, ddq):
        f = self._forward(q, dq, ddq)
        return np.stack([np.concatenate([f["al"][l], f["aa"][l]], 1) for l in range(self.L)], 1)

    def joint_torque(self, q, dq, ddq, wrenches=False):
        """Textbook Newton-Euler about each centre of mass, base frame.  wrenches=True also returns the force / moment
        transmitted through every link ((N, L, 6): [force; moment about the link origin], base-frame coordinates) -- what
        Chain::getWrench (primitives_impl.h:1225-1262) returns without external loads; the base link carries the total."""
        f = self._forward(q, dq, ddq)
        N = f["N"]
        F = np.zeros((N, 3))   # force transmitted through joint l (on link l from its parent), base frame
        Mo = np.zeros((N, 3))  # moment about origin of link l
        tau_chain = np.zeros((N, self.nJ))
        W = np.zeros((N, self.L, 6))
        pn = None
        for l in range(self.L - 1, 0, -1):
            Rl = f["R"][l]
            w, dw = f["va"][l], f["aa"][l]
            # classical acceleration of the link origin = spatial lin + w x v_origin
            a_o = f["al"][l] + np.cross(w, f["vl"][l])
            rc = np.einsum("nij,j->ni", Rl, self.com[l])
            a_c = a_o + np.cross(dw, rc) + np.cross(w, np.cross(w, rc))
            Ib = Rl @ self.Icom[l] @ np.transpose(Rl, (0, 2, 1))
            f_net = self.mass[l] * (a_c - self.g)
            n_net = np.einsum("nij,nj->ni", Ib, dw) + np.cross(w, np.einsum("nij,nj->ni", Ib, w))
            if pn is not None:
                # child joint force/moment (about child origin) carried to this link's origin
                Mo = Mo + np.cross(pn - f["p"][l], F)
            F = F + f_net
            Mo = Mo + n_net + np.cross(rc, f_net)
            pn = f["p"][l]
            W[:, l, :3], W[:, l, 3:] = F, Mo
            j = l - 1
            if self.jtype[j] == "R":
                tau_chain[:, j] = np.einsum("ni,ni->n", Mo, f["z"][l])
            elif self.jtype[j] == "P":
                tau_chain[:, j] = np.einsum("ni,ni->n", F, f["z"][l])
        if wrenches:
            W[:, 0, :3] = F                                           # base link: the same force, moment about ITS origin
            W[:, 0, 3:] = Mo + (np.cross(pn - f["p"][0], F) if pn is not None else 0.0)
            return tau_chain[:, self.spec.input_chain_index], W
        return tau_chain[:, self.spec.input_chain_index]

    def regressor(self, q, dq, ddq):
        f = self._forward(q, dq, ddq)
        N = f["N"]
        Yext = np.zeros((N, self.nJ, self.P))
        W = {}  # running, already shifted blocks (N,6,10) in base frame, reference point = current link origin
        for l in range(self.L - 1, 0, -1):
            Rl = f["R"][l]
            Rt = np.transpose(Rl, (0, 2, 1))
            w = np.einsum("nij,nj->ni", Rt, f["va"][l])
            v = np.einsum("nij,nj->ni", Rt, f["vl"][l])
            al = np.einsum("nij,nj->ni", Rt, f["aa"][l])
            a = np.einsum("nij,nj->ni", Rt, f["al"][l])
            d = a + np.cross(w, v) - np.einsum("nij,j->ni", Rt, self.g)
            Wl = np.zeros((N, 6, 10))
            Wl[:, :3, 0] = d
            wx = _skew_b(w)
            Wl[:, :3, 1:4] = _skew_b(al) + wx @ wx
            Wl[:, 3:, 1:4] = -_skew_b(d)
            Wl[:, 3:, 4:] = _L(al) + wx @ _L(w)
            Wb = np.concatenate([Rl @ Wl[:, :3], Rl @ Wl[:, 3:]], 1)
            # shift the following links' blocks from origin l+1 to origin l: ang += lin x (p_l - p_{l+1})
            if l + 1 < self.L:
                dd = f["p"][l] - f["p"][l + 1]
                for k in W:
                    W[k][:, 3:] += np.cross(W[k][:, :3], dd[:, :, None], axisa=1, axisb=1, axisc=1)
            W[l] = Wb
            j = l - 1
            for k in W:
                if self.jtype[j] == "R":
                    Yext[:, j, 10 * (k - 1):10 * k] = np.einsum("ni,nip->np", f["z"][l], W[k][:, 3:])
                elif self.jtype[j] == "P":
                    Yext[:, j, 10 * (k - 1):10 * k] = np.einsum("ni,nip->np", f["z"][l], W[k][:, :3])
        return Yext[:, self.spec.input_chain_index, :]

    def joint_inertia(self, q):
        q = np.atleast_2d(np.asarray(q, dtype=float))
        zero = np.zeros_like(q)
        t0 = self.joint_torque(q, zero, zero)
        M = np.zeros((q.shape[0], self.n, self.n))
        for k in range(self.n):
            e = zero.copy()
            e[:, k] = 1.0
            M[:, :, k] = self.joint_torque(q, zero, e) - t0
        return M

    def nominal_parameters(self):
        pi = np.zeros(self.P)
        for l in range(1, self.L):
            I = self.Iorigin[l]
            pi[10 * (l - 1):10 * l] = [self.mass[l], *(self.mass[l] * self.com[l]),
                                       I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]]
        return pi

    def potential_energy(self, q):
        f = self._forward(q)
        U = np.zeros(f["N"])
        for l in range(1, self.L):
            pc = f["p"][l] + np.einsum("nij,j->ni", f["R"][l], self.com[l])
            U -= self.mass[l] * (pc @ self.g)
        return U
