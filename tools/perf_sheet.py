#!/usr/bin/env python3
"""One MI355X, one table: every batched entry point at N = 1e6 (UR10-like base_link -> wrist_3_link, n = 6, P = 60,
element-major unless stated), time per call, calls/s per sample and the algorithmic HBM bytes each one moves."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain                 # noqa: E402
from rosdyn_amd.components import ComponentSet  # noqa: E402
from tools.probe import timeit               # noqa: E402

N, n, L, P = 1000000, 6, 7, 60
chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
q, dq, ddq, dddq = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
qs, dqs, ddqs = (t.T.contiguous() for t in (q, dq, ddq))
ext = torch.rand((L, 6, N), dtype=torch.float64, device="cuda")
tau = chain.getJointTorque(q, dq, ddq, layout="element")
rows = []
comps = ComponentSet([dict(type=0, joint=j, min_velocity=1e-3, max_velocity=10.0, parameters=[0.1, 0.2]) for j in range(n)], n)


def row(name, fn, bytes_per_eval, reps=10):
    t = timeit(fn, reps=reps, warm=2)
    rows.append((name, t * 1e6, N / t, bytes_per_eval, bytes_per_eval * N / t / 1e9))


E = "element"
row("getRegressor + tau (element-major Y)", lambda: chain.getRegressor(q, dq, ddq, layout=E, with_torque=True), 3072)
row("getRegressor + tau (stacked column-major Y)", lambda: chain.getRegressor(qs, dqs, ddqs, y_layout="stacked", with_torque=True), 3072)
row("getRegressor + tau (per-sample Eigen image)", lambda: chain.getRegressor(qs, dqs, ddqs, with_torque=True), 3072)
row("getJointTorque", lambda: chain.getJointTorque(q, dq, ddq, layout=E), 192)
row("getJointTorque with external wrenches", lambda: chain.getJointTorqueExt(q, dq, ddq, ext, layout=E), 192 + 48 * L)
row("getJointTorqueNonLinearPart", lambda: chain.getJointTorqueNonLinearPart(q, dq, layout=E), 144)
row("getJointInertia", lambda: chain.getJointInertia(q, layout=E), 48 + 288)
row("getWrench (all links, external wrenches)", lambda: chain.getWrench(q, dq, ddq, ext, layout=E), 144 + 96 * L)
row("getTransformation (tool)", lambda: chain.getTransformation(q, layout=E), 48 + 96)
row("getTransformations (all links)", lambda: chain.getTransformations(q, layout=E), 48 + 96 * L)
row("getJacobian", lambda: chain.getJacobian(q, layout=E), 48 + 288)
row("getTwist (all links)", lambda: chain.getTwist(q, dq, layout=E), 96 + 48 * L)
row("getDTwist (all links)", lambda: chain.getDTwist(q, dq, ddq, layout=E), 144 + 48 * L)
row("getDDTwist (all links)", lambda: chain.getDDTwist(q, dq, ddq, dddq, layout=E), 192 + 48 * L)
row("regressor -> Gram [A|tau]'[A|tau] fused", lambda: chain.getRegressorGram(q, dq, ddq, tau, layout=E), 192)
row("identification Gram [Y | 6 friction comps | tau] one call", lambda: chain.getIdentificationGram(comps, q, dq, ddq, tau, layout=E), 192)
row("regressor -> TSQR R factor of [A | tau] (no A'A)", lambda: chain.getRegressorTsqr(q, dq, ddq, tau, layout=E), 192, reps=5)
row("identification TSQR [Y | 6 friction comps | tau]", lambda: chain.getIdentificationTsqr(comps, q, dq, ddq, tau, layout=E), 192, reps=5)
row("friction components (6 x first order, dense n x K)", lambda: comps.getRegressor(q, dq, layout=E), 96 + 8 * n * comps.columns)
T = chain.getTransformation(q, layout=E)
seeds = q + 0.25 * (torch.rand_like(q) * 2 - 1)
row("computeLocalIk, <= 8 updates, seeds within 0.25 rad", lambda: chain.computeLocalIk(T, seeds, toll=1e-6, max_iterations=8, layout=E), 96 + 48 + 48 + 8, reps=5)

# round 3: the reference's own benchmark chain in its public URDF form (fixed head joint, fixed flange + tool0: 9 joints, 6 inputs, P = 90)
pub = Chain(os.path.join(ROOT, "tests/fixtures/ur10_public.urdf"), "base_link", "tool0", (0, 0, -9.806))
Bp = 3 * 48 + 48 + 6 * 90 * 8
row("ur10_public base_link->tool0: getRegressor + tau (stacked)", lambda: pub.getRegressor(qs, dqs, ddqs, y_layout="stacked", with_torque=True), Bp)
row("ur10_public base_link->tool0: getRegressor + tau (images)", lambda: pub.getRegressor(qs, dqs, ddqs, with_torque=True), Bp)
row("ur10_public base_link->tool0: regressor -> Gram (P = 90)", lambda: pub.getRegressorGram(q, dq, ddq, tau, layout=E), 192)
row("ur10_public base_link->tool0: R factor of [A | tau]", lambda: pub.getRegressorTsqr(q, dq, ddq, tau, layout=E), 192, reps=5)

# round 4: the identification step's R factor on the reference's own chains (friction columns beside getRegressor), and rdyn_tsqr on
# materialised matrices beyond 64 columns
from rosdyn_amd.gram import tsqr            # noqa: E402
comps_pub = ComponentSet([dict(type=j % 3, joint=j, min_velocity=1e-3, max_velocity=10.0, parameters=[0.1, 0.2, 0.01][:3 if j % 3 == 1 else 2]) for j in range(6)], 6)
row("ur10_public base_link->tool0: identification R factor, 6 mixed comps", lambda: pub.getIdentificationTsqr(comps_pub, q, dq, ddq, tau, layout=E), 192, reps=5)
row("ur10_public base_link->tool0: identification Gram, 6 mixed comps", lambda: pub.getIdentificationGram(comps_pub, q, dq, ddq, tau, layout=E), 192, reps=5)
n7 = 7
q7, dq7, ddq7 = (torch.rand((n7, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
comps7 = ComponentSet([dict(type=0, joint=j, min_velocity=1e-3, max_velocity=10.0, parameters=[0.1, 0.2]) for j in range(n7)], n7)
for tool in ("link7", "hand"):
    pa = Chain(os.path.join(ROOT, "tests/fixtures/panda_like.urdf"), "link0", tool, (0, 0, -9.806))
    tau7 = pa.getJointTorque(q7, dq7, ddq7, layout=E)
    row("panda link0->%s: regressor -> Gram" % tool, lambda: pa.getRegressorGram(q7, dq7, ddq7, tau7, layout=E), 224, reps=5)
    row("panda link0->%s: R factor of [A | tau]" % tool, lambda: pa.getRegressorTsqr(q7, dq7, ddq7, tau7, layout=E), 224, reps=5)
    row("panda link0->%s: identification Gram, 7 friction comps" % tool, lambda: pa.getIdentificationGram(comps7, q7, dq7, ddq7, tau7, layout=E), 224, reps=5)
    row("panda link0->%s: identification R factor, 7 friction comps" % tool, lambda: pa.getIdentificationTsqr(comps7, q7, dq7, ddq7, tau7, layout=E), 224, reps=5)
    if tool == "link7":
        small = [t[:, :2000].contiguous() for t in (q7, dq7, ddq7, tau7)]
        t = timeit(lambda: pa.getIdentificationTsqr(comps7, *small, layout=E), reps=5, warm=2)
        rows.append(("panda link0->link7: identification R factor, N = 2 000 (LDS-resident folds)", t * 1e6, 2000 / t, 224, 224 * 2000 / t / 1e9))
for rws, cols in ((6000000, 60), (6000000, 85), (1000000, 111)):
    Am = torch.rand((cols, rws), dtype=torch.float64, device="cuda")
    bm = torch.rand((rws,), dtype=torch.float64, device="cuda")
    t = timeit(lambda: tsqr(Am, bm), reps=3, warm=1)
    rows.append(("rdyn_tsqr: %d rows x (%d + 1) columns from memory" % (rws, cols), t * 1e6, rws / t, 8 * (cols + 1), 8 * (cols + 1) * rws / t / 1e9))
    del Am, bm

# round 4: what the chains OFF the fast kernels cost -- a fixed frame in the MIDDLE of the chain, permuted input joints, and a chain
# longer than the kernels sweep (14 joints: the reduced companion is swept, the folded links' columns restored by 10 x 10 blocks)
longc = Chain(os.path.join(ROOT, "tests/fixtures/ur10_public_long.urdf"), "base_link", "tcp", (0, 0, -9.806))
Bl = 3 * 48 + 48 + 6 * 140 * 8
row("ur10_public_long base_link->tcp (14 joints, P = 140): getRegressor + tau (element-major)", lambda: longc.getRegressor(q, dq, ddq, layout=E, with_torque=True), Bl, reps=5)
row("ur10_public_long base_link->tcp: getRegressor + tau (per-sample images)", lambda: longc.getRegressor(qs, dqs, ddqs, with_torque=True), Bl, reps=5)
row("ur10_public_long base_link->tcp: getRegressor + tau (stacked)", lambda: longc.getRegressor(qs, dqs, ddqs, y_layout="stacked", with_torque=True), Bl, reps=5)
row("ur10_public_long base_link->tcp: getJointTorque", lambda: longc.getJointTorque(q, dq, ddq, layout=E), 192)
row("ur10_public_long base_link->tcp: regressor -> Gram (P = 140)", lambda: longc.getRegressorGram(q, dq, ddq, tau, layout=E), 192, reps=5)
row("ur10_public_long base_link->tcp: R factor of [A | tau] (141 x 141)", lambda: longc.getRegressorTsqr(q, dq, ddq, tau, layout=E), 192, reps=5)
midc = Chain(os.path.join(ROOT, "tests/fixtures/ur10_public_long.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))   # 8 joints, fixed head + fixed MIDDLE
Bm = 3 * 48 + 48 + 6 * 80 * 8
row("fixed frame mid-chain (8 joints, P = 80): getRegressor + tau (stacked)", lambda: midc.getRegressor(qs, dqs, ddqs, y_layout="stacked", with_torque=True), Bm, reps=5)
row("fixed frame mid-chain: getRegressor + tau (per-sample images)", lambda: midc.getRegressor(qs, dqs, ddqs, with_torque=True), Bm, reps=5)
row("fixed frame mid-chain: regressor -> Gram", lambda: midc.getRegressorGram(q, dq, ddq, tau, layout=E), 192, reps=5)
row("fixed frame mid-chain: R factor of [A | tau]", lambda: midc.getRegressorTsqr(q, dq, ddq, tau, layout=E), 192, reps=5)
perm = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
perm.setInputJointsName(list(reversed(perm.getActiveJointsName())))
row("permuted input joints (6 joints): getRegressor + tau (stacked)", lambda: perm.getRegressor(qs, dqs, ddqs, y_layout="stacked", with_torque=True), 3072, reps=5)
row("permuted input joints: getRegressor + tau (per-sample images)", lambda: perm.getRegressor(qs, dqs, ddqs, with_torque=True), 3072, reps=5)
row("permuted input joints: getRegressor + tau (element-major)", lambda: perm.getRegressor(q, dq, ddq, layout=E, with_torque=True), 3072, reps=5)
row("permuted input joints: regressor -> Gram", lambda: perm.getRegressorGram(q, dq, ddq, tau, layout=E), 192, reps=5)
row("permuted input joints: R factor of [A | tau]", lambda: perm.getRegressorTsqr(q, dq, ddq, tau, layout=E), 192, reps=5)

# round 5: the by-link kinematic outputs of a chain longer than the unrolled kernels sweep (run-time-length kernels, rdyn_long_kin.hip):
# 14 joints / 15 links, 6 input joints
Ll = 15
extl = torch.rand((Ll, 6, N), dtype=torch.float64, device="cuda")
row("ur10_public_long (15 links): getTransformation (tool)", lambda: longc.getTransformation(q, layout=E), 48 + 96, reps=5)
row("ur10_public_long: getTransformations (all links)", lambda: longc.getTransformations(q, layout=E), 48 + 96 * Ll, reps=5)
row("ur10_public_long: getJacobian", lambda: longc.getJacobian(q, layout=E), 48 + 288, reps=5)
row("ur10_public_long: getTwist (all links)", lambda: longc.getTwist(q, dq, layout=E), 96 + 48 * Ll, reps=5)
row("ur10_public_long: getDTwist (all links)", lambda: longc.getDTwist(q, dq, ddq, layout=E), 144 + 48 * Ll, reps=5)
row("ur10_public_long: getDDTwist (all links)", lambda: longc.getDDTwist(q, dq, ddq, dddq, layout=E), 192 + 48 * Ll, reps=5)
row("ur10_public_long: getWrench (all links, external wrenches)", lambda: longc.getWrench(q, dq, ddq, extl, layout=E), 144 + 96 * Ll, reps=5)
row("ur10_public_long: getJointTorque with external wrenches", lambda: longc.getJointTorqueExt(q, dq, ddq, extl, layout=E), 192 + 48 * Ll, reps=5)
Tl = longc.getTransformation(q, layout=E)
row("ur10_public_long: computeLocalIk, <= 8 updates", lambda: longc.computeLocalIk(Tl, seeds, toll=1e-6, max_iterations=8, layout=E), 200, reps=3)
del extl

print("%-60s %10s %14s %10s %10s" % ("entry point (N = 1e6 per call)", "us / call", "evals/s", "B / eval", "GB/s"))
for r in rows:
    print("%-60s %10.1f %14.3e %10d %10.0f" % r)

# round 5: small batches -- what a call costs when the batch does not fill the chip (device-resident inputs, time per CALL; the C++
# facade's single-sample getters add the host round trip: rdyn_speed_test prints those beside the reference's README figures)
print()
print("%-60s %12s %12s %12s" % ("small batches: us per call", "N = 1", "N = 64", "N = 4096"))
small = []
for name, fn in (("getRegressor + tau (per-sample Eigen image)", lambda a, b, c, t: chain.getRegressor(a, b, c, with_torque=True)),
                 ("getJointTorque", lambda a, b, c, t: chain.getJointTorque(a, b, c)),
                 ("getJointInertia", lambda a, b, c, t: chain.getJointInertia(a)),
                 ("getTransformations (all links)", lambda a, b, c, t: chain.getTransformations(a)),
                 ("getJacobian", lambda a, b, c, t: chain.getJacobian(a)),
                 ("getDTwist (all links)", lambda a, b, c, t: chain.getDTwist(a, b, c)),
                 ("regressor -> Gram [A|tau]'[A|tau]", lambda a, b, c, t: chain.getRegressorGram(a, b, c, t)),
                 ("regressor -> R factor of [A | tau]", lambda a, b, c, t: chain.getRegressorTsqr(a, b, c, t))):
    cells = []
    for Ns in (1, 64, 4096):
        args = [x[:Ns].contiguous() for x in (qs, dqs, ddqs)] + [tau.T[:Ns].contiguous()]
        cells.append(timeit(lambda: fn(*args), reps=200, warm=10) * 1e6)
    print("%-60s %12.1f %12.1f %12.1f" % ((name,) + tuple(cells)))
