#!/usr/bin/env python3
"""The sweep kernels behind the reference's own timed calls (rosdyn_speed_test.cpp:109-192: pose, jacobian, twists, the split
acceleration twists, acceleration twists, jerk twists, joint torque, joint inertia) plus getRegressor / getTransformations / getWrench,
at N = 1e6 on one MI355X, in BOTH layouts of include/rdyn.h:

  sample   RDYN_LAYOUT_SAMPLE_MAJOR -- a sample's record is contiguous: the memory image of the Eigen objects a rosdyn::Chain caller
           holds (the drop-in layout; primitives_impl.h:884-912, 981-1013)
  element  RDYN_LAYOUT_ELEMENT_MAJOR -- element e of all samples contiguous

For every call: time per launch (torch events on the current stream = the stream the C-ABI launches on), the algorithmic bytes of one
evaluation (inputs read once + the record written once), the roofline that bounds it -- "hbm" for all of them but the joint torque
(and its non-linear part), whose 192 B per evaluation make it fp64-issue bound (~1 500 fp64 instructions per sample)
-- and the fraction of that roofline.  measure_sweeps() is imported by bench.py (extras.sweeps); run as a script it prints the table.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12
# fp64 issue: one wave64 fp64 instruction per SIMD every 4.76 cycles at 2.4 GHz (tools/fp64_issue.hip, profiles/r4), 1 024 SIMDs
FP64_INSTR_PER_S = 1024 * 2.4e9 / 4.76 * 64


def measure_sweeps(N=1000000, reps=10, warm=3, fixture="ur10_like.urdf", base="base_link", tool="wrist_3_link"):
    import torch
    from rosdyn_amd import Chain

    chain = Chain(os.path.join(ROOT, "tests", "fixtures", fixture), base, tool, (0.0, 0.0, -9.806))
    n, L, P = chain.getActiveJointsNumber(), chain.getLinksNumber(), 10 * chain.getJointsNumber()
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0002)

    def timeit(fn):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    out = {}
    for lay in ("sample", "element"):
        shape = (N, n) if lay == "sample" else (n, N)
        q, dq, ddq, dddq = (torch.rand(shape, dtype=torch.float64, device=dev, generator=g) * 2 - 1 for _ in range(4))
        ext = torch.rand((N, L, 6) if lay == "sample" else (L, 6, N), dtype=torch.float64, device=dev, generator=g)

        def buf(*rec):
            return torch.empty(((N,) + rec) if lay == "sample" else (rec + (N,)), dtype=torch.float64, device=dev)
        T1, TL, J, TW, TAU, M, W = buf(4, 3), buf(L, 4, 3), buf(n, 6), buf(L, 6), buf(n), buf(n, n), buf(L, 6)
        Y = torch.empty((N, P, n) if lay == "sample" else (P, n, N), dtype=torch.float64, device=dev)
        # name -> (call, bytes per evaluation, fp64 instructions per sample when the call is issue bound)
        calls = [
            ("getTransformation", lambda: chain.getTransformation(q, layout=lay, out=T1), 8 * n + 96, 0),
            ("getTransformations", lambda: chain.getTransformations(q, layout=lay, out=TL), 8 * n + 96 * L, 0),
            ("getJacobian", lambda: chain.getJacobian(q, layout=lay, out=J), 8 * n + 48 * n, 0),
            ("getTwist", lambda: chain.getTwist(q, dq, layout=lay, out=TW), 16 * n + 48 * L, 0),
            ("getDTwistLinearPart", lambda: chain.getDTwistLinearPart(q, ddq, layout=lay), 16 * n + 48 * L, 0),
            ("getDTwistNonLinearPart", lambda: chain.getDTwistNonLinearPart(q, dq, layout=lay), 16 * n + 48 * L, 0),
            ("getDTwist", lambda: chain.getDTwist(q, dq, ddq, layout=lay, out=TW), 24 * n + 48 * L, 0),
            ("getDDTwist", lambda: chain.getDDTwist(q, dq, ddq, dddq, layout=lay), 32 * n + 48 * L, 0),
            ("getJointTorque", lambda: chain.getJointTorque(q, dq, ddq, layout=lay, out=TAU), 32 * n, 1500),
            ("getJointTorqueNonLinearPart", lambda: chain.getJointTorqueNonLinearPart(q, dq, layout=lay, out=TAU), 24 * n, 1400),
            ("getJointInertia", lambda: chain.getJointInertia(q, layout=lay, out=M), 8 * n + 8 * n * n, 0),
            ("getWrench", lambda: chain.getWrench(q, dq, ddq, ext, layout=lay, out=W), 24 * n + 96 * L, 0),
            ("getRegressor", lambda: chain.getRegressor(q, dq, ddq, layout=lay, out=Y, tau_out=TAU, with_torque=True), 32 * n + 8 * n * P, 0),
        ]
        rows = {}
        for name, fn, nbytes, instr in calls:
            t = timeit(fn)
            hbm = nbytes * N / t / HBM_PEAK
            if instr:
                issue = instr * N / t / FP64_INSTR_PER_S
                bound, frac = ("fp64-issue", issue) if issue > hbm else ("hbm", hbm)
            else:
                bound, frac = "hbm", hbm
            rows[name] = {"ms": round(t * 1e3, 4), "bytes_per_eval": nbytes, "GBps": round(nbytes * N / t / 1e9, 1), "bound": bound, "frac": round(frac, 3)}
        out[lay] = rows
        del q, dq, ddq, dddq, ext, T1, TL, J, TW, TAU, M, W, Y
    return {"n_samples": N, "chain": "%s %s->%s" % (fixture, base, tool), "layouts": out}


def main():
    r = measure_sweeps()
    print("sweep kernels, N = %d, %s (us per launch | GB/s algorithmic | bound, fraction)" % (r["n_samples"], r["chain"]))
    print("%-30s %34s   %34s   %s" % ("call", "sample-major (drop-in)", "element-major", "sample / element"))
    for name in r["layouts"]["sample"]:
        s, e = r["layouts"]["sample"][name], r["layouts"]["element"][name]
        print("%-30s %9.1f %7.0f %-10s %5.3f   %9.1f %7.0f %-10s %5.3f   %6.2f" % (
            name, s["ms"] * 1e3, s["GBps"], s["bound"], s["frac"], e["ms"] * 1e3, e["GBps"], e["bound"], e["frac"], s["ms"] / e["ms"]))


if __name__ == "__main__":
    main()
