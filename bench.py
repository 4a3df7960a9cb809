#!/usr/bin/env python3
"""bench.py -- RNEA joint torque + inertial regressor evaluations/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (rdyn_regressor: fused getJointTorque + getRegressor, one HIP kernel
launch) over one batch of synthetic (q, Dq, DDq) samples that are already resident in HBM.

Workload (BASELINE.json configs[1]): 6-DOF chain (tests/fixtures/ur10_like.urdf cut at wrist_3_link:
n = 6 active joints, 6 chain joints, P = 60 parameters), 1e6 samples per GPU, fp64, dense Y.
Default layouts = SURVEY section 8(d) config 2 as written: inputs AoS [N][6] (sample-major), tau [N][6], Y = the stacked
column-major (6 N) x 60 regressor A (k_rowpair_sweep<6>); --y-layout element selects the SoA form (k_local_sweep<6, REGRESSOR>,
~2 % slower on the same box).
Multi-GPU: the batch shards trivially (i.i.d. samples) -> every rank evaluates its own 1e6 samples, no
data-path collective ("weak" scaling); the only exchange is the max-over-ranks of the elapsed time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--samples S] [--y-layout element|stacked|per_sample]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
GRAVITY = (0.0, 0.0, -9.806)   # rosdyn_speed_test.cpp:61-62


def shard_sizes(total, world):
    """Contiguous split of `total` samples over `world` ranks."""
    base, rem = divmod(total, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def max_over_ranks(value, dist, device):
    """MAX all-reduce of a python float (the bench contract: time = slowest rank)."""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def usable_cores():
    """Host cores this process may actually use: min(CPU affinity, cgroup CPU quota).  The GPU boxes expose 256
    logical CPUs but cap the container at 16 (cpu.max = 1600000 100000): 256 OpenMP threads there only thrash."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def measured_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_latest.json, written by
    tools/summarize_prof.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command;
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md section HBM).  None if no profile matches."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        e = d.get(kernel_tag)
        return float(e["traffic_bytes_per_launch"]) if e else None
    except (OSError, ValueError, KeyError):
        return None


def algorithmic_bytes_per_eval(n, P):
    """3 n doubles read (q, Dq, DDq) + n written (tau) + n P written (dense Y)  -- SURVEY section 8(d)."""
    return 3 * n * 8 + n * 8 + n * P * 8


def cpu_baseline(urdf, base, tool, n, seconds, chunk=262144):
    """Times the CPU oracle (oracle/rosdyn_oracle.c, a port of the reference algorithm -- the reference itself
    cannot be built without Eigen/ROS) on a bounded sample of the same workload: repeated passes over one
    chunk of seeded samples (outputs overwritten, pages pre-touched) for about `seconds` of wall time on all
    host cores (OpenMP static over samples), plus a single-thread rate on a small slice."""
    import numpy as np
    from oracle.oracle import OracleChain
    from rosdyn_amd.samples import trajectory_batch
    cores = usable_cores()
    ref = OracleChain(urdf, base, tool, GRAVITY)
    q, dq, ddq = trajectory_batch(0x5EED0002, chunk, n)
    bufs = (np.ones((chunk, n)), np.ones((chunk, ref.P, n)))   # np.ones touches every output page
    m = 8192
    b1 = (bufs[0][:m], bufs[1][:m])
    ref.batch_torque_regressor(q[:m], dq[:m], ddq[:m], threads=1, bufs=b1)
    t1 = time.perf_counter()   # single thread first (idle OpenMP workers spin after a parallel region)
    ref.batch_torque_regressor(q[:m], dq[:m], ddq[:m], threads=1, bufs=b1)
    dt1 = time.perf_counter() - t1
    ref.batch_torque_regressor(q, dq, ddq, threads=cores, bufs=bufs)   # spin up the thread pool
    passes, used = 0, cores
    t0 = time.perf_counter()
    while True:
        _, _, used = ref.batch_torque_regressor(q, dq, ddq, threads=cores, bufs=bufs)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": passes * chunk / dt, "unit": "evals/s", "cores": int(used), "kind": "port",
            "sample": "%d passes over %d seeded U[-1,1] samples of the same workload (%.1f s wall), "
                      "getJointTorque + getRegressor per sample, OpenMP static over samples on %d threads; "
                      "single-thread rate %.3e evals/s" % (passes, chunk, dt, used, m / dt1)}


def gram_extras(chain, q, dq, ddq, tau, in_layout, N, n, P, world, dist, dev):
    """Outside the timed region, informational only (BASELINE.json configs[3]: normal equations of every rank's shard
    on the fp64 matrix cores + ONE all-reduce of [G | c | bb | count] = P*P + P + 2 doubles).  Never fails the bench."""
    import torch
    try:
        from rosdyn_amd._lib import lib
        from rosdyn_amd.gram import allreduce_normal_equations
        ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, 0),), dtype=torch.uint8, device=dev)
        acc = chain.getRegressorGram(q, dq, ddq, tau, layout=in_layout, workspace=ws)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            chain.getRegressorGram(q, dq, ddq, tau, layout=in_layout, out=acc, workspace=ws)
        torch.cuda.synchronize()
        t_gram = (time.perf_counter() - t0) / reps
        ex = {"regressor_gram_ms_per_rank": t_gram * 1e3, "regressor_gram_evals_per_s": N * world / max_over_ranks(t_gram, dist, dev),
              "gram_flop_per_eval_dense_syrk": n * P * (P + 1) + 2 * n * P,
              "allreduce_doubles": P * P + P + 2, "allreduce_us": None}
        if dist is not None:
            for _ in range(3):
                allreduce_normal_equations(acc[0], acc[1], acc[2], N, dist)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                allreduce_normal_equations(acc[0], acc[1], acc[2], N, dist)
            torch.cuda.synchronize()
            ex["allreduce_us"] = max_over_ranks((time.perf_counter() - t0) / 20, dist, dev) * 1e6
        return ex
    except Exception as e:   # informational leg: report, do not fail the run
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=1000000, help="samples per GPU")
    ap.add_argument("--y-layout", default="stacked", choices=["element", "stacked", "per_sample"])
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall time of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational Gram/all-reduce leg after the timed region")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rosdyn_amd import Chain

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    urdf = os.path.join(ROOT, "tests", "fixtures", "ur10_like.urdf")
    base, tool = "base_link", "wrist_3_link"
    chain = Chain(urdf, base, tool, GRAVITY)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    N = args.samples
    elem = args.y_layout == "element"
    in_layout = "element" if elem else "sample"

    gen = torch.Generator(device=dev).manual_seed(0x5EED0002 + rank)
    shape = (n, N) if elem else (N, n)
    q, dq, ddq = (torch.rand(shape, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3))
    y_shape = {"element": (P, n, N), "stacked": (P, N * n), "per_sample": (N, P, n)}[args.y_layout]
    Y = torch.empty(y_shape, dtype=torch.float64, device=dev)
    tau = torch.empty(shape, dtype=torch.float64, device=dev)

    def step():
        chain.getRegressor(q, dq, ddq, layout=in_layout, y_layout=args.y_layout, out=Y, tau_out=tau)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()          # same stream the kernels are launched on (torch's current stream is passed to the C-ABI)
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dev_ms = ev0.elapsed_time(ev1)
    wall = max_over_ranks(wall, dist if world > 1 else None, dev)
    dev_ms = max_over_ranks(dev_ms, dist if world > 1 else None, dev)

    total_evals = N * world * args.steps
    value = total_evals / wall
    b_eval = algorithmic_bytes_per_eval(n, P)
    kernel_ms = dev_ms / args.steps                       # one launch per step, back to back on one stream
    achieved = b_eval * N / (kernel_ms * 1e-3) / 1e9      # GB/s, algorithmic bytes per launch / launch duration

    out = {
        "metric": "RNEA+regressor evals/s (6-DOF, batch 1e6)", "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1]: 6-DOF chain (ur10_like base_link->wrist_3_link, n=6, P=60), "
                               "batch %d samples per GPU, fp64 getJointTorque + dense getRegressor, "
                               "inputs %s-major, Y layout %s" % (N, in_layout, args.y_layout),
                   "samples_per_gpu": N, "n_active": n, "n_params": P, "parallelism": "sample-sharded x%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic("regressor_%s_n%d_P%d_N%d" % (args.y_layout, n, P, N)),
                     "kernel": "k_local_sweep<6, REGRESSOR>" if elem else "k_rowpair_sweep<6>", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": b_eval * N},
    }
    if not args.no_extras:
        out["extras"] = gram_extras(chain, q, dq, ddq, tau, in_layout, N, n, P, world, dist if world > 1 else None, dev)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(urdf, base, tool, n, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
