#!/usr/bin/env python3
"""bench.py -- RNEA joint torque + inertial regressor evaluations/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (rdyn_regressor: fused getJointTorque + getRegressor, one HIP kernel
launch) over one batch of synthetic (q, Dq, DDq) samples that are already resident in HBM.

Workload of `value` (BASELINE.json configs[1]): 6-DOF chain (tests/fixtures/ur10_like.urdf cut at wrist_3_link:
n = 6 active joints, 6 chain joints, P = 60 parameters), 1e6 samples per GPU, fp64, dense Y.
Default layouts = SURVEY section 8(d) config 2 as written: inputs AoS [N][6] (sample-major), tau [N][6], Y = the stacked
column-major (6 N) x 60 regressor A (k_image_sweep<6, 6>: one thread per sample, link blocks staged in LDS, whole-line nontemporal
copy-out); --y-layout element selects the SoA form (k_local_sweep<6, REGRESSOR>), --y-layout per_sample the drop-in Eigen image.
Multi-GPU: the batch shards trivially (i.i.d. samples) -> every rank evaluates its own 1e6 samples, no
data-path collective ("weak" scaling); the only exchange is the max-over-ranks of the elapsed time.

`value` is measured into the FIRST output allocation the process gets -- what a rosdyn::Chain caller has.  The speed of the
stacked store pattern depends on the physical backing of the 2.88 GB output; the best of a few probed
allocations is reported beside it as "tuned_output_placement" and is never `value`.

Beside `value` the line carries
  * "config4": BASELINE.json configs[3] -- every rank's regressor -> fp64 Gram of its shard (the regressor never reaches
    HBM) written straight into the packed payload [G | c | bb | count], followed by ONE in-place all-reduce of it
    (P*P + P + 2 doubles, RCCL over xGMI; no packing kernels, no host synchronisation inside a step), timed as its own
    barrier-bracketed region (max over ranks), with the all-reduce latency separately;
  * "config4_library": the same exchange through the library's own multi-device path (rdyn_regressor_gram_multi: one process,
    ncclCommInitAll over the N devices, one stream per device, one grouped ncclAllReduce) -- measured by a child process
    `bench.py --single-process --gpus N` that rank 0 starts after the torch.distributed legs are done and torn down;
  * "config3_sharded" / "config5_sharded" (N > 1 only): configs[2]'s robust R factor with every rank's 4e6-sample shard factored on
    its GPU, ONE all-gather of the (P + 1)^2 factors and the fold of the stack on every rank (the TSQR alternative);
    configs[4]'s 256 chains split 256 / N per rank (no collective) -- both timed like the headline (barrier, max over ranks);
  * "extras" (rank 0's GPU, outside every timed region of the headline): configs[2] (7-DOF, N = 4e6, Gram) and configs[4]
    (256 distinct chains x 4 096 samples) once each with their own roofline blocks;
  * "cpu_baseline": the C oracle on rank 0's host cores (every line, also for N > 1).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--samples S] [--y-layout element|stacked|per_sample]
      --gpus N > 1 without a torch.distributed environment: this process starts
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...` as a CHILD
      (before anything touches the GPU), relays rank 0's JSON line and exits with the child's return code.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (what the driver runs)
  python bench.py --gpus 2 --backend gloo --dry     launcher / process-group / all-reduce plumbing only, no GPU (CPU tests)
  python bench.py --single-process --gpus N         configs[3], configs[2]'s R factor and configs[4] through the library's in-process
                                                    multi-device paths only (one process, one JSON line)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix (= vector) datasheet peak
GRAVITY = (0.0, 0.0, -9.806)     # rosdyn_speed_test.cpp:61-62


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


def ranks_seen(dist, device):
    """SUM all-reduce of a one per rank: how many ranks really took part in the collective."""
    import torch
    t = torch.ones(1, dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


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


def committed_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel from the COMMITTED PMC passes (profiles/pmc_latest.json, written by
    tools/summarize_prof.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command on the
    builder's box; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md section HBM).  Not measured in
    this run (PMC needs rocprofv3 around the process): the line says so in roofline.traffic_source.  None if no profile matches."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        e = d.get(kernel_tag)
        return float(e["traffic_bytes_per_launch"]) if e else None
    except (OSError, ValueError, KeyError):
        return None


def committed_counter(tag, key):
    """A counter-derived figure of profiles/pmc_latest.json (written from separate rocprofv3 --pmc passes on the builder's box)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            e = json.load(f).get(tag)
        return float(e[key]) if e and key in e else None
    except (OSError, ValueError, KeyError):
        return None


# v_mfma_f64_16x16x4_f64 instructions per 16-sample tile: the fused regressor -> Gram kernel (rdyn_duo_gram.hip: descending column order,
# zero bands on 16-column boundaries) and pass B of the preconditioned R factor (rdyn_cholqr.hip: product + Gram of the product)
# (checked against SQ_INSTS_VALU_MFMA_MOPS_F64 / 4 of the committed counter passes: 8.25e6 per 1e6 samples at 6 joints, 4.8e7 per 4e6 at 7)
GRAM_MFMA_PER_TILE = {6: 132, 7: 192}
PASS_B_MFMA_PER_TILE = {7: 384}


def algorithmic_bytes_per_eval(n, P):
    """3 n doubles read (q, Dq, DDq) + n written (tau) + n P written (dense Y)  -- SURVEY section 8(d)."""
    return 3 * n * 8 + n * 8 + n * P * 8


def gram_flop_per_eval(n, P):
    """Dense-syrk convention of SURVEY section 8(d): n P (P + 1) for A'A + 2 n P for A'tau."""
    return n * P * (P + 1) + 2 * n * P


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


def time_region(fn, steps, warmup, world, dist, dev):
    """`warmup` untimed calls, then exactly `steps` calls bracketed by barrier + synchronize on both sides.
    Returns (wall seconds, device milliseconds), each the MAX over ranks."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()          # same stream the kernels are launched on (torch's current stream is passed to the C-ABI)
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    d = dist if world > 1 else None
    return max_over_ranks(wall, d, dev), max_over_ranks(ev0.elapsed_time(ev1), d, dev)


def config4_block(chain, q, dq, ddq, tau, in_layout, N, n, P, world, dist, dev, steps, warmup):
    """BASELINE.json configs[3]: the 6-DOF batch sharded over the ranks (weak: N per rank), every rank's normal equations
    on the fp64 matrix cores without the regressor leaving the chip, written straight into the packed payload, then ONE in-place
    all-reduce of P*P + P + 2 doubles.  Nothing in a step touches the host."""
    import torch
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import allreduce_packed, packed_buffer, unpack_normal_equations
    d = dist if world > 1 else None
    ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, 0),), dtype=torch.uint8, device=dev)
    buf = packed_buffer(P, N, dev)

    def gram_only():
        chain.getRegressorGram(q, dq, ddq, tau, layout=in_layout, packed_out=buf, workspace=ws)

    def step():
        gram_only()
        allreduce_packed(buf, d, count=N if world > 1 else None)   # world 1: nothing to reduce, the count slot stays N

    def allreduce_only():
        allreduce_packed(buf, d)

    wall, _ = time_region(step, steps, warmup, world, dist, dev)
    count = unpack_normal_equations(buf, P)[3]                     # after the timed region: one host read
    _, gram_ms = time_region(gram_only, steps, 1, world, dist, dev)
    ar_wall, _ = time_region(allreduce_only, 20, 3, world, dist, dev)
    f_eval = gram_flop_per_eval(n, P)
    kernel_ms = gram_ms / steps
    tf = f_eval * N / (kernel_ms * 1e-3) / 1e12
    return {"workload": "configs[3]: 6-DOF chain, %d samples per GPU x %d GPUs, getRegressor -> Gram [A'A | A'tau | tau'tau] per rank "
                        "(regressor stays on chip) + one in-place all-reduce of %d doubles" % (N, world, P * P + P + 2),
            "value": N * world * steps / wall, "unit": "evals/s", "ms_per_step": wall / steps * 1e3,
            "regressor_gram_ms_per_rank": kernel_ms, "allreduce_us": ar_wall / 20 * 1e6, "allreduce_doubles": P * P + P + 2,
            "samples_reduced": count, "backend": "torch.distributed nccl(RCCL)" if world > 1 else "none (1 rank)",
            "roofline": {"bound": "fp64-matrix", "achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_MATRIX_PEAK_TFLOPS, "flop_per_eval_dense_syrk": f_eval, "kernel_ms": kernel_ms}}


def config4_library(n_dev, N, steps, warmup):
    """BASELINE.json configs[3] through the library's OWN multi-device path: one process, rdyn_multi_gpu_create (ncclCommInitAll over
    the n_dev devices, one stream and one event per device) and rdyn_regressor_gram_multi per step -- every device the fused
    regressor -> Gram of its shard, ONE grouped ncclAllReduce of P*P + P + 2 doubles in place.  Steps are queued back to back
    (asynchronous C-ABI), the context is synchronised at both ends of the timed region."""
    import torch
    from rosdyn_amd import Chain
    from rosdyn_amd.gram import MultiGpuGram
    chain = Chain(os.path.join(ROOT, "tests", "fixtures", "ur10_like.urdf"), "base_link", "wrist_3_link", GRAVITY)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    shards = []
    for d in range(n_dev):
        dev = torch.device("cuda", d)
        gen = torch.Generator(device=dev).manual_seed(0x5EED0002 + d)
        shards.append(tuple(torch.rand((N, n), dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(4)))
    ctx = MultiGpuGram(list(range(n_dev)))
    acc = ctx.regressor_gram(chain, shards)
    for _ in range(warmup):
        ctx.regressor_gram(chain, shards, acc=acc, sync=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.regressor_gram(chain, shards, acc=acc, sync=False)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    count = float(acc[0][-1].item())
    # every device must hold the same sums
    same = all(bool(torch.equal(acc[0].cpu(), a.cpu())) for a in acc[1:])
    f_eval = gram_flop_per_eval(n, P)
    tf = f_eval * N * n_dev * steps / wall / 1e12
    return {"workload": "configs[3] in the library: 6-DOF chain, %d samples per GPU x %d GPUs in ONE process, rdyn_regressor_gram_multi "
                        "(regressor -> Gram per device + one grouped ncclAllReduce of %d doubles)" % (N, n_dev, P * P + P + 2),
            "value": N * n_dev * steps / wall, "unit": "evals/s", "ms_per_step": wall / steps * 1e3, "n_gpus": n_dev, "steps": steps,
            "samples_reduced": count, "all_devices_agree": same, "backend": "rccl-in-library (ncclCommInitAll, single process)",
            "roofline": {"bound": "fp64-matrix", "achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS * n_dev, "unit": "TFLOP/s",
                         "frac": tf / (FP64_MATRIX_PEAK_TFLOPS * n_dev), "flop_per_eval_dense_syrk": f_eval}}


def process_gone(pid):
    """The process does not exist any more, or is a zombie (its GPU contexts are gone; the launcher has not reaped it yet)."""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[0] == "Z"
    except (OSError, IndexError):
        return True


def wait_for_exit(pids, timeout_s):
    """True once none of the processes `pids` runs any more (they are siblings, not children: /proc is polled)."""
    t_end = time.time() + timeout_s
    while time.time() < t_end:
        if all(process_gone(p) for p in pids):
            return True
        time.sleep(0.05)
    return all(process_gone(p) for p in pids)


def config4_library_child(n_dev, N, steps, timeout_s=400, no_extras=False):
    """Runs config4_library in a CHILD process (`bench.py --single-process`): a fresh process owns all n_dev devices and its own RCCL
    communicator, and a failure or hang there cannot take the bench line with it."""
    cmd = [sys.executable, os.path.abspath(__file__), "--single-process", "--gpus", str(n_dev), "--samples", str(N), "--steps", str(steps)]
    if no_extras:
        cmd.append("--no-extras")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "child timed out after %d s" % timeout_s}
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": "child rc %d: %s" % (p.returncode, p.stderr[-400:])}
    return json.loads(lines[-1])


def config3_sharded(world, rank, dist, dev, N=4000000, steps=4):
    """BASELINE.json configs[2] over the ranks (weak: N per rank): every rank the robust R factor of [A | tau] of ITS shard
    (rdyn_regressor_tsqr: preconditioned CholeskyQR on the matrix cores), ONE all-gather of the (P + 1)^2 factors and the fold of the
    stack on every rank -- the R-factor counterpart of config4's all-reduce."""
    import torch
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import allgather_fold_r_factors
    chain = Chain(os.path.join(ROOT, "tests", "fixtures", "panda_like.urdf"), "link0", "link7", GRAVITY)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    gen = torch.Generator(device=dev).manual_seed(0x5EED0003 + rank)
    q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(4))
    ws = torch.empty((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h),), dtype=torch.uint8, device=dev)
    d = dist if world > 1 else None
    state = {}

    def factor_only():
        state["R"] = chain.getRegressorTsqr(q, dq, ddq, tau, workspace=ws)

    def step():
        factor_only()
        state["Rall"] = allgather_fold_r_factors(state["R"], d, device_fold=True)

    def exchange_only():
        allgather_fold_r_factors(state["R"], d, device_fold=True)

    wall, _ = time_region(step, steps, 1, world, dist, dev)
    _, f_ms = time_region(factor_only, steps, 1, world, dist, dev)
    x_wall, _ = time_region(exchange_only, 10, 2, world, dist, dev)
    Rall = state["Rall"]
    # every rank must hold the same factor: the max over ranks of |R - R of rank 0|
    ref = Rall.clone()
    if world > 1:
        dist.broadcast(ref, src=0)
    agree = max_over_ranks(float((Rall - ref).abs().max().item()), d, dev) == 0.0
    qr_flop = 2.0 * n * N * (P + 1) ** 2
    tf = qr_flop / (f_ms / steps * 1e-3) / 1e12
    return {"workload": "configs[2] sharded: 7-DOF panda_like link0->link7 (n=7, P=70), %d samples per GPU x %d GPUs, R factor of [A | tau] per rank "
                        "+ one all-gather of %d doubles per rank + the fold of the stack on every rank" % (N, world, (P + 1) ** 2),
            "value": N * world * steps / wall, "unit": "evals/s", "ms_per_step": wall / steps * 1e3, "factor_ms_per_rank": f_ms / steps,
            "gather_and_fold_us": x_wall / 10 * 1e6, "all_ranks_agree_bitwise": agree, "scaling": "weak",
            "backend": "torch.distributed nccl(RCCL)" if world > 1 else "none (1 rank)",
            "roofline": {"bound": "fp64-matrix", "achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MATRIX_PEAK_TFLOPS,
                         "flop_dense_householder_per_rank": qr_flop, "kernel_ms": f_ms / steps}}


def config5_sharded(world, rank, dist, dev, steps=10, n_chains=256, S=4096):
    """BASELINE.json configs[4] split over the ranks as SURVEY.md section 8(e) says (256 / N chains per GPU, strong scaling: the 256
    chains are the job): every rank one plan over ITS chains, no collective."""
    import torch
    from rosdyn_amd import Chain
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.urdf_gen import mixed_chain_set
    sizes = shard_sizes(n_chains, world)
    lo = sum(sizes[:rank])
    mine = mixed_chain_set(os.path.join(ROOT, "tests", "fixtures"), n_chains)[lo:lo + sizes[rank]]
    items, nbytes = [], 0
    gen = torch.Generator(device=dev).manual_seed(0x5EED0005 + rank)
    for xml, base, tool in mine:
        c = Chain(xml, base, tool, GRAVITY)
        n, P = c.getActiveJointsNumber(), 10 * c.getJointsNumber()
        q, dq, ddq = (torch.rand((n, S), dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3))
        items.append((c, q, dq, ddq))
        nbytes += S * algorithmic_bytes_per_eval(n, P)
    plan = MultiChainRegressor(items, y_layout="stacked")
    wall, dev_ms = time_region(plan.run, steps, 2, world, dist, dev)
    total_bytes = nbytes
    if world > 1:
        t = torch.tensor([float(nbytes)], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        total_bytes = float(t.item())
    gbps = total_bytes / (wall / steps) / 1e9
    return {"workload": "configs[4] sharded: %d distinct 6-/7-DOF chains x %d samples split %s chains per rank over %d GPUs, stacked matrices, no collective"
                        % (n_chains, S, "/".join(str(x) for x in sorted(set(sizes), reverse=True)), world),
            "value": n_chains * S * steps / wall, "unit": "evals/s", "ms_per_step": wall / steps * 1e3, "kernel_ms_slowest_rank": dev_ms / steps,
            "scaling": "strong", "chains_per_rank": sizes,
            "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS * world, "unit": "GB/s", "frac": gbps / (HBM_PEAK_GBPS * world),
                         "algorithmic_bytes_per_step": total_bytes}}


def library_multi_legs(n_dev, steps=4, N3=4000000, n_chains=256, S=4096):
    """--single-process: the other two multi-device legs through the library alone -- configs[2]'s R factor with
    rdyn_regressor_tsqr_multi (factor per device, ONE grouped ncclAllGather, the fold on every device) and configs[4] with one plan per
    device launched from the one process (no collective)."""
    import torch
    from rosdyn_amd import Chain
    from rosdyn_amd.gram import MultiGpuGram
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.urdf_gen import mixed_chain_set
    out = {}
    chain = Chain(os.path.join(ROOT, "tests", "fixtures", "panda_like.urdf"), "link0", "link7", GRAVITY)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    shards = []
    for d in range(n_dev):
        dev = torch.device("cuda", d)
        gen = torch.Generator(device=dev).manual_seed(0x5EED0003 + d)
        shards.append(tuple(torch.rand((N3, n), dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(4)))
    ctx = MultiGpuGram(list(range(n_dev)))
    R = ctx.identification_tsqr(chain, shards)
    t0 = time.perf_counter()
    for _ in range(steps):
        R = ctx.identification_tsqr(chain, shards, sync=False)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    same = all(bool(torch.equal(R[0].cpu(), r.cpu())) for r in R[1:])
    out["config3_library"] = {"workload": "configs[2] in the library: %d samples per GPU x %d GPUs in ONE process, rdyn_regressor_tsqr_multi (R factor per device + one "
                                          "grouped ncclAllGather of %d doubles + the fold on every device)" % (N3, n_dev, (P + 1) ** 2),
                              "value": N3 * n_dev * steps / wall, "unit": "evals/s", "ms_per_step": wall / steps * 1e3, "all_devices_agree_bitwise": same}
    del shards, R, ctx
    torch.cuda.empty_cache()
    sizes = shard_sizes(n_chains, n_dev)
    all_items = mixed_chain_set(os.path.join(ROOT, "tests", "fixtures"), n_chains)
    plans, lo = [], 0
    for d in range(n_dev):
        dev = torch.device("cuda", d)
        gen = torch.Generator(device=dev).manual_seed(0x5EED0005 + d)
        items = []
        for xml, base, tool in all_items[lo:lo + sizes[d]]:
            c = Chain(xml, base, tool, GRAVITY)
            nn = c.getActiveJointsNumber()
            items.append((c,) + tuple(torch.rand((nn, S), dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)))
        lo += sizes[d]
        with torch.cuda.device(dev):
            plans.append(MultiChainRegressor(items, y_layout="stacked"))

    def run_all():
        for d, pl in enumerate(plans):
            with torch.cuda.device(d):
                pl.run()

    def sync_all():
        for d in range(n_dev):
            torch.cuda.synchronize(d)

    run_all()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps * 3):
        run_all()
    sync_all()
    wall = time.perf_counter() - t0
    out["config5_library"] = {"workload": "configs[4] in ONE process: %d chains x %d samples, one plan per device (%s chains each), no collective"
                                          % (n_chains, S, "/".join(str(x) for x in sorted(set(sizes), reverse=True))),
                              "value": n_chains * S * steps * 3 / wall, "unit": "evals/s", "ms_per_step": wall / (steps * 3) * 1e3}
    return out


def extras_config3(dev, steps=5):
    """BASELINE.json configs[2]: Panda-like 7-DOF chain (link0 -> link7: n = 7, 7 chain joints, P = 70),
    N = 4e6, regressor -> Gram on the fp64 matrix cores.  Once, outside the headline's timed region."""
    import torch
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import lib
    chain = Chain(os.path.join(ROOT, "tests", "fixtures", "panda_like.urdf"), "link0", "link7", GRAVITY)
    n, P, N = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber(), 4000000
    gen = torch.Generator(device=dev).manual_seed(0x5EED0003)
    q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(4))
    ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, 0),), dtype=torch.uint8, device=dev)
    acc = chain.getRegressorGram(q, dq, ddq, tau, workspace=ws)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        chain.getRegressorGram(q, dq, ddq, tau, out=acc, workspace=ws)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / steps
    # the same batch through the tall-skinny QR (the R factor of [A | tau] without forming A'A): the robust route, not the default
    from rosdyn_amd._lib import lib
    ws_qr = torch.empty((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h),), dtype=torch.uint8, device=dev)
    chain.getRegressorTsqr(q, dq, ddq, tau, workspace=ws_qr)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(4):
        chain.getRegressorTsqr(q, dq, ddq, tau, workspace=ws_qr)
    ev1.record()
    torch.cuda.synchronize()
    tsqr_ms = ev0.elapsed_time(ev1) / 4
    report = chain.lastTsqrReport(N, ws_qr)   # which stage of the factorisation vouched for the result (decided on the device)
    f_eval = gram_flop_per_eval(n, P)
    tf = f_eval * N / (ms * 1e-3) / 1e12
    # dense Householder convention for the factor of the (n N) x (P + 1) matrix [A | tau]: 2 rows cols^2 -- a CONVENTION (the
    # preconditioned CholeskyQR that runs does far fewer flops): kept as a labelled side value.  The utilisation figure is the MFMA-issue
    # fraction: the v_mfma_f64_16x16x4_f64 instructions the call issues (pass A over the subsample + pass B over all rows: 384 per
    # 16-sample tile at 7 joints, rdyn_cholqr.hip) x 64 issue cycles each, over the 1 024 SIMDs, against the time of the whole call
    qr_flop = 2.0 * n * N * (P + 1) ** 2
    qr_tf = qr_flop / (tsqr_ms * 1e-3) / 1e12
    mfma_pass_b = (N / 16.0) * PASS_B_MFMA_PER_TILE.get(n, 0)
    mfma_issue_ms = mfma_pass_b * 64.0 / (1024 * 2.4e9) * 1e3
    gram_mfma = (N / 16.0) * GRAM_MFMA_PER_TILE.get(n, 0)
    gram_issue_ms = gram_mfma * 64.0 / (1024 * 2.4e9) * 1e3
    busy = committed_counter("gram_duo_n%d" % n, "mfma_busy")
    busy_b = committed_counter("pgram_n%d" % n, "mfma_busy")
    return {"workload": "configs[2]: 7-DOF panda_like link0->link7 (n=7, P=70), N=%d, getRegressor -> Gram" % N,
            "value": N / (ms * 1e-3), "unit": "evals/s", "ms_per_step": ms, "tsqr_ms_per_step": tsqr_ms,
            "roofline": {"bound": "fp64-matrix", "achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_MATRIX_PEAK_TFLOPS, "flop_per_eval_dense_syrk": f_eval, "kernel_ms": ms,
                         # what the kernel really issues: MFMAs x 64 cycles over 1 024 SIMDs / launch time (zero bands skipped: fewer
                         # flops than the dense-syrk convention credits), and the committed SQ_VALU_MFMA_BUSY_CYCLES fraction
                         "mfma_issue_frac": gram_issue_ms / ms if gram_mfma else None, "mfma_per_launch": gram_mfma or None,
                         "mfma_busy": busy, "mfma_busy_source": "committed counter pass (profiles/pmc_latest.json), not measured in this run" if busy else None},
            "tsqr_roofline": {"bound": "fp64-matrix", "unit": "fraction of the fp64 MFMA issue rate",
                              "frac": mfma_issue_ms / tsqr_ms if mfma_pass_b else None, "mfma_pass_b": mfma_pass_b or None, "ms": tsqr_ms,
                              "mfma_busy_pass_b": busy_b,
                              "dense_householder_convention": {"achieved_tflops": qr_tf, "frac": qr_tf / FP64_MATRIX_PEAK_TFLOPS, "flop": qr_flop,
                                                               "note": "2 rows cols^2 credited to a route that does fewer flops: a convention, not a utilisation"},
                              "route": "rdyn_regressor_tsqr (R factor of [A | tau] without the normal equations)",
                              "report": report}}


def extras_sweeps():
    """tools/sweep_sheet.py: every getter of the reference's harness at N = 1e6 in the sample-major (drop-in) and the element-major layout:
    ms per launch, the roofline that bounds it ("hbm" | "fp64-issue") and the fraction of it."""
    from tools.sweep_sheet import measure_sweeps
    return measure_sweeps()


def extras_real_chains(dev, steps=5, N=1000000):
    """The reference's own benchmark chains in their public URDF form (rosdyn_speed_test.cpp:44-45, test.cpp:47-48): ur10
    base_link -> tool0 with the fixed base_link_inertia joint in front and the fixed flange / tool0 joints behind (9 chain joints,
    6 input joints, P = 90) and a Panda link0 -> hand (9 joints, 7 input joints, P = 90): dense regressor (per-sample drop-in
    image and stacked matrix) and the regressor -> Gram, 1e6 samples each."""
    import torch
    from rosdyn_amd import Chain
    out = {}
    for name, urdf, base, tool in (("ur10_public_base_link_tool0", "ur10_public.urdf", "base_link", "tool0"),
                                   ("panda_link0_hand", "panda_like.urdf", "link0", "hand")):
        chain = Chain(os.path.join(ROOT, "tests", "fixtures", urdf), base, tool, GRAVITY)
        n, nJ = chain.getActiveJointsNumber(), chain.getJointsNumber()
        P = 10 * nJ
        gen = torch.Generator(device=dev).manual_seed(0x5EED0006)
        q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(4))
        tau_o = torch.empty((N, n), dtype=torch.float64, device=dev)
        res = {"n_active": n, "chain_joints": nJ, "n_params": P}
        for lay, shape in (("stacked", (P, N * n)), ("per_sample", (N, P, n))):
            Y = torch.empty(shape, dtype=torch.float64, device=dev)
            chain.getRegressor(q, dq, ddq, y_layout=lay, out=Y, tau_out=tau_o)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(steps):
                chain.getRegressor(q, dq, ddq, y_layout=lay, out=Y, tau_out=tau_o)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / steps
            gbps = algorithmic_bytes_per_eval(n, P) * N / (ms * 1e-3) / 1e9
            res["regressor_" + lay] = {"ms": ms, "evals_per_s": N / (ms * 1e-3), "GBps": gbps, "hbm_frac": gbps / HBM_PEAK_GBPS}
            del Y
        acc = chain.getRegressorGram(q, dq, ddq, tau)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(steps):
            chain.getRegressorGram(q, dq, ddq, tau, out=acc)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / steps
        # flops actually multiplied: the kernels sweep the reduced chain (10 n columns, fixed links folded in)
        f_red = gram_flop_per_eval(n, 10 * n)
        res["regressor_gram"] = {"ms": ms, "evals_per_s": N / (ms * 1e-3), "TFLOPs_reduced_chain_dense_syrk": f_red * N / (ms * 1e-3) / 1e12,
                                 "note": "normal equations of all %d columns = E' G_red E from the %d columns of the reduced chain" % (P, 10 * n)}
        out[name] = res
        torch.cuda.empty_cache()
    return out


def extras_config5(dev, steps=5, n_chains=256, S=4096):
    """BASELINE.json configs[4]: 256 distinct perturbed 6-/7-DOF chains x 4 096 samples, one launch per joint-count group."""
    import torch
    from rosdyn_amd import Chain
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.urdf_gen import mixed_chain_set
    items, nbytes = [], 0
    gen = torch.Generator(device=dev).manual_seed(0x5EED0005)
    for xml, base, tool in mixed_chain_set(os.path.join(ROOT, "tests", "fixtures"), n_chains):
        c = Chain(xml, base, tool, GRAVITY)
        n, P = c.getActiveJointsNumber(), 10 * c.getJointsNumber()
        q, dq, ddq = (torch.rand((n, S), dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3))
        items.append((c, q, dq, ddq))
        nbytes += S * algorithmic_bytes_per_eval(n, P)

    def time_plan(y_layout):
        plan = MultiChainRegressor(items, y_layout=y_layout)
        plan.run()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(steps):
            plan.run()
        ev1.record()
        torch.cuda.synchronize()
        t = ev0.elapsed_time(ev1) / steps
        del plan
        return t

    # two passes over the three layouts, the better one each (the first plan of a process pays for fresh device memory)
    t = {lay: min(time_plan(lay), time_plan(lay)) for lay in ("element", "per_sample", "stacked")}
    ms = t["stacked"]            # the layout of the headline (stacked column-major A), one matrix per chain
    ms_image = t["per_sample"]   # the drop-in layout: one Eigen image per sample
    ms_element = t["element"]
    gbps = nbytes / (ms * 1e-3) / 1e9
    traffic = committed_traffic("config5_stacked") if (n_chains, S) == (256, 4096) else None
    return {"workload": "configs[4]: %d distinct 6-/7-DOF chains x %d samples, getJointTorque + dense getRegressor, "
                        "one stacked column-major (S n) x P matrix per chain, one launch per joint-count group" % (n_chains, S),
            "value": n_chains * S / (ms * 1e-3), "unit": "evals/s", "ms_per_step": ms,
            "ms_per_step_per_sample_images": ms_image, "ms_per_step_element_major": ms_element,
            "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_step": nbytes, "kernel_ms": ms, "traffic": traffic,
                         "traffic_source": "committed profile (profiles/pmc_latest.json, builder's box), not measured in this run" if traffic else None}}


def step_into(chain, q, dq, ddq, in_layout, y_layout, Y, tau):
    def step():
        chain.getRegressor(q, dq, ddq, layout=in_layout, y_layout=y_layout, out=Y, tau_out=tau)
    return step


def guarded(fn, *a, **k):
    """Informational legs never fail the run: they report their error instead."""
    try:
        return fn(*a, **k)
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """--gpus N > 1 without a torch.distributed environment: run the ranks as CHILD processes of this one (never exec:
    a process that has touched the GPU must not be replaced, and this one has not touched it yet), relay rank 0's JSON
    line, return the child's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True)
    line = None
    for out in p.stdout:
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(out)
    rc = p.wait()
    if line is not None:
        print(line)
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
    return rc


def dry_run(args, world, rank):
    """No GPU: the launcher, the process group (gloo) and the ONE collective of the path (all-reduce of the packed normal
    equations) on small CPU tensors.  Used by tests/test_bench_launch.py; `value` is null."""
    import torch
    import torch.distributed as dist
    from rosdyn_amd.gram import allreduce_normal_equations
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=args.backend)
    dev = torch.device("cpu")
    P, N = 60, shard_sizes(args.samples * world, world)[rank]
    g = torch.Generator().manual_seed(1234 + rank)
    A = torch.rand((64, P), dtype=torch.float64, generator=g)
    b = torch.rand((64,), dtype=torch.float64, generator=g)
    G, c, bb, count = allreduce_normal_equations(A.T @ A, A.T @ b, (b @ b).reshape(1), N, dist if world > 1 else None)
    seen = ranks_seen(dist if world > 1 else None, dev)
    # the R-factor exchange of config3_sharded: every rank the factor of ITS rows, one all-gather, the fold of the stack (host fold here)
    import numpy as np
    from rosdyn_amd.gram import allgather_fold_r_factors
    M = torch.cat([A, b[:, None]], dim=1).numpy()
    Rr = torch.from_numpy(np.linalg.qr(M, mode="r"))
    Rall = allgather_fold_r_factors(Rr, dist if world > 1 else None)
    gram_all = torch.from_numpy(M.T @ M)
    if world > 1:
        dist.all_reduce(gram_all)
    fold_err = float((Rall.T @ Rall - gram_all).abs().max() / gram_all.abs().max())
    # config5_sharded's split of the 256 chains, and the process ids rank 0 waits on before it starts the library child
    chains = shard_sizes(256, world)
    mine = torch.tensor([float(chains[rank])], dtype=torch.float64)
    pids = [os.getpid()]
    if world > 1:
        dist.all_reduce(mine)
        pids = [None] * world
        dist.all_gather_object(pids, os.getpid())
    out = {"metric": "RNEA+regressor evals/s (6-DOF, batch 1e6)", "value": None, "unit": "evals/s", "n_gpus": world, "dry": True,
           "backend": args.backend, "n_ranks_seen": seen, "samples_reduced": count, "gram_trace": float(torch.trace(G).item()),
           "steps": args.steps, "warmup": args.warmup,
           "legs": ["config4"] + (["config3_sharded", "config5_sharded"] if world > 1 and not args.no_extras else []),
           "r_factor_gather_fold_rel_err": fold_err, "config5_chains_per_rank": chains, "config5_chains_total": int(mine.item()),
           "distinct_pids": len(set(pids))}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["other_ranks_exited"] = wait_for_exit([p for p in pids if p != os.getpid()], 30.0) if world > 1 else True
        print(json.dumps(out))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=1000000, help="samples per GPU")
    ap.add_argument("--y-layout", default="stacked", choices=["element", "stacked", "per_sample"])
    ap.add_argument("--placements", type=int, default=12, help="candidate output buffers probed for the tuned_output_placement side block AFTER the headline (0 or 1 = skip); never affects `value`")
    ap.add_argument("--single-process", action="store_true", help="configs[3] through the library's in-process RCCL path (rdyn_regressor_gram_multi over --gpus devices) only")
    ap.add_argument("--no-library-config4", action="store_true", help="skip the config4_library child process")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall time of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[2] / configs[4] legs after the timed region")
    ap.add_argument("--no-config4", action="store_true", help="skip the regressor -> Gram -> all-reduce block (configs[3])")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo: --dry only)")
    ap.add_argument("--sharded-legs", action="store_true", help="run config3_sharded / config5_sharded also with one rank (they run by themselves for N > 1)")
    ap.add_argument("--dry", action="store_true", help="no GPU work: launcher + process group + all-reduce plumbing (CPU tests)")
    args = ap.parse_args()

    if args.single_process:
        # one process, every device: the library's own communicator (nothing of torch.distributed is initialised)
        if args.dry:
            print(json.dumps({"metric": "config4_library", "dry": True, "n_gpus": args.gpus, "steps": args.steps, "samples_per_gpu": args.samples,
                              "legs": ["config4_library"] + ([] if args.no_extras else ["config3_library", "config5_library"])}))
            sys.exit(0)
        line = config4_library(args.gpus, args.samples, max(1, args.steps), 3)
        if not args.no_extras:
            line.update(guarded(library_multi_legs, args.gpus))
        print(json.dumps(line))
        sys.exit(0)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))   # before anything touches the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
        sys.exit(2)
    if args.dry:
        sys.exit(dry_run(args, world, rank))
    if args.backend != "nccl":
        sys.stderr.write("bench.py: --backend gloo is for --dry only\n")
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from rosdyn_amd import Chain

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    seen = ranks_seen(dist if world > 1 else None, dev)   # one RCCL all-reduce of ones: who is really here

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
    tau = torch.empty(shape, dtype=torch.float64, device=dev)

    # HEADLINE = the first allocation (what a caller gets).  Where the 2.88 GB output lands in HBM matters for this store pattern
    # (profiles/r2/placement.txt): the best of a few probed allocations is a SIDE block further down.
    Y = torch.empty(y_shape, dtype=torch.float64, device=dev)

    def step():
        chain.getRegressor(q, dq, ddq, layout=in_layout, y_layout=args.y_layout, out=Y, tau_out=tau)

    wall, dev_ms = time_region(step, args.steps, args.warmup, world, dist, dev)

    total_evals = N * world * args.steps
    value = total_evals / wall
    b_eval = algorithmic_bytes_per_eval(n, P)
    kernel_ms = dev_ms / args.steps                       # one launch per step, back to back on one stream
    achieved = b_eval * N / (kernel_ms * 1e-3) / 1e9      # GB/s, algorithmic bytes per launch / launch duration
    # the names rocprofv3 --kernel-trace reports (profiles/r6/bench_default_summary.txt); template arguments of k_image_sweep: NJ, FIX
    # (mask of joints that are not input joints), NT (nontemporal copy-out), STACKED, MAP (run-time row map, -1 = none), EXPAND
    kernel = {"element": "void (anonymous namespace)::k_local_sweep<6, 0>(RdynSweepArgs)",
              "stacked": "void (anonymous namespace)::k_image_sweep<6, 0u, true, true, -1, false>(RdynSweepArgs)",
              "per_sample": "void (anonymous namespace)::k_image_sweep<6, 0u, true, false, -1, false>(RdynSweepArgs)"}[args.y_layout]
    traffic = committed_traffic("regressor_%s_n%d_P%d_N%d" % (args.y_layout, n, P, N))

    out = {
        "metric": "RNEA+regressor evals/s (6-DOF, batch 1e6)", "value": value, "unit": "evals/s",
        "n_gpus": world, "n_ranks_seen": seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1]: 6-DOF chain (ur10_like base_link->wrist_3_link, n=6, P=60), "
                               "batch %d samples per GPU, fp64 getJointTorque + dense getRegressor, "
                               "inputs %s-major, Y layout %s" % (N, in_layout, args.y_layout),
                   "samples_per_gpu": N, "n_active": n, "n_params": P, "parallelism": "sample-sharded x%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": "committed profile (profiles/pmc_latest.json, builder's box), not measured in this run" if traffic else None,
                     "kernel": kernel, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": b_eval * N},
    }
    if args.y_layout == "stacked" and world == 1 and not args.no_extras:
        # side block, never `value`: the same evaluation into the per-sample DROP-IN images (what a rosdyn::Chain caller receives per sample,
        # primitives_impl.h:1350-1354) -- its own first allocation, same step count, same timing protocol
        def drop_in():
            Yp = torch.empty((N, P, n), dtype=torch.float64, device=dev)
            w_p, ms_p = time_region(step_into(chain, q, dq, ddq, in_layout, "per_sample", Yp, tau), args.steps, args.warmup, world, dist, dev)
            k_ms = ms_p / args.steps
            return {"layout": "per-sample column-major n x P images (N, P, n)", "value": N * args.steps / w_p, "unit": "evals/s", "ms_per_step": w_p / args.steps * 1e3,
                    "roofline": {"bound": "hbm", "achieved": b_eval * N / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": b_eval * N / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel_ms": k_ms,
                                 "kernel": "void (anonymous namespace)::k_image_sweep<6, 0u, true, false, -1, false>(RdynSweepArgs)",
                                 "traffic": committed_traffic("regressor_per_sample_n%d_P%d_N%d" % (n, P, N))},
                    "note": "side block: the headline is the stacked matrix of SURVEY.md section 8(d)"}
        out["drop_in_layout"] = guarded(drop_in)
        torch.cuda.empty_cache()
    if args.placements > 1:
        # side block, never `value`: the same launch into the best of a few candidate output allocations (rosdyn_amd/placement.py)
        del Y
        torch.cuda.empty_cache()
        from rosdyn_amd.placement import pick_output_buffer
        Y, placement = pick_output_buffer(lambda Yc: chain.getRegressor(q, dq, ddq, layout=in_layout, y_layout=args.y_layout, out=Yc, tau_out=tau),
                                          y_shape, dev, max_candidates=args.placements)
        t_wall, t_ms = time_region(step_into(chain, q, dq, ddq, in_layout, args.y_layout, Y, tau), args.steps, 1, world, dist, dev)
        t_kernel_ms = t_ms / args.steps
        out["tuned_output_placement"] = {"value": total_evals / t_wall, "ms_per_step": t_wall / args.steps * 1e3,
                                         "frac": b_eval * N / (t_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "probe_ms": placement["probe_ms"],
                                         "candidates": placement["candidates"], "chosen": placement["chosen"],
                                         "note": "same launch, same device, best of the probed output allocations; NOT the headline"}
    def collective_leg(name, fn, *a):
        """A secondary leg that holds collectives: not guarded per rank when world > 1 (a rank that swallowed an exception would leave
        the others waiting inside the collective) -- but the headline measured above is not lost with it: rank 0 prints the line as
        it stands, marked, before the error takes the job down."""
        if world == 1:
            out[name] = guarded(fn, *a)
            return
        try:
            out[name] = fn(*a)
        except Exception as e:   # noqa: BLE001
            if rank == 0:
                out[name] = {"error": repr(e)}
                out["incomplete"] = "leg %s raised; the legs behind it were not run" % name
                print(json.dumps(out))
                sys.stdout.flush()
            raise

    if not args.no_config4:
        # measured torques of this rank's shard: tau of the evaluation just timed (noise-free: exact normal equations)
        collective_leg("config4", config4_block, chain, q, dq, ddq, tau, in_layout, N, n, P, world, dist, dev, max(50, args.steps), 3)
    del Y
    pids = [os.getpid()]
    if (world > 1 or args.sharded_legs) and not args.no_extras:
        del q, dq, ddq
        torch.cuda.empty_cache()
        collective_leg("config3_sharded", config3_sharded, world, rank, dist, dev)
        torch.cuda.empty_cache()
        collective_leg("config5_sharded", config5_sharded, world, rank, dist, dev)
        q = dq = ddq = None
    if world > 1:
        pids = [None] * world
        dist.all_gather_object(pids, os.getpid())
        dist.barrier()
        dist.destroy_process_group()   # everything collective is done: the other ranks leave, rank 0 goes on alone
    if rank != 0:
        return
    del q, dq, ddq, tau
    torch.cuda.empty_cache()
    if not args.no_extras:
        out["extras"] = {"config3": guarded(extras_config3, dev), "config5": guarded(extras_config5, dev),
                         "real_chains": guarded(extras_real_chains, dev),
                         # the sweep kernels behind the reference's own timed calls (rosdyn_speed_test.cpp:109-192), both layouts, N = 1e6
                         "sweeps": guarded(extras_sweeps)}
        torch.cuda.empty_cache()
    if not args.no_config4 and not args.no_library_config4:
        # the child owns ALL devices with its own communicator: it starts only once the other ranks' processes are gone (a process that
        # has exited has torn its GPU context down -- nothing of theirs is still being dismantled on the devices the child opens)
        gone = wait_for_exit([p for p in pids if p != os.getpid()], 60.0)
        out["config4_library"] = config4_library_child(world, N, max(50, args.steps), no_extras=args.no_extras)
        if isinstance(out["config4_library"], dict):
            out["config4_library"]["other_ranks_exited_before_start"] = gone
    if args.cpu_seconds > 0:
        out["cpu_baseline"] = guarded(cpu_baseline, urdf, base, tool, n, args.cpu_seconds)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
