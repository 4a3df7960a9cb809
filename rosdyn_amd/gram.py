"""Normal equations of the stacked regressor: fp64-MFMA Gram reduction (librdyn_hip.so), the one multi-GPU
exchange of the path (a single all-reduce of P*P + P + 2 doubles over RCCL/xGMI), and the small host-side
least-squares solve.  No reference counterpart inside rosdyn_core (the identification step lived in the
external rosdyn_identification, README.md:15); BASELINE.json's north star asks for it.
"""
import ctypes as C

import numpy as np

from ._lib import check, lib


def gram(A, b=None, out=None, accumulate=False, workspace=None):
    """A: (P, R) torch.float64 CUDA tensor = column-major R x P matrix (e.g. getRegressor(..., y_layout="element")
    viewed as (P, n*N)); b: (R,) or None.  Returns (G (P, P), c (P,), bb (1,))."""
    import torch
    assert A.is_cuda and A.dtype == torch.float64 and A.dim() == 2 and A.is_contiguous()
    P, R = A.shape
    if out is None:
        out = (torch.empty((P, P), dtype=torch.float64, device=A.device), torch.empty((P,), dtype=torch.float64, device=A.device),
               torch.empty((1,), dtype=torch.float64, device=A.device))
        assert not accumulate
    G, c, bb = out
    nbytes = lib().rdyn_gram_workspace_bytes(P)
    if workspace is None:
        workspace = torch.empty((nbytes,), dtype=torch.uint8, device=A.device)
    if b is not None:
        assert b.is_cuda and b.dtype == torch.float64 and b.numel() == R and b.is_contiguous()
    check(lib().rdyn_gram(A.data_ptr(), R, R, P, b.data_ptr() if b is not None else None, G.data_ptr(), c.data_ptr(), bb.data_ptr(),
                          1 if accumulate else 0, workspace.data_ptr(), workspace.numel(),
                          A.device.index if A.device.index is not None else -1, torch.cuda.current_stream(A.device).cuda_stream))
    return G, c, bb


def pack_normal_equations(G, c, bb, count):
    """[G | c | bb | count] as ONE flat fp64 buffer: the all-reduce payload."""
    import torch
    P = G.shape[0]
    buf = torch.empty((P * P + P + 2,), dtype=torch.float64, device=G.device)
    buf[:P * P] = G.reshape(-1)
    buf[P * P:P * P + P] = c
    buf[P * P + P] = bb.reshape(-1)[0]
    buf[P * P + P + 1] = float(count)
    return buf


def unpack_normal_equations(buf, P):
    return buf[:P * P].reshape(P, P), buf[P * P:P * P + P], buf[P * P + P:P * P + P + 1], float(buf[P * P + P + 1].item())


def packed_buffer(P, count, device):
    """The all-reduce payload [G (P*P) | c (P) | bb | count] as one zeroed fp64 buffer with the count slot filled ONCE (the shard
    size does not change from step to step); pass it as getRegressorGram(packed_out=...) and reduce it with allreduce_packed."""
    import torch
    buf = torch.zeros((P * P + P + 2,), dtype=torch.float64, device=device)
    buf[P * P + P + 1] = float(count)
    return buf


def allreduce_packed(buf, dist=None, count=None):
    """ONE in-place all-reduce of a packed accumulator (RCCL over xGMI on GPUs, gloo in the CPU tests): no packing kernels, no host
    synchronisation.  The kernels overwrite G, c, bb every step; the count slot holds the SUM after a reduction, so a caller that
    re-uses the buffer passes its shard size again (`count`: one fill kernel, still no synchronisation)."""
    if count is not None:
        buf[-1] = float(count)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def allreduce_normal_equations(G, c, bb, count, dist=None):
    """Sum of every rank's accumulators: one all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests).
    ~29 KB for P = 60: latency-bound, independent of the batch size."""
    buf = pack_normal_equations(G, c, bb, count)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return unpack_normal_equations(buf, G.shape[0])


def allgather_fold_r_factors(R1, dist=None, device_fold=False):
    """The R-factor exchange of SURVEY.md section 8(e) under torch.distributed: ONE all-gather of every rank's (n1, n1) factor (math
    layout; RCCL over xGMI on GPUs, gloo in the CPU tests) and the fold of the stack in rank order -- every rank folds the same stack in
    the same order, so all ranks hold the same bits.  device_fold=True folds on the GPU (rdyn_tsqr over the stacked factors as one
    (world n1) x n1 matrix); default: rdyn_tsqr_combine_host on the host."""
    import torch
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    if world == 1:
        return R1
    R1 = R1.contiguous()
    parts = [torch.empty_like(R1) for _ in range(world)]
    dist.all_gather(parts, R1)
    if device_fold and R1.is_cuda:
        n1 = R1.shape[0]
        stack = torch.cat(parts, dim=0)                     # (world n1, n1) math layout = rows of the stacked factors
        return tsqr(stack.t().contiguous())                 # column-major (world n1) x n1 as the (n1, world n1) tensor tsqr() takes
    out = tsqr_combine_host([p.cpu().numpy() for p in parts])
    return torch.from_numpy(out).to(R1.device)


def r_factor(G, rtol=1e-10):
    """Rank-revealing R factor of the stacked regressor A from its Gram G = A'A (what a tall-skinny QR of A returns, up to
    the signs of the rows): pivoted Cholesky on the host, R'R = G[perm][:, perm] restricted to the numerical rank.
    Returns (R (rank x P, upper trapezoidal in the permuted column order), perm, rank).  Going through the Gram squares
    the condition number (CholeskyQR): fine for cond(A) << 1e8 in fp64, which identification trajectories satisfy after
    the rank truncation; BASELINE.json configs[2] names this factor."""
    Gh = np.array(G.detach().cpu().numpy() if hasattr(G, "detach") else G, dtype=np.float64)
    Gh = 0.5 * (Gh + Gh.T)
    P = Gh.shape[0]
    perm = np.arange(P)
    R = np.zeros((P, P))
    d = np.diag(Gh).copy()
    dmax = d.max() if P else 0.0
    rank = 0
    for k in range(P):
        j = k + int(np.argmax(d[k:]))
        if d[j] <= rtol * dmax:
            break
        if j != k:
            perm[[k, j]] = perm[[j, k]]
            d[[k, j]] = d[[j, k]]
            R[:, [k, j]] = R[:, [j, k]]
        R[k, k] = np.sqrt(d[k])
        row = Gh[perm[k], perm[k + 1:]] - R[:k, k] @ R[:k, k + 1:]
        R[k, k + 1:] = row / R[k, k]
        d[k + 1:] -= R[k, k + 1:] ** 2
        rank += 1
    return R[:rank], perm, rank


def solve_normal_equations_abi(G, c, rtol=1e-10):
    """The same minimum-norm solve through the C-ABI (rdyn_solve_normal_equations: host C++, what a C++ caller uses).
    Returns (x, rank)."""
    Gh = np.ascontiguousarray(np.asarray(G.detach().cpu() if hasattr(G, "detach") else G, dtype=np.float64).T)   # column-major
    ch = np.ascontiguousarray(np.asarray(c.detach().cpu() if hasattr(c, "detach") else c, dtype=np.float64))
    n = ch.shape[0]
    x = np.zeros(n)
    rank = C.c_int(0)
    dp = C.POINTER(C.c_double)
    check(lib().rdyn_solve_normal_equations(Gh.ctypes.data_as(dp), ch.ctypes.data_as(dp), n, float(rtol), x.ctypes.data_as(dp), C.byref(rank)))
    return x, rank.value


def r_factor_abi(G, rtol=1e-10):
    """rdyn_gram_r_factor through the C-ABI: (R (rank x P), perm, rank) like r_factor()."""
    Gh = np.ascontiguousarray(np.asarray(G.detach().cpu() if hasattr(G, "detach") else G, dtype=np.float64).T)
    n = Gh.shape[0]
    R = np.zeros((n, n))          # column-major on the C side == the transpose here
    perm = np.zeros(n, dtype=np.int32)
    rank = C.c_int(0)
    dp = C.POINTER(C.c_double)
    check(lib().rdyn_gram_r_factor(Gh.ctypes.data_as(dp), n, float(rtol), R.ctypes.data_as(dp), perm.ctypes.data_as(C.POINTER(C.c_int32)),
                                   C.byref(rank)))
    return R.T[:rank.value].copy(), perm.astype(np.int64), rank.value


def residual_sum_of_squares(G, c, bb, x):
    """|A x - b|^2 from the normal-equation accumulators alone: bb - 2 c'x + x'G x (no second pass over the batch).
    Cancellation limits it to ~1e-12 |b|^2; that is far below any measurement noise the identification works with."""
    Gh = np.asarray(G.detach().cpu().numpy() if hasattr(G, "detach") else G, dtype=np.float64)
    ch = np.asarray(c.detach().cpu().numpy() if hasattr(c, "detach") else c, dtype=np.float64)
    b2 = float(bb.reshape(-1)[0].item() if hasattr(bb, "detach") else np.asarray(bb).reshape(-1)[0])
    x = np.asarray(x, dtype=np.float64)
    return b2 - 2.0 * float(ch @ x) + float(x @ (Gh @ x))


def solve_base_parameters(G, c, rtol=1e-10):
    """Minimum-norm least-squares solution of G x = c on the host (P <= 100): the stacked regressor is structurally
    rank deficient (unobservable base-link parameters, fixed tail links), so the symmetric eigen-decomposition is
    truncated at rtol * lambda_max.  Returns (x, rank)."""
    Gh = np.asarray(G.detach().cpu() if hasattr(G, "detach") else G, dtype=np.float64)
    ch = np.asarray(c.detach().cpu() if hasattr(c, "detach") else c, dtype=np.float64)
    Gh = 0.5 * (Gh + Gh.T)
    w, V = np.linalg.eigh(Gh)
    keep = w > rtol * w.max()
    x = V[:, keep] @ ((V[:, keep].T @ ch) / w[keep])
    return x, int(keep.sum())


def tsqr(A, b=None, out=None, accumulate=False, workspace=None):
    """rdyn_tsqr: R factor of [A | b] without forming A'A.  A: (P, R) torch.float64 CUDA tensor = column-major R x P matrix (as
    gram()); b: (R,) or None.  Returns R1 as a (n1, n1) tensor in MATH layout (R1[i, j], upper triangular), n1 = P + (b is not None)."""
    import torch
    assert A.is_cuda and A.dtype == torch.float64 and A.dim() == 2 and A.is_contiguous()
    P, rows = A.shape
    n1 = P + (1 if b is not None else 0)
    buf = torch.zeros((n1, n1), dtype=torch.float64, device=A.device) if out is None else out.t().contiguous()
    nbytes = lib().rdyn_tsqr_workspace_bytes(n1)
    if nbytes == 0:
        raise ValueError("rdyn_tsqr: at most 112 columns (right-hand side included)")
    if workspace is None:
        workspace = torch.empty((nbytes,), dtype=torch.uint8, device=A.device)
    check(lib().rdyn_tsqr(A.data_ptr(), rows, rows, P, b.data_ptr() if b is not None else None, buf.data_ptr(), 1 if accumulate else 0,
                          workspace.data_ptr(), workspace.numel(), A.device.index if A.device.index is not None else -1,
                          torch.cuda.current_stream(A.device).cuda_stream))
    if out is not None:     # the C side is column-major: `buf` was a transposed copy of `out`
        out.copy_(buf.t())
        return out
    return buf.t()


def tsqr_last_report(n1, rows, workspace):
    """rdyn_tsqr_rows_last_report: what the last tsqr() call that used `workspace` did (route / stage / gamma / rho); synchronises."""
    import torch
    from ._lib import RdynTsqrReport
    rep = RdynTsqrReport()
    check(lib().rdyn_tsqr_rows_last_report(int(n1), int(rows), workspace.data_ptr(), workspace.device.index or 0,
                                           torch.cuda.current_stream(workspace.device).cuda_stream, C.byref(rep)))
    return dict(route=rep.route, stage=rep.stage, n_deferred=rep.n_deferred, gamma=tuple(rep.gamma), rho=tuple(rep.rho))


def tsqr_combine_host(factors):
    """rdyn_tsqr_combine_host: folds upper-triangular factors (list of (n, n) numpy arrays in math layout) into one."""
    n = factors[0].shape[0]
    stack = np.ascontiguousarray(np.stack([np.asarray(f, dtype=np.float64).T for f in factors]))   # each column-major
    out = np.zeros((n, n))
    dp = C.POINTER(C.c_double)
    check(lib().rdyn_tsqr_combine_host(stack.ctypes.data_as(dp), len(factors), n, out.ctypes.data_as(dp)))
    return out.T.copy()


def solve_r_factor(R1, n_cols, rtol=1e-10):
    """Minimum-norm least-squares solution from R1 = [R d; 0 rho] (math layout, numpy or tensor): rdyn_solve_r_factor."""
    Rh = np.asarray(R1.detach().cpu() if hasattr(R1, "detach") else R1, dtype=np.float64)
    Rf = np.asfortranarray(Rh[:n_cols, :n_cols])
    d = np.ascontiguousarray(Rh[:n_cols, n_cols])
    x = np.zeros(n_cols)
    rank = C.c_int(0)
    dp = C.POINTER(C.c_double)
    check(lib().rdyn_solve_r_factor(Rf.ctypes.data_as(dp), n_cols, n_cols, n_cols, d.ctypes.data_as(dp), float(rtol), x.ctypes.data_as(dp),
                                    C.byref(rank)))
    return x, rank.value


class MultiGpuGram(object):
    """include/rdyn.h: rdyn_multi_gpu_* -- one process, the batch sharded over `devices`, every device the fused regressor -> Gram of
    its shard, ONE ncclAllReduce of [G | c | bb | count] inside the library (RCCL resolved at run time)."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        check(lib().rdyn_multi_gpu_create(arr, len(self.devices), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().rdyn_multi_gpu_destroy(h)
            self._h = None

    def regressor_gram(self, chain, shards, layout="sample", acc=None, sync=True, accumulate=False):
        """shards[i] = (q, Dq, DDq, tau_meas) float64 CUDA tensors on devices[i].  Returns the list of per-device accumulators
        (P*P + P + 2,) (acc= re-uses the caller's): once the context is synchronised every one holds the sums over all shards
        (accumulate=True: added to what acc held -- a batch streamed through the devices in pieces).
        The library orders its streams behind torch's current stream of every device; sync=False returns without waiting
        (call synchronize() before reading)."""
        import torch
        from ._lib import Batch, LAYOUT_ELEMENT_MAJOR, LAYOUT_SAMPLE_MAJOR
        assert len(shards) == len(self.devices)
        P = 10 * chain.getJointsNumber()
        n = chain.getActiveJointsNumber()
        batches = (Batch * len(shards))()
        taus = (C.c_void_p * len(shards))()
        accs = (C.c_void_p * len(shards))()
        out = []
        for i, (q, dq, ddq, tau) in enumerate(shards):
            for t in (q, dq, ddq, tau):
                assert t.is_cuda and t.device.index == self.devices[i] and t.dtype == torch.float64 and t.is_contiguous()
            b = batches[i]
            b.n_samples = q.shape[1] if layout == "element" else q.shape[0]
            assert (q.shape[0] if layout == "element" else q.shape[1]) == n
            b.q, b.dq, b.ddq = q.data_ptr(), dq.data_ptr(), ddq.data_ptr()
            b.layout = LAYOUT_ELEMENT_MAJOR if layout == "element" else LAYOUT_SAMPLE_MAJOR
            b.device = self.devices[i]
            b.stream = torch.cuda.current_stream(q.device).cuda_stream      # the library waits for what is queued here
            taus[i] = tau.data_ptr()
            a = acc[i] if acc is not None else torch.empty((P * P + P + 2,), dtype=torch.float64, device=q.device)
            assert a.is_cuda and a.device.index == self.devices[i] and a.dtype == torch.float64 and a.numel() == P * P + P + 2 and a.is_contiguous()
            accs[i] = a.data_ptr()
            out.append(a)
        assert not accumulate or acc is not None, "accumulate needs the caller's accumulators"
        check(lib().rdyn_regressor_gram_multi_accumulate(self._h, chain._h, batches, taus, accs, 1 if accumulate else 0))
        if sync:
            self.synchronize()
        return out

    def identification_tsqr(self, chain, shards, components=None, layout="sample", out=None, accumulate=False, sync=True):
        """include/rdyn.h: rdyn_identification_tsqr_multi (components=None: rdyn_regressor_tsqr_multi) -- every device the robust R
        factor of its shard, ONE ncclAllGather of the factors, every device folds the stack in the same order.  Returns the list of
        per-device factors in MATH layout ((n1, n1) upper triangular, n1 = P + K + 1), bitwise identical on all devices.  The call
        leaves torch's current stream of every device waiting (on the device) for the collective, so torch work queued afterwards
        sees the results; sync=True additionally waits on the host."""
        import torch
        from ._lib import Batch, LAYOUT_ELEMENT_MAJOR, LAYOUT_SAMPLE_MAJOR
        assert len(shards) == len(self.devices)
        n = chain.getActiveJointsNumber()
        n1 = 10 * chain.getJointsNumber() + (components.columns if components is not None else 0) + 1
        arr, n_comps = (C.cast(components._arr, C.c_void_p), components.n_comps) if components is not None else (None, 0)
        batches = (Batch * len(shards))()
        taus = (C.c_void_p * len(shards))()
        ptrs = (C.c_void_p * len(shards))()
        bufs = []
        for i, (q, dq, ddq, tau) in enumerate(shards):
            for t in (q, dq, ddq, tau):
                assert t.is_cuda and t.device.index == self.devices[i] and t.dtype == torch.float64 and t.is_contiguous()
            b = batches[i]
            b.n_samples = q.shape[1] if layout == "element" else q.shape[0]
            assert (q.shape[0] if layout == "element" else q.shape[1]) == n
            b.q, b.dq, b.ddq = q.data_ptr(), dq.data_ptr(), ddq.data_ptr()
            b.layout = LAYOUT_ELEMENT_MAJOR if layout == "element" else LAYOUT_SAMPLE_MAJOR
            b.device = self.devices[i]
            b.stream = torch.cuda.current_stream(q.device).cuda_stream
            taus[i] = tau.data_ptr()
            with torch.cuda.device(q.device):
                buf = torch.zeros((n1, n1), dtype=torch.float64, device=q.device) if out is None else out[i].t().contiguous()   # column-major for the library
            ptrs[i] = buf.data_ptr()
            bufs.append(buf)
        check(lib().rdyn_identification_tsqr_multi(self._h, chain._h, arr, n_comps, batches, taus, ptrs, 1 if accumulate else 0))
        if sync:
            self.synchronize()
        return [b.t() for b in bufs]

    def synchronize(self):
        check(lib().rdyn_multi_gpu_synchronize(self._h))
