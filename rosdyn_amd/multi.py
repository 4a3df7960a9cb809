"""Mixed-chain batch (BASELINE.json configs[4]): many (chain, batch) items evaluated by one kernel launch per
joint-count group -- include/rdyn.h: rdyn_multi_plan_*."""
import ctypes as C

from ._lib import LAYOUT_ELEMENT_MAJOR, MultiItem, RegressorLayout, check, lib


class MultiChainRegressor(object):
    """Freezes a list of items [(chain, q, Dq, DDq)] (element-major (n, S) float64 CUDA tensors) and allocates their
    outputs: tau[i] (n, S) and Y[i] -- (P, n, S) element-major (default), (S, P, n) per-sample images (y_layout="per_sample": the
    memory image of the reference's Eigen matrix per sample) or the stacked column-major (S n) x P matrix as a (P, S n) tensor
    (y_layout="stacked").  run() launches one kernel per (joint count, layout kind) group; nothing is copied."""

    def __init__(self, items, with_torque=True, y_layout="element"):
        import torch
        self._keep = []
        self.tau, self.Y = [], []
        arr = (MultiItem * len(items))()
        for i, (chain, q, dq, ddq) in enumerate(items):
            n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
            for t in (q, dq, ddq):
                assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.shape == q.shape and t.shape[0] == n
            S = q.shape[1]
            shape, lay = {"element": ((P, n, S), (1, S, n * S)), "per_sample": ((S, P, n), (n * P, 1, n)),
                          "stacked": ((P, S * n), (n, 1, S * n))}[y_layout]
            Y = torch.empty(shape, dtype=torch.float64, device=q.device)
            tau = torch.empty((n, S), dtype=torch.float64, device=q.device) if with_torque else None
            it = arr[i]
            it.chain = chain._h
            it.batch.n_samples = S
            it.batch.q, it.batch.dq, it.batch.ddq = q.data_ptr(), dq.data_ptr(), ddq.data_ptr()
            it.batch.layout = LAYOUT_ELEMENT_MAJOR
            it.batch.device = q.device.index if q.device.index is not None else -1
            it.tau = tau.data_ptr() if tau is not None else None
            it.Y = Y.data_ptr()
            it.y_layout = RegressorLayout(*lay)
            self._keep.append((chain, q, dq, ddq))
            self.tau.append(tau)
            self.Y.append(Y)
        self._device = items[0][1].device
        self._h = C.c_void_p()
        check(lib().rdyn_multi_plan_create(C.cast(arr, C.c_void_p), len(items), C.byref(self._h)))

    def run(self):
        import torch
        check(lib().rdyn_multi_plan_regressor(self._h, torch.cuda.current_stream(self._device).cuda_stream))
        return self.Y, self.tau

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().rdyn_multi_plan_destroy(h)
            self._h = None
