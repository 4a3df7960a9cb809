"""Per-joint additive components (friction, spring) -- batched mirror of rosdyn_core's ComponentBase family
(friction_polynomial1.h, friction_polynomial2.h, ideal_spring.h) over include/rdyn.h: rdyn_components_regressor."""
import ctypes as C

from ._lib import LAYOUT_ELEMENT_MAJOR, LAYOUT_SAMPLE_MAJOR, Batch, Component, RegressorLayout, check, lib

FRICTION1, FRICTION2, SPRING = 0, 1, 2


class ComponentSet(object):
    """components: list of dicts {type, joint, min_velocity, max_velocity, parameters}; `joint` indexes the ACTIVE joints."""

    def __init__(self, components, n_active):
        self.n_active = n_active
        self._arr = (Component * len(components))()
        for i, c in enumerate(components):
            a = self._arr[i]
            a.type, a.joint = int(c["type"]), int(c["joint"])
            a.min_velocity, a.max_velocity = float(c.get("min_velocity", 0.0)), float(c.get("max_velocity", 0.0))
            p = list(c["parameters"]) + [0.0, 0.0, 0.0]
            a.parameters[:] = p[:3]
        self.n_comps = len(components)
        self.columns = lib().rdyn_components_columns(C.cast(self._arr, C.c_void_p), self.n_comps)

    def getNominalParameters(self):
        out = []
        for a in self._arr:
            out += list(a.parameters[:3 if a.type == FRICTION2 else 2])
        return out

    def getRegressor(self, q, Dq, layout="sample", out=None, tau_add=None):
        """Returns C: layout="sample" -> (N, K, n) (per-sample column-major n x K); "element" -> (K, n, N)."""
        import torch
        lay = LAYOUT_ELEMENT_MAJOR if layout == "element" else LAYOUT_SAMPLE_MAJOR
        N = q.shape[1] if lay == LAYOUT_ELEMENT_MAJOR else q.shape[0]
        n, K = self.n_active, self.columns
        assert (q.shape[0] if lay == LAYOUT_ELEMENT_MAJOR else q.shape[1]) == n and Dq.shape == q.shape
        b = Batch(N, q.data_ptr(), Dq.data_ptr(), None, lay, q.device.index if q.device.index is not None else -1,
                  torch.cuda.current_stream(q.device).cuda_stream)
        if lay == LAYOUT_ELEMENT_MAJOR:
            shape, yl = (K, n, N), RegressorLayout(1, N, n * N)
        else:
            shape, yl = (N, K, n), RegressorLayout(n * K, 1, n)
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=q.device)
        check(lib().rdyn_components_regressor(C.cast(self._arr, C.c_void_p), self.n_comps, n, C.byref(b), out.data_ptr(), C.byref(yl),
                                              tau_add.data_ptr() if tau_add is not None else None))
        return out
