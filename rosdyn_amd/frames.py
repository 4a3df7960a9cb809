"""Batched mirrors of the free functions of rosdyn_core/include/rosdyn_core/frame_distance.h over include/rdyn.h:
rdyn_frame_distance.  Frames are the records Chain.getTransformation returns: (N, 4, 3) sample-major or (4, 3, N)
element-major float64 CUDA tensors (columns of the 3 x 4 [R | p])."""
from ._lib import LAYOUT_ELEMENT_MAJOR, LAYOUT_SAMPLE_MAJOR, check, lib

AXIS_ANGLE, QUAT, QUAT_JAC = 0, 1, 2


def _run(T_wa, T_wb, layout, kind, with_jacobian):
    import torch
    elem = layout == "element"
    N = T_wa.shape[-1] if elem else T_wa.shape[0]
    shape = (4, 3, N) if elem else (N, 4, 3)
    for t in (T_wa, T_wb):
        if tuple(t.shape) != shape or t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("frames must be contiguous float64 CUDA tensors of shape %s" % (shape,))
    d = torch.empty((6, N) if elem else (N, 6), dtype=torch.float64, device=T_wa.device)
    J = torch.empty((6, 6, N) if elem else (N, 6, 6), dtype=torch.float64, device=T_wa.device) if with_jacobian else None
    check(lib().rdyn_frame_distance(N, T_wa.data_ptr(), T_wb.data_ptr(), LAYOUT_ELEMENT_MAJOR if elem else LAYOUT_SAMPLE_MAJOR, kind,
                                    d.data_ptr(), J.data_ptr() if J is not None else None,
                                    T_wa.device.index if T_wa.device.index is not None else -1,
                                    torch.cuda.current_stream(T_wa.device).cuda_stream))
    return (d, J) if with_jacobian else d


def getFrameDistance(T_wa, T_wb, layout="sample"):              # frame_distance.h:44
    return _run(T_wa, T_wb, layout, AXIS_ANGLE, False)


def getFrameDistanceQuat(T_wa, T_wb, layout="sample"):          # frame_distance.h:73
    return _run(T_wa, T_wb, layout, QUAT, False)


def getFrameDistanceQuatJac(T_wa, T_wb, layout="sample"):       # frame_distance.h:112 -> (distance, jacobian[s] = J^T image: J(r, c) at [s, c, r])
    return _run(T_wa, T_wb, layout, QUAT_JAC, True)
