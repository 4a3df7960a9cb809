"""ctypes binding of librdyn_hip.so (include/rdyn.h).  There is no CPU fallback: if the HIP library is
missing the import fails loudly -- build it with ``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C rosdyn_amd/csrc``."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RDYN_LIB_PATH", os.path.join(_HERE, "librdyn_hip.so"))  # override: A/B builds only

RDYN_MAX_JOINTS = 32
RDYN_MAX_SWEPT_JOINTS = 10
LAYOUT_SAMPLE_MAJOR = 0
LAYOUT_ELEMENT_MAJOR = 1
OK = 0
STATUS_NAMES = {0: "RDYN_OK", 1: "RDYN_ERR_INVALID_ARGUMENT", 2: "RDYN_ERR_BASE_NOT_FOUND", 3: "RDYN_ERR_TOOL_NOT_FOUND",
                4: "RDYN_ERR_URDF", 5: "RDYN_ERR_UNSUPPORTED", 6: "RDYN_ERR_JOINT_NOT_FOUND", 7: "RDYN_ERR_NO_DEVICE",
                8: "RDYN_ERR_HIP"}


class Batch(C.Structure):
    _fields_ = [("n_samples", C.c_int64), ("q", C.c_void_p), ("dq", C.c_void_p), ("ddq", C.c_void_p),
                ("layout", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p)]


class RegressorLayout(C.Structure):
    _fields_ = [("stride_sample", C.c_int64), ("stride_row", C.c_int64), ("stride_col", C.c_int64)]


class Component(C.Structure):
    _fields_ = [("type", C.c_int32), ("joint", C.c_int32), ("min_velocity", C.c_double), ("max_velocity", C.c_double),
                ("parameters", C.c_double * 3)]


class MultiItem(C.Structure):
    _fields_ = [("chain", C.c_void_p), ("batch", Batch), ("tau", C.c_void_p), ("Y", C.c_void_p), ("y_layout", RegressorLayout)]


class JointDesc(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("urdf_type", C.c_int32), ("origin_xyz", C.c_double * 3),
                ("origin_quat", C.c_double * 4), ("axis", C.c_double * 3), ("has_limits", C.c_int32),
                ("lower", C.c_double), ("upper", C.c_double), ("velocity", C.c_double), ("effort", C.c_double)]


class LinkDesc(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("has_inertial", C.c_int32), ("mass", C.c_double), ("com_xyz", C.c_double * 3),
                ("com_quat", C.c_double * 4), ("ixx", C.c_double), ("ixy", C.c_double), ("ixz", C.c_double),
                ("iyy", C.c_double), ("iyz", C.c_double), ("izz", C.c_double)]


class ChainDesc(C.Structure):
    _fields_ = [("n_joints", C.c_int32), ("joints", C.POINTER(JointDesc)), ("links", C.POINTER(LinkDesc)),
                ("gravity", C.c_double * 3)]


# every symbol include/rdyn.h declares: name -> (restype, argtypes)
_VP, _I, _DP, _CP = C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_char_p
_BP, _YP = C.POINTER(Batch), C.POINTER(RegressorLayout)
SYMBOLS = {
    "rdyn_chain_from_urdf": (_I, [_CP, _CP, _CP, _DP, C.POINTER(_VP)]),
    "rdyn_chain_from_desc": (_I, [C.POINTER(ChainDesc), C.POINTER(_VP)]),
    "rdyn_chain_clone": (_I, [_VP, C.POINTER(_VP)]),
    "rdyn_chain_destroy": (None, [_VP]),
    "rdyn_last_error": (_CP, []),
    "rdyn_chain_links_number": (_I, [_VP]),
    "rdyn_chain_joints_number": (_I, [_VP]),
    "rdyn_chain_active_joints_number": (_I, [_VP]),
    "rdyn_chain_moveable_joints_number": (_I, [_VP]),
    "rdyn_chain_link_name": (_CP, [_VP, _I]),
    "rdyn_chain_joint_name": (_CP, [_VP, _I]),
    "rdyn_chain_moveable_joint_name": (_CP, [_VP, _I]),
    "rdyn_chain_active_joint_name": (_CP, [_VP, _I]),
    "rdyn_chain_joint_type": (_I, [_VP, _I]),
    "rdyn_chain_gravity": (_I, [_VP, _DP]),
    "rdyn_chain_set_input_joints": (_I, [_VP, C.POINTER(_CP), _I]),
    "rdyn_chain_limits": (_I, [_VP, _DP, _DP, _DP, _DP, _DP]),
    "rdyn_nominal_parameters": (_I, [_VP, _DP]),
    "rdyn_chain_reduction": (_I, [_VP, _VP, _VP, _VP]),
    "rdyn_transformation": (_I, [_VP, _BP, _VP, _VP]),
    "rdyn_jacobian": (_I, [_VP, _BP, _VP]),
    "rdyn_jacobian_link": (_I, [_VP, _BP, _I, _VP]),
    "rdyn_twist": (_I, [_VP, _BP, _VP, _VP]),
    "rdyn_twist_parts": (_I, [_VP, _BP, _VP, _VP, _VP, _VP]),
    "rdyn_jerk_parts": (_I, [_VP, _BP, _VP, _VP, _VP]),
    "rdyn_wrench": (_I, [_VP, _BP, _VP, _VP]),
    "rdyn_joint_torque_ext": (_I, [_VP, _BP, _VP, _VP]),
    "rdyn_joint_torque": (_I, [_VP, _BP, _VP]),
    "rdyn_joint_torque_nonlinear": (_I, [_VP, _BP, _VP]),
    "rdyn_regressor": (_I, [_VP, _BP, _VP, _VP, _YP]),
    "rdyn_chain_joint_constants": (_I, [_VP, _I, _DP, _DP, _DP, _DP]),
    "rdyn_chain_link_parameters": (_I, [_VP, _I, _DP, _DP, _DP]),
    "rdyn_joint_inertia": (_I, [_VP, _BP, _VP]),
    "rdyn_local_ik": (_I, [_VP, _BP, _VP, _DP, C.c_double, _I, _VP, _VP, _VP]),
    "rdyn_local_ik_damped": (_I, [_VP, _BP, _VP, _DP, C.c_double, C.c_double, _I, _VP, _VP, _VP]),
    "rdyn_frame_distance": (_I, [C.c_int64, _VP, _VP, _I, _I, _VP, _VP, _I, _VP]),
    "rdyn_evaluate_all": (_I, [_VP, _BP, _VP]),
    "rdyn_components_columns": (_I, [_VP, _I]),
    "rdyn_components_regressor": (_I, [_VP, _I, _I, _BP, _VP, _YP, _VP]),
    "rdyn_multi_plan_create": (_I, [_VP, _I, C.POINTER(_VP)]),
    "rdyn_multi_plan_regressor": (_I, [_VP, _VP]),
    "rdyn_multi_plan_destroy": (None, [_VP]),
    "rdyn_gram_workspace_bytes": (C.c_size_t, [_I]),
    "rdyn_gram": (_I, [_VP, C.c_int64, C.c_int64, _I, _VP, _VP, _VP, _VP, _I, _VP, C.c_size_t, _I, _VP]),
    "rdyn_identification_gram_workspace_bytes": (C.c_size_t, [_VP, _VP, _I]),
    "rdyn_identification_gram": (_I, [_VP, _VP, _I, _BP, _VP, _VP, _VP, _VP, _I, _VP, C.c_size_t]),
    "rdyn_regressor_gram_workspace_bytes": (C.c_size_t, [_VP, C.c_int64]),
    "rdyn_regressor_gram": (_I, [_VP, _BP, _VP, _VP, _VP, _VP, _I, C.c_int64, _VP, C.c_size_t]),
    "rdyn_multi_gpu_create": (_I, [C.POINTER(C.c_int), _I, C.POINTER(_VP)]),
    "rdyn_multi_gpu_destroy": (None, [_VP]),
    "rdyn_multi_gpu_device_count": (_I, [_VP]),
    "rdyn_multi_gpu_synchronize": (_I, [_VP]),
    "rdyn_regressor_gram_multi": (_I, [_VP, _VP, _BP, C.POINTER(_VP), C.POINTER(_VP)]),
    "rdyn_regressor_gram_multi_accumulate": (_I, [_VP, _VP, _BP, C.POINTER(_VP), C.POINTER(_VP), _I]),
    "rdyn_identification_tsqr_multi": (_I, [_VP, _VP, _VP, _I, _BP, C.POINTER(_VP), C.POINTER(_VP), _I]),
    "rdyn_regressor_tsqr_multi": (_I, [_VP, _VP, _BP, C.POINTER(_VP), C.POINTER(_VP), _I]),
    "rdyn_tsqr_workspace_bytes": (C.c_size_t, [_I]),
    "rdyn_tsqr": (_I, [_VP, C.c_int64, C.c_int64, _I, _VP, _VP, _I, _VP, C.c_size_t, _I, _VP]),
    "rdyn_regressor_tsqr_workspace_bytes": (C.c_size_t, [_VP]),
    "rdyn_regressor_tsqr": (_I, [_VP, _BP, _VP, _VP, _I, _VP, C.c_size_t]),
    "rdyn_identification_tsqr_workspace_bytes": (C.c_size_t, [_VP, _VP, _I]),
    "rdyn_identification_tsqr": (_I, [_VP, _VP, _I, _BP, _VP, _VP, _I, _VP, C.c_size_t]),
    "rdyn_tsqr_last_report": (_I, [_VP, _VP, _I, C.c_int64, _VP, _I, _VP, _VP]),
    "rdyn_tsqr_rows_last_report": (_I, [_I, C.c_int64, _VP, _I, _VP, _VP]),
    "rdyn_tsqr_combine_host": (_I, [_DP, _I, _I, _DP]),
    "rdyn_solve_normal_equations": (_I, [_DP, _DP, _I, C.c_double, _DP, C.POINTER(C.c_int)]),
    "rdyn_gram_r_factor": (_I, [_DP, _I, C.c_double, _DP, C.POINTER(C.c_int32), C.POINTER(C.c_int)]),
    "rdyn_solve_r_factor": (_I, [_DP, C.c_int64, _I, _I, _DP, C.c_double, _DP, C.POINTER(C.c_int)]),
}

class RdynTsqrReport(C.Structure):
    _fields_ = [("route", C.c_int32), ("stage", C.c_int32), ("n_deferred", C.c_int32), ("reserved", C.c_int32),
                ("gamma", C.c_double * 2), ("rho", C.c_double * 2)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("librdyn_hip.so not built (expected at %s); run __graft_entry__.build() -- "
                              "there is deliberately no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: torch wheels bundle their own libamdhip64.so.7 (same SONAME as
        # /opt/rocm's).  Whichever is loaded first serves both; loading ours first and torch's second left
        # this library without a device (hipGetDevice failed on MI355X/ROCm 7.2), so let torch load first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


class RdynError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status
        self.message = message


def check(status):
    if status != OK:
        msg = lib().rdyn_last_error().decode()
        # same exception types the reference throws for the same conditions
        if status in (2, 3):   # std::runtime_error, primitives_impl.h:486-501
            raise RdynError(status, msg)
        if status == 1:        # std::invalid_argument, primitives_impl.h:1302
            raise ValueError(msg)
        raise RdynError(status, msg)
