"""Loader for libdcx.so (the HIP implementation) via ctypes.

There is no CPU implementation of the hot path in this package: if the library is missing
or no GPU is visible, every op raises.  `import torch` happens first on purpose — torch
bundles its own libamdhip64; loading ours afterwards binds to the already-mapped runtime so
both share streams and allocations.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see docstring)

from ._fkdesc import FkDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCX_LIB: developer override used by tools/sweep.py to A/B kernel variants; never a CPU path
LIB_PATH = os.environ.get("DCX_LIB") or os.path.join(_HERE, "libdcx.so")

# every symbol include/dcx.h declares: (restype, argtypes)
_c_fp = C.c_void_p  # device/host float pointers travel as raw addresses
SYMBOLS = {
    "dcx_version": (C.c_int, []),
    "dcx_last_error": (C.c_char_p, []),
    "dcx_device_count": (C.c_int, []),
    "dcx_debug_set": (C.c_int, [C.c_char_p, C.c_int64]),
    "dcx_debug_clock_probe": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_int32), C.c_void_p]),
    "dcx_model_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(FkDesc), C.c_int, C.POINTER(C.c_float),
                                   _c_fp, _c_fp, C.c_int64, C.c_int32, C.c_int32]),
    "dcx_model_create_ex": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(FkDesc), C.c_int, C.POINTER(C.c_float),
                                      _c_fp, _c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "dcx_model_update": (C.c_int, [C.c_void_p, _c_fp, _c_fp, C.c_int64, C.c_void_p]),
    "dcx_model_destroy": (None, [C.c_void_p]),
    "dcx_model_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dcx_score": (C.c_int, [C.c_void_p, _c_fp, C.c_int64, _c_fp, C.c_void_p]),
    "dcx_score_grad": (C.c_int, [C.c_void_p, _c_fp, C.c_int64, _c_fp, _c_fp, _c_fp, C.c_void_p]),
    "dcx_score_jac": (C.c_int, [C.c_void_p, _c_fp, C.c_int64, _c_fp, _c_fp, C.c_void_p]),
    "dcx_score_hess": (C.c_int, [C.c_void_p, _c_fp, C.c_int64, _c_fp, _c_fp, _c_fp, C.c_void_p]),
    "dcx_score_hinge_grad": (C.c_int, [C.c_void_p, _c_fp, C.c_int64, C.c_float, C.c_float, _c_fp, _c_fp, C.c_void_p]),
    "dcx_score_hinge_grad_mc": (C.c_int, [C.c_void_p, _c_fp, C.c_int64, C.POINTER(C.c_float), C.c_float, _c_fp, _c_fp, C.c_void_p]),
    "dcx_traj_adam_run_mc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_void_p]),
    "dcx_traj_adam_step_mc": (C.c_int, [C.c_int, C.POINTER(FkDesc), C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_int32,
                                        C.c_void_p]),
    "dcx_traj_adam_step": (C.c_int, [C.c_int, C.POINTER(FkDesc), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "dcx_traj_adam_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dcx_escape_work_bytes": (C.c_size_t, [C.c_void_p, C.c_int64]),
    "dcx_escape_adam": (C.c_int, [C.c_void_p, _c_fp, C.c_int64, _c_fp, C.c_void_p, C.c_void_p, C.c_size_t, _c_fp, C.c_void_p,
                                  C.c_void_p]),
    "dcx_train_perceptron": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, _c_fp, C.c_int64, C.c_int32, _c_fp,
                                       C.c_int32, _c_fp, _c_fp, _c_fp, C.c_int32, _c_fp, C.c_void_p]),
    "dcx_train_perceptron_ex": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, _c_fp, C.c_int64, C.c_int32, _c_fp,
                                          C.c_int32, _c_fp, _c_fp, _c_fp, C.c_int32, _c_fp, C.c_int32, C.c_void_p]),
    "dcx_fkine": (C.c_int, [C.c_int, C.POINTER(FkDesc), _c_fp, C.c_int64, _c_fp, C.c_void_p]),
    "dcx_fkine_vjp": (C.c_int, [C.c_int, C.POINTER(FkDesc), _c_fp, _c_fp, C.c_int64, _c_fp, C.c_void_p]),
    "dcx_dh_frames": (C.c_int, [C.c_int, _c_fp, C.c_int64, C.c_int32, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, C.c_void_p]),
    "dcx_dh_frames_vjp": (C.c_int, [C.c_int, _c_fp, C.c_int64, C.c_int32, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, C.c_void_p]),
    "dcx_euler_frames": (C.c_int, [C.c_int, _c_fp, C.c_int64, _c_fp, C.c_void_p]),
    "dcx_euler_frames_vjp": (C.c_int, [C.c_int, _c_fp, _c_fp, C.c_int64, _c_fp, C.c_void_p]),
    "dcx_kernel_matrix": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float), _c_fp, C.c_int64, _c_fp, C.c_int64,
                                    C.c_int32, _c_fp, C.c_void_p]),
    "dcx_solve_work_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "dcx_solve": (C.c_int, [C.c_int, _c_fp, _c_fp, C.c_int64, C.c_int64, _c_fp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32,
                            C.c_void_p]),
}



class TrajState(C.Structure):
    """ctypes mirror of dcx_traj_state (include/dcx.h)"""
    _fields_ = [("n_paths", C.c_int32), ("n_waypoints", C.c_int32)] + [
        (n, C.c_void_p) for n in ("path", "adam_m", "adam_v", "limits", "col_score", "col_grad", "stats", "lowest_loss",
                                  "lowest_obj", "lowest_path", "best_valid_obj", "best_valid_path", "done", "steps")]


class TrajOpts(C.Structure):
    """ctypes mirror of dcx_traj_opts (include/dcx.h)"""
    _fields_ = [(n, C.c_float) for n in ("lr", "beta1", "beta2", "eps", "w_diff", "w_collision", "w_max_move",
                                         "w_joint_limit", "safety_margin", "max_speed", "valid_tol", "grad_tol")]


class EscapeOpts(C.Structure):
    """ctypes mirror of dcx_escape_opts (include/dcx.h)"""
    _fields_ = [(n, C.c_float) for n in ("lr", "beta1", "beta2", "eps")] + [
        (n, C.c_int32) for n in ("n_steps", "record_freq", "joint", "compact_every")] + [("wrap_mask", C.c_uint64)]


_lib = None


class DcxError(RuntimeError):
    pass


class DcxUnsupported(DcxError):
    """DCX_ERR_UNSUPPORTED: the shape / kind is outside what the library is compiled for"""


def load():
    """Load libdcx.so (once) and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DcxError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C diffco_amd/csrc -j8`.  diffco_amd has no CPU fallback for the score/grad path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().dcx_last_error()
        raise (DcxUnsupported if rc == 2 else DcxError)(f"libdcx error {rc}: {msg.decode() if msg else '?'}")


_gpu_ok = False


def require_gpu():
    """the loaded library, after checking ONCE per process that a GPU is there (every op calls this: the check itself -
    torch.cuda.is_available + hipGetDeviceCount - cost 2 us of a 10 us call)"""
    global _gpu_ok
    if _gpu_ok:
        return _lib
    lib = load()
    if not torch.cuda.is_available() or lib.dcx_device_count() < 1:
        raise DcxError("diffco_amd: no MI355X/HIP device visible — the score/grad path is HIP-only "
                       "(there is deliberately no CPU fallback)")
    _gpu_ok = True
    return lib
