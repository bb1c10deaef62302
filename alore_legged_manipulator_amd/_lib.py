"""ctypes binding of the C ABI (include/alore_nmpc.h) of libalore_nmpc.so.

There is deliberately no fallback: if the HIP library is missing or cannot be
loaded the import of the solver fails loudly (``NmpcLibraryError``)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ALORE_NMPC_LIB: another build of the same C ABI (A/B timing of kernel variants); never a CPU library
LIB_PATH = os.environ.get("ALORE_NMPC_LIB") or os.path.join(HERE, "libalore_nmpc.so")

FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int)


class NmpcLibraryError(RuntimeError):
    pass


class NmpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"alore_nmpc error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("N", C.c_int), ("dt", C.c_float), ("device", C.c_int), ("max_as_iter", C.c_int),
                ("lanes_per_problem", C.c_int), ("warm_start_steps", C.c_int)]


BATCH_FLOAT_MEMBERS = ("x", "u", "od", "y", "yN", "W", "WN", "x0", "lbValues", "ubValues", "dual")
BATCH_MEMBERS = BATCH_FLOAT_MEMBERS + ("status", "n_iter", "kkt", "obj")


class Batch(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in BATCH_MEMBERS]


class LinOut(C.Structure):
    _fields_ = [("d", C.c_void_p), ("evGx", C.c_void_p), ("evGu", C.c_void_p)]


class PolynomeMsg(C.Structure):
    """alore_polynome (include/alore_nmpc.h): one planner message, P/utils/carstatemsgs/msg/Polynome.msg"""
    _fields_ = [("n_pieces", C.c_int), ("innerpoints", C.c_void_p), ("t_pts", C.c_void_p),
                ("init_p", C.c_double * 2), ("init_v", C.c_double * 2), ("init_a", C.c_double * 2),
                ("tail_p", C.c_double * 2), ("tail_v", C.c_double * 2), ("tail_a", C.c_double * 2),
                ("start_position", C.c_double * 3), ("ICR", C.c_double * 3), ("traj_start_time", C.c_double)]


class PlantParams(C.Structure):
    """alore_plant_params (include/alore_nmpc.h): the reference's simulator node constants"""
    _fields_ = [("max_acc", C.c_double), ("max_domega", C.c_double), ("pose_pub_period", C.c_double),
                ("state_propa_period", C.c_double), ("substeps", C.c_int)]


class LaunchInfo(C.Structure):
    _fields_ = [("lanes_per_problem", C.c_int), ("problems_per_block", C.c_int), ("threads_per_block", C.c_int),
                ("grid", C.c_int), ("lds_bytes_per_block", C.c_int), ("last_kernel_ms", C.c_float)]


class TwoPhaseInfo(C.Structure):
    """alore_nmpc_two_phase_info (include/alore_nmpc.h)"""
    _fields_ = [("last_grid_two_phase", C.c_int), ("two_phase_batches", C.c_int), ("tail_workgroups_per_batch", C.c_int),
                ("lag_units", C.c_int), ("tail_share", C.c_float)]


# every symbol include/alore_nmpc.h declares: (name, restype, argtypes)
SYMBOLS = (
    ("alore_nmpc_create", C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    ("alore_nmpc_destroy", C.c_int, [C.c_void_p]),
    ("alore_nmpc_last_error", C.c_char_p, [C.c_void_p]),
    ("alore_nmpc_version", C.c_char_p, []),
    ("alore_nmpc_batch_alloc", C.c_int, [C.c_void_p, C.c_int, C.POINTER(Batch)]),
    ("alore_nmpc_batch_free", C.c_int, [C.c_void_p, C.POINTER(Batch)]),
    ("alore_nmpc_host_alloc", C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    ("alore_nmpc_host_free", C.c_int, [C.c_void_p]),
    ("alore_nmpc_comm_unique_id", C.c_int, [C.c_char_p]),
    ("alore_nmpc_comm_create", C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    ("alore_nmpc_comm_destroy", C.c_int, [C.c_void_p]),
    ("alore_nmpc_comm_all_gather", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_int, C.POINTER(Batch), C.c_void_p]),
    ("alore_nmpc_comm_last_error", C.c_char_p, []),
    ("alore_nmpc_batch_upload", C.c_int, [C.c_void_p, C.POINTER(Batch), C.POINTER(Batch), C.c_int, C.c_void_p]),
    ("alore_nmpc_batch_download", C.c_int, [C.c_void_p, C.POINTER(Batch), C.POINTER(Batch), C.c_int, C.c_void_p]),
    ("alore_nmpc_batch_default_bounds", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_void_p]),
    ("alore_nmpc_rti", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_int, C.c_void_p]),
    ("alore_nmpc_rti_many", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("alore_nmpc_set_launch_overlap", C.c_int, [C.c_void_p, C.c_int]),
    ("alore_nmpc_rti_many_prepare", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_int]),
    ("alore_nmpc_synchronize", C.c_int, [C.c_void_p, C.c_void_p]),
    ("alore_nmpc_set_many_mode", C.c_int, [C.c_void_p, C.c_int]),
    ("alore_nmpc_set_two_phase", C.c_int, [C.c_void_p, C.c_int]),
    ("alore_nmpc_get_two_phase_info", C.c_int, [C.c_void_p, C.POINTER(TwoPhaseInfo)]),
    ("alore_nmpc_set_problem_mask", C.c_int, [C.c_void_p, C.c_void_p]),
    ("alore_nmpc_condense", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_dense_qp", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_input_column", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p]),
    ("alore_nmpc_linearize", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.POINTER(LinOut), C.c_void_p]),
    ("alore_nmpc_forward_simulate", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_void_p]),
    ("alore_nmpc_shift", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_set_shared_members", C.c_int, [C.c_void_p, C.c_uint]),
    ("alore_nmpc_refs_init", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    ("alore_nmpc_refs_set_trajectory", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                 C.c_double, C.c_double, C.c_double, C.c_void_p]),
    ("alore_nmpc_refs_set_polynomes", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]),
    ("alore_nmpc_refs_set_from_backend", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    ("alore_nmpc_refs_download", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_plant_init", C.c_int, [C.c_void_p, C.c_void_p]),
    ("alore_nmpc_plant_set_state", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_plant_get_state", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_closed_loop_tick", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_double, C.c_int, C.c_void_p]),
    ("alore_nmpc_refs_at_goal", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_refs_eval", C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_closed_loop_reset", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_closed_loop_run", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p]),
    ("alore_nmpc_refs_sample", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p]),
    ("alore_nmpc_set_linearization_point", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("alore_nmpc_get_launch_info", C.c_int, [C.c_void_p, C.POINTER(LaunchInfo)]),
    ("alore_nmpc_set_timing", C.c_int, [C.c_void_p, C.c_int]),
)

_lib = None


def load():
    """Load libalore_nmpc.so and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (same soname as /opt/rocm's): import it
    # first so that the whole process runs on ONE runtime
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NmpcLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise NmpcLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NmpcLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
