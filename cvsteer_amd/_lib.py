"""ctypes binding of libcvsteer_hip.so (the C ABI declared in include/cvsteer_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("CVSTEER_HIP_LIB", os.path.join(_HERE, "libcvsteer_hip.so"))  # override: diagnostic twin only

OK, E_BADARG, E_SIZE, E_HIP, E_NOMEM, E_STATE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
KIND_G2, KIND_G4 = 2, 4
MEM_HOST, MEM_DEVICE = 0, 1
DEPTH_U8 = 0x100
ABI_VERSION = 2
OPT_ATAN_MODE, OPT_STRIP_ROWS, OPT_FIND_ON, OPT_G4_EXTENSIONS, OPT_BLOCK_ORDER, OPT_PERSIST_STATE = 1, 2, 3, 6, 8, 9
OPT_AUTOTUNE = 12
OPT_HOST_OVERLAP = 13
OPT_STATE_LAYOUT = 14
ORDER_PLAIN, ORDER_XCD_COLUMNS, ORDER_DYNAMIC_TAIL = 0, 1000000, 2000000
PLANE_BASIS0, PLANE_C1, PLANE_C2, PLANE_C3, PLANE_THETA, PLANE_STRENGTH = 0, 32, 33, 34, 35, 36


class Plane(C.Structure):
    """struct cvs_plane"""
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32),
                ("step", C.c_size_t), ("mem", C.c_int32)]


class CvsError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: %s (%d)" % (where, lib().cvs_status_string(status).decode(), status)
        if detail:
            msg += ": " + detail
        super().__init__(msg)


class LaunchInfo(C.Structure):
    """struct cvs_launch_info"""
    _fields_ = [("struct_size", C.c_uint32), ("block_order", C.c_int32), ("strip_rows", C.c_int32), ("nt_stores", C.c_int32),
                ("state_layout", C.c_int32), ("warm", C.c_int32), ("tuning_launches", C.c_int32), ("tuned", C.c_int32), ("tune_state", C.c_int32), ("wg_per_cu", C.c_int32), ("literal_taps", C.c_int32)]


_PP = C.POINTER(Plane)
_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int)

# every symbol include/cvsteer_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "cvs_abi_version": (C.c_int, []),
    "cvs_status_string": (C.c_char_p, [C.c_int]),
    "cvs_num_basis": (C.c_int, [C.c_int]),
    "cvs_make_taps": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, _FP]),
    "cvs_basis_taps": (C.c_int, [C.c_int, C.c_int, _IP, _IP]),
    "cvs_steer_weights": (C.c_int, [C.c_int, C.c_float, _FP]),
    "cvs_create": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_void_p)]),
    "cvs_destroy": (C.c_int, [C.c_void_p]),
    "cvs_release_cached_memory": (C.c_int, []),
    "cvs_last_error": (C.c_char_p, [C.c_void_p]),
    "cvs_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cvs_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cvs_get_option": (C.c_int, [C.c_void_p, C.c_int, _IP]),
    "cvs_get_launch_info": (C.c_int, [C.c_void_p, C.POINTER(LaunchInfo)]),
    "cvs_taps": (C.c_int, [C.c_void_p, C.c_int, _FP]),
    "cvs_kind": (C.c_int, [C.c_void_p, _IP, _IP, _FP]),
    "cvs_shape": (C.c_int, [C.c_void_p, _IP, _IP]),
    "cvs_sync": (C.c_int, [C.c_void_p]),
    "cvs_setup": (C.c_int, [C.c_void_p, _PP, C.c_uint]),
    "cvs_setup_steer": (C.c_int, [C.c_void_p, _PP, C.c_uint, C.c_float, _PP, _PP]),
    "cvs_setup_rows": (C.c_int, [C.c_void_p, _PP, C.c_uint, C.c_int, C.c_int]),
    "cvs_state_plane": (C.c_int, [C.c_void_p, C.c_int, _PP]),
    "cvs_read_state": (C.c_int, [C.c_void_p, C.c_int, _PP]),
    "cvs_steer_scalar": (C.c_int, [C.c_void_p, C.c_float, _PP, _PP, _PP, _PP, _PP]),
    "cvs_steer_map": (C.c_int, [C.c_void_p, _PP, _PP, _PP, _PP, _PP, _PP]),
    "cvs_steer_point": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, _FP]),
    "cvs_mag_phase": (C.c_int, [C.c_void_p, _PP, _PP, _PP, _PP]),
    "cvs_wrap": (C.c_int, [C.c_void_p, _PP, _PP]),
    "cvs_phase_weights": (C.c_int, [C.c_void_p, _PP, _PP, C.c_float, C.c_int, C.c_float]),
    "cvs_find": (C.c_int, [C.c_void_p, _PP, _PP, _PP, _PP, _PP]),
    "cvs_pipeline": (C.c_int, [C.c_void_p, _PP, C.POINTER(_PP)]),
    "cvs_pipeline_batch": (C.c_int, [C.c_void_p, _PP, C.c_int, _PP]),
    "cvs_select_frame": (C.c_int, [C.c_void_p, C.c_int]),
    "cvs_num_frames": (C.c_int, [C.c_void_p, _IP]),
    "cvs_pyr_down": (C.c_int, [C.c_void_p, _PP, _PP]),
    "cvs_setup_pyr": (C.c_int, [C.c_void_p, _PP, C.c_uint, _PP]),
    "cvs_pyramid_setup": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _PP, C.c_uint, _PP]),
    "cvs_normalize_u8": (C.c_int, [C.c_void_p, _PP, C.c_void_p, C.c_size_t, C.c_int]),
    "cvs_convert_u8": (C.c_int, [C.c_void_p, _PP, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_int]),
    "cvs_normalize_u8_batch": (C.c_int, [C.c_void_p, _PP, C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_int]),
    "cvs_convert_u8_batch": (C.c_int, [C.c_void_p, _PP, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_void_p), C.c_size_t, C.c_int]),
}



class BatchCfg(C.Structure):
    """struct cvs_batch_cfg"""
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("n_frames", C.c_int32), ("outputs", C.c_uint32),
                ("root", C.c_int32), ("gather", C.c_int32), ("self_via_transport", C.c_int32)]


class BatchTiming(C.Structure):
    """struct cvs_batch_timing"""
    _fields_ = [("scatter_ms", C.c_double), ("compute_ms", C.c_double), ("gather_ms", C.c_double)]


BATCH_ID_BYTES = 128
TRANSPORT_NONE, TRANSPORT_RCCL, TRANSPORT_COPY = 0, 1, 2
_VPP = C.POINTER(C.c_void_p)
SIGNATURES.update({
    "cvs_batch_create_local": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, _IP, _VPP]),
    "cvs_batch_unique_id": (C.c_int, [C.c_void_p]),
    "cvs_batch_create_rank": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, _VPP]),
    "cvs_batch_destroy": (C.c_int, [C.c_void_p]),
    "cvs_batch_last_error": (C.c_char_p, [C.c_void_p]),
    "cvs_batch_info": (C.c_int, [C.c_void_p, _IP, _IP, _IP]),
    "cvs_batch_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cvs_batch_set_u8_gain": (C.c_int, [C.c_void_p, C.c_float]),
    "cvs_batch_run": (C.c_int, [C.c_void_p, C.POINTER(BatchCfg), _PP, _PP, C.POINTER(BatchTiming)]),
    "cvs_batch_local_result": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(_FP), _IP, _IP, _IP, _IP]),
    "cvs_batch_pyramid_setup": (C.c_int, [C.c_void_p, _PP, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.POINTER(BatchTiming)]),
    "cvs_batch_level": (C.c_int, [C.c_void_p, C.c_int, _VPP, _PP]),
})

_lib = None


def lib_path():
    return _SO


def lib():
    """Load libcvsteer_hip.so.  Fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(
                "cvsteer_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C cvsteer_amd/csrc`.  There is no CPU fallback." % _SO)
        L = C.CDLL(_SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        if L.cvs_abi_version() != ABI_VERSION:
            raise ImportError("cvsteer_amd: ABI version mismatch: the library says %d, this binding was written for %d" % (L.cvs_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def abi_version():
    return lib().cvs_abi_version()
