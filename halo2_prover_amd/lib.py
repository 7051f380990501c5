"""ctypes loader for the C-ABI library (include/h2hip.h).

The product path has no CPU fallback: if libh2hip.so is missing or cannot be loaded this
module raises, and every compute entry point returns an error status without a GPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libh2hip.so")

H2_BN254, H2_PALLAS, H2_VESTA = 0, 1, 2
CURVES = {"bn254": H2_BN254, "pallas": H2_PALLAS, "vesta": H2_VESTA}

H2_OK = 0
STATUS_NAMES = {0: "H2_OK", -1: "H2_EINVAL", -2: "H2_ENOMEM", -3: "H2_EDEVICE", -4: "H2_EHANDLE", -5: "H2_ENOTINIT",
                -6: "H2_EPROOF"}

# every symbol include/h2hip.h declares: name -> (restype, argtypes)
_P, _Z, _I, _U32, _U64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64


class Profile(ctypes.Structure):
    _fields_ = [("launches", _U64), ("kernel_ms", ctypes.c_double), ("algorithmic_bytes", ctypes.c_double)]


class MsmPlan(ctypes.Structure):
    _fields_ = [("window_bits", _U32), ("windows", _U32), ("buckets", _U32), ("table_bytes", _U64)]


SYMBOLS = {
    "h2_init": (_I, [_I]),
    "h2_init_devices": (_I, [_I, _P]),
    "h2_device_count": (_I, []),
    "h2_shutdown": (_I, []),
    "h2_strerror": (ctypes.c_char_p, [_I]),
    "h2_last_device_error": (ctypes.c_char_p, []),
    "h2_version": (_I, []),
    "h2_bases_register": (_I, [_I, _P, _Z, ctypes.POINTER(_U64)]),
    "h2_bases_register_device": (_I, [_I, _P, _Z, ctypes.POINTER(_U64)]),
    "h2_bases_release": (_I, [_U64]),
    "h2_bases_len": (ctypes.c_int64, [_U64]),
    "h2_msm": (_I, [_I, _U64, _P, _Z, _P]),
    "h2_msm_batch": (_I, [_I, _U64, _P, _Z, _Z, _P]),
    "h2_msm_device": (_I, [_I, _U64, _P, _Z, _Z, _P, _P]),
    "h2_msm_device_range": (_I, [_I, _U64, _P, _Z, _Z, _Z, _Z, _P, _P]),
    "h2_points_sum_device": (_I, [_I, _P, _Z, _Z, _P, _P]),
    "h2_stream_wait_msm_tail": (_I, [_P]),
    "h2_msm_device_multi": (_I, [_I, _P, _P, _Z, _Z, _Z, _Z, _P, _P]),
    "h2_ntt": (_I, [_I, _P, _P, _U32]),
    "h2_ntt_batch": (_I, [_I, _P, _Z, _P, _U32]),
    "h2_ntt_device": (_I, [_I, _P, _Z, _P, _U32, _P]),
    "h2_fft_group": (_I, [_I, _P, _P, _U32]),
    "h2_fft_group_device": (_I, [_I, _P, _P, _U32, _P]),
    "h2_ntt_scaled_device": (_I, [_I, _P, _Z, _P, _U32, _P, _P]),
    "h2_poly_scale_device": (_I, [_I, _P, _Z, _Z, _P, _P]),
    "h2_poly_coset_device": (_I, [_I, _P, _Z, _Z, _P, _P]),
    "h2_poly_mul_periodic_device": (_I, [_I, _P, _Z, _Z, _P, _Z, _P]),
    "h2_poly_pointwise_device": (_I, [_I, _I, _P, _P, _Z, _P]),
    "h2_poly_divide_linear_device": (_I, [_I, _P, _Z, _P, _P, _P]),
    "h2_poly_prefix_product_device": (_I, [_I, _P, _Z, _P, _P]),
    "h2_chacha20_scalars_device": (_I, [_I, _P, _U64, _Z, _P, _P]),
    "h2_poly_inverse_device": (_I, [_I, _P, _Z, _P]),
    "h2_msm_plan": (_I, [_U64, ctypes.POINTER(MsmPlan)]),
    "h2_srs_generate": (_I, [_I, _P, _Z, _P, _P]),
    "h2_fixed_base_mul": (_I, [_I, _P, _Z, _P, _P]),
    "h2_setup": (_I, [_U32, _P, _P, _P, _Z, ctypes.POINTER(_Z)]),
    "h2_generate_proof": (_I, [_P, _Z, ctypes.c_char_p, _I, _P, _P, _P, _Z, ctypes.POINTER(_Z)]),
    "h2_verify_proof": (_I, [_P, _Z, _P, _Z, ctypes.c_char_p, _I, ctypes.POINTER(_I)]),
    "h2_simulate": (_I, [ctypes.c_char_p, _I, _P, _Z, ctypes.POINTER(_Z)]),
    "h2_circuit_count": (_I, []),
    "h2_params_cache_clear": (_I, []),
    "h2_key_cache": (_I, [_I]),
    "h2_profile_enable": (_I, [_I]),
    "h2_profile_read": (_I, [ctypes.POINTER(Profile)]),
}
# include/h2hip_selftest.h (host instantiation of the device templates; not a compute path)
SELFTEST_SYMBOLS = {
    "h2_selftest_field_op": (_I, [_I, _I, _P, _P, _P]),
    "h2_selftest_curve_op": (_I, [_I, _I, _P, _P, _P]),
    "h2_selftest_digits": (_I, [_I, _P, _Z, _P, _U32]),
    "h2_selftest_field_op_device": (_I, [_I, _I, _P, _P, _P, _Z]),
    "h2_selftest_curve_op_device": (_I, [_I, _I, _P, _P, _P, _Z]),
    "h2_selftest_set_msm_max_entries": (_I, [_U64]),
    "h2_selftest_modmul_rate": (_I, [_I, _I, _I, ctypes.POINTER(ctypes.c_double)]),
    "h2_selftest_host": (_I, [_I, _P, _Z, _P, _Z, ctypes.POINTER(_Z)]),
    "h2_selftest_sharded_commits": (_U64, []),
    "h2_selftest_set_shard_min_rows": (_I, [_Z]),
    "h2_selftest_msm_check": (_I, [_I, _Z, _Z, _Z, _Z, _I, _P]),
    "h2_selftest_msm_tiles": (_I, [_U32, _U32]),
    "h2_selftest_msm_guard": (_I, [_I]),
    "h2_selftest_msm_guard_report": (_I, [_P, _P, _Z]),
    "h2_selftest_arena_stats": (_I, [_P]),
}

# the RNG callback of the product surface: void (*)(void* ctx, uint8_t* out, size_t n)
RNG_FILL = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)

_lib = None


class H2Error(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        detail = ""
        try:
            msg = load().h2_strerror(status).decode()
            dev = load().h2_last_device_error().decode()
            detail = msg + (" [" + dev + "]" if dev and status == -3 else "")
        except Exception:  # pragma: no cover
            pass
        super().__init__("%s failed: %s (%s)" % (where, STATUS_NAMES.get(status, status), detail))


def load():
    """Load libh2hip.so; raises (never falls back) when the HIP extension is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "halo2_prover_amd: %s not found -- build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.  When PyTorch is installed, load it
        # first so that libh2hip.so binds to the same copy (two runtimes in one process do not see the GPU).
        try:
            import torch  # noqa: F401
        except ImportError:  # the C ABI itself does not need PyTorch (e.g. a Rust host)
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(SELFTEST_SYMBOLS.items()):
            fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status, where):
    if status != H2_OK:
        raise H2Error(status, where)
