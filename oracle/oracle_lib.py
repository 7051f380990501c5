"""ctypes binding of oracle/libh2oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never
from halo2_prover_amd/.  Arrays are numpy uint64 in the reference's in-memory layout
(4 LE limbs per field element, Montgomery form; affine = 8 limbs; Jacobian = 12 limbs).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FIELD_IDS = {"bn254_fq": 0, "bn254_fr": 1, "pasta_fp": 2, "pasta_fq": 3}
CURVE_IDS = {"bn254": 0, "pallas": 1, "vesta": 2}
CURVE_SCALAR_FIELD = {0: 1, 1: 3, 2: 2}
CURVE_BASE_FIELD = {0: 0, 1: 2, 2: 3}


def build(force=False):
    so = os.path.join(_HERE, "libh2oracle.so")
    src = os.path.join(_HERE, "h2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libh2oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libh2oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        P, Z, I, U = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
        U64 = ctypes.c_uint64
        sigs = {
            "h2o_best_multiexp": [I, P, P, Z, I, P],
            "h2o_best_fft": [I, P, P, U, I],
            "h2o_group_fft": [I, P, P, U],
            "h2o_field_op": [I, I, P, P, P],
            "h2o_field_mul_many": [I, P, P, Z, P],
            "h2o_eval_polynomial": [I, P, Z, P, P],
            "h2o_to_affine": [I, P, Z, P],
            "h2o_is_on_curve": [I, P, Z],
            "h2o_scalar_mul": [I, P, P, P],
            "h2o_jac_add": [I, P, P, P],
            "h2o_powers_of_s": [I, P, Z, I, P],
            "h2o_scale_points": [I, P, P, Z],
            "h2o_synth_scalars": [I, U64, Z, P],
            "h2o_synth_bases": [I, U64, Z, I, P],
        }
        for name, args in sigs.items():
            fn = getattr(_LIB, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def int_to_limbs(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def limbs_to_int(a):
    return sum(int(x) << (64 * i) for i, x in enumerate(a))


def best_multiexp(cid, coeffs, bases, threads=1):
    coeffs, bases = _u64(coeffs), _u64(bases)
    n = coeffs.size // 4
    assert bases.size == n * 8
    out = np.zeros(12, dtype=np.uint64)
    assert lib().h2o_best_multiexp(cid, _p(coeffs), _p(bases), n, threads, _p(out)) == 0
    return out


def best_fft(fid, a, omega, log_n, threads=1):
    a = _u64(a).reshape(-1).copy()
    omega = _u64(omega)
    assert a.size == 4 << log_n
    assert lib().h2o_best_fft(fid, _p(a), _p(omega), log_n, threads) == 0
    return a


def group_fft(cid, pts_jac, omega, log_n):
    a = _u64(pts_jac).copy()
    omega = _u64(omega)
    assert lib().h2o_group_fft(cid, _p(a), _p(omega), log_n) == 0
    return a


def field_op(fid, op, a, b=None):
    ops = {"add": 0, "sub": 1, "mul": 2, "inv": 3, "to_mont": 4, "from_mont": 5, "neg": 6}
    a = _u64(a)
    b = _u64(b) if b is not None else a
    out = np.zeros(4, dtype=np.uint64)
    assert lib().h2o_field_op(fid, ops[op], _p(a), _p(b), _p(out)) == 0
    return out


def field_mul_many(fid, a, b):
    a, b = _u64(a), _u64(b)
    out = np.zeros_like(a)
    assert lib().h2o_field_mul_many(fid, _p(a), _p(b), a.size // 4, _p(out)) == 0
    return out


def eval_polynomial(fid, coeffs, x_mont):
    """sum_i coeffs[i] x^i (Montgomery limbs in and out)."""
    coeffs, x = _u64(coeffs), _u64(x_mont)
    out = np.zeros(4, dtype=np.uint64)
    assert lib().h2o_eval_polynomial(fid, _p(coeffs), coeffs.size // 4, _p(x), _p(out)) == 0
    return out


def to_affine(cid, jac):
    jac = _u64(jac)
    n = jac.size // 12
    out = np.zeros(n * 8, dtype=np.uint64)
    assert lib().h2o_to_affine(cid, _p(jac), n, _p(out)) == 0
    return out


def is_on_curve(cid, aff):
    aff = _u64(aff)
    return lib().h2o_is_on_curve(cid, _p(aff), aff.size // 8) == 1


def scalar_mul(cid, k_mont, aff):
    out = np.zeros(12, dtype=np.uint64)
    assert lib().h2o_scalar_mul(cid, _p(_u64(k_mont)), _p(_u64(aff)), _p(out)) == 0
    return out


def jac_add(cid, p, q):
    out = np.zeros(12, dtype=np.uint64)
    assert lib().h2o_jac_add(cid, _p(_u64(p)), _p(_u64(q)), _p(out)) == 0
    return out


def powers_of_s(cid, s_mont, n, threads=8):
    out = np.zeros(n * 12, dtype=np.uint64)
    assert lib().h2o_powers_of_s(cid, _p(_u64(s_mont)), n, threads, _p(out)) == 0
    return out


def scale_points(cid, k_mont, pts):
    pts = _u64(pts).copy()
    assert lib().h2o_scale_points(cid, _p(_u64(k_mont)), _p(pts), pts.size // 12) == 0
    return pts


def synth_scalars(fid, seed, n):
    out = np.zeros(n * 4, dtype=np.uint64)
    assert lib().h2o_synth_scalars(fid, seed, n, _p(out)) == 0
    return out


def synth_bases(cid, seed, n, threads=8):
    out = np.zeros(n * 8, dtype=np.uint64)
    assert lib().h2o_synth_bases(cid, seed, n, threads, _p(out)) == 0
    return out
