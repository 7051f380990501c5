"""Host-side mirror of the reference's interface for the MSM/NTT hot path.

The reference is Rust and there is no Rust toolchain in this image, so the host side above
the C ABI is written in Python with the reference's names, argument meaning and error
behaviour:

    halo2_proofs::arithmetic::best_multiexp(coeffs, bases) -> C::Curve
    halo2_proofs::arithmetic::best_fft(a, omega, log_n)            (in place)
    ParamsKZG::{read, write, k, n, get_g, commit, commit_lagrange}
        (constructed at /root/reference/circuits/src/utils.rs:59-61, read at wasm.rs:79-80)

Arrays are numpy uint64 in halo2curves' in-memory layout: scalars (n, 4), affine points
(n, 8), Jacobian points (12,), Montgomery form.  Length mismatches raise ValueError where the
reference panics (assert_eq!).  Everything computes on the GPU through libh2hip.so; there is
no CPU fallback.
"""
import ctypes
import os

import numpy as np

from . import lib as _lib

_initialised = False


def init(device=None):
    """Bind this process to one GPU (default: LOCAL_RANK, else 0)."""
    global _initialised
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    _lib.check(_lib.load().h2_init(int(device)), "h2_init")
    _initialised = True


def _ensure_init():
    if not _initialised:
        init()


def _curve_id(curve):
    if isinstance(curve, str):
        return _lib.CURVES[curve]
    return int(curve)


def _as_u64(a, cols, name):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        if a.size % cols:
            raise ValueError("%s: size %d is not a multiple of %d limbs" % (name, a.size, cols))
        a = a.reshape(-1, cols)
    if a.ndim != 2 or a.shape[1] != cols:
        raise ValueError("%s: expected shape (n, %d), got %r" % (name, cols, a.shape))
    return a


class Bases:
    """A registered set of affine bases kept resident in HBM (an SRS vector g or g_lagrange)."""

    def __init__(self, curve, affine):
        _ensure_init()
        self.curve = _curve_id(curve)
        affine = _as_u64(affine, 8, "bases")
        if affine.shape[0] == 0:
            raise ValueError("bases: empty")
        self.n = affine.shape[0]
        h = ctypes.c_uint64(0)
        st = _lib.load().h2_bases_register(self.curve, affine.ctypes.data, self.n, ctypes.byref(h))
        _lib.check(st, "h2_bases_register")
        self.handle = h.value

    @classmethod
    def from_device(cls, curve, d_ptr, n):
        _ensure_init()
        self = cls.__new__(cls)
        self.curve = _curve_id(curve)
        self.n = int(n)
        h = ctypes.c_uint64(0)
        st = _lib.load().h2_bases_register_device(self.curve, ctypes.c_void_p(d_ptr), self.n, ctypes.byref(h))
        _lib.check(st, "h2_bases_register_device")
        self.handle = h.value
        return self

    def plan(self):
        p = _lib.MsmPlan()
        _lib.check(_lib.load().h2_msm_plan(self.handle, ctypes.byref(p)), "h2_msm_plan")
        return {"window_bits": p.window_bits, "windows": p.windows, "buckets": p.buckets,
                "table_bytes": p.table_bytes}

    def release(self):
        if getattr(self, "handle", 0):
            _lib.load().h2_bases_release(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # ---- MSM -------------------------------------------------------------------------------
    def msm(self, scalars):
        """sum_i scalars[i] * bases[i] -> Jacobian (12,) ; len(scalars) <= len(bases)."""
        scalars = _as_u64(scalars, 4, "coeffs")
        n = scalars.shape[0]
        if n > self.n:
            raise ValueError("coeffs longer than bases (%d > %d)" % (n, self.n))
        out = np.zeros(12, dtype=np.uint64)
        st = _lib.load().h2_msm(self.curve, self.handle, scalars.ctypes.data if n else None, n, out.ctypes.data)
        _lib.check(st, "h2_msm")
        return out

    def msm_batch(self, columns):
        """m columns of equal length -> (m, 8) normalised affine commitments."""
        cols = [_as_u64(c, 4, "column") for c in columns]
        m = len(cols)
        if m == 0:
            return np.zeros((0, 8), dtype=np.uint64)
        n = cols[0].shape[0]
        if any(c.shape[0] != n for c in cols):
            raise ValueError("columns of unequal length")
        if n > self.n:
            raise ValueError("coeffs longer than bases (%d > %d)" % (n, self.n))
        ptrs = (ctypes.c_void_p * m)(*[c.ctypes.data for c in cols])
        out = np.zeros((m, 8), dtype=np.uint64)
        st = _lib.load().h2_msm_batch(self.curve, self.handle, ptrs, n, m, out.ctypes.data)
        _lib.check(st, "h2_msm_batch")
        return out

    def msm_device(self, d_scalars, n, m, d_out_jac, stream=0):
        """Device-resident: m columns (stride n*32 B) -> m Jacobian points; asynchronous."""
        st = _lib.load().h2_msm_device(self.curve, self.handle, ctypes.c_void_p(d_scalars), n, m,
                                       ctypes.c_void_p(d_out_jac), ctypes.c_void_p(stream))
        _lib.check(st, "h2_msm_device")


    def msm_device_range(self, d_scalars, first_base, n, col_stride, m, d_out_jac, stream=0):
        """m columns (col_stride elements apart) against bases[first_base : first_base + n] -> m Jacobian points"""
        st = _lib.load().h2_msm_device_range(self.curve, self.handle, ctypes.c_void_p(d_scalars), first_base, n,
                                             col_stride, m, ctypes.c_void_p(d_out_jac), ctypes.c_void_p(stream))
        _lib.check(st, "h2_msm_device_range")

    def points_sum_device(self, d_in_jac, groups, count, d_out_jac, stream=0):
        """out[j] = sum_g in[g * count + j]: adds the all-gathered partial sums of a range-split MSM"""
        st = _lib.load().h2_points_sum_device(self.curve, ctypes.c_void_p(d_in_jac), groups, count,
                                              ctypes.c_void_p(d_out_jac), ctypes.c_void_p(stream))
        _lib.check(st, "h2_points_sum_device")


def msm_device_multi(bases_list, d_scalars, first_base, n, col_stride, d_out_jac, stream=0):
    """m = len(bases_list) columns in ONE launch sequence, column j against bases_list[j] (same length and curve):
    commitments that do not wait for each other although they use different SRS vectors"""
    m = len(bases_list)
    handles = (ctypes.c_uint64 * m)(*[b.handle for b in bases_list])
    st = _lib.load().h2_msm_device_multi(bases_list[0].curve, handles, ctypes.c_void_p(d_scalars), first_base, n, col_stride,
                                         m, ctypes.c_void_p(d_out_jac), ctypes.c_void_p(stream))
    _lib.check(st, "h2_msm_device_multi")


def init_devices(device_ids):
    """One process, several GPUs (h2_init_devices): host-pointer batches are then sharded over the devices."""
    global _initialised
    ids = (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
    _lib.check(_lib.load().h2_init_devices(len(device_ids), ids), "h2_init_devices")
    _initialised = True


def best_multiexp(coeffs, bases, curve="bn254"):
    """best_multiexp(coeffs, bases) -> Jacobian point (12 limbs).  One-shot form: registers the
    bases, runs the MSM and releases them (use `Bases` / `ParamsKZG` to keep an SRS resident)."""
    coeffs = _as_u64(coeffs, 4, "coeffs")
    bases = _as_u64(bases, 8, "bases")
    if coeffs.shape[0] != bases.shape[0]:
        raise ValueError("best_multiexp: coeffs.len() != bases.len() (%d vs %d)" % (coeffs.shape[0], bases.shape[0]))
    if coeffs.shape[0] == 0:
        return np.zeros(12, dtype=np.uint64)
    b = Bases(curve, bases)
    try:
        return b.msm(coeffs)
    finally:
        b.release()


def best_fft(a, omega, log_n, curve="bn254"):
    """best_fft(a, omega, log_n): in-place NTT of a (n, 4) uint64 array over the curve's scalar
    field; natural order in and out, unscaled."""
    _ensure_init()
    if not isinstance(a, np.ndarray) or a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"]:
        raise ValueError("best_fft: a must be a C-contiguous numpy uint64 array (it is transformed in place)")
    if a.size != 4 << log_n:
        raise ValueError("best_fft: a.len() != 1 << log_n (%d vs %d)" % (a.size // 4, 1 << log_n))
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    st = _lib.load().h2_ntt(_curve_id(curve), a.ctypes.data, omega.ctypes.data, log_n)
    _lib.check(st, "h2_ntt")
    return a


def best_fft_group(points_jac, omega, log_n, curve="bn254"):
    """best_fft over group elements (G = C::Curve, the FftGroup impl g_to_lagrange uses): in-place transform of a
    (n, 12) uint64 array of Jacobian points; natural order in and out, unscaled."""
    _ensure_init()
    a = points_jac
    if not isinstance(a, np.ndarray) or a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"]:
        raise ValueError("best_fft_group: points must be a C-contiguous numpy uint64 array (transformed in place)")
    if a.size != 12 << log_n:
        raise ValueError("best_fft_group: a.len() != 1 << log_n (%d vs %d)" % (a.size // 12, 1 << log_n))
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    st = _lib.load().h2_fft_group(_curve_id(curve), a.ctypes.data, omega.ctypes.data, log_n)
    _lib.check(st, "h2_fft_group")
    return a


def best_fft_batch(columns, omega, log_n, curve="bn254"):
    """The same transform over m independent columns in one launch sequence."""
    _ensure_init()
    for c in columns:
        if not isinstance(c, np.ndarray) or c.dtype != np.uint64 or not c.flags["C_CONTIGUOUS"]:
            raise ValueError("best_fft_batch: columns must be C-contiguous numpy uint64 arrays")
        if c.size != 4 << log_n:
            raise ValueError("best_fft_batch: column length != 1 << log_n")
    m = len(columns)
    if m == 0:
        return columns
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    ptrs = (ctypes.c_void_p * m)(*[c.ctypes.data for c in columns])
    st = _lib.load().h2_ntt_batch(_curve_id(curve), ptrs, m, omega.ctypes.data, log_n)
    _lib.check(st, "h2_ntt_batch")
    return columns


def ntt_device(d_ptr, m, omega, log_n, curve="bn254", stream=0):
    """Device-resident NTT of m columns (stride (1<<log_n)*32 B), in place, asynchronous."""
    _ensure_init()
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    st = _lib.load().h2_ntt_device(_curve_id(curve), ctypes.c_void_p(d_ptr), m, omega.ctypes.data, log_n,
                                   ctypes.c_void_p(stream))
    _lib.check(st, "h2_ntt_device")


class ParamsKZG:
    """Mirror of halo2_proofs::poly::kzg::commitment::ParamsKZG<Bn256> for the prover side.

    Wire format (written at /root/reference/circuits/src/wasm.rs:52, read at
    wasm.rs:79-80): k:u32 LE || g[0..n) || g_lagrange[0..n) || g2 || s_g2, G1 points as raw
    Montgomery limbs (64 B), the 256-byte G2 tail kept opaque.
    """

    def __init__(self, k, g, g_lagrange, g2_tail=b""):
        self.k = int(k)
        self.n = 1 << self.k
        self.g = _as_u64(g, 8, "g")
        self.g_lagrange = _as_u64(g_lagrange, 8, "g_lagrange")
        if self.g.shape[0] != self.n or self.g_lagrange.shape[0] != self.n:
            raise ValueError("ParamsKZG: g / g_lagrange length != 1 << k")
        self.g2_tail = bytes(g2_tail)
        self._g = Bases("bn254", self.g)
        self._gl = Bases("bn254", self.g_lagrange)

    @classmethod
    def read(cls, data):
        data = bytes(data)
        if len(data) < 4:
            raise ValueError("ParamsKZG.read: truncated")
        k = int.from_bytes(data[:4], "little")
        if k > 28:
            raise ValueError("ParamsKZG.read: k=%d out of range" % k)
        n = 1 << k
        need = 4 + 128 * n + 256
        if len(data) != need:
            raise ValueError("ParamsKZG.read: expected %d bytes for k=%d, got %d" % (need, k, len(data)))
        g = np.frombuffer(data, dtype=np.uint64, count=8 * n, offset=4).reshape(n, 8).copy()
        gl = np.frombuffer(data, dtype=np.uint64, count=8 * n, offset=4 + 64 * n).reshape(n, 8).copy()
        return cls(k, g, gl, data[4 + 128 * n:])

    def write(self):
        return self.k.to_bytes(4, "little") + self.g.tobytes() + self.g_lagrange.tobytes() + self.g2_tail

    def get_g(self):
        return self.g

    def commit(self, poly):
        """commit(poly in coefficient basis, blind ignored for KZG) = best_multiexp(poly, g)."""
        poly = _as_u64(poly, 4, "poly")
        if poly.shape[0] != self.n:
            raise ValueError("commit: poly.len() != n")
        return self._g.msm(poly)

    def commit_lagrange(self, poly):
        """commit_lagrange(poly in Lagrange basis) = best_multiexp(poly, g_lagrange)."""
        poly = _as_u64(poly, 4, "poly")
        if poly.shape[0] != self.n:
            raise ValueError("commit_lagrange: poly.len() != n")
        return self._gl.msm(poly)

    def commit_many(self, polys, lagrange):
        """m columns of one proof phase in one batched launch -> (m, 8) affine commitments."""
        return (self._gl if lagrange else self._g).msm_batch(polys)
