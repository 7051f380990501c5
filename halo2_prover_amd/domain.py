"""Host mirror of halo2_proofs::poly::EvaluationDomain for device-resident columns.

Upstream (halo2_proofs @6b43b6b, src/poly/domain.rs -- un-vendored; behaviour restated in SURVEY.md App. A.3;
constructed inside keygen / create_proof reached from /root/reference/circuits/src/utils.rs:63-70,83-91,105-120):

    new(j, k)                   quotient_poly_degree = j - 1, extended_k = smallest with 2^extended_k >= n (j-1),
                                omega / extended_omega from ROOT_OF_UNITY, g_coset = ZETA (a cube root of unity)
    lagrange_to_coeff(a)        ifft: best_fft(a, omega^-1, k) then * n^-1
    coeff_to_extended(a)        zero-extend to 2^extended_k, a[i] *= g_coset^i, best_fft(a, extended_omega)
    extended_to_coeff(a)        ifft on the extended domain, a[i] *= g_coset^-i, truncate to n (j-1)
    divide_by_vanishing_poly(a) a[i] *= t_evaluations[i mod 2^(extended_k-k)],  t_i = 1 / ((zeta w_ext^i)^n - 1)

Columns are torch int64 CUDA tensors of shape (..., n, 4) viewing halo2curves' 4 x u64 Montgomery limbs; they
stay in HBM across the calls (PyTorch only provides the device memory and the stream).  The value of ZETA is
(R)ecalled, not verifiable offline: the quotient polynomial is unique whatever coset is
used, so proof bytes do not depend on it.
"""
import ctypes

import numpy as np

from . import lib as _lib
from .api import _curve_id, _ensure_init

_FIELDS = {
    0: (0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 7, 28,
        0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23),   # bn256::Fr, ZETA as recalled
    1: (0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001, 5, 32, None),  # pallas scalar = Fq
    2: (0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001, 5, 32, None),  # vesta scalar = Fp
}


def _limbs(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class EvaluationDomain:
    def __init__(self, j, k, curve="bn254"):
        import torch
        _ensure_init()
        self.curve = _curve_id(curve)
        p, gen, S, zeta = _FIELDS[self.curve]
        self.p, self.R = p, (1 << 256) % p
        self.k, self.n = k, 1 << k
        self.quotient_poly_degree = j - 1
        self.extended_k = k
        while (1 << self.extended_k) < self.n * self.quotient_poly_degree:
            self.extended_k += 1
        if self.extended_k > S:
            raise ValueError("extended_k exceeds the field's two-adicity")
        root = pow(gen, (p - 1) >> S, p)
        self.extended_omega = pow(root, 1 << (S - self.extended_k), p)
        self.omega = pow(self.extended_omega, 1 << (self.extended_k - k), p)
        self.omega_inv = pow(self.omega, -1, p)
        self.extended_omega_inv = pow(self.extended_omega, -1, p)
        self.g_coset = zeta if zeta is not None else pow(gen, (p - 1) // 3, p)
        assert pow(self.g_coset, 3, p) == 1 and self.g_coset != 1
        self.g_coset_inv = self.g_coset * self.g_coset % p
        self.ifft_divisor = pow(self.n, -1, p)
        self.extended_ifft_divisor = pow(1 << self.extended_k, -1, p)
        period = 1 << (self.extended_k - k)
        t = [pow((pow(self.g_coset * pow(self.extended_omega, i, p) % p, self.n, p) - 1) % p, -1, p)
             for i in range(period)]
        self.t_evaluations_int = t
        tv = np.stack([_limbs(x * self.R % p) for x in t])
        self.t_evaluations = torch.from_numpy(tv.view(np.int64)).cuda()
        self._L = _lib.load()
        # limb arrays handed to the C ABI must outlive the call: keep them as attributes
        self._m = {name: self.mont(getattr(self, name)) for name in
                   ("omega", "omega_inv", "extended_omega", "extended_omega_inv", "g_coset", "g_coset_inv",
                    "ifft_divisor", "extended_ifft_divisor")}

    # ---- helpers -------------------------------------------------------------------------------
    def mont(self, v):
        return _limbs(v % self.p * self.R % self.p)

    def _stream(self):
        """the caller's current stream as a raw handle (the library orders its launches on it)"""
        import torch
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # ~0.3 us against ~9 us for the object
        if raw is not None:
            return ctypes.c_void_p(raw(torch.cuda.current_device()))
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _shape(a, n):
        if a.dtype.__str__() != "torch.int64" or not a.is_cuda or not a.is_contiguous():
            raise ValueError("columns must be contiguous int64 CUDA tensors")
        if a.shape[-1] != 4 or a.shape[-2] != n:
            raise ValueError("expected shape (..., %d, 4), got %r" % (n, tuple(a.shape)))
        return int(a.numel() // (4 * n))

    def to_device(self, arr):
        import torch
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)).cuda()

    # ---- EvaluationDomain methods --------------------------------------------------------------
    def lagrange_to_coeff(self, a):
        """in place: Lagrange basis -> coefficients (ifft with the n^-1 scaling fused into the last NTT pass)"""
        m = self._shape(a, self.n)
        st = self._L.h2_ntt_scaled_device(self.curve, ctypes.c_void_p(a.data_ptr()), m, self._m["omega_inv"].ctypes.data,
                                          self.k, self._m["ifft_divisor"].ctypes.data, self._stream())
        _lib.check(st, "h2_ntt_scaled_device")
        return a

    def coeff_to_lagrange(self, a):
        m = self._shape(a, self.n)
        st = self._L.h2_ntt_device(self.curve, ctypes.c_void_p(a.data_ptr()), m, self._m["omega"].ctypes.data, self.k,
                                   self._stream())
        _lib.check(st, "h2_ntt_device")
        return a

    def coeff_to_extended(self, a):
        """returns a new (..., 2^extended_k, 4) tensor: evaluations of a on the coset g * <extended_omega>"""
        import torch
        m = self._shape(a, self.n)
        en = 1 << self.extended_k
        out = torch.zeros(a.shape[:-2] + (en, 4), dtype=torch.int64, device=a.device)
        out[..., : self.n, :] = a
        ptr = ctypes.c_void_p(out.data_ptr())
        _lib.check(self._L.h2_poly_coset_device(self.curve, ptr, en, m, self._m["g_coset"].ctypes.data, self._stream()),
                   "h2_poly_coset_device")
        _lib.check(self._L.h2_ntt_device(self.curve, ptr, m, self._m["extended_omega"].ctypes.data, self.extended_k,
                                         self._stream()), "h2_ntt_device")
        return out

    def extended_to_coeff(self, a):
        """extended-coset evaluations -> the n*(j-1) coefficients (new tensor); a is overwritten"""
        en = 1 << self.extended_k
        m = self._shape(a, en)
        ptr = ctypes.c_void_p(a.data_ptr())
        _lib.check(self._L.h2_ntt_scaled_device(self.curve, ptr, m, self._m["extended_omega_inv"].ctypes.data,
                                                self.extended_k, self._m["extended_ifft_divisor"].ctypes.data,
                                                self._stream()), "h2_ntt_scaled_device")
        _lib.check(self._L.h2_poly_coset_device(self.curve, ptr, en, m, self._m["g_coset_inv"].ctypes.data,
                                                self._stream()), "h2_poly_coset_device")
        return a[..., : self.n * self.quotient_poly_degree, :].contiguous()

    def divide_by_vanishing_poly(self, a):
        en = 1 << self.extended_k
        m = self._shape(a, en)
        st = self._L.h2_poly_mul_periodic_device(self.curve, ctypes.c_void_p(a.data_ptr()), en, m,
                                                 ctypes.c_void_p(self.t_evaluations.data_ptr()),
                                                 1 << (self.extended_k - self.k), self._stream())
        _lib.check(st, "h2_poly_mul_periodic_device")
        return a

    def pointwise(self, op, a, b):
        """a = a (op) b elementwise, op in {'add', 'sub', 'mul'}"""
        if a.shape != b.shape:
            raise ValueError("shape mismatch")
        code = {"add": 0, "sub": 1, "mul": 2}[op]
        st = self._L.h2_poly_pointwise_device(self.curve, code, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                              a.numel() // 4, self._stream())
        _lib.check(st, "h2_poly_pointwise_device")
        return a

    def scale(self, a, c):
        n = a.shape[-2]
        m = a.numel() // (4 * n)
        cm = self.mont(c)
        st = self._L.h2_poly_scale_device(self.curve, ctypes.c_void_p(a.data_ptr()), n, m, cm.ctypes.data,
                                          self._stream())
        _lib.check(st, "h2_poly_scale_device")
        return a
