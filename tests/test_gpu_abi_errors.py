"""GPU: error behaviour of the C ABI.  The reference panics (assert_eq! on lengths, .expect()); the ABI must return
status codes and stay usable afterwards -- never abort or throw across the boundary."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
EINVAL, EHANDLE = -1, -4


def test_status_codes_and_recovery(h2):
    L = h2.load()
    n = 64
    b = O.synth_bases(0, 0x48324D53000000B5, n).reshape(n, 8)
    s = O.synth_scalars(1, 0x48324D5300000001, n).reshape(n, 4)
    out = np.zeros(12, dtype=np.uint64)
    h = ctypes.c_uint64(0)
    # registration: unknown curve, null pointer, n = 0
    assert L.h2_bases_register(9, b.ctypes.data, n, ctypes.byref(h)) == EINVAL
    assert L.h2_bases_register(0, None, n, ctypes.byref(h)) == EINVAL
    assert L.h2_bases_register(0, b.ctypes.data, 0, ctypes.byref(h)) == EINVAL
    assert L.h2_bases_register(0, b.ctypes.data, n, ctypes.byref(h)) == 0
    handle = h.value
    assert L.h2_bases_len(handle) == n
    # MSM: wrong curve for the handle, more scalars than bases, null out, bad handle
    assert L.h2_msm(1, handle, s.ctypes.data, n, out.ctypes.data) == EINVAL
    assert L.h2_msm(0, handle, s.ctypes.data, n + 1, out.ctypes.data) == EINVAL
    assert L.h2_msm(0, handle, s.ctypes.data, n, None) == EINVAL
    assert L.h2_msm(0, handle + 12345, s.ctypes.data, n, out.ctypes.data) == EHANDLE
    assert L.h2_msm_batch(0, handle, None, n, 1, out.ctypes.data) == EINVAL
    # n = 0 -> the identity
    out[:] = 7
    assert L.h2_msm(0, handle, None, 0, out.ctypes.data) == 0 and not out.any()
    # NTT: null, log_n too large, unknown curve
    w = np.zeros(4, dtype=np.uint64)
    assert L.h2_ntt(0, None, w.ctypes.data, 3) == EINVAL
    assert L.h2_ntt(0, s.ctypes.data, None, 3) == EINVAL
    assert L.h2_ntt(0, s.ctypes.data, w.ctypes.data, 31) == EINVAL
    assert L.h2_ntt(5, s.ctypes.data, w.ctypes.data, 3) == EINVAL
    assert L.h2_poly_mul_periodic_device(0, ctypes.c_void_p(1), 8, 1, ctypes.c_void_p(1), 3, None) == EINVAL  # period not 2^j
    # still healthy: a correct MSM after all the refusals
    assert L.h2_msm(0, handle, s.ctypes.data, n, out.ctypes.data) == 0
    assert np.array_equal(O.to_affine(0, out), O.to_affine(0, O.best_multiexp(0, s, b)))
    # release, double release, use after release
    assert L.h2_bases_release(handle) == 0
    assert L.h2_bases_release(handle) == EHANDLE
    assert L.h2_msm(0, handle, s.ctypes.data, n, out.ctypes.data) == EHANDLE
    assert L.h2_bases_len(handle) == EHANDLE
    assert L.h2_init(0) == 0                       # idempotent for the same device
    assert L.h2_init(1 << 20) == EINVAL            # a different device in the same process is refused
    assert L.h2_strerror(-4).decode().startswith("unknown bases handle")


def test_python_mirror_raises_where_the_reference_panics(h2):
    b = O.synth_bases(0, 1, 8).reshape(8, 8)
    s = O.synth_scalars(1, 2, 8).reshape(8, 4)
    with pytest.raises(ValueError):
        h2.best_multiexp(s[:7], b, "bn254")                      # assert_eq!(coeffs.len(), bases.len())
    with pytest.raises(ValueError):
        h2.best_fft(s.copy(), np.zeros(4, dtype=np.uint64), 4, "bn254")   # assert_eq!(a.len(), 1 << log_n)
    with pytest.raises(ValueError):
        h2.ParamsKZG.read(b"\\x04\\x00\\x00\\x00" + bytes(10))     # truncated params file
    with pytest.raises(KeyError):
        h2.best_multiexp(s, b, "bls12-381")


def test_bases_off_the_curve_are_rejected(h2):
    """h2_bases_register checks every point on the device while it builds the table (the reference reads params with
    SerdeFormat::RawBytes, i.e. with curve checks)"""
    import oracle_lib as O
    n = 300
    b = O.synth_bases(0, 0x48324D53000000E5, n).reshape(n, 8).copy()
    h2.Bases("bn254", b).release()
    bad = b.copy()
    bad[137, 0] ^= np.uint64(1)                                  # x of one point off by one
    with pytest.raises(h2.H2Error) as err:
        h2.Bases("bn254", bad)
    assert err.value.status == -1
    ident = b.copy()
    ident[5] = 0                                                 # the identity (0, 0) is a valid base
    h2.Bases("bn254", ident).release()
    noncanon = b.copy()
    noncanon[9, 0:4] = np.array([0xFFFFFFFFFFFFFFFF] * 4, dtype=np.uint64)   # x >= p
    with pytest.raises(h2.H2Error):
        h2.Bases("bn254", noncanon)
