"""CPU (no GPU): the HOST instantiation of the library's field / curve templates against the oracle,
and the C-ABI surface: libh2hip.so loads and exports every symbol include/*.h declares."""
import os
import random
import re

import numpy as np
import pytest

import oracle_lib as O
import pyref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import halo2_prover_amd
    return halo2_prover_amd.load()


def test_every_declared_symbol_is_exported(lib):
    import halo2_prover_amd
    declared = set()
    for hdr in ("h2hip.h", "h2hip_selftest.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        declared |= set(re.findall(r"\b(h2_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libh2hip.so does not export " + name
    assert declared == set(halo2_prover_amd.SYMBOLS) | set(halo2_prover_amd.lib.SELFTEST_SYMBOLS)


def test_compute_calls_fail_loudly_without_init(lib):
    """No GPU / no h2_init: the product path reports an error instead of falling back."""
    out = np.zeros(12, dtype=np.uint64)
    s = np.zeros(4, dtype=np.uint64)
    st = lib.h2_msm(0, 1, s.ctypes.data, 1, out.ctypes.data)
    assert st in (-5, -4)  # H2_ENOTINIT (or H2_EHANDLE if another test initialised a device)
    assert lib.h2_strerror(-5).decode().startswith("h2_init")
    assert lib.h2_version() >= 1000


def _fop(lib, fid, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    assert lib.h2_selftest_field_op(fid, op, a.ctypes.data, b.ctypes.data, out.ctypes.data) == 0
    return out


@pytest.mark.parametrize("name", list(R.FIELDS))
def test_host_field_ops_match_oracle(lib, name):
    f = R.FIELDS[name]
    fid = O.FIELD_IDS[name]
    rng = random.Random(fid)
    special = [0, 1, 2, f.p - 1, f.p - 2, (1 << 255) % f.p, (1 << 32) - 1, 1 << 32, (1 << 64) - 1, 1 << 224]
    vals = special + [rng.randrange(f.p) for _ in range(150)]
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        am = np.array(f.limbs(a), dtype=np.uint64)
        bm = np.array(f.limbs(b), dtype=np.uint64)
        for op, name_ in ((0, "add"), (1, "sub"), (2, "mul"), (9, "mul"), (10, "add"), (11, "sub")):
            # ops 9-11: the same through the MSM's working form (9 x 29-bit limbs, R' = 2^261, h2_field29.hpp)
            assert np.array_equal(_fop(lib, fid, op, am, bm), O.field_op(fid, name_, am, bm)), (op, name_, a, b)
        chain = ((a - b) ** 2 - a * b - 2 * b) % f.p                     # op 12: a lazy chain, normalised once
        assert O.limbs_to_int(_fop(lib, fid, 12, am, bm)) == f.to_mont(chain), (a, b)
        assert np.array_equal(_fop(lib, fid, 6, am), O.field_op(fid, "neg", am))
        assert np.array_equal(_fop(lib, fid, 5, am), O.field_op(fid, "from_mont", am))
        assert np.array_equal(_fop(lib, fid, 4, am), O.field_op(fid, "to_mont", am))
    for a in vals[1:40]:
        am = np.array(f.limbs(a), dtype=np.uint64)
        assert O.limbs_to_int(_fop(lib, fid, 3, am)) == f.to_mont(pow(a, -1, f.p))


def _cop(lib, cid, op, p, q):
    out = np.zeros(8, dtype=np.uint64)
    p = np.ascontiguousarray(p, dtype=np.uint64)
    q = np.ascontiguousarray(q, dtype=np.uint64)
    assert lib.h2_selftest_curve_op(cid, op, p.ctypes.data, q.ctypes.data, out.ctypes.data) == 0
    return out


def _aff(c, P):
    return np.frombuffer(c.affine_bytes(P), dtype=np.uint64).copy()


@pytest.mark.parametrize("name", list(R.CURVES))
def test_host_curve_ops_match_bigint_reference(lib, name):
    c = R.CURVES[name]
    cid = O.CURVE_IDS[name]
    G = c.gen
    P = c.mul(0x1234567, G)
    Q = c.mul(0x7654321, G)
    ident = np.zeros(8, dtype=np.uint64)
    assert np.array_equal(_cop(lib, cid, 0, _aff(c, P), _aff(c, Q)), _aff(c, c.add(P, Q)))
    assert np.array_equal(_cop(lib, cid, 0, _aff(c, P), _aff(c, P)), _aff(c, c.add(P, P)))          # P + P
    assert np.array_equal(_cop(lib, cid, 0, _aff(c, P), _aff(c, c.neg(P))), ident)                   # P - P
    assert np.array_equal(_cop(lib, cid, 0, ident, _aff(c, Q)), _aff(c, Q))                          # O + Q
    assert np.array_equal(_cop(lib, cid, 0, _aff(c, P), ident), _aff(c, P))                          # P + O
    assert np.array_equal(_cop(lib, cid, 1, _aff(c, P), ident), _aff(c, c.add(P, P)))
    assert np.array_equal(_cop(lib, cid, 2, _aff(c, P), _aff(c, Q)), _aff(c, c.add(c.add(P, Q), Q)))
    assert np.array_equal(_cop(lib, cid, 2, _aff(c, Q), _aff(c, Q)), _aff(c, c.mul(3, Q)))           # doubling inside add
    for k in (1, 2, 3, 0xFFFF, 0x10001, 0xFFFFFFFF):
        kq = np.zeros(8, dtype=np.uint64)
        kq[0] = k
        assert np.array_equal(_cop(lib, cid, 3, _aff(c, P), kq), _aff(c, c.mul(k, P))), k
    # ops 10-13: the same four operations on the MSM's working representation (h2_curve29.hpp)
    assert np.array_equal(_cop(lib, cid, 10, _aff(c, P), _aff(c, Q)), _aff(c, c.add(P, Q)))
    assert np.array_equal(_cop(lib, cid, 10, _aff(c, P), _aff(c, P)), _aff(c, c.add(P, P)))
    assert np.array_equal(_cop(lib, cid, 10, _aff(c, P), _aff(c, c.neg(P))), ident)
    assert np.array_equal(_cop(lib, cid, 10, ident, _aff(c, Q)), _aff(c, Q))
    assert np.array_equal(_cop(lib, cid, 10, _aff(c, P), ident), _aff(c, P))
    assert np.array_equal(_cop(lib, cid, 11, _aff(c, P), ident), _aff(c, c.add(P, P)))
    assert np.array_equal(_cop(lib, cid, 12, _aff(c, P), _aff(c, Q)), _aff(c, c.add(c.add(P, Q), Q)))
    assert np.array_equal(_cop(lib, cid, 12, _aff(c, Q), _aff(c, Q)), _aff(c, c.mul(3, Q)))
    for k in (1, 2, 3, 0xFFFF, 0x10001, 0xFFFFFFFF, 0xAAAAAAAA, 0x80000001):
        kq = np.zeros(8, dtype=np.uint64)
        kq[0] = k
        assert np.array_equal(_cop(lib, cid, 13, _aff(c, P), kq), _aff(c, c.mul(k, P))), k
    # group order: [q]G = O  (SURVEY.md section 8(a) asks for this assertion on the Pasta curves)
    assert c.mul(c.scalar.p, G) is None


@pytest.mark.parametrize("name", list(R.CURVES))
def test_signed_digit_decomposition_reconstructs_the_scalar(lib, name):
    """host run of msm_digit_step: sum_w d_w 2^(c w) == scalar, |d| <= 2^(c-1), bucket index in range,
    no carry out of the top window -- for every window size the geometry can choose."""
    c = R.CURVES[name]
    f = c.scalar
    cid = O.CURVE_IDS[name]
    rng = random.Random(cid + 17)
    out = np.zeros(160, dtype=np.uint32)
    for n_geom in (1, 1 << 10, 1 << 11, 1 << 13, 1 << 16, 1 << 17, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
        specials = [0, 1, f.p - 1, f.p - 2, (1 << 253) - 1, (1 << 128) - 1, (1 << 200) - (1 << 13)]
        for trial in range(60):
            v = specials[trial] if trial < len(specials) else rng.randrange(f.p)
            if trial in (10, 11, 12):  # long runs of one bits: raw == 2^width with carry-in
                v = ((1 << 250) - 1) & ~((1 << rng.randrange(1, 200)) - 1)
                v |= 1 << rng.randrange(0, 8)
                v %= f.p
            s = np.array(f.limbs(v), dtype=np.uint64)
            rc = lib.h2_selftest_digits(cid, s.ctypes.data, n_geom, out.ctypes.data, 160)
            assert rc == 0, (n_geom, hex(v), rc)
            cbits, W, B, nbits = (int(x) for x in out[:4])
            offs = [int(x) for x in out[4 + W:4 + 2 * W]]
            widths = [int(x) for x in out[4 + 2 * W:4 + 3 * W]]
            assert B == 1 << (cbits - 1) and nbits == f.num_bits and cbits <= 19
            # balanced, contiguous windows covering nbits + 1 bits
            assert offs[0] == 0 and all(offs[w + 1] == offs[w] + widths[w] for w in range(W - 1))
            assert offs[-1] + widths[-1] == nbits + 1 and max(widths) == cbits and min(widths) >= cbits - 1
            total = 0
            for w in range(W):
                enc = int(out[4 + w])
                mag = enc & 0x7FFFFFFF
                assert mag <= 1 << (widths[w] - 1) <= B
                if enc:
                    assert 1 <= mag
                total += (-mag if enc >> 31 else mag) << offs[w]
            assert total == v, (n_geom, hex(v))
