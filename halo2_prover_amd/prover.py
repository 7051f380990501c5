"""The reference crate's prove() surface on the GPU backend.

Mirrors /root/reference/circuits/src/utils.rs and wasm.rs:
    generate_params(k)                                               utils.rs:59-61   ParamsKZG::new (setup)
    generate_keys(params, circuit)                                   utils.rs:63-70   keygen_vk + keygen_pk
    generate_proof_with_instance(params, pk, circuit, public_input)  utils.rs:95-123  create_proof, KZG + GWC, Blake2b
    generate_proof(params, pk, circuit)                              utils.rs:72-93   create_proof, KZG + SHPLONK
    wasm_generate_proof(params_bytes, json, circuit_index)           wasm.rs:77-122   circuits 0, 1, 2
for the Collatz circuit (collatz.rs), the arithmetic circuit (arithmetic_circuit.rs) and the Poseidon circuit
(poseidon_circuit.rs + Pow5 chip).  Phase order, transcript, RNG schedule, vk digest and the GWC / SHPLONK
openings follow SURVEY.md App. A.4-A.8.

What runs where:
  * every commitment  -> ParamsKZG.commit_many -> h2_msm_batch            (GPU, one launch sequence per phase)
  * Lagrange -> coefficient form of every column -> h2_ntt_scaled_device  (GPU, batched)
  * the quotient h(X): coset NTTs of every column, the gate / permutation expressions with the pointwise kernels,
    divide_by_vanishing_poly, inverse coset NTT                          (GPU, EvaluationDomain)
  * setup: g = [s^i]G and g_lagrange = [L_i(s)]G                         (GPU, h2_srs_generate / h2_fixed_base_mul)
  * evaluations at x (powers-multiply + folding adds), the v-power combinations of the openings, the
    grand-product ratios (per-element inverse kernel) and their running product (prefix-product scan), the
    openings' synthetic divisions, the blinding polynomial's ChaCha20 draws  (GPU, kernels on resident columns)
  * transcript hashing, witness synthesis, the permutation union-find    (host Python, big integers)
There is no CPU fallback for the GPU parts.  Columns stay resident in HBM as 4 x u64 Montgomery limbs; the
conversion from / to canonical integers runs on the device.
"""
import ctypes
import hashlib
import json
import os

import numpy as np

from . import lib as _lib
from .api import ParamsKZG
from . import sharded as _sharded
from .domain import EvaluationDomain

P = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001       # bn256::Fr modulus
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47       # bn256::Fq modulus
R_P = (1 << 256) % P
R_P_INV = pow(R_P, -1, P)
R_Q = (1 << 256) % Q
R_Q_INV = pow(R_Q, -1, Q)
GENERATOR, TWO_ADICITY = 7, 28
ROOT_OF_UNITY = pow(GENERATOR, (P - 1) >> TWO_ADICITY, P)
DELTA = pow(GENERATOR, 1 << TWO_ADICITY, P)


# ------------------------------------------------------------------------------------------- host helpers ----
def _limbs_of(vals):
    """canonical ints -> (n, 4) uint64 Montgomery limbs"""
    buf = b"".join((v % P * R_P % P).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


class _Col(list):
    """A length-n host column of canonical integers, zero except where it was written; it remembers those rows, so
    that uploading it does not have to scan 2^k entries (witness columns use a few dozen rows of 65 536)."""
    __slots__ = ("touched",)

    def __init__(self, n):
        super().__init__([0] * n)
        self.touched = set()

    def __setitem__(self, i, v):
        list.__setitem__(self, i, v)
        if isinstance(i, int):
            self.touched.add(i if i >= 0 else i + len(self))
        else:                                   # a slice: forget the shortcut, column() scans
            self.touched = None


def _horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def _batch_inverse(vals):
    prefix, acc = [], 1
    for v in vals:
        prefix.append(acc)
        acc = acc * v % P
    inv = pow(acc, -1, P)
    out = [0] * len(vals)
    for i in range(len(vals) - 1, -1, -1):
        out[i] = prefix[i] * inv % P
        inv = inv * vals[i] % P
    return out


class OsRng:
    """rand::rngs::OsRng as the reference uses it (utils.rs:89,116): Fr::random draws 8 x next_u64."""

    def fill(self, nbytes):
        return os.urandom(nbytes)

    def fr_random(self, _field=None):
        v = 0
        for i in range(8):
            v |= int.from_bytes(self.fill(8), "little") << (64 * i)
        return v % P


class _Transcript:
    """Blake2bWrite<Vec<u8>, G1Affine, Challenge255<_>> (utils.rs:103-104)"""

    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.bytes = bytearray()

    def common_scalar(self, s):
        self.state.update(b"\x02" + int(s % P).to_bytes(32, "little"))

    def write_scalar(self, s):
        self.common_scalar(s)
        self.bytes += int(s % P).to_bytes(32, "little")

    def write_point(self, pt):
        x, y = pt if pt is not None else (0, 0)
        self.state.update(b"\x01" + x.to_bytes(32, "little") + y.to_bytes(32, "little"))
        enc = bytearray(x.to_bytes(32, "little"))
        enc[31] |= (y & 1) << 6
        self.bytes += enc

    def squeeze_challenge(self):
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % P


# -------------------------------------------------------------------------------------------- expressions ----
# halo2 Expression trees as nested tuples.  Rust operators: a + b -> sum, a - b -> sum(a, neg(b)), a * b -> prod,
# expr * F -> scaled.
def _const(v): return ("const", v % P)
def _adv(qi, col, rot): return ("advice", qi, col, rot)
def _fix(qi, col, rot): return ("fixed", qi, col, rot)
def _sum(a, b): return ("sum", a, b)
def _sub(a, b): return ("sum", a, ("neg", b))
def _prod(a, b): return ("prod", a, b)
def _scaled(a, c): return ("scaled", a, c % P)


def _expr_debug(e):
    t = e[0]
    if t == "const":
        return "Constant(0x%064x)" % e[1]
    if t in ("advice", "fixed", "instance"):
        return "%s { query_index: %d, column_index: %d, rotation: Rotation(%d) }" % (t.capitalize(), e[1], e[2], e[3])
    if t == "neg":
        return "Negated(%s)" % _expr_debug(e[1])
    if t == "sum":
        return "Sum(%s, %s)" % (_expr_debug(e[1]), _expr_debug(e[2]))
    if t == "prod":
        return "Product(%s, %s)" % (_expr_debug(e[1]), _expr_debug(e[2]))
    if t == "scaled":
        return "Scaled(%s, 0x%064x)" % (_expr_debug(e[1]), e[2])
    raise ValueError(t)


# ------------------------------------------------------------------------------------------------ circuits ----
class Circuit:
    name = ""
    num_advice = num_fixed = num_instance = num_selectors = 0
    degree = 3
    permutation_columns = []
    advice_queries = []
    fixed_queries = []
    instance_queries = [(0, 0)]
    constants = []
    gates = []

    def blinding_factors(self):
        per_col = {}
        for col, rot in self.advice_queries:
            per_col.setdefault(col, set()).add(rot)
        return max(3, max(len(v) for v in per_col.values())) + 2


class ArithmeticCircuit(Circuit):
    """arithmetic_circuit.rs: advice l, r, o; fixed sm, sl, sr, so, sc (creation order :196-200); instance PI."""

    name = "arithmetic"
    num_advice, num_fixed, num_instance, num_selectors = 3, 5, 1, 0
    degree = 3
    SM, SL, SR, SO, SC = range(5)
    permutation_columns = [("advice", 0), ("advice", 1), ("advice", 2), ("instance", 0)]
    advice_queries = [(0, 0), (1, 0), (2, 0)]
    fixed_queries = [(1, 0), (2, 0), (3, 0), (0, 0), (4, 0)]

    def __init__(self, x=None, y=None, constant=0):
        self.x, self.y, self.constant = x, y, constant
        l, r, o = _adv(0, 0, 0), _adv(1, 1, 0), _adv(2, 2, 0)
        sl, sr, so, sm, sc = _fix(0, 1, 0), _fix(1, 2, 0), _fix(2, 3, 0), _fix(3, 0, 0), _fix(4, 4, 0)
        # :216  l*sl + r*sr + l*r*sm + (o*so*(-1)) + sc
        self.gates = [_sum(_sum(_sum(_sum(_prod(l, sl), _prod(r, sr)), _prod(_prod(l, r), sm)),
                                _scaled(_prod(o, so), P - 1)), sc)]

    @classmethod
    def from_json(cls, s):
        v = json.loads(s)
        return cls(int(v["x"]), int(v["y"]), int(v["constant"]))

    def public_inputs(self, s):
        v = json.loads(s)
        return [int(v["constant"]), int(v["z"])]            # wasm.rs:93-94

    def synthesize_fixed(self, n):
        cols = [_Col(n) for _ in range(5)]
        for row in range(3):
            cols[self.SM][row] = cols[self.SO][row] = 1
        cols[self.SL][3] = cols[self.SR][3] = cols[self.SO][3] = 1
        return cols

    def synthesize_advice(self, n):
        x, y, c = self.x % P, self.y % P, self.constant % P
        xx, yy = x * x % P, y * y % P
        prod = xx * yy % P
        rows = [(x, x, xx), (y, y, yy), (xx, yy, prod), (prod, c, (prod + c) % P)]
        cols = [_Col(n) for _ in range(3)]
        for i, row in enumerate(rows):
            for j in range(3):
                cols[j][i] = row[j]
        return cols

    def copy_constraints(self):
        a = lambda col, row: (("advice", col), row)  # noqa: E731
        return [(a(0, 0), a(1, 0)), (a(0, 1), a(1, 1)), (a(2, 0), a(0, 2)), (a(2, 1), a(1, 2)), (a(2, 2), a(0, 3)),
                (a(1, 3), (("instance", 0), 0)), (a(2, 3), (("instance", 0), 1))]


class _Grain:
    """Grain LFSR of the Poseidon reference parameter generation (poseidon/primitives/grain.rs:52-137)"""

    def __init__(self, num_bits, t, r_f, r_p):
        bits = []
        for width, value in ((2, 1), (4, 0), (12, num_bits), (12, t), (10, r_f), (10, r_p)):
            bits += [(value >> (width - 1 - i)) & 1 for i in range(width)]
        self.state = bits + [1] * 30
        self.num_bits = num_bits
        for _ in range(160):
            self._raw()

    def _raw(self):
        s = self.state
        b = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        self.state = s[1:] + [b]
        return b

    def _bit(self):
        while True:
            if self._raw():
                return self._raw()
            self._raw()

    def take(self):
        v = 0
        for _ in range(self.num_bits):
            v = (v << 1) | self._bit()
        return v


def _poseidon_constants(t=3, r_f=8, r_p=60):
    """round constants, Cauchy MDS and its inverse over bn256::Fr (primitives.rs:57-84, mds.rs:5-102)"""
    g = _Grain(254, t, r_f, r_p)
    rcs = []
    for _ in range(r_f + r_p):
        row = []
        while len(row) < t:
            v = g.take()
            if v < P:
                row.append(v)
        rcs.append(row)
    while True:
        vals = [g.take() % P for _ in range(2 * t)]
        if len(set(vals)) == len(vals):
            break
    xs, ys = vals[:t], vals[t:]
    mds = [[pow(xs[i] + ys[j], -1, P) for j in range(t)] for i in range(t)]

    def lag(pts, j, x):
        acc = 1
        for m, xm in enumerate(pts):
            if m != j:
                acc = acc * (x - xm) % P * pow(pts[j] - xm, -1, P) % P
        return acc

    nys = [(-y) % P for y in ys]
    minv = [[(xs[j] - nys[i]) * lag(xs, j, nys[i]) % P * lag(nys, i, xs[j]) % P for j in range(t)] for i in range(t)]
    return rcs, mds, minv


class PoseidonCircuit(Circuit):
    """poseidon_circuit.rs (:68-123) with the Pow5 chip: WIDTH 3, RATE 2, L 2, R_F 8, R_P 60 (:19-25,129-149).
    advice state0..2 = 0..2, partial_sbox = 3; fixed rc_a = 0..2, rc_b = 3..5, selector columns 6..8."""

    name = "poseidon"
    num_advice, num_fixed, num_instance, num_selectors = 4, 9, 1, 3
    degree = 6
    permutation_columns = [("instance", 0), ("fixed", 3), ("advice", 0), ("advice", 1), ("advice", 2),
                           ("fixed", 4), ("fixed", 5)]
    advice_queries = [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1), (3, 0), (2, -1), (0, -1), (1, -1)]
    fixed_queries = [(3, 0), (4, 0), (5, 0), (0, 0), (1, 0), (2, 0), (6, 0), (7, 0), (8, 0)]
    constants = [3]
    _CONSTS = None

    def __init__(self, message=None):
        self.message = None if message is None else [m % P for m in message]
        if PoseidonCircuit._CONSTS is None:
            PoseidonCircuit._CONSTS = _poseidon_constants()
        self.rcs, self.mds, self.minv = PoseidonCircuit._CONSTS
        mds, minv = self.mds, self.minv
        s_cur = [_adv(i, i, 0) for i in range(3)]
        s_next = [_adv(3 + i, i, 1) for i in range(3)]
        ps = _adv(6, 3, 0)
        s_prev = [_adv(8, 0, -1), _adv(9, 1, -1), _adv(7, 2, -1)]
        rc_b = [_fix(i, 3 + i, 0) for i in range(3)]
        rc_a = [_fix(3 + i, i, 0) for i in range(3)]
        s_full, s_partial, s_pad = _fix(6, 6, 0), _fix(7, 7, 0), _fix(8, 8, 0)

        def pow5(v):
            v2 = _prod(v, v)
            return _prod(_prod(v2, v2), v)

        gates = []
        for nx in range(3):
            terms = [_scaled(pow5(_sum(s_cur[i], rc_a[i])), mds[nx][i]) for i in range(3)]
            gates.append(_prod(s_full, _sub(_sum(_sum(terms[0], terms[1]), terms[2]), s_next[nx])))

        def mid(i):
            acc = _scaled(ps, mds[i][0])
            for c in (1, 2):
                acc = _sum(acc, _scaled(_sum(s_cur[c], rc_a[c]), mds[i][c]))
            return acc

        def nxt(i):
            return _sum(_sum(_scaled(s_next[0], minv[i][0]), _scaled(s_next[1], minv[i][1])),
                        _scaled(s_next[2], minv[i][2]))

        partial = [_sub(pow5(_sum(s_cur[0], rc_a[0])), ps), _sub(pow5(_sum(mid(0), rc_b[0])), nxt(0))]
        partial += [_sub(_sum(mid(i), rc_b[i]), nxt(i)) for i in (1, 2)]
        gates += [_prod(s_partial, g) for g in partial]
        pad = [_sub(_sum(s_prev[i], s_cur[i]), s_next[i]) for i in (0, 1)] + [_sub(s_prev[2], s_next[2])]
        gates += [_prod(s_pad, g) for g in pad]
        self.gates = gates

    @classmethod
    def from_json(cls, s):
        return cls([int(v) for v in json.loads(s)["x"]])     # poseidon_circuit.rs:37-41, 232-246

    def public_inputs(self, s):
        out = json.loads(s).get("output")
        return [int(out, 16)] if out else [self.output()]     # wasm.rs:116 hex_to_fr(output)

    def _permutation_rows(self):
        rcs, mds = self.rcs, self.mds
        cap = (2 << 64) % P
        state = [self.message[0], self.message[1], cap]
        rows, sbox = [list(state)], {}
        mix = lambda v: [sum(mds[i][j] * v[j] for j in range(3)) % P for i in range(3)]  # noqa: E731
        for r in range(4):
            state = mix([pow((state[i] + rcs[r][i]) % P, 5, P) for i in range(3)])
            rows.append(list(state))
        for r in range(30):
            rnd = 4 + 2 * r
            r0 = [pow((state[0] + rcs[rnd][0]) % P, 5, P)] + [(state[i] + rcs[rnd][i]) % P for i in (1, 2)]
            sbox[4 + r] = r0[0]
            m = mix(r0)
            state = mix([pow((m[0] + rcs[rnd + 1][0]) % P, 5, P)] + [(m[i] + rcs[rnd + 1][i]) % P for i in (1, 2)])
            rows.append(list(state))
        for r in range(4):
            state = mix([pow((state[i] + rcs[64 + r][i]) % P, 5, P) for i in range(3)])
            rows.append(list(state))
        return rows, sbox

    def output(self):
        return self._permutation_rows()[0][-1][0]

    def synthesize_advice(self, n):
        adv = [_Col(n) for _ in range(4)]
        m0, m1 = self.message
        cap = (2 << 64) % P
        adv[0][0], adv[1][0] = m0, m1
        adv[2][1] = cap
        adv[2][2] = cap
        adv[0][3], adv[1][3] = m0, m1
        adv[0][4], adv[1][4], adv[2][4] = m0, m1, cap
        rows, sbox = self._permutation_rows()
        for off, st in enumerate(rows):
            for i in range(3):
                adv[i][5 + off] = st[i]
        for off, v in sbox.items():
            adv[3][5 + off] = v
        return adv

    def synthesize_fixed(self, n):
        f = [_Col(n) for _ in range(9)]
        rcs = self.rcs
        f[3][2] = (2 << 64) % P
        f[8][3] = 1
        for r in range(4):
            for i in range(3):
                f[i][5 + r] = rcs[r][i]
            f[6][5 + r] = 1
        for r in range(30):
            off, rnd = 4 + r, 4 + 2 * r
            for i in range(3):
                f[i][5 + off] = rcs[rnd][i]
                f[3 + i][5 + off] = rcs[rnd + 1][i]
            f[7][5 + off] = 1
        for r in range(4):
            for i in range(3):
                f[i][5 + 34 + r] = rcs[64 + r][i]
            f[6][5 + 34 + r] = 1
        return f

    def copy_constraints(self):
        a = lambda c, r: (("advice", c), r)  # noqa: E731
        out = [((("fixed", 3), i), a(i, 1)) for i in range(3)]
        out += [(a(i, 2), a(i, 1)) for i in range(3)]
        out += [(a(i, 3), a(i, 0)) for i in range(2)]
        out += [(a(i, 5), a(i, 4)) for i in range(3)]
        out.append((a(0, 43), (("instance", 0), 0)))
        return out


class CollatzCircuit(Circuit):
    """collatz.rs: advice witness, is_odd, is_one; selectors final_entry (0) / selector (1) compressed into fixed
    columns 0 / 1; four gates, degree 4; equality on `witness`, no copies, no instance column.  Region i of the
    SimpleFloorPlanner starts at row i(i+3)/2 and uses offsets i, i+1 (:119-134, :180-198); the last at 527."""

    name = "collatz"
    num_advice, num_fixed, num_instance, num_selectors = 3, 2, 0, 2
    degree = 4
    permutation_columns = [("advice", 0)]
    advice_queries = [(0, 0), (0, 1), (1, 0), (2, 0)]
    fixed_queries = [(0, 0), (1, 0)]
    instance_queries = []

    def __init__(self, seq=None):
        seq = list(seq or [])[:32]
        self.x = [v % P for v in seq] + [1] * (32 - len(seq))           # collatz.rs:256-261
        x, y, is_odd, is_one = _adv(0, 0, 0), _adv(1, 0, 1), _adv(2, 1, 0), _adv(3, 2, 0)
        fin, sel, one = _fix(0, 0, 0), _fix(1, 1, 0), _const(1)
        self.gates = [
            _prod(sel, _prod(_sub(one, is_odd), _sub(x, _prod(_const(2), y)))),
            _prod(_prod(sel, _sub(one, is_one)), _prod(is_odd, _sub(_sum(_prod(_const(3), x), one), y))),
            _prod(_prod(sel, is_one), _sum(_sub(x, y), _sub(x, one))),
            _prod(fin, _sub(one, x)),
        ]

    @classmethod
    def from_json(cls, s):
        return cls([int(v) for v in json.loads(s)["x"]])

    def public_inputs(self, s):
        return []

    def synthesize_advice(self, n):
        adv = [_Col(n) for _ in range(3)]
        for i in range(31):
            row = i * (i + 3) // 2 + i
            adv[0][row], adv[0][row + 1] = self.x[i], self.x[i + 1]
            adv[1][row] = self.x[i] & 1
            adv[2][row] = 1 if self.x[i] == 1 else 0
        adv[0][527 + 31] = self.x[31]
        return adv

    def synthesize_fixed(self, n):
        f = [_Col(n) for _ in range(2)]
        for i in range(31):
            f[1][i * (i + 3) // 2 + i] = 1
        f[0][527 + 31] = 1
        return f

    def copy_constraints(self):
        return []


# ----------------------------------------------------------------------------------- extended-domain ops ----
class _ExtOps:
    """expression evaluation on extended-coset evaluation vectors held in HBM"""

    def __init__(self, dom):
        self.dom = dom
        self.step = 1 << (dom.extended_k - dom.k)
        self.en = 1 << dom.extended_k

    def mul(self, a, b):
        return self.dom.pointwise("mul", a.clone(), b)

    def add(self, a, b):
        return self.dom.pointwise("add", a.clone(), b)

    def sub(self, a, b):
        return self.dom.pointwise("sub", a.clone(), b)

    def scale(self, a, c):
        return self.dom.scale(a.clone(), c)

    def constant(self, c):
        import torch
        row = torch.from_numpy(_limbs_of([c]).view(np.int64)).cuda()
        return row.expand(self.en, 4).contiguous()

    def rotate(self, a, rot):
        """evaluations of f(w^rot X) from those of f(X)"""
        import torch
        return torch.roll(a, shifts=-rot * self.step, dims=0).contiguous()

    def x_column(self):
        """evaluations of the polynomial X on the coset: zeta * extended_omega^i"""
        col = self.constant(self.dom.g_coset)
        w = self.dom._m["extended_omega"]
        st = self.dom._L.h2_poly_coset_device(self.dom.curve, ctypes.c_void_p(col.data_ptr()), self.en, 1,
                                              w.ctypes.data, self.dom._stream())
        _lib.check(st, "h2_poly_coset_device")
        return col

    def evaluate(self, e, cols, cache):
        """a gate expression on the extended-domain columns cols[kind][column]"""
        if e in cache:
            return cache[e]
        t = e[0]
        if t == "const":
            r = self.constant(e[1])
        elif t in ("advice", "fixed", "instance"):
            base = cols[t][e[2]]
            r = base if e[3] == 0 else self.rotate(base, e[3])
        elif t == "neg":
            r = self.scale(self.evaluate(e[1], cols, cache), P - 1)
        elif t == "sum":
            r = self.add(self.evaluate(e[1], cols, cache), self.evaluate(e[2], cols, cache))
        elif t == "prod":
            r = self.mul(self.evaluate(e[1], cols, cache), self.evaluate(e[2], cols, cache))
        elif t == "scaled":
            r = self.scale(self.evaluate(e[1], cols, cache), e[2])
        else:
            raise ValueError(t)
        cache[e] = r
        return r


# ------------------------------------------------------------------------------------------------- keygen ----
def _permutation_mapping(circuit, n):
    """Assembly::copy of halo2_proofs/src/plonk/permutation/keygen.rs"""
    cols = circuit.permutation_columns
    index = {c: i for i, c in enumerate(cols)}
    mapping, aux, sizes = {}, {}, {}
    for (lc, lr), (rc, rr) in circuit.copy_constraints():
        left, right = (index[lc], lr), (index[rc], rr)
        for cell in (left, right):
            if cell not in mapping:
                mapping[cell], aux[cell], sizes[cell] = cell, cell, 1
        if aux[left] == aux[right]:
            continue
        big, small = aux[left], aux[right]
        if sizes[big] < sizes[small]:
            big, small = small, big
        sizes[big] += sizes[small]
        i = small
        while True:
            aux[i] = big
            i = mapping[i]
            if i == small:
                break
        mapping[left], mapping[right] = mapping[right], mapping[left]
    return mapping            # cells not present map to themselves


def vk_debug_string(circuit, k, fixed_commitments, sigma_commitments):
    """format!("{:?}", vk.pinned()) of halo2_proofs @6b43b6b"""
    col = lambda kind, i: "Column { index: %d, column_type: %s }" % (i, kind.capitalize())  # noqa: E731
    pt = lambda q: "Infinity" if q is None else "(0x%064x, 0x%064x)" % q                     # noqa: E731
    ext_k = k
    while (1 << ext_k) < (1 << k) * (circuit.degree - 1):
        ext_k += 1
    omega = pow(ROOT_OF_UNITY, 1 << (TWO_ADICITY - k), P)
    s = ('PinnedVerificationKey { base_modulus: "0x%x", scalar_modulus: "0x%x", domain: PinnedEvaluationDomain '
         '{ k: %d, extended_k: %d, omega: 0x%064x }, ' % (Q, P, k, ext_k, omega))
    s += ("cs: PinnedConstraintSystem { num_fixed_columns: %d, num_advice_columns: %d, num_instance_columns: %d, "
          "num_selectors: %d, gates: [%s], " % (circuit.num_fixed, circuit.num_advice, circuit.num_instance,
                                                circuit.num_selectors, ", ".join(_expr_debug(g) for g in circuit.gates)))
    s += "advice_queries: [%s], " % ", ".join("(%s, Rotation(%d))" % (col("advice", c), r) for c, r in circuit.advice_queries)
    s += "instance_queries: [%s], " % ", ".join("(%s, Rotation(%d))" % (col("instance", c), r) for c, r in circuit.instance_queries)
    s += "fixed_queries: [%s], " % ", ".join("(%s, Rotation(%d))" % (col("fixed", c), r) for c, r in circuit.fixed_queries)
    s += "permutation: Argument { columns: [%s] }, " % ", ".join(col(kd, i) for kd, i in circuit.permutation_columns)
    s += "lookups: [], constants: [%s], minimum_degree: None }, " % ", ".join(col("fixed", c) for c in circuit.constants)
    s += "fixed_commitments: [%s], " % ", ".join(pt(q) for q in fixed_commitments)
    s += "permutation: VerifyingKey { commitments: [%s] } }" % ", ".join(pt(q) for q in sigma_commitments)
    return s


class _Dev:
    """Device-resident columns: torch int64 CUDA tensors (..., n, 4) of Montgomery limbs.  Everything numeric goes
    through the C ABI (pointwise kernels, NTT, MSM); torch only owns the memory and moves bytes."""

    R2_RAW = None
    ONE_RAW = None

    def __init__(self, dom, params):
        import torch
        self.torch, self.dom, self.params = torch, dom, params
        self.L, self.curve, self.n = dom._L, dom.curve, dom.n
        if _Dev.R2_RAW is None:
            _Dev.R2_RAW = np.frombuffer((R_P * R_P % P).to_bytes(32, "little"), dtype=np.uint64).copy()
            _Dev.ONE_RAW = np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint64).copy()

    def _p(self, t):
        return ctypes.c_void_p(t.data_ptr())

    def _scale_raw(self, t, limbs):
        _lib.check(self.L.h2_poly_scale_device(self.curve, self._p(t), t.numel() // 4, 1, limbs.ctypes.data,
                                               self.dom._stream()), "h2_poly_scale_device")
        return t

    # ---- host <-> device -------------------------------------------------------------------------------------
    def from_ints(self, vals):
        """canonical ints -> (len, 4) Montgomery column; the multiplication by R runs on the device"""
        buf = b"".join(v.to_bytes(32, "little") for v in vals)
        t = self.torch.from_numpy(np.frombuffer(buf, dtype=np.int64).reshape(-1, 4).copy()).cuda()
        return self._scale_raw(t, _Dev.R2_RAW)          # mont_mul(x, R^2) = x R

    def column(self, vals):
        """a length-n host column (list of canonical ints); sparse columns only convert their non-zero rows"""
        touched = getattr(vals, "touched", None)
        rows = sorted(i for i in touched if vals[i]) if touched is not None else [i for i, v in enumerate(vals) if v]
        if len(rows) * 8 > len(vals):
            return self.from_ints(vals)
        t = self.torch.zeros((len(vals), 4), dtype=self.torch.int64, device="cuda")
        if rows:
            t[self.torch.tensor(rows, device="cuda")] = self.from_ints([vals[i] for i in rows])
        return t

    def to_ints(self, t):
        c = self._scale_raw(t.clone().contiguous(), _Dev.ONE_RAW)   # mont_mul(x R, 1) = x
        buf = c.cpu().numpy().tobytes()
        return [int.from_bytes(buf[i:i + 32], "little") for i in range(0, len(buf), 32)]

    def const(self, c, n=None):
        return self.from_ints([c % P]).expand(n or self.n, 4).contiguous()

    def powers(self, g, n=None):
        """the column g^i"""
        t = self.const(1, n)
        gm = _limbs_of([g])
        _lib.check(self.L.h2_poly_coset_device(self.curve, self._p(t), t.shape[0], 1, gm.ctypes.data,
                                               self.dom._stream()), "h2_poly_coset_device")
        return t

    # ---- arithmetic (new tensors unless named *_) ---------------------------------------------------------------
    def _pw(self, op, a, b):
        _lib.check(self.L.h2_poly_pointwise_device(self.curve, op, self._p(a), self._p(b), a.numel() // 4,
                                                   self.dom._stream()), "h2_poly_pointwise_device")
        return a

    def add(self, a, b):
        return self._pw(0, a.clone(), b)

    def sub(self, a, b):
        return self._pw(1, a.clone(), b)

    def mul(self, a, b):
        return self._pw(2, a.clone(), b)

    def add_(self, a, b):
        return self._pw(0, a, b)

    def mul_(self, a, b):
        return self._pw(2, a, b)

    def scale(self, a, c):
        cm = _limbs_of([c])
        t = a.clone()
        _lib.check(self.L.h2_poly_scale_device(self.curve, self._p(t), t.numel() // 4, 1, cm.ctypes.data,
                                               self.dom._stream()), "h2_poly_scale_device")
        return t

    def divide_linear(self, col, z):
        """(col - col(z)) / (X - z) as a new column: kate_division on the device"""
        col = col.contiguous()
        q = self.torch.empty_like(col)
        zm = _limbs_of([z])
        _lib.check(self.L.h2_poly_divide_linear_device(self.curve, self._p(col), col.shape[0], zm.ctypes.data,
                                                       self._p(q), self.dom._stream()), "h2_poly_divide_linear_device")
        return q

    def prefix_product(self, col):
        """the column prod_{j < i} col[j] (1 in row 0)"""
        col = col.contiguous()
        out = self.torch.empty_like(col)
        _lib.check(self.L.h2_poly_prefix_product_device(self.curve, self._p(col), col.shape[0], self._p(out),
                                                        self.dom._stream()), "h2_poly_prefix_product_device")
        return out

    def random_scalars(self, seed32, count):
        """`count` draws of Fr::random(ChaCha20Rng::from_seed(seed32)) as a device column"""
        t = self.torch.empty((count, 4), dtype=self.torch.int64, device="cuda")
        _lib.check(self.L.h2_chacha20_scalars_device(self.curve, seed32, 0, count, self._p(t), self.dom._stream()),
                   "h2_chacha20_scalars_device")
        return t

    def inverse_(self, t):
        _lib.check(self.L.h2_poly_inverse_device(self.curve, self._p(t), t.numel() // 4, self.dom._stream()),
                   "h2_poly_inverse_device")
        return t

    def evals(self, cols, point):
        """f_j(point) for the m coefficient-form columns of `cols` (m, n, 4): multiply by the powers of the point
        and fold the halves together (log2 n pointwise additions over all columns at once)"""
        m, n = cols.shape[0], cols.shape[1]
        # (n, m, 4) so that halves are contiguous; always a fresh copy: the fold below works in place
        a = cols.permute(1, 0, 2).clone(memory_format=self.torch.contiguous_format)
        pw = self.powers(point, n).unsqueeze(1).expand(n, m, 4).contiguous()
        self.mul_(a, pw)
        length = n
        while length > 1:
            half = length // 2
            lo, hi = a[:half], a[half:length]
            _lib.check(self.L.h2_poly_pointwise_device(self.curve, 0, self._p(lo), self._p(hi), half * m,
                                                       self.dom._stream()), "h2_poly_pointwise_device")
            length = half
        return self.to_ints(a[0])

    def commit(self, cols, lagrange):
        """m device columns (m, n, 4) -> m affine points (canonical ints); the MSM result never leaves the device
        until it is 96 bytes per column"""
        import torch.distributed as dist
        m, n = cols.shape[0], cols.shape[1]
        bases = self.params._gl if lagrange else self.params._g
        sharded = (dist.is_available() and dist.is_initialized() and
                   (dist.get_world_size() > 1 or _sharded.FORCE_GATHER))
        cols = cols.contiguous()
        stream = self.dom._stream().value or 0
        if sharded:
            # one process per GPU: whole columns per rank when m divides evenly, else every rank takes a point range of
            # every column; ONE all-gather of 96-byte points per phase either way
            mode = None if dist.get_world_size() > 1 else "range"
            out = _sharded.msm_phase_device(bases, cols.data_ptr(), n, m, stream, mode=mode)
        else:
            out = self.torch.zeros((m, 12), dtype=self.torch.int64, device="cuda")
            bases.msm_device(cols.data_ptr(), n, m, out.data_ptr(), stream)
        self.torch.cuda.synchronize()
        raw = out.cpu().numpy().tobytes()
        pts = []
        for j in range(m):
            X, Y, Z = (int.from_bytes(raw[96 * j + 32 * i:96 * j + 32 * i + 32], "little") * R_Q_INV % Q for i in range(3))
            if Z == 0:
                pts.append(None)
                continue
            zi = pow(Z, -1, Q)
            pts.append((X * zi * zi % Q, Y * zi * zi % Q * zi % Q))
        return pts


class ProvingKey:
    """what keygen_vk + keygen_pk leave behind, resident in HBM: fixed / permutation columns in Lagrange and
    coefficient form, their commitments and the vk digest"""

    def __init__(self, params, circuit):
        import torch
        self.params, self.circuit = params, circuit
        self.k, self.n = params.k, params.n
        self.domain = EvaluationDomain(circuit.degree, params.k, "bn254")
        self.omega = self.domain.omega
        n = self.n
        dev = self.dev = _Dev(self.domain, params)
        self.fixed_values = torch.stack([dev.column(c) for c in circuit.synthesize_fixed(n)])
        # sigma_j[i] = delta^j w^i except on the cells the copy constraints permute
        mapping = _permutation_mapping(circuit, n)
        ncols = len(circuit.permutation_columns)
        self.omega_col = dev.powers(self.omega)
        sig = torch.stack([dev.scale(self.omega_col, pow(DELTA, j, P)) for j in range(ncols)])
        moved = [(cell, tgt) for cell, tgt in mapping.items() if cell != tgt]
        if moved:
            vals = dev.from_ints([pow(DELTA, tj, P) * pow(self.omega, trow, P) % P for _, (tj, trow) in moved])
            cj = torch.tensor([c[0] for c, _ in moved], device="cuda")
            rw = torch.tensor([c[1] for c, _ in moved], device="cuda")
            sig[cj, rw] = vals
        self.sigma_values = sig
        cols = torch.cat([self.fixed_values, self.sigma_values])
        commits = dev.commit(cols, lagrange=True)
        polys = self.domain.lagrange_to_coeff(cols.clone())
        nf = self.fixed_values.shape[0]
        self.fixed_commitments, self.sigma_commitments = commits[:nf], commits[nf:]
        self.fixed_polys, self.sigma_polys = polys[:nf], polys[nf:]
        s = vk_debug_string(circuit, self.k, self.fixed_commitments, self.sigma_commitments)
        h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
        h.update(len(s).to_bytes(8, "little"))
        h.update(s.encode())
        self.transcript_repr = int.from_bytes(h.digest(), "little") % P


def generate_keys(params, circuit):
    """utils.rs:63-70 generate_keys(params, circuit) -> pk (the vk's digest rides along)"""
    return ProvingKey(params, circuit)


# ------------------------------------------------------------------------------------------------- setup -----
_G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634),
           (8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _g2_scalar_mul(k, pt):
    """[k] pt on the BN254 twist y^2 = x^3 + 3/(9+u) over Fq2 = Fq[u]/(u^2+1) (affine, host big integers)"""
    mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)  # noqa: E731
    sub = lambda a, b: ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)                              # noqa: E731

    def inv(a):
        d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
        return (a[0] * d % Q, -a[1] * d % Q)

    def add(p1, p2):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        (x1, y1), (x2, y2) = p1, p2
        if x1 == x2:
            if ((y1[0] + y2[0]) % Q, (y1[1] + y2[1]) % Q) == (0, 0):
                return None
            lam = mul(mul((3, 0), mul(x1, x1)), inv(((2 * y1[0]) % Q, (2 * y1[1]) % Q)))
        else:
            lam = mul(sub(y2, y1), inv(sub(x2, x1)))
        x3 = sub(sub(mul(lam, lam), x1), x2)
        return (x3, sub(mul(lam, sub(x1, x3)), y1))

    acc = None
    while k:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def _g2_bytes(pt):
    (x0, x1), (y0, y1) = pt
    return b"".join((v * R_Q % Q).to_bytes(32, "little") for v in (x0, x1, y0, y1))


def generate_params(k, rng=None):
    """utils.rs:59-61 generate_params(k) = ParamsKZG::<Bn256>::new(k): one Fr::random s, g[i] = [s^i]G,
    g_lagrange[i] = [L_i(s)]G, g2, [s]g2 -- the 2 * 2^k fixed-base multiplications run on the GPU."""
    import torch
    from .api import _ensure_init
    _ensure_init()
    rng = rng or OsRng()
    s = rng.fr_random()
    n = 1 << k
    L = _lib.load()
    # every library call below goes to torch's CURRENT stream: the inputs are produced by torch kernels on that stream
    # (the library's own blocking stream would only be ordered against the null stream)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g_dev = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    s_m = _limbs_of([s])
    _lib.check(L.h2_srs_generate(0, s_m.ctypes.data, n, ctypes.c_void_p(g_dev.data_ptr()), st), "h2_srs_generate")
    # L_i(s) = w^i (s^n - 1) / (n (s - w^i)), all 2^k of them on the device: the column w^i (coset kernel on a column
    # of ones), s - w^i, a per-element inversion, two pointwise products
    omega = pow(ROOT_OF_UNITY, 1 << (TWO_ADICITY - k), P)
    if pow(s, n, P) == 1:                                    # s on the domain: L_i(s) is an indicator (host, never hit)
        lag = [1 if pow(omega, i, P) == s else 0 for i in range(n)]
        sc_dev = torch.from_numpy(_limbs_of(lag).view(np.int64)).cuda()
    else:
        def col_of(c):
            return torch.from_numpy(_limbs_of([c]).view(np.int64)).cuda().expand(n, 4).contiguous()

        def ptr(t):
            return ctypes.c_void_p(t.data_ptr())

        ws = col_of(1)
        _lib.check(L.h2_poly_coset_device(0, ptr(ws), n, 1, _limbs_of([omega]).ctypes.data, st), "h2_poly_coset_device")
        den = col_of(s)
        _lib.check(L.h2_poly_pointwise_device(0, 1, ptr(den), ptr(ws), n, st), "h2_poly_pointwise_device")   # s - w^i
        _lib.check(L.h2_poly_inverse_device(0, ptr(den), n, st), "h2_poly_inverse_device")
        _lib.check(L.h2_poly_pointwise_device(0, 2, ptr(den), ptr(ws), n, st), "h2_poly_pointwise_device")   # w^i / (s - w^i)
        t = (pow(s, n, P) - 1) * pow(n, -1, P) % P
        _lib.check(L.h2_poly_scale_device(0, ptr(den), n, 1, _limbs_of([t]).ctypes.data, st), "h2_poly_scale_device")
        sc_dev = den
    gl_dev = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    _lib.check(L.h2_fixed_base_mul(0, ctypes.c_void_p(sc_dev.data_ptr()), n, ctypes.c_void_p(gl_dev.data_ptr()), st),
               "h2_fixed_base_mul")
    torch.cuda.synchronize()
    g = g_dev.cpu().numpy().view(np.uint64)
    gl = gl_dev.cpu().numpy().view(np.uint64)
    tail = _g2_bytes(_G2_GEN) + _g2_bytes(_g2_scalar_mul(s, _G2_GEN))
    return ParamsKZG(k, g, gl, tail)


# ------------------------------------------------------------------------------------------- create_proof ----
def generate_proof_with_instance(params, pk, circuit, public_input, rng=None, trace=None):
    """utils.rs:95-123: create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC, Challenge255, _, Blake2bWrite, _>"""
    return _create_proof(params, pk, circuit, public_input, rng, trace, "gwc")


def generate_proof(params, pk, circuit, rng=None, trace=None):
    """utils.rs:72-93: create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK, ...> with instances &[&[]]"""
    return _create_proof(params, pk, circuit, [], rng, trace, "shplonk")


def _interpolate(points, values):
    """coefficients of the polynomial of degree < len(points) through (points[i], values[i])"""
    out = [0] * len(points)
    for i, (xi, yi) in enumerate(zip(points, values)):
        term, den = [1], 1
        for j, xj in enumerate(points):
            if j != i:
                nt = [0] * (len(term) + 1)
                for d, c in enumerate(term):
                    nt[d] = (nt[d] - c * xj) % P
                    nt[d + 1] = (nt[d + 1] + c) % P
                term = nt
                den = den * (xi - xj) % P
        scale = yi * pow(den, -1, P) % P
        for d, c in enumerate(term):
            out[d] = (out[d] + c * scale) % P
    return out


def _divide_linear_device(dev, col, points):
    """col / prod (X - p) for a device column known to vanish at the points: one kate_division per point, on the
    device (the quotient keeps the column's length, its top coefficients are zero)"""
    q = col
    for pt in points:
        q = dev.divide_linear(q, pt)
    return q


def _shplonk_open(tr, dev, n, queries, evals, trace):
    """ProverSHPLONK::create_proof.  queries: (point, device column, key); evals[(key, point)]
    are the already computed evaluations.  Linear combinations run on the device, the two quotient MSMs too."""
    y = tr.squeeze_challenge()
    v = tr.squeeze_challenge()
    trace.update(shplonk_y=y, v=v)
    polys = []
    for pt, col, key in queries:
        for entry in polys:
            if entry[0] == key:
                if pt not in entry[2]:
                    entry[2].append(pt)
                break
        else:
            polys.append((key, col, [pt]))
    groups = []
    for key, col, pts in polys:
        pset = sorted(pts)
        for g in groups:
            if g[0] == pset:
                g[1].append((key, col))
                break
        else:
            groups.append((pset, [(key, col)]))
    T = sorted({pt for pset, _ in groups for pt in pset})
    h, vp, per_set = None, 1, []
    for pset, members in groups:
        acc, yp, rems = None, 1, []
        for key, col in members:
            r = _interpolate(pset, [evals[(key, pt)] for pt in pset])
            rems.append(r)
            term = dev.scale(col, yp)
            acc = term if acc is None else dev.add_(acc, term)
            yp = yp * y % P
        # subtract sum_j y^j R_ij (a polynomial of degree < |S_i|)
        rsum = [0] * len(pset)
        yp = 1
        for r in rems:
            rsum = [(a + yp * b) % P for a, b in zip(rsum, r)]
            yp = yp * y % P
        acc[:len(pset)] = dev.sub(acc[:len(pset)].contiguous(), dev.from_ints(rsum))
        q = _divide_linear_device(dev, acc, pset)
        term = dev.scale(q, vp)
        h = term if h is None else dev.add_(h, term)
        vp = vp * v % P
        per_set.append((pset, members, rems))
    tr.write_point(dev.commit(h.unsqueeze(0), lagrange=False)[0])
    u = tr.squeeze_challenge()
    trace.update(u=u)
    zt = 1
    for pt in T:
        zt = zt * (u - pt) % P
    L, vp, z0 = None, 1, None
    for pset, members, rems in per_set:
        z_i = 1
        for pt in T:
            if pt not in pset:
                z_i = z_i * (u - pt) % P
        if z0 is None:
            z0 = z_i
        inner, yp, const = None, 1, 0
        for (key, col), r in zip(members, rems):
            term = dev.scale(col, yp)
            inner = term if inner is None else dev.add_(inner, term)
            const = (const + yp * _horner(r, u)) % P
            yp = yp * y % P
        inner[:1] = dev.sub(inner[:1].contiguous(), dev.from_ints([const]))
        term = dev.scale(inner, vp * z_i % P)
        L = term if L is None else dev.add_(L, term)
        vp = vp * v % P
    L = dev.sub(L, dev.scale(h, zt))
    w = dev.scale(_divide_linear_device(dev, L, [u]), pow(z0, -1, P))
    tr.write_point(dev.commit(w.unsqueeze(0), lagrange=False)[0])


def _create_proof(params, pk, circuit, public_input, rng, trace, opening):
    import torch
    rng = rng or OsRng()
    trace = trace if trace is not None else {}
    n, omega, dom, dev = pk.n, pk.omega, pk.domain, pk.dev
    bf, d = circuit.blinding_factors(), circuit.degree
    tr = _Transcript()
    tr.common_scalar(pk.transcript_repr)
    instance_cols = []
    if circuit.num_instance:
        inst = _Col(n)
        for i, v in enumerate(public_input):
            inst[i] = v % P
        instance_cols = [dev.column(inst)]
        for v in public_input:
            tr.common_scalar(v)
    instance_values = torch.stack(instance_cols) if instance_cols else torch.zeros((0, n, 4), dtype=torch.int64, device="cuda")

    # advice: synthesize, blind the last bf + 1 rows, commit
    advice_host = circuit.synthesize_advice(n)
    for col in advice_host:
        for row in range(n - (bf + 1), n):
            col[row] = rng.fr_random()
    for _ in advice_host:
        rng.fr_random()
    advice_values = torch.stack([dev.column(c) for c in advice_host])
    for pt in dev.commit(advice_values, lagrange=True):
        tr.write_point(pt)
    theta, beta, gamma = tr.squeeze_challenge(), tr.squeeze_challenge(), tr.squeeze_challenge()
    trace.update(theta=theta, beta=beta, gamma=gamma)

    # permutation grand products, d - 2 columns per set: per-row ratios and their running product on the device
    values_of = {"advice": advice_values, "fixed": pk.fixed_values, "instance": instance_values}
    pcols = circuit.permutation_columns
    sets = [list(range(s, min(s + d - 2, len(pcols)))) for s in range(0, len(pcols), d - 2)]
    gamma_col = dev.const(gamma)
    z_cols, last_z = [], 1
    for cols in sets:
        num = den = None
        for j in cols:
            v = values_of[pcols[j][0]][pcols[j][1]]
            vg = dev.add(v, gamma_col)
            tn = dev.add_(dev.scale(pk.omega_col, pow(DELTA, j, P) * beta % P), vg)
            td = dev.add_(dev.scale(pk.sigma_values[j], beta), vg)
            num = tn if num is None else dev.mul_(num, tn)
            den = td if den is None else dev.mul_(den, td)
        # z[i] = last_z * prod_{j < i} ratio[j] on the usable rows: a prefix product on the device; only the 32 bytes of
        # its last usable row come back (the next set starts from there)
        pref = dev.prefix_product(dev.mul_(num, dev.inverse_(den)))
        z = dev.scale(pref, last_z) if last_z != 1 else pref
        z[n - bf:] = dev.from_ints([rng.fr_random() for _ in range(bf)])
        last_z = last_z * dev.to_ints(pref[n - bf - 1:n - bf])[0] % P
        rng.fr_random()
        z_cols.append(z)
    z_values = torch.stack(z_cols)
    for pt in dev.commit(z_values, lagrange=True):
        tr.write_point(pt)

    # random polynomial of the vanishing argument (one thread chunk: one seed, n sequential draws)
    random_poly = dev.random_scalars(rng.fill(32), n)
    rng.fr_random()
    tr.write_point(dev.commit(random_poly.unsqueeze(0), lagrange=False)[0])

    # coefficient forms (one batched inverse NTT on the GPU)
    basis = []
    for rows in ([0], [n - bf - 1], list(range(n - bf, n))):
        v = _Col(n)
        for r in rows:
            v[r] = 1
        basis.append(dev.column(v))
    na, ni, nz = advice_values.shape[0], instance_values.shape[0], z_values.shape[0]
    coeffs = dom.lagrange_to_coeff(torch.cat([advice_values, instance_values, z_values, torch.stack(basis)]))
    advice_polys, instance_polys = coeffs[:na], coeffs[na:na + ni]
    z_polys, basis_polys = coeffs[na + ni:na + ni + nz], coeffs[na + ni + nz:]

    # quotient on the extended coset
    y = tr.squeeze_challenge()
    trace.update(y=y)
    ops = _ExtOps(dom)
    adv_e, fix_e = dom.coeff_to_extended(advice_polys), dom.coeff_to_extended(pk.fixed_polys)
    inst_e = dom.coeff_to_extended(instance_polys) if ni else []
    sig_e, z_e = dom.coeff_to_extended(pk.sigma_polys), dom.coeff_to_extended(z_polys)
    l0_e, l_last_e, l_blind_e = dom.coeff_to_extended(basis_polys)
    one = ops.constant(1)
    l_active_e = ops.sub(ops.sub(one, l_last_e), l_blind_e)
    ext_of = {"advice": adv_e, "fixed": fix_e, "instance": inst_e}
    cache = {}
    terms = [ops.evaluate(g, ext_of, cache) for g in circuit.gates]
    cache.clear()
    terms.append(ops.mul(l0_e, ops.sub(one, z_e[0])))
    terms.append(ops.mul(l_last_e, ops.sub(ops.mul(z_e[-1], z_e[-1]), z_e[-1])))
    for i in range(1, len(sets)):
        terms.append(ops.mul(l0_e, ops.sub(z_e[i], ops.rotate(z_e[i - 1], -(bf + 1)))))
    x_col = ops.x_column()
    gamma_ext = ops.constant(gamma)
    for i, cols in enumerate(sets):
        left, right = ops.rotate(z_e[i], 1), z_e[i]
        for j in cols:
            v = ext_of[pcols[j][0]][pcols[j][1]]
            left = ops.mul(left, ops.add(ops.add(v, ops.scale(sig_e[j], beta)), gamma_ext))
            right = ops.mul(right, ops.add(ops.add(v, ops.scale(x_col, pow(DELTA, j, P) * beta % P)), gamma_ext))
        terms.append(ops.mul(l_active_e, ops.sub(left, right)))
    numer = terms[0]
    for t in terms[1:]:
        numer = ops.add(ops.scale(numer, y), t)
    h_pieces = dom.extended_to_coeff(dom.divide_by_vanishing_poly(numer)).view(d - 1, n, 4)
    del terms, numer, adv_e, fix_e, inst_e, sig_e, z_e, ext_of
    for pt in dev.commit(h_pieces, lagrange=False):
        tr.write_point(pt)
    for _ in range(d - 1):
        rng.fr_random()

    # evaluations at x: every (polynomial, point) pair once, batched per point on the device
    x = tr.squeeze_challenge()
    trace.update(x=x)
    w_back = pow(omega, -(bf + 1), P)
    rot_point = lambda rot: x * pow(omega, rot, P) % P  # noqa: E731
    wanted = []                                            # (key, device column, point) in transcript order
    for col, rot in circuit.advice_queries:
        wanted.append((("advice", col), advice_polys[col], rot_point(rot)))
    for col, rot in circuit.fixed_queries:
        wanted.append((("fixed", col), pk.fixed_polys[col], rot_point(rot)))
    wanted.append((("random", 0), random_poly, x))
    for j in range(pk.sigma_polys.shape[0]):
        wanted.append((("sigma", j), pk.sigma_polys[j], x))
    for i in range(nz):
        wanted.append((("z", i), z_polys[i], x))
        wanted.append((("z", i), z_polys[i], x * omega % P))
        if i + 1 < nz:
            wanted.append((("z", i), z_polys[i], x * w_back % P))
    evals = {}
    by_point = {}
    for key, col, pt in wanted:
        by_point.setdefault(pt, []).append((key, col))
    for pt, items in by_point.items():
        vals = dev.evals(torch.stack([c for _, c in items]), pt)
        for (key, _), val in zip(items, vals):
            evals[(key, pt)] = val
    for key, _, pt in wanted:
        tr.write_scalar(evals[(key, pt)])

    # multiopen
    xn = pow(x, n, P)
    h_poly = h_pieces[d - 2].clone()
    for i in range(d - 3, -1, -1):
        h_poly = dev.add_(dev.scale(h_poly, xn), h_pieces[i])
    queries = [(rot_point(rot), advice_polys[col], ("advice", col)) for col, rot in circuit.advice_queries]
    for i in range(nz):
        queries += [(x, z_polys[i], ("z", i)), (x * omega % P, z_polys[i], ("z", i))]
    for i in range(nz - 2, -1, -1):
        queries.append((x * w_back % P, z_polys[i], ("z", i)))
    queries += [(rot_point(rot), pk.fixed_polys[col], ("fixed", col)) for col, rot in circuit.fixed_queries]
    queries += [(x, pk.sigma_polys[j], ("sigma", j)) for j in range(pk.sigma_polys.shape[0])]
    queries += [(x, h_poly, ("h", 0)), (x, random_poly, ("random", 0))]
    if opening == "shplonk":
        evals[(("h", 0), x)] = dev.evals(h_poly.unsqueeze(0), x)[0]
        _shplonk_open(tr, dev, n, queries, evals, trace)
        return bytes(tr.bytes)
    # GWC: one witness polynomial per distinct point; the v-power combinations on the device
    v = tr.squeeze_challenge()
    trace.update(v=v)
    points = []
    for pt, _, _ in queries:
        if pt not in points:
            points.append(pt)
    witnesses = []
    for pt in points:
        acc, vp = None, 1
        for qpt, col, _ in queries:
            if qpt == pt:
                term = dev.scale(col, vp)
                acc = term if acc is None else dev.add_(acc, term)
                vp = vp * v % P
        witnesses.append(_divide_linear_device(dev, acc, [pt]))
    for pt in dev.commit(torch.stack(witnesses), lagrange=False):
        tr.write_point(pt)
    return bytes(tr.bytes)


def wasm_generate_proof(params_bytes, s, circuit_index, rng=None):
    """wasm.rs:77-122: read params, keygen on the empty circuit, prove
    (circuit 0 = Collatz / SHPLONK, 1 = arithmetic / GWC, anything else = Poseidon / GWC, as the reference's match)"""
    params = ParamsKZG.read(params_bytes)
    if circuit_index == 0:
        circuit = CollatzCircuit.from_json(s)
        return generate_proof(params, generate_keys(params, circuit), circuit, rng)
    circuit = ArithmeticCircuit.from_json(s) if circuit_index == 1 else PoseidonCircuit.from_json(s)
    pk = generate_keys(params, circuit)
    return generate_proof_with_instance(params, pk, circuit, circuit.public_inputs(s), rng)
