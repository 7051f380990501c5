"""The reference crate's verify() surface.

Mirrors /root/reference/circuits/src/utils.rs and wasm.rs:
    verify(params, pk, proof)                                 utils.rs:125-140  verify_proof, KZG + SHPLONK
    verify_with_instance(params, pk, proof, public_input)     utils.rs:141-158  verify_proof, KZG + GWC
    wasm_verify_proof(params_bytes, proof, json, circuit)     wasm.rs:125-179   re-keygen, recompute the public input
    wasm_simulate_circuit(json, circuit) / get_circuit_count  wasm.rs:68-74,182
halo2_proofs::plonk::verify_proof itself is in the un-vendored dependency (halo2_proofs @6b43b6b); what it does is
fixed by the proof stream the prover writes: replay the Blake2b transcript, recompute the
gate and permutation expressions from the evaluations, check h(x) (x^n - 1) against them, and check the KZG openings
with one pairing equation  e(left, [s]G2) = e(right, G2).

What runs where: the transcript and the few hundred scalar operations on the host; the two small MSMs that build
`left` and `right` (about 35 points: the proof's commitments and the verifying key's) on the GPU through the C ABI
(h2_msm, the same kernels as the prover); the pairing on the host (pairing.py).  The reference's
verify_with_instance unwraps the result (utils.rs:150-157: an invalid proof is a trap there); here every failure is
`False`.
"""
import json

import numpy as np

from . import pairing as _pairing
from .api import ParamsKZG, best_multiexp
from .lib import H2Error
from .prover import (DELTA, P, Q, R_P, R_Q, R_Q_INV, ArithmeticCircuit, CollatzCircuit, PoseidonCircuit,
                     _horner, _interpolate, generate_keys)


class VerifyError(Exception):
    pass


class _ReadTranscript:
    """Blake2bRead<&[u8], G1Affine, Challenge255<_>> (utils.rs:132,147)"""

    def __init__(self, proof):
        import hashlib
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.buf, self.pos = bytes(proof), 0

    def common_scalar(self, s):
        self.state.update(b"\x02" + int(s % P).to_bytes(32, "little"))

    def _take(self):
        if self.pos + 32 > len(self.buf):
            raise VerifyError("proof too short")
        chunk = self.buf[self.pos:self.pos + 32]
        self.pos += 32
        return chunk

    def read_scalar(self):
        v = int.from_bytes(self._take(), "little")
        if v >= P:
            raise VerifyError("scalar not canonical")
        self.common_scalar(v)
        return v

    def read_point(self):
        """32-byte compressed G1 point: x little-endian, bit 6 of the last byte = parity of y, bit 7 = identity"""
        b = bytearray(self._take())
        sign, inf = (b[31] >> 6) & 1, (b[31] >> 7) & 1
        b[31] &= 0x3F
        x = int.from_bytes(b, "little")
        if x >= Q:
            raise VerifyError("point x not canonical")
        if inf or (x == 0 and sign == 0):
            # Blake2bRead::read_point -> common_point refuses the point at infinity: the reference rejects such a proof
            raise VerifyError("point at infinity in the proof")
        else:
            y2 = (x * x * x + 3) % Q
            y = pow(y2, (Q + 1) // 4, Q)
            if y * y % Q != y2:
                raise VerifyError("point not on the curve")
            if (y & 1) != sign:
                y = Q - y
            pt = (x, y)
        px, py = pt if pt is not None else (0, 0)
        self.state.update(b"\x01" + px.to_bytes(32, "little") + py.to_bytes(32, "little"))
        return pt

    def squeeze_challenge(self):
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % P


class _Msm:
    """MSMKZG: a list of (scalar, point) terms; points are affine int pairs or None"""

    def __init__(self):
        self.terms = []

    def append(self, s, pt):
        self.terms.append((s % P, pt))

    def scale(self, f):
        self.terms = [(s * f % P, pt) for s, pt in self.terms]

    def add_msm(self, other):
        self.terms += other.terms

    def copy(self):
        m = _Msm()
        m.terms = list(self.terms)
        return m

    def eval(self):
        """the group element, on the GPU (h2_bases_register + h2_msm through best_multiexp)"""
        acc = {}
        for s, pt in self.terms:
            if pt is not None and s:
                acc[pt] = (acc.get(pt, 0) + s) % P
        if not acc:
            return None
        pts = list(acc)
        scalars = np.frombuffer(b"".join((acc[p] * R_P % P).to_bytes(32, "little") for p in pts), dtype=np.uint64).reshape(-1, 4)
        bases = np.frombuffer(b"".join((x * R_Q % Q).to_bytes(32, "little") + (y * R_Q % Q).to_bytes(32, "little")
                                       for x, y in pts), dtype=np.uint64).reshape(-1, 8)
        jac = best_multiexp(scalars.copy(), bases.copy(), "bn254")
        raw = jac.tobytes()
        X, Y, Z = (int.from_bytes(raw[32 * i:32 * i + 32], "little") * R_Q_INV % Q for i in range(3))
        if Z == 0:
            return None
        zi = pow(Z, -1, Q)
        return (X * zi * zi % Q, Y * zi * zi % Q * zi % Q)


def _eval_expr(e, adv, fix, inst):
    t = e[0]
    if t == "const":
        return e[1]
    if t == "advice":
        return adv[e[1]]
    if t == "fixed":
        return fix[e[1]]
    if t == "instance":
        return inst[e[1]]
    if t == "neg":
        return -_eval_expr(e[1], adv, fix, inst) % P
    if t == "sum":
        return (_eval_expr(e[1], adv, fix, inst) + _eval_expr(e[2], adv, fix, inst)) % P
    if t == "prod":
        return _eval_expr(e[1], adv, fix, inst) * _eval_expr(e[2], adv, fix, inst) % P
    if t == "scaled":
        return _eval_expr(e[1], adv, fix, inst) * e[2] % P
    raise ValueError(t)


def _g2_points(params):
    """g2 and [s]g2 from the params tail (4 x 32-byte Montgomery limbs each: x.c0, x.c1, y.c0, y.c1)"""
    tail = params.g2_tail
    if len(tail) != 256:
        raise VerifyError("params carry no G2 points")
    vals = [int.from_bytes(tail[32 * i:32 * i + 32], "little") * R_Q_INV % Q for i in range(8)]
    g2 = ((vals[0], vals[1]), (vals[2], vals[3]))
    s_g2 = ((vals[4], vals[5]), (vals[6], vals[7]))
    if not (_pairing.g2_is_on_curve(g2) and _pairing.g2_is_on_curve(s_g2)):
        raise VerifyError("params: G2 point not on the curve")
    return g2, s_g2


def _affine_of_limbs(row):
    raw = row.tobytes()
    x = int.from_bytes(raw[:32], "little") * R_Q_INV % Q
    y = int.from_bytes(raw[32:], "little") * R_Q_INV % Q
    return None if (x == 0 and y == 0) else (x, y)


def _verify_proof(params, pk, proof, public_input, opening):
    circuit = pk.circuit
    n, k, omega = pk.n, pk.k, pk.omega
    bf, d = circuit.blinding_factors(), circuit.degree
    if len(public_input) > n - (bf + 1):
        raise VerifyError("instance too long")
    tr = _ReadTranscript(proof)
    tr.common_scalar(pk.transcript_repr)
    instance = [v % P for v in public_input]
    if circuit.num_instance:
        for v in instance:
            tr.common_scalar(v)
    elif instance:
        raise VerifyError("circuit has no instance column")
    advice_c = [tr.read_point() for _ in range(circuit.num_advice)]
    theta, beta, gamma = tr.squeeze_challenge(), tr.squeeze_challenge(), tr.squeeze_challenge()
    pcols = circuit.permutation_columns
    chunk = d - 2
    sets = [list(range(s, min(s + chunk, len(pcols)))) for s in range(0, len(pcols), chunk)]
    z_c = [tr.read_point() for _ in sets]
    random_c = tr.read_point()
    y = tr.squeeze_challenge()
    h_c = [tr.read_point() for _ in range(d - 1)]
    x = tr.squeeze_challenge()
    xn = pow(x, n, P)
    if xn == 1:
        raise VerifyError("challenge on the domain")

    # instance evaluations are recomputed from the values (KZG: instance columns are not committed)
    def lagrange_at(rows):
        """[L_row(x) for row in rows], L_i(x) = w^i (x^n - 1) / (n (x - w^i))"""
        common = (xn - 1) * pow(n, -1, P) % P
        out = []
        for r in rows:
            wi = pow(omega, r % n, P)
            out.append(wi * common % P * pow((x - wi) % P, -1, P) % P)
        return out

    inst_evals = []
    for col, rot in circuit.instance_queries:
        ls = lagrange_at([i - rot for i in range(len(instance))])
        inst_evals.append(sum(v * l for v, l in zip(instance, ls)) % P)
    adv_evals = [tr.read_scalar() for _ in circuit.advice_queries]
    fix_evals = [tr.read_scalar() for _ in circuit.fixed_queries]
    random_eval = tr.read_scalar()
    sigma_evals = [tr.read_scalar() for _ in pcols]
    z_evals = []
    for i in range(len(sets)):
        ev, nxt = tr.read_scalar(), tr.read_scalar()
        last = tr.read_scalar() if i + 1 < len(sets) else None
        z_evals.append((ev, nxt, last))

    # the vanishing argument: every gate and permutation expression at x, folded with y
    l_last, *l_blind, l_0 = lagrange_at(range(-(bf + 1), 1))
    l_blind = sum(l_blind) % P
    exprs = [_eval_expr(g, adv_evals, fix_evals, inst_evals) for g in circuit.gates]
    if sets:
        exprs.append(l_0 * (1 - z_evals[0][0]) % P)
        exprs.append(l_last * (z_evals[-1][0] * z_evals[-1][0] - z_evals[-1][0]) % P)
        for i in range(1, len(sets)):
            exprs.append(l_0 * (z_evals[i][0] - z_evals[i - 1][2]) % P)

        def column_eval(kind, idx):
            qs = {"advice": circuit.advice_queries, "fixed": circuit.fixed_queries, "instance": circuit.instance_queries}[kind]
            qi = qs.index((idx, 0))
            return {"advice": adv_evals, "fixed": fix_evals, "instance": inst_evals}[kind][qi]

        for i, cols in enumerate(sets):
            left, right = z_evals[i][1], z_evals[i][0]
            for j in cols:
                v = column_eval(*pcols[j])
                left = left * ((v + beta * sigma_evals[j] + gamma) % P) % P
                right = right * ((v + pow(DELTA, j, P) * beta % P * x + gamma) % P) % P
            exprs.append((left - right) * (1 - (l_last + l_blind)) % P)
    folded = 0
    for e in exprs:
        folded = (folded * y + e) % P
    expected_h = folded * pow(xn - 1, -1, P) % P
    h_msm = _Msm()
    for c in reversed(h_c):
        h_msm.scale(xn)
        h_msm.append(1, c)

    # the opening queries, in the order the prover batches them
    w_back = pow(omega, -(bf + 1), P)
    rot_point = lambda rot: x * pow(omega, rot, P) % P  # noqa: E731
    queries = []                                            # (point, key, commitment | _Msm, eval)
    for qi, (col, rot) in enumerate(circuit.advice_queries):
        queries.append((rot_point(rot), ("advice", col), advice_c[col], adv_evals[qi]))
    for i in range(len(sets)):
        queries.append((x, ("z", i), z_c[i], z_evals[i][0]))
        queries.append((x * omega % P, ("z", i), z_c[i], z_evals[i][1]))
    for i in range(len(sets) - 2, -1, -1):
        queries.append((x * w_back % P, ("z", i), z_c[i], z_evals[i][2]))
    for qi, (col, rot) in enumerate(circuit.fixed_queries):
        queries.append((rot_point(rot), ("fixed", col), pk.fixed_commitments[col], fix_evals[qi]))
    for j in range(len(pcols)):
        queries.append((x, ("sigma", j), pk.sigma_commitments[j], sigma_evals[j]))
    queries.append((x, ("h", 0), h_msm, expected_h))
    queries.append((x, ("random", 0), random_c, random_eval))

    def as_msm(c, factor):
        if isinstance(c, _Msm):
            m = c.copy()
            m.scale(factor)
            return m
        m = _Msm()
        m.append(factor, c)
        return m

    g0 = _affine_of_limbs(params.g[0])
    left, right = _Msm(), _Msm()
    if opening == "gwc":
        # VerifierGWC (halo2_proofs src/poly/kzg/multiopen/gwc/verifier.rs)
        v = tr.squeeze_challenge()
        points = []
        for pt, *_ in queries:
            if pt not in points:
                points.append(pt)
        ws = [tr.read_point() for _ in points]
        u = tr.squeeze_challenge()
        eval_multi, up = 0, 1
        for pt, wi in zip(points, ws):
            vp, batch, ev = 1, _Msm(), 0
            for qpt, _, c, e in queries:
                if qpt == pt:
                    batch.add_msm(as_msm(c, vp))
                    ev = (ev + vp * e) % P
                    vp = vp * v % P
            batch.scale(up)
            right.add_msm(batch)
            eval_multi = (eval_multi + up * ev) % P
            right.append(up * pt, wi)
            left.append(up, wi)
            up = up * u % P
        right.append(-eval_multi, g0)
    else:
        # VerifierSHPLONK (src/poly/kzg/multiopen/shplonk/verifier.rs)
        polys = []
        for pt, key, c, e in queries:
            for entry in polys:
                if entry[0] == key:
                    if pt not in entry[2]:
                        entry[2].append(pt)
                        entry[3][pt] = e
                    break
            else:
                polys.append((key, c, [pt], {pt: e}))
        groups = []
        for key, c, pts, evs in polys:
            pset = sorted(pts)
            for g in groups:
                if g[0] == pset:
                    g[1].append((c, evs))
                    break
            else:
                groups.append((pset, [(c, evs)]))
        T = sorted({pt for pset, _ in groups for pt in pset})
        y_ch = tr.squeeze_challenge()
        v = tr.squeeze_challenge()
        h1 = tr.read_point()
        u = tr.squeeze_challenge()
        h2 = tr.read_point()
        outer, r_outer, vp = _Msm(), 0, 1
        z_0 = z_0_diff_inv = None
        for i, (pset, members) in enumerate(groups):
            z_diff = 1
            for pt in T:
                if pt not in pset:
                    z_diff = z_diff * (u - pt) % P
            if i == 0:
                z_0 = 1
                for pt in pset:
                    z_0 = z_0 * (u - pt) % P
                if z_diff == 0:
                    raise VerifyError("challenge hits an opening point")
                z_0_diff_inv = pow(z_diff, -1, P)
                z_diff = 1
            else:
                z_diff = z_diff * z_0_diff_inv % P
            inner, r_inner, yp = _Msm(), 0, 1
            for c, evs in members:
                r_x = _interpolate(pset, [evs[pt] for pt in pset])
                r_inner = (r_inner + yp * _horner(r_x, u)) % P
                inner.add_msm(as_msm(c, yp))
                yp = yp * y_ch % P
            inner.scale(vp * z_diff % P)
            outer.add_msm(inner)
            r_outer = (r_outer + vp * r_inner % P * z_diff) % P
            vp = vp * v % P
        outer.append(-r_outer, g0)
        outer.append(-z_0, h1)
        outer.append(u, h2)
        left.append(1, h2)
        right.add_msm(outer)
    g2, s_g2 = _g2_points(params)
    return _pairing.pairing_check([(left.eval(), s_g2), (right.eval(), _pairing.g2_neg(g2))])


def verify(params, pk, proof):
    """utils.rs:125-140 verify(params, pk, proof): SHPLONK, instances &[&[]]; False where the reference returns Err"""
    try:
        return _verify_proof(params, pk, proof, [], "shplonk")
    except VerifyError:
        return False


def verify_with_instance(params, pk, proof, public_input):
    """utils.rs:141-158 verify_with_instance(params, pk, proof, public_input): GWC; the reference unwraps the
    result (an invalid proof traps, :150-157) -- here it is False"""
    try:
        return _verify_proof(params, pk, proof, list(public_input), "gwc")
    except VerifyError:
        return False


def wasm_verify_proof(params_bytes, proof, s, circuit_index):
    """wasm.rs:125-179: read params, keygen on the empty circuit, recompute the public input from the JSON
    (arithmetic: [constant, z]; Poseidon: the hash of x, not the claimed `output`), verify"""
    try:
        params = ParamsKZG.read(params_bytes)
        if circuit_index == 0:
            circuit = CollatzCircuit()
            return verify(params, generate_keys(params, circuit), proof)
        if circuit_index == 1:
            v = json.loads(s)
            circuit = ArithmeticCircuit(None, None, int(v["constant"]))
            return verify_with_instance(params, generate_keys(params, circuit), proof, [int(v["constant"]), int(v["z"])])
        circuit = PoseidonCircuit([int(t) for t in json.loads(s)["x"]])
        return verify_with_instance(params, generate_keys(params, circuit), proof, [circuit.output()])
    except (ValueError, KeyError, TypeError):
        return False
    except H2Error as e:
        if e.status == -1:          # params hold a point that is not on the curve (ParamsKZG::read would fail)
            return False
        raise


def wasm_simulate_circuit(s, circuit_index):
    """wasm.rs:68-74: Collatz "N/A" (collatz.rs:248-250); arithmetic x^2 y^2 + constant in u64 arithmetic, decimal
    (arithmetic_circuit.rs:298-301); Poseidon the native hash as Fr's Debug form (poseidon_circuit.rs:269-299)"""
    if circuit_index == 0:
        return "N/A"
    v = json.loads(s)
    if circuit_index == 1:
        x, y, c = int(v["x"]), int(v["y"]), int(v["constant"])
        r = x * x * y * y + c
        if r >= 1 << 64:
            raise OverflowError("u64 overflow (the reference panics)")
        return str(r)
    return "0x%064x" % PoseidonCircuit([int(t) for t in v["x"]]).output()


def get_circuit_count():
    """wasm.rs:182"""
    return 3
