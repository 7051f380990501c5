"""Big-integer restatement of halo2's keygen + create_proof (KZG / GWC) around the MSM/NTT hot path.

TEST INFRASTRUCTURE ONLY (never imported by halo2_prover_amd/).  It restates, from SURVEY.md Appendix A.4-A.7,
what `halo2_proofs::plonk::{keygen_vk, keygen_pk, create_proof}` @6b43b6b do when the reference calls them at
/root/reference/circuits/src/utils.rs:63-70 (keygen) and :95-123 (generate_proof_with_instance, GWC), for the
arithmetic circuit of /root/reference/circuits/src/arithmetic_circuit.rs (configure :187-230, synthesize
:232-267).  Everything is plain Python integers and coefficient lists; it is meant for k = 4..6.

Pins (tests/test_proof_pins.py): under the deterministic RNG stream of SURVEY.md App. B.2 the proof of
`{"x":6,"y":9,"constant":7,"z":2923}` at k = 4 must reproduce the challenge checkpoints of App. B.5 and the
proof sha256 of App. B.2.  The verifying-key digest `transcript_repr` is taken from App. A.6 (it is Blake2b over
the Rust `{:?}` rendering of the pinned vk; re-deriving that string is not attempted here).

The MSM and NTT calls go through a small backend object so that the same prover logic can run on the CPU oracle
(this file) or on the GPU library (halo2_prover_amd/prover.py uses its own copy of the host logic).
"""
import hashlib

import pyref as R

FR = R.BN_FR
P = FR.p
DELTA = pow(FR.gen, 1 << FR.S, P)  # 7^(2^28): generator of the "column shift" cosets of the permutation argument


# ---------------------------------------------------------------- polynomial helpers (coefficient lists) ----
def padd(a, b):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % P for i in range(n)]


def psub(a, b):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % P for i in range(n)]


def pscale(a, c):
    return [x * c % P for x in a]


def pmul(a, b):
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % P
    return out


def peval(a, x):
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % P
    return acc


def protate(a, w):
    """f(w * X): coefficient i times w^i"""
    out, cur = [], 1
    for c in a:
        out.append(c * cur % P)
        cur = cur * w % P
    return out


def pdiv_linear(a, z):
    """(a(X) - a(z)) / (X - z) by synthetic division (kate_division)"""
    q = [0] * (len(a) - 1)
    acc = 0
    for i in range(len(a) - 1, 0, -1):
        acc = (a[i] + acc * z) % P
        q[i - 1] = acc
    return q


def pdiv_vanishing(a, n):
    """a / (X^n - 1), exact"""
    a = list(a)
    q = [0] * max(0, len(a) - n)
    for i in range(len(a) - 1, n - 1, -1):
        c = a[i]
        q[i - n] = c
        a[i] = 0
        a[i - n] = (a[i - n] + c) % P
    assert not any(a), "quotient by the vanishing polynomial is not exact"
    return q


# ---------------------------------------------------------------- transcript + RNG ---------------------------
class Blake2bTranscript:
    """Blake2bWrite<Vec<u8>, G1Affine, Challenge255>."""

    def __init__(self):
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = b""

    def common_scalar(self, s):
        self.h.update(b"\x02" + int(s % P).to_bytes(32, "little"))

    def common_point(self, pt):
        x, y = pt if pt is not None else (0, 0)
        self.h.update(b"\x01" + x.to_bytes(32, "little") + y.to_bytes(32, "little"))

    def write_scalar(self, s):
        self.common_scalar(s)
        self.proof += int(s % P).to_bytes(32, "little")

    def write_point(self, pt):
        self.common_point(pt)
        self.proof += R.BN254.compress(pt)

    def squeeze(self):
        self.h.update(b"\x00")
        return int.from_bytes(self.h.copy().digest(), "little") % P


def chacha20_block(key_words, counter):
    def rotl(v, c):
        return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))

    def qr(s, a, b, c, d):
        s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = rotl(s[d] ^ s[a], 16)
        s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = rotl(s[b] ^ s[c], 12)
        s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = rotl(s[d] ^ s[a], 8)
        s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = rotl(s[b] ^ s[c], 7)

    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + \
           [counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, 0, 0]
    s = list(init)
    for _ in range(10):
        qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15)
        qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14)
    return [(s[i] + init[i]) & 0xFFFFFFFF for i in range(16)]


class ChaCha20Rng:
    """rand_chacha 0.3.1 ChaCha20Rng::from_seed: 64-bit block counter from 0, stream 0."""

    def __init__(self, seed32):
        self.key = [int.from_bytes(seed32[4 * i:4 * i + 4], "little") for i in range(8)]
        self.counter = 0
        self.buf = []

    def next_u32(self):
        if not self.buf:
            self.buf = chacha20_block(self.key, self.counter)
            self.counter += 1
        return self.buf.pop(0)

    def fr_random(self):
        v = 0
        for i in range(16):
            v |= self.next_u32() << (32 * i)
        return v % P


# ---------------------------------------------------------------- circuit description -----------------------
class ArithmeticCircuit:
    """/root/reference/circuits/src/arithmetic_circuit.rs: 3 advice (l, r, o), 5 fixed (sm, sl, sr, so, sc in
    creation order :196-200), 1 instance column; one degree-3 gate; equality on l, r, o, PI."""

    num_advice, num_fixed, num_instance = 3, 5, 1
    degree = 3                               # max(gate degree 3, permutation argument 3)
    SM, SL, SR, SO, SC = 0, 1, 2, 3, 4
    # permutation columns in enable_equality order (:192-194, :203)
    perm_columns = [("advice", 0), ("advice", 1), ("advice", 2), ("instance", 0)]
    advice_queries = [(0, 0), (1, 0), (2, 0)]            # (column, rotation), first-use order
    fixed_queries = [(1, 0), (2, 0), (3, 0), (0, 0), (4, 0)]   # sl, sr, so, sm, sc (:210-214)

    def __init__(self, x, y, constant):
        self.x, self.y, self.constant = x, y, constant

    def blinding_factors(self):
        return max(3, 1) + 2                 # max distinct queries of any advice column is 1

    def fixed_columns(self, n):
        f = [[0] * n for _ in range(5)]
        for row in (0, 1, 2):                # three `mul` regions (:243-251)
            f[self.SM][row] = 1
            f[self.SO][row] = 1
        f[self.SL][3] = f[self.SR][3] = f[self.SO][3] = 1   # the `add` region (:256-259)
        return f

    def witness(self, n):
        x, y, c = self.x, self.y, self.constant
        xx, yy = x * x % P, y * y % P
        l = [x, y, xx, xx * yy % P]
        r = [x, y, yy, c]
        o = [xx, yy, xx * yy % P, (xx * yy + c) % P]
        return [col + [0] * (n - 4) for col in (l, r, o)]

    def copies(self):
        A = lambda c, r: (("advice", c), r)  # noqa: E731
        I = lambda r: (("instance", 0), r)   # noqa: E731
        return [(A(0, 0), A(1, 0)), (A(0, 1), A(1, 1)), (A(2, 0), A(0, 2)), (A(2, 1), A(1, 2)),
                (A(2, 2), A(0, 3)), (A(1, 3), I(0)), (A(2, 3), I(1))]

    def gate_polys(self, adv, fix, inst, rot):
        """gate polynomials on coefficient-form columns (:216): l*sl + r*sr + l*r*sm + (o*so*(-1)) + sc"""
        l, r, o = adv
        sm, sl, sr, so, sc = fix
        t = padd(pmul(l, sl), pmul(r, sr))
        t = padd(t, pmul(pmul(l, r), sm))
        t = padd(t, pscale(pmul(o, so), P - 1))
        return [padd(t, sc)]


# ---------------------------------------------------------------- CPU backend --------------------------------
class OracleBackend:
    """MSM / NTT through the pinned CPU oracle (oracle/h2_oracle.c) -- ints in, ints out."""

    def __init__(self, params_bytes):
        import numpy as np
        import oracle_lib as O
        self.np, self.O = np, O
        k = int.from_bytes(params_bytes[:4], "little")
        n = 1 << k
        self.k, self.n = k, n
        self.g = np.frombuffer(params_bytes, dtype=np.uint64, count=8 * n, offset=4).reshape(n, 8)
        self.g_lagrange = np.frombuffer(params_bytes, dtype=np.uint64, count=8 * n, offset=4 + 64 * n).reshape(n, 8)

    def _scalars(self, vals):
        return self.np.array([FR.limbs(v) for v in vals], dtype=self.np.uint64)

    def _point(self, aff):
        x = R.BN_FQ.from_mont(self.O.limbs_to_int(aff[:4]))
        y = R.BN_FQ.from_mont(self.O.limbs_to_int(aff[4:]))
        return None if x == 0 and y == 0 else (x, y)

    def commit(self, coeffs):
        c = list(coeffs) + [0] * (self.n - len(coeffs))
        assert len(c) == self.n
        return self._point(self.O.to_affine(0, self.O.best_multiexp(0, self._scalars(c), self.g)))

    def commit_lagrange(self, values):
        assert len(values) == self.n
        return self._point(self.O.to_affine(0, self.O.best_multiexp(0, self._scalars(values), self.g_lagrange)))

    def lagrange_to_coeff(self, values):
        k = self.k
        out = self.O.best_fft(1, self._scalars(values), self.np.array(FR.limbs(pow(FR.omega(k), -1, P)), dtype=self.np.uint64), k)
        ninv = pow(self.n, -1, P)
        return [FR.from_mont(self.O.limbs_to_int(out[4 * i:4 * i + 4])) * ninv % P for i in range(self.n)]


# ---------------------------------------------------------------- keygen -------------------------------------
def permutation_mapping(circuit, n):
    """Assembly::copy union-find of halo2_proofs/src/plonk/permutation/keygen.rs."""
    cols = circuit.perm_columns
    idx = {c: i for i, c in enumerate(cols)}
    mapping = [[(c, r) for r in range(n)] for c in range(len(cols))]
    aux = [[(c, r) for r in range(n)] for c in range(len(cols))]
    sizes = [[1] * n for _ in cols]
    for (lc, lr), (rc, rr) in circuit.copies():
        left, right = (idx[lc], lr), (idx[rc], rr)
        if aux[left[0]][left[1]] == aux[right[0]][right[1]]:
            continue
        lcy, rcy = aux[left[0]][left[1]], aux[right[0]][right[1]]
        if sizes[lcy[0]][lcy[1]] < sizes[rcy[0]][rcy[1]]:
            lcy, rcy = rcy, lcy
        sizes[lcy[0]][lcy[1]] += sizes[rcy[0]][rcy[1]]
        i = rcy
        while True:
            aux[i[0]][i[1]] = lcy
            i = mapping[i[0]][i[1]]
            if i == rcy:
                break
        mapping[left[0]][left[1]], mapping[right[0]][right[1]] = mapping[right[0]][right[1]], mapping[left[0]][left[1]]
    return mapping


class ProvingKey:
    def __init__(self, circuit, backend, transcript_repr):
        n, k = backend.n, backend.k
        self.circuit, self.n, self.k = circuit, n, k
        self.omega = FR.omega(k)
        self.transcript_repr = transcript_repr
        self.fixed_values = circuit.fixed_columns(n)
        self.fixed_polys = [backend.lagrange_to_coeff(v) for v in self.fixed_values]
        self.fixed_commitments = [backend.commit_lagrange(v) for v in self.fixed_values]
        mapping = permutation_mapping(circuit, n)
        self.sigma_values = [[pow(DELTA, mapping[j][i][0], P) * pow(self.omega, mapping[j][i][1], P) % P
                              for i in range(n)] for j in range(len(circuit.perm_columns))]
        self.sigma_polys = [backend.lagrange_to_coeff(v) for v in self.sigma_values]
        self.sigma_commitments = [backend.commit_lagrange(v) for v in self.sigma_values]


# ---------------------------------------------------------------- create_proof (GWC) -------------------------
def create_proof(pk, backend, instances, rng, trace=None):
    """SURVEY.md App. A.4 (phase order, RNG schedule), A.7 (quotient, GWC).  `rng` follows SURVEY App. B.2's
    stream interface (fr_random(field), fill(nbytes)).  Returns the proof bytes."""
    c, n, k, omega = pk.circuit, pk.n, pk.k, pk.omega
    bf = c.blinding_factors()
    d = c.degree
    tr = Blake2bTranscript()
    trace = trace if trace is not None else {}

    # 0-1: vk digest, instance values (KZG: not committed)
    tr.common_scalar(pk.transcript_repr)
    inst_values = []
    for col in instances:
        for v in col:
            tr.common_scalar(v)
        inst_values.append(list(col) + [0] * (n - len(col)))
    inst_polys = [backend.lagrange_to_coeff(v) for v in inst_values]

    # 2: advice columns, blinded rows, commitments, unused blinds
    adv_values = c.witness(n)
    for col in adv_values:
        for row in range(n - (bf + 1), n):
            col[row] = rng.fr_random(FR)
    for _ in adv_values:
        rng.fr_random(FR)                     # Blind (unused by KZG, but drawn)
    adv_commit = [backend.commit_lagrange(v) for v in adv_values]
    for pt in adv_commit:
        tr.write_point(pt)
    adv_polys = [backend.lagrange_to_coeff(v) for v in adv_values]

    # 3: challenges
    theta = tr.squeeze(); beta = tr.squeeze(); gamma = tr.squeeze()
    trace.update(theta=theta, beta=beta, gamma=gamma)

    # 4: permutation argument
    def column_values(col):
        kind, i = col
        return {"advice": adv_values, "fixed": pk.fixed_values, "instance": inst_values}[kind][i]

    def column_poly(col):
        kind, i = col
        return {"advice": adv_polys, "fixed": pk.fixed_polys, "instance": inst_polys}[kind][i]

    chunk = d - 2
    sets = [list(range(s, min(s + chunk, len(c.perm_columns)))) for s in range(0, len(c.perm_columns), chunk)]
    z_values, last_z = [], 1
    for cols in sets:
        m = [1] * n
        for i in range(n):
            num = den = 1
            for j in cols:
                v = column_values(c.perm_columns[j])[i]
                num = num * ((pow(DELTA, j, P) * pow(omega, i, P) % P) * beta + gamma + v) % P
                den = den * (beta * pk.sigma_values[j][i] + gamma + v) % P
            m[i] = num * pow(den, -1, P) % P
        z = [last_z]
        for row in range(1, n):
            z.append(z[row - 1] * m[row - 1] % P)
        for row in range(n - bf, n):
            z[row] = rng.fr_random(FR)
        last_z = z[n - bf - 1]
        rng.fr_random(FR)                     # Blind
        z_values.append(z)
    z_commit = [backend.commit_lagrange(z) for z in z_values]
    for pt in z_commit:
        tr.write_point(pt)
    z_polys = [backend.lagrange_to_coeff(z) for z in z_values]

    # 5: vanishing argument's random polynomial (one thread chunk: one 32-byte seed)
    chacha = ChaCha20Rng(rng.fill(32))
    random_poly = [chacha.fr_random() for _ in range(n)]
    rng.fr_random(FR)                         # Blind
    tr.write_point(backend.commit(random_poly))

    # 6: quotient
    y = tr.squeeze()
    trace.update(y=y)
    fixed_polys = pk.fixed_polys

    def lagrange_basis(rows):
        v = [0] * n
        for r in rows:
            v[r] = 1
        return backend.lagrange_to_coeff(v)

    l0 = lagrange_basis([0])
    l_last = lagrange_basis([n - bf - 1])
    l_blind = lagrange_basis(range(n - bf, n))
    l_active = psub(psub([1], l_last), l_blind)
    terms = list(c.gate_polys(adv_polys, fixed_polys, inst_polys, None))
    terms.append(pmul(l0, psub([1], z_polys[0])))
    zl = z_polys[-1]
    terms.append(pmul(l_last, psub(pmul(zl, zl), zl)))
    w_back = pow(omega, -(bf + 1), P)
    for i in range(1, len(sets)):
        terms.append(pmul(l0, psub(z_polys[i], protate(z_polys[i - 1], w_back))))
    for i, cols in enumerate(sets):
        left = protate(z_polys[i], omega)
        right = z_polys[i]
        for j in cols:
            v = column_poly(c.perm_columns[j])
            left = pmul(left, padd(padd(v, pscale(pk.sigma_polys[j], beta)), [gamma]))
            right = pmul(right, padd(padd(v, [0, pow(DELTA, j, P) * beta % P]), [gamma]))
        terms.append(pmul(l_active, psub(left, right)))
    numer = []
    for t in terms:
        numer = padd(pscale(numer, y), t)
    h = pdiv_vanishing(numer, n)
    h = h + [0] * (n * (d - 1) - len(h))
    assert len(h) == n * (d - 1), "quotient degree too large"
    h_pieces = [h[i * n:(i + 1) * n] for i in range(d - 1)]
    for piece in h_pieces:
        tr.write_point(backend.commit(piece))
    for _ in h_pieces:
        rng.fr_random(FR)                     # Blinds

    # 7: evaluations
    x = tr.squeeze()
    trace.update(x=x)
    xn = pow(x, n, P)
    for col, rot in c.advice_queries:
        tr.write_scalar(peval(adv_polys[col], x * pow(omega, rot, P) % P))
    for col, rot in c.fixed_queries:
        tr.write_scalar(peval(fixed_polys[col], x * pow(omega, rot, P) % P))
    tr.write_scalar(peval(random_poly, x))
    for s in pk.sigma_polys:
        tr.write_scalar(peval(s, x))
    for i, zp in enumerate(z_polys):
        tr.write_scalar(peval(zp, x))
        tr.write_scalar(peval(zp, x * omega % P))
        if i + 1 < len(z_polys):
            tr.write_scalar(peval(zp, x * w_back % P))

    # 8: GWC multiopen
    v = tr.squeeze()
    trace.update(v=v)
    h_poly = []
    for piece in reversed(h_pieces):
        h_poly = padd(pscale(h_poly, xn), piece)
    queries = [(x * pow(omega, rot, P) % P, adv_polys[col]) for col, rot in c.advice_queries]
    for zp in z_polys:
        queries.append((x, zp))
        queries.append((x * omega % P, zp))
    for zp in reversed(z_polys[:-1]):
        queries.append((x * w_back % P, zp))
    queries += [(x * pow(omega, rot, P) % P, fixed_polys[col]) for col, rot in c.fixed_queries]
    queries += [(x, s) for s in pk.sigma_polys]
    queries.append((x, h_poly))
    queries.append((x, random_poly))
    points = []
    for pt, _ in queries:
        if pt not in points:
            points.append(pt)
    for pt in points:
        acc, vp = [], 1
        for qpt, poly in queries:
            if qpt == pt:
                acc = padd(acc, pscale(poly, vp))
                vp = vp * v % P
        tr.write_point(backend.commit(pdiv_linear(acc, pt)))
    return tr.proof


# transcript_repr of the pinned verifying keys under the SURVEY App. B.2 stream
TRANSCRIPT_REPR = {
    ("arithmetic", 4): 0x29FDBC4FAA50E4E635114C86B4655A8CC4C5B56751D66E7F06C91C80076930F9,
}
