"""Big-integer restatement of the arithmetic under the MSM/NTT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``halo2_prover_amd/`` may import this
module; it is used by ``tests/``, by ``tools/`` fixture generators and to pin the C
oracle (``oracle/h2_oracle.c``).  It is plain Python integers, so it is slow and
only meant for small sizes (n <= 2^11 or so).

What it restates (upstream code is NOT in /root/reference -- the hot path lives in the
git dependency halo2_proofs@6b43b6b / halo2curves 0.3.2 / pasta_curves 0.5.1, pinned at
/root/reference/circuits/Cargo.lock:836-838,854-856,1126-1128):

* 4x64-bit Montgomery fields (R = 2^256) for BN254 Fq/Fr and Pasta Fp/Fq
  (constants from SURVEY.md section 8(a) "Constants").
* a = 0 short Weierstrass curves: BN254 G1 (y^2=x^3+3, gen (1,2)), Pallas/Vesta
  (y^2=x^3+5, gen (-1,2)).
* ``best_multiexp`` result semantics: sum_i coeffs[i] * bases[i].
* ``best_fft`` result semantics: A[i] = sum_j a[j] w^(ij), natural
  order in and out, unscaled.
* ``ParamsKZG::new(k)`` + ``write`` so that the params
  sha256 values recorded in SURVEY.md App. B.2 can be re-derived.
* Poseidon (Grain LFSR, Cauchy MDS, permutation, ConstantLength sponge) following
  /root/reference/circuits/src/poseidon/primitives/grain.rs:52-167, mds.rs:5-102 and
  primitives.rs:57-132,204-390 -- used only as a known-answer harness for field arithmetic.
"""
import hashlib

MASK64 = (1 << 64) - 1
R_BITS = 256


class Field:
    """Prime field description; elements are plain ints in [0, p)."""

    def __init__(self, name, p, gen, two_adicity):
        self.name = name
        self.p = p
        self.gen = gen  # multiplicative generator
        self.S = two_adicity
        self.R = (1 << R_BITS) % p
        self.R2 = (self.R * self.R) % p
        self.Rinv = pow(self.R, -1, p)
        self.inv64 = (-pow(p, -1, 1 << 64)) & MASK64
        self.inv32 = (-pow(p, -1, 1 << 32)) & 0xFFFFFFFF
        t = (p - 1) >> two_adicity
        assert (p - 1) == t << two_adicity and t & 1
        self.root_of_unity = pow(gen, t, p)  # primitive 2^S-th root
        self.num_bits = p.bit_length()

    # representation helpers -------------------------------------------------
    def to_mont(self, x):
        return (x * self.R) % self.p

    def from_mont(self, x):
        return (x * self.Rinv) % self.p

    def limbs(self, x):
        """canonical int -> 4 little-endian u64 limbs of its Montgomery form."""
        m = self.to_mont(x)
        return [(m >> (64 * i)) & MASK64 for i in range(4)]

    def mont_bytes(self, x):
        return self.to_mont(x).to_bytes(32, "little")

    def from_mont_bytes(self, b):
        return self.from_mont(int.from_bytes(b, "little"))

    def omega(self, log_n):
        """w_k = ROOT_OF_UNITY^(2^(S-k))."""
        assert log_n <= self.S
        return pow(self.root_of_unity, 1 << (self.S - log_n), self.p)

    def inv(self, x):
        return pow(x, -1, self.p)

    def sqrt(self, a):
        """Tonelli-Shanks; returns None when a is a non-residue."""
        p = self.p
        a %= p
        if a == 0:
            return 0
        if pow(a, (p - 1) // 2, p) != 1:
            return None
        if p % 4 == 3:
            return pow(a, (p + 1) // 4, p)
        q, s = p - 1, 0
        while q % 2 == 0:
            q //= 2
            s += 1
        z = self.gen  # a generator is a non-residue
        m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
        while t != 1:
            i, t2 = 0, t
            while t2 != 1:
                t2 = t2 * t2 % p
                i += 1
            b = pow(c, 1 << (m - i - 1), p)
            m, c = i, b * b % p
            t, r = t * c % p, r * b % p
        return r


# ---- the four fields ---------------------------------------------------------------
P_BN = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_BN = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
P_PASTA = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q_PASTA = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001

BN_FQ = Field("bn254_fq", P_BN, 3, 1)
BN_FR = Field("bn254_fr", R_BN, 7, 28)
PA_FP = Field("pasta_fp", P_PASTA, 5, 32)
PA_FQ = Field("pasta_fq", Q_PASTA, 5, 32)
FIELDS = {f.name: f for f in (BN_FQ, BN_FR, PA_FP, PA_FQ)}


class Curve:
    """y^2 = x^3 + b over `base`, prime order group with scalar field `scalar`."""

    def __init__(self, name, cid, base, scalar, b, gen):
        self.name, self.cid, self.base, self.scalar, self.b, self.gen = name, cid, base, scalar, b, gen

    def is_on_curve(self, P):
        if P is None:
            return True
        x, y = P
        p = self.base.p
        return (y * y - x * x * x - self.b) % p == 0

    def neg(self, P):
        return None if P is None else (P[0], (-P[1]) % self.base.p)

    def add(self, P, Q):
        p = self.base.p
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def mul(self, k, P):
        k %= self.scalar.p
        acc = None
        while k:
            if k & 1:
                acc = self.add(acc, P)
            P = self.add(P, P)
            k >>= 1
        return acc

    def msm(self, scalars, points):
        """Result semantics of best_multiexp."""
        assert len(scalars) == len(points)
        acc = None
        for s, P in zip(scalars, points):
            if s % self.scalar.p and P is not None:
                acc = self.add(acc, self.mul(s, P))
        return acc

    # affine (x,y) Montgomery limbs, identity = (0,0)  
    def affine_bytes(self, P):
        if P is None:
            return bytes(64)
        return self.base.mont_bytes(P[0]) + self.base.mont_bytes(P[1])

    def affine_from_bytes(self, b):
        x = self.base.from_mont_bytes(b[:32])
        y = self.base.from_mont_bytes(b[32:64])
        return None if (x == 0 and y == 0) else (x, y)

    def compress(self, P):
        """Proof wire format: canonical x LE, bit 6 of byte 31 = y&1."""
        if P is None:
            return bytes(32)
        b = bytearray(P[0].to_bytes(32, "little"))
        b[31] |= (P[1] & 1) << 6
        return bytes(b)

    def point_from_x(self, x):
        """try-and-increment base generator helper: smallest x' >= x on the curve, even y."""
        p = self.base.p
        while True:
            y = self.base.sqrt((x * x * x + self.b) % p)
            if y is not None:
                if y & 1:
                    y = p - y
                return (x % p, y)
            x += 1


BN254 = Curve("bn254", 0, BN_FQ, BN_FR, 3, (1, 2))
PALLAS = Curve("pallas", 1, PA_FP, PA_FQ, 5, (P_PASTA - 1, 2))
VESTA = Curve("vesta", 2, PA_FQ, PA_FP, 5, (Q_PASTA - 1, 2))
CURVES = {c.name: c for c in (BN254, PALLAS, VESTA)}


# ---- NTT ---------------------------------------------------------------------------
def dft_naive(a, omega, p):
    n = len(a)
    return [sum(a[j] * pow(omega, i * j, p) for j in range(n)) % p for i in range(n)]


def best_fft(a, omega, log_n, p):
    """Iterative form of SURVEY.md App. A.2 (bit-reverse, radix-2 DIT). Returns new list."""
    n = 1 << log_n
    assert len(a) == n
    a = list(a)
    for k in range(n):
        rk = int(format(k, "0%db" % log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), p)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % p
                a[k + j + m] = (a[k + j] - t) % p
                a[k + j] = (a[k + j] + t) % p
                w = w * w_m % p
        m *= 2
    return a


def group_fft(curve, pts, omega, log_n):
    """best_fft over group elements (FftGroup for G1). O(n log n) adds."""
    n = 1 << log_n
    q = curve.scalar.p
    a = list(pts)
    for k in range(n):
        rk = int(format(k, "0%db" % log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), q)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = curve.mul(w, a[k + j + m])
                a[k + j + m] = curve.add(a[k + j], curve.neg(t))
                a[k + j] = curve.add(a[k + j], t)
                w = w * w_m % q
        m *= 2
    return a


# ---- deterministic RNG stream of SURVEY.md App. B.2 ------------------------------
class SurveyStream:
    """call i fills its buffer with SHA256("seed0-" + str(counter)) digests."""

    def __init__(self, start=0):
        self.counter = start

    def fill(self, nbytes):
        out = b""
        while len(out) < nbytes:
            out += hashlib.sha256(b"seed0-%d" % self.counter).digest()
            self.counter += 1
        return out[:nbytes]

    def next_u64(self):
        return int.from_bytes(self.fill(8), "little")

    def fr_random(self, field):
        """Fr::random = 8 x next_u64 (LE) -> 512-bit integer mod r."""
        v = 0
        for i in range(8):
            v |= self.next_u64() << (64 * i)
        return v % field.p


# ---- BN254 G2 (only to round-trip the 256-byte tail of the params file) ----------
class Fq2:
    p = P_BN

    @staticmethod
    def add(a, b):
        return ((a[0] + b[0]) % P_BN, (a[1] + b[1]) % P_BN)

    @staticmethod
    def sub(a, b):
        return ((a[0] - b[0]) % P_BN, (a[1] - b[1]) % P_BN)

    @staticmethod
    def mul(a, b):  # u^2 = -1
        return ((a[0] * b[0] - a[1] * b[1]) % P_BN, (a[0] * b[1] + a[1] * b[0]) % P_BN)

    @staticmethod
    def inv(a):
        d = pow(a[0] * a[0] + a[1] * a[1], -1, P_BN)
        return (a[0] * d % P_BN, (-a[1]) * d % P_BN)


G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def g2_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if Fq2.add(y1, y2) == (0, 0):
            return None
        three_x2 = Fq2.mul((3, 0), Fq2.mul(x1, x1))
        lam = Fq2.mul(three_x2, Fq2.inv(Fq2.add(y1, y1)))
    else:
        lam = Fq2.mul(Fq2.sub(y2, y1), Fq2.inv(Fq2.sub(x2, x1)))
    x3 = Fq2.sub(Fq2.sub(Fq2.mul(lam, lam), x1), x2)
    y3 = Fq2.sub(Fq2.mul(lam, Fq2.sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul(k, P):
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, P)
        P = g2_add(P, P)
        k >>= 1
    return acc


def g2_bytes(P):
    (x0, x1), (y0, y1) = P
    return b"".join(BN_FQ.mont_bytes(v) for v in (x0, x1, y0, y1))


def params_kzg_bytes(k, s, g_lagrange_fn=None):
    """ParamsKZG::<Bn256>::new(k).write() for toxic scalar s.

    k:u32 LE || g[0..n) || g_lagrange[0..n) || g2 || s*g2, G1 as raw Montgomery limbs.
    """
    n = 1 << k
    r = R_BN
    g = []
    cur = 1
    for _ in range(n):
        g.append(BN254.mul(cur, BN254.gen))
        cur = cur * s % r
    if g_lagrange_fn is None:
        omega_inv = pow(BN_FR.omega(k), -1, r)
        gl = group_fft(BN254, g, omega_inv, k)
        n_inv = pow(n, -1, r)
        gl = [BN254.mul(n_inv, P) for P in gl]
    else:
        gl = g_lagrange_fn(g)
    out = k.to_bytes(4, "little")
    out += b"".join(BN254.affine_bytes(P) for P in g)
    out += b"".join(BN254.affine_bytes(P) for P in gl)
    out += g2_bytes(G2_GEN) + g2_bytes(g2_mul(s, G2_GEN))
    return out, g, gl


# ---- Poseidon as a field-arithmetic KAT harness ------------------------------------
class Grain:
    """grain.rs:52-137 -- 80-bit LFSR, MSB-first bit handling."""

    def __init__(self, field, t, r_f, r_p, sbox_tag=0):
        self.f = field
        bits = []

        def put(width, value):
            bits.extend(((value >> (width - 1 - i)) & 1) for i in range(width))

        put(2, 1)  # FieldType::PrimeOrder tag
        put(4, sbox_tag)
        put(12, field.num_bits)
        put(12, t)
        put(10, r_f)
        put(10, r_p)
        bits.extend([1] * 30)
        assert len(bits) == 80
        self.state = bits
        for _ in range(160):
            self._raw()

    def _raw(self):
        s = self.state
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        self.state = s[1:] + [nb]
        return nb

    def _bit(self):
        while True:
            if self._raw():
                return self._raw()
            self._raw()

    def _take(self):
        v = 0
        for _ in range(self.f.num_bits):
            v = (v << 1) | self._bit()
        return v

    def next_field_element(self):
        while True:
            v = self._take()
            if v < self.f.p:
                return v

    def next_without_rejection(self):
        return self._take() % self.f.p


def poseidon_constants(field, t, r_f, r_p, secure_mds=0):
    """primitives.rs:57-84 + mds.rs:5-63."""
    p = field.p
    g = Grain(field, t, r_f, r_p)
    rcs = [[g.next_field_element() for _ in range(t)] for _ in range(r_f + r_p)]
    select = secure_mds
    while True:
        while True:
            vals = [g.next_without_rejection() for _ in range(2 * t)]
            if len(set(vals)) == len(vals):
                break
        if select:
            select -= 1
            continue
        xs, ys = vals[:t], vals[t:]
        mds = [[pow(xs[i] + ys[j], -1, p) for j in range(t)] for i in range(t)]
        return rcs, mds


def poseidon_permute(field, state, rcs, mds, r_f, r_p):
    """primitives.rs:87-132, x^5 S-box."""
    p = field.p
    t = len(state)

    def apply_mds(st):
        return [sum(mds[i][j] * st[j] for j in range(t)) % p for i in range(t)]

    st = list(state)
    half = r_f // 2
    for r in range(r_f + r_p):
        rc = rcs[r]
        if r < half or r >= half + r_p:
            st = [pow((st[i] + rc[i]) % p, 5, p) for i in range(t)]
        else:
            st = [(st[i] + rc[i]) % p for i in range(t)]
            st[0] = pow(st[0], 5, p)
        st = apply_mds(st)
    return st


def poseidon_hash_const_len(field, msg, rcs, mds, r_f, r_p, t=3, rate=2):
    """ConstantLength<L> sponge (primitives.rs:204-390): capacity = L * 2^64, zero pad."""
    p = field.p
    L = len(msg)
    state = [0] * t
    state[rate] = (L << 64) % p
    padded = list(msg) + [0] * ((-L) % rate)
    for off in range(0, len(padded), rate):
        for i in range(rate):
            state[i] = (state[i] + padded[off + i]) % p
        state = poseidon_permute(field, state, rcs, mds, r_f, r_p)
    return state[0]


# ---- synthetic inputs (SURVEY.md section 8(d) "Synthetic inputs") ----------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)


def synth_scalar(rng, p):
    """4 limbs, top limb masked to 62 bits, one conditional subtract of the modulus."""
    v = 0
    for i in range(4):
        limb = rng.next()
        if i == 3:
            limb &= (1 << 62) - 1
        v |= limb << (64 * i)
    return v - p if v >= p else v
