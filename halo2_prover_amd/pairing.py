"""BN254 optimal-ate pairing check on the host (Python integers): the last step of KZG verification,
e(left, [s]G2) * e(right, -G2) == 1  (halo2_proofs src/poly/kzg/msm.rs `DualMSM::check`, reached from
/root/reference/circuits/src/utils.rs:125-158 through verify_proof).

Not on the hot path: two Miller loops and one final exponentiation per proof.  The arithmetic follows the published
construction (Vercauteren's optimal ate for Barreto-Naehrig curves, halo2curves 0.3.2 bn256: Fq2 = Fq[u]/(u^2+1),
Fq6 = Fq2[v]/(v^3 - xi) with xi = 9 + u, Fq12 = Fq6[w]/(w^2 - v), D-type twist y^2 = x^3 + 3/xi); nothing in
/root/reference pins it directly ("parity unpinned" for the pairing on its own) -- it is checked by bilinearity and
by accepting the proofs recorded from the reference's build while rejecting corrupted ones (tests/test_verifier.py).
The C++ verifier (csrc/h2_pairing.hpp) is a transcription of this file.
"""
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
BN_X = 4965661367192848881
ATE_LOOP = 6 * BN_X + 2                       # 29793968203157093288


# ---- Fq2: pairs (a0, a1) = a0 + a1 u, u^2 = -1 ---------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_neg(a): return (-a[0] % Q, -a[1] % Q)
def f2_conj(a): return (a[0], -a[1] % Q)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2_sqr(a): return ((a[0] + a[1]) * (a[0] - a[1]) % Q, 2 * a[0] * a[1] % Q)
def f2_scale(a, k): return (a[0] * k % Q, a[1] * k % Q)
def f2_mul_xi(a): return ((9 * a[0] - a[1]) % Q, (a[0] + 9 * a[1]) % Q)      # times xi = 9 + u


def f2_inv(a):
    t = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * t % Q, -a[1] * t % Q)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (9, 1)


# ---- Fq6: triples of Fq2, c0 + c1 v + c2 v^2, v^3 = xi ---------------------------------------------------------
def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_neg(a): return tuple(f2_neg(x) for x in a)


def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0, t1, t2 = f2_mul(a0, b0), f2_mul(a1, b1), f2_mul(a2, b2)
    c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_mul(f2_add(a1, a2), f2_add(b1, b2)), f2_add(t1, t2))))
    c1 = f2_add(f2_sub(f2_mul(f2_add(a0, a1), f2_add(b0, b1)), f2_add(t0, t1)), f2_mul_xi(t2))
    c2 = f2_add(f2_sub(f2_mul(f2_add(a0, a2), f2_add(b0, b2)), f2_add(t0, t2)), t1)
    return (c0, c1, c2)


def f6_mul_v(a): return (f2_mul_xi(a[2]), a[0], a[1])                          # times v


def f6_inv(a):
    a0, a1, a2 = a
    c0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    c1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    c2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    t = f2_inv(f2_add(f2_mul(a0, c0), f2_mul_xi(f2_add(f2_mul(a2, c1), f2_mul(a1, c2)))))
    return (f2_mul(c0, t), f2_mul(c1, t), f2_mul(c2, t))


F6_ZERO, F6_ONE = (F2_ZERO, F2_ZERO, F2_ZERO), (F2_ONE, F2_ZERO, F2_ZERO)


# ---- Fq12: pairs of Fq6, c0 + c1 w, w^2 = v --------------------------------------------------------------------
def f12_mul(a, b):
    t0, t1 = f6_mul(a[0], b[0]), f6_mul(a[1], b[1])
    c1 = f6_sub(f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1])), f6_add(t0, t1))
    return (f6_add(t0, f6_mul_v(t1)), c1)


def f12_sqr(a): return f12_mul(a, a)
def f12_conj(a): return (a[0], f6_neg(a[1]))                                   # the p^6 Frobenius


def f12_inv(a):
    t = f6_inv(f6_sub(f6_mul(a[0], a[0]), f6_mul_v(f6_mul(a[1], a[1]))))
    return (f6_mul(a[0], t), f6_neg(f6_mul(a[1], t)))


def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


F12_ONE = (F6_ONE, F6_ZERO)

# Frobenius constants for the twist points: pi(x, y) = (conj(x) xi^((p-1)/3), conj(y) xi^((p-1)/2))
_G12 = f2_pow(XI, (Q - 1) // 3)
_G13 = f2_pow(XI, (Q - 1) // 2)
_G22 = f2_pow(XI, (Q * Q - 1) // 3)
_G23 = f2_pow(XI, (Q * Q - 1) // 2)
TWIST_B = f2_mul((3, 0), f2_inv(XI))


def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sqr(y) == f2_add(f2_mul(f2_sqr(x), x), TWIST_B)


def _line(t, q, p):
    """the line through the untwisted points t, q (t == q: the tangent) evaluated at p = (xP, yP) in E(Fq), and
    t + q on the twist.  psi(x', y') = (x' w^2, y' w^3): l(P) = yP - lambda xP w + (lambda x1 - y1) w^3."""
    (x1, y1), (x2, y2) = t, q
    if t == q:
        lam = f2_mul(f2_scale(f2_sqr(x1), 3), f2_inv(f2_scale(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    xp, yp = p
    c0 = ((yp % Q, 0), F2_ZERO, F2_ZERO)
    c1 = (f2_neg(f2_scale(lam, xp)), f2_sub(f2_mul(lam, x1), y1), F2_ZERO)
    return (c0, c1), (x3, y3)


def miller_loop(p, q):
    """f_{6x+2, Q}(P) times the two Frobenius lines; p in G1 (affine ints), q in G2 (affine Fq2 pairs); None = identity"""
    if p is None or q is None:
        return F12_ONE
    f, t = F12_ONE, q
    for bit in bin(ATE_LOOP)[3:]:
        l, t = _line(t, t, p)
        f = f12_mul(f12_sqr(f), l)
        if bit == "1":
            l, t = _line(t, q, p)
            f = f12_mul(f, l)
    q1 = (f2_mul(f2_conj(q[0]), _G12), f2_mul(f2_conj(q[1]), _G13))
    q2 = (f2_mul(q[0], _G22), f2_neg(f2_mul(q[1], _G23)))                       # -pi^2(Q)
    l, t = _line(t, q1, p)
    f = f12_mul(f, l)
    l, t = _line(t, q2, p)
    return f12_mul(f, l)


def final_exponentiation(f):
    """f^((p^12 - 1) / r): the p^6 - 1 part by conjugation and one inversion, the rest by plain exponentiation"""
    f = f12_mul(f12_conj(f), f12_inv(f))
    return f12_pow(f, (Q ** 6 + 1) // R)


def pairing(p, q):
    return final_exponentiation(miller_loop(p, q))


def pairing_check(pairs):
    """prod e(P_i, Q_i) == 1 with one shared final exponentiation"""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(p, q))
    return final_exponentiation(f) == F12_ONE


def g2_neg(pt):
    return None if pt is None else (pt[0], f2_neg(pt[1]))
