"""BN254 ("BN256" in pairing_ce) arithmetic on Python big integers.

TEST INFRASTRUCTURE ONLY.  Nothing under zinc_amd/ may import this module; it exists so
tests, the smoke check and the CPU-baseline leg of the benchmark have an independent
statement of the arithmetic to compare the HIP path against.

PARITY STATUS: "parity unpinned".  The prover this restates lives in third-party crates that
are not part of the reference tree (pairing_ce 0.17.0 / ff_ce 0.6.0, pinned by
Cargo.lock:2514-2517 and Cargo.lock:1363-1366 of the reference and reached from
zinc-vm/Cargo.toml:29).  The reference holds no golden vectors for this path, so what pins
this file is: the public BN254 constants (checked in tests/test_oracle_constants.py), the
curve equations, bilinearity of the pairing, and the Groth16 verification equation.

Representation: field elements are plain ints in [0, p); Fq2 elements are (c0, c1) tuples
meaning c0 + c1*u with u^2 = -1; curve points are Jacobian triples (X, Y, Z) over the base
field of the group, with Z == 0 (or zero tuple) meaning the identity.
"""

# --------------------------------------------------------------------------- constants
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq
MONT_R = 1 << 256
FR_S = 28
FR_GEN = 7
FR_ROOT_OF_UNITY = pow(FR_GEN, (R_MOD - 1) >> FR_S, R_MOD)
FR_INV64 = (-pow(R_MOD, -1, 1 << 64)) % (1 << 64)
FQ_INV64 = (-pow(Q_MOD, -1, 1 << 64)) % (1 << 64)
FR_INV32 = FR_INV64 & 0xFFFFFFFF
FQ_INV32 = FQ_INV64 & 0xFFFFFFFF

G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
B_G1 = 3


def to_mont(x, p):
    return (x * MONT_R) % p


def from_mont(x, p):
    return (x * pow(MONT_R, -1, p)) % p


def limbs64(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs64(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


# --------------------------------------------------------------------------- Fq2
def fq2_add(a, b):
    return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)


def fq2_sub(a, b):
    return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)


def fq2_neg(a):
    return ((-a[0]) % Q_MOD, (-a[1]) % Q_MOD)


def fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def fq2_sqr(a):
    return fq2_mul(a, a)


def fq2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % Q_MOD, -1, Q_MOD)
    return ((a[0] * n) % Q_MOD, (-a[1] * n) % Q_MOD)


def fq2_scale(a, k):
    return ((a[0] * k) % Q_MOD, (a[1] * k) % Q_MOD)


FQ2_ZERO = (0, 0)
FQ2_ONE = (1, 0)
# b' = 3 / (9 + u): the sextic twist coefficient of G2
B_G2 = fq2_scale(fq2_inv((9, 1)), 3)


class _Fq:
    """Field-operation bundle so the curve code below is written once."""
    zero = 0
    one = 1
    add = staticmethod(lambda a, b: (a + b) % Q_MOD)
    sub = staticmethod(lambda a, b: (a - b) % Q_MOD)
    neg = staticmethod(lambda a: (-a) % Q_MOD)
    mul = staticmethod(lambda a, b: (a * b) % Q_MOD)
    inv = staticmethod(lambda a: pow(a, -1, Q_MOD))
    is_zero = staticmethod(lambda a: a % Q_MOD == 0)
    b = B_G1


class _Fq2:
    zero = FQ2_ZERO
    one = FQ2_ONE
    add = staticmethod(fq2_add)
    sub = staticmethod(fq2_sub)
    neg = staticmethod(fq2_neg)
    mul = staticmethod(fq2_mul)
    inv = staticmethod(fq2_inv)
    is_zero = staticmethod(lambda a: a[0] % Q_MOD == 0 and a[1] % Q_MOD == 0)
    b = B_G2


class Group:
    """Short-Weierstrass y^2 = x^3 + b in Jacobian coordinates over field bundle F."""

    def __init__(self, F, gen):
        self.F = F
        self.gen_affine = gen
        self.identity = (F.zero, F.one, F.zero)
        self.gen = (gen[0], gen[1], F.one)

    def is_identity(self, P):
        return self.F.is_zero(P[2])

    def from_affine(self, a):
        if a is None:
            return self.identity
        return (a[0], a[1], self.F.one)

    def to_affine(self, P):
        F = self.F
        if self.is_identity(P):
            return None
        zi = F.inv(P[2])
        zi2 = F.mul(zi, zi)
        return (F.mul(P[0], zi2), F.mul(P[1], F.mul(zi2, zi)))

    def on_curve_affine(self, a):
        F = self.F
        if a is None:
            return True
        x, y = a
        return F.sub(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), F.b)) == F.zero

    def neg(self, P):
        return (P[0], self.F.neg(P[1]), P[2])

    def double(self, P):
        F = self.F
        if self.is_identity(P):
            return P
        X, Y, Z = P
        A = F.mul(X, X)
        B = F.mul(Y, Y)
        C = F.mul(B, B)
        t = F.add(X, B)
        D = F.sub(F.sub(F.mul(t, t), A), C)
        D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fv = F.mul(E, E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(Y, Z)
        Z3 = F.add(Z3, Z3)
        return (X3, Y3, Z3)

    def add(self, P, Q):
        F = self.F
        if self.is_identity(P):
            return Q
        if self.is_identity(Q):
            return P
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        Z1Z1 = F.mul(Z1, Z1)
        Z2Z2 = F.mul(Z2, Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
        S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        if U1 == U2:
            if S1 == S2:
                return self.double(P)
            return self.identity
        H = F.sub(U2, U1)
        Rr = F.sub(S2, S1)
        HH = F.mul(H, H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.mul(Rr, Rr), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def mul(self, P, k):
        k = int(k)
        if k < 0:
            return self.mul(self.neg(P), -k)
        acc = self.identity
        for bit in bin(k)[2:] if k else "":
            acc = self.double(acc)
            if bit == "1":
                acc = self.add(acc, P)
        return acc

    def eq(self, P, Q):
        return self.to_affine(P) == self.to_affine(Q)

    def sum(self, pts):
        acc = self.identity
        for p in pts:
            acc = self.add(acc, p)
        return acc

    def batch_to_affine(self, pts):
        """Montgomery's trick; identities map to None."""
        F = self.F
        live = [i for i, P in enumerate(pts) if not self.is_identity(P)]
        prefix = []  # prefix[k] = product of Z over live[0..k)
        acc = F.one
        for i in live:
            prefix.append(acc)
            acc = F.mul(acc, pts[i][2])
        inv = F.inv(acc) if live else F.one
        out = [None] * len(pts)
        for k in range(len(live) - 1, -1, -1):
            P = pts[live[k]]
            zi = F.mul(inv, prefix[k])
            inv = F.mul(inv, P[2])
            zi2 = F.mul(zi, zi)
            out[live[k]] = (F.mul(P[0], zi2), F.mul(P[1], F.mul(zi2, zi)))
        return out


G1 = Group(_Fq, G1_GEN)
G2 = Group(_Fq2, G2_GEN)


# --------------------------------------------------------------------------- Fq12 and pairing
# Fq12 = Fq[w] / (w^12 - 18 w^6 + 82).  With x = w^6, x^2 - 18x + 82 = 0, i.e. x = 9 + u.
def _f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for i in range(22, 11, -1):
        v = t[i]
        if v:
            t[i - 6] += 18 * v
            t[i - 12] -= 82 * v
    return [v % Q_MOD for v in t[:12]]


F12_ONE = [1] + [0] * 11


def _f12_pow(a, e):
    res = F12_ONE
    base = a
    while e:
        if e & 1:
            res = _f12_mul(res, base)
        base = _f12_mul(base, base)
        e >>= 1
    return res


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def _f12_inv(a):
    # extended Euclid over Fq[w]
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, (-18) % Q_MOD, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low):
        # r = high / low (polynomial rounded division)
        dl = _poly_deg(low)
        temp = list(high)
        r = [0] * 13
        inv_lead = pow(low[dl], -1, Q_MOD)
        for i in range(_poly_deg(temp) - dl, -1, -1):
            c = (temp[dl + i] * inv_lead) % Q_MOD
            r[i] = c
            for j in range(dl + 1):
                temp[i + j] = (temp[i + j] - c * low[j]) % Q_MOD
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] -= lm[i] * r[j]
                new[i + j] -= low[i] * r[j]
        nm = [v % Q_MOD for v in nm]
        new = [v % Q_MOD for v in new]
        lm, low, hm, high = nm, new, lm, low
    c = pow(low[0], -1, Q_MOD)
    return [(v * c) % Q_MOD for v in lm[:12]]


def _f12_sub(a, b):
    return [(x - y) % Q_MOD for x, y in zip(a, b)]


def _f12_add(a, b):
    return [(x + y) % Q_MOD for x, y in zip(a, b)]


def _f12_scalar(k):
    return [k % Q_MOD] + [0] * 11


_W = [0, 1] + [0] * 10
_W2 = _f12_mul(_W, _W)
_W3 = _f12_mul(_W2, _W)
ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE_LOOP_COUNT = 63


def _twist(q_affine):
    """Map an affine G2 point over Fq2 onto the curve y^2 = x^3 + 3 over Fq12."""
    (x0, x1), (y0, y1) = q_affine
    nx = [(x0 - 9 * x1) % Q_MOD] + [0] * 5 + [x1] + [0] * 5
    ny = [(y0 - 9 * y1) % Q_MOD] + [0] * 5 + [y1] + [0] * 5
    return (_f12_mul(nx, _W2), _f12_mul(ny, _W3))


def _a12_double(P):
    x, y = P
    lam = _f12_mul(_f12_mul(_f12_scalar(3), _f12_mul(x, x)), _f12_inv(_f12_add(y, y)))
    nx = _f12_sub(_f12_mul(lam, lam), _f12_add(x, x))
    ny = _f12_sub(_f12_mul(lam, _f12_sub(x, nx)), y)
    return (nx, ny)


def _a12_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if y1 == y2:
            return _a12_double(P)
        return None
    lam = _f12_mul(_f12_sub(y2, y1), _f12_inv(_f12_sub(x2, x1)))
    nx = _f12_sub(_f12_sub(_f12_mul(lam, lam), x1), x2)
    ny = _f12_sub(_f12_mul(lam, _f12_sub(x1, nx)), y1)
    return (nx, ny)


def _linefunc(P1, P2, T):
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if x1 != x2:
        m = _f12_mul(_f12_sub(y2, y1), _f12_inv(_f12_sub(x2, x1)))
        return _f12_sub(_f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))
    if y1 == y2:
        m = _f12_mul(_f12_mul(_f12_scalar(3), _f12_mul(x1, x1)), _f12_inv(_f12_add(y1, y1)))
        return _f12_sub(_f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))
    return _f12_sub(xt, x1)


def miller_loop(q_affine, p_affine):
    """Un-exponentiated optimal-ate Miller value for (P in G1, Q in G2), both affine or None."""
    if q_affine is None or p_affine is None:
        return list(F12_ONE)
    Q = _twist(q_affine)
    P = (_f12_scalar(p_affine[0]), _f12_scalar(p_affine[1]))
    Rp = Q
    f = list(F12_ONE)
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = _f12_mul(_f12_mul(f, f), _linefunc(Rp, Rp, P))
        Rp = _a12_double(Rp)
        if ATE_LOOP_COUNT & (1 << i):
            f = _f12_mul(f, _linefunc(Rp, Q, P))
            Rp = _a12_add(Rp, Q)
    Q1 = (_f12_pow(Q[0], Q_MOD), _f12_pow(Q[1], Q_MOD))
    nQ2 = (_f12_pow(Q1[0], Q_MOD), [(-v) % Q_MOD for v in _f12_pow(Q1[1], Q_MOD)])
    f = _f12_mul(f, _linefunc(Rp, Q1, P))
    Rp = _a12_add(Rp, Q1)
    f = _f12_mul(f, _linefunc(Rp, nQ2, P))
    return f


def final_exponentiation(f):
    return _f12_pow(f, (Q_MOD ** 12 - 1) // R_MOD)


def pairing(p_affine, q_affine):
    return final_exponentiation(miller_loop(q_affine, p_affine))


def pairing_product_is_one(pairs):
    """True iff prod e(P_i, Q_i) == 1, with one shared final exponentiation."""
    f = list(F12_ONE)
    for p_aff, q_aff in pairs:
        f = _f12_mul(f, miller_loop(q_aff, p_aff))
    return final_exponentiation(f) == F12_ONE
