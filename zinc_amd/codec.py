"""Byte formats of bellman's `Proof`, `VerifyingKey` and `Parameters` for BN256.

These are the files and the stdout hex that `zvm setup / prove / verify` exchange:
`proof.write` at zinc-vm/src/zvm/arguments/command/prove.rs:171-174, `Proof::read` at verify.rs:61,
`params.write` / `params.vk.write` at setup.rs:67-75, `Parameters::read(file, true)` at prove.rs:73-76.
Layouts (pairing_ce 0.17.0 / bellman_ce 0.3.0, which are not part of the reference tree):
  * coordinates are 32-byte big-endian canonical integers; Fq2 is written c1 then c0;
  * byte 0 bit 6 = point at infinity; compressed form also uses bit 7 = "y is the larger of
    (y, -y)" (Fq2 ordered by c1, then c0);
  * Proof = A (G1 compressed) || B (G2 compressed) || C (G1 compressed), 128 bytes;
  * VerifyingKey = alpha_g1, beta_g1 (G1 uncompressed), beta_g2, gamma_g2 (G2 uncompressed),
    delta_g1, delta_g2, u32 big-endian count, ic points (G1 uncompressed);
  * Parameters = VerifyingKey, then h, l, a, b_g1 (G1) and b_g2 (G2), each u32-BE count + points.
Points are Python tuples of canonical integers (None = infinity) on this side of the codec.
"""
import io

from .fields import Q_MOD

B_G1 = 3


def _fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def _fq2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = _fq2_mul(r, a)
        a = _fq2_mul(a, a)
        e >>= 1
    return r


def _fq2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % Q_MOD, -1, Q_MOD)
    return (a[0] * n % Q_MOD, -a[1] * n % Q_MOD)


B_G2 = _fq2_mul((3, 0), _fq2_inv((9, 1)))


def _fq_sqrt(a):
    y = pow(a, (Q_MOD + 1) // 4, Q_MOD)
    return y if y * y % Q_MOD == a % Q_MOD else None


def _fq2_sqrt(a):
    """Square root in Fq[u]/(u^2+1) for q = 3 mod 4 (complex method)."""
    if a == (0, 0):
        return (0, 0)
    a1 = _fq2_pow(a, (Q_MOD - 3) // 4)
    alpha = _fq2_mul(_fq2_mul(a1, a1), a)
    a0 = _fq2_mul((alpha[0], -alpha[1] % Q_MOD), alpha)  # alpha^q * alpha
    if a0 == (Q_MOD - 1, 0):
        return None
    x0 = _fq2_mul(a1, a)
    if alpha == (Q_MOD - 1, 0):
        return _fq2_mul((0, 1), x0)
    b = _fq2_pow(((1 + alpha[0]) % Q_MOD, alpha[1]), (Q_MOD - 1) // 2)
    return _fq2_mul(b, x0)


def _fq2_gt(a, b):
    return (a[1], a[0]) > (b[1], b[0])


class DecodeError(ValueError):
    pass


R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _g2_times_order_is_identity(p):
    """r * P == O for an affine point of the twist (Jacobian double-and-add over Fq2)."""
    sub = lambda a, b: ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)
    add = lambda a, b: ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)
    mul, sqr = _fq2_mul, lambda a: _fq2_mul(a, a)
    dbl2 = lambda a: add(a, a)
    px, py = p
    X, Y, Z = None, None, None  # the identity
    for bit in bin(R_ORDER)[2:]:
        if X is not None:
            if Y == (0, 0):
                X = None
            else:  # dbl-2009-l (a = 0)
                A, B = sqr(X), sqr(Y)
                Cc = sqr(B)
                D = dbl2(sub(sub(sqr(add(X, B)), A), Cc))
                E = add(dbl2(A), A)
                X3 = sub(sqr(E), dbl2(D))
                Y3 = sub(mul(E, sub(D, X3)), dbl2(dbl2(dbl2(Cc))))
                Z = dbl2(mul(Y, Z))
                X, Y = X3, Y3
        if bit == "1":
            if X is None:
                X, Y, Z = px, py, (1, 0)
            else:  # mixed addition
                Z2 = sqr(Z)
                U2, S2 = mul(px, Z2), mul(py, mul(Z2, Z))
                if U2 == X:
                    if S2 != Y:
                        X = None
                        continue
                    raise AssertionError("doubling inside an addition cannot happen for a point of odd prime order")
                H, Rr = sub(U2, X), sub(S2, Y)
                H2 = sqr(H)
                H3, V = mul(H, H2), mul(X, H2)
                X3 = sub(sub(sqr(Rr), H3), dbl2(V))
                Y = sub(mul(Rr, sub(V, X3)), mul(Y, H3))
                X, Z = X3, mul(Z, H)
    return X is None or Z == (0, 0)


# ------------------------------------------------------------------------------ single points
def g1_write(p, compressed):
    if p is None:
        return bytes([0x40]) + bytes(31 if compressed else 63)
    x, y = p
    if not compressed:
        return x.to_bytes(32, "big") + y.to_bytes(32, "big")
    out = bytearray(x.to_bytes(32, "big"))
    if y > (-y) % Q_MOD:
        out[0] |= 0x80
    return bytes(out)


def g2_write(p, compressed):
    if p is None:
        return bytes([0x40]) + bytes(63 if compressed else 127)
    (x0, x1), (y0, y1) = p
    if not compressed:
        return b"".join(v.to_bytes(32, "big") for v in (x1, x0, y1, y0))
    out = bytearray(x1.to_bytes(32, "big") + x0.to_bytes(32, "big"))
    if _fq2_gt((y0, y1), ((-y0) % Q_MOD, (-y1) % Q_MOD)):
        out[0] |= 0x80
    return bytes(out)


def _int(b):
    v = int.from_bytes(b, "big")
    if v >= Q_MOD:
        raise DecodeError("coordinate is not in the field")
    return v


def g1_read(buf, compressed, checked=True):
    buf = bytearray(buf)
    flags, buf[0] = buf[0] & 0xC0, buf[0] & 0x3F
    if flags & 0x40:
        if any(buf):
            raise DecodeError("infinity flag with non-zero coordinates")
        return None
    if compressed:
        x = _int(buf[:32])
        y = _fq_sqrt((x * x * x + B_G1) % Q_MOD)
        if y is None:
            raise DecodeError("x is not on the curve")
        if (y > (-y) % Q_MOD) != bool(flags & 0x80):
            y = (-y) % Q_MOD
        return (x, y)
    x, y = _int(buf[:32]), _int(buf[32:64])
    if checked and (y * y - x * x * x - B_G1) % Q_MOD:
        raise DecodeError("point is not on the curve")
    return (x, y)


def g2_read(buf, compressed, checked=True):
    buf = bytearray(buf)
    flags, buf[0] = buf[0] & 0xC0, buf[0] & 0x3F
    if flags & 0x40:
        if any(buf):
            raise DecodeError("infinity flag with non-zero coordinates")
        return None
    x = (_int(buf[32:64]), _int(buf[:32]))
    rhs = _fq2_mul(_fq2_mul(x, x), x)
    rhs = ((rhs[0] + B_G2[0]) % Q_MOD, (rhs[1] + B_G2[1]) % Q_MOD)
    if compressed:
        y = _fq2_sqrt(rhs)
        if y is None:
            raise DecodeError("x is not on the curve")
        ny = ((-y[0]) % Q_MOD, (-y[1]) % Q_MOD)
        if _fq2_gt(y, ny) != bool(flags & 0x80):
            y = ny
        # pairing_ce's G2Compressed::into_affine refuses a point outside the prime-order subgroup, so bellman's
        # Proof::read never hands one to the verifier
        if checked and not _g2_times_order_is_identity((x, y)):
            raise DecodeError("point is not in the prime-order subgroup")
        return (x, y)
    y = (_int(buf[96:128]), _int(buf[64:96]))
    if checked:
        if _fq2_mul(y, y) != rhs:
            raise DecodeError("point is not on the curve")
        if not _g2_times_order_is_identity((x, y)):
            raise DecodeError("point is not in the prime-order subgroup")
    return (x, y)


# ------------------------------------------------------------------------------ bellman structs
def proof_write(proof):
    """proof: (A, B, C) affine integer points -> 128 bytes (hex of this is what `zvm prove` prints)."""
    a, b, c = proof
    return g1_write(a, True) + g2_write(b, True) + g1_write(c, True)


def proof_read(data):
    if len(data) != 128:
        raise DecodeError("a BN256 Groth16 proof is 128 bytes")
    proof = (g1_read(data[:32], True), g2_read(data[32:96], True), g1_read(data[96:], True))
    if any(p is None for p in proof):  # bellman's Proof::read: "point at infinity" is InvalidData
        raise DecodeError("point at infinity in a proof")
    return proof


class KeyPoints:
    """Plain container with bellman's field names (affine integer points)."""

    def __init__(self):
        self.alpha_g1 = self.beta_g1 = self.delta_g1 = None
        self.beta_g2 = self.gamma_g2 = self.delta_g2 = None
        self.ic, self.h, self.l, self.a, self.b_g1, self.b_g2 = [], [], [], [], [], []


def vk_write(pk):
    out = [g1_write(pk.alpha_g1, False), g1_write(pk.beta_g1, False), g2_write(pk.beta_g2, False),
           g2_write(pk.gamma_g2, False), g1_write(pk.delta_g1, False), g2_write(pk.delta_g2, False),
           len(pk.ic).to_bytes(4, "big")]
    out += [g1_write(p, False) for p in pk.ic]
    return b"".join(out)


def _read_exact(f, n):
    b = f.read(n)
    if len(b) != n:
        raise DecodeError("unexpected end of key data")
    return b


def vk_read(f, pk=None):
    """bellman `VerifyingKey::read`: every point goes through `into_affine()` (on the curve, in the
    subgroup) whatever the caller's `checked` flag for the query vectors says, and ic may not hold
    the point at infinity."""
    pk = pk or KeyPoints()
    pk.alpha_g1 = g1_read(_read_exact(f, 64), False, True)
    pk.beta_g1 = g1_read(_read_exact(f, 64), False, True)
    pk.beta_g2 = g2_read(_read_exact(f, 128), False, True)
    pk.gamma_g2 = g2_read(_read_exact(f, 128), False, True)
    pk.delta_g1 = g1_read(_read_exact(f, 64), False, True)
    pk.delta_g2 = g2_read(_read_exact(f, 128), False, True)
    n = int.from_bytes(_read_exact(f, 4), "big")
    pk.ic = [g1_read(_read_exact(f, 64), False, True) for _ in range(n)]
    if any(p is None for p in pk.ic):
        raise DecodeError("point at infinity in ic")
    return pk


def params_write(pk):
    out = [vk_write(pk)]
    for vec in (pk.h, pk.l, pk.a, pk.b_g1):
        out.append(len(vec).to_bytes(4, "big"))
        out += [g1_write(p, False) for p in vec]
    out.append(len(pk.b_g2).to_bytes(4, "big"))
    out += [g2_write(p, False) for p in pk.b_g2]
    return b"".join(out)


def params_read(data, checked=True):
    """bellman `Parameters::read(reader, checked)`: with `checked`, points must be on the curve and
    the query vectors may not contain the point at infinity."""
    f = io.BytesIO(data) if isinstance(data, (bytes, bytearray)) else data
    pk = vk_read(f, KeyPoints())
    for name in ("h", "l", "a", "b_g1"):
        n = int.from_bytes(_read_exact(f, 4), "big")
        vec = [g1_read(_read_exact(f, 64), False, checked) for _ in range(n)]
        if checked and any(p is None for p in vec):
            raise DecodeError("point at infinity in %s" % name)
        setattr(pk, name, vec)
    n = int.from_bytes(_read_exact(f, 4), "big")
    pk.b_g2 = [g2_read(_read_exact(f, 128), False, checked) for _ in range(n)]
    if checked and any(p is None for p in pk.b_g2):
        raise DecodeError("point at infinity in b_g2")
    return pk
