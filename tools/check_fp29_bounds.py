"""Worst-case bound check of the lazy 29-bit-limb arithmetic in zinc_amd/csrc/{fp29,curve29}.hpp.

Every field value is tracked as (v, L): v = upper bound of the value in units of p, L = upper bound
of a limb below the top one.  The operations mirror the device code one to one and assert
  * no 64-bit column sum of a product overflows,
  * no limb-wise subtraction goes negative (the multiple K dominates the subtrahend limb by limb
    and in value), no lazy limb reaches 2^32,
  * every product result stays below 2p (what the zero test and the constants K rely on),
and the curve formulas are iterated until the bounds of the stored coordinates stop growing.
Run:  python tools/check_fp29_bounds.py   (exit status 0 = all bounds hold)
"""
import sys

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
C = Q / 2.0 ** 261            # a product of values a*p, b*p comes out below (1 + C*a*b) * p
NORM = (1 << 29) + 7          # limb bound after norm()
STRICT = (1 << 29) - 1        # limb bound of a product result
K_MIN = (1 << 31) - 4         # smallest low limb of a K constant (2^31 lent, 4 taken)
K_MAX = (1 << 31) + (1 << 29)
KS_MIN = (1 << 30) - 2
KS_MAX = (1 << 30) + (1 << 29)
MARGIN = 1e-3                 # value margin (in p) that keeps the top limb of K - b non-negative


class V:
    def __init__(self, v, L):
        self.v, self.L = v, L


def mul(a, b):
    assert 9 * a.L * b.L + 9 * STRICT * STRICT + (1 << 36) < 1 << 64, "column overflow"
    assert a.v * b.v <= 169.0, "product operands too large: %.1f * %.1f" % (a.v, b.v)
    return V(1 + C * a.v * b.v, STRICT)


def sqr(a):
    return mul(a, a)  # the doubled cross terms add up to the same column sums


def mul_pair(a, b, ks):
    """lane-pair Fq2 product: a*v + ao*z with z = b or KS - bo"""
    assert b.L <= KS_MIN, "KS does not dominate limbs"
    assert ks >= b.v + MARGIN, "KS does not dominate value"
    zL = KS_MAX
    assert 9 * a.L * b.L + 9 * a.L * zL + 9 * STRICT * STRICT + (1 << 36) < 1 << 64, "column overflow"
    s = a.v * (b.v + ks)
    assert s <= 169.0, "pair product operands too large: %.1f" % s
    return V(1 + C * s, STRICT)


def sqr_pair(a, k, limit=169.0):
    """complex squaring on a lane pair: (a0 + a1) * norm(a0 + K - a1) | (2 a0) * a1, one product per lane"""
    assert a.L <= K_MIN and k >= a.v + MARGIN, "K does not dominate"
    assert a.L + K_MAX < 1 << 32
    assert 9 * (2 * a.L) * NORM + 9 * STRICT * STRICT + (1 << 36) < 1 << 64, "column overflow"
    s = 2 * a.v * (a.v + k)
    assert s <= limit, "squaring operands too large: %.1f" % s
    return V(1 + C * s, STRICT)


def mul_sub_pair(a, b, kb, c, kc, d):
    """a*b - c*d over Fq2 in one scan; the negations norm(K - .) are formed by sub_k"""
    sub_k(ZERO, b, kb); sub_k(ZERO, c, kc)
    for x in (a, b, c, d):
        assert x.L <= NORM, "operands must be normalised"
    assert 4 * 9 * NORM * NORM + 9 * STRICT * STRICT + (1 << 36) < 1 << 64, "column overflow"
    s = a.v * (b.v + kb) + 2 * kc * d.v
    assert s <= 169.0, "fused pair product operands too large: %.1f" % s
    return V(1 + C * s, STRICT)


def mul_sub(a, b, c, d, ks):
    """a*b + (KS - c)*d in one column scan with one reduction"""
    assert c.L <= KS_MIN, "KS does not dominate limbs"
    assert ks >= c.v + MARGIN, "KS%d does not dominate value %.2f" % (ks, c.v)
    assert d.L <= NORM, "d must be normalised"
    assert 9 * a.L * b.L + 9 * KS_MAX * d.L + 9 * STRICT * STRICT + (1 << 36) < 1 << 64, "column overflow"
    s = a.v * b.v + ks * d.v
    assert s <= 169.0, "fused product operands too large: %.1f" % s
    return V(1 + C * s, STRICT)


def sub_k(a, b, k):
    assert b.L <= K_MIN and k >= b.v + MARGIN, "K%d does not dominate (%.2f)" % (k, b.v)
    assert a.L + K_MAX < 1 << 32
    return V(a.v + k, NORM)


def sub2_k(a, b, c, k):
    assert b.L + 2 * c.L <= K_MIN and k >= b.v + 2 * c.v + MARGIN, "K%d does not dominate" % k
    assert a.L + K_MAX < 1 << 32
    return V(a.v + k, NORM)


def mul_small_k(a, k):
    assert a.L * k < 1 << 32
    return V(a.v * k, NORM)


ZERO = V(0.0, 0)


class Pt:
    def __init__(self, x, y, zz, zzz):
        self.x, self.y, self.zz, self.zzz = x, y, zz, zzz


def g1_dbl(p):
    u = mul_small_k(p.y, 2); v = sqr(u); w = mul(u, v); s = mul(p.x, v)
    m = mul_small_k(sqr(p.x), 3)
    x3 = sub2_k(sqr(m), ZERO, s, 6); t = sub_k(s, x3, 8)
    y3 = sub_k(mul(m, t), mul(w, p.y), 3)
    return Pt(x3, y3, mul(v, p.zz), mul(w, p.zzz))


def g1_madd(acc, bx, by):
    u2 = mul(bx, acc.zz); s2 = mul(by, acc.zzz)
    pp_ = sub_k(u2, acc.x, 8); pp = sqr(pp_)
    r = sub_k(s2, acc.y, 6); sqr(r)
    ppp = mul(pp_, pp); q = mul(acc.x, pp)
    x3 = sub2_k(sqr(r), ppp, q, 6); t = sub_k(q, x3, 8)
    y3 = mul_sub(r, t, acc.y, ppp, 6)
    return Pt(x3, y3, mul(acc.zz, pp), mul(acc.zzz, ppp))


def g1_add(a, p):
    u1 = mul(a.x, p.zz); u2 = mul(p.x, a.zz); s1 = mul(a.y, p.zzz); s2 = mul(p.y, a.zzz)
    pp_ = sub_k(u2, u1, 3); pp = sqr(pp_); r = sub_k(s2, s1, 3); sqr(r)
    ppp = mul(pp_, pp); q = mul(u1, pp)
    x3 = sub2_k(sqr(r), ppp, q, 6); t = sub_k(q, x3, 8)
    y3 = mul_sub(r, t, s1, ppp, 2)
    return Pt(x3, y3, mul(mul(a.zz, p.zz), pp), mul(mul(a.zzz, p.zzz), ppp))


def g2_dbl(p):
    u = mul_small_k(p.y, 2); v = mul_pair(u, u, 8); w = mul_pair(u, v, 2); s = mul_pair(p.x, v, 2)
    m = mul_small_k(mul_pair(p.x, p.x, 6), 3)
    x3 = sub2_k(mul_pair(m, m, 6), ZERO, s, 4); t = sub_k(s, x3, 6)
    y3 = sub_k(mul_pair(t, m, 6), mul_pair(p.y, w, 2), 2)
    return Pt(x3, y3, mul_pair(p.zz, v, 2), mul_pair(p.zzz, w, 2))


def g2_dbl_affine(bx, by):
    u = mul_small_k(by, 2); v = mul_pair(u, u, 3); w = mul_pair(u, v, 2); s = mul_pair(bx, v, 2)
    m = mul_small_k(mul_pair(bx, bx, 2), 3)
    x3 = sub2_k(mul_pair(m, m, 4), ZERO, s, 4); t = sub_k(s, x3, 6)
    y3 = sub_k(mul_pair(t, m, 4), mul_pair(by, w, 2), 2)
    return Pt(x3, y3, v, w)


def g2_madd(acc, bx, by):
    u2 = mul_pair(acc.zz, bx, 2); s2 = mul_pair(acc.zzz, by, 2)
    pp_ = sub_k(u2, acc.x, 6); pp = sqr_pair(pp_, 8, limit=220.0)   # below 2.3p: its users take KS3
    assert pp.v < 3 - MARGIN
    r = sub_k(s2, acc.y, 4)
    ppp = mul_pair(pp_, pp, 3); q = mul_pair(acc.x, pp, 3)
    x3 = sub2_k(sqr_pair(r, 6), ppp, q, 4); t = sub_k(q, x3, 6)
    y3 = mul_sub_pair(t, r, 6, acc.y, 4, ppp)
    return Pt(x3, y3, mul_pair(acc.zz, pp, 3), mul_pair(acc.zzz, ppp, 2))


def g2_add(a, p):
    u1 = mul_pair(a.x, p.zz, 2); u2 = mul_pair(p.x, a.zz, 2)
    s1 = mul_pair(a.y, p.zzz, 2); s2 = mul_pair(p.y, a.zzz, 2)
    pp_ = sub_k(u2, u1, 2); pp = mul_pair(pp_, pp_, 4); r = sub_k(s2, s1, 2)
    ppp = mul_pair(pp_, pp, 2); q = mul_pair(u1, pp, 2)
    x3 = sub2_k(mul_pair(r, r, 4), ppp, q, 4); t = sub_k(q, x3, 6)
    y3 = sub_k(mul_pair(t, r, 4), mul_pair(s1, ppp, 2), 2)
    return Pt(x3, y3, mul_pair(mul_pair(a.zz, p.zz, 2), pp, 2), mul_pair(mul_pair(a.zzz, p.zzz, 2), ppp, 2))


def g1_quad_add(a, p):
    """quad_add (curve29.hpp): the steps of g1_add, four products at a time; lane 3's last product goes through the
    fused scan with a zero subtrahend"""
    m1 = [mul(a.x, p.zz), mul(a.y, p.zzz), mul(a.zz, p.x), mul(a.zzz, p.y)]
    d = [sub_k(m1[2], m1[0], 3), sub_k(m1[3], m1[1], 3)]
    m2 = [mul(d[0], d[0]), mul(d[1], d[1]), mul(a.zz, p.zz), mul(a.zzz, p.zzz)]
    m3 = [mul(m1[0], m2[0]), mul(d[0], m2[0]), mul(m2[2], m2[0])]
    x3 = sub2_k(m2[1], m3[1], m3[0], 6); t = sub_k(m3[0], x3, 8)
    y3 = mul_sub(d[1], t, m1[1], m3[1], 2)
    zzz3 = mul_sub(m2[3], m3[1], m1[3], ZERO, 2)
    return Pt(x3, y3, m3[2], zzz3)


def g2_oct_add(a, p):
    """oct_add: round 1 pairs every coordinate with the widest constant, round 4 is one plain product per position"""
    m1 = [mul_pair(a.x, p.zz, 8), mul_pair(a.y, p.zzz, 8), mul_pair(a.zzz, p.y, 8), mul_pair(a.zz, p.x, 8)]
    d = [sub_k(m1[3], m1[0], 2), sub_k(m1[2], m1[1], 2)]
    m2 = [mul_pair(d[0], d[0], 4), mul_pair(d[1], d[1], 4), mul_pair(a.zzz, p.zzz, 4), mul_pair(a.zz, p.zz, 4)]
    m3 = [mul_pair(m1[0], m2[0], 2), mul_pair(d[0], m2[0], 2), mul_pair(m2[3], m2[0], 2)]
    x3 = sub2_k(m2[1], m3[1], m3[0], 4); t = sub_k(m3[0], x3, 6)
    m4 = [mul_pair(m1[1], m3[1], 4), mul_pair(t, d[1], 4), mul_pair(m2[2], m3[1], 4)]
    y3 = sub_k(m4[1], m4[0], 2)
    return Pt(x3, y3, m3[2], m4[2])


def join(a, b):
    return Pt(*[V(max(p.v, q.v), max(p.L, q.L)) for p, q in
                zip((a.x, a.y, a.zz, a.zzz), (b.x, b.y, b.zz, b.zzz))])


def fixed_point(first, madd, add, dbl, base_y, stored_x, stored_y):
    base_x = V(1.0, STRICT)
    cur = first
    for _ in range(20):
        nxt = join(cur, madd(cur, base_x, base_y))
        for f in add:
            nxt = join(nxt, f(cur, cur))
        nxt = join(nxt, dbl(cur))
        same = all(abs(p.v - q.v) < 1e-9 for p, q in zip((cur.x, cur.y, cur.zz, cur.zzz),
                                                         (nxt.x, nxt.y, nxt.zz, nxt.zzz)))
        cur = nxt
        if same:
            break
    else:
        raise AssertionError("bounds keep growing")
    assert cur.x.v <= stored_x and cur.y.v <= stored_y and cur.zz.v < 2 and cur.zzz.v < 2, \
        "stored bounds exceeded: X %.2f Y %.2f" % (cur.x.v, cur.y.v)
    return cur


def main():
    one = V(1.0, STRICT)
    # G1: the base's y may be 3p - y (negated digit), normalised
    g1_first = Pt(V(1.0, STRICT), V(3.0, NORM), one, one)
    r = fixed_point(g1_first, g1_madd, (g1_add, g1_quad_add), g1_dbl, V(3.0, NORM), 8.0, 5.0)
    print("G1 stored bounds: X < %.2fp  Y < %.2fp  ZZ < %.2fp  ZZZ < %.2fp" % (r.x.v, r.y.v, r.zz.v, r.zzz.v))
    # G2 on lane pairs: canonical bases (negation is done on the packed form)
    g2_first = Pt(one, one, one, one)
    g2_dbl_affine(one, one)
    r = fixed_point(g2_first, g2_madd, (g2_add, g2_oct_add), g2_dbl, one, 6.0, 4.0)
    print("G2 stored bounds: X < %.2fp  Y < %.2fp  ZZ < %.2fp  ZZZ < %.2fp (per component)" % (r.x.v, r.y.v, r.zz.v, r.zzz.v))
    # radix change for the host: to_r256 multiplies a stored coordinate by a canonical constant
    mul(V(8.0, NORM), one)
    print("all bounds hold")
    return 0


if __name__ == "__main__":
    sys.exit(main())
