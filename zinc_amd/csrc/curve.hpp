// BN254 G1 / G2 group arithmetic for host and gfx950 device code.
//
// Bases cross the ABI as affine points (x || y, Montgomery limbs, all-zero bytes = the point at
// infinity; (0,0) is on neither curve).  Accumulators use extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): the mixed addition that dominates bucket accumulation
// costs 8M + 2S there against 7M + 4S in plain Jacobian, and ZZ == 0 encodes the identity so a
// zero-filled bucket array is an array of identities.
//
// Group results are exact, so any addition order / coordinate system yields the same affine
// point bellman_ce's Jacobian code produces (the reference calls it through
// zinc-vm/src/core/circuit/facade.rs:153).
#pragma once
#include "field.hpp"

// Group operations are inlined into their callers; the field product underneath may be out of line
// (see ZG_OUTLINE_FIELD_MUL in field.hpp).
#define ZG_GROUP_OP ZG_HD

namespace zg {

template <class F>
struct Affine {
  F x, y;
  ZG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  ZG_HD static Affine inf() { return {F::zero(), F::zero()}; }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  ZG_HD bool is_inf() const { return zz.is_zero(); }
  ZG_HD static XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  ZG_HD static XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return {p.x, p.y, F::one(), F::one()};
  }
};

template <class F>
ZG_HD Affine<F> neg(const Affine<F>& p) {
  return {p.x, neg(p.y)};
}

template <class F>
ZG_HD XYZZ<F> neg(const XYZZ<F>& p) {
  return {p.x, neg(p.y), p.zz, p.zzz};
}

// 2*(x, y) for an affine, non-identity input (mdbl-2008-s-1, a = 0)
template <class F>
ZG_GROUP_OP XYZZ<F> dbl_affine(const Affine<F>& p) {
  F u = dbl(p.y);
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(p.x, v);
  F xx = sqr(p.x);
  F m = add(dbl(xx), xx);
  F x3 = sub(sqr(m), dbl(s));
  F y3 = sub(mul(m, sub(s, x3)), mul(w, p.y));
  return {x3, y3, v, w};
}

// dbl-2008-s-1, a = 0
template <class F>
ZG_GROUP_OP XYZZ<F> dbl(const XYZZ<F>& p) {
  if (p.is_inf()) return p;
  F u = dbl(p.y);
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(p.x, v);
  F xx = sqr(p.x);
  F m = add(dbl(xx), xx);
  F x3 = sub(sqr(m), dbl(s));
  F y3 = sub(mul(m, sub(s, x3)), mul(w, p.y));
  return {x3, y3, mul(v, p.zz), mul(w, p.zzz)};
}

// acc += p (affine).  madd-2008-s with the exceptional cases handled exactly.
template <class F>
ZG_GROUP_OP void add_mixed(XYZZ<F>& acc, const Affine<F>& p) {
  if (p.is_inf()) return;
  if (acc.is_inf()) {
    acc = {p.x, p.y, F::one(), F::one()};
    return;
  }
  F u2 = mul(p.x, acc.zz);
  F s2 = mul(p.y, acc.zzz);
  F pp_ = sub(u2, acc.x);
  F r = sub(s2, acc.y);
  if (pp_.is_zero()) {
    if (r.is_zero())
      acc = dbl_affine(p);
    else
      acc = XYZZ<F>::inf();
    return;
  }
  F pp = sqr(pp_);
  F ppp = mul(pp_, pp);
  F q = mul(acc.x, pp);
  F x3 = sub(sub(sqr(r), ppp), dbl(q));
  F y3 = sub(mul(r, sub(q, x3)), mul(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul(acc.zz, pp);
  acc.zzz = mul(acc.zzz, ppp);
}

// acc += p (XYZZ).  add-2008-s with the exceptional cases handled exactly.
template <class F>
ZG_GROUP_OP void add_full(XYZZ<F>& acc, const XYZZ<F>& p) {
  if (p.is_inf()) return;
  if (acc.is_inf()) {
    acc = p;
    return;
  }
  F u1 = mul(acc.x, p.zz);
  F u2 = mul(p.x, acc.zz);
  F s1 = mul(acc.y, p.zzz);
  F s2 = mul(p.y, acc.zzz);
  F pp_ = sub(u2, u1);
  F r = sub(s2, s1);
  if (pp_.is_zero()) {
    if (r.is_zero())
      acc = dbl(acc);
    else
      acc = XYZZ<F>::inf();
    return;
  }
  F pp = sqr(pp_);
  F ppp = mul(pp_, pp);
  F q = mul(u1, pp);
  F x3 = sub(sub(sqr(r), ppp), dbl(q));
  F y3 = sub(mul(r, sub(q, x3)), mul(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul(mul(acc.zz, p.zz), pp);
  acc.zzz = mul(mul(acc.zzz, p.zzz), ppp);
}

template <class F>
ZG_HD Affine<F> to_affine(const XYZZ<F>& p) {
  if (p.is_inf()) return Affine<F>::inf();
  // 1/ZZ and 1/ZZZ from one inversion
  F t = inv(mul(p.zz, p.zzz));
  F izz = mul(t, p.zzz);
  F izzz = mul(t, p.zz);
  return {mul(p.x, izz), mul(p.y, izzz)};
}

// k * p for a canonical (non-Montgomery) little-endian scalar of 8 u32 limbs.
template <class F>
ZG_HD XYZZ<F> scalar_mul(const XYZZ<F>& p, const uint32_t* k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = 7; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      acc = dbl(acc);
      if ((k[i] >> b) & 1) add_full(acc, p);
    }
  return acc;
}

// The same with 4-bit windows: 14 additions for the table, then 4 doublings and at most one addition per
// nibble (about 250 doublings + 60 additions instead of 254 + 127).
template <class F>
ZG_HD XYZZ<F> scalar_mul_w4(const XYZZ<F>& p, const uint32_t* k) {
  XYZZ<F> tab[16];
  tab[0] = XYZZ<F>::inf();
  tab[1] = p;
  for (int i = 2; i < 16; i++) {
    tab[i] = tab[i - 1];
    add_full(tab[i], p);
  }
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = 7; i >= 0; i--)
    for (int n = 7; n >= 0; n--) {
      if (!acc.is_inf())
        for (int d = 0; d < 4; d++) acc = dbl(acc);
      const uint32_t digit = (k[i] >> (4 * n)) & 15;
      if (digit) add_full(acc, tab[digit]);
    }
  return acc;
}

// k * P for a point known in advance: t[w][j - 1] = j * 16^w * P, so the product is one addition per non-zero
// nibble of k and no doubling (64 additions instead of ~250 doublings + 60 additions).
template <class F>
struct FixedBaseTable {
  XYZZ<F> t[64][15];
  void build(const Affine<F>& p) {
    XYZZ<F> base = XYZZ<F>::from_affine(p);
    for (int w = 0; w < 64; w++) {
      t[w][0] = base;
      for (int j = 1; j < 15; j++) {
        t[w][j] = t[w][j - 1];
        add_full(t[w][j], base);
      }
      base = t[w][14];
      add_full(base, t[w][0]);  // 16 * 16^w * P
    }
  }
  XYZZ<F> mul(const uint32_t* k) const {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = 0; w < 64; w++) {
      const uint32_t digit = (k[w >> 3] >> (4 * (w & 7))) & 15;
      if (digit) add_full(acc, t[w][digit - 1]);
    }
    return acc;
  }
};

// k1 * p + k2 * q in one pass over both scalars, two bits of each at a time: 15 additions for the table of
// i*p + j*q (i, j < 4), then 2 doublings and at most one addition per step (about 254 doublings + 120 additions
// instead of twice 250 + 60).
template <class F>
ZG_HD XYZZ<F> double_scalar_mul(const XYZZ<F>& p, const uint32_t* k1, const XYZZ<F>& q, const uint32_t* k2) {
  XYZZ<F> tab[4][4];
  tab[0][0] = XYZZ<F>::inf();
  for (int i = 0; i < 4; i++) {
    if (i) {
      tab[i][0] = tab[i - 1][0];
      add_full(tab[i][0], p);
    }
    for (int j = 1; j < 4; j++) {
      tab[i][j] = tab[i][j - 1];
      add_full(tab[i][j], q);
    }
  }
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int bit = 254; bit >= 0; bit -= 2) {
    if (!acc.is_inf()) acc = dbl(dbl(acc));
    const uint32_t d1 = (k1[bit >> 5] >> (bit & 31)) & 3, d2 = (k2[bit >> 5] >> (bit & 31)) & 3;
    if (d1 | d2) add_full(acc, tab[d1][d2]);
  }
  return acc;
}

template <class F>
ZG_HD bool on_curve(const Affine<F>& p, const F& b) {
  if (p.is_inf()) return true;
  return sqr(p.y) == add(mul(sqr(p.x), p.x), b);
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

// ---- membership of the prime-order subgroup of the twist (what pairing_ce's `into_affine()` checks) ----
// psi = untwist, q-power Frobenius, twist:  psi(x, y) = (conj(x) * xi^((q-1)/3), conj(y) * xi^((q-1)/2)), xi = 9 + u.
// On the whole twist curve psi^2 - t psi + q = 0 with t = q + 1 - r = 6 x^2 + 1 (x the BN parameter), so a point with
// psi(P) = [6 x^2] P satisfies ((6x^2)^2 - (6x^2 + 1) 6x^2 + q) P = (q - 6x^2) P = r P = O; and on the order-r
// subgroup psi IS multiplication by q = 6 x^2 (mod r).  The test "on the curve and psi(P) = [6x^2] P" therefore
// accepts exactly the points that "on the curve and r P = O" accepts, with a 127-bit scalar instead of a 254-bit one.
struct G2SubgroupConsts {
  Fq2 curve_b;  // 3 / xi
  Fq2 psi_x;    // xi^((q - 1) / 3)
  Fq2 psi_y;    // xi^((q - 1) / 2)
};

ZG_HD bool g2_in_subgroup(const Affine<Fq2>& p, const G2SubgroupConsts& k) {
  if (p.is_inf()) return true;
  // 6 x^2 = 0x6f4d8248eeb859fbf83e9682e87cfd46, x = 4965661367192848881
  const uint32_t e[4] = {0xe87cfd46u, 0xf83e9682u, 0xeeb859fbu, 0x6f4d8248u};
  XYZZ<Fq2> acc = XYZZ<Fq2>::inf();
  for (int i = 3; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      acc = dbl(acc);
      if ((e[i] >> b) & 1) add_mixed(acc, p);
    }
  if (acc.is_inf()) return false;  // psi(P) is a finite point
  const Fq2 px = mul(Fq2{p.x.c0, neg(p.x.c1)}, k.psi_x), py = mul(Fq2{p.y.c0, neg(p.y.c1)}, k.psi_y);
  return mul(px, acc.zz) == acc.x && mul(py, acc.zzz) == acc.y;
}

// Host: the three constants, computed once per process.
inline const G2SubgroupConsts& g2_subgroup_consts() {
  static const G2SubgroupConsts consts = [] {
    const Fq2 xi = {from_u64<FqParams>(9), Fq::one()};
    const Fq2 three = {from_u64<FqParams>(3), Fq::zero()};
    auto xi_pow_q_minus_1_over = [&](uint64_t d) {
      uint32_t e[8];
      for (int i = 0; i < 8; i++) e[i] = FqParams::mod(i);
      e[0] -= 1;  // q is odd
      uint64_t rem = 0;
      for (int i = 7; i >= 0; i--) {  // exact: q = 1 (mod 6)
        const uint64_t cur = (rem << 32) | e[i];
        e[i] = (uint32_t)(cur / d);
        rem = cur % d;
      }
      Fq2 g = Fq2::one();
      for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
          g = sqr(g);
          if ((e[i] >> b) & 1) g = mul(g, xi);
        }
      return g;
    };
    G2SubgroupConsts c;
    c.curve_b = mul(three, inv(xi));
    c.psi_x = xi_pow_q_minus_1_over(3);
    c.psi_y = xi_pow_q_minus_1_over(2);
    return c;
  }();
  return consts;
}

}  // namespace zg
