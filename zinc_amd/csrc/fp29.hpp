// BN254 prime fields on the gfx950 VALU in an unsaturated representation: nine 29-bit limbs in
// 32-bit registers, Montgomery radix R = 2^261.
//
// Why.  The widest integer multiply of CDNA4 is v_mad_u64_u32 (32 x 32 + 64 -> 64) and it has no
// carry input, so with saturated 32-bit limbs every product term needs a second instruction to
// propagate its carry (field.hpp's device product: 128 multiply-adds + 128 carry adds + moves).
// With 29-bit limbs a whole column of the product scan -- nine a_i*b_j terms and nine m_i*p_j
// terms, each below 2^58 -- accumulates in one 64-bit register with no carry at all, so a term is
// exactly one instruction: 162 multiply-adds and ~60 shifts / masks per Montgomery product.  The
// five spare bits of the radix (2^261 against a 254-bit modulus) also remove the final conditional
// subtraction and make additions and subtractions limb-wise without carry chains.
//
// Value discipline (checked exhaustively by tools/check_fp29_bounds.py):
//   * a product's operands may be any residues below 13p whose limb products fit the column bound;
//     its result is below 2p with limbs strictly below 2^29 (top limb below 2^25).
//   * sub / the fused forms add a multiple of p large enough to stay non-negative limb by limb, then
//     `norm` brings every limb back below 2^29 + 8 by one parallel carry step.
//   * nothing here is canonical: equality with zero is tested modulo p where the curve code needs it.
//
// Only device code uses this file.  Memory formats: bases are read from the packed 8 x 32-bit
// canonical form (in radix-2^261 Montgomery form, see to_r261); accumulators are stored as nine
// limbs per coordinate.
#pragma once
#include "field.hpp"

namespace zg {

#include "fp29_consts.inc"

template <class P>
struct Fp29 {
  uint32_t v[9];

  ZG_D static Fp29 zero() {
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
  }
  ZG_D static Fp29 one() {
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = P::ONE[i];
    return r;
  }
  // all limbs zero (the encoding of a value that was stored as zero, e.g. ZZ of the identity)
  ZG_D bool is_zero_limbs() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= v[i];
    return o == 0;
  }
};

static constexpr uint32_t MASK29 = 0x1fffffffu;

// One parallel carry step: limbs below 2^32 in, limbs below 2^29 + 8 out (top limb keeps its excess).
template <class P>
ZG_D Fp29<P> norm(const Fp29<P>& a) {
  Fp29<P> r;
  r.v[0] = a.v[0] & MASK29;
#pragma unroll
  for (int i = 1; i < 8; i++) r.v[i] = (a.v[i] & MASK29) + (a.v[i - 1] >> 29);
  r.v[8] = a.v[8] + (a.v[7] >> 29);
  return r;
}

// Limb-wise sum, no carries: the caller keeps limbs below 2^32 and normalises before a product.
template <class P>
ZG_D Fp29<P> add_lazy(const Fp29<P>& a, const Fp29<P>& b) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}

// norm(a + K - b) for a multiple K of p (one of P::K*) that dominates b limb by limb.
template <class P>
ZG_D Fp29<P> sub_k(const Fp29<P>& a, const Fp29<P>& b, const uint32_t (&K)[9]) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + K[i] - b.v[i];
  return norm(r);
}

// norm(a + K - b - 2c)
template <class P>
ZG_D Fp29<P> sub2_k(const Fp29<P>& a, const Fp29<P>& b, const Fp29<P>& c, const uint32_t (&K)[9]) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + K[i] - b.v[i] - 2 * c.v[i];
  return norm(r);
}

// norm(k * a) for a small constant k (limbs of a below 2^29 + 8, k <= 4)
template <class P>
ZG_D Fp29<P> mul_small_k(const Fp29<P>& a, uint32_t k) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] * k;
  return norm(r);
}

ZG_D uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }

// Montgomery product a*b*2^-261 mod p, result below 2p with normalised limbs.  Column sums: at most
// nine a_i*b_j and nine m_i*p_j terms plus the carried-in word; the bound script proves < 2^64 for
// the operand classes the curve code uses (one operand with limbs up to 2^31, the other normalised).
template <class P>
ZG_D Fp29<P> mul(const Fp29<P>& a, const Fp29<P>& b) {
  uint32_t m[9];
  Fp29<P> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (j >= 0 && j < 9) acc = mad64(a.v[i], b.v[j], acc);
    }
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      m[k] = ((uint32_t)acc * P::INV) & MASK29;
      acc = mad64(m[k], P::MOD[0], acc);
    } else {
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      r.v[k - 9] = (uint32_t)acc & MASK29;
    }
    acc >>= 29;
  }
  r.v[8] = (uint32_t)acc;
  return r;
}

// a^2: the 36 cross terms are taken once against a doubled operand.
template <class P>
ZG_D Fp29<P> sqr(const Fp29<P>& a) {
  uint32_t m[9], a2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = a.v[i] << 1;
  Fp29<P> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (j >= 0 && j < 9 && i <= j) acc = i == j ? mad64(a.v[i], a.v[i], acc) : mad64(a.v[i], a2[j], acc);
    }
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      m[k] = ((uint32_t)acc * P::INV) & MASK29;
      acc = mad64(m[k], P::MOD[0], acc);
    } else {
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      r.v[k - 9] = (uint32_t)acc & MASK29;
    }
    acc >>= 29;
  }
  r.v[8] = (uint32_t)acc;
  return r;
}

// (a*b - c*d) * 2^-261 mod p as ONE column scan with one Montgomery reduction: a*b + (KS - c)*d, where KS (one of
// P::KS*) is a multiple of p that dominates c limb by limb and in value.  Saves the second product's 81
// reduction multiply-adds.  Result below 2p with normalised limbs when
// (bound of a)*(bound of b) + (KS / p)*(bound of d) <= 169; c, d must have normalised limbs.
template <class P>
ZG_D Fp29<P> mul_sub(const Fp29<P>& a, const Fp29<P>& b, const Fp29<P>& c, const Fp29<P>& d, const uint32_t (&KS)[9]) {
  uint32_t z[9];
#pragma unroll
  for (int i = 0; i < 9; i++) z[i] = KS[i] - c.v[i];
  uint32_t m[9];
  Fp29<P> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (j >= 0 && j < 9) {
        acc = mad64(a.v[i], b.v[j], acc);
        acc = mad64(z[i], d.v[j], acc);
      }
    }
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      m[k] = ((uint32_t)acc * P::INV) & MASK29;
      acc = mad64(m[k], P::MOD[0], acc);
    } else {
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      r.v[k - 9] = (uint32_t)acc & MASK29;
    }
    acc >>= 29;
  }
  r.v[8] = (uint32_t)acc;
  return r;
}

// ---- Fq2 = Fq[u]/(u^2 + 1) spread over a lane pair ---------------------------------------------------
// Lane 2k holds the real component of every Fq2 value, lane 2k+1 the imaginary one.  Additions
// and subtractions are the single-lane ones; a product needs the partner's components, fetched
// with one DPP move per limb, and then both lanes run the same code:
//   real: a0*b0 + a1*(K - b1)      imaginary: a1*b0 + a0*b1
// as ONE column scan over both products with one Montgomery reduction (243 multiply-adds).
ZG_D uint32_t pair_swap(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
}
template <class P>
ZG_D Fp29<P> pair_swap(const Fp29<P>& a) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = pair_swap(a.v[i]);
  return r;
}
ZG_D bool pair_both(bool x) { return x && pair_swap((uint32_t)x) != 0; }

// This lane's component of a*b.  b's components must have normalised limbs and lie below the multiple
// KS of p (one of P::KS*) with a margin; the result is below 2p when
// (bound of a) * (bound of b + KS / p) <= 169, and has normalised limbs.
// (ao, bo: the pair partner's components of a and b, for callers that hold them already.)
template <class P>
ZG_D Fp29<P> mul_pair_x(const Fp29<P>& a, const Fp29<P>& ao, const Fp29<P>& b, const Fp29<P>& bo, const uint32_t (&KS)[9]) {
  const bool im = (threadIdx.x & 1) != 0;
  uint32_t v[9], z[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    v[i] = im ? bo.v[i] : b.v[i];
    z[i] = im ? b.v[i] : KS[i] - bo.v[i];
  }
  uint32_t m[9];
  Fp29<P> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (j >= 0 && j < 9) {
        acc = mad64(a.v[i], v[j], acc);
        acc = mad64(ao.v[i], z[j], acc);
      }
    }
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      m[k] = ((uint32_t)acc * P::INV) & MASK29;
      acc = mad64(m[k], P::MOD[0], acc);
    } else {
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      r.v[k - 9] = (uint32_t)acc & MASK29;
    }
    acc >>= 29;
  }
  r.v[8] = (uint32_t)acc;
  return r;
}
template <class P>
ZG_D Fp29<P> mul_pair(const Fp29<P>& a, const Fp29<P>& b, const uint32_t (&KS)[9]) {
  return mul_pair_x(a, pair_swap(a), b, pair_swap(b), KS);
}

// This lane's component of a^2 by the complex squaring (a0 + a1)(a0 - a1) + 2 a0 a1 u: ONE product per lane (162
// multiply-adds against mul_pair's 243).  K (one of P::K*) dominates a's components limb by limb; the result is
// below (1 + C s) p with s = 2 (bound of a) * (bound of a + K / p), which may exceed 169 (the callers say how far).
template <class P>
ZG_D Fp29<P> sqr_pair(const Fp29<P>& a, const uint32_t (&K)[9]) {
  const bool im = (threadIdx.x & 1) != 0;
  const Fp29<P> ao = pair_swap(a);
  Fp29<P> x, y;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    x.v[i] = im ? 2 * ao.v[i] : a.v[i] + ao.v[i];          // a0 + a1 | 2 a0
    y.v[i] = im ? a.v[i] : a.v[i] + K[i] - ao.v[i];        // a0 - a1 | a1
  }
  return mul(x, norm(y));
}

// This lane's component of a*b - c*d over Fq2 as ONE column scan with one Montgomery reduction (405 multiply-adds
// against 486 for two products).  nb and nc are component-wise negations of b and c, norm(K - .) for any multiple K
// of p that dominates them; every operand has normalised limbs.  Result below 2p when
// (bound of a) * (bound of b + K_b / p) + 2 (K_c / p) (bound of d) <= 169.
template <class P>
ZG_D Fp29<P> mul_sub_pair(const Fp29<P>& a, const Fp29<P>& b, const Fp29<P>& nb, const Fp29<P>& c, const Fp29<P>& nc,
                          const Fp29<P>& d) {
  const bool im = (threadIdx.x & 1) != 0;
  const Fp29<P> ao = pair_swap(a), bo = pair_swap(b), nbo = pair_swap(nb), co = pair_swap(c), nco = pair_swap(nc),
                dd = pair_swap(d);
  // real: a0 b0 + a1 (-b1) + (-c0) d0 + c1 d1        imaginary: a1 b0 + a0 b1 + (-c1) d0 + (-c0) d1
  uint32_t v1[9], v2[9], v3[9], u4[9], v4[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    v1[i] = im ? bo.v[i] : b.v[i];     // times a
    v2[i] = im ? b.v[i] : nbo.v[i];    // times ao
    v3[i] = im ? dd.v[i] : d.v[i];     // times nc
    u4[i] = im ? nco.v[i] : co.v[i];
    v4[i] = im ? d.v[i] : dd.v[i];     // times u4
  }
  uint32_t m[9];
  Fp29<P> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (j >= 0 && j < 9) {
        acc = mad64(a.v[i], v1[j], acc);
        acc = mad64(ao.v[i], v2[j], acc);
        acc = mad64(nc.v[i], v3[j], acc);
        acc = mad64(u4[i], v4[j], acc);
      }
    }
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      m[k] = ((uint32_t)acc * P::INV) & MASK29;
      acc = mad64(m[k], P::MOD[0], acc);
    } else {
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::MOD[k - i], acc);
      r.v[k - 9] = (uint32_t)acc & MASK29;
    }
    acc >>= 29;
  }
  r.v[8] = (uint32_t)acc;
  return r;
}

// Is the product-normalised value a (below 3p, limbs strictly below 2^29) zero modulo p?
template <class P>
ZG_D bool is_zero_mod_p3(const Fp29<P>& a) {
  uint32_t z = 0, e = 0, f = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    z |= a.v[i];
    e |= a.v[i] ^ P::MOD[i];
    f |= a.v[i] ^ P::MOD2[i];
  }
  return z == 0 || e == 0 || f == 0;
}

// Is the product-normalised value a (below 2p, limbs strictly below 2^29) zero modulo p?
template <class P>
ZG_D bool is_zero_mod_p(const Fp29<P>& a) {
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    z |= a.v[i];
    e |= a.v[i] ^ P::MOD[i];
  }
  return z == 0 || e == 0;
}

// ---- packed 8 x 32-bit canonical form <-> nine 29-bit limbs --------------------------------------
template <class P, class Q>
ZG_D Fp29<P> unpack29(const Fp<Q>& x) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, w = bit >> 5, s = bit & 31;
    uint32_t lo = x.v[w] >> s;
    if (s > 3 && w + 1 < 8) lo |= x.v[w + 1] << (32 - s);
    r.v[i] = i < 8 ? (lo & MASK29) : lo;
  }
  return r;
}

// a (below 2p, limbs strictly normalised) -> the canonical residue packed in eight 32-bit words
template <class P, class Q>
ZG_D Fp<Q> pack29(const Fp29<P>& a) {
  // conditional subtraction of p with a borrow chain over 29-bit limbs
  uint32_t d[9];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t t = a.v[i] - P::MOD[i] - borrow;
    borrow = t >> 31;
    d[i] = i < 8 ? (t & MASK29) : t;
  }
  uint32_t l[9];
#pragma unroll
  for (int i = 0; i < 9; i++) l[i] = borrow ? a.v[i] : d[i];
  Fp<Q> r;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int bit = 32 * w, i = bit / 29, s = bit % 29;
    uint32_t x = l[i] >> s;
    x |= l[i + 1] << (29 - s);
    if (29 - s + 29 < 32 && i + 2 < 9) x |= l[i + 2] << (58 - s);
    r.v[w] = x;
  }
  return r;
}

// Montgomery radix change.  Packed values cross the library boundary in radix-2^256 form
// (ff_ce's layout); the 29-bit code works in radix 2^261.
template <class P, class Q>
ZG_D Fp29<P> from_r256(const Fp<Q>& x) {  // one product
  Fp29<P> c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.v[i] = P::R266[i];
  return mul(unpack29<P>(x), c);
}
template <class P, class Q>
ZG_D Fp<Q> to_r256(const Fp29<P>& a) {  // a below 13p with normalised limbs
  Fp29<P> c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.v[i] = P::R256[i];
  return pack29<P, Q>(mul(a, c));
}

using Fq29 = Fp29<Fq29Params>;
using Fr29 = Fp29<Fr29Params>;

}  // namespace zg
