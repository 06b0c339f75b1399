// BN254 G1 group operations over the 29-bit-limb field of fp29.hpp (device only).
//
// Same formulas as curve.hpp (extended Jacobian XYZZ: madd-2008-s, add-2008-s, dbl-2008-s-1), with
// the value bounds of the lazy field written next to each step in units of p:
//   stored points:  X < 8p,  Y < 5p,  ZZ, ZZZ < 2p,  limbs normalised;  ZZ == 0 (all limbs) <=> identity
//   affine bases:   x, y < p canonical
// Every product keeps (bound of a) * (bound of b) <= 169, which is what makes its result < 2p.
// Exact group results do not depend on any of this: the affine point is the one bellman_ce computes
// (reached from zinc-vm/src/core/circuit/facade.rs:153).
#pragma once
#include "curve.hpp"
#include "fp29.hpp"

namespace zg {

template <class P>
struct Affine29 {
  Fp29<P> x, y;
};

template <class P>
struct XYZZ29 {
  Fp29<P> x, y, zz, zzz;
  ZG_D bool is_inf() const { return zz.is_zero_limbs(); }
  ZG_D static XYZZ29 inf() { return {Fp29<P>::zero(), Fp29<P>::zero(), Fp29<P>::zero(), Fp29<P>::zero()}; }
};

// 2*(x, y) for an affine, non-identity input
template <class P>
ZG_D XYZZ29<P> dbl_affine29(const Affine29<P>& p) {
  using F = Fp29<P>;
  const F u = mul_small_k(p.y, 2);                    // 2
  const F v = sqr(u);                                 // <2
  const F w = mul(u, v);                              // <2
  const F s = mul(p.x, v);                            // <2
  const F m = mul_small_k(sqr(p.x), 3);               // <6
  const F x3 = sub2_k(sqr(m), F::zero(), s, P::K6);   // m^2 - 2s : < 2 + 6 = 8
  const F t = sub_k(s, x3, P::K8);                    // < 2 + 8 = 10
  const F y3 = sub_k(mul(m, t), mul(w, p.y), P::K3);  // < 2 + 3 = 5
  return {x3, y3, v, w};
}

template <class P>
ZG_D XYZZ29<P> dbl29(const XYZZ29<P>& p) {
  using F = Fp29<P>;
  if (p.is_inf()) return p;
  const F u = mul_small_k(p.y, 2);                    // <10
  const F v = sqr(u);                                 // 100 -> <2
  const F w = mul(u, v);                              // <2
  const F s = mul(p.x, v);                            // 16 -> <2
  const F m = mul_small_k(sqr(p.x), 3);               // 64 -> <2, times 3: <6
  const F x3 = sub2_k(sqr(m), F::zero(), s, P::K6);   // <8
  const F t = sub_k(s, x3, P::K8);                    // <10
  const F y3 = sub_k(mul(m, t), mul(w, p.y), P::K3);  // 60, 10 -> <5
  return {x3, y3, mul(v, p.zz), mul(w, p.zzz)};
}

// acc += p (affine, not the identity)
template <class P>
ZG_D void add_mixed29(XYZZ29<P>& acc, const Affine29<P>& p) {
  using F = Fp29<P>;
  if (acc.is_inf()) {
    acc = {p.x, p.y, F::one(), F::one()};
    return;
  }
  const F u2 = mul(p.x, acc.zz);                  // <2
  const F s2 = mul(p.y, acc.zzz);                 // <2
  const F pp_ = sub_k(u2, acc.x, P::K8);          // <10
  const F pp = sqr(pp_);                          // <2
  if (is_zero_mod_p(pp)) {                        // same x: doubling or cancellation (rare)
    const F r0 = sub_k(s2, acc.y, P::K6);
    const F rr = sqr(r0);
    if (is_zero_mod_p(rr))
      acc = dbl_affine29(p);
    else
      acc = XYZZ29<P>::inf();
    return;
  }
  const F r = sub_k(s2, acc.y, P::K6);            // <8
  const F ppp = mul(pp_, pp);                     // <2
  const F q = mul(acc.x, pp);                     // <2
  const F x3 = sub2_k(sqr(r), ppp, q, P::K6);     // r^2 - ppp - 2q : < 2 + 6 = 8
  const F t = sub_k(q, x3, P::K8);                // <10
  const F y3 = mul_sub(r, t, acc.y, ppp, P::KS6); // r*t - y*ppp in one scan: 80 + 6*2 -> <2
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul(acc.zz, pp);
  acc.zzz = mul(acc.zzz, ppp);
}

// acc += p (XYZZ)
template <class P>
ZG_D void add_full29(XYZZ29<P>& acc, const XYZZ29<P>& p) {
  using F = Fp29<P>;
  if (p.is_inf()) return;
  if (acc.is_inf()) {
    acc = p;
    return;
  }
  const F u1 = mul(acc.x, p.zz);                  // 16 -> <2
  const F u2 = mul(p.x, acc.zz);
  const F s1 = mul(acc.y, p.zzz);
  const F s2 = mul(p.y, acc.zzz);
  const F pp_ = sub_k(u2, u1, P::K3);             // <5
  const F pp = sqr(pp_);
  if (is_zero_mod_p(pp)) {
    const F r0 = sub_k(s2, s1, P::K3);
    if (is_zero_mod_p(sqr(r0)))
      acc = dbl29(acc);
    else
      acc = XYZZ29<P>::inf();
    return;
  }
  const F r = sub_k(s2, s1, P::K3);               // <5
  const F ppp = mul(pp_, pp);
  const F q = mul(u1, pp);
  const F x3 = sub2_k(sqr(r), ppp, q, P::K6);     // <8
  const F t = sub_k(q, x3, P::K8);                // <10
  const F y3 = mul_sub(r, t, s1, ppp, P::KS2);    // r*t - s1*ppp in one scan: 50 + 2*2 -> <2
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul(mul(acc.zz, p.zz), pp);
  acc.zzz = mul(mul(acc.zzz, p.zzz), ppp);
}

// ---- G1 over a lane quad ----------------------------------------------------------------------
// The sums that END a multiexp (bucket folds, workgroup trees) are chains of dependent additions with few points
// in flight: a wave that holds one point per lane spends the 14 products of an addition one after another
// (about 7 us) however few of its lanes carry a live point.  Spread over four lanes -- lane q of a quad holds
// coordinate q of (X, Y, ZZ, ZZZ), nine registers instead of 36 -- the products of add-2008-s run four at a time:
//   round 1   u1 = X1 ZZ2 | s1 = Y1 ZZZ2 | u2 = X2 ZZ1 | s2 = Y2 ZZZ1      (each lane: own coordinate of one
//             operand times the coordinate two lanes over of the other)
//   round 2   pp = (u2-u1)^2 | rr = (s2-s1)^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
//   round 3   q = u1 pp | ppp = (u2-u1) pp | ZZ3 = ZZ1 ZZ2 pp | ppp again
//   round 4   X3 = rr - ppp - 2q (no product) | Y3 = r (q - X3) - s1 ppp (one fused scan) | - | ZZZ3 = ZZZ1 ZZZ2 ppp
// 4.5 product times instead of 13.5, plus 54 DPP moves.  Same bounds as add_full29.
template <int CTRL>
ZG_D uint32_t quad_perm(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
template <int CTRL, class P>
ZG_D Fp29<P> quad_perm(const Fp29<P>& a) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = quad_perm<CTRL>(a.v[i]);
  return r;
}
static constexpr int QUAD_X2 = 0x4E;  // quad_perm [2,3,0,1]: the lane two over
static constexpr int QUAD_X1 = 0xB1;  // quad_perm [1,0,3,2]
static constexpr int QUAD_B0 = 0x00, QUAD_B1 = 0x55, QUAD_B2 = 0xAA, QUAD_B3 = 0xFF;  // lane k of the quad, in all four

template <class P>
ZG_D Fp29<P> pick(bool c, const Fp29<P>& a, const Fp29<P>& b) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

// this lane's coordinate of a point that every lane of the quad holds whole
template <class P>
ZG_D Fp29<P> quad_own(const XYZZ29<P>& p) {
  const uint32_t c = threadIdx.x & 3;
  return pick(c < 2, pick(c == 0, p.x, p.y), pick(c == 2, p.zz, p.zzz));
}
// the whole point of a quad, in each of its lanes
template <class P>
ZG_D XYZZ29<P> quad_whole(const Fp29<P>& own) {
  return {quad_perm<QUAD_B0>(own), quad_perm<QUAD_B1>(own), quad_perm<QUAD_B2>(own), quad_perm<QUAD_B3>(own)};
}
template <class P>
ZG_D bool quad_is_inf(const Fp29<P>& own) {
  return quad_perm<QUAD_B2>((uint32_t)own.is_zero_limbs()) != 0;
}

// a += b, both spread over the quad.  No branch on the common paths but one that the whole wave takes together
// (identity operands are selected at the end), so every lane of a wave walks the same instructions whatever its
// quad holds.
template <class P>
ZG_D void quad_add(Fp29<P>& a, const Fp29<P>& b) {
  using F = Fp29<P>;
  const uint32_t c = threadIdx.x & 3;
  const bool hi = c >= 2;
  const bool ainf = quad_is_inf(a), binf = quad_is_inf(b);
  if (__all(ainf || binf)) {  // no quad of the wave has two points to add (sums over empty ranges, thin trees)
    a = pick(binf, a, b);
    return;
  }
  const F bx = quad_perm<QUAD_X2>(b);
  const F m1 = mul(a, bx);                          // u1 | s1 | u2 | s2 : 8*2, 5*2 -> <2
  const F mx = quad_perm<QUAD_X2>(m1);
  const F d = sub_k(pick(hi, m1, mx), pick(hi, mx, m1), P::K3);  // u2 - u1 | s2 - s1 | u2 - u1 | s2 - s1 : <5
  const F m2 = mul(pick(hi, a, d), pick(hi, b, d));  // pp | rr | ZZ1 ZZ2 | ZZZ1 ZZZ2 : 25, 4 -> <2
  if (quad_perm<QUAD_B0>((uint32_t)is_zero_mod_p(m2)) != 0 && !ainf && !binf) {
    // same x: doubling or cancellation (rare) -- every lane takes the whole points and the one-lane formulas
    XYZZ29<P> wa = quad_whole(a);
    add_full29(wa, quad_whole(b));
    a = quad_own(wa);
    return;
  }
  const F pp0 = quad_perm<QUAD_B0>(m2), d0 = quad_perm<QUAD_B0>(d);
  const F m3 = mul(pick(c == 0, m1, pick(c == 2, m2, d0)), pp0);  // q | ppp | ZZ3 | ppp : 2*2, 5*2 -> <2
  const F m2s = quad_perm<QUAD_X1>(m2), m3s = quad_perm<QUAD_X1>(m3);
  const bool l0 = c == 0;                           // lanes 0 and 1 both form X3 (lane 0 keeps it, lane 1 needs it)
  const F q = pick(l0, m3, m3s), ppp = pick(l0, m3s, m3);
  const F x3 = sub2_k(pick(l0, m2s, m2), ppp, q, P::K6);  // rr - ppp - 2q : <8
  const F t = sub_k(q, x3, P::K8);                  // <10
  const bool l3 = c == 3;
  // lane 1: r t - s1 ppp (50 + 2*2); lane 3: ZZZ1 ZZZ2 ppp through the same scan with nothing subtracted.
  // (Measured and dropped: s1 ppp on the idle lane 0 and a plain product in every lane -- half a product time less.
  // It is exact (checked limb by limb on the host, and alone on the GPU), but k_sum_range built with it by hipcc 7.2
  // returned wrong sums, so the fused scan stays.)
  const F m4 = mul_sub(pick(l3, m2, d), pick(l3, m3, t), m1, pick(l3, F::zero(), m3), P::KS2);
  const F sum = pick(l0, x3, pick(c == 2, m3, m4));
  a = pick(binf, a, pick(ainf, b, sum));
}

// ---- G2 over a lane pair -----------------------------------------------------------------------
// The same formulas over Fq2 with every value split across lanes 2k (real) and 2k+1 (imaginary), see
// mul_pair.  XYZZ29 / Affine29 then hold ONE lane's components.  Bounds in units of p, per component:
//   stored points:  X < 6,  Y < 4,  ZZ, ZZZ < 2;   affine bases canonical (< 1)
// The comment after a product is (bound of a) * (bound of b + multiple used for -b) <= 169.
template <class P>
ZG_D bool pair_is_inf(const XYZZ29<P>& p) {
  return pair_both(p.zz.is_zero_limbs());
}
template <class P>
ZG_D Fp29<P> pair_one() {
  return (threadIdx.x & 1) ? Fp29<P>::zero() : Fp29<P>::one();
}

template <class P>
ZG_D XYZZ29<P> dbl_affine_pair(const Affine29<P>& p) {
  using F = Fp29<P>;
  const F u = mul_small_k(p.y, 2);                          // <2
  const F v = mul_pair(u, u, P::KS3);                       // 2*5
  const F w = mul_pair(u, v, P::KS2);                       // 2*4
  const F s = mul_pair(p.x, v, P::KS2);                     // 1*4
  const F m = mul_small_k(mul_pair(p.x, p.x, P::KS2), 3);   // 1*3 -> <1.1, times 3: <3.3
  const F x3 = sub2_k(mul_pair(m, m, P::KS4), F::zero(), s, P::K4);  // 3.3*7.3 -> <1.2 ; < 5.2
  const F t = sub_k(s, x3, P::K6);                          // <7.1
  const F y3 = sub_k(mul_pair(t, m, P::KS4), mul_pair(p.y, w, P::KS2), P::K2);  // 7.1*7.3 ; <3.4
  return {x3, y3, v, w};
}

template <class P>
ZG_D XYZZ29<P> dbl_pair(const XYZZ29<P>& p) {
  using F = Fp29<P>;
  if (pair_is_inf(p)) return p;
  const F u = mul_small_k(p.y, 2);                          // <8
  const F v = mul_pair(u, u, P::KS8);                       // 8*16 = 128 -> <1.76
  const F w = mul_pair(u, v, P::KS2);                       // 8*3.76
  const F s = mul_pair(p.x, v, P::KS2);                     // 6*3.76
  const F m = mul_small_k(mul_pair(p.x, p.x, P::KS6), 3);   // 6*12 = 72 -> <1.43, times 3: <4.3
  const F x3 = sub2_k(mul_pair(m, m, P::KS6), F::zero(), s, P::K4);  // 4.3*10.3 -> <1.27 ; 2s < 2.3 ; <5.3
  const F t = sub_k(s, x3, P::K6);                          // <7.2
  const F y3 = sub_k(mul_pair(t, m, P::KS6), mul_pair(p.y, w, P::KS2), P::K2);  // 7.2*10.3 = 74 ; 4*3.2 ; <3.5
  return {x3, y3, mul_pair(p.zz, v, P::KS2), mul_pair(p.zzz, w, P::KS2)};
}

// acc += p (affine, not the identity)
template <class P>
ZG_D void add_mixed_pair(XYZZ29<P>& acc, const Affine29<P>& p) {
  using F = Fp29<P>;
  if (pair_is_inf(acc)) {
    acc = {p.x, p.y, pair_one<P>(), pair_one<P>()};
    return;
  }
  const F u2 = mul_pair(acc.zz, p.x, P::KS2);       // 2*3
  const F s2 = mul_pair(acc.zzz, p.y, P::KS2);
  const F pp_ = sub_k(u2, acc.x, P::K6);            // <7.1
  // the two squares by the complex squaring (one product per lane each); P^2 comes out below 2.27p -- real
  // component 14.2 * 15.1 = 214 -- and its three users take it with the next constant up
  const F pp = sqr_pair(pp_, P::K8);                // <2.27
  if (pair_both(is_zero_mod_p3(pp))) {              // same x: doubling or cancellation (rare)
    const F r0 = sub_k(s2, acc.y, P::K4);
    if (pair_both(is_zero_mod_p(mul_pair(r0, r0, P::KS6))))
      acc = dbl_affine_pair(p);
    else
      acc = XYZZ29<P>::inf();
    return;
  }
  const F r = sub_k(s2, acc.y, P::K4);              // <5.1
  const F ppp = mul_pair(pp_, pp, P::KS3);          // 7.1*5.27
  const F q = mul_pair(acc.x, pp, P::KS3);          // 6*5.27
  const F x3 = sub2_k(sqr_pair(r, P::K6), ppp, q, P::K4);  // 10.2*11.1 = 113 -> <1.67 ; ppp + 2q < 3.7 ; <5.7
  const F t = sub_k(q, x3, P::K6);                  // <7.2
  // t r - Y ppp in one scan: 7.2 * (5.1 + 6) + 2 * 4 * 1.3 = 90
  const F y3 = mul_sub_pair(t, r, sub_k(F::zero(), r, P::K6), acc.y, sub_k(F::zero(), acc.y, P::K4), ppp);
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul_pair(acc.zz, pp, P::KS3);
  acc.zzz = mul_pair(acc.zzz, ppp, P::KS2);
}

// acc += p (XYZZ)
template <class P>
ZG_D void add_full_pair(XYZZ29<P>& acc, const XYZZ29<P>& p) {
  using F = Fp29<P>;
  if (pair_is_inf(p)) return;
  if (pair_is_inf(acc)) {
    acc = p;
    return;
  }
  const F u1 = mul_pair(acc.x, p.zz, P::KS2);       // 6*4
  const F u2 = mul_pair(p.x, acc.zz, P::KS2);
  const F s1 = mul_pair(acc.y, p.zzz, P::KS2);
  const F s2 = mul_pair(p.y, acc.zzz, P::KS2);
  const F pp_ = sub_k(u2, u1, P::K2);               // <3.2
  const F pp = mul_pair(pp_, pp_, P::KS4);          // 3.2*7.2
  if (pair_both(is_zero_mod_p(pp))) {
    const F r0 = sub_k(s2, s1, P::K2);
    if (pair_both(is_zero_mod_p(mul_pair(r0, r0, P::KS4))))
      acc = dbl_pair(acc);
    else
      acc = XYZZ29<P>::inf();
    return;
  }
  const F r = sub_k(s2, s1, P::K2);                 // <3.2
  const F ppp = mul_pair(pp_, pp, P::KS2);
  const F q = mul_pair(u1, pp, P::KS2);
  const F x3 = sub2_k(mul_pair(r, r, P::KS4), ppp, q, P::K4);  // <5.2
  const F t = sub_k(q, x3, P::K6);                  // <7.2
  const F y3 = sub_k(mul_pair(t, r, P::KS4), mul_pair(s1, ppp, P::KS2), P::K2);  // 7.2*7.2 ; <3.4
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul_pair(mul_pair(acc.zz, p.zz, P::KS2), pp, P::KS2);
  acc.zzz = mul_pair(mul_pair(acc.zzz, p.zzz, P::KS2), ppp, P::KS2);
}

// ---- G2 over a lane octet ---------------------------------------------------------------------
// quad_add's schedule with every Fq2 coordinate on a lane pair: lane 2 pos + comp of an octet holds component comp
// (0 real, 1 imaginary) of the coordinate at position pos, positions being X, Y, ZZZ, ZZ -- so that the coordinate an
// addition's first products pair up (X with ZZ, Y with ZZZ) sits at position 3 - pos, which together with the other
// component is the mirror image inside the octet (one DPP move).  Four mul_pair times instead of fourteen.
//   stored points:  X < 6,  Y < 4,  ZZ, ZZZ < 2  per component, as for the lane-pair form
ZG_D uint32_t oct_mirror(uint32_t x) {  // lane 7 - l of the octet: position 3 - pos, the OTHER component
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141 /* row_half_mirror */, 0xf, 0xf, true);
}
template <int POS>
ZG_D uint32_t oct_from(uint32_t x) {  // position POS of the octet, same component
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x0019 | (POS << 6));  // lane = (lane & 0b11001) | POS << 1
}
template <class P>
ZG_D Fp29<P> oct_mirror(const Fp29<P>& a) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = oct_mirror(a.v[i]);
  return r;
}
template <int POS, class P>
ZG_D Fp29<P> oct_from(const Fp29<P>& a) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = oct_from<POS>(a.v[i]);
  return r;
}
ZG_D uint32_t oct_pos() { return (threadIdx.x >> 1) & 3; }
// which of x, y, zz, zzz a position holds
ZG_D uint32_t oct_coord(uint32_t pos) { return pos < 2 ? pos : 5 - pos; }

// this lane's part of a point that every lane pair of the octet holds whole (in the lane-pair form)
template <class P>
ZG_D Fp29<P> oct_own(const XYZZ29<P>& p) {
  const uint32_t pos = oct_pos();
  return pick(pos < 2, pick(pos == 0, p.x, p.y), pick(pos == 2, p.zzz, p.zz));
}
// the whole point of an octet in the lane-pair form, in each of its pairs
template <class P>
ZG_D XYZZ29<P> oct_whole(const Fp29<P>& own) {
  return {oct_from<0>(own), oct_from<1>(own), oct_from<3>(own), oct_from<2>(own)};
}
template <class P>
ZG_D bool oct_is_inf(const Fp29<P>& own) {
  return pair_both(oct_from<3>((uint32_t)own.is_zero_limbs()) != 0);
}

// a += b, both spread over the octet; no branch on the common paths (see quad_add)
template <class P>
ZG_D void oct_add(Fp29<P>& a, const Fp29<P>& b) {
  using F = Fp29<P>;
  const uint32_t pos = oct_pos();
  const bool lo = pos < 2;
  const bool ainf = oct_is_inf(a), binf = oct_is_inf(b);
  if (__all(ainf || binf)) {  // no octet of the wave has two points to add
    a = pick(binf, a, b);
    return;
  }
  const F bxo = oct_mirror(b), bx = pair_swap(bxo);  // b's coordinate at 3 - pos: the partner's component, this lane's
  // u1 = X1 ZZ2 | s1 = Y1 ZZZ2 | s2 = ZZZ1 Y2 | u2 = ZZ1 X2 : 6 * (6 + 8) -> < 2 (the widest pairing bounds them all)
  const F m1 = mul_pair_x(a, pair_swap(a), bx, bxo, P::KS8);
  const F mx = pair_swap(oct_mirror(m1));
  const F d = sub_k(pick(lo, mx, m1), pick(lo, m1, mx), P::K2);  // u2 - u1 | s2 - s1 | s2 - s1 | u2 - u1 : < 3.2
  const F m2 = mul_pair(pick(lo, d, a), pick(lo, d, b), P::KS4);  // pp | rr | ZZZ1 ZZZ2 | ZZ1 ZZ2 : 3.2 * 7.2
  if (pair_both(oct_from<0>((uint32_t)is_zero_mod_p(m2)) != 0) && !ainf && !binf) {
    // same x: doubling or cancellation (rare) -- every pair takes the whole points and the lane-pair formulas
    XYZZ29<P> wa = oct_whole(a);
    add_full_pair(wa, oct_whole(b));
    a = oct_own(wa);
    return;
  }
  const F pp0 = oct_from<0>(m2), d0 = oct_from<0>(d);
  // q = u1 pp | ppp | ppp | ZZ3 = ZZ1 ZZ2 pp : 3.2 * (2 + 2)
  const F m3 = mul_pair(pick(pos == 0, m1, pick(pos == 3, m2, d0)), pp0, P::KS2);
  const F m1s = quad_perm<QUAD_X2>(m1), m2s = quad_perm<QUAD_X2>(m2), m3s = quad_perm<QUAD_X2>(m3);  // positions 0 <-> 1
  const bool p0 = pos == 0;
  const F q = pick(p0, m3, m3s), ppp = pick(p0, m3s, m3);
  const F x3 = sub2_k(pick(p0, m2s, m2), ppp, q, P::K4);  // rr - ppp - 2q : < 5.2
  const F t = sub_k(q, x3, P::K6);                        // < 7.2
  // s1 ppp | t r | ZZZ1 ZZZ2 ppp | - : 7.2 * (3.2 + 4)
  const F m4 = mul_pair(pick(p0, m1s, pick(pos == 1, t, m2)), pick(p0, m3s, pick(pos == 1, d, m3)), P::KS4);
  const F y3 = sub_k(m4, quad_perm<QUAD_X2>(m4), P::K2);  // position 1: t r - s1 ppp : < 3.4
  const F sum = pick(p0, x3, pick(pos == 1, y3, pick(pos == 2, m4, m3)));
  a = pick(binf, a, pick(ainf, b, sum));
}

// ---- the two sides of the G2 subgroup test psi(P) = [6x^2] P (see g2_in_subgroup in curve.hpp), on a lane pair ----
// [6 x^2] P for an affine, non-identity P; 6 x^2 = 0x6f4d8248eeb859fb f83e9682e87cfd46 (127 bits), x the BN parameter.
template <class P>
ZG_D XYZZ29<P> g2_times_6x2_pair(const Affine29<P>& base) {
  const uint64_t e_hi = 0x6f4d8248eeb859fbull, e_lo = 0xf83e9682e87cfd46ull;
  XYZZ29<P> acc = XYZZ29<P>::inf();
  add_mixed_pair(acc, base);  // bit 126
#pragma unroll 1
  for (int bit = 125; bit >= 0; bit--) {
    acc = dbl_pair(acc);
    if (((bit < 64 ? e_lo >> bit : e_hi >> (bit - 64)) & 1) != 0) add_mixed_pair(acc, base);
  }
  return acc;
}
// psi(P) = (conj(x) kx, conj(y) ky) with (kx, ky) this lane's components of xi^((q-1)/3), xi^((q-1)/2) (canonical,
// radix-2^261 Montgomery form); components below 2p with normalised limbs.
template <class P>
ZG_D Affine29<P> g2_psi_pair(const Affine29<P>& base, const Fp29<P>& kx, const Fp29<P>& ky) {
  using F = Fp29<P>;
  const bool im = (threadIdx.x & 1) != 0;
  const F cx = im ? sub_k(F::zero(), base.x, P::K2) : base.x;   // conj: the imaginary lane holds 2p - x1
  const F cy = im ? sub_k(F::zero(), base.y, P::K2) : base.y;
  return {mul_pair(cx, kx, P::KS2), mul_pair(cy, ky, P::KS2)};   // 2 * (1 + 2)
}
// Is the point (X / ZZ, Y / ZZZ) of acc (not the identity) the affine point q (components below 2p)?
template <class P>
ZG_D bool pair_equals_affine(const XYZZ29<P>& acc, const Affine29<P>& q) {
  using F = Fp29<P>;
  const F dx = sub_k(mul_pair(acc.zz, q.x, P::KS3), acc.x, P::K6);    // 2 * (2 + 3); X < 6p; < 8p
  const F dy = sub_k(mul_pair(acc.zzz, q.y, P::KS3), acc.y, P::K4);   // Y < 4p; < 6p
  const F one = pair_one<P>();
  // a product by one brings the differences below 2p with strictly normalised limbs, where zero is 0 or p
  const bool zx = is_zero_mod_p(mul_pair(dx, one, P::KS2)), zy = is_zero_mod_p(mul_pair(dy, one, P::KS2));  // 8 * (1 + 2)
  return pair_both(zx && zy);
}

// ---- memory formats -------------------------------------------------------------------------
// Packed affine base (x || y, 8 x 32-bit words each, canonical, radix-2^261 Montgomery form).
template <class P, class Q>
ZG_D Affine29<P> unpack_affine(const Affine<Fp<Q>>& a) {
  return {unpack29<P>(a.x), unpack29<P>(a.y)};
}

// XYZZ29 -> the library's XYZZ<Fq> (radix-2^256 Montgomery form, canonical), e.g. for the host.
template <class P, class Q>
ZG_D XYZZ<Fp<Q>> export_xyzz(const XYZZ29<P>& p) {
  if (p.is_inf()) return XYZZ<Fp<Q>>::inf();
  return {to_r256<P, Q>(p.x), to_r256<P, Q>(p.y), to_r256<P, Q>(p.zz), to_r256<P, Q>(p.zzz)};
}

using G1Affine29 = Affine29<Fq29Params>;
using G1XYZZ29 = XYZZ29<Fq29Params>;

}  // namespace zg
