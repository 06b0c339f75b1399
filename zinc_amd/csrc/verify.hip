// Groth16 verification on the host: bellman_ce 0.3.0 `verify_proof(&prepare_verifying_key(&vk), &proof, &inputs)`
// as the reference calls it (zinc-vm/src/core/facade.rs:32-38; zinc-tester/src/runners/proof_check.rs:243-264 is
// the check every proof test ends in).
//
// The work is four Miller loops and one final exponentiation whatever the circuit size -- milliseconds of one host
// core -- so nothing here runs on the GPU; the code is in the library so that a host that proves through it can
// also check what it produced without linking a second pairing implementation.
//
//   accepts iff  e(A, B) * e(acc, -gamma) * e(C, -delta) == e(alpha, beta),   acc = ic[0] + sum_i x_i ic[i+1]
// computed as  final_exp( ML(B, A) * ML(gamma, -acc) * ML(delta, -C) * ML(beta, -alpha) ) == 1.
//
// Pairing: optimal ate on BN254 (pairing_ce bn256: loop count 6x + 2 with x = 4965661367192848881, then the two
// Frobenius-twisted additions), written twice.
//   * The PLAIN form, for obviousness: G2 points are untwisted into E(Fq12) : y^2 = x^3 + 3, the loop uses affine
//     chord-and-tangent steps there, Frobenius is a power by q and the final exponentiation a power by
//     (q^12 - 1) / r, so no Frobenius or cyclotomic constants are needed.  37 ms per proof.
//   * The form that runs: the same chords and tangents on the twist itself (slopes in Fq2: the untwisted slope is
//     the twist's slope times w, so the lines are the same field elements), the four Miller loops as ONE loop that
//     squares once per bit and inverts the four slope denominators together, Frobenius through the six constants
//     xi^(i (q-1) / 6), and the final exponentiation as the easy part (q^6 - 1)(q^2 + 1) followed by the
//     q^3 + lambda_2 q^2 + lambda_1 q + lambda_0 expansion of (q^4 - q^2 + 1) / r in x (three powers by x).
//     Same value as the plain form, about a tenth of the time.
// ZG16_VERIFY_PLAIN=1 runs the plain form; ZG16_VERIFY_CHECK=1 runs both and reports ZG16_ERR_DEVICE unless the two
// Miller products and the two exponentiations agree element for element (tests/test_host.py).
// Tower: Fq2 = Fq[u] / (u^2 + 1), Fq6 = Fq2[v] / (v^3 - xi) with xi = 9 + u, Fq12 = Fq6[w] / (w^2 - v).
#include <stdlib.h>
#include <string.h>

#include "curve.hpp"
#include "../../include/zinc_groth16.h"

namespace zg {
namespace {

#include "pairing_consts.inc"

Fq2 mul_xi(const Fq2& a) {  // (a0 + a1 u)(9 + u)
  const Fq n0 = add(dbl(dbl(dbl(a.c0))), a.c0), n1 = add(dbl(dbl(dbl(a.c1))), a.c1);
  return {sub(n0, a.c1), add(n1, a.c0)};
}

struct Fq6 {
  Fq2 c0, c1, c2;
  static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
  static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
};
Fq6 add(const Fq6& a, const Fq6& b) { return {add(a.c0, b.c0), add(a.c1, b.c1), add(a.c2, b.c2)}; }
Fq6 sub(const Fq6& a, const Fq6& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1), sub(a.c2, b.c2)}; }
Fq6 neg(const Fq6& a) { return {neg(a.c0), neg(a.c1), neg(a.c2)}; }
Fq6 mul_v(const Fq6& a) { return {mul_xi(a.c2), a.c0, a.c1}; }
Fq6 mul(const Fq6& a, const Fq6& b) {
  const Fq2 t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1), t2 = mul(a.c2, b.c2);
  const Fq2 c0 = add(t0, mul_xi(sub(sub(mul(add(a.c1, a.c2), add(b.c1, b.c2)), t1), t2)));
  const Fq2 c1 = add(sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), t0), t1), mul_xi(t2));
  const Fq2 c2 = add(sub(sub(mul(add(a.c0, a.c2), add(b.c0, b.c2)), t0), t2), t1);
  return {c0, c1, c2};
}
Fq6 inv(const Fq6& a) {
  const Fq2 t0 = sub(sqr(a.c0), mul_xi(mul(a.c1, a.c2)));
  const Fq2 t1 = sub(mul_xi(sqr(a.c2)), mul(a.c0, a.c1));
  const Fq2 t2 = sub(sqr(a.c1), mul(a.c0, a.c2));
  const Fq2 d = inv(add(mul(a.c0, t0), mul_xi(add(mul(a.c2, t1), mul(a.c1, t2)))));
  return {mul(t0, d), mul(t1, d), mul(t2, d)};
}

struct Fq12 {
  Fq6 c0, c1;
  static Fq12 zero() { return {Fq6::zero(), Fq6::zero()}; }
  static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool is_one() const {
    return c0.c0 == Fq2::one() && c0.c1.is_zero() && c0.c2.is_zero() && c1.is_zero();
  }
};
Fq12 add(const Fq12& a, const Fq12& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
Fq12 sub(const Fq12& a, const Fq12& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
Fq12 mul(const Fq12& a, const Fq12& b) {
  const Fq6 t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
  return {add(t0, mul_v(t1)), sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), t0), t1)};
}
Fq12 inv(const Fq12& a) {
  const Fq6 d = inv(sub(mul(a.c0, a.c0), mul_v(mul(a.c1, a.c1))));
  return {mul(a.c0, d), neg(mul(a.c1, d))};
}
Fq12 pow(const Fq12& a, const uint64_t* e, int limbs) {
  Fq12 r = Fq12::one();
  bool started = false;
  for (int i = limbs - 1; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      if (started) r = mul(r, r);
      if ((e[i] >> b) & 1) {
        r = started ? mul(r, a) : a;
        started = true;
      }
    }
  return r;
}
Fq12 frobenius(const Fq12& a) {  // a^q
  uint64_t q[4];
  for (int i = 0; i < 4; i++) q[i] = (uint64_t)FqParams::mod(2 * i) | ((uint64_t)FqParams::mod(2 * i + 1) << 32);
  return pow(a, q, 4);
}
Fq12 from_fq(const Fq& x) {
  Fq12 r = Fq12::zero();
  r.c0.c0.c0 = x;
  return r;
}

// A point of E(Fq12) : y^2 = x^3 + 3, affine.
struct P12 {
  Fq12 x, y;
};
// (x', y') on the twist y'^2 = x'^3 + 3 / xi  ->  (x' w^2, y' w^3) on E  (w^2 = v, w^3 = v w, w^6 = xi)
P12 untwist(const G2Affine& q) {
  P12 p = {Fq12::zero(), Fq12::zero()};
  p.x.c0.c1 = q.x;  // x' v
  p.y.c1.c1 = q.y;  // y' v w
  return p;
}
Fq12 three() {
  const Fq one = Fq::one();
  return from_fq(add(add(one, one), one));
}

// The line through t and p (the tangent when they coincide) evaluated at (px, py), and t <- t + p.
// Returns false when the step is not a chord or a tangent (vertical line): cannot happen for points of prime order
// inside the loop, and the caller treats it as a failed verification.
bool step(P12& t, const P12& p, const Fq12& px, const Fq12& py, Fq12* line) {
  Fq12 m;
  const Fq12 dx = sub(p.x, t.x), dy = sub(p.y, t.y);
  if (!dx.is_zero()) {
    m = mul(dy, inv(dx));
  } else if (dy.is_zero()) {
    const Fq12 two_y = add(t.y, t.y);
    if (two_y.is_zero()) return false;
    m = mul(mul(three(), mul(t.x, t.x)), inv(two_y));
  } else {
    return false;
  }
  *line = sub(mul(m, sub(px, t.x)), sub(py, t.y));
  const Fq12 x3 = sub(sub(mul(m, m), t.x), p.x);
  const Fq12 y3 = sub(mul(m, sub(t.x, x3)), t.y);
  t = {x3, y3};
  return true;
}

// f_{6x+2, Q}(P) times the two Frobenius lines; Q in G2, P in G1, neither the identity.
bool miller_loop(const G2Affine& q2, const G1Affine& p1, Fq12* out) {
  static const uint64_t ATE_LOOP_COUNT_HI = 1, ATE_LOOP_COUNT_LO = 0x9d797039be763ba8ull;  // 29793968203157093288 = 2^64 + lo
  const P12 q = untwist(q2);
  const Fq12 px = from_fq(p1.x), py = from_fq(p1.y);
  P12 t = q;
  Fq12 f = Fq12::one(), line;
  (void)ATE_LOOP_COUNT_HI;  // the top bit (bit 64) is where the loop starts: t = q, f = 1
  for (int i = 63; i >= 0; i--) {
    const P12 tt = t;
    if (!step(t, tt, px, py, &line)) return false;
    f = mul(mul(f, f), line);
    if ((ATE_LOOP_COUNT_LO >> i) & 1) {
      if (!step(t, q, px, py, &line)) return false;
      f = mul(f, line);
    }
  }
  const P12 q1 = {frobenius(q.x), frobenius(q.y)};
  P12 nq2 = {frobenius(q1.x), frobenius(q1.y)};
  nq2.y = sub(Fq12::zero(), nq2.y);
  if (!step(t, q1, px, py, &line)) return false;
  f = mul(f, line);
  // the last line only: the point it leads to is not needed
  P12 t2 = t;
  if (!step(t2, nq2, px, py, &line)) {
    // t + (-pi^2(Q)) may be the identity: the line is then the vertical through t
    line = sub(px, t.x);
  }
  f = mul(f, line);
  *out = f;
  return true;
}

// ---- the form that runs ----------------------------------------------------------------------------------------
Fq2 conj(const Fq2& a) { return {a.c0, neg(a.c1)}; }
Fq2 mul_fq(const Fq2& a, const Fq& k) { return {mul(a.c0, k), mul(a.c1, k)}; }
Fq12 conj(const Fq12& a) { return {a.c0, neg(a.c1)}; }  // a^(q^6)

// gamma^i for gamma = xi^((q - 1) / 6): w^q = gamma w, so (a_i w^i)^q = conj(a_i) gamma^i w^i.
struct FrobConsts {
  Fq2 g[6];
};
const FrobConsts& frob_consts() {
  static const FrobConsts consts = [] {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FqParams::mod(i);
    e[0] -= 1;  // q is odd
    uint64_t rem = 0;
    for (int i = 7; i >= 0; i--) {  // (q - 1) / 6, exact
      const uint64_t cur = (rem << 32) | e[i];
      e[i] = (uint32_t)(cur / 6);
      rem = cur % 6;
    }
    const Fq2 xi = {from_u64<FqParams>(9), Fq::one()};
    Fq2 g = Fq2::one();
    for (int i = 7; i >= 0; i--)
      for (int b = 31; b >= 0; b--) {
        g = sqr(g);
        if ((e[i] >> b) & 1) g = mul(g, xi);
      }
    FrobConsts c;
    c.g[0] = Fq2::one();
    for (int i = 1; i < 6; i++) c.g[i] = mul(c.g[i - 1], g);
    return c;
  }();
  return consts;
}
// tower slot <-> power of w: c0 = (w^0, w^2, w^4), c1 = (w^1, w^3, w^5)
Fq12 frobenius_fast(const Fq12& a) {
  const FrobConsts& F = frob_consts();
  return {{conj(a.c0.c0), mul(conj(a.c0.c1), F.g[2]), mul(conj(a.c0.c2), F.g[4])},
          {mul(conj(a.c1.c0), F.g[1]), mul(conj(a.c1.c1), F.g[3]), mul(conj(a.c1.c2), F.g[5])}};
}
Fq12 frobenius2_fast(const Fq12& a) { return frobenius_fast(frobenius_fast(a)); }
// pi(x' w^2, y' w^3) = (conj(x') gamma^2 w^2, conj(y') gamma^3 w^3): the Frobenius of E(Fq12) seen from the twist
G2Affine twist_frobenius(const G2Affine& q) {
  const FrobConsts& F = frob_consts();
  return {mul(conj(q.x), F.g[2]), mul(conj(q.y), F.g[3])};
}

Fq12 pow_x(const Fq12& a) {
  static const uint64_t BN_X = 4965661367192848881ull;
  return pow(a, &BN_X, 1);
}

// f^((q^12 - 1) / r).  Easy part: f^(q^6 - 1) = conj(f) / f, then ^(q^2 + 1).  Hard part (q^4 - q^2 + 1) / r =
// q^3 + (6x^2 + 1) q^2 + (-36x^3 - 18x^2 - 12x + 1) q + (-36x^3 - 30x^2 - 18x - 2), evaluated with three powers by x
// and the addition chain of Devegili, Scott and Dahab ("Implementing cryptographic pairings over Barreto-Naehrig
// curves", section on the final exponentiation); inverses of unitary elements are conjugates.
Fq12 final_exp_fast(const Fq12& f) {
  Fq12 t1 = mul(conj(f), inv(f));
  t1 = mul(frobenius2_fast(t1), t1);
  const Fq12 fp = frobenius_fast(t1), fp2 = frobenius2_fast(t1), fp3 = frobenius_fast(fp2);
  const Fq12 fu = pow_x(t1), fu2 = pow_x(fu), fu3 = pow_x(fu2);
  const Fq12 y3 = conj(frobenius_fast(fu)), fu2p = frobenius_fast(fu2), fu3p = frobenius_fast(fu3);
  const Fq12 y2 = frobenius2_fast(fu2);
  const Fq12 y0 = mul(mul(fp, fp2), fp3), y1 = conj(t1), y5 = conj(fu2);
  const Fq12 y4 = conj(mul(fu, fu2p)), y6 = conj(mul(fu3, fu3p));
  Fq12 t0 = mul(mul(mul(y6, y6), y4), y5);
  Fq12 t = mul(mul(y3, y5), t0);
  t0 = mul(t0, y2);
  t = mul(mul(t, t), t0);
  t = mul(t, t);
  t0 = mul(t, y1);
  t = mul(t, y0);
  t0 = mul(t0, t0);
  return mul(t0, t);
}

// d[k] <- 1 / d[k] with one inversion; false if any is zero.
bool batch_inv(Fq2* d, int n) {
  Fq2 pre[4];
  Fq2 run = Fq2::one();
  for (int k = 0; k < n; k++) {
    if (d[k].is_zero()) return false;
    pre[k] = run;
    run = mul(run, d[k]);
  }
  Fq2 r = inv(run);
  for (int k = n - 1; k >= 0; k--) {
    const Fq2 dk = d[k];
    d[k] = mul(r, pre[k]);
    r = mul(r, dk);
  }
  return true;
}

struct MillerPair {
  G2Affine q;  // on the twist
  G1Affine p;
};
// The line of slope lam * w through the untwisted image of t = (x, y), at (px, py):
//   lam w (px - x w^2) - (py - y w^3) = -py + (lam px) w + (y - lam x) w^3
Fq12 line_at(const Fq2& lam, const G2Affine& t, const G1Affine& p) {
  Fq12 l = Fq12::zero();
  l.c0.c0 = {neg(p.y), Fq::zero()};
  l.c1.c0 = mul_fq(lam, p.x);
  l.c1.c1 = sub(t.y, mul(lam, t.x));
  return l;
}
// t <- t + o along the line of slope lam (o = t for a tangent)
void advance(G2Affine& t, const G2Affine& o, const Fq2& lam) {
  const Fq2 x3 = sub(sub(sqr(lam), t.x), o.x);
  t = {x3, sub(mul(lam, sub(t.x, x3)), t.y)};
}

// prod_k f_{6x+2, Q_k}(P_k) with the Frobenius lines; no pair holds an identity; n <= 4.
bool multi_miller(const MillerPair* pr, int n, Fq12* out) {
  static const uint64_t ATE_LOOP_COUNT_LO = 0x9d797039be763ba8ull;  // the loop count is 2^64 + this
  G2Affine t[4];
  Fq2 den[4];
  for (int k = 0; k < n; k++) t[k] = pr[k].q;
  Fq12 f = Fq12::one();
  for (int i = 63; i >= 0; i--) {
    f = mul(f, f);
    for (int k = 0; k < n; k++) den[k] = dbl(t[k].y);
    if (!batch_inv(den, n)) return false;
    for (int k = 0; k < n; k++) {
      const Fq2 xx = sqr(t[k].x);
      const Fq2 lam = mul(add(dbl(xx), xx), den[k]);
      f = mul(f, line_at(lam, t[k], pr[k].p));
      advance(t[k], G2Affine(t[k]), lam);
    }
    if ((ATE_LOOP_COUNT_LO >> i) & 1) {
      for (int k = 0; k < n; k++) den[k] = sub(pr[k].q.x, t[k].x);
      if (!batch_inv(den, n)) return false;
      for (int k = 0; k < n; k++) {
        const Fq2 lam = mul(sub(pr[k].q.y, t[k].y), den[k]);
        f = mul(f, line_at(lam, t[k], pr[k].p));
        advance(t[k], pr[k].q, lam);
      }
    }
  }
  for (int k = 0; k < n; k++) {
    const G2Affine q1 = twist_frobenius(pr[k].q);
    G2Affine nq2 = twist_frobenius(q1);
    nq2.y = neg(nq2.y);
    Fq2 d = sub(q1.x, t[k].x);
    if (d.is_zero()) return false;
    Fq2 lam = mul(sub(q1.y, t[k].y), inv(d));
    f = mul(f, line_at(lam, t[k], pr[k].p));
    advance(t[k], q1, lam);
    // the last line only: the point it leads to is not needed
    d = sub(nq2.x, t[k].x);
    if (!d.is_zero()) {
      lam = mul(sub(nq2.y, t[k].y), inv(d));
      f = mul(f, line_at(lam, t[k], pr[k].p));
    } else if (sub(nq2.y, t[k].y).is_zero()) {
      // the same point: the tangent
      const Fq2 two_y = dbl(t[k].y);
      if (two_y.is_zero()) return false;
      const Fq2 xx = sqr(t[k].x);
      f = mul(f, line_at(mul(add(dbl(xx), xx), inv(two_y)), t[k], pr[k].p));
    } else {
      // t + (-pi^2(Q)) is the identity: the vertical through t, px - x w^2
      Fq12 l = Fq12::zero();
      l.c0.c0 = {pr[k].p.x, Fq::zero()};
      l.c0.c1 = neg(t[k].x);
      f = mul(f, l);
    }
  }
  *out = f;
  return true;
}

bool equal12(const Fq12& a, const Fq12& b) { return sub(a, b).is_zero(); }

template <class A>
A load_point(const uint64_t* p) {
  A r;
  memcpy(&r, p, sizeof r);
  return r;
}

bool on_curve_g1(const G1Affine& p) {
  if (p.is_inf()) return true;
  const Fq one = Fq::one(), b = add(add(one, one), one);
  return sqr(p.y) == add(mul(sqr(p.x), p.x), b);
}

}  // namespace
}  // namespace zg

using namespace zg;

extern "C" zg16_status zg16_verify_proof(const zg16_vk_view* vk, const zg16_proof* proof, const uint64_t* public_inputs,
                                         size_t n_inputs, int* accepted) {
  if (!vk || !proof || !accepted || (n_inputs && !public_inputs)) return ZG16_ERR_INVALID_ARGUMENT;
  *accepted = 0;
  if (!vk->alpha_g1 || !vk->beta_g2 || !vk->gamma_g2 || !vk->delta_g2 || !vk->ic) return ZG16_ERR_MALFORMED_KEY;
  if (n_inputs + 1 != vk->n_ic) return ZG16_ERR_MALFORMED_KEY;  // bellman: MalformedVerifyingKey
  // acc = ic[0] + sum x_i ic[i + 1]
  G1XYZZ acc = G1XYZZ::from_affine(load_point<G1Affine>(vk->ic));
  for (size_t i = 0; i < n_inputs; i++) {
    Fr x;
    memcpy(x.v, public_inputs + 4 * i, 32);
    if (!is_canonical<FrParams>(x.v)) return ZG16_ERR_INVALID_ARGUMENT;
    const Fr xc = from_mont(x);
    add_full(acc, scalar_mul_w4(G1XYZZ::from_affine(load_point<G1Affine>(vk->ic + 8 * (i + 1))), xc.v));
  }
  const G1Affine a = load_point<G1Affine>(proof->a), c = load_point<G1Affine>(proof->c), alpha = load_point<G1Affine>(vk->alpha_g1);
  const G2Affine b = load_point<G2Affine>(proof->b), beta = load_point<G2Affine>(vk->beta_g2),
                 gamma = load_point<G2Affine>(vk->gamma_g2), delta = load_point<G2Affine>(vk->delta_g2);
  // What bellman's `Proof::read` refuses never reaches its verifier (pairing_ce's `into_affine()`: coordinates below
  // q, on the curve, in the prime-order subgroup); a caller that hands over raw limbs gets the same answer here:
  // not a proof, rejected.  B is the one untrusted G2 point; a Miller loop on a point outside G2 proves nothing.
  for (const uint64_t* limbs : {proof->a, proof->a + 4, proof->c, proof->c + 4, proof->b, proof->b + 4, proof->b + 8, proof->b + 12})
    if (!is_canonical<FqParams>((const uint32_t*)limbs)) return ZG16_OK;
  if (!on_curve_g1(a) || !on_curve_g1(c)) return ZG16_OK;
  if (!on_curve(b, g2_subgroup_consts().curve_b) || !g2_in_subgroup(b, g2_subgroup_consts())) return ZG16_OK;
  // e(P, Q) with either argument the identity is one: such a factor is skipped
  struct Pair {
    G2Affine q;
    G1Affine p;
  } pairs[4] = {{b, a}, {gamma, neg(to_affine(acc))}, {delta, neg(c)}, {beta, neg(alpha)}};
  static const bool plain = getenv("ZG16_VERIFY_PLAIN") && atoi(getenv("ZG16_VERIFY_PLAIN"));
  static const bool check = getenv("ZG16_VERIFY_CHECK") && atoi(getenv("ZG16_VERIFY_CHECK"));
  MillerPair live[4];
  int n_live = 0;
  for (const Pair& pr : pairs)
    if (!pr.p.is_inf() && !pr.q.is_inf()) live[n_live++] = {pr.q, pr.p};
  Fq12 f_plain = Fq12::one(), f = Fq12::one();
  bool ok_plain = true, ok = true;
  if (plain || check) {
    for (int k = 0; k < n_live && ok_plain; k++) {
      Fq12 ml;
      ok_plain = miller_loop(live[k].q, live[k].p, &ml);
      if (ok_plain) f_plain = mul(f_plain, ml);
    }
  }
  if (!plain || check) ok = n_live ? multi_miller(live, n_live, &f) : true;
  if (check && (ok != ok_plain || (ok && !equal12(f, f_plain)))) return ZG16_ERR_DEVICE;
  if (plain) {
    ok = ok_plain;
    f = f_plain;
  }
  if (!ok) return ZG16_OK;  // a degenerate step: not a valid proof / key
  Fq12 e = plain ? pow(f, FINAL_EXP, FINAL_EXP_LIMBS) : final_exp_fast(f);
  if (check && !equal12(e, pow(f, FINAL_EXP, FINAL_EXP_LIMBS))) return ZG16_ERR_DEVICE;
  *accepted = e.is_one() ? 1 : 0;
  return ZG16_OK;
}
