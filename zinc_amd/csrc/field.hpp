// BN254 prime fields in Montgomery form (R = 2^256) for host and gfx950 device code.
//
// Memory layout of one element is 8 little-endian u32 limbs = the 4 x u64 little-endian
// Montgomery limbs that ff_ce 0.6.0 keeps in memory (Cargo.lock:1363-1366 of the reference pins
// it), so a Rust `&[Fr]` can be handed to the kernels without conversion.
//
// The multiplier is an 8 x 32-bit CIOS loop: the only wide integer multiply the CDNA4 VALU has
// is v_mad_u64_u32 (32 x 32 + 64 -> 64), so 32-bit limbs are the native digit size; each
// `(uint64_t)a * b + c` below is written so that it lowers to exactly one of them.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZG_HD __host__ __device__ __forceinline__
#define ZG_D __device__ __forceinline__
#else
#define ZG_HD inline
#define ZG_D inline
#endif

namespace zg {

struct FrParams {
  // r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
  ZG_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return m[i];
  }
  // R mod r  (Montgomery form of 1)
  ZG_HD static constexpr uint32_t one(int i) {
    constexpr uint32_t m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                               0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return m[i];
  }
  // R^2 mod r
  ZG_HD static constexpr uint32_t r2(int i) {
    constexpr uint32_t m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                               0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    return m[i];
  }
  static constexpr uint32_t inv = 0xefffffffu;  // -r^-1 mod 2^32
};

struct FqParams {
  // q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
  ZG_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return m[i];
  }
  ZG_HD static constexpr uint32_t one(int i) {
    constexpr uint32_t m[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                               0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return m[i];
  }
  ZG_HD static constexpr uint32_t r2(int i) {
    constexpr uint32_t m[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                               0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    return m[i];
  }
  static constexpr uint32_t inv = 0xe4866389u;  // -q^-1 mod 2^32
};

template <class P>
struct alignas(16) Fp {
  uint32_t v[8];

  ZG_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
  }
  ZG_HD static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = P::one(i);
    return r;
  }
  ZG_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i];
    return o == 0;
  }
  ZG_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
  ZG_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = a - mod if a >= mod else a   (a < 2*mod < 2^256)
template <class P>
ZG_HD void reduce_once(uint32_t* a) {
  uint32_t d[8];
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a[i] - P::mod(i) - borrow;
    d[i] = (uint32_t)t;
    borrow = (t >> 32) & 1;
  }
  if (!borrow) {
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = d[i];
  }
}

template <class P>
ZG_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a.v[i] + b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  reduce_once<P>(r.v);  // a + b < 2*mod < 2^255: no carry out of limb 7
  return r;
}

template <class P>
ZG_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
    r.v[i] = (uint32_t)t;
    borrow = (t >> 32) & 1;
  }
  uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)r.v[i] + (P::mod(i) & mask);
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

template <class P>
ZG_HD Fp<P> neg(const Fp<P>& a) {
  return a.is_zero() ? a : sub(Fp<P>::zero(), a);
}

template <class P>
ZG_HD Fp<P> dbl(const Fp<P>& a) {
  return add(a, a);
}

#if defined(__HIP_DEVICE_COMPILE__)
// Montgomery product a*b*R^-1 mod p on the device: finely integrated product scanning.  Column k
// sums a_i*b_(k-i) and m_i*p_(k-i) into a 96-bit accumulator (acc, ex), derives m_k so that the
// column's low word vanishes, then shifts the accumulator down one word.  Every term is one
// v_mad_u64_u32 whose addend and result are the same register pair plus one add-with-carry: 136
// multiply-adds, 136 carry adds, three moves per column.  It is written as generated inline asm
// (tools/gen_field_mul.py) because the C form of a column sum makes hipcc re-pair 64-bit registers
// with two v_mov per term (gfx950 wants 64-bit VGPR operands even-aligned); modulus limbs ride in
// scalar registers.
template <class P>
ZG_D Fp<P> mul_inline(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
#include "field_mul_gfx950.inc"
  // the result is below 2p < 2^255, so nothing is left in the accumulator
  reduce_once<P>(r.v);
  return r;
}
#else
// Montgomery product a*b*R^-1 mod p on the host: CIOS over four 64-bit digits (the same bytes as the
// eight 32-bit limbs on a little-endian machine).  Because p < 2^254 the running value stays below
// 2p, so one spare digit is enough.  Used by the final assembly of A, B, C and by table set-up.
template <class P>
inline Fp<P> mul_inline(const Fp<P>& a, const Fp<P>& b) {
  typedef unsigned __int128 u128;
  uint64_t x[4], y[4], p[4], t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    x[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
    y[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
    p[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
  }
  // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: inv64 = inv32 * (2 + p * inv32)
  const uint64_t i32 = P::inv;
  const uint64_t inv64 = i32 * (2 + p[0] * i32);
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)x[j] * y[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * inv64;
    c = (u128)m * p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * p[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  uint32_t r32[8];
  for (int i = 0; i < 4; i++) {
    r32[2 * i] = (uint32_t)t[i];
    r32[2 * i + 1] = (uint32_t)(t[i] >> 32);
  }
  reduce_once<P>(r32);
  Fp<P> r;
  for (int i = 0; i < 8; i++) r.v[i] = r32[i];
  return r;
}
#endif

// Translation units whose kernels chain dozens of products per lane (curve arithmetic) define
// ZG_OUTLINE_FIELD_MUL: the product then is ONE out-of-line leaf function per field (operands and
// result travel in VGPRs, ~650 instructions, few registers), every group operation is inlined around
// calls to it, and kernels stay small enough for hipcc to allocate registers for 4 waves per SIMD.
// Inlining the product everywhere instead makes a G2 addition ~25k instructions and out-of-line
// group operations make every caller inherit their ~240 VGPRs (1-2 waves per SIMD).
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZG_OUTLINE_FIELD_MUL)
template <class P>
__device__ __noinline__ Fp<P> mul(const Fp<P> a, const Fp<P> b) {
  return mul_inline(a, b);
}
#else
template <class P>
ZG_HD Fp<P> mul(const Fp<P>& a, const Fp<P>& b) {
  return mul_inline(a, b);
}
#endif

template <class P>
ZG_HD Fp<P> sqr(const Fp<P>& a) {
  return mul(a, a);
}

// Montgomery form -> canonical integer (multiply by 1).
template <class P>
ZG_HD Fp<P> from_mont(const Fp<P>& a) {
  Fp<P> o = Fp<P>::zero();
  o.v[0] = 1;
  return mul(a, o);
}

template <class P>
ZG_HD Fp<P> to_mont(const Fp<P>& a) {
  Fp<P> r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.v[i] = P::r2(i);
  return mul(a, r2);
}

// a^e for a little-endian u32 exponent of `nlimbs` limbs (square and multiply, MSB first).
template <class P>
ZG_HD Fp<P> pow_limbs(const Fp<P>& a, const uint32_t* e, int nlimbs) {
  Fp<P> r = Fp<P>::one();
  bool started = false;
  for (int i = nlimbs - 1; i >= 0; i--) {
    for (int b = 31; b >= 0; b--) {
      if (started) r = sqr(r);
      if ((e[i] >> b) & 1) {
        r = started ? mul(r, a) : a;
        started = true;
      }
    }
  }
  return r;
}

template <class P>
ZG_HD bool is_canonical(const uint32_t* a);

#if !defined(__HIP_DEVICE_COMPILE__)
// Host only: a^-1 by the binary extended Euclidean algorithm on the Montgomery representation read as an integer
// (shifts, additions and subtractions of 256-bit integers: about a microsecond, against ~380 field products for
// Fermat's power).  The integer inverse of a*R is a^-1 * R^-1; one product by R^3 brings it to a^-1 * R.
namespace host_inv {
struct U256 {
  uint64_t w[4];
};
inline bool is_even(const U256& a) { return (a.w[0] & 1) == 0; }
inline bool is_one(const U256& a) { return a.w[0] == 1 && (a.w[1] | a.w[2] | a.w[3]) == 0; }
inline void shr1(U256& a) {
  for (int i = 0; i < 3; i++) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 63);
  a.w[3] >>= 1;
}
inline bool geq(const U256& a, const U256& b) {
  for (int i = 3; i >= 0; i--)
    if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
  return true;
}
inline void add_to(U256& a, const U256& b) {
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a.w[i] + b.w[i];
    a.w[i] = (uint64_t)c;
    c >>= 64;
  }
}
inline void sub_from(U256& a, const U256& b) {  // a >= b
  unsigned __int128 borrow = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned __int128 t = (unsigned __int128)a.w[i] - b.w[i] - borrow;
    a.w[i] = (uint64_t)t;
    borrow = (t >> 64) & 1;
  }
}
// x <- x / 2 mod p (p odd, x < p < 2^255)
inline void half_mod(U256& x, const U256& p) {
  if (!is_even(x)) add_to(x, p);
  shr1(x);
}
// x <- x - y mod p (x, y < p)
inline void sub_mod(U256& x, const U256& y, const U256& p) {
  if (!geq(x, y)) add_to(x, p);
  sub_from(x, y);
}
}  // namespace host_inv

template <class P>
inline Fp<P> inv_host(const Fp<P>& a) {
  using namespace host_inv;
  if (a.is_zero()) return a;  // as the power does
  // limbs that are not a residue below p (never produced by this library's arithmetic; a caller's raw bytes can
  // be): the loop below ends only for inputs coprime to p, and u = p is not -- answer as for zero
  if (!is_canonical<P>(a.v)) return Fp<P>::zero();
  U256 u, v, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
  for (int i = 0; i < 4; i++) {
    u.w[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
    v.w[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
  }
  const U256 p = v;
  while (!is_one(u) && !is_one(v)) {
    while (is_even(u)) {
      shr1(u);
      half_mod(x1, p);
    }
    while (is_even(v)) {
      shr1(v);
      half_mod(x2, p);
    }
    if (geq(u, v)) {
      sub_from(u, v);
      sub_mod(x1, x2, p);
    } else {
      sub_from(v, u);
      sub_mod(x2, x1, p);
    }
  }
  const U256& t = is_one(u) ? x1 : x2;
  Fp<P> ti, r2;
  for (int i = 0; i < 4; i++) {
    ti.v[2 * i] = (uint32_t)t.w[i];
    ti.v[2 * i + 1] = (uint32_t)(t.w[i] >> 32);
  }
  for (int i = 0; i < 8; i++) r2.v[i] = P::r2(i);
  return mul(ti, mul(r2, r2));  // t * R^3 * R^-1 = a^-1 * R
}
#endif

// a^-1 (a != 0; zero gives zero): the binary Euclidean algorithm on the host, Fermat's power on the device (used
// there for one-off per-launch scalars and the table builders only).
template <class P>
ZG_HD Fp<P> inv(const Fp<P>& a) {
#if !defined(__HIP_DEVICE_COMPILE__)
  return inv_host(a);
#else
  uint32_t e[8];
  uint64_t borrow = 2;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)P::mod(i) - borrow;
    e[i] = (uint32_t)t;
    borrow = (t >> 32) & 1;
  }
  return pow_limbs(a, e, 8);
#endif
}

template <class P>
ZG_HD Fp<P> pow_u64(const Fp<P>& a, uint64_t e) {
  uint32_t l[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
  return pow_limbs(a, l, 2);
}

template <class P>
ZG_HD Fp<P> from_u64(uint64_t x) {
  Fp<P> r = Fp<P>::zero();
  r.v[0] = (uint32_t)x;
  r.v[1] = (uint32_t)(x >> 32);
  return to_mont(r);
}

// canonical comparison helper: is a (any residue representative < 2^256) below the modulus?
template <class P>
ZG_HD bool is_canonical(const uint32_t* a) {
  for (int i = 7; i >= 0; i--) {
    if (a[i] < P::mod(i)) return true;
    if (a[i] > P::mod(i)) return false;
  }
  return false;
}

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1)
struct Fq2 {
  Fq c0, c1;
  ZG_HD static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
  ZG_HD static Fq2 one() { return {Fq::one(), Fq::zero()}; }
  ZG_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZG_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  ZG_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
};

ZG_HD Fq2 add(const Fq2& a, const Fq2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
ZG_HD Fq2 sub(const Fq2& a, const Fq2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
ZG_HD Fq2 neg(const Fq2& a) { return {neg(a.c0), neg(a.c1)}; }
ZG_HD Fq2 dbl(const Fq2& a) { return {dbl(a.c0), dbl(a.c1)}; }
// Karatsuba: 3 base-field products
ZG_HD Fq2 mul(const Fq2& a, const Fq2& b) {
  Fq aa = mul(a.c0, b.c0);
  Fq bb = mul(a.c1, b.c1);
  Fq s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
  return {sub(aa, bb), sub(sub(s, aa), bb)};
}
// (c0 + c1 u)^2 = (c0+c1)(c0-c1) + 2 c0 c1 u : 2 base-field products
ZG_HD Fq2 sqr(const Fq2& a) {
  Fq p = mul(add(a.c0, a.c1), sub(a.c0, a.c1));
  Fq m = mul(a.c0, a.c1);
  return {p, dbl(m)};
}
ZG_HD Fq2 inv(const Fq2& a) {
  Fq n = inv(add(sqr(a.c0), sqr(a.c1)));
  return {mul(a.c0, n), neg(mul(a.c1, n))};
}

}  // namespace zg
