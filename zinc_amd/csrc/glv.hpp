// k1 * P + k2 * Q on BN254 G1 with the curve's endomorphism (host code: the proof's final assembly).
//
// y^2 = x^3 + 3 over Fq with q = 1 mod 3 has phi(x, y) = (beta x, y), beta a cube root of unity in Fq, and on the
// group of order r phi is multiplication by lambda, a cube root of unity in Fr.  Every scalar k < 2^256 splits as
// k = u + v lambda (mod r) with |u|, |v| < 2^128: (u, v) = (k, 0) - c1 v1 - c2 v2 for the reduced basis v1, v2 of
// the lattice {(a, b): a + b lambda = 0 mod r} and c_i the nearest integers to k b2 / r and -k b1 / r (taken from
// 2^256-scaled quotients; an error of one in c_i only makes u, v a little longer).  k1 P + k2 Q is then a sum of
// four points with 128-bit scalars: 128 doublings and at most 128 additions from a table of the 15 subset sums,
// against 254 doublings and ~120 additions for two bits of two full-length scalars at a time.  The result is the
// same group element; whatever u, v come out, u + v lambda = k holds exactly, and a split that came out longer than
// 128 bits (it cannot, by the bound above) falls back to the plain routine.
#pragma once
#include "curve.hpp"

namespace zg {
// (host code only; the device pass of a translation unit sees the declaration)
G1XYZZ double_scalar_mul_glv(const G1XYZZ& p, const uint32_t* k1, const G1XYZZ& q, const uint32_t* k2);
}  // namespace zg

#if !defined(__HIP_DEVICE_COMPILE__)
namespace zg {
namespace glv {

// lambda = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd, beta = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
// v1 = (A1, -NB1), v2 = (A2, B2); det = r.  G1C = floor(2^256 B2 / r), G2C = floor(2^256 NB1 / r).
static const uint64_t A1[1] = {0x89d3256894d213e3ull};
static const uint64_t NB1[2] = {0x8211bbeb7d4f1128ull, 0x6f4d8248eeb859fcull};
static const uint64_t A2[2] = {0x0be4e1541221250bull, 0x6f4d8248eeb859fdull};
static const uint64_t B2[1] = {0x89d3256894d213e3ull};
static const uint64_t G1C[2] = {0xd91d232ec7e0b3d7ull, 0x0000000000000002ull};
static const uint64_t G2C[3] = {0x7a7bd9d4391eb18dull, 0x4ccef014a773d2cfull, 0x0000000000000002ull};
static const uint64_t BETA[4] = {0x5763473177fffffeull, 0xd4f263f1acdb5c4full, 0x59e26bcea0d48bacull, 0};

// out[0 .. na + nb) = a * b
inline void umul(const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* out) {
  for (int i = 0; i < na + nb; i++) out[i] = 0;
  for (int i = 0; i < na; i++) {
    unsigned __int128 c = 0;
    for (int j = 0; j < nb; j++) {
      c += (unsigned __int128)a[i] * b[j] + out[i + j];
      out[i + j] = (uint64_t)c;
      c >>= 64;
    }
    out[i + nb] = (uint64_t)c;
  }
}

// five-digit two's complement: x += sign * y (y: n digits, unsigned)
inline void acc5(uint64_t* x, const uint64_t* y, int n, bool subtract) {
  unsigned __int128 c = subtract ? 1 : 0;
  for (int i = 0; i < 5; i++) {
    const uint64_t yi = i < n ? y[i] : 0;
    c += (unsigned __int128)x[i] + (subtract ? ~yi : yi);
    x[i] = (uint64_t)c;
    c >>= 64;
  }
}

struct Half {
  uint64_t mag[2];
  bool neg;
};

// x (five-digit two's complement) as sign and 128-bit magnitude; false when it does not fit
inline bool to_half(const uint64_t* x, Half* h) {
  uint64_t m[5];
  h->neg = (x[4] >> 63) != 0;
  if (h->neg) {
    unsigned __int128 c = 1;
    for (int i = 0; i < 5; i++) {
      c += (uint64_t)~x[i];
      m[i] = (uint64_t)c;
      c >>= 64;
    }
  } else {
    for (int i = 0; i < 5; i++) m[i] = x[i];
  }
  h->mag[0] = m[0];
  h->mag[1] = m[1];
  return (m[2] | m[3] | m[4]) == 0;
}

// k (canonical, 8 u32 limbs, any value below 2^256) = u + v lambda mod r
inline bool split(const uint32_t* k32, Half* u, Half* v) {
  uint64_t k[4];
  for (int i = 0; i < 4; i++) k[i] = (uint64_t)k32[2 * i] | ((uint64_t)k32[2 * i + 1] << 32);
  // c1 = (k G1C + 2^255) >> 256, c2 = (k G2C + 2^255) >> 256
  uint64_t t1[6], t2[7], c1[2], c2[3];
  umul(k, 4, G1C, 2, t1);
  umul(k, 4, G2C, 3, t2);
  {
    unsigned __int128 c = (unsigned __int128)t1[3] + (1ull << 63);
    c >>= 64;
    for (int i = 0; i < 2; i++) {
      c += t1[4 + i];
      c1[i] = (uint64_t)c;
      c >>= 64;
    }
    c = (unsigned __int128)t2[3] + (1ull << 63);
    c >>= 64;
    for (int i = 0; i < 3; i++) {
      c += t2[4 + i];
      c2[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  // u = k - c1 A1 - c2 A2,  v = c1 NB1 - c2 B2
  uint64_t p[5], x[5] = {k[0], k[1], k[2], k[3], 0}, y[5] = {0, 0, 0, 0, 0};
  umul(c1, 2, A1, 1, p);
  acc5(x, p, 3, true);
  umul(c2, 3, A2, 2, p);
  acc5(x, p, 5, true);
  umul(c1, 2, NB1, 2, p);
  acc5(y, p, 4, false);
  umul(c2, 3, B2, 1, p);
  acc5(y, p, 4, true);
  return to_half(x, u) && to_half(y, v);
}

inline G1XYZZ endo(const G1XYZZ& p) {
  static const Fq beta_mont = [] {
    Fq b;
    for (int i = 0; i < 4; i++) {
      b.v[2 * i] = (uint32_t)BETA[i];
      b.v[2 * i + 1] = (uint32_t)(BETA[i] >> 32);
    }
    return to_mont(b);
  }();
  G1XYZZ r = p;
  if (!r.is_inf()) r.x = mul(r.x, beta_mont);
  return r;
}

inline G1XYZZ negate_if(const G1XYZZ& p, bool n) {
  G1XYZZ r = p;
  if (n && !r.is_inf()) r.y = neg(r.y);
  return r;
}

}  // namespace glv

// k1 * p + k2 * q for canonical little-endian scalars of 8 u32 limbs (the same group element as
// double_scalar_mul(p, k1, q, k2))
inline G1XYZZ double_scalar_mul_glv(const G1XYZZ& p, const uint32_t* k1, const G1XYZZ& q, const uint32_t* k2) {
  glv::Half h[4];
  if (!glv::split(k1, &h[0], &h[1]) || !glv::split(k2, &h[2], &h[3])) return double_scalar_mul(p, k1, q, k2);
  G1XYZZ tab[16];
  tab[0] = G1XYZZ::inf();
  tab[1] = glv::negate_if(p, h[0].neg);
  tab[2] = glv::negate_if(glv::endo(p), h[1].neg);
  tab[4] = glv::negate_if(q, h[2].neg);
  tab[8] = glv::negate_if(glv::endo(q), h[3].neg);
  for (int m = 3; m < 16; m++) {
    if ((m & (m - 1)) == 0) continue;
    tab[m] = tab[m & (m - 1)];
    add_full(tab[m], tab[m & -m]);
  }
  G1XYZZ acc = G1XYZZ::inf();
  for (int bit = 127; bit >= 0; bit--) {
    if (!acc.is_inf()) acc = dbl(acc);
    int m = 0;
    for (int i = 0; i < 4; i++) m |= (int)((h[i].mag[bit >> 6] >> (bit & 63)) & 1) << i;
    if (m) add_full(acc, tab[m]);
  }
  return acc;
}

}  // namespace zg
#endif
