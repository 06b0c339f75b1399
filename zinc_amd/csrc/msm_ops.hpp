// Point arithmetic as the multiexp kernels see it: one traits class per curve hides which field
// representation the accumulators use and how a point is laid out in memory.
//
//   Pt      an accumulator as one lane holds it in registers
//   PtMem   the same in device memory (bucket sums, partial sums); all-zero bytes = identity
//   Aff     an affine base as one lane holds it
//   LP      lanes that cooperate on one point (kernels index logical threads as lane / LP)
//
// The spread form, for the dependent additions that end a multiexp (see quad_add): a point lies over SL adjacent
// lanes and every lane keeps its part (Own) only.  own_* are the operations on it; every lane of a group of SL
// lanes must take part in each of them.
//
// Bases are read from the "internal packed" form the key tables are stored in (to_internal).
#pragma once
#include "curve29.hpp"

namespace zg {

template <class F>
struct MsmOps;

// a word as another workgroup of this launch last wrote it (after its fence), not a copy that this compute unit's
// vector cache may still hold
ZG_D uint32_t load_fresh(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// G1: 29-bit-limb field (fp29.hpp), one lane per point.  Packed bases are in radix-2^261
// Montgomery form, i.e. the ABI's radix-2^256 value times 32.
template <>
struct MsmOps<Fq> {
  using Pt = G1XYZZ29;
  using PtMem = G1XYZZ29;
  using Aff = G1Affine29;
  static constexpr int LP = 1;
  static constexpr int MIN_WAVES = 4;

  ZG_D static Pt inf() { return Pt::inf(); }
  ZG_D static bool is_inf(const Pt& p) { return p.is_inf(); }
  // false when the stored base is the point at infinity
  ZG_D static bool load_base(const Affine<Fq>* src, bool negate, Aff* out) {
    const Affine<Fq> a = *src;
    if (a.is_inf()) return false;
    out->x = unpack29<Fq29Params>(a.x);
    const Fq29 y = unpack29<Fq29Params>(a.y);
    Fq29 ny;
#pragma unroll
    for (int i = 0; i < 9; i++) ny.v[i] = negate ? Fq29Params::K3[i] - y.v[i] : y.v[i];
    out->y = norm(ny);  // y or 3p - y
    return true;
  }
  ZG_D static void madd(Pt& acc, const Aff& b) { add_mixed29(acc, b); }
  ZG_D static void add(Pt& acc, const Pt& o) { add_full29(acc, o); }
  ZG_D static Pt dbl(const Pt& p) { return dbl29(p); }
  ZG_D static Pt load(const PtMem* p) { return *p; }
  ZG_D static void store(PtMem* p, const Pt& v) { *p = v; }
  ZG_D static Pt shfl_down(const Pt& p, int d) {
    Pt o;
    const uint32_t* src = (const uint32_t*)&p;
    uint32_t* dst = (uint32_t*)&o;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Pt) / 4); i++) dst[i] = __shfl_down(src[i], d);
    return o;
  }
  // logical thread 0's copy of p, in every lane of the wave
  ZG_D static Pt first_thread(const Pt& p) {
    Pt o;
    const uint32_t* src = (const uint32_t*)&p;
    uint32_t* dst = (uint32_t*)&o;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Pt) / 4); i++) dst[i] = __shfl(src[i], 0);
    return o;
  }
  // the library-wide XYZZ form (radix-2^256 Montgomery, canonical) for the host
  ZG_D static void store_std(XYZZ<Fq>* dst, const Pt& p) { *dst = export_xyzz<Fq29Params, FqParams>(p); }
  // ABI form -> internal packed form
  ZG_HD static Affine<Fq> to_internal(Affine<Fq> a) {
    for (int k = 0; k < 5; k++) {
      a.x = zg::dbl(a.x);
      a.y = zg::dbl(a.y);
    }
    return a;
  }

  // spread form: lane q of a quad holds coordinate q of (X, Y, ZZ, ZZZ)
  static constexpr int SL = 4;
  using Own = Fq29;
  ZG_D static Own own_inf() { return Fq29::zero(); }
  ZG_D static Own own_load(const PtMem* p) { return ((const Fq29*)p)[threadIdx.x & 3]; }
  ZG_D static Own own_load_fresh(const PtMem* p) {
    Own r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = load_fresh(((const Fq29*)p)[threadIdx.x & 3].v + i);
    return r;
  }
  ZG_D static void own_store(PtMem* p, const Own& v) { ((Fq29*)p)[threadIdx.x & 3] = v; }
  ZG_D static Own own_of(const Pt& p) { return quad_own(p); }  // p whole in every lane of the quad
  ZG_D static void own_add(Own& a, const Own& b) { quad_add(a, b); }
  // a += b in the groups where `active` holds; every lane of the wave walks the addition either way
  ZG_D static void own_add_if(Own& a, const Own& b, bool active) { quad_add(a, pick(active, b, Fq29::zero())); }
  ZG_D static Own own_shfl_down(const Own& v, int lanes) {
    Own o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.v[i] = __shfl_down(v.v[i], lanes);
    return o;
  }
  ZG_D static void own_store_std(XYZZ<Fq>* dst, const Own& v) {
    const bool inf = quad_is_inf(v);
    ((Fq*)dst)[threadIdx.x & 3] = inf ? Fq::zero() : to_r256<Fq29Params, FqParams>(v);
  }
};

// G2: the same 29-bit-limb field with every Fq2 value spread over a lane pair (even lane: real
// components, odd lane: imaginary ones), so a G2 lane has the register footprint of a G1 lane.
// Packed bases are x.c0 | x.c1 | y.c0 | y.c1 in radix-2^261 Montgomery form.
template <>
struct MsmOps<Fq2> {
  using Pt = G1XYZZ29;  // one lane's components
  struct PtMem {
    Pt half[2];
  };
  using Aff = G1Affine29;
  static constexpr int LP = 2;
  static constexpr int MIN_WAVES = 3;

  ZG_D static int sub() { return threadIdx.x & 1; }
  ZG_D static Pt inf() { return Pt::inf(); }
  ZG_D static bool is_inf(const Pt& p) { return pair_is_inf(p); }
  ZG_D static bool load_base(const Affine<Fq2>* src, bool negate, Aff* out) {
    const Fq* c = (const Fq*)src;
    const Fq x = c[sub()];
    Fq y = c[2 + sub()];
    if (pair_both(x.is_zero() && y.is_zero())) return false;
    if (negate) y = neg(y);
    out->x = unpack29<Fq29Params>(x);
    out->y = unpack29<Fq29Params>(y);
    return true;
  }
  ZG_D static void madd(Pt& acc, const Aff& b) { add_mixed_pair(acc, b); }
  ZG_D static void add(Pt& acc, const Pt& o) { add_full_pair(acc, o); }
  ZG_D static Pt dbl(const Pt& p) { return dbl_pair(p); }
  ZG_D static Pt load(const PtMem* p) { return p->half[sub()]; }
  ZG_D static void store(PtMem* p, const Pt& v) { p->half[sub()] = v; }
  ZG_D static Pt shfl_down(const Pt& p, int d) {
    Pt o;
    const uint32_t* src = (const uint32_t*)&p;
    uint32_t* dst = (uint32_t*)&o;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Pt) / 4); i++) dst[i] = __shfl_down(src[i], d);
    return o;
  }
  ZG_D static Pt first_thread(const Pt& p) {  // lanes 0 and 1 hold logical thread 0's two components
    Pt o;
    const uint32_t* src = (const uint32_t*)&p;
    uint32_t* dst = (uint32_t*)&o;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Pt) / 4); i++) dst[i] = __shfl(src[i], (int)sub());
    return o;
  }
  ZG_D static void store_std(XYZZ<Fq2>* dst, const Pt& p) {
    Fq* c = (Fq*)dst;
    const bool inf = pair_is_inf(p);
    c[0 + sub()] = inf ? Fq::zero() : to_r256<Fq29Params, FqParams>(p.x);
    c[2 + sub()] = inf ? Fq::zero() : to_r256<Fq29Params, FqParams>(p.y);
    c[4 + sub()] = inf ? Fq::zero() : to_r256<Fq29Params, FqParams>(p.zz);
    c[6 + sub()] = inf ? Fq::zero() : to_r256<Fq29Params, FqParams>(p.zzz);
  }
  // spread form: lane 2 pos + comp of an octet holds component comp of the coordinate at position pos (oct_add)
  static constexpr int SL = 8;
  using Own = Fq29;
  ZG_D static Own own_inf() { return Fq29::zero(); }
  ZG_D static const Fq29* own_slot(const PtMem* p) { return (const Fq29*)&p->half[sub()] + oct_coord(oct_pos()); }
  ZG_D static Own own_load(const PtMem* p) { return *own_slot(p); }
  ZG_D static Own own_load_fresh(const PtMem* p) {
    Own r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = load_fresh(own_slot(p)->v + i);
    return r;
  }
  ZG_D static void own_store(PtMem* p, const Own& v) { *(Fq29*)own_slot(p) = v; }
  ZG_D static Own own_of(const Pt& p) { return oct_own(p); }  // p whole in every lane pair of the octet
  ZG_D static void own_add(Own& a, const Own& b) { oct_add(a, b); }
  ZG_D static void own_add_if(Own& a, const Own& b, bool active) { oct_add(a, pick(active, b, Fq29::zero())); }
  ZG_D static Own own_shfl_down(const Own& v, int lanes) {
    Own o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.v[i] = __shfl_down(v.v[i], lanes);
    return o;
  }
  ZG_D static void own_store_std(XYZZ<Fq2>* dst, const Own& v) {
    const bool inf = oct_is_inf(v);
    ((Fq*)dst)[2 * oct_coord(oct_pos()) + sub()] = inf ? Fq::zero() : to_r256<Fq29Params, FqParams>(v);
  }
  ZG_HD static Affine<Fq2> to_internal(Affine<Fq2> a) {
    for (int k = 0; k < 5; k++) {
      a.x.c0 = zg::dbl(a.x.c0);
      a.x.c1 = zg::dbl(a.x.c1);
      a.y.c0 = zg::dbl(a.y.c0);
      a.y.c1 = zg::dbl(a.y.c1);
    }
    return a;
  }
};

}  // namespace zg
