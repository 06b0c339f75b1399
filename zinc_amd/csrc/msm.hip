// Bucket multiexp over BN254 G1 / G2.
//
// Replaces bellman_ce 0.3.0 `multiexp` (Pippenger over a CPU pool) as used by `create_proof` for
// the h, l, a, b_g1 and b_g2 queries (reached from zinc-vm/src/core/circuit/facade.rs:153).  The
// sum of scalar*base terms is an exact group element, so the signed-digit windows, bucket order,
// precomputed tables and coordinate system used here give the same affine point as upstream's
// unsigned per-window buckets.
//
// Two bucket layouts:
//  * shared buckets (proving-key queries): the key holds 2^(c*w) * P for every window w
//    (built once at upload by k_table_windows / k_table_affine -- 288 GB of HBM makes the W-fold copy cheap), so a
//    digit d of window w is simply "add table[w][i] to bucket |d|".  All W windows share ONE set of
//    2^(c-1) buckets: the bucket reduction shrinks W-fold and there is no window recombination.
//  * per-window buckets (stand-alone multiexp on caller-supplied bases): W bucket sets, host Horner.
//
// Pipeline (one stream, no host synchronisation until the result is read):
//   1. k_sort_* / k_fine_* : Montgomery scalar -> canonical -> W signed c-bit digits -> two-level
//                          counting sort of the (bucket, window, base index, sign) entries
//   2. k_scan_*          : exclusive scan of bucket sizes -> bucket offsets
//   3. k_accumulate      : segmented reduction over the sorted entries.  Every lane owns a fixed
//                          chunk of consecutive entries whatever the bucket sizes are, so a
//                          skewed witness (zinc witnesses are mostly 0/1 bits) cannot unbalance
//                          the launch.  A run inside one chunk finishes its bucket; a run cut by a
//                          chunk edge leaves a partial sum in slot 2t / 2t+1.
//   4. k_combine         : the pieces of a bucket cut at chunk edges (at most 32 of them) are added by four lanes
//                          (G2: eight) that hold one point between them; larger buckets get one or more
//                          workgroups each in k_heavy.
//   5. k_sum_range       : plain tree sum for scalars equal to one (about half of a zinc witness);
//                          they reach their bucket through a wave-aggregated counter, not a digit.
//   6. k_split_sums x2   : each bucket set folded to a few dozen plain sums (no scalar multiplications)
//      Steps 4-6 are chains of dependent additions on few points: they run in the spread form of quad_add /
//      oct_add (a point over 4 / 8 lanes, the products of an addition four at a time).
//   7. host              : sum_b (b+1)*S_b from the folded sums; Horner over the W window sums
//                          (per-window layout only)
#define ZG_OUTLINE_FIELD_MUL 1
#include <string.h>

#include <algorithm>

#include "r1cs.hpp"
#include "msm_ops.hpp"

namespace zg {

static constexpr uint32_t ACC_KMIN = 8;   // fewest entries a lane of k_accumulate takes
static constexpr uint32_t LIGHT_PIECES = 32;  // buckets cut into at most this many pieces: k_combine, more: k_heavy
static constexpr int MAX_JOBS = 4;
static constexpr uint32_t KEY_INVALID = 0xffffffffu;
static constexpr uint32_t KEY_MASK = 0x00ffffffu;  // level-0 keys carry the window in the top byte

// Minimum waves per SIMD requested from the register allocator for the curve kernels.
template <class F>
constexpr int kMinWaves = MsmOps<F>::MIN_WAVES;
// The two launches of a direct-table multiexp (a small circuit's proof) are chains of dependent group additions on
// few waves with nothing else on the GPU: compiled for two waves per SIMD (256 registers) they do not spill.  The
// latency-bound kernels that END a bucket multiexp (k_combine, k_heavy, k_split_sums, k_sum_range) spill 40-144 bytes
// per lane at the four-wave register cut too, but they run beside other pipelines' accumulations, and waves twice as
// wide there take the registers of two accumulating waves: measured, a 2^20-row proof went from 4.2 to 4.45 ms.
constexpr int kTailWaves = 2;

template <class F>
struct JobTable {
  const Affine<F>* bases[MAX_JOBS];
  const Fr* scalars[MAX_JOBS];
  const uint32_t* index[MAX_JOBS];
  uint32_t stride[MAX_JOBS];     // table stride (points per window); unused with per-window buckets
  uint32_t start[MAX_JOBS + 1];  // prefix offsets of the concatenated term list
  uint32_t group[MAX_JOBS];      // which result (bucket set) the job's terms are summed into
  const Affine<F>* direct[MAX_JOBS];  // direct tables (k_direct_sum), stride `stride`
  uint32_t windows;                   // windows a table holds (1 when the buckets are per-window)
  int njobs;
};

struct MsmShape {
  uint32_t n;        // total terms
  uint32_t c;        // window bits
  uint32_t W;        // windows
  uint32_t B;        // buckets per bucket set = 2^(c-1)
  uint32_t nb;       // digit buckets (G*B shared, W*B per-window); bucket nb + g collects group g's unit scalars
  uint32_t shared;   // 1: the windows of a group share `f` bucket sets (table lookup), 0: one set per window
  uint32_t G;        // groups = independent results (1 unless shared)
  uint32_t f;        // shared only: the tables keep every f-th window; window w uses table window w / f and set w mod f
};

template <class F>
ZG_D int find_job(const JobTable<F>& jt, uint32_t i) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < MAX_JOBS; k++)
    if (k < jt.njobs && i >= jt.start[k]) j = k;
  return j;
}

template <class F>
ZG_D Fr load_scalar(const JobTable<F>& jt, uint32_t i, int j) {
  const uint32_t local = i - jt.start[j];
  const uint32_t si = jt.index[j] ? jt.index[j][local] : local;
  return jt.scalars[j][si];
}

// address of the base of term i for window w (w = 0 when the buckets are per-window).  The job's fields are picked
// by a chain of selects over constant indices: indexing the argument arrays with a per-lane job number makes the
// compiler read them back from memory, one more dependent load in front of every base.
template <class F>
ZG_D const Affine<F>* base_ptr(const JobTable<F>& jt, uint32_t i, uint32_t w) {
  const Affine<F>* b = jt.bases[0];
  uint32_t stride = jt.stride[0], start = jt.start[0], end = jt.start[1];
#pragma unroll
  for (int k = 1; k < MAX_JOBS; k++) {
    const bool in = k < jt.njobs && i >= jt.start[k];
    b = in ? jt.bases[k] : b;
    stride = in ? jt.stride[k] : stride;
    start = in ? jt.start[k] : start;
    end = in ? jt.start[k + 1] : end;
  }
  // (an index beyond the term list or the table can only come from a damaged sort entry; it is clamped so that such
  // a bug reads a wrong base instead of faulting the GPU)
  const uint32_t local = i - start, last = end - 1 - start;
  return b + ((size_t)(w < jt.windows ? w : 0) * stride + (local < last ? local : last));
}

// bits [lo, lo+c) of a 256-bit little-endian integer (c <= 24)
ZG_D uint32_t extract_bits(const uint32_t* v, uint32_t lo, uint32_t c) {
  const uint32_t word = lo >> 5, sh = lo & 31;
  uint64_t x = v[word];
  if (word + 1 < 8) x |= (uint64_t)v[word + 1] << 32;
  return (uint32_t)(x >> sh) & ((1u << c) - 1);
}

// ---- bucket sort of the (term, window) entries --------------------------------------------------
// An entry is 8 bytes: x = bucket | window << 24, y = term index | sign << 31.  Order inside a
// bucket is irrelevant (group addition is exact), so this is a two-level counting sort whose global
// atomics are per (workgroup, bin) instead of per entry:
//   k_sort_count    terms -> digits -> LDS histogram of coarse bins (bucket >> shift, at most
//                   SORT_MAX_COARSE of them) -> the workgroup's row of the (bin, workgroup) histogram
//                   (the coarse bin starts are read off the scanned per-workgroup histograms by the first
//                   workgroup of the next kernel)
//   k_sort_scatter  same digits again; a scan of the per-workgroup histograms k_sort_count left behind gives
//                   the workgroup its run in every coarse bin, and it writes its entries there (ranks by LDS
//                   atomics, no global atomic at all).  Unit scalars go straight to their final
//                   place behind all digit entries.
//   k_fine_count / k_fine_scatter   tiles of the coarse-sorted entries span a few coarse bins, so
//                   their buckets are counted / ranked in LDS and again cost one global atomic per
//                   (tile, bucket).  A tile that spans more buckets than LDS holds (tiny multiexps)
//                   falls back to one atomic per entry.
static constexpr uint32_t SORT_TERMS_SMALL = 4;     // terms per lane in k_sort_count / k_sort_scatter ...
static constexpr uint32_t SORT_TERMS_LARGE = 16;    // ... and from 2^22 terms on: a workgroup's run in a coarse bin is
                                                    // then several 128-byte lines long, not a fraction of one
static constexpr uint32_t SORT_MAX_COARSE = 1024;
static constexpr uint32_t FINE_TILE = 4096;         // entries per workgroup in k_fine_*
static constexpr uint32_t FINE_PER_LANE = FINE_TILE / 256;
static constexpr uint32_t FINE_LDS = 8192;          // LDS counters in k_fine_*

struct SortShape {
  uint32_t shift;    // coarse bin = bucket >> shift
  uint32_t ncoarse;  // coarse bins covering the digit buckets; bins ncoarse + g count group g's unit scalars
  uint32_t terms;    // terms per lane of k_sort_count / k_sort_scatter
};

// Term i's scalar: 0 when it is zero (the term produces nothing), 1 when it is one (the term goes to its group's
// unit-scalar bucket), else 2 with the canonical value in *s.  *job: the job the term belongs to.
template <class F>
ZG_D int term_scalar(const JobTable<F>& jt, uint32_t i, Fr* s, int* job) {
  const int j = find_job(jt, i);
  *job = j;
  // zero and one -- four fifths of a zinc witness -- are recognised in Montgomery form, before the product that
  // takes a scalar out of it
  const Fr sm = load_scalar(jt, i, j);
  uint32_t nz = 0, n1 = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    nz |= sm.v[k];
    n1 |= sm.v[k] ^ FrParams::one(k);
  }
  if (nz == 0) return 0;
  if (n1 == 0) return 1;
  *s = from_mont(sm);
  return 2;
}

// Calls digit(bucket, window, negative) for every non-zero signed digit of the canonical scalar s of a term of
// group `group`.
template <class FD>
ZG_D void scalar_digits(const MsmShape& sh, uint32_t group, const Fr& s, FD digit) {
  const uint32_t set_base = sh.shared ? group * sh.f * sh.B : 0;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < sh.W; w++) {
    uint32_t d = extract_bits(s.v, w * sh.c, sh.c) + carry;
    uint32_t neg = 0;
    carry = 0;
    if (d > sh.B) {
      d = (1u << sh.c) - d;
      neg = 1;
      carry = 1;
    }
    if (d) {
      if (sh.shared)
        digit(set_base + (w % sh.f) * sh.B + d - 1, w / sh.f, neg);
      else
        digit(w * sh.B + d - 1, w, neg);
    }
  }
}

// Calls digit(bucket, window, negative) for every non-zero signed digit of term i and one(group) if
// the scalar equals one.  Scalars equal to zero produce nothing.
template <class F, class FD, class FO>
ZG_D void term_digits(const JobTable<F>& jt, const MsmShape& sh, uint32_t i, FD digit, FO one) {
  Fr s;
  int j;
  const int kind = term_scalar(jt, i, &s, &j);
  if (kind == 1) one(jt.group[j]);
  if (kind == 2) scalar_digits(sh, jt.group[j], s, digit);
}

template <class F>
__global__ __launch_bounds__(256) void k_sort_count(JobTable<F> jt, MsmShape sh, SortShape ss,
                                                    uint32_t* __restrict__ wg_hist) {
  const uint32_t SORT_TERMS = ss.terms;
  __shared__ uint32_t lh[SORT_MAX_COARSE + MAX_JOBS];
  const uint32_t tid = threadIdx.x, nbins = ss.ncoarse + sh.G;
  for (uint32_t j = tid; j < nbins; j += 256) lh[j] = 0;
  __syncthreads();
  const uint32_t tile = blockIdx.x * (256 * SORT_TERMS);
  for (uint32_t k = 0; k < SORT_TERMS; k++) {
    const uint32_t i = tile + k * 256 + tid;
    if (i < sh.n)
      term_digits(jt, sh, i, [&](uint32_t b, uint32_t, uint32_t) { atomicAdd(&lh[b >> ss.shift], 1u); },
                  [&](uint32_t g) { atomicAdd(&lh[ss.ncoarse + g], 1u); });
  }
  __syncthreads();
  // bin-major, so that one exclusive scan over the whole array gives every (bin, workgroup) its run's start
  for (uint32_t j = tid; j < nbins; j += 256) {
    wg_hist[(size_t)j * gridDim.x + blockIdx.x] = lh[j];
  }
}

// The coarse offsets fall out of the scanned per-workgroup histograms: bin j starts where its first workgroup's run
// starts.  coff[nbins] = total, and hist[nb + g] = number of unit scalars of group g so that the bucket scan places
// them too.  (A global histogram kept by atomics beside the per-workgroup ones cost k_sort_count half its time: a
// thousand workgroups adding to the same few hundred counters.)  Done by the first workgroup of k_sort_scatter on its
// way in -- a launch of its own was one more link in a chain of launches.
ZG_D void coarse_offsets(const uint32_t* __restrict__ wg_off, uint32_t sgrid, uint32_t n_wgh, uint32_t ncoarse, uint32_t G,
                         uint32_t* __restrict__ coff, uint32_t* __restrict__ hist, uint32_t nb) {
  const uint32_t nbins = ncoarse + G;
  for (uint32_t j = threadIdx.x; j <= nbins; j += 256) coff[j] = j < nbins ? wg_off[(size_t)j * sgrid] : wg_off[n_wgh];
  if (threadIdx.x < G) {
    const uint32_t j = ncoarse + threadIdx.x;
    const uint32_t end = j + 1 < nbins ? wg_off[(size_t)(j + 1) * sgrid] : wg_off[n_wgh];
    hist[nb + threadIdx.x] = end - wg_off[(size_t)j * sgrid];
  }
}

// The coarse scatter: every lane stores its entries straight to their bins' runs (ranks by LDS atomics).
// (Measured and dropped: ordering a chunk's entries by bin in LDS first -- count, scan, place, then copy out by
// consecutive lanes -- so that runs are written as whole lines.  It does bring the bytes written down from 2.0x the
// entry bytes, and it is half as fast again: 0.97 against 0.66 ms of sort per 2^20-row proof, 11.0 against 7.7 ms at
// 2^24.  The scatter is bound by its digit arithmetic, LDS atomics and barriers, not by HBM writes.)
template <class F>
__global__ __launch_bounds__(256) void k_sort_scatter(JobTable<F> jt, MsmShape sh, SortShape ss,
                                                             const uint32_t* __restrict__ wg_off, uint32_t n_wgh,
                                                             uint2* __restrict__ rec_coarse, uint2* __restrict__ rec_final,
                                                             uint32_t* __restrict__ coff, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[SORT_MAX_COARSE + MAX_JOBS];     // cursor inside this workgroup's run of each bin
  __shared__ uint32_t lbase[SORT_MAX_COARSE + MAX_JOBS];  // start of that run (scanned k_sort_count histogram)
  const uint32_t tid = threadIdx.x, nbins = ss.ncoarse + sh.G;
  if (blockIdx.x == 0) coarse_offsets(wg_off, gridDim.x, n_wgh, ss.ncoarse, sh.G, coff, hist, sh.nb);
  for (uint32_t j = tid; j < nbins; j += 256) {
    lh[j] = 0;
    lbase[j] = wg_off[(size_t)j * gridDim.x + blockIdx.x];
  }
  __syncthreads();
  const uint32_t tile = blockIdx.x * (256 * ss.terms);
  for (uint32_t k = 0; k < ss.terms; k++) {
    const uint32_t i = tile + k * 256 + tid;
    if (i < sh.n)
      term_digits(
          jt, sh, i,
          [&](uint32_t b, uint32_t w, uint32_t neg) {
            const uint32_t bin = b >> ss.shift;
            const uint32_t pos = lbase[bin] + atomicAdd(&lh[bin], 1u);
            rec_coarse[pos] = make_uint2(sh.shared ? (b | (w << 24)) : b, i | (neg << 31));
          },
          [&](uint32_t g) {
            const uint32_t bin = ss.ncoarse + g;
            const uint32_t pos = lbase[bin] + atomicAdd(&lh[bin], 1u);
            rec_final[pos] = make_uint2(sh.nb + g, i);
          });
  }
}

// Bucket range [base, base + span) that the coarse-sorted tile [lo, hi) can touch.
ZG_D void tile_span(const uint2* __restrict__ rec, uint32_t lo, uint32_t hi, uint32_t shift, uint32_t* base,
                    uint32_t* span) {
  const uint32_t first = (rec[lo].x & KEY_MASK) >> shift, last = (rec[hi - 1].x & KEY_MASK) >> shift;
  *base = first << shift;
  *span = (last - first + 1) << shift;
}

__global__ __launch_bounds__(256) void k_fine_count(const uint2* __restrict__ rec, const uint32_t* __restrict__ coff,
                                                    SortShape ss, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[FINE_LDS];
  const uint32_t tid = threadIdx.x, total = coff[ss.ncoarse];
  const uint32_t lo = blockIdx.x * FINE_TILE;
  if (lo >= total) return;
  const uint32_t hi = lo + FINE_TILE < total ? lo + FINE_TILE : total;
  uint32_t base, span;
  tile_span(rec, lo, hi, ss.shift, &base, &span);
  if (span > FINE_LDS) {
    for (uint32_t i = lo + tid; i < hi; i += 256) atomicAdd(&hist[rec[i].x & KEY_MASK], 1u);
    return;
  }
  for (uint32_t j = tid; j < span; j += 256) lh[j] = 0;
  __syncthreads();
  for (uint32_t i = lo + tid; i < hi; i += 256) atomicAdd(&lh[(rec[i].x & KEY_MASK) - base], 1u);
  __syncthreads();
  for (uint32_t j = tid; j < span; j += 256)
    if (lh[j]) atomicAdd(&hist[base + j], lh[j]);
}

__global__ __launch_bounds__(256) void k_fine_scatter(const uint2* __restrict__ rec, const uint32_t* __restrict__ coff,
                                                      SortShape ss, uint32_t* __restrict__ cursor,
                                                      uint2* __restrict__ out) {
  __shared__ uint32_t lh[FINE_LDS];
  const uint32_t tid = threadIdx.x, total = coff[ss.ncoarse];
  const uint32_t lo = blockIdx.x * FINE_TILE;
  if (lo >= total) return;
  const uint32_t hi = lo + FINE_TILE < total ? lo + FINE_TILE : total;
  uint32_t base, span;
  tile_span(rec, lo, hi, ss.shift, &base, &span);
  if (span > FINE_LDS) {
    for (uint32_t i = lo + tid; i < hi; i += 256) {
      const uint2 r = rec[i];
      out[atomicAdd(&cursor[r.x & KEY_MASK], 1u)] = r;
    }
    return;
  }
  for (uint32_t j = tid; j < span; j += 256) lh[j] = 0;
  __syncthreads();
  uint2 r[FINE_PER_LANE];
  uint32_t rank[FINE_PER_LANE];
#pragma unroll
  for (uint32_t k = 0; k < FINE_PER_LANE; k++) {
    const uint32_t i = lo + k * 256 + tid;
    if (i < hi) {
      r[k] = rec[i];
      rank[k] = atomicAdd(&lh[(r[k].x & KEY_MASK) - base], 1u);
    }
  }
  __syncthreads();
  for (uint32_t j = tid; j < span; j += 256) {
    const uint32_t cnt = lh[j];
    if (cnt) lh[j] = atomicAdd(&cursor[base + j], cnt);  // the counter becomes the start of this tile's run
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < FINE_PER_LANE; k++) {
    const uint32_t i = lo + k * 256 + tid;
    if (i < hi) out[lh[(r[k].x & KEY_MASK) - base] + rank[k]] = r[k];
  }
}

// Exclusive scan of hist[0..count) into off[0..count] (and a copy in cursor[0..count)) in two
// launches: per-workgroup scans of 4096 counters, then a fix-up by the sums of the tiles before.
static constexpr uint32_t SCAN_TILE = 4096;

// (hist and off may be the same array: a lane reads its sixteen counters before it writes them)
__global__ __launch_bounds__(256) void k_scan_local(const uint32_t* hist, uint32_t count, uint32_t* off,
                                                    uint32_t* __restrict__ tile_sum) {
  __shared__ uint32_t part[256];
  const uint32_t tid = threadIdx.x;
  const uint32_t base = blockIdx.x * SCAN_TILE + tid * 16;
  uint32_t v[16], s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    v[i] = base + i < count ? hist[base + i] : 0;
    s += v[i];
  }
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t x = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += x;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (base + i < count) off[base + i] = run;
    run += v[i];
  }
  if (tid == 255) tile_sum[blockIdx.x] = part[255];
}

// Second and last launch of the scan: every workgroup adds up the sums of the tiles before its own (at most a few
// hundred values -- cheaper than a launch of its own for a scan of them) and adds that to its 256 locally scanned
// counters.  (cursor may be the same array as off.)
__global__ __launch_bounds__(256) void k_scan_fixup(uint32_t* off, uint32_t* cursor, uint32_t count,
                                                    const uint32_t* __restrict__ tile_sum, uint32_t ntiles) {
  __shared__ uint32_t part[256];
  const uint32_t tid = threadIdx.x;
  const uint32_t i = blockIdx.x * blockDim.x + tid;
  // (SCAN_TILE is a multiple of the workgroup size: all of a workgroup's counters lie in one tile; the workgroup
  // that holds index `count` needs every tile)
  const uint32_t first = blockIdx.x * blockDim.x;
  const uint32_t before = first >= count ? ntiles : first / SCAN_TILE;
  uint32_t s = 0;
  for (uint32_t t = tid; t < before; t += 256) s += tile_sum[t];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 128; d >= 1; d >>= 1) {
    if (tid < d) part[tid] += part[tid + d];
    __syncthreads();
  }
  const uint32_t base = part[0];
  if (i < count) {
    const uint32_t v = off[i] + base;
    off[i] = v;
    cursor[i] = v;
  } else if (i == count) {
    // (first < count: the tiles before this workgroup's, its own tile's total on top)
    uint32_t total = base;
    if (first < count)
      for (uint32_t t = before; t < ntiles; t++) total += tile_sum[t];
    off[count] = total;
  }
}

// Entries per lane of k_accumulate: the launch has a fixed number of lanes (sized to fill the GPU once) and the
// sorted entries are dealt out evenly, so the chunk length follows the entry count -- but never so short that an
// average bucket is cut into more than ~24 pieces (a sweep of 8 .. 28 was flat within noise above 12): mid-sized multiexps (2^15 .. 2^17 dense terms) have few,
// large buckets, and with 8-entry chunks every one of them went down the one-workgroup-per-bucket path
// (k_heavy: 0.49 ms of a 1.7 ms proof at 2^16 rows) instead of k_combine's.
ZG_D uint32_t chunk_len(uint32_t entries, uint32_t lanes, uint32_t nb) {
  uint32_t k = (entries + lanes - 1) / lanes;
  const uint32_t per_bucket = entries / nb / 24;
  if (k < per_bucket) k = per_bucket;
  return k < ACC_KMIN ? ACC_KMIN : k;
}

// A bucket [s, e) of the sorted list that level 0 cut at chunk edges has one partial sum per chunk it
// touches: piece 0 in the first chunk's tail slot (head slot if the bucket starts the chunk -- it is
// then that chunk's first run), every later piece in its chunk's head slot.
ZG_D uint32_t piece_slot(uint32_t s, uint32_t K, uint32_t piece) {
  const uint32_t cf = s / K;
  if (piece == 0) return 2 * cf + (s % K != 0);
  return 2 * (cf + piece);
}

// Level 0 of the segmented reduction: logical thread t owns sorted entries [K t, K t + K) whatever
// the bucket sizes are (K = chunk_len of the entry count off[nb], read on the device).  A run that
// lies inside the chunk finishes its bucket; a run cut by a chunk edge leaves a partial sum in slot
// 2t (first run) or 2t+1 (last run) for k_combine / k_heavy.
//
// (Measured and dropped: fetching the base of entry i + 1 by direct-to-LDS loads while entry i is added, sort
// entries in an LDS ring two ahead, no load left inside the loop.  Bit-identical, and 3 % SLOWER at 2^20 rows: with
// four waves per SIMD the gather latency is already hidden -- the fifth of the issue slots this kernel leaves empty
// is the 64-bit multiply-add's own issue rate, two thirds of the 32-bit rate, not a wait for memory.)
//
// off points at the offsets of the nb buckets this launch sums, whose keys in the entries are key0 .. key0 + nb - 1:
// the whole sorted list (key0 = 0, off[0] = 0) or, when the sort was another launch's, one group's part of it.
template <class F>
__global__ __launch_bounds__(256, kMinWaves<F>) void k_accumulate(JobTable<F> jt, uint32_t nb, uint32_t lanes,
                                                                  const uint2* __restrict__ rec,
                                                                  const uint32_t* __restrict__ off, uint32_t key0,
                                                                  typename MsmOps<F>::PtMem* __restrict__ pts_out,
                                                                  typename MsmOps<F>::PtMem* __restrict__ bucket_sum,
                                                                  uint32_t* __restrict__ err_flag) {
  using O = MsmOps<F>;
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / O::LP;
  const uint32_t e0 = off[0];
  const uint32_t nslots = off[nb] - e0;  // digit entries; the unit-scalar entries behind them go to k_sum_range
  if (blockIdx.x == 0 && threadIdx.x == 0) err_flag[3] = nslots;  // the additions of this launch, for the host's counters
  const uint32_t K = chunk_len(nslots, lanes, nb);
  const uint64_t lo64 = (uint64_t)t * K;
  if (lo64 >= nslots) return;
  const uint32_t lo = e0 + (uint32_t)lo64;
  const uint32_t hi = nslots - (uint32_t)lo64 > K ? lo + K : e0 + nslots;
  uint32_t cur = KEY_INVALID;
  typename O::Pt acc = O::inf();
  bool first_run = true;

  auto flush = [&]() {
    if (off[cur] >= lo && off[cur + 1] <= hi) {
      O::store(bucket_sum + cur, acc);
    } else {
      const int slot = first_run ? 0 : 1;  // a cut run is the first or (otherwise) the last of its chunk
      O::store(pts_out + (2 * (size_t)t + slot), acc);
    }
    first_run = false;
  };

  for (uint32_t i = lo; i < hi; i++) {
    const uint2 rc = rec[i];
    const uint32_t kw = rc.x, e = rc.y;
    // (a key outside this launch's buckets can only come from a damaged entry or a wrong offset table: clamped, so
    // that such a bug gives a wrong sum instead of a read far outside `off`)
    const uint32_t krel = (kw & KEY_MASK) - key0;
    const uint32_t k = krel < nb ? krel : nb - 1;
    if (k != cur) {
      if (cur != KEY_INVALID) flush();
      cur = k;
      acc = O::inf();
    }
    typename O::Aff p;
    if (O::load_base(base_ptr(jt, e & 0x7fffffffu, kw >> 24), (e >> 31) != 0, &p))
      O::madd(acc, p);
    else
      *err_flag = 1;  // identity base consumed by a non-zero scalar
  }
  if (cur != KEY_INVALID) flush();
}

// Q adjacent groups of SL lanes per bucket add, in the spread form, the pieces level 0 left of a bucket it cut at
// chunk edges: group q takes pieces q, q + Q, ... and the groups fold by shuffles.  Q is 1 when there are buckets
// enough to fill the GPU with one group each (a 2^20-point query: 65 536 buckets) and up to 8 for the few, deeply cut
// buckets of a mid-sized query (8192 buckets in ~25 pieces: a chain of 25 additions became 4 + 3).  Buckets in more
// than LIGHT_PIECES pieces are appended to heavy_list for k_heavy.  Q * SL divides 64.
template <class F>
__global__ __launch_bounds__(256, kMinWaves<F>) void k_combine(uint32_t nb, uint32_t lanes, uint32_t Q,
                                                                 const uint32_t* __restrict__ off,
                                                                 const typename MsmOps<F>::PtMem* __restrict__ pts1,
                                                                 typename MsmOps<F>::PtMem* __restrict__ bucket_sum,
                                                                 uint32_t* __restrict__ heavy_count,
                                                                 uint32_t* __restrict__ heavy_list) {
  using O = MsmOps<F>;
  const uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) / O::SL;
  const uint32_t b = grp / Q, q = grp % Q;
  const uint32_t e0 = off[0];  // positions below are counted from the first entry of this launch's buckets
  const uint32_t K = chunk_len(off[nb] - e0, lanes, nb);
  // (every lane of a wave stays for the shuffles below; a bucket with nothing to add here has no pieces)
  uint32_t s = 0, pieces = 0;
  if (b < nb) {
    s = off[b] - e0;
    const uint32_t e = off[b + 1] - e0;
    if (e > s) pieces = (e - 1) / K - s / K + 1;
    if (pieces > LIGHT_PIECES) {
      if (q == 0 && threadIdx.x % O::SL == 0) heavy_list[atomicAdd(heavy_count, 1u)] = b;
      pieces = 0;
    }
    if (pieces == 1) pieces = 0;  // level 0 finished the bucket inside one chunk
  }
  // (the next piece is on its way while the current one is added: the chain is one addition after another)
  typename O::Own acc = O::own_inf(), next = O::own_inf();
  if (q < pieces) next = O::own_load(pts1 + piece_slot(s, K, q));
  for (uint32_t p = q; p < pieces; p += Q) {
    const typename O::Own cur = next;
    if (p + Q < pieces) next = O::own_load(pts1 + piece_slot(s, K, p + Q));
    O::own_add(acc, cur);
  }
#pragma unroll 1
  for (uint32_t d = Q / 2; d >= 1; d >>= 1) {
    const typename O::Own o = O::own_shfl_down(acc, (int)(d * O::SL));
    O::own_add_if(acc, o, q < d);
  }
  if (pieces && q == 0) O::own_store(bucket_sum + b, acc);
}

// p * k for a small unsigned k
template <class F>
ZG_D typename MsmOps<F>::Pt mul_small(const typename MsmOps<F>::Pt& p, uint32_t k) {
  using O = MsmOps<F>;
  typename O::Pt acc = O::inf();
  for (int b = 31 - __clz(k | 1); b >= 0; b--) {
    acc = O::dbl(acc);
    if ((k >> b) & 1) O::add(acc, p);
  }
  return acc;
}

// ---- sums in the spread form (see quad_add) ----------------------------------------------------------------------
// What a workgroup sum needs in LDS: a parking slot per logical thread and the four waves' sums.
template <class F>
struct SpreadLds {
  typename MsmOps<F>::PtMem park[256 / MsmOps<F>::LP];
  typename MsmOps<F>::Own fold[4][MsmOps<F>::SL];
};

// Sum of `count` points by the 256 lanes of a workgroup (every lane must call); own(i) is this lane's part of point
// i.  Groups of SL lanes stride over the points, then the groups fold: inside each wave by shuffles, across the
// four waves through `fold`.  The sum is valid in the first SL lanes of the workgroup.
template <class F, class OwnOf>
ZG_D typename MsmOps<F>::Own wg_spread_sum(uint32_t count, OwnOf own, typename MsmOps<F>::Own (*fold)[MsmOps<F>::SL]) {
  using O = MsmOps<F>;
  constexpr uint32_t SL = O::SL, GW = 64 / SL, G = 256 / SL;  // groups per wave, per workgroup
  const uint32_t g = threadIdx.x / SL, gw = g % GW, wave = threadIdx.x / 64, sub = threadIdx.x % SL;
  typename O::Own acc = O::own_inf();
  for (uint32_t i = g; i < count; i += G) O::own_add(acc, own(i));
  // the fold starts at the first level that still has a point in its upper half
  const uint32_t n_act = count < G ? count : G;  // groups that hold a point
  uint32_t top = GW;
  if (n_act < GW) {
    top = 1;
    while (top < n_act) top <<= 1;
  }
#pragma unroll 1
  for (uint32_t d = top / 2; d >= 1; d >>= 1) {
    const typename O::Own o = O::own_shfl_down(acc, (int)(d * SL));
    O::own_add_if(acc, o, gw < d);
  }
  if (n_act > GW) {  // (the same in every lane of the workgroup)
    if (gw == 0) fold[wave][sub] = acc;
    __syncthreads();
    if (wave == 0) {
      acc = g < 4 ? fold[g][sub] : O::own_inf();
#pragma unroll 1
      for (uint32_t d = 2; d >= 1; d >>= 1) {
        const typename O::Own o = O::own_shfl_down(acc, (int)(d * SL));
        O::own_add_if(acc, o, gw < d);
      }
    }
    __syncthreads();  // `fold` may be written again by the caller's next sum
  }
  return acc;
}

// Sum of the accumulators the logical threads of a workgroup hold (one lane, or lane pair, per point): each parks
// its point in LDS and the workgroup adds them in the spread form.  Valid in the first SL lanes.
template <class F>
ZG_D typename MsmOps<F>::Own block_sum_spread(const typename MsmOps<F>::Pt& total, SpreadLds<F>& lds) {
  using O = MsmOps<F>;
  O::store(lds.park + threadIdx.x / O::LP, total);
  __syncthreads();
  const typename O::Own r = wg_spread_sum<F>(256 / O::LP, [&](uint32_t i) { return O::own_load(lds.park + i); }, lds.fold);
  __syncthreads();  // `park` may be written again
  return r;
}

// The buckets k_combine left: a bucket in thousands of pieces (the few buckets of a narrow top window, the small
// values of a witness) is a long chain for one workgroup, so with fewer heavy buckets than workgroups a bucket is cut
// into slices of about HEAVY_PER_THREAD pieces per logical thread, one workgroup each: its threads stride over the
// slice's pieces, the workgroup adds them in the spread form, and the workgroup that finishes a bucket's last slice (a
// counter per bucket, zero at launch) adds the slice sums.  No workgroup waits for another; workgroups beyond a
// bucket's slices leave at once.  slice_sum and slices_done hold gridDim.x slots.
static constexpr uint32_t HEAVY_PER_THREAD = 6;
template <class F>
__global__ __launch_bounds__(256, kMinWaves<F>) void k_heavy(const uint32_t* __restrict__ heavy_count,
                                                               const uint32_t* __restrict__ heavy_list,
                                                               const uint32_t* __restrict__ off, uint32_t nb,
                                                               uint32_t lanes,
                                                               const typename MsmOps<F>::PtMem* __restrict__ pts1,
                                                               typename MsmOps<F>::PtMem* __restrict__ bucket_sum,
                                                               typename MsmOps<F>::PtMem* slice_sum,
                                                               uint32_t* __restrict__ slices_done) {
  using O = MsmOps<F>;
  __shared__ SpreadLds<F> lds;
  __shared__ uint32_t finished_bucket;
  const uint32_t n = *heavy_count;
  if (!n) return;
  const uint32_t max_slices = n < gridDim.x ? gridDim.x / n : 1;
  const uint32_t e0 = off[0];
  const uint32_t K = chunk_len(off[nb] - e0, lanes, nb);
  const uint32_t lt = threadIdx.x / O::LP, nlt = 256 / O::LP;
  for (uint32_t item = blockIdx.x; item < n * max_slices; item += gridDim.x) {  // (max_slices > 1: one item at most)
    const uint32_t h = item / max_slices, slice = item % max_slices;
    const uint32_t b = heavy_list[h];
    const uint32_t s = off[b] - e0, e = off[b + 1] - e0;
    const uint32_t pieces = (e - 1) / K - s / K + 1;
    uint32_t S = (pieces + nlt * HEAVY_PER_THREAD - 1) / (nlt * HEAVY_PER_THREAD);
    if (S > max_slices) S = max_slices;
    if (slice >= S) continue;
    typename O::Pt acc = O::inf();
    for (uint32_t p = slice * nlt + lt; p < pieces; p += S * nlt) O::add(acc, O::load(pts1 + piece_slot(s, K, p)));
    const typename O::Own sum = block_sum_spread<F>(acc, lds);
    if (S == 1) {
      if (threadIdx.x < O::SL) O::own_store(bucket_sum + b, sum);
      continue;
    }
    if (threadIdx.x < O::SL) O::own_store(slice_sum + item, sum);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) finished_bucket = atomicAdd(&slices_done[h], 1u) == S - 1;
    __syncthreads();
    if (finished_bucket) {  // (the same in every lane of the workgroup)
      __threadfence();
      const typename O::Own total =
          wg_spread_sum<F>(S, [&](uint32_t i) { return O::own_load_fresh(slice_sum + (size_t)h * max_slices + i); }, lds.fold);
      if (threadIdx.x < O::SL) O::own_store(bucket_sum + b, total);
    }
  }
}

// Weighted bucket sums without scalar multiplications.  sum_b (b+1)*S_b over B = R*H buckets with
// b = hi*R + lo splits into  sum_lo lo*U_lo + sum_hi (R*hi + 1)*V_hi,  U_lo = sum_hi S_b (column
// sums), V_hi = sum_lo S_b (row sums): two plain-sum arrays of R and H points.  The weights of U are the
// numbers lo < R and those of V are R * hi (+ 1), so bit by bit:  sum_p 2^p T_p + sum of all buckets,  T_p = the
// sum of the U_lo with bit p of lo set (p < log R), of the V_hi with bit p - log R of hi set (above): c - 1 plain
// sums of R / 2 or H / 2 points each (k_bit_sums), and the host finishes with c - 1 doublings and c additions per
// bucket set (msm_collect) -- a CPU core runs such a dependent chain an order of magnitude faster than one wave
// does, and the device part is two launches of log-depth sums.
//
// k_split_sums: each workgroup produces one output point (wg_spread_sum).  Array a of set blockIdx.z holds count[a] =
// r[a] * h[a] points at in + set*in_stride + in_off[a]; outputs are r[a] column sums (index mod r)
// followed by h[a] row sums (index div r) at out_off[a].  With out_std the outputs are written in
// the library-wide XYZZ form for the host instead.
struct SplitShape {
  uint32_t narr;
  uint32_t in_off[2], r[2], h[2], out_off[2];
  uint32_t in_stride, out_stride;
};

template <class F>
__global__ __launch_bounds__(256, kMinWaves<F>) void k_split_sums(const typename MsmOps<F>::PtMem* __restrict__ in,
                                                                    SplitShape ss,
                                                                    typename MsmOps<F>::PtMem* __restrict__ out,
                                                                    XYZZ<F>* __restrict__ out_std) {
  using O = MsmOps<F>;
  __shared__ typename O::Own fold[4][O::SL];
  const uint32_t a = blockIdx.y, set = blockIdx.z;
  const uint32_t r = ss.r[a], h = ss.h[a];
  const uint32_t task = blockIdx.x;
  if (task >= r + h) return;
  const typename O::PtMem* X = in + (size_t)set * ss.in_stride + ss.in_off[a];
  typename O::Own acc;
  if (task < r) {
    acc = wg_spread_sum<F>(h, [&](uint32_t k) { return O::own_load(X + (size_t)k * r + task); }, fold);
  } else {
    const typename O::PtMem* row = X + (size_t)(task - r) * r;
    acc = wg_spread_sum<F>(r, [&](uint32_t k) { return O::own_load(row + k); }, fold);
  }
  if (threadIdx.x < O::SL) {
    const size_t dst = (size_t)set * ss.out_stride + ss.out_off[a] + task;
    if (out_std)
      O::own_store_std(out_std + dst, acc);
    else
      O::own_store(out + dst, acc);
  }
}

// Set blockIdx.y holds U (2^lR points) then V (2^lH points) at in + set * in_stride.  Workgroup blockIdx.x = p
// writes T_p (see above) for p < lR + lH and the sum of all of U for p = lR + lH, in the library-wide XYZZ form, at
// out_std[set * (lR + lH + 1) + p].
template <class F>
__global__ __launch_bounds__(256, kMinWaves<F>) void k_bit_sums(const typename MsmOps<F>::PtMem* __restrict__ in,
                                                                  uint32_t lR, uint32_t lH, uint32_t in_stride,
                                                                  XYZZ<F>* __restrict__ out_std) {
  using O = MsmOps<F>;
  __shared__ typename O::Own fold[4][O::SL];
  const uint32_t p = blockIdx.x, set = blockIdx.y, bits = lR + lH;
  const typename O::PtMem* U = in + (size_t)set * in_stride;
  typename O::Own acc;
  if (p == bits) {
    acc = wg_spread_sum<F>(1u << lR, [&](uint32_t k) { return O::own_load(U + k); }, fold);
  } else {
    const typename O::PtMem* X = p < lR ? U : U + ((size_t)1 << lR);
    const uint32_t t = p < lR ? p : p - lR, len = p < lR ? lR : lH;
    // the k-th index with bit t set: the bits of k above t move up by one
    acc = wg_spread_sum<F>((1u << len) >> 1, [&](uint32_t k) {
      return O::own_load(X + ((((k >> t) << 1) | 1u) << t | (k & ((1u << t) - 1))));
    }, fold);
  }
  if (threadIdx.x < O::SL) O::own_store_std(out_std + ((size_t)set * (bits + 1) + p), acc);
}

// out_std[w] = sum of part[w][0..count), in the library-wide XYZZ form.  One workgroup per w.
template <class F>
__global__ __launch_bounds__(256, kTailWaves) void k_sum_partials(const typename MsmOps<F>::PtMem* __restrict__ part,
                                                                      uint32_t count, XYZZ<F>* __restrict__ out_std) {
  using O = MsmOps<F>;
  __shared__ typename O::Own fold[4][O::SL];
  const uint32_t w = blockIdx.x;
  const typename O::Own acc = wg_spread_sum<F>(count, [&](uint32_t i) { return O::own_load(part + ((size_t)w * count + i)); }, fold);
  if (threadIdx.x < O::SL) O::own_store_std(out_std + w, acc);
}

// Plain sum of the bases whose scalar equals one, group blockIdx.y: the entries are
// rec[off[nb+g] .. off[nb+g+1]).  Each logical thread adds a contiguous share, the workgroup
// tree-adds, one partial per workgroup at part[g * gridDim.x + blockIdx.x].
template <class F>
__global__ __launch_bounds__(256, kMinWaves<F>) void k_sum_range(JobTable<F> jt, const uint2* __restrict__ rec,
                                                                   const uint32_t* __restrict__ off, uint32_t nb,
                                                                   typename MsmOps<F>::PtMem* __restrict__ part,
                                                                   uint32_t* __restrict__ err_flag) {
  using O = MsmOps<F>;
  __shared__ SpreadLds<F> lds;
  const uint32_t g = blockIdx.y, lt = threadIdx.x / O::LP, nlt = 256 / O::LP;
  const uint32_t lo = off[nb + g], hi = off[nb + g + 1];
  if (lo == hi) {  // no scalar of the group equals one (the h query's are uniform): nothing to add in any workgroup
    if (threadIdx.x < O::SL) O::own_store(part + ((size_t)g * gridDim.x + blockIdx.x), O::own_inf());
    return;
  }
  const uint32_t nthreads = gridDim.x * nlt, t = blockIdx.x * nlt + lt;
  const uint32_t per = (hi - lo + nthreads - 1) / nthreads;
  const uint32_t a = lo + t * per < hi ? lo + t * per : hi;
  const uint32_t b = a + per < hi ? a + per : hi;
  typename O::Pt acc = O::inf();
  for (uint32_t i = a; i < b; i++) {
    typename O::Aff p;
    if (O::load_base(base_ptr(jt, rec[i].y & 0x7fffffffu, 0), false, &p))
      O::madd(acc, p);
    else
      *err_flag = 1;
  }
  const typename O::Own sum = block_sum_spread<F>(acc, lds);
  if (threadIdx.x < O::SL) O::own_store(part + ((size_t)g * gridDim.x + blockIdx.x), sum);
}


// ---- direct-lookup multiexp for small queries ---------------------------------------------------
// Below ~2^14 points a bucket multiexp is ~20 dependent launches of almost no work each, and that chain is what a
// small circuit's proof waits for.  With 288 GB of HBM such a query can instead keep EVERY multiple a digit can
// ask for: with signed 5-bit digits, direct[(w * 16 + (|d| - 1)) * n + k] = |d| * 2^(5 w) * P_k for 52 windows --
// 52 KiB per G1 point.  The multiexp is then a plain sum of one looked-up point per non-zero digit: one launch of
// independent mixed additions plus the workgroup trees, one launch to add the workgroups' partial sums.
// Digits: s' = s + sum_{w < 51} 16 * 32^w makes every window w < 51 of s' an unsigned digit u with d = u - 16 in
// [-16, 15]; window 51 (bits 255 and up of s' < 2^256) is 0 or 1 and is taken as it is.
static constexpr uint32_t DIRECT_C = 5, DIRECT_M = 16, DIRECT_WB = 51, DIRECT_W = 52;
static constexpr uint32_t DIRECT_PER_THREAD = 8;  // table look-ups per logical thread of k_direct_sum

struct DirectBias {
  uint32_t v[8];  // sum_{w < 51} 16 * 32^w
};

static DirectBias direct_bias() {
  DirectBias b = {};
  for (uint32_t w = 0; w < DIRECT_WB; w++) {
    const uint32_t bit = w * DIRECT_C + (DIRECT_C - 1);
    b.v[bit >> 5] |= 1u << (bit & 31);
  }
  return b;
}

size_t msm_direct_max() {
  const char* e = getenv("ZG16_DIRECT_MAX_LOG");
  const int lg = e ? atoi(e) : 14;
  return lg <= 0 ? 0 : (size_t)1 << (lg > 16 ? 16 : lg);
}

// Job blockIdx.y: every logical thread adds the looked-up multiples of a contiguous share of the job's
// (window, term) pairs, the workgroup tree-adds, one partial per workgroup at part[job * gridDim.x + blockIdx.x].
template <class F>
__global__ __launch_bounds__(256, kTailWaves) void k_direct_sum(JobTable<F> jt, DirectBias bias,
                                                                    typename MsmOps<F>::PtMem* __restrict__ part,
                                                                    uint32_t* __restrict__ err_flag) {
  using O = MsmOps<F>;
  __shared__ SpreadLds<F> lds;
  const int j = blockIdx.y;
  const uint32_t n = jt.start[j + 1] - jt.start[j];
  const uint32_t lt = threadIdx.x / O::LP, nlt = 256 / O::LP;
  const uint32_t total = n * DIRECT_W;
  const uint32_t nthreads = gridDim.x * nlt, t = blockIdx.x * nlt + lt;
  const uint32_t per = (total + nthreads - 1) / nthreads;
  const uint32_t a = (uint64_t)t * per < total ? t * per : total;
  const uint32_t b = total - a > per ? a + per : total;
  typename O::Pt acc = O::inf();
  uint32_t w = n ? a / n : 0, k = n ? a - w * n : 0;
  for (uint32_t e = a; e < b; e++) {
    const Fr s = from_mont(load_scalar(jt, jt.start[j] + k, j));
    uint32_t sp[8];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      carry += (uint64_t)s.v[i] + bias.v[i];
      sp[i] = (uint32_t)carry;
      carry >>= 32;
    }
    const uint32_t u = extract_bits(sp, w * DIRECT_C, DIRECT_C);
    const int d = w < DIRECT_WB ? (int)u - (int)DIRECT_M : (int)u;
    if (d != 0) {
      const uint32_t mag = (uint32_t)(d < 0 ? -d : d) - 1;
      typename O::Aff p;
      if (O::load_base(jt.direct[j] + ((size_t)(w * DIRECT_M + mag) * jt.stride[j] + k), d < 0, &p))
        O::madd(acc, p);
      else
        *err_flag = 1;  // identity base consumed by a non-zero scalar
    }
    if (++k == n) {
      k = 0;
      w++;
    }
  }
  const typename O::Own sum = block_sum_spread<F>(acc, lds);
  if (threadIdx.x < O::SL) O::own_store(part + ((size_t)j * gridDim.x + blockIdx.x), sum);
}

// table[w*stride + i] = 2^(c*w) * table[i] for w in 1..W-1, every window in the internal packed form
// (window 0 arrives in the ABI form and is rewritten).  W = 1 just converts the points.  (c here is the distance
// between two stored windows in bits: the digit width times the table step.)
//
// Bringing a window back to affine form is a field inversion -- three times the cost of the c doublings that lead
// to it -- so it is not done per window.  Three kernels:
//   k_table_convert   window 0 into the packed form the multiexp kernels read;
//   k_table_windows   one logical thread per point (a lane pair for G2) runs the doublings in the 29-bit-limb
//                     arithmetic of the multiexp kernels and leaves every window in XYZZ form in a scratch block
//                     (window-major, so neighbouring threads store neighbouring slots);
//   k_table_affine    per point: the running product of the windows' denominators ZZ*ZZZ, ONE inversion, and a
//                     walk back from the last window that peels the single inverses off it (Montgomery's trick):
//                     an inversion and ten products per point and window became thirteen products.
// The coordinates are canonical field values, so the table does not depend on how they were reached.
// Thread j of a launch handles point first + j; xyzz and prefix hold (W - 1) * chunk slots.
template <class F>
__global__ __launch_bounds__(256) void k_table_convert(Affine<F>* __restrict__ table, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) table[i] = MsmOps<F>::to_internal(table[i]);
}

template <class F>
__global__ __launch_bounds__(256, kMinWaves<F>) void k_table_windows(const Affine<F>* __restrict__ table, uint32_t n,
                                                                     uint32_t W, uint32_t c, uint32_t first, uint32_t chunk,
                                                                     XYZZ<F>* __restrict__ xyzz) {
  using Ops = MsmOps<F>;
  const uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) / Ops::LP, i = first + j;
  if (j >= chunk || i >= n) return;  // (the lanes of a pair share j)
  typename Ops::Aff base;
  typename Ops::Pt acc = Ops::inf();
  if (Ops::load_base(table + i, false, &base)) Ops::madd(acc, base);
  for (uint32_t w = 1; w < W; w++) {
    for (uint32_t k = 0; k < c; k++) acc = Ops::dbl(acc);
    Ops::store_std(xyzz + (size_t)(w - 1) * chunk + j, acc);
  }
}

// (group: 0 for a window table -- thread first + j owns point i = first + j.  For a direct table the threads are
// (window, point) pairs t = w * npts + k whose W entries start at slot (w * group) * npts + k, npts apart.)
ZG_D uint32_t table_slot(uint32_t t, uint32_t group, uint32_t npts) {
  return group ? (t / npts) * group * npts + t % npts : t;
}

template <class F>
__global__ __launch_bounds__(256, 1) void k_table_affine(Affine<F>* __restrict__ table, uint32_t n, uint32_t stride,
                                                         uint32_t W, uint32_t first, uint32_t chunk,
                                                         const XYZZ<F>* __restrict__ xyzz, F* __restrict__ prefix,
                                                         uint32_t group, uint32_t npts) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= chunk || first + j >= n) return;
  const uint32_t i = table_slot(first + j, group, npts);
  F run = F::one();
  for (uint32_t w = 1; w < W; w++) {
    const XYZZ<F> a = xyzz[(size_t)(w - 1) * chunk + j];
    // (a point of the prime-order group never doubles to the identity; a stray point that does is kept out of
    // the product and stored as the identity below)
    if (!a.is_inf()) run = mul(run, mul(a.zz, a.zzz));
    prefix[(size_t)(w - 1) * chunk + j] = run;
  }
  F inv_run = inv(run);  // 1 / (t_1 ... t_(W-1)), t_w = ZZ_w * ZZZ_w
  for (uint32_t w = W - 1; w >= 1; w--) {
    const XYZZ<F> a = xyzz[(size_t)(w - 1) * chunk + j];
    Affine<F> out = Affine<F>::inf();
    if (!a.is_inf()) {
      const F before = w > 1 ? prefix[(size_t)(w - 2) * chunk + j] : F::one();
      const F it = mul(inv_run, before);         // 1 / t_w
      inv_run = mul(inv_run, mul(a.zz, a.zzz));  // 1 / (t_1 ... t_(w-1))
      out = {mul(a.x, mul(it, a.zzz)), mul(a.y, mul(it, a.zz))};
    }
    table[(size_t)w * stride + i] = MsmOps<F>::to_internal(out);
  }
}

static uint32_t pick_window(size_t n) {
  const char* e = getenv("ZG16_MSM_WINDOW");
  if (e) {
    int v = atoi(e);
    if (v >= 2 && v <= 16) return (uint32_t)v;
  }
  unsigned lg = 0;
  while (((size_t)2 << lg) <= n) lg++;
  int c = (int)lg - 4;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  return (uint32_t)c;
}

// Table window for a query of `points` points.  The h query meets dense scalars: about 2^4 terms
// per window per bucket keeps buckets inside k_combine's reach and the bucket work small.  The
// queries over the witness (role 1: l, a, b_g1 share one launch and one width; role 2: b_g2) meet
// mostly bits, which bypass the buckets, so they get a narrower window (one bit narrower: two bits narrower measured
// the same on bit-heavy witnesses from 2^14 to 2^24 rows, 2-3 % slower at 2^16 rows and on dense witnesses, and its
// tables are a window larger).
// Returns 0 when tables are switched off (ZG16_MSM_TABLE_WINDOW=0): per-window buckets are used.
// ZG16_MSM_TABLE_WINDOW sets every width, ZG16_MSM_TABLE_WINDOW_H / _W / _G2 one role's.
uint32_t msm_table_window(size_t points, int role) {
  static const char* const role_env[3] = {"ZG16_MSM_TABLE_WINDOW_H", "ZG16_MSM_TABLE_WINDOW_W", "ZG16_MSM_TABLE_WINDOW_G2"};
  const char* e = getenv("ZG16_MSM_TABLE_WINDOW");
  if (e && atoi(e) == 0) return 0;
  const char* er = getenv(role_env[role]);
  if (er) e = er;
  if (e) {
    int v = atoi(e);
    if (v >= 8 && v <= 20) return (uint32_t)v;
  }
  unsigned lg = 0;
  while (((size_t)2 << lg) <= points) lg++;
  int c = (int)lg - (role == 0 ? 3 : 4);
  if (c < 8) c = 8;
  if (c > 20) c = 20;
  // A width whose last window holds only a few of the scalar's 254 bits sends every term's top digit to a handful
  // of buckets (c = 14: 19 windows, 2 bits in the last, four buckets with a quarter of the terms each): those are
  // the heavy buckets of a uniform query.  Widen until the last window is at least half full.
  while (c < 20 && 254 - ((255 + c - 1) / c - 1) * c < (c + 1) / 2) c++;
  return (uint32_t)c;
}

uint32_t msm_table_windows(uint32_t c, uint32_t step) {
  const uint32_t W = (255 + c - 1) / c;
  return (W + step - 1) / step;
}

uint32_t msm_table_step(const TableQuery* q, int count) {
  size_t budget;
  const char* e = getenv("ZG16_TABLE_BUDGET_MB");
  if (e) {
    budget = (size_t)atol(e) << 20;
  } else {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 1;
    budget = free_b / 5 * 4;
  }
  uint32_t max_w = 1, max_b = 1;
  for (int i = 0; i < count; i++) {
    if (!q[i].c || !q[i].points) continue;
    max_w = std::max(max_w, msm_table_windows(q[i].c));
    max_b = std::max(max_b, 1u << (q[i].c - 1));
  }
  // MAX_JOBS groups of `step` bucket sets each must stay below the 24-bit bucket index of a sort entry
  const uint32_t max_step = std::max(1u, std::min(max_w, (KEY_MASK - 16) / ((uint32_t)MAX_JOBS * max_b)));
  for (uint32_t step = 1; step <= max_step; step++) {
    size_t bytes = 0;
    for (int i = 0; i < count; i++)
      if (q[i].c) bytes += q[i].points * msm_table_windows(q[i].c, step) * q[i].point_bytes;
    if (bytes <= budget) return step;
  }
  return max_step;  // the fewest windows there are; an allocation that still does not fit reports out of memory
}

template <class F>
zg16_status msm_build_table(zg16_ctx* ctx, Affine<F>* table, size_t n, uint32_t c, uint32_t step) {
  if (!n) return ZG16_OK;
  const uint32_t W = c ? msm_table_windows(c, step) : 1;
  // points per launch: the scratch block of a launch (W - 1 windows in XYZZ form and their prefix products) stays
  // below about 1 GiB; freed again before returning -- a key upload is no place to keep it
  size_t chunk = (size_t)1 << 18;
  if (chunk > n) chunk = n;
  XYZZ<F>* xyzz = nullptr;
  F* prefix = nullptr;
  if (W > 1) {
    ZG_HIP(ctx, hipMalloc((void**)&xyzz, (size_t)(W - 1) * chunk * sizeof(XYZZ<F>)));
    const hipError_t e = hipMalloc((void**)&prefix, (size_t)(W - 1) * chunk * sizeof(F));
    if (e != hipSuccess) {
      (void)hipFree(xyzz);
      ZG_HIP(ctx, e);
    }
  }
  hipStream_t st = ctx->stream;
  k_table_convert<F><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(table, (uint32_t)n);
  for (size_t first = 0; W > 1 && first < n; first += chunk) {
    const size_t here = n - first < chunk ? n - first : chunk;
    k_table_windows<F><<<(unsigned)((here * MsmOps<F>::LP + 255) / 256), 256, 0, st>>>(table, (uint32_t)n, W, c * step, (uint32_t)first,
                                                                                 (uint32_t)chunk, xyzz);
    k_table_affine<F><<<(unsigned)((here + 255) / 256), 256, 0, st>>>(table, (uint32_t)n, (uint32_t)n, W, (uint32_t)first,
                                                                   (uint32_t)chunk, xyzz, prefix, 0, 0);
  }
  if (ctx->deferred_frees) {  // the caller waits for the stream and frees the scratch
    ctx->deferred_frees->push_back(xyzz);
    ctx->deferred_frees->push_back(prefix);
    ZG_HIP(ctx, hipGetLastError());
    return ZG16_OK;
  }
  const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ctx->stream);
  (void)hipFree(xyzz);
  (void)hipFree(prefix);
  ZG_HIP(ctx, e1);
  ZG_HIP(ctx, e2);
  return ZG16_OK;
}
template zg16_status msm_build_table<Fq>(zg16_ctx*, Affine<Fq>*, size_t, uint32_t, uint32_t);
template zg16_status msm_build_table<Fq2>(zg16_ctx*, Affine<Fq2>*, size_t, uint32_t, uint32_t);

// ---- direct tables, built -----------------------------------------------------------------------------------
// direct[(w * 16 + m) * n + k] = (m + 1) * 2^(5 w) * P_k.  Built like a window table, twice over: the 52 window
// bases 2^(5 w) * P_k are a window table of width 5 whose windows lie 16 n slots apart (k_table_windows /
// k_table_affine: one chain of 255 doublings and one inversion per point); then one logical thread per (window,
// point) adds the base to itself fifteen times (k_direct_multiples, 29-bit arithmetic) and k_table_affine brings
// those fifteen multiples back to affine form with ONE inversion.  (The first version gave every (window, point)
// thread its own 5 w doublings from P_k and an inversion per multiple: 75 ms for the five queries of a 1.8 k-row
// circuit and 316 ms at 2^14 rows -- more than a hundred proofs' worth of what the table saves.)
template <class F>
__global__ __launch_bounds__(256, kMinWaves<F>) void k_direct_multiples(const Affine<F>* __restrict__ direct, uint32_t total,
                                                                        uint32_t npts, uint32_t first, uint32_t chunk,
                                                                        XYZZ<F>* __restrict__ xyzz) {
  using Ops = MsmOps<F>;
  const uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) / Ops::LP, t = first + j;
  if (j >= chunk || t >= total) return;  // (the lanes of a pair share j)
  typename Ops::Aff base;
  typename Ops::Pt acc = Ops::inf();
  const bool finite = Ops::load_base(direct + table_slot(t, DIRECT_M, npts), false, &base);
  if (finite) Ops::madd(acc, base);
  for (uint32_t m = 1; m < DIRECT_M; m++) {
    if (finite) Ops::madd(acc, base);  // (m = 1 adds the base to itself: the addition's doubling case)
    Ops::store_std(xyzz + (size_t)(m - 1) * chunk + j, acc);
  }
}

template <class F>
zg16_status msm_build_direct(zg16_ctx* ctx, const Affine<F>* points, size_t n, Affine<F>** out) {
  *out = nullptr;
  if (!n || n > msm_direct_max()) return ZG16_OK;
  const size_t entries = n * DIRECT_W * DIRECT_M, total = n * DIRECT_W;
  hipStream_t st = ctx->stream;
  Affine<F>* direct = nullptr;
  ZG_HIP(ctx, hipMalloc((void**)&direct, entries * sizeof(Affine<F>)));
  // scratch: the larger of the two uses -- DIRECT_W - 1 window bases per point, DIRECT_M - 1 multiples per chunk
  // of (window, point) threads
  size_t chunk = (size_t)1 << 18;
  if (chunk > total) chunk = total;
  const size_t slots = std::max((size_t)(DIRECT_W - 1) * n, (size_t)(DIRECT_M - 1) * chunk);
  XYZZ<F>* xyzz = nullptr;
  F* prefix = nullptr;
  hipError_t e = hipMalloc((void**)&xyzz, slots * sizeof(XYZZ<F>));
  if (e == hipSuccess) e = hipMalloc((void**)&prefix, slots * sizeof(F));
  if (e == hipSuccess) e = hipMemcpyAsync(direct, points, n * sizeof(Affine<F>), hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) {
    // window bases: slot (w * 16) * n + k
    k_table_convert<F><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(direct, (uint32_t)n);
    k_table_windows<F><<<(unsigned)((n * MsmOps<F>::LP + 255) / 256), 256, 0, st>>>(direct, (uint32_t)n, DIRECT_W, DIRECT_C, 0,
                                                                              (uint32_t)n, xyzz);
    k_table_affine<F><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(direct, (uint32_t)n, (uint32_t)(DIRECT_M * n), DIRECT_W, 0,
                                                                (uint32_t)n, xyzz, prefix, 0, 0);
    // the other fifteen multiples of every base: slots n apart
    for (size_t first = 0; first < total; first += chunk) {
      const size_t here = total - first < chunk ? total - first : chunk;
      k_direct_multiples<F><<<(unsigned)((here * MsmOps<F>::LP + 255) / 256), 256, 0, st>>>(direct, (uint32_t)total, (uint32_t)n,
                                                                                      (uint32_t)first, (uint32_t)chunk, xyzz);
      k_table_affine<F><<<(unsigned)((here + 255) / 256), 256, 0, st>>>(direct, (uint32_t)total, (uint32_t)n, DIRECT_M,
                                                                     (uint32_t)first, (uint32_t)chunk, xyzz, prefix, DIRECT_M,
                                                                     (uint32_t)n);
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess && ctx->deferred_frees) {  // the caller waits for the stream and frees the scratch
    ctx->deferred_frees->push_back(xyzz);
    ctx->deferred_frees->push_back(prefix);
    *out = direct;
    return ZG16_OK;
  }
  const hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(xyzz);
  (void)hipFree(prefix);
  if (e != hipSuccess || e2 != hipSuccess) {
    (void)hipFree(direct);
    ZG_HIP(ctx, e);
    ZG_HIP(ctx, e2);
  }
  *out = direct;
  return ZG16_OK;
}
template zg16_status msm_build_direct<Fq>(zg16_ctx*, const Affine<Fq>*, size_t, Affine<Fq>**);
template zg16_status msm_build_direct<Fq2>(zg16_ctx*, const Affine<Fq2>*, size_t, Affine<Fq2>**);

// How a set of 2^bits buckets is folded on the device: R*H buckets -> U (R) and V (H), both powers of two
// (logarithms held here), then one point per bit of the bucket number and the sum of all buckets.
struct FoldShape {
  uint32_t lR, lH;
  explicit FoldShape(uint32_t bits) {
    lR = (bits + 1) / 2;
    lH = bits - lR;
  }
  uint32_t mid() const { return (1u << lR) + (1u << lH); }
  uint32_t leaves() const { return lR + lH + 1; }
};

// Result block in page-locked host memory: the launch's 256 bytes of flags ([0] identity base met, [3] entry count),
// then sets * leaves XYZZ points (the folded bucket sets) and one unit-scalar sum per group.
static constexpr size_t PINNED_POINTS = 256;

template <class F>
zg16_status msm_launch(zg16_ctx* ctx, const MsmStreams& ms, int scratch_slot, void* pinned, const MsmJob<F>* jobs,
                       int njobs, int ngroups, MsmPending* pend, const MsmWorkspace* fixed, MsmSortShare* lend,
                       const MsmSortShare* borrow) {
  hipStream_t st = ms.main;
  if (lend) lend->valid = false;
  if (njobs > MAX_JOBS) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "too many multiexp jobs");
  if (ngroups < 1 || ngroups > MAX_JOBS) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad multiexp group count");
  JobTable<F> jt = {};
  size_t n = 0;
  int nj = 0;
  bool all_tables = true, all_direct = true;  // direct: every job has a direct table and is its own group
  uint32_t table_c = 0, table_step = 0;
  for (int j = 0; j < njobs; j++) {
    if (!jobs[j].n) continue;
    if (jobs[j].group < 0 || jobs[j].group >= ngroups) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad multiexp group");
    jt.bases[nj] = jobs[j].bases;
    jt.scalars[nj] = jobs[j].scalars;
    jt.index[nj] = jobs[j].index;
    jt.stride[nj] = (uint32_t)jobs[j].table_stride;
    jt.start[nj] = (uint32_t)n;
    jt.group[nj] = (uint32_t)jobs[j].group;
    jt.direct[nj] = jobs[j].direct;
    if (!jobs[j].direct || jobs[j].group != nj) all_direct = false;
    if (!jobs[j].table_stride || (table_c && (jobs[j].table_c != table_c || jobs[j].table_step != table_step))) all_tables = false;
    table_c = jobs[j].table_c;
    table_step = jobs[j].table_step;
    n += jobs[j].n;
    nj++;
  }
  jt.njobs = nj;
  jt.windows = 1;
  for (int j = nj; j <= MAX_JOBS; j++) jt.start[j] = (uint32_t)n;
  pend->pinned = pinned;
  pend->ngroups = (uint32_t)ngroups;
  pend->nsets = 0;
  pend->direct = 0;
  if (n == 0) return ZG16_OK;
  if (n >= 0x7fffffffu) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "multiexp of %zu terms exceeds 2^31", n);
  if (!all_tables && ngroups != 1) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "grouped multiexp needs window tables");
  if (all_direct && nj == ngroups) {
    // small queries with direct tables: two launches (see k_direct_sum)
    using PtMem = typename MsmOps<F>::PtMem;
    size_t largest = 0;
    for (int j = 0; j < nj; j++) largest = std::max<size_t>(largest, jt.start[j + 1] - jt.start[j]);
    // the launch is a latency chain (look-ups per thread, then the workgroup tree): tiny queries take half the
    // look-ups per thread
    const size_t per_thread = largest * DIRECT_W < ((size_t)1 << 17) ? DIRECT_PER_THREAD / 2 : DIRECT_PER_THREAD;
    const size_t per_block = (size_t)(256 / MsmOps<F>::LP) * per_thread;
    uint32_t blocks = (uint32_t)((largest * DIRECT_W + per_block - 1) / per_block);
    if (blocks > 1024) blocks = 1024;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // flags (256 bytes) and results lie back to back and go to the host in one copy
    const size_t o_out = PINNED_POINTS, o_part = o_out + align((size_t)nj * sizeof(XYZZ<F>));
    const size_t need = o_part + align((size_t)nj * blocks * sizeof(PtMem));
    char* base;
    if (fixed) {
      // a caller-supplied workspace: nothing is allocated (the per-kernel timers create events, so they stay off)
      if (need > fixed->bytes) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "multiexp needs %zu bytes of workspace, %zu given", need, fixed->bytes);
      base = fixed->ptr;
    } else {
      ZG_TRY(scratch_get(ctx, scratch_slot, need, (void**)&base));
    }
    uint32_t* err = (uint32_t*)base;
    PtMem* part = (PtMem*)(base + o_part);
    XYZZ<F>* wstd = (XYZZ<F>*)(base + o_out);
    ZG_HIP(ctx, hipMemsetAsync(base, 0, 256, st));
    if (fixed) {
      k_direct_sum<F><<<dim3(blocks, nj), 256, 0, st>>>(jt, direct_bias(), part, err);
      k_sum_partials<F><<<nj, 256, 0, st>>>(part, blocks, wstd);
    } else {
      KernelTimer kt(ctx, sizeof(F) > sizeof(Fq) ? K_MSM_ONES_G2 : K_MSM_ONES_G1, (double)n, st);
      k_direct_sum<F><<<dim3(blocks, nj), 256, 0, st>>>(jt, direct_bias(), part, err);
      k_sum_partials<F><<<nj, 256, 0, st>>>(part, blocks, wstd);
    }
    ZG_HIP(ctx, hipGetLastError());
    ZG_HIP(ctx, hipMemcpyAsync(pinned, base, PINNED_POINTS + (size_t)nj * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
    pend->nsets = 1;
    pend->direct = 1;
    pend->acc_kernel = sizeof(F) > sizeof(Fq) ? K_MSM_ACC0_G2 : K_MSM_ACC0_G1;
    return ZG16_OK;
  }

  if (fixed) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "a multiexp in a fixed workspace needs a direct table for every job");
  MsmShape sh;
  sh.n = (uint32_t)n;
  sh.shared = all_tables ? 1 : 0;
  sh.G = (uint32_t)ngroups;
  sh.c = sh.shared ? table_c : pick_window(n);
  sh.W = (255 + sh.c - 1) / sh.c;
  sh.B = 1u << (sh.c - 1);
  sh.f = sh.shared ? std::max(1u, std::min(table_step, sh.W)) : 1;
  sh.nb = sh.shared ? sh.G * sh.f * sh.B : sh.W * sh.B;
  jt.windows = sh.shared ? msm_table_windows(sh.c, sh.f) : 1;
  if (!sh.shared)
    for (int j = 0; j < nj; j++) jt.stride[j] = 0;
  const uint32_t nsets = sh.shared ? sh.G * sh.f : sh.W;
  const size_t max_entries = (size_t)sh.W * n;
  if (max_entries >= 0xfffffff0ull) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "multiexp too large for 32-bit entry offsets");
  if (sh.nb + sh.G > KEY_MASK) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "too many multiexp buckets");
  const FoldShape fold(sh.c - 1);
  const size_t nresult = (size_t)nsets * fold.leaves() + sh.G;
  if (nresult * sizeof(XYZZ<F>) + PINNED_POINTS > MSM_PINNED_BYTES)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "multiexp result block too large");
  const uint32_t ncount = sh.nb + sh.G;  // counters: digit buckets, then one unit-scalar bucket per group
  if (borrow) {
    if (!borrow->valid || !sh.shared || nj != 1 || sh.G != 1 || borrow->c != sh.c || borrow->f != sh.f || borrow->B != sh.B ||
        borrow->term_count != n || (uint64_t)(borrow->group + 1) * sh.nb > borrow->nb)
      return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "this multiexp cannot use the other one's sort (different width, step or terms)");
    // the entries name the lender's terms
    jt.start[0] = borrow->term_start;
    for (int j = 1; j <= MAX_JOBS; j++) jt.start[j] = borrow->term_start + (uint32_t)n;
  }

  // Coarse bins: about 256 (a workgroup's run in a bin is then a few lines long; 1024 bins measured 5 % slower at
  // 2^24 rows), as long as two bins' buckets still fit the LDS counters of the fine passes.
  SortShape ss;
  static const uint32_t want_bins = getenv("ZG16_SORT_BINS") ? (uint32_t)atoi(getenv("ZG16_SORT_BINS")) : 256;
  ss.shift = 7;
  while (((sh.nb + (1u << ss.shift) - 1) >> ss.shift) > want_bins && (4u << ss.shift) <= FINE_LDS) ss.shift++;
  while (((sh.nb + (1u << ss.shift) - 1) >> ss.shift) > SORT_MAX_COARSE) ss.shift++;
  ss.ncoarse = (sh.nb + (1u << ss.shift) - 1) >> ss.shift;
  const uint32_t nbins = ss.ncoarse + sh.G;
  ss.terms = n >= ((size_t)1 << 22) ? SORT_TERMS_LARGE : SORT_TERMS_SMALL;
  const uint32_t SORT_TERMS = ss.terms;

  // scratch layout
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t o = 0;
  const size_t o_hist = o;   o += align((size_t)ncount * 4);
  const size_t o_off = o;    o += align((size_t)(ncount + 1) * 4);
  const size_t o_cur = o;    o += align((size_t)ncount * 4);
  const size_t o_coff = o;   o += align((size_t)(nbins + 1) * 4);
  constexpr uint32_t HEAVY_GRID = 256;
  const size_t o_done = o;   o += align(HEAVY_GRID * 4);  // k_heavy: slices finished, per heavy bucket
  using PtMem = typename MsmOps<F>::PtMem;
  const size_t o_bsum = o;   o += align((size_t)sh.nb * sizeof(PtMem));  // bucket sums (identity = zero bytes)
  // [0] identity-base flag, [2] number of heavy buckets, [3] entry count; the result block follows it directly, so
  // that flags and results go to the host in one copy
  const size_t o_err = o;    o += PINNED_POINTS;
  const size_t o_zero_end = o;
  const size_t o_wstd = o;   o += align(nresult * sizeof(XYZZ<F>));  // the result block for the host
  const size_t o_rec0 = o;   o += align(max_entries * sizeof(uint2));  // coarse-sorted entries
  const size_t o_rec = o;    o += align(max_entries * sizeof(uint2));  // bucket-sorted entries
  // k_accumulate runs a fixed number of logical threads: one full round of the GPU at four waves per SIMD.
  constexpr bool is_g2 = sizeof(F) > sizeof(Fq);
  const uint32_t lanes = (uint32_t)ctx->cu_count * 4 * 64 * 4 / MsmOps<F>::LP;
  const size_t o_pts1 = o;   o += align(2 * (size_t)lanes * sizeof(PtMem));
  const size_t o_heavy = o;  o += align((size_t)sh.nb * 4);
  const size_t o_slice = o;  o += align(HEAVY_GRID * sizeof(PtMem));
  const size_t o_mid = o;    o += align((size_t)nsets * fold.mid() * sizeof(PtMem));
  const uint32_t ntiles = (ncount + SCAN_TILE - 1) / SCAN_TILE;
  constexpr uint32_t LP = MsmOps<F>::LP;
  // k_sum_range: sized so that a logical thread adds about 8 bases when half of a group's scalars equal one (a zinc
  // witness): the workgroup sum that follows costs a few additions per thread, so far fewer bases per thread waste
  // the launch on it and many more make a long serial tail (measured with the spread-form sum, same box: 4 / 8 / 12 /
  // 16 per thread -> 1.204 / 1.213 / 1.236 / 1.248 ms at 2^17 rows, flat within noise at 2^20, 52.6 vs 52.4 ms for
  // 8 vs 16 at 2^24)
  constexpr size_t ONES_PER_THREAD = 8;
  const size_t ones_div = (size_t)sh.G * 2 * (256 / LP) * ONES_PER_THREAD;
  uint32_t ones_blocks = (uint32_t)((n + ones_div - 1) / ones_div);
  if (ones_blocks > 1024) ones_blocks = 1024;
  const size_t o_tiles = o;  o += align((size_t)(ntiles + 1) * 4);
  // per-workgroup coarse histograms of the sort (bin-major) and the tile sums of their scan
  const unsigned sgrid = (unsigned)((n + 256 * SORT_TERMS - 1) / (256 * SORT_TERMS));
  const size_t n_wgh = (size_t)nbins * sgrid;
  if (n_wgh >= 0xffffffffull) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "multiexp too large for the sort's histogram scan");
  const uint32_t wg_tiles = (uint32_t)((n_wgh + SCAN_TILE - 1) / SCAN_TILE);
  const size_t o_wgh = o;    o += align((n_wgh + 1) * 4);
  const size_t o_wgt = o;    o += align((size_t)(wg_tiles + 1) * 4);
  const size_t o_part = o;   o += align((size_t)sh.G * ones_blocks * sizeof(PtMem));
  char* base;
  ZG_TRY(scratch_get(ctx, scratch_slot, o, (void**)&base));
  uint32_t* hist = (uint32_t*)(base + o_hist);
  uint32_t* off = (uint32_t*)(base + o_off);
  uint32_t* cursor = (uint32_t*)(base + o_cur);
  uint32_t* coff = (uint32_t*)(base + o_coff);
  uint32_t* err = (uint32_t*)(base + o_err);
  uint2* rec0 = (uint2*)(base + o_rec0);
  uint2* rec = (uint2*)(base + o_rec);
  PtMem* pts1 = (PtMem*)(base + o_pts1);
  PtMem* bsum = (PtMem*)(base + o_bsum);
  PtMem* mid = (PtMem*)(base + o_mid);
  XYZZ<F>* wstd = (XYZZ<F>*)(base + o_wstd);
  uint32_t* tiles = (uint32_t*)(base + o_tiles);
  uint32_t* wg_hist = (uint32_t*)(base + o_wgh);
  uint32_t* wg_tiles_sum = (uint32_t*)(base + o_wgt);
  PtMem* ones_part = (PtMem*)(base + o_part);

  ZG_HIP(ctx, hipMemsetAsync(base, 0, o_zero_end, st));  // every counter and flag, and the bucket sums
  // whose sorted entries the kernels below walk: this launch's, or one group's part of the lender's
  const uint2* walk_rec = rec;
  const uint32_t* walk_off = off;        // offsets of the nb digit buckets (and the end of the last one)
  const uint32_t* ones_off = off;        // ones_off[ones_nb + g] .. [ones_nb + g + 1]: the unit-scalar entries of group g
  uint32_t key0 = 0, ones_nb = sh.nb;
  if (borrow) {
    key0 = borrow->group * sh.nb;
    walk_rec = (const uint2*)borrow->rec;
    walk_off = borrow->off + key0;
    ones_off = borrow->off + borrow->group;
    ones_nb = borrow->nb;
    ZG_HIP(ctx, hipStreamWaitEvent(st, borrow->sorted, 0));
  } else {
    KernelTimer kt(ctx, K_MSM_DIGITS, (double)n, st);
    const unsigned fgrid = (unsigned)((max_entries + FINE_TILE - 1) / FINE_TILE);
    k_sort_count<F><<<sgrid, 256, 0, st>>>(jt, sh, ss, wg_hist);
    // exclusive scan of the (bin, workgroup) histogram, in place: every workgroup's run start in every bin
    k_scan_local<<<wg_tiles, 256, 0, st>>>(wg_hist, (uint32_t)n_wgh, wg_hist, wg_tiles_sum);
    k_scan_fixup<<<(unsigned)((n_wgh + 1 + 255) / 256), 256, 0, st>>>(wg_hist, wg_hist, (uint32_t)n_wgh, wg_tiles_sum, wg_tiles);
    k_sort_scatter<F><<<sgrid, 256, 0, st>>>(jt, sh, ss, wg_hist, (uint32_t)n_wgh, rec0, rec, coff, hist);
    k_fine_count<<<fgrid, 256, 0, st>>>(rec0, coff, ss, hist);
    k_scan_local<<<ntiles, 256, 0, st>>>(hist, ncount, off, tiles);
    k_scan_fixup<<<(ncount + 1 + 255) / 256, 256, 0, st>>>(off, cursor, ncount, tiles, ntiles);
    k_fine_scatter<<<fgrid, 256, 0, st>>>(rec0, coff, ss, cursor, rec);
  }
  if (lend && sh.shared && lend->sorted && lend->group < sh.G) {
    for (int j = 0; j < nj; j++)
      if (jt.group[j] == lend->group) {
        lend->term_start = jt.start[j];
        lend->term_count = jt.start[j + 1] - jt.start[j];
        lend->valid = true;
      }
    lend->rec = rec;
    lend->off = off;
    lend->nb = sh.nb;
    lend->c = sh.c;
    lend->f = sh.f;
    lend->B = sh.B;
    if (lend->valid) ZG_HIP(ctx, hipEventRecord(lend->sorted, st));
  }
  // Everything below sizes its grid for the largest possible entry count and reads the actual
  // counts from `off` on the device, so the host never waits in the middle of a multiexp.
  // The scalars equal to one: a plain tree sum per group, kept out of the segmented reduction.  It
  // depends only on the sort.  With a side stream it runs beside the bucket pipeline; without one it
  // follows it, so that its two latency-bound launches do not hold back the accumulation.
  auto unit_sums = [&](hipStream_t so) -> zg16_status {
    KernelTimer kt(ctx, is_g2 ? K_MSM_ONES_G2 : K_MSM_ONES_G1, (double)n, so);
    k_sum_range<F><<<dim3(ones_blocks, sh.G), 256, 0, so>>>(jt, walk_rec, ones_off, ones_nb, ones_part, err);
    k_sum_partials<F><<<sh.G, 256, 0, so>>>(ones_part, ones_blocks, wstd + (size_t)nsets * fold.leaves());
    return ZG16_OK;
  };
  if (ms.side) {
    ZG_HIP(ctx, hipEventRecord(ms.ev_fork, st));
    ZG_HIP(ctx, hipStreamWaitEvent(ms.side, ms.ev_fork, 0));
    ZG_TRY(unit_sums(ms.side));
    ZG_HIP(ctx, hipEventRecord(ms.ev_join, ms.side));
  }
  {
    // work units (additions actually performed) are added by msm_collect from the entry count
    KernelTimer kt(ctx, is_g2 ? K_MSM_ACC0_G2 : K_MSM_ACC0_G1, 0.0, st);
    k_accumulate<F><<<lanes * LP / 256, 256, 0, st>>>(jt, sh.nb, lanes, walk_rec, walk_off, key0, pts1, bsum, err);
  }
  {
    KernelTimer kt(ctx, is_g2 ? K_MSM_ACCN_G2 : K_MSM_ACCN_G1, (double)sh.nb, st);
    // lane groups per bucket: as many as still fit one resident round of the GPU, at most 8 (and a wave's worth)
    uint32_t Q = 1;
    while (Q < 8 && 2 * Q * MsmOps<F>::SL <= 64 && (size_t)sh.nb * MsmOps<F>::SL * 2 * Q <= (size_t)lanes * LP) Q *= 2;
    k_combine<F><<<(unsigned)(((size_t)sh.nb * Q * MsmOps<F>::SL + 255) / 256), 256, 0, st>>>(
        sh.nb, lanes, Q, walk_off, pts1, bsum, err + 2, (uint32_t*)(base + o_heavy));
    k_heavy<F><<<HEAVY_GRID, 256, 0, st>>>(err + 2, (const uint32_t*)(base + o_heavy), walk_off, sh.nb, lanes, pts1, bsum,
                                           (PtMem*)(base + o_slice), (uint32_t*)(base + o_done));
  }
  {
    KernelTimer kt(ctx, is_g2 ? K_MSM_REDUCE_G2 : K_MSM_REDUCE_G1, (double)sh.nb, st);
    SplitShape s1 = {};
    s1.narr = 1;
    s1.r[0] = 1u << fold.lR;
    s1.h[0] = 1u << fold.lH;
    s1.in_stride = sh.B;
    s1.out_stride = fold.mid();
    k_split_sums<F><<<dim3(fold.mid(), 1, nsets), 256, 0, st>>>(bsum, s1, mid, nullptr);
    k_bit_sums<F><<<dim3(fold.leaves(), nsets), 256, 0, st>>>(mid, fold.lR, fold.lH, fold.mid(), wstd);
  }
  ZG_HIP(ctx, hipGetLastError());
  if (ms.side)
    ZG_HIP(ctx, hipStreamWaitEvent(st, ms.ev_join, 0));
  else
    ZG_TRY(unit_sums(st));
  ZG_HIP(ctx, hipMemcpyAsync(pinned, err, PINNED_POINTS + nresult * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
  pend->nsets = nsets;
  pend->acc_kernel = is_g2 ? K_MSM_ACC0_G2 : K_MSM_ACC0_G1;
  pend->c = sh.c;
  pend->shared = sh.shared;
  pend->step = sh.f;
  return ZG16_OK;
}

// The weighted sum of one bucket set from the points k_bit_sums left: sum_p 2^p leaf[p] + leaf[bits] by Horner.
template <class F>
static XYZZ<F> fold_finish(const XYZZ<F>* leaf, const FoldShape& f) {
  const uint32_t bits = f.lR + f.lH;
  XYZZ<F> t = XYZZ<F>::inf();
  for (uint32_t p = bits; p-- > 0;) {
    if (!t.is_inf()) t = dbl(t);
    add_full(t, leaf[p]);
  }
  add_full(t, leaf[bits]);
  return t;
}

template <class F>
zg16_status msm_collect(zg16_ctx* ctx, const MsmPending* pend, XYZZ<F>* out) {
  for (uint32_t g = 0; g < pend->ngroups; g++) out[g] = XYZZ<F>::inf();
  if (!pend->nsets) return ZG16_OK;  // no terms
  const uint32_t* head = (const uint32_t*)pend->pinned;
  if (head[0]) return fail(ctx, ZG16_ERR_UNEXPECTED_IDENTITY, "multiexp consumed a base at infinity");
  if (pend->direct) {
    const XYZZ<F>* sums = (const XYZZ<F>*)((const char*)pend->pinned + PINNED_POINTS);
    for (uint32_t g = 0; g < pend->ngroups; g++) out[g] = sums[g];
    return ZG16_OK;
  }
  if (ctx->profile) ctx->kstat[pend->acc_kernel].units += head[3];
  const XYZZ<F>* hw = (const XYZZ<F>*)((const char*)pend->pinned + PINNED_POINTS);
  const FoldShape fold(pend->c - 1);
  const XYZZ<F>* ones = hw + (size_t)pend->nsets * fold.leaves();
  if (pend->shared) {
    const uint32_t f = pend->step;
    for (uint32_t g = 0; g < pend->ngroups; g++) {
      // bucket set j of the group collects the windows w = j mod f: sum_j 2^(c j) set_j, by Horner
      XYZZ<F> acc = XYZZ<F>::inf();
      for (uint32_t j = f; j-- > 0;) {
        if (!acc.is_inf())
          for (uint32_t k = 0; k < pend->c; k++) acc = dbl(acc);
        add_full(acc, fold_finish(hw + ((size_t)g * f + j) * fold.leaves(), fold));
      }
      add_full(acc, ones[g]);  // the scalars equal to one
      out[g] = acc;
    }
    return ZG16_OK;
  }
  XYZZ<F> total = XYZZ<F>::inf();
  for (int w = (int)pend->nsets - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < pend->c; k++) total = dbl(total);
    add_full(total, fold_finish(hw + (size_t)w * fold.leaves(), fold));
  }
  add_full(total, ones[0]);
  out[0] = total;
  return ZG16_OK;
}

template zg16_status msm_launch<Fq>(zg16_ctx*, const MsmStreams&, int, void*, const MsmJob<Fq>*, int, int, MsmPending*,
                                    const MsmWorkspace*, MsmSortShare*, const MsmSortShare*);
template zg16_status msm_launch<Fq2>(zg16_ctx*, const MsmStreams&, int, void*, const MsmJob<Fq2>*, int, int, MsmPending*,
                                     const MsmWorkspace*, MsmSortShare*, const MsmSortShare*);
template zg16_status msm_collect<Fq>(zg16_ctx*, const MsmPending*, XYZZ<Fq>*);
template zg16_status msm_collect<Fq2>(zg16_ctx*, const MsmPending*, XYZZ<Fq2>*);

template <class F>
static zg16_status msm_blocking(zg16_ctx* ctx, int slot, const MsmJob<F>* jobs, int njobs, XYZZ<F>* out_host) {
  MsmPending pend;
  MsmStreams ms = {ctx->stream, nullptr, nullptr, nullptr};
  ZG_TRY(msm_launch<F>(ctx, ms, slot, ctx->pinned, jobs, njobs, 1, &pend));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  profile_collect(ctx);
  return msm_collect<F>(ctx, &pend, out_host);
}

zg16_status msm_g1_device(zg16_ctx* ctx, const MsmJob<Fq>* jobs, int njobs, G1XYZZ* out_host) {
  return msm_blocking<Fq>(ctx, 1, jobs, njobs, out_host);
}
zg16_status msm_g2_device(zg16_ctx* ctx, const MsmJob<Fq2>* jobs, int njobs, G2XYZZ* out_host) {
  return msm_blocking<Fq2>(ctx, 2, jobs, njobs, out_host);
}


// ---- a multiexp that allocates nothing ------------------------------------------------------------
// zg16_msm_resident_noalloc works in one file-scope device array instead of the context's grow-only scratch
// arena: the partial sums of k_direct_sum's workgroups, the identity-base flag, the result block and (for host
// scalars) the staged scalars.  Every GPU holds its own copy of the array for the life of the process; its device
// address is looked up once per device when a context is created there, and only that address is ever handed to a
// kernel, a memset or a copy.  All calls on one GPU share the array, so calls that overlap on one GPU race.
static constexpr size_t NOALLOC_MAX_POINTS = (size_t)1 << 14;
static constexpr size_t NOALLOC_MSM_BYTES = 256 + 1024 * sizeof(MsmOps<Fq2>::PtMem) + 256;  // flags, partials, result
static constexpr size_t NOALLOC_BYTES = NOALLOC_MSM_BYTES + NOALLOC_MAX_POINTS * sizeof(Fr);
static constexpr int NOALLOC_MAX_DEVICES = 64;
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];
static unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// Called by zg16_ctx_create with the context's device current.
zg16_status msm_noalloc_init(zg16_ctx* ctx) {
  if (ctx->device < 0 || ctx->device >= NOALLOC_MAX_DEVICES) return ZG16_OK;  // the entry point reports it
  if (g_scratch_ptr[ctx->device]) return ZG16_OK;
  unsigned char* ptr = nullptr;
  ZG_HIP(ctx, hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
  g_scratch_ptr[ctx->device] = ptr;
  return ZG16_OK;
}

}  // namespace zg

using namespace zg;

extern "C" zg16_status zg16_msm_resident_noalloc(zg16_ctx* ctx, const zg16_bases* b, const uint64_t* scalars,
                                                 int on_device, uint64_t* out) {
  if (!ctx || !b || !out || (b->n && !scalars)) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_msm_resident_noalloc");
  if (b->ctx != ctx) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "handle belongs to another context");
  if (b->n > NOALLOC_MAX_POINTS)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_msm_resident_noalloc takes at most %zu points, the base set has %zu",
                NOALLOC_MAX_POINTS, b->n);
  if (b->n && !b->direct)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_msm_resident_noalloc needs a base set uploaded with window tables and small "
                                                "enough for a direct table (%zu points)", msm_direct_max());
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  unsigned char* ws = ctx->device < NOALLOC_MAX_DEVICES ? g_scratch_ptr[ctx->device] : nullptr;
  if (!ws) return fail(ctx, ZG16_ERR_DEVICE, "no static workspace on GPU %d", ctx->device);
  const Fr* d_s = (const Fr*)scalars;
  if (!on_device && b->n) {
    // through the context's page-locked block (its third megabyte; the result block below uses the first): a copy
    // from pageable memory is staged by the runtime in buffers of its own, which it may allocate on the way
    static_assert(NOALLOC_MAX_POINTS * sizeof(Fr) <= MSM_PINNED_BYTES, "the staged scalars fit one pinned block");
    void* host_stage = (char*)ctx->pinned + 2 * MSM_PINNED_BYTES;
    memcpy(host_stage, scalars, b->n * sizeof(Fr));
    Fr* staged = (Fr*)(ws + NOALLOC_MSM_BYTES);
    ZG_HIP(ctx, hipMemcpyAsync(staged, host_stage, b->n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    d_s = staged;
  }
  const MsmWorkspace fixed = {(char*)ws, NOALLOC_MSM_BYTES};
  const MsmStreams ms = {ctx->stream, nullptr, nullptr, nullptr};
  MsmPending pend;
  if (b->group == 1) {
    const MsmJob<Fq> job = {(const Affine<Fq>*)b->points, d_s, nullptr, b->n, b->n, b->table_c, 0, (const Affine<Fq>*)b->direct, b->table_step};
    ZG_TRY(msm_launch<Fq>(ctx, ms, 1, ctx->pinned, &job, 1, 1, &pend, &fixed));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    G1XYZZ res;
    ZG_TRY(msm_collect<Fq>(ctx, &pend, &res));
    const G1Affine a = to_affine(res);
    memcpy(out, &a, sizeof a);
  } else {
    const MsmJob<Fq2> job = {(const Affine<Fq2>*)b->points, d_s, nullptr, b->n, b->n, b->table_c, 0, (const Affine<Fq2>*)b->direct, b->table_step};
    ZG_TRY(msm_launch<Fq2>(ctx, ms, 2, ctx->pinned, &job, 1, 1, &pend, &fixed));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    G2XYZZ res;
    ZG_TRY(msm_collect<Fq2>(ctx, &pend, &res));
    const G2Affine a = to_affine(res);
    memcpy(out, &a, sizeof a);
  }
  return ZG16_OK;
}
