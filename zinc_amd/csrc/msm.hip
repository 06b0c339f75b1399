// Bucket multiexp over BN254 G1 / G2.
//
// Replaces bellman_ce 0.3.0 `multiexp` (Pippenger over a CPU pool) as used by `create_proof` for
// the h, l, a, b_g1 and b_g2 queries (reached from zinc-vm/src/core/circuit/facade.rs:153).  The
// sum of scalar*base terms is an exact group element, so the signed-digit windows, bucket order
// and coordinate system used here give the same affine point as upstream's unsigned windows.
//
// Pipeline (all on one stream):
//   1. k_digits<COUNT>   : Montgomery scalar -> canonical -> W signed c-bit digits; histogram
//   2. k_scan            : exclusive scan of bucket sizes -> bucket offsets
//   3. k_digits<SCATTER> : counting sort of (bucket, base index, sign) entries
//   4. k_accumulate      : segmented reduction over the sorted entries.  Every lane owns a fixed
//                          chunk of K consecutive entries whatever the bucket sizes are, so a
//                          skewed witness (zinc witnesses are mostly 0/1 bits) cannot unbalance
//                          the launch.  Runs that lie inside one chunk finish a bucket; runs cut
//                          by a chunk edge go to the next level, which is reduced the same way.
//   5. k_bucket_reduce   : sum_b (b+1)*S_b per window by segment running sums + LDS tree
//   6. host              : Horner over the W window sums (W*c doublings)
// Scalars equal to one (about half of a zinc witness) skip the digit path: they are appended to
// one extra bucket through a wave-aggregated counter, which keeps them off a single hot atomic.
#include "common.cuh"

namespace zg {

static constexpr int ACC_K = 32;      // entries per lane per level
static constexpr int MAX_JOBS = 4;
static constexpr int MAX_LEVELS = 8;
static constexpr uint32_t KEY_INVALID = 0xffffffffu;
static constexpr int RED_SEG = 8;     // buckets per lane in the window reduction

template <class F>
struct JobTable {
  const Affine<F>* bases[MAX_JOBS];
  const Fr* scalars[MAX_JOBS];
  const uint32_t* index[MAX_JOBS];
  uint32_t start[MAX_JOBS + 1];  // prefix offsets of the concatenated term list
  int njobs;
};

struct MsmShape {
  uint32_t n;        // total terms
  uint32_t c;        // window bits
  uint32_t W;        // windows
  uint32_t B;        // buckets per window = 2^(c-1)
  uint32_t nb;       // W*B digit buckets; bucket nb collects the scalars equal to one
};

struct MsmScratch {
  void* mem = nullptr;
  size_t bytes = 0;
};

template <class F>
ZG_D Fr load_scalar(const JobTable<F>& jt, uint32_t i, int* job_out, uint32_t* local_out) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < MAX_JOBS; k++)
    if (k < jt.njobs && i >= jt.start[k]) j = k;
  const uint32_t local = i - jt.start[j];
  *job_out = j;
  *local_out = local;
  const uint32_t si = jt.index[j] ? jt.index[j][local] : local;
  return jt.scalars[j][si];
}

template <class F>
ZG_D Affine<F> load_base(const JobTable<F>& jt, uint32_t i) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < MAX_JOBS; k++)
    if (k < jt.njobs && i >= jt.start[k]) j = k;
  return jt.bases[j][i - jt.start[j]];
}

// bits [lo, lo+c) of a 256-bit little-endian integer (c <= 24)
ZG_D uint32_t extract_bits(const uint32_t* v, uint32_t lo, uint32_t c) {
  const uint32_t word = lo >> 5, sh = lo & 31;
  uint64_t x = v[word];
  if (word + 1 < 8) x |= (uint64_t)v[word + 1] << 32;
  return (uint32_t)(x >> sh) & ((1u << c) - 1);
}

// Wave-aggregated `atomicAdd(ctr, 1)` for the lanes that call it together.
ZG_D uint32_t wave_agg_inc(uint32_t* ctr) {
  const unsigned long long mask = __ballot(1);
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  const uint32_t rank = __popcll(mask & ((1ull << lane) - 1));
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(ctr, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  return base + rank;
}

template <class F, bool SCATTER>
__global__ __launch_bounds__(256) void k_digits(JobTable<F> jt, MsmShape sh, uint32_t* __restrict__ hist,
                                                uint32_t* __restrict__ cursor,
                                                uint32_t* __restrict__ ent, uint32_t* __restrict__ keys) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < sh.n; i += gridDim.x * blockDim.x) {
    int job;
    uint32_t local;
    const Fr s = from_mont(load_scalar(jt, i, &job, &local));
    if (s.is_zero()) continue;
    bool is_one = s.v[0] == 1;
#pragma unroll
    for (int k = 1; k < 8; k++) is_one = is_one && s.v[k] == 0;
    if (is_one) {
      if (SCATTER) {
        const uint32_t pos = wave_agg_inc(&cursor[sh.nb]);
        ent[pos] = i;
        keys[pos] = sh.nb;
      } else {
        wave_agg_inc(&hist[sh.nb]);
      }
      continue;
    }
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
        const uint32_t key = w * sh.B + d - 1;
        if (SCATTER) {
          const uint32_t pos = atomicAdd(&cursor[key], 1u);
          ent[pos] = i | (neg << 31);
          keys[pos] = key;
        } else {
          atomicAdd(&hist[key], 1u);
        }
      }
    }
  }
}

// Exclusive scan of hist[0..count) into off[0..count] (and a copy in cursor[0..count)) in three
// launches: per-workgroup scans of 4096 counters, a scan of the workgroup totals, a fix-up.
static constexpr uint32_t SCAN_TILE = 4096;

__global__ __launch_bounds__(256) void k_scan_local(const uint32_t* __restrict__ hist, uint32_t count,
                                                    uint32_t* __restrict__ off, uint32_t* __restrict__ tile_sum) {
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

// One workgroup: exclusive scan of tile_sum[0..ntiles) in place; tile_sum[ntiles] = grand total.
__global__ __launch_bounds__(1024) void k_scan_tiles(uint32_t* __restrict__ tile_sum, uint32_t ntiles) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (ntiles + 1023) / 1024;
  const uint32_t lo = tid * per, hi = (lo + per < ntiles) ? lo + per : ntiles;
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += tile_sum[i];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0;
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t v = tile_sum[i];
    tile_sum[i] = run;
    run += v;
  }
  if (tid == 1023) tile_sum[ntiles] = part[1023];
}

__global__ __launch_bounds__(256) void k_scan_fixup(uint32_t* __restrict__ off, uint32_t* __restrict__ cursor,
                                                    uint32_t count, const uint32_t* __restrict__ tile_sum,
                                                    uint32_t ntiles) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) {
    const uint32_t v = off[i] + tile_sum[i / SCAN_TILE];
    off[i] = v;
    cursor[i] = v;
  } else if (i == count) {
    off[count] = tile_sum[ntiles];
  }
}

// Slot range that bucket [s, e) of level 0 occupies at `level`.
ZG_D void level_range(uint32_t s, uint32_t e, int level, uint32_t* rs, uint32_t* re) {
  for (int l = 0; l < level; l++) {
    s = 2 * (s / ACC_K);
    e = 2 * ((e - 1) / ACC_K) + 2;
  }
  *rs = s;
  *re = e;
}

// One level of the segmented reduction.  LEVEL0: entries are (base index, sign) into the job
// table; otherwise entries are XYZZ partial sums.  Lane t owns slots [t*K, (t+1)*K).
template <class F, bool LEVEL0>
__global__ __launch_bounds__(256) void k_accumulate(JobTable<F> jt, int level, uint32_t nslots,
                                                    const uint32_t* __restrict__ keys_in,
                                                    const uint32_t* __restrict__ ent_in,
                                                    const XYZZ<F>* __restrict__ pts_in,
                                                    const uint32_t* __restrict__ off,
                                                    uint32_t* __restrict__ keys_out,
                                                    XYZZ<F>* __restrict__ pts_out,
                                                    XYZZ<F>* __restrict__ bucket_sum,
                                                    uint32_t* __restrict__ err_flag) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = t * ACC_K;
  if (lo >= nslots) return;
  const uint32_t hi = lo + ACC_K < nslots ? lo + ACC_K : nslots;
  uint32_t cur = KEY_INVALID;
  XYZZ<F> acc = XYZZ<F>::inf();
  bool first_run = true, head_used = false;
  uint32_t out_key[2] = {KEY_INVALID, KEY_INVALID};

  auto flush = [&](bool last) {
    uint32_t rs, re;
    level_range(off[cur], off[cur + 1], level, &rs, &re);
    if (rs >= lo && re <= lo + ACC_K) {
      bucket_sum[cur] = acc;
    } else {
      // cut by a chunk edge: first run -> head slot, otherwise (necessarily the last run) tail slot
      const int slot = (first_run && !head_used) ? 0 : 1;
      if (slot == 0) head_used = true;
      out_key[slot] = cur;
      pts_out[2 * (size_t)t + slot] = acc;
    }
    first_run = false;
    (void)last;
  };

  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t k = keys_in[i];
    if (k == KEY_INVALID) continue;
    if (k != cur) {
      if (cur != KEY_INVALID) flush(false);
      cur = k;
      acc = XYZZ<F>::inf();
    }
    if (LEVEL0) {
      const uint32_t e = ent_in[i];
      Affine<F> p = load_base(jt, e & 0x7fffffffu);
      if (p.is_inf()) *err_flag = 1;  // identity base consumed by a non-zero scalar
      if (e >> 31) p.y = neg(p.y);
      add_mixed(acc, p);
    } else {
      add_full(acc, pts_in[i]);
    }
  }
  if (cur != KEY_INVALID) flush(true);
  keys_out[2 * (size_t)t] = out_key[0];
  keys_out[2 * (size_t)t + 1] = out_key[1];
}

// p * k for a small unsigned k
template <class F>
ZG_D XYZZ<F> mul_small(const XYZZ<F>& p, uint32_t k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int b = 31 - __clz(k | 1); b >= 0; b--) {
    acc = dbl(acc);
    if ((k >> b) & 1) add_full(acc, p);
  }
  return acc;
}

// Workgroup tree sum through LDS; the result is valid in thread 0.
template <class F>
ZG_D void block_sum(XYZZ<F>& total, XYZZ<F>* sh /* [128] */) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t stride = 128; stride >= 1; stride >>= 1) {
    if (tid >= stride && tid < 2 * stride) sh[tid - stride] = total;
    __syncthreads();
    if (tid < stride) add_full(total, sh[tid]);
    __syncthreads();
  }
}

// Window sums, stage 1: part[w][blk] = sum over this workgroup's buckets of (b+1) * S[w*B + b].
// Grid (blocks per window, W).  A lane folds RED_SEG consecutive buckets by running sums and adds
// seg_start * (segment sum); the workgroup then tree-adds through LDS.
template <class F>
__global__ __launch_bounds__(256) void k_bucket_reduce(const XYZZ<F>* __restrict__ bucket_sum, uint32_t B,
                                                       XYZZ<F>* __restrict__ part) {
  __shared__ XYZZ<F> sh[128];
  const uint32_t w = blockIdx.y, tid = threadIdx.x;
  const XYZZ<F>* S = bucket_sum + (size_t)w * B;
  const uint32_t seg = (blockIdx.x * 256 + tid) * RED_SEG;
  XYZZ<F> acc = XYZZ<F>::inf();
  if (seg < B) {
    const uint32_t hi = seg + RED_SEG < B ? seg + RED_SEG : B;
    XYZZ<F> run = XYZZ<F>::inf();
    for (uint32_t b = hi; b-- > seg;) {
      add_full(run, S[b]);
      add_full(acc, run);
    }
    // acc = sum (b - seg + 1) S_b ; add seg * run
    if (seg) add_full(acc, mul_small(run, seg));
  }
  block_sum(acc, sh);
  if (tid == 0) part[(size_t)w * gridDim.x + blockIdx.x] = acc;
}

// out[w] = sum of part[w][0..count).  One workgroup per w.
template <class F>
__global__ __launch_bounds__(256) void k_sum_partials(const XYZZ<F>* __restrict__ part, uint32_t count,
                                                      XYZZ<F>* __restrict__ out) {
  __shared__ XYZZ<F> sh[128];
  const uint32_t w = blockIdx.x, tid = threadIdx.x;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t i = tid; i < count; i += 256) add_full(acc, part[(size_t)w * count + i]);
  block_sum(acc, sh);
  if (tid == 0) out[w] = acc;
}

// Plain sum of the bases named by ent[lo..hi) (the scalars equal to one): each lane adds a
// contiguous share, the workgroup tree-adds, one partial per workgroup.
template <class F>
__global__ __launch_bounds__(256) void k_sum_range(JobTable<F> jt, const uint32_t* __restrict__ ent, uint32_t lo,
                                                   uint32_t hi, XYZZ<F>* __restrict__ part,
                                                   uint32_t* __restrict__ err_flag) {
  __shared__ XYZZ<F> sh[128];
  const uint32_t nthreads = gridDim.x * 256, t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t per = (hi - lo + nthreads - 1) / nthreads;
  const uint32_t a = lo + t * per < hi ? lo + t * per : hi;
  const uint32_t b = a + per < hi ? a + per : hi;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t i = a; i < b; i++) {
    const Affine<F> p = load_base(jt, ent[i] & 0x7fffffffu);
    if (p.is_inf()) *err_flag = 1;
    add_mixed(acc, p);
  }
  block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

static uint32_t pick_window(size_t n) {
  const char* e = getenv("ZG16_MSM_WINDOW");
  if (e) {
    int v = atoi(e);
    if (v >= 2 && v <= 20) return (uint32_t)v;
  }
  unsigned lg = 0;
  while (((size_t)2 << lg) <= n) lg++;
  int c = (int)lg - 4;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  return (uint32_t)c;
}

template <class F>
static zg16_status msm_run(zg16_ctx* ctx, int scratch_slot, const MsmJob<F>* jobs, int njobs,
                           XYZZ<F>* out_host) {
  if (njobs > MAX_JOBS) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "too many multiexp jobs");
  JobTable<F> jt = {};
  size_t n = 0;
  int nj = 0;
  for (int j = 0; j < njobs; j++) {
    if (!jobs[j].n) continue;
    jt.bases[nj] = jobs[j].bases;
    jt.scalars[nj] = jobs[j].scalars;
    jt.index[nj] = jobs[j].index;
    jt.start[nj] = (uint32_t)n;
    n += jobs[j].n;
    nj++;
  }
  jt.njobs = nj;
  for (int j = nj; j <= MAX_JOBS; j++) jt.start[j] = (uint32_t)n;
  *out_host = XYZZ<F>::inf();
  if (n == 0) return ZG16_OK;
  if (n >= 0x7fffffffu) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "multiexp of %zu terms exceeds 2^31", n);

  MsmShape sh;
  sh.n = (uint32_t)n;
  sh.c = pick_window(n);
  sh.W = (255 + sh.c - 1) / sh.c;
  sh.B = 1u << (sh.c - 1);
  sh.nb = sh.W * sh.B;
  const size_t max_entries = (size_t)sh.W * n;
  if (max_entries >= 0xfffffff0ull) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "multiexp too large for 32-bit entry offsets");

  // level sizes (slots) for the worst case
  size_t lvl[MAX_LEVELS + 1];
  int nlevels = 0;
  lvl[0] = max_entries;
  while (true) {
    const size_t chunks = (lvl[nlevels] + ACC_K - 1) / ACC_K;
    lvl[nlevels + 1] = 2 * chunks;
    nlevels++;
    if (chunks <= 1) break;
    if (nlevels >= MAX_LEVELS) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "multiexp level overflow");
  }
  // scratch layout
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t o = 0;
  const size_t o_hist = o;   o += align((sh.nb + 1) * 4);
  const size_t o_off = o;    o += align((sh.nb + 2) * 4);
  const size_t o_cur = o;    o += align((sh.nb + 1) * 4);
  const size_t o_err = o;    o += 256;
  const size_t o_ent = o;    o += align(max_entries * 4);
  const size_t o_key0 = o;   o += align(max_entries * 4);
  size_t o_keys[MAX_LEVELS + 1], o_pts[MAX_LEVELS + 1];
  for (int l = 1; l <= nlevels; l++) {
    o_keys[l] = o; o += align(lvl[l] * 4);
    o_pts[l] = o;  o += align(lvl[l] * sizeof(XYZZ<F>));
  }
  const size_t o_bsum = o;   o += align((size_t)(sh.nb + 1) * sizeof(XYZZ<F>));
  const size_t o_wsum = o;   o += align((size_t)(sh.W + 1) * sizeof(XYZZ<F>));
  const uint32_t ntiles = (sh.nb + 1 + SCAN_TILE - 1) / SCAN_TILE;
  const uint32_t red_blocks = (sh.B + 256 * RED_SEG - 1) / (256 * RED_SEG);
  const uint32_t max_ones_blocks = 1024;
  const size_t o_tiles = o;  o += align((size_t)(ntiles + 1) * 4);
  const size_t o_part = o;   o += align((size_t)(sh.W * red_blocks + max_ones_blocks) * sizeof(XYZZ<F>));
  char* base;
  ZG_TRY(scratch_get(ctx, scratch_slot, o, (void**)&base));
  uint32_t* hist = (uint32_t*)(base + o_hist);
  uint32_t* off = (uint32_t*)(base + o_off);
  uint32_t* cursor = (uint32_t*)(base + o_cur);
  uint32_t* err = (uint32_t*)(base + o_err);
  uint32_t* ent = (uint32_t*)(base + o_ent);
  uint32_t* key0 = (uint32_t*)(base + o_key0);
  XYZZ<F>* bsum = (XYZZ<F>*)(base + o_bsum);
  XYZZ<F>* wsum = (XYZZ<F>*)(base + o_wsum);
  uint32_t* tiles = (uint32_t*)(base + o_tiles);
  XYZZ<F>* part = (XYZZ<F>*)(base + o_part);
  hipStream_t st = ctx->stream;

  ZG_HIP(ctx, hipMemsetAsync(base, 0, o_ent, st));  // hist, off, cursor, err
  ZG_HIP(ctx, hipMemsetAsync(bsum, 0, (size_t)(sh.nb + 1) * sizeof(XYZZ<F>), st));
  const unsigned dgrid = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  constexpr bool is_g2 = sizeof(F) > sizeof(Fq);
  {
    KernelTimer kt(ctx, K_MSM_DIGITS, (double)n);
    k_digits<F, false><<<dgrid, 256, 0, st>>>(jt, sh, hist, cursor, ent, key0);
  }
  {
    KernelTimer kt(ctx, K_MSM_SCAN, (double)sh.nb);
    k_scan_local<<<ntiles, 256, 0, st>>>(hist, sh.nb + 1, off, tiles);
    k_scan_tiles<<<1, 1024, 0, st>>>(tiles, ntiles);
    k_scan_fixup<<<(sh.nb + 2 + 255) / 256, 256, 0, st>>>(off, cursor, sh.nb + 1, tiles, ntiles);
  }
  {
    KernelTimer kt(ctx, K_MSM_DIGITS, (double)n);
    k_digits<F, true><<<dgrid, 256, 0, st>>>(jt, sh, hist, cursor, ent, key0);
  }
  uint32_t tail[2] = {0, 0};  // start of the ones range, total entries
  ZG_HIP(ctx, hipMemcpyAsync(tail, off + sh.nb, 8, hipMemcpyDeviceToHost, st));
  ZG_HIP(ctx, hipStreamSynchronize(st));
  // the scalars equal to one: a plain tree sum, kept out of the segmented reduction
  const uint32_t n_ones = tail[1] - tail[0];
  if (n_ones) {
    uint32_t g = (n_ones + 256 * 16 - 1) / (256 * 16);
    if (g > max_ones_blocks) g = max_ones_blocks;
    KernelTimer kt(ctx, is_g2 ? K_MSM_ONES_G2 : K_MSM_ONES_G1, (double)n_ones);
    k_sum_range<F><<<g, 256, 0, st>>>(jt, ent, tail[0], tail[1], part + (size_t)sh.W * red_blocks, err);
    k_sum_partials<F><<<1, 256, 0, st>>>(part + (size_t)sh.W * red_blocks, g, wsum + sh.W);
  } else {
    ZG_HIP(ctx, hipMemsetAsync(wsum + sh.W, 0, sizeof(XYZZ<F>), st));
  }
  // levels, sized by the actual entry count of the digit buckets
  size_t slots = tail[0];
  for (int l = 0; slots > 0; l++) {
    const size_t chunks = (slots + ACC_K - 1) / ACC_K;
    const unsigned grid = (unsigned)((chunks + 255) / 256);
    KernelTimer kt(ctx, l == 0 ? (is_g2 ? K_MSM_ACC0_G2 : K_MSM_ACC0_G1) : (is_g2 ? K_MSM_ACCN_G2 : K_MSM_ACCN_G1),
                   (double)slots);
    if (l == 0)
      k_accumulate<F, true><<<grid, 256, 0, st>>>(jt, 0, (uint32_t)slots, key0, ent, nullptr, off,
                                                  (uint32_t*)(base + o_keys[1]), (XYZZ<F>*)(base + o_pts[1]), bsum, err);
    else
      k_accumulate<F, false><<<grid, 256, 0, st>>>(jt, l, (uint32_t)slots, (uint32_t*)(base + o_keys[l]), nullptr,
                                                   (XYZZ<F>*)(base + o_pts[l]), off, (uint32_t*)(base + o_keys[l + 1]),
                                                   (XYZZ<F>*)(base + o_pts[l + 1]), bsum, err);
    if (chunks <= 1) break;
    slots = 2 * chunks;
  }
  {
    KernelTimer kt(ctx, is_g2 ? K_MSM_REDUCE_G2 : K_MSM_REDUCE_G1, (double)sh.nb);
    k_bucket_reduce<F><<<dim3(red_blocks, sh.W), 256, 0, st>>>(bsum, sh.B, part);
    k_sum_partials<F><<<sh.W, 256, 0, st>>>(part, red_blocks, wsum);
  }
  ZG_HIP(ctx, hipGetLastError());
  std::vector<XYZZ<F>> hw(sh.W + 1);
  uint32_t herr = 0;
  ZG_HIP(ctx, hipMemcpyAsync(hw.data(), wsum, (sh.W + 1) * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
  ZG_HIP(ctx, hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, st));
  ZG_HIP(ctx, hipStreamSynchronize(st));
  if (herr) return fail(ctx, ZG16_ERR_UNEXPECTED_IDENTITY, "multiexp consumed a base at infinity");
  XYZZ<F> total = XYZZ<F>::inf();
  for (int w = (int)sh.W - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < sh.c; k++) total = dbl(total);
    add_full(total, hw[w]);
  }
  add_full(total, hw[sh.W]);  // the scalars equal to one
  *out_host = total;
  return ZG16_OK;
}

zg16_status msm_g1_device(zg16_ctx* ctx, const MsmJob<Fq>* jobs, int njobs, G1XYZZ* out_host) {
  return msm_run<Fq>(ctx, 1, jobs, njobs, out_host);
}
zg16_status msm_g2_device(zg16_ctx* ctx, const MsmJob<Fq2>* jobs, int njobs, G2XYZZ* out_host) {
  return msm_run<Fq2>(ctx, 2, jobs, njobs, out_host);
}

void msm_free_scratch(zg16_ctx*) {}

}  // namespace zg
