// Full-catalog multi-head scoring, exact top-k and cross-head merge for gfx950.
//
// Reference path replaced (file:line under code/REC/): model/IDNet/hstu.py:965-1015 (fp32 normalise, [B,H,N]
// score matmul, tag / given-prior / switch -inf masks), trainer/trainer.py:724-726 (pad + history
// suppression) and evaluator/collector.py:241-282 (per-head top-k, flatten, sort, first-occurrence dedup).
// The reference writes the [B,H,N] fp32 score tensor (1.86 GB per 256-user batch at N = 454k) and sweeps it
// at least five times; here scores exist only in MFMA accumulators:
//
//   catalog_score_emit : streaming bf16 MFMA GEMM (users stationary in registers, item tiles through LDS);
//                        the epilogue compares every score with a per-row threshold tau and appends the
//                        few survivors that also pass the tag / pad / history predicates to per-row lists;
//   topk_select        : exact per-row radix select + bitonic sort of a candidate list
//                        (value descending, index ascending);
//   multihead_merge_dedup : sort of the H*k per-head winners, first-occurrence dedup, first k.
//
// The host side (ops.catalog_topk) derives tau from two strided sample passes through the SAME emit kernel
// and verifies exactness afterwards (k <= candidates <= capacity for every row), re-running flagged rows
// with tau = -inf; so results never depend on the sampling.
#include "mhr_common.h"
#include "stream_gemm.h"

namespace {

// order-preserving map float -> uint32 (larger float -> larger key)
__device__ __forceinline__ uint32_t okey(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

// ------------------------------------------------------------------------------------------
// emit
// ------------------------------------------------------------------------------------------
template <int NKS>
__global__ __launch_bounds__(256, 2) void catalog_emit_kernel(
    const bf16_t* __restrict__ users, int n_rows, int H, const bf16_t* __restrict__ items, int64_t n_items,
    int64_t item_begin, int64_t item_stride, int n_tiles, int R, int n_slices, const uint32_t* __restrict__ tag_bits,
    const uint32_t* __restrict__ row_bits, const float* __restrict__ tau, const int32_t* __restrict__ hist_ptr,
    const int64_t* __restrict__ hist_items, float* __restrict__ cand_val, int32_t* __restrict__ cand_idx,
    int32_t* __restrict__ cand_cnt, int cap) {
  using T = sg::Tile<NKS>;
  constexpr int RF = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles = smem;                                            // 3 x T::BYTES (LDS-DMA ring)

  // XCD-aware decode: workgroups with equal blockIdx % 8 share an XCD (L2); the R row tiles that stream the
  // same item slice are placed on one XCD so the slice is fetched from HBM once and re-read from that L2.
  const int w = blockIdx.x, xcd = w & 7, j = w >> 3;
  const int rt = j % R, slice = (j / R) * 8 + xcd;
  const int tps = (n_tiles + n_slices - 1) / n_slices;
  const int t0 = slice * tps, t1 = min(n_tiles, t0 + tps);
  if (t0 >= t1) return;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;

  bf16x8 frag[RF][NKS];
  int row[RF];
  float my_tau[RF];
  uint32_t my_bits[RF];
  int hp0[RF], hp1[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    row[f] = rt * 256 + wave * 64 + f * 32 + r;
    const bool live = row[f] < n_rows;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      frag[f][ks] = live ? *reinterpret_cast<const bf16x8*>(users + (int64_t)row[f] * T::DIM + ks * 16 + 8 * half) : sg::zero8();
    my_bits[f] = live ? row_bits[row[f]] : 0u;
    my_tau[f] = (live && my_bits[f] != 0u) ? tau[row[f]] : INFINITY;     // rows switched off never pass the threshold test
    hp0[f] = hp1[f] = 0;
    if (live && hist_ptr) {
      const int b = row[f] / H;
      hp0[f] = hist_ptr[b];
      hp1[f] = hist_ptr[b + 1];
    }
  }

  auto item_of = [&](int tile, int rr) -> int64_t { return item_begin + ((int64_t)tile * 32 + rr) * item_stride; };
  // item tiles stream through a 3-deep LDS-DMA ring; rows past the catalog are clamped and never emitted
  using D = sg::Dma<NKS>;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto row_ptr_for = [&](int tile) {
    return [=](int rr) -> const bf16_t* {
      const int64_t n = item_begin + ((int64_t)tile * 32 + rr) * item_stride;
      return items + (n < n_items ? n : n_items - 1) * T::DIM;
    };
  };
  sg::LaneAddr<NKS> la;
  la.init(lane);
  const int n_loc = t1 - t0;
  D::issue(tiles, row_ptr_for(t0), wv, lane);
  if (n_loc > 1) D::issue(tiles + T::BYTES, row_ptr_for(t0 + 1), wv, lane);
  sg::ring_loop<3>(n_loc, [&](auto slot_c, int i) {
    constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 3;
    const int t = t0 + i;
    if (i + 1 < n_loc) sg::wait_vmcnt<D::PW>(); else sg::wait_vmcnt<0>();
    sg::ring_barrier();
    if (i + 2 < n_loc) D::issue(tiles + nxt * T::BYTES, row_ptr_for(t + 2), wv, lane);
    f32x16 acc[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) acc[f] = sg::zero16();
    sg::mma_tile<NKS, RF>(tiles + cur * T::BYTES, la, frag, acc);

#pragma unroll
    for (int f = 0; f < RF; ++f) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float s = acc[f][g];
        if (s >= my_tau[f]) {                                // rare once tau is set: everything below is off the hot path
          const int64_t n = item_of(t, sg::crow(g, half));
          if (n < n_items && n != 0) {                       // n == 0: the pad id (trainer.py:724)
            const uint32_t tb = tag_bits ? tag_bits[n] : 0x80000000u;
            bool ok = (tb & my_bits[f]) != 0u;
            if (ok && hist_items) {                          // trainer.py:725-726: the user's own history
              int lo = hp0[f], hi = hp1[f];
              while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int64_t hv = hist_items[mid];
                if (hv < n) lo = mid + 1;
                else hi = mid;
              }
              ok = !(lo < hp1[f] && hist_items[lo] == n);
            }
            if (ok) {
              const int pos = atomicAdd(cand_cnt + row[f], 1);
              if (pos < cap) {
                cand_val[(int64_t)row[f] * cap + pos] = s;
                cand_idx[(int64_t)row[f] * cap + pos] = (int32_t)n;
              }
            }
          }
        }
      }
    }
  });
}

// ------------------------------------------------------------------------------------------
// emit, sliced lists: the fast path of ops.catalog_topk
// ------------------------------------------------------------------------------------------
// The kernel above serialises every threshold hit on global-memory round trips (tag word, history binary search, a
// returning atomic for the list slot): at cfg1 about a million hits per batch, 1.3 ms for 95 us worth of MFMA work.
// Here a hit costs two plain stores: every (row, item slice, lane half) triple has its OWN short list whose fill count
// lives in a register of the lane that owns it (no atomics), the tag words of a tile arrive in LDS with the tile (LDS-DMA),
// and the history filter moves to the select kernel, where one workgroup per row checks its few thousand candidates
// in parallel.  The DMA ring is branch-free with SGPR-base addressing (catalog padded to whole tiles) as in nce_fwd.
template <int NKS, bool STRIDED>
__global__ __launch_bounds__(256, 2) void catalog_emit_sliced_kernel(
    const bf16_t* __restrict__ users, int n_rows, const bf16_t* __restrict__ items, int64_t n_items, int64_t item_begin,
    int64_t item_stride, int n_tiles, int R, int n_slices, const uint32_t* __restrict__ tag_bits,
    const uint32_t* __restrict__ row_bits, const float* __restrict__ tau, float* __restrict__ cand_val,
    int32_t* __restrict__ cand_idx, int32_t* __restrict__ cand_cnt, int cap_s) {
  using T = sg::Tile<NKS>;
  constexpr int RF = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles = smem;                              // 3 x T::BYTES (LDS-DMA ring)
  unsigned char* words = smem + 3 * T::BYTES;               // 3 x [4 waves][64] tag words of the tile's 32 items

  const int w = blockIdx.x, xcd = w & 7, j = w >> 3;        // XCD-aware decode, as above
  const int rt = j % R, slice = (j / R) * 8 + xcd;
  const int tps = (n_tiles + n_slices - 1) / n_slices;
  const int t0 = slice * tps, t1 = min(n_tiles, t0 + tps);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;

  bf16x8 frag[RF][NKS];
  int row[RF], cnt[RF] = {0, 0};
  float my_tau[RF];
  uint32_t my_bits[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    row[f] = rt * 256 + wave * 64 + f * 32 + r;
    const bool live = row[f] < n_rows;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      frag[f][ks] = live ? *reinterpret_cast<const bf16x8*>(users + (int64_t)row[f] * T::DIM + ks * 16 + 8 * half) : sg::zero8();
    my_bits[f] = live ? row_bits[row[f]] : 0u;
    my_tau[f] = (live && my_bits[f] != 0u) ? tau[row[f]] : INFINITY;     // rows switched off never pass the threshold test
  }
  if (t0 < t1) {
    using P = sg::DmaPieces<NKS>;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    P dp;
    dp.init(wv, lane);
    const uint32_t* tag_src = tag_bits ? tag_bits : row_bits;            // no tags: any readable word (never tested)
    auto item_of = [&](int tile, int rr) -> int64_t { return item_begin + ((int64_t)tile * 32 + rr) * item_stride; };
    auto dma_tile = [&](auto slot_c, int tn) {
      constexpr int slot = decltype(slot_c)::value;
      auto f = [&](auto k_c) {
        constexpr int k = decltype(k_c)::value;
        if constexpr (STRIDED) {
          dp.template piece_rows<k>(tiles + slot * T::BYTES, [=](int rr) {
            const int64_t n = item_begin + ((int64_t)tn * 32 + rr) * item_stride;
            return items + (n < n_items ? n : n_items - 1) * T::DIM;
          }, lane);
        } else {
          dp.template piece<k>(tiles + slot * T::BYTES, reinterpret_cast<const char*>(items) + (item_begin + (int64_t)tn * 32) * T::ROW_BYTES);
        }
      };
      sg::static_for<P::PW>(f);
      const int64_t n = item_of(tn, r);
      sg::dma_words(tag_src + (tag_bits ? (n < n_items ? n : n_items - 1) : 0), words + slot * 1024 + wv * 256);
    };
    sg::LaneAddr<NKS> la;
    la.init(lane);
    sg::RowAddr<NKS> ra;
    ra.init(la, tiles);
    const int n_loc = t1 - t0, t_last = t1 - 1;
    dma_tile(std::integral_constant<int, 0>{}, t0);
    dma_tile(std::integral_constant<int, 1>{}, min(t0 + 1, t_last));
    sg::ring_loop<3>(n_loc, [&](auto slot_c, int i) {
      constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 3;
      const int t = t0 + i;
      sg::wait_vmcnt<P::PW + 1>();
      sg::ring_barrier();
      dma_tile(std::integral_constant<int, nxt>{}, min(t + 2, t_last));
      f32x16 acc[RF];
#pragma unroll
      for (int f = 0; f < RF; ++f) acc[f] = sg::zero16();
      sg::mma_tile_asm<NKS, RF, cur * T::BYTES>(ra, frag, acc);
      const uint32_t* tw = reinterpret_cast<const uint32_t*>(words + cur * 1024 + wv * 256);
#pragma unroll
      for (int f = 0; f < RF; ++f) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const float sc = acc[f][g];
          if (sc >= my_tau[f]) {                               // a hit: two stores, no round trip
            const int rr = sg::crow(g, half);
            const int64_t n = item_of(t, rr);
            const bool tag_ok = !tag_bits || (tw[rr] & my_bits[f]) != 0u;
            if (n < n_items && n != 0 && tag_ok) {             // n == 0: the pad id (trainer.py:724)
              const int pos = cnt[f]++;
              if (pos < cap_s) {
                const int64_t o = ((int64_t)row[f] * (2 * n_slices) + 2 * slice + half) * cap_s + pos;
                cand_val[o] = sc;
                cand_idx[o] = (int32_t)n;
              }
            }
          }
        }
      }
    });
    sg::wait_vmcnt<0>();
  }
  // the two lane halves of a row see different items of every tile: each half owns its own list (2 lists per slice)
#pragma unroll
  for (int f = 0; f < RF; ++f)
    if (row[f] < n_rows) cand_cnt[(int64_t)row[f] * (2 * n_slices) + 2 * slice + half] = cnt[f];
}

// ------------------------------------------------------------------------------------------
// exact top-k of a candidate list: 8-pass radix select on (okey(value) << 32 | ~index), then bitonic sort
// ------------------------------------------------------------------------------------------
constexpr int SEL_MAX_K = 1024;

// One radix-select digit step, all 256 threads: thread d owns bin d of `hist`; picks the largest digit d with
// suffix(d) = sum_{j >= d} hist[j] >= kk and returns it with kk - suffix(d + 1) (a serial scan of the 256 bins by one
// thread cost ~100 cycles of LDS latency per bin, 8 times per select: most of the kernel).
__device__ __forceinline__ void pick_digit(const uint32_t* hist, int kk, int* s_wave_tot /*[4]*/, int* s_digit, int* s_kk) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int v = (int)hist[t];
  int suf = v;                                             // inclusive suffix sum inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int x = __shfl_down(suf, o, 64);
    if (lane + o < 64) suf += x;
  }
  if (lane == 0) s_wave_tot[wv] = suf;                     // the wave's total
  __syncthreads();
  int above = 0;                                           // bins of higher waves
  for (int w2 = wv + 1; w2 < 4; ++w2) above += s_wave_tot[w2];
  const int incl = suf + above, excl = incl - v;           // suffix(d), suffix(d + 1)
  if (incl >= kk && excl < kk) {                           // exactly one thread (bin 0 catches kk > total: cannot happen, n > k)
    *s_digit = t;
    *s_kk = kk - excl;
  }
  __syncthreads();
}


__device__ __forceinline__ void bitonic_sort_desc(uint64_t* a, int n_pow2) {
  for (int size = 2; size <= n_pow2; size <<= 1) {
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < n_pow2 / 2; i += blockDim.x) {
        const int lo = (i / strd) * strd * 2 + (i % strd);
        const int hi = lo + strd;
        const bool desc = ((lo & size) == 0);
        const uint64_t x = a[lo], y = a[hi];
        if (desc ? (x < y) : (x > y)) {
          a[lo] = y;
          a[hi] = x;
        }
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ cand_val, const int32_t* __restrict__ cand_idx,
                                                          const int32_t* __restrict__ cand_cnt, int cap, int k, int kp2,
                                                          float* __restrict__ out_val, int64_t* __restrict__ out_idx,
                                                          float* __restrict__ kth_val, int32_t* __restrict__ status) {
  __shared__ uint32_t hist[256];
  __shared__ uint64_t sel[SEL_MAX_K];
  __shared__ uint64_t s_prefix;
  __shared__ int s_kk, s_nsel, s_kk2, s_digit, s_wtot[4];
  __shared__ int s_scan[256];

  const int row = blockIdx.x;
  const int cnt = cand_cnt[row];
  const int n = min(cnt, cap);
  const float* v = cand_val + (int64_t)row * cap;
  const int32_t* ix = cand_idx + (int64_t)row * cap;
  auto key_of = [&](int i) -> uint64_t { return ((uint64_t)okey(v[i]) << 32) | (uint32_t)(~(uint32_t)ix[i]); };

  uint64_t kth_key = 0;
  if (n > k) {
    if (threadIdx.x == 0) {
      s_prefix = 0;
      s_kk = k;
    }
    uint64_t mask = 0;
    for (int p = 7; p >= 0; --p) {
      hist[threadIdx.x] = 0;
      __syncthreads();
      const uint64_t prefix = s_prefix;
      for (int i = threadIdx.x; i < n; i += 256) {
        const uint64_t key = key_of(i);
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * p)) & 255], 1u);
      }
      __syncthreads();
      pick_digit(hist, s_kk, s_wtot, &s_digit, &s_kk2);
      if (threadIdx.x == 0) {
        s_kk = s_kk2;
        s_prefix = prefix | ((uint64_t)s_digit << (8 * p));
      }
      mask |= (uint64_t)0xFF << (8 * p);
      __syncthreads();
    }
    kth_key = s_prefix;
  }
  // compact the selected keys (all keys are distinct, so exactly min(n,k) qualify)
  if (threadIdx.x == 0) s_nsel = 0;
  for (int i = threadIdx.x; i < kp2; i += 256) sel[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint64_t key = key_of(i);
    if (key >= kth_key) {
      const int pos = atomicAdd(&s_nsel, 1);
      if (pos < kp2) sel[pos] = key;
    }
  }
  __syncthreads();
  const int nsel = min(s_nsel, k);
  bitonic_sort_desc(sel, kp2);

  for (int i = threadIdx.x; i < nsel; i += 256) {
    const uint64_t key = sel[i];
    out_val[(int64_t)row * k + i] = okey_inv((uint32_t)(key >> 32));
    out_idx[(int64_t)row * k + i] = (int64_t)(uint32_t)(~(uint32_t)key);
  }
  if (nsel < k) {
    // complete with (-inf, lowest item ids not in the list): among ids [0, k) at most nsel are taken
    const int need = k - nsel;
    int base = 0;   // ids handled so far
    int filled = 0;
    for (int id0 = 0; id0 < k && filled < need; id0 += 256) {
      const int id = id0 + threadIdx.x;
      int free_ = 0;
      if (id < k) {
        free_ = 1;
        for (int q = 0; q < nsel; ++q)
          if ((uint32_t)(~(uint32_t)sel[q]) == (uint32_t)id) {
            free_ = 0;
            break;
          }
      }
      s_scan[threadIdx.x] = free_;
      __syncthreads();
      for (int o = 1; o < 256; o <<= 1) {   // inclusive Hillis-Steele scan
        int add = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += add;
        __syncthreads();
      }
      const int my = filled + s_scan[threadIdx.x] - free_;   // exclusive position
      if (free_ && my < need) {
        out_val[(int64_t)row * k + nsel + my] = -INFINITY;
        out_idx[(int64_t)row * k + nsel + my] = id;
      }
      filled += s_scan[255];
      __syncthreads();
      (void)base;
    }
  }
  if (threadIdx.x == 0) {
    if (kth_val) kth_val[row] = (n >= k) ? okey_inv((uint32_t)(sel[k - 1] >> 32)) : -INFINITY;
    if (status) status[row] = cnt > cap ? 1 : 0;
  }
}

// exact top-k of a row's sliced candidate lists.  One workgroup per row: gather the slices' entries into an LDS key array
// (dropping the user's own history items: trainer.py:725-726, one binary search per candidate, all in parallel), then the
// same 8-pass radix select + bitonic sort, on LDS.  count_out = valid candidates gathered; status = 1 when a slice list
// or the LDS array overflowed (the caller re-runs such rows exactly).
constexpr int SEL_CAP = 8192;
constexpr int HIST_LDS = 2048;   // history entries kept on chip per user (longer histories are searched in global memory)
__global__ __launch_bounds__(256) void topk_select_sliced_kernel(const float* __restrict__ cand_val, const int32_t* __restrict__ cand_idx,
                                                                 const int32_t* __restrict__ cand_cnt, int n_slices, int cap_s,
                                                                 int H, const int32_t* __restrict__ hist_ptr,
                                                                 const int64_t* __restrict__ hist_items, int k, int kp2,
                                                                 float* __restrict__ out_val, int64_t* __restrict__ out_idx,
                                                                 float* __restrict__ kth_val, int32_t* __restrict__ count_out,
                                                                 int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);            // [SEL_CAP]
  uint64_t* sel = keys + SEL_CAP;                                // [kp2]
  __shared__ uint32_t hist[256];
  __shared__ uint64_t s_prefix;
  __shared__ int s_kk, s_nsel, s_n, s_over, s_kk2, s_digit, s_wtot[4];
  __shared__ int s_scan[256];

  const int row = blockIdx.x;
  const int32_t* cc = cand_cnt + (int64_t)row * n_slices;
  int hp0 = 0, hp1 = 0;
  if (hist_ptr) {
    hp0 = hist_ptr[row / H];
    hp1 = hist_ptr[row / H + 1];
  }
  if (threadIdx.x == 0) s_n = s_over = 0;
  // the user's history goes to LDS once (sorted ascending): every candidate is then checked by an on-chip binary search
  int32_t* s_hist = reinterpret_cast<int32_t*>(sel + kp2);        // [HIST_LDS]
  const int nh = hp1 - hp0;
  const bool hist_in_lds = nh <= HIST_LDS;
  if (hist_items && hist_in_lds)
    for (int i = threadIdx.x; i < nh; i += 256) s_hist[i] = (int32_t)hist_items[hp0 + i];
  __syncthreads();
  // gather: one thread per list (lists are short: a few entries each)
  for (int l = threadIdx.x; l < n_slices; l += 256) {
    int c = cc[l];
    if (c > cap_s) {
      s_over = 1;
      c = cap_s;
    }
    const int64_t base = ((int64_t)row * n_slices + l) * cap_s;
    for (int jj = 0; jj < c; ++jj) {
      const int32_t n = cand_idx[base + jj];
      bool ok = true;
      if (hist_items) {
        int lo = 0, hi = nh;
        if (hist_in_lds) {
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_hist[mid] < n) lo = mid + 1;
            else hi = mid;
          }
          ok = !(lo < nh && s_hist[lo] == n);
        } else {
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (hist_items[hp0 + mid] < (int64_t)n) lo = mid + 1;
            else hi = mid;
          }
          ok = !(lo < nh && hist_items[hp0 + lo] == (int64_t)n);
        }
      }
      if (ok) {
        const int pos = atomicAdd(&s_n, 1);
        if (pos < SEL_CAP) keys[pos] = ((uint64_t)okey(cand_val[base + jj]) << 32) | (uint32_t)(~(uint32_t)n);
      }
    }
  }
  __syncthreads();
  const int cnt = s_n;
  const int n = min(cnt, SEL_CAP);

  uint64_t kth_key = 0;
  if (n > k) {
    if (threadIdx.x == 0) {
      s_prefix = 0;
      s_kk = k;
    }
    uint64_t mask = 0;
    for (int p = 7; p >= 0; --p) {
      hist[threadIdx.x] = 0;
      __syncthreads();
      const uint64_t prefix = s_prefix;
      for (int i = threadIdx.x; i < n; i += 256) {
        const uint64_t key = keys[i];
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * p)) & 255], 1u);
      }
      __syncthreads();
      pick_digit(hist, s_kk, s_wtot, &s_digit, &s_kk2);
      if (threadIdx.x == 0) {
        s_kk = s_kk2;
        s_prefix = prefix | ((uint64_t)s_digit << (8 * p));
      }
      mask |= (uint64_t)0xFF << (8 * p);
      __syncthreads();
    }
    kth_key = s_prefix;
  }
  if (threadIdx.x == 0) s_nsel = 0;
  for (int i = threadIdx.x; i < kp2; i += 256) sel[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint64_t key = keys[i];
    if (key >= kth_key) {
      const int pos = atomicAdd(&s_nsel, 1);
      if (pos < kp2) sel[pos] = key;
    }
  }
  __syncthreads();
  const int nsel = min(s_nsel, k);
  bitonic_sort_desc(sel, kp2);
  for (int i = threadIdx.x; i < nsel; i += 256) {
    const uint64_t key = sel[i];
    out_val[(int64_t)row * k + i] = okey_inv((uint32_t)(key >> 32));
    out_idx[(int64_t)row * k + i] = (int64_t)(uint32_t)(~(uint32_t)key);
  }
  if (nsel < k) {
    // complete with (-inf, lowest item ids not in the list): among ids [0, k) at most nsel are taken
    const int need = k - nsel;
    int filled = 0;
    for (int id0 = 0; id0 < k && filled < need; id0 += 256) {
      const int id = id0 + threadIdx.x;
      int free_ = 0;
      if (id < k) {
        free_ = 1;
        for (int q = 0; q < nsel; ++q)
          if ((uint32_t)(~(uint32_t)sel[q]) == (uint32_t)id) {
            free_ = 0;
            break;
          }
      }
      s_scan[threadIdx.x] = free_;
      __syncthreads();
      for (int o = 1; o < 256; o <<= 1) {   // inclusive Hillis-Steele scan
        int add = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += add;
        __syncthreads();
      }
      const int my = filled + s_scan[threadIdx.x] - free_;   // exclusive position
      if (free_ && my < need) {
        out_val[(int64_t)row * k + nsel + my] = -INFINITY;
        out_idx[(int64_t)row * k + nsel + my] = id;
      }
      filled += s_scan[255];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    if (kth_val) kth_val[row] = (n >= k) ? okey_inv((uint32_t)(sel[k - 1] >> 32)) : -INFINITY;
    if (count_out) count_out[row] = cnt;
    if (status) status[row] = (s_over || cnt > SEL_CAP) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------
// cross-head merge + first-occurrence dedup
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void bitonic_sort_desc_dyn(uint64_t* a, int n_pow2) { bitonic_sort_desc(a, n_pow2); }

__global__ __launch_bounds__(256) void merge_dedup_kernel(const float* __restrict__ vals, const int64_t* __restrict__ idx, int H,
                                                          int k, int m_pow2, int64_t* __restrict__ out_idx,
                                                          float* __restrict__ out_val, int32_t* __restrict__ out_src,
                                                          int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* k1 = reinterpret_cast<uint64_t*>(smem);          // [m_pow2] by value
  uint64_t* k2 = k1 + m_pow2;                                // [m_pow2] by item
  uint8_t* first = reinterpret_cast<uint8_t*>(k2 + m_pow2);  // [m_pow2]
  __shared__ int s_scan[256];

  const int b = blockIdx.x;
  const int m = H * k;
  const float* v = vals + (int64_t)b * m;
  const int64_t* ix = idx + (int64_t)b * m;
  // key: value descending, then flattened (head, rank) position ascending  (collector.py:251-258)
  for (int i = threadIdx.x; i < m_pow2; i += 256)
    k1[i] = i < m ? (((uint64_t)okey(v[i]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i)) : 0ull;
  bitonic_sort_desc(k1, m_pow2);
  // group equal items: sort (item, sorted position) ascending == descending on the complement
  for (int i = threadIdx.x; i < m_pow2; i += 256) {
    if (i < m) {
      const uint32_t pos = 0xFFFFFFFFu - (uint32_t)k1[i];
      const uint64_t it = (uint64_t)(uint32_t)ix[pos];
      k2[i] = ~((it << 32) | (uint32_t)i);
    } else {
      k2[i] = 0ull;
    }
    first[i] = 0;
  }
  bitonic_sort_desc(k2, m_pow2);   // ascending in (item, position); padding (0) sinks to the end
  for (int jx = threadIdx.x; jx < m; jx += 256) {
    const uint64_t cur = ~k2[jx];
    const bool is_first = jx == 0 || ((~k2[jx - 1]) >> 32) != (cur >> 32);
    if (is_first) first[(uint32_t)cur] = 1;                  // collector.py:262-268 first occurrence wins
  }
  __syncthreads();
  // exclusive scan of `first` in value order, 256 threads x chunk
  const int chunk = (m + 255) / 256;
  int local = 0;
  for (int c = 0; c < chunk; ++c) {
    const int i = threadIdx.x * chunk + c;
    if (i < m) local += first[i];
  }
  s_scan[threadIdx.x] = local;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    int add = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
    __syncthreads();
    s_scan[threadIdx.x] += add;
    __syncthreads();
  }
  int pos_out = s_scan[threadIdx.x] - local;
  for (int c = 0; c < chunk; ++c) {
    const int i = threadIdx.x * chunk + c;
    if (i < m && first[i]) {
      if (pos_out < k) {
        const uint32_t pos = 0xFFFFFFFFu - (uint32_t)k1[i];
        out_idx[(int64_t)b * k + pos_out] = ix[pos];
        out_val[(int64_t)b * k + pos_out] = v[pos];
        out_src[(int64_t)b * k + pos_out] = (int32_t)(pos / k);
      }
      ++pos_out;
    }
  }
  if (threadIdx.x == 255 && status) status[b] = s_scan[255];
}

__global__ void hit_matrix_kernel(const int64_t* __restrict__ topk_idx, int B, int k, const int64_t* __restrict__ positives,
                                  int pos_stride, int n_pos, uint8_t* __restrict__ hit) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * k) return;
  const int b = (int)(i / k);
  const int64_t it = topk_idx[i];
  uint8_t h = 0;
  for (int p = 0; p < n_pos; ++p) h |= (positives[(int64_t)b * pos_stride + p] == it);
  hit[i] = h;
}

inline int next_pow2(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// fp32 re-score of a row's candidates (exactness against the reference's fp32 score path, hstu.py:965-979)
// ------------------------------------------------------------------------------------------
// The scorers rank on bf16 operands: |s_bf16 - s_fp32| <= 2^-8 for unit vectors, so the fp32 top-k is a subset of the
// candidates within 2^-7 of the k-th bf16 score.  Those few hundred candidates per row are re-scored here from the fp32
// rows - one wave per (row, candidate), 16 bytes per lane, wave-shuffle sum - and the final select runs on these values.
__global__ __launch_bounds__(256) void rescore_f32_kernel(const float* __restrict__ users, const float* __restrict__ items, int dim,
                                                          int64_t n_items, const int64_t* __restrict__ cand_idx, int k2,
                                                          const int32_t* __restrict__ cand_cnt, int64_t n_pairs,
                                                          float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t pr = wave; pr < n_pairs; pr += n_waves) {
    const int64_t row = pr / k2;
    const int j = (int)(pr - row * k2);
    if (j >= cand_cnt[row]) continue;
    const int64_t n = cand_idx[pr];
    float acc = 0.f;
    if (n >= 0 && n < n_items) {
      const float* u = users + row * dim;
      const float* it = items + n * dim;
      for (int c = lane * 4; c < dim; c += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(u + c), b = *reinterpret_cast<const f32x4*>(it + c);
        acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      out_val[pr] = acc;
      out_idx[pr] = (int32_t)n;
    }
  }
}

extern "C" int mhr_rescore_f32(const float* users, const float* items, int dim, int64_t n_items, const int64_t* cand_idx,
                               int n_rows, int k2, const int32_t* cand_cnt, float* out_val, int32_t* out_idx, void* stream) {
  MHR_REQUIRE(users && items && cand_idx && cand_cnt && out_val && out_idx, "rescore_f32: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && n_rows >= 0 && k2 >= 1 && n_items > 0, "rescore_f32: bad sizes (dim=%d)", dim);
  if (n_rows == 0) return MHR_OK;
  const int64_t n_pairs = (int64_t)n_rows * k2;
  hipLaunchKernelGGL(rescore_f32_kernel, dim3(mhr_grid_for(n_pairs, 4)), dim3(256), 0, (hipStream_t)stream, users, items, dim, n_items,
                     cand_idx, k2, cand_cnt, n_pairs, out_val, out_idx);
  MHR_CHECK_LAUNCH("rescore_f32");
  return MHR_OK;
}

extern "C" int mhr_catalog_score_emit(const void* users, int n_rows, int H, const void* items, int64_t n_items, int dim,
                                      int64_t item_begin, int64_t item_stride, const uint32_t* tag_bits,
                                      const uint32_t* row_bits, const float* tau, const int32_t* hist_ptr,
                                      const int64_t* hist_items, float* cand_val, int32_t* cand_idx, int32_t* cand_cnt,
                                      int cap, void* stream) {
  MHR_REQUIRE(users && items && row_bits && tau && cand_val && cand_idx && cand_cnt, "catalog_score_emit: null pointer");
  MHR_REQUIRE(n_rows > 0 && H > 0 && n_rows % H == 0, "catalog_score_emit: n_rows=%d must be a positive multiple of H=%d", n_rows, H);
  MHR_REQUIRE(dim == 16 || dim == 32 || dim == 64 || dim == 128 || dim == 256,
              "catalog_score_emit: dim=%d unsupported (16/32/64/128/256)", dim);
  MHR_REQUIRE(n_items > 0 && n_items < (1ll << 31) && item_begin >= 0 && item_stride >= 1 && cap > 0,
              "catalog_score_emit: bad item range");
  MHR_REQUIRE((hist_ptr == nullptr) == (hist_items == nullptr), "catalog_score_emit: hist_ptr/hist_items must both be set or null");
  if (item_begin >= n_items) return MHR_OK;
  const int64_t n_sel = (n_items - item_begin + item_stride - 1) / item_stride;
  const int n_tiles = (int)((n_sel + 31) / 32);
  const int R = (n_rows + 255) / 256;
  int SL = 64 / R;                       // ~512 workgroups = 2 per CU
  if (SL < 1) SL = 1;
  while (SL > 1 && 8 * (SL - 1) >= n_tiles) --SL;
  const int n_slices = 8 * SL;
  const int grid = 8 * R * SL;
  hipStream_t s = (hipStream_t)stream;
#define L_(NKS)                                                                                                          \
  {                                                                                                                      \
    size_t lds = 3 * sg::Tile<NKS>::BYTES;                                                                               \
    hipLaunchKernelGGL((catalog_emit_kernel<NKS>), dim3(grid), dim3(256), lds, s, (const bf16_t*)users, n_rows, H,       \
                       (const bf16_t*)items, n_items, item_begin, item_stride, n_tiles, R, n_slices, tag_bits, row_bits, \
                       tau, hist_ptr, hist_items, cand_val, cand_idx, cand_cnt, cap);                                    \
  }
  switch (dim) {
    case 16: L_(1); break;
    case 32: L_(2); break;
    case 64: L_(4); break;
    case 128: L_(8); break;
    default: L_(16); break;
  }
#undef L_
  MHR_CHECK_LAUNCH("catalog_score_emit");
  return MHR_OK;
}

extern "C" int mhr_catalog_score_emit_sliced(const void* users, int n_rows, const void* items, int64_t n_items,
                                             int64_t items_alloc_rows, int dim, int64_t item_begin, int64_t item_stride,
                                             const uint32_t* tag_bits, const uint32_t* row_bits, const float* tau,
                                             float* cand_val, int32_t* cand_idx, int32_t* cand_cnt, int n_slices, int cap_s,
                                             void* stream) {
  MHR_REQUIRE(users && items && row_bits && tau && cand_val && cand_idx && cand_cnt, "catalog_score_emit_sliced: null pointer");
  MHR_REQUIRE(dim == 16 || dim == 32 || dim == 64 || dim == 128 || dim == 256,
              "catalog_score_emit_sliced: dim=%d unsupported (16/32/64/128/256)", dim);
  MHR_REQUIRE(n_rows > 0 && n_items > 0 && n_items < (1ll << 31) && item_begin >= 0 && item_begin < n_items && item_stride >= 1,
              "catalog_score_emit_sliced: bad item range");
  MHR_REQUIRE(n_slices >= 8 && n_slices % 8 == 0 && cap_s >= 1, "catalog_score_emit_sliced: n_slices=%d must be a positive multiple of 8", n_slices);
  const int64_t n_sel = (n_items - item_begin + item_stride - 1) / item_stride;
  const int n_tiles = (int)((n_sel + 31) / 32);
  const int R = (n_rows + 255) / 256;
  const int grid = R * n_slices;
  // contiguous whole-tile streaming needs the catalog readable up to the end of the last tile
  const bool contiguous = item_stride == 1 && item_begin + (int64_t)n_tiles * 32 <= items_alloc_rows;
  hipStream_t s = (hipStream_t)stream;
#define L_(NKS)                                                                                                            \
  {                                                                                                                        \
    size_t lds = 3 * sg::Tile<NKS>::BYTES + 3 * 1024;                                                                      \
    if (contiguous)                                                                                                        \
      hipLaunchKernelGGL((catalog_emit_sliced_kernel<NKS, false>), dim3(grid), dim3(256), lds, s, (const bf16_t*)users,    \
                         n_rows, (const bf16_t*)items, n_items, item_begin, item_stride, n_tiles, R, n_slices, tag_bits,   \
                         row_bits, tau, cand_val, cand_idx, cand_cnt, cap_s);                                              \
    else                                                                                                                   \
      hipLaunchKernelGGL((catalog_emit_sliced_kernel<NKS, true>), dim3(grid), dim3(256), lds, s, (const bf16_t*)users,     \
                         n_rows, (const bf16_t*)items, n_items, item_begin, item_stride, n_tiles, R, n_slices, tag_bits,   \
                         row_bits, tau, cand_val, cand_idx, cand_cnt, cap_s);                                              \
  }
  switch (dim) {
    case 16: L_(1); break;
    case 32: L_(2); break;
    case 64: L_(4); break;
    case 128: L_(8); break;
    default: L_(16); break;
  }
#undef L_
  MHR_CHECK_LAUNCH("catalog_score_emit_sliced");
  return MHR_OK;
}

// Workspace queries (SURVEY.md 8b contract: no allocation inside, sizes queried): the scorers' candidate lists are the only
// caller-provided scratch whose size depends on a choice made here (how the item tiles are split into slices).
extern "C" int mhr_catalog_emit_slices(int n_rows, int64_t n_sel_items) {
  // about 512 workgroups (2 per CU), a multiple of 8 (one slice group per XCD), never more slices than item tiles
  const int64_t n_tiles = (n_sel_items + 31) / 32;
  const int R = (n_rows + 255) / 256;
  int SL = 64 / (R < 1 ? 1 : R);
  if (SL < 1) SL = 1;
  while (SL > 1 && 8 * (int64_t)(SL - 1) >= n_tiles) --SL;
  return 8 * SL;
}

extern "C" int64_t mhr_catalog_score_emit_sliced_workspace_bytes(int n_rows, int64_t n_sel_items, int cap_s) {
  if (n_rows <= 0 || n_sel_items <= 0 || cap_s <= 0) return 0;
  const int64_t lists = (int64_t)n_rows * 2 * mhr_catalog_emit_slices(n_rows, n_sel_items);
  return lists * cap_s * 8 + lists * 4;                     // cand_val f32 + cand_idx i32 per slot, cand_cnt i32 per list
}

extern "C" int64_t mhr_catalog_score_emit_workspace_bytes(int n_rows, int cap) {
  if (n_rows <= 0 || cap <= 0) return 0;
  return (int64_t)n_rows * cap * 8 + (int64_t)n_rows * 4;  // one list per row (cand_cnt must be zeroed by the caller)
}

extern "C" int mhr_topk_select_sliced(const float* cand_val, const int32_t* cand_idx, const int32_t* cand_cnt, int n_slices,
                                      int cap_s, int n_rows, int H, const int32_t* hist_ptr, const int64_t* hist_items, int k,
                                      float* out_val, int64_t* out_idx, float* kth_val, int32_t* count_out, int32_t* status,
                                      void* stream) {
  MHR_REQUIRE(cand_val && cand_idx && cand_cnt && out_val && out_idx, "topk_select_sliced: null pointer");
  MHR_REQUIRE(k >= 1 && k <= SEL_MAX_K && n_slices >= 1 && cap_s >= 1 && n_rows >= 0 && H >= 1,
              "topk_select_sliced: k=%d must be in [1,%d]", k, SEL_MAX_K);
  MHR_REQUIRE((hist_ptr == nullptr) == (hist_items == nullptr), "topk_select_sliced: hist_ptr/hist_items must both be set or null");
  if (n_rows == 0) return MHR_OK;
  const int kp2 = next_pow2(k);
  const size_t lds = (size_t)(SEL_CAP + kp2) * 8 + (size_t)HIST_LDS * 4;
  auto kern = topk_select_sliced_kernel;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(n_rows), dim3(256), lds, (hipStream_t)stream, cand_val, cand_idx, cand_cnt, n_slices, cap_s, H,
                     hist_ptr, hist_items, k, kp2, out_val, out_idx, kth_val, count_out, status);
  MHR_CHECK_LAUNCH("topk_select_sliced");
  return MHR_OK;
}

extern "C" int mhr_topk_select(const float* cand_val, const int32_t* cand_idx, const int32_t* cand_cnt, int cap, int n_rows,
                               int k, float* out_val, int64_t* out_idx, float* kth_val, int32_t* status, void* stream) {
  MHR_REQUIRE(cand_val && cand_idx && cand_cnt && out_val && out_idx, "topk_select: null pointer");
  MHR_REQUIRE(k >= 1 && k <= SEL_MAX_K && cap >= 1 && n_rows >= 0, "topk_select: k=%d must be in [1,%d]", k, SEL_MAX_K);
  if (n_rows == 0) return MHR_OK;
  hipLaunchKernelGGL(topk_select_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, cand_val, cand_idx, cand_cnt, cap,
                     k, next_pow2(k), out_val, out_idx, kth_val, status);
  MHR_CHECK_LAUNCH("topk_select");
  return MHR_OK;
}

extern "C" int mhr_multihead_merge_dedup(const float* vals, const int64_t* idx, int B, int H, int k, int64_t* out_idx,
                                         float* out_val, int32_t* out_src, int32_t* status, void* stream) {
  MHR_REQUIRE(vals && idx && out_idx && out_val && out_src, "multihead_merge_dedup: null pointer");
  MHR_REQUIRE(B >= 0 && H >= 1 && k >= 1, "multihead_merge_dedup: bad sizes");
  const int m = H * k, mp = next_pow2(m);
  MHR_REQUIRE(mp <= 8192, "multihead_merge_dedup: H*k=%d too large (<= 8192)", m);
  if (B == 0) return MHR_OK;
  size_t lds = (size_t)mp * 8 * 2 + mp;
  auto kern = merge_dedup_kernel;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, vals, idx, H, k, mp, out_idx, out_val, out_src, status);
  MHR_CHECK_LAUNCH("multihead_merge_dedup");
  return MHR_OK;
}

extern "C" int mhr_hit_matrix(const int64_t* topk_idx, int B, int k, const int64_t* positives, int pos_stride, int n_pos,
                              uint8_t* hit, void* stream) {
  MHR_REQUIRE(topk_idx && positives && hit, "hit_matrix: null pointer");
  MHR_REQUIRE(B >= 0 && k >= 1 && n_pos >= 0 && pos_stride >= n_pos, "hit_matrix: bad sizes");
  if (B == 0) return MHR_OK;
  const int64_t n = (int64_t)B * k;
  hipLaunchKernelGGL(hit_matrix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, topk_idx, B, k,
                     positives, pos_stride, n_pos, hit);
  MHR_CHECK_LAUNCH("hit_matrix");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// The decode's exactness bookkeeping, one launch per decision instead of 7-12 elementwise launches each (the eval step is a
// chain of few-microsecond launches around its three big ones, and every host read of a flag drains the queue):
//   pick_tau:      emit threshold per row from the two sample passes (kth2 if both selects were clean and kth2 is finite, else
//                  kth1 if clean, else -inf)
//   flag:          rows whose candidate lists cannot be trusted (list overflow, or fewer than k_min candidates above a finite
//                  threshold for an admissible row); any[slot] = 1 if there is one (written, not or-ed: one workgroup)
//   margin_count:  cnt[r] = finite entries of the sorted list bv[r, :] that are >= bv[r, kk - 1] - margin (a prefix)
//   uncertified:   rows whose margin set reaches below the emit threshold (tau = -inf: exact list, nothing below; NaN / +inf: no
//                  threshold known, uncertified) or fills all k2 slots; any[slot] likewise
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void topk_pick_tau_kernel(const float* __restrict__ kth1, const float* __restrict__ kth2,
                                                            const int32_t* __restrict__ st1, const int32_t* __restrict__ st2, int n,
                                                            float* __restrict__ tau) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const bool ok = st1[r] == 0 && st2[r] == 0;
  const float k2 = kth2[r];
  tau[r] = (ok && isfinite(k2)) ? k2 : (ok ? kth1[r] : -INFINITY);
}

__device__ __forceinline__ void block_any_write(bool mine, int32_t* __restrict__ any_out) {
  __shared__ int hit;
  if (threadIdx.x == 0) hit = 0;
  __syncthreads();
  if (mine) atomicOr(&hit, 1);
  __syncthreads();
  if (threadIdx.x == 0) any_out[0] = hit;
}

__global__ __launch_bounds__(1024) void topk_flag_kernel(const int32_t* __restrict__ st, const int32_t* __restrict__ cnt,
                                                         const int32_t* __restrict__ row_bits, const float* __restrict__ tau, int k_min,
                                                         int n, uint8_t* __restrict__ flagged, int32_t* __restrict__ any_out) {
  bool mine = false;
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    const bool f = st[r] != 0 || (cnt[r] < k_min && row_bits[r] != 0 && isfinite(tau[r]));
    flagged[r] = f ? 1 : 0;
    mine |= f;
  }
  block_any_write(mine, any_out);
}

__global__ __launch_bounds__(256) void topk_margin_count_kernel(const float* __restrict__ bv, int n, int k2, int kk, float margin,
                                                                int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float* row = bv + (int64_t)r * k2;
  const float lo = row[kk - 1] - margin;
  int c = 0;
  for (int j = lane; j < k2; j += 64) {
    const float v = row[j];
    c += (v >= lo && isfinite(v)) ? 1 : 0;
  }
  c = wave_sum_i(c);
  if (lane == 0) cnt[r] = c;
}

__global__ __launch_bounds__(1024) void topk_uncertified_kernel(const int32_t* __restrict__ cnt, const float* __restrict__ bv, int k2, int kk,
                                                                const float* __restrict__ tau, int list_can_fill, float margin, int n,
                                                                uint8_t* __restrict__ full, int32_t* __restrict__ any_out) {
  bool mine = false;
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    const float kth = bv[(int64_t)r * k2 + kk - 1], t = tau[r];
    // t = -inf: every admissible item was a candidate (exact list); anything else must lie BELOW the margin band - a NaN or
    // +inf threshold (a scorer that reports none) certifies nothing
    const bool f = (cnt[r] >= k2 && list_can_fill) || (isfinite(kth) && t != -INFINITY && !(kth - margin >= t));
    full[r] = f ? 1 : 0;
    mine |= f;
  }
  block_any_write(mine, any_out);
}
}  // namespace

extern "C" int mhr_topk_pick_tau(const float* kth1, const float* kth2, const int32_t* st1, const int32_t* st2, int n_rows, float* tau,
                                 void* stream) {
  MHR_REQUIRE(kth1 && kth2 && st1 && st2 && tau && n_rows > 0, "topk_pick_tau: null pointer / bad size");
  hipLaunchKernelGGL(topk_pick_tau_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, kth1, kth2, st1, st2, n_rows, tau);
  MHR_CHECK_LAUNCH("topk_pick_tau");
  return MHR_OK;
}

extern "C" int mhr_topk_flag(const int32_t* status, const int32_t* count, const int32_t* row_bits, const float* tau, int k_min, int n_rows,
                             uint8_t* flagged, int32_t* any_out, void* stream) {
  MHR_REQUIRE(status && count && row_bits && tau && flagged && any_out && n_rows > 0, "topk_flag: null pointer / bad size");
  hipLaunchKernelGGL(topk_flag_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, status, count, row_bits, tau, k_min, n_rows, flagged,
                     any_out);
  MHR_CHECK_LAUNCH("topk_flag");
  return MHR_OK;
}

extern "C" int mhr_topk_margin_count(const float* sorted_vals, int n_rows, int k2, int kk, float margin, int32_t* count, void* stream) {
  MHR_REQUIRE(sorted_vals && count && n_rows > 0 && k2 > 0 && kk >= 1 && kk <= k2, "topk_margin_count: null pointer / bad sizes");
  hipLaunchKernelGGL(topk_margin_count_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, sorted_vals, n_rows, k2, kk, margin,
                     count);
  MHR_CHECK_LAUNCH("topk_margin_count");
  return MHR_OK;
}

extern "C" int mhr_topk_uncertified(const int32_t* count, const float* sorted_vals, int k2, int kk, const float* tau, int list_can_fill,
                                    float margin, int n_rows, uint8_t* full, int32_t* any_out, void* stream) {
  MHR_REQUIRE(count && sorted_vals && tau && full && any_out && n_rows > 0 && k2 > 0 && kk >= 1 && kk <= k2,
              "topk_uncertified: null pointer / bad sizes");
  hipLaunchKernelGGL(topk_uncertified_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, count, sorted_vals, k2, kk, tau, list_can_fill,
                     margin, n_rows, full, any_out);
  MHR_CHECK_LAUNCH("topk_uncertified");
  return MHR_OK;
}
