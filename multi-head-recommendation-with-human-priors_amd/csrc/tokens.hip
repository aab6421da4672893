// Ordered compaction of the live (group, token slot) pairs into per-group index lists, on the device.
//
// Reference (file:line under code/REC/): model/IDNet/hstu.py:688-690 and 814-829 select the live tokens of a prior
// category with boolean-mask indexing (`x[mask]`, a host-synchronising nonzero + gather per tensor) and branch on
// `mask.sum() == 0` on the host.  Here the live count never leaves the device: two small kernels produce, for every
// group, the ascending list of live slots translated through the static tables (query row, target row, offset).
//   count   : one workgroup per (4096-slot chunk, group): population count of the chunk's mask bytes
//   scatter : same grid; start offset = sum of the earlier chunks' counts (<= a few dozen), then an in-workgroup
//             exclusive scan (16 slots per thread, wave shuffles + one LDS hop) and the ordered writes.
// (torch.cumsum on a [4, 204800] int64 tensor plus three scatter_ calls took 0.6 ms per step at cfg1.)
#include "mhr_common.h"

namespace {

constexpr int CHUNK = 4096;   // slots per workgroup = 256 threads x 16

__global__ __launch_bounds__(256) void token_count_kernel(const uint8_t* __restrict__ mask, int n_slots, int n_chunks,
                                                          int32_t* __restrict__ chunk_cnt) {
  const int chunk = blockIdx.x, grp = blockIdx.y;
  const uint8_t* m = mask + (int64_t)grp * n_slots;
  const int base = chunk * CHUNK + threadIdx.x * 16;
  int c = 0;
  if (base + 16 <= n_slots && ((reinterpret_cast<uintptr_t>(m + base) & 15) == 0)) {
    const uint4 v = *reinterpret_cast<const uint4*>(m + base);       // mask bytes are 0 / 1
    c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  } else {
    for (int i = 0; i < 16; ++i) c += (base + i < n_slots && m[base + i]) ? 1 : 0;
  }
  c = wave_sum_i(c);
  __shared__ int s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) chunk_cnt[grp * n_chunks + chunk] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void token_scatter_kernel(const uint8_t* __restrict__ mask, const int32_t* __restrict__ q_all,
                                                            const int32_t* __restrict__ p_all, const int32_t* __restrict__ o_all,
                                                            int n_slots, int n_chunks, int tok_cap,
                                                            const int32_t* __restrict__ chunk_cnt, int32_t* __restrict__ q_idx,
                                                            int32_t* __restrict__ p_idx, int32_t* __restrict__ o_idx,
                                                            int32_t* __restrict__ n_tok, int32_t* __restrict__ tok_of_slot) {
  const int chunk = blockIdx.x, grp = blockIdx.y;
  const uint8_t* m = mask + (int64_t)grp * n_slots;
  const int32_t* cc = chunk_cnt + grp * n_chunks;
  __shared__ int s_wave[4];
  __shared__ int s_start;
  if (threadIdx.x < 64) {                                 // wave 0: start offset of this chunk (and the group total)
    int before = 0, total = 0;
    for (int c = threadIdx.x; c < n_chunks; c += 64) {
      const int v = cc[c];
      total += v;
      before += c < chunk ? v : 0;
    }
    before = wave_sum_i(before);
    total = wave_sum_i(total);
    if (threadIdx.x == 0) {
      s_start = before;
      if (chunk == 0) n_tok[grp] = min(total, tok_cap);
    }
  }
  const int base = chunk * CHUNK + threadIdx.x * 16;
  uint32_t bits = 0;                                       // bit i: slot base+i is live
#pragma unroll
  for (int i = 0; i < 16; ++i) bits |= (base + i < n_slots && m[base + i]) ? (1u << i) : 0u;
  const int mine = __popc(bits);
  // exclusive scan over the workgroup
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if ((threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  int off = s_start + incl - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += s_wave[w];
  int32_t* qd = q_idx + (int64_t)grp * tok_cap;
  int32_t* pd = p_idx + (int64_t)grp * tok_cap;
  int32_t* od = o_idx + (int64_t)grp * tok_cap;
  const int32_t* qa = q_all + (int64_t)grp * n_slots;
  if (tok_of_slot) {                                       // inverse map: list position of a live slot, -1 for the others
    int32_t* ts = tok_of_slot + (int64_t)grp * n_slots;
    int o2 = off;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (base + i < n_slots) {
        const bool on = (bits >> i) & 1u;
        ts[base + i] = (on && o2 < tok_cap) ? o2 : -1;
        o2 += on ? 1 : 0;
      }
    }
  }
  while (bits) {
    const int i = __ffs(bits) - 1;
    bits &= bits - 1;
    if (off < tok_cap) {
      qd[off] = qa[base + i];
      pd[off] = p_all[base + i];
      od[off] = o_all[base + i];
    }
    ++off;
  }
}


// ------------------------------------------------------------------------------------------
// Row maps of the compacted token lists (query-row sharing, nce_shared.hip): a ROW is a run of consecutive live tokens with
// the same query row.  Same two-launch structure as the compaction: heads per 4096-token chunk, then prefix + in-workgroup scan.
//   tok2row[t] = row of token t (0 beyond n_tok), row_q[r] = the run's query row, row_first[r] = its first token,
//   row_first[n_row] = n_tok, n_row[g] = number of rows.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t head_bits(const int32_t* __restrict__ q, int n_tok, int base) {
  uint32_t bits = 0;
  if (base < n_tok) {
    int prev = base > 0 ? q[base - 1] : -1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (base + i < n_tok) {
        const int cur = q[base + i];
        bits |= (base + i == 0 || cur != prev) ? (1u << i) : 0u;
        prev = cur;
      }
    }
  }
  return bits;
}

__global__ __launch_bounds__(256) void row_count_kernel(const int32_t* __restrict__ q_idx, const int32_t* __restrict__ n_tok_dev,
                                                        int tok_cap, int n_chunks, int32_t* __restrict__ chunk_cnt) {
  const int chunk = blockIdx.x, grp = blockIdx.y;
  const int n_tok = min(n_tok_dev[grp], tok_cap);
  int c = __popc(head_bits(q_idx + (int64_t)grp * tok_cap, n_tok, chunk * CHUNK + threadIdx.x * 16));
  c = wave_sum_i(c);
  __shared__ int s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) chunk_cnt[grp * n_chunks + chunk] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void row_scatter_kernel(const int32_t* __restrict__ q_idx, const int32_t* __restrict__ n_tok_dev,
                                                          int tok_cap, int row_cap, int n_chunks,
                                                          const int32_t* __restrict__ chunk_cnt, int32_t* __restrict__ row_q,
                                                          int32_t* __restrict__ row_first, int32_t* __restrict__ tok2row,
                                                          int32_t* __restrict__ n_row) {
  const int chunk = blockIdx.x, grp = blockIdx.y;
  const int n_tok = min(n_tok_dev[grp], tok_cap);
  const int32_t* q = q_idx + (int64_t)grp * tok_cap;
  const int32_t* cc = chunk_cnt + grp * n_chunks;
  int32_t* rq = row_q + (int64_t)grp * row_cap;
  int32_t* rf = row_first + (int64_t)grp * row_cap;
  int32_t* t2r = tok2row + (int64_t)grp * tok_cap;
  __shared__ int s_wave[4];
  __shared__ int s_start;
  if (threadIdx.x < 64) {
    int before = 0, total = 0;
    for (int c = threadIdx.x; c < n_chunks; c += 64) {
      const int v = cc[c];
      total += v;
      before += c < chunk ? v : 0;
    }
    before = wave_sum_i(before);
    total = wave_sum_i(total);
    if (threadIdx.x == 0) {
      s_start = before;
      if (chunk == 0) {
        n_row[grp] = min(total, row_cap - 1);
        if (total < row_cap) rf[total] = n_tok;
      }
    }
  }
  const int base = chunk * CHUNK + threadIdx.x * 16;
  const uint32_t bits = head_bits(q, n_tok, base);
  const int mine = __popc(bits);
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if ((threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  int rid = s_start + incl - mine - 1;                   // row of the token in front of my first one
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) rid += s_wave[w];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = base + i;
    if (t >= tok_cap) break;
    if (t < n_tok) {
      if ((bits >> i) & 1u) {
        ++rid;
        if (rid < row_cap - 1) {
          rq[rid] = q[t];
          rf[rid] = t;
        }
      }
      t2r[t] = min(rid, row_cap - 2);
    } else {
      t2r[t] = 0;
    }
  }
}

// Training-time log counters of one group (reference hstu.py:621-629: nce_samples and top-k accuracy over the tokens of
// prediction offset 0): out[0] = mean n_valid, out[1 + i] = mean(rank < ks[i]) over the live tokens with offset 0.
// Replaces about twenty-five small torch kernels (mask, casts, concatenation, GEMV, divisions) per step.  The workgroups add
// their integer partial sums into `scratch` (8 x uint64, all zero between launches); the last one to arrive - a ticket in
// scratch[7] - divides and zeroes the scratch again.  Integer sums: the result does not depend on the arrival order.
// (As ONE workgroup striding over the ~67 k tokens of a cfg1 group this was 141 us of dependent-load latency per step.)
__global__ __launch_bounds__(256) void nce_log_counters_kernel(const int32_t* __restrict__ n_valid, const int32_t* __restrict__ rank,
                                                               const int32_t* __restrict__ o_idx, const int32_t* __restrict__ n_tok_dev,
                                                               int group, int tok_cap, int k0, int k1, int k2, int k3, int k4, int n_k,
                                                               unsigned long long* __restrict__ scratch, float* __restrict__ out) {
  __shared__ unsigned long long red[4][7];
  __shared__ bool last;
  const int64_t base = (int64_t)group * tok_cap;
  const int nt = min(n_tok_dev[group], tok_cap);
  const int ks[5] = {k0, k1, k2, k3, k4};
  unsigned long long acc[7] = {0, 0, 0, 0, 0, 0, 0};          // count, sum n_valid, hits per k
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += gridDim.x * blockDim.x) {
    if (o_idx[base + t] != 0) continue;
    acc[0] += 1;
    acc[1] += (unsigned long long)max(n_valid[base + t], 0);
    const int r = rank[base + t];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[2 + i] += (i < n_k && r < ks[i]) ? 1 : 0;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < 7; ++i) red[wave][i] = acc[i];
  __syncthreads();
  if (threadIdx.x < 7) atomicAdd(scratch + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(scratch + 7, 1ull) == (unsigned long long)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < 1 + n_k) {
    const unsigned long long cnt = atomicAdd(scratch + 0, 0ull), v = atomicAdd(scratch + 1 + threadIdx.x, 0ull);   // device-scope reads
    out[threadIdx.x] = (float)((double)v / (double)(cnt > 0 ? cnt : 1));
  }
  __syncthreads();
  if (threadIdx.x < 8) scratch[threadIdx.x] = 0;              // ready for the next launch
}

// Weighted total of the per-(group, offset) mean losses and the logged partial sums, ONE launch (reference hstu.py:697-723,
// 836-870: mean per offset, horizon discount x head weight, per-segment / per-head sums).  The sums are taken in index order by
// one wave: at most a few hundred terms, and the same bits on every run.
//   total[0];  out = [ per_gp G*P | seg_all G*S | g_tot G | seg_sum S ],  per_gp = sum / max(cnt, 1) * weight
__global__ __launch_bounds__(64) void loss_reduce_kernel(const float* __restrict__ bsum, const float* __restrict__ bcnt,
                                                         const float* __restrict__ weight, int G, int P, int S,
                                                         float* __restrict__ total, float* __restrict__ out) {
  const int n = G * P, seg_len = P / S;
  float* per_gp = out;
  float* seg_all = per_gp + n;
  float* g_tot = seg_all + G * S;
  float* seg_sum = g_tot + G;
  for (int i = threadIdx.x; i < n; i += 64) per_gp[i] = bsum[i] / fmaxf(bcnt[i], 1.0f) * weight[i];
  __syncthreads();
  for (int i = threadIdx.x; i < G * S; i += 64) {
    const int g = i / S, sg = i % S;
    float a = 0.0f;
    for (int p = sg * seg_len; p < (sg + 1) * seg_len; ++p) a += per_gp[g * P + p];
    seg_all[i] = a;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 64) {
    float a = 0.0f;
    for (int sg = 0; sg < S; ++sg) a += seg_all[g * S + sg];
    g_tot[g] = a;
  }
  for (int sg = threadIdx.x; sg < S; sg += 64) {
    float a = 0.0f;
    for (int g = 0; g < G; ++g) a += seg_all[g * S + sg];
    seg_sum[sg] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f;
    for (int g = 0; g < G; ++g) a += g_tot[g];
    total[0] = a;
  }
}

// its backward: the weight of every token of bucket (g, p) = d_total * weight[g, p] / max(cnt[g, p], 1)
__global__ __launch_bounds__(64) void loss_reduce_bwd_kernel(const float* __restrict__ d_total, const float* __restrict__ bcnt,
                                                             const float* __restrict__ weight, int n, float* __restrict__ w_out) {
  const float d = d_total[0];
  for (int i = threadIdx.x; i < n; i += 64) w_out[i] = d * weight[i] / fmaxf(bcnt[i], 1.0f);
}

}  // namespace

extern "C" int mhr_loss_reduce(const float* bucket_sum, const float* bucket_cnt, const float* weight, int n_groups, int n_buckets,
                               int n_segments, float* total, float* out, void* stream) {
  MHR_REQUIRE(bucket_sum && bucket_cnt && weight && total && out, "loss_reduce: null pointer");
  MHR_REQUIRE(n_groups >= 1 && n_buckets >= 1 && n_segments >= 1 && n_buckets % n_segments == 0 && (int64_t)n_groups * n_buckets <= 65536,
              "loss_reduce: bad sizes (groups=%d buckets=%d segments=%d)", n_groups, n_buckets, n_segments);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bucket_sum, bucket_cnt, weight, n_groups, n_buckets,
                     n_segments, total, out);
  MHR_CHECK_LAUNCH("loss_reduce");
  return MHR_OK;
}

extern "C" int mhr_loss_reduce_bwd(const float* d_total, const float* bucket_cnt, const float* weight, int n, float* w_out, void* stream) {
  MHR_REQUIRE(d_total && bucket_cnt && weight && w_out && n >= 1, "loss_reduce_bwd: null pointer / bad size");
  hipLaunchKernelGGL(loss_reduce_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_total, bucket_cnt, weight, n, w_out);
  MHR_CHECK_LAUNCH("loss_reduce_bwd");
  return MHR_OK;
}

extern "C" int mhr_nce_log_counters(const int32_t* n_valid, const int32_t* rank, const int32_t* o_idx, const int32_t* n_tok_dev,
                                    int group, int tok_cap, const int32_t* ks_host, int n_k, uint64_t* scratch8, float* out,
                                    void* stream) {
  MHR_REQUIRE(n_valid && rank && o_idx && n_tok_dev && out && ks_host && scratch8, "nce_log_counters: null pointer");
  MHR_REQUIRE(group >= 0 && tok_cap > 0 && n_k >= 0 && n_k <= 5, "nce_log_counters: bad sizes (n_k=%d)", n_k);
  int k[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n_k; ++i) k[i] = ks_host[i];
  const int blocks = (int)((tok_cap + 1023) / 1024 < 128 ? (tok_cap + 1023) / 1024 : 128);
  hipLaunchKernelGGL(nce_log_counters_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_valid, rank, o_idx, n_tok_dev, group,
                     tok_cap, k[0], k[1], k[2], k[3], k[4], n_k, (unsigned long long*)scratch8, out);
  MHR_CHECK_LAUNCH("nce_log_counters");
  return MHR_OK;
}

extern "C" int mhr_token_compact(const uint8_t* mask, const int32_t* q_all, const int32_t* p_all, const int32_t* o_all,
                                 int n_groups, int n_slots, int tok_cap, int32_t* q_idx, int32_t* p_idx, int32_t* o_idx,
                                 int32_t* n_tok, int32_t* scratch, int32_t* tok_of_slot, void* stream) {
  MHR_REQUIRE(mask && q_all && p_all && o_all && q_idx && p_idx && o_idx && n_tok && scratch, "token_compact: null pointer");
  MHR_REQUIRE(n_groups >= 1 && n_groups <= 65535 && n_slots > 0 && tok_cap > 0, "token_compact: bad sizes");
  const int n_chunks = (n_slots + CHUNK - 1) / CHUNK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(token_count_kernel, dim3(n_chunks, n_groups), dim3(256), 0, s, mask, n_slots, n_chunks, scratch);
  hipLaunchKernelGGL(token_scatter_kernel, dim3(n_chunks, n_groups), dim3(256), 0, s, mask, q_all, p_all, o_all, n_slots,
                     n_chunks, tok_cap, scratch, q_idx, p_idx, o_idx, n_tok, tok_of_slot);
  MHR_CHECK_LAUNCH("token_compact");
  return MHR_OK;
}

extern "C" int mhr_row_maps(const int32_t* q_idx, const int32_t* n_tok_dev, int n_groups, int tok_cap, int row_cap,
                            int32_t* row_q, int32_t* row_first, int32_t* tok2row, int32_t* n_row, int32_t* scratch,
                            void* stream) {
  MHR_REQUIRE(q_idx && n_tok_dev && row_q && row_first && tok2row && n_row && scratch, "row_maps: null pointer");
  MHR_REQUIRE(n_groups >= 1 && n_groups <= 65535 && tok_cap > 0 && row_cap > 1, "row_maps: bad sizes");
  const int n_chunks = (tok_cap + CHUNK - 1) / CHUNK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(row_count_kernel, dim3(n_chunks, n_groups), dim3(256), 0, s, q_idx, n_tok_dev, tok_cap, n_chunks, scratch);
  hipLaunchKernelGGL(row_scatter_kernel, dim3(n_chunks, n_groups), dim3(256), 0, s, q_idx, n_tok_dev, tok_cap, row_cap, n_chunks,
                     scratch, row_q, row_first, tok2row, n_row);
  MHR_CHECK_LAUNCH("row_maps");
  return MHR_OK;
}
