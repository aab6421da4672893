// Item-embedding kernels: row gather (+ fused position add), dense atomic scatter-add,
// deterministic sparse segment-sum, AdamW over the table (sparse gradient, dense update) and over
// the flat dense-parameter buffer.  All HBM-bound: one wave per 1 KiB-class row, 16 B per lane.
//
// Reference ops replaced (file:line under code/REC/): nn.Embedding forward + position add
// model/IDNet/hstu.py:413,637-643,670,752,883; embedding_dense_backward (autograd);
// optimizer step trainer/trainer.py:292-299,532.
#include "mhr_common.h"

// ------------------------------------------------------------------------------------------
// gather
// ------------------------------------------------------------------------------------------
// ids outside [0, n_rows) seen by the gather kernels since the last reset (mhr_bad_id_count)
__device__ unsigned int g_bad_ids = 0;

template <typename OT, typename XT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int64_t n_rows, int dim,
                                                          const int64_t* __restrict__ ids, int64_t n_ids,
                                                          OT* __restrict__ out, const float* __restrict__ pos,
                                                          int seq_len, int window_len, XT* __restrict__ x_out, int64_t n_x_ids) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  constexpr int U = 4;  // rows in flight per wave
  for (int64_t r0 = wave * U; r0 < n_ids; r0 += n_waves * U) {
    int64_t id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t r = r0 + u;
      int64_t v = r < n_ids ? ids[r] : 0;
      if (v < 0 || v >= n_rows) {                          // nn.Embedding raises here; a kernel cannot: count, clamp, and let the
        if (lane == 0) atomicAdd(&g_bad_ids, 1u);          // host ask (mhr_bad_id_count) where it synchronises anyway
        v = v < 0 ? 0 : n_rows - 1;
      }
      id[u] = v;
    }
    for (int c = lane * 4; c < dim; c += 256) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r0 + u < n_ids) v[u] = *reinterpret_cast<const f32x4*>(table + id[u] * dim + c);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t r = r0 + u;
        if (r >= n_ids) continue;
        if (out) Vec4IO<OT>::store(out + r * dim + c, v[u]);
        if (x_out && r < n_x_ids) {
          int64_t b = r / window_len;
          int l = (int)(r - b * window_len);
          if (l < seq_len) {
            f32x4 p = *reinterpret_cast<const f32x4*>(pos + (int64_t)l * dim + c);
            Vec4IO<XT>::store(x_out + (b * seq_len + l) * dim + c, v[u] + p);
          }
        }
      }
    }
  }
}

// One launch for everything a training step reads from the item table (reference hstu.py:637-643 item windows + position
// add, 670-672 / 752-754 negative pools gathered and L2-normalised): workgroups [0, grid_items) run the item-window gather
// above on ids[0, n_item_ids); workgroups [grid_items, ...) turn ids[n_item_ids, n_ids) into normalised bf16 rows + their norms
// (one wave per row, the row in registers: the fp32 copy of the negatives' rows is never written).  The normalisation is
// l2norm_kernel's arithmetic in its order (csrc/norm.hip), so the values are bitwise gather + mhr_l2norm_rows.
template <int NC>
__global__ __launch_bounds__(256) void gather_step_kernel(const float* __restrict__ table, int64_t n_rows, int dim,
                                                          const int64_t* __restrict__ ids, int64_t n_ids, int64_t n_item_ids,
                                                          float* __restrict__ out, const float* __restrict__ pos, int seq_len,
                                                          int window_len, float* __restrict__ x_out, bf16_t* __restrict__ neg_out,
                                                          float* __restrict__ neg_norms, int grid_items) {
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x < grid_items) {
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)grid_items * 4;
    constexpr int U = 4;
    for (int64_t r0 = wave * U; r0 < n_item_ids; r0 += n_waves * U) {
      int64_t id[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t v = r0 + u < n_item_ids ? ids[r0 + u] : 0;
        if (v < 0 || v >= n_rows) {
          if (lane == 0) atomicAdd(&g_bad_ids, 1u);
          v = v < 0 ? 0 : n_rows - 1;
        }
        id[u] = v;
      }
      for (int c = lane * 4; c < dim; c += 256) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (r0 + u < n_item_ids) v[u] = *reinterpret_cast<const f32x4*>(table + id[u] * dim + c);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = r0 + u;
          if (r >= n_item_ids) continue;
          *reinterpret_cast<f32x4*>(out + r * dim + c) = v[u];
          const int64_t b = r / window_len;
          const int l = (int)(r - b * window_len);
          if (l < seq_len) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (int64_t)l * dim + c);
            *reinterpret_cast<f32x4*>(x_out + (b * seq_len + l) * dim + c) = v[u] + p;
          }
        }
      }
    }
    return;
  }
  const int64_t n_neg = n_ids - n_item_ids;
  const int64_t wave0 = (int64_t)((int)blockIdx.x - grid_items) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)((int)gridDim.x - grid_items) * 4;
  constexpr int U = 2;                                     // rows in flight per wave
  for (int64_t r0 = wave0 * U; r0 < n_neg; r0 += n_waves * U) {
    f32x4 v[U][NC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t id = r0 + u < n_neg ? ids[n_item_ids + r0 + u] : 0;
      if (id < 0 || id >= n_rows) {
        if (lane == 0) atomicAdd(&g_bad_ids, 1u);
        id = id < 0 ? 0 : n_rows - 1;
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane * 4 + i * 256;
        v[u][i] = c < dim ? *reinterpret_cast<const f32x4*>(table + id * dim + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + u >= n_neg) continue;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) s += v[u][i][k] * v[u][i][k];
      const float nrm = sqrtf(wave_sum(s));
      const float inv = 1.0f / nrm;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < dim) {
          f32x4 y = {v[u][i][0] * inv, v[u][i][1] * inv, v[u][i][2] * inv, v[u][i][3] * inv};
          Vec4IO<bf16_t>::store(neg_out + (r0 + u) * dim + c, y);
        }
      }
      if (lane == 0) neg_norms[r0 + u] = nrm;
    }
  }
}

extern "C" int mhr_embedding_gather_step(const float* table, int64_t n_rows, int dim, const int64_t* ids, int64_t n_ids,
                                         int64_t n_item_ids, float* rows_out, const float* pos_table, int seq_len, int window_len,
                                         float* x_out, void* neg_out, float* neg_norms, void* stream) {
  MHR_REQUIRE(table && ids && rows_out && pos_table && x_out, "embedding_gather_step: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048, "embedding_gather_step: dim=%d unsupported (multiple of 4, <= 2048)", dim);
  MHR_REQUIRE(n_rows > 0 && n_item_ids >= 0 && n_ids >= n_item_ids && window_len > 0 && seq_len > 0 && seq_len <= window_len &&
              n_item_ids % window_len == 0, "embedding_gather_step: bad sizes (n_ids=%lld n_item_ids=%lld window=%d seq=%d)",
              (long long)n_ids, (long long)n_item_ids, window_len, seq_len);
  MHR_REQUIRE(n_ids == n_item_ids || (neg_out && neg_norms), "embedding_gather_step: negative ids without neg_out / neg_norms");
  if (n_ids == 0) return MHR_OK;
  const int grid_items = n_item_ids ? mhr_grid_for(n_item_ids, 16) : 0;
  const int grid_negs = n_ids > n_item_ids ? mhr_grid_for(n_ids - n_item_ids, 8) : 0;
  hipStream_t s = (hipStream_t)stream;
#define LG(NC)                                                                                                              \
  hipLaunchKernelGGL((gather_step_kernel<NC>), dim3(grid_items + grid_negs), dim3(256), 0, s, table, n_rows, dim, ids, n_ids, \
                     n_item_ids, rows_out, pos_table, seq_len, window_len, x_out, (bf16_t*)neg_out, neg_norms, grid_items)
  const int nc = (dim + 255) / 256;
  if (nc <= 1) LG(1);
  else if (nc <= 2) LG(2);
  else if (nc <= 4) LG(4);
  else LG(8);
#undef LG
  MHR_CHECK_LAUNCH("embedding_gather_step");
  return MHR_OK;
}

unsigned int* mhr_bad_id_counter_addr() {
  static unsigned int* addr = nullptr;
  if (!addr && hipGetSymbolAddress((void**)&addr, HIP_SYMBOL(g_bad_ids)) != hipSuccess) addr = nullptr;
  return addr;
}

extern "C" int mhr_bad_id_count(int64_t* host_count, int reset) {
  MHR_REQUIRE(host_count, "bad_id_count: null pointer");
  unsigned int c = 0;
  MHR_REQUIRE(hipMemcpyFromSymbol(&c, HIP_SYMBOL(g_bad_ids), sizeof(c)) == hipSuccess, "bad_id_count: hipMemcpyFromSymbol failed");
  *host_count = c;
  if (reset && c) {
    const unsigned int z = 0;
    MHR_REQUIRE(hipMemcpyToSymbol(HIP_SYMBOL(g_bad_ids), &z, sizeof(z)) == hipSuccess, "bad_id_count: hipMemcpyToSymbol failed");
  }
  return MHR_OK;
}

extern "C" int mhr_embedding_gather_fwd(const float* table, int64_t n_rows, int dim, const int64_t* ids, int64_t n_ids,
                                        void* out, int out_dtype, const float* pos_table, int seq_len, int window_len,
                                        void* x_out, int x_dtype, int64_t n_x_ids, void* stream) {
  MHR_REQUIRE(table && ids, "embedding_gather_fwd: null table/ids");
  MHR_REQUIRE(out || x_out, "embedding_gather_fwd: no output requested");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0, "embedding_gather_fwd: dim=%d must be a positive multiple of 4", dim);
  MHR_REQUIRE(n_rows > 0 && n_ids >= 0, "embedding_gather_fwd: bad sizes");
  if (x_out) {
    const int64_t nx = (n_x_ids <= 0 || n_x_ids > n_ids) ? n_ids : n_x_ids;
    MHR_REQUIRE(pos_table && window_len > 0 && seq_len > 0 && seq_len <= window_len && nx % window_len == 0,
                "embedding_gather_fwd: x_out needs pos_table and n_x_ids %% window_len == 0 (n_x_ids=%lld window=%d seq=%d)",
                (long long)nx, window_len, seq_len);
  }
  if (n_ids == 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(n_ids, 16);
  if (window_len <= 0) window_len = 1;
  if (n_x_ids <= 0 || n_x_ids > n_ids) n_x_ids = n_ids;
#define LAUNCH(OT, XT)                                                                                     \
  hipLaunchKernelGGL((gather_rows_kernel<OT, XT>), dim3(grid), dim3(256), 0, s, table, n_rows, dim, ids,   \
                     n_ids, (OT*)out, pos_table, seq_len, window_len, (XT*)x_out, n_x_ids)
  bool ob = out && out_dtype == MHR_BF16, xb = x_out && x_dtype == MHR_BF16;
  if (ob && xb) LAUNCH(bf16_t, bf16_t);
  else if (ob) LAUNCH(bf16_t, float);
  else if (xb) LAUNCH(float, bf16_t);
  else LAUNCH(float, float);
#undef LAUNCH
  MHR_CHECK_LAUNCH("embedding_gather_fwd");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// dense scatter-add with float atomics (each wave-instruction adds 256 contiguous bytes of one row)
// ------------------------------------------------------------------------------------------
template <typename GT>
__global__ __launch_bounds__(256) void scatter_add_kernel(const GT* __restrict__ g, const int64_t* __restrict__ ids,
                                                          int64_t n_ids, float* __restrict__ gt, int64_t n_rows, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = wave; r < n_ids; r += n_waves) {
    int64_t id = ids[r];
    if (id < 0 || id >= n_rows) continue;
    for (int c = lane; c < dim; c += 64) atomicAdd(gt + id * dim + c, (float)g[r * dim + c]);
  }
}

extern "C" int mhr_embedding_scatter_add_bwd(const void* grad_rows, int grad_dtype, const int64_t* ids, int64_t n_ids,
                                             float* grad_table, int64_t n_rows, int dim, void* stream) {
  MHR_REQUIRE(grad_rows && ids && grad_table, "embedding_scatter_add_bwd: null pointer");
  MHR_REQUIRE(dim > 0 && n_rows > 0, "embedding_scatter_add_bwd: bad sizes");
  if (n_ids == 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(n_ids, 4);
  if (grad_dtype == MHR_BF16)
    hipLaunchKernelGGL((scatter_add_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)grad_rows, ids, n_ids,
                       grad_table, n_rows, dim);
  else
    hipLaunchKernelGGL((scatter_add_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)grad_rows, ids, n_ids,
                       grad_table, n_rows, dim);
  MHR_CHECK_LAUNCH("embedding_scatter_add_bwd");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// deterministic sparse segment sum over sorted ids
// ------------------------------------------------------------------------------------------
// One wave per CHUNK of 32 consecutive sorted positions (not per segment: under Zipf-distributed ids one item can own
// thousands of rows and a wave per segment serialises them - measured 1.1 ms of a 21 ms step).  Runs of equal ids
// inside a chunk are summed in registers (in list order) and stored once.  A run that is cut by chunk borders leaves one
// partial sum per chunk - at the run's head in its first chunk, at the chunk's first position in every later one (a
// position no other wave writes) - and a second pass adds a run's partials to its head IN CHUNK ORDER and zeroes them:
// no atomics, so the sum of a hot id (hundreds of duplicates per step under Zipf ids) has one fixed order and the
// data-parallel replicas, which each run this reduction on the same exchanged list, stay bitwise identical.
constexpr int SEG_CHUNK = 32;

// Latency, not bandwidth, bounded the first form of this kernel (one dependent perm load and one row load at a time per wave:
// 0.57 TB/s on the 245 k-row list of an 8-rank exchange): a chunk's 32 ids and source-row indices now arrive with ONE coalesced
// load per wave and are broadcast lane by lane (v_readlane), and the rows are fetched EIGHT positions at a time before they are
// added - in list order, from zero, exactly as before, so the sums keep their bits.
__device__ __forceinline__ int64_t readlane64(int64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <typename AT, typename BT>
__global__ __launch_bounds__(256) void segment_sum_kernel(const int64_t* __restrict__ sorted_ids,
                                                          const int64_t* __restrict__ perm, int64_t n_ids,
                                                          const AT* __restrict__ ga, int64_t n_a,
                                                          const BT* __restrict__ gb, int64_t n_b,
                                                          const float* __restrict__ xg, int seq_len, int window_len,
                                                          float* __restrict__ out_rows, int32_t* __restrict__ row_slot,
                                                          int64_t n_rows, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t n_chunks = (n_ids + SEG_CHUNK - 1) / SEG_CHUNK;
  constexpr int U = 8;                                           // positions whose rows are in flight together
  for (int64_t ch = wave; ch < n_chunks; ch += n_waves) {
    const int64_t i0 = ch * SEG_CHUNK;
    const int n_here = (int)(min(n_ids, i0 + SEG_CHUNK) - i0);
    // lanes 0 .. 31: the chunk's ids and source rows; lane 32: the id in front of the chunk (a run may come in from the left)
    int64_t my_id = -1, my_src = 0;
    if (lane < n_here) {
      my_id = sorted_ids[i0 + lane];
      my_src = perm[i0 + lane];
    } else if (lane == 32 && i0 > 0) {
      my_id = sorted_ids[i0 - 1];
    }
    const int64_t id_before = readlane64(my_id, 32);
    const bool has_before = i0 > 0;
    for (int c0 = 0; c0 < dim; c0 += 256) {
      const int c = c0 + lane * 4;
      const bool col = c < dim;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int run_head = 0;                                          // position (in the chunk) the open run started at
      int64_t run_id = readlane64(my_id, 0);
      for (int p0 = 0; p0 < n_here; p0 += U) {
        f32x4 v[U], vx[U];
        bool use_x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          vx[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          use_x[u] = false;
          if (p0 + u < n_here && col) {
            const int64_t r = readlane64(my_src, p0 + u);
            if (r < n_a) {
              v[u] = Vec4IO<AT>::load(ga + r * dim + c);
              if (xg) {
                const int64_t b = r / window_len;
                const int l = (int)(r - b * window_len);
                if (l < seq_len) {
                  vx[u] = *reinterpret_cast<const f32x4*>(xg + (b * seq_len + l) * dim + c);
                  use_x[u] = true;
                }
              }
            } else if (r - n_a < n_b) {
              v[u] = Vec4IO<BT>::load(gb + (r - n_a) * dim + c);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int p = p0 + u;
          if (p >= n_here) break;
          const int64_t id = readlane64(my_id, p);
          if (id != run_id) {                                    // the open run ends in front of p: store it, open the next
            if (col) *reinterpret_cast<f32x4*>(out_rows + (i0 + run_head) * dim + c) = acc;
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            run_head = p;
            run_id = id;
          }
          acc += v[u];
          if (use_x[u]) acc += vx[u];
        }
      }
      if (col) *reinterpret_cast<f32x4*>(out_rows + (i0 + run_head) * dim + c) = acc;   // whole run, or this chunk's partial of a cut run
    }
    // slots: every run that STARTS in this chunk (ids outside the table - a data-layer bug; nn.Embedding would have raised in
    // the forward - get none: their rows stay out of the update instead of writing row_slot out of bounds)
    const int64_t up = __shfl_up(my_id, 1, 64);                  // (all lanes: no shuffle under divergence)
    if (lane < n_here) {
      const int64_t left = lane == 0 ? (has_before ? id_before : my_id - 1) : up;
      if (left != my_id && my_id >= 0 && my_id < n_rows) row_slot[my_id] = (int32_t)(i0 + lane);
    }
  }
}

// Second pass: the WORKGROUP of the chunk in which a cut run STARTS adds the run's partials (stored at the first positions of
// its later chunks) to the head row and clears them, so non-head rows are zero again.  Under Zipf ids one item can own a tenth
// of an exchanged list (21 k of 213 k positions at eight ranks: a run over 660 chunks): one wave walking those partials one
// after the other was the serial tail of the whole reduction, so the run's chunk range is dealt to the 16 waves of the
// workgroup in contiguous pieces - each summed in chunk order - and the 16 piece sums are added to the head in wave order.
// The association order is fixed by the list alone: bitwise reproducible, identical on every data-parallel replica.
constexpr int FIX_WAVES = 16;
__global__ __launch_bounds__(64 * FIX_WAVES) void segment_fixup_kernel(const int64_t* __restrict__ sorted_ids, int64_t n_ids,
                                                                       float* __restrict__ out_rows, int dim) {
  __shared__ f32x4 piece[FIX_WAVES][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_chunks = (n_ids + SEG_CHUNK - 1) / SEG_CHUNK;
  for (int64_t ch = blockIdx.x; ch + 1 < n_chunks; ch += gridDim.x) {
    const int64_t i0 = ch * SEG_CHUNK, i1 = i0 + SEG_CHUNK;
    const int64_t id = sorted_ids[i1 - 1];
    if (sorted_ids[i1] != id) continue;                          // the chunk's last run ends with the chunk
    if (sorted_ids[i0] == id && i0 > 0 && sorted_ids[i0 - 1] == id) continue;   // the run started earlier: not ours
    int64_t head = i1 - 1;
    while (head > i0 && sorted_ids[head - 1] == id) --head;
    int64_t last = ch + 1;                                       // chunks ch + 1 .. last hold partials of this run
    if (last + 1 < n_chunks && sorted_ids[(last + 1) * SEG_CHUNK] == id) {
      // a long run: gallop, then bisect for the last chunk whose first position still holds this id (ids are sorted)
      int64_t lo = last + 1, step = 1;
      while (lo + step < n_chunks && sorted_ids[(lo + step) * SEG_CHUNK] == id) {
        lo += step;
        step <<= 1;
      }
      int64_t hi = min(n_chunks - 1, lo + step);               // lo holds the id; hi may not
      while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (sorted_ids[mid * SEG_CHUNK] == id) lo = mid;
        else hi = mid - 1;
      }
      last = lo;
    }
    const int64_t K = last - ch;                                 // partial rows of this run
    const int64_t per = (K + FIX_WAVES - 1) / FIX_WAVES;
    const int64_t k_lo = ch + 1 + wave * per, k_hi = min(last, k_lo + per - 1);
    constexpr int U = 8;                                         // partial rows in flight per wave
    for (int c0 = 0; c0 < dim; c0 += 256) {
      const int c = c0 + lane * 4;
      const bool col = c < dim;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int64_t k0 = k_lo; k0 <= k_hi; k0 += U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k0 + u <= k_hi && col) v[u] = *reinterpret_cast<const f32x4*>(out_rows + (k0 + u) * SEG_CHUNK * dim + c);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k0 + u <= k_hi && col) {
            acc += v[u];
            *reinterpret_cast<f32x4*>(out_rows + (k0 + u) * SEG_CHUNK * dim + c) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
      piece[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && col) {
        f32x4 tot = *reinterpret_cast<const f32x4*>(out_rows + head * dim + c);
        const int n_pieces = (int)min((int64_t)FIX_WAVES, (K + per - 1) / per);
        for (int w = 0; w < n_pieces; ++w) tot += piece[w][lane];
        *reinterpret_cast<f32x4*>(out_rows + head * dim + c) = tot;
      }
      __syncthreads();
    }
  }
}

extern "C" int mhr_sparse_rows_segment_sum(const int64_t* sorted_ids, const int64_t* perm, int64_t n_ids,
                                           const void* grad_a, int a_dtype, int64_t n_a, const void* grad_b, int b_dtype,
                                           int64_t n_b, const float* x_grad, int seq_len, int window_len, float* out_rows,
                                           int32_t* row_slot, int64_t n_rows, int dim, void* stream) {
  MHR_REQUIRE(sorted_ids && perm && out_rows && row_slot, "sparse_rows_segment_sum: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && n_rows > 0, "sparse_rows_segment_sum: dim=%d must be a multiple of 4", dim);
  MHR_REQUIRE((n_a == 0 || grad_a) && (n_b == 0 || grad_b), "sparse_rows_segment_sum: null gradient buffer");
  MHR_REQUIRE(n_a + n_b >= n_ids, "sparse_rows_segment_sum: n_a+n_b < n_ids");
  if (x_grad) MHR_REQUIRE(window_len > 0 && seq_len > 0 && n_a % window_len == 0, "sparse_rows_segment_sum: bad window");
  if (n_ids == 0) return MHR_OK;
  if (window_len <= 0) window_len = 1;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for((n_ids + SEG_CHUNK - 1) / SEG_CHUNK, 4);
#define LAUNCH(AT, BT)                                                                                               \
  hipLaunchKernelGGL((segment_sum_kernel<AT, BT>), dim3(grid), dim3(256), 0, s, sorted_ids, perm, n_ids,             \
                     (const AT*)grad_a, n_a, (const BT*)grad_b, n_b, x_grad, seq_len, window_len, out_rows, row_slot, \
                     n_rows, dim)
  bool ab = a_dtype == MHR_BF16, bb = b_dtype == MHR_BF16;
  if (ab && bb) LAUNCH(bf16_t, bf16_t);
  else if (ab) LAUNCH(bf16_t, float);
  else if (bb) LAUNCH(float, bf16_t);
  else LAUNCH(float, float);
#undef LAUNCH
  MHR_CHECK_LAUNCH("sparse_rows_segment_sum");
  if (n_ids > SEG_CHUNK) {
    const int fgrid = mhr_grid_for((n_ids + SEG_CHUNK - 1) / SEG_CHUNK, 1);       // one workgroup per chunk that may head a cut run
    hipLaunchKernelGGL(segment_fixup_kernel, dim3(fgrid), dim3(64 * FIX_WAVES), 0, s, sorted_ids, n_ids, out_rows, dim);
    MHR_CHECK_LAUNCH("sparse_rows_segment_sum (fix-up)");
  }
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// AdamW
// ------------------------------------------------------------------------------------------
struct AdamConst {
  float beta1, beta2, one_m_beta1, one_m_beta2, eps, decay_mul, step_size, inv_sqrt_bc2, grad_scale;
};

static AdamConst make_adam(float lr, float beta1, float beta2, float eps, float wd, int step, float grad_scale) {
  AdamConst a;
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.one_m_beta1 = 1.0f - beta1;
  a.one_m_beta2 = 1.0f - beta2;
  a.eps = eps;
  a.decay_mul = 1.0f - lr * wd;
  a.step_size = (float)((double)lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  a.grad_scale = grad_scale;
  return a;
}

// Every operation is spelled with a rounding-explicit intrinsic: under -ffp-contract=fast the compiler may fuse
// (w * decay) - step * q either way round, and it chose differently in different kernels - the lazy replay
// (adam_rows_lazy_kernel) must reproduce the dense kernel's bits, so the order is fixed here once for all of them.
__device__ __forceinline__ void adam_update4(f32x4& w, f32x4& m, f32x4& v, f32x4 g, const AdamConst& a) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gk = __fmul_rn(g[k], a.grad_scale);
    const float wk = __fmul_rn(w[k], a.decay_mul);
    const float mk = __fmaf_rn(a.beta1, m[k], __fmul_rn(a.one_m_beta1, gk));
    const float vk = __fmaf_rn(a.beta2, v[k], __fmul_rn(__fmul_rn(a.one_m_beta2, gk), gk));
    const float denom = __fmaf_rn(__fsqrt_rn(vk), a.inv_sqrt_bc2, a.eps);
    w[k] = __fmaf_rn(-a.step_size, __fdiv_rn(mk, denom), wk);
    m[k] = mk;
    v[k] = vk;
  }
}

__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                        int64_t n_rows, int dim, const float* __restrict__ grad_rows,
                                                        int32_t* __restrict__ row_slot, AdamConst a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = wave; r < n_rows; r += n_waves) {
    int64_t slot = row_slot ? (int64_t)row_slot[r] : r;
    for (int c = lane * 4; c < dim; c += 256) {
      const int64_t o = r * dim + c;
      f32x4 wv = *reinterpret_cast<const f32x4*>(w + o);
      f32x4 mv = *reinterpret_cast<const f32x4*>(m + o);
      f32x4 vv = *reinterpret_cast<const f32x4*>(v + o);
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (slot >= 0) g = *reinterpret_cast<const f32x4*>(grad_rows + slot * dim + c);
      adam_update4(wv, mv, vv, g, a);
      *reinterpret_cast<f32x4*>(w + o) = wv;
      *reinterpret_cast<f32x4*>(m + o) = mv;
      *reinterpret_cast<f32x4*>(v + o) = vv;
    }
    if (row_slot && slot >= 0 && lane == 0) row_slot[r] = -1;
  }
}

extern "C" int mhr_adam_rows(float* w, float* m, float* v, int64_t n_rows, int dim, const float* grad_rows,
                             int32_t* row_slot, float grad_scale, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, void* stream) {
  MHR_REQUIRE(w && m && v && grad_rows, "adam_rows: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && n_rows > 0 && step >= 1, "adam_rows: bad sizes (dim=%d step=%d)", dim, step);
  AdamConst a = make_adam(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  int grid = mhr_grid_for(n_rows, 4 * 4);
  hipLaunchKernelGGL(adam_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, m, v, n_rows, dim, grad_rows,
                     row_slot, a);
  MHR_CHECK_LAUNCH("adam_rows");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// Lazy form of the dense table update.  The reference updates EVERY row every step (dense gradient, trainer.py:292-299): a
// row without gradient still decays its moments and moves by its momentum.  That update needs no gradient, so it can be
// replayed later: last_step[row] remembers the last step applied to a row and `hist` keeps the per-step constants of the last
// `hist_len` steps.  A row is brought up to date (same adam_update4, same constants, same order: bitwise the dense result)
//   * before the forward reads it  (mode 0: the ids of the batch, up to step - 1),
//   * when its gradient is applied (mode 1: the touched rows = segment heads of the sorted ids, through step),
//   * when everything is flushed   (mode 2: all rows through step; evaluation, checkpoints, every hist_len steps).
// Per step this touches the ~60 k rows of the batch instead of all 454 k (2.8 GB of HBM traffic at cfg1).
// ------------------------------------------------------------------------------------------
struct AdamLazy {
  float beta1, beta2, one_m_beta1, one_m_beta2, eps, grad_scale;
  int hist_len, step, mode;
};

__device__ __forceinline__ AdamConst lazy_consts(const AdamLazy& a, const float* __restrict__ hist, int s, float grad_scale) {
  const float* h = hist + (int64_t)(s % a.hist_len) * 4;
  AdamConst c;
  c.beta1 = a.beta1; c.beta2 = a.beta2; c.one_m_beta1 = a.one_m_beta1; c.one_m_beta2 = a.one_m_beta2; c.eps = a.eps;
  c.decay_mul = h[0]; c.step_size = h[1]; c.inv_sqrt_bc2 = h[2]; c.grad_scale = grad_scale;
  return c;
}

__global__ __launch_bounds__(256) void adam_rows_lazy_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                             int64_t n_rows, int dim, const int64_t* __restrict__ ids,
                                                             int64_t n_ids, const float* __restrict__ grad_rows,
                                                             int32_t* __restrict__ row_slot, int32_t* __restrict__ last_step,
                                                             const float* __restrict__ hist, AdamLazy a,
                                                             const int64_t* __restrict__ step_dev) {
  if (step_dev) a.step = (int)step_dev[0];      // hipGraph-replayed step: the step number lives in device memory
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t n_work = a.mode == 2 ? n_rows : n_ids;
  for (int64_t i = wave; i < n_work; i += n_waves) {
    int64_t row = i;
    if (a.mode != 2) {
      row = ids[i];
      if (row < 0 || row >= n_rows) continue;
      if (a.mode == 1 && i > 0 && ids[i - 1] == row) continue;          // not the head of its segment
    }
    const int target = a.mode == 0 ? a.step - 1 : a.step;               // last step this visit leaves applied
    int last = last_step[row];
    if (last >= target) continue;
    if (a.mode == 0) {                                                  // duplicates in the id list: one wave claims the row
      int won = 0;
      if (lane == 0) won = atomicCAS(last_step + row, last, target) == last ? 1 : 0;
      won = __builtin_amdgcn_readfirstlane(won);
      if (!won) continue;
    }
    for (int c = lane * 4; c < dim; c += 256) {
      const int64_t o = row * dim + c;
      f32x4 wv = *reinterpret_cast<const f32x4*>(w + o);
      f32x4 mv = *reinterpret_cast<const f32x4*>(m + o);
      f32x4 vv = *reinterpret_cast<const f32x4*>(v + o);
      const int replay_to = a.mode == 1 ? a.step - 1 : target;          // steps without gradient
      for (int s = last + 1; s <= replay_to; ++s) {
        const AdamConst cs = lazy_consts(a, hist, s, 1.0f);
        adam_update4(wv, mv, vv, f32x4{0.f, 0.f, 0.f, 0.f}, cs);
      }
      if (a.mode == 1) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(grad_rows + i * dim + c);   // the segment head holds the summed row
        const AdamConst cs = lazy_consts(a, hist, a.step, a.grad_scale);
        adam_update4(wv, mv, vv, g, cs);
      }
      *reinterpret_cast<f32x4*>(w + o) = wv;
      *reinterpret_cast<f32x4*>(m + o) = mv;
      *reinterpret_cast<f32x4*>(v + o) = vv;
    }
    if (a.mode != 0 && lane == 0) last_step[row] = target;
    if (a.mode == 1 && row_slot && lane == 0) row_slot[row] = -1;
  }
}

extern "C" int mhr_adam_consts(float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* out4) {
  MHR_REQUIRE(out4 && step >= 1, "adam_consts: bad arguments");
  const AdamConst a = make_adam(lr, beta1, beta2, eps, weight_decay, step, 1.0f);     // host only: the constants of one step
  out4[0] = a.decay_mul; out4[1] = a.step_size; out4[2] = a.inv_sqrt_bc2; out4[3] = lr;
  return MHR_OK;
}

extern "C" int mhr_adam_rows_lazy(float* w, float* m, float* v, int64_t n_rows, int dim, const int64_t* ids, int64_t n_ids,
                                  const float* grad_rows, int32_t* row_slot, int32_t* last_step, const float* hist,
                                  int hist_len, int step, float grad_scale, float beta1, float beta2, float eps, int mode,
                                  const int64_t* step_dev, void* stream) {
  MHR_REQUIRE(w && m && v && last_step && hist, "adam_rows_lazy: null pointer");
  MHR_REQUIRE(mode >= 0 && mode <= 2 && (mode == 2 || ids) && (mode != 1 || grad_rows), "adam_rows_lazy: bad mode / inputs");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && n_rows > 0 && hist_len > 0 && step >= 0, "adam_rows_lazy: bad sizes");
  const int64_t n_work = mode == 2 ? n_rows : n_ids;
  if (n_work <= 0) return MHR_OK;
  AdamLazy a;
  a.beta1 = beta1; a.beta2 = beta2; a.one_m_beta1 = 1.0f - beta1; a.one_m_beta2 = 1.0f - beta2; a.eps = eps;
  a.grad_scale = grad_scale; a.hist_len = hist_len; a.step = step; a.mode = mode;
  int grid = mhr_grid_for(n_work, 4 * 4);
  hipLaunchKernelGGL(adam_rows_lazy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, m, v, n_rows, dim, ids, n_ids,
                     grad_rows, row_slot, last_step, hist, a, step_dev);
  MHR_CHECK_LAUNCH("adam_rows_lazy");
  return MHR_OK;
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        AdamConst a, bf16_t* __restrict__ w16,
                                                        const float* __restrict__ hist, int hist_len,
                                                        const int64_t* __restrict__ step_dev) {
  if (step_dev) {       // hipGraph-replayed step: this step's constants from the device-side history (mhr_adam_consts rows)
    const float* h = hist + (step_dev[0] % hist_len) * 4;
    a.decay_mul = h[0]; a.step_size = h[1]; a.inv_sqrt_bc2 = h[2];
  }
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nt = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t i = tid; i < n4; i += nt) {
    f32x4 wv = reinterpret_cast<f32x4*>(w)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    adam_update4(wv, mv, vv, gv, a);
    reinterpret_cast<f32x4*>(w)[i] = wv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (w16) Vec4IO<bf16_t>::store(w16 + i * 4, wv);        // bf16 shadow of the updated weights (the GEMM operand)
  }
  for (int64_t i = n4 * 4 + tid; i < n; i += nt) {  // tail
    float gk = g[i] * a.grad_scale;
    float wk = w[i] * a.decay_mul;
    float mk = a.beta1 * m[i] + a.one_m_beta1 * gk;
    float vk = a.beta2 * v[i] + a.one_m_beta2 * gk * gk;
    w[i] = wk - a.step_size * (mk / (sqrtf(vk) * a.inv_sqrt_bc2 + a.eps));
    m[i] = mk;
    v[i] = vk;
    if (w16) w16[i] = (bf16_t)w[i];
  }
}

extern "C" int mhr_adam_flat(float* w, const float* g, float* m, float* v, int64_t n, float grad_scale, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, void* w_bf16,
                             const float* hist, int hist_len, const int64_t* step_dev, void* stream) {
  MHR_REQUIRE(w && g && m && v, "adam_flat: null pointer");
  MHR_REQUIRE(!step_dev || (hist && hist_len > 0), "adam_flat: step_dev needs the constants' history");
  MHR_REQUIRE(!w_bf16 || (uintptr_t)w_bf16 % 8 == 0, "adam_flat: the bf16 shadow must be 8-byte aligned");
  MHR_REQUIRE(n >= 0 && step >= 1, "adam_flat: bad sizes");
  MHR_REQUIRE(((uintptr_t)w % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
              "adam_flat: buffers must be 16-byte aligned");
  if (n == 0) return MHR_OK;
  AdamConst a = make_adam(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  int grid = mhr_grid_for(n / 4 + 1, 256, 2048);
  hipLaunchKernelGGL(adam_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, a, (bf16_t*)w_bf16,
                     hist, hist_len, step_dev);
  MHR_CHECK_LAUNCH("adam_flat");
  return MHR_OK;
}
