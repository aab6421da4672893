// Packed ("jagged") token rows for the encoder: the valid positions of the B windows back to back in one [capacity, D] buffer.
//
// Reference path (file:line under code/REC/): the loaders pad every user window to MAX_ITEM_LIST_LENGTH at the front
// (data/dataset/trainset.py:111-137, evalset.py:34-41) and model/IDNet/hstu.py:221-328 runs every layer over all B x L rows; the
// padding rows are masked out of the attention (hstu.py:137-160), of the loss tokens (hstu.py:682-690) and of the decode
// (hstu.py:879-913).  At the synthetic cfg1 batch 37 % of the rows are such padding.  The encoder can run on the valid rows
// only: `mhr_seq_pack_maps` numbers them (sequence by sequence, in window order), `mhr_rows_gather_masked` moves rows between
// the two layouts (both directions of both passes are gathers: the map is injective), the attention kernels address the
// sequences through `cu_rows`, every other layer kernel is row-wise.  The capacity is chosen by the host (a bucketed upper
// bound of the batch's valid positions, known to the loader): shapes stay static for the replayed step graph.
#include "mhr_common.h"

namespace {

// lens[b] = valid positions of sequence b: one wave per sequence, ballots over the mask bytes (written into cu_rows[b + 1])
__global__ __launch_bounds__(64) void seq_len_kernel(const uint8_t* __restrict__ key_valid, int L, int32_t* __restrict__ cu_rows) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const uint8_t* kv = key_valid + (int64_t)b * L;
  int n = 0;
  for (int l0 = 0; l0 < L; l0 += 64) {
    const int l = l0 + lane;
    n += __popcll(__ballot(l < L && kv[l] != 0));
  }
  if (lane == 0) cu_rows[b + 1] = n;
}

// in place: cu_rows[1 .. B] hold the lengths -> cu_rows[b] = number of valid positions in sequences 0 .. b-1, clamped to the capacity
// (one workgroup; B <= 65536: a thread owns B / 256 consecutive sequences)
__global__ __launch_bounds__(256) void seq_cu_kernel(int B, int cap, int32_t* __restrict__ cu_rows, int32_t* __restrict__ overflow) {
  __shared__ int s_part[256];
  const int t = threadIdx.x;
  const int per = (B + 255) / 256;
  const int b0 = min(B, t * per), b1 = min(B, b0 + per);
  int mine = 0;
  for (int b = b0; b < b1; ++b) mine += cu_rows[b + 1];
  s_part[t] = mine;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int c = s_part[i];
      s_part[i] = run;
      run += c;
    }
    if (overflow) overflow[0] = run > cap ? run : 0;      // the batch does not fit the capacity: the rows past it are DROPPED
  }
  __syncthreads();
  int run = s_part[t];
  for (int b = b0; b < b1; ++b) {                          // (reads its own b + 1 before any other thread could have written it:
    const int n = cu_rows[b + 1];                          //  thread t writes entries b0 + 1 .. b1 only, and entry b0 + 1 after reading it)
    run += n;
    cu_rows[b + 1] = min(run, cap);
  }
  if (t == 0) cu_rows[0] = 0;
}

// one wave per sequence: row_of[(b, l)] = packed row of a valid position (-1: padding / past the capacity), src_of[packed] = b L + l;
// blocks >= B clear the rows of src_of behind the last sequence
__global__ __launch_bounds__(64) void seq_maps_kernel(const uint8_t* __restrict__ key_valid, int B, int L, int cap,
                                                      const int32_t* __restrict__ cu_rows, int32_t* __restrict__ src_of,
                                                      int32_t* __restrict__ row_of) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= B) {
    const int tail0 = cu_rows[B];
    for (int r = tail0 + ((int)blockIdx.x - B) * 64 + lane; r < cap; r += ((int)gridDim.x - B) * 64) src_of[r] = -1;
    return;
  }
  const int b = blockIdx.x;
  const uint8_t* kv = key_valid + (int64_t)b * L;
  int at = cu_rows[b];
  const int end = cu_rows[b + 1];                          // (clamped to the capacity by seq_cu_kernel)
  for (int l0 = 0; l0 < L; l0 += 64) {
    const int l = l0 + lane;
    const bool v = l < L && kv[l] != 0;
    const unsigned long long m = __ballot(v);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    const int r = at + rank;
    if (l < L) {
      const bool keep = v && r < end;
      row_of[(int64_t)b * L + l] = keep ? r : -1;
      if (keep) src_of[r] = b * L + l;
    }
    at += __popcll(m);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_masked_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx,
                                                            T* __restrict__ out, int64_t n_out, int dim) {
  constexpr int V = 16 / sizeof(T);                        // elements per 16-byte access
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  constexpr int U = 4;                                     // rows in flight per wave
  for (int64_t r0 = wave * U; r0 < n_out; r0 += n_waves * U) {
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = r0 + u < n_out ? idx[r0 + u] : -1;
    for (int c = lane * V; c < dim; c += 64 * V) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = id[u] >= 0 ? *reinterpret_cast<const f32x4*>(src + (int64_t)id[u] * dim + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r0 + u < n_out) *reinterpret_cast<f32x4*>(out + (r0 + u) * dim + c) = v[u];
    }
  }
}

}  // namespace

extern "C" int mhr_seq_pack_maps(const uint8_t* key_valid, int B, int L, int capacity, int32_t* cu_rows, int32_t* src_of,
                                 int32_t* row_of, int32_t* overflow, void* stream) {
  MHR_REQUIRE(key_valid && cu_rows && src_of && row_of, "seq_pack_maps: null pointer");
  MHR_REQUIRE(B > 0 && B <= 65536 && L > 0 && capacity > 0 && (int64_t)B * L < (1ll << 31), "seq_pack_maps: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(seq_len_kernel, dim3(B), dim3(64), 0, s, key_valid, L, cu_rows);
  hipLaunchKernelGGL(seq_cu_kernel, dim3(1), dim3(256), 0, s, B, capacity, cu_rows, overflow);
  hipLaunchKernelGGL(seq_maps_kernel, dim3(B + 16), dim3(64), 0, s, key_valid, B, L, capacity, cu_rows, src_of, row_of);
  MHR_CHECK_LAUNCH("seq_pack_maps");
  return MHR_OK;
}

extern "C" int mhr_rows_gather_masked(const void* src, int dtype, const int32_t* idx, void* out, int64_t n_out, int dim,
                                      void* stream) {
  MHR_REQUIRE(src && idx && out, "rows_gather_masked: null pointer");
  MHR_REQUIRE(dtype == MHR_F32 || dtype == MHR_BF16, "rows_gather_masked: dtype %d unsupported", dtype);
  MHR_REQUIRE(dim > 0 && dim % (dtype == MHR_F32 ? 4 : 8) == 0, "rows_gather_masked: dim=%d must fill whole 16-byte pieces", dim);
  MHR_REQUIRE(((uintptr_t)src | (uintptr_t)out) % 16 == 0, "rows_gather_masked: operands must be 16-byte aligned");
  if (n_out <= 0) return MHR_OK;
  const int grid = mhr_grid_for((n_out + 3) / 4, 4);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MHR_F32)
    hipLaunchKernelGGL(gather_masked_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, idx, (float*)out, n_out, dim);
  else
    hipLaunchKernelGGL(gather_masked_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)src, idx, (bf16_t*)out, n_out, dim);
  MHR_CHECK_LAUNCH("rows_gather_masked");
  return MHR_OK;
}
