// Row-stationary streaming GEMM core shared by the catalog scorer and the sampled-softmax kernels.
//
//   S^T[streamed row, stationary col] = sum_k  T[streamed row][k] * U[stationary col][k]
//
// One 256-thread workgroup = 4 waves.  Each wave keeps RF x 32 "stationary" rows (users / tokens) as
// MFMA B-operand fragments in registers for the whole kernel (RF*NKS*4 VGPRs, K = 16*NKS <= 256) and the
// workgroup streams 32-row tiles of the other matrix (items / negatives) through a double-buffered,
// XOR-swizzled LDS image, so each 16-byte ds_read_b128 of the tile feeds RF MFMAs.  Accumulators hold
// S^T: streamed rows on the registers, stationary rows on the lanes - so per-stationary-row state
// (threshold, running sum, mask word) is one VGPR per lane, and no cross-lane work happens in the loop.
//
// v_mfma_f32_32x32x16_bf16 lane maps (cdna guide section 3):
//   A: lane l holds A[row l&31][k = 8*(l>>5) + j];  B: lane l holds B[k = 8*(l>>5) + j][col l&31]
//   C: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
#pragma once
#include <type_traits>

#include "mhr_common.h"

namespace sg {

__device__ __forceinline__ int crow(int g, int half) { return (g & 3) + 8 * (g >> 2) + 4 * half; }

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.0f;
  return z;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

template <int NKS>
struct Tile {
  static constexpr int DIM = NKS * 16;
  static constexpr int CH = NKS * 2;                       // 16-byte chunks per row
  static constexpr int SW = CH >= 16 ? 15 : CH - 1;        // swizzle mask
  static constexpr int ROW_BYTES = DIM * 2;
  static constexpr int BYTES = 32 * ROW_BYTES;             // one 32-row tile
  static constexpr int CHUNKS = 32 * CH;                   // 16-byte chunks per tile
  static constexpr int PER_THREAD = (CHUNKS + 255) / 256;  // staging chunks per thread

  // XOR swizzle key of a row: bits [3:2] = row & 3, bits [1:0] = (row >> 2) & 3.  With it BOTH access shapes are
  // bank-conflict free on the 256-byte bank row: a ds_read_b128 lane group (16 distinct rows, one chunk) and a
  // ds_read_b64_tr_b16 32-lane group (4 consecutive rows x 4 consecutive chunks) each touch 16 distinct 16-byte
  // slots.  (key = row & 15 serves the row reads but puts the 4 rows of a transposed read on the same 4 slots:
  // measured 54 % of LDS cycles lost to conflicts in the sampled-softmax backward.)
  static __device__ __forceinline__ int key(int row) { return (((row & 3) << 2) | ((row >> 2) & 3)) & SW; }
  // byte offset of chunk c of row `row` inside the swizzled tile image
  static __device__ __forceinline__ int off(int row, int c) { return row * ROW_BYTES + ((c ^ key(row)) << 4); }

  // A-operand fragment of k-step ks for lane (r, half)
  static __device__ __forceinline__ bf16x8 read_a(const unsigned char* tile, int r, int half, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + off(r, ks * 2 + half));
  }
};

// Staging registers for one tile: each thread moves PER_THREAD 16-byte chunks global -> regs -> LDS.
template <int NKS>
struct Stage {
  bf16x8 v[Tile<NKS>::PER_THREAD];

  // row_ptr(row) returns the global address of the 32-row tile's row `row` (bf16, DIM contiguous) or nullptr.
  template <typename RowPtr>
  __device__ __forceinline__ void load(RowPtr row_ptr) {
    using T = Tile<NKS>;
#pragma unroll
    for (int i = 0; i < T::PER_THREAD; ++i) {
      const int id = threadIdx.x + i * 256;
      if (id < T::CHUNKS) {
        const int row = id / T::CH, c = id % T::CH;
        const bf16_t* p = row_ptr(row);
        v[i] = p ? *reinterpret_cast<const bf16x8*>(p + c * 8) : zero8();
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* tile) const {
    using T = Tile<NKS>;
#pragma unroll
    for (int i = 0; i < T::PER_THREAD; ++i) {
      const int id = threadIdx.x + i * 256;
      if (id < T::CHUNKS) {
        const int row = id / T::CH, c = id % T::CH;
        *reinterpret_cast<bf16x8*>(tile + T::off(row, c)) = v[i];
      }
    }
  }
};

// Per-lane LDS byte offsets of every fragment read, computed ONCE per kernel.  The swizzle is an XOR on the low four
// chunk bits, so all reads of a tile are `lane offset + compile-time immediate`:
//   row read of k-step ks      : a[ks & 7] + 256*(ks >> 3)
//   transposed read (dc, s, hi) : t[hi][dc & 3] + 256*(dc >> 2) + 16*s*ROW_BYTES        (hi = rows +8)
// With the ring slot a compile-time constant too (tile loops are unrolled by the ring depth) the main loops carry no
// address arithmetic at all; recomputing the XORs per read cost 150-200 VALU instructions per 32-MFMA tile.
template <int NKS>
struct LaneAddr {
  using T = Tile<NKS>;
  static constexpr int NA = NKS < 8 ? NKS : 8;
  int a[NA];
  int t[2][4];
  __device__ __forceinline__ void init(int lane) {
    const int r = lane & 31, half = lane >> 5;
#pragma unroll
    for (int j = 0; j < NA; ++j) a[j] = T::off(r, 2 * j + half);
    const int i = lane & 15, q = i >> 2, p = i & 3, g1 = (lane >> 4) & 1;
    const int cl = 2 * g1 + (p >> 1), bo = (p & 1) * 8;
    const int row0 = 4 * half + q;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      t[0][d] = T::off(row0, d * 4 + cl) + bo;
      t[1][d] = T::off(row0 + 8, d * 4 + cl) + bo;
    }
  }
  __device__ __forceinline__ bf16x8 read_a(const unsigned char* tile, int ks) const {
    return *reinterpret_cast<const bf16x8*>(tile + a[ks & 7] + 256 * (ks >> 3));
  }
  // Transposed fragment straight from the row-major swizzled tile with ds_read_b64_tr_b16 (cdna guide T10): the
  // B operand of a product that sums over the ROW index of an accumulator tile (rows = this tile's 32 rows):
  // element j of lane half h must be tile[row 16*s + 8*(j>>2) + 4*h + (j&3)][col dc*32 + (lane&31)].
  // Per 16-lane group the instruction reads a 4-row x 16-column block: lane 4q+p supplies the address of row q,
  // columns 4p..4p+3, and lane i receives column i of the 4 rows (probed on gfx950: tools/probe_tr.hip).  Two reads
  // (rows +0 and +8) make one fragment.  EXEC must be all ones.  Columns beyond DIM (only when DIM < 32) read
  // neighbouring rows / zeros: callers mask them.
  // (The _v4i16 form of the builtin followed by per-element bit casts is mis-optimised by hipcc 7.2 - all four
  // elements collapse to element 0 - so the _v4bf16 form and a whole-vector shuffle are used.)
  __device__ __forceinline__ bf16x8 read_tr(const unsigned char* tile, int dc, int s) const {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int imm = 256 * (dc >> 2) + 16 * s * T::ROW_BYTES;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + t[0][dc & 3] + imm));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + t[1][dc & 3] + imm));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

// stand-alone form (tests / probes): same result as LaneAddr::read_tr
template <int NKS>
__device__ __forceinline__ bf16x8 read_tr_frag(const unsigned char* tile, int dc, int s, int lane) {
  LaneAddr<NKS> la;
  la.init(lane);
  return la.read_tr(tile, dc, s);
}

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA staging (global_load_lds_dwordx4): tiles go HBM/L2 -> LDS without passing through registers.
// One wave-instruction writes 1 KiB of LDS linearly (wave-uniform base + lane*16), so the XOR swizzle is applied
// on the per-lane SOURCE address (cdna guide section 5.4 rule 21): the lane that fills slot `s` of row `row`
// fetches chunk s ^ key(row).  Tiles rotate through a 3-deep ring: while tile t is consumed, t+1 and t+2 are in
// flight; a wave waits for its own pieces with a counted `s_waitcnt vmcnt(N)` and a raw s_barrier publishes them
// (no __syncthreads(): its implicit vmcnt(0) would drain the ring).  Inside such a loop there must be no ordinary
// global load (hipcc would wait vmcnt(0) for it); stores and atomics only make the counted wait conservative.
// ---------------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 10, "extend wait_vmcnt");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
}

__device__ __forceinline__ void ring_barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int NKS>
struct Dma {
  using T = Tile<NKS>;
  static constexpr int PIECES = T::BYTES / 1024;          // 1-KiB pieces per tile
  static constexpr int PW = (PIECES + 3) / 4;             // pieces issued per wave per tile (waves beyond PIECES issue none)
  static_assert(T::BYTES % 1024 == 0, "tile must be a whole number of 1-KiB pieces");

  // row_ptr(row) must return a VALID global address for every row 0..31 (clamp out-of-range rows; their scores are
  // masked by the caller) - a DMA cannot zero-fill.
  template <typename RowPtr>
  static __device__ __forceinline__ void issue(unsigned char* tile, RowPtr row_ptr, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int pc = wave * PW + i;                        // wave-uniform
      if (pc < PIECES) {
        const int pos = pc * 1024 + lane * 16;
        const int row = pos / T::ROW_BYTES, slot = (pos % T::ROW_BYTES) >> 4;
        const bf16_t* src = row_ptr(row) + ((slot ^ T::key(row)) << 3);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + pc * 1024), 16, 0, 0);
      }
    }
  }
};

// one 4-byte word per lane -> 256 bytes of LDS at a wave-uniform base (per-token scalars / mask words)
__device__ __forceinline__ void dma_words(const void* lane_src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)lane_src, (lptr_t)lds_wave_base, 4, 0, 0);
}

// acc[f] (+)= tile . frag[f]^T for the RF stationary fragments.  The A fragments are read PF k-steps ahead of the
// MFMAs that consume them (two register sets), so LDS latency hides under the matrix pipe instead of every MFMA
// waiting for its own ds_read (what hipcc emits for the naive loop at one wave per SIMD).
template <int NKS, int RF, int PF = 2>
__device__ __forceinline__ void mma_tile(const unsigned char* tile, const LaneAddr<NKS>& la, const bf16x8 (&frag)[RF][NKS],
                                         f32x16 (&acc)[RF]) {
  constexpr int P = PF < NKS ? PF : NKS;
  constexpr int NB = (NKS + P - 1) / P;
  bf16x8 a[2][P];
#pragma unroll
  for (int j = 0; j < P; ++j) a[0][j] = la.read_a(tile, j);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) {
#pragma unroll
      for (int j = 0; j < P; ++j)
        if ((b + 1) * P + j < NKS) a[(b + 1) & 1][j] = la.read_a(tile, (b + 1) * P + j);
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int ks = b * P + j;
      if (ks < NKS) {
#pragma unroll
        for (int f = 0; f < RF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[b & 1][j], frag[f][ks], acc[f], 0, 0, 0);
      }
    }
  }
}

// Same product with an independent VALU job pinned into the MFMA gaps: after the MFMA of k-step ks the caller's
// epi(e) runs for accumulator elements e = ks*EPK .. ks*EPK+EPK-1 of ANOTHER (already complete) accumulator, and a
// sched_barrier fixes that order.  At one wave per SIMD the in-order issue stage cannot overlap a batch of MFMAs with
// VALU work that follows the batch in program order; placing one element's epilogue (~8 VALU, one v_exp) in each
// 32-cycle MFMA gap hides it behind the matrix pipe (cdna guide: "budget every MFMA gap ... and place them").
template <int NKS, int PF, typename Epi>
__device__ __forceinline__ void mma_tile_epi(const unsigned char* tile, const LaneAddr<NKS>& la, const bf16x8 (&frag)[1][NKS],
                                             f32x16& acc, Epi epi) {
  constexpr int P = PF < NKS ? PF : NKS;
  constexpr int NB = (NKS + P - 1) / P;
  constexpr int EPK = (16 + NKS - 1) / NKS;                // accumulator elements handled per k-step
  bf16x8 a[2][P];
#pragma unroll
  for (int j = 0; j < P; ++j) a[0][j] = la.read_a(tile, j);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) {
#pragma unroll
      for (int j = 0; j < P; ++j)
        if ((b + 1) * P + j < NKS) a[(b + 1) & 1][j] = la.read_a(tile, (b + 1) * P + j);
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int ks = b * P + j;
      if (ks < NKS) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[b & 1][j], frag[0][ks], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < EPK; ++e)
          if (ks * EPK + e < 16) epi(ks * EPK + e);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// out[dc] += A(g0,g1) . tile^T-fragments: the second product of the backward kernels (sums over the tile's 32 rows).
// g0/g1 are the two k-step fragments of the gated tile; B fragments come from LaneAddr::read_tr, read one dc ahead.
template <int NKS, int ND>
__device__ __forceinline__ void mma_tile_tr(const unsigned char* tile, const LaneAddr<NKS>& la, bf16x8 g0, bf16x8 g1,
                                            f32x16 (&out)[ND]) {
  bf16x8 b[2][2];
  b[0][0] = la.read_tr(tile, 0, 0);
  b[0][1] = la.read_tr(tile, 0, 1);
#pragma unroll
  for (int dc = 0; dc < ND; ++dc) {
    if (dc + 1 < ND) {
      b[(dc + 1) & 1][0] = la.read_tr(tile, dc + 1, 0);
      b[(dc + 1) & 1][1] = la.read_tr(tile, dc + 1, 1);
    }
    out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, b[dc & 1][0], out[dc], 0, 0, 0);
    out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, b[dc & 1][1], out[dc], 0, 0, 0);
  }
}

// Runs body(slot_constant, i) for i = 0..n-1 with slot = i % DEPTH as a COMPILE-TIME constant (the loop is unrolled
// by the ring depth), so ring-slot bases fold into the immediate offset field of the LDS instructions.
template <int DEPTH, typename Body>
__device__ __forceinline__ void ring_loop(int n, Body body) {
  for (int i0 = 0; i0 < n; i0 += DEPTH) {
    body(std::integral_constant<int, 0>{}, i0);
    if constexpr (DEPTH > 1) { if (i0 + 1 < n) body(std::integral_constant<int, 1>{}, i0 + 1); }
    if constexpr (DEPTH > 2) { if (i0 + 2 < n) body(std::integral_constant<int, 2>{}, i0 + 2); }
    if constexpr (DEPTH > 3) { if (i0 + 3 < n) body(std::integral_constant<int, 3>{}, i0 + 3); }
  }
}

// Scheduling hint for a region that holds one 16-MFMA product (with its transposed LDS reads) and an independent
// VALU epilogue: ask hipcc to emit them as 16 x {1 MFMA, 2 DS reads, `valu` VALU} so the epilogue runs in the issue
// slots the matrix pipe leaves free (cdna guide T19; at one wave per SIMD nothing else can fill those slots).
template <int VALU_PER_MFMA>
__device__ __forceinline__ void interleave_mfma_valu_16() {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);              // DS read
    __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);  // VALU
  }
}

}  // namespace sg
