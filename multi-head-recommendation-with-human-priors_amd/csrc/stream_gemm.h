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
#include <initializer_list>
#include <type_traits>
#include <utility>

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
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
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
  static_assert(N >= 0 && N <= 63, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
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

// Per-piece form for loops that place single DMA instructions into MFMA gaps.  Every wave issues exactly PW pieces
// per tile (waves beyond the tile's piece count repeat an earlier piece: same bytes to the same LDS address), so one
// compile-time `s_waitcnt vmcnt(n)` is right for all four waves.
//   Fast path: source rows contiguous (row stride = DIM): address = wave-uniform tile base (SGPR pair) + a per-lane
//   32-bit offset computed once per kernel -> `global_load_lds_dwordx4 v_off, s[base]`, no per-tile address VALU.
template <int NKS>
struct DmaPieces {
  using T = Tile<NKS>;
  static constexpr int PIECES = T::BYTES / 1024;
  static constexpr int PW = (PIECES + 3) / 4;
  uint32_t off[PW];        // byte offset of my 16 bytes inside the (unswizzled) source tile, per piece
  int pc[PW];              // piece index (wave-uniform)
  __device__ __forceinline__ void init(int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      pc[i] = (wave * PW + i) % PIECES;
      const int pos = pc[i] * 1024 + lane * 16;
      const int row = pos / T::ROW_BYTES, slot = (pos % T::ROW_BYTES) >> 4;
      off[i] = (uint32_t)(row * T::ROW_BYTES + ((slot ^ T::key(row)) << 4));
    }
  }
  template <int K>
  __device__ __forceinline__ void piece(unsigned char* tile_lds, const char* tile_src) const {
    __builtin_amdgcn_global_load_lds((gptr_t)(tile_src + off[K]), (lptr_t)(tile_lds + pc[K] * 1024), 16, 0, 0);
  }
  // rows through a callback (clamped / gathered rows): per-piece address arithmetic, same instruction count
  template <int K, typename RowPtr>
  __device__ __forceinline__ void piece_rows(unsigned char* tile_lds, RowPtr row_ptr, int lane) const {
    const int pos = pc[K] * 1024 + lane * 16;
    const int row = pos / T::ROW_BYTES, slot = (pos % T::ROW_BYTES) >> 4;
    const bf16_t* src = row_ptr(row) + ((slot ^ T::key(row)) << 3);
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile_lds + pc[K] * 1024), 16, 0, 0);
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

// ---------------------------------------------------------------------------------------------------------
// The same product with the transposed reads emitted as inline asm.  hipcc 7.2 guards every
// __builtin_amdgcn_ds_read_tr16_b64 with `s_waitcnt vmcnt(0)` when LDS-DMA loads are in flight (it cannot prove the
// read does not alias the DMA destination), which drains the whole prefetch ring once per tile - in the
// sampled-softmax backward kernels that exposed a full L2/HBM round trip per 32 MFMAs.  The asm form carries no
// memory operand, so only the counted waits of the ring remain.  The price is doing the LDS wait counting by hand:
// reads are issued PD column chunks ahead and `s_waitcnt lgkmcnt(4*PD)` (4 reads per chunk) precedes each MFMA pair;
// the waited registers are threaded through the wait statement as "+v" operands so the MFMAs cannot be hoisted
// above it.  LDS instructions hipcc adds on its own only make the counted wait stricter, never weaker.
//   lds_base : LDS byte address of the ring (see lds_addr()), OFF: compile-time byte offset of the tile in the ring.
// ---------------------------------------------------------------------------------------------------------
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <int IMM>
__device__ __forceinline__ u32x2 ds_read_tr_asm(uint32_t addr) {
  static_assert(IMM >= 0 && IMM < 65536, "LDS offset field is 16 bits");
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return v;
}

template <int N>
__device__ __forceinline__ void wait_lgkm(u32x2& a, u32x2& b, u32x2& c, u32x2& d) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

template <int N>
__device__ __forceinline__ void wait_lgkm2(u32x2& a, u32x2& b) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

template <int NKS>
struct TrAddr {           // per-lane LDS byte ADDRESSES (ring base included) of the transposed reads
  uint32_t t[2][4];
  __device__ __forceinline__ void init(const LaneAddr<NKS>& la, const void* ring) {
    const uint32_t b = lds_addr(ring);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int d = 0; d < 4; ++d) t[h][d] = b + (uint32_t)la.t[h][d];
  }
};

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int NKS, int ND, int OFF, int PD = 2>
__device__ __forceinline__ void mma_tile_tr_asm(const TrAddr<NKS>& ta, bf16x8 g0, bf16x8 g1, f32x16 (&out)[ND]) {
  using T = Tile<NKS>;
  constexpr int P = PD < ND ? PD : ND;
  u32x2 r[P + 1][4];      // [.][0,1] = rows +0/+8 of k-step 0, [.][2,3] = of k-step 1
  auto issue = [&](auto dc_c) {
    constexpr int dc = decltype(dc_c)::value;
    constexpr int imm = OFF + 256 * (dc >> 2), s1 = 16 * T::ROW_BYTES;
    u32x2* q = r[dc % (P + 1)];
    q[0] = ds_read_tr_asm<imm>(ta.t[0][dc & 3]);
    q[1] = ds_read_tr_asm<imm>(ta.t[1][dc & 3]);
    q[2] = ds_read_tr_asm<imm + s1>(ta.t[0][dc & 3]);
    q[3] = ds_read_tr_asm<imm + s1>(ta.t[1][dc & 3]);
  };
  auto step = [&](auto dc_c) {
    constexpr int dc = decltype(dc_c)::value;
    if constexpr (dc + P < ND) issue(std::integral_constant<int, dc + P>{});
    constexpr int ahead = (ND - 1 - dc) < P ? (ND - 1 - dc) : P;     // chunks issued after this one
    u32x2* q = r[dc % (P + 1)];
    wait_lgkm<4 * ahead>(q[0], q[1], q[2], q[3]);
    const u32x4 b0 = {q[0].x, q[0].y, q[1].x, q[1].y}, b1 = {q[2].x, q[2].y, q[3].x, q[3].y};
    out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, __builtin_bit_cast(bf16x8, b0), out[dc], 0, 0, 0);
    out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, __builtin_bit_cast(bf16x8, b1), out[dc], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<P>(issue);
  static_for<ND>(step);
}

// ---------------------------------------------------------------------------------------------------------
// One tile step of the backward kernels, hand-ordered (everything that touches LDS is inline asm, so the only waits
// are the counted ones written here):
//     S(t)   = tile_cur . frag^T            NKS chained MFMAs, row fragments read PA k-steps ahead,
//     epi(e)                                 the caller's per-element epilogue of S(t-1), EPK elements per MFMA gap,
//     out   += G(t-1) . tile_prv             ND x 2 MFMAs, transposed fragments read PT chunks ahead - the first PT
//                                            chunks are already requested during the last S gaps, so the second
//                                            product starts without an LDS round trip.
// `mid()` runs after the first PA row reads are in flight (the place for the next tile's DMA issue: its issue cost
// overlaps the LDS latency).  `pack(g0, g1)` converts the finished epilogue into the two A fragments of G(t-1).
// LDS return order is issue order, so `s_waitcnt lgkmcnt(n)` with n = reads issued after the one needed is exact.
// ---------------------------------------------------------------------------------------------------------
template <int IMM>
__device__ __forceinline__ u32x4 ds_read_b128_asm(uint32_t addr) {
  static_assert(IMM >= 0 && IMM < 65536, "LDS offset field is 16 bits");
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return v;
}
template <int IMM>
__device__ __forceinline__ uint32_t ds_read_b32_asm(uint32_t addr) {
  static_assert(IMM >= 0 && IMM < 65536, "LDS offset field is 16 bits");
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm1(u32x4& a) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}

template <int NKS>
struct RowAddr {          // per-lane LDS byte ADDRESSES (ring base included) of the row-fragment reads
  static constexpr int NA = NKS < 8 ? NKS : 8;
  uint32_t a[NA];
  __device__ __forceinline__ void init(const LaneAddr<NKS>& la, const void* ring) {
    const uint32_t b = lds_addr(ring);
#pragma unroll
    for (int j = 0; j < NA; ++j) a[j] = b + (uint32_t)la.a[j];
  }
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// epi(e) -> float : element e (accumulator register index) of the gated previous tile, already masked.
// dma(k)          : called once in every gap k = 0..2*ND-1 of the second product (and for k up to NDMA-1 afterwards if
//                   there are fewer gaps): those gaps carry only two transposed reads, so the caller places the next
//                   tile's LDS-DMA instructions there (k < its DMA count) and any VALU work that does not fit the S gaps.
// acc[f] (+)= tile . frag[f]^T like mma_tile, with the row-fragment reads as inline asm (PA k-steps ahead, counted
// lgkmcnt waits) and one sched_barrier per k-step: hipcc's own schedule of the builtin form leaves the MFMAs waiting on
// `s_waitcnt lgkmcnt(0)` after short read batches.  OFF: compile-time byte offset of the tile in the ring whose LDS
// addresses are in `ra`.
// SWAP: the stationary fragment is the A operand - acc holds S (stationary rows on the registers, streamed rows on the lanes).
template <int NKS, int RF, int OFF, bool SWAP = false>
__device__ __forceinline__ void mma_tile_asm(const RowAddr<NKS>& ra, const bf16x8 (&frag)[RF][NKS], f32x16 (&acc)[RF]) {
  constexpr int PA = NKS < 4 ? NKS : 4;
  u32x4 a[PA + 1];
  auto issue_a = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    a[ks % (PA + 1)] = ds_read_b128_asm<OFF + 256 * (ks >> 3)>(ra.a[ks & 7]);
  };
  static_for<PA>(issue_a);
  auto step = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    if constexpr (ks + PA < NKS) issue_a(std::integral_constant<int, ks + PA>{});
    constexpr int a_after = (ks + PA < NKS ? ks + PA : NKS - 1) - ks;
    wait_lgkm1<a_after>(a[ks % (PA + 1)]);
    const bf16x8 av = __builtin_bit_cast(bf16x8, a[ks % (PA + 1)]);
#pragma unroll
    for (int f = 0; f < RF; ++f)
      acc[f] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag[f][ks], av, acc[f], 0, 0, 0)
                    : __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, frag[f][ks], acc[f], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<NKS>(step);
}

// Values read from LDS by the caller's own inline-asm reads (issued BEFORE bwd_tile) must be passed through this after
// the wait that covers them: an asm output looks "ready" to hipcc at the read itself, so ordinary code or non-volatile
// asm consuming it could otherwise be scheduled above the wait.  `ready(n)` in bwd_tile is the place: n = LDS reads
// issued after the caller's.
template <typename V>
__device__ __forceinline__ void redefine(V& v) {
  asm volatile("" : "+v"(v));
}
template <int N, typename... V>
__device__ __forceinline__ void wait_lgkm_values(V&... v) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
  (redefine(v), ...);
}

struct EpiIdentity {
  __device__ __forceinline__ float operator()(int, float g) const { return g; }
};
struct NoMid {
  __device__ __forceinline__ void operator()() const {}
};

// One stationary fragment set, epilogue spread over BOTH products.  The gated tile G(t-1) enters the second product as two
// A fragments: g0 = elements 0..7, g1 = elements 8..15.  Sweeping the second product as {all column chunks x g0} then
// {all column chunks x g1} means g1 is not needed before the second sweep, so only HALF of the per-element epilogue has to
// finish inside the S gaps; the other half rides the gaps of the first sweep and the next tile's DMA instructions those of
// the second.  Stamped at D = 256 before the split: S phase 1040 cycles for 16 MFMAs (512 in the matrix pipe) - each gap
// carried a whole element (fma, quarter-rate exp, bit test, mask, running sum, convert: ~34 issue cycles against a 32-cycle
// MFMA) - while the gaps of the second product carried two LDS reads and, in 5 of 16, one DMA instruction.
template <int NKS, int ND, int OFF_CUR, int OFF_PRV, int NDMA, typename Ready, typename Epi, typename DmaFn, typename Mid>
__device__ __forceinline__ void tile_step_split(const RowAddr<NKS>& ra, const TrAddr<NKS>& ta, const bf16x8 (&frag)[1][NKS],
                                                f32x16 (&accs)[1], f32x16 (&out)[ND], Ready ready, Epi epi, DmaFn dma, Mid mid) {
  using T = Tile<NKS>;
  constexpr int PA = NKS < 4 ? NKS : 4;
  constexpr int NCH = 2 * ND;                         // transposed chunks (two reads each) in sweep order: c = sweep * ND + dc
  constexpr int PT = NKS < 2 ? 1 : 2;                 // chunks requested ahead (NCH >= 2 always)
  constexpr int ES = (8 + NKS - 1) / NKS;             // elements 0..7 per S gap
  constexpr int EU = (8 + ND - 1) / ND;               // elements 8..15 per gap of the first sweep
  constexpr int T0 = NKS - PT;                        // S step after which chunk 0 is requested
  u32x4 a[PA + 1];
  u32x2 r[PT + 1][2];
  uint32_t pk[8];
  float even = 0.f;
  auto issue_a = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    a[ks % (PA + 1)] = ds_read_b128_asm<OFF_CUR + 256 * (ks >> 3)>(ra.a[ks & 7]);
  };
  auto issue_t = [&](auto c_c) {
    constexpr int c = decltype(c_c)::value, dc = c % ND, sw = c / ND;
    constexpr int imm = OFF_PRV + 256 * (dc >> 2) + sw * 16 * T::ROW_BYTES;
    u32x2* q = r[c % (PT + 1)];
    q[0] = ds_read_tr_asm<imm>(ta.t[0][dc & 3]);
    q[1] = ds_read_tr_asm<imm>(ta.t[1][dc & 3]);
  };
  auto element = [&](int e) {
    const float g = epi(e);
    if (e & 1) pk[e >> 1] = cvt_pk_bf16(even, g); else even = g;
  };
  static_for<PA>(issue_a);
  ready(std::integral_constant<int, PA>{});
  auto s_step = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    if constexpr (ks + PA < NKS) issue_a(std::integral_constant<int, ks + PA>{});
    constexpr int a_after = (ks + PA < NKS ? ks + PA : NKS - 1) - ks;
    constexpr int t_chunks = ks > T0 ? ks - T0 : 0;
    wait_lgkm1<a_after + 2 * t_chunks>(a[ks % (PA + 1)]);
    const bf16x8 av = __builtin_bit_cast(bf16x8, a[ks % (PA + 1)]);
    accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, frag[0][ks], accs[0], 0, 0, 0);
#pragma unroll
    for (int e = ks * ES; e < ks * ES + ES && e < 8; ++e) element(e);
    if constexpr (ks >= T0) issue_t(std::integral_constant<int, ks - T0>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<NKS>(s_step);
  mid();
  bf16x8 g0, g1;
  {
    const u32x4 g0v = {pk[0], pk[1], pk[2], pk[3]};
    g0 = __builtin_bit_cast(bf16x8, g0v);
  }
  auto t_step = [&](auto c_c) {
    constexpr int c = decltype(c_c)::value, dc = c % ND, sw = c / ND;
    if constexpr (c + PT < NCH) issue_t(std::integral_constant<int, c + PT>{});
    constexpr int ahead = (NCH - 1 - c) < PT ? (NCH - 1 - c) : PT;
    u32x2* q = r[c % (PT + 1)];
    wait_lgkm2<2 * ahead>(q[0], q[1]);
    const u32x4 bv = {q[0].x, q[0].y, q[1].x, q[1].y};
    if constexpr (sw == 0) {
      out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, __builtin_bit_cast(bf16x8, bv), out[dc], 0, 0, 0);
#pragma unroll
      for (int e = 8 + dc * EU; e < 8 + dc * EU + EU && e < 16; ++e) element(e);
    } else {
      if constexpr (dc == 0) {
        const u32x4 g1v = {pk[4], pk[5], pk[6], pk[7]};
        g1 = __builtin_bit_cast(bf16x8, g1v);
      }
      out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, __builtin_bit_cast(bf16x8, bv), out[dc], 0, 0, 0);
      dma(std::integral_constant<int, dc>{});
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<NCH>(t_step);
  auto rest = [&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    if constexpr (k >= ND) dma(k_c);
  };
  static_for<NDMA>(rest);
}

// The split tile step with the NEXT tile's first row-fragment reads issued from its last gaps, for loops whose ring barrier sits
// in the MIDDLE of the step (`mid`, between the two products) instead of at the top.  With the barrier at the top every tile
// began with an empty matrix pipe: counted DMA wait, barrier, then a full LDS round trip before the first MFMA (stamped: ~410
// cycles of loop top + ~150 of wait / barrier per ~2250-cycle tile).  Here the barrier that publishes tile t + 1 is crossed
// while tile t's second product still has its MFMAs to issue, the first PA fragments of tile t + 1 are requested in the last PA
// gaps of that product, and the next step starts with them in flight.
//   a      : in/out, the PA prefetched fragments (k-steps 0..PA-1) of the tile at OFF_CUR on entry, of the tile at OFF_NXT on exit
//   tail() : the caller's own LDS reads for the NEXT step (NW of them), issued right before the prefetch; `ready(PA)` of the next
//            step waits for them (exactly PA reads are issued after them)
// Counted waits: the transposed reads of chunk c are issued at the start of step c - PT; the prefetch reads are issued in the
// steps F .. NCH - 1 of the second sweep after their MFMA (PP per step; the tail reads in step F, before the first), so a
// wait in step c also lets the prefetch / tail reads of steps c - PT .. c - 1 stay outstanding.
template <int NKS, int ND, int NW>
struct PfCount {
  static constexpr int PA = NKS < 4 ? NKS : 4;
  static constexpr int NCH = 2 * ND;
  static constexpr int PT = NKS < 2 ? 1 : 2;        // transposed chunks requested ahead (3 was tried: -2 %, +-0)
  // first step that carries prefetch reads: inside the SECOND sweep (the caller's tail reads may overwrite registers the
  // epilogue elements of the first sweep still read), as late as the PA reads allow
  static constexpr int F = (NCH - PA) > ND ? (NCH - PA) : ND;
  static constexpr int PP = (PA + (NCH - F) - 1) / (NCH - F);           // prefetch reads per step
  static constexpr int first(int st) { return (st - F) * PP; }          // index of the first prefetch read of step st
  static constexpr int count(int st) {                                   // prefetch reads issued in step st
    if (st < F) return 0;
    const int left = PA - first(st);
    return left <= 0 ? 0 : (left < PP ? left : PP);
  }
  static constexpr int extra(int c) {        // prefetch / tail reads issued after chunk c's reads and before its wait
    int lo = c - PT < 0 ? 0 : c - PT, n = 0;
    for (int st = lo; st < c; ++st) n += count(st) + (st == F ? NW : 0);
    return n;
  }
};

template <int NKS, int ND, int OFF_CUR, int OFF_PRV, int OFF_NXT, int NDMA, int NW, typename Ready, typename Epi, typename DmaFn,
          typename Mid, typename Tail>
__device__ __forceinline__ void tile_step_pf(const RowAddr<NKS>& ra, const TrAddr<NKS>& ta, const bf16x8 (&frag)[1][NKS],
                                             f32x16 (&accs)[1], f32x16 (&out)[ND], u32x4 (&a)[(NKS < 4 ? NKS : 4) + 1], Ready ready,
                                             Epi epi, DmaFn dma, Mid mid, Tail tail) {
  using T = Tile<NKS>;
  using C = PfCount<NKS, ND, NW>;
  constexpr int PA = C::PA, NCH = C::NCH, PT = C::PT;
  constexpr int ES = (8 + NKS - 1) / NKS;
  constexpr int EU = (8 + ND - 1) / ND;
  constexpr int T0 = NKS - PT;
  u32x2 r[PT + 1][2];
  uint32_t pk[8];
  float even = 0.f;
  auto issue_a = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    a[ks % (PA + 1)] = ds_read_b128_asm<OFF_CUR + 256 * (ks >> 3)>(ra.a[ks & 7]);
  };
  auto issue_t = [&](auto c_c) {
    constexpr int c = decltype(c_c)::value, dc = c % ND, sw = c / ND;
    constexpr int imm = OFF_PRV + 256 * (dc >> 2) + sw * 16 * T::ROW_BYTES;
    u32x2* q = r[c % (PT + 1)];
    q[0] = ds_read_tr_asm<imm>(ta.t[0][dc & 3]);
    q[1] = ds_read_tr_asm<imm>(ta.t[1][dc & 3]);
  };
  auto element = [&](int e) {
    const float g = epi(e);
    if (e & 1) pk[e >> 1] = cvt_pk_bf16(even, g); else even = g;
  };
  ready(std::integral_constant<int, PA>{});
  auto s_step = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    if constexpr (ks + PA < NKS) issue_a(std::integral_constant<int, ks + PA>{});
    constexpr int a_after = (ks + PA < NKS ? ks + PA : NKS - 1) - ks;
    constexpr int t_chunks = ks > T0 ? ks - T0 : 0;
    wait_lgkm1<a_after + 2 * t_chunks>(a[ks % (PA + 1)]);
    const bf16x8 av = __builtin_bit_cast(bf16x8, a[ks % (PA + 1)]);
    accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, frag[0][ks], accs[0], 0, 0, 0);
#pragma unroll
    for (int e = ks * ES; e < ks * ES + ES && e < 8; ++e) element(e);
    if constexpr (ks >= T0) issue_t(std::integral_constant<int, ks - T0>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<NKS>(s_step);
  mid();
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 g0, g1;
  {
    const u32x4 g0v = {pk[0], pk[1], pk[2], pk[3]};
    g0 = __builtin_bit_cast(bf16x8, g0v);
  }
  auto t_step = [&](auto c_c) {
    constexpr int c = decltype(c_c)::value, dc = c % ND, sw = c / ND;
    if constexpr (c + PT < NCH) issue_t(std::integral_constant<int, c + PT>{});
    constexpr int ahead = (NCH - 1 - c) < PT ? (NCH - 1 - c) : PT;
    u32x2* q = r[c % (PT + 1)];
    wait_lgkm2<2 * ahead + C::extra(c)>(q[0], q[1]);
    const u32x4 bv = {q[0].x, q[0].y, q[1].x, q[1].y};
    if constexpr (sw == 0) {
      out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, __builtin_bit_cast(bf16x8, bv), out[dc], 0, 0, 0);
#pragma unroll
      for (int e = 8 + dc * EU; e < 8 + dc * EU + EU && e < 16; ++e) element(e);
    } else {
      if constexpr (dc == 0) {
        const u32x4 g1v = {pk[4], pk[5], pk[6], pk[7]};
        g1 = __builtin_bit_cast(bf16x8, g1v);
      }
      out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, __builtin_bit_cast(bf16x8, bv), out[dc], 0, 0, 0);
      dma(std::integral_constant<int, dc>{});
    }
    if constexpr (c >= C::F) {
      if constexpr (c == C::F) tail();
      auto pf = [&](auto j_c) {
        constexpr int k = C::first(c) + decltype(j_c)::value;
        a[k] = ds_read_b128_asm<OFF_NXT + 256 * (k >> 3)>(ra.a[k & 7]);
      };
      static_for<C::count(c)>(pf);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<NCH>(t_step);
  auto rest = [&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    if constexpr (k >= ND) dma(k_c);
  };
  static_for<NDMA>(rest);
}

// the prologue of such a loop: the PA fragments of the first tile
template <int NKS, int OFF>
__device__ __forceinline__ void prefetch_first(const RowAddr<NKS>& ra, u32x4 (&a)[(NKS < 4 ? NKS : 4) + 1]) {
  constexpr int PA = NKS < 4 ? NKS : 4;
  auto f = [&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    a[k] = ds_read_b128_asm<OFF + 256 * (k >> 3)>(ra.a[k & 7]);
  };
  static_for<PA>(f);
}

// RF = 1: one stationary fragment set (the backward kernels).  RF = 2: two sets sharing every row-fragment read (the
// fused forward: s = q.n and f = p.n); the per-element epilogue is then split over the two MFMA gaps of a k-step:
// epi(e) after the first MFMA, epi2(e, value) after the second.
template <int NKS, int ND, int OFF_CUR, int OFF_PRV, int NDMA, int RF, typename Ready, typename Epi, typename DmaFn,
          typename Epi2 = EpiIdentity, typename Mid = NoMid>
__device__ __forceinline__ void tile_step(const RowAddr<NKS>& ra, const TrAddr<NKS>& ta, const bf16x8 (&frag)[RF][NKS],
                                          f32x16 (&accs)[RF], f32x16 (&out)[ND], Ready ready, Epi epi, DmaFn dma,
                                          Epi2 epi2 = Epi2{}, Mid mid = Mid{}) {
  if constexpr (RF == 1) {          // one stationary set: the split order (what follows is the RF = 2 order)
    tile_step_split<NKS, ND, OFF_CUR, OFF_PRV, NDMA>(ra, ta, frag, accs, out, ready, epi, dma, mid);
    return;
  }
  using T = Tile<NKS>;
  constexpr int PA = NKS < 4 ? NKS : 4;
  constexpr int PT = ND < 2 ? ND : 2;
  constexpr int EPK = (16 + NKS - 1) / NKS;
  constexpr int T0 = NKS - PT;             // S step after which transposed chunk 0 is requested
  u32x4 a[PA + 1];
  u32x2 r[PT + 1][4];
  uint32_t pk[8];                          // bf16 pairs of the gated tile: pk[0..3] = k-step 0, pk[4..7] = k-step 1
  float even = 0.f;
  auto issue_a = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    a[ks % (PA + 1)] = ds_read_b128_asm<OFF_CUR + 256 * (ks >> 3)>(ra.a[ks & 7]);
  };
  auto issue_t = [&](auto dc_c) {
    constexpr int dc = decltype(dc_c)::value;
    constexpr int imm = OFF_PRV + 256 * (dc >> 2), s1 = 16 * T::ROW_BYTES;
    u32x2* q = r[dc % (PT + 1)];
    q[0] = ds_read_tr_asm<imm>(ta.t[0][dc & 3]);
    q[1] = ds_read_tr_asm<imm>(ta.t[1][dc & 3]);
    q[2] = ds_read_tr_asm<imm + s1>(ta.t[0][dc & 3]);
    q[3] = ds_read_tr_asm<imm + s1>(ta.t[1][dc & 3]);
  };
  static_for<PA>(issue_a);
  ready(std::integral_constant<int, PA>{});
  auto s_step = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    if constexpr (ks + PA < NKS) issue_a(std::integral_constant<int, ks + PA>{});
    constexpr int a_after = (ks + PA < NKS ? ks + PA : NKS - 1) - ks;
    constexpr int t_chunks = ks > T0 ? ks - T0 : 0;
    wait_lgkm1<a_after + 4 * t_chunks>(a[ks % (PA + 1)]);
    const bf16x8 av = __builtin_bit_cast(bf16x8, a[ks % (PA + 1)]);
    accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, frag[0][ks], accs[0], 0, 0, 0);
    if constexpr (RF == 1) {
#pragma unroll
      for (int e = ks * EPK; e < ks * EPK + EPK && e < 16; ++e) {
        const float g = epi(e);
        if (e & 1) pk[e >> 1] = cvt_pk_bf16(even, g); else even = g;
      }
    } else {
      float half_done[EPK];
#pragma unroll
      for (int e = ks * EPK; e < ks * EPK + EPK && e < 16; ++e) half_done[e - ks * EPK] = epi(e);
      __builtin_amdgcn_sched_barrier(0);
      accs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, frag[1][ks], accs[1], 0, 0, 0);
#pragma unroll
      for (int e = ks * EPK; e < ks * EPK + EPK && e < 16; ++e) {
        const float g = epi2(e, half_done[e - ks * EPK]);
        if (e & 1) pk[e >> 1] = cvt_pk_bf16(even, g); else even = g;
      }
    }
    if constexpr (ks >= T0) issue_t(std::integral_constant<int, ks - T0>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<NKS>(s_step);
  mid();                                   // (phase-timing hook of the stamped builds; nothing in the product)
  const u32x4 g0v = {pk[0], pk[1], pk[2], pk[3]}, g1v = {pk[4], pk[5], pk[6], pk[7]};
  const bf16x8 g0 = __builtin_bit_cast(bf16x8, g0v), g1 = __builtin_bit_cast(bf16x8, g1v);
  auto t_step = [&](auto dc_c) {
    constexpr int dc = decltype(dc_c)::value;
    if constexpr (dc + PT < ND) issue_t(std::integral_constant<int, dc + PT>{});
    constexpr int ahead = (ND - 1 - dc) < PT ? (ND - 1 - dc) : PT;
    u32x2* q = r[dc % (PT + 1)];
    wait_lgkm<4 * ahead>(q[0], q[1], q[2], q[3]);
    const u32x4 b0 = {q[0].x, q[0].y, q[1].x, q[1].y}, b1 = {q[2].x, q[2].y, q[3].x, q[3].y};
    out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, __builtin_bit_cast(bf16x8, b0), out[dc], 0, 0, 0);
    dma(std::integral_constant<int, 2 * dc>{});
    __builtin_amdgcn_sched_barrier(0);
    out[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, __builtin_bit_cast(bf16x8, b1), out[dc], 0, 0, 0);
    dma(std::integral_constant<int, 2 * dc + 1>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<ND>(t_step);
  // callbacks beyond the 2*ND gaps (narrow feature dims have few gaps)
  auto rest = [&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    if constexpr (k >= 2 * ND) dma(k_c);
  };
  static_for<NDMA>(rest);
}

template <int NKS, int ND, int OFF_CUR, int OFF_PRV, int NDMA, typename Ready, typename Epi, typename DmaFn>
__device__ __forceinline__ void bwd_tile(const RowAddr<NKS>& ra, const TrAddr<NKS>& ta, const bf16x8 (&frag)[1][NKS],
                                         f32x16& acc, f32x16 (&out)[ND], Ready ready, Epi epi, DmaFn dma) {
  f32x16 accs[1] = {acc};
  tile_step<NKS, ND, OFF_CUR, OFF_PRV, NDMA, 1>(ra, ta, frag, accs, out, ready, epi, dma);
  acc = accs[0];
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
