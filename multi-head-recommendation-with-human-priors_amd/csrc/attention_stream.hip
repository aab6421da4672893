// HSTU pointwise-gated attention for LONG sequences: the streamed operand of every product moves through a two-slot LDS
// ring, so the kernels need 16 KB of LDS at any sequence length (the resident form of attention.hip keeps whole
// [L, head_dim] operand images in LDS: one workgroup per CU at L = 512 x head_dim 64, nothing beyond L = 640).
//
//   forward   : one workgroup per (sequence, head, block of 128 queries): 4 waves x 32 queries keep their Q fragments and
//               O^T accumulators in registers; the K / V tiles of key blocks 0 .. last query block stream through LDS
//               (SiLU applied while staging), shared by the four waves.
//   backward A: one workgroup per (sequence, head, block of 128 keys): K / V fragments and the dK^T / dV^T accumulators in
//               registers, the Q / dO tiles of the query blocks at or after the key block stream through LDS.
//   backward B: one workgroup per (sequence, head, block of 128 queries): Q / dO fragments and dQ accumulators in
//               registers, K / V tiles stream.
// Same tile maths, masks and 1/L scaling as attention.hip (reference model/IDNet/hstu.py:137-160); gradient tiles leave
// through the same wave-private scratch (store_grad_tile).  No atomics: bitwise reproducible.
//
// Ring protocol: every thread loads its 16-byte chunks of tile t + 1 into registers right after the barrier that publishes
// tile t (the loads fly under tile t's MFMAs and gate arithmetic), and stores them - activated - into the other slot at the
// top of the next iteration; one barrier per tile.  Blocks of padding keys are skipped by the whole workgroup.
#include "mhr_common.h"
#include "stream_gemm.h"
#include "attn_tiles.h"

namespace {

using namespace attn;

// (sequence, head) of a flat index, XCD-aware like attention.hip: the heads of a sequence share an L2
__device__ __forceinline__ void decode_bh(int i, int n_bh, int n_heads, int& b, int& head) {
  const int full = (n_bh / (8 * n_heads)) * (8 * n_heads);
  if (i < full) {
    const int xcd = i & 7, slot = i >> 3;
    head = slot % n_heads;
    b = (slot / n_heads) * 8 + xcd;
  } else {
    b = i / n_heads;
    head = i % n_heads;
  }
}

// One streamed 32-row tile of a [L, hd] operand: global -> registers (issued early) -> SiLU -> swizzled LDS image
template <int NKS>
struct TileRegs {
  using T = sg::Tile<NKS>;
  static constexpr int PER = (T::CHUNKS + 255) / 256;
  bf16x8 v[PER];
  __device__ __forceinline__ void load(const bf16_t* src, int64_t stride, int tile, int L, int hd) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = threadIdx.x + i * 256;
      const int row = id / T::CH, c = id % T::CH, m = tile * 32 + row;
      v[i] = zero8();
      if (id < T::CHUNKS && m < L && c * 8 < hd) v[i] = *reinterpret_cast<const bf16x8*>(src + (int64_t)m * stride + c * 8);
    }
  }
  __device__ __forceinline__ void save(bf16_t* dst, int64_t stride, int tile, int L, int hd, bool do_silu) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = threadIdx.x + i * 256;
      const int row = id / T::CH, c = id % T::CH, m = tile * 32 + row;
      if (id < T::CHUNKS && m < L && c * 8 < hd) *reinterpret_cast<bf16x8*>(dst + (int64_t)m * stride + c * 8) = do_silu ? silu8(v[i]) : v[i];
    }
  }
  __device__ __forceinline__ void store(unsigned char* dst, bool do_silu) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = threadIdx.x + i * 256;
      if (id < T::CHUNKS) {
        const int row = id / T::CH, c = id % T::CH;
        *reinterpret_cast<bf16x8*>(dst + T::off(row, c)) = do_silu ? silu8(v[i]) : v[i];
      }
    }
  }
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NKS, int ND>
__global__ __launch_bounds__(256) void hstu_attn_fwd_stream_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                   const bf16_t* __restrict__ v, int64_t stride,
                                                                   const uint8_t* __restrict__ key_valid, bf16_t* __restrict__ out,
                                                                   int64_t out_stride, bf16_t* act_q, bf16_t* act_k, bf16_t* act_v,
                                                                   int64_t act_stride, int L, int n_heads, int hd, int apply_silu,
                                                                   float inv_n, int n_bh) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = (L + 31) >> 5, nsb = (nb + 3) >> 2;
  unsigned char* Kb = smem;                        // 2 slots
  unsigned char* Vb = smem + 2 * T::BYTES;         // 2 slots
  uint32_t* vmask = reinterpret_cast<uint32_t*>(smem + 4 * T::BYTES);

  const int qs = nsb - 1 - (int)(blockIdx.x / n_bh);              // heavy (late) query blocks first
  int b, head;
  decode_bh(blockIdx.x % n_bh, n_bh, n_heads, b, head);
  const int64_t row0 = (int64_t)b * L;
  const bf16_t* qp = q + row0 * stride + head * hd;
  const bf16_t* kp = k + row0 * stride + head * hd;
  const bf16_t* vp = v + row0 * stride + head * hd;
  const bool do_silu = apply_silu != 0;
  // saved activations (optional): a key tile is written by the workgroup whose own query blocks contain it
  bf16_t* aq = act_q ? act_q + row0 * act_stride + head * hd : nullptr;
  bf16_t* ak = act_k ? act_k + row0 * act_stride + head * hd : nullptr;
  bf16_t* av = act_v ? act_v + row0 * act_stride + head * hd : nullptr;
  build_valid_mask(vmask, key_valid + row0, L, nb);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int qb = qs * 4 + wave, qrow = qb * 32 + r;
  const bool active = qb < nb;
  sg::LaneAddr<NKS> la;
  la.init(lane);
  bf16x8 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int koff = ks * 16 + 8 * half;
    qf[ks] = load_frag(qp, stride, active ? qrow : L, L, koff, hd);
    if (do_silu) qf[ks] = silu8(qf[ks]);
    if (aq && active && qrow < L && koff < hd) *reinterpret_cast<bf16x8*>(aq + (int64_t)qrow * act_stride + koff) = qf[ks];
  }
  f32x16 o[ND];
#pragma unroll
  for (int dc = 0; dc < ND; ++dc) o[dc] = zero16();

  const int kb_end = min(nb - 1, qs * 4 + 3);
  TileRegs<NKS> rk, rv;
  if (ak) {                                                        // this workgroup's own key rows
    for (int t = qs * 4; t <= kb_end; ++t) {
      rk.load(kp, stride, t, L, hd);
      rv.load(vp, stride, t, L, hd);
      rk.save(ak, act_stride, t, L, hd, do_silu);
      rv.save(av, act_stride, t, L, hd, do_silu);
    }
  }
  int kb = 0;
  while (kb <= kb_end && vmask[kb] == 0) ++kb;                     // blocks of padding keys contribute nothing
  if (kb <= kb_end) {
    rk.load(kp, stride, kb, L, hd);
    rv.load(vp, stride, kb, L, hd);
  }
  int slot = 0;
  while (kb <= kb_end) {
    rk.store(Kb + slot * T::BYTES, do_silu);
    rv.store(Vb + slot * T::BYTES, do_silu);
    __syncthreads();
    int nx = kb + 1;
    while (nx <= kb_end && vmask[nx] == 0) ++nx;
    if (nx <= kb_end) {
      rk.load(kp, stride, nx, L, hd);
      rv.load(vp, stride, nx, L, hd);
    }
    if (active && kb <= qb) {
      const uint32_t vm = vmask[kb];
      const unsigned char* kt = Kb + slot * T::BYTES;
      const unsigned char* vt = Vb + slot * T::BYTES;
      f32x16 s = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(kt, ks), qf[ks], s, 0, 0, 0);   // S^T: rows = keys, cols = queries
      if (kb < qb && vm == 0xFFFFFFFFu) {
#pragma unroll
        for (int g = 0; g < 16; ++g) s[g] *= fast_sigmoid(s[g]);
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int kl = crow(g, half);
          const bool ok = ((vm >> kl) & 1u) && (kb * 32 + kl <= qrow);
          s[g] = ok ? s[g] * fast_sigmoid(s[g]) : 0.f;
        }
      }
      bf16x8 p0, p1;
      pack_acc(s, p0, p1);
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        o[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_tr(vt, dc, 0), p0, o[dc], 0, 0, 0);   // O^T += V^T . P^T
        o[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_tr(vt, dc, 1), p1, o[dc], 0, 0, 0);
      }
    }
    kb = nx;
    slot ^= 1;
  }
  if (active && qrow < L) {
    bf16_t* orow = out + (row0 + qrow) * out_stride + head * hd;
#pragma unroll
    for (int dc = 0; dc < ND; ++dc)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = dc * 32 + 8 * g4 + 4 * half;
        if (d0 < hd) {
          bf16x4 w = {(bf16_t)(o[dc][4 * g4] * inv_n), (bf16_t)(o[dc][4 * g4 + 1] * inv_n), (bf16_t)(o[dc][4 * g4 + 2] * inv_n),
                      (bf16_t)(o[dc][4 * g4 + 3] * inv_n)};
          *reinterpret_cast<bf16x4*>(orow + d0) = w;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------
// backward A: dK, dV of a block of 128 keys
// ------------------------------------------------------------------------------------------
template <int NKS, int ND>
__global__ __launch_bounds__(256, NKS <= 4 ? 2 : 1) void hstu_attn_bwd_kv_stream_kernel(
    const bf16_t* __restrict__ q_pre, const bf16_t* __restrict__ k_pre, const bf16_t* __restrict__ v_pre, int64_t stride,
    const bf16_t* __restrict__ act_q, const bf16_t* __restrict__ act_k, const bf16_t* __restrict__ act_v, int64_t act_stride,
    const uint8_t* __restrict__ key_valid, const bf16_t* __restrict__ d_out, int64_t do_stride, bf16_t* __restrict__ dk,
    bf16_t* __restrict__ dv, int64_t d_stride, int L, int n_heads, int hd, int apply_silu, float inv_n, int n_bh) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = (L + 31) >> 5;
  unsigned char* Qb = smem;                        // 2 slots: activated Q tiles
  unsigned char* Db = smem + 2 * T::BYTES;         // 2 slots: dO tiles
  uint32_t* vmask = reinterpret_cast<uint32_t*>(smem + 4 * T::BYTES);
  float* gscratch = reinterpret_cast<float*>(smem + 4 * T::BYTES + ((nb * 4 + 15) & ~15)) + (threadIdx.x >> 6) * (32 * GS);

  const int ksb = (int)(blockIdx.x / n_bh);                        // early key blocks (the heavy ones) first
  int b, head;
  decode_bh(blockIdx.x % n_bh, n_bh, n_heads, b, head);
  const int64_t row0 = (int64_t)b * L;
  const int hoff = head * hd;
  // operands: saved activations when the forward kept them, else silu(pre) recomputed while staging / loading
  const bool redo = act_q == nullptr, chain = apply_silu != 0;
  const int64_t a_stride = redo ? stride : act_stride;
  const bf16_t* qp = redo ? q_pre + row0 * stride + hoff : act_q + row0 * act_stride + hoff;
  const bf16_t* kp = redo ? k_pre + row0 * stride + hoff : act_k + row0 * act_stride + hoff;
  const bf16_t* vp = redo ? v_pre + row0 * stride + hoff : act_v + row0 * act_stride + hoff;
  const bf16_t* qpre_h = q_pre + (chain ? row0 * stride + hoff : 0);
  const bf16_t* kpre_h = k_pre + (chain ? row0 * stride + hoff : 0);
  const bf16_t* vpre_h = v_pre + (chain ? row0 * stride + hoff : 0);
  const bf16_t* dop = d_out + row0 * do_stride + hoff;
  build_valid_mask(vmask, key_valid + row0, L, nb);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int kb = ksb * 4 + wave, key = kb * 32 + r;
  const bool active = kb < nb;
  sg::LaneAddr<NKS> la;
  la.init(lane);
  const uint32_t vm_kb = active ? vmask[kb] : 0u;
  const bool kvalid = (vm_kb >> r) & 1u;
  bf16x8 kf[NKS], vf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    kf[ks] = load_frag(kp, a_stride, active ? key : L, L, ks * 16 + 8 * half, hd);
    vf[ks] = load_frag(vp, a_stride, active ? key : L, L, ks * 16 + 8 * half, hd);
    if (redo) {
      kf[ks] = silu8(kf[ks]);
      vf[ks] = silu8(vf[ks]);
    }
  }
  f32x16 dvacc[ND], dkacc[ND];
#pragma unroll
  for (int dc = 0; dc < ND; ++dc) {
    dvacc[dc] = zero16();
    dkacc[dc] = zero16();
  }
  // any valid key in the workgroup's four blocks?  (uniform: otherwise nothing is streamed and zeros are stored)
  uint32_t any_valid = 0;
  for (int w = 0; w < 4; ++w) any_valid |= (ksb * 4 + w < nb) ? vmask[ksb * 4 + w] : 0u;

  TileRegs<NKS> rq, rd;
  int qb = ksb * 4, slot = 0;
  if (any_valid && qb < nb) {
    rq.load(qp, a_stride, qb, L, hd);
    rd.load(dop, do_stride, qb, L, hd);
  }
  while (any_valid && qb < nb) {
    rq.store(Qb + slot * T::BYTES, redo);
    rd.store(Db + slot * T::BYTES, false);
    __syncthreads();
    if (qb + 1 < nb) {
      rq.load(qp, a_stride, qb + 1, L, hd);
      rd.load(dop, do_stride, qb + 1, L, hd);
    }
    if (active && qb >= kb && vm_kb != 0) {
      const unsigned char* qt = Qb + slot * T::BYTES;
      const unsigned char* dot = Db + slot * T::BYTES;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(qt, ks), kf[ks], s, 0, 0, 0);      // S: rows = queries, cols = keys
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(dot, ks), vf[ks], dp, 0, 0, 0);   // dP = dO . V^T
      }
      if (qb > kb && vm_kb == 0xFFFFFFFFu) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const float x = s[g];
          const float sig = fast_sigmoid(x);
          s[g] = x * sig;
          dp[g] = dp[g] * sig * (1.0f + x * (1.0f - sig));
        }
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int qi = qb * 32 + crow(g, half);
          const bool ok = kvalid && key <= qi;
          const float x = s[g];
          const float sig = fast_sigmoid(x);
          s[g] = ok ? x * sig : 0.f;
          dp[g] = ok ? dp[g] * sig * (1.0f + x * (1.0f - sig)) : 0.f;
        }
      }
      bf16x8 pa0, pa1, da0, da1;
      pack_acc(s, pa0, pa1);
      pack_acc(dp, da0, da1);
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        dvacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0, la.read_tr(dot, dc, 0), dvacc[dc], 0, 0, 0);   // dV += P^T . dO
        dvacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, la.read_tr(dot, dc, 1), dvacc[dc], 0, 0, 0);
        dkacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da0, la.read_tr(qt, dc, 0), dkacc[dc], 0, 0, 0);    // dK += dS^T . Q
        dkacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da1, la.read_tr(qt, dc, 1), dkacc[dc], 0, 0, 0);
      }
    }
    ++qb;
    slot ^= 1;
  }
  if (active) {
    // pre-activations of the SiLU' chain: fetched here, not held across the (long) tile loop - their registers are what
    // lets two workgroups share a CU
    GradPre kpre[ND], vpre[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      kpre[dc] = prefetch_pre(kpre_h, stride, kb * 32, L, dc * 32, hd, chain, lane);
      vpre[dc] = prefetch_pre(vpre_h, stride, kb * 32, L, dc * 32, hd, chain, lane);
    }
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      store_grad_tile(gscratch, dvacc[dc], inv_n, dv + row0 * d_stride + hoff, d_stride, vpre[dc], kb * 32, L, dc * 32, hd, chain, lane);
      store_grad_tile(gscratch, dkacc[dc], inv_n, dk + row0 * d_stride + hoff, d_stride, kpre[dc], kb * 32, L, dc * 32, hd, chain, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward B: dQ of a block of 128 queries
// ------------------------------------------------------------------------------------------
template <int NKS, int ND>
__global__ __launch_bounds__(256, NKS <= 4 ? 2 : 1) void hstu_attn_bwd_q_stream_kernel(
    const bf16_t* __restrict__ q_pre, const bf16_t* __restrict__ k_pre, const bf16_t* __restrict__ v_pre, int64_t stride,
    const bf16_t* __restrict__ act_q, const bf16_t* __restrict__ act_k, const bf16_t* __restrict__ act_v, int64_t act_stride,
    const uint8_t* __restrict__ key_valid, const bf16_t* __restrict__ d_out, int64_t do_stride, bf16_t* __restrict__ dq,
    int64_t d_stride, int L, int n_heads, int hd, int apply_silu, float inv_n, int n_bh) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = (L + 31) >> 5, nsb = (nb + 3) >> 2;
  unsigned char* Kb = smem;
  unsigned char* Vb = smem + 2 * T::BYTES;
  uint32_t* vmask = reinterpret_cast<uint32_t*>(smem + 4 * T::BYTES);
  float* gscratch = reinterpret_cast<float*>(smem + 4 * T::BYTES + ((nb * 4 + 15) & ~15)) + (threadIdx.x >> 6) * (32 * GS);

  const int qs = nsb - 1 - (int)(blockIdx.x / n_bh);
  int b, head;
  decode_bh(blockIdx.x % n_bh, n_bh, n_heads, b, head);
  const int64_t row0 = (int64_t)b * L;
  const int hoff = head * hd;
  // operands: saved activations when the forward kept them, else silu(pre) recomputed while staging / loading
  const bool redo = act_q == nullptr, chain = apply_silu != 0;
  const int64_t a_stride = redo ? stride : act_stride;
  const bf16_t* qp = redo ? q_pre + row0 * stride + hoff : act_q + row0 * act_stride + hoff;
  const bf16_t* kp = redo ? k_pre + row0 * stride + hoff : act_k + row0 * act_stride + hoff;
  const bf16_t* vp = redo ? v_pre + row0 * stride + hoff : act_v + row0 * act_stride + hoff;
  const bf16_t* qpre_h = q_pre + (chain ? row0 * stride + hoff : 0);
  const bf16_t* kpre_h = k_pre + (chain ? row0 * stride + hoff : 0);
  const bf16_t* vpre_h = v_pre + (chain ? row0 * stride + hoff : 0);
  const bf16_t* dop = d_out + row0 * do_stride + hoff;
  build_valid_mask(vmask, key_valid + row0, L, nb);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int qb = qs * 4 + wave, qcol = qb * 32 + r;
  const bool active = qb < nb;
  sg::LaneAddr<NKS> la;
  la.init(lane);
  bf16x8 qf[NKS], dof[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    qf[ks] = load_frag(qp, a_stride, active ? qcol : L, L, ks * 16 + 8 * half, hd);
    if (redo) qf[ks] = silu8(qf[ks]);
    dof[ks] = load_frag(dop, do_stride, active ? qcol : L, L, ks * 16 + 8 * half, hd);
  }
  f32x16 dqacc[ND];
#pragma unroll
  for (int dc = 0; dc < ND; ++dc) dqacc[dc] = zero16();

  const int kb_end = min(nb - 1, qs * 4 + 3);
  TileRegs<NKS> rk, rv;
  int kb = 0, slot = 0;
  while (kb <= kb_end && vmask[kb] == 0) ++kb;
  if (kb <= kb_end) {
    rk.load(kp, a_stride, kb, L, hd);
    rv.load(vp, a_stride, kb, L, hd);
  }
  while (kb <= kb_end) {
    rk.store(Kb + slot * T::BYTES, redo);
    rv.store(Vb + slot * T::BYTES, redo);
    __syncthreads();
    int nx = kb + 1;
    while (nx <= kb_end && vmask[nx] == 0) ++nx;
    if (nx <= kb_end) {
      rk.load(kp, a_stride, nx, L, hd);
      rv.load(vp, a_stride, nx, L, hd);
    }
    if (active && kb <= qb) {
      const uint32_t vm = vmask[kb];
      const unsigned char* kt = Kb + slot * T::BYTES;
      const unsigned char* vt = Vb + slot * T::BYTES;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(kt, ks), qf[ks], s, 0, 0, 0);      // S^T: rows = keys, cols = queries
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(vt, ks), dof[ks], dp, 0, 0, 0);   // dP^T = V . dO^T
      }
      if (kb < qb && vm == 0xFFFFFFFFu) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const float x = s[g];
          const float sig = fast_sigmoid(x);
          dp[g] = dp[g] * sig * (1.0f + x * (1.0f - sig));
        }
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int kl = crow(g, half);
          const bool ok = ((vm >> kl) & 1u) && (kb * 32 + kl <= qcol);
          const float x = s[g];
          const float sig = fast_sigmoid(x);
          dp[g] = ok ? dp[g] * sig * (1.0f + x * (1.0f - sig)) : 0.f;
        }
      }
      bf16x8 a0, a1;
      pack_acc(dp, a0, a1);
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        dqacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, la.read_tr(kt, dc, 0), dqacc[dc], 0, 0, 0);   // dQ += dS . K
        dqacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, la.read_tr(kt, dc, 1), dqacc[dc], 0, 0, 0);
      }
    }
    kb = nx;
    slot ^= 1;
  }
  if (active) {
    GradPre qpre[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) qpre[dc] = prefetch_pre(qpre_h, stride, qb * 32, L, dc * 32, hd, chain, lane);
#pragma unroll
    for (int dc = 0; dc < ND; ++dc)
      store_grad_tile(gscratch, dqacc[dc], inv_n, dq + row0 * d_stride + hoff, d_stride, qpre[dc], qb * 32, L, dc * 32, hd, chain, lane);
  }
}

}  // namespace

// launchers called by the C-ABI entry points of attention.hip (same arguments, already validated)
int mhr_attn_stream_fwd(const void* q, const void* k, const void* v, int64_t row_stride, const uint8_t* key_valid, void* out,
                        void* act_q, void* act_k, void* act_v, int64_t act_stride, int B, int L, int n_heads, int head_dim,
                        int apply_silu, hipStream_t s) {
  attn::AttnShape sh;
  attn::attn_shape(head_dim, sh);
  const int nb = (L + 31) / 32, nsb = (nb + 3) / 4, n_bh = B * n_heads;
  const float inv_n = 1.0f / (float)L;
  const int64_t out_stride = (int64_t)n_heads * head_dim;
#define L_(NKS, ND)                                                                                                          \
  {                                                                                                                          \
    const size_t lds = 4 * (size_t)sg::Tile<NKS>::BYTES + (size_t)nb * 4 + 16;                                               \
    hipLaunchKernelGGL((hstu_attn_fwd_stream_kernel<NKS, ND>), dim3(nsb * n_bh), dim3(256), lds, s, (const bf16_t*)q,        \
                       (const bf16_t*)k, (const bf16_t*)v, row_stride, key_valid, (bf16_t*)out, out_stride, (bf16_t*)act_q,  \
                       (bf16_t*)act_k, (bf16_t*)act_v, act_stride, L, n_heads, head_dim, apply_silu, inv_n, n_bh);           \
  }
  ATTN_DISPATCH(sh, L_);
#undef L_
  return 0;
}

int mhr_attn_stream_bwd(const void* q_pre, const void* k_pre, const void* v_pre, int64_t row_stride, const void* act_q,
                        const void* act_k, const void* act_v, int64_t act_stride, const uint8_t* key_valid, const void* d_out,
                        void* dq, void* dk, void* dv, int64_t d_stride, int B, int L, int n_heads, int head_dim, int apply_silu,
                        hipStream_t s) {
  attn::AttnShape sh;
  attn::attn_shape(head_dim, sh);
  const int nb = (L + 31) / 32, nsb = (nb + 3) / 4, n_bh = B * n_heads;
  const float inv_n = 1.0f / (float)L;
  const int64_t do_stride = (int64_t)n_heads * head_dim;
#define L_(NKS, ND)                                                                                                          \
  {                                                                                                                          \
    const size_t lds = 4 * (size_t)sg::Tile<NKS>::BYTES + (((size_t)nb * 4 + 15) & ~(size_t)15) + 4 * 32 * 36 * sizeof(float); \
    hipLaunchKernelGGL((hstu_attn_bwd_kv_stream_kernel<NKS, ND>), dim3(nsb * n_bh), dim3(256), lds, s, (const bf16_t*)q_pre, \
                       (const bf16_t*)k_pre, (const bf16_t*)v_pre, row_stride, (const bf16_t*)act_q, (const bf16_t*)act_k,   \
                       (const bf16_t*)act_v, act_stride, key_valid, (const bf16_t*)d_out, do_stride, (bf16_t*)dk,           \
                       (bf16_t*)dv, d_stride, L, n_heads, head_dim, apply_silu, inv_n, n_bh);                                \
    hipLaunchKernelGGL((hstu_attn_bwd_q_stream_kernel<NKS, ND>), dim3(nsb * n_bh), dim3(256), lds, s, (const bf16_t*)q_pre,  \
                       (const bf16_t*)k_pre, (const bf16_t*)v_pre, row_stride, (const bf16_t*)act_q, (const bf16_t*)act_k,   \
                       (const bf16_t*)act_v, act_stride, key_valid, (const bf16_t*)d_out, do_stride, (bf16_t*)dq, d_stride, \
                       L, n_heads, head_dim, apply_silu, inv_n, n_bh);                                                       \
  }
  ATTN_DISPATCH(sh, L_);
#undef L_
  return 0;
}
