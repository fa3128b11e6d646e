// Weight-streaming GEMV kernels for M <= 8 tokens (decode): HBM-bound, one pass over the weights.
//
// Replaces, per decode token and layer, the reference's separate launches for RMSNorm
// (transformer_layers.py:115-120), the q/k/v/o and w1/w2/w3 nn.Linear GEMVs (:66,:93,:105-106), RoPE
// (rope.py:13-23), the ring write (cache.py:83-92), the residual adds (:166,:168) and silu*mul.
//
// Structure (cdna_hip_programming.md "GEMV / M<=16 decode weights"): weights go straight HBM -> VGPR
// with 16-byte non-temporal loads in batches of 8 per lane; a wave always has two batches (16 KiB) in
// flight, across the prologue and across unit boundaries (the load cursor runs over the flattened
// (unit, batch) sequence, two batches ahead of the FMAs).  The (optionally RMS-normalised) activation
// vector lives in LDS; its loads are issued before the first weight batch so the prologue finishes under
// the HBM latency of the weights.  A wave owns "units" (a pair of weight rows, or one row for small N)
// strided over the whole grid, so at any instant the chip streams one contiguous weight region.
#include <cstdlib>

#include "common.cuh"

// A second compile with -DGEMV_F16=1 (build_native.py: gemv_f16.o) is the same weight-streaming kernel for fp16 storage:
// v_dot2_f32_f16 for v_dot2c_f32_bf16 and half conversions / rounding points, under launch_gemv_f16 (api.hip: decode steps of
// fp16 models in mi_forward_generic).  Every 16-bit access of gemv_core.cuh goes through the helpers renamed here; the default
// compile is untouched by this block (ISA hash checked) and gemv_core.cuh itself - shared with the frozen decode engine - is
// not edited.
#ifndef GEMV_F16
#define GEMV_F16 0
#endif
#if GEMV_F16
typedef _Float16 gemv_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float gemv_h_to_f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t gemv_h_from_f(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float gemv_h_round(float f) { return (float)(_Float16)f; }
__device__ __forceinline__ uint32_t gemv_h_pack2(float lo, float hi) {
  return (uint32_t)gemv_h_from_f(lo) | ((uint32_t)gemv_h_from_f(hi) << 16);
}
__device__ __forceinline__ float gemv_h_lo(uint32_t u) { return gemv_h_to_f((uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float gemv_h_hi(uint32_t u) { return gemv_h_to_f((uint16_t)(u >> 16)); }
__device__ __forceinline__ float gemv_h_dot2(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(gemv_half2, a), __builtin_bit_cast(gemv_half2, b), acc, false);
}
__device__ __forceinline__ float gemv_h_swiglu(float acc1, float acc3) {  // swiglu_bf with fp16 rounding points
  const float a = gemv_h_round(acc1), b = gemv_h_round(acc3);
  const float s = gemv_h_round(a / (1.0f + expf(-a)));
  return s * b;
}
#define dot2_bf16 gemv_h_dot2
#define bf_lo gemv_h_lo
#define bf_hi gemv_h_hi
#define pack_bf2 gemv_h_pack2
#define bf_round gemv_h_round
#define f_to_bf gemv_h_from_f
#define bf_to_f gemv_h_to_f
#define swiglu_bf gemv_h_swiglu
#define gemv_core gemv_core_f16
#define gemv_max_tokens gemv_max_tokens_f16
#define launch_gemv launch_gemv_f16
#endif

#include "gemv_core.cuh"

namespace {

using namespace gemv_core;

// DMA: the activation rows go to LDS by LDS-DMA (gemv_core.cuh; instantiated for 2..8 tokens, picked when they do not fit the registers)
template <int TT, int MODE, int ROWS, bool DMA>
__global__ __launch_bounds__(256, (TT == 1 ? 4 : (TT <= 3 ? 3 : 2))) void gemv_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemv_body<TT, MODE, ROWS, DMA>(a, smem, blockIdx.x, gridDim.x, blockIdx.y);
}

// Blocks of 5 .. 7 waves for the plain modes at one token (gemv_core.cuh NWV): row counts that 4-wave blocks cannot split evenly
// over the CUs.
template <int MODE, int NWV>
__global__ __launch_bounds__(NWV * 64, 4) void gemv_kernel_nw(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemv_body<1, MODE, 2, false, NWV>(a, smem, blockIdx.x, gridDim.x, blockIdx.y);
}

// MoE down-projection + combine for one token per blockIdx.y (moe.py:28-32 at decode):
// out[t] = bf16(h[t] + R),  R = sum over the token's experts in ascending id of bf16(w_e * bf16(W2_e . g_e)),
// accumulated in bf16 starting from zero.
template <int TOPK>
__global__ __launch_bounds__(256) void moe_w2_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [TOPK][K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = blockIdx.y;
  const int nwaves = gridDim.x * 4;
  const int units = (a.N + 1) >> 1;
  constexpr int U = BATCH / 2;
  const int nch = (a.K + 511) >> 9;
  const int nb = (nch + U - 1) / U;

  int eid[TOPK];
  float ew[TOPK];
  const bf16_t* w2[TOPK];
#pragma unroll
  for (int k = 0; k < TOPK; ++k) {
    eid[k] = a.sel_idx[t * TOPK + k];
    ew[k] = a.sel_w[t * TOPK + k];
  }
  // visit experts in ascending id (moe.py:29 loop order)
#pragma unroll
  for (int i = 0; i < TOPK; ++i)
#pragma unroll
    for (int j = i + 1; j < TOPK; ++j)
      if (eid[j] < eid[i]) {
        const int te = eid[i]; eid[i] = eid[j]; eid[j] = te;
        const float tw = ew[i]; ew[i] = ew[j]; ew[j] = tw;
      }
  int slot_of[TOPK];  // which hidden row belongs to sorted position k
#pragma unroll
  for (int k = 0; k < TOPK; ++k) {
    slot_of[k] = 0;
#pragma unroll
    for (int s = 0; s < TOPK; ++s)
      if (a.sel_idx[t * TOPK + s] == eid[k]) slot_of[k] = s;
    w2[k] = reinterpret_cast<const bf16_t*>(a.expert_tab[eid[k] * 3 + 1]);
  }

  // flattened load cursor over (unit, expert, batch)
  int u = blockIdx.x * 4 + wid;
  int ul = u, kl = 0, jl = 0;
  u32x4 bufA[BATCH], bufB[BATCH];
  auto issue = [&](u32x4 (&buf)[BATCH]) {  // branch-free around the loads (see gemv_core.cuh)
    const bool live = ul < units;
    const bf16_t* base = w2[0];
#pragma unroll
    for (int k = 1; k < TOPK; ++k) base = (kl == k) ? w2[k] : base;
    const int ue = live ? ul : 0;
    RowPair r;
    r.a = live ? base + (size_t)(2 * ue) * a.K : a.x;
    r.b = live ? ((2 * ue + 1 < a.N) ? base + (size_t)(2 * ue + 1) * a.K : r.a) : a.x;
    load_batch<2>(r, live ? jl * U : 0, live ? a.K : 8, live ? lane : 0, buf);
    if (live && ++jl == nb) {
      jl = 0;
      if (++kl == TOPK) {
        kl = 0;
        ul += nwaves;
      }
    }
  };
  // stage the TOPK hidden rows of this token (rows of a.x are [T*TOPK, K], slot-major per token) by LDS-DMA, one row per sorted
  // position: all pieces in flight at once (it was a load -> store loop of K / 2048 L2 round trips), issued BEFORE the two weight
  // batches so that vmcnt(16) = "the rows have landed" leaves the weights in flight
#pragma unroll
  for (int k = 0; k < TOPK; ++k)
    dma_rows_to_lds(a.x + ((size_t)t * TOPK + slot_of[k]) * a.ldx, 0, 1, 1, a.K >> 3, reinterpret_cast<char*>(xs + (size_t)k * a.K));
  issue(bufA);
  issue(bufB);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __syncthreads();

  Acc<1> acc;
  acc.v[0][0] = acc.v[1][0] = 0.f;
  int jc = 0, kc = 0;
  float r0 = 0.f, r1 = 0.f;
  auto step = [&](u32x4 (&buf)[BATCH]) {
    fma_batch<1, 2>(buf, jc * U, xs + (size_t)kc * a.K, a.K, lane, acc);
    issue(buf);
    if (++jc == nb) {
      jc = 0;
      const float y0 = wave_sum(acc.v[0][0]), y1 = wave_sum(acc.v[1][0]);
      acc.v[0][0] = acc.v[1][0] = 0.f;
      float w = ew[0];
#pragma unroll
      for (int k = 1; k < TOPK; ++k) w = (kc == k) ? ew[k] : w;
      r0 = bf_round(r0 + bf_round(w * bf_round(y0)));
      r1 = bf_round(r1 + bf_round(w * bf_round(y1)));
      if (++kc == TOPK) {
        kc = 0;
        if (lane == 0 && u < units) {
          const int n = 2 * u;
          const bf16_t* rs = a.residual + (size_t)t * a.ldo + n;
          bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)t * a.ldo + n;
          o[0] = f_to_bf(bf_to_f(rs[0]) + r0);
          if (n + 1 < a.N) o[1] = f_to_bf(bf_to_f(rs[1]) + r1);
        }
        r0 = r1 = 0.f;
        u += nwaves;
      }
    }
  };
  if (u < units) {  // both steps per trip: see gemv_core.cuh (a step past the last unit stores nothing)
    do {
      step(bufA);
      step(bufB);
    } while (u < units);
  }
}

template <int TT, int MODE, int ROWS, bool DMA>
hipError_t launch_tt(const GemvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  if (lds > 64 * 1024) {  // more than the default dynamic-LDS limit: an opt-in per function AND per device
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<TT, MODE, ROWS, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)GEMV_LDS_BUDGET + 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL((gemv_kernel<TT, MODE, ROWS, DMA>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}
// Rows that fit the prologue's register set (gemv_core.cuh: 4 x 256 pieces with a fused RMSNorm, 8 x 256 without) are staged
// through registers; 2..8 rows that do not, by LDS-DMA.
template <int TT, int MODE, int ROWS>
hipError_t launch_stage(const GemvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  if constexpr (TT > 1) {
    constexpr bool norm_mode = MODE == GEMV_QKV_ROPE || MODE == GEMV_SWIGLU || MODE == GEMV_LOGITS || MODE == GEMV_MOE_W13;  // (gemv_body's kNormMode)
    if ((size_t)TT * (a.K >> 3) > (norm_mode ? 4u : 8u) * 256u) return launch_tt<TT, MODE, ROWS, true>(a, grid, lds, s);
  }
  return launch_tt<TT, MODE, ROWS, false>(a, grid, lds, s);
}
template <int MODE, int ROWS>
hipError_t launch_mode(const GemvArgs& a, int TT, dim3 grid, size_t lds, hipStream_t s) {
  switch (TT) {
    case 1: return launch_stage<1, MODE, ROWS>(a, grid, lds, s);
    case 2: return launch_stage<2, MODE, ROWS>(a, grid, lds, s);
    case 3: return launch_stage<3, MODE, ROWS>(a, grid, lds, s);
    case 4: return launch_stage<4, MODE, ROWS>(a, grid, lds, s);
    case 6: return launch_stage<6, MODE, ROWS>(a, grid, lds, s);
    default: return launch_stage<8, MODE, ROWS>(a, grid, lds, s);
  }
}

template <int MODE>
hipError_t launch_nw(const GemvArgs& a, int nw, dim3 grid, size_t lds, hipStream_t s) {
  switch (nw) {
    case 5: hipLaunchKernelGGL((gemv_kernel_nw<MODE, 5>), grid, dim3(320), lds, s, a); break;
    case 6: hipLaunchKernelGGL((gemv_kernel_nw<MODE, 6>), grid, dim3(384), lds, s, a); break;
    case 7: hipLaunchKernelGGL((gemv_kernel_nw<MODE, 7>), grid, dim3(448), lds, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

int g_gemv_max_blocks = 0;

}  // namespace

int gemv_max_tokens(int K) {
  int t = (int)(GEMV_LDS_BUDGET / ((size_t)K * 2));
  if (t == 5 || t == 7) --t;  // (5 and 7 rows run on the 6- and 8-row instantiations, which stage 6 / 8 rows in LDS)
  return t < 1 ? 1 : (t > GEMV_MAX_T ? GEMV_MAX_T : t);
}

// One launch; a.T must be <= gemv_max_tokens(K).
hipError_t launch_gemv(const GemvArgs& a, hipStream_t s) {
  if (g_gemv_max_blocks == 0) {
    const char* e = getenv("MI_GEMV_MAX_BLOCKS");
    g_gemv_max_blocks = e ? atoi(e) : 2 * device_cus();  // 2 blocks per CU measured best on MI355X
    if (g_gemv_max_blocks <= 0) g_gemv_max_blocks = 2 * device_cus();
  }
  const bool pair_mode = !(a.mode == GEMV_SWIGLU || a.mode == GEMV_MOE_W13);
  // single-row units when row pairs would leave CUs without a full set of waves (256 CUs x 4 blocks x 4 waves)
  static int single_below = -1;  // MI_GEMV_SINGLE_BELOW: row-pair count under which units become single rows
  if (single_below < 0) {
    const char* e = getenv("MI_GEMV_SINGLE_BELOW");
    single_below = e ? atoi(e) : 0;  // measured: row pairs are at least as fast on MI355X
  }
  const bool single = pair_mode && a.mode != GEMV_QKV_ROPE && a.mode != GEMV_MOE_W2 && (a.N + 1) / 2 < single_below;
  const int units = pair_mode ? (single ? a.N : (a.N + 1) / 2) : a.N;
  // Persistent-style grid: every wave gets the same number k of units (no partially filled last round of blocks; the
  // two-batch load pipeline runs across a wave's units).  Preferred: the smallest k for which the block count is a
  // multiple of the 256 CUs (even load per CU), at most 4 per CU, and divides the units exactly - e.g. W1|W3: 14336
  // units -> 512 blocks x 4 waves x 7 units; q|k|v: 3072 units -> 768 blocks x 1.  Otherwise the smallest k that fits
  // g_gemv_max_blocks (512 = 2 blocks per CU measured best on MI355X).
  const int cus = device_cus();
  int blocks = 0;
  for (int k = 1; k <= 64 && !blocks; ++k) {
    const int b = (units + 4 * k - 1) / (4 * k);
    if (b <= 4 * cus && b % cus == 0 && b * 4 * k == units && (b <= g_gemv_max_blocks || k == 1)) blocks = b;
  }
  int nw = 4;  // waves per block
  if (!blocks) {
    // No even split over 4-wave blocks (Mistral-Nemo's Wo / W2: 5120 rows = 2560 pairs = 256 CUs x 10).  The fallback below
    // gives 320 blocks - 64 CUs with two blocks, 192 with one - and the doubly loaded CUs set the kernel's time at their own
    // ingest rate (W2: 30 us for 147 MB).  One token, plain modes: blocks of 5 .. 7 waves that DO split evenly, all resident.
    static int nw_ok = -1;  // MI_GEMV_NW=0: 4-wave blocks only (A/B testing)
    if (nw_ok < 0) {
      const char* e = getenv("MI_GEMV_NW");
      nw_ok = e ? atoi(e) : 1;
    }
    const bool plain1 = nw_ok && a.T == 1 && (a.mode == GEMV_STORE || a.mode == GEMV_RESIDUAL) && !single && a.norm_w == nullptr &&
                        (size_t)a.K * 2 <= 48 * 1024;
    for (int w = 5; plain1 && w <= 7 && nw == 4; ++w)
      for (int k = 1; k <= 16 && nw == 4; ++k) {
        const int b = units / (w * k);
        if (b >= cus && b % cus == 0 && b * w * k == units && b * w <= 16 * cus) {
          nw = w;
          blocks = b;
        }
      }
  }
  if (!blocks) {
    const int k = (units + 4 * g_gemv_max_blocks - 1) / (4 * g_gemv_max_blocks);
    blocks = (units + 4 * k - 1) / (4 * k);
  }
  if (blocks < 1) blocks = 1;
  if (a.mode == GEMV_MOE_W2) {
    const size_t lds = (size_t)a.top_k * a.K * 2;
    dim3 grid(blocks, a.T);
    switch (a.top_k) {
      case 1: hipLaunchKernelGGL((moe_w2_kernel<1>), grid, dim3(256), lds, s, a); break;
      case 2: hipLaunchKernelGGL((moe_w2_kernel<2>), grid, dim3(256), lds, s, a); break;
      case 4: hipLaunchKernelGGL((moe_w2_kernel<4>), grid, dim3(256), lds, s, a); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  int TT = a.T;
  if (a.mode == GEMV_MOE_W13) TT = 1;
  if (TT == 5) TT = 6;
  if (TT == 7) TT = 8;
  // [TT][K] activations, 4 x TT partial sums, and (norm modes at TT > 1: the DMA staging path of gemv_core.cuh) K norm weights
  const size_t lds = (size_t)TT * a.K * 2 + 4 * TT * sizeof(float) + ((TT > 1 && a.norm_w) ? (size_t)a.K * 2 : 0);
  if (lds > 80 * 1024 && blocks > cus) {  // one such block fits a CU: one block per CU, every wave the same number of units
    const int k = (units + 4 * cus - 1) / (4 * cus);
    blocks = (units + 4 * k - 1) / (4 * k);
  }
  dim3 grid(blocks, a.mode == GEMV_MOE_W13 ? a.T * a.top_k : 1);
  if (nw != 4) return a.mode == GEMV_STORE ? launch_nw<GEMV_STORE>(a, nw, grid, lds, s) : launch_nw<GEMV_RESIDUAL>(a, nw, grid, lds, s);
  switch (a.mode) {
    case GEMV_STORE:
      return single ? launch_mode<GEMV_STORE, 1>(a, TT, grid, lds, s) : launch_mode<GEMV_STORE, 2>(a, TT, grid, lds, s);
    case GEMV_RESIDUAL:
      return single ? launch_mode<GEMV_RESIDUAL, 1>(a, TT, grid, lds, s) : launch_mode<GEMV_RESIDUAL, 2>(a, TT, grid, lds, s);
    case GEMV_LOGITS:
      return single ? launch_mode<GEMV_LOGITS, 1>(a, TT, grid, lds, s) : launch_mode<GEMV_LOGITS, 2>(a, TT, grid, lds, s);
    case GEMV_SWIGLU: return launch_mode<GEMV_SWIGLU, 2>(a, TT, grid, lds, s);
    case GEMV_QKV_ROPE: return launch_mode<GEMV_QKV_ROPE, 2>(a, TT, grid, lds, s);
    case GEMV_MOE_W13: return launch_mode<GEMV_MOE_W13, 2>(a, 1, grid, lds, s);
    default: return hipErrorInvalidValue;
  }
}
