// Generic-storage layer kernels: the same operator sequence as the bf16 hot path for models whose storage dtype is fp32 or
// fp16 (reference transformer.py:303,338: `from_folder(dtype=...)` keeps whatever dtype the caller asks for, and the
// reference's own tests build fp32 models, tests/test_generate.py:51,100), and for bf16 models of a shape the tuned gfx950
// kernels decline (head_dim != 128, more than 16 experts, top_k = 3: csrc/api.hip check_model).
//
// Numerics contract = the reference's execution in the storage dtype T (the test oracle
// restates it): every nn.Linear accumulates in fp32 and rounds its output to T; RMSNorm normalises in fp32, rounds to T,
// multiplies by the weight and rounds again (transformer_layers.py:115-120); RoPE is an fp32 complex multiply rounded to T
// (rope.py:13-23); attention is fp32 softmax(q k^T / sqrt(Dh) + mask) v rounded to T; SiLU, the gate product, both
// residual adds and the MoE accumulation (moe.py:24-32) each round to T.  With T = fp32 nothing is rounded and the logits
// agree with the reference's stored fp32 outputs to ~1e-5 (tests/test_gpu_generic.py).
//
// These kernels are written for correctness and reasonable streaming behaviour, not for the roofline: BASELINE.json's
// configurations are all bf16 and run on the tuned kernels (gemv_core.cuh, gemm256.hip, decode_engine.hip, ...).
//   * linear, more than 8 rows: 128 x 128 tiles on the matrix cores (v_mfma_f32_32x32x16 f16 / bf16, v_mfma_f32_32x32x2 f32),
//     operands staged through LDS in the storage type; rows that are not 16-byte aligned: a 64 x 64 fp32-FMA tile kernel;
//     fp16 with at least 256 rows: the tuned 256 x 256 8-phase kernel itself, compiled for fp16 payloads (gemm256.hip);
//   * linear, up to 8 rows (decode): one wave per two weight rows, templated on the row count, eight 16-byte weight loads per
//     lane issued before anything waits, lanes stride K, wave reduction; fused forms: RMSNorm prologue, q | k | v from three
//     matrices, gate | up | SiLU | product;
//   * attention: one block per (query token, head); 4 or 16 waves split the visible keys, 64 lanes span the head dimension
//     (one coalesced load per K / V row, score by wave reduction), per-wave online softmax merged at the end; launches with
//     few (token, head) pairs also split the keys over blocks and merge in a second launch.  Keys older than this forward
//     come from the ring at slot position % W, newer ones from the post-RoPE activation rows (same visibility rule as
//     attn_prefill.hip), so the one kernel serves first prefills, later chunks, decode steps and the cache=None call; fp16
//     prefills with 128-wide heads: the MFMA flash kernel itself, compiled for fp16 payloads (attn_prefill.hip).
#include <cstdlib>
#include <cstring>

#include "common.cuh"
#include "kernels.h"

namespace {

struct sbf16 {
  uint16_t v;
};

template <typename T>
struct St;
template <>
struct St<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
  static __device__ __forceinline__ void ld8(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i] = a[i];
      o[4 + i] = b[i];
    }
  }
};
template <>
struct St<_Float16> {
  static __device__ __forceinline__ float ld(const _Float16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(_Float16* p, float v) { *p = (_Float16)v; }
  static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
  static __device__ __forceinline__ void ld8(const _Float16* p, float (&o)[8]) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 a = *reinterpret_cast<const h8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)a[i];
  }
};
template <>
struct St<sbf16> {
  static __device__ __forceinline__ float ld(const sbf16* p) { return bf_to_f(p->v); }
  static __device__ __forceinline__ void st(sbf16* p, float v) { p->v = f_to_bf(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf_round(v); }
  static __device__ __forceinline__ void ld8(const sbf16* p, float (&o)[8]) {
    const u32x4 a = ld16(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = bf_lo(a[i]);
      o[2 * i + 1] = bf_hi(a[i]);
    }
  }
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------------------------ embedding
// transformer.py:193.  Out-of-range ids: flagged in *bad_id (1 + token index) and clamped, as embedding_kernel does.
template <typename T>
__global__ __launch_bounds__(256) void g_embedding_kernel(T* out, const T* table, const int64_t* ids, int D, int vocab,
                                                          uint32_t* bad_id) {
  const int t = blockIdx.x;
  long id = ids[t];
  if (id < 0 || id >= vocab) {
    if (bad_id && threadIdx.x == 0) atomicMax(bad_id, (uint32_t)t + 1u);
    id = id < 0 ? 0 : vocab - 1;
  }
  const T* src = table + (size_t)id * D;
  T* dst = out + (size_t)t * D;
  for (int i = threadIdx.x; i < D; i += 256) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------ RMSNorm
// transformer_layers.py:115-120: one block per row.
template <typename T>
__global__ __launch_bounds__(256) void g_rmsnorm_kernel(T* out, const T* x, const T* w, int D, float eps) {
  __shared__ float part[4];
  const int t = blockIdx.x;
  const T* xr = x + (size_t)t * D;
  float ss = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float a = St<T>::ld(xr + i);
    ss = fmaf(a, a, ss);
  }
  ss = wave_sum_f(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = (part[0] + part[1]) + (part[2] + part[3]);
  const float inv = 1.0f / sqrtf(tot / (float)D + eps);
  T* o = out + (size_t)t * D;
  for (int i = threadIdx.x; i < D; i += 256)
    St<T>::st(o + i, St<T>::rnd(St<T>::ld(xr + i) * inv) * St<T>::ld(w + i));
}

// ------------------------------------------------------------------------------------------------ linear
// out[m, n] = epilogue(sum_k x[m, k] * w[n, k]): STORE rounds to T, RESIDUAL adds the (T) residual to the rounded product
// and rounds again (`h + wo(...)`), LOGITS writes fp32 of the T-rounded value (`output(...).float()`, transformer.py:235-242).
// `active` (MoE): tiles / launches none of whose rows is active do nothing; rows of a computed tile are all written.
template <typename T, int EPI>
__device__ __forceinline__ void g_store(const GLinearArgs& g, int m, int n, float acc) {
  const float v = St<T>::rnd(acc);
  if (EPI == G_EPI_LOGITS) {
    reinterpret_cast<float*>(g.out)[(size_t)m * g.ldo + n] = v;
  } else if (EPI == G_EPI_RESIDUAL) {
    const float r = St<T>::ld(reinterpret_cast<const T*>(g.residual) + (size_t)m * g.ldr + n);
    St<T>::st(reinterpret_cast<T*>(g.out) + (size_t)m * g.ldo + n, r + v);
  } else {
    St<T>::st(reinterpret_cast<T*>(g.out) + (size_t)m * g.ldo + n, v);
  }
}

template <typename T, int EPI>
__global__ __launch_bounds__(256) void g_gemm_kernel(GLinearArgs g) {
  constexpr int BM = 64, BN = 64, BK = 16, PAD = 4;
  __shared__ float As[BK][BM + PAD];
  __shared__ float Ws[BK][BN + PAD];
  __shared__ int any_active;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (g.active) {
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < BM && m0 + tid < g.M && g.active[m0 + tid]) any_active = 1;
    __syncthreads();
    if (!any_active) return;
  }
  const T* x = reinterpret_cast<const T*>(g.x);
  const T* w = reinterpret_cast<const T*>(g.w);
  // staging: thread -> (row tid >> 2, k offset (tid & 3) * 4) of both tiles
  const int sr = tid >> 2, sk = (tid & 3) * 4;
  const int am = min(m0 + sr, g.M - 1), wn = min(n0 + sr, g.N - 1);
  const T* ap = x + (size_t)am * g.ldx + sk;
  const T* wp = w + (size_t)wn * g.K + sk;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in = k0 + sk + i < g.K;
      As[sk + i][sr] = in ? St<T>::ld(ap + k0 + i) : 0.f;
      Ws[sk + i][sr] = in ? St<T>::ld(wp + k0 + i) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&As[kk][ty * 4]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&Ws[kk][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n < g.N) g_store<T, EPI>(g, m, n, acc[i][j]);
    }
  }
}

// ---- the same contraction on the matrix cores, for more than 8 rows with 16-byte-aligned rows (K % 8 == 0): 128 x 128
// output tile per block, four waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32), K in slabs of 32 staged through LDS in the
// STORAGE type (rows padded to 80 / 132 bytes: conflict-free 16-byte / 4-byte fragment reads); the next slab's global
// loads are issued before the MFMAs of this one and written to LDS after them (single buffer, two barriers per slab).
// fp16 / bf16: v_mfma_f32_32x32x16 (products of 16-bit values are exact in fp32, fp32 accumulate: what rocBLAS / torch do
// for a half GEMM); fp32: v_mfma_f32_32x32x2_f32 (full fp32 products).  C[row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col =
// lane & 31] per 16-register accumulator, A-operand rows = tokens, B-operand rows = output features.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename T>
struct Mf;
template <>
struct Mf<_Float16> {
  static constexpr int KPI = 16;
  typedef f16x8 frag;
  static __device__ __forceinline__ frag ldf(const char* p) { return __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p)); }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <>
struct Mf<sbf16> {
  static constexpr int KPI = 16;
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag ldf(const char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p)); }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct Mf<float> {
  static constexpr int KPI = 2;
  typedef float frag;
  static __device__ __forceinline__ frag ldf(const char* p) { return *reinterpret_cast<const float*>(p); }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};

template <typename T, int EPI>
__global__ __launch_bounds__(256) void g_gemm_mfma_kernel(GLinearArgs g) {
  constexpr int BM = 128, BN = 128, BK = 32, ES = (int)sizeof(T);
  constexpr int EPP = 16 / ES;                   // elements per 16-byte piece
  constexpr int PPR = BK / EPP;                  // pieces per tile row: 4 (16-bit) / 8 (fp32)
  constexpr int PPT = BM * PPR / 256;            // pieces per thread and tile: 2 / 4
  constexpr int ROWB = BK * ES + (ES == 2 ? 16 : 4);
  constexpr int KPL = Mf<T>::KPI / 2;            // k elements a lane supplies per MFMA: 8 / 1
  __shared__ __attribute__((aligned(16))) char As[BM * ROWB];
  __shared__ __attribute__((aligned(16))) char Ws[BN * ROWB];
  __shared__ int any_active;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (g.active) {
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < BM && m0 + tid < g.M && g.active[m0 + tid]) any_active = 1;
    __syncthreads();
    if (!any_active) return;
  }
  const T* x = reinterpret_cast<const T*>(g.x);
  const T* w = reinterpret_cast<const T*>(g.w);
  const T* asrc[PPT];
  const T* wsrc[PPT];
  int koff[PPT], loff[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = tid + 256 * j, row = p / PPR, pc = p % PPR;
    koff[j] = pc * EPP;
    loff[j] = row * ROWB + pc * 16;
    asrc[j] = x + (size_t)min(m0 + row, g.M - 1) * g.ldx + koff[j];
    wsrc[j] = w + (size_t)min(n0 + row, g.N - 1) * g.K + koff[j];
  }
  u32x4 ra[PPT], rw[PPT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const bool in = k0 + koff[j] < g.K;  // (K % 8 == 0: a piece is inside or outside as a whole)
      const int kc = in ? k0 : -koff[j];   // outside: the row's first piece (always there), zeroed below
      ra[j] = ld16(asrc[j] + kc);
      rw[j] = ld16(wsrc[j] + kc);
      if (!in) {
        ra[j] = u32x4{0u, 0u, 0u, 0u};
        rw[j] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      if (ES == 2) {
        st16(As + loff[j], ra[j]);
        st16(Ws + loff[j], rw[j]);
      } else {  // rows are 132 bytes apart: 4-byte stores
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          *reinterpret_cast<uint32_t*>(As + loff[j] + 4 * c) = ra[j][c];
          *reinterpret_cast<uint32_t*>(Ws + loff[j] + 4 * c) = rw[j][c];
        }
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const char* a_base = As + (wr * 64 + (lane & 31)) * ROWB + (lane >> 5) * KPL * ES;
  const char* w_base = Ws + (wc * 64 + (lane & 31)) * ROWB + (lane >> 5) * KPL * ES;
  gload(0);
  lstore();
  __syncthreads();
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    const bool more = k0 + BK < g.K;
    if (more) gload(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK / Mf<T>::KPI; ++ks) {
      typename Mf<T>::frag af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = Mf<T>::ldf(a_base + i * 32 * ROWB + ks * Mf<T>::KPI * ES);
        bf[i] = Mf<T>::ldf(w_base + i * 32 * ROWB + ks * Mf<T>::KPI * ES);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mf<T>::mma(af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
    if (more) {
      lstore();
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + (lane & 31);
        if (n < g.N) g_store<T, EPI>(g, m, n, acc[i][j][r]);
      }
    }
}

// M <= 8 rows, K a multiple of 8: one wave per TWO weight rows - two adjacent output columns, or (SwiGLU) the gate and the up
// row of ONE output column -, rows walked in chunks of 4 pieces per lane (eight independent 16-byte weight loads per lane
// in flight before anything waits; the activation rows come out of L1 / L2).  Optional prologue: RMSNorm of the rows (transformer_layers.py:115-120,
// every wave recomputes the 1 / rms of its M rows - 8 KB of L1-resident reads against 16+ KB of streamed weights), which
// removes a launch and a round trip of the normalised row per contraction.  Output columns may come from up to three weight
// matrices (q | k | v in one launch).
template <typename T>
__device__ __forceinline__ const T* g_seg_row(const GLinearArgs& g, int n) {
  if (!g.w1 || n < g.n0) return reinterpret_cast<const T*>(g.w) + (size_t)n * g.K;
  if (!g.w2 || n < g.n1) return reinterpret_cast<const T*>(g.w1) + (size_t)(n - g.n0) * g.K;
  return reinterpret_cast<const T*>(g.w2) + (size_t)(n - g.n1) * g.K;
}

// 16-byte pieces: EPP elements of T (8 of a 16-bit type, 4 floats)
template <typename T>
__device__ __forceinline__ void g_cvt_piece(const u32x4& raw, float (&o)[16 / sizeof(T)]) {
  constexpr int EPP = 16 / (int)sizeof(T);
  T tmp[EPP];
  __builtin_memcpy(tmp, &raw, 16);
#pragma unroll
  for (int i = 0; i < EPP; ++i) o[i] = St<T>::ld(&tmp[i]);
}

template <typename T, int EPI, int MR>  // MR = rows compiled in (1, 2, 4, 8 >= M; the surplus rows repeat row M - 1 and are not stored)
__global__ __launch_bounds__(256) void g_gemv_kernel(GLinearArgs g) {
  constexpr int NC = 2;
  constexpr bool SW = EPI == G_EPI_SWIGLU;
  constexpr int EPP = 16 / (int)sizeof(T);  // elements per 16-byte piece
  constexpr int UNR = MR <= 2 ? 4 : 2;      // pieces per lane and row in flight: 2 x UNR x 16 B of weights per lane
  const int lane = threadIdx.x & 63;
  const int wv_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n0 = SW ? wv_id : wv_id * NC;
  if (n0 >= g.N) return;
  if (g.active) {
    bool any = false;
    for (int m = 0; m < g.M; ++m) any = any || g.active[m] != 0;
    if (!any) return;
  }
  const T* x = reinterpret_cast<const T*>(g.x);
  const T* nw = reinterpret_cast<const T*>(g.norm_w);
  const T* wr[NC];
  if (SW) {
    wr[0] = reinterpret_cast<const T*>(g.w) + (size_t)n0 * g.K;
    wr[1] = reinterpret_cast<const T*>(g.w1) + (size_t)n0 * g.K;
  } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) wr[c] = g_seg_row<T>(g, min(n0 + c, g.N - 1));
  }
  const int PR = g.K / EPP;  // pieces per row (K % 8 == 0)
  // ---- 1 / rms of every row (fused RMSNorm): UNR independent loads per lane and round trip
  float inv[MR];
  const T* xrow[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    inv[m] = 1.f;
    xrow[m] = x + (size_t)min(m, g.M - 1) * g.ldx;
  }
  if (nw) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      {
        float ss = 0.f;
        for (int p0 = lane; p0 < PR; p0 += 64 * UNR) {
          u32x4 raw[UNR];
#pragma unroll
          for (int u = 0; u < UNR; ++u) raw[u] = ld16(xrow[m] + (size_t)min(p0 + 64 * u, PR - 1) * EPP);
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            float xv[EPP];
            g_cvt_piece<T>(raw[u], xv);
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < EPP; ++i) s1 = fmaf(xv[i], xv[i], s1);
            ss += (p0 + 64 * u < PR) ? s1 : 0.f;
          }
        }
        ss = wave_sum_f(ss);
        inv[m] = 1.0f / sqrtf(ss / (float)g.K + g.eps);
      }
    }
  }
  float acc[NC][MR];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[c][m] = 0.f;
  // ---- the contraction: all 2 x UNR weight pieces of a chunk are issued before anything waits (clamped addresses, the
  // pieces past the row's end contribute zero), then the activation pieces row by row out of L1 / L2
  for (int p0 = lane; p0 < PR; p0 += 64 * UNR) {
    u32x4 wraw[NC][UNR], nraw[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const size_t off = (size_t)min(p0 + 64 * u, PR - 1) * EPP;
#pragma unroll
      for (int c = 0; c < NC; ++c) wraw[c][u] = ld16_nt(wr[c] + off);
      if (nw) nraw[u] = ld16(nw + off);
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      {
        u32x4 xraw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) xraw[u] = ld16(xrow[m] + (size_t)min(p0 + 64 * u, PR - 1) * EPP);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          float xv[EPP];
          g_cvt_piece<T>(xraw[u], xv);
          if (nw) {
            float nv[EPP];
            g_cvt_piece<T>(nraw[u], nv);
#pragma unroll
            for (int i = 0; i < EPP; ++i) xv[i] = St<T>::rnd(St<T>::rnd(xv[i] * inv[m]) * nv[i]);
          }
          const bool in = p0 + 64 * u < PR;
#pragma unroll
          for (int i = 0; i < EPP; ++i) xv[i] = in ? xv[i] : 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            float wv[EPP];
            g_cvt_piece<T>(wraw[c][u], wv);
#pragma unroll
            for (int i = 0; i < EPP; ++i) acc[c][m] = fmaf(xv[i], wv[i], acc[c][m]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (m < g.M) {
      const float v0 = wave_sum_f(acc[0][m]), v1 = wave_sum_f(acc[1][m]);
      if (lane == 0) {
        if (SW) {  // transformer_layers.py:106: silu(w1 x) * w3 x, every intermediate rounded to T
          const float a = St<T>::rnd(v0), b = St<T>::rnd(v1);
          const float sl = St<T>::rnd(a / (1.0f + expf(-a)));
          St<T>::st(reinterpret_cast<T*>(g.out) + (size_t)m * g.ldo + n0, sl * b);
        } else {
          g_store<T, EPI>(g, m, n0, v0);
          if (n0 + 1 < g.N) g_store<T, EPI>(g, m, n0 + 1, v1);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ RoPE
// rope.py:13-23, in place on the first n_rot_cols columns (q | k) of the fused buffer; one thread per pair.
template <typename T>
__global__ __launch_bounds__(256) void g_rope_kernel(T* qkv, int ld, int T_rows, int n_rot_cols, int Dh, const float* rope_cs,
                                                     const int32_t* tok_pos) {
  const int pairs = n_rot_cols >> 1;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)T_rows * pairs) return;
  const int t = (int)(gid / pairs), p = (int)(gid % pairs);
  const int i = p % (Dh >> 1);
  T* ptr = qkv + (size_t)t * ld + 2 * p;
  const float* cs = rope_cs + ((size_t)tok_pos[t] * (Dh >> 1) + i) * 2;
  float re, im;
  rope_pair(St<T>::ld(ptr), St<T>::ld(ptr + 1), cs[0], cs[1], re, im);
  St<T>::st(ptr, re);
  St<T>::st(ptr + 1, im);
}

// ------------------------------------------------------------------------------------------------ ring write
// cache.py:83-92 + 226-235 (kv_write_kernel for any element type): only the last W tokens of a chunk are stored.
template <typename T>
__global__ __launch_bounds__(256) void g_kv_write_kernel(T* ck, T* cv, int W, const T* k, const T* v, int ld, int T_rows, int kv_dim,
                                                         const int32_t* tok_seq, const int32_t* tok_pos, const int32_t* q_start,
                                                         int layout, int Dh) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)T_rows * kv_dim) return;
  const int t = (int)(gid / kv_dim), c = (int)(gid % kv_dim);
  const int b = tok_seq[t];
  const int i = t - q_start[b];
  const int s = q_start[b + 1] - q_start[b];
  if (i < s - W) return;
  const size_t off = kv_offset(layout, W, kv_dim, Dh, (size_t)b, tok_pos[t] % W, c);
  ck[off] = k[(size_t)t * ld + c];
  cv[off] = v[(size_t)t * ld + c];
}

// ------------------------------------------------------------------------------------------------ attention
// transformer_layers.py:74-89 for every branch.  Block = (query token, q head), NW waves; wave w takes the visible keys in
// groups of four - group w, w + NW, ... - the 64 lanes span the head dimension (EPL = head_dim / 64 adjacent elements per
// lane: a K / V row is ONE coalesced load instruction; head dims that are not a multiple of 64: lane, lane + 64, ...
// element by element), the score is a wave reduction, each wave keeps its own online-softmax state and the NW states are
// merged through LDS at the end.  Two register sets of four K and four V rows are refilled in place (the next group's
// loads are in flight while this one is reduced).  NW = 4 when the launch has many (token, head) blocks, 16 for the few
// blocks of a decode step.
template <int BYTES>
struct RawN;
template <>
struct RawN<2> {
  typedef uint16_t type;
};
template <>
struct RawN<4> {
  typedef uint32_t type;
};
template <>
struct RawN<8> {
  typedef u32x2 type;
};
template <>
struct RawN<16> {
  typedef u32x4 type;
};

// EPL > 0: lane's EPL adjacent elements of a row in one load; EPL == 0: elements lane + 64 i, i < 4, below Dh
template <typename T, int EPL>
__device__ __forceinline__ void g_row_load(const T* row, int lane, int Dh, float (&o)[EPL > 0 ? EPL : 4]) {
  if constexpr (EPL > 0) {
    typedef typename RawN<EPL * (int)sizeof(T)>::type raw_t;
    const raw_t r = *reinterpret_cast<const raw_t*>(row + lane * EPL);
    T tmp[EPL];
    __builtin_memcpy(tmp, &r, sizeof(r));
#pragma unroll
    for (int i = 0; i < EPL; ++i) o[i] = St<T>::ld(&tmp[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = lane + 64 * i;
      o[i] = d < Dh ? St<T>::ld(row + d) : 0.f;
    }
  }
}

template <typename T, int EPL, int NW, bool SPLIT>
__global__ __launch_bounds__(NW * 64) void g_attention_kernel(GAttnArgs a) {
  constexpr int NI = EPL > 0 ? EPL : 4;  // elements per lane
  constexpr int U = 4;                   // keys per group
  __shared__ float sm_m[NW], sm_l[NW];
  __shared__ float sm_acc[NW][256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = blockIdx.x, h = blockIdx.y;
  const int Dh = a.Dh, kvh = h / (a.H / a.Hkv);
  const int nq = a.H * Dh, kv_dim = a.Hkv * Dh;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  int row0, p_b, qp, b, W;
  if (a.causal) {
    b = a.tok_seq[t];
    row0 = a.q_start[b];
    p_b = a.kv_before[b];
    qp = a.tok_pos[t];
    W = a.W;
  } else {
    b = 0;
    row0 = 0;
    p_b = 0;
    qp = a.T - 1;
    W = a.T;
  }
  const int n_old = min(p_b, W);
  const int kp_lo = a.causal ? max(p_b - n_old, qp - W + 1) : 0;
  const int kp_hi = qp;
  float q[NI], acc[NI];
  g_row_load<T, EPL>(qkv + (size_t)t * a.ld + (size_t)h * Dh, lane, Dh, q);
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  const size_t ring0 = kv_offset(a.kv_layout, W, kv_dim, Dh, (size_t)b, 0, kvh * Dh);
  const size_t ring_stride = a.kv_layout ? (size_t)Dh : (size_t)kv_dim;  // elements between consecutive ring slots of this kv head
  const T* ring_k = a.cache_k ? reinterpret_cast<const T*>(a.cache_k) + ring0 : qkv;
  const T* ring_v = a.cache_v ? reinterpret_cast<const T*>(a.cache_v) + ring0 : qkv;
  const T* act_k = qkv + nq + (size_t)kvh * Dh;
  const T* act_v = act_k + kv_dim;
  float m_run = -INFINITY, l_run = 0.f;
  const int n_groups = (kp_hi - kp_lo + U) / U;  // groups of U keys in the visible range
  // ... of which this wave takes every GS-th: GS = waves of the block x key splits of the launch (SPLIT: gridDim.z blocks
  // per (token, head), each leaving an un-normalised partial for g_attention_combine_kernel)
  const int gw = (SPLIT ? (int)blockIdx.z * NW : 0) + wid, GS = (SPLIT ? (int)gridDim.z : 1) * NW;
  const int n_mine = gw < n_groups ? (n_groups - gw + GS - 1) / GS : 0;
  struct Set {
    float k[U][NI], v[U][NI];
  };
  Set A, B;
  auto load = [&](Set& s, int it) {  // always 2 U row loads from clamped (valid) positions; masked in reduce
    const int base = kp_lo + (gw + GS * it) * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kp = min(base + u, kp_hi);
      const T *kr, *vr;
      if (kp < p_b) {  // wave-uniform
        const size_t off = (size_t)(kp % W) * ring_stride;
        kr = ring_k + off;
        vr = ring_v + off;
      } else {
        const size_t off = (size_t)(row0 + kp - p_b) * a.ld;
        kr = act_k + off;
        vr = act_v + off;
      }
      g_row_load<T, EPL>(kr, lane, Dh, s.k[u]);
      g_row_load<T, EPL>(vr, lane, Dh, s.v[u]);
    }
  };
  auto reduce = [&](const Set& s, int it) {
    if (it >= n_mine) return;  // wave-uniform; (keeps an all-masked group away from m_run = -inf: exp(-inf + inf))
    const int base = kp_lo + (gw + GS * it) * U;
    float sc[U], mx = m_run;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) part = fmaf(q[i], s.k[u][i], part);
      const float sv = wave_sum_f(part) * a.scale;
      sc[u] = (base + u <= kp_hi) ? sv : -INFINITY;
      mx = fmaxf(mx, sc[u]);
    }
    const float alpha = expf(m_run - mx);  // (first group: exp(-inf) = 0 against zero accumulators)
    m_run = mx;
    l_run *= alpha;
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] *= alpha;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float p = expf(sc[u] - mx);  // masked key: exp(-inf) = 0
      l_run += p;
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = fmaf(p, s.v[u][i], acc[i]);
    }
  };
  load(A, 0);
  load(B, 1);
  for (int it = 0; it < n_mine; it += 2) {
    reduce(A, it);
    load(A, it + 2);
    reduce(B, it + 1);
    load(B, it + 3);
  }
  if (lane == 0) {
    sm_m[wid] = m_run;
    sm_l[wid] = l_run;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) sm_acc[wid][EPL > 0 ? lane * EPL + i : lane + 64 * i] = acc[i];
  __syncthreads();
  if (tid < Dh) {
    float M = sm_m[0];  // (unsplit: finite - wave 0 always owns the group that starts at kp_lo)
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, sm_m[w]);
    float L = 0.f, o = 0.f;
    if (M > -INFINITY) {  // (a split block whose waves found no keys: exp(-inf + inf) must not be formed)
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float f = expf(sm_m[w] - M);  // a wave without keys: exp(-inf) = 0
        L = fmaf(sm_l[w], f, L);
        o = fmaf(sm_acc[w][tid], f, o);
      }
    }
    if (SPLIT) {  // partial of this key split: [Dh] un-normalised outputs, then (max, sum)
      float* pp = a.partial + (((size_t)t * a.H + h) * gridDim.z + blockIdx.z) * (Dh + 2);
      pp[tid] = o;
      if (tid == 0) {
        pp[Dh] = M;
        pp[Dh + 1] = L;
      }
    } else {
      St<T>::st(reinterpret_cast<T*>(a.out) + (size_t)t * a.ldo + (size_t)h * Dh + tid, o / L);
    }
  }
}

// merge of the key splits of one (token, head): the same max / rescale / sum as the in-block merge above
template <typename T>
__global__ __launch_bounds__(256) void g_attention_combine_kernel(GAttnArgs a, int S) {
  const int tid = threadIdx.x, t = blockIdx.x, h = blockIdx.y, Dh = a.Dh;
  if (tid >= Dh) return;
  const float* pp = a.partial + (((size_t)t * a.H + h) * S) * (Dh + 2);
  float M = -INFINITY;
  for (int s = 0; s < S; ++s) M = fmaxf(M, pp[(size_t)s * (Dh + 2) + Dh]);
  float L = 0.f, o = 0.f;
  for (int s = 0; s < S; ++s) {
    const float f = expf(pp[(size_t)s * (Dh + 2) + Dh] - M);  // (split 0 always holds the key kp_lo: M is finite)
    L = fmaf(pp[(size_t)s * (Dh + 2) + Dh + 1], f, L);
    o = fmaf(pp[(size_t)s * (Dh + 2) + tid], f, o);
  }
  St<T>::st(reinterpret_cast<T*>(a.out) + (size_t)t * a.ldo + (size_t)h * Dh + tid, o / L);
}

constexpr int G_ATTN_SPLIT_BELOW = 256;  // (token, head) pairs under which the keys are also split over blocks
constexpr int G_ATTN_MAX_SPLITS = 16;

template <typename T, int EPL>
void attention_nw(const GAttnArgs& a, hipStream_t s) {
  const long pairs = (long)a.T * a.H;
  if (pairs >= 1024) {
    hipLaunchKernelGGL((g_attention_kernel<T, EPL, 4, false>), dim3(a.T, a.H), dim3(256), 0, s, a);
  } else if (pairs >= G_ATTN_SPLIT_BELOW || !a.partial) {
    hipLaunchKernelGGL((g_attention_kernel<T, EPL, 16, false>), dim3(a.T, a.H), dim3(1024), 0, s, a);
  } else {
    // a decode step: 32 (token, head) pairs would leave 7/8 of the chip idle and every wave with a long chain of dependent
    // round trips - split the keys over enough blocks to cover the CUs, merge in a second tiny launch
    int S = (int)((2 * G_ATTN_SPLIT_BELOW + pairs - 1) / pairs);
    S = S > G_ATTN_MAX_SPLITS ? G_ATTN_MAX_SPLITS : S;
    hipLaunchKernelGGL((g_attention_kernel<T, EPL, 4, true>), dim3(a.T, a.H, S), dim3(256), 0, s, a);
    hipLaunchKernelGGL((g_attention_combine_kernel<T>), dim3(a.T, a.H), dim3(256), 0, s, a, S);
  }
}
template <typename T>
void attention_t(const GAttnArgs& a, hipStream_t s) {
  // one load per row needs rows that start on a multiple of the lane's load width: every row offset here is a multiple of
  // Dh elements, so Dh % 64 == 0 and 16-byte aligned bases are enough
  const bool al = ((reinterpret_cast<size_t>(a.qkv) | reinterpret_cast<size_t>(a.cache_k) | reinterpret_cast<size_t>(a.cache_v)) & 15) == 0 &&
                  a.ld % 8 == 0;
  if (al && a.Dh == 64) attention_nw<T, 1>(a, s);
  else if (al && a.Dh == 128) attention_nw<T, 2>(a, s);
  else if (al && a.Dh == 256) attention_nw<T, 4>(a, s);
  else attention_nw<T, 0>(a, s);
}

// ------------------------------------------------------------------------------------------------ SwiGLU
// transformer_layers.py:106: silu(w1 x) * w3 x with both intermediates rounded to T.  In place on `a`.
template <typename T>
__global__ __launch_bounds__(256) void g_swiglu_kernel(T* a, const T* b, int T_rows, int F, const int32_t* active) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)T_rows * F) return;
  if (active && !active[gid / F]) return;
  const float x = St<T>::ld(a + gid);
  const float sl = St<T>::rnd(x / (1.0f + expf(-x)));
  St<T>::st(a + gid, sl * St<T>::ld(b + gid));
}

// ------------------------------------------------------------------------------------------------ MoE
// moe.py:26-27: top-k of the (T-rounded) router logits, fp32 softmax over the picked ones, rounded to T.  One thread per token;
// ties go to the lower expert id.
template <typename T>
__global__ __launch_bounds__(64) void g_moe_topk_kernel(const T* logits, int T_rows, int E, int k, int32_t* sel_idx, float* sel_w) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= T_rows) return;
  const T* row = logits + (size_t)t * E;
  float mx = -INFINITY;
  for (int j = 0; j < k; ++j) {
    int best = -1;
    float bv = -INFINITY;
    for (int e = 0; e < E; ++e) {
      bool taken = false;
      for (int q = 0; q < j; ++q) taken = taken || sel_idx[(size_t)t * k + q] == e;
      const float v = St<T>::ld(row + e);
      if (!taken && (best < 0 || v > bv)) {
        best = e;
        bv = v;
      }
    }
    sel_idx[(size_t)t * k + j] = best;
    sel_w[(size_t)t * k + j] = bv;
    mx = fmaxf(mx, bv);
  }
  float sum = 0.f;
  for (int j = 0; j < k; ++j) {
    const float p = expf(sel_w[(size_t)t * k + j] - mx);
    sel_w[(size_t)t * k + j] = p;
    sum += p;
  }
  for (int j = 0; j < k; ++j) sel_w[(size_t)t * k + j] = St<T>::rnd(sel_w[(size_t)t * k + j] / sum);
}

// rows that picked expert e, and their weight
__global__ __launch_bounds__(256) void g_moe_mask_kernel(const int32_t* sel_idx, const float* sel_w, int T_rows, int k, int e,
                                                         int32_t* active, float* wt) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T_rows) return;
  int on = 0;
  float w = 0.f;
  for (int j = 0; j < k; ++j)
    if (sel_idx[(size_t)t * k + j] == e) {
      on = 1;
      w = sel_w[(size_t)t * k + j];
    }
  active[t] = on;
  wt[t] = w;
}

// moe.py:31: results[rows] += weight * expert(rows), product and sum each rounded to T
template <typename T>
__global__ __launch_bounds__(256) void g_moe_accum_kernel(T* results, const T* y, const int32_t* active, const float* wt, int T_rows,
                                                          int D) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)T_rows * D) return;
  const int t = (int)(gid / D);
  if (!active[t]) return;
  const float prod = St<T>::rnd(wt[t] * St<T>::ld(y + gid));
  St<T>::st(results + gid, St<T>::ld(results + gid) + prod);
}

// out = a + b, rounded to T (`h + r`, transformer_layers.py:166-168)
template <typename T>
__global__ __launch_bounds__(256) void g_add_kernel(T* out, const T* a, const T* b, size_t n) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid < n) St<T>::st(out + gid, St<T>::ld(a + gid) + St<T>::ld(b + gid));
}

// nn.GELU() (exact erf form), fp32 evaluation rounded once (vision_encoder.py:112-116), in place on rows of N elements
template <typename T>
__global__ __launch_bounds__(256) void g_gelu_kernel(T* x, int ldx, int N) {
  T* row = x + (size_t)blockIdx.x * ldx;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float v = St<T>::ld(row + i);
    St<T>::st(row + i, 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void g_zero_kernel(T* p, size_t n) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid < n) St<T>::st(p + gid, 0.f);
}

template <typename T, int EPI>
hipError_t linear_t(const GLinearArgs& g, hipStream_t s) {
  static int use_mfma = -1;  // MI_GENERIC_MFMA=0: every multi-row contraction on the fp32-FMA tile kernel (A/B, tests)
  if (use_mfma < 0) {
    const char* e = getenv("MI_GENERIC_MFMA");
    use_mfma = e ? atoi(e) : 1;
  }
  const bool aligned = g.K % 8 == 0 && g.ldx % 8 == 0 && (reinterpret_cast<size_t>(g.x) & 15) == 0 && (reinterpret_cast<size_t>(g.w) & 15) == 0;
  const bool special = g.w1 || g.w2 || g.norm_w || EPI == G_EPI_SWIGLU;  // forms of the M <= 8 kernel only
  if (g.M <= 8 && aligned) {
    const dim3 grid(EPI == G_EPI_SWIGLU ? (g.N + 3) / 4 : (g.N + 7) / 8);
    if (g.M == 1) hipLaunchKernelGGL((g_gemv_kernel<T, EPI, 1>), grid, dim3(256), 0, s, g);
    else if (g.M == 2) hipLaunchKernelGGL((g_gemv_kernel<T, EPI, 2>), grid, dim3(256), 0, s, g);
    else if (g.M <= 4) hipLaunchKernelGGL((g_gemv_kernel<T, EPI, 4>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((g_gemv_kernel<T, EPI, 8>), grid, dim3(256), 0, s, g);
  } else if (special) {
    return hipErrorInvalidValue;
  } else if constexpr (EPI == G_EPI_SWIGLU) {
    return hipErrorInvalidValue;
  } else if (aligned && use_mfma) {
    hipLaunchKernelGGL((g_gemm_mfma_kernel<T, EPI>), dim3((g.N + 127) / 128, (g.M + 127) / 128), dim3(256), 0, s, g);
  } else {
    hipLaunchKernelGGL((g_gemm_kernel<T, EPI>), dim3((g.N + 63) / 64, (g.M + 63) / 64), dim3(256), 0, s, g);
  }
  return hipGetLastError();
}
template <typename T>
hipError_t linear_e(const GLinearArgs& g, hipStream_t s) {
  switch (g.epi) {
    case G_EPI_STORE: return linear_t<T, G_EPI_STORE>(g, s);
    case G_EPI_RESIDUAL: return linear_t<T, G_EPI_RESIDUAL>(g, s);
    case G_EPI_LOGITS: return linear_t<T, G_EPI_LOGITS>(g, s);
    case G_EPI_SWIGLU: return linear_t<T, G_EPI_SWIGLU>(g, s);
    default: return hipErrorInvalidValue;
  }
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

#define G_DISPATCH(dt, CALL)                                   \
  switch (dt) {                                                \
    case G_DT_BF16: { using T = sbf16; CALL; break; }          \
    case G_DT_FP16: { using T = _Float16; CALL; break; }       \
    case G_DT_FP32: { using T = float; CALL; break; }          \
    default: return hipErrorInvalidValue;                      \
  }                                                            \
  return hipGetLastError();

size_t g_elem_bytes(int dt) { return dt == G_DT_FP32 ? 4 : 2; }
bool g_gemv_takes(int M, int K, int ldx) { return M <= 8 && K % 8 == 0 && ldx % 8 == 0; }
size_t g_attn_partial_floats(int T, int H, int Dh) {
  const long pairs = (long)T * H;
  return pairs < G_ATTN_SPLIT_BELOW ? (size_t)pairs * G_ATTN_MAX_SPLITS * (Dh + 2) : 0;
}

hipError_t launch_g_embedding(int dt, void* out, const void* table, const int64_t* ids, int T_rows, int D, int vocab, uint32_t* bad_id,
                              hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_embedding_kernel<T>), dim3(T_rows), dim3(256), 0, s, (T*)out, (const T*)table, ids, D, vocab, bad_id))
}

hipError_t launch_g_rmsnorm(int dt, void* out, const void* x, const void* w, int T_rows, int D, float eps, hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_rmsnorm_kernel<T>), dim3(T_rows), dim3(256), 0, s, (T*)out, (const T*)x, (const T*)w, D, eps))
}

// fp16 prefills of at least 256 rows: the tuned 256 x 256 8-phase MFMA kernel, compiled a second time for fp16 payloads
// (gemm256.hip, -DG256_F16=1: same DMA / LDS / barrier schedule, v_mfma_f32_16x16x32_f16, half rounding points).  It also
// takes the multi-matrix (q | k | v) and the fused SwiGLU forms.  MI_GENERIC_G256=0 keeps everything on the kernels above.
static bool g256_f16_args(const GLinearArgs& g, GemmArgs& a) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MI_GENERIC_G256");
    on = e ? atoi(e) : 1;
  }
  if (!on || g.active || g.norm_w || g.M < 256) return false;
  if (g.ldx % 8 || (reinterpret_cast<size_t>(g.x) & 15) || (reinterpret_cast<size_t>(g.w) & 15)) return false;
  memset(&a, 0, sizeof(a));
  a.M = g.M; a.N = g.N; a.K = g.K; a.a = reinterpret_cast<const bf16_t*>(g.x); a.lda = g.ldx;
  a.out = g.out; a.ldo = g.ldo;
  a.w0 = reinterpret_cast<const bf16_t*>(g.w);
  switch (g.epi) {
    case G_EPI_STORE: a.epi = GEMM_STORE; break;
    case G_EPI_RESIDUAL:
      if (g.ldr != g.ldo) return false;
      a.epi = GEMM_RESIDUAL; a.residual = reinterpret_cast<const bf16_t*>(g.residual); break;
    case G_EPI_LOGITS: a.epi = GEMM_LOGITS; break;
    case G_EPI_SWIGLU: a.epi = GEMM_SWIGLU; break;
    default: return false;
  }
  if (g.epi == G_EPI_SWIGLU) {
    if (!g.w1) return false;
    a.w1 = reinterpret_cast<const bf16_t*>(g.w1); a.n0 = a.n1 = g.N;
  } else if (g.w1) {
    a.w1 = reinterpret_cast<const bf16_t*>(g.w1); a.w2 = reinterpret_cast<const bf16_t*>(g.w2);
    a.n0 = g.n0; a.n1 = g.w2 ? g.n1 : g.N;
  } else {
    a.n0 = a.n1 = g.N;
  }
  return gemm256_applicable_f16(a);
}

bool g_linear_fused_ok(int dt, const GLinearArgs& g) {
  if (g_gemv_takes(g.M, g.K, g.ldx)) return true;
  GemmArgs a;
  return dt == G_DT_FP16 && g256_f16_args(g, a);
}

hipError_t launch_g_linear(int dt, const GLinearArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return hipErrorInvalidValue;
  if (dt == G_DT_FP16) {
    GemmArgs a;
    if (g256_f16_args(g, a)) return launch_gemm256_f16(a, s);
  }
  switch (dt) {
    case G_DT_BF16: return linear_e<sbf16>(g, s);
    case G_DT_FP16: return linear_e<_Float16>(g, s);
    case G_DT_FP32: return linear_e<float>(g, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_g_rope(int dt, void* qkv, int ld, int T_rows, int n_rot_cols, int Dh, const float* rope_cs, const int32_t* tok_pos,
                         hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_rope_kernel<T>), dim3(blocks_for((size_t)T_rows * (n_rot_cols >> 1))), dim3(256), 0, s, (T*)qkv, ld,
                                    T_rows, n_rot_cols, Dh, rope_cs, tok_pos))
}

hipError_t launch_g_kv_write(int dt, void* ck, void* cv, int W, const void* k, const void* v, int ld, int T_rows, int kv_dim,
                             const int32_t* tok_seq, const int32_t* tok_pos, const int32_t* q_start, int kv_layout, int Dh, hipStream_t s) {
  if (Dh <= 0 || kv_dim % Dh) return hipErrorInvalidValue;
  G_DISPATCH(dt, hipLaunchKernelGGL((g_kv_write_kernel<T>), dim3(blocks_for((size_t)T_rows * kv_dim)), dim3(256), 0, s, (T*)ck, (T*)cv, W,
                                    (const T*)k, (const T*)v, ld, T_rows, kv_dim, tok_seq, tok_pos, q_start, kv_layout, Dh))
}

hipError_t launch_g_attention(int dt, const GAttnArgs& a, hipStream_t s) {
  if (a.Dh > 256 || a.H % a.Hkv) return hipErrorInvalidValue;
  // fp16 prefills with 128-wide heads: the tuned MFMA flash kernel compiled for fp16 payloads (attn_prefill.hip, -DATTN_F16=1;
  // P is rounded to fp16 for P.V as fp16 flash kernels do).  MI_GENERIC_ATTN_MFMA=0 keeps the fp32-P kernel below.
  static int mfma_on = -1;
  if (mfma_on < 0) {
    const char* e = getenv("MI_GENERIC_ATTN_MFMA");
    mfma_on = e ? atoi(e) : 1;
  }
  if (mfma_on && dt == G_DT_FP16 && a.Dh == 128 && a.T >= 128 && a.B > 0 && a.max_q_len > 0 && a.ld % 8 == 0 &&
      ((reinterpret_cast<size_t>(a.qkv) | reinterpret_cast<size_t>(a.cache_k) | reinterpret_cast<size_t>(a.cache_v) |
        reinterpret_cast<size_t>(a.out)) & 15) == 0 &&
      (size_t)a.W * a.Hkv * a.Dh < (1ull << 31) && (size_t)a.T * (size_t)a.ld < (1ull << 31)) {
    AttnPrefillArgs p;
    p.out = a.out; p.qkv = reinterpret_cast<const bf16_t*>(a.qkv); p.ld = a.ld;
    p.cache_k = reinterpret_cast<const bf16_t*>(a.cache_k); p.cache_v = reinterpret_cast<const bf16_t*>(a.cache_v);
    p.kv_layout = a.kv_layout;
    p.W = a.W; p.B = a.B; p.max_q_len = a.max_q_len; p.H = a.H; p.Hkv = a.Hkv; p.Dh = a.Dh;
    p.q_start = a.q_start; p.kv_before = a.kv_before; p.causal = a.causal; p.scale = a.scale;
    return launch_attn_prefill_f16(p, s);
  }
  G_DISPATCH(dt, attention_t<T>(a, s))
}

hipError_t launch_g_swiglu(int dt, void* a, const void* b, int T_rows, int F, const int32_t* active, hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_swiglu_kernel<T>), dim3(blocks_for((size_t)T_rows * F)), dim3(256), 0, s, (T*)a, (const T*)b, T_rows, F,
                                    active))
}

hipError_t launch_g_moe_topk(int dt, const void* logits, int T_rows, int E, int k, int32_t* sel_idx, float* sel_w, hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_moe_topk_kernel<T>), dim3((T_rows + 63) / 64), dim3(64), 0, s, (const T*)logits, T_rows, E, k, sel_idx,
                                    sel_w))
}

hipError_t launch_g_moe_mask(const int32_t* sel_idx, const float* sel_w, int T_rows, int k, int e, int32_t* active, float* wt,
                             hipStream_t s) {
  hipLaunchKernelGGL(g_moe_mask_kernel, dim3((T_rows + 255) / 256), dim3(256), 0, s, sel_idx, sel_w, T_rows, k, e, active, wt);
  return hipGetLastError();
}

hipError_t launch_g_moe_accum(int dt, void* results, const void* y, const int32_t* active, const float* wt, int T_rows, int D,
                              hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_moe_accum_kernel<T>), dim3(blocks_for((size_t)T_rows * D)), dim3(256), 0, s, (T*)results, (const T*)y,
                                    active, wt, T_rows, D))
}

hipError_t launch_g_add(int dt, void* out, const void* a, const void* b, size_t n, hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_add_kernel<T>), dim3(blocks_for(n)), dim3(256), 0, s, (T*)out, (const T*)a, (const T*)b, n))
}

hipError_t launch_g_gelu(int dt, void* x, int ldx, int T_rows, int N, hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_gelu_kernel<T>), dim3(T_rows), dim3(256), 0, s, (T*)x, ldx, N))
}

hipError_t launch_g_zero(int dt, void* p, size_t n, hipStream_t s) {
  G_DISPATCH(dt, hipLaunchKernelGGL((g_zero_kernel<T>), dim3(blocks_for(n)), dim3(256), 0, s, (T*)p, n))
}
