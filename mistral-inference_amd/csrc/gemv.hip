// Weight-streaming GEMV kernels for M <= 8 tokens (decode): HBM-bound, one pass over the weights.
//
// Replaces, per decode token and layer, the reference's separate launches for RMSNorm
// (transformer_layers.py:115-120), the q/k/v/o and w1/w2/w3 nn.Linear GEMVs (:66,:93,:105-106), RoPE
// (rope.py:13-23), the ring write (cache.py:83-92), the residual adds (:166,:168) and silu*mul.
//
// Structure (cdna_hip_programming.md "GEMV / M<=16 decode weights"): weights go straight HBM -> VGPR
// with 16-byte non-temporal loads, two rows per wave in flight, U chunks deep, double buffered; the
// (optionally RMS-normalised) activation vector lives in LDS; the first weight batch is issued before
// the prologue so the x staging hides under HBM latency.  A wave owns "units" (a pair of weight rows)
// strided over the whole grid, so at any instant the chip streams one contiguous weight region.
#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int U = 4;  // 16-byte chunks in flight per row per buffer

template <int TT>
struct Acc {
  float v[2][TT];
};

struct RowPair {
  const bf16_t* a;
  const bf16_t* b;  // nullptr when the unit has one row
};

__device__ __forceinline__ void load_batch(const RowPair& r, int c0, int K, int lane, u32x4 (&ca)[U], u32x4 (&cb)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = ((c0 + u) * 64 + lane) * 8;
    const bool ok = e < K;
    u32x4 z = {0u, 0u, 0u, 0u};
    ca[u] = ok ? ld16_nt(r.a + e) : z;
    cb[u] = (ok && r.b) ? ld16_nt(r.b + e) : z;
  }
}

template <int TT>
__device__ __forceinline__ void fma_chunk(const u32x4& wa, const u32x4& wb, const bf16_t* xs, int K, int e,
                                          Acc<TT>& acc) {
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[2 * i] = bf_lo(wa[i]);
    a[2 * i + 1] = bf_hi(wa[i]);
    b[2 * i] = bf_lo(wb[i]);
    b[2 * i + 1] = bf_hi(wb[i]);
  }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + (size_t)t * K + e);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x0 = bf_lo(xv[i]), x1 = bf_hi(xv[i]);
      acc.v[0][t] = fmaf(a[2 * i], x0, acc.v[0][t]);
      acc.v[0][t] = fmaf(a[2 * i + 1], x1, acc.v[0][t]);
      acc.v[1][t] = fmaf(b[2 * i], x0, acc.v[1][t]);
      acc.v[1][t] = fmaf(b[2 * i + 1], x1, acc.v[1][t]);
    }
  }
}

// Finish the dot products of one unit whose first batch (ca, cb) is already in flight.
template <int TT>
__device__ __forceinline__ void dot_unit(const RowPair& r, const bf16_t* xs, int K, int lane, u32x4 (&ca)[U],
                                         u32x4 (&cb)[U], Acc<TT>& acc) {
  const int nch = (K + 511) >> 9;
#pragma unroll
  for (int t = 0; t < TT; ++t) acc.v[0][t] = acc.v[1][t] = 0.f;
  for (int c0 = 0; c0 < nch; c0 += U) {
    u32x4 na[U], nb[U];
    if (c0 + U < nch) load_batch(r, c0 + U, K, lane, na, nb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = ((c0 + u) * 64 + lane) * 8;
      if (e < K) fma_chunk<TT>(ca[u], cb[u], xs, K, e, acc);
    }
    if (c0 + U < nch) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ca[u] = na[u];
        cb[u] = nb[u];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    acc.v[0][t] = wave_sum(acc.v[0][t]);
    acc.v[1][t] = wave_sum(acc.v[1][t]);
  }
}

// Stage x[T, K] (rows x + t*ldx; rows t >= T are zero) into LDS, optionally RMS-normalised:
// bf16( bf16(x * rsqrt(mean(x^2) + eps)) * w )   (transformer_layers.py:115-120)
template <int TT>
__device__ __forceinline__ void stage_x(bf16_t* xs, float* red, const bf16_t* x, int ldx, int T, int K,
                                        const bf16_t* norm_w, float eps) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int npieces = K >> 3;
  float ss[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) ss[t] = 0.f;
  for (int p = tid; p < npieces; p += 256) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (t < T) v = ld16(x + (size_t)t * ldx + p * 8);
      st16(xs + (size_t)t * K + p * 8, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = bf_lo(v[i]), b = bf_hi(v[i]);
        ss[t] = fmaf(a, a, ss[t]);
        ss[t] = fmaf(b, b, ss[t]);
      }
    }
  }
  if (norm_w == nullptr) {
    __syncthreads();
    return;
  }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const float s = wave_sum(ss[t]);
    if (lane == 0) red[wid * TT + t] = s;
  }
  __syncthreads();
  float inv[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const float s = red[t] + red[TT + t] + red[2 * TT + t] + red[3 * TT + t];
    inv[t] = 1.0f / sqrtf(s / (float)K + eps);
  }
  for (int p = tid; p < npieces; p += 256) {
    const u32x4 wv = ld16(norm_w + p * 8);
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      u32x4 v = *reinterpret_cast<const u32x4*>(xs + (size_t)t * K + p * 8);
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y0 = bf_round(bf_lo(v[i]) * inv[t]) * bf_lo(wv[i]);
        const float y1 = bf_round(bf_hi(v[i]) * inv[t]) * bf_hi(wv[i]);
        o[i] = pack_bf2(y0, y1);
      }
      st16(xs + (size_t)t * K + p * 8, o);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ const bf16_t* seg_row(const GemvArgs& a, int r) {
  if (r < a.n0) return a.w0 + (size_t)r * a.K;
  if (r < a.n1) return a.w1 + (size_t)(r - a.n0) * a.K;
  return a.w2 + (size_t)(r - a.n1) * a.K;
}

template <int MODE>
__device__ __forceinline__ RowPair unit_rows(const GemvArgs& a, int u, const bf16_t* e1, const bf16_t* e3) {
  RowPair r;
  if (MODE == GEMV_SWIGLU) {
    r.a = a.w0 + (size_t)u * a.K;
    r.b = a.w1 + (size_t)u * a.K;
  } else if (MODE == GEMV_MOE_W13) {
    r.a = e1 + (size_t)u * a.K;
    r.b = e3 + (size_t)u * a.K;
  } else {
    r.a = seg_row(a, 2 * u);
    r.b = (2 * u + 1 < a.N) ? seg_row(a, 2 * u + 1) : nullptr;
  }
  return r;
}

template <int TT, int MODE>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + (size_t)TT * a.K * 2);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nwaves = gridDim.x * 4;
  const int units = (MODE == GEMV_SWIGLU || MODE == GEMV_MOE_W13) ? a.N : (a.N + 1) >> 1;

  // MoE: blockIdx.y is the problem (token, slot); pick this problem's expert and input row
  const bf16_t* x = a.x;
  const bf16_t *e1 = nullptr, *e3 = nullptr;
  char* outp = reinterpret_cast<char*>(a.out);
  if (MODE == GEMV_MOE_W13) {
    const int prob = blockIdx.y;
    const int e = a.sel_idx[prob];
    e1 = reinterpret_cast<const bf16_t*>(a.expert_tab[e * 3 + 0]);
    e3 = reinterpret_cast<const bf16_t*>(a.expert_tab[e * 3 + 2]);
    x = a.x + (size_t)(prob / a.top_k) * a.ldx;
    outp += (size_t)prob * a.ldo * 2;
  }

  int u = blockIdx.x * 4 + wid;
  u32x4 ca[U], cb[U];
  RowPair rp = {nullptr, nullptr};
  if (u < units) {
    rp = unit_rows<MODE>(a, u, e1, e3);
    load_batch(rp, 0, a.K, lane, ca, cb);
  }
  stage_x<TT>(xs, red, x, a.ldx, (MODE == GEMV_MOE_W13) ? 1 : a.T, a.K, a.norm_w, a.eps);

  while (u < units) {
    Acc<TT> acc;
    dot_unit<TT>(rp, xs, a.K, lane, ca, cb, acc);
    const int un = u + nwaves;
    if (un < units) {
      rp = unit_rows<MODE>(a, un, e1, e3);
      load_batch(rp, 0, a.K, lane, ca, cb);
    }
    // ---- epilogue: lane t finishes token t
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (lane == t) {
        v0 = acc.v[0][t];
        v1 = acc.v[1][t];
      }
    }
    const int T = (MODE == GEMV_MOE_W13) ? 1 : a.T;
    if (lane < T) {
      const int t = lane;
      if (MODE == GEMV_SWIGLU || MODE == GEMV_MOE_W13) {
        reinterpret_cast<bf16_t*>(outp)[(size_t)t * a.ldo + u] = f_to_bf(swiglu_bf(v0, v1));
      } else {
        const int r0 = 2 * u;
        const bool two = (r0 + 1 < a.N);
        if (MODE == GEMV_LOGITS) {
          float* o = reinterpret_cast<float*>(outp) + (size_t)t * a.ldo + r0;
          o[0] = bf_round(v0);
          if (two) o[1] = bf_round(v1);
        } else {
          float y0 = bf_round(v0), y1 = bf_round(v1);
          bf16_t* o = reinterpret_cast<bf16_t*>(outp) + (size_t)t * a.ldo + r0;
          if (MODE == GEMV_RESIDUAL) {
            const bf16_t* rs = a.residual + (size_t)t * a.ldo + r0;
            y0 = bf_to_f(rs[0]) + y0;
            if (two) y1 = bf_to_f(rs[1]) + y1;
          }
          if (MODE == GEMV_QKV_ROPE) {
            const int pos = a.tok_pos[t];
            if (r0 < a.n1) {  // q or k rows: rotate the adjacent pair (rope.py:13-23)
              const int i = (r0 % a.head_dim) >> 1;
              const float2 cs = *reinterpret_cast<const float2*>(a.rope_cs + ((size_t)pos * (a.head_dim >> 1) + i) * 2);
              const float re = __fsub_rn(__fmul_rn(y0, cs.x), __fmul_rn(y1, cs.y));
              const float im = __fadd_rn(__fmul_rn(y0, cs.y), __fmul_rn(y1, cs.x));
              y0 = re;
              y1 = im;
            }
            if (a.write_kv && r0 >= a.n0) {  // cache.py:83-92: ring slot pos % W of this sequence's row
              const int kv_dim = a.n1 - a.n0;
              const int seq = a.tok_seq ? a.tok_seq[t] : t;
              const size_t slot = (size_t)seq * a.W + (pos % a.W);
              bf16_t* ring = (r0 < a.n1) ? reinterpret_cast<bf16_t*>(a.cache_k) + slot * kv_dim + (r0 - a.n0)
                                         : reinterpret_cast<bf16_t*>(a.cache_v) + slot * kv_dim + (r0 - a.n1);
              *reinterpret_cast<uint32_t*>(ring) = pack_bf2(y0, y1);
            }
          }
          if (two) {
            *reinterpret_cast<uint32_t*>(o) = pack_bf2(y0, y1);
          } else {
            o[0] = f_to_bf(y0);
          }
        }
      }
    }
    u = un;
  }
}

// MoE down-projection + combine for one token per blockIdx.y (moe.py:28-32 at decode):
// out[t] = bf16(h[t] + R),  R = sum over the token's experts in ascending id of bf16(w_e * bf16(W2_e . g_e)),
// accumulated in bf16 starting from zero.
template <int TOPK>
__global__ __launch_bounds__(256) void moe_w2_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [TOPK][K]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int t = blockIdx.y;
  const int nwaves = gridDim.x * 4;
  const int units = (a.N + 1) >> 1;

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

  int u = blockIdx.x * 4 + wid;
  u32x4 ca[U], cb[U];
  RowPair rp = {nullptr, nullptr};
  auto rows = [&](int k, int uu) {
    RowPair r;
    r.a = w2[k] + (size_t)(2 * uu) * a.K;
    r.b = (2 * uu + 1 < a.N) ? w2[k] + (size_t)(2 * uu + 1) * a.K : nullptr;
    return r;
  };
  if (u < units) {
    rp = rows(0, u);
    load_batch(rp, 0, a.K, lane, ca, cb);
  }
  // stage the TOPK hidden rows of this token (rows of a.x are [T*TOPK, K], slot-major per token)
  for (int p = tid; p < (a.K >> 3) * TOPK; p += 256) {
    const int k = p / (a.K >> 3), pp = p % (a.K >> 3);
    st16(xs + (size_t)k * a.K + pp * 8, ld16(a.x + ((size_t)t * TOPK + slot_of[k]) * a.ldx + pp * 8));
  }
  __syncthreads();

  while (u < units) {
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int k = 0; k < TOPK; ++k) {
      Acc<1> acc;
      dot_unit<1>(rp, xs + (size_t)k * a.K, a.K, lane, ca, cb, acc);
      // next rows: next expert of this unit, or first expert of the next unit
      const int nk = (k + 1 < TOPK) ? k + 1 : 0;
      const int nu = (k + 1 < TOPK) ? u : u + nwaves;
      if (nu < units) {
        rp = rows(nk, nu);
        load_batch(rp, 0, a.K, lane, ca, cb);
      }
      r0 = bf_round(r0 + bf_round(ew[k] * bf_round(acc.v[0][0])));
      r1 = bf_round(r1 + bf_round(ew[k] * bf_round(acc.v[1][0])));
    }
    if (lane == 0) {
      const int n = 2 * u;
      const bf16_t* rs = a.residual + (size_t)t * a.ldo + n;
      bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)t * a.ldo + n;
      o[0] = f_to_bf(bf_to_f(rs[0]) + r0);
      if (n + 1 < a.N) o[1] = f_to_bf(bf_to_f(rs[1]) + r1);
    }
    u += nwaves;
  }
}

template <int MODE>
hipError_t launch_mode(const GemvArgs& a, int TT, dim3 grid, size_t lds, hipStream_t s) {
  switch (TT) {
    case 1: hipLaunchKernelGGL((gemv_kernel<1, MODE>), grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL((gemv_kernel<2, MODE>), grid, dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL((gemv_kernel<3, MODE>), grid, dim3(256), lds, s, a); break;
    case 4: hipLaunchKernelGGL((gemv_kernel<4, MODE>), grid, dim3(256), lds, s, a); break;
    case 6: hipLaunchKernelGGL((gemv_kernel<6, MODE>), grid, dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL((gemv_kernel<8, MODE>), grid, dim3(256), lds, s, a); break;
  }
  return hipGetLastError();
}

int g_gemv_max_blocks = 0;

}  // namespace

int gemv_max_tokens(int K) {
  int t = (int)(GEMV_LDS_BUDGET / ((size_t)K * 2));
  return t < 1 ? 1 : (t > GEMV_MAX_T ? GEMV_MAX_T : t);
}

// One launch; a.T must be <= gemv_max_tokens(K).
hipError_t launch_gemv(const GemvArgs& a, hipStream_t s) {
  if (g_gemv_max_blocks == 0) {
    const char* e = getenv("MI_GEMV_MAX_BLOCKS");
    g_gemv_max_blocks = e ? atoi(e) : 2048;
    if (g_gemv_max_blocks <= 0) g_gemv_max_blocks = 2048;
  }
  const bool pair_mode = !(a.mode == GEMV_SWIGLU || a.mode == GEMV_MOE_W13);
  const int units = pair_mode ? (a.N + 1) / 2 : a.N;
  int blocks = (units + 3) / 4;
  if (blocks > g_gemv_max_blocks) blocks = g_gemv_max_blocks;
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
  const size_t lds = (size_t)TT * a.K * 2 + 4 * TT * sizeof(float);
  dim3 grid(blocks, a.mode == GEMV_MOE_W13 ? a.T * a.top_k : 1);
  switch (a.mode) {
    case GEMV_STORE: return launch_mode<GEMV_STORE>(a, TT, grid, lds, s);
    case GEMV_RESIDUAL: return launch_mode<GEMV_RESIDUAL>(a, TT, grid, lds, s);
    case GEMV_SWIGLU: return launch_mode<GEMV_SWIGLU>(a, TT, grid, lds, s);
    case GEMV_LOGITS: return launch_mode<GEMV_LOGITS>(a, TT, grid, lds, s);
    case GEMV_QKV_ROPE: return launch_mode<GEMV_QKV_ROPE>(a, TT, grid, lds, s);
    case GEMV_MOE_W13: return launch_mode<GEMV_MOE_W13>(a, 1, grid, lds, s);
    default: return hipErrorInvalidValue;
  }
}
