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
#include "kernels.h"

namespace {

constexpr int BATCH = 8;  // 16-byte loads per lane per batch (ROWS rows x BATCH/ROWS chunks); two batches in flight

template <int TT>
struct Acc {
  float v[2][TT];
};

struct RowPair {
  const bf16_t* a;
  const bf16_t* b;  // nullptr when the unit has one row
};

// One batch = chunks [c0, c0 + BATCH/ROWS) of each of the unit's ROWS rows: always exactly BATCH asm loads, so the
// hand-written vmcnt counts are static.  Chunk offsets past K are clamped to the row's last 16 bytes (fma_batch
// skips them) and a missing second row aliases the first (the epilogue drops it).  Never a `cond ? load : 0`: that
// makes hipcc branch around each load and wait vmcnt(0) after it (cdna_hip_programming.md, ".s-level traps" (c)).
template <int ROWS>
__device__ __forceinline__ void load_batch(const RowPair& r, int c0, int K, int lane, u32x4 (&buf)[BATCH]) {
  constexpr int U = BATCH / ROWS;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = min(((c0 + u) * 64 + lane) * 8, K - 8);
    ld16_asm_nt(buf[u], r.a + e);
    if (ROWS == 2) ld16_asm_nt(buf[U + u], r.b + e);
  }
}

template <int TT, int ROWS>
__device__ __forceinline__ void fma_batch(const u32x4 (&buf)[BATCH], int c0, const bf16_t* xs, int K, int lane,
                                          Acc<TT>& acc) {
  constexpr int U = BATCH / ROWS;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = ((c0 + u) * 64 + lane) * 8;
    if (e < K) {
      float a[8], b[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[2 * i] = bf_lo(buf[u][i]);
        a[2 * i + 1] = bf_hi(buf[u][i]);
        if (ROWS == 2) {
          b[2 * i] = bf_lo(buf[U + u][i]);
          b[2 * i + 1] = bf_hi(buf[U + u][i]);
        }
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + (size_t)t * K + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x0 = bf_lo(xv[i]), x1 = bf_hi(xv[i]);
          acc.v[0][t] = fmaf(a[2 * i], x0, acc.v[0][t]);
          acc.v[0][t] = fmaf(a[2 * i + 1], x1, acc.v[0][t]);
          if (ROWS == 2) {
            acc.v[1][t] = fmaf(b[2 * i], x0, acc.v[1][t]);
            acc.v[1][t] = fmaf(b[2 * i + 1], x1, acc.v[1][t]);
          }
        }
      }
    }
  }
}

// ---- activation staging.  x[T, K] (rows t >= T are zero) goes to LDS, optionally RMS-normalised:
// bf16( bf16(x * rsqrt(mean(x^2) + eps)) * w )   (transformer_layers.py:115-120).
// Split in two so that the x (and norm weight) loads are the FIRST loads the wave issues - they are L2 hits and
// return long before the HBM weight batches issued right after them, so the whole prologue runs under the
// weight latency instead of in front of it.
// Activation pieces (16 B) a thread holds in registers while the weight batches are issued.  Modes that fuse the
// RMSNorm (K = model dim <= 8192 for one token) hold NX = 4 x pieces + NW = 4 norm-weight pieces; the plain modes
// (Wo, W2: K up to 16384) hold NX = 8 x pieces and no norm weights.  Anything larger takes the in-loop path.
template <int NX, int NW>
struct XRegs {
  u32x4 x[NX];
  u32x4 w[NW > 0 ? NW : 1];
};

// Issues exactly NX + NW asm loads (clamped / dummy where there is nothing to load).
template <int TT, int NX, int NW>
__device__ __forceinline__ bool x_issue(XRegs<NX, NW>& xr, const bf16_t* x, int ldx, int T, int K, const bf16_t* norm_w) {
  const int npieces = K >> 3;
  const int total = TT * npieces;
  const bool fits = total <= NX * 256 && (norm_w == nullptr || total <= NW * 256);
  const bf16_t* wsrc = norm_w ? norm_w : x;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int q = min((int)threadIdx.x + i * 256, total - 1);
    const int t = q / npieces, p = q - t * npieces;
    ld16_asm(xr.x[i], x + (size_t)min(t, T - 1) * ldx + p * 8);
    if (i < NW) ld16_asm(xr.w[i], wsrc + p * 8);
  }
  return fits;
}

template <int NX, int NW, int AFTER>
__device__ __forceinline__ void x_wait(XRegs<NX, NW>& xr) {
  if constexpr (NX == 8) vm_wait8<AFTER>(xr.x);
  if constexpr (NX == 4) vm_wait4<AFTER>(xr.x);
  if constexpr (NW == 4) vm_wait4<AFTER>(xr.w);
}

// Exactly two weight batches (2 * BATCH loads) are issued between x_issue and x_finish.
template <int TT, int NX, int NW>
__device__ __forceinline__ void x_finish(bool in_regs, XRegs<NX, NW>& xr, bf16_t* xs, float* red,
                                         const bf16_t* x, int ldx, int T, int K, const bf16_t* norm_w, float eps) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int npieces = K >> 3;
  float ss[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) ss[t] = 0.f;
  x_wait<NX, NW, 2 * BATCH>(xr);  // both weight batches stay in flight under the prologue
  if (in_regs) {
    const int total = TT * npieces;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = tid + i * 256;
      if (q < total) {
        const int t = q / npieces;
        if (t >= T) xr.x[i] = u32x4{0u, 0u, 0u, 0u};
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = bf_lo(xr.x[i][c]), b = bf_hi(xr.x[i][c]);
          s = fmaf(a, a, s);
          s = fmaf(b, b, s);
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) ss[tt] += (tt == t) ? s : 0.f;
        if (!norm_w) st16(xs + (size_t)q * 8, xr.x[i]);  // [t][K] row-major == q * 8
      }
    }
  } else {
    for (int p = tid; p < npieces; p += 256) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const u32x4 ld = ld16(x + (size_t)min(t, T - 1) * ldx + p * 8);
        const u32x4 z = {0u, 0u, 0u, 0u};
        const u32x4 v = (t < T) ? ld : z;
        st16(xs + (size_t)t * K + p * 8, v);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = bf_lo(v[c]), b = bf_hi(v[c]);
          ss[t] = fmaf(a, a, ss[t]);
          ss[t] = fmaf(b, b, ss[t]);
        }
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
  if (in_regs) {
    if constexpr (NW > 0) {
      const int total = TT * npieces;
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int q = tid + i * 256;
        if (q < total) {
          const int t = q / npieces;
          float iv = 0.f;
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) iv = (tt == t) ? inv[tt] : iv;
          u32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            o[c] = pack_bf2(bf_round(bf_lo(xr.x[i][c]) * iv) * bf_lo(xr.w[i < NW ? i : 0][c]),
                            bf_round(bf_hi(xr.x[i][c]) * iv) * bf_hi(xr.w[i < NW ? i : 0][c]));
          st16(xs + (size_t)q * 8, o);
        }
      }
    }
  } else {
    for (int p = tid; p < npieces; p += 256) {
      const u32x4 wv = ld16(norm_w + p * 8);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(xs + (size_t)t * K + p * 8);
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          o[c] = pack_bf2(bf_round(bf_lo(v[c]) * inv[t]) * bf_lo(wv[c]), bf_round(bf_hi(v[c]) * inv[t]) * bf_hi(wv[c]));
        st16(xs + (size_t)t * K + p * 8, o);
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ const bf16_t* seg_row(const GemvArgs& a, int r) {
  if (r < a.n0) return a.w0 + (size_t)r * a.K;
  if (r < a.n1) return a.w1 + (size_t)(r - a.n0) * a.K;
  return a.w2 + (size_t)(r - a.n1) * a.K;
}

template <int MODE, int ROWS>
__device__ __forceinline__ RowPair unit_rows(const GemvArgs& a, int u, const bf16_t* e1, const bf16_t* e3) {
  RowPair r;
  if (MODE == GEMV_SWIGLU) {
    r.a = a.w0 + (size_t)u * a.K;
    r.b = a.w1 + (size_t)u * a.K;
  } else if (MODE == GEMV_MOE_W13) {
    r.a = e1 + (size_t)u * a.K;
    r.b = e3 + (size_t)u * a.K;
  } else if (ROWS == 1) {
    r.a = seg_row(a, u);
    r.b = r.a;
  } else {
    r.a = seg_row(a, 2 * u);
    r.b = (2 * u + 1 < a.N) ? seg_row(a, 2 * u + 1) : r.a;  // odd N: alias, result dropped in the epilogue
  }
  return r;
}

// ROWS = rows per unit (2 everywhere except the plain/residual/logits modes on small N, where single-row units
// double the number of waves so that a 4096-row matrix still fills 256 CUs x 16 waves).
template <int TT, int MODE, int ROWS>
__global__ __launch_bounds__(256, (TT == 1 ? 4 : (TT <= 3 ? 3 : 2))) void gemv_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + (size_t)TT * a.K * 2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: unit loops become scalar
  const int nwaves = gridDim.x * 4;
  constexpr bool kPairOut = !(MODE == GEMV_SWIGLU || MODE == GEMV_MOE_W13);
  const int units = kPairOut ? (ROWS == 2 ? (a.N + 1) >> 1 : a.N) : a.N;
  constexpr int U = BATCH / ROWS;
  const int nch = (a.K + 511) >> 9;
  const int nb = (nch + U - 1) / U;  // batches per unit

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
  const int T = (MODE == GEMV_MOE_W13) ? 1 : a.T;

  // 1. activation (and norm weight) loads first, 2. two weight batches, 3. finish the prologue under them
  constexpr bool kNormMode = MODE == GEMV_QKV_ROPE || MODE == GEMV_SWIGLU || MODE == GEMV_LOGITS || MODE == GEMV_MOE_W13;
  constexpr int NX = kNormMode ? 4 : 8, NW = kNormMode ? 4 : 0;
  XRegs<NX, NW> xr;
  const bool in_regs = x_issue<TT, NX, NW>(xr, x, a.ldx, T, a.K, a.norm_w);

  // load cursor over the flattened (unit, batch) sequence of this wave: always two batches ahead of the math
  int u = blockIdx.x * 4 + wid;
  int ul = u, jl = 0;
  RowPair rpl = unit_rows<MODE, ROWS>(a, min(ul, units - 1), e1, e3);
  u32x4 bufA[BATCH], bufB[BATCH];
  // Past the wave's last unit `issue` loads BATCH times one L2-resident line instead, so that every wait below can
  // use the static count "the other buffer's BATCH loads may stay in flight" (branching between counted and draining
  // waits makes hipcc spill the buffers).  Those trailing loads are never consumed; that is safe because bufA/bufB
  // are loop-carried (their registers are not reused inside the loop) and nothing executes after the loop.
  const RowPair dummy = {x, x};
  auto issue = [&](u32x4 (&buf)[BATCH]) {
    if (ul < units) {
      load_batch<ROWS>(rpl, jl * U, a.K, lane, buf);
      if (++jl == nb) {
        jl = 0;
        ul += nwaves;
        if (ul < units) rpl = unit_rows<MODE, ROWS>(a, ul, e1, e3);
      }
    } else {
      load_batch<ROWS>(dummy, 0, 8, 0, buf);
    }
  };
  issue(bufA);
  issue(bufB);
  x_finish<TT, NX, NW>(in_regs, xr, xs, red, x, a.ldx, T, a.K, a.norm_w, a.eps);

  Acc<TT> acc;
#pragma unroll
  for (int t = 0; t < TT; ++t) acc.v[0][t] = acc.v[1][t] = 0.f;
  int jc = 0;

  auto finish_unit = [&]() {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      acc.v[0][t] = wave_sum(acc.v[0][t]);
      if (ROWS == 2) acc.v[1][t] = wave_sum(acc.v[1][t]);
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
    if (lane < T) {
      const int t = lane;
      if (MODE == GEMV_SWIGLU || MODE == GEMV_MOE_W13) {
        reinterpret_cast<bf16_t*>(outp)[(size_t)t * a.ldo + u] = f_to_bf(swiglu_bf(v0, v1));
      } else {
        const int r0 = (ROWS == 2) ? 2 * u : u;
        const bool two = (ROWS == 2) && (r0 + 1 < a.N);
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
#pragma unroll
    for (int t = 0; t < TT; ++t) acc.v[0][t] = acc.v[1][t] = 0.f;
  };

  // One step = consume the oldest batch, refill the same registers with the batch two ahead (ping-pong between
  // bufA and bufB: no register copies, so the compiler's wait for bufA leaves bufB's eight loads in flight).
  auto step = [&](u32x4 (&buf)[BATCH]) {
    vm_wait8<BATCH>(buf);  // the other buffer's BATCH loads were issued after this one's and may stay in flight
    fma_batch<TT, ROWS>(buf, jc * U, xs, a.K, lane, acc);
    issue(buf);
    if (++jc == nb) {
      jc = 0;
      finish_unit();
      u += nwaves;
    }
  };
  while (u < units) {
    step(bufA);
    if (u < units) step(bufB);
  }
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
  auto issue = [&](u32x4 (&buf)[BATCH]) {
    if (ul < units) {
      RowPair r;
      const bf16_t* base = w2[0];
#pragma unroll
      for (int k = 1; k < TOPK; ++k) base = (kl == k) ? w2[k] : base;
      r.a = base + (size_t)(2 * ul) * a.K;
      r.b = (2 * ul + 1 < a.N) ? base + (size_t)(2 * ul + 1) * a.K : r.a;
      load_batch<2>(r, jl * U, a.K, lane, buf);
      if (++jl == nb) {
        jl = 0;
        if (++kl == TOPK) {
          kl = 0;
          ul += nwaves;
        }
      }
    } else {
      const RowPair dummy = {a.x, a.x};  // see gemv_kernel: keeps the wait counts static
      load_batch<2>(dummy, 0, 8, 0, buf);
    }
  };
  issue(bufA);
  issue(bufB);
  // stage the TOPK hidden rows of this token (rows of a.x are [T*TOPK, K], slot-major per token)
  for (int p = tid; p < (a.K >> 3) * TOPK; p += 256) {
    const int k = p / (a.K >> 3), pp = p % (a.K >> 3);
    int so = slot_of[0];
#pragma unroll
    for (int kk = 1; kk < TOPK; ++kk) so = (k == kk) ? slot_of[kk] : so;
    st16(xs + (size_t)k * a.K + pp * 8, ld16(a.x + ((size_t)t * TOPK + so) * a.ldx + pp * 8));
  }
  __syncthreads();

  Acc<1> acc;
  acc.v[0][0] = acc.v[1][0] = 0.f;
  int jc = 0, kc = 0;
  float r0 = 0.f, r1 = 0.f;
  auto step = [&](u32x4 (&buf)[BATCH]) {
    vm_wait8<BATCH>(buf);
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
        if (lane == 0) {
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
  while (u < units) {
    step(bufA);
    if (u < units) step(bufB);
  }
}

template <int MODE, int ROWS>
hipError_t launch_mode(const GemvArgs& a, int TT, dim3 grid, size_t lds, hipStream_t s) {
  switch (TT) {
    case 1: hipLaunchKernelGGL((gemv_kernel<1, MODE, ROWS>), grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL((gemv_kernel<2, MODE, ROWS>), grid, dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL((gemv_kernel<3, MODE, ROWS>), grid, dim3(256), lds, s, a); break;
    case 4: hipLaunchKernelGGL((gemv_kernel<4, MODE, ROWS>), grid, dim3(256), lds, s, a); break;
    case 6: hipLaunchKernelGGL((gemv_kernel<6, MODE, ROWS>), grid, dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL((gemv_kernel<8, MODE, ROWS>), grid, dim3(256), lds, s, a); break;
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
    g_gemv_max_blocks = e ? atoi(e) : 4096;
    if (g_gemv_max_blocks <= 0) g_gemv_max_blocks = 4096;
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
