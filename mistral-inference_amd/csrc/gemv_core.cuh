// Core of the weight-streaming GEMV (see gemv.hip for the design notes): batch loads, FMA, activation staging and the
// per-wave unit loop.
#pragma once
#include "common.cuh"
#include "kernels.h"

namespace gemv_core {

constexpr int BATCH = 8;  // 16-byte loads per lane per batch (ROWS rows x BATCH/ROWS chunks); two batches in flight

template <int TT>
struct Acc {
  float v[2][TT];
};

struct RowPair {
  const bf16_t* a;
  const bf16_t* b;  // nullptr when the unit has one row
};

// One batch = chunks [c0, c0 + BATCH/ROWS) of each of the unit's ROWS rows: always exactly BATCH unconditional loads.
// Chunk offsets past K are clamped to the row's last 16 bytes (fma_batch skips them) and a missing second row aliases
// the first (the epilogue drops it).  Never a `cond ? load : 0`: that makes hipcc branch around each load and wait
// vmcnt(0) after it (cdna_hip_programming.md, ".s-level traps" (c)).
template <int ROWS>
__device__ __forceinline__ void load_batch(const RowPair& r, int c0, int K, int lane, u32x4 (&buf)[BATCH]) {
  constexpr int U = BATCH / ROWS;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = min(((c0 + u) * 64 + lane) * 8, K - 8);
    buf[u] = ld16_nt(r.a + e);
    if (ROWS == 2) buf[U + u] = ld16_nt(r.b + e);
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
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + (size_t)t * K + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc.v[0][t] = dot2_bf16(buf[u][i], xv[i], acc.v[0][t]);
          if (ROWS == 2) acc.v[1][t] = dot2_bf16(buf[U + u][i], xv[i], acc.v[1][t]);
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

// Activations that do not fit the register set (batches of >= 3 tokens; W2 rows of >= 2 tokens) go to LDS by LDS-DMA
// (global_load_lds_dwordx4: a wave-instruction moves 64 consecutive 16-byte pieces, no registers, all of them in flight at
// once), the norm weights behind them; x_finish waits ONCE and runs the RMSNorm passes on LDS.  Before round 6 this case was
// a load -> wait -> store loop per 256 pieces: up to 21 dependent L2 round trips in front of the first FMA of every GEMV of a
// batch-3 decode step (q|k|v 19.3 us against 13.0 at batch 1).  Piece q of `rows` rows goes to dst + q * 16; rows >= T are zeros.
__device__ __forceinline__ void dma_rows_to_lds(const bf16_t* src, int ld, int rows, int T, int npieces, char* dst) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int total = rows * npieces;
  for (int q0 = wid * 64; q0 < total; q0 += 256) {
    const int q = q0 + lane;
    const int t = q / npieces, pc = q - t * npieces;
    if (q < total) {
      if (t < T) {
        // from inline asm (M0 = the wave's LDS base, saved and restored): hipcc does not track it, so it neither drains the
        // weight loads at the prologue's barriers (as it does behind the builtin) nor counts it - the waits behind these DMAs are
        // the caller's (x_finish: vmcnt(16) with exactly the two weight batches issued after them)
        unsigned keep;
        const bf16_t* sp = src + (size_t)t * ld + pc * 8;
        const uint32_t lds_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(dst + (size_t)q0 * 16);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(sp), "s"(lds_addr) : "memory");
      } else {
        st16(dst + (size_t)q * 16, u32x4{0u, 0u, 0u, 0u});
      }
    }
  }
}

// Issues exactly NX + NW loads (clamped / dummy where there is nothing to load); when the rows do not fit them, the DMAs above.
// xs: the LDS image [TT][K]; ws: K norm weights behind it (used by the DMA path of the norm modes only).
// DMA is a property of the INSTANTIATION (launch_gemv picks it when the rows cannot fit the registers): with an LDS-DMA anywhere in
// a function hipcc waits vmcnt(0) at every barrier and gives up the counted waits of the register path.
template <int TT, int NX, int NW, bool DMA, int NT = 256>
__device__ __forceinline__ bool x_issue(XRegs<NX, NW>& xr, const bf16_t* x, int ldx, int T, int K, const bf16_t* norm_w, bf16_t* xs,
                                        bf16_t* ws) {
  const int npieces = K >> 3;
  const int total = TT * npieces;
  if constexpr (DMA) {
    dma_rows_to_lds(x, ldx, TT, T, npieces, reinterpret_cast<char*>(xs));
    if (norm_w) dma_rows_to_lds(norm_w, 0, 1, 1, npieces, reinterpret_cast<char*>(ws));
    return false;
  }
  const bool fits = total <= NX * NT && (norm_w == nullptr || total <= NW * NT);
  const bf16_t* wsrc = norm_w ? norm_w : x;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int q = min((int)threadIdx.x + i * NT, total - 1);
    const int t = q / npieces, p = q - t * npieces;
    xr.x[i] = ld16(x + (size_t)min(t, T - 1) * ldx + p * 8);
    if (i < NW) xr.w[i] = ld16(wsrc + p * 8);
  }
  return fits;
}

template <int TT, int NX, int NW, bool DMA, int NT = 256>
__device__ __forceinline__ void x_finish(bool in_regs, XRegs<NX, NW>& xr, bf16_t* xs, float* red, const bf16_t* ws,
                                         const bf16_t* x, int ldx, int T, int K, const bf16_t* norm_w, float eps) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int npieces = K >> 3;
  float ss[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) ss[t] = 0.f;
  if (in_regs) {
    const int total = TT * npieces;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = tid + i * NT;
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
  } else if constexpr (DMA) {
    // the rows (and norm weights) were sent to LDS by x_issue's DMAs.  vmcnt retires in order and EXACTLY the two weight batches
    // (2 x BATCH unconditional loads: gemv_body) were issued behind them: vmcnt(2 * BATCH) = the DMAs have landed, the weights
    // stay in flight under the passes below, which read LDS only.
    static_assert(BATCH == 8, "the wait below counts the two weight batches");
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
    if (norm_w != nullptr) {
      for (int p = tid; p < npieces; p += 256) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(xs + (size_t)t * K + p * 8);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float a = bf_lo(v[c]), b = bf_hi(v[c]);
            ss[t] = fmaf(a, a, ss[t]);
            ss[t] = fmaf(b, b, ss[t]);
          }
        }
      }
    }
  } else {
    for (int p = tid; p < npieces; p += NT) {
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
    if (!DMA) __syncthreads();  // (the DMA path has had its barrier)
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
      const u32x4 wv = DMA ? *reinterpret_cast<const u32x4*>(ws + p * 8) : ld16(norm_w + p * 8);
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
// Order: activation loads first (L2 hits), then two weight batches, and the prologue finishes under them.
// NWV = waves per block: 4; 5 .. 8 for the plain modes (no fused RMSNorm: its passes are written for 256 threads) when the row
// count has no even split over 4-wave blocks - Mistral-Nemo's 5120 rows are 2560 pairs = 256 CUs x 10 (launch_gemv).
template <int TT, int MODE, int ROWS, bool DMA, int NWV = 4>
__device__ __forceinline__ void gemv_body(const GemvArgs& a, char* smem, int block_id, int n_blocks, int problem) {
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + (size_t)TT * a.K * 2);
  bf16_t* ws = reinterpret_cast<bf16_t*>(smem + (size_t)TT * a.K * 2 + 16 * TT);  // (launch_gemv reserves it for TT > 1)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: unit loops become scalar
  const int nwaves = n_blocks * NWV;
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
    const int prob = problem;
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
  static_assert(NWV == 4 || (!kNormMode && !DMA), "other block sizes: plain modes on the register staging path");
  XRegs<NX, NW> xr;
  bool in_regs = false;
  // q|k|v epilogue operands that depend on nothing: the token's position (RoPE row, ring slot) and its sequence (ring row) are the
  // FIRST loads of the wave.  Loaded behind the weight batches (rounds 1-6), the position - which the first RoPE prefetch needs
  // for its address - made hipcc wait vmcnt(0) in front of the first FMA: both weight batches drained, then two dependent L2
  // round trips (position, RoPE entry), and a third per unit for the sequence in the epilogue.  As the oldest loads they are
  // waited for with a counted vmcnt that leaves the weights in flight.
  const int tl = lane < T ? lane : 0;
  int ep_pos = 0, ep_seq = 0;
  if (MODE == GEMV_QKV_ROPE) {
    ep_pos = a.tok_pos[tl];
    ep_seq = a.tok_seq ? a.tok_seq[tl] : tl;
  }
  in_regs = x_issue<TT, NX, NW, DMA, NWV * 64>(xr, x, a.ldx, T, a.K, a.norm_w, xs, ws);

  // load cursor over the flattened (unit, batch) sequence of this wave: always two batches ahead of the math
  int u = block_id * NWV + wid;
  int ul = u, jl = 0;
  RowPair rpl = unit_rows<MODE, ROWS>(a, min(ul, units - 1), e1, e3);
  u32x4 bufA[BATCH], bufB[BATCH];
  // Past the wave's last unit `issue` loads BATCH times one L2-resident line instead of branching around the loads:
  // the row pointers and chunk offset are SELECTED (real rows, or one dummy line) and the BATCH loads are issued
  // unconditionally, which keeps the loop body one basic block and the compiler's wait for one buffer at
  // "the other buffer's BATCH loads may stay in flight".  The trailing loads are never consumed.
  const RowPair dummy = {x, x};
  auto issue = [&](u32x4 (&buf)[BATCH]) {
    const bool live = ul < units;
    RowPair r;
    r.a = live ? rpl.a : dummy.a;
    r.b = live ? rpl.b : dummy.b;
    load_batch<ROWS>(r, live ? jl * U : 0, live ? a.K : 8, live ? lane : 0, buf);
    if (live && ++jl == nb) {
      jl = 0;
      ul += nwaves;
      if (ul < units) rpl = unit_rows<MODE, ROWS>(a, ul, e1, e3);
    }
  };
  issue(bufA);
  issue(bufB);
  x_finish<TT, NX, NW, DMA, NWV * 64>(in_regs, xr, xs, red, ws, x, a.ldx, T, a.K, a.norm_w, a.eps);

  Acc<TT> acc;
#pragma unroll
  for (int t = 0; t < TT; ++t) acc.v[0][t] = acc.v[1][t] = 0.f;
  int jc = 0;

  // Epilogue operands are fetched EARLY (token position once, the unit's RoPE entry / residual pair when the unit
  // starts) so that the end of a unit is arithmetic + one store instead of a chain of dependent loads.
  float2 ep_cs = make_float2(1.f, 0.f);
  uint32_t ep_res = 0;
  auto prefetch_epilogue = [&](int uu) {
    const int r0 = (ROWS == 2) ? 2 * uu : uu;
    if (MODE == GEMV_QKV_ROPE && r0 < a.n1) {
      const int i = (r0 % a.head_dim) >> 1;
      ep_cs = *reinterpret_cast<const float2*>(a.rope_cs + ((size_t)ep_pos * (a.head_dim >> 1) + i) * 2);
    }
    if (MODE == GEMV_RESIDUAL) {
      const bf16_t* rs = a.residual + (size_t)tl * a.ldo + r0;
      if (ROWS == 2 && r0 + 1 < a.N) ep_res = *reinterpret_cast<const uint32_t*>(rs);
      else ep_res = rs[0];
    }
  };
  if (u < units) prefetch_epilogue(u);

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
        bf16_t* o = reinterpret_cast<bf16_t*>(outp) + (size_t)t * a.ldo + u;
        *o = f_to_bf(swiglu_bf(v0, v1));
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
            y0 = bf_lo(ep_res) + y0;
            if (two) y1 = bf_hi(ep_res) + y1;
          }
          if (MODE == GEMV_QKV_ROPE) {
            const int pos = ep_pos;
            if (r0 < a.n1) {  // q or k rows: rotate the adjacent pair (rope.py:13-23)
              const float2 cs = ep_cs;
              float re, im;
              rope_pair(y0, y1, cs.x, cs.y, re, im);
              y0 = re;
              y1 = im;
            }
            if (a.write_kv && r0 >= a.n0) {  // cache.py:83-92: ring slot pos % W of this sequence's row
              const int kv_dim = a.n1 - a.n0;
              const size_t off = kv_offset(a.kv_layout, a.W, kv_dim, a.head_dim, (size_t)ep_seq, pos % a.W, (r0 < a.n1) ? r0 - a.n0 : r0 - a.n1);
              bf16_t* ring = ((r0 < a.n1) ? reinterpret_cast<bf16_t*>(a.cache_k) : reinterpret_cast<bf16_t*>(a.cache_v)) + off;
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
    fma_batch<TT, ROWS>(buf, jc * U, xs, a.K, lane, acc);
    issue(buf);
    if (++jc == nb) {
      jc = 0;
      if (u < units) finish_unit();
      u += nwaves;
      if (u < units) prefetch_epilogue(u);
    }
  };
  // ALWAYS both steps per trip (a step past the wave's last unit multiplies dummy lines and stores nothing).  With the second
  // step under `if (u < units)` - rounds 1-6 - hipcc's wait-count pass merged the skip edge into the loop header and guarded
  // the FIRST step's operands with vmcnt(6..0) instead of vmcnt(14..8): every trip drained both batches before its first FMA
  // and the "two batches in flight" of the design was one.
  if (u < units) {
    do {
      step(bufA);
      step(bufB);
    } while (u < units);
  }
}

}  // namespace gemv_core
