// Memory-bound elementwise / row kernels of the hot path.  All bf16 traffic is 16 bytes per lane.
#include "common.cuh"
#include "kernels.h"

namespace {

// transformer.py:193
// An id outside [0, vocab) makes nn.Embedding raise (reference transformer.py:193).  A kernel cannot raise: it records
// 1 + the token index in *bad_id (when given), reads row 0 / vocab-1 instead of faulting, and the host turns the flag
// into the IndexError at its next synchronisation point (ids that live on the host are checked before the launch).
__device__ __forceinline__ long checked_id(long id, int vocab, int t, uint32_t* bad_id) {
  if (id < 0 || id >= vocab) {
    if (bad_id && threadIdx.x == 0) atomicMax(bad_id, (uint32_t)t + 1u);
    id = id < 0 ? 0 : vocab - 1;
  }
  return id;
}

__global__ __launch_bounds__(256) void embedding_kernel(bf16_t* out, const bf16_t* table, const int64_t* ids, int D,
                                                        int vocab, uint32_t* bad_id) {
  const int t = blockIdx.x;
  const long id = checked_id(ids[t], vocab, t, bad_id);
  const bf16_t* src = table + (size_t)id * D;
  bf16_t* dst = out + (size_t)t * D;
  for (int p = threadIdx.x; p < (D >> 3); p += 256) st16(dst + p * 8, ld16(src + p * 8));
}

// transformer_layers.py:115-120.  One WAVE per row (4 rows per block): the row sits in registers between the two passes,
// the sum of squares is a wave reduction - no LDS, no barrier - and every lane's NP 16-byte loads are issued
// unconditionally from clamped addresses before anything waits (a `p < np ? load : 0` form costs a branch and a full
// vmcnt drain per load, cdna_hip_programming.md ".s-level traps" (c)).
template <int NP>  // 16-byte pieces per lane: D <= 64 * 8 * NP
__global__ __launch_bounds__(256) void rmsnorm_kernel(bf16_t* out, const bf16_t* x, const bf16_t* w, int T, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (t >= T) return;
  const bf16_t* xr = x + (size_t)t * D;
  const int np = D >> 3;
  u32x4 v[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) v[j] = ld16(xr + min(lane + j * 64, np - 1) * 8);
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = bf_lo(v[j][i]), b = bf_hi(v[j][i]);
      s = fmaf(a, a, s);
      s = fmaf(b, b, s);
    }
    ss += (lane + j * 64 < np) ? s : 0.f;
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss / (float)D + eps);
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int p = lane + j * 64;
    if (p < np) {
      const u32x4 wv = ld16(w + p * 8);
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        o[i] = pack_bf2(bf_round(bf_lo(v[j][i]) * inv) * bf_lo(wv[i]), bf_round(bf_hi(v[j][i]) * inv) * bf_hi(wv[i]));
      st16(out + (size_t)t * D + p * 8, o);
    }
  }
}

// rope.py:13-23, in place on the q|k columns of the fused buffer.  One thread rotates 4 adjacent pairs.
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* qkv, int ld, int T, int n_rot_cols, int Dh,
                                                   const float* rope_cs, const int32_t* tok_pos) {
  const int pieces_per_row = n_rot_cols >> 3;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)T * pieces_per_row) return;
  const int t = (int)(gid / pieces_per_row), p = (int)(gid % pieces_per_row);
  const int col = p * 8;
  const int i0 = (col % Dh) >> 1;
  bf16_t* ptr = qkv + (size_t)t * ld + col;
  const u32x4 v = ld16(ptr);
  const float* cs = rope_cs + ((size_t)tok_pos[t] * (Dh >> 1) + i0) * 2;
  const f32x4 c01 = *reinterpret_cast<const f32x4*>(cs);
  const f32x4 c23 = *reinterpret_cast<const f32x4*>(cs + 4);
  const float cc[4] = {c01[0], c01[2], c23[0], c23[2]};
  const float sn[4] = {c01[1], c01[3], c23[1], c23[3]};
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = bf_lo(v[i]), b = bf_hi(v[i]);
    float re, im;
    rope_pair(a, b, cc[i], sn[i], re, im);
    o[i] = pack_bf2(re, im);
  }
  st16(ptr, o);
}

// cache.py:83-92 + 226-235
__global__ __launch_bounds__(256) void kv_write_kernel(bf16_t* ck, bf16_t* cv, int W, const bf16_t* k, const bf16_t* v,
                                                       int ld, int T, int kv_dim, const int32_t* tok_seq,
                                                       const int32_t* tok_pos, const int32_t* q_start, int layout, int Dh) {
  const int pieces = kv_dim >> 3;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)T * pieces) return;
  const int t = (int)(gid / pieces), p = (int)(gid % pieces);
  const int b = tok_seq[t];
  const int i = t - q_start[b];
  const int s = q_start[b + 1] - q_start[b];
  if (i < s - W) return;  // to_cache_mask: only the last W tokens of the chunk are stored
  const size_t off = kv_offset(layout, W, kv_dim, Dh, (size_t)b, tok_pos[t] % W, p * 8);  // (Dh % 8 == 0: a piece stays inside a head)
  st16(ck + off, ld16(k + (size_t)t * ld + p * 8));
  st16(cv + off, ld16(v + (size_t)t * ld + p * 8));
}

// Decode step metadata from the device-resident kv_seqlens (no host round trip), then
// kv_seqlens += 1 (cache.py:193-195 update_seqlens for seqlens = [1]*B).
__global__ void decode_prep_kernel(int64_t* kv_seqlens, int32_t* q_start, int32_t* kv_before, int32_t* tok_seq,
                                   int32_t* tok_pos, int B, uint32_t* engine_ctrl) {
  const int b = threadIdx.x;
  if (b == 0 && engine_ctrl) {
    engine_ctrl[0] += 1;
    engine_ctrl[2] = 0;
    engine_ctrl[5] += 1;  // decode steps started on this workspace (index + 1 into the greedy history ring)
  }
  if (b < B) {
    const int p = (int)kv_seqlens[b];
    kv_before[b] = p;
    tok_pos[b] = p;
    tok_seq[b] = b;
    q_start[b] = b;
    kv_seqlens[b] = p + 1;
  }
  if (b == 0) q_start[B] = B;
}

// Decode step on the rank that owns the embedding: decode_prep + embedding gather in ONE launch (block t = sequence t;
// T == B at decode).  Saves a launch per token; the metadata words are consumed only by later launches.
__global__ __launch_bounds__(256) void decode_prep_embedding_kernel(int64_t* kv_seqlens, int32_t* q_start, int32_t* kv_before,
                                                                    int32_t* tok_seq, int32_t* tok_pos, int B, bf16_t* out,
                                                                    const bf16_t* table, const int64_t* ids, int D, int vocab,
                                                                    uint32_t* engine_ctrl) {
  const int t = blockIdx.x;
  if (threadIdx.x == 0) {
    if (t == 0 && engine_ctrl) {
      engine_ctrl[0] += 1;
      engine_ctrl[2] = 0;
      engine_ctrl[5] += 1;
    }
    const int p = (int)kv_seqlens[t];
    kv_before[t] = p;
    tok_pos[t] = p;
    tok_seq[t] = t;
    q_start[t] = t;
    kv_seqlens[t] = p + 1;
    if (t == 0) q_start[B] = B;
  }
  const long id = checked_id(ids[t], vocab, t, engine_ctrl ? engine_ctrl + 3 : nullptr);
  const bf16_t* src = table + (size_t)id * D;
  bf16_t* dst = out + (size_t)t * D;
  for (int p = threadIdx.x; p < (D >> 3); p += 256) st16(dst + p * 8, ld16(src + p * 8));
}

// Greedy sampling behind the LM head on the launch path (the persistent engine does the same in its own epilogue,
// decode_engine.hip): block b reduces logits row b to (max, FIRST index of the max, sum exp(x - max)) - generate.py:124
// `torch.argmax` and :134-136 `log_softmax(...)[token]`, which at the argmax is -log(sum).
__global__ __launch_bounds__(1024) void greedy_rows_kernel(const float* logits, int ld, int B, int V, int64_t* tok, float* lp,
                                                           int64_t* hist_tok, float* hist_lp, int hist_len,
                                                           const uint32_t* ctrl) {
  __shared__ float s_m[16], s_s[16];
  __shared__ int s_i[16];
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* row = logits + (size_t)b * ld;
  float M = -INFINITY, S = 0.f;
  int I = 0x7fffffff;
  int i0 = threadIdx.x;
  if ((V & 3) == 0 && (reinterpret_cast<size_t>(row) & 15) == 0) {
    // 16 logits per trip from four independent 16-byte loads (round 6: the scalar loop below was one dependent 4-byte load per
    // trip - V / 1024 L2 round trips: 46.6 us per step at Mistral-Nemo's 131072 logits, 14.4 at 32768).  Ascending indices per
    // thread, `>` on the trip's maximum and the first position holding it: the FIRST maximum, as before.
    const int V4 = V >> 2;
    for (int q = threadIdx.x; q < V4; q += 4096) {
      float x[16];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int qk = q + k * 1024;
        const float4 v = *reinterpret_cast<const float4*>(row + (size_t)min(qk, V4 - 1) * 4);  // (unconditional load, masked value)
        const bool in = qk < V4;
        x[4 * k + 0] = in ? v.x : -INFINITY;
        x[4 * k + 1] = in ? v.y : -INFINITY;
        x[4 * k + 2] = in ? v.z : -INFINITY;
        x[4 * k + 3] = in ? v.w : -INFINITY;
      }
      float cm = x[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) cm = fmaxf(cm, x[e]);
      if (cm > M) {
        S *= __expf(M - cm);  // (M = -inf at the start: exp(-inf) = 0 and S = 0)
        M = cm;
        int first = 15;
#pragma unroll
        for (int e = 14; e >= 0; --e) first = (x[e] == cm) ? e : first;
        I = (q + (first >> 2) * 1024) * 4 + (first & 3);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) S += __expf(x[e] - M);
    }
    i0 = V;  // nothing left for the scalar loop
  }
  for (int i = i0; i < V; i += 1024) {  // ascending indices per thread: `>` keeps the first maximum
    const float x = row[i];
    if (x > M) {
      S = S * __expf(M - x) + 1.f;
      M = x;
      I = i;
    } else {
      S += __expf(x - M);
    }
  }
  auto merge = [&](float m2, int i2, float s2) {
    const bool take = m2 > M || (m2 == M && i2 < I);
    const float Mn = take ? m2 : M;
    S = S * (M == Mn ? 1.f : __expf(M - Mn)) + s2 * (m2 == Mn ? 1.f : __expf(m2 - Mn));
    M = Mn;
    I = take ? i2 : I;
  };
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) merge(__shfl_xor(M, o, 64), __shfl_xor(I, o, 64), __shfl_xor(S, o, 64));
  if (lane == 0) {
    s_m[w] = M;
    s_i[w] = I;
    s_s[w] = S;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int j = 1; j < 16; ++j) merge(s_m[j], s_i[j], s_s[j]);
    const float l = -__logf(S);
    tok[b] = I;
    lp[b] = l;
    if (hist_tok && hist_len > 0) {
      const uint32_t step = (ctrl[5] - 1u) % (uint32_t)hist_len;
      hist_tok[(size_t)step * B + b] = I;
      hist_lp[(size_t)step * B + b] = l;
    }
  }
}

__global__ void engine_ctrl_reset_kernel(uint32_t* ctrl) {
  ctrl[0] += 16;  // tags of the failed step (epoch + 1) can never match again
  ctrl[1] = 0;
  ctrl[2] = 0;
  ctrl[6] = 0;
}

// out = bf16(a + b) (transformer_layers.py:168 for the MoE prefill path)
__global__ __launch_bounds__(256) void add_rows_kernel(bf16_t* out, const bf16_t* a, const bf16_t* b, size_t npieces) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npieces) return;
  const u32x4 x = ld16(a + p * 8), y = ld16(b + p * 8);
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack_bf2(bf_lo(x[i]) + bf_lo(y[i]), bf_hi(x[i]) + bf_hi(y[i]));
  st16(out + p * 8, o);
}

// moe.py:25-27.  One block per token, one wave per expert: logit_e = bf16(x . Wg[e]) ; top-k ; softmax over the k ;
// bf16.  (A single wave per token serialises E dot products behind one another: 30 us per decode layer.)
constexpr int MOE_MAX_E = 16;
__global__ __launch_bounds__(1024) void moe_router_kernel(int32_t* sel_idx, float* sel_w, const bf16_t* x, int ldx, int D,
                                                          const bf16_t* gate, int E, int top_k, const bf16_t* norm_w,
                                                          float eps) {
  __shared__ float s_logit[MOE_MAX_E];
  const int t = blockIdx.x, lane = threadIdx.x & 63, e = threadIdx.x >> 6;  // wave e <-> expert e
  const bf16_t* xr = x + (size_t)t * ldx;
  const bf16_t* gr = gate + (size_t)e * D;
  const int np = D >> 3;
  float ss = 0.f;
  if (norm_w) {  // every wave needs the same 1/rms: recomputing it (8 KB from L2) is cheaper than a block reduction
    for (int p = lane; p < np; p += 64) {
      const u32x4 v = ld16(xr + p * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = bf_lo(v[i]), b = bf_hi(v[i]);
        ss = fmaf(a, a, ss);
        ss = fmaf(b, b, ss);
      }
    }
    ss = wave_sum(ss);
  }
  const float inv = norm_w ? 1.0f / sqrtf(ss / (float)D + eps) : 1.f;
  float acc = 0.f;
  for (int p = lane; p < np; p += 64) {
    const u32x4 v = ld16(xr + p * 8);
    const u32x4 g = ld16(gr + p * 8);
    const u32x4 wv = norm_w ? ld16(norm_w + p * 8) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x0 = bf_lo(v[i]), x1 = bf_hi(v[i]);
      if (norm_w) {
        x0 = bf_round(bf_round(x0 * inv) * bf_lo(wv[i]));
        x1 = bf_round(bf_round(x1 * inv) * bf_hi(wv[i]));
      }
      acc = fmaf(bf_lo(g[i]), x0, acc);
      acc = fmaf(bf_hi(g[i]), x1, acc);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) s_logit[e] = bf_round(acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    float tw[4];
    int ti[4];
    unsigned taken = 0;
    for (int k = 0; k < top_k; ++k) {
      int best = -1;
      float bv = -INFINITY;
      for (int j = 0; j < E; ++j)
        if (!((taken >> j) & 1u) && (best < 0 || s_logit[j] > bv)) {  // ties: lowest expert id
          best = j;
          bv = s_logit[j];
        }
      taken |= 1u << best;
      ti[k] = best;
      tw[k] = bv;
    }
    float den = 0.f, ex[4];
    for (int k = 0; k < top_k; ++k) {
      ex[k] = expf(tw[k] - tw[0]);
      den += ex[k];
    }
    for (int k = 0; k < top_k; ++k) {
      sel_idx[t * top_k + k] = ti[k];
      sel_w[t * top_k + k] = bf_round(ex[k] / den);
    }
  }
}

// Sort the T*top_k (token, slot) pairs by expert for the token-grouped prefill GEMMs (moe.py:29-31 `torch.where` per
// expert, which costs the reference one host sync per expert per layer).  Single block: LDS histogram, exclusive
// scan, LDS-cursor scatter.  Row order inside an expert is arbitrary (it only decides which MFMA tile a row lands
// in, never its value); the bf16 accumulation ORDER of moe.py is reproduced later by moe_combine_kernel.
__global__ __launch_bounds__(1024) void moe_lists_kernel(const int32_t* sel_idx, int T, int E, int top_k, int32_t* tok_of,
                                                         int32_t* row_of, int32_t* tile_tab, int32_t* n_tiles, int tile_rows) {
  __shared__ int cnt[MOE_MAX_E];
  __shared__ int off[MOE_MAX_E + 1];
  __shared__ int cur[MOE_MAX_E];
  const int tid = threadIdx.x, n = T * top_k;
  if (tid < MOE_MAX_E) cnt[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) atomicAdd(&cnt[sel_idx[i]], 1);
  __syncthreads();
  if (tid == 0) {
    int o = 0, nt = 0;
    for (int e = 0; e < E; ++e) {
      off[e] = o;
      cur[e] = o;
      for (int r = 0; r < cnt[e]; r += tile_rows) {  // m-tiles of this expert
        tile_tab[nt * 4 + 0] = e;
        tile_tab[nt * 4 + 1] = o + r;
        tile_tab[nt * 4 + 2] = min(tile_rows, cnt[e] - r);
        tile_tab[nt * 4 + 3] = 0;
        ++nt;
      }
      o += cnt[e];
    }
    off[E] = o;
    *n_tiles = nt;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int r = atomicAdd(&cur[sel_idx[i]], 1);
    tok_of[r] = i / top_k;
    row_of[i] = r;
  }
}

// moe.py:28-32 accumulation + transformer_layers.py:168: per token, experts in ascending id, bf16 running sum from 0.
__global__ __launch_bounds__(256) void moe_combine_kernel(bf16_t* out, const bf16_t* h, const bf16_t* y, const int32_t* sel_idx,
                                                          const float* sel_w, const int32_t* row_of, int D, int top_k) {
  const int t = blockIdx.x;
  int eid[4], row[4];
  float w[4];
  for (int k = 0; k < top_k; ++k) {
    eid[k] = sel_idx[t * top_k + k];
    row[k] = row_of[t * top_k + k];
    w[k] = sel_w[t * top_k + k];
  }
  for (int i = 0; i < top_k; ++i)
    for (int j = i + 1; j < top_k; ++j)
      if (eid[j] < eid[i]) {
        int te = eid[i]; eid[i] = eid[j]; eid[j] = te;
        te = row[i]; row[i] = row[j]; row[j] = te;
        const float tw = w[i]; w[i] = w[j]; w[j] = tw;
      }
  for (int p = threadIdx.x; p < (D >> 3); p += 256) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = 0.f;
    for (int k = 0; k < top_k; ++k) {
      const u32x4 v = ld16(y + (size_t)row[k] * D + p * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r[2 * i] = bf_round(r[2 * i] + bf_round(w[k] * bf_lo(v[i])));
        r[2 * i + 1] = bf_round(r[2 * i + 1] + bf_round(w[k] * bf_hi(v[i])));
      }
    }
    const u32x4 hv = ld16(h + (size_t)t * D + p * 8);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf2(bf_lo(hv[i]) + r[2 * i], bf_hi(hv[i]) + r[2 * i + 1]);
    st16(out + (size_t)t * D + p * 8, o);
  }
}

// nn.GELU() (exact erf form) on a bf16 tensor: evaluated in fp32, rounded once (vision_encoder.py:112-116).
__global__ __launch_bounds__(256) void gelu_kernel(bf16_t* x, int ldx, int N) {
  bf16_t* row = x + (size_t)blockIdx.x * ldx;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float v = bf_to_f(row[i]);
    row[i] = f_to_bf(0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)));
  }
}

// logprob[m] = x_t - (M + log sum_j s_j exp(m_j - M)) from the per-tile partials of GEMM_LOGPROB: one wave per row.
__global__ __launch_bounds__(64) void logprob_finalize_kernel(float* out, const float2* partial, const float* tgt, int n_tiles) {
  const int m = blockIdx.x, lane = threadIdx.x;
  const float2* p = partial + (size_t)m * n_tiles;
  float M = -INFINITY;
  for (int j = lane; j < n_tiles; j += 64) M = fmaxf(M, p[j].x);
  M = wave_max(M);
  float S = 0.f;
  for (int j = lane; j < n_tiles; j += 64) S += p[j].y * __expf(p[j].x - M);
  S = wave_sum(S);
  if (lane == 0) out[m] = tgt[m] - (M + logf(S));
}

// Same result from a full fp32 logits row (small-M path): one block per row.
__global__ __launch_bounds__(256) void logprob_rows_kernel(float* out, const float* logits, int ld, const int32_t* target, int V) {
  __shared__ float red[8];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* x = logits + (size_t)m * ld;
  float M = -INFINITY;
  for (int j = tid; j < V; j += 256) M = fmaxf(M, x[j]);
  M = wave_max(M);
  if (lane == 0) red[wid] = M;
  __syncthreads();
  M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float S = 0.f;
  for (int j = tid; j < V; j += 256) S += __expf(x[j] - M);
  S = wave_sum(S);
  if (lane == 0) red[4 + wid] = S;
  __syncthreads();
  if (tid == 0) {
    const int t = target[m];
    out[m] = (t >= 0 && t < V ? x[t] : 0.f) - (M + logf(red[4] + red[5] + red[6] + red[7]));
  }
}

}  // namespace

hipError_t launch_logprob_finalize(float* out, const float2* partial, const float* tgt, int M, int n_tiles, hipStream_t s) {
  hipLaunchKernelGGL(logprob_finalize_kernel, dim3(M), dim3(64), 0, s, out, partial, tgt, n_tiles);
  return hipGetLastError();
}
hipError_t launch_logprob_rows(float* out, const float* logits, int ld, const int32_t* target, int M, int V, hipStream_t s) {
  hipLaunchKernelGGL(logprob_rows_kernel, dim3(M), dim3(256), 0, s, out, logits, ld, target, V);
  return hipGetLastError();
}

hipError_t launch_gelu(void* x, int ldx, int T, int N, hipStream_t s) {
  hipLaunchKernelGGL(gelu_kernel, dim3(T), dim3(256), 0, s, (bf16_t*)x, ldx, N);
  return hipGetLastError();
}

hipError_t launch_embedding(void* out, const void* table, const int64_t* ids, int T, int D, int vocab, uint32_t* bad_id,
                            hipStream_t s) {
  hipLaunchKernelGGL(embedding_kernel, dim3(T), dim3(256), 0, s, (bf16_t*)out, (const bf16_t*)table, ids, D, vocab, bad_id);
  return hipGetLastError();
}
hipError_t launch_rmsnorm(void* out, const void* x, const void* w, int T, int D, float eps, hipStream_t s) {
  const dim3 grid((T + 3) / 4), block(256);
  bf16_t* o = (bf16_t*)out;
  const bf16_t *xx = (const bf16_t*)x, *ww = (const bf16_t*)w;
  if (D % 8 || D > 16384) return hipErrorInvalidValue;
  if (D <= 4096) hipLaunchKernelGGL((rmsnorm_kernel<8>), grid, block, 0, s, o, xx, ww, T, D, eps);
  else if (D <= 6144) hipLaunchKernelGGL((rmsnorm_kernel<12>), grid, block, 0, s, o, xx, ww, T, D, eps);
  else if (D <= 8192) hipLaunchKernelGGL((rmsnorm_kernel<16>), grid, block, 0, s, o, xx, ww, T, D, eps);
  else hipLaunchKernelGGL((rmsnorm_kernel<32>), grid, block, 0, s, o, xx, ww, T, D, eps);
  return hipGetLastError();
}
hipError_t launch_rope(void* qkv, int ld, int T, int H, int Hkv, int Dh, const float* rope_cs, const int32_t* tok_pos,
                       hipStream_t s) {
  const int cols = (H + Hkv) * Dh;
  const long n = (long)T * (cols >> 3);
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (bf16_t*)qkv, ld, T, cols, Dh,
                     rope_cs, tok_pos);
  return hipGetLastError();
}
hipError_t launch_kv_write(void* ck, void* cv, int W, const void* k, const void* v, int ld, int T, int kv_dim,
                           const int32_t* tok_seq, const int32_t* tok_pos, const int32_t* q_start, int kv_layout, int Dh, hipStream_t s) {
  if (Dh <= 0 || Dh % 8 || kv_dim % Dh) return hipErrorInvalidValue;
  const long n = (long)T * (kv_dim >> 3);
  hipLaunchKernelGGL(kv_write_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (bf16_t*)ck, (bf16_t*)cv, W,
                     (const bf16_t*)k, (const bf16_t*)v, ld, T, kv_dim, tok_seq, tok_pos, q_start, kv_layout, Dh);
  return hipGetLastError();
}
hipError_t launch_decode_prep(int64_t* kv_seqlens, int32_t* q_start, int32_t* kv_before, int32_t* tok_seq,
                              int32_t* tok_pos, int B, uint32_t* engine_ctrl, hipStream_t s) {
  if (B > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(decode_prep_kernel, dim3(1), dim3(((B + 63) / 64) * 64), 0, s, kv_seqlens, q_start, kv_before,
                     tok_seq, tok_pos, B, engine_ctrl);
  return hipGetLastError();
}
hipError_t launch_decode_prep_embedding(int64_t* kv_seqlens, int32_t* q_start, int32_t* kv_before, int32_t* tok_seq,
                                        int32_t* tok_pos, int B, void* out, const void* table, const int64_t* ids, int D,
                                        int vocab, uint32_t* engine_ctrl, hipStream_t s) {
  hipLaunchKernelGGL(decode_prep_embedding_kernel, dim3(B), dim3(256), 0, s, kv_seqlens, q_start, kv_before, tok_seq, tok_pos, B,
                     (bf16_t*)out, (const bf16_t*)table, ids, D, vocab, engine_ctrl);
  return hipGetLastError();
}
hipError_t launch_greedy_rows(const float* logits, int ld, int B, int V, int64_t* tok, float* lp, int64_t* hist_tok,
                              float* hist_lp, int hist_len, const uint32_t* ctrl, hipStream_t s) {
  hipLaunchKernelGGL(greedy_rows_kernel, dim3(B), dim3(1024), 0, s, logits, ld, B, V, tok, lp, hist_tok, hist_lp, hist_len, ctrl);
  return hipGetLastError();
}
hipError_t launch_engine_ctrl_reset(uint32_t* ctrl, hipStream_t s) {
  hipLaunchKernelGGL(engine_ctrl_reset_kernel, dim3(1), dim3(1), 0, s, ctrl);
  return hipGetLastError();
}
hipError_t launch_add_rows(void* out, const void* a, const void* b, size_t n, hipStream_t s) {
  const size_t np = n >> 3;
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, (bf16_t*)out,
                     (const bf16_t*)a, (const bf16_t*)b, np);
  return hipGetLastError();
}
hipError_t launch_moe_router(int32_t* sel_idx, float* sel_w, const void* x, int ldx, int T, int D, const void* gate,
                             int E, int top_k, const void* norm_w, float eps, hipStream_t s) {
  if (E > MOE_MAX_E || top_k > 4 || top_k > E) return hipErrorInvalidValue;
  hipLaunchKernelGGL(moe_router_kernel, dim3(T), dim3(64 * E), 0, s, sel_idx, sel_w, (const bf16_t*)x, ldx, D,
                     (const bf16_t*)gate, E, top_k, (const bf16_t*)norm_w, eps);
  return hipGetLastError();
}
hipError_t launch_moe_lists(const int32_t* sel_idx, int T, int E, int top_k, int32_t* tok_of, int32_t* row_of,
                            int32_t* tile_tab, int32_t* n_tiles, int tile_rows, hipStream_t s) {
  hipLaunchKernelGGL(moe_lists_kernel, dim3(1), dim3(1024), 0, s, sel_idx, T, E, top_k, tok_of, row_of, tile_tab, n_tiles,
                     tile_rows);
  return hipGetLastError();
}
hipError_t launch_moe_combine(void* out, const void* h, const void* y, const int32_t* sel_idx, const float* sel_w,
                              const int32_t* row_of, int T, int D, int top_k, hipStream_t s) {
  hipLaunchKernelGGL(moe_combine_kernel, dim3(T), dim3(256), 0, s, (bf16_t*)out, (const bf16_t*)h, (const bf16_t*)y,
                     sel_idx, sel_w, row_of, D, top_k);
  return hipGetLastError();
}
