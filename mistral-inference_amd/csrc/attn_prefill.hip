// Prefill-branch attention (first prefill, later chunks, and the cache=None call) on MFMA.
//
// Replaces transformer_layers.py:74-76,84-89 at prefill: CacheView.interleave_kv (cache.py:94-117,
// unrotate + per-sequence torch.cat), repeat_kv (:16-19) and xformers FMHA under
// BlockDiagonalCausalMask.make_local_attention / BlockDiagonalMask.make_local_attention_from_bottomright
// (cache.py:238-248).  All three reduce to: key position kp of the same sequence is visible to query
// position qp iff qp - W < kp <= qp.  Keys older than this forward (kp < p_b) are
// read straight from the ring at slot kp % W, new keys from the post-RoPE activation rows - nothing is
// concatenated or replicated.
//
// Block = (128-query tile, q head, sequence), 4 waves x 32 query rows.  Per 64-key tile:
//   S^T = K . Q^T   (mfma 32x32x16; swapped so that a lane owns ONE query column: the softmax row
//                    reductions are in-lane plus one exchange with lane^32 - guide T12)
//   O^T += V^T . P^T with P^T taken directly from the S^T accumulator registers: the contraction index
//                    <-> key mapping of the B operand is permuted to match the C layout, and the A operand
//                    (V^T, staged transposed in LDS) is read with the same permutation.
// K tile: row-major in LDS, 16-byte slots XOR-swizzled by (key & 15) -> conflict-free ds_read_b128.
// V tile: transposed on the way in (4 keys x 8 d per thread, register transpose, ds_write_b64), row
// stride 136 B -> conflict-free ds_read_b64.
#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int DH = 128;
constexpr int KT = 64;         // keys per tile
constexpr int VT_STRIDE = 136; // bytes per d-row of the transposed V tile
constexpr int KS_BYTES = KT * DH * 2;
constexpr int VT_BYTES = DH * VT_STRIDE;
constexpr float LOG2E = 1.4426950408889634f;

__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(AttnPrefillArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[KS_BYTES + VT_BYTES];
  char* Ks = smem;
  char* Vt = smem + KS_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int hf = lane >> 5, ql = lane & 31;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int R = a.H / a.Hkv, kvh = h / R;
  const int nq_cols = a.H * DH, kv_dim = a.Hkv * DH;

  int row0, s_b, p_b;
  if (a.causal) {
    row0 = a.q_start[b];
    s_b = a.q_start[b + 1] - row0;
    p_b = a.kv_before[b];
  } else {
    row0 = 0;
    s_b = a.max_q_len;
    p_b = 0;
  }
  if (qt * 128 >= s_b) return;
  const int W = a.W;
  const int n_old = min(p_b, W);

  // key-position range this block needs
  const int qp_blk_lo = p_b + qt * 128;
  const int qp_blk_hi = p_b + min(qt * 128 + 127, s_b - 1);
  const int kp_lo = a.causal ? max(p_b - n_old, qp_blk_lo - W + 1) : 0;
  const int kp_hi = a.causal ? qp_blk_hi : s_b - 1;
  const int n_tiles = (kp_hi - kp_lo + KT) / KT;

  // this wave's queries
  const int qi = qt * 128 + wid * 32 + ql;            // index inside the sequence
  const int qi_c = min(qi, s_b - 1);
  const int qp = p_b + qi_c;
  const int qp_w_lo = p_b + qt * 128 + wid * 32;       // wave's lowest / highest query position
  const int qp_w_hi = p_b + min(qt * 128 + wid * 32 + 31, s_b - 1);
  const bool wave_active = (qt * 128 + wid * 32) < s_b;

  bf16x8 qf[8];
  {
    const bf16_t* qrow = a.qkv + (size_t)(row0 + qi_c) * a.ld + (size_t)h * DH + hf * 8;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, ld16(qrow + kk * 16));
  }

  f32x16 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sc = rsqrtf((float)DH) * LOG2E;

  // ---- staging assignment
  const int k_slot = tid & 15, k_key0 = tid >> 4;   // K: keys k_key0 + 16 j, 16-byte slot k_slot
  const int v_kq = tid & 15, v_ds = tid >> 4;       // V: keys 4 v_kq + j, d slice v_ds*8..+8
  u32x4 rk[4], rv[4];
  // Source row of key position kp: ring slot (kp % W) for keys older than this forward, activation row otherwise.
  // Both candidate addresses are formed and SELECTED (v_cndmask), and the load itself is unconditional from a
  // clamped position: a `cond ? load : 0` makes hipcc branch around every load and wait vmcnt(0) after each one
  // (cdna_hip_programming.md ".s-level traps" (c)), which serialised this staging to one load in flight.
  // Keys past kp_hi are masked in the softmax (their P is exactly 0), so the clamped duplicates are harmless.
  const bf16_t* ring_k0 = a.cache_k ? a.cache_k + ((size_t)b * W) * kv_dim + (size_t)kvh * DH : a.qkv;
  const bf16_t* ring_v0 = a.cache_v ? a.cache_v + ((size_t)b * W) * kv_dim + (size_t)kvh * DH : a.qkv;
  const bf16_t* act_k0 = a.qkv + nq_cols + (size_t)kvh * DH;
  const bf16_t* act_v0 = act_k0 + kv_dim;
  auto key_ptr = [&](int kp, bool is_v) -> const bf16_t* {
    const int slot = (kp < p_b) ? (kp % W) : 0;                  // only cached keys live in the ring
    const int arow = (kp < p_b) ? 0 : (row0 + kp - p_b);
    const bf16_t* r = (is_v ? ring_v0 : ring_k0) + (size_t)slot * kv_dim;
    const bf16_t* x = (is_v ? act_v0 : act_k0) + (size_t)arow * a.ld;
    return (kp < p_b) ? r : x;
  };
  auto gload = [&](int it) {
    const int t_lo = kp_lo + it * KT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kpk = min(t_lo + k_key0 + 16 * j, kp_hi);
      rk[j] = ld16(key_ptr(kpk, false) + k_slot * 8);
      const int kpv = min(t_lo + v_kq * 4 + j, kp_hi);
      rv[j] = ld16(key_ptr(kpv, true) + v_ds * 8);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = k_key0 + 16 * j;
      st16(Ks + key * (DH * 2) + ((k_slot ^ (key & 15)) << 4), rk[j]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // d = v_ds*8 + 2c (low halves) and 2c+1 (high halves); 4 keys packed per row
      u32x2 lo, hi;
      lo[0] = (rv[0][c] & 0xffffu) | (rv[1][c] << 16);
      lo[1] = (rv[2][c] & 0xffffu) | (rv[3][c] << 16);
      hi[0] = (rv[0][c] >> 16) | (rv[1][c] & 0xffff0000u);
      hi[1] = (rv[2][c] >> 16) | (rv[3][c] & 0xffff0000u);
      *reinterpret_cast<u32x2*>(Vt + (v_ds * 8 + 2 * c) * VT_STRIDE + v_kq * 8) = lo;
      *reinterpret_cast<u32x2*>(Vt + (v_ds * 8 + 2 * c + 1) * VT_STRIDE + v_kq * 8) = hi;
    }
  };

  // No register prefetch of the next tile: it would push the kernel past 256 registers, i.e. to one wave per SIMD.
  // Two resident blocks per CU (2 waves per SIMD) hide each other's staging latency and barriers instead.
  for (int it = 0; it < n_tiles; ++it) {
    gload(it);
    lstore();
    __syncthreads();

    const int t_lo = kp_lo + it * KT;
    const int t_hi = min(t_lo + KT - 1, kp_hi);
    const bool skip = !wave_active || (a.causal && (t_lo > qp_w_hi || t_hi <= qp_w_lo - W));
    if (!skip) {
      // ---- S^T = K . Q^T
      f32x16 st[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[mb][r] = 0.f;
        const int key = mb * 32 + ql;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int slot = kk * 2 + hf;
          const bf16x8 kf = __builtin_bit_cast(
              bf16x8, *reinterpret_cast<const u32x4*>(Ks + key * (DH * 2) + ((slot ^ (key & 15)) << 4)));
          st[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[mb], 0, 0, 0);
        }
      }
      // ---- mask, online softmax (this lane: query ql, keys of its half)
      const bool full = a.causal ? (t_hi <= qp_w_lo && t_lo > qp_w_hi - W && t_lo + KT - 1 <= kp_hi) : (t_lo + KT - 1 <= kp_hi);
      float mx = -INFINITY;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float s = st[mb][r] * sc;
          if (!full) {
            const int kp = t_lo + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const bool vis = (kp <= kp_hi) && (!a.causal || (kp <= qp && kp > qp - W));
            s = vis ? s : -INFINITY;
          }
          st[mb][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2(m_run - m_new);
      const bool moved = m_new != m_run;
      m_run = m_new;
      float psum = 0.f;
      uint32_t pb[2][2][4];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = fast_exp2(st[mb][r] - m_new), p1 = fast_exp2(st[mb][r + 1] - m_new);
          psum += p0 + p1;
          pb[mb][r >> 3][(r & 7) >> 1] = cvt_pk_bf16(p0, p1);
        }
      l_run = l_run * alpha + psum;
      if (__any(moved)) {  // exact: when no lane's running max moved, every alpha is 1 and the rescale is a no-op
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
      }
      // ---- O^T += V^T . P^T
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          u32x4 pw = {pb[mb][kb][0], pb[mb][kb][1], pb[mb][kb][2], pb[mb][kb][3]};
          const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
          const int kbase = mb * 32 + kb * 16 + 4 * hf;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const char* vrow = Vt + (dt * 32 + ql) * VT_STRIDE + kbase * 2;
            const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow);
            const u32x2 v1 = *reinterpret_cast<const u32x2*>(vrow + 16);
            u32x4 vw = {v0[0], v0[1], v1[0], v1[1]};
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), pf, acc[dt], 0, 0, 0);
          }
        }
    }
    __syncthreads();
  }

  // ---- epilogue: O^T[d][q] / l  ->  out[q][h*128 + d]
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (wave_active && qi < s_b) {
    const float inv = 1.0f / l_tot;
    bf16_t* orow = reinterpret_cast<bf16_t*>(a.out) + (size_t)(row0 + qi) * nq_cols + (size_t)h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * hf;
        u32x2 o;
        o[0] = pack_bf2(acc[dt][4 * g] * inv, acc[dt][4 * g + 1] * inv);
        o[1] = pack_bf2(acc[dt][4 * g + 2] * inv, acc[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(orow + d) = o;
      }
  }
}

}  // namespace

hipError_t launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t s) {
  if (a.Dh != DH || a.H % a.Hkv != 0) return hipErrorInvalidValue;
  dim3 grid((a.max_q_len + 127) / 128, a.H, a.causal ? a.B : 1), block(256);
  hipLaunchKernelGGL(attn_prefill_kernel, grid, block, 0, s, a);
  return hipGetLastError();
}
