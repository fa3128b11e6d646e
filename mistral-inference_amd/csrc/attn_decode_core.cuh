// Arithmetic of the decode-branch attention, shared by the stand-alone kernels (attn_decode.hip) and the persistent
// decode engine (decode_engine.hip) so that both produce bit-identical results: every product/sum that could be
// contracted either way by the compiler is written as an explicit fmaf / __fmul_rn.
//
// Decomposition (transformer_layers.py:77-89 under the mask of cache.py:249-254): ring slots are cut into splits; inside
// a split, slot s_begin + 4 * p + g belongs to lane group g (16 lanes, 8 head dims each) of "virtual wave" p % 4, which
// visits its slots in ascending order with an online softmax; lane groups -> wave (butterfly), 4 waves -> split partial
// (fp32 m, l, acc[128] per query head), splits -> output in ascending split order.
#pragma once
#include "common.cuh"

namespace attn_core {

constexpr int DH = 128;
constexpr float LOG2E = 1.4426950408889634f;
// Online softmax with a LAGGING reference maximum (cdna_hip_programming.md T13): a lane group rescales its sums only
// when a score exceeds the reference by more than 2^RESCALE_T, so the common step has no rescale multiplies; the
// triple (m, l, acc) stays self-consistent (p <= 2^RESCALE_T, far inside fp32 range) and every merge works on it as on
// an exact maximum.
constexpr float RESCALE_T = 8.0f;

// 2^x as ONE v_exp_f32 (arguments are <= RESCALE_T; results below the normal range flush to zero, which is what a softmax
// weight of 2^-127 is anyway).  The library routine spends ~5 more instructions on denormal range fix-ups.
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int R>
struct State {
  float m[R], l[R], acc[R][8];
};

template <int R>
__device__ __forceinline__ void init_state(State<R>& st) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    st.m[r] = -1e30f;
    st.l[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) st.acc[r][i] = 0.f;
  }
}

// q row pieces of this lane (dims dl*8 .. +8 of each of the R query heads), pre-scaled by Dh^-1/2 * log2(e)
template <int R>
__device__ __forceinline__ void load_q(float (&qf)[R][8], const u32x4 (&qraw)[R]) {
  const float sc = rsqrtf((float)DH) * LOG2E;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qf[r][2 * i] = __fmul_rn(bf_lo(qraw[r][i]), sc);
      qf[r][2 * i + 1] = __fmul_rn(bf_hi(qraw[r][i]), sc);
    }
}

// One ring slot (K row piece + V row piece of this lane's 8 dims) folded into the lane group's online softmax.
template <int R>
__device__ __forceinline__ void reduce_slot(State<R>& st, const float (&qf)[R][8], const u32x4 kraw, const u32x4 vraw,
                                            bool valid) {
  float kf[8], vf[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    kf[2 * i] = bf_lo(kraw[i]);
    kf[2 * i + 1] = bf_hi(kraw[i]);
    vf[2 * i] = bf_lo(vraw[i]);
    vf[2 * i + 1] = bf_hi(vraw[i]);
  }
  float d[R];
  bool grow[R], any_grow = false;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t = fmaf(qf[r][i], kf[i], t);
    d[r] = row16_sum(t);
    grow[r] = valid && (d[r] > st.m[r] + RESCALE_T);
    any_grow |= grow[r];
  }
  // ONE wave-uniform branch per slot for all R heads (a ballot + branch per head costs a scalar round trip each);
  // rare after a lane group's first slots.  Heads / lanes that keep their reference multiply by exactly 1.
  if (__builtin_amdgcn_ballot_w64(any_grow) != 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float mn = grow[r] ? d[r] : st.m[r];
      const float alpha = ex2(st.m[r] - mn);
      st.l[r] *= alpha;
#pragma unroll
      for (int i = 0; i < 8; ++i) st.acc[r][i] *= alpha;
      st.m[r] = mn;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float p = valid ? ex2(d[r] - st.m[r]) : 0.f;
    st.l[r] += p;
#pragma unroll
    for (int i = 0; i < 8; ++i) st.acc[r][i] = fmaf(p, vf[i], st.acc[r][i]);
  }
}

// Value of lane (lane ^ 16) / (lane ^ 32): gfx950 permlane swaps (VALU rate) instead of ds_bpermute round trips.
// v_permlane16_swap(v, v) returns {rows (0,0,2,2), rows (1,1,3,3)}; v_permlane32_swap(v, v) {halves (lo,lo), (hi,hi)}.
template <int OFF>
__device__ __forceinline__ float lane_xor(float v, bool upper) {
  static_assert(OFF == 16 || OFF == 32, "cross-row exchange");
  const uint32_t u = __float_as_uint(v);
  if (OFF == 16) {
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(upper ? a[0] : a[1]);
  }
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(upper ? a[0] : a[1]);
}

template <int R, int OFF>
__device__ __forceinline__ void merge_from(State<R>& s, int lane) {
  const bool upper = (lane & OFF) != 0;  // this lane sits in the odd row / upper half: its partner is the other one
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float mo = lane_xor<OFF>(s.m[r], upper);
    const float lo = lane_xor<OFF>(s.l[r], upper);
    const float M = fmaxf(s.m[r], mo);
    const float a1 = ex2(s.m[r] - M), a2 = ex2(mo - M);
    s.l[r] = fmaf(s.l[r], a1, __fmul_rn(lo, a2));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float ao = lane_xor<OFF>(s.acc[r][i], upper);
      s.acc[r][i] = fmaf(s.acc[r][i], a1, __fmul_rn(ao, a2));
    }
    s.m[r] = M;
  }
}

// LDS image of one virtual wave's merged state: sm_m[4][R], sm_l[4][R], sm_acc[4][R][DH]
// (FP: float* into __shared__ arrays, or an explicit address_space(3) pointer)
template <int R, class FP>
__device__ __forceinline__ void wave_state_to_lds(State<R>& st, int vw, int lane, FP sm_m, FP sm_l, FP sm_acc) {
  merge_from<R, 16>(st, lane);
  merge_from<R, 32>(st, lane);
  const int g = lane >> 4, dl = lane & 15;
  if (g == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (dl == 0) {
        sm_m[vw * R + r] = st.m[r];
        sm_l[vw * R + r] = st.l[r];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) sm_acc[(vw * R + r) * DH + dl * 8 + i] = st.acc[r][i];
    }
  }
}

// The same for RP of the RT query heads of a kv head (heads r_off .. r_off + RP): a block that serves its heads in several
// passes over the same K/V slots (fewer live registers per pass) leaves the LDS image of the one-pass form.  Heads do not
// interact in reduce_slot (a head that keeps its reference maximum is multiplied by exactly 1), so the results are the same bits.
template <int RP, int RT, class FP>
__device__ __forceinline__ void wave_state_to_lds_heads(State<RP>& st, int vw, int lane, int r_off, FP sm_m, FP sm_l, FP sm_acc) {
  merge_from<RP, 16>(st, lane);
  merge_from<RP, 32>(st, lane);
  const int g = lane >> 4, dl = lane & 15;
  if (g == 0) {
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      if (dl == 0) {
        sm_m[vw * RT + r_off + r] = st.m[r];
        sm_l[vw * RT + r_off + r] = st.l[r];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) sm_acc[(vw * RT + r_off + r) * DH + dl * 8 + i] = st.acc[r][i];
    }
  }
}

// 4 virtual waves -> the split's partial for flat index idx = r * DH + d.
template <int R, class FP>
__device__ __forceinline__ void split_partial(int idx, FP sm_m, FP sm_l, FP sm_acc, float& A, float& M, float& L) {
  const int r = idx / DH, d = idx % DH;
  M = fmaxf(fmaxf(sm_m[0 * R + r], sm_m[1 * R + r]), fmaxf(sm_m[2 * R + r], sm_m[3 * R + r]));
  L = 0.f;
  A = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float e = ex2(sm_m[w * R + r] - M);
    L = fmaf(sm_l[w * R + r], e, L);
    A = fmaf(sm_acc[(w * R + r) * DH + d], e, A);
  }
}

// splits -> one output element.  m/l/v: per-split values (entries >= n are ignored).
template <int NS>
__device__ __forceinline__ float combine_splits(const float (&m)[NS], const float (&l)[NS], const float (&v)[NS], int n) {
  float M = -1e30f;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) M = fmaxf(M, (sp < n) ? m[sp] : -1e30f);
  float L = 0.f, A = 0.f;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) {
    const float e = ex2(((sp < n) ? m[sp] : -1e30f) - M);
    L = fmaf((sp < n) ? l[sp] : 0.f, e, L);
    A = fmaf((sp < n) ? v[sp] : 0.f, e, A);
  }
  return A / L;
}

// The same merge with the per-split values behind accessors (LDS-staged in the engine): identical operations in the
// identical order - entries >= n of the array form contribute exactly nothing.
template <class FM, class FL, class FV>
__device__ __forceinline__ float combine_stream(int n, FM m, FL l, FV v) {
  float M = -1e30f;
  for (int sp = 0; sp < n; ++sp) M = fmaxf(M, m(sp));
  float L = 0.f, A = 0.f;
  for (int sp = 0; sp < n; ++sp) {
    const float e = ex2(m(sp) - M);
    L = fmaf(l(sp), e, L);
    A = fmaf(v(sp), e, A);
  }
  return A / L;
}

// Split geometry shared by both paths: slots per split (multiple of 16) for a ring of W slots cut into n_splits.
__host__ __device__ inline int split_chunk(int W, int n_splits) {
  int chunk = (W + n_splits - 1) / n_splits;
  return (chunk + 15) & ~15;
}

}  // namespace attn_core
