// Decode-branch attention: one query per sequence against its rotating K/V ring.
//
// Replaces transformer_layers.py:77-89 at decode: `repeat_kv` (the reference materialises R copies of
// the whole padded ring per token, :16-19,84) plus xformers FMHA under
// BlockDiagonalCausalWithOffsetPaddedKeysMask (cache.py:249-254): sequence b sees ring slots
// [0, min(pos_b + 1, W)) of its own row; slot order is irrelevant (RoPE applied before caching).
//
// HBM-bound (K and V of the ring are read exactly once): grid = (splits, kv_head, sequence); each
// block streams a contiguous slot range for ONE kv head and serves all R = H/Hkv query heads from
// the same registers.  A wave-load covers 4 slots x 256 B; 16 lanes share a slot, scores are reduced
// with DPP row rotations; online softmax is kept per 16-lane group (no cross-lane sync in the loop)
// and merged once at the end: lanes -> waves (LDS) -> splits (global fp32 partials, merged by a second, tiny
// launch: the kernel boundary is the cross-XCD visibility point.  An in-kernel "last block combines" variant with an
// agent-scope release/acquire ticket measured slower than that boundary and was dropped).
#include <cstdlib>

#include "attn_decode_core.cuh"
#include "kernels.h"

namespace {

using namespace attn_core;

// SMALL (round 5, batches of >= 3 sequences - mistral-demo decodes three): the kernel sits at 170 VGPRs = 2 blocks per CU, so
// 3 x 256 blocks ran as 1.5 rounds of the grid (20.7 us per layer against 9.0 at batch 1).  With half-size load sets (UK = 2:
// 8 instead of 16 KiB in flight per wave) it fits 3 blocks per CU without spilling and the batch is ONE round.  A lane group
// still visits its slots in ascending order: bit-identical results.
template <int R, bool SMALL>
__global__ __launch_bounds__(256, SMALL ? 3 : 1) void attn_decode_kernel(AttnDecodeArgs a) {
  __shared__ float sm_m[4 * R];
  __shared__ float sm_l[4 * R];
  __shared__ float sm_acc[4 * R * DH];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar control flow
  const int g = lane >> 4, dl = lane & 15;
  // Workgroup i runs on XCD i % 8: consecutive ids cycle through the kv heads, so (Hkv = 8) every split of a kv head
  // - and the combine block that merges them - sits on ONE XCD and the fp32 partials are re-read from that XCD's L2.
  const int kvh = blockIdx.x % a.Hkv, split = blockIdx.x / a.Hkv, b = blockIdx.y;
  const int pos = a.tok_pos[b];
  const int kv_len = min(pos + 1, a.W);
  const int chunk = split_chunk(a.W, a.n_splits);
  const int s_begin = split * chunk;
  const int s_end = min(s_begin + chunk, kv_len);

  float qf[R][8];
  {
    u32x4 qraw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) qraw[r] = ld16(a.q + (size_t)b * a.ldq + (size_t)(kvh * R + r) * DH + dl * 8);
    load_q<R>(qf, qraw);
  }
  State<R> st;
  init_state<R>(st);

  // `kvh` is a SCHEDULED kv head: when a GQA ratio is split into groups, kv_groups consecutive scheduled heads read the
  // same real head's K/V (the repeat hits the XCD's L2) and own consecutive slices of its query heads.
  const int kv_real = kvh / a.kv_groups;
  // elements between consecutive slots: a whole row of kv heads in the reference's layout, ONE head row in the head-major
  // layout (common.cuh) - there the four lane groups of a wave read four consecutive slots = one contiguous KiB per load
  const int hkv_real = a.Hkv / a.kv_groups;
  const size_t row_stride = a.kv_layout ? (size_t)DH : (size_t)hkv_real * DH;
  const size_t ring0 = kv_offset(a.kv_layout, a.W, hkv_real * DH, DH, (size_t)b, 0, kv_real * DH) + dl * 8;
  const bf16_t* kbase = a.cache_k + ring0;
  const bf16_t* vbase = a.cache_v + ring0;

  // Each lane group walks slots s_begin + wid*4 + g + 16*j, UK slots (K and V rows = 2*UK loads) per step, two
  // steps in flight (ping-pong register sets A/B refilled in place, 16 KiB per wave outstanding): the kernel is pure
  // HBM latency/bandwidth, so depth is what matters.
  constexpr int UK = (R <= 4 && !SMALL) ? 4 : 2;  // R >= 6 needs the registers for its accumulators: half-size sets, no AGPR spills
  const int s_first = s_begin + wid * 4 + g;
  const int s_clamp = max(kv_len - 1, 0);
  const int n_steps = (s_end > s_begin) ? (s_end - s_begin + 16 * UK - 1) / (16 * UK) : 0;  // block-uniform
  u32x4 setA[2 * UK], setB[2 * UK];  // [0, UK): K rows, [UK, 2UK): V rows
  // ALWAYS exactly 2*UK unconditional loads (slots past the block's range are clamped to a valid slot and masked in
  // reduce_step): no branch around a load, so hipcc's wait for one set leaves the other set's loads in flight.
  auto load_step = [&](int it, u32x4 (&kv)[2 * UK]) {
    const int s0 = s_first + it * 16 * UK;
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const int sl = min(s0 + 16 * u, s_clamp);
      kv[u] = ld16_nt(kbase + (size_t)sl * row_stride);
      kv[UK + u] = ld16_nt(vbase + (size_t)sl * row_stride);
    }
  };
  auto reduce_step = [&](int it, const u32x4 (&kv)[2 * UK]) {
    const int s0 = s_first + it * 16 * UK;
#pragma unroll
    for (int u = 0; u < UK; ++u) reduce_slot<R>(st, qf, kv[u], kv[UK + u], (s0 + 16 * u) < s_end);
  };
  // Two sets in flight.  Steps beyond n_steps reduce nothing (every slot is masked).
  load_step(0, setA);
  load_step(1, setB);
  for (int it = 0; it < n_steps; it += 2) {
    reduce_step(it, setA);
    load_step(it + 2, setA);
    reduce_step(it + 1, setB);
    load_step(it + 3, setB);
  }

  // 4 lane groups -> wave -> LDS
  wave_state_to_lds<R>(st, wid, lane, sm_m, sm_l, sm_acc);
  __syncthreads();

  // 4 waves -> block partial in global scratch
  const int bh = b * a.Hkv + kvh;
  float* p_acc = a.partial + ((size_t)bh * a.n_splits + split) * R * DH;
  float* p_ml = a.partial + (size_t)a.B * a.Hkv * a.n_splits * R * DH + ((size_t)bh * a.n_splits + split) * R * 2;
  for (int idx = tid; idx < R * DH; idx += 256) {
    float A, M, L;
    split_partial<R>(idx, sm_m, sm_l, sm_acc, A, M, L);
    p_acc[idx] = A;
    if (idx % DH == 0) {
      p_ml[(idx / DH) * 2] = M;
      p_ml[(idx / DH) * 2 + 1] = L;
    }
  }
}

// ALL-IN form (round 6; batches of >= 3 sequences on rings of <= 4096 slots: 128 slots per block = 8 per lane group): the block's
// whole K/V share - 16 loads per lane - is issued at once and the query heads are served in passes of two over the SAME registers
// (heads do not interact in reduce_slot, a lane group still visits its slots in ascending order: bit-identical results).  The
// SMALL form above walks the same slots in four dependent steps with two in flight: three memory round trips where this has
// one (batch 3: 16.9 us per layer for 50 MB).  Fits 3 blocks per CU.
template <int R>
__global__ __launch_bounds__(256, 3) void attn_decode_allin_kernel(AttnDecodeArgs a) {
  static_assert(R == 2 || R == 4, "passes of two query heads");
  __shared__ float sm_m[4 * R];
  __shared__ float sm_l[4 * R];
  __shared__ float sm_acc[4 * R * DH];
  constexpr int NS = 8;  // slots per lane group: 4 waves x 4 groups x 8 = 128 slots per block

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, dl = lane & 15;
  const int kvh = blockIdx.x % a.Hkv, split = blockIdx.x / a.Hkv, b = blockIdx.y;
  const int pos = a.tok_pos[b];
  const int kv_len = min(pos + 1, a.W);
  const int chunk = split_chunk(a.W, a.n_splits);  // <= 128 (launch_r)
  const int s_begin = split * chunk;
  const int s_end = min(s_begin + chunk, kv_len);

  u32x4 qraw[R];
#pragma unroll
  for (int r = 0; r < R; ++r) qraw[r] = ld16(a.q + (size_t)b * a.ldq + (size_t)(kvh * R + r) * DH + dl * 8);

  const int kv_real = kvh / a.kv_groups;
  const int hkv_real = a.Hkv / a.kv_groups;
  const size_t row_stride = a.kv_layout ? (size_t)DH : (size_t)hkv_real * DH;
  const size_t ring0 = kv_offset(a.kv_layout, a.W, hkv_real * DH, DH, (size_t)b, 0, kv_real * DH) + dl * 8;
  const bf16_t* kbase = a.cache_k + ring0;
  const bf16_t* vbase = a.cache_v + ring0;
  const int s_first = s_begin + wid * 4 + g;
  const int s_clamp = max(kv_len - 1, 0);
  u32x4 kk[NS], vv[NS];  // ALWAYS 2 * NS unconditional loads (slots past the block's range: clamped, masked below)
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int sl = min(s_first + 16 * j, s_clamp);
    kk[j] = ld16_nt(kbase + (size_t)sl * row_stride);
    vv[j] = ld16_nt(vbase + (size_t)sl * row_stride);
  }
#pragma unroll
  for (int rp = 0; rp < R; rp += 2) {
    if (rp > 0) {  // the passes run one after the other on the RAW rows: shared, hipcc keeps every row's 16 converted floats live (145 spills)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NS; ++j) asm volatile("" : "+v"(kk[j]), "+v"(vv[j]));
    }
    float qf[2][8];
    const u32x4 qpair[2] = {qraw[rp], qraw[rp + 1]};
    load_q<2>(qf, qpair);
    State<2> st;
    init_state<2>(st);
#pragma unroll
    for (int j = 0; j < NS; ++j) reduce_slot<2>(st, qf, kk[j], vv[j], (s_first + 16 * j) < s_end);
    wave_state_to_lds_heads<2, R>(st, wid, lane, rp, sm_m, sm_l, sm_acc);
  }
  __syncthreads();

  const int bh = b * a.Hkv + kvh;
  float* p_acc = a.partial + ((size_t)bh * a.n_splits + split) * R * DH;
  float* p_ml = a.partial + (size_t)a.B * a.Hkv * a.n_splits * R * DH + ((size_t)bh * a.n_splits + split) * R * 2;
  for (int idx = tid; idx < R * DH; idx += 256) {
    float A, M, L;
    split_partial<R>(idx, sm_m, sm_l, sm_acc, A, M, L);
    p_acc[idx] = A;
    if (idx % DH == 0) {
      p_ml[(idx / DH) * 2] = M;
      p_ml[(idx / DH) * 2 + 1] = L;
    }
  }
}

// Second launch: one block per (sequence, q head), one thread per output element.  All of a
// thread's loads (the head's (m, l) pairs - same address across the block, so one transaction each - and its own
// accumulator column) are independent and issued together: one memory round trip instead of a chain.
template <int NS>  // upper bound on n_splits held in registers
__global__ __launch_bounds__(128) void attn_decode_combine_kernel(AttnDecodeArgs a) {
  const int d = threadIdx.x, b = blockIdx.y;
  const int R = a.H / a.Hkv, kvh = blockIdx.x % a.Hkv, r = blockIdx.x / a.Hkv, h = kvh * R + r;  // same XCD as the splits
  const int bh = b * a.Hkv + kvh;
  const float* all_acc = a.partial + (size_t)bh * a.n_splits * R * DH + (size_t)r * DH + d;
  const float* all_ml = a.partial + (size_t)a.B * a.Hkv * a.n_splits * R * DH + (size_t)bh * a.n_splits * R * 2 + r * 2;
  float m[NS], l[NS], v[NS];
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) {
    const int s2 = min(sp, a.n_splits - 1);  // clamped, never a conditional load
    const float2 ml = *reinterpret_cast<const float2*>(all_ml + (size_t)s2 * R * 2);
    m[sp] = (sp < a.n_splits) ? ml.x : -1e30f;
    l[sp] = (sp < a.n_splits) ? ml.y : 0.f;
    v[sp] = all_acc[(size_t)s2 * R * DH];
  }
  const float o = combine_splits<NS>(m, l, v, a.n_splits);
  reinterpret_cast<bf16_t*>(a.out)[(size_t)b * a.H * DH + (size_t)h * DH + d] = f_to_bf(o);
}

template <int R>
void launch_r(const AttnDecodeArgs& a, hipStream_t s) {
  dim3 grid(a.n_splits * a.Hkv, a.B), block(256);
  static int allin = -1;  // MI_ATTN_ALLIN=0: the stepping form for batches >= 3 (A/B testing)
  if (allin < 0) {
    const char* e = getenv("MI_ATTN_ALLIN");
    allin = e ? atoi(e) : 1;
  }
  if constexpr (R == 2 || R == 4) {
    if (a.B >= 3 && allin && attn_core::split_chunk(a.W, a.n_splits) <= 128) {
      hipLaunchKernelGGL((attn_decode_allin_kernel<R>), grid, block, 0, s, a);
      goto combine;
    }
  }
  if (R <= 4 && a.B >= 3) hipLaunchKernelGGL((attn_decode_kernel<R, (R <= 4)>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn_decode_kernel<R, false>), grid, block, 0, s, a);
combine:
  if (a.n_splits <= 16) hipLaunchKernelGGL((attn_decode_combine_kernel<16>), dim3(a.H, a.B), dim3(128), 0, s, a);
  else hipLaunchKernelGGL((attn_decode_combine_kernel<32>), dim3(a.H, a.B), dim3(128), 0, s, a);  // n_splits <= 32
}

}  // namespace

int attn_decode_splits(int W) {
  static int slots = 0;  // minimum ring slots per block; MI_ATTN_SPLIT_SLOTS overrides (tuning)
  if (slots == 0) {
    const char* e = getenv("MI_ATTN_SPLIT_SLOTS");
    slots = e ? atoi(e) : 128;
    if (slots < 16) slots = 128;
  }
  // at most 32 splits (the one-round-trip combine holds 32 partials per thread): widen the blocks beyond that
  int per = slots;
  if ((W + per - 1) / per > 32) per = (((W + 31) / 32) + 15) & ~15;
  int n = (W + per - 1) / per;
  if (n < 1) n = 1;
  return n;
}

size_t attn_decode_partial_floats(int B, int H, int Hkv, int Dh, int W) {
  const int R = H / Hkv;
  return (size_t)B * Hkv * attn_decode_splits(W) * R * (Dh + 2);
}

// Largest per-block query-head count in {8, 6, 4, 2, 1} that divides the GQA ratio: 12 (Mistral-Large) -> 2 groups of 6,
// 16 -> 2 x 8, 3 -> 3 x 1.  Each group is scheduled as a kv head of its own.
int attn_decode_group(int R) {
  for (int g : {8, 6, 4, 2}) if (R % g == 0) return g;
  return 1;
}

hipError_t launch_attn_decode(const AttnDecodeArgs& a_in, hipStream_t s) {
  if (a_in.Dh != DH || a_in.H % a_in.Hkv != 0) return hipErrorInvalidValue;
  AttnDecodeArgs a = a_in;
  const int R = attn_decode_group(a.H / a.Hkv);
  a.kv_groups = (a.H / a.Hkv) / R;
  a.Hkv = a_in.Hkv * a.kv_groups;
  switch (R) {
    case 1: launch_r<1>(a, s); break;
    case 2: launch_r<2>(a, s); break;
    case 4: launch_r<4>(a, s); break;
    case 6: launch_r<6>(a, s); break;
    case 8: launch_r<8>(a, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
