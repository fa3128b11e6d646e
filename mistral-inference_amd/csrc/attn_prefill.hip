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
// Block = (query tile, q head, sequence), NW waves x 32 query rows: 8 waves (256 queries) when the launch still fills
// the chip that way, else 4 waves (128 queries, two blocks per CU) - a bigger query tile halves the K/V staging and
// barrier cost per flop.  K/V tiles are double-buffered in LDS: the next tile's global loads are issued before this
// tile's math, written to the OTHER buffer after it, and one barrier per tile publishes them.  Per 64-key tile:
//   S^T = K . Q^T   (mfma 32x32x16; swapped so that a lane owns ONE query column: the softmax row
//                    reductions are in-lane plus one exchange with lane^32 - guide T12)
//   O^T += V^T . P^T with P^T taken directly from the S^T accumulator registers: the contraction index
//                    <-> key mapping of the B operand is permuted to match the C layout, and the A operand
//                    (V^T, staged transposed in LDS) is read with the same permutation.
// K tile: row-major in LDS, 16-byte slots XOR-swizzled by (key & 15) -> conflict-free ds_read_b128.
// V tile: transposed on the way in (2 or 4 keys x 8 d per thread, register transpose), the 4-key quads of every 16-key
// group stored in the order 0, 2, 1, 3 so that a lane's 8 keys of one MFMA are one ds_read_b128; row stride 144 B ->
// conflict-free (see VT_STRIDE).
#include <cstdlib>
#include <type_traits>

#include "common.cuh"
#include "kernels.h"

// A second compile with -DATTN_F16=1 (build_native.py: attn_prefill_f16.o) is the same kernel for fp16 storage: the MFMA
// opcode (v_mfma_f32_32x32x16_f16), P rounded to fp16 instead of bf16 for P.V (what fp16 flash kernels do), fp16 output
// packing; entry point launch_attn_prefill_f16 (generic.hip: prefills of fp16 models with 128-wide heads).  The default
// compile is untouched by this block.
#ifndef ATTN_F16
#define ATTN_F16 0
#endif
#if ATTN_F16
typedef _Float16 attn_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t attn_h_pack2(float lo, float hi) {
  return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)lo) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)hi) << 16);
}
#define bf16x8 attn_f16x8
#define cvt_pk_bf16 attn_h_pack2
#define pack_bf2 attn_h_pack2
#define ATTN_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define launch_attn_prefill launch_attn_prefill_f16
#define attn_prefill_set_mode attn_prefill_set_mode_f16
#else
#define ATTN_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif

#ifndef ATT_FASTLOAD
#define ATT_FASTLOAD 1  // 0 = every tile through the general (per-lane source selection) staging path: the round 1-5 form
#endif
#ifndef ATT_PRIO
#define ATT_PRIO 0  // experiment switch: 1 = s_setprio(1) around the two MFMA clusters, 2 = static priority 1 for waves 4..7
#endif

#ifndef ATT_TRACE
#define ATT_TRACE 0  // 1 = (probe builds only) every block records {start, end} of s_memrealtime, its XCC / CU ids: scripts/attn_prefill_trace.py
#endif
#if ATT_TRACE
__device__ unsigned long long g_attn_trace[16384 * 4];
extern "C" int mi_probe_attn_trace(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_trace), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

namespace {

constexpr int DH = 128;
constexpr int KT = 64;         // keys per tile
#ifndef ATT_VB128
#define ATT_VB128 1  // 0 = V^T rows in key order, 136-byte stride, fragments as two 8-byte reads (rounds 1-6: hipcc fuses them to ds_read2_b64)
#endif
// bytes per d-row of the transposed V tile.  ATT_VB128: inside every 16-key group the four 4-key quads are stored in the order
// 0, 2, 1, 3, so that the 8 keys one lane feeds to a P.V MFMA ({0..3, 8..11} + 4 * half) are 16 CONTIGUOUS bytes: one
// ds_read_b128 (4 LDS cycles per wave, conflict-free at a 144-byte row stride) instead of a ds_read2_b64, which the LDS
// serves in 16 cycles (cdna_hip_programming.md, LDS: "16 bytes cost 16 as ds_read2_b64 and 4 as ds_read_b128") - the V^T
// fragments were 2048 of the ~4300 cycles a key tile takes on a CU.
constexpr int VT_STRIDE = ATT_VB128 ? 144 : 136;
constexpr int KS_BYTES = KT * DH * 2;
constexpr int VT_BYTES = DH * VT_STRIDE;
constexpr float LOG2E = 1.4426950408889634f;

constexpr int BUF_BYTES = KS_BYTES + VT_BYTES;  // one K tile + one transposed V tile

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_prefill_kernel(AttnPrefillArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x BUF_BYTES
  constexpr int NT = NW * 64, QB = NW * 32;  // threads, queries per block

  const int tid = threadIdx.x, lane = tid & 63;
#if ATT_TRACE
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  // provably wave-uniform: everything derived from it (this wave's query range, `skip`, `full`) becomes scalar control
  // flow; with a VGPR-derived wave id hipcc predicated the 32-score masking code lane by lane on every tile.
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hf = lane >> 5, ql = lane & 31;
  // Block -> (query tile, head).  Workgroup i runs on XCD i % 8 and every XCD has its own L2: consecutive ids cycle
  // through the KV heads (id % Hkv), so with Hkv = 8 all blocks of one KV head share one XCD and its K/V (2 KiB per
  // position) stays L2-resident instead of every XCD streaming all heads.  Within a KV head the heaviest query tiles
  // (causal work grows with the tile index) are handed out first so that short ones fill the tail of the launch.
  const int R = a.H / a.Hkv;
  const int kvh = blockIdx.x % a.Hkv, wq = blockIdx.x / a.Hkv;
  const int n_qt = gridDim.x / a.H;
  const int qt = n_qt - 1 - wq / R, h = kvh * R + wq % R, b = blockIdx.y;
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
  if (qt * QB >= s_b) return;
  const int W = a.W;
  const int n_old = min(p_b, W);

  // key-position range this block needs
  const int qp_blk_lo = p_b + qt * QB;
  const int qp_blk_hi = p_b + min(qt * QB + QB - 1, s_b - 1);
  const int kp_lo = a.causal ? max(p_b - n_old, qp_blk_lo - W + 1) : 0;
  const int kp_hi = a.causal ? qp_blk_hi : s_b - 1;
  const int n_tiles = (kp_hi - kp_lo + KT) / KT;

  // this wave's queries
  const int qi = qt * QB + wid * 32 + ql;             // index inside the sequence
  const int qi_c = min(qi, s_b - 1);
  const int qp = p_b + qi_c;
  const int qp_w_lo = p_b + qt * QB + wid * 32;        // wave's lowest / highest query position
  const int qp_w_hi = p_b + min(qt * QB + wid * 32 + 31, s_b - 1);
  const bool wave_active = (qt * QB + wid * 32) < s_b;

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
  const float sc = a.scale * LOG2E;

  // ---- staging assignment.  K: 16-byte piece p = tid + NT*j -> key p >> 4, slot p & 15.  V: a thread takes the same
  // 16-byte d slice of KPT consecutive keys (4 with 256 threads, 2 with 512) and transposes them in registers.
  constexpr int PPT = 1024 / NT;  // K pieces (and V keys) per thread
  constexpr int KPT = PPT;
  const int k_slot = tid & 15, k_key0 = tid >> 4;
  const int v_kg = tid % (64 / KPT), v_ds = tid / (64 / KPT);  // key group, d slice (8 d)
  u32x4 rk[PPT], rv[KPT];
  // Source row of key position kp: ring slot (kp % W) for keys older than this forward, activation row otherwise.
  // Both candidate addresses are formed and SELECTED (v_cndmask), and the load itself is unconditional from a
  // clamped position: a `cond ? load : 0` makes hipcc branch around every load and wait vmcnt(0) after each one
  // (cdna_hip_programming.md ".s-level traps" (c)), which serialised this staging to one load in flight.
  // Keys past kp_hi are masked in the softmax (their P is exactly 0), so the clamped duplicates are harmless.
  const size_t ring0 = kv_offset(a.kv_layout, W, kv_dim, DH, (size_t)b, 0, kvh * DH);
  const int ring_stride = a.kv_layout ? DH : kv_dim;  // elements between consecutive ring slots of this kv head (common.cuh)
  const bf16_t* ring_k0 = a.cache_k ? a.cache_k + ring0 : a.qkv;
  const bf16_t* ring_v0 = a.cache_v ? a.cache_v + ring0 : a.qkv;
  const bf16_t* act_k0 = a.qkv + nq_cols + (size_t)kvh * DH;
  const bf16_t* act_v0 = act_k0 + kv_dim;
  // kp % W costs ~17 VALU instructions per load when done per lane (there is no integer divide); the tile's first key
  // is wave-uniform, so the modulo is taken ONCE per tile on the scalar side and a lane only adds its key offset and
  // wraps (offsets are < 64: one conditional subtract when W >= 64, the general modulo only for toy windows).
  // Element offsets are 32-bit: slot * kv_dim < W * kv_dim and row * ld < T * ld are checked on the host (< 2^31).
  const bool big_w = W >= KT;
  auto gload_impl = [&](int it, auto big) {
    constexpr bool BIG = decltype(big)::value;
    const int t_lo = kp_lo + it * KT;
    const int slot_lo = __builtin_amdgcn_readfirstlane(t_lo % W);
    auto key_off = [&](int kp) -> int {  // element offset relative to the ring / activation base
      int slot = slot_lo + (kp - t_lo);
      if constexpr (BIG) slot = (slot >= W) ? slot - W : slot;
      else slot = kp % W;
      const int arow = row0 + kp - p_b;
      return (kp < p_b) ? slot * ring_stride : arow * a.ld;
    };
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int kpk = min(t_lo + k_key0 + (NT / 16) * j, kp_hi);
      const bf16_t* kb = (kpk < p_b) ? ring_k0 : act_k0;
      rk[j] = ld16(kb + key_off(kpk) + k_slot * 8);
      const int kpv = min(t_lo + v_kg * KPT + j, kp_hi);
      const bf16_t* vb = (kpv < p_b) ? ring_v0 : act_v0;
      rv[j] = ld16(vb + key_off(kpv) + v_ds * 8);
    }
  };
#if ATT_FASTLOAD
  // The common tile - 64 consecutive rows of ONE source (activation rows, or ring slots that do not wrap), none past kp_hi -
  // needs no per-lane selection at all: a wave-uniform base (row of the tile's first key) on the scalar side, one multiply-add
  // per lane for (key in tile, 16-byte column) and the loads.  The general path above costs ~110 VALU + ~80 SALU
  // instructions per tile (divergent selects, v_mul_lo_u32, a scalar modulo) - 40 % of the loop's VALU issue - and stays for
  // the tiles that straddle the ring / activation seam, wrap, or end at kp_hi.  The ring slot of a tile's first key is
  // carried from tile to tile (gload is called with it = 0, 1, 2, ... in order) instead of a modulo per tile.
  int ld_slot = __builtin_amdgcn_readfirstlane(kp_lo % W);
  auto gload = [&](int it) {
    const int t_lo = kp_lo + it * KT;
    const int slot_lo = ld_slot;
    ld_slot += KT;
    if (ld_slot >= W) ld_slot -= W;  // (exact when W >= KT; the !big_w path does not use it)
    const bool all_act = t_lo >= p_b;
    const bool all_ring = t_lo + KT - 1 < p_b && slot_lo + KT - 1 < W;
    if (big_w && t_lo + KT - 1 <= kp_hi && (all_act || all_ring)) {
      const int stride = all_act ? a.ld : ring_stride;
      const size_t first = all_act ? (size_t)(row0 + t_lo - p_b) * a.ld : (size_t)slot_lo * ring_stride;
      const bf16_t* kb = (all_act ? act_k0 : ring_k0) + first;
      const bf16_t* vb = (all_act ? act_v0 : ring_v0) + first;
      const uint32_t ko = (uint32_t)(k_key0 * stride + k_slot * 8);
      const uint32_t vo = (uint32_t)(v_kg * KPT * stride + v_ds * 8);
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        rk[j] = ld16(kb + (size_t)((NT / 16) * j) * stride + ko);
        rv[j] = ld16(vb + (size_t)j * stride + vo);
      }
    } else if (big_w) {
      gload_impl(it, std::true_type{});
    } else {
      gload_impl(it, std::false_type{});
    }
  };
#else
  auto gload = [&](int it) {  // one scalar branch per tile instead of one per load
    if (big_w) gload_impl(it, std::true_type{});
    else gload_impl(it, std::false_type{});
  };
#endif
  auto lstore = [&](char* Ks, char* Vt) {
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int key = k_key0 + (NT / 16) * j;
      st16(Ks + key * (DH * 2) + ((k_slot ^ (key & 15)) << 4), rk[j]);
    }
    // byte offset of this thread's first key inside a V^T row (ATT_VB128: quads of a 16-key group in the order 0, 2, 1, 3)
    const int vk0 = v_kg * KPT;
    const int v_pos = ATT_VB128 ? ((vk0 & ~15) | (((vk0 >> 2) & 1) << 3) | (((vk0 >> 3) & 1) << 2) | (vk0 & 3)) * 2 : vk0 * 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // d = v_ds*8 + 2c (low halves) and 2c+1 (high halves); KPT keys packed per row
      if constexpr (KPT == 4) {
        u32x2 lo, hi;
        lo[0] = (rv[0][c] & 0xffffu) | (rv[1][c] << 16);
        lo[1] = (rv[2][c] & 0xffffu) | (rv[3][c] << 16);
        hi[0] = (rv[0][c] >> 16) | (rv[1][c] & 0xffff0000u);
        hi[1] = (rv[2][c] >> 16) | (rv[3][c] & 0xffff0000u);
        *reinterpret_cast<u32x2*>(Vt + (v_ds * 8 + 2 * c) * VT_STRIDE + v_pos) = lo;
        *reinterpret_cast<u32x2*>(Vt + (v_ds * 8 + 2 * c + 1) * VT_STRIDE + v_pos) = hi;
      } else {
        const uint32_t lo = (rv[0][c] & 0xffffu) | (rv[1][c] << 16);
        const uint32_t hi = (rv[0][c] >> 16) | (rv[1][c] & 0xffff0000u);
        *reinterpret_cast<uint32_t*>(Vt + (v_ds * 8 + 2 * c) * VT_STRIDE + v_pos) = lo;
        *reinterpret_cast<uint32_t*>(Vt + (v_ds * 8 + 2 * c + 1) * VT_STRIDE + v_pos) = hi;
      }
    }
  };

  // Split staging (guide T14) into a double-buffered LDS image: tile it+1's global loads are issued before tile it's
  // math and sit in registers (rk/rv) until the math is done; they are then written to the OTHER buffer - whose last
  // readers passed the barrier that ended tile it-1 - and the barrier that ends tile it publishes them.
  gload(0);
  lstore(smem, smem + KS_BYTES);
  __syncthreads();
  if (ATT_PRIO == 2 && wid >= 4) __builtin_amdgcn_s_setprio(1);
#if ATT_TRACE == 2
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter();
#define ATT_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - ph_t; ph_t = now_; }
#else
#define ATT_STAMP(i)
#endif
  for (int it = 0; it < n_tiles; ++it) {
    if (it + 1 < n_tiles) gload(it + 1);
    ATT_STAMP(0)
    const char* Ks = smem + (it & 1) * BUF_BYTES;
    const char* Vt = Ks + KS_BYTES;

    const int t_lo = kp_lo + it * KT;
    const int t_hi = min(t_lo + KT - 1, kp_hi);
    const bool skip = !wave_active || (a.causal && (t_lo > qp_w_hi || t_hi <= qp_w_lo - W));
    if (!skip) {
      // ---- S^T = K . Q^T
      // Fragment reads are issued four k-steps (8 ds_read_b128) ahead of the MFMAs that consume them and the two
      // 32-key accumulators alternate, so neither the LDS latency nor the dependent-accumulate latency of a
      // 32x32x16 MFMA sits between consecutive MFMAs (the straightforward read -> wait -> mfma loop ran this phase
      // as one serial chain).
      f32x16 st[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[mb][r] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        bf16x8 kf[4][2];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const int key = mb * 32 + ql, slot = (kh * 4 + k4) * 2 + hf;
            kf[k4][mb] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(Ks + key * (DH * 2) + ((slot ^ (key & 15)) << 4)));
          }
        __builtin_amdgcn_sched_barrier(0);  // keep the 8 reads ahead of the MFMAs (hipcc sinks them back otherwise)
        if (ATT_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            st[mb] = ATTN_MFMA(kf[k4][mb], qf[kh * 4 + k4], st[mb], 0, 0, 0);
        if (ATT_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      }
      ATT_STAMP(1)
      // ---- mask, online softmax (this lane: query ql, keys of its half).  Scores stay RAW (unscaled): the
      // 1/sqrt(d) * log2(e) factor is folded into the exponent's fma, p = exp2(s * sc - m * sc).
      const bool full = a.causal ? (t_hi <= qp_w_lo && t_lo > qp_w_hi - W && t_lo + KT - 1 <= kp_hi) : (t_lo + KT - 1 <= kp_hi);
      // visible keys of this lane's query: vis_lo < kp <= vis_hi
      const int vis_hi = a.causal ? min(kp_hi, qp) : kp_hi;
      const int vis_lo = a.causal ? qp - W : -1;
      if (!full) {  // ONE scalar branch around the whole masking pass (diagonal / window-edge / ragged tiles only)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kp = t_lo + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const bool vis = (kp <= vis_hi) & (kp > vis_lo);
            st[mb][r] = vis ? st[mb][r] : -INFINITY;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[mb][r]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // max with lane ^ 32 (guide T12)
      }
      // Deferred rescale (guide T13): keep the old running max while no lane's maximum grew by more than 2^8 in the
      // exponent domain - P is then bounded by 256 instead of 1, which bf16/fp32 hold without loss of relative
      // precision - and skip the 64-multiply rescale of O.  The decision is wave-uniform, taken before this tile's
      // P.V, and l and O always see the same factor.
      const float m_cand = fmaxf(m_run, mx);
      float alpha = 1.0f;
      if (__any((m_cand - m_run) * sc > 8.0f)) {
        alpha = fast_exp2((m_run - m_cand) * sc);
        m_run = m_cand;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
      }
      ATT_STAMP(2)
      const float nm = -m_run * sc;
      float psum = 0.f;
      uint32_t pb[2][2][4];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = fast_exp2(fmaf(st[mb][r], sc, nm)), p1 = fast_exp2(fmaf(st[mb][r + 1], sc, nm));
          psum += p0 + p1;
          pb[mb][r >> 3][(r & 7) >> 1] = cvt_pk_bf16(p0, p1);
        }
      l_run = l_run * alpha + psum;
      if (ATT_PRIO == 1) __builtin_amdgcn_s_setprio(1);
      // ---- O^T += V^T . P^T  (left to hipcc's scheduler, which overlaps the second key half's exponentials with the
      // first half's MFMAs; hand-pipelining the V^T fragment reads one group ahead measured no better)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          u32x4 pw = {pb[mb][kb][0], pb[mb][kb][1], pb[mb][kb][2], pb[mb][kb][3]};
          const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
#if ATT_VB128
            const u32x4 vw = *reinterpret_cast<const u32x4*>(Vt + (dt * 32 + ql) * VT_STRIDE + (mb * 32 + kb * 16 + 8 * hf) * 2);
#else
            const char* vrow = Vt + (dt * 32 + ql) * VT_STRIDE + (mb * 32 + kb * 16 + 4 * hf) * 2;
            const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow);
            const u32x2 v1 = *reinterpret_cast<const u32x2*>(vrow + 16);
            const u32x4 vw = {v0[0], v0[1], v1[0], v1[1]};
#endif
            acc[dt] = ATTN_MFMA(__builtin_bit_cast(bf16x8, vw), pf, acc[dt], 0, 0, 0);
          }
        }
      if (ATT_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    }
    ATT_STAMP(3)
    if (it + 1 < n_tiles) lstore(smem + ((it + 1) & 1) * BUF_BYTES, smem + ((it + 1) & 1) * BUF_BYTES + KS_BYTES);
    ATT_STAMP(4)
    __syncthreads();
    ATT_STAMP(5)
  }
#if ATT_TRACE == 2
  if (lane == 0 && (wid == 0 || wid == 7)) {
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (bid < 1024)
      for (int i = 0; i < 6; ++i) g_attn_trace[4096 * 4 + (bid * 2 + (wid ? 1 : 0)) * 6 + i] = ph[i];
  }
#endif

#if ATT_TRACE
  if (tid == 0) {
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (bid < 16384) {
      unsigned hw = 0, xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_attn_trace[bid * 4 + 0] = tr_t0;
      g_attn_trace[bid * 4 + 1] = __builtin_amdgcn_s_memrealtime();
      g_attn_trace[bid * 4 + 2] = ((unsigned long long)xcc << 32) | hw;
      g_attn_trace[bid * 4 + 3] = (unsigned long long)n_tiles;
    }
  }
#endif
  // ---- epilogue: O^T[d][q] / l  ->  out[q][h*128 + d]
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);  // (once per block: not worth a permlane)
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

template <int NW>
static hipError_t launch_nw(const AttnPrefillArgs& a, hipStream_t s) {
  // 2 x 34 KiB of dynamic LDS: above the 64 KiB default limit.  The opt-in is per function AND per device.
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_prefill_kernel<NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  constexpr int QB = NW * 32;
  dim3 grid(((a.max_q_len + QB - 1) / QB) * a.H, a.causal ? a.B : 1), block(NW * 64);
  hipLaunchKernelGGL((attn_prefill_kernel<NW>), grid, block, 2 * BUF_BYTES, s, a);
  return hipGetLastError();
}

static int g_force_waves = -1;
void attn_prefill_set_mode(int waves) {
  if (waves >= 0) g_force_waves = waves;
}

hipError_t launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t s) {
  if (a.Dh != DH || a.H % a.Hkv != 0) return hipErrorInvalidValue;
  int& force = g_force_waves;  // MI_ATTN_PREFILL_WAVES=4|8 / mi_debug_set_prefill_kernels pin the block shape (A/B testing)
  if (force < 0) {
    const char* e = getenv("MI_ATTN_PREFILL_WAVES");
    force = e ? atoi(e) : 0;
  }
  // 256-query blocks (one 8-wave block per CU) once they come in at least two rounds of the chip, else 128-query blocks (two
  // 4-wave blocks per CU, each other's prologue / epilogue cover): 2048 tokens x 32 heads = one round of 256-query blocks
  // measured 75.9 us, as 128-query blocks 70.6; 4096 tokens (two rounds) 180 against 183 (round 6, scripts/attn_prefill_probe.py)
  const long blocks8 = (long)((a.max_q_len + 255) / 256) * a.H * (a.causal ? a.B : 1);
  const bool eight = force ? force == 8 : blocks8 >= 2 * device_cus();
  return eight ? launch_nw<8>(a, s) : launch_nw<4>(a, s);
}
