// bf16 MFMA GEMM, 256x256x64 block tile, for the large prefill shapes: out[M,N] = epilogue(A[M,K] . W[N,K]^T).
//
// Same contract and epilogues as gemm.hip (transformer_layers.py:66,93,105-106; transformer.py:235); launch_gemm
// picks this kernel when M >= 256 and K % 64 == 0, and for the token-grouped MoE form (moe.py:28-32, one launch for all
// experts) when its tile table was built with 256-row m-tiles.
//
// Epilogues: store / residual add / SiLU*mul / fp32 logits as in gemm.hip, plus GEMM_LOGPROB (the LM head reduced to
// log-softmax pieces without storing logits, generate.py:101-118).
//
// Why a second tile shape: the 128x128 kernel moves 32 KiB of operands through LDS per 128x128x64 MACs and is bound by
// LDS bandwidth (DMA writes ~64-85 B/clk + fragment reads 256 B/clk against 16 clk per MFMA): ~37 % of the MFMA peak.
// 256x256 halves the LDS bytes per flop.  Structure (cdna_hip_programming.md section 5, "256^2 8-phase template",
// written from its description):
//   * 8 waves = 2 (M) x 4 (N); a wave owns 64 rows of each 128-row A half and 32 columns of each 128-column B half,
//     i.e. four 64x32 quadrants (ha, hb), 16 MFMA 16x16x32 each per K tile;
//   * LDS = 2 stages x {A0, A1, B0, B1} x 16 KiB half tiles (128 rows x 128 B, 16-byte slot XOR-swizzled by row & 7,
//     the swizzle applied on the DMA's SOURCE address);
//   * one K tile = 4 phases, one quadrant each.  A phase (a) reads the fragments it is missing, (b) re-stages ONE
//     half tile whose last reader finished a phase earlier (2 `global_load_lds_dwordx4` per lane), (c) lgkmcnt(0) +
//     raw s_barrier, (d) 16 MFMAs.  The DMAs are never drained inside the loop: the single `s_waitcnt vmcnt(6)` per
//     K tile (phase 4) retires the NEXT tile's four halves and leaves the three halves issued after them in flight
//     across the barrier.
// Hazards, by construction:  RAW - tile t+1's halves are waited for (vmcnt) by every wave BEFORE phase 4's barrier and
// first read AFTER it.  WAR - a half is re-staged in the phase after the one whose barrier followed its last reads
// (lgkmcnt(0) precedes every barrier).
#include <cstdlib>

#include "common.cuh"
#include "kernels.h"

// A second compile of this file with -DG256_F16=1 (build_native.py: gemm256_f16.o) is the SAME kernel for fp16 storage - the
// data movement of two 16-bit types is identical; what changes is the MFMA opcode (v_mfma_f32_16x16x32_f16) and the half
// conversions / rounding points of the epilogues - under the entry points launch_gemm256_f16 & co., used by the generic
// storage path (generic.hip) for prefills of fp16 models.  The default compile is untouched by this block (its ISA hash is
// checked like the decode engine's).
#ifndef G256_F16
#define G256_F16 0
#endif
#if G256_F16
typedef _Float16 g256_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float g256_h_to_f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t g256_h_from_f(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float g256_h_round(float f) { return (float)(_Float16)f; }
__device__ __forceinline__ uint32_t g256_h_pack2(float lo, float hi) {
  return (uint32_t)g256_h_from_f(lo) | ((uint32_t)g256_h_from_f(hi) << 16);
}
__device__ __forceinline__ float g256_h_lo(uint32_t u) { return g256_h_to_f((uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float g256_h_hi(uint32_t u) { return g256_h_to_f((uint16_t)(u >> 16)); }
__device__ __forceinline__ float g256_h_swiglu(float acc1, float acc3) {  // swiglu_bf_fast with fp16 rounding points
  const float a = g256_h_round(acc1), b = g256_h_round(acc3);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * a);
  const float s = g256_h_round(a * __builtin_amdgcn_rcpf(1.0f + e));
  return s * b;
}
#define bf_round g256_h_round
#define pack_bf2 g256_h_pack2
#define f_to_bf g256_h_from_f
#define bf_to_f g256_h_to_f
#define bf_lo g256_h_lo
#define bf_hi g256_h_hi
#define swiglu_bf_fast g256_h_swiglu
#define bf16x8 g256_f16x8
#define G256_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#define gemm256_applicable gemm256_applicable_f16
#define gemm256_half_applicable gemm256_half_applicable_f16
#define launch_gemm256_half launch_gemm256_half_f16
#define launch_gemm256 launch_gemm256_f16
#else
#define G256_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

// Build-time experiment switches (all off in the shipped build; the experimental main loops live OUTSIDE the product tree, in scripts/probes/gemm256_experiments.inc):
#ifndef G256_PRIO
#define G256_PRIO 0
#endif
#ifndef G256_BAL
#define G256_BAL 0
#endif
#ifndef G256_DMA_AFTER
#define G256_DMA_AFTER 1   // 1 (shipped): a read segment issues its fragment reads before its half-tile DMA
#endif
#ifndef G256_SPLIT
#define G256_SPLIT 0
#endif
#ifndef G256_PIPE
#define G256_PIPE 0
#endif
#ifndef G256_ABL
#define G256_ABL 0
#endif
#ifndef G256_CLK
#define G256_CLK 0         // 1: block 0 records shader-clock and 100 MHz ticks across its main loop (mi_debug_gemm_clock)
#endif

#if G256_CLK
__device__ unsigned long long g256_clk[2];
extern "C" int mi_debug_gemm_clock(unsigned long long* out2) {
  return (int)hipMemcpyFromSymbol(out2, HIP_SYMBOL(g256_clk), sizeof(g256_clk));
}
#endif

namespace {

constexpr int BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;     // 16 KiB: 128 rows x 128 B
constexpr int STAGE_BYTES = 4 * HALF_BYTES;  // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // 128 KiB
// Half-height form (MH = 1, round 4): a 128 x 256 output tile = ONE A half against both B halves, two phases per K tile,
// three stages of {A0, B0, B1} (144 KiB).  It exists for the LAST, partly filled round of a launch: 384 tiles of
// 256 x 256 (q|k|v of Mistral-7B at 4096 tokens) are 1.5 rounds of 256 CUs; the columns of the half round used to go
// to the 128 x 128 kernel (twice the LDS bytes per flop: 0.72 PF); as 128 x 256 tiles they are exactly one full round
// at 1.5x the LDS bytes per flop of the square tile.  Same MFMA shape, same k order: bit-identical outputs.
constexpr int STAGE_BYTES_H = 3 * HALF_BYTES;
constexpr int LDS_BYTES_H = 3 * STAGE_BYTES_H;  // 144 KiB

__device__ __forceinline__ const bf16_t* seg_row256(const GemmArgs& g, int r) {
  if (r < g.n0) return g.w0 + (size_t)r * g.K;
  if (r < g.n1) return g.w1 + (size_t)(r - g.n0) * g.K;
  return g.w2 + (size_t)(r - g.n1) * g.K;
}

__device__ __forceinline__ void dma16(const bf16_t* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_barrier is not a memory operation for the compiler: the asm statements (memory clobber) on both sides keep every
// LDS read and every DMA on its side of the barrier.
__device__ __forceinline__ void raw_barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_reads_done_then_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();
}

template <int EPI, bool STAGGER, int MH = 2>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NOUT = (EPI == GEMM_SWIGLU) ? 128 : 256;  // output columns per block
  constexpr int MROWS = MH * 128;                         // output rows per block
  constexpr int STG = (MH + 2) * HALF_BYTES;              // bytes per stage: the A halves, B0, B1
  static_assert(MH == 2 || (EPI == GEMM_STORE || EPI == GEMM_RESIDUAL), "half-height tiles: store / residual epilogues");
  // bf16 outputs: MFMAs issued as W-fragment x A-fragment (transposed 16x16 result tiles, see the epilogue); fp32
  // logits: A x W, whose 4-byte stores already cover 64-byte row segments and measured faster than 16-byte ones.
  constexpr bool kSwap = EPI != GEMM_LOGITS && EPI != GEMM_LOGPROB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
#if G256_CLK
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- block -> tile.  Block b runs on XCD b % 8; a 4 (m) x 8 (n) supertile = the 32 blocks one XCD runs at a time
  // stays on one XCD so that its blocks share 4 A panels and 8 W panels in that XCD's L2 (guide T1; speed only).
  const bool grouped = g.tile_tab != nullptr;  // token-grouped MoE form: m-tiles come from the device tile table
  const int m_tiles = grouped ? g.max_m_tiles : (g.M + MROWS - 1) / MROWS, n_tiles = (g.N + NOUT - 1) / NOUT;
  const int MS = (m_tiles + 3) >> 2, NS = (n_tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int st = (q >> 5) * 8 + xcd, wi = q & 31;
  if (st >= MS * NS) return;
  const int m_tile = (st % MS) * 4 + (wi & 3), n_tile = (st / MS) * 8 + (wi >> 2);
  if (m_tile >= m_tiles || n_tile >= n_tiles) return;
  int row0, rows_valid;
  const bf16_t *w0 = g.w0, *w1 = g.w1;
  if (grouped) {  // one tile of one expert: {expert, first compact row, valid rows (<= 256)}
    if (m_tile >= *g.n_tiles_ptr) return;
    const int e = g.tile_tab[m_tile * 4];
    row0 = g.tile_tab[m_tile * 4 + 1];
    rows_valid = g.tile_tab[m_tile * 4 + 2];
    w0 = reinterpret_cast<const bf16_t*>(g.expert_tab[e * 3 + g.w_sel0]);
    if (g.w_sel1 >= 0) w1 = reinterpret_cast<const bf16_t*>(g.expert_tab[e * 3 + g.w_sel1]);
  } else {
    row0 = m_tile * MROWS;
    rows_valid = min(MROWS, g.M - row0);
  }

  // ---- DMA sources.  One instruction fills 8 consecutive 128-B rows of a half tile; wave w, piece j covers rows
  // (2w + j) * 8 .. + 8; lane l lands at (row + (l >> 3), slot l & 7) and fetches global slot (l & 7) ^ (row & 7).
  const int sslot = (lane & 7) ^ ((lane >> 3) & 7);
  const bf16_t* src[MH + 2][2];  // halves of a stage: A0 (A1) B0 B1
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wid * 2 + j) * 8 + (lane >> 3);
      if (h < MH) {
        int m = row0 + min(h * 128 + r, rows_valid - 1);
        if (g.a_gather) m = g.a_gather[m];
        src[h][j] = g.a + (size_t)m * g.lda + sslot * 8;
      }
      if (EPI == GEMM_SWIGLU) {
        const int n = min(n_tile * 128 + r, g.N - 1);
        src[MH + h][j] = (h == 0 ? w0 : w1) + (size_t)n * g.K + sslot * 8;
      } else {
        const int n = min(n_tile * 256 + h * 128 + r, g.N - 1);
        src[MH + h][j] = (grouped ? w0 + (size_t)n * g.K : seg_row256(g, n)) + sslot * 8;
      }
    }
  char* const my_piece = smem + wid * 2048;  // this wave's two 1-KiB pieces inside any half tile
#define STAGE_HALF(HALF, KT, STAGE)                                                              \
  do {                                                                                           \
    char* dst_ = my_piece + (STAGE) * STG + (HALF) * HALF_BYTES;                                 \
    dma16(src[HALF][0] + (size_t)(KT) * BK, dst_);                                               \
    dma16(src[HALF][1] + (size_t)(KT) * BK, dst_ + 1024);                                        \
  } while (0)

  // ---- fragment addressing (MFMA 16x16x32: lane holds row lane & 15, k = (lane >> 4) * 8 .. + 8 of each 32-wide k step)
  const int fq = lane >> 4;
  const int sw0 = ((fq) ^ (lane & 7)) << 4, sw1 = ((4 + fq) ^ (lane & 7)) << 4;
  const int a_off = (wr * 64 + (lane & 15)) * 128;
  const int b_off = MH * HALF_BYTES + (wc * 32 + (lane & 15)) * 128;

  f32x4 acc[MH][2][4][2];
#pragma unroll
  for (int a = 0; a < MH; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if G256_PIPE || G256_BAL || G256_SPLIT || G256_ABL || G256_PRIO || !G256_DMA_AFTER
#include "../../scripts/probes/gemm256_experiments.inc"  // probe builds only (scripts/build_variants.py gemm): never part of the shipped library
#else
  bf16x8 af[2][4];             // [k step][row fragment] of the A half in use
  bf16x8 b0x[2][2], b1[2][2];  // [k step][column fragment] of B half 0 / 1
#define LDS_FRAG(ADDR) __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ADDR))
#define READ_A(HA, SB)                                                               \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
    af[0][i] = LDS_FRAG((SB) + (HA) * HALF_BYTES + a_off + i * 2048 + sw0);          \
    af[1][i] = LDS_FRAG((SB) + (HA) * HALF_BYTES + a_off + i * 2048 + sw1);          \
  }
#define READ_B(DST, HB, SB)                                                          \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                    \
    DST[0][j] = LDS_FRAG((SB) + (HB) * HALF_BYTES + b_off + j * 2048 + sw0);         \
    DST[1][j] = LDS_FRAG((SB) + (HB) * HALF_BYTES + b_off + j * 2048 + sw1);         \
  }
#define SEGMENT_END()                          \
  do {                                         \
    lds_reads_done_then_barrier();             \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)
#define MFMA_END()                             \
  do {                                         \
    if (STAGGER) {                             \
      __builtin_amdgcn_sched_barrier(0);       \
      raw_barrier();                           \
      __builtin_amdgcn_sched_barrier(0);       \
    }                                          \
  } while (0)
#define MFMA_QUAD(HA, HB, BF)                                                                                   \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
    acc[HA][HB][i][j] = kSwap ? G256_MFMA(BF[ks][j], af[ks][i], acc[HA][HB][i][j], 0, 0, 0) \
                              : G256_MFMA(af[ks][i], BF[ks][j], acc[HA][HB][i][j], 0, 0, 0);

  const int nk = g.K / BK;
  if constexpr (MH == 1) {
    // ---- half-height tile: stage = {A0, B0, B1} (half indices 0, 1, 2), three stages, tile t in stage t % 3.  Tile
    // t + 2 is staged during tile t into the stage tile t - 1 was read from (its last reads - by the group that runs one
    // barrier behind - ended with the barrier before this tile's first read segment), and is waited for one tile later
    // (vmcnt(6): everything but the six pieces issued since), two barriers before its first read.
    STAGE_HALF(0, 0, 0);
    STAGE_HALF(1, 0, 0);
    STAGE_HALF(2, 0, 0);
    if (nk > 1) {
      STAGE_HALF(0, 1, 1);
      STAGE_HALF(1, 1, 1);
      STAGE_HALF(2, 1, 1);
      wait_vm<6>();
    } else {
      wait_vm<0>();
    }
    raw_barrier();
    if (STAGGER && wr == 1) raw_barrier();
    int s = 0, s2 = 2;  // stage of tile t, of tile t + 2
    for (int t = 0; t < nk; ++t) {
      const char* sb = smem + s * STG;
      const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
      // phase 1: quadrant (0, 0)
      READ_B(b0x, 0, sb)
      READ_A(0, sb)
      if (more2) STAGE_HALF(0, t + 2, s2);
      SEGMENT_END();
      MFMA_QUAD(0, 0, b0x)
      MFMA_END();
      // phase 2: quadrant (0, 1); retire tile t + 1
      READ_B(b1, 1, sb)
      if (more2) {
        STAGE_HALF(1, t + 2, s2);
        STAGE_HALF(2, t + 2, s2);
        wait_vm<6>();
      } else if (more1) {
        wait_vm<0>();
      }
      SEGMENT_END();
      MFMA_QUAD(0, 1, b1)
      MFMA_END();
      s = (s == 2) ? 0 : s + 1;
      s2 = (s2 == 2) ? 0 : s2 + 1;
    }
    if (STAGGER && wr == 0) raw_barrier();
  } else {
  // ---- prologue: all of tile 0, then A0 B0 B1 of tile 1 (its A1 is staged by phase 1 of tile 0)
  STAGE_HALF(0, 0, 0);
  STAGE_HALF(2, 0, 0);
  STAGE_HALF(3, 0, 0);
  STAGE_HALF(1, 0, 0);
  if (nk > 1) {
    STAGE_HALF(0, 1, 1);
    STAGE_HALF(2, 1, 1);
    STAGE_HALF(3, 1, 1);
    wait_vm<6>();
  } else {
    wait_vm<0>();
  }
  raw_barrier();
  // Every phase is two segments, [fragment reads + one half tile's DMA] | barrier | [16 MFMAs] | barrier.  The wr = 1 waves
  // run one barrier behind the wr = 0 waves (each SIMD hosts one wave of either group), so while one wave of a SIMD issues
  // its MFMAs the other one does its LDS reads, and the matrix pipe never waits for a read segment.  The hazard rules of
  // the header still hold with one barrier of slack less: a half tile is re-staged in the read segment after the one (two
  // barriers earlier for the same wave, one for the other group) in which it was last read, and tile t + 1 is waited for
  // in phase 4's first segment and read two barriers later.  Inside a read segment the fragment reads come FIRST: their
  // latency then runs under the ~120 cycles that the DMA's issue takes (the group's four waves queue their 1-KiB pieces
  // on the CU's one vector-memory path right after the barrier): -9 % cycles per K tile, profiles/EXPERIMENTS.md.
  if (STAGGER && wr == 1) raw_barrier();
  for (int t = 0; t < nk; ++t) {
    const int s = t & 1;
    const char* sb = smem + s * STAGE_BYTES;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    // phase 1: quadrant (0, 0)
    READ_B(b0x, 0, sb)
    READ_A(0, sb)
    if (more1) STAGE_HALF(1, t + 1, s ^ 1);  // A1 of the other stage: last read in phase 3 of tile t - 1
    SEGMENT_END();
    MFMA_QUAD(0, 0, b0x)
    MFMA_END();
    // phase 2: quadrant (0, 1)
    READ_B(b1, 1, sb)
    if (more2) STAGE_HALF(0, t + 2, s);  // A0: read in phase 1
    SEGMENT_END();
    MFMA_QUAD(0, 1, b1)
    MFMA_END();
    // phase 3: quadrant (1, 1)
    READ_A(1, sb)
    if (more2) STAGE_HALF(2, t + 2, s);  // B0: read in phase 1
    SEGMENT_END();
    MFMA_QUAD(1, 1, b1)
    MFMA_END();
    // phase 4: quadrant (1, 0); retire tile t + 1 (everything issued before the last three halves)
    if (more2) {
      STAGE_HALF(3, t + 2, s);  // B1: read in phase 2
      wait_vm<6>();
    } else {
      wait_vm<0>();
    }
    SEGMENT_END();
    MFMA_QUAD(1, 0, b0x)
    MFMA_END();
  }
  if (STAGGER && wr == 0) raw_barrier();
  }  // MH == 2
#undef STAGE_HALF
#undef READ_A
#undef READ_B
#undef LDS_FRAG
#undef MFMA_QUAD
#undef SEGMENT_END
#undef MFMA_END
#if G256_CLK
  if (blockIdx.x == 0 && tid == 0) {
    g256_clk[0] = __builtin_readcyclecounter() - clk_c0;
    g256_clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
#endif  // experiments

  if constexpr (EPI == GEMM_LOGPROB) {
    // Log-softmax pieces of this tile, nothing stored (transformer.py:235-242 + generate.py:101-118: the LM head's
    // bf16-rounded logits are only ever reduced to log-probabilities of given tokens).  A row's 256 columns live in the
    // four wc waves: 4 values per lane x 16 lanes per wave.  Per wave: lane-local then DPP row reductions; the four
    // waves meet in LDS (free again: every fragment read was waited for before the loop's last barrier).
    __syncthreads();
    float2* part = reinterpret_cast<float2*>(smem);  // [256 rows][4 waves]
#pragma unroll
    for (int ha = 0; ha < MH; ++ha)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = ha * 128 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
          const int row = min(row0 + rl, g.M - 1);
          float v[2][2], mx = -INFINITY;
#pragma unroll
          for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int n = n_tile * 256 + hb * 128 + wc * 32 + j * 16 + (lane & 15);
              v[hb][j] = (n < g.N) ? bf_round(acc[ha][hb][i][j][r]) : -INFINITY;
              mx = fmaxf(mx, v[hb][j]);
              if (n == g.lp_target[row] && row0 + rl < g.M) g.lp_tgt[row] = v[hb][j];
            }
          mx = row16_max(mx);
          float sm = 0.f;
#pragma unroll
          for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 2; ++j) sm += (v[hb][j] == -INFINITY) ? 0.f : __expf(v[hb][j] - mx);
          sm = row16_sum(sm);
          if ((lane & 15) == 0) part[rl * 4 + wc] = make_float2(mx, sm);
        }
    __syncthreads();
    if (tid < 256 && row0 + tid < g.M) {
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) M = fmaxf(M, part[tid * 4 + w].x);
      float S = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) S += (part[tid * 4 + w].x == -INFINITY) ? 0.f : part[tid * 4 + w].y * __expf(part[tid * 4 + w].x - M);
      g.lp_partial[(size_t)(row0 + tid) * n_tiles + n_tile] = make_float2(M, S);
    }
    return;
  }
  if constexpr (!kSwap) {
    // acc[ha][hb][i][j][r]: tile row ha*128 + wr*64 + i*16 + (lane>>4)*4 + r, tile column hb*128 + wc*32 + j*16 + (lane & 15)
#pragma unroll
    for (int ha = 0; ha < MH; ++ha)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = ha * 128 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
          if (rl >= rows_valid) continue;
          const int row = row0 + rl;
          float* o = reinterpret_cast<float*>(g.out) + (size_t)row * g.ldo;
#pragma unroll
          for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int n = n_tile * 256 + hb * 128 + wc * 32 + j * 16 + (lane & 15);
              if (n < g.N) o[n] = bf_round(acc[ha][hb][i][j][r]);
            }
        }
    return;
  }
  // ---- epilogue.  The MFMAs were issued as W-fragment x A-fragment, i.e. they produced the transposed 16x16 tiles:
  // acc[ha][hb][i][j][r] is tile row ha*128 + wr*64 + i*16 + (lane & 15), tile column hb*128 + wc*32 + j*16 +
  // (lane >> 4)*4 + r - four CONSECUTIVE output columns per lane, so results (and the residual) move as 8-byte
  // (bf16) / 16-byte (fp32 logits) accesses instead of 2-byte ones.
  const int cq = (lane >> 4) * 4;
  const bool rope = EPI == GEMM_STORE && g.rope_cs != nullptr;
  const bool wide_ok = (g.ldo & 3) == 0 && (reinterpret_cast<size_t>(g.out) & 15) == 0 &&
                       (EPI != GEMM_RESIDUAL || (reinterpret_cast<size_t>(g.residual) & 7) == 0);
#pragma unroll
  for (int ha = 0; ha < MH; ++ha)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = ha * 128 + wr * 64 + i * 16 + (lane & 15);
      if (rl >= rows_valid) continue;
      const int row = row0 + rl;
      const size_t ob = (size_t)row * g.ldo;
      const int tp = rope ? g.tok_pos[row] : 0;
      if (EPI == GEMM_SWIGLU) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n_tile * 128 + wc * 32 + j * 16 + cq;
          float y[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = swiglu_bf_fast(acc[ha][0][i][j][r], acc[ha][1][i][j][r]);
          bf16_t* o = reinterpret_cast<bf16_t*>(g.out) + ob + n;
          if (n + 3 < g.N && wide_ok) {
            *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]));
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < g.N) o[r] = f_to_bf(y[r]);
          }
        }
      } else {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int n = n_tile * 256 + hb * 128 + wc * 32 + j * 16 + cq;
            if (n >= g.N) continue;
            const bool full = (n + 3 < g.N) && wide_ok;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = bf_round(acc[ha][hb][i][j][r]);
            if (EPI == GEMM_STORE && rope && n < g.rope_cols) {
              // rotary embedding on the bf16-rounded projection (transformer_layers.py:66-70): this lane's four columns
              // are two (re, im) pairs of one head; their (cos, sin) entries are 16 contiguous bytes of the table
              const int i0 = ((n + g.rope_col0) % g.rope_dh) >> 1;
              const f32x4 cs = *reinterpret_cast<const f32x4*>(g.rope_cs + ((size_t)tp * (g.rope_dh >> 1) + i0) * 2);
              float re, im;
              rope_pair(y[0], y[1], cs[0], cs[1], re, im);
              y[0] = bf_round(re); y[1] = bf_round(im);
              rope_pair(y[2], y[3], cs[2], cs[3], re, im);
              y[2] = bf_round(re); y[3] = bf_round(im);
            }
            if (EPI == GEMM_LOGITS) {
              float* o = reinterpret_cast<float*>(g.out) + ob + n;
              if (full) {
                *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (n + r < g.N) o[r] = y[r];
              }
            } else {
              bf16_t* o = reinterpret_cast<bf16_t*>(g.out) + ob + n;
              if (full) {
                if (EPI == GEMM_RESIDUAL) {
                  const uint2 rs = *reinterpret_cast<const uint2*>(g.residual + ob + n);
                  y[0] += bf_lo(rs.x); y[1] += bf_hi(rs.x); y[2] += bf_lo(rs.y); y[3] += bf_hi(rs.y);
                }
                *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]));
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (n + r < g.N) o[r] = f_to_bf(EPI == GEMM_RESIDUAL ? bf_to_f(g.residual[ob + n + r]) + y[r] : y[r]);
              }
            }
          }
      }
    }
}

template <int EPI, bool STAGGER, int MH = 2>
hipError_t launch_var(const GemmArgs& g, dim3 grid, hipStream_t s) {
  constexpr int lds = MH == 2 ? LDS_BYTES : LDS_BYTES_H;
  static bool attr_set = false;  // > 64 KiB of dynamic LDS needs the opt-in once per kernel
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI, STAGGER, MH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<EPI, STAGGER, MH>), grid, dim3(512), lds, s, g);
  return hipGetLastError();
}
template <int EPI, int MH = 2>
hipError_t launch_one(const GemmArgs& g, dim3 grid, hipStream_t s) {
  static int stagger = -1;  // MI_GEMM_STAGGER=0: both wave groups in lockstep (A/B testing)
  if (stagger < 0) {
    const char* e = getenv("MI_GEMM_STAGGER");
    stagger = e ? atoi(e) : 1;
  }
  return stagger ? launch_var<EPI, true, MH>(g, grid, s) : launch_var<EPI, false, MH>(g, grid, s);
}

}  // namespace

bool gemm256_applicable(const GemmArgs& g) {
  if (g.K % BK != 0 || g.K < 2 * BK) return false;
  if (g.tile_tab != nullptr) return g.tile_rows == 256;  // token-grouped form: the tile table decides
  return g.a_gather == nullptr && g.M >= 256;
}

// 128 x 256 tiles (MH = 1) for the columns of a launch's partly filled last round: plain (not token-grouped) store /
// residual problems only.
bool gemm256_half_applicable(const GemmArgs& g) {
  return gemm256_applicable(g) && g.tile_tab == nullptr && (g.epi == GEMM_STORE || g.epi == GEMM_RESIDUAL);
}

hipError_t launch_gemm256_half(const GemmArgs& g, hipStream_t s) {
  const int m_tiles = (g.M + 127) >> 7, n_tiles = (g.N + 255) >> 8;
  const int supertiles = ((m_tiles + 3) >> 2) * ((n_tiles + 7) >> 3);
  const dim3 grid((unsigned)(((supertiles + 7) / 8) * 8 * 32));
  switch (g.epi) {
    case GEMM_STORE: return launch_one<GEMM_STORE, 1>(g, grid, s);
    case GEMM_RESIDUAL: return launch_one<GEMM_RESIDUAL, 1>(g, grid, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemm256(const GemmArgs& g, hipStream_t s) {
  const int nout = (g.epi == GEMM_SWIGLU) ? 128 : 256;
  const int m_tiles = g.tile_tab ? g.max_m_tiles : (g.M + 255) >> 8, n_tiles = (g.N + nout - 1) / nout;
  const int supertiles = ((m_tiles + 3) >> 2) * ((n_tiles + 7) >> 3);
  const dim3 grid((unsigned)(((supertiles + 7) / 8) * 8 * 32));
  switch (g.epi) {
    case GEMM_STORE: return launch_one<GEMM_STORE>(g, grid, s);
    case GEMM_RESIDUAL: return launch_one<GEMM_RESIDUAL>(g, grid, s);
    case GEMM_SWIGLU: return launch_one<GEMM_SWIGLU>(g, grid, s);
    case GEMM_LOGITS: return launch_one<GEMM_LOGITS>(g, grid, s);
    case GEMM_LOGPROB: return launch_one<GEMM_LOGPROB>(g, grid, s);
    default: return hipErrorInvalidValue;
  }
}
