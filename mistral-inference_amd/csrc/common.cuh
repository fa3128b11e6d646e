// Device-side helpers shared by the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 payload
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MI_WAVE 64

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_to_f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// fp32 -> bf16, round to nearest even (what torch's .to(bfloat16) does); NaN stays NaN.
__device__ __forceinline__ uint32_t f_to_bf_bits(float f) {
  uint32_t u = __float_as_uint(f);
  uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  r = (f != f) ? (u | 0x00400000u) : r;
  return r >> 16;
}
__device__ __forceinline__ bf16_t f_to_bf(float f) { return (bf16_t)f_to_bf_bits(f); }
// value after a round trip through bf16 (a rounding point of the reference)
__device__ __forceinline__ float bf_round(float f) { return __uint_as_float(f_to_bf_bits(f) << 16); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return f_to_bf_bits(lo) | (f_to_bf_bits(hi) << 16); }

// acc + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs, fp32 accumulate: ONE v_dot2c_f32_bf16 instead of four unpacks and
// two FMAs.  Every weight-streaming dot product (GEMV kernels and the persistent decode engine) goes through this helper
// in the same element order, which is what keeps the two paths bit-identical.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}

// K/V ring layouts (data layout in HBM is ours to choose; a kv head's head_dim elements are contiguous in both):
//   0  [max_batch, W, n_kv_heads, head_dim] - the reference's (cache.py:163-167); leaf operators called with a caller's own rings
//   1  [max_batch, n_kv_heads, W, head_dim] - head-major: the slots of ONE kv head are contiguous, so a decode split's keys are a
//      single run (4 slots = one 1-KiB DMA piece, 16 slots = a 4-KiB run: what the persistent engine's loader streams at the
//      weight rate; the strided form costs it 1.2 us per 16 KiB against 0.66 - profiles/EXPERIMENTS.md round 6).  BufferCache
//      allocates this form and exposes it as a permuted view of the reference's shape.
#ifndef MI_KV_SLOT_MAJOR
#define MI_KV_SLOT_MAJOR 0
#define MI_KV_HEAD_MAJOR 1
#endif
// element offset of (sequence, slot, column c = kv_head * head_dim + d) inside a ring of W slots
__host__ __device__ __forceinline__ size_t kv_offset(int layout, int W, int kv_dim, int Dh, size_t seq, int slot, int c) {
  return layout ? ((seq * (size_t)(kv_dim / Dh) + (size_t)(c / Dh)) * (size_t)W + (size_t)slot) * (size_t)Dh + (size_t)(c % Dh)
                : (seq * (size_t)W + (size_t)slot) * (size_t)kv_dim + (size_t)c;
}

// 16-byte loads.  `nt` marks streamed-once data (weights at decode): MI355X_MICROARCH "nt-weights".
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x4 ld16_nt(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// The weight / KV streams use ld16_nt directly and leave the s_waitcnt bookkeeping to hipcc.  With unconditional
// (clamped-address) loads, ping-pong register sets refilled in place and no register copies in the loop, the
// compiler's own wait placement keeps two batches in flight (checked in the ISA: the wait for one set is
// `vmcnt(<loads of the other set>)`).  A hand-counted inline-asm variant of these loads (cdna_hip_programming.md
// section 5.7) was tried and measured ~1 % faster end to end, and was removed: an asm load's destination registers are
// unprotected until the hand-written wait, and hipcc moved such registers while the data was in flight in three
// different places during development (long prologue between issue and wait; AGPR shuffling at 256 registers; the
// R = 2 decode-attention instance on rings wider than ~5000 slots) - silent, timing-dependent corruption.

// Sum over each aligned group of 16 lanes (a DPP "row"); every lane of the row gets the total.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);  // row_ror:8
  v += __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, true);  // row_ror:4
  v += __builtin_amdgcn_mov_dpp(v, 0x122, 0xf, 0xf, true);  // row_ror:2
  v += __builtin_amdgcn_mov_dpp(v, 0x121, 0xf, 0xf, true);  // row_ror:1
  return v;
}
// Sum over the 64 lanes of a wave; every lane gets the SAME total (lane 0's association order, broadcast).
// 4 DPP row rotations, then the two cross-row steps as gfx950 permlane swaps (v_permlane16_swap: rows 0|1 and 2|3,
// v_permlane32_swap: halves) - 8 VALU-rate instructions instead of six dependent ds_bpermute round trips.
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true));  // row_ror:8
  v = fmaxf(v, __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, true));  // row_ror:4
  v = fmaxf(v, __builtin_amdgcn_mov_dpp(v, 0x122, 0xf, 0xf, true));  // row_ror:2
  v = fmaxf(v, __builtin_amdgcn_mov_dpp(v, 0x121, 0xf, 0xf, true));  // row_ror:1
  return v;
}

// rope.py:13-23 on one (even, odd) pair: complex multiply with INDIVIDUALLY ROUNDED products.  HIP's __fmul_rn /
// __fsub_rn are plain operators, so under the default -ffp-contract=fast the compiler may or may not fuse
// `y0 * c - y1 * s` into an FMA depending on the surrounding code - two kernels then disagree in the last bit on
// near-cancelling pairs.  contract(off) pins the four products and the two sums everywhere this helper is inlined.
__device__ __forceinline__ void rope_pair(float y0, float y1, float c, float s, float& re, float& im) {
#pragma clang fp contract(off)
  const float a = y0 * c;
  const float b = y1 * s;
  const float d = y0 * s;
  const float e = y1 * c;
  re = a - b;
  im = d + e;
}

// Packed fp32 -> bf16 (RNE) in one instruction; gfx950 has no builtin for it (guide T12).
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// raw v_exp_f32 (2^x): no denormal range fix-up - callers pass x <= 0 where flushing tiny results to 0 is intended
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// The reference's SiLU runs on a bf16 tensor: silu evaluated in fp32, rounded to bf16; then the product
// with the bf16 up-projection is rounded again (transformer_layers.py:105-106).
__device__ __forceinline__ float swiglu_bf(float acc1, float acc3) {
  float a = bf_round(acc1), b = bf_round(acc3);
  float s = bf_round(a / (1.0f + expf(-a)));
  return s * b;  // caller rounds to bf16
}
// Same rounding points with hardware exp2 / rcp (relative error ~1e-6, three orders below a bf16 ulp): for the MFMA
// GEMM epilogue, where 64 results per lane made the IEEE expf + division chain (~40 instructions each) a visible
// part of every output tile.
__device__ __forceinline__ float swiglu_bf_fast(float acc1, float acc3) {
  const float a = bf_round(acc1), b = bf_round(acc3);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * a);
  const float s = bf_round(a * __builtin_amdgcn_rcpf(1.0f + e));
  return s * b;
}
