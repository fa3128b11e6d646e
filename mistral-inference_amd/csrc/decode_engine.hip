// Persistent decode engine: ONE launch runs every local layer of a batch-1 decode step (and the LM head).
//
// Replaces, for T = B = 1 on dense models, the 6 dependent launches per layer of the launch path (api.hip) - i.e. the
// whole of TransformerBlock.forward (transformer_layers.py:158-169) for every layer, CacheView.update (cache.py:83-92),
// the decode attention (transformer_layers.py:77-89) and the LM head (transformer.py:219,235).  Same arithmetic, same
// summation orders: the results are BIT-IDENTICAL to the launch path (tests/test_gpu_engine.py).
//
// Why: a decode layer is ~436 MB of weights streamed through six all-to-all dependencies; as separate launches each
// dependency costs a kernel boundary plus the ramp-up / drain of a 256-CU grid (~4.7 us).  Here
// the HBM stream never stops at a dependency (cdna_hip_programming.md section 5.6, MI355X_MICROARCH.md rows
// prefetch-credit / engine-vs-launches):
//
//   * grid = one workgroup per CU (8 waves): waves 0-3 are CONSUMERS, wave 4 is the LOADER, waves 5-7 are HOLDERS
//     (each keeps one W1|W3 unit of the layer in registers, fetched while the attention block runs: holder_units()).
//   * The loader walks a fixed per-CU program - this CU's row slab of Wq|Wk|Wv, its K/V ring slice, its rows of Wo,
//     W1|W3, W2 for every layer, then of the LM head - and copies it HBM -> LDS with `global_load_lds_dwordx4` (1 KiB
//     "pieces", non-temporal) into a ring of 8 x 16 KiB "fills".  Weights and old K/V do not depend on activations, so
//     the loader runs up to a full ring (~5 us of stream) ahead of the consumers across every dependency.
//   * Consumers take whole rows out of the ring (fp32 FMA chain per lane in the launch path's order, butterfly sum) and
//     run the epilogues (RMSNorm, RoPE, ring write, SwiGLU, residual).
//   * Dependencies between CUs are 8-byte {value, tag} granules: ONE write-through (sc1) store publishes 2 bf16 (or 1
//     fp32) together with its tag; consumers sweep the granule array with sc1 loads until every tag matches
//     (Guideline 16 R2: the data is the flag - no fences, no barriers, placement-independent).  tag = (step epoch << 12)
//     | (layer * 8 + edge + 1): unique per step and edge, so buffers are never re-initialised.  The epoch lives in
//     device memory and is bumped by the step's first kernel (decode_prep), so hipGraph replays see fresh tags.
//   * Every spin is bounded; on a timeout the block raises ctrl[1] (sticky) / ctrl[2] (per-step broadcast), all waits
//     fall through and the launch drains.  mi_engine_status() reads the sticky word.
//
// Work split per layer (NB = CUs): q/k/v/Wo/W2/LM-head rows in contiguous slabs of row PAIRS per CU; W1|W3 in slabs of
// unit pairs (rows w1[2j], w3[2j], w1[2j+1], w3[2j+1]); attention by (scheduled kv head, split) exactly as the
// stand-alone kernel's grid; split merge by slabs of output pairs.
#include <cstdlib>
#include <cstring>

#include "attn_decode_core.cuh"
#include "kernels.h"

// ENG_WIDE = 1: this file compiled a SECOND time (build_native.py -> decode_engine_wide.o) under other entry-point names, for
// the model shapes the shipped instantiations decline: GQA ratio 6 and a hid vector of 32 KiB (Mixtral-8x22B: dim 6144, 48 / 8
// heads, hidden 16384 - needs a 7-fill ring), rows that are not a multiple of 4 pieces (Mistral-Nemo: dim 5120, streamed as
// contiguous units instead of 2-piece groups).  Every difference sits behind `#if ENG_WIDE`, so the default compile of this
// file is token for token what it was: this kernel's speed moves by several per cent with ANY change of its code (see
// ENG_TRACE below), and the headline configuration must not pay for shapes it never runs.  scripts/engine_isa_hash.sh
// prints a hash of the default object's ISA; it has not changed since round 3.
#ifndef ENG_WIDE
#define ENG_WIDE 0
#endif
// ENG_WIDE = 2: the same additions with the shipped 8-fill ring, MoE models only (decode_engine_moe.o): at dim 4096 the four
// consumer waves' W1|W3 units span exactly 8 fills, and the 7-fill ring of ENG_WIDE = 1 costs Mixtral-8x7B 6 % of its W1|W3
// streaming rate (25.6 vs 27.2 GB/s per CU, profiles/r04_engine_trace_8x7b_*) - as much as the batched router saves.
// ENG_HEADLINE_ONLY = 2 (round 6, decode_engine_nemo.o: + ENG_WIDE = 2, ENG_SADDR = 2): only decode_engine_kernel<4, dense, rows NOT
// all multiples of 4 pieces> on the 8-fill ring - Mistral-Nemo (dim 5120 = rows of 10 pieces, streamed as contiguous units).
// ENG_SUFFIX (round 5): further compiles of this source under their own entry-point names - `_next` (decode_engine_next.o: the
// dense GQA-4 headline shape with the round-5 switches below: ENG_ABORT_RARE, ENG_CONS_PRIO, ENG_HOLD_STAGE, ENG_SADDR = 2 and
// ENG_TRACE = 0 - build_native.py: ENGINE_NEXT_FLAGS) and the `_x<N>` slots of an experiment library (scripts/build_variants.py
// engine_slots, scripts/engine_ab.py).  ENG_HEADLINE_ONLY = 1 instantiates only decode_engine_kernel<4, dense, all rows
// multiples of 4 pieces>.  Why the default object is kept frozen and what made builds of this file differ by up to 30 % in speed
// (hipcc's `s_waitcnt vmcnt(0)` in the loader's issue loop, ENG_SADDR below): DESIGN.md section 3, profiles/EXPERIMENTS.md round 5.
#ifndef ENG_HEADLINE_ONLY
#define ENG_HEADLINE_ONLY 0
#endif
#define ENG_CAT_(a, b) a##b
#define ENG_CAT(a, b) ENG_CAT_(a, b)
#if defined(ENG_SUFFIX)
#define ENG_NAME(x) ENG_CAT(x, ENG_SUFFIX)
#elif ENG_WIDE == 1
#define ENG_NAME(x) x##_wide
#elif ENG_WIDE == 2
#define ENG_NAME(x) x##_moe
#endif
#if ENG_WIDE || defined(ENG_SUFFIX)
#define launch_decode_engine ENG_NAME(launch_decode_engine)
#define decode_engine_applicable ENG_NAME(decode_engine_applicable)
#define decode_engine_granule_bytes ENG_NAME(decode_engine_granule_bytes)
#define decode_engine_set_holders ENG_NAME(decode_engine_set_holders)
#define decode_engine_set_knobs ENG_NAME(decode_engine_set_knobs)
#define decode_engine_set_trace ENG_NAME(decode_engine_set_trace)
#define decode_engine_trace_bytes ENG_NAME(decode_engine_trace_bytes)
#define decode_engine_census_detail ENG_NAME(decode_engine_census_detail)
#define decode_engine_forget_census ENG_NAME(decode_engine_forget_census)
#define engine_census_kernel ENG_NAME(engine_census_kernel)
#endif

namespace {

using namespace attn_core;

constexpr int NCONS = 4;
#ifndef ENG_HOLDERS
#define ENG_HOLDERS 3  // holder waves per workgroup (0: none)
#endif
// (Experiments that measured slower or within noise - sparse re-polls, lean barriers, a flag barrier, the consumer marks as one
// 16-byte line, cached marks, per-site sleep lengths, 8-piece fills, the stamp-site bisect mask, a finer holder check - were
// removed from this file in round 6: scripts/probes/decode_engine_experiments.patch restores them; profiles/EXPERIMENTS.md has
// every number.)
// ENG_TRACE = 1 (default, wide and MoE builds): the phase-timeline stamp sites (mi_debug_set_engine_trace, scripts/engine_trace.py)
// stay in the kernel although they cost a test of a null pointer each.  MEASURED: compiling them out makes THOSE builds 14-19 %
// SLOWER (profiles/EXPERIMENTS.md rounds 3-5): without the sites hipcc places `s_waitcnt vmcnt(0)` at the top of the loader's
// per-unit loops (in front of the rewrite of a DMA's 64-bit VGPR address pair) - a drain of the DMA queue per unit.  With the
// DMAs out of hipcc's sight (ENG_SADDR = 2) the build without the sites is as fast as the one with them: the `next` build ships
// ENG_TRACE = 0.  ENG_TRACE = 2 replaces the stamps by bare compiler barriers (experiment).
#ifndef ENG_TRACE
#define ENG_TRACE 1
#endif
// ENG_ALL4: instantiations whose weight rows are all multiples of 4 pieces (dim, n_heads*128 and hidden_dim multiples of
// 2048 - every BASELINE model but Nemo) drop the generic-group path from the five row loops of the consumers.
#ifndef ENG_ALL4
#define ENG_ALL4 1
#endif
#ifndef ENG_ABORT_RARE
#define ENG_ABORT_RARE 0  // 1: the abort word is read on every 1024th iteration of a spin only (one LDS read less per poll)
#endif
#ifndef ENG_CONS_PRIO
#define ENG_CONS_PRIO 0   // s_setprio of the consumer waves (the loader runs at 3, holders at 0)
#endif
// ENG_QKV_HOLD = 1 (MoE models; their holder waves have no W1|W3 unit to keep - which experts stream is decided late): the
// three holder waves keep the LAST SIX q|k|v row-pair units of the NEXT layer in registers (two units = 4 rows = 128 VGPRs
// each), fetched in the one window of a MoE layer in which HBM idles - the router bubble: the loader has flushed behind
// the Wo rows and waits for the expert decision (~6 us, profiles/r04_engine_trace_8x7b_*).  When attention_norm(h) of that
// layer stands in LDS they reduce their rows from registers (the arithmetic of Cons::unit_dot<2>: bit-identical) and run
// the consumers' epilogue (RoPE, ring write, granule).  96 KiB per workgroup and layer that no longer pass through the ring
// in the loader-bound q|k|v phase.
#ifndef ENG_QKV_HOLD
#define ENG_QKV_HOLD 0
#endif
// ENG_SADDR = 1: the loader's weight DMAs in the SGPR-base form (`global_load_lds_dwordx4 v_lane_offset, s[base:base+1]`): the
// per-unit / per-group address arithmetic becomes scalar and no VGPR that an in-flight DMA names is ever rewritten.  (With
// 64-bit VGPR addresses hipcc guards every rewrite of the address pair with `s_waitcnt vmcnt(0)` - it treats the pair as
// the destination of a load - which drains the DMA queue once per unit in some builds and not in others: one of the
// mechanisms behind this kernel's "regimes", profiles/EXPERIMENTS.md round 5.)
#ifndef ENG_SADDR
#define ENG_SADDR 0
#endif
// ENG_NOSTOP (bit mask: 1 h, 2 q, 4 split merge, 8 attn, 16 h1, 32 hid): sweeps during which this workgroup's loader is NOT stopped
#ifndef ENG_NOSTOP
#define ENG_NOSTOP 0
#endif
#define GATHER_FLAG(bit) ((ENG_NOSTOP & (bit)) ? 0u : 1u)
// ENG_CLEAN_ENTRY = 1: run_loader begins with a wait-count instruction that hipcc models (see there; round 6)
#ifndef ENG_CLEAN_ENTRY
#define ENG_CLEAN_ENTRY 0
#endif
#ifndef ENG_HOLD_STAGE
#define ENG_HOLD_STAGE 3  // the holders' fetch of a layer's units may begin when the loader has issued: 0 nothing yet, 1 q|k|v, 2 + K/V, 3 + Wo
#endif
constexpr int NHOLD = ENG_HOLDERS;
constexpr int NTHREADS = (NCONS + 1 + NHOLD) * 64;
constexpr int PIECE = 1024;          // bytes per DMA instruction: 64 lanes x 16 B
constexpr int FILL = 16;             // pieces per fill
#if ENG_WIDE == 1
constexpr int RING_FILLS = 7;        // 112 KiB ring: 47 KiB left for the activation region (a 32 KiB hid vector fits)
#define RING_IDX(sh, x) ((uint32_t)(x) % (uint32_t)(RING_FILLS * FILL))  // not a power of two: a constant modulo (scalar ALU)
#else
constexpr int RING_FILLS = 128 / FILL;  // 128 KiB ring
#define RING_IDX(sh, x) ((x) & (sh).ring_mask)
#endif
constexpr int LDS_TOTAL = 160 * 1024;
constexpr int CTL_BYTES = 256;
constexpr int RES_BYTES = 1024;      // this CU's residual rows (bf16), <= 512 rows
constexpr int XS_OFF = CTL_BYTES + RES_BYTES;
constexpr uint32_t SPIN_LIMIT = 1u << 22;

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;
// Every LDS access goes through an explicit address_space(3) pointer: a generic pointer makes hipcc emit FLAT loads,
// which count in vmcnt (the loader's DMA bookkeeping) and are waited for with vmcnt(0).
#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS char lchar;
typedef LDS_AS uint32_t lu32;
typedef LDS_AS volatile uint32_t lvu32;
typedef LDS_AS float lf32;
typedef LDS_AS volatile float lvf32;
typedef LDS_AS u32x4 lu32x4;
typedef LDS_AS bf16_t lbf16;

// control words at the start of the LDS
enum : int {
  C_LANDED = 0,     // fills completely in LDS (loader -> consumers)
  C_DONE = 1,       // [NCONS] first piece index each consumer wave may still read (consumers -> loader)
  C_CBAR = 5,       // consumer-wave barrier counter
  C_GATHERING = 6,  // consumers are sweeping granules: the loader keeps one fill outstanding
  C_ABORT = 7,
  C_RED = 8,        // [4] fp32 wave totals of the RMSNorm
  C_LSTAGE = 12,    // loader progress: (layer + 1) once this layer's Wo rows are issued (loader -> holders)
  C_XREADY = 13,    // (layer + 1) once ffn_norm(h1) of that layer stands in the activation region (consumers -> holders)
  C_HDONE = 14,     // W1|W3 units finished by holder waves since the launch began (holders -> consumers)
  C_ARRIVED = 15,   // 1 once every workgroup of the launch is known to be resident (consumer wave 0 -> the other waves)
  C_XA = 21,        // ENG_QKV_HOLD: (layer + 1) once attention_norm(h) of that layer stands in the activation region (-> holders)
  C_HGO = 23,       // ENG_QKV_HOLD = 2: (layer + 1) once this workgroup has h1 of that layer (the router runs next: no sweep for a while)
  C_HQDONE = 22,    // ENG_QKV_HOLD: holder waves done with their q|k|v units since the launch began (-> consumers)
  C_EXPERT = 20,    // MoE: ((layer + 1) << 16) | expert A | expert B << 8 (ascending ids) once the router has decided (consumers -> loader)
  C_RLOGIT = 24     // [16] MoE: bf16-rounded router logits of the layer (fp32 words)
};
// global control words (workspace): [0] step epoch, [1] sticky status, [2] abort broadcast, [3] bad token id, [4] engine
// launches completed, [5] decode steps committed (index into the greedy history ring), [6] workgroup arrivals
// [7] test hook: engine launches that shall fail their residency gate (mi_debug_engine_sabotage)
enum : int { G_EPOCH = 0, G_STATUS = 1, G_ABORT = 2, G_BADID = 3, G_LAUNCHES = 4, G_STEPS = 5, G_ARRIVE = 6, G_SABOTAGE = 7 };
constexpr uint32_t ARRIVE_POLLS = 1u << 16;  // ~50 ms: far beyond the ~1 us over which a resident grid starts

// Optional timeline (mi_debug_set_engine_trace): trace[c][layer][event] = 100 MHz wall clock.  Consumer wave 0 writes
// events [0, TR_CONS), the loader events [TR_CONS, TR_EVENTS).  The buffer is reached through an explicit
// global-address-space pointer held in `Shared`: a store through a generic pointer loaded from the by-value argument
// struct makes hipcc spill the whole struct to scratch.
constexpr int TR_CONS = 18, TR_EVENTS = 26;

struct Shared {
  lvu32* ctl;
  lbf16* res;   // residual rows of this CU
  lchar* xs;    // activation vector / attention scratch
  lchar* ring;
  uint32_t ring_mask;  // ring pieces - 1
  gu32* ctrl;          // global control words (G_* below)
  gu64* trace;         // optional timeline buffer
};

__device__ __forceinline__ void trace_ev(const Shared& sh, int c, int layer, int ev, bool who) {
#if ENG_TRACE == 1
  if (sh.trace && who) sh.trace[((size_t)c * ENG_MAXL + layer) * TR_EVENTS + ev] = __builtin_amdgcn_s_memrealtime();
#elif ENG_TRACE == 2
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ void raise_abort(const Shared& sh, uint32_t code) {
  sh.ctl[C_ABORT] = 1;
  __hip_atomic_store(sh.ctrl + 2, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t expected = 0;
  __hip_atomic_compare_exchange_strong(sh.ctrl + 1, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
}

// One iteration of a bounded spin.  Returns false when the wait must be abandoned.
__device__ __forceinline__ bool spin_ok(const Shared& sh, uint32_t& spins, uint32_t code) {
  __builtin_amdgcn_s_sleep(1);  // (64 clocks; 0 / 2 / 4 / 8 per kind of wait all measured within +-0.2 %)
  if (!ENG_ABORT_RARE && sh.ctl[C_ABORT]) return false;
  ++spins;
  if ((spins & 1023u) == 0) {
    if (ENG_ABORT_RARE && sh.ctl[C_ABORT]) return false;
    if (__hip_atomic_load(sh.ctrl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
      sh.ctl[C_ABORT] = 1;
      return false;
    }
    if (spins >= SPIN_LIMIT) {
      raise_abort(sh, code);
      return false;
    }
  }
  return true;
}

__device__ __forceinline__ u32x4 lds16(const LDS_AS void* p) { return *reinterpret_cast<const lu32x4*>(p); }
__device__ __forceinline__ void lds_st16(LDS_AS void* p, u32x4 v) { *reinterpret_cast<lu32x4*>(p) = v; }

// ------------------------------------------------------------------------------------------------ work split
__device__ __forceinline__ void slab(int units, int c, int nb, int& u0, int& u1) {
  u0 = (int)((long)units * c / nb);
  u1 = (int)((long)units * (c + 1) / nb);
}

struct LayerPlan {  // this CU's share of one layer, in units and pieces
  int q0, q1, k0, k1, v0, v1;  // row-pair slabs of Wq, Wk, Wv
  int o0, o1;                  // row-pair slab of Wo and W2 (outputs of dim D)
  int f0, f1;                  // unit-pair slab of W1|W3
  int e0, e1;                  // output-pair slab of the split merge
  int att;                     // 1: this CU owns an attention work item
  int kvh, split, s_begin, s_end, n_att;  // scheduled kv head, split, slot range, K (= V) pieces
  int cur_slot;
};

__device__ __forceinline__ void plan_layer(const EngArgs& a, const EngLayer& L, int c, int pos, LayerPlan& p) {
  slab(a.H * DH / 2, c, a.NB, p.q0, p.q1);
  slab(a.Hkv * DH / 2, c, a.NB, p.k0, p.k1);
  p.v0 = p.k0;
  p.v1 = p.k1;
  slab(a.D / 2, c, a.NB, p.o0, p.o1);
  slab(a.F / 2, c, a.NB, p.f0, p.f1);
  slab(a.H * DH / 2, c, a.NB, p.e0, p.e1);
  const int kv_len = min(pos + 1, L.W);
  p.cur_slot = pos % L.W;
  p.att = c < a.Hs * L.n_splits;
  p.kvh = c % a.Hs;
  p.split = c / a.Hs;
  p.s_begin = p.split * L.chunk;
  p.s_end = min(p.s_begin + L.chunk, kv_len);
  p.n_att = (p.att && p.s_end > p.s_begin) ? (p.s_end - p.s_begin + 3) >> 2 : 0;
}

// pieces per group of a unit's interleaved stream (loader and consumers must agree)
__device__ __forceinline__ int unit_group(int P) { return (P & 3) == 0 ? 4 : ((P & 1) == 0 ? 2 : 1); }

// K/V pieces of a workgroup's attention work item.  Ring layout MI_KV_HEAD_MAJOR (common.cuh): the slots of a kv head are
// contiguous, a piece (4 slots x 256 B) is ONE KiB of memory and 4 pieces a 4-KiB run - streamed like a weight row, four pieces
// per address computation, as [K j .. j+3][V j .. j+3] (kv_runs): 0.66 us per 16 KiB instead of 1.2 for single strided pieces
// (the K/V issue sat on the attention block's critical path: profiles/EXPERIMENTS.md round 6).  Otherwise (the reference's
// layout, a split that is not whole groups of 16 slots, a stream position that is not a multiple of 4): [K j][V j] single pieces.
__device__ __forceinline__ bool kv_runs(const EngLayer& L, const LayerPlan& p, uint32_t g) {
  return L.kv_layout == MI_KV_HEAD_MAJOR && p.n_att > 0 && (p.n_att & 3) == 0 && (g & 3u) == 0 && p.s_begin + 4 * p.n_att <= L.W;
}
// ring piece (relative to the first K/V piece) of K piece j / V piece j
__device__ __forceinline__ uint32_t kv_piece_k(bool runs, int j) { return runs ? (uint32_t)(8 * (j >> 2) + (j & 3)) : (uint32_t)(2 * j); }
__device__ __forceinline__ uint32_t kv_piece_v(bool runs, int j) { return runs ? (uint32_t)(8 * (j >> 2) + (j & 3) + 4) : (uint32_t)(2 * j + 1); }

// HOLDER waves: the last holder_units() W1|W3 units of a CU's slab never pass through the ring.  A holder wave fetches
// its unit (4 rows x D bf16 = 32 KiB at D = 4096) straight into 128 of its VGPRs while the attention block of the layer
// keeps the consumers busy with hand-offs and the ring is full, and reduces it from registers when ffn_norm(h1) arrives:
// 96 KiB per CU and layer that the loader no longer has to squeeze through the HBM-bound W1|W3 phase.
constexpr int HOLD_GROUPS = 2;  // 4-piece groups per row a holder can keep: D <= 4096
__device__ __forceinline__ int holder_units(const EngArgs& a, int n_f) {
  const int P = a.D >> 9;
  return (NHOLD > 0 && a.holders && a.E == 0 && unit_group(P) == 4 && (P >> 2) <= HOLD_GROUPS && n_f >= 2 * NCONS + NHOLD) ? NHOLD : 0;
}

#if ENG_QKV_HOLD
// q|k|v units (row pairs) of a workgroup that the holder waves keep in registers: the last 2 * NHOLD of the list q.., k.., v..
__device__ __forceinline__ int qkv_held(const EngArgs& a, int n_u) {
  const int P = a.D >> 9;
#if ENG_WIDE == 1
  // rows of 9-12 pieces (Mixtral-8x22B: dim 6144): a holder keeps ONE unit (2 rows x 12 pieces = 96 VGPRs): run_qkv_holder1
  if (NHOLD > 0 && a.holders && a.E > 0 && unit_group(P) == 4 && (P >> 2) == HOLD_GROUPS + 1 && n_u >= NCONS + NHOLD) return NHOLD;
#endif
  return (NHOLD > 0 && a.holders && a.E > 0 && unit_group(P) == 4 && (P >> 2) <= HOLD_GROUPS && n_u >= NCONS + 2 * NHOLD) ? 2 * NHOLD : 0;
}
#endif
// list index k of a workgroup's q|k|v units -> kind (0 q, 1 k, 2 v) and global row pair inside that matrix
__device__ __forceinline__ void qkv_unit(const LayerPlan& p, int k, int& kind, int& u) {
  const int nq_u = p.q1 - p.q0, nk_u = p.k1 - p.k0;
  if (k < nq_u) { kind = 0; u = p.q0 + k; }
  else if (k < nq_u + nk_u) { kind = 1; u = p.k0 + (k - nq_u); }
  else { kind = 2; u = p.v0 + (k - nq_u - nk_u); }
}

// ------------------------------------------------------------------------------------------------ loader wave
struct Loader {
  const Shared& sh;
  int lane, ring_fills, thin, depth;
  uint32_t g = 0;    // pieces issued
  uint32_t pub = 0;  // fills published
  uint32_t stalls = 0;  // fills that had to wait for a free ring slot (trace only)
#if ENG_SADDR
  uint32_t lane16 = 0;  // this lane's byte offset inside a piece (set by run_loader)
#endif

  __device__ __forceinline__ uint32_t min_done() const {
    return min(min(sh.ctl[C_DONE + 0], sh.ctl[C_DONE + 1]), min(sh.ctl[C_DONE + 2], sh.ctl[C_DONE + 3]));
  }
  __device__ __forceinline__ void publish(uint32_t fills) {
    if (fills > pub) {
      pub = fills;
      sh.ctl[C_LANDED] = fills;
    }
  }
  // Before the first piece of a fill: its ring slot must have been consumed completely.
  __device__ __forceinline__ void fill_begin() {
    const uint32_t f = g / FILL;
    if (f >= (uint32_t)ring_fills) {
      const uint32_t need = (f - ring_fills + 1) * FILL;
      if (min_done() < need) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(f);  // everything issued has landed: consumers must not starve while we wait for them
        ++stalls;
        uint32_t spins = 0;
        while (min_done() < need)
          if (!spin_ok(sh, spins, 0x100)) break;
      }
    }
  }
  // After the last piece of a fill: retire older fills (DMA completes in order; 16 instructions per fill).
  __device__ __forceinline__ void fill_end() {
    const uint32_t f = g / FILL;  // fills issued so far
    if (thin >= 2 && sh.ctl[C_GATHERING]) {
      // STOP the stream while this CU's consumers sweep granules: the sweep's loads queue in the CU's vector memory
      // pipeline behind whatever the loader has in flight, and the hand-off is the critical path - the ring is full
      // for most of that wait anyway (2.70-2.75 ms against 2.74-2.79 ms with one fill kept in flight, same box)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      publish(f);
      uint32_t spins = 0;
      while (sh.ctl[C_GATHERING])
        if (!spin_ok(sh, spins, 0x100)) break;
    } else if (thin && sh.ctl[C_GATHERING]) {  // thin == 1: keep one fill in flight during sweeps (A/B)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      publish(f - 16 / FILL);
    } else if (depth >= 3) {
      asm volatile("s_waitcnt vmcnt(47)" ::: "memory");  // <= 3 fills in flight (the counter saturates at 63)
      if (f >= 48 / FILL) publish(f - 48 / FILL);
    } else {
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      if (f >= 32 / FILL) publish(f - 32 / FILL);
    }
  }
  // The DMA itself: hipcc's builtin, or (ENG_SADDR >= 2) inline asm (cdna_hip_programming.md section 5.7 recipe: M0 is written in
  // the statement that reads it and restored), which keeps the DMA out of hipcc's s_waitcnt bookkeeping.
  template <int N>
  __device__ __forceinline__ void dma_n(const void* src_lane, lchar* dst) {
#if ENG_SADDR >= 2
    // ENG_SADDR >= 2: NO builtin LDS-DMA is left in the loader.  The pieces that still carry per-lane addresses - K/V pieces of
    // rings in the reference's layout and of splits that are not whole 16-slot groups (kv_runs) - are rare at the shapes this
    // build serves, but as builtins they are what hipcc's wait-count pass tracks: it guards every LDS read and every rewrite
    // of an address register that MAY follow one with `s_waitcnt vmcnt(0)`, e.g. in fill_begin - a drain of the DMA queue per
    // fill, +30 % per step (scripts/engine_loader_waits.py finds such waits statically; tests/test_engine_build.py).
    unsigned keep;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<size_t>(dst);
    if constexpr (N == 1)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off nt\n\t"
                   "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src_lane), "s"(lds_addr) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off nt\n\t"
                   "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                   "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
                   "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
                   "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src_lane), "s"(lds_addr) : "memory");
#else
    dma<0>(src_lane, dst);
    if constexpr (N == 4) {
      dma<PIECE>(src_lane, dst);
      dma<2 * PIECE>(src_lane, dst);
      dma<3 * PIECE>(src_lane, dst);
    }
#endif
  }
  template <int OFF>
  __device__ __forceinline__ void dma(const void* src_lane, lchar* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lane, (LDS_AS void*)dst, 16, OFF, 2 /* nt */);
  }
#if ENG_SADDR
  // wave-uniform base (SGPR pair) + this lane's 32-bit byte offset (one VGPR, written once per launch)
  template <int OFF>
  __device__ __forceinline__ void dma_s(const char* sbase, lchar* dst) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(sbase);  // (readfirstlane: an opaque, provably uniform pair)
    const unsigned long long bu = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const __attribute__((address_space(1))) char* gp = (const __attribute__((address_space(1))) char*)bu + (uint32_t)(lane * 16);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (LDS_AS void*)dst, 16, OFF, 2 /* nt */);
  }
  __device__ __forceinline__ void piece_s(const char* sbase) {
    if ((g & (FILL - 1)) == 0) fill_begin();
#if ENG_SADDR >= 2
    unsigned keep;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<size_t>(slot_of(g));
    const unsigned long long b = reinterpret_cast<unsigned long long>(sbase);
    const unsigned long long bu = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                 "global_load_lds_dwordx4 %1, %2 nt\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane16), "s"(bu), "s"(lds_addr) : "memory");
#else
    dma_s<0>(sbase, slot_of(g));
#endif
    ++g;
    if ((g & (FILL - 1)) == 0) fill_end();
  }
  __device__ __forceinline__ void piece4_s(const char* sbase) {
    if ((g & (FILL - 1)) == 0) fill_begin();
    lchar* dst = slot_of(g);
#if ENG_SADDR == 2
    // (hipcc's lowering of the builtin keeps the 64-bit VGPR address even for an SGPR base + zext(VGPR) sum: the SGPR-base
    // form is written out.  M0 carries the LDS address; it is written in the statement that reads it and restored
    // (cdna_hip_programming.md section 5.7).  hipcc does not count these loads: every wait for them is the explicit
    // vmcnt(N) of fill_begin / fill_end / flush, which assume nothing else.)
    unsigned keep;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<size_t>(dst);
    const unsigned long long b = reinterpret_cast<unsigned long long>(sbase);
    const unsigned long long bu = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    // (s_nop 4: hipcc does not pad hazards for the operands of an asm statement - should the base pair ever come straight
    // from a v_readfirstlane, a VMEM read of a VALU-written SGPR needs 5 wait states; the M0 write needs 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                 "global_load_lds_dwordx4 %1, %2 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane16), "s"(bu), "s"(lds_addr) : "memory");
#else
    dma_s<0>(sbase, dst);
    dma_s<PIECE>(sbase, dst);
    dma_s<2 * PIECE>(sbase, dst);
    dma_s<3 * PIECE>(sbase, dst);
#endif
    g += 4;
    if ((g & (FILL - 1)) == 0) fill_end();
  }
#endif
  __device__ __forceinline__ lchar* slot_of(uint32_t piece_idx) {  // wave-uniform (becomes M0)
    return sh.ring + (uint32_t)__builtin_amdgcn_readfirstlane((int)RING_IDX(sh, piece_idx)) * PIECE;
  }
  __device__ __forceinline__ void piece(const void* src_lane) {
    if ((g & (FILL - 1)) == 0) fill_begin();
    dma_n<1>(src_lane, slot_of(g));
    ++g;
    if ((g & (FILL - 1)) == 0) fill_end();
  }
  // 4 consecutive pieces (4 KiB contiguous in memory AND in the ring: g % 4 == 0): one address, immediate offsets
  __device__ __forceinline__ void piece4(const void* src_lane) {
    if ((g & (FILL - 1)) == 0) fill_begin();
    dma_n<4>(src_lane, slot_of(g));
    g += 4;
    if ((g & (FILL - 1)) == 0) fill_end();
  }
  // One UNIT = NR weight rows of P pieces each that a consumer wave reduces together.  Stream order inside a unit: groups
  // of G pieces, row after row - rows[0][0..G), rows[1][0..G), ..., rows[0][G..2G), ... - so that the consumer can start on
  // the first group while the rest is in flight and hand ring space back group by group (a unit of W2 is 56 pieces: four
  // waves each pinning a whole unit would need 14 fills of the 8-fill ring).
#if ENG_WIDE
  // n pieces that are contiguous in memory (and land contiguously in the ring): four per address computation wherever the
  // stream position allows it
  __device__ __forceinline__ void seg(const bf16_t* base, int n) {
#if ENG_SADDR
    const char* sb = reinterpret_cast<const char*>(base);  // (wave-uniform base; the lane's offset is in lane16)
    int i = 0;
    while (i < n) {
      if ((g & 3) == 0 && n - i >= 4) {
        piece4_s(sb + (size_t)i * PIECE);
        i += 4;
      } else {
        piece_s(sb + (size_t)i * PIECE);
        ++i;
      }
    }
#else
    const char* src = reinterpret_cast<const char*>(base) + lane * 16;
    int i = 0;
    while (i < n) {
      if ((g & 3) == 0 && n - i >= 4) {
        piece4(src + (size_t)i * PIECE);
        i += 4;
      } else {
        piece(src + (size_t)i * PIECE);
        ++i;
      }
    }
#endif
  }
#endif
  template <int NR>
  __device__ __forceinline__ void unit(const bf16_t* const (&rp)[NR], int P) {
    const int G = unit_group(P);
#if ENG_WIDE
    // Rows that are not a multiple of 4 pieces (Mistral-Nemo: dim 5120 = 10 pieces).  The shipped form streams such rows in
    // 2-piece groups, piece by piece, and the loader's cost is per GROUP (~200 cycles of address / slot / fill bookkeeping,
    // profiles/EXPERIMENTS.md): 13.5 GB/s per CU instead of 28.  But the two rows of a pair ARE contiguous in memory
    // (rows 2u, 2u + 1 of a row-major matrix; rows 2j, 2j + 1 of W1 and of W3 for a gate/up unit): a unit is one (or two)
    // contiguous runs of 2P pieces, streamed four pieces per address computation like every other row, laid out in the
    // ring linearly - row r of a pair at unit + r * P.  The consumer waits for the whole unit (20 or 40 KiB: < 1.5 us of
    // stream, hidden behind the previous unit's reduction) instead of group by group.
    if (G != 4) {
      seg(rp[0], 2 * P);                      // NR == 2: rows (2u, 2u + 1); NR == 4: W1 rows (2j, 2j + 1)
      if constexpr (NR == 4) seg(rp[1], 2 * P);  // W3 rows (2j, 2j + 1)
      return;
    }
#endif
    for (int p0 = 0; p0 < P; p0 += G) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#if ENG_SADDR
        const char* sb = reinterpret_cast<const char*>(rp[r]) + (size_t)p0 * PIECE;
        if (G == 4 && (g & 3) == 0) {
          piece4_s(sb);
        } else {
          for (int i = 0; i < G; ++i) piece_s(sb + (size_t)i * PIECE);
        }
#else
        const char* src = reinterpret_cast<const char*>(rp[r]) + (size_t)p0 * PIECE + lane * 16;
        if (G == 4 && (g & 3) == 0) {
          piece4(src);
        } else {
          for (int i = 0; i < G; ++i) piece(src + (size_t)i * PIECE);
        }
#endif
      }
    }
  }
  // row-pair units u0 .. u1 of a [rows, K] matrix
  __device__ __forceinline__ void pairs(const bf16_t* base, int u0, int u1, int K) {
    for (int u = u0; u < u1; ++u) {
      const bf16_t* const rp[2] = {base + (size_t)(2 * u) * K, base + (size_t)(2 * u + 1) * K};
      unit<2>(rp, K >> 9);
    }
  }
  __device__ __forceinline__ void flush() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish((g + FILL - 1) / FILL);
  }
};

template <bool MOE>
__device__ __forceinline__ void run_loader(const EngArgs& a, const Shared& sh, int c, int lane, int pos, int seq) {
  Loader ld{sh, lane, a.ring_fills, a.thin, a.depth};
#if ENG_SADDR
  ld.lane16 = (uint32_t)lane * 16u;
#endif
#if ENG_CLEAN_ENTRY
  // hipcc structurizes the role split of the kernel into a chain of `Flow` blocks, so that - statically - the loader's code is
  // reachable from the holders' and the consumers' code, and its wait-count pass carries THEIR outstanding register loads into
  // this function: wherever the loader first writes a VGPR that such a load names, it places `s_waitcnt vmcnt(N)` - inside
  // the loader's issue loops, where the same instruction then drains the DMA queue on every pass (N = 4: +30 % per step; which
  // registers collide changes with every edit: the "regimes" of rounds 3-5, profiles/EXPERIMENTS.md round 6).  One wait that the
  // pass understands, executed once at the role's entry, empties its scoreboard.  (Measured 0.3-0.4 % slower than a build whose
  // loader happens to be clean without it - scripts/engine_loader_waits.py checks that statically - so the shipped builds leave
  // it off and are held to the check instead: tests/test_engine_build.py.)
  __builtin_amdgcn_s_waitcnt(0);
#endif
  __builtin_amdgcn_s_setprio(3);  // the loader shares a SIMD with one consumer wave: its few instructions go first
  const int PD = a.D >> 9;
  for (int l = 0; l < a.n_layers; ++l) {
    const EngLayer& L = a.L[l];
    if (sh.ctl[C_ABORT]) break;  // the consumers have given up (residency gate / a timed-out wait): nothing left to feed
    LayerPlan p;
    plan_layer(a, L, c, pos, p);
    const bool tr = lane == 0;
    trace_ev(sh, c, l, TR_CONS + 0, tr);
    if (NHOLD && ENG_HOLD_STAGE == 0) sh.ctl[C_LSTAGE] = (uint32_t)(l + 1);
#if ENG_QKV_HOLD
    if (const int n_held = qkv_held(a, (p.q1 - p.q0) + 2 * (p.k1 - p.k0))) {  // the list without its last n_held units
      const int n_s = (p.q1 - p.q0) + 2 * (p.k1 - p.k0) - n_held;
      for (int k = 0; k < n_s; ++k) {
        int kind, u;
        qkv_unit(p, k, kind, u);
        const bf16_t* base = kind == 0 ? L.wq : (kind == 1 ? L.wk : L.wv);
        ld.pairs(base, u, u + 1, a.D);
      }
    } else
#endif
    {
      ld.pairs(L.wq, p.q0, p.q1, a.D);
      ld.pairs(L.wk, p.k0, p.k1, a.D);
      ld.pairs(L.wv, p.v0, p.v1, a.D);
    }
    trace_ev(sh, c, l, TR_CONS + 1, tr);
    if (NHOLD && ENG_HOLD_STAGE == 1) sh.ctl[C_LSTAGE] = (uint32_t)(l + 1);
    if (p.n_att) {  // K piece j, V piece j: 4 ring slots x 256 B each (slots past the ring end are clamped; masked later)
      const int kv_real = p.kvh / a.kv_groups;
      const size_t row_stride = L.kv_layout ? (size_t)DH : (size_t)a.Hkv * DH;  // elements between consecutive slots of this kv head
      const size_t ring0 = kv_offset(L.kv_layout, L.W, a.Hkv * DH, DH, (size_t)seq, 0, kv_real * DH);
      if (kv_runs(L, p, ld.g)) {  // (head-major rings) 16 slots = 4 pieces = one 4-KiB run, K then V
        const bf16_t* kb = L.ck + ring0 + (size_t)p.s_begin * DH;
        const bf16_t* vb = L.cv + ring0 + (size_t)p.s_begin * DH;
        for (int j = 0; j < p.n_att; j += 4) {
#if ENG_SADDR
          ld.piece4_s(reinterpret_cast<const char*>(kb) + (size_t)j * PIECE);
          ld.piece4_s(reinterpret_cast<const char*>(vb) + (size_t)j * PIECE);
#else
          ld.piece4(reinterpret_cast<const char*>(kb) + (size_t)j * PIECE + lane * 16);
          ld.piece4(reinterpret_cast<const char*>(vb) + (size_t)j * PIECE + lane * 16);
#endif
        }
      } else {
        const size_t base = ring0 + (lane & 15) * 8;
        for (int j = 0; j < p.n_att; ++j) {
          const int slot = min(p.s_begin + 4 * j + (lane >> 4), L.W - 1);
          ld.piece(L.ck + base + (size_t)slot * row_stride);
          ld.piece(L.cv + base + (size_t)slot * row_stride);
        }
      }
    }
    trace_ev(sh, c, l, TR_CONS + 2, tr);
    if (NHOLD && ENG_HOLD_STAGE == 2) sh.ctl[C_LSTAGE] = (uint32_t)(l + 1);
    ld.pairs(L.wo, p.o0, p.o1, a.H * DH);
    trace_ev(sh, c, l, TR_CONS + 3, tr);
    if (NHOLD && ENG_HOLD_STAGE == 3) sh.ctl[C_LSTAGE] = (uint32_t)(l + 1);  // the latency-critical small phases are issued: holders may fetch
    if constexpr (!MOE) {
      const int f_ring = p.f1 - holder_units(a, p.f1 - p.f0);
      for (int j = p.f0; j < f_ring; ++j) {
        const size_t r0 = (size_t)(2 * j) * a.D, r1 = r0 + a.D;
        const bf16_t* const rp[4] = {L.w1 + r0, L.w3 + r0, L.w1 + r1, L.w3 + r1};
        ld.template unit<4>(rp, a.D >> 9);
      }
      trace_ev(sh, c, l, TR_CONS + 4, tr);
      ld.pairs(L.w2, p.o0, p.o1, a.F);
    } else {
      // MoE (moe.py:24-32): WHICH expert matrices stream next is decided by the router, i.e. by this layer's activations:
      // the one edge where the loader cannot run ahead.  Everything issued so far must be visible to the consumers (they
      // need the Wo rows to get to the router at all), so the stream is flushed and restarts on a fill boundary - a
      // published fill must never gain pieces afterwards (the consumers skip to the same boundary).
      ld.flush();
      ld.g = (ld.g + FILL - 1) & ~(uint32_t)(FILL - 1);
      uint32_t word = 0, spins = 0;
      while (((word = sh.ctl[C_EXPERT]) >> 16) != (uint32_t)(l + 1))
        if (!spin_ok(sh, spins, 0x100)) break;
      const void* const* tab = reinterpret_cast<const void* const*>(L.w2);  // device table [E][3] of (w1, w2, w3)
      const int ex[2] = {__builtin_amdgcn_readfirstlane((int)(word & 0xffu)), __builtin_amdgcn_readfirstlane((int)((word >> 8) & 0xffu))};
      for (int q = 0; q < 2; ++q) {  // ascending expert id: the order of moe.py's accumulation, and of the consumers
        const bf16_t* e1 = reinterpret_cast<const bf16_t*>(tab[ex[q] * 3 + 0]);
        const bf16_t* e3 = reinterpret_cast<const bf16_t*>(tab[ex[q] * 3 + 2]);
        for (int j = p.f0; j < p.f1; ++j) {
          const size_t r0 = (size_t)(2 * j) * a.D, r1 = r0 + a.D;
          const bf16_t* const rp[4] = {e1 + r0, e3 + r0, e1 + r1, e3 + r1};
          ld.template unit<4>(rp, a.D >> 9);
        }
      }
      trace_ev(sh, c, l, TR_CONS + 4, tr);
      for (int q = 0; q < 2; ++q) ld.pairs(reinterpret_cast<const bf16_t*>(tab[ex[q] * 3 + 1]), p.o0, p.o1, a.F);
    }
    trace_ev(sh, c, l, TR_CONS + 5, tr);
#if ENG_TRACE == 1
    if (sh.trace && tr) sh.trace[((size_t)c * ENG_MAXL + l) * TR_EVENTS + TR_CONS + 6] = ld.stalls;
#endif
  }
  if (a.head) {
    int v0, v1;
    slab(a.V / 2, c, a.NB, v0, v1);
    ld.pairs(a.output, v0, v1, a.D);
  }
  (void)PD;
  ld.flush();
}

// ------------------------------------------------------------------------------------------------ consumer waves
struct Cons {
  const Shared& sh;
  int w, lane;
  uint32_t landed = 0;   // cached copy of ctl->landed
  uint32_t cbar_target = 0;

  __device__ __forceinline__ void need_fill(uint32_t piece_idx) {  // wait until the fill holding piece_idx has landed
    const uint32_t need = piece_idx / FILL + 1;
    if (landed >= need) return;
    uint32_t spins = 0;
    for (;;) {
      landed = sh.ctl[C_LANDED];
      if (landed >= need) return;
      if (!spin_ok(sh, spins, 0x200)) return;
    }
  }
  __device__ __forceinline__ void set_done(uint32_t piece_idx) { sh.ctl[C_DONE + w] = piece_idx; }

  // barrier among the NCONS consumer waves (the loader and the holders never take part, so s_barrier is out)
  __device__ __forceinline__ void cbar() {  // one shared counter; waiting waves sleep between polls, which leaves the SIMD's issue slots to the wave still working
    cbar_target += NCONS;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((lu32*)(sh.ctl + C_CBAR), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t spins = 0;
    while (sh.ctl[C_CBAR] < cbar_target)
      if (!spin_ok(sh, spins, 0x300)) break;
    asm volatile("" ::: "memory");
  }

  // fp32 dots of NR consecutive streamed weight rows (P pieces each, the first at piece g0) with the activation vector
  // in LDS.  Lane owns elements (p * 64 + lane) * 8 .. + 8 of every 512-element piece p and accumulates them pairwise
  // (dot2_bf16) in ascending order - the launch path's order per row.  The NR rows advance in lockstep: NR independent
  // accumulation chains (a single chain is latency-bound: 28 GB/s per CU measured, below the HBM stream).
  // A4: every row of this instantiation has a multiple of 4 pieces - the generic-group path is not compiled
  template <int NR, bool A4>
  __device__ __forceinline__ void unit_dot(uint32_t g0, int P, const lbf16* xs, float (&out)[NR]) {
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.f;
    const lchar* xl = reinterpret_cast<const lchar*>(xs) + lane * 16;
    const lchar* wl = sh.ring + lane * 16;
    auto step = [&](const u32x4& xv, const u32x4 (&wv)[NR]) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[r] = dot2_bf16(wv[r][i], xv[i], acc[r]);
    };
    // the unit arrives in groups of G pieces per row (Loader::unit): wait for a group, reduce it, hand its ring space back
    const int G = A4 ? 4 : unit_group(P);
#if ENG_WIDE
    if (!A4 && G != 4) {
      // linear unit (Loader::unit): rows of a pair are P pieces apart; a gate/up unit is [w1 2j | w1 2j+1 | w3 2j | w3 2j+1]
      // while the accumulators are ordered (w1 2j, w3 2j, w1 2j+1, w3 2j+1)
      need_fill(g0 + NR * P - 1);
      int off[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) off[r] = (NR == 2 ? r : ((r & 1) * 2 + (r >> 1))) * P;
      int p = 0;
      for (; p + 1 < P; p += 2) {
        u32x4 xv[2], wv[2][NR];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          xv[j] = lds16(xl + (p + j) * PIECE);
#pragma unroll
          for (int r = 0; r < NR; ++r) wv[j][r] = lds16(wl + RING_IDX(sh, g0 + off[r] + p + j) * PIECE);
        }
        step(xv[0], wv[0]);
        step(xv[1], wv[1]);
      }
      if (p < P) {
        u32x4 wv[NR];
        const u32x4 xv = lds16(xl + p * PIECE);
#pragma unroll
        for (int r = 0; r < NR; ++r) wv[r] = lds16(wl + RING_IDX(sh, g0 + off[r] + p) * PIECE);
        step(xv, wv);
      }
      set_done(g0 + NR * P);
#pragma unroll
      for (int r = 0; r < NR; ++r) out[r] = wave_sum(acc[r]);
      return;
    }
#endif
    constexpr int S = (NR <= 2) ? 4 : 2;  // pieces per row whose LDS reads are issued together
    uint32_t gg = g0;                     // first ring piece of the current group
    for (int p0 = 0; p0 < P; p0 += G, gg += NR * G) {
      need_fill(gg + NR * G - 1);
      if (A4 || G == 4) {
#pragma unroll
        for (int h = 0; h < 4; h += S) {
          u32x4 xv[S], wv[S][NR];
#pragma unroll
          for (int j = 0; j < S; ++j) {
            xv[j] = lds16(xl + (p0 + h + j) * PIECE);
#pragma unroll
            for (int r = 0; r < NR; ++r) wv[j][r] = lds16(wl + RING_IDX(sh, gg + r * 4 + h + j) * PIECE);
          }
#pragma unroll
          for (int j = 0; j < S; ++j) step(xv[j], wv[j]);
        }
      } else {
        for (int i = 0; i < G; ++i) {
          u32x4 wv[NR];
          const u32x4 xv = lds16(xl + (p0 + i) * PIECE);
#pragma unroll
          for (int r = 0; r < NR; ++r) wv[r] = lds16(wl + RING_IDX(sh, gg + r * G + i) * PIECE);
          step(xv, wv);
        }
      }
      set_done(gg + NR * G);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) out[r] = wave_sum(acc[r]);
  }

  __device__ __forceinline__ void publish(gu64* g, uint32_t tag, uint32_t value) {
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // All consumer waves: copy n granules (granule i lives at addr(i); its tag must match) into LDS words dst[0, n).
  // A sweep that finds some tags missing is repeated as a whole (all loads unconditional - a predicated load would
  // serialise them; re-polling only the missing granules measured slower).
  template <int NL, class AddrFn>
  __device__ __forceinline__ void gather_fn(int n, uint32_t tag, lu32* dst, AddrFn addr) {
    for (int k0 = 0; k0 * NCONS * 64 < n; k0 += NL) {
      unsigned long long x[NL];
      bool have[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        x[k] = 0;
        have[k] = ((k0 + k) * NCONS + w) * 64 + lane >= n;
      }
      uint32_t spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          const int i = ((k0 + k) * NCONS + w) * 64 + lane;
          const unsigned long long y =
              __hip_atomic_load(addr(min(i, n - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!have[k]) x[k] = y;
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          have[k] = have[k] || ((uint32_t)(x[k] >> 32) == tag);
          ok &= have[k];
        }
        if (__all(ok)) break;
        if (!spin_ok(sh, spins, 0x400)) break;
      }
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int i = ((k0 + k) * NCONS + w) * 64 + lane;
        if (i < n) dst[i] = (uint32_t)x[k];
      }
    }
  }
  // Contiguous granule array (n even): 16-byte sc1 loads, two granules each - half the load instructions and wider
  // transactions than the 8-byte sweep (each 8-byte half is still validated by its own tag).
  template <int NL = 4>
  __device__ __forceinline__ void gather(const gu64* src, int n, uint32_t tag, lu32* dst) {
    const int n2 = n >> 1;  // granule pairs
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 8, 0x00020000);
    for (int k0 = 0; k0 * NCONS * 64 < n2; k0 += NL) {
      u32x4 x[NL];
      bool have[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        x[k] = u32x4{0u, 0u, 0u, 0u};
        have[k] = ((k0 + k) * NCONS + w) * 64 + lane >= n2;
      }
      uint32_t spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          const int i = ((k0 + k) * NCONS + w) * 64 + lane;
          const int off = min(i, n2 - 1) * 16;
          const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16 /* sc1 */);
          if (!have[k]) x[k] = y;
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          have[k] = have[k] || (x[k][1] == tag && x[k][3] == tag);
          ok &= have[k];
        }
        if (__all(ok)) break;
        if (!spin_ok(sh, spins, 0x400)) break;
      }
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int i = ((k0 + k) * NCONS + w) * 64 + lane;
        if (i < n2) *reinterpret_cast<LDS_AS u32x2*>(dst + 2 * i) = u32x2{x[k][0], x[k][2]};
      }
    }
  }

  // RMSNorm of the K-element bf16 vector in LDS, in place (transformer_layers.py:115-120), with the launch path's
  // reduction tree: 256 "threads" own 16-byte pieces vt + i * 256, per-piece sums, wave butterfly, 4 wave totals.
  // The norm weights do not depend on the activations: norm_prefetch issues their loads BEFORE the hand-off sweep so
  // that the global round trip is over when the vector arrives.
  struct NormW {
    u32x4 w[4];
  };
  __device__ __forceinline__ void norm_prefetch(NormW& nw, int K, const bf16_t* norm_w) {
    const int vt = w * 64 + lane, npieces = K >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) nw.w[i] = ld16(norm_w + (size_t)min(vt + i * 256, npieces - 1) * 8);
  }
  // Pieces of the K-element vector this lane owns in the norm's reduction tree (q = vt + i * 256), from LDS ...
  __device__ __forceinline__ void norm_load_lds(u32x4 (&xr)[4], const lbf16* xs, int K) {
    const int vt = w * 64 + lane, npieces = K >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = vt + i * 256;
      if (q < npieces) xr[i] = lds16(xs + q * 8);
    }
  }
  // ... or straight from the hand-off granules: piece q = granules 4q .. 4q+3 = two 16-byte sc1 loads of this lane, so the
  // sweep IS the norm's first pass (no LDS round trip and no barrier between them).
  __device__ __forceinline__ void norm_load_granules(u32x4 (&xr)[4], const gu64* src, int K, uint32_t tag) {
    const int vt = w * 64 + lane, npieces = K >> 3;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (K / 2) * 8, 0x00020000);
    u32x4 lo[4], hi[4];
    bool have[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lo[i] = hi[i] = u32x4{0u, 0u, 0u, 0u};
      have[i] = vt + i * 256 >= npieces;
    }
    uint32_t spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = min(vt + i * 256, npieces - 1);
        const int off = q * 32;
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16 /* sc1 */);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 16);
        if (!have[i]) {
          lo[i] = a;
          hi[i] = b;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        have[i] = have[i] || (lo[i][1] == tag && lo[i][3] == tag && hi[i][1] == tag && hi[i][3] == tag);
        ok &= have[i];
      }
      if (__all(ok)) break;
      if (!spin_ok(sh, spins, 0x400)) break;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = u32x4{lo[i][0], lo[i][2], hi[i][0], hi[i][2]};
  }
  // RMSNorm of the pieces in xr, written to xs (transformer_layers.py:115-120), with the launch path's reduction tree:
  // 256 "threads" own 16-byte pieces vt + i * 256, per-piece sums, wave butterfly, 4 wave totals.
  __device__ __forceinline__ void rmsnorm_store(const u32x4 (&xr)[4], lbf16* xs, int K, const NormW& nw, float eps) {
    const int vt = w * 64 + lane, npieces = K >> 3;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = vt + i * 256;
      if (q < npieces) {
        float s = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float x0 = bf_lo(xr[i][cc]), x1 = bf_hi(xr[i][cc]);
          s = fmaf(x0, x0, s);
          s = fmaf(x1, x1, s);
        }
        ss += s;
      }
    }
    ss = wave_sum(ss);
    if (lane == 0) reinterpret_cast<lvf32*>(sh.ctl + C_RED)[w] = ss;
    cbar();
    lvf32* red = reinterpret_cast<lvf32*>(sh.ctl + C_RED);
    const float s = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / sqrtf(s / (float)K + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = vt + i * 256;
      if (q < npieces) {
        u32x4 o;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          o[cc] = pack_bf2(bf_round(bf_lo(xr[i][cc]) * inv) * bf_lo(nw.w[i][cc]), bf_round(bf_hi(xr[i][cc]) * inv) * bf_hi(nw.w[i][cc]));
        lds_st16(xs + q * 8, o);
      }
    }
    cbar();
  }
};

// ---------------------------------------------------------------------------------------------------- MoE feed-forward
// moe.py:24-32 for the one token of a decode step (top-2 routing), behind ffn_norm - whose output stands in the activation
// region, with the raw h1 pieces of this lane still in `xr`:
//   router   every workgroup computes the E gate logits itself (the gate matrix is 64 KB, read from L2 by everybody - no
//            hand-off), with the arithmetic of the launch path's moe_router_kernel: its own sum of squares over the lane-
//            strided pieces, fp32 fmaf chain per lane, wave sum, bf16 logit; top-k with ties to the lower id; softmax over
//            the picks in fp32, rounded to bf16.  The two picks, in ASCENDING id, go to the loader (C_EXPERT).
//   W1|W3    this workgroup's unit slab of expert A, then of expert B -> two hid vectors (granule arrays g_hid, g_hid2)
//   W2       expert A rows with hid A (gathered long after its last producer finished: the sweep is one pass), then
//            expert B rows with hid B; r = bf16(bf16(0 + bf16(wA yA)) + bf16(wB yB)); h = bf16(h1 + r)  (moe.py:28-32 +
//            transformer_layers.py:168, the order of the launch path's moe_w2_kernel)
template <bool ALL4>
__device__ __forceinline__ void moe_ffn(const EngArgs& a, const Shared& sh, Cons& cs, const EngLayer& L, const LayerPlan& p, int l,
                                        int c, int w, int lane, uint32_t& g, uint32_t tag_hid, uint32_t tag_h,
                                        const u32x4 (&xr)[4], bool trc) {
  lbf16* xs = reinterpret_cast<lbf16*>(sh.xs);
  lu32* xs32 = reinterpret_cast<lu32*>(sh.xs);
  gu64* G = (gu64*)a.gran;
  const int PD = a.D >> 9, PF = a.F >> 9, np = a.D >> 3;
  // ---- raw h1 next to the normalised vector: the router normalises on its own (different reduction tree)
  lbf16* raw = xs + a.D;
  {
    const int vt = w * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = vt + i * 256;
      if (q < np) lds_st16(raw + q * 8, xr[i]);
    }
  }
  cs.cbar();
  // ---- router (moe_router_kernel: one wave per expert there, experts w, w + 4, ... per wave here)
#if ENG_WIDE
  // Round 4 (timelines of an 8x7B / 8x22B stage, profiles/r04_engine_trace_*).  The shipped form below took 11-15 us per layer
  // - time in which the loader has nothing to stream (it waits for this decision): (1) the gate row and the norm weights
  // were fetched INSIDE the per-piece loop, one expert after the other: 16-24 dependent L2 round trips; (2) every wave
  // normalised the WHOLE vector again for every one of its experts (two bf16 roundings per element, ~3 us of VALU).
  // Here: every wave computes the sum of squares (the launch path's per-wave order), normalises ITS QUARTER of the vector
  // once into LDS - the same numbers the shipped form recomputes - and after one barrier runs its two experts' dot
  // products together over four pieces per batch, the first batch of gate rows requested before that barrier.  Per
  // expert the fmaf chain runs over the same values in the same order: bit-identical.  (Two more forms were measured and
  // dropped: prefetching the gate rows before the h1 sweep spills the consumers' registers; the whole router on the idle
  // holder waves is compute-bound there - three experts per wave - and came out slower: profiles/EXPERIMENTS.md.)
  {
    const bf16_t* gate = L.w1;
    lbf16* xn = raw + a.D;  // router-normalised activations (bf16), behind the raw copy
    constexpr int CH = 4;
    const int ea0 = w, eb0 = (w + NCONS < a.E) ? w + NCONS : w;
    u32x4 wnq[4];  // norm weights of this wave's quarter of the pieces (dim <= 8192: at most 4 per lane)
#pragma unroll
    for (int i = 0; i < 4; ++i) wnq[i] = ld16(L.fn + (size_t)min(w * 64 + lane + NCONS * 64 * i, np - 1) * 8);
    float ss = 0.f;
    for (int pp = lane; pp < np; pp += 64) {
      const u32x4 v = lds16(raw + pp * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = bf_lo(v[i]), x1 = bf_hi(v[i]);
        ss = fmaf(x0, x0, ss);
        ss = fmaf(x1, x1, ss);
      }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)a.D + a.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // this wave's quarter of the pieces
      const int pp = w * 64 + lane + NCONS * 64 * i;
      if (pp < np) {
        const u32x4 v = lds16(raw + pp * 8);
        u32x4 o;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          o[cc] = pack_bf2(bf_round(bf_lo(v[cc]) * inv) * bf_lo(wnq[i][cc]), bf_round(bf_hi(v[cc]) * inv) * bf_hi(wnq[i][cc]));
        lds_st16(xn + pp * 8, o);
      }
    }
    u32x4 va0[CH], vb0[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {  // first batch of this wave's first two experts: in flight across the barrier
      const int pp = min(lane + 64 * i, np - 1);
      va0[i] = ld16(gate + (size_t)ea0 * a.D + pp * 8);
      vb0[i] = ld16(gate + (size_t)eb0 * a.D + pp * 8);
    }
    cs.cbar();
    for (int ea = w; ea < a.E; ea += 2 * NCONS) {
      const int eb = ea + NCONS;
      const bool two = eb < a.E;
      const bf16_t* ga = gate + (size_t)ea * a.D;
      const bf16_t* gb = gate + (size_t)(two ? eb : ea) * a.D;
      float acc_a = 0.f, acc_b = 0.f;
      auto fold = [&](int p0, const u32x4 (&va)[CH], const u32x4 (&vb)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          if (p0 + 64 * i < np) {  // (np is a multiple of 64: a batch is inside the vector or outside it for the whole wave)
            const u32x4 xv = lds16(xn + (p0 + 64 * i) * 8);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const float x0 = bf_lo(xv[cc]), x1 = bf_hi(xv[cc]);
              acc_a = fmaf(bf_lo(va[i][cc]), x0, acc_a);
              acc_a = fmaf(bf_hi(va[i][cc]), x1, acc_a);
              acc_b = fmaf(bf_lo(vb[i][cc]), x0, acc_b);
              acc_b = fmaf(bf_hi(vb[i][cc]), x1, acc_b);
            }
          }
        }
      };
      int p0 = lane;
      if (ea == w) {  // the batch that has been in flight since before the barrier
        fold(p0, va0, vb0);
        p0 += 64 * CH;
      }
      for (; p0 < np; p0 += 64 * CH) {
        u32x4 va[CH], vb[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int pp = min(p0 + 64 * i, np - 1);
          va[i] = ld16(ga + pp * 8);
          vb[i] = ld16(gb + pp * 8);
        }
        fold(p0, va, vb);
      }
      acc_a = wave_sum(acc_a);
      acc_b = wave_sum(acc_b);
      if (lane == 0) {
        reinterpret_cast<lvf32*>(sh.ctl + C_RLOGIT)[ea] = bf_round(acc_a);
        if (two) reinterpret_cast<lvf32*>(sh.ctl + C_RLOGIT)[eb] = bf_round(acc_b);
      }
    }
    cs.cbar();
  }
#else
  {
    const bf16_t* gate = L.w1;  // (MoE layers carry the gate in the w1 slot and the expert table in the w2 slot)
    float ss = 0.f;
    for (int pp = lane; pp < np; pp += 64) {
      const u32x4 v = lds16(raw + pp * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = bf_lo(v[i]), x1 = bf_hi(v[i]);
        ss = fmaf(x0, x0, ss);
        ss = fmaf(x1, x1, ss);
      }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)a.D + a.eps);
    for (int e = w; e < a.E; e += NCONS) {
      const bf16_t* gr = gate + (size_t)e * a.D;
      float acc = 0.f;
      for (int pp = lane; pp < np; pp += 64) {
        const u32x4 v = lds16(raw + pp * 8);
        const u32x4 gv = ld16(gr + pp * 8);
        const u32x4 wv = ld16(L.fn + pp * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x0 = bf_round(bf_round(bf_lo(v[i]) * inv) * bf_lo(wv[i]));
          const float x1 = bf_round(bf_round(bf_hi(v[i]) * inv) * bf_hi(wv[i]));
          acc = fmaf(bf_lo(gv[i]), x0, acc);
          acc = fmaf(bf_hi(gv[i]), x1, acc);
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) reinterpret_cast<lvf32*>(sh.ctl + C_RLOGIT)[e] = bf_round(acc);
    }
  }
  cs.cbar();
#endif
  int eA, eB;
  float wA, wB;
  {
#if ENG_WIDE
    // (the 16 logits in four 16-byte LDS reads instead of 2 x E dependent volatile reads)
    float lg[16];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const u32x4 t = *reinterpret_cast<const LDS_AS volatile u32x4*>(sh.ctl + C_RLOGIT + 4 * q4);
#pragma unroll
      for (int i = 0; i < 4; ++i) lg[4 * q4 + i] = __uint_as_float(t[i]);
    }
#else
    const lvf32* lg = reinterpret_cast<const lvf32*>(sh.ctl + C_RLOGIT);
#endif
    int ti[2];
    float tw[2];
    unsigned taken = 0;
    for (int k = 0; k < 2; ++k) {
      int best = -1;
      float bv = -INFINITY;
#if ENG_WIDE
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j >= a.E) break;
#else
      for (int j = 0; j < a.E; ++j) {
#endif
        const float v = lg[j];
        if (!((taken >> j) & 1u) && (best < 0 || v > bv)) {  // ties: lowest expert id
          best = j;
          bv = v;
        }
      }
      taken |= 1u << best;
      ti[k] = best;
      tw[k] = bv;
    }
    const float ex0 = expf(tw[0] - tw[0]), ex1 = expf(tw[1] - tw[0]);
    float den = 0.f;
    den += ex0;
    den += ex1;
    const float w0 = bf_round(ex0 / den), w1 = bf_round(ex1 / den);
    const bool swap = ti[1] < ti[0];
    eA = swap ? ti[1] : ti[0];
    eB = swap ? ti[0] : ti[1];
    wA = swap ? w1 : w0;
    wB = swap ? w0 : w1;
  }
  if (w == 0 && lane == 0) sh.ctl[C_EXPERT] = ((uint32_t)(l + 1) << 16) | (uint32_t)eA | ((uint32_t)eB << 8);
  trace_ev(sh, c, l, 13, trc);
  // ---- W1|W3 of the two experts (the loader restarts on a fill boundary behind the router edge)
  g = (g + FILL - 1) & ~(uint32_t)(FILL - 1);
  {
    const int n_u = p.f1 - p.f0;
    for (int k = w; k < 2 * n_u; k += NCONS) {
      const uint32_t ga = g + (uint32_t)(4 * k) * PD;
      cs.set_done(ga);
      float vv[4];
      cs.template unit_dot<4, ALL4>(ga, PD, xs, vv);
      if (lane == 0) {
        const uint32_t packed = (uint32_t)f_to_bf(swiglu_bf(vv[0], vv[1])) | ((uint32_t)f_to_bf(swiglu_bf(vv[2], vv[3])) << 16);
        const int q = k >= n_u, j = k - q * n_u;
        cs.publish(G + (q ? a.g_hid2 : a.g_hid) + p.f0 + j, tag_hid, packed);
      }
    }
    g += (uint32_t)(4 * 2 * n_u) * PD;
    cs.set_done(g);
  }
  trace_ev(sh, c, l, 14, trc);
  // ---- W2 of expert A, then of expert B
  lf32* keep = reinterpret_cast<lf32*>(sh.xs + (size_t)a.F * 2);  // r after expert A, two floats per unit of this workgroup
  const int n_o = p.o1 - p.o0;
  const bool to_global = (l == a.n_layers - 1);
  for (int q = 0; q < 2; ++q) {
    cs.cbar();  // the region's previous content (x, or hid A) is dead
    sh.ctl[C_GATHERING] = 1;
    cs.template gather<14>(G + (q ? a.g_hid2 : a.g_hid), a.F / 2, tag_hid, xs32);
    cs.cbar();
    sh.ctl[C_GATHERING] = 0;
    if (q == 0) trace_ev(sh, c, l, 15, trc);
    const float wq = q ? wB : wA;
    for (int k = w; k < n_o; k += NCONS) {
      const uint32_t ga = g + (uint32_t)(2 * k) * PF;
      cs.set_done(ga);
      float vv[2];
      cs.template unit_dot<2, ALL4>(ga, PF, xs, vv);
      if (lane == 0) {
        const float t0 = bf_round(wq * bf_round(vv[0])), t1 = bf_round(wq * bf_round(vv[1]));
        if (q == 0) {
          keep[2 * k] = bf_round(0.f + t0);
          keep[2 * k + 1] = bf_round(0.f + t1);
        } else {
          const float r0 = bf_round(keep[2 * k] + t0), r1 = bf_round(keep[2 * k + 1] + t1);
          const uint32_t rs = *reinterpret_cast<const lu32*>(sh.res + 2 * k);
          const uint32_t packed = pack_bf2(bf_lo(rs) + r0, bf_hi(rs) + r1);
          *reinterpret_cast<lu32*>(sh.res + 2 * k) = packed;  // residual of the next layer's Wo epilogue
          cs.publish(G + a.g_h + p.o0 + k, tag_h, packed);
          if (to_global) *reinterpret_cast<uint32_t*>(a.h + 2 * (size_t)(p.o0 + k)) = packed;
        }
      }
    }
    g += (uint32_t)(2 * n_o) * PF;
    cs.set_done(g);
  }
}

template <int R, bool MOE, bool ALL4>
__device__ __forceinline__ void run_consumer(const EngArgs& a, const Shared& sh, int c, int w, int lane, int pos, int seq, uint32_t epoch,
                                             uint32_t arrive_target) {
  Cons cs{sh, w, lane};
  if (ENG_CONS_PRIO) __builtin_amdgcn_s_setprio(ENG_CONS_PRIO);
  lbf16* xs = reinterpret_cast<lbf16*>(sh.xs);
  lu32* xs32 = reinterpret_cast<lu32*>(sh.xs);
  gu64* G = (gu64*)a.gran;
  const int PD = a.D >> 9, PA = (a.H * DH) >> 9, PF = a.F >> 9;
  const int nq = a.H * DH, nkv = a.Hkv * DH;
  uint32_t g = 0;  // first piece of the current segment
  uint32_t hold_target = 0;  // W1|W3 units the holder waves must have finished (cumulative)
#if ENG_QKV_HOLD
  uint32_t hq_target = 0;    // holder waves that must have finished their q|k|v units (cumulative)
#endif
  long greedy_token = 0;     // the fused greedy sample (workgroup 0, wave 0, lane 0)
  float greedy_logprob = 0.f;
  bool greedy_valid = false;
  auto tag_of = [&](int layer, int edge) { return (epoch << 12) | (uint32_t)((a.seq_base + layer) * 8 + edge + 1); };

  // attention scratch inside the activation region (free between the q|k|v rows and the Wo gather)
  lu32* q_lds = xs32;                                            // R * 64 words
  lu32* kn_lds = xs32 + R * 64;                                  // 64 words
  lu32* vn_lds = kn_lds + 64;                                    // 64 words
  lf32* sm_m = reinterpret_cast<lf32*>(vn_lds + 64);             // 4 R
  lf32* sm_l = sm_m + 4 * R;                                     // 4 R
  lf32* sm_acc = sm_l + 4 * R;                                   // 4 R DH
  lu32* cmb_lds = reinterpret_cast<lu32*>(sm_acc + 4 * R * DH);  // split merge staging: 3 * n_splits * ne words

  for (int l = 0; l < a.n_layers; ++l) {
    const EngLayer& L = a.L[l];
    LayerPlan p;
    plan_layer(a, L, c, pos, p);
    const bool trc = (w == 0) && (lane == 0);
    trace_ev(sh, c, l, 0, trc);

    // ================================================================ attention_norm + q|k|v + RoPE + ring write
    typename Cons::NormW nw;
    cs.norm_prefetch(nw, a.D, L.an);
    u32x4 xr[4];
    if (l == 0 && a.first) {  // the step's input comes from global memory: the embedding row of this step's token
      // (transformer.py:193), or h as the previous stage / previous launch left it
      const bf16_t* hin = a.h;
      if (a.emb) {
        long id = (long)a.ids[0];
        if (id < 0 || id >= a.V) {  // the reference's nn.Embedding raises IndexError: flagged for the host, row clamped
          if (c == 0 && w == 0 && lane == 0) atomicMax((uint32_t*)a.ctrl + G_BADID, 1u);
          id = id < 0 ? 0 : a.V - 1;
        }
        hin = a.emb + (size_t)id * a.D;
      }
      const int vt = w * 64 + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[i] = ld16(hin + (size_t)min(vt + i * 256, (a.D >> 3) - 1) * 8);
      for (int r = 2 * p.o0 + vt; r < 2 * p.o1; r += NCONS * 64) sh.res[r - 2 * p.o0] = hin[r];
    } else {
      sh.ctl[C_GATHERING] = GATHER_FLAG(1);
      cs.norm_load_granules(xr, G + a.g_h, a.D, tag_of(l - 1, 0));
      sh.ctl[C_GATHERING] = 0;
    }
    trace_ev(sh, c, l, 1, trc);
    cs.rmsnorm_store(xr, xs, a.D, nw, a.eps);
    trace_ev(sh, c, l, 2, trc);
    if (l == 0) {
      // RESIDENCY GATE.  Every hand-off below assumes that all NB workgroups run at the same time (one per CU).  Nothing
      // has been written yet - no ring row, no granule - so a launch that finds a workgroup missing (a CU masked or busy
      // with another process) gives up here WITHOUT side effects: the step can be re-run on the launch path from
      // unchanged state (Transformer._recover_engine).  The arrival count was taken at kernel entry; by now (a norm and
      // ~1 us later) it is complete on a healthy chip and the wait is one L2 read.
      if (w == 0) {
        uint32_t polls = 0;
        while ((int)(__hip_atomic_load(sh.ctrl + G_ARRIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - arrive_target) < 0) {
          __builtin_amdgcn_s_sleep(8);
          if (sh.ctl[C_ABORT] || ++polls >= ARRIVE_POLLS) {
            raise_abort(sh, 0x700);
            if (c == 0 && lane == 0 && sh.ctrl[G_SABOTAGE] != 0) sh.ctrl[G_SABOTAGE] -= 1;  // (every workgroup read it at entry)
            break;
          }
        }
        sh.ctl[C_ARRIVED] = 1;
      } else {
        uint32_t spins = 0;
        while (!sh.ctl[C_ARRIVED])
          if (!spin_ok(sh, spins, 0x700)) break;
      }
      if (sh.ctl[C_ABORT]) {
        cs.set_done(0xffffffffu);
        return;
      }
    }
    {
      const int nq_u = p.q1 - p.q0, nk_u = p.k1 - p.k0;
#if ENG_QKV_HOLD
      const int n_held = qkv_held(a, nq_u + 2 * nk_u);
      const int n_u = nq_u + 2 * nk_u - n_held;  // the holder waves reduce the last n_held units from their registers
      // (rmsnorm_store ended with a barrier of the consumer waves; layer 0: the residency gate has been passed, so the
      // holders' side effects - granules, ring rows - come after it as well)
      if (n_held && w == 0) sh.ctl[C_XA] = (uint32_t)(l + 1);
#else
      const int n_u = nq_u + 2 * nk_u;
#endif
      for (int k = w; k < n_u; k += NCONS) {
        const uint32_t ga = g + (uint32_t)(2 * k) * PD;
        cs.set_done(ga);
        int kind, u;  // 0 q, 1 k, 2 v; u = global row pair inside that matrix
        if (k < nq_u) { kind = 0; u = p.q0 + k; }
        else if (k < nq_u + nk_u) { kind = 1; u = p.k0 + (k - nq_u); }
        else { kind = 2; u = p.v0 + (k - nq_u - nk_u); }
        const int r0 = 2 * u;
        // the rotary entry is fetched BEFORE the dot products (an L2 round trip otherwise sits in every unit's tail)
        const float2 cs2 = *reinterpret_cast<const float2*>(a.rope_cs + ((size_t)pos * (DH >> 1) + ((r0 % DH) >> 1)) * 2);
        float v[2];
        cs.template unit_dot<2, ALL4>(ga, PD, xs, v);
        if (lane == 0) {
          float y0 = bf_round(v[0]), y1 = bf_round(v[1]);
          if (kind < 2) {  // rope.py:13-23 on the adjacent pair
            float re, im;
            rope_pair(y0, y1, cs2.x, cs2.y, re, im);
            y0 = re;
            y1 = im;
          }
          const uint32_t packed = pack_bf2(y0, y1);
          if (kind > 0) {  // cache.py:83-92
            bf16_t* ring = (kind == 1 ? L.ck : L.cv) + kv_offset(L.kv_layout, L.W, nkv, DH, (size_t)seq, p.cur_slot, r0);
            *reinterpret_cast<uint32_t*>(ring) = packed;
          }
          const int gi = (kind == 0 ? 0 : (kind == 1 ? nq / 2 : nq / 2 + nkv / 2)) + u;
          cs.publish(G + a.g_qkv + gi, tag_of(l, 1), packed);
        }
      }
      g += (uint32_t)(2 * n_u) * PD;
      cs.set_done(g);
    }
    trace_ev(sh, c, l, 3, trc);

    // ================================================================ attention: this CU's (kv head, split)
    cs.cbar();  // every wave is done with the normalised activations: the region becomes attention scratch
#if ENG_QKV_HOLD
    if (qkv_held(a, (p.q1 - p.q0) + 2 * (p.k1 - p.k0))) {  // ... and so are the holder waves (they finished long ago: 4 rows each)
      hq_target += (uint32_t)NHOLD;
      uint32_t spins = 0;
      while (sh.ctl[C_HQDONE] < hq_target)
        if (!spin_ok(sh, spins, 0x500)) break;
    }
#endif
    trace_ev(sh, c, l, 4, trc);
    if (p.att) {
      const int kv_real = p.kvh / a.kv_groups;
      const uint32_t tq = tag_of(l, 1);
      sh.ctl[C_GATHERING] = GATHER_FLAG(2);
      {  // q of the R query heads | this step's k row | v row of the kv head: ONE sweep (q_lds, kn_lds, vn_lds are contiguous)
        const gu64* qg = G + a.g_qkv + (size_t)p.kvh * R * 64;
        const gu64* kg = G + a.g_qkv + nq / 2 + (size_t)kv_real * 64;
        const gu64* vg = kg + nkv / 2;
        cs.template gather_fn<2>(R * 64 + 128, tq, q_lds,
                                 [&](int i) { return i < R * 64 ? qg + i : (i < R * 64 + 64 ? kg + (i - R * 64) : vg + (i - R * 64 - 64)); });
      }
      cs.cbar();
      sh.ctl[C_GATHERING] = 0;
      trace_ev(sh, c, l, 5, trc);
      const int gl = lane >> 4, dl = lane & 15;
#if ENG_WIDE
      // GQA ratio 6 (Mixtral-8x22B): six heads' q rows and running sums in one pass need ~140 live registers on top of the
      // kernel's standing state and the instantiation spills 45 of them - not only here: a 7-layer 8x22B stage ran 2.00 ms
      // per step with the spilling one-pass form against 1.88 ms with this one (and 1.94-1.97 ms on the launch path; same
      // boxes, profiles/EXPERIMENTS.md round 4), although the attention pieces themselves take the same ~10 us either way.
      // The heads are served in two passes of three over the same ring slots: heads do not interact in reduce_slot, so the
      // partials are the same bits.
      if constexpr (R == 6) {
        {  // (never `streamed`: decode_engine_applicable declines rings whose split does not fit the LDS ring at this ratio)
          const bool runs = kv_runs(L, p, g);
          if (p.n_att) cs.need_fill(g + 2 * p.n_att - 1);
#pragma unroll
          for (int pass = 0; pass < 2; ++pass) {
            constexpr int RP = 3;
            const int r_off = pass * RP;
            float qh[RP][8];
            {
              u32x4 qraw[RP];
#pragma unroll
              for (int r = 0; r < RP; ++r) qraw[r] = lds16(q_lds + (r_off + r) * 64 + dl * 4);
              load_q<RP>(qh, qraw);
            }
            State<RP> sp;
            init_state<RP>(sp);
            for (int j = w; j < p.n_att; j += NCONS) {
              u32x4 kraw = lds16(sh.ring + RING_IDX(sh, g + kv_piece_k(runs, j)) * PIECE + lane * 16);
              u32x4 vraw = lds16(sh.ring + RING_IDX(sh, g + kv_piece_v(runs, j)) * PIECE + lane * 16);
              const int slot = p.s_begin + 4 * j + gl;
              if (slot == p.cur_slot) {
                kraw = lds16(kn_lds + dl * 4);
                vraw = lds16(vn_lds + dl * 4);
              }
              const bool valid = slot < p.s_end;
              if (!valid) {
                kraw = u32x4{0u, 0u, 0u, 0u};
                vraw = u32x4{0u, 0u, 0u, 0u};
              }
              reduce_slot<RP>(sp, qh, kraw, vraw, valid);
            }
            wave_state_to_lds_heads<RP, R>(sp, w, lane, r_off, sm_m, sm_l, sm_acc);
          }
          g += 2 * p.n_att;
          cs.set_done(g);
          trace_ev(sh, c, l, 6, trc);
        }
      } else {
#endif
      float qf[R][8];
      {
        u32x4 qraw[R];
#pragma unroll
        for (int r = 0; r < R; ++r) qraw[r] = lds16(q_lds + r * 64 + dl * 4);
        load_q<R>(qf, qraw);
      }
      State<R> st;
      init_state<R>(st);
      // this CU's K/V pieces were the first thing the loader fetched after the q|k|v weights: they landed long ago -
      // ONE wait for the last of them instead of a wait + ring bookkeeping per piece ... as long as they all fit the ring
      // while this wave pins its position (a 272-slot split of an 8K ring is 136 pieces: per-piece bookkeeping then)
      const bool streamed = 2 * p.n_att > (RING_FILLS - 2) * FILL;
      const bool runs = kv_runs(L, p, g);  // the loader's piece order: [K j .. j+3][V j .. j+3] runs or [K j][V j]
      if (p.n_att && !streamed) cs.need_fill(g + 2 * p.n_att - 1);
      for (int j = w; j < p.n_att; j += NCONS) {  // virtual wave w of the stand-alone kernel
        const uint32_t gk = g + kv_piece_k(runs, j), gv = g + kv_piece_v(runs, j);
        if (streamed) {  // (runs: this wave's next pieces lie in the group of 8 that begins at gk - (j & 3))
          cs.set_done(runs ? gk - (uint32_t)(j & 3) : gk);
          cs.need_fill(gv);
        }
        u32x4 kraw = lds16(sh.ring + RING_IDX(sh, gk) * PIECE + lane * 16);
        u32x4 vraw = lds16(sh.ring + RING_IDX(sh, gv) * PIECE + lane * 16);
        const int slot = p.s_begin + 4 * j + gl;
        if (slot == p.cur_slot) {  // this step's K/V row: taken from the granules, its ring write may still be in flight
          kraw = lds16(kn_lds + dl * 4);
          vraw = lds16(vn_lds + dl * 4);
        }
        const bool valid = slot < p.s_end;
        if (!valid) {  // never-written ring slots may hold NaN bit patterns (torch.empty): 0 * NaN would poison the sums
          kraw = u32x4{0u, 0u, 0u, 0u};
          vraw = u32x4{0u, 0u, 0u, 0u};
        }
        reduce_slot<R>(st, qf, kraw, vraw, valid);
      }
      g += 2 * p.n_att;
      cs.set_done(g);
      trace_ev(sh, c, l, 6, trc);
      wave_state_to_lds<R>(st, w, lane, sm_m, sm_l, sm_acc);
#if ENG_WIDE
      }
#endif
      cs.cbar();
      const uint32_t tp = tag_of(l, 2);
      const size_t bh = p.kvh;  // batch 1
      gu64* p_acc = G + a.g_part + (bh * L.n_splits + p.split) * R * DH;
      gu64* p_ml = G + a.g_part + (size_t)a.Hs * L.n_splits * R * DH + (bh * L.n_splits + p.split) * R * 2;
      for (int idx = w * 64 + lane; idx < R * DH; idx += NCONS * 64) {
        float A, M, Lsum;
        split_partial<R>(idx, sm_m, sm_l, sm_acc, A, M, Lsum);
        cs.publish(p_acc + idx, tp, __float_as_uint(A));
        if (idx % DH == 0) {
          cs.publish(p_ml + (idx / DH) * 2, tp, __float_as_uint(M));
          cs.publish(p_ml + (idx / DH) * 2 + 1, tp, __float_as_uint(Lsum));
        }
      }
    }
    trace_ev(sh, c, l, 7, trc);

    // ================================================================ split merge: this CU's slab of output pairs
    {
      const int ne = 2 * (p.e1 - p.e0), ns = L.n_splits;  // output elements of this CU (<= 64), splits
      if (ne > 0) {
        const uint32_t tp = tag_of(l, 2);
        // word t = (which * ns + sp) * ne + el: which = 0 acc, 1 m, 2 l of split sp for element 2 * e0 + el
        const gu64* part = G + a.g_part;
        const size_t ml_base = (size_t)a.Hs * ns * R * DH;
        sh.ctl[C_GATHERING] = GATHER_FLAG(4);
        cs.gather_fn<8>(3 * ns * ne, tp, cmb_lds, [&](int t) {
          const int el = t % ne, sp = (t / ne) % ns, which = t / (ne * ns);
          const int e = 2 * p.e0 + el, hh = e / DH, d = e % DH, kvh = hh / R, r = hh % R;
          const size_t blk = (size_t)kvh * ns + sp;
          return which == 0 ? part + blk * R * DH + (size_t)r * DH + d : part + ml_base + blk * R * 2 + r * 2 + (which - 1);
        });
        cs.cbar();
        sh.ctl[C_GATHERING] = 0;
        trace_ev(sh, c, l, 8, trc);
        if (w == 0) {
          const int el = min(lane, ne - 1);
          const lf32* ca = reinterpret_cast<const lf32*>(cmb_lds) + el;
          const lf32* cm = ca + ns * ne;
          const lf32* cl = cm + ns * ne;
          float mm[32], ll[32], vv[32];  // all reads independent: one LDS round trip instead of a chain of 64
#pragma unroll
          for (int sp = 0; sp < 32; ++sp) {
            const int s2 = min(sp, ns - 1) * ne;
            mm[sp] = cm[s2];
            ll[sp] = cl[s2];
            vv[sp] = ca[s2];
          }
          const float o = combine_splits<32>(mm, ll, vv, ns);
          const float o_hi = __shfl_down(o, 1, 64);
          if (lane < ne && !(lane & 1)) cs.publish(G + a.g_att + p.e0 + lane / 2, tag_of(l, 3), pack_bf2(o, o_hi));
        }
      }
    }

    // ================================================================ h1 = h + attn @ Wo^T
    trace_ev(sh, c, l, 9, trc);
    // the attention scratch is dead once wave 0 has merged
    cs.cbar();
    sh.ctl[C_GATHERING] = GATHER_FLAG(8);
    cs.gather(G + a.g_att, nq / 2, tag_of(l, 3), xs32);
    cs.cbar();
    sh.ctl[C_GATHERING] = 0;
    trace_ev(sh, c, l, 10, trc);
    {
      const int n_u = p.o1 - p.o0;
      for (int k = w; k < n_u; k += NCONS) {
        const uint32_t ga = g + (uint32_t)(2 * k) * PA;
        cs.set_done(ga);
        float vv[2];
        cs.template unit_dot<2, ALL4>(ga, PA, xs, vv);
        const float v0 = vv[0], v1 = vv[1];
        if (lane == 0) {
          const uint32_t rs = *reinterpret_cast<const lu32*>(sh.res + 2 * k);
          const uint32_t packed = pack_bf2(bf_lo(rs) + bf_round(v0), bf_hi(rs) + bf_round(v1));
          *reinterpret_cast<lu32*>(sh.res + 2 * k) = packed;  // residual of the W2 epilogue
          cs.publish(G + a.g_h1 + p.o0 + k, tag_of(l, 4), packed);
        }
      }
      g += (uint32_t)(2 * n_u) * PA;
      cs.set_done(g);
    }
    trace_ev(sh, c, l, 11, trc);

    // ================================================================ hid = silu(W1 x) * (W3 x), x = ffn_norm(h1)
    cs.norm_prefetch(nw, a.D, L.fn);
    cs.cbar();
    sh.ctl[C_GATHERING] = GATHER_FLAG(16);
    cs.norm_load_granules(xr, G + a.g_h1, a.D, tag_of(l, 4));
    sh.ctl[C_GATHERING] = 0;
#if ENG_QKV_HOLD == 2
    if (MOE && w == 0) sh.ctl[C_HGO] = (uint32_t)(l + 1);
#endif
    trace_ev(sh, c, l, 12, trc);
    cs.rmsnorm_store(xr, xs, a.D, nw, a.eps);
    trace_ev(sh, c, l, 13, trc);
    if constexpr (!MOE) {
      const int n_hold = holder_units(a, p.f1 - p.f0);
      if (n_hold && w == 0) sh.ctl[C_XREADY] = (uint32_t)(l + 1);  // (rmsnorm_store ends with a barrier of the consumer waves)
      {
        const int n_u = p.f1 - p.f0 - n_hold;  // the slab's last n_hold units belong to the holder waves
        for (int k = w; k < n_u; k += NCONS) {
          const uint32_t ga = g + (uint32_t)(4 * k) * PD;
          cs.set_done(ga);
          float vv[4];
          cs.template unit_dot<4, ALL4>(ga, PD, xs, vv);
          const float a0 = vv[0], b0 = vv[1], a1 = vv[2], b1 = vv[3];
          if (lane == 0) {
            const uint32_t packed = (uint32_t)f_to_bf(swiglu_bf(a0, b0)) | ((uint32_t)f_to_bf(swiglu_bf(a1, b1)) << 16);
            cs.publish(G + a.g_hid + p.f0 + k, tag_of(l, 5), packed);
          }
        }
        g += (uint32_t)(4 * n_u) * PD;
        cs.set_done(g);
      }
      trace_ev(sh, c, l, 14, trc);

      // ================================================================ h = h1 + hid @ W2^T
      cs.cbar();
      if (n_hold) {  // the holder waves read the activation region too: it is overwritten only when they are done with it
        hold_target += (uint32_t)n_hold;
        uint32_t spins = 0;
        while (sh.ctl[C_HDONE] < hold_target)
          if (!spin_ok(sh, spins, 0x500)) break;
      }
      sh.ctl[C_GATHERING] = GATHER_FLAG(32);
      cs.gather<14>(G + a.g_hid, a.F / 2, tag_of(l, 5), xs32);
      cs.cbar();
      sh.ctl[C_GATHERING] = 0;
      trace_ev(sh, c, l, 15, trc);
      {
        const int n_u = p.o1 - p.o0;
        const bool to_global = (l == a.n_layers - 1);
        for (int k = w; k < n_u; k += NCONS) {
          const uint32_t ga = g + (uint32_t)(2 * k) * PF;
          cs.set_done(ga);
          float vv[2];
          cs.template unit_dot<2, ALL4>(ga, PF, xs, vv);
          const float v0 = vv[0], v1 = vv[1];
          if (lane == 0) {
            const uint32_t rs = *reinterpret_cast<const lu32*>(sh.res + 2 * k);
            const uint32_t packed = pack_bf2(bf_lo(rs) + bf_round(v0), bf_hi(rs) + bf_round(v1));
            *reinterpret_cast<lu32*>(sh.res + 2 * k) = packed;  // residual of the next layer's Wo epilogue
            cs.publish(G + a.g_h + p.o0 + k, tag_of(l, 0), packed);
            if (to_global) *reinterpret_cast<uint32_t*>(a.h + 2 * (size_t)(p.o0 + k)) = packed;
          }
        }
        g += (uint32_t)(2 * n_u) * PF;
        cs.set_done(g);
      }
    } else {
      moe_ffn<ALL4>(a, sh, cs, L, p, l, c, w, lane, g, tag_of(l, 5), tag_of(l, 0), xr, trc);
    }
    trace_ev(sh, c, l, 16, trc);
    cs.cbar();
    trace_ev(sh, c, l, 17, trc);
  }

  // ================================================================ final norm + LM head (transformer.py:219,235,242)
  if (a.head) {
    typename Cons::NormW nw;
    cs.norm_prefetch(nw, a.D, a.final_norm);
    u32x4 xr[4];
    sh.ctl[C_GATHERING] = 1;
    cs.norm_load_granules(xr, G + a.g_h, a.D, tag_of(a.n_layers - 1, 0));
    sh.ctl[C_GATHERING] = 0;
    cs.rmsnorm_store(xr, xs, a.D, nw, a.eps);
    int v0, v1;
    slab(a.V / 2, c, a.NB, v0, v1);
    // greedy sampling rides on the LM head (generate.py:124-136 at temperature 0).  Inside the row loop the only extra work
    // is one LDS store of the two logits (a running max / sum-exp in the loop cost 18 us per step: it sits in the
    // dependency chain of every unit); the reduction runs once, afterwards, on wave 0.
    lf32* lg_lds = reinterpret_cast<lf32*>(sh.xs + (size_t)a.D * 2);  // behind the normalised activations: 2 (v1 - v0) floats
    const bool greedy = a.greedy_tok != nullptr;
    for (int k = w; k < v1 - v0; k += NCONS) {
      const uint32_t ga = g + (uint32_t)(2 * k) * PD;
      cs.set_done(ga);
      float vv[2];
      cs.template unit_dot<2, ALL4>(ga, PD, xs, vv);
      if (lane == 0) {
        const float y0 = bf_round(vv[0]), y1 = bf_round(vv[1]);
        *reinterpret_cast<float2*>(a.logits + 2 * (size_t)(v0 + k)) = make_float2(y0, y1);
        if (greedy) *reinterpret_cast<LDS_AS u32x2*>(lg_lds + 2 * k) = u32x2{__float_as_uint(y0), __float_as_uint(y1)};
      }
    }
    g += (uint32_t)(2 * (v1 - v0)) * PD;
    cs.set_done(0xffffffffu);
    if (greedy) {
      // workgroup partial (max, FIRST index of the max, sum exp(x - max)) -> three granules -> workgroup 0 reduces them all.
      // Ties: the lower index wins at every level (torch.argmax returns the first maximal element).
      cs.cbar();
      const uint32_t tg = tag_of(a.n_layers - 1, 6);
      if (w == 0) {
        const int n = 2 * (v1 - v0);
        float M = -INFINITY, S = 0.f;
        int I = 0x7fffffff;
        for (int i = lane; i < n; i += 64) {  // ascending indices per lane: `>` keeps the first maximum
          const float x = lg_lds[i];
          if (x > M) {
            S = S * __expf(M - x) + 1.f;
            M = x;
            I = 2 * v0 + i;
          } else {
            S += __expf(x - M);
          }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float m2 = __shfl_xor(M, o, 64), s2 = __shfl_xor(S, o, 64);
          const int i2 = __shfl_xor(I, o, 64);
          const bool take = m2 > M || (m2 == M && i2 < I);
          const float Mn = take ? m2 : M;
          S = S * (M == Mn ? 1.f : __expf(M - Mn)) + s2 * (m2 == Mn ? 1.f : __expf(m2 - Mn));
          M = Mn;
          I = take ? i2 : I;
        }
        if (lane < 3)
          cs.publish(G + a.g_amax + (size_t)c * 4 + lane, tg, lane == 0 ? __float_as_uint(M) : (lane == 1 ? (uint32_t)I : __float_as_uint(S)));
      }
      if (c == 0) {
        // staging behind the logits stash (wave 0 may still be reading the stash when waves 1-3 begin the sweep)
        lu32* red = reinterpret_cast<lu32*>(lg_lds + ((2 * ((a.V / 2 + a.NB - 1) / a.NB) + 3) & ~3));
        sh.ctl[C_GATHERING] = 1;
        cs.gather_fn<4>(4 * a.NB, tg, red, [&](int i) { return G + a.g_amax + ((i & 3) == 3 ? i - 3 : i); });
        cs.cbar();
        sh.ctl[C_GATHERING] = 0;
        if (w == 0) {
          float M = -INFINITY, S = 0.f;
          int I = 0x7fffffff;
          for (int cc = lane; cc < a.NB; cc += 64) {  // ascending workgroups = ascending vocabulary slabs
            const float m2 = __uint_as_float(red[cc * 4]), s2 = __uint_as_float(red[cc * 4 + 2]);
            const int i2 = (int)red[cc * 4 + 1];
            if (m2 > M || (m2 == M && i2 < I)) {
              S = S * __expf(M - m2) + s2;
              M = m2;
              I = i2;
            } else if (s2 > 0.f) {
              S += s2 * __expf(m2 - M);
            }
          }
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const float m2 = __shfl_xor(M, o, 64), s2 = __shfl_xor(S, o, 64);
            const int i2 = __shfl_xor(I, o, 64);
            const bool take = m2 > M || (m2 == M && i2 < I);
            const float Mn = take ? m2 : M;
            S = S * (M == Mn ? 1.f : __expf(M - Mn)) + s2 * (m2 == Mn ? 1.f : __expf(m2 - Mn));  // (-inf partials: lanes without work)
            M = Mn;
            I = take ? i2 : I;
          }
          if (lane == 0 && !sh.ctl[C_ABORT]) {
            greedy_token = I;
            greedy_logprob = -__logf(S);  // log_softmax at the argmax: x - max - log(sum exp(x - max)) with x == max
            greedy_valid = true;
          }
        }
      }
    }
  } else {
    cs.set_done(0xffffffffu);
  }

  // ================================================================ commit (workgroup 0, one lane): the step becomes visible
  // Everything a LATER launch reads to find its place - the position, the tag epoch, the token id when the caller chains
  // greedy_tok back in as ids - is written here, after the last all-to-all of the step: every workgroup read those words
  // at its entry, long before any workgroup can pass that edge.  An aborted step commits nothing.
  if (a.commit && c == 0 && w == 0 && lane == 0 && !sh.ctl[C_ABORT] &&
      __hip_atomic_load(sh.ctrl + G_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    const uint32_t step = sh.ctrl[G_STEPS];
    if (greedy_valid) {
      a.greedy_tok[0] = greedy_token;
      a.greedy_lp[0] = greedy_logprob;
      if (a.hist_tok && a.hist_len > 0) {
        a.hist_tok[step % (uint32_t)a.hist_len] = greedy_token;
        a.hist_lp[step % (uint32_t)a.hist_len] = greedy_logprob;
      }
    }
    a.kv_seqlens[0] = (int64_t)pos + 1;
    if (a.tok_pos) {
      a.q_start[0] = 0;
      a.q_start[1] = 1;
      a.kv_before[0] = pos;
      a.tok_seq[0] = 0;
      a.tok_pos[0] = pos;
    }
    sh.ctrl[G_STEPS] = step + 1;
    sh.ctrl[G_ARRIVE] = 0;  // every workgroup of this launch has been counted and the next launch has not begun: no wrap
    __hip_atomic_store(sh.ctrl + G_EPOCH, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------ holder waves
// Holder hi owns W1|W3 unit f1 - n_hold + hi of every layer (rows w1[2j], w3[2j], w1[2j+1], w3[2j+1]).  Same arithmetic
// as Cons::unit_dot<4>: per row, pieces in ascending order, four dot2_bf16 per piece, then wave_sum - bit-identical.
#if ENG_QKV_HOLD
// MoE models: holder hi keeps q|k|v units n_u - 6 + 2 hi and + 1 of the list (q.., k.., v..) of every layer, fetched during the
// PREVIOUS layer's router bubble (layer 0: at once).  Rows reduce as in Cons::unit_dot<2>; epilogue as in run_consumer.
__device__ __forceinline__ void run_qkv_holder(const EngArgs& a, const Shared& sh, int c, int hi, int lane, int pos, int seq, uint32_t epoch) {
  gu64* G = (gu64*)a.gran;
  const int PD = a.D >> 9;
  const lchar* xl = sh.xs + lane * 16;
  const int nq = a.H * DH, nkv = a.Hkv * DH;
  {
    LayerPlan p0;
    plan_layer(a, a.L[0], c, pos, p0);
    if (!qkv_held(a, (p0.q1 - p0.q0) + 2 * (p0.k1 - p0.k0))) return;  // (the same answer for every layer: shapes do not change)
  }
  u32x4 hw[HOLD_GROUPS][4][4];  // [group][row: unit 0 rows 0-1, unit 1 rows 0-1][piece in group]: constant indices only
  // iteration l: reduce and publish the units of layer l (held since iteration l - 1), then fetch those of layer l + 1
  for (int l = -1; l < a.n_layers; ++l) {
    uint32_t spins = 0;
    if (l >= 0) {
      const EngLayer& L = a.L[l];
      LayerPlan p;
      plan_layer(a, L, c, pos, p);
      const int n_u = (p.q1 - p.q0) + 2 * (p.k1 - p.k0);
      int kind0, u0, kind1, u1;
      qkv_unit(p, n_u - 2 * NHOLD + 2 * hi, kind0, u0);
      qkv_unit(p, n_u - 2 * NHOLD + 2 * hi + 1, kind1, u1);
      const float2 cs0 = *reinterpret_cast<const float2*>(a.rope_cs + ((size_t)pos * (DH >> 1) + (((2 * u0) % DH) >> 1)) * 2);
      const float2 cs1 = *reinterpret_cast<const float2*>(a.rope_cs + ((size_t)pos * (DH >> 1) + (((2 * u1) % DH) >> 1)) * 2);
      while (sh.ctl[C_XA] < (uint32_t)(l + 1))
        if (!spin_ok(sh, spins, 0x600)) return;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int grp = 0; grp < HOLD_GROUPS; ++grp)
        if (grp * 4 < PD) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 xv = lds16(xl + (grp * 4 + q) * PIECE);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[r] = dot2_bf16(hw[grp][r][q][i], xv[i], acc[r]);
          }
        }
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = wave_sum(acc[r]);
      if (lane == 0) {
        const uint32_t tag = (epoch << 12) | (uint32_t)((a.seq_base + l) * 8 + 1 + 1);
        auto finish = [&](int kind, int u, float2 cs2, float d0, float d1) {
          float y0 = bf_round(d0), y1 = bf_round(d1);
          if (kind < 2) {  // rope.py:13-23 on the adjacent pair
            float re, im;
            rope_pair(y0, y1, cs2.x, cs2.y, re, im);
            y0 = re;
            y1 = im;
          }
          const uint32_t packed = pack_bf2(y0, y1);
          if (kind > 0) {  // cache.py:83-92
            bf16_t* ring = (kind == 1 ? L.ck : L.cv) + kv_offset(L.kv_layout, L.W, nkv, DH, (size_t)seq, p.cur_slot, 2 * u);
            *reinterpret_cast<uint32_t*>(ring) = packed;
          }
          const int gi = (kind == 0 ? 0 : (kind == 1 ? nq / 2 : nq / 2 + nkv / 2)) + u;
          __hip_atomic_store(G + a.g_qkv + gi, ((unsigned long long)tag << 32) | packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        finish(kind0, u0, cs0, v[0], v[1]);
        finish(kind1, u1, cs1, v[2], v[3]);
        __hip_atomic_fetch_add((lu32*)(sh.ctl + C_HQDONE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (l + 1 < a.n_layers) {
      if (l >= 0) {
        spins = 0;
        // ENG_QKV_HOLD = 1: the loader has issued layer l's Wo rows (it flushes and waits for the router next); 2: this workgroup
        // has gathered h1 - the attention block's sweeps are over, the router's arithmetic begins
        while (sh.ctl[ENG_QKV_HOLD == 2 ? C_HGO : C_LSTAGE] < (uint32_t)(l + 1))
          if (!spin_ok(sh, spins, 0x600)) return;
      }
      const EngLayer& L = a.L[l + 1];
      LayerPlan p;
      plan_layer(a, L, c, pos, p);
      const int n_u = (p.q1 - p.q0) + 2 * (p.k1 - p.k0);
      int kind0, u0, kind1, u1;
      qkv_unit(p, n_u - 2 * NHOLD + 2 * hi, kind0, u0);
      qkv_unit(p, n_u - 2 * NHOLD + 2 * hi + 1, kind1, u1);
      const bf16_t* b0 = (kind0 == 0 ? L.wq : (kind0 == 1 ? L.wk : L.wv)) + (size_t)(2 * u0) * a.D + lane * 8;
      const bf16_t* b1 = (kind1 == 0 ? L.wq : (kind1 == 1 ? L.wk : L.wv)) + (size_t)(2 * u1) * a.D + lane * 8;
      const bf16_t* rows[4] = {b0, b0 + a.D, b1, b1 + a.D};
#pragma unroll
      for (int grp = 0; grp < HOLD_GROUPS; ++grp) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // 8 loads, then look whether this CU's consumers are sweeping a hand-off
          spins = 0;
          while (sh.ctl[C_GATHERING])
            if (!spin_ok(sh, spins, 0x600)) return;
#pragma unroll
          for (int r = 2 * half; r < 2 * half + 2; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) hw[grp][r][q] = ld16_nt(rows[r] + (size_t)min(grp * 4 + q, PD - 1) * 512);
        }
      }
    }
  }
}
#if ENG_WIDE == 1
// The same for rows of 9-12 pieces (wide build, Mixtral-8x22B): ONE unit per holder - list index n_u - 3 + hi - in three 4-piece groups.
__device__ __forceinline__ void run_qkv_holder1(const EngArgs& a, const Shared& sh, int c, int hi, int lane, int pos, int seq, uint32_t epoch) {
  gu64* G = (gu64*)a.gran;
  const int PD = a.D >> 9;
  const lchar* xl = sh.xs + lane * 16;
  const int nq = a.H * DH, nkv = a.Hkv * DH;
  {
    LayerPlan p0;
    plan_layer(a, a.L[0], c, pos, p0);
    if (!qkv_held(a, (p0.q1 - p0.q0) + 2 * (p0.k1 - p0.k0))) return;  // (the same answer for every layer: shapes do not change)
  }
  u32x4 hw[HOLD_GROUPS + 1][2][4];  // [group][row][piece in group]: constant indices only
  // iteration l: reduce and publish the units of layer l (held since iteration l - 1), then fetch those of layer l + 1
  for (int l = -1; l < a.n_layers; ++l) {
    uint32_t spins = 0;
    if (l >= 0) {
      const EngLayer& L = a.L[l];
      LayerPlan p;
      plan_layer(a, L, c, pos, p);
      const int n_u = (p.q1 - p.q0) + 2 * (p.k1 - p.k0);
      int kind0, u0;
      qkv_unit(p, n_u - NHOLD + hi, kind0, u0);
      const float2 cs0 = *reinterpret_cast<const float2*>(a.rope_cs + ((size_t)pos * (DH >> 1) + (((2 * u0) % DH) >> 1)) * 2);
      while (sh.ctl[C_XA] < (uint32_t)(l + 1))
        if (!spin_ok(sh, spins, 0x600)) return;
      float acc[2] = {0.f, 0.f};
#pragma unroll
      for (int grp = 0; grp < HOLD_GROUPS + 1; ++grp)
        if (grp * 4 < PD) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 xv = lds16(xl + (grp * 4 + q) * PIECE);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[r] = dot2_bf16(hw[grp][r][q][i], xv[i], acc[r]);
          }
        }
      float v[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) v[r] = wave_sum(acc[r]);
      if (lane == 0) {
        const uint32_t tag = (epoch << 12) | (uint32_t)((a.seq_base + l) * 8 + 1 + 1);
        auto finish = [&](int kind, int u, float2 cs2, float d0, float d1) {
          float y0 = bf_round(d0), y1 = bf_round(d1);
          if (kind < 2) {  // rope.py:13-23 on the adjacent pair
            float re, im;
            rope_pair(y0, y1, cs2.x, cs2.y, re, im);
            y0 = re;
            y1 = im;
          }
          const uint32_t packed = pack_bf2(y0, y1);
          if (kind > 0) {  // cache.py:83-92
            bf16_t* ring = (kind == 1 ? L.ck : L.cv) + kv_offset(L.kv_layout, L.W, nkv, DH, (size_t)seq, p.cur_slot, 2 * u);
            *reinterpret_cast<uint32_t*>(ring) = packed;
          }
          const int gi = (kind == 0 ? 0 : (kind == 1 ? nq / 2 : nq / 2 + nkv / 2)) + u;
          __hip_atomic_store(G + a.g_qkv + gi, ((unsigned long long)tag << 32) | packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        finish(kind0, u0, cs0, v[0], v[1]);
        __hip_atomic_fetch_add((lu32*)(sh.ctl + C_HQDONE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (l + 1 < a.n_layers) {
      if (l >= 0) {
        spins = 0;
        // ENG_QKV_HOLD = 1: the loader has issued layer l's Wo rows (it flushes and waits for the router next); 2: this workgroup
        // has gathered h1 - the attention block's sweeps are over, the router's arithmetic begins
        while (sh.ctl[ENG_QKV_HOLD == 2 ? C_HGO : C_LSTAGE] < (uint32_t)(l + 1))
          if (!spin_ok(sh, spins, 0x600)) return;
      }
      const EngLayer& L = a.L[l + 1];
      LayerPlan p;
      plan_layer(a, L, c, pos, p);
      const int n_u = (p.q1 - p.q0) + 2 * (p.k1 - p.k0);
      int kind0, u0;
      qkv_unit(p, n_u - NHOLD + hi, kind0, u0);
      const bf16_t* b0 = (kind0 == 0 ? L.wq : (kind0 == 1 ? L.wk : L.wv)) + (size_t)(2 * u0) * a.D + lane * 8;
      const bf16_t* rows[2] = {b0, b0 + a.D};
#pragma unroll
      for (int grp = 0; grp < HOLD_GROUPS + 1; ++grp) {
        {  // 8 loads, then look whether this CU's consumers are sweeping a hand-off
          spins = 0;
          while (sh.ctl[C_GATHERING])
            if (!spin_ok(sh, spins, 0x600)) return;
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) hw[grp][r][q] = ld16_nt(rows[r] + (size_t)min(grp * 4 + q, PD - 1) * 512);
        }
      }
    }
  }
}
#endif
#endif

__device__ __forceinline__ void run_holder(const EngArgs& a, const Shared& sh, int c, int hi, int lane, int pos, uint32_t epoch) {
  gu64* G = (gu64*)a.gran;
  const int PD = a.D >> 9;
  const lchar* xl = sh.xs + lane * 16;
  for (int l = 0; l < a.n_layers; ++l) {
    const EngLayer& L = a.L[l];
    LayerPlan p;
    plan_layer(a, L, c, pos, p);
    const int n_hold = holder_units(a, p.f1 - p.f0);
    if (hi >= n_hold) continue;
    const int j = p.f1 - n_hold + hi;
    uint32_t spins = 0;
    while (sh.ctl[C_LSTAGE] < (uint32_t)(l + 1))  // not before the layer's q|k|v, K/V and Wo streams are on their way
      if (!spin_ok(sh, spins, 0x600)) return;
    const size_t r0 = (size_t)(2 * j) * a.D + lane * 8;
    const bf16_t* rows[4] = {L.w1 + r0, L.w3 + r0, L.w1 + r0 + a.D, L.w3 + r0 + a.D};
    u32x4 hw[HOLD_GROUPS][4][4];  // [group][row][piece in group]: constant indices only -> registers
#pragma unroll
    for (int grp = 0; grp < HOLD_GROUPS; ++grp) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // 8 loads, then look whether this CU's consumers are sweeping a hand-off
        spins = 0;
        while (sh.ctl[C_GATHERING])
          if (!spin_ok(sh, spins, 0x600)) return;
#pragma unroll
        for (int r = 2 * half; r < 2 * half + 2; ++r) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int piece = min(grp * 4 + q, PD - 1);  // groups beyond the row are never used (holder_units: PD <= 8)
            hw[grp][r][q] = ld16_nt(rows[r] + (size_t)piece * 512);
          }
        }
      }
    }
    spins = 0;
    while (sh.ctl[C_XREADY] < (uint32_t)(l + 1))
      if (!spin_ok(sh, spins, 0x600)) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int grp = 0; grp < HOLD_GROUPS; ++grp)
      if (grp * 4 < PD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4 xv = lds16(xl + (grp * 4 + q) * PIECE);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[r] = dot2_bf16(hw[grp][r][q][i], xv[i], acc[r]);
        }
      }
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = wave_sum(acc[r]);
    if (lane == 0) {
      const uint32_t tag = (epoch << 12) | (uint32_t)((a.seq_base + l) * 8 + 5 + 1);
      const uint32_t packed = (uint32_t)f_to_bf(swiglu_bf(v[0], v[1])) | ((uint32_t)f_to_bf(swiglu_bf(v[2], v[3])) << 16);
      __hip_atomic_store(G + a.g_hid + j, ((unsigned long long)tag << 32) | packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the LDS reads of x above were consumed by the dots: the region may be overwritten once every holder says so
      __hip_atomic_fetch_add((lu32*)(sh.ctl + C_HDONE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// MOE is a separate instantiation: the dense kernel must not pay registers for the router / two-expert code (it sits at
// 247 of 256 VGPRs and spilled with the MoE path compiled in)
template <int R, bool MOE, bool ALL4>
__global__ __launch_bounds__(NTHREADS, 1) void decode_engine_kernel(const EngArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Shared sh;
  lchar* lds = (lchar*)smem;
  sh.ctl = reinterpret_cast<lvu32*>(lds);
  sh.res = reinterpret_cast<lbf16*>(lds + CTL_BYTES);
  sh.xs = lds + XS_OFF;
  sh.ring = lds + (LDS_TOTAL - a.ring_fills * FILL * PIECE);
  sh.ring_mask = a.ring_fills * FILL - 1;
  sh.ctrl = (gu32*)a.ctrl;
  sh.trace = (gu64*)a.trace;
  if (threadIdx.x < CTL_BYTES / 4) sh.ctl[threadIdx.x] = 0;
  __syncthreads();  // the only workgroup barrier: roles split below

  // A workspace whose status word is raised is poisoned until the host has dealt with it (Transformer._recover_engine):
  // later launches leave at once, before any side effect - the device state stays the one of the first failed step.
  if (__hip_atomic_load(sh.ctrl + G_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  // The step prepares itself (round 2 ran a decode_prep kernel first): position = kv_seqlens[0], batch row 0, tags of
  // epoch + 1.  Workgroup 0 commits position + 1 / epoch + 1 at the very end (run_consumer), so a launch that does not
  // complete leaves both untouched; the launches of one step (> 32 layers) all see the same values.
  const int pos = (int)a.kv_seqlens[0];
  const int seq = 0;
  const uint32_t epoch = (__hip_atomic_load(sh.ctrl + G_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u) & 0xfffffu;
  if (w == NCONS) run_loader<MOE>(a, sh, c, lane, pos, seq);
#if ENG_QKV_HOLD
#if ENG_WIDE == 1
  else if (w > NCONS && MOE && ((a.D >> 9) >> 2) == HOLD_GROUPS + 1) run_qkv_holder1(a, sh, c, w - NCONS - 1, lane, pos, seq, epoch);
#endif
  else if (w > NCONS && MOE) run_qkv_holder(a, sh, c, w - NCONS - 1, lane, pos, seq, epoch);
#endif
  else if (w > NCONS) run_holder(a, sh, c, w - NCONS - 1, lane, pos, epoch);
  else {
    // residency census: every workgroup counts itself in; consumers check the total before their first side effect
    uint32_t arrive_target = 0;
    if (w == 0) {
      uint32_t old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(sh.ctrl + G_ARRIVE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
      arrive_target = (old / (uint32_t)a.NB + 1u) * (uint32_t)a.NB;
      // test hook: wait for one workgroup more than exist - the gate fails exactly as it would with one missing
      if (__hip_atomic_load(sh.ctrl + G_SABOTAGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) arrive_target += 1u;
    }
    run_consumer<R, MOE, ALL4>(a, sh, c, w, lane, pos, seq, epoch, arrive_target);
  }
  // launches completed by the engine (one per <= 32 layers of a step): how a caller tells which path ran
  if (c == 0 && threadIdx.x == 0 && !sh.ctl[C_ABORT])
    __hip_atomic_fetch_add(sh.ctrl + G_LAUNCHES, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
namespace {
struct GranLayout {
  uint32_t g_h, g_qkv, g_att, g_h1, g_hid, g_hid2, g_part, g_amax, total;
};
constexpr int AMAX_MAX_NB = 1024;  // workgroups the greedy-sampling edge is sized for (decode_engine_applicable: NB <= 1024)
GranLayout gran_layout(int D, int H, int Hkv, int F, int max_splits) {
  using attn_core::DH;
  const int Rtot = H / Hkv, R = attn_decode_group(Rtot), Hs = Hkv * (Rtot / R);
  GranLayout g;
  uint32_t off = 0;
  g.g_h = off;    off += D / 2;
  g.g_qkv = off;  off += (H + 2 * Hkv) * DH / 2;
  g.g_att = off;  off += H * DH / 2;
  g.g_h1 = off;   off += D / 2;
  g.g_hid = off;  off += F / 2;
  g.g_hid2 = off; off += F / 2;  // MoE: hid of the second expert
  g.g_part = off; off += (uint32_t)((size_t)Hs * max_splits * R * (DH + 2));
  g.g_amax = off; off += 4 * AMAX_MAX_NB;  // (max logit, argmax, sum exp, pad) per workgroup
  g.total = off;
  return g;
}
}  // namespace

size_t decode_engine_granule_bytes(int D, int H, int Hkv, int F, int maxW) {
  if (Hkv <= 0 || H % Hkv) return 0;
  (void)maxW;  // sized for the maximum of 32 splits so that the layout depends on the model only
  return (size_t)gran_layout(D, H, Hkv, F, 32).total * 8;
}

bool decode_engine_applicable(const EngProblem& pr, char* why, size_t why_len) {
  auto no = [&](const char* m) {
    if (why) snprintf(why, why_len, "%s", m);
    return false;
  };
  using attn_core::DH;
  if (pr.D % 512 || pr.F % 512 || (pr.H * DH) % 512) return no("dim / hidden_dim / n_heads*128 not a multiple of 512");
  if (pr.D > 8192) return no("dim > 8192 (fused RMSNorm holds 4 pieces per thread)");
  // Large dims that are not a multiple of 2048 (Mistral-Nemo: 5120 = rows of 10 pieces, streamed in 2-piece groups, no holder
  // waves): 7.6-8.0 ms per step on the engine against 5.0 ms on the launch path (profiles/EXPERIMENTS.md) - such models
  // take the launch path.  Small dims (the parity tests) stay on the engine.
#if !ENG_WIDE
#if ENG_HEADLINE_ONLY != 2
  if (pr.D > 3072 && ((pr.D >> 9) & 3) != 0) return no("dim > 3072 and not a multiple of 2048: the launch path is faster (Nemo dims)");
#endif
#else
  // The wide build streams such rows as contiguous units (Loader::unit) - the third form tried for the Nemo dims, and still
  // slower than the launch path there: 5.47 vs 4.94 ms per step at an 8192-token context (profiles/EXPERIMENTS.md round 4;
  // 2-piece groups: 7.6-8.0 ms, 4 + 4 + 2 groups: 5.62 ms).  Declined unless the caller forces the wide build (traces, tests).
#if ENG_HEADLINE_ONLY != 2  // (the `nemo` build exists for exactly these dims)
  if (pr.D > 3072 && ((pr.D >> 9) & 3) != 0 && !pr.forced) return no("dim > 3072 and not a multiple of 2048: the launch path is faster (Nemo dims)");
#endif
  if (((pr.D >> 9) & 1) && pr.D > 3072) return no("odd number of 512-element pieces per row at a large dim");
#endif
  if (pr.V % 2) return no("odd vocab");
#if ENG_HEADLINE_ONLY == 1
  if (pr.E || pr.H != 4 * pr.Hkv || pr.D % 2048 || (pr.H * DH) % 2048 || pr.F % 2048)
    return no("this build instantiates the dense GQA-4 kernel for rows of 4-piece groups only");
#elif ENG_HEADLINE_ONLY == 2
  if (pr.E || pr.H != 4 * pr.Hkv || !(pr.D % 2048 || (pr.H * DH) % 2048 || pr.F % 2048) || pr.D <= 3072)
    return no("this build instantiates the dense GQA-4 kernel for large dims whose rows are not all 4-piece groups only (Mistral-Nemo)");
#elif ENG_WIDE == 2
  if (!pr.E) return no("the 8-fill MoE build takes MoE models only");
#endif
  const int kmax = pr.D > pr.F ? (pr.D > pr.H * DH ? pr.D : pr.H * DH) : (pr.F > pr.H * DH ? pr.F : pr.H * DH);
  const size_t region = (size_t)LDS_TOTAL - RING_FILLS * FILL * PIECE - XS_OFF;  // activation vector / attention scratch
  if ((size_t)kmax * 2 > region) return no("activation vector does not fit beside the 8-fill ring");
  if (pr.E) {
    if (pr.E > 16 || pr.top_k != 2) return no("MoE: at most 16 experts, top-2 routing");
#if ENG_WIDE
    if ((size_t)pr.D * 6 > region) return no("MoE: normalised + raw + router-normalised activation vector");
#else
    if ((size_t)pr.D * 4 > region) return no("MoE: normalised + raw activation vector");
#endif
    if ((size_t)pr.F * 2 + (size_t)((pr.D / 2 + pr.NB - 1) / pr.NB) * 8 > region) return no("MoE: hid vector + per-unit partial sums");
  }
  if ((size_t)pr.D * 2 + (size_t)((pr.V / 2 + pr.NB - 1) / pr.NB) * 8 + 16 + (size_t)pr.NB * 16 > region)
    return no("LM-head logits stash + greedy reduction staging");
  const int NB = pr.NB;
  if (NB < 8 || NB > 1024) return no("CU count");
  if (((pr.D / 2 + NB - 1) / NB) * 2 * 2 > RES_BYTES) return no("residual slab");
  if (pr.Hkv <= 0 || pr.H % pr.Hkv) return no("heads");
  const int Rtot = pr.H / pr.Hkv;
  const int R = attn_decode_group(Rtot);
#if ENG_WIDE == 2
  if (R != 4) return no("GQA group size (the 8-fill MoE build instantiates 4)");
#elif ENG_WIDE
  if (R != 4 && R != 6) return no("GQA group size (the wide build instantiates 4 and 6)");
#else
  if (R != 1 && R != 2 && R != 4 && R != 8) return no("GQA group size");
#endif
  const int Hs = pr.Hkv * (Rtot / R);
  const int ne_max = ((pr.H * DH / 2 + NB - 1) / NB) * 2;
  if (ne_max > 64) return no("split-merge slab");
#if ENG_WIDE
  // q (R*64 words) | k, v of this step (128 words) | m, l (8R floats) | acc (4 R DH floats) | merge staging (3 * 32 * ne words)
  if ((size_t)R * 256 + 512 + (size_t)R * 32 + (size_t)4 * R * DH * 4 + (size_t)3 * 32 * ne_max * 4 > region) return no("attention scratch");
#else
  if ((size_t)R * (256 + 2 * DH * 16 + 32) + 512 + (size_t)3 * 32 * ne_max * 4 > region) return no("attention scratch");
#endif
  for (int l = 0; l < pr.n_layers; ++l) {
    const int ns = attn_decode_splits(pr.W[l]);
    if (ns > 32 || Hs * ns > NB) return no("more attention work items than CUs");
#if ENG_WIDE
    // ratio 6 serves its heads in two passes over K/V pieces that must all sit in the LDS ring (run_consumer)
    if (R == 6 && 2 * ((attn_core::split_chunk(pr.W[l], ns) + 3) >> 2) > (RING_FILLS - 2) * FILL) return no("GQA ratio 6: ring longer than 5120 slots");
#endif
  }
  return true;
}

namespace {
uint64_t* g_trace = nullptr;
int g_thin = -1, g_depth = -1, g_holders = -1;
}
void decode_engine_set_holders(int on) { g_holders = on; }
void decode_engine_set_knobs(int thin, int depth) {
  g_thin = thin;
  g_depth = depth;
}
void decode_engine_set_trace(void* dev_buffer) { g_trace = (uint64_t*)dev_buffer; }
size_t decode_engine_trace_bytes(int NB) { return (size_t)NB * ENG_MAXL * TR_EVENTS * sizeof(uint64_t); }

// ---- residency census (once per device and process, outside any stream capture) ------------------------------------------
// The engine's hand-offs need all NB workgroups resident at once.  hipOccupancy... answers for an empty device; a CU mask,
// a partition mode or a co-tenant kernel changes the real answer, and a plain launch applies no check (MI355X_MICROARCH.md
// "Residency and cooperative launch").  So before the engine is used on a device the first time, a probe kernel with the
// engine's exact launch shape (512 threads, 160 KiB LDS) is run: every workgroup counts itself in and waits (bounded) for
// the others.  A failed census disables the engine for this device - the launch path is bit-identical and needs nothing.
// The per-step residency gate in the kernel covers what changes later (run_consumer).
__global__ __launch_bounds__(NTHREADS, 1) void engine_census_kernel(uint32_t* words, int nb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  (void)smem;
  if (threadIdx.x != 0) return;
  __hip_atomic_fetch_add(words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t polls = 0; polls < ARRIVE_POLLS; ++polls) {
    if (__hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (uint32_t)nb) return;
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_store(words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // somebody never showed up
}

namespace {
// 0 unknown, 1 passed, -1 failed; per device
int g_census[64] = {};
char g_census_why[160] = "";

int engine_census(int dev, int nb, uint32_t* ctrl, hipStream_t s) {
  if (dev < 0 || dev >= 64) return -1;
  if (g_census[dev] != 0) return g_census[dev];
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return 0;  // cannot probe now
  const void* fn = (const void*)engine_census_kernel;
  int per_cu = 0;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NTHREADS, LDS_TOTAL) != hipSuccess || per_cu < 1) {
    snprintf(g_census_why, sizeof(g_census_why), "occupancy query: %d workgroups of %d threads + %d B LDS per CU", per_cu, NTHREADS, LDS_TOTAL);
    return g_census[dev] = -1;
  }
  uint32_t* words = ctrl + 8;  // two spare control words of the caller's workspace (the ABI never allocates)
  uint32_t host[2] = {0, 0};
  bool ok = hipMemsetAsync(words, 0, 8, s) == hipSuccess;
  if (ok) {
    void* params[] = {(void*)&words, (void*)&nb};
    ok = hipLaunchKernel(fn, dim3(nb), dim3(NTHREADS), params, LDS_TOTAL, s) == hipSuccess &&
         hipMemcpyAsync(host, words, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (!ok || host[1] != 0 || host[0] != (uint32_t)nb) {
    snprintf(g_census_why, sizeof(g_census_why), "census: %u of %d workgroups resident together (CU mask / partition / co-tenant?)",
             host[0], nb);
    return g_census[dev] = -1;
  }
  return g_census[dev] = 1;
}
}  // namespace
const char* decode_engine_census_detail() { return g_census_why; }
void decode_engine_forget_census() { memset(g_census, 0, sizeof(g_census)); }

hipError_t launch_decode_engine(const EngProblem& pr, hipStream_t s, bool* declined) {
  if (declined) *declined = false;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (engine_census(dev, pr.NB, pr.ctrl, s) < 0) {  // not all workgroups can be resident: the caller takes the launch path
    if (declined) *declined = true;
    return hipSuccess;
  }
  EngArgs a;
  memset(&a, 0, sizeof(a));
  a.trace = (unsigned long long*)g_trace;
  if (g_thin < 0) {
    const char* e = getenv("MI_ENGINE_THIN");
    g_thin = e ? atoi(e) : 2;  // 0 stream through sweeps, 1 one fill in flight, 2 stop (measured best)
  }
  if (g_depth < 0) {
    const char* e = getenv("MI_ENGINE_DEPTH");
    g_depth = e ? atoi(e) : 2;  // measured: 2 fills in flight (plus the one being issued) beats 3
  }
  a.thin = g_thin;
  a.depth = g_depth;
  if (g_holders < 0) {
    const char* e = getenv("MI_ENGINE_HOLDERS");
    g_holders = e ? (atoi(e) != 0) : 1;
  }
  a.holders = g_holders;
  a.D = pr.D; a.H = pr.H; a.Hkv = pr.Hkv; a.F = pr.F; a.V = pr.V; a.eps = pr.eps; a.NB = pr.NB;
  const int Rtot = pr.H / pr.Hkv;
  a.R = attn_decode_group(Rtot);
  a.kv_groups = Rtot / a.R;
  a.Hs = pr.Hkv * a.kv_groups;
  a.ring_fills = RING_FILLS;
  a.h = (bf16_t*)pr.h; a.rope_cs = pr.rope_cs;
  a.kv_seqlens = pr.kv_seqlens; a.q_start = pr.q_start; a.kv_before = pr.kv_before; a.tok_seq = pr.tok_seq; a.tok_pos = pr.tok_pos;
  a.final_norm = (const bf16_t*)pr.final_norm; a.output = (const bf16_t*)pr.output; a.logits = pr.logits;
  a.gran = (uint64_t*)pr.granules; a.ctrl = pr.ctrl;
  const GranLayout gl = gran_layout(pr.D, pr.H, pr.Hkv, pr.F, 32);
  a.g_h = gl.g_h; a.g_qkv = gl.g_qkv; a.g_att = gl.g_att; a.g_h1 = gl.g_h1; a.g_hid = gl.g_hid; a.g_part = gl.g_part;
  a.g_amax = gl.g_amax; a.g_hid2 = gl.g_hid2; a.E = pr.E;
  if ((size_t)gl.total * 8 > pr.granule_bytes || !pr.kv_seqlens) return hipErrorInvalidValue;

  for (int l0 = 0; l0 < pr.n_layers; l0 += ENG_MAXL) {
    const int nl = pr.n_layers - l0 < ENG_MAXL ? pr.n_layers - l0 : ENG_MAXL;
    const bool last = l0 + nl == pr.n_layers;
    a.n_layers = nl;
    a.seq_base = l0;
    a.first = 1;  // every launch starts from the residual stream in global memory (the first one: or the embedding row)
    a.emb = l0 == 0 ? (const bf16_t*)pr.emb : nullptr;
    a.ids = pr.ids;
    a.commit = last;
    a.head = last && pr.logits != nullptr;
    const bool greedy = a.head && pr.greedy_tok && pr.greedy_lp;
    a.greedy_tok = greedy ? pr.greedy_tok : nullptr;
    a.greedy_lp = greedy ? pr.greedy_lp : nullptr;
    a.hist_tok = greedy && pr.hist_lp ? pr.hist_tok : nullptr;
    a.hist_lp = greedy && pr.hist_tok ? pr.hist_lp : nullptr;
    a.hist_len = pr.hist_len;
    for (int l = 0; l < nl; ++l) {
      const mi_layer_t& M = pr.layers[l0 + l];
      EngLayer& L = a.L[l];
      L.an = (const bf16_t*)M.attention_norm; L.wq = (const bf16_t*)M.wq; L.wk = (const bf16_t*)M.wk;
      L.wv = (const bf16_t*)M.wv; L.wo = (const bf16_t*)M.wo; L.fn = (const bf16_t*)M.ffn_norm;
      if (pr.E) {  // MoE layers: the gate in the w1 slot, the DEVICE table [E][3] of expert matrices in the w2 slot
        L.w1 = (const bf16_t*)M.gate; L.w2 = (const bf16_t*)M.expert_w_dev; L.w3 = nullptr;
      } else {
        L.w1 = (const bf16_t*)M.w1; L.w2 = (const bf16_t*)M.w2; L.w3 = (const bf16_t*)M.w3;
      }
      L.ck = (bf16_t*)pr.cache_k[l0 + l]; L.cv = (bf16_t*)pr.cache_v[l0 + l];
      L.W = pr.W[l0 + l];
      L.kv_layout = pr.kv_layout;
      L.n_splits = attn_decode_splits(L.W);
      L.chunk = attn_core::split_chunk(L.W, L.n_splits);
    }
    const void* fn = nullptr;
    const bool moe = pr.E > 0;
    const bool all4 = ENG_ALL4 && pr.D % 2048 == 0 && (pr.H * attn_core::DH) % 2048 == 0 && pr.F % 2048 == 0;
#define ENG_PICK(RR)                                                                                                       \
  (moe ? (all4 ? (const void*)decode_engine_kernel<RR, true, true> : (const void*)decode_engine_kernel<RR, true, false>) \
       : (all4 ? (const void*)decode_engine_kernel<RR, false, true> : (const void*)decode_engine_kernel<RR, false, false>))
    switch (a.R) {
#if ENG_HEADLINE_ONLY == 1
      case 4: fn = (!moe && all4) ? (const void*)decode_engine_kernel<4, false, true> : nullptr; break;
#elif ENG_HEADLINE_ONLY == 2
      case 4: fn = (!moe && !all4) ? (const void*)decode_engine_kernel<4, false, false> : nullptr; break;
#elif ENG_WIDE == 2
      case 4: fn = !moe ? nullptr : (all4 ? (const void*)decode_engine_kernel<4, true, true> : (const void*)decode_engine_kernel<4, true, false>); break;
#elif ENG_WIDE
      case 4: fn = ENG_PICK(4); break;
      case 6: fn = ENG_PICK(6); break;
#else
      case 1: fn = ENG_PICK(1); break;
      case 2: fn = ENG_PICK(2); break;
      case 4: fn = ENG_PICK(4); break;
      case 8: fn = ENG_PICK(8); break;
#endif
      default: return hipErrorInvalidValue;
    }
#undef ENG_PICK
    if (!fn) return hipErrorInvalidValue;
    // 160 KiB of dynamic LDS is an opt-in per function AND per device
    static bool attr_set[64][36] = {};
    const int slot = a.R + (moe ? 9 : 0) + (all4 ? 18 : 0);
    if (dev < 0 || dev >= 64 || !attr_set[dev][slot]) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_set[dev][slot] = true;
    }
    void* params[] = {(void*)&a};
    hipError_t e = hipLaunchKernel(fn, dim3(pr.NB), dim3(NTHREADS), params, LDS_TOTAL, s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
