// Internal (non-ABI) declarations shared by the kernel translation units and the runner.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mistral_hip.h"

typedef uint16_t bf16_t;

// ---------------------------------------------------------------------------------------------- GEMV
enum GemvMode {
  GEMV_STORE = 0,
  GEMV_RESIDUAL = 1,
  GEMV_SWIGLU = 2,
  GEMV_LOGITS = 3,
  GEMV_QKV_ROPE = 4,  // fused q|k|v projection + RoPE (+ ring write at decode)
  GEMV_MOE_W13 = 5,   // per (token, slot) expert gate/up projection + SwiGLU
  GEMV_MOE_W2 = 6     // per token: expert down projections, weighted bf16 combine, residual
};

constexpr int GEMV_MAX_T = 8;
// LDS a GEMV block may use for its T activation rows.  Round 5: up to 144 KiB (opt-in above 64 KiB per function) instead of 64:
// with hidden_dim 14336 a row is 28 KiB, and a batch of three sequences (mistral-demo) ran the W2 GEMV as TWO passes - the
// weights streamed twice per decode step (+24 us per layer).  Above 80 KiB one block fits a CU: the grid is one block per CU.
constexpr size_t GEMV_LDS_BUDGET = 144 * 1024;

struct GemvArgs {
  int mode;
  int T;                // tokens in this launch
  int K;                // contraction length (multiple of 8)
  int N;                // output rows (SWIGLU: hidden_dim)
  const bf16_t* x;      // [T, ldx]
  int ldx;
  const bf16_t* norm_w; // fused RMSNorm weight [K] or nullptr
  float eps;
  const bf16_t* w0;     // rows [0, n0)
  const bf16_t* w1;     // rows [n0, n1)   (SWIGLU: W3)
  const bf16_t* w2;     // rows [n1, N)
  int n0, n1;
  void* out;            // bf16 [T, ldo] (fp32 for LOGITS)
  int ldo;
  const bf16_t* residual;
  // QKV_ROPE
  const float* rope_cs; // fp32 [rope_len, head_dim/2, 2]
  const int32_t* tok_pos;
  const int32_t* tok_seq;  // nullptr: sequence id == token index
  int head_dim;
  int write_kv;
  void* cache_k;
  void* cache_v;
  int W;
  int kv_layout;  // MI_KV_SLOT_MAJOR / MI_KV_HEAD_MAJOR (common.cuh)
  // MoE
  const void* const* expert_tab;  // device [E][3]
  const int32_t* sel_idx;         // device [T*top_k]
  const float* sel_w;             // device [T*top_k]
  int top_k;
};

int gemv_max_tokens(int K);
hipError_t launch_gemv(const GemvArgs& a, hipStream_t s);
// gemv.hip compiled with -DGEMV_F16=1: the same kernels on fp16 payloads (mi_forward_generic: decode steps of fp16 models)
int gemv_max_tokens_f16(int K);
hipError_t launch_gemv_f16(const GemvArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------- GEMM
enum GemmEpi { GEMM_STORE = 0, GEMM_RESIDUAL = 1, GEMM_SWIGLU = 2, GEMM_LOGITS = 3, GEMM_LOGPROB = 4 };

struct GemmArgs {
  int epi;
  int M, N, K;          // N: output columns (SWIGLU: hidden_dim)
  const bf16_t* a;      // [M, lda]
  int lda;
  const bf16_t* w0;     // row segments as in GemvArgs (SWIGLU: w0 = W1, w1 = W3)
  const bf16_t* w1;
  const bf16_t* w2;
  int n0, n1;
  void* out;
  int ldo;
  const bf16_t* residual;
  // Token-grouped form (MoE prefill, moe.py:28-32): ONE launch covers every expert.  Compact rows are the (token,
  // slot) pairs sorted by expert; tile i of `tile_tab` is {expert, first compact row, valid rows (<= tile_rows), 0}.
  const int32_t* tile_tab;        // device [max_m_tiles][4] or nullptr (plain GEMM)
  const int32_t* n_tiles_ptr;     // device: number of valid entries in tile_tab
  int max_m_tiles;                // host upper bound (grid sizing)
  int tile_rows;                  // rows per m-tile the table was built for: 128 (gemm.hip) or 256 (gemm256.hip)
  const void* const* expert_tab;  // device [E][3] (w1, w2, w3) pointers
  int w_sel0, w_sel1;             // which of the three matrices feed w0 / w1 (w_sel1 < 0: unused)
  const int32_t* a_gather;        // device: A row of compact row r is a[a_gather[r]] (nullptr: a[r])
  // GEMM_LOGPROB (gemm256 only): nothing is stored; per (row, 256-column tile) the max and sum-exp of the bf16-rounded
  // logits go to lp_partial[row][n_tile] and the logit of column lp_target[row] to lp_tgt[row]
  const int32_t* lp_target;       // device [M], -1: none
  float2* lp_partial;             // device [M][ceil(N / 256)]
  float* lp_tgt;                  // device [M]
  // GEMM_STORE only: rotary embedding in the epilogue (transformer_layers.py:70 apply_rotary_emb on the bf16 q, k).  Output
  // columns n < rope_cols are (re, im) pairs of heads of rope_dh columns; pair i of a head is turned by the table entry
  // rope_cs[tok_pos[row]][i] = (cos, sin), exactly as rope_kernel does in a separate pass.  nullptr: plain store.
  const float* rope_cs;           // device fp32 [rope_len][rope_dh / 2][2]
  const int32_t* tok_pos;         // device [M]
  int rope_cols, rope_dh;         // rope_cols % rope_dh == 0, rope_dh % 4 == 0
  int rope_col0;                  // column of the whole problem at which this (column-sliced) launch starts (% rope_dh == 0)
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
bool gemm256_applicable(const GemmArgs& g);  // gemm256.hip: 256x256 tile for the large prefill shapes
hipError_t launch_gemm256(const GemmArgs& g, hipStream_t s);
bool gemm256_half_applicable(const GemmArgs& g);  // 128x256 tiles of the same kernel: the columns of a half-filled last round
hipError_t launch_gemm256_half(const GemmArgs& g, hipStream_t s);
void gemm_set_tail_mode(int mode);  // mi_debug_set_prefill_kernels
// gemm256.hip compiled with -DG256_F16=1: the same kernel on fp16 payloads (generic.hip: prefills of fp16 models)
bool gemm256_applicable_f16(const GemmArgs& g);
hipError_t launch_gemm256_f16(const GemmArgs& g, hipStream_t s);

// ---------------------------------------------------------------------------------------------- attention
struct AttnDecodeArgs {
  void* out;            // [B, H*Dh]
  const bf16_t* q;      // [B, ldq]
  int ldq;
  const bf16_t* cache_k;  // [maxB, W, Hkv*Dh] (kv_layout 0) or [maxB, Hkv, W, Dh] (1)
  const bf16_t* cache_v;
  int kv_layout;
  int W, B, H, Hkv, Dh;    // Hkv: kv heads AS SCHEDULED = real kv heads x kv_groups (launch_attn_decode sets both)
  int kv_groups;           // query-head groups per real kv head (1 unless the GQA ratio is split, see launch_attn_decode)
  const int32_t* tok_pos;  // [B]
  float* partial;          // scratch
  int32_t* tickets;        // reserved: first 4 KiB of the scratch (arrival counters of a removed in-kernel combine)
  int n_splits;
};
int attn_decode_splits(int W);
int attn_decode_group(int R);  // query heads served per block: the largest of {8, 6, 4, 2, 1} dividing the GQA ratio
size_t attn_decode_partial_floats(int B, int H, int Hkv, int Dh, int W);
hipError_t launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s);

struct AttnPrefillArgs {
  void* out;            // [T, H*Dh]
  const bf16_t* qkv;    // [T, ld]  q | k | v (post-RoPE)
  int ld;
  const bf16_t* cache_k;
  const bf16_t* cache_v;
  int kv_layout;  // MI_KV_SLOT_MAJOR / MI_KV_HEAD_MAJOR (common.cuh)
  int W, B, max_q_len, H, Hkv, Dh;
  const int32_t* q_start;    // [B+1]
  const int32_t* kv_before;  // [B]
  int causal;
  float scale;  // softmax scale (already validated: > 0)
};
hipError_t launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t s);
void attn_prefill_set_mode(int waves);  // mi_debug_set_prefill_kernels (negative: keep)
hipError_t launch_attn_prefill_f16(const AttnPrefillArgs& a, hipStream_t s);  // attn_prefill.hip compiled with -DATTN_F16=1

// Compute units of the current device (256 on a full MI355X; fewer in partitioned modes).  Grid-shaping heuristics
// (rounds of blocks, tail splitting) use it; results never depend on it.
inline int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

// ---------------------------------------------------------------------------------------------- elementwise
hipError_t launch_gelu(void* x, int ldx, int T, int N, hipStream_t s);
// log-softmax gather: out[m] = x_t - logsumexp(row m), from per-tile (max, sum-exp) partials or from a full logits row
hipError_t launch_logprob_finalize(float* out, const float2* partial, const float* tgt, int M, int n_tiles, hipStream_t s);
hipError_t launch_logprob_rows(float* out, const float* logits, int ld, const int32_t* target, int M, int V, hipStream_t s);
// bad_id (nullable): receives 1 + the index of an out-of-range token id (atomic max); such ids read row 0 / vocab-1
hipError_t launch_embedding(void* out, const void* table, const int64_t* ids, int T, int D, int vocab, uint32_t* bad_id,
                            hipStream_t s);
hipError_t launch_rmsnorm(void* out, const void* x, const void* w, int T, int D, float eps, hipStream_t s);
hipError_t launch_rope(void* qkv, int ld, int T, int H, int Hkv, int Dh, const float* rope_cs, const int32_t* tok_pos,
                       hipStream_t s);
hipError_t launch_kv_write(void* ck, void* cv, int W, const void* k, const void* v, int ld, int T, int kv_dim,
                           const int32_t* tok_seq, const int32_t* tok_pos, const int32_t* q_start, int kv_layout, int Dh, hipStream_t s);
// engine_ctrl (nullable): control words of the workspace - [0] step epoch of the persistent decode engine (incremented
// here), [2] its per-step abort broadcast (cleared here), [3] out-of-range token id flag (launch_embedding)
hipError_t launch_decode_prep(int64_t* kv_seqlens, int32_t* q_start, int32_t* kv_before, int32_t* tok_seq,
                              int32_t* tok_pos, int B, uint32_t* engine_ctrl, hipStream_t s);
hipError_t launch_decode_prep_embedding(int64_t* kv_seqlens, int32_t* q_start, int32_t* kv_before, int32_t* tok_seq,
                                        int32_t* tok_pos, int B, void* out, const void* table, const int64_t* ids, int D,
                                        int vocab, uint32_t* engine_ctrl, hipStream_t s);
hipError_t launch_add_rows(void* out, const void* a, const void* b, size_t n, hipStream_t s);
// Greedy sampling of B logits rows (generate.py:124-136 at temperature 0): tok[b] = first index of the row maximum
// (torch.argmax), lp[b] = log_softmax(row)[tok[b]]; also stored at entry (ctrl[5] - 1) % hist_len of the [hist_len, B]
// history rings when given (ctrl[5] = decode steps started on this workspace, advanced by decode_prep).
hipError_t launch_greedy_rows(const float* logits, int ld, int B, int V, int64_t* tok, float* lp, int64_t* hist_tok,
                              float* hist_lp, int hist_len, const uint32_t* ctrl, hipStream_t s);
// Nucleus sampling of B logits rows (generate.py:151-170 + the logprob of :134-136), csrc/sampling.hip: tok[b] drawn from
// softmax(row / temperature) restricted to the top-p prefix; uniforms (nullable): u[b] in [0, 1) instead of the Philox draw
// keyed by (seed; offset + ctrl[5], b); history rings as launch_greedy_rows.
hipError_t launch_sample_top_p(const float* logits, int ld, int B, int V, float temperature, float top_p, uint64_t seed,
                               uint64_t offset, const float* uniforms, int64_t* tok, float* lp, int64_t* hist_tok, float* hist_lp,
                               int hist_len, const uint32_t* ctrl, hipStream_t s);
// control words of a workspace after a raised engine status: status / abort / arrivals cleared, epoch advanced
hipError_t launch_engine_ctrl_reset(uint32_t* ctrl, hipStream_t s);

// ---------------------------------------------------------------------------------------------- MoE
hipError_t launch_moe_router(int32_t* sel_idx, float* sel_w, const void* x, int ldx, int T, int D, const void* gate,
                             int E, int top_k, const void* norm_w, float eps, hipStream_t s);
// Sort the (token, slot) pairs by expert: tok_of[r] = token of compact row r, row_of[t*top_k + kk] = compact row of
// that pair, tile_tab / n_tiles = the grouped GEMM's m-tile table (128 rows per tile, tiles never span experts).
hipError_t launch_moe_lists(const int32_t* sel_idx, int T, int E, int top_k, int32_t* tok_of, int32_t* row_of,
                            int32_t* tile_tab, int32_t* n_tiles, int tile_rows, hipStream_t s);
// out[t] = bf16(h[t] + R_t), R_t = sum over the token's experts in ascending id of bf16(w * y[row]), accumulated in
// bf16 from zero (moe.py:28-32 + transformer_layers.py:168)
hipError_t launch_moe_combine(void* out, const void* h, const void* y, const int32_t* sel_idx, const float* sel_w,
                              const int32_t* row_of, int T, int D, int top_k, hipStream_t s);


// ---------------------------------------------------------------------------------------------- persistent decode engine
// decode_engine.hip: one launch per (up to ENG_MAXL) layers of a batch-1 decode step on a dense model.
constexpr int ENG_MAXL = 32;  // layers per launch: bounded by the 4 KiB kernel-argument segment

struct EngLayer {
  const bf16_t *an, *wq, *wk, *wv, *wo, *fn, *w1, *w2, *w3;
  bf16_t *ck, *cv;
  int W, n_splits, chunk, kv_layout;  // kv_layout: MI_KV_SLOT_MAJOR / MI_KV_HEAD_MAJOR (common.cuh)
};

struct EngArgs {
  int n_layers, D, H, Hkv, F, V;
  int R, kv_groups, Hs;   // query heads per scheduled kv head; groups per real kv head; scheduled kv heads
  int NB, ring_fills;
  int seq_base;           // global index of this launch's first layer (tag sequence)
  int first, head;
  int thin, depth;        // loader knobs (A/B): during hand-off sweeps 0 stream / 1 one fill in flight / 2 stop; fills in flight (2 or 3)
  int holders;            // 1: holder waves take the last W1|W3 units of every CU's slab (A/B: MI_ENGINE_HOLDERS=0)
  int commit;             // 1: this launch ends the decode step: workgroup 0 advances kv_seqlens / epoch / step counter
  int hist_len;           // entries of the greedy history ring (0: none)
  float eps;
  bf16_t* h;              // [D] residual stream, in (first layer, unless emb) / out (last layer)
  const bf16_t* emb;      // token embedding table [V, D]: the first layer's input is row ids[0] (nullptr: read h)
  const int64_t* ids;     // [1] token id of this step (may alias greedy_tok: it is read before anything is committed)
  int64_t* kv_seqlens;    // [1] position of this step's token; + 1 at commit (cache.py:193-195)
  int32_t *q_start, *kv_before, *tok_seq, *tok_pos;  // the step's metadata words, written at commit for callers (launch-path layout)
  int64_t* greedy_tok;    // [1] argmax of the logits (first maximal index), or nullptr
  float* greedy_lp;       // [1] log_softmax(logits)[argmax]
  int64_t* hist_tok;      // [hist_len] ring indexed by the step counter (ctrl[5]) or nullptr
  float* hist_lp;
  const float* rope_cs;
  const bf16_t* final_norm;
  const bf16_t* output;
  float* logits;
  uint64_t* gran;         // granule regions
  uint32_t* ctrl;         // [0] epoch, [1] sticky status, [2] per-step abort
  uint32_t g_h, g_qkv, g_att, g_h1, g_hid, g_hid2, g_part, g_amax;
  int E;                  // experts (0: dense).  MoE layers: EngLayer.w1 = gate [E, D], .w2 = device table [E][3] of (w1, w2, w3)
  unsigned long long* trace;  // optional timeline buffer (debug)
  EngLayer L[ENG_MAXL];
};

struct EngProblem {
  int D, H, Hkv, F, V, n_layers, NB;
  int E, top_k;              // MoE (0, 0: dense)
  float eps;
  const mi_layer_t* layers;  // host
  void* const* cache_k;      // host [n_layers]
  void* const* cache_v;
  const int32_t* W;          // host [n_layers]
  int kv_layout;             // layout of every ring (common.cuh)
  void* h;
  const float* rope_cs;
  const void* emb;           // nullptr: h holds the step's input
  const int64_t* ids;
  int64_t* kv_seqlens;
  int32_t *q_start, *kv_before, *tok_seq, *tok_pos;  // decode metadata words (kept consistent for callers)
  const void* final_norm;
  const void* output;
  float* logits;             // nullptr: no LM head
  int64_t* greedy_tok;       // optional fused greedy sampling (needs logits)
  float* greedy_lp;
  int64_t* hist_tok;
  float* hist_lp;
  int hist_len;
  void* granules;
  size_t granule_bytes;
  uint32_t* ctrl;
  int forced;                // mi_debug_set_engine_variant(1): take shapes that measured slower than the launch path too (traces, tests)
};
static_assert(sizeof(EngArgs) <= 4096, "EngArgs must fit the kernel-argument segment");
size_t decode_engine_granule_bytes(int D, int H, int Hkv, int F, int maxW);
bool decode_engine_applicable(const EngProblem& pr, char* why, size_t why_len);
// *declined = true (and nothing enqueued): the residency census failed on this device - take the launch path
hipError_t launch_decode_engine(const EngProblem& pr, hipStream_t s, bool* declined);
const char* decode_engine_census_detail();
void decode_engine_forget_census();  // tests
void decode_engine_set_trace(void* dev_buffer);  // debug: nullptr disables
void decode_engine_set_knobs(int thin, int depth);  // debug / tuning
void decode_engine_set_holders(int on);             // debug / A/B: -1 = environment default

size_t decode_engine_trace_bytes(int NB);
// The same source compiled a second time with -DENG_WIDE=1 (decode_engine_wide.o): the shapes the shipped instantiations
// decline - GQA ratio 6 with a 32 KiB hid vector (Mixtral-8x22B: 7-fill ring), rows of an even number of pieces that is not a
// multiple of 4 (Mistral-Nemo: contiguous units).  Same EngProblem, same granule layout, bit-identical results.
bool decode_engine_applicable_wide(const EngProblem& pr, char* why, size_t why_len);
hipError_t launch_decode_engine_wide(const EngProblem& pr, hipStream_t s, bool* declined);
const char* decode_engine_census_detail_wide();
void decode_engine_forget_census_wide();
void decode_engine_set_trace_wide(void* dev_buffer);
void decode_engine_set_knobs_wide(int thin, int depth);
void decode_engine_set_holders_wide(int on);
// ... and a third time with -DENG_WIDE=2 (decode_engine_moe.o): the wide build's additions on the shipped 8-fill ring, for MoE
// models whose hid vector fits beside it (Mixtral-8x7B: the four consumers' W1|W3 units span exactly 8 fills at dim 4096).
bool decode_engine_applicable_moe(const EngProblem& pr, char* why, size_t why_len);
hipError_t launch_decode_engine_moe(const EngProblem& pr, hipStream_t s, bool* declined);
const char* decode_engine_census_detail_moe();
void decode_engine_forget_census_moe();
void decode_engine_set_trace_moe(void* dev_buffer);
void decode_engine_set_knobs_moe(int thin, int depth);
void decode_engine_set_holders_moe(int on);
// ... and a fourth time (decode_engine_next.o: -DENG_SUFFIX=_next -DENG_HEADLINE_ONLY=1 + build_native.ENGINE_NEXT_FLAGS:
// ENG_ABORT_RARE, ENG_CONS_PRIO, ENG_HOLD_STAGE=2, ENG_SADDR=2, ENG_TRACE=0, ENG_NOSTOP=32): the dense GQA-4 shapes with rows of
// 4-piece groups, i.e. the headline model.  The default object stays in the library (every other dense shape;
// MI_ENGINE_VARIANT=2 routes the headline to it for A/B).
bool decode_engine_applicable_next(const EngProblem& pr, char* why, size_t why_len);
hipError_t launch_decode_engine_next(const EngProblem& pr, hipStream_t s, bool* declined);
const char* decode_engine_census_detail_next();
void decode_engine_forget_census_next();
void decode_engine_set_trace_next(void* dev_buffer);
void decode_engine_set_knobs_next(int thin, int depth);
void decode_engine_set_holders_next(int on);
// ... and a fifth time (round 6, decode_engine_nemo.o: build_native.ENGINE_NEMO_FLAGS): dense GQA-4 models of a large dim whose
// rows are not multiples of 4 pieces - Mistral-Nemo (dim 5120).  Rounds 4-5 measured such dims slower on the wide build than on
// the launch path; round 6 found that build's dense kernels in the slow regime (hipcc drains the builtin-DMA queue in
// fill_begin: scripts/engine_loader_waits.py) - this build issues every DMA from inline asm.
bool decode_engine_applicable_nemo(const EngProblem& pr, char* why, size_t why_len);
hipError_t launch_decode_engine_nemo(const EngProblem& pr, hipStream_t s, bool* declined);
const char* decode_engine_census_detail_nemo();
void decode_engine_forget_census_nemo();
void decode_engine_set_trace_nemo(void* dev_buffer);
void decode_engine_set_knobs_nemo(int thin, int depth);
void decode_engine_set_holders_nemo(int on);

// ---------------------------------------------------------------------------------------------- generic storage dtype
// generic.hip: the operator sequence of the hot path for fp32 / fp16 storage (and for bf16 shapes the tuned kernels
// decline), every rounding point in the storage dtype as the reference executes it.  mi_forward_generic (api.hip).
enum { G_DT_BF16 = 0, G_DT_FP16 = 1, G_DT_FP32 = 2 };
enum { G_EPI_STORE = 0, G_EPI_RESIDUAL = 1, G_EPI_LOGITS = 2, G_EPI_SWIGLU = 3 };
struct GLinearArgs {
  const void* x;         // [M, K], row stride ldx
  int ldx;
  const void* w;         // [N, K] dense
  // --- the forms below exist on the M <= 8 kernel only (launch_g_linear refuses them otherwise; g_gemv_takes())
  const void* w1;        // optional second / third weight matrix: output columns [n0, n1) are rows of w1, [n1, N) rows of w2
  const void* w2;        //   (q | k | v in one launch).  G_EPI_SWIGLU: out[m, n] = silu(x . w[n]) * (x . w1[n]), N rows each
  int n0, n1;
  const void* norm_w;    // optional fused RMSNorm of x ([K] weights, `eps`): the contraction runs on round(round(x inv) w)
  float eps;
  void* out;             // [M, N] storage dtype (fp32 for G_EPI_LOGITS), row stride ldo
  int ldo;
  const void* residual;  // G_EPI_RESIDUAL: [M, N], row stride ldr (may alias out)
  int ldr;
  int M, N, K;
  int epi;
  const int32_t* active; // optional [M]: MoE row mask (nothing is done when no row of a tile is active)
};
struct GAttnArgs {
  void* out;             // [T, H * Dh]
  int ldo;
  const void* qkv;       // [T, ld]: q | k | v after RoPE
  int ld;
  const void* cache_k;   // rings [max_batch, W, Hkv, Dh] or head-major [max_batch, Hkv, W, Dh] (nullptr: cache=None call)
  const void* cache_v;
  int kv_layout;
  int W, T, H, Hkv, Dh;
  const int32_t* q_start;
  const int32_t* kv_before;
  const int32_t* tok_seq;
  const int32_t* tok_pos;
  int causal;            // 0: the cache=None call (every token sees every token, transformer_layers.py:165)
  float scale;
  float* partial;        // scratch of g_attn_partial_floats(T, H, Dh) floats: launches with few (token, head) pairs split the keys
  int B, max_q_len;      // sequences / longest new-token run of the forward (0: unknown - the MFMA prefill route is not taken)
};
size_t g_elem_bytes(int dt);
size_t g_attn_partial_floats(int T, int H, int Dh);
bool g_gemv_takes(int M, int K, int ldx);  // the M <= 8 kernel (fused norm / multi-matrix / SwiGLU forms) applies
bool g_linear_fused_ok(int dt, const GLinearArgs& g);  // a multi-matrix / SwiGLU request has a kernel (M <= 8 rows, or fp16 with >= 256 rows)
hipError_t launch_g_embedding(int dt, void* out, const void* table, const int64_t* ids, int T, int D, int vocab, uint32_t* bad_id,
                              hipStream_t s);
hipError_t launch_g_rmsnorm(int dt, void* out, const void* x, const void* w, int T, int D, float eps, hipStream_t s);
hipError_t launch_g_linear(int dt, const GLinearArgs& g, hipStream_t s);
hipError_t launch_g_rope(int dt, void* qkv, int ld, int T, int n_rot_cols, int Dh, const float* rope_cs, const int32_t* tok_pos,
                         hipStream_t s);
hipError_t launch_g_kv_write(int dt, void* ck, void* cv, int W, const void* k, const void* v, int ld, int T, int kv_dim,
                             const int32_t* tok_seq, const int32_t* tok_pos, const int32_t* q_start, int kv_layout, int Dh, hipStream_t s);
hipError_t launch_g_attention(int dt, const GAttnArgs& a, hipStream_t s);
hipError_t launch_g_swiglu(int dt, void* a, const void* b, int T, int F, const int32_t* active, hipStream_t s);
hipError_t launch_g_moe_topk(int dt, const void* logits, int T, int E, int k, int32_t* sel_idx, float* sel_w, hipStream_t s);
hipError_t launch_g_moe_mask(const int32_t* sel_idx, const float* sel_w, int T, int k, int e, int32_t* active, float* wt, hipStream_t s);
hipError_t launch_g_moe_accum(int dt, void* results, const void* y, const int32_t* active, const float* wt, int T, int D, hipStream_t s);
hipError_t launch_g_add(int dt, void* out, const void* a, const void* b, size_t n, hipStream_t s);
hipError_t launch_g_zero(int dt, void* p, size_t n, hipStream_t s);
hipError_t launch_g_gelu(int dt, void* x, int ldx, int T, int N, hipStream_t s);
