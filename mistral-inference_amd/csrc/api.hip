// C-ABI entry points of libmistral_hip.so (include/mistral_hip.h) and the layer-stack runner.
//
// mi_forward sequences, for every local layer, the launches that replace TransformerBlock.forward
// (reference transformer_layers.py:158-169) and, around the stack, Transformer.forward_partial /
// forward (transformer.py:163-242).  Per decode token and dense layer that is 6 launches:
//   [RMSNorm + Wq|Wk|Wv GEMV + RoPE + ring write] [split-KV GQA attention] [split combine] [Wo GEMV + residual]
//   [RMSNorm + W1|W3 GEMV + SiLU*mul] [W2 GEMV + residual]
// with no host synchronisation and no per-step host metadata (positions come from the device-resident
// kv_seqlens), so a decode step can also be captured in a hipGraph by the caller.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mistral_hip.h"
#include "../../include/mistral_hip_debug.h"
#include "kernels.h"

#ifdef MI_SLOT_LIST  // experiment libraries: see g_slots below
#define X(n)                                                                         \
  bool decode_engine_applicable_x##n(const EngProblem&, char*, size_t);             \
  hipError_t launch_decode_engine_x##n(const EngProblem&, hipStream_t, bool*);      \
  void decode_engine_set_trace_x##n(void*);                                          \
  void decode_engine_set_knobs_x##n(int, int);                                       \
  void decode_engine_set_holders_x##n(int);
MI_SLOT_LIST
#undef X
#endif

namespace {

thread_local char g_detail[512] = "";

bool kv_layout_ok(int layout) { return layout == MI_KV_SLOT_MAJOR || layout == MI_KV_HEAD_MAJOR; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_detail, sizeof(g_detail), fmt, ap);
  va_end(ap);
  return code;
}

inline int hip_rc(hipError_t e, const char* what) {
  if (e == hipSuccess) return MI_OK;
  snprintf(g_detail, sizeof(g_detail), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

#define MI_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc != MI_OK) return _rc; \
  } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

constexpr size_t TICKET_BYTES = 4096;  // first words: control block of the persistent decode engine (epoch, status, abort)

// 1 (default): batch-1 decode steps of dense models run on the persistent engine (decode_engine.hip) when the shapes
// allow it; 0: always the launch path.  MI_DECODE_ENGINE sets the initial value, mi_set_decode_engine changes it.
int g_engine_mode = -1;
bool nemo_engine_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MI_ENGINE_NEMO");
    on = e ? atoi(e) != 0 : 0;  // measured (round 6): 5.15-5.17 ms per step against 4.94-4.95 on the launch path at the Nemo-12B dims
  }
  return on != 0;
}
int g_engine_variant = -1;  // 0 (default): the shipped engine build first (MoE: the wide build first); 1: wide first; 2: shipped first
int engine_variant() {
  if (g_engine_variant < 0) {
    const char* e = getenv("MI_ENGINE_VARIANT");
    g_engine_variant = e ? atoi(e) : 0;
  }
  return g_engine_variant;
}
// Experiment libraries only (scripts/build_variants.py engine_slots -> -DMI_SLOT_LIST="X(0) X(1) ..."): further compiles of the
// engine source under the names *_x<N>, selected at run time by mi_debug_set_engine_slot (scripts/engine_ab.py: one process,
// one set of weights, every variant timed in turn on the same box).  Not compiled into the shipped library.
#ifdef MI_SLOT_LIST
struct EngSlot {
  bool (*applicable)(const EngProblem&, char*, size_t);
  hipError_t (*launch)(const EngProblem&, hipStream_t, bool*);
  void (*set_trace)(void*);
  void (*set_knobs)(int, int);
  void (*set_holders)(int);
};
#define X(n) {decode_engine_applicable_x##n, launch_decode_engine_x##n, decode_engine_set_trace_x##n, decode_engine_set_knobs_x##n, decode_engine_set_holders_x##n},
const EngSlot g_slots[] = {MI_SLOT_LIST};
#undef X
constexpr int N_SLOTS = (int)(sizeof(g_slots) / sizeof(g_slots[0]));
int g_slot = -1;
#endif
int engine_mode() {
  if (g_engine_mode < 0) {
    const char* e = getenv("MI_DECODE_ENGINE");
    g_engine_mode = e ? (atoi(e) != 0) : 1;
  }
  return g_engine_mode;
}

// MI_FUSE_ROPE=0: RoPE as a separate pass after the prefill q|k|v GEMM instead of in its epilogue (A/B testing)
bool fuse_rope_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MI_FUSE_ROPE");
    v = e ? (atoi(e) != 0) : 1;
  }
  return v != 0;
}

struct Workspace {
  int32_t* tickets;
  bf16_t* xn;       // [T, D]
  bf16_t* qkv;      // [T, (H + 2 Hkv) Dh]
  bf16_t* attn;     // [T, H Dh]
  bf16_t* hid;      // [T * max(1, top_k), F]
  float* partial;   // decode attention partials
  int32_t* sel_idx; // MoE: router picks [T, top_k]
  float* sel_w;
  int32_t* tok_of;   // token of compact row r            [T * top_k]
  int32_t* row_of;   // compact row of (token, slot)      [T * top_k]
  int32_t* tile_tab; // grouped-GEMM m-tile table         [max_tiles][4]
  int32_t* n_tiles;
  bf16_t* moe_y;     // expert outputs, compact rows      [T * top_k, D]
  void* gran;        // decode-engine granule regions (dense models)
  size_t gran_bytes;
  int max_tiles;
  size_t total;
};

Workspace carve(const mi_model_t* m, int T, int B, int maxW, char* base) {
  Workspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  const int qkv_cols = (m->n_heads + 2 * m->n_kv_heads) * m->head_dim;
  const int slots = m->top_k > 0 ? m->top_k : 1;
  w.tickets = (int32_t*)take(TICKET_BYTES);
  // engine granules at a FIXED offset (independent of T): nothing else ever writes them, so a stale word can never
  // look like a valid {value, tag} granule
  w.gran_bytes = (m->n_kv_heads > 0 && m->n_heads % m->n_kv_heads == 0)
                     ? decode_engine_granule_bytes(m->dim, m->n_heads, m->n_kv_heads, m->hidden_dim, maxW) : 0;
  w.gran = take(w.gran_bytes);
  w.xn = (bf16_t*)take((size_t)T * m->dim * 2);
  w.qkv = (bf16_t*)take((size_t)T * qkv_cols * 2);
  w.attn = (bf16_t*)take((size_t)T * m->n_heads * m->head_dim * 2);
  w.hid = (bf16_t*)take((size_t)T * slots * m->hidden_dim * 2);
  w.partial = (float*)take(attn_decode_partial_floats(B, m->n_heads, m->n_kv_heads, m->head_dim, maxW) * sizeof(float));
  w.sel_idx = (int32_t*)take((size_t)T * slots * 4);
  w.sel_w = (float*)take((size_t)T * slots * 4);
  w.max_tiles = (T * slots + 127) / 128 + (m->num_experts > 0 ? m->num_experts : 0);
  w.tok_of = (int32_t*)take((size_t)T * slots * 4);
  w.row_of = (int32_t*)take((size_t)T * slots * 4);
  w.tile_tab = (int32_t*)take((size_t)w.max_tiles * 16);
  w.n_tiles = (int32_t*)take(256);
  w.moe_y = (bf16_t*)take(m->num_experts > 0 ? (size_t)T * slots * m->dim * 2 : 0);
  w.total = off;
  return w;
}

// GEMV over T <= 8 tokens, in passes when T * K does not fit the LDS budget.
int gemv_passes(GemvArgs a, int T, hipStream_t s, const char* what) {
  const int cap = gemv_max_tokens(a.K);
  const size_t out_elt = (a.mode == GEMV_LOGITS) ? 4 : 2;
  for (int t0 = 0; t0 < T; t0 += cap) {
    GemvArgs p = a;
    p.T = (T - t0 < cap) ? T - t0 : cap;
    p.x = a.x + (size_t)t0 * a.ldx;
    p.out = (char*)a.out + (size_t)t0 * a.ldo * out_elt;
    if (a.residual) p.residual = a.residual + (size_t)t0 * a.ldo;
    if (a.tok_pos) p.tok_pos = a.tok_pos + t0;
    if (a.tok_seq) p.tok_seq = a.tok_seq + t0;
    MI_TRY(hip_rc(launch_gemv(p, s), what));
  }
  return MI_OK;
}

int check_model(const mi_model_t* m) {
  if (!m || !m->layers) return fail(MI_ERR_ARG, "null model");
  if (m->head_dim != 128) return fail(MI_ERR_SHAPE, "head_dim %d: kernels are built for 128", m->head_dim);
  if (m->n_heads % m->n_kv_heads) return fail(MI_ERR_SHAPE, "n_heads %% n_kv_heads != 0");
  if (m->dim % 8 || m->hidden_dim % 8) return fail(MI_ERR_SHAPE, "dim/hidden_dim must be multiples of 8");
  if (m->dim > 16384) return fail(MI_ERR_SHAPE, "dim > 16384");
  if (m->num_experts > 16 || m->top_k > 4 || (m->top_k == 3)) return fail(MI_ERR_SHAPE, "MoE: E <= 16, top_k in {1,2,4}");
  if (m->num_experts > 0 && (size_t)m->top_k * m->hidden_dim * 2 > 65536)
    return fail(MI_ERR_SHAPE, "MoE: top_k * hidden_dim too large for the decode combine kernel");
  return MI_OK;
}

}  // namespace

extern "C" {

int mi_abi_version(void) { return MI_ABI_VERSION; }

const char* mi_last_error_detail(void) { return g_detail; }

const char* mi_error_string(int code) {
  switch (code) {
    case MI_OK: return "ok";
    case MI_ERR_ARG: return "invalid argument";
    case MI_ERR_SHAPE: return "shape not supported by the gfx950 kernels";
    case MI_ERR_WORKSPACE: return "workspace too small";
    case MI_ERR_UNSUPPORTED: return "unsupported";
    case MI_ERR_RCCL: return "RCCL call failed (see mi_rccl_last_error)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

int mi_embedding(void* out, const void* table, const int64_t* ids, int T, int D, int vocab, mi_stream_t stream) {
  if (!out || !table || !ids || T <= 0 || D <= 0 || D % 8) return fail(MI_ERR_ARG, "mi_embedding");
  return hip_rc(launch_embedding(out, table, ids, T, D, vocab, nullptr, (hipStream_t)stream), "embedding");
}

int mi_rmsnorm(void* out, const void* x, const void* w, int T, int D, float eps, mi_stream_t stream) {
  if (!out || !x || !w || T <= 0 || D <= 0 || D % 8) return fail(MI_ERR_ARG, "mi_rmsnorm");
  return hip_rc(launch_rmsnorm(out, x, w, T, D, eps, (hipStream_t)stream), "rmsnorm");
}

int mi_rope_inplace(void* qkv, int ld, int T, int n_heads, int n_kv_heads, int head_dim, const float* rope_cs,
                    int rope_len, const int32_t* tok_pos, mi_stream_t stream) {
  if (!qkv || !rope_cs || !tok_pos || T <= 0 || head_dim % 8 || rope_len <= 0) return fail(MI_ERR_ARG, "mi_rope_inplace");
  return hip_rc(launch_rope(qkv, ld, T, n_heads, n_kv_heads, head_dim, rope_cs, tok_pos, (hipStream_t)stream), "rope");
}

int mi_kv_write(void* cache_k, void* cache_v, int W, const void* k, const void* v, int ld, int T, int kv_dim,
                const int32_t* tok_seq, const int32_t* tok_pos, const int32_t* q_start, int kv_layout, int head_dim,
                mi_stream_t stream) {
  if (!cache_k || !cache_v || !k || !v || !tok_seq || !tok_pos || !q_start || W <= 0 || T <= 0 || kv_dim % 8)
    return fail(MI_ERR_ARG, "mi_kv_write");
  if (!kv_layout_ok(kv_layout) || head_dim <= 0 || head_dim % 8 || kv_dim % head_dim) return fail(MI_ERR_ARG, "mi_kv_write: layout / head_dim");
  return hip_rc(launch_kv_write(cache_k, cache_v, W, k, v, ld, T, kv_dim, tok_seq, tok_pos, q_start, kv_layout, head_dim,
                                (hipStream_t)stream), "kv_write");
}

int mi_linear(void* out, int ldo, const void* x, int ldx, int M, int K, const void* const w[3], const int n_rows[3],
              int epilogue, const void* residual, const void* norm_w, float eps, mi_stream_t stream) {
  if (!out || !x || !w || !n_rows || !w[0] || M <= 0 || K <= 0 || K % 8) return fail(MI_ERR_ARG, "mi_linear");
  if (epilogue == MI_EPI_RESIDUAL && !residual) return fail(MI_ERR_ARG, "mi_linear: residual epilogue without residual");
  if (epilogue == MI_EPI_SWIGLU && (!w[1] || n_rows[0] != n_rows[1])) return fail(MI_ERR_ARG, "mi_linear: swiglu needs W1, W3");
  hipStream_t s = (hipStream_t)stream;
  const int n0 = n_rows[0], n1 = n0 + (w[1] ? n_rows[1] : 0), n2 = n1 + (w[2] ? n_rows[2] : 0);
  if (M <= GEMV_MAX_T) {
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.K = K; a.x = (const bf16_t*)x; a.ldx = ldx; a.norm_w = (const bf16_t*)norm_w; a.eps = eps;
    a.w0 = (const bf16_t*)w[0]; a.w1 = (const bf16_t*)w[1]; a.w2 = (const bf16_t*)w[2];
    a.out = out; a.ldo = ldo; a.residual = (const bf16_t*)residual;
    if (epilogue == MI_EPI_SWIGLU) {
      a.mode = GEMV_SWIGLU; a.N = n0; a.n0 = a.n1 = n0;
    } else {
      a.mode = epilogue == MI_EPI_STORE ? GEMV_STORE : epilogue == MI_EPI_RESIDUAL ? GEMV_RESIDUAL : GEMV_LOGITS;
      a.N = n2; a.n0 = n0; a.n1 = n1;
    }
    return gemv_passes(a, M, s, "gemv");
  }
  if (norm_w) return fail(MI_ERR_UNSUPPORTED, "mi_linear: fused RMSNorm only on the M <= 8 path");
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.K = K; g.a = (const bf16_t*)x; g.lda = ldx;
  g.w0 = (const bf16_t*)w[0]; g.w1 = (const bf16_t*)w[1]; g.w2 = (const bf16_t*)w[2];
  g.out = out; g.ldo = ldo; g.residual = (const bf16_t*)residual;
  if (epilogue == MI_EPI_SWIGLU) {
    g.epi = GEMM_SWIGLU; g.N = n0; g.n0 = g.n1 = n0;
  } else {
    g.epi = epilogue == MI_EPI_STORE ? GEMM_STORE : epilogue == MI_EPI_RESIDUAL ? GEMM_RESIDUAL : GEMM_LOGITS;
    g.N = n2; g.n0 = n0; g.n1 = n1;
  }
  return hip_rc(launch_gemm(g, s), "gemm");
}

namespace {
constexpr int LOGPROB_ROW_CHUNK = 128;  // rows per pass of the unfused route (bounds its scratch)
}

size_t mi_lm_head_logprobs_scratch_bytes(int M, int vocab) {
  if (M <= 0 || vocab <= 0) return 0;
  const size_t n_tiles = (size_t)(vocab + 255) / 256;
  const size_t fused = align_up((size_t)M * n_tiles * sizeof(float2)) + align_up((size_t)M * sizeof(float));
  const size_t rows = (size_t)(M < LOGPROB_ROW_CHUNK ? M : LOGPROB_ROW_CHUNK) * vocab * sizeof(float);
  return fused > rows ? fused : rows;  // either route may be taken (the fused one needs M >= 256 and K % 64 == 0)
}

int mi_lm_head_logprobs(float* logprob, const void* x, int ldx, int M, int K, const void* w, int vocab,
                        const int32_t* target, void* scratch, size_t scratch_bytes, mi_stream_t stream) {
  if (!logprob || !x || !w || !target || !scratch || M <= 0 || K <= 0 || K % 8 || vocab <= 0)
    return fail(MI_ERR_ARG, "mi_lm_head_logprobs");
  if (scratch_bytes < mi_lm_head_logprobs_scratch_bytes(M, vocab))
    return fail(MI_ERR_WORKSPACE, "mi_lm_head_logprobs: scratch %zu < required %zu", scratch_bytes,
                mi_lm_head_logprobs_scratch_bytes(M, vocab));
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.epi = GEMM_LOGPROB; g.M = M; g.N = vocab; g.K = K; g.a = (const bf16_t*)x; g.lda = ldx;
  g.w0 = (const bf16_t*)w; g.n0 = g.n1 = vocab;
  if (M >= 256 && gemm256_applicable(g)) {
    // one pass over the LM head: per-tile (max, sum-exp) partials + the target logit, then one wave per row
    const int n_tiles = (vocab + 255) / 256;
    g.lp_target = target;
    g.lp_partial = (float2*)scratch;
    g.lp_tgt = (float*)((char*)scratch + align_up((size_t)M * n_tiles * sizeof(float2)));
    MI_TRY(hip_rc(hipMemsetAsync(g.lp_tgt, 0, (size_t)M * sizeof(float), s), "logprob target memset"));
    MI_TRY(hip_rc(launch_gemm256(g, s), "lm head logprob gemm"));
    return hip_rc(launch_logprob_finalize(logprob, g.lp_partial, g.lp_tgt, M, n_tiles, s), "logprob finalize");
  }
  // few rows (or a K the 256-tile kernel does not take): fp32 logits of up to 128 rows at a time into the scratch
  // (GEMV / 128-tile GEMM), then a row-wise log-softmax gather
  const void* ws[3] = {w, nullptr, nullptr};
  const int nr[3] = {vocab, 0, 0};
  for (int r0 = 0; r0 < M; r0 += LOGPROB_ROW_CHUNK) {
    const int rows = (M - r0 < LOGPROB_ROW_CHUNK) ? M - r0 : LOGPROB_ROW_CHUNK;
    const int rc = mi_linear(scratch, vocab, (const bf16_t*)x + (size_t)r0 * ldx, ldx, rows, K, ws, nr, MI_EPI_LOGITS, nullptr,
                             nullptr, 0.f, stream);
    if (rc) return rc;
    MI_TRY(hip_rc(launch_logprob_rows(logprob + r0, (const float*)scratch, vocab, target + r0, rows, vocab, s), "logprob rows"));
  }
  return MI_OK;
}

size_t mi_attn_decode_scratch_bytes(int B, int n_heads, int n_kv_heads, int head_dim, int W) {
  return TICKET_BYTES + align_up(attn_decode_partial_floats(B, n_heads, n_kv_heads, head_dim, W) * sizeof(float));
}

int mi_attn_decode(void* out, const void* q, int ldq, const void* cache_k, const void* cache_v, int W, int B,
                   int n_heads, int n_kv_heads, int head_dim, const int32_t* tok_pos, void* scratch, int kv_layout,
                   mi_stream_t stream) {
  if (!out || !q || !cache_k || !cache_v || !tok_pos || !scratch || W <= 0 || B <= 0 || !kv_layout_ok(kv_layout))
    return fail(MI_ERR_ARG, "mi_attn_decode");
  if (head_dim != 128) return fail(MI_ERR_SHAPE, "head_dim must be 128");
  if ((size_t)B * n_kv_heads * 4 > TICKET_BYTES) return fail(MI_ERR_SHAPE, "B * n_kv_heads > 1024");
  AttnDecodeArgs a;
  a.out = out; a.q = (const bf16_t*)q; a.ldq = ldq; a.cache_k = (const bf16_t*)cache_k; a.cache_v = (const bf16_t*)cache_v;
  a.kv_layout = kv_layout;
  a.W = W; a.B = B; a.H = n_heads; a.Hkv = n_kv_heads; a.Dh = head_dim; a.tok_pos = tok_pos;
  a.tickets = (int32_t*)scratch; a.partial = (float*)((char*)scratch + TICKET_BYTES);
  a.n_splits = attn_decode_splits(W);
  return hip_rc(launch_attn_decode(a, (hipStream_t)stream), "attn_decode");
}

int mi_attn_prefill(void* out, const void* qkv, int ld, const void* cache_k, const void* cache_v, int W, int B,
                    int max_q_len, int n_heads, int n_kv_heads, int head_dim, const int32_t* q_start,
                    const int32_t* kv_before, int causal, float softmax_scale, int kv_layout, mi_stream_t stream) {
  if (!out || !qkv || B <= 0 || max_q_len <= 0 || W <= 0 || !kv_layout_ok(kv_layout)) return fail(MI_ERR_ARG, "mi_attn_prefill");
  if (causal && (!q_start || !kv_before)) return fail(MI_ERR_ARG, "mi_attn_prefill: metadata");
  if (head_dim != 128) return fail(MI_ERR_SHAPE, "head_dim must be 128");
  // the kernel forms 32-bit element offsets inside one ring (W * kv_dim) and inside the activation matrix (rows * ld)
  if ((size_t)W * n_kv_heads * head_dim >= (1ull << 31) || (size_t)B * max_q_len * (size_t)ld >= (1ull << 31))
    return fail(MI_ERR_UNSUPPORTED, "mi_attn_prefill: ring or activation matrix larger than 2^31 elements");
  AttnPrefillArgs a;
  a.out = out; a.qkv = (const bf16_t*)qkv; a.ld = ld; a.cache_k = (const bf16_t*)cache_k; a.cache_v = (const bf16_t*)cache_v;
  a.kv_layout = kv_layout;
  a.W = W; a.B = B; a.max_q_len = max_q_len; a.H = n_heads; a.Hkv = n_kv_heads; a.Dh = head_dim;
  a.q_start = q_start; a.kv_before = kv_before; a.causal = causal;
  a.scale = softmax_scale > 0.f ? softmax_scale : 1.0f / sqrtf((float)head_dim);
  return hip_rc(launch_attn_prefill(a, (hipStream_t)stream), "attn_prefill");
}

int mi_gelu(void* x, int ldx, int T, int N, mi_stream_t stream) {
  if (!x || T <= 0 || N <= 0 || ldx < N) return fail(MI_ERR_ARG, "mi_gelu");
  return hip_rc(launch_gelu(x, ldx, T, N, (hipStream_t)stream), "gelu");
}

int mi_moe_router(int32_t* sel_idx, float* sel_w, const void* x, int ldx, int T, int D, const void* gate, int E,
                  int top_k, const void* norm_w, float eps, mi_stream_t stream) {
  if (!sel_idx || !sel_w || !x || !gate || T <= 0 || D % 8) return fail(MI_ERR_ARG, "mi_moe_router");
  if (E > 16 || top_k > 4 || top_k > E || top_k < 1) return fail(MI_ERR_SHAPE, "router: E <= 16, top_k <= 4");
  return hip_rc(launch_moe_router(sel_idx, sel_w, x, ldx, T, D, gate, E, top_k, norm_w, eps, (hipStream_t)stream), "moe_router");
}

int mi_qkv_rope_kvwrite(void* qkv, int ldo, const void* x, int ldx, int T, int D, const void* wq, const void* wk,
                        const void* wv, int n_heads, int n_kv_heads, int head_dim, const void* norm_w, float eps,
                        const float* rope_cs, int rope_len, const int32_t* tok_pos, const int32_t* tok_seq, void* cache_k,
                        void* cache_v, int W, int kv_layout, mi_stream_t stream) {
  if (!qkv || !x || !wq || !wk || !wv || !rope_cs || !tok_pos || T <= 0 || D <= 0 || D % 8 || rope_len <= 0 || !kv_layout_ok(kv_layout))
    return fail(MI_ERR_ARG, "mi_qkv_rope_kvwrite");
  if (head_dim != 128) return fail(MI_ERR_SHAPE, "head_dim must be 128");
  if ((cache_k == nullptr) != (cache_v == nullptr) || (cache_k && W <= 0)) return fail(MI_ERR_ARG, "mi_qkv_rope_kvwrite: cache");
  if (T > GEMV_MAX_T)
    return fail(MI_ERR_UNSUPPORTED, "mi_qkv_rope_kvwrite: T = %d > %d (the prefill path is mi_rmsnorm + mi_linear + "
                "mi_rope_inplace + mi_kv_write)", T, GEMV_MAX_T);
  const int nq = n_heads * head_dim, nkv = n_kv_heads * head_dim;
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.mode = GEMV_QKV_ROPE; a.K = D; a.N = nq + 2 * nkv; a.x = (const bf16_t*)x; a.ldx = ldx;
  a.norm_w = (const bf16_t*)norm_w; a.eps = eps;
  a.w0 = (const bf16_t*)wq; a.w1 = (const bf16_t*)wk; a.w2 = (const bf16_t*)wv; a.n0 = nq; a.n1 = nq + nkv;
  a.out = qkv; a.ldo = ldo;
  a.rope_cs = rope_cs; a.tok_pos = tok_pos; a.tok_seq = tok_seq; a.head_dim = head_dim;
  a.write_kv = cache_k != nullptr; a.cache_k = cache_k; a.cache_v = cache_v; a.W = W; a.kv_layout = kv_layout;
  return gemv_passes(a, T, (hipStream_t)stream, "qkv gemv");
}

int mi_moe_experts_decode(void* out, const void* residual, const void* x, int ldx, int T, int D, int F,
                          const void* const* expert_w_dev, const int32_t* sel_idx, const float* sel_w, int top_k,
                          const void* norm_w, float eps, void* hidden_scratch, mi_stream_t stream) {
  if (!out || !residual || !x || !expert_w_dev || !sel_idx || !sel_w || !hidden_scratch || T <= 0 || D % 8 || F % 8)
    return fail(MI_ERR_ARG, "mi_moe_experts_decode");
  if (T > GEMV_MAX_T) return fail(MI_ERR_UNSUPPORTED, "mi_moe_experts_decode: T = %d > %d (use mi_moe_grouped_gemm)", T, GEMV_MAX_T);
  if (!(top_k == 1 || top_k == 2 || top_k == 4)) return fail(MI_ERR_SHAPE, "MoE: top_k in {1,2,4}");
  if ((size_t)top_k * F * 2 > 65536) return fail(MI_ERR_SHAPE, "MoE: top_k * hidden_dim too large for the decode combine kernel");
  hipStream_t s = (hipStream_t)stream;
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.mode = GEMV_MOE_W13; a.T = T; a.K = D; a.N = F; a.x = (const bf16_t*)x; a.ldx = ldx;
  a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.out = hidden_scratch; a.ldo = F;
  a.expert_tab = expert_w_dev; a.sel_idx = sel_idx; a.sel_w = sel_w; a.top_k = top_k;
  MI_TRY(hip_rc(launch_gemv(a, s), "moe w13 gemv"));
  memset(&a, 0, sizeof(a));
  a.mode = GEMV_MOE_W2; a.T = T; a.K = F; a.N = D; a.x = (const bf16_t*)hidden_scratch; a.ldx = F; a.out = out; a.ldo = D;
  a.residual = (const bf16_t*)residual;
  a.expert_tab = expert_w_dev; a.sel_idx = sel_idx; a.sel_w = sel_w; a.top_k = top_k;
  return hip_rc(launch_gemv(a, s), "moe w2 gemv");
}

namespace {
struct MoeScratch {
  bf16_t* hid;
  bf16_t* y;
  int32_t *tok_of, *row_of, *tile_tab, *n_tiles;
  int max_tiles;
  size_t total;
};
MoeScratch moe_carve(int T, int D, int F, int E, int top_k, char* base) {
  MoeScratch w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  const size_t rows = (size_t)T * top_k;
  w.hid = (bf16_t*)take(rows * F * 2);
  w.y = (bf16_t*)take(rows * D * 2);
  w.tok_of = (int32_t*)take(rows * 4);
  w.row_of = (int32_t*)take(rows * 4);
  w.max_tiles = (int)((rows + 127) / 128) + E;
  w.tile_tab = (int32_t*)take((size_t)w.max_tiles * 16);
  w.n_tiles = (int32_t*)take(256);
  w.total = off;
  return w;
}
}  // namespace

size_t mi_moe_grouped_gemm_scratch_bytes(int T, int D, int F, int E, int top_k) {
  if (T <= 0 || D <= 0 || F <= 0 || E <= 0 || top_k <= 0) return 0;
  return moe_carve(T, D, F, E, top_k, nullptr).total;
}

int mi_moe_grouped_gemm(void* out, const void* residual, const void* x, int ldx, int T, int D, int F, int E, int top_k,
                        const void* const* expert_w_dev, const int32_t* sel_idx, const float* sel_w, void* scratch,
                        size_t scratch_bytes, mi_stream_t stream) {
  if (!out || !residual || !x || !expert_w_dev || !sel_idx || !sel_w || !scratch || T <= 0 || D % 8 || F % 8 || ldx != D)
    return fail(MI_ERR_ARG, "mi_moe_grouped_gemm (x must be dense [T, D])");
  if (E > 16 || top_k > 4 || top_k > E || top_k < 1) return fail(MI_ERR_SHAPE, "MoE: E <= 16, top_k <= 4");
  MoeScratch w = moe_carve(T, D, F, E, top_k, (char*)scratch);
  if (w.total > scratch_bytes) return fail(MI_ERR_WORKSPACE, "mi_moe_grouped_gemm: scratch %zu < required %zu", scratch_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  const int k = top_k;
  // 256-row m-tiles (gemm256.hip) once an expert averages a few of them, else 128-row tiles (gemm.hip)
  const int tile_rows = ((long)T * k >= 512L * E && D % 64 == 0 && F % 64 == 0) ? 256 : 128;
  const int max_m_tiles = (T * k + tile_rows - 1) / tile_rows + E;
  MI_TRY(hip_rc(launch_moe_lists(sel_idx, T, E, k, w.tok_of, w.row_of, w.tile_tab, w.n_tiles, tile_rows, s), "moe_lists"));
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.epi = GEMM_SWIGLU; g.M = T * k; g.N = F; g.K = D; g.a = (const bf16_t*)x; g.lda = D; g.n0 = g.n1 = F;
  g.out = w.hid; g.ldo = F;
  g.tile_tab = w.tile_tab; g.n_tiles_ptr = w.n_tiles; g.max_m_tiles = max_m_tiles; g.tile_rows = tile_rows;
  g.expert_tab = expert_w_dev; g.w_sel0 = 0; g.w_sel1 = 2; g.a_gather = w.tok_of;
  MI_TRY(hip_rc(launch_gemm(g, s), "moe w13 grouped gemm"));
  memset(&g, 0, sizeof(g));
  g.epi = GEMM_STORE; g.M = T * k; g.N = D; g.K = F; g.a = w.hid; g.lda = F; g.n0 = g.n1 = D;
  g.out = w.y; g.ldo = D;
  g.tile_tab = w.tile_tab; g.n_tiles_ptr = w.n_tiles; g.max_m_tiles = max_m_tiles; g.tile_rows = tile_rows;
  g.expert_tab = expert_w_dev; g.w_sel0 = 1; g.w_sel1 = -1;
  MI_TRY(hip_rc(launch_gemm(g, s), "moe w2 grouped gemm"));
  return hip_rc(launch_moe_combine(out, residual, w.y, sel_idx, sel_w, w.row_of, T, D, k, s), "moe combine");
}

int mi_set_decode_engine(int enabled) {
  const int prev = engine_mode();
  g_engine_mode = enabled != 0;
  return prev;
}

int mi_decode_engine_reset(void* workspace, mi_stream_t stream) {
  if (!workspace) return fail(MI_ERR_ARG, "mi_decode_engine_reset");
  hipStream_t s = (hipStream_t)stream;
  // a raised status poisons the workspace (every later engine launch leaves at once): clear it, clear the abort broadcast
  // and the arrival count, and move to an epoch whose tags no granule of the failed step can carry
  MI_TRY(hip_rc(launch_engine_ctrl_reset(reinterpret_cast<uint32_t*>(workspace), s), "engine reset"));
  return MI_OK;
}

int mi_greedy_sample(const float* logits, int ld, int B, int vocab, int64_t* token, float* logprob, mi_stream_t stream) {
  if (!logits || !token || !logprob || B <= 0 || vocab <= 0 || ld < vocab) return fail(MI_ERR_ARG, "mi_greedy_sample");
  return hip_rc(launch_greedy_rows(logits, ld, B, vocab, token, logprob, nullptr, nullptr, 0, nullptr, (hipStream_t)stream),
                "greedy sample");
}

int mi_sample_top_p(const float* logits, int ld, int B, int vocab, float temperature, float top_p, uint64_t seed,
                    uint64_t offset, const float* uniforms, int64_t* token, float* logprob, mi_stream_t stream) {
  if (!logits || !token || !logprob || B <= 0 || vocab <= 0 || ld < vocab) return fail(MI_ERR_ARG, "mi_sample_top_p");
  if (!(temperature > 0.f) || !(top_p >= 0.f && top_p <= 1.f))
    return fail(MI_ERR_ARG, "mi_sample_top_p: temperature %g must be > 0 and top_p %g in [0, 1] (generate.py:163)", (double)temperature,
                (double)top_p);
  return hip_rc(launch_sample_top_p(logits, ld, B, vocab, temperature, top_p, seed, offset, uniforms, token, logprob, nullptr,
                                    nullptr, 0, nullptr, (hipStream_t)stream),
                "top-p sample");
}

int mi_debug_engine_sabotage(void* workspace, int launches, mi_stream_t stream) {
  if (!workspace || launches < 0) return fail(MI_ERR_ARG, "mi_debug_engine_sabotage");
  static thread_local uint32_t v;
  v = (uint32_t)launches;
  MI_TRY(hip_rc(hipMemcpyAsync((uint32_t*)workspace + 7, &v, 4, hipMemcpyHostToDevice, (hipStream_t)stream), "sabotage word"));
  return hip_rc(hipStreamSynchronize((hipStream_t)stream), "sabotage sync");
}

int mi_decode_engine_census(int forget) {
  if (forget) decode_engine_forget_census_next();
  if (forget) decode_engine_forget_census_nemo();
  if (forget) decode_engine_forget_census_wide();
  if (forget) decode_engine_forget_census_moe();
  if (forget) decode_engine_forget_census();
  return MI_OK;
}

int mi_decode_engine_status(const void* workspace, mi_stream_t stream, uint32_t status[8]) {
  if (!workspace || !status) return fail(MI_ERR_ARG, "mi_decode_engine_status");
  MI_TRY(hip_rc(hipMemcpyAsync(status, workspace, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream), "status copy"));
  return hip_rc(hipStreamSynchronize((hipStream_t)stream), "status sync");
}

size_t mi_debug_engine_trace_bytes(void) { return decode_engine_trace_bytes(device_cus()); }
int mi_debug_set_engine_knobs(int thin, int depth) {
  decode_engine_set_knobs(thin, depth);
  decode_engine_set_knobs_next(thin, depth);
  decode_engine_set_knobs_nemo(thin, depth);
#ifdef MI_SLOT_LIST
  for (const EngSlot& sl : g_slots) sl.set_knobs(thin, depth);
#endif
  decode_engine_set_knobs_wide(thin, depth);
  decode_engine_set_knobs_moe(thin, depth);
  return MI_OK;
}
int mi_debug_set_engine_holders(int on) {
  decode_engine_set_holders(on);
  decode_engine_set_holders_next(on);
  decode_engine_set_holders_nemo(on);
#ifdef MI_SLOT_LIST
  for (const EngSlot& sl : g_slots) sl.set_holders(on);
#endif
  decode_engine_set_holders_wide(on);
  decode_engine_set_holders_moe(on);
  return MI_OK;
}
int mi_debug_set_engine_variant(int variant) {
  const int prev = engine_variant();
  g_engine_variant = variant < 0 || variant > 3 ? 0 : variant;
  return prev;
}
#ifdef MI_SLOT_LIST
extern "C" int mi_debug_set_engine_slot(int slot) {  // -1: the library's own routing; returns the number of slots
  g_slot = slot < 0 || slot >= N_SLOTS ? -1 : slot;
  return N_SLOTS;
}
#endif
int mi_debug_set_prefill_kernels(int attn_waves, int gemm_tail) {
  attn_prefill_set_mode(attn_waves);
  if (gemm_tail >= 0) gemm_set_tail_mode(gemm_tail);
  return MI_OK;
}
int mi_debug_set_engine_trace(void* dev_buffer) {
  decode_engine_set_trace(dev_buffer);
  decode_engine_set_trace_next(dev_buffer);
  decode_engine_set_trace_nemo(dev_buffer);
#ifdef MI_SLOT_LIST
  for (const EngSlot& sl : g_slots) sl.set_trace(dev_buffer);
#endif
  decode_engine_set_trace_wide(dev_buffer);
  decode_engine_set_trace_moe(dev_buffer);
  return MI_OK;
}

size_t mi_workspace_bytes(const mi_model_t* model, int T, int B, int max_cache_size) {
  if (!model || T <= 0 || B <= 0) return 0;
  return carve(model, T, B, max_cache_size > 0 ? max_cache_size : 1, nullptr).total;
}

int mi_forward(const mi_model_t* m, const mi_batch_t* bt, mi_stream_t stream) {
  MI_TRY(check_model(m));
  if (!bt || bt->T <= 0 || bt->B <= 0 || !bt->h || !bt->workspace) return fail(MI_ERR_ARG, "mi_forward: batch");
  if (!bt->q_start || !bt->kv_before || !bt->tok_seq || !bt->tok_pos) return fail(MI_ERR_ARG, "mi_forward: metadata");
  const int T = bt->T, B = bt->B, branch = bt->branch;
  const bool has_cache = branch != MI_BRANCH_NOCACHE;
  if (has_cache && (!bt->cache_k || !bt->cache_v || !bt->cache_sizes)) return fail(MI_ERR_ARG, "mi_forward: cache");
  if (!kv_layout_ok(bt->kv_layout)) return fail(MI_ERR_ARG, "mi_forward: kv_layout");
  const int kvl = bt->kv_layout;
  if (branch == MI_BRANCH_DECODE && (T != B || !bt->kv_seqlens)) return fail(MI_ERR_ARG, "mi_forward: decode needs T == B");
  if ((size_t)B * m->n_kv_heads * 4 > TICKET_BYTES) return fail(MI_ERR_SHAPE, "B * n_kv_heads > 1024");
  if (bt->logits && (!m->final_norm || !m->output)) return fail(MI_ERR_ARG, "mi_forward: logits on a rank without LM head");
  hipStream_t s = (hipStream_t)stream;

  int maxW = 1;
  if (has_cache)
    for (int l = 0; l < m->n_layers; ++l) maxW = bt->cache_sizes[l] > maxW ? bt->cache_sizes[l] : maxW;
  Workspace ws = carve(m, T, B, maxW, (char*)bt->workspace);
  if (ws.total > bt->workspace_bytes)
    return fail(MI_ERR_WORKSPACE, "workspace %zu < required %zu", bt->workspace_bytes, ws.total);

  const int D = m->dim, H = m->n_heads, Hkv = m->n_kv_heads, Dh = m->head_dim, F = m->hidden_dim;
  const int nq = H * Dh, nkv = Hkv * Dh, qkv_cols = nq + 2 * nkv;
  const bool gemv = T <= GEMV_MAX_T;
  bf16_t* h = (bf16_t*)bt->h;

  // input_ids == NULL: h already holds this stage's input - received from the previous pipeline rank, or the multimodal
  // embeddings of transformer.py:190-191 (text rows from mi_embedding, image rows from the vision tower)
  const bool embed = m->tok_embeddings && bt->input_ids;
  uint32_t* engine_ctrl = reinterpret_cast<uint32_t*>(ws.tickets);
  const bool want_greedy = branch == MI_BRANCH_DECODE && bt->logits && bt->greedy_token && bt->greedy_logprob;
  if (bt->greedy_token && !want_greedy)
    return fail(MI_ERR_ARG, "mi_forward: greedy_token needs the DECODE branch, logits and greedy_logprob");
  if (want_greedy && bt->hist_len > 0 && (!bt->hist_token || !bt->hist_logprob))
    return fail(MI_ERR_ARG, "mi_forward: hist_len > 0 without history buffers");
  // ABI v5: the step's sample is a nucleus draw instead of the argmax (generate.py:126 at temperature > 0)
  const bool want_topp = want_greedy && bt->sample_temperature > 0.f;
  if (bt->sample_temperature < 0.f || (want_topp && !(bt->sample_top_p >= 0.f && bt->sample_top_p <= 1.f)))
    return fail(MI_ERR_ARG, "mi_forward: sample_temperature %g / sample_top_p %g", (double)bt->sample_temperature, (double)bt->sample_top_p);
  auto sample_step = [&]() -> int {  // behind the LM head of either decode path; reads the step counter the step advanced
    if (want_topp)
      return hip_rc(launch_sample_top_p(bt->logits, m->vocab_size, B, m->vocab_size, bt->sample_temperature, bt->sample_top_p,
                                        bt->sample_seed, bt->sample_offset, nullptr, bt->greedy_token, bt->greedy_logprob, bt->hist_token,
                                        bt->hist_logprob, bt->hist_len, engine_ctrl, s), "top-p sample");
    return hip_rc(launch_greedy_rows(bt->logits, m->vocab_size, B, m->vocab_size, bt->greedy_token, bt->greedy_logprob,
                                     bt->hist_token, bt->hist_logprob, bt->hist_len, engine_ctrl, s), "greedy sample");
  };

  // ---- batch-1 decode step of a dense model: every layer (and the LM head) in ONE persistent launch, which also does
  // the step's bookkeeping (position, embedding row, greedy sample): nothing else is enqueued for the token
  if (branch == MI_BRANCH_DECODE && T == 1 && B == 1 && m->n_layers > 0 && engine_mode()) {
    EngProblem pr;
    memset(&pr, 0, sizeof(pr));
    pr.D = D; pr.H = H; pr.Hkv = Hkv; pr.F = F; pr.V = m->vocab_size; pr.n_layers = m->n_layers; pr.NB = device_cus();
    pr.eps = m->norm_eps; pr.layers = m->layers; pr.cache_k = bt->cache_k; pr.cache_v = bt->cache_v; pr.W = bt->cache_sizes; pr.kv_layout = kvl;
    pr.h = h; pr.rope_cs = m->rope_cs;
    pr.emb = embed ? m->tok_embeddings : nullptr; pr.ids = bt->input_ids; pr.kv_seqlens = bt->kv_seqlens;
    pr.q_start = bt->q_start; pr.kv_before = bt->kv_before; pr.tok_seq = bt->tok_seq; pr.tok_pos = bt->tok_pos;
    pr.final_norm = m->final_norm; pr.output = m->output; pr.logits = bt->logits;
    if (want_greedy && !want_topp) {  // (a nucleus draw is its own small kernel behind the engine launch, below)
      pr.greedy_tok = bt->greedy_token; pr.greedy_lp = bt->greedy_logprob;
      pr.hist_tok = bt->hist_token; pr.hist_lp = bt->hist_logprob; pr.hist_len = bt->hist_len;
    }
    pr.granules = ws.gran; pr.granule_bytes = ws.gran_bytes; pr.ctrl = engine_ctrl;
    pr.E = m->num_experts; pr.top_k = m->top_k;
    pr.forced = engine_variant() == 1;
    bool dense_ok = true;
    for (int l = 0; l < m->n_layers; ++l)
      dense_ok = dense_ok && (m->num_experts ? (m->layers[l].gate && m->layers[l].expert_w_dev)
                                             : (m->layers[l].w1 && m->layers[l].w2 && m->layers[l].w3));
    // the shipped build of the engine first (the headline shapes); the "wide" build of the same source for what it declines
    // (GQA ratio 6 + 32 KiB hid vector: Mixtral-8x22B; rows of 10 pieces: Mistral-Nemo).  g_engine_variant = 1 (tests) prefers
    // the wide build wherever it applies, so that its code paths can be compared bit for bit at small sizes.
    // MoE models take the wide build wherever it applies: it carries the round-4 router (two experts per wave, batched loads:
    // -8..-11 us per layer), which the shipped object - frozen, see decode_engine.hip - does not.  Variant 2 = shipped build
    // first for every model (the A/B of that choice).
    // (MoE: the 8-fill MoE build where the model fits it - Mixtral-8x7B -, else the 7-fill wide build - Mixtral-8x22B.)
#ifdef MI_SLOT_LIST
    if (g_slot >= 0 && dense_ok && g_slots[g_slot].applicable(pr, nullptr, 0)) {
      bool declined = false;
      MI_TRY(hip_rc(g_slots[g_slot].launch(pr, s, &declined), "decode engine (experiment slot)"));
      if (!declined) {
        if (m->final_norm && !bt->logits) MI_TRY(hip_rc(launch_rmsnorm(h, h, m->final_norm, T, D, m->norm_eps, s), "final norm"));
        if (want_topp) MI_TRY(sample_step());
        return MI_OK;
      }
    }
#endif
    // the dense GQA-4 headline shapes take the `next` compile (build_native.ENGINE_NEXT_FLAGS: abort word read rarely, consumers at
    // s_setprio 1, holders fetch from the K/V stage on, every DMA from inline asm in the SGPR-base form, no stamp sites, the
    // loader not stopped during the hid sweep)
    const bool next_ok = dense_ok && m->num_experts == 0 && engine_variant() == 0 && decode_engine_applicable_next(pr, nullptr, 0);
    if (next_ok) {
      bool declined = false;
      MI_TRY(hip_rc(launch_decode_engine_next(pr, s, &declined), "decode engine"));
      if (!declined) {
        if (m->final_norm && !bt->logits) MI_TRY(hip_rc(launch_rmsnorm(h, h, m->final_norm, T, D, m->norm_eps, s), "final norm"));
        if (want_topp) MI_TRY(sample_step());
        return MI_OK;
      }
      snprintf(g_detail, sizeof(g_detail), "decode engine declined: %s", decode_engine_census_detail_next());  // informational
    }
    // large dims whose rows are not multiples of 4 pieces (Mistral-Nemo): the `nemo` compile - bit-equal, in a clean regime
    // (scripts/engine_loader_waits.py) and still 4 % slower than the launch path at those dims (contiguous 20- / 40-piece units:
    // four consumer waves cannot all hold one in the 128-piece ring), so it is opt-in: MI_ENGINE_NEMO=1 or engine variant 3 (tests)
    const bool nemo_ok = dense_ok && m->num_experts == 0 && (engine_variant() == 3 || (engine_variant() == 0 && nemo_engine_enabled())) &&
                         decode_engine_applicable_nemo(pr, nullptr, 0);
    if (nemo_ok) {
      bool declined = false;
      MI_TRY(hip_rc(launch_decode_engine_nemo(pr, s, &declined), "decode engine"));
      if (!declined) {
        if (m->final_norm && !bt->logits) MI_TRY(hip_rc(launch_rmsnorm(h, h, m->final_norm, T, D, m->norm_eps, s), "final norm"));
        if (want_topp) MI_TRY(sample_step());
        return MI_OK;
      }
      snprintf(g_detail, sizeof(g_detail), "decode engine declined: %s", decode_engine_census_detail_nemo());  // informational
    }
    const bool moe_ok = dense_ok && m->num_experts > 0 && engine_variant() == 0 && decode_engine_applicable_moe(pr, nullptr, 0);
    const bool wide_ok = !moe_ok && dense_ok && decode_engine_applicable_wide(pr, nullptr, 0);
    const bool wide_first = engine_variant() == 1 || (engine_variant() == 0 && m->num_experts > 0);
    const bool base_ok = !moe_ok && dense_ok && !(wide_first && wide_ok) && decode_engine_applicable(pr, nullptr, 0);
    if (moe_ok || base_ok || wide_ok) {
      bool declined = false;
      MI_TRY(hip_rc(moe_ok ? launch_decode_engine_moe(pr, s, &declined)
                           : (base_ok ? launch_decode_engine(pr, s, &declined) : launch_decode_engine_wide(pr, s, &declined)), "decode engine"));
      if (!declined) {
        if (m->final_norm && !bt->logits) MI_TRY(hip_rc(launch_rmsnorm(h, h, m->final_norm, T, D, m->norm_eps, s), "final norm"));
        if (want_topp) MI_TRY(sample_step());
        return MI_OK;
      }
      snprintf(g_detail, sizeof(g_detail), "decode engine declined: %s",
               moe_ok ? decode_engine_census_detail_moe() : (base_ok ? decode_engine_census_detail() : decode_engine_census_detail_wide()));  // informational
    }
  }

  if (branch == MI_BRANCH_DECODE && embed) {
    MI_TRY(hip_rc(launch_decode_prep_embedding(bt->kv_seqlens, bt->q_start, bt->kv_before, bt->tok_seq, bt->tok_pos, B, h,
                                               m->tok_embeddings, bt->input_ids, D, m->vocab_size, engine_ctrl, s),
                  "decode_prep+embedding"));
  } else {
    if (branch == MI_BRANCH_DECODE)
      MI_TRY(hip_rc(launch_decode_prep(bt->kv_seqlens, bt->q_start, bt->kv_before, bt->tok_seq, bt->tok_pos, B, engine_ctrl, s),
                    "decode_prep"));
    if (embed)
      MI_TRY(hip_rc(launch_embedding(h, m->tok_embeddings, bt->input_ids, T, D, m->vocab_size, engine_ctrl + 3, s), "embedding"));
  }

  for (int l = 0; l < m->n_layers; ++l) {
    const mi_layer_t& L = m->layers[l];
    const int W = has_cache ? bt->cache_sizes[l] : 1;
    void* ck = has_cache ? bt->cache_k[l] : nullptr;
    void* cv = has_cache ? bt->cache_v[l] : nullptr;

    // ---- attention_norm + q|k|v + RoPE (+ ring write at decode)
    if (gemv) {
      GemvArgs a;
      memset(&a, 0, sizeof(a));
      a.mode = GEMV_QKV_ROPE; a.K = D; a.N = qkv_cols; a.x = h; a.ldx = D;
      a.norm_w = (const bf16_t*)L.attention_norm; a.eps = m->norm_eps;
      a.w0 = (const bf16_t*)L.wq; a.w1 = (const bf16_t*)L.wk; a.w2 = (const bf16_t*)L.wv; a.n0 = nq; a.n1 = nq + nkv;
      a.out = ws.qkv; a.ldo = qkv_cols;
      a.rope_cs = m->rope_cs; a.tok_pos = bt->tok_pos; a.tok_seq = bt->tok_seq; a.head_dim = Dh;
      a.write_kv = branch == MI_BRANCH_DECODE; a.cache_k = ck; a.cache_v = cv; a.W = W; a.kv_layout = kvl;
      MI_TRY(gemv_passes(a, T, s, "qkv gemv"));
    } else {
      MI_TRY(hip_rc(launch_rmsnorm(ws.xn, h, L.attention_norm, T, D, m->norm_eps, s), "attention_norm"));
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.epi = GEMM_STORE; g.M = T; g.N = qkv_cols; g.K = D; g.a = ws.xn; g.lda = D;
      g.w0 = (const bf16_t*)L.wq; g.w1 = (const bf16_t*)L.wk; g.w2 = (const bf16_t*)L.wv; g.n0 = nq; g.n1 = nq + nkv;
      g.out = ws.qkv; g.ldo = qkv_cols;
      // RoPE rides on the GEMM's epilogue (same arithmetic as rope_kernel on the same bf16-rounded values; it was a
      // separate 20 us pass over q|k per layer at 4096 tokens).  Heads of a size the epilogues do not take: separate pass.
      const bool fused_rope = fuse_rope_enabled() && Dh % 16 == 0;
      if (fused_rope) { g.rope_cs = m->rope_cs; g.tok_pos = bt->tok_pos; g.rope_cols = nq + nkv; g.rope_dh = Dh; }
      MI_TRY(hip_rc(launch_gemm(g, s), "qkv gemm"));
      if (!fused_rope) MI_TRY(hip_rc(launch_rope(ws.qkv, qkv_cols, T, H, Hkv, Dh, m->rope_cs, bt->tok_pos, s), "rope"));
      // decode branch with more than 8 sequences: the ring write that the GEMV epilogue does otherwise; it must
      // precede the attention (cache.py:83-92 `update` then read, transformer_layers.py:77-81)
      if (branch == MI_BRANCH_DECODE)
        MI_TRY(hip_rc(launch_kv_write(ck, cv, W, ws.qkv + nq, ws.qkv + nq + nkv, qkv_cols, T, nkv, bt->tok_seq, bt->tok_pos,
                                      bt->q_start, kvl, Dh, s), "kv_write (decode)"));
    }

    // ---- attention
    if (branch == MI_BRANCH_DECODE) {
      AttnDecodeArgs a;
      a.out = ws.attn; a.q = ws.qkv; a.ldq = qkv_cols; a.cache_k = (const bf16_t*)ck; a.cache_v = (const bf16_t*)cv;
      a.kv_layout = kvl;
      a.W = W; a.B = B; a.H = H; a.Hkv = Hkv; a.Dh = Dh; a.tok_pos = bt->tok_pos;
      a.partial = ws.partial; a.tickets = ws.tickets; a.n_splits = attn_decode_splits(W);
      MI_TRY(hip_rc(launch_attn_decode(a, s), "attn_decode"));
    } else {
      AttnPrefillArgs a;
      a.out = ws.attn; a.qkv = ws.qkv; a.ld = qkv_cols; a.cache_k = (const bf16_t*)ck; a.cache_v = (const bf16_t*)cv;
      a.kv_layout = kvl;
      a.W = has_cache ? W : T; a.B = B; a.max_q_len = has_cache ? bt->max_q_len : T; a.H = H; a.Hkv = Hkv; a.Dh = Dh;
      a.q_start = bt->q_start; a.kv_before = bt->kv_before; a.causal = has_cache ? 1 : 0;
      a.scale = 1.0f / sqrtf((float)m->head_dim);
      MI_TRY(hip_rc(launch_attn_prefill(a, s), "attn_prefill"));
      if (has_cache)
        MI_TRY(hip_rc(launch_kv_write(ck, cv, W, ws.qkv + nq, ws.qkv + nq + nkv, qkv_cols, T, nkv, bt->tok_seq, bt->tok_pos,
                                      bt->q_start, kvl, Dh, s), "kv_write"));
    }

    // ---- h = h + attn @ Wo^T
    if (gemv) {
      GemvArgs a;
      memset(&a, 0, sizeof(a));
      a.mode = GEMV_RESIDUAL; a.K = nq; a.N = D; a.x = ws.attn; a.ldx = nq;
      a.w0 = (const bf16_t*)L.wo; a.n0 = a.n1 = D; a.out = h; a.ldo = D; a.residual = h;
      MI_TRY(gemv_passes(a, T, s, "wo gemv"));
    } else {
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.epi = GEMM_RESIDUAL; g.M = T; g.N = D; g.K = nq; g.a = ws.attn; g.lda = nq;
      g.w0 = (const bf16_t*)L.wo; g.n0 = g.n1 = D; g.out = h; g.ldo = D; g.residual = h;
      MI_TRY(hip_rc(launch_gemm(g, s), "wo gemm"));
    }

    // ---- h = h + FFN(ffn_norm(h))
    const bool moe = m->num_experts > 0;
    if (!moe) {
      if (gemv) {
        GemvArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = GEMV_SWIGLU; a.K = D; a.N = F; a.x = h; a.ldx = D;
        a.norm_w = (const bf16_t*)L.ffn_norm; a.eps = m->norm_eps;
        a.w0 = (const bf16_t*)L.w1; a.w1 = (const bf16_t*)L.w3; a.n0 = a.n1 = F; a.out = ws.hid; a.ldo = F;
        MI_TRY(gemv_passes(a, T, s, "w13 gemv"));
        memset(&a, 0, sizeof(a));
        a.mode = GEMV_RESIDUAL; a.K = F; a.N = D; a.x = ws.hid; a.ldx = F;
        a.w0 = (const bf16_t*)L.w2; a.n0 = a.n1 = D; a.out = h; a.ldo = D; a.residual = h;
        MI_TRY(gemv_passes(a, T, s, "w2 gemv"));
      } else {
        MI_TRY(hip_rc(launch_rmsnorm(ws.xn, h, L.ffn_norm, T, D, m->norm_eps, s), "ffn_norm"));
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.epi = GEMM_SWIGLU; g.M = T; g.N = F; g.K = D; g.a = ws.xn; g.lda = D;
        g.w0 = (const bf16_t*)L.w1; g.w1 = (const bf16_t*)L.w3; g.n0 = g.n1 = F; g.out = ws.hid; g.ldo = F;
        MI_TRY(hip_rc(launch_gemm(g, s), "w13 gemm"));
        memset(&g, 0, sizeof(g));
        g.epi = GEMM_RESIDUAL; g.M = T; g.N = D; g.K = F; g.a = ws.hid; g.lda = F;
        g.w0 = (const bf16_t*)L.w2; g.n0 = g.n1 = D; g.out = h; g.ldo = D; g.residual = h;
        MI_TRY(hip_rc(launch_gemm(g, s), "w2 gemm"));
      }
    } else {
      const int E = m->num_experts, k = m->top_k;
      if (!L.gate || !L.expert_w_dev || !L.expert_w_host) return fail(MI_ERR_ARG, "mi_forward: MoE layer tables");
      if (gemv) {
        MI_TRY(hip_rc(launch_moe_router(ws.sel_idx, ws.sel_w, h, D, T, D, L.gate, E, k, L.ffn_norm, m->norm_eps, s), "moe_router"));
        GemvArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = GEMV_MOE_W13; a.T = T; a.K = D; a.N = F; a.x = h; a.ldx = D;
        a.norm_w = (const bf16_t*)L.ffn_norm; a.eps = m->norm_eps; a.out = ws.hid; a.ldo = F;
        a.expert_tab = L.expert_w_dev; a.sel_idx = ws.sel_idx; a.sel_w = ws.sel_w; a.top_k = k;
        MI_TRY(hip_rc(launch_gemv(a, s), "moe w13 gemv"));
        memset(&a, 0, sizeof(a));
        a.mode = GEMV_MOE_W2; a.T = T; a.K = F; a.N = D; a.x = ws.hid; a.ldx = F; a.out = h; a.ldo = D; a.residual = h;
        a.expert_tab = L.expert_w_dev; a.sel_idx = ws.sel_idx; a.sel_w = ws.sel_w; a.top_k = k;
        MI_TRY(hip_rc(launch_gemv(a, s), "moe w2 gemv"));
      } else {
        MI_TRY(hip_rc(launch_rmsnorm(ws.xn, h, L.ffn_norm, T, D, m->norm_eps, s), "ffn_norm"));
        MI_TRY(hip_rc(launch_moe_router(ws.sel_idx, ws.sel_w, ws.xn, D, T, D, L.gate, E, k, nullptr, 0.f, s), "moe_router"));
        // 256-row m-tiles (gemm256.hip) once an expert averages a few of them, else 128-row tiles (gemm.hip)
        const int tile_rows = ((long)T * k >= 512L * E && D % 64 == 0 && F % 64 == 0) ? 256 : 128;
        const int max_m_tiles = (T * k + tile_rows - 1) / tile_rows + E;  // <= ws.max_tiles (sized for 128-row tiles)
        MI_TRY(hip_rc(launch_moe_lists(ws.sel_idx, T, E, k, ws.tok_of, ws.row_of, ws.tile_tab, ws.n_tiles, tile_rows, s), "moe_lists"));
        // one token-grouped launch per projection covers all experts (tile table built on the device: no host sync)
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.epi = GEMM_SWIGLU; g.M = T * k; g.N = F; g.K = D; g.a = ws.xn; g.lda = D; g.n0 = g.n1 = F;
        g.out = ws.hid; g.ldo = F;
        g.tile_tab = ws.tile_tab; g.n_tiles_ptr = ws.n_tiles; g.max_m_tiles = max_m_tiles; g.tile_rows = tile_rows;
        g.expert_tab = L.expert_w_dev; g.w_sel0 = 0; g.w_sel1 = 2; g.a_gather = ws.tok_of;
        MI_TRY(hip_rc(launch_gemm(g, s), "moe w13 grouped gemm"));
        memset(&g, 0, sizeof(g));
        g.epi = GEMM_STORE; g.M = T * k; g.N = D; g.K = F; g.a = ws.hid; g.lda = F; g.n0 = g.n1 = D;
        g.out = ws.moe_y; g.ldo = D;
        g.tile_tab = ws.tile_tab; g.n_tiles_ptr = ws.n_tiles; g.max_m_tiles = max_m_tiles; g.tile_rows = tile_rows;
        g.expert_tab = L.expert_w_dev; g.w_sel0 = 1; g.w_sel1 = -1;
        MI_TRY(hip_rc(launch_gemm(g, s), "moe w2 grouped gemm"));
        MI_TRY(hip_rc(launch_moe_combine(h, h, ws.moe_y, ws.sel_idx, ws.sel_w, ws.row_of, T, D, k, s), "moe combine"));
      }
    }
  }

  // ---- final norm (+ LM head)
  if (m->final_norm) {
    if (bt->logits) {
      if (gemv) {
        GemvArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = GEMV_LOGITS; a.K = D; a.N = m->vocab_size; a.x = h; a.ldx = D;
        a.norm_w = (const bf16_t*)m->final_norm; a.eps = m->norm_eps;
        a.w0 = (const bf16_t*)m->output; a.n0 = a.n1 = m->vocab_size; a.out = bt->logits; a.ldo = m->vocab_size;
        MI_TRY(gemv_passes(a, T, s, "lm head gemv"));
        if (want_greedy)  // generate.py:124-136 at temperature 0, fused behind the LM head (one block per sequence)
          MI_TRY(sample_step());
      } else {
        MI_TRY(hip_rc(launch_rmsnorm(ws.xn, h, m->final_norm, T, D, m->norm_eps, s), "final norm"));
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.epi = GEMM_LOGITS; g.M = T; g.N = m->vocab_size; g.K = D; g.a = ws.xn; g.lda = D;
        g.w0 = (const bf16_t*)m->output; g.n0 = g.n1 = m->vocab_size; g.out = bt->logits; g.ldo = m->vocab_size;
        MI_TRY(hip_rc(launch_gemm(g, s), "lm head gemm"));
        if (want_greedy)
          MI_TRY(sample_step());
      }
    } else {
      MI_TRY(hip_rc(launch_rmsnorm(h, h, m->final_norm, T, D, m->norm_eps, s), "final norm"));
    }
  }
  return MI_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- generic storage dtype
// mi_forward for fp32 / fp16 storage (and for bf16 models of a shape check_model declines): launch by launch over the
// kernels of generic.hip.  Same mi_model_t / mi_batch_t contract (pointers are to `dtype` elements), same metadata
// protocol (DECODE: positions from kv_seqlens on the device), same sample epilogue behind the LM head.
namespace {

struct GWorkspace {
  uint32_t* ctrl;
  char *xn, *qkv, *attn, *a, *b, *y, *res, *glog;
  int32_t *sel_idx, *active;
  float *sel_w, *wt, *attn_partial;
  size_t total;
};

GWorkspace carve_generic(const mi_model_t* m, int T, size_t es, char* base) {
  GWorkspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  const size_t qkv_cols = (size_t)(m->n_heads + 2 * m->n_kv_heads) * m->head_dim;
  const bool moe = m->num_experts > 0;
  w.ctrl = (uint32_t*)take(TICKET_BYTES);  // same control words as mi_forward (step counter, bad-id flag)
  w.xn = take((size_t)T * m->dim * es);
  w.qkv = take((size_t)T * qkv_cols * es);
  w.attn = take((size_t)T * m->n_heads * m->head_dim * es);
  w.a = take((size_t)T * m->hidden_dim * es);
  w.b = take((size_t)T * m->hidden_dim * es);
  w.y = take(moe ? (size_t)T * m->dim * es : 0);
  w.res = take(moe ? (size_t)T * m->dim * es : 0);
  w.glog = take(moe ? (size_t)T * m->num_experts * es : 0);
  w.sel_idx = (int32_t*)take(moe ? (size_t)T * m->top_k * 4 : 0);
  w.sel_w = (float*)take(moe ? (size_t)T * m->top_k * 4 : 0);
  w.active = (int32_t*)take(moe ? (size_t)T * 4 : 0);
  w.wt = (float*)take(moe ? (size_t)T * 4 : 0);
  w.attn_partial = (float*)take(g_attn_partial_floats(T, m->n_heads, m->head_dim) * sizeof(float));
  w.total = off;
  return w;
}

int check_model_generic(const mi_model_t* m, int dtype) {
  if (!m || !m->layers) return fail(MI_ERR_ARG, "null model");
  if (dtype != G_DT_BF16 && dtype != G_DT_FP16 && dtype != G_DT_FP32) return fail(MI_ERR_ARG, "storage dtype %d", dtype);
  if (m->head_dim <= 0 || m->head_dim > 256 || m->head_dim % 8) return fail(MI_ERR_SHAPE, "head_dim %d: multiple of 8, <= 256", m->head_dim);
  if (m->n_kv_heads <= 0 || m->n_heads % m->n_kv_heads) return fail(MI_ERR_SHAPE, "n_heads %% n_kv_heads != 0");
  if (m->dim % 8 || m->hidden_dim % 8) return fail(MI_ERR_SHAPE, "dim/hidden_dim must be multiples of 8");
  if (m->num_experts > 0 && (m->top_k <= 0 || m->top_k > m->num_experts)) return fail(MI_ERR_SHAPE, "MoE: 0 < top_k <= num_experts");
  return MI_OK;
}

}  // namespace

extern "C" {

// ---- leaf operators in any storage dtype (ABI v6): what module-level callers of an fp16 / fp32 model bind - the Pixtral tower
// (vision_encoder.py), RMSNorm / FeedForward modules used stand-alone.  Same kernels as mi_forward_generic.
static int check_dt(int dtype, const char* what) {
  if (dtype != G_DT_BF16 && dtype != G_DT_FP16 && dtype != G_DT_FP32) return fail(MI_ERR_ARG, "%s: storage dtype %d", what, dtype);
  return MI_OK;
}

int mi_embedding_generic(void* out, const void* table, const int64_t* ids, int T, int D, int vocab, int dtype, mi_stream_t stream) {
  MI_TRY(check_dt(dtype, "mi_embedding_generic"));
  if (!out || !table || !ids || T <= 0 || D <= 0) return fail(MI_ERR_ARG, "mi_embedding_generic");
  return hip_rc(launch_g_embedding(dtype, out, table, ids, T, D, vocab, nullptr, (hipStream_t)stream), "embedding");
}

int mi_rmsnorm_generic(void* out, const void* x, const void* w, int T, int D, float eps, int dtype, mi_stream_t stream) {
  MI_TRY(check_dt(dtype, "mi_rmsnorm_generic"));
  if (!out || !x || !w || T <= 0 || D <= 0) return fail(MI_ERR_ARG, "mi_rmsnorm_generic");
  return hip_rc(launch_g_rmsnorm(dtype, out, x, w, T, D, eps, (hipStream_t)stream), "rmsnorm");
}

/* out[:, columns of w[i]] = epilogue(x @ w[i]^T), i < 3 (w[i] == NULL ends the list): MI_EPI_STORE, MI_EPI_RESIDUAL (residual
 * [M, N] with row stride ldo) or MI_EPI_LOGITS (fp32 out).  One launch per matrix. */
int mi_linear_generic(void* out, int ldo, const void* x, int ldx, int M, int K, const void* const w[3], const int n_rows[3],
                      int epilogue, const void* residual, int dtype, mi_stream_t stream) {
  MI_TRY(check_dt(dtype, "mi_linear_generic"));
  if (!out || !x || !w || !n_rows || !w[0] || M <= 0 || K <= 0) return fail(MI_ERR_ARG, "mi_linear_generic");
  int epi;
  switch (epilogue) {
    case MI_EPI_STORE: epi = G_EPI_STORE; break;
    case MI_EPI_RESIDUAL: epi = G_EPI_RESIDUAL; break;
    case MI_EPI_LOGITS: epi = G_EPI_LOGITS; break;
    default: return fail(MI_ERR_UNSUPPORTED, "mi_linear_generic: epilogue %d (SwiGLU: two calls + mi_swiglu_generic)", epilogue);
  }
  if (epi == G_EPI_RESIDUAL && !residual) return fail(MI_ERR_ARG, "mi_linear_generic: residual");
  const size_t es = g_elem_bytes(dtype), oes = epi == G_EPI_LOGITS ? 4 : es;
  int col = 0;
  for (int i = 0; i < 3 && w[i]; ++i) {
    if (n_rows[i] <= 0) return fail(MI_ERR_ARG, "mi_linear_generic: n_rows[%d]", i);
    GLinearArgs g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.ldx = ldx; g.w = w[i]; g.out = (char*)out + (size_t)col * oes; g.ldo = ldo;
    g.residual = residual ? (const char*)residual + (size_t)col * es : nullptr; g.ldr = ldo;
    g.M = M; g.N = n_rows[i]; g.K = K; g.epi = epi;
    MI_TRY(hip_rc(launch_g_linear(dtype, g, (hipStream_t)stream), "linear"));
    col += n_rows[i];
  }
  return MI_OK;
}

/* rope.py:13-23 in place on the first n_rot_cols columns (heads of head_dim): pair i of a head turns by rope_cs[tok_pos[t], i] */
int mi_rope_inplace_generic(void* qkv, int ld, int T, int n_rot_cols, int head_dim, const float* rope_cs, const int32_t* tok_pos,
                            int dtype, mi_stream_t stream) {
  MI_TRY(check_dt(dtype, "mi_rope_inplace_generic"));
  if (!qkv || !rope_cs || !tok_pos || T <= 0 || head_dim <= 0 || head_dim % 2 || n_rot_cols % head_dim || n_rot_cols > ld)
    return fail(MI_ERR_ARG, "mi_rope_inplace_generic");
  return hip_rc(launch_g_rope(dtype, qkv, ld, T, n_rot_cols, head_dim, rope_cs, tok_pos, (hipStream_t)stream), "rope");
}

/* the cache=None attention (transformer_layers.py:72-73,165: every token sees every token): qkv [T, ld] = q | k | v after RoPE */
int mi_attention_nocache_generic(void* out, const void* qkv, int ld, int T, int n_heads, int n_kv_heads, int head_dim,
                                 float softmax_scale, int dtype, mi_stream_t stream) {
  MI_TRY(check_dt(dtype, "mi_attention_nocache_generic"));
  if (!out || !qkv || T <= 0 || n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads || head_dim <= 0 || head_dim > 256)
    return fail(MI_ERR_ARG, "mi_attention_nocache_generic");
  GAttnArgs a;
  memset(&a, 0, sizeof(a));
  a.out = out; a.ldo = n_heads * head_dim; a.qkv = qkv; a.ld = ld; a.W = T; a.T = T; a.H = n_heads; a.Hkv = n_kv_heads; a.Dh = head_dim;
  a.causal = 0;
  a.scale = softmax_scale > 0.f ? softmax_scale : 1.0f / sqrtf((float)head_dim);
  return hip_rc(launch_g_attention(dtype, a, (hipStream_t)stream), "attention");
}

/* a <- silu(a) * b on [T, F] dense rows (transformer_layers.py:106) */
int mi_swiglu_generic(void* a, const void* b, int T, int F, int dtype, mi_stream_t stream) {
  MI_TRY(check_dt(dtype, "mi_swiglu_generic"));
  if (!a || !b || T <= 0 || F <= 0) return fail(MI_ERR_ARG, "mi_swiglu_generic");
  return hip_rc(launch_g_swiglu(dtype, a, b, T, F, nullptr, (hipStream_t)stream), "swiglu");
}

int mi_gelu_generic(void* x, int ldx, int T, int N, int dtype, mi_stream_t stream) {
  MI_TRY(check_dt(dtype, "mi_gelu_generic"));
  if (!x || T <= 0 || N <= 0 || ldx < N) return fail(MI_ERR_ARG, "mi_gelu_generic");
  return hip_rc(launch_g_gelu(dtype, x, ldx, T, N, (hipStream_t)stream), "gelu");
}

size_t mi_workspace_bytes_generic(const mi_model_t* model, int T, int dtype) {
  if (!model || T <= 0) return 0;
  return carve_generic(model, T, g_elem_bytes(dtype), nullptr).total;
}

int mi_forward_generic(const mi_model_t* m, const mi_batch_t* bt, int dtype, mi_stream_t stream) {
  MI_TRY(check_model_generic(m, dtype));
  if (!bt || bt->T <= 0 || bt->B <= 0 || !bt->h || !bt->workspace) return fail(MI_ERR_ARG, "mi_forward_generic: batch");
  if (!bt->q_start || !bt->kv_before || !bt->tok_seq || !bt->tok_pos) return fail(MI_ERR_ARG, "mi_forward_generic: metadata");
  const int T = bt->T, B = bt->B, branch = bt->branch;
  const bool has_cache = branch != MI_BRANCH_NOCACHE;
  if (has_cache && (!bt->cache_k || !bt->cache_v || !bt->cache_sizes)) return fail(MI_ERR_ARG, "mi_forward_generic: cache");
  if (!kv_layout_ok(bt->kv_layout)) return fail(MI_ERR_ARG, "mi_forward_generic: kv_layout");
  const int kvl = bt->kv_layout;
  if (branch == MI_BRANCH_DECODE && (T != B || !bt->kv_seqlens)) return fail(MI_ERR_ARG, "mi_forward_generic: decode needs T == B");
  if (bt->logits && (!m->final_norm || !m->output)) return fail(MI_ERR_ARG, "mi_forward_generic: logits on a rank without LM head");
  hipStream_t s = (hipStream_t)stream;
  const int dt = dtype;
  const size_t es = g_elem_bytes(dt);
  GWorkspace ws = carve_generic(m, T, es, (char*)bt->workspace);
  if (ws.total > bt->workspace_bytes)
    return fail(MI_ERR_WORKSPACE, "workspace %zu < required %zu", bt->workspace_bytes, ws.total);

  const int D = m->dim, H = m->n_heads, Hkv = m->n_kv_heads, Dh = m->head_dim, F = m->hidden_dim;
  const int nq = H * Dh, nkv = Hkv * Dh, qkv_cols = nq + 2 * nkv;
  char* h = (char*)bt->h;
  const bool want_sample = branch == MI_BRANCH_DECODE && bt->logits && bt->greedy_token && bt->greedy_logprob;
  if (bt->greedy_token && !want_sample)
    return fail(MI_ERR_ARG, "mi_forward_generic: greedy_token needs the DECODE branch, logits and greedy_logprob");
  if (want_sample && bt->hist_len > 0 && (!bt->hist_token || !bt->hist_logprob))
    return fail(MI_ERR_ARG, "mi_forward_generic: hist_len > 0 without history buffers");
  const bool want_topp = want_sample && bt->sample_temperature > 0.f;
  if (bt->sample_temperature < 0.f || (want_topp && !(bt->sample_top_p >= 0.f && bt->sample_top_p <= 1.f)))
    return fail(MI_ERR_ARG, "mi_forward_generic: sample_temperature / sample_top_p");

  auto linear = [&](const void* x, int ldx, const void* w, void* out, int ldo, int N, int K, int epi, const void* residual,
                    const int32_t* active, const char* what) -> int {
    GLinearArgs g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.ldx = ldx; g.w = w; g.out = out; g.ldo = ldo; g.residual = residual; g.ldr = ldo;
    g.M = T; g.N = N; g.K = K; g.epi = epi; g.active = active;
    return hip_rc(launch_g_linear(dt, g, s), what);
  };
  // T <= 8 (decode steps, tiny prompts): the row kernel's fused forms - RMSNorm in the prologue, q | k | v in one launch,
  // gate / up / SiLU / product in one launch: 8 launches per dense layer instead of 13
  const bool rows = g_gemv_takes(T, D, D);
  // ... and for fp16 those launches are the TUNED weight-streaming kernels themselves (gemv.hip compiled for fp16 payloads,
  // launch_gemv_f16): RMSNorm + q | k | v + RoPE in one launch, Wo / W2 with the residual, RMSNorm + gate | up + SiLU, the LM
  // head - the launch path of mi_forward minus its bf16 attention kernels.  MI_GENERIC_GEMV_TUNED=0: the generic row kernel.
  static int tuned_rows = -1;
  if (tuned_rows < 0) {
    const char* e = getenv("MI_GENERIC_GEMV_TUNED");
    tuned_rows = e ? atoi(e) : 1;
  }
  const bool rows16 = rows && dt == G_DT_FP16 && tuned_rows != 0 && Dh % 2 == 0;
  auto gemv16 = [&](GemvArgs a, const char* what) -> int {  // gemv_passes for the fp16 compile
    const int cap = gemv_max_tokens_f16(a.K);
    const size_t out_elt = (a.mode == GEMV_LOGITS) ? 4 : 2;
    for (int t0 = 0; t0 < T; t0 += cap) {
      GemvArgs p = a;
      p.T = (T - t0 < cap) ? T - t0 : cap;
      p.x = a.x + (size_t)t0 * a.ldx;
      p.out = (char*)a.out + (size_t)t0 * a.ldo * out_elt;
      if (a.residual) p.residual = a.residual + (size_t)t0 * a.ldo;
      if (a.tok_pos) p.tok_pos = a.tok_pos + t0;
      if (a.tok_seq) p.tok_seq = a.tok_seq + t0;
      MI_TRY(hip_rc(launch_gemv_f16(p, s), what));
    }
    return MI_OK;
  };

  if (branch == MI_BRANCH_DECODE)
    MI_TRY(hip_rc(launch_decode_prep(bt->kv_seqlens, bt->q_start, bt->kv_before, bt->tok_seq, bt->tok_pos, B, ws.ctrl, s), "decode_prep"));
  if (m->tok_embeddings && bt->input_ids)
    MI_TRY(hip_rc(launch_g_embedding(dt, h, m->tok_embeddings, bt->input_ids, T, D, m->vocab_size, ws.ctrl + 3, s), "embedding"));

  for (int l = 0; l < m->n_layers; ++l) {
    const mi_layer_t& L = m->layers[l];
    const int W = has_cache ? bt->cache_sizes[l] : T;
    void* ck = has_cache ? bt->cache_k[l] : nullptr;
    void* cv = has_cache ? bt->cache_v[l] : nullptr;
    // ---- attention_norm, q | k | v, RoPE (transformer_layers.py:66-70)
    if (rows16) {
      GemvArgs a;
      memset(&a, 0, sizeof(a));
      a.mode = GEMV_QKV_ROPE; a.K = D; a.N = qkv_cols; a.x = (const bf16_t*)h; a.ldx = D;
      a.norm_w = (const bf16_t*)L.attention_norm; a.eps = m->norm_eps;
      a.w0 = (const bf16_t*)L.wq; a.w1 = (const bf16_t*)L.wk; a.w2 = (const bf16_t*)L.wv; a.n0 = nq; a.n1 = nq + nkv;
      a.out = ws.qkv; a.ldo = qkv_cols;
      a.rope_cs = m->rope_cs; a.tok_pos = bt->tok_pos; a.tok_seq = bt->tok_seq; a.head_dim = Dh;
      a.write_kv = 0;  // (the ring write follows the attention here: g_kv_write below)
      MI_TRY(gemv16(a, "norm + q|k|v + rope (fp16 gemv)"));
    } else if (rows) {
      GLinearArgs g;
      memset(&g, 0, sizeof(g));
      g.x = h; g.ldx = D; g.w = L.wq; g.w1 = L.wk; g.w2 = L.wv; g.n0 = nq; g.n1 = nq + nkv; g.norm_w = L.attention_norm; g.eps = m->norm_eps;
      g.out = ws.qkv; g.ldo = qkv_cols; g.M = T; g.N = qkv_cols; g.K = D; g.epi = G_EPI_STORE;
      MI_TRY(hip_rc(launch_g_linear(dt, g, s), "norm + q|k|v"));
    } else {
      MI_TRY(hip_rc(launch_g_rmsnorm(dt, ws.xn, h, L.attention_norm, T, D, m->norm_eps, s), "attention_norm"));
      GLinearArgs g;
      memset(&g, 0, sizeof(g));
      g.x = ws.xn; g.ldx = D; g.w = L.wq; g.w1 = L.wk; g.w2 = L.wv; g.n0 = nq; g.n1 = nq + nkv;
      g.out = ws.qkv; g.ldo = qkv_cols; g.M = T; g.N = qkv_cols; g.K = D; g.epi = G_EPI_STORE;
      if (g_linear_fused_ok(dt, g)) {  // (fp16, at least 256 rows: one launch of the 256-tile kernel)
        MI_TRY(hip_rc(launch_g_linear(dt, g, s), "q|k|v"));
      } else {
        MI_TRY(linear(ws.xn, D, L.wq, ws.qkv, qkv_cols, nq, D, G_EPI_STORE, nullptr, nullptr, "wq"));
        MI_TRY(linear(ws.xn, D, L.wk, ws.qkv + (size_t)nq * es, qkv_cols, nkv, D, G_EPI_STORE, nullptr, nullptr, "wk"));
        MI_TRY(linear(ws.xn, D, L.wv, ws.qkv + (size_t)(nq + nkv) * es, qkv_cols, nkv, D, G_EPI_STORE, nullptr, nullptr, "wv"));
      }
    }
    if (!rows16) MI_TRY(hip_rc(launch_g_rope(dt, ws.qkv, qkv_cols, T, nq + nkv, Dh, m->rope_cs, bt->tok_pos, s), "rope"));
    // ---- attention over [surviving ring entries ++ this forward's keys], then the ring write (cache.py:83-117)
    GAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.out = ws.attn; a.ldo = nq; a.qkv = ws.qkv; a.ld = qkv_cols; a.cache_k = ck; a.cache_v = cv; a.kv_layout = kvl;
    a.W = W; a.T = T; a.H = H; a.Hkv = Hkv; a.Dh = Dh;
    a.q_start = bt->q_start; a.kv_before = bt->kv_before; a.tok_seq = bt->tok_seq; a.tok_pos = bt->tok_pos;
    a.causal = has_cache ? 1 : 0;
    a.scale = 1.0f / sqrtf((float)Dh);
    a.partial = ws.attn_partial;
    a.B = has_cache ? B : 1; a.max_q_len = has_cache ? bt->max_q_len : T;
    MI_TRY(hip_rc(launch_g_attention(dt, a, s), "attention"));
    if (has_cache)
      MI_TRY(hip_rc(launch_g_kv_write(dt, ck, cv, W, ws.qkv + (size_t)nq * es, ws.qkv + (size_t)(nq + nkv) * es, qkv_cols, T, nkv,
                                      bt->tok_seq, bt->tok_pos, bt->q_start, kvl, Dh, s), "kv_write"));
    // ---- h = h + wo(attn)
    if (rows16) {
      GemvArgs a;
      memset(&a, 0, sizeof(a));
      a.mode = GEMV_RESIDUAL; a.K = nq; a.N = D; a.x = (const bf16_t*)ws.attn; a.ldx = nq;
      a.w0 = (const bf16_t*)L.wo; a.n0 = a.n1 = D; a.out = h; a.ldo = D; a.residual = (const bf16_t*)h;
      MI_TRY(gemv16(a, "wo (fp16 gemv)"));
    } else {
      MI_TRY(linear(ws.attn, nq, L.wo, h, D, D, nq, G_EPI_RESIDUAL, h, nullptr, "wo"));
    }
    // ---- h = h + FFN(ffn_norm(h))
    if (m->num_experts == 0) {
      if (!L.w1 || !L.w2 || !L.w3) return fail(MI_ERR_ARG, "mi_forward_generic: dense layer without w1/w2/w3");
      if (rows16) {
        GemvArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = GEMV_SWIGLU; a.K = D; a.N = F; a.x = (const bf16_t*)h; a.ldx = D;
        a.norm_w = (const bf16_t*)L.ffn_norm; a.eps = m->norm_eps;
        a.w0 = (const bf16_t*)L.w1; a.w1 = (const bf16_t*)L.w3; a.n0 = a.n1 = F; a.out = ws.a; a.ldo = F;
        MI_TRY(gemv16(a, "norm + w1|w3 + swiglu (fp16 gemv)"));
        memset(&a, 0, sizeof(a));
        a.mode = GEMV_RESIDUAL; a.K = F; a.N = D; a.x = (const bf16_t*)ws.a; a.ldx = F;
        a.w0 = (const bf16_t*)L.w2; a.n0 = a.n1 = D; a.out = h; a.ldo = D; a.residual = (const bf16_t*)h;
        MI_TRY(gemv16(a, "w2 (fp16 gemv)"));
      } else if (rows) {
        GLinearArgs g;
        memset(&g, 0, sizeof(g));
        g.x = h; g.ldx = D; g.w = L.w1; g.w1 = L.w3; g.norm_w = L.ffn_norm; g.eps = m->norm_eps;
        g.out = ws.a; g.ldo = F; g.M = T; g.N = F; g.K = D; g.epi = G_EPI_SWIGLU;
        MI_TRY(hip_rc(launch_g_linear(dt, g, s), "norm + w1|w3 + swiglu"));
      } else {
        MI_TRY(hip_rc(launch_g_rmsnorm(dt, ws.xn, h, L.ffn_norm, T, D, m->norm_eps, s), "ffn_norm"));
        GLinearArgs g;
        memset(&g, 0, sizeof(g));
        g.x = ws.xn; g.ldx = D; g.w = L.w1; g.w1 = L.w3; g.out = ws.a; g.ldo = F; g.M = T; g.N = F; g.K = D; g.epi = G_EPI_SWIGLU;
        if (g_linear_fused_ok(dt, g)) {
          MI_TRY(hip_rc(launch_g_linear(dt, g, s), "w1|w3 + swiglu"));
        } else {
          MI_TRY(linear(ws.xn, D, L.w1, ws.a, F, F, D, G_EPI_STORE, nullptr, nullptr, "w1"));
          MI_TRY(linear(ws.xn, D, L.w3, ws.b, F, F, D, G_EPI_STORE, nullptr, nullptr, "w3"));
          MI_TRY(hip_rc(launch_g_swiglu(dt, ws.a, ws.b, T, F, nullptr, s), "swiglu"));
        }
      }
      if (!rows16) MI_TRY(linear(ws.a, F, L.w2, h, D, D, F, G_EPI_RESIDUAL, h, nullptr, "w2"));
    } else {
      MI_TRY(hip_rc(launch_g_rmsnorm(dt, ws.xn, h, L.ffn_norm, T, D, m->norm_eps, s), "ffn_norm"));
      // moe.py:24-32: experts in ascending id, each adding round(weight * expert(x)) for the rows that picked it
      const int E = m->num_experts, k = m->top_k;
      if (!L.gate || !L.expert_w_host) return fail(MI_ERR_ARG, "mi_forward_generic: MoE layer tables");
      MI_TRY(linear(ws.xn, D, L.gate, ws.glog, E, E, D, G_EPI_STORE, nullptr, nullptr, "gate"));
      MI_TRY(hip_rc(launch_g_moe_topk(dt, ws.glog, T, E, k, ws.sel_idx, ws.sel_w, s), "moe top-k"));
      MI_TRY(hip_rc(launch_g_zero(dt, ws.res, (size_t)T * D, s), "moe zero"));
      for (int e = 0; e < E; ++e) {
        const void* w1 = L.expert_w_host[e * 3 + 0];
        const void* w2 = L.expert_w_host[e * 3 + 1];
        const void* w3 = L.expert_w_host[e * 3 + 2];
        MI_TRY(hip_rc(launch_g_moe_mask(ws.sel_idx, ws.sel_w, T, k, e, ws.active, ws.wt, s), "moe mask"));
        MI_TRY(linear(ws.xn, D, w1, ws.a, F, F, D, G_EPI_STORE, nullptr, ws.active, "expert w1"));
        MI_TRY(linear(ws.xn, D, w3, ws.b, F, F, D, G_EPI_STORE, nullptr, ws.active, "expert w3"));
        MI_TRY(hip_rc(launch_g_swiglu(dt, ws.a, ws.b, T, F, ws.active, s), "expert swiglu"));
        MI_TRY(linear(ws.a, F, w2, ws.y, D, D, F, G_EPI_STORE, nullptr, ws.active, "expert w2"));
        MI_TRY(hip_rc(launch_g_moe_accum(dt, ws.res, ws.y, ws.active, ws.wt, T, D, s), "moe accumulate"));
      }
      MI_TRY(hip_rc(launch_g_add(dt, h, h, ws.res, (size_t)T * D, s), "moe residual"));
    }
  }

  if (m->final_norm) {
    if (bt->logits) {
      if (rows16) {
        GemvArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = GEMV_LOGITS; a.K = D; a.N = m->vocab_size; a.x = (const bf16_t*)h; a.ldx = D;
        a.norm_w = (const bf16_t*)m->final_norm; a.eps = m->norm_eps;
        a.w0 = (const bf16_t*)m->output; a.n0 = a.n1 = m->vocab_size; a.out = bt->logits; a.ldo = m->vocab_size;
        MI_TRY(gemv16(a, "final norm + lm head (fp16 gemv)"));
      } else if (rows) {
        GLinearArgs g;
        memset(&g, 0, sizeof(g));
        g.x = h; g.ldx = D; g.w = m->output; g.norm_w = m->final_norm; g.eps = m->norm_eps;
        g.out = bt->logits; g.ldo = m->vocab_size; g.M = T; g.N = m->vocab_size; g.K = D; g.epi = G_EPI_LOGITS;
        MI_TRY(hip_rc(launch_g_linear(dt, g, s), "final norm + lm head"));
      } else {
        MI_TRY(hip_rc(launch_g_rmsnorm(dt, ws.xn, h, m->final_norm, T, D, m->norm_eps, s), "final norm"));
        MI_TRY(linear(ws.xn, D, m->output, bt->logits, m->vocab_size, m->vocab_size, D, G_EPI_LOGITS, nullptr, nullptr, "lm head"));
      }
      if (want_sample) {
        if (want_topp)
          MI_TRY(hip_rc(launch_sample_top_p(bt->logits, m->vocab_size, B, m->vocab_size, bt->sample_temperature, bt->sample_top_p,
                                            bt->sample_seed, bt->sample_offset, nullptr, bt->greedy_token, bt->greedy_logprob,
                                            bt->hist_token, bt->hist_logprob, bt->hist_len, ws.ctrl, s), "top-p sample"));
        else
          MI_TRY(hip_rc(launch_greedy_rows(bt->logits, m->vocab_size, B, m->vocab_size, bt->greedy_token, bt->greedy_logprob,
                                           bt->hist_token, bt->hist_logprob, bt->hist_len, ws.ctrl, s), "greedy sample"));
      }
    } else {
      MI_TRY(hip_rc(launch_g_rmsnorm(dt, h, h, m->final_norm, T, D, m->norm_eps, s), "final norm"));
    }
  }
  return MI_OK;
}

}  // extern "C"
