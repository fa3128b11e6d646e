/*
 * mistral_hip.h -- C ABI of libmistral_hip.so: the MI355X (gfx950) implementation of
 * mistral-inference's `Transformer.forward_partial` hot path.
 *
 * The reference (mistralai/mistral-inference v1.6.0) has NO FFI/plugin layer of its own: its hot
 * path calls torch/xformers kernels from Python.  This header is therefore the boundary a
 * maintainer would bind (ctypes stub in INTEGRATION.md); every entry point names the reference
 * lines whose work it replaces.  Paths are relative to src/mistral_inference/ of the reference.
 *
 * Conventions
 *   - plain C types only; all tensor arguments are raw DEVICE pointers unless marked "host";
 *   - storage dtype is bf16 (uint16 payload), row-major, innermost dimension contiguous;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates,
 *     frees or synchronises, and keeps no state between calls (scratch comes from the caller);
 *   - return value: 0 = ok, >0 = hipError_t of a failed launch, <0 = MI_ERR_* argument check;
 *     `mi_error_string` turns any of them into text.  The Python host raises RuntimeError.
 *   - numerics contract (rounding points): SURVEY.md Appendix A / DESIGN.md section 4.
 */
#ifndef MISTRAL_HIP_H
#define MISTRAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 7

#define MI_OK 0
#define MI_ERR_ARG (-1)        /* null pointer / non-positive size                        */
#define MI_ERR_SHAPE (-2)      /* shape not supported by the gfx950 kernels (see message) */
#define MI_ERR_WORKSPACE (-3)  /* caller workspace too small                              */
#define MI_ERR_UNSUPPORTED (-4)
#define MI_ERR_RCCL (-5)       /* a RCCL call failed: text in mi_rccl_last_error()        */

typedef void* mi_stream_t; /* hipStream_t */

int mi_abi_version(void);
const char* mi_error_string(int code);
/* text of the last MI_ERR_* raised on this thread (empty string if none) */
const char* mi_last_error_detail(void);

/* ------------------------------------------------------------------------------------------------
 * Leaf operators (each replaces one group of torch/xformers launches of the reference)
 * ---------------------------------------------------------------------------------------------- */

/* transformer.py:193  h = tok_embeddings(input_ids).  out[T,D] = table[ids[t], :].  ids must lie in [0, vocab): the
 * reference raises IndexError otherwise, which a kernel cannot - this leaf reads row 0 / vocab-1 for such an id (callers
 * validate; mi_forward additionally records the offending token index in the workspace, see mi_decode_engine_status). */
int mi_embedding(void* out, const void* table, const int64_t* ids, int T, int D, int vocab, mi_stream_t stream);

/* transformer_layers.py:115-120 RMSNorm.forward: out = bf16(bf16(x_f32 * rsqrt(mean(x^2)+eps)) * w).
 * In-place (out == x) is allowed. */
int mi_rmsnorm(void* out, const void* x, const void* w, int T, int D, float eps, mi_stream_t stream);

/* rope.py:13-23 apply_rotary_emb, in place on the q and k column blocks of a fused activation
 * buffer qkv[T, ld] = [ q (n_heads*Dh) | k (n_kv_heads*Dh) | v ... ].
 * rope_cs: fp32 [rope_len, Dh/2, 2] = (cos, sin), i.e. view_as_real of the reference's complex64
 * freqs_cis table (rope.py:6-10, transformer.py:113-116); tok_pos[T] absolute positions. */
int mi_rope_inplace(void* qkv, int ld, int T, int n_heads, int n_kv_heads, int head_dim, const float* rope_cs,
                    int rope_len, const int32_t* tok_pos, mi_stream_t stream);

/* cache.py:83-92 CacheView.update with to_cache_mask / cache_positions of cache.py:226-235:
 * token t (sequence b = tok_seq[t], index i = t - q_start[b] of s_b = q_start[b+1]-q_start[b] new
 * tokens) is stored iff i >= s_b - W, into ring slot tok_pos[t] % W of row b.
 * k/v: [T, ld] activation views (already at the k / v column), cache_k/v: one layer's rings in `kv_layout` (below). */
int mi_kv_write(void* cache_k, void* cache_v, int W, const void* k, const void* v, int ld, int T, int kv_dim,
                const int32_t* tok_seq, const int32_t* tok_pos, const int32_t* q_start, int kv_layout, int head_dim,
                mi_stream_t stream);

/* ABI v7 - layout of a layer's K/V rings in HBM.  A kv head's head_dim elements are contiguous in both:
 *   MI_KV_SLOT_MAJOR  [max_batch, W, n_kv_heads, head_dim]: the reference's torch.empty shape (cache.py:163-167)
 *   MI_KV_HEAD_MAJOR  [max_batch, n_kv_heads, W, head_dim]: the slots of ONE kv head are contiguous, so the keys a decode work
 *                     item (kv head, range of slots) reads are a single run - 4-KiB runs that the persistent engine streams at
 *                     its weight rate; the strided form costs it 1.2 us per 16 KiB against 0.66.  What
 *                     mistral_inference.cache.BufferCache allocates (exposed to Python as a permuted view of the reference's shape).
 * Every entry point that touches a ring takes the layout; all rings of one mi_forward call share it (mi_batch_t.kv_layout). */
#define MI_KV_SLOT_MAJOR 0
#define MI_KV_HEAD_MAJOR 1

/* Epilogues of the dense contractions */
enum mi_epilogue {
  MI_EPI_STORE = 0,    /* out = bf16(acc)                                  nn.Linear                       */
  MI_EPI_RESIDUAL = 1, /* out = bf16(residual + bf16(acc))                 transformer_layers.py:166,168   */
  MI_EPI_SWIGLU = 2,   /* out = bf16(bf16(silu(bf16(acc1))) * bf16(acc3))  transformer_layers.py:105-106   */
  MI_EPI_LOGITS = 3    /* out_f32 = float(bf16(acc))                       transformer.py:235,239-242      */
};

/* out[M,N] = epilogue(x[M,K] @ W^T).  W is given as up to three row blocks (w[i] has n_rows[i]
 * rows of K bf16; used for the fused Wq|Wk|Wv projection, transformer_layers.py:66) except for
 * MI_EPI_SWIGLU where w[0]=W1, w[1]=W3 (both [N,K]).  M <= 8 runs the weight-streaming GEMV kernels
 * (HBM-bound), M > 8 the MFMA GEMM.  norm_w != NULL fuses RMSNorm(x; norm_w, eps) in front
 * (GEMV path only; the GEMM path requires norm_w == NULL).
 * residual: [M, ldo] (may alias out).  out is bf16 [M, ldo] (fp32 for MI_EPI_LOGITS). */
int mi_linear(void* out, int ldo, const void* x, int ldx, int M, int K, const void* const w[3], const int n_rows[3],
              int epilogue, const void* residual, const void* norm_w, float eps, mi_stream_t stream);

/* generate.py:101-118 needs, of the prompt's [T, vocab] logits (transformer.py:235-242), only log_softmax(logits)[t, next
 * token]: logprob[m] = l[m, target[m]] - logsumexp(l[m, :]) with l = float(bf16(x @ W^T)), computed in ONE pass over the
 * LM head without materialising the logits (per-tile max / sum-exp partials in `scratch`).  M < 256 rows take the plain
 * logits + row-reduction route inside the same call.  target[m] outside [0, vocab) yields -logsumexp. */
size_t mi_lm_head_logprobs_scratch_bytes(int M, int vocab);
int mi_lm_head_logprobs(float* logprob, const void* x, int ldx, int M, int K, const void* w, int vocab,
                        const int32_t* target, void* scratch, size_t scratch_bytes, mi_stream_t stream);

/* generate.py:124 + :134-136 at temperature 0 for B rows of fp32 logits [B, ld]: token[b] = FIRST index of the row maximum
 * (torch.argmax), logprob[b] = log_softmax(row)[token[b]] - one block per row.  mi_forward fuses the same reduction
 * behind its LM head (mi_batch_t.greedy_token). */
int mi_greedy_sample(const float* logits, int ld, int B, int vocab, int64_t* token, float* logprob, mi_stream_t stream);

/* generate.py:151-170 `sample` / `sample_top_p` at temperature > 0 for B rows of fp32 logits [B, ld], ONE launch (the
 * reference: softmax, a sort of the whole vocabulary, cumsum, masked fill, renormalise, torch.multinomial, gathers):
 *   probs = softmax(row / temperature); the tokens kept are the prefix of the descending order whose mass BEFORE the token
 *   is <= top_p (generate.py:165-167; ties in probability are ordered by ascending index - torch.sort leaves them
 *   unspecified); token[b] is drawn from the renormalised kept masses by inverse CDF in that order;
 *   logprob[b] = log_softmax(row)[token[b]] of the UNSCALED logits (generate.py:134-136).
 * The uniform variate is Philox4x32-10(seed; counter = offset, row b) unless `uniforms` (device fp32 [B], each in [0, 1))
 * is given - which is how tests pin a draw.  Masses are 40-bit fixed point: the result is bit-reproducible for a given
 * seed/offset (torch.multinomial's own stream cannot be reproduced; parity is distributional, tests/test_gpu_sampling.py).
 * mi_forward fuses the same kernel behind its LM head (mi_batch_t.sample_temperature > 0). */
int mi_sample_top_p(const float* logits, int ld, int B, int vocab, float temperature, float top_p, uint64_t seed,
                    uint64_t offset, const float* uniforms, int64_t* token, float* logprob, mi_stream_t stream);

/* Decode-branch attention (transformer_layers.py:77-89 with the mask of cache.py:249-254):
 * one query per sequence, keys = ring slots [0, min(pos+1, W)) of its row, GQA by kv = h / (H/Hkv)
 * (replaces repeat_kv, transformer_layers.py:16-19,84).  q: [B, ldq] (H*Dh used), out: [B, H*Dh].
 * tok_pos[b] = position of the new token (already written to the ring).  scratch: fp32, at least
 * mi_attn_decode_scratch_bytes(...) bytes; its first `B*Hkv` int32 words are arrival counters that
 * must be zero at the first call (the kernel leaves them zero). */
size_t mi_attn_decode_scratch_bytes(int B, int n_heads, int n_kv_heads, int head_dim, int W);
int mi_attn_decode(void* out, const void* q, int ldq, const void* cache_k, const void* cache_v, int W, int B,
                   int n_heads, int n_kv_heads, int head_dim, const int32_t* tok_pos, void* scratch, int kv_layout,
                   mi_stream_t stream);

/* Prefill-branch attention (transformer_layers.py:74-76,84-89; keys of cache.py:94-117 interleave_kv;
 * masks of cache.py:238-248).  For sequence b with s_b new tokens (rows q_start[b]..) and p_b =
 * kv_before[b] tokens already seen, key position kp is visible to query position qp iff
 * qp - W < kp <= qp; keys kp < p_b are read from the ring (slot kp % W, only the last min(p_b,W)
 * exist), keys kp >= p_b from the activation rows.  qkv: [T, ld] fused buffer (post-RoPE).
 * causal == 0 is the cache=None quirk (transformer_layers.py:165): one segment, no mask - also what one image of the
 * vision tower's block-diagonal mask is (vision_encoder.py:96-99).
 * softmax_scale <= 0 selects head_dim^-1/2 (the xformers default the reference relies on, transformer_layers.py:48,88);
 * the vision tower runs its 64-wide heads zero-padded to 128 and passes 64^-1/2.
 * out: [T, H*Dh]. */
int mi_attn_prefill(void* out, const void* qkv, int ld, const void* cache_k, const void* cache_v, int W, int B,
                    int max_q_len, int n_heads, int n_kv_heads, int head_dim, const int32_t* q_start,
                    const int32_t* kv_before, int causal, float softmax_scale, int kv_layout, mi_stream_t stream);

/* nn.GELU() of the vision-language adapter (vision_encoder.py:112-116; exact erf form): x <- bf16(gelu(x)) in place
 * over [T, N] bf16 rows with row pitch ldx. */
int mi_gelu(void* x, int ldx, int T, int N, mi_stream_t stream);

/* moe.py:25-27: logits = bf16(x @ Wg^T); top-k on them; fp32 softmax over the k picked, rounded to
 * bf16.  x is [T, D] (norm_w != NULL fuses the RMSNorm).  sel_idx int32 [T, k], sel_w fp32 [T, k]
 * (bf16-rounded values).  Ties: the lowest expert id wins (torch.topk leaves tie order unspecified). */
int mi_moe_router(int32_t* sel_idx, float* sel_w, const void* x, int ldx, int T, int D, const void* gate, int E,
                  int top_k, const void* norm_w, float eps, mi_stream_t stream);

/* transformer_layers.py:66-70,165 + rope.py:13-23 + cache.py:83-92 for T <= 8 tokens (the decode step) in ONE launch:
 * qkv[t] = [ rope(Wq xn) | rope(Wk xn) | Wv xn ] with xn = RMSNorm(x[t]; norm_w, eps) (norm_w == NULL: xn = x), every
 * projection rounded to bf16 before the rotation; when cache_k/cache_v are given the new K/V rows are also stored into
 * ring slot tok_pos[t] % W of row tok_seq[t] (tok_seq == NULL: row t) - CacheView.update at decode.
 * qkv: [T, ldo] bf16, ldo >= (n_heads + 2 n_kv_heads) * head_dim.  T > 8: MI_ERR_UNSUPPORTED (the prefill path is
 * mi_rmsnorm + mi_linear + mi_rope_inplace + mi_kv_write). */
int mi_qkv_rope_kvwrite(void* qkv, int ldo, const void* x, int ldx, int T, int D, const void* wq, const void* wk,
                        const void* wv, int n_heads, int n_kv_heads, int head_dim, const void* norm_w, float eps,
                        const float* rope_cs, int rope_len, const int32_t* tok_pos, const int32_t* tok_seq, void* cache_k,
                        void* cache_v, int W, int kv_layout, mi_stream_t stream);

/* moe.py:28-32 (+ the residual add of transformer_layers.py:168) for T <= 8 tokens, two launches, no host sync:
 * out[t] = bf16(residual[t] + R_t), R_t = sum over the token's picked experts in ascending expert id of
 * bf16(w * bf16(W2_e . bf16(silu(W1_e xn) * W3_e xn))), accumulated in bf16 from zero; xn = RMSNorm(x[t]) when norm_w is
 * given.  expert_w_dev: DEVICE array [E][3] of (w1, w2, w3) pointers; sel_idx/sel_w: mi_moe_router's output [T, top_k];
 * hidden_scratch: bf16 [T * top_k, F].  out may alias residual. */
int mi_moe_experts_decode(void* out, const void* residual, const void* x, int ldx, int T, int D, int F,
                          const void* const* expert_w_dev, const int32_t* sel_idx, const float* sel_w, int top_k,
                          const void* norm_w, float eps, void* hidden_scratch, mi_stream_t stream);

/* The same layer for any T (prefill): (token, slot) pairs sorted by expert ON THE DEVICE (the reference does one
 * torch.where host sync per expert per layer, moe.py:30), ONE token-grouped MFMA GEMM launch per projection over all
 * experts, then the ordered bf16 combine + residual.  x: dense [T, D] (already ffn-normalised), ldx == D. */
size_t mi_moe_grouped_gemm_scratch_bytes(int T, int D, int F, int E, int top_k);
int mi_moe_grouped_gemm(void* out, const void* residual, const void* x, int ldx, int T, int D, int F, int E, int top_k,
                        const void* const* expert_w_dev, const int32_t* sel_idx, const float* sel_w, void* scratch,
                        size_t scratch_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Whole local layer stack: Transformer.forward_partial (transformer.py:163-219) + the LM head of
 * Transformer.forward (transformer.py:229-242)
 * ---------------------------------------------------------------------------------------------- */
typedef struct mi_layer {
  const void* attention_norm; /* [D]            transformer_layers.py:143 */
  const void* wq;             /* [H*Dh, D]      transformer_layers.py:51  */
  const void* wk;             /* [Hkv*Dh, D]                         :52  */
  const void* wv;             /* [Hkv*Dh, D]                         :53  */
  const void* wo;             /* [D, H*Dh]                           :54  */
  const void* ffn_norm;       /* [D]                                 :144 */
  const void* w1;             /* [F, D]  dense FFN (NULL when MoE)   :101 */
  const void* w2;             /* [D, F]                              :102 */
  const void* w3;             /* [F, D]                              :103 */
  const void* gate;           /* [E, D]  MoE router (NULL when dense) moe.py:20 */
  const void* const* expert_w_dev;  /* DEVICE array [E][3] of (w1,w2,w3) pointers, moe.py:19 */
  const void* const* expert_w_host; /* the same table in host memory */
} mi_layer_t;

typedef struct mi_model {
  int32_t dim, n_heads, n_kv_heads, head_dim, hidden_dim, vocab_size;
  int32_t n_layers;               /* layers local to this pipeline rank (transformer.py:94-98) */
  int32_t num_experts, top_k;     /* 0,0 for dense */
  float norm_eps;
  const void* tok_embeddings;     /* [V, D] or NULL when pipeline_rank > 0 (transformer.py:56-57) */
  const void* final_norm;         /* [D]    or NULL when not the last rank (transformer.py:77-79) */
  const void* output;             /* [V, D] or NULL */
  const float* rope_cs;           /* fp32 [rope_len, Dh/2, 2] */
  int32_t rope_len;
  const mi_layer_t* layers;       /* host array [n_layers] */
} mi_model_t;

enum mi_branch {
  MI_BRANCH_NOCACHE = 0, /* cache=None: positions restart per sequence, attention unmasked         */
  MI_BRANCH_PREFILL = 1, /* first or subsequent prefill chunk (cache.py:236-248)                   */
  MI_BRANCH_DECODE = 2   /* all seqlens == 1 and tokens already cached (cache.py:249-254)          */
};

typedef struct mi_batch {
  int32_t T, B, branch;
  int32_t max_q_len;            /* host: max(seqlens) */
  const int64_t* input_ids;     /* dev [T]; embedded into h when model->tok_embeddings != NULL; NULL: h is the input */
  /* sequence metadata, device int32.  PREFILL/NOCACHE: written by the host (one H2D copy per
   * forward -- the reference rebuilds five tensors per LAYER, cache.py:226-263).
   * DECODE: filled on the device from kv_seqlens by the first kernel of the step. */
  int32_t* q_start;             /* [B+1] */
  int32_t* kv_before;           /* [B]   */
  int32_t* tok_seq;             /* [T]   */
  int32_t* tok_pos;             /* [T]   */
  int64_t* kv_seqlens;          /* dev [B] BufferCache.kv_seqlens (cache.py:170,193-195); DECODE reads
                                   positions from it and adds 1 on the device; NULL for NOCACHE */
  void* const* cache_k;         /* host array [n_layers] of dev rings: [max_batch, W_l, Hkv, Dh] (cache.py:163-167) or head-major
                                   [max_batch, Hkv, W_l, Dh], as kv_layout (last field) says */
  void* const* cache_v;
  const int32_t* cache_sizes;   /* host [n_layers] W_l (cache.py:13-24) */
  void* h;                      /* dev [T, D] bf16, in/out: the residual stream.  rank 0: overwritten by
                                   the embedding; rank>0: holds the received activations.  On return: the
                                   block-stack output, RMS-normalised when final_norm != NULL and
                                   logits == NULL (transformer.py:213-219) */
  float* logits;                /* dev [T, V] fp32 or NULL (transformer.py:235-242) */
  void* workspace;              /* dev scratch, >= mi_workspace_bytes(model, T, B, max W) */
  size_t workspace_bytes;
  /* ABI v4 - greedy sampling fused behind the LM head (generate.py:124 `torch.argmax(logits)` and :134-136
   * `log_softmax(logits)[token]` at temperature 0).  Optional (NULL: not computed); DECODE branch with `logits` only.
   *   greedy_token[b]   = first index of the maximum of logits[b, :]            (torch.argmax's tie rule)
   *   greedy_logprob[b] = log_softmax(logits[b, :])[greedy_token[b]]
   * `input_ids` MAY ALIAS `greedy_token`: the ids are consumed before the outputs are written, so a caller can chain
   * steps (the next step's input is this step's sample) without any copy between two mi_forward calls - the shape of
   * generate()'s greedy loop, and of a captured hipGraph that is replayed once per token.
   * hist_token / hist_logprob: optional rings [hist_len, B]; the step's sample is also stored at row
   * (steps % hist_len), `steps` being the number of decode steps this workspace has run before this one
   * (mi_decode_engine_status word 5; the caller may zero that word) - so a generation loop reads its tokens back in
   * ONE copy at the end instead of one per step (generate.py:137 `generated_tensors.append`). */
  int64_t* greedy_token;        /* dev [B] or NULL */
  float* greedy_logprob;        /* dev [B] */
  int64_t* hist_token;          /* dev [hist_len, B] or NULL */
  float* hist_logprob;          /* dev [hist_len, B] or NULL */
  int32_t hist_len;
  /* ABI v5 - nucleus sampling instead of the argmax (generate.py:126 `sample(logits, temperature, top_p=0.8)`): with
   * sample_temperature > 0 the token written to greedy_token / the history ring is DRAWN as mi_sample_top_p describes
   * (one more small kernel behind the LM head, inside the same captured step), greedy_logprob is its log-probability under
   * the unscaled logits.  The variate of a step, row b is Philox(sample_seed; sample_offset + n, b) with n = the workspace's
   * decode-step counter (mi_decode_engine_status word 5) - a device value, so hipGraph replays draw fresh numbers; a caller
   * that wants the stream of a generation to start at the same point every time passes sample_offset = -(counter at its
   * start).  sample_temperature == 0: the ABI v4 behaviour (argmax). */
  float sample_temperature;
  float sample_top_p;
  uint64_t sample_seed;
  uint64_t sample_offset;
  /* ABI v7 */
  int32_t kv_layout;            /* MI_KV_SLOT_MAJOR (0, the reference's shape) or MI_KV_HEAD_MAJOR: layout of EVERY ring in cache_k / cache_v */
} mi_batch_t;

size_t mi_workspace_bytes(const mi_model_t* model, int T, int B, int max_cache_size);
/* The workspace must be ZERO-FILLED once after allocation (the first 4 KiB hold the control words of the persistent
 * decode engine - step epoch, status - and the engine's hand-off granules carry tags that must never match garbage). */
int mi_forward(const mi_model_t* model, const mi_batch_t* batch, mi_stream_t stream);

/* ABI v6 - storage dtypes other than bf16 (reference transformer.py:303,338: `from_folder(dtype=...)` keeps the dtype the
 * caller asks for; the reference's own tests build fp32 models, tests/test_generate.py:51,100) and bf16 models of a shape
 * mi_forward declines with MI_ERR_SHAPE (head_dim != 128, more than 16 experts, top_k = 3).  Same structs, same metadata
 * protocol, same sample epilogue; every weight / activation / cache pointer is to `dtype` elements, `logits` stays fp32.
 * Rounding points are the reference's in that dtype (csrc/generic.hip); with MI_DTYPE_FP32 nothing is rounded.  Launch by
 * launch (capturable in a hipGraph by the caller), not tuned to the roofline: BASELINE's configurations are bf16. */
enum mi_dtype { MI_DTYPE_BF16 = 0, MI_DTYPE_FP16 = 1, MI_DTYPE_FP32 = 2 };
size_t mi_workspace_bytes_generic(const mi_model_t* model, int T, int dtype);
int mi_forward_generic(const mi_model_t* model, const mi_batch_t* batch, int dtype, mi_stream_t stream);
/* Leaf operators in any storage dtype - what module-level callers of an fp16 / fp32 model bind (the Pixtral tower of
 * vision_encoder.py:76-204, RMSNorm / FeedForward used stand-alone); argument meaning as the bf16 entry points above.
 * mi_linear_generic: one launch per weight matrix, epilogues STORE / RESIDUAL / LOGITS (SwiGLU = two calls +
 * mi_swiglu_generic: a <- silu(a) * b).  mi_attention_nocache_generic: the cache=None attention, every token sees every
 * token (transformer_layers.py:72-73,165), any head_dim <= 256, softmax_scale <= 0 = head_dim^-1/2. */
int mi_embedding_generic(void* out, const void* table, const int64_t* ids, int T, int D, int vocab, int dtype, mi_stream_t stream);
int mi_rmsnorm_generic(void* out, const void* x, const void* w, int T, int D, float eps, int dtype, mi_stream_t stream);
int mi_linear_generic(void* out, int ldo, const void* x, int ldx, int M, int K, const void* const w[3], const int n_rows[3],
                      int epilogue, const void* residual, int dtype, mi_stream_t stream);
int mi_rope_inplace_generic(void* qkv, int ld, int T, int n_rot_cols, int head_dim, const float* rope_cs, const int32_t* tok_pos,
                            int dtype, mi_stream_t stream);
int mi_attention_nocache_generic(void* out, const void* qkv, int ld, int T, int n_heads, int n_kv_heads, int head_dim,
                                 float softmax_scale, int dtype, mi_stream_t stream);
int mi_swiglu_generic(void* a, const void* b, int T, int F, int dtype, mi_stream_t stream);
int mi_gelu_generic(void* x, int ldx, int T, int N, int dtype, mi_stream_t stream);

/* Persistent decode engine (csrc/decode_engine.hip).  A DECODE-branch mi_forward with T == B == 1 on a dense model runs
 * all local layers - TransformerBlock.forward (transformer_layers.py:158-169) x n_layers, the ring write (cache.py:83-92)
 * and the LM head (transformer.py:235) - as ONE persistent launch when the shapes allow it (dim, hidden_dim and
 * n_heads*128 multiples of 512, dim <= 8192, at most one attention work item per CU); bit-identical to the launch path.
 * mi_set_decode_engine(0) forces the launch path (A/B measurements, tests); returns the previous setting.  Initial value:
 * environment MI_DECODE_ENGINE (default 1). */
int mi_set_decode_engine(int enabled);
/* Residency.  The engine's workgroups wait for each other, so all of them (one per CU) must be resident at once; a plain
 * launch checks nothing.  Two guards: (1) before its first use on a device the library runs a census kernel with the
 * engine's launch shape (one stream synchronisation, outside any capture): if the workgroups do not all meet - a CU mask, a
 * partition mode, a co-tenant - the engine is not used on that device and mi_forward takes the launch path;
 * mi_decode_engine_census(1) forgets the verdicts (tests, or after the device's situation changed).  (2) Every engine
 * launch repeats the census on itself before its first side effect; if it fails, the launch ends having written NOTHING
 * (no ring row, no position advance, no sample), raises status word 1 = 0x700, and every later engine launch on that
 * workspace leaves at once until the caller has run mi_decode_engine_reset - the device state stays exactly that of the
 * first failed step, which the caller can therefore re-run on the launch path (mi_set_decode_engine(0)); the number of
 * steps that did complete is status word 5.  Timeouts AFTER the census (codes 0x100-0x600) should not exist; they poison
 * the workspace the same way but may leave a half-written step. */
int mi_decode_engine_census(int forget);
int mi_decode_engine_reset(void* workspace, mi_stream_t stream);
/* Copies the engine's control words out of a workspace and synchronises `stream` (a health check, NOT part of the hot
 * path): status[0] = step epoch, status[1] = 0 or the code of the first bounded wait that ever timed out
 * (0x100 loader / 0x200 ring / 0x300 consumer barrier / 0x400 hand-off sweep / 0x500-0x600 holder waves / 0x700 residency
 * census of a step: nothing was written), status[2] = abort flag of the last step,
 * status[3] = 0 or 1 + the index of a token whose id was outside [0, vocab) in some mi_forward call (sticky until the
 * caller zeroes the word: the host raises the reference's IndexError from it), status[4] = launches completed by the
 * engine since the workspace was zeroed (one per <= 32 layers of a decode step; unchanged when the launch path ran),
 * status[5] = decode steps run on this workspace (engine: completed; launch path: started) = next row of the greedy
 * history ring, status[6] = workgroup arrivals of the current engine step, status[7] reserved. */
int mi_decode_engine_status(const void* workspace, mi_stream_t stream, uint32_t status[8]);
/* Engine diagnostics (timeline trace, loader knobs, residency-gate test hook) are NOT part of the product boundary:
 * include/mistral_hip_debug.h. */

/* ------------------------------------------------------------------------------------------------
 * Pipeline-parallel exchange steps over RCCL (xGMI between the GPUs of a node)
 *
 * Replaces the reference's torch.distributed calls on the hot path: recv of the previous stage's activations
 * (transformer.py:196), send to the next stage (:214), logits broadcast from the last stage (:237); process-group
 * set-up of main.py:110-118.  One communicator per process (one process per GPU).  Every transfer is enqueued on the
 * caller's HIP stream - no host synchronisation, and capturable in a hipGraph with the decode step.  librccl is
 * dlopen()ed at the first call (MI_RCCL_LIB overrides the name): MI_ERR_UNSUPPORTED when it is absent.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mi_rccl_comm* mi_rccl_t;
#define MI_RCCL_ID_BYTES 128
/* rank 0 creates the 128-byte rendezvous id; the host distributes it to the other ranks (any side channel:
 * torch.distributed object broadcast, a file, MPI) before each rank calls mi_rccl_init with the same id. */
int mi_rccl_unique_id(void* id128);
int mi_rccl_init(mi_rccl_t* comm, int world_size, int rank, const void* id128);
int mi_rccl_destroy(mi_rccl_t comm);
/* point-to-point: `bytes` from/to device memory, matched by the peer's recv/send on ITS stream */
int mi_rccl_send(mi_rccl_t comm, const void* buf, size_t bytes, int peer, mi_stream_t stream);
int mi_rccl_recv(mi_rccl_t comm, void* buf, size_t bytes, int peer, mi_stream_t stream);
/* in-place broadcast of `bytes` from `root` */
int mi_rccl_bcast(mi_rccl_t comm, void* buf, size_t bytes, int root, mi_stream_t stream);
/* ncclGroupStart / ncclGroupEnd: a send and a recv that must progress together (a rank exchanging with itself or with
 * both neighbours at once) are issued between the two */
int mi_rccl_group_start(void);
int mi_rccl_group_end(void);
const char* mi_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MISTRAL_HIP_H */
