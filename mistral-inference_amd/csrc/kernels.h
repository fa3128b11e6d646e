// Internal (non-ABI) declarations shared by the kernel translation units and the runner.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

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
constexpr size_t GEMV_LDS_BUDGET = 64 * 1024 - 256;

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
  // MoE
  const void* const* expert_tab;  // device [E][3]
  const int32_t* sel_idx;         // device [T*top_k]
  const float* sel_w;             // device [T*top_k]
  int top_k;
};

int gemv_max_tokens(int K);
hipError_t launch_gemv(const GemvArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------- GEMM
enum GemmEpi { GEMM_STORE = 0, GEMM_RESIDUAL = 1, GEMM_SWIGLU = 2, GEMM_LOGITS = 3, GEMM_MOE_ACCUM = 4 };

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
  // grouped form (MoE prefill): this launch handles compact rows [base, base + count) of an expert
  const int32_t* m_count;     // device: actual row count (<= M) or nullptr
  const int32_t* row_base;    // device: first compact row of this expert, or nullptr (0)
  const int32_t* a_gather;    // device: A row = a_gather[base + m]   (nullptr: base + m)
  const int32_t* out_scatter; // device: out row = out_scatter[base + m] (nullptr: base + m)
  const float* row_scale;     // device: expert weight of compact row base + m (MOE_ACCUM)
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------- attention
struct AttnDecodeArgs {
  void* out;            // [B, H*Dh]
  const bf16_t* q;      // [B, ldq]
  int ldq;
  const bf16_t* cache_k;  // [maxB, W, Hkv*Dh]
  const bf16_t* cache_v;
  int W, B, H, Hkv, Dh;
  const int32_t* tok_pos;  // [B]
  float* partial;          // scratch
  int32_t* tickets;        // [B*Hkv], zero on entry, left zero
  int n_splits;
};
int attn_decode_splits(int W);
size_t attn_decode_partial_floats(int B, int H, int Hkv, int Dh, int W);
hipError_t launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s);

struct AttnPrefillArgs {
  void* out;            // [T, H*Dh]
  const bf16_t* qkv;    // [T, ld]  q | k | v (post-RoPE)
  int ld;
  const bf16_t* cache_k;
  const bf16_t* cache_v;
  int W, B, max_q_len, H, Hkv, Dh;
  const int32_t* q_start;    // [B+1]
  const int32_t* kv_before;  // [B]
  int causal;
};
hipError_t launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------- elementwise
hipError_t launch_embedding(void* out, const void* table, const int64_t* ids, int T, int D, int vocab, hipStream_t s);
hipError_t launch_rmsnorm(void* out, const void* x, const void* w, int T, int D, float eps, hipStream_t s);
hipError_t launch_rope(void* qkv, int ld, int T, int H, int Hkv, int Dh, const float* rope_cs, const int32_t* tok_pos,
                       hipStream_t s);
hipError_t launch_kv_write(void* ck, void* cv, int W, const void* k, const void* v, int ld, int T, int kv_dim,
                           const int32_t* tok_seq, const int32_t* tok_pos, const int32_t* q_start, hipStream_t s);
hipError_t launch_decode_prep(int64_t* kv_seqlens, int32_t* q_start, int32_t* kv_before, int32_t* tok_seq,
                              int32_t* tok_pos, int B, hipStream_t s);
hipError_t launch_add_rows(void* out, const void* a, const void* b, size_t n, hipStream_t s);

// ---------------------------------------------------------------------------------------------- MoE
hipError_t launch_moe_router(int32_t* sel_idx, float* sel_w, const void* x, int ldx, int T, int D, const void* gate,
                             int E, int top_k, const void* norm_w, float eps, hipStream_t s);
// builds per-expert token lists for the grouped prefill GEMMs
hipError_t launch_moe_lists(const int32_t* sel_idx, const float* sel_w, int T, int E, int top_k, int32_t* counts,
                            int32_t* offsets, int32_t* tok_of, float* w_of, hipStream_t s);
