/* Minimal C consumer of libmistral_hip.so: what a non-Python host (or a cgo / JNI shim) does first.
 *
 *   gcc -std=c99 -I include examples/abi_probe.c -ldl -o abi_probe && ./abi_probe mistral-inference_amd/lib/libmistral_hip.so
 *
 * It resolves the entry points by name, checks the ABI version against the header it was compiled with, and exercises the
 * argument validation of mi_forward (which happens before any device work, so no GPU is needed for this program). */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "mistral_hip.h"

typedef int (*abi_fn)(void);
typedef const char* (*str_fn)(int);
typedef const char* (*detail_fn)(void);
typedef int (*forward_fn)(const mi_model_t*, const mi_batch_t*, mi_stream_t);
typedef size_t (*ws_fn)(const mi_model_t*, int, int, int);

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "libmistral_hip.so";
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fprintf(stderr, "dlopen(%s): %s\n", path, dlerror());
    return 2;
  }
  abi_fn abi = (abi_fn)dlsym(h, "mi_abi_version");
  str_fn err = (str_fn)dlsym(h, "mi_error_string");
  detail_fn detail = (detail_fn)dlsym(h, "mi_last_error_detail");
  forward_fn forward = (forward_fn)dlsym(h, "mi_forward");
  ws_fn ws = (ws_fn)dlsym(h, "mi_workspace_bytes");
  if (!abi || !err || !detail || !forward || !ws) {
    fprintf(stderr, "missing symbol\n");
    return 3;
  }
  /* ABI v3: fused decode / MoE leaf operators, the RCCL transport and the decode-engine controls are part of the surface */
  static const char* const v3[] = {"mi_qkv_rope_kvwrite", "mi_moe_experts_decode", "mi_moe_grouped_gemm", "mi_rccl_unique_id",
                                   "mi_rccl_init", "mi_rccl_send", "mi_rccl_recv", "mi_rccl_bcast", "mi_rccl_destroy",
                                   "mi_set_decode_engine", "mi_decode_engine_status"};
  for (size_t i = 0; i < sizeof(v3) / sizeof(v3[0]); ++i)
    if (!dlsym(h, v3[i])) {
      fprintf(stderr, "missing symbol %s\n", v3[i]);
      return 3;
    }
  if (abi() != MI_ABI_VERSION) {
    fprintf(stderr, "ABI %d, header %d\n", abi(), MI_ABI_VERSION);
    return 4;
  }
  /* Mistral-7B shapes: the library sizes the caller-owned workspace for a 4096-token prefill */
  mi_layer_t layer;
  mi_model_t m;
  mi_batch_t b;
  memset(&layer, 0, sizeof(layer));
  memset(&m, 0, sizeof(m));
  memset(&b, 0, sizeof(b));
  m.dim = 4096; m.n_heads = 32; m.n_kv_heads = 8; m.head_dim = 128; m.hidden_dim = 14336; m.vocab_size = 32768;
  m.n_layers = 1; m.norm_eps = 1e-5f; m.layers = &layer;
  printf("abi %d; workspace for T=4096, B=1, W=4096: %zu bytes\n", abi(), ws(&m, 4096, 1, 4096));
  /* an unsupported head size is refused with a message, before anything is launched */
  m.head_dim = 96;
  int rc = forward(&m, &b, NULL);
  printf("head_dim 96 -> rc %d (%s): %s\n", rc, err(rc), detail());
  return rc == MI_ERR_SHAPE ? 0 : 5;
}
