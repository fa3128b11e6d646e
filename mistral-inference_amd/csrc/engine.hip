// Persistent fused decode operators: several dependent weight-streaming GEMVs in ONE launch, separated by grid
// barriers, with the next operator's first weight batches issued between barrier arrive and wait (weights never
// depend on activations) - the per-launch fixed cost (about 4 us per operator) is replaced
// by a barrier whose latency is covered by weights already in flight.
#include <cstdlib>

#include "gemv_core.cuh"
#include "grid_barrier.cuh"

using namespace gemv_core;

namespace {

// [RMSNorm + W1|W3 + SiLU*mul] -> barrier -> [W2 + residual]   (transformer_layers.py:105-106,167-168), one token set
template <int TT>
__global__ __launch_bounds__(256, 2) void fused_ffn_kernel(GemvArgs a13, GemvArgs a2, GridBarrierState* bar) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GridBarrier gb{bar, (int)blockIdx.x, (int)gridDim.x, 0u};
  gemv_body<TT, GEMV_SWIGLU, 2, true>(a13, smem, blockIdx.x, gridDim.x, 0, NoSync{});
  gemv_body<TT, GEMV_RESIDUAL, 2, true>(a2, smem, blockIdx.x, gridDim.x, 0, gb);
}

}  // namespace

// debug / probe entry (not part of the public ABI yet)
extern "C" int mi_debug_fused_ffn(void* h, const void* x_in, const void* norm_w, float eps, const void* w1, const void* w3,
                                  const void* w2, void* hid, int D, int F, void* barrier_state, int n_blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GemvArgs a13 = {}, a2 = {};
  a13.mode = GEMV_SWIGLU; a13.T = 1; a13.K = D; a13.N = F; a13.x = (const bf16_t*)x_in; a13.ldx = D;
  a13.norm_w = (const bf16_t*)norm_w; a13.eps = eps; a13.w0 = (const bf16_t*)w1; a13.w1 = (const bf16_t*)w3;
  a13.n0 = a13.n1 = F; a13.out = hid; a13.ldo = F;
  a2.mode = GEMV_RESIDUAL; a2.T = 1; a2.K = F; a2.N = D; a2.x = (const bf16_t*)hid; a2.ldx = F;
  a2.w0 = (const bf16_t*)w2; a2.n0 = a2.n1 = D; a2.out = h; a2.ldo = D; a2.residual = (const bf16_t*)h;
  hipError_t e = hipMemsetAsync(barrier_state, 0, sizeof(GridBarrierState), s);
  if (e != hipSuccess) return (int)e;
  const size_t lds = (size_t)((F > D ? F : D) * 2 + 64);
  hipLaunchKernelGGL((fused_ffn_kernel<1>), dim3(n_blocks), dim3(256), lds, s, a13, a2, (GridBarrierState*)barrier_state);
  return (int)hipGetLastError();
}
