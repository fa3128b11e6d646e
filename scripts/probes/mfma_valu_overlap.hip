// Micro-probe: how do MFMA bursts and VALU work of the waves of one SIMD overlap on gfx950?
// One workgroup per CU, WPS waves per SIMD (4 * WPS waves).  Per iteration a wave issues 16 MFMAs on 2 accumulators
// (the S^T phase of the prefill attention), NV dependent-free VALU fmas (+ NE v_exp), 16 MFMAs on 4 accumulators (P.V).
// Prints shader cycles per iteration for several (WPS, NV, NE, stagger).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_overlap.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV, int NE, int STAG>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + lane * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
  f32x16 s0 = {}, s1 = {}, o0 = {}, o1 = {}, o2 = {}, o3 = {};
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = seed + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (STAG && wid >= 4 && it == 0) {  // half a phase of head start difference
      for (int i = 0; i < 16; ++i) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 31] = __builtin_fmaf(v[i & 31], 1.0001f, s0[i & 15] * 1e-30f);
#pragma unroll
    for (int i = 0; i < NE; ++i) v[i & 31] = __builtin_amdgcn_exp2f(v[i & 31] * 1e-3f);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 p = b;
    p[0] = (__bf16)v[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, o1, 0, 0, 0);
      o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, o2, 0, 0, 0);
      o3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, o3, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += s0[i] + s1[i] + o0[i] + o1[i] + o2[i] + o3[i];
  for (int i = 0; i < 32; ++i) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NE, int STAG>
void run(int waves, const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<NV, NE, STAG>), dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters, 1.0f);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NV, NE, STAG>), dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
  const double per_simd_mfma = 32.0 * (waves / 4);
  printf("%-28s waves/SIMD %d  NV %3d NE %2d stag %d: %7.0f cycles/iter (MFMA-only floor %4.0f)  %.1f cycles per MFMA  clock %.0f MHz  %.0f TFLOP/s\n", name, waves / 4, NV, NE,
         STAG, c / iters, per_simd_mfma * 32, c / iters / per_simd_mfma, c / (ms * 1e3), 256.0 * waves * 32 * iters * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0, 0>(4, "mfma only");
  run<0, 0, 0>(8, "mfma only");
  run<64, 0, 0>(4, "64 fma");
  run<64, 0, 0>(8, "64 fma");
  run<128, 0, 0>(4, "128 fma");
  run<128, 0, 0>(8, "128 fma");
  run<128, 0, 1>(8, "128 fma, staggered");
  run<96, 32, 0>(4, "96 fma + 32 exp");
  run<96, 32, 0>(8, "96 fma + 32 exp");
  run<96, 32, 1>(8, "96 fma + 32 exp, staggered");
  run<160, 32, 0>(8, "160 fma + 32 exp");
  return 0;
}
