// Probe: can two consecutive kernels of the launch path overlap on gfx950, and through which launch form?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/anyorder_probe scripts/probes/anyorder_probe.hip && /tmp/anyorder_probe
// A `spin` kernel (bounded: it gives up after `timeout_ns`) waits for a flag that the NEXT launch sets.  If the spin sees the flag
// before its timeout, the second launch ran while the first was still resident.  Forms: (1) one stream, plain launches (control:
// must time out); (2) one stream, hipExtLaunchKernel with hipExtAnyOrderLaunch on the second launch; (3) two streams; (4)-(6) the
// same three captured into a hipGraph and replayed.  Also: time per launch of a chain of 200 empty dependent kernels in each form.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(unsigned* flag, unsigned long long timeout_ns, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  unsigned seen = 0;
  unsigned long long t = t0;
  while (true) {
    seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_s_memrealtime();
    if (seen || (t - t0) * 10ull > timeout_ns) break;
    __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = seen;
    out[1] = (t - t0) * 10ull;
  }
}
__global__ void set_kernel(unsigned* flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void empty_kernel(unsigned* p) {
  if (p == nullptr) p[threadIdx.x] = 0;
}

static void launch_set(unsigned* flag, hipStream_t s, int any_order) {
  void* args[] = {&flag};
  if (any_order) CK(hipExtLaunchKernel((const void*)set_kernel, dim3(1), dim3(64), args, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch));
  else hipLaunchKernelGGL(set_kernel, dim3(1), dim3(64), 0, s, flag);
}

int main() {
  unsigned* flag;
  unsigned long long* out;
  CK(hipMalloc(&flag, 4));
  CK(hipMalloc(&out, 16));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev, ev2;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
  const unsigned long long timeout_ns = 3000000ull;  // 3 ms
  unsigned long long h[2];

  auto report = [&](const char* what) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    printf("%-58s seen=%llu after %8.1f us -> %s\n", what, h[0], h[1] / 1000.0, h[0] ? "OVERLAP" : "serialized (timed out)");
  };
  for (int blocks : {1, 256, 2048}) {
    printf("--- spin grid %d blocks x 256\n", blocks);
    // (1) control
    CK(hipMemsetAsync(flag, 0, 4, s1));
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s1, flag, timeout_ns, out);
    launch_set(flag, s1, 0);
    report("one stream, plain");
    // (2) any-order
    CK(hipMemsetAsync(flag, 0, 4, s1));
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s1, flag, timeout_ns, out);
    launch_set(flag, s1, 1);
    report("one stream, hipExtAnyOrderLaunch on the setter");
    // (3) two streams
    CK(hipMemsetAsync(flag, 0, 4, s1));
    CK(hipEventRecord(ev, s1));
    CK(hipStreamWaitEvent(s2, ev, 0));
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s1, flag, timeout_ns, out);
    launch_set(flag, s2, 0);
    report("two streams");
    // (4)-(6) graphs
    for (int form = 0; form < 3; ++form) {
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
      CK(hipMemsetAsync(flag, 0, 4, s1));
      if (form == 2) {
        CK(hipEventRecord(ev, s1));
        CK(hipStreamWaitEvent(s2, ev, 0));
      }
      hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s1, flag, timeout_ns, out);
      if (form == 2) {
        launch_set(flag, s2, 0);
        CK(hipEventRecord(ev2, s2));
        CK(hipStreamWaitEvent(s1, ev2, 0));
      } else {
        launch_set(flag, s1, form);
      }
      CK(hipStreamEndCapture(s1, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s1));
      report(form == 0 ? "graph: one stream, plain" : (form == 1 ? "graph: one stream, any-order setter" : "graph: fork / join over two streams"));
      CK(hipGraphLaunch(ge, s1));
      report("   (second replay)");
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
  }

  // chain cost: 200 empty kernels
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  const int N = 200;
  for (int form = 0; form < 4; ++form) {
    float best = 1e9f;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    unsigned* pf = flag;
    void* args[] = {&pf};
    auto chain = [&](int any) {
      for (int i = 0; i < N; ++i) {
        if (any) CK(hipExtLaunchKernel((const void*)empty_kernel, dim3(256), dim3(256), args, 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch));
        else hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s1, flag);
      }
    };
    if (form >= 2) {
      CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
      chain(form == 3);
      CK(hipStreamEndCapture(s1, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s1));
      if (form < 2) chain(form == 1);
      else CK(hipGraphLaunch(ge, s1));
      CK(hipEventRecord(t1, s1));
      CK(hipEventSynchronize(t1));
      float ms;
      CK(hipEventElapsedTime(&ms, t0, t1));
      best = ms < best ? ms : best;
    }
    const char* names[] = {"plain launches", "any-order launches", "graph of plain launches", "graph of any-order launches"};
    printf("chain of %d empty kernels, %-30s %7.2f us per kernel\n", N, names[form], best * 1000.f / N);
  }
  return 0;
}
