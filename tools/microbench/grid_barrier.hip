// Cost of a device-wide barrier inside one persistent kernel on MI355X (8 XCDs, non-coherent L2s):
// every workgroup publishes a value, arrives on a monotonic counter with agent-scope release, spins with
// agent-scope acquire, then reads a value another workgroup (on another XCD) published.  Compared with the
// 1.56 us empty-kernel boundary of launch_floor.hip this decides whether a persistent decoder-layer kernel pays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, unsigned* timeout_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22)) { *timeout_flag = 1; break; }   // bounded: never hang the box
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) barrier_loop(unsigned* counter, float* slots, int iters, unsigned* timeout_flag, float* sink) {
  const unsigned G = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (threadIdx.x == 0) slots[blockIdx.x * 16] = (float)(it + blockIdx.x);
    grid_barrier(counter, (unsigned)(it + 1) * G, timeout_flag);
    acc += slots[((blockIdx.x + 1 + (it & 7)) % G) * 16];      // neighbour on another XCD
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

// variant: relaxed arrive + explicit fences only (to see what the release/acquire cache maintenance costs)
__global__ void __launch_bounds__(256) barrier_loop_relaxed(unsigned* counter, int iters, unsigned* timeout_flag) {
  const unsigned G = gridDim.x;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0, target = (unsigned)(it + 1) * G;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        if (++spins > (1u << 22)) { *timeout_flag = 1; break; }
    }
    __syncthreads();
  }
}

int main() {
  unsigned *counter, *flag; float *slots, *sink;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&slots, 4096 * 64)); CK(hipMalloc(&sink, 4096 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {64, 128, 256, 512}) {
    for (int variant = 0; variant < 2; ++variant) {
      const int iters = 2000;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(flag, 0, 4));
        CK(hipEventRecord(e0));
        if (variant == 0) barrier_loop<<<G, 256>>>(counter, slots, iters, flag, sink);
        else barrier_loop_relaxed<<<G, 256>>>(counter, iters, flag);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      unsigned f; CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
      printf("G=%3d %-16s %.3f us/barrier%s\n", G, variant == 0 ? "release/acquire" : "relaxed", best * 1e3 / iters, f ? "  (TIMEOUT hit)" : "");
    }
  }
  return 0;
}
