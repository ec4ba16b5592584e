// Cost of a device-wide barrier between the phases of one persistent kernel on MI355X (all workgroups
// co-resident: 480 of 256 threads, two per CU), against the gap between two dependent kernel launches
// replayed from a hipGraph -- the two ways of sequencing the towers' 31 same-shape convolutions.
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
  __threadfence();                       // release: this wave's stores reach the memory side
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) { ok = false; break; }   // never hang: give up after ~1 s
    }
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);   // every wave: drop stale lines before reading the others' output
  return ok;
}

__global__ __launch_bounds__(256, 2) void phases(float* buf, unsigned* counter, int nphase, int work) {
  const int n = gridDim.x * blockDim.x, i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  for (int p = 0; p < nphase; ++p) {
    // a token amount of dependent memory work per phase: read what another workgroup wrote in the last phase
    for (int k = 0; k < work; ++k) v += buf[(i + 4099 * (k + 1)) % n];
    buf[i] = v + 1.f;
    grid_barrier(counter, (unsigned)(p + 1) * gridDim.x);
  }
}
__global__ __launch_bounds__(256, 2) void one_phase(float* buf, int work) {
  const int n = gridDim.x * blockDim.x, i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  for (int k = 0; k < work; ++k) v += buf[(i + 4099 * (k + 1)) % n];
  buf[i] = v + 1.f;
}

int main() {
  const int G = 480, NPH = 31;
  float* buf; unsigned* counter;
  hipMalloc(&buf, G * 256 * 4); hipMalloc(&counter, 4);
  hipMemset(buf, 0, G * 256 * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int work : {0, 4}) {
    float ms;
    // persistent kernel with grid barriers
    for (int rep = 0; rep < 3; ++rep) {
      hipMemsetAsync(counter, 0, 4, s);
      hipEventRecord(e0, s);
      hipLaunchKernelGGL(phases, dim3(G), dim3(256), 0, s, buf, counter, NPH, work);
      hipEventRecord(e1, s); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("work %d: one kernel, %d phases with device-wide barriers: %.1f us per phase\n", work, NPH, ms * 1e3 / NPH);
    // the same phases as dependent launches replayed from a graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int p = 0; p < NPH; ++p) hipLaunchKernelGGL(one_phase, dim3(G), dim3(256), 0, s, buf, work);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("work %d: %d dependent launches replayed from a hipGraph:   %.1f us per launch\n", work, NPH, ms * 1e3 / NPH);
  }
  unsigned c; hipMemcpy(&c, counter, 4, hipMemcpyDeviceToHost);
  printf("counter %u (expected %u)\n", c, (unsigned)G * NPH);
  return 0;
}
