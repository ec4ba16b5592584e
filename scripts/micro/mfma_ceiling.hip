// Sustained rate of v_mfma_f32_16x16x32_f16 on every SIMD of the chip (one wave per SIMD, 8 independent
// accumulators, operands in registers), on zero, small-random and wide-random operand data: the
// practical ceiling the convolution kernels' MFMA-busy fractions are to be read against
// (profiles/r03_ablation.md).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_ceiling mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void mfma_loop(const f16x8* __restrict__ in, f32x4* __restrict__ out, int iters) {
  const int t = threadIdx.x;
  f16x8 a[2], b[NACC];
  a[0] = in[t]; a[1] = in[256 + t];
#pragma unroll
  for (int i = 0; i < NACC; ++i) b[i] = in[512 + ((i * 256 + t) & 2047)];
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rep & 1], b[i], acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + t] = s;
}

int main() {
  const int n = 512 + 2048;
  std::vector<_Float16> h(n * 8);
  f16x8* din; f32x4* dout;
  hipMalloc(&din, n * 16); hipMalloc(&dout, 256 * 256 * 16 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"zeros", "random in [-1e-3, 1e-3]", "random in [-4, 4]"};
  for (int mode = 0; mode < 3; ++mode) {
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
      const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
      h[i] = (_Float16)(mode == 0 ? 0.f : mode == 1 ? u * 1e-3f : u * 4.f);
    }
    hipMemcpy(din, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) {
      const int iters = 20000;
      hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, 0, din, dout, 2000);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfmas = (double)iters * 32;                       // per wave
      const double flops = mfmas * 16384.0 * blocks * 4;
      const double ns_per = ms * 1e6 / mfmas / (blocks / 256.0);
      printf("%-26s %3d workgroups: %8.3f ms  %7.1f TFLOP/s  %.2f ns per MFMA per SIMD (16 cycles at 2.4 GHz = 6.67 ns)\n",
             names[mode], blocks, ms, flops / ms / 1e9, ns_per);
    }
  }
  return 0;
}
