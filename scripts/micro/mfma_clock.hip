// Sustained rate of back-to-back v_mfma_f32_32x32x16_bf16 on every SIMD of the chip (one or two waves per SIMD, operands in
// registers, random or zero data): time per MFMA in nominal 2.4-GHz cycles tells the clock the chip actually holds under
// matrix load - the ceiling for any kernel of csrc/gemm_bf3.hip.   build: hipcc --offload-arch=gfx950 -O3 -o mfma_clock mfma_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) mfma_loop(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + u) & 3]), __builtin_bit_cast(bf16x8, b[i]), acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const int iters = 20000;
  u32x4* in; float* out;
  hipMalloc(&in, 4096 * sizeof(u32x4)); hipMalloc(&out, 1024 * 256 * sizeof(float));
  unsigned* h = (unsigned*)malloc(4096 * 16);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 4096 * 4; ++i) {   // mode 0: zeros; mode 1: random bf16 in [-1, 1)
      unsigned lo = 0x3f00u | (rand() & 0x7f) | ((rand() & 1) << 15), hi = 0x3f00u | (rand() & 0x7f) | ((rand() & 1) << 15);
      h[i] = mode ? (lo | (hi << 16)) : 0u;
    }
    hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int wgs = 256; wgs <= 512; wgs *= 2) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      mfma_loop<<<wgs, 256>>>(in, out, 200); hipDeviceSynchronize();
      hipEventRecord(e0);
      mfma_loop<<<wgs, 256>>>(in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double per_simd = (double)iters * 16 * (wgs / 256);            // MFMAs per SIMD
      const double cyc = ms * 1e-3 * 2.4e9 / per_simd;
      const double tf = (double)iters * 16 * wgs * 4 * 32768.0 / (ms * 1e-3) / 1e12;
      printf("%s operands, %d wave(s)/SIMD: %.2f ms, %.1f nominal cycles per MFMA (32 = 2.4 GHz) -> %.2f GHz, %.0f TFLOP/s bf16 = %.2f of 2.5 PF\n",
             mode ? "random" : "zero", wgs / 256, ms, cyc, 2.4 * 32.0 / cyc, tf, tf / 2500.0);
    }
  }
  return 0;
}
