// Which bf16 MFMA shape does more work under the chip's power / clock limit?  v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16,
// same FLOP per instruction-cycle (1024 FLOP/clk/SIMD), on every SIMD of the chip, random operands:
//   bare     operands in registers (the pure matrix-pipe ceiling);
//   lds      the fragment traffic of csrc/gemm_bf3.hip's consumer waves: one ds_read_b128 per two 32x32x16 MFMAs
//            (= per four 16x16x32 MFMAs), 64x64 wave tile, conflict-free LDS image;
//   lds+idle the same with the matrix pipe ~55 % idle (s_sleep after every K tile), i.e. the duty cycle of the real kernels.
// Prints TFLOP/s per variant: the ratio of the two shapes at equal structure is what a conversion of the kernels could gain.
// build: hipcc --offload-arch=gfx950 -O3 -o bin/mfma_shape_power mfma_shape_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int LDS, int IDLE>      // SHAPE 32 / 16; LDS 0 / 1; IDLE 0 / 1
__global__ void __launch_bounds__(256) loop(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ u32x4 sm[4096];                  // 64 KB
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // one K tile of the real kernel: 48 MFMAs (2 k-steps x 2x2 blocks x 6 products), 24 fragment reads
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (LDS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {       // 12 reads per k-step: 4 here x 3 planes
            a[i] = sm[(wave * 1024 + ((it * 2 + ks) & 3) * 256 + i * 64 + lane) & 4095];
            b[i] = sm[(wave * 1024 + ((it * 2 + ks + 1) & 3) * 256 + i * 64 + lane) & 4095];
          }
          u32x4 c = sm[(wave * 1024 + 512 + lane + it) & 4095], d = sm[(wave * 1024 + 640 + lane + it) & 4095];
          u32x4 e = sm[(wave * 1024 + 768 + lane + it) & 4095], f = sm[(wave * 1024 + 896 + lane + it) & 4095];
          a[0].x ^= c.x & 1u; b[0].x ^= d.x & 1u; a[1].x ^= e.x & 1u; b[1].x ^= f.x & 1u;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + u) & 3]), __builtin_bit_cast(bf16x8, b[i]), acc[i], 0, 0, 0);
      }
      if constexpr (IDLE) __builtin_amdgcn_s_sleep(28);       // 28 x 64 = 1792 cycles idle per 1536 cycles of MFMA
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // the same K tile with 16x16x32: 96 MFMAs (4x4 blocks x 6 products), the same 24 fragment reads
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (LDS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            a[i] = sm[(wave * 1024 + ((it * 2 + ks) & 3) * 256 + i * 64 + lane) & 4095];
            b[i] = sm[(wave * 1024 + ((it * 2 + ks + 1) & 3) * 256 + i * 64 + lane) & 4095];
          }
          u32x4 c = sm[(wave * 1024 + 512 + lane + it) & 4095], d = sm[(wave * 1024 + 640 + lane + it) & 4095];
          u32x4 e = sm[(wave * 1024 + 768 + lane + it) & 4095], f = sm[(wave * 1024 + 896 + lane + it) & 4095];
          a[0].x ^= c.x & 1u; b[0].x ^= d.x & 1u; a[1].x ^= e.x & 1u; b[1].x ^= f.x & 1u;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(i + u) & 3]), __builtin_bit_cast(bf16x8, b[(i >> 2) & 3]), acc[i], 0, 0, 0);
      }
      if constexpr (IDLE) __builtin_amdgcn_s_sleep(28);
    }
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int LDS, int IDLE>
static void run(const char* name, const u32x4* in, float* out, int wgs) {
  const int iters = IDLE ? 6000 : 12000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    loop<SHAPE, LDS, IDLE><<<wgs, 256>>>(in, out, rep ? iters : 300);
    hipDeviceSynchronize();
  }
  hipEventRecord(e0);
  for (int rep = 0; rep < 4; ++rep) loop<SHAPE, LDS, IDLE><<<wgs, 256>>>(in, out, iters);      // ~100-300 ms: long enough for the clock to settle
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 4.0 * iters * 48.0 * 32768.0 * 4 * wgs;       // per K tile and wave: 48 x 32768 (= 96 x 16384) FLOP
  printf("%-34s %3d workgroups: %8.2f ms  %7.0f TFLOP/s bf16 = %.3f of 2.5 PF\n", name, wgs, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15);
}

int main() {
  u32x4* in; float* out;
  hipMalloc(&in, 4096 * sizeof(u32x4)); hipMalloc(&out, 1024 * 256 * sizeof(float));
  unsigned* h = (unsigned*)malloc(4096 * 16);
  for (int i = 0; i < 4096 * 4; ++i) {
    unsigned lo = 0x3f00u | (rand() & 0x7f) | ((rand() & 1) << 15), hi = 0x3f00u | (rand() & 0x7f) | ((rand() & 1) << 15);
    h[i] = lo | (hi << 16);
  }
  hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  for (int wgs : {256, 196, 98}) {
    run<32, 0, 0>("32x32x16 bare", in, out, wgs);
    run<16, 0, 0>("16x16x32 bare", in, out, wgs);
    run<32, 1, 0>("32x32x16 + LDS fragment reads", in, out, wgs);
    run<16, 1, 0>("16x16x32 + LDS fragment reads", in, out, wgs);
    run<32, 1, 1>("32x32x16 + LDS, pipe ~45% busy", in, out, wgs);
    run<16, 1, 1>("16x16x32 + LDS, pipe ~45% busy", in, out, wgs);
  }
  return 0;
}
