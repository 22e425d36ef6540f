// Do a byte-bound and a matrix-bound workgroup overlap when they SHARE a CU?  (the library's contraction kernels hold a
// whole CU each, so the byte-bound and the matrix-bound kernels of different forwards take turns.)  Two synthetic kernels, 256 threads and no
// LDS each - one workgroup of each fits every CU:
//   stream: 2 read streams + 1 write stream over 128-row tiles of row-major matrices, 8 x 16 B per thread in flight (the on-the-fly 1x1
//           kernel's access mix, scripts/micro/stride_stream.hip), grid = 256 persistent workgroups striding over the tiles;
//   mfma:   back-to-back v_mfma_f32_32x32x16_f16 on random operands, one wave per SIMD, grid = 256 workgroups.
// Each alone, then both at once on two streams (started together; the pair's time is the later end).
// build: hipcc --offload-arch=gfx950 -O3 -o bin/coresident coresident.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) stream_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, float* out,
                                                int tiles, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rsub = lane >> 3, cl = lane & 7;               // 8 rows x 128 B per wave instruction
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int ksteps = K / 32;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long total = (long long)ksteps * 4;          // wave instructions per stream and tile (32 rows per wave, 8 at a time)
    for (long long i0 = 0; i0 < total; i0 += 8) {
      f32x4 v[8], u[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const long long i = i0 + d;
        const int ks = (int)(i >> 2), rg = (int)(i & 3);
        const long long row = (long long)tile * 128 + wave * 32 + rg * 8 + rsub;
        const long long off = row * K + (long long)ks * 32 + cl * 4;
        v[d] = *reinterpret_cast<const f32x4*>(a + off);
        u[d] = *reinterpret_cast<const f32x4*>(b + off);
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const long long i = i0 + d;
        const int ks = (int)(i >> 2), rg = (int)(i & 3);
        const long long row = (long long)tile * 128 + wave * 32 + rg * 8 + rsub;
        const f32x4 r = v[d] + u[d];
        acc += r;
        *reinterpret_cast<f32x4*>(c + row * K + (long long)ks * 32 + cl * 4) = r;
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

template <int GAP>      // GAP: s_nop-cycles of idle issue behind every MFMA (0 = back to back; 8 = ~50 % duty of the matrix pipe)
__global__ void __launch_bounds__(256) mfma_k(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(i + u) & 3]), __builtin_bit_cast(f16x8, b[i]), acc[i], 0, 0, 0);
        if constexpr (GAP > 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(GAP); __builtin_amdgcn_sched_barrier(0); }
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int K = 1024, M = 128 * 1024, tiles = M / 128;      // 512 MiB per buffer: far beyond the Infinity Cache
  float *a, *b, *c, *out, *out2;
  u32x4* in;
  const size_t n = (size_t)M * K;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&out, 64); hipMalloc(&out2, 512 * 256 * 4);
  hipMalloc(&in, 4096 * sizeof(u32x4));
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, n * 4);
  unsigned* h = (unsigned*)malloc(4096 * 16);
  for (int i = 0; i < 4096 * 4; ++i) {                      // random fp16 in +-[0.5, 1)
    const unsigned lo = 0x3800u | (rand() & 0x3ff) | ((rand() & 1) << 15), hi = 0x3800u | (rand() & 0x3ff) | ((rand() & 1) << 15);
    h[i] = lo | (hi << 16);
  }
  hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t b1, e1, b2, e2;
  hipEventCreate(&b1); hipEventCreate(&e1); hipEventCreate(&b2); hipEventCreate(&e2);
  const double sbytes = (double)n * 4 * 3;
  auto run = [&](const char* what, bool do_s, int do_m, int wgs_s, int iters, int wgs_m = 256) {      // do_m: 0 none, 1 back-to-back, 2 with gaps
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      if (do_s) { hipEventRecord(b1, s1); hipLaunchKernelGGL(stream_k, dim3(wgs_s), dim3(256), 0, s1, a, b, c, out, tiles, K); hipEventRecord(e1, s1); }
      if (do_m) {
        hipEventRecord(b2, s2);
        if (do_m == 1) hipLaunchKernelGGL(mfma_k<0>, dim3(wgs_m), dim3(256), 0, s2, in, out2, iters);
        else hipLaunchKernelGGL(mfma_k<1>, dim3(wgs_m), dim3(256), 0, s2, in, out2, iters);
        hipEventRecord(e2, s2);
      }
      hipDeviceSynchronize();
    }
    float ms1 = 0.f, ms2 = 0.f;
    if (do_s) hipEventElapsedTime(&ms1, b1, e1);
    if (do_m) hipEventElapsedTime(&ms2, b2, e2);
    printf("%-58s", what);
    if (do_s) printf("  stream %7.1f us (%.2f TB/s)", ms1 * 1e3, sbytes / (ms1 * 1e-3) / 1e12);
    if (do_m) printf("  mfma %7.1f us (%.0f TFLOP/s)", ms2 * 1e3, (double)iters * 16 * wgs_m * 4 * 32768.0 / (ms2 * 1e-3) / 1e12);
    printf("\n");
  };
  const int iters = 900;                                    // ~ the stream kernel's time
  run("stream alone, 256 workgroups", true, 0, 256, iters);
  run("stream alone, 128 workgroups", true, 0, 128, iters);
  run("mfma alone, 256 workgroups (1 wave per SIMD)", false, 1, 256, iters);
  run("mfma alone, 128 workgroups", false, 1, 256, iters, 128);
  run("mfma with an s_sleep behind every MFMA, alone", false, 2, 256, iters);
  run("both at once: stream 256 + mfma 256", true, 1, 256, iters);
  run("both at once: stream 256 + mfma-with-gaps 256", true, 2, 256, iters);
  run("both at once: stream 128 + mfma 128 (room for disjoint CUs)", true, 1, 128, iters, 128);
  run("both at once: stream 256 + mfma 128", true, 1, 256, iters, 128);
  return 0;
}
