// How fast can 128-row tiles of a row-major fp32 matrix [M][K] be streamed when each workgroup walks its rows along K in pieces of
// PB bytes per row?  The on-the-fly-operand convolution (csrc/gemm_bf3.hip, OPK_ROWK_BN) reads its input as 8 rows x 128 B per wave
// instruction (PB = 128): 128 concurrent sequential streams per workgroup, each advancing one cache line per K tile.  In the ResNet
// forward that kernel sits at 3.1 TB/s of algorithmic bytes; this probe separates the access pattern from everything else.
//   pattern PB = 128 / 256 / 512 / 1024: a wave instruction covers 1024 / PB rows x PB bytes; one workgroup = 128 rows, 4 waves;
//   every thread keeps DEPTH 16-byte loads in flight; buffers far beyond the 256-MiB Infinity Cache; optional second read stream
//   (the residual) and a write stream of the same shape (the fp32 copy).
// build: hipcc --offload-arch=gfx950 -O3 -o bin/stride_stream stride_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PB, int DEPTH, int STREAMS, int WRITE>
__global__ void __launch_bounds__(256) stream(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, float* out, int M, int K) {
  constexpr int LPR = PB / 16;            // lanes per row
  constexpr int RPI = 64 / LPR;           // rows per wave instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;            // 128 rows
  const int rsub = lane / LPR, cl = lane % LPR;
  // wave w owns rows w*32 .. w*32+31 of the tile, RPI at a time
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int ksteps = K * 4 / PB;          // pieces along a row
  const int rgroups = 32 / RPI;
  const long long total = (long long)ksteps * rgroups;      // wave instructions per stream
  for (long long i0 = 0; i0 < total; i0 += DEPTH) {
    f32x4 v[DEPTH], u[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long long i = i0 + d < total ? i0 + d : total - 1;
      const int ks = (int)(i / rgroups), rg = (int)(i % rgroups);      // K-major walk: all row groups of a K piece, then the next piece
      const long long row = (long long)tile * 128 + wave * 32 + rg * RPI + rsub;
      const long long off = row * K + (long long)ks * (PB / 4) + cl * 4;
      v[d] = *reinterpret_cast<const f32x4*>(a + off);
      if (STREAMS > 1) u[d] = *reinterpret_cast<const f32x4*>(b + off);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      f32x4 r = v[d];
      if (STREAMS > 1) r += u[d];
      acc += r;
      if (WRITE) {
        const long long i = i0 + d < total ? i0 + d : total - 1;
        const int ks = (int)(i / rgroups), rg = (int)(i % rgroups);
        const long long row = (long long)tile * 128 + wave * 32 + rg * RPI + rsub;
        *reinterpret_cast<f32x4*>(c + row * K + (long long)ks * (PB / 4) + cl * 4) = r;
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

template <int PB, int DEPTH, int STREAMS, int WRITE>
static void run(const char* name, const float* a, const float* b, float* c, float* out, int M, int K) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<PB, DEPTH, STREAMS, WRITE>), dim3(M / 128), dim3(256), 0, 0, a, b, c, out, M, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)M * K * 4.0 * (STREAMS + WRITE);
  printf("%-44s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 1024;
  const int M = argc > 2 ? atoi(argv[2]) : 128 * 1024;          // 128K rows x 4 KB = 512 MiB per buffer
  float *a, *b, *c, *out;
  const size_t n = (size_t)M * K;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&out, 64);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, n * 4);
  printf("M = %d rows x K = %d floats (%.0f MiB per buffer), %d workgroups of 128 rows\n", M, K, n * 4.0 / (1 << 20), M / 128);
  run<128, 8, 1, 0>("1 read stream, 8 rows x 128 B, depth 8", a, b, c, out, M, K);
  run<256, 8, 1, 0>("1 read stream, 4 rows x 256 B, depth 8", a, b, c, out, M, K);
  run<512, 8, 1, 0>("1 read stream, 2 rows x 512 B, depth 8", a, b, c, out, M, K);
  run<1024, 8, 1, 0>("1 read stream, 1 row x 1 KB, depth 8", a, b, c, out, M, K);
  run<128, 16, 1, 0>("1 read stream, 8 rows x 128 B, depth 16", a, b, c, out, M, K);
  run<128, 8, 2, 0>("2 read streams, 8 rows x 128 B, depth 8", a, b, c, out, M, K);
  run<512, 8, 2, 0>("2 read streams, 2 rows x 512 B, depth 8", a, b, c, out, M, K);
  run<128, 8, 2, 1>("2 read + 1 write stream, 8 rows x 128 B", a, b, c, out, M, K);
  run<256, 8, 2, 1>("2 read + 1 write stream, 4 rows x 256 B", a, b, c, out, M, K);
  run<512, 8, 2, 1>("2 read + 1 write stream, 2 rows x 512 B", a, b, c, out, M, K);
  run<1024, 8, 2, 1>("2 read + 1 write stream, 1 row x 1 KB", a, b, c, out, M, K);
  return 0;
}
