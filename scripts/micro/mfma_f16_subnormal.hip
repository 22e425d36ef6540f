// Does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 subnormal inputs?  A = one subnormal value s everywhere, B = 1024 everywhere:
// every output element must be 16 * s * 1024 (exactly representable); a flushing matrix core would give 0.
// hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_f16_subnormal.hip -o scripts/micro/bin/mfma_f16_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* vals, float* out, int n) {
  for (int t = 0; t < n; ++t) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = vals[t]; b[i] = (_Float16)1024.f; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[t] = acc[0];
  }
}
int main() {
  const int n = 6;
  _Float16 h[n];
  const float src[n] = {1.0f, 6.103515625e-05f /* 2^-14 min normal */, 3.0517578125e-05f /* 2^-15 */, 9.5367431640625e-07f /* 2^-20 */,
                        5.9604644775390625e-08f /* 2^-24 smallest subnormal */, 1.7881393432617188e-07f /* 3 * 2^-24 */};
  for (int i = 0; i < n; ++i) h[i] = (_Float16)src[i];
  _Float16* dv; float* dout; float out[n];
  hipMalloc(&dv, sizeof(h)); hipMalloc(&dout, sizeof(out));
  hipMemcpy(dv, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dv, dout, n);
  hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const float want = 16.f * (float)h[i] * 1024.f;
    printf("a = %.10e  ->  %.10e  (exact %.10e)%s\n", (float)h[i], out[i], want, out[i] == want ? "" : "   <-- differs");
    bad += out[i] != want;
  }
  printf(bad ? "fp16 subnormal inputs are NOT honoured\n" : "fp16 subnormal inputs are honoured\n");
  return 0;
}
