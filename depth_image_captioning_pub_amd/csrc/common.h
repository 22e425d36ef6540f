// Shared host/device helpers for libdic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace dic {

// ---- error reporting across the C ABI: never throw, return negative codes -------------------
enum : int { DIC_OK = 0, DIC_ERR_ARG = -1, DIC_ERR_HIP = -2, DIC_ERR_WORKSPACE = -3, DIC_ERR_UNSUPPORTED = -4 };

void set_last_error(const char* fmt, ...);
const char* last_error();

#define DIC_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::dic::set_last_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return ::dic::DIC_ERR_HIP;                                                         \
    }                                                                                    \
  } while (0)

#define DIC_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      ::dic::set_last_error(__VA_ARGS__);     \
      return ::dic::DIC_ERR_ARG;              \
    }                                         \
  } while (0)

#define DIC_TRY(expr)            \
  do {                           \
    int _rc = (expr);            \
    if (_rc != 0) return _rc;    \
  } while (0)

#define DIC_LAUNCH_CHECK() DIC_CHECK_HIP(hipGetLastError())

// ---- workspace carving (caller-owned device memory, 256-B aligned slices) --------------------
struct Carver {
  char* base;
  size_t off = 0;
  size_t cap;
  bool overflow = false;
  Carver(void* p, size_t bytes) : base(static_cast<char*>(p)), cap(bytes) {}
  template <typename T>
  T* take(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
    T* r = reinterpret_cast<T*>(base + off);
    off += bytes;
    if (base != nullptr && off > cap) overflow = true;
    return base ? r : nullptr;
  }
};

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {   // sum over each 32-lane half
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// fp32 -> three bf16 planes (hi, mid, lo), round-to-nearest-even at each step; hi+mid+lo == x exactly
__device__ __forceinline__ unsigned short bf16_rne_bits(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);       // inf / nan: truncate
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ void split3_bf16(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = bf16_rne_bits(v);
  const float r1 = v - bf16_bits_to_f32(h);
  m = bf16_rne_bits(r1);
  l = bf16_rne_bits(r1 - bf16_bits_to_f32(m));
}

constexpr float kF16ActScale = 4.0f;       // f16x2 ACTIVATION planes hold 4 * x: |x| up to 16376 (larger values become inf: a loud failure); weights carry a
                                           // per-layer scale that puts their largest magnitude in (2^13, 2^14]
constexpr float kF16Max = 65504.0f;        // largest finite fp16: a plane value beyond it is inf and every product it enters inf - inf = NaN
// Overflow guard of the f16x2 format.  Every kernel that WRITES f16x2 planes (bn_apply_planes, bn_relu_maxpool, the producer waves of the
// on-the-fly-operand convolution, split_f16x2_paired) checks |s * v| <= 65504 for the values it splits and raises a caller-owned status
// word otherwise (NaN counts as a violation).  The word is what keeps the failure loud: downstream ReLUs are fmaxf(v, 0), which turn the
// NaN of an overflowed product into 0.  dic_resnet_fwd* fill their output with NaN when it is set; dic_adamw_step / dic_bn_ema_update
// skip their update when given the word (include/dic.h).  Rare path: one atomic per offending thread.
// Bits of the word say who raised it: 1 bn_apply_planes, 2 bn_relu_maxpool, 4 producer waves of the on-the-fly operand, 8 BatchNorm
// statistics not finite, 16 split_f16x2_paired (any non-zero value means "raised").
__device__ __forceinline__ void f16x2_raise(unsigned* status, unsigned who = 1u) { if (status) atomicOr(status, who); }
__device__ __forceinline__ bool f16x2_out_of_range(float v, float s) { return !(fabsf(v) * s <= kF16Max); }

// fp32 -> two fp16 planes of s*v (s a power of two chosen by the caller): h1 = rn(s*v), h2 = rn(s*v - h1); |s*v - h1 - h2| <= 2^-22 |s*v|
// (h2 may be subnormal: the matrix cores honour it).  Values beyond the fp16 range become inf - the caller's scale must prevent that.
__device__ __forceinline__ void split2_f16(float v, float s, unsigned short& h1, unsigned short& h2) {
  const float x = v * s;
  const _Float16 a = (_Float16)x;
  const _Float16 b = (_Float16)(x - (float)a);
  h1 = __builtin_bit_cast(unsigned short, a);
  h2 = __builtin_bit_cast(unsigned short, b);
}

// Bijective XCD-aware block remap (guide T1): blocks b and b+8 share an XCD (speed only).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}
#endif

}  // namespace dic
