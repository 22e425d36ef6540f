// Reproducer for the packed-fp32 operand-select defect described in build.py (scripts/diag_pk_fp32_opsel.py).  This file is
// the one place where the affected instruction forms are written on purpose; build.py's audit exempts it by name.
#include "common.h"
#include <cstdint>

namespace dic {
// Modes 0..4: eight wave-uniform ds_read2_b32 pairs of a known LDS image, consumed behind counted / full lgkmcnt waits by packed
// FMAs (low result from the pair's second dword, as conv1_depth.hip's first form did) or by scalar adds.  Modes 5..12: packed
// fp32 instructions on register operands with every operand-select form.  bad[0] counts lanes whose result is wrong.
typedef float probe_f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) lds_wait_probe_kernel(unsigned* bad, int iters, int mode) {
  extern __shared__ float img[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) img[i] = (float)((i * 7 + 3) % 61);
  __syncthreads();
  const probe_f2 w = {1.0f, 1.0f};
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    const int base = ((it * 37 + (tid >> 6) * 11) % 250) * 16;          // wave-uniform: 16 consecutive floats
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(img + base);
    probe_f2 acc = {0.f, 0.f}, t0, t1, t2, t3, t4, t5, t6, t7;
    if (mode >= 2) {
    } else if (mode == 0) {
      asm volatile(
          "ds_read2_b32 %1, %9 offset0:0 offset1:1\n\tds_read2_b32 %2, %9 offset0:2 offset1:3\n\t"
          "ds_read2_b32 %3, %9 offset0:4 offset1:5\n\tds_read2_b32 %4, %9 offset0:6 offset1:7\n\t"
          "ds_read2_b32 %5, %9 offset0:8 offset1:9\n\tds_read2_b32 %6, %9 offset0:10 offset1:11\n\t"
          "ds_read2_b32 %7, %9 offset0:12 offset1:13\n\tds_read2_b32 %8, %9 offset0:14 offset1:15\n\t"
          "s_waitcnt lgkmcnt(7)\n\tv_pk_fma_f32 %0, %10, %1, %0 op_sel:[0,1,0]\n\t"
          "s_waitcnt lgkmcnt(6)\n\tv_pk_fma_f32 %0, %10, %2, %0 op_sel:[0,1,0]\n\t"
          "s_waitcnt lgkmcnt(5)\n\tv_pk_fma_f32 %0, %10, %3, %0 op_sel:[0,1,0]\n\t"
          "s_waitcnt lgkmcnt(4)\n\tv_pk_fma_f32 %0, %10, %4, %0 op_sel:[0,1,0]\n\t"
          "s_waitcnt lgkmcnt(3)\n\tv_pk_fma_f32 %0, %10, %5, %0 op_sel:[0,1,0]\n\t"
          "s_waitcnt lgkmcnt(2)\n\tv_pk_fma_f32 %0, %10, %6, %0 op_sel:[0,1,0]\n\t"
          "s_waitcnt lgkmcnt(1)\n\tv_pk_fma_f32 %0, %10, %7, %0 op_sel:[0,1,0]\n\t"
          "s_waitcnt lgkmcnt(0)\n\tv_pk_fma_f32 %0, %10, %8, %0 op_sel:[0,1,0]"
          : "+v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
          : "v"(addr), "v"(w)
          : "memory");
    } else {
      asm volatile(
          "ds_read2_b32 %1, %9 offset0:0 offset1:1\n\tds_read2_b32 %2, %9 offset0:2 offset1:3\n\t"
          "ds_read2_b32 %3, %9 offset0:4 offset1:5\n\tds_read2_b32 %4, %9 offset0:6 offset1:7\n\t"
          "ds_read2_b32 %5, %9 offset0:8 offset1:9\n\tds_read2_b32 %6, %9 offset0:10 offset1:11\n\t"
          "ds_read2_b32 %7, %9 offset0:12 offset1:13\n\tds_read2_b32 %8, %9 offset0:14 offset1:15\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_pk_fma_f32 %0, %10, %1, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %10, %2, %0 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %0, %10, %3, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %10, %4, %0 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %0, %10, %5, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %10, %6, %0 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %0, %10, %7, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %10, %8, %0 op_sel:[0,1,0]"
          : "+v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
          : "v"(addr), "v"(w)
          : "memory");
    }
    if (mode >= 5) {      // no LDS at all: packed fp32 instructions on register operands with known values
      const float a0 = (float)((it * 13 + tid) % 17), a1 = (float)((it * 5 + 2 * tid) % 19);
      probe_f2 pa = {a0, a1}, pb = {3.f, 5.f}, r = {1.f, 2.f};
      float want_x, want_y;
      if (mode == 5) {          // v_pk_fma_f32 with op_sel:[0,1,0] (src1's upper register feeds both results)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(r) : "v"(pb), "v"(pa));
        want_x = 1.f + 8.f * 3.f * a1; want_y = 2.f + 8.f * 5.f * a1;
      } else if (mode == 6) {   // plain v_pk_fma_f32
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(pb), "v"(pa));
        want_x = 1.f + 8.f * 3.f * a0; want_y = 2.f + 8.f * 5.f * a1;
      } else if (mode >= 8) {   // other operand-select forms (expected values follow the ISA's op_sel / op_sel_hi semantics)
        const probe_f2 pc = {7.f, 11.f};
        probe_f2 d = {0.f, 0.f};
        if (mode == 8) {        // op_sel_hi:[1,0,1]: high result takes src1's LOW half (the compiler's usual broadcast)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(pb), "v"(pa), "v"(pc));
          want_x = 3.f * a0 + 7.f; want_y = 5.f * a0 + 11.f;
        } else if (mode == 9) { // v_pk_mul_f32 op_sel:[1,0]: low result takes src0's HIGH half
          asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(pa), "v"(pb));
          want_x = a1 * 3.f; want_y = a1 * 5.f;
        } else if (mode == 10) {// v_pk_add_f32 op_sel:[0,1]: low result takes src1's HIGH half
          asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(pb), "v"(pa));
          want_x = 3.f + a1; want_y = 5.f + a1;
        } else if (mode == 11) {// v_pk_fma_f32 op_sel:[1,0,0]
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(pa), "v"(pb), "v"(pc));
          want_x = a1 * 3.f + 7.f; want_y = a1 * 5.f + 11.f;
        } else if (mode == 13) {// op_sel_hi:[0,1,1]: high result takes src0's LOW half
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(pa), "v"(pb), "v"(pc));
          want_x = a0 * 3.f + 7.f; want_y = a0 * 5.f + 11.f;
        } else if (mode == 14) {// op_sel_hi:[1,1,0]: high result takes src2's LOW half
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(pb), "v"(pc), "v"(pa));
          want_x = 3.f * 7.f + a0; want_y = 5.f * 11.f + a0;
        } else if (mode == 15) {// op_sel_hi:[1,0,0]
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pb), "v"(pa), "v"(pc));
          want_x = 3.f * a0 + 7.f; want_y = 5.f * a0 + 7.f;
        } else if (mode == 16) {// v_pk_mul_f32 op_sel_hi:[0,1]
          asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(pa), "v"(pb));
          want_x = a0 * 3.f; want_y = a0 * 5.f;
        } else if (mode == 17) {// v_pk_mul_f32 op_sel_hi:[1,0]
          asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(pb), "v"(pa));
          want_x = 3.f * a0; want_y = 5.f * a0;
        } else if (mode == 18) {// op_sel:[0,1,0] together with op_sel_hi:[1,0,1] (src1 swapped: low result from hi, high from lo)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(pb), "v"(pa), "v"(pc));
          want_x = 3.f * a1 + 7.f; want_y = 5.f * a0 + 11.f;
        } else {                // mode 12: v_pk_fma_f32 op_sel:[0,0,1]
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d) : "v"(pb), "v"(pc), "v"(pa));
          want_x = 3.f * 7.f + a1; want_y = 5.f * 11.f + a1;
        }
        r = d;
      } else {                  // mode 7: the same arithmetic with two scalar v_fma_f32
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r.x) : "v"(pb.x), "v"(pa.x));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r.y) : "v"(pb.y), "v"(pa.y));
        }
        want_x = 1.f + 8.f * 3.f * a0; want_y = 2.f + 8.f * 5.f * a1;
      }
      if (r.x != want_x || r.y != want_y) ++wrong;
      continue;
    }
    if (mode >= 2) {      // mode 0/1 result discarded; other instruction mixes:
      acc = probe_f2{0.f, 0.f};
      if (mode == 2) {    // as mode 1, but 16 wait states between the wait and the first use
        asm volatile(
            "ds_read2_b32 %1, %9 offset0:0 offset1:1\n\tds_read2_b32 %2, %9 offset0:2 offset1:3\n\t"
            "ds_read2_b32 %3, %9 offset0:4 offset1:5\n\tds_read2_b32 %4, %9 offset0:6 offset1:7\n\t"
            "ds_read2_b32 %5, %9 offset0:8 offset1:9\n\tds_read2_b32 %6, %9 offset0:10 offset1:11\n\t"
            "ds_read2_b32 %7, %9 offset0:12 offset1:13\n\tds_read2_b32 %8, %9 offset0:14 offset1:15\n\t"
            "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_pk_fma_f32 %0, %10, %1, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %10, %2, %0 op_sel:[0,1,0]\n\t"
            "v_pk_fma_f32 %0, %10, %3, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %10, %4, %0 op_sel:[0,1,0]\n\t"
            "v_pk_fma_f32 %0, %10, %5, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %10, %6, %0 op_sel:[0,1,0]\n\t"
            "v_pk_fma_f32 %0, %10, %7, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %10, %8, %0 op_sel:[0,1,0]"
            : "+v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
            : "v"(addr), "v"(w)
            : "memory");
      } else if (mode == 3) {   // ds_read2_b32, one lgkmcnt(0), plain v_add_f32 of the second dwords (no packed instruction)
        float sx = 0.f;
        asm volatile(
            "ds_read2_b32 %1, %9 offset0:0 offset1:1\n\tds_read2_b32 %2, %9 offset0:2 offset1:3\n\t"
            "ds_read2_b32 %3, %9 offset0:4 offset1:5\n\tds_read2_b32 %4, %9 offset0:6 offset1:7\n\t"
            "ds_read2_b32 %5, %9 offset0:8 offset1:9\n\tds_read2_b32 %6, %9 offset0:10 offset1:11\n\t"
            "ds_read2_b32 %7, %9 offset0:12 offset1:13\n\tds_read2_b32 %8, %9 offset0:14 offset1:15\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "+v"(sx), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
            : "v"(addr)
            : "memory");
        sx = ((t0.y + t1.y) + (t2.y + t3.y)) + ((t4.y + t5.y) + (t6.y + t7.y));
        acc = probe_f2{sx, sx};
      } else {                  // mode 4: sixteen ds_read_b32 (no read2), one lgkmcnt(0), packed FMAs on explicit pairs
        float u[16];
        asm volatile(
            "ds_read_b32 %0, %16\n\tds_read_b32 %1, %16 offset:4\n\tds_read_b32 %2, %16 offset:8\n\tds_read_b32 %3, %16 offset:12\n\t"
            "ds_read_b32 %4, %16 offset:16\n\tds_read_b32 %5, %16 offset:20\n\tds_read_b32 %6, %16 offset:24\n\tds_read_b32 %7, %16 offset:28\n\t"
            "ds_read_b32 %8, %16 offset:32\n\tds_read_b32 %9, %16 offset:36\n\tds_read_b32 %10, %16 offset:40\n\tds_read_b32 %11, %16 offset:44\n\t"
            "ds_read_b32 %12, %16 offset:48\n\tds_read_b32 %13, %16 offset:52\n\tds_read_b32 %14, %16 offset:56\n\tds_read_b32 %15, %16 offset:60\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]), "=&v"(u[8]),
              "=&v"(u[9]), "=&v"(u[10]), "=&v"(u[11]), "=&v"(u[12]), "=&v"(u[13]), "=&v"(u[14]), "=&v"(u[15])
            : "v"(addr)
            : "memory");
        float sx = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sx += u[2 * k + 1];
        acc = probe_f2{sx, sx};
      }
    }
    // op_sel:[0,1,0]: low result += w.x * pair.y (second dword), high result += w.y * pair.y
    float want = 0.f;
    for (int k = 0; k < 8; ++k) want += (float)(((base + 2 * k + 1) * 7 + 3) % 61);
    if (acc.x != want || acc.y != want) ++wrong;
  }
  if (wrong) atomicAdd(bad, wrong);
}
}  // namespace dic

extern "C" {
using namespace dic;
int dic_debug_lds_wait_probe(unsigned* bad, int blocks, int iters, int mode, void* stream) {
  lds_wait_probe_kernel<<<blocks, 256, 4096 * sizeof(float), (hipStream_t)stream>>>(bad, iters, mode);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // extern "C"
