// Kernel-argument structs of the weight-gradient family, shared by wgrad.hip (direct slab kernels, host planning) and
// wino_wgrad.hip (Winograd slab kernel).
#pragma once

#include "common.hpp"

namespace crdr {

struct FastDiv {  // unsigned division by a runtime constant: n / d == umulhi(n, mul) >> sh   (n < 2^31)
  unsigned mul, sh, d;
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f; f.d = d;
  if (d == 1) { f.mul = 0; f.sh = 0; return f; }
  unsigned l = 0; while ((1u << l) < d) ++l;               // ceil(log2 d)
  const unsigned long long m = ((1ull << (32 + l)) + d - 1) / d;  // in (2^32, 2^33)
  f.mul = (unsigned)(m - (1ull << 32)); f.sh = l;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  if (f.d == 1) return n;
  const unsigned t = __umulhi(n, f.mul);
  return (t + ((n - t) >> 1)) >> (f.sh - 1);
}

struct WgradGroup {  // per-problem operands of a grouped launch; indexed with the workgroup-uniform problem index only
  const float* p[CRDR_MAX_GROUP];
  const float* q[CRDR_MAX_GROUP];
};

struct WgradArgs {
  const float* p;
  const float* q;
  float* ws;
  int ngroup;
  long long slab_elems;  // floats of one problem's slab
  int N, PH, PW, PC, ldp;
  int QH, QW, QC, ldq;
  int kw, stride, pad, T;
  int M;        // N*PH*PW
  int ntiles;   // pixel tiles of 32
  int nsplit;
  int jtiles;
  FastDiv d_hw, d_w;
  unsigned p_bytes, q_bytes;  // extents of the two buffer descriptors (range-checked loads)
  int smallj;  // 1: QC <= 4 (RGB operand): the taps are folded into the GEMM columns, column = 4 tap + channel, so one
               // launch covers all taps instead of one 32-column (>= 87 % padding) GEMM per tap
};


// Winograd F(3x3, 2x2) slab kernel (wino_wgrad.hip): same slab layout as wgrad_kernel, grid = (I tiles of 64 x J tiles of 64,
// strip splits, problems)
void wino_wgrad_launch(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t s);
// Winograd F(3x3, 4x4) slab kernel (wino4_wgrad.hip): the same, grid = (I tiles of 64 x J tiles of 32, strip splits, problems)
void wino4_wgrad_launch(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t s);

}  // namespace crdr
