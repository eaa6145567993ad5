// Packed fp32 forms of the F(4x4, 3x3) / F(3x3, 4x4) transforms shared by wino4.hip and wino4_wgrad.hip.
#pragma once
#include "common.hpp"

namespace crdr {
namespace {

// 1-D data transform B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]] in PACKED fp32
// (v_pk_fma_f32 / v_pk_add_f32: two lanes of arithmetic per instruction).  What the issue probe (tools/experiments/issue_probe_gen.py,
// profiles/r5_issue_probe.txt) says about this part: a wave's own vector instructions do NOT overlap its exact-fp32 MFMAs -- every
// v_fma_f32 beside v_mfma_f32_16x16x4_f32 costs ~5 cycles of matrix time plus ~7 per gap that holds any (72 gaps with two each: +1 217
// cycles on 2 304; the same 144 in 9 clusters: +772), a v_pk_fma_f32 costs the same ~5.7 as a scalar one (72 in 9 clusters: +412), while LDS
// reads and LDS-DMA pieces issued DIRECTLY behind an MFMA are free (behind a VALU instruction they wait for it: +1 050 per sub-step).  So:
// the transform is written in packed form (72 instead of 144 instructions per sub-step), in 10 clusters, and every memory instruction sits
// right behind an MFMA.  Same operations in the same association as the scalar form (every step an fma or an exact add): bit-identical.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v pk_fma(f32x2v a, f32x2v b, f32x2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2v pk_bc(float v) { return f32x2v{v, v}; }
// The clusters are single asm statements: left to itself the compiler turns about a third of the packed operations back into scalar pairs
// (measured: 41 v_pk_fma_f32 + 38 v_fma_f32 where 60 packed ones were written).  Inside a statement no packed result is read by the very next
// instruction (gfx950 wants one wait state there -- the compiler inserts it for its own code and, conservatively, behind an asm statement
// whose outputs the next VALU instruction reads).  Constant pairs live in scalar registers (one scalar source per instruction).
struct W4Consts { f32x2v k41, k12, kn12, k5; };   // (-4, -1), (1, 2), (-1, -2), (-5, -5)
// vertical pass, two patch columns at once: D[i] = (d[i][b], d[i][b + 1]) -> T[xi] = (t[xi][b], t[xi][b + 1]): 12 packed operations
__device__ __forceinline__ void bt6_cols(const f32x2v (&D)[6], f32x2v (&T)[6], const W4Consts& kc) {
  f32x2v a, b, x, c, e, y;
  asm volatile(
      "v_pk_fma_f32 %6, %14, -4.0, %16 op_sel_hi:[1,0,1]\n"     // a = d4 - 4 d2
      "v_pk_fma_f32 %7, %13, -4.0, %15 op_sel_hi:[1,0,1]\n"     // b = d3 - 4 d1
      "v_pk_fma_f32 %8, %14, %18, %16 op_sel_hi:[1,0,1]\n"      // x = d4 - 5 d2
      "v_pk_add_f32 %9, %16, %14 neg_lo:[0,1] neg_hi:[0,1]\n"   // c = d4 - d2
      "v_pk_add_f32 %10, %15, %13 neg_lo:[0,1] neg_hi:[0,1]\n"  // e = d3 - d1
      "v_pk_fma_f32 %11, %15, %18, %17 op_sel_hi:[1,0,1]\n"     // y = d5 - 5 d3
      "v_pk_add_f32 %1, %6, %7\n"                                // t1 = a + b
      "v_pk_add_f32 %2, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n"     // t2 = a - b
      "v_pk_fma_f32 %0, %12, 4.0, %8 op_sel_hi:[1,0,1]\n"       // t0 = 4 d0 + x
      "v_pk_fma_f32 %3, %10, 2.0, %9 op_sel_hi:[1,0,1]\n"       // t3 = c + 2 e
      "v_pk_fma_f32 %4, %10, -2.0, %9 op_sel_hi:[1,0,1]\n"      // t4 = c - 2 e
      "v_pk_fma_f32 %5, %13, 4.0, %11 op_sel_hi:[1,0,1]"         // t5 = 4 d1 + y
      : "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(a), "=&v"(b), "=&v"(x), "=&v"(c), "=&v"(e), "=&v"(y)
      : "v"(D[0]), "v"(D[1]), "v"(D[2]), "v"(D[3]), "v"(D[4]), "v"(D[5]), "s"(kc.k5));
}
// horizontal pass of one row, inside the vector: P0 = (t0, t1), P1 = (t2, t3), P2 = (t4, t5) -> V[0] = (v0, v5), V[1] = (v1, v3), V[2] = (v2, v4):
// 6 packed operations (the half selects travel in the instructions' op_sel bits)
__device__ __forceinline__ void bt6_row(const f32x2v P0, const f32x2v P1, const f32x2v P2, f32x2v (&V)[3], const W4Consts& kc) {
  f32x2v ac, be, xy;
  asm volatile(
      "v_pk_fma_f32 %3, %7, %9, %8 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n"   // (a, c) = t2 (-4, -1) + t4
      "v_pk_fma_f32 %4, %6, %9, %7 op_sel:[1,0,1] op_sel_hi:[1,1,1]\n"   // (b, e) = t1 (-4, -1) + t3
      "v_pk_fma_f32 %5, %7, %12, %8 op_sel_hi:[1,0,1]\n"                  // (x, y) = (t4, t5) - 5 (t2, t3)
      "v_pk_fma_f32 %1, %4, %10, %3\n"                                    // (v1, v3) = (a + b, c + 2 e)
      "v_pk_fma_f32 %2, %4, %11, %3\n"                                    // (v2, v4) = (a - b, c - 2 e)
      "v_pk_fma_f32 %0, %6, 4.0, %5 op_sel_hi:[1,0,1]"                    // (v0, v5) = 4 (t0, t1) + (x, y)
      : "=&v"(V[0]), "=&v"(V[1]), "=&v"(V[2]), "=&v"(ac), "=&v"(be), "=&v"(xy)
      : "v"(P0), "v"(P1), "v"(P2), "s"(kc.k41), "s"(kc.k12), "s"(kc.kn12), "s"(kc.k5));
}
// element y of a transformed row kept as V[0] = (v0, v5), V[1] = (v1, v3), V[2] = (v2, v4)
__device__ __forceinline__ float v_elem(const f32x2v (&V)[3], int y) { return y == 0 ? V[0].x : y == 1 ? V[1].x : y == 2 ? V[2].x : y == 3 ? V[1].y : y == 4 ? V[2].y : V[0].y; }

}  // namespace
}  // namespace crdr
