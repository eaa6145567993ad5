// Packed fp32 forms of the F(4x4, 3x3) / F(3x3, 4x4) transforms shared by wino4.hip and wino4_wgrad.hip.
#pragma once
#include "common.hpp"

namespace crdr {
namespace {

// Interpolation points of the F(4x4, 3x3) / F(3x3, 4x4) kernels: 0, +-a, +-b, infinity with a = 3/4, b = 5/4 (round 5; rounds 3-4: Lavin &
// Gray's 0, +-1, +-2).  The point set decides how far the fp32 result strays from the exact one: the products are accumulated over K in the
// transform domain and the output transform then cancels large terms, so the accumulation's rounding error comes out amplified -- by ~15 with
// +-1, +-2 (measured 1.2e-5 .. 2.3e-5 of the output scale on the 5x5 layers, 5x the direct kernels).  A kernel-faithful fp32 model of the
// arithmetic in numpy (tests/test_winograd_identities.py::test_point_set_error_model) over all dyadic pairs (a, b) puts a broad optimum at
// a ~ 0.6 .. 0.8, b ~ 1.25 .. 1.6; (3/4, 5/4): 4.5e-6 at K = 1 024 where (1, 2) gives 2.0e-5, weight gradients 3x lower as well.  Symmetric
// points keep the even / odd structure of the transforms -- the SAME instruction count -- and dyadic ones make every constant below exact
// in fp32 (a^2 - b^2 = -1 is a bonus).  Unnormalised Toom-Cook rows (the filter side carries the 1 / N_j):
//   B^T = [[a2b2, 0, -s2, 0, 1, 0], [0, -a b2, -b2, a, 1, 0], [0, a b2, -b2, -a, 1, 0], [0, -a2 b, -a2, b, 1, 0], [0, a2 b, -a2, -b, 1, 0],
//          [0, a2b2, 0, -s2, 0, 1]],   a2 = a^2, b2 = b^2, a2b2 = a^2 b^2, s2 = a^2 + b^2
//   rows +-a = E_a +- a O_a with E_a = d4 - b2 d2, O_a = d3 - b2 d1; rows +-b likewise with a2; 12 fma per 1-D transform.
constexpr float kWa = 0.75f, kWb = 1.25f, kWa2 = 0.5625f, kWb2 = 1.5625f, kWa2b2 = 0.87890625f, kWs2 = 2.125f, kWa3 = 0.421875f, kWb3 = 1.953125f;
constexpr double kWNa = 2.0 * 0.5625 * (0.5625 - 1.5625), kWNb = 2.0 * 1.5625 * (1.5625 - 0.5625), kWN0 = 0.87890625;   // N_j = prod_{l != j} (p_j - p_l)

// In PACKED fp32 (v_pk_fma_f32 / v_pk_add_f32: two lanes of arithmetic per instruction).  What the issue probe
// (tools/experiments/issue_probe_gen.py, profiles/r5_issue_probe.txt) says about this part: a wave's own vector instructions do NOT overlap
// its exact-fp32 MFMAs -- every v_fma_f32 beside v_mfma_f32_16x16x4_f32 costs ~5 cycles of matrix time plus ~7 per gap that holds any (72
// gaps with two each: +1 217 cycles on 2 304; the same 144 in 9 clusters: +772), a v_pk_fma_f32 costs the same ~5.7 as a scalar one (72 in 9
// clusters: +412), while LDS reads and LDS-DMA pieces issued DIRECTLY behind an MFMA are free (behind a VALU instruction they wait for it:
// +1 050 per sub-step).  So: the transform is written in packed form (72 instead of 144 instructions per sub-step), in 10 clusters, and
// every memory instruction sits right behind an MFMA.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v pk_fma(f32x2v a, f32x2v b, f32x2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2v pk_bc(float v) { return f32x2v{v, v}; }
// The clusters are single asm statements: left to itself the compiler turns about a third of the packed operations back into scalar pairs
// (measured: 41 v_pk_fma_f32 + 38 v_fma_f32 where 60 packed ones were written).  Inside a statement no packed result is read by the very next
// instruction (gfx950 wants one wait state there -- the compiler inserts it for its own code and, conservatively, behind an asm statement
// whose outputs the next VALU instruction reads).  Constant pairs live in scalar registers (one scalar source per instruction; a pair's
// low / high half is broadcast to both lanes with op_sel_hi:[.,0,.] / op_sel:[.,1,.]).
struct W4Consts { f32x2v k1, k2, k3, k4, k5; };   // (-b2, -a2), (a, b), (-a, -b), (a2b2, -s2), (a2, b2)
__device__ __forceinline__ W4Consts w4_consts() {
  return W4Consts{f32x2v{-kWb2, -kWa2}, f32x2v{kWa, kWb}, f32x2v{-kWa, -kWb}, f32x2v{kWa2b2, -kWs2}, f32x2v{kWa2, kWb2}};
}
// vertical pass, two patch columns at once: D[i] = (d[i][b], d[i][b + 1]) -> T[xi] = (t[xi][b], t[xi][b + 1]): 12 packed operations
__device__ __forceinline__ void bt6_cols(const f32x2v (&D)[6], f32x2v (&T)[6], const W4Consts& kc) {
  f32x2v ea, oa, eb, ob, x, y;
  asm volatile(
      "v_pk_fma_f32 %6, %14, %18, %16 op_sel_hi:[1,0,1]\n"                    // E_a = d4 - b2 d2
      "v_pk_fma_f32 %7, %13, %18, %15 op_sel_hi:[1,0,1]\n"                    // O_a = d3 - b2 d1
      "v_pk_fma_f32 %8, %14, %18, %16 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"     // E_b = d4 - a2 d2
      "v_pk_fma_f32 %9, %13, %18, %15 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"     // O_b = d3 - a2 d1
      "v_pk_fma_f32 %10, %14, %21, %16 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"    // x = d4 - s2 d2
      "v_pk_fma_f32 %11, %15, %21, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"    // y = d5 - s2 d3
      "v_pk_fma_f32 %1, %7, %19, %6 op_sel_hi:[1,0,1]\n"                      // t1 = E_a + a O_a
      "v_pk_fma_f32 %2, %7, %20, %6 op_sel_hi:[1,0,1]\n"                      // t2 = E_a - a O_a
      "v_pk_fma_f32 %3, %9, %19, %8 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"       // t3 = E_b + b O_b
      "v_pk_fma_f32 %4, %9, %20, %8 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"       // t4 = E_b - b O_b
      "v_pk_fma_f32 %0, %12, %21, %10 op_sel_hi:[1,0,1]\n"                    // t0 = a2b2 d0 + x
      "v_pk_fma_f32 %5, %13, %21, %11 op_sel_hi:[1,0,1]"                       // t5 = a2b2 d1 + y
      : "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(ea), "=&v"(oa), "=&v"(eb), "=&v"(ob), "=&v"(x), "=&v"(y)
      : "v"(D[0]), "v"(D[1]), "v"(D[2]), "v"(D[3]), "v"(D[4]), "v"(D[5]), "s"(kc.k1), "s"(kc.k2), "s"(kc.k3), "s"(kc.k4));
}
// horizontal pass of one row, inside the vector: P0 = (t0, t1), P1 = (t2, t3), P2 = (t4, t5) -> V[0] = (v0, v5), V[1] = (v1, v3), V[2] = (v2, v4):
// 6 packed operations (the half selects travel in the instructions' op_sel bits)
__device__ __forceinline__ void bt6_row(const f32x2v P0, const f32x2v P1, const f32x2v P2, f32x2v (&V)[3], const W4Consts& kc) {
  f32x2v e, o, xy;
  asm volatile(
      "v_pk_fma_f32 %3, %7, %9, %8 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n"     // (E_a, E_b) = t2 (-b2, -a2) + t4
      "v_pk_fma_f32 %4, %6, %9, %7 op_sel:[1,0,1] op_sel_hi:[1,1,1]\n"     // (O_a, O_b) = t1 (-b2, -a2) + t3
      "v_pk_fma_f32 %5, %7, %12, %8 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"    // (x, y) = (t4, t5) - s2 (t2, t3)
      "v_pk_fma_f32 %1, %4, %10, %3\n"                                      // (v1, v3) = (E_a + a O_a, E_b + b O_b)
      "v_pk_fma_f32 %2, %4, %11, %3\n"                                      // (v2, v4) = (E_a - a O_a, E_b - b O_b)
      "v_pk_fma_f32 %0, %6, %12, %5 op_sel_hi:[1,0,1]"                      // (v0, v5) = a2b2 (t0, t1) + (x, y)
      : "=&v"(V[0]), "=&v"(V[1]), "=&v"(V[2]), "=&v"(e), "=&v"(o), "=&v"(xy)
      : "v"(P0), "v"(P1), "v"(P2), "s"(kc.k1), "s"(kc.k2), "s"(kc.k3), "s"(kc.k4));
}
// element y of a transformed row kept as V[0] = (v0, v5), V[1] = (v1, v3), V[2] = (v2, v4)
__device__ __forceinline__ float v_elem(const f32x2v (&V)[3], int y) { return y == 0 ? V[0].x : y == 1 ? V[1].x : y == 2 ? V[2].x : y == 3 ? V[1].y : y == 4 ? V[2].y : V[0].y; }

}  // namespace
}  // namespace crdr
