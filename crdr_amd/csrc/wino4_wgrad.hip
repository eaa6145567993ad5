// Weight gradient of 3x3 stride-1 convolutions by Winograd minimal filtering F(3x3, 4x4) on the exact-fp32 matrix cores -- the transposition of
// wino4.hip's F(4x4, 3x3), as wino_wgrad.hip's F(3x3, 2x2) is the transposition of wino.hip's F(2x2, 3x3):
//
//   g[i][j][r][s] = sum_{n, a, b} P[n, a, b, i] * Q[n, a - pad + r, b - pad + s, j]          (wgrad.hip's statement, stride 1)
//
// Cut P (dy of a Conv2d) into 4 x 4 tiles and Q (x) into the 6 x 6 patches they meet: per tile the 3 x 3 taps are the correlation of the patch
// with the tile, which minimal filtering does in 36 products instead of 144:
//   g = sum_tiles A^T [ (G p G^T) . (B^T q B) ] A,   A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,1]],   B^T = wino4.hip's,
//   G = D V,  V = rows [1, x, x^2, x^3] at x = 0, 1, -1, 2, -2 and [0,0,0,1],  D = diag(1/4, -1/6, -1/6, 1/24, 1/24, 1)
// i.e. 36 GEMMs [I x tiles] . [tiles x J] whose reduction axis is the tile index, and ONE output transform per workgroup.  D is left out of
// the K loop: the accumulators carry (V p V^T) . (B^T q B) and the output transform scales position (xi, nu) by D[xi] D[nu] once.
//
// Workgroup = 4 waves (one per SIMD, 512 registers) = 64 channels of P x 32 channels of Q x all 36 positions, over a contiguous range of
// strips (a strip = 4 consecutive tiles of one tile row = 4 x 16 pixels of P and the 6 x 18 pixels of Q around them = ONE k-step of
// v_mfma_f32_16x16x4_f32: lane group kg holds tile kg).  Wave (iw, jw): P channels 32 iw .. + 31 (two blocks of 16), Q channels 16 jw .. + 15:
// 72 accumulator blocks (position x P block).  A strip is one LDS stage ([pixel][channels], exactly the NHWC memory layout, LDS-DMA, double
// buffered); a lane reads its channels' 2 x 16 + 36 raw values with ds_read_b32, transforms them in registers (V: 9 operations per 1-D
// transform, B^T: 12) and feeds them as the A (P side) and B (Q side) operands of 72 MFMAs.  Epilogue: pure register arithmetic per lane
// (it holds all 36 positions of its (i, j) pairs), written into the slab of this split -- the layout of wgrad_kernel's slabs
// ([split][tap][PC][QC]), so the deferred fixed-order reduce (wgrad_reduce_batched) and everything behind it are unchanged.
//
// The 5x5 layers run through the same loop as four 3x3 sub-problems (blockIdx.x carries the sub-problem), each writing its share of the 25 taps:
//   * 5x5 stride 1 (the slice transforms of the context model): taps (3 bi + a, 3 bj + b) -- sub-filter (bi, bj) correlates P with the patch of Q
//     displaced by (3 bi, 3 bj);
//   * 5x5 stride 2 (elic_autoencoder.py:42-52 / elic_layers.py:14-21: Q has twice P's resolution): taps (2 a + ph, 2 b + pw) -- sub-filter
//     (ph, pw) correlates P with the parity plane Q[2 y + ph][2 x + pw] at pad 1: the Q strip is read with pixel stride 2.
// In both, Q row u / column c of a strip is image pixel qs (4 tr + u) + oy / qs (16 sc + c) + ox with (qs, oy, ox) per sub-problem.
#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "wgrad_args.hpp"

namespace crdr {

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned kOob = 0x80000000u;
constexpr int kPC = 64, kQC = 32;                          // channels of P / Q per workgroup
constexpr int kPFloats = 4 * 16 * kPC, kQFloats = 6 * 18 * kQC;
constexpr int kUsed = kPFloats + kQFloats;                 // floats of a strip (7 552)
constexpr int kPieces = kUsed / 4;                         // 16-byte pieces (1 888)
constexpr int kPasses = (kPieces + 255) / 256;             // 8 (the last one partial: its wave 1 is half used)
constexpr int kStage = ((kPieces + 63) / 64) * 64 * 4;     // floats per stage, whole wave instructions (7 680)

// V p: rows [1, x, x^2, x^3] at x = 0, 1, -1, 2, -2, inf
__device__ __forceinline__ void v4(const float p0, const float p1, const float p2, const float p3, float (&o)[6]) {
  const float e = p0 + p2, d = p1 + p3;
  const float e2 = __builtin_fmaf(4.0f, p2, p0), d2 = __builtin_fmaf(4.0f, p3, p1);
  o[0] = p0;
  o[1] = e + d;
  o[2] = e - d;
  o[3] = __builtin_fmaf(2.0f, d2, e2);
  o[4] = __builtin_fmaf(-2.0f, d2, e2);
  o[5] = p3;
}
// B^T d (wino4.hip's data transform)
__device__ __forceinline__ void bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5, float (&o)[6]) {
  const float a = __builtin_fmaf(-4.0f, d2, d4), b = __builtin_fmaf(-4.0f, d1, d3);
  o[0] = __builtin_fmaf(4.0f, d0, __builtin_fmaf(-5.0f, d2, d4));
  o[1] = a + b;
  o[2] = a - b;
  const float c = d4 - d2, e = d3 - d1;
  o[3] = __builtin_fmaf(2.0f, e, c);
  o[4] = __builtin_fmaf(-2.0f, e, c);
  o[5] = __builtin_fmaf(4.0f, d1, __builtin_fmaf(-5.0f, d3, d5));
}

__device__ __forceinline__ float acc_read(float v) {
  float o;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(o) : "a"(v));
  return o;
}

__global__ __launch_bounds__(256) void wino4_wgrad_kernel(const WgradArgs p_, const WgradGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int iw = wave & 1, jw = wave >> 1;
  WgradArgs p = p_;
  const int gidx = blockIdx.z;
  if (p.ngroup > 1) { p.p = grp.p[gidx]; p.q = grp.q[gidx]; }
  // sub-problem (one for 3x3; four for 5x5): Q pixel stride, origin of the strip's Q patch, tap of sub-filter element (0, 0) and tap steps
  const int nij = ((p.PC + kPC - 1) / kPC) * p.jtiles;
  const int sub = blockIdx.x / nij, bxy = blockIdx.x - sub * nij;
  const int sa = sub >> 1, sb = sub & 1;
  int qs = 1, oy = -p.pad, ox = -p.pad, tap0 = 0, tstep = 1, na = 3, nb = 3;   // taps (a, b) -> tap0 + tstep (a kw + b), a < na, b < nb
  if (p.kw == 5 && p.stride == 1) { oy += 3 * sa; ox += 3 * sb; tap0 = (3 * sa) * 5 + 3 * sb; na = sa ? 2 : 3; nb = sb ? 2 : 3; }
  if (p.kw == 5 && p.stride == 2) { qs = 2; oy = sa - 2; ox = sb - 2; tap0 = sa * 5 + sb; tstep = 2; na = sa ? 2 : 3; nb = sb ? 2 : 3; }
  const int it = bxy / p.jtiles, jt = bxy - it * p.jtiles;
  const int i0 = it * kPC, j0 = jt * kQC;
  const int split = blockIdx.y;
  const int TR = (p.PH + 3) >> 2, SC = (p.PW + 15) >> 4;
  const long long S = (long long)p.N * TR * SC;
  const int s0 = (int)(S * split / p.nsplit), s1 = (int)(S * (split + 1) / p.nsplit);
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.p), 0, p.p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.q), 0, p.q_bytes, 0x00020000);

  // staging: piece S = tid + 256 pass -> P: [row 4][col 16][16 pieces], Q: [row 6][col 18][8 pieces]
  int s_row[kPasses], s_col[kPasses];
  bool s_isp[kPasses];
  unsigned s_off[kPasses];
#pragma unroll
  for (int j = 0; j < kPasses; ++j) {
    const int Sx = tid + 256 * j;
    const bool isp = Sx < kPFloats / 4;
    const int px = isp ? Sx >> 4 : (Sx - kPFloats / 4) >> 3;
    const int ch = isp ? (Sx & 15) * 4 : ((Sx - kPFloats / 4) & 7) * 4;
    s_isp[j] = isp;
    s_row[j] = isp ? px >> 4 : px / 18;
    s_col[j] = isp ? px & 15 : px - (px / 18) * 18;
    // lane part of the byte offsets (strip (n, tr, sc) = (0, 0, 0)); the strip's displacement is uniform
    if (isp) s_off[j] = i0 + ch < p.PC ? (unsigned)(((s_row[j] * p.PW + s_col[j]) * p.ldp + i0 + ch) * 4) : kOob;
    else s_off[j] = j0 + ch < p.QC ? (unsigned)((((qs * s_row[j] + oy) * p.QW + qs * s_col[j] + ox) * p.ldq + j0 + ch) * 4) : kOob;   // (may wrap: added to the strip's displacement)
    if (Sx >= kPieces) s_off[j] = kOob;
  }
  int f_n, f_tr, f_sc;   // strip counters of the NEXT strip to fetch: no division in the loop
  {
    const int n = s0 / (TR * SC), rem = s0 - n * (TR * SC);
    f_n = n; f_tr = rem / SC; f_sc = rem - f_tr * SC;
  }
  auto fetch = [&](int buf) __attribute__((always_inline)) {
    const int n = f_n, tr = f_tr, sc = f_sc;
    if (++f_sc == SC) { f_sc = 0; if (++f_tr == TR) { f_tr = 0; ++f_n; } }
    float* st = smem + buf * kStage;
    const unsigned dp = (unsigned)((((n * p.PH + 4 * tr) * p.PW + 16 * sc) * p.ldp) * 4);
    const unsigned dq = (unsigned)((((n * p.QH + qs * 4 * tr) * p.QW + qs * 16 * sc) * p.ldq) * 4);
    // interior strip: its 4 x 16 pixels of P and the 6 x 18 pixels of Q around them all lie inside the images
    const bool inner = 4 * tr + 4 <= p.PH && 16 * sc + 16 <= p.PW && qs * 4 * tr + oy >= 0 && qs * (4 * tr + 5) + oy < p.QH &&
                       qs * 16 * sc + ox >= 0 && qs * (16 * sc + 17) + ox < p.QW;
#pragma unroll
    for (int j = 0; j < kPasses; ++j) {
      if (j * 256 + wave * 64 >= kPieces) break;   // (wave-uniform)
      unsigned off = s_off[j];
      if (!inner) {
        if (s_isp[j]) {
          const int a = 4 * tr + s_row[j], b = 16 * sc + s_col[j];
          if (!(a < p.PH && b < p.PW)) off = kOob;
        } else {
          const int a = qs * (4 * tr + s_row[j]) + oy, b = qs * (16 * sc + s_col[j]) + ox;
          if (!((unsigned)a < (unsigned)p.QH && (unsigned)b < (unsigned)p.QW)) off = kOob;
        }
      }
      if (s_isp[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr_t)(st + (j * 256 + wave * 64) * 4), 16, (int)off, (int)dp, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)(st + (j * 256 + wave * 64) * 4), 16, (int)(off == kOob ? kOob : off + dq), 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x4 acc[64], accv[8];
#pragma unroll
  for (int j = 0; j < 64; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) accv[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int m = lane & 15, kg = lane >> 4;
  const int pb = (4 * kg) * kPC + 32 * iw + m, qb = kPFloats + (4 * kg) * kQC + 16 * jw + m;   // this lane's tile / channel inside the P / Q images
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float* st = smem + buf * kStage;
    float A[2][6][6], B[6][6];
    // P side: V p V^T of the lane's tile for its two channel blocks
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      float t[6][4];   // vertical pass: [xi][column]
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float o[6];
        v4(st[pb + (0 * 16 + v) * kPC + 16 * ib], st[pb + (1 * 16 + v) * kPC + 16 * ib], st[pb + (2 * 16 + v) * kPC + 16 * ib],
           st[pb + (3 * 16 + v) * kPC + 16 * ib], o);
#pragma unroll
        for (int x = 0; x < 6; ++x) t[x][v] = o[x];
      }
#pragma unroll
      for (int x = 0; x < 6; ++x) v4(t[x][0], t[x][1], t[x][2], t[x][3], A[ib][x]);
    }
    // Q side: B^T q B of the lane's patch
    {
      float t[6][6];
#pragma unroll
      for (int v = 0; v < 6; ++v) {
        float o[6];
        bt6(st[qb + (0 * 18 + v) * kQC], st[qb + (1 * 18 + v) * kQC], st[qb + (2 * 18 + v) * kQC], st[qb + (3 * 18 + v) * kQC],
            st[qb + (4 * 18 + v) * kQC], st[qb + (5 * 18 + v) * kQC], o);
#pragma unroll
        for (int x = 0; x < 6; ++x) t[x][v] = o[x];
      }
#pragma unroll
      for (int x = 0; x < 6; ++x) bt6(t[x][0], t[x][1], t[x][2], t[x][3], t[x][4], t[x][5], B[x]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 72; ++s) {
      const int pos = s >> 1, ib = s & 1, x = pos / 6, y = pos - 6 * x;
      if (s < 64) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ib][x][y], B[x][y], acc[s], 0, 0, 0);
      else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accv[s - 64]) : "v"(A[ib][x][y]), "v"(B[x][y]));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  if (s0 < s1) fetch(0);
  __syncthreads();
  for (int s = s0; s < s1; ++s) {
    const int buf = (s - s0) & 1;
    if (s + 1 < s1) fetch(buf ^ 1);
    compute(buf);
    __syncthreads();
  }

  // ---- output transform g = A^T (D U D) A per (i, j) pair of the lane: accumulator element r of block (pos, ib) is
  // (i = i0 + 32 iw + 16 ib + 4 kg + r, j = j0 + 16 jw + m)
  float* slab = p.ws + (size_t)gidx * p.slab_elems + (size_t)split * p.T * p.PC * p.QC;
  const int jc = j0 + 16 * jw + m;
  const float D[6] = {0.25f, -1.0f / 6, -1.0f / 6, 1.0f / 24, 1.0f / 24, 1.0f};
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv[3][6];   // A^T (D U D), rows a, columns nu
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) {
        float u[6];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
          const int blk = 2 * (xi * 6 + nu) + ib;
          const float raw = blk < 64 ? acc_read(acc[blk < 64 ? blk : 0][r]) : accv[blk >= 64 ? blk - 64 : 0][r];
          u[xi] = raw * (D[xi] * D[nu]);
        }
        const float pp = u[1] + u[2], qq = u[1] - u[2], uu = u[3] + u[4], ww = u[3] - u[4];
        sv[0][nu] = u[0] + pp + uu;
        sv[1][nu] = __builtin_fmaf(2.0f, ww, qq);
        sv[2][nu] = __builtin_fmaf(4.0f, uu, pp) + u[5];
      }
      const int i = i0 + 32 * iw + 16 * ib + 4 * kg + r;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float pp = sv[a][1] + sv[a][2], qq = sv[a][1] - sv[a][2], uu = sv[a][3] + sv[a][4], ww = sv[a][3] - sv[a][4];
        const float g0 = sv[a][0] + pp + uu, g1 = __builtin_fmaf(2.0f, ww, qq), g2 = __builtin_fmaf(4.0f, uu, pp) + sv[a][5];
        if (i < p.PC && jc < p.QC && a < na) {
          const int t0 = tap0 + tstep * (a * p.kw);
          slab[((size_t)t0 * p.PC + i) * p.QC + jc] = g0;
          slab[((size_t)(t0 + tstep) * p.PC + i) * p.QC + jc] = g1;
          if (nb > 2) slab[((size_t)(t0 + 2 * tstep) * p.PC + i) * p.QC + jc] = g2;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

void wino4_wgrad_launch(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t s) {
  static std::atomic<bool> attr_done{false};
  if (!attr_done.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(wino4_wgrad_kernel, grid, dim3(256), (size_t)2 * kStage * sizeof(float), s, a, grp);
}

}  // namespace crdr
