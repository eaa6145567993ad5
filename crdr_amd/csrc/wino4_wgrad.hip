// Weight gradient of 3x3 stride-1 convolutions by Winograd minimal filtering F(3x3, 4x4) on the exact-fp32 matrix cores -- the transposition of
// wino4.hip's F(4x4, 3x3), as wino_wgrad.hip's F(3x3, 2x2) is the transposition of wino.hip's F(2x2, 3x3):
//
//   g[i][j][r][s] = sum_{n, a, b} P[n, a, b, i] * Q[n, a - pad + r, b - pad + s, j]          (wgrad.hip's statement, stride 1)
//
// Cut P (dy of a Conv2d) into 4 x 4 tiles and Q (x) into the 6 x 6 patches they meet: per tile the 3 x 3 taps are the correlation of the patch
// with the tile, which minimal filtering does in 36 products instead of 144:
//   g = sum_tiles A^T [ (G p G^T) . (B^T q B) ] A,   A^T = [[1,1,1,1,1,0],[0,a,-a,b,-b,0],[0,a2,a2,b2,b2,1]],   B^T = wino4.hip's,
//   G = D V,  V = rows [1, x, x^2, x^3] at x = 0, +-a, +-b and [0,0,0,1],  D = diag(1 / N_j)   (points: wino4_xform.hpp, a = 3/4, b = 5/4)
// i.e. 36 GEMMs [I x tiles] . [tiles x J] whose reduction axis is the tile index, and ONE output transform per workgroup.  D is left out of
// the K loop: the accumulators carry (V p V^T) . (B^T q B) and the output transform scales position (xi, nu) by D[xi] D[nu] once.
//
// Workgroup = 4 waves (one per SIMD, 512 registers) = 64 channels of P x 32 channels of Q x all 36 positions, over a contiguous range of
// strips (a strip = 4 consecutive tiles of one tile row = 4 x 16 pixels of P and the 6 x 18 pixels of Q around them = ONE k-step of
// v_mfma_f32_16x16x4_f32: lane group kg holds tile kg).  Wave (iw, jw): P channels 32 iw .. + 31 (two blocks of 16), Q channels 16 jw .. + 15:
// 72 accumulator blocks (position x P block).  A strip is one LDS stage ([pixel][channels], exactly the NHWC memory layout, LDS-DMA, double
// buffered); a lane reads its channels' 2 x 16 + 36 raw values as 34 pairs (ds_read2_b32: the two P channel blocks of a pixel, two
// neighbouring Q pixels), transforms them in registers in PACKED fp32 (V: 8 operations per 1-D transform, over the two channel blocks at
// once; B^T: wino4_xform.hpp) and feeds them as the A (P side) and B (Q side) operands of 72 MFMAs.
// Round 5 (the issue probe, tools/experiments/issue_probe_gen.py: a wave's vector instructions never hide behind its own fp32 MFMAs, its LDS
// reads and DMA requests do when they sit directly behind one): a k-step is [transforms of strip s: 152 packed instructions] [MFMAs 0..35]
// [vmcnt(0) + barrier: strip s + 1 has landed, everybody is done reading strip s] [MFMAs 36..71, each followed by one of: the 8 DMA
// requests of strip s + 2, the 34 raw reads of strip s + 1].  Before: 324 scalar transform instructions + 68 reads in front of 72 bare MFMAs
// (matrix pipe 37 % busy).  Epilogue: pure register arithmetic per lane
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
#include <type_traits>

#include "common.hpp"
#include "wgrad_args.hpp"
#include "wino4_xform.hpp"

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

// V p: rows [1, x, x^2, x^3] at x = 0, +-a, +-b, inf (wino4_xform.hpp: a = 3/4, b = 5/4), on two channel blocks at once (packed fp32; o[0] = p0
// and o[5] = p3 are the inputs): rows +-a = E_a +- a D_a with E_a = p0 + a2 p2, D_a = p1 + a2 p3
__device__ __forceinline__ void v4_pk(const f32x2v p0, const f32x2v p1, const f32x2v p2, const f32x2v p3, f32x2v (&o)[6], const W4Consts& kc) {
  f32x2v ea, da, eb, db;
  asm volatile(
      "v_pk_fma_f32 %4, %10, %14, %8 op_sel_hi:[1,0,1]\n"                     // E_a = p0 + a2 p2
      "v_pk_fma_f32 %5, %11, %14, %9 op_sel_hi:[1,0,1]\n"                     // D_a = p1 + a2 p3
      "v_pk_fma_f32 %6, %10, %14, %8 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"      // E_b = p0 + b2 p2
      "v_pk_fma_f32 %7, %11, %14, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"      // D_b = p1 + b2 p3
      "v_pk_fma_f32 %0, %5, %12, %4 op_sel_hi:[1,0,1]\n"                      // o1 = E_a + a D_a
      "v_pk_fma_f32 %1, %5, %13, %4 op_sel_hi:[1,0,1]\n"                      // o2 = E_a - a D_a
      "v_pk_fma_f32 %2, %7, %12, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"       // o3 = E_b + b D_b
      "v_pk_fma_f32 %3, %7, %13, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]"          // o4 = E_b - b D_b
      : "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(ea), "=&v"(da), "=&v"(eb), "=&v"(db)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(kc.k2), "s"(kc.k3), "s"(kc.k5));
  o[0] = p0;
  o[5] = p3;
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
  static_assert(kPFloats / 4 == 4 * 256, "staging passes 0..3 are P pieces, 4..7 Q pieces");
  unsigned s_off[kPasses];
#pragma unroll
  for (int j = 0; j < kPasses; ++j) {
    const int Sx = tid + 256 * j;
    const bool isp = Sx < kPFloats / 4;
    const int px = isp ? Sx >> 4 : (Sx - kPFloats / 4) >> 3;
    const int ch = isp ? (Sx & 15) * 4 : ((Sx - kPFloats / 4) & 7) * 4;
    s_row[j] = isp ? px >> 4 : px / 18;
    s_col[j] = isp ? px & 15 : px - (px / 18) * 18;
    // lane part of the byte offsets (strip (n, tr, sc) = (0, 0, 0)); the strip's displacement is uniform
    if (isp) s_off[j] = i0 + ch < p.PC ? (unsigned)(((s_row[j] * p.PW + s_col[j]) * p.ldp + i0 + ch) * 4) : kOob;
    // (Q: relative to the patch's first pixel -- never negative; the patch origin (oy, ox) travels with the strip's scalar displacement)
    else s_off[j] = j0 + ch < p.QC ? (unsigned)(((qs * s_row[j] * p.QW + qs * s_col[j]) * p.ldq + j0 + ch) * 4) : kOob;
    if (Sx >= kPieces) s_off[j] = kOob;
  }
  // the last pass is partial (pieces 1 792 .. 1 887: waves 0 and 1): waves 2 and 3 repeat their pass 6 instead -- every wave issues the same
  // number of requests and the request sequence has no branch in it
  const int last_pass = wave < 2 ? kPasses - 1 : kPasses - 2;
  if (wave >= 2) { s_off[kPasses - 1] = s_off[kPasses - 2]; s_row[kPasses - 1] = s_row[kPasses - 2]; s_col[kPasses - 1] = s_col[kPasses - 2]; }
  int f_n, f_tr, f_sc;   // strip counters of the NEXT strip to fetch: no division in the loop
  {
    const int n = s0 / (TR * SC), rem = s0 - n * (TR * SC);
    f_n = n; f_tr = rem / SC; f_sc = rem - f_tr * SC;
  }
  // DMA requests of one strip, one at a time (request j of this wave = staging pass j): fetch_begin() takes the next strip's counters,
  // fetch_piece(j) issues pass j.  `live` false (past the workgroup's last strip): an empty descriptor -- zeros land in a stage nobody reads.
  int fn = 0, ftr = 0, fsc = 0;
  unsigned f_dp = 0, f_dq = 0;
  bool f_inner = true, f_live = true;
  auto fetch_begin = [&](bool live) __attribute__((always_inline)) {
    fn = f_n; ftr = f_tr; fsc = f_sc;
    if (++f_sc == SC) { f_sc = 0; if (++f_tr == TR) { f_tr = 0; ++f_n; } }
    f_live = live;
    f_dp = (unsigned)((((fn * p.PH + 4 * ftr) * p.PW + 16 * fsc) * p.ldp) * 4);
    // interior strip: its 4 x 16 pixels of P and the 6 x 18 pixels of Q around them all lie inside the images
    f_inner = 4 * ftr + 4 <= p.PH && 16 * fsc + 16 <= p.PW && qs * 4 * ftr + oy >= 0 && qs * (4 * ftr + 5) + oy < p.QH &&
              qs * 16 * fsc + ox >= 0 && qs * (16 * fsc + 17) + ox < p.QW;
    // (signed: the first patch pixel of an edge strip may lie above / left of the image -- those strips add it per lane below)
    f_dq = (unsigned)((((fn * p.QH + qs * 4 * ftr + oy) * p.QW + qs * 16 * fsc + ox) * p.ldq) * 4);
  };
  // per-lane offsets of the strip taken by fetch_begin: the staging offsets themselves for an interior strip (its displacement travels in
  // the requests' scalar offsets); for an edge strip -- the only place with vector arithmetic and a branch, kept OUT of the MFMA sequence
  // so that that stays one basic block -- out-of-image pixels get an out-of-range offset and the Q displacement (which may point above /
  // left of the image) is added per lane
  unsigned off8[kPasses];
  unsigned f_sq = 0;   // scalar offset of the Q requests
  auto fetch_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < kPasses; ++j) off8[j] = s_off[j];
    f_sq = f_dq;
    if (!f_inner) {
      f_sq = 0;
#pragma unroll
      for (int j = 0; j < kPasses; ++j) {
        if (j < 4) {
          const int a = 4 * ftr + s_row[j], b = 16 * fsc + s_col[j];
          if (!(a < p.PH && b < p.PW)) off8[j] = kOob;
        } else {
          const int a = qs * (4 * ftr + s_row[j]) + oy, b = qs * (16 * fsc + s_col[j]) + ox;
          off8[j] = ((unsigned)a < (unsigned)p.QH && (unsigned)b < (unsigned)p.QW && s_off[j] != kOob) ? s_off[j] + f_dq : kOob;
        }
      }
    }
  };
  auto fetch_piece = [&](int buf, int j) __attribute__((always_inline)) {
    const int jd = j == kPasses - 1 ? last_pass : j;   // (destination pass)
    float* st = smem + buf * kStage;
    const __amdgpu_buffer_rsrc_t rl = f_live ? (j < 4 ? rp : rq) : __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_ptr_t)(st + (jd * 256 + wave * 64) * 4), 16, (int)off8[j], (int)(j < 4 ? f_dp : f_sq), 0, 0);
  };

  f32x4 acc[64], accv[8];
#pragma unroll
  for (int j = 0; j < 64; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) accv[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int m = lane & 15, kg = lane >> 4;
  const int pb = (4 * kg) * kPC + 32 * iw + m, qb = kPFloats + (4 * kg) * kQC + 16 * jw + m;   // this lane's tile / channel inside the P / Q images
  const W4Consts kc = w4_consts();
  f32x2v PR[4][4];   // raw P: [row][column] = (channel block 0, channel block 1) of the lane's tile
  f32x2v QR[3][6];   // raw Q: [column pair c][row] = pixels (row, 2 c), (row, 2 c + 1) of the lane's patch
  f32x2v Ap[6][6];   // V p V^T: [xi][nu] = (block 0, block 1)
  f32x2v Vq[6][3];   // B^T q B: row xi as (v0, v5), (v1, v3), (v2, v4)
  // raw read r of a strip (34 of them): 0..15 P pixel (r / 4, r % 4), 16..33 Q pair (column pair (r - 16) / 6, row (r - 16) % 6)
  auto raw_read = [&](const float* st, int r) __attribute__((always_inline)) {
    if (r < 16) {
      const int u = r >> 2, v = r & 3;
      PR[u][v] = f32x2v{st[pb + (u * 16 + v) * kPC], st[pb + (u * 16 + v) * kPC + 16]};
    } else {
      const int c = (r - 16) / 6, u = (r - 16) - 6 * c;
      QR[c][u] = f32x2v{st[qb + (u * 18 + 2 * c) * kQC], st[qb + (u * 18 + 2 * c + 1) * kQC]};
    }
  };
  auto transforms = [&]() __attribute__((always_inline)) {
    {   // P side: V p V^T, both channel blocks at once
      f32x2v t[6][4];   // vertical pass: [xi][column]
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        f32x2v o[6];
        v4_pk(PR[0][v], PR[1][v], PR[2][v], PR[3][v], o, kc);
#pragma unroll
        for (int x = 0; x < 6; ++x) t[x][v] = o[x];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 6; ++x) v4_pk(t[x][0], t[x][1], t[x][2], t[x][3], Ap[x], kc);
      __builtin_amdgcn_sched_barrier(0);
    }
    {   // Q side: B^T q B
      f32x2v T[6][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        f32x2v t6[6];
        bt6_cols(QR[c], t6, kc);
#pragma unroll
        for (int x = 0; x < 6; ++x) T[x][c] = t6[x];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 6; ++x) bt6_row(T[x][0], T[x][1], T[x][2], Vq[x], kc);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto mfma_slot = [&](int s) __attribute__((always_inline)) {
    const int pos = s >> 1, ib = s & 1, x = pos / 6, y = pos - 6 * x;
    const float av = ib ? Ap[x][y].y : Ap[x][y].x;
    if (s < 64) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v_elem(Vq[x], y), acc[s], 0, 0, 0);
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accv[s - 64]) : "v"(av), "v"(v_elem(Vq[x], y)));
  };
  // ---- prologue: strip s0 into stage 0, then strip s0 + 1 into stage 1 (in flight while strip s0 is read and transformed)
  if (s0 < s1) {
    fetch_begin(true);
    fetch_offsets();
#pragma unroll
    for (int j = 0; j < kPasses; ++j) fetch_piece(0, j);
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    __syncthreads();
    fetch_begin(s0 + 1 < s1);
    fetch_offsets();
#pragma unroll
    for (int j = 0; j < kPasses; ++j) fetch_piece(1, j);
#pragma unroll
    for (int r = 0; r < 34; ++r) raw_read(smem, r);
  }
  for (int s = s0; s < s1; ++s) {
    const int buf = (s - s0) & 1;
    const float* stn = smem + (buf ^ 1) * kStage;   // strip s + 1
    // (the offsets of strip s + 2 first: the only branch of the k-step -- edge strips -- then everything below is ONE basic block; with
    // the branch between the two MFMA halves the register allocator copied every accumulator of the first half, 140 moves per k-step)
    fetch_begin(s + 2 < s1);
    fetch_offsets();
    __builtin_amdgcn_sched_barrier(0);
    transforms();
#pragma unroll
    for (int t = 0; t < 36; ++t) {
      mfma_slot(t);
      __builtin_amdgcn_sched_barrier(0);
    }
    // strip s + 1 has landed (requested one k-step ago), and every wave is done reading strip s (its raw reads were the last thing of the
    // previous k-step): stage `buf` is free for strip s + 2
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 36; t < 72; ++t) {
      mfma_slot(t);
      // the slot's memory instructions, directly behind the MFMA: 8 DMA requests of strip s + 2, then the 34 raw reads of strip s + 1
      // (34 + 8 = 42 in 36 slots: the first six read slots take two)
      const int q = t - 36;
      if (q < kPasses) fetch_piece(buf, q);
      else {
        const int r0 = q - kPasses < 6 ? 2 * (q - kPasses) : (q - kPasses) + 6;
        raw_read(stn, r0);
        if (q - kPasses < 6) raw_read(stn, r0 + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // (the reads behind the last k-step's MFMAs fetched a stage nobody needs; nothing is waited for)

  // ---- output transform g = A^T (D U D) A per (i, j) pair of the lane: accumulator element r of block (pos, ib) is
  // (i = i0 + 32 iw + 16 ib + 4 kg + r, j = j0 + 16 jw + m)
  float* slab = p.ws + (size_t)gidx * p.slab_elems + (size_t)split * p.T * p.PC * p.QC;
  const int jc = j0 + 16 * jw + m;
  const double D[6] = {1.0 / kWN0, 1.0 / kWNa, 1.0 / kWNa, 1.0 / kWNb, 1.0 / kWNb, 1.0};   // 1 / N_j (wino4_xform.hpp), left out of the K loop
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv[3][6];   // A^T (D U D), rows a, columns nu;  A^T = [[1, 1, 1, 1, 1, 0], [0, a, -a, b, -b, 0], [0, a2, a2, b2, b2, 1]]
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) {
        float u[6];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
          const int blk = 2 * (xi * 6 + nu) + ib;
          const float raw = blk < 64 ? acc_read(acc[blk < 64 ? blk : 0][r]) : accv[blk >= 64 ? blk - 64 : 0][r];
          u[xi] = raw * (float)(D[xi] * D[nu]);
        }
        const float pp = u[1] + u[2], qq = u[1] - u[2], uu = u[3] + u[4], ww = u[3] - u[4];
        sv[0][nu] = u[0] + pp + uu;
        sv[1][nu] = __builtin_fmaf(kWb, ww, kWa * qq);
        sv[2][nu] = __builtin_fmaf(kWb2, uu, kWa2 * pp) + u[5];
      }
      const int i = i0 + 32 * iw + 16 * ib + 4 * kg + r;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float pp = sv[a][1] + sv[a][2], qq = sv[a][1] - sv[a][2], uu = sv[a][3] + sv[a][4], ww = sv[a][3] - sv[a][4];
        const float g0 = sv[a][0] + pp + uu, g1 = __builtin_fmaf(kWb, ww, kWa * qq), g2 = __builtin_fmaf(kWb2, uu, kWa2 * pp) + sv[a][5];
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
