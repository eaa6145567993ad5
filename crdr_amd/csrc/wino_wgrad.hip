// Weight gradient of 3x3 stride-1 convolutions by Winograd minimal filtering F(3x3, 2x2) on the exact-fp32 matrix cores.
//
//   g[i][j][r][s] = sum_{n, a, b} P[n, a, b, i] * Q[n, a - pad + r, b - pad + s, j]          (wgrad.hip's statement, stride 1)
//
// Cut P (dy of a Conv2d) into 2 x 2 tiles and Q (x) into the 4 x 4 patches they meet: per tile the 3 x 3 taps are the
// correlation of the patch with the 2 x 2 tile, which minimal filtering does in 16 products instead of 36:
//   g = sum_tiles A^T [ (G p G^T) . (B^T q B) ] A,   A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]],  G = [[1,0],[.5,.5],[.5,-.5],[0,1]],
//   B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,-1,0,1]]     (Toom-Cook points 0, 1, -1, inf: the transposition of wino.hip's F(2x2, 3x3))
// i.e. 16 GEMMs [I x tiles] . [tiles x J] whose reduction axis is the tile index, and ONE output transform per workgroup.
//
// Workgroup = 8 waves = 64 channels of P x 64 channels of Q x all 16 positions, over a contiguous range of strips (a strip = 8
// consecutive tiles of one tile row = 2 x 16 pixels of P and the 4 x 18 pixels of Q around them).  wave (ph, wn, wm): transform
// rows xi in {2 ph, 2 ph + 1}, P channels 32 wm .., Q channels 32 wn ..: 8 accumulator blocks of 32 x 32.  A strip is one LDS
// stage ([pixel][64 channels], exactly the NHWC memory layout, LDS-DMA, double buffered); an MFMA consumes two tiles (k = 2: lane
// half kh holds tile 2 j + kh), so a strip is 4 k-steps of 8 MFMAs per wave.  Each lane reads its channel's 4 + 12 raw values
// of its tile with ds_read_b32 (32 consecutive channels per half-wave: conflict free), transforms them in registers and feeds
// them as the A (P side) and B (Q side) operands.  Epilogue: the two waves of a (wm, wn) pair each hold half of the xi sum; per tap
// row the ph = 1 wave hands its part of A^T U A over through LDS and the ph = 0 wave writes the sum into the slab of this split --
// the layout of wgrad_kernel's slabs ([split][tap][PC][QC]), so the deferred fixed-order reduce (wgrad_reduce_batched) and
// everything behind it are unchanged.  fp32 arithmetic throughout (transforms: +-1 and halves).
#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "wgrad_args.hpp"

namespace crdr {

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned kOob = 0x80000000u;
constexpr int kPFloats = 2 * 16 * 64, kQFloats = 4 * 18 * 64;
constexpr int kStage = kPFloats + kQFloats;        // floats per stage
constexpr int kPieces = kStage / 4;                // 16-byte pieces per stage (1664 = 26 wave instructions)
constexpr int kPasses = (kPieces + 511) / 512;
constexpr int kXFloats = 4 * 48 * 64;              // epilogue hand-over: [pair][3 taps x 16 registers][64 lanes]
constexpr int kLdsFloats = (2 * kStage > kXFloats ? 2 * kStage : kXFloats);

template <int PH>
__device__ __forceinline__ void wino_wgrad_loop(const WgradArgs& p, float* smem, int s0, int s1, int i0, int j0, const __amdgpu_buffer_rsrc_t rp,
                                                const __amdgpu_buffer_rsrc_t rq, int tid, int lane, int wave, int wm, int wn, f32x16 (&acc)[8]) {
  const int TR = (p.PH + 1) >> 1, SC = (p.PW + 15) >> 4;
  // staging: piece S = tid + 512 pass -> P: [row 2][col 16][16 pieces], Q: [row 4][col 18][16 pieces]
  int s_row[kPasses], s_col[kPasses], s_ch[kPasses];
  bool s_isp[kPasses];
#pragma unroll
  for (int j = 0; j < kPasses; ++j) {
    const int S = tid + 512 * j;
    const bool isp = S < kPFloats / 4;
    const int px = (isp ? S : S - kPFloats / 4) >> 4;
    s_isp[j] = isp;
    s_ch[j] = (S & 15) * 4;
    s_row[j] = isp ? px >> 4 : px / 18;
    s_col[j] = isp ? px & 15 : px - (px / 18) * 18;
  }
  // per-thread constants of the staging: lane part of the byte offsets (strip (n, tr, sc) = (0, 0, 0)) and whether the piece's
  // channels exist.  The strip's displacement is uniform and travels in the instruction's scalar offset; bounds are only
  // tested for strips that touch an image border (every non-MFMA instruction of the loop costs matrix time).
  unsigned s_off[kPasses];
  bool s_chok[kPasses];
#pragma unroll
  for (int j = 0; j < kPasses; ++j) {
    if (s_isp[j]) {
      s_chok[j] = i0 + s_ch[j] < p.PC;
      s_off[j] = (unsigned)(((s_row[j] * p.PW + s_col[j]) * p.ldp + i0 + s_ch[j]) * 4);
    } else {
      s_chok[j] = j0 + s_ch[j] < p.QC;
      s_off[j] = (unsigned)((((s_row[j] - p.pad) * p.QW + s_col[j] - p.pad) * p.ldq + j0 + s_ch[j]) * 4);   // (may wrap: added to the strip's displacement below)
    }
    if (!s_chok[j]) s_off[j] = kOob;
  }
  // strip counters (n, tr, sc) of the NEXT strip to fetch: no division in the loop
  int f_n, f_tr, f_sc;
  {
    const int n = s0 / (TR * SC), rem = s0 - n * (TR * SC);
    f_n = n; f_tr = rem / SC; f_sc = rem - f_tr * SC;
  }
  auto fetch = [&](int buf) __attribute__((always_inline)) {
    const int n = f_n, tr = f_tr, sc = f_sc;
    if (++f_sc == SC) { f_sc = 0; if (++f_tr == TR) { f_tr = 0; ++f_n; } }
    float* st = smem + buf * kStage;
    const unsigned dp = (unsigned)((((n * p.PH + 2 * tr) * p.PW + 16 * sc) * p.ldp) * 4);
    const unsigned dq = (unsigned)((((n * p.QH + 2 * tr) * p.QW + 16 * sc) * p.ldq) * 4);
    // interior strip: its 2 x 16 pixels of P and the 4 x 18 pixels of Q around them all lie inside the images
    const bool inner = 2 * tr + 2 <= p.PH && 16 * sc + 16 <= p.PW && 2 * tr - p.pad >= 0 && 2 * tr - p.pad + 4 <= p.QH &&
                       16 * sc - p.pad >= 0 && 16 * sc - p.pad + 18 <= p.QW;
#pragma unroll
    for (int j = 0; j < kPasses; ++j) {
      if (j * 512 + wave * 64 >= kPieces) break;   // (wave-uniform)
      unsigned off = s_off[j];
      if (!inner) {
        if (s_isp[j]) {
          const int a = 2 * tr + s_row[j], b = 16 * sc + s_col[j];
          if (!(a < p.PH && b < p.PW)) off = kOob;
        } else {
          const int a = 2 * tr - p.pad + s_row[j], b = 16 * sc - p.pad + s_col[j];
          if (!((unsigned)a < (unsigned)p.QH && (unsigned)b < (unsigned)p.QW)) off = kOob;
        }
      }
      // (Q's lane offset may have wrapped below zero for pad > 0: lane offset and displacement are added before the range check
      // only in the vector operand, so the sum goes there)
      if (s_isp[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr_t)(st + (j * 512 + wave * 64) * 4), 16, (int)off, (int)dp, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)(st + (j * 512 + wave * 64) * 4), 16, (int)(off == kOob ? kOob : off + dq), 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  const int m = lane & 31, kh = lane >> 5;
  const int po = wm * 32 + m, qo = kPFloats + wn * 32 + m;   // this lane's channel inside the P / Q images
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float* st = smem + buf * kStage;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tc = 2 * j + kh;   // this lane's tile of the k-step
      float dy[2][2], d[3][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) dy[i][jj] = st[po + ((i * 16) + 2 * tc + jj) * 64];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) d[a][b] = st[qo + (((PH + a) * 18) + 2 * tc + b) * 64];
      // P side: rows xi = 2 PH, 2 PH + 1 of G p G^T -- WITHOUT G's halves: positions (xi, nu) with xi in {1, 2} and / or nu in
      // {1, 2} are accumulated 2x / 4x too large and scaled once per workgroup in the output transform (a power of two commutes
      // with every rounding of the sum), which takes 24 multiplies out of every k-step
      float zr[2][2], z[2][4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if constexpr (PH == 0) { zr[0][jj] = dy[0][jj]; zr[1][jj] = dy[0][jj] + dy[1][jj]; }
        else { zr[0][jj] = dy[0][jj] - dy[1][jj]; zr[1][jj] = dy[1][jj]; }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        z[a][0] = zr[a][0]; z[a][1] = zr[a][0] + zr[a][1]; z[a][2] = zr[a][0] - zr[a][1]; z[a][3] = zr[a][1];
      }
      // Q side: rows xi = 2 PH, 2 PH + 1 of B^T q B (raw rows PH .. PH + 2)
      float t[2][4], v[2][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if constexpr (PH == 0) { t[0][b] = d[0][b] - d[2][b]; t[1][b] = d[1][b] + d[2][b]; }   // xi = 0, 1 from rows 0, 1, 2
        else { t[0][b] = d[1][b] - d[0][b]; t[1][b] = d[2][b] - d[0][b]; }                     // xi = 2, 3 from rows 1, 2, 3
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        v[a][0] = t[a][0] - t[a][2]; v[a][1] = t[a][1] + t[a][2]; v[a][2] = t[a][2] - t[a][1]; v[a][3] = t[a][3] - t[a][1];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) acc[a * 4 + nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(z[a][nu], v[a][nu], acc[a * 4 + nu], 0, 0, 0);
    }
  };
  if (s0 < s1) fetch(0);
  __syncthreads();
  for (int s = s0; s < s1; ++s) {
    const int buf = (s - s0) & 1;
    if (s + 1 < s1) fetch(buf ^ 1);
    compute(buf);
    __syncthreads();
  }
}

__global__ __launch_bounds__(512) void wino_wgrad_kernel(const WgradArgs p_, const WgradGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ph = wave & 1, wn = (wave >> 1) & 1, wm = wave >> 2;
  WgradArgs p = p_;
  const int gidx = blockIdx.z;
  if (p.ngroup > 1) { p.p = grp.p[gidx]; p.q = grp.q[gidx]; }
  const int it = blockIdx.x / p.jtiles, jt = blockIdx.x - it * p.jtiles;
  const int i0 = it * 64, j0 = jt * 64;
  const int split = blockIdx.y;
  const int TR = (p.PH + 1) >> 1, SC = (p.PW + 15) >> 4;
  const long long S = (long long)p.N * TR * SC;
  const int s0 = (int)(S * split / p.nsplit), s1 = (int)(S * (split + 1) / p.nsplit);
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.p), 0, p.p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.q), 0, p.q_bytes, 0x00020000);

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  if (ph == 0) wino_wgrad_loop<0>(p, smem, s0, s1, i0, j0, rp, rq, tid, lane, wave, wm, wn, acc);
  else wino_wgrad_loop<1>(p, smem, s0, s1, i0, j0, rp, rq, tid, lane, wave, wm, wn, acc);
  // (the loop ends with a barrier: the stages are free)

  // ---- output transform A^T U A and hand-over.  This wave holds U[xi][nu] for xi = 2 ph, 2 ph + 1; over xi, tap row a takes
  //   a = 0: U0 + U1 + U2,  a = 1: U1 - U2,  a = 2: U1 + U2 + U3;   ph 0 holds (U0, U1), ph 1 (U2, U3).
  float* slab = p.ws + (size_t)gidx * p.slab_elems + (size_t)split * 9 * p.PC * p.QC;
  float* sX = smem + (wave >> 1) * (48 * 64);
  const int c = j0 + wn * 32 + (lane & 31), fh = lane >> 5;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float pt[3][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sv[4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        // G's halves, left out of the K loop: row xi = 2 ph + a carries 1/2 for xi in {1, 2} (a = 1 of ph 0, a = 0 of ph 1), column nu likewise
        const float fc = (nu == 1 || nu == 2) ? 0.5f : 1.0f;
        const float ua = acc[nu][r] * (ph == 0 ? fc : 0.5f * fc), ub = acc[4 + nu][r] * (ph == 0 ? 0.5f * fc : fc);
        if (ph == 0) sv[nu] = a == 0 ? ua + ub : ub;
        else sv[nu] = a == 0 ? ua : (a == 1 ? -ua : ua + ub);
      }
      pt[0][r] = sv[0] + sv[1] + sv[2];
      pt[1][r] = sv[1] - sv[2];
      pt[2][r] = sv[1] + sv[2] + sv[3];
    }
    if (ph == 1) {
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sX[(b * 16 + r) * 64 + lane] = pt[b][r];
    }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = pt[b][r] + sX[(b * 16 + r) * 64 + lane];   // (part of xi 0, 1) + (part of xi 2, 3)
          const int i = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (i < p.PC && c < p.QC) slab[((size_t)(a * 3 + b) * p.PC + i) * p.QC + c] = v;
        }
    }
    __syncthreads();
  }
}

}  // namespace

void wino_wgrad_launch(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t s) {
  static std::atomic<bool> attr_done{false};
  if (!attr_done.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(wino_wgrad_kernel, grid, dim3(512), (size_t)kLdsFloats * sizeof(float), s, a, grp);
}

}  // namespace crdr
