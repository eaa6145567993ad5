// Winograd F(4x4, 3x3) convolution on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) for the 3x3 stride-1 layers at >= 64 pixels
// of width (elic_layers.py:23-36, cheng_nlam.py:31-46, clic21_gvae_discriminator.py:27-40) and their input gradients:
//   Y = A^T [ (G g G^T) . (B^T d B) ] A   with 6x6 transforms (Lavin & Gray 2016, interpolation points 0, +-1, +-2, inf):
// 36 element-wise products per 4x4 outputs and channel pair instead of 144 -- 4x fewer MFMAs than the implicit GEMM, 1.78x fewer
// than the F(2x2, 3x3) kernel of wino.hip.  Arithmetic is fp32 throughout (the filter transform is evaluated in double and rounded
// once); the transforms carry the constants 4, 5, 8 and 1/4 ... 1/24, so the result deviates from the direct form by ~5e-6 .. 1e-5
// of the output scale at 96 .. 256 input channels (the direct kernels: ~1e-6): a tuner candidate for TRAINING launches only (the
// codec never runs tuned plans), behind the forced-algorithm id of Winograd variant 2, tests/test_gpu_wino.py.
//
// One output tile = 8 rows x 64 columns of output pixels (2 x 16 Winograd tiles of 4 x 4 = the 32 MFMA columns) of one image x 32
// output channels, on FOUR waves (one per SIMD: 512 registers each, no co-resident wave to share the matrix pipe with):
//   wave (ph, pw): transform rows xi in {3 ph .. 3 ph + 2}, columns nu in {3 pw .. 3 pw + 2}: 9 of the 36 positions, i.e. 9
//   accumulator blocks of 32 channels x 32 tiles.  MFMA operand A = filter fragment (row = channel), B = transformed data (column =
//   tile): a lane ends up with 16 channels (4 groups of 4 consecutive ones) of ONE tile, so the epilogue works with 16-byte accesses.
// K loop: sub-steps of 8 input channels, one barrier each, operands double buffered in LDS and filled by LDS-DMA:
//   * the raw 10 x 66 x 8 input patch, stored by pixel class (row & 3, column & 3): [half h = channels 4h..4h+3][class 16][3 rows][17
//     slots]; the 16 tiles of a tile row read patch pixel (i, j) from 16 consecutive 16-byte slots (conflict free);
//   * the transformed filters of the sub-step: [position 36][h][channel 32][4] = 36 KiB, contiguous in memory and in LDS.
//   Every wave reads the 5 x 5 patch pixels its 3 x 3 positions depend on (rows ph .. ph + 4, columns pw .. pw + 4), applies the
//   two 1-D transforms in registers (6 fma-class operations per 5 inputs and 3 outputs) and issues 36 MFMAs.
// Epilogue: every wave forms its part of A^T M A for all 16 output pixels of a tile (partial sums over its own xi, nu); wave w
// finishes output row w of every tile: three rounds of hand-over through LDS in fixed order, then the element-wise epilogue of
// the implicit-GEMM kernel (same order of operations) with 16-byte buffer loads / stores.
#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "igemm_args.hpp"
#include "wino.hpp"

namespace crdr {

namespace {

constexpr int kNT4 = 256;
constexpr int kTY = 2, kTX = 16;                       // Winograd tiles per output tile (rows, columns)
constexpr int kInUsed4 = 2 * 16 * 3 * 17;              // 16-byte slots of the input patch image per stage
constexpr int kInPieces = (kInUsed4 + 63) / 64;        // DMA instructions (1 KiB each) for it: 26
constexpr int kInSlots4 = kInPieces * 64;
constexpr int kUSlots4 = 36 * 2 * 32;                  // slots of one filter block (8 channels x 32 output channels): 36 pieces
constexpr int kUPieces = kUSlots4 / 64;
constexpr int kStageFloats4 = (kInSlots4 + kUSlots4) * 4;
constexpr int kXFloats4 = 2 * 4 * 64 * 64;             // hand-over area: 2 regions x 4 waves x 64 values x 64 lanes (128 KiB)
constexpr int kLdsFloats4 = (2 * kStageFloats4 > kXFloats4 ? 2 * kStageFloats4 : kXFloats4);

__device__ __forceinline__ void lds_barrier4() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// 1-D data transform, three of the six outputs of B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],
// [0,4,0,-5,0,1]] from the five inputs they depend on.  HALF 0: outputs 0, 1, 2 from inputs d0..d4; HALF 1: outputs 3, 4, 5 from
// inputs d1..d5 (passed as e0..e4).
template <int HALF>
__device__ __forceinline__ void bt6(const f32x4 e0, const f32x4 e1, const f32x4 e2, const f32x4 e3, const f32x4 e4, f32x4 (&o)[3]) {
  if constexpr (HALF == 0) {
    const f32x4 a = e4 - 4.0f * e2, b = e3 - 4.0f * e1;
    o[0] = 4.0f * e0 + (e4 - 5.0f * e2);
    o[1] = a + b;
    o[2] = a - b;
  } else {   // e0..e4 = d1..d5
    const f32x4 c = e3 - e1, e = e2 - e0;
    o[0] = c + 2.0f * e;
    o[1] = c - 2.0f * e;
    o[2] = 4.0f * e0 + (e4 - 5.0f * e2);
  }
}

struct Wino4Tile { int gidx, n, oh0, ow0, n0, patch; };

__device__ __forceinline__ Wino4Tile wino4_tile(const IgemmArgs& p, int vb, int gx, int gyn, int gz) {
  // XCD-aware order (see igemm_kernel.hpp): the hardware deals workgroups round-robin over the 8 XCDs; every XCD walks a contiguous
  // range of (patch, N tile) pairs, the N tiles of a patch back to back (they re-read the patch out of that XCD's L2)
  Wino4Tile t;
  const int T = gx * gyn, nwg = T * gz, cpx = nwg >> 3;
  const int q = vb < cpx * 8 ? (vb & 7) * cpx + (vb >> 3) : vb;
  t.gidx = q / T;
  const int r = q - t.gidx * T;
  const int tn = r % gyn;
  t.patch = r / gyn;
  const int ppi = p.GH * p.GW;
  t.n = t.patch / ppi;
  const int prem = t.patch - t.n * ppi, by = prem / p.GW, bx = prem - by * p.GW;
  t.oh0 = by * (4 * kTY);
  t.ow0 = bx * (4 * kTX);
  t.n0 = tn * 32;
  return t;
}

// byte offset (into the input tensor's descriptor) of the pixel a DMA lane stages for input piece `piece`: slot S = piece * 64 + lane ->
// (h, class (ci, cj), R, Cc) -> patch pixel (4 R + ci, 4 Cc + cj), channels 4h .. 4h + 3; out of range where the slot is unused or the
// pixel lies outside the image (the range check of the buffer load then delivers zeros: padding)
__device__ __forceinline__ unsigned wino4_in_off(const IgemmArgs& p, const Wino4Tile& t, int piece, int lane) {
  const int S = piece * 64 + lane;
  const int h = S / (16 * 51), rem = S - h * (16 * 51), cls = rem / 51, r2 = rem - cls * 51, R = r2 / 17, Cc = r2 - R * 17;
  const int ci = cls >> 2, cj = cls & 3;
  const int pi = 4 * R + ci, pj = 4 * Cc + cj;
  const int ih = t.oh0 - p.si + pi, iw = t.ow0 - p.si + pj;
  const bool ok = S < kInUsed4 && pi < 4 * kTY + 2 && pj < 4 * kTX + 2 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
  return ok ? (unsigned)((((t.n * p.H + ih) * p.W + iw) * p.ldx + 4 * h) * 4) : kOobOffset;
}

template <int PH, int PW>
__device__ __forceinline__ void wino4_loop(const IgemmArgs& p, float* smem, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ru,
                                           const unsigned (&a_off)[7], unsigned u_off0, int lane, int wave, f32x16 (&acc)[3][3]) {
  const int K8 = p.kchunks;
  const int m = lane & 31, fh = lane >> 5;
  const int ty = m >> 4, tx = m & 15;
  // float offsets of this lane's 25 raw reads inside a stage: patch pixel (PH + a, PW + b), a, b < 5, of tile (ty, tx)
  int ro[5][5];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int i = PH + a, j = PW + b;
      const int cls = (i & 3) * 4 + (j & 3);
      ro[a][b] = (((fh * 16 + cls) * 3 + ty + (i >> 2)) * 17 + tx + (j >> 2)) * 4;
    }
  // filter fragment of position (xi, nu) = (3 PH + x, 3 PW + y): + ((xi * 6 + nu) * 64) * 4 floats
  const int bo = kInSlots4 * 4 + (fh * 32 + m) * 4;

  auto issue = [&](int k8) __attribute__((always_inline)) {
    float* st = smem + (k8 & 1) * kStageFloats4;
    const unsigned dch = (unsigned)(k8 * 32);
    const bool tail = k8 * 8 + 8 > p.Cin;   // (uniform) the chunk's upper half lies past Cin
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int piece = wave + 4 * j;
      if (piece < kInPieces) {
        unsigned off = a_off[j];
        // the channel half of a slot: h = S / 816 with S = piece * 64 + lane
        if (tail && (piece * 64 + lane) >= 16 * 51) off = kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(st + piece * 256), 16, (int)off, (int)dch, 0, 0);
      }
    }
    const unsigned ub = u_off0 + (unsigned)k8 * (kUSlots4 * 16u);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int piece = wave * 9 + j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(st + kInSlots4 * 4 + piece * 256), 16, (int)((piece * 64 + lane) * 16), (int)ub, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float* st = smem + buf * kStageFloats4;
    // vertical pass column by column: t[x][b] = sum_a BT[3 PH + x][PH + a] d[a][b]
    f32x4 t[3][5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      f32x4 d[5];
#pragma unroll
      for (int a = 0; a < 5; ++a) d[a] = *reinterpret_cast<const f32x4*>(st + ro[a][b]);
      f32x4 o[3];
      bt6<PH>(d[0], d[1], d[2], d[3], d[4], o);
#pragma unroll
      for (int x = 0; x < 3; ++x) t[x][b] = o[x];
    }
    // horizontal pass: v[x][y] = sum_b BT[3 PW + y][PW + b] t[x][b]
    f32x4 v[3][3];
#pragma unroll
    for (int x = 0; x < 3; ++x) bt6<PW>(t[x][0], t[x][1], t[x][2], t[x][3], t[x][4], v[x]);
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 3; ++y) {
        const int pos = (3 * PH + x) * 6 + 3 * PW + y;
        const f32x4 uf = *reinterpret_cast<const f32x4*>(st + bo + pos * 256);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[s], v[x][y][s], acc[x][y], 0, 0, 0);
      }
  };

  issue(0);
  __syncthreads();
  for (int k8 = 0; k8 < K8; ++k8) {
    if (k8 + 1 < K8) issue(k8 + 1);
    compute(k8 & 1);
    __syncthreads();
  }
}

// Output transform rows of A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]] restricted to a wave's three transform
// indices: HALF 0: indices 0, 1, 2; HALF 1: indices 3, 4, 5.  o[a] = sum_x AT[a][3 HALF + x] m[x].
template <int HALF>
__device__ __forceinline__ void at6(const float m0, const float m1, const float m2, float (&o)[4]) {
  if constexpr (HALF == 0) {
    const float s = m1 + m2, d = m1 - m2;
    o[0] = m0 + s; o[1] = d; o[2] = s; o[3] = d;
  } else {
    const float s = m0 + m1, d = m0 - m1;
    o[0] = s; o[1] = 2.0f * d; o[2] = 4.0f * s; o[3] = 8.0f * d + m2;
  }
}

template <int PH, int PW>
__device__ __forceinline__ void wino4_finish(const IgemmArgs& p, const Wino4Tile& tl, float* smem, const float* sV, float* sS, int lane, int wave,
                                             f32x16 (&acc)[3][3]) {
  // this wave's partial sums of all 16 output pixels (a, b) of its tiles, per accumulator register r
  // row w stays (own[b][r]), rows w + 1, w + 2, w + 3 (mod 4) go to the waves that finish them: three rounds through LDS
  float own[4][16];
  float* sX = smem;   // [region 2][wave 4][64 values][64 lanes]
#pragma unroll
  for (int rnd = 0; rnd < 4; ++rnd) {
    // the output row handled in this round: rnd 0 = own row, rnd k = row (wave + k) & 3 written for its owner
    float part[4][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s[3][4];   // s[y][a]: vertical output transform of column y
#pragma unroll
      for (int y = 0; y < 3; ++y) at6<PH>(acc[0][y][r], acc[1][y][r], acc[2][y][r], s[y]);
      // the rows are selected at run time per wave (uniform): compute all four, pick below
      float yb[4][4];   // [a][b]
#pragma unroll
      for (int a = 0; a < 4; ++a) at6<PW>(s[0][a], s[1][a], s[2][a], yb[a]);
      const int arow = (wave + rnd) & 3;
#pragma unroll
      for (int b = 0; b < 4; ++b) part[b][r] = arow == 0 ? yb[0][b] : arow == 1 ? yb[1][b] : arow == 2 ? yb[2][b] : yb[3][b];
    }
    if (rnd == 0) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) own[b][r] = part[b][r];
    } else {
      float* dst = sX + (size_t)((rnd & 1) * 4 + wave) * 4096;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(b * 16 + r) * 64 + lane] = part[b][r];
      lds_barrier4();
      const float* src = sX + (size_t)((rnd & 1) * 4 + ((wave - rnd) & 3)) * 4096;   // the wave whose round-rnd row is mine
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) own[b][r] += src[(b * 16 + r) * 64 + lane];
      // (round rnd + 1 writes the other region; round rnd + 2 reuses this one after the barrier of round rnd + 1)
    }
  }

  // ---- element-wise epilogue + stores (order of operations: epilogue_store of igemm_kernel.hpp).  This lane: tile (ty, tx), output
  // row `wave` of it, pixels b = 0..3, channels n0 + 8 g + 4 fh + e (register r = 4 g + e).
  const int f = p.flags;
  const bool has_res = (f & CRDR_EPI_RES) != 0, has_mask = (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) != 0;
  const bool do_cs = (f & CRDR_EPI_COLSUM) != 0, accum = (f & CRDR_EPI_ACCUM) != 0;
  const int m = lane & 31, fh = lane >> 5, ty = m >> 4, tx = m & 15;
  const int oy = tl.oh0 + 4 * ty + wave, ox0 = tl.ow0 + 4 * tx;
  const bool row_ok = oy < p.OH;
  const size_t pix_row = ((size_t)tl.n * p.OH + oy) * p.OW;
  auto tdesc = [&](const float* base, int ld) __attribute__((always_inline)) {
    const unsigned long long bytes = (((unsigned long long)p.N * p.OH * p.OW - 1) * ld + p.Cout) * 4ull;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (unsigned)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ry = tdesc(p.y, p.ldy);
  auto off = [&](int ld, int b, int g) __attribute__((always_inline)) {
    const int c = tl.n0 + 8 * g + 4 * fh;
    return (row_ok && ox0 + b < p.OW && c < p.Cout) ? (unsigned)(((pix_row + ox0 + b) * ld + c) * 4) : kOobOffset;
  };
  f32x4 bias[4], vec2[4], scale[4], shift[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bias[g] = *reinterpret_cast<const f32x4*>(sV + 0 * 32 + 8 * g + 4 * fh);
    vec2[g] = *reinterpret_cast<const f32x4*>(sV + 1 * 32 + 8 * g + 4 * fh);
    scale[g] = *reinterpret_cast<const f32x4*>(sV + 2 * 32 + 8 * g + 4 * fh);
    shift[g] = *reinterpret_cast<const f32x4*>(sV + 3 * 32 + 8 * g + 4 * fh);
  }
  float cpre[16], cpost[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) cpre[r] = cpost[r] = 0.f;
  const float moff_on = (f & CRDR_EPI_MASKOFF) ? 1.f : 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    f32x4 o[4], resv[4], mskv[4], oldv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) o[g] = f32x4{own[b][4 * g], own[b][4 * g + 1], own[b][4 * g + 2], own[b][4 * g + 3]};
    if (has_res) {
      const __amdgpu_buffer_rsrc_t rr = tdesc(p.res, p.ldres);
#pragma unroll
      for (int g = 0; g < 4; ++g) resv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off(p.ldres, b, g), 0, 0));
    }
    if (has_mask) {
      const __amdgpu_buffer_rsrc_t rm = tdesc(p.mask, p.ldmask);
#pragma unroll
      for (int g = 0; g < 4; ++g) mskv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, off(p.ldmask, b, g), 0, 0));
    }
    if (accum) {
#pragma unroll
      for (int g = 0; g < 4; ++g) oldv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, off(p.ldy, b, g), 0, 0));
    }
    const bool pix_ok = row_ok && ox0 + b < p.OW;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = o[g];
      if (f & CRDR_EPI_BIAS) v += bias[g];
      if (f & CRDR_EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
      }
      if (f & CRDR_EPI_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.2f * v[e];
      }
      if (f & CRDR_EPI_VEC2) v += vec2[g];
      if (has_res) v += resv[g];
      if (f & CRDR_EPI_AFFINE) v = v * scale[g] + shift[g];
      if (do_cs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cpre[4 * g + e] += (pix_ok && tl.n0 + 8 * g + 4 * fh + e < p.Cout) ? v[e] : 0.f;
      }
      if (has_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float mv = mskv[g][e] - moff_on * vec2[g][e];
          v[e] = mv > 0.0f ? v[e] : ((f & CRDR_EPI_LRELUMASK) ? 0.2f * v[e] : 0.0f);
        }
      }
      if (do_cs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cpost[4 * g + e] += (pix_ok && tl.n0 + 8 * g + 4 * fh + e < p.Cout) ? v[e] : 0.f;
      }
      if (accum) v += oldv[g];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ry, off(p.ldy, b, g), 0, 0);
    }
  }
  if (do_cs) {
    // column sums of the tile: lane (tile m, half fh) holds 16 channels; fixed-order sum over the 32 tiles of a half and the 4 waves
    lds_barrier4();   // (the hand-over area is free again)
    float* sC = smem;   // [wave 4][which 2][16 r][64 lanes]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sC[((wave * 2 + 0) * 16 + r) * 64 + lane] = cpre[r];
      sC[((wave * 2 + 1) * 16 + r) * 64 + lane] = cpost[r];
    }
    lds_barrier4();
    const int tid = wave * 64 + lane;
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;          // channel c = 8 g + 4 fh + e
      const int g = c >> 3, h2 = (c >> 2) & 1, e = c & 3, r = 4 * g + e;
      float v = 0.f;
      for (int w2 = 0; w2 < 4; ++w2)
        for (int t2 = 0; t2 < 32; ++t2) v += sC[((w2 * 2 + which) * 16 + r) * 64 + h2 * 32 + t2];
      if (tl.n0 + c < p.Cout) p.cs[((size_t)tl.patch * 2 + which) * p.cs_ld + tl.n0 + c] = v;
    }
  }
  (void)sS;
}

__global__ __launch_bounds__(kNT4) void wino4_kernel(const IgemmArgs p_, const IgemmGroup grp, int gx, int gyn, int gz) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int total = gx * gyn * gz;
  float* sV = smem + kLdsFloats4;   // [4][32]: bias, vec2, scale, shift
  for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
    const int tid = wave * 64 + lane;
    const Wino4Tile tl = wino4_tile(p_, vb, gx, gyn, gz);
    IgemmArgs p = p_;
    if (p.ngroup > 1) {
      const int g = tl.gidx;
      p.x = grp.x[g]; p.y = grp.y[g]; p.bias = grp.bias[g]; p.mask = grp.mask[g]; p.res = grp.res[g]; p.cs = grp.cs[g];
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((((unsigned long long)p.N * p.H * p.W - 1) * p.ldx + p.Cin) * 4ull), 0x00020000);
    // transformed filters of group gidx: [N tile][chunk][2304 slots of 16 B]
    const size_t ublock = (size_t)gyn * p.kchunks * kUSlots4 * 4;   // floats per group
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_.w) + (size_t)tl.gidx * ublock, 0, (unsigned)(ublock * 4), 0x00020000);
    const unsigned u_off0 = (unsigned)(tl.n0 / 32) * (unsigned)p.kchunks * (kUSlots4 * 16u);
    unsigned a_off[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) a_off[j] = wave + 4 * j < kInPieces ? wino4_in_off(p, tl, wave + 4 * j, lane) : kOobOffset;

    f32x16 acc[3][3];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 3; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    {
      const int f0 = p.flags;
      if (tid < 32) {
        const bool live = tl.n0 + tid < p.Cout;
        sV[0 * 32 + tid] = (live && (f0 & CRDR_EPI_BIAS)) ? p.bias[tl.n0 + tid] : 0.f;
        sV[1 * 32 + tid] = (live && (f0 & (CRDR_EPI_VEC2 | CRDR_EPI_MASKOFF))) ? p.vec2[tl.n0 + tid] : 0.f;
        sV[2 * 32 + tid] = (live && (f0 & CRDR_EPI_AFFINE)) ? p.scale[tl.n0 + tid] : 1.f;
        sV[3 * 32 + tid] = (live && (f0 & CRDR_EPI_AFFINE)) ? p.shift[tl.n0 + tid] : 0.f;
      }
    }
    const int role = wave;
    if (role == 0) wino4_loop<0, 0>(p, smem, rx, ru, a_off, u_off0, lane, wave, acc);
    else if (role == 1) wino4_loop<0, 1>(p, smem, rx, ru, a_off, u_off0, lane, wave, acc);
    else if (role == 2) wino4_loop<1, 0>(p, smem, rx, ru, a_off, u_off0, lane, wave, acc);
    else wino4_loop<1, 1>(p, smem, rx, ru, a_off, u_off0, lane, wave, acc);
    // (the loop ends with a barrier: every wave is past its last LDS read, the stages are free for the hand-over)
    if (role == 0) wino4_finish<0, 0>(p, tl, smem, sV, nullptr, lane, wave, acc);
    else if (role == 1) wino4_finish<0, 1>(p, tl, smem, sV, nullptr, lane, wave, acc);
    else if (role == 2) wino4_finish<1, 0>(p, tl, smem, sV, nullptr, lane, wave, acc);
    else wino4_finish<1, 1>(p, tl, smem, sV, nullptr, lane, wave, acc);
    __syncthreads();   // hand-over area, sV: free for the next tile (and the last stores need not be waited for)
  }
}

// Filter transform U = G g G^T, G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], evaluated in double
// and rounded once, from the implicit-GEMM weight pack (tap-major [tap][wrows][wcols]) into the stage-block layout of wino4_kernel:
// [N tile of 32][chunk of 8 channels][position 36][h 2][oc 32][4].  One thread per (N tile, chunk, h, oc).
struct Wino4Taps { int widx[9]; };
__global__ void wino4_filter_kernel(const IgemmGroup grp, int ngroup, const float* w0, float* u, int Cin, int Cout, int wrows, int wcols, int kchunks,
                                    int ntile, Wino4Taps tp) {
  const long long total = (long long)ntile * kchunks * 64;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int g = blockIdx.y;
  const float* w = ngroup > 1 ? grp.w[g] : w0;
  const int oc32 = (int)(id & 31), h = (int)((id >> 5) & 1);
  const long long blk = id >> 6;   // (N tile, chunk)
  const int kc = (int)(blk % kchunks), ct = (int)(blk / kchunks);
  const int oc = ct * 32 + oc32, c0 = kc * 8 + h * 4;
  const bool live = oc < Cout && c0 < Cin;
  double g9[3][3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int wi = tp.widx[a * 3 + b];
      if (live) v = *reinterpret_cast<const f32x4*>(w + ((size_t)wi * wrows + oc) * wcols + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) g9[a][b][e] = c0 + e < Cin ? (double)v[e] : 0.0;
    }
  const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  float* dst = u + ((size_t)g * ntile * kchunks + (size_t)blk) * (kUSlots4 * 4) + ((size_t)h * 32 + oc32) * 4;
  for (int xi = 0; xi < 6; ++xi) {
    double t[3][4];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[b][e] = G[xi][0] * g9[0][b][e] + G[xi][1] * g9[1][b][e] + G[xi][2] * g9[2][b][e];
    for (int nu = 0; nu < 6; ++nu) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (float)(G[nu][0] * t[0][e] + G[nu][1] * t[1][e] + G[nu][2] * t[2][e]);
      *reinterpret_cast<f32x4*>(dst + (size_t)(xi * 6 + nu) * 256) = o;
    }
  }
}

}  // namespace

bool wino4_eligible(const crdr_conv_desc* d, int G, bool vec_ok) {
  if (!(d->kh == 3 && d->kw == 3) || d->stride != 1 || d->wlayout != 0) return false;
  const int grow = d->transposed ? 2 - 2 * d->pad : 2 * d->pad - 2;   // (a stride-1 transposed conv = a conv with pad k - 1 - pad)
  if (d->OH != d->H + grow || d->OW != d->W + grow) return false;
  if (d->pad < 0 || d->pad > 2) return false;
  if (d->C % 4 != 0 || d->ldx % 4 != 0 || d->OC % 4 != 0 || d->ldy % 4 != 0) return false;
  if (d->OW < 48) return false;   // the 8 x 64 output tile wants wide images (the F(2x2) kernel serves the rest)
  if ((d->flags & CRDR_EPI_RES) && d->ldres % 4 != 0) return false;
  if ((d->flags & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) && d->ldmask % 4 != 0) return false;
  if (d->flags & (CRDR_EPI_GATE | CRDR_EPI_PREADD | CRDR_CONV_BF16X3)) return false;
  if (G > 1 && (d->flags & (CRDR_EPI_VEC2 | CRDR_EPI_AFFINE | CRDR_EPI_MASKOFF))) return false;
  if (!vec_ok) return false;
  const long long img = ((long long)d->N * d->H + 8) * d->W * d->ldx * 4;   // one descriptor spans a whole tensor
  const long long oimg = (long long)d->N * d->OH * d->OW * std::max(std::max(d->ldy, d->ldres), d->ldmask) * 4;
  if (img >= (1ll << 31) || oimg >= (1ll << 31)) return false;
  if ((long long)wino4_workspace(d, G) / G >= (1ll << 31)) return false;
  return true;
}

size_t wino4_workspace(const crdr_conv_desc* d, int G) { return (size_t)G * cdiv(d->OC, 32) * cdiv(d->C, 8) * kUSlots4 * 16; }

int wino4_colsum_rows(const crdr_conv_desc* d) { return d->N * cdiv(d->OH, 4 * kTY) * cdiv(d->OW, 4 * kTX); }

int wino4_launch(const crdr_conv_desc* d, IgemmArgs a, const IgemmTaps& taps, const IgemmGroup& grp, int G, float* u, hipStream_t s) {
  CRDR_REQUIRE(wino4_eligible(d, G, a.vec_epi != 0), "conv2d: the F(4x4, 3x3) Winograd kernel takes 3x3 stride-1 convolutions of >= 48 output columns with "
               "C, OC %% 4 == 0, 16-byte aligned operand rows and no gate / pre-add epilogue");
  Wino4Taps wt;
  int dmin = 127;
  for (int t = 0; t < 9; ++t) dmin = std::min(dmin, (int)(signed char)(taps.packed[t] & 0xff));
  for (int t = 0; t < 9; ++t) wt.widx[t] = -1;
  for (int t = 0; t < 9; ++t) {
    const int v = taps.packed[t];
    const int dh = (int)(signed char)(v & 0xff) - dmin, dw = (int)(signed char)((v >> 8) & 0xff) - dmin;
    CRDR_REQUIRE(dh >= 0 && dh < 3 && dw >= 0 && dw < 3, "conv2d: Winograd F(4x4): tap offsets are not a 3x3 window");
    wt.widx[dh * 3 + dw] = v >> 16;
  }
  for (int t = 0; t < 9; ++t) CRDR_REQUIRE(wt.widx[t] >= 0, "conv2d: Winograd F(4x4): incomplete 3x3 window");
  const int ntile = cdiv(d->OC, 32), kchunks = cdiv(d->C, 8);
  {
    const long long total = (long long)ntile * kchunks * 64;
    hipLaunchKernelGGL(wino4_filter_kernel, dim3((unsigned)cdiv64(total, 256), G), dim3(256), 0, s, grp, G, a.w, u, d->C, d->OC, d->wrows, d->wcols, kchunks,
                       ntile, wt);
    CRDR_CHECK_LAUNCH("wino4_filter_kernel");
  }
  a.w = u;
  a.kchunks = kchunks;
  a.nphase = 1;
  a.GH = cdiv(d->OH, 4 * kTY);
  a.GW = cdiv(d->OW, 4 * kTX);
  a.si = -dmin;   // the patch starts `si` pixels above / left of its first output pixel
  a.cs_rows = wino4_colsum_rows(d);
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    return n / 8 * 8;
  }();
  const int gx = d->N * a.GH * a.GW;
  const int total = gx * ntile * G;
  static std::atomic<bool> attr_done;
  if (!attr_done.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.store(true, std::memory_order_release);
  }
  const size_t lds = (size_t)(kLdsFloats4 + 4 * 32) * sizeof(float);
  hipLaunchKernelGGL(wino4_kernel, dim3(std::min(total, ncu)), dim3(kNT4), lds, s, a, grp, gx, ntile, G);
  CRDR_CHECK_LAUNCH("wino4_kernel");
  return 0;
}

}  // namespace crdr
