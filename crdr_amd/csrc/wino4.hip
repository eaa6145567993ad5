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
constexpr int kInPieces = 28;                          // DMA instructions (1 KiB each) for it: 26 needed, 28 = 7 per wave (uniform vmcnt counts)
constexpr int kInSlots4 = kInPieces * 64;
constexpr int kUSlots4 = 36 * 2 * 32;                  // slots of one filter block (8 channels x 32 output channels): 36 pieces
constexpr int kRawBufs = 3, kFiltBufs = 2;            // raw patches are requested two sub-steps ahead (they come from HBM), filters one (L2)
constexpr int kRingFloats4 = (kRawBufs * kInSlots4 + kFiltBufs * kUSlots4) * 4;
constexpr int kXFloats4 = 4 * 4 * 4 * 64 * 4;          // hand-over area of the epilogue: [destination 4][source 4][4][64 lanes][4] floats (64 KiB), in the filter buffers
constexpr int kLdsFloats4 = (kRingFloats4 > kXFloats4 ? kRingFloats4 : kXFloats4);

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

// scalar forms of the 1-D data transform (bt6 above on one component): packed f32 instructions issued beside MFMAs cost several
// times their scalar pair on this part (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'), so the K loop works on scalars
template <int HALF>
__device__ __forceinline__ void bt6s(const float e0, const float e1, const float e2, const float e3, const float e4, float& o0, float& o1, float& o2) {
  if constexpr (HALF == 0) {
    const float a = __builtin_fmaf(-4.0f, e2, e4), b = __builtin_fmaf(-4.0f, e1, e3);
    o0 = __builtin_fmaf(4.0f, e0, __builtin_fmaf(-5.0f, e2, e4));
    o1 = a + b;
    o2 = a - b;
  } else {   // e0..e4 = d1..d5
    const float c = e3 - e1, e = e2 - e0;
    o0 = __builtin_fmaf(2.0f, e, c);
    o1 = __builtin_fmaf(-2.0f, e, c);
    o2 = __builtin_fmaf(4.0f, e0, __builtin_fmaf(-5.0f, e2, e4));
  }
}

// where the DMA of one output tile reads: descriptors of its image tensor and filter blocks, this lane's seven raw-piece offsets
struct Wino4Src {
  __amdgpu_buffer_rsrc_t rx, ru;
  unsigned u_off0;     // byte offset of the N tile's first filter block
  unsigned a_off[7];   // raw piece wave + 4 j: byte offset of this lane's pixel (chunk 0), or out of range
};

// raw piece j (0..6) of this wave, patch kr -> raw buffer rbuf; live false: the sub-step does not exist (zeros land in a buffer nobody reads)
__device__ __forceinline__ void wino4_dma_raw(float* smem, const Wino4Src& sr, int Cin, int j, int kr, int rbuf, bool live, int lane, int wave) {
  const int piece = wave + 4 * j;
  unsigned off = live ? sr.a_off[j] : kOobOffset;
  if (kr * 8 + 8 > Cin && (piece * 64 + lane) >= 16 * 51) off = kOobOffset;   // the chunk's upper half lies past Cin
  __builtin_amdgcn_raw_ptr_buffer_load_lds(sr.rx, (lds_ptr_t)(smem + rbuf * (kInSlots4 * 4) + piece * 256), 16, (int)off, (int)(kr * 32), 0, 0);
}
// filter piece 9 wave + j (j = 0..8) of block kf -> F[kf & 1]
__device__ __forceinline__ void wino4_dma_filt(float* smem, const Wino4Src& sr, int j, int kf, bool live, int lane, int wave) {
  const int piece = wave * 9 + j;
  const unsigned off = live ? (unsigned)((piece * 64 + lane) * 16) : kOobOffset;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(sr.ru, (lds_ptr_t)(smem + kRawBufs * (kInSlots4 * 4) + (kf & 1) * (kUSlots4 * 4) + piece * 256), 16, (int)off,
                                           (int)(sr.u_off0 + (unsigned)kf * (kUSlots4 * 16u)), 0, 0);
}

// the requests a tile starts with: raw patches 0, 1, 2 (-> R0, R1, R2) and / or filter block 0 (-> F0)
__device__ __forceinline__ void wino4_prologue_dma(float* smem, const Wino4Src& sr, int Cin, int K8, int lane, int wave, bool raw, bool filt) {
  if (raw) {
#pragma unroll
    for (int kr = 0; kr < 3; ++kr)
#pragma unroll
      for (int j = 0; j < 7; ++j) wino4_dma_raw(smem, sr, Cin, j, kr, kr, kr < K8, lane, wave);
  }
  if (filt) {
#pragma unroll
    for (int j = 0; j < 9; ++j) wino4_dma_filt(smem, sr, j, 0, true, lane, wave);
  }
}

// K loop of one wave (the only wave of its SIMD: nothing else hides its latencies, so the loop is software pipelined by hand).
// LDS: three raw-patch buffers R0..R2 and two filter buffers F0, F1.  Sub-step k multiplies V_k (registers) with the filters of F[k & 1];
// in the shadow of those 36 MFMAs the wave reads raw patch k + 1 from R[(k + 1) % 3] and transforms it into V_{k+1}, and issues the DMA of
// filters k + 1 (-> F[(k + 1) & 1], last read in sub-step k - 1; they come out of L2) and of raw patch k + 3 (-> R[k % 3], last read in
// sub-step k - 1; raw patches come from HBM / the Infinity Cache: with one sub-step of lead the loop stood waiting for them).  The
// sub-step ends with vmcnt(7) + barrier: the seven raw pieces just requested stay in flight, everything older has landed.
// The body is laid out as 36 slots of one MFMA + its share of the other work, pinned by sched_barrier:
//   slot s: MFMA of position j = s / 4 (row x = j / 3), channel pair s % 4; filter fragment j + 2 requested at slot 4 j + 3;
//   raw column b (5 pixels) requested in slots 4 b .. 4 b + 2, its vertical transform in slots 4 b + 4 .. 4 b + 7 (one component each);
//   horizontal transform of row 0 / row 1 of V_{k+1} in slots 24..27 / 28..31 (straight into the registers of V_k's rows, whose MFMAs
//   are done by then); row 2 follows at the top of the next sub-step (its MFMAs run last); DMA instructions in slots 0..15.
template <int PH, int PW>
__device__ __forceinline__ void wino4_loop(const IgemmArgs& p, float* smem, const Wino4Src& sr, bool prefetched, int lane, int wave, f32x16 (&acc)[3][3]) {
  const int K8 = p.kchunks;
  const int m = lane & 31, fh = lane >> 5;
  const int ty = m >> 4, tx = m & 15;
  constexpr int kRF = kInSlots4 * 4, kFF = kUSlots4 * 4;   // floats of a raw buffer / a filter buffer
  // this lane's raw reads: patch pixel (PH + a, PW + b), a, b < 5, of tile (ty, tx) sits at float offset rbase + ro(a, b) of a raw buffer
  const int rbase = ((fh * 48 + ty) * 17 + tx) * 4;
  auto ro = [](int a, int b) constexpr { return ((((PH + a) & 3) * 4 + ((PW + b) & 3)) * 51 + ((PH + a) >> 2) * 17 + ((PW + b) >> 2)) * 4; };
  const int fbase = kRawBufs * kRF + (fh * 32 + m) * 4;   // filter fragment of position pos: + (k & 1) * kFF + pos * 256
  constexpr int pos0 = (3 * PH) * 6 + 3 * PW;      // position (x, y) of this wave: pos0 + 6 x + y

  // DMA instruction q (0..15) of this wave in sub-step k: q < 9: filter piece q of block k + 1; else raw piece q - 9 of patch k + 3
  // (filters first: the sub-step's closing vmcnt(7) then covers them and leaves the raw pieces in flight)
  auto dma = [&](int q, int k, int rbuf, bool live_f, bool live_r) __attribute__((always_inline)) {
    if (q < 9) wino4_dma_filt(smem, sr, q, k + 1, live_f, lane, wave);
    else wino4_dma_raw(smem, sr, p.Cin, q - 9, k + 3, rbuf, live_r, lane, wave);
  };

  float t[3][5][4];   // vertical pass of the patch being transformed
  float v[3][3][4];   // V of the current sub-step (rows 0, 1: replaced in place by the next one's during the sub-step)
  f32x4 dcol[2][5];   // raw pixels of one patch column, double buffered by column parity
  auto vread = [&](const float* rp, int b, int a) __attribute__((always_inline)) { dcol[b & 1][a] = *reinterpret_cast<const f32x4*>(rp + ro(a, b)); };
  auto vpass = [&](int b, int c) __attribute__((always_inline)) {
    bt6s<PH>(dcol[b & 1][0][c], dcol[b & 1][1][c], dcol[b & 1][2][c], dcol[b & 1][3][c], dcol[b & 1][4][c], t[0][b][c], t[1][b][c], t[2][b][c]);
  };
  auto hpass = [&](int x, int c) __attribute__((always_inline)) {
    bt6s<PW>(t[x][0][c], t[x][1][c], t[x][2][c], t[x][3][c], t[x][4][c], v[x][0][c], v[x][1][c], v[x][2][c]);
  };

  // ---- prologue: filters 0 and patches 0, 1, 2 (requested by the previous tile's epilogue where there was one: then at least 16
  // younger vector-memory operations -- that tile's stores -- are in flight and need not be waited for); V_0 rows 0, 1 and the vertical
  // pass of row 2
  if (!prefetched) {
    wino4_prologue_dma(smem, sr, p.Cin, K8, lane, wave, true, true);
    __builtin_amdgcn_s_waitcnt(0xC07F & ~0xC00F);   // vmcnt(0)
  } else {
    __builtin_amdgcn_s_waitcnt((0xC07F & ~0xC00F) | 0x4000);   // vmcnt(16)
  }
  lds_barrier4();
  {
    const float* rp = smem + rbase;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
#pragma unroll
      for (int a = 0; a < 5; ++a) vread(rp, b, a);
#pragma unroll
      for (int c = 0; c < 4; ++c) vpass(b, c);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int c = 0; c < 4; ++c) hpass(x, c);
  }

  int r1 = 1, r0 = 0;   // raw buffer of patch k + 1 / of patch k (= the one patch k + 3 goes to)
  for (int k = 0; k < K8; ++k) {
    const float* rp = smem + rbase + r1 * kRF;                      // raw patch k + 1
    const float* fp = smem + fbase + (k & 1) * kFF + pos0 * 256;    // filters k
    const bool live1 = k + 1 < K8, live3 = k + 3 < K8;
    f32x4 uf[3];
    uf[0] = *reinterpret_cast<const f32x4*>(fp);
    uf[1] = *reinterpret_cast<const f32x4*>(fp + 1 * 256);
#pragma unroll
    for (int c = 0; c < 4; ++c) hpass(2, c);   // row 2 of V_k (its vertical pass was done during the previous sub-step)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      const int j = s >> 2, c = s & 3, x = j / 3, y = j - 3 * x;
      acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[j % 3][c], v[x][y][c], acc[x][y], 0, 0, 0);
      if (c == 3 && j + 2 < 9) uf[(j + 2) % 3] = *reinterpret_cast<const f32x4*>(fp + ((j + 2) / 3 * 6 + (j + 2) % 3) * 256);
      // raw column b: pixels 0, 1 requested at slot 4 b, 2, 3 at 4 b + 1, 4 at 4 b + 2 (into the buffer the vertical pass of column b - 2
      // finished with at slot 4 b - 1); vertical pass of column b at slots 4 b + 4 + c
      if (s < 20) {
        const int b = s >> 2;
        if ((s & 3) == 0) { vread(rp, b, 0); vread(rp, b, 1); }
        if ((s & 3) == 1) { vread(rp, b, 2); vread(rp, b, 3); }
        if ((s & 3) == 2) vread(rp, b, 4);
      }
      if (s >= 4 && s < 24) vpass((s - 4) >> 2, (s - 4) & 3);
      if (s >= 24 && s < 32) hpass((s - 24) >> 2, (s - 24) & 3);
      if (s < 16) dma(s, k, r0, live1, live3);
      __builtin_amdgcn_sched_barrier(0);
    }
    r0 = r1;
    r1 = r1 == 2 ? 0 : r1 + 1;
    // filters k + 1 (and everything older: patch k + 2) have landed; the seven raw pieces of patch k + 3 stay in flight
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0077);   // vmcnt(7) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  // (the dead tail requests -- zeros for sub-steps past the last -- that may still be in flight go to raw buffers; whatever is requested
  // into those next comes from the same wave and lands behind them)
}

// The accumulators live in the accumulation half of the register file; the output transform takes them out one register at a time
// (left to itself the compiler copies all 144 into vector registers at the loop exit and spills them to scratch)
__device__ __forceinline__ float acc_read(float v) {
  float o;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(o) : "a"(v));
  return o;
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

template <int PH, int PW, typename AfterHandover>
__device__ __forceinline__ void wino4_finish(const IgemmArgs& p, const Wino4Tile& tl, float* smem, const float* sV, int lane, f32x16 (&acc)[3][3],
                                             AfterHandover after_handover) {
  // This wave's partial sums of all 16 output pixels (a, b) of its tiles: row a = wave stays in registers (own[b][r]), the other three
  // rows go to the waves that finish them, through LDS, in four passes of 4 accumulator registers each (16-byte accesses):
  // sX[destination wave 4][source wave 4][b 4][64 lanes][4 registers] = 64 KiB, placed in the filter buffers (the raw buffers already
  // receive the next tile's first patches); the source = destination blocks are not used.
  constexpr int kMe = 2 * PH + PW;   // = wave
  float own[4][16];
  float* sX = smem + kRawBufs * kInSlots4 * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 part[4][4];   // [a][b]: registers 4 q .. 4 q + 3
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int r = 4 * q + r4;
      float sv[3][4];   // sv[y][a]: vertical output transform of position column y
#pragma unroll
      for (int y = 0; y < 3; ++y) at6<PH>(acc_read(acc[0][y][r]), acc_read(acc[1][y][r]), acc_read(acc[2][y][r]), sv[y]);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float yb[4];
        at6<PW>(sv[0][a], sv[1][a], sv[2][a], yb);
#pragma unroll
        for (int b = 0; b < 4; ++b) part[a][b][r4] = yb[b];
      }
      __builtin_amdgcn_sched_barrier(0);   // (one accumulator register at a time: the accumulators live in the other register file)
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (a == kMe) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) own[b][4 * q + r4] = part[a][b][r4];
        } else {
          *reinterpret_cast<f32x4*>(sX + ((((a * 4 + kMe) * 4 + b) * 64 + lane) * 4)) = part[a][b];
        }
      }
    lds_barrier4();
#pragma unroll
    for (int src = 0; src < 4; ++src) {   // fixed order of the sum: own part, then the other waves in ascending order
      if (src == kMe) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(sX + ((((kMe * 4 + src) * 4 + b) * 64 + lane) * 4));
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) own[b][4 * q + r4] += o[r4];
      }
    }
    lds_barrier4();   // (the next pass overwrites the area)
  }
  after_handover();   // (the filter buffers are free again: the next tile's first filter block is requested here)

  // ---- element-wise epilogue + stores (order of operations: epilogue_store of igemm_kernel.hpp).  This lane: tile (ty, tx), output
  // row `wave` of it, pixels b = 0..3, channels n0 + 8 g + 4 fh + e (register r = 4 g + e).  One channel group g at a time (keeps the
  // live registers of this part small: the accumulators still sit in the other half of the register file), the residual / mask /
  // accumulate operands of group g + 1 requested before group g is computed.
  const int f = p.flags;
  const bool has_res = (f & CRDR_EPI_RES) != 0, has_mask = (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) != 0;
  const bool do_cs = (f & CRDR_EPI_COLSUM) != 0, accum = (f & CRDR_EPI_ACCUM) != 0;
  const int m = lane & 31, fh = lane >> 5, ty = m >> 4, tx = m & 15;
  const int oy = tl.oh0 + 4 * ty + kMe, ox0 = tl.ow0 + 4 * tx;
  const bool row_ok = oy < p.OH;
  const unsigned pix0 = (unsigned)(((size_t)tl.n * p.OH + oy) * p.OW + ox0);
  auto tdesc = [&](const float* base, int ld) __attribute__((always_inline)) {
    const unsigned long long bytes = (((unsigned long long)p.N * p.OH * p.OW - 1) * ld + p.Cout) * 4ull;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (unsigned)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ry = tdesc(p.y, p.ldy), rr = tdesc(has_res ? p.res : p.y, p.ldres), rm = tdesc(has_mask ? p.mask : p.y, p.ldmask);
  unsigned okb = 0;   // bit b: pixel b of this lane's row lies inside the image
#pragma unroll
  for (int b2 = 0; b2 < 4; ++b2) okb |= (row_ok && ox0 + b2 < p.OW) ? (1u << b2) : 0u;
  auto off = [&](int ld, int b2, int g) __attribute__((always_inline)) {
    const int c = tl.n0 + 8 * g + 4 * fh;
    return (((okb >> b2) & 1u) && c < p.Cout) ? (unsigned)(((pix0 + b2) * ld + c) * 4) : kOobOffset;
  };
  f32x4 resv[2][4], mskv[2][4], oldv[2][4];
  auto request = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) {
      if (has_res) resv[g & 1][b2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off(p.ldres, b2, g), 0, 0));
      if (has_mask) mskv[g & 1][b2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, off(p.ldmask, b2, g), 0, 0));
      if (accum) oldv[g & 1][b2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, off(p.ldy, b2, g), 0, 0));
    }
  };
  float* sC = smem + kRawBufs * kInSlots4 * 4 + kUSlots4 * 4;   // column sums: [wave 4][which 2][16 r][64 lanes] = 32 KiB in filter buffer F1 (F0 receives the next tile's block 0)
  request(0);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g + 1 < 4) request(g + 1);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(sV + 0 * 32 + 8 * g + 4 * fh);
    const f32x4 vec2 = *reinterpret_cast<const f32x4*>(sV + 1 * 32 + 8 * g + 4 * fh);
    const f32x4 scale = *reinterpret_cast<const f32x4*>(sV + 2 * 32 + 8 * g + 4 * fh);
    const f32x4 shift = *reinterpret_cast<const f32x4*>(sV + 3 * 32 + 8 * g + 4 * fh);
    f32x4 cpre = {0.f, 0.f, 0.f, 0.f}, cpost = {0.f, 0.f, 0.f, 0.f};
    const bool c_ok[4] = {tl.n0 + 8 * g + 4 * fh + 0 < p.Cout, tl.n0 + 8 * g + 4 * fh + 1 < p.Cout, tl.n0 + 8 * g + 4 * fh + 2 < p.Cout,
                          tl.n0 + 8 * g + 4 * fh + 3 < p.Cout};
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) {
      f32x4 v = {own[b2][4 * g], own[b2][4 * g + 1], own[b2][4 * g + 2], own[b2][4 * g + 3]};
      const bool pix_ok = ((okb >> b2) & 1u) != 0;
      if (f & CRDR_EPI_BIAS) v += bias;
      if (f & CRDR_EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
      }
      if (f & CRDR_EPI_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.2f * v[e];
      }
      if (f & CRDR_EPI_VEC2) v += vec2;
      if (has_res) v += resv[g & 1][b2];
      if (f & CRDR_EPI_AFFINE) v = v * scale + shift;
      if (do_cs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cpre[e] += (pix_ok && c_ok[e]) ? v[e] : 0.f;
      }
      if (has_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float mv = mskv[g & 1][b2][e] - ((f & CRDR_EPI_MASKOFF) ? vec2[e] : 0.f);
          v[e] = mv > 0.0f ? v[e] : ((f & CRDR_EPI_LRELUMASK) ? 0.2f * v[e] : 0.0f);
        }
      }
      if (do_cs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cpost[e] += (pix_ok && c_ok[e]) ? v[e] : 0.f;
      }
      if (accum) v += oldv[g & 1][b2];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ry, off(p.ldy, b2, g), 0, 0);
    }
    if (do_cs) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sC[((kMe * 2 + 0) * 16 + 4 * g + e) * 64 + lane] = cpre[e];
        sC[((kMe * 2 + 1) * 16 + 4 * g + e) * 64 + lane] = cpost[e];
      }
    }
  }
  if (do_cs) {
    // column sums of the tile: lane (tile m, half fh) held 16 channels; fixed-order sum over the 32 tiles of a half and the 4 waves
    lds_barrier4();
    const int tid = kMe * 64 + lane;
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;          // channel c = 8 g + 4 fh + e
      const int g = c >> 3, h2 = (c >> 2) & 1, e = c & 3, r = 4 * g + e;
      float v = 0.f;
      for (int w2 = 0; w2 < 4; ++w2)
        for (int t2 = 0; t2 < 32; ++t2) v += sC[((w2 * 2 + which) * 16 + r) * 64 + h2 * 32 + t2];
      if (tl.n0 + c < p.Cout) p.cs[((size_t)tl.patch * 2 + which) * p.cs_ld + tl.n0 + c] = v;
    }
  }
}

// DMA sources of tile tl (group pointers resolved)
__device__ __forceinline__ Wino4Src wino4_src(const IgemmArgs& p, const IgemmGroup& grp, const Wino4Tile& tl, int gyn, int lane, int wave) {
  Wino4Src sr;
  const float* x = p.ngroup > 1 ? grp.x[tl.gidx] : p.x;
  sr.rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((((unsigned long long)p.N * p.H * p.W - 1) * p.ldx + p.Cin) * 4ull), 0x00020000);
  // transformed filters of group gidx: [N tile][chunk][2304 slots of 16 B]
  const size_t ublock = (size_t)gyn * p.kchunks * kUSlots4 * 4;   // floats per group
  sr.ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w) + (size_t)tl.gidx * ublock, 0, (unsigned)(ublock * 4), 0x00020000);
  sr.u_off0 = (unsigned)(tl.n0 / 32) * (unsigned)p.kchunks * (kUSlots4 * 16u);
#pragma unroll
  for (int j = 0; j < 7; ++j) sr.a_off[j] = wino4_in_off(p, tl, wave + 4 * j, lane);
  return sr;
}

// per-column epilogue vectors of tile tl -> sV[4][32] (bias, vec2, scale, shift)
__device__ __forceinline__ void wino4_vectors(const IgemmArgs& p, const IgemmGroup& grp, const Wino4Tile& tl, float* sV, int tid) {
  if (tid < 32) {
    const int f0 = p.flags;
    const bool live = tl.n0 + tid < p.Cout;
    const float* bias = p.ngroup > 1 ? grp.bias[tl.gidx] : p.bias;
    sV[0 * 32 + tid] = (live && (f0 & CRDR_EPI_BIAS)) ? bias[tl.n0 + tid] : 0.f;
    sV[1 * 32 + tid] = (live && (f0 & (CRDR_EPI_VEC2 | CRDR_EPI_MASKOFF))) ? p.vec2[tl.n0 + tid] : 0.f;
    sV[2 * 32 + tid] = (live && (f0 & CRDR_EPI_AFFINE)) ? p.scale[tl.n0 + tid] : 1.f;
    sV[3 * 32 + tid] = (live && (f0 & CRDR_EPI_AFFINE)) ? p.shift[tl.n0 + tid] : 0.f;
  }
}

// Persistent: at most one workgroup per CU, each walks the tiles vb = blockIdx.x, + gridDim.x, ...  Between the K loop and the
// epilogue of a tile the waves request the next tile's first raw patches (into the raw buffers, free by then) and its epilogue vectors,
// after the hand-over its first filter block: the DMA latency of a fresh tile and the memory latency of the stores hide behind each other.
__global__ __launch_bounds__(kNT4) void wino4_kernel(const IgemmArgs p_, const IgemmGroup grp, int gx, int gyn, int gz) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane_ = threadIdx.x & 63, wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int total = gx * gyn * gz;
  float* sVb = smem + kLdsFloats4;   // [2][4][32]: bias, vec2, scale, shift of the current / the next tile
  int cur = 0;
  bool prefetched = false;
  Wino4Src sr;
  for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
    // (per-lane constants are re-derived per tile instead of staying live across the register-hungry epilogue)
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(wave));
    const int tid = wave * 64 + lane;
    const Wino4Tile tl = wino4_tile(p_, vb, gx, gyn, gz);
    IgemmArgs p = p_;
    if (p.ngroup > 1) {
      const int g = tl.gidx;
      p.x = grp.x[g]; p.y = grp.y[g]; p.bias = grp.bias[g]; p.mask = grp.mask[g]; p.res = grp.res[g]; p.cs = grp.cs[g];
    }
    float* sV = sVb + cur * 128;
    if (!prefetched) {
      sr = wino4_src(p_, grp, tl, gyn, lane, wave);
      wino4_vectors(p_, grp, tl, sV, tid);   // (published by the K loop's first barrier)
    }

    f32x16 acc[3][3];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 3; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    const int role = wave;
#ifndef W4_SKIP_LOOP
    if (role == 0) wino4_loop<0, 0>(p, smem, sr, prefetched, lane, wave, acc);
    else if (role == 1) wino4_loop<0, 1>(p, smem, sr, prefetched, lane, wave, acc);
    else if (role == 2) wino4_loop<1, 0>(p, smem, sr, prefetched, lane, wave, acc);
    else wino4_loop<1, 1>(p, smem, sr, prefetched, lane, wave, acc);
#endif
    // (the loop ends with a barrier: every wave is past its last LDS read, raw and filter buffers are free)
    const bool more = vb + (int)gridDim.x < total;
    if (more) {   // the next tile: raw patches 0, 1, 2 and the epilogue vectors now, filter block 0 after the hand-over
      const Wino4Tile tn = wino4_tile(p_, vb + (int)gridDim.x, gx, gyn, gz);
      sr = wino4_src(p_, grp, tn, gyn, lane, wave);
      wino4_prologue_dma(smem, sr, p_.Cin, p_.kchunks, lane, wave, true, false);
      wino4_vectors(p_, grp, tn, sVb + (cur ^ 1) * 128, tid);
    }
    auto after = [&]() __attribute__((always_inline)) {
      if (more) wino4_prologue_dma(smem, sr, p_.Cin, p_.kchunks, lane, wave, false, true);
    };
#ifdef W4_SAMEROLE
    wino4_finish<0, 0>(p, tl, smem, sV, lane, acc, after);
#elif !defined(W4_SKIP_FINISH)
    if (role == 0) wino4_finish<0, 0>(p, tl, smem, sV, lane, acc, after);
    else if (role == 1) wino4_finish<0, 1>(p, tl, smem, sV, lane, acc, after);
    else if (role == 2) wino4_finish<1, 0>(p, tl, smem, sV, lane, acc, after);
    else wino4_finish<1, 1>(p, tl, smem, sV, lane, acc, after);
#else
    after();
    if (acc[0][0][0] == 1.2345f && acc[1][1][3] == 2.5f && acc[2][2][7] == 0.3f) p.y[lane] = acc[0][1][1];
#endif
    prefetched = more;
    cur ^= 1;
    lds_barrier4();   // column-sum area, sV of this tile: free (the stores stay in flight: the next tile's first wait is vmcnt(16))
  }
}

// Filter transform U = G g G^T, G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], evaluated in double
// and rounded once, from the implicit-GEMM weight pack (tap-major [tap][wrows][wcols]) into the stage-block layout of wino4_kernel:
// [N tile of 32][chunk of 8 channels][position 36][h 2][oc 32][4].  One thread per (N tile, chunk, h, oc).
struct Wino4Taps { int widx[9]; };
__global__ void wino4_filter_kernel(const IgemmGroup grp, int ngroup, const float* w0, float* u, int Cin, int Cout, int wrows, int wcols, int kchunks,
                                    int ntile, Wino4Taps tp) {
  // one thread per (N tile, chunk, h, oc, channel e of the half): consecutive threads read consecutive input channels of one weight-pack
  // row and write consecutive floats of a 16-byte slot
  const long long total = (long long)ntile * kchunks * 256;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int g = blockIdx.y;
  const float* w = ngroup > 1 ? grp.w[g] : w0;
  const int e = (int)(id & 3), oc32 = (int)((id >> 2) & 31), h = (int)((id >> 7) & 1);
  const long long blk = id >> 8;   // (N tile, chunk)
  const int kc = (int)(blk % kchunks), ct = (int)(blk / kchunks);
  const int oc = ct * 32 + oc32, c = kc * 8 + h * 4 + e;
  const bool live = oc < Cout && c < Cin;
  double g9[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g9[a][b] = live ? (double)w[((size_t)tp.widx[a * 3 + b] * wrows + oc) * wcols + c] : 0.0;
  const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  float* dst = u + ((size_t)g * ntile * kchunks + (size_t)blk) * (kUSlots4 * 4) + ((size_t)h * 32 + oc32) * 4 + e;
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    double t[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) t[b] = G[xi][0] * g9[0][b] + G[xi][1] * g9[1][b] + G[xi][2] * g9[2][b];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) dst[(size_t)(xi * 6 + nu) * 256] = (float)(G[nu][0] * t[0] + G[nu][1] * t[1] + G[nu][2] * t[2]);
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
    const long long total = (long long)ntile * kchunks * 256;
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
  const size_t lds = (size_t)(kLdsFloats4 + 2 * 4 * 32) * sizeof(float);
  hipLaunchKernelGGL(wino4_kernel, dim3(std::min(total, ncu)), dim3(kNT4), lds, s, a, grp, gx, ntile, G);
  CRDR_CHECK_LAUNCH("wino4_kernel");
  return 0;
}

}  // namespace crdr
