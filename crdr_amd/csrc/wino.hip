// Winograd F(2x2, 3x3) convolution on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) for 3x3 stride-1 layers and their
// input gradients: Y = A^T [ (G g G^T) . (B^T d B) ] A  (Lavin & Gray 2016, the minimal-filtering form cuDNN's
// CUDNN_CONVOLUTION_*_ALGO_WINOGRAD uses, which the reference reaches through cudnn.benchmark, base_trainer.py:20).
// 16 element-wise products per 2x2 outputs and channel instead of 36: 2.25x fewer MFMAs than the implicit GEMM.  The data
// transform has +-1 coefficients only, the filter transform halves (exact), so the arithmetic stays fp32 throughout; the
// association of the sums differs from the direct form (typical deviation a few 1e-7 of the output scale, tests/test_gpu_conv.py).
//
// A 5x5 stride-1 kernel (Charm transforms) runs as 2 x 2 sub-filters of 3x3 (the outer ones zero padded), each reading the patch 3
// pixels further down / right: 4 x 16 = 64 products per 2x2 outputs instead of 100, same transforms, the sub-filters are just more
// reduction steps.  (Measured: not faster than the implicit GEMM on the 16x16 Charm shapes -- few patches, 2.56x larger filter
// blocks -- so the tuner rarely takes it there; tools/bench_wino.py --k5.)
//
// One output tile = 16 x 16 output pixels (8 x 8 Winograd tiles) of one image x 64 output channels, computed by 8 waves:
//   wave (ph, wn, wm): transform rows xi in {2 ph, 2 ph + 1} (8 of the 16 positions), tile columns 4 wm .. 4 wm + 3 (8 rows x 4
//   columns = 32 tiles = the 32 MFMA rows), output channels 32 wn .. 32 wn + 31: 8 accumulator blocks of 32 x 32, two waves per SIMD.
// The launch is persistent (one workgroup per CU walks the tiles in an XCD-aware order); before the epilogue of a tile the
// waves already issue the first sub-step of the next one.
// K loop: sub-steps of 8 input channels (of one sub-filter), one barrier each, operands double buffered in LDS and filled by LDS-DMA:
//   * the raw 18 x 18 x 8 input patch, stored by pixel parity class so that the 32 tiles' reads of patch pixel (i, j) are
//     consecutive 16-B slots: [half h = channels 4h..4h+3][class (i&1, j&1)][9 rows][12 slots (9 used)]; with 8 x 4 tiles per
//     wave a row pitch of 12 slots makes every 16-lane group of a ds_read_b128 hit 16 different slots mod 16 (conflict free);
//     pixels outside the image (padding) and channels past Cin read zeros through the buffer range check;
//   * the transformed filters of the sub-step, one contiguous 32 KiB block in memory and in LDS: [position 16][h][oc 64][4].
//   Every wave builds B^T d B for its 8 positions in registers from 12 ds_read_b128 (lane = tile, half-wave = channel half, the
//   operand layout of the implicit-GEMM kernel), reads 8 filter fragments and issues 32 MFMAs per sub-step.
//   (Measured alternatives at the same speed, +-1 %: 32-channel input chunks staged as full 128-B lines with an XOR swizzle; the
//   two waves of a SIMD half a sub-step apart, one loading while the other multiplies; all DMA issued between the MFMAs of one role.)
// Epilogue: the two waves of a (wm, wn) pair each hold half of the xi sum; each forms its part of A^T M A for both output rows,
//   hands the part of the partner's row over through LDS and finishes its own row (ph = output row inside the tile), then runs
//   the element-wise epilogue of the implicit-GEMM kernel (same order of operations) with dword buffer accesses whose per-element
//   offset is a scalar: 32 lanes = 32 consecutive channels of one pixel.
#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "igemm_args.hpp"
#include "wino.hpp"

namespace crdr {

namespace {

constexpr int kInUsed = 2 * 4 * 9 * 12;     // 16-byte slots of the input patch image per stage: [h][class][9 rows][12 slots, 9 used] ...
constexpr int kInSlots = 1024;              // ... rounded up to 4 DMA instructions for each of the 4 issuing waves
constexpr int kUSlots = 16 * 2 * 64;        // of one filter block (8 channels): 8 DMA instructions per issuing wave
constexpr int kInFloats = kInSlots * 4, kUFloats = kUSlots * 4;
// stage = [input patch A][input patch B (kernel variant with pair tiles only)][filters]
template <bool PAIRS> struct WinoLds {
  static constexpr int kNIn = PAIRS ? 2 : 1;
  static constexpr int kStageFloats = kNIn * kInFloats + kUFloats;
  static constexpr int kXFloats = 8 * 32 * 64;       // epilogue hand-over area, placed behind stage 0 (which then already receives the next tile)
  static constexpr int kTileFloats = kStageFloats + kXFloats > 2 * kStageFloats ? kStageFloats + kXFloats : 2 * kStageFloats;
};

// workgroup barrier that publishes LDS writes but does not wait for DMA / global stores in flight
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) (vmcnt, expcnt untouched)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
constexpr int kNT = 512;

// in-place 1-D data transform of four f32x4 (B^T rows applied along one axis): (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
__device__ __forceinline__ void bt4(const f32x4 d0, const f32x4 d1, const f32x4 d2, const f32x4 d3, f32x4 (&o)[4]) {
  o[0] = d0 - d2; o[1] = d1 + d2; o[2] = d2 - d1; o[3] = d1 - d3;
}

// K loop of one wave.  The two waves of a SIMD (ROLE 0: waves 0-3, ROLE 1: waves 4-7) run half a sub-step apart -- while one
// reads its raw pixels and filter fragments from LDS and transforms the pixels (latency-bound LDS / VALU work), the other issues
// its 32 MFMAs from registers -- so the matrix pipe of the SIMD always has one wave feeding it.  Two barriers per sub-step k
// (8 channels): A(k) (its operands have landed) and B(k); phase [A(k), B(k)): role 0 loads k, role 1 multiplies k - 1; phase
// [B(k), A(k + 1)): role 0 multiplies k, role 1 loads k.  The role-1 waves issue ALL the DMA of sub-step k + 1 between their
// MFMAs of phase [A(k), B(k)) (where an issue slot is nearly free) and drain it before A(k + 1); the stage it overwrites was last
// read in phase [B(k - 1), A(k)).  B barriers hand no data over: a bare s_barrier that does not drain the DMA in flight.
// what the DMA of one output tile needs: descriptors of its image and filter blocks, and this lane's two input slots
struct WinoIO {
  __amdgpu_buffer_rsrc_t rx, ru;
  unsigned a_off[2][2];  // [patch A / B][j]: byte offset of slot j's pixel (sub-filter 0, chunk 0) in the tensor; may wrap where the pixel is outside: used only where ok
  unsigned okmask;       // bits 8 patch + 2 sub + j: slot j holds a pixel of the image for sub-filter sub; bits 16, 17: its channel half h; bit 18: pair tile
  unsigned u_off0;       // byte offset of the N tile's first filter block
};

// the DMA of sub-step k8, this wave's share: input instructions 2 wave + j (slots (2 wave + j) * 64 + lane), filter instructions 4 wave + j
template <bool PAIRS>
__device__ __forceinline__ void wino_issue(const IgemmArgs& p, float* smem, const WinoIO& io, int k8, int sub, int kc, int lane, int wave) {
  constexpr int kStageFloats = WinoLds<PAIRS>::kStageFloats, kNIn = WinoLds<PAIRS>::kNIn;
  float* st = smem + (k8 & 1) * kStageFloats;
  // k8 = sub * kchunks + kc (the caller keeps both counters: no division in the loop).  Sub-filter (sa, sb) of a 5x5 kernel
  // (2 x 2 of them) reads the patch 3 sa rows / 3 sb columns further down / right
  const int sa = sub >> 1, sb = sub & 1;
  // Every non-MFMA instruction costs matrix time, so the per-piece address work is kept minimal: the chunk / sub-filter
  // displacement is uniform and travels in the instruction's SCALAR offset, a slot outside the image has an out-of-range lane
  // offset for the whole tile (io.a_off, set up once per tile and sub-filter), and the channel-tail test only runs in the last chunk.
  const unsigned dsp = (unsigned)(((3 * sa * p.W + 3 * sb) * p.ldx) * 4), dch = (unsigned)(kc * 32);
  const bool tail = kc * 8 + 8 > p.Cin;   // (uniform) the chunk's upper half lies past Cin
  if (p.nphase == 1 && !tail) {   // (uniform) the common case: nothing to decide per lane
#pragma unroll
    for (int pb = 0; pb < kNIn; ++pb) {
      if (pb == 1 && !((io.okmask >> 18) & 1u)) break;   // (uniform) the second patch of a pair tile
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(io.rx, (lds_ptr_t)(st + pb * kInFloats + ((2 * wave + j) * 64) * 4), 16, (int)io.a_off[pb][j], (int)dch, 0, 0);
    }
  } else {
#pragma unroll
    for (int pb = 0; pb < kNIn; ++pb) {
      if (pb == 1 && !((io.okmask >> 18) & 1u)) break;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // (several sub-filters: a_off may have wrapped below zero for a slot that only a displaced sub-filter brings into the
        // image, so the displacement is added in the lane offset there; the range check does not see the scalar offset wrap)
        unsigned off = p.nphase == 1 ? io.a_off[pb][j] : (((io.okmask >> (8 * pb + 2 * sub + j)) & 1u) ? io.a_off[pb][j] + dsp : kOobOffset);
        if (tail && ((io.okmask >> (16 + j)) & 1u)) off = kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(io.rx, (lds_ptr_t)(st + pb * kInFloats + ((2 * wave + j) * 64) * 4), 16, (int)off, (int)dch, 0, 0);
      }
    }
  }
  const unsigned ub = io.u_off0 + (unsigned)k8 * (kUSlots * 16u);
  const unsigned ul = (unsigned)(4 * wave * 64 + lane) * 16u;
  // four consecutive KiB: the instruction's immediate offset (a literal) advances the global AND the LDS address, so one M0 serves all four
  float* su = st + kNIn * kInFloats + (4 * wave * 64) * 4;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(io.ru, (lds_ptr_t)su, 16, (int)ul, (int)ub, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(io.ru, (lds_ptr_t)su, 16, (int)ul, (int)ub, 1024, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(io.ru, (lds_ptr_t)su, 16, (int)ul, (int)ub, 2048, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(io.ru, (lds_ptr_t)su, 16, (int)ul, (int)ub, 3072, 0);
  __builtin_amdgcn_sched_barrier(0);
}

// K loop of one wave: one barrier per sub-step k (8 channels of one sub-filter); the DMA of sub-step k + 1 is issued first, then
// the wave reads its 12 raw pixels, transforms them, and issues its 32 MFMAs, reading one filter fragment per position.
template <int PH, bool PAIRS>
__device__ __forceinline__ void wino_loop(const IgemmArgs& p, float* smem, const WinoIO& io, bool stage0_issued,
                                          int lane, int wave, int wm, int wn, f32x16 (&acc)[8]) {
  const int K8 = p.kchunks * p.nphase;   // sub-steps: (sub-filter, 8-channel chunk), sub-filter outermost
  const int m = lane & 31, fh = lane >> 5;
  const int ty = m >> 2, tx = (m & 3) + 4 * wm;
  constexpr int kStageFloats = WinoLds<PAIRS>::kStageFloats, kNIn = WinoLds<PAIRS>::kNIn;
  const bool pair = PAIRS && ((io.okmask >> 18) & 1u);
  // float offsets of this lane's 12 raw reads (rows PH .. PH + 2 of the 4 x 4 patch, all 4 columns) inside a stage
  int ro[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = PH + a;
      const int cls = (i & 1) * 2 + (j & 1);
      ro[a][j] = (pair && wn ? kInFloats : 0) + (fh * 432 + cls * 108 + (ty + (i >> 1)) * 12 + tx + (j >> 1)) * 4;
    }
  // position p = (2 PH + a) * 4 + nu: + (a * 4 + nu) * 512; in a pair tile both wn take the tile's first 32 channels (of two patches)
  const int bo = kNIn * kInFloats + ((2 * PH * 4 * 2 + fh) * 64 + (pair ? 0 : wn * 32) + m) * 4;

  // One sub-step.  Every instruction beside the MFMAs costs matrix time, hazard no-ops and waits included, so the
  // three parts are kept apart: all LDS reads of the first half, the whole data transform, then the 32 MFMAs back to back with
  // the second half of the filter fragments read behind the first MFMAs.
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float* st = smem + buf * kStageFloats;
    f32x4 d[3][4], bfa[4], bfb[4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d[a][j] = *reinterpret_cast<const f32x4*>(st + ro[a][j]);
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) bfa[q] = *reinterpret_cast<const f32x4*>(st + bo + q * 512);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 t[2][4], v[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (PH == 0) { t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; }   // xi = 0, 1 from rows 0, 1, 2
      else { t[0][j] = d[1][j] - d[0][j]; t[1][j] = d[0][j] - d[2][j]; }                     // xi = 2, 3 from rows 1, 2, 3
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) bt4(t[a][0], t[a][1], t[a][2], t[a][3], v[a]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) bfb[q] = *reinterpret_cast<const f32x4*>(st + bo + (4 + q) * 512);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0][q][s], bfa[q][s], acc[q], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1][q][s], bfb[q][s], acc[4 + q], 0, 0, 0);
  };

  if (!stage0_issued) wino_issue<PAIRS>(p, smem, io, 0, 0, 0, lane, wave);
  __syncthreads();
  int sub = 0, kc = 0;   // of sub-step k8 + 1
  for (int k8 = 0; k8 < K8; ++k8) {
    if (++kc == p.kchunks) { kc = 0; ++sub; }
    if (k8 + 1 < K8) wino_issue<PAIRS>(p, smem, io, k8 + 1, sub, kc, lane, wave);
    compute(k8 & 1);
    __syncthreads();
  }
}

// where an output tile (virtual block id vb) lies.  A group's tiles: gx patches x gyn N tiles of 64 channels, then -- when the
// channel count leaves a tail of <= 32 -- npair PAIR tiles: the tail's 32 channels for TWO patches (2 i, 2 i + 1), the wn waves
// taking one patch each instead of one 32-channel half each, so that a 96-channel layer costs 1.5 tiles per patch, not 2.
struct WinoTile { int patch, tile_n, gidx, n, oh0, ow0, n0; int pair, nB, oh0B, ow0B, patchB; };
__device__ __forceinline__ WinoTile wino_tile(const IgemmArgs& p, int vb, int gx, int gyn, int npair, int gz) {
  // XCD-aware order (see igemm_kernel): the hardware deals workgroups round-robin over the 8 XCDs; every XCD walks a contiguous
  // range of (patch, N tile) pairs, the N tiles of a patch back to back on one L2.  The persistent grid is a multiple of 8
  // workgroups, so vb & 7 is still the XCD of the workgroup that runs tile vb.
  WinoTile t;
  const int T = gx * gyn + npair, nwg = T * gz, cpx = nwg >> 3;
  const int q = vb < cpx * 8 ? (vb & 7) * cpx + (vb >> 3) : vb;
  t.gidx = q / T;
  const int r = q - t.gidx * T;
  const int ppi = p.GH * p.GW;
  t.pair = r >= gx * gyn;
  if (t.pair) {
    t.patch = 2 * (r - gx * gyn);
    t.tile_n = gyn;
  } else if (p.m_inner) {   // filters dominate the traffic (Charm hoists): the patches of one N tile run together and share its filter blocks in L2
    t.patch = r % gx;
    t.tile_n = r / gx;
  } else {
    t.tile_n = r % gyn;
    t.patch = r / gyn;
  }
  t.n = t.patch / ppi;
  const int prem = t.patch - t.n * ppi, by = prem / p.GW, bx = prem - by * p.GW;
  t.oh0 = by * 16; t.ow0 = bx * 16; t.n0 = t.tile_n * 64;
  t.patchB = -1; t.nB = 0; t.oh0B = 0; t.ow0B = 0;
  if (t.pair) {
    const int pb = t.patch + 1;
    t.nB = pb / ppi;
    const int premB = pb - t.nB * ppi, byB = premB / p.GW, bxB = premB - byB * p.GW;
    t.oh0B = byB * 16; t.ow0B = bxB * 16;
    t.patchB = pb < gx ? pb : -1;   // no second patch (odd patch count): every access of it falls out of range
  }
  return t;
}

template <bool PAIRS>
__device__ __forceinline__ WinoIO wino_io(const IgemmArgs& p, const IgemmGroup& grp, const WinoTile& t, int gy, int wave, int lane) {
  WinoIO io;
  const int H = p.H, W = p.W, ldx = p.ldx;
  const float* x = p.ngroup > 1 ? grp.x[t.gidx] : p.x;
  io.rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((((unsigned long long)p.N * H * W - 1) * ldx + p.Cin) * 4ull), 0x00020000);
  // transformed filters of group gidx: [N tile][sub-filter][chunk][2048 slots of 16 B]
  const size_t ublock = (size_t)gy * p.nphase * p.kchunks * kUSlots * 4;   // floats per group
  io.ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w) + (size_t)t.gidx * ublock, 0, (unsigned)(ublock * 4), 0x00020000);
  io.u_off0 = (unsigned)t.tile_n * (unsigned)(p.nphase * p.kchunks) * (kUSlots * 16u);
  // input staging: DMA instruction j of wave w fills slots (2 w + j) * 64 + lane -> (h, class, r, c) -> patch pixel
  // (2 r + pi, 2 c + pj), channels 4h .. 4h + 3 of the sub-step
  io.okmask = t.pair ? (1u << 18) : 0u;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int S = (2 * wave + j) * 64 + lane;
    const int h = S / 432, rem = S - h * 432, cls = rem / 108, r2 = rem - cls * 108, r = r2 / 12, c = r2 - r * 12;
    io.okmask |= (h ? 1u : 0u) << (16 + j);
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      if (pb == 1 && (!PAIRS || !t.pair)) { io.a_off[1][j] = 0; break; }   // (uniform)
      const int nn = pb ? t.nB : t.n, ih0 = (pb ? t.oh0B : t.oh0) - p.si, iw0 = (pb ? t.ow0B : t.ow0) - p.si;
      const bool live = pb ? t.patchB >= 0 : true;
      const int ih = ih0 + 2 * r + (cls >> 1), iw = iw0 + 2 * c + (cls & 1);
      for (int sub = 0; sub < p.nphase; ++sub) {
        const int sa = sub >> 1, sb = sub & 1;
        const bool ok = live && S < kInUsed && c < 9 && (unsigned)(ih + 3 * sa) < (unsigned)H && (unsigned)(iw + 3 * sb) < (unsigned)W;
        io.okmask |= (ok ? 1u : 0u) << (8 * pb + 2 * sub + j);
      }
      io.a_off[pb][j] = (unsigned)((((nn * H + ih) * W + iw) * ldx + 4 * h) * 4);
      if (p.nphase == 1 && !((io.okmask >> (8 * pb + j)) & 1u)) io.a_off[pb][j] = kOobOffset;   // (one sub-filter: decided once per tile)
    }
  }
  return io;
}

// Everything after the K loop of one wave, compiled per PH (no per-element selects on the wave's role) and organised as one pass
// over the wave's 32 outputs per epilogue flag (a flag is tested once per tile, not once per element): every instruction here is
// matrix time lost, and this part used to be a third of a 16-sub-step tile.
//   Output transform: this wave holds M[xi][nu] for xi = 2 PH, 2 PH + 1.  Row sums of A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]:
//   PH 0: s0 = M0 + M1, s1 = M1;   PH 1: s0 = M2, s1 = -M2 - M3;   then along nu: (s[0] + s[1] + s[2], s[1] - s[2] - s[3]).
//   The wave finishes output row i = PH of its tiles and hands its part of row 1 - PH to the partner wave through LDS.
template <int PH>
__device__ __forceinline__ void wino_finish(const IgemmArgs& p, bool pair, int patch, int n, int oh0, int ow0, int n0, float* sXb, const float* sV,
                                            int lane, int wave, int wm, int wn, f32x16 (&acc)[8], float& cpre, float& cpost) {
  float own[32], give[32];   // index jj * 16 + r: column jj of the tile of accumulator register r
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s0[4], s1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const float ma = acc[nu][r], mb = acc[4 + nu][r];
      if constexpr (PH == 0) { s0[nu] = ma + mb; s1[nu] = mb; }
      else { s0[nu] = ma; s1[nu] = -ma - mb; }
    }
    const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
    const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
    if constexpr (PH == 0) { own[r] = y00; own[16 + r] = y01; give[r] = y10; give[16 + r] = y11; }
    else { own[r] = y10; own[16 + r] = y11; give[r] = y00; give[16 + r] = y01; }
  }
  float* sX = sXb + wave * (32 * 64);   // [32][64 lanes]
#pragma unroll
  for (int q = 0; q < 32; ++q) sX[q * 64 + lane] = give[q];
  lds_barrier();   // (not __syncthreads: the next tile's DMA stays in flight)
  const float* sP = sXb + (wave ^ 1) * (32 * 64);
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const float o = sP[q * 64 + lane];
    own[q] = PH == 0 ? own[q] + o : o + own[q];   // (part of xi 0, 1) + (part of xi 2, 3)
  }

  // ---- element-wise epilogue + stores (order of operations: epilogue_store of igemm_kernel.hpp).
  // This lane's 32 outputs: channel cn of pixels (oh0 + 4 (r >> 2) + 2 fh + PH, ow0 + 2 (r & 3) + 8 wm + jj), r < 16, jj < 2.  The descriptors
  // are based at the tile's first pixel and channel; the per-lane part of the offset is fixed and the (r, jj) part is uniform, so
  // it travels in the instruction's scalar offset: no per-element address arithmetic.  Pixels past the image / channels past Cout
  // get an out-of-range lane offset (rows past OH fall off the end of the image-sized descriptor by themselves).
  const int f = p.flags;
  const bool has_res = (f & CRDR_EPI_RES) != 0, has_mask = (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) != 0;
  const bool do_cs = (f & CRDR_EPI_COLSUM) != 0, accum = (f & CRDR_EPI_ACCUM) != 0;
  const int fh = lane >> 5, cn = (pair ? 0 : wn * 32) + (lane & 31);
  const bool oc_ok = n0 + cn < p.Cout;
  const size_t pix0 = ((size_t)n * p.OH + oh0) * p.OW + ow0;             // first pixel of the tile
  const unsigned rows_left = (unsigned)(p.OH - oh0);                     // image rows from the tile's first one
  auto desc = [&](const float* base, int ld) __attribute__((always_inline)) {
    const unsigned long long bytes = patch < 0 ? 0ull : ((unsigned long long)rows_left * p.OW - ow0) * (unsigned long long)ld * 4ull;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + pix0 * ld + n0, 0, (unsigned)(bytes < 0x7fffffffull ? bytes : 0x7fffffffull), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ry = desc(p.y, p.ldy);
  const int lrow = 2 * fh + PH, lcol = 8 * wm;                           // lane part of the pixel position inside the tile
  const bool full_w = ow0 + 16 <= p.OW;                                  // (uniform) no column of the tile hangs over the image
  // element q = jj * 16 + r sits 4 (r >> 2) rows down and 2 (r & 3) + jj columns right of the lane's first pixel
  auto pixs = [&](int q) __attribute__((always_inline)) { return 4 * ((q & 15) >> 2) * p.OW + 2 * (q & 3) + (q >> 4); };
  // lane offsets per element: the fixed lane part, or out of range (a column past the image would alias the next row: only tiles
  // on the right edge of a ragged image test per element)
  auto lane_off = [&](int ld, int q) __attribute__((always_inline)) {
    const unsigned v = oc_ok ? (unsigned)(((lrow * p.OW + lcol) * ld + cn) * 4) : kOobOffset;
    return (full_w || ow0 + lcol + 2 * (q & 3) + (q >> 4) < p.OW) ? v : kOobOffset;
  };
  const float bias = sV[0 * 64 + cn], vec2 = sV[1 * 64 + cn], scale = sV[2 * 64 + cn], shift = sV[3 * 64 + cn];
  // (rows past OH / columns past OW / channels past Cout must not enter the column sums; everything else about them is harmless)
  auto live = [&](int q) __attribute__((always_inline)) {
    return oc_ok && (full_w || ow0 + lcol + 2 * (q & 3) + (q >> 4) < p.OW) && oh0 + 4 * ((q & 15) >> 2) + lrow < p.OH;
  };
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {   // the two tile columns jj, 16 outputs each (keeps the operand arrays small)
    float* o = own + hb * 16;
    float resv[16], mskv[16];
    if (has_res) {
      const __amdgpu_buffer_rsrc_t rr = desc(p.res, p.ldres);
#pragma unroll
      for (int r = 0; r < 16; ++r) resv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off(p.ldres, hb * 16 + r), pixs(hb * 16 + r) * p.ldres * 4, 0));
    }
    if (has_mask) {
      const __amdgpu_buffer_rsrc_t rm = desc(p.mask, p.ldmask);
#pragma unroll
      for (int r = 0; r < 16; ++r) mskv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, lane_off(p.ldmask, hb * 16 + r), pixs(hb * 16 + r) * p.ldmask * 4, 0));
    }
    if (f & CRDR_EPI_BIAS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] += bias;
    }
    if (f & CRDR_EPI_RELU) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = fmaxf(o[r], 0.0f);
    }
    if (f & CRDR_EPI_LRELU) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = o[r] > 0.0f ? o[r] : 0.2f * o[r];
    }
    if (f & CRDR_EPI_VEC2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] += vec2;
    }
    if (has_res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] += resv[r];
    }
    if (f & CRDR_EPI_AFFINE) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = o[r] * scale + shift;
    }
    if (do_cs) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cpre += live(hb * 16 + r) ? o[r] : 0.f;
    }
    if (has_mask) {
      const float moff = (f & CRDR_EPI_MASKOFF) ? vec2 : 0.f;
      if (f & CRDR_EPI_LRELUMASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (mskv[r] - moff) > 0.0f ? o[r] : 0.2f * o[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (mskv[r] - moff) > 0.0f ? o[r] : 0.0f;
      }
    }
    if (do_cs) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cpost += live(hb * 16 + r) ? o[r] : 0.f;
    }
    if (accum) {
      float oldv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) oldv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, lane_off(p.ldy, hb * 16 + r), pixs(hb * 16 + r) * p.ldy * 4, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] += oldv[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[r]), ry, lane_off(p.ldy, hb * 16 + r), pixs(hb * 16 + r) * p.ldy * 4, 0);
  }
}

// Persistent: the launch has at most one workgroup per CU, each walks the tiles vb = blockIdx.x, + gridDim.x, ...  Before the
// epilogue of a tile the waves already issue the first sub-step of the next one, so its DMA latency (and the dispatch of a
// fresh workgroup) is hidden behind the output transform and the stores.
template <bool PAIRS>
__global__ __launch_bounds__(kNT) void wino_kernel(const IgemmArgs p_, const IgemmGroup grp, int gx, int gyn, int npair, int gz) {
  constexpr int kStageFloats = WinoLds<PAIRS>::kStageFloats, kLdsTileFloats = WinoLds<PAIRS>::kTileFloats;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane_ = threadIdx.x & 63, wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int total = (gx * gyn + npair) * gz, gy = gyn + (npair ? 1 : 0);   // gy: filter tiles per group
  float* sV = smem + kLdsTileFloats;          // [4][64]: bias, vec2, scale, shift
  float* sS = sV + 4 * 64;                    // [8 waves][2][32] column sums
  bool stage0_issued = false;
  for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
  // (per-lane constants of the K loop are re-derived per tile instead of staying live across the register-hungry epilogue)
  int lane = lane_, wave = wave_;
  asm volatile("" : "+v"(lane));
  asm volatile("" : "+s"(wave));
  const int tid = wave * 64 + lane;
  const int ph = wave & 1, wn = (wave >> 1) & 1, wm = wave >> 2;
  const WinoTile tl = wino_tile(p_, vb, gx, gyn, npair, gz);
  const WinoIO io = wino_io<PAIRS>(p_, grp, tl, gy, wave, lane);
  IgemmArgs p = p_;
  const int gidx = tl.gidx;
  if (p.ngroup > 1) {
    p.x = grp.x[gidx]; p.y = grp.y[gidx]; p.bias = grp.bias[gidx]; p.mask = grp.mask[gidx]; p.res = grp.res[gidx]; p.cs = grp.cs[gidx];
  }
  // (a pair tile: the wn = 1 waves work on the tile's second patch)
  const bool second = tl.pair && wn;
  const int n = second ? tl.nB : tl.n, oh0 = second ? tl.oh0B : tl.oh0, ow0 = second ? tl.ow0B : tl.ow0, n0 = tl.n0;
  const int patch = second ? tl.patchB : tl.patch;   // (-1: no such patch)

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // per-column epilogue vectors (the previous tile's epilogue ended with a barrier; the K loop's barriers publish them)
  {
    const int f0 = p.flags;
    if (tid < 64) {
      const bool live = n0 + tid < p.Cout;
      sV[0 * 64 + tid] = (live && (f0 & CRDR_EPI_BIAS)) ? p.bias[n0 + tid] : 0.f;
      sV[1 * 64 + tid] = (live && (f0 & (CRDR_EPI_VEC2 | CRDR_EPI_MASKOFF))) ? p.vec2[n0 + tid] : 0.f;
      sV[2 * 64 + tid] = (live && (f0 & CRDR_EPI_AFFINE)) ? p.scale[n0 + tid] : 1.f;
      sV[3 * 64 + tid] = (live && (f0 & CRDR_EPI_AFFINE)) ? p.shift[n0 + tid] : 0.f;
    }
  }
  if (ph == 0) wino_loop<0, PAIRS>(p, smem, io, stage0_issued, lane, wave, wm, wn, acc);
  else wino_loop<1, PAIRS>(p, smem, io, stage0_issued, lane, wave, wm, wn, acc);
  // (the loop ends with a barrier: every wave is past its last LDS read, the stages are free)
  // the next tile's first sub-step goes out now (stage 0; the epilogue below works in the region behind it)
  stage0_issued = false;
  if (vb + (int)gridDim.x < total) {   // (recomputed at the top of the next iteration: nothing of it stays live across the epilogue)
    const WinoTile tn = wino_tile(p_, vb + (int)gridDim.x, gx, gyn, npair, gz);
    const WinoIO ion = wino_io<PAIRS>(p_, grp, tn, gy, wave, lane);
    wino_issue<PAIRS>(p_, smem, ion, 0, 0, 0, lane, wave);
    stage0_issued = true;
  }

  // ---- output transform, hand-over between the two waves of a pair, element-wise epilogue and stores (wino_finish below)
  const int f = p.flags;
  const bool do_cs = (f & CRDR_EPI_COLSUM) != 0;
  float cpre = 0.f, cpost = 0.f;
  {
    float* sXb = smem + kStageFloats;
    if (ph == 0) wino_finish<0>(p, tl.pair, patch, n, oh0, ow0, n0, sXb, sV, lane, wave, wm, wn, acc, cpre, cpost);
    else wino_finish<1>(p, tl.pair, patch, n, oh0, ow0, n0, sXb, sV, lane, wave, wm, wn, acc, cpre, cpost);
  }
  if (do_cs) {   // lane -> wave (the two half-waves hold different tiles of the same channel) -> workgroup, fixed order
    cpre += __shfl_xor(cpre, 32, 64);
    cpost += __shfl_xor(cpost, 32, 64);
    if (lane < 32) { sS[(wave * 2 + 0) * 32 + lane] = cpre; sS[(wave * 2 + 1) * 32 + lane] = cpost; }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63, w2 = c >> 5, cc = c & 31;
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // waves (ph, wn = w2, wm): wave = ph + 2 wn + 4 wm, in the order (wm, ph)
        const int wv = (q & 1) + 2 * w2 + 4 * (q >> 1);
        v += sS[(wv * 2 + which) * 32 + cc];
      }
      // plain tile: the wn = w2 waves hold channels 32 w2 + cc of the tile's patch; pair tile: channels cc of patch A / B
      const int prow = tl.pair ? (w2 ? tl.patchB : tl.patch) : tl.patch, col = n0 + (tl.pair ? cc : c);
      if (prow >= 0 && col < p.Cout) p.cs[((size_t)prow * 2 + which) * p.cs_ld + col] = v;
    }
  }
  lds_barrier();   // the hand-over area, sV and sS are free for the next tile
  }  // tile loop
}

// Filter transform U = G g G^T, G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]], from the implicit-GEMM weight pack
// (tap-major [tap][wrows][wcols]) into the stage-block layout of wino_kernel: [N tile][chunk of 8 channels][position 16][h 2][oc 64][4].
// One thread per (N tile, chunk, h, oc): 9 x 16 B in, 16 x 16 B out (consecutive threads = consecutive oc: coalesced both ways).
struct WinoTaps { int widx[36]; };   // weight index of kernel offset (a, b) on the (3 ks) x (3 ks) grid, -1 outside the kernel
__global__ void wino_filter_kernel(const IgemmGroup grp, int ngroup, const float* w0, float* u, int Cin, int Cout, int wrows, int wcols, int kchunks,
                                   int ntile, int ks, WinoTaps tp) {
  const int nsub = ks * ks;
  const long long total = (long long)ntile * nsub * kchunks * 128;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int g = blockIdx.y;
  const float* w = ngroup > 1 ? grp.w[g] : w0;
  const int oc64 = (int)(id & 63), h = (int)((id >> 6) & 1);
  const long long blk = id >> 7;   // (N tile, sub-filter, chunk)
  const int kc = (int)(blk % kchunks), sub = (int)((blk / kchunks) % nsub), ct = (int)(blk / ((long long)kchunks * nsub));
  const int sa = sub / ks, sb = sub - sa * ks;
  const int oc = ct * 64 + oc64, c0 = kc * 8 + h * 4;
  f32x4 g9[3][3];
  const bool live = oc < Cout && c0 < Cin;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int wi = tp.widx[(3 * sa + a) * (3 * ks) + 3 * sb + b];
      if (live && wi >= 0) v = *reinterpret_cast<const f32x4*>(w + ((size_t)wi * wrows + oc) * wcols + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e >= Cin) v[e] = 0.f;
      g9[a][b] = v;
    }
  f32x4 t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g9[0][b];
    t[1][b] = 0.5f * (g9[0][b] + g9[1][b] + g9[2][b]);
    t[2][b] = 0.5f * (g9[0][b] - g9[1][b] + g9[2][b]);
    t[3][b] = g9[2][b];
  }
  float* dst = u + ((size_t)g * ntile * nsub * kchunks + (size_t)blk) * (kUSlots * 4) + ((size_t)h * 64 + oc64) * 4;
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {
    f32x4 o[4];
    o[0] = t[xi][0];
    o[1] = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
    o[2] = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
    o[3] = t[xi][2];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) *reinterpret_cast<f32x4*>(dst + (size_t)(xi * 4 + nu) * 512) = o[nu];
  }
}

}  // namespace

static int wino_ks(const crdr_conv_desc* d) { return d->kh == 5 ? 2 : 1; }   // sub-filters per axis

bool wino_eligible(const crdr_conv_desc* d, int G) {
  if (!((d->kh == 3 && d->kw == 3) || (d->kh == 5 && d->kw == 5)) || d->stride != 1 || d->wlayout != 0) return false;
  const int k = d->kh;
  const int grow = d->transposed ? (k - 1) - 2 * d->pad : 2 * d->pad - (k - 1);   // (a stride-1 transposed conv = a conv with pad k - 1 - pad)
  if (d->OH != d->H + grow || d->OW != d->W + grow) return false;
  if (d->pad < 0 || d->pad > k - 1) return false;
  if (d->C % 4 != 0 || d->ldx % 4 != 0) return false;
  if (d->flags & (CRDR_EPI_GATE | CRDR_EPI_PREADD | CRDR_CONV_BF16X3)) return false;
  if (G > 1 && (d->flags & (CRDR_EPI_VEC2 | CRDR_EPI_AFFINE | CRDR_EPI_MASKOFF))) return false;
  const long long img = ((long long)d->N * d->H + 8) * d->W * d->ldx * 4;   // one descriptor spans the whole input tensor
  const long long oimg = (long long)d->OH * d->OW * std::max(std::max(d->ldy, d->ldres), d->ldmask) * 4;
  if (img >= (1ll << 31) || oimg >= (1ll << 31)) return false;
  if ((long long)wino_workspace(d, G) / G >= (1ll << 31)) return false;
  return true;
}

size_t wino_workspace(const crdr_conv_desc* d, int G) {
  return (size_t)G * cdiv(d->OC, 64) * wino_ks(d) * wino_ks(d) * cdiv(d->C, 8) * kUSlots * 16;
}

int wino_colsum_rows(const crdr_conv_desc* d) { return d->N * cdiv(d->OH, 16) * cdiv(d->OW, 16); }

// variant 1 (pair tiles): the channel count leaves a tail of <= 32 and there is more than one patch to pair
bool wino_pairs_ok(const crdr_conv_desc* d) {
  const int tail = d->OC % 64;
  return tail > 0 && tail <= 32 && d->N * cdiv(d->OH, 16) * cdiv(d->OW, 16) > 1;
}

// a: the argument block of the implicit-GEMM plan with every pointer / stride / flag filled in; taps: the plan's tap table
int wino_launch(const crdr_conv_desc* d, int variant, IgemmArgs a, const IgemmTaps& taps, const IgemmGroup& grp, int G, float* u, hipStream_t s) {
  CRDR_REQUIRE(wino_eligible(d, G), "conv2d: the Winograd kernel takes 3x3 / 5x5 stride-1 convolutions (C %% 4 == 0, no gate / pre-add epilogue)");
  const int ks = wino_ks(d), kk = d->kh, nt = kk * kk;
  WinoTaps wt;
  int dmin = 127;
  for (int t = 0; t < nt; ++t) dmin = std::min(dmin, (int)(signed char)(taps.packed[t] & 0xff));
  for (int t = 0; t < 36; ++t) wt.widx[t] = -1;
  for (int t = 0; t < nt; ++t) {
    const int v = taps.packed[t];
    const int dh = (int)(signed char)(v & 0xff) - dmin, dw = (int)(signed char)((v >> 8) & 0xff) - dmin;
    CRDR_REQUIRE(dh >= 0 && dh < kk && dw >= 0 && dw < kk, "conv2d: Winograd: tap offsets are not a %dx%d window", kk, kk);
    wt.widx[dh * (3 * ks) + dw] = v >> 16;
  }
  for (int a2 = 0; a2 < kk; ++a2)
    for (int b2 = 0; b2 < kk; ++b2) CRDR_REQUIRE(wt.widx[a2 * (3 * ks) + b2] >= 0, "conv2d: Winograd: incomplete %dx%d window", kk, kk);
  const int ntile = cdiv(d->OC, 64), kchunks = cdiv(d->C, 8);
  {
    const long long total = (long long)ntile * ks * ks * kchunks * 128;
    hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)cdiv64(total, 256), G), dim3(256), 0, s, grp, G, a.w, u, d->C, d->OC, d->wrows, d->wcols, kchunks,
                       ntile, ks, wt);
    CRDR_CHECK_LAUNCH("wino_filter_kernel");
  }
  a.w = u;
  a.kchunks = kchunks;
  {
    const double A = (double)d->N * d->H * d->W * d->C * 4.0, B = (double)wino_workspace(d, 1);
    a.m_inner = (G == 1 && B >= 32.0e6 && A * 4.0 <= B) ? 1 : 0;
  }
  a.nphase = ks * ks;   // sub-filters (a 5x5 kernel = 2 x 2 sub-filters of 3x3, the outer ones zero padded, 3 pixels apart)
  a.so = ks;
  a.GH = cdiv(d->OH, 16);
  a.GW = cdiv(d->OW, 16);
  a.si = -dmin;   // the patch starts `si` pixels above / left of its first output pixel
  a.cs_rows = wino_colsum_rows(d);
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    return n / 8 * 8;
  }();
  // N tiles: full 64-channel ones, then either one padded tile for the tail (variant 0) or pair tiles (variant 1, two patches each)
  const int gx = d->N * a.GH * a.GW;
  const bool pairs = variant == 1;
  CRDR_REQUIRE(!pairs || wino_pairs_ok(d), "conv2d: Winograd pair-tile variant: needs a channel tail of 1..32 and more than one patch");
  const int gyn = pairs ? d->OC / 64 : ntile, npair = pairs ? (gx + 1) / 2 : 0;
  const int total = (gx * gyn + npair) * G;
  static std::atomic<bool> attr_done[2];
  if (!attr_done[variant].load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(pairs ? reinterpret_cast<const void*>(wino_kernel<true>) : reinterpret_cast<const void*>(wino_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[variant].store(true, std::memory_order_release);
  }
  if (pairs) {
    const size_t lds = (size_t)(WinoLds<true>::kTileFloats + 4 * 64 + 8 * 2 * 32) * sizeof(float);
    hipLaunchKernelGGL(wino_kernel<true>, dim3(std::min(total, ncu)), dim3(kNT), lds, s, a, grp, gx, gyn, npair, G);
  } else {
    const size_t lds = (size_t)(WinoLds<false>::kTileFloats + 4 * 64 + 8 * 2 * 32) * sizeof(float);
    hipLaunchKernelGGL(wino_kernel<false>, dim3(std::min(total, ncu)), dim3(kNT), lds, s, a, grp, gx, gyn, npair, G);
  }
  CRDR_CHECK_LAUNCH("wino_kernel");
  return 0;
}

}  // namespace crdr
