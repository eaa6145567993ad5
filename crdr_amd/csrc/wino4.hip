// Winograd F(4x4, 3x3) convolution on the exact-fp32 matrix cores for the 3x3 stride-1 layers (and, as 3x3 pieces, the 5x5 stride-2 and
// stride-1 layers: wino4_mode below; tile geometries: W4Geo; K splits inside the launch: wino4_finish)
// (elic_layers.py:23-36, cheng_nlam.py:31-46, clic21_gvae_discriminator.py:27-40) and their input gradients:
//   Y = A^T [ (G g G^T) . (B^T d B) ] A   with 6x6 transforms (Lavin & Gray 2016; interpolation points 0, +-3/4, +-5/4, inf since round 5:
//   wino4_xform.hpp says why):
// 36 element-wise products per 4x4 outputs and channel pair instead of 144 -- 4x fewer MFMAs than the implicit GEMM, 1.78x fewer
// than the F(2x2, 3x3) kernel of wino.hip.  Arithmetic is fp32 throughout (the filter transform is evaluated in double and rounded
// once); with the round-5 points the result deviates from float64 by ~1.5e-6 .. 5e-6 of the output scale at 96 .. 1 024 accumulated
// channels (the points 0, +-1, +-2 of rounds 3-4: 5e-6 .. 2.3e-5; the direct kernels: ~1e-6 .. 4e-6): a tuner candidate for TRAINING launches
// only (the codec never runs tuned plans), behind the forced-algorithm id of Winograd variant 2, tests/test_gpu_wino.py.
//
// One output tile = 8 rows x 64 columns of output pixels (2 x 16 Winograd tiles of 4 x 4) of one image x 64 output channels, on FOUR
// waves (one per SIMD: 512 registers each), all running the SAME code: wave (th, oh) owns the 16 tiles of tile row th x the 32
// channels of half oh x ALL 36 transform positions = 72 accumulator blocks of v_mfma_f32_16x16x4_f32 (A = filter fragment: row =
// channel; B = transformed data: column = tile; k = the 4 input channels of a sub-step, one per 16-lane group).  A lane ends up with
// 2 x 4 consecutive channels of ONE tile at all 36 positions: the output transform is pure register arithmetic (no hand-over between
// waves) and the epilogue works with 16-byte accesses.
// History of the form (measured, 128 -> 128 channels at 128 x 128, bs 16; the F(2x2) kernel: 396 us): positions split over the waves
// (9 per wave, MFMA 32x32x2, four role-specialised epilogues + LDS hand-over: instruction-cache bound) 470 us; this symmetric form
// with 16 channels per wave and 8-channel sub-steps 349 us -- one wave per SIMD issues EVERYTHING itself, and 288 transform
// operations + 71 LDS reads + 16 DMA per 72 MFMAs made the loop issue-bound at 1.8x the MFMA time; 32 channels per wave and
// 4-channel sub-steps halve the non-MFMA instructions per MFMA.
// K loop: sub-steps of 4 input channels, one barrier each; LDS holds three raw-patch buffers and two filter buffers filled by LDS-DMA:
//   * the raw 10 x 66 x 4 input patch, stored by pixel class (row & 3, column & 3): [class 16][64 slots of 16 B: 3 rows x 17 used]; lane (tile tx,
//     channel kg) reads 4 bytes of patch pixel (i, j): the 16 tiles of a row x 4 channels cover 256 consecutive bytes (conflict free);
//   * the transformed filters of the sub-step: [position 36][channel 4][oh 2][tx 16][ob 2] (output channel 32 oh + 16 ob + tx) = 36 KiB,
//     contiguous in memory and in LDS.
//   Every lane transforms the 6 x 6 patch of its tile for its channel in registers (12 fma-class operations per 1-D transform) and
//   the wave issues 72 MFMAs; the work is laid out by hand in 72 slots of one MFMA + its share of loads / transform / DMA.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.hpp"
#include "igemm_args.hpp"
#include "wino.hpp"
#include "wino4_xform.hpp"

namespace crdr {

namespace {

constexpr int kNT4 = 256;

// How the 32 Winograd tiles (4 x 4 outputs each) of an output tile lie over the image -- geometry 0: 2 rows x 16 columns (8 x 64 pixels:
// images of >= 33 columns), geometry 1: 4 rows x 8 columns (16 x 32 pixels: the 32-column images, where geometry 0 would compute a
// half-empty tile).  A wave owns 16 of them: tile row th (geometry 0) or tile rows 2 th, 2 th + 1 (geometry 1); lane (tx, kg) -> tile
// (ty, txx).  The raw patch keeps its layout [class 16][64 slots], a class row being TX + 1 slots (17 x 3 = 51 / 9 x 5 = 45 used); with
// geometry 1 the two tile rows of a wave read 2 x 128 bytes 144 bytes apart: four banks are touched twice (one extra cycle per read).
// Geometry 2: TWO images of at most 16 x 16 pixels (4 x 4 tiles each), wave th = image th of the pair: the 16 x 16 stage of the context
// model; a class holds 2 x 32 slots, image th's 5 x 5 used ones at 32 th.
template <int GEO> struct W4Geo {
  static constexpr int TY = GEO == 0 ? 2 : 4, TX = GEO == 0 ? 16 : (GEO == 1 ? 8 : 4), PR = TX + 1, USED = GEO == 2 ? 57 : (TY + 1) * (TX + 1);
};
template <int GEO> __device__ __forceinline__ int w4_ty(int th, int tx) { return GEO == 0 ? th : (GEO == 1 ? 2 * th + (tx >> 3) : tx >> 2); }
template <int GEO> __device__ __forceinline__ int w4_tx(int tx) { return GEO == 0 ? tx : (GEO == 1 ? tx & 7 : tx & 3); }
template <int GEO> __device__ __forceinline__ int w4_slot0(int th) { return GEO == 2 ? 32 * th : 0; }   // first slot of the wave's image inside a class
constexpr int kBN4 = 64;                               // output channels per tile
constexpr int kClsSlots = 64;                          // slots per pixel class: 3 rows x 17 = 51 used; 64 = 1 KiB apart, so that two pixels of a patch column
                                                       // (different classes) are one ds_read2st64_b32
constexpr int kInUsed4 = 16 * kClsSlots;               // 16-byte slots (4 channels of one pixel) of the input patch image per buffer
constexpr int kInPieces = 16;                          // DMA instructions (1 KiB each) for it, 4 per wave
constexpr int kInSlots4 = kInPieces * 64;
constexpr int kUSlots4 = 36 * 4 * kBN4 / 4;            // slots of one filter block (4 channels x 64 output channels): 36 pieces
constexpr int kRawBufs = 3, kFiltBufs = 3;            // raw patches are requested three sub-steps ahead, filter blocks two: nothing requested during a sub-step is waited for at its end
constexpr int kRF = kInSlots4 * 4, kFF = kUSlots4 * 4;   // floats of a raw buffer / a filter buffer
constexpr int kLdsFloats4 = kRawBufs * kRF + kFiltBufs * kFF;

__device__ __forceinline__ void lds_barrier4() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// 1-D output transform A^T = [[1, 1, 1, 1, 1, 0], [0, a, -a, b, -b, 0], [0, a2, a2, b2, b2, 0], [0, a3, -a3, b3, -b3, 1]] (wino4_xform.hpp: a = 3/4,
// b = 5/4), two channels at once in packed fp32 (the epilogue's vector instructions are not hidden behind anything either): 13 operations
__device__ __forceinline__ void at6_pk(const f32x2v m0, const f32x2v m1, const f32x2v m2, const f32x2v m3, const f32x2v m4, const f32x2v m5,
                                       f32x2v& o0, f32x2v& o1, f32x2v& o2, f32x2v& o3) {
  const f32x2v p = m1 + m2, q = m1 - m2, u = m3 + m4, w = m3 - m4;
  o0 = m0 + p + u;
  o1 = pk_fma(pk_bc(kWb), w, pk_bc(kWa) * q);
  o2 = pk_fma(pk_bc(kWb2), u, pk_bc(kWa2) * p);
  o3 = pk_fma(pk_bc(kWb3), w, pk_bc(kWa3) * q) + m5;
}

struct Wino4Tile { int gidx, n, oh0, ow0, n0, patch, tn, phase, lin, ks, kbeg, kcnt; };   // lin: index of the tile among the launch's tiles; K split ks works sub-steps [kbeg, kbeg + kcnt)   // phase: output phase of a stride-2 transposed conv (0 otherwise)

template <int GEO>
__device__ __forceinline__ Wino4Tile wino4_tile(const IgemmArgs& p, int vb, int gx, int gyn_, int gz) {
  // gyn_ < 0: FILTER-STATIONARY order -- the patches of an N tile back to back instead of the N tiles of a patch: where the transformed
  // filters outweigh the input (the hoisted convs of the context model: 784 MB of filters against 5 MB of latents) each XCD then streams
  // its share of the filters once instead of all of them (6.5 GB of HBM reads per launch, measured)
  const bool fstat = gyn_ < 0;
  const int gyn = fstat ? -gyn_ : gyn_;
  // XCD-aware order (see igemm_kernel.hpp): the hardware deals workgroups round-robin over the 8 XCDs; every XCD walks a contiguous
  // range of (patch, N tile) pairs, the N tiles of a patch back to back (they re-read the patch out of that XCD's L2)
  Wino4Tile t;
  const int nph = p.so * p.so;   // output phases (4 for a stride-2 transposed conv: each is a stride-1 conv of its own with the same input)
  // (split K, p.nsplit > 1: the work items are (tile, split), the splits of a tile back to back -- they share the patch)
  const int T = gx * gyn * nph, S = p.nsplit, nwg = T * gz * S, cpx = nwg >> 3;
  const int qs = vb < cpx * 8 ? (vb & 7) * cpx + (vb >> 3) : vb;
  const int q = qs / S;
  t.ks = qs - q * S;
  t.lin = q;
  {
    const int cnt = (p.kchunks + S - 1) / S;
    t.kbeg = t.ks * cnt;
    t.kcnt = min(cnt, p.kchunks - t.kbeg);
  }
  t.gidx = q / T;
  const int r = q - t.gidx * T;
  // filter-stationary: (N tile, phase) outermost -- ONE block of transformed filters (36 positions x K x 64 channels: 2.4 MB at K = 256) stays in
  // the XCD's L2 while the patches stream past it (round 6; before, the phases of a patch ran back to back and an XCD's 32 concurrent
  // workgroups cycled through every block of the launch each round: T 256->256 k5 s2 fetched 2.9 GB for 0.34 GB of operands)
  int tn;
  if (fstat) {
    const int combo = r / gx;
    t.patch = r - combo * gx;
    tn = combo / nph;
    t.phase = combo - tn * nph;
  } else {
    t.phase = r % nph;
    const int r1 = r / nph;
    tn = r1 % gyn;
    t.patch = r1 / gyn;
  }
  t.tn = tn;
  if constexpr (GEO == 2) {
    t.n = 2 * t.patch;   // (first image of the pair; wave th works on image n + th)
    t.oh0 = 0;
    t.ow0 = 0;
  } else {
    const int ppi = p.GH * p.GW;
    t.n = t.patch / ppi;
    const int prem = t.patch - t.n * ppi, by = prem / p.GW, bx = prem - by * p.GW;
    t.oh0 = by * (4 * W4Geo<GEO>::TY);
    t.ow0 = bx * (4 * W4Geo<GEO>::TX);
  }
  t.n0 = tn * kBN4;
  return t;
}

// byte offset (into the input tensor's descriptor) of the pixel a DMA lane stages for input piece `piece`: slot S = piece * 64 + lane ->
// (class (ci, cj), R, Cc) -> patch pixel (4 R + ci, 4 Cc + cj), channels 0..3 of the sub-step; out of range where the slot is unused or
// the pixel lies outside the image (the range check of the buffer load then delivers zeros: padding)
// FORM 1 (a 5x5 stride-1 layer as four 3x3 sub-filters, taps 3 bi + a, 3 bj + b): sub-filter `sub` = (bi, bj) reads the patch displaced by
// (3 bi, 3 bj) pixels; whether a pixel is padding then depends on the sub-filter, so the lane offsets are re-derived when the K loop
// moves to the next one (four times per tile).
// (the slot decode depends on the lane and the piece only: wino4_piece_geo does it once per kernel -- pi | pj << 8 | image of the pair << 16 |
// slot in use << 17 --, wino4_in_off finishes it per tile / sub-filter: the decode was two thirds of the 1 700 cycles a tile spent here)
template <int GEO>
__device__ __forceinline__ unsigned wino4_piece_geo(int piece, int lane) {
  using G = W4Geo<GEO>;
  const int S = piece * 64 + lane;
  const int cls = S / kClsSlots, r2f = S - cls * kClsSlots;
  const int img = GEO == 2 ? r2f >> 5 : 0, r2 = GEO == 2 ? r2f & 31 : r2f;
  const int R = r2 / G::PR, Cc = r2 - R * G::PR;
  const int ci = cls >> 2, cj = cls & 3;
  const bool used = S < kInUsed4 && r2 < (GEO == 2 ? 25 : G::USED);
  return (unsigned)(4 * R + ci) | ((unsigned)(4 * Cc + cj) << 8) | ((unsigned)img << 16) | ((used ? 1u : 0u) << 17);
}
template <int GEO, int FORM>
__device__ __forceinline__ unsigned wino4_in_off(const IgemmArgs& p, const Wino4Tile& t, unsigned geo, int sub) {
  using G = W4Geo<GEO>;
  const int pi = (int)(geo & 0xff) + (FORM == 1 ? 3 * (sub >> 1) : 0), pj = (int)((geo >> 8) & 0xff) + (FORM == 1 ? 3 * (sub & 1) : 0);
  const int pmax_i = 4 * G::TY + 2 + (FORM == 1 ? 3 * (sub >> 1) : 0), pmax_j = 4 * G::TX + 2 + (FORM == 1 ? 3 * (sub & 1) : 0);
  const int n = t.n + (int)((geo >> 16) & 1);
  // (5x5 stride-2 conv as four parity sub-filters: plane pixel m of parity (ph, pw) is image pixel 2 m + parity; the parity displacement
  // is wave-uniform and travels in the request's scalar offset; H and W are even there, so validity does not depend on the parity)
  const int ist = (FORM == 0 && p.nphase == 4) ? 2 : 1;
  const int ih = ist * (t.oh0 - p.si + pi), iw = ist * (t.ow0 - p.si + pj);
  const bool ok = ((geo >> 17) & 1) && pi < pmax_i && pj < pmax_j && n < p.N && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
  return ok ? (unsigned)((((n * p.H + ih) * p.W + iw) * p.ldx) * 4) : kOobOffset;
}

// where the DMA of one output tile reads: descriptors of its image tensor and filter blocks, this lane's four raw-piece offsets
struct Wino4Src {
  __amdgpu_buffer_rsrc_t rx, ru;
  unsigned u_off0;     // byte offset of the N tile's first filter block
  unsigned a_off[4];   // raw piece wave + 4 j: byte offset of this lane's pixel (chunk 0), or out of range
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wino4_empty_rsrc() { return __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000); }

// raw piece j (0..3) of this wave, patch kr -> raw buffer rbuf through descriptor rx: the tile's own (sr.rx), or an EMPTY one where the
// sub-step does not exist (every lane out of range: zeros land in a buffer nobody reads) -- liveness is wave-uniform, so it travels in the
// scalar descriptor instead of a per-lane select (every vector instruction of the K loop costs matrix time)
__device__ __forceinline__ void wino4_dma_raw(float* smem, const Wino4Src& sr, __amdgpu_buffer_rsrc_t rx, int j, int soff, int rbuf, int lane, int wave) {
  const int piece = wave + 4 * j;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(smem + rbuf * kRF + piece * 256), 16, (int)sr.a_off[j], soff, 0, 0);
}
// scalar byte offset of sub-step kr's raw patch: channel chunk kr % Kc, parity sub-filter kr / Kc (stride-2 conv: + (ph W + pw) pixels)
template <int FORM>
__device__ __forceinline__ int wino4_raw_soff(const IgemmArgs& p, int kr) {
  if (p.nphase == 1) return kr * 16;
  const int Kc = p.kchunks / p.nphase, sub = kr / Kc, ch = kr - sub * Kc;
  if (FORM == 1) return ch * 16;   // (the displacement of a shifted sub-filter sits in the lane offsets: wino4_in_off)
  return ch * 16 + ((sub >> 1) * p.W + (sub & 1)) * p.ldx * 4;
}
// filter piece 9 wave + j (j = 0..8) of block kf -> filter buffer fbuf (the piece's displacement is wave-uniform: it travels in the scalar
// offset, the lane part is the same register for every piece -- no vector add per request)
__device__ __forceinline__ void wino4_dma_filt(float* smem, const Wino4Src& sr, __amdgpu_buffer_rsrc_t ru, int j, int kf, int fbuf, int lane, int wave) {
  const int piece = wave * 9 + j;
  // (u_off0 is wave-uniform; said so explicitly, or the compiler wraps every request in a waterfall loop over the scalar offset)
  const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)sr.u_off0) + (unsigned)kf * (kUSlots4 * 16u) + (unsigned)piece * 1024u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(smem + kRawBufs * kRF + fbuf * kFF + piece * 256), 16, lane * 16, (int)so, 0, 0);
}
// the requests a tile starts with: raw patches 0, 1, 2 (-> R0, R1, R2) and filter blocks 0, 1 (-> F0, F1): 30 per wave; request i: 0..11 raw
// piece i % 4 of patch i / 4, 12..29 filter piece (i - 12) % 9 of block (i - 12) / 9
constexpr int kW4PrologueReqs = 30;
template <int FORM>
__device__ __forceinline__ void wino4_prologue_req(const IgemmArgs& p, float* smem, const Wino4Src& sr, int kbeg, int kcnt, int lane, int wave, int i) {
  if (i < 12) {
    const int kr = i >> 2;
    wino4_dma_raw(smem, sr, kr < kcnt ? sr.rx : wino4_empty_rsrc(), i & 3, wino4_raw_soff<FORM>(p, kbeg + kr), kr, lane, wave);
  } else {
    const int kf = (i - 12) / 9;
    wino4_dma_filt(smem, sr, kf < kcnt ? sr.ru : wino4_empty_rsrc(), (i - 12) - 9 * kf, kbeg + kf, kf, lane, wave);
  }
}
template <int FORM>
__device__ __forceinline__ void wino4_prologue_dma(const IgemmArgs& p, float* smem, const Wino4Src& sr, int kbeg, int kcnt, int lane, int wave) {
#pragma unroll
  for (int i = 0; i < kW4PrologueReqs; ++i) wino4_prologue_req<FORM>(p, smem, sr, kbeg, kcnt, lane, wave, i);
}
// the next tile's first requests, handed to the epilogue of the current one: issued back to back they hold the wave for ~160 cycles each (the
// CU's load path takes 1 KiB per request: 4 800 - 5 000 cycles per tile by the stamps, tools/experiments/w4_stamps.py); spread through the
// vector arithmetic of the first half's output transform they cost their issue slots
struct Wino4Next { bool more; int kbeg, kcnt; };

// K loop of one wave (the only wave of its SIMD: nothing else hides its latencies, so the loop is software pipelined by hand).
// LDS: raw-patch buffers R0..R2, filter buffers F0..F2.  Sub-step k multiplies V_k (registers) with the filters of F[k % 3]; beside those 72
// MFMAs the wave reads raw patch k + 1 from R[(k + 1) % 3] and transforms it into V_{k+1}, and issues the DMA of filter block k + 2 and of
// raw patch k + 3.  Slot 65 of the sub-step is vmcnt(13) + barrier: what it requested itself stays in flight, everything older has landed; the
// last six MFMAs cover the reads of the next sub-step's first four filter fragments.
// Schedule of a sub-step (round 5, after the issue probe: VALU never hides behind an fp32 MFMA of the same wave, memory instructions do if
// they are issued directly behind one), 72 slots pinned by sched_barrier:
//   slot s: MFMA of position j = s / 2 (transform row x = j / 6), channel block s % 2, followed by AT MOST ONE memory instruction --
//     odd s < 64: the two filter fragments of position j + 4 (one ds_read_b64, four positions ahead of their use);
//     even s < 36: raw pair r = s / 2 = pixels (i, 2 c), (i, 2 c + 1) of patch row i = r % 6, column pair c = r / 6 (one ds_read2st64_b32);
//     even s in 36 .. 60: DMA instruction (s - 36) / 2 (9 filter pieces, then 4 raw pieces);
//   then the slot's VALU cluster, if any: vertical pass of column pair c (12 packed operations) at s = 15 + 12 c; horizontal pass of row
//     x < 5 of V_{k+1} (6 packed operations) at s = 41, 44, 47, 50, 62 -- straight into the registers of V_k's row x, whose MFMAs (slots 12 x
//     .. 12 x + 11) are done by then; row 5 follows at the top of the next sub-step.
template <int GEO, int FORM>
__device__ __forceinline__ void wino4_loop(const IgemmArgs& p, const Wino4Tile& tl, float* smem, Wino4Src& sr, bool prefetched, int lane, int wave,
                                           f32x4 (&acc)[64], f32x4 (&accv)[8], const unsigned (&geo)[4]) {
  const int K4 = tl.kcnt, kbeg = tl.kbeg;   // this work item's sub-steps: kbeg .. kbeg + K4 - 1 of the tile's p.kchunks
  const int tx = lane & 15, kg = lane >> 4, th = wave >> 1, oh = wave & 1;
  // this lane's raw reads: patch pixel (i, j) of tile (th, tx), channel kg: float offset rbase + ro(i, j) of a raw buffer
  const int rbase = (w4_slot0<GEO>(th) + w4_ty<GEO>(th, tx) * W4Geo<GEO>::PR + w4_tx<GEO>(tx)) * 4 + kg;
  auto ro = [](int i, int j) constexpr { return (((i & 3) * 4 + (j & 3)) * kClsSlots + (i >> 2) * W4Geo<GEO>::PR + (j >> 2)) * 4; };
  // filter fragments of position pos (channel kg of the sub-step, output channels 32 oh + 16 ob + tx, ob = 0, 1: adjacent, one 8-byte read):
  // + buffer * kFF + pos * 256
  const int fbase = kRawBufs * kRF + kg * kBN4 + 32 * oh + 2 * tx;

  const W4Consts kc = w4_consts();
  f32x2v T[6][3];      // vertical pass of the patch being transformed: T[xi][c] = (t[xi][2 c], t[xi][2 c + 1])
  f32x2v V[6][3];      // V of the current sub-step, row x as (v0, v5), (v1, v3), (v2, v4) (rows 0..4: replaced in place by the next one's during the sub-step)
  f32x2v D[2][6];      // raw pixels of one column pair, double buffered by the pair's parity: D[c & 1][i] = (d[i][2 c], d[i][2 c + 1])
  // (the two pixels are one class apart -- 1 KiB = 4 x 64 floats: one ds_read2st64_b32; columns 2 c, 2 c + 1 never straddle a slot)
  auto vread = [&](const float* rp, int r) __attribute__((always_inline)) {
    const int c = r / 6, i = r - 6 * c;
    D[c & 1][i] = f32x2v{rp[ro(i, 2 * c)], rp[ro(i, 2 * c + 1)]};
  };
  auto vpass = [&](int c) __attribute__((always_inline)) {
    f32x2v t6[6];
    bt6_cols(D[c & 1], t6, kc);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) T[xi][c] = t6[xi];
  };
  auto hpass = [&](int x) __attribute__((always_inline)) { bt6_row(T[x][0], T[x][1], T[x][2], V[x], kc); };

  // ---- prologue: filter block 0 and patches 0, 1, 2 (requested by the previous tile's epilogue where there was one: then at least 32
  // younger vector-memory operations -- that tile's stores -- are in flight and need not be waited for); V_0 rows 0..4 and the
  // vertical pass of row 5
  if (!prefetched) {
    wino4_prologue_dma<FORM>(p, smem, sr, kbeg, K4, lane, wave);
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
  } else {
    __builtin_amdgcn_s_waitcnt(0x8070);   // vmcnt(32)
  }
  lds_barrier4();
  {
    const float* rp = smem + rbase;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int i = 0; i < 6; ++i) vread(rp, 6 * c + i);
      vpass(c);
    }
#pragma unroll
    for (int x = 0; x < 5; ++x) hpass(x);
  }

  float uf[9][2];   // filter fragments of positions j .. j + 3 in flight (ring of 9, indexed j % 9: 36 = 0 mod 9, so the next sub-step's
  //                   positions 0..3 land in entries the current one has long left); alive across sub-steps
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 u2 = *reinterpret_cast<const f32x2*>(smem + fbase + j * 256);   // (filter block 0 of the tile: buffer 0)
    uf[j][0] = u2[0];
    uf[j][1] = u2[1];
  }
  int r1 = 1, r0 = 0;   // raw buffer of patch k + 1 / of patch k (= the one patch k + 3 goes to)
  const int Kc = p.kchunks / p.nphase;   // channel chunks per parity sub-filter (stride-2 conv: 4 sub-filters; else 1)
  int ch3 = (kbeg + 3) % Kc, sub3 = (kbeg + 3) / Kc;   // chunk / sub-filter of sub-step k + 3 (counters: no division in the loop)
  if constexpr (FORM == 1) {   // (12 input channels: the first patch requested inside the loop already belongs to the second sub-filter)
    if (sub3 != kbeg / Kc && sub3 < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sr.a_off[j] = wino4_in_off<GEO, FORM>(p, tl, geo[j], sub3);
    }
  }
  // per-sub-step scalar state, computed one sub-step AHEAD behind the last MFMAs of the previous one (slots 66 .. 70): ~30 scalar instructions
  // and three address adds used to sit at the loop top, where nothing covers them -- the matrix pipe ran dry for ~150 cycles per sub-step
  // (in-loop stamps, tools/experiments/w4_stamps.py)
  struct SubState {
    int rpo, fpo, fpno;                  // float offsets of raw patch k + 1, filter block k, filter block k + 1 (opaque to the compiler, see below)
    __amdgpu_buffer_rsrc_t ru2, rx3;     // descriptors of the requests for filter block k + 2 / raw patch k + 3 (empty past the last sub-step)
    int soff3, f2;                       // scalar offset of raw patch k + 3; buffer of filter block k + 2
  };
  auto sub_state = [&](int k, int r0_, int r1_, int ch3_, int sub3_) __attribute__((always_inline)) {
    SubState st;
    // (the base offsets are made opaque: every LDS read of the sub-step is then `base register + 16-bit immediate`; left visible,
    // the compiler folds the buffer constants into the offsets, overflows the immediate and spends an add per read)
    st.rpo = rbase + r1_ * kRF;
    st.fpo = fbase + r0_ * kFF;
    st.fpno = fbase + r1_ * kFF;
    asm volatile("" : "+v"(st.rpo));
    asm volatile("" : "+v"(st.fpo));
    asm volatile("" : "+v"(st.fpno));
    st.ru2 = k + 2 < K4 ? sr.ru : wino4_empty_rsrc();
    st.rx3 = k + 3 < K4 ? sr.rx : wino4_empty_rsrc();
    st.soff3 = FORM == 1 ? ch3_ * 16 : ch3_ * 16 + ((sub3_ >> 1) * p.W + (sub3_ & 1)) * p.ldx * 4;   // (sub3 = 0 unless the conv is the stride-2 form)
    st.f2 = r1_ == 2 ? 0 : r1_ + 1;
    return st;
  };
  SubState cur = sub_state(0, r0, r1, ch3, sub3), nxt = cur;
  for (int k = 0; k < K4; ++k) {
    const float* rp = smem + cur.rpo;                     // raw patch k + 1
    const float* fp = smem + cur.fpo;                     // filter block k (the filter ring turns with the raw ring: block k in buffer k % 3)
    const __amdgpu_buffer_rsrc_t ru2 = cur.ru2, rx3 = cur.rx3;
    const int soff3 = cur.soff3, f2 = cur.f2;
    const float* fpn = smem + cur.fpno;                   // filter block k + 1 (its first four positions are fetched at the end of this sub-step)
    bool newsub = false;                                  // the requests of sub-step k + 1 start a new shifted sub-filter (FORM 1)
    hpass(5);   // row 5 of V_k (its vertical pass was done during the previous sub-step; its MFMAs are the last ones)
    __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
    for (int s = 0; s < 72; ++s) {
      const int j = s >> 1, ob = s & 1, x = j / 6, y = j - 6 * x;
      // 72 blocks x 4 = 288 accumulator registers: 64 blocks in the accumulation half of the register file, the last 8 (positions 32..35)
      // pinned to ordinary vector registers (left to the compiler they bounce between the two files: 72 moves per sub-step)
      if (s < 64) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[j % 9][ob], v_elem(V[x], y), acc[s], 0, 0, 0);
      else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accv[s - 64]) : "v"(uf[j % 9][ob]), "v"(v_elem(V[x], y)));
      // the slot's memory instruction, directly behind the MFMA
      if (ob == 1 && j + 4 < 36) {   // four positions (8 slots) ahead of their use
        const f32x2 u2 = *reinterpret_cast<const f32x2*>(fp + (j + 4) * 256);
        uf[(j + 4) % 9][0] = u2[0];
        uf[(j + 4) % 9][1] = u2[1];
      }
      if (s == 70) {   // the next sub-step's scalar state, behind an MFMA (the ring turns, the chunk / sub-filter counters advance)
        r0 = r1;
        r1 = r1 == 2 ? 0 : r1 + 1;
        if (++ch3 == Kc) {
          ch3 = 0;
          ++sub3;
          newsub = true;
        }
        nxt = sub_state(k + 1, r0, r1, ch3, sub3);
      }
      if (s >= 66 && s < 70) {   // behind the sub-step's barrier (slot 65): positions 0..3 of the NEXT sub-step's filter block
        const f32x2 u2 = *reinterpret_cast<const f32x2*>(fpn + (s - 66) * 256);
        uf[s - 66][0] = u2[0];
        uf[s - 66][1] = u2[1];
      }
      if (ob == 0 && s < 36) vread(rp, s >> 1);
      if (ob == 0 && s >= 36 && s <= 60) {
        const int q = (s - 36) >> 1;
        if (q < 9) wino4_dma_filt(smem, sr, ru2, q, kbeg + k + 2, f2, lane, wave);
        else wino4_dma_raw(smem, sr, rx3, q - 9, soff3, r0, lane, wave);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the slot's VALU cluster
      if (s == 15) vpass(0);
      if (s == 27) vpass(1);
      if (s == 39) vpass(2);
      if (s == 41) hpass(0);
      if (s == 44) hpass(1);
      if (s == 47) hpass(2);
      if (s == 50) hpass(3);
      if (s == 62) hpass(4);
      if (s == 15 || s == 27 || s == 39 || s == 41 || s == 44 || s == 47 || s == 50 || s == 62) __builtin_amdgcn_sched_barrier(0);
      if (s == 65) {
        // the sub-step's barrier sits HERE, not at its end: the 13 requests of this sub-step (filter block k + 2, patch k + 3: slots 36 .. 60)
        // stay in flight, everything older (filter block k + 1, patch k + 2) has landed; every wave is past its last read of filter block k
        // (slot 63) and of patch k + 1 (slot 34), which the next sub-step's requests overwrite.  The six MFMAs behind it cover the latency of
        // the next sub-step's first filter fragments -- with the barrier at the end the first MFMA of every sub-step waited for its operands
        __builtin_amdgcn_s_waitcnt(0x007D);   // vmcnt(13) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur = nxt;
    if constexpr (FORM == 1) {   // the patches requested from here on belong to the next shifted sub-filter: its padding pattern
      if (newsub && sub3 < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sr.a_off[j] = wino4_in_off<GEO, FORM>(p, tl, geo[j], sub3);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // (the last sub-step's barrier is 6 MFMAs back: what follows -- the next tile's requests into the raw / filter buffers, the epilogue -- touches
  // no LDS a wave could still be reading for a purpose: the four fragment reads behind that barrier fetched a block nobody multiplies)
  // (the dead tail requests -- zeros for sub-steps past the last -- that may still be in flight go to raw buffers; whatever is requested
  // into those next comes from the same wave and lands behind them)
}

// The accumulators live in the accumulation half of the register file; the output transform takes them out one register at a time
// (left to itself the compiler copies them all into vector registers at the loop exit and spills to scratch)
__device__ __forceinline__ float acc_read(float v) {
  float o;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(o) : "a"(v));
  return o;
}

// Output transform + element-wise epilogue + stores of one wave: lane (tile tx of row th, channel group kg) holds M[36] for channels
// n0 + 32 oh + 16 ob + 4 kg + r, ob < 2, r < 4.  Order of the element-wise operations: epilogue_store of igemm_kernel.hpp.
// SPLIT (p.nsplit > 1 work items per tile, each over a part of the K range): the output transform is linear, so every work item publishes
// its transformed partial tile -- 2 x 16 float4 per lane, in the lane's own order: 128 KiB per tile and split -- with write-through
// stores, takes a ticket on the tile, and the last arriver adds the slabs in split order and runs the epilogue (the hand-off protocol of
// igemm_kernel.hpp's in-launch split-K: sc1 stores, vmcnt(0), barrier, one agent-scope atomic, acquire, sc1 loads; a workgroup either
// leaves or reduces, nobody spins).
// Epilogue classes: the flags a launch MAY carry, as a compile-time superset (the kernel tests `p.flags & FM`, so the passes of every other
// flag disappear from the code: the general epilogue is 51 KB of instructions beside a 2 KB loop -- a 128-channel tile ran 5.6 us faster
// with only its own flags compiled in).  0: bias / (Leaky)ReLU / vec2 / affine (no operand tensors), 1: + residual, 2: the input-gradient
// launches of a conv chain (ReLU masks, column sums, accumulate), 3: everything.
constexpr int kW4EpiA = CRDR_EPI_BIAS | CRDR_EPI_RELU | CRDR_EPI_LRELU | CRDR_EPI_VEC2 | CRDR_EPI_AFFINE;
constexpr int kW4EpiC = CRDR_EPI_BIAS | CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK | CRDR_EPI_MASKOFF | CRDR_EPI_COLSUM | CRDR_EPI_ACCUM;
constexpr int kW4EpiAll = kW4EpiA | kW4EpiC | CRDR_EPI_RES;
constexpr int w4_epi_mask(int cls) { return cls == 0 ? kW4EpiA : cls == 1 ? (kW4EpiA | CRDR_EPI_RES) : cls == 2 ? kW4EpiC : kW4EpiAll; }
constexpr int w4_epi_class(int flags) {
  const int f = flags & kW4EpiAll;
  return (f & ~w4_epi_mask(0)) == 0 ? 0 : (f & ~w4_epi_mask(1)) == 0 ? 1 : (f & ~w4_epi_mask(2)) == 0 ? 2 : 3;
}

template <int GEO, int FORM, bool SPLIT, int FM>
__device__ __forceinline__ void wino4_finish(const IgemmArgs& p, const Wino4Tile& tl, float* smem, const float* sV, int lane, int wave, f32x4 (&acc)[64],
                                             f32x4 (&accv)[8], const Wino4Src& srn, const Wino4Next nx) {
  const int f = p.flags & FM;
  const bool has_res = (f & CRDR_EPI_RES) != 0, has_mask = (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) != 0;
  const bool do_cs = (f & CRDR_EPI_COLSUM) != 0, accum = (f & CRDR_EPI_ACCUM) != 0;
  const int tx = lane & 15, kg = lane >> 4, th = wave >> 1, oh = wave & 1;
  const int oy0 = tl.oh0 + 4 * w4_ty<GEO>(th, tx), ox0 = tl.ow0 + 4 * w4_tx<GEO>(tx);       // first output pixel of this lane's tile
  auto tdesc = [&](const float* base, int ld) __attribute__((always_inline)) {
    const unsigned long long bytes = (((unsigned long long)p.N * p.OH * p.OW - 1) * ld + p.Cout) * 4ull;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (unsigned)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ry = tdesc(p.y, p.ldy), rr = tdesc(has_res ? p.res : p.y, p.ldres), rm = tdesc(has_mask ? p.mask : p.y, p.ldmask);
  // output pixel (a, b) of the lane's tile: (so (oy0 + a) + py, so (ox0 + b) + px) -- so = 2 and (py, px) = the tile's phase for a
  // stride-2 transposed conv, so = 1 otherwise
  const int so = p.so, py = tl.phase >> 1, px = tl.phase & 1;
  const int img = tl.n + (GEO == 2 ? th : 0);   // (geometry 2: the wave's image of the pair)
  const unsigned pix00 = (unsigned)(((size_t)img * p.OH + so * oy0 + py) * p.OW + so * ox0 + px);
  // validity of this lane's 16 output pixels (bit 4 a + b); the (a, b) displacement of an access is wave-uniform and travels as the
  // buffer instruction's scalar offset (the range check only sees the lane part)
  unsigned pixm = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) pixm |= (img < p.N && so * (oy0 + a) + py < p.OH && so * (ox0 + b) + px < p.OW) ? (1u << (4 * a + b)) : 0u;
  auto soff = [&](int ld, int a, int b) __attribute__((always_inline)) { return so * (a * p.OW + b) * ld * 4; };
  float* sC = smem + kRawBufs * kRF + 2 * kFF;   // column sums: [wave 4][which 2][ob 2][64 lanes][4 r] = 16 KiB in filter buffer F2 (F0, F1 receive the next tile's blocks 0, 1)

  // output transform of half ob, two channel registers (r, r + 1) at a time in packed fp32: s[a][nu] = sum_xi AT[a][xi] M[xi][nu], then
  // Y[a][b] = sum_nu AT[b][nu] s[a][nu]
  // (`reqs`: the 30 requests of the next tile ride in this call, three every two of its 20 arithmetic groups)
  auto next_reqs = [&](int grp) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = (3 * grp) / 2; i < (3 * (grp + 1)) / 2; ++i) wino4_prologue_req<FORM>(p, smem, srn, nx.kbeg, nx.kcnt, lane, wave, i);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto out_transform = [&](int ob, float (&yv)[4][4][4], auto reqs_c) __attribute__((always_inline)) {
    constexpr bool REQS = decltype(reqs_c)::value;
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
      f32x2v sv[4][6];
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) {
        if constexpr (REQS) next_reqs(10 * rp + nu);
        f32x2v mcol[6];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
          const int blk = 2 * (xi * 6 + nu) + ob;
#pragma unroll
          for (int e = 0; e < 2; ++e)
            mcol[xi][e] = blk < 64 ? acc_read(acc[blk < 64 ? blk : 0][2 * rp + e]) : accv[blk >= 64 ? blk - 64 : 0][2 * rp + e];
        }
        at6_pk(mcol[0], mcol[1], mcol[2], mcol[3], mcol[4], mcol[5], sv[0][nu], sv[1][nu], sv[2][nu], sv[3][nu]);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if constexpr (REQS) next_reqs(10 * rp + 6 + a);
        f32x2v y0, y1, y2, y3;
        at6_pk(sv[a][0], sv[a][1], sv[a][2], sv[a][3], sv[a][4], sv[a][5], y0, y1, y2, y3);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          yv[a][0][2 * rp + e] = y0[e];
          yv[a][1][2 * rp + e] = y1[e];
          yv[a][2][2 * rp + e] = y2[e];
          yv[a][3][2 * rp + e] = y3[e];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  __amdgpu_buffer_rsrc_t rws = ry;
  unsigned slab_bytes = 0;
  auto slab_off = [&](int ob, int q) __attribute__((always_inline)) { return (unsigned)((((wave * 2 + ob) * 16 + q) * 64 + lane) * 16); };
  if constexpr (SPLIT) {
    if (nx.more) wino4_prologue_dma<FORM>(p, smem, srn, nx.kbeg, nx.kcnt, lane, wave);
    constexpr unsigned kTileSlab = 4u * 2u * 16u * 64u * 16u;   // bytes of one tile's partial result
    slab_bytes = (unsigned)p.ws_ld * kTileSlab;                   // one split's slab: every tile of the launch
    rws = __builtin_amdgcn_make_buffer_rsrc(p.ws + (size_t)tl.lin * (kTileSlab / 4), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      float yv[4][4][4];
      out_transform(ob, yv, std::false_type{});
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const f32x4 v = {yv[q >> 2][q & 3][0], yv[q >> 2][q & 3][1], yv[q >> 2][q & 3][2], yv[q >> 2][q & 3][3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rws, slab_off(ob, q) + (unsigned)tl.ks * slab_bytes, 0, 16 /* sc1 */);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier4();
    int* sFlag = reinterpret_cast<int*>(smem + kLdsFloats4 + 2 * 4 * kBN4);
    if (wave == 0 && lane == 0) {
      int* cnt = p.counters + tl.lin;
      const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == p.nsplit - 1;
      if (last) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch finds zeros again
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *sFlag = last;
    }
    lds_barrier4();
    const int last = *sFlag;
    if (!last) return;   // (workgroup-uniform)
  }

#pragma unroll
  for (int ob = 0; ob < 2; ++ob) {
    const int cl = 32 * oh + 16 * ob + 4 * kg;                  // first of this lane's four channels inside the tile
    const int c0 = tl.n0 + cl;
    const bool c_ok = c0 < p.Cout;   // (Cout % 4 == 0: a group of four channels is in or out as a whole)
    const unsigned okm = c_ok ? pixm : 0u;
    const unsigned base_y = (pix00 * (unsigned)p.ldy + (unsigned)c0) * 4u, base_r = (pix00 * (unsigned)p.ldres + (unsigned)c0) * 4u,
                   base_m = (pix00 * (unsigned)p.ldmask + (unsigned)c0) * 4u;
    auto voff = [&](unsigned base, int a, int b) __attribute__((always_inline)) { return ((okm >> (4 * a + b)) & 1u) ? base : kOobOffset; };
    // residual / mask / accumulate operands, 8 pixels (two output rows) at a time.  Launches carry at most one of them as a rule (the
    // ReLU mask of an input-gradient conv, the residual of a forward one): that one's first half is requested up front and its latency
    // hides behind the output transform, the second half behind the first half's arithmetic; with several present they are fetched where
    // they are used.
    const int nopnd = (has_res ? 1 : 0) + (has_mask ? 1 : 0) + (accum ? 1 : 0);
    f32x4 opnd[2][8];
    auto fetch = [&](f32x4 (&dst)[8], __amdgpu_buffer_rsrc_t rs, unsigned base, int ld, int hf) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int a = 2 * hf + (q >> 2), b = q & 3;
        dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff(base, a, b), soff(ld, a, b), 0));
      }
    };
    auto fetch_single = [&](int hf) __attribute__((always_inline)) {
      if (has_res) fetch(opnd[hf], rr, base_r, p.ldres, hf);
      else if (has_mask) fetch(opnd[hf], rm, base_m, p.ldmask, hf);
      else fetch(opnd[hf], ry, base_y, p.ldy, hf);
    };
    if (nopnd == 1) fetch_single(0);   // (half 1 follows behind half 0's arithmetic: 32 registers less across the output transform)
    const f32x4 bias = *reinterpret_cast<const f32x4*>(sV + 0 * kBN4 + cl);
    const f32x4 vec2 = *reinterpret_cast<const f32x4*>(sV + 1 * kBN4 + cl);
    const f32x4 scale = *reinterpret_cast<const f32x4*>(sV + 2 * kBN4 + cl);
    const f32x4 shift = *reinterpret_cast<const f32x4*>(sV + 3 * kBN4 + cl);
    f32x4 cpre = {0.f, 0.f, 0.f, 0.f}, cpost = {0.f, 0.f, 0.f, 0.f};
    float yv[4][4][4];   // [a][b][r]
    if constexpr (SPLIT) {   // the sum of the partial tiles, in split order, four splits of loads in flight
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < p.nsplit; s0 += 4) {
          f32x4 l[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            l[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, s0 + j < p.nsplit ? slab_off(ob, q) + (unsigned)(s0 + j) * slab_bytes : kOobOffset, 0, 16));
#pragma unroll
          for (int j = 0; j < 4; ++j) v += l[j];   // (splits past nsplit read zeros: + 0.f is exact)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[q >> 2][q & 3][r] = v[r];
      }
    } else if (ob == 0 && nx.more) {   // (wave-uniform; the other half and the last tile of a workgroup: the plain form)
      out_transform(ob, yv, std::true_type{});
    } else {
      out_transform(ob, yv, std::false_type{});
    }
    // element-wise part: ONE PASS PER EPILOGUE FLAG over 8 outputs pixels at a time (a flag is tested once per half tile, not once per
    // element: every instruction here is matrix time lost)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (nopnd == 1 && hf == 0) fetch_single(1);
      f32x4 o[8];
      bool pix_ok[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int a = 2 * hf + (q >> 2), b = q & 3;
        o[q] = f32x4{yv[a][b][0], yv[a][b][1], yv[a][b][2], yv[a][b][3]};
        pix_ok[q] = ((okm >> (4 * a + b)) & 1u) != 0;
      }
      if (f & CRDR_EPI_BIAS) {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += bias;
      }
      if (f & CRDR_EPI_RELU) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[q][e] = fmaxf(o[q][e], 0.0f);
      }
      if (f & CRDR_EPI_LRELU) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[q][e] = o[q][e] > 0.0f ? o[q][e] : 0.2f * o[q][e];
      }
      if (f & CRDR_EPI_VEC2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += vec2;
      }
      if (has_res) {
        if (nopnd > 1) fetch(opnd[hf], rr, base_r, p.ldres, hf);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += opnd[hf][q];
      }
      if (f & CRDR_EPI_AFFINE) {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = o[q] * scale + shift;
      }
      if (do_cs) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) cpre[e] += pix_ok[q] ? o[q][e] : 0.f;
      }
      if (has_mask) {
        if (nopnd > 1) fetch(opnd[hf], rm, base_m, p.ldmask, hf);
        const f32x4 moff = (f & CRDR_EPI_MASKOFF) ? vec2 : f32x4{0.f, 0.f, 0.f, 0.f};
        if (f & CRDR_EPI_LRELUMASK) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[q][e] = (opnd[hf][q][e] - moff[e]) > 0.0f ? o[q][e] : 0.2f * o[q][e];
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[q][e] = (opnd[hf][q][e] - moff[e]) > 0.0f ? o[q][e] : 0.0f;
        }
      }
      if (do_cs) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) cpost[e] += pix_ok[q] ? o[q][e] : 0.f;
      }
      if (accum) {
        if (nopnd > 1) fetch(opnd[hf], ry, base_y, p.ldy, hf);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += opnd[hf][q];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int a = 2 * hf + (q >> 2), b = q & 3;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o[q]), ry, voff(base_y, a, b), soff(p.ldy, a, b), 0);
      }
    }
    if (do_cs) {
      *reinterpret_cast<f32x4*>(sC + (((wave * 2 + 0) * 2 + ob) * 64 + lane) * 4) = cpre;
      *reinterpret_cast<f32x4*>(sC + (((wave * 2 + 1) * 2 + ob) * 64 + lane) * 4) = cpost;
    }
  }
  if (do_cs) {
    // column sums of the tile, fixed order: channel c = 32 oh + 16 ob + 4 kg + r <- waves (th = 0, 1; oh), lanes kg * 16 + tx, tx = 0..15
    lds_barrier4();
    const int tid = wave * 64 + lane;
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63, oh2 = c >> 5, ob2 = (c >> 4) & 1, kg2 = (c >> 2) & 3, r = c & 3;
      float sum = 0.f;
      for (int th2 = 0; th2 < 2; ++th2)
        for (int t2 = 0; t2 < 16; ++t2) sum += sC[((((th2 * 2 + oh2) * 2 + which) * 2 + ob2) * 64 + kg2 * 16 + t2) * 4 + r];
      if (tl.n0 + c < p.Cout) p.cs[(((size_t)tl.patch * (p.so * p.so) + tl.phase) * 2 + which) * p.cs_ld + tl.n0 + c] = sum;
    }
  }
}

// DMA sources of tile tl (group pointers resolved)
template <int GEO, int FORM>
__device__ __forceinline__ Wino4Src wino4_src(const IgemmArgs& p, const IgemmGroup& grp, const Wino4Tile& tl, int gyn, const unsigned (&geo)[4]) {
  Wino4Src sr;
  const float* x = p.ngroup > 1 ? grp.x[tl.gidx] : p.x;
  sr.rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((((unsigned long long)p.N * p.H * p.W - 1) * p.ldx + p.Cin) * 4ull), 0x00020000);
  // transformed filters of group gidx: [N tile][chunk][2304 slots of 16 B]
  const int nph = p.so * p.so;
  const size_t ublock = (size_t)gyn * nph * p.kchunks * kUSlots4 * 4;   // floats per group: [N tile][output phase][sub-step][2304 slots]
  sr.ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w) + (size_t)tl.gidx * ublock, 0, (unsigned)(ublock * 4), 0x00020000);
  sr.u_off0 = (unsigned)(tl.tn * nph + tl.phase) * (unsigned)p.kchunks * (kUSlots4 * 16u);
  const int sub0 = FORM == 1 ? tl.kbeg / (p.kchunks / p.nphase) : 0;   // (the shifted sub-filter this work item starts in)
#pragma unroll
  for (int j = 0; j < 4; ++j) sr.a_off[j] = wino4_in_off<GEO, FORM>(p, tl, geo[j], sub0);
  return sr;
}

// per-column epilogue vectors of tile tl -> sV[4][64] (bias, vec2, scale, shift), in two halves: the loads are issued, something else is done
// while they are in flight (the next tile's source set-up), then they go to LDS
struct Wino4Vec { float v[4]; };
__device__ __forceinline__ Wino4Vec wino4_vectors_load(const IgemmArgs& p, const IgemmGroup& grp, const Wino4Tile& tl, int tid) {
  Wino4Vec r = {{0.f, 0.f, 1.f, 0.f}};
  if (tid < kBN4) {
    const int f0 = p.flags;
    const bool live = tl.n0 + tid < p.Cout;
    const float* bias = p.ngroup > 1 ? grp.bias[tl.gidx] : p.bias;
    if (live && (f0 & CRDR_EPI_BIAS)) r.v[0] = bias[tl.n0 + tid];
    if (live && (f0 & (CRDR_EPI_VEC2 | CRDR_EPI_MASKOFF))) r.v[1] = p.vec2[tl.n0 + tid];
    if (live && (f0 & CRDR_EPI_AFFINE)) {
      r.v[2] = p.scale[tl.n0 + tid];
      r.v[3] = p.shift[tl.n0 + tid];
    }
  }
  return r;
}
__device__ __forceinline__ void wino4_vectors_store(const Wino4Vec& r, float* sV, int tid) {
  if (tid < kBN4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sV[k * kBN4 + tid] = r.v[k];
  }
}

// Persistent: at most one workgroup per CU, each walks the tiles vb = blockIdx.x, + gridDim.x, ...  Between the K loop and the
// epilogue of a tile the waves request the next tile's first raw patches and filter block (the buffers are free by then) and its
// epilogue vectors: the DMA latency of a fresh tile and the memory latency of the stores hide behind each other.
template <int GEO, int FORM, bool SPLIT, int EC>
__global__ __launch_bounds__(kNT4) void wino4_kernel(const IgemmArgs p_, const IgemmGroup grp, int gx, int gyn, int gz) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane_ = threadIdx.x & 63, wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gyn_abs = gyn < 0 ? -gyn : gyn;
  const int total = gx * gyn_abs * gz * p_.so * p_.so * (SPLIT ? p_.nsplit : 1);
  float* sVb = smem + kLdsFloats4;   // [2][4][64]: bias, vec2, scale, shift of the current / the next tile
  int cur = 0;
  bool prefetched = false;
  Wino4Src sr;
  unsigned geo[4];   // this lane's slots of the four raw pieces of its wave (tile independent)
#pragma unroll
  for (int j = 0; j < 4; ++j) geo[j] = wino4_piece_geo<GEO>(wave_ + 4 * j, lane_);
  for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
    // (per-lane constants are re-derived per tile instead of staying live across the register-hungry epilogue)
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(wave));
    const int tid = wave * 64 + lane;
    const Wino4Tile tl = wino4_tile<GEO>(p_, vb, gx, gyn, gz);
    IgemmArgs p = p_;
    if (p.ngroup > 1) {
      const int g = tl.gidx;
      p.x = grp.x[g]; p.y = grp.y[g]; p.bias = grp.bias[g]; p.mask = grp.mask[g]; p.res = grp.res[g]; p.cs = grp.cs[g];
    }
    float* sV = sVb + cur * (4 * kBN4);
    if (!prefetched) {
      const Wino4Vec vec = wino4_vectors_load(p_, grp, tl, tid);
      sr = wino4_src<GEO, FORM>(p_, grp, tl, gyn_abs, geo);
      wino4_vectors_store(vec, sV, tid);   // (published by the K loop's first barrier)
    }
    f32x4 acc[64], accv[8];
#pragma unroll
    for (int j = 0; j < 64; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) accv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    wino4_loop<GEO, FORM>(p, tl, smem, sr, prefetched, lane, wave, acc, accv, geo);
    // (the loop ends with a barrier: every wave is past its last LDS read, raw and filter buffers are free)
    const bool more = vb + (int)gridDim.x < total;
    Wino4Next nx{false, 0, 0};
    if (more) {   // the next tile: raw patches 0, 1, 2, filter block 0 and the epilogue vectors
      const Wino4Tile tn = wino4_tile<GEO>(p_, vb + (int)gridDim.x, gx, gyn, gz);
      // (the vectors' loads FIRST -- vector-memory operations retire in issue order: behind the next tile's requests their wait would sit out
      // the whole DMA latency, 9 000 cycles per tile by the stamps -- and the source set-up while they are in flight)
      const Wino4Vec vec = wino4_vectors_load(p_, grp, tn, tid);
      sr = wino4_src<GEO, FORM>(p_, grp, tn, gyn_abs, geo);
      wino4_vectors_store(vec, sVb + (cur ^ 1) * (4 * kBN4), tid);
      nx = Wino4Next{true, tn.kbeg, tn.kcnt};
    }
    wino4_finish<GEO, FORM, SPLIT, w4_epi_mask(EC)>(p, tl, smem, sV, lane, wave, acc, accv, sr, nx);
    prefetched = more;
    cur ^= 1;
    lds_barrier4();   // column-sum area, sV of this tile: free (the stores stay in flight: the next tile's first wait is vmcnt(32))
  }
}

// Filter transform U = G g G^T, G = rows [1, p_j, p_j^2] / N_j at the points 0, +-a, +-b and [0, 0, 1] (wino4_xform.hpp), evaluated in double
// and rounded once, from the implicit-GEMM weight pack (tap-major [tap][wrows][wcols]) into the block layout of wino4_kernel:
// [N tile of 64][chunk of 4 channels][position 36][channel 4][oh 2][tx 16][ob 2], output channel = 32 oh + 16 ob + tx.
struct Wino4Taps { int widx[4][9]; };   // [variant][3 a + b]: weight-pack tap of sub-filter element (a, b), -1 = zero
// one thread: (unit blk = (N tile, variant, chunk), oc, channel c of the chunk); variant = parity sub-filter of a stride-2 conv (blocks of one
// tile: [sub][chunk] ... the K loop walks them) or output phase of a stride-2 transposed conv ([phase][chunk] as well, each phase being a
// tile of its own); consecutive threads read consecutive input channels of one weight-pack row
__device__ __forceinline__ void wino4_filter_thread(const float* w, float* ug, long long blk, int t256, int Cin, int Cout, int wrows, int wcols,
                                                    int kchunks, int nvar, const int (&widx)[4][9]) {
  const int c4 = t256 & 3, oc64 = (t256 >> 2) & 63;
  const int kc = (int)(blk % kchunks), var = (int)((blk / kchunks) % nvar), ct = (int)(blk / ((long long)kchunks * nvar));
  const int oc = ct * kBN4 + oc64, c = kc * 4 + c4;
  const bool live = oc < Cout && c < Cin;
  double g9[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int wi = widx[var][a * 3 + b];
      g9[a][b] = (live && wi >= 0) ? (double)w[((size_t)wi * wrows + oc) * wcols + c] : 0.0;
    }
  const double wa = kWa, wb = kWb;
  const double G[6][3] = {{1.0 / kWN0, 0.0, 0.0}, {1.0 / kWNa, wa / kWNa, wa * wa / kWNa}, {1.0 / kWNa, -wa / kWNa, wa * wa / kWNa},
                          {1.0 / kWNb, wb / kWNb, wb * wb / kWNb}, {1.0 / kWNb, -wb / kWNb, wb * wb / kWNb}, {0.0, 0.0, 1.0}};
  float* dst = ug + (size_t)blk * (kUSlots4 * 4) + (size_t)c4 * kBN4 + (oc64 >> 5) * 32 + (oc64 & 15) * 2 + ((oc64 >> 4) & 1);
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    double t[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) t[b] = G[xi][0] * g9[0][b] + G[xi][1] * g9[1][b] + G[xi][2] * g9[2][b];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) dst[(size_t)(xi * 6 + nu) * 256] = (float)(G[nu][0] * t[0] + G[nu][1] * t[1] + G[nu][2] * t[2]);
  }
}
__global__ void wino4_filter_kernel(const IgemmGroup grp, int ngroup, const float* w0, float* u, int Cin, int Cout, int wrows, int wcols, int kchunks,
                                    int ntile, int nvar, int var_inner, Wino4Taps tp) {
  const long long total = (long long)ntile * nvar * kchunks * 256;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  (void)var_inner;
  const int g = blockIdx.y;
  const float* w = ngroup > 1 ? grp.w[g] : w0;
  wino4_filter_thread(w, u + (size_t)g * ntile * nvar * kchunks * (kUSlots4 * 4), id >> 8, (int)(id & 255), Cin, Cout, wrows, wcols, kchunks, nvar, tp.widx);
}
// every filter cache of an optimiser in one launch (crdr_w4_filters_batched): a workgroup = one unit (256 threads) at a time, units dealt
// round robin over a persistent grid; the item of a unit by binary search over the prefix table (block-uniform)
__global__ __launch_bounds__(256) void wino4_filter_batched_kernel(const crdr_w4_filter_item* items, const long long* prefix, const long long* meta) {
  const int n = (int)meta[0];
  const long long total = meta[1];
  for (long long gu = blockIdx.x; gu < total; gu += gridDim.x) {
    int lo = 0, hi = n - 1;   // last item with prefix[item] <= gu
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= gu) lo = mid; else hi = mid - 1;
    }
    const crdr_w4_filter_item& it = items[lo];
    const long long ul = gu - prefix[lo];
    const long long per = (long long)it.ntile * it.nvar * it.kchunks;   // units of one problem of the group
    const int g = (int)(ul / per);
    wino4_filter_thread(it.w[g], it.u + (size_t)g * per * (kUSlots4 * 4), ul - (long long)g * per, (int)threadIdx.x, it.Cin, it.Cout, it.wrows, it.wcols,
                        it.kchunks, it.nvar, it.widx);
  }
}

}  // namespace

// the forms the kernel takes: 0 = none, 1 = 3x3 stride 1 (conv or its transposed twin), 2 = 5x5 stride-2 conv (pad 2, even H / W: four
// parity sub-filters of 3x3 accumulated), 3 = 5x5 stride-2 transposed conv (pad 2, output = 2 x input: four output phases, each a 3x3
// stride-1 conv of the input -- elic_layers.py:14-21 up_conv, elic_autoencoder.py:42-52 and their input gradients), 4 = 5x5 stride 1 pad 2
// (conv or transposed twin: four 3x3 sub-filters over the taps 3 bi + a, 3 bj + b, the last row / column of the padded 6x6 being zero,
// accumulated over patches displaced by (3 bi, 3 bj) -- the slice transforms of the context model,
// minnen20_charm_context_model.py:26-38)
static int wino4_mode(const crdr_conv_desc* d) {
  if (d->wlayout != 0) return 0;
  if (d->kh == 3 && d->kw == 3 && d->stride == 1) {
    const int grow = d->transposed ? 2 - 2 * d->pad : 2 * d->pad - 2;   // (a stride-1 transposed conv = a conv with pad k - 1 - pad)
    return (d->OH == d->H + grow && d->OW == d->W + grow && d->pad >= 0 && d->pad <= 2) ? 1 : 0;
  }
  if (d->kh == 5 && d->kw == 5 && d->stride == 1 && d->pad == 2) return (d->OH == d->H && d->OW == d->W && d->C >= 12) ? 4 : 0;
  if (d->kh == 5 && d->kw == 5 && d->stride == 2 && d->pad == 2) {
    if (!d->transposed) return (d->H % 2 == 0 && d->W % 2 == 0 && d->OH == d->H / 2 && d->OW == d->W / 2) ? 2 : 0;
    return (d->OH == 2 * d->H && d->OW == 2 * d->W) ? 3 : 0;
  }
  return 0;
}

// tile geometry of a launch (W4Geo): by the width of the grid the tiles cover -- the output, or one output phase
// (geometry 2 -- two whole images of <= 16 x 16 per tile -- for the stride-1 forms only)
static int wino4_geo(const crdr_conv_desc* d, int mode) {
  const int gh = mode == 3 ? d->H : d->OH, gw = mode == 3 ? d->W : d->OW;
  if ((mode == 1 || mode == 4) && gh <= 16 && gw <= 16) return 2;
  return gw <= 32 ? 1 : 0;
}
static int wino4_tile_rows(int geo) { return geo == 0 ? 8 : 16; }
static int wino4_tile_cols(int geo) { return geo == 0 ? 64 : (geo == 1 ? 32 : 16); }
// output tiles of a launch (per N tile and output phase) = rows of its column-sum partials
static int wino4_patches(const crdr_conv_desc* d, int mode) {
  const int geo = wino4_geo(d, mode), gh = mode == 3 ? d->H : d->OH, gw = mode == 3 ? d->W : d->OW;
  if (geo == 2) return cdiv(d->N, 2);
  return d->N * cdiv(gh, wino4_tile_rows(geo)) * cdiv(gw, wino4_tile_cols(geo));
}

static size_t wino4_filter_bytes(const crdr_conv_desc* d, int G) {
  const int nvar = wino4_mode(d) >= 2 ? 4 : 1;   // parity or shifted sub-filters / output phases
  return (size_t)G * cdiv(d->OC, kBN4) * nvar * cdiv(d->C, 4) * kUSlots4 * 16;
}
// tiles of a launch (each the ticket of its K splits) and bytes of one tile's published partial result
static long long wino4_tiles(const crdr_conv_desc* d, int G) {
  const int mode = wino4_mode(d);
  return (long long)wino4_patches(d, mode) * cdiv(d->OC, kBN4) * G * (mode == 3 ? 4 : 1);
}
constexpr size_t kTileSlabBytes = 4 * 2 * 16 * 64 * 16;

// K splits (1 = none): the K range of a tile -- p.kchunks sub-steps of 4 channels, sub-filters included -- in `nsplit` equal parts
bool wino4_split_ok(const crdr_conv_desc* d, int G, int nsplit) {
  if (nsplit == 1) return true;
  if (d->flags & CRDR_CONV_NOSPLIT) return false;   // (the caller's workspace has no zeroed ticket head)
  const int mode = wino4_mode(d);
  if (!mode || nsplit < 1 || nsplit > 16) return false;
  const int nph = (mode == 2 || mode == 4) ? 4 : 1, Kc = cdiv(d->C, 4), K4 = Kc * nph, cnt = cdiv(K4, nsplit);
  if (K4 - (nsplit - 1) * cnt < 1) return false;                       // every split has work
  const long long tiles = wino4_tiles(d, G);
  if (tiles > CRDR_CONV_TICKETS || tiles * nsplit * (long long)kTileSlabBytes >= (1ll << 31)) return false;
  if (mode == 4)   // a work item's first three patches share one shifted sub-filter (their lane offsets are derived once)
    for (int ks = 1; ks < nsplit; ++ks)
      if ((ks * cnt) % Kc + 3 > Kc && (ks * cnt) % Kc != 0) return false;
  return true;
}

size_t wino4_workspace(const crdr_conv_desc* d, int G, int nsplit) {
  return wino4_filter_bytes(d, G) + (nsplit > 1 ? (size_t)wino4_tiles(d, G) * nsplit * kTileSlabBytes : 0);
}

bool wino4_eligible(const crdr_conv_desc* d, int G, bool vec_ok) {
  const int mode = wino4_mode(d);
  if (!mode) return false;
  if (d->C % 4 != 0 || d->ldx % 4 != 0 || d->OC % 4 != 0 || d->ldy % 4 != 0) return false;
  {   // the 8 x 64 / 16 x 32 tile (of the output, or of one output phase) wants wide images; two-image tiles want both dimensions >= 9
    const int gh = mode == 3 ? d->H : d->OH, gw = mode == 3 ? d->W : d->OW;
    if (wino4_geo(d, mode) == 2 ? (gh < 9 || gw < 9) : gw < 24) return false;
  }
  if ((d->flags & CRDR_EPI_RES) && d->ldres % 4 != 0) return false;
  if ((d->flags & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) && d->ldmask % 4 != 0) return false;
  if (d->flags & (CRDR_EPI_GATE | CRDR_EPI_PREADD | CRDR_CONV_BF16X3)) return false;
  if (G > 1 && (d->flags & (CRDR_EPI_VEC2 | CRDR_EPI_AFFINE | CRDR_EPI_MASKOFF))) return false;
  if (!vec_ok) return false;
  const long long img = ((long long)d->N * d->H + 8) * d->W * d->ldx * 4;   // one descriptor spans a whole tensor
  const long long oimg = (long long)d->N * d->OH * d->OW * std::max(std::max(d->ldy, d->ldres), d->ldmask) * 4;
  if (img >= (1ll << 31) || oimg >= (1ll << 31)) return false;
  if ((long long)wino4_filter_bytes(d, G) / G >= (1ll << 31)) return false;
  return true;
}


int wino4_colsum_rows(const crdr_conv_desc* d) {
  const int mode = wino4_mode(d);
  return wino4_patches(d, mode) * (mode == 3 ? 4 : 1);
}

// which weight-pack tap feeds element (a, b) of sub-filter / phase v, and how far above / left of its first output pixel the patch starts
static int wino4_taps(const crdr_conv_desc* d, const IgemmTaps& taps, Wino4Taps& wt, int& si) {
  const int mode = wino4_mode(d);
  for (int v = 0; v < 4; ++v)
    for (int t = 0; t < 9; ++t) wt.widx[v][t] = -1;
  si = 1;
  if (mode == 1) {
    int dmin = 127;
    for (int t = 0; t < 9; ++t) dmin = std::min(dmin, (int)(signed char)(taps.packed[t] & 0xff));
    for (int t = 0; t < 9; ++t) {
      const int v = taps.packed[t];
      const int dh = (int)(signed char)(v & 0xff) - dmin, dw = (int)(signed char)((v >> 8) & 0xff) - dmin;
      CRDR_REQUIRE(dh >= 0 && dh < 3 && dw >= 0 && dw < 3, "conv2d: Winograd F(4x4): tap offsets are not a 3x3 window");
      wt.widx[0][dh * 3 + dw] = v >> 16;
    }
    for (int t = 0; t < 9; ++t) CRDR_REQUIRE(wt.widx[0][t] >= 0, "conv2d: Winograd F(4x4): incomplete 3x3 window");
    si = -dmin;   // the patch starts `si` pixels above / left of its first output pixel
  } else if (mode == 4) {
    // taps (dh, dw) relative to the window's first: sub-filter (bi, bj) element (a, b) = tap (3 bi + a, 3 bj + b), absent past the 5th
    int dmin = 127;
    for (int t = 0; t < 25; ++t) dmin = std::min(dmin, (int)(signed char)(taps.packed[t] & 0xff));
    for (int t = 0; t < 25; ++t) {
      const int v = taps.packed[t];
      const int dh = (int)(signed char)(v & 0xff) - dmin, dw = (int)(signed char)((v >> 8) & 0xff) - dmin;
      CRDR_REQUIRE(dh >= 0 && dh < 5 && dw >= 0 && dw < 5, "conv2d: Winograd F(4x4): tap offsets are not a 5x5 window");
      wt.widx[(dh / 3) * 2 + dw / 3][(dh % 3) * 3 + dw % 3] = v >> 16;
    }
    si = -dmin;
  } else if (mode == 2) {
    // out[o] = sum_t w[t] x[2 o - 2 + t], t = 2 a + p: sub-filter (ph, pw) element (a, b) = w[2 a + ph][2 b + pw] over the parity plane
    // x[2 m + ph], a 3-tap 'pad 1' correlation; the pack's tap index of kernel element (r, s) is 5 r + s
    for (int sub = 0; sub < 4; ++sub)
      for (int a2 = 0; a2 < 3; ++a2)
        for (int b2 = 0; b2 < 3; ++b2) {
          const int r = 2 * a2 + (sub >> 1), c = 2 * b2 + (sub & 1);
          if (r < 5 && c < 5) wt.widx[sub][a2 * 3 + b2] = r * 5 + c;
        }
  } else {
    // out[2 u + py] = sum_{a'} w[2 (2 - a') + py] in[u - 1 + a']: phase (py, px) element (a', b') = w[2 (2 - a') + py][2 (2 - b') + px]
    for (int ph = 0; ph < 4; ++ph)
      for (int a2 = 0; a2 < 3; ++a2)
        for (int b2 = 0; b2 < 3; ++b2) {
          const int r = 2 * (2 - a2) + (ph >> 1), c = 2 * (2 - b2) + (ph & 1);
          if (r < 5 && c < 5) wt.widx[ph][a2 * 3 + b2] = r * 5 + c;
        }
  }
  return 0;
}

// description of one filter cache for the batched rebuild (crdr_conv2d_filter_item): everything but the pointers
int wino4_filter_item(const crdr_conv_desc* d, const IgemmTaps& taps, int G, crdr_w4_filter_item* it) {
  CRDR_REQUIRE(wino4_eligible(d, G, true), "conv2d_filter_item: not a convolution the F(4x4, 3x3) kernel takes");
  Wino4Taps wt;
  int si = 1;
  if (int rc = wino4_taps(d, taps, wt, si)) return rc;
  const int mode = wino4_mode(d);
  for (int g = 0; g < CRDR_MAX_GROUP; ++g) it->w[g] = nullptr;
  it->u = nullptr;
  it->G = G; it->Cin = d->C; it->Cout = d->OC; it->wrows = d->wrows; it->wcols = d->wcols;
  it->kchunks = cdiv(d->C, 4); it->ntile = cdiv(d->OC, kBN4); it->nvar = mode >= 2 ? 4 : 1;
  for (int v = 0; v < 4; ++v)
    for (int t = 0; t < 9; ++t) it->widx[v][t] = wt.widx[v][t];
  it->units = (long long)G * it->ntile * it->nvar * it->kchunks;
  return 0;
}

int wino4_filters_batched(const crdr_w4_filter_item* items, const long long* prefix, const long long* meta, hipStream_t s) {
  hipLaunchKernelGGL(wino4_filter_batched_kernel, dim3(4096), dim3(256), 0, s, items, prefix, meta);
  CRDR_CHECK_LAUNCH("wino4_filter_batched_kernel");
  return 0;
}

int wino4_launch(const crdr_conv_desc* d, IgemmArgs a, const IgemmTaps& taps, const IgemmGroup& grp, int G, float* u, float* slabs, int nsplit,
                 bool filters_ready, hipStream_t s) {
  CRDR_REQUIRE(wino4_eligible(d, G, a.vec_epi != 0), "conv2d: the F(4x4, 3x3) Winograd kernel takes 3x3 / 5x5 stride-1 and 5x5 stride-2 (pad 2) convolutions of >= 24 "
               "output (phase) columns (or whole images of 9..16 pixels a side) with C, OC %% 4 == 0, 16-byte aligned operand rows and no gate / pre-add epilogue");
  const int mode = wino4_mode(d);
  Wino4Taps wt;
  int si = 1;
  if (int rc = wino4_taps(d, taps, wt, si)) return rc;
  const int nvar = mode >= 2 ? 4 : 1;
  const int ntile = cdiv(d->OC, kBN4), kchunks = cdiv(d->C, 4);
  if (!filters_ready) {   // (a caller that kept the transformed filters of these weights from an earlier launch skips this)
    const long long total = (long long)ntile * nvar * kchunks * 256;
    hipLaunchKernelGGL(wino4_filter_kernel, dim3((unsigned)cdiv64(total, 256), G), dim3(256), 0, s, grp, G, a.w, u, d->C, d->OC, d->wrows, d->wcols, kchunks,
                       ntile, nvar, 0, wt);
    CRDR_CHECK_LAUNCH("wino4_filter_kernel");
  }
  a.w = u;
  a.nsplit = nsplit;
  a.ws = slabs;   // partial tiles of a split launch
  a.ws_ld = (int)wino4_tiles(d, G);
  a.nphase = (mode == 2 || mode == 4) ? 4 : 1;   // parity / shifted sub-filters the K loop accumulates
  a.kchunks = kchunks * a.nphase;             // sub-steps of a tile
  a.so = mode == 3 ? 2 : 1;                   // output stride (4 output phases = 4 tiles per patch and N tile)
  const int gh = mode == 3 ? d->H : d->OH, gw = mode == 3 ? d->W : d->OW;   // the grid the 8 x 64 tiles cover
  const int geo = wino4_geo(d, mode);
  a.GH = geo == 2 ? 1 : cdiv(gh, wino4_tile_rows(geo));
  a.GW = geo == 2 ? 1 : cdiv(gw, wino4_tile_cols(geo));
  a.si = si;
  a.cs_rows = wino4_colsum_rows(d);
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    return n / 8 * 8;
  }();
  const int gx = wino4_patches(d, mode);
  const int total = gx * ntile * G * a.so * a.so;
  using Kern = void (*)(const IgemmArgs, const IgemmGroup, int, int, int);
#define W4_ROW(SP, F, E) {wino4_kernel<0, F, SP, E>, wino4_kernel<1, F, SP, E>, wino4_kernel<2, F, SP, E>}
#define W4_CLS(E) {{W4_ROW(false, 0, E), W4_ROW(false, 1, E)}, {W4_ROW(true, 0, E), W4_ROW(true, 1, E)}}
  static const Kern kerns[4][2][2][3] = {W4_CLS(0), W4_CLS(1), W4_CLS(2), W4_CLS(3)};   // [epilogue class][split][form][geometry]
#undef W4_CLS
#undef W4_ROW
  static std::atomic<bool> attr_done;
  if (!attr_done.load(std::memory_order_acquire)) {
    for (int ec = 0; ec < 4; ++ec)
      for (int sp = 0; sp < 2; ++sp)
        for (int f = 0; f < 2; ++f)
          for (int g = 0; g < 3; ++g)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[ec][sp][f][g]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.store(true, std::memory_order_release);
  }
  const size_t lds = (size_t)(kLdsFloats4 + 2 * 4 * kBN4 + 4) * sizeof(float);   // (+ the split-K ticket flag)
  // tile order by a traffic estimate (each XCD has its own 4 MB L2; an XCD's ~32 concurrent workgroups walk consecutive tiles).  Patch-major: the
  // input once, but every round of resident workgroups touches every filter block of the launch again on every XCD unless they all fit L2
  // together.  Filter-stationary: the filters once, the input once per (N tile, phase).
  const double u_bytes = (double)ntile * a.so * a.so * a.kchunks * (kUSlots4 * 16.0), x_bytes = (double)d->N * d->H * d->W * d->C * 4.0;
  const double ncombo = (double)ntile * a.so * a.so, rounds = std::max(1.0, (double)total * nsplit / std::max(ncu, 1));
  const double est_patch = x_bytes + 8.0 * u_bytes * (u_bytes > 3.0e6 ? rounds : 1.0), est_fstat = u_bytes + ncombo * x_bytes;
  static const int force_order = [] { const char* e = getenv("CRDR_W4_ORDER"); return e ? atoi(e) : -1; }();   // experiments: 0 patch-major, 1 filter-stationary
  const int gyn_arg = (force_order >= 0 ? force_order == 1 : est_fstat < est_patch) ? -ntile : ntile;
  hipLaunchKernelGGL(kerns[w4_epi_class(a.flags)][nsplit > 1 ? 1 : 0][mode == 4 ? 1 : 0][geo], dim3(std::min(total * nsplit, ncu)), dim3(kNT4), lds, s, a, grp, gx, gyn_arg, G);
  CRDR_CHECK_LAUNCH("wino4_kernel");
  return 0;
}


}  // namespace crdr
