// Implicit-GEMM convolution on CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
//   out[pix][oc] = epilogue( sum_{tap, c} in[gather(pix, tap)][c] * wpack[tap][oc][c] )
//
// One kernel covers Conv2d forward, ConvTranspose2d forward and both input gradients: a launch is a set of
// "phases" (sub-pixel output grids); every phase owns the taps that can reach it, so a stride-2 transposed
// conv does only the (k/2)^2-ish taps per output instead of zero-stuffing.
//
// Tiling: block = WM x WN waves (one wave per SIMD), wave tile = (32 MB) x (32 NB) made of 32x32 MFMA blocks,
// K-tile = 32 input channels of one tap.  A (gathered activations) and B (packed weights) K-tiles are both
// "rows of 128 B": staged global -> registers -> LDS (zero fill for padding taps), double buffered, one
// barrier per K-tile.  Each lane reads 16 B (4 consecutive k) per 32-row fragment and feeds 4 MFMAs: lane half
// h owns k = 4h..4h+3 of every 8-wide k group, identically for A and B, so the sum over k is complete.
// Split-K (blockIdx.z): every split publishes its raw partial tile (write-through stores), takes a ticket on the tile's
// counter, and the workgroup that arrives last adds the slabs in split order and runs the epilogue -- one launch, and the
// result does not depend on which split came last.

#include <algorithm>
#include <atomic>
#include <type_traits>

#pragma once

#include "common.hpp"
#include "igemm_args.hpp"


namespace crdr {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// epilogue on one element; opix = output pixel index, oc = channel
__device__ __forceinline__ void epilogue_store(const IgemmArgs& p, size_t opix, int oc, float v, float& vpre, float& vpost) {
  const int f = p.flags;
  if (f & CRDR_EPI_PREADD) v += p.pre[opix * p.ldpre + oc];
  if (f & CRDR_EPI_BIAS) v += p.bias[oc];
  if (f & CRDR_EPI_RELU) v = fmaxf(v, 0.0f);
  if (f & CRDR_EPI_LRELU) v = v > 0.0f ? v : 0.2f * v;
  if (f & CRDR_EPI_VEC2) v += p.vec2[oc];
  if (f & CRDR_EPI_RES) v += p.res[opix * p.ldres + oc];
  if (f & CRDR_EPI_GATE) {
    const float s = 1.0f / (1.0f + expf(-v));
    p.sig[opix * p.ldg + oc] = s;
    v = p.gx[opix * p.ldg + oc] + p.gt[opix * p.ldg + oc] * s;
  }
  if (f & CRDR_EPI_AFFINE) v = v * p.scale[oc] + p.shift[oc];
  vpre = v;
  if (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) {
    const float mv = p.mask[opix * p.ldmask + oc] - ((f & CRDR_EPI_MASKOFF) ? p.vec2[oc] : 0.0f);
    v = mv > 0.0f ? v : ((f & CRDR_EPI_LRELUMASK) ? 0.2f * v : 0.0f);
  }
  vpost = v;
  float* dst = p.y + opix * p.ldy + oc;
  if (f & CRDR_EPI_ACCUM) v += *dst;
  *dst = v;
}

// GEMM row m -> (image n, grid row a, grid column b).  gfx950 has no integer divide: a division by a runtime value is ~25 scalar or ~20
// vector instructions behind a v_rcp_f32, and a tile used to run 8 of them per thread in front of its first request plus two per 16-byte
// output group in the epilogue of a phased (stride-2 transposed) launch -- on the discriminator's k3 s2 input gradients, whose phases have K
// loops of 2 .. 8 K-tiles, that was most of the tile's time (PMC: waves 26 % "active" at 36 % MFMA busy).  Power-of-two grids take shifts
// (the branch is workgroup-uniform).
__device__ __forceinline__ void split_row(const IgemmArgs& p, int m, int& n, int& a, int& b) {
  if (p.hw_sh >= 0) {
    n = m >> p.hw_sh;
    const int rem = m & ((1 << p.hw_sh) - 1);
    a = rem >> p.gw_sh;
    b = rem & ((1 << p.gw_sh) - 1);
  } else {
    const int hw = p.GH * p.GW;
    n = m / hw;
    const int rem = m - n * hw;
    a = rem / p.GW;
    b = rem - a * p.GW;
  }
}

// Staging is LDS-DMA: `buffer_load_dwordx4 ... lds` moves 16 bytes per lane straight from global memory into LDS
// (1 KiB = 8 tile rows per wave instruction, lane-linear destination), so the K loop carries no staging registers,
// no ds_write pass and no zero-fill selects: rows / channel chunks / taps that fall outside the tensor get a byte
// offset beyond the buffer descriptor's range and the hardware range check returns zeros for them.  The XOR swizzle
// of the tile image (lds_off) is applied on the SOURCE side: the lane that fills slot s of row r fetches chunk
// s ^ ((r >> 1) & 7).

// K-loop stages of a tile configuration (host and kernel agree through this): see the kernel.
// BF3: the products run as split-bf16 triples on the bf16 matrix path (common.hpp, split_bf16x8): same staging, same K order,
// same epilogues -- only the fragment-to-MFMA step differs.
// FAST: the launch is unsplit and takes the straight-line epilogue (p.fast_epi, p.nsplit == 1 -- the host checks): the split-K hand-off and
// the general epilogue are not compiled in.  The full kernel is ~16 k instructions (~125 KB) beside a 64 KB instruction cache shared by two
// CUs; a tile walks its prologue and epilogue once, so every tile re-fetched most of what it executed outside the K loop -- a constant
// ~6.8 us of CU time per tile whatever its K (C 64->64 k3 s2: 22.9 us per tile for 18 K-tiles of 0.9 us; its transposed twin with four
// times the tiles: the same 6.7 us per tile), i.e. most of the time of every launch whose tiles have short K loops.
// Register budget of the 8-wave configurations with two accumulator blocks per wave (128 x 128 and 256 x 64 tiles): two workgroups per CU =
// four waves per SIMD = 128 registers.  They sat at 121; the row-split branches took the allocator to 133-135 and the second workgroup
// away (the bf16x3 step, which lives on these tiles, went from 241 to 219 img/s) -- the bound is stated instead of hoped for.
// PREC = 6 ("bf16x6", common.hpp): fp32-equivalent products on the bf16 matrix path.  At 3/8 of the fp32 matrix time a K-tile lasts ~0.8 us per
// workgroup, well under the loaded-memory latency, and an LDS landing zone deep enough for that does not fit beside the operands -- so this loop
// stages global -> REGISTERS -> LDS: a thread requests its pieces of tile t + 2 into one of two register sets (plain range-checked buffer loads:
// 2-3 us of lead), splits the set that holds tile t + 1 into three bf16 pieces between the MFMAs of tile t, and writes them to the other of TWO LDS
// stages of three bf16 planes each (hi / mid / lo, 192 bytes per tile row and K-tile).  A thread owns a CHUNK PAIR (8 consecutive k of one row):
// its three 16-byte results are exactly what a lane of v_mfma_f32_32x32x16_bf16 feeds -- three ds_read_b128 per fragment, no register shuffles.
// Tiles whose rows are not whole passes of NT / 4 rows or whose two stages pass 160 KiB have no bf16x6 form (igemm_bf6_ok).
constexpr int igemm_min_waves(int waves, int blocks, int prec) { return prec == 6 ? (waves == 8 ? 2 : 1) : ((waves == 8 && blocks <= 2) ? 4 : 1); }
// floats of LDS in front of the per-column vectors (host and kernel agree through this)
constexpr int igemm_staging_floats(int BM, int BN, int prec) { return prec == 6 ? 2 * (BM + BN) * 48 + 132 : 2 * (BM + BN) * 32 + 132; }
constexpr bool igemm_bf6_ok(int wm, int wn, int mb, int nb) {
  const int BM = 32 * wm * mb, BN = 32 * wn * nb, R = 16 * wm * wn;
  return BM % R == 0 && BN % R == 0 && (igemm_staging_floats(BM, BN, 6) + 4 * BN + 8) * 4 <= 160 * 1024;
}
template <int WM, int WN, int MB, int NB, bool SMALLC, int PREC = 0, bool FAST = false>
__global__ __launch_bounds__(64 * WM * WN, igemm_min_waves(WM * WN, MB * NB, PREC)) void igemm_kernel(const IgemmArgs p_, const IgemmTaps tp, const IgemmGroup grp) {
  static_assert(PREC == 0 || PREC == 3 || PREC == 6, "PREC: 0 exact fp32, 3 split-bf16 triples, 6 split-bf16 sextuples");
  constexpr bool BF3 = PREC == 3, BF6 = PREC == 6;
  static_assert(!BF6 || (!SMALLC && !FAST && igemm_bf6_ok(WM, WN, MB, NB)), "no bf16x6 form of this configuration");
  constexpr int BM = 32 * WM * MB, BN = 32 * WN * NB, NT = 64 * WM * WN;
  constexpr int AV = BM * 8 / NT, BV = BN * 8 / NT;  // 16-byte pieces per thread per K-tile
  static_assert(AV * NT == BM * 8 && BV * NT == BN * 8, "tile/threads mismatch");
  constexpr int RPP = NT / 8;  // tile rows filled by one pass of the whole block
  static_assert(RPP % 16 == 0, "the swizzle term must not depend on the pass");

  constexpr int ST = 2;   // LDS stages of the K loop (a 3-4 deep ring with counted vmcnt waits was measured slower, also for BF3)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                                        // [ST][BM*32]
  float* sB = smem + ST * BM * 32;                         // [ST][BN*32]
  int* sTap = reinterpret_cast<int*>(smem + (BF6 ? 2 * (BM + BN) * 48 : ST * (BM + BN) * 32));  // [<=132] packed (dh, dw, widx) of this phase

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the DMA destinations and fragment bases derived from it then cost no vector instructions)
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: the hardware deals workgroups round-robin over the 8 XCDs (each with its own L2) in linear
  // block-id order; remap so that every XCD walks a CONTIGUOUS range of tiles ordered (M tile, phase/split, N tile):
  // the N tiles, phases and K splits that read the same activation rows then run back to back on one L2.
  int tile_m, tile_n, tile_z;
  {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int nwg = gx * gy * gz, bid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int cpx = nwg >> 3;
    const int t = bid < cpx * 8 ? (bid & 7) * cpx + (bid >> 3) : bid;
    if (p_.m_inner) {  // weights dominate the traffic: keep one weight tile in L2 while the M tiles stream past it
      tile_m = t % gx;
      tile_z = (t / gx) % gz;
      tile_n = t / (gx * gz);
    } else {
      tile_n = t % gy;
      tile_z = (t / gy) % gz;
      tile_m = t / (gy * gz);
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // problem of a grouped launch (workgroup-uniform): its pointers replace the ones in the argument block
  IgemmArgs p = p_;
  const int nsplit = FAST ? 1 : p.nsplit;
  const int zper = p.nphase * nsplit;
  const int gidx = tile_z / zper;
  tile_z -= gidx * zper;
  if (p.ngroup > 1) {
    p.x = grp.x[gidx]; p.w = grp.w[gidx]; p.y = grp.y[gidx];
    p.bias = grp.bias[gidx]; p.pre = grp.pre[gidx]; p.mask = grp.mask[gidx]; p.res = grp.res[gidx]; p.cs = grp.cs[gidx];
  }
  const int phase = nsplit == 1 ? tile_z : tile_z / nsplit, split = nsplit == 1 ? 0 : tile_z % nsplit;
  const int tb = tp.tap_begin[phase], te = tp.tap_begin[phase + 1];
  const int ntap = te - tb;
  const int KT = SMALLC ? p.kchunks : ntap * p.kchunks;
  // (KT <= 25 taps x 133 chunks, split < 64: 32-bit products; a 64-bit division is ~150 instructions here)
  const int it0 = nsplit == 1 ? 0 : KT * split / nsplit, it1 = nsplit == 1 ? KT : KT * (split + 1) / nsplit;
  const int poh = tp.poh[phase], pow_ = tp.pow[phase];
  const int H = p.H, W = p.W, ldx = p.ldx, Cin = p.Cin, kchunks = p.kchunks;

  int mintap = 0;  // most negative pixel offset any tap of this phase reaches (wave-uniform)
  for (int t = 0; t < ntap; ++t) {  // wave-uniform index: scalar loads from the kernarg segment
    const int v = tp.packed[tb + t];
    if (tid == 0) sTap[t] = v;
    mintap = min(mintap, (int)(signed char)(v & 0xff) * W + (int)(signed char)((v >> 8) & 0xff));
  }
  if (tid < 4) sTap[ntap + tid] = 0;  // the cursor may run one K-tile past the end (range-checked, never consumed)

  // The input descriptor is re-based at the first pixel this workgroup can touch, so the 32-bit byte offsets below only
  // have to span one tile (+ halo) and the tensor itself may be of any size.
  long long base_pix;
  {
    int nn, a0, b0;
    split_row(p, m0, nn, a0, b0);
    base_pix = (long long)(nn * H + a0 * p.si) * W + b0 * p.si + mintap;
    base_pix = base_pix < 0 ? 0 : base_pix;
  }
  const unsigned long long base_bytes = (unsigned long long)base_pix * ldx * 4ull;
  const unsigned long long left = p.x_bytes - base_bytes;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x) + base_pix * ldx, 0, (unsigned)(left < 0x7fffffffull ? left : 0x7fffffffull), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  // ---- staging assignment: thread fills slot (tid & 7) of rows (tid >> 3) + j * RPP
  const int srow = tid >> 3;
  const int csrc = (tid & 7) ^ ((srow >> 1) & 7);  // source chunk landing in this thread's slot
  unsigned a_off[AV];
  int a_ih[AV], a_iw[AV];
#pragma unroll
  for (int j = 0; j < AV; ++j) {
    const int m = m0 + srow + j * RPP;
    int n, a, b;
    split_row(p, m, n, a, b);
    const bool ok = (m < p.M) && (a * p.so + poh < p.OH) && (b * p.so + pow_ < p.OW);
    a_ih[j] = ok ? a * p.si : -(1 << 24);  // a dead row fails every bounds test below
    a_iw[j] = ok ? b * p.si : 0;
    a_off[j] = ok ? (unsigned)(((long long)(n * H + a * p.si) * W + b * p.si - base_pix) * ldx + (SMALLC ? 0 : csrc * 4)) * 4u : 0u;
  }
  unsigned b_off[BV];
#pragma unroll
  for (int j = 0; j < BV; ++j) {
    const int oc = n0 + srow + j * RPP;
    b_off[j] = oc < p.wrows ? (unsigned)(oc * p.wcols + csrc * 4) * 4u : kOobOffset;
  }
  // per-column epilogue vectors of this N tile (neutral values where a flag is off or past Cout), read by the fast epilogue
  constexpr int kG = NB < 4 ? NB : 4;
  constexpr int kStagingFloats = igemm_staging_floats(BM, BN, PREC);
  static_assert(ST == 2, "igemm_staging_floats assumes two stages of the fp32 / bf16x3 loops");
  constexpr int kEpiFloats = WM * WN * 32 * 32 * kG + WM * 2 * BN;
  constexpr int kSvOff = ((kStagingFloats > kEpiFloats ? kStagingFloats : kEpiFloats) + 3) & ~3;
  float* sV = smem + kSvOff;  // [4][BN]: bias, vec2, scale, shift
  __syncthreads();  // sTap visible

  // K-iteration cursor of the NEXT tile to fetch: tap index (relative to tb) and channel chunk
  // K order (kmajor = p.k_cmajor): tap-major walks all channel chunks of one tap before the next tap; channel-major walks all
  // taps of one 32-channel chunk first -- the ntap shifted reads of a chunk then hit the same few KB of pixel rows (L1 / L2)
  // instead of re-streaming the whole activation tile once per tap.  Same products, different fp32 summation order.
  // The cursor is kept as (inner, outer) counters -- which of them is the tap depends on the order -- so that it stays in registers.
  const bool cmajor = !SMALLC && p.k_cmajor != 0;
  const int inner_n = SMALLC ? (1 << 30) : (cmajor ? ntap : kchunks);
  int ci = SMALLC ? it0 : it0 % inner_n, co = SMALLC ? 0 : it0 / inner_n;
  auto fetch = [&](int buf) __attribute__((always_inline)) {
    float* a = sA + buf * BM * 32 + wave * 8 * 32;
    float* b = sB + buf * BN * 32 + wave * 8 * 32;
    int dh, dw;
    unsigned toff, woff;
    bool cok;
    const int lt = cmajor ? ci : co, lc = (SMALLC || !cmajor) ? ci : co;
    if constexpr (SMALLC) {  // this thread's piece is tap (8 lc + csrc), channels 0..3
      const int ti = lc * 8 + csrc;
      cok = ti < ntap;
      const int t = sTap[cok ? ti : 0];
      dh = (int)(signed char)(t & 0xff); dw = (int)(signed char)((t >> 8) & 0xff);
      toff = (unsigned)((dh * W + dw) * ldx) * 4u;
      woff = (unsigned)(lc * 32) * 4u;
    } else {
      const int t = __builtin_amdgcn_readfirstlane(sTap[lt]);
      dh = (int)(signed char)(t & 0xff); dw = (int)(signed char)((t >> 8) & 0xff);
      const int wi = t >> 16;
      cok = lc * 32 + csrc * 4 < Cin;
      toff = (unsigned)((dh * W + dw) * ldx + lc * 32) * 4u;
      woff = (unsigned)(wi * p.wrows * p.wcols + lc * 32) * 4u;
    }
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      const int ih = a_ih[j] + dh, iw = a_iw[j] + dw;
      const bool ok = cok & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(a + j * RPP * 32), 16, (int)(ok ? a_off[j] + toff : kOobOffset),
                                               0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BV; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(b + j * RPP * 32), 16, (int)(b_off[j] + woff), 0, 0, 0);
    if (++ci == inner_n) { ci = 0; ++co; }
    __builtin_amdgcn_sched_barrier(0);  // keep the DMA issue ahead of the MFMA stream that hides its latency
  };

  // BF3: every thread splits the pieces IT staged, in place, once its DMA has landed (its own data: no barrier needed in
  // between; the step's barrier then publishes the split image).  Each element is converted once per workgroup instead of once
  // per wave that reads it, and the MFMA loop carries no VALU work.  Slot layout: {pk(hi0, hi1), pk(hi2, hi3), pk(lo0, lo1), pk(lo2, lo3)}.
  auto presplit = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of the stage have landed
    asm volatile("" ::: "memory");
    float* a = sA + buf * BM * 32 + wave * 8 * 32 + lane * 4;
    float* b = sB + buf * BN * 32 + wave * 8 * 32 + lane * 4;
#pragma unroll
    for (int j = 0; j < AV + BV; ++j) {
      float* q = j < AV ? a + j * RPP * 32 : b + (j - AV) * RPP * 32;
      const f32x4 x = *reinterpret_cast<const f32x4*>(q);
      u32x4_t o;
      unsigned h0, h1, l0, l1;
      split_bf16x2(x[0], x[1], h0, l0);
      split_bf16x2(x[2], x[3], h1, l1);
      o[0] = h0; o[1] = h1; o[2] = l0; o[3] = l1;
      *reinterpret_cast<u32x4_t*>(q) = o;
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- bf16x6 (PREC = 6): staging assignment, requests, split and MFMA step (see the header comment).
  // Stage s of A: planes [hi | mid | lo] of BM rows x 64 bytes at s6A + s * BM * 48 floats; chunk pair q of row r sits at 16-byte slot
  // q ^ ((r >> 2) & 3) of its row in every plane (the rows a ds_read_b128 lane group reads then cover all 64 banks once).
  constexpr int RPP6 = NT / 4;                                   // tile rows per request pass: four threads (= four chunk pairs) per row
  constexpr int PA6 = BF6 ? BM / RPP6 : 1, PB6 = BF6 ? BN / RPP6 : 1;
  float* s6A = smem;                                             // [2][3 planes][BM * 16]
  float* s6B = smem + 2 * BM * 48;                               // [2][3 planes][BN * 16]
  const int srow6 = tid >> 2, q6 = tid & 3;                      // this thread's row (of every pass) and chunk pair
  unsigned a6_off[PA6], b6_off[PB6];
  int a6_ih[PA6], a6_iw[PA6];
  u32x4 ra6[2][PA6][2], rb6[2][PB6][2];                          // two register sets: the tiles in flight
  // The descriptors again, from pointers and extents forced into scalar registers: left to the compiler they sat in vector registers in this
  // instantiation, and every buffer load was wrapped in a waterfall loop (four v_readfirstlane, two 64-bit compares, exec save / restore and a
  // branch per request -- and a basic-block boundary between the MFMAs).
  auto uni64 = [](unsigned long long v) __attribute__((always_inline)) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  const __amdgpu_buffer_rsrc_t rx6 = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<float*>(uni64(reinterpret_cast<unsigned long long>(p.x + base_pix * ldx))), 0,
      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(left < 0x7fffffffull ? left : 0x7fffffffull)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw6 = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<float*>(uni64(reinterpret_cast<unsigned long long>(p.w))), 0, (unsigned)__builtin_amdgcn_readfirstlane((int)p.w_bytes), 0x00020000);
  if constexpr (BF6) {
    static_assert(RPP6 % 16 == 0, "the slot swizzle must not depend on the pass");
#pragma unroll
    for (int j = 0; j < PA6; ++j) {
      const int m = m0 + srow6 + j * RPP6;
      int n, a, b;
      split_row(p, m, n, a, b);
      const bool ok = (m < p.M) && (a * p.so + poh < p.OH) && (b * p.so + pow_ < p.OW);
      a6_ih[j] = ok ? a * p.si : -(1 << 24);
      a6_iw[j] = ok ? b * p.si : 0;
      a6_off[j] = ok ? (unsigned)(((long long)(n * H + a * p.si) * W + b * p.si - base_pix) * ldx + q6 * 8) * 4u : 0u;
    }
#pragma unroll
    for (int j = 0; j < PB6; ++j) {
      const int oc = n0 + srow6 + j * RPP6;
      b6_off[j] = oc < p.wrows ? (unsigned)(oc * p.wcols + q6 * 8) * 4u : kOobOffset;
    }
  }
  auto load6 = [&](auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    const int lt = cmajor ? ci : co, lc = cmajor ? co : ci;
    const int t = __builtin_amdgcn_readfirstlane(sTap[lt]);
    const int dh = (int)(signed char)(t & 0xff), dw = (int)(signed char)((t >> 8) & 0xff);
    const int wi = t >> 16;
    const bool cok0 = lc * 32 + q6 * 8 < Cin, cok1 = lc * 32 + q6 * 8 + 4 < Cin;
    const unsigned toff = (unsigned)((dh * W + dw) * ldx + lc * 32) * 4u;
    const unsigned woff = (unsigned)(wi * p.wrows * p.wcols + lc * 32) * 4u;
#pragma unroll
    for (int j = 0; j < PA6; ++j) {
      const int ih = a6_ih[j] + dh, iw = a6_iw[j] + dw;
      const bool ok = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
      ra6[set][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rx6, (ok & cok0) ? a6_off[j] + toff : kOobOffset, 0, 0);
      ra6[set][j][1] = __builtin_amdgcn_raw_buffer_load_b128(rx6, (ok & cok1) ? a6_off[j] + toff + 16u : kOobOffset, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PB6; ++j) {
      rb6[set][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rw6, b6_off[j] == kOobOffset ? kOobOffset : b6_off[j] + woff, 0, 0);
      rb6[set][j][1] = __builtin_amdgcn_raw_buffer_load_b128(rw6, b6_off[j] == kOobOffset ? kOobOffset : b6_off[j] + woff + 16u, 0, 0);
    }
    if (++ci == inner_n) { ci = 0; ++co; }
    __builtin_amdgcn_sched_barrier(0);   // requests first: the MFMA stream behind them hides their latency
  };
  // register set `set` -> the three bf16 planes of LDS stage `buf`
  auto split6 = [&](auto setc, auto bufc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value, buf = decltype(bufc)::value;
    const int slot = (srow6 * 4 + (q6 ^ ((srow6 >> 2) & 3))) * 4;   // floats: (row, swizzled slot) inside a plane
    float* a = s6A + buf * BM * 48 + slot;
    float* b = s6B + buf * BN * 48 + slot;
#pragma unroll
    for (int j = 0; j < PA6 + PB6; ++j) {
      float* q = j < PA6 ? a + j * RPP6 * 16 : b + (j - PA6) * RPP6 * 16;
      constexpr int kPlaneA = BM * 16, kPlaneB = BN * 16;
      const int plane = j < PA6 ? kPlaneA : kPlaneB;
      const f32x4 x0 = __builtin_bit_cast(f32x4, j < PA6 ? ra6[set][j < PA6 ? j : 0][0] : rb6[set][j < PA6 ? 0 : j - PA6][0]);
      const f32x4 x1 = __builtin_bit_cast(f32x4, j < PA6 ? ra6[set][j < PA6 ? j : 0][1] : rb6[set][j < PA6 ? 0 : j - PA6][1]);
      const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      bf16x8 h, m, l;
      split3_bf16x8(x, h, m, l);
      *reinterpret_cast<bf16x8*>(q) = h;
      *reinterpret_cast<bf16x8*>(q + plane) = m;
      *reinterpret_cast<bf16x8*>(q + 2 * plane) = l;
    }
  };
  int fo6[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) fo6[j] = (lane & 31) * 16 + (((2 * j + (lane >> 5)) ^ (((lane & 31) >> 2) & 3)) << 2);
  // the MFMAs of stage `buf`: lane half h takes chunk pair 2 j + h of both operands for the j-th 16 k (the same k for A and B)
  auto compute6 = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    const float* fa6 = s6A + buf * BM * 48 + (wm * MB * 32) * 16;
    const float* fb6 = s6B + buf * BN * 48 + (wn * NB * 32) * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 ah[MB], am[MB], al[MB], bh[NB], bm[NB], bl[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(fa6 + fo6[j] + i * 512);
        am[i] = *reinterpret_cast<const bf16x8*>(fa6 + fo6[j] + i * 512 + BM * 16);
        al[i] = *reinterpret_cast<const bf16x8*>(fa6 + fo6[j] + i * 512 + 2 * BM * 16);
      }
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        bh[jj] = *reinterpret_cast<const bf16x8*>(fb6 + fo6[j] + jj * 512);
        bm[jj] = *reinterpret_cast<const bf16x8*>(fb6 + fo6[j] + jj * 512 + BN * 16);
        bl[jj] = *reinterpret_cast<const bf16x8*>(fb6 + fo6[j] + jj * 512 + 2 * BN * 16);
      }
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) acc[i][jj] = mfma_bf16x6(ah[i], am[i], al[i], bh[jj], bm[jj], bl[jj], acc[i][jj]);
    }
  };

  const int frow = lane & 31, fh = lane >> 5;
  const float* fa = sA + (wm * MB * 32) * 32;
  const float* fb = sB + (wn * NB * 32) * 32;
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = lds_off(frow, kk * 2 + fh);
  // one K-tile: kk = [K0, K1) quarter steps of 8 K values each, out of LDS buffer `buf`
  auto compute = [&](auto bufc, auto k0c, auto k1c) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value, K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
    if constexpr (BF3) {
      // one bf16 MFMA covers 16 k = the quarter steps kk, kk + 1.  The stage holds the PRE-SPLIT image (presplit below): every
      // 16-byte slot is {bf16 hi of its 4 floats, bf16 lo of its 4 floats}, so a lane's 8 + 8 operand values are the first /
      // second halves of the two slots it reads -- no conversion in the MFMA loop.  Lane half h contributes the same k for A
      // and B, so the sum over k is complete whatever their order inside the instruction.
      static_assert(K0 % 2 == 0 && K1 % 2 == 0, "bf16x3 steps are pairs of quarter steps");
#pragma unroll
      for (int kk = K0; kk < K1; kk += 2) {
        bf16x8 ah[MB], al[MB], bh[NB], bl[NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const u32x4_t r0 = *reinterpret_cast<const u32x4_t*>(fa + fo[kk] + (buf * BM * 32 + i * 1024));
          const u32x4_t r1 = *reinterpret_cast<const u32x4_t*>(fa + fo[kk + 1] + (buf * BM * 32 + i * 1024));
          const u32x4_t h = {r0[0], r0[1], r1[0], r1[1]}, l = {r0[2], r0[3], r1[2], r1[3]};
          ah[i] = __builtin_bit_cast(bf16x8, h);
          al[i] = __builtin_bit_cast(bf16x8, l);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const u32x4_t r0 = *reinterpret_cast<const u32x4_t*>(fb + fo[kk] + (buf * BN * 32 + j * 1024));
          const u32x4_t r1 = *reinterpret_cast<const u32x4_t*>(fb + fo[kk + 1] + (buf * BN * 32 + j * 1024));
          const u32x4_t h = {r0[0], r0[1], r1[0], r1[1]}, l = {r0[2], r0[3], r1[2], r1[3]};
          bh[j] = __builtin_bit_cast(bf16x8, h);
          bl[j] = __builtin_bit_cast(bf16x8, l);
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = mfma_bf16x3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
      }
      return;
    }
#pragma unroll
    for (int kk = K0; kk < K1; ++kk) {
      f32x4 af[MB], bf[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) af[i] = *reinterpret_cast<const f32x4*>(fa + fo[kk] + (buf * BM * 32 + i * 1024));
#pragma unroll
      for (int j = 0; j < NB; ++j) bf[j] = *reinterpret_cast<const f32x4*>(fb + fo[kk] + (buf * BN * 32 + j * 1024));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  using I4 = integral_constant<int, 4>;
  // a step = the MFMAs of one tile with the DMA of the next one issued behind the first quarter, so that its address
  // arithmetic and issue slots run in the shadow of MFMAs already queued
  auto step = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    fetch(buf ^ 1);
    compute(bufc, I0{}, I4{});
    if constexpr (BF3) presplit(integral_constant<int, buf ^ 1>{});
    __syncthreads();
  };
  if constexpr (BF6) {
    // prologue: tiles 0 and 1 requested (a request past the K range is range-checked garbage that is never multiplied)
    if (it0 < it1) { load6(I0{}); load6(I1{}); }
  } else {
    if (it0 < it1) fetch(0);
  }
  // (behind the first DMA so that their latencies overlap; the barrier below publishes both)
  if (FAST || p.fast_epi) {
    const int f0 = p.flags;
    for (int c = tid; c < BN; c += NT) {
      const bool live = n0 + c < p.Cout;
      sV[0 * BN + c] = (live && (f0 & CRDR_EPI_BIAS)) ? p.bias[n0 + c] : 0.f;
      sV[1 * BN + c] = (live && (f0 & (CRDR_EPI_VEC2 | CRDR_EPI_MASKOFF))) ? p.vec2[n0 + c] : 0.f;
      sV[2 * BN + c] = (live && (f0 & CRDR_EPI_AFFINE)) ? p.scale[n0 + c] : 1.f;
      sV[3 * BN + c] = (live && (f0 & CRDR_EPI_AFFINE)) ? p.shift[n0 + c] : 0.f;
    }
  }
  int it = it0;
  if constexpr (BF6) {
    if (it0 < it1) split6(I0{}, I0{});
    __syncthreads();
    // a step = [requests of tile t + 2 into the register set tile t left] [MFMAs of tile t out of stage t & 1, with the split of tile t + 1 -- in
    // the other register set since the previous step -- into the other stage between them: one basic block, the compiler interleaves]
    // [barrier: publishes that stage, frees this one]
    auto step6 = [&](auto bufc) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value;
      load6(bufc);
      compute6(bufc);
      split6(integral_constant<int, buf ^ 1>{}, integral_constant<int, buf ^ 1>{});
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): my LDS writes are done (the requests stay in flight: no __syncthreads here)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    for (; it + 2 <= it1; it += 2) {
      step6(I0{});
      step6(I1{});
    }
    if (it < it1) step6(I0{});
    __syncthreads();
  } else {
  if constexpr (BF3) presplit(I0{});
  __syncthreads();
  for (; it + 2 <= it1; it += 2) {
    step(I0{});
    step(I1{});
  }
  if (it < it1) {
    compute(I0{}, I0{}, I4{});
    __syncthreads();
  }
  }

  // ---- epilogue.  The accumulators go through LDS (the staging buffers are free now) so that every lane handles
  // 4 consecutive channels of one pixel: 16-byte residual / gate loads and 16-byte stores, 512 B contiguous per
  // 32 lanes.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  constexpr int G = NB < 4 ? NB : 4;          // 32-column blocks staged per pass
  constexpr int CLD = 32 * G;                 // floats per staged row
  float* sC = smem + wave * (32 * CLD);       // per-wave [32][CLD]
  const int f = p.flags;
  const bool direct = (p.nphase == 1) && (p.so == 1);  // output pixel index == GEMM row
  const bool vec = p.vec_epi != 0;

  // ---- split-K inside the launch.  Slabs: ws[group][phase][split][M][ws_ld], this tile's part addressed through ONE buffer
  // descriptor based at split 0 (build_plan keeps nsplit * M * ws_ld * 4 below 2 GiB).  Hand-off (MI355X_MICROARCH.md,
  // "Workgroup dispatch, XCD placement & inter-workgroup visibility"): every slab byte is stored sc1 (write-through: no L2
  // write-back needed), every storing wave drains vmcnt, the workgroup barrier, then ONE lane takes the ticket with an
  // agent-scope atomic; the workgroup whose ticket is nsplit - 1 acquires (buffer_inv sc1) and reads all slabs with sc1 loads
  // in split order.  Placement independent, no spinning: a workgroup either leaves or reduces.
  const bool splitk = !FAST && p.nsplit > 1;
  const unsigned slab_bytes = splitk ? (unsigned)((size_t)p.M * p.ws_ld * 4u) : 0u;
  __amdgpu_buffer_rsrc_t rws = rw;
  if (splitk)
    rws = __builtin_amdgcn_make_buffer_rsrc(
        p.ws + (size_t)(gidx * p.nphase + phase) * p.nsplit * ((size_t)p.M * p.ws_ld) + (size_t)m0 * p.ws_ld + n0, 0, 0x7fffffff, 0x00020000);
  // byte offset (relative to rws) of the 16-byte group that lane position q of pass (I, JG, GC) owns in split 0's slab
  auto slab_off = [&](int I, int JG, int GC, int q) __attribute__((always_inline)) {
    const int row = q / (8 * GC), c4 = q - row * (8 * GC);
    const int rr = (wm * MB + I) * 32 + row;
    return (m0 + rr < p.M) ? (unsigned)(rr * p.ws_ld + (wn * NB + JG) * 32 + c4 * 4) * 4u : kOobOffset;
  };
  if (splitk) {
    auto wpass = [&](auto I_, auto JG_, auto GC_) __attribute__((always_inline)) {
      constexpr int I = decltype(I_)::value, JG = decltype(JG_)::value, GC = decltype(GC_)::value;
#pragma unroll
      for (int jj = 0; jj < GC; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sC[((r & 3) + 8 * (r >> 2) + 4 * fh) * CLD + jj * 32 + frow] = acc[I][JG + jj][r];
      // sC is private to the wave: program order is enough
#pragma unroll
      for (int k = 0; k < 4 * GC; ++k) {
        const int q = lane + 64 * k;
        const int row = q / (8 * GC), c4 = q - row * (8 * GC);
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(sC + row * CLD + c4 * 4);
        const unsigned o = slab_off(I, JG, GC, q);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a4), rws, o == kOobOffset ? o : o + (unsigned)split * slab_bytes, 0,
                                               16 /* sc1 */);
      }
    };
    wpass(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, G>{});
    if constexpr (NB > 4) wpass(integral_constant<int, 0>{}, integral_constant<int, 4>{}, integral_constant<int, NB - 4>{});
    if constexpr (MB > 1) {
      wpass(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, G>{});
      if constexpr (NB > 4) wpass(integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, NB - 4>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* sFlag = reinterpret_cast<int*>(smem + kSvOff + 4 * BN);
    if (tid == 0) {
      int* cnt = p.counters + ((gidx * p.nphase + phase) * (int)gridDim.x + tile_m) * (int)gridDim.y + tile_n;
      const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == p.nsplit - 1;
      if (last) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next launch finds zeros again
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *sFlag = last;
    }
    __syncthreads();
    if (!*sFlag) return;
  }
  // what the reducer of a split tile puts into sC instead of its accumulators: the sum of the slabs in split order
  // (((s0 + s1) + s2) + ...), four splits of loads in flight at a time
  auto rfill = [&](auto I_, auto JG_, auto GC_) __attribute__((always_inline)) {
    constexpr int I = decltype(I_)::value, JG = decltype(JG_)::value, GC = decltype(GC_)::value;
#pragma unroll
    for (int k = 0; k < 4 * GC; ++k) {
      const int q = lane + 64 * k;
      const int row = q / (8 * GC), c4 = q - row * (8 * GC);
      const unsigned o = slab_off(I, JG, GC, q);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < p.nsplit; s0 += 4) {
        f32x4 l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          l[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                               rws, (o == kOobOffset || s0 + j >= p.nsplit) ? kOobOffset : o + (unsigned)(s0 + j) * slab_bytes, 0, 16));
#pragma unroll
        for (int j = 0; j < 4; ++j) v += l[j];   // splits past nsplit read zeros: + 0.f is exact
      }
      *reinterpret_cast<f32x4*>(sC + row * CLD + c4 * 4) = v;
    }
  };
  // CRDR_EPI_COLSUM: column sums of this tile's outputs (value before / after the ReLU mask), reduced lane -> wave ->
  // workgroup in a fixed order and written as one partial row per (phase, M tile); crdr_colsum_finish adds the rows up.
  // A lane must keep one 4-channel column group for a whole pass, which holds for passes of 1, 2 or 4 column blocks; a pass
  // of 3 runs as three single-block passes when column sums are requested.
  const bool do_cs = (f & CRDR_EPI_COLSUM) != 0;   // (of a split tile: by its reducer, from the summed values)
  float* sS = smem + WM * WN * 32 * CLD;  // [WM][2][BN] behind the staged accumulators
  // ---- fast epilogue (p.fast_epi): every global access is a buffer instruction issued by all lanes -- dead rows / column
  // groups get an out-of-range offset, loads return 0 and stores are dropped -- relative to this tile's first output pixel;
  // bias / vec2 / scale / shift come from sV.  The code is straight-line: no wait on a store anywhere, one wait per batch of
  // four row groups on the res / mask operands.  Arithmetic and order are those of the general path below.
  if (FAST || p.fast_epi) {
    const bool has_res = (f & CRDR_EPI_RES) != 0, has_mask = (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) != 0;
    long long opix0 = m0;
    if (!direct) {
      int n, ga, gb;
      split_row(p, m0, n, ga, gb);
      opix0 = ((long long)n * p.OH + ga * p.so + poh) * p.OW + gb * p.so + pow_;
    }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + opix0 * p.ldy, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_res ? p.res + opix0 * p.ldres : p.y), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_mask ? p.mask + opix0 * p.ldmask : p.y), 0, 0x7fffffff, 0x00020000);
    auto fpass = [&](auto I_, auto JG_, auto GC_) __attribute__((always_inline)) {
      constexpr int I = decltype(I_)::value, JG = decltype(JG_)::value, GC = decltype(GC_)::value;
      if (splitk) {
        rfill(I_, JG_, GC_);
      } else {
#pragma unroll
        for (int jj = 0; jj < GC; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            sC[((r & 3) + 8 * (r >> 2) + 4 * fh) * CLD + jj * 32 + frow] = acc[I][JG + jj][r];
      }
      // sC is private to the wave: program order is enough
      f32x4 cpre = {0.f, 0.f, 0.f, 0.f}, cpost = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < GC; ++kb) {
        f32x4 res4[4], msk4[4];
        unsigned yoff[4];
        int cc[4];
        bool okk[4];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          const int q = lane + 64 * (kb * 4 + kq);
          const int row = q / (8 * GC), c4 = q - row * (8 * GC);
          const int m = m0 + (wm * MB + I) * 32 + row;
          const int oc0 = n0 + (wn * NB + JG) * 32 + c4 * 4;
          bool live_row = m < p.M;
          unsigned rel = (unsigned)(m - m0);
          if (!direct) {
            const int mm = live_row ? m : m0;
            int n, ga, gb;
            split_row(p, mm, n, ga, gb);
            const int oh = ga * p.so + poh, ow = gb * p.so + pow_;
            live_row = live_row && (oh < p.OH) && (ow < p.OW);
            rel = (unsigned)((((long long)n * p.OH + oh) * p.OW + ow) - opix0);
          }
          const bool ok = live_row && oc0 < p.Cout;
          okk[kq] = ok;
          cc[kq] = (wn * NB + JG) * 32 + c4 * 4;
          yoff[kq] = ok ? (rel * p.ldy + oc0) * 4u : kOobOffset;
          if (has_res)
            res4[kq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ok ? (rel * p.ldres + oc0) * 4u : kOobOffset, 0, 0));
          if (has_mask)
            msk4[kq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, ok ? (rel * p.ldmask + oc0) * 4u : kOobOffset, 0, 0));
        }
        // One pass over the batch's 16 outputs per epilogue flag: a flag is tested once per batch, not once per element (per-element
        // tests compile to a select each, and every non-MFMA instruction costs matrix time -- DESIGN 4e).  Same operations in the
        // same order per element.
        f32x4 o[4];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          const int q = lane + 64 * (kb * 4 + kq);
          const int row = q / (8 * GC), c4 = q - row * (8 * GC);
          o[kq] = *reinterpret_cast<const f32x4*>(sC + row * CLD + c4 * 4);
        }
        if (f & CRDR_EPI_BIAS) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) o[kq] += *reinterpret_cast<const f32x4*>(sV + 0 * BN + cc[kq]);
        }
        if (f & CRDR_EPI_RELU) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[kq][e] = fmaxf(o[kq][e], 0.0f);
        }
        if (f & CRDR_EPI_LRELU) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[kq][e] = o[kq][e] > 0.0f ? o[kq][e] : 0.2f * o[kq][e];
        }
        if (f & CRDR_EPI_VEC2) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) o[kq] += *reinterpret_cast<const f32x4*>(sV + 1 * BN + cc[kq]);
        }
        if (has_res) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) o[kq] += res4[kq];
        }
        if (f & CRDR_EPI_AFFINE) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) {
            const f32x4 scale4 = *reinterpret_cast<const f32x4*>(sV + 2 * BN + cc[kq]);
            const f32x4 shift4 = *reinterpret_cast<const f32x4*>(sV + 3 * BN + cc[kq]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[kq][e] = o[kq][e] * scale4[e] + shift4[e];
          }
        }
        if (do_cs) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int e = 0; e < 4; ++e) cpre[e] += okk[kq] ? o[kq][e] : 0.f;
        }
        if (has_mask) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) {
            f32x4 mv = msk4[kq];
            if (f & CRDR_EPI_MASKOFF) mv -= *reinterpret_cast<const f32x4*>(sV + 1 * BN + cc[kq]);
            if (f & CRDR_EPI_LRELUMASK) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[kq][e] = mv[e] > 0.0f ? o[kq][e] : 0.2f * o[kq][e];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[kq][e] = mv[e] > 0.0f ? o[kq][e] : 0.0f;
            }
          }
        }
        if (do_cs) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int e = 0; e < 4; ++e) cpost[e] += okk[kq] ? o[kq][e] : 0.f;
        }
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[kq]), ry, yoff[kq], 0, 0);
      }
      if (do_cs) {
        if constexpr (GC == 1 || GC == 2 || GC == 4) {
#pragma unroll
          for (int off = 32; off >= 8 * GC; off >>= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              cpre[e] += __shfl_xor(cpre[e], off, 64);
              cpost[e] += __shfl_xor(cpost[e], off, 64);
            }
          if (lane < 8 * GC) {
            float* d0 = sS + (wm * 2 + 0) * BN + (wn * NB + JG) * 32 + lane * 4;
            float* d1 = sS + (wm * 2 + 1) * BN + (wn * NB + JG) * 32 + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (I == 0) { d0[e] = cpre[e]; d1[e] = cpost[e]; }
              else { d0[e] += cpre[e]; d1[e] += cpost[e]; }
            }
          }
        }
      }
    };
    // a pass of three column blocks cannot keep one column group per lane: with column sums it runs as three single-block passes
    auto fgroup = [&](auto I_, auto JG_, auto GC_) __attribute__((always_inline)) {
      constexpr int JG = decltype(JG_)::value, GC = decltype(GC_)::value;
      if constexpr (GC == 3) {
        if (do_cs) {
          fpass(I_, integral_constant<int, JG>{}, integral_constant<int, 1>{});
          fpass(I_, integral_constant<int, JG + 1>{}, integral_constant<int, 1>{});
          fpass(I_, integral_constant<int, JG + 2>{}, integral_constant<int, 1>{});
          return;
        }
      }
      fpass(I_, JG_, GC_);
    };
    fgroup(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, G>{});
    if constexpr (NB > 4) fgroup(integral_constant<int, 0>{}, integral_constant<int, 4>{}, integral_constant<int, NB - 4>{});
    if constexpr (MB > 1) {
      fgroup(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, G>{});
      if constexpr (NB > 4) fgroup(integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, NB - 4>{});
    }
    if (do_cs) {
      __syncthreads();
      float* dst = p.cs + ((size_t)(phase * gridDim.x + tile_m) * 2) * p.cs_ld;
      for (int t = tid; t < 2 * BN; t += NT) {
        const int which = t / BN, c = t - which * BN;
        float v = sS[(0 * 2 + which) * BN + c];
#pragma unroll
        for (int w2 = 1; w2 < WM; ++w2) v += sS[(w2 * 2 + which) * BN + c];
        if (n0 + c < p.Cout) dst[(size_t)which * p.cs_ld + n0 + c] = v;
      }
    }
    return;
  }
  if constexpr (!FAST) {
  // one pass = GC column blocks [JG, JG+GC) of accumulator row-block I (all compile-time so acc stays in registers)
  auto pass = [&](auto I_, auto JG_, auto GC_) __attribute__((always_inline)) {
    constexpr int I = decltype(I_)::value, JG = decltype(JG_)::value, GC = decltype(GC_)::value;
    if (splitk) {
      rfill(I_, JG_, GC_);
    } else {
#pragma unroll
      for (int jj = 0; jj < GC; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sC[((r & 3) + 8 * (r >> 2) + 4 * fh) * CLD + jj * 32 + frow] = acc[I][JG + jj][r];
    }
    __syncthreads();
    f32x4 cpre = {0.f, 0.f, 0.f, 0.f}, cpost = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4 * GC; ++k) {
      const int q = lane + 64 * k;
      const int row = q / (8 * GC), c4 = q - row * (8 * GC);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(sC + row * CLD + c4 * 4);
      const int m = m0 + (wm * MB + I) * 32 + row;
      const int oc0 = n0 + (wn * NB + JG) * 32 + c4 * 4;
      bool live_row = m < p.M;
      {
        size_t opix = (size_t)m;
        if (!direct) {
          const int mm = live_row ? m : 0;
          int n, ga, gb;
          split_row(p, mm, n, ga, gb);
          const int oh = ga * p.so + poh, ow = gb * p.so + pow_;
          live_row = live_row && (oh < p.OH) && (ow < p.OW);
          opix = ((size_t)n * p.OH + oh) * p.OW + ow;
        }
        if (live_row && oc0 < p.Cout) {
          const bool full = vec && (oc0 + 3 < p.Cout);
          f32x4 res4 = {0.f, 0.f, 0.f, 0.f}, gx4 = res4, gt4 = res4, old4 = res4, pre4 = res4, msk4 = res4;
          if (full) {
            if (f & CRDR_EPI_PREADD) pre4 = *reinterpret_cast<const f32x4*>(p.pre + opix * p.ldpre + oc0);
            if (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) msk4 = *reinterpret_cast<const f32x4*>(p.mask + opix * p.ldmask + oc0);
            if (f & CRDR_EPI_RES) res4 = *reinterpret_cast<const f32x4*>(p.res + opix * p.ldres + oc0);
            if (f & CRDR_EPI_GATE) {
              gx4 = *reinterpret_cast<const f32x4*>(p.gx + opix * p.ldg + oc0);
              gt4 = *reinterpret_cast<const f32x4*>(p.gt + opix * p.ldg + oc0);
            }
            if (f & CRDR_EPI_ACCUM) old4 = *reinterpret_cast<const f32x4*>(p.y + opix * p.ldy + oc0);
          }
          f32x4 o4, s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int oc = oc0 + e;
            const bool live = oc < p.Cout;
            float v = a4[e];
            if (f & CRDR_EPI_PREADD) v += full ? pre4[e] : (live ? p.pre[opix * p.ldpre + oc] : 0.f);
            if (f & CRDR_EPI_BIAS) v += live ? p.bias[oc] : 0.f;
            if (f & CRDR_EPI_RELU) v = fmaxf(v, 0.0f);
            if (f & CRDR_EPI_LRELU) v = v > 0.0f ? v : 0.2f * v;
            if (f & CRDR_EPI_VEC2) v += live ? p.vec2[oc] : 0.f;
            if (f & CRDR_EPI_RES) v += full ? res4[e] : (live ? p.res[opix * p.ldres + oc] : 0.f);
            if (f & CRDR_EPI_GATE) {
              const float sgm = 1.0f / (1.0f + expf(-v));
              s4[e] = sgm;
              const float gxv = full ? gx4[e] : (live ? p.gx[opix * p.ldg + oc] : 0.f);
              const float gtv = full ? gt4[e] : (live ? p.gt[opix * p.ldg + oc] : 0.f);
              v = gxv + gtv * sgm;
            }
            if (f & CRDR_EPI_AFFINE) v = live ? v * p.scale[oc] + p.shift[oc] : v;
            if (do_cs && live) cpre[e] += v;
            if (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) {
              float mv = full ? msk4[e] : (live ? p.mask[opix * p.ldmask + oc] : 0.f);
              if (f & CRDR_EPI_MASKOFF) mv -= live ? p.vec2[oc] : 0.f;
              v = mv > 0.0f ? v : ((f & CRDR_EPI_LRELUMASK) ? 0.2f * v : 0.0f);
            }
            if (do_cs && live) cpost[e] += v;
            if (f & CRDR_EPI_ACCUM) v += full ? old4[e] : (live ? p.y[opix * p.ldy + oc] : 0.f);
            o4[e] = v;
          }
          if (full) {
            *reinterpret_cast<f32x4*>(p.y + opix * p.ldy + oc0) = o4;
            if (f & CRDR_EPI_GATE) *reinterpret_cast<f32x4*>(p.sig + opix * p.ldg + oc0) = s4;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (oc0 + e < p.Cout) {
                p.y[opix * p.ldy + oc0 + e] = o4[e];
                if (f & CRDR_EPI_GATE) p.sig[opix * p.ldg + oc0 + e] = s4[e];
              }
          }
        }
      }
    }
    if (do_cs) {
      if constexpr (GC == 1 || GC == 2 || GC == 4) {
        // lanes that share a column group differ in the lane bits >= 8 GC: fixed-order butterfly
#pragma unroll
        for (int off = 32; off >= 8 * GC; off >>= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cpre[e] += __shfl_xor(cpre[e], off, 64);
            cpost[e] += __shfl_xor(cpost[e], off, 64);
          }
        if (lane < 8 * GC) {
          float* d0 = sS + (wm * 2 + 0) * BN + (wn * NB + JG) * 32 + lane * 4;
          float* d1 = sS + (wm * 2 + 1) * BN + (wn * NB + JG) * 32 + lane * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (I == 0) { d0[e] = cpre[e]; d1[e] = cpost[e]; }
            else { d0[e] += cpre[e]; d1[e] += cpost[e]; }
          }
        }
      }
    }
    __syncthreads();
  };
  auto group = [&](auto I_, auto JG_, auto GC_) __attribute__((always_inline)) {
    constexpr int JG = decltype(JG_)::value, GC = decltype(GC_)::value;
    if constexpr (GC == 3) {
      if (do_cs) {  // (see the fast path)
        pass(I_, integral_constant<int, JG>{}, integral_constant<int, 1>{});
        pass(I_, integral_constant<int, JG + 1>{}, integral_constant<int, 1>{});
        pass(I_, integral_constant<int, JG + 2>{}, integral_constant<int, 1>{});
        return;
      }
    }
    pass(I_, JG_, GC_);
  };
  group(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, G>{});
  if constexpr (NB > 4) group(integral_constant<int, 0>{}, integral_constant<int, 4>{}, integral_constant<int, NB - 4>{});
  if constexpr (MB > 1) {
    group(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, G>{});
    if constexpr (NB > 4) group(integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, NB - 4>{});
  }
  static_assert(MB <= 2 && NB <= 8, "epilogue passes are written out for MB <= 2, NB <= 8");
  if (do_cs) {  // (the last pass ended with a barrier) sum the WM row groups in order, one partial row per (phase, M tile)
    float* dst = p.cs + ((size_t)(phase * gridDim.x + tile_m) * 2) * p.cs_ld;
    for (int t = tid; t < 2 * BN; t += NT) {
      const int which = t / BN, c = t - which * BN;
      float v = sS[(0 * 2 + which) * BN + c];
#pragma unroll
      for (int w2 = 1; w2 < WM; ++w2) v += sS[(w2 * 2 + which) * BN + c];
      if (n0 + c < p.Cout) dst[(size_t)which * p.cs_ld + n0 + c] = v;
    }
  }
  }   // !FAST
}

}  // namespace crdr
