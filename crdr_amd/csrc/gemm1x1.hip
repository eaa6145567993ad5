// Streaming 1x1 convolution kernel of the implicit-GEMM family (see igemm.hip for the tiled kernels and the host side).
//
// The throw-away build switches DESIGN.md 4c's cost breakdown was measured with (ring not refilled, results not stored, no epilogue,
// L2-resident tiles) live in tools/experiments/kernel_experiment_switches.patch, not here.  CRDR_STORE_AUX=2: non-temporal stores.

#include <atomic>

#include "common.hpp"
#include "igemm_args.hpp"

#ifndef CRDR_STORE_AUX
#define CRDR_STORE_AUX 0
#endif

namespace crdr {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// Streaming 1x1 convolution (a plain GEMM  out[m][oc] = epi(sum_c in[m][c] w[oc][c]),  K = Cin <= 320).
//
// The tiled kernel ran the 1x1 layers of the bottleneck blocks at ~40 % MFMA occupancy and ~2 TB/s: a K loop of 3..8
// iterations cannot hide the load latency behind a double buffer, and every tile pays its own ramp and epilogue (PMC:
// profiles/r2_f_pmc_1x1_vs_3x3.txt).  Here a workgroup is PERSISTENT: it loads its weight tile [BN][K] into LDS once, then
// walks its share of the M tiles (32 NW rows each); every wave streams the 32 activation rows it multiplies through its own
// ring of S [32][32]-float stages filled by LDS-DMA, which keeps running across tile boundaries -- the next tile's rows are
// in flight during the epilogue of the current one.  The ring being private to the wave, the loop has no workgroup barrier,
// only counted vmcnt waits (vmcnt counts loads, DMA pieces and stores in issue order, see the loop).  Wave tile = 32 rows x
// BN columns (NB = BN / 32 accumulators); fragment layout and epilogue arithmetic are those of the tiled kernel, so the
// results are bit-identical to its unsplit configurations.  The epilogue is straight-line: per-column vectors staged in LDS
// once, operands and results moved by buffer instructions issued by all lanes (out-of-range offsets for dead rows / column
// groups), the ring slot consumed last doubles as the accumulator-transpose buffer.  Workgroups that share M tiles
// (different N tiles) sit on one XCD, so the activation rows come from HBM once.  A grouped launch spreads its problems over
// the workgroups.  No split-K, no gate / pre-add / accumulate epilogue.  Design notes and measurements: DESIGN.md 4c.
// ------------------------------------------------------------------------------------------------------------
// OPS: operands read by the epilogue besides the per-column vectors: bit 0 = res (CRDR_EPI_RES), bit 1 = mask (the ReLU masks)
// NW: waves per workgroup (4: one per SIMD; 8: two per SIMD, so one wave's epilogue and stores overlap the other's MFMAs)
// PREC = 6: fp32-equivalent split-bf16 products (common.hpp, CRDR_CONV_BF16X6).  The stages stay fp32 -- a wave's activation rows are its own and are
// read once, so they are split where they are used, in registers; the weight fragments likewise (a pre-split weight tile would be 1.5x the LDS
// this kernel already fills).  Six bf16 MFMAs of 32 cycles per 16 k instead of eight fp32 ones of 64: the layer is then bound by its HBM stream.
template <int NB, int S, int OPS, int NW, int PREC = 0>
__global__ __launch_bounds__(64 * NW) void gemm1x1_kernel(const IgemmArgs p_, const StreamArgs sa, const IgemmGroup grp) {
  constexpr int BM = 32 * NW, BN = 32 * NB, NT = 64 * NW, AV = BM * 8 / NT;
  static_assert(AV == 4, "a wave fetches its own 32 rows in four instructions");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KT = p_.kchunks;
  float* sB = smem;                            // [KT][BN * 32]   weight tile, one swizzled image per 32-channel chunk
  float* sA = sB + KT * BN * 32;               // [S][BM * 32]    activation ring
  float* sS = sA + S * BM * 32;                // [NW][2][BN]     column sums of the row groups
  float* sV = sS + NW * 2 * BN;                // [4][BN]         bias, vec2, scale, shift of this column tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;        // consecutive workgroup ids rotate over the XCDs
  const int tile_n = slot % sa.gridN;
  const int nlanes = sa.nlanes;
  const int lslot = slot / sa.gridN;           // (problem, group of 8 lanes)
  const int gidx = lslot / (nlanes >> 3);
  const int mlane = (lslot - gidx * (nlanes >> 3)) * 8 + xcd;
  IgemmArgs p = p_;
  if (p.ngroup > 1) {  // problem of a grouped launch (workgroup-uniform)
    p.x = grp.x[gidx]; p.w = grp.w[gidx]; p.y = grp.y[gidx];
    p.bias = grp.bias[gidx]; p.mask = grp.mask[gidx]; p.res = grp.res[gidx]; p.cs = grp.cs[gidx];
  }
  const int n0 = tile_n * BN;
  const int mtiles = (p.M + BM - 1) / BM;
  const int my_tiles = mlane < mtiles ? (mtiles - mlane + nlanes - 1) / nlanes : 0;
  const int ldx = p.ldx;

  const int srow = tid >> 3;
  const int csrc = (tid & 7) ^ ((srow >> 1) & 7);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
  // ---- weight tile, once
  for (int kc = 0; kc < KT; ++kc) {
    for (int r0 = wave * 8; r0 < BN; r0 += NT / 8) {  // 8 rows per wave instruction; srow = r0 + lane / 8 on the first pass
      const int oc = n0 + r0 + (lane >> 3);
      const unsigned off = oc < p.wrows ? (unsigned)(oc * p.wcols + kc * 32 + csrc * 4) * 4u : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sB + kc * BN * 32 + r0 * 32), 16, (int)off, 0, 0, 0);
    }
  }
  // ---- activation ring: every call loads the next (tile, K chunk) of this workgroup into the next slot; past the end it
  // issues out-of-range loads (they write zeros) so that the number of pieces in flight per thread stays uniform
  int f_t = 0, f_kc = 0, f_slot = 0;
  auto fetch = [&]() __attribute__((always_inline)) {
    const bool live = f_t < my_tiles;
    const long long m0 = (long long)(mlane + f_t * nlanes) * BM;
    const unsigned long long base_bytes = live ? (unsigned long long)m0 * ldx * 4ull : 0ull;
    const unsigned long long left = live ? p.x_bytes - base_bytes : 0ull;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (live ? m0 * ldx : 0), 0, (unsigned)(left < 0x7fffffffull ? left : 0x7fffffffull), 0x00020000);
    // a wave fetches exactly the 32 rows it multiplies (8 rows per instruction): the ring needs no workgroup barrier
    float* a = sA + f_slot * BM * 32 + wave * 32 * 32;
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      const int r = wave * 32 + (lane >> 3) + j * 8;
      const int cs = (lane & 7) ^ ((((lane >> 3) + j * 8) >> 1) & 7);
      const bool ok = live && (m0 + r < p.M);
      const unsigned off = (unsigned)(r * ldx + f_kc * 32 + cs * 4) * 4u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(a + j * 8 * 32), 16, (int)(ok ? off : kOobOffset), 0, 0, 0);
    }
    if (++f_kc == KT) { f_kc = 0; ++f_t; }
    if (++f_slot == S) f_slot = 0;
  };
#pragma unroll
  for (int g = 0; g < S - 1; ++g) fetch();

  f32x16 acc[NB];
  const int frow = lane & 31, fh = lane >> 5;
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = lds_off(frow, kk * 2 + fh);
  const int f = p.flags;
  const bool do_cs = (f & CRDR_EPI_COLSUM) != 0;
  constexpr bool HAS_RES = (OPS & 1) != 0, HAS_MASK = (OPS & 2) != 0;
  // per-column epilogue vectors, staged once (neutral values beyond Cout): the epilogue never waits on them
  for (int c = tid; c < BN; c += NT) {
    const bool live = n0 + c < p.Cout;
    sV[0 * BN + c] = (live && (f & CRDR_EPI_BIAS)) ? p.bias[n0 + c] : 0.f;
    sV[1 * BN + c] = (live && (f & (CRDR_EPI_VEC2 | CRDR_EPI_MASKOFF))) ? p.vec2[n0 + c] : 0.f;
    sV[2 * BN + c] = (live && (f & CRDR_EPI_AFFINE)) ? p.scale[n0 + c] : 1.f;
    sV[3 * BN + c] = (live && (f & CRDR_EPI_AFFINE)) ? p.shift[n0 + c] : 0.f;
  }
  {  // the weight tile and the vectors are shared: everyone's pieces must have landed (the ring stays in flight)
    constexpr int kPro = (S - 1) * AV;
    static_assert(kPro <= 15, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt(0x0070 | kPro);      // vmcnt(kPro), lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  constexpr int kWait = (S - 2) * AV;  // pieces that may still be in flight when the stage to be consumed must have landed
  constexpr int kWaitEpi = kWait + 4 * NB;  // ... plus the stores of an epilogue issued since that stage was requested
  static_assert(kWaitEpi <= 63, "vmcnt immediate");
  constexpr int kImmWait = 0x0F70 | (kWait & 15) | ((kWait >> 4) << 14);
  constexpr int kImmWaitEpi = 0x0F70 | (kWaitEpi & 15) | ((kWaitEpi >> 4) << 14);
  const int c4 = lane & 7, rbase = lane >> 3;  // epilogue: a lane keeps one 4-channel column group, rows rbase + 8 k

  int c_slot = 0;
  for (int t = 0; t < my_tiles; ++t) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int kc = 0; kc < KT; ++kc) {
      // vmcnt counts loads, DMA pieces and stores in issue order: the stage consumed now is older than S - 2 stages and,
      // during the first S - 1 iterations after an epilogue, than that epilogue's 4 NB stores
      if (t > 0 && kc < S - 1) __builtin_amdgcn_s_waitcnt(kImmWaitEpi);
      else __builtin_amdgcn_s_waitcnt(kImmWait);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const float* fa = sA + c_slot * BM * 32 + (wave * 32) * 32;
      const float* fb = sB + kc * BN * 32;
      if constexpr (PREC == 6) {
        // one bf16 MFMA covers the quarter steps kk, kk + 1: a lane's 8 values are the two 16-byte slots it reads (the same k for A and B)
#pragma unroll
        for (int kk = 0; kk < 4; kk += 2) {
          bf16x8 ah, am, al;
          {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(fa + fo[kk]), a1 = *reinterpret_cast<const f32x4*>(fa + fo[kk + 1]);
            const float x[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            split3_bf16x8(x, ah, am, al);
          }
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(fb + fo[kk] + j * 1024), b1 = *reinterpret_cast<const f32x4*>(fb + fo[kk + 1] + j * 1024);
            const float x[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            bf16x8 bh, bm, bl;
            split3_bf16x8(x, bh, bm, bl);
            acc[j] = mfma_bf16x6(ah, am, al, bh, bm, bl, acc[j]);
          }
          if (kk == 0) fetch();
        }
        if (++c_slot == S) c_slot = 0;
        continue;
      }
      // fragments of k group kk + 1 are requested before the MFMAs of group kk are issued
      f32x4 af[2], bf[2][NB];
      af[0] = *reinterpret_cast<const f32x4*>(fa + fo[0]);
#pragma unroll
      for (int j = 0; j < NB; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(fb + fo[0] + j * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
          af[(kk + 1) & 1] = *reinterpret_cast<const f32x4*>(fa + fo[kk + 1]);
#pragma unroll
          for (int j = 0; j < NB; ++j) bf[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(fb + fo[kk + 1] + j * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][s2], bf[kk & 1][j][s2], acc[j], 0, 0, 0);
        if (kk == 0) fetch();                       // issued in the shadow of the first MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
      if (++c_slot == S) c_slot = 0;
    }
    // ---- epilogue of this M tile (the ring keeps filling meanwhile).  sC is private to the wave: no barriers.  Every
    // global access is a buffer instruction issued by all lanes (rows past M / column groups past Cout get an out-of-range
    // offset: loads return 0, stores are dropped), so the code is straight-line, the compiler's vmcnt waits are exact and
    // the operands of pass j + 1 are in flight while pass j is computed and stored.
    const int mt = mlane + t * nlanes;
    // the ring slot consumed last is free until the next fetch: it stages this wave's transposed accumulators
    float* sC = sA + (c_slot == 0 ? S - 1 : c_slot - 1) * BM * 32 + wave * 1024;
    const long long mw = (long long)mt * BM + wave * 32;  // first row of this wave
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + mw * p.ldy, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_RES ? p.res + mw * p.ldres : p.y), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_MASK ? p.mask + mw * p.ldmask : p.y), 0, 0x7fffffff, 0x00020000);
    bool rowok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) rowok[k] = mw + rbase + 8 * k < p.M;
    f32x4 res4[2][4], msk4[2][4];
    auto load_ops = [&](int j) __attribute__((always_inline)) {
      const int oc0 = n0 + j * 32 + c4 * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = rowok[k] && oc0 < p.Cout;
        if constexpr (HAS_RES)
          res4[j & 1][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              rr, ok ? (unsigned)((rbase + 8 * k) * p.ldres + oc0) * 4u : kOobOffset, 0, 0));
        if constexpr (HAS_MASK)
          msk4[j & 1][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              rm, ok ? (unsigned)((rbase + 8 * k) * p.ldmask + oc0) * 4u : kOobOffset, 0, 0));
      }
    };
    load_ops(0);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int oc0 = n0 + j * 32 + c4 * 4;
      const bool colok = oc0 < p.Cout;  // Cout % 4 == 0 (checked by the host): column groups are whole
      if (j + 1 < NB) load_ops(j + 1);
      const f32x4 bias4 = *reinterpret_cast<const f32x4*>(sV + 0 * BN + j * 32 + c4 * 4);
      const f32x4 vec24 = *reinterpret_cast<const f32x4*>(sV + 1 * BN + j * 32 + c4 * 4);
      const f32x4 scale4 = *reinterpret_cast<const f32x4*>(sV + 2 * BN + j * 32 + c4 * 4);
      const f32x4 shift4 = *reinterpret_cast<const f32x4*>(sV + 3 * BN + j * 32 + c4 * 4);
#pragma unroll
      for (int r = 0; r < 16; ++r) sC[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + frow] = acc[j][r];
      f32x4 cpre = {0.f, 0.f, 0.f, 0.f}, cpost = {0.f, 0.f, 0.f, 0.f};
      // One pass over this lane's 16 outputs of the column block per epilogue flag: a flag is tested once per block, not once per
      // element (per-element tests compiled to ~6 selects per output -- a fifth of this kernel's time on the 192 -> 96 layers;
      // every non-MFMA instruction costs matrix time).  Same operations in the same order per element.
      f32x4 o[4];
      bool okk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = *reinterpret_cast<const f32x4*>(sC + (rbase + 8 * k) * 32 + c4 * 4);
        okk[k] = rowok[k] && colok;
      }
      if (f & CRDR_EPI_BIAS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += bias4;
      }
      if (f & CRDR_EPI_RELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[k][e] = fmaxf(o[k][e], 0.0f);
      }
      if (f & CRDR_EPI_LRELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[k][e] = o[k][e] > 0.0f ? o[k][e] : 0.2f * o[k][e];
      }
      if (f & CRDR_EPI_VEC2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += vec24;
      }
      if constexpr (HAS_RES) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += res4[j & 1][k];
      }
      if (f & CRDR_EPI_AFFINE) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[k][e] = o[k][e] * scale4[e] + shift4[e];
      }
      if (do_cs) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) cpre[e] += okk[k] ? o[k][e] : 0.f;
      }
      if constexpr (HAS_MASK) {
        f32x4 moff = {0.f, 0.f, 0.f, 0.f};
        if (f & CRDR_EPI_MASKOFF) moff = vec24;
        if (f & CRDR_EPI_LRELUMASK) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[k][e] = (msk4[j & 1][k][e] - moff[e]) > 0.0f ? o[k][e] : 0.2f * o[k][e];
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[k][e] = (msk4[j & 1][k][e] - moff[e]) > 0.0f ? o[k][e] : 0.0f;
        }
      }
      if (do_cs) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) cpost[e] += okk[k] ? o[k][e] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[k]), ry,
                                               okk[k] ? (unsigned)((rbase + 8 * k) * p.ldy + oc0) * 4u : kOobOffset, 0, CRDR_STORE_AUX);
      if (do_cs) {
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cpre[e] += __shfl_xor(cpre[e], off, 64);
            cpost[e] += __shfl_xor(cpost[e], off, 64);
          }
        if (lane < 8) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sS[(wave * 2 + 0) * BN + j * 32 + lane * 4 + e] = cpre[e];
            sS[(wave * 2 + 1) * BN + j * 32 + lane * 4 + e] = cpost[e];
          }
        }
      }
    }
    if (do_cs) {
      __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): my sS writes are done
      __builtin_amdgcn_s_barrier();
      float* dst = p.cs + ((size_t)mt * 2) * p.cs_ld;
      for (int t2 = tid; t2 < 2 * BN; t2 += NT) {
        const int which = t2 / BN, c = t2 - which * BN;
        float v = sS[(0 * 2 + which) * BN + c];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) v += sS[(w2 * 2 + which) * BN + c];
        if (n0 + c < p.Cout) dst[(size_t)which * p.cs_ld + n0 + c] = v;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();              // sS is free for the next tile
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): no DMA may land in LDS after the workgroup is gone
}

struct StreamCfg {
  int nb, stages, nw;
  void (*kern[4])(const IgemmArgs, const StreamArgs, const IgemmGroup);  // by OPS
  void (*kern6[4])(const IgemmArgs, const StreamArgs, const IgemmGroup);  // ... with bf16x6 products
};
#define SCFG(nb, st, nw) \
  {nb, st, nw, {gemm1x1_kernel<nb, st, 0, nw>, gemm1x1_kernel<nb, st, 1, nw>, gemm1x1_kernel<nb, st, 2, nw>, gemm1x1_kernel<nb, st, 3, nw>}, \
   {gemm1x1_kernel<nb, st, 0, nw, 6>, gemm1x1_kernel<nb, st, 1, nw, 6>, gemm1x1_kernel<nb, st, 2, nw, 6>, gemm1x1_kernel<nb, st, 3, nw, 6>}}
static const StreamCfg kStreamCfgs[] = {
    SCFG(2, 4, 4), SCFG(3, 4, 4), SCFG(4, 4, 4), SCFG(6, 3, 4),   // one wave per SIMD, deep ring
    SCFG(2, 2, 8), SCFG(3, 2, 8), SCFG(4, 2, 8), SCFG(5, 2, 8),   // two waves per SIMD, double buffer
};
#undef SCFG
static const int kNumStreamCfgs = sizeof(kStreamCfgs) / sizeof(kStreamCfgs[0]);

int stream_num_variants() { return kNumStreamCfgs; }

void stream_variant_shape(int v, int* nb, int* stages, int* nw) {
  *nb = kStreamCfgs[v].nb;
  *stages = kStreamCfgs[v].stages;
  *nw = kStreamCfgs[v].nw;
}

void stream_launch(int v, const IgemmArgs& a, const StreamArgs& sa, const IgemmGroup& grp, unsigned grid, size_t lds, hipStream_t s) {
  const StreamCfg& sc = kStreamCfgs[v];
  const int ops = ((a.flags & CRDR_EPI_RES) ? 1 : 0) | ((a.flags & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) ? 2 : 0);
  const int b6 = (a.flags & CRDR_CONV_BF16X6) ? 1 : 0;
  auto kern = b6 ? sc.kern6[ops] : sc.kern[ops];
  static std::atomic<bool> attr_done[2][16][4];
  if (!attr_done[b6][v][ops].load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[b6][v][ops].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * sc.nw), lds, s, a, sa, grp);
}

}  // namespace crdr
