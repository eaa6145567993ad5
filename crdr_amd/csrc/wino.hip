// Winograd F(2x2, 3x3) convolution on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) for 3x3 stride-1 layers and their
// input gradients: Y = A^T [ (G g G^T) . (B^T d B) ] A  (Lavin & Gray 2016, the minimal-filtering form cuDNN's
// CUDNN_CONVOLUTION_*_ALGO_WINOGRAD uses, which the reference reaches through cudnn.benchmark, base_trainer.py:20).
// 16 element-wise products per 2x2 outputs and channel instead of 36: 2.25x fewer MFMAs than the implicit GEMM.  The data
// transform has +-1 coefficients only, the filter transform halves (exact), so the arithmetic stays fp32 throughout; the
// association of the sums differs from the direct form (typical deviation a few 1e-7 of the output scale, tests/test_gpu_conv.py).
//
// One workgroup = 8 waves = 16 x 16 output pixels (8 x 8 Winograd tiles) of one image x 64 output channels.
//   wave (ph, wm, wn): transform rows xi in {2 ph, 2 ph + 1} (8 of the 16 positions), tile columns 4 wm .. 4 wm + 3 (8 rows x 4
//   columns = 32 tiles = the 32 MFMA rows), output channels 32 wn .. 32 wn + 31: 8 accumulator blocks of 32 x 32 = 128 AGPRs,
//   two waves per SIMD.
// K loop: sub-steps of 8 input channels, operands double buffered in LDS and filled by LDS-DMA:
//   * the raw 18 x 18 x 8 input patch, stored by pixel parity class so that the 32 tiles' reads of patch pixel (i, j) are
//     consecutive 16-B slots: [half h = channels 4h..4h+3][class (i&1, j&1)][9 rows][12 slots (9 used)]; with 8 x 4 tiles per
//     wave a row pitch of 12 slots makes every 16-lane group of a ds_read_b128 hit 16 different slots mod 16 (conflict free);
//     pixels outside the image (padding) and channels past Cin read zeros through the buffer range check;
//   * the transformed filters of the sub-step, one contiguous 32 KiB block in memory and in LDS: [position 16][h][oc 64][4].
//   Every wave builds B^T d B for its 8 positions in registers from 12 ds_read_b128 (lane = tile, half-wave = channel half, the
//   operand layout of the implicit-GEMM kernel), reads 8 filter fragments and issues 32 MFMAs per stage.
// Epilogue: the two waves of a (wm, wn) pair each hold half of the xi sum; each forms its part of A^T M A for both output rows,
//   hands the part of the partner's row over through LDS and finishes its own row (ph = output row inside the tile), then runs
//   the element-wise epilogue of the implicit-GEMM kernel (same order of operations) with dword buffer stores: 32 lanes = 32
//   consecutive channels of one pixel.
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
constexpr int kStageFloats = kInFloats + kUFloats;
constexpr int kStagingFloats = 2 * kStageFloats;
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
template <int PH, int ROLE>
__device__ __forceinline__ void wino_loop(const IgemmArgs& p, float* smem, const __amdgpu_buffer_rsrc_t rx, const __amdgpu_buffer_rsrc_t ru,
                                          const unsigned (&a_off)[4], const unsigned a_okmask, unsigned u_off0,
                                          int lane, int wave, int wm, int wn, f32x16 (&acc)[8]) {
  const int K8 = p.kchunks;
  const int m = lane & 31, fh = lane >> 5;
  const int ty = m >> 2, tx = (m & 3) + 4 * wm;
  // float offsets of this lane's 12 raw reads (rows PH .. PH + 2 of the 4 x 4 patch, all 4 columns) inside a stage
  int ro[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = PH + a;
      const int cls = (i & 1) * 2 + (j & 1);
      ro[a][j] = (fh * 432 + cls * 108 + (ty + (i >> 1)) * 12 + tx + (j >> 1)) * 4;
    }
  const int bo = kInFloats + ((2 * PH * 4 * 2 + fh) * 64 + wn * 32 + m) * 4;  // position p = (2 PH + a) * 4 + nu: + (a * 4 + nu) * 512
  const int w1 = wave & 3;

  // DMA instruction idx (0..11) of this (role-1) wave for sub-step k8: 0..3 input (slots (4 idx + w1) * 64 + lane), 4..11 filters
  auto issue = [&](int k8, int idx) __attribute__((always_inline)) {
#ifdef WINO_EXP_NODMA
    if (k8 > 0) return;
#endif
    float* st = smem + (k8 & 1) * kStageFloats;
    if (idx < 4) {
      const bool ok = ((a_okmask >> idx) & 1u) && (k8 * 8 + (int)((a_okmask >> (8 + idx)) & 1u) * 4 < p.Cin);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(st + ((4 * idx + w1) * 64) * 4), 16, (int)(ok ? a_off[idx] + (unsigned)k8 * 32u : kOobOffset), 0, 0, 0);
    } else {
      const int j = idx - 4;
      const unsigned ub = u_off0 + (unsigned)k8 * (kUSlots * 16u) + (unsigned)((4 * j + w1) * 64 + lane) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(st + kInFloats + ((4 * j + w1) * 64) * 4), 16, (int)ub, 0, 0, 0);
    }
  };

  f32x4 v[2][4], bfr[8];   // B^T d B and the filter fragments of the sub-step: everything its MFMAs read
  auto loads = [&](int k8) __attribute__((always_inline)) {
    const float* st = smem + (k8 & 1) * kStageFloats;
    f32x4 d[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#ifdef WINO_EXP_NORAW
        d[a][j] = f32x4{(float)k8, 1.f, 2.f, (float)(a + j)};
#else
        d[a][j] = *reinterpret_cast<const f32x4*>(st + ro[a][j]);
#endif
      }
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) bfr[q8] = *reinterpret_cast<const f32x4*>(st + bo + q8 * 512);
    f32x4 t[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (PH == 0) { t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; }   // xi = 0, 1 from rows 0, 1, 2
      else { t[0][j] = d[1][j] - d[0][j]; t[1][j] = d[0][j] - d[2][j]; }                     // xi = 2, 3 from rows 1, 2, 3
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) bt4(t[a][0], t[a][1], t[a][2], t[a][3], v[a]);
  };
  // the 32 MFMAs of a sub-step from v and bfr; kd >= 0: the DMA of sub-step kd is issued between them
  auto mfmas = [&](int kd) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        const f32x4 bf = bfr[a * 4 + nu];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#ifdef WINO_EXP_NOMFMA
          acc[a * 4 + nu][s] += v[a][nu][s] * bf[s];
#else
          acc[a * 4 + nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[a][nu][s], bf[s], acc[a * 4 + nu], 0, 0, 0);
#endif
          if constexpr (ROLE == 1) {
            const int g = (a * 4 + nu) * 4 + s;   // one DMA instruction behind every second MFMA of the first 24
            if ((g & 1) && g < 24 && kd >= 0) issue(kd, g >> 1);
          }
        }
      }
  };
  auto pbarrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if constexpr (ROLE == 0) {
    __syncthreads();   // A(0): sub-step 0 has landed (issued by the role-1 waves)
    for (int k8 = 0; k8 < K8; ++k8) {
      loads(k8);
      pbarrier();        // B(k8)
      mfmas(-1);
      __syncthreads();   // A(k8 + 1)
    }
  } else {
#pragma unroll
    for (int idx = 0; idx < 12; ++idx) issue(0, idx);
    __syncthreads();   // A(0)
    for (int k8 = 0; k8 < K8; ++k8) {
      const int kd = k8 + 1 < K8 ? k8 + 1 : -1;
      if (k8 > 0) mfmas(kd);
      else if (kd >= 0) {
#pragma unroll
        for (int idx = 0; idx < 12; ++idx) issue(kd, idx);
      }
      pbarrier();        // B(k8)
      loads(k8);
      __syncthreads();   // A(k8 + 1) (with the DMA of sub-step k8 + 1 drained)
    }
    mfmas(-1);
  }
}

__global__ __launch_bounds__(kNT) void wino_kernel(const IgemmArgs p_, const IgemmGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ph = wave & 1, wn = (wave >> 1) & 1, wm = wave >> 2;
  // XCD-aware order (see igemm_kernel): every XCD walks a contiguous range of (patch, N tile) pairs, the N tiles of a patch
  // back to back on one L2
  int patch, tile_n, gidx;
  {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int nwg = gx * gy * gz, bid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int cpx = nwg >> 3;
    const int t = bid < cpx * 8 ? (bid & 7) * cpx + (bid >> 3) : bid;
    tile_n = t % gy;
    patch = (t / gy) % gx;
    gidx = t / (gx * gy);
  }
  IgemmArgs p = p_;
  if (p.ngroup > 1) {
    p.x = grp.x[gidx]; p.y = grp.y[gidx]; p.bias = grp.bias[gidx]; p.mask = grp.mask[gidx]; p.res = grp.res[gidx]; p.cs = grp.cs[gidx];
  }
  const int ppi = p.GH * p.GW;
  const int n = patch / ppi, prem = patch - n * ppi, by = prem / p.GW, bx = prem - by * p.GW;
  const int oh0 = by * 16, ow0 = bx * 16, ih0 = oh0 - p.si, iw0 = ow0 - p.si;
  const int n0 = tile_n * 64;
  const int H = p.H, W = p.W, ldx = p.ldx;

  const unsigned long long img = (unsigned long long)H * W * ldx * 4ull;   // bytes of one image (< 2 GiB: wino_plan)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (size_t)n * H * W * ldx, 0,
                                                                      (unsigned)(((unsigned long long)(H * W - 1) * ldx + p.Cin) * 4ull), 0x00020000);
  (void)img;
  // transformed filters of group gidx: [N tile][chunk][2048 slots of 16 B]
  const size_t ublock = (size_t)gridDim.y * p.kchunks * kUSlots * 4;   // floats per group
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w) + (size_t)gidx * ublock, 0, (unsigned)(ublock * 4), 0x00020000);
  const unsigned u_off0 = (unsigned)tile_n * (unsigned)p.kchunks * (kUSlots * 16u);

  // input staging (waves 4-7): DMA instruction i of wave 4 + w1 fills slots (4 i + w1) * 64 + lane -> (h, class, r, c) -> patch
  // pixel (2 r + pi, 2 c + pj), channels 4h .. 4h + 3 of the sub-step
  unsigned a_off[4], a_okmask = 0;   // bits 0..3: slot holds a pixel of the image; bits 8..11: its channel half h
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int S = (4 * i + (wave & 3)) * 64 + lane;
    const int h = S / 432, rem = S - h * 432, cls = rem / 108, r2 = rem - cls * 108, r = r2 / 12, c = r2 - r * 12;
    const int ih = ih0 + 2 * r + (cls >> 1), iw = iw0 + 2 * c + (cls & 1);
    const bool ok = S < kInUsed && c < 9 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    a_okmask |= (ok ? 1u : 0u) << i;
    a_okmask |= (h ? 1u : 0u) << (8 + i);
    a_off[i] = ok ? (unsigned)(((ih * W + iw) * ldx + 4 * h) * 4) : 0u;
  }

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // per-column epilogue vectors (behind the two stages)
  float* sV = smem + kStagingFloats;          // [4][64]: bias, vec2, scale, shift
  float* sS = sV + 4 * 64;                    // [8 waves][2][32] column sums
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
  if (wave < 4) {
    if (ph == 0) wino_loop<0, 0>(p, smem, rx, ru, a_off, a_okmask, u_off0, lane, wave, wm, wn, acc);
    else wino_loop<1, 0>(p, smem, rx, ru, a_off, a_okmask, u_off0, lane, wave, wm, wn, acc);
  } else {
    if (ph == 0) wino_loop<0, 1>(p, smem, rx, ru, a_off, a_okmask, u_off0, lane, wave, wm, wn, acc);
    else wino_loop<1, 1>(p, smem, rx, ru, a_off, a_okmask, u_off0, lane, wave, wm, wn, acc);
  }
  __syncthreads();   // every wave is past its last LDS read: the staging area is free

  // ---- output transform.  This wave holds M[xi][nu] for xi = 2 ph, 2 ph + 1.  Row sums of A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]:
  //   ph 0: s0 = M0 + M1, s1 = M1;   ph 1: s0 = M2, s1 = -M2 - M3;   then along nu: (s[0] + s[1] + s[2], s[1] - s[2] - s[3]).
  // yp[i][jj][r]: this wave's part of output row i, column jj of the tile of accumulator register r.
  float own[2][16], give[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s0[4], s1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const float ma = acc[nu][r], mb = acc[4 + nu][r];
      if (ph == 0) { s0[nu] = ma + mb; s1[nu] = mb; }
      else { s0[nu] = ma; s1[nu] = -ma - mb; }
    }
    const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
    const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
    // this wave finishes output row i = ph and hands row 1 - ph to its partner
    own[0][r] = ph == 0 ? y00 : y10; own[1][r] = ph == 0 ? y01 : y11;
    give[0][r] = ph == 0 ? y10 : y00; give[1][r] = ph == 0 ? y11 : y01;
  }
  float* sX = smem + wave * (32 * 64);   // [32][64 lanes]
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int r = 0; r < 16; ++r) sX[(jj * 16 + r) * 64 + lane] = give[jj][r];
  __syncthreads();
  const float* sP = smem + (wave ^ 1) * (32 * 64);
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float o = sP[(jj * 16 + r) * 64 + lane];
      own[jj][r] = ph == 0 ? own[jj][r] + o : o + own[jj][r];   // (part of xi 0, 1) + (part of xi 2, 3)
    }

  // ---- element-wise epilogue + stores (order of operations: epilogue_store of igemm_kernel.hpp)
  const int f = p.flags;
  const bool has_res = (f & CRDR_EPI_RES) != 0, has_mask = (f & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) != 0;
  const bool do_cs = (f & CRDR_EPI_COLSUM) != 0, accum = (f & CRDR_EPI_ACCUM) != 0;
  const size_t opix_img = (size_t)n * p.OH * p.OW;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + opix_img * p.ldy, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_res ? p.res + opix_img * p.ldres : p.y), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_mask ? p.mask + opix_img * p.ldmask : p.y), 0, 0x7fffffff, 0x00020000);
  const int cn = wn * 32 + (lane & 31), oc = n0 + cn, fh = lane >> 5;
  const bool oc_ok = oc < p.Cout;
  const float bias = sV[0 * 64 + cn], vec2 = sV[1 * 64 + cn], scale = sV[2 * 64 + cn], shift = sV[3 * 64 + cn];
  float cpre = 0.f, cpost = 0.f;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {   // batches of 4 accumulator registers x 2 columns = 8 pixels: loads first, then the stores
    float resv[8], mskv[8], oldv[8];
    unsigned yo[8];
    bool okk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = rb * 4 + (q >> 1), jj = q & 1;
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * fh;   // MFMA row = tile of the wave's 8 x 4 block
      const int oh = oh0 + 2 * (trow >> 2) + ph, ow = ow0 + 2 * ((trow & 3) + 4 * wm) + jj;
      const bool ok = oc_ok && oh < p.OH && ow < p.OW;
      const unsigned pix = (unsigned)(oh * p.OW + ow);
      okk[q] = ok;
      yo[q] = ok ? (pix * (unsigned)p.ldy + (unsigned)oc) * 4u : kOobOffset;
      if (has_res) resv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, ok ? (pix * (unsigned)p.ldres + (unsigned)oc) * 4u : kOobOffset, 0, 0));
      if (has_mask) mskv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, ok ? (pix * (unsigned)p.ldmask + (unsigned)oc) * 4u : kOobOffset, 0, 0));
      if (accum) oldv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, yo[q], 0, 0));
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = rb * 4 + (q >> 1), jj = q & 1;
      float v = own[jj][r];
      if (f & CRDR_EPI_BIAS) v += bias;
      if (f & CRDR_EPI_RELU) v = fmaxf(v, 0.0f);
      if (f & CRDR_EPI_LRELU) v = v > 0.0f ? v : 0.2f * v;
      if (f & CRDR_EPI_VEC2) v += vec2;
      if (has_res) v += resv[q];
      if (f & CRDR_EPI_AFFINE) v = v * scale + shift;
      if (do_cs) cpre += okk[q] ? v : 0.f;
      if (has_mask) {
        float mv = mskv[q];
        if (f & CRDR_EPI_MASKOFF) mv -= vec2;
        v = mv > 0.0f ? v : ((f & CRDR_EPI_LRELUMASK) ? 0.2f * v : 0.0f);
      }
      if (do_cs) cpost += okk[q] ? v : 0.f;
      if (accum) v += oldv[q];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, yo[q], 0, 0);
    }
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
      if (n0 + c < p.Cout) p.cs[((size_t)patch * 2 + which) * p.cs_ld + n0 + c] = v;
    }
  }
}

// Filter transform U = G g G^T, G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]], from the implicit-GEMM weight pack
// (tap-major [tap][wrows][wcols]) into the stage-block layout of wino_kernel: [N tile][chunk of 8 channels][position 16][h 2][oc 64][4].
// One thread per (N tile, chunk, h, oc): 9 x 16 B in, 16 x 16 B out (consecutive threads = consecutive oc: coalesced both ways).
struct WinoTaps { int widx[9]; };   // weight index of tap (a, b) = patch offset (a, b) relative to the first tap
__global__ void wino_filter_kernel(const IgemmGroup grp, int ngroup, const float* w0, float* u, int Cin, int Cout, int wrows, int wcols, int kchunks,
                                   int ntile, WinoTaps tp) {
  const long long total = (long long)ntile * kchunks * 128;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int g = blockIdx.y;
  const float* w = ngroup > 1 ? grp.w[g] : w0;
  const int oc64 = (int)(id & 63), h = (int)((id >> 6) & 1);
  const long long blk = id >> 7;   // (N tile, chunk)
  const int kc = (int)(blk % kchunks), ct = (int)(blk / kchunks);
  const int oc = ct * 64 + oc64, c0 = kc * 8 + h * 4;
  f32x4 g9[3][3];
  const bool live = oc < Cout && c0 < Cin;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (live) v = *reinterpret_cast<const f32x4*>(w + ((size_t)tp.widx[a * 3 + b] * wrows + oc) * wcols + c0);
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
  float* dst = u + ((size_t)g * ntile * kchunks + (size_t)blk) * (kUSlots * 4) + ((size_t)h * 64 + oc64) * 4;
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

bool wino_eligible(const crdr_conv_desc* d, int G) {
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->wlayout != 0) return false;
  const int grow = d->transposed ? 2 - 2 * d->pad : 2 * d->pad - 2;   // (a stride-1 transposed conv = a conv with pad 2 - pad)
  if (d->OH != d->H + grow || d->OW != d->W + grow) return false;
  if (d->pad < 0 || d->pad > 2) return false;
  if (d->C % 4 != 0 || d->ldx % 4 != 0) return false;
  if (d->flags & (CRDR_EPI_GATE | CRDR_EPI_PREADD | CRDR_CONV_BF16X3)) return false;
  if (G > 1 && (d->flags & (CRDR_EPI_VEC2 | CRDR_EPI_AFFINE | CRDR_EPI_MASKOFF))) return false;
  const long long img = (long long)d->H * d->W * d->ldx * 4;
  const long long oimg = (long long)d->OH * d->OW * std::max(std::max(d->ldy, d->ldres), d->ldmask) * 4;
  if (img >= (1ll << 31) || oimg >= (1ll << 31)) return false;
  const long long ub = (long long)cdiv(d->OC, 64) * cdiv(d->C, 8) * kUSlots * 16;
  if (ub >= (1ll << 31)) return false;
  return true;
}

size_t wino_workspace(const crdr_conv_desc* d, int G) {
  return (size_t)G * cdiv(d->OC, 64) * cdiv(d->C, 8) * kUSlots * 16;
}

int wino_colsum_rows(const crdr_conv_desc* d) { return d->N * cdiv(d->OH, 16) * cdiv(d->OW, 16); }

// a: the argument block of the implicit-GEMM plan with every pointer / stride / flag filled in; taps: the plan's tap table
int wino_launch(const crdr_conv_desc* d, IgemmArgs a, const IgemmTaps& taps, const IgemmGroup& grp, int G, float* u, hipStream_t s) {
  CRDR_REQUIRE(wino_eligible(d, G), "conv2d: the Winograd kernel takes 3x3 stride-1 convolutions (C %% 4 == 0, no gate / pre-add epilogue)");
  WinoTaps wt;
  int dmin = 127;
  for (int t = 0; t < 9; ++t) dmin = std::min(dmin, (int)(signed char)(taps.packed[t] & 0xff));
  for (int t = 0; t < 9; ++t) wt.widx[t] = -1;
  for (int t = 0; t < 9; ++t) {
    const int v = taps.packed[t];
    const int dh = (int)(signed char)(v & 0xff) - dmin, dw = (int)(signed char)((v >> 8) & 0xff) - dmin;
    CRDR_REQUIRE(dh >= 0 && dh < 3 && dw >= 0 && dw < 3, "conv2d: Winograd: tap offsets are not a 3x3 window");
    wt.widx[dh * 3 + dw] = v >> 16;
  }
  for (int t = 0; t < 9; ++t) CRDR_REQUIRE(wt.widx[t] >= 0, "conv2d: Winograd: incomplete 3x3 window");
  const int ntile = cdiv(d->OC, 64), kchunks = cdiv(d->C, 8);
  {
    const long long total = (long long)ntile * kchunks * 128;
    hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)cdiv64(total, 256), G), dim3(256), 0, s, grp, G, a.w, u, d->C, d->OC, d->wrows, d->wcols, kchunks,
                       ntile, wt);
    CRDR_CHECK_LAUNCH("wino_filter_kernel");
  }
  a.w = u;
  a.kchunks = kchunks;
  a.GH = cdiv(d->OH, 16);
  a.GW = cdiv(d->OW, 16);
  a.si = -dmin;   // the patch starts `si` pixels above / left of its first output pixel
  a.cs_rows = wino_colsum_rows(d);
  const size_t lds = (size_t)(kStagingFloats + 4 * 64 + 8 * 2 * 32) * sizeof(float);
  static std::atomic<bool> attr_done{false};
  if (!attr_done.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(wino_kernel, dim3(d->N * a.GH * a.GW, ntile, G), dim3(kNT), lds, s, a, grp);
  CRDR_CHECK_LAUNCH("wino_kernel");
  return 0;
}

}  // namespace crdr
