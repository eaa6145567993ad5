// Generalised divisive normalisation (Balle et al. 2016) as used by the reference's ablation transforms through
// compressai.layers.GDN (src/models/subnet/autoencoder/balle18_autoencoder.py:16-20,37-41; src/models/layer/
// cheng_resblock.py:8-15):
//
//   n[p][i] = beta_i + sum_j gamma_ij x[p][j]^2,    y = x * n^(-1/2)  (GDN)   or   y = x * n^(1/2)  (inverse, "IGDN")
//
// with compressai's NonNegativeParametrizer on both parameters: v_eff = max(v_param, bound)^2 - pedestal,
// pedestal = reparam_offset^2, bound_beta = sqrt(beta_min + pedestal), bound_gamma = reparam_offset; the max() carries the
// LowerBound gradient rule (pass where v >= bound or the gradient would raise v).
//
// Up to 192 channels the forward and the backward are fused persistent kernels (below); beyond, the channel mix (a C x C GEMV
// per pixel) runs on the fp32 matrix cores as a 1x1 launch of the implicit-GEMM kernel on x^2 and the rest are single-pass
// elementwise kernels (16 B per lane).  Not on the CRDR training path -- the ELIC
// transforms use ReLU bottlenecks -- so this is a registered optional op, parity-tested and profiled on its own.

#include <algorithm>
#include <atomic>
#include <cmath>

#include "common.hpp"

namespace crdr {

__global__ __launch_bounds__(256) void gdn_reparam_kernel(const float* beta_p, const float* gamma_p, int C, int CP, float bound_b,
                                                          float bound_g, float ped, float* beta_eff, float* pack_f, float* pack_b) {
  const int total = CP * CP;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int i = e / CP, j = e - i * CP;
    float g = 0.f;
    if (i < C && j < C) {
      const float v = fmaxf(gamma_p[i * C + j], bound_g);
      g = v * v - ped;
    }
    pack_f[i * CP + j] = g;   // rows = output channel i, cols = input channel j
    pack_b[j * CP + i] = g;   // transposed: the input-gradient operand
    if (j == 0 && i < C) {
      const float v = fmaxf(beta_p[i], bound_b);
      beta_eff[i] = v * v - ped;
    }
  }
}

__global__ __launch_bounds__(256) void gdn_square_kernel(const float* x, int ldx, int64_t M, int C4, float* x2) {
  const int64_t total = M * C4;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int64_t m = e / C4;
    const int c = (int)(e - m * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ldx + c);
    *reinterpret_cast<f32x4*>(x2 + m * (C4 * 4) + c) = v * v;
  }
}

__global__ __launch_bounds__(256) void gdn_apply_kernel(const float* x, int ldx, const float* norm, int64_t M, int C4, int inverse,
                                                        float* y, int ldy) {
  const int64_t total = M * C4;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int64_t m = e / C4;
    const int c = (int)(e - m * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ldx + c);
    const f32x4 n = *reinterpret_cast<const f32x4*>(norm + m * (C4 * 4) + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = inverse ? v[k] * sqrtf(n[k]) : v[k] / sqrtf(n[k]);
    *reinterpret_cast<f32x4*>(y + m * ldy + c) = o;
  }
}

// dn = dL/dn, u = dy * dy/dx|_n
__global__ __launch_bounds__(256) void gdn_bwd_prep_kernel(const float* x, int ldx, const float* norm, const float* dy, int lddy,
                                                           int64_t M, int C4, int inverse, float* dn, float* u) {
  const int64_t total = M * C4;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int64_t m = e / C4;
    const int c = (int)(e - m * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ldx + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + m * lddy + c);
    const f32x4 n = *reinterpret_cast<const f32x4*>(norm + m * (C4 * 4) + c);
    f32x4 a, b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float r = sqrtf(n[k]);
      if (inverse) { a[k] = 0.5f * g[k] * v[k] / r; b[k] = g[k] * r; }
      else { a[k] = -0.5f * g[k] * v[k] / (n[k] * r); b[k] = g[k] / r; }
    }
    *reinterpret_cast<f32x4*>(dn + m * (C4 * 4) + c) = a;
    *reinterpret_cast<f32x4*>(u + m * (C4 * 4) + c) = b;
  }
}

// dx = u + 2 x w,  w = gamma^T dn
__global__ __launch_bounds__(256) void gdn_bwd_finish_kernel(const float* x, int ldx, const float* u, const float* w, int64_t M,
                                                             int C4, float* dx, int lddx) {
  const int64_t total = M * C4;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int64_t m = e / C4;
    const int c = (int)(e - m * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ldx + c);
    const f32x4 a = *reinterpret_cast<const f32x4*>(u + m * (C4 * 4) + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w + m * (C4 * 4) + c);
    *reinterpret_cast<f32x4*>(dx + m * lddx + c) = a + 2.0f * v * b;
  }
}

// chain the gradients of the effective parameters through max(v, bound)^2 - pedestal (LowerBound rule), accumulating
__global__ __launch_bounds__(256) void gdn_reparam_bwd_kernel(const float* dgamma_eff, const float* dbeta_eff, const float* gamma_p,
                                                              const float* beta_p, int C, float bound_b, float bound_g,
                                                              float* dgamma_p, float* dbeta_p) {
  const int total = C * C;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const float v = gamma_p[e], g = dgamma_eff[e] * 2.0f * fmaxf(v, bound_g);
    dgamma_p[e] += (v >= bound_g || g < 0.f) ? g : 0.f;
    if (e < C) {
      const float b = beta_p[e], gb = dbeta_eff[e] * 2.0f * fmaxf(b, bound_b);
      dbeta_p[e] += (b >= bound_b || gb < 0.f) ? gb : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Fused forward: ONE kernel reads x once and writes y once (8 B per element of HBM traffic; the four-kernel form above
// moves 28 B).  A persistent workgroup (4 waves, one per SIMD) keeps gamma_eff IN REGISTERS for the whole launch -- wave w owns the
// output channels [16 NCB w, 16 NCB (w + 1)) and holds their gamma rows as A fragments of v_mfma_f32_16x16x4_f32 (NCB x NS x 4
// registers per lane: 144 at 192 channels) -- and walks tiles of 64 pixels: the x tile goes to LDS by LDS-DMA as chunk images [64 rows]
// [32 floats] (the igemm A-tile image: 128-byte rows, XOR-swizzled 16-byte slots), double buffered, every wave reads each pixel row's 16
// channels of a K step with one ds_read_b128 and squares them on the way to the matrix cores (exact fp32 MFMA, the pixel tile as the B
// operand: a lane's four accumulator elements are four consecutive channels of one pixel), and the epilogue -- n = acc + beta,
// y = x rsqrt(n) or x sqrt(n) -- takes x from the LDS tile with one 16-byte read per (row block, channel block) and keeps y in registers.
// Every vector-memory request of a tile is issued INSIDE the matrix loop, one per group of MFMAs and behind them: the next tile's NS
// LDS-DMA requests in the first half of the groups, the previous tile's NS result stores (64-byte row segments) in the second -- issued
// in a row at the top / bottom of the tile they cost 22 + 10 us of a 200 us launch.  One barrier per tile (the buffer hand-over); no LDS
// traffic for gamma.  History: 128-row tiles with gamma streamed through LDS in six K slabs (a barrier each): 65 - 77 TFLOP/s of the mix;
// gamma in registers, results rewritten in the LDS tile in place (three barriers per tile): 92 - 100; this form: 107 - 118.
// Roofline: 2 C^2 FLOP against 8 C bytes per pixel = C / 4 FLOP per byte: at C = 192 the exact-fp32 matrix peak
// (157 TFLOP/s) caps the op at 3.3 TB/s = 0.41 of the HBM peak, at C = 128 at 0.61.
// mode 0: y = GDN / IGDN(x); mode 1: y = n (the norm, for the unfused backward).  C <= 192; K and the channel blocks are padded to
// NS = 4, 8 or 12 steps of 16 (the padding multiplies zeros: x beyond C is zero-filled by the DMA's range check).
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* gdn_lds_ptr_t;
struct GdnFusedArgs {
  const float* x;
  const float* pack;   // gamma_eff [CP][CP] (row = output channel, K-contiguous)
  const float* beta;   // beta_eff [C]
  float* y;
  long long M;
  int C, CP, ldx, ldy, inverse, mode, tiles;
  unsigned x_bytes, pack_bytes;
};

constexpr int kGdnBM = 64;

// x^2 of a fragment on the way to the matrix cores: two v_pk_mul_f32.  Nothing vector hides beside an exact-fp32 MFMA and a packed
// instruction costs the wave what a scalar one does, so the fewer the better: four v_mul_f32 by inline assembly measured
// 2.5 % slower on the forward kernel (184 against 180 us)
__device__ __forceinline__ f32x4 gdn_square4(f32x4 v) { return v * v; }

template <int VM>
__device__ __forceinline__ void gdn_wait_barrier() {   // vmcnt(VM) lgkmcnt(0), then the workgroup barrier
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt((VM & 15) | ((VM >> 4) << 14) | 0x70);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

template <int NS, int OP>   // OP 0: y = x n^(-1/2) (GDN); 1: y = x n^(1/2) (IGDN); 2: y = n
__global__ __launch_bounds__(256) void gdn_fused_fwd_kernel(const GdnFusedArgs p) {
  constexpr int BM = kGdnBM, NCB = NS / 4, NBX = NS / 2, XF = NBX * BM * 32;   // NBX chunk images of 32 channels; XF floats per x tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;                    // [2][NBX][BM * 32]
  float* sBeta = smem + 2 * XF;        // [64 NCB]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lg = lane >> 4;
  const int col0 = wave * 16 * NCB;    // first output channel of this wave
  if ((int)blockIdx.x >= p.tiles) return;
  for (int c = tid; c < 64 * NCB; c += 256) sBeta[c] = c < p.C ? p.beta[c] : 1.f;
  // gamma rows of this wave's channels, once: B[cb][s][e] = gamma[col0 + 16 cb + ln][16 s + 4 lg + e] (rows past the pack: zeros by the range
  // check; columns past CP inside a row meet x = 0)
  f32x4 B[NCB][NS];
  {
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack), 0, p.pack_bytes, 0x00020000);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int row = col0 + 16 * cb + ln, k = 16 * s + 4 * lg;
        const bool ok = row < p.CP && k < p.CP;
        B[cb][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? ((unsigned)row * p.CP + k) * 4u : 0x80000000u, 0, 0));
      }
  }
  // staging assignment (as igemm): thread fills slot (tid & 7) of rows (tid >> 3) + 32 j with source chunk slot ^ swizzle(row)
  const int srow = tid >> 3;
  const int csrc = (tid & 7) ^ ((srow >> 1) & 7);
  // tile-independent lane offsets, once per launch (every instruction between two tiles' MFMAs is matrix time lost).  Rows past the
  // tensor's end need no test of their own: the descriptors of a tile end at its last valid row, so the range check drops them.
  unsigned xoff[NBX][2];
#pragma unroll
  for (int kc = 0; kc < NBX; ++kc)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int chx = 32 * kc + csrc * 4;                                       // DMA: this lane's source chunk (swizzled)
      xoff[kc][j] = chx < p.C ? ((unsigned)(srow + 32 * j) * p.ldx + chx) * 4u : 0x80000000u;
    }
  // the lane's results: channels ch(cb) = col0 + 16 cb + 4 lg .. + 3 of pixel row 16 rb + ln (gamma is the A operand of the MFMAs, the pixel
  // tile the B operand: four accumulator elements = four consecutive channels, one 16-byte access per (rb, cb) on either side)
  bool chok[NCB];
  int xe[NCB];   // float offset of (row ln, channels ch(cb)) inside an x tile; row block rb adds 512 rb (the swizzle repeats every 16 rows)
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int ch = col0 + 16 * cb + 4 * lg;
    chok[cb] = ch < p.C;
    xe[cb] = (ch >> 5) * BM * 32 + lds_off(ln, (ch & 31) >> 2);
  }
  const unsigned ybase = ((unsigned)ln * p.ldy + col0 + 4 * lg) * 4u;
  auto tile_rsrc = [&](const float* base, int ld, long long m0) __attribute__((always_inline)) {
    const long long left = p.M - m0;
    const long long bytes = left <= 0 ? 0 : ((left < BM ? left : BM) - 1) * ld * 4ll + p.C * 4ll;   // (a tile past the last one: no bytes at all)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + (left <= 0 ? 0 : m0 * ld), 0, (unsigned)std::min<long long>(bytes, 0x7fffffffll), 0x00020000);
  };
  // an x tile: chunk image kc, rows srow + 32 j, j < 2: NS requests
  // request q (< NS) of a tile: chunk image q >> 1, rows srow + 32 (q & 1)
  auto fetch_piece = [&](const __amdgpu_buffer_rsrc_t rx, int buf, int q) __attribute__((always_inline)) {
    const int kc = q >> 1, j = q & 1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (gdn_lds_ptr_t)(sX + buf * XF + kc * BM * 32 + wave * 8 * 32 + j * 32 * 32), 16, (int)xoff[kc][j], 0, 0, 0);
  };
  auto fetch_x = [&](long long m0, int buf) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rx = tile_rsrc(p.x, p.ldx, m0);
#pragma unroll
    for (int q = 0; q < NS; ++q) fetch_piece(rx, buf, q);
  };
  int t = blockIdx.x, cur = 0;
  fetch_x((long long)t * BM, 0);
  bool first = true;
  // the results of a tile leave during the NEXT tile's matrix loop (one store per group, behind its MFMAs): before the first tile the
  // descriptor is empty and the NS stores are dropped, so that every tile issues the same requests
  f32x4 yout[4][NCB];
  __amdgpu_buffer_rsrc_t ry_prev = tile_rsrc(p.y, p.ldy, p.M);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) yout[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto store_piece = [&](const __amdgpu_buffer_rsrc_t ry, int q) __attribute__((always_inline)) {
    const int rb = q / NCB, cb = q % NCB;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, yout[rb][cb]), ry,
                                           chok[cb] ? ybase + (unsigned)(16 * rb * p.ldy * 4 + 64 * cb) : 0x80000000u, 0, 0);
  };
  for (; t < p.tiles; t += gridDim.x, cur ^= 1) {
    const long long m0 = (long long)t * BM;
    // this tile's x has landed (everything but the NS result stores issued after its requests, the youngest ones); every wave is done with the other
    // buffer (matrix loop AND epilogue: the results leave from registers, nothing is rewritten in LDS -- this is the only barrier of a tile)
    if (first) gdn_wait_barrier<0>();
    else gdn_wait_barrier<NS>();
    first = false;
    // the next tile's NS requests go out one per group of the matrix loop, behind that group's MFMAs (an LDS-DMA request issued among bare
    // MFMAs costs ~60 cycles of issue; twelve in a row at the top of the tile were 22 us of the launch).  No branch: past the last tile the
    // descriptor is empty
    const __amdgpu_buffer_rsrc_t rnext = tile_rsrc(p.x, p.ldx, (long long)(t + (int)gridDim.x) * BM);
    f32x4 acc[4][NCB];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* xt = sX + cur * XF;
    // 2 NS groups (K step s, row blocks rb, rb + 1) of 8 NCB MFMAs on 2 NCB independent accumulators (with one row block per group -- three
    // accumulators in turn -- the loop ran at 44 cycles per MFMA, with two at 40; the instruction's issue rate is 32, its dependent latency
    // 40).  The pixel fragments of group g + 1 (pixel rows 16 rb + ln, channels 16 s + 4 lg .. + 3: chunk image s >> 1, slot 4 (s & 1) + lg) are
    // requested BEFORE the MFMAs of group g are issued and squared after them -- left to the compiler every ds_read_b128 sat directly in
    // front of its use with a full lgkmcnt(0) wait
    auto a_ptr = [&](int s, int rb) __attribute__((always_inline)) {
      return reinterpret_cast<const f32x4*>(xt + (s >> 1) * BM * 32 + lds_off(16 * rb + ln, 4 * (s & 1) + lg));
    };
    f32x4 n0 = *a_ptr(0, 0), n1 = *a_ptr(0, 1);
#pragma unroll
    for (int g = 0; g < NS * 2; ++g) {
      const int s = g >> 1, rb = 2 * (g & 1);
      const f32x4 a0 = gdn_square4(n0), a1 = gdn_square4(n1);
      if (g + 1 < NS * 2) { n0 = *a_ptr((g + 1) >> 1, 2 * ((g + 1) & 1)); n1 = *a_ptr((g + 1) >> 1, 2 * ((g + 1) & 1) + 1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[cb][s][e], a0[e], acc[rb][cb], 0, 0, 0);
          acc[rb + 1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[cb][s][e], a1[e], acc[rb + 1][cb], 0, 0, 0);
        }
      if (g < NS) fetch_piece(rnext, cur ^ 1, g);
      else store_piece(ry_prev, g - NS);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: acc[rb][cb][i] is (pixel row 16 rb + ln, channel ch(cb) + i); x comes from the LDS tile again, the result leaves with one
    // 16-byte store per (rb, cb) -- 64-byte row segments, the two halves of a 128-byte line in consecutive instructions of the wave.
    // (v_rsq_f32 / v_sqrt_f32: 1 ulp, one instruction each; n >= beta_min > 0: no denormal path)
    const __amdgpu_buffer_rsrc_t ry = tile_rsrc(p.y, p.ldy, m0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xt + xe[cb] + 512 * rb);
        const f32x4 bta = *reinterpret_cast<const f32x4*>(sBeta + col0 + 16 * cb + 4 * lg);
        f32x4 y;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float n = acc[rb][cb][i] + bta[i];
          y[i] = OP == 2 ? n : (OP == 1 ? xv[i] * __builtin_amdgcn_sqrtf(n) : xv[i] * __builtin_amdgcn_rsqf(n));
        }
        yout[rb][cb] = y;
      }
    ry_prev = ry;
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) store_piece(ry_prev, q);   // the last tile's results
}

template <int NS, int OP>
static void gdn_fused_launch_op(const GdnFusedArgs& a, hipStream_t s) {
  const size_t lds = ((size_t)2 * (NS / 2) * kGdnBM * 32 + 64 * (NS / 4)) * sizeof(float);
  static std::atomic<bool> done{false};
  if (!done.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gdn_fused_fwd_kernel<NS, OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((gdn_fused_fwd_kernel<NS, OP>), dim3(std::min(a.tiles, 256)), dim3(256), lds, s, a);
}

template <int NS>
static void gdn_fused_launch(const GdnFusedArgs& a, hipStream_t s) {
  if (a.mode) gdn_fused_launch_op<NS, 2>(a, s);
  else if (a.inverse) gdn_fused_launch_op<NS, 1>(a, s);
  else gdn_fused_launch_op<NS, 0>(a, s);
}

// ------------------------------------------------------------------------------------------------------------
// The backward in ONE pass: both channel mixes of a tile inside the same workgroup.  gamma (for n = beta + gamma x^2) stays in registers
// as in the forward kernel; its transpose (for w = gamma^T dn) does NOT fit beside it (2 x 144 registers per lane at 192 channels left the
// allocator 89 spills) and is STREAMED instead: every K step's fragments (NCB x 16 bytes per lane) come from the L2-resident pack through a
// ring of three steps, requested three steps (144 MFMAs) ahead.  dn crosses the waves through a third LDS tile (each wave writes its
// channels in the tile image the matrix loop reads; one barrier), u = dy n^(-1/2) never leaves the registers that received dy.
// HBM traffic: x and dy in, dn (for the gamma gradient) and dx out -- 16 B per element, against 32 B for a two-pass form (x, dy -> dn, u;
// dn, x, u -> dx: measured 248 + 236 us at 16 x 192 x 128 x 128, both passes bound by their 16 B per element at ~5 TB/s with the matrix loop
// only partly hidden; this kernel: 375 - 400 us), and the two matrix loops of a tile (2 x 576 MFMAs) hide the tile's 192 KB.
// ------------------------------------------------------------------------------------------------------------
struct GdnBwd1Args {
  const float* x;
  const float* dy;
  const float* pack_f;   // gamma_eff [CP][CP] (row = output channel i)
  const float* pack_b;   // its transpose (row = input channel j)
  const float* beta;     // beta_eff [C]
  float* dn;             // [M][ld_dn]
  float* dx;
  float* colpart;        // [gridDim.x][CP]
  long long M;
  int C, CP, ldx, lddy, ld_dn, lddx, inverse, tiles;
  unsigned pack_bytes;
};

template <int NS, bool INV>
__global__ __launch_bounds__(256) void gdn_bwd_onepass_kernel(const GdnBwd1Args p) {
  constexpr int BM = kGdnBM, NCB = NS / 4, NBX = NS / 2, XF = NBX * BM * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;             // [2][NBX][BM * 32]
  float* sD = smem + 2 * XF;    // [NBX][BM * 32]: dn of the current tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lg = lane >> 4;
  const int col0 = wave * 16 * NCB;
  if ((int)blockIdx.x >= p.tiles) return;
  f32x4 Bf[NCB][NS];
  {
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack_f), 0, p.pack_bytes, 0x00020000);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int row = col0 + 16 * cb + ln, k = 16 * s + 4 * lg;
        const unsigned off = (row < p.CP && k < p.CP) ? ((unsigned)row * p.CP + k) * 4u : 0x80000000u;
        Bf[cb][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rf, off, 0, 0));
      }
  }
  // the transposed pack, streamed: K step s of channel block cb is 16 bytes at boff[cb] + 64 s (K steps past the pack: zeros)
  const __amdgpu_buffer_rsrc_t rpb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack_b), 0, p.pack_bytes, 0x00020000);
  unsigned boff[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int row = col0 + 16 * cb + ln;
    boff[cb] = row < p.CP ? ((unsigned)row * p.CP + 4 * lg) * 4u : 0x80000000u;
  }
  constexpr int RING = 3;
  auto load_b = [&](int s, f32x4 (&dst)[NCB]) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
      dst[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpb, 16 * s < p.CP ? boff[cb] + 64u * s : 0x80000000u, 0, 0));
  };
  const int srow = tid >> 3;
  const int csrc = (tid & 7) ^ ((srow >> 1) & 7);
  // request offsets are rebuilt from a few per-lane bases inside the tile loop (the bases are made opaque there): hoisted out of the loop as
  // loop invariants, the ~60 offsets of a tile's requests do not fit beside gamma and come back from scratch memory
  unsigned xbase = ((unsigned)srow * p.ldx + csrc * 4) * 4u;
  float* sBeta = smem + 3 * XF;   // [64 NCB]
  for (int c = tid; c < 64 * NCB; c += 256) sBeta[c] = c < p.C ? p.beta[c] : 1.f;
  bool chok[NCB];
  int xe[NCB];   // float offset of (row ln, channels ch(cb) .. + 3) inside a tile image; row block rb adds 512 rb
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int ch = col0 + 16 * cb + 4 * lg;
    chok[cb] = ch < p.C;
    xe[cb] = (ch >> 5) * BM * 32 + lds_off(ln, (ch & 31) >> 2);
  }
  unsigned lb_dy = ((unsigned)ln * p.lddy + col0 + 4 * lg) * 4u, lb_dn = ((unsigned)ln * p.ld_dn + col0 + 4 * lg) * 4u,
           lb_dx = ((unsigned)ln * p.lddx + col0 + 4 * lg) * 4u;
  auto lane_off = [&](unsigned lb, int ld, int cb, int rb) __attribute__((always_inline)) {
    return chok[cb] ? lb + (unsigned)(16 * rb * ld * 4 + 64 * cb) : 0x80000000u;
  };
  auto tile_rsrc = [&](const float* base, int ld, long long m0) __attribute__((always_inline)) {
    const long long left = p.M - m0;
    const long long bytes = left <= 0 ? 0 : ((left < BM ? left : BM) - 1) * ld * 4ll + p.C * 4ll;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + (left <= 0 ? 0 : m0 * ld), 0, (unsigned)std::min<long long>(bytes, 0x7fffffffll), 0x00020000);
  };
  // request q (< NS) of an x tile: chunk image q >> 1, rows srow + 32 (q & 1)
  auto fetch_piece = [&](const __amdgpu_buffer_rsrc_t rx, int buf, int q) __attribute__((always_inline)) {
    const int kc = q >> 1, j = q & 1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (gdn_lds_ptr_t)(sX + buf * XF + kc * BM * 32 + wave * 8 * 32 + j * 32 * 32), 16,
                                             (int)(32 * kc + csrc * 4 < p.C ? xbase + (unsigned)(128 * kc + j * 32 * p.ldx * 4) : 0x80000000u), 0, 0, 0);
  };
  auto fetch = [&](long long m0, int buf) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rx = tile_rsrc(p.x, p.ldx, m0);
#pragma unroll
    for (int q = 0; q < NS; ++q) fetch_piece(rx, buf, q);
  };
  f32x4 csum[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) csum[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  int t = blockIdx.x, cur = 0;
  fetch((long long)t * BM, 0);
  bool first = true;
  for (; t < p.tiles; t += gridDim.x, cur ^= 1) {
    const long long m0 = (long long)t * BM;
    asm volatile("" : "+v"(xbase), "+v"(lb_dy), "+v"(lb_dn), "+v"(lb_dx));
    // x of this tile has landed (everything but the previous tile's 2 NS stores); every wave is done with the other x buffer and with sD
    if (first) gdn_wait_barrier<0>();
    else gdn_wait_barrier<2 * NS>();
    first = false;
    // dy (then u) in registers; its NS requests and the next x tile's NS LDS-DMA requests go out one of each per group of the first matrix
    // loop, behind that group's MFMAs (issued in a row at the top of the tile they cost the launch 40 us)
    f32x4 g[4][NCB];
    const __amdgpu_buffer_rsrc_t q0 = tile_rsrc(p.dy, p.lddy, m0);
    const __amdgpu_buffer_rsrc_t rnext = tile_rsrc(p.x, p.ldx, (long long)(t + (int)gridDim.x) * BM);
    float* xt = sX + cur * XF;
    f32x4 acc[4][NCB];
    auto a_ptr = [&](const float* tile, int s, int rb) __attribute__((always_inline)) {
      return reinterpret_cast<const f32x4*>(tile + (s >> 1) * BM * 32 + lds_off(16 * rb + ln, 4 * (s & 1) + lg));
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // first mix: acc[rb][cb][i] = sum_k gamma[ch][k] x^2(pixel row 16 rb + ln, k); 2 NS groups (K step, row-block pair) of 8 NCB MFMAs on 2 NCB
    // independent accumulators, fragments of the next group requested before a group's MFMAs (see the forward kernel)
    zero_acc();
    {
      f32x4 n0 = *a_ptr(xt, 0, 0), n1 = *a_ptr(xt, 0, 1);
#pragma unroll
      for (int gi = 0; gi < NS * 2; ++gi) {
        const int s = gi >> 1, rb = 2 * (gi & 1);
        const f32x4 a0 = gdn_square4(n0), a1 = gdn_square4(n1);
        if (gi + 1 < NS * 2) { n0 = *a_ptr(xt, (gi + 1) >> 1, 2 * ((gi + 1) & 1)); n1 = *a_ptr(xt, (gi + 1) >> 1, 2 * ((gi + 1) & 1) + 1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bf[cb][s][e], a0[e], acc[rb][cb], 0, 0, 0);
            acc[rb + 1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bf[cb][s][e], a1[e], acc[rb + 1][cb], 0, 0, 0);
          }
        if (gi < NS) {
          const int qr = gi / NCB, qc = gi % NCB;
          g[qr][qc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q0, lane_off(lb_dy, p.lddy, qc, qr), 0, 0));
          fetch_piece(rnext, cur ^ 1, gi);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the second mix's first K steps, requested ahead of the epilogue's stores (requests return in order)
    f32x4 ring[RING][NCB];
#pragma unroll
    for (int r = 0; r < RING; ++r) load_b(r, ring[r]);
    {
      const __amdgpu_buffer_rsrc_t w0 = tile_rsrc(p.dn, p.ld_dn, m0);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xt + xe[cb] + 512 * rb);
          const f32x4 gy = g[rb][cb];
          const f32x4 bta = *reinterpret_cast<const f32x4*>(sBeta + col0 + 16 * cb + 4 * lg);
          f32x4 dn, u;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float n = acc[rb][cb][i] + bta[i];
            const float rs = __builtin_amdgcn_rsqf(n);
            if constexpr (INV) { u[i] = gy[i] * __builtin_amdgcn_sqrtf(n); dn[i] = 0.5f * gy[i] * xv[i] * rs; }
            else { u[i] = gy[i] * rs; dn[i] = -0.5f * u[i] * xv[i] * (rs * rs); }
          }
          g[rb][cb] = u;
          csum[cb] += dn;
          *reinterpret_cast<f32x4*>(sD + xe[cb] + 512 * rb) = dn;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, dn), w0, lane_off(lb_dn, p.ld_dn, cb, rb), 0, 0);
        }
    }
    gdn_wait_barrier<63>();   // every wave's dn is in sD (lgkmcnt(0) + barrier; vector memory is not waited for)
    // second mix: acc[rb][cb][i] = sum_k gamma[k][ch] dn(pixel row 16 rb + ln, k)
    zero_acc();
    {
      f32x4 n0 = *a_ptr(sD, 0, 0), n1 = *a_ptr(sD, 0, 1);
#pragma unroll
      for (int gi = 0; gi < NS * 2; ++gi) {
        const int s = gi >> 1, rb = 2 * (gi & 1);
        const f32x4 a0 = n0, a1 = n1;
        if (gi + 1 < NS * 2) { n0 = *a_ptr(sD, (gi + 1) >> 1, 2 * ((gi + 1) & 1)); n1 = *a_ptr(sD, (gi + 1) >> 1, 2 * ((gi + 1) & 1) + 1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[s % RING][cb][e], a0[e], acc[rb][cb], 0, 0, 0);
            acc[rb + 1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[s % RING][cb][e], a1[e], acc[rb + 1][cb], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (rb == 2 && s + RING < NS) load_b(s + RING, ring[s % RING]);
      }
    }
    {
      const __amdgpu_buffer_rsrc_t w1 = tile_rsrc(p.dx, p.lddx, m0);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xt + xe[cb] + 512 * rb);
          const f32x4 dx = g[rb][cb] + 2.0f * xv * acc[rb][cb];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, dx), w1, lane_off(lb_dx, p.lddx, cb, rb), 0, 0);
        }
    }
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = csum[cb][i];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
      csum[cb][i] = v;
    }
    const int ch = col0 + 16 * cb + 4 * lg;
    if (ln == 0 && ch < p.CP) *reinterpret_cast<f32x4*>(p.colpart + (size_t)blockIdx.x * p.CP + ch) = csum[cb];
  }
}

template <int NS, bool INV>
static void gdn_bwd1_launch_dir(const GdnBwd1Args& a, hipStream_t s) {
  const size_t lds = ((size_t)3 * (NS / 2) * kGdnBM * 32 + 64 * (NS / 4)) * sizeof(float);
  static std::atomic<bool> done{false};
  if (!done.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gdn_bwd_onepass_kernel<NS, INV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((gdn_bwd_onepass_kernel<NS, INV>), dim3(std::min(a.tiles, 256)), dim3(256), lds, s, a);
}

template <int NS>
static void gdn_bwd1_launch(const GdnBwd1Args& a, hipStream_t s) {
  if (a.inverse) gdn_bwd1_launch_dir<NS, true>(a, s);
  else gdn_bwd1_launch_dir<NS, false>(a, s);
}

// ------------------------------------------------------------------------------------------------------------
// The gamma gradient dgamma_eff[i][j] = sum_p dn[p][i] x[p][j]^2 (a C x C result over M pixels) as a persistent kernel: the WHOLE result
// stays in the accumulators of one workgroup -- wave (wi, wj) of 2 x 2 owns the (8 NS x 8 NS)-channel quadrant as (NS / 2)^2 blocks of
// v_mfma_f32_16x16x4_f32 (144 registers per lane at 192 channels) -- while the workgroup walks 32-pixel tiles of dn and x (LDS-DMA, three
// stages, the chunk images of the kernels above; the requests ride behind the K steps' MFMAs).  A K step is four pixels: NS / 2 fragments of
// dn (A operand: lane (ln, lg) holds dn[pixel 4 k + lg][i0 + ln]) and NS / 2 of x (B operand, squared on the way), one 4-byte LDS read
// each, for (NS / 2)^2 MFMAs.  Every workgroup leaves its partial C x C sum; gdn_dgamma_finish_kernel adds them in a fixed order and chains
// the result through the parametrisation.  Against the tiled weight-gradient kernel with CRDR_WGRAD_SQUARE_Q (no LDS-resident result,
// split over 128 slabs and a separate reduction): 268 us at 16 x 192 x 128 x 128.
// ------------------------------------------------------------------------------------------------------------
struct GdnDgArgs {
  const float* dn;     // [M][ld_dn]
  const float* x;      // [M][ldx]
  float* part;         // [gridDim.x][CH * CH], CH = 16 NS
  long long M;
  int C, ld_dn, ldx, tiles;   // tiles of 32 pixels
};

constexpr int kGdnDgBP = 32;

template <int NS>
__global__ __launch_bounds__(256) void gdn_dgamma_kernel(const GdnDgArgs p) {
  constexpr int BP = kGdnDgBP, NBX = NS / 2, TF = NBX * BP * 32, HB = NS / 2, CH = 16 * NS;   // TF floats per operand tile; HB blocks per quadrant side
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sP = smem;            // [3][NBX][BP * 32]  dn
  float* sQ = smem + 3 * TF;   // [3][NBX][BP * 32]  x
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lg = lane >> 4;
  const int i0 = (wave >> 1) * 16 * HB, j0 = (wave & 1) * 16 * HB;
  if ((int)blockIdx.x >= p.tiles) return;
  // staging: thread fills slot (tid & 7) of row (tid >> 3) (32 rows: one request per chunk image and operand) with source chunk slot ^ swizzle(row)
  const int srow = tid >> 3;
  const int csrc = (tid & 7) ^ ((srow >> 1) & 7);
  unsigned pbase = ((unsigned)srow * p.ld_dn + csrc * 4) * 4u, qbase = ((unsigned)srow * p.ldx + csrc * 4) * 4u;
  auto tile_rsrc = [&](const float* base, int ld, long long m0) __attribute__((always_inline)) {
    const long long left = p.M - m0;
    const long long bytes = left <= 0 ? 0 : ((left < BP ? left : BP) - 1) * ld * 4ll + p.C * 4ll;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + (left <= 0 ? 0 : m0 * ld), 0, (unsigned)std::min<long long>(bytes, 0x7fffffffll), 0x00020000);
  };
  // request q (< 2 NBX) of a tile pair: operand q & 1 (0: dn, 1: x), chunk image q >> 1
  auto fetch_piece = [&](const __amdgpu_buffer_rsrc_t rp, const __amdgpu_buffer_rsrc_t rq, int buf, int q) __attribute__((always_inline)) {
    const int kc = q >> 1;
    const bool ok = 32 * kc + csrc * 4 < p.C;
    if (q & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (gdn_lds_ptr_t)(sQ + buf * TF + kc * BP * 32 + wave * 8 * 32), 16, (int)(ok ? qbase + 128u * kc : 0x80000000u), 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (gdn_lds_ptr_t)(sP + buf * TF + kc * BP * 32 + wave * 8 * 32), 16, (int)(ok ? pbase + 128u * kc : 0x80000000u), 0, 0, 0);
  };
  // fragment offsets inside an operand tile: block b of a quadrant, lane's channel c = c0 + 16 b + ln, pixel row 4 k + lg: chunk image c >> 5,
  // slot (c & 31) >> 2 swizzled by the row, element c & 3.  The swizzle term ((row >> 1) & 7 = (2 k + (lg >> 1)) & 7) depends on k: the offset
  // is rebuilt per K step from the lane's slot and element (two integer instructions per fragment)
  int pim[HB], pslot[HB], qim[HB], qslot[HB];
#pragma unroll
  for (int b = 0; b < HB; ++b) {
    const int ci = i0 + 16 * b + ln, cj = j0 + 16 * b + ln;
    pim[b] = (ci >> 5) * BP * 32 + (ci & 3) + 32 * lg; pslot[b] = (ci & 31) >> 2;
    qim[b] = (cj >> 5) * BP * 32 + (cj & 3) + 32 * lg; qslot[b] = (cj & 31) >> 2;
  }
  f32x4 acc[HB][HB];
#pragma unroll
  for (int a = 0; a < HB; ++a)
#pragma unroll
    for (int b = 0; b < HB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // three stages: tile n is read while tile n + 1 is landing and tile n + 2 is requested (a 32-pixel tile computes for ~4 us: with two
  // stages the next tile's requests, issued during this one, would be waited for at its end)
  int t = blockIdx.x, cur = 0;
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const long long m0 = (long long)(t + st * (int)gridDim.x) * BP;
    const __amdgpu_buffer_rsrc_t rp = tile_rsrc(p.dn, p.ld_dn, m0), rq = tile_rsrc(p.x, p.ldx, m0);
#pragma unroll
    for (int q = 0; q < 2 * NBX; ++q) fetch_piece(rp, rq, st, q);
  }
  for (; t < p.tiles; t += gridDim.x, cur = cur == 2 ? 0 : cur + 1) {
    asm volatile("" : "+v"(pbase), "+v"(qbase));
    gdn_wait_barrier<NS>();   // this tile pair has landed (the next one's NS requests may be in flight); every wave is done with the stage requested next
    const long long mn = (long long)(t + 2 * (int)gridDim.x) * BP;
    const __amdgpu_buffer_rsrc_t rp = tile_rsrc(p.dn, p.ld_dn, mn), rq = tile_rsrc(p.x, p.ldx, mn);
    const int nxt = cur == 0 ? 2 : cur - 1;   // (cur + 2) mod 3
    const float* tp = sP + cur * TF;
    const float* tq = sQ + cur * TF;
    auto frags = [&](int k, float (&fa)[HB], float (&fb)[HB]) __attribute__((always_inline)) {
      const int sw = (2 * k + (lg >> 1)) & 7;
#pragma unroll
      for (int b = 0; b < HB; ++b) {
        fa[b] = tp[pim[b] + 128 * k + ((pslot[b] ^ sw) << 2)];
        fb[b] = tq[qim[b] + 128 * k + ((qslot[b] ^ sw) << 2)];
      }
    };
    float na[HB], nb[HB];
    frags(0, na, nb);
#pragma unroll
    for (int k = 0; k < BP / 4; ++k) {
      float fa[HB], fb[HB];
#pragma unroll
      for (int b = 0; b < HB; ++b) { fa[b] = na[b]; fb[b] = nb[b] * nb[b]; }
      if (k + 1 < BP / 4) frags(k + 1, na, nb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < HB; ++a)
#pragma unroll
        for (int b = 0; b < HB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      // the tile pair after next: 2 NBX requests over the BP / 4 = 8 K steps
#pragma unroll
      for (int q = k * (2 * NBX) / (BP / 4); q < (k + 1) * (2 * NBX) / (BP / 4); ++q) fetch_piece(rp, rq, nxt, q);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // acc[a][b][r] is (channel i0 + 16 a + 4 lg + r, channel j0 + 16 b + ln)
  float* out = p.part + (size_t)blockIdx.x * CH * CH;
#pragma unroll
  for (int a = 0; a < HB; ++a)
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)(i0 + 16 * a + 4 * lg + r) * CH + j0 + 16 * b + ln] = acc[a][b][r];
}

// dgamma_p[i][j] += chain(sum over the workgroups' partials), as gdn_reparam_bwd_kernel does for the gamma half.  A workgroup owns 64 consecutive
// results; its four waves take every fourth partial (four loads in flight each) and meet in LDS, added in a fixed order
__global__ __launch_bounds__(256) void gdn_dgamma_finish_kernel(const float* part, int parts, int CH, const float* gamma_p, int C, float bound_g,
                                                                float* dgamma_p) {
  __shared__ float red[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const bool ok = e < C * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (ok) {
    const int i = e / C, j = e - i * C;
    const float* src = part + (size_t)i * CH + j;
    const size_t st = (size_t)CH * CH;
    int q = w;
    for (; q + 12 < parts; q += 16) {
      s0 += src[(size_t)q * st]; s1 += src[(size_t)(q + 4) * st]; s2 += src[(size_t)(q + 8) * st]; s3 += src[(size_t)(q + 12) * st];
    }
    for (; q < parts; q += 4) s0 += src[(size_t)q * st];
  }
  red[w][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && ok) {
    const float sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const float v = gamma_p[e], g = sum * 2.0f * fmaxf(v, bound_g);
    dgamma_p[e] += (v >= bound_g || g < 0.f) ? g : 0.f;
  }
}

// workgroups of the gamma-gradient kernel: every one leaves a CH x CH partial that the finish kernel reads back, so small problems use fewer
// (at least four 32-pixel tiles each)
static int gdn_dgamma_grid(int tiles) { return std::max(1, std::min((tiles + 3) / 4, 256)); }

template <int NS>
static void gdn_dgamma_launch(const GdnDgArgs& a, hipStream_t s) {
  const size_t lds = (size_t)6 * (NS / 2) * kGdnDgBP * 32 * sizeof(float);
  static std::atomic<bool> done{false};
  if (!done.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gdn_dgamma_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(gdn_dgamma_kernel<NS>, dim3(gdn_dgamma_grid(a.tiles)), dim3(256), lds, s, a);
}

// the beta gradient: the first pass's per-workgroup column sums [parts][CP], added in a fixed order (32 channels per workgroup; 8 row
// groups, then the groups)
__global__ __launch_bounds__(256) void gdn_colpart_sum_kernel(const float* colpart, int parts, int CP, int C, float* dbeta_eff) {
  __shared__ float part[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
  float sum = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int q = r; q < parts; q += 8) sum += colpart[(size_t)q * CP + c];
  }
  part[r][threadIdx.x & 31] = sum;
  __syncthreads();
  if (r == 0 && c < C) {
    float v = part[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += part[k][threadIdx.x];
    dbeta_eff[c] = v;
  }
}

static inline int grid1(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(cdiv64(n, 256), 1), 8192); }

// dbeta_p[c] += chain(sum over the workgroups' column sums), as gdn_reparam_bwd_kernel does for the beta half
__global__ __launch_bounds__(256) void gdn_dbeta_finish_kernel(const float* colpart, int parts, int CP, int C, const float* beta_p, float bound_b,
                                                               float* dbeta_p) {
  __shared__ float part[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
  float sum = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int q = r; q < parts; q += 8) sum += colpart[(size_t)q * CP + c];
  }
  part[r][threadIdx.x & 31] = sum;
  __syncthreads();
  if (r == 0 && c < C) {
    float v = part[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += part[k][threadIdx.x];
    const float b = beta_p[c], gb = v * 2.0f * fmaxf(b, bound_b);
    dbeta_p[c] += (b >= bound_b || gb < 0.f) ? gb : 0.f;
  }
}

struct GdnLayout {
  int CP;
  size_t beta_eff, pack_f, pack_b, x2, norm, dn, u, w, dg, db, colpart, gpart, conv_ws, end;
  size_t conv_ws_bytes;
};

static int gdn_conv_desc(const crdr_gdn_desc* d, crdr_conv_desc* cd, int CP) {
  memset(cd, 0, sizeof(*cd));
  CRDR_REQUIRE(d->M > 0 && d->M < (1ll << 31), "gdn: pixel count out of range");
  cd->N = (int32_t)d->M; cd->H = 1; cd->W = 1; cd->C = d->C; cd->OH = 1; cd->OW = 1; cd->OC = d->C;
  cd->kh = 1; cd->kw = 1; cd->stride = 1; cd->pad = 0; cd->transposed = 0;
  cd->ldx = d->C; cd->ldy = d->C; cd->wrows = CP; cd->wcols = CP;
  cd->flags = CRDR_CONV_NOSPLIT;   // the scratch of these launches is shared with other kernels: no ticket area
  return 0;
}

static int gdn_layout(const crdr_gdn_desc* d, int backward, GdnLayout* L) {
  CRDR_REQUIRE(d->C > 0 && d->C % 4 == 0 && d->ldx % 4 == 0 && d->ldx >= d->C, "gdn: C (%d) and ldx (%d) must be multiples of 4", d->C, d->ldx);
  const int CP = round_up(d->C, 32);
  L->CP = CP;
  size_t off = 0;
  auto take = [&](size_t floats) { const size_t o = off; off += (floats * 4 + 255) / 256 * 256; return o; };
  const size_t MC = (size_t)d->M * d->C;
  L->beta_eff = take(CP); L->pack_f = take((size_t)CP * CP); L->pack_b = take((size_t)CP * CP);
  L->x2 = take(MC); L->norm = take(MC);
  L->dn = L->u = L->w = L->dg = L->db = L->colpart = L->gpart = 0;
  if (backward) {
    L->dn = take(MC); L->u = take(MC); L->w = take(MC); L->dg = take((size_t)d->C * d->C); L->db = take(d->C);
    L->colpart = take((size_t)256 * CP);
    if (CP <= 192) {   // the persistent gamma-gradient kernel's per-workgroup partial sums
      const int ch = CP <= 64 ? 64 : (CP <= 128 ? 128 : 192);
      L->gpart = take((size_t)std::min<long long>((d->M + kGdnDgBP - 1) / kGdnDgBP, 256) * ch * ch);
    }
  }
  crdr_conv_desc cd;
  if (int rc = gdn_conv_desc(d, &cd, CP)) return rc;
  size_t cw = crdr_conv2d_workspace(&cd);
  if (backward) {
    crdr_wgrad_desc wd;
    memset(&wd, 0, sizeof(wd));
    wd.N = (int32_t)d->M; wd.PH = 1; wd.PW = 1; wd.PC = d->C; wd.ldp = d->C; wd.QH = 1; wd.QW = 1; wd.QC = d->C; wd.ldq = d->C;
    wd.kh = 1; wd.kw = 1; wd.stride = 1; wd.pad = 0; wd.gI = d->C; wd.gJ = d->C;
    cw = std::max(cw, crdr_conv2d_wgrad_workspace(&wd));
    wd.algo = CRDR_WGRAD_SQUARE_Q; wd.ldq = d->ldx;   // the fused backward's launch
    cw = std::max(cw, crdr_conv2d_wgrad_workspace(&wd));
    cw = std::max(cw, crdr_colsum_workspace(d->M, d->C));
  }
  L->conv_ws_bytes = cw;
  L->conv_ws = take((cw + 3) / 4);
  L->end = off;
  return 0;
}

// CRDR_GDN_UNFUSED_BWD=1: the nine-launch backward (the form the fused one is A/B-tested against; read at every call)
static bool gdn_unfused_backward() {
  const char* e = getenv("CRDR_GDN_UNFUSED_BWD");
  return e && e[0] == '1';
}

static bool gdn_fused_ok(const crdr_gdn_desc* d, const GdnLayout& L) {
  return L.CP <= 192 && (long long)128 * d->ldx * 4 < (1ll << 31) && (long long)128 * d->ldy * 4 < (1ll << 31);
}

// y (mode 0) or n (mode 1) by the fused kernel; gamma_eff / beta_eff must already be in the workspace
static int gdn_fused(const crdr_gdn_desc* d, const GdnLayout& L, char* ws, const float* x, float* out, int ldo, int mode, crdr_stream_t s) {
  GdnFusedArgs a;
  a.x = x; a.pack = (const float*)(ws + L.pack_f); a.beta = (const float*)(ws + L.beta_eff); a.y = out;
  a.M = d->M; a.C = d->C; a.ldx = d->ldx; a.ldy = ldo; a.inverse = d->inverse; a.mode = mode;
  a.tiles = (int)((d->M + kGdnBM - 1) / kGdnBM);
  a.CP = L.CP;
  a.x_bytes = 0; a.pack_bytes = (unsigned)((size_t)L.CP * L.CP * 4);
  if (d->C <= 64) gdn_fused_launch<4>(a, as_stream(s));
  else if (d->C <= 128) gdn_fused_launch<8>(a, as_stream(s));
  else gdn_fused_launch<12>(a, as_stream(s));
  CRDR_CHECK_LAUNCH("gdn_fused_fwd");
  return 0;
}

static int gdn_norm(const crdr_gdn_desc* d, const GdnLayout& L, char* ws, const float* x, const float* beta, const float* gamma,
                    crdr_stream_t s) {
  const float ped = d->reparam_offset * d->reparam_offset;
  const float bb = sqrtf(d->beta_min + ped), bg = d->reparam_offset;
  float* beta_eff = (float*)(ws + L.beta_eff);
  hipLaunchKernelGGL(gdn_reparam_kernel, dim3(grid1((int64_t)L.CP * L.CP)), dim3(256), 0, as_stream(s), beta, gamma, d->C, L.CP, bb, bg,
                     ped, beta_eff, (float*)(ws + L.pack_f), (float*)(ws + L.pack_b));
  CRDR_CHECK_LAUNCH("gdn_reparam");
  if (gdn_fused_ok(d, L)) return gdn_fused(d, L, ws, x, (float*)(ws + L.norm), d->C, 1, s);   // n in one pass over x
  hipLaunchKernelGGL(gdn_square_kernel, dim3(grid1(d->M * (d->C / 4))), dim3(256), 0, as_stream(s), x, d->ldx, d->M, d->C / 4,
                     (float*)(ws + L.x2));
  CRDR_CHECK_LAUNCH("gdn_square");
  crdr_conv_desc cd;
  if (int rc = gdn_conv_desc(d, &cd, L.CP)) return rc;
  cd.flags |= CRDR_EPI_BIAS;
  crdr_conv_io io;
  memset(&io, 0, sizeof(io));
  io.x = (const float*)(ws + L.x2); io.w = (const float*)(ws + L.pack_f); io.y = (float*)(ws + L.norm); io.bias = beta_eff;
  return crdr_conv2d(&cd, &io, ws + L.conv_ws, L.conv_ws_bytes, s);
}

}  // namespace crdr

using namespace crdr;

extern "C" size_t crdr_gdn_workspace(const crdr_gdn_desc* d, int backward) {
  GdnLayout L;
  if (!d || gdn_layout(d, backward, &L)) return 0;
  return L.end;
}

extern "C" int crdr_gdn_fwd(const crdr_gdn_desc* d, const float* x, const float* beta, const float* gamma, float* y, void* ws,
                            size_t ws_bytes, crdr_stream_t s) {
  CRDR_REQUIRE(d && x && beta && gamma && y && ws, "gdn_fwd: null pointer");
  GdnLayout L;
  if (int rc = gdn_layout(d, 0, &L)) return rc;
  CRDR_REQUIRE(ws_bytes >= L.end && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "gdn_fwd: workspace too small or misaligned (%zu < %zu)", ws_bytes, L.end);
  CRDR_REQUIRE(d->ldy % 4 == 0 && d->ldy >= d->C, "gdn_fwd: ldy");
  char* w8 = (char*)ws;
  if (gdn_fused_ok(d, L)) {   // reparametrisation (C x C, tiny) + ONE pass over x
    const float ped = d->reparam_offset * d->reparam_offset;
    hipLaunchKernelGGL(gdn_reparam_kernel, dim3(grid1((int64_t)L.CP * L.CP)), dim3(256), 0, as_stream(s), beta, gamma, d->C, L.CP,
                       sqrtf(d->beta_min + ped), d->reparam_offset, ped, (float*)(w8 + L.beta_eff), (float*)(w8 + L.pack_f), (float*)(w8 + L.pack_b));
    CRDR_CHECK_LAUNCH("gdn_reparam");
    return gdn_fused(d, L, w8, x, y, d->ldy, 0, s);
  }
  if (int rc = gdn_norm(d, L, w8, x, beta, gamma, s)) return rc;
  hipLaunchKernelGGL(gdn_apply_kernel, dim3(grid1(d->M * (d->C / 4))), dim3(256), 0, as_stream(s), x, d->ldx, (const float*)(w8 + L.norm),
                     d->M, d->C / 4, d->inverse, y, d->ldy);
  CRDR_CHECK_LAUNCH("gdn_apply");
  return 0;
}

extern "C" int crdr_gdn_bwd(const crdr_gdn_desc* d, const float* x, const float* beta, const float* gamma, const float* dy, int lddy,
                            float* dx, int lddx, float* dbeta, float* dgamma, void* ws, size_t ws_bytes, crdr_stream_t s) {
  CRDR_REQUIRE(d && x && beta && gamma && dy && dx && dbeta && dgamma && ws, "gdn_bwd: null pointer");
  GdnLayout L;
  if (int rc = gdn_layout(d, 1, &L)) return rc;
  CRDR_REQUIRE(ws_bytes >= L.end && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "gdn_bwd: workspace too small or misaligned (%zu < %zu)", ws_bytes, L.end);
  CRDR_REQUIRE(lddy % 4 == 0 && lddx % 4 == 0, "gdn_bwd: strides must be multiples of 4");
  char* w8 = (char*)ws;
  crdr_wgrad_desc wd;
  memset(&wd, 0, sizeof(wd));
  wd.N = (int32_t)d->M; wd.PH = 1; wd.PW = 1; wd.PC = d->C; wd.ldp = d->C; wd.QH = 1; wd.QW = 1; wd.QC = d->C; wd.ldq = d->C;
  wd.kh = 1; wd.kw = 1; wd.stride = 1; wd.pad = 0; wd.gI = d->C; wd.gJ = d->C; wd.accumulate = 0;
  const float ped = d->reparam_offset * d->reparam_offset;
  if (gdn_fused_ok(d, L) && (long long)128 * lddy * 4 < (1ll << 31) && (long long)128 * lddx * 4 < (1ll << 31) && !gdn_unfused_backward()) {
    // reparametrisation, ONE pass over (x, dy) -> (dx, dn, column sums), the gamma gradient straight from (dn, x) with the square taken in the
    // kernel, the chain rule through the parametrisation
    hipLaunchKernelGGL(gdn_reparam_kernel, dim3(grid1((int64_t)L.CP * L.CP)), dim3(256), 0, as_stream(s), beta, gamma, d->C, L.CP,
                       sqrtf(d->beta_min + ped), d->reparam_offset, ped, (float*)(w8 + L.beta_eff), (float*)(w8 + L.pack_f), (float*)(w8 + L.pack_b));
    CRDR_CHECK_LAUNCH("gdn_reparam");
    float *dn = (float*)(w8 + L.dn), *dg = (float*)(w8 + L.dg), *parts = (float*)(w8 + L.colpart);
    GdnBwd1Args b;
    memset(&b, 0, sizeof(b));
    b.x = x; b.dy = dy; b.pack_f = (const float*)(w8 + L.pack_f); b.pack_b = (const float*)(w8 + L.pack_b); b.beta = (const float*)(w8 + L.beta_eff);
    b.dn = dn; b.dx = dx; b.colpart = parts; b.M = d->M; b.C = d->C; b.CP = L.CP; b.ldx = d->ldx; b.lddy = lddy; b.ld_dn = d->C; b.lddx = lddx;
    b.inverse = d->inverse; b.tiles = (int)((d->M + kGdnBM - 1) / kGdnBM); b.pack_bytes = (unsigned)((size_t)L.CP * L.CP * 4);
    if (d->C <= 64) gdn_bwd1_launch<4>(b, as_stream(s));
    else if (d->C <= 128) gdn_bwd1_launch<8>(b, as_stream(s));
    else gdn_bwd1_launch<12>(b, as_stream(s));
    CRDR_CHECK_LAUNCH("gdn_bwd_onepass");
    const char* via = getenv("CRDR_GDN_DGAMMA");
    if (via && via[0] == 'w') {   // CRDR_GDN_DGAMMA=wgrad: the gamma gradient by the tiled weight-gradient kernel (the form the persistent one is measured against)
      wd.ldq = d->ldx; wd.algo = CRDR_WGRAD_SQUARE_Q;
      if (int rc = crdr_conv2d_wgrad(&wd, dn, x, dg, w8 + L.conv_ws, L.conv_ws_bytes, s)) return rc;
      float* db = (float*)(w8 + L.db);
      hipLaunchKernelGGL(gdn_colpart_sum_kernel, dim3((d->C + 31) / 32), dim3(256), 0, as_stream(s), (const float*)parts, std::min(b.tiles, 256), L.CP,
                         d->C, db);
      CRDR_CHECK_LAUNCH("gdn_colpart_sum");
      hipLaunchKernelGGL(gdn_reparam_bwd_kernel, dim3(grid1((int64_t)d->C * d->C)), dim3(256), 0, as_stream(s), (const float*)dg,
                         (const float*)db, gamma, beta, d->C, sqrtf(d->beta_min + ped), d->reparam_offset, dgamma, dbeta);
      CRDR_CHECK_LAUNCH("gdn_reparam_bwd");
      return 0;
    }
    GdnDgArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.dn = dn; ga.x = x; ga.part = (float*)(w8 + L.gpart); ga.M = d->M; ga.C = d->C; ga.ld_dn = d->C; ga.ldx = d->ldx;
    ga.tiles = (int)((d->M + kGdnDgBP - 1) / kGdnDgBP);
    const int ch = d->C <= 64 ? 64 : (d->C <= 128 ? 128 : 192);
    if (d->C <= 64) gdn_dgamma_launch<4>(ga, as_stream(s));
    else if (d->C <= 128) gdn_dgamma_launch<8>(ga, as_stream(s));
    else gdn_dgamma_launch<12>(ga, as_stream(s));
    CRDR_CHECK_LAUNCH("gdn_dgamma");
    hipLaunchKernelGGL(gdn_dgamma_finish_kernel, dim3((d->C * d->C + 63) / 64), dim3(256), 0, as_stream(s), (const float*)ga.part,
                       gdn_dgamma_grid(ga.tiles), ch, gamma, d->C, d->reparam_offset, dgamma);
    CRDR_CHECK_LAUNCH("gdn_dgamma_finish");
    hipLaunchKernelGGL(gdn_dbeta_finish_kernel, dim3((d->C + 31) / 32), dim3(256), 0, as_stream(s), (const float*)parts, std::min(b.tiles, 256), L.CP,
                       d->C, beta, sqrtf(d->beta_min + ped), dbeta);
    CRDR_CHECK_LAUNCH("gdn_dbeta_finish");
    return 0;
  }
  if (int rc = gdn_norm(d, L, w8, x, beta, gamma, s)) return rc;  // recomputed: cheaper than keeping M x C floats alive
  const int g = grid1(d->M * (d->C / 4));
  float *dn = (float*)(w8 + L.dn), *u = (float*)(w8 + L.u), *w = (float*)(w8 + L.w);
  hipLaunchKernelGGL(gdn_bwd_prep_kernel, dim3(g), dim3(256), 0, as_stream(s), x, d->ldx, (const float*)(w8 + L.norm), dy, lddy, d->M,
                     d->C / 4, d->inverse, dn, u);
  CRDR_CHECK_LAUNCH("gdn_bwd_prep");
  crdr_conv_desc cd;
  if (int rc = gdn_conv_desc(d, &cd, L.CP)) return rc;
  crdr_conv_io io;
  memset(&io, 0, sizeof(io));
  io.x = dn; io.w = (const float*)(w8 + L.pack_b); io.y = w;
  if (int rc = crdr_conv2d(&cd, &io, w8 + L.conv_ws, L.conv_ws_bytes, s)) return rc;  // w[p][j] = sum_i gamma_ij dn[p][i]
  hipLaunchKernelGGL(gdn_bwd_finish_kernel, dim3(g), dim3(256), 0, as_stream(s), x, d->ldx, (const float*)u, (const float*)w, d->M,
                     d->C / 4, dx, lddx);
  CRDR_CHECK_LAUNCH("gdn_bwd_finish");
  float *dg = (float*)(w8 + L.dg), *db = (float*)(w8 + L.db);
  if (gdn_fused_ok(d, L)) {   // the fused norm pass squares on the fly: the weight gradient still wants x^2 in memory
    hipLaunchKernelGGL(gdn_square_kernel, dim3(grid1(d->M * (d->C / 4))), dim3(256), 0, as_stream(s), x, d->ldx, d->M, d->C / 4,
                       (float*)(w8 + L.x2));
    CRDR_CHECK_LAUNCH("gdn_square");
  }
  if (int rc = crdr_conv2d_wgrad(&wd, dn, (const float*)(w8 + L.x2), dg, w8 + L.conv_ws, L.conv_ws_bytes, s)) return rc;
  if (int rc = crdr_colsum(dn, d->C, d->M, d->C, db, 0, w8 + L.conv_ws, L.conv_ws_bytes, s)) return rc;
  hipLaunchKernelGGL(gdn_reparam_bwd_kernel, dim3(grid1((int64_t)d->C * d->C)), dim3(256), 0, as_stream(s), (const float*)dg,
                     (const float*)db, gamma, beta, d->C, sqrtf(d->beta_min + ped), d->reparam_offset, dgamma, dbeta);
  CRDR_CHECK_LAUNCH("gdn_reparam_bwd");
  return 0;
}
