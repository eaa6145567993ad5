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
// The channel mix (a C x C GEMV per pixel) runs on the fp32 matrix cores as a 1x1 launch of the implicit-GEMM kernel on
// x^2; the rest are single-pass elementwise kernels (16 B per lane).  Not on the CRDR training path -- the ELIC
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
// moves 28 B).  A persistent workgroup (4 waves) takes tiles of 128 pixels x all channels: the x tile goes to LDS by
// LDS-DMA as NB chunk images [128 rows][32 floats] (the igemm A-tile image: 128-byte rows, XOR-swizzled 16-byte slots),
// gamma_eff streams through LDS in 32-channel K slabs [CP rows][32] (double buffered), A fragments are squared on the way
// from LDS to the matrix cores (exact fp32 MFMA), and the epilogue -- n = acc + beta, y = x rsqrt(n) or x sqrt(n) -- takes
// x from the LDS tile again, writes y over it in place and leaves with 16-byte stores of whole 128-byte row segments.
// Roofline: 2 C^2 FLOP against 8 C bytes per pixel = C / 4 FLOP per byte: at C = 192 the exact-fp32 matrix peak
// (157 TFLOP/s) caps the op at 3.3 TB/s = 0.41 of the HBM peak, at C = 128 it is HBM-bound.
// mode 0: y = GDN / IGDN(x); mode 1: y = n (the norm, for the backward pass).  CP = 32 NB <= 192.
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* gdn_lds_ptr_t;
struct GdnFusedArgs {
  const float* x;
  const float* pack;   // gamma_eff [CP][CP] (row = output channel, K-contiguous)
  const float* beta;   // beta_eff [C]
  float* y;
  long long M;
  int C, ldx, ldy, inverse, mode, tiles;
  unsigned x_bytes, pack_bytes;
};

template <int NB>
__global__ __launch_bounds__(256) void gdn_fused_fwd_kernel(const GdnFusedArgs p) {
  constexpr int CP = 32 * NB, BM = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;                    // [NB][BM * 32]
  float* sG = smem + NB * BM * 32;     // [2][CP * 32]
  float* sBeta = sG + 2 * CP * 32;     // [CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack), 0, p.pack_bytes, 0x00020000);
  for (int c = tid; c < CP; c += 256) sBeta[c] = c < p.C ? p.beta[c] : 1.f;
  // staging assignment (as igemm): thread fills slot (tid & 7) of rows (tid >> 3) + 32 j with source chunk slot ^ swizzle(row)
  const int srow = tid >> 3;
  const int csrc = (tid & 7) ^ ((srow >> 1) & 7);
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = lds_off(frow, kk * 2 + fh);
  auto fetch_g = [&](int kc, int buf) __attribute__((always_inline)) {   // gamma rows 0..CP-1, columns 32 kc .. +31
    float* b = sG + buf * CP * 32 + wave * 8 * 32;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (gdn_lds_ptr_t)(b + j * 32 * 32), 16,
                                               (int)(((unsigned)(srow + 32 * j) * CP + 32 * kc + csrc * 4) * 4u), 0, 0, 0);
  };
  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    const long long m0 = (long long)t * BM;
    const long long left = p.M - m0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + m0 * p.ldx, 0, (unsigned)std::min<long long>(((left < BM ? left : BM) - 1) * p.ldx * 4ll + p.C * 4ll, 0x7fffffffll), 0x00020000);
    __syncthreads();   // the previous tile's epilogue is done with sX / sG
    // x tile: chunk image kc, rows srow + 32 j (rows / channels past the tensor read zeros through the range check)
#pragma unroll
    for (int kc = 0; kc < NB; ++kc) {
      float* a = sX + kc * BM * 32 + wave * 8 * 32;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = srow + 32 * j, ch = 32 * kc + csrc * 4;
        const bool ok = (row < left) && (ch < p.C);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (gdn_lds_ptr_t)(a + j * 32 * 32), 16, (int)(ok ? ((unsigned)row * p.ldx + ch) * 4u : 0x80000000u),
                                                 0, 0, 0);
      }
    }
    fetch_g(0, 0);
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    __syncthreads();   // (waits for the DMA: x tile and slab 0 have landed)
#pragma unroll 1
    for (int kc = 0; kc < NB; ++kc) {
      if (kc + 1 < NB) fetch_g(kc + 1, (kc + 1) & 1);
      const float* fa = sX + kc * BM * 32 + wave * 32 * 32;
      const float* fb = sG + (kc & 1) * CP * 32;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        f32x4 a = *reinterpret_cast<const f32x4*>(fa + fo[kk]);
        a = a * a;
        f32x4 b[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) b[j] = *reinterpret_cast<const f32x4*>(fb + fo[kk] + j * 1024);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], b[j][s2], acc[j], 0, 0, 0);
      }
      __syncthreads();
    }
    // epilogue: accumulator element r of block j is (row (r & 3) + 8 (r >> 2) + 4 fh, column 32 j + frow) of this wave's 32 rows;
    // x sits in chunk image j at slot frow >> 2 (swizzled), element frow & 3 -- rewritten in place with y
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float bta = sBeta[32 * j + frow];
      float* img = sX + j * BM * 32 + wave * 32 * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
        float* q = img + lds_off(row, frow >> 2) + (frow & 3);
        const float n = acc[j][r] + bta;
        const float xv = *q;
        // (v_rsq_f32 / v_sqrt_f32: 1 ulp, one instruction each; the library sqrtf + division pair is ~25 and the epilogue of a wave that
        // owns its SIMD is not hidden behind anything.  n >= beta_min > 0: no denormal path)
        *q = p.mode ? n : (p.inverse ? xv * __builtin_amdgcn_sqrtf(n) : xv * __builtin_amdgcn_rsqf(n));
      }
    }
    // (the rows are private to the wave: program order is enough) 16-byte stores, 8 rows x 128 B per wave instruction
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + m0 * p.ldy, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float* img = sX + j * BM * 32 + wave * 32 * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = (lane >> 3) + 8 * k, slot = lane & 7;
        const int grow = wave * 32 + row, ch = 32 * j + ((slot ^ ((row >> 1) & 7)) << 2);
        const f32x4 v = *reinterpret_cast<const f32x4*>(img + row * 32 + slot * 4);
        const bool ok = (grow < left) && (ch < p.C);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ry, ok ? ((unsigned)grow * p.ldy + ch) * 4u : 0x80000000u, 0, 0);
      }
    }
  }
}

template <int NB>
static void gdn_fused_launch(const GdnFusedArgs& a, hipStream_t s) {
  constexpr int CP = 32 * NB;
  const size_t lds = ((size_t)NB * 128 * 32 + 2 * CP * 32 + CP) * sizeof(float);
  static std::atomic<bool> done{false};
  if (!done.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gdn_fused_fwd_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(gdn_fused_fwd_kernel<NB>, dim3(std::min(a.tiles, 256)), dim3(256), lds, s, a);
}

static inline int grid1(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(cdiv64(n, 256), 1), 8192); }

struct GdnLayout {
  int CP;
  size_t beta_eff, pack_f, pack_b, x2, norm, dn, u, w, dg, db, conv_ws, end;
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
  L->dn = L->u = L->w = L->dg = L->db = 0;
  if (backward) { L->dn = take(MC); L->u = take(MC); L->w = take(MC); L->dg = take((size_t)d->C * d->C); L->db = take(d->C); }
  crdr_conv_desc cd;
  if (int rc = gdn_conv_desc(d, &cd, CP)) return rc;
  size_t cw = crdr_conv2d_workspace(&cd);
  if (backward) {
    crdr_wgrad_desc wd;
    memset(&wd, 0, sizeof(wd));
    wd.N = (int32_t)d->M; wd.PH = 1; wd.PW = 1; wd.PC = d->C; wd.ldp = d->C; wd.QH = 1; wd.QW = 1; wd.QC = d->C; wd.ldq = d->C;
    wd.kh = 1; wd.kw = 1; wd.stride = 1; wd.pad = 0; wd.gI = d->C; wd.gJ = d->C;
    cw = std::max(cw, crdr_conv2d_wgrad_workspace(&wd));
    cw = std::max(cw, crdr_colsum_workspace(d->M, d->C));
  }
  L->conv_ws_bytes = cw;
  L->conv_ws = take((cw + 3) / 4);
  L->end = off;
  return 0;
}

static bool gdn_fused_ok(const crdr_gdn_desc* d, const GdnLayout& L) {
  return L.CP <= 192 && (long long)128 * d->ldx * 4 < (1ll << 31) && (long long)128 * d->ldy * 4 < (1ll << 31);
}

// y (mode 0) or n (mode 1) by the fused kernel; gamma_eff / beta_eff must already be in the workspace
static int gdn_fused(const crdr_gdn_desc* d, const GdnLayout& L, char* ws, const float* x, float* out, int ldo, int mode, crdr_stream_t s) {
  GdnFusedArgs a;
  a.x = x; a.pack = (const float*)(ws + L.pack_f); a.beta = (const float*)(ws + L.beta_eff); a.y = out;
  a.M = d->M; a.C = d->C; a.ldx = d->ldx; a.ldy = ldo; a.inverse = d->inverse; a.mode = mode;
  a.tiles = (int)((d->M + 127) / 128);
  a.x_bytes = 0; a.pack_bytes = (unsigned)((size_t)L.CP * L.CP * 4);
  switch (L.CP / 32) {
    case 1: gdn_fused_launch<1>(a, as_stream(s)); break;
    case 2: gdn_fused_launch<2>(a, as_stream(s)); break;
    case 3: gdn_fused_launch<3>(a, as_stream(s)); break;
    case 4: gdn_fused_launch<4>(a, as_stream(s)); break;
    case 5: gdn_fused_launch<5>(a, as_stream(s)); break;
    default: gdn_fused_launch<6>(a, as_stream(s)); break;
  }
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
  crdr_wgrad_desc wd;
  memset(&wd, 0, sizeof(wd));
  wd.N = (int32_t)d->M; wd.PH = 1; wd.PW = 1; wd.PC = d->C; wd.ldp = d->C; wd.QH = 1; wd.QW = 1; wd.QC = d->C; wd.ldq = d->C;
  wd.kh = 1; wd.kw = 1; wd.stride = 1; wd.pad = 0; wd.gI = d->C; wd.gJ = d->C; wd.accumulate = 0;
  float *dg = (float*)(w8 + L.dg), *db = (float*)(w8 + L.db);
  if (gdn_fused_ok(d, L)) {   // the fused norm pass squares on the fly: the weight gradient still wants x^2 in memory
    hipLaunchKernelGGL(gdn_square_kernel, dim3(grid1(d->M * (d->C / 4))), dim3(256), 0, as_stream(s), x, d->ldx, d->M, d->C / 4,
                       (float*)(w8 + L.x2));
    CRDR_CHECK_LAUNCH("gdn_square");
  }
  if (int rc = crdr_conv2d_wgrad(&wd, dn, (const float*)(w8 + L.x2), dg, w8 + L.conv_ws, L.conv_ws_bytes, s)) return rc;
  if (int rc = crdr_colsum(dn, d->C, d->M, d->C, db, 0, w8 + L.conv_ws, L.conv_ws_bytes, s)) return rc;
  const float ped = d->reparam_offset * d->reparam_offset;
  hipLaunchKernelGGL(gdn_reparam_bwd_kernel, dim3(grid1((int64_t)d->C * d->C)), dim3(256), 0, as_stream(s), (const float*)dg,
                     (const float*)db, gamma, beta, d->C, sqrtf(d->beta_min + ped), d->reparam_offset, dgamma, dbeta);
  CRDR_CHECK_LAUNCH("gdn_reparam_bwd");
  return 0;
}
