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

static int gdn_norm(const crdr_gdn_desc* d, const GdnLayout& L, char* ws, const float* x, const float* beta, const float* gamma,
                    crdr_stream_t s) {
  const float ped = d->reparam_offset * d->reparam_offset;
  const float bb = sqrtf(d->beta_min + ped), bg = d->reparam_offset;
  float* beta_eff = (float*)(ws + L.beta_eff);
  hipLaunchKernelGGL(gdn_reparam_kernel, dim3(grid1((int64_t)L.CP * L.CP)), dim3(256), 0, as_stream(s), beta, gamma, d->C, L.CP, bb, bg,
                     ped, beta_eff, (float*)(ws + L.pack_f), (float*)(ws + L.pack_b));
  CRDR_CHECK_LAUNCH("gdn_reparam");
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
  if (int rc = crdr_conv2d_wgrad(&wd, dn, (const float*)(w8 + L.x2), dg, w8 + L.conv_ws, L.conv_ws_bytes, s)) return rc;
  if (int rc = crdr_colsum(dn, d->C, d->M, d->C, db, 0, w8 + L.conv_ws, L.conv_ws_bytes, s)) return rc;
  const float ped = d->reparam_offset * d->reparam_offset;
  hipLaunchKernelGGL(gdn_reparam_bwd_kernel, dim3(grid1((int64_t)d->C * d->C)), dim3(256), 0, as_stream(s), (const float*)dg,
                     (const float*)db, gamma, beta, d->C, sqrtf(d->beta_min + ped), d->reparam_offset, dgamma, dbeta);
  CRDR_CHECK_LAUNCH("gdn_reparam_bwd");
  return 0;
}
