// Entropy-model arithmetic of the CRDR training step (fp32, HBM/latency bound, tiny tensors at 256^2 crops).
//
// GaussianConditional (mean + scale, CompressAI 1.2.4 semantics as subclassed by the reference at
// src/models/subnet/entropy_model/{gaussian_conditional,ste_gaussian_conditional}.py) and the factorised
// prior EntropyBottleneck (src/models/subnet/entropy_model/entropy_bottleneck.py).  One launch produces what
// the reference obtains from ~20 ATen kernels x 2 calls: STE-rounded latent, noisy and quantised likelihoods
// and their per-image bit sums.

#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace crdr {

__device__ __forceinline__ float std_cdf(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }
__device__ __forceinline__ float std_pdf(float x) { return 0.39894228040143267794f * expf(-0.5f * x * x); }

__device__ __forceinline__ float block_sum(float v, float* red) {  // any block size multiple of 64, <= 1024
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}

struct GcArgs {
  crdr_gc_desc d;
  const float *y, *mu, *sigma, *noise;
  float *yhat, *lik_noisy, *lik_quant, *bits_noisy, *bits_quant;
  const float *gbits, *dyhat;
  int lddyhat;
  float *dy, *dmu, *dsigma;
};

// grid (B, N): block b of image n strides over that image's HW*C elements
__global__ __launch_bounds__(1024) void gauss_cond_fwd_kernel(const GcArgs p) {
  __shared__ float red[16];
  const int n = blockIdx.y, C = p.d.C;
  const int per_img = p.d.HW * C;
  const float inv_ln2 = 1.4426950408889634f;
  float sn = 0.f, sq = 0.f;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < per_img; e += gridDim.x * blockDim.x) {
    const int px = e / C, c = e - px * C;
    const size_t pix = (size_t)n * p.d.HW + px;
    const float yv = p.y[pix * p.d.ldy + c], m = p.mu[pix * p.d.ldmu + c];
    const float sg = fmaxf(p.sigma[pix * p.d.ldsigma + c], p.d.scale_bound);
    const float q = rintf(yv - m);  // torch.round: half to even
    if (p.yhat) p.yhat[pix * p.d.ldyhat + c] = q + m;
    {
      const float a = fabsf(q);  // |round(y - mu) + mu - mu|
      float l = std_cdf((0.5f - a) / sg) - std_cdf((-0.5f - a) / sg);
      l = fmaxf(l, p.d.likelihood_bound);
      if (p.lik_quant) p.lik_quant[pix * C + c] = l;
      sq -= logf(l) * inv_ln2;
    }
    if (p.noise) {
      const float a = fabsf(yv + p.noise[pix * C + c] - m);
      float l = std_cdf((0.5f - a) / sg) - std_cdf((-0.5f - a) / sg);
      l = fmaxf(l, p.d.likelihood_bound);
      if (p.lik_noisy) p.lik_noisy[pix * C + c] = l;
      sn -= logf(l) * inv_ln2;
    }
  }
  sn = block_sum(sn, red);
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) {  // the training shapes: deterministic
      if (p.bits_noisy && p.noise) p.bits_noisy[n] += sn;
      if (p.bits_quant) p.bits_quant[n] += sq;
    } else {
      if (p.bits_noisy && p.noise) atomicAdd(p.bits_noisy + n, sn);
      if (p.bits_quant) atomicAdd(p.bits_quant + n, sq);
    }
  }
}

__global__ __launch_bounds__(256) void gauss_cond_bwd_kernel(const GcArgs p) {
  const int C = p.d.C;
  const int per_img = p.d.HW * C;
  const int64_t total = (int64_t)p.d.N * per_img;
  const float inv_ln2 = 1.4426950408889634f;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int n = (int)(e / per_img);
    const size_t pix = (size_t)(e / C);
    const int c = (int)(e - (int64_t)pix * C);
    const float yv = p.y[pix * p.d.ldy + c], m = p.mu[pix * p.d.ldmu + c];
    const float sraw = p.sigma[pix * p.d.ldsigma + c];
    const float sg = fmaxf(sraw, p.d.scale_bound);
    const float dlt = yv + p.noise[pix * C + c] - m;
    const float a = fabsf(dlt), sgn = dlt > 0.f ? 1.f : (dlt < 0.f ? -1.f : 0.f);
    const float zu = (0.5f - a) / sg, zl = (-0.5f - a) / sg;
    const float lraw = std_cdf(zu) - std_cdf(zl);
    const float l = fmaxf(lraw, p.d.likelihood_bound);
    const float glik = p.gbits[n] * (-inv_ln2 / l);                      // d(-log2 l)/dl scaled
    const float graw = (lraw >= p.d.likelihood_bound || glik < 0.f) ? glik : 0.f;  // LowerBound backward
    const float pu = std_pdf(zu), pl = std_pdf(zl);
    const float dl_da = (pl - pu) / sg;
    const float dl_dsg = (zl * pl - zu * pu) / sg;
    const float gy = graw * dl_da * sgn;
    const float gsg = graw * dl_dsg;
    const float gsig = (sraw >= p.d.scale_bound || gsg < 0.f) ? gsg : 0.f;
    const float ste = p.dyhat ? p.dyhat[pix * p.lddyhat + c] : 0.f;
    p.dy[pix * C + c] = gy + ste;   // yhat = ste_round(y - mu) + mu: d/dy = 1, d/dmu = 0
    p.dmu[pix * C + c] = -gy;
    p.dsigma[pix * C + c] = gsig;
  }
}

// ------------------------------------------------------------------------------------------------------------
// factorised prior
// ------------------------------------------------------------------------------------------------------------
struct EbParams {  // one channel, transformed
  float spM[4][9];  // layer 0 uses [0][0..2], layers 1..3 use 3x3 [out][in]
  float sgM[4][9];  // softplus' = sigmoid(raw M)
  float b[4][3];
  float tf[4][3];
  float spM4[3], sgM4[3], b4;
};
__device__ __forceinline__ float softplus_(float w) { return w > 20.f ? w : log1pf(expf(w)); }
__device__ __forceinline__ float sigmoid_(float w) { return 1.f / (1.f + expf(-w)); }

__device__ void eb_load(const float* raw, EbParams& P) {
  // raw layout (CRDR_EB_PARAMS = 58): L0 {M[3] b[3] f[3]} L1..L3 {M[9] b[3] f[3]} L4 {M[3] b[1]}
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    P.spM[0][i] = softplus_(raw[i]); P.sgM[0][i] = sigmoid_(raw[i]);
    P.b[0][i] = raw[3 + i]; P.tf[0][i] = tanhf(raw[6 + i]);
  }
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const float* r = raw + 9 + (k - 1) * 15;
#pragma unroll
    for (int i = 0; i < 9; ++i) { P.spM[k][i] = softplus_(r[i]); P.sgM[k][i] = sigmoid_(r[i]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) { P.b[k][i] = r[9 + i]; P.tf[k][i] = tanhf(r[12 + i]); }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { P.spM4[i] = softplus_(raw[54 + i]); P.sgM4[i] = sigmoid_(raw[54 + i]); }
  P.b4 = raw[57];
}

struct EbTape { float a[4][3]; float th[4][3]; };  // activations after each gated layer, tanh(h)

__device__ __forceinline__ float eb_logits(const EbParams& P, float x, EbTape& T) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float h = P.spM[0][i] * x + P.b[0][i];
    T.th[0][i] = tanhf(h);
    T.a[0][i] = h + P.tf[0][i] * T.th[0][i];
  }
#pragma unroll
  for (int k = 1; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float h = P.b[k][i];
#pragma unroll
      for (int j = 0; j < 3; ++j) h += P.spM[k][i * 3 + j] * T.a[k - 1][j];
      T.th[k][i] = tanhf(h);
      T.a[k][i] = h + P.tf[k][i] * T.th[k][i];
    }
  float o = P.b4;
#pragma unroll
  for (int j = 0; j < 3; ++j) o += P.spM4[j] * T.a[3][j];
  return o;
}

// accumulate d(out)/d(raw params) * go into g[58]; returns d(out)/dx * go
__device__ __forceinline__ float eb_logits_bwd(const EbParams& P, float x, const EbTape& T, float go, float* g) {
  float da[3];
  g[57] += go;
#pragma unroll
  for (int j = 0; j < 3; ++j) { g[54 + j] += go * T.a[3][j] * P.sgM4[j]; da[j] = go * P.spM4[j]; }
#pragma unroll
  for (int k = 3; k >= 1; --k) {
    float* gk = g + 9 + (k - 1) * 15;
    float dprev[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float th = T.th[k][i];
      const float dh = da[i] * (1.f + P.tf[k][i] * (1.f - th * th));
      gk[12 + i] += da[i] * th * (1.f - P.tf[k][i] * P.tf[k][i]);
      gk[9 + i] += dh;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        gk[i * 3 + j] += dh * T.a[k - 1][j] * P.sgM[k][i * 3 + j];
        dprev[j] += dh * P.spM[k][i * 3 + j];
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) da[j] = dprev[j];
  }
  float dx = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float th = T.th[0][i];
    const float dh = da[i] * (1.f + P.tf[0][i] * (1.f - th * th));
    g[6 + i] += da[i] * th * (1.f - P.tf[0][i] * P.tf[0][i]);
    g[3 + i] += dh;
    g[i] += dh * x * P.sgM[0][i];
    dx += dh * P.spM[0][i];
  }
  return dx;
}

// block per channel
__global__ __launch_bounds__(256) void eb_fwd_kernel(const float* z, const float* noise, const float* params,
                                                     const float* medians, int N, int HW, int C, float bound,
                                                     float* zhat, float* lik) {
  const int c = blockIdx.x;
  EbParams P;
  eb_load(params + (size_t)c * CRDR_EB_PARAMS, P);
  const float med = medians[c];
  EbTape T;
  for (int e = threadIdx.x; e < N * HW; e += blockDim.x) {
    const size_t idx = (size_t)e * C + c;
    const float zv = z[idx];
    const float q = rintf(zv - med) + med;
    if (zhat) zhat[idx] = q;
    const float v = noise ? zv + noise[idx] : q;
    const float lo = eb_logits(P, v - 0.5f, T), up = eb_logits(P, v + 0.5f, T);
    const float sm = lo + up;
    const float s = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
    const float l = fabsf(sigmoid_(s * up) - sigmoid_(s * lo));
    lik[idx] = fmaxf(l, bound);
  }
}

__global__ __launch_bounds__(256) void eb_bwd_kernel(const float* z, const float* noise, const float* params, int N,
                                                     int HW, int C, float bound, const float* gbits,
                                                     const float* dzhat, float* dz, float* dparams) {
  __shared__ float red[4][CRDR_EB_PARAMS];
  const int c = blockIdx.x;
  EbParams P;
  eb_load(params + (size_t)c * CRDR_EB_PARAMS, P);
  float g[CRDR_EB_PARAMS];
#pragma unroll
  for (int i = 0; i < CRDR_EB_PARAMS; ++i) g[i] = 0.f;
  const float inv_ln2 = 1.4426950408889634f;
  EbTape Tl, Tu;
  for (int e = threadIdx.x; e < N * HW; e += blockDim.x) {
    const size_t idx = (size_t)e * C + c;
    const int n = e / HW;
    const float v = z[idx] + noise[idx];
    const float lo = eb_logits(P, v - 0.5f, Tl), up = eb_logits(P, v + 0.5f, Tu);
    const float sm = lo + up;
    const float s = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
    const float A = sigmoid_(s * up), B = sigmoid_(s * lo);
    const float D = A - B, lraw = fabsf(D);
    const float l = fmaxf(lraw, bound);
    const float glik = gbits[n] * (-inv_ln2 / l);
    const float graw = (lraw >= bound || glik < 0.f) ? glik : 0.f;
    const float sd = D > 0.f ? 1.f : (D < 0.f ? -1.f : 0.f);
    const float gu = graw * sd * A * (1.f - A) * s;
    const float gl = -graw * sd * B * (1.f - B) * s;
    float dx = eb_logits_bwd(P, v + 0.5f, Tu, gu, g);
    dx += eb_logits_bwd(P, v - 0.5f, Tl, gl, g);
    dz[idx] = dx + (dzhat ? dzhat[idx] : 0.f);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < CRDR_EB_PARAMS; ++i) {
    float v = g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[w][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < CRDR_EB_PARAMS) {
    const int i = threadIdx.x;
    dparams[(size_t)c * CRDR_EB_PARAMS + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

// bits[n] += sum_e -log2(lik[n][e]) ; one block per image (deterministic)
__global__ __launch_bounds__(1024) void bits_kernel(const float* lik, int per_img, float* bits) {
  __shared__ float red[16];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int e = threadIdx.x; e < per_img; e += blockDim.x) s -= logf(lik[(size_t)n * per_img + e]);
  s = block_sum(s, red);
  if (threadIdx.x == 0) bits[n] += s * 1.4426950408889634f;
}

// single block: loss = sum |logits(q) - target|, dq = sign(.) * dlogits/dq
__global__ __launch_bounds__(256) void eb_quantile_kernel(const float* quantiles, const float* params,
                                                          const float* target, int C, float* loss, float* dq) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int e = threadIdx.x; e < C * 3; e += blockDim.x) {
    const int c = e / 3, j = e - c * 3;
    EbParams P;
    eb_load(params + (size_t)c * CRDR_EB_PARAMS, P);
    EbTape T;
    float g[CRDR_EB_PARAMS];
#pragma unroll
    for (int i = 0; i < CRDR_EB_PARAMS; ++i) g[i] = 0.f;
    const float x = quantiles[e];
    const float d = eb_logits(P, x, T) - target[j];
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    acc += fabsf(d);
    dq[e] = eb_logits_bwd(P, x, T, sgn, g);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = acc;
}

}  // namespace crdr

using namespace crdr;

extern "C" int crdr_eb_quantile_loss(const float* quantiles, const float* params, const float* target, int C,
                                     float* loss, float* dquantiles, crdr_stream_t s) {
  CRDR_REQUIRE(quantiles && params && target && loss && dquantiles, "eb_quantile_loss: null pointer");
  hipLaunchKernelGGL(eb_quantile_kernel, dim3(1), dim3(256), 0, as_stream(s), quantiles, params, target, C, loss,
                     dquantiles);
  CRDR_CHECK_LAUNCH("eb_quantile_loss");
  return 0;
}

extern "C" int crdr_gauss_cond_fwd(const crdr_gc_desc* d, const float* y, const float* mu, const float* sigma,
                                   const float* noise, float* yhat, float* lik_noisy, float* lik_quant,
                                   float* bits_noisy, float* bits_quant, crdr_stream_t s) {
  CRDR_REQUIRE(y && mu && sigma, "gauss_cond_fwd: null input");
  GcArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d; a.y = y; a.mu = mu; a.sigma = sigma; a.noise = noise;
  a.yhat = yhat; a.lik_noisy = lik_noisy; a.lik_quant = lik_quant; a.bits_noisy = bits_noisy; a.bits_quant = bits_quant;
  const int per_img = d->HW * d->C;
  if (per_img == 0 || d->N == 0) return 0;
  const int B = std::max(1, std::min(cdiv(per_img, 16384), 128));
  hipLaunchKernelGGL(gauss_cond_fwd_kernel, dim3(B, d->N), dim3(1024), 0, as_stream(s), a);
  CRDR_CHECK_LAUNCH("gauss_cond_fwd");
  return 0;
}

extern "C" int crdr_gauss_cond_bwd(const crdr_gc_desc* d, const float* y, const float* mu, const float* sigma,
                                   const float* noise, const float* gbits, const float* dyhat, int lddyhat, float* dy,
                                   float* dmu, float* dsigma, crdr_stream_t s) {
  CRDR_REQUIRE(y && mu && sigma && noise && gbits && dy && dmu && dsigma, "gauss_cond_bwd: null pointer");
  GcArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d; a.y = y; a.mu = mu; a.sigma = sigma; a.noise = noise;
  a.gbits = gbits; a.dyhat = dyhat; a.lddyhat = lddyhat; a.dy = dy; a.dmu = dmu; a.dsigma = dsigma;
  const int64_t total = (int64_t)d->N * d->HW * d->C;
  if (total == 0) return 0;
  const int nb = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(gauss_cond_bwd_kernel, dim3(nb), dim3(256), 0, as_stream(s), a);
  CRDR_CHECK_LAUNCH("gauss_cond_bwd");
  return 0;
}

extern "C" int crdr_entropy_bottleneck_fwd(const float* z, const float* noise, const float* params,
                                           const float* medians, int N, int HW, int C, float likelihood_bound,
                                           float* zhat, float* lik, float* bits, crdr_stream_t s) {
  CRDR_REQUIRE(z && params && medians && lik, "entropy_bottleneck_fwd: null pointer");
  if (N * HW * C == 0) return 0;
  hipLaunchKernelGGL(eb_fwd_kernel, dim3(C), dim3(256), 0, as_stream(s), z, noise, params, medians, N, HW, C,
                     likelihood_bound, zhat, lik);
  CRDR_CHECK_LAUNCH("eb_fwd");
  if (bits) {
    hipLaunchKernelGGL(bits_kernel, dim3(N), dim3(1024), 0, as_stream(s), (const float*)lik, HW * C, bits);
    CRDR_CHECK_LAUNCH("bits_kernel");
  }
  return 0;
}

extern "C" int crdr_entropy_bottleneck_bwd(const float* z, const float* noise, const float* params, int N, int HW,
                                           int C, float likelihood_bound, const float* gbits, const float* dzhat,
                                           float* dz, float* dparams, crdr_stream_t s) {
  CRDR_REQUIRE(z && noise && params && gbits && dz && dparams, "entropy_bottleneck_bwd: null pointer");
  if (N * HW * C == 0) return 0;
  hipLaunchKernelGGL(eb_bwd_kernel, dim3(C), dim3(256), 0, as_stream(s), z, noise, params, N, HW, C, likelihood_bound,
                     gbits, dzhat, dz, dparams);
  CRDR_CHECK_LAUNCH("eb_bwd");
  return 0;
}
