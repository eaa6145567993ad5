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

// ---- the interval likelihood of the training kernels ------------------------------------------------------------------------------
// Phi(zu) - Phi(zl), zu = (1/2 - a) / sigma, zl = (-1/2 - a) / sigma, a = |v| (compressai's sign trick keeps both arguments on the
// side where the tail is relatively accurate; GaussianConditional._likelihood as the reference subclasses it,
// src/models/subnet/entropy_model/ste_gaussian_conditional.py:20-27) = 1/2 [erfc(xl) - erfc(xh)], xl = (a - 1/2) rs, xh = (a + 1/2) rs,
// rs = 1 / (sqrt 2 sigma).  Two library erfcf calls per likelihood are ~4 divergent branches each on this target; here
//   erfc(x) = exp(-x^2) erfcx(x), x >= 0,   erfc(x) = 2 - exp(-x^2) erfcx(-x), x < 0,
// branch-free: erfcx by one polynomial over the whole half line, (1 + 2 x) erfcx(x) = 1 + p(q), q = (x - 2) / (x + 2) in [-1, 1)
// (degree 11, Chebyshev interpolant converted to monomials: 1.7e-8 of erfcx in exact arithmetic; the variable change is the one of
// Shepherd & Laframboise's (1 + 2x) exp(x^2) erfc x expansion), and exp(-x^2) with the rounding error of x^2 and of the change of base
// carried into the exponent (expf(-x * x) alone loses x^2 2^-24: 2e-6 at the likelihood floor).  The two exponentials are the Gaussian
// densities the backward needs.  Against float64 (tests/test_gpu_model.py::test_gauss_cond_likelihood_against_float64): the same
// accuracy class as the erfcf form -- relative in the tail, 2e-7 absolute where the two erfc values are O(1).
__device__ __forceinline__ float exp_neg_sq(float x) {   // exp(-x^2), |x| <= 1e4
  const float kL2Ehi = 1.44269502162933349609375f, kL2Elo = 1.92596299112661746e-8f;   // log2(e) = hi + lo
  const float s = x * x, slo = __builtin_fmaf(x, x, -s);
  const float ti = __builtin_rintf(s * kL2Ehi);
  float tf = __builtin_fmaf(s, kL2Ehi, -ti);
  tf = __builtin_fmaf(s, kL2Elo, tf);
  tf = __builtin_fmaf(slo, kL2Ehi, tf);
  return __builtin_ldexpf(__builtin_amdgcn_exp2f(-tf), -(int)ti);   // (|tf| <= 1/2 + ...: no denormal path needed in v_exp_f32)
}
__device__ __forceinline__ float erfcx_pos(float a) {   // exp(a^2) erfc(a), 0 <= a <= 1e4
  const float q = (a - 2.f) * __builtin_amdgcn_rcpf(a + 2.f);
  float p = 6.807514774e-05f;
  p = __builtin_fmaf(p, q, 1.608403462e-04f);
  p = __builtin_fmaf(p, q, -3.709284118e-04f);
  p = __builtin_fmaf(p, q, -1.395805088e-03f);
  p = __builtin_fmaf(p, q, 1.230503218e-03f);
  p = __builtin_fmaf(p, q, 8.689252695e-03f);
  p = __builtin_fmaf(p, q, -8.024739103e-03f);
  p = __builtin_fmaf(p, q, -5.421199524e-02f);
  p = __builtin_fmaf(p, q, 1.640504971e-01f);
  p = __builtin_fmaf(p, q, -1.660310904e-01f);
  p = __builtin_fmaf(p, q, -9.276381574e-02f);
  p = __builtin_fmaf(p, q, 2.769783912e-01f);
  const float r = __builtin_amdgcn_rcpf(__builtin_fmaf(2.f, a, 1.f));
  return __builtin_fmaf(p, r, r);
}
// 1 / (sqrt 2 sigma): hardware reciprocal + one Newton step (the argument error is amplified by 2 x^2 in the tail)
__device__ __forceinline__ float gc_rs(float sg) {
  const float r0 = __builtin_amdgcn_rcpf(sg);
  return __builtin_fmaf(__builtin_fmaf(-sg, r0, 1.f), r0, r0) * 0.70710678118654752440f;
}
// -> the raw likelihood; e_lo = exp(-xl^2) = exp(-zu^2 / 2), e_hi = exp(-xh^2) = exp(-zl^2 / 2)
__device__ __forceinline__ float gc_lik(float a, float rs, float& e_lo, float& e_hi) {
  const float xh = fminf((a + 0.5f) * rs, 1e4f), xl = fminf(fmaxf((a - 0.5f) * rs, -1e4f), 1e4f);
  e_hi = exp_neg_sq(xh);
  e_lo = exp_neg_sq(xl);
  const float th = e_hi * erfcx_pos(xh), tl = e_lo * erfcx_pos(fabsf(xl));
  const float l = 0.5f * ((xl < 0.f ? 2.f - tl : tl) - th);
  return a != a ? a : l;   // (the clamps above would swallow a NaN latent)
}

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
  crdr_gc_desc2 d;
  crdr_gc_io io;
};

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so the backward pass regenerates the forward's samples from
// (seed, offset, element index) instead of storing 4 B per latent element.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// U(-1/2, 1/2) sample of global element `idx` (24 random bits, like torch.rand's fp32 path: [0, 1) - 1/2)
__device__ __forceinline__ float philox_uniform(const uint64_t* ph, unsigned long long idx) {
  const unsigned long long seed = ph[0], ctr = ph[1] + (idx >> 2);
  unsigned o[4];
  philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
  const unsigned r = o[idx & 3];
  return (float)(r >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

__device__ __forceinline__ float gc_noise(const GcArgs& p, size_t pix, int c) {
  if (p.io.noise) return p.io.noise[pix * p.d.ldnoise + c];
  return philox_uniform(p.io.philox, (unsigned long long)pix * p.d.Ctot + p.d.c0 + c);
}

// One element of the forward: quantise, both likelihoods, bit terms.
__device__ __forceinline__ void gc_elem(const GcArgs& p, float yv, float m, float sraw, float u, bool noisy, bool want_q, float& q_plus_m,
                                        float& lq, float& ln, float& sn, float& sq) {
  const float sg = fmaxf(sraw, p.d.scale_bound);
  const float rs = gc_rs(sg);
  const float q = rintf(yv - m);  // torch.round: half to even
  q_plus_m = q + m;
  float e0, e1;
  if (want_q) {
    float l = gc_lik(fabsf(q), rs, e0, e1);  // |round(y - mu) + mu - mu|
    l = fmaxf(l, p.d.likelihood_bound);
    lq = l;
    sq -= __builtin_amdgcn_logf(l);   // v_log_f32 = log2 (l >= the bound: never denormal)
  }
  if (noisy) {
    float l = gc_lik(fabsf(yv + u - m), rs, e0, e1);
    l = fmaxf(l, p.d.likelihood_bound);
    ln = l;
    sn -= __builtin_amdgcn_logf(l);
  }
}

// grid (B, N), 256 threads: block b of image n strides over that image's HW*C elements.  VEC: four consecutive channels per
// thread through 16-byte accesses (every pointer 16-byte aligned, C / strides / Ctot / c0 multiples of 4), one Philox call
// per four samples.  The two bit sums of a block go to part[(n * B + b) * 2 + {0, 1}] when `part` is set (finished in fixed
// order by gauss_cond_finish_kernel: no float atomics, any size, bit-reproducible run to run); with B == 1 and no `part` the
// block adds straight into bits[n].
template <bool VEC>
__global__ __launch_bounds__(256) void gauss_cond_fwd_kernel(const GcArgs p, float* part) {
  __shared__ float red[16];
  const int n = blockIdx.y, C = p.d.C;
  const int per_img = p.d.HW * C;
  const bool noisy = p.io.noise || p.io.philox;
  const bool want_q = p.io.lik_quant || p.io.bits_quant;
  float sn = 0.f, sq = 0.f;
  if (VEC) {
    const int C4 = C >> 2;
    for (int e4 = blockIdx.x * 256 + threadIdx.x; e4 < (per_img >> 2); e4 += gridDim.x * 256) {
      const int px = e4 / C4, c = (e4 - px * C4) << 2;
      const size_t pix = (size_t)n * p.d.HW + px;
      const float4 yv = *reinterpret_cast<const float4*>(p.io.y + pix * p.d.ldy + c);
      const float4 mv = *reinterpret_cast<const float4*>(p.io.mu + pix * p.d.ldmu + c);
      const float4 sv = *reinterpret_cast<const float4*>(p.io.sigma + pix * p.d.ldsigma + c);
      float u[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.io.noise) {
        const float4 nv = *reinterpret_cast<const float4*>(p.io.noise + pix * p.d.ldnoise + c);
        u[0] = nv.x; u[1] = nv.y; u[2] = nv.z; u[3] = nv.w;
      } else if (p.io.philox) {  // element index = pix * Ctot + c0 + c, a multiple of 4: the four samples of one counter
        const unsigned long long idx = (unsigned long long)pix * p.d.Ctot + p.d.c0 + c;
        const unsigned long long seed = p.io.philox[0], ctr = p.io.philox[1] + (idx >> 2);
        unsigned o[4];
        philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = (float)(o[j] >> 8) * (1.0f / 16777216.0f) - 0.5f;
      }
      const float ya[4] = {yv.x, yv.y, yv.z, yv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, sa[4] = {sv.x, sv.y, sv.z, sv.w};
      float qm[4], lq[4], ln[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) gc_elem(p, ya[j], ma[j], sa[j], u[j], noisy, want_q, qm[j], lq[j], ln[j], sn, sq);
      const float4 qv = make_float4(qm[0], qm[1], qm[2], qm[3]);
      if (p.io.yhat) *reinterpret_cast<float4*>(p.io.yhat + pix * p.d.ldyhat + c) = qv;
      if (p.io.yhat2) *reinterpret_cast<float4*>(p.io.yhat2 + pix * p.d.ldyhat2 + c) = qv;
      if (p.io.lik_quant) *reinterpret_cast<float4*>(p.io.lik_quant + pix * p.d.ldlik + c) = make_float4(lq[0], lq[1], lq[2], lq[3]);
      if (p.io.lik_noisy && noisy) *reinterpret_cast<float4*>(p.io.lik_noisy + pix * p.d.ldlik + c) = make_float4(ln[0], ln[1], ln[2], ln[3]);
    }
  } else {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per_img; e += gridDim.x * 256) {
      const int px = e / C, c = e - px * C;
      const size_t pix = (size_t)n * p.d.HW + px;
      float qm, lq = 0.f, ln = 0.f;
      gc_elem(p, p.io.y[pix * p.d.ldy + c], p.io.mu[pix * p.d.ldmu + c], p.io.sigma[pix * p.d.ldsigma + c],
              noisy ? gc_noise(p, pix, c) : 0.f, noisy, want_q, qm, lq, ln, sn, sq);
      if (p.io.yhat) p.io.yhat[pix * p.d.ldyhat + c] = qm;
      if (p.io.yhat2) p.io.yhat2[pix * p.d.ldyhat2 + c] = qm;
      if (p.io.lik_quant) p.io.lik_quant[pix * p.d.ldlik + c] = lq;
      if (p.io.lik_noisy && noisy) p.io.lik_noisy[pix * p.d.ldlik + c] = ln;
    }
  }
  sn = block_sum(sn, red);
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) {
    if (part) {
      float* o = part + ((size_t)n * gridDim.x + blockIdx.x) * 2;
      o[0] = sn; o[1] = sq;
    } else {  // gridDim.x == 1
      if (p.io.bits_noisy && noisy) p.io.bits_noisy[n] += sn;
      if (p.io.bits_quant) p.io.bits_quant[n] += sq;
    }
  }
}

// one block per image: thread t adds the partials of blocks t, t + 256, ... in that order (every load in flight at once: a single
// wave walking 2 048 partials one after the other took 8.7 us at the codec's sizes), then the fixed shuffle + LDS tree of block_sum
__global__ __launch_bounds__(256) void gauss_cond_finish_kernel(const float* part, int B, float* bits_noisy, float* bits_quant) {
  __shared__ float red[16];
  const int n = blockIdx.x;
  float sn = 0.f, sq = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)n * B + b) * 2);
    sn += v.x; sq += v.y;
  }
  sn = block_sum(sn, red);
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) {
    if (bits_noisy) bits_noisy[n] += sn;
    if (bits_quant) bits_quant[n] += sq;
  }
}

__device__ __forceinline__ void gc_elem_bwd(const GcArgs& p, float yv, float m, float sraw, float u, float gb, float ste, float& gy_out,
                                            float& gmu, float& gsig) {
  const float inv_ln2 = 1.4426950408889634f;
  const float sg = fmaxf(sraw, p.d.scale_bound);
  const float dlt = yv + u - m;
  const float a = fabsf(dlt), sgn = dlt > 0.f ? 1.f : (dlt < 0.f ? -1.f : 0.f);
  const float rs = gc_rs(sg), inv_sg = rs * 1.41421356237309504880f;
  const float zu = (0.5f - a) * inv_sg, zl = (-0.5f - a) * inv_sg;
  float e_lo, e_hi;
  const float lraw = gc_lik(a, rs, e_lo, e_hi);
  const float l = fmaxf(lraw, p.d.likelihood_bound);
  const float glik = gb * (-inv_ln2 / l);                                         // d(-log2 l)/dl scaled
  const float graw = (lraw >= p.d.likelihood_bound || glik < 0.f) ? glik : 0.f;  // LowerBound backward
  const float pu = 0.39894228040143267794f * e_lo, pl = 0.39894228040143267794f * e_hi;   // the densities at zu, zl
  const float dl_da = (pl - pu) * inv_sg;
  const float dl_dsg = (zl * pl - zu * pu) * inv_sg;
  const float gy = graw * dl_da * sgn;
  const float gsg = graw * dl_dsg;
  gsig = (sraw >= p.d.scale_bound || gsg < 0.f) ? gsg : 0.f;
  gy_out = gy + ste;   // yhat = ste_round(y - mu) + mu: d/dy = 1, d/dmu = 0
  gmu = -gy;
}

template <bool VEC>
__global__ __launch_bounds__(256) void gauss_cond_bwd_kernel(const GcArgs p) {
  const int C = p.d.C;
  const int per_img = p.d.HW * C;
  const int64_t total = (int64_t)p.d.N * per_img;
  if (VEC) {
    const int C4 = C >> 2;
    for (int64_t e4 = blockIdx.x * 256ll + threadIdx.x; e4 < (total >> 2); e4 += gridDim.x * 256ll) {
      const size_t pix = (size_t)(e4 / C4);
      const int c = (int)(e4 - (int64_t)pix * C4) << 2;
      const int n = (int)(pix / p.d.HW);
      const float4 yv = *reinterpret_cast<const float4*>(p.io.y + pix * p.d.ldy + c);
      const float4 mv = *reinterpret_cast<const float4*>(p.io.mu + pix * p.d.ldmu + c);
      const float4 sv = *reinterpret_cast<const float4*>(p.io.sigma + pix * p.d.ldsigma + c);
      float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.io.dyhat) dv = *reinterpret_cast<const float4*>(p.io.dyhat + pix * p.d.lddyhat + c);
      float u[4];
      if (p.io.noise) {
        const float4 nv = *reinterpret_cast<const float4*>(p.io.noise + pix * p.d.ldnoise + c);
        u[0] = nv.x; u[1] = nv.y; u[2] = nv.z; u[3] = nv.w;
      } else {
        const unsigned long long idx = (unsigned long long)pix * p.d.Ctot + p.d.c0 + c;
        const unsigned long long seed = p.io.philox[0], ctr = p.io.philox[1] + (idx >> 2);
        unsigned o[4];
        philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = (float)(o[j] >> 8) * (1.0f / 16777216.0f) - 0.5f;
      }
      const float gb = p.io.gbits[n];
      const float ya[4] = {yv.x, yv.y, yv.z, yv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, sa[4] = {sv.x, sv.y, sv.z, sv.w};
      const float da[4] = {dv.x, dv.y, dv.z, dv.w};
      float gy[4], gm[4], gs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) gc_elem_bwd(p, ya[j], ma[j], sa[j], u[j], gb, da[j], gy[j], gm[j], gs[j]);
      *reinterpret_cast<float4*>(p.io.dy + pix * p.d.ldgrad + c) = make_float4(gy[0], gy[1], gy[2], gy[3]);
      *reinterpret_cast<float4*>(p.io.dmu + pix * p.d.ldgrad + c) = make_float4(gm[0], gm[1], gm[2], gm[3]);
      *reinterpret_cast<float4*>(p.io.dsigma + pix * p.d.ldgrad + c) = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
    return;
  }
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int n = (int)(e / per_img);
    const size_t pix = (size_t)(e / C);
    const int c = (int)(e - (int64_t)pix * C);
    const float ste = p.io.dyhat ? p.io.dyhat[pix * p.d.lddyhat + c] : 0.f;
    float gy, gm, gs;
    gc_elem_bwd(p, p.io.y[pix * p.d.ldy + c], p.io.mu[pix * p.d.ldmu + c], p.io.sigma[pix * p.d.ldsigma + c], gc_noise(p, pix, c),
                p.io.gbits[n], ste, gy, gm, gs);
    p.io.dy[pix * p.d.ldgrad + c] = gy;
    p.io.dmu[pix * p.d.ldgrad + c] = gm;
    p.io.dsigma[pix * p.d.ldgrad + c] = gs;
  }
}

__global__ __launch_bounds__(256) void philox_uniform_kernel(const uint64_t* ph, int64_t M, int C, int Ctot, int c0, float* out, int ld) {
  const int64_t total = M * C;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int64_t pix = e / C;
    const int c = (int)(e - pix * C);
    out[pix * ld + c] = philox_uniform(ph, (unsigned long long)pix * Ctot + c0 + c);
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

static void gc_fill_defaults(crdr_gc_desc2& d) {
  if (!d.ldnoise) d.ldnoise = d.C;
  if (!d.ldlik) d.ldlik = d.C;
  if (!d.ldgrad) d.ldgrad = d.C;
  if (!d.Ctot) d.Ctot = d.C;
}

// blocks per image: one 16-byte group per thread, a whole number of waves over the chip at the codec's sizes
static int gc_blocks(const crdr_gc_desc2* d) {
  const long long per_img = (long long)d->HW * d->C;
  return (int)std::max<long long>(1, std::min<long long>(cdiv64(per_img, 1024), 2048));
}

extern "C" size_t crdr_gauss_cond_fwd_workspace(const crdr_gc_desc2* d) {
  if (!d) return 0;
  return (size_t)d->N * gc_blocks(d) * 2 * sizeof(float);
}

extern "C" int crdr_gauss_cond_fwd2(const crdr_gc_desc2* d, const crdr_gc_io* io, crdr_stream_t s) {
  CRDR_REQUIRE(d && io && io->y && io->mu && io->sigma, "gauss_cond_fwd: null input");
  GcArgs a;
  a.d = *d; a.io = *io;
  gc_fill_defaults(a.d);
  const long long per_img = (long long)d->HW * d->C;
  if (per_img == 0 || d->N == 0) return 0;
  CRDR_REQUIRE(per_img < (1ll << 31), "gauss_cond_fwd: %lld elements per image", per_img);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const crdr_gc_desc2& g = a.d;
  const bool vec = g.C % 4 == 0 && g.ldy % 4 == 0 && g.ldmu % 4 == 0 && g.ldsigma % 4 == 0 && g.Ctot % 4 == 0 && g.c0 % 4 == 0 &&
                   al16(io->y) && al16(io->mu) && al16(io->sigma) && (!io->noise || (g.ldnoise % 4 == 0 && al16(io->noise))) &&
                   (!io->yhat || (g.ldyhat % 4 == 0 && al16(io->yhat))) && (!io->yhat2 || (g.ldyhat2 % 4 == 0 && al16(io->yhat2))) &&
                   ((!io->lik_noisy && !io->lik_quant) || (g.ldlik % 4 == 0 && al16(io->lik_noisy) && al16(io->lik_quant)));
  // bit sums: per-block partials in the caller's workspace + a fixed-order finishing pass.  Without a workspace one block
  // per image adds in place -- equally deterministic, slow for big images.  No float atomics either way.
  int B = gc_blocks(d);
  float* part = io->ws;
  if (!part || io->ws_bytes < crdr_gauss_cond_fwd_workspace(d) || !(io->bits_noisy || io->bits_quant)) {
    part = nullptr;
    if (io->bits_noisy || io->bits_quant) B = 1;
  }
  if (vec)
    hipLaunchKernelGGL(gauss_cond_fwd_kernel<true>, dim3(B, d->N), dim3(256), 0, as_stream(s), a, part);
  else
    hipLaunchKernelGGL(gauss_cond_fwd_kernel<false>, dim3(B, d->N), dim3(256), 0, as_stream(s), a, part);
  CRDR_CHECK_LAUNCH("gauss_cond_fwd");
  if (part) {
    const bool noisy = io->noise || io->philox;
    hipLaunchKernelGGL(gauss_cond_finish_kernel, dim3(d->N), dim3(256), 0, as_stream(s), part, B, noisy ? io->bits_noisy : nullptr,
                       io->bits_quant);
    CRDR_CHECK_LAUNCH("gauss_cond_finish");
  }
  return 0;
}

extern "C" int crdr_gauss_cond_bwd2(const crdr_gc_desc2* d, const crdr_gc_io* io, crdr_stream_t s) {
  CRDR_REQUIRE(d && io && io->y && io->mu && io->sigma && (io->noise || io->philox) && io->gbits && io->dy && io->dmu && io->dsigma,
               "gauss_cond_bwd: null pointer");
  GcArgs a;
  a.d = *d; a.io = *io;
  gc_fill_defaults(a.d);
  const int64_t total = (int64_t)d->N * d->HW * d->C;
  if (total == 0) return 0;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const crdr_gc_desc2& g = a.d;
  const bool vec = g.C % 4 == 0 && g.ldy % 4 == 0 && g.ldmu % 4 == 0 && g.ldsigma % 4 == 0 && g.ldgrad % 4 == 0 && g.Ctot % 4 == 0 &&
                   g.c0 % 4 == 0 && al16(io->y) && al16(io->mu) && al16(io->sigma) && al16(io->dy) && al16(io->dmu) && al16(io->dsigma) &&
                   (!io->noise || (g.ldnoise % 4 == 0 && al16(io->noise))) && (!io->dyhat || (g.lddyhat % 4 == 0 && al16(io->dyhat)));
  const int nb = (int)std::min<int64_t>(cdiv64(vec ? total / 4 : total, 256), 4096);
  if (vec)
    hipLaunchKernelGGL(gauss_cond_bwd_kernel<true>, dim3(nb), dim3(256), 0, as_stream(s), a);
  else
    hipLaunchKernelGGL(gauss_cond_bwd_kernel<false>, dim3(nb), dim3(256), 0, as_stream(s), a);
  CRDR_CHECK_LAUNCH("gauss_cond_bwd");
  return 0;
}

// symbols / CDF indexes of the coded latent, written in the (channel, row, column) order the host coder consumes
// (compressai GaussianConditional.quantize(..., "symbols", means) and build_indexes as called from
// minnen20_charm_context_model.py:186-187,197-199): sym = round(y - mu) as int32, idx = #{table entries < max(sigma, bound)}
// capped at levels - 1.  Thread -> output element (pixel fastest): coalesced 4-byte stores, strided NHWC loads (the tensors
// are a few MB at most).
__global__ __launch_bounds__(256) void gauss_symbols_kernel(const float* y, int ldy, const float* mu, int ldmu, const float* sigma,
                                                            int ldsg, const float* table, int levels, float bound, int N, int HW,
                                                            int C, int* sym, int* idx) {
  __shared__ float st[256];
  for (int i = threadIdx.x; i < levels; i += 256) st[i] = table[i];
  __syncthreads();
  const long long total = (long long)N * C * HW;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int px = (int)(e % HW);
    const long long r = e / HW;
    const int c = (int)(r % C), n = (int)(r / C);
    const size_t pix = (size_t)n * HW + px;
    if (sym) sym[e] = (int)rintf(y[pix * ldy + c] - mu[pix * ldmu + c]);
    if (idx) {
      const float s = fmaxf(sigma[pix * ldsg + c], bound);
      int k = 0;
      for (int i = 0; i < levels - 1; ++i) k += (s > st[i]) ? 1 : 0;
      idx[e] = k;
    }
  }
}

extern "C" int crdr_gauss_symbols(const float* y, int ldy, const float* mu, int ldmu, const float* sigma, int ldsigma,
                                  const float* scale_table, int levels, float scale_bound, int N, int HW, int C, int32_t* symbols,
                                  int32_t* indexes, crdr_stream_t s) {
  CRDR_REQUIRE((!symbols || (y && mu)) && (!indexes || (sigma && scale_table)) && levels >= 1 && levels <= 256, "gauss_symbols: bad arguments");
  const long long total = (long long)N * C * HW;
  if (total == 0) return 0;
  hipLaunchKernelGGL(gauss_symbols_kernel, dim3((int)std::min<long long>(cdiv64(total, 256), 8192)), dim3(256), 0, as_stream(s), y, ldy, mu,
                     ldmu, sigma, ldsigma, scale_table, levels, scale_bound, N, HW, C, symbols, indexes);
  CRDR_CHECK_LAUNCH("gauss_symbols");
  return 0;
}

__global__ void philox_fork_kernel(uint64_t* state, uint64_t* call, unsigned long long inc) {
  call[0] = state[0];
  call[1] = state[1];
  state[1] += inc;
}

extern "C" int crdr_philox_fork(uint64_t* state, uint64_t* call, uint64_t inc, crdr_stream_t s) {
  CRDR_REQUIRE(state && call, "philox_fork: null pointer");
  hipLaunchKernelGGL(philox_fork_kernel, dim3(1), dim3(1), 0, as_stream(s), state, call, (unsigned long long)inc);
  CRDR_CHECK_LAUNCH("philox_fork");
  return 0;
}

extern "C" int crdr_philox_uniform(const uint64_t* philox, int N, int HW, int C, int Ctot, int c0, float* out, int ld,
                                   crdr_stream_t s) {
  CRDR_REQUIRE(philox && out, "philox_uniform: null pointer");
  const int64_t M = (int64_t)N * HW, total = M * C;
  if (total == 0) return 0;
  hipLaunchKernelGGL(philox_uniform_kernel, dim3((int)std::min<int64_t>(cdiv64(total, 256), 4096)), dim3(256), 0, as_stream(s),
                     philox, M, C, Ctot, c0, out, ld);
  CRDR_CHECK_LAUNCH("philox_uniform");
  return 0;
}

extern "C" int crdr_gauss_cond_fwd(const crdr_gc_desc* d, const float* y, const float* mu, const float* sigma,
                                   const float* noise, float* yhat, float* lik_noisy, float* lik_quant,
                                   float* bits_noisy, float* bits_quant, crdr_stream_t s) {
  crdr_gc_desc2 d2;
  memset(&d2, 0, sizeof(d2));
  d2.N = d->N; d2.HW = d->HW; d2.C = d->C; d2.ldy = d->ldy; d2.ldmu = d->ldmu; d2.ldsigma = d->ldsigma; d2.ldyhat = d->ldyhat;
  d2.scale_bound = d->scale_bound; d2.likelihood_bound = d->likelihood_bound;
  crdr_gc_io io;
  memset(&io, 0, sizeof(io));
  io.y = y; io.mu = mu; io.sigma = sigma; io.noise = noise; io.yhat = yhat; io.lik_noisy = lik_noisy; io.lik_quant = lik_quant;
  io.bits_noisy = bits_noisy; io.bits_quant = bits_quant;
  return crdr_gauss_cond_fwd2(&d2, &io, s);
}

extern "C" int crdr_gauss_cond_bwd(const crdr_gc_desc* d, const float* y, const float* mu, const float* sigma,
                                   const float* noise, const float* gbits, const float* dyhat, int lddyhat, float* dy,
                                   float* dmu, float* dsigma, crdr_stream_t s) {
  crdr_gc_desc2 d2;
  memset(&d2, 0, sizeof(d2));
  d2.N = d->N; d2.HW = d->HW; d2.C = d->C; d2.ldy = d->ldy; d2.ldmu = d->ldmu; d2.ldsigma = d->ldsigma; d2.ldyhat = d->ldyhat;
  d2.lddyhat = lddyhat; d2.scale_bound = d->scale_bound; d2.likelihood_bound = d->likelihood_bound;
  crdr_gc_io io;
  memset(&io, 0, sizeof(io));
  io.y = y; io.mu = mu; io.sigma = sigma; io.noise = noise; io.gbits = gbits; io.dyhat = dyhat; io.dy = dy; io.dmu = dmu;
  io.dsigma = dsigma;
  return crdr_gauss_cond_bwd2(&d2, &io, s);
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
