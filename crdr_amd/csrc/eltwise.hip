// HBM-bound fused elementwise / reduction kernels of the CRDR training step (fp32).
// All reductions are two-stage (per-block partials in a caller workspace, then a fixed-order final sum) so
// results are bit-reproducible run to run -- no float atomics anywhere in this file.

#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace crdr {

// ------------------------------------------------------------------------------------------------------------
// epilogue backward
// ------------------------------------------------------------------------------------------------------------
struct EbwdArgs {
  crdr_ebwd_desc d;
  crdr_ebwd_io io;
  float* partial;  // [nblocks][4][C]
  int rows_per_block;
};

// Generic (any C, any ld) path: block = 64 (channels) x 4 (rows); grid (strips, 1).
__global__ __launch_bounds__(256) void ebwd_kernel(const EbwdArgs p) {
  __shared__ float red[4][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int f = p.d.flags, C = p.d.C;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.d.M ? r0 + p.rows_per_block : p.d.M;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
      const float sc = (f & CRDR_EPI_AFFINE) ? p.io.scale[c] : 1.f;
      const float sh = (f & CRDR_EPI_AFFINE) ? p.io.shift[c] : 0.f;
      const float v2 = (f & CRDR_EPI_VEC2) ? p.io.vec2[c] : 0.f;
      for (int64_t m = r0 + ty; m < r1; m += 4) {
        float g = p.io.dout[m * p.d.lddout + c];
        float o = (f & (CRDR_EPI_AFFINE | CRDR_EPI_RELU | CRDR_EPI_LRELU)) ? p.io.out[m * p.d.ldout + c] : 0.f;
        if (f & CRDR_EPI_AFFINE) {
          const float u = (o - sh) / sc;
          s2 += g * u;
          s3 += g;
          g *= sc;
          o = u;
        }
        if (f & CRDR_EPI_GATE) {
          const float sg = p.io.sig[m * p.d.ldg + c], tr = p.io.gt[m * p.d.ldg + c];
          p.io.gres[m * p.d.ldgres + c] = g;
          p.io.dgt[m * p.d.ldg + c] = g * sg;
          g = g * tr * sg * (1.f - sg);
        } else if ((f & CRDR_EPI_AFFINE) && (f & CRDR_EPI_RES)) {
          p.io.gres[m * p.d.ldgres + c] = g;
        }
        if (f & CRDR_EPI_VEC2) s1 += g;
        if (f & CRDR_EPI_RELU) g = (o - v2) > 0.f ? g : 0.f;
        if (f & CRDR_EPI_LRELU) g = o > 0.f ? g : 0.2f * g;
        if (p.io.dz) p.io.dz[m * p.d.lddz + c] = g;
        s0 += g;
      }
    }
    red[0][ty][tx] = s0; red[1][ty][tx] = s1; red[2][ty][tx] = s2; red[3][ty][tx] = s3;
    __syncthreads();
    if (ty == 0 && c < C) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        p.partial[((size_t)blockIdx.x * 4 + k) * C + c] = red[k][0][tx] + red[k][1][tx] + red[k][2][tx] + red[k][3][tx];
    }
    __syncthreads();
  }
}

// Narrow tensors (C <= 4: the RGB ends of the generator and the discriminator), no gate: one thread per ROW.  The generic kernel's 64
// channel lanes leave 61 of 64 idle there (258 us for the 16 x 3 x 256 x 256 image gradient of the step, 0.05 TB/s).  Same per-element
// arithmetic; the rows of a block are summed per thread (stride 256) and then over the threads in a fixed tree.
__global__ __launch_bounds__(256) void ebwd_kernel_narrow(const EbwdArgs p) {
  __shared__ float red[16][256];   // [4 sums x 4 channels][thread]
  const int f = p.d.flags, C = p.d.C, tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.d.M ? r0 + p.rows_per_block : p.d.M;
  float s[4][4];
  float sc[4], sh[4], v2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool live = c < C;
    sc[c] = (live && (f & CRDR_EPI_AFFINE)) ? p.io.scale[c] : 1.f;
    sh[c] = (live && (f & CRDR_EPI_AFFINE)) ? p.io.shift[c] : 0.f;
    v2[c] = (live && (f & CRDR_EPI_VEC2)) ? p.io.vec2[c] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k][c] = 0.f;
  }
  const bool need_out = f & (CRDR_EPI_AFFINE | CRDR_EPI_RELU | CRDR_EPI_LRELU);
  for (int64_t m = r0 + tid; m < r1; m += 256) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c >= C) continue;
      float g = p.io.dout[m * p.d.lddout + c];
      float o = need_out ? p.io.out[m * p.d.ldout + c] : 0.f;
      if (f & CRDR_EPI_AFFINE) {
        const float u = (o - sh[c]) / sc[c];
        s[2][c] += g * u;
        s[3][c] += g;
        g *= sc[c];
        o = u;
        if (f & CRDR_EPI_RES) p.io.gres[m * p.d.ldgres + c] = g;
      }
      if (f & CRDR_EPI_VEC2) s[1][c] += g;
      if (f & CRDR_EPI_RELU) g = (o - v2[c]) > 0.f ? g : 0.f;
      if (f & CRDR_EPI_LRELU) g = o > 0.f ? g : 0.2f * g;
      if (p.io.dz) p.io.dz[m * p.d.lddz + c] = g;
      s[0][c] += g;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) red[k * 4 + c][tid] = s[k][c];
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) {
#pragma unroll
      for (int e = 0; e < 16; ++e) red[e][tid] += red[e][tid + w];
    }
    __syncthreads();
  }
  if (tid < 16 && (tid & 3) < C) p.partial[((size_t)blockIdx.x * 4 + (tid >> 2)) * C + (tid & 3)] = red[tid][0];
}

// Vector path (C % 4 == 0, all strides % 4 == 0): 16 B per lane.  block = 16 channel-quads (64 channels) x 16 rows;
// grid (strips, ceil(C/64)).  Each thread keeps 4 sums x 4 channels; rows are reduced through LDS in fixed order.
__global__ __launch_bounds__(256) void ebwd_kernel_v4(const EbwdArgs p) {
  __shared__ f32x4 red[4][16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int f = p.d.flags, C = p.d.C;
  const int c = blockIdx.y * 64 + tx * 4;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.d.M ? r0 + p.rows_per_block : p.d.M;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
  f32x4 s0 = zero, s1 = zero, s2 = zero, s3 = zero;
  if (c < C) {
    const f32x4 sc = (f & CRDR_EPI_AFFINE) ? *reinterpret_cast<const f32x4*>(p.io.scale + c) : one;
    const f32x4 sh = (f & CRDR_EPI_AFFINE) ? *reinterpret_cast<const f32x4*>(p.io.shift + c) : zero;
    const f32x4 v2 = (f & CRDR_EPI_VEC2) ? *reinterpret_cast<const f32x4*>(p.io.vec2 + c) : zero;
    const bool need_out = f & (CRDR_EPI_AFFINE | CRDR_EPI_RELU | CRDR_EPI_LRELU);
#pragma unroll 2
    for (int64_t m = r0 + ty; m < r1; m += 16) {
      f32x4 g = *reinterpret_cast<const f32x4*>(p.io.dout + m * p.d.lddout + c);
      f32x4 o = need_out ? *reinterpret_cast<const f32x4*>(p.io.out + m * p.d.ldout + c) : zero;
      if (f & CRDR_EPI_AFFINE) {
        const f32x4 u = (o - sh) / sc;
        s2 += g * u;
        s3 += g;
        g *= sc;
        o = u;
      }
      if (f & CRDR_EPI_GATE) {
        const f32x4 sg = *reinterpret_cast<const f32x4*>(p.io.sig + m * p.d.ldg + c);
        const f32x4 tr = *reinterpret_cast<const f32x4*>(p.io.gt + m * p.d.ldg + c);
        *reinterpret_cast<f32x4*>(p.io.gres + m * p.d.ldgres + c) = g;
        *reinterpret_cast<f32x4*>(p.io.dgt + m * p.d.ldg + c) = g * sg;
        g = g * tr * sg * (one - sg);
      } else if ((f & CRDR_EPI_AFFINE) && (f & CRDR_EPI_RES)) {
        *reinterpret_cast<f32x4*>(p.io.gres + m * p.d.ldgres + c) = g;
      }
      if (f & CRDR_EPI_VEC2) s1 += g;
      if (f & CRDR_EPI_RELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = (o[k] - v2[k]) > 0.f ? g[k] : 0.f;
      }
      if (f & CRDR_EPI_LRELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = o[k] > 0.f ? g[k] : 0.2f * g[k];
      }
      if (p.io.dz) *reinterpret_cast<f32x4*>(p.io.dz + m * p.d.lddz + c) = g;
      s0 += g;
    }
  }
  red[0][ty][tx] = s0; red[1][ty][tx] = s1; red[2][ty][tx] = s2; red[3][ty][tx] = s3;
  __syncthreads();
  if (threadIdx.x < 64) {  // thread -> (sum k = tid >> 4, channel quad tx)
    const int k = threadIdx.x >> 4;
    f32x4 a = red[k][0][tx];
#pragma unroll
    for (int r = 1; r < 16; ++r) a += red[k][r][tx];
    if (c < C) *reinterpret_cast<f32x4*>(p.partial + ((size_t)blockIdx.x * 4 + k) * C + c) = a;
  }
}

// out[e] (+)= sum_b partial[b][e], e over K*C columns; 16 columns x 16 block-groups per workgroup, fixed order
// (acc0, optional: columns of the first of the K rows are also added into acc0[0..C))
__global__ __launch_bounds__(256) void colsum_final(const float* partial, int nblocks, int K, int C, float* out,
                                                    int accumulate, float* acc0) {
  __shared__ float red[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + tx;
  const int KC = K * C;
  // (four loads in flight per thread, four partial sums combined in a fixed order: up to 512 partial rows used to be 32 dependent
  // load -> add steps per thread, 7.8 us for a few kilobytes)
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (e < KC) {
    int b = ty;
    for (; b + 48 < nblocks; b += 64) {
      const float a0 = partial[(size_t)b * KC + e], a1 = partial[(size_t)(b + 16) * KC + e], a2 = partial[(size_t)(b + 32) * KC + e],
                  a3 = partial[(size_t)(b + 48) * KC + e];
      v0 += a0; v1 += a1; v2 += a2; v3 += a3;
    }
    for (; b < nblocks; b += 16) v0 += partial[(size_t)b * KC + e];
  }
  red[ty][tx] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (ty == 0 && e < KC) {
    float t = red[0][tx];
#pragma unroll
    for (int r = 1; r < 16; ++r) t += red[r][tx];
    out[e] = accumulate ? out[e] + t : t;
    if (acc0 && e < C) acc0[e] += t;
  }
}

// the same second stage writing through a pointer table: column e goes to outs[e / block][e % block]
__global__ __launch_bounds__(256) void colsum_final_scatter(const float* partial, int nblocks, int C, int block,
                                                            float* const* outs, int accumulate) {
  __shared__ float red[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + tx;
  float v = 0.f;
  if (e < C)
    for (int b = ty; b < nblocks; b += 16) v += partial[(size_t)b * C + e];
  red[ty][tx] = v;
  __syncthreads();
  if (ty == 0 && e < C) {
    float t = red[0][tx];
#pragma unroll
    for (int r = 1; r < 16; ++r) t += red[r][tx];
    float* o = outs[e / block];
    if (o) {
      o += e % block;
      *o = accumulate ? *o + t : t;
    }
  }
}

// Finish the in-epilogue column sums of the conv kernels (CRDR_EPI_COLSUM): all pending jobs in two launches.  A conv
// leaves one partial row per (phase, M tile) -- up to 16 K rows for the discriminator's 2 M-pixel layers -- so pass A sums
// slabs of kColsumSlab rows (tile = 64 columns x one slab, four interleaved row lanes added in order) into a scratch row
// per slab, pass B adds the slab rows in order and writes / accumulates the targets.  Fixed order throughout.
constexpr int kColsumSlab = 128;
__device__ __forceinline__ int find_job(const long long* prefix, int n, long long tl) {
  int lo = 0, hi = n - 1;  // last job with prefix[job] <= tl  (block-uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= tl) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void colsum_finish_a_kernel(const crdr_colsum_job* jobs, const long long* prefix_a,
                                                              const long long* meta, float* scratch) {
  __shared__ float red[2][4][64];
  const int n = (int)meta[0];
  const long long total = meta[1];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (long long tl = blockIdx.x; tl < total; tl += gridDim.x) {
    const int lo = find_job(prefix_a, n, tl);
    const crdr_colsum_job jb = jobs[lo];
    const int rel = (int)(tl - prefix_a[lo]);
    const int cb = rel / jb.nslab, slab = rel - cb * jb.nslab;
    const int c = cb * 64 + tx;
    const int r0 = slab * kColsumSlab, r1 = min(jb.rows, r0 + kColsumSlab);
    float a = 0.f, b = 0.f;
    if (c < jb.C)
      for (int r = r0 + ty; r < r1; r += 4) {
        a += jb.cs[((size_t)r * 2 + 0) * jb.ld + c];
        b += jb.cs[((size_t)r * 2 + 1) * jb.ld + c];
      }
    red[0][ty][tx] = a; red[1][ty][tx] = b;
    __syncthreads();
    if (ty == 0) {
      float* dst = scratch + jb.scratch_off + ((size_t)slab * 2) * jb.cpad + c;
      dst[0] = ((red[0][0][tx] + red[0][1][tx]) + red[0][2][tx]) + red[0][3][tx];
      dst[jb.cpad] = ((red[1][0][tx] + red[1][1][tx]) + red[1][2][tx]) + red[1][3][tx];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void colsum_finish_b_kernel(const crdr_colsum_job* jobs, const long long* prefix_b,
                                                              const long long* meta, const float* scratch) {
  __shared__ float red[2][4][64];
  const int n = (int)meta[0];
  const long long total = meta[2];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (long long tl = blockIdx.x; tl < total; tl += gridDim.x) {
    const int lo = find_job(prefix_b, n, tl);
    const crdr_colsum_job jb = jobs[lo];
    const int c = (int)(tl - prefix_b[lo]) * 64 + tx;
    const float* src = scratch + jb.scratch_off + c;
    float a = 0.f, b = 0.f;
    for (int sl = ty; sl < jb.nslab; sl += 4) {
      a += src[((size_t)sl * 2 + 0) * jb.cpad];
      b += src[((size_t)sl * 2 + 1) * jb.cpad];
    }
    red[0][ty][tx] = a; red[1][ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c < jb.C) {
      const float sa = ((red[0][0][tx] + red[0][1][tx]) + red[0][2][tx]) + red[0][3][tx];
      const float sb = ((red[1][0][tx] + red[1][1][tx]) + red[1][2][tx]) + red[1][3][tx];
      if (jb.out_pre) jb.out_pre[c] = jb.accumulate ? jb.out_pre[c] + sa : sa;
      if (jb.out_post) jb.out_post[c] = jb.accumulate ? jb.out_post[c] + sb : sb;
    }
    __syncthreads();
  }
}

// rows per strip: large enough that the second stage sums <= 512 partials per column
static int ebwd_blocks(int64_t M, int* rpb) {
  int r = 64;
  while (cdiv64(M, r) > 512) r *= 2;
  *rpb = r;
  return (int)cdiv64(M, r);
}

// ------------------------------------------------------------------------------------------------------------
// simple maps
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_kernel(const float* x, int ldx, const float* scale, const float* shift,
                                                     float* y, int ldy, int64_t M, int C) {
  const int64_t total = M * C;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int64_t m = e / C;
    const int c = (int)(e - m * C);
    y[m * ldy + c] = x[m * ldx + c] * scale[c] + shift[c];
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int ldx, int64_t M, int C, float* partial,
                                                     int rows_per_block) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + tx;
    float s = 0.f;
    if (c < C)
      for (int64_t m = r0 + ty; m < r1; m += 4) s += x[m * ldx + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) partial[(size_t)blockIdx.x * C + c] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void colsum_kernel_v4(const float* x, int ldx, int64_t M, int C, float* partial,
                                                        int rows_per_block) {
  __shared__ f32x4 red[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + tx * 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int64_t m = r0 + ty; m < r1; m += 16) s += *reinterpret_cast<const f32x4*>(x + m * ldx + c);
  red[ty][tx] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    f32x4 a = red[0][tx];
#pragma unroll
    for (int r = 1; r < 16; ++r) a += red[r][tx];
    *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.x * C + c) = a;
  }
}

__device__ __forceinline__ float softplusf_(float w) { return w > 20.f ? w : log1pf(expf(w)); }

__global__ void interp_ca_params_kernel(const float* W, const float* B, int L, int C, float q, float* scale,
                                        float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float lf = floorf(q);
  const float rf = fminf(lf + 1.f, (float)(L - 1));
  const float a = rf - q;
  const int l = (int)lf, r = (int)rf;
  const float w = W[l * C + c] * a + W[r * C + c] * (1.f - a);
  scale[c] = softplusf_(w);
  shift[c] = B ? B[l * C + c] * a + B[r * C + c] * (1.f - a) : 0.f;
}

__global__ void interp_ca_params_bwd_kernel(const float* W, int L, int C, float q, const float* dscale,
                                            const float* dshift, float* dW, float* dB) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float lf = floorf(q);
  const float rf = fminf(lf + 1.f, (float)(L - 1));
  const float a = rf - q;
  const int l = (int)lf, r = (int)rf;
  const float w = W[l * C + c] * a + W[r * C + c] * (1.f - a);
  const float dw = dscale[c] * (w > 20.f ? 1.f : 1.f / (1.f + expf(-w)));  // softplus' = sigmoid
  // l == r (q == L-1) => a = 0: the reference's out = t[l]*a + t[r]*(1-a) sends the whole gradient through r
  dW[l * C + c] += dw * a;
  dW[r * C + c] += dw * (1.f - a);
  if (dB) {
    dB[l * C + c] += dshift[c] * a;
    dB[r * C + c] += dshift[c] * (1.f - a);
  }
}

__global__ __launch_bounds__(256) void lrp_kernel(const float* a, int lda, const float* z, int ldz, float* y, int ldy,
                                                  int64_t M, int C) {
  const int64_t total = M * C;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int64_t m = e / C;
    const int c = (int)(e - m * C);
    y[m * ldy + c] = a[m * lda + c] + 0.5f * tanhf(z[m * ldz + c]);
  }
}
__global__ __launch_bounds__(256) void lrp_bwd_kernel(const float* dy, int lddy, const float* z, int ldz, float* dz,
                                                      int lddz, int64_t M, int C) {
  const int64_t total = M * C;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int64_t m = e / C;
    const int c = (int)(e - m * C);
    const float t = tanhf(z[m * ldz + c]);
    dz[m * lddz + c] = dy[m * lddy + c] * 0.5f * (1.f - t * t);
  }
}

// ------------------------------------------------------------------------------------------------------------
// scalar reductions: out = sum f(...)  (partials per block, then one block sums them in order)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

enum { RED_SQDIFF = 0, RED_BCE = 1, RED_SQNORM = 2, RED_DOT = 3 };
template <int OP>
__global__ __launch_bounds__(256) void reduce_kernel(const float* a, const float* b, int64_t n, float target,
                                                     float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < n; e += gridDim.x * 256ll) {
    if (OP == RED_SQDIFF) { const float d = a[e] - b[e]; s += d * d; }
    if (OP == RED_SQNORM) { const float d = a[e]; s += d * d; }
    if (OP == RED_DOT) s += a[e] * b[e];
    if (OP == RED_BCE) {  // BCEWithLogits(x, t) = max(x,0) - x t + log(1 + exp(-|x|))
      const float x = a[e] - b[e];
      s += fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x)));
    }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void reduce_final(const float* partial, int nb, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int e = threadIdx.x; e < nb; e += 256) s += partial[e];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s;
}
static int reduce_blocks(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(cdiv64(n, 1024), 1), 1024); }

__global__ __launch_bounds__(256) void sqdiff_bwd_kernel(const float* a, const float* b, int64_t n, const float* g,
                                                         float gscale, float* da, float* db) {
  const float k = 2.f * gscale * (g ? g[0] : 1.f);
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < n; e += gridDim.x * 256ll) {
    const float d = (a[e] - b[e]) * k;
    if (da) da[e] = d;
    if (db) db[e] = -d;
  }
}
__global__ __launch_bounds__(256) void bce_diff_bwd_kernel(const float* p, const float* q, int64_t n, float target,
                                                           const float* g, float gscale, float* dp, float* dq) {
  const float k = gscale * (g ? g[0] : 1.f);
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < n; e += gridDim.x * 256ll) {
    const float x = p[e] - q[e];
    const float d = (1.f / (1.f + expf(-x)) - target) * k;
    if (dp) dp[e] = d;
    if (dq) dq[e] = -d;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Adam
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   const float* sqnorm, float max_norm) {
  float clip = 1.f;
  if (sqnorm) {
    const float c = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
    clip = c < 1.f ? c : 1.f;
  }
  const float step_size = lr / bc1;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < n; e += gridDim.x * 256ll) {
    const float gr = g[e] * clip;
    const float mm = m[e] + (gr - m[e]) * (1.f - b1);  // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vv = v[e] * b2 + (1.f - b2) * gr * gr;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[e] = mm;
    v[e] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[e] -= step_size * (mm / denom);
  }
}

// Same update with the step-dependent scalars read on the device (so the launch can live inside a HIP graph):
// dyn[0] = lr, dyn[1] = step count (>= 1, already incremented), dyn[2] != 0 -> skip the update entirely.
__global__ __launch_bounds__(256) void adam_dyn_kernel(float* p, const float* g, float* m, float* v, int64_t n, float b1,
                                                       float b2, float eps, const float* dyn, const float* sqnorm,
                                                       float max_norm) {
  if (dyn[2] != 0.f) return;
  const float st = dyn[1];
  const float bc1 = 1.f - powf(b1, st), bc2_sqrt = sqrtf(1.f - powf(b2, st));
  float clip = 1.f;
  if (sqnorm) {
    const float c = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
    clip = c < 1.f ? c : 1.f;
  }
  const float step_size = dyn[0] / bc1;
  // 16-byte accesses on all seven streams where the four pointers allow it (the flat parameter buffers always do); the arithmetic per
  // element is the scalar loop's, operation for operation
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const int64_t n4 = vec ? n >> 2 : 0;
  for (int64_t e4 = blockIdx.x * 256ll + threadIdx.x; e4 < n4; e4 += gridDim.x * 256ll) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[e4], mv = reinterpret_cast<const f32x4*>(m)[e4], vv4 = reinterpret_cast<const f32x4*>(v)[e4];
    f32x4 pv = reinterpret_cast<const f32x4*>(p)[e4], mo, vo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = gv[k] * clip;
      const float mm = mv[k] + (gr - mv[k]) * (1.f - b1);
      const float vv = vv4[k] * b2 + (1.f - b2) * gr * gr;
      mo[k] = mm;
      vo[k] = vv;
      pv[k] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    }
    reinterpret_cast<f32x4*>(m)[e4] = mo;
    reinterpret_cast<f32x4*>(v)[e4] = vo;
    reinterpret_cast<f32x4*>(p)[e4] = pv;
  }
  for (int64_t e = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; e < n; e += gridDim.x * 256ll) {
    const float gr = g[e] * clip;
    const float mm = m[e] + (gr - m[e]) * (1.f - b1);
    const float vv = v[e] * b2 + (1.f - b2) * gr * gr;
    m[e] = mm;
    v[e] = vv;
    p[e] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}

// ------------------------------------------------------------------------------------------------------------
// LPIPS helpers (NHWC)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float* x, float* y, int N, int H, int W, int C,
                                                             int OH, int OW) {
  // (32-bit element index, checked by the host: a 64-bit division is ~150 instructions on this ISA and there were three per element)
  const unsigned total = (unsigned)N * OH * OW * C;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    unsigned r = e / (unsigned)C;
    const int c = (int)(e - r * C);
    unsigned q = r / (unsigned)OW;
    const int ow = (int)(r - q * OW);
    const int n = (int)(q / (unsigned)OH), oh = (int)(q - (unsigned)n * OH);
    float best = -INFINITY;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const int ih = oh * 2 + i, iw = ow * 2 + j;
        if (ih < H && iw < W) best = fmaxf(best, x[(((int64_t)n * H + ih) * W + iw) * C + c]);
      }
    y[e] = best;
  }
}
// gather form: dx[pixel] = sum over the (<=4) windows containing it whose argmax (first max in scan order) is it
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* x, const float* dy, float* dx, int N, int H,
                                                             int W, int C, int OH, int OW) {
  const unsigned total = (unsigned)N * H * W * C;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    unsigned r = e / (unsigned)C;
    const int c = (int)(e - r * C);
    unsigned q = r / (unsigned)W;
    const int iw = (int)(r - q * W);
    const int n = (int)(q / (unsigned)H), ih = (int)(q - (unsigned)n * H);
    float acc = 0.f;
    for (int oh = (ih >= 2 ? (ih - 1) / 2 : 0); oh <= ih / 2 && oh < OH; ++oh)
      for (int ow = (iw >= 2 ? (iw - 1) / 2 : 0); ow <= iw / 2 && ow < OW; ++ow) {
        // argmax of window (oh, ow), first occurrence in row-major scan
        float best = -INFINITY; int bi = -1, bj = -1;
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) {
            const int yy = oh * 2 + i, xx = ow * 2 + j;
            if (yy < H && xx < W) {
              const float v = x[(((int64_t)n * H + yy) * W + xx) * C + c];
              if (v > best) { best = v; bi = yy; bj = xx; }
            }
          }
        if (bi == ih && bj == iw) acc += dy[(((int64_t)n * OH + oh) * OW + ow) * C + c];
      }
    dx[e] = acc;
  }
}

// one wave per pixel: unit-normalise both feature vectors over C, weighted squared difference
__global__ __launch_bounds__(256) void lpips_layer_fwd_kernel(const float* f0, const float* f1, const float* lin,
                                                              int N, int HW, int C, float* partial) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.y;
  float acc = 0.f;
  for (int px = blockIdx.x * 4 + w; px < HW; px += gridDim.x * 4) {
    const float* a = f0 + ((size_t)n * HW + px) * C;
    const float* b = f1 + ((size_t)n * HW + px) * C;
    float sa = 0.f, sb = 0.f;
    for (int c = lane; c < C; c += 64) { sa += a[c] * a[c]; sb += b[c] * b[c]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
    const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
    float d = 0.f;
    for (int c = lane; c < C; c += 64) { const float t = a[c] * ia - b[c] * ib; d += lin[c] * t * t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    acc += d;
  }
  if (lane == 0) red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void lpips_layer_final(const float* partial, int nb, int HW, float* out) {
  const int n = blockIdx.x;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partial[(size_t)n * nb + b];
    out[n] += s / (float)HW;
  }
}
// gradient w.r.t. f1 only (f0 = features of the real image carries no gradient in the training step)
__global__ __launch_bounds__(256) void lpips_layer_bwd_kernel(const float* f0, const float* f1, const float* lin,
                                                              int N, int HW, int C, const float* gout, float* df1) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const float g = gout[n] / (float)HW;
  for (int px = blockIdx.x * 4 + w; px < HW; px += gridDim.x * 4) {
    const float* a = f0 + ((size_t)n * HW + px) * C;
    const float* b = f1 + ((size_t)n * HW + px) * C;
    float* d = df1 + ((size_t)n * HW + px) * C;
    float sa = 0.f, sb = 0.f;
    for (int c = lane; c < C; c += 64) { sa += a[c] * a[c]; sb += b[c] * b[c]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
    const float nb_ = sqrtf(sb);
    const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (nb_ + 1e-10f);
    // e_c = dL/d(bhat_c) = -2 g lin_c (ahat_c - bhat_c); bhat = b * ib; d ib / d b_c = -ib^2 * b_c / |b|
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float e = -2.f * g * lin[c] * (a[c] * ia - b[c] * ib);
      dot += e * b[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    const float k = nb_ > 0.f ? dot * ib * ib / nb_ : 0.f;
    for (int c = lane; c < C; c += 64) {
      const float e = -2.f * g * lin[c] * (a[c] * ia - b[c] * ib);
      d[c] = e * ib - k * b[c];
    }
  }
}


// ---- linear layers on <= 16 row vectors (GEMV style) ---------------------------------------------------------
constexpr int kLinMaxM = 16;
// one wave per output feature: lanes stride over the input features, butterfly reduction at the end
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* x, int M, int I, int ldx, const float* w, const float* b,
                                                         float* y, int O, int ldy, int relu) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  float acc[kLinMaxM];
#pragma unroll
  for (int m = 0; m < kLinMaxM; ++m) acc[m] = 0.f;
  for (int i = lane; i < I; i += 64) {
    const float wv = w[(size_t)o * I + i];
#pragma unroll
    for (int m = 0; m < kLinMaxM; ++m)
      if (m < M) acc[m] += wv * x[(size_t)m * ldx + i];
  }
#pragma unroll
  for (int m = 0; m < kLinMaxM; ++m) {
    float v = acc[m];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0 && m < M) {
      v += b ? b[o] : 0.f;
      y[(size_t)m * ldy + o] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}
__device__ __forceinline__ float lin_g(const float* dy, int lddy, const float* y, int ldy, int m, int o) {
  const float g = dy[(size_t)m * lddy + o];
  return (y && !(y[(size_t)m * ldy + o] > 0.f)) ? 0.f : g;
}
// dw / db: one thread per (o, i)
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* x, int M, int I, int ldx, const float* dy, int lddy,
                                                           const float* y, int ldy, int O, float* dw, float* db) {
  const long long e = blockIdx.x * 256ll + threadIdx.x;
  if (e >= (long long)O * I) return;
  const int o = (int)(e / I), i = (int)(e % I);
  float a = 0.f, s = 0.f;
  for (int m = 0; m < M; ++m) {
    const float g = lin_g(dy, lddy, y, ldy, m, o);
    a += g * x[(size_t)m * ldx + i];
    s += g;
  }
  if (dw) dw[e] += a;
  if (db && i == 0) db[o] += s;
}
// dx: block = 64 input features, the 16 waves split the output features, LDS reduction in wave order
__global__ __launch_bounds__(1024) void linear_bwd_x_kernel(const float* w, int M, int I, const float* dy, int lddy, const float* y,
                                                            int ldy, int O, float* dx, int lddx) {
  __shared__ float red[16][kLinMaxM][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = blockIdx.x * 64 + lane;
  float acc[kLinMaxM];
#pragma unroll
  for (int m = 0; m < kLinMaxM; ++m) acc[m] = 0.f;
  if (i < I)
    for (int o = wave; o < O; o += 16) {
      const float wv = w[(size_t)o * I + i];
#pragma unroll
      for (int m = 0; m < kLinMaxM; ++m)
        if (m < M) acc[m] += wv * lin_g(dy, lddy, y, ldy, m, o);
    }
#pragma unroll
  for (int m = 0; m < kLinMaxM; ++m) red[wave][m][lane] = acc[m];
  __syncthreads();
  if (wave == 0 && i < I)
    for (int m = 0; m < M; ++m) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r][m][lane];
      dx[(size_t)m * lddx + i] = t;
    }
}

// grouped variants (crdr_linear_group_*): blockIdx.y = problem; the shared input rows are read by every problem
__global__ __launch_bounds__(256) void linear_group_fwd_kernel(const float* x, int M, int I, int ldx, const crdr_linear_group g) {
  const int gi = blockIdx.y, O = g.O[gi];
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  const float* w = g.w[gi];
  float acc[kLinMaxM];
#pragma unroll
  for (int m = 0; m < kLinMaxM; ++m) acc[m] = 0.f;
  for (int i = lane; i < I; i += 64) {
    const float wv = w[(size_t)o * I + i];
#pragma unroll
    for (int m = 0; m < kLinMaxM; ++m)
      if (m < M) acc[m] += wv * x[(size_t)m * ldx + i];
  }
#pragma unroll
  for (int m = 0; m < kLinMaxM; ++m) {
    float v = acc[m];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0 && m < M) g.y[gi][(size_t)m * O + o] = v + (g.b[gi] ? g.b[gi][o] : 0.f);
  }
}
__global__ __launch_bounds__(256) void linear_group_bwd_w_kernel(const float* x, int M, int I, int ldx, const crdr_linear_group g) {
  const int gi = blockIdx.y, O = g.O[gi];
  const long long e = blockIdx.x * 256ll + threadIdx.x;
  if (e >= (long long)O * I) return;
  const int o = (int)(e / I), i = (int)(e % I);
  const float* dy = g.dy[gi];
  float a = 0.f, s = 0.f;
  for (int m = 0; m < M; ++m) {
    const float gv = dy[(size_t)m * O + o];
    a += gv * x[(size_t)m * ldx + i];
    s += gv;
  }
  if (g.dw[gi]) g.dw[gi][e] += a;
  if (g.db[gi] && i == 0) g.db[gi][o] += s;
}
// dx over all problems: block = 64 input features, the 16 waves split every problem's output features, fixed-order LDS reduce
__global__ __launch_bounds__(1024) void linear_group_bwd_x_kernel(int M, int I, const crdr_linear_group g, int G, float* dx, int lddx) {
  __shared__ float red[16][kLinMaxM][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = blockIdx.x * 64 + lane;
  float acc[kLinMaxM];
#pragma unroll
  for (int m = 0; m < kLinMaxM; ++m) acc[m] = 0.f;
  if (i < I)
    for (int gi = 0; gi < G; ++gi) {
      const int O = g.O[gi];
      const float* w = g.w[gi];
      const float* dy = g.dy[gi];
      for (int o = wave; o < O; o += 16) {
        const float wv = w[(size_t)o * I + i];
#pragma unroll
        for (int m = 0; m < kLinMaxM; ++m)
          if (m < M) acc[m] += wv * dy[(size_t)m * O + o];
      }
    }
#pragma unroll
  for (int m = 0; m < kLinMaxM; ++m) red[wave][m][lane] = acc[m];
  __syncthreads();
  if (wave == 0 && i < I)
    for (int m = 0; m < M; ++m) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r][m][lane];
      dx[(size_t)m * lddx + i] = t;
    }
}

// gather half of the GEMM + scatter formulation of RGB-output transposed ops (see crdr_col2im_rgb).  SF = 2: stride 2 (the decoder's last layer):
// the taps that reach an output pixel are those of its parity -- (kh / 2 + 1) x (kw / 2 + 1) candidates instead of kh x kw, shifts instead of
// the `% S`, `/ S` of the general form; the pixel index is decoded in 32 bits (the host checks the extent).  The general form spent its time
// in integer divisions (three 64-bit ones per pixel, four 32-bit ones per tap: 97 us for 16 x 256 x 256 pixels, 1.1 TB/s).
template <int SF>   // SF = 1 / 2: stride known at compile time (every tap / the taps of the pixel's parity, no divisions); 0: any stride
__global__ __launch_bounds__(256) void col2im_rgb_kernel(const float* cols, int ldc, int N, int H, int W, int kh, int kw, int S,
                                                         int P, const float* bias, float* out, int ldo, int OH, int OW, int C) {
  const unsigned total = (unsigned)N * OH * OW;
  const f32x4 b4 = {bias ? bias[0] : 0.f, (bias && C > 1) ? bias[1] : 0.f, (bias && C > 2) ? bias[2] : 0.f, (bias && C > 3) ? bias[3] : 0.f};
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    const unsigned r2 = e / (unsigned)OW;
    const int ow = (int)(e - r2 * OW);
    const int n = (int)(r2 / (unsigned)OH), oh = (int)(r2 - (unsigned)n * OH);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* base = cols + (size_t)n * H * W * ldc;
    if constexpr (SF == 1) {
      for (int r = 0; r < kh; ++r) {
        const int ih = oh + P - r;
        if (ih < 0 || ih >= H) continue;
        for (int s = 0; s < kw; ++s) {
          const int iw = ow + P - s;
          if (iw < 0 || iw >= W) continue;
          acc += *reinterpret_cast<const f32x4*>(base + ((size_t)ih * W + iw) * ldc + 4 * (r * kw + s));
        }
      }
    } else if constexpr (SF == 2) {
      for (int r = (oh + P) & 1; r < kh; r += 2) {
        const int ih = (oh + P - r) >> 1;
        if (oh + P - r < 0 || ih >= H) continue;
        for (int s = (ow + P) & 1; s < kw; s += 2) {
          const int iw = (ow + P - s) >> 1;
          if (ow + P - s < 0 || iw >= W) continue;
          acc += *reinterpret_cast<const f32x4*>(base + ((size_t)ih * W + iw) * ldc + 4 * (r * kw + s));
        }
      }
    } else {
      for (int r = 0; r < kh; ++r) {
        const int th = oh + P - r;
        if (th < 0 || th % S) continue;
        const int ih = th / S;
        if (ih >= H) continue;
        for (int s = 0; s < kw; ++s) {
          const int tw = ow + P - s;
          if (tw < 0 || tw % S) continue;
          const int iw = tw / S;
          if (iw >= W) continue;
          acc += *reinterpret_cast<const f32x4*>(base + ((size_t)ih * W + iw) * ldc + 4 * (r * kw + s));
        }
      }
    }
    acc += b4;
    float* o = out + (size_t)e * ldo;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) o[c] = acc[c];
  }
}


// batch of random crops (+ horizontal flip, normalisation to [-1, 1]) out of a uint8 RGB image pool
__global__ __launch_bounds__(256) void crop_flip_normalize_kernel(const uint8_t* pool, const long long* items, int N, int ch, int cw,
                                                                  float* out, int ldo) {
  const long long total = (long long)N * ch * cw;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int c = (int)(e % cw);
    const long long r2 = e / cw;
    const int r = (int)(r2 % ch), n = (int)(r2 / ch);
    const long long* it = items + (size_t)n * 6;
    const int H = (int)it[1], W = (int)it[2];
    int y = (int)it[3] + r, x = (int)it[4] + (it[5] ? cw - 1 - c : c);
    y = y < 0 ? -y : y; y = y >= H ? 2 * (H - 1) - y : y;
    x = x < 0 ? -x : x; x = x >= W ? 2 * (W - 1) - x : x;
    const uint8_t* px = pool + it[0] + ((size_t)y * W + x) * 3;
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = ((float)px[k] / 255.0f - 0.5f) / 0.5f;
    v[3] = 0.f;
    *reinterpret_cast<f32x4*>(out + (size_t)e * ldo) = v;
  }
}


// ---- spectral norm (power iteration on w[O][K]) --------------------------------------------------------------
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* w, const float* u, int O, int K, float* t) {  // t[k] = sum_o w[o][k] u[o]
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float a = 0.f;
  for (int o = 0; o < O; ++o) a += w[(size_t)o * K + k] * u[o];
  t[k] = a;
}
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* w, const float* v, int O, int K, float* t) {  // t[o] = sum_k w[o][k] v[k]
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  float a = 0.f;
  for (int k = lane; k < K; k += 64) a += w[(size_t)o * K + k] * v[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  if (lane == 0) t[o] = a;
}
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < 16; ++i) t += red[i];
  return t;
}
// one block: x <- t / max(||t||, eps)  (normalize != 0), and *dot_out = sum x[i] t[i] (dot_with_t) -- fixed order
__global__ __launch_bounds__(1024) void sn_normalize_kernel(const float* t, int n, float eps, float* x, int normalize, float* dot_out) {
  __shared__ float red[16];
  float a = 0.f;
  if (normalize) {
    for (int i = threadIdx.x; i < n; i += 1024) a += t[i] * t[i];
    const float nrm = fmaxf(sqrtf(block_sum_1024(a, red)), eps);
    for (int i = threadIdx.x; i < n; i += 1024) x[i] = t[i] / nrm;
    __syncthreads();
  }
  if (dot_out) {
    float d = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) d += x[i] * t[i];
    d = block_sum_1024(d, red);
    if (threadIdx.x == 0) *dot_out = d;
  }
}
__global__ __launch_bounds__(256) void sn_scale_kernel(const float* w, const float* sigma, long long n, float* out) {
  const float s = *sigma;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) out[i] = w[i] / s;
}
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* dw_sn, const float* u, const float* v, const float* sigma,
                                                     const float* dot, int O, int K, float* dw) {
  const float s = *sigma, c = *dot / (s * s);
  const long long n = (long long)O * K;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const int o = (int)(i / K), k = (int)(i % K);
    dw[i] += dw_sn[i] / s - c * u[o] * v[k];
  }
}

}  // namespace crdr

using namespace crdr;

static inline int grid_for(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(cdiv64(n, 256), 1), 8192); }

extern "C" size_t crdr_epilogue_bwd_workspace(const crdr_ebwd_desc* d) {
  int rpb;
  const int nb = ebwd_blocks(d->M, &rpb);
  return (size_t)nb * 4 * d->C * sizeof(float);
}

extern "C" int crdr_epilogue_bwd(const crdr_ebwd_desc* d, const crdr_ebwd_io* io, void* ws, size_t ws_bytes,
                                 crdr_stream_t s) {
  const int f = d->flags;
  CRDR_REQUIRE(io->dout && io->colsums, "epilogue_bwd: null dout/colsums");
  CRDR_REQUIRE(!((f & (CRDR_EPI_RELU | CRDR_EPI_LRELU)) && (f & (CRDR_EPI_RES | CRDR_EPI_GATE | CRDR_EPI_AFFINE))),
               "epilogue_bwd: activation mask cannot be recovered when RES/GATE/AFFINE follow it");
  CRDR_REQUIRE(!(f & (CRDR_EPI_AFFINE | CRDR_EPI_RELU | CRDR_EPI_LRELU)) || io->out, "epilogue_bwd: needs saved out");
  CRDR_REQUIRE(!(f & CRDR_EPI_AFFINE) || (io->scale && io->shift), "epilogue_bwd: AFFINE needs scale/shift");
  CRDR_REQUIRE(!(f & CRDR_EPI_GATE) || (io->gt && io->sig && io->gres && io->dgt), "epilogue_bwd: GATE needs gt/sig/gres/dgt");
  CRDR_REQUIRE(!((f & CRDR_EPI_AFFINE) && (f & CRDR_EPI_RES)) || io->gres, "epilogue_bwd: AFFINE+RES needs gres");
  CRDR_REQUIRE(!(f & CRDR_EPI_VEC2) || io->vec2, "epilogue_bwd: VEC2 needs vec2");
  EbwdArgs a;
  a.d = *d; a.io = *io;
  const int nb = ebwd_blocks(d->M, &a.rows_per_block);
  CRDR_REQUIRE(ws_bytes >= (size_t)nb * 4 * d->C * sizeof(float), "epilogue_bwd: workspace too small");
  a.partial = (float*)ws;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool vec = (d->C % 4 == 0) && (d->lddout % 4 == 0) && (d->ldout % 4 == 0) && (d->lddz % 4 == 0) &&
                   (d->ldgres % 4 == 0) && (d->ldg % 4 == 0) && al16(io->dout) && al16(io->out) && al16(io->dz) &&
                   al16(io->gres) && al16(io->dgt) && al16(io->gt) && al16(io->sig) && al16(io->vec2) &&
                   al16(io->scale) && al16(io->shift);
  if (d->M > 0) {
    if (vec) hipLaunchKernelGGL(ebwd_kernel_v4, dim3(nb, cdiv(d->C, 64)), dim3(256), 0, as_stream(s), a);
    else if (d->C <= 4 && !(d->flags & CRDR_EPI_GATE)) hipLaunchKernelGGL(ebwd_kernel_narrow, dim3(nb), dim3(256), 0, as_stream(s), a);
    else hipLaunchKernelGGL(ebwd_kernel, dim3(nb), dim3(256), 0, as_stream(s), a);
    CRDR_CHECK_LAUNCH("ebwd_kernel");
  }
  hipLaunchKernelGGL(colsum_final, dim3(cdiv(4 * d->C, 16)), dim3(256), 0, as_stream(s), (const float*)ws,
                     d->M > 0 ? nb : 0, 4, d->C, io->colsums, 0, io->dbias_accum);
  CRDR_CHECK_LAUNCH("colsum_final");
  return 0;
}

extern "C" int crdr_col2im_rgb(const float* cols, int ldc, int N, int H, int W, int kh, int kw, int stride, int pad,
                               const float* bias, float* out, int ldo, int OH, int OW, int C, crdr_stream_t s) {
  CRDR_REQUIRE(cols && out, "col2im_rgb: null pointer");
  CRDR_REQUIRE(C >= 1 && C <= 4 && ldc % 4 == 0 && ldc >= 4 * kh * kw && stride >= 1, "col2im_rgb: bad geometry (C=%d ldc=%d)", C, ldc);
  CRDR_REQUIRE((reinterpret_cast<uintptr_t>(cols) & 15) == 0, "col2im_rgb: cols must be 16-byte aligned");
  const long long total = (long long)N * OH * OW;
  if (total == 0) return 0;
  CRDR_REQUIRE(total < (1ll << 31), "col2im_rgb: %lld output pixels (32-bit pixel index)", total);
  const dim3 grid((unsigned)std::min<long long>(cdiv64(total, 256), 65535));
  if (stride == 2) hipLaunchKernelGGL(col2im_rgb_kernel<2>, grid, dim3(256), 0, as_stream(s), cols, ldc, N, H, W, kh, kw, stride, pad, bias, out, ldo, OH, OW, C);
  else if (stride == 1) hipLaunchKernelGGL(col2im_rgb_kernel<1>, grid, dim3(256), 0, as_stream(s), cols, ldc, N, H, W, kh, kw, stride, pad, bias, out, ldo, OH, OW, C);
  else hipLaunchKernelGGL(col2im_rgb_kernel<0>, grid, dim3(256), 0, as_stream(s), cols, ldc, N, H, W, kh, kw, stride, pad, bias, out, ldo, OH, OW, C);
  CRDR_CHECK_LAUNCH("col2im_rgb_kernel");
  return 0;
}

extern "C" int crdr_crop_flip_normalize(const uint8_t* pool, const int64_t* items, int N, int crop_h, int crop_w, float* out,
                                        int ldo, crdr_stream_t s) {
  CRDR_REQUIRE(pool && items && out, "crop_flip_normalize: null pointer");
  CRDR_REQUIRE(ldo >= 4 && ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "crop_flip_normalize: out must be 16-byte aligned rows");
  const long long total = (long long)N * crop_h * crop_w;
  if (total == 0) return 0;
  hipLaunchKernelGGL(crop_flip_normalize_kernel, dim3((unsigned)std::min<long long>(cdiv64(total, 256), 65535)), dim3(256), 0,
                     as_stream(s), pool, reinterpret_cast<const long long*>(items), N, crop_h, crop_w, out, ldo);
  CRDR_CHECK_LAUNCH("crop_flip_normalize_kernel");
  return 0;
}

extern "C" int crdr_linear_fwd(const float* x, int M, int I, int ldx, const float* w, const float* b, float* y, int O,
                               int ldy, int relu, crdr_stream_t s) {
  CRDR_REQUIRE(x && w && y, "linear_fwd: null pointer");
  CRDR_REQUIRE(M >= 1 && M <= kLinMaxM, "linear_fwd: M = %d rows (1..%d supported; use crdr_conv2d)", M, kLinMaxM);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(O, 4)), dim3(256), 0, as_stream(s), x, M, I, ldx, w, b, y, O, ldy, relu);
  CRDR_CHECK_LAUNCH("linear_fwd_kernel");
  return 0;
}

extern "C" int crdr_linear_bwd(const float* x, int M, int I, int ldx, const float* w, const float* dy, int lddy,
                               const float* y, int ldy, int O, float* dx, int lddx, float* dw, float* db, crdr_stream_t s) {
  CRDR_REQUIRE(x && w && dy, "linear_bwd: null pointer");
  CRDR_REQUIRE(M >= 1 && M <= kLinMaxM, "linear_bwd: M = %d rows (1..%d supported)", M, kLinMaxM);
  if (dw || db) {
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)cdiv64((long long)O * I, 256)), dim3(256), 0, as_stream(s), x, M, I,
                       ldx, dy, lddy, y, ldy, O, dw, db);
    CRDR_CHECK_LAUNCH("linear_bwd_w_kernel");
  }
  if (dx) {
    hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(cdiv(I, 64)), dim3(1024), 0, as_stream(s), w, M, I, dy, lddy, y, ldy, O, dx,
                       lddx);
    CRDR_CHECK_LAUNCH("linear_bwd_x_kernel");
  }
  return 0;
}

extern "C" int crdr_linear_group_fwd(const float* x, int M, int I, int ldx, const crdr_linear_group* g, int G, crdr_stream_t s) {
  CRDR_REQUIRE(x && g, "linear_group_fwd: null pointer");
  CRDR_REQUIRE(M >= 1 && M <= kLinMaxM, "linear_group_fwd: M = %d rows (1..%d supported)", M, kLinMaxM);
  CRDR_REQUIRE(G >= 1 && G <= CRDR_MAX_GROUP, "linear_group_fwd: %d problems (1..%d)", G, CRDR_MAX_GROUP);
  int omax = 0;
  for (int i = 0; i < G; ++i) {
    CRDR_REQUIRE(g->w[i] && g->y[i] && g->O[i] >= 1, "linear_group_fwd: problem %d incomplete", i);
    omax = std::max(omax, (int)g->O[i]);
  }
  hipLaunchKernelGGL(linear_group_fwd_kernel, dim3(cdiv(omax, 4), G), dim3(256), 0, as_stream(s), x, M, I, ldx, *g);
  CRDR_CHECK_LAUNCH("linear_group_fwd_kernel");
  return 0;
}

extern "C" int crdr_linear_group_bwd(const float* x, int M, int I, int ldx, const crdr_linear_group* g, int G, float* dx, int lddx,
                                     crdr_stream_t s) {
  CRDR_REQUIRE(x && g, "linear_group_bwd: null pointer");
  CRDR_REQUIRE(M >= 1 && M <= kLinMaxM, "linear_group_bwd: M = %d rows (1..%d supported)", M, kLinMaxM);
  CRDR_REQUIRE(G >= 1 && G <= CRDR_MAX_GROUP, "linear_group_bwd: %d problems (1..%d)", G, CRDR_MAX_GROUP);
  int omax = 0;
  bool any_w = false;
  for (int i = 0; i < G; ++i) {
    CRDR_REQUIRE(g->w[i] && g->dy[i] && g->O[i] >= 1, "linear_group_bwd: problem %d incomplete", i);
    omax = std::max(omax, (int)g->O[i]);
    any_w = any_w || g->dw[i] || g->db[i];
  }
  if (any_w) {
    hipLaunchKernelGGL(linear_group_bwd_w_kernel, dim3((unsigned)cdiv64((long long)omax * I, 256), G), dim3(256), 0, as_stream(s), x, M,
                       I, ldx, *g);
    CRDR_CHECK_LAUNCH("linear_group_bwd_w_kernel");
  }
  if (dx) {
    hipLaunchKernelGGL(linear_group_bwd_x_kernel, dim3(cdiv(I, 64)), dim3(1024), 0, as_stream(s), M, I, *g, G, dx, lddx);
    CRDR_CHECK_LAUNCH("linear_group_bwd_x_kernel");
  }
  return 0;
}

extern "C" int crdr_affine(const float* x, int ldx, const float* scale, const float* shift, float* y, int ldy,
                           int64_t M, int C, crdr_stream_t s) {
  CRDR_REQUIRE(x && y && scale && shift, "affine: null pointer");
  if (M * C == 0) return 0;
  hipLaunchKernelGGL(affine_kernel, dim3(grid_for(M * C)), dim3(256), 0, as_stream(s), x, ldx, scale, shift, y, ldy, M, C);
  CRDR_CHECK_LAUNCH("affine");
  return 0;
}

extern "C" size_t crdr_colsum_workspace(int64_t M, int C) {
  int rpb;
  return (size_t)ebwd_blocks(M, &rpb) * C * sizeof(float);
}
extern "C" int crdr_colsum(const float* x, int ldx, int64_t M, int C, float* out, int accumulate, void* ws,
                           size_t ws_bytes, crdr_stream_t s) {
  int rpb;
  const int nb = ebwd_blocks(M, &rpb);
  CRDR_REQUIRE(x && out && ws_bytes >= (size_t)nb * C * sizeof(float), "colsum: bad arguments");
  if (M > 0) {
    if ((C % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0))
      hipLaunchKernelGGL(colsum_kernel_v4, dim3(nb, cdiv(C, 64)), dim3(256), 0, as_stream(s), x, ldx, M, C, (float*)ws, rpb);
    else
      hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, as_stream(s), x, ldx, M, C, (float*)ws, rpb);
    CRDR_CHECK_LAUNCH("colsum");
  }
  hipLaunchKernelGGL(colsum_final, dim3(cdiv(C, 16)), dim3(256), 0, as_stream(s), (const float*)ws, M > 0 ? nb : 0, 1,
                     C, out, accumulate, (float*)nullptr);
  CRDR_CHECK_LAUNCH("colsum_final");
  return 0;
}

extern "C" int crdr_colsum_slab_rows(void) { return kColsumSlab; }

extern "C" int crdr_colsum_finish_batched(const crdr_colsum_job* jobs, const int64_t* prefix_a, const int64_t* prefix_b,
                                          const int64_t* meta, float* scratch, crdr_stream_t s) {
  CRDR_REQUIRE(jobs && prefix_a && prefix_b && meta && scratch, "colsum_finish_batched: null pointer");
  hipLaunchKernelGGL(colsum_finish_a_kernel, dim3(4096), dim3(256), 0, as_stream(s), jobs, reinterpret_cast<const long long*>(prefix_a),
                     reinterpret_cast<const long long*>(meta), scratch);
  CRDR_CHECK_LAUNCH("colsum_finish_a_kernel");
  hipLaunchKernelGGL(colsum_finish_b_kernel, dim3(512), dim3(256), 0, as_stream(s), jobs, reinterpret_cast<const long long*>(prefix_b),
                     reinterpret_cast<const long long*>(meta), (const float*)scratch);
  CRDR_CHECK_LAUNCH("colsum_finish_b_kernel");
  return 0;
}

extern "C" int crdr_colsum_scatter(const float* x, int ldx, int64_t M, int C, int block, float* const* outs, int accumulate,
                                   void* ws, size_t ws_bytes, crdr_stream_t s) {
  int rpb;
  const int nb = ebwd_blocks(M, &rpb);
  CRDR_REQUIRE(x && outs && block > 0 && ws_bytes >= (size_t)nb * C * sizeof(float), "colsum_scatter: bad arguments");
  if (M > 0) {
    if ((C % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0))
      hipLaunchKernelGGL(colsum_kernel_v4, dim3(nb, cdiv(C, 64)), dim3(256), 0, as_stream(s), x, ldx, M, C, (float*)ws, rpb);
    else
      hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, as_stream(s), x, ldx, M, C, (float*)ws, rpb);
    CRDR_CHECK_LAUNCH("colsum");
  }
  hipLaunchKernelGGL(colsum_final_scatter, dim3(cdiv(C, 16)), dim3(256), 0, as_stream(s), (const float*)ws, M > 0 ? nb : 0, C,
                     block, outs, accumulate);
  CRDR_CHECK_LAUNCH("colsum_final_scatter");
  return 0;
}

extern "C" int crdr_interp_ca_params(const float* W, const float* B, int L, int C, float q, float* scale,
                                     float* shift, crdr_stream_t s) {
  CRDR_REQUIRE(W && scale && shift, "interp_ca_params: null pointer");
  CRDR_REQUIRE(q >= 0.f && q <= (float)(L - 1), "interp_ca_params: rate_ind %f outside [0, %d]", q, L - 1);
  hipLaunchKernelGGL(interp_ca_params_kernel, dim3(cdiv(C, 64)), dim3(64), 0, as_stream(s), W, B, L, C, q, scale, shift);
  CRDR_CHECK_LAUNCH("interp_ca_params");
  return 0;
}
extern "C" int crdr_interp_ca_params_bwd(const float* W, int L, int C, float q, const float* dscale,
                                         const float* dshift, float* dW, float* dB, crdr_stream_t s) {
  CRDR_REQUIRE(W && dscale && dW, "interp_ca_params_bwd: null pointer");
  hipLaunchKernelGGL(interp_ca_params_bwd_kernel, dim3(cdiv(C, 64)), dim3(64), 0, as_stream(s), W, L, C, q, dscale,
                     dshift, dW, dB);
  CRDR_CHECK_LAUNCH("interp_ca_params_bwd");
  return 0;
}

extern "C" int crdr_lrp(const float* a, int lda, const float* z, int ldz, float* y, int ldy, int64_t M, int C,
                        crdr_stream_t s) {
  CRDR_REQUIRE(a && z && y, "lrp: null pointer");
  if (M * C == 0) return 0;
  hipLaunchKernelGGL(lrp_kernel, dim3(grid_for(M * C)), dim3(256), 0, as_stream(s), a, lda, z, ldz, y, ldy, M, C);
  CRDR_CHECK_LAUNCH("lrp");
  return 0;
}
extern "C" int crdr_lrp_bwd(const float* dy, int lddy, const float* z, int ldz, float* dz, int lddz, int64_t M, int C,
                            crdr_stream_t s) {
  CRDR_REQUIRE(dy && z && dz, "lrp_bwd: null pointer");
  if (M * C == 0) return 0;
  hipLaunchKernelGGL(lrp_bwd_kernel, dim3(grid_for(M * C)), dim3(256), 0, as_stream(s), dy, lddy, z, ldz, dz, lddz, M, C);
  CRDR_CHECK_LAUNCH("lrp_bwd");
  return 0;
}

// a[m][c0_k + c] = max(a[m][c0_k + c] + bias_k[c], 0) for up to CRDR_MAX_GROUP channel ranges ("slots") of one wide NHWC buffer:
// the first-layer pre-activations of the Charm's transforms are accumulated by several launches (hoisted hyper-prior part,
// one wide conv per decoded support slice); this finishes the slots that are complete.  16-byte accesses, 8 B per element.
struct BiasReluArgs {
  float* a;
  long long M;
  int ld, C, n;
  int c0[CRDR_MAX_GROUP];
  const float* bias[CRDR_MAX_GROUP];
};
__global__ __launch_bounds__(256) void bias_relu_slots_kernel(const BiasReluArgs p) {
  const int C4 = p.C >> 2;
  const long long per = p.M * C4, total = per * p.n;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int k = (int)(e / per);
    const long long r = e - k * per;
    const long long m = r / C4;
    const int c = (int)(r - m * C4) << 2;
    float* q = p.a + m * p.ld + p.c0[k] + c;
    const f32x4 v = *reinterpret_cast<const f32x4*>(q);
    const float* b = p.bias[k] + c;   // (a parameter inside a flat buffer: 4-byte aligned only)
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaxf(v[j] + b[j], 0.f);
    *reinterpret_cast<f32x4*>(q) = o;
  }
}

extern "C" int crdr_bias_relu_slots(float* a, int ld, int64_t M, int C, int n, const int32_t* c0, const float* const* bias, crdr_stream_t s) {
  CRDR_REQUIRE(a && c0 && bias, "bias_relu_slots: null pointer");
  CRDR_REQUIRE(n >= 1 && n <= CRDR_MAX_GROUP && C % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0,
               "bias_relu_slots: %d slots of %d channels (ld %d)", n, C, ld);
  if (M * C == 0) return 0;
  BiasReluArgs p;
  p.a = a; p.M = M; p.ld = ld; p.C = C; p.n = n;
  for (int k = 0; k < n; ++k) {
    CRDR_REQUIRE(bias[k] && c0[k] % 4 == 0, "bias_relu_slots: slot %d misaligned", k);
    p.c0[k] = c0[k]; p.bias[k] = bias[k];
  }
  hipLaunchKernelGGL(bias_relu_slots_kernel, dim3(grid_for(M * (C / 4) * n)), dim3(256), 0, as_stream(s), p);
  CRDR_CHECK_LAUNCH("bias_relu_slots");
  return 0;
}

extern "C" size_t crdr_reduce_workspace(int64_t n) { return (size_t)reduce_blocks(n) * sizeof(float); }

template <int OP>
static int run_reduce(const float* a, const float* b, int64_t n, float target, float* out, void* ws, size_t ws_bytes,
                      crdr_stream_t s, const char* what) {
  const int nb = reduce_blocks(n);
  CRDR_REQUIRE(a && out && ws_bytes >= (size_t)nb * sizeof(float), "%s: bad arguments", what);
  hipLaunchKernelGGL(reduce_kernel<OP>, dim3(nb), dim3(256), 0, as_stream(s), a, b, n, target, (float*)ws);
  CRDR_CHECK_LAUNCH(what);
  hipLaunchKernelGGL(reduce_final, dim3(1), dim3(256), 0, as_stream(s), (const float*)ws, nb, out);
  CRDR_CHECK_LAUNCH(what);
  return 0;
}
extern "C" int crdr_spectral_norm_fwd(const float* w, int O, int K, float* u, float* v, int power_iter, float eps, float* w_out,
                                      float* sigma_out, void* ws, size_t ws_bytes, crdr_stream_t s) {
  CRDR_REQUIRE(w && u && v && w_out && sigma_out && ws, "spectral_norm_fwd: null pointer");
  CRDR_REQUIRE(ws_bytes >= (size_t)(O + K + 2) * sizeof(float), "spectral_norm_fwd: workspace too small");
  float* tv = (float*)ws;       // [K]
  float* tu = tv + K;           // [O]
  hipStream_t st = as_stream(s);
  if (power_iter) {
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3(cdiv(K, 256)), dim3(256), 0, st, w, u, O, K, tv);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, tv, K, eps, v, 1, (float*)nullptr);
  }
  hipLaunchKernelGGL(sn_w_v_kernel, dim3(cdiv(O, 4)), dim3(256), 0, st, w, v, O, K, tu);
  hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, tu, O, eps, u, power_iter ? 1 : 0, sigma_out);  // sigma = u . (w v)
  hipLaunchKernelGGL(sn_scale_kernel, dim3(grid_for((int64_t)O * K)), dim3(256), 0, st, w, sigma_out, (long long)O * K, w_out);
  CRDR_CHECK_LAUNCH("spectral_norm_fwd");
  return 0;
}

extern "C" int crdr_spectral_norm_bwd(const float* dw_sn, const float* w, const float* u, const float* v, const float* sigma,
                                      int O, int K, float* dw, void* ws, size_t ws_bytes, crdr_stream_t s) {
  CRDR_REQUIRE(dw_sn && w && u && v && sigma && dw && ws, "spectral_norm_bwd: null pointer");
  const int64_t n = (int64_t)O * K;
  const int nb = reduce_blocks(n);
  CRDR_REQUIRE(ws_bytes >= (size_t)(nb + 1) * sizeof(float), "spectral_norm_bwd: workspace too small");
  float* part = (float*)ws;
  float* dot = part + nb;
  hipStream_t st = as_stream(s);
  hipLaunchKernelGGL(reduce_kernel<RED_DOT>, dim3(nb), dim3(256), 0, st, dw_sn, w, n, 0.f, part);
  hipLaunchKernelGGL(reduce_final, dim3(1), dim3(256), 0, st, (const float*)part, nb, dot);
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, dw_sn, u, v, sigma, (const float*)dot, O, K, dw);
  CRDR_CHECK_LAUNCH("spectral_norm_bwd");
  return 0;
}

extern "C" int crdr_sqdiff_sum(const float* a, const float* b, int64_t n, float* out, void* ws, size_t ws_bytes,
                               crdr_stream_t s) {
  CRDR_REQUIRE(b, "sqdiff_sum: null pointer");
  return run_reduce<RED_SQDIFF>(a, b, n, 0.f, out, ws, ws_bytes, s, "sqdiff_sum");
}
extern "C" int crdr_bce_diff_sum(const float* p, const float* q, int64_t n, float target, float* out, void* ws,
                                 size_t ws_bytes, crdr_stream_t s) {
  CRDR_REQUIRE(q, "bce_diff_sum: null pointer");
  return run_reduce<RED_BCE>(p, q, n, target, out, ws, ws_bytes, s, "bce_diff_sum");
}
extern "C" int crdr_sqnorm(const float* g, int64_t n, float* out, void* ws, size_t ws_bytes, crdr_stream_t s) {
  return run_reduce<RED_SQNORM>(g, nullptr, n, 0.f, out, ws, ws_bytes, s, "sqnorm");
}
extern "C" int crdr_sqdiff_bwd(const float* a, const float* b, int64_t n, const float* g, float gscale, float* da,
                               float* db, crdr_stream_t s) {
  CRDR_REQUIRE(a && b, "sqdiff_bwd: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(sqdiff_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(s), a, b, n, g, gscale, da, db);
  CRDR_CHECK_LAUNCH("sqdiff_bwd");
  return 0;
}
extern "C" int crdr_bce_diff_bwd(const float* p, const float* q, int64_t n, float target, const float* g, float gscale,
                                 float* dp, float* dq, crdr_stream_t s) {
  CRDR_REQUIRE(p && q, "bce_diff_bwd: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(bce_diff_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(s), p, q, n, target, g, gscale, dp, dq);
  CRDR_CHECK_LAUNCH("bce_diff_bwd");
  return 0;
}

extern "C" int crdr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, int step, const float* sqnorm, float max_norm, crdr_stream_t s) {
  CRDR_REQUIRE(p && g && m && v && step >= 1, "adam_step: bad arguments");
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(s), p, g, m, v, n, lr, beta1, beta2, eps,
                     bc1, bc2_sqrt, sqnorm, max_norm);
  CRDR_CHECK_LAUNCH("adam_step");
  return 0;
}

extern "C" int crdr_adam_step_dyn(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2,
                                  float eps, const float* dyn, const float* sqnorm, float max_norm, crdr_stream_t s) {
  CRDR_REQUIRE(p && g && m && v && dyn, "adam_step_dyn: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(adam_dyn_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(s), p, g, m, v, n, beta1, beta2, eps, dyn,
                     sqnorm, max_norm);
  CRDR_CHECK_LAUNCH("adam_step_dyn");
  return 0;
}

extern "C" int crdr_maxpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, crdr_stream_t s) {
  CRDR_REQUIRE(x && y && H >= 3 && W >= 3, "maxpool: bad arguments");
  CRDR_REQUIRE((long long)N * H * W * C < (1ll << 31), "maxpool: %lld elements (32-bit element index)", (long long)N * H * W * C);
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(grid_for((int64_t)N * OH * OW * C)), dim3(256), 0, as_stream(s), x, y, N,
                     H, W, C, OH, OW);
  CRDR_CHECK_LAUNCH("maxpool_fwd");
  return 0;
}
extern "C" int crdr_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C,
                                   crdr_stream_t s) {
  CRDR_REQUIRE(x && dy && dx && H >= 3 && W >= 3, "maxpool_bwd: bad arguments");
  CRDR_REQUIRE((long long)N * H * W * C < (1ll << 31), "maxpool_bwd: %lld elements (32-bit element index)", (long long)N * H * W * C);
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, as_stream(s), x, dy, dx,
                     N, H, W, C, OH, OW);
  CRDR_CHECK_LAUNCH("maxpool_bwd");
  return 0;
}

extern "C" int crdr_lpips_layer_fwd(const float* f0, const float* f1, const float* lin, int N, int HW, int C,
                                    float* out, void* ws, size_t ws_bytes, crdr_stream_t s) {
  CRDR_REQUIRE(f0 && f1 && lin && out && ws, "lpips_layer_fwd: bad arguments");
  const int nb = std::min(cdiv(HW, 4), 64);
  CRDR_REQUIRE(ws_bytes >= (size_t)N * nb * sizeof(float), "lpips_layer_fwd: workspace too small");
  float* part = (float*)ws;
  hipLaunchKernelGGL(lpips_layer_fwd_kernel, dim3(nb, N), dim3(256), 0, as_stream(s), f0, f1, lin, N, HW, C, part);
  CRDR_CHECK_LAUNCH("lpips_layer_fwd");
  hipLaunchKernelGGL(lpips_layer_final, dim3(N), dim3(64), 0, as_stream(s), (const float*)part, nb, HW, out);
  CRDR_CHECK_LAUNCH("lpips_layer_final");
  return 0;
}
extern "C" int crdr_lpips_layer_bwd(const float* f0, const float* f1, const float* lin, int N, int HW, int C,
                                    const float* gout, float* df1, crdr_stream_t s) {
  CRDR_REQUIRE(f0 && f1 && lin && gout && df1, "lpips_layer_bwd: bad arguments");
  const int nb = std::min(cdiv(HW, 4), 256);
  hipLaunchKernelGGL(lpips_layer_bwd_kernel, dim3(nb, N), dim3(256), 0, as_stream(s), f0, f1, lin, N, HW, C, gout, df1);
  CRDR_CHECK_LAUNCH("lpips_layer_bwd");
  return 0;
}
