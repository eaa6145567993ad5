// Weight gradient of Conv2d / ConvTranspose2d on fp32 MFMA, plus the weight (re)packing kernels.
//
//   g[i][j][t] (+)= sum_{n,a,b} P[n,a,b,i] * Q[n, a*S - pad + r(t), b*S - pad + s(t), j]
//
// GEMM view per tap: rows i (channels of the dense operand P), cols j (channels of the gathered operand Q),
// reduction over pixels.  K-tile = 32 pixels; both operand tiles are staged as [32 px][channels] (exactly how
// NHWC memory is laid out) and MFMA fragments are read down the pixel axis with ds_read_b32: 32 consecutive
// channels per half-wave -> conflict free.  grid = (I-tiles x J-tiles, taps, pixel splits); every block writes
// its partial [BI][BJ] tile to a slab, wgrad_reduce sums the slabs in split order (deterministic) and scatters
// into the parameter layout [I][J][T].

#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "wgrad_args.hpp"

namespace crdr {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
static constexpr unsigned kOob = 0x80000000u;  // >= any descriptor size accepted by build_wplan (< 2 GiB)

// PREC: 0 exact fp32, 3 split-bf16 triples (CRDR_WGRAD_BF16X3), 6 fp32-equivalent split-bf16 sextuples (CRDR_WGRAD_BF16X6); the fragments run
// down the pixel axis, so the pieces are made in registers from the lane's eight ds_read_b32 values
// SQQ (CRDR_WGRAD_SQUARE_Q): the gathered operand enters SQUARED (g = sum P Q^2: the gamma gradient of a GDN layer, gdn.hip, without an x^2 tensor
// in memory); exact fp32 only, built for the configurations the GDN sizes take (CFGQ below)
template <int WM, int WN, int MB, int NB, int PREC = 0, bool SQQ = false>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_kernel(const WgradArgs p_, const WgradGroup grp) {
  constexpr bool BF3 = PREC == 3, BF6 = PREC == 6;
  static_assert(!SQQ || PREC == 0, "the squared form is exact fp32");
  constexpr int BI = 32 * WM * MB, BJ = 32 * WN * NB, NT = 64 * WM * WN;
  constexpr int PV = (8 * BI + NT - 1) / NT, QV = (8 * BJ + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sP = smem;               // [2][32][BI]
  float* sQ = smem + 2 * 32 * BI; // [2][32][BJ]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the DMA destinations and fragment bases derived from it then cost no vector instructions)
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware order (see igemm.hip): every XCD walks a contiguous range of (pixel split, tap, channel tile) so that the
  // workgroups re-reading one pixel range -- all taps and channel tiles of a split -- share an L2.
  int bx, by, bz;
  {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int nwg = gx * gy * gz, bid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int cpx = nwg >> 3;
    const int t = bid < cpx * 8 ? (bid & 7) * cpx + (bid >> 3) : bid;
    bx = t % gx;
    by = (t / gx) % gy;
    bz = t / (gx * gy);
  }
  WgradArgs p = p_;
  const int gidx = bz / p.nsplit;
  bz -= gidx * p.nsplit;
  if (p.ngroup > 1) {
    p.p = grp.p[gidx]; p.q = grp.q[gidx];
    p.ws += (size_t)gidx * p.slab_elems;
  }
  const int it = bx / p.jtiles, jt = bx % p.jtiles;
  const int i0 = it * BI, j0 = jt * BJ;
  const int tap = p.smallj ? 0 : by, split = bz;
  const int ncols = p.smallj ? p.T * 4 : p.QC;  // GEMM columns = width of a slab row
  const int t0 = (int)((long long)p.ntiles * split / p.nsplit), t1 = (int)((long long)p.ntiles * (split + 1) / p.nsplit);

  // Staging is LDS-DMA (buffer_load ... lds, 16 B per lane, lane-linear destination = exactly the [32 px][channels]
  // image): no staging registers, and pixels / channels / taps outside the tensors get an offset beyond the buffer
  // descriptor's range, for which the hardware returns zeros.
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.p), 0, p.p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.q), 0, p.q_bytes, 0x00020000);
  unsigned p_off[PV], q_coff[QV];
  int q_prow[QV], q_dh[QV], q_dw[QV];
#pragma unroll
  for (int k = 0; k < PV; ++k) {
    const int e = tid + k * NT;
    const int prow = e / (BI / 4), c = i0 + (e % (BI / 4)) * 4;
    p_off[k] = (c < p.PC) ? (unsigned)(prow * p.ldp + c) * 4u : kOob;  // rows past M fall off the end by themselves
  }
#pragma unroll
  for (int k = 0; k < QV; ++k) {
    const int e = tid + k * NT;
    const int c = j0 + (e % (BJ / 4)) * 4;
    q_prow[k] = e / (BJ / 4);
    const int tk = p.smallj ? (c >> 2) : by;  // tap of this piece
    q_dh[k] = tk / p.kw - p.pad;
    q_dw[k] = tk % p.kw - p.pad;
    q_coff[k] = p.smallj ? (tk < p.T ? 0u : kOob) : ((c < p.QC) ? (unsigned)c * 4u : kOob);
  }
  auto fetch = [&](int t, int buf) __attribute__((always_inline)) {
    const int mbase = t * 32;
    float* a = sP + buf * 32 * BI + wave * 256;
    float* b = sQ + buf * 32 * BJ + wave * 256;
    const unsigned pbase = (unsigned)(mbase * p.ldp) * 4u;
#pragma unroll
    for (int k = 0; k < PV; ++k)
      if (tid + k * NT < 8 * BI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rP, (lds_ptr_t)(a + k * NT * 4), 16, (int)(p_off[k] + pbase), 0, 0, 0);
#pragma unroll
    for (int k = 0; k < QV; ++k) {
      const int m = mbase + q_prow[k];
      const unsigned n = fdiv((unsigned)m, p.d_hw), rem = m - n * p.d_hw.d;
      const unsigned a_ = fdiv(rem, p.d_w), b_ = rem - a_ * p.d_w.d;
      const int ih = (int)a_ * p.stride + q_dh[k], iw = (int)b_ * p.stride + q_dw[k];
      const bool ok = (m < p.M) & ((unsigned)ih < (unsigned)p.QH) & ((unsigned)iw < (unsigned)p.QW);
      const unsigned off = (unsigned)(((int)n * p.QH + ih) * p.QW + iw) * (unsigned)(p.ldq * 4) + q_coff[k];
      if (tid + k * NT < 8 * BJ)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rQ, (lds_ptr_t)(b + k * NT * 4), 16, (int)(ok ? off : kOob), 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the DMA issue ahead of the MFMA stream that hides its latency
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fcol = lane & 31, fh = lane >> 5;
  const float* fa = sP + wm * MB * 32 + fcol + fh * BI;
  const float* fb = sQ + wn * NB * 32 + fcol + fh * BJ;
  auto compute = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    if constexpr (BF6) {  // fp32-equivalent split-bf16 products (common.hpp): three exact pieces per operand, six MFMAs per 16 pixels
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += 8) {
        bf16x8 ah[MB], am[MB], al[MB], bh[NB], bm[NB], bl[NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fa[buf * 32 * BI + 2 * (s0 + e) * BI + i * 32];
          split3_bf16x8(x, ah[i], am[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fb[buf * 32 * BJ + 2 * (s0 + e) * BJ + j * 32];
          split3_bf16x8(x, bh[j], bm[j], bl[j]);
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = mfma_bf16x6(ah[i], am[i], al[i], bh[j], bm[j], bl[j], acc[i][j]);
      }
      return;
    }
    if constexpr (BF3) {  // split-bf16 products (common.hpp): one bf16 MFMA covers the 16 pixels of eight fp32 steps
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += 8) {
        bf16x8 ah[MB], al[MB], bh[NB], bl[NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fa[buf * 32 * BI + 2 * (s0 + e) * BI + i * 32];
          split_bf16x8(x, ah[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fb[buf * 32 * BJ + 2 * (s0 + e) * BJ + j * 32];
          split_bf16x8(x, bh[j], bl[j]);
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = mfma_bf16x3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float af[MB], bf[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) af[i] = fa[buf * 32 * BI + 2 * s * BI + i * 32];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bf[j] = fb[buf * 32 * BJ + 2 * s * BJ + j * 32];
        if constexpr (SQQ) bf[j] *= bf[j];
      }
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (t0 < t1) fetch(t0, 0);
  __syncthreads();
  int t = t0;
  for (; t + 2 <= t1; t += 2) {
    fetch(t + 1, 1);  // t + 1 < t1 here
    compute(I0{});
    __syncthreads();
    fetch(t + 2, 0);  // may be one tile past this block's range: fetched (range-checked), never consumed
    compute(I1{});
    __syncthreads();
  }
  if (t < t1) {
    compute(I0{});
    __syncthreads();
  }

  // partial tile -> slab ws[split][tap][PC][QC]
  float* dst = p.ws + ((size_t)split * (p.smallj ? 1 : p.T) + tap) * p.PC * ncols;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + (wm * MB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (row >= p.PC) continue;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = j0 + (wn * NB + j) * 32 + fcol;
        if (col < ncols) dst[(size_t)row * ncols + col] = acc[i][j][r];
      }
    }
}

// g[(i*gJ + j)*T + t] (+)= sum_s ws[((s*T + t)*PC + i)*QC + j]
// (smallj: the slab row is [T][4] instead: ws[(s*PC + i)*4T + 4t + j])
__global__ __launch_bounds__(256) void wgrad_reduce(const float* ws, float* g, int PC, int QC, int gI, int gJ, int T,
                                                    int nsplit, int accumulate, int smallj) {
  const long long total = (long long)gI * gJ * T;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int j = (int)(e % gJ);
    const long long r = e / gJ;
    const int i = (int)(r % gI), t = (int)(r / gI);
    float v = 0.f;
    if (smallj)
      for (int s = 0; s < nsplit; ++s) v += ws[((size_t)s * PC + i) * (4 * T) + 4 * t + j];
    else
      for (int s = 0; s < nsplit; ++s) v += ws[((size_t)(s * T + t) * PC + i) * QC + j];
    float* d = g + ((size_t)i * gJ + j) * T + t;
    *d = accumulate ? *d + v : v;
  }
}

// All pending reductions of a backward pass in one launch.  Work unit ("tile") of a regular job = one output row i x 64
// input channels j x ALL T taps: the slab is read in 256-byte row segments (j contiguous), the sums go through LDS and
// leave as ONE contiguous run of 64 T floats of the parameter gradient g[i][j0 .. j0+63][0 .. T) -- both sides fully
// coalesced (a direct j-fastest mapping writes 4-byte pieces T floats apart and re-touches every output line T times).
// RGB jobs (smallj: slab rows are [T][4]) keep 256 consecutive outputs per tile.  Tile count per job (host side):
// smallj ? ceil(gI gJ T / 256) : gI * ceil(gJ / 64).  Splits are summed in a fixed order (deterministic).
// Batched kernels find the table entry of a work unit by binary search over the prefix sums.  From global memory that is a
// chain of ~log2(n) dependent loads (4-6 us) per unit -- more than the unit's own traffic takes for small layers -- so a
// workgroup first copies the prefix table (n <= kPrefixCache entries, totals < 2^31) to LDS.
constexpr int kPrefixCache = 1024;
__device__ __forceinline__ bool cache_prefix(const long long* prefix, int n, int* spre) {
  const bool cached = n + 1 <= kPrefixCache;
  if (cached) {
    for (int k = threadIdx.x; k <= n; k += blockDim.x) spre[k] = (int)prefix[k];
    __syncthreads();
  }
  return cached;
}
__device__ __forceinline__ int find_entry(const long long* prefix, const int* spre, bool cached, int n, long long g) {
  int lo = 0, hi = n - 1;  // last entry with prefix[entry] <= g  (block-uniform)
  if (cached) {
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (spre[mid] <= (int)g) lo = mid; else hi = mid - 1;
    }
  } else {
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= g) lo = mid; else hi = mid - 1;
    }
  }
  return lo;
}

constexpr int kRedMaxT = 32;
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const crdr_wgrad_job* jobs, const long long* prefix,
                                                                   const long long* meta) {
  __shared__ float tile[64 * kRedMaxT + 64];
  __shared__ int spre[kPrefixCache];
  const int n = (int)meta[0];
  const long long total = meta[1];
  const int tid = threadIdx.x;
  const bool cached = cache_prefix(prefix, n, spre);
  for (long long tl = blockIdx.x; tl < total; tl += gridDim.x) {
    const int lo = find_entry(prefix, spre, cached, n, tl);
    const crdr_wgrad_job jb = jobs[lo];
    const long long rel = tl - (cached ? (long long)spre[lo] : prefix[lo]);
    const int gJt = jb.gJtot ? jb.gJtot : jb.gJ;
    if (jb.smallj || jb.T > kRedMaxT) {
      const long long e = rel * 256 + tid;
      if (e >= (long long)jb.gI * jb.gJ * jb.T) continue;
      const int j = (int)(e % jb.gJ);
      const long long r = e / jb.gJ;
      const int i = (int)(r % jb.gI), t = (int)(r / jb.gI);
      const float* base = jb.smallj ? jb.slab + (size_t)i * (4 * jb.T) + 4 * t + j : jb.slab + ((size_t)t * jb.PC + i) * jb.QC + j;
      const size_t stride = jb.smallj ? (size_t)jb.PC * 4 * jb.T : (size_t)jb.T * jb.PC * jb.QC;
      float v = 0.f;
      for (int s = 0; s < jb.nsplit; ++s) v += base[(size_t)s * stride];
      float* d = jb.g + ((size_t)i * gJt + j) * jb.T + t;
      *d = jb.accumulate ? *d + v : v;
      continue;
    }
    const int jt = (jb.gJ + 63) >> 6;
    const int i = (int)(rel / jt), j0 = (int)(rel % jt) * 64;
    const int T = jb.T, jn = min(64, jb.gJ - j0);
    const size_t stride = (size_t)T * jb.PC * jb.QC;
    // phase 1: sums over splits in split order (four partial sums, combined (v0 + v1) + (v2 + v3): the order never changes).
    // Vector form: thread -> (tap tid >> 4, four channels 4 (tid & 15)): 16-byte loads, 16 taps per pass
    if ((jb.QC & 3) == 0 && (reinterpret_cast<uintptr_t>(jb.slab) & 15) == 0) {
      const int l16 = tid & 15;
      const bool in_row = j0 + 4 * l16 < jb.QC;  // the whole float4 lies inside the (padded) slab row
      for (int t = tid >> 4; t < T; t += 16) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
        if (in_row) {
          const float* base = jb.slab + ((size_t)t * jb.PC + i) * jb.QC + j0 + 4 * l16;
          int s = 0;
          for (; s + 4 <= jb.nsplit; s += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(base + (size_t)s * stride);
            const f32x4 b = *reinterpret_cast<const f32x4*>(base + (size_t)(s + 1) * stride);
            const f32x4 c = *reinterpret_cast<const f32x4*>(base + (size_t)(s + 2) * stride);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(base + (size_t)(s + 3) * stride);
            v0 += a; v1 += b; v2 += c; v3 += d4;
          }
          for (; s < jb.nsplit; ++s) v0 += *reinterpret_cast<const f32x4*>(base + (size_t)s * stride);
        }
        const f32x4 v = (v0 + v1) + (v2 + v3);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[(4 * l16 + e) * (T + 1) + t] = v[e];
      }
    } else {
    // scalar form: thread -> (tap group tid >> 6, channel tid & 63), four loads in flight
    const int jl = tid & 63;
    for (int t = tid >> 6; t < T; t += 4) {
      float v = 0.f;
      if (jl < jn) {
        const float* base = jb.slab + ((size_t)t * jb.PC + i) * jb.QC + j0 + jl;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int s = 0;
        for (; s + 4 <= jb.nsplit; s += 4) {
          v0 += base[(size_t)s * stride];
          v1 += base[(size_t)(s + 1) * stride];
          v2 += base[(size_t)(s + 2) * stride];
          v3 += base[(size_t)(s + 3) * stride];
        }
        for (; s < jb.nsplit; ++s) v0 += base[(size_t)s * stride];
        v = (v0 + v1) + (v2 + v3);
      }
      tile[jl * (T + 1) + t] = v;  // (T + 1): the transposed read below walks T-strided rows
    }
    }
    __syncthreads();
    // phase 2: the jn * T outputs of this tile are contiguous in g
    float* d = jb.g + ((size_t)i * gJt + j0) * T;
    const int cnt = jn * T;
    for (int e = tid; e < cnt; e += 256) {
      const int j = e / T, t = e - j * T;
      const float v = tile[j * (T + 1) + t];
      d[e] = jb.accumulate ? d[e] + v : v;
    }
    __syncthreads();
  }
}

// dst[t][rows][cols] <- src[I][J][T]
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* src, float* dst, int I, int J, int T, int rows,
                                                          int cols, int transpose) {
  const long long total = (long long)T * rows * cols;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int c = (int)(e % cols);
    const long long r2 = e / cols;
    const int r = (int)(r2 % rows), t = (int)(r2 / rows);
    const int i = transpose ? c : r, j = transpose ? r : c;
    dst[e] = (i < I && j < J) ? src[((size_t)i * J + j) * T + t] : 0.f;
  }
}

// one pack item with sub-block strides (crdr_pack_item), modes 0 / 1, any T
__global__ __launch_bounds__(256) void pack_weight_item_kernel(const crdr_pack_item it) {
  const long long total = (long long)it.T * it.rows * it.cols;
  const int sJ = it.srcJ ? it.srcJ : it.J;
  const long long dld = it.dld ? it.dld : it.cols, ts = it.tstride ? it.tstride : (long long)it.rows * it.cols;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int c = (int)(e % it.cols);
    const long long r2 = e / it.cols;
    const int r = (int)(r2 % it.rows), t = (int)(r2 / it.rows);
    const int i = it.mode ? c : r, j = it.mode ? r : c;
    it.dst[t * ts + r * dld + c] = (i < it.I && j < it.J) ? it.src[((size_t)i * sJ + j) * it.T + t] : 0.f;
  }
}

// tap-major pack for C <= 4 inputs: dst[i][t*4 + j] <- src[i][j][t]   (one "tap" of rows x cols, cols >= 4*T)
__global__ __launch_bounds__(256) void pack_weight_tapmajor_kernel(const float* src, float* dst, int I, int J, int T,
                                                                   int rows, int cols) {
  const long long total = (long long)rows * cols;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int c = (int)(e % cols), r = (int)(e / cols);
    const int t = c >> 2, j = c & 3;
    dst[e] = (r < I && j < J && t < T) ? src[((size_t)r * J + j) * T + t] : 0.f;
  }
}

// Every weight pack of one optimiser in a single launch.  Work unit = one tile of 8 pack rows x 32 pack columns x all
// T taps, staged through LDS so that both sides are coalesced: the parameter is read in runs of 32 T (mode 0) or 8 T
// (mode 1) contiguous floats, the pack is written in 128-byte row segments.
constexpr int kPackMaxT = 32;
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const crdr_pack_item* items, const long long* prefix,
                                                                   const long long* meta) {
  __shared__ float tile[256 * kPackMaxT + 32];
  __shared__ int spre[kPrefixCache];
  const int n = (int)meta[0];
  const long long total = meta[1];
  const int tid = threadIdx.x;
  const bool cached = cache_prefix(prefix, n, spre);
  for (long long g = blockIdx.x; g < total; g += gridDim.x) {
    const int lo = find_entry(prefix, spre, cached, n, g);
    const crdr_pack_item it = items[lo];
    const int tl = (int)(g - (cached ? (long long)spre[lo] : prefix[lo]));
    const int ctiles = it.cols >> 5;
    const int r0 = (tl / ctiles) * 8, c0 = (tl % ctiles) * 32;
    const int T = it.T;
    const int sJ = it.srcJ ? it.srcJ : it.J;
    const size_t dld = it.dld ? it.dld : it.cols;
    const size_t tstride = it.tstride ? (size_t)it.tstride : (size_t)it.rows * it.cols;
    // 16-byte accesses on both sides when every source run and every destination row segment is 16-byte aligned
    const bool al = ((sJ * T) & 3) == 0 && (reinterpret_cast<uintptr_t>(it.src) & 15) == 0 && (reinterpret_cast<uintptr_t>(it.dst) & 15) == 0 &&
                    (dld & 3) == 0 && (tstride & 3) == 0;
    // LDS image: mode 0 [8 r][32 c][T], mode 1 [32 c][8 r][T]; each outer row padded by one float, which makes the strided
    // reads of the write phase conflict free for the odd tap counts (1, 9, 25) these layers have
    if (it.mode == 0) {  // pack row = i, pack col = j
      const int run = 32 * T;
      const bool vec = al && c0 + 32 <= it.J;   // whole runs inside the parameter row
      if (vec) {
        for (int e4 = tid; e4 < 8 * (run >> 2); e4 += 256) {
          const int r = e4 / (run >> 2), rest = (e4 - r * (run >> 2)) << 2;
          const int i = r0 + r;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (i < it.I) v = *reinterpret_cast<const f32x4*>(it.src + ((size_t)i * sJ + c0) * T + rest);
          float* d = tile + r * (run + 1) + rest;
          d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
      } else {
        for (int e = tid; e < 8 * run; e += 256) {
          const int r = e / run, rest = e - r * run;
          const int i = r0 + r, j = c0 + rest / T;
          tile[r * (run + 1) + rest] = (i < it.I && j < it.J) ? it.src[((size_t)i * sJ + c0) * T + rest] : 0.f;
        }
      }
    } else {             // pack row = j, pack col = i
      const int run = 8 * T;
      const bool vec = al && r0 + 8 <= it.J;
      if (vec) {
        for (int e4 = tid; e4 < 32 * (run >> 2); e4 += 256) {
          const int c = e4 / (run >> 2), rest = (e4 - c * (run >> 2)) << 2;
          const int i = c0 + c;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (i < it.I) v = *reinterpret_cast<const f32x4*>(it.src + ((size_t)i * sJ + r0) * T + rest);
          float* d = tile + c * (run + 1) + rest;
          d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
      } else {
        for (int e = tid; e < 32 * run; e += 256) {
          const int c = e / run, rest = e - c * run;
          const int i = c0 + c, j = r0 + rest / T;
          tile[c * (run + 1) + rest] = (i < it.I && j < it.J) ? it.src[((size_t)i * sJ + r0) * T + rest] : 0.f;
        }
      }
    }
    __syncthreads();
    if (al) {  // thread -> (tap lane tid >> 6, row (tid >> 3) & 7, four columns 4 (tid & 7)): one 128-byte row segment per 8 lanes
      const int r = (tid >> 3) & 7, c4 = (tid & 7) << 2;
      float* dp = it.dst + (size_t)(r0 + r) * dld + c0 + c4;
      for (int t = tid >> 6; t < T; t += 4) {
        f32x4 v;
        if (it.mode == 0) {
          const float* sp = tile + r * (32 * T + 1) + c4 * T + t;
          v[0] = sp[0]; v[1] = sp[T]; v[2] = sp[2 * T]; v[3] = sp[3 * T];
        } else {
          const float* sp = tile + c4 * (8 * T + 1) + r * T + t;
          v[0] = sp[0]; v[1] = sp[8 * T + 1]; v[2] = sp[2 * (8 * T + 1)]; v[3] = sp[3 * (8 * T + 1)];
        }
        *reinterpret_cast<f32x4*>(dp + (size_t)t * tstride) = v;
      }
    } else {
      const int r = tid >> 5, c = tid & 31;
      const float* sp = it.mode == 0 ? tile + r * (32 * T + 1) + c * T : tile + c * (8 * T + 1) + r * T;
      float* dp = it.dst + ((size_t)(r0 + r)) * dld + c0 + c;
      for (int t = 0; t < T; ++t) dp[t * tstride] = sp[t];
    }
    __syncthreads();
  }
}

// scatter pack for RGB-output transposed ops: dst[4 t + j][i] <- src[i][j][t]   (rows >= 4*T, cols >= I, J <= 4)
__global__ __launch_bounds__(256) void pack_weight_scatter_kernel(const float* src, float* dst, int I, int J, int T, int rows,
                                                                  int cols) {
  const long long total = (long long)rows * cols;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int i = (int)(e % cols), r = (int)(e / cols);
    const int t = r >> 2, j = r & 3;
    dst[e] = (i < I && j < J && t < T) ? src[((size_t)i * J + j) * T + t] : 0.f;
  }
}

struct WCfg {
  int wm, wn, mb, nb;
  void (*kern)(const WgradArgs, const WgradGroup);
  void (*kern_bf3)(const WgradArgs, const WgradGroup);
  void (*kern_bf6)(const WgradArgs, const WgradGroup);
  void (*kern_sq)(const WgradArgs, const WgradGroup);   // CRDR_WGRAD_SQUARE_Q (nullptr where not built)
};
#define CFG(a, b, c, d) {a, b, c, d, wgrad_kernel<a, b, c, d, 0>, wgrad_kernel<a, b, c, d, 3>, wgrad_kernel<a, b, c, d, 6>, nullptr}
#define CFGQ(a, b, c, d) {a, b, c, d, wgrad_kernel<a, b, c, d, 0>, wgrad_kernel<a, b, c, d, 3>, wgrad_kernel<a, b, c, d, 6>, wgrad_kernel<a, b, c, d, 0, true>}
static const WCfg kWCfgs[] = {
    CFG(1, 1, 1, 1),  // 32x32
    CFGQ(2, 2, 1, 1),  // 64x64
    CFGQ(2, 2, 2, 2),  // 128x128
    CFG(2, 2, 2, 1),  // 128x64
    CFG(2, 2, 1, 2),  // 64x128
    CFGQ(4, 1, 1, 1),  // 128x32
    CFG(1, 4, 1, 1),  // 32x128
    CFG(3, 1, 1, 3),  // 96x96
    CFG(3, 1, 1, 2),  // 96x64
    CFG(2, 3, 1, 1),  // 64x96
    CFG(5, 1, 1, 2),  // 160x64
    CFG(5, 1, 1, 1),  // 160x32
    CFGQ(7, 1, 1, 1),  // 224x32
    CFG(7, 1, 1, 2),  // 224x64
    CFGQ(4, 2, 1, 1),  // 128x64 (8 waves)
    CFG(2, 4, 1, 1),  // 64x128 (8 waves)
    CFG(4, 2, 1, 2),  // 128x128 (8 waves)
    CFG(3, 2, 1, 1),  // 96x64 (6 waves)
    CFGQ(3, 3, 1, 1),  // 96x96 (9 waves)
    CFG(2, 2, 1, 1),  // dup guard (kept for index stability)
    CFG(4, 4, 1, 1),  // 128x128 (16 waves)
    CFG(2, 1, 1, 1),  // 64x32
    CFG(1, 2, 1, 1),  // 32x64
};
#undef CFG
#undef CFGQ
static const int kNumWCfgs = sizeof(kWCfgs) / sizeof(kWCfgs[0]);

struct WPlan {
  WgradArgs a;
  int cfg;
  int wino;   // 1: the Winograd F(3x3, 2x2) slab kernel, 2: the F(3x3, 4x4) one (cfg = -1)
  dim3 grid;
  size_t lds, ws_bytes;
};

static int build_wplan(const crdr_wgrad_desc* d, WPlan* pl, int G = 1) {
  WgradArgs& a = pl->a;
  memset(&a, 0, sizeof(a));
  CRDR_REQUIRE(d->PC % 4 == 0 && d->QC % 4 == 0 && d->ldp % 4 == 0 && d->ldq % 4 == 0,
               "wgrad: channel counts / strides must be multiples of 4 (PC=%d QC=%d)", d->PC, d->QC);
  CRDR_REQUIRE(d->gI <= d->PC && d->gJ <= d->QC && d->gI > 0 && d->gJ > 0, "wgrad: bad g dims");
  a.N = d->N; a.PH = d->PH; a.PW = d->PW; a.PC = d->PC; a.ldp = d->ldp;
  a.QH = d->QH; a.QW = d->QW; a.QC = d->QC; a.ldq = d->ldq;
  a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.T = d->kh * d->kw;
  const long long M64 = (long long)d->N * d->PH * d->PW;
  CRDR_REQUIRE(M64 < (1ll << 31) && (long long)d->N * d->QH * d->QW < (1ll << 31), "wgrad: too many pixels");
  a.M = (int)M64;
  {
    const long long pb = ((M64 - 1) * d->ldp + d->PC) * 4, qb = (((long long)d->N * d->QH * d->QW - 1) * d->ldq + d->QC) * 4;
    CRDR_REQUIRE(pb < (1ll << 31) && qb < (1ll << 31), "wgrad: an operand (%lld / %lld B) reaches 2 GiB; split the batch on the host", pb, qb);
    a.p_bytes = (unsigned)pb;
    a.q_bytes = (unsigned)qb;
  }
  a.ntiles = cdiv(a.M, 32);
  a.smallj = (d->QC <= 4 && a.T > 1) ? 1 : 0;
  const int ncols = a.smallj ? a.T * 4 : d->QC;   // GEMM columns per launch
  const int ntapg = a.smallj ? 1 : a.T;           // tap groups = grid.y = slabs per split
  a.d_hw = make_fastdiv((unsigned)(d->PH * d->PW));
  a.d_w = make_fastdiv((unsigned)d->PW);
  double best = 1e300; int bc = -1, bs = 1;
  const bool sqq = (d->algo & CRDR_WGRAD_SQUARE_Q) != 0;
  CRDR_REQUIRE(!sqq || !(d->algo & (CRDR_WGRAD_BF16X3 | CRDR_WGRAD_BF16X6)), "wgrad: CRDR_WGRAD_SQUARE_Q is exact fp32");
  for (int c = 0; c < kNumWCfgs; ++c) {
    const WCfg& t = kWCfgs[c];
    if (sqq && !t.kern_sq) continue;
    const int BI = 32 * t.wm * t.mb, BJ = 32 * t.wn * t.nb;
    const long long tiles = (long long)cdiv(d->PC, BI) * cdiv(ncols, BJ) * ntapg;
    const int waves_per_block = t.wm * t.wn;
    for (int ns = 1; ns <= 256; ns *= 2) {
      if (ns > 1 && a.ntiles / ns < 4) break;
      const long long blocks = tiles * ns * G;
      const double slots = 256.0 * std::max(1, 4 / waves_per_block);  // blocks that run at full MFMA rate at once
      const double per_tile = 16.0 * t.mb * t.nb * 64.0 * ((d->algo & CRDR_WGRAD_BF16X3) ? 0.3 : ((d->algo & CRDR_WGRAD_BF16X6) ? 0.5 : 1.0)) + 400.0;
      double cost = std::ceil(blocks / slots) * ((double)cdiv(a.ntiles, ns) * per_tile + 4000.0);
      cost += (double)ns * ntapg * d->PC * ncols * 4.0 / 1500.0;  // slab write + read
      if (cost < best) { best = cost; bc = c; bs = ns; }
    }
  }
  pl->wino = 0;
  if ((d->algo & 0xff) - 1 == kNumWCfgs + 1) {  // forced: the Winograd F(3x3, 4x4) slab kernel (wino4_wgrad.hip), strips of 4 tiles split 2^k ways
    const bool k3 = d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad >= 0 && d->pad <= 2;
    const bool k5s1 = d->kh == 5 && d->kw == 5 && d->stride == 1 && d->pad == 2;
    const bool k5s2 = d->kh == 5 && d->kw == 5 && d->stride == 2 && d->pad == 2;
    CRDR_REQUIRE((k3 || k5s1 || k5s2) && !a.smallj && !(d->algo & CRDR_WGRAD_BF16X3),
                 "wgrad: the Winograd F(3x3, 4x4) kernel takes 3x3 stride-1 and 5x5 (pad 2, stride 1 or 2) weight gradients with QC > 4 (exact fp32 only)");
    bs = 1 << ((d->algo >> 8) & 0xf);
    const long long strips = (long long)d->N * ((d->PH + 3) / 4) * ((d->PW + 15) / 16);
    CRDR_REQUIRE(strips / bs >= 1, "wgrad: forced split %d too deep for %lld strips", bs, strips);
    pl->wino = 2; pl->cfg = -1;
    a.nsplit = bs; a.jtiles = cdiv(d->QC, 32);
    pl->grid = dim3(cdiv(d->PC, 64) * a.jtiles * (k3 ? 1 : 4), bs, G);
    pl->lds = 0;
    a.ngroup = G;
    a.slab_elems = (long long)bs * a.T * d->PC * d->QC;
    pl->ws_bytes = (size_t)G * bs * a.T * d->PC * d->QC * sizeof(float);
    return 0;
  }
  if ((d->algo & 0xff) - 1 == kNumWCfgs) {  // forced: the Winograd F(3x3, 2x2) slab kernel (wino_wgrad.hip), strips split 2^k ways
    CRDR_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && !a.smallj && !(d->algo & CRDR_WGRAD_BF16X3) && d->pad >= 0 && d->pad <= 2,
                 "wgrad: the Winograd kernel takes 3x3 stride-1 weight gradients with QC > 4 (exact fp32 only)");
    bs = 1 << ((d->algo >> 8) & 0xf);
    const long long strips = (long long)d->N * ((d->PH + 1) / 2) * ((d->PW + 15) / 16);
    CRDR_REQUIRE(strips / bs >= 1, "wgrad: forced split %d too deep for %lld strips", bs, strips);
    pl->wino = 1; pl->cfg = -1;
    a.nsplit = bs; a.jtiles = cdiv(d->QC, 64);
    pl->grid = dim3(cdiv(d->PC, 64) * a.jtiles, bs, G);
    pl->lds = 0;
    a.ngroup = G;
    a.slab_elems = (long long)bs * a.T * d->PC * d->QC;
    pl->ws_bytes = (size_t)G * bs * a.T * d->PC * d->QC * sizeof(float);
    return 0;
  }
  if ((d->algo & 0xffff) != 0) {  // caller-forced algorithm (autotuner): (config index + 1) | log2(split) << 8
    bc = (d->algo & 0xff) - 1;
    bs = 1 << ((d->algo >> 8) & 0xf);
    CRDR_REQUIRE(bc >= 0 && bc < kNumWCfgs, "wgrad: forced config %d out of range", bc);
    CRDR_REQUIRE(!sqq || kWCfgs[bc].kern_sq, "wgrad: config %d has no CRDR_WGRAD_SQUARE_Q form", bc);
    CRDR_REQUIRE(bs == 1 || a.ntiles / bs >= 1, "wgrad: forced split %d too deep for %d pixel tiles", bs, a.ntiles);
  }
  CRDR_REQUIRE(bc >= 0, "wgrad: no tile config");
  const WCfg& t = kWCfgs[bc];
  const int BI = 32 * t.wm * t.mb, BJ = 32 * t.wn * t.nb;
  pl->cfg = bc; a.nsplit = bs; a.jtiles = cdiv(ncols, BJ);
  pl->grid = dim3(cdiv(d->PC, BI) * a.jtiles, ntapg, bs * G);
  pl->lds = (size_t)2 * 32 * (BI + BJ) * sizeof(float);
  a.ngroup = G;
  a.slab_elems = (long long)bs * ntapg * d->PC * ncols;
  pl->ws_bytes = (size_t)G * bs * ntapg * d->PC * ncols * sizeof(float);
  return 0;
}

}  // namespace crdr

using namespace crdr;

extern "C" int crdr_conv2d_wgrad_num_configs(void) { return kNumWCfgs + 1; }   // (+ the Winograd F(3x3, 2x2) slab kernel, last)
// Winograd slab kernels: ids crdr_conv2d_wgrad_num_configs() (F(3x3, 2x2)) .. + crdr_conv2d_wgrad_num_wino_configs() - 1 (F(3x3, 4x4))
extern "C" int crdr_conv2d_wgrad_num_wino_configs(void) { return 2; }

extern "C" size_t crdr_conv2d_wgrad_workspace(const crdr_wgrad_desc* d) {
  WPlan pl;
  if (build_wplan(d, &pl)) return 0;
  return pl.ws_bytes;
}

static int launch_wgrad_slabs(const crdr_wgrad_desc* d, const float* const* ps, const float* const* qs, int G, void* ws,
                              size_t ws_bytes, WPlan& pl, crdr_stream_t s) {
  if (int rc = build_wplan(d, &pl, G)) return rc;
  CRDR_REQUIRE(ws, "wgrad: null pointer");
  CRDR_REQUIRE(pl.ws_bytes <= ws_bytes, "wgrad: workspace too small (%zu < %zu)", ws_bytes, pl.ws_bytes);
  WgradArgs& a = pl.a;
  WgradGroup grp;
  memset(&grp, 0, sizeof(grp));
  for (int g = 0; g < G; ++g) {
    CRDR_REQUIRE(ps[g] && qs[g], "wgrad: null operand (problem %d)", g);
    grp.p[g] = ps[g]; grp.q[g] = qs[g];
  }
  a.p = ps[0]; a.q = qs[0]; a.ws = (float*)ws;
  if (pl.wino == 2) {
    wino4_wgrad_launch(a, grp, pl.grid, as_stream(s));
    CRDR_CHECK_LAUNCH("wino4_wgrad_kernel");
    return 0;
  } else if (pl.wino) {
    wino_wgrad_launch(a, grp, pl.grid, as_stream(s));
    CRDR_CHECK_LAUNCH("wino_wgrad_kernel");
    return 0;
  }
  const WCfg& t = kWCfgs[pl.cfg];
  CRDR_REQUIRE(!((d->algo & CRDR_WGRAD_BF16X3) && (d->algo & CRDR_WGRAD_BF16X6)), "wgrad: CRDR_WGRAD_BF16X3 and CRDR_WGRAD_BF16X6 are exclusive");
  const int bf3 = (d->algo & CRDR_WGRAD_BF16X3) ? 1 : ((d->algo & CRDR_WGRAD_BF16X6) ? 2 : 0);
  const bool sq = (d->algo & CRDR_WGRAD_SQUARE_Q) != 0;
  CRDR_REQUIRE(!sq || pl.wino == 0, "wgrad: CRDR_WGRAD_SQUARE_Q with a Winograd id");
  auto kern = sq ? t.kern_sq : (bf3 == 1 ? t.kern_bf3 : (bf3 == 2 ? t.kern_bf6 : t.kern));
  static std::atomic<bool> attr_done[4][64];
  if (!attr_done[sq ? 3 : bf3][pl.cfg].load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[sq ? 3 : bf3][pl.cfg].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, pl.grid, dim3(64 * t.wm * t.wn), pl.lds, as_stream(s), a, grp);
  CRDR_CHECK_LAUNCH("wgrad_kernel");
  return 0;
}

extern "C" int crdr_conv2d_wgrad(const crdr_wgrad_desc* d, const float* p, const float* q, float* g, void* ws,
                                 size_t ws_bytes, crdr_stream_t s) {
  CRDR_REQUIRE(g, "wgrad: null pointer");
  WPlan pl;
  void* prof = profile_begin(as_stream(s));
  if (int rc = launch_wgrad_slabs(d, &p, &q, 1, ws, ws_bytes, pl, s)) return rc;
  const WgradArgs& a = pl.a;
  const long long total = (long long)d->gI * d->gJ * a.T;
  const int blocks = (int)std::min<long long>(cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(wgrad_reduce, dim3(blocks), dim3(256), 0, as_stream(s), (const float*)ws, g, d->PC, d->QC, d->gI,
                     d->gJ, a.T, a.nsplit, d->accumulate, a.smallj);
  CRDR_CHECK_LAUNCH("wgrad_reduce");
  profile_end(pl.wino == 2 ? (d->kh == 5 ? 8 : 7) : pl.wino ? 4 : 1, 2.0 * (double)a.M * d->gI * d->gJ * a.T, prof, as_stream(s));
  return 0;
}

extern "C" int crdr_conv2d_wgrad_partial(const crdr_wgrad_desc* d, const float* p, const float* q, float* g, void* slab,
                                         size_t slab_bytes, crdr_wgrad_job* job, crdr_stream_t s) {
  CRDR_REQUIRE(g && job, "wgrad_partial: null pointer");
  WPlan pl;
  void* prof = profile_begin(as_stream(s));
  if (int rc = launch_wgrad_slabs(d, &p, &q, 1, slab, slab_bytes, pl, s)) return rc;
  const WgradArgs& a = pl.a;
  job->slab = (const float*)slab; job->g = g;
  job->PC = d->PC; job->QC = d->QC; job->gI = d->gI; job->gJ = d->gJ; job->T = a.T; job->nsplit = a.nsplit;
  job->smallj = a.smallj; job->accumulate = d->accumulate; job->gJtot = 0; job->reserved = 0;
  profile_end(pl.wino == 2 ? (d->kh == 5 ? 8 : 7) : pl.wino ? 4 : 1, 2.0 * (double)a.M * d->gI * d->gJ * a.T, prof, as_stream(s));
  return 0;
}

extern "C" size_t crdr_conv2d_wgrad_grouped_workspace(const crdr_wgrad_desc* d, int G) {
  WPlan pl;
  if (G < 1 || G > CRDR_MAX_GROUP || build_wplan(d, &pl, G)) return 0;
  return pl.ws_bytes;
}

extern "C" int crdr_conv2d_wgrad_partial_grouped(const crdr_wgrad_desc* d, const float* const* ps, const float* const* qs,
                                                 float* const* gs, int G, void* slab, size_t slab_bytes, crdr_wgrad_job* jobs,
                                                 crdr_stream_t s) {
  CRDR_REQUIRE(ps && qs && gs && jobs, "wgrad_partial_grouped: null pointer");
  CRDR_REQUIRE(G >= 1 && G <= CRDR_MAX_GROUP, "wgrad_partial_grouped: %d problems (1..%d)", G, CRDR_MAX_GROUP);
  WPlan pl;
  void* prof = profile_begin(as_stream(s));
  if (int rc = launch_wgrad_slabs(d, ps, qs, G, slab, slab_bytes, pl, s)) return rc;
  const WgradArgs& a = pl.a;
  for (int g = 0; g < G; ++g) {
    crdr_wgrad_job* job = jobs + g;
    CRDR_REQUIRE(gs[g], "wgrad_partial_grouped: null gradient (problem %d)", g);
    job->slab = (const float*)slab + (size_t)g * a.slab_elems; job->g = gs[g];
    job->PC = d->PC; job->QC = d->QC; job->gI = d->gI; job->gJ = d->gJ; job->T = a.T; job->nsplit = a.nsplit;
    job->smallj = a.smallj; job->accumulate = d->accumulate; job->gJtot = 0; job->reserved = 0;
  }
  profile_end(pl.wino == 2 ? (d->kh == 5 ? 8 : 7) : pl.wino ? 4 : 1, 2.0 * (double)G * a.M * d->gI * d->gJ * a.T, prof, as_stream(s));
  return 0;
}

extern "C" int crdr_wgrad_reduce_batched(const crdr_wgrad_job* jobs, const int64_t* prefix, const int64_t* meta,
                                         crdr_stream_t s) {
  CRDR_REQUIRE(jobs && prefix && meta, "wgrad_reduce_batched: null pointer");
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(16384), dim3(256), 0, as_stream(s), jobs,
                     reinterpret_cast<const long long*>(prefix), reinterpret_cast<const long long*>(meta));
  CRDR_CHECK_LAUNCH("wgrad_reduce_batched_kernel");
  return 0;
}

extern "C" int crdr_pack_weights_batched(const crdr_pack_item* items, const int64_t* prefix, const int64_t* meta,
                                         crdr_stream_t s) {
  CRDR_REQUIRE(items && prefix && meta, "pack_weights_batched: null pointer");
  hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(2048), dim3(256), 0, as_stream(s), items,
                     reinterpret_cast<const long long*>(prefix), reinterpret_cast<const long long*>(meta));
  CRDR_CHECK_LAUNCH("pack_weights_batched_kernel");
  return 0;
}

extern "C" int crdr_pack_weight_item(const crdr_pack_item* item, crdr_stream_t s) {
  CRDR_REQUIRE(item && item->src && item->dst, "pack_weight_item: null pointer");
  CRDR_REQUIRE(item->mode == 0 || item->mode == 1, "pack_weight_item: mode %d", item->mode);
  CRDR_REQUIRE(item->rows >= (item->mode ? item->J : item->I) && item->cols >= (item->mode ? item->I : item->J),
               "pack_weight_item: pack smaller than source");
  const long long total = (long long)item->T * item->rows * item->cols;
  hipLaunchKernelGGL(pack_weight_item_kernel, dim3((int)std::min<long long>(cdiv64(total, 256), 8192)), dim3(256), 0,
                     as_stream(s), *item);
  CRDR_CHECK_LAUNCH("pack_weight_item");
  return 0;
}

extern "C" int crdr_pack_weight(const float* src, float* dst, int I, int J, int T, int rows, int cols, int transpose,
                                crdr_stream_t s) {
  CRDR_REQUIRE(src && dst, "pack_weight: null pointer");
  if (transpose == 2) {
    CRDR_REQUIRE(J <= 4 && rows >= I && cols >= 4 * T, "pack_weight: tap-major pack needs J <= 4, rows >= I, cols >= 4*T");
    const long long tot = (long long)rows * cols;
    hipLaunchKernelGGL(pack_weight_tapmajor_kernel, dim3((int)std::min<long long>(cdiv64(tot, 256), 8192)), dim3(256), 0,
                       as_stream(s), src, dst, I, J, T, rows, cols);
    CRDR_CHECK_LAUNCH("pack_weight_tapmajor");
    return 0;
  }
  if (transpose == 3) {
    CRDR_REQUIRE(J <= 4 && rows >= 4 * T && cols >= I, "pack_weight: scatter pack needs J <= 4, rows >= 4*T, cols >= I");
    const long long tot = (long long)rows * cols;
    hipLaunchKernelGGL(pack_weight_scatter_kernel, dim3((int)std::min<long long>(cdiv64(tot, 256), 8192)), dim3(256), 0,
                       as_stream(s), src, dst, I, J, T, rows, cols);
    CRDR_CHECK_LAUNCH("pack_weight_scatter");
    return 0;
  }
  CRDR_REQUIRE(rows >= (transpose ? J : I) && cols >= (transpose ? I : J), "pack_weight: pack smaller than source");
  const long long total = (long long)T * rows * cols;
  const int blocks = (int)std::min<long long>(cdiv64(total, 256), 8192);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(s), src, dst, I, J, T, rows, cols,
                     transpose);
  CRDR_CHECK_LAUNCH("pack_weight");
  return 0;
}
