// Shared helpers for the gfx950 kernels of libcrdr_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "crdr_hip.h"

namespace crdr {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(crdr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define CRDR_CHECK_LAUNCH(what)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      crdr::set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
      return -2;                                                             \
    }                                                                        \
  } while (0)

#define CRDR_REQUIRE(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      crdr::set_error(__VA_ARGS__);  \
      return -1;                     \
    }                                \
  } while (0)

// Optional launch timing (bench.py): when enabled, conv / wgrad launches are bracketed by HIP events on the launch
// stream and (flops, events) are kept until crdr_profile_read.  Off by default; never used under graph capture.
bool profile_on();
void* profile_begin(hipStream_t s);
void profile_end(int kind, double flops, void* token, hipStream_t s);

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// LDS tile image shared by the GEMM kernels: rows of 32 floats (128 B) cut in eight 16-B chunks; chunk c of
// row r lives at chunk slot c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads (16-lane groups
// over 16 different rows, same chunk) conflict free.
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }

// ---- split-bf16 ("bf16x3") operands for the bf16 matrix path: x = hi + lo + eps with hi = bf16(x) (round to nearest even),
// lo = bf16(x - hi), |eps| <= 2^-16 |x|.  a b ~= ah bh + ah bl + al bh (the dropped al bl term is <= 2^-16 |a b| too): three
// v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulation -- per-product relative error <= 3 * 2^-16 at 3/16 of the exact-fp32
// MFMA cost.  Opt-in (CRDR_CONV_BF16X3); the default path stays exact fp32.
__device__ __forceinline__ void split_bf16x2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f32x2 x = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));   // v_cvt_pk_bf16_f32: low half = bf16(x0)
  const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
  const f32x2 r = {x0 - h0, x1 - h1};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split_bf16x2(x[0], x[1], h0, l0);
  split_bf16x2(x[2], x[3], h1, l1);
  split_bf16x2(x[4], x[5], h2, l2);
  split_bf16x2(x[6], x[7], h3, l3);
  const u32x4_t h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
// acc += a b for one 32x32 block over 16 k (small terms first)
__device__ __forceinline__ f32x16 mfma_bf16x3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}


// ---- "bf16x6": fp32-equivalent products on the bf16 matrix path.  x = hi + mid + lo EXACTLY (hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid:
// three 8-bit pieces of the 24-bit significand; both residuals are exact in fp32 and the last one is a bf16 number), and
// a b = (ah + am + al)(bh + bm + bl) is evaluated as the six products of weight >= 2^-16, small terms first:  am bm, al bh, ah bl, am bh, ah bm,
// ah bh.  Every bf16 x bf16 product is exact in the fp32 accumulator; dropped are am bl, al bm (<= 2^-24 |a b| each) and al bl (2^-32):
// per-product relative error <= ~2^-23, the size of one fp32 rounding -- what the exact-fp32 MFMA chain pays per accumulation step anyway.
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16 k against eight v_mfma_f32_32x32x2_f32 (64 cycles each): 3/8 of the matrix time.
// Opt-in (CRDR_CONV_BF16X6 / CRDR_WGRAD_BF16X6); the default path stays the exact fp32 instruction.
__device__ __forceinline__ void split3_bf16x2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 x = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const f32x2 h = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
  const f32x2 r = x - h;   // exact: x and bf16(x) share their leading 8 bits
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const f32x2 m = {__builtin_bit_cast(float, mid << 16), __builtin_bit_cast(float, mid & 0xffff0000u)};
  const f32x2 r2 = r - m;  // exact, and at most 8 significant bits are left: the conversion below does not round
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}
__device__ __forceinline__ void split3_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4_t h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned a, b, c;
    split3_bf16x2(x[2 * e], x[2 * e + 1], a, b, c);
    h[e] = a; m[e] = b; l[e] = c;
  }
  hi = __builtin_bit_cast(bf16x8, h);
  mid = __builtin_bit_cast(bf16x8, m);
  lo = __builtin_bit_cast(bf16x8, l);
}
// acc += a b for one 32x32 block over 16 k (small terms first)
__device__ __forceinline__ f32x16 mfma_bf16x6(const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh, const bf16x8 bm, const bf16x8 bl,
                                              f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

}  // namespace crdr
