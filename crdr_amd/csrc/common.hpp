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

}  // namespace crdr
