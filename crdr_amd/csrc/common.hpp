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

// LDS tile image shared by the GEMM kernels: rows of 32 floats (128 B) cut in eight 16-B chunks; chunk c of
// row r lives at chunk slot c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads (16-lane groups
// over 16 different rows, same chunk) conflict free.
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }

}  // namespace crdr
