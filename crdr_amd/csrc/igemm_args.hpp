// Kernel-argument structs of the implicit-GEMM family, shared by igemm.hip (tiled kernels, host planning) and
// gemm1x1.hip (streaming 1x1 kernel).
#pragma once

#include "common.hpp"

namespace crdr {

struct IgemmArgs {
  const float* x;
  const float* w;
  float* y;
  float* ws;
  const float* bias;
  const float* vec2;
  const float* res;
  const float* scale;
  const float* shift;
  const float* gx;
  const float* gt;
  float* sig;
  const float* pre;
  const float* mask;
  int* counters;   // split-K: one ticket per (group, phase, M tile, N tile), zero between launches (head of the workspace)
  float* cs;       // CRDR_EPI_COLSUM: per-tile partial column sums [rows][2][cs_ld] (pre-mask, post-mask)
  int ldpre, ldmask;
  int cs_ld, cs_rows;
  int ngroup;
  int N, H, W, Cin, ldx;
  int GH, GW, so, OH, OW, ldy, Cout;
  int si;
  int wrows, wcols;
  int ldres, ldg;
  int flags;
  int M;  // rows per phase = N*GH*GW
  int nphase, nsplit;
  int kchunks;
  int ws_ld;  // columns of a partial slab row (= gridDim.y * BN)
  int hw_sh, gw_sh;  // log2(GH * GW), log2(GW) when both are powers of two (every layer of the model but the 15 / 31 / 63-pixel ones), else -1:
                     // the tiled kernel then splits a GEMM row into (image, grid row, grid column) with shifts instead of integer divisions
  int vec_epi;  // 1: y / res / gx / gt / sig rows are 16-byte aligned -> vector epilogue
  int m_inner;   // 1: consecutive workgroups walk the M tiles of one (N tile, phase/split) -- weight-heavy shapes (see build_plan)
  int fast_epi;  // 1: vec_epi, Cout % 4 == 0, unsplit, no gate / pre-add / accumulate: straight-line buffer-op epilogue
  int k_cmajor;  // 1: K loop walks (channel chunk, tap) instead of (tap, channel chunk), see the kernel
  int smallc;   // 1: Cin <= 4 and the weight pack is tap-major ([rows][taps*4]): a K-tile covers 8 taps x 4 channels
  unsigned long long x_bytes;  // extent of the input tensor (the descriptor is re-based per workgroup, see kernel)
  unsigned w_bytes;            // extent of the weight pack's buffer descriptor
};

// Tap / phase tables travel as a second by-value kernel argument that is only ever indexed with wave-uniform
// indices in the kernel prologue (keeps the scalar argument block above in SGPRs).
struct IgemmTaps {
  int packed[128];  // (dh & 0xff) | (dw & 0xff) << 8 | widx << 16
  short tap_begin[17];
  int8_t poh[16], pow[16];
};

// Per-problem pointers of a grouped launch (crdr_conv2d_grouped); a plain launch is a group of one.  Only ever indexed
// with the workgroup-uniform problem index.
struct IgemmGroup {
  const float* x[CRDR_MAX_GROUP];
  const float* w[CRDR_MAX_GROUP];
  float* y[CRDR_MAX_GROUP];
  const float* bias[CRDR_MAX_GROUP];
  const float* pre[CRDR_MAX_GROUP];
  const float* mask[CRDR_MAX_GROUP];
  const float* res[CRDR_MAX_GROUP];
  float* cs[CRDR_MAX_GROUP];
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
static constexpr unsigned kOobOffset = 0x80000000u;  // >= any descriptor size accepted by build_plan (< 2 GiB)

// streaming 1x1 kernel (gemm1x1.hip)
struct StreamArgs {
  int gridN;   // N tiles
  int nlanes;  // M-tile lanes per problem (multiple of 8): lane l owns M tiles l, l + nlanes, ...
};
int stream_num_variants();
void stream_variant_shape(int v, int* nb, int* stages, int* nw);
void stream_launch(int v, const IgemmArgs& a, const StreamArgs& sa, const IgemmGroup& grp, unsigned grid, size_t lds, hipStream_t s);

}  // namespace crdr
