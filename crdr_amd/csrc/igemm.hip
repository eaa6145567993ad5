// Implicit-GEMM convolution: host side (geometry, plan, launch).  The kernel template lives in igemm_kernel.hpp and is
// instantiated in igemm_p*.hip.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "igemm_kernel.hpp"
#include "wino.hpp"

namespace crdr {

// BEGIN GENERATED (gen_igemm_parts.py)
extern template __global__ void igemm_kernel<4, 1, 1, 1, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 2, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 3, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 4, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 5, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 6, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 7, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 2, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 3, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 4, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 2, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 1, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 2, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 3, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 4, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<1, 4, 1, 1, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<1, 4, 1, 2, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 1, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 2, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 3, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 2, 1, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 2, 2, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 1, 1, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 2, 1, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 1, 2, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 1, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 3, false, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 1, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 2, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 2, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 2, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 1, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 2, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 2, 1, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 2, 2, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 2, 1, true, 0, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 1, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 2, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 3, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 4, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 5, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 6, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 7, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 2, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 3, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 4, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 2, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 1, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 2, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 3, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 4, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<1, 4, 1, 1, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<1, 4, 1, 2, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 1, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 2, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 3, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 2, 1, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 2, 2, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 1, 1, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 2, 1, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 1, 2, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 1, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 3, false, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 2, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 4, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 2, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 1, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 2, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 3, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 2, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 2, 1, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 1, false, 6, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 1, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 2, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 3, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 4, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 5, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 6, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 1, 7, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 2, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 3, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 1, 2, 4, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 2, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 1, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 2, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 3, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 1, 4, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<1, 4, 1, 1, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<1, 4, 1, 2, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 1, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 2, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 1, 3, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 2, 1, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<4, 2, 2, 2, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 1, 1, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 2, 1, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 4, 1, 2, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 1, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
extern template __global__ void igemm_kernel<2, 2, 2, 3, false, 0, true>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
// END GENERATED

// ------------------------------------------------------------------------------------------------------------
// host side: geometry, config choice, launch
// ------------------------------------------------------------------------------------------------------------
struct TileCfg {
  int wm, wn, mb, nb;
  void (*kern)(const IgemmArgs, const IgemmTaps, const IgemmGroup);
  void (*kern_smallc)(const IgemmArgs, const IgemmTaps, const IgemmGroup);  // tap-major variant (Cin <= 4), nullptr where not built
  void (*kern_bf3)(const IgemmArgs, const IgemmTaps, const IgemmGroup);     // split-bf16 products (CRDR_CONV_BF16X3)
  void (*kern_fast)(const IgemmArgs, const IgemmTaps, const IgemmGroup);    // unsplit launches on the straight-line epilogue: the kernel without its split-K / general-epilogue code
  void (*kern_bf6)(const IgemmArgs, const IgemmTaps, const IgemmGroup);     // fp32-equivalent split-bf16 products (CRDR_CONV_BF16X6), nullptr where the stages do not fit LDS
};
// LDS bytes of a tile configuration (K-loop staging + tap table overlaid by the epilogue's transposed accumulators + column sums, then the
// per-column vectors sV at kSvOff in the kernel, + the reducer flag of a split launch)
static constexpr size_t cfg_lds(int wm, int wn, int mb, int nb, int prec) {
  const size_t BM = 32 * wm * mb, BN = 32 * wn * nb;
  const size_t staging = igemm_staging_floats((int)BM, (int)BN, prec);
  const size_t epi = (size_t)wm * wn * 32 * 32 * (nb < 4 ? nb : 4) + (size_t)wm * 2 * BN;
  const size_t sv_off = ((staging > epi ? staging : epi) + 3) & ~(size_t)3;
  return (sv_off + (size_t)4 * BN + 4) * sizeof(float);
}
template <int a, int b, int c, int d, bool FITS = igemm_bf6_ok(a, b, c, d) && (cfg_lds(a, b, c, d, 6) <= 160 * 1024)>
struct Bf6Kern { static constexpr void (*fn)(const IgemmArgs, const IgemmTaps, const IgemmGroup) = igemm_kernel<a, b, c, d, false, 6>; };
template <int a, int b, int c, int d>
struct Bf6Kern<a, b, c, d, false> { static constexpr void (*fn)(const IgemmArgs, const IgemmTaps, const IgemmGroup) = nullptr; };
#define CFG(a, b, c, d) {a, b, c, d, igemm_kernel<a, b, c, d, false, 0>, nullptr, igemm_kernel<a, b, c, d, false, 3>, igemm_kernel<a, b, c, d, false, 0, true>, Bf6Kern<a, b, c, d>::fn}
#define CFGS(a, b, c, d) {a, b, c, d, igemm_kernel<a, b, c, d, false, 0>, igemm_kernel<a, b, c, d, true, 0>, igemm_kernel<a, b, c, d, false, 3>, igemm_kernel<a, b, c, d, false, 0, true>, Bf6Kern<a, b, c, d>::fn}
static const TileCfg kCfgs[] = {
    // BM=128 family (one 32-row strip per wave), BN = 32..224
    CFGS(4, 1, 1, 1), CFGS(4, 1, 1, 2), CFG(4, 1, 1, 3), CFG(4, 1, 1, 4), CFG(4, 1, 1, 5), CFG(4, 1, 1, 6),
    CFG(4, 1, 1, 7),
    // BM=256: two strips per wave
    CFGS(4, 1, 2, 2), CFG(4, 1, 2, 3), CFG(4, 1, 2, 4),
    // 2x2 waves
    CFGS(2, 2, 2, 2),  // 128x128
    CFG(2, 2, 1, 1),  // 64x64
    CFG(2, 2, 1, 2),  // 64x128
    CFG(2, 2, 1, 3),  // 64x192
    CFG(2, 2, 1, 4),  // 64x256
    // BM=32: small-M layers
    CFG(1, 4, 1, 1),  // 32x128
    CFG(1, 4, 1, 2),  // 32x256
    // 8-wave blocks: two waves per SIMD share one staged tile (more MFMA work per staged byte, latency hiding)
    CFGS(4, 2, 1, 1),  // 128x64
    CFGS(4, 2, 1, 2),  // 128x128
    CFG(4, 2, 1, 3),  // 128x192
    CFGS(4, 2, 2, 1),  // 256x64
    CFGS(4, 2, 2, 2),  // 256x128
    CFG(2, 4, 1, 1),  // 64x128
    CFGS(2, 4, 2, 1),  // 128x128
    CFG(2, 4, 1, 2),  // 64x256
    CFG(2, 2, 2, 1),  // 128x64 (4 waves)
    CFG(2, 2, 2, 3),  // 128x192 (4 waves)
};
#undef CFG
#undef CFGS
static const int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

struct Plan {
  IgemmArgs a;
  IgemmTaps t;
  int cfg;
  int stream;  // index into kStreamCfgs, or -1: the tiled kernel
  int wino;    // 1 + variant: the Winograd kernel (wino.hip); cfg and stream are -1
  StreamArgs sa;
  dim3 grid;
  size_t lds;
  size_t ws_bytes;
};

// split-K plans: tickets fit the workspace head and one buffer descriptor spans all slabs of a (group, phase)
static bool splitk_ok(long long tiles, int M, int ws_ld, int ns) {
  return tiles <= CRDR_CONV_TICKETS && (long long)ns * M * ws_ld * 4 < (1ll << 31) - (1 << 24);
}

static int floordiv(int a, int b) {
  int q = a / b, r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

// Measured large-M throughput of every tile configuration on an unpadded 3x3 layer (TFLOP/s, tools/sweep_conv.py --dump on
// k3_128to{64,96,128,192,256,320}): what the built-in cost model knows about occupancy, LDS traffic and epilogue cost.
static const float kCfgTflops[] = {107, 120, 122, 120, 105, 99, 94, 90, 101, 106, 102, 116, 123, 120, 107, 109, 111, 115, 126, 121,
                                   111, 116, 119, 127, 114, 119, 108};

// fallback = true: the caller found unaligned operands after the queries answered for the streaming kernel: plan an unsplit
// tiled launch (no workspace, never the streaming kernel)
static int build_plan(const crdr_conv_desc* d, Plan* pl, int G = 1, bool fallback = false) {
  IgemmArgs& a = pl->a;
  IgemmTaps& tp = pl->t;
  memset(&a, 0, sizeof(a));
  memset(&tp, 0, sizeof(tp));
  auto pack_tap = [](int dh, int dw, int widx) { return (dh & 0xff) | ((dw & 0xff) << 8) | (widx << 16); };
  CRDR_REQUIRE(d->kh * d->kw <= 128, "conv2d: kernel %dx%d has more than 128 taps", d->kh, d->kw);
  CRDR_REQUIRE(d->stride >= 1 && d->stride <= 4, "conv2d: stride %d unsupported", d->stride);
  CRDR_REQUIRE(d->C % 4 == 0 && d->ldx % 4 == 0, "conv2d: C (%d) and ldx (%d) must be multiples of 4", d->C, d->ldx);
  CRDR_REQUIRE(d->wcols % 32 == 0 && d->wcols >= d->C && d->wrows >= d->OC, "conv2d: bad weight pack %dx%d", d->wrows,
               d->wcols);
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->C; a.ldx = d->ldx;
  a.OH = d->OH; a.OW = d->OW; a.ldy = d->ldy; a.Cout = d->OC;
  a.wrows = d->wrows; a.wcols = d->wcols;
  a.ldres = d->ldres; a.ldg = d->ldg; a.flags = d->flags;
  a.ldpre = d->ldpre; a.ldmask = d->ldmask; a.ngroup = G;
  const bool want_cs = (d->flags & CRDR_EPI_COLSUM) != 0;
  CRDR_REQUIRE(G >= 1 && G <= CRDR_MAX_GROUP, "conv2d: group of %d problems (max %d)", G, CRDR_MAX_GROUP);
  a.smallc = d->wlayout == 1;
  CRDR_REQUIRE(!a.smallc || (!d->transposed && d->C <= 4 && d->wcols >= 4 * d->kh * d->kw),
               "conv2d: tap-major weight layout needs a non-transposed conv with C <= 4 and wcols >= 4*taps");
  a.kchunks = a.smallc ? cdiv(d->kh * d->kw, 8) : cdiv(d->C, 32);
  const int S = d->stride, P = d->pad;
  int nt = 0;
  if (!d->transposed) {
    a.nphase = 1; a.GH = d->OH; a.GW = d->OW; a.so = 1; a.si = S;
    tp.poh[0] = 0; tp.pow[0] = 0; tp.tap_begin[0] = 0;
    for (int r = 0; r < d->kh; ++r)
      for (int s = 0; s < d->kw; ++s) {
        tp.packed[nt++] = pack_tap(r - P, s - P, r * d->kw + s);
      }
    tp.tap_begin[1] = (short)nt;
  } else {
    a.nphase = S * S; a.GH = cdiv(d->OH, S); a.GW = cdiv(d->OW, S); a.so = S; a.si = 1;
    int ph = 0;
    for (int py = 0; py < S; ++py)
      for (int px = 0; px < S; ++px, ++ph) {
        tp.poh[ph] = (int8_t)py; tp.pow[ph] = (int8_t)px; tp.tap_begin[ph] = (short)nt;
        for (int r = 0; r < d->kh; ++r) {
          if (((py + P - r) % S + S) % S) continue;
          for (int s = 0; s < d->kw; ++s) {
            if (((px + P - s) % S + S) % S) continue;
            tp.packed[nt++] = pack_tap(floordiv(py + P - r, S), floordiv(px + P - s, S), r * d->kw + s);
          }
        }
      }
    tp.tap_begin[ph] = (short)nt;
  }
  {  // power-of-two grids: the kernel splits GEMM rows with shifts (igemm_kernel.hpp, split_row)
    const long long hw = (long long)a.GH * a.GW;
    const bool p2 = a.GW > 0 && hw > 0 && hw < (1ll << 30) && (a.GW & (a.GW - 1)) == 0 && (hw & (hw - 1)) == 0;
    a.gw_sh = p2 ? __builtin_ctz((unsigned)a.GW) : -1;
    a.hw_sh = p2 ? __builtin_ctzll((unsigned long long)hw) : -1;
  }
  {  // extents of the range-checked buffer descriptors (the input's is re-based per workgroup: any size)
    const long long xb = (((long long)d->N * d->H * d->W - 1) * d->ldx + d->C) * 4;
    const long long wb = (long long)(a.smallc ? 1 : d->kh * d->kw) * d->wrows * d->wcols * 4;
    CRDR_REQUIRE(wb < (1ll << 31), "conv2d: weight pack (%lld B) reaches 2 GiB", wb);
    CRDR_REQUIRE((long long)d->W * d->ldx * 4 * (d->kh + 2) < (1ll << 30), "conv2d: an image row (%d px x %d floats) is too wide", d->W, d->ldx);
    a.x_bytes = (unsigned long long)xb;
    a.w_bytes = (unsigned)wb;
  }
  const long long M64 = (long long)d->N * a.GH * a.GW;
  CRDR_REQUIRE(M64 < (1ll << 31) && (long long)d->N * d->H * d->W < (1ll << 31), "conv2d: too many pixels");
  a.M = (int)M64;

  // ---- choose tile config + split-K with a small cost model (MFMA cycles per block x waves of blocks)
  int maxtaps = 0;
  for (int ph = 0; ph < a.nphase; ++ph) maxtaps = std::max(maxtaps, (int)(tp.tap_begin[ph + 1] - tp.tap_begin[ph]));
  const int KT = a.smallc ? a.kchunks : maxtaps * a.kchunks;
  double best = 1e300; int bc = -1, bs = 1;
  const bool bf6 = (d->flags & CRDR_CONV_BF16X6) && !a.smallc;
  CRDR_REQUIRE(!((d->flags & CRDR_CONV_BF16X6) && (d->flags & CRDR_CONV_BF16X3)), "conv2d: CRDR_CONV_BF16X3 and CRDR_CONV_BF16X6 are exclusive");
  for (int c = 0; c < kNumCfgs; ++c) {
    const TileCfg& t = kCfgs[c];
    if (a.smallc && !t.kern_smallc) continue;
    if (bf6 && !t.kern_bf6) continue;
    const int BM = 32 * t.wm * t.mb, BN = 32 * t.wn * t.nb;
    const long long tiles = (long long)cdiv(a.M, BM) * cdiv(d->OC, BN) * a.nphase;
    static_assert(sizeof(kCfgTflops) / sizeof(kCfgTflops[0]) == sizeof(kCfgs) / sizeof(kCfgs[0]), "one figure per configuration");
    for (int ns = 1; ns <= ((fallback || (d->flags & CRDR_CONV_NOSPLIT)) ? 1 : 16); ns *= 2) {
      if (ns > 1 && KT / ns < 4) break;
      if (ns > 1 && !splitk_ok(tiles * G, a.M, cdiv(d->OC, BN) * BN, ns)) break;
      const long long blocks = tiles * ns * G;
      // MFMA cycles of one workgroup per K-tile (waves beyond four share the SIMDs), scaled by the measured efficiency
      // (split-bf16 products: 6 MFMAs of 32 cycles per block and K-tile instead of 16 of 64, plus the operand splitting)
      // (bf16x6: 12 MFMAs of 32 cycles per block and K-tile, the operand splitting between them)
      const double per_iter = 16.0 * t.mb * t.nb * 64.0 * std::max(1.0, t.wm * t.wn / 4.0) * (133.0 / kCfgTflops[c]) *
                              ((d->flags & CRDR_CONV_BF16X3) && !a.smallc ? 0.3 : (bf6 ? 0.5 : 1.0));
      const double waves = (double)cdiv64(blocks, 256);
      double cost = waves * ((double)cdiv(KT, ns) * per_iter + 3000.0);
      // in-launch reduce: publish + ticket + acquire (~3 us) and the last arriver's slab reads (~100 GB/s per workgroup)
      if (ns > 1) cost += 7000.0 + (double)BM * BN * ns * 4.0 / 40.0;
      if (cost < best) { best = cost; bc = c; bs = ns; }
    }
  }
  pl->stream = -1;
  pl->wino = 0;
  const int wino_variant = d->reserved != 0 ? (d->reserved & 0xff) - 1 - kNumCfgs - stream_num_variants() : -1;
  if (wino_variant == 2) {   // forced: the Winograd F(4x4, 3x3) kernel (wino4.hip)
    CRDR_REQUIRE(!fallback && wino4_eligible(d, G, true), "conv2d: forced F(4x4, 3x3) Winograd kernel: not a 3x3 / 5x5 stride-1 or 5x5 stride-2 convolution it takes (wino4_eligible)");
    const int w4split = ((d->reserved >> 8) & 0xf) + 1;   // K splits per tile (1 = none), reduced inside the launch
    CRDR_REQUIRE(wino4_split_ok(d, G, w4split), "conv2d: F(4x4, 3x3) Winograd kernel: %d K splits do not fit this shape", w4split);
    pl->wino = 3;
    pl->cfg = -1;
    a.nsplit = w4split;
    a.ws_ld = 0;
    pl->grid = dim3(1, 1, 1);
    pl->lds = 0;
    a.cs_ld = round_up(d->OC, 32);
    a.cs_rows = want_cs ? wino4_colsum_rows(d) : 0;
    pl->ws_bytes = (size_t)CRDR_CONV_TICKETS * sizeof(int) + wino4_workspace(d, G, w4split);
    return 0;
  }
  if (wino_variant >= 0 && wino_variant < 2) {  // forced: the Winograd F(2x2, 3x3) kernel (wino.hip); variant 1 = pair tiles for a channel tail <= 32
    CRDR_REQUIRE(!fallback && wino_eligible(d, G), "conv2d: forced Winograd kernel: not a 3x3 / 5x5 stride-1 convolution it takes");
    CRDR_REQUIRE(wino_variant == 0 || wino_pairs_ok(d), "conv2d: Winograd pair-tile variant: needs a channel tail of 1..32 and more than one patch");
    CRDR_REQUIRE(((d->reserved >> 8) & 0xf) == 0, "conv2d: the Winograd kernel has no split-K");
    pl->wino = 1 + wino_variant;
    pl->cfg = -1;
    a.nsplit = 1;
    a.ws_ld = 0;
    pl->grid = dim3(1, 1, 1);
    pl->lds = 0;
    a.cs_ld = round_up(d->OC, 32);
    a.cs_rows = want_cs ? wino_colsum_rows(d) : 0;
    pl->ws_bytes = (size_t)CRDR_CONV_TICKETS * sizeof(int) + wino_workspace(d, G);
    return 0;
  }
  int auto_sv = -1;
  if (d->reserved == 0 && !fallback && !want_cs && !a.smallc && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 &&
      d->C % 32 == 0 && d->OC % 4 == 0 && !(d->flags & (CRDR_EPI_GATE | CRDR_EPI_PREADD | CRDR_EPI_ACCUM)) &&
      (long long)a.M * G >= 16384 && d->ldy % 4 == 0 && (!(d->flags & CRDR_EPI_RES) || d->ldres % 4 == 0) &&
      (!(d->flags & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) || d->ldmask % 4 == 0)) {
    // built-in choice for the 1x1 layers: the 8-wave streaming variant with the fewest padded columns (ties: BN 96, 128, 64, 160);
    // column-sum launches stay tiled so that crdr_conv2d_colsum_layout does not depend on operand alignment
    static const int pref[] = {1, 2, 0, 3};
    long long bestpad = 1ll << 60;
    for (int q = 0; q < 4; ++q) {
      const int sv = 4 + pref[q];
      if (sv >= stream_num_variants()) continue;
      int snb, sst, snw;
      stream_variant_shape(sv, &snb, &sst, &snw);
      const int BN = 32 * snb;
      const size_t lds = ((size_t)a.kchunks * BN * 32 + (size_t)sst * 32 * snw * 32 + snw * 2 * BN + 4 * BN) * sizeof(float);
      if (lds > 160 * 1024 || cdiv(d->OC, BN) > 32) continue;
      const long long pad = (long long)cdiv(d->OC, BN) * BN;
      if (pad < bestpad) { bestpad = pad; auto_sv = sv; }
    }
  }
  if (auto_sv >= 0 || (d->reserved != 0 && (d->reserved & 0xff) - 1 >= kNumCfgs)) {  // streaming 1x1 variant (forced or built-in)
    const int sv = auto_sv >= 0 ? auto_sv : (d->reserved & 0xff) - 1 - kNumCfgs;
    CRDR_REQUIRE(sv < stream_num_variants(), "conv2d: forced config %d out of range", sv + kNumCfgs);
    CRDR_REQUIRE(((d->reserved >> 8) & 0xf) == 0, "conv2d: the streaming 1x1 kernel has no split-K");
    CRDR_REQUIRE(!a.smallc && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->C % 32 == 0 &&
                     !(d->flags & (CRDR_EPI_GATE | CRDR_EPI_PREADD | CRDR_EPI_ACCUM)) && d->OC % 4 == 0,
                 "conv2d: the streaming kernel takes 1x1 stride-1 convolutions with C %% 32 == 0, OC %% 4 == 0 and no "
                 "gate / pre-add / accumulate epilogue");
    int snb, sstages, snw;
    stream_variant_shape(sv, &snb, &sstages, &snw);
    const int BN = 32 * snb, SBM = 32 * snw;
    const size_t lds = ((size_t)a.kchunks * BN * 32 + (size_t)sstages * SBM * 32 + snw * 2 * BN + 4 * BN) * sizeof(float);
    CRDR_REQUIRE(lds <= 160 * 1024, "conv2d: streaming variant %d needs %zu B of LDS for C = %d", sv, lds, d->C);
    const int gridN = cdiv(d->OC, BN), mtiles = cdiv(a.M, SBM);
    CRDR_REQUIRE(gridN <= 32, "conv2d: streaming variant %d: %d column tiles", sv, gridN);
    const int nlanes = std::max(8, std::min(256 / (gridN * G) / 8 * 8, round_up(mtiles, 8)));  // per problem
    pl->stream = sv;
    pl->cfg = -1;
    pl->sa.gridN = gridN;
    pl->sa.nlanes = nlanes;
    a.nsplit = 1;
    a.ws_ld = 0;
    pl->grid = dim3(G * gridN * nlanes, 1, 1);
    pl->lds = lds;
    a.cs_ld = round_up(d->OC, 32);
    a.cs_rows = want_cs ? mtiles : 0;
    pl->ws_bytes = 0;
    return 0;
  }
  if (d->reserved != 0) {  // caller-forced algorithm (autotuner): (config index + 1) | log2(split) << 8
    bc = (d->reserved & 0xff) - 1;
    bs = 1 << ((d->reserved >> 8) & 0xf);
    CRDR_REQUIRE(bc >= 0 && bc < kNumCfgs, "conv2d: forced config %d out of range", bc);
    CRDR_REQUIRE(!a.smallc || kCfgs[bc].kern_smallc, "conv2d: config %d has no tap-major variant", bc);
    CRDR_REQUIRE(!bf6 || kCfgs[bc].kern_bf6, "conv2d: config %d has no bf16x6 variant (its stages do not fit LDS)", bc);
    CRDR_REQUIRE(bs == 1 || KT / bs >= 2, "conv2d: forced split %d too deep for %d K-iterations", bs, KT);
    CRDR_REQUIRE(bs == 1 || !(d->flags & CRDR_CONV_NOSPLIT), "conv2d: forced split %d with CRDR_CONV_NOSPLIT", bs);
    if (bs > 1) {
      const TileCfg& ft = kCfgs[bc];
      const int fBM = 32 * ft.wm * ft.mb, fBN = 32 * ft.wn * ft.nb;
      CRDR_REQUIRE(splitk_ok((long long)cdiv(a.M, fBM) * cdiv(d->OC, fBN) * a.nphase * G, a.M, cdiv(d->OC, fBN) * fBN, bs),
                   "conv2d: forced split %d: too many tiles or slabs too large for the in-launch reduce", bs);
    }
  }
  CRDR_REQUIRE(bc >= 0, "conv2d: no tile config");
  const TileCfg& t = kCfgs[bc];
  const int BM = 32 * t.wm * t.mb, BN = 32 * t.wn * t.nb;
  pl->cfg = bc;
  a.nsplit = bs;
  pl->grid = dim3(cdiv(a.M, BM), cdiv(d->OC, BN), a.nphase * bs * G);
  {  // tile order inside an XCD's contiguous range.  N-inner (default): the activation tile of an M tile stays in L2 while
     // its N tiles / phases / splits run, and every M tile streams the weight pack -- right while the pack is L2 / MALL
     // sized.  M-inner: one weight tile stays while the M tiles stream past it -- chosen only when the pack is far beyond
     // any cache and dwarfs the activations (hoisted Charm convs: a 136 MB pack against 5 MB of activations; measured L2-miss
     // traffic of that launch 8.8 -> 4.4 GB, +2 % speed).  Not for their input gradients (70 MB of activations re-read per
     // tap: M-inner raised the traffic from 5.7 to 7.3 GB) nor for in-between shapes (5x5 s2 transposed, 6.5 MB pack).
    const double A = (double)d->N * d->H * d->W * d->C * 4.0, B = (double)a.w_bytes;
    a.m_inner = (G == 1 && B >= 32.0e6 && A * 4.0 <= B) ? 1 : 0;
  }
  a.ws_ld = pl->grid.y * BN;
  {  // K order: experiment switch CRDR_K_CMAJOR = 0 (tap-major) / 1 (channel-major wherever it applies) / unset: built-in rule
    static const int mode = [] { const char* e = getenv("CRDR_K_CMAJOR"); return e ? atoi(e) : -1; }();
    const bool applies = !a.smallc && maxtaps > 1 && a.kchunks > 1;
    a.k_cmajor = applies && (mode == 1 || (mode < 0 && d->C >= 1024)) ? 1 : 0;
  }
  pl->lds = cfg_lds(t.wm, t.wn, t.mb, t.nb, bf6 ? 6 : 0);
  a.cs_ld = round_up(d->OC, 32);
  a.cs_rows = want_cs ? a.nphase * (int)pl->grid.x : 0;
  // split-K workspace: [tickets: CRDR_CONV_TICKETS ints, zero between launches][slabs]
  pl->ws_bytes = bs > 1 ? (size_t)CRDR_CONV_TICKETS * sizeof(int) + (size_t)G * a.nphase * bs * a.M * a.ws_ld * sizeof(float) : 0;
  return 0;
}

}  // namespace crdr

using namespace crdr;

extern "C" int crdr_conv2d_num_configs(void) { return kNumCfgs; }
extern "C" int crdr_conv2d_num_stream_configs(void) { return stream_num_variants(); }
extern "C" int crdr_conv2d_num_wino_configs(void) { return 3; }

extern "C" size_t crdr_conv2d_workspace(const crdr_conv_desc* d) {
  Plan pl;
  if (build_plan(d, &pl)) return 0;
  return pl.ws_bytes;
}

extern "C" size_t crdr_conv2d_grouped_workspace(const crdr_conv_desc* d, int G) {
  Plan pl;
  if (build_plan(d, &pl, G)) return 0;
  return pl.ws_bytes;
}

extern "C" int crdr_conv2d_colsum_layout(const crdr_conv_desc* d, int G, int* rows, int* ld) {
  CRDR_REQUIRE(d && rows && ld, "conv2d_colsum_layout: null pointer");
  Plan pl;
  if (int rc = build_plan(d, &pl, G)) return rc;
  *rows = pl.a.cs_rows;
  *ld = pl.a.cs_ld;
  return 0;
}

extern "C" int crdr_conv2d_choose_algo(const crdr_conv_desc* d, int G) {
  Plan pl;
  if (!d || build_plan(d, &pl, G)) return 0;
  if (pl.wino) return (kNumCfgs + 1 + stream_num_variants() + (pl.wino - 1)) | ((pl.a.nsplit - 1) << 8);   // (F(4x4): bits 8..11 = K splits - 1)
  if (pl.stream >= 0) return kNumCfgs + 1 + pl.stream;
  int ls = 0;
  while ((1 << ls) < pl.a.nsplit) ++ls;
  return (pl.cfg + 1) | (ls << 8);
}

extern "C" double crdr_conv2d_flops(const crdr_conv_desc* d) {
  // exact count of in-bounds multiply-accumulates is shape dependent only at the borders; report the dense count
  if (!d->transposed) return 2.0 * d->N * d->OH * d->OW * (double)d->OC * d->C * d->kh * d->kw;
  return 2.0 * d->N * d->H * d->W * (double)d->OC * d->C * d->kh * d->kw;
}

static int launch_conv(const crdr_conv_desc* d, const crdr_conv_io* ios, int G, void* ws, size_t ws_bytes, crdr_stream_t s,
                       float* ucache = nullptr, size_t ucache_bytes = 0, int ucache_valid = 0) {
  Plan pl;
  if (int rc = build_plan(d, &pl, G)) return rc;
  IgemmArgs& a = pl.a;
  const crdr_conv_io* io = ios;
  a.x = io->x; a.w = io->w; a.y = io->y;
  a.counters = (int*)ws;
  a.ws = (float*)ws + CRDR_CONV_TICKETS;
  a.bias = io->bias; a.vec2 = io->vec2; a.res = io->res; a.scale = io->scale; a.shift = io->shift;
  a.gx = io->gx; a.gt = io->gt; a.sig = io->sig; a.pre = io->pre; a.mask = io->mask; a.cs = io->cs;
  if (a.M == 0) return 0;  // empty batch: nothing to compute (tensors may legitimately be null)
  CRDR_REQUIRE(!(a.flags & CRDR_EPI_VEC2) || a.vec2, "conv2d: VEC2 flag without vec2");
  CRDR_REQUIRE(!(a.flags & CRDR_EPI_RES) || a.res, "conv2d: RES flag without res");
  CRDR_REQUIRE(!((a.flags & CRDR_EPI_RELUMASK) && (a.flags & CRDR_EPI_LRELUMASK)), "conv2d: RELUMASK and LRELUMASK are exclusive");
  CRDR_REQUIRE(!(a.flags & CRDR_EPI_AFFINE) || (a.scale && a.shift), "conv2d: AFFINE flag without scale/shift");
  CRDR_REQUIRE(!(a.flags & CRDR_EPI_GATE) || (a.gx && a.gt && a.sig), "conv2d: GATE flag without gx/gt/sig");
  CRDR_REQUIRE(G == 1 || !(a.flags & (CRDR_EPI_VEC2 | CRDR_EPI_AFFINE | CRDR_EPI_GATE | CRDR_EPI_MASKOFF)),
               "conv2d_grouped: epilogue flags %d not supported in a grouped launch", a.flags);
  CRDR_REQUIRE(!(a.flags & CRDR_EPI_MASKOFF) || ((a.flags & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) && a.vec2 && !(a.flags & CRDR_EPI_VEC2)),
               "conv2d: MASKOFF needs a mask flag and vec2, and excludes VEC2");
  CRDR_REQUIRE(pl.ws_bytes <= ws_bytes, "conv2d: workspace too small (%zu < %zu)", ws_bytes, pl.ws_bytes);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  IgemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  bool v = (a.ldy % 4 == 0);
  if (a.flags & CRDR_EPI_RES) v = v && (a.ldres % 4 == 0) && al16(a.res);
  if (a.flags & CRDR_EPI_GATE) v = v && (a.ldg % 4 == 0) && al16(a.gx) && al16(a.gt) && al16(a.sig);
  if (a.flags & CRDR_EPI_PREADD) v = v && (a.ldpre % 4 == 0);
  if (a.flags & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) v = v && (a.ldmask % 4 == 0);
  for (int g = 0; g < G; ++g) {
    const crdr_conv_io& q = ios[g];
    CRDR_REQUIRE(q.x && q.w && q.y, "conv2d: null tensor (problem %d)", g);
    // both kernels stage x and w with 16-byte LDS-DMA: a view at a channel offset that is not a multiple of 4 floats
    // would be mis-addressed, not slow
    CRDR_REQUIRE(al16(q.x) && al16(q.w), "conv2d: x and w must be 16-byte aligned (problem %d)", g);
    CRDR_REQUIRE(!(a.flags & CRDR_EPI_BIAS) || q.bias, "conv2d: BIAS flag without bias (problem %d)", g);
    CRDR_REQUIRE(!(a.flags & CRDR_EPI_PREADD) || q.pre, "conv2d: PREADD flag without pre (problem %d)", g);
    CRDR_REQUIRE(!(a.flags & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) || q.mask, "conv2d: mask flag without mask (problem %d)", g);
    CRDR_REQUIRE(!(a.flags & CRDR_EPI_RES) || q.res, "conv2d: RES flag without res (problem %d)", g);
    CRDR_REQUIRE(!(a.flags & CRDR_EPI_COLSUM) || q.cs, "conv2d: COLSUM flag without cs (problem %d)", g);
    grp.x[g] = q.x; grp.w[g] = q.w; grp.y[g] = q.y; grp.bias[g] = q.bias; grp.pre[g] = q.pre; grp.mask[g] = q.mask;
    grp.res[g] = q.res; grp.cs[g] = q.cs;
    v = v && al16(q.y);
    if (a.flags & CRDR_EPI_RES) v = v && al16(q.res);
    if (a.flags & CRDR_EPI_PREADD) v = v && al16(q.pre);
    if (a.flags & (CRDR_EPI_RELUMASK | CRDR_EPI_LRELUMASK)) v = v && al16(q.mask);
  }
  a.vec_epi = v ? 1 : 0;
  // 32-bit byte offsets of the fast epilogue span one tile of output pixels (<= 256 rows, a few image rows when transposed)
  // a tile of BM <= 256 GEMM rows covers at most BM / GW + 2 grid rows, each of them `so` output rows of OW pixels
  const long long span_px = (256 / std::max(a.GW, 1) + 2) * (long long)std::max(a.so, 1) * a.OW + a.OW;
  const long long span = span_px * std::max(std::max(a.ldy, a.ldres), a.ldmask) * 4;
  a.fast_epi = (v && a.Cout % 4 == 0 && span < (1ll << 31) && !(a.flags & (CRDR_EPI_GATE | CRDR_EPI_PREADD | CRDR_EPI_ACCUM))) ? 1 : 0;
  if (pl.wino) {
    void* prof = profile_begin(as_stream(s));
    if (pl.wino == 3) {
      // transformed filters: in the workspace, or in the caller's buffer (crdr_conv2d_grouped_ex: kept across launches that share
      // weights -- `ucache_valid` skips the transform); the partial tiles of a split launch always live in the workspace
      float* u = (float*)ws + CRDR_CONV_TICKETS;
      if (ucache) {
        CRDR_REQUIRE(ucache_bytes >= wino4_workspace(d, G, 1), "conv2d: filter cache of %zu bytes, the launch needs %zu", ucache_bytes, wino4_workspace(d, G, 1));
        u = ucache;
      }
      float* slabs = (float*)ws + CRDR_CONV_TICKETS + wino4_workspace(d, G, 1) / 4;
      if (int rc = wino4_launch(d, a, pl.t, grp, G, u, slabs, a.nsplit, ucache && ucache_valid, as_stream(s))) return rc;
    } else if (int rc = wino_launch(d, pl.wino - 1, a, pl.t, grp, G, (float*)ws + CRDR_CONV_TICKETS, as_stream(s))) return rc;
    // kind 3 / 5 / 6: filter transform + Winograd F(2x2, 3x3) / F(4x4, 3x3) kernel (6: a 5x5 stride-2 layer through it), direct-convolution
    // flop count
    profile_end(pl.wino == 3 ? (d->kh == 5 ? 6 : 5) : 3, G * crdr_conv2d_flops(d), prof, as_stream(s));
    return 0;
  }
  if (pl.stream >= 0 && !a.vec_epi && d->reserved == 0) {
    // the built-in choice assumed 16-byte aligned operands (it only knows the strides): take the tiled kernel instead
    Plan fb;
    if (int rc = build_plan(d, &fb, G, true)) return rc;
    fb.a.x = a.x; fb.a.w = a.w; fb.a.y = a.y; fb.a.ws = a.ws; fb.a.counters = a.counters; fb.a.bias = a.bias; fb.a.vec2 = a.vec2; fb.a.res = a.res;
    fb.a.scale = a.scale; fb.a.shift = a.shift; fb.a.gx = a.gx; fb.a.gt = a.gt; fb.a.sig = a.sig; fb.a.pre = a.pre;
    fb.a.mask = a.mask; fb.a.cs = a.cs; fb.a.vec_epi = 0; fb.a.fast_epi = 0;
    pl = fb;
  }
  if (pl.stream >= 0) {
    CRDR_REQUIRE(a.vec_epi, "conv2d: the streaming kernel needs 16-byte aligned operand rows");
    void* prof = profile_begin(as_stream(s));
    stream_launch(pl.stream, a, pl.sa, grp, pl.grid.x, pl.lds, as_stream(s));
    CRDR_CHECK_LAUNCH("gemm1x1_kernel");
    profile_end(0, G * crdr_conv2d_flops(d), prof, as_stream(s));
    return 0;
  }
  const TileCfg& t = kCfgs[pl.cfg];
  // (RGB-input layers stay exact: K is tiny there); 3: the FAST form of the plain kernel (same arithmetic, same order: bit-identical results)
  const int variant = a.smallc ? 1 : ((d->flags & CRDR_CONV_BF16X3) ? 2 : ((d->flags & CRDR_CONV_BF16X6) ? 4 : ((a.fast_epi && a.nsplit == 1) ? 3 : 0)));
  auto kern = variant == 1 ? t.kern_smallc : (variant == 2 ? t.kern_bf3 : (variant == 3 ? t.kern_fast : (variant == 4 ? t.kern_bf6 : t.kern)));
  CRDR_REQUIRE(kern, "conv2d: config %d has no kernel for this precision", pl.cfg);
  static std::atomic<bool> attr_done[5][64];
  if (!attr_done[variant][pl.cfg].load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[variant][pl.cfg].store(true, std::memory_order_release);
  }
  void* prof = profile_begin(as_stream(s));
  hipLaunchKernelGGL(kern, pl.grid, dim3(64 * t.wm * t.wn), pl.lds, as_stream(s), a, pl.t, grp);
  CRDR_CHECK_LAUNCH("igemm_kernel");
  profile_end(0, G * crdr_conv2d_flops(d), prof, as_stream(s));  // kind 0 = the igemm kernel alone (what rocprofv3 lists)
  return 0;
}

extern "C" int crdr_conv2d(const crdr_conv_desc* d, const crdr_conv_io* io, void* ws, size_t ws_bytes,
                           crdr_stream_t s) {
  CRDR_REQUIRE(d && io, "conv2d: null descriptor");
  return launch_conv(d, io, 1, ws, ws_bytes, s);
}

extern "C" int crdr_conv2d_grouped(const crdr_conv_desc* d, const crdr_conv_io* ios, int G, void* ws, size_t ws_bytes,
                                   crdr_stream_t s) {
  CRDR_REQUIRE(d && ios, "conv2d_grouped: null descriptor");
  CRDR_REQUIRE(G >= 1 && G <= CRDR_MAX_GROUP, "conv2d_grouped: %d problems (1..%d)", G, CRDR_MAX_GROUP);
  return launch_conv(d, ios, G, ws, ws_bytes, s);
}

extern "C" size_t crdr_conv2d_filter_cache_bytes(const crdr_conv_desc* d, int G) {
  Plan pl;
  if (!d || G < 1 || G > CRDR_MAX_GROUP || build_plan(d, &pl, G) || pl.wino != 3) return 0;
  return wino4_workspace(d, G, 1);
}

extern "C" int crdr_conv2d_filter_item(const crdr_conv_desc* d, int G, crdr_w4_filter_item* item) {
  CRDR_REQUIRE(d && item, "conv2d_filter_item: null pointer");
  CRDR_REQUIRE(G >= 1 && G <= CRDR_MAX_GROUP, "conv2d_filter_item: %d problems (1..%d)", G, CRDR_MAX_GROUP);
  Plan pl;
  if (int rc = build_plan(d, &pl, G)) return rc;
  CRDR_REQUIRE(pl.wino == 3, "conv2d_filter_item: the descriptor's `reserved` does not force the F(4x4, 3x3) kernel");
  return wino4_filter_item(d, pl.t, G, item);
}

extern "C" int crdr_w4_filters_batched(const crdr_w4_filter_item* items, const int64_t* prefix, const int64_t* meta, crdr_stream_t s) {
  CRDR_REQUIRE(items && prefix && meta, "w4_filters_batched: null pointer");
  return wino4_filters_batched(items, reinterpret_cast<const long long*>(prefix), reinterpret_cast<const long long*>(meta), as_stream(s));
}

extern "C" int crdr_conv2d_grouped_ex(const crdr_conv_desc* d, const crdr_conv_io* ios, int G, void* ws, size_t ws_bytes, float* filter_cache,
                                      size_t filter_cache_bytes, int filter_cache_valid, crdr_stream_t s) {
  CRDR_REQUIRE(d && ios, "conv2d_grouped_ex: null descriptor");
  CRDR_REQUIRE(G >= 1 && G <= CRDR_MAX_GROUP, "conv2d_grouped_ex: %d problems (1..%d)", G, CRDR_MAX_GROUP);
  CRDR_REQUIRE(!filter_cache || (reinterpret_cast<uintptr_t>(filter_cache) & 15) == 0, "conv2d_grouped_ex: the filter cache must be 16-byte aligned");
  return launch_conv(d, ios, G, ws, ws_bytes, s, filter_cache, filter_cache_bytes, filter_cache_valid);
}
