/*
 * crdr_hip.h -- C ABI of libcrdr_hip.so, the MI355X (gfx950) arithmetic behind the CRDR codec hot path.
 *
 * Boundary contract:
 *   - flat extern "C", plain pointers + sizes, no torch / C++ types, no exceptions cross the boundary;
 *   - every device entry enqueues on the caller's hipStream_t only, allocates nothing, retains no pointer;
 *   - return 0 on success, negative on error; message via crdr_last_error() (thread local);
 *   - activations are NHWC fp32 (a torch channels_last tensor), `ld*` = pixel stride in elements so that
 *     channel slices of a wider tensor can be read / written in place (no concat copies);
 *   - host entries (rANS, pmf_to_quantized_cdf) take no stream.
 *
 * Each entry cites the reference interface it replaces (paths relative to the iwa-shi/CRDR tree).
 */
#ifndef CRDR_HIP_H
#define CRDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* crdr_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------------------------------------ */
/* library                                                                                          */
/* ------------------------------------------------------------------------------------------------ */
const char* crdr_last_error(void);
int crdr_version(void);
const char* crdr_arch(void); /* "gfx950" */
/* measurement aid: bracket conv launches with HIP events on their stream; read = sum over launches of `kind`
 * (0: igemm kernel of a conv forward / input gradient, 1: weight-gradient slab kernel (+ its reduce when not deferred),
 * 2: split-K epilogue kernels, 3: Winograd launches of a conv forward / input gradient -- filter transform + kernel, counted with the
 * direct convolution's FLOPs, 2.25x what the matrix cores execute, 4: Winograd weight-gradient slab launches, likewise, 5: F(4x4, 3x3)
 * Winograd launches of a 3x3 layer (4x what is executed), 6: of a 5x5 layer as four 3x3 sub-filters / phases (25/9 x), 7 / 8: F(3x3, 4x4)
 * Winograd weight-gradient slab launches of 3x3 / 5x5 layers (4x / 25/9 x)) of
 * algorithmic FLOPs and elapsed ms, then clear */
void crdr_profile_enable(int on);
int crdr_profile_read(int kind, double* flops, double* ms, long long* launches);

/* ------------------------------------------------------------------------------------------------ */
/* implicit-GEMM convolution family (fp32 in / fp32 MFMA v_mfma_f32_32x32x2_f32 / fp32 out)          */
/*   replaces nn.Conv2d / nn.ConvTranspose2d forward + input-gradient as used by                     */
/*   src/models/layer/elic_layers.py:14-36, src/models/layer/cheng_nlam.py:31-46,                    */
/*   src/models/subnet/autoencoder/elic_autoencoder.py:42-53,                                         */
/*   src/models/subnet/hyperprior/minnen20_hyperprior.py:16-18,50-52,                                 */
/*   src/models/subnet/context_model/minnen20_charm_context_model.py:29-35,                           */
/*   src/models/discriminator/clic21_gvae_discriminator.py:12-40                                      */
/* ------------------------------------------------------------------------------------------------ */

/* epilogue flags, applied in this order on v = acc:                                                 */
#define CRDR_EPI_BIAS 1    /* v += bias[oc]                                                          */
#define CRDR_EPI_RELU 2    /* v = max(v, 0)                                                          */
#define CRDR_EPI_LRELU 4   /* v = v > 0 ? v : 0.2 v        (clic21_gvae_discriminator.py:24)        */
#define CRDR_EPI_VEC2 8    /* v += vec2[oc]  (beta-cond proj added AFTER the ReLU,                   */
                           /*                 elic_interpca_beta_cond_autoencoder.py:56-66)          */
#define CRDR_EPI_RES 16    /* v += res[pix][oc]            (elic_layers.py:36, cheng_nlam.py:45)    */
#define CRDR_EPI_GATE 32   /* s = sigmoid(v); sig[pix][oc] = s; v = gx + gt * s (cheng_nlam.py:23-29) */
#define CRDR_EPI_AFFINE 64 /* v = v * scale[oc] + shift[oc] (InterpChAtt, interp_channel_attention.py:68-72) */
#define CRDR_EPI_ACCUM 128 /* y[pix][oc] += v  instead of  = v (gradient accumulation)              */
/* v = acc + pre[pix][oc] BEFORE the bias: the K range of a conv may be split over several launches (the Charm's
 * first-layer convs: the hyper-prior channels of all 28 slice transforms are computed up front as two wide convs,
 * the support channels follow inside the slice loop, minnen20_charm_context_model.py:88-141); pre may alias y */
#define CRDR_EPI_PREADD 256
/* v = mask[pix][oc] > 0 ? v : 0, applied last (before ACCUM): the ReLU backward of the layer below, fused into the
 * input-gradient conv that produces its output gradient (mask = that layer's saved post-ReLU activation) */
#define CRDR_EPI_RELUMASK 512
#define CRDR_EPI_LRELUMASK 1024 /* v = mask > 0 ? v : 0.2 v: the LeakyReLU(0.2) backward, same position as RELUMASK */
#define CRDR_EPI_MASKOFF 2048   /* with (L)RELUMASK: compare mask[pix][oc] - vec2[oc] (the layer below added vec2 AFTER its
                                 * ReLU, so its saved output is relu(z) + vec2); vec2 is NOT added to v in this case */
/* cs[row][0][oc] = sum over the rows of one (phase, M tile) of v right before the mask step, cs[row][1][oc] = after it
 * (both before ACCUM): partial column sums of what this launch writes, e.g. the bias / beta-vector gradients of the layer
 * whose output gradient an input-gradient launch produces.  crdr_conv2d_colsum_layout gives the number of partial rows
 * and their stride; crdr_colsum_finish_batched adds them up in row order (deterministic). */
#define CRDR_EPI_COLSUM 4096
#define CRDR_CONV_BF16X3 16384 /* opt-in reduced-cost products: operands split on the fly into (hi, lo) bf16 pairs, a b ~= ah bh + ah bl +
                                * al bh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (per-product relative error <= 3 * 2^-16) at
                                * 3/16 of the exact-fp32 MFMA cost.  Off = exact fp32 (the default, the codec path, every parity claim).
                                * The reference's own GPU convolutions run TF32 (base_trainer.py:20 + torch 1.12 defaults), a coarser
                                * format than this.  Ignored by the streaming 1x1 kernel and by RGB-input (tap-major) layers. */
#define CRDR_CONV_BF16X6 32768 /* opt-in fp32-EQUIVALENT products on the bf16 matrix path: operands split exactly into three bf16 pieces
                                * (hi + mid + lo = x: 3 x 8 bits of the 24-bit significand), a b evaluated as the six products of weight >= 2^-16
                                * (am bm, al bh, ah bl, am bh, ah bm, ah bh, fp32 accumulate, small terms first); dropped terms <= 2^-23 |a b|, one
                                * fp32 rounding.  3/8 of the exact-fp32 MFMA time.  Exclusive with CRDR_CONV_BF16X3.  The Winograd kernels and the
                                * RGB-input (tap-major) layers ignore it and stay on the exact fp32 instruction. */
#define CRDR_CONV_NOSPLIT 8192 /* plan without split-K (the caller's workspace cannot hold the zeroed tickets, see CRDR_CONV_TICKETS) */

typedef struct crdr_conv_desc {
  /* "in" tensor [N][H][W][C] (NHWC, pixel stride ldx) and "out" tensor [N][OH][OW][OC] (pixel stride ldy) */
  int32_t N, H, W, C;
  int32_t OH, OW, OC;
  int32_t kh, kw, stride, pad;
  /* transposed = 0: out[o] = sum_r in[o*stride - pad + r] * w[r]            (Conv2d fwd, ConvT dgrad)
   * transposed = 1: out[i*stride - pad + r] += in[i] * w[r]                 (ConvT fwd, Conv2d dgrad) */
  int32_t transposed;
  int32_t ldx, ldy;
  /* weight pack [kh*kw][wrows][wcols] fp32, wrows >= OC (multiple of 32), wcols >= C (multiple of 32),
   * made by crdr_pack_weight; element [t][oc][c] multiplies in-channel c for out-channel oc at tap t */
  int32_t wrows, wcols;
  int32_t flags;
  int32_t ldres; /* pixel stride of res                                    */
  int32_t ldg;   /* pixel stride of gx, gt and sig                          */
  int32_t wlayout;  /* 0: weight pack [kh*kw][wrows][wcols] (below); 1: tap-major [wrows][wcols >= 4*kh*kw] for C <= 4
                     * (RGB inputs): a K-tile then covers 8 taps x 4 channels instead of 1 tap x 32 mostly-zero channels */
  int32_t reserved; /* 0: built-in heuristic; else a forced algorithm = (config index + 1) | log2(split-K) << 8
                     * (what cudnn.benchmark=True does for the reference, base_trainer.py:20: time the candidates
                     * once per shape and keep the fastest; see crdr_amd/hip/ops.py) */
  int32_t ldpre;    /* pixel stride of pre  (CRDR_EPI_PREADD)   */
  int32_t ldmask;   /* pixel stride of mask (CRDR_EPI_RELUMASK) */
} crdr_conv_desc;

/* Split-K plans (crdr_conv2d_workspace(d) > 0) reduce inside the launch: every K split publishes its partial tile, takes a
 * ticket, and the workgroup that arrives last adds the slabs in split order and runs the epilogue.  The tickets are the first
 * CRDR_CONV_TICKETS int32 of the workspace: they must be ZERO when the first launch that uses a workspace buffer starts, and
 * every launch leaves them zero -- hand the same (initially zeroed) buffer to successive launches of one stream. */
#define CRDR_CONV_TICKETS 16384

typedef struct crdr_conv_io {
  const float* x;
  const float* w;
  float* y;
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
  float* cs; /* CRDR_EPI_COLSUM: [rows][2][ld] floats, see crdr_conv2d_colsum_layout */
} crdr_conv_io;

/* number of tile configurations a forced algorithm may name */
int crdr_conv2d_num_configs(void);
/* the forced-algorithm id (crdr_conv_desc.reserved) of what the built-in heuristic picks for a launch of G problems of d
 * (d->reserved is honoured if set); 0 on error.  Results of a launch depend on its tile configuration and split depth (fp32
 * summation order): a caller that needs two launches of different group size to agree bit for bit forces one plan on both. */
int crdr_conv2d_choose_algo(const crdr_conv_desc* d, int G);
/* number of streaming 1x1 variants: forced algorithm ids crdr_conv2d_num_configs() + 1 + v, no split bits.  A persistent
 * workgroup keeps its weight tile in LDS and streams the activation rows through per-wave DMA rings (the 1x1 layers of
 * ResidualBottleneck, elic_layers.py:17-36, and of the NLAM branches, cheng_nlam.py); plain and grouped launches.  Rejected
 * (error return, as for any unsuitable forced id) unless the convolution is 1x1 stride 1 with C % 32 == 0, OC % 4 == 0,
 * 16-byte aligned rows, no GATE / PREADD / ACCUM epilogue and a weight tile that fits LDS.  Results are bit-identical to
 * the unsplit tile configurations; CRDR_EPI_COLSUM rows are per 128- or 256-row tile (crdr_conv2d_colsum_layout) */
int crdr_conv2d_num_stream_configs(void);
/* number of Winograd variants (3): forced algorithm id crdr_conv2d_num_configs() + 1 + crdr_conv2d_num_stream_configs() + v, no split bits.
 * v = 0, 1: F(2x2, 3x3) minimal filtering on the exact-fp32 matrix cores for 3x3 stride-1 convolutions and their input gradients (2.25x
 * fewer multiply-accumulates than the implicit GEMM; what cuDNN's WINOGRAD algorithms do for the reference, base_trainer.py:20
 * cudnn.benchmark); v = 1 covers a channel tail of <= 32 with PAIR tiles -- the tail's channels of two 16x16 patches per workgroup -- so
 * that e.g. 96 channels cost 1.5 tiles per patch, not 2.  fp32 arithmetic throughout (data transform +-1, filter transform halves); the
 * sums are associated differently from the direct form.  Rejected unless kh = kw = 3 (or 5, as 2 x 2 sub-filters), stride 1, C % 4 == 0
 * and the epilogue has no GATE / PREADD (grouped: nor VEC2 / AFFINE / MASKOFF).
 * v = 2: F(4x4, 3x3) (wino4.hip): 4x fewer multiply-accumulates than the implicit GEMM on a 3x3 stride-1 layer; also takes the 5x5 layers
 * as 3x3 pieces (2.78x fewer): 5x5 stride-2 pad-2 convolutions of even-sized inputs (four parity sub-filters, elic_autoencoder.py:42-52),
 * 5x5 stride-2 pad-2 transposed convolutions with output = 2 x input (four output phases, elic_layers.py:14-21), 5x5 stride-1 pad-2 layers
 * of >= 12 input channels (four shifted sub-filters, minnen20_charm_context_model.py:26-38).  Output tiles of 8 x 64 pixels x 64 channels
 * (>= 33 output / phase columns), 16 x 32 pixels (24..32 columns) or, for the stride-1 forms, two whole images of 9..16 pixels a side;
 * needs C % 4 == 0, OC % 4 == 0 and 16-byte aligned operand rows.  Bits 8..11 of the forced id = K splits - 1 (1..16 work items per tile
 * over equal parts of the K range, reduced inside the launch in split order: bit-identical run to run; the ticket head of the workspace must
 * be zero, as for the implicit GEMM's split-K).  Interpolation points 0, +-3/4, +-5/4, infinity (round 5; rounds 3-4: 0, +-1, +-2): results
 * deviate from float64 by ~0.4e-6 .. 5e-6 of the output scale (the direct kernels: ~1e-6 .. 4e-6) -- still a candidate of the training-side
 * tuner only, never of the built-in plan (the codec runs built-in plans).
 * The transformed filters are rebuilt from the weight pack into the workspace by every launch (crdr_conv2d_workspace with the same
 * `reserved`; crdr_conv2d_grouped_ex keeps them in a caller's buffer instead); CRDR_EPI_COLSUM rows are per output tile (and phase),
 * crdr_conv2d_colsum_layout. */
int crdr_conv2d_num_wino_configs(void);
/* bytes of workspace crdr_conv2d needs for this problem (split-K partial slabs; may be 0) */
size_t crdr_conv2d_workspace(const crdr_conv_desc* d);
int crdr_conv2d(const crdr_conv_desc* d, const crdr_conv_io* io, void* ws, size_t ws_bytes, crdr_stream_t s);
/* partial-row count and row stride (floats) of io.cs for this problem (depends on the tile configuration the plan picks,
 * so pass the same desc -- including `reserved` -- and group size as the launch) */
int crdr_conv2d_colsum_layout(const crdr_conv_desc* d, int G, int* rows, int* ld);
/* Finish pending column sums, all jobs in two launches: pass A adds slabs of crdr_colsum_slab_rows() partial rows (stride
 * 2 * ld) in row order into scratch[scratch_off + (slab * 2 + which) * cpad + c], pass B adds a job's nslab scratch rows in
 * order and writes (accumulate != 0: adds to) out_pre[c] / out_post[c], c < C (either may be NULL).  nslab = ceil(rows /
 * slab rows), cpad = C rounded up to 64; scratch_off (floats) must give every job its own nslab * 2 * cpad floats.  `jobs`,
 * `prefix_a` (first pass-A tile of each job; a job has cpad / 64 * nslab of them), `prefix_b` (first pass-B tile; cpad / 64 per
 * job) and `meta` = {number of jobs, total pass-A tiles, total pass-B tiles} are DEVICE arrays (a table rewritten in place keeps
 * working under a captured HIP graph), like crdr_wgrad_reduce_batched. */
typedef struct crdr_colsum_job {
  const float* cs;
  float* out_pre;
  float* out_post;
  int32_t rows, ld, C, accumulate;
  int32_t nslab, cpad;
  int64_t scratch_off;
} crdr_colsum_job;
int crdr_colsum_slab_rows(void);
int crdr_colsum_finish_batched(const crdr_colsum_job* jobs, const int64_t* prefix_a, const int64_t* prefix_b, const int64_t* meta,
                               float* scratch, crdr_stream_t s);

/* G <= CRDR_MAX_GROUP independent convolutions of ONE geometry (same desc) in one launch; problem g reads
 * ios[g].{x, w, bias, pre, mask, res, cs} and writes ios[g].y (the other io fields must be unused: flags limited to BIAS,
 * RELU, LRELU, RES, PREADD, RELUMASK, LRELUMASK, COLSUM, ACCUM).  The mean and scale transforms of a Charm slice -- and, from slice 5 on, those of
 * all remaining slices, whose support no longer grows (minnen20_charm_context_model.py:104-105) -- are independent
 * and identically shaped: one launch fills the chip where 2..15 small ones each pay their own ramp.
 * Workspace: crdr_conv2d_grouped_workspace(d, G).  No two problems may write overlapping outputs. */
#define CRDR_MAX_GROUP 16
size_t crdr_conv2d_grouped_workspace(const crdr_conv_desc* d, int G);
int crdr_conv2d_grouped(const crdr_conv_desc* d, const crdr_conv_io* ios, int G, void* ws, size_t ws_bytes, crdr_stream_t s);
/* The same launch (G = 1: crdr_conv2d) with the Winograd F(4x4) kernel's transformed filters kept in a caller-owned buffer instead of the
 * workspace: crdr_conv2d_filter_cache_bytes(d, G) bytes (0 unless d->reserved forces that kernel), 16-byte aligned.  With
 * filter_cache_valid == 0 the launch fills it, with != 0 it trusts it -- for launches that share weights between two updates (the reference runs
 * its generator twice per step, multirate_hr_rgan_beta_cond_rate_distortion_trainer.py:42-47; cuDNN re-transforms there as well).  The caller
 * owns validity: same weights, same descriptor, same forced algorithm.  filter_cache == NULL: exactly crdr_conv2d_grouped. */
size_t crdr_conv2d_filter_cache_bytes(const crdr_conv_desc* d, int G);
/* Transformed filters as PERSISTENT packs (round 5): they go stale exactly when the weight packs they are derived from do -- right behind an
 * optimiser's update -- so every filter cache of that optimiser is rebuilt by ONE launch behind crdr_pack_weights_batched instead of one
 * transform launch in front of every convolution (round 4: 80.8 launches, 2.1 ms per stage-3 step).
 * crdr_conv2d_filter_item fills the description of one cache (a HOST struct; everything but the pointers `w[g]` -- the G weight packs, the
 * launch's crdr_conv_io.w -- and `u`, the cache of crdr_conv2d_filter_cache_bytes(d, G) bytes) for a descriptor whose `reserved` forces
 * the F(4x4) kernel.  crdr_w4_filters_batched takes DEVICE arrays like crdr_pack_weights_batched: item k owns the work units (one per
 * problem, 64-channel N tile, sub-filter / phase and 4-channel chunk: `units`) numbered [prefix[k], prefix[k+1]); meta = {number of items,
 * prefix[number of items]}.  A launch that finds its cache filled this way passes filter_cache_valid != 0 to crdr_conv2d_grouped_ex. */
typedef struct crdr_w4_filter_item {
  const float* w[16]; /* CRDR_MAX_GROUP source packs (the first G used) */
  float* u;
  int32_t G, Cin, Cout, wrows, wcols, kchunks, ntile, nvar;
  int32_t widx[4][9]; /* [sub-filter / phase][3 a + b]: tap of the weight pack, -1 = zero */
  int64_t units;      /* G * ntile * nvar * kchunks */
} crdr_w4_filter_item;
int crdr_conv2d_filter_item(const crdr_conv_desc* d, int G, crdr_w4_filter_item* item);
int crdr_w4_filters_batched(const crdr_w4_filter_item* items, const int64_t* prefix, const int64_t* meta, crdr_stream_t s);
int crdr_conv2d_grouped_ex(const crdr_conv_desc* d, const crdr_conv_io* ios, int G, void* ws, size_t ws_bytes, float* filter_cache,
                           size_t filter_cache_bytes, int filter_cache_valid, crdr_stream_t s);
/* algorithmic FLOPs of the call (2 * MACs actually needed, padding taps excluded approx.) for roofline maths */
double crdr_conv2d_flops(const crdr_conv_desc* d);

/* weight gradient.  P = dense operand [N][PH][PW][PC] (dy for Conv2d, x for ConvT), Q = gathered operand
 * [N][QH][QW][QC] (x for Conv2d, dy for ConvT): g[i][j][t] (+)= sum_{n,a,b} P[n,a,b,i] * Q[n,a*stride-pad+r,b*stride-pad+s,j]
 * which is exactly Conv2d.weight.grad [OC][IC][kh][kw] resp. ConvTranspose2d.weight.grad [IC][OC][kh][kw]. */
typedef struct crdr_wgrad_desc {
  int32_t N, PH, PW, PC, ldp;
  int32_t QH, QW, QC, ldq;
  int32_t kh, kw, stride, pad;
  int32_t gI, gJ;     /* dims of g (<= PC, QC): channels beyond them are layout padding and are dropped */
  int32_t accumulate; /* 1: g += ; 0: g = */
  int32_t algo;       /* low 16 bits 0: heuristic; else forced (config index + 1) | log2(pixel split) << 8;
                       * bit 16 (CRDR_WGRAD_BF16X3): split-bf16 products, see CRDR_CONV_BF16X3 */
} crdr_wgrad_desc;
#define CRDR_WGRAD_BF16X3 (1 << 16)
#define CRDR_WGRAD_SQUARE_Q (1 << 18) /* the gathered operand enters squared: g = sum P Q^2 (the gamma gradient of a GDN layer without an x^2 tensor in
                                      * memory); exact fp32, direct kernels, a subset of the tile configurations (others are refused when forced) */
#define CRDR_WGRAD_BF16X6 (1 << 17) /* fp32-equivalent split-bf16 products (see CRDR_CONV_BF16X6) in the direct weight-gradient kernels; the Winograd
                                     * slab kernels ignore it */
/* number of forced configurations; the LAST one (index crdr_conv2d_wgrad_num_configs() - 1) is the Winograd F(3x3, 2x2) slab kernel
 * (csrc/wino_wgrad.hip: 16 products per 2x2 tile of P and tap set instead of 36; same slabs, same deferred reduce), accepted for
 * kh = kw = 3, stride 1, QC > 4, exact fp32; its split bits divide the strips of 8 tiles instead of the 32-pixel tiles */
int crdr_conv2d_wgrad_num_configs(void);
/* Winograd weight-gradient slab kernels (3x3 stride 1, QC > 4, exact fp32): forced ids crdr_conv2d_wgrad_num_configs() = F(3x3, 2x2)
 * (wino_wgrad.hip: 2.25x fewer products) and + 1 = F(3x3, 4x4) (wino4_wgrad.hip: 4x fewer; its transforms carry 2, 4, 8 and 1/4 .. 1/24,
 * deviation from float64 ~1e-5 of the result's scale); bits 8..11 = log2 of the strip splits; same slabs / reduce as the direct kernels */
int crdr_conv2d_wgrad_num_wino_configs(void);
size_t crdr_conv2d_wgrad_workspace(const crdr_wgrad_desc* d);
int crdr_conv2d_wgrad(const crdr_wgrad_desc* d, const float* p, const float* q, float* g, void* ws, size_t ws_bytes,
                      crdr_stream_t s);

/* Deferred reduction.  crdr_conv2d_wgrad = partial-slab kernel + a small reduce launch per layer; a backward pass makes
 * hundreds of those.  crdr_conv2d_wgrad_partial runs only the slab kernel (into `slab`, crdr_conv2d_wgrad_workspace bytes,
 * which must stay untouched until the reduce) and describes the pending reduction in *job; crdr_wgrad_reduce_batched
 * then finishes ALL pending layers in one launch.  `jobs`, `prefix` (first tile of each job; a job has gI * ceil(gJ / 64) tiles --
 * one output row x 64 input channels x all taps -- or ceil(gI gJ T / 256) when smallj is set or T > 32) and
 * `meta` = {number of jobs, total tiles} are DEVICE arrays.  No two jobs of one batch may write the same g. */
typedef struct crdr_wgrad_job {
  const float* slab;
  float* g;
  int32_t PC, QC, gI, gJ, T, nsplit, smallj, accumulate;
  int32_t gJtot;   /* second dim of the parameter g points into (0 = gJ): g[(i * gJtot + j) * T + t]; with `g` offset by
                    * j0 * T the job fills the input-channel range [j0, j0 + gJ) of a wider weight, and with `slab` offset
                    * by i0 * QC it takes rows [i0, i0 + gI) of the slab: one slab launch may feed several parameters */
  int32_t reserved;
} crdr_wgrad_job;
int crdr_conv2d_wgrad_partial(const crdr_wgrad_desc* d, const float* p, const float* q, float* g, void* slab,
                              size_t slab_bytes, crdr_wgrad_job* job, crdr_stream_t s);
int crdr_wgrad_reduce_batched(const crdr_wgrad_job* jobs, const int64_t* prefix, const int64_t* meta, crdr_stream_t s);
/* G <= CRDR_MAX_GROUP weight gradients of one geometry in one slab launch: problem g = (ps[g], qs[g]) -> gs[g]; `slab`
 * holds G x crdr_conv2d_wgrad_workspace(d) bytes, jobs[g] describes the pending reduction of problem g. */
size_t crdr_conv2d_wgrad_grouped_workspace(const crdr_wgrad_desc* d, int G);
int crdr_conv2d_wgrad_partial_grouped(const crdr_wgrad_desc* d, const float* const* ps, const float* const* qs,
                                      float* const* gs, int G, void* slab, size_t slab_bytes, crdr_wgrad_job* jobs,
                                      crdr_stream_t s);

/* src[I][J][T] (a Conv2d / ConvTranspose2d parameter, T = kh*kw) -> dst[T][rows][cols] zero padded.
 * transpose = 0: dst[t][i][j] = src[i][j][t] (rows >= I, cols >= J); transpose = 1: dst[t][j][i] = src[i][j][t];
 * transpose = 2 (J <= 4): tap-major dst[i][4 t + j] = src[i][j][t] (rows >= I, cols >= 4 T), see crdr_conv_desc.wlayout;
 * transpose = 3 (J <= 4): scatter pack dst[4 t + j][i] = src[i][j][t] (rows >= 4 T, cols >= I), see crdr_col2im_rgb */
int crdr_pack_weight(const float* src, float* dst, int I, int J, int T, int rows, int cols, int transpose,
                     crdr_stream_t s);

/* Second half of an RGB-output transposed op done as GEMM + scatter (ConvTranspose2d C -> 3, or the input gradient of
 * a Conv2d 3 -> C): a 1x1 conv with the scatter pack above first produces cols[n][ih][iw][4 t + c] = sum_i u[n][ih][iw][i]
 * w[i][c][t] (4 T columns instead of T GEMMs padded from 3 to 32 columns); this kernel then gathers, per output pixel,
 *   out[n][oh][ow][c] = bias[c] + sum_{t=(r,s): (oh+pad-r) % stride == 0, (ow+pad-s) % stride == 0}
 *                                  cols[n][(oh+pad-r)/stride][(ow+pad-s)/stride][4 t + c],          c < C <= 4.
 * H, W: the grid of `cols`; OH, OW: the output; ldc / ldo: pixel strides in floats (multiples of 4). */
int crdr_col2im_rgb(const float* cols, int ldc, int N, int H, int W, int kh, int kw, int stride, int pad, const float* bias,
                    float* out, int ldo, int OH, int OW, int C, crdr_stream_t s);

/* All weight packs of an optimiser in one launch (they all go stale together, right after its fused Adam update).
 * `items` and `prefix` are DEVICE arrays: item k packs like crdr_pack_weight(src, dst, I, J, T, rows, cols, mode) with
 * mode 0 or 1, T <= 32, rows % 8 == 0, cols % 32 == 0, and owns the tiles (8 pack rows x 32 pack columns x T) numbered
 * [prefix[k], prefix[k+1]); `meta` is a 2-element DEVICE array {number of items, prefix[number of items]} read by
 * the kernel, so a table that is rewritten in place keeps working under a captured HIP graph. */
typedef struct crdr_pack_item {
  const float* src;
  float* dst;
  int32_t I, J, T, rows, cols, mode;
  /* sub-blocks: element (i, j, t) of the source block is src[(i * srcJ + j) * T + t] (srcJ = 0: J), i.e. `src` may point
   * at input-channel j0 of a wider parameter; pack element (t, r, c) is dst[t * tstride + r * dld + c] (0: rows * cols
   * resp. cols), i.e. `dst` may point at a row / column offset inside a wider pack shared by several parameters */
  int32_t srcJ, dld;
  int64_t tstride;
} crdr_pack_item;
int crdr_pack_weights_batched(const crdr_pack_item* items, const int64_t* prefix, const int64_t* meta, crdr_stream_t s);
/* one item (a HOST struct, modes 0 / 1, any T) */
int crdr_pack_weight_item(const crdr_pack_item* item, crdr_stream_t s);

/* ------------------------------------------------------------------------------------------------ */
/* fused elementwise / reductions (HBM-bound)                                                        */
/* ------------------------------------------------------------------------------------------------ */

/* backward of the fused conv epilogue: given dout and the saved forward output, produce dz (gradient at the
 * accumulator) plus per-channel sums.  Mirrors the flag order above in reverse.
 *   colsums layout (floats, each C long, zero-initialised by the call): [0]=sum dz (dbias) [1]=sum g after
 *   affine/res (dvec2) [2]=sum dout*u (dscale) [3]=sum dout (dshift).  gres (optional) receives the gradient
 *   flowing to res / gx when AFFINE or GATE is set (otherwise it equals dout and is not written).         */
typedef struct crdr_ebwd_desc {
  int64_t M; /* pixels */
  int32_t C;
  int32_t flags;
  int32_t lddout, ldout, lddz, ldgres, ldg;
} crdr_ebwd_desc;
typedef struct crdr_ebwd_io {
  const float* dout;
  const float* out;
  const float* vec2;
  const float* scale;
  const float* shift;
  const float* gt;  /* trunk  */
  const float* sig; /* saved sigmoid */
  float* dz;
  float* gres; /* grad to res / gx */
  float* dgt;  /* grad to trunk    */
  float* colsums; /* [4][C] */
  float* dbias_accum; /* optional [C]: += sum dz, i.e. the bias gradient goes straight into its (flat) gradient slot */
} crdr_ebwd_io;
size_t crdr_epilogue_bwd_workspace(const crdr_ebwd_desc* d);
int crdr_epilogue_bwd(const crdr_ebwd_desc* d, const crdr_ebwd_io* io, void* ws, size_t ws_bytes, crdr_stream_t s);

/* torch.nn.utils.spectral_norm on a conv weight viewed as w[O][K] (hific_discriminator.py:10-13; n_power_iterations = 1,
 * eps = 1e-12).  power_iter != 0 (training-mode forward): v <- normalize(w^T u), u <- normalize(w v), both updated in
 * place; then sigma = u . (w v) and w_out = w / sigma.  ws: (O + K + 2) floats.  The backward treats u, v as constants:
 * dw[o][k] += dw_sn[o][k] / sigma - (sum(dw_sn * w) / sigma^2) u[o] v[k];  ws: crdr_reduce_workspace(O*K) + 4 bytes. */
int crdr_spectral_norm_fwd(const float* w, int O, int K, float* u, float* v, int power_iter, float eps, float* w_out,
                           float* sigma_out, void* ws, size_t ws_bytes, crdr_stream_t s);
int crdr_spectral_norm_bwd(const float* dw_sn, const float* w, const float* u, const float* v, const float* sigma, int O, int K,
                           float* dw, void* ws, size_t ws_bytes, crdr_stream_t s);

/* Device-side input pipeline (data_transform.py:19-45: RandomCrop(size, pad_if_needed, reflect) -> HFlip -> ToTensor ->
 * (x - 0.5) / 0.5) over a pool of decoded uint8 RGB images that lives in HBM: one launch cuts a whole batch.
 * items: DEVICE array [N][6] of int64 {byte offset of the image in pool, H, W, sy0, sx0, flip}; (sy0, sx0) is the crop
 * origin in image coordinates and may be negative / reach past the image, where torchvision's reflect padding
 * (no edge repeat) applies.  out: NHWC fp32, pixel stride ldo >= 4, channel 3 zeroed. */
int crdr_crop_flip_normalize(const uint8_t* pool, const int64_t* items, int N, int crop_h, int crop_w, float* out, int ldo,
                             crdr_stream_t s);

/* Linear layer on a handful of row vectors (M <= 16: the beta-conditioning MLP and the per-block projections of the
 * [1, 512] conditioning vector, elic_interpca_beta_cond_autoencoder.py:42-84, fourier_cond.py:12-37) -- a 1x1 conv
 * over M pixels is launch-latency bound on the implicit-GEMM path, these run as GEMV-style kernels on the raw [O][I]
 * parameter (no weight pack).   y[m][o] = act(sum_i x[m][i] w[o][i] + b[o]),  act = ReLU if relu != 0. */
int crdr_linear_fwd(const float* x, int M, int I, int ldx, const float* w, const float* b, float* y, int O, int ldy,
                    int relu, crdr_stream_t s);
/* g = dy (masked by y > 0 when y is given: the ReLU of the forward);  dx[m][i] = sum_o g[m][o] w[o][i] (if dx);
 * dw[o][i] += sum_m g[m][o] x[m][i] (if dw);  db[o] += sum_m g[m][o] (if db). */
int crdr_linear_bwd(const float* x, int M, int I, int ldx, const float* w, const float* dy, int lddy, const float* y,
                    int ldy, int O, float* dx, int lddx, float* dw, float* db, crdr_stream_t s);

/* Up to CRDR_MAX_GROUP linear layers that read the SAME input rows, one launch per direction: the three projections of each of
 * the three blocks of a beta-conditioned bottleneck stack (BetaCondBaseBlock.proj_{1,2,3}, elic_interpca_beta_cond_autoencoder.py
 * :42-84; 27 projections of one [1, 512] vector per decoder pass).  y[g]: dense [M][O[g]].  Backward: dx[m][i] = sum_g sum_o
 * dy[g][m][o] w[g][o][i] (written, problems added in index order), dw[g] / db[g] accumulated (null = skipped). */
typedef struct crdr_linear_group {
  const float* w[CRDR_MAX_GROUP];
  const float* b[CRDR_MAX_GROUP]; /* may be null */
  float* y[CRDR_MAX_GROUP];       /* forward only */
  const float* dy[CRDR_MAX_GROUP]; /* backward only, dense [M][O[g]] */
  float* dw[CRDR_MAX_GROUP];
  float* db[CRDR_MAX_GROUP];
  int32_t O[CRDR_MAX_GROUP];
} crdr_linear_group;
int crdr_linear_group_fwd(const float* x, int M, int I, int ldx, const crdr_linear_group* g, int G, crdr_stream_t s);
int crdr_linear_group_bwd(const float* x, int M, int I, int ldx, const crdr_linear_group* g, int G, float* dx, int lddx,
                          crdr_stream_t s);

/* y[m][c] = x[m][c] * scale[c] + shift[c]   (stand-alone InterpChAtt apply) */
int crdr_affine(const float* x, int ldx, const float* scale, const float* shift, float* y, int ldy, int64_t M, int C,
                crdr_stream_t s);
/* out[c] (+)= sum_m x[m][c] */
size_t crdr_colsum_workspace(int64_t M, int C);
int crdr_colsum(const float* x, int ldx, int64_t M, int C, float* out, int accumulate, void* ws, size_t ws_bytes,
                crdr_stream_t s);
/* the same sums scattered block-wise: outs[c / block][c % block] (+)= sum_m x[m][c]; `outs` is a DEVICE array of
 * ceil(C / block) pointers (NULL = skip that block).  One pass over a wide activation-gradient buffer yields the bias
 * gradients of every conv whose output lives in it. */
int crdr_colsum_scatter(const float* x, int ldx, int64_t M, int C, int block, float* const* outs, int accumulate, void* ws,
                        size_t ws_bytes, crdr_stream_t s);

/* InterpChAtt parameters -> per-channel scale/shift (interp_channel_attention.py:39-73):
 *   l = floor(q), r = min(l+1, L-1), a = r - q; scale = softplus(a W[l] + (1-a) W[r]); shift = a B[l] + (1-a) B[r] */
int crdr_interp_ca_params(const float* W, const float* B, int L, int C, float q, float* scale, float* shift,
                          crdr_stream_t s);
/* dW[l] += a * dscale * sigmoid(w);  dW[r] += (1-a) * ...;  dB likewise (accumulating) */
int crdr_interp_ca_params_bwd(const float* W, int L, int C, float q, const float* dscale, const float* dshift,
                              float* dW, float* dB, crdr_stream_t s);

/* a[m][c0[k] + c] = max(a[m][c0[k] + c] + bias[k][c], 0), k < n <= CRDR_MAX_GROUP channel ranges of C channels of one NHWC
 * buffer with pixel stride ld (host arrays c0 / bias; a and the c0 are 16-byte aligned): finishes first-layer pre-activations that several
 * launches accumulated (the Charm's SliceTransform first convs, minnen20_charm_context_model.py:26-38, whose input is the
 * concatenation [hyper, support slices] -- the conv is linear in it, so each part is added by the launch that has it) */
int crdr_bias_relu_slots(float* a, int ld, int64_t M, int C, int n, const int32_t* c0, const float* const* bias, crdr_stream_t s);

/* y = a + 0.5*tanh(z)  (latent residual prediction, minnen20_charm_context_model.py:127-131) and its backward */
int crdr_lrp(const float* a, int lda, const float* z, int ldz, float* y, int ldy, int64_t M, int C, crdr_stream_t s);
int crdr_lrp_bwd(const float* dy, int lddy, const float* z, int ldz, float* dz, int lddz, int64_t M, int C,
                 crdr_stream_t s);

/* ------------------------------------------------------------------------------------------------ */
/* entropy models (CompressAI 1.2.4 semantics as called from the reference)                          */
/* ------------------------------------------------------------------------------------------------ */

/* GaussianConditional, mean+scale, STE output.  Replaces SteGaussianMeanScaleConditional.forward
 * (ste_gaussian_conditional.py:20-27) called twice per slice (noisy + quantised,
 * minnen20_charm_context_model.py:118-124) and likelihood_to_bit (hyperprior_model.py:41-46).
 *   yhat = round(y - mu) + mu
 *   lik_noisy = P(y + noise | mu, max(sigma, bound)), lik_quant = P(yhat | ...), both floored at 1e-9
 *   bits_noisy[n] += sum -log2 lik_noisy, bits_quant[n] += sum -log2 lik_quant (per image; caller zeroes)
 * noise: U(-1/2,1/2) samples, same shape as y (dense, pixel stride C); NULL = quantised outputs only (eval). */
typedef struct crdr_gc_desc {
  int32_t N;
  int32_t HW; /* pixels per image */
  int32_t C;
  int32_t ldy, ldmu, ldsigma, ldyhat;
  float scale_bound;      /* 0.11 */
  float likelihood_bound; /* 1e-9 */
} crdr_gc_desc;
int crdr_gauss_cond_fwd(const crdr_gc_desc* d, const float* y, const float* mu, const float* sigma, const float* noise,
                        float* yhat, float* lik_noisy, float* lik_quant, float* bits_noisy, float* bits_quant,
                        crdr_stream_t s);
/* backward of bits_noisy w.r.t. (y, mu, sigma) scaled by gbits[n], plus the STE path dyhat -> dy.
 * LowerBound gradients follow CompressAI's rule (pass if x >= bound or grad < 0). Outputs are dense (ld = C). */
int crdr_gauss_cond_bwd(const crdr_gc_desc* d, const float* y, const float* mu, const float* sigma, const float* noise,
                        const float* gbits, const float* dyhat, int lddyhat, float* dy, float* dmu, float* dsigma,
                        crdr_stream_t s);

/* The same two entries over channel slices of wider tensors and with in-kernel noise.
 *   io.noise != NULL : explicit U(-1/2,1/2) samples (pixel stride ldnoise) -- the parity tests' path;
 *   io.noise == NULL && io.philox != NULL : noise drawn in the kernel, Philox4x32-10 keyed by philox[0] (seed) with
 *     counter (philox[1] + global element index / 4), element index = (n * HW + px) * Ctot + c0 + c; philox is a DEVICE
 *     array of two uint64 (a captured HIP graph replays with fresh noise when the host bumps philox[1]); the backward
 *     regenerates the same samples, nothing is stored;
 *   both NULL : quantised outputs only (eval).
 * yhat2 (optional) receives a second copy of yhat (pixel stride ldyhat2). lik_* / dy, dmu, dsigma use pixel strides ldlik /
 * ldgrad (0 = C, dense). */
typedef struct crdr_gc_desc2 {
  int32_t N, HW, C;
  int32_t ldy, ldmu, ldsigma, ldyhat, ldyhat2, ldnoise, ldlik, ldgrad, lddyhat;
  int32_t Ctot, c0; /* width of the full latent and first channel of this slice (Philox indexing) */
  float scale_bound, likelihood_bound;
} crdr_gc_desc2;
typedef struct crdr_gc_io {
  const float *y, *mu, *sigma, *noise;
  const uint64_t* philox;
  float *yhat, *yhat2, *lik_noisy, *lik_quant, *bits_noisy, *bits_quant;
  const float *gbits, *dyhat;
  float *dy, *dmu, *dsigma;
  /* forward only: scratch of crdr_gauss_cond_fwd_workspace(d) bytes for the per-block partial bit sums, which a finishing pass
   * adds in fixed order (no float atomics: bit sums are bit-reproducible at any size).  NULL / too small: one block per image
   * adds in place -- same determinism, slow for large images. */
  float* ws;
  size_t ws_bytes;
} crdr_gc_io;
size_t crdr_gauss_cond_fwd_workspace(const crdr_gc_desc2* d);
int crdr_gauss_cond_fwd2(const crdr_gc_desc2* d, const crdr_gc_io* io, crdr_stream_t s);
int crdr_gauss_cond_bwd2(const crdr_gc_desc2* d, const crdr_gc_io* io, crdr_stream_t s);
/* U(-1/2, 1/2) samples of the generator above written out: out[(n * HW + px) * ld + c] for c < C (tests, and the noisy
 * latent the reference's non-STE GaussianConditional returns) */
/* What the host rANS coder consumes, straight from the device tensors: symbols[n][c][p] = (int32) round(y - mu) and
 * indexes[n][c][p] = number of scale-table entries below max(sigma, scale_bound), capped at levels - 1 (compressai
 * GaussianConditional.quantize(.., "symbols", means) / build_indexes, minnen20_charm_context_model.py:186-187,197-199), in
 * (channel, pixel) order so that one contiguous device-to-host copy per array suffices.  Either output may be NULL. */
int crdr_gauss_symbols(const float* y, int ldy, const float* mu, int ldmu, const float* sigma, int ldsigma, const float* scale_table,
                       int levels, float scale_bound, int N, int HW, int C, int32_t* symbols, int32_t* indexes, crdr_stream_t s);
/* call[0..1] = state[0..1]; state[1] += inc -- one launch hands a forward pass its own (seed, offset) pair (kept for the
 * backward) and moves the generator on, so a captured HIP graph draws fresh noise at every replay */
int crdr_philox_fork(uint64_t* state, uint64_t* call, uint64_t inc, crdr_stream_t s);
int crdr_philox_uniform(const uint64_t* philox, int N, int HW, int C, int Ctot, int c0, float* out, int ld, crdr_stream_t s);

/* factorised prior (EntropyBottleneck, filters (3,3,3,3)); parameters are passed as one packed block per
 * channel made by the host: see crdr_amd/models/subnet/entropy_model. Replaces SteEntropyBottleneck.forward
 * (entropy_bottleneck.py:18-30).  z is NHWC [N][HW][C]; noise NULL => quantised. */
#define CRDR_EB_PARAMS 58 /* 3+3+3 + 9+3+3 + 9+3+3 + 9+3+3 + 3+1 = 58 floats per channel */
int crdr_entropy_bottleneck_fwd(const float* z, const float* noise, const float* params, const float* medians, int N,
                                int HW, int C, float likelihood_bound, float* zhat, float* lik, float* bits,
                                crdr_stream_t s);
int crdr_entropy_bottleneck_bwd(const float* z, const float* noise, const float* params, int N, int HW, int C,
                                float likelihood_bound, const float* gbits, const float* dzhat, float* dz,
                                float* dparams, crdr_stream_t s);

/* aux ("quantile") loss of the factorised prior, compressai EntropyBottleneck.loss() as called from
 * base_model.py:68-78: loss = sum_{c,j} |logits_c(quantiles[c][j]) - target[j]|, dq = d loss / d quantiles
 * (density parameters are constants here). */
int crdr_eb_quantile_loss(const float* quantiles, const float* params, const float* target, int C, float* loss,
                          float* dquantiles, crdr_stream_t s);

/* ------------------------------------------------------------------------------------------------ */
/* GDN / IGDN (optional registered op: only the reference's Balle18 / Cheng20 ablation transforms use it,          */
/* balle18_autoencoder.py:16-20,37-41, cheng_resblock.py:8-15, through compressai.layers.GDN)                     */
/*   n[p][i] = beta_i + sum_j gamma_ij x[p][j]^2 ;  y = x / sqrt(n)  (inverse = 0)  or  y = x * sqrt(n)  (inverse = 1)  */
/*   beta, gamma are the STORED parameters of compressai's NonNegativeParametrizer:                                  */
/*   v_eff = max(v, bound)^2 - reparam_offset^2, bound_beta = sqrt(beta_min + reparam_offset^2), bound_gamma =        */
/*   reparam_offset (defaults beta_min 1e-6, reparam_offset 2^-18); the backward applies the LowerBound gradient rule.  */
/*   C <= 192: forward and backward are fused persistent kernels (gamma in registers, one pass over x / over x and dy);  */
/*   larger C: the channel mix runs as a 1x1 launch of the MFMA conv kernel on x^2.  dbeta / dgamma ACCUMULATE.          */
/* ------------------------------------------------------------------------------------------------ */
typedef struct crdr_gdn_desc {
  int64_t M; /* pixels */
  int32_t C, ldx, ldy, inverse;
  float beta_min, reparam_offset;
} crdr_gdn_desc;
size_t crdr_gdn_workspace(const crdr_gdn_desc* d, int backward); /* 256-byte aligned buffer of this many bytes */
int crdr_gdn_fwd(const crdr_gdn_desc* d, const float* x, const float* beta, const float* gamma, float* y, void* ws, size_t ws_bytes,
                 crdr_stream_t s);
int crdr_gdn_bwd(const crdr_gdn_desc* d, const float* x, const float* beta, const float* gamma, const float* dy, int lddy, float* dx,
                 int lddx, float* dbeta, float* dgamma, void* ws, size_t ws_bytes, crdr_stream_t s);

/* ------------------------------------------------------------------------------------------------ */
/* losses                                                                                            */
/* ------------------------------------------------------------------------------------------------ */
/* out[0] (+)= sum (a-b)^2 ; backward da = 2 (a-b) g, db = -da   (distortion_loss.py:41-46)         */
size_t crdr_reduce_workspace(int64_t n);
int crdr_sqdiff_sum(const float* a, const float* b, int64_t n, float* out, void* ws, size_t ws_bytes, crdr_stream_t s);
int crdr_sqdiff_bwd(const float* a, const float* b, int64_t n, const float* g, float gscale, float* da, float* db,
                    crdr_stream_t s);
/* BCEWithLogits(sign * (p - q), target) summed: out[0] = sum softplus-form loss; gradient wrt p and q
 * (gan_loss.py:28-31 applied to logit differences, multirate_hr_rgan_beta_cond_rate_distortion_trainer.py:56-62) */
int crdr_bce_diff_sum(const float* p, const float* q, int64_t n, float target, float* out, void* ws, size_t ws_bytes,
                      crdr_stream_t s);
int crdr_bce_diff_bwd(const float* p, const float* q, int64_t n, float target, const float* g, float gscale, float* dp,
                      float* dq, crdr_stream_t s);

/* ------------------------------------------------------------------------------------------------ */
/* optimiser (build_optimizer_scheduler.py:11-19; rate_distortion_trainer.py:83-85)                  */
/* ------------------------------------------------------------------------------------------------ */
/* out[0] = sum g^2 (deterministic two-stage) */
int crdr_sqnorm(const float* g, int64_t n, float* out, void* ws, size_t ws_bytes, crdr_stream_t s);
/* torch.optim.Adam step (no amsgrad, no weight decay) on a flat buffer; the gradient is multiplied by
 * min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) when sqnorm != NULL (clip_grad_norm_ semantics).              */
int crdr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, int step, const float* sqnorm, float max_norm, crdr_stream_t s);

/* The same update with lr / step count / skip flag read from device memory (dyn[0..2]) so that the launch can be
 * captured in a HIP graph and replayed: dyn[1] is the already-incremented step count, dyn[2] != 0 skips the update
 * (the reference's NaN / Inf / huge-loss gate, base_trainer.py:228-238, taken on the device). */
int crdr_adam_step_dyn(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                       const float* dyn, const float* sqnorm, float max_norm, crdr_stream_t s);

/* ------------------------------------------------------------------------------------------------ */
/* LPIPS helpers (perceptual_loss.py:25-30; lpips 0.1.4 AlexNet)                                      */
/* ------------------------------------------------------------------------------------------------ */
int crdr_maxpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, crdr_stream_t s);
int crdr_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, crdr_stream_t s);
/* per-pixel: d = sum_c lin[c] * (f0/|f0| - f1/|f1|)^2 ; out[n] += mean_pixels d */
/* ws: N*64 floats */
int crdr_lpips_layer_fwd(const float* f0, const float* f1, const float* lin, int N, int HW, int C, float* out,
                         void* ws, size_t ws_bytes, crdr_stream_t s);
int crdr_lpips_layer_bwd(const float* f0, const float* f1, const float* lin, int N, int HW, int C, const float* gout,
                         float* df1, crdr_stream_t s);

/* ------------------------------------------------------------------------------------------------ */
/* host-side entropy coder (third-party in the reference: compressai 1.2.4 `compressai.ans`,         */
/* call sites hyperprior_model.py:150-155,190-198; minnen20_charm_context_model.py:186-187,201-224)  */
/* ------------------------------------------------------------------------------------------------ */
/* cdf_out has pmf_len+1 entries. returns 0 / <0 */
int crdr_pmf_to_quantized_cdf(const float* pmf, int pmf_len, int precision, uint32_t* cdf_out);
/* encode; cdfs is [ncdf][cdf_stride] int32. returns bytes written, or -needed if out_cap too small, < -2^30 on error */
int64_t crdr_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                      int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int ncdf,
                                      uint8_t* out, int64_t out_cap);
typedef struct crdr_rans_decoder crdr_rans_decoder;
crdr_rans_decoder* crdr_rans_decoder_create(void);
void crdr_rans_decoder_destroy(crdr_rans_decoder*);
int crdr_rans_decoder_set_stream(crdr_rans_decoder*, const uint8_t* data, int64_t nbytes);
int crdr_rans_decoder_decode_stream(crdr_rans_decoder*, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                    int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int ncdf,
                                    int32_t* out);
int crdr_rans_decode_with_indexes(const uint8_t* data, int64_t nbytes, const int32_t* indexes, int64_t n,
                                  const int32_t* cdfs, int cdf_stride, const int32_t* cdf_sizes,
                                  const int32_t* offsets, int ncdf, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* CRDR_HIP_H */
