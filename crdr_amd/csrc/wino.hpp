// Winograd F(2x2, 3x3) path of crdr_conv2d (wino.hip), planned and launched from igemm.hip.
#pragma once
#include "common.hpp"
#include "igemm_args.hpp"

namespace crdr {

bool wino_eligible(const crdr_conv_desc* d, int G);
size_t wino_workspace(const crdr_conv_desc* d, int G);   // bytes of transformed filters
int wino_colsum_rows(const crdr_conv_desc* d);
bool wino_pairs_ok(const crdr_conv_desc* d);   // variant 1 applies
int wino_launch(const crdr_conv_desc* d, int variant, IgemmArgs a, const IgemmTaps& taps, const IgemmGroup& grp, int G, float* u, hipStream_t s);

// Winograd F(4x4, 3x3) path (wino4.hip): variant 2 of the forced Winograd ids.  vec_ok: every operand row is 16-byte aligned (known at
// launch; planning passes true)
bool wino4_eligible(const crdr_conv_desc* d, int G, bool vec_ok);
size_t wino4_workspace(const crdr_conv_desc* d, int G, int nsplit);   // bytes of transformed filters (+ the partial tiles of a K-split launch)
bool wino4_split_ok(const crdr_conv_desc* d, int G, int nsplit);      // nsplit K splits per tile (forced id: bits 8..11 = nsplit - 1)
int wino4_colsum_rows(const crdr_conv_desc* d);
int wino4_filter_item(const crdr_conv_desc* d, const IgemmTaps& taps, int G, crdr_w4_filter_item* it);   // crdr_conv2d_filter_item
int wino4_filters_batched(const crdr_w4_filter_item* items, const long long* prefix, const long long* meta, hipStream_t s);
int wino4_launch(const crdr_conv_desc* d, IgemmArgs a, const IgemmTaps& taps, const IgemmGroup& grp, int G, float* u, float* slabs, int nsplit,
                 bool filters_ready, hipStream_t s);

}  // namespace crdr
