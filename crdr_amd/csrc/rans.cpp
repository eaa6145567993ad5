// Host-side range-ANS entropy coder + pmf quantiser.
//
// The reference does not contain this code: it calls the third-party package compressai==1.2.4
// (pyproject.toml:16) -- `compressai.ans.RansEncoder/RansDecoder` and `compressai._CXX.pmf_to_quantized_cdf`
// -- at hyperprior_model.py:150-155,190-198 and minnen20_charm_context_model.py:186-187,201-224.  This file
// restates that published algorithm (ryg_rans rans64 with 32-bit renormalisation words, 16-bit CDF precision,
// and a 4-bit "bypass" escape for symbols outside the table) so bitstreams have the same structure.
// PARITY UNPINNED: no compressai install or bitstream fixture exists in the reference tree to confirm
// byte-for-byte equality; round-trip identity and analytic cases are what the tests pin.

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <numeric>
#include <vector>

#include "crdr_hip.h"

namespace crdr {
void set_error(const char* fmt, ...);
}

namespace {

constexpr int kPrecision = 16;
constexpr int kBypassPrecision = 4;
constexpr int kMaxBypassVal = (1 << kBypassPrecision) - 1;
constexpr uint64_t kRansL = 1ull << 31;

struct Sym {
  uint16_t start, range;
  bool bypass;
};

inline void enc_put(uint64_t& x, uint32_t*& ptr, uint32_t start, uint32_t freq, uint32_t scale_bits) {
  const uint64_t x_max = ((kRansL >> scale_bits) << 32) * freq;
  if (x >= x_max) {
    *--ptr = (uint32_t)x;
    x >>= 32;
  }
  x = ((x / freq) << scale_bits) + (x % freq) + start;
}
inline void enc_put_bits(uint64_t& x, uint32_t*& ptr, uint32_t val, uint32_t nbits) {
  const uint32_t freq = 1u << (16 - nbits);
  const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
  if (x >= x_max) {
    *--ptr = (uint32_t)x;
    x >>= 32;
  }
  x = (x << nbits) | val;
}

struct Decoder {
  std::vector<uint32_t> words;
  size_t pos = 0;
  uint64_t x = 0;
  bool ready = false;

  inline uint32_t next_word() { return pos < words.size() ? words[pos++] : 0u; }
  inline uint32_t get(uint32_t scale_bits) const { return (uint32_t)(x & ((1u << scale_bits) - 1)); }
  inline void advance(uint32_t start, uint32_t freq, uint32_t scale_bits) {
    const uint64_t mask = (1ull << scale_bits) - 1;
    x = freq * (x >> scale_bits) + (x & mask) - start;
    if (x < kRansL) x = (x << 32) | next_word();
  }
  inline uint32_t get_bits(uint32_t nbits) {
    const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < kRansL) x = (x << 32) | next_word();
    return val;
  }
};

int decode_n(Decoder& d, const int32_t* indexes, int64_t n, const int32_t* cdfs, int cdf_stride,
             const int32_t* cdf_sizes, const int32_t* offsets, int ncdf, int32_t* out) {
  if (!d.ready) { crdr::set_error("rans decode: no stream set"); return -1; }
  for (int64_t i = 0; i < n; ++i) {
    const int32_t ci = indexes[i];
    if (ci < 0 || ci >= ncdf) { crdr::set_error("rans decode: cdf index %d out of range", ci); return -1; }
    const int32_t* cdf = cdfs + (size_t)ci * cdf_stride;
    const int32_t size = cdf_sizes[ci];
    const int32_t max_value = size - 2;
    const uint32_t cum = d.get(kPrecision);
    int32_t s = 0;  // first entry greater than cum, minus one
    while (s < size && (uint32_t)cdf[s] <= cum) ++s;
    s -= 1;
    if (s < 0 || s + 1 >= size) { crdr::set_error("rans decode: corrupt stream"); return -1; }
    d.advance((uint32_t)cdf[s], (uint32_t)(cdf[s + 1] - cdf[s]), kPrecision);
    int32_t value = s;
    if (value == max_value) {
      int32_t val = (int32_t)d.get_bits(kBypassPrecision);
      int32_t n_bypass = val;
      while (val == kMaxBypassVal) {
        val = (int32_t)d.get_bits(kBypassPrecision);
        n_bypass += val;
      }
      int32_t raw = 0;
      for (int j = 0; j < n_bypass; ++j) {
        val = (int32_t)d.get_bits(kBypassPrecision);
        raw |= val << (j * kBypassPrecision);
      }
      value = raw >> 1;
      if (raw & 1) value = -value - 1;
      else value += max_value;
    }
    out[i] = value + offsets[ci];
  }
  return 0;
}

int set_stream(Decoder& d, const uint8_t* data, int64_t nbytes) {
  if (nbytes < 8 || (nbytes % 4) != 0) { crdr::set_error("rans: stream of %lld bytes is not a rans64 stream", (long long)nbytes); return -1; }
  d.words.resize((size_t)nbytes / 4);
  memcpy(d.words.data(), data, (size_t)nbytes);
  d.x = (uint64_t)d.words[0] | ((uint64_t)d.words[1] << 32);
  d.pos = 2;
  d.ready = true;
  return 0;
}

}  // namespace

struct crdr_rans_decoder {
  Decoder d;
};

extern "C" int crdr_pmf_to_quantized_cdf(const float* pmf, int pmf_len, int precision, uint32_t* cdf) {
  if (!pmf || !cdf || pmf_len <= 0 || precision <= 0 || precision > 16) { crdr::set_error("pmf_to_quantized_cdf: bad arguments"); return -1; }
  for (int i = 0; i < pmf_len; ++i)
    if (!(pmf[i] >= 0.f) || !std::isfinite(pmf[i])) { crdr::set_error("pmf_to_quantized_cdf: invalid pmf[%d]", i); return -1; }
  const int n = pmf_len + 1;
  cdf[0] = 0;
  for (int i = 0; i < pmf_len; ++i) cdf[i + 1] = (uint32_t)std::round(pmf[i] * (float)(1 << precision));
  uint32_t total = 0;
  for (int i = 0; i < n; ++i) total += cdf[i];
  if (total == 0) { crdr::set_error("pmf_to_quantized_cdf: zero total frequency"); return -1; }
  for (int i = 0; i < n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);
  for (int i = 1; i < n; ++i) cdf[i] += cdf[i - 1];
  cdf[n - 1] = 1u << precision;
  for (int i = 0; i < n - 1; ++i) {
    if (cdf[i] == cdf[i + 1]) {  // zero-frequency symbol: steal one count from the smallest symbol with freq > 1
      uint32_t best_freq = ~0u;
      int best = -1;
      for (int j = 0; j < n - 1; ++j) {
        const uint32_t freq = cdf[j + 1] - cdf[j];
        if (freq > 1 && freq < best_freq) { best_freq = freq; best = j; }
      }
      if (best < 0) { crdr::set_error("pmf_to_quantized_cdf: cannot repair zero frequency"); return -1; }
      if (best < i) { for (int j = best + 1; j <= i; ++j) cdf[j]--; }
      else { for (int j = i + 1; j <= best; ++j) cdf[j]++; }
    }
  }
  return 0;
}

extern "C" int64_t crdr_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n,
                                                 const int32_t* cdfs, int cdf_stride, const int32_t* cdf_sizes,
                                                 const int32_t* offsets, int ncdf, uint8_t* out, int64_t out_cap) {
  const int64_t kErr = -(1ll << 40);
  if (n < 0 || (n > 0 && (!symbols || !indexes)) || !cdfs || !cdf_sizes || !offsets) { crdr::set_error("rans encode: null pointer"); return kErr; }
  std::vector<Sym> syms;
  syms.reserve((size_t)n + 16);
  for (int64_t i = 0; i < n; ++i) {
    const int32_t ci = indexes[i];
    if (ci < 0 || ci >= ncdf) { crdr::set_error("rans encode: cdf index %d out of range", ci); return kErr; }
    const int32_t* cdf = cdfs + (size_t)ci * cdf_stride;
    const int32_t max_value = cdf_sizes[ci] - 2;
    if (max_value < 0 || cdf_sizes[ci] > cdf_stride) { crdr::set_error("rans encode: bad cdf size"); return kErr; }
    int32_t value = symbols[i] - offsets[ci];
    uint32_t raw = 0;
    if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
    else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
    syms.push_back({(uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), false});
    if (value == max_value) {
      int32_t n_bypass = 0;
      while ((raw >> (n_bypass * kBypassPrecision)) != 0) ++n_bypass;
      int32_t val = n_bypass;
      while (val >= kMaxBypassVal) { syms.push_back({(uint16_t)kMaxBypassVal, (uint16_t)(kMaxBypassVal + 1), true}); val -= kMaxBypassVal; }
      syms.push_back({(uint16_t)val, (uint16_t)(val + 1), true});
      for (int32_t j = 0; j < n_bypass; ++j) {
        const int32_t v = (int32_t)((raw >> (j * kBypassPrecision)) & kMaxBypassVal);
        syms.push_back({(uint16_t)v, (uint16_t)(v + 1), true});
      }
    }
  }
  std::vector<uint32_t> buf(syms.size() + 2, 0xCCu);
  uint32_t* ptr = buf.data() + buf.size();
  uint64_t x = kRansL;
  for (size_t k = syms.size(); k-- > 0;) {
    const Sym& s = syms[k];
    if (!s.bypass) {
      if (s.range == 0) { crdr::set_error("rans encode: zero-frequency symbol"); return kErr; }
      enc_put(x, ptr, s.start, s.range, kPrecision);
    } else {
      enc_put_bits(x, ptr, s.start, kBypassPrecision);
    }
  }
  ptr -= 2;
  ptr[0] = (uint32_t)x;
  ptr[1] = (uint32_t)(x >> 32);
  const int64_t nbytes = (int64_t)(buf.data() + buf.size() - ptr) * 4;
  if (!out || nbytes > out_cap) return -nbytes;
  memcpy(out, ptr, (size_t)nbytes);
  return nbytes;
}

extern "C" crdr_rans_decoder* crdr_rans_decoder_create(void) { return new crdr_rans_decoder(); }
extern "C" void crdr_rans_decoder_destroy(crdr_rans_decoder* d) { delete d; }
extern "C" int crdr_rans_decoder_set_stream(crdr_rans_decoder* d, const uint8_t* data, int64_t nbytes) {
  if (!d || !data) { crdr::set_error("rans: null pointer"); return -1; }
  return set_stream(d->d, data, nbytes);
}
extern "C" int crdr_rans_decoder_decode_stream(crdr_rans_decoder* d, const int32_t* indexes, int64_t n,
                                               const int32_t* cdfs, int cdf_stride, const int32_t* cdf_sizes,
                                               const int32_t* offsets, int ncdf, int32_t* out) {
  if (!d || !indexes || !cdfs || !cdf_sizes || !offsets || !out) { crdr::set_error("rans: null pointer"); return -1; }
  return decode_n(d->d, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, ncdf, out);
}
extern "C" int crdr_rans_decode_with_indexes(const uint8_t* data, int64_t nbytes, const int32_t* indexes, int64_t n,
                                             const int32_t* cdfs, int cdf_stride, const int32_t* cdf_sizes,
                                             const int32_t* offsets, int ncdf, int32_t* out) {
  if (!data || !indexes || !cdfs || !cdf_sizes || !offsets || !out) { crdr::set_error("rans: null pointer"); return -1; }
  Decoder d;
  if (int rc = set_stream(d, data, nbytes)) return rc;
  return decode_n(d, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, ncdf, out);
}
