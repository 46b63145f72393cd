// inflate_common.hpp — pieces shared by the inflate kernels (gfx950).
// RFC1951 constant tables (reference lib/de.ml:237-325), the packed LUT entry
// format and the LUT builder (De.Inf.huffman, lib/de.ml:523-638).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdeflate.h"

namespace md {

constexpr int kWave = 64;
constexpr uint32_t kFlush = 1024;  // flush granularity in output bytes
constexpr uint32_t kNearSlack = 128;

// 16-bit LUT entry: leaf  = sym[8:0] | len[12:9]
//                   link  = off[9:0] | sub[13:10] | 0x8000
//                   kBad  = unreachable slot of an incomplete table
constexpr uint16_t kLink = 0x8000;
constexpr uint16_t kBad = 0x7fff;

enum { K_CODES = 0, K_LENS = 1, K_DISTS = 2 };

// RFC1951 tables, reference lib/de.ml:237-325 (base_length is "+3"-biased there).
__constant__ uint8_t c_zigzag[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
__constant__ uint16_t c_base_length[32] = {0,  1,  2,  3,  4,  5,  6,   7,   8,   10,  12,
                                           14, 16, 20, 24, 28, 32, 40,  48,  56,  64,  80,
                                           96, 112, 128, 160, 192, 224, 255, 0,   0,   0};
__constant__ uint8_t c_extra_lbits[32] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                          3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0};
__constant__ uint8_t c_extra_dbits[32] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,  4,
                                          4, 5, 5, 6, 6, 7, 7,  8,  8,  9,  9,
                                          10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
// base_dist + 1 (lib/de.ml:321-325); entries 30,31 are -1+1 = 0 => "Invalid distance code"
__constant__ uint16_t c_base_dist1[32] = {1,    2,    3,    4,    5,    7,     9,     13,
                                          17,   25,   33,   49,   65,   97,    129,   193,
                                          257,  385,  513,  769,  1025, 1537,  2049,  3073,
                                          4097, 6145, 8193, 12289, 16385, 24577, 0,   0};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

struct Lut {
  uint16_t *t;
  uint32_t mask;  // (1 << root) - 1
  uint32_t root;
  uint32_t maxl;  // longest code (Lookup.l)
};

// Per-wavefront LDS block (dynamic shared memory): ring first, tables after.
struct Scratch {
  uint16_t lit[852];
  uint16_t dist[592];
  uint16_t codes[128];
  uint16_t work[320];
  uint8_t lens[320];
  uint16_t cnt[16];
  uint16_t offs[16];
};

// ---------------------------------------------------------------------------
// LUT construction: De.Inf.huffman (lib/de.ml:523-638) with 16-bit entries.
// Runs wave-uniform; lane 0 performs the LDS stores.  Returns false for
// Invalid_huffman.  `lens` are the code lengths of `codes` symbols.
__device__ __noinline__ bool build_lut(int kind, const uint8_t *lens, uint32_t codes, Scratch *s, Lut *out,
                          uint32_t lane, uint16_t *tbl_override = nullptr) {
  // tbl_override: build the lit/len or distance table somewhere else than Scratch::lit/dist
  // (the split decode kernel builds them inside its fat-LUT area and never touches s->lit/dist)
  uint16_t *tbl = tbl_override ? tbl_override : kind == K_LENS ? s->lit : kind == K_DISTS ? s->dist : s->codes;
  uint16_t *cnt = s->cnt, *offs = s->offs, *work = s->work;
  if (lane < 16) cnt[lane] = 0;
  // histogram of code lengths (one lane: the table is tiny)
  if (lane == 0)
    for (uint32_t sym = 0; sym < codes; sym++) cnt[lens[sym]]++;
  uint32_t max = 15, min = 1;
  while (max >= 1 && uni(cnt[max]) == 0) max--;
  if (max == 0) {
    // empty_table (lib/de.ml:521): one 1-bit code for symbol 0; the other slot is
    // out of bounds in the reference (documented divergence D2: reported as an error)
    if (lane == 0) {
      tbl[0] = (uint16_t)((1u << 9) | 0);
      tbl[1] = kBad;
    }
    out->t = tbl;
    out->mask = 1;
    out->root = 1;
    out->maxl = 1;
    return true;
  }
  int left = 1;
  for (uint32_t i = 1; i <= 15; i++) {
    left = (left << 1) - (int)uni(cnt[i]);
    if (left < 0) return false;
  }
  if (left > 0 && (kind == K_CODES || max != 1)) return false;
  while (min <= 15 && uni(cnt[min]) == 0) min++;
  if (lane == 0) {
    offs[0] = 0;
    offs[1] = 0;
    for (uint32_t i = 1; i <= 14; i++) offs[i + 1] = offs[i] + cnt[i];
    for (uint32_t sym = 0; sym < codes; sym++) {
      uint32_t l = lens[sym];
      if (l) work[offs[l]++] = (uint16_t)sym;
    }
  }
  uint32_t root = kind == K_LENS ? 9 : kind == K_DISTS ? 6 : 7;
  if (root > max) root = max;
  if (root < min) root = min;
  uint32_t size;
  if (max <= root) size = 1u << max;
  else size = kind == K_LENS ? 852 : kind == K_DISTS ? 592 : (1u << max);
  for (uint32_t i = lane; i < size; i += kWave) tbl[i] = 0;

  uint32_t huff = 0, sym = 0, len = min, next = 0, curr = root, drop = 0;
  int low = -1;
  const uint32_t mask = (1u << root) - 1;
  bool finished = false;
  while (!finished) {
    uint32_t value = uni(work[sym]);
    uint16_t entry = (uint16_t)((len << 9) | value);
    uint32_t step = 1u << (len - drop);
    uint32_t fill_size = 1u << curr;
    uint32_t base = next + (huff >> drop);
    if (base + fill_size - step >= size) return false;  // D3
    // replicate: indices base + k*step, k < fill_size/step — spread over the lanes
    for (uint32_t k = lane; k * step < fill_size; k += kWave) tbl[base + k * step] = entry;
    uint32_t inc = 1u << (len - 1);
    while (huff & inc) inc >>= 1;
    huff = inc ? (huff & (inc - 1)) + inc : 0;
    sym++;
    uint32_t c = uni(cnt[len]) - 1;
    if (lane == 0) cnt[len] = (uint16_t)c;
    if (c == 0) {
      if (len == max) finished = true;
      else len = uni(lens[uni(work[sym])]);
    }
    if (!finished && len > root && (int)(huff & mask) != low) {
      if (drop == 0) drop = root;
      next += fill_size;
      curr = len - drop;
      int l2 = 1 << curr;
      while (curr + drop < max) {
        l2 -= (int)uni(cnt[curr + drop]);
        if (l2 <= 0) break;
        curr++;
        l2 <<= 1;
      }
      low = (int)(huff & mask);
      if (next + (1u << curr) > size) return false;  // D3
      if (lane == 0) tbl[low] = (uint16_t)(kLink | (curr << 10) | next);
    }
  }
  // an incomplete 1-bit code leaves slot 1 empty: the reference reads a zero
  // entry there (len 0, sym 0); keep that behaviour (tbl is zero-filled).
  out->t = tbl;
  out->mask = mask;
  out->root = root;
  out->maxl = max;
  return true;
}

__device__ __forceinline__ uint32_t lut_lookup(const Lut &l, uint64_t hold) {
  uint32_t e = uni(l.t[(uint32_t)hold & l.mask]);
  if (e & kLink) {
    uint32_t sub = (e >> 10) & 15;
    uint32_t off = e & 1023;
    e = uni(l.t[off + (((uint32_t)(hold >> l.root)) & ((1u << sub) - 1))]);
  }
  return e;
}


}  // namespace md
