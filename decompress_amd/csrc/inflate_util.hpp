// inflate_util.hpp — small device helpers of the wave inflate kernel (gfx950): LDS / output-buffer
// copies with LZ77 forward-byte semantics, wave scans, the optional in-kernel phase profile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdeflate.h"

namespace md {
namespace wv {

constexpr int kWave = 64;
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

#define MD_LDS __attribute__((address_space(3)))
typedef MD_LDS uint32_t lds_u32;
typedef MD_LDS uint16_t lds_u16;
typedef MD_LDS uint8_t lds_u8;
typedef uint64_t u64_u __attribute__((aligned(1)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef uint16_t u16_u __attribute__((aligned(1)));

enum { P_ENSURE = 0, P_DECODE1, P_DECODE2, P_EMIT_A, P_FAR, P_NEAR, P_ADLER, P_HEADER, P_NEAR_FAST, P_NEAR_SLOW, P_NEAR_UPD, P_NEAR_LOAD, P_HDR_LENS, P_HDR_LIT, P_FAR_REC, P_FAR_LOAD, P_COUNT };
enum { C_ROUNDS = 0, C_PASSES, C_LANES, C_TOKENS, C_SLOTS, C_NEAR_IT,
       C_END_CHAIN, C_END_FIT, C_END_RECORDS, C_END_STAGE, C_END_EOB, C_LONG_NEAR, C_COUNT };  // why rounds ended short
template <bool ON>
struct Prof {
  static constexpr bool on = true;
  uint64_t t0;
  uint64_t acc[P_COUNT];
  uint32_t cnt[C_COUNT];
  __device__ __forceinline__ void init() {
    for (int i = 0; i < P_COUNT; i++) acc[i] = 0;
    for (int i = 0; i < C_COUNT; i++) cnt[i] = 0;
    t0 = clock64();
  }
  __device__ __forceinline__ void tick(int i) {
    uint64_t t = clock64();
    acc[i] += t - t0;
    t0 = t;
  }
  __device__ __forceinline__ void tick_lds(int i) {  // after the LDS queue has drained: the phase pays for its own accesses
    __builtin_amdgcn_s_waitcnt(0xc07f);
    tick(i);
  }
  __device__ __forceinline__ void tick_all(int i) {  // after every outstanding memory access has completed
    __builtin_amdgcn_s_waitcnt(0);
    tick(i);
  }
  __device__ __forceinline__ void count(int i, uint32_t n = 1) { cnt[i] += n; }
};
template <>
struct Prof<false> {
  static constexpr bool on = false;
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ void tick(int) {}
  __device__ __forceinline__ void tick_lds(int) {}
  __device__ __forceinline__ void tick_all(int) {}
  __device__ __forceinline__ void count(int, uint32_t = 1) {}
};

// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane) {
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += t;
  }
  return x - v;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l));
}

// L1-bypassing (nt) accesses to the output buffer: it is written and re-read by
// different lanes of this wavefront, and the same 128-B line can be cached by the
// CU's L1 before a later store completes it.
__device__ __forceinline__ uint64_t out_ld64(const uint8_t *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u64_u *>(p));
}
__device__ __forceinline__ uint32_t out_ld8(const uint8_t *p) { return __builtin_nontemporal_load(p); }
// n (1..8) bytes at g[x ..) without touching g[cap ..)
__device__ __forceinline__ uint64_t out_ld_guard(const uint8_t *g, uint32_t x, uint32_t n, uint32_t cap) {
  if (x + 8 <= cap) return out_ld64(g + x);
  uint64_t v = 0;
  for (uint32_t j = 0; j < n && j < 8; j++) v |= (uint64_t)out_ld8(g + x + j) << (8 * j);
  return v;
}
// store the low n (1..8) bytes of v
__device__ __forceinline__ void out_st(uint8_t *p, uint64_t v, uint32_t n) {
  if (n >= 8) {
    *reinterpret_cast<u64_u *>(p) = v;
    return;
  }
  if (n & 4) {
    *reinterpret_cast<u32_u *>(p) = (uint32_t)v;
    p += 4;
    v >>= 32;
  }
  if (n & 2) {
    *reinterpret_cast<u16_u *>(p) = (uint16_t)v;
    p += 2;
    v >>= 16;
  }
  if (n & 1) *p = (uint8_t)v;
}
// LZ77 copy of ml bytes to g[q ..) from d bytes back.  Every source byte is read
// from [q-d, q) — final and visible — never from bytes this copy writes itself.
// Loads of up to 32 bytes are in flight together.
__device__ __forceinline__ void copy_match(uint8_t *g, uint32_t q, uint32_t ml, uint32_t d, uint32_t cap) {
  const uint32_t src = q - d;
  if (d >= 8) {
    for (uint32_t o = 0; o < ml; o += d) {  // periodic: dst[o + j] = orig[j], j < d
      const uint32_t n = ml - o < d ? ml - o : d;
      for (uint32_t j = 0; j < n; j += 32) {
        const uint32_t m = n - j;
        uint64_t v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) v[u] = m > 8 * u ? out_ld_guard(g, src + j + 8 * u, m - 8 * u, cap) : 0;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++)
          if (m > 8 * u) out_st(g + q + o + j + 8 * u, v[u], m - 8 * u);
      }
    }
  } else {
    // period d < 8: replicate the last d bytes into a 64-bit pattern
    uint64_t v = out_ld_guard(g, src, d, cap);
    const uint32_t sh = 8 * d;
    v &= (1ull << sh) - 1;
    v |= v << sh;
    if (2 * sh < 64) v |= v << (2 * sh);
    if (4 * sh < 64) v |= v << (4 * sh);
    const uint32_t adv = d * (8 / d);  // largest multiple of the period that fits 8 bytes
    uint32_t j = 0;
    for (; j + 8 <= ml; j += adv) out_st(g + q + j, v, 8);
    if (j < ml) out_st(g + q + j, v, ml - j);
  }
}

__device__ __forceinline__ uint64_t lds_ld64(const lds_u8 *p) { return *reinterpret_cast<const MD_LDS u64_u *>(p); }
__device__ __forceinline__ void lds_st64(lds_u8 *p, uint64_t v) { *reinterpret_cast<MD_LDS u64_u *>(p) = v; }
// store the low r (< 8) bytes of v
__device__ __forceinline__ void lds_st_tail(lds_u8 *p, uint64_t v, uint32_t r) {
  if (r & 4) {
    *reinterpret_cast<MD_LDS u32_u *>(p) = (uint32_t)v;
    p += 4;
    v >>= 32;
  }
  if (r & 2) {
    *reinterpret_cast<MD_LDS u16_u *>(p) = (uint16_t)v;
    p += 2;
    v >>= 16;
  }
  if (r & 1) *p = (uint8_t)v;
}
__device__ __forceinline__ void lds_st(lds_u8 *p, uint64_t v, uint32_t n) {
  if (n >= 8) lds_st64(p, v);
  else lds_st_tail(p, v, n);
}
// staging -> staging LZ77 copy with forward-byte semantics (overlap allowed)
__device__ __forceinline__ void copy_near(lds_u8 *dst, const lds_u8 *src, uint32_t ml, uint32_t d) {
  if (d >= 8) {
    uint32_t j = 0;
    for (; j + 8 <= ml; j += 8) lds_st64(dst + j, lds_ld64(src + j));
    if (j < ml) lds_st_tail(dst + j, lds_ld64(src + j), ml - j);
  } else {
    // period d < 8: replicate the last d bytes into a 64-bit pattern
    uint64_t v = lds_ld64(src);
    const uint32_t sh = 8 * d;
    v &= (1ull << sh) - 1;
    v |= v << sh;
    if (2 * sh < 64) v |= v << (2 * sh);
    if (4 * sh < 64) v |= v << (4 * sh);
    const uint32_t adv = d * (8 / d);  // largest multiple of the period that fits 8 bytes
    uint32_t j = 0;
    for (; j + 8 <= ml; j += adv) lds_st64(dst + j, v);
    if (j < ml) lds_st_tail(dst + j, v, ml - j);
  }
}

}  // namespace wv
}  // namespace md
