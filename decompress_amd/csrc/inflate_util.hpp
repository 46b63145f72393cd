// inflate_util.hpp — small device helpers of the wave inflate kernel (gfx950): LDS / output-buffer
// copies with LZ77 forward-byte semantics, wave scans, the optional in-kernel phase profile.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
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
// unaligned AND aliasing: the staging buffer is written as bytes / halfwords / words and read back as 8-byte words;
// without may_alias the compiler may reorder such a load before the stores it depends on (TBAA)
typedef uint64_t u64_u __attribute__((aligned(1), may_alias));
typedef uint32_t u32_u __attribute__((aligned(1), may_alias));
typedef uint16_t u16_u __attribute__((aligned(1), may_alias));

enum { P_ENSURE = 0, P_DECODE1, P_DECODE2, P_EMIT_A, P_FAR, P_NEAR, P_ADLER, P_HEADER, P_NEAR_FAST, P_NEAR_SLOW, P_NEAR_UPD, P_NEAR_LOAD, P_HDR_LENS, P_HDR_LIT, P_FAR_REC, P_FAR_LOAD, P_WAIT_DEC, P_WAIT_COPY, P_COUNT };
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
// Sums over the wavefront on the DPP network (row shifts, then gfx9's row broadcasts): no LDS round trips - as six
// ds_bpermute steps each of them was ~700 cycles of nothing else on its wavefront's chain.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  uint32_t v = x;
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {  // (a lane read: the compiler knows it is the same in all lanes)
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t) { return wave_incl_scan(v) - v; }
// the value of the lane below (lane 0: 0)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
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
__device__ __forceinline__ uint64_t lds_ld64(const lds_u8 *p) { return *reinterpret_cast<const MD_LDS u64_u *>(p); }

// exact store of the low n (1..8) bytes of v at an arbitrary LDS address: two overlapping 4-byte (or 1 + 2-byte) stores
__device__ __forceinline__ void lds_put(lds_u8 *p, uint64_t v, uint32_t n) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  if (n >= 4) {
    const uint32_t k = n - 4;  // 0..4
    const uint32_t t = k >= 4 ? hi : __builtin_amdgcn_alignbyte(hi, lo, k);
    *reinterpret_cast<MD_LDS u32_u *>(p) = lo;
    *reinterpret_cast<MD_LDS u32_u *>(p + k) = t;
  } else {
    *p = (uint8_t)lo;
    if (n > 1) *reinterpret_cast<MD_LDS u16_u *>(p + n - 2) = (uint16_t)(lo >> (8 * (n - 2)));
  }
}

}  // namespace wv
}  // namespace md
