// inflate_lane.hpp — the lane-parallel inflate machinery shared by the fused kernel
// (inflate_v4.hip) and the two-kernel split (inflate_split.hip): LDS input ring, fat LUT,
// speculative decode pass, staging sink, round emission, block headers.
#pragma once
#include "inflate_common.hpp"

namespace md {
namespace v4 {

#define MD_LDS __attribute__((address_space(3)))
typedef MD_LDS uint32_t lds_u32;
typedef MD_LDS uint16_t lds_u16;
typedef MD_LDS uint8_t lds_u8;
typedef uint64_t u64_u __attribute__((aligned(1)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef uint16_t u16_u __attribute__((aligned(1)));

constexpr uint32_t kDistBase = 852;  // Scratch::dist follows Scratch::lit
constexpr uint32_t kStopEob = 100;   // per-lane stop reasons; values < 100 are MD_* status codes
constexpr int kStatusLogFull = 50;   // internal: the split path ran out of token log (stream is redone fused)

// match record: gap[31:24] | len-3[23:16] | near[15] | dist-1[14:0]
constexpr uint32_t kNear = 0x8000u;

enum { P_ENSURE = 0, P_DECODE1, P_DECODE2, P_EMIT_A, P_FAR, P_NEAR, P_ADLER, P_HEADER, P_COUNT };
enum { C_ROUNDS = 0, C_PASSES, C_LANES, C_TOKENS, C_SLOTS, C_NEAR_IT, C_COUNT };
template <bool ON>
struct Prof {
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
  __device__ __forceinline__ void count(int i, uint32_t n = 1) { cnt[i] += n; }
};
template <>
struct Prof<false> {
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ void tick(int) {}
  __device__ __forceinline__ void count(int, uint32_t = 1) {}
};

template <int S_, int LMAX_, int MMAX_, int KMAX_, int INB_, int PASSES_, int STAGE_>
struct Cfg {
  static constexpr uint32_t S = S_;        // bits per lane zone
  static constexpr uint32_t LMAX = LMAX_;  // literals per lane per round
  static constexpr uint32_t MMAX = MMAX_;  // matches per lane per round
  static constexpr uint32_t KMAX = KMAX_;  // decode slots per pass
  static constexpr uint32_t PASSES = PASSES_;  // A2 passes after A1
  static constexpr uint32_t STAGE = STAGE_;    // staging bytes (one round of output)
  static constexpr uint32_t BMAX = STAGE_ - 288;  // a lane stops once it has produced this many bytes
  static constexpr uint32_t IN_BYTES = INB_;   // compressed-input ring
  static constexpr uint32_t IN_WORDS = INB_ / 4;
  static constexpr uint32_t CHUNK = INB_ >= 4096 ? 1024 : 512;  // refill granularity
  static constexpr uint32_t CHUNK_LANE = CHUNK / 64;
  static constexpr uint32_t NEED = 8 * S_ + 32 > 640 ? 8 * S_ + 32 : 640;  // bytes a round / header may touch
  static_assert(LMAX_ + MMAX_ <= 64, "token type mask is one 64-bit register");
  static_assert(NEED + CHUNK <= IN_BYTES, "input ring too small");
  static_assert(CHUNK_LANE == 8 || CHUNK_LANE == 16, "refill is 8 or 16 bytes per lane");
};

// fat LUT entry: base[15:0] | xbits[19:16] | len[23:20] | type[26:24]
//   base = literal byte / length base (+3) / distance base (+1) / sub-table offset (LINK)
enum : uint32_t { T_LIT = 0, T_LEN = 1, T_EOB = 2, T_LINK = 3, T_BAD = 4, T_DIST = 5 };

template <class C>
struct Smem {
  uint32_t inring[C::IN_WORDS + 4];  // +1 mirror word (ring[IN_WORDS] == ring[0]), padded
  uint32_t lut[852 + 592];           // fat lit/len LUT, then fat distance LUT
  union U {
    Scratch sc;  // packed 16-bit LUTs + construction scratch: live only while a header is parsed
    struct T {
      uint32_t mrec[C::MMAX * kWave];  // match tokens, [m][lane]
      uint8_t lits[C::LMAX * kWave];   // literal tokens, [i][lane]
      alignas(16) uint8_t stage[C::STAGE + 16];  // +16: 8-byte copies may read a little past the data
      uint8_t owner[C::STAGE / 32 + 8];          // producer lane of each 32-byte staging block
    } t;
  } u;
};

// RFC1951 length / distance symbol -> (base, extra bits) (lib/de.ml:293-325; +3 / +1 folded in)
__device__ __forceinline__ uint32_t fat_lit(uint32_t e) {
  if (e & kLink) return (T_LINK << 24) | (((e >> 10) & 15) << 16) | (e & 1023);
  const uint32_t len = (e >> 9) & 15, sym = e & 511;
  if (sym < 256) return (T_LIT << 24) | (len << 20) | sym;
  if (sym == 256) return (T_EOB << 24) | (len << 20);
  const uint32_t l = (sym - 257) & 31;
  const uint32_t xb = (l >= 8 && l < 28) ? (l - 4) >> 2 : 0;
  const uint32_t base = (l < 8 ? l : l < 28 ? (4 + (l & 3)) << xb : l == 28 ? 255 : 0) + 3;
  return (T_LEN << 24) | (len << 20) | (xb << 16) | base;
}
__device__ __forceinline__ uint32_t fat_dist(uint32_t e) {
  if (e == kBad) return T_BAD << 24;
  if (e & kLink) return (T_LINK << 24) | (((e >> 10) & 15) << 16) | (e & 1023);
  const uint32_t len = (e >> 9) & 15, dv = e & 31;
  const uint32_t xb = (dv >= 4 && dv < 30) ? (dv - 2) >> 1 : 0;
  const uint32_t base = dv < 4 ? dv + 1 : dv < 30 ? ((2 + (dv & 1)) << xb) + 1 : 0;
  return (T_DIST << 24) | (len << 20) | (xb << 16) | base;
}

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

// ---------------------------------------------------------------------------
// Compressed input: an LDS ring addressed by absolute bit position.
template <class C>
struct Input {
  const uint8_t *p;
  uint32_t nbytes;
  uint32_t lane;
  lds_u32 *ring;
  uint32_t in_hi;  // stream bytes [.., in_hi) are in the ring (multiple of CHUNK)

  __device__ __forceinline__ void load_chunk() {
    constexpr uint32_t N = C::CHUNK_LANE;
    uint32_t off = in_hi + lane * N;
    uint32_t w[4] = {0, 0, 0, 0};
    if (off + N <= nbytes) {
      __builtin_memcpy(w, p + off, N);
    } else if (off < nbytes) {
      for (uint32_t k = 0; k < N && off + k < nbytes; k++) w[k >> 2] |= (uint32_t)p[off + k] << (8 * (k & 3));
    }
    uint32_t r = (off & (C::IN_BYTES - 1)) >> 2;
    ring[r] = w[0];
    ring[r + 1] = w[1];
    if (N == 16) {
      ring[r + 2] = w[2];
      ring[r + 3] = w[3];
    }
    if (r == 0) ring[C::IN_WORDS] = w[0];  // mirror of word 0 for the wrap-around peek
    in_hi += C::CHUNK;
  }
  __device__ __forceinline__ void reset(uint32_t byte_pos) { in_hi = byte_pos & ~(C::CHUNK - 1); }
  __device__ __forceinline__ void ensure(uint32_t byte_pos) {
    while (in_hi < byte_pos + C::NEED) load_chunk();
  }
  // 32 bits of the stream starting at absolute bit position bp (zero beyond the end)
  __device__ __forceinline__ uint32_t peek(uint32_t bp) const {
    uint32_t w = (bp >> 5) & (C::IN_WORDS - 1);
    uint32_t lo = ring[w], hi = ring[w + 1];
    return __builtin_amdgcn_alignbit(hi, lo, bp & 31);
  }
};

// wave-uniform bit cursor over the ring (block headers)
template <class C>
struct UReader {
  const Input<C> *in;
  uint32_t bp;
  uint32_t total;  // total real bits of the stream
  __device__ __forceinline__ int64_t avail() const { return (int64_t)total - (int64_t)bp; }
  __device__ __forceinline__ uint32_t peek(uint32_t n) const { return uni(in->peek(bp)) & ((1u << n) - 1); }
  __device__ __forceinline__ void drop(uint32_t n) { bp += n; }
};

struct LaneState {
  uint32_t start, end, nlit, nmat, nb, stop;
  uint64_t tmask;  // bit t set = token t is a match
};

// One speculative decode pass of this lane's zone [start, limit).  The slot body
// is straight-line: one LUT entry per slot (a 2nd-level entry or the distance
// code of a match take another slot), all state updates are selects.
template <class C, class PF, class MP, class LP>
__device__ __forceinline__ void decode_pass(const Input<C> &in, const lds_u32 *lut, MP mrec,
                                            LP lits, uint32_t lane, uint32_t total_bits,
                                            uint32_t lmask, uint32_t lroot, uint32_t dmask,
                                            uint32_t droot, bool go, uint32_t limit, LaneState &ls,
                                            PF &pf) {
  uint32_t p = ls.start, ptok = ls.start, nlit = 0, nmat = 0, nb = 0, stop = 0;
  uint64_t tmask = 0;
  // table cursor: (shift, tbase, tmsk) index the next entry; (croot, ctb) describe the table in use
  uint32_t shift = 0, tbase = 0, tmsk = lmask, croot = lroot, ctb = 0, mlen = 0;
  bool tokstart = true, run = go;
  for (uint32_t slot = 0; slot < C::KMAX; ++slot) {
    run = run && !(tokstart && (p >= limit || nlit == C::LMAX || nmat == C::MMAX || nb >= C::BMAX));
    if (!__any(run)) break;
    pf.count(C_SLOTS);
    if (run) {
      const uint32_t w = in.peek(p);
      const uint32_t e = lut[tbase + ((w >> shift) & tmsk)];
      const uint32_t type = (e >> 24) & 7, len = (e >> 20) & 15, xb = (e >> 16) & 15, base = e & 0xffff;
      const uint32_t val = base + __builtin_amdgcn_ubfe(w >> len, 0, xb);
      const uint32_t pn = p + len + xb;
      const bool is_link = type == T_LINK;
      // oracle order: empty distance slot (D2), then end of input, then distance code 30/31
      const uint32_t err = type == T_BAD                 ? (uint32_t)MD_INVALID_DISTANCE_CODE
                           : (!is_link && pn > total_bits) ? (uint32_t)MD_UNEXPECTED_END_OF_INPUT
                           : (type == T_DIST && val == 0)  ? (uint32_t)MD_INVALID_DISTANCE_CODE
                                                           : 0u;
      const bool ok = !err && !is_link;
      const bool c_lit = ok && type == T_LIT, c_mat = ok && type == T_DIST;
      const bool is_len = ok && type == T_LEN, is_eob = ok && type == T_EOB;
      if (c_lit) lits[nlit * kWave + lane] = (uint8_t)val;
      if (c_mat) mrec[nmat * kWave + lane] = ((mlen - 3) << 16) | (val - 1);
      tmask |= (uint64_t)c_mat << (nlit + nmat);
      nb += c_lit ? 1u : c_mat ? mlen : 0u;
      nlit += c_lit;
      nmat += c_mat;
      ptok = (c_lit || c_mat || is_eob) ? pn : ptok;
      p = ok ? pn : p;
      mlen = is_len ? val : mlen;
      stop = err ? err : is_eob ? kStopEob : 0u;
      shift = is_link ? croot : 0u;
      tbase = is_link ? ctb + base : is_len ? kDistBase : 0u;
      tmsk = is_link ? (1u << xb) - 1 : is_len ? dmask : lmask;
      croot = is_link ? croot : is_len ? droot : lroot;
      ctb = is_link ? ctb : is_len ? kDistBase : 0u;
      tokstart = c_lit || c_mat;
      run = stop == 0;
    }
  }
  if (go) {
    ls.end = ptok;
    ls.nlit = nlit;
    ls.nmat = nmat;
    ls.nb = nb;
    ls.stop = stop;
    ls.tmask = tmask;
  }
}

// ---------------------------------------------------------------------------
struct Sink {
  lds_u8 *stage;
  uint8_t *g;
  uint32_t cap;
  uint32_t pos;  // bytes produced and flushed
  uint32_t lane;
  uint32_t a, b;
  bool want_adler;

  __device__ __forceinline__ void adler_fold(uint32_t s1, uint32_t s2, uint32_t n) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    b = (b + n * a + s2) % 65521u;
    a = (a + s1) % 65521u;
  }
  // staging index of output position x is x - (pos & ~15)
  __device__ __forceinline__ uint32_t sbase() const { return pos & ~15u; }

  // write stage[...] for positions [pos, pos+total) to HBM, fold Adler-32, advance pos
  __device__ __forceinline__ void flush(uint32_t total) {
    const uint32_t rb = sbase();
    const uint32_t endp = pos + total;
    for (uint32_t ps = rb; ps < endp; ps += 1024) {
      const uint32_t a0 = ps > pos ? ps : pos;
      const uint32_t b0 = ps + 1024 < endp ? ps + 1024 : endp;
      const uint32_t cpos = ps + lane * 16;
      const uint32_t lo = cpos > a0 ? cpos : a0;
      const uint32_t hi = cpos + 16 < b0 ? cpos + 16 : b0;
      uint32_t s1 = 0, s2 = 0;
      if (lo < hi) {
        const lds_u32 *sp = reinterpret_cast<const lds_u32 *>(stage + (cpos - rb));  // 16-byte aligned
        const uint32_t w[4] = {sp[0], sp[1], sp[2], sp[3]};
        if (hi - lo == 16) {
          const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
          __builtin_memcpy(g + cpos, &v, 16);
          if (want_adler) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
              uint32_t d = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
              s1 += d;
              s2 += (b0 - (cpos + k)) * d;
            }
          }
        } else {
          for (uint32_t x = lo; x < hi; x++) {
            uint32_t k = x - cpos;
            uint32_t d = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
            g[x] = (uint8_t)d;
            s1 += d;
            s2 += (b0 - x) * d;
          }
        }
      }
      if (want_adler) adler_fold(s1, s2, b0 - a0);
    }
    pos = endp;
  }
};

// ---------------------------------------------------------------------------
// Phase B of a round: place the accepted tokens into the staging buffer.  Returns
// MD_OK or the status of the first failing token (stream order); *emitted = bytes
// produced; *nvalid_out = lanes actually accepted.
template <class C, class PF>
__device__ __forceinline__ int emit_round(lds_u32 *mrec, const lds_u8 *lits, lds_u8 *owner, Sink &sk,
                                          uint32_t lane, uint32_t lo, uint32_t nvalid, const LaneState &ls,
                                          uint32_t *nvalid_out, uint32_t *emitted, PF &pf) {
  lds_u8 *stage = sk.stage;
  const uint32_t R0 = sk.pos, rb = sk.sbase(), cap = sk.cap;
  const uint8_t *g = sk.g;

  // lanes [lo, nvalid) are placed (lo > 0 only when an earlier call accepted a prefix)
  uint32_t mynb = (lane >= lo && lane < nvalid) ? ls.nb : 0;
  const uint32_t off = wave_excl_scan(mynb, lane);
  {  // staging capacity: keep the largest prefix of lanes that fits
    const uint64_t fits = __ballot(off + mynb <= C::STAGE - 16);
    const uint32_t nfit = fits == ~0ull ? 64 : (uint32_t)__builtin_ctzll(~fits);
    if (nfit < nvalid) nvalid = nfit;  // lane `lo` always fits alone (BMAX)
  }
  const bool mine = lane >= lo && lane < nvalid;
  if (!mine) mynb = 0;
  const uint32_t q0 = R0 + off;
  const uint32_t ntok = mine ? ls.nlit + ls.nmat : 0;

  // owner table: the lane that produces the first byte of every 32-byte staging
  // block — a lower bound of the producer of any byte in that block
  if (mynb) {
    const uint32_t b0 = (q0 - rb + 31) >> 5, b1 = (q0 + mynb - 1 - rb) >> 5;
    for (uint32_t bb = b0; bb <= b1; bb++) owner[bb] = (uint8_t)lane;
  }
  if (lane == 0) owner[0] = 0;

  // (a) literals into the staging buffer; matches become records (gap of literals
  //     in front, near/far flag); position-dependent checks in stream order
  uint32_t q = q0, li = 0, nm = 0, gap = 0, fail = 0, good = 0;  // good = bytes before the failing token
  for (uint32_t t = 0; t < C::LMAX + C::MMAX; t++) {
    if (!__any(t < ntok && !fail)) break;
    if (t < ntok && !fail) {
      if (!((ls.tmask >> t) & 1)) {
        if (q >= cap) fail = MD_UNEXPECTED_END_OF_OUTPUT;
        else {
          stage[q - rb] = lits[li * kWave + lane];
          li++;
          q++;
          gap++;
        }
      } else {
        const uint32_t tk = mrec[nm * kWave + lane];
        const uint32_t d = (tk & 0x7fff) + 1, ml = ((tk >> 16) & 0xff) + 3;
        const uint32_t lim = q < 32768u ? q : 32768u;
        if (d > lim) fail = MD_INVALID_DISTANCE;
        else if (ml > cap - q) fail = MD_UNEXPECTED_END_OF_OUTPUT;
        else {
          mrec[nm * kWave + lane] = tk | (gap << 24) | ((q - d + ml > R0) ? kNear : 0u);
          nm++;
          gap = 0;
          q += ml;
        }
      }
      if (!fail) good = q - q0;
    }
  }
  // first failing lane (stream order) truncates the round
  int rc = MD_OK;
  uint32_t total;
  {
    const uint64_t fm = __ballot(fail != 0);
    if (fm) {
      const uint32_t fl = __builtin_ctzll(fm);
      rc = (int)rdlane(fail, fl);
      total = rdlane(off, fl) + rdlane(good, fl);
      if (lane > fl) nm = 0;  // later lanes are void
      nvalid = fl + 1;
    } else {
      total = rdlane(off + mynb, nvalid - 1);  // inclusive sum at the last accepted lane
    }
  }
  pf.tick(P_EMIT_A);

  // (b) far matches: the whole source is older than this round — final in HBM/L2,
  //     visible once the flush of earlier rounds has been waited for; d >= ml.
  //     The loads of up to 4 matches per lane are in flight together.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  {
    uint32_t qq = q0, m = 0;
    while (__any(m < nm)) {
      uint64_t v0[4], v1[4];
      uint32_t dq[4], dl[4], ds[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        dl[u] = 0;
        dq[u] = 0;
        ds[u] = 0;
        v0[u] = 0;
        v1[u] = 0;
        while (m < nm) {  // advance to this lane's next far record
          const uint32_t tk = mrec[m * kWave + lane];
          const uint32_t d = (tk & 0x7fff) + 1, ml = ((tk >> 16) & 0xff) + 3;
          qq += tk >> 24;
          const uint32_t cq = qq;
          qq += ml;
          m++;
          if (tk & kNear) continue;
          v0[u] = out_ld_guard(g, cq - d, ml, cap);
          if (ml > 8) v1[u] = out_ld_guard(g, cq - d + 8, ml - 8, cap);
          dq[u] = cq;
          dl[u] = ml;
          ds[u] = cq - d;
          break;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (dl[u]) {
          lds_u8 *dd = stage + (dq[u] - rb);
          const uint32_t ml = dl[u];
          lds_st(dd, v0[u], ml);
          if (ml > 8) lds_st(dd + 8, v1[u], ml - 8);
          for (uint32_t j = 16; j < ml; j += 8)  // long far match: stream the rest (rare)
            lds_st(dd + j, out_ld_guard(g, ds[u] + j, ml - j, cap), ml - j);
        }
      }
    }
  }
  pf.tick(P_FAR);

  // (c) near matches: the source reaches into this round's staging buffer.  `done`
  //     is the position of this lane's first unresolved match: everything the lane
  //     produces before it is final.  A match may run when the first unresolved
  //     lane at or after the producer of its source is itself, or has progressed
  //     beyond the source's end.
  {
    uint32_t m = 0, qq = q0;
    uint32_t d = 0, ml = 0, qm = 0, ja = 0;
    bool pending = false;
    auto advance = [&]() {
      pending = false;
      while (m < nm) {
        const uint32_t tk = mrec[m * kWave + lane];
        d = (tk & 0x7fff) + 1;
        ml = ((tk >> 16) & 0xff) + 3;
        qm = qq + (tk >> 24);
        m++;
        qq = qm + ml;
        if (tk & kNear) {
          const uint32_t src = qm - d;
          ja = src >= rb ? owner[(src - rb) >> 5] : 0;
          pending = true;
          break;
        }
      }
    };
    advance();
    for (;;) {
      const uint64_t pm = __ballot(pending);
      if (!pm) break;
      pf.count(C_NEAR_IT);
      const uint32_t done = pending ? qm : 0xffffffffu;
      uint32_t f = lane;
      if (pending) f = ja + (uint32_t)__builtin_ctzll(pm >> ja);  // bit `lane` is set: pm >> ja != 0
      const uint32_t df = __shfl(done, f);
      if (pending && (f >= lane || df >= qm - d + ml)) {
        const uint32_t src = qm - d;
        lds_u8 *dd = stage + (qm - rb);
        if (src >= R0) {
          copy_near(dd, stage + (src - rb), ml, d);
        } else {
          // straddles the round start: the first bytes come from HBM
          const uint32_t ng = R0 - src;
          for (uint32_t j = 0; j < ng; j++) dd[j] = (uint8_t)out_ld8(g + src + j);
          copy_near(dd + ng, stage + (R0 - rb), ml - ng, d);
        }
        advance();
      }
    }
  }
  pf.tick(P_NEAR);
  *nvalid_out = nvalid;
  *emitted = total;
  return rc;
}

// ---------------------------------------------------------------------------
// All rounds of one Huffman block.  On return *bp_io is the bit after the EOB.
template <class C, class PF>
__device__ __forceinline__ int inflate_block(Smem<C> *smg, Input<C> &in, Sink &sk, uint32_t lmask,
                                             uint32_t lroot, uint32_t dmask, uint32_t droot,
                                             uint32_t lane, uint32_t total_bits, uint32_t *bp_io,
                                             PF &pf) {
  uint32_t bp = *bp_io;
  const lds_u32 *lut = (const lds_u32 *)smg->lut;
  lds_u32 *mrec = (lds_u32 *)smg->u.t.mrec;
  lds_u8 *lits = (lds_u8 *)smg->u.t.lits;
  for (;;) {
    in.ensure(bp >> 3);
    pf.tick(P_ENSURE);
    pf.count(C_ROUNDS);
    pf.count(C_PASSES);
    LaneState ls;
    ls.start = bp + lane * C::S;
    const uint32_t limit = bp + (lane + 1) * C::S;
    decode_pass<C>(in, lut, mrec, lits, lane, total_bits, lmask, lroot, dmask, droot, true, limit, ls, pf);
    pf.tick(P_DECODE1);
    // A2: chain the lanes
    for (uint32_t it = 0; it < C::PASSES; it++) {
      const uint32_t pe = __shfl_up(ls.end, 1), ps = __shfl_up(ls.stop, 1);
      const bool redo = lane > 0 && ps == 0 && pe != ls.start;
      if (!__any(redo)) break;
      if (redo) ls.start = pe;
      pf.count(C_PASSES);
      decode_pass<C>(in, lut, mrec, lits, lane, total_bits, lmask, lroot, dmask, droot, redo, limit, ls, pf);
    }
    pf.tick(P_DECODE2);
    uint32_t nvalid;
    {
      const uint32_t pe = __shfl_up(ls.end, 1), ps = __shfl_up(ls.stop, 1);
      const uint64_t bad = __ballot(lane > 0 && (ps != 0 || pe != ls.start));
      nvalid = bad ? (uint32_t)__builtin_ctzll(bad) : 64;
    }
    uint32_t emitted;
    int rc = emit_round<C>(mrec, lits, (lds_u8 *)smg->u.t.owner, sk, lane, 0, nvalid, ls, &nvalid, &emitted, pf);
    sk.flush(emitted);
    pf.tick(P_ADLER);
    pf.count(C_LANES, nvalid);
    pf.count(C_TOKENS, wave_sum(lane < nvalid ? ls.nlit + ls.nmat : 0));
    if (rc != MD_OK) return rc;
    const uint32_t lastl = nvalid - 1;
    const uint32_t lstop = rdlane(ls.stop, lastl);
    bp = rdlane(ls.end, lastl);
    if (lstop == kStopEob) break;
    if (lstop != 0) return (int)lstop;
  }
  *bp_io = bp;
  return MD_OK;
}

// Dynamic block header (lib/de.ml:1733-1793), wave-uniform over the LDS ring.
template <class C>
__device__ __noinline__ int dynamic_header(UReader<C> &ur, Scratch *s, Lut *lit, Lut *dist, uint32_t lane,
                                           uint16_t *lit_tbl = nullptr, uint16_t *dist_tbl = nullptr) {
  if (ur.avail() < 14) return MD_UNEXPECTED_END_OF_INPUT;
  uint32_t hlit = ur.peek(5) + 257;
  ur.drop(5);
  uint32_t hdist = ur.peek(5) + 1;
  ur.drop(5);
  uint32_t hclen = ur.peek(4) + 4;
  ur.drop(4);
  if (lane < 19) s->lens[lane] = 0;
  for (uint32_t i = 0; i < hclen; i++) {
    if (ur.avail() < 3) return MD_UNEXPECTED_END_OF_INPUT;
    uint32_t v = ur.peek(3);
    ur.drop(3);
    if (lane == 0) s->lens[c_zigzag[i]] = (uint8_t)v;
  }
  Lut cl;
  if (!build_lut(K_CODES, s->lens, 19, s, &cl, lane)) return MD_INVALID_DICTIONARY;
  const uint32_t max_res = hlit + hdist;
  uint32_t i = 0, prev = 0;
  while (i < max_res) {
    if (ur.avail() < (int64_t)cl.maxl) return MD_UNEXPECTED_END_OF_INPUT;
    uint32_t e = uni(cl.t[ur.peek(cl.maxl)]);
    if (e == kBad) return MD_INVALID_DICTIONARY;
    uint32_t sym = e & 511, len = (e >> 9) & 15;
    ur.drop(len);
    if (sym < 16) {
      if (lane == 0) s->lens[i] = (uint8_t)sym;
      prev = sym;
      i++;
    } else {
      uint32_t nb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
      if (sym == 16 && i == 0) return MD_INVALID_DICTIONARY;
      if (ur.avail() < (int64_t)nb) return MD_UNEXPECTED_END_OF_INPUT;
      uint32_t copy = ur.peek(nb) + (sym == 18 ? 11 : 3);
      ur.drop(nb);
      uint32_t val = sym == 16 ? prev : 0;
      if (i + copy > max_res) return MD_INVALID_DICTIONARY;
      for (uint32_t x = lane; x < copy; x += kWave) s->lens[i + x] = (uint8_t)val;
      prev = val;
      i += copy;
    }
  }
  if (uni(s->lens[256]) == 0) return MD_INVALID_DICTIONARY;
  if (!build_lut(K_LENS, s->lens, hlit, s, lit, lane, lit_tbl)) return MD_INVALID_DICTIONARY;
  if (!build_lut(K_DISTS, s->lens + hlit, hdist, s, dist, lane, dist_tbl)) return MD_INVALID_DICTIONARY;
  return MD_OK;
}

__device__ __noinline__ void fixed_tables(Scratch *s, Lut *lit, Lut *dist, uint32_t lane,
                                          uint16_t *lit_tbl = nullptr, uint16_t *dist_tbl = nullptr) {
  for (uint32_t n = lane; n < 288; n += kWave) s->lens[n] = n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8;
  build_lut(K_LENS, s->lens, 288, s, lit, lane, lit_tbl);
  uint16_t *dt = dist_tbl ? dist_tbl : s->dist;
  if (lane < 32) dt[lane] = (uint16_t)((5u << 9) | (__brev(lane) >> 27));
  dist->t = dt;
  dist->mask = 31;
  dist->root = 5;
  dist->maxl = 5;
}


}  // namespace v4
}  // namespace md
