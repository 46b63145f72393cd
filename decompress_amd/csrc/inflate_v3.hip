// inflate_v3.hip — lane-parallel batched RFC1951 inflate for gfx950 (MI355X).
//
// One independent stream per wavefront; inside the wavefront the 64 lanes decode
// 64 consecutive sub-sequences of the compressed block AT ONCE and stitch them
// together through Huffman self-synchronisation:
//
//   decode   lane i starts at bit bp + i*S (only lane 0's start is known to be a
//            token boundary), marks every token start it visits inside its own
//            S-bit zone in a per-lane LDS bitmap, and keeps decoding PAST its zone
//            until it lands on a position the owner of that zone has marked —
//            from there on both lanes would decode the same tokens, so it stops
//            ("joins" lane j).  Lane 0 is right by construction, hence so is the
//            chain 0 -> join(0) -> join(join(0)) ... : it IS the serial decode of
//            the reference loop (`inflate`, lib/de.ml:1667-1712).
//   chain    lane j on the chain emits its tokens from the join position onward
//            (token index = rank of that position in its bitmap).
//   emit     wave prefix-sum of the emitted sizes gives every token its output
//            position; literals go to an LDS staging buffer; matches whose source
//            is older than this round are read from already-flushed output in
//            L2/HBM, matches into this round resolve lane-parallel with exact
//            dependency tracking; then the round is flushed with 16-byte coalesced
//            stores and folded into the Adler-32 (WInf.update, lib/de.ml:453-455).
//
// Error behaviour keeps the oracle's order: the first failing token in stream
// order decides the status, and everything before it is written.
// Block headers, LUT construction (lib/de.ml:523-638, 1733-1793), stored blocks
// (lib/de.ml:1613-1627) and the zlib frame (lib/zl.ml:400-417) run wave-uniform
// between rounds.
#include "inflate_common.hpp"

namespace md {
namespace v3 {

#define MD_LDS __attribute__((address_space(3)))
typedef MD_LDS uint32_t lds_u32;
typedef MD_LDS uint8_t lds_u8;
typedef uint64_t u64_u __attribute__((aligned(1)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef uint16_t u16_u __attribute__((aligned(1)));

// fat LUT entry: base[15:0] | xbits[19:16] | len[23:20] | type[26:24]
enum : uint32_t { T_LIT = 0, T_LEN = 1, T_EOB = 2, T_LINK = 3, T_BAD = 4, T_DIST = 5 };
constexpr uint32_t kDistBase = 852;  // dist LUT follows the lit LUT

// per-lane stop reasons; values < 100 are MD_* status codes
constexpr uint32_t kStopEob = 100, kStopJoin = 101, kStopRoundEnd = 102, kStopBudget = 103;

constexpr uint32_t kTokMatch = 0x80000000u;  // token: bit31 match | gap[30:24] | len-3[23:16] | near[15] | dist-1[14:0]

enum { P_ENSURE = 0, P_DECODE, P_CHAIN, P_EMIT_A, P_FAR, P_NEAR, P_FLUSH, P_HEADER, P_COUNT };
enum { C_ROUNDS = 0, C_SLOTS, C_LANES, C_TOKENS, C_FAR, C_NEAR, C_NEAR_IT, C_SLOWCHAIN, C_COUNT };
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

template <int S_, int TMAX_, int KMAX_, int STAGE_, int INB_>
struct Cfg {
  static constexpr uint32_t S = S_;          // bits per lane zone
  static constexpr uint32_t BW = S_ / 32;    // bitmap words per lane
  static constexpr uint32_t TMAX = TMAX_;    // tokens per lane per round
  static constexpr uint32_t KMAX = KMAX_;    // decode slots per round
  static constexpr uint32_t STAGE = STAGE_;  // staging bytes
  static constexpr uint32_t BMAX = STAGE_ - 288;  // a lane stops once it has produced this many bytes
  static constexpr uint32_t IN_BYTES = INB_;      // compressed-input ring
  static constexpr uint32_t IN_WORDS = INB_ / 4;
  static constexpr uint32_t CHUNK = INB_ / 4;     // refill granularity
  static constexpr uint32_t CHUNK_LANE = CHUNK / 64;  // bytes per lane per refill (8 or 16)
  static constexpr uint32_t NEED = 8 * S_ + 32 > 640 ? 8 * S_ + 32 : 640;  // bytes a round / header may touch
  static_assert(S_ % 32 == 0, "zone must be whole bitmap words");
  static_assert(NEED + CHUNK <= IN_BYTES, "input ring too small");
  static_assert(CHUNK_LANE == 8 || CHUNK_LANE == 16, "refill is 8 or 16 bytes per lane");
};

template <class C>
struct Smem {
  uint32_t inring[C::IN_WORDS + 4];  // +1 mirror word (ring[IN_WORDS] == ring[0]), padded
  uint32_t lut[852 + 592];           // fat lit/len + distance LUTs
  uint32_t bitmap[kWave * C::BW];    // token starts seen by each lane inside its own zone
  union U {
    Scratch sc;  // LUT construction scratch: only live while a block header is parsed
    struct R {
      uint32_t tok[C::TMAX * kWave];
      alignas(16) uint8_t stage[C::STAGE + 16];  // +16: 8-byte copies may read a little past the data
      uint8_t owner[C::STAGE / 32 + 8];          // producer lane of each 32-byte staging block
    } r;
  } u;
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
// L1-bypassing (nt) reads of already-flushed output: the same 128-B line may have
// been cached by this CU before a later flush completed it
__device__ __forceinline__ uint64_t hbm_ld64(const uint8_t *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u64_u *>(p));
}
__device__ __forceinline__ uint32_t hbm_ld8(const uint8_t *p) { return __builtin_nontemporal_load(p); }
// up to 8 bytes at g[x .. x+n) without touching g[cap ..)
__device__ __forceinline__ uint64_t hbm_ld_guard(const uint8_t *g, uint32_t x, uint32_t n, uint32_t cap) {
  if (x + 8 <= cap) return hbm_ld64(g + x);
  uint64_t v = 0;
  for (uint32_t j = 0; j < n && j < 8; j++) v |= (uint64_t)hbm_ld8(g + x + j) << (8 * j);
  return v;
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

// RFC1951 length / distance symbol -> (base, extra bits) (lib/de.ml:293-325; +3 / +1 folded in)
__device__ __forceinline__ void len_sym(uint32_t l, uint32_t *base3, uint32_t *xb) {
  uint32_t x = (l >= 8 && l < 28) ? (l - 4) >> 2 : 0;
  uint32_t b = l < 8 ? l : l < 28 ? (4 + (l & 3)) << x : l == 28 ? 255 : 0;
  *base3 = b + 3;
  *xb = x;
}
__device__ __forceinline__ void dist_sym(uint32_t dv, uint32_t *base1, uint32_t *xb) {
  uint32_t x = (dv >= 4 && dv < 30) ? (dv - 2) >> 1 : 0;
  uint32_t b = dv < 4 ? dv + 1 : dv < 30 ? ((2 + (dv & 1)) << x) + 1 : 0;
  *base1 = b;
  *xb = x;
}
__device__ __forceinline__ uint32_t fat_lit(uint32_t e) {
  if (e & kLink) return (T_LINK << 24) | (((e >> 10) & 15) << 16) | (e & 1023);
  uint32_t len = (e >> 9) & 15, sym = e & 511;
  if (sym < 256) return (T_LIT << 24) | (len << 20) | sym;
  if (sym == 256) return (T_EOB << 24) | (len << 20);
  uint32_t b3, xl;
  len_sym((sym - 257) & 31, &b3, &xl);
  return (T_LEN << 24) | (len << 20) | (xl << 16) | b3;
}
__device__ __forceinline__ uint32_t fat_dist(uint32_t e) {
  if (e == kBad) return T_BAD << 24;
  if (e & kLink) return (T_LINK << 24) | (((e >> 10) & 15) << 16) | (e & 1023);
  uint32_t len = (e >> 9) & 15, b1, xd;
  dist_sym(e & 31, &b1, &xd);
  return (T_DIST << 24) | (len << 20) | (xd << 16) | b1;
}

struct Lane {
  uint32_t ptok, k, nb, stop, jl;
};

// ---------------------------------------------------------------------------
// The fused speculative decode of one round.
template <class C, class PF>
__device__ __forceinline__ void decode_round(const Input<C> &in, const lds_u32 *lut, lds_u32 *bitmap,
                                             lds_u32 *tok, uint32_t lane, uint32_t bp,
                                             uint32_t total_bits, uint32_t lmask, uint32_t lroot,
                                             uint32_t dmask, uint32_t droot, Lane &out, PF &pf) {
  const uint32_t L = bp + lane * C::S;
  lds_u32 *mybits = bitmap + lane * C::BW;
#pragma unroll
  for (uint32_t w = 0; w < C::BW; w++) mybits[w] = w == 0 ? 1u : 0u;  // my start is my first boundary
  uint32_t p = L, ptok = L, k = 0, nb = 0, stop = 0, jl = 0, mlen = 0;
  uint32_t shift = 0, tbase = 0, tmask = lmask, croot = lroot, ctb = 0;
  for (uint32_t slot = 0; slot < C::KMAX; ++slot) {
    const bool run = stop == 0;
    if (!__any(run)) break;
    pf.count(C_SLOTS);
    if (run) {
      const uint32_t w = in.peek(p);
      const uint32_t e = lut[tbase + ((w >> shift) & tmask)];
      const uint32_t type = (e >> 24) & 7, len = (e >> 20) & 15, xb = (e >> 16) & 15, base = e & 0xffff;
      const uint32_t val = base + __builtin_amdgcn_ubfe(w >> len, 0, xb);
      const uint32_t pn = p + len + xb;
      if (type == T_LINK) {
        shift = croot;
        tbase = ctb + base;
        tmask = (1u << xb) - 1;
      } else {
        bool commit = false;
        uint32_t tk = 0, add = 0;
        if (type == T_BAD) stop = MD_INVALID_DISTANCE_CODE;
        else if (pn > total_bits) stop = MD_UNEXPECTED_END_OF_INPUT;
        else if (type == T_LEN) {
          mlen = val;
          p = pn;
          shift = 0;
          tbase = kDistBase;
          tmask = dmask;
          croot = droot;
          ctb = kDistBase;
        } else if (type == T_EOB) {
          ptok = pn;
          stop = kStopEob;
        } else if (type == T_DIST) {
          if (val == 0) stop = MD_INVALID_DISTANCE_CODE;
          else {
            commit = true;
            tk = kTokMatch | ((mlen - 3) << 16) | (val - 1);
            add = mlen;
          }
        } else {
          commit = true;
          tk = val;
          add = 1;
        }
        if (commit) {
          tok[k * kWave + lane] = tk;
          k++;
          nb += add;
          p = pn;
          ptok = pn;
          shift = 0;
          tbase = 0;
          tmask = lmask;
          croot = lroot;
          ctb = 0;
          const uint32_t rel = pn - L;
          if (rel < C::S) {
            __hip_atomic_fetch_or(mybits + (rel >> 5), 1u << (rel & 31), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            const uint32_t dz = rel / C::S, r2 = rel - dz * C::S, j = lane + dz;
            if (j >= kWave) stop = kStopRoundEnd;
            else if ((bitmap[j * C::BW + (r2 >> 5)] >> (r2 & 31)) & 1) {
              stop = kStopJoin;
              jl = j;
            }
          }
          if (!stop && (k == C::TMAX || nb >= C::BMAX)) stop = kStopBudget;
        }
      }
    }
  }
  if (!stop) stop = kStopBudget;  // out of slots (possibly mid-token: ptok is the last boundary)
  out.ptok = ptok;
  out.k = k;
  out.nb = nb;
  out.stop = stop;
  out.jl = jl;
}

// ---------------------------------------------------------------------------
template <class C>
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
        const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
        if (hi - lo == 16) {
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
// Place the tokens [s, e) of every chain lane.  Returns MD_OK or the status of the
// first failing token (stream order); *emitted = bytes produced; *last_acc = the
// last chain lane whose tokens were (at least partly) accepted.
template <class C, class PF>
__device__ __forceinline__ int emit_round(Smem<C> *smg, Sink<C> &sk, uint32_t lane, bool on,
                                          uint32_t s, uint32_t e, uint32_t nb_emit,
                                          uint32_t *last_acc, bool *cut, uint32_t *emitted, PF &pf) {
  lds_u32 *tok = (lds_u32 *)smg->u.r.tok;
  lds_u8 *stage = (lds_u8 *)smg->u.r.stage;
  lds_u8 *owner = (lds_u8 *)smg->u.r.owner;
  const uint32_t R0 = sk.pos, rb = sk.sbase();
  const uint32_t cap = sk.cap;

  uint32_t mynb = on ? nb_emit : 0;
  const uint32_t off = wave_excl_scan(mynb, lane);
  // staging capacity: keep the largest prefix of lanes that fits
  const uint64_t fits = __ballot(off + mynb <= C::STAGE - 16);
  const uint32_t nfit = fits == ~0ull ? 64 : (uint32_t)__builtin_ctzll(~fits);
  const bool mine = on && lane < nfit;
  if (!mine) mynb = 0;
  const uint64_t mm = __ballot(mine);
  uint32_t lastl = 63 - (uint32_t)__builtin_clzll(mm);  // lane 0 always fits
  *cut = __ballot(on) != mm;
  const uint32_t q0 = R0 + off;

  // owner table: the lane that produces the first byte of every 32-byte staging
  // block — a lower bound of the producer of any byte in that block
  if (mynb) {
    uint32_t b0 = (q0 - rb + 31) >> 5, b1 = (q0 + mynb - 1 - rb) >> 5;
    for (uint32_t bb = b0; bb <= b1; bb++) owner[bb] = (uint8_t)lane;
  }
  if (lane == 0) owner[0] = 0;

  // (a) literals into the staging buffer, matches compacted to records with the
  //     literal gap in front of them; position-dependent checks in stream order
  const uint32_t ntok = mine ? e - s : 0;
  uint32_t q = q0, nm = 0, gap = 0, fail = 0, good = 0;  // good = bytes before the failing token
  for (uint32_t t = 0; t < C::TMAX; t++) {
    if (!__any(t < ntok && !fail)) break;
    if (t < ntok && !fail) {
      uint32_t tk = tok[(s + t) * kWave + lane];
      if (!(tk & kTokMatch)) {
        if (q >= cap) {
          fail = MD_UNEXPECTED_END_OF_OUTPUT;
        } else {
          stage[q - rb] = (uint8_t)tk;
          q++;
          gap++;
        }
      } else {
        uint32_t d = (tk & 0x7fff) + 1, ml = ((tk >> 16) & 0xff) + 3;
        uint32_t lim = q < 32768u ? q : 32768u;
        if (d > lim) fail = MD_INVALID_DISTANCE;
        else if (ml > cap - q) fail = MD_UNEXPECTED_END_OF_OUTPUT;
        else {
          // bit15: 1 = near (source reaches into this round), 0 = far
          uint32_t near = (q - d + ml > R0) ? 0x8000u : 0u;
          tok[nm * kWave + lane] = tk | (gap << 24) | near;
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
    uint64_t fm = __ballot(fail != 0);
    if (fm) {
      uint32_t fl = __builtin_ctzll(fm);
      rc = (int)rdlane(fail, fl);
      total = rdlane(off, fl) + rdlane(good, fl);
      if (lane > fl) nm = 0;  // later lanes are void
      lastl = fl;
    } else {
      total = rdlane(off + mynb, lastl);  // inclusive sum at the last accepted lane
    }
  }
  pf.tick(P_EMIT_A);

  // (b) far matches: the whole source lies in already-flushed output (HBM/L2).
  //     The loads of up to 4 matches per lane are in flight together.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // earlier flush stores have landed
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
          uint32_t tk = tok[m * kWave + lane];
          uint32_t ml = ((tk >> 16) & 0xff) + 3;
          qq += (tk >> 24) & 0x7f;
          m++;
          if (!(tk & 0x8000u)) {
            uint32_t src = qq - ((tk & 0x7fff) + 1);
            v0[u] = hbm_ld_guard(sk.g, src, ml, cap);
            if (ml > 8) v1[u] = hbm_ld_guard(sk.g, src + 8, ml - 8, cap);
            dl[u] = ml;
            dq[u] = qq;
            ds[u] = src;
            qq += ml;
            pf.count(C_FAR);
            break;
          }
          qq += ml;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (dl[u]) {
          lds_u8 *dd = stage + (dq[u] - rb);
          uint32_t ml = dl[u];
          if (ml >= 8) lds_st64(dd, v0[u]);
          else lds_st_tail(dd, v0[u], ml);
          if (ml >= 16) lds_st64(dd + 8, v1[u]);
          else if (ml > 8) lds_st_tail(dd + 8, v1[u], ml - 8);
          for (uint32_t j = 16; j < ml; j += 8) {  // long far match: stream the rest (rare)
            uint64_t v = hbm_ld_guard(sk.g, ds[u] + j, ml - j, cap);
            if (ml - j >= 8) lds_st64(dd + j, v);
            else lds_st_tail(dd + j, v, ml - j);
          }
        }
      }
    }
  }
  pf.tick(P_FAR);

  // (c) near matches: the source reaches into this round's staging buffer.
  //     Exact dependency tracking: `done` is the position of this lane's first
  //     unresolved match (everything this lane produces before it is final);
  //     a match may run when the first unresolved lane at or after the producer
  //     of its source is itself, or has progressed beyond the source's end.
  {
    uint32_t m = 0, qq = q0;
    uint32_t d = 0, ml = 0, qm = 0, ja = 0;
    bool pending = false;
    auto advance = [&]() {
      pending = false;
      while (m < nm) {
        uint32_t tk = tok[m * kWave + lane];
        d = (tk & 0x7fff) + 1;
        ml = ((tk >> 16) & 0xff) + 3;
        qm = qq + ((tk >> 24) & 0x7f);
        m++;
        qq = qm + ml;
        if (tk & 0x8000u) {
          uint32_t src = qm - d;
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
      if (pending) f = ja + (uint32_t)__builtin_ctzll(pm >> ja);  // bit `lane` is set, so pm >> ja != 0
      const uint32_t df = __shfl(done, f);
      if (pending && (f >= lane || df >= qm - d + ml)) {
        const uint32_t src = qm - d;
        lds_u8 *dd = stage + (qm - rb);
        if (src >= R0) {
          copy_near(dd, stage + (src - rb), ml, d);
        } else {
          // straddles the round start: the first bytes come from HBM
          const uint32_t ng = R0 - src;
          for (uint32_t j = 0; j < ng; j++) dd[j] = (uint8_t)hbm_ld8(sk.g + src + j);
          copy_near(dd + ng, stage + (R0 - rb), ml - ng, d);
        }
        pf.count(C_NEAR);
        advance();
      }
    }
  }
  pf.tick(P_NEAR);
  *last_acc = lastl;
  *emitted = total;
  return rc;
}

// ---------------------------------------------------------------------------
// All rounds of one Huffman block.  On return *bp_io is the bit after the EOB.
template <class C, class PF>
__device__ __forceinline__ int inflate_block(Smem<C> *smg, Input<C> &in, Sink<C> &sk, uint32_t lmask,
                                          uint32_t lroot, uint32_t dmask, uint32_t droot,
                                          uint32_t lane, uint32_t total_bits, uint32_t *bp_io, PF &pf) {
  uint32_t bp = *bp_io;
  const lds_u32 *lut = (const lds_u32 *)smg->lut;
  lds_u32 *bitmap = (lds_u32 *)smg->bitmap;
  lds_u32 *tok = (lds_u32 *)smg->u.r.tok;
  for (;;) {
    in.ensure(bp >> 3);
    pf.tick(P_ENSURE);
    pf.count(C_ROUNDS);
    Lane ln;
    decode_round<C>(in, lut, bitmap, tok, lane, bp, total_bits, lmask, lroot, dmask, droot, ln, pf);
    pf.tick(P_DECODE);

    // ---- chain: 0 -> join(0) -> ...  (fast path: everybody joined the next lane)
    const uint32_t pe = __shfl_up(ln.ptok, 1);
    const uint64_t jn = __ballot(ln.stop == kStopJoin && ln.jl == lane + 1);
    uint32_t c = jn == ~0ull ? 64 : (uint32_t)__builtin_ctzll(~jn);  // lanes 0..c are on the chain
    bool on = lane <= c;
    uint32_t xstart = lane == 0 ? bp : pe;
    uint32_t cur = c;
    while (rdlane(ln.stop, cur) == kStopJoin) {  // a join that skips lanes: follow it serially
      pf.count(C_SLOWCHAIN);
      const uint32_t nxt = rdlane(ln.jl, cur), x = rdlane(ln.ptok, cur);
      const uint64_t m2 = jn >> nxt;
      const uint32_t c2 = (uint32_t)__builtin_ctzll(~m2);  // join-next run starting at nxt
      if (lane == nxt) {
        on = true;
        xstart = x;
      } else if (lane > nxt && lane <= nxt + c2) {
        on = true;
        xstart = pe;
      }
      cur = nxt + c2;
    }
    const uint32_t last = cur;
    // token range of a chain lane: from the join position (rank in its bitmap) to its end
    uint32_t s = 0, nb_emit = ln.nb;
    if (on && lane > 0) {
      const uint32_t rel = xstart - (bp + lane * C::S);
#pragma unroll
      for (uint32_t w = 0; w < C::BW; w++) {
        uint32_t word = bitmap[lane * C::BW + w];
        if (w * 32 + 32 <= rel) s += __builtin_popcount(word);
        else if (w * 32 < rel) s += __builtin_popcount(word & ((1u << (rel - w * 32)) - 1));
      }
    }
    {  // bytes of the skipped prefix
      const uint32_t sk_n = on ? s : 0;
      for (uint32_t t = 0; t < C::TMAX; t++) {
        if (!__any(t < sk_n)) break;
        if (t < sk_n) {
          uint32_t tk = tok[t * kWave + lane];
          nb_emit -= (tk & kTokMatch) ? ((tk >> 16) & 0xff) + 3 : 1;
        }
      }
    }
    pf.tick(P_CHAIN);

    uint32_t emitted, last_acc;
    bool cut;
    int rc = emit_round<C>(smg, sk, lane, on, s, ln.k, nb_emit, &last_acc, &cut, &emitted, pf);
    sk.flush(emitted);
    pf.tick(P_FLUSH);
    pf.count(C_LANES, last_acc + 1);
    pf.count(C_TOKENS, wave_sum(on && lane <= last_acc ? ln.k - s : 0));
    if (rc != MD_OK) return rc;
    bp = rdlane(ln.ptok, last_acc);
    if (last_acc == last && !cut) {
      const uint32_t lstop = rdlane(ln.stop, last);
      if (lstop == kStopEob) break;
      if (lstop < 100) return (int)lstop;  // a decode error on the chain
    }
  }
  *bp_io = bp;
  return MD_OK;
}

// Dynamic block header (lib/de.ml:1733-1793), wave-uniform over the LDS ring.
template <class C>
__device__ __noinline__ int dynamic_header(UReader<C> &ur, Scratch *s, Lut *lit, Lut *dist, uint32_t lane) {
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
  if (!build_lut(K_LENS, s->lens, hlit, s, lit, lane)) return MD_INVALID_DICTIONARY;
  if (!build_lut(K_DISTS, s->lens + hlit, hdist, s, dist, lane)) return MD_INVALID_DICTIONARY;
  return MD_OK;
}

__device__ __noinline__ void fixed_tables(Scratch *s, Lut *lit, Lut *dist, uint32_t lane) {
  for (uint32_t n = lane; n < 288; n += kWave) s->lens[n] = n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8;
  build_lut(K_LENS, s->lens, 288, s, lit, lane);
  if (lane < 32) s->dist[lane] = (uint16_t)((5u << 9) | (__brev(lane) >> 27));
  dist->t = s->dist;
  dist->mask = 31;
  dist->root = 5;
  dist->maxl = 5;
}

template <class C, bool PROF>
__global__ __launch_bounds__(kWave) void inflate_v3_kernel(
    int format, uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *__restrict__ out,
    const uint64_t *__restrict__ out_off, const uint64_t *__restrict__ out_cap,
    uint64_t *__restrict__ out_len, uint64_t *__restrict__ consumed, int32_t *__restrict__ status,
    uint32_t *__restrict__ checksum, uint64_t *__restrict__ dbg) {
  extern __shared__ __align__(16) uint8_t smem_raw[];
  Smem<C> *smg = reinterpret_cast<Smem<C> *>(smem_raw);
  Prof<PROF> pf;
  pf.init();
  const uint32_t lane = threadIdx.x;
  const uint32_t sid = blockIdx.x;
  if (sid >= n) return;

  const uint8_t *src = in + in_off[sid];
  uint64_t slen64 = in_len[sid], cap64 = out_cap[sid];
  uint32_t slen = slen64 > 0x1ffffff0ull ? 0x1ffffff0u : (uint32_t)slen64;
  uint32_t cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;

  int rc = MD_OK;
  uint32_t body_off = 0, body_len = slen;
  if (format == MD_FORMAT_ZLIB) {  // Zl.Inf.Ns.inflate, lib/zl.ml:400-417
    if (slen < 2) rc = MD_UNEXPECTED_END_OF_INPUT;
    else {
      uint32_t cmf = src[0], flg = src[1];
      if (((cmf << 8) + flg) % 31 != 0 || (cmf & 0xf) != 8) rc = MD_INVALID_HEADER;
      else if (slen < 6) rc = MD_UNEXPECTED_END_OF_INPUT;
      else {
        body_off = 2;
        body_len = slen - 6;
      }
    }
  }

  Sink<C> sk;
  sk.stage = (lds_u8 *)smg->u.r.stage;
  sk.g = out + out_off[sid];
  sk.cap = cap;
  sk.pos = 0;
  sk.lane = lane;
  sk.a = 1;
  sk.b = 0;
  sk.want_adler = (checksum != nullptr) || format == MD_FORMAT_ZLIB;

  Input<C> inp;
  inp.p = src + body_off;
  inp.nbytes = body_len;
  inp.lane = lane;
  inp.ring = (lds_u32 *)smg->inring;
  inp.reset(0);
  const uint32_t total_bits = body_len * 8;
  uint32_t bp = 0;

  if (rc == MD_OK) {
    bool last = false;
    while (!last && rc == MD_OK) {
      inp.ensure(bp >> 3);
      UReader<C> ur{&inp, bp, total_bits};
      if (ur.avail() < 3) {
        rc = MD_UNEXPECTED_END_OF_INPUT;
        break;
      }
      last = ur.peek(1);
      ur.drop(1);
      uint32_t type = ur.peek(2);
      ur.drop(2);
      bp = ur.bp;
      if (type == 0) {
        // flat, lib/de.ml:1613-1627
        uint32_t p = (bp + 7) >> 3;
        if (body_len - p < 4) {
          rc = MD_UNEXPECTED_END_OF_INPUT;
          break;
        }
        uint32_t hdr = uni(inp.peek(p * 8));
        uint32_t len = hdr & 0xffff, nlen = hdr >> 16;
        p += 4;
        if (nlen != 0xffff - len) rc = MD_INVALID_COMPLEMENT_OF_LENGTH;
        else if (len > body_len - p) rc = MD_UNEXPECTED_END_OF_INPUT;
        else if (len > sk.cap - sk.pos) rc = MD_UNEXPECTED_END_OF_OUTPUT;
        else {
          const uint8_t *q = inp.p + p;
          uint32_t left = len;
          while (left) {
            uint32_t seg = left < C::STAGE - 16 ? left : C::STAGE - 16;
            uint32_t s0 = sk.pos - sk.sbase();
            for (uint32_t j = lane; j < seg; j += kWave) sk.stage[s0 + j] = q[j];
            sk.flush(seg);
            q += seg;
            left -= seg;
          }
          p += len;
          bp = p * 8;
          inp.reset(p);
        }
      } else if (type == 3) {
        rc = MD_INVALID_KIND_OF_BLOCK;
      } else {
        Lut lit, dist;
        if (type == 1) fixed_tables(&smg->u.sc, &lit, &dist, lane);
        else {
          rc = dynamic_header<C>(ur, &smg->u.sc, &lit, &dist, lane);
          bp = ur.bp;
        }
        if (rc == MD_OK) {
          // fat LUTs (base / extra bits / type per entry) from the packed 16-bit ones
          for (uint32_t i = lane; i < 852; i += kWave) smg->lut[i] = fat_lit(smg->u.sc.lit[i]);
          for (uint32_t i = lane; i < 592; i += kWave) smg->lut[kDistBase + i] = fat_dist(smg->u.sc.dist[i]);
          pf.tick(P_HEADER);
          rc = inflate_block<C>(smg, inp, sk, uni(lit.mask), uni(lit.root), uni(dist.mask), uni(dist.root),
                                lane, total_bits, &bp, pf);
        }
      }
    }
  }
  uint32_t used = (bp + 7) >> 3;  // i_pos - (bits lsr 3), lib/de.ml:1805
  uint32_t adler = (sk.b << 16) | sk.a;
  if (rc == MD_OK && format == MD_FORMAT_ZLIB) {
    const uint8_t *t = src + 2 + used;
    uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    if (want != adler) rc = MD_INVALID_CHECKSUM;
    used += 6;
  }
  if (lane == 0) {
    out_len[sid] = sk.pos;
    consumed[sid] = rc == MD_OK ? used : 0;
    status[sid] = rc;
    if (checksum) checksum[sid] = adler;
  }
  if constexpr (PROF) {
    if (lane == 0 && sid == 0 && dbg) {
      for (int i = 0; i < P_COUNT; i++) dbg[i] = pf.acc[i];
      for (int i = 0; i < C_COUNT; i++) dbg[P_COUNT + i] = pf.cnt[i];
    }
  }
}

}  // namespace v3
}  // namespace md

extern "C" int md_launch_inflate_v2(int variant, int format, uint32_t n, const uint8_t *in,
                                    const uint64_t *in_off, const uint64_t *in_len, uint8_t *out,
                                    const uint64_t *out_off, const uint64_t *out_cap,
                                    uint64_t *out_len, uint64_t *consumed, int32_t *status,
                                    uint32_t *checksum, uint64_t *dbg, hipStream_t stream) {
  if (n == 0) return 0;
  dim3 grid(n), block(md::kWave);
#define MD_LAUNCH_V3(CFG)                                                                        \
  do {                                                                                           \
    if (dbg)                                                                                     \
      hipLaunchKernelGGL((md::v3::inflate_v3_kernel<CFG, true>), grid, block,                    \
                         sizeof(md::v3::Smem<CFG>), stream, format, n, in, in_off, in_len, out,  \
                         out_off, out_cap, out_len, consumed, status, checksum, dbg);            \
    else                                                                                         \
      hipLaunchKernelGGL((md::v3::inflate_v3_kernel<CFG, false>), grid, block,                   \
                         sizeof(md::v3::Smem<CFG>), stream, format, n, in, in_off, in_len, out,  \
                         out_off, out_cap, out_len, consumed, status, checksum, dbg);            \
  } while (0)
  //                       S   TMAX KMAX STAGE  IN_BYTES
  using A = md::v3::Cfg<128, 24, 48, 6144, 2048>;
  using B = md::v3::Cfg<96, 20, 40, 5120, 2048>;
  using Cc = md::v3::Cfg<160, 28, 56, 7168, 4096>;
  switch (variant) {
  case 0: MD_LAUNCH_V3(A); break;
  case 1: MD_LAUNCH_V3(B); break;
  case 2: MD_LAUNCH_V3(Cc); break;
  default: return -1;
  }
#undef MD_LAUNCH_V3
  return (int)hipGetLastError();
}
