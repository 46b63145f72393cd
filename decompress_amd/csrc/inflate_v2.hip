// inflate_v2.hip — lane-parallel batched RFC1951 inflate for gfx950 (MI355X).
//
// Still one independent stream per wavefront, but the 64 lanes now decode 64
// consecutive sub-sequences of the compressed block at once (speculative,
// self-synchronising Huffman decode) instead of idling behind lane 0:
//
//   round:  A1  lane i decodes tokens from bit  bp + i*S  up to  bp + (i+1)*S
//               (only lane 0's start is known to be a token boundary)
//           A2  every lane compares its start with its left neighbour's end and
//               re-decodes from there until the chain start_i == end_{i-1} holds
//               (Huffman streams re-synchronise within a few symbols, so this
//               is 1-2 extra passes); the consistent prefix of lanes is accepted
//           B   wave prefix-sum of the lanes' output sizes gives every token its
//               output position; literals go to an LDS staging buffer, matches
//               are resolved lane-parallel (far sources from already-flushed
//               output in L2/HBM, near sources inside the staging buffer by
//               multi-round resolution), then the round is flushed to HBM with
//               16-byte coalesced stores and folded into the Adler-32.
//
// The decode is bit-exact with the serial reference loop (`inflate`,
// lib/de.ml:1667-1712): a consistent chain of token boundaries starting at a
// known boundary IS the serial decode.  Error cases keep the oracle's order:
// the first failing token in stream order decides the status and everything
// before it is written.
//
// Block headers, table construction (lib/de.ml:523-638, 1733-1793), stored
// blocks (lib/de.ml:1613-1627) and the zlib frame (lib/zl.ml:400-417) are
// handled wave-uniform between rounds.
#include "inflate_common.hpp"

namespace md {
namespace v2 {

constexpr uint32_t kInBytes = 4096;  // compressed-input ring in LDS
constexpr uint32_t kInWords = kInBytes / 4;
constexpr uint32_t kChunk = 1024;  // refill granularity (16 B per lane)

enum : uint32_t { ST_LIT = 0, ST_LIT2 = 1, ST_DIST = 2, ST_DIST2 = 3 };
// per-lane stop reasons (>= 1); MD_* status codes are used directly for errors
constexpr uint32_t kStopEob = 100;

constexpr uint32_t kTokMatch = 0x80000000u;  // token: bit31 match | gap[30:24] | len-3 [23:16] | dist-1 [14:0]

// optional in-kernel profile (stream 0 only): cycles per phase + event counts
enum { P_ENSURE = 0, P_DECODE1, P_DECODE2, P_EMIT_A, P_FAR, P_NEAR, P_FLUSH, P_HEADER, P_COUNT };
enum { C_ROUNDS = 0, C_PASSES, C_LANES, C_TOKENS, C_FAR, C_NEAR, C_NEAR_IT, C_COUNT };
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

template <int S_, int TMAX_, int KMAX_, int STAGE_>
struct Cfg {
  static constexpr uint32_t S = S_;          // bits per lane sub-sequence
  static constexpr uint32_t TMAX = TMAX_;    // tokens per lane per round
  static constexpr uint32_t KMAX = KMAX_;    // decode slots per lane per pass
  static constexpr uint32_t STAGE = STAGE_;  // staging bytes
  static constexpr uint32_t NEED = 8 * S_ + 64 > 640 ? 8 * S_ + 64 : 640;  // input bytes a round/header may touch
  static_assert(TMAX_ * 258 + 16 <= STAGE_, "lane 0 must always fit the staging buffer");
  static_assert(NEED + kChunk <= kInBytes, "input ring too small");
};

template <class C>
struct Smem {
  uint32_t inring[kInWords + 4];  // +1 mirror word (ring[kInWords] == ring[0]), padded
  Scratch sc;
  uint32_t tok[C::TMAX * kWave];
  alignas(16) uint8_t stage[C::STAGE + 16];  // +16: 8-byte copies may read a little past the data
  uint8_t owner[C::STAGE / 32 + 8];          // producer lane of each 32-byte staging block
};

// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = __shfl_xor(v, o);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// exclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane) {
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += t;
  }
  return x - v;
}

// ---------------------------------------------------------------------------
// Compressed input: a 4 KiB LDS ring addressed by absolute bit position.
struct Input {
  const uint8_t *p;
  uint32_t nbytes;
  uint32_t lane;
  uint32_t *ring;
  uint32_t in_hi;  // stream bytes [.., in_hi) are in the ring (multiple of kChunk)

  __device__ __forceinline__ void load_chunk() {
    uint32_t off = in_hi + lane * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (off + 16 <= nbytes) {
      __builtin_memcpy(&v, p + off, 16);
    } else if (off < nbytes) {
      uint32_t w[4] = {0, 0, 0, 0};
      for (uint32_t k = 0; k < 16 && off + k < nbytes; k++) w[k >> 2] |= (uint32_t)p[off + k] << (8 * (k & 3));
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    uint32_t r = off & (kInBytes - 1);
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(ring) + r) = v;
    if (r == 0) ring[kInWords] = v.x;  // mirror of word 0 for the wrap-around peek
    in_hi += kChunk;
  }
  __device__ __forceinline__ void reset(uint32_t byte_pos) { in_hi = byte_pos & ~(kChunk - 1); }
  __device__ __forceinline__ void ensure(uint32_t byte_pos, uint32_t need) {
    while (in_hi < byte_pos + need) load_chunk();
  }
  // 32 bits of the stream starting at absolute bit position bp (zero beyond the end)
  __device__ __forceinline__ uint32_t peek(uint32_t bp) const {
    uint32_t w = (bp >> 5) & (kInWords - 1);
    uint32_t lo = ring[w], hi = ring[w + 1];
    return __builtin_amdgcn_alignbit(hi, lo, bp & 31);
  }
};

// wave-uniform bit cursor over the ring (block headers)
struct UReader {
  const Input *in;
  uint32_t bp;
  uint32_t total;  // total real bits of the stream
  __device__ __forceinline__ int64_t avail() const { return (int64_t)total - (int64_t)bp; }
  __device__ __forceinline__ uint32_t peek(uint32_t n) const { return uni(in->peek(bp)) & ((1u << n) - 1); }
  __device__ __forceinline__ void drop(uint32_t n) { bp += n; }
};

// RFC1951 length / distance symbol -> (base, extra bits), computed instead of
// looked up (the tables are lib/de.ml:293-325; +3 / +1 biases folded in)
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

struct LaneState {
  uint32_t start, end, ntok, nb, stop;
};

// One speculative decode pass of this lane's sub-sequence [start, limit).
template <class C>
__device__ __forceinline__ void decode_pass(const Input &in, const uint16_t *lut, const Lut &lit,
                                            const Lut &dist, uint32_t *tok, uint32_t lane,
                                            uint32_t total_bits, bool go, uint32_t limit,
                                            LaneState &ls) {
  uint32_t p = ls.start, ptok = ls.start, k = 0, nb = 0, stop = 0;
  uint32_t state = ST_LIT, sub_off = 0, sub_bits = 0, mlen = 0;
  bool run = go;
  for (uint32_t slot = 0; slot < C::KMAX; ++slot) {
    if (run && state == ST_LIT && (p >= limit || k == C::TMAX)) run = false;
    if (!__any(run)) break;
    if (run) {
      const uint32_t w = in.peek(p);
      const bool isdist = state >= ST_DIST;
      const uint32_t tb = isdist ? 852u : 0u;  // dist LUT follows the lit LUT in Scratch
      const uint32_t root = isdist ? dist.root : lit.root;
      uint32_t idx;
      if (state & 1) idx = tb + sub_off + ((w >> root) & ((1u << sub_bits) - 1));
      else idx = tb + (w & (isdist ? dist.mask : lit.mask));
      const uint32_t e = lut[idx];
      if (e & kLink) {
        sub_off = e & 1023;
        sub_bits = (e >> 10) & 15;
        state |= 1;
      } else {
        const uint32_t len = (e >> 9) & 15, sym = e & 511;
        if (!isdist) {
          if (sym < 256) {
            p += len;
            if (p > total_bits) stop = MD_UNEXPECTED_END_OF_INPUT;
            else {
              tok[k * kWave + lane] = sym;
              k++;
              nb++;
              ptok = p;
            }
            state = ST_LIT;
          } else if (sym == 256) {
            p += len;
            if (p > total_bits) stop = MD_UNEXPECTED_END_OF_INPUT;
            else {
              stop = kStopEob;
              ptok = p;
            }
          } else {
            uint32_t b3, xl;
            len_sym((sym - 257) & 31, &b3, &xl);
            mlen = b3 + ((w >> len) & ((1u << xl) - 1));
            p += len + xl;
            if (p > total_bits) stop = MD_UNEXPECTED_END_OF_INPUT;
            state = ST_DIST;
          }
        } else {
          if (e == kBad) stop = MD_INVALID_DISTANCE_CODE;
          else {
            uint32_t b1, xd;
            dist_sym(sym & 31, &b1, &xd);
            const uint32_t d = b1 + ((w >> len) & ((1u << xd) - 1));
            p += len + xd;
            if (p > total_bits) stop = MD_UNEXPECTED_END_OF_INPUT;
            else if (d == 0) stop = MD_INVALID_DISTANCE_CODE;
            else {
              tok[k * kWave + lane] = kTokMatch | ((mlen - 3) << 16) | (d - 1);
              k++;
              nb += mlen;
              ptok = p;
            }
          }
          state = ST_LIT;
        }
        if (stop) run = false;
      }
    }
  }
  if (go) {
    ls.end = ptok;
    ls.ntok = k;
    ls.nb = nb;
    ls.stop = stop;
  }
}

// ---------------------------------------------------------------------------
template <class C>
struct Sink {
  uint8_t *stage;
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
        const uint4 v = *reinterpret_cast<const uint4 *>(stage + (cpos - rb));
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (hi - lo == 16) {
          __builtin_memcpy(g + cpos, &v, 16);
#pragma unroll
          for (int k = 0; k < 16; k++) {
            uint32_t d = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
            s1 += d;
            s2 += (b0 - (cpos + k)) * d;
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
// Small unaligned LDS / HBM accessors (gfx950 runs DS and global memory in
// unaligned mode: misaligned b16/b32/b64 accesses are legal — tools/probes/unaligned_lds.hip)
typedef uint64_t u64_u __attribute__((aligned(1)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef uint16_t u16_u __attribute__((aligned(1)));

__device__ __forceinline__ uint64_t lds_ld64(const uint8_t *p) { return *reinterpret_cast<const u64_u *>(p); }
__device__ __forceinline__ void lds_st64(uint8_t *p, uint64_t v) { *reinterpret_cast<u64_u *>(p) = v; }
// store the low r (< 8) bytes of v
__device__ __forceinline__ void lds_st_tail(uint8_t *p, uint64_t v, uint32_t r) {
  if (r & 4) {
    *reinterpret_cast<u32_u *>(p) = (uint32_t)v;
    p += 4;
    v >>= 32;
  }
  if (r & 2) {
    *reinterpret_cast<u16_u *>(p) = (uint16_t)v;
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
__device__ __forceinline__ void copy_near(uint8_t *dst, const uint8_t *src, uint32_t ml, uint32_t d) {
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
// Phase B of a round: place the accepted tokens.  Returns MD_OK or the status
// of the first failing token (stream order); *emitted = bytes produced.
template <class C, class PF>
__device__ __forceinline__ int emit_round(Smem<C> *sm, Sink<C> &sk, uint32_t lane, uint32_t nvalid,
                                          const LaneState &ls, uint32_t *nvalid_out,
                                          uint32_t *emitted, PF &pf) {
  uint32_t *tok = sm->tok;
  uint8_t *stage = sm->stage;
  const uint32_t R0 = sk.pos, rb = sk.sbase();
  const uint32_t cap = sk.cap;

  uint32_t mynb = lane < nvalid ? ls.nb : 0;
  uint32_t off = wave_excl_scan(mynb, lane);
  // staging capacity: keep the largest prefix of lanes that fits
  {
    uint64_t fits = __ballot(off + mynb <= C::STAGE - 16);
    uint32_t nfit = fits == ~0ull ? 64 : __builtin_ctzll(~fits);
    if (nfit < nvalid) nvalid = nfit;
  }
  if (lane >= nvalid) mynb = 0;
  const bool mine = lane < nvalid;
  const uint32_t ntok = mine ? ls.ntok : 0;
  const uint32_t q0 = R0 + off;

  // owner table: the lane that produces the first byte of every 32-byte staging
  // block — a lower bound of the producer of any byte in that block
  if (mynb) {
    uint32_t b0 = (q0 - rb + 31) >> 5, b1 = (q0 + mynb - 1 - rb) >> 5;
    for (uint32_t bb = b0; bb <= b1; bb++) sm->owner[bb] = (uint8_t)lane;
  }
  if (lane == 0) sm->owner[0] = 0;

  // (a) literals into the staging buffer, matches compacted to records with the
  //     literal gap in front of them; position-dependent checks in stream order
  uint32_t q = q0, nm = 0, gap = 0, fail = 0, good = 0;  // good = bytes before the failing token
  for (uint32_t t = 0; t < C::TMAX; t++) {
    if (!__any(t < ntok && !fail)) break;
    if (t < ntok && !fail) {
      uint32_t tk = tok[t * kWave + lane];
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
      rc = (int)__shfl(fail, fl);
      uint32_t foff = __shfl(off, fl), fgood = __shfl(good, fl);
      total = foff + fgood;
      if (lane > fl) nm = 0;  // later lanes are void
      nvalid = fl + 1;
    } else {
      total = __shfl(off + mynb, nvalid - 1);  // inclusive sum at the last accepted lane
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
          uint8_t *dd = stage + (dq[u] - rb);
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
          ja = src >= rb ? sm->owner[(src - rb) >> 5] : 0;
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
        uint8_t *dd = stage + (qm - rb);
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
  *nvalid_out = nvalid;
  *emitted = total;
  return rc;
}

// ---------------------------------------------------------------------------
// All rounds of one Huffman block.  On return *bp is the bit after the EOB.
template <class C, class PF>
__device__ int inflate_block_v2(Smem<C> *sm, Input &in, Sink<C> &sk, const Lut &lit,
                                const Lut &dist, uint32_t lane, uint32_t total_bits,
                                uint32_t *bp_io, PF &pf) {
  uint32_t bp = *bp_io;
  const uint16_t *lut = sm->sc.lit;  // lit[852] then dist[592], contiguous in Scratch
  for (;;) {
    in.ensure(bp >> 3, C::NEED);
    pf.tick(P_ENSURE);
    pf.count(C_ROUNDS);
    pf.count(C_PASSES);
    LaneState ls;
    ls.start = bp + lane * C::S;
    const uint32_t limit = bp + (lane + 1) * C::S;
    decode_pass<C>(in, lut, lit, dist, sm->tok, lane, total_bits, true, limit, ls);
    pf.tick(P_DECODE1);
    // A2: chain the lanes
    for (int it = 0; it < 3; it++) {
      uint32_t pe = __shfl_up(ls.end, 1), ps = __shfl_up(ls.stop, 1);
      bool redo = lane > 0 && ps == 0 && pe != ls.start;
      if (!__any(redo)) break;
      if (redo) ls.start = pe;
      pf.count(C_PASSES);
      decode_pass<C>(in, lut, lit, dist, sm->tok, lane, total_bits, redo, limit, ls);
    }
    pf.tick(P_DECODE2);
    uint32_t nvalid;
    {
      uint32_t pe = __shfl_up(ls.end, 1), ps = __shfl_up(ls.stop, 1);
      uint64_t bad = __ballot(lane > 0 && (ps != 0 || pe != ls.start));
      nvalid = bad ? __builtin_ctzll(bad) : 64;
    }
    uint32_t emitted;
    int rc = emit_round<C>(sm, sk, lane, nvalid, ls, &nvalid, &emitted, pf);
    sk.flush(emitted);
    pf.tick(P_FLUSH);
    pf.count(C_LANES, nvalid);
    pf.count(C_TOKENS, wave_sum(lane < nvalid ? ls.ntok : 0));
    if (rc != MD_OK) return rc;
    const uint32_t lastl = nvalid - 1;
    const uint32_t lstop = __shfl(ls.stop, lastl);
    bp = __shfl(ls.end, lastl);
    if (lstop == kStopEob) break;
    if (lstop != 0) return (int)lstop;
  }
  *bp_io = bp;
  return MD_OK;
}

// Dynamic block header (lib/de.ml:1733-1793), wave-uniform over the LDS ring.
__device__ int dynamic_header_v2(UReader &ur, Scratch *s, Lut *lit, Lut *dist, uint32_t lane) {
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

__device__ void fixed_tables_v2(Scratch *s, Lut *lit, Lut *dist, uint32_t lane) {
  for (uint32_t n = lane; n < 288; n += kWave) s->lens[n] = n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8;
  build_lut(K_LENS, s->lens, 288, s, lit, lane);
  if (lane < 32) s->dist[lane] = (uint16_t)((5u << 9) | (__brev(lane) >> 27));
  dist->t = s->dist;
  dist->mask = 31;
  dist->root = 5;
  dist->maxl = 5;
}

template <class C, bool PROF>
__global__ __launch_bounds__(kWave) void inflate_v2_kernel(
    int format, uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *__restrict__ out,
    const uint64_t *__restrict__ out_off, const uint64_t *__restrict__ out_cap,
    uint64_t *__restrict__ out_len, uint64_t *__restrict__ consumed, int32_t *__restrict__ status,
    uint32_t *__restrict__ checksum, uint64_t *__restrict__ dbg) {
  extern __shared__ __align__(16) uint8_t smem_raw[];
  Smem<C> *sm = reinterpret_cast<Smem<C> *>(smem_raw);
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
  sk.stage = sm->stage;
  sk.g = out + out_off[sid];
  sk.cap = cap;
  sk.pos = 0;
  sk.lane = lane;
  sk.a = 1;
  sk.b = 0;
  sk.want_adler = (checksum != nullptr) || format == MD_FORMAT_ZLIB;

  Input inp;
  inp.p = src + body_off;
  inp.nbytes = body_len;
  inp.lane = lane;
  inp.ring = sm->inring;
  inp.reset(0);
  const uint32_t total_bits = body_len * 8;
  uint32_t bp = 0;

  if (rc == MD_OK) {
    Lut lit, dist;
    bool last = false;
    while (!last && rc == MD_OK) {
      inp.ensure(bp >> 3, C::NEED);
      UReader ur{&inp, bp, total_bits};
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
            for (uint32_t j = lane; j < seg; j += kWave) sm->stage[s0 + j] = q[j];
            sk.flush(seg);
            q += seg;
            left -= seg;
          }
          p += len;
          bp = p * 8;
          inp.reset(p);
        }
      } else if (type == 1) {
        fixed_tables_v2(&sm->sc, &lit, &dist, lane);
        pf.tick(P_HEADER);
        rc = inflate_block_v2<C>(sm, inp, sk, lit, dist, lane, total_bits, &bp, pf);
      } else if (type == 2) {
        rc = dynamic_header_v2(ur, &sm->sc, &lit, &dist, lane);
        bp = ur.bp;
        pf.tick(P_HEADER);
        if (rc == MD_OK) rc = inflate_block_v2<C>(sm, inp, sk, lit, dist, lane, total_bits, &bp, pf);
      } else {
        rc = MD_INVALID_KIND_OF_BLOCK;
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

}  // namespace v2
}  // namespace md

extern "C" int md_launch_inflate_v2(int variant, int format, uint32_t n, const uint8_t *in,
                                    const uint64_t *in_off, const uint64_t *in_len, uint8_t *out,
                                    const uint64_t *out_off, const uint64_t *out_cap,
                                    uint64_t *out_len, uint64_t *consumed, int32_t *status,
                                    uint32_t *checksum, uint64_t *dbg, hipStream_t stream) {
  if (n == 0) return 0;
  dim3 grid(n), block(md::kWave);
#define MD_LAUNCH_V2(CFG)                                                                        \
  do {                                                                                           \
    if (dbg)                                                                                     \
      hipLaunchKernelGGL((md::v2::inflate_v2_kernel<CFG, true>), grid, block,                    \
                         sizeof(md::v2::Smem<CFG>), stream, format, n, in, in_off, in_len, out,  \
                         out_off, out_cap, out_len, consumed, status, checksum, dbg);            \
    else                                                                                         \
      hipLaunchKernelGGL((md::v2::inflate_v2_kernel<CFG, false>), grid, block,                   \
                         sizeof(md::v2::Smem<CFG>), stream, format, n, in, in_off, in_len, out,  \
                         out_off, out_cap, out_len, consumed, status, checksum, dbg);            \
  } while (0)
  using A = md::v2::Cfg<128, 15, 40, 4096>;
  using B = md::v2::Cfg<96, 15, 36, 4096>;
  using Cc = md::v2::Cfg<160, 23, 48, 6144>;
  switch (variant) {
  case 0: MD_LAUNCH_V2(A); break;
  case 1: MD_LAUNCH_V2(B); break;
  case 2: MD_LAUNCH_V2(Cc); break;
  default: return -1;
  }
#undef MD_LAUNCH_V2
  return (int)hipGetLastError();
}
