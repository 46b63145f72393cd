// inflate_wave.hip — batched RFC1951 inflate for gfx950 (MI355X): one kernel, one workgroup of two wavefronts per stream.
//
// The DECODER wavefront finds and decodes tokens (sync, emit below); the COPIER wavefront resolves the matches and
// writes the round out (copy, flush).  They share the stream's LDS (staging buffer, match records) and hand rounds
// over through a mailbox in LDS (struct Mail): while the copier works on round r the decoder is already walking the
// zones of round r + 1, and it waits for the copier only before it writes round r + 1 into the staging buffer.  One
// stream's rounds are serial, its two halves are not: a stream finishes 1.4x sooner than with one wavefront doing both
// in turn (that form is still here: md_set_option "inflate_waves" = 1, same results), and with 8 streams = 16
// wavefronts per CU the kernel is 1.13x faster on C2.
//
// The decoder's 64 lanes decode 64 consecutive zones (48 .. S bits, re-sized every round from what the last one produced) of
// the compressed block per round:
//
//   sync   Light passes that only FIND TOKEN BOUNDARIES.  Pass 0: lane i walks the tokens from bit
//          bp + i*S until it crosses bp + (i+1)*S (only lane 0 starts on a boundary, the others
//          are speculative).  Pass k >= 1: a lane whose left neighbour ended somewhere else than
//          where it started walks again from there and now also counts the bytes and matches its
//          tokens will produce.  Huffman streams re-synchronise (p ~ 0.9 inside a full zone), so the
//          chain start_i == end_{i-1} settles after ~4 passes; the consistent prefix of lanes is
//          accepted.  A chain of token boundaries that starts on a known boundary IS the serial
//          decode of the reference loop (`inflate`, lib/de.ml:1667-1712).  A walk step is one
//          32-bit peek of the input window in LDS and one 32-bit LUT entry that carries the bits
//          to skip and the descriptor (index width, base) of the table the NEXT step indexes, so
//          the step has no per-type branches or selects.
//   emit   One pass over the accepted lanes from their validated starts.  A wave prefix sum of
//          the counts has given every lane its output position and its first record, so literals
//          go straight to their final place in the LDS staging buffer and matches become (length,
//          distance, position) records in one pool, in stream order.  Anything unusual (end of
//          block, an invalid code, a distance or length the output cannot take, a full buffer)
//          stops the lane in front of the token; the token is then looked at again with the checks
//          in the reference's order.
//   copy   By record, 64 per step across the wave.  Matches whose source is older than the round
//          read it from already flushed output (L2/HBM, up to 32 bytes per record, all of the
//          round's records in flight together); matches into the round itself are listed and wait
//          on a bitmap of the staging bytes that are still to be produced; long and self-overlapping
//          ones are copied by the whole wave.
//   flush  16-byte coalesced stores, Adler-32 folded in the same pass with v_dot4_u32_u8
//          (WInf.update / tail, lib/de.ml:453-455, 499-505).
//
// The window is the output buffer itself (De.Inf.Ns semantics, lib/de.ml:1534).  Block headers
// (lib/de.ml:1733-1793) are parsed wave-uniform on the scalar unit; the Huffman tables
// (De.Inf.huffman, lib/de.ml:523-638) are built by all lanes from the canonical code.  Stored
// blocks (lib/de.ml:1613-1627) and the zlib frame (lib/zl.ml:400-417) run between rounds.  The first
// failing token in stream order decides the status and everything before it is written
// (oracle/de_inflate.c).
#include <stdlib.h>
#include <string.h>
#include "inflate_util.hpp"

namespace md {
namespace wv {

// ---- geometry ---------------------------------------------------------------------------------
constexpr uint32_t S = 264;        // bits per lane zone at most: 8.25 dwords, so the 64 cursors start on distinct LDS banks
constexpr uint32_t SMIN = 48;      // ... and at least: where the data expands so much that 64 full zones overflow the
                                   // staging buffer, the zones shrink so that all 64 lanes still have work (cost then
                                   // follows the output, not the number of zones thrown away)
constexpr uint32_t KMAX = 64;      // walk steps per lane per pass
#ifndef MD_EMIT_SPLIT
#define MD_EMIT_SPLIT 1  // the two wavefronts of a stream emit a round together, each one half of every zone (inflate_block)
#endif
#if defined(MD_DEBUG_KNOWN_BOUNDS)
#undef MD_EMIT_SPLIT
#define MD_EMIT_SPLIT 0  // (the replayed sync results carry no mid-zone boundaries)
#endif
#ifndef MD_SPLIT_B
#define MD_SPLIT_B 4  // eighths of a zone that the copier emits
#endif
#ifndef MD_RUNIN_NUM
#define MD_RUNIN_NUM 3
#define MD_RUNIN_DEN 2
#endif
constexpr uint32_t RUNIN_NUM = MD_RUNIN_NUM, RUNIN_DEN = MD_RUNIN_DEN;  // run-in of the speculative pass, as a fraction of the zone
constexpr uint32_t PASSES = 5;     // walks after the first one: at least this many are allowed, more when the zones are small
constexpr uint32_t PASS_BITS = 1600, PASSES_MAX = 16;  // (a walk costs in proportion to the zone size)
constexpr uint32_t RMAX = 768;     // match records per round, all lanes together (in stream order)
constexpr uint32_t STAGE = 5568;   // staging bytes (one round of output)
constexpr uint32_t WIN_WORDS = 544;  // input window: 31 + 64*S + 47 bits and the two words a peek touches

// LUT entry: n[4:0] | xb[8:5] | val8[16:9] | next.nbits[20:17] | next.tb[31:21]
//   n      bits this step consumes: the code (a LINK entry: the root bits, the sub-table entry behind it: the rest of
//          the code) and the extra bits that follow it
//   xb     how many of them are extra bits (the last xb of the n)
//   val8   literal byte | length base - 3 (0..255) | distance base m (0..3), base = (m << xb) + 1
//   next   the table the following step indexes: the lit/len root, the distance root, a sub-table (this is a LINK
//          entry then), or one of the two self-looping STOP entries (end of block / no such code)
// The tables lie in this order: lit/len sub-tables, distance root and sub-tables, lit/len root, the STOP entries -
// and every table starts on an EVEN entry.  So (1) `e >> 17` has the next index width in its low FIVE bits (what
// v_bfe_u32 takes as its width operand) and `e >> 21` is the next table: a step needs no masks; (2) "this step ended a
// token at or beyond `lim`, or ran into a STOP entry" is ONE unsigned compare: (e & kTbMask) | p >= (kLitB << 21) | lim.
constexpr uint32_t kLitSize = 852, kDistSize = 592;  // zlib ENOUGH (lib/de.ml:579-580)
constexpr uint32_t kLitRootMax = 512;                // the lit/len root is at most 9 bits wide (De.Inf.huffman's root)
constexpr uint32_t kLitSubB = 0;                     // lit/len sub-tables: they exist only behind a 9-bit root
constexpr uint32_t kDistB = kLitSize - kLitRootMax;  // 340
constexpr uint32_t kLitB = kDistB + kDistSize;       // 932
constexpr uint32_t kStopEobI = kLitB + kLitRootMax;  // 1444
constexpr uint32_t kStopBadI = kStopEobI + 2;
constexpr uint32_t kLutWords = kStopBadI + 2;
constexpr uint32_t kTbMask = 0xffe00000u;
constexpr uint32_t kLoopy = 0x100u;  // with the root width of a block: an incomplete code, a walk can stand still
static_assert(kDistB % 2 == 0 && kLitB % 2 == 0 && kStopEobI % 2 == 0 && kLutWords == 1448, "even table bases");
constexpr uint32_t kStEob = 100, kStTrunc = 101;  // lane stop reasons; < 100 = MD_* status
constexpr uint32_t kCountMatch = 1u << 20;         // a walk counts bytes | matches << 20 (64 lanes: < 2^20 bytes, < 2^12 matches)
constexpr uint32_t kNearBit = 0x8000u;            // match record: len-3[23:16] | near[15] | dist-1[14:0]

// The two wavefronts of a stream talk through this: the decoder posts jobs, the copier reports them done.  A wavefront's
// LDS operations execute in order, so a job's records and literals are in LDS before `emitted` says so, and the
// copier's last read of them is over before `copied` does.
struct Mail {
  uint32_t emitted;  // jobs posted by the decoder
  uint32_t copied;   // jobs finished by the copier: staging buffer and records are free again
  uint32_t kind;     // kJobRound | kJobStored | kJobQuit | kJobEmit
  uint32_t total;    // bytes the job produces
  uint32_t x;        // match records of the round | body offset of the stored bytes
  uint32_t stuck;    // the copier's defensive verdict
  uint32_t a, b;     // Adler-32 state after job `copied`
};
constexpr uint32_t kJobRound = 0, kJobStored = 1, kJobQuit = 2, kJobEmit = 3;
// kJobEmit: the copier emits the second halves of the round's zones while the decoder emits the first ones.  What it needs
// lies in `list` (the copier's own array, idle between two rounds): four words per lane - start, limit, output position,
// first record (0xffffffff: nothing to do) -, then six wave-uniform words; the lanes' results come back in the same place.
constexpr uint32_t kHxUni = 4 * 64;  // word index of the uniform part: lroot, tot, rb, R0, cap, checked
struct Smem {  // the kernel's only LDS object: it sits at LDS address 0
  uint32_t win[WIN_WORDS];
  uint32_t lut[kLutWords];
  uint32_t mrec[RMAX];                     // the round's match records in stream order
  uint16_t mpos[RMAX];                     // their staging positions
  alignas(16) uint8_t stage[STAGE + 16];   // one round of output; header scratch while a header is parsed
  uint16_t list[RMAX];                     // the round's near matches (indices into mrec), in stream order
  Mail mail;
};
struct HScratch {       // aliases the tail of Smem::win (hscratch_of)
  uint8_t lens[384];    // code lengths: lit/len symbols, then the distance symbols
  uint16_t work[320];   // symbols sorted by (code length, symbol)
  uint32_t ctr;         // sub-table allocation counter
};
static_assert(sizeof(Smem) <= 20480, "8 streams (16 wavefronts) per CU");
// A dynamic header is at most 17 + 19 x 3 + 320 x 14 bits = 570 bytes and starts in the window's first word: the window's
// tail is free while a header is parsed (the staging buffer is not: the copier wavefront may still be writing a round out)
constexpr uint32_t kHScratchAt = 1136;
static_assert(kHScratchAt >= 4 + 572 + 8 && kHScratchAt % 16 == 0 && kHScratchAt + sizeof(HScratch) <= WIN_WORDS * 4, "header scratch lives behind the header's bits");
typedef MD_LDS Smem lds_smem;
typedef MD_LDS HScratch lds_hscratch;
__device__ __forceinline__ lds_hscratch *hscratch_of(lds_smem *sm) {
  return reinterpret_cast<lds_hscratch *>(reinterpret_cast<lds_u8 *>(sm->win) + kHScratchAt);
}

#ifdef MD_DEBUG_KNOWN_BOUNDS
// Measurement build only (tools/dbg/build_known_bounds.sh): mode & 3 == 1 records what the sync
// passes of every round found (start, end, stop, counts per lane), == 2 replays the record instead of walking - the
// time of that launch is what a PERFECT sync would leave of the kernel.  mode & 4: the copier skips the matches (far and
// near) and only flushes; mode & 8: the copier does nothing at all (the decoder's time alone).  Results are only right in
// modes 0..2.
constexpr uint32_t KB_ROUNDS = 512;
__device__ uint4 *g_kb_buf;
__device__ int g_kb_mode;
#endif
__device__ __forceinline__ uint32_t mk_entry(uint32_t n, uint32_t xb, uint32_t val8, uint32_t nbits, uint32_t tb) {
  return n | (xb << 5) | (val8 << 9) | (nbits << 17) | (tb << 21);
}
__device__ __forceinline__ uint32_t e_n(uint32_t e) { return e & 31; }
__device__ __forceinline__ uint32_t e_xb(uint32_t e) { return (e >> 5) & 15; }
__device__ __forceinline__ uint32_t e_val(uint32_t e) { return (e >> 9) & 255; }
__device__ __forceinline__ uint32_t e_tb(uint32_t e) { return e >> 21; }
// a LINK entry leads to a sub-table: neither a root nor a STOP entry
__device__ __forceinline__ bool e_link(uint32_t e) { return e_tb(e) < kLitB && e_tb(e) != kDistB; }
// the state a walk starts in: "the next step indexes the lit/len root"
__device__ __forceinline__ uint32_t e_root(uint32_t lroot) { return mk_entry(0, 0, 0, lroot, kLitB); }
// leaf entries; `codelen` = bits of the code this step consumes
__device__ __forceinline__ uint32_t lit_leaf(uint32_t sym, uint32_t codelen, uint32_t lroot, uint32_t droot) {
  if (sym < 256) return mk_entry(codelen, 0, sym, lroot, kLitB);
  if (sym == 256) return mk_entry(codelen, 0, 0, 0, kStopEobI);
  const uint32_t l = (sym - 257) & 31;  // lib/de.ml:293-311 (29,30 -> length 3)
  const uint32_t xb = (l >= 8 && l < 28) ? (l - 4) >> 2 : 0;
  const uint32_t base3 = l < 8 ? l : l < 28 ? (4 + (l & 3)) << xb : l == 28 ? 255 : 0;  // length base - 3
  return mk_entry(codelen + xb, xb, base3, droot, kDistB);
}
// a distance leaf carries the base of its symbol as `m`, base = (m << xb) + 1: m = the symbol itself for 0..3, 2 or 3
// for 4..29 (lib/de.ml:313-325); the invalid symbols 30 and 31 lead to the STOP entry of "no such code" and, unlike an
// empty table slot, consume their code (slow_token tells them apart by that)
__device__ __forceinline__ uint32_t dist_leaf(uint32_t dv, uint32_t codelen, uint32_t lroot) {
  dv &= 31;
  if (dv >= 30) return mk_entry(codelen, 0, 0, 0, kStopBadI);
  const uint32_t xb = dv >= 4 ? (dv - 2) >> 1 : 0;
  const uint32_t m = dv < 4 ? dv : ((dv & 1) | 2);
  return mk_entry(codelen + xb, xb, m, lroot, kLitB);
}
// leaf value + extra bits -> distance (lib/de.ml:321-325, +1 folded in)
__device__ __forceinline__ uint32_t dist_value(uint32_t m, uint32_t xb, uint32_t x) { return (m << xb) + 1 + x; }

// ---- input window -----------------------------------------------------------------------------
// The window holds body bytes [base, base + 4*WIN_WORDS), base a multiple of 4; zero beyond the body.  A lane loads
// 32 bytes of it (and the first lanes one more word): fetch() starts the loads into registers, put() stores them to
// LDS.  A round fetches the window of the next one as soon as it knows where that begins: the loads are in flight
// while the round is handed over (or copied).
struct Window {
  uint32_t w[8], wx;
  uint32_t base;  // of the fetched words; 0xffffffff = nothing fetched
#ifdef MD_DEBUG_KNOWN_BOUNDS
  uint32_t kb_round, kb_sid;
#endif
  __device__ __forceinline__ void fetch(const uint8_t *__restrict__ body, uint32_t nbytes, uint32_t b, uint32_t lane) {
    base = b;
    const uint32_t off = b + lane * 32;
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = 0;
    if (off + 32 <= nbytes) __builtin_memcpy(w, body + off, 32);
    else if (off < nbytes)
      for (uint32_t k = 0; k < 32 && off + k < nbytes; k++) w[k >> 2] |= (uint32_t)body[off + k] << (8 * (k & 3));
    wx = 0;
    if (lane < WIN_WORDS - 512) {
      const uint32_t ox = b + 2048 + lane * 4;
      if (ox + 4 <= nbytes) __builtin_memcpy(&wx, body + ox, 4);
      else if (ox < nbytes)
        for (uint32_t k = 0; k < 4 && ox + k < nbytes; k++) wx |= (uint32_t)body[ox + k] << (8 * k);
    }
  }
  __device__ __forceinline__ void put(lds_u32 *win, uint32_t lane) const {
    lds_u32 *dst = win + lane * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k] = w[k];
    if (lane < WIN_WORDS - 512) win[512 + lane] = wx;
  }
  // the window at `b` in LDS, from the fetched words if they are the right ones
  __device__ __forceinline__ void ensure(lds_u32 *win, const uint8_t *__restrict__ body, uint32_t nbytes, uint32_t b, uint32_t lane) {
    if (base != b) fetch(body, nbytes, b, lane);
    put(win, lane);
    base = 0xffffffffu;
  }
};
// 32 bits of the window starting at window-relative bit position p
__device__ __forceinline__ uint32_t peek(const lds_u32 *win, uint32_t p) {
  const lds_u32 *q = reinterpret_cast<const lds_u32 *>(reinterpret_cast<const lds_u8 *>(win) + ((p >> 3) & ~3u));
  return __builtin_amdgcn_alignbit(q[1], q[0], p & 31);
}
__device__ __forceinline__ uint32_t lut_at(const lds_u32 *lut, uint32_t tb, uint32_t idx) {
  return *reinterpret_cast<const lds_u32 *>(reinterpret_cast<const lds_u8 *>(lut) + ((tb + idx) << 2));
}

// wave-uniform bit cursor over the window (block headers); values live in SGPRs
struct UBits {
  const lds_u32 *win;
  uint64_t buf;
  uint32_t n;   // valid bits in buf
  uint32_t wp;  // next window word
  __device__ __forceinline__ void init(const lds_u32 *w, uint32_t bp) {
    win = w;
    wp = bp >> 5;
    buf = ((uint64_t)uni(win[wp]) | ((uint64_t)uni(win[wp + 1]) << 32)) >> (bp & 31);
    n = 64 - (bp & 31);
    wp += 2;
  }
  __device__ __forceinline__ uint32_t pos() const { return wp * 32 - n; }
  __device__ __forceinline__ void fill() {  // afterwards n > 32
    if (n <= 32) {
      buf |= (uint64_t)uni(win[wp]) << n;
      n += 32;
      wp++;
    }
  }
  __device__ __forceinline__ uint32_t peekb(uint32_t k) const { return (uint32_t)buf & ((1u << k) - 1); }
  __device__ __forceinline__ void drop(uint32_t k) {
    buf >>= k;
    n -= k;
  }
};

// ---- Huffman tables from the canonical code ---------------------------------------------------
// De.Inf.huffman (lib/de.ml:523-638) accepts a set of code lengths iff it is not over-subscribed and
// is complete (or is a single 1-bit code, for the lit/len and distance alphabets), and then decodes
// the canonical prefix code; zlib's table layout is an implementation detail except for its size
// (documented divergence D3: more than ENOUGH entries => Invalid_dictionary).  All lanes build the
// walk table directly: per-length counts by ballot, codes by rank, root slots by a canonical search,
// sub-tables sized by the longest code behind each root slot (what zlib's `curr` loop computes for a
// complete code).
struct Canon {
  uint32_t cnt[16];    // codes per length
  uint32_t first[16];  // first canonical code (MSB first) per length
  uint32_t offs[16];   // rank of the first symbol of that length in `work`
  uint32_t max, min, root, left;
};
__device__ __forceinline__ uint32_t lane_rank(uint64_t m) {  // set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
template <int K>
__device__ __forceinline__ void canon_counts(const uint32_t (&len)[K], uint32_t rootpref, Canon &c) {
#pragma unroll
  for (int l = 1; l < 16; l++) {
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < K; k++) n += (uint32_t)__builtin_popcountll(__ballot(len[k] == (uint32_t)l));
    c.cnt[l] = n;
  }
  c.cnt[0] = 0;
  c.max = 0;
  c.min = 16;
#pragma unroll
  for (int l = 15; l >= 1; l--) {
    if (c.cnt[l] && c.max == 0) c.max = l;
    if (c.cnt[l]) c.min = l;
  }
  int left = 1;
  bool over = false;
  uint32_t code = 0, off = 0;
  c.first[0] = 0;
  c.offs[0] = 0;
#pragma unroll
  for (int l = 1; l < 16; l++) {
    left = (left << 1) - (int)c.cnt[l];
    over = over || left < 0;
    code = (code + c.cnt[l - 1]) << 1;
    c.first[l] = code;
    c.offs[l] = off;
    off += c.cnt[l];
  }
  c.left = over ? 0xffffffffu : (uint32_t)left;  // 0xffffffff = over-subscribed
  uint32_t root = rootpref;
  if (root > c.max) root = c.max;
  if (root < c.min) root = c.min;
  c.root = root;
}

// Builds one walk table: its root into lut[tb0 ..), its sub-tables into lut[sub0 ..).  len[k] = code length of symbol
// lane + 64k (0 beyond the alphabet).  LEAF(sym, codelen) encodes a leaf.  Returns false when the table would need
// more than `size` entries (D3).
template <int K, class LEAF>
__device__ __forceinline__ bool build_walk(const uint32_t (&len)[K], const Canon &c, lds_u32 *lut, uint32_t tb0, uint32_t sub0,
                                           uint32_t size, lds_hscratch *hs, uint32_t lane, uint32_t zero_entry, LEAF leaf) {
  const uint32_t root = c.root, rmask = (1u << root) - 1;
  // codes by rank inside their length class; symbols sorted by (length, symbol) for the root search
  uint32_t rev[K];
#pragma unroll
  for (int k = 0; k < K; k++) rev[k] = 0;
#pragma unroll
  for (int l = 1; l < 16; l++) {
    if (c.cnt[l]) {
      uint32_t base = 0;
#pragma unroll
      for (int k = 0; k < K; k++) {
        const uint64_t m = __ballot(len[k] == (uint32_t)l);
        if (len[k] == (uint32_t)l) {
          const uint32_t rank = base + lane_rank(m);
          rev[k] = __brev(c.first[l] + rank) >> (32 - l);
          hs->work[c.offs[l] + rank] = (uint16_t)(lane + 64 * k);
        }
        base += (uint32_t)__builtin_popcountll(m);
      }
    }
  }
  const uint32_t nroot = 1u << root;
  for (uint32_t i = lane; i < nroot; i += kWave) lut[tb0 + i] = 0;
  if (lane == 0) hs->ctr = 0;
  // longest code behind each root slot that leads to a sub-table
  if (c.max > root) {
#pragma unroll
    for (int k = 0; k < K; k++)
      if (len[k] > root)
        __hip_atomic_fetch_max(&lut[tb0 + (rev[k] & rmask)], len[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // root slots: a link (sub-tables get their place from an LDS counter) or the leaf found by canonical search
  for (uint32_t i = lane; i < nroot; i += kWave) {
    const uint32_t v = lut[tb0 + i];
    uint32_t e;
    if (v) {
      const uint32_t sub = v - root;
      const uint32_t off = __hip_atomic_fetch_add(&hs->ctr, 1u << sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      e = mk_entry(root, 0, 0, sub, (sub0 + off) & 2047);  // (sub-table sizes are even, so are their places)
    } else {
      const uint32_t cw = __brev(i) >> (32 - root);  // the root bits as an MSB-first code prefix
      uint32_t fl = 0, ft = 0;
#pragma unroll
      for (int l = 15; l >= 1; l--) {  // at most one length matches in a prefix code
        if (c.cnt[l] && (uint32_t)l <= root) {
          const uint32_t t = (cw >> (root - l)) - c.first[l];
          if (t < c.cnt[l]) {
            fl = l;
            ft = c.offs[l] + t;
          }
        }
      }
      e = fl ? leaf((uint32_t)hs->work[ft], fl) : zero_entry;
    }
    lut[tb0 + i] = e;
  }
  const uint32_t used = nroot + uni(hs->ctr);
  if (used > size) return false;  // D3
  // sub-table entries
  if (c.max > root) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      if (len[k] > root) {
        const uint32_t link = lut[tb0 + (rev[k] & rmask)];
        const uint32_t sub = (link >> 17) & 15, tb = link >> 21;
        const uint32_t e = leaf(lane + 64 * k, len[k] - root);
        for (uint32_t j = rev[k] >> root; j < (1u << sub); j += 1u << (len[k] - root)) lut[tb + j] = e;
      }
    }
  }
  return true;
}

// the order of the code-length code lengths (lib/de.ml:225-231), 5 bits each
constexpr uint64_t kZig0 = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                           10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
constexpr uint64_t kZig1 = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;

// Dynamic block header (lib/de.ml:1733-1793) at window bit `bp`.  On MD_OK the walk tables are in lut,
// *lroot_out is the index width of the lit/len root table and *bp_out the bit after the header.
// *lroot_out also says (bit 8) that one of the two codes is incomplete: a walk over its zero entries needs a step budget.
template <class PF>
__device__ __noinline__ int dynamic_tables(lds_smem *sm, uint32_t bp_arg, uint32_t tot_arg, uint32_t lane, uint32_t *bp_out,
                                           uint32_t *lroot_out, PF &pf) {
  // arguments of a function arrive in vector registers: say that these two are wave-uniform, or the whole header
  // parse below is compiled as divergent vector code instead of scalar code
  const uint32_t bp = uni(bp_arg), tot = uni(tot_arg);
  const lds_u32 *win = (const lds_u32 *)sm->win;
  lds_hscratch *hs = hscratch_of(sm);
  UBits ub;
  ub.init(win, bp);
  if ((int32_t)(tot - ub.pos()) < 14) return MD_UNEXPECTED_END_OF_INPUT;
  const uint32_t hlit = ub.peekb(5) + 257;
  ub.drop(5);
  const uint32_t hdist = ub.peekb(5) + 1;
  ub.drop(5);
  const uint32_t hclen = ub.peekb(4) + 4;
  ub.drop(4);
  // code-length code lengths: lane j holds the length of symbol j
  uint32_t cl[1] = {0};
  for (uint32_t i = 0; i < hclen; i++) {
    ub.fill();
    if ((int32_t)(tot - ub.pos()) < 3) return MD_UNEXPECTED_END_OF_INPUT;
    const uint32_t v = ub.peekb(3);
    ub.drop(3);
    const uint32_t z = (uint32_t)((i < 12 ? kZig0 >> (5 * i) : kZig1 >> (5 * (i - 12))) & 31);
    if (lane == z) cl[0] = v;
  }
  // its decode table: 128 direct entries (root 7 >= longest code), two per lane: sym | len << 8, 0xffff = unreachable
  Canon cc;
  canon_counts<1>(cl, 7, cc);
  uint32_t cmaxl, t_lo, t_hi;
  if (cc.max == 0) {  // empty_table (lib/de.ml:521): a 1-bit code for symbol 0, the other slot out of bounds (D2)
    cmaxl = 1;
    t_lo = lane == 0 ? (1u << 8) : 0xffffu;
    t_hi = 0xffffu;
  } else {
    if (cc.left != 0) return MD_INVALID_DICTIONARY;  // over-subscribed or incomplete
    cmaxl = cc.max;
    // rank of each symbol inside its length class, then the table by canonical search (codes <= 7 bits)
#pragma unroll
    for (int l = 1; l < 8; l++) {
      const uint64_t m = __ballot(cl[0] == (uint32_t)l);
      if (cl[0] == (uint32_t)l) hs->work[cc.offs[l] + lane_rank(m)] = (uint16_t)lane;
    }
    uint32_t t[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t i = lane + 64 * h;
      const uint32_t cw = __brev(i & ((1u << cmaxl) - 1)) >> (32 - cmaxl);
      uint32_t fl = 0, ft = 0;
#pragma unroll
      for (int l = 7; l >= 1; l--) {
        if (cc.cnt[l] && (uint32_t)l <= cmaxl) {
          const uint32_t x = (cw >> (cmaxl - l)) - cc.first[l];
          if (x < cc.cnt[l]) {
            fl = l;
            ft = cc.offs[l] + x;
          }
        }
      }
      t[h] = fl ? ((uint32_t)hs->work[ft] | (fl << 8)) : 0xffffu;
    }
    t_lo = t[0];
    t_hi = t[1];
  }
  // the hlit + hdist code lengths, run-length coded (lib/de.ml:1733-1769)
  const uint32_t max_res = hlit + hdist;
  for (uint32_t x = lane; x < 384; x += kWave) hs->lens[x] = 0;
  // One scalar step per symbol.  With enough input left for the longest header there can be (316 symbols of 7 + 7
  // bits) the end-of-input tests are left out of the loop.
  auto run_lengths = [&](auto checked) -> int {
    constexpr bool CHECK = decltype(checked)::value;
    uint32_t i = 0, prev = 0;
    while (i < max_res) {
      ub.fill();
      if (CHECK && (int32_t)(tot - ub.pos()) < (int32_t)cmaxl) return MD_UNEXPECTED_END_OF_INPUT;
      const uint32_t idx = ub.peekb(cmaxl);
      const uint32_t e_lo = __builtin_amdgcn_readlane(t_lo, idx & 63), e_hi = __builtin_amdgcn_readlane(t_hi, idx & 63);
      const uint32_t e = idx < 64 ? e_lo : e_hi;
      if (e == 0xffffu) return MD_INVALID_DICTIONARY;
      const uint32_t sym = e & 0xff;
      ub.drop(e >> 8);
      if (sym < 16) {
        hs->lens[i] = (uint8_t)sym;  // every lane stores the same byte: no exec-mask change on the scalar path
        prev = sym;
        i++;
      } else {
        const uint32_t nb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
        if (sym == 16 && i == 0) return MD_INVALID_DICTIONARY;
        ub.fill();
        if (CHECK && (int32_t)(tot - ub.pos()) < (int32_t)nb) return MD_UNEXPECTED_END_OF_INPUT;
        const uint32_t copy = ub.peekb(nb) + (sym == 18 ? 11 : 3);
        ub.drop(nb);
        const uint32_t val = sym == 16 ? prev : 0;
        if (i + copy > max_res) return MD_INVALID_DICTIONARY;
        if (val)
          for (uint32_t x = lane; x < copy; x += kWave) hs->lens[i + x] = (uint8_t)val;
        prev = val;
        i += copy;
      }
    }
    return MD_OK;
  };
  {
    const bool plenty = (int32_t)(tot - ub.pos()) >= 316 * 14 + 64;
    const int rc = plenty ? run_lengths(std::false_type{}) : run_lengths(std::true_type{});
    if (rc != MD_OK) return rc;
  }
  *bp_out = ub.pos();
  pf.tick_lds(P_HDR_LENS);
  if (uni(hs->lens[256]) == 0) return MD_INVALID_DICTIONARY;
  // the two alphabets, lane-parallel
  uint32_t ll[5], dl[1];
#pragma unroll
  for (int k = 0; k < 5; k++) ll[k] = lane + 64 * k < hlit ? (uint32_t)hs->lens[lane + 64 * k] : 0u;
  dl[0] = lane < hdist ? (uint32_t)hs->lens[hlit + lane] : 0u;
  Canon cl_, cd_;
  canon_counts<5>(ll, 9, cl_);
  canon_counts<1>(dl, 6, cd_);
  // De.Inf.huffman's verdicts (lib/de.ml:549-550): over-subscribed, or incomplete unless the longest code is 1 bit
  if (cl_.left == 0xffffffffu || (cl_.left > 0 && cl_.max != 1)) return MD_INVALID_DICTIONARY;
  const uint32_t lroot = cl_.root;
  uint32_t droot;
  lds_u32 *lut = (lds_u32 *)sm->lut;
  if (cd_.max == 0) {
    droot = 1;
  } else {
    if (cd_.left == 0xffffffffu || (cd_.left > 0 && cd_.max != 1)) return MD_INVALID_DICTIONARY;
    droot = cd_.root;
  }
  if (!build_walk<5>(ll, cl_, lut, kLitB, kLitSubB, kLitSize, hs, lane, e_root(lroot),
                     [=](uint32_t sym, uint32_t codelen) { return lit_leaf(sym, codelen, lroot, droot); }))
    return MD_INVALID_DICTIONARY;
  pf.tick_lds(P_HDR_LIT);
  if (cd_.max == 0) {  // empty_table: symbol 0 on a 1-bit code, the other slot is an error (D2)
    if (lane == 0) {
      lut[kDistB] = dist_leaf(0, 1, lroot);
      lut[kDistB + 1] = mk_entry(0, 0, 0, 0, kStopBadI);
    }
  } else if (!build_walk<1>(dl, cd_, lut, kDistB, kDistB + (1u << droot), kDistSize, hs, lane, e_root(lroot),
                            [=](uint32_t sym, uint32_t codelen) { return dist_leaf(sym, codelen, lroot); }))
    return MD_INVALID_DICTIONARY;
  const bool loopy = cl_.left > 0 || (cd_.max != 0 && cd_.left > 0);
  *lroot_out = lroot | (loopy ? kLoopy : 0u);
  return MD_OK;
}

// fixed_lit / fixed_dist (lib/de.ml:821-833): 288 lit/len codes of 8/9/7/8 bits, 32 distance codes of 5 bits
__device__ __noinline__ void fixed_tables(lds_smem *sm, uint32_t lane, uint32_t *lroot_out) {
  lds_hscratch *hs = hscratch_of(sm);
  uint32_t ll[5], dl[1];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint32_t n = lane + 64 * k;
    ll[k] = n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : n < 288 ? 8 : 0;
  }
  dl[0] = lane < 32 ? 5 : 0;
  Canon cl_, cd_;
  canon_counts<5>(ll, 9, cl_);
  canon_counts<1>(dl, 6, cd_);
  const uint32_t lroot = cl_.root, droot = cd_.root;
  lds_u32 *lut = (lds_u32 *)sm->lut;
  build_walk<5>(ll, cl_, lut, kLitB, kLitSubB, kLitSize, hs, lane, e_root(lroot),
                [=](uint32_t sym, uint32_t codelen) { return lit_leaf(sym, codelen, lroot, droot); });
  build_walk<1>(dl, cd_, lut, kDistB, kDistB + (1u << droot), kDistSize, hs, lane, e_root(lroot),
                [=](uint32_t sym, uint32_t codelen) { return dist_leaf(sym, codelen, lroot); });
  *lroot_out = lroot;
}

// ---- the walks ----------------------------------------------------------------------------------
// A lane's bit cursor: two consecutive window words in registers and the one after them on its way (fetched a step
// ahead), so the only LDS access a step waits for is its table entry.
struct Cursor {
  uint32_t w0, w1, w2, wa;  // window words at byte address wa, wa + 4, wa + 8
  __device__ __forceinline__ void init(const lds_u32 *win, uint32_t p) {
    wa = (p >> 5) << 2;
    const lds_u32 *q = reinterpret_cast<const lds_u32 *>(reinterpret_cast<const lds_u8 *>(win) + wa);
    w0 = q[0];
    w1 = q[1];
    w2 = q[2];
  }
  __device__ __forceinline__ uint32_t peek(uint32_t p) const { return __builtin_amdgcn_alignbit(w1, w0, p); }  // (uses p & 31)
  __device__ __forceinline__ void seek(const lds_u32 *win, uint32_t pn) {  // pn at most one word further on
    const uint32_t wan = (pn >> 5) << 2;
    const bool ge = wan != wa;
    w0 = ge ? w1 : w0;
    w1 = ge ? w2 : w1;
    wa = wan;
    w2 = *reinterpret_cast<const lds_u32 *>(reinterpret_cast<const lds_u8 *>(win) + wa + 8);
  }
};
// the table entry the step in state `e` (the entry of the step before) finds at the bits w
__device__ __forceinline__ uint32_t lut_step(const lds_u32 *lut, uint32_t e, uint32_t w) {
  return lut_at(lut, e >> 21, __builtin_amdgcn_ubfe(w, 0, e >> 17));  // (v_bfe_u32 takes the low 5 bits of the width: the next index width)
}

// One token-boundary walk of this lane's zone [start, limit).  COUNT adds what the tokens produce: a byte per return
// to the lit/len root (a literal, or the last byte of a match), length - 1 and a match at every length code.
// A divergent per-lane loop: finished lanes leave the exec mask, the wave leaves when it is empty.  The walk goes on
// while `key` < `thr` (see the table layout); every step of a complete code consumes a bit, so it ends.  BUDGET: the
// block has an incomplete code (allowed when the only code is 1 bit long, lib/de.ml:549-550), whose unused slot
// consumes nothing (lib/de.ml:521: a zero entry is "0 bits, symbol 0"): the walk is then limited to KMAX steps.
// MID (counting passes): also the first token boundary at or beyond `mthr` (the middle of the zone) and what the tokens
// before it produce - where the zone is cut when the stream's two wavefronts emit it together (0xffffffff: none, the zone's
// tokens end before the middle or the walk stopped there).
template <bool COUNT, bool BUDGET, bool MID = false>
__device__ __forceinline__ void sync_pass(const lds_u32 *win, const lds_u32 *lut, uint32_t lroot, bool go, uint32_t start,
                                          uint32_t limit, uint32_t &end, uint32_t &stop, uint32_t &nb, uint32_t mthr = 0,
                                          uint32_t *midp = nullptr, uint32_t *midcnt = nullptr) {
  if (go) {
    uint32_t p = start, e = e_root(lroot), cnt = 0;
    uint32_t mp = 0xffffffffu, mc = 0;
    bool got = false;
    if (MID && start >= mthr) {  // (the token before reached beyond the middle: everything is the second half's)
      mp = start;
      got = true;
    }
    if (p < limit) {
      const uint32_t thr = (kLitB << 21) | limit;
      uint32_t slot = 0, key;
      Cursor c;
      c.init(win, start);
      do {
        const uint32_t w = c.peek(p);
        const uint32_t en = lut_step(lut, e, w);
        const uint32_t n = e_n(en);
        if (COUNT) {  // bytes in the low 20 bits, matches above (kCountMatch)
          const uint32_t xb = e_xb(en);
          const uint32_t len1 = e_val(en) + __builtin_amdgcn_ubfe(w, n - xb, xb) + (kCountMatch + 2);  // length - 1, and a match
          cnt += (e_tb(en) == kLitB ? 1u : 0u) + (e_tb(en) == kDistB ? len1 : 0u);
        }
        p += n;
        if (MID) {
          const bool cross = (e_tb(en) == kLitB) & !got & (p >= mthr);
          mp = cross ? p : mp;
          mc = cross ? cnt : mc;
          got = got | cross;
        }
        c.seek(win, p);
        e = en;
        key = (en & kTbMask) | p;
        if (BUDGET && ++slot >= KMAX && e_tb(en) == kLitB) break;
      } while (key < thr);
    }
    end = p;
    stop = e_tb(e) >= kStopEobI ? e_tb(e) : 0u;
    if (COUNT) nb = cnt;
    if (MID) {
      *midp = mp;
      *midcnt = mc;
    }
  }
}

struct LaneOut {
  uint32_t endp;   // bit after the last token taken (a token boundary)
  uint32_t stopc;  // 0 = zone done, kStEob, kStTrunc (round capacity), else MD_* status of the failing token
  uint32_t bytes;  // bytes produced (up to the failing token)
  uint32_t nm;     // match records written (from the lane's first record on)
};

// The token at ptok, with every check in the oracle's order (oracle/de_inflate.c ns_inflate_block): called for
// the lanes the emit pass stopped.  Returns the stop reason (< 256) | the bit after an end-of-block code << 8 - in ONE
// value: as an out-parameter of this call the caller's variable lived in scratch memory, a store and a load of every
// emit pass whether anything had stopped or not.
__device__ __noinline__ uint32_t slow_token(const lds_u32 *win, const lds_u32 *lut, uint32_t lroot, uint32_t ptok, uint32_t q,
                                            uint32_t tot, uint32_t cap) {
  uint32_t p = ptok;
  uint32_t w = peek(win, p);
  uint32_t e = lut_step(lut, e_root(lroot), w);
  if (e_link(e)) {
    p += e_n(e);
    w = peek(win, p);
    e = lut_step(lut, e, w);
  }
  uint32_t n = e_n(e), xb = e_xb(e), ntb = e_tb(e);
  uint32_t pn = p + n;
  if (pn > tot) return MD_UNEXPECTED_END_OF_INPUT;  // D1
  if (ntb == kStopEobI) return kStEob | (pn << 8);
  if (ntb != kDistB) {  // literal
    if (q >= cap) return MD_UNEXPECTED_END_OF_OUTPUT;
    return kStTrunc;
  }
  const uint32_t mlen = e_val(e) + 3 + __builtin_amdgcn_ubfe(w, n - xb, xb);
  p = pn;
  w = peek(win, p);
  e = lut_step(lut, e, w);
  if (e_tb(e) == kStopBadI && e_n(e) == 0) return MD_INVALID_DISTANCE_CODE;  // D2: an empty slot
  if (e_link(e)) {
    p += e_n(e);
    w = peek(win, p);
    e = lut_step(lut, e, w);
  }
  n = e_n(e), xb = e_xb(e);
  pn = p + n;
  if (pn > tot) return MD_UNEXPECTED_END_OF_INPUT;
  if (e_tb(e) == kStopBadI) return MD_INVALID_DISTANCE_CODE;  // the symbols 30 and 31
  const uint32_t d = dist_value(e_val(e), xb, __builtin_amdgcn_ubfe(w, n - xb, xb));
  const uint32_t lim = q < 32768u ? q : 32768u;
  if (d > lim) return MD_INVALID_DISTANCE;
  if (mlen > cap - q) return MD_UNEXPECTED_END_OF_OUTPUT;
  return kStTrunc;  // the token is fine: the round's staging buffer is full
}

// The emit pass: the same walk from a validated start, producing output.  Straight-line per step; anything
// unusual stops the lane in front of the token (ptok) and is classified afterwards by slow_token.  CHECKED = false
// leaves out the tests the caller has ruled out for the whole round: the end of the input is beyond the window, the
// output position is past 32 KiB (no distance can reach before the start) and everything counted fits the staging
// buffer and the output capacity; the only stops left are the STOP entries, which end the loop like the zone's end
// does.  A length code leaves the match length in mlen until its distance code returns to the root: a step that
// returns to the root with mlen == 0 is a literal.
template <bool CHECKED, bool BUDGET>
__device__ __forceinline__ void emit_pass(const lds_u32 *win, const lds_u32 *lut, uint32_t lroot, lds_u32 *mrec,
                                          lds_u16 *mpos, lds_u8 *stage, uint32_t rec0, uint32_t tot, bool go, uint32_t start,
                                          uint32_t limit, uint32_t q0, uint32_t rb, uint32_t R0, uint32_t cap, LaneOut &lo) {
  // rec0 = the lane's first record: the walk before counted the matches, the lanes' records follow each other
  uint32_t p = start, ptok = start, e = e_root(lroot);
  uint32_t q = q0, rec = rec0, mlen = 0;
  bool stopped = false;
  const uint32_t qlim = rb + STAGE - 16;  // the staging buffer holds output positions [rb, qlim)
  const uint32_t qmax = cap < qlim ? cap : qlim;
  if (go && p < limit) {
    const uint32_t thr = (kLitB << 21) | limit;
    uint32_t slot = 0, key;
    Cursor c;
    c.init(win, start);
    do {
      const uint32_t w = c.peek(p);
      const uint32_t en = lut_step(lut, e, w);
      const uint32_t n = e_n(en), xb = e_xb(en), ntb = e_tb(en);
      const uint32_t x = __builtin_amdgcn_ubfe(w, n - xb, xb);
      const uint32_t pn = p + n;
      const bool to_root = ntb == kLitB, is_len = ntb == kDistB;
      const bool mat = to_root & (mlen != 0), lit = to_root & (mlen == 0);
      const uint32_t d1 = (e_val(en) << xb) + x;  // distance - 1 (when this is the distance step)
      if (CHECKED) {
        const uint32_t lim = q < 32768u ? q : 32768u;
        const uint32_t need = mat ? mlen : 1u;
        stopped = (ntb >= kStopEobI) | (pn > tot) | (mat & (d1 >= lim)) | (to_root & (q + need > qmax));
        if (stopped) break;
      }
      stage[lit ? q - rb : STAGE + 15] = (uint8_t)(en >> 9);  // (a slack byte when the step is not a literal: no branch)
      if (mat) {
        mrec[rec] = ((mlen - 3) << 16) | ((q - d1 + mlen > R0 + 1) ? kNearBit : 0u) | d1;
        mpos[rec] = (uint16_t)(q - rb);
        rec++;
      }
      q += to_root ? (mlen > 1u ? mlen : 1u) : 0u;
      mlen = is_len ? e_val(en) + 3 + x : to_root ? 0u : mlen;
      p = pn;
      ptok = to_root ? pn : ptok;  // the start of the token the next step belongs to
      c.seek(win, p);
      e = en;
      key = (en & kTbMask) | p;
      if (BUDGET && ++slot >= KMAX && to_root) break;
    } while (key < thr);
    // (unchecked: the walk has gone through the code that leads to a STOP entry - an end-of-block code or an invalid
    // distance symbol; nothing was written for that token, which began at ptok)
    if (!CHECKED) stopped = e_tb(e) >= kStopEobI;
  }
  uint32_t stopc = 0, endp = p;
  if (go && stopped) {
    const uint32_t r = slow_token(win, lut, lroot, ptok, q, tot, cap);
    stopc = r & 0xffu;
    endp = stopc == kStEob ? r >> 8 : ptok;
  }
  if (go) {
    lo.endp = endp;
    lo.stopc = stopc;
    lo.bytes = q - q0;
    lo.nm = rec - rec0;
  }
}

// ---- output -------------------------------------------------------------------------------------
struct Sink {
  lds_u8 *stage;
  uint8_t *g;
  uint32_t cap;
  uint32_t pos;  // bytes produced and flushed
  uint32_t lane;
  uint32_t a, b;
  bool want_adler;
  __device__ __forceinline__ uint32_t sbase() const { return pos & ~15u; }  // staging index of position x is x - sbase()

  // write stage[...] for positions [pos, pos+total) to HBM, fold Adler-32, advance pos: whole 16-byte chunks one per
  // lane, the bytes of the (at most two) partial chunks at either end one per lane
  __device__ __forceinline__ void flush(uint32_t total) {
    const uint32_t rb = sbase();
    const uint32_t endp = pos + total;
    const uint32_t full0 = (pos + 15) & ~15u, full1 = endp & ~15u;  // whole chunks: [full0, full1) (if full0 < full1)
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t ps = full0; ps < full1; ps += 1024) {
      const uint32_t cpos = ps + lane * 16;
      if (cpos < full1) {
        const lds_u32 *sp = reinterpret_cast<const lds_u32 *>(stage + (cpos - rb));  // 16-byte aligned
        const uint32_t w0 = sp[0], w1 = sp[1], w2 = sp[2], w3 = sp[3];
        const uint4 v = make_uint4(w0, w1, w2, w3);
        __builtin_memcpy(g + cpos, &v, 16);
        if (want_adler) {
          // sum d_k and sum k*d_k over the 16 bytes
          uint32_t t1 = __builtin_amdgcn_udot4(w0, 0x01010101u, 0u, false);
          t1 = __builtin_amdgcn_udot4(w1, 0x01010101u, t1, false);
          t1 = __builtin_amdgcn_udot4(w2, 0x01010101u, t1, false);
          t1 = __builtin_amdgcn_udot4(w3, 0x01010101u, t1, false);
          uint32_t tk = __builtin_amdgcn_udot4(w0, 0x03020100u, 0u, false);
          tk = __builtin_amdgcn_udot4(w1, 0x07060504u, tk, false);
          tk = __builtin_amdgcn_udot4(w2, 0x0b0a0908u, tk, false);
          tk = __builtin_amdgcn_udot4(w3, 0x0f0e0d0cu, tk, false);
          s1 += t1;
          s2 += (endp - cpos) * t1 - tk;
        }
      }
    }
    {  // lanes 0..15: the bytes in front of the first whole chunk; lanes 16..31: the bytes behind the last one
      const uint32_t head1 = full0 < endp ? full0 : endp;     // head: [pos, head1)
      const uint32_t tail0 = full1 > head1 ? full1 : head1;   // tail: [tail0, endp)
      const uint32_t x = lane < 16 ? pos + lane : tail0 + (lane - 16);
      if (lane < 16 ? x < head1 : (lane < 32 && x < endp)) {
        const uint32_t d = stage[x - rb];
        g[x] = (uint8_t)d;
        s1 += d;
        s2 += (endp - x) * d;
      }
    }
    if (want_adler) {  // per lane: <= 96 bytes x 255 x 6160 < 2^32
      s1 = wave_sum(s1);
      s2 = wave_sum(s2 % 65521u);
      b = (b + total * a + s2) % 65521u;
      a = (a + s1) % 65521u;
    }
    pos = endp;
  }
};

// Far matches (the whole source is older than this round: final in HBM/L2 once the flush of earlier rounds has
// been waited for; d >= ml) and the heads of matches that straddle the round start.  The round's records are taken
// in stream order, one per lane and row of 64.  FIRST the loads of ALL rows are started - 2 LDS reads per record, one
// unaligned 16-byte load each: the source lies in L2 or beyond, two microseconds away on a busy chip, and a round
// should wait for that once, not once per group of rows (three sets of four rows, two of them in flight, spent more
// time waiting than working) - and while they are in flight the near matches are listed,
// in stream order, for copy_near_all (*nnear_out of them).  THEN the rows are gone through again as their data
// arrives: the record is read once more (cheaper than keeping three more registers per row) and its bytes are stored
// with exact-length LDS stores; bytes 16 .. 31 of the longer records follow in a second batch of the same kind, and
// what is beyond 32 bytes is copied by the whole wave, 8 bytes per lane, four records at a time.
// The loads read whole 8-byte words, up to 15 bytes past a record's source: the caller makes sure that stays inside
// the output buffer.
constexpr int FR = RMAX / kWave;  // rows of a full record pool
struct FarRow {
  uint32_t qs, n, src;
  bool near, has;
  uint32_t ml;
};
__device__ __forceinline__ FarRow far_row(const lds_u32 *mrec, const lds_u16 *mpos, uint32_t r, uint32_t nrec, uint32_t rb, uint32_t R0) {
  FarRow f;
  f.has = r < nrec;
  const uint32_t tk = mrec[f.has ? r : 0u];
  f.qs = mpos[f.has ? r : 0u];
  const uint32_t d = (tk & 0x7fff) + 1;
  f.ml = ((tk >> 16) & 0xff) + 3;
  const uint32_t s = rb + f.qs - d;
  f.near = f.has && (tk & kNearBit);
  const uint32_t head = s < R0 ? R0 - s : 0u;  // a near match that begins before the round start: its head is final too
  f.n = !f.has ? 0u : f.near ? head : f.ml;
  f.src = f.n ? s : 0u;  // (a lane without a record reads g[0..15]: no branch around the load)
  return f;
}
template <class PF>
__device__ __forceinline__ void copy_far(const lds_u32 *mrec, const lds_u16 *mpos, lds_u16 *list, const Sink &sk,
                                         uint32_t lane, uint32_t nrec, uint32_t *nnear_out, PF &pf) {
  lds_u8 *stage = sk.stage;
  const uint32_t R0 = sk.pos, rb = sk.sbase();
  const uint8_t *g = sk.g;
  uint32_t nnear = 0;
  uint64_t v0[FR], v1[FR];
  // ONE pass over the records' rows: the loads of a row are started, the row's near matches and long records are listed
  // while they fly, and what the stores will need - staging position and byte count - stays in a register per row, so
  // that a record is read from LDS once, not three times (rounds 3-4: a pass to start the loads, one to list, one to store;
  // the same time within noise, seven registers fewer).  Straight-line: a load behind a branch makes the compiler wait
  // for it at the join; a row beyond the last record reads g[0..15] with every lane, one cache line
  uint32_t pk[FR];
  uint32_t nlong = 0;
  bool spill = false;
#pragma unroll
  for (int u = 0; u < FR; u++) {
    const FarRow f = far_row(mrec, mpos, u * kWave + lane, nrec, rb, R0);
    v0[u] = out_ld64(g + f.src);
    v1[u] = out_ld64(g + f.src + 8);
    pk[u] = f.qs | (f.n << 16);
    const uint64_t nb = __ballot(f.near);
    if (f.near) list[nnear + lane_rank(nb)] = (uint16_t)(u * kWave + lane);
    nnear += (uint32_t)__builtin_popcountll(nb);
    const uint64_t lb = __ballot(f.n > 16);
    const uint32_t lc = (uint32_t)__builtin_popcountll(lb);
    if (nnear + nlong + lc > RMAX) {  // the two lists would meet (a record can be in both): the long ones go the slow way
      spill = true;
      nlong = 0;
    }
    if (lb && !spill) {
      if (f.n > 16) list[RMAX - 1 - (nlong + lane_rank(lb))] = (uint16_t)(u * kWave + lane);
      nlong += lc;
    }
  }
  pf.tick_lds(P_FAR_REC);
#pragma unroll
  for (int u = 0; u < FR; u++) {
    if ((uint32_t)u * kWave < nrec) {
      const uint32_t qs = pk[u] & 0xffffu, n = pk[u] >> 16;
      if constexpr (PF::on) {
        if (u == 0) {
          asm volatile("" ::"v"(v0[u]), "v"(v1[u]));
          pf.tick_all(P_FAR_LOAD);
        }
      }
      if (n) {
        lds_u8 *dd = stage + qs;
        lds_put(dd, v0[u], n < 8 ? n : 8);
        if (n > 8) lds_put(dd + 8, v1[u], n < 16 ? n - 8 : 8);
      }
    }
  }
  pf.tick_lds(P_FAR);
  // bytes 16 .. 31 of the long ones: one record per lane again, all loads first
  for (uint32_t c0 = 0; c0 < nlong; c0 += kWave) {
    const bool has = c0 + lane < nlong;
    const uint32_t r = has ? list[RMAX - 1 - (c0 + lane)] : 0u;
    const FarRow f = far_row(mrec, mpos, r, nrec, rb, R0);
    const uint32_t n = has ? f.n : 0u;  // > 16 where there is a record
    const uint64_t w0 = out_ld64(g + (n ? f.src + 16 : 0u)), w1 = out_ld64(g + (n ? f.src + 24 : 0u));
    if (n) {
      lds_u8 *dd = stage + f.qs + 16;
      lds_put(dd, w0, n < 24 ? n - 16 : 8);
      if (n > 24) lds_put(dd + 8, w1, n < 32 ? n - 24 : 8);
    }
    // what is beyond 32 bytes: by the whole wave, 8 bytes per lane, four records' loads in flight at a time
    for (uint64_t lm = __ballot(n > 32); lm;) {
      uint32_t ln[4], lsrc[4], lqs[4];
      uint64_t d[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool any = lm != 0;
        const uint32_t l = any ? (uint32_t)__builtin_ctzll(lm) : 0u;
        lm &= lm - 1;  // (0 stays 0)
        ln[k] = any ? rdlane(n, l) : 0u;
        lsrc[k] = rdlane(f.src, l);
        lqs[k] = rdlane(f.qs, l);
      }
      const uint32_t j = 32 + lane * 8;
#pragma unroll
      for (int k = 0; k < 4; k++) d[k] = out_ld64(g + (j < ln[k] ? lsrc[k] + j : 0u));
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (j < ln[k]) lds_put(stage + lqs[k] + j, d[k], ln[k] - j < 8 ? ln[k] - j : 8);
    }
  }
  if (spill) {  // (never seen: more long records than `list` has room for) one record at a time, all its bytes beyond 16
    for (uint32_t c0 = 0; c0 < nrec; c0 += kWave) {
      const FarRow f = far_row(mrec, mpos, c0 + lane, nrec, rb, R0);
      for (uint64_t lm = __ballot(f.n > 16); lm; lm &= lm - 1) {
        const uint32_t l = (uint32_t)__builtin_ctzll(lm);
        const uint32_t ln = rdlane(f.n, l), lsrc = rdlane(f.src, l), lqs = rdlane(f.qs, l);
        const uint32_t j = 16 + lane * 8;
        if (j < ln) lds_put(stage + lqs + j, out_ld64(g + lsrc + j), ln - j < 8 ? ln - j : 8);
      }
    }
  }
  pf.tick_all(P_FAR);
  *nnear_out = nnear;
}
// the same for the last rounds of a stream, where an 8-byte load could reach past the output buffer: guarded loads
template <class PF>
__device__ __noinline__ void copy_far_guarded(const lds_u32 *mrec, const lds_u16 *mpos, lds_u16 *list,
                                              lds_u8 *stage, const uint8_t *g, uint32_t R0, uint32_t rb, uint32_t cap,
                                              uint32_t lane, uint32_t nrec, uint32_t *nnear_out) {
  uint32_t nnear = 0;
  for (uint32_t c0 = 0; c0 < nrec; c0 += kWave) {
    const uint32_t r = c0 + lane;
    const bool has = r < nrec;
    const uint32_t tk = mrec[has ? r : 0u], qs = mpos[has ? r : 0u];
    const uint32_t d = (tk & 0x7fff) + 1, ml = ((tk >> 16) & 0xff) + 3;
    const uint32_t s = rb + qs - d;
    const bool near = has && (tk & kNearBit);
    uint32_t n = has ? ml : 0u;
    const uint64_t nb = __ballot(near);
    if (near) {
      list[nnear + lane_rank(nb)] = (uint16_t)r;
      n = s < R0 ? R0 - s : 0u;
    }
    nnear += (uint32_t)__builtin_popcountll(nb);
    for (uint32_t j = 0; j < n; j += 8) lds_put(stage + qs + j, out_ld_guard(g, s + j, n - j, cap), n - j < 8 ? n - j : 8);
  }
  *nnear_out = nnear;
}

// staging -> staging LZ77 copy of n bytes with forward-byte semantics (dst - src = d; overlap allowed), by the whole
// wave: one long or self-overlapping match at a time (dd, ss, n, d are wave-uniform).  Every destination byte j is
// stage[ss + j mod d], and the first period [ss, ss + min(d, n)) is final when the match is ready, so the lanes are
// independent of each other: 8 bytes per lane without overlap, one whole number of periods per lane for d < 8, 4
// gathered bytes per lane otherwise.
__device__ __forceinline__ void wave_copy(lds_u8 *stage, uint32_t dd, uint32_t ss, uint32_t n, uint32_t d, uint32_t lane) {
  if (d >= n) {
    const uint32_t j = lane * 8;
    if (j < n) lds_put(stage + dd + j, lds_ld64(stage + ss + j), n - j < 8 ? n - j : 8);
  } else if (d < 8) {
    uint64_t v = lds_ld64(stage + ss);  // the same 8 bytes for every lane
    const uint32_t sh = 8 * d;
    v &= (1ull << sh) - 1;
    v |= v << sh;
    if (2 * sh < 64) v |= v << (2 * sh);
    if (4 * sh < 64) v |= v << (4 * sh);
    const uint32_t adv = (0x76586880u >> (4 * d)) & 15;  // d * (8 / d): 5..8 bytes, 64 lanes cover 258
    const uint32_t j = lane * adv;
    if (j < n) lds_put(stage + dd + j, v, n - j < adv ? n - j : adv);
  } else {
    const float inv = 1.0f / (float)d;
    for (uint32_t j = lane * 4; j < n; j += 4 * kWave) {
      uint32_t r = j - (uint32_t)((float)j * inv) * d;  // j mod d, the quotient may be one off either way
      r = (int32_t)r < 0 ? r + d : r;
      r = r >= d ? r - d : r;
      uint32_t w = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t x = r + k >= d ? r + k - d : r + k;  // d >= 8: one wrap at most
        w |= (uint32_t)stage[ss + x] << (8 * k);
      }
      lds_put(stage + dd + j, w, n - j < 4 ? n - j : 4);
    }
  }
}

// Near matches: the source reaches into this round's staging buffer.  copy_far listed them in stream order; they
// are taken 64 at a time, one per lane, whatever lane decoded them.  Everything a record of a group can depend on is
// an earlier group (done) or the group itself.  Inside a group the records' destinations are disjoint and ascending
// with the lane, so the records whose output a lane's source range [sa, sb) touches are a RANGE of lower lanes: found
// once per group by two binary searches over the lanes (ds_bpermute), kept as a 64-bit mask.  A record is copied in the
// pass in which all of them are done (a wave-uniform mask of finished lanes); the lowest lane left never waits, so every
// pass makes progress.  (Until round 5 this was a bitmap of pending staging bytes in LDS: two atomics to set, two loads
// to test and two atomics to clear per record and pass.)
template <class PF>
__device__ __forceinline__ void copy_near_all(const lds_u32 *mrec, const lds_u16 *mpos, const lds_u16 *list,
                                              const Sink &sk, uint32_t lane, uint32_t count, bool *stuck, PF &pf) {
  lds_u8 *stage = sk.stage;
  const uint32_t head = sk.pos - sk.sbase();  // staging index of the round start
  for (uint32_t c0 = 0; c0 < count; c0 += kWave) {
    bool todo = c0 + lane < count;
    const uint32_t e = todo ? list[c0 + lane] : 0u;
    const uint32_t tk = mrec[e], qs = mpos[e];
    const uint32_t d = (tk & 0x7fff) + 1, ml = ((tk >> 16) & 0xff) + 3;
    // source bytes: staging [qs - d, min(qs - d + ml, qs)); the part before the round start came from HBM (copy_far)
    const int32_t s0 = (int32_t)qs - (int32_t)d;
    const uint32_t skip = s0 < (int32_t)head ? head - (uint32_t)s0 : 0u;
    const uint32_t sa = (uint32_t)(s0 + (int32_t)skip), sb = d < ml ? qs : (uint32_t)s0 + ml;
    const uint32_t n = ml - skip;         // skip < ml: a near match reaches into the round
    const bool fast = d >= n && n <= 32;  // the usual case: source and destination do not overlap
    // what the lanes of this group still have to produce: [qb, qe), ascending with the lane (idle lanes: nothing, on top)
    const uint32_t qb = todo ? qs + skip : 0xffffffffu, qe = todo ? qs + ml : 0xffffffffu;
    uint32_t jlo = 0, jhi = 0;  // lanes [jlo, jhi): qe > sa and qb < sb
#pragma unroll
    for (uint32_t st = 32; st > 0; st >>= 1) {
      const uint32_t ve = __shfl(qe, jlo + st - 1), vb = __shfl(qb, jhi + st - 1);
      jlo = ve <= sa ? jlo + st : jlo;
      jhi = vb < sb ? jhi + st : jhi;
    }
    jhi = jhi < lane ? jhi : lane;  // (lower lanes only: a record's own output is not its source)
    uint64_t need = 0;
    if (sa < sb && jlo < jhi) need = ((jhi >= 64 ? 0ull : (1ull << jhi)) - 1ull) & ~((1ull << jlo) - 1ull);
    uint64_t done = 0;  // wave-uniform: lanes of this group that are copied
    pf.tick_lds(P_NEAR_LOAD);
    for (uint32_t guard = 0; __ballot(todo) != 0; guard++) {
      if (guard > kWave) {  // cannot happen
        *stuck = true;
        return;
      }
      pf.count(C_NEAR_IT);
      const bool ready = todo && (need & ~done) == 0;
      if constexpr (PF::on) {
        if (__ballot(ready && !fast)) pf.count(C_LONG_NEAR);
      }
      pf.tick_lds(P_NEAR);
      if (ready && fast) {  // all loads in flight together
        lds_u8 *dd = stage + qs + skip;
        const lds_u8 *ss = stage + sa;
        const uint64_t v0 = lds_ld64(ss), v1 = lds_ld64(ss + 8), v2 = lds_ld64(ss + 16), v3 = lds_ld64(ss + 24);
        lds_put(dd, v0, n < 8 ? n : 8);
        if (n > 8) lds_put(dd + 8, v1, n < 16 ? n - 8 : 8);
        if (n > 16) {
          lds_put(dd + 16, v2, n < 24 ? n - 16 : 8);
          if (n > 24) lds_put(dd + 24, v3, n - 24);
        }
      }
      pf.tick_lds(P_NEAR_FAST);
      for (uint64_t lm = __ballot(ready && !fast); lm; lm &= lm - 1) {  // long or self-overlapping: the wave takes them one by one
        const uint32_t l = (uint32_t)__builtin_ctzll(lm);
        wave_copy(stage, rdlane(qs + skip, l), rdlane(sa, l), rdlane(n, l), rdlane(d, l), lane);
      }
      pf.tick_lds(P_NEAR_SLOW);
      done |= __ballot(ready);
      todo = todo && !ready;
      pf.tick_lds(P_NEAR_UPD);
    }
  }
  pf.tick(P_NEAR);
}

// The second half of a round: far matches from HBM, near matches inside the staging buffer, then the round goes out.
template <class PF>
__device__ __forceinline__ void copy_round(lds_smem *sm, Sink &sk, uint32_t lane, uint32_t total, uint32_t nrec, bool *stuck, PF &pf) {
  const lds_u32 *mrec = (const lds_u32 *)sm->mrec;
  const lds_u16 *mpos = (const lds_u16 *)sm->mpos;
  lds_u16 *list = (lds_u16 *)sm->list;
  const uint32_t R0 = sk.pos, rb = sk.sbase();
  uint32_t nnear = 0;
#ifdef MD_DEBUG_KNOWN_BOUNDS
  {
    const int kbm = __builtin_amdgcn_readfirstlane(g_kb_mode);
    if (kbm & 8) {
      sk.pos += total;
      return;
    }
    if (kbm & 4) {
      sk.flush(total);
      return;
    }
  }
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // earlier rounds' flushes have landed
  if (R0 + 40 > sk.cap) {
    copy_far_guarded<PF>(mrec, mpos, list, sk.stage, sk.g, R0, rb, sk.cap, lane, nrec, &nnear);
    pf.tick(P_FAR);
  } else {
    copy_far(mrec, mpos, list, sk, lane, nrec, &nnear, pf);
  }
  copy_near_all(mrec, mpos, list, sk, lane, nnear, stuck, pf);
  sk.flush(total);
  pf.tick(P_ADLER);
}

// ---- the decoder / copier hand-over -------------------------------------------------------------
__device__ __forceinline__ uint32_t mail_ld(const MD_LDS uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mail_st(MD_LDS uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// decoder: until the copier has finished everything posted so far (false = the copier gave up)
__device__ __forceinline__ bool mail_wait_idle(lds_smem *sm, uint32_t sent) {
  while (uni(mail_ld(&sm->mail.copied)) != sent) __builtin_amdgcn_s_sleep(2);  // (a poll costs issue slots the other wavefronts want)
  asm volatile("" ::: "memory");
  return uni(mail_ld(&sm->mail.stuck)) == 0;
}
// decoder: everything the job needs is in LDS; hand it over
__device__ __forceinline__ void mail_post(lds_smem *sm, uint32_t lane, uint32_t kind, uint32_t total, uint32_t x, uint32_t &sent) {
  if (lane == 0) {
    sm->mail.kind = kind;
    sm->mail.total = total;
    sm->mail.x = x;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  sent++;
  if (lane == 0) mail_st(&sm->mail.emitted, sent);
}

// ---------------------------------------------------------------------------
// All rounds of one Huffman block.  bp is a bit position of the body; on return *bp_io is the bit after the EOB.
// PAIR: this wavefront only decodes — a round's records and literals go to the copier wavefront (copier_main), which
// copies the matches, folds the checksum and writes the round out while this one is already walking the next round's
// zones; sk is then the decoder's own idea of the output position, and `sent` counts the jobs posted.
// BUDGET: the block has an incomplete code (see sync_pass); that form is not made for speed.
template <class PF, bool PAIR, bool BUDGET>
__device__ __forceinline__ int inflate_block(lds_smem *sm, const uint8_t *__restrict__ body, uint32_t body_len, Sink &sk,
                                             uint32_t lroot, uint32_t lane, uint32_t *bp_io, uint32_t *zone_io, Window &wnd,
                                             uint32_t &sent, PF &pf) {
  uint32_t bp = *bp_io;
  lds_u32 *win = (lds_u32 *)sm->win;
  const lds_u32 *lut = (const lds_u32 *)sm->lut;
  lds_u32 *mrec = (lds_u32 *)sm->mrec;
  lds_u16 *mpos = (lds_u16 *)sm->mpos;
  const uint32_t total_bits = body_len * 8;
  uint32_t zs = *zone_io;  // zone size of this round (wave-uniform), adapted to the expansion of the last one
  constexpr bool kSplit = PAIR && !BUDGET && MD_EMIT_SPLIT != 0;
  for (;;) {
    const uint32_t base = (bp >> 5) << 2;  // window start (byte of the body, multiple of 4)
    wnd.ensure(win, body, body_len, base, lane);
    const uint32_t rbp = bp - base * 8, tot = total_bits - base * 8;  // window-relative
    pf.tick(P_ENSURE);
    pf.count(C_ROUNDS);
    pf.count(C_PASSES);
    uint32_t start = rbp + lane * zs, end = 0, stop = 0, nb = 0;
    uint32_t midp = 0xffffffffu, midcnt = 0;  // (kSplit) where the zone is cut in two, and what its first half produces
    const uint32_t limit = rbp + (lane + 1) * zs;
    // the speculative pass starts a RUN-IN ahead of the zone (inside the zone before): all it has to deliver is the
    // boundary at the zone's END, and the longer the walk, the likelier that it has synchronised by then - fewer lanes
    // walk again in passes 2+, which cost a full pass each for a handful of lanes (the pass without counts is the
    // cheap one: 24 instructions a step against 34)
    const uint32_t runin_ = (zs * RUNIN_NUM) / RUNIN_DEN, runin = runin_ < lane * zs ? runin_ : lane * zs;  // (not before the round's start)
    bool counted = false;
#ifdef MD_DEBUG_KNOWN_BOUNDS
    const int kbm = __builtin_amdgcn_readfirstlane(g_kb_mode);
    const bool kb_can = wnd.kb_round < KB_ROUNDS;
    uint4 *kbp = g_kb_buf + ((size_t)wnd.kb_sid * KB_ROUNDS + wnd.kb_round) * kWave + lane;
    wnd.kb_round++;
    if ((kbm & 3) == 2 && kb_can) {
      const uint4 v = *kbp;
      start = v.x;
      end = v.y;
      stop = v.z & 0xffffu;
      counted = (v.z >> 16) != 0;
      nb = v.w;
    } else {
#endif
    sync_pass<false, BUDGET>(win, lut, lroot, true, start - runin, limit, end, stop, nb);
    pf.tick(P_DECODE1);
    const uint32_t passes = zs * PASSES >= PASS_BITS ? PASSES : PASS_BITS / zs > PASSES_MAX ? PASSES_MAX : PASS_BITS / zs;
    for (uint32_t it = 0; it < passes; it++) {
      const uint32_t pe = wave_shr1(end), ps = wave_shr1(stop);
      const bool redo = lane == 0 ? !counted : (ps == 0 && (pe != start || !counted));
      if (__ballot(redo) == 0) break;
      if (redo && lane > 0) start = pe;
      pf.count(C_PASSES);
      if constexpr (kSplit) sync_pass<true, BUDGET, true>(win, lut, lroot, redo, start, limit, end, stop, nb, limit - (zs * MD_SPLIT_B) / 8, &midp, &midcnt);
      else sync_pass<true, BUDGET>(win, lut, lroot, redo, start, limit, end, stop, nb);
      counted = counted || redo;
    }
#ifdef MD_DEBUG_KNOWN_BOUNDS
      if ((kbm & 3) == 1 && kb_can) *kbp = make_uint4(start, end, stop | (counted ? 0x10000u : 0u), nb);
    }
#endif
    pf.tick(P_DECODE2);
    uint32_t nvalid;
    {
      const uint32_t pe = wave_shr1(end), ps = wave_shr1(stop);
      const uint64_t bad = __ballot(!counted || (lane > 0 && (ps != 0 || pe != start)));
      nvalid = bad ? (uint32_t)__builtin_ctzll(bad) : 64;  // lane 0 is always counted: nvalid >= 1
    }
    const uint32_t R0 = sk.pos, rb = sk.sbase();
    uint32_t mynb = lane < nvalid ? nb : 0;  // bytes | matches << 20
    const uint32_t off = wave_excl_scan(mynb, lane);
    if (nvalid < 64) pf.count(C_END_CHAIN);
    {  // staging and record capacity: keep the largest prefix of lanes that fits (a lone first lane truncates itself;
       // its matches always fit)
      const uint32_t bend = ((off + mynb) & (kCountMatch - 1)), rend = (off + mynb) >> 20;
      const uint64_t fits = __ballot((R0 - rb) + bend <= STAGE - 16 && rend <= RMAX);
      const uint32_t nfit = fits == ~0ull ? 64 : (uint32_t)__builtin_ctzll(~fits);
      if (nfit < nvalid) {
        pf.count(rdlane(rend, nfit) > RMAX ? C_END_RECORDS : C_END_FIT);
        nvalid = nfit ? nfit : 1;
      }
    }
    const bool mine = lane < nvalid;
    if (!mine) mynb = 0;
    const uint32_t boff = off & (kCountMatch - 1), roff = off >> 20;
    const uint32_t q0 = R0 + boff;
    LaneOut lo;
    lo.endp = end;
    lo.stopc = 0;
    lo.bytes = 0;
    lo.nm = 0;
    if constexpr (PAIR) {  // the staging buffer and the record arrays are the copier's until it has written the last round out
      pf.tick(P_DECODE2);
      if (!mail_wait_idle(sm, sent)) return MD_E_HIP;
      pf.tick(P_WAIT_DEC);
    }
    // Two wavefronts, one round: the copier has nothing to do until the round is emitted, so it emits the second half of
    // every zone while this wavefront emits the first (the counting pass left the boundary in the middle and what lies
    // before it: the halves' output positions and records are known) - the emit pass is half as long.  Small zones are
    // not worth the hand-over.
    bool split = false;
    uint32_t bq = 0, br = 0;  // the second half's bytes | records before it, relative to the lane's
    LaneOut lob;
    lob.endp = 0, lob.stopc = 0, lob.bytes = 0, lob.nm = 0;
    bool gob = false;
    if constexpr (kSplit) split = zs >= 128;
    {
      const uint32_t all = rdlane(off + mynb, nvalid - 1) & (kCountMatch - 1);  // bytes the round will produce
      const bool plain = tot >= rbp + kWave * zs + 64 && R0 >= 32768u && all <= sk.cap - R0 && (R0 - rb) + all <= STAGE - 16;
      uint32_t alimit = limit;
      if constexpr (kSplit) {
        if (split) {
          gob = mine && midp != 0xffffffffu && midp < limit && midp >= start;
          bq = midcnt & (kCountMatch - 1);
          br = midcnt >> 20;
          lds_u32 *hx = reinterpret_cast<lds_u32 *>(sm->list);
          hx[4 * lane + 0] = gob ? midp : 0xffffffffu;
          hx[4 * lane + 1] = limit;
          hx[4 * lane + 2] = q0 + bq;
          hx[4 * lane + 3] = roff + br;
          if (lane < 6) hx[kHxUni + lane] = lane == 0 ? lroot : lane == 1 ? tot : lane == 2 ? rb : lane == 3 ? R0 : lane == 4 ? sk.cap : (plain ? 0u : 1u);
          mail_post(sm, lane, kJobEmit, 0, 0, sent);
          alimit = gob ? midp : limit;
        }
      }
      if (plain && !BUDGET) emit_pass<false, false>(win, lut, lroot, mrec, mpos, sk.stage, roff, tot, mine, start, alimit, q0, rb, R0, sk.cap, lo);
      else emit_pass<true, BUDGET>(win, lut, lroot, mrec, mpos, sk.stage, roff, tot, mine, start, alimit, q0, rb, R0, sk.cap, lo);
      if constexpr (kSplit) {
        if (split) {
          if (!mail_wait_idle(sm, sent)) return MD_E_HIP;
          const lds_u32 *hx = reinterpret_cast<const lds_u32 *>(sm->list);
          if (gob) {
            lob.endp = hx[4 * lane + 0];
            lob.stopc = hx[4 * lane + 1];
            lob.bytes = hx[4 * lane + 2];
            lob.nm = hx[4 * lane + 3];
          }
          // a lane's verdict: its first half's if that stopped, else its second half's
          const bool sa = lo.stopc != 0;
          if (!sa && gob) {
            lo.endp = lob.endp;
            lo.stopc = lob.stopc;
            lo.bytes = bq + lob.bytes;
            lo.nm = br + lob.nm;
          }
        }
      }
    }
    pf.tick(P_EMIT_A);
    // the first stopped lane (stream order) ends the round
    uint32_t total, lstop, nrec;
    {
      const uint64_t fm = __ballot(mine && lo.stopc != 0);
      if (fm) {
        const uint32_t fl = __builtin_ctzll(fm);
        lstop = rdlane(lo.stopc, fl);
        total = rdlane(boff, fl) + rdlane(lo.bytes, fl);
        nrec = rdlane(roff, fl) + rdlane(lo.nm, fl);  // later lanes are void
        if (lstop == kStEob) pf.count(C_END_EOB);
        else if (lstop == kStTrunc) pf.count(C_END_STAGE);
        nvalid = fl + 1;
      } else {
        lstop = 0;
        const uint32_t all = rdlane(off + mynb, nvalid - 1);
        total = all & (kCountMatch - 1);
        nrec = all >> 20;
      }
    }
    const uint32_t nbp = base * 8 + rdlane(lo.endp, nvalid - 1);
    wnd.fetch(body, body_len, (nbp >> 5) << 2, lane);  // the next round's (or the next block header's) window: on its way
    bool stuck = false;
    if constexpr (PAIR) {
      mail_post(sm, lane, kJobRound, total, nrec, sent);
      sk.pos += total;
    } else {
      copy_round(sm, sk, lane, total, nrec, &stuck, pf);
    }
    pf.count(C_LANES, nvalid);
    if (stuck || (nbp == bp && total == 0 && (lstop == 0 || lstop == kStTrunc))) return MD_E_HIP;  // defensive: no progress
    {  // next round: 64 zones that fill about 7/8 of the staging buffer at this round's bytes-per-bit
      const uint32_t bits = nbp - bp;
      const uint64_t want = (uint64_t)(STAGE - STAGE / 8) * bits / ((uint64_t)kWave * (total ? total : 1u));
      zs = want > S ? S : want < SMIN ? SMIN : (uint32_t)want;
    }
    bp = nbp;
    if (lstop == kStEob) break;
    if (lstop != 0 && lstop != kStTrunc) return (int)lstop;
  }
  *bp_io = bp;
  *zone_io = zs;
  return MD_OK;
}

// stored bytes body[p, p + len) -> staging buffer -> output (lib/de.ml:1613-1627), a staging buffer at a time
__device__ __forceinline__ void copy_stored(Sink &sk, const uint8_t *__restrict__ body, uint32_t p, uint32_t len, uint32_t lane) {
  const uint8_t *q = body + p;
  uint32_t left = len;
  while (left) {
    const uint32_t seg = left < STAGE - 16 ? left : STAGE - 16;
    const uint32_t s0 = sk.pos - sk.sbase();
    for (uint32_t j = lane; j < seg; j += kWave) sk.stage[s0 + j] = q[j];
    sk.flush(seg);
    q += seg;
    left -= seg;
  }
}

// The copier wavefront of a stream: takes the decoder's jobs one by one until it is told to quit.
template <class PF>
__device__ __forceinline__ void copier_main(lds_smem *sm, const uint8_t *__restrict__ body, Sink &sk, uint32_t lane, PF &pf) {
  uint32_t done = 0;
  for (;;) {
    while (uni(mail_ld(&sm->mail.emitted)) == done) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
    const uint32_t kind = uni(sm->mail.kind), total = uni(sm->mail.total), x = uni(sm->mail.x);
    pf.tick(P_WAIT_COPY);
    if (kind == kJobQuit) break;
    bool stuck = false;
    if (kind == kJobRound) copy_round(sm, sk, lane, total, x, &stuck, pf);
    else if (kind == kJobEmit) {
      if constexpr (MD_EMIT_SPLIT != 0) {
        lds_u32 *hx = reinterpret_cast<lds_u32 *>(sm->list);
        const uint32_t bstart = hx[4 * lane + 0], blimit = hx[4 * lane + 1], bq0 = hx[4 * lane + 2], brec = hx[4 * lane + 3];
        const uint32_t lroot = uni(hx[kHxUni + 0]), tot = uni(hx[kHxUni + 1]), rb = uni(hx[kHxUni + 2]), R0 = uni(hx[kHxUni + 3]),
                       cap = uni(hx[kHxUni + 4]), checked = uni(hx[kHxUni + 5]);
        const bool go = bstart != 0xffffffffu;
        LaneOut lo;
        lo.endp = bstart, lo.stopc = 0, lo.bytes = 0, lo.nm = 0;
        const lds_u32 *win = (const lds_u32 *)sm->win;
        const lds_u32 *lut = (const lds_u32 *)sm->lut;
        if (checked) emit_pass<true, false>(win, lut, lroot, (lds_u32 *)sm->mrec, (lds_u16 *)sm->mpos, sk.stage, brec, tot, go, go ? bstart : 0u, blimit, bq0, rb, R0, cap, lo);
        else emit_pass<false, false>(win, lut, lroot, (lds_u32 *)sm->mrec, (lds_u16 *)sm->mpos, sk.stage, brec, tot, go, go ? bstart : 0u, blimit, bq0, rb, R0, cap, lo);
        hx[4 * lane + 0] = lo.endp;
        hx[4 * lane + 1] = lo.stopc;
        hx[4 * lane + 2] = lo.bytes;
        hx[4 * lane + 3] = lo.nm;
      }
    } else copy_stored(sk, body, x, total, lane);
    if (lane == 0) {
      sm->mail.a = sk.a;
      sm->mail.b = sk.b;
      if (stuck) sm->mail.stuck = 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    done++;
    if (lane == 0) mail_st(&sm->mail.copied, done);
  }
}

// A stream decoded in pieces (md_de_inf_continue_host: the `Flush steps of De.Inf.decode while input is still arriving,
// lib/de.ml:1427-1474): the piece starts start_bit bits into its first byte, the output buffer begins with hist_len
// bytes of what was decoded before (the window), the checksum goes on from adler_in; the kernel says where the last
// block that was complete in this piece ended (bit position, output position, checksum state there) so that the next
// piece can start at that block boundary.  All pointers null: a whole stream, nothing to report (the batch path).
struct Cont {
  const uint32_t *start_bit, *hist_len, *adler_in;
  uint64_t *resume_bits, *resume_out;
  uint32_t *resume_adler, *resume_last;
};

// PAIR = two wavefronts per stream (decoder + copier, see inflate_block); otherwise one wavefront does both in turn.
template <bool PROF, bool PAIR>
__global__ __launch_bounds__(PAIR ? 2 * kWave : kWave, PAIR ? 4 : 2) void inflate_wave_kernel(
    int format, uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *out, const uint64_t *__restrict__ out_off,
    const uint64_t *__restrict__ out_cap, uint64_t *__restrict__ out_len,
    uint64_t *__restrict__ consumed, int32_t *__restrict__ status, uint32_t *__restrict__ checksum,
    uint64_t *__restrict__ dbg, const uint32_t *__restrict__ order, Cont cont) {
  __shared__ Smem smem;  // static: LDS addresses fold into the instructions' offset fields
  lds_smem *sm = (lds_smem *)&smem;
  Prof<PROF> pf;
  pf.init();
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = PAIR ? uni(threadIdx.x / kWave) : 0u;  // 0 = decoder, 1 = copier
  if (blockIdx.x >= n) return;
  const uint32_t sid = order ? order[blockIdx.x] : blockIdx.x;  // workgroups start in index order: longest streams first

  const uint8_t *src = in + in_off[sid];
  uint64_t slen64 = in_len[sid], cap64 = out_cap[sid];
  if (slen64 > MD_MAX_INFLATE_IN) {  // 32-bit bit positions: the descriptor is out of range (mdeflate.h)
    if (threadIdx.x == 0) {
      out_len[sid] = 0;
      consumed[sid] = 0;
      status[sid] = MD_E_INVALID_ARGUMENT;
      if (checksum) checksum[sid] = 0;
    }
    return;
  }
  const uint32_t slen = (uint32_t)slen64;
  const bool cap_clamped = cap64 > MD_MAX_STREAM;  // more room than 32-bit cursors can use
  const uint32_t cap = cap_clamped ? (uint32_t)MD_MAX_STREAM : (uint32_t)cap64;

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
  const uint8_t *body = src + body_off;

  Sink sk;  // PAIR: the copier's is the real one, the decoder's only follows the output position
  sk.stage = (lds_u8 *)sm->stage;
  sk.g = out + out_off[sid];
  sk.cap = cap;
  const uint32_t hist = cont.hist_len ? cont.hist_len[sid] : 0u, adler0 = cont.adler_in ? cont.adler_in[sid] : 1u;
  sk.pos = hist < cap ? hist : cap;  // (the history is the caller's: it fits the buffer)
  sk.lane = lane;
  sk.a = adler0 & 0xffffu;
  sk.b = adler0 >> 16;
  sk.want_adler = (checksum != nullptr) || format == MD_FORMAT_ZLIB || cont.resume_adler != nullptr;

  if (threadIdx.x < 4) sm->lut[kStopEobI + threadIdx.x] = mk_entry(0, 0, 0, 0, kStopEobI + (threadIdx.x & 2));  // the self-looping STOP entries
  if constexpr (PAIR) {
    if (threadIdx.x < sizeof(Mail) / 4) ((lds_u32 *)&sm->mail)[threadIdx.x] = threadIdx.x == 6 ? sk.a : threadIdx.x == 7 ? sk.b : 0u;
    __syncthreads();
    if (wave == 1) {
      copier_main(sm, body, sk, lane, pf);
      if constexpr (PROF) {
        if (lane == 0 && sid == 0 && dbg) {
          const int mine[] = {P_FAR, P_NEAR, P_ADLER, P_NEAR_FAST, P_NEAR_SLOW, P_NEAR_UPD, P_NEAR_LOAD, P_FAR_REC, P_FAR_LOAD, P_WAIT_COPY};
          for (int i : mine) dbg[i] = pf.acc[i];
          dbg[P_COUNT + C_NEAR_IT] = pf.cnt[C_NEAR_IT];
          dbg[P_COUNT + C_LONG_NEAR] = pf.cnt[C_LONG_NEAR];
        }
      }
      return;
    }
  }

  lds_u32 *win = (lds_u32 *)sm->win;
  const uint32_t total_bits = body_len * 8;
  uint32_t bp = cont.start_bit ? cont.start_bit[sid] & 7u : 0u;
  uint32_t zone = S;
  uint32_t sent = 0;  // PAIR: jobs posted
  uint32_t fixed_lroot = 0;  // != 0: the tables in LDS are the fixed ones (and this is their root width)
  Window wnd;
  wnd.base = 0xffffffffu;
#ifdef MD_DEBUG_KNOWN_BOUNDS
  wnd.kb_round = 0;
  wnd.kb_sid = sid;
#endif
  if (cont.resume_bits && lane == 0) {  // no block complete yet
    cont.resume_bits[sid] = bp;
    cont.resume_out[sid] = sk.pos;
    cont.resume_adler[sid] = adler0;
    cont.resume_last[sid] = 0;
  }

  if (rc == MD_OK) {
    bool last = false;
    uint32_t lds_base = 0xffffffffu;  // the window in LDS begins at this byte of the body (nothing else has been loaded since)
    while (!last && rc == MD_OK) {
      // a block header that still lies in the first part of the window in LDS is read from there: runs of tiny blocks
      // (flush points, empty blocks) would otherwise load 2 KiB from the input for every few bytes of it
      uint32_t base = (bp >> 5) << 2;
      if (lds_base != 0xffffffffu && bp >= lds_base * 8 && bp - lds_base * 8 <= 4096) base = lds_base;
      else {
        wnd.ensure(win, body, body_len, base, lane);
        lds_base = base;
      }
      uint32_t rbp = bp - base * 8;
      const uint32_t tot = total_bits - base * 8;
      if ((int32_t)(tot - rbp) < 3) {
        rc = MD_UNEXPECTED_END_OF_INPUT;
        break;
      }
      const uint32_t hdr = uni(peek(win, rbp));
      last = hdr & 1;
      uint32_t type = (hdr >> 1) & 3;
      bp += 3;
      rbp += 3;
      bool block_done = false;  // the block was dealt with here (an empty fixed one)
      if (type == 1) {
        // a run of EMPTY fixed blocks - header, then the 7-bit end-of-block code 0000000 (lib/de.ml:821-833) - is walked
        // through right here: ten bits a block instead of a round each
        for (;;) {
          if (rbp + 7 > tot || rbp + 64 > WIN_WORDS * 32 - 64 || (uni(peek(win, rbp)) & 0x7fu) != 0) break;  // not one of them
          rbp += 7;
          bp += 7;
          block_done = true;  // the block whose header was consumed last is complete
          if (last || rbp + 3 > tot) break;
          const uint32_t h = uni(peek(win, rbp)) & 7u;
          if (((h >> 1) & 3) != 1) break;  // another kind of block: back to the loop above
          last = h & 1;
          rbp += 3;
          bp += 3;
          block_done = false;
        }
      }
      if (block_done) {
        // nothing to decode
      } else if (type == 0) {
        // flat, lib/de.ml:1613-1627
        uint32_t p = (bp + 7) >> 3;
        if (body_len - p < 4) {
          rc = MD_UNEXPECTED_END_OF_INPUT;
          break;
        }
        uint32_t h4;
        if (p + 4 <= base + (WIN_WORDS - 2) * 4) h4 = uni(peek(win, (p - base) * 8));
        else h4 = (uint32_t)body[p] | ((uint32_t)body[p + 1] << 8) | ((uint32_t)body[p + 2] << 16) | ((uint32_t)body[p + 3] << 24);
        uint32_t len = h4 & 0xffff, nlen = h4 >> 16;
        p += 4;
        if (nlen != 0xffff - len) rc = MD_INVALID_COMPLEMENT_OF_LENGTH;
        else if (len > body_len - p) rc = MD_UNEXPECTED_END_OF_INPUT;
        else if (len > sk.cap - sk.pos) rc = MD_UNEXPECTED_END_OF_OUTPUT;
        else {
          if (len) {  // (an empty stored block - a flush point - is its header and nothing else)
            if constexpr (PAIR) {
              if (!mail_wait_idle(sm, sent)) rc = MD_E_HIP;
              mail_post(sm, lane, kJobStored, len, p, sent);
              sk.pos += len;
            } else {
              copy_stored(sk, body, p, len, lane);
            }
          }
          p += len;
          bp = p * 8;
        }
      } else if (type == 3) {
        rc = MD_INVALID_KIND_OF_BLOCK;
      } else {
        uint32_t lroot = 0;
        if (type == 1) {
          if (fixed_lroot == 0) {  // (a run of fixed blocks builds their tables once)
            fixed_tables(sm, lane, &lroot);
            fixed_lroot = uni(lroot);
          }
          lroot = fixed_lroot;
        } else {
          uint32_t hend = 0;
          fixed_lroot = 0;
          rc = dynamic_tables(sm, rbp, tot, lane, &hend, &lroot, pf);
          bp = base * 8 + hend;
        }
        lroot = uni(lroot);
        pf.tick(P_HEADER);
        if (rc == MD_OK) {
          if (lroot & kLoopy) rc = inflate_block<Prof<PROF>, PAIR, true>(sm, body, body_len, sk, lroot & 15, lane, &bp, &zone, wnd, sent, pf);
          else rc = inflate_block<Prof<PROF>, PAIR, false>(sm, body, body_len, sk, lroot, lane, &bp, &zone, wnd, sent, pf);
          lds_base = 0xffffffffu;
        }
      }
      if (cont.resume_bits && rc == MD_OK) {  // a block is complete: the next piece of the stream could start here
        if constexpr (PAIR) {                 // (the checksum state is the copier's)
          if (!mail_wait_idle(sm, sent)) rc = MD_E_HIP;
          sk.a = uni(sm->mail.a);
          sk.b = uni(sm->mail.b);
        }
        if (lane == 0) {
          cont.resume_bits[sid] = bp;
          cont.resume_out[sid] = sk.pos;
          cont.resume_adler[sid] = (sk.b << 16) | sk.a;
          cont.resume_last[sid] = last ? 1u : 0u;
        }
      }
    }
  }
  if constexpr (PAIR) {  // everything posted is written out; the checksum state comes back, the copier goes home
    if (!mail_wait_idle(sm, sent) && rc == MD_OK) rc = MD_E_HIP;
    sk.a = uni(sm->mail.a);
    sk.b = uni(sm->mail.b);
    mail_post(sm, lane, kJobQuit, 0, 0, sent);
  }
  uint32_t used = (bp + 7) >> 3;  // i_pos - (bits lsr 3), lib/de.ml:1805
  uint32_t adler = (sk.b << 16) | sk.a;
  if (rc == MD_OK && format == MD_FORMAT_ZLIB) {
    const uint8_t *t = src + 2 + used;
    uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    if (want != adler) rc = MD_INVALID_CHECKSUM;
    used += 6;
  }
  if (rc == MD_UNEXPECTED_END_OF_OUTPUT && cap_clamped) rc = MD_E_INVALID_ARGUMENT;  // a > 4 GiB stream
  if (lane == 0) {
    out_len[sid] = sk.pos;
    consumed[sid] = rc == MD_OK ? used : 0;
    status[sid] = rc;
    if (checksum) checksum[sid] = adler;
  }
  if constexpr (PROF) {
    if (lane == 0 && sid == 0 && dbg) {
      if constexpr (PAIR) {
        const int mine[] = {P_ENSURE, P_DECODE1, P_DECODE2, P_EMIT_A, P_HEADER, P_HDR_LENS, P_HDR_LIT, P_WAIT_DEC};
        for (int i : mine) dbg[i] = pf.acc[i];
        for (int i = 0; i < C_COUNT; i++)
          if (i != C_NEAR_IT && i != C_LONG_NEAR) dbg[P_COUNT + i] = pf.cnt[i];
      } else {
        for (int i = 0; i < P_COUNT; i++) dbg[i] = pf.acc[i];
        for (int i = 0; i < C_COUNT; i++) dbg[P_COUNT + i] = pf.cnt[i];
      }
    }
  }
}

// Launch order of a batch: a stream's time grows with its compressed length, and a batch of more streams than the
// chip holds at once (8 per CU) ends when its last wavefront does.  Workgroups are started in index order, so the
// streams are handed out longest first (256 length classes, counting sort; the order inside a class is whatever
// the atomics give — nothing a stream computes depends on it).
constexpr uint32_t kOrderThreads = 1024, kOrderClasses = 256;
__global__ __launch_bounds__(kOrderThreads) void inflate_order_kernel(uint32_t n, const uint64_t *__restrict__ in_len,
                                                                      uint32_t *__restrict__ order) {
  __shared__ uint32_t cls[kOrderClasses];
  __shared__ uint32_t longest;
  const uint32_t t = threadIdx.x;
  if (t < kOrderClasses) cls[t] = 0;
  if (t == 0) longest = 0;
  __syncthreads();
  uint32_t m = 0;
  for (uint32_t i = t; i < n; i += kOrderThreads) {
    const uint64_t l = in_len[i];
    const uint32_t l32 = l > MD_MAX_INFLATE_IN ? 0u : (uint32_t)l;  // rejected at once by the kernel
    m = l32 > m ? l32 : m;
  }
  atomicMax(&longest, m);
  __syncthreads();
  const uint64_t span = (uint64_t)longest + 1;
  auto klass = [&](uint32_t i) {
    const uint64_t l = in_len[i];
    const uint32_t l32 = l > MD_MAX_INFLATE_IN ? 0u : (uint32_t)l;
    return (kOrderClasses - 1) - (uint32_t)((uint64_t)l32 * kOrderClasses / span);  // class 0 = the longest
  };
  for (uint32_t i = t; i < n; i += kOrderThreads) atomicAdd(&cls[klass(i)], 1u);
  __syncthreads();
  if (t == 0) {  // exclusive scan of 256 counters
    uint32_t acc = 0;
    for (uint32_t c = 0; c < kOrderClasses; c++) {
      const uint32_t v = cls[c];
      cls[c] = acc;
      acc += v;
    }
  }
  __syncthreads();
  for (uint32_t i = t; i < n; i += kOrderThreads) order[atomicAdd(&cls[klass(i)], 1u)] = i;
}

}  // namespace wv
}  // namespace md

// the launch order of a batch (longest first) on its own: the deflate kernels take it too
extern "C" int md_launch_stream_order(uint32_t n, const uint64_t *in_len, uint32_t *order, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::wv::inflate_order_kernel, dim3(1), dim3(md::wv::kOrderThreads), 0, stream, n, in_len, order);
  return (int)hipGetLastError();
}

// measurement only (md_set_option "debug_inflate_lds_pad", tools/dbg/inflate_occupancy.py): bytes of unused dynamic LDS
// per workgroup, which lowers the number of streams a CU holds at once
static uint32_t g_debug_lds_pad = 0;
extern "C" int md_i_debug_inflate_lds_pad(uint32_t bytes) {
  using namespace md::wv;
  if (bytes + sizeof(Smem) > 160u * 1024u) return 1;
  hipError_t e = hipFuncSetAttribute((const void *)inflate_wave_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)inflate_wave_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  g_debug_lds_pad = bytes;
  return 0;
}

extern "C" int md_i_debug_known_bounds(int mode, uint32_t nstreams) {
#ifdef MD_DEBUG_KNOWN_BOUNDS
  static uint4 *buf = nullptr;
  static uint32_t cap = 0;
  if (nstreams > cap) {
    if (buf) hipFree(buf);
    if (hipMalloc((void **)&buf, (size_t)nstreams * md::wv::KB_ROUNDS * 64 * sizeof(uint4)) != hipSuccess) return 2;
    cap = nstreams;
    if (hipMemcpyToSymbol(HIP_SYMBOL(md::wv::g_kb_buf), &buf, sizeof buf) != hipSuccess) return 3;
  }
  return hipMemcpyToSymbol(HIP_SYMBOL(md::wv::g_kb_mode), &mode, sizeof mode) == hipSuccess ? 0 : 4;
#else
  (void)mode, (void)nstreams;
  return 1;  // not a measurement build
#endif
}

// `order` = n words of device scratch, or null for index order; waves = wavefronts per stream (2, or 1)
extern "C" int md_launch_inflate_wave(int format, uint32_t n, const uint8_t *in, const uint64_t *in_off,
                                      const uint64_t *in_len, uint8_t *out, const uint64_t *out_off,
                                      const uint64_t *out_cap, uint64_t *out_len, uint64_t *consumed,
                                      int32_t *status, uint32_t *checksum, uint64_t *dbg, uint32_t *order,
                                      int waves, const void *cont_ptrs, hipStream_t stream) {
  if (n == 0) return 0;
  using namespace md::wv;
  const bool single = waves == 1;  // the one-wavefront form of the kernel: same results, kept for comparison
  dim3 grid(n), block(single ? kWave : 2 * kWave);
  if (order) hipLaunchKernelGGL(inflate_order_kernel, dim3(1), dim3(kOrderThreads), 0, stream, n, in_len, order);
  const uint32_t *ord = dbg ? nullptr : order;
  Cont cont{};  // (seven device pointers in the order of struct Cont, or null)
  if (cont_ptrs) memcpy(&cont, cont_ptrs, sizeof cont);
  const uint32_t lds_pad = g_debug_lds_pad;
#define MD_LAUNCH_INFLATE(P, Q)                                                                                          \
  hipLaunchKernelGGL((inflate_wave_kernel<P, Q>), grid, block, (P) ? 0u : lds_pad, stream, format, n, in, in_off, in_len, out, out_off, \
                     out_cap, out_len, consumed, status, checksum, dbg, ord, cont)
  if (dbg) {
    if (single) MD_LAUNCH_INFLATE(true, false);
    else MD_LAUNCH_INFLATE(true, true);
  } else {
    if (single) MD_LAUNCH_INFLATE(false, false);
    else MD_LAUNCH_INFLATE(false, true);
  }
#undef MD_LAUNCH_INFLATE
  return (int)hipGetLastError();
}
