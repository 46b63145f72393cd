// lzo_kernels.hip — batched LZO1X for gfx950: the reference's Lzo.uncompress / Lzo.compress
// (lib/lzo.ml), one independent stream per wavefront.
//
//   lzo_uncompress_kernel   The instruction stream of LZO is byte-serial, so its interpreter
//       (`run` over `fiber`, lib/lzo.ml:246-393) is wave-uniform: every lane follows the same
//       instruction, the opcode bytes come from a 256-byte window of the input held in registers
//       (one dword per lane, v_readlane to pick a byte).  The work of an instruction is
//       lane-parallel: a literal run is a coalesced copy input -> output, a match is a copy inside
//       the output where lane k takes byte k — source byte `o_pos - off + (k mod off)` is always
//       older than the instruction, whatever the overlap.
//   lzo_compress_kernel     lzo1x-1 (lib/lzo.ml:578-660): the probe loop is serial by construction
//       (what is inserted in the dictionary depends on where the previous match ended), so it runs
//       wave-uniformly with the 16 K-entry u16 dictionary of the current 48 KiB chunk in an HBM
//       workspace (in LDS it would cap residency at 5 wavefronts per CU); match extension
//       compares 8 bytes per lane (512 per step), literal runs and the trailer are coalesced
//       copies.  Within 32 bytes of a match - where text spends its time - the step is near_step:
//       8 probes from an input ring in LDS, 16 bytes at the reference, the match's last bytes
//       held back for the count of the literals behind it.
// Semantics, error cases and quirks are those of the reference as restated in oracle/lzo.c
// (the decoder state after a literal run behaves as 3, the EOI guard on every instruction, the
// `< 238` first-byte form, the match extension that stops 20 bytes before the end of a chunk).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "inflate_util.hpp"
#include "mdeflate.h"

namespace md {
namespace lzo {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint8_t ld_nt8(const uint8_t *p) { return __builtin_nontemporal_load(p); }
// Workgroup -> stream: PERSISTENT workgroups draw streams from a counter in device memory.  With one workgroup per
// stream a batch whose cheap and expensive streams alternate (C5: text, noise, text, ...) ended up with all its expensive
// streams on some of the CUs - a slot freed by a stream that ends at once is refilled at once, by the next workgroup in
// line: 8 192 mixed streams took as long as 8 192 text streams, twice their 4 096 text streams alone.
__device__ __forceinline__ uint32_t next_stream(uint32_t *counter, uint32_t lane) {
  uint32_t v = 0;
  if (lane == 0) v = atomicAdd(counter, 1u);
  return uni(v);
}

// 256-byte window of the input in registers: lane l holds bytes [wbase + 4l, wbase + 4l + 4)
struct Win {
  const uint8_t *src;
  uint32_t n, lane;
  uint32_t wbase;  // multiple of 4
  uint32_t w;
  __device__ __forceinline__ void fill(uint32_t pos) {
    wbase = pos & ~3u;
    const uint32_t a = wbase + 4 * lane;
    uint32_t v = 0;
    if (a + 4 <= n) __builtin_memcpy(&v, src + a, 4);
    else
      for (uint32_t k = 0; a + k < n; k++) v |= (uint32_t)src[a + k] << (8 * k);
    w = v;
  }
  // byte at pos < n (wave-uniform)
  __device__ __forceinline__ uint32_t byte(uint32_t pos) {
    if (pos - wbase >= 256u) fill(pos);
    const uint32_t d = pos - wbase;
    return ((uint32_t)__builtin_amdgcn_readlane((int)w, (int)(d >> 2)) >> (8 * (d & 3))) & 0xff;
  }
};

struct Dec {
  Win in;
  uint8_t *dst;
  uint32_t cap, i_pos, o_pos, lane;
  int state;
#ifdef MD_LZO_PROF
  uint32_t prof[8];
  uint64_t prof_t;
#endif
};

// transmit (lib/lzo.ml:188-192) with blit's bounds (:81-89).  Long runs (incompressible input is ONE literal run) go 16
// bytes per lane and step: unaligned loads, stores on the destination's alignment.
__device__ __forceinline__ int transmit(Dec &d, uint32_t len) {
  if (d.i_pos > d.in.n || len > d.in.n - d.i_pos || len > d.cap - d.o_pos) return MD_LZO_OUT_OF_BOUND;
  const uint8_t *s = d.in.src + d.i_pos;
  uint8_t *o = d.dst + d.o_pos;
  if (len < 256) {
    for (uint32_t k = d.lane; k < len; k += kWave) o[k] = s[k];
  } else {
    const uint32_t head = (16u - (uint32_t)((uintptr_t)o & 15)) & 15u;
    const uint32_t body = (len - head) & ~15u;
    if (d.lane < head) o[d.lane] = s[d.lane];
    for (uint32_t k = head + d.lane * 16; k < head + body; k += kWave * 16) {
      uint4 v;
      __builtin_memcpy(&v, s + k, 16);
      *reinterpret_cast<uint4 *>(o + k) = v;
    }
    const uint32_t k = head + body + d.lane;
    if (k < len) o[k] = s[k];
  }
  d.i_pos += len;
  d.o_pos += len;
  return MD_OK;
}
// copy (lib/lzo.ml:194-197): byte-serial LZ77 semantics, all bytes at once
__device__ __forceinline__ int copy(Dec &d, uint32_t off, uint32_t len) {
  if (off > d.o_pos || len > d.cap - d.o_pos) return MD_LZO_OUT_OF_BOUND;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // earlier instructions' bytes have landed
  const uint8_t *s = d.dst + (d.o_pos - off);
  for (uint32_t k = d.lane; k < len; k += kWave) d.dst[d.o_pos + k] = ld_nt8(s + (off >= len ? k : k % off));
  d.o_pos += len;
  return MD_OK;
}
// count (lib/lzo.ml:218-236)
__device__ __forceinline__ int count(Dec &d, uint32_t *out) {
  uint32_t idx = d.i_pos, res = 0;
  const uint32_t max = d.in.n;
  while (idx < max && d.in.byte(idx) == 0) {  // the reference's 4-byte then 1-byte scans count the same zeros
    idx++;
    res++;
  }
  if (idx < max) {
    d.i_pos = idx + 1;
    *out = res * 255 + d.in.byte(idx);
    return MD_OK;
  }
  return MD_LZO_INVALID_INPUT;
}

#define LZ_EOI() \
  if (d.i_pos >= d.in.n) return MD_UNEXPECTED_END_OF_INPUT;
#define LZ_TRY(e)             \
  {                           \
    const int rc_ = (e);      \
    if (rc_) return rc_;      \
  }

// ONE instruction of `fiber` (lib/lzo.ml:315-369) with every check in the reference's order, straight on the output
// buffer: the interpreter of rounds 2-5, now the slow path - what the batched decoder below does not take (an instruction
// near the end of the input, a length that goes on over zero bytes, the end marker, anything that fails) comes here, so
// the error cases and their order are the reference's by construction.  *end: the end-of-stream instruction.
__device__ __forceinline__ int fiber_step(Dec &d, bool *end) {
  LZ_EOI()
  uint32_t chr = d.in.byte(d.i_pos++);
  const int st = d.state & 3;  // -1 land 3 = 3
  uint32_t len, off, cnt;
  int nstate;
  if (chr < 16 && st == 0) {
    if (chr == 0) {
      LZ_EOI()
      LZ_TRY(count(d, &cnt))
      len = 3 + 15 + cnt;
    } else len = chr + 3;
    d.state = -1;
    LZ_EOI()
    LZ_TRY(transmit(d, len))
    return MD_OK;
  }
  if (chr < 16) {
    LZ_EOI()
    const uint32_t h = d.in.byte(d.i_pos++);
    off = (h << 2) + (chr >> 2) + 1;
    len = 0;
    nstate = (int)(chr & 3);
  } else if (chr < 32) {
    len = chr & 7;
    if (len == 0) {
      LZ_EOI()
      LZ_TRY(count(d, &cnt))
      len = 7 + cnt;
    }
    LZ_EOI()
    if (d.i_pos + 2 > d.in.n) return MD_LZO_OUT_OF_BOUND;
    const uint32_t s = d.in.byte(d.i_pos) | (d.in.byte(d.i_pos + 1) << 8);
    d.i_pos += 2;
    off = 16384 + (((chr & 8) >> 3) << 14) + (s >> 2);
    nstate = (int)(s & 0xff);
    if (off == 16384) {  // end_of_lzo
      *end = true;
      return MD_OK;
    }
  } else if (chr < 64) {
    len = chr & 31;
    if (len == 0) {
      LZ_EOI()
      LZ_TRY(count(d, &cnt))
      len = 31 + cnt;
    }
    LZ_EOI()
    if (d.i_pos + 2 > d.in.n) return MD_LZO_OUT_OF_BOUND;
    const uint32_t s = d.in.byte(d.i_pos) | (d.in.byte(d.i_pos + 1) << 8);
    d.i_pos += 2;
    nstate = (int)(s & 0xff);
    off = (s >> 2) + 1;
  } else {
    nstate = (int)chr;
    len = (chr >> 5) - 1;
    LZ_EOI()
    const uint32_t h = d.in.byte(d.i_pos++);
    off = (h << 3) + ((chr >> 2) & 7) + 1;
  }
  // Copy (lib/lzo.ml:283-288): len + 2 bytes, then copy_done = transmit (state land 3)
  LZ_EOI()
  d.state = nstate;
  LZ_TRY(copy(d, off, len + 2))
  LZ_TRY(transmit(d, (uint32_t)(nstate & 3)))
  return MD_OK;
}

// ---- the batched decoder (round 6) ---------------------------------------------------------------------------------
// LZO's instruction stream is byte-aligned, and what an instruction DOES (copy `len` bytes from `off` back, then 0..3
// literals; or a run of literals) is independent of its neighbours once its first byte and the decoder's state (zero /
// not zero, lib/lzo.ml:322-336) are known.  So a BATCH of the stream is taken in three steps:
//   decode  lane l decodes the instruction that WOULD start at input byte base + l, in both states (they differ only for
//           opcodes below 16) - opcode length, offset, lengths, the state it leaves - and packs "bytes to advance, bytes
//           produced, state left, not for the fast path" into one word per state;
//   walk    the scalar unit follows the chain from lane 0 (base IS an instruction start): one v_readlane and a dozen
//           scalar instructions per instruction, marking the lanes that really are instructions, until the chain leaves
//           the 64 bytes, the batch's output is full, or it meets an instruction the fast path does not take;
//   copy    the marked lanes know their output position from a prefix sum.  Literals go from the input window in LDS to
//           a staging buffer in LDS; matches whose source is older than the batch are read from the output buffer (all
//           lanes at once, unaligned 8-byte loads); matches that reach into the batch itself are taken in stream order
//           by the whole wave, byte j from staging or the output buffer; then the batch leaves in 16-byte chunks.
// Everything else is fiber_step's: one instruction, the reference's checks in the reference's order.  An instruction is
// taken by the fast path only if all of its opcode bytes and literals lie inside the input AND one more byte follows
// (the EOI guard before Copy), its lengths need no zero-byte continuation, it is not the end marker, and - checked after
// the prefix sum - its offset and its bytes fit the output: what the fast path takes cannot fail.
constexpr uint32_t kInRing = 2048, kInBlk = 1024;  // input window: a ring of two 1 KiB blocks, the next one on its way
constexpr uint32_t kStage = 3072;                  // staging: one batch of output, 16-byte aligned with the output buffer
constexpr uint32_t kBatchMax = kStage - 16;        // bytes a batch may produce
constexpr uint32_t kNextZeroBit = 14, kExotic = 1u << 15, kNextZero = 1u << kNextZeroBit;  // (a walk word is 16 bits: advance | flags)
constexpr int kWindows = 4;                       // windows of 64 input bytes per batch (6: the same, 8: registers for 3 wavefronts per SIMD, slower)
struct LSmem {
  alignas(16) uint8_t in[kInRing + 32];  // (+ the ring's first bytes again: an access may run over the end)
  alignas(16) uint8_t stage[kStage + 32];
};
typedef wv::lds_u8 lds_u8;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));  // (HIP's uint4 has no assignment in address space 3)
using wv::lds_ld64;
using wv::lds_put;
using wv::out_ld8;
using wv::out_ld_guard;

struct InRing {
  lds_u8 *ring;
  const uint8_t *src;
  uint32_t n, lane;
  uint32_t lo;   // blocks lo and lo + 1 are in the ring (slot = block & 1); 0xffffffff: nothing
  uint4 ahead;   // this lane's 16 bytes of block lo + 2
  __device__ __forceinline__ uint4 load(uint32_t blk) const {
    const uint64_t a = (uint64_t)blk * kInBlk + lane * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (a + 16 <= n) __builtin_memcpy(&v, src + a, 16);
    else if (a < n) {
      uint32_t w[4] = {0, 0, 0, 0};
      for (uint32_t k = 0; a + k < n; k++) w[k >> 2] |= (uint32_t)src[a + k] << (8 * (k & 3));
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
  }
  __device__ __forceinline__ void put(uint32_t blk, const uint4 &v) {
    const uint32_t slot = (blk & 1) * kInBlk;
    const v4u x = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<MD_LDS v4u *>(ring + slot + lane * 16) = x;
    if (slot == 0 && lane < 2) *reinterpret_cast<MD_LDS v4u *>(ring + kInRing + lane * 16) = x;
  }
  // input bytes [pos, pos + 1 KiB) readable from the ring
  __device__ __forceinline__ void ensure(uint32_t pos) {
    const uint32_t need = pos / kInBlk;
    if (need == lo) return;
    if (need == lo + 1) {
      put(lo + 2, ahead);
      lo = need;
    } else {
      put(need, load(need));
      put(need + 1, load(need + 1));
      lo = need;
    }
    ahead = load(lo + 2);
  }
  __device__ __forceinline__ const lds_u8 *at(uint32_t pos) const { return ring + (pos & (kInRing - 1)); }
};

__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// The instruction that would start with the bytes b (b0 = opcode): opcodes from 16 on are the same in both states of the
// decoder, below 16 a run of literals in state zero and a two-byte match otherwise (lib/lzo.ml:322-369).  Straight-line
// selects: as four branchy decodes per window (two states, before and after the walk) this was 40 divergent branches
// and a third of the decoder's time.
struct InsAll {
  uint32_t kc, offc, mlenc, litc;  // opcode >= 16: opcode bytes, offset, match bytes, literals that follow
  uint32_t kz, litz;               // opcode < 16, state zero: a run of literals (no match)
  uint32_t offn, litn;             // opcode < 16, state not zero: 2 opcode bytes, a match of 2
  bool low, exc, exz;              // opcode < 16; not for the fast path (a length that goes on over zero bytes, the end marker)
};
__device__ __forceinline__ InsAll decode_all(uint32_t b) {
  const uint32_t chr = b & 0xff, b1 = (b >> 8) & 0xff;
  InsAll r;
  r.low = chr < 16;
  const bool m2 = chr >= 64, m3 = chr >= 32;
  uint32_t L = m3 ? chr & 31 : chr & 7;
  const bool ext = L == 0;
  L = ext ? (m3 ? 31u : 7u) + b1 : L;
  const uint32_t s = ext ? b >> 16 : (b >> 8) & 0xffff;
  const uint32_t off34 = m3 ? (s >> 2) + 1 : 16384 + ((chr & 8) << 11) + (s >> 2);
  r.kc = m2 ? 2u : ext ? 4u : 3u;
  r.offc = m2 ? (b1 << 3) + ((chr >> 2) & 7) + 1 : off34;
  r.mlenc = m2 ? (chr >> 5) + 1 : L + 2;
  r.litc = m2 ? chr & 3 : s & 3;
  r.exc = !m2 && ((ext && b1 == 0) || off34 == 16384);
  const bool extz = chr == 0;
  r.kz = extz ? 2u : 1u;
  r.litz = extz ? 18 + b1 : chr + 3;
  r.exz = extz && b1 == 0;
  r.offn = (b1 << 2) + (chr >> 2) + 1;
  r.litn = chr & 3;
  return r;
}
// the walk's word of an instruction at input position p: bytes to advance | the state it leaves; what is not for the fast
// path "advances" out of the window, so that the walk needs no test of its own for it
template <bool CHECK>
__device__ __forceinline__ uint32_t walk_word(uint32_t k, uint32_t lit, bool match, bool exotic, uint32_t p, uint32_t n) {
  const bool inside = !CHECK || (p < n && p + k < n && p + k + lit <= n);
  // a run of literals leaves the state -1 (not zero), a match the number of its literals (lib/lzo.ml:283-288, :322-336)
  const uint32_t fast = (k + lit) | ((match && lit == 0) ? kNextZero : 0u);
  return (exotic || !inside) ? (64u | kExotic) : fast;
}
#ifdef MD_LZO_PROF  // measurement build (tools/dbg/lzo_phases.py): cycles per phase of the decoder, left in out_len
#define LZ_PROF_MARK(k)                                   \
  {                                                       \
    const uint64_t now_ = __builtin_readcyclecounter();   \
    d.prof[k] += (uint32_t)(now_ - d.prof_t);             \
    d.prof_t = now_;                                      \
  }
#else
#define LZ_PROF_MARK(k)
#endif
__device__ __forceinline__ int uncompress_stream(Dec &d, LSmem MD_LDS *sm) {
  LZ_EOI()
  uint32_t chr = d.in.byte(0);
  if (chr == 16) return MD_LZO_NO_DICTIONARY;
  if (chr >= 18) {  // the first byte, lib/lzo.ml:372-393
    d.i_pos = 1;
    d.state = 0;
    LZ_EOI()
    LZ_TRY(transmit(d, chr - 17))
  }
  const uint32_t lane = d.lane;
  lds_u8 *stage = (lds_u8 *)sm->stage;
  InRing ir;
  ir.ring = (lds_u8 *)sm->in;
  ir.src = d.in.src;
  ir.n = d.in.n;
  ir.lane = lane;
  ir.lo = 0xfffffff0u;
  bool slow = false;
  for (;;) {
    if (slow) {
      bool end = false;
      LZ_TRY(fiber_step(d, &end))
      if (end) return MD_OK;
      slow = false;
      continue;
    }
    // ---- a batch: up to kWindows windows of 64 input bytes, one after the other: decode, walk, places, literals - then the
    // matches of all of them and ONE write-out (the fence, the far loads' round trip and the flush are per batch, not per
    // 64 bytes of input: a lone stream spent most of its time on them)
    const uint32_t o0 = d.o_pos, rb = o0 & ~15u, n = d.in.n;
    LZ_PROF_MARK(3)
    uint32_t osum = 0;             // bytes of the batch so far
    uint32_t rec_m[kWindows], rec_p[kWindows];  // per window and lane: off | mlen << 16 ; staging index (0xffffffff: no match)
#pragma unroll
    for (int w = 0; w < kWindows; w++) rec_m[w] = 0, rec_p[w] = 0xffffffffu;
    bool more = true;
#pragma unroll
    for (int w = 0; w < kWindows; w++) {
      if (more) {
        // decode: the 64 instructions that would start at base .. base + 63
        const uint32_t base = d.i_pos;
        ir.ensure(base);
        const uint32_t p = base + lane;
        const uint32_t b = *reinterpret_cast<const MD_LDS wv::u32_u *>(ir.at(p));
        const InsAll ia = decode_all(b);
        const uint32_t kzz = ia.low ? ia.kz : ia.kc, lzz = ia.low ? ia.litz : ia.litc, knn = ia.low ? 2u : ia.kc, lnn = ia.low ? ia.litn : ia.litc;
        const bool ezz = ia.low ? ia.exz : ia.exc, enn = !ia.low && ia.exc;
        uint32_t wz, wn;
        if (base + 64u + 4u + 273u + 1u <= n) {  // far from the input's end (an instruction is <= 4 + 273 bytes): nothing to test
          wz = walk_word<false>(kzz, lzz, !ia.low, ezz, p, n);
          wn = walk_word<false>(knn, lnn, true, enn, p, n);
        } else {
          wz = walk_word<true>(kzz, lzz, !ia.low, ezz, p, n);
          wn = walk_word<true>(knn, lnn, true, enn, p, n);
        }
        // walk (wave-uniform): a dozen scalar instructions per instruction - the two ways out of the fast path are not
        // tested here: an instruction that is not for the fast path leaves the window by itself (pack_ins) and is looked
        // at after the loop, a batch that is full is cut where the places are known (the walk is the stream's own chain:
        // with both tests, and what the compiler made of the three exits, it was 30)
        uint32_t cur = 0, zero = (d.state & 3) == 0 ? 1u : 0u, wd = 0;
        uint64_t taken = 0, zmask = 0;
        const uint32_t wboth = wn | (wz << 16);  // (one lane read per instruction: the word of the state is picked by a shift)
        LZ_PROF_MARK(4)  // (measurement build: the window's decode)
        do {
          wd = rdl(wboth, cur) >> (zero << 4);
          taken |= 1ull << cur;
          zmask |= (uint64_t)zero << cur;
          cur += wd & 1023;
          zero = (wd >> kNextZeroBit) & 1;
        } while (cur < 64);
        if (wd & kExotic) {  // the slow path's: the batch ends in front of it
          const uint32_t last = 63u - (uint32_t)__builtin_clzll(taken);
          taken &= ~(1ull << last);
          cur = last;
          zero = (uint32_t)((zmask >> last) & 1);
          slow = true;
        }
        LZ_PROF_MARK(5)  // (the walk)
        // the marked lanes: their instruction, their place in the output
        bool mine = (taken >> lane) & 1;
        const bool mz = (zmask >> lane) & 1;
        const uint32_t k = ia.low ? (mz ? ia.kz : 2u) : ia.kc, off = ia.low ? (mz ? 0u : ia.offn) : ia.offc;
        const uint32_t mlen = ia.low ? (mz ? 0u : 2u) : ia.mlenc, lit = ia.low ? (mz ? ia.litz : ia.litn) : ia.litc;
        const uint32_t orel = osum + wv::wave_excl_scan(mine ? mlen + lit : 0u, lane);
        const uint32_t oabs = o0 + orel;
        {  // the batch is full: the next one starts at that instruction.  What does not fit the output is the slow path's
           // (it fails there, with the reference's error).  Whichever comes first.
          const uint64_t full = __ballot(mine && orel + mlen + lit > kBatchMax);
          // (the output has room for a whole batch, and every offset - at most 49 151 - has that much output behind it:
          // nothing to test)
          uint64_t bad = 0;
          if (o0 < 49152u || d.cap - o0 < kStage)
            bad = __ballot(mine && ((mlen != 0 && off > oabs) || mlen + lit > d.cap - oabs || oabs > d.cap));
          if (full | bad) {
            const uint32_t fb = (uint32_t)__builtin_ctzll(full | bad);
            taken &= (1ull << fb) - 1;
            mine = (taken >> lane) & 1;
            cur = fb;
            zero = (uint32_t)((zmask >> fb) & 1);
            slow = ((bad >> fb) & 1) && !((full >> fb) & 1);
            if (!slow) more = false;
          }
        }
        uint32_t wsum = 0;
        if (taken) {
          const uint32_t lt = 63u - (uint32_t)__builtin_clzll(taken);
          wsum = rdl(orel + mlen + lit, lt) - osum;
        }
        LZ_PROF_MARK(6)  // (places and cuts)
        const uint32_t sidx = oabs - rb;  // staging index of this lane's first byte
        if (taken) {  // literals: input window -> staging
          const uint32_t lsrc = p + k, ldst = sidx + mlen;
          if (mine && lit != 0 && lit <= 16) {
            const uint64_t v0 = lds_ld64(ir.at(lsrc));
            lds_put(stage + ldst, v0, lit < 8 ? lit : 8);
            if (lit > 8) lds_put(stage + ldst + 8, lds_ld64(ir.at(lsrc + 8)), lit - 8);
          }
          for (uint64_t lm = __ballot(mine && lit > 16); lm; lm &= lm - 1) {  // a long run: by the whole wave (<= 273 bytes)
            const uint32_t l = (uint32_t)__builtin_ctzll(lm);
            const uint32_t sp = rdl(lsrc, l), t = rdl(ldst, l), c = rdl(lit, l), j = lane * 8;
            if (j < c) lds_put(stage + t + j, lds_ld64(ir.at(sp + j)), c - j < 8 ? c - j : 8);
          }
          if (mine && mlen != 0) {
            rec_m[w] = off | (mlen << 16);
            rec_p[w] = sidx;
          }
        }
        osum += wsum;
        d.i_pos = base + cur;
        d.state = zero ? 0 : 1;
        if (slow || taken == 0) more = false;
        LZ_PROF_MARK(0)  // (literals into staging, records)
      }
    }
    LZ_PROF_MARK(0)
    if (osum) {
      // ---- matches.  Older than the batch: from the output buffer, every lane its own, the first 16 bytes of all windows'
      // records in flight together; into the batch: in stream order by the whole wave, byte j = source byte j mod off
      bool anyfar = false;
#pragma unroll
      for (int w = 0; w < kWindows; w++) {
        const uint32_t mlen = rec_m[w] >> 16, off = rec_m[w] & 0xffffu;
        anyfar = anyfar || (rec_p[w] != 0xffffffffu && rb + rec_p[w] - off < o0);
      }
      if (__ballot(anyfar)) {  // some source byte lies in the output buffer
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // earlier batches' bytes have landed
        if (o0 + 8 <= d.cap) {  // whole 8-byte words: up to 7 bytes past a source, which ends at o0 at the latest
          uint64_t v0[kWindows], v1[kWindows];
#pragma unroll
          for (int w = 0; w < kWindows; w++) {
            const uint32_t mlen = rec_m[w] >> 16, off = rec_m[w] & 0xffffu, sabs = rb + rec_p[w] - off;
            const bool far = rec_p[w] != 0xffffffffu && sabs + mlen <= o0;
            const uint8_t *g = d.dst + (far ? sabs : 0u);
            v0[w] = wv::out_ld64(g);
            v1[w] = wv::out_ld64(g + (far && mlen > 8 ? 8 : 0));
          }
#pragma unroll
          for (int w = 0; w < kWindows; w++) {
            const uint32_t mlen = rec_m[w] >> 16, off = rec_m[w] & 0xffffu, sabs = rb + rec_p[w] - off;
            const bool far = rec_p[w] != 0xffffffffu && sabs + mlen <= o0;
            if (far) {
              lds_u8 *t = stage + rec_p[w];
              lds_put(t, v0[w], mlen < 8 ? mlen : 8);
              if (mlen > 8) lds_put(t + 8, v1[w], mlen < 16 ? mlen - 8 : 8);
              for (uint32_t j = 16; j < mlen; j += 8) lds_put(t + j, wv::out_ld64(d.dst + sabs + j), mlen - j < 8 ? mlen - j : 8);
            }
          }
        } else {  // the last bytes of the output buffer: nothing is read beyond it
#pragma unroll
          for (int w = 0; w < kWindows; w++) {
            const uint32_t mlen = rec_m[w] >> 16, off = rec_m[w] & 0xffffu, sabs = rb + rec_p[w] - off;
            if (rec_p[w] != 0xffffffffu && sabs + mlen <= o0)
              for (uint32_t j = 0; j < mlen; j += 8)
                lds_put(stage + rec_p[w] + j, out_ld_guard(d.dst, sabs + j, mlen - j, d.cap), mlen - j < 8 ? mlen - j : 8);
          }
        }
      }
      LZ_PROF_MARK(1)
#pragma unroll
      for (int w = 0; w < kWindows; w++) {
        const uint32_t mlen = rec_m[w] >> 16, off = rec_m[w] & 0xffffu, sidx = rec_p[w], sabs = rb + sidx - off;
        const bool near = sidx != 0xffffffffu && sabs + mlen > o0;
        for (uint64_t nm = __ballot(near); nm; nm &= nm - 1) {
          const uint32_t l = (uint32_t)__builtin_ctzll(nm);
          const uint32_t sp = rdl(sabs, l), t = rdl(sidx, l), c = rdl(mlen, l), f = rdl(off, l);
          if (f >= c && sp >= o0 && c <= (uint32_t)kWave) {  // the usual one: inside the batch, no overlap, one step
            if (lane < c) stage[t + lane] = stage[sp - rb + lane];
            continue;
          }
          const float inv = 1.0f / (float)f;
          for (uint32_t j = lane; j < c; j += kWave) {
            uint32_t r = j;
            if (f < c) {
              r = j - (uint32_t)((float)j * inv) * f;  // j mod f, the quotient may be one off either way
              r = (int32_t)r < 0 ? r + f : r;
              r = r >= f ? r - f : r;
            }
            const uint32_t x = sp + r;
            stage[t + j] = x >= o0 ? stage[x - rb] : (uint8_t)out_ld8(d.dst + x);
          }
        }
      }
      LZ_PROF_MARK(2)
      // the batch leaves: whole 16-byte chunks one per lane, the bytes in front of the first and behind the last one by one
      {
        const uint32_t endp = o0 + osum;
        const uint32_t full0 = (o0 + 15) & ~15u, full1 = endp & ~15u;
        for (uint32_t ps = full0; ps < full1; ps += kWave * 16) {
          const uint32_t c = ps + lane * 16;
          if (c < full1) {  // (16-byte aligned offset; the buffer itself may start anywhere)
            const v4u v = *reinterpret_cast<const MD_LDS v4u *>(stage + (c - rb));
            __builtin_memcpy(d.dst + c, &v, 16);
          }
        }
        const uint32_t head1 = full0 < endp ? full0 : endp, tail0 = full1 > head1 ? full1 : head1;
        const uint32_t x = lane < 16 ? o0 + lane : tail0 + (lane - 16);
        if (lane < 16 ? x < head1 : (lane < 32 && x < endp)) d.dst[x] = stage[x - rb];
        d.o_pos = endp;
      }
    }
  }
}

__global__ __launch_bounds__(kWave) void lzo_uncompress_kernel(
    uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    const uint64_t *__restrict__ out_cap, uint64_t *__restrict__ out_len, int32_t *__restrict__ status, uint32_t *counter) {
  __shared__ LSmem smem;
  const uint32_t lane = threadIdx.x;
  for (;;) {
    const uint32_t sid = next_stream(counter, lane);
    if (sid >= n) return;
    Dec d;
    d.in.src = in + in_off[sid];
    const uint64_t l64 = in_len[sid], c64 = out_cap[sid];
    if (l64 > MD_MAX_STREAM) {  // 32-bit cursors: the descriptor is out of range (mdeflate.h)
      if (lane == 0) {
        status[sid] = MD_E_INVALID_ARGUMENT;
        out_len[sid] = 0;
      }
      continue;
    }
    d.in.n = (uint32_t)l64;
    d.in.lane = lane;
    d.in.fill(0);
    d.dst = out + out_off[sid];
    d.cap = c64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)c64;
    d.i_pos = d.o_pos = 0;
    d.lane = lane;
    d.state = 0;
#ifdef MD_LZO_PROF
    for (int k_ = 0; k_ < 8; k_++) d.prof[k_] = 0;
    d.prof_t = __builtin_readcyclecounter();
#endif
    const int st = uncompress_stream(d, (LSmem MD_LDS *)&smem);
    if (lane == 0) {
      status[sid] = st;
      out_len[sid] = st == MD_OK ? d.o_pos : 0;
#ifdef MD_LZO_PROF  // 8 x 8 bits: a phase's share of the stream's cycles in 1/255 (tools/dbg/lzo_phases.py names them)
      {
        uint64_t tot_ = 0, pk_ = 0;
        for (int k_ = 0; k_ < 8; k_++) tot_ += d.prof[k_];
        for (int k_ = 0; k_ < 8; k_++) pk_ |= (uint64_t)((uint64_t)d.prof[k_] * 255 / (tot_ ? tot_ : 1)) << (8 * k_);
        out_len[sid] = pk_;
        status[sid] = (int32_t)(tot_ >> 10);
      }
#endif
    }
  }
}


// ---------------------------------------------------------------------------------------------
// compress

struct Cmp {
  const uint8_t *src;
  uint32_t n;  // whole input
  uint8_t *dst;
  uint32_t cap, lane;
  uint32_t op;   // output position
  bool oob;      // Out_of_bound (lib/lzo.ml:655)
  // the byte at op - 2 when it is the second-to-last byte of the latest match (record_literals /
  // record_trailer OR the literal count into its two low bits, lib/lzo.ml:509, :548)
  uint32_t patch;
  // The last bytes of the latest match (its offset bytes, and its marker when that is one byte) wait HERE until the run
  // of literals behind it is known: the count of a short run goes into the second-to-last of them, and then they leave
  // with one store (lane k takes byte k) instead of a byte store each and another one for the count.  `op` has them
  // counted already.
  uint32_t pend, pend_n, pend_pos;
};
__device__ __forceinline__ void c_set(Cmp &c, uint32_t pos, uint32_t b) {  // wave-uniform byte store
  if (pos >= c.cap) c.oob = true;
  else if (c.lane == 0) c.dst[pos] = (uint8_t)b;
}
__device__ __forceinline__ void flush_pending(Cmp &c) {
  if (c.pend_n == 0) return;
  if (c.pend_pos + c.pend_n > c.cap) c.oob = true;
  if (c.lane < c.pend_n && c.pend_pos + c.lane < c.cap) c.dst[c.pend_pos + c.lane] = (uint8_t)(c.pend >> (8 * c.lane));
  c.pend_n = 0;
}
// the count of a run of 1..3 literals goes into the latest match (lib/lzo.ml:509, :548)
__device__ __forceinline__ void patch_pending(Cmp &c, uint32_t len) {
  if (c.pend_n) c.pend |= len << (8 * (c.pend_n - 2));
  else if (c.op < 2) c.oob = true;
  else c_set(c, c.op - 2, c.patch | len);
}
// blit in_data off out_data op len (lib/lzo.ml:81-89) — `room` bytes must fit both buffers, `len`
// of them are kept (the reference copies 4 / 16 bytes for short runs and overwrites the excess)
__device__ __forceinline__ bool blit_fits(Cmp &c, uint32_t off, uint32_t room) {
  // (room >= 1 at every call; written without a borrow so that it stays on the scalar unit)
  const uint32_t in_left = c.n > off ? c.n - off : 0u, out_left = c.cap > c.op ? c.cap - c.op : 0u;
  if (room > in_left || room > out_left) {
    c.oob = true;
    return false;
  }
  return true;
}
__device__ __forceinline__ void c_blit(Cmp &c, uint32_t off, uint32_t len, uint32_t room) {
  if (!blit_fits(c, off, room)) return;
  const uint8_t *s = c.src + off;
  uint8_t *o = c.dst + c.op;
  if (len < 256) {
#pragma clang loop vectorize(disable) unroll(disable)
    for (uint32_t k = c.lane; k < len; k += kWave) o[k] = s[k];
    return;
  }
  // a long run (incompressible input is one run per chunk): 16 bytes per lane and step, stores on the output's alignment
  const uint32_t head = (16u - (uint32_t)((uintptr_t)o & 15)) & 15u;
  const uint32_t body = (len - head) & ~15u;
  if (c.lane < head) o[c.lane] = s[c.lane];
#pragma clang loop vectorize(disable) unroll(disable)
  for (uint32_t k = head + c.lane * 16; k < head + body; k += kWave * 16) {
    uint4 v;
    __builtin_memcpy(&v, s + k, 16);
    *reinterpret_cast<uint4 *>(o + k) = v;
  }
  const uint32_t k = head + body + c.lane;
  if (k < len) o[k] = s[k];
}
__device__ __forceinline__ void long_run(Cmp &c, uint32_t len) {  // 19+ literals: 0, 0..., rest
  uint32_t l = len - 18;
  c_set(c, c.op++, 0);
  while (l > 255) {
    l -= 255;
    c_set(c, c.op++, 0);
  }
  c_set(c, c.op++, l);
}
// record_literals, lib/lzo.ml:502-538
__device__ __forceinline__ void record_literals(Cmp &c, uint32_t off, uint32_t len) {
  if (len == 0) return;
  if (len <= 3) {
    patch_pending(c, len);
    flush_pending(c);
    c_blit(c, off, len, 4);
  } else if (len <= 16) {
    flush_pending(c);
    c_set(c, c.op++, len - 3);
    c_blit(c, off, len, 16);
  } else {
    flush_pending(c);
    if (len <= 18) c_set(c, c.op++, len - 3);
    else long_run(c, len);
    c_blit(c, off, len, len);
  }
  c.op += len;
}
// record_match, lib/lzo.ml:443-500
__device__ __forceinline__ void record_match(Cmp &c, uint32_t off, uint32_t len) {
  flush_pending(c);
  if (len <= 8 && off <= 0x0800) {
    off -= 1;
    c.patch = ((len - 1) << 5) | ((off & 7) << 2);
    c.pend = c.patch | ((off >> 3) << 8);
    c.pend_n = 2;
    c.pend_pos = c.op;
    c.op += 2;
    return;
  }
  uint32_t marker, maxl;
  if (off <= 0x4000) {
    off -= 1;
    marker = 32;
    maxl = 33;
  } else {
    off -= 0x4000;
    marker = 16 | ((off >> 11) & 8);
    maxl = 9;
  }
  uint32_t head = 0, head_n = 0;
  if (len <= maxl) {
    head = marker | (len - 2);
    head_n = 1;
  } else {
    uint32_t l = len - maxl;
    c_set(c, c.op++, marker);
    while (l > 255) {
      l -= 255;
      c_set(c, c.op++, 0);
    }
    c_set(c, c.op++, l);
  }
  c.patch = (off << 2) & 0xff;
  c.pend = head | (((off << 2) & 0xff) << (8 * head_n)) | (((off >> 6) & 0xff) << (8 * head_n + 8));
  c.pend_n = head_n + 2;
  c.pend_pos = c.op;
  c.op += head_n + 2;
}
// record_trailer, lib/lzo.ml:540-576
__device__ __forceinline__ void record_trailer(Cmp &c, uint32_t off, uint32_t len) {
  if (len > 0) {
    if (c.op == 0 && len < 238) c_set(c, c.op++, 17 + len);
    else if (len <= 3) patch_pending(c, len);
    else {
      flush_pending(c);
      if (len <= 18) c_set(c, c.op++, len - 3);
      else long_run(c, len);
    }
    flush_pending(c);
    c_blit(c, off, len, len);
  }
  flush_pending(c);
  c.op += len;
  c_set(c, c.op++, 16 | 1);
  c_set(c, c.op++, 0);
  c_set(c, c.op++, 0);
}

// A dictionary entry as it is in L2 (the wavefront's own stores are there, in order).  As a non-temporal load - the way
// past a stale line in the CU's cache until now - the entries were also the first to leave L2: with the dictionaries of
// 16 and more streams per CU at four times the L2 of their XCD, 63 % of all L2 requests missed (4 096 text streams: 31.1
// ms; like this: 25.7).
__device__ __forceinline__ uint16_t dict_ld(const uint16_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct CSmem {
  alignas(16) uint8_t in[kInRing + 32];  // the input around the probe position (InRing)
  uint8_t tbl[1024];                     // which lane probed a dictionary slot in this step
};
#ifndef MD_LZO_FIRST
#define MD_LZO_FIRST 8
#endif
#ifndef MD_LZO_C_WAVES
#define MD_LZO_C_WAVES 6
#endif
constexpr uint32_t kFirstProbes = MD_LZO_FIRST;  // probes taken in the step right behind a match

// The steps right behind a match (`next:` / `literal:` of lib/lzo.ml:596-640 while the stride is still 1, that is up to 32
// bytes behind the match).  In text the next match starts a few bytes on (C5's text: 48 % of the matches at once, 89 %
// within 8 bytes, 98 % within 16; 2.6 literals between matches, matches of 6.7 bytes), so this step decides the kernel.  It
// takes kFirstProbes positions and keeps every dependent memory round trip off the chain that it can:
//   * the bytes at the probe positions, and the 12 behind them, come from the input ring in LDS;
//   * a lane reads 16 bytes at its reference, not 4: a match shorter than 16 (99 %) needs no second look at memory;
//   * right behind a match the literals up to the next one are the low bytes of the lanes in front of the hit;
//   * two lanes on one dictionary slot (then the later one's reference is the earlier lane, not the dictionary's entry)
//     are looked for with one byte store and load per lane in a small table in LDS; if there are any in front of the hit
//     - or a false alarm of the table - the step is the general one's.
// Returns 0: done, go on at `first`; 1: the chunk ends (*ret); 2: nothing done, take the general step.
__device__ __forceinline__ int near_step(Cmp &c, InRing &ring, lds_u8 *tbl, uint16_t *dict, uint32_t in_pos, uint32_t in_len,
                                         uint32_t idx_end, uint32_t &first, uint32_t &idx1, uint32_t &t, uint32_t *ret) {
  const uint32_t lane = c.lane;
  const uint32_t room = first - in_pos < idx_end ? idx_end - (first - in_pos) : 0u;
  const uint32_t nvalid = room < kFirstProbes ? room : kFirstProbes;
  if (nvalid == 0) {
    idx1 -= t;
    *ret = in_len - (idx1 - in_pos);
    return 1;
  }
  ring.ensure(first);
  const bool valid = lane < nvalid;
  const uint32_t mine = first + lane;
  uint32_t w0 = 0, x1 = 0, x2 = 0, x3 = 0, r0 = 0, ref = 0, index = 0, back = lane;
  if (valid) {
    const lds_u8 *q = ring.at(mine);
    w0 = *reinterpret_cast<const MD_LDS wv::u32_u *>(q);
    index = ((uint32_t)(0x1824429du * w0) >> 18) & 0x3fff;
    ref = (uint32_t)dict_ld(dict + index) + in_pos;
    volatile lds_u8 *slot = tbl + (index & 1023u);  // (volatile: what comes back is the LAST lane's, not this one's)
    *slot = (uint8_t)lane;
    // (a reference lies in front of its probe, and a probe 20 bytes in front of the chunk's end: 16 bytes are there)
    uint32_t r[4];
    __builtin_memcpy(r, c.src + ref, 16);
    r0 = r[0];
    x1 = r[1] ^ *reinterpret_cast<const MD_LDS wv::u32_u *>(q + 4);
    x2 = r[2] ^ *reinterpret_cast<const MD_LDS wv::u32_u *>(q + 8);
    x3 = r[3] ^ *reinterpret_cast<const MD_LDS wv::u32_u *>(q + 12);
    back = *slot;
  }
  const uint64_t hit = __ballot(valid && r0 == w0);
  const uint64_t dup = __ballot(back != lane);
  const uint32_t stop = hit ? (uint32_t)__builtin_ctzll(hit) : nvalid - 1;
  if (dup & ((2ull << stop) - 1)) return 2;
  if (valid && lane <= stop) dict[index] = (uint16_t)(mine - in_pos);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (!hit) {
    first += nvalid;
    return 0;
  }
  const uint32_t idx0 = first + stop;
  if (first != idx1 || t != 0) {
    idx1 -= t;
    t = 0;
    record_literals(c, idx1, idx0 - idx1);
  } else if (stop != 0) {
    // ---- stop literals (record_literals, lib/lzo.ml:502-538) with the waiting bytes of the match in front of them
    bool fits;
    if (stop <= 3) {
      patch_pending(c, stop);
      fits = blit_fits(c, idx1, 4);
    } else {
      if (c.pend_n) {
        c.pend |= (stop - 3) << (8 * c.pend_n);
        c.pend_n++;
      } else {
        c.pend = stop - 3, c.pend_n = 1, c.pend_pos = c.op;
      }
      c.op++;
      fits = blit_fits(c, idx1, 16);
    }
    flush_pending(c);
    if (fits && lane < stop) c.dst[c.op + lane] = (uint8_t)w0;
    c.op += stop;
  }
  // ---- the match: 4 bytes are known to agree, the next 12 are compared in lane `stop` (the loop of lib/lzo.ml:616-631
  // goes 8 bytes at a time while the probe is in front of idx_end, then counts the bytes of the next 8 that still agree)
  const uint64_t x = ((uint64_t)x2 << 32) | x1;
  const bool in_front = mine + 4 - in_pos < idx_end;
  uint32_t mlen;  // bit 31: go on from 12 with memory
  if (x) mlen = 4 + ((uint32_t)__builtin_ctzll(x) >> 3);
  else if (!in_front) mlen = 4;
  else if (x3) mlen = 12 + ((uint32_t)__builtin_ctz(x3) >> 3);
  else mlen = 0x80000000u;
  const uint32_t mref = rdl(ref, stop);
  uint32_t len = rdl(mlen, stop);
  if (len >> 31) {
    len = 12;
    for (;;) {  // 8 bytes per lane, stops at the first lane that fails
      const uint32_t o = len + 8 * lane;
      const bool inb = idx0 + o + 8 <= c.n;
      uint64_t a = 0, b = 0;
      if (inb) {
        __builtin_memcpy(&a, c.src + idx0 + o, 8);
        __builtin_memcpy(&b, c.src + mref + o, 8);
      }
      const bool go = idx0 + o - in_pos < idx_end && inb && a == b;
      const uint64_t fm = __ballot(!go);
      if (fm == 0) {
        len += 8 * kWave;
        continue;
      }
      const uint32_t L = (uint32_t)__builtin_ctzll(fm);
      len += 8 * L;
      const uint64_t y = a ^ b;
      const uint32_t extra = (inb && y) ? (uint32_t)__builtin_ctzll(y) >> 3 : 0u;
      if (idx0 + len - in_pos < in_len) len += rdl(extra, L);
      break;
    }
  }
  record_match(c, idx0 - mref, len);
  first = idx0 + len;
  idx1 = first;
  return 0;
}

// One 48 KiB chunk (lib/lzo.ml:578-640).  The probe positions of a literal run are known in advance
// (`literal: ip += 1 + ((ip - ii) >> 5)` does not look at the data), so 64 of them are probed per
// step: lane j hashes its position, the dictionary entry it would see is the latest earlier lane of
// the step with the same hash or else the LDS entry, and the first lane whose 4 bytes agree with its
// reference is where the serial loop finds its match; the lanes up to it enter the dictionary in
// order.  After a match the next probe is the match end itself (`goto next`).
__device__ __forceinline__ uint32_t compress_chunk(Cmp &c, InRing &ring, lds_u8 *tbl, uint16_t *dict, uint32_t in_pos, uint32_t in_len,
                                                uint32_t t) {
  const uint32_t idx_end = in_len > 20 ? in_len - 20 : 0, lane = c.lane;
  uint32_t idx1 = in_pos;
  uint32_t first = in_pos + (t < 4 ? 4 - t : 0);
  first += 1 + ((first - idx1) >> 5);
  for (;;) {
    if (first - idx1 <= 32u - kFirstProbes) {  // behind a match (or the chunk's start), stride 1
      uint32_t ret = 0;
      const int how = near_step(c, ring, tbl, dict, in_pos, in_len, idx_end, first, idx1, t, &ret);
      if (how == 0) continue;
      if (how == 1) return ret;
    }
    // ---- the next probe positions of this run (wave-uniform recurrence, lane j keeps p_j)
    uint32_t p = first, mine = 0, nvalid = 0;
    if (first - idx1 < 24u) {
      // right after a match (the usual case in compressible input, where the first few probes already hit): the
      // stride is 1 until the run is 32 long, so the probes up to there need no recurrence; the rest of the run,
      // if it goes on, is the next step's
      const uint32_t ahead = 32u - (first - idx1), room = first - in_pos < idx_end ? idx_end - (first - in_pos) : 0u;
      nvalid = ahead < room ? ahead : room;
      mine = first + lane;
      p = first + nvalid;
    } else {
      for (uint32_t j = 0; j < (uint32_t)kWave; j++) {
        if (p - in_pos >= idx_end) break;
        if (lane == j) mine = p;
        nvalid++;
        p += 1 + ((p - idx1) >> 5);
      }
    }
    if (nvalid == 0) {  // the run reaches the end of the chunk
      idx1 -= t;
      return in_len - (idx1 - in_pos);
    }
    const bool valid = lane < nvalid;
    const uint64_t vm = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1);
    uint32_t v = 0;
    if (valid) __builtin_memcpy(&v, c.src + mine, 4);
    const uint32_t index = ((uint32_t)(0x1824429du * v) >> 18) & 0x3fff;
    // latest earlier lane of the step with the same dictionary slot
    uint64_t same = vm;
#pragma unroll
    for (int bit = 0; bit < 14; bit++) {
      const bool bset = (index >> bit) & 1;
      const uint64_t bal = __ballot(bset);
      same &= bset ? bal : ~bal;
    }
    const uint64_t below = valid ? same & ((1ull << lane) - 1) : 0;
    const uint32_t pred = (uint32_t)__shfl((int)mine, below ? 63 - (int)__builtin_clzll(below) : 0);
    uint32_t ref = 0, rv = 0;
    if (valid) {
      ref = below ? pred : (uint32_t)dict_ld(dict + index) + in_pos;
      __builtin_memcpy(&rv, c.src + ref, 4);
    }
    const uint64_t hit = __ballot(valid && rv == v);
    const uint32_t stop = hit ? (uint32_t)__builtin_ctzll(hit) : nvalid - 1;
    {  // lanes up to the stop enter the dictionary; of equal slots the last one stays
      const uint64_t upto = stop == 63 ? ~0ull : ((2ull << stop) - 1);
      const uint64_t later = lane == 63 ? 0 : same & upto & ~((2ull << lane) - 1);
      if (valid && lane <= stop && later == 0) dict[index] = (uint16_t)(mine - in_pos);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the entries have landed before the next step reads
    if (!hit) {
      first = p;
      continue;
    }
    // ---- the match at lane `stop`
    const uint32_t idx0 = rdl(mine, stop);  // (readlane: the compiler knows the result is the same in all lanes)
    const uint32_t mref = rdl(ref, stop);
    idx1 -= t;
    t = 0;
    record_literals(c, idx1, idx0 - idx1);
    uint32_t len = 4;
    for (;;) {  // 8 bytes per lane; the loop of lib/lzo.ml:616-621 stops at the first lane that fails
      const uint32_t o = len + 8 * lane;
      const bool inb = idx0 + o + 8 <= c.n;
      uint64_t a = 0, b = 0;
      if (inb) {
        __builtin_memcpy(&a, c.src + idx0 + o, 8);
        __builtin_memcpy(&b, c.src + mref + o, 8);
      }
      const bool go = idx0 + o - in_pos < idx_end && inb && a == b;
      const uint64_t fm = __ballot(!go);
      if (fm == 0) {
        len += 8 * kWave;
        continue;
      }
      const uint32_t L = (uint32_t)__builtin_ctzll(fm);
      len += 8 * L;
      // then the bytes of the next 8 that still agree (lib/lzo.ml:624-631; ctz 0 = 0)
      const uint64_t x = a ^ b;
      const uint32_t extra = (inb && x) ? (uint32_t)__builtin_ctzll(x) >> 3 : 0u;
      if (idx0 + len - in_pos < in_len) len += rdl(extra, L);
      break;
    }
    record_match(c, idx0 - mref, len);
    first = idx0 + len;  // `goto next`: the match end is probed as it is
    idx1 = first;
  }
}

__global__ __launch_bounds__(kWave, MD_LZO_C_WAVES) void lzo_compress_kernel(
    uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    const uint64_t *__restrict__ out_cap, uint64_t *__restrict__ out_len, int32_t *__restrict__ status,
    uint16_t *__restrict__ ws_dict, uint32_t *counter) {
  const uint32_t lane = threadIdx.x;
  __shared__ CSmem smem;
  CSmem MD_LDS *sm = (CSmem MD_LDS *)&smem;
  // make_wrkmem (lib/lzo.ml:645-646): 16 K u16 entries per (persistent) workgroup in an HBM workspace — in LDS the
  // 32 KiB would hold residency to 5 wavefronts per CU, and the probe loop lives on occupancy
  uint16_t *dict = ws_dict + (size_t)blockIdx.x * (1u << 14);
  for (;;) {
    const uint32_t sid = next_stream(counter, lane);
    if (sid >= n) return;
    Cmp c;
    c.src = in + in_off[sid];
    const uint64_t l64 = in_len[sid], c64 = out_cap[sid];
    if (l64 > MD_MAX_STREAM) {  // 32-bit cursors: the descriptor is out of range (mdeflate.h)
      if (lane == 0) {
        status[sid] = MD_E_INVALID_ARGUMENT;
        out_len[sid] = 0;
      }
      continue;
    }
    c.n = (uint32_t)l64;
    c.dst = out + out_off[sid];
    c.cap = c64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)c64;
    c.lane = lane;
    c.op = 0;
    c.oob = false;
    c.patch = 0;
    c.pend = 0, c.pend_n = 0, c.pend_pos = 0;
    InRing ring;
    ring.ring = sm->in;
    ring.src = c.src;
    ring.n = c.n;
    ring.lane = lane;
    ring.lo = 0xfffffff0u;
    ring.ahead = make_uint4(0, 0, 0, 0);
    // Lzo.compress, lib/lzo.ml:648-660
    uint32_t idx = 0, len = c.n, t = 0;
    while (len > 20) {
      const uint32_t ll = len < 49152u ? len : 49152u;
      if (((t + ll) >> 5) == 0) break;
      for (uint32_t i = lane; i < (1u << 11); i += kWave) reinterpret_cast<uint4 *>(dict)[i] = make_uint4(0, 0, 0, 0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      t = compress_chunk(c, ring, sm->tbl, dict, idx, ll, t);
      idx += ll;
      len -= ll;
    }
    t += len;
    record_trailer(c, c.n - t, t);
    if (lane == 0) {
      status[sid] = c.oob ? MD_LZO_OUT_OF_BOUND : MD_OK;
      out_len[sid] = c.oob ? 0 : c.op;
    }
  }
}

}  // namespace lzo
}  // namespace md

// grid: persistent workgroups, as many as the device holds at once (asked of the runtime: registers and LDS decide),
// never more than there are streams
template <class K>
static uint32_t resident_workgroups(K kernel, uint32_t cus) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, md::lzo::kWave, 0) != hipSuccess || per_cu < 1) per_cu = 8;
  return cus * (uint32_t)per_cu;
}
extern "C" uint32_t md_lzo_slots(int compress, uint32_t cus) {
  static uint32_t per_cu[2] = {0, 0};  // (per CU: the same on every device this library runs on)
  uint32_t &v = per_cu[compress ? 1 : 0];
  if (!v) v = compress ? resident_workgroups(md::lzo::lzo_compress_kernel, 1) : resident_workgroups(md::lzo::lzo_uncompress_kernel, 1);
  // The compressor is a chain of dependent memory round trips per stream: beyond four wavefronts per SIMD a CU gains
  // little (32 streams per CU: +12 % over 16) and a batch loses more at its end - the last streams run alone at the
  // slow rate, and in a mixed batch the CUs end up with unequal numbers of the expensive streams (C5: 14 to 20 text
  // streams per CU with 28 slots, 16 each with 16: 32.0 -> 26.8 ms).
  if (compress && v > 16) v = 16;
  return v * cus;
}
extern "C" int md_launch_lzo_uncompress(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                        uint8_t *out, const uint64_t *out_off, const uint64_t *out_cap,
                                        uint64_t *out_len, int32_t *status, uint32_t *counter, uint32_t slots, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::lzo::lzo_uncompress_kernel, dim3(n < slots ? n : slots), dim3(md::lzo::kWave), 0, stream, n, in, in_off, in_len,
                     out, out_off, out_cap, out_len, status, counter);
  return (int)hipGetLastError();
}

extern "C" int md_launch_lzo_compress(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                      uint8_t *out, const uint64_t *out_off, const uint64_t *out_cap,
                                      uint64_t *out_len, int32_t *status, uint16_t *ws_dict, uint32_t *counter, uint32_t slots,
                                      hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::lzo::lzo_compress_kernel, dim3(n < slots ? n : slots), dim3(md::lzo::kWave), 0, stream, n, in, in_off, in_len, out,
                     out_off, out_cap, out_len, status, ws_dict, counter);
  return (int)hipGetLastError();
}
