// deflate_common.hpp — what the three kernels of the deflate path share (gfx950).
//
//   deflate_plan_kernel   per-stream slots of the position-indexed workspace (a scan over the batch)
//   deflate_link_kernel   hash chains: link[p] = distance to the latest earlier position with the hash of p (+ fp16)
//                         (insert_string, lib/de.ml:4220-4226; head table in LDS, one workgroup per stream)
//   deflate_match_kernel  longest_match (lib/de.ml:4110-4174) for EVERY position, every position independent
//   deflate_kernel        the sequential part: lazy evaluation, De.Queue, De.T, De.Def, the drivers
//
// In deflate_slow every position p < p_end is inserted into the hash chains exactly once and in order, whatever
// the matcher decides, so the chain structure is a pure function of the input: the first two kernels compute it
// (and what longest_match would find at each position) for the whole batch at full occupancy, the third one only
// takes the decisions.  See DESIGN.md 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdeflate.h"

namespace md {
namespace defl {

constexpr int kWave = 64;
constexpr int MIN_MATCH = 3, MAX_MATCH = 258, MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1;
constexpr int HASH_BITS = 15, HASH_SIZE = 1 << HASH_BITS, TOO_FAR = 4096;
constexpr int WSIZE = 1 << 15, WMASK = WSIZE - 1, MAX_DIST = WSIZE - MIN_LOOKAHEAD;

// look-ahead verdict of a position (flg[]): FL_ENDED no candidate of its chain shares 3 bytes with it (longest_match
// cannot improve on any prev_length), FL_MATCH the whole longest_match was run ahead and its result is in m[] / mq[]
// (full / quartered chain, (length << 16) | distance), 0 the matcher has to look for itself
enum { FL_ENDED = 2, FL_MATCH = 4 };
constexpr int KSPEC = 128;  // chain links walked ahead per position at most (min(max_chain, KSPEC))
#ifndef MD_PGM
#define MD_PGM 4
#endif
#ifndef MD_MATCH_WAVES
#define MD_MATCH_WAVES 6
#endif
constexpr int PGM = MD_PGM;  // steps of 64 positions one wavefront of the match kernel keeps in flight
constexpr uint32_t kChunk = PGM * kWave;

// Position-indexed workspace of a batch.  Stream i owns positions [slot[i], slot[i + 1]) of link / flg / m / mq.
struct Front {
  const uint32_t *p_end;   // [n]     positions < p_end[i] are inserted ahead (and have a verdict)
  const uint64_t *slot;    // [n + 1]
  const uint32_t *chunk0;  // [n + 1] first kChunk-position chunk of stream i in the match kernel's grid
  const uint32_t *tail;    // [2 n]   hash head of position len - 3 (De matcher): 4th byte 0 / the byte 32 KiB earlier (H7)
  const uint32_t *flags;   // [1]     bit 0: the batch needs more workspace than the caller's size hint allowed
  uint32_t *link;          // low 16 bits: distance to the previous position with the same hash, 0 = none within
                           // 32767; high 16 bits: a fingerprint of the 3 bytes at the position (fp16)
  uint8_t *flg;            // FL_*
  uint32_t *m, *mq;        // longest_match ahead over the full / quartered chain, valid where flg == FL_MATCH
};

// A stream compressed IN PIECES (the encoder of `Def.encode` while input is still arriving, lib/de.ml:4294-4349,
// lib/zl.ml:523-555): the sequential kernel's state - two LDS structs - is written out when the matcher wants input the
// piece does not hold, and read back when the next piece begins.  Every pointer null: whole streams (the batch path).
//   flags[i]  bit 0: the stream's first piece, bit 1: its last one (the end of the input has been signalled), bit 2: the
//             Adler-32 is the kernel's own, carried in the state (else sum[2i] is the caller's running checksum), bit 3:
//             nothing to do for this stream (it ended in an earlier launch of a batch in slices)
//   state     kPieceState bytes per state slot
//   pos[4i]   w0: position of the first byte the input buffer holds (in_off[i] points at it): the piece brings the
//             32 KiB window (and a margin) along; in_len[i] is the length of the input so far, counted like w0
//   pos[4i+1] rebase: positions are 32-bit, so a long stream's origin moves now and then - w0 and in_len count from the
//             new origin, and this (a multiple of 64 KiB, at most the window base) is what the positions in the state
//             the piece before left have to come down by
//   pos[4i+2] the stream's state slot, pos[4i+3] its command queue's slot (whole streams: queue i)
//   sum[2i]   the checksum of the whole input so far (Adler-32 / CRC-32; for gzip always the caller's), [2i+1] its length
//             mod 2^32 (the trailer of the last piece)
// status[i] = MD_PIECE_AWAIT when the piece ended with the matcher waiting for more input.
struct Piece {
  const uint32_t *flags;
  uint8_t *state;
  const uint64_t *pos;
  const uint32_t *sum;
};
constexpr uint32_t kPieceState = 12288;
constexpr int MD_PIECE_AWAIT = 1000;

// hash of the string at a, from its little-endian first 4 bytes:
//   De.Lz77  hash4, lib/de.ml:4067-4071: 4 bytes multiplied by 0x9e3779b1, top 15 bits;
//   Lz       update_hash (lib/lz.ml:153-155, shift 5, 15 bits) rolled over 3 bytes by fill_window's
//            priming (lib/lz.ml:399-401) + insert_string (lib/lz.ml:297-304) — every string that is
//            inserted at all is inserted right after its predecessor, so the rolled state is this
//            pure function of the 3 bytes.
__device__ __forceinline__ uint32_t hash_of(int matcher, uint32_t w4) {
  if (matcher == MD_MATCHER_LZ)
    return (((w4 & 0xff) << 10) ^ (((w4 >> 8) & 0xff) << 5) ^ ((w4 >> 16) & 0xff)) & (HASH_SIZE - 1);
  return (uint32_t)(w4 * 0x9e3779b1u) >> (32 - HASH_BITS);
}
// positions < p_end are inserted whatever the matcher decides: the last ones are not (lookahead < MIN_MATCH), and
// De's hash of position len - 3 reads a byte beyond the data (H7): that one is left to the sequential kernel
__device__ __forceinline__ uint32_t stream_p_end(int matcher, uint32_t eff_level, uint32_t len) {
  if (matcher == MD_MATCHER_LZ) return len >= 3 ? len - 2 : 0;
  return (eff_level != 0 && len >= 4) ? len - 3 : 0;
}
__device__ __forceinline__ uint32_t effective_level(int driver, int matcher, int level) {
  if (driver == 1 /* DRV_HIGHER */) return 4;               // H6: De.Higher.compress has no ?level
  if (matcher == MD_MATCHER_LZ && level < 4) return 4;      // Lz.state: 0..4 are _4 (lib/lz.ml:535)
  return (uint32_t)level;
}

__device__ __forceinline__ uint64_t lanes_below(uint32_t lane) { return (1ull << lane) - 1; }
// 16-bit fingerprint of the 3 bytes a candidate must share with the position (lib/de.ml:4133-4137): a candidate whose
// fingerprint differs cannot pass the test, so the walk reads one word per candidate and touches the text only for
// the few that may
__device__ __forceinline__ uint32_t fp16(uint32_t w4) { return ((w4 & 0xffffffu) * 0x9E3779B1u) >> 16; }

// length of the common prefix of a and b, at most MAX_MATCH: what longest_match's compare loop
// arrives at (lib/de.ml:4139-4155; its 4-byte steps land exactly on str_end).  Reads a[0..259].
__device__ __forceinline__ uint32_t lcp258(const uint8_t *a, const uint8_t *b) {
#pragma clang loop unroll(disable)
  for (uint32_t k = 0; k < 256; k += 4) {
    uint32_t x, y;
    __builtin_memcpy(&x, a + k, 4);
    __builtin_memcpy(&y, b + k, 4);
    if (x != y) return k + ((uint32_t)__builtin_ctz(x ^ y) >> 3);
  }
  if (a[256] != b[256]) return 256;
  return a[257] != b[257] ? 257 : 258;
}

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// common prefix of two 16-byte strings held in registers (0..16)
__device__ __forceinline__ uint32_t prefix16(v4u a, v4u b) {
  const uint64_t lo = (uint64_t)(a.x ^ b.x) | ((uint64_t)(a.y ^ b.y) << 32);
  const uint64_t hi = (uint64_t)(a.z ^ b.z) | ((uint64_t)(a.w ^ b.w) << 32);
  return lo ? (uint32_t)__builtin_ctzll(lo) >> 3 : hi ? 8u + ((uint32_t)__builtin_ctzll(hi) >> 3) : 16u;
}
// lcp258 for strings known to agree in their first 16 bytes
__device__ __forceinline__ uint32_t lcp258_from16(const uint8_t *a, const uint8_t *b) {
#pragma clang loop unroll(disable)
  for (uint32_t k = 16; k < 256; k += 8) {
    uint64_t x, y;
    __builtin_memcpy(&x, a + k, 8);
    __builtin_memcpy(&y, b + k, 8);
    if (x != y) return k + ((uint32_t)__builtin_ctzll(x ^ y) >> 3);
  }
  if (a[256] != b[256]) return 256;
  return a[257] != b[257] ? 257 : 258;
}

}  // namespace defl
}  // namespace md
