// inflate_chunked.hip — ONE long DEFLATE stream on the whole chip (round 6; VERDICT r5 "missing" 1).
//
// De.Higher.uncompress / Zl.Higher.uncompress / Gz on one big input (lib/de.ml:4555-4571, lib/zl.ml:650-666) is one
// serial chain of blocks for the reference, and was one pair of wavefronts here (126-160 MiB/s whatever the chip).  A
// DEFLATE stream can be decoded from any BLOCK START if the 32 KiB of output in front of it are known - and where they are
// not, the decode can still run with a PLACEHOLDER window and be corrected afterwards, because every output byte is either a
// literal of the stream or a copy (of a copy ...) of exactly one window byte.  So (capi.cpp `inflate_parallel` drives this):
//
//   1. find_blocks_kernel   the compressed body is cut into chunks of K bytes; for every chunk but the first, the first
//      bit position at or behind the chunk's start where a non-final dynamic block header parses completely (HLIT / HDIST /
//      HCLEN, a complete code-length code, lengths that decode without overrun, a complete literal/length code with an
//      end-of-block code, a complete or single-code distance code - De.Inf's own rules, lib/de.ml:1733-1793, :523-638,
//      made a little stricter) is a CANDIDATE block start; so is the byte behind an empty stored block (a flush marker).
//   2. the batch kernel (inflate_wave.hip, unchanged) decodes every piece [candidate c, candidate c + 1) as a stream of its
//      own (md_inflate_continue_batch_device: starting bit, 32 KiB of history in front of the output) - the first piece
//      with the real (empty) window straight into the output buffer, every other piece TWICE into scratch buffers, once
//      with each of two placeholder windows A and B (fill_window_kernel) built so that A[j] != B[j] for every window
//      position j and (A[j], B[j]) names j.  A piece must end - last complete block - exactly on the next candidate: the
//      chain of pieces from the true start of the stream is then the stream's own chain of blocks, and a false candidate
//      shows up as a piece that runs past it (the host drops it and decodes again).
//   3. window_chain_kernel  one workgroup walks the pieces in order and resolves the LAST 32 KiB of each: the window of
//      the piece behind it.  resolve_kernel then rewrites every piece in parallel: where the two decodes agree the byte is
//      a literal (or a copy of one), where they differ it is window byte j = A-byte | (B-byte >> 1) << 8.
//   4. adler_segments_kernel / the CRC-32 kernel give the checksum of the final bytes in segments that the host joins.
// Anything unexpected - a status the pieces should not give, a reference in front of the stream's start, a checksum that
// does not match - sends the caller back to the serial path, whose statuses and counts are the reference's: this path only
// ever returns a finished, verified stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdeflate.h"

namespace md {
namespace chunked {

constexpr uint32_t kWin = 32768;

// placeholder windows: A[j] = j & 255; B[j] = (j >> 8) << 1 | f with f chosen so that B[j] != A[j]
__host__ __device__ __forceinline__ uint32_t pat_a(uint32_t j) { return j & 255u; }
__host__ __device__ __forceinline__ uint32_t pat_b(uint32_t j) {
  const uint32_t a = j & 255u, h = (j >> 8) << 1;
  return h == (a & 0xfeu) ? h | (~a & 1u) : h;
}
// the decodes with window A and window B produced bytes a and b at some position: *j = the window byte it copies
__device__ __forceinline__ bool is_marker(uint32_t a, uint32_t b, uint32_t *j) {
  *j = a | ((b >> 1) << 8);
  return a != b;
}

// ---- 1. candidate block starts -----------------------------------------------------------------------------------------
struct BitRd {  // LSB-first bit reader over body[0, n), zeros beyond the end (`over` says so)
  const uint8_t *p;
  uint64_t n, byte;  // next byte to load
  uint64_t buf;
  uint32_t cnt;
  bool over;
  __device__ __forceinline__ void init(const uint8_t *body, uint64_t nbytes, uint64_t bit) {
    p = body;
    n = nbytes;
    byte = bit >> 3;
    buf = 0;
    cnt = 0;
    over = false;
    fill();
    drop((uint32_t)(bit & 7));
  }
  __device__ __forceinline__ void fill() {
    if (cnt <= 56 && byte + 8 <= n) {  // as many whole bytes as fit, from ONE unaligned load
      uint64_t w;
      __builtin_memcpy(&w, p + byte, 8);
      const uint32_t take = (64 - cnt) >> 3;  // 1 .. 8 bytes
      buf |= (take == 8 ? w : w & ((1ull << (8 * take)) - 1)) << cnt;
      byte += take;
      cnt += 8 * take;
      return;
    }
    while (cnt <= 56) {
      uint64_t b = 0;
      if (byte < n) b = p[byte];
      buf |= b << cnt;
      byte++;
      cnt += 8;
    }
  }
  __device__ __forceinline__ uint32_t peek(uint32_t k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
  __device__ __forceinline__ void drop(uint32_t k) {
    buf >>= k;
    cnt -= k;
  }
  __device__ __forceinline__ uint32_t take(uint32_t k) {
    if (cnt < k) fill();
    const uint32_t v = peek(k);
    drop(k);
    return v;
  }
  // bits consumed so far lie inside the input
  __device__ __forceinline__ bool inside() const { return byte * 8 - cnt <= n * 8; }
};

// a complete dynamic block header at bit q of the body?  (called for the few positions whose first 17 + 3 HCLEN bits pass)
// Everything the code-length code needs lives in packed 64-bit registers - lengths 3 bits a symbol, counts / first codes /
// offsets 8 bits a length, the symbols sorted by (length, symbol) 5 bits each: as arrays indexed at run time they were
// scratch memory, a round trip per access, and the finder took longer than the decode it prepares (8 MiB of stored text:
// 33 ms of 34).
constexpr uint32_t kLutCols = 128;  // threads of a workgroup that validate side by side (a column of the look-up table each)
__device__ __noinline__ bool header_parses(const uint8_t *body, uint64_t nbytes, uint64_t q, uint8_t *lut /* [128][kLutCols], this thread's column */) {
  BitRd r;
  r.init(body, nbytes, q);
  r.take(3);
  const uint32_t hlit = r.take(5) + 257, hdist = r.take(5) + 1, hclen = r.take(4) + 4;
  constexpr uint64_t kZigA = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                             5ull << 45 | 11ull << 50 | 4ull << 55;
  constexpr uint64_t kZigB = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
  uint64_t cl = 0;  // 3 bits per symbol
  for (uint32_t i = 0; i < hclen; i++) {
    const uint32_t sym = (uint32_t)((i < 12 ? kZigA >> (5 * i) : kZigB >> (5 * (i - 12))) & 31);
    cl |= (uint64_t)r.take(3) << (3 * sym);
  }
  uint64_t cnt = 0;  // 8 bits per length
  for (uint32_t sy = 0; sy < 19; sy++) {
    const uint32_t l = (uint32_t)(cl >> (3 * sy)) & 7;
    if (l) cnt += 1ull << (8 * l);
  }
  // a 7-bit look-up table of the code-length code in LDS (entry = symbol | length << 5; column of this thread, the rows
  // 256 bytes apart: conflict-free): a symbol of the 300 that follow is one look-up instead of up to seven compare steps
  {
    uint64_t next = 0;  // next canonical code per length, 8 bits each
    uint32_t code = 0;
    for (uint32_t l = 1; l < 8; l++) {
      code = (code + (uint32_t)((cnt >> (8 * (l - 1))) & 255)) << 1;
      next |= (uint64_t)(code & 255) << (8 * l);
    }
    for (uint32_t sy = 0; sy < 19; sy++) {
      const uint32_t l = (uint32_t)(cl >> (3 * sy)) & 7;
      if (!l) continue;
      const uint32_t c = (uint32_t)(next >> (8 * l)) & 255;
      next += 1ull << (8 * l);
      const uint32_t rev = __builtin_bitreverse32(c) >> (32 - l);  // the stream carries a code most significant bit first
      for (uint32_t k = rev; k < 128; k += 1u << l) lut[k * kLutCols] = (uint8_t)(sy | (l << 5));
    }
  }
  // the lengths of both alphabets, run-length coded (lib/de.ml:1291-1345): Kraft sums and the end-of-block code on the way
  uint32_t lsum = 0, dsum = 0, dcodes = 0, d1 = 0, prev = 0, i = 0;
  bool eob = false;
  const uint32_t total = hlit + hdist;
  while (i < total) {
    if (r.cnt < 16) r.fill();
    const uint32_t ent = lut[r.peek(7) * kLutCols];
    const uint32_t sym = ent & 31;
    r.drop(ent >> 5);
    uint32_t rep = 1, val = sym;
    if (sym == 16) {
      if (i == 0) return false;
      rep = 3 + r.take(2);
      val = prev;
    } else if (sym == 17) {
      rep = 3 + r.take(3);
      val = 0;
    } else if (sym == 18) {
      rep = 11 + r.take(7);
      val = 0;
    }
    if (i + rep > total) return false;
    if (val) {  // rep symbols of length val from symbol i on: how many fall into each alphabet
      const uint32_t nl = i < hlit ? (i + rep <= hlit ? rep : hlit - i) : 0u, nd = rep - nl;
      lsum += nl * (32768u >> val);
      dsum += nd * (32768u >> val);
      dcodes += nd;
      if (val == 1) d1 += nd;
      if (i <= 256 && 256 < i + rep) eob = true;
      if (lsum > 32768u || dsum > 32768u) return false;  // over-subscribed already
    }
    i += rep;
    prev = val;
  }
  if (!r.inside()) return false;
  if (!eob || lsum != 32768u) return false;
  return dsum == 32768u || dcodes == 0 || (dcodes == 1 && d1 == 1);
}

// chunk c of the grid covers body bytes [(c + 1) K, (c + 2) K): cand[c] = the first candidate bit in it, ~0 if none.
// A round takes kRoundTiles x 256 bit positions: every thread runs the cheap filter (header bits, a complete code-length
// code) over its positions and appends what passes to a list in LDS; then the list is VALIDATED with one position per
// thread.  (Validating inside the filter loop kept 63 lanes waiting for the one that had something to validate - one
// position in ~900 passes the filter, a validation is thousands of instructions: 4 ms for 3 MB of input, four times the
// decode it prepares.)  The first valid position of the first round that has one is the chunk's candidate.
constexpr uint32_t kRoundTiles = 64, kListCap = 1024, kPreCap = 8192;
// the filter at bit q: 1 = a dynamic header that deserves validation, 2 = the byte behind an EMPTY STORED BLOCK (00 00 ff ff:
// what Z_SYNC_FLUSH / Z_FULL_FLUSH leave, pigz between its blocks; its three header bits and padding lie in the byte in
// front: the block behind it starts here, byte-aligned - needs no second look), 0 = neither
struct Bits24 {  // the 8 bytes in front of byte `by` and the 16 from it on
  uint64_t pre, lo, hi;
};
__device__ __forceinline__ Bits24 header_bytes(const uint8_t *__restrict__ body, uint64_t nbytes, uint64_t by) {
  Bits24 v{0, 0, 0};
  if (by >= 8 && by + 16 <= nbytes) {
    __builtin_memcpy(&v.pre, body + by - 8, 8);
    __builtin_memcpy(&v.lo, body + by, 8);
    __builtin_memcpy(&v.hi, body + by + 8, 8);
  } else {
    for (uint32_t k = 0; k < 8 && k < by; k++) v.pre |= (uint64_t)body[by - 1 - k] << (8 * (7 - k));
    for (uint32_t k = 0; k < 16 && by + k < nbytes; k++) {
      if (k < 8) v.lo |= (uint64_t)body[by + k] << (8 * k);
      else v.hi |= (uint64_t)body[by + k] << (8 * (k - 8));
    }
  }
  return v;
}
// bit s (0 .. 7) of the byte the words were loaded for.  The filter in two steps: the header's first 17 bits (one position
// in nine passes), then the code-length code's completeness - up to 19 three-bit fields, ~250 instructions, which a
// wavefront paid in full for every column of positions while ONE of its lanes had passed the first step: the survivors go
// to a list first and the second step takes them one per thread (64 MiB of text: finder 2.2 -> see DESIGN 3b).
__device__ __forceinline__ uint32_t header_first(const Bits24 &w, uint64_t by, uint32_t s) {
  // pre holds bytes by - 8 .. by - 1 (byte by - 1 on top): 00 00 ff ff in front, and the top three bits of the byte before zero
  if (s == 0 && by >= 5 && (w.pre >> 32) == 0xffff0000ull && ((w.pre >> 24) & 0xe0) == 0) return 2;
  const uint32_t v = (uint32_t)((s ? (w.lo >> s) | (w.hi << (64 - s)) : w.lo));
  if ((v & 7) != 4) return 0;  // BFINAL = 0, BTYPE = 2
  const uint32_t hlit = (v >> 3) & 31, hdist = (v >> 8) & 31;
  return hlit <= 29 && hdist <= 29 ? 1u : 0u;
}
__device__ __forceinline__ bool header_precode_complete(const Bits24 &w, uint32_t s) {
  const uint64_t v0 = s ? (w.lo >> s) | (w.hi << (64 - s)) : w.lo, v1 = w.hi >> s;
  const uint32_t hclen = ((uint32_t)(v0 >> 13) & 15) + 4;
  uint32_t kraft = 0;  // the code-length code is complete (kind CODES, lib/de.ml:549-550)
  for (uint32_t i = 0; i < hclen; i++) {
    const uint32_t at = 17 + 3 * i;  // 17 .. 71
    const uint32_t l = (uint32_t)(at < 64 ? (v0 >> at) | (v1 << (64 - at)) : v1 >> (at - 64)) & 7;
    kraft += l ? 128u >> l : 0u;
  }
  return kraft == 128u;
}
// grid = (parts, chunks behind the first): workgroup (s, c) searches the s-th part of chunk c's bits and lowers cand[c]
// (set to ~0 before the launch) to its first candidate - the parts of a chunk run side by side, a small input has few chunks
__global__ __launch_bounds__(256) void find_blocks_kernel(const uint8_t *__restrict__ body, uint64_t nbytes, uint64_t K,
                                                          unsigned long long *__restrict__ cand) {
  __shared__ uint32_t found, nlist, npre;
  __shared__ uint32_t list[kListCap];
  __shared__ alignas(16) uint8_t cl_lut[128 * kLutCols];  // (one wavefront validates: the rest of the table was three quarters of the
                                                           // workgroup's LDS, and LDS is what limits the workgroups per CU here)
  // positions of the round (relative to its start) that passed the first step: in the validation's table space, which is
  // not in use while they are
  uint16_t *pre = reinterpret_cast<uint16_t *>(cl_lut);
  static_assert(kPreCap * 2 <= 128 * kLutCols, "the list fits the table");
  const uint64_t c = blockIdx.y;
  const uint32_t kSubRanges = gridDim.x;  // parts of a chunk side by side: only while the chunks alone would leave the chip empty
  const uint64_t part = ((K * 8 / kSubRanges) + 255) & ~(uint64_t)255;
  const uint64_t c0 = (c + 1) * K * 8, c1x = (c + 2) * K * 8, nbits = nbytes * 8, c1 = c1x < nbits ? c1x : nbits;
  const uint64_t b0 = c0 + blockIdx.x * part, b1y = blockIdx.x + 1 == kSubRanges ? c1 : b0 + part, b1 = b1y < c1 ? b1y : c1;
  if (b0 >= b1) return;
  if (threadIdx.x == 0) {
    found = 0xffffffffu;
    nlist = 0;
    npre = 0;
  }
  __syncthreads();
  // a position that is complete goes on the list of the round's candidates
  auto to_list = [&](uint64_t q) {
    const uint32_t at = atomicAdd(&nlist, 1u);
    if (at < kListCap) list[at] = (uint32_t)(q - b0);
    else atomicMin(&found, 0xfffffffeu);  // (cannot be: a list of a thousand in 16 K positions; the part has no candidate then)
  };
  for (uint64_t r0 = b0; r0 < b1; r0 += (uint64_t)kRoundTiles * 256) {
    if (blockIdx.x && r0 > b0 && __hip_atomic_load(&cand[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < b0) break;  // an earlier part has one
    // a thread takes the eight bit positions of ONE byte from one set of loads (a load per position was a memory round trip
    // per 256 positions: 3 ms for 25 MB), the loads of the next byte column on their way meanwhile
    Bits24 nx = header_bytes(body, nbytes, (r0 >> 3) + threadIdx.x);
    for (uint32_t j = 0; j < kRoundTiles / 8; j++) {
      const uint64_t by = (r0 >> 3) + (uint64_t)j * 256 + threadIdx.x;
      const Bits24 w = nx;
      if (j + 1 < kRoundTiles / 8) nx = header_bytes(body, nbytes, by + 256);
#pragma unroll 1
      for (uint32_t sb = 0; sb < 8; sb++) {
        const uint64_t q = by * 8 + sb;
        const uint32_t f = q + 3 + 14 + 12 <= c1 && q < b1 ? header_first(w, by, sb) : 0u;
        if (f == 2) atomicMin(&found, (uint32_t)(q - b0));
        {  // survivors of the first step: one LDS atomic per wavefront and column, not one per survivor
          const uint64_t m = __ballot(f == 1);
          if (m) {
            const uint32_t lane = threadIdx.x & 63u, cnt = (uint32_t)__popcll(m);
            uint32_t base = 0;
            if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&npre, cnt);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
            if (f == 1) {
              const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
              if (at < kPreCap) pre[at] = (uint16_t)(q - r0);
              else if (header_precode_complete(w, sb)) to_list(q);  // (a quarter of the round: checked where it stands)
            }
          }
        }
      }
    }
    __syncthreads();
    {
      const uint32_t n1 = npre < kPreCap ? npre : kPreCap;
      for (uint32_t k = threadIdx.x; k < n1; k += 256) {
        const uint64_t q = r0 + pre[k];
        if (header_precode_complete(header_bytes(body, nbytes, q >> 3), (uint32_t)(q & 7))) to_list(q);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) npre = 0;
    const uint32_t n = nlist < kListCap ? nlist : kListCap;
    if (threadIdx.x < kLutCols)
      for (uint32_t k = threadIdx.x; k < n; k += kLutCols) {
        const uint32_t rel = list[k];
        if (rel < found && header_parses(body, nbytes, b0 + rel, cl_lut + threadIdx.x)) atomicMin(&found, rel);
      }
    __syncthreads();
    const uint32_t f = found;
    __syncthreads();
    if (threadIdx.x == 0) nlist = 0;
    if (f != 0xffffffffu) break;
    __syncthreads();
  }
  if (threadIdx.x == 0 && found < 0xfffffffeu) atomicMin(&cand[c], (unsigned long long)(b0 + found));
}

// ---- 2. placeholder windows in front of the pieces' scratch outputs -------------------------------------------------
// variant[i]: 0 = leave alone (the first piece: its window is real), 1 = window A, 2 = window B
__global__ __launch_bounds__(256) void fill_window_kernel(uint32_t n, uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
                                                          const uint8_t *__restrict__ variant) {
  const uint32_t i = blockIdx.y;
  if (i >= n || variant[i] == 0) return;
  uint8_t *o = out + out_off[i];
  const bool b = variant[i] == 2;
  for (uint32_t j = (blockIdx.x * 256 + threadIdx.x) * 4; j < kWin; j += gridDim.x * 256 * 4) {
    uint32_t w = 0;
    for (uint32_t k = 0; k < 4; k++) w |= (b ? pat_b(j + k) : pat_a(j + k)) << (8 * k);
    __builtin_memcpy(o + j, &w, 4);
  }
}

// ---- 3. windows, then every byte ------------------------------------------------------------------------------------------
// Piece p (p >= 1): decoded bytes A at scratch + offa[p], B at scratch + offb[p], u[p] of them, final place dst + pos[p].
// Piece 0 is final already: dst[0, u[0]).  wins[p] = the 32 KiB in front of piece p (right-aligned: wins[p][kWin - 1] is the
// byte just before it), valid[p] of them real.  flag[0] != 0: a reference in front of the stream's start (the serial path
// reports Invalid_distance for it).
// Thread t owns window bytes [32 t, 32 t + 32).  The pieces are a serial chain, but only through the window in LDS: the
// tails of piece p + 1's two decodes are loaded (two 16-byte loads each) while piece p is resolved - the first version read
// them byte by byte when it got there, a memory round trip per piece and byte column: 13 us a piece, 5 ms of the 11 a
// 64 MiB stream took.
struct TailRegs {
  uint32_t a[8], b[8];
  uint64_t up;
};
struct PieceDesc {
  uint64_t offa, offb, up;
};
__device__ __forceinline__ PieceDesc desc_load(uint32_t p, uint32_t npieces, const uint64_t *__restrict__ offa, const uint64_t *__restrict__ offb,
                                               const uint64_t *__restrict__ u) {
  PieceDesc d{0, 0, 0};
  if (p < npieces) {
    d.offa = offa[p];
    d.offb = offb[p];
    d.up = u[p];
  }
  return d;
}
__device__ __forceinline__ void tail_load(TailRegs &r, const PieceDesc &d, const uint8_t *__restrict__ scratch, uint32_t j0) {
#pragma unroll
  for (int k = 0; k < 8; k++) r.a[k] = r.b[k] = 0;
  r.up = d.up;
  const uint8_t *a = scratch + d.offa, *b = scratch + d.offb;
  // window byte j is piece byte up - (kWin - j); a piece shorter than the window is not fetched ahead (the loop takes it
  // byte by byte: rare, and its registers would be indexed at run time)
  if (r.up >= kWin) {
    const uint64_t i = r.up - (kWin - j0);
    __builtin_memcpy(r.a, a + i, 32);
    __builtin_memcpy(r.b, b + i, 32);
  }
}
// 512 threads, two 32-byte segments of the window each (bytes 32 t .. and 16 K + 32 t ..); the tails of the pieces are
// loaded kAhead pieces ahead - a load issued one piece ahead still left 6 us a piece, the memory's latency, not the work
constexpr int kAhead = 3, kSegs = 2;
constexpr uint32_t kChainThreads = 512;
__global__ __launch_bounds__(kChainThreads) void window_chain_kernel(uint32_t npieces, const uint8_t *__restrict__ dst, const uint8_t *__restrict__ scratch,
                                                                     const uint64_t *__restrict__ offa, const uint64_t *__restrict__ offb,
                                                                     const uint64_t *__restrict__ u, uint8_t *__restrict__ wins,
                                                                     uint32_t *__restrict__ flag) {
  __shared__ __attribute__((aligned(16))) uint8_t w[2][kWin];
  const uint32_t t = threadIdx.x;
  uint32_t cur = 0;
  uint64_t valid = u[0] < kWin ? u[0] : kWin;
  for (uint32_t j = t; j < kWin; j += kChainThreads) w[0][j] = j >= kWin - valid ? dst[u[0] - (kWin - j)] : 0;
  TailRegs r[kAhead][kSegs];  // r[0]: the piece at hand, r[1]: the next one, ...
#pragma unroll
  for (int a = 0; a < kAhead; a++) {
    const PieceDesc d = desc_load(1 + a, npieces, offa, offb, u);
#pragma unroll
    for (int sg = 0; sg < kSegs; sg++) tail_load(r[a][sg], d, scratch, 32 * t + sg * (kWin / kSegs));
  }
  PieceDesc dn = desc_load(1 + kAhead, npieces, offa, offb, u);
  __syncthreads();
  bool bad = false;
  for (uint32_t p = 1; p < npieces; p++) {
    TailRegs now[kSegs];
#pragma unroll
    for (int sg = 0; sg < kSegs; sg++) now[sg] = r[0][sg];
#pragma unroll
    for (int a = 0; a + 1 < kAhead; a++)
#pragma unroll
      for (int sg = 0; sg < kSegs; sg++) r[a][sg] = r[a + 1][sg];
#pragma unroll
    for (int sg = 0; sg < kSegs; sg++) tail_load(r[kAhead - 1][sg], dn, scratch, 32 * t + sg * (kWin / kSegs));  // piece p + kAhead
    dn = desc_load(p + 1 + kAhead, npieces, offa, offb, u);
    const uint64_t up = now[0].up;
#pragma unroll
    for (int sg = 0; sg < kSegs; sg++) {  // the window in front of piece p, for resolve_kernel (behind the loads: waiting for them must not wait for this)
      const uint32_t j0 = 32 * t + sg * (kWin / kSegs);
      const uint4 *src = reinterpret_cast<const uint4 *>(&w[cur][j0]);
      uint4 *d = reinterpret_cast<uint4 *>(wins + (uint64_t)p * kWin + j0);
      d[0] = src[0];
      d[1] = src[1];
    }
    if (up >= kWin) {  // (wave-uniform) the piece's last 32 KiB are the next window
#pragma unroll
      for (int sg = 0; sg < kSegs; sg++) {
        const uint32_t j0 = 32 * t + sg * (kWin / kSegs);
        uint32_t out[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          if (now[sg].a[q] == now[sg].b[q]) {  // four bytes, none of them from the window: the usual case
            out[q] = now[sg].a[q];
            continue;
          }
          uint32_t word = 0;
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            uint32_t idx, byte = (now[sg].a[q] >> (8 * kk)) & 255;
            if (is_marker(byte, (now[sg].b[q] >> (8 * kk)) & 255, &idx)) {
              if (idx < kWin - valid) bad = true;
              byte = w[cur][idx];
            }
            word |= byte << (8 * kk);
          }
          out[q] = word;
        }
        uint4 *o = reinterpret_cast<uint4 *>(&w[cur ^ 1][j0]);
        o[0] = make_uint4(out[0], out[1], out[2], out[3]);
        o[1] = make_uint4(out[4], out[5], out[6], out[7]);
      }
    } else {  // a piece shorter than the window: the old window slides, the piece's bytes come straight from memory
      const PieceDesc dd = desc_load(p, npieces, offa, offb, u);
      const uint8_t *a = scratch + dd.offa, *b = scratch + dd.offb;
      for (uint32_t j = t; j < kWin; j += kChainThreads) {
        uint32_t byte;
        if (j < kWin - up) byte = w[cur][j + up];
        else {
          const uint64_t i = up - (kWin - j);
          uint32_t idx;
          byte = a[i];
          if (is_marker(byte, b[i], &idx)) {
            if (idx < kWin - valid) bad = true;
            byte = w[cur][idx];
          }
        }
        w[cur ^ 1][j] = (uint8_t)byte;
      }
    }
    valid = valid + up < kWin ? valid + up : kWin;
    cur ^= 1;
    __syncthreads();
  }
  if (bad) atomicOr(flag, 1u);
}

// ---- the windows in PARALLEL (pointer jumping) -------------------------------------------------------------------------
// The chain above is one workgroup: with word text two bytes in five of a piece's last 32 KiB still descend from the window
// in front of it (a frequent word is a copy of a copy of ... its first occurrence), so a piece is 13 000 dependent LDS
// look-ups and the chain 6 us a piece whatever is fetched ahead (64 MiB: 2.4 ms of 7.8).  The same thing as a table: T[q][j] =
// window byte j behind piece g0 + q - 1 (T[0]: the window in front of the group) is either a LITERAL byte or a REFERENCE
// (t, i) "the same as T[t][i]" with t < q.  One round replaces every reference by what it points at; after ceil(log2(pieces))
// rounds every entry is a literal - every round one launch over all entries, the whole chip.
constexpr uint32_t kLit = 0x80000000u, kInvalid = 0x40000000u;
__global__ __launch_bounds__(256) void win_first_kernel(const uint8_t *__restrict__ dst, uint64_t u0, uint8_t *__restrict__ win1) {
  const uint64_t valid = u0 < kWin ? u0 : kWin;
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < kWin; j += gridDim.x * 256) win1[j] = j >= kWin - valid ? dst[u0 - (kWin - j)] : 0;
}
// grid (x, cnt + 1): row q of the table of the group of pieces g0 .. g0 + cnt - 1
__global__ __launch_bounds__(256) void jump_init_kernel(uint32_t g0, const uint8_t *__restrict__ scratch, const uint64_t *__restrict__ offa,
                                                        const uint64_t *__restrict__ offb, const uint64_t *__restrict__ u,
                                                        const uint64_t *__restrict__ pos, const uint8_t *__restrict__ wins, uint32_t *__restrict__ T) {
  const uint32_t q = blockIdx.y;
  uint32_t *row = T + (uint64_t)q * kWin;
  if (q == 0) {  // the window in front of the group: bytes, of which the stream's first pos[g0] are real
    const uint64_t valid = pos[g0] < kWin ? pos[g0] : kWin;
    const uint8_t *w = wins + (uint64_t)g0 * kWin;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < kWin; j += gridDim.x * 256) row[j] = kLit | (j < kWin - valid ? kInvalid : 0u) | w[j];
    return;
  }
  const uint32_t p = g0 + q - 1;
  const uint64_t up = u[p], take = up < kWin ? up : kWin;
  const uint8_t *a = scratch + offa[p], *b = scratch + offb[p];
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < kWin; j += gridDim.x * 256) {
    uint32_t e;
    if (j < kWin - take) e = ((q - 1) << 15) | (uint32_t)(j + take);  // (a piece shorter than the window: the old one slides)
    else {
      const uint64_t i = up - (kWin - j);
      uint32_t idx;
      const uint32_t x = a[i];
      e = is_marker(x, b[i], &idx) ? ((q - 1) << 15) | idx : kLit | x;
    }
    row[j] = e;
  }
}
__global__ __launch_bounds__(256) void jump_round_kernel(uint64_t entries, const uint32_t *__restrict__ Tin, uint32_t *__restrict__ Tout) {
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < entries; k += (uint64_t)gridDim.x * 256) {
    uint32_t e = Tin[k];
    if (!(e & kLit)) e = Tin[(uint64_t)(e >> 15) * kWin + (e & (kWin - 1))];
    Tout[k] = e;
  }
}
// rows 1 .. cnt of the table are the windows in front of pieces g0 + 1 .. g0 + cnt (those that exist)
__global__ __launch_bounds__(256) void jump_final_kernel(uint32_t g0, uint32_t npieces, const uint32_t *__restrict__ T, const uint64_t *__restrict__ pos,
                                                         uint8_t *__restrict__ wins, uint32_t *__restrict__ flag) {
  const uint32_t q = blockIdx.y + 1, p1 = g0 + q;  // the window in front of piece p1
  if (p1 >= npieces) return;
  const uint32_t *row = T + (uint64_t)q * kWin;
  const uint64_t valid = pos[p1] < kWin ? pos[p1] : kWin;
  uint8_t *w = wins + (uint64_t)p1 * kWin;
  bool bad = false;
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < kWin; j += gridDim.x * 256) {
    const uint32_t e = row[j];
    bad = bad || !(e & kLit) || ((e & kInvalid) && j >= kWin - valid);  // a byte of the stream that copies what lies in front of its start
    w[j] = (uint8_t)e;
  }
  if (bad) atomicOr(flag, 1u);
}

// grid.y = piece - 1, grid.x strides over the piece in steps of 4 KiB per workgroup
__global__ __launch_bounds__(256) void resolve_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ scratch,
                                                      const uint64_t *__restrict__ offa, const uint64_t *__restrict__ offb,
                                                      const uint64_t *__restrict__ u, const uint64_t *__restrict__ pos,
                                                      const uint8_t *__restrict__ wins, uint32_t *__restrict__ flag) {
  const uint32_t p = blockIdx.y + 1;
  const uint64_t up = u[p];
  const uint8_t *a = scratch + offa[p], *b = scratch + offb[p], *w = wins + (uint64_t)p * kWin;
  uint8_t *o = dst + pos[p];
  const uint64_t lim = pos[p] < kWin ? kWin - pos[p] : 0;  // window positions below this lie in front of the stream
  bool bad = false;
  for (uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; i0 < up; i0 += (uint64_t)gridDim.x * 256 * 16) {
    if (i0 + 16 <= up) {
      uint4 va, vb;
      __builtin_memcpy(&va, a + i0, 16);
      __builtin_memcpy(&vb, b + i0, 16);
      uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (wa[q] == wb[q]) continue;
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint32_t x = (wa[q] >> (8 * k)) & 255, j;
          if (is_marker(x, (wb[q] >> (8 * k)) & 255, &j)) {
            bad = bad || j < lim;
            x = w[j];
          }
          r |= x << (8 * k);
        }
        wa[q] = r;
      }
      const uint4 vo = make_uint4(wa[0], wa[1], wa[2], wa[3]);
      __builtin_memcpy(o + i0, &vo, 16);
    } else {
      for (uint64_t i = i0; i < up; i++) {
        uint32_t x = a[i], j;
        if (is_marker(x, b[i], &j)) {
          bad = bad || j < lim;
          x = w[j];
        }
        o[i] = (uint8_t)x;
      }
    }
  }
  if (bad) atomicOr(flag, 1u);
}

// ---- 4. Adler-32 in segments ----------------------------------------------------------------------------------------------
// segment s = data[s * seg, min((s + 1) * seg, n)): sums[2 s] = sum of its bytes, sums[2 s + 1] = sum of (len - i) * byte i,
// both mod 65521 (the host joins them: b += len * a + S2, a += S1; lib/de.ml:453-455's update over the whole output)
__global__ __launch_bounds__(256) void adler_segments_kernel(const uint8_t *__restrict__ data, uint64_t n, uint32_t seg,
                                                             uint32_t *__restrict__ sums) {
  __shared__ uint32_t sh1[256], sh2[256];
  const uint64_t s0 = (uint64_t)blockIdx.x * seg, s1 = s0 + seg < n ? s0 + seg : n;
  const uint32_t len = (uint32_t)(s1 - s0);
  // thread t takes the contiguous slice [t * per, (t + 1) * per) of the segment, per a multiple of 16
  const uint32_t per = ((len + 255) / 256 + 15) & ~15u;
  const uint32_t a0 = threadIdx.x * per, a1 = a0 + per < len ? a0 + per : len;
  uint32_t t1 = 0, t2 = 0;  // of the slice: sum d, sum (slice_len - i) d_i
  if (a0 < len) {
    const uint8_t *p = data + s0;
    const uint32_t sl = a1 - a0;
    uint32_t i = a0;
    for (; i + 16 <= a1; i += 16) {
      uint4 v;
      __builtin_memcpy(&v, p + i, 16);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t c1 = 0, ck = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        c1 = __builtin_amdgcn_udot4(w[q], 0x01010101u, c1, false);
        ck = __builtin_amdgcn_udot4(w[q], 0x03020100u + 0x04040404u * q, ck, false);
      }
      t1 += c1;
      t2 = (t2 + (sl - (i - a0)) * c1 - ck) % 65521u;  // (<= 16 x 255 x 2^20 per step: no overflow before the reduction)
    }
    for (; i < a1; i++) {
      const uint32_t d = p[i];
      t1 += d;
      t2 = (t2 + (sl - (i - a0)) * d) % 65521u;
    }
    // the slice seen from the segment: every byte counts (len - a1) more times
    t2 = (t2 + (uint64_t)(len - a1) % 65521u * (t1 % 65521u)) % 65521u;
    t1 %= 65521u;
  }
  sh1[threadIdx.x] = t1;
  sh2[threadIdx.x] = t2;
  __syncthreads();
  for (uint32_t o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh1[threadIdx.x] = (sh1[threadIdx.x] + sh1[threadIdx.x + o]) % 65521u;
      sh2[threadIdx.x] = (sh2[threadIdx.x] + sh2[threadIdx.x + o]) % 65521u;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[2 * blockIdx.x] = sh1[0];
    sums[2 * blockIdx.x + 1] = sh2[0];
  }
}

}  // namespace chunked
}  // namespace md

extern "C" int md_launch_find_blocks(const uint8_t *body, uint64_t nbytes, uint64_t K, uint32_t nchunks_behind_first, uint64_t *cand,
                                     hipStream_t stream) {
  if (nchunks_behind_first == 0) return 0;
  if (hipMemsetAsync(cand, 0xff, (size_t)nchunks_behind_first * 8, stream) != hipSuccess) return (int)hipGetLastError();
  const uint32_t parts = nchunks_behind_first >= 1024 ? 1u : nchunks_behind_first >= 512 ? 2u : 4u;
  hipLaunchKernelGGL(md::chunked::find_blocks_kernel, dim3(parts, nchunks_behind_first), dim3(256), 0, stream, body, nbytes, K,
                     (unsigned long long *)cand);
  return (int)hipGetLastError();
}
extern "C" int md_launch_fill_windows(uint32_t n, uint8_t *out, const uint64_t *out_off, const uint8_t *variant, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::chunked::fill_window_kernel, dim3(4, n), dim3(256), 0, stream, n, out, out_off, variant);
  return (int)hipGetLastError();
}
extern "C" int md_launch_window_chain(uint32_t npieces, const uint8_t *dst, const uint8_t *scratch, const uint64_t *offa,
                                      const uint64_t *offb, const uint64_t *u, uint8_t *wins, uint32_t *flag, hipStream_t stream) {
  hipLaunchKernelGGL(md::chunked::window_chain_kernel, dim3(1), dim3(md::chunked::kChainThreads), 0, stream, npieces, dst, scratch, offa, offb, u, wins, flag);
  return (int)hipGetLastError();
}
// the same windows by pointer jumping, in groups of at most `group` pieces (work: 2 x (group + 1) x 32 Ki 32-bit entries);
// u0 = u[0] (the host knows it), pos = the pieces' places in the output
extern "C" size_t md_windows_work_bytes(uint32_t npieces, uint32_t group) {
  const size_t g = npieces - 1 < group ? npieces - 1 : group;
  return 2 * (g + 1) * (size_t)md::chunked::kWin * 4;
}
extern "C" int md_launch_windows_parallel(uint32_t npieces, uint32_t group, const uint8_t *dst, uint64_t u0, const uint8_t *scratch, const uint64_t *offa,
                                          const uint64_t *offb, const uint64_t *u, const uint64_t *pos, uint8_t *wins, uint32_t *work, uint32_t *flag,
                                          hipStream_t stream) {
  using namespace md::chunked;
  if (npieces < 2) return 0;
  hipLaunchKernelGGL(win_first_kernel, dim3(16), dim3(256), 0, stream, dst, u0, wins + kWin);
  const size_t gmax = npieces - 1 < group ? npieces - 1 : group;
  uint32_t *T0 = work, *T1 = work + (gmax + 1) * (size_t)kWin;
  for (uint32_t g0 = 1; g0 < npieces; g0 += group) {
    const uint32_t cnt = npieces - g0 < group ? npieces - g0 : group;  // pieces g0 .. g0 + cnt - 1
    hipLaunchKernelGGL(jump_init_kernel, dim3(8, cnt + 1), dim3(256), 0, stream, g0, scratch, offa, offb, u, pos, wins, T0);
    uint32_t *in = T0, *out = T1;
    const uint64_t entries = (uint64_t)(cnt + 1) * kWin;
    for (uint32_t span = 1; span < cnt + 1; span *= 2) {
      hipLaunchKernelGGL(jump_round_kernel, dim3((uint32_t)((entries + 256 * 8 - 1) / (256 * 8))), dim3(256), 0, stream, entries, in, out);
      uint32_t *t = in;
      in = out;
      out = t;
    }
    hipLaunchKernelGGL(jump_final_kernel, dim3(8, cnt), dim3(256), 0, stream, g0, npieces, in, pos, wins, flag);
  }
  return (int)hipGetLastError();
}
extern "C" int md_launch_resolve(uint32_t npieces, uint8_t *dst, const uint8_t *scratch, const uint64_t *offa, const uint64_t *offb,
                                 const uint64_t *u, const uint64_t *pos, const uint8_t *wins, uint32_t *flag, hipStream_t stream) {
  if (npieces < 2) return 0;
  hipLaunchKernelGGL(md::chunked::resolve_kernel, dim3(32, npieces - 1), dim3(256), 0, stream, dst, scratch, offa, offb, u, pos, wins, flag);
  return (int)hipGetLastError();
}
extern "C" int md_launch_adler_segments(const uint8_t *data, uint64_t n, uint32_t seg, uint32_t *sums, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::chunked::adler_segments_kernel, dim3((uint32_t)((n + seg - 1) / seg)), dim3(256), 0, stream, data, n, seg, sums);
  return (int)hipGetLastError();
}
