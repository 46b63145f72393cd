// deflate_kernel.hip — batched RFC1951 deflate for gfx950 (MI355X): the sequential kernel.
//
// Bit-exact with the reference's streaming compressor:
//   De.Lz77 (zlib deflate_slow, 4-byte multiplicative hash)      lib/de.ml:4013-4515
//   De.Queue command encoding                                    lib/de.ml:2245-2266
//   De.T     Huffman tree construction                           lib/de.ml:1828-2192
//   De.Def   block choice + LSB-first bit encoder                 lib/de.ml:2354-3038
//   drivers  Zl.Def (lib/zl.ml:509-555), De.Higher (lib/de.ml:4518-4553), CLI (bin/decompress.ml:47-75)
// including the parity hazards H1-H8 of SURVEY.md 8(c) (queue-sized blocks,
// cumulative and mutated histograms, the odd cost formula, driver-dependent
// empty blocks, deterministic stale bytes past the end of input).
//
// Layout: one independent stream per wavefront.  The lazy matcher and the driver
// of a stream are sequential state machines (every decision depends on the
// previous one): lane 0 runs them.  What does NOT depend on decisions has been computed
// for the whole batch by deflate_front.hip before this kernel starts: link[p] (the hash
// chains: hash_head(p) and every chain link are pure functions of the input) and, per
// position, what longest_match finds there (flg / m / mq).  This kernel streams those
// arrays into an LDS ring, takes literal runs 64 positions at a time, parses the lazy
// evaluation from the verdicts, builds the Huffman trees (all but zlib's heap merge loop)
// and packs the bits of a queue fill 64 commands per step.  The command queue lives in a
// per-stream HBM workspace; histograms, heap and code tables live in LDS.
// The window is the input buffer itself: w[rel] = in[base + rel]; bytes the
// reference reads beyond the data (H7) follow its 64 KiB sliding buffer exactly
// (zero before the first slide, the byte 32 KiB earlier after it).  See DESIGN.md 4.
#include <string.h>
#include "deflate_common.hpp"

namespace md {
namespace defl {

constexpr int MAX_BITS = 15, L_CODES = 286, D_CODES = 30, BL_CODES = 19, HEAP_SIZE = 2 * L_CODES + 1;
constexpr int Q_EOB = 256, Q_COPY = 0x2000000;

__constant__ uint8_t c_zigzag[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
__constant__ uint8_t c_base_length[32] = {0,  1,  2,  3,  4,  5,  6,   7,   8,   10,  12,
                                          14, 16, 20, 24, 28, 32, 40,  48,  56,  64,  80,
                                          96, 112, 128, 160, 192, 224, 255, 0,   0,   0};
__constant__ uint8_t c_extra_lbits[32] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                          3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0};
__constant__ uint8_t c_extra_dbits[32] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,  4,
                                          4, 5, 5, 6, 6, 7, 7,  8,  8,  9,  9,
                                          10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
__constant__ uint16_t c_base_dist[32] = {0,    1,    2,    3,    4,    6,     8,     12,
                                         16,   24,   32,   48,   64,   96,    128,   192,
                                         256,  384,  512,  768,  1024, 1536,  2048,  3072,
                                         4096, 6144, 8192, 12288, 16384, 24576, 0,   0};
// level table lib/de.ml:4030-4049: {max_chain, max_lazy, good_length, nice_length}
__constant__ uint16_t c_levels[10][4] = {{0, 0, 0, 0},         {4, 4, 4, 8},        {8, 5, 4, 16},
                                         {32, 6, 4, 32},       {16, 4, 4, 16},      {32, 16, 8, 32},
                                         {128, 16, 8, 128},    {256, 32, 8, 128},   {1024, 128, 32, 258},
                                         {4096, 258, 32, 258}};

enum { KIND_FLAT = 0, KIND_FIXED = 1, KIND_DYNAMIC = 2 };
enum { DRV_ZL = 0, DRV_HIGHER = 1, DRV_CLI = 2,
       DRV_LZ77 = 3,     // De.Lz77.compress alone: the queue fills are written out as commands (md_de_lz77_compress)
       DRV_ENCODE = 4,   // De.Def.encode alone: the input IS the command list of one last block (md_de_def_encode)
       DRV_SCRIPT = 5 }; // De.Def.encode driven step by step: the input is a list of operations (md_de_def_run)
// operations of DRV_SCRIPT, 32-bit words (mdeflate.h MD_OP_*)
enum { OP_FILL = 1, OP_BLOCK = 2, OP_FLUSH = 3, OP_SUCC_LITERAL = 4, OP_SUCC_LENGTH = 5, OP_SUCC_DISTANCE = 6, OP_NEW_FREQS = 7,
       OP_QUEUE_RESET = 8 };
enum { K_FIRST_ENTRY, K_ENCODE, K_BLOCK, K_FLAT_DONE };
enum { R_OK, R_BLOCK };
enum { V_AWAIT, V_FLUSH, V_BLOCK };
enum { LZ_FLUSH, LZ_END, LZ_NEED, LZ_AWAIT };
enum { LK_ENOUGH, LK_FILL };

template <int N>
struct TreeT {  // one Huffman tree as the encoder needs it
  uint16_t lengths[N + 2];  // T.tree.lengths (patched by T.scan's 0xffff guard)
  uint16_t codes[N + 2];
  uint8_t clen[N + 2];      // Lookup lengths as of T.make
  int max_code;
};
struct TreeRef {
  uint16_t *lengths, *codes;
  uint8_t *clen;
  int *max_code;
};
template <int N>
__device__ __forceinline__ TreeRef tref(TreeT<N> *t) {
  return TreeRef{t->lengths, t->codes, t->clen, &t->max_code};
}

constexpr uint32_t RING = 1024;
constexpr int PG = 8;    // steps of 64 positions loaded into the ring together

// LDS of one stream: 9.6 KiB, so that 16 streams share a CU (4 096 streams = one generation on 256 CUs).  What is
// live only while the trees of a block are built (heap, parent links, lengths) shares its space with what is live
// only while the matcher runs (the ring of look-ahead inputs, the parse window): building trees kills the ring, and
// it is loaded again from the front workspace before the matcher goes on (`ring_dead`).
struct DS {
  int lits[L_CODES];     // live literal/length histogram (make_literals, lib/de.ml:2333), mutated by T.make's pkzip rule
  int dsts[D_CODES];
  int blf[BL_CODES];
  TreeT<L_CODES> lt;           // trees of the CURRENT block (e.blk)
  TreeT<D_CODES> dt;
  TreeT<BL_CODES> bt;
  int kind_result;             // block kind chosen by trees_wave
  uint32_t tp[8];              // profile of trees_wave (ticks), only when profiling
  int nsymbols, h_lit, h_dst, h_len;
  uint32_t ctl[5];       // [0] machine strstart, [1] machine state (1 = finished), [2] prepared_end, [3] action,
                         // [4] tree mode of ACT_TREES
  uint32_t ring_dead;    // the ring's space was used by the tree builder: load it again from strstart - 1 on
  // wave-parallel bit packing of one queue fill (enc_write_wave)
  uint8_t t_xl[32], t_bl[32], t_xd[32];
  uint16_t t_bd[32];
  struct ZS {            // matcher state the wave needs for bulk literal runs (lane 0 <-> wave)
    uint32_t strstart, lookahead, base, trivial, qw, qr, bulked;
    // trivial: 0 = the matcher must run; 1 = literal-run state (match_available, match_length 2);
    //          2 = fresh state (no pending literal, match_length 2: right after a match)
  } zs;
  struct WCtl {
    uint64_t hold;
    uint32_t bits, o_pos, o_cap, qr, qw, qc, kind, last, overflow, rc, k, hdr;
    uint8_t *o;
    int *q;
  } w;
  union alignas(16) {
    struct {  // while trees are built (T.make)
      alignas(16) uint64_t hk[L_CODES + 2];  // the heap as keys (freq << 32 | depth << 16 | node), 1-based
      uint16_t heap[HEAP_SIZE];              // its sorted tail: nodes in the order they left the heap
      uint16_t dads[HEAP_SIZE];
      uint16_t tlen[HEAP_SIZE];  // tree_lengths of the tree being built (leaves and internal nodes)
      int bl_count[MAX_BITS + 1];
    };
    // from the end of trees_wave to the header's bit packing: (len << 8) | code of the code-length stream, over the
    // heap (dead by then)
    uint16_t symbols[L_CODES + D_CODES + 8];
    struct {  // while the matcher runs
      // ring of decision-independent matcher inputs, indexed by position & (RING-1)
      uint16_t hl[RING];     // link[p] = p - hash_head(p), 0 = none
      uint8_t flg[RING];     // look-ahead verdict of p (FL_*)
      uint8_t byt[RING];     // the byte at p (pending literal of the next position)
      // window of the parse step: what the look-ahead knows about positions s0 .. s0 + 79
      uint16_t pm_lf[80], pm_df[80], pm_lq[80], pm_dq[80];  // longest_match ahead: full / quartered chain
      uint8_t pm_k[80];      // bit 0: verdict known, bit 1: hash_head valid (lib/de.ml:4365-4369)
      // bit buffer of one 128-command packing step (<= 15 + 128*48 bits): behind the ring (a queue fill is packed
      // while the ring is alive) and behind `symbols` (a header is packed from them)
      uint32_t bb[200];
    };
  };
};
static_assert(sizeof(DS) <= 10240 - 320, "16 streams per CU (with the lane-0 state next to it)");

// _length (lib/de.ml:240-256) of a match length 3..258 and _distance (lib/de.ml:258-291) of distance - 1, by arithmetic:
// beyond the first codes a code is two bits (one bit) under the leading one of the value
__device__ __forceinline__ int length_code_of(int len) {
  const int l = len - 3;
  if (l < 8) return l;
  if (l == 255) return 28;
  const int k = 31 - __builtin_clz((unsigned)l);
  return 4 * (k - 1) + ((l >> (k - 2)) & 3);
}
__device__ __forceinline__ int distance_code(const DS *, int d1) {
  if (d1 < 4) return d1;
  const int k = 31 - __builtin_clz((unsigned)d1);
  return 2 * k + ((d1 >> (k - 1)) & 1);
}
__device__ __forceinline__ void static_lit(int sym, int *len, int *code) {  // lib/de.ml:373-409
  int l, c;
  if (sym < 144) { l = 8; c = 0x30 + sym; }
  else if (sym < 256) { l = 9; c = 0x190 + (sym - 144); }
  else if (sym < 280) { l = 7; c = sym - 256; }
  else { l = 8; c = 0xc0 + (sym - 280); }
  *len = l;
  *code = (int)(__brev((unsigned)c) >> (32 - l));
}

// ---------------------------------------------------------------------------
// De.T (lib/de.ml:1828-2068): heap, lengths with overflow fix-up, reversed codes — built by the
// whole wave.  Only the merge loop (two smallest out, their parent in) is inherently serial and
// stays on lane 0, on packed 64-bit keys so that `smaller` is one compare and both children of
// a heap node come in one 16-byte LDS read.  Everything else is data-parallel: compaction of the
// used symbols (ballot), heapify level by level (the sub-heaps of one level are disjoint), code
// lengths as parent-chain depths, codes by per-length ranks, the run-length pass by position.
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint64_t hkey(uint32_t f, uint32_t depth, uint32_t n) {
  return ((uint64_t)f << 32) | (uint64_t)((depth << 16) | n);
}
// smaller, lib/de.ml:1876-1877: freq, then depth, ties count as smaller
__device__ __forceinline__ bool hsmaller(uint64_t a, uint64_t b) { return a <= (b | 0xffffull); }  // (a >> 16) <= (b >> 16)
// pqdownheap, lib/de.ml:1879-1899.  (Fetching the four grandchildren with the two children resolves two levels per
// LDS round trip and is faster for a stream alone, but costs a third more instructions per level: at 16 streams per
// CU, where the instruction issue is what the streams share, the batch got slower — measured, not kept.)
__device__ void heap_down(DS *s, int hlen, int k) {
  const uint64_t v = s->hk[k];
  int j = k << 1;
  while (j <= hlen) {
    const u64x2 ab = *(const u64x2 *)&s->hk[j];  // children j, j+1 (j even: 16-byte aligned)
    uint64_t a = ab.x;
    if (j < hlen && hsmaller(ab.y, a)) {
      j++;
      a = ab.y;
    }
    if (hsmaller(v, a)) break;
    s->hk[k] = a;
    k = j;
    j <<= 1;
  }
  s->hk[k] = v;
}
// The merge loop's sifts by the whole wavefront.  Every node has a choice: the child that a sinking key meets
// (pqdownheap's: the right one when it is not larger), the node itself for a leaf.  The choices of nodes 1 .. 63 live in
// a register, lane = node (`row`), those of deeper nodes - a heap of 128 and more - in LDS (nx).  A sift is then: the
// path from the root by lane reads of the register; its keys, one read by lane d = depth; where the key stops, one
// ballot; the keys moving up, one write; and every lane makes its node's choice again from its children's keys.  Two
// LDS round trips and some 60 instructions, where lane 0 walking down takes a round trip and 30 instructions per level:
// a stream's wavefront is alone with its latencies (one stream per CU runs this kernel only 30 % faster than sixteen).
// Returns the key at the root afterwards.
__device__ __forceinline__ int heap_choice(const DS *s, int hlen, uint32_t j) {
  const uint32_t c = j << 1;
  if ((int)c > hlen || j == 0) return (int)j;
  const u64x2 ab = *(const u64x2 *)&s->hk[c];
  return (int)(((int)c < hlen && hsmaller(ab.y, ab.x)) ? c + 1 : c);
}
// the same for node = lane, without branches: a lane beyond the heap reads the pair at the heap's end and drops it
__device__ __forceinline__ int heap_choice_row(const DS *s, int hlen, uint32_t lane) {
  const uint32_t c = lane << 1;
  const bool leaf = (int)c > hlen || lane == 0;
  const u64x2 ab = *(const u64x2 *)&s->hk[leaf ? 2u : c];
  const uint32_t pick = ((int)c < hlen && hsmaller(ab.y, ab.x)) ? c + 1 : c;
  return (int)(leaf ? lane : pick);
}
// hnext: the heap's length when the choices made at the end are used (the sift before a pop leaves them for the heap
// without its last slot)
template <bool DEEP>
__device__ __forceinline__ uint64_t heap_sink_wave(DS *s, uint16_t *nx, int hlen, int hnext, uint64_t v, uint32_t lane, int &row) {
  int p = 1;
#pragma unroll
  for (int d = 1; d <= 6; d++) p = __builtin_amdgcn_readlane(row, p);  // (a leaf's choice is the leaf: p stays)
  if (DEEP) {  // a heap of 128 and more
    if (p >= 64) p = __builtin_amdgcn_readfirstlane((int)nx[p]);
    if (p >= 64) p = __builtin_amdgcn_readfirstlane((int)nx[p]);
  }
  // the path is the ancestors of where it ends: lane d takes the one of depth d (lanes beyond the end: the end again)
  const int depth = 31 - __builtin_clz((unsigned)p);
  const int sh = depth - (int)lane;
  const int myp = p >> (sh > 0 ? sh : 0);
  const int up = __builtin_amdgcn_update_dpp(0, myp, 0x111, 0xf, 0xf, false);  // row_shr:1: p of depth - 1
  const uint64_t k = s->hk[myp];
  const bool stop = lane >= 1 && lane <= 8 && (myp == up || hsmaller(v, k));
  const uint64_t bm = __builtin_amdgcn_ballot_w64(stop);
  const uint32_t t = bm ? (uint32_t)__builtin_ctzll(bm) - 1u : 8u;  // v stops at depth t
  const uint32_t klo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, 0x101, 0xf, 0xf, false);  // row_shl:1: the key below
  const uint32_t khi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), 0x101, 0xf, 0xf, false);
  const uint64_t w = lane < t ? (((uint64_t)khi << 32) | klo) : v;
  if (lane <= t) s->hk[myp] = w;
  if (DEEP && lane < t && myp >= 64) nx[myp] = (uint16_t)heap_choice(s, hlen, (uint32_t)myp);
  row = heap_choice_row(s, hnext, lane);
  // the key at the root now: what lane 0 wrote
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
}
// one step of the merge loop, lib/de.ml:2027-2049: the two smallest out, their parent in
template <bool DEEP>
__device__ __forceinline__ void merge_step(DS *s, uint16_t *nx, int &hlen, int &hm, int &node, uint64_t &root, int &row, uint32_t lane) {
  const uint64_t n = root;
  const uint64_t last = s->hk[hlen];
  const int L = hlen--;
  // the parent of the slot that went has one child left, or is a leaf now (the register's choices were made for that)
  if (DEEP && (L >> 1) >= 64 && lane == 0) nx[L >> 1] = (uint16_t)((L & 1) ? L - 1 : (L >> 1));
  const uint64_t m = heap_sink_wave<DEEP>(s, nx, hlen, hlen, last, lane, row);
  const uint32_t ni = (uint32_t)n & 0xffff, mi = (uint32_t)m & 0xffff;
  const uint32_t fs = (uint32_t)(n >> 32) + (uint32_t)(m >> 32);
  const uint32_t dn = ((uint32_t)n >> 16), dm = ((uint32_t)m >> 16);
  hm -= 2;
  // (every lane writes the same values: no branch)
  s->heap[hm] = (uint16_t)mi;
  s->heap[hm + 1] = (uint16_t)ni;
  s->dads[ni] = (uint16_t)node;
  s->dads[mi] = (uint16_t)node;
  root = heap_sink_wave<DEEP>(s, nx, hlen, hlen - 1, hkey(fs, (dn >= dm ? dn : dm) + 1, (uint32_t)node), lane, row);
  node++;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

#define TP_MARK(i)                                          \
  if (TPROF) {                                              \
    const uint64_t t_now = wall_clock64();                  \
    if (lane == 0) s->tp[i] += (uint32_t)(t_now - t_prev);  \
    t_prev = t_now;                                         \
  }
template <bool TPROF>
__device__ void tree_make_wave(DS *s, int length, int max_length, int *f, TreeRef t, uint32_t lane) {
  uint64_t t_prev = TPROF ? wall_clock64() : 0;
  for (int n = (int)lane; n < HEAP_SIZE; n += kWave) s->dads[n] = 0;
  uint16_t *nx = s->tlen;  // (the lengths' space until the merge is over)
  if (lane <= (uint32_t)MAX_BITS) s->bl_count[lane] = 0;
  int hlen = 0, max_code = -1;
  for (int base = 0; base < length; base += kWave) {
    const int n = base + (int)lane;
    const bool nz = n < length && f[n] != 0;
    const uint64_t m = __ballot(nz);
    if (nz) s->hk[hlen + 1 + __popcll(m & lanes_below(lane))] = hkey((uint32_t)f[n], 0, (uint32_t)n);
    hlen += __popcll(m);
    if (m) max_code = base + 63 - __builtin_clzll(m);
  }
  while (hlen < 2) {  // pkzip, lib/de.ml:1863-1874 (writes into the live histogram: H2)
    const int node = max_code < 2 ? ++max_code : 0;
    if (lane == 0) {
      f[node] = 1;
      s->hk[hlen + 1] = hkey(1, 0, (uint32_t)node);
    }
    hlen++;
  }
  const int hlen0 = hlen;
  __syncthreads();
  for (int top = 1 << (31 - __builtin_clz((unsigned)(hlen >> 1))); top >= 1; top >>= 1) {
    for (int k = top + (int)lane; k < 2 * top && k <= (hlen >> 1); k += kWave) heap_down(s, hlen, k);
    __syncthreads();
  }
  const int hmax = HEAP_SIZE - (2 * hlen0 - 1);
  TP_MARK(0)
  // the merge loop, lib/de.ml:2027-2049: two smallest out, their parent in — the reference's own sequence of heap
  // operations (ties are broken by heap position), by lane 0 on the LDS heap.  (A heap held in registers with
  // scalar-unit sifts was measured: no LDS round trips, but ~30 scalar instructions per level on the one scalar unit
  // the CU's 16 streams share — not faster.)
  for (int j = 64 + (int)lane; j <= hlen; j += kWave) nx[j] = (uint16_t)heap_choice(s, hlen, (uint32_t)j);
  __syncthreads();
  {
    int hm = HEAP_SIZE, node = length;
    hlen = __builtin_amdgcn_readfirstlane(hlen);  // (a count of ballots: the same on every lane)
    int row = heap_choice_row(s, hlen - 1, lane);  // (for the heap after the first pop)
    uint64_t root = s->hk[1];
    while (hlen >= 128) merge_step<true>(s, nx, hlen, hm, node, root, row, lane);
    while (hlen >= 2) merge_step<false>(s, nx, hlen, hm, node, root, row, lane);
    if (lane == 0) s->heap[--hm] = (uint16_t)((uint32_t)root & 0xffff);
  }
  __syncthreads();
  for (int n = (int)lane; n < HEAP_SIZE; n += kWave) s->tlen[n] = 0;
  __syncthreads();
  TP_MARK(1)
  // generate_lengths, lib/de.ml:1952-2009: a node's length is its depth below the root
  const int root = s->heap[hmax];
  bool over = false;
  for (int h = hmax + 1 + (int)lane; h < HEAP_SIZE; h += kWave) {
    const int n = s->heap[h];
    int d = 0;
    for (int p = n; p != root; p = s->dads[p]) d++;
    if (d > max_length) over = true;
    else {
      s->tlen[n] = (uint16_t)d;
      if (n <= max_code) atomicAdd(&s->bl_count[d], 1);
    }
  }
  if (__ballot(over)) {  // too deep: the reference's serial fix-up, as it stands
    __syncthreads();
    if (lane == 0) {
      int overflow = 0;
      for (int i = 0; i <= MAX_BITS; i++) s->bl_count[i] = 0;
      for (int h = hmax + 1; h < HEAP_SIZE; h++) {
        int n = s->heap[h];
        int bits = s->tlen[s->dads[n]] + 1;
        if (bits > max_length) {
          overflow++;
          bits = max_length;
        }
        s->tlen[n] = (uint16_t)bits;
        if (n <= max_code) s->bl_count[bits]++;
      }
      do {
        int bits = max_length - 1;
        while (s->bl_count[bits] == 0) bits--;
        s->bl_count[bits]--;
        s->bl_count[bits + 1] += 2;
        s->bl_count[max_length]--;
        overflow -= 2;
      } while (overflow > 0);
      int h = HEAP_SIZE;
      for (int bits = max_length; bits >= 1; bits--) {
        int n = s->bl_count[bits];
        while (n != 0) {
          int m = s->heap[--h];
          if (m <= max_code) {
            s->tlen[m] = (uint16_t)bits;
            n--;
          }
        }
      }
    }
  }
  __syncthreads();
  TP_MARK(2)
  // generate_codes, lib/de.ml:1926-1950: lane b keeps next_code[b]; a symbol's code is that plus
  // the number of earlier symbols of the same length
  uint32_t ncr = 0;
  {
    uint32_t code = 0;
#pragma unroll
    for (int bits = 1; bits <= MAX_BITS; bits++) {
      code = (code + (uint32_t)s->bl_count[bits - 1]) << 1;
      if (lane == (uint32_t)bits) ncr = code & 0xffff;
    }
  }
  const int lim = length + 2 < L_CODES + 2 ? length + 2 : L_CODES + 2;
  for (int base = 0; base < lim; base += kWave) {
    const int n = base + (int)lane;
    const bool inr = n < lim;
    const uint32_t tl = inr ? s->tlen[n] : 0;
    const uint32_t len = n < length ? tl : 0;
    const bool valid = inr && n <= max_code && len > 0;
    const uint32_t basec = (uint32_t)__shfl((int)ncr, (int)(len & 15));
    uint32_t rank = 0, add = 0;
#pragma unroll
    for (uint32_t L = 1; L <= (uint32_t)MAX_BITS; L++) {
      const uint64_t m = __ballot(valid && len == L);
      if (len == L) rank = (uint32_t)__popcll(m & lanes_below(lane));
      if (lane == L) add = (uint32_t)__popcll(m);
    }
    ncr += add;
    if (inr) {
      t.lengths[n] = (uint16_t)tl;
      t.clen[n] = (uint8_t)len;
      t.codes[n] = valid ? (uint16_t)(__brev(basec + rank) >> (32 - len)) : (uint16_t)0;
    }
  }
  if (lane == 0) *t.max_code = max_code;
  __syncthreads();
  TP_MARK(3)
}

// T.scan (lib/de.ml:2070-2117) and T.symbols (lib/de.ml:2122-2191) by position.  The serial state
// machine cuts every run of equal lengths into chunks: zero runs into 138s, other runs into a
// first chunk of up to 7 and then 6s.  A position knows from its offset in its run whether it
// closes a chunk and how long that chunk is, so every chunk is emitted independently:
// emit == false counts into blf (scan), emit == true writes (len << 8) | code entries from
// out_i on and returns the new end.
__device__ int tree_rle_wave(DS *s, TreeRef t, int out_i, bool emit, uint32_t lane) {
#define BLSYM(c) (uint16_t)((s->bt.clen[c] << 8) | s->bt.codes[c])
  const int max_code = *t.max_code;
  if (!emit) {
    if (lane == 0) t.lengths[max_code + 1] = 0xffff;  // the guard T.scan leaves in the lengths
    __syncthreads();
  }
  int carry_rs = 0;
  for (int base = 0; base <= max_code; base += kWave) {
    const int n = base + (int)lane;
    const bool in = n <= max_code;
    const uint32_t cur = in ? t.lengths[n] : 0xfffeu;
    const uint32_t nxt = in ? t.lengths[n + 1] : 0u;
    const bool is_start = in && (n == 0 || t.lengths[n - 1] != cur);
    const uint64_t sm = __ballot(is_start);
    const uint64_t mine = sm & (~0ull >> (63 - lane));
    const int rs = mine ? base + 63 - __builtin_clzll(mine) : carry_rs;
    if (sm) carry_rs = base + 63 - __builtin_clzll(sm);
    const int i = n - rs;  // offset in the run
    const bool run_end = nxt != cur;
    bool first = false, chunk_end;
    int count;
    if (cur == 0) {
      const int k = i % 138;
      chunk_end = run_end || k == 137;
      count = k + 1;
    } else if (i <= 6) {
      first = true;
      chunk_end = run_end || i == 6;
      count = i + 1;
    } else {
      const int k = (i - 7) % 6;
      chunk_end = run_end || k == 5;
      count = k + 1;
    }
    const bool em = in && chunk_end;
    const int min_count = cur == 0 ? 3 : first ? 4 : 3;
    if (!emit) {
      if (em) {
        if (count < min_count) atomicAdd(&s->blf[cur], count);
        else if (cur != 0) {
          if (first) atomicAdd(&s->blf[cur], 1);
          atomicAdd(&s->blf[16], 1);
        } else if (count <= 10) atomicAdd(&s->blf[17], 1);
        else atomicAdd(&s->blf[18], 1);
      }
    } else {
      int cnt = 0;
      if (em) cnt = count < min_count ? count : cur != 0 ? (first ? 3 : 2) : 2;
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const int u = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += u;
      }
      int o = out_i + incl - cnt;
      out_i += __shfl(incl, kWave - 1);
      if (em) {
        if (count < min_count) {
          for (int r = 0; r < count; r++) s->symbols[o + r] = BLSYM(cur);
        } else if (cur != 0) {
          int c = count;
          if (first) {
            s->symbols[o++] = BLSYM(cur);
            c--;
          }
          s->symbols[o++] = BLSYM(16);
          s->symbols[o] = (uint16_t)((2 << 8) | (c - 3));
        } else if (count <= 10) {
          s->symbols[o++] = BLSYM(17);
          s->symbols[o] = (uint16_t)((3 << 8) | (count - 3));
        } else {
          s->symbols[o++] = BLSYM(18);
          s->symbols[o] = (uint16_t)((7 << 8) | (count - 11));
        }
      }
    }
  }
  __syncthreads();
  return out_i;
#undef BLSYM
}

enum { TM_NONE = 0, TM_DYNAMIC = 1, TM_CHOOSE = 2 };

// Def.dynamic_of_frequencies, lib/de.ml:2387-2407 (into the current block's trees); with
// TM_CHOOSE also the cost comparison of lib/de.ml:2415-2449 with the H3 quirk
// (`distances[i] + len`).  Leaves the block kind in s->kind_result.
template <bool TPROF>
__device__ void trees_wave(DS *s, int mode, uint32_t lane) {
  tree_make_wave<TPROF>(s, L_CODES, MAX_BITS, s->lits, tref(&s->lt), lane);
  tree_make_wave<TPROF>(s, D_CODES, MAX_BITS, s->dsts, tref(&s->dt), lane);
  uint64_t t_prev = TPROF ? wall_clock64() : 0;
  if (lane < BL_CODES) s->blf[lane] = 0;
  __syncthreads();
  tree_rle_wave(s, tref(&s->lt), 0, false, lane);
  tree_rle_wave(s, tref(&s->dt), 0, false, lane);
  TP_MARK(4)
  tree_make_wave<TPROF>(s, BL_CODES, 7, s->blf, tref(&s->bt), lane);
  if (TPROF) t_prev = wall_clock64();
  const uint64_t used = __ballot(lane < (uint32_t)BL_CODES && s->bt.clen[c_zigzag[lane < 19 ? lane : 0]] != 0);
  int max_blindex = used ? 63 - __builtin_clzll(used) : 0;
  if (max_blindex < 2) max_blindex = 2;
  int i = tree_rle_wave(s, tref(&s->lt), 0, true, lane);
  i = tree_rle_wave(s, tref(&s->dt), i, true, lane);
  const int h_len = max_blindex + 1;
  int kind = KIND_DYNAMIC;
  if (mode == TM_CHOOSE) {
    uint64_t dyn = 0, sta = 0;
    for (int k = (int)lane; k < i; k += kWave) dyn += s->symbols[k] >> 8;
    for (int k = (int)lane; k < L_CODES; k += kWave) {
      const uint32_t fq = (uint32_t)s->lits[k];
      if (fq != 0) {
        int l, c;
        static_lit(k, &l, &c);
        sta += (uint64_t)fq * (uint32_t)l;
        dyn += (uint64_t)fq * s->lt.lengths[k];
      }
    }
    if (lane < (uint32_t)D_CODES && s->dsts[lane] != 0) {
      sta += (uint64_t)(uint32_t)s->dsts[lane] + 5;
      dyn += (uint64_t)(uint32_t)s->dsts[lane] + s->dt.lengths[lane];
    }
    dyn = wave_sum64(dyn) + (uint64_t)(5 + 5 + 4 + h_len * 3);
    sta = wave_sum64(sta);
    kind = dyn <= sta ? KIND_DYNAMIC : KIND_FIXED;
  }
  if (lane == 0) {
    s->nsymbols = i;
    s->h_lit = s->lt.max_code + 1;
    s->h_dst = s->dt.max_code + 1;
    s->h_len = h_len;
    s->kind_result = kind;
  }
  __syncthreads();
  TP_MARK(5)
}
#undef TP_MARK

// ---------------------------------------------------------------------------
// workspace in HBM, per stream
struct Ws {
  const uint32_t *link;   // [slot] low half of link[p] = p - hash_head(p), 0 = NIL / out of reach (deflate_link_kernel)
  const uint8_t *flg;     // [slot] look-ahead verdict FL_* of p (deflate_match_kernel)
  const uint32_t *m, *mq; // [slot] longest_match run ahead: (length << 16) | distance, full / quartered chain
  int *queue;             // [qcap]
  uint32_t tail0, tail1;  // hash head of position len - 3 when its 4th byte is 0 / the byte 32 KiB earlier (H7)
};
__device__ __forceinline__ uint32_t g_ld(const uint32_t *p) { return __builtin_nontemporal_load(p); }
// hash_head(p) = the position p's chain link points at (absolute; 0 = NIL, like the reference's position 0)
__device__ __forceinline__ uint32_t link_at(const Ws *ws, uint32_t p) {
  const uint32_t l = __builtin_nontemporal_load(ws->link + p) & 0xffffu;
  return l ? p - l : 0u;
}
__device__ __forceinline__ int g_ldi(const int *p) { return __builtin_nontemporal_load(p); }

struct Enc {  // De.Def.encoder, lib/de.ml:2465-2478
  int kind, last;
  uint64_t hold;
  int bits, flat, fmax, k;
  uint8_t *o;
  uint32_t o_pos, o_cap;
  bool overflow;
  bool hdr;             // the code-length part of a dynamic header is still to be packed (by the wave)
  unsigned qw, qr, qc;  // the queue's cursors (shared with the matcher)
  int *q;
};
__device__ __forceinline__ void out_byte(Enc *e, unsigned b) {
  if (e->o_pos < e->o_cap) e->o[e->o_pos] = (uint8_t)b;
  else e->overflow = true;
  e->o_pos++;
}
__device__ __forceinline__ void put_bits(Enc *e, unsigned v, int n) {
  e->hold |= (uint64_t)v << e->bits;
  e->bits += n;
  while (e->bits >= 16) {
    out_byte(e, (unsigned)e->hold & 0xff);
    out_byte(e, (unsigned)(e->hold >> 8) & 0xff);
    e->hold >>= 16;
    e->bits -= 16;
  }
}
__device__ __forceinline__ void align_bits(Enc *e) {
  if (e->bits > 8) {
    out_byte(e, (unsigned)e->hold & 0xff);
    out_byte(e, (unsigned)(e->hold >> 8) & 0xff);
  } else if (e->bits > 0) out_byte(e, (unsigned)e->hold & 0xff);
  e->hold = 0;
  e->bits = 0;
}
__device__ __forceinline__ void lit_code(const DS *s, const Enc *e, int sym, int *len, int *code) {
  if (e->kind == KIND_DYNAMIC) {
    *len = s->lt.clen[sym];
    *code = s->lt.codes[sym];
  } else static_lit(sym, len, code);
}
__device__ __forceinline__ void dst_code(const DS *s, const Enc *e, int sym, int *len, int *code) {
  if (e->kind == KIND_DYNAMIC) {
    *len = s->dt.clen[sym];
    *code = s->dt.codes[sym];
  } else {
    *len = 5;
    *code = (int)(__brev((unsigned)sym) >> 27);
  }
}
__device__ __forceinline__ bool cmd_exists(const DS *s, const Enc *e, int cmd) {  // Def.exists, lib/de.ml:2451-2463
  if (e->kind != KIND_DYNAMIC || cmd == Q_EOB) return true;
  if (!(cmd & Q_COPY)) return s->lt.clen[cmd & 0xff] > 0;
  int off = cmd & 0xffff, len = (cmd >> 16) & 0x1ff;
  return s->lt.clen[257 + length_code_of(len + 3)] > 0 && s->dt.clen[distance_code(s, off)] > 0;
}
__device__ __forceinline__ void emit_eob(const DS *s, Enc *e) {
  int l, c;
  lit_code(s, e, 256, &l, &c);
  put_bits(e, (unsigned)c, l);
}
__device__ __forceinline__ void emit_header(const DS *s, Enc *e) {  // lib/de.ml:2566-2633
  put_bits(e, e->last ? 1 : 0, 1);
  if (e->kind == KIND_FIXED) put_bits(e, 1, 2);
  else if (e->kind == KIND_DYNAMIC) {
    put_bits(e, 2, 2);
    put_bits(e, (unsigned)(s->h_lit - 257), 5);
    put_bits(e, (unsigned)(s->h_dst - 1), 5);
    put_bits(e, (unsigned)(s->h_len - 4), 4);
    e->hdr = true;  // h_len x 3 bits + the code-length symbols: enc_write_wave, before the commands
  } else {
    put_bits(e, 0, 2);
    align_bits(e);
    out_byte(e, e->fmax & 0xff);
    out_byte(e, (e->fmax >> 8) & 0xff);
    out_byte(e, (~e->fmax) & 0xff);
    out_byte(e, ((~e->fmax) >> 8) & 0xff);
    e->flat = 0;
  }
}
__device__ __forceinline__ int enc_write_flat(Enc *e) {  // lib/de.ml:2927-2962
  while (e->qw != e->qr && e->flat < e->fmax) {
    int cmd = g_ldi(e->q + (e->qr++ & (e->qc - 1)));
    if (cmd != Q_EOB) {
      out_byte(e, cmd & 0xff);
      e->flat++;
    }
  }
  if (e->flat == e->fmax) {
    e->fmax = 0;
    if (!e->last) e->k = K_FLAT_DONE;
  }
  return R_OK;
}
__device__ __forceinline__ void flat_len(Enc *e) {
  if (e->qw != e->qr && g_ldi(e->q + ((e->qw - 1) & (e->qc - 1))) == Q_EOB) e->qw--;
  unsigned len = e->qw - e->qr;
  e->fmax = len < 0xffff ? (int)len : 0xffff;
}
constexpr int W_PENDING = 2;  // enc_begin: the command loop (write, lib/de.ml:2708-2897) is still to run

// block, lib/de.ml:2657-2684, up to the command loop
__device__ __forceinline__ int enc_block_begin(const DS *s, Enc *e, int kind, int last) {
  e->kind = kind;
  e->last = last;
  if (kind == KIND_FLAT) flat_len(e);
  emit_header(s, e);
  e->k = K_ENCODE;
  return kind == KIND_FLAT ? enc_write_flat(e) : W_PENDING;
}
// Def.encode, lib/de.ml:2965-3038.  For `Block the caller has already built the new block's
// trees in DS — but force must close the OLD block first, whose EOB code comes from the old
// trees; the caller passes it in.  Returns R_OK / R_BLOCK, or W_PENDING when the command
// loop of a Fixed/Dynamic block has to run next (enc_write_wave, all lanes).
__device__ __forceinline__ int enc_begin(const DS *s, Enc *e, int v, int kind, int last, int old_eob_len, int old_eob_code) {
  for (;;) {
    switch (e->k) {
    case K_FIRST_ENTRY:
      if (v == V_BLOCK) return enc_block_begin(s, e, kind, last);
      emit_header(s, e);  // the initial {Fixed; last = false} block
      e->k = K_ENCODE;
      continue;
    case K_BLOCK:
      if (v == V_BLOCK) return enc_block_begin(s, e, kind, last);
      e->k = K_ENCODE;
      continue;
    case K_FLAT_DONE:
      if (v == V_BLOCK) return enc_block_begin(s, e, kind, last);
      e->k = K_BLOCK;
      return R_BLOCK;
    default:
      if (v == V_AWAIT) return R_OK;
      if (v == V_FLUSH) return e->kind == KIND_FLAT ? enc_write_flat(e) : W_PENDING;
      // force, lib/de.ml:2899-2924: close the open block with ITS end-of-block code
      if (e->kind != KIND_FLAT) put_bits(e, (unsigned)old_eob_code, old_eob_len);
      return enc_block_begin(s, e, kind, last);
    }
  }
}

// The output side of the encoder while the wave owns it.
struct Pack {
  uint64_t hold;
  uint32_t bits, o_pos, o_cap, overflow;
  uint8_t *o;
};
// Appends one item of nb <= 48 bits per lane, in lane order: a wave prefix sum of the lengths
// gives each item its bit offset, the items are OR-ed into an LDS bit buffer behind the bits
// still held, and the finished bytes go out 4 per lane.  pad = pending_bits of the last block
// (lib/de.ml:2635-2653): the final partial byte goes out too.
// inclusive prefix sum over the wavefront in six data-parallel-primitive steps (row shifts inside the rows of 16 lanes,
// then the two row broadcasts): no LDS round trip — a shuffle-based scan is six dependent ones
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
// (One wavefront owns the bit buffer, and a wavefront's LDS operations execute in program order: the phases need no
// barrier.  A workgroup barrier here also waits for the step's global stores to complete — a microsecond per step.)
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
__device__ void pack_step(DS *s, uint32_t lane, uint64_t v, uint32_t nb, uint64_t v1, uint32_t nb1, bool pad, Pack &p) {
  // (two items per lane, the second right behind the first: a step of the command loop takes 128 commands - what a step
  // costs is its chain of dependent operations, hardly longer for two items than for one)
  const uint32_t incl = wave_incl_scan(nb + nb1);
  const uint32_t total = p.bits + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const uint32_t boff = p.bits + incl - nb - nb1;
  const uint32_t nwords = (total + 31) / 32 + 1;
  for (uint32_t i = lane; i < nwords; i += kWave) s->bb[i] = i == 0 ? (uint32_t)p.hold : 0u;
  lds_order();
  auto put = [&](uint32_t at, uint64_t x, uint32_t n) {
    if (n) {
      const uint32_t w = at >> 5, sh = at & 31;
      const uint64_t lo = x << sh;
      atomicOr(&s->bb[w], (uint32_t)lo);
      if ((lo >> 32) != 0) atomicOr(&s->bb[w + 1], (uint32_t)(lo >> 32));
      if (sh + n > 64) atomicOr(&s->bb[w + 2], (uint32_t)(x >> (64 - sh)));
    }
  };
  put(boff, v, nb);
  put(boff + nb, v1, nb1);
  lds_order();
  uint32_t nbytes = total >> 3, rem = total & 7;
  if (pad) {
    nbytes = (total + 7) >> 3;
    rem = 0;
  }
  for (uint32_t i = lane * 4; i < nbytes; i += kWave * 4) {
    const uint32_t wv = __hip_atomic_load(&s->bb[i >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (i + 4 <= nbytes && p.o_pos + i + 4 <= p.o_cap) {
      __builtin_memcpy(p.o + p.o_pos + i, &wv, 4);
    } else {
      for (uint32_t k = 0; k < 4 && i + k < nbytes; k++) {
        if (p.o_pos + i + k < p.o_cap) p.o[p.o_pos + i + k] = (uint8_t)(wv >> (8 * k));
      }
    }
  }
  if (p.o_pos + nbytes > p.o_cap) p.overflow = 1;
  p.hold = rem ? ((__hip_atomic_load(&s->bb[nbytes >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (8 * (nbytes & 3))) & ((1u << rem) - 1)) : 0;
  p.bits = rem;
  p.o_pos += nbytes;
  lds_order();
}

// write, lib/de.ml:2708-2897, by the whole wave: 64 queue commands per step.  Every lane
// turns its command into (bits, length) — a literal code, or length code + extra + distance
// code + extra (<= 48 bits).  The step stops at the first command without a code in the
// current tree (Leave) or at the end-of-block command (End); that lane contributes the EOB
// code.  A dynamic header's code lengths (lib/de.ml:2585-2633) go first when still pending.
__device__ void enc_write_wave(DS *s, uint32_t lane) {
  Pack p{s->w.hold, s->w.bits, s->w.o_pos, s->w.o_cap, s->w.overflow, s->w.o};
  uint32_t qr = s->w.qr;
  const uint32_t qw = s->w.qw, qc = s->w.qc, kind = s->w.kind, last = s->w.last;
  const int *q = s->w.q;
  uint32_t rc = R_OK, knew = K_ENCODE;
  if (s->w.hdr) {
    const uint32_t h_len = (uint32_t)s->h_len, items = h_len + (uint32_t)s->nsymbols;
    for (uint32_t base = 0; base < items; base += kWave) {
      const uint32_t k = base + lane;
      uint64_t v = 0;
      uint32_t nb = 0;
      if (k < h_len) {
        v = s->bt.clen[c_zigzag[k]];
        nb = 3;
      } else if (k < items) {
        const uint32_t sym = s->symbols[k - h_len];
        v = sym & 0xff;
        nb = sym >> 8;
      }
      pack_step(s, lane, v, nb, 0, 0, false, p);
    }
  }
  // the commands of the next step are on their way while this one is packed (the queue lives in HBM / L2; a ring
  // index is always inside it, so the load needs no guard)
  // two commands per lane: 2 * lane and 2 * lane + 1 of the step's 128
  int nxt0 = g_ldi(q + ((qr + 2 * lane) & (qc - 1))), nxt1 = g_ldi(q + ((qr + 2 * lane + 1) & (qc - 1)));
  struct Cmd {
    bool act, is_eob, is_copy, ex;
    int cmd, off, ml, lcode, dcode;
  };
  auto decode = [&](bool act, int raw) {
    Cmd c;
    c.act = act;
    c.cmd = act ? raw : 0;
    c.is_eob = act && c.cmd == Q_EOB;
    c.is_copy = act && (c.cmd & Q_COPY) != 0;
    c.off = c.cmd & 0xffff;
    c.ml = (c.cmd >> 16) & 0x1ff;
    c.lcode = c.is_copy ? length_code_of(c.ml + 3) : 0;
    c.dcode = c.is_copy ? distance_code(s, c.off) : 0;
    c.ex = true;  // Def.exists, lib/de.ml:2451-2463
    if (act && kind == KIND_DYNAMIC && !c.is_eob)
      c.ex = c.is_copy ? (s->lt.clen[257 + c.lcode] > 0 && s->dt.clen[c.dcode] > 0) : s->lt.clen[c.cmd & 0xff] > 0;
    return c;
  };
  // the item of a command that is written: its codes and extra bits, or the end-of-block code where the step stops
  auto item = [&](const Cmd &c, bool is_stop, uint64_t &v, uint32_t &nb) {
    const int sym = is_stop ? 256 : c.is_copy ? 257 + c.lcode : c.cmd;
    int l0, c0;
    if (kind == KIND_DYNAMIC) {
      l0 = s->lt.clen[sym];
      c0 = s->lt.codes[sym];
    } else static_lit(sym, &l0, &c0);
    v = (uint64_t)(uint32_t)c0;
    nb = (uint32_t)l0;
    if (!is_stop && c.is_copy) {
      const uint32_t l1 = s->t_xl[c.lcode], v1 = (uint32_t)(c.ml - s->t_bl[c.lcode & 0x1f]);
      int l2, c2;
      if (kind == KIND_DYNAMIC) {
        l2 = s->dt.clen[c.dcode];
        c2 = s->dt.codes[c.dcode];
      } else {
        l2 = 5;
        c2 = (int)(__brev((unsigned)c.dcode) >> 27);
      }
      const uint32_t l3 = s->t_xd[c.dcode & 0x1f], v3 = (uint32_t)(c.off - s->t_bd[c.dcode]);
      v |= (uint64_t)v1 << nb;
      nb += l1;
      v |= (uint64_t)(uint32_t)c2 << nb;
      nb += (uint32_t)l2;
      v |= (uint64_t)v3 << nb;
      nb += l3;
    }
  };
  constexpr uint32_t kStep = 2 * kWave;
  for (;;) {
    const uint32_t avail = qw - qr;
    if (avail == 0) break;
    const Cmd a0 = decode(2 * lane < avail, nxt0), a1 = decode(2 * lane + 1 < avail, nxt1);
    nxt0 = g_ldi(q + ((qr + kStep + 2 * lane) & (qc - 1)));
    nxt1 = g_ldi(q + ((qr + kStep + 2 * lane + 1) & (qc - 1)));
    // the step stops at the first command that is the end-of-block command (End) or has no code (Leave)
    const uint64_t m0 = __builtin_amdgcn_ballot_w64(a0.act && (a0.is_eob || !a0.ex)), m1 = __builtin_amdgcn_ballot_w64(a1.act && (a1.is_eob || !a1.ex));
    const uint32_t sp0 = m0 ? 2u * (uint32_t)__builtin_ctzll(m0) : kStep, sp1 = m1 ? 2u * (uint32_t)__builtin_ctzll(m1) + 1u : kStep;
    const uint32_t sp = sp0 < sp1 ? sp0 : sp1;
    uint64_t v0 = 0, v1 = 0;
    uint32_t nb0 = 0, nb1 = 0;
    if (a0.act && 2 * lane <= sp) item(a0, 2 * lane == sp, v0, nb0);
    if (a1.act && 2 * lane + 1 <= sp) item(a1, 2 * lane + 1 == sp, v1, nb1);
    const bool stop = sp < kStep;
    const uint64_t e0 = __builtin_amdgcn_ballot_w64(a0.is_eob), e1 = __builtin_amdgcn_ballot_w64(a1.is_eob);
    const bool end_cmd = stop && ((((sp & 1) ? e1 : e0) >> (sp >> 1)) & 1) != 0;
    pack_step(s, lane, v0, nb0, v1, nb1, end_cmd && last, p);
    qr += stop ? sp + (end_cmd ? 1u : 0u) : (avail < kStep ? avail : kStep);
    if (stop) {
      if (end_cmd && last) {
        rc = R_OK;
        knew = K_ENCODE;
      } else {
        rc = R_BLOCK;
        knew = K_BLOCK;
      }
      break;
    }
  }
  if (lane == 0) {
    s->w.hold = p.hold;
    s->w.bits = p.bits;
    s->w.o_pos = p.o_pos;
    s->w.qr = qr;
    s->w.overflow = p.overflow;
    s->w.rc = rc;
    s->w.k = knew;
    s->w.hdr = 0;
  }
}

// ---------------------------------------------------------------------------
// De.Lz77, lib/de.ml:4013-4515, in absolute positions.
struct Lz {
  int level;
  int max_chain, max_lazy, good_length, nice_length;
  const uint8_t *in;
  uint32_t n;
  uint32_t base;     // absolute position of window index 0 (32 KiB per slide)
  uint32_t filled;   // absolute end of the data copied into the window
  uint32_t strstart; // absolute
  int lookahead;
  uint32_t match_start, prev_match;
  int match_length, prev_length, match_available;
  bool eoi;
  bool more;              // a stream in pieces: input beyond n will come (`Await instead of the end of the input)
  int k;
  uint32_t prepared_end;  // positions < prepared_end have their hash_head in the LDS ring
  uint32_t p_end;         // positions < p_end (= n - 3; n - 2 for the Lz matcher) can be prepared ahead
  int matcher;            // MD_MATCHER_DE: De.Lz77 (lib/de.ml:4013-4515); MD_MATCHER_LZ: Lz (lib/lz.ml)
  int steps;              // steps executed in this lz_compress call (bulk-yield policy)
  uint32_t n_steps, n_lm, n_chain;  // profile: matcher steps, longest_match calls, chain links walked
};
// window byte at absolute position a (H7: beyond the data the reference reads what its
// 64 KiB buffer holds: zero before the first slide, the byte 32 KiB earlier after it)
__device__ __forceinline__ unsigned W(const Lz *z, uint32_t a) {
  if (a < z->filled) return z->in[a];
  if (z->base == 0 || a < (uint32_t)WSIZE) return 0;
  return a - WSIZE < z->filled ? z->in[a - WSIZE] : 0;
}
__device__ __forceinline__ unsigned W16(const Lz *z, uint32_t a) {
  if (a + 2 <= z->filled) return z->in[a] | (z->in[a + 1] << 8);
  return W(z, a) | (W(z, a + 1) << 8);
}
__device__ __forceinline__ uint32_t W32(const Lz *z, uint32_t a) {
  if (a + 4 <= z->filled) {
    uint32_t v;
    __builtin_memcpy(&v, z->in + a, 4);
    return v;
  }
  return W(z, a) | (W(z, a + 1) << 8) | (W(z, a + 2) << 16) | (W(z, a + 3) << 24);
}
// insert_string, lib/de.ml:4220-4226: the chains were built ahead (deflate_link_kernel), inserting is looking up
__device__ __forceinline__ uint32_t insert_string(const DS *s, const Lz *z, const Ws *ws, uint32_t str) {
  if (str < z->prepared_end) {
    const uint32_t l = s->hl[str & (RING - 1)];
    return l ? str - l : 0u;
  }
  if (str < z->p_end) return link_at(ws, str);
  // De's last string (position len - 3, lookahead 3): its hash reads one byte beyond the data (H7)
  return W(z, str + 3) == 0 ? ws->tail0 : ws->tail1;
}
// longest_match, lib/de.ml:4110-4174
__device__ __forceinline__ int longest_match(Lz *z, const Ws *ws, uint32_t cur_match) {
  const uint32_t ss = z->strstart;
  const uint32_t str_end = ss + (MAX_MATCH - 1);
  const uint32_t rel = ss - z->base;
  const uint32_t limit = z->base + (rel > (uint32_t)MAX_DIST ? rel - MAX_DIST : 0);
  int chain_length = z->prev_length >= z->good_length ? z->max_chain >> 2 : z->max_chain;
  z->n_lm++;
  const unsigned scan_start = W16(z, ss);
  unsigned scan_end = W16(z, ss + z->prev_length - 1);
  int best_len = z->prev_length;
  for (;;) {
    uint32_t m = cur_match;
    if (W16(z, m + best_len - 1) == scan_end && W16(z, m) == scan_start) {
      uint32_t scan = ss + 1;
      m++;
      while (scan < str_end && W32(z, scan) == W32(z, m)) { scan += 4; m += 4; }
      while (scan < str_end && W16(z, scan) == W16(z, m)) { scan += 2; m += 2; }
      while (scan < str_end && W(z, scan) == W(z, m)) { scan++; m++; }
      if (W(z, scan) == W(z, m)) scan++;
      int len = MAX_MATCH - 1 - (int)(str_end - scan);
      if (len > best_len) {
        z->match_start = cur_match;
        best_len = len;
        if (len >= z->nice_length) break;
        scan_end = W16(z, ss + best_len - 1);
      }
    }
    cur_match = link_at(ws, cur_match);
    z->n_chain++;
    chain_length--;
    if (!(cur_match > limit && chain_length != 0)) break;
  }
  return best_len <= z->lookahead ? best_len : z->lookahead;
}
__device__ __forceinline__ bool q_push_auto(DS *s, Enc *e, int v) {  // emit_*: push + auto EOB when one cell is left
  e->q[e->qw++ & (e->qc - 1)] = v;
  if (e->qc - (e->qw - e->qr) == 1) {
    e->q[e->qw++ & (e->qc - 1)] = Q_EOB;
    return true;
  }
  return false;
}
__device__ __forceinline__ bool emit_match(DS *s, Enc *e, int off, int len) {
  s->lits[257 + length_code_of(len)]++;
  s->dsts[distance_code(s, off - 1)]++;
  return q_push_auto(s, e, ((len - 3) << 16) | (off - 1) | Q_COPY);
}
__device__ __forceinline__ bool emit_literal(DS *s, Enc *e, int chr) {
  s->lits[chr]++;
  return q_push_auto(s, e, chr);
}
// deflate (one position), lib/de.ml:4351-4410
__device__ __forceinline__ bool lz_deflate(DS *s, Enc *e, Lz *z, const Ws *ws) {
  uint32_t hash_head = 0;
  if (z->lookahead >= MIN_MATCH) hash_head = insert_string(s, z, ws, z->strstart);
  z->prev_length = z->match_length;
  z->prev_match = z->match_start;
  z->match_length = MIN_MATCH - 1;
  // hash_head != 0 in window terms: absolute position strictly above the window base
  if (hash_head > z->base && z->prev_length < z->max_lazy && z->strstart - hash_head <= (uint32_t)MAX_DIST) {
    int ml;
    bool fast = false;
    ml = z->prev_length;
    if (z->strstart < z->prepared_end && z->lookahead >= MIN_LOOKAHEAD) {
      // the wave has run this position's chain ahead (kernel main loop): FL_ENDED — no candidate
      // shares 3 bytes, nothing can beat prev_length; FL_MATCH — the best candidate of the full
      // and of the quartered chain (lib/de.ml:4117-4119) are in the match ring.  With prev_length
      // as the bar, longest_match returns that candidate iff it is strictly longer.
      const uint32_t r = z->strstart & (RING - 1);
      const uint32_t f = s->flg[r];
      if (f == FL_ENDED) fast = true;
      else if (f == FL_MATCH) {
        const uint32_t m = g_ld((z->prev_length >= z->good_length ? ws->mq : ws->m) + z->strstart);
        if ((int)(m >> 16) > z->prev_length) {
          ml = (int)(m >> 16);
          z->match_start = z->strstart - (m & 0xffff);
        }
        fast = true;
      }
    }
    if (!fast) ml = longest_match(z, ws, hash_head);
    if (ml <= 5 && ml == MIN_MATCH && z->strstart - z->match_start > (uint32_t)TOO_FAR) z->match_length = MIN_MATCH - 1;
    else z->match_length = ml;
  }
  if (z->prev_length >= MIN_MATCH && z->match_length <= z->prev_length) {
    uint32_t max_insert = z->strstart + z->lookahead - MIN_MATCH;
    bool flush = emit_match(s, e, (int)(z->strstart - 1 - z->prev_match), z->prev_length);
    z->lookahead -= z->prev_length - 1;
    z->prev_length -= 2;
    do {
      z->strstart++;
      if (z->strstart <= max_insert) insert_string(s, z, ws, z->strstart);
    } while (--z->prev_length != 0);
    z->match_available = 0;
    z->match_length = MIN_MATCH - 1;
    z->strstart++;
    return flush;
  } else if (z->match_available) {
    // the pending literal: from the look-ahead ring when the position was prepared (no HBM round trip)
    const uint32_t lp = z->strstart - 1;
    bool flush = emit_literal(s, e, lp < z->prepared_end ? (int)s->byt[lp & (RING - 1)] : (int)W(z, lp));
    z->strstart++;
    z->lookahead--;
    return flush;
  }
  z->match_available = 1;
  z->strstart++;
  z->lookahead--;
  return false;
}
__device__ __forceinline__ bool lz_copy(DS *s, Enc *e, Lz *z) {  // level 0, lib/de.ml:4412-4423
  bool flush = (e->qc - (e->qw - e->qr)) <= 1;
  while (!flush && z->lookahead > 0) {
    flush = emit_literal(s, e, (int)W(z, z->strstart));
    z->strstart++;
    z->lookahead--;
  }
  return flush;
}
// Lz77.compress until `Flush or `End; the whole input is available, `Await = end of input
__device__ __forceinline__ int lz_compress(DS *s, Enc *e, Lz *z, const Ws *ws) {
  for (;;) {
    if (!(z->k == LK_ENOUGH && z->lookahead >= MIN_LOOKAHEAD)) {
      // fill_window, lib/de.ml:4294-4342
      uint32_t rel = z->strstart - z->base;
      int more = 2 * WSIZE - z->lookahead - (int)rel;
      if (rel >= (uint32_t)(WSIZE + MAX_DIST)) {
        z->base += WSIZE;  // the slide: heads/chains hold absolute positions, nothing to rewrite
        more += WSIZE;
      }
      if (z->filled >= z->n) {
        if (z->eoi) {
          if (z->lookahead == 0) {
            // trailing, lib/de.ml:4257-4266
            // (Lz.trailing pushes no end-of-block command, lib/lz.ml:348-354: the driver does)
            if (z->match_available) {
              if (!emit_literal(s, e, (int)W(z, z->strstart - 1)) && z->matcher == MD_MATCHER_DE)
                e->q[e->qw++ & (e->qc - 1)] = Q_EOB;
            } else if (z->matcher == MD_MATCHER_DE) e->q[e->qw++ & (e->qc - 1)] = Q_EOB;
            return LZ_END;
          }
        } else if (z->more) {
          return LZ_AWAIT;  // (a stream in pieces: the state is as the reference's when it answers `Await)
        } else {
          z->eoi = true;
          z->k = LK_FILL;
          continue;
        }
      } else {
        uint32_t rem = z->n - z->filled;
        uint32_t len = (uint32_t)more < rem ? (uint32_t)more : rem;
        z->filled += len;
        z->lookahead += (int)len;
        if (z->lookahead < MIN_LOOKAHEAD) {
          z->eoi = !z->more && z->filled >= z->n;
          z->k = LK_FILL;
          continue;
        }
      }
    }
    z->k = LK_ENOUGH;
    // one step may insert up to 258 positions: make sure the ring covers them (or the tail began)
    if (z->level != 0 && z->prepared_end < z->p_end && z->strstart + 260 > z->prepared_end) return LZ_NEED;
    // literal-run state (previous position had no match, its literal is pending): the wave can
    // take the following no-match positions 64 at a time — hand over after one step
    // (and in the fresh state after a match: the wave parses ahead from the look-ahead's verdicts)
    if (z->level != 0 && z->steps > 0 && z->match_length == MIN_MATCH - 1 &&
        z->lookahead > MIN_LOOKAHEAD && z->strstart + 1 < z->prepared_end && e->qc - (e->qw - e->qr) >= 8 &&
        s->flg[z->strstart & (RING - 1)] != 0 && s->flg[(z->strstart + 1) & (RING - 1)] != 0)
      return LZ_NEED;
    z->steps++;
    z->n_steps++;
    if (z->level == 0 ? lz_copy(s, e, z) : lz_deflate(s, e, z, ws)) return LZ_FLUSH;
  }
}

// make_block of the three drivers: the kind when it is known without trees (negated tree mode
// otherwise: the wave builds the trees and, for TM_CHOOSE, picks the kind)
__device__ __forceinline__ int block_plan(int driver, int dynamic, int level, int last) {
  if (driver == DRV_ENCODE) return dynamic == 0 ? KIND_FLAT : dynamic == 1 ? KIND_FIXED : -TM_DYNAMIC;  // the caller's block kind
  if (driver == DRV_CLI) return last ? KIND_FIXED : -TM_DYNAMIC;
  if (driver == DRV_ZL && level == 0) return KIND_FLAT;
  if (driver == DRV_ZL && !dynamic) return KIND_FIXED;
  return -TM_CHOOSE;
}

enum { ACT_PREP = 0, ACT_WRITE = 1, ACT_DONE = 2, ACT_TREES = 3, ACT_AWAIT = 4 };
enum { PH_LZ = 0, PH_FLUSH = 1, PH_END = 2, PH_TREES = 3, PH_TREES_END = 4, PH_SC_TREES = 5, PH_SC_WRITE = 6 };

struct Run {  // the two state machines of one stream (lane 0's registers)
  Enc e;
  Lz z;
  bool first;
  int phase;
  bool qfull;  // the reference would have raised Queue.Full
  int ol, oc;  // end-of-block code of the block that was open when new trees were asked for
  int mode;    // tree mode asked for (TM_*)
  uint32_t ncmd;  // DRV_LZ77: commands written out so far
  uint32_t sc_pos, sc_nrc;  // DRV_SCRIPT: next operation word, results reported so far
  int sc_last;
  bool sc_bad;              // DRV_SCRIPT: a malformed operation list
};

__device__ __forceinline__ void stream_begin(Run *r, const Ws *ws, const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap,
                             int level, int qcap, int driver, int matcher) {
  Enc &e = r->e;
  e.kind = KIND_FIXED;
  e.last = 0;
  e.hold = 0;
  e.bits = 0;
  e.flat = 0;
  e.fmax = 0;
  e.k = K_FIRST_ENTRY;
  e.o = out;
  e.o_pos = 0;
  e.o_cap = cap;
  e.overflow = false;
  e.hdr = false;
  e.qw = e.qr = 0;
  e.qc = (unsigned)qcap;
  e.q = ws->queue;
  Lz &z = r->z;
  z.level = driver == DRV_HIGHER ? 4 : level;  // H6: De.Higher.compress has no ?level
  z.matcher = matcher;
  if (matcher == MD_MATCHER_LZ && z.level < 4) z.level = 4;  // Lz.state: 0..4 are _4, no copy mode (lib/lz.ml:535)
  z.max_chain = c_levels[z.level][0];
  z.max_lazy = c_levels[z.level][1];
  z.good_length = c_levels[z.level][2];
  z.nice_length = c_levels[z.level][3];
  z.in = in;
  z.n = n;
  z.base = 0;
  z.filled = 0;
  z.strstart = 0;
  z.lookahead = 0;
  z.match_start = z.prev_match = 0;
  z.match_length = z.prev_length = matcher == MD_MATCHER_LZ ? MIN_MATCH - 1 : 0;  // lib/lz.ml:563-566
  z.match_available = 0;
  z.eoi = n == 0;
  z.more = false;
  z.k = LK_ENOUGH;
  z.prepared_end = 0;
  z.steps = 0;
  z.n_steps = z.n_lm = z.n_chain = 0;
  z.p_end = matcher == MD_MATCHER_LZ ? (n >= 3 ? n - 2 : 0) : (z.level != 0 && n >= 4) ? n - 3 : 0;
  r->first = true;
  r->qfull = false;
  r->ncmd = 0;
  r->sc_pos = r->sc_nrc = 0;
  r->sc_last = 0;
  r->sc_bad = false;
  r->phase = PH_LZ;
}

// DRV_SCRIPT: De.Def.encode (lib/de.ml:2965-3038) driven the way the reference's tests drive it
// (test/test_ns.ml:388-615): the caller fills the queue, counts frequencies (succ_literal / succ_length /
// succ_distance, lib/de.ml:2339-2351) and passes `Block {kind; last} / `Flush; every encode answers `Ok (0) or
// `Block (1) into res[1..], res[0] = their number.  A Dynamic block is dynamic_of_frequencies of the live
// histograms at that moment (and mutates them, H2).
__device__ __forceinline__ int script_step(DS *s, Run *r, uint32_t *res, uint32_t res_cap) {
  Enc &e = r->e;
  const uint8_t *ops = r->z.in;
  const uint32_t nops = r->z.n / 4;
  auto word = [&](uint32_t i) {
    uint32_t v;
    __builtin_memcpy(&v, ops + 4 * (size_t)i, 4);
    return v;
  };
  int rc;
  if (r->phase == PH_SC_TREES) {
    rc = enc_begin(s, &e, V_BLOCK, KIND_DYNAMIC, r->sc_last, r->ol, r->oc);
    goto have_rc;
  }
  if (r->phase == PH_SC_WRITE) {  // a command loop just finished: take its result
    e.hold = s->w.hold;
    e.bits = (int)s->w.bits;
    e.o_pos = s->w.o_pos;
    e.qr = s->w.qr;
    e.overflow = e.overflow || s->w.overflow;
    e.k = (int)s->w.k;
    rc = (int)s->w.rc;
    goto record;
  }
  for (;;) {
    if (r->sc_pos >= nops) return ACT_DONE;
    {
      const uint32_t op = word(r->sc_pos++);
      const uint32_t left = nops - r->sc_pos;
      switch (op) {
      case OP_FILL: {
        if (left < 1 || word(r->sc_pos) > left - 1) goto bad;
        const uint32_t n = word(r->sc_pos++);
        for (uint32_t i = 0; i < n; i++) {
          if (e.qc - (e.qw - e.qr) == 0) {  // Queue.push_exn: Queue.Full, lib/de.ml:2214-2217
            r->qfull = true;
            return ACT_DONE;
          }
          const uint32_t c = word(r->sc_pos++);
          const bool ok = (c & Q_COPY) ? ((c & ~0x2ffffffu) == 0 && ((c >> 16) & 0x1ff) <= 255 && (c & 0xffff) <= 32767) : c <= 256;
          if (!ok) goto bad;
          e.q[e.qw++ & (e.qc - 1)] = (int)c;
        }
        continue;
      }
      case OP_SUCC_LITERAL:
        if (left < 1 || word(r->sc_pos) > 255) goto bad;
        s->lits[word(r->sc_pos++)]++;
        continue;
      case OP_SUCC_LENGTH:
        if (left < 1 || word(r->sc_pos) < 3 || word(r->sc_pos) > 258) goto bad;
        s->lits[257 + length_code_of((int)word(r->sc_pos++))]++;
        continue;
      case OP_SUCC_DISTANCE:
        if (left < 1 || word(r->sc_pos) < 1 || word(r->sc_pos) > 32768) goto bad;
        s->dsts[distance_code(s, (int)word(r->sc_pos++) - 1)]++;
        continue;
      case OP_NEW_FREQS:  // make_literals () / make_distances (), lib/de.ml:2333-2346
        for (int i = 0; i < L_CODES; i++) s->lits[i] = i == 256 ? 1 : 0;
        for (int i = 0; i < D_CODES; i++) s->dsts[i] = 0;
        continue;
      case OP_QUEUE_RESET:
        e.qw = e.qr = 0;
        continue;
      case OP_FLUSH:
        lit_code(s, &e, 256, &r->ol, &r->oc);
        rc = enc_begin(s, &e, V_FLUSH, 0, 0, r->ol, r->oc);
        break;
      case OP_BLOCK: {
        if (left < 2 || word(r->sc_pos) > 2) goto bad;
        const int kind = (int)word(r->sc_pos++), last = word(r->sc_pos++) ? 1 : 0;
        lit_code(s, &e, 256, &r->ol, &r->oc);
        if (kind == KIND_DYNAMIC) {
          r->sc_last = last;
          r->mode = TM_DYNAMIC;
          r->phase = PH_SC_TREES;
          return ACT_TREES;
        }
        rc = enc_begin(s, &e, V_BLOCK, kind, last, r->ol, r->oc);
        break;
      }
      default:
        goto bad;
      }
    }
  have_rc:
    if (rc == W_PENDING) {
      r->phase = PH_SC_WRITE;
      return ACT_WRITE;
    }
  record:
    r->phase = PH_LZ;
    r->sc_nrc++;
    if (res && r->sc_nrc < res_cap) res[r->sc_nrc] = rc == R_BLOCK ? 1u : 0u;
  }
bad:
  r->sc_bad = true;
  return ACT_DONE;
}

// Lane 0: runs the matcher and the encoder's serial parts until the wave has to do something:
// more look-ahead (ACT_PREP), the command loop of a block (ACT_WRITE), or nothing more (ACT_DONE).
// `phase` remembers which driver step a pending command loop belongs to.
__device__ __forceinline__ int stream_step(DS *s, const Ws *ws, Run *r, int driver, int dynamic) {
  Enc &e = r->e;
  Lz &z = r->z;
  int rc, res, kind;
  if (r->phase == PH_TREES_END) {  // the last block's trees are there
    rc = enc_begin(s, &e, V_BLOCK, s->kind_result, 1, r->ol, r->oc);
    goto last_block_sent;
  }
  if (r->phase == PH_TREES) {  // trees of a new block are there
    rc = enc_begin(s, &e, V_BLOCK, s->kind_result, 0, r->ol, r->oc);
    goto block_sent;
  }
  if (r->phase != PH_LZ) {
    // a command loop just finished: take its result
    e.hold = s->w.hold;
    e.bits = (int)s->w.bits;
    e.o_pos = s->w.o_pos;
    e.qr = s->w.qr;
    e.overflow = e.overflow || s->w.overflow;
    e.k = (int)s->w.k;
    rc = (int)s->w.rc;
    if (r->phase == PH_END) return ACT_DONE;
    goto after_flush_write;
  }
  for (;;) {
    {
      res = driver == DRV_ENCODE ? (int)LZ_END : lz_compress(s, &e, &z, ws);
      if (res == LZ_NEED) return ACT_PREP;
      if (res == LZ_AWAIT) return ACT_AWAIT;
      if (driver == DRV_LZ77) {
        // `Flush (the queue is full) or `End: hand the queue's commands to the caller and empty it, as the
        // caller of De.Lz77.compress does before it calls again (lib/de.mli:453-524)
        uint32_t *co = reinterpret_cast<uint32_t *>(e.o);
        for (; e.qr != e.qw; e.qr++) {
          if ((r->ncmd + 1) * 4 <= e.o_cap) co[r->ncmd] = (uint32_t)g_ldi(e.q + (e.qr & (e.qc - 1)));
          else e.overflow = true;
          r->ncmd++;
        }
        if (res == LZ_END) return ACT_DONE;
        continue;
      }
      // the end-of-block code of the block that is open right now (force needs it after the
      // new trees have replaced the old ones in DS)
      lit_code(s, &e, 256, &r->ol, &r->oc);
      if (res == LZ_END) {
        if (driver == DRV_CLI) {  // bin/decompress.ml:67
          if (e.qc - (e.qw - e.qr) == 0) {  // Queue.push_exn raises Queue.Full, lib/de.ml:2214-2217
            r->qfull = true;
            return ACT_DONE;
          }
          e.q[e.qw++ & (e.qc - 1)] = Q_EOB;
        }
        else if (z.matcher == MD_MATCHER_LZ &&
                 !(e.qw != e.qr && g_ldi(e.q + ((e.qw - 1) & (e.qc - 1))) == Q_EOB))
          e.q[e.qw++ & (e.qc - 1)] = Q_EOB;  // Lz leaves the end-of-block command to its driver
        kind = block_plan(driver, dynamic, z.level, 1);
        if (kind < 0) {
          r->mode = -kind;
          r->phase = PH_TREES_END;
          return ACT_TREES;
        }
        rc = enc_begin(s, &e, V_BLOCK, kind, 1, r->ol, r->oc);
      last_block_sent:
        if (rc == W_PENDING) {
          r->phase = PH_END;
          return ACT_WRITE;
        }
        return ACT_DONE;
      }
      if (driver == DRV_ZL && !r->first) rc = enc_begin(s, &e, V_FLUSH, 0, 0, r->ol, r->oc);
      else {
        r->first = false;
        kind = block_plan(driver, dynamic, z.level, 0);
        if (kind < 0) {
          r->mode = -kind;
          r->phase = PH_TREES;
          return ACT_TREES;
        }
        rc = enc_begin(s, &e, V_BLOCK, kind, 0, r->ol, r->oc);
      }
    }
    for (;;) {
    block_sent:
      if (rc == W_PENDING) {
        r->phase = PH_FLUSH;
        return ACT_WRITE;
      }
    after_flush_write:
      r->phase = PH_LZ;
      if (rc == R_BLOCK && driver != DRV_CLI) {  // `Block reply: send a block again
        lit_code(s, &e, 256, &r->ol, &r->oc);
        kind = block_plan(driver, dynamic, z.level, 0);
        if (kind < 0) {
          r->mode = -kind;
          r->phase = PH_TREES;
          return ACT_TREES;
        }
        rc = enc_begin(s, &e, V_BLOCK, kind, 0, r->ol, r->oc);
        continue;
      }
      break;
    }
  }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// the next group of the ring, on its way from the front workspace: 8 positions per lane
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
struct RingPf {
  v4u la, lb;   // link words of positions q0 .. q0 + 7
  v2u f;        // their verdicts
  uint2 b;      // their bytes
  uint32_t pe;  // the group these belong to (0xffffffff: none)
  __device__ __forceinline__ void fetch(const Ws &ws, const uint8_t *__restrict__ src, uint32_t slen, uint32_t slot_len,
                                        uint32_t at, uint32_t lane) {
    pe = at;
    const uint32_t q0 = at + lane * 8;
    la = lb = v4u{0, 0, 0, 0};
    f = v2u{0, 0};
    b = make_uint2(0, 0);
    if (q0 + 8 <= slot_len) {  // (every position a fill uses is inside the slot: it ends 64 past the input at least)
      la = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(ws.link + q0));
      lb = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(ws.link + q0 + 4));
      f = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(ws.flg + q0));
    }
    if (q0 + 8 <= slen) __builtin_memcpy(&b, src + q0, 8);
    else
      for (uint32_t k = 0; k < 8 && q0 + k < slen; k++) (k < 4 ? b.x : b.y) |= (uint32_t)src[q0 + k] << (8 * (k & 3));
  }
};

template <bool PROF>
__global__ __launch_bounds__(kWave, 4) void deflate_kernel(
    int format, int level, int qcap, int driver, int dynamic, uint32_t n, const uint8_t *__restrict__ in,
    const uint64_t *__restrict__ in_off, const uint64_t *__restrict__ in_len, uint8_t *__restrict__ out,
    const uint64_t *__restrict__ out_off, const uint64_t *__restrict__ out_cap,
    uint64_t *__restrict__ out_len, int32_t *__restrict__ status, uint32_t *__restrict__ checksum,
    Front fr, int *__restrict__ ws_queue, uint64_t *__restrict__ dbg, const uint8_t *__restrict__ gz_hdr, uint32_t gz_hdr_len,
    const uint32_t *__restrict__ gz_crc, int matcher, uint32_t *__restrict__ hist, const uint32_t *__restrict__ order, Piece pcs) {
  __shared__ DS ds;
  // optional phase profile of stream 0 (md_set_option "profile"): [0] setup [1] look-ahead [2] bulk
  // literal runs [3] matcher/driver (lane 0) [4] bit packing [5] trees, in clock ticks; [8..] event counts
  const bool prof = PROF && dbg != nullptr && blockIdx.x == 0;
  uint64_t pt[6] = {0, 0, 0, 0, 0, 0}, pc[4] = {0, 0, 0, 0}, pa[4] = {0, 0, 0, 0}, pn[4] = {0, 0, 0, 0}, pmax = 0, pq[3] = {0, 0, 0};
  uint64_t t_prev = prof ? wall_clock64() : 0;
#define PROF_MARK(i)                      \
  if (prof) {                             \
    const uint64_t t_now = wall_clock64(); \
    pt[i] += t_now - t_prev;              \
    t_prev = t_now;                       \
  }
  const uint32_t lane = threadIdx.x;
  if (blockIdx.x >= n) return;
  const uint32_t sid = order ? order[blockIdx.x] : blockIdx.x;  // workgroups start in index order: longest streams first
  // a stream in pieces (struct Piece): in_off points at the byte of absolute position w0, in_len is the absolute length
  // so far; positions stay absolute everywhere below, the text and the front workspace are addressed from w0
  const bool piece = pcs.flags != nullptr;
  const uint32_t pflags = piece ? pcs.flags[sid] : 3u;
  const bool p_first = pflags & 1, p_last = (pflags & 2) != 0;
  if (pflags & 8) return;  // (a stream that ended in an earlier slice of the batch)
  const uint32_t w0 = piece ? (uint32_t)pcs.pos[4 * sid] : 0u, rebase = piece ? (uint32_t)pcs.pos[4 * sid + 1] : 0u;
  const size_t sslot = piece ? (size_t)pcs.pos[4 * sid + 2] : 0, qslot = piece ? (size_t)pcs.pos[4 * sid + 3] : sid;
  const uint8_t *src = in + in_off[sid] - w0;
  if (in_len[sid] > MD_MAX_STREAM || fr.flags[0]) {  // 32-bit cursors: the descriptor is out of range (mdeflate.h);
                                                       // or the batch is larger than md_deflate_params.total_in_bytes said
    if (lane == 0) {
      status[sid] = MD_E_INVALID_ARGUMENT;
      out_len[sid] = 0;
      if (checksum) checksum[sid] = 0;
    }
    return;
  }
  const uint32_t slen = (uint32_t)in_len[sid];
  uint8_t *dst = out + out_off[sid];
  uint64_t cap64 = out_cap[sid];
  uint32_t cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;  // more room than 32-bit cursors can use
  const uint64_t so = fr.slot[sid];
  const uint32_t slot_len = (uint32_t)(fr.slot[sid + 1] - so) + w0;  // positions [w0, slot_len) have a cell
  // (the front kernels saw the piece as a stream of its own beginning at w0: their positions - the tails - are relative)
  const uint32_t tl0 = fr.tail[2 * sid], tl1 = fr.tail[2 * sid + 1];
  Ws ws{fr.link + so - w0, fr.flg + so - w0, fr.m + so - w0, fr.mq + so - w0, ws_queue + qslot * qcap,
        tl0 ? tl0 + w0 : 0u, tl1 ? tl1 + w0 : 0u};

  // ---- cooperative setup: histograms, code tables, Adler-32 of the input
  for (uint32_t i = lane; i < (uint32_t)L_CODES; i += kWave) ds.lits[i] = i == 256 ? 1 : 0;  // make_literals
  if (lane < D_CODES) ds.dsts[lane] = 0;
  if (lane < 32) {
    ds.t_xl[lane] = c_extra_lbits[lane];
    ds.t_bl[lane] = c_base_length[lane];
    ds.t_xd[lane] = c_extra_dbits[lane];
    ds.t_bd[lane] = c_base_dist[lane];
  }
  // Lz77's update_crc (Adler-32 of the input), lib/de.ml:4217-4218: of [from, slen), continuing `seed`
  // 4 KiB a step, 4 x 16 bytes per lane (the loads of the next step are under way while this one is summed: a stream's
  // only wavefront would otherwise wait out an HBM round trip per step); sums by dot products: with k the index of a
  // byte in its 16, sum (end - pos) * byte = (end - first) * sum(byte) - sum(k * byte)
  auto adler_load = [&](uint32_t ps, v4u (&w)[4]) {
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
      const uint32_t x0 = ps + j * 1024 + lane * 16;
      w[j] = v4u{0u, 0u, 0u, 0u};
      if (x0 + 16 <= slen) __builtin_memcpy(&w[j], src + x0, 16);
      else
        for (uint32_t k = 0; x0 + k < slen && k < 16; k++) w[j][k >> 2] |= (uint32_t)src[x0 + k] << (8 * (k & 3));
    }
  };
  auto adler_of = [&](uint32_t from, uint32_t seed) {
    uint32_t a = seed & 0xffff, b = seed >> 16;
    v4u cur[4], nxt[4];
    if (from < slen) adler_load(from, cur);
    for (uint32_t ps = from; ps < slen; ps += 4096) {
      const uint32_t b0 = ps + 4096 < slen ? ps + 4096 : slen;
      if (b0 < slen) adler_load(b0, nxt);
      uint32_t s1 = 0, s2 = 0;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const uint32_t x0 = ps + j * 1024 + lane * 16;
        uint32_t t1 = 0, tk = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          t1 = __builtin_amdgcn_udot4(cur[j][q], 0x01010101u, t1, false);
          tk = __builtin_amdgcn_udot4(cur[j][q], 0x03020100u + 0x04040404u * q, tk, false);
        }
        s1 += t1;
        s2 += (b0 - x0) * t1 - tk;  // (nothing at or beyond the end: t1 = tk = 0)
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      b = (b + (b0 - ps) * a + s2) % 65521u;
      a = (a + s1) % 65521u;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) cur[j] = nxt[j];
    }
    return (b << 16) | a;
  };
  // in pieces: the host's running checksum, or (P_SUM: the batch in slices) ours, carried in the state
  const bool own_sum = piece && (pflags & 4) && format != MD_FORMAT_GZIP;
  uint32_t adler = !piece ? adler_of(0, 1u) : own_sum ? (p_first ? adler_of(0, 1u) : 1u) : pcs.sum[2 * sid];
  const uint32_t isize = piece ? pcs.sum[2 * sid + 1] : slen;
  __threadfence_block();
  __syncthreads();

  uint32_t hdr = 0;
  if (!p_first) {
    // (the frame's header went out with the first piece)
  } else if (format == MD_FORMAT_ZLIB) {
    // Zl.Def header, lib/zl.ml:511-517 (FLEVEL map lib/zl.ml:580-581)
    int flevel = level == 0 ? 0 : level <= 5 ? 1 : level == 6 ? 2 : 3;
    unsigned h = (8 + ((15 - 8) << 4)) << 8;
    h |= (unsigned)flevel << 6;
    h += 31 - (h % 31);
    if (lane == 0 && cap >= 2) {
      dst[0] = (uint8_t)(h >> 8);
      dst[1] = (uint8_t)h;
    }
    hdr = 2;
  } else if (format == MD_FORMAT_GZIP) {
    // Gz.Def header (lib/gz.ml:796-812), prepared by md_gz_set_header
    for (uint32_t i = lane; i < gz_hdr_len && i < cap; i += kWave) dst[i] = gz_hdr[i];
    hdr = gz_hdr_len;
  }
  // the two state machines of the stream: lane 0's, kept in LDS — as registers they cost every lane of the wave ~100
  // VGPRs (a value only lane 0 uses still takes a whole vector register) and with them half of the occupancy
  __shared__ Run run;
  const bool room = cap >= hdr;
  if (piece && !p_first) {
    // the state the piece before left: the two structs as they were, then what belongs to this launch
    static_assert(sizeof(DS) % 4 == 0 && sizeof(Run) % 4 == 0 && sizeof(DS) + sizeof(Run) + 4 <= kPieceState, "piece state");
    const uint32_t *st = reinterpret_cast<const uint32_t *>(pcs.state + sslot * kPieceState);
    __syncthreads();
    for (uint32_t i = lane; i < sizeof(DS) / 4; i += kWave) reinterpret_cast<uint32_t *>(&ds)[i] = st[i];
    for (uint32_t i = lane; i < sizeof(Run) / 4; i += kWave) reinterpret_cast<uint32_t *>(&run)[i] = st[sizeof(DS) / 4 + i];
    __syncthreads();
    if (lane == 0) {
      run.e.o = dst;  // the encoder's output cursor counts within the piece
      run.e.o_pos = 0;
      run.e.o_cap = cap;
      run.e.q = ws.queue;
      // the origin of the positions moved (Lz's are absolute; differences of them are all that is ever used)
      run.z.base -= rebase;
      run.z.filled -= rebase;
      run.z.strstart -= rebase;
      run.z.match_start -= rebase;
      run.z.prev_match -= rebase;
      run.z.prepared_end -= rebase;
      ds.zs.strstart -= rebase;
      ds.zs.base -= rebase;
      run.z.in = src;
      run.z.n = slen;
      run.z.p_end = matcher == MD_MATCHER_LZ ? (slen >= 3 ? slen - 2 : 0) : (run.z.level != 0 && slen >= 4) ? slen - 3 : 0;
      run.z.more = !p_last;
      ds.ctl[0] = run.z.strstart;
      ds.ctl[1] = 0;
      ds.ctl[2] = 0;
      ds.ctl[3] = ACT_PREP;
      ds.ring_dead = 1;  // the ring is loaded again, from the new front workspace
      ds.w.o = run.e.o;
      ds.w.q = run.e.q;
      ds.w.o_pos = 0;
      ds.w.o_cap = cap;
    }
    if (own_sum) {  // the bytes that are new: from where the slice before ended (the matcher had filled up to there)
      __syncthreads();
      const uint32_t from = run.z.filled;
      adler = adler_of(from < slen ? from : slen, st[(sizeof(DS) + sizeof(Run)) / 4]);
    }
  } else if (lane == 0) {
    if (room) stream_begin(&run, &ws, src, slen, dst + hdr, cap - hdr, level, qcap, driver, matcher);
    if (room && piece && !p_last) {
      run.z.more = true;
      run.z.eoi = false;
    }
    ds.ctl[0] = 0;
    ds.ctl[1] = room ? 0 : 1;
    ds.ctl[2] = 0;
    ds.ctl[3] = ACT_PREP;
    ds.ring_dead = 0;
    ds.zs.trivial = 0;
    ds.zs.bulked = 0;
    for (int i = 0; i < 8; i++) ds.tp[i] = 0;
  }
  if (driver == DRV_ENCODE && room) {
    // the input is a list of De.Queue commands (lib/de.ml:2245-2266) for ONE last block: load the queue and count the
    // frequencies the way test/test.ml's `encode_dynamic` does (:84-95); nothing is left for the matcher
    const uint32_t ncmds = slen / 4;
    __syncthreads();
    if (lane == 0) {
      for (uint32_t i = 0; i < ncmds && i < (uint32_t)qcap; i++) {
        uint32_t c;
        __builtin_memcpy(&c, src + 4 * i, 4);
        ws.queue[i] = (int)c;
        if (c == (uint32_t)Q_EOB) continue;
        if (c & Q_COPY) {
          ds.lits[257 + length_code_of((int)((c >> 16) & 0x1ff) + 3)]++;
          ds.dsts[distance_code(&ds, (int)(c & 0xffff))]++;
        } else ds.lits[c & 0xff]++;
      }
      run.e.qw = ncmds < (uint32_t)qcap ? ncmds : (uint32_t)qcap;
      run.z.n = 0;
      run.z.eoi = 1;
      run.z.p_end = 0;
      if (ncmds > (uint32_t)qcap) run.qfull = true;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
  }
  const bool no_text = driver == DRV_ENCODE || driver == DRV_SCRIPT;
  const uint32_t eff_level = driver == DRV_HIGHER ? 4 : (matcher == MD_MATCHER_LZ && level < 4) ? 4 : level;
  const uint32_t p_end = no_text ? 0
                         : matcher == MD_MATCHER_LZ ? (slen >= 3 ? slen - 2 : 0)
                                                    : (eff_level != 0 && slen >= 4) ? slen - 3 : 0;
  __syncthreads();
  PROF_MARK(0)
  RingPf rp;
  rp.pe = 0xffffffffu;
  rp.la = rp.lb = v4u{0, 0, 0, 0};
  rp.f = v2u{0, 0};
  rp.b = make_uint2(0, 0);
  for (;;) {
    if (ds.ctl[1]) break;
    const uint32_t ss = ds.ctl[0];
    uint32_t pe = ds.ctl[2];
    // the tree builder used the ring's space, and the code-length symbols of the header it made sit there until the
    // command loop that follows ACT_TREES has packed them: then the ring is loaded again, from the pending literal on
    const bool hold = ds.ctl[3] == ACT_TREES;
    if (ds.ring_dead && !hold) {
      pe = (ss ? ss - 1 : 0u) & ~63u;
      if (pe > p_end) pe = p_end;
      __syncthreads();
      if (lane == 0) ds.ring_dead = 0;
    }
    // ---- the ring runs ahead of the matcher: hash heads, verdicts and bytes of the next positions, PG steps of 64
    //      positions per load (all of it was computed for the whole batch by deflate_front.hip)
    for (;;) {
      uint32_t nb = 0;
      if (pe < p_end && !hold) {
        const uint32_t room = (ss + RING - 1 - pe) / kWave;  // slot of position ss - 1 stays intact
        const uint32_t left = (p_end - pe + kWave - 1) / kWave;
        nb = room < left ? room : left;
        if (nb > (uint32_t)PG) nb = PG;
        // wait for a full group unless the matcher is about to run dry or this is the tail
        if (nb < (uint32_t)PG && nb != left && pe >= ss + 324) nb = 0;
      }
      if (nb == 0) break;
      // a lane takes 8 consecutive positions: two 16-byte loads of link words, one 8-byte load of verdicts, one of
      // input bytes, and three LDS stores — for the whole group (pe is a multiple of 64 here, so a lane's 8 ring cells
      // are contiguous and aligned).  The loads were started when the previous group was stored: the fill does not
      // wait for HBM unless the ring was restarted somewhere else.
      if (rp.pe != pe) rp.fetch(ws, src, slen, slot_len, pe, lane);
      if (lane * 8 < nb * kWave) {
        const uint32_t r0 = (pe + lane * 8) & (RING - 1);
        // the low halves of the 8 link words (the high halves are the match kernel's fingerprints)
        const uint4 l8 = make_uint4((rp.la.x & 0xffffu) | (rp.la.y << 16),
                                    (rp.la.z & 0xffffu) | (rp.la.w << 16), (rp.lb.x & 0xffffu) | (rp.lb.y << 16),
                                    (rp.lb.z & 0xffffu) | (rp.lb.w << 16));
        *reinterpret_cast<uint4 *>(&ds.hl[r0]) = l8;
        *reinterpret_cast<uint2 *>(&ds.flg[r0]) = make_uint2(rp.f.x, rp.f.y);
        *reinterpret_cast<uint2 *>(&ds.byt[r0]) = rp.b;
      }
      {
        const uint32_t pn = pe + nb * kWave;
        rp.pe = 0xffffffffu;
        if (pn < p_end) rp.fetch(ws, src, slen, slot_len, pn, lane);
      }
      pe = pe + nb * kWave < p_end ? pe + nb * kWave : p_end;
      pc[0] += nb;
    }
    __syncthreads();
    PROF_MARK(1)
    // ---- bulk literal run: in the literal-run state a position is trivial when its chain is
    //      empty / out of reach, or none of its pre-walked candidates passes the 3-byte filter
    //      and the chain ends within them (longest_match would return prev_length = 2)
    if (ds.zs.trivial == 2) {  // fresh state: nothing for the literal-run step, the parse step counts from 0
      __syncthreads();
      if (lane == 0) ds.zs.bulked = 0;
      __syncthreads();
    }
    bool short_step = true;  // the literal run ended inside a step (or did not run): the parse step may go on from there
    if (ds.zs.trivial == 1) {
      // up to 8 steps of 64 positions per turn of the main loop (its fixed cost per turn was a third of a literal-only
      // stream's time): the state stays in registers between the steps
      uint32_t s0 = ds.zs.strstart, la = ds.zs.lookahead, qw = ds.zs.qw, total = 0;
      const uint32_t base = ds.zs.base, qr = ds.zs.qr;
      for (int rep = 0; rep < 8; rep++) {
        const uint32_t avail = (uint32_t)qcap - (qw - qr);
        uint32_t maxk = la > (uint32_t)MIN_LOOKAHEAD ? la - MIN_LOOKAHEAD + 1 : 0;
        if (maxk > (uint32_t)kWave) maxk = kWave;
        if (avail < 3) maxk = 0;
        else if (maxk > avail - 2) maxk = avail - 2;
        if (s0 >= pe) maxk = 0;
        else if (maxk > pe - s0) maxk = pe - s0;
        const uint32_t p = s0 + lane, r = p & (RING - 1);
        const uint32_t hlv = ds.hl[r], hhv = hlv ? p - hlv : 0u;
        bool triv = lane < maxk;
        if (triv && hhv > base && p - hhv <= (uint32_t)MAX_DIST) triv = ds.flg[r] == FL_ENDED;
        const uint64_t nt = __ballot(!triv);
        const uint32_t K = nt ? (uint32_t)__builtin_ctzll(nt) : (uint32_t)kWave;
        if (lane < K) {
          const uint32_t byte = ds.byt[(p - 1) & (RING - 1)];  // the pending literal of the previous position
          ws.queue[(qw + lane) & ((uint32_t)qcap - 1)] = (int)byte;
          atomicAdd(&ds.lits[byte], 1);
        }
        s0 += K;
        la -= K;
        qw += K;
        total += K;
        pc[1]++;
        short_step = K < (uint32_t)kWave;
        if (short_step) break;
      }
      if (lane == 0) {
        ds.zs.strstart = s0;
        ds.zs.lookahead = la;
        ds.zs.qw = qw;
        ds.zs.bulked = total;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      pc[2] += total;
    }
    // ---- parse step: lazy evaluation (lib/de.ml:4351-4410) of up to 64 positions from the
    //      look-ahead's verdicts, when the literal run above stopped at a match.  Every position
    //      evaluates, as if it were reached with prev_length 2, its own lazy chain: a match at p is
    //      deferred while the next position's match is strictly longer (full chain below
    //      good_length, quartered from there on, no search from max_lazy on), and is emitted when
    //      it is not.  The path from s0 through these chains is then walked and emitted.
    if (ds.zs.trivial && (ds.zs.trivial == 2 || short_step)) {
      const uint32_t s0 = ds.zs.strstart, la = ds.zs.lookahead, base = ds.zs.base;
      uint32_t qw = ds.zs.qw;
      const uint32_t qavail = (uint32_t)qcap - (qw - ds.zs.qr);
      // positions that may be evaluated: prepared with room for a full match behind them
      // (insert_string of lib/de.ml:4386-4391 stays inside the ring) and MIN_LOOKAHEAD ahead
      uint32_t E = la >= (uint32_t)MIN_LOOKAHEAD ? s0 + la - MIN_LOOKAHEAD + 1 : s0;
      if (pe < p_end) {
        const uint32_t lim = pe > 260 ? pe - 260 : 0;
        if (E > lim) E = lim;
      } else if (E > pe) E = pe;
      __syncthreads();
      {
        // what the look-ahead knows about positions s0 .. s0 + 79: lane -> position s0 + lane, lanes 0 .. 15 also
        // s0 + 64 + lane; the four loads of match results are issued before anything waits for them (no branch around
        // them: a position without a match reads its cell and drops it)
        const uint32_t q0 = s0 + lane, q1 = s0 + 64 + (lane & 15);
        const uint32_t g0 = q0 < slot_len ? q0 : slot_len - 1, g1 = q1 < slot_len ? q1 : slot_len - 1;
        const uint32_t ma0 = g_ld(ws.m + g0), mb0 = g_ld(ws.mq + g0), ma1 = g_ld(ws.m + g1), mb1 = g_ld(ws.mq + g1);
        auto fill = [&](uint32_t j, uint32_t q, uint32_t a, uint32_t b) {
          const uint32_t r = q & (RING - 1);
          uint32_t k = 0, lf = 2, df = 0, lq = 2, dq = 0;
          if (q < E) {
            const uint32_t f = ds.flg[r], l1v = ds.hl[r], c1v = l1v ? q - l1v : 0u;
            if (f == FL_ENDED || f == FL_MATCH) k = 1;
            if (c1v > base && q - c1v <= (uint32_t)MAX_DIST) k |= 2;
            if (f == FL_MATCH) {
              lf = a >> 16;
              df = a & 0xffff;
              lq = b >> 16;
              dq = b & 0xffff;
            }
          }
          ds.pm_k[j] = (uint8_t)k;
          ds.pm_lf[j] = (uint16_t)lf;
          ds.pm_df[j] = (uint16_t)df;
          ds.pm_lq[j] = (uint16_t)lq;
          ds.pm_dq[j] = (uint16_t)dq;
        };
        fill(lane, q0, ma0, mb0);
        if (lane < 16) fill(64 + lane, q1, ma1, mb1);
      }
      __syncthreads();
      // the lazy chain that starts at window position `lane`
      const uint32_t lv_max_lazy = c_levels[eff_level][1], lv_good = c_levels[eff_level][2];
      uint32_t typ = 2, ck = 0, cl = 0, cd = 0;  // typ 0 literal, 1 match chain, 2 unknown
      {
        const uint32_t k0 = ds.pm_k[lane];
        if (k0 & 1) {
          uint32_t L = (k0 & 2) ? ds.pm_lf[lane] : 2u, d = ds.pm_df[lane];
          if (L == (uint32_t)MIN_MATCH && d > (uint32_t)TOO_FAR) L = 2;  // lib/de.ml:4363-4367
          if (L < (uint32_t)MIN_MATCH) typ = 0;
          else {
            uint32_t t = lane + 1;
            typ = 1;
            for (;;) {
              if (t >= 80 || !(ds.pm_k[t] & 1)) {
                typ = 2;
                break;
              }
              if (L >= lv_max_lazy) break;  // prev_length >= max_lazy: no search
              const bool quart = L >= lv_good;
              const uint32_t m = (ds.pm_k[t] & 2) ? (quart ? ds.pm_lq[t] : ds.pm_lf[t]) : 2u;
              if (m <= L) break;
              d = quart ? ds.pm_dq[t] : ds.pm_df[t];
              L = m;
              t++;
            }
            ck = t - 1 - lane;  // positions lane .. t-2 become literals, the match sits at t-1
            cl = L;
            cd = d;
          }
        }
      }
      // ---- the path from s0 through these chains, and its commands, by the whole wave.  The matcher arrives at a
      //      position, and leaves it for the next one (no match there: its literal waits, `pending`) or for the end of
      //      the chain's match.  Which positions it arrives at is a walk along these jumps from position 0: six rounds
      //      of pointer doubling (a position's jump and the set of arrivals from it, over 1, 2, 4 ... jumps).  An arrival
      //      writes the literal that waited (the position before was an arrival without a match) and, at a chain, the
      //      chain's literals and its match; a prefix sum of these counts places them in the queue, and the arrivals that
      //      fit the queue (an arrival's commands all or none) are the ones that happen.  (It was one wave-uniform loop
      //      turn per literal run and per chain, ~80 dependent instructions each: three quarters of a text stream's time.)
      const uint32_t nxt1 = typ == 2 ? 64u : typ == 1 ? lane + ck + cl : lane + 1;  // (>= 64: out of the window)
      uint64_t arr = typ == 2 ? 0ull : 1ull << lane;
      {
        uint32_t J = nxt1 < 64u ? nxt1 : 64u;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const int from = (int)(J & 63u);
          const uint32_t alo = (uint32_t)__shfl((int)(uint32_t)arr, from), ahi = (uint32_t)__shfl((int)(uint32_t)(arr >> 32), from);
          const uint32_t jn = (uint32_t)__shfl((int)J, from);
          if (J < 64u) {
            arr |= ((uint64_t)ahi << 32) | alo;
            J = jn;
          }
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)J) >= 64u) break;  // the walk from position 0 is complete
        }
      }
      const uint64_t vis = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(arr >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)arr);  // the arrivals from position 0
      const uint32_t pending0 = ds.zs.trivial == 1 ? 1u : 0u;
      const uint64_t waits = vis & __builtin_amdgcn_ballot_w64(typ == 0);  // arrivals that leave their literal waiting
      const bool here = (vis >> lane) & 1;
      const uint32_t pend_in = lane ? (uint32_t)((waits >> (lane - 1)) & 1) : pending0;
      const uint32_t mine = here ? pend_in + (typ == 1 ? ck + 1 : 0u) : 0u;
      const uint32_t upto = wave_incl_scan(mine);
      const uint32_t cmax = qavail >= 3 ? qavail - 2 : 0;
      const bool happens = here && upto <= cmax;
      const uint64_t hm = __builtin_amdgcn_ballot_w64(happens);
      uint32_t cur = 0, pending = pending0, cmds = 0;
      if (hm) {
        const int last = 63 - __builtin_clzll(hm);
        cmds = (uint32_t)__builtin_amdgcn_readlane((int)upto, last);
        cur = (uint32_t)__builtin_amdgcn_readlane((int)nxt1, last);
        pending = (uint32_t)((waits >> last) & 1);
        if (happens) {
          uint32_t at = qw + upto - mine;
          if (pend_in) {
            const uint32_t byte = ds.byt[(s0 + lane - 1) & (RING - 1)];
            ws.queue[at++ & ((uint32_t)qcap - 1)] = (int)byte;
            atomicAdd(&ds.lits[byte], 1);
          }
          if (typ == 1) {
            for (uint32_t j = 0; j < ck; j++) {
              const uint32_t byte = ds.byt[(s0 + lane + j) & (RING - 1)];
              ws.queue[at++ & ((uint32_t)qcap - 1)] = (int)byte;
              atomicAdd(&ds.lits[byte], 1);
            }
            // emit_match, lib/de.ml:4236-4245
            ws.queue[at & ((uint32_t)qcap - 1)] = (int)(((cl - 3) << 16) | (cd - 1) | Q_COPY);
            atomicAdd(&ds.lits[257 + length_code_of((int)cl)], 1);
            atomicAdd(&ds.dsts[distance_code(&ds, (int)(cd - 1))], 1);
          }
        }
      }
      if (lane == 0 && cur > 0) {
        ds.zs.strstart = s0 + cur;
        ds.zs.lookahead = la - cur;
        ds.zs.qw = qw + cmds;
        ds.zs.trivial = pending ? 1u : 2u;
        ds.zs.bulked += cur;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      pc[2] += cmds;
    }
    PROF_MARK(2)
    // would the matcher hand straight back (lz_compress's two LZ_NEED exits)?  Then don't run it:
    // its state machine is a long walk through divergent code even when it has nothing to do
    if (ds.zs.trivial && ds.zs.bulked > 0) {
      const uint32_t la = ds.zs.lookahead, s2 = ds.zs.strstart;
      const uint32_t avail = (uint32_t)qcap - (ds.zs.qw - ds.zs.qr);
      const bool need_ring = pe < p_end && s2 + 260 > pe;
      const bool next_bulk = la > (uint32_t)MIN_LOOKAHEAD && s2 + 1 < pe && avail >= 8 &&
                             ds.flg[s2 & (RING - 1)] != 0 && ds.flg[(s2 + 1) & (RING - 1)] != 0;
      if (la >= (uint32_t)MIN_LOOKAHEAD && (need_ring || next_bulk)) {
        __syncthreads();
        if (lane == 0) {
          ds.ctl[0] = s2;
          ds.ctl[2] = pe;
        }
        __syncthreads();
        continue;
      }
    }
    if (lane == 0) {
      run.z.prepared_end = pe;
      if (ds.zs.trivial) {
        run.z.strstart = ds.zs.strstart;
        run.z.lookahead = (int)ds.zs.lookahead;
        run.e.qw = ds.zs.qw;
        run.z.match_available = ds.zs.trivial == 1 ? 1 : 0;  // the parse step may have changed the state
        run.z.match_length = MIN_MATCH - 1;
      }
      // after a bulk run that made progress the wave gets another go before the matcher steps
      run.z.steps = (ds.zs.trivial && ds.zs.bulked > 0) ? 1 : 0;
      const uint64_t q0 = prof ? wall_clock64() : 0;
      int act = driver == DRV_SCRIPT ? script_step(&ds, &run, hist ? hist + (size_t)sid * (L_CODES + D_CODES) : nullptr, L_CODES + D_CODES)
                                     : stream_step(&ds, &ws, &run, driver, dynamic);
      const uint64_t q1 = prof ? wall_clock64() : 0;
      if (prof) {
        pq[0] += q0 - t_prev;
        pq[1] += q1 - q0;
      }
      if (act == ACT_WRITE) {
        Enc &e = run.e;
        ds.w.hold = e.hold;
        ds.w.bits = (uint32_t)e.bits;
        ds.w.o_pos = e.o_pos;
        ds.w.o_cap = e.o_cap;
        ds.w.qr = e.qr;
        ds.w.qw = e.qw;
        ds.w.qc = e.qc;
        ds.w.kind = (uint32_t)e.kind;
        ds.w.last = (uint32_t)e.last;
        ds.w.overflow = 0;
        ds.w.o = e.o;
        ds.w.q = e.q;
        ds.w.hdr = e.hdr ? 1u : 0u;
        e.hdr = false;
      }
      if (act == ACT_TREES) ds.ctl[4] = (uint32_t)run.mode;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // chain links / queue commands have landed
      ds.ctl[0] = run.z.strstart;
      ds.ctl[1] = act == ACT_DONE ? 1 : act == ACT_AWAIT ? 2 : 0;
      ds.ctl[2] = pe;
      ds.ctl[3] = (uint32_t)act;
      ds.zs.strstart = run.z.strstart;
      ds.zs.lookahead = (uint32_t)run.z.lookahead;
      ds.zs.base = run.z.base;
      ds.zs.qw = run.e.qw;
      ds.zs.qr = run.e.qr;
      ds.zs.bulked = 0;
      ds.zs.trivial = (act == ACT_PREP && run.z.level != 0 && run.z.match_length == MIN_MATCH - 1 &&
                       run.phase == PH_LZ)
                          ? (run.z.match_available ? 1u : 2u)
                          : 0u;
      if (prof) pq[2] += wall_clock64() - q1;
    }
    __syncthreads();
    if (prof) {  // lane-0 turn time by the action it ended with
      const uint64_t t_now = wall_clock64(), dt = t_now - t_prev;
      const uint32_t a = ds.ctl[3] & 3;
      pa[a] += dt;
      pn[a]++;
      if (dt > pmax) pmax = dt;
    }
    PROF_MARK(3)
    pc[3]++;
    if (ds.ctl[3] == ACT_WRITE) {
      enc_write_wave(&ds, lane);
      __syncthreads();
      PROF_MARK(4)
    } else if (ds.ctl[3] == ACT_TREES) {
      trees_wave<PROF>(&ds, (int)ds.ctl[4], lane);
      if (lane == 0) ds.ring_dead = 1;
      __syncthreads();
      PROF_MARK(5)
    }
  }
  if (prof && lane == 0) {
    for (int i = 0; i < 6; i++) dbg[i] = pt[i];
    for (int i = 0; i < 4; i++) dbg[8 + i] = pc[i];
    for (int i = 0; i < 4; i++) {
      dbg[16 + i] = pa[i];
      dbg[20 + i] = pn[i];
    }
    dbg[24] = pmax;
    for (int i = 0; i < 4; i++) dbg[28 + i] = ((uint64_t)ds.tp[2 * i + 1] << 32) | ds.tp[2 * i];
    for (int i = 0; i < 3; i++) dbg[25 + i] = pq[i];
    dbg[12] = run.z.n_steps;
    dbg[13] = run.z.n_lm;
    dbg[14] = run.z.n_chain;
  }
#undef PROF_MARK
  if (driver == DRV_LZ77) {  // the commands are out; De.Lz77.literals / distances (lib/de.mli:464-470) for the caller
    if (hist) {
      for (uint32_t i = lane; i < (uint32_t)L_CODES; i += kWave) hist[(size_t)sid * (L_CODES + D_CODES) + i] = (uint32_t)ds.lits[i];
      if (lane < (uint32_t)D_CODES) hist[(size_t)sid * (L_CODES + D_CODES) + L_CODES + lane] = (uint32_t)ds.dsts[lane];
    }
    if (lane == 0) {
      const bool ok = room && !run.e.overflow;
      out_len[sid] = room ? 4ull * run.ncmd : 0;  // on overflow: the size the caller needs (mdeflate.h)
      status[sid] = ok ? MD_OK : MD_UNEXPECTED_END_OF_OUTPUT;
      if (checksum) checksum[sid] = adler;
    }
    return;
  }
  if (piece && ds.ctl[1] == 2) {  // the matcher waits for input the piece does not hold: the state goes out, nothing else
    uint32_t *st = reinterpret_cast<uint32_t *>(pcs.state + sslot * kPieceState);
    __syncthreads();
    for (uint32_t i = lane; i < sizeof(DS) / 4; i += kWave) st[i] = reinterpret_cast<const uint32_t *>(&ds)[i];
    for (uint32_t i = lane; i < sizeof(Run) / 4; i += kWave) st[sizeof(DS) / 4 + i] = reinterpret_cast<const uint32_t *>(&run)[i];
    if (lane == 0) {
      st[(sizeof(DS) + sizeof(Run)) / 4] = adler;
      out_len[sid] = hdr + run.e.o_pos;
      status[sid] = run.e.overflow ? MD_UNEXPECTED_END_OF_OUTPUT : MD_PIECE_AWAIT;
      if (checksum) checksum[sid] = adler;
    }
    return;
  }
  if (lane == 0) {
    uint32_t body = room ? run.e.o_pos : 0;
    int st = !room || run.e.overflow ? MD_UNEXPECTED_END_OF_OUTPUT : MD_OK;
    if (room && run.qfull) st = MD_QUEUE_FULL;
    if (driver == DRV_SCRIPT) {
      if (room && run.sc_bad) st = MD_E_INVALID_ARGUMENT;
      if (hist) hist[(size_t)sid * (L_CODES + D_CODES)] = room ? run.sc_nrc : 0u;
    }
    uint32_t total = hdr + body;
    if (format == MD_FORMAT_ZLIB && st == MD_OK) {
      if (cap - total < 4) st = MD_UNEXPECTED_END_OF_OUTPUT;
      else {
        dst[total] = (uint8_t)(adler >> 24);
        dst[total + 1] = (uint8_t)(adler >> 16);
        dst[total + 2] = (uint8_t)(adler >> 8);
        dst[total + 3] = (uint8_t)adler;
        total += 4;
      }
    }
    uint32_t sum = adler;
    if (format == MD_FORMAT_GZIP) {
      sum = piece ? adler : gz_crc[sid];  // (in pieces: the host's running CRC-32)
      if (st == MD_OK) {
        // Gz.Def checksum (lib/gz.ml:715-722): CRC-32, then the input size mod 2^32, little-endian
        if (cap - total < 8) st = MD_UNEXPECTED_END_OF_OUTPUT;
        else {
          for (int k = 0; k < 4; k++) {
            dst[total + k] = (uint8_t)(sum >> (8 * k));
            dst[total + 4 + k] = (uint8_t)(isize >> (8 * k));
          }
          total += 8;
        }
      }
    }
    out_len[sid] = st == MD_OK ? total : 0;
    status[sid] = st;
    if (checksum) checksum[sid] = sum;
  }
}

}  // namespace defl
}  // namespace md

extern "C" size_t md_deflate_queue_bytes(uint32_t n, int qcap) { return (size_t)n * (size_t)qcap * 4; }

// fr: the front workspace of this batch, filled by md_launch_deflate_plan + md_launch_deflate_front
extern "C" int md_launch_deflate(int format, int level, int qcap, int driver, int dynamic, uint32_t n,
                                 const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                 uint8_t *out, const uint64_t *out_off, const uint64_t *out_cap,
                                 uint64_t *out_len, int32_t *status, uint32_t *checksum, const md::defl::Front *fr,
                                 void *queue_ws, uint64_t *dbg, const uint8_t *gz_hdr, uint32_t gz_hdr_len,
                                 const uint32_t *gz_crc, int matcher, uint32_t *hist, const uint32_t *order, const void *piece_ptrs,
                                 hipStream_t stream) {
  if (n == 0) return 0;
  if (dbg) order = nullptr;  // (the profile is stream 0's)
  md::defl::Piece pcs{};  // (four device pointers in the order of struct Piece, or null: whole streams)
  if (piece_ptrs) memcpy(&pcs, piece_ptrs, sizeof pcs);
  if (dbg)
    hipLaunchKernelGGL(md::defl::deflate_kernel<true>, dim3(n), dim3(md::defl::kWave), 0, stream, format, level,
                       qcap, driver, dynamic, n, in, in_off, in_len, out, out_off, out_cap, out_len, status,
                       checksum, *fr, (int *)queue_ws, dbg, gz_hdr, gz_hdr_len, gz_crc, matcher, hist, order, pcs);
  else
    hipLaunchKernelGGL(md::defl::deflate_kernel<false>, dim3(n), dim3(md::defl::kWave), 0, stream, format, level,
                       qcap, driver, dynamic, n, in, in_off, in_len, out, out_off, out_cap, out_len, status,
                       checksum, *fr, (int *)queue_ws, dbg, gz_hdr, gz_hdr_len, gz_crc, matcher, hist, order, pcs);
  return (int)hipGetLastError();
}

// max_chain / nice_length of the level the matcher runs at (lib/de.ml:4030-4049) for the front kernels
extern "C" void md_deflate_level_params(int driver, int matcher, int level, uint32_t *max_chain, uint32_t *nice) {
  static const uint16_t lv[10][2] = {{0, 0}, {4, 8}, {8, 16}, {32, 32}, {16, 16}, {32, 32}, {128, 128}, {256, 128}, {1024, 258}, {4096, 258}};
  int eff = driver == 1 ? 4 : (matcher == MD_MATCHER_LZ && level < 4) ? 4 : level;
  *max_chain = lv[eff][0];
  *nice = lv[eff][1];
}
