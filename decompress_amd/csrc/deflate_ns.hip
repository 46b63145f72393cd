// deflate_ns.hip — De.Def.Ns.deflate / Zl.Def.Ns.deflate (lib/de.ml:3040-4010, lib/zl.ml:596-629) for gfx950 (MI355X).
//
// The reference's whole-buffer compressor is an OCaml port of libdeflate's greedy path: a hash-chain match finder over
// a 32 KiB window (hc_matchfinder, lib/de.ml:3765-3856), a greedy parser (compress_greedy, :3875-3925), block splitting
// by observation statistics (:3705-3751), its own length-limited Huffman construction (:3170-3348) and a choice
// between dynamic, static and uncompressed blocks (flush_block, :3622-3703).  Levels 1-4 differ in search depth and
// nice length; levels 5-12 are stubs upstream (Ok 0), level 0 and every uncompressed block can only end in
// `Unexpected_end_of_output upstream (write_uncompressed_blocks never advances its input, :3411-3420): both are
// reproduced as results, see md_de_def_ns_deflate in mdeflate.h.
//
// Same split as De.Lz77's path (deflate_front.hip): every position with 5 bytes left is inserted into the chains in
// order whatever the parser decides (longest_match inserts the position it is asked about, skip_positions the ones
// a match covers; the exception is the last match of the input, after which nothing is searched any more), so
//   deflate_link_kernel<NS>   builds the chains (two passes over the 16-bit hash range, head table in LDS),
//   ns_match_kernel           runs hc_matchfinder_longest_match for EVERY position (depth-limited, nice length),
//   ns_kernel                 takes the greedy decisions, splits blocks, builds the codes and writes the bits: one
//                             stream per wavefront.  Per block, pass 0 goes through the positions 64 at a time: every
//                             lane holds the jump of its position (match length or 1), the chain of token starts is
//                             followed on the scalar unit (a v_readlane per token), the lanes that are token starts
//                             count their symbols (LDS atomics) and flag their word of the match array; lane 0 builds
//                             the codes and the header; pass 1 reads the flags back and all lanes encode their tokens
//                             at once (prefix sum of the bit lengths, OR into an LDS bit buffer, coalesced stores).
#include "deflate_common.hpp"

namespace md {
namespace ns {
using md::defl::Front;
using md::defl::fp16;
using md::defl::kWave;

constexpr int MIN_BLOCK_LENGTH = 10000, END_PADDING = 8, NUM_LITLEN_SYMS = 288, MAX_LITLEN_CODEWORD_LEN = 14;
constexpr int NUM_OFFSET_SYMS = 32, MAX_OFFSET_CODEWORD_LEN = 15, NUM_SYMBOL_BITS = 10, SYMBOL_MASK = 0x3ff;
constexpr int MIN_MATCH_LEN = 3, MAX_MATCH_LEN = 258, SOFT_MAX_BLOCK_LENGTH = 300000, NUM_PRECODE_SYMS = 19;
constexpr int END_OF_BLOCK = 256, MAX_PRE_CODEWORD_LEN = 7, MAX_MAX_CODEWORD_LEN = 15;
constexpr uint32_t kChunkNs = 4 * kWave;
constexpr int NG = 4;  // groups of 64 positions per step of the sequential kernel's two passes

__constant__ uint8_t c_zigzag[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
__constant__ uint8_t c_base_length[31] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  10,  12,  14,  16,  20,  24, 28,
                                          32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 255, 0,   0};
__constant__ uint8_t c_extra_lbits[32] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0};
__constant__ uint8_t c_extra_dbits[32] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6, 6,
                                          7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
__constant__ int c_base_dist[32] = {0,   1,   2,   3,   4,    6,    8,    12,   16,   24,   32,   48,    64,    96,    128, 192,
                                    256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, -1,  -1};

// _length (lib/de.ml:240-256) of a match length and the offset slot (offset_slot_fast, lib/de.ml:3351-3358) of an offset
__device__ __forceinline__ int length_slot(int len) {
  const int l = len - 3;
  if (l < 8) return l;
  if (l == 255) return 28;
  const int k = 31 - __builtin_clz((unsigned)l);
  return 4 * (k - 1) + ((l >> (k - 2)) & 3);
}
__device__ __forceinline__ int offset_slot(int offset) {
  const int d1 = offset - 1;
  if (d1 < 4) return d1;
  const int k = 31 - __builtin_clz((unsigned)d1);
  return 2 * k + ((d1 >> (k - 1)) & 1);
}

// lz_extend (lib/de.ml:3753-3763): common prefix of the strings at a and b, at most max_len
__device__ __forceinline__ uint32_t lz_extend(const uint8_t *a, const uint8_t *b, uint32_t max_len) {
  uint32_t len = 0;
  while (max_len - len >= 4) {
    uint32_t x, y;
    __builtin_memcpy(&x, a + len, 4);
    __builtin_memcpy(&y, b + len, 4);
    if (x != y) return len + ((uint32_t)__builtin_ctz(x ^ y) >> 3);
    len += 4;
  }
  while (len < max_len && a[len] == b[len]) len++;
  return len;
}

// ---- hc_matchfinder_longest_match (lib/de.ml:3770-3833) for every position p < p_end, as the greedy parser would call
// it: lens.best = 2, lens.max = min 258 (len - p), lens.nice = min nice max.  A candidate that does not share 3 bytes
// cannot be longer than best (its fingerprint says so without touching the text); every candidate costs one unit of
// depth; the walk ends at a candidate of nice length, when the depth is used up or where the chain leaves the window
// (cur_node <= cutoff: 32768 positions back or more).  m[p] = best << 16 | offset, best = 2: no match.
__global__ __launch_bounds__(kWave) void ns_match_kernel(uint32_t n, uint32_t nchunks_max, const uint8_t *__restrict__ in,
                                                         const uint64_t *__restrict__ in_off, const uint64_t *__restrict__ in_len,
                                                         const uint32_t *__restrict__ p_end_a, const uint64_t *__restrict__ slot,
                                                         const uint32_t *__restrict__ chunk0, const uint32_t *__restrict__ link,
                                                         uint32_t *__restrict__ m, const uint32_t *__restrict__ flags,
                                                         uint32_t max_depth, uint32_t nice_level) {
  const uint32_t lane = threadIdx.x;
  if (flags[0]) return;
  const uint32_t per = (nchunks_max + 7) / 8;  // every XCD takes one contiguous eighth of the chunks (shared windows meet in one L2)
  const uint32_t c = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (c >= chunk0[n]) return;
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (chunk0[mid] <= c) lo = mid;
    else hi = mid;
  }
  const uint32_t sid = lo, p_end = p_end_a[sid];
  const uint32_t pe = (c - chunk0[sid]) * kChunkNs;
  if (pe >= p_end) return;
  const uint32_t slen = (uint32_t)in_len[sid];
  const uint8_t *src = in + in_off[sid];
  const uint64_t so = slot[sid];
  const uint32_t *lk = link + so;
  for (uint32_t g = 0; g < 4; g++) {
    const uint32_t pos = pe + g * kWave + lane;
    const bool in_range = pos < p_end;
    const uint32_t ps = in_range ? pos : 0u;  // (p_end <= len - 4: four bytes are there)
    uint32_t w4;
    __builtin_memcpy(&w4, src + ps, 4);
    const uint32_t myfp = fp16(w4);
    const uint32_t maxl = slen - ps < (uint32_t)MAX_MATCH_LEN ? slen - ps : (uint32_t)MAX_MATCH_LEN;
    const uint32_t nice = nice_level < maxl ? nice_level : maxl;
    const uint32_t d0 = lk[ps] & 0xffffu;
    md::defl::v4u own = {0u, 0u, 0u, 0u};  // the 16 bytes at the position: most comparisons end inside them
    if (maxl >= 16u) __builtin_memcpy(&own, src + ps, 16);
    uint32_t cand = ps - d0, best = MIN_MATCH_LEN - 1, off = 0, depth = max_depth;
    bool act = in_range && d0 != 0;
    while (__ballot(act)) {
      const uint32_t a = act ? cand : 0u;
      const uint32_t rec = lk[a];
      if (act) {
        if ((rec >> 16) == myfp) {
          uint32_t len;
          if (maxl >= 16u) {  // one 16-byte load of the candidate settles every match shorter than 16
            md::defl::v4u cv;
            __builtin_memcpy(&cv, src + cand, 16);
            len = md::defl::prefix16(own, cv);
            if (len == 16u) len = 16u + lz_extend(src + pos + 16, src + cand + 16, maxl - 16u);
          } else {
            len = lz_extend(src + pos, src + cand, maxl);
          }
          if (len >= nice) {
            best = len;
            off = pos - cand;
            act = false;
          } else if (len > best) {
            best = len;
            off = pos - cand;
          }
        }
        const uint32_t l = rec & 0xffffu;
        depth--;
        if (act && (l == 0 || pos - (cand - l) > 32767u || depth == 0)) act = false;
        cand -= l;
      }
    }
    if (in_range) m[so + pos] = (best << 16) | off;
  }
}

// ---- the sequential kernel ----------------------------------------------------------------------------
struct NsS {
  int freq_l[NUM_LITLEN_SYMS], freq_o[NUM_OFFSET_SYMS];
  uint16_t len_l[NUM_LITLEN_SYMS], len_o[NUM_OFFSET_SYMS], cw_l[NUM_LITLEN_SYMS], cw_o[NUM_OFFSET_SYMS];
  int a[NUM_LITLEN_SYMS];              // work array of the code construction (symbol | freq << 10, then tree links)
  int counters[NUM_LITLEN_SYMS + 4];
  int lens_tmp[NUM_LITLEN_SYMS];       // lens of the code being built (int, like the work array)
  int len_counts[MAX_MAX_CODEWORD_LEN + 2];
  int pre_freq[NUM_PRECODE_SYMS], pre_len[NUM_PRECODE_SYMS], pre_cw[NUM_PRECODE_SYMS];
  uint16_t pre_items[NUM_LITLEN_SYMS + NUM_OFFSET_SYMS];
  uint16_t both[NUM_LITLEN_SYMS + NUM_OFFSET_SYMS];  // litlen lens followed by offset lens, as the header run-length codes them
  int new_obs[10], obs[10], num_new_obs, num_obs;
  int num_litlen_syms, num_offset_syms, num_explicit_lens, num_precode_items;
  uint32_t bb[104];      // bit buffer of one encode step: 64 tokens of at most 47 bits behind at most 31 held bits
};

struct Os {  // output_bitstream, lib/de.ml:3100-3109
  uint8_t *o;
  uint32_t o_pos, o_len;
  uint64_t hold;
  int bits;
  bool failed;
};
__device__ __forceinline__ void add_bits(Os &os, uint32_t bits, int num_bits) {  // lib/de.ml:3360-3369
  if (os.failed) return;
  os.hold |= (uint64_t)bits << os.bits;
  os.bits += num_bits;
  if (os.bits >= 16) {
    if (os.o_pos + 1 >= os.o_len) {
      os.failed = true;
      return;
    }
    os.o[os.o_pos] = (uint8_t)os.hold;
    os.o[os.o_pos + 1] = (uint8_t)(os.hold >> 8);
    os.o_pos += 2;
    os.bits -= 16;
    os.hold >>= 16;
  }
}
__device__ __forceinline__ void flush_bits(Os &os) {
  if (os.failed) return;
  if (os.bits >= 8) {
    if (os.o_pos >= os.o_len) {
      os.failed = true;
      return;
    }
    os.o[os.o_pos++] = (uint8_t)os.hold;
    os.bits -= 8;
    os.hold >>= 8;
  }
}
__device__ __forceinline__ uint32_t flush_output(Os &os) {  // lib/de.ml:3422-3431
  while (!os.failed && os.bits > 0) {
    if (os.o_pos >= os.o_len) {
      os.failed = true;
      break;
    }
    os.o[os.o_pos++] = (uint8_t)os.hold;
    os.bits -= 8;
    os.hold >>= 8;
  }
  return os.o_pos;
}
// one uncompressed block of the `len` bytes at src (len < 65536): write_uncompressed_block, lib/de.ml:3399-3409
__device__ __forceinline__ void write_uncompressed_block(Os &os, const uint8_t *src, uint32_t len, bool is_final) {
  add_bits(os, is_final ? 1 : 0, 1);
  add_bits(os, 0, 2);
  os.bits += (-os.bits) & 7;  // align_bitstream
  flush_bits(os);
  if (os.failed) return;
  if (4 + len >= os.o_len - os.o_pos) {
    os.failed = true;
    return;
  }
  os.o[os.o_pos] = (uint8_t)len;
  os.o[os.o_pos + 1] = (uint8_t)(len >> 8);
  os.o[os.o_pos + 2] = (uint8_t)~len;
  os.o[os.o_pos + 3] = (uint8_t)(~len >> 8);
  os.o_pos += 4;
  for (uint32_t k = 0; k < len; k++) os.o[os.o_pos + k] = src[k];
  os.o_pos += len;
}

// the fixed Huffman code (what init_static_codes builds, lib/de.ml:3331-3349): length and bit-reversed codeword
__device__ __forceinline__ void static_litlen(int sym, int *len, int *cw) {
  int l, c;
  if (sym < 144) { l = 8; c = 0x30 + sym; }
  else if (sym < 256) { l = 9; c = 0x190 + (sym - 144); }
  else if (sym < 280) { l = 7; c = sym - 256; }
  else { l = 8; c = 0xc0 + (sym - 280); }
  *len = l;
  *cw = (int)(__brev((unsigned)c) >> (32 - l));
}

// make_huffman_code, lib/de.ml:3170-3329 (lane 0): counting sort by frequency, in-place two-queue tree, length counts
// with the length limit, canonical codewords, bit reversal.  lens_out / cw_out get 16-bit results.
__device__ void make_huffman_code(NsS *s, int num_syms, int max_codeword_len, const int *freqs, uint16_t *lens_out, uint16_t *cw_out) {
  int *a = s->a, *counters = s->counters, *lens = s->lens_tmp;
  const int num_counters = (num_syms + (3 / 4) + 3) & ~3;  // get_num_counter (sic: 3 / 4 = 0)
  for (int i = 0; i < num_counters; i++) counters[i] = 0;
  for (int sym = 0; sym < num_syms; sym++) {
    const int f = freqs[sym];
    counters[f < num_counters - 1 ? f : num_counters - 1]++;
  }
  int num_used = 0;
  for (int i = 1; i < num_counters; i++) {
    const int count = counters[i];
    counters[i] = num_used;
    num_used += count;
  }
  for (int sym = 0; sym < num_syms; sym++) {
    const int f = freqs[sym];
    if (f != 0) {
      const int i = f < num_counters - 1 ? f : num_counters - 1;
      a[counters[i]++] = sym | (f << NUM_SYMBOL_BITS);
    } else lens[sym] = 0;
  }
  {  // the last bucket holds every frequency >= num_counters - 1: sorted by (frequency, symbol); keys are distinct
    const int pos = counters[num_counters - 2], len = counters[num_counters - 1] - counters[num_counters - 2];
    for (int i = 1; i < len; i++) {
      const int v = a[pos + i];
      int j = i;
      for (; j > 0 && a[pos + j - 1] > v; j--) a[pos + j] = a[pos + j - 1];
      a[pos + j] = v;
    }
  }
  if (num_used == 1) {
    const int sym = a[0] & SYMBOL_MASK, nz = sym > 1 ? sym : 1;
    a[0] = 0;
    lens[0] = 1;
    a[nz] = 1;
    lens[nz] = 1;
  } else if (num_used > 1) {
    {  // build_tree
      int i = 0, b = 0, e = 0;
      const int sym_count = num_used;
      while (sym_count - e > 1) {
        int mm, nn;
        if (i != sym_count && (b == e || (a[i] >> NUM_SYMBOL_BITS) <= (a[b] >> NUM_SYMBOL_BITS))) mm = i++;
        else mm = b++;
        if (i != sym_count && (b == e || (a[i] >> NUM_SYMBOL_BITS) <= (a[b] >> NUM_SYMBOL_BITS))) nn = i++;
        else nn = b++;
        const int freq_shifted = (a[mm] & ~SYMBOL_MASK) + (a[nn] & ~SYMBOL_MASK);
        a[mm] = (a[mm] & SYMBOL_MASK) | (e << NUM_SYMBOL_BITS);
        a[nn] = (a[nn] & SYMBOL_MASK) | (e << NUM_SYMBOL_BITS);
        a[e] = (a[e] & SYMBOL_MASK) | freq_shifted;
        e++;
      }
    }
    int *len_counts = s->len_counts;
    for (int i = 0; i <= MAX_MAX_CODEWORD_LEN + 1; i++) len_counts[i] = 0;
    {  // compute_length_counts
      const int root_idx = num_used - 2;
      len_counts[1] = 2;
      a[root_idx] &= SYMBOL_MASK;
      for (int node = root_idx - 1; node >= 0; node--) {
        const int parent = a[node] >> NUM_SYMBOL_BITS;
        const int depth = (a[parent] >> NUM_SYMBOL_BITS) + 1;
        int len = depth;
        a[node] = (a[node] & SYMBOL_MASK) | (depth << NUM_SYMBOL_BITS);
        if (len >= max_codeword_len) {
          len = max_codeword_len - 1;
          while (len_counts[len] == 0) len--;
        }
        len_counts[len]--;
        len_counts[len + 1] += 2;
      }
    }
    {  // gen_codewords
      int next_codewords[MAX_MAX_CODEWORD_LEN + 1];
      int i = 0;
      for (int len = max_codeword_len; len != 0; len--)
        for (int count = len_counts[len]; count != 0; count--) lens[a[i++] & SYMBOL_MASK] = len;
      next_codewords[0] = 0;
      next_codewords[1] = 0;
      for (int len = 2; len <= max_codeword_len; len++) next_codewords[len] = (next_codewords[len - 1] + len_counts[len - 1]) << 1;
      for (int sym = 0; sym < num_syms; sym++) a[sym] = next_codewords[lens[sym]]++;
    }
  }
  for (int sym = 0; sym < num_syms; sym++) {
    const int l = lens[sym];
    lens_out[sym] = (uint16_t)l;
    cw_out[sym] = (uint16_t)(((unsigned)__brev((unsigned)a[sym] & 0xffffu) >> 16) >> (16 - l));  // reverse_codeword
  }
}

// compute_precode_items, lib/de.ml:3434-3487
__device__ int compute_precode_items(NsS *s, const uint16_t *lens, int num_lens) {
  for (int i = 0; i < NUM_PRECODE_SYMS; i++) s->pre_freq[i] = 0;
  int itemptr = 0, run_start = 0;
  while (run_start != num_lens) {
    const int len = lens[run_start];
    int run_end = run_start;
    while (run_end != num_lens && len == lens[run_end]) run_end++;
    if (len == 0) {
      while (run_end - run_start >= 11) {
        const int extra = run_end - run_start - 11 < 0x7F ? run_end - run_start - 11 : 0x7F;
        s->pre_freq[18]++;
        s->pre_items[itemptr++] = (uint16_t)(18 | (extra << 5));
        run_start += 11 + extra;
      }
      if (run_end - run_start >= 3) {
        const int extra = run_end - run_start - 3 < 0x7 ? run_end - run_start - 3 : 0x7;
        s->pre_freq[17]++;
        s->pre_items[itemptr++] = (uint16_t)(17 | (extra << 5));
        run_start += 3 + extra;
      }
    } else if (run_end - run_start >= 4) {
      s->pre_freq[len]++;
      s->pre_items[itemptr++] = (uint16_t)len;
      run_start++;
      while (run_end - run_start >= 3) {
        const int extra = run_end - run_start - 3 < 0x3 ? run_end - run_start - 3 : 0x3;
        s->pre_freq[16]++;
        s->pre_items[itemptr++] = (uint16_t)(16 | (extra << 5));
        run_start += 3 + extra;
      }
    }
    while (run_start != run_end) {
      s->pre_freq[len]++;
      s->pre_items[itemptr++] = (uint16_t)len;
      run_start++;
    }
  }
  return itemptr;
}

// precompute_huffman_header, lib/de.ml:3489-3536: the lens the header codes are the first num_litlen_syms litlen lens
// followed by the first num_offset_syms offset lens (upstream shuffles the two arrays so that they are contiguous and
// puts the offset lens back afterwards — the litlen lens it overwrote belong to unused symbols and are set again by the
// next block's code; here the concatenation is built aside)
__device__ void precompute_huffman_header(NsS *s) {
  int n = NUM_LITLEN_SYMS;
  while (!(n == 257 || s->len_l[n - 1] != 0)) n--;
  s->num_litlen_syms = n;
  n = NUM_OFFSET_SYMS;
  while (!(n == 1 || s->len_o[n - 1] != 0)) n--;
  s->num_offset_syms = n;
  for (int i = 0; i < s->num_litlen_syms; i++) s->both[i] = s->len_l[i];
  for (int i = 0; i < s->num_offset_syms; i++) s->both[s->num_litlen_syms + i] = s->len_o[i];
  s->num_precode_items = compute_precode_items(s, s->both, s->num_litlen_syms + s->num_offset_syms);
  uint16_t pl[NUM_PRECODE_SYMS], pc[NUM_PRECODE_SYMS];
  make_huffman_code(s, NUM_PRECODE_SYMS, MAX_PRE_CODEWORD_LEN, s->pre_freq, pl, pc);
  for (int i = 0; i < NUM_PRECODE_SYMS; i++) {
    s->pre_len[i] = pl[i];
    s->pre_cw[i] = pc[i];
  }
  n = NUM_PRECODE_SYMS;
  while (n > 1 && s->pre_len[c_zigzag[n - 1]] == 0) n--;
  s->num_explicit_lens = n;
}

constexpr uint32_t kNoMatch = (uint32_t)(MIN_MATCH_LEN - 1) << 16;  // a match word that says "literal"
constexpr uint32_t kTokenFlag = 0x80000000u;                          // set by pass 0 in the words of the positions the parse stops at
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// lane 0 has moved the output state: every lane takes it over
__device__ __forceinline__ void os_bcast(Os &os) {
  os.o_pos = uni(os.o_pos);
  os.bits = (int)uni((uint32_t)os.bits);
  os.hold = ((uint64_t)uni((uint32_t)(os.hold >> 32)) << 32) | uni((uint32_t)os.hold);
  os.failed = uni(os.failed ? 1u : 0u) != 0;
}
// (one wavefront owns the LDS state and a wavefront's LDS operations execute in program order: no barrier)
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
// The output side while all lanes write: bits held (< 32) in front of byte o_pos, nothing is stored at or beyond o_cap.
struct Pack {
  uint64_t hold;
  uint32_t bits, o_pos, o_cap;
  uint8_t *o;
};
// Appends one item of nb <= 47 bits per lane, in lane order (the bit stream add_bits builds, lib/de.ml:3360-3369): a
// prefix sum of the lengths gives each item its bit offset, the items are OR-ed into the LDS bit buffer behind the bits
// still held, the finished bytes go out 4 per lane.  Returns the number of bits appended.
__device__ __forceinline__ uint32_t pack_step(NsS *s, uint32_t lane, uint64_t v, uint32_t nb, Pack &p) {
  const uint32_t incl = wave_incl_scan(nb);
  const uint32_t added = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const uint32_t total = p.bits + added;
  const uint32_t boff = p.bits + incl - nb;
  const uint32_t nwords = (total + 31) / 32 + 1;
  for (uint32_t i = lane; i < nwords; i += kWave) s->bb[i] = i == 0 ? (uint32_t)p.hold : 0u;
  lds_order();
  if (nb) {
    const uint32_t w = boff >> 5, sh = boff & 31;
    const uint64_t lo = v << sh;
    atomicOr(&s->bb[w], (uint32_t)lo);
    if ((lo >> 32) != 0) atomicOr(&s->bb[w + 1], (uint32_t)(lo >> 32));
    if (sh + nb > 64) atomicOr(&s->bb[w + 2], (uint32_t)(v >> (64 - sh)));
  }
  lds_order();
  const uint32_t nbytes = total >> 3, rem = total & 7;
  for (uint32_t i = lane * 4; i < nbytes; i += kWave * 4) {
    const uint32_t wv = __hip_atomic_load(&s->bb[i >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (i + 4 <= nbytes && p.o_pos + i + 4 <= p.o_cap) {
      __builtin_memcpy(p.o + p.o_pos + i, &wv, 4);
    } else {
      for (uint32_t k = 0; k < 4 && i + k < nbytes; k++)
        if (p.o_pos + i + k < p.o_cap) p.o[p.o_pos + i + k] = (uint8_t)(wv >> (8 * k));
    }
  }
  p.hold = rem ? ((__hip_atomic_load(&s->bb[nbytes >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (8 * (nbytes & 3))) & ((1u << rem) - 1)) : 0;
  p.bits = rem;
  p.o_pos += nbytes;
  lds_order();
  return added;
}

// do_end_block_check / should_end_block, lib/de.ml:3717-3751
__device__ __forceinline__ bool should_end_block(NsS *s, uint32_t in_block_begin, uint32_t in_next, uint32_t in_end) {
  if (s->num_new_obs < 512 || in_next - in_block_begin < (uint32_t)MIN_BLOCK_LENGTH || in_end - in_next < (uint32_t)MIN_BLOCK_LENGTH) return false;
  const long block_length = (long)(in_next - in_block_begin);
  if (s->num_obs > 0) {
    long total_delta = 0;
    for (int i = 0; i < 10; i++) {
      const long expected = (long)s->obs[i] * s->num_new_obs, actual = (long)s->new_obs[i] * s->num_obs;
      total_delta += actual > expected ? actual - expected : expected - actual;
    }
    if (total_delta + (block_length / 4096 * s->num_obs) >= (long)(512 * 200 / 512) * s->num_obs) return true;
  }
  for (int i = 0; i < 10; i++) {
    s->num_obs += s->new_obs[i];
    s->obs[i] += s->new_obs[i];
    s->new_obs[i] = 0;
  }
  s->num_new_obs = 0;
  return false;
}


// format: MD_FORMAT_DEFLATE (De.Def.Ns.deflate) or MD_FORMAT_ZLIB (Zl.Def.Ns.deflate: 2 header bytes, the body, the
// Adler-32 of the input big-endian)
__global__ __launch_bounds__(kWave) void ns_kernel(int format, int level, uint32_t n, const uint8_t *__restrict__ in,
                                                   const uint64_t *__restrict__ in_off, const uint64_t *__restrict__ in_len,
                                                   uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
                                                   const uint64_t *__restrict__ out_cap, uint64_t *__restrict__ out_len,
                                                   int32_t *__restrict__ status, uint32_t *__restrict__ checksum, Front fr) {
  __shared__ NsS ds;
  NsS *s = &ds;
  const uint32_t lane = threadIdx.x, sid = blockIdx.x;
  if (sid >= n) return;
  if (in_len[sid] > MD_MAX_STREAM || fr.flags[0]) {
    if (lane == 0) {
      status[sid] = MD_E_INVALID_ARGUMENT;
      out_len[sid] = 0;
      if (checksum) checksum[sid] = 0;
    }
    return;
  }
  const uint8_t *src = in + in_off[sid];
  const uint32_t slen = (uint32_t)in_len[sid];
  uint8_t *dst0 = out + out_off[sid];
  const uint64_t cap64 = out_cap[sid];
  const uint32_t cap0 = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;
  const uint64_t so = fr.slot[sid];
  const uint32_t *mw = fr.m + so;
  const uint32_t p_end = fr.p_end[sid];

  // Adler-32 of the input (Zl.Def.Ns, lib/zl.ml:618-621), by the wave
  uint32_t adler = 1;
  if (format == MD_FORMAT_ZLIB || checksum) {
    uint32_t a = 1, b = 0;
    for (uint32_t ps = 0; ps < slen; ps += 1024) {
      const uint32_t b0 = ps + 1024 < slen ? ps + 1024 : slen;
      uint32_t s1 = 0, s2 = 0;
      for (uint32_t k = 0; k < 16; k++) {
        const uint32_t x = ps + lane * 16 + k;
        if (x < b0) {
          const uint32_t d = src[x];
          s1 += d;
          s2 += (b0 - x) * d;
        }
      }
      for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
      }
      b = (b + (b0 - ps) * a + s2) % 65521u;
      a = (a + s1) % 65521u;
    }
    adler = (b << 16) | a;
  }
  uint32_t hdr = 0;
  bool room = true;
  if (format == MD_FORMAT_ZLIB) {  // Zl.Def.Ns.header, lib/zl.ml:602-610 (FLEVEL: 0|1 -> 0, 2..5 -> 1, 6 -> 2, else 3)
    if (cap0 < 2) room = false;
    else {
      unsigned h = (8 + ((15 - 8) << 4)) << 8;
      const unsigned lv = (level == 0 || level == 1) ? 0 : (level >= 2 && level <= 5) ? 1 : level == 6 ? 2 : 3;
      h |= lv << 6;
      h += 31 - (h % 31);
      if (lane == 0) {
        dst0[0] = (uint8_t)(h >> 8);
        dst0[1] = (uint8_t)h;
      }
      hdr = 2;
    }
  }
  uint8_t *dst = dst0 + hdr;
  const uint32_t cap = room ? cap0 - hdr : 0;

  // ---- lane 0's state
  Os os{dst, 0, cap >= (uint32_t)END_PADDING ? cap - END_PADDING : 0, 0, 0, false};
  uint32_t res = 0;        // the `Ok n` of De.Def.Ns.deflate
  bool simple_done = false;
  const int min_size_to_compress = 56 - level * 4;
  if (!room) simple_done = true;
  else if (cap < (uint32_t)END_PADDING) simple_done = true;  // Ok 0
  else if (slen < (uint32_t)(min_size_to_compress > 0 ? min_size_to_compress : 0)) {
    if (lane == 0) {
      write_uncompressed_block(os, src, slen, true);
      res = flush_output(os);
    }
    simple_done = true;
  } else if (level == 0) {  // compress_none: write_uncompressed_blocks with input left ends in `Unexpected_end_of_output
    os.failed = true;
    simple_done = true;
  } else if (level > 4) simple_done = true;  // compress_lazy: "TO DO" upstream, Ok 0
  if (!simple_done) {
    // compress_greedy, lib/de.ml:3875-3925
    uint32_t *mwr = fr.m + so;
    for (uint32_t q = p_end + lane; q < slen; q += kWave) mwr[q] = kNoMatch;  // no match where fewer than 5 bytes are left
    uint32_t i_pos = 0;
    while (!os.failed && i_pos < slen) {
      // ---- a block: in_block_begin = i_pos
      const uint32_t blk_begin = i_pos;
      uint32_t p = i_pos;
      for (uint32_t k = lane; k < (uint32_t)NUM_LITLEN_SYMS; k += kWave) s->freq_l[k] = 0;
      if (lane < (uint32_t)NUM_OFFSET_SYMS) s->freq_o[lane] = 0;
      if (lane < 10) s->new_obs[lane] = s->obs[lane] = 0;
      if (lane == 0) s->num_new_obs = s->num_obs = 0;
      lds_order();
      uint32_t num_new_obs = 0;  // (wave-uniform copy of s->num_new_obs)
      const uint32_t in_max_block_end = blk_begin + (slen - blk_begin < (uint32_t)SOFT_MAX_BLOCK_LENGTH ? slen - blk_begin : (uint32_t)SOFT_MAX_BLOCK_LENGTH);
      // ---- pass 0: the greedy parse, NG x 64 positions at a time (one round trip to the match array per step)
      while (p < in_max_block_end) {
        if (num_new_obs >= 512u && p - blk_begin >= (uint32_t)MIN_BLOCK_LENGTH && slen - p >= (uint32_t)MIN_BLOCK_LENGTH) {
          uint32_t end = 0;  // should_end_block has something to decide before the token at p
          if (lane == 0) {
            s->num_new_obs = (int)num_new_obs;
            end = should_end_block(s, blk_begin, p, slen) ? 1u : 0u;
          }
          lds_order();
          end = uni(end);
          num_new_obs = uni((uint32_t)s->num_new_obs);
          if (end) break;
        }
        uint32_t mm[NG], byte[NG], jump[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
          const uint32_t q = p + g * kWave + lane;
          mm[g] = q < p_end ? __builtin_nontemporal_load(mwr + q) : kNoMatch;
          byte[g] = q < slen ? src[q] : 0u;
        }
#pragma unroll
        for (int g = 0; g < NG; g++) jump[g] = (mm[g] >> 16) >= (uint32_t)MIN_MATCH_LEN ? mm[g] >> 16 : 1u;
        // the token starts among these positions (scalar unit; the test in front of a token is should_end_block's
        // `num_new_obs < 512 || ...` with the observations of this step counted in)
        uint32_t t = 0, cnt = 0;
        uint64_t mask[NG];
        // can should_end_block become due inside this step?  Not while fewer than 512 observations can have been made
        // at its end, and not before the block (or after the input's end minus) MIN_BLOCK_LENGTH: the usual step
        // walks without asking
        const uint32_t step_end = p + (uint32_t)(NG * kWave) + (uint32_t)MAX_MATCH_LEN;
        const bool may_be_due = num_new_obs + (uint32_t)(NG * kWave) >= 512u && step_end - blk_begin >= (uint32_t)MIN_BLOCK_LENGTH &&
                                slen - p >= (uint32_t)MIN_BLOCK_LENGTH;
        const uint32_t t_end = in_max_block_end - p;  // > 0
        if (!may_be_due) {
#pragma unroll
          for (int g = 0; g < NG; g++) {
            mask[g] = 0;
            const uint32_t lim = t_end < (uint32_t)(g + 1) * kWave ? t_end : (uint32_t)(g + 1) * kWave;
            while (t < lim) {
              mask[g] |= 1ull << (t - g * kWave);
              cnt++;
              t += (uint32_t)__builtin_amdgcn_readlane((int)jump[g], (int)(t - g * kWave));
            }
          }
        } else {
          bool due = false;
#pragma unroll
          for (int g = 0; g < NG; g++) {
            mask[g] = 0;
            while (!due && t < (uint32_t)(g + 1) * kWave && t < t_end) {
              if (cnt && num_new_obs + cnt >= 512u && p + t - blk_begin >= (uint32_t)MIN_BLOCK_LENGTH && slen - (p + t) >= (uint32_t)MIN_BLOCK_LENGTH) {
                due = true;
                break;
              }
              mask[g] |= 1ull << (t - g * kWave);
              cnt++;
              t += (uint32_t)__builtin_amdgcn_readlane((int)jump[g], (int)(t - g * kWave));
            }
          }
        }
        uint32_t n8 = 0, n9 = 0, n0 = 0, n1 = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
          if (mask[g] == 0) continue;  // (wave-uniform)
          const uint32_t q = p + g * kWave + lane;
          const uint32_t best = mm[g] >> 16, off = mm[g] & 0xffffu;
          const bool is_match = best >= (uint32_t)MIN_MATCH_LEN;
          const bool tok = (mask[g] >> lane) & 1;
          if (tok) {  // choose_match / choose_literal: the symbol counts, and the flag pass 1 finds the token by
            if (is_match) {
              atomicAdd(&s->freq_l[257 + length_slot((int)best)], 1);
              atomicAdd(&s->freq_o[offset_slot((int)off)], 1);
            } else {
              atomicAdd(&s->freq_l[byte[g]], 1);
            }
            mwr[q] = mm[g] | kTokenFlag;
          }
          // observe_match / observe_literal (sic: upstream observes the POSITION, shifted left: its lowest bit)
          n8 += (uint32_t)__builtin_popcountll(__ballot(tok && is_match && best < 9u));
          n9 += (uint32_t)__builtin_popcountll(__ballot(tok && is_match && best >= 9u));
          n0 += (uint32_t)__builtin_popcountll(__ballot(tok && !is_match && (q & 1u) == 0u));
          n1 += (uint32_t)__builtin_popcountll(__ballot(tok && !is_match && (q & 1u) != 0u));
        }
        if (lane == 0) {
          s->new_obs[8] += (int)n8;
          s->new_obs[9] += (int)n9;
          s->new_obs[0] += (int)n0;
          s->new_obs[1] += (int)n1;
        }
        num_new_obs += cnt;
        p += t;
      }
      lds_order();
      // ---- flush_block, lib/de.ml:3622-3703 (lane 0): codes, header, the three costs
      const uint32_t blk_end = p;
      uint32_t block_type = 0;
      if (lane == 0) {
        const bool is_final = blk_end == slen;
        const long block_length = (long)(blk_end - blk_begin);
        s->num_new_obs = (int)num_new_obs;
        s->freq_l[END_OF_BLOCK]++;
        make_huffman_code(s, NUM_LITLEN_SYMS, MAX_LITLEN_CODEWORD_LEN, s->freq_l, s->len_l, s->cw_l);
        make_huffman_code(s, NUM_OFFSET_SYMS, MAX_OFFSET_CODEWORD_LEN, s->freq_o, s->len_o, s->cw_o);
        precompute_huffman_header(s);
        long dynamic_cost = 5 + 5 + 4 + 3 * s->num_explicit_lens, static_cost = 0, uncompressed_cost = 0;
        for (int sym = 0; sym < NUM_PRECODE_SYMS; sym++) {
          const int extra = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
          dynamic_cost += (long)s->pre_freq[sym] * (extra + s->pre_len[sym]);
        }
        for (int sym = 0; sym <= 255; sym++) dynamic_cost += (long)s->freq_l[sym] * s->len_l[sym];
        for (int sym = 0; sym <= 143; sym++) static_cost += (long)s->freq_l[sym] * 8;
        for (int sym = 144; sym <= 255; sym++) static_cost += (long)s->freq_l[sym] * 9;
        dynamic_cost += s->len_l[256];
        static_cost += 7;
        for (int sym = 257; sym <= 257 + 32 - 3; sym++) {
          const int extra = c_extra_lbits[sym - 257];
          int sl, sc;
          static_litlen(sym, &sl, &sc);
          dynamic_cost += (long)s->freq_l[sym] * (extra + s->len_l[sym]);
          static_cost += (long)s->freq_l[sym] * (extra + sl);
        }
        for (int sym = 0; sym <= 32 - 3; sym++) {
          const int extra = c_extra_dbits[sym];
          dynamic_cost += (long)s->freq_o[sym] * (extra + s->len_o[sym]);
          static_cost += (long)s->freq_o[sym] * (extra + 5);
        }
        uncompressed_cost += ((-(os.bits + 3)) & 7) + 32 + 40 * (((block_length + 65535 - 1) / 65535) - 1) + 8 * block_length;
        const long ms = static_cost < uncompressed_cost ? static_cost : uncompressed_cost;
        block_type = dynamic_cost < ms ? 2 : static_cost < uncompressed_cost ? 1 : 0;
        if (block_type == 0) {
          // write_uncompressed_blocks from block_begin with input left: upstream never advances its input
          // cursor and leaves by `Unexpected_end_of_output (lib/de.ml:3411-3420)
          os.failed = true;
        } else {
          add_bits(os, is_final ? 1 : 0, 1);
          add_bits(os, block_type, 2);
          if (block_type == 2) {  // write_huffman_header, lib/de.ml:3538-3556
            add_bits(os, (uint32_t)(s->num_litlen_syms - 257), 5);
            add_bits(os, (uint32_t)(s->num_offset_syms - 1), 5);
            add_bits(os, (uint32_t)(s->num_explicit_lens - 4) & 0xf, 4);
            for (int i = 0; i < s->num_explicit_lens; i++) add_bits(os, (uint32_t)s->pre_len[c_zigzag[i]], 3);
            for (int i = 0; i < s->num_precode_items; i++) {
              const int item = s->pre_items[i], sym = item & 0x1F;
              add_bits(os, (uint32_t)s->pre_cw[sym], s->pre_len[sym]);
              if (sym >= 16) add_bits(os, (uint32_t)(item >> 5), sym == 16 ? 2 : sym == 17 ? 3 : 7);
            }
          }
        }
      }
      block_type = uni(block_type);
      os_bcast(os);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the codes are in LDS, the token flags in the match array
      if (os.failed) break;
      // ---- pass 1: write_sequences (lib/de.ml:3558-3614) and write_end_of_block, every lane its token
      {
        const uint32_t o1 = os.o_pos, b1 = (uint32_t)os.bits;
        Pack pk{os.hold, b1, o1, os.o_len, os.o};
        uint32_t sbits = 0;
        // the words and bytes of the next step are on their way while this one is packed (branch-free: a lane beyond
        // the block reads position 0)
        auto fetch = [&](uint32_t base, uint32_t (&mw)[NG], uint32_t (&lit)[NG]) {
#pragma unroll
          for (int g = 0; g < NG; g++) {
            const uint32_t q = base + g * kWave + lane;
            const uint32_t a = q < blk_end ? q : 0u;  // (blk_end <= slen)
            mw[g] = __builtin_nontemporal_load(mwr + a);
            lit[g] = src[a];
          }
        };
        uint32_t nmw[NG], nlit[NG];
        fetch(blk_begin, nmw, nlit);
        for (uint32_t base = blk_begin; base < blk_end; base += NG * kWave) {
          uint32_t mw4[NG], lit4[NG];
#pragma unroll
          for (int g = 0; g < NG; g++) {
            const uint32_t q = base + g * kWave + lane;
            mw4[g] = q < blk_end ? nmw[g] : 0u;
            lit4[g] = nlit[g];
          }
          fetch(base + NG * kWave, nmw, nlit);
#pragma unroll
          for (int g = 0; g < NG; g++) {
            if (base + g * kWave >= blk_end) break;  // (wave-uniform)
            const uint32_t mm = mw4[g];
            uint64_t v = 0;
            uint32_t nb = 0;
            if (mm & kTokenFlag) {
              const uint32_t best = (mm >> 16) & 0x1ffu, off = mm & 0xffffu;
              if (best >= (uint32_t)MIN_MATCH_LEN) {
                const int lslot = length_slot((int)best), oslot = offset_slot((int)off);
                int l, c, l2, c2;
                if (block_type == 2) {
                  l = s->len_l[257 + lslot];
                  c = s->cw_l[257 + lslot];
                  l2 = s->len_o[oslot];
                  c2 = s->cw_o[oslot];
                } else {
                  static_litlen(257 + lslot, &l, &c);
                  l2 = 5;
                  c2 = (int)(__brev((unsigned)oslot) >> 27);
                }
                v = (uint64_t)(uint32_t)c;
                nb = (uint32_t)l;
                v |= (uint64_t)(best - c_base_length[lslot] - 3u) << nb;
                nb += c_extra_lbits[lslot];
                v |= (uint64_t)(uint32_t)c2 << nb;
                nb += (uint32_t)l2;
                v |= (uint64_t)(off - (uint32_t)c_base_dist[oslot] - 1u) << nb;
                nb += c_extra_dbits[oslot];
              } else {
                const int lit = (int)lit4[g];
                int l, c;
                if (block_type == 2) {
                  l = s->len_l[lit];
                  c = s->cw_l[lit];
                } else static_litlen(lit, &l, &c);
                v = (uint64_t)(uint32_t)c;
                nb = (uint32_t)l;
              }
            }
            sbits += pack_step(s, lane, v, nb, pk);
          }
        }
        {  // write_end_of_block
          uint64_t v = 0;
          uint32_t nb = 0;
          if (lane == 0) {
            int l, c;
            if (block_type == 2) {
              l = s->len_l[END_OF_BLOCK];
              c = s->cw_l[END_OF_BLOCK];
            } else static_litlen(END_OF_BLOCK, &l, &c);
            v = (uint64_t)(uint32_t)c;
            nb = (uint32_t)l;
          }
          sbits += pack_step(s, lane, v, nb, pk);
        }
        // what add_bits and flush_bits would have said on the way: the first group of 16 bits goes out when 16 are
        // held and needs two bytes of room each time; flush_bits then one more byte for 8 held bits
        const uint32_t groups = (b1 + sbits) >> 4, held = (b1 + sbits) & 15u;
        if ((groups >= 1 && o1 + 2 * groups > os.o_len) || (held >= 8 && o1 + 2 * groups >= os.o_len)) os.failed = true;
        os.o_pos = pk.o_pos;
        os.bits = (int)pk.bits;
        os.hold = pk.hold;
      }
      i_pos = blk_end;
    }
    if (lane == 0) res = flush_output(os);
  }
  if (lane == 0) {
    int st = (!room || os.failed) ? MD_UNEXPECTED_END_OF_OUTPUT : MD_OK;
    uint32_t total = st == MD_OK ? hdr + res : 0;
    if (st == MD_OK && format == MD_FORMAT_ZLIB) {
      // lib/zl.ml:622-627 (upstream asks for 2 bytes of room and writes 4: with 2 or 3 left it writes out of bounds —
      // `Unexpected_end_of_output here)
      if (cap - res < 4) {
        st = MD_UNEXPECTED_END_OF_OUTPUT;
        total = 0;
      } else {
        dst[res] = (uint8_t)(adler >> 24);
        dst[res + 1] = (uint8_t)(adler >> 16);
        dst[res + 2] = (uint8_t)(adler >> 8);
        dst[res + 3] = (uint8_t)adler;
        total = res + 6;
      }
    }
    out_len[sid] = total;
    status[sid] = st;
    if (checksum) checksum[sid] = adler;
  }
}

}  // namespace ns
}  // namespace md

extern "C" int md_launch_link_ns(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len, const md::defl::Front *f,
                                 hipStream_t stream);

// links (deflate_front.hip, NS variant), matches, then the sequential kernel.  max_depth / nice: lib/de.ml:3933-3936.
extern "C" int md_launch_def_ns(int format, int level, uint32_t n, uint32_t nchunks_max, const uint8_t *in, const uint64_t *in_off,
                                const uint64_t *in_len, uint8_t *out, const uint64_t *out_off, const uint64_t *out_cap,
                                uint64_t *out_len, int32_t *status, uint32_t *checksum, const md::defl::Front *f, hipStream_t stream) {
  using namespace md::ns;
  if (n == 0) return 0;
  static const uint32_t depth[5] = {0, 2, 6, 12, 24}, nice[5] = {0, 8, 10, 14, 24};
  if (level >= 1 && level <= 4 && nchunks_max != 0) {
    int e = md_launch_link_ns(n, in, in_off, in_len, f, stream);
    if (e != 0) return e;
    const uint32_t per = (nchunks_max + 7) / 8;
    hipLaunchKernelGGL(ns_match_kernel, dim3(per * 8), dim3(kWave), 0, stream, n, nchunks_max, in, in_off, in_len, f->p_end, f->slot,
                       f->chunk0, f->link, f->m, f->flags, depth[level], nice[level]);
  }
  hipLaunchKernelGGL(ns_kernel, dim3(n), dim3(kWave), 0, stream, format, level, n, in, in_off, in_len, out, out_off, out_cap, out_len,
                     status, checksum, *f);
  return (int)hipGetLastError();
}
