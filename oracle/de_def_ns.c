/* oracle/de_def_ns.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of De.Def.Ns (lib/de.ml:3040-4010) and Zl.Def.Ns (lib/zl.ml:596-629): the reference's
 * whole-buffer compressor, an OCaml port of libdeflate's greedy path — hash-chain match finder over a 32 KiB window
 * (hash of 4 bytes, 16 bits, multiplier 0x1E35A7BD), greedy parser, block splitting by observation statistics, its own
 * length-limited Huffman construction, a choice between dynamic / static / uncompressed blocks.  Levels 1..4 are
 * implemented upstream; levels 5..12 are stubs that return Ok 0 (lib/de.ml:3927); level 0 and every uncompressed
 * block go through write_uncompressed_blocks, which upstream never advances its input cursor (lib/de.ml:3411-3420)
 * and so can only end by running out of output: `Unexpected_end_of_output.  All of that is restated as it stands.
 *
 * Parity pinning: the reference holds two compressed-byte vectors for this path (test/test_ns.ml:1189-1222, both
 * reproduced: tests/golden/def_ns.json); everything else upstream is round trips.  Beyond those two vectors the byte
 * parity of this file is UNPINNED.
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define MIN_BLOCK_LENGTH 10000
#define END_PADDING 8
#define NUM_LITLEN_SYMS 288
#define MAX_LITLEN_CODEWORD_LEN 14
#define NUM_OFFSET_SYMS 32
#define MAX_OFFSET_CODEWORD_LEN 15
#define MAX_NUM_SYMS 288
#define NUM_SYMBOL_BITS 10
#define SYMBOL_MASK 0x3ff
#define MIN_MATCH_LEN 3
#define MAX_MATCH_LEN 258
#define SOFT_MAX_BLOCK_LENGTH 300000
#define NUM_PRECODE_SYMS 19
#define END_OF_BLOCK 256
#define MAX_PRE_CODEWORD_LEN 7
#define MAX_MAX_CODEWORD_LEN 15
#define WINDOW_SIZE 32768
#define HASH4_ORDER 16

/* lib/de.ml:237-325 */
static const uint8_t zigzag[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
static const int base_length[31] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  10,  12,  14,  16,  20,  24, 28,
                                    32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 255, 0,   0};
static const int extra_lbits[32] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0};
static const int extra_dbits[32] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6, 6,
                                    7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
static const int base_dist[32] = {0,   1,   2,   3,   4,    6,    8,    12,   16,   24,   32,   48,    64,    96,    128, 192,
                                  256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, -1,  -1};
static uint8_t length_slot_of[259]; /* _length, lib/de.ml:240-256 */
static uint8_t offset_slot_fast[32769];

typedef struct {
  int litlen[NUM_LITLEN_SYMS], offset[NUM_OFFSET_SYMS];
} lit_off;
typedef struct {
  lit_off codewords, lens;
} codes_t;
typedef struct {
  int level, min_size_to_compress, max_search_depth, nice_match_length;
  lit_off freqs;
  codes_t codes, static_codes;
  int precode_freqs[NUM_PRECODE_SYMS], precode_lens[NUM_PRECODE_SYMS], precode_codewords[NUM_PRECODE_SYMS];
  int precode_items[NUM_LITLEN_SYMS + NUM_OFFSET_SYMS];
  int num_litlen_syms, num_offset_syms, num_explicit_lens, num_precode_items;
} encoder_t;
typedef struct {
  const uint8_t *i;
  long i_pos, i_len;
  uint8_t *o;
  long o_pos, o_len;
  uint64_t hold;
  int bits;
  int failed; /* Malformed `Unexpected_end_of_output was raised */
} os_t;
typedef struct {
  int litrunlen_and_length, offset, offset_symbol, length_slot;
} seq_t;
typedef struct {
  int new_observations[10], observations[10], num_new_observations, num_observations;
} split_stats_t;
typedef struct {
  int best, nice, max;
} lens_t;
typedef struct {
  int *hash4_tab; /* [1 << 16] */
  int next_hash4;
  int *next_tab; /* [WINDOW_SIZE] */
} hc_mf;

static void init_tables(void) {
  static int done = 0;
  if (done) return;
  for (int len = 0; len < 259; len++) {
    int c = 0;
    if (len >= 3) {
      int l = len - 3;
      if (l == 255) c = 28;
      else
        for (c = 27; c > 0 && base_length[c] > l; c--) {}
    }
    length_slot_of[len] = (uint8_t)c;
  }
  /* init_offset_slot_fast, lib/de.ml:3351-3358 */
  for (int slot = 0; slot <= 32 - 3; slot++) {
    int off = base_dist[slot] + 1, end = off + (1 << extra_dbits[slot]);
    for (int k = off; k < end; k++) offset_slot_fast[k] = (uint8_t)slot;
  }
  done = 1;
}

/* ---- canonical Huffman code, lib/de.ml:3170-3348 ---- */
static int get_num_counter(int num_syms) { return (num_syms + (3 / 4) + 3) & ~3; } /* (sic: 3 / 4 = 0) */

static int cmp_symout(const void *pa, const void *pb) {
  int a = *(const int *)pa, b = *(const int *)pb;
  if (a == 0) return 1;
  if (b == 0) return -1;
  return a - b;
}
static int sort_symbols(int num_syms, const int *freqs, int *lens, int *symout) {
  int counters[MAX_NUM_SYMS + 4];
  memset(counters, 0, sizeof counters);
  const int num_counters = get_num_counter(num_syms);
  for (int sym = 0; sym < num_syms; sym++) {
    int i = freqs[sym] < num_counters - 1 ? freqs[sym] : num_counters - 1;
    counters[i]++;
  }
  int num_used_syms = 0;
  for (int i = 1; i < num_counters; i++) {
    int count = counters[i];
    counters[i] = num_used_syms;
    num_used_syms += count;
  }
  for (int sym = 0; sym < num_syms; sym++) {
    int freq = freqs[sym];
    if (freq != 0) {
      int i = freq < num_counters - 1 ? freq : num_counters - 1;
      symout[counters[i]] = sym | (freq << NUM_SYMBOL_BITS);
      counters[i]++;
    } else lens[sym] = 0;
  }
  int pos = counters[num_counters - 2], len = counters[num_counters - 1] - counters[num_counters - 2];
  qsort(symout + pos, (size_t)len, sizeof(int), cmp_symout); /* keys are distinct: any sort gives the same order */
  return num_used_syms;
}
static void build_tree(int *a, int sym_count) {
  int i = 0, b = 0, e = 0;
  while (sym_count - e > 1) {
    int m, n;
    if (i != sym_count && (b == e || (a[i] >> NUM_SYMBOL_BITS) <= (a[b] >> NUM_SYMBOL_BITS))) m = i++;
    else m = b++;
    if (i != sym_count && (b == e || (a[i] >> NUM_SYMBOL_BITS) <= (a[b] >> NUM_SYMBOL_BITS))) n = i++;
    else n = b++;
    int freq_shifted = (a[m] & ~SYMBOL_MASK) + (a[n] & ~SYMBOL_MASK);
    a[m] = (a[m] & SYMBOL_MASK) | (e << NUM_SYMBOL_BITS);
    a[n] = (a[n] & SYMBOL_MASK) | (e << NUM_SYMBOL_BITS);
    a[e] = (a[e] & SYMBOL_MASK) | freq_shifted;
    e++;
  }
}
static void compute_length_counts(int *a, int root_idx, int *len_counts, int max_codeword) {
  len_counts[1] = 2;
  a[root_idx] &= SYMBOL_MASK;
  for (int node = root_idx - 1; node >= 0; node--) {
    int parent = a[node] >> NUM_SYMBOL_BITS;
    int parent_depth = a[parent] >> NUM_SYMBOL_BITS;
    int depth = parent_depth + 1, len = depth;
    a[node] = (a[node] & SYMBOL_MASK) | (depth << NUM_SYMBOL_BITS);
    if (len >= max_codeword) {
      len = max_codeword - 1;
      while (len_counts[len] == 0) len--;
    }
    len_counts[len]--;
    len_counts[len + 1] += 2;
  }
}
static void gen_codewords(int *a, int *lens, const int *len_counts, int max_codeword_len, int num_syms) {
  int next_codewords[MAX_MAX_CODEWORD_LEN + 1];
  memset(next_codewords, 0, sizeof next_codewords);
  int i = 0;
  for (int len = max_codeword_len; len != 0; len--)
    for (int count = len_counts[len]; count != 0; count--) lens[a[i++] & SYMBOL_MASK] = len;
  next_codewords[0] = 0;
  next_codewords[1] = 0;
  for (int len = 2; len <= max_codeword_len; len++) next_codewords[len] = (next_codewords[len - 1] + len_counts[len - 1]) << 1;
  for (int sym = 0; sym < num_syms; sym++) a[sym] = next_codewords[lens[sym]]++;
}
static int reverse_codeword(int codeword, int len) {
  codeword = ((codeword & 0x5555) << 1) | ((codeword & 0xAAAA) >> 1);
  codeword = ((codeword & 0x3333) << 2) | ((codeword & 0xCCCC) >> 2);
  codeword = ((codeword & 0x0F0F) << 4) | ((codeword & 0xF0F0) >> 4);
  codeword = ((codeword & 0x00FF) << 8) | ((codeword & 0xFF00) >> 8);
  return (int)((unsigned)codeword >> (16 - len));
}
static void make_huffman_code(int num_syms, int max_codeword_len, const int *freqs, int *lens, int *codewords) {
  int num_used_syms = sort_symbols(num_syms, freqs, lens, codewords);
  if (num_used_syms == 1) {
    int sym = codewords[0] & SYMBOL_MASK;
    int nonzero_idx = sym > 1 ? sym : 1;
    codewords[0] = 0;
    lens[0] = 1;
    codewords[nonzero_idx] = 1;
    lens[nonzero_idx] = 1;
  } else if (num_used_syms > 1) {
    int len_counts[MAX_MAX_CODEWORD_LEN + 2];
    memset(len_counts, 0, sizeof len_counts);
    build_tree(codewords, num_used_syms);
    compute_length_counts(codewords, num_used_syms - 2, len_counts, max_codeword_len);
    gen_codewords(codewords, lens, len_counts, max_codeword_len, num_syms);
  }
  for (int sym = 0; sym < num_syms; sym++) codewords[sym] = reverse_codeword(codewords[sym] & 0xffff, lens[sym]);
}
static void make_huffman_codes(const lit_off *freqs, codes_t *codes) {
  make_huffman_code(NUM_LITLEN_SYMS, MAX_LITLEN_CODEWORD_LEN, freqs->litlen, codes->lens.litlen, codes->codewords.litlen);
  make_huffman_code(NUM_OFFSET_SYMS, MAX_OFFSET_CODEWORD_LEN, freqs->offset, codes->lens.offset, codes->codewords.offset);
}
static void init_static_codes(lit_off *freqs, codes_t *static_codes) { /* lib/de.ml:3331-3349 */
  for (int i = 0; i <= 143; i++) freqs->litlen[i] = 1 << (9 - 8);
  for (int i = 144; i <= 255; i++) freqs->litlen[i] = 1 << (9 - 9);
  for (int i = 256; i <= 279; i++) freqs->litlen[i] = 1 << (9 - 7);
  for (int i = 280; i <= 287; i++) freqs->litlen[i] = 1 << (9 - 8);
  for (int i = 0; i <= 31; i++) freqs->offset[i] = 1 << (5 - 5);
  make_huffman_codes(freqs, static_codes);
}

/* ---- output bitstream, lib/de.ml:3360-3432 ---- */
static void add_bits(os_t *os, uint64_t bits, int num_bits) {
  if (os->failed) return;
  os->hold |= bits << os->bits;
  os->bits += num_bits;
  if (os->bits >= 16) {
    if (os->o_pos + 1 >= os->o_len) {
      os->failed = 1;
      return;
    }
    os->o[os->o_pos] = (uint8_t)os->hold;
    os->o[os->o_pos + 1] = (uint8_t)(os->hold >> 8);
    os->o_pos += 2;
    os->bits -= 16;
    os->hold >>= 16;
  }
}
static void flush_bits(os_t *os) {
  if (os->failed) return;
  if (os->bits >= 8) {
    if (os->o_pos >= os->o_len) {
      os->failed = 1;
      return;
    }
    os->o[os->o_pos++] = (uint8_t)os->hold;
    os->bits -= 8;
    os->hold >>= 8;
  }
}
static void write_block_header(os_t *os, int is_final_block, int block_type) {
  add_bits(os, is_final_block ? 1 : 0, 1);
  add_bits(os, (uint64_t)block_type, 2);
}
static void align_bitstream(os_t *os) {
  os->bits += (-os->bits) & 7;
  flush_bits(os);
}
static void write_uncompressed_block(os_t *os, long len, int is_final_block) {
  write_block_header(os, is_final_block, 0);
  align_bitstream(os);
  if (os->failed) return;
  if (4 + len >= os->o_len - os->o_pos) {
    os->failed = 1;
    return;
  }
  os->o[os->o_pos] = (uint8_t)(len & 0xff);
  os->o[os->o_pos + 1] = (uint8_t)((len >> 8) & 0xff);
  os->o[os->o_pos + 2] = (uint8_t)(~len & 0xff);
  os->o[os->o_pos + 3] = (uint8_t)((~len >> 8) & 0xff);
  os->o_pos += 4;
  /* memcpy os.i ~src_off:os.i_pos: upstream reads `len` bytes whatever is left of the input; what lies beyond it is
   * undefined there (and the call then fails anyway, see write_uncompressed_blocks): zeros here */
  for (long k = 0; k < len; k++) os->o[os->o_pos + k] = os->i_pos + k < os->i_len ? os->i[os->i_pos + k] : 0;
  os->o_pos += len;
}
/* lib/de.ml:3411-3420 — (sic) i_pos is never advanced: with input left this can only end by `Unexpected_end_of_output */
static void write_uncompressed_blocks(os_t *os, long block_length, int is_final_block) {
  while (!os->failed && os->i_len - os->i_pos != 0) {
    long len = block_length < 65535 ? block_length : 65535;
    write_uncompressed_block(os, len, is_final_block && os->i_pos + len == os->i_len);
  }
}
static long flush_output(os_t *os) {
  while (!os->failed && os->bits > 0) {
    if (os->o_pos >= os->o_len) {
      os->failed = 1;
      break;
    }
    os->o[os->o_pos++] = (uint8_t)os->hold;
    os->bits -= 8;
    os->hold >>= 8;
  }
  return os->o_pos;
}

/* ---- dynamic header, lib/de.ml:3434-3556 ---- */
static int compute_precode_items(const int *lens, int num_lens, int *precode_freqs, int *precode_items) {
  memset(precode_freqs, 0, NUM_PRECODE_SYMS * sizeof(int));
  int itemptr = 0, run_start = 0;
  while (run_start != num_lens) {
    int len = lens[run_start], run_end = run_start;
    while (run_end != num_lens && len == lens[run_end]) run_end++;
    if (len == 0) {
      while (run_end - run_start >= 11) {
        int extra_bits = run_end - run_start - 11 < 0x7F ? run_end - run_start - 11 : 0x7F;
        precode_freqs[18]++;
        precode_items[itemptr++] = 18 | (extra_bits << 5);
        run_start += 11 + extra_bits;
      }
      if (run_end - run_start >= 3) {
        int extra_bits = run_end - run_start - 3 < 0x7 ? run_end - run_start - 3 : 0x7;
        precode_freqs[17]++;
        precode_items[itemptr++] = 17 | (extra_bits << 5);
        run_start += 3 + extra_bits;
      }
    } else if (run_end - run_start >= 4) {
      precode_freqs[len]++;
      precode_items[itemptr++] = len;
      run_start++;
      while (run_end - run_start >= 3) {
        int extra_bits = run_end - run_start - 3 < 0x3 ? run_end - run_start - 3 : 0x3;
        precode_freqs[16]++;
        precode_items[itemptr++] = 16 | (extra_bits << 5);
        run_start += 3 + extra_bits;
      }
    }
    while (run_start != run_end) {
      precode_freqs[len]++;
      precode_items[itemptr++] = len;
      run_start++;
    }
  }
  return itemptr;
}
static void precompute_huffman_header(encoder_t *c) {
  int n = NUM_LITLEN_SYMS;
  while (!(n == 257 || c->codes.lens.litlen[n - 1] != 0)) n--;
  c->num_litlen_syms = n;
  n = NUM_OFFSET_SYMS;
  while (!(n == 1 || c->codes.lens.offset[n - 1] != 0)) n--;
  c->num_offset_syms = n;
  const int nl = c->num_litlen_syms, no = c->num_offset_syms;
  if (nl != NUM_LITLEN_SYMS) {
    int max1 = (nl + no < NUM_LITLEN_SYMS ? nl + no : NUM_LITLEN_SYMS) - nl, max2 = no - max1;
    for (int i = 0; i < max1; i++) c->codes.lens.litlen[nl + i] = c->codes.lens.offset[i];
    for (int i = 0; i < max2; i++) c->codes.lens.offset[i] = c->codes.lens.offset[max1 + i];
  }
  int both[NUM_LITLEN_SYMS + NUM_OFFSET_SYMS]; /* Array.append litlen offset */
  memcpy(both, c->codes.lens.litlen, sizeof c->codes.lens.litlen);
  memcpy(both + NUM_LITLEN_SYMS, c->codes.lens.offset, sizeof c->codes.lens.offset);
  c->num_precode_items = compute_precode_items(both, nl + no, c->precode_freqs, c->precode_items);
  make_huffman_code(NUM_PRECODE_SYMS, MAX_PRE_CODEWORD_LEN, c->precode_freqs, c->precode_lens, c->precode_codewords);
  n = NUM_PRECODE_SYMS;
  while (n > 1 && c->precode_lens[zigzag[n - 1]] == 0) n--; /* (upstream would index -1 on an all-zero precode: cannot happen) */
  c->num_explicit_lens = n;
  if (nl != NUM_LITLEN_SYMS) {
    int max1 = NUM_LITLEN_SYMS - nl < no ? NUM_LITLEN_SYMS - nl : no;
    int max2 = no - max1 > 0 ? no - max1 : 0;
    for (int i = 0; i < max2; i++) c->codes.lens.offset[max1 + max2 - 1 - i] = c->codes.lens.offset[max2 - 1 - i];
    for (int i = 0; i < max1; i++) c->codes.lens.offset[i] = c->codes.lens.litlen[nl + i];
  }
}
static void write_huffman_header(encoder_t *c, os_t *os) {
  add_bits(os, (uint64_t)(c->num_litlen_syms - 257), 5);
  add_bits(os, (uint64_t)(c->num_offset_syms - 1), 5);
  add_bits(os, (uint64_t)(c->num_explicit_lens - 4) & 0xf, 4);
  for (int i = 0; i < c->num_explicit_lens; i++) add_bits(os, (uint64_t)c->precode_lens[zigzag[i]], 3);
  for (int i = 0; i < c->num_precode_items; i++) {
    int item = c->precode_items[i], sym = item & 0x1F;
    add_bits(os, (uint64_t)c->precode_codewords[sym], c->precode_lens[sym]);
    if (sym >= 16) add_bits(os, (uint64_t)(item >> 5), sym == 16 ? 2 : sym == 17 ? 3 : 7);
  }
}
static void write_sequences(os_t *os, const codes_t *codes, const seq_t *seqs, size_t nseq, const uint8_t *in_next, long *in_next_i) {
  for (size_t s = 0; s < nseq; s++) {
    int litrunlen = seqs[s].litrunlen_and_length & 0x7FFF, length = seqs[s].litrunlen_and_length >> 15;
    for (; litrunlen > 0; litrunlen--) {
      int lit = in_next[(*in_next_i)++];
      add_bits(os, (uint64_t)codes->codewords.litlen[lit], codes->lens.litlen[lit]);
    }
    if (length != 0) {
      *in_next_i += length;
      int slot = seqs[s].length_slot, sym = 257 + slot;
      add_bits(os, (uint64_t)codes->codewords.litlen[sym], codes->lens.litlen[sym]);
      add_bits(os, (uint64_t)(length - base_length[slot] - 3), extra_lbits[slot]);
      int osym = seqs[s].offset_symbol;
      add_bits(os, (uint64_t)codes->codewords.offset[osym], codes->lens.offset[osym]);
      add_bits(os, (uint64_t)(seqs[s].offset - base_dist[osym] - 1), extra_dbits[osym]);
    }
  }
}

/* flush_block, lib/de.ml:3622-3703 */
static void flush_block(encoder_t *c, os_t *os, long *block_begin, long block_length, int is_final_block, const seq_t *seqs, size_t nseq) {
  static const int extra_precode_bits[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
  long dynamic_cost = 0, static_cost = 0, uncompressed_cost = 0;
  c->freqs.litlen[END_OF_BLOCK]++;
  make_huffman_codes(&c->freqs, &c->codes);
  precompute_huffman_header(c);
  dynamic_cost += 5 + 5 + 4 + 3 * c->num_explicit_lens;
  for (int sym = 0; sym < NUM_PRECODE_SYMS; sym++) dynamic_cost += (long)c->precode_freqs[sym] * (extra_precode_bits[sym] + c->precode_lens[sym]);
  for (int sym = 0; sym <= 255; sym++) dynamic_cost += (long)c->freqs.litlen[sym] * c->codes.lens.litlen[sym];
  for (int sym = 0; sym <= 143; sym++) static_cost += (long)c->freqs.litlen[sym] * 8;
  for (int sym = 144; sym <= 255; sym++) static_cost += (long)c->freqs.litlen[sym] * 9;
  dynamic_cost += c->codes.lens.litlen[256];
  static_cost += 7;
  for (int sym = 257; sym <= 257 + 32 - 3; sym++) {
    int extra = extra_lbits[sym - 257];
    dynamic_cost += (long)c->freqs.litlen[sym] * (extra + c->codes.lens.litlen[sym]);
    static_cost += (long)c->freqs.litlen[sym] * (extra + c->static_codes.lens.litlen[sym]);
  }
  for (int sym = 0; sym <= 32 - 3; sym++) {
    int extra = extra_dbits[sym];
    dynamic_cost += (long)c->freqs.offset[sym] * (extra + c->codes.lens.offset[sym]);
    static_cost += (long)c->freqs.offset[sym] * (extra + 5);
  }
  uncompressed_cost += ((-(os->bits + 3)) & 7) + 32 + 40 * (((block_length + 65535 - 1) / 65535) - 1) + 8 * block_length;
  long ms = static_cost < uncompressed_cost ? static_cost : uncompressed_cost;
  int block_type = dynamic_cost < ms ? 2 : static_cost < uncompressed_cost ? 1 : 0;
  if (block_type == 0) {
    os->i_pos = *block_begin;
    write_uncompressed_blocks(os, block_length, is_final_block);
  } else {
    write_block_header(os, is_final_block, block_type);
    const codes_t *codes = &c->static_codes;
    if (block_type == 2) {
      write_huffman_header(c, os);
      codes = &c->codes;
    }
    write_sequences(os, codes, seqs, nseq, os->i, block_begin);
    add_bits(os, (uint64_t)codes->codewords.litlen[END_OF_BLOCK], codes->lens.litlen[END_OF_BLOCK]); /* write_end_of_block */
    flush_bits(os);
  }
}

/* ---- block splitting, lib/de.ml:3705-3751 ---- */
static int do_end_block_check(split_stats_t *st, long block_length) {
  if (st->num_observations > 0) {
    long total_delta = 0;
    for (int i = 0; i < 10; i++) {
      long expected = (long)st->observations[i] * st->num_new_observations;
      long actual = (long)st->new_observations[i] * st->num_observations;
      total_delta += actual > expected ? actual - expected : expected - actual;
    }
    if (total_delta + (block_length / 4096 * st->num_observations) >= (long)(512 * 200 / 512) * st->num_observations) return 1;
  }
  for (int i = 0; i < 10; i++) {
    st->num_observations += st->new_observations[i];
    st->observations[i] += st->new_observations[i];
    st->new_observations[i] = 0;
  }
  st->num_new_observations = 0;
  return 0;
}
static int should_end_block(split_stats_t *st, long in_block_begin, long in_next, long in_end) {
  if (st->num_new_observations < 512 || in_next - in_block_begin < MIN_BLOCK_LENGTH || in_end - in_next < MIN_BLOCK_LENGTH) return 0;
  return do_end_block_check(st, in_next - in_block_begin);
}

/* ---- hash-chain match finder, lib/de.ml:3753-3852 ---- */
static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static int lz_hash(const uint8_t *i, long pos, int num_bits) { return (int)((uint32_t)(rd32(i + pos) * 0x1E35A7BDu) >> (32 - num_bits)); }
static int lz_extend(const uint8_t *i, long start_pos, long match_pos, int len, int max_len) {
  while (max_len - len >= 4 && rd32(i + match_pos + len) == rd32(i + start_pos + len)) len += 4;
  while (len < max_len && i[match_pos + len] == i[start_pos + len]) len++;
  return len;
}
static void slide_window(hc_mf *mf) {
  /* entries only ever compare against cutoff >= -WINDOW_SIZE: keep them from running away on multi-GiB inputs */
  for (int k = 0; k < (1 << HASH4_ORDER); k++) mf->hash4_tab[k] = mf->hash4_tab[k] < INT_MIN / 2 ? mf->hash4_tab[k] : mf->hash4_tab[k] - WINDOW_SIZE;
  for (int k = 0; k < WINDOW_SIZE; k++) mf->next_tab[k] = mf->next_tab[k] < INT_MIN / 2 ? mf->next_tab[k] : mf->next_tab[k] - WINDOW_SIZE;
}
static long longest_match(hc_mf *mf, os_t *os, lens_t *lens, int max_search_depth) {
  long best_matchptr = os->i_pos;
  int cur_pos = (int)(os->i_pos & (WINDOW_SIZE - 1));
  if (cur_pos == 0 && os->i_pos != 0) slide_window(mf);
  int cutoff = cur_pos - WINDOW_SIZE;
  if (lens->max < 5) return os->i_pos - best_matchptr;
  int cur_node = mf->hash4_tab[mf->next_hash4];
  mf->hash4_tab[mf->next_hash4] = cur_pos;
  mf->next_tab[cur_pos] = cur_node;
  mf->next_hash4 = lz_hash(os->i, os->i_pos + 1, HASH4_ORDER);
  if (cur_node <= cutoff || lens->best >= lens->nice) return os->i_pos - best_matchptr;
  int depth_remaining = max_search_depth;
  for (;;) {
    long matchptr = (os->i_pos & ~(long)(WINDOW_SIZE - 1)) + cur_node;
    if (os->i[matchptr + lens->best] == os->i[os->i_pos + lens->best]) {
      int len = lz_extend(os->i, os->i_pos, matchptr, 0, lens->max);
      if (len >= lens->nice) {
        lens->best = len;
        return os->i_pos - matchptr;
      }
      if (len > lens->best) {
        lens->best = len;
        best_matchptr = matchptr;
      }
    }
    cur_node = mf->next_tab[cur_node & (WINDOW_SIZE - 1)];
    depth_remaining--;
    if (cur_node <= cutoff || depth_remaining == 0) return os->i_pos - best_matchptr;
  }
}
static void skip_positions(hc_mf *mf, os_t *os, int count) {
  if (count + 5 > os->i_len - os->i_pos) {
    os->i_pos += count;
    return;
  }
  for (; count > 0; count--) {
    int cur_pos = (int)(os->i_pos & (WINDOW_SIZE - 1));
    if (cur_pos == 0 && os->i_pos != 0) slide_window(mf);
    mf->next_tab[cur_pos] = mf->hash4_tab[mf->next_hash4];
    mf->hash4_tab[mf->next_hash4] = cur_pos;
    os->i_pos++;
    mf->next_hash4 = lz_hash(os->i, os->i_pos, HASH4_ORDER);
  }
}

/* compress_greedy, lib/de.ml:3875-3925 */
static long compress_greedy(encoder_t *c, os_t *os) {
  lens_t lens = {0, c->nice_match_length < MAX_MATCH_LEN ? c->nice_match_length : MAX_MATCH_LEN, MAX_MATCH_LEN};
  hc_mf mf;
  mf.hash4_tab = (int *)malloc(sizeof(int) << HASH4_ORDER);
  mf.next_tab = (int *)calloc(WINDOW_SIZE, sizeof(int));
  mf.next_hash4 = 0;
  for (int k = 0; k < (1 << HASH4_ORDER); k++) mf.hash4_tab[k] = -WINDOW_SIZE;
  split_stats_t st;
  size_t cap = 1024, nseq;
  seq_t *seqs = (seq_t *)malloc(cap * sizeof *seqs);
  while (!os->failed && os->i_pos != os->i_len) {
    long in_block_begin = os->i_pos;
    long rest = os->i_len - os->i_pos;
    long in_max_block_end = os->i_pos + (rest < SOFT_MAX_BLOCK_LENGTH ? rest : SOFT_MAX_BLOCK_LENGTH);
    int litrunlen = 0;
    nseq = 0;
    memset(&st, 0, sizeof st);
    memset(&c->freqs, 0, sizeof c->freqs);
    while (os->i_pos < in_max_block_end && !should_end_block(&st, in_block_begin, os->i_pos, os->i_len)) {
      if (lens.max > os->i_len - os->i_pos) {
        lens.max = (int)(os->i_len - os->i_pos);
        lens.nice = lens.nice < lens.max ? lens.nice : lens.max;
      }
      lens.best = MIN_MATCH_LEN - 1;
      long offset = longest_match(&mf, os, &lens, c->max_search_depth);
      if (lens.best >= MIN_MATCH_LEN) {
        if (nseq + 2 > cap) seqs = (seq_t *)realloc(seqs, (cap *= 2) * sizeof *seqs);
        int slot = length_slot_of[lens.best], oslot = offset_slot_fast[offset];
        c->freqs.litlen[257 + slot]++;
        c->freqs.offset[oslot]++;
        seqs[nseq++] = (seq_t){(lens.best << 15) | litrunlen, (int)offset, oslot, slot};
        litrunlen = 0;
        st.new_observations[8 + (lens.best >= 9 ? 1 : 0)]++; /* observe_match */
        st.num_new_observations++;
        os->i_pos++;
        skip_positions(&mf, os, lens.best - 1);
      } else {
        c->freqs.litlen[os->i[os->i_pos]]++; /* choose_literal */
        litrunlen++;
        /* observe_literal split_stats os.i_pos — (sic) the position, and `lsl 5`: the index is its lowest bit */
        st.new_observations[(int)(((os->i_pos << 5) & 0x6) | (os->i_pos & 1))]++;
        st.num_new_observations++;
        os->i_pos++;
      }
    }
    if (nseq + 2 > cap) seqs = (seq_t *)realloc(seqs, (cap *= 2) * sizeof *seqs);
    seqs[nseq++] = (seq_t){litrunlen, 0, 0, 0};
    flush_block(c, os, &in_block_begin, os->i_pos - in_block_begin, os->i_pos == os->i_len, seqs, nseq);
  }
  free(seqs);
  free(mf.hash4_tab);
  free(mf.next_tab);
  return flush_output(os);
}

/* De.Def.Ns.deflate ?level src dst (lib/de.ml:3999-4010).  Returns ORC_OK with *out_len = the `Ok n`,
 * ORC_UNEXPECTED_END_OF_OUTPUT, or -1 for `Invalid_compression_level. */
int orc_de_def_ns_deflate(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap, int level, size_t *out_len) {
  init_tables();
  *out_len = 0;
  if (level < 0 || level > 12) return -1;
  static const int depth[13] = {0, 2, 6, 12, 24, 20, 40, 100, 150, 200, 200, 200, 200};
  static const int nice[13] = {0, 8, 10, 14, 24, 30, 65, 130, 200, 258, 258, 258, 258};
  if (dst_cap < END_PADDING) return ORC_OK; /* Ok 0 */
  encoder_t *c = (encoder_t *)calloc(1, sizeof *c);
  c->level = level;
  c->min_size_to_compress = 56 - level * 4;
  c->max_search_depth = depth[level];
  c->nice_match_length = nice[level];
  init_static_codes(&c->freqs, &c->static_codes);
  os_t os = {src, 0, (long)n, dst, 0, (long)dst_cap - END_PADDING, 0, 0, 0};
  long res;
  if ((long)n < c->min_size_to_compress) {
    write_uncompressed_block(&os, os.i_len - os.i_pos, 1);
    res = flush_output(&os);
  } else if (level == 0) { /* compress_none */
    write_uncompressed_blocks(&os, os.o_len, 1);
    res = flush_output(&os);
  } else if (level <= 4) res = compress_greedy(c, &os);
  else res = 0; /* compress_lazy: "clecat: TO DO", lib/de.ml:3927 */
  free(c);
  if (os.failed) return ORC_UNEXPECTED_END_OF_OUTPUT;
  *out_len = (size_t)res;
  return ORC_OK;
}

/* De.Def.Ns.compress_bound (lib/de.ml:3994-3997) */
size_t orc_de_def_ns_compress_bound(size_t len) {
  size_t max_blocks = (len + MIN_BLOCK_LENGTH - 1) / MIN_BLOCK_LENGTH;
  if (max_blocks < 1) max_blocks = 1;
  return 5 * max_blocks + len + 1 + END_PADDING;
}

/* Zl.Def.Ns.deflate (lib/zl.ml:602-629): header (FLEVEL map 0|1 -> 0, 2..5 -> 1, 6 -> 2, else 3: H9), the body into
 * dst + 2, the Adler-32 of the input big-endian.  Upstream checks for 2 bytes of room before it writes the 4 of the
 * checksum (an out-of-bounds write with 2 or 3 left): `Unexpected_end_of_output here. */
int orc_zl_def_ns_deflate(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap, int level, size_t *out_len) {
  *out_len = 0;
  if (dst_cap < 2) return ORC_UNEXPECTED_END_OF_OUTPUT;
  unsigned header = (8 + ((15 - 8) << 4)) << 8;
  unsigned lv = (level == 0 || level == 1) ? 0 : (level >= 2 && level <= 5) ? 1 : level == 6 ? 2 : 3;
  header |= lv << 6;
  header += 31 - (header % 31);
  dst[0] = (uint8_t)(header >> 8);
  dst[1] = (uint8_t)header;
  size_t res = 0;
  int rc = orc_de_def_ns_deflate(src, n, dst + 2, dst_cap - 2, level, &res);
  if (rc != ORC_OK) return rc;
  if (dst_cap - 2 - res < 4) return ORC_UNEXPECTED_END_OF_OUTPUT;
  uint32_t a = orc_adler32(1, src, n);
  dst[2 + res] = (uint8_t)(a >> 24);
  dst[2 + res + 1] = (uint8_t)(a >> 16);
  dst[2 + res + 2] = (uint8_t)(a >> 8);
  dst[2 + res + 3] = (uint8_t)a;
  *out_len = res + 6;
  return ORC_OK;
}
