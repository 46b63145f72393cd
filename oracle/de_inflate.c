/* oracle/de_inflate.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of De.Inf.Ns (whole-buffer inflate, lib/de.ml:1534-1823),
 * its table builder De.Inf.huffman (lib/de.ml:523-638), the RFC1951 constant
 * tables (lib/de.ml:237-325) and Zl.Inf.Ns (lib/zl.ml:391-417).
 *
 * Documented divergences from the reference — all on malformed input only,
 * where the reference's own behaviour is an escaping OCaml exception or an
 * unbounded negative bit count:
 *  D1  De.Inf.Ns.__fill_bits (lib/de.ml:1643-1654) does not fail at end of
 *      input; the reference then keeps decoding zero bits with a negative bit
 *      count.  Here, consuming more bits than the input holds returns
 *      Unexpected_end_of_input (the streaming decoder's answer, lib/de.ml:762).
 *  D2  All-zero distance lengths give the 1-entry empty_table (lib/de.ml:521);
 *      indexing it with bit 1 raises Invalid_argument in OCaml.  Here:
 *      Invalid_distance_code.
 *  D3  HLIT > 286 / HDIST > 30 can overflow the fixed 852/592-entry tables
 *      (lib/de.ml:579-580) -> OCaml Invalid_argument.  Here: Invalid_dictionary.
 */
#include "oracle.h"
#include <string.h>

/* lib/de.ml:237-238 */
static const uint8_t zigzag[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5,
                                   11, 4, 12, 3, 13, 2, 14, 1, 15};
/* lib/de.ml:293-297 (31 entries: two zero pads after 255) */
static const int base_length[32] = {0,  1,  2,  3,  4,  5,  6,   7,   8,  10, 12,
                                    14, 16, 20, 24, 28, 32, 40,  48,  56, 64, 80,
                                    96, 112, 128, 160, 192, 224, 255, 0,  0,  0};
/* lib/de.ml:307-311 */
static const int extra_lbits[32] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                    3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0};
/* lib/de.ml:313-317 */
static const int extra_dbits[32] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,  4,
                                    4, 5, 5, 6, 6, 7, 7,  8,  8,  9,  9,
                                    10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
/* lib/de.ml:321-325 */
static const int base_dist[32] = {0,    1,    2,    3,    4,    6,     8,     12,
                                  16,   24,   32,   48,   64,   96,    128,   192,
                                  256,  384,  512,  768,  1024, 1536,  2048,  3072,
                                  4096, 6144, 8192, 12288, 16384, 24576, -1,  -1};

#define MAX_BITS 15
#define LINK_FLAG (1u << 20) /* lib/de.ml:520 */
#define K_CODES 0
#define K_LENS 1
#define K_DISTS 2

/* lib/de.ml:523-638 */
int orc_inf_huffman(int kind, const uint8_t *lens, int codes, uint32_t *tbl,
                    int *size_out, int *root_out, int *max_out) {
  int bl_count[16];
  int offs[16];
  uint16_t work[320];
  int max = 15, min = 1;
  memset(bl_count, 0, sizeof bl_count);
  for (int sym = 0; sym < codes; sym++) bl_count[lens[sym]]++;
  while (max >= 1 && bl_count[max] == 0) max--;
  if (max == 0) { /* empty_table, lib/de.ml:521,542 */
    tbl[0] = 1u << MAX_BITS;
    *size_out = 1;
    *root_out = 1;
    *max_out = 1;
    return 0;
  }
  int left = 1;
  for (int i = 1; i <= 15; i++) {
    left = (left << 1) - bl_count[i];
    if (left < 0) return -1;
  }
  if (left > 0 && (kind == K_CODES || max != 1)) return -1;
  while (min <= 15 && bl_count[min] == 0) min++;
  memset(offs, 0, sizeof offs);
  for (int idx = 1; idx <= 14; idx++) offs[idx + 1] = offs[idx] + bl_count[idx];
  for (int sym = 0; sym < codes; sym++) {
    int l = lens[sym];
    if (l != 0) work[offs[l]++] = (uint16_t)sym;
  }
  int root = kind == K_LENS ? 9 : kind == K_DISTS ? 6 : 7;
  if (root > max) root = max;
  if (root < min) root = min;
  int size;
  if (max <= root) size = 1 << max;
  else size = kind == K_LENS ? 852 : kind == K_DISTS ? 592 : (1 << max);
  memset(tbl, 0, (size_t)size * sizeof *tbl);

  unsigned huff = 0;
  int sym = 0, len = min, next = 0, curr = root, drop = 0;
  int low = -1;
  unsigned mask = (1u << root) - 1;
  int finished = 0;
  int cnt[16];
  memcpy(cnt, bl_count, sizeof cnt);
  while (!finished) {
    uint32_t entry = ((uint32_t)len << 15) | work[sym];
    int step = 1 << (len - drop);
    int fill_size = 1 << curr;
    int fill = fill_size;
    do {
      fill -= step;
      int idx = next + (int)(huff >> drop) + fill;
      if (idx >= size) return -1; /* D3 */
      tbl[idx] = entry;
    } while (fill != 0);
    unsigned inc = 1u << (len - 1);
    while (huff & inc) inc >>= 1;
    huff = inc != 0 ? (huff & (inc - 1)) + inc : 0;
    sym++;
    if (--cnt[len] == 0) {
      if (len == max) finished = 1;
      else len = lens[work[sym]];
    }
    if (!finished && len > root && (int)(huff & mask) != low) {
      if (drop == 0) drop = root;
      next += fill_size;
      curr = len - drop;
      int l2 = 1 << curr;
      while (curr + drop < max) {
        l2 -= cnt[curr + drop];
        if (l2 <= 0) break;
        curr++;
        l2 <<= 1;
      }
      low = (int)(huff & mask);
      if (next + (1 << curr) > size) return -1; /* D3 */
      tbl[low] = LINK_FLAG | ((uint32_t)curr << 15) | (uint32_t)next;
    }
  }
  *size_out = size;
  *root_out = root;
  *max_out = max;
  return 0;
}

typedef struct {
  uint32_t t[852];
  int size, m, l, root;
} lookup_t; /* Lookup.t, lib/de.ml:349-371 */

typedef struct {
  const uint8_t *i;
  size_t i_pos, i_len;
  uint64_t hold;
  int bits;
  uint8_t *o;
  size_t o_pos, o_len;
} ns_t; /* lib/de.ml:1535-1544 */

#define TRY(e)            \
  do {                    \
    int rc_ = (e);        \
    if (rc_) return rc_;  \
  } while (0)

/* lib/de.ml:1643-1654 (never fails) */
static void fill_bits_nofail(ns_t *d, int n) {
  if (d->bits < n) {
    size_t rem = d->i_len - d->i_pos;
    if (rem > 1) {
      uint64_t v = (uint64_t)d->i[d->i_pos] | ((uint64_t)d->i[d->i_pos + 1] << 8);
      d->hold |= v << d->bits;
      d->i_pos += 2;
      d->bits += 16;
    } else if (rem == 1) {
      d->hold |= (uint64_t)d->i[d->i_pos] << d->bits;
      d->i_pos += 1;
      d->bits += 8;
    }
  }
}
/* lib/de.ml:1629-1641 */
static int fill_bits(ns_t *d, int n) {
  if (d->bits < n) {
    if (d->i_len - d->i_pos == 0) return ORC_UNEXPECTED_END_OF_INPUT;
    fill_bits_nofail(d, n);
  }
  return 0;
}
/* lib/de.ml:1656-1661; D1: popping bits the input does not hold is EOI */
static int pop_bits(ns_t *d, int n, int *v) {
  if (d->bits < n) return ORC_UNEXPECTED_END_OF_INPUT;
  *v = (int)(d->hold & ((1ull << n) - 1));
  d->hold >>= n;
  d->bits -= n;
  return 0;
}
/* lib/de.ml:640-647; returns 0 and *e, or -1 when out of the table (D2) */
static int resolve(const lookup_t *lk, uint64_t hold, uint32_t *e) {
  uint32_t idx = (uint32_t)(hold & (uint64_t)lk->m);
  if ((int)idx >= lk->size) return -1;
  uint32_t v = lk->t[idx];
  if (v & LINK_FLAG) {
    int sub = (v >> 15) & 0x1f;
    idx = (v & 0x7fff) + (uint32_t)((hold >> lk->root) & ((1u << sub) - 1));
    if ((int)idx >= lk->size) return -1;
    v = lk->t[idx];
  }
  *e = v;
  return 0;
}

/* _blit, lib/de.ml:1595-1611: forward copy, 32-bit words when dst-src >= 4,
 * bytes otherwise; either way equivalent to a forward byte copy. */
static void blit_fwd(uint8_t *o, size_t src, size_t dst, size_t len) {
  for (size_t k = 0; k < len; k++) o[dst + k] = o[src + k];
}

/* lib/de.ml:1667-1712 */
static int ns_inflate_block(ns_t *d, const lookup_t *lit, const lookup_t *dist) {
  for (;;) {
    uint32_t code;
    fill_bits_nofail(d, lit->l);
    if (resolve(lit, d->hold, &code)) return ORC_INVALID_DICTIONARY;
    int value = code & 0x7fff;
    int len = code >> 15;
    if (d->bits < len) return ORC_UNEXPECTED_END_OF_INPUT; /* D1 */
    d->hold >>= len;
    d->bits -= len;
    if (value < 256) {
      if (d->o_pos >= d->o_len) return ORC_UNEXPECTED_END_OF_OUTPUT;
      d->o[d->o_pos++] = (uint8_t)value;
    } else if (value == 256) {
      return 0;
    } else {
      int l = value - 257;
      int extra;
      int xl = extra_lbits[l & 0x1f];
      fill_bits_nofail(d, xl);
      TRY(pop_bits(d, xl, &extra));
      l = base_length[l & 0x1f] + 3 + extra;
      fill_bits_nofail(d, dist->l);
      if (resolve(dist, d->hold, &code)) return ORC_INVALID_DISTANCE_CODE; /* D2 */
      int dv = code & 0x7fff;
      len = code >> 15;
      if (d->bits < len) return ORC_UNEXPECTED_END_OF_INPUT; /* D1 */
      d->hold >>= len;
      d->bits -= len;
      int xd = extra_dbits[dv & 0x1f];
      fill_bits_nofail(d, xd);
      TRY(pop_bits(d, xd, &extra));
      int dd = base_dist[dv & 0x1f] + 1 + extra;
      if (dd == 0) return ORC_INVALID_DISTANCE_CODE;
      size_t lim = d->o_pos < 32768 ? d->o_pos : 32768;
      if ((size_t)dd > lim) return ORC_INVALID_DISTANCE;
      if ((size_t)l > d->o_len - d->o_pos) return ORC_UNEXPECTED_END_OF_OUTPUT;
      blit_fwd(d->o, d->o_pos - (size_t)dd, d->o_pos, (size_t)l);
      d->o_pos += (size_t)l;
    }
  }
}

/* lib/de.ml:1613-1627 */
static int ns_flat(ns_t *d) {
  d->i_pos -= (size_t)(d->bits / 8);
  d->hold = 0;
  d->bits = 0;
  if (d->i_len - d->i_pos < 4) return ORC_UNEXPECTED_END_OF_INPUT;
  unsigned len = d->i[d->i_pos] | (d->i[d->i_pos + 1] << 8);
  unsigned nlen = d->i[d->i_pos + 2] | (d->i[d->i_pos + 3] << 8);
  d->i_pos += 4;
  if (nlen != 0xffff - len) return ORC_INVALID_COMPLEMENT_OF_LENGTH;
  if (len > d->i_len - d->i_pos) return ORC_UNEXPECTED_END_OF_INPUT;
  if (len > d->o_len - d->o_pos) return ORC_UNEXPECTED_END_OF_OUTPUT;
  memcpy(d->o + d->o_pos, d->i + d->i_pos, len);
  d->o_pos += len;
  d->i_pos += len;
  return 0;
}

/* fixed_lit / fixed_dist, lib/de.ml:821-833 */
static void fixed_tables(lookup_t *lit, lookup_t *dist) {
  uint8_t l[288];
  for (int n = 0; n < 288; n++) l[n] = n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8;
  orc_inf_huffman(K_LENS, l, 288, lit->t, &lit->size, &lit->root, &lit->l);
  lit->m = (1 << lit->root) - 1;
  for (int i = 0; i < 32; i++) {
    unsigned r = 0, v = (unsigned)i << 3; /* reverse_bits (i lsl 3) */
    for (int b = 0; b < 8; b++) r |= ((v >> b) & 1u) << (7 - b);
    dist->t[i] = (5u << 15) | r;
  }
  dist->size = 32;
  dist->root = 5;
  dist->l = 5;
  dist->m = 31;
}

/* lib/de.ml:1718-1793: dynamic -> table -> inflate_table -> make_table */
static int ns_dynamic(ns_t *d) {
  int hlit, hdist, hclen, v;
  TRY(fill_bits(d, 14));
  TRY(pop_bits(d, 5, &hlit));
  TRY(pop_bits(d, 5, &hdist));
  TRY(pop_bits(d, 4, &hclen));
  hlit += 257;
  hdist += 1;
  hclen += 4;
  uint8_t cl[19];
  memset(cl, 0, sizeof cl);
  for (int i = 0; i < hclen; i++) {
    TRY(fill_bits(d, 3));
    TRY(pop_bits(d, 3, &v));
    cl[zigzag[i]] = (uint8_t)v;
  }
  uint32_t ct[128];
  int csize, croot, cmax;
  if (orc_inf_huffman(K_CODES, cl, 19, ct, &csize, &croot, &cmax))
    return ORC_INVALID_DICTIONARY;
  /* inflate_table, lib/de.ml:1733-1769 */
  uint8_t res[320];
  memset(res, 0, sizeof res);
  int max_res = hlit + hdist;
  unsigned cmask = (1u << cmax) - 1;
  int i = 0;
  while (i < max_res) {
    TRY(fill_bits(d, cmax));
    uint32_t idx = (uint32_t)(d->hold & cmask);
    if ((int)idx >= csize) return ORC_INVALID_DICTIONARY; /* empty_table OOB */
    uint32_t e = ct[idx];
    int sym = e & 0x7fff, len = e >> 15;
    if (d->bits < len) return ORC_UNEXPECTED_END_OF_INPUT;
    d->hold >>= len;
    d->bits -= len;
    if (sym < 16) {
      res[i++] = (uint8_t)sym;
    } else {
      int copy, val;
      if (sym == 16) {
        if (i == 0) return ORC_INVALID_DICTIONARY;
        TRY(fill_bits(d, 2));
        TRY(pop_bits(d, 2, &v));
        copy = v + 3;
        val = res[i - 1];
      } else if (sym == 17) {
        TRY(fill_bits(d, 3));
        TRY(pop_bits(d, 3, &v));
        copy = v + 3;
        val = 0;
      } else {
        TRY(fill_bits(d, 7));
        TRY(pop_bits(d, 7, &v));
        copy = v + 11;
        val = 0;
      }
      if (i + copy > max_res) return ORC_INVALID_DICTIONARY;
      for (int x = 0; x < copy; x++) res[i + x] = (uint8_t)val;
      i += copy;
    }
  }
  /* make_table, lib/de.ml:1718-1731 */
  if (res[256] == 0) return ORC_INVALID_DICTIONARY;
  static _Thread_local lookup_t lit, dist;
  if (orc_inf_huffman(K_LENS, res, hlit, lit.t, &lit.size, &lit.root, &lit.l))
    return ORC_INVALID_DICTIONARY;
  lit.m = (1 << lit.root) - 1;
  if (orc_inf_huffman(K_DISTS, res + hlit, hdist, dist.t, &dist.size, &dist.root, &dist.l))
    return ORC_INVALID_DICTIONARY;
  dist.m = (1 << dist.root) - 1;
  return ns_inflate_block(d, &lit, &dist);
}

/* lib/de.ml:1795-1822 */
int orc_de_inf_ns_inflate(const uint8_t *src, size_t src_len, uint8_t *dst,
                          size_t dst_cap, size_t *consumed, size_t *written) {
  ns_t d = {src, 0, src_len, 0, 0, dst, 0, dst_cap};
  static _Thread_local lookup_t flit, fdist;
  static _Thread_local int fixed_ready = 0;
  *consumed = 0;
  *written = 0;
  for (;;) {
    int last = 0, type = 0, rc;
    /* a failure in the 3 header bits reports what the earlier blocks wrote, like a failure inside a block
     * (the reference returns `Error e` without counts either way; the C ABI reports the bytes produced) */
    rc = fill_bits(&d, 3);
    if (!rc) rc = pop_bits(&d, 1, &last);
    if (!rc) rc = pop_bits(&d, 2, &type);
    if (rc) {
      *written = d.o_pos;
      return rc;
    }
    switch (type) {
    case 0: rc = ns_flat(&d); break;
    case 1:
      if (!fixed_ready) {
        fixed_tables(&flit, &fdist);
        fixed_ready = 1;
      }
      rc = ns_inflate_block(&d, &flit, &fdist);
      break;
    case 2: rc = ns_dynamic(&d); break;
    default: rc = ORC_INVALID_KIND_OF_BLOCK; break;
    }
    if (rc) {
      *written = d.o_pos;
      return rc;
    }
    if (last) {
      d.i_pos -= (size_t)(d.bits >> 3);
      break;
    }
  }
  *consumed = d.i_pos;
  *written = d.o_pos;
  return ORC_OK;
}

/* RFC1950 Adler-32 (checkseum; call sites lib/de.ml:453-455, lib/zl.ml:414) */
uint32_t orc_adler32(uint32_t adler, const uint8_t *buf, size_t len) {
  uint32_t a = adler & 0xffff, b = adler >> 16;
  while (len) {
    size_t n = len < 5552 ? len : 5552;
    len -= n;
    while (n--) {
      a += *buf++;
      b += a;
    }
    a %= 65521;
    b %= 65521;
  }
  return (b << 16) | a;
}

/* RFC1952 CRC-32 (checkseum; call sites lib/gz.ml:428) */
uint32_t orc_crc32(uint32_t crc, const uint8_t *buf, size_t len) {
  static uint32_t tab[256];
  static int ready = 0;
  if (!ready) {
    for (uint32_t n = 0; n < 256; n++) {
      uint32_t c = n;
      for (int k = 0; k < 8; k++) c = c & 1 ? 0xedb88320u ^ (c >> 1) : c >> 1;
      tab[n] = c;
    }
    ready = 1;
  }
  crc = ~crc;
  while (len--) crc = tab[(crc ^ *buf++) & 0xff] ^ (crc >> 8);
  return ~crc;
}

/* Zl.Inf.Ns.inflate, lib/zl.ml:391-417 */
int orc_zl_inf_ns_inflate(const uint8_t *src, size_t src_len, uint8_t *dst,
                          size_t dst_cap, size_t *consumed, size_t *written) {
  *consumed = 0;
  *written = 0;
  if (src_len < 2) return ORC_UNEXPECTED_END_OF_INPUT;
  unsigned cmf = src[0], flg = src[1];
  if (((cmf << 8) + flg) % 31 != 0 || (cmf & 0xf) != 8) return ORC_INVALID_HEADER;
  /* bigstring_sub src 2 (len - 6): OCaml raises when len < 6; here EOI */
  if (src_len < 6) return ORC_UNEXPECTED_END_OF_INPUT;
  size_t i, o;
  int rc = orc_de_inf_ns_inflate(src + 2, src_len - 6, dst, dst_cap, &i, &o);
  *written = o;
  if (rc) return rc;
  if (src_len < i + 6) return ORC_UNEXPECTED_END_OF_INPUT;
  uint32_t want = ((uint32_t)src[i + 2] << 24) | ((uint32_t)src[i + 3] << 16) |
                  ((uint32_t)src[i + 4] << 8) | src[i + 5];
  if (want != orc_adler32(1, dst, o)) return ORC_INVALID_CHECKSUM;
  *consumed = i + 6;
  return ORC_OK;
}

/* bench.py's cpu_baseline leg: n zlib streams of one blob inflated back to back into one scratch
 * buffer (one call per worker thread, so that the timing is C code, not the Python binding) */
size_t orc_zl_inf_ns_inflate_batch(const uint8_t *blob, const uint64_t *off, const uint64_t *len, size_t n,
                                   uint8_t *scratch, size_t cap, uint64_t *total_out) {
  size_t bad = 0;
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) {
    size_t used, wrote;
    if (orc_zl_inf_ns_inflate(blob + off[i], (size_t)len[i], scratch, cap, &used, &wrote) != ORC_OK) bad++;
    total += wrote;
  }
  *total_out = total;
  return bad;
}

/* error strings: lib/de.ml:1557-1567, lib/zl.ml:385-389 */
const char *orc_status_string(int s) {
  switch (s) {
  case ORC_OK: return "Ok";
  case ORC_UNEXPECTED_END_OF_INPUT: return "Unexpected end of input";
  case ORC_UNEXPECTED_END_OF_OUTPUT: return "Unexpected end of output";
  case ORC_INVALID_KIND_OF_BLOCK: return "Invalid kind of block";
  case ORC_INVALID_DICTIONARY: return "Invalid dictionary";
  case ORC_INVALID_COMPLEMENT_OF_LENGTH: return "Invalid complement of length";
  case ORC_INVALID_DISTANCE: return "Invalid distance";
  case ORC_INVALID_DISTANCE_CODE: return "Invalid distance code";
  case ORC_INVALID_HEADER: return "Invalid header";
  case ORC_INVALID_CHECKSUM: return "Invalid checksum";
  case ORC_INVALID_GZIP_HEADER: return "Invalid GZip header";
  case ORC_INVALID_GZIP_HEADER_CHECKSUM: return "Invalid GZip header checksum";
  case ORC_INVALID_SIZE: return "Invalid input size";
  case ORC_QUEUE_FULL: return "Queue.Full";
  case ORC_LZO_INVALID_INPUT: return "Invalid input";
  case ORC_LZO_NO_DICTIONARY: return "No dictionary at offset 0 available";
  case ORC_LZO_OUT_OF_BOUND: return "Input is malformed or output is not large enough";
  default: return "?";
  }
}
