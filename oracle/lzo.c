/* oracle/lzo.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's LZO1X codec (mirage/decompress v1.6.0, lib/lzo.ml):
 *   orc_lzo_uncompress  Lzo.uncompress (lib/lzo.ml:395-403): the instruction interpreter `run`
 *                       (:246-293) over `fiber` (:315-393), `count` (:218-236)
 *   orc_lzo_compress    Lzo.compress (lib/lzo.ml:594-660): lzo1x-1, 16 K-entry u16 dictionary,
 *                       49 152-byte chunks, record_literals / record_match / record_trailer
 *                       (:443-592)
 * Reference behaviour kept on purpose (it differs from liblzo on streams lzo1x-1 never emits):
 *   - the decoder state after a literal run is -1 and `-1 land 3 = 3`, so a byte < 16 after a
 *     literal run is a 2-byte match at offset <= 0x400 (:334-338; the `-1` arm :339-343 is
 *     unreachable), and after the first-byte literal run the state is 0 (:376-393);
 *   - every instruction but State/Return fails with "Unexpected end of input" once the input
 *     position has reached the end (:273-274);
 *   - any out-of-bounds access is `Invalid_argument "Input is malformed or output is not large
 *     enough"` (:401-402);
 *   - record_trailer takes the short first-byte form only for len < 238 (:565; liblzo: <= 238).
 * One divergence: the reference's `copy` into a bigstring moves 4 bytes at a time (unsafe_blit,
 * :65-79), which reads bytes it has not written yet when the offset is below 4 — the result then
 * depends on what the caller's output buffer held.  Here (as in Lzo.uncompress_with_buffer,
 * :206-216, and in liblzo) a match is the usual byte-serial LZ77 copy.
 * Parity pinning: the reference's vector (test/test.ml:2033-2065) and, as the reference's own
 * tests and fuzzers do (test/test.ml:2067-2097, fuzz/fuzz_lzo.ml), cross-checks against minilzo
 * — built from the reference tree into oracle/_ref/libminilzo.so.
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t ld64(const uint8_t *p) { return (uint64_t)ld32(p) | ((uint64_t)ld32(p + 4) << 32); }

/* ------------------------------------------------------------------ uncompress */
typedef struct {
  const uint8_t *i;
  size_t i_len, i_pos;
  uint8_t *o;
  size_t o_len, o_pos;
  int state;
} lzo_v;

/* transmit (lib/lzo.ml:188-192) with blit's bounds (:81-89) */
static int lz_transmit(lzo_v *v, size_t len) {
  if (v->i_pos > v->i_len || len > v->i_len - v->i_pos || len > v->o_len - v->o_pos) return ORC_LZO_OUT_OF_BOUND;
  memcpy(v->o + v->o_pos, v->i + v->i_pos, len);
  v->i_pos += len;
  v->o_pos += len;
  return ORC_OK;
}
/* copy (lib/lzo.ml:194-197): byte-serial, see the header */
static int lz_copy(lzo_v *v, size_t off, size_t len) {
  if (off > v->o_pos || len > v->o_len - v->o_pos) return ORC_LZO_OUT_OF_BOUND;
  for (size_t k = 0; k < len; k++) v->o[v->o_pos + k] = v->o[v->o_pos - off + k];
  v->o_pos += len;
  return ORC_OK;
}
/* count (lib/lzo.ml:218-236) */
static int lz_count(lzo_v *v, size_t *out) {
  size_t res = 0, idx = v->i_pos, max = v->i_len;
  while (idx + 4 <= max && ld32(v->i + idx) == 0) {
    idx += 4;
    res += 4;
  }
  while (idx + 1 <= max && v->i[idx] == 0) {
    idx++;
    res++;
  }
  if (idx < max) {
    v->i_pos = idx + 1;
    *out = res * 255 + v->i[idx];
    return ORC_OK;
  }
  return ORC_LZO_INVALID_INPUT;
}
#define EOI_GUARD() do { if (v.i_pos >= v.i_len) return ORC_UNEXPECTED_END_OF_INPUT; } while (0)
#define TRY(e) do { int rc_ = (e); if (rc_) return rc_; } while (0)

int orc_lzo_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *written) {
  lzo_v v = {src, n, 0, dst, cap, 0, 0};
  *written = 0;
  /* the first byte, lib/lzo.ml:372-393 */
  EOI_GUARD();
  unsigned chr = src[0];
  if (chr == 16) return ORC_LZO_NO_DICTIONARY;
  if (chr >= 18) {
    v.i_pos = 1;
    v.state = 0;
    EOI_GUARD(); /* Transmit is guarded like every instruction, :273-274 */
    TRY(lz_transmit(&v, chr - 17));
  }
  for (;;) { /* fiber, lib/lzo.ml:315-369 */
    EOI_GUARD();
    chr = v.i[v.i_pos++];
    const int st = v.state & 3; /* -1 land 3 = 3 */
    size_t len, off, cnt;
    int nstate;
    if (chr < 16 && st == 0) {
      if (chr == 0) {
        EOI_GUARD();
        TRY(lz_count(&v, &cnt));
        len = 3 + 15 + cnt;
      } else len = chr + 3;
      v.state = -1;
      EOI_GUARD();
      TRY(lz_transmit(&v, len));
      continue;
    }
    if (chr < 16) { /* st in 1..3 */
      EOI_GUARD();
      const unsigned h = v.i[v.i_pos++];
      off = ((size_t)h << 2) + (chr >> 2) + 1;
      len = 0;
      nstate = chr & 3;
    } else if (chr < 32) {
      len = chr & 7;
      if (len == 0) {
        EOI_GUARD();
        TRY(lz_count(&v, &cnt));
        len = 7 + cnt;
      }
      EOI_GUARD();
      if (v.i_pos + 2 > v.i_len) return ORC_LZO_OUT_OF_BOUND; /* get_int16 / Junk Short */
      const unsigned s = v.i[v.i_pos] | (v.i[v.i_pos + 1] << 8);
      v.i_pos += 2;
      off = 16384 + ((size_t)((chr & 8) >> 3) << 14) + (s >> 2);
      nstate = s & 0xff;
      if (off == 16384) break; /* end_of_lzo */
    } else if (chr < 64) {
      len = chr & 31;
      if (len == 0) {
        EOI_GUARD();
        TRY(lz_count(&v, &cnt));
        len = 31 + cnt;
      }
      EOI_GUARD();
      if (v.i_pos + 2 > v.i_len) return ORC_LZO_OUT_OF_BOUND;
      const unsigned s = v.i[v.i_pos] | (v.i[v.i_pos + 1] << 8);
      v.i_pos += 2;
      nstate = s & 0xff;
      off = (s >> 2) + 1;
    } else {
      nstate = chr;
      len = (chr >> 5) - 1;
      EOI_GUARD();
      const unsigned h = v.i[v.i_pos++];
      off = ((size_t)h << 3) + ((chr >> 2) & 7) + 1;
    }
    /* Copy (:283-288): copy len + 2 bytes, then copy_done = transmit (state land 3) */
    EOI_GUARD();
    v.state = nstate;
    TRY(lz_copy(&v, off, len + 2));
    TRY(lz_transmit(&v, (size_t)(nstate & 3)));
  }
  *written = v.o_pos;
  return ORC_OK;
}

/* ------------------------------------------------------------------ compress */
typedef struct {
  const uint8_t *in;
  size_t in_total;
  uint8_t *out;
  size_t out_cap;
  int oob; /* Out_of_bound: "lzo: output is not large enough" (lib/lzo.ml:655) */
} lzo_c;

static void c_set(lzo_c *c, long op, unsigned b) {
  if (op < 0 || (size_t)op >= c->out_cap) c->oob = 1;
  else c->out[op] = (uint8_t)b;
}
static unsigned c_get(lzo_c *c, long op) {
  if (op < 0 || (size_t)op >= c->out_cap) {
    c->oob = 1;
    return 0;
  }
  return c->out[op];
}
static void c_blit(lzo_c *c, size_t off, long op, size_t len) { /* blit in_data off out_data op len, :81-89 */
  if (off > c->in_total || len > c->in_total - off || op < 0 || (size_t)op > c->out_cap || len > c->out_cap - (size_t)op) {
    c->oob = 1;
    return;
  }
  memmove(c->out + op, c->in + off, len);
}
/* record_match, lib/lzo.ml:443-500 */
static long record_match(lzo_c *c, size_t off, size_t len, long op) {
  if (len <= 8 && off <= 0x0800) {
    off -= 1;
    c_set(c, op++, (unsigned)(((len - 1) << 5) | ((off & 7) << 2)));
    c_set(c, op++, (unsigned)(off >> 3));
  } else if (off <= 0x4000) {
    off -= 1;
    if (len <= 33) c_set(c, op++, (unsigned)(32 | (len - 2)));
    else {
      size_t l = len - 33;
      c_set(c, op++, 32);
      while (l > 255) {
        l -= 255;
        c_set(c, op++, 0);
      }
      c_set(c, op++, (unsigned)l);
    }
    c_set(c, op++, (unsigned)((off << 2) & 0xff));
    c_set(c, op++, (unsigned)((off >> 6) & 0xff));
  } else {
    off -= 0x4000;
    if (len <= 9) c_set(c, op++, (unsigned)(16 | ((off >> 11) & 8) | (len - 2)));
    else {
      size_t l = len - 9;
      c_set(c, op++, (unsigned)(16 | ((off >> 11) & 8)));
      while (l > 255) {
        l -= 255;
        c_set(c, op++, 0);
      }
      c_set(c, op++, (unsigned)l);
    }
    c_set(c, op++, (unsigned)((off << 2) & 0xff));
    c_set(c, op++, (unsigned)((off >> 6) & 0xff));
  }
  return op;
}
static long long_run(lzo_c *c, size_t len, long op) { /* 18+ literals: 0, 0..., rest */
  size_t l = len - 18;
  c_set(c, op++, 0);
  while (l > 255) {
    l -= 255;
    c_set(c, op++, 0);
  }
  c_set(c, op++, (unsigned)l);
  return op;
}
/* record_literals, lib/lzo.ml:502-538 (the 4 / 16-byte over-copies need their room too) */
static long record_literals(lzo_c *c, size_t off, size_t len, long op) {
  if (len == 0) return op;
  if (len <= 3) {
    c_set(c, op - 2, c_get(c, op - 2) | (unsigned)len);
    c_blit(c, off, op, 4);
    return op + (long)len;
  }
  if (len <= 16) {
    c_set(c, op++, (unsigned)(len - 3));
    c_blit(c, off, op, 8);
    c_blit(c, off + 8, op + 8, 8);
    return op + (long)len;
  }
  if (len <= 18) c_set(c, op++, (unsigned)(len - 3));
  else op = long_run(c, len, op);
  c_blit(c, off, op, len);
  return op + (long)len;
}
/* record_trailer, lib/lzo.ml:540-576 */
static long record_trailer(lzo_c *c, size_t off, size_t len, long op) {
  if (len > 0) {
    if (op == 0 && len < 238) c_set(c, op++, (unsigned)(17 + len));
    else if (len <= 3) c_set(c, op - 2, c_get(c, op - 2) | (unsigned)len);
    else if (len <= 18) c_set(c, op++, (unsigned)(len - 3));
    else op = long_run(c, len, op);
    c_blit(c, off, op, len);
  }
  op += (long)len;
  c_set(c, op++, 16 | 1);
  c_set(c, op++, 0);
  c_set(c, op++, 0);
  return op;
}
static const int ctz_index[64] = {0,  1,  2,  53, 3,  7,  54, 27, 4,  38, 41, 8,  34, 55, 48, 28, 62, 5,  39, 46, 44, 42,
                                  22, 9,  24, 35, 59, 56, 49, 18, 29, 11, 63, 52, 6,  26, 37, 40, 33, 47, 61, 45, 43, 21,
                                  23, 58, 17, 10, 51, 25, 36, 32, 60, 20, 57, 16, 50, 31, 19, 15, 30, 14, 13, 12};
static int lzo_ctz(uint64_t v) { /* lib/lzo.ml:436-442 (0 -> index.(0) = 0) */
  return ctz_index[((v & (0 - v)) * 0x022fdd63cc95386dull) >> 58];
}
/* the chunk compressor, lib/lzo.ml:578-640: returns the unrecorded tail length, *pop = out position */
static size_t compress_chunk(lzo_c *c, size_t in_pos, size_t in_len, long *pop, size_t t, uint16_t *wrkmem) {
  const size_t idx_end = in_len > 20 ? in_len - 20 : 0;
  long op = *pop;
  size_t idx0 = in_pos + (t < 4 ? 4 - t : 0), idx1 = in_pos;
  for (;;) {
    idx0 += 1 + ((idx0 - idx1) >> 5); /* literal: */
    for (;;) {                        /* next: */
      if (idx0 - in_pos >= idx_end) {
        idx1 -= t;
        *pop = op;
        return in_len - (idx1 - in_pos);
      }
      if (idx0 + 4 > c->in_total) { /* get_int32 raises Out_of_bound */
        c->oob = 1;
        *pop = op;
        return 0;
      }
      const uint32_t v = ld32(c->in + idx0);
      const uint32_t index = ((uint32_t)(0x1824429du * v) >> 18) & 0x3fff;
      const size_t ref = wrkmem[index] + in_pos;
      wrkmem[index] = (uint16_t)(idx0 - in_pos);
      if (v != ld32(c->in + ref)) break; /* -> literal */
      idx1 -= t;
      t = 0;
      op = record_literals(c, idx1, idx0 - idx1, op);
      size_t len = 4;
      while (idx0 + len - in_pos < idx_end && ld64(c->in + idx0 + len) == ld64(c->in + ref + len)) len += 8;
      if (idx0 + len - in_pos < in_len) {
        if (idx0 + len + 8 > c->in_total) { /* get_int64 raises Out_of_bound */
          c->oob = 1;
          *pop = op;
          return 0;
        }
        len += (size_t)lzo_ctz(ld64(c->in + idx0 + len) ^ ld64(c->in + ref + len)) / 8;
      }
      op = record_match(c, idx0 - ref, len, op);
      idx0 += len;
      idx1 = idx0;
    }
  }
}

/* Lzo.compress in_data out_data wrkmem (lib/lzo.ml:642-660).  Returns ORC_OK and *out_len, or
 * ORC_LZO_OUT_OF_BOUND ("lzo: output is not large enough"). */
int orc_lzo_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len) {
  lzo_c c = {src, n, dst, cap, 0};
  uint16_t *wrkmem = (uint16_t *)malloc(16384 * sizeof(uint16_t));
  size_t idx = 0, len = n, t = 0;
  long op = 0;
  *out_len = 0;
  while (len > 20) {
    const size_t ll = len < 49152 ? len : 49152;
    if (((t + ll) >> 5) == 0) break;
    memset(wrkmem, 0, 16384 * sizeof(uint16_t));
    t = compress_chunk(&c, idx, ll, &op, t, wrkmem);
    if (c.oob) break;
    idx += ll;
    len -= ll;
  }
  free(wrkmem);
  if (!c.oob) {
    t += len;
    op = record_trailer(&c, n - t, t, op);
  }
  if (c.oob) return ORC_LZO_OUT_OF_BOUND;
  *out_len = (size_t)op;
  return ORC_OK;
}
