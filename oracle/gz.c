/* oracle/gz.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's GZip framing (mirage/decompress v1.6.0,
 * lib/gz.ml) around the DEFLATE body of de_inflate.c / de_deflate.c.
 *
 * Whole-buffer form of Gz.Inf (lib/gz.ml:248-633) and Gz.Def (lib/gz.ml:636-918):
 * the reference's decoder/encoder are streaming state machines; here the whole
 * input is present, `Await at the end of a `String source is "Unexpected end of
 * input", and the body follows De.Inf.Ns status semantics (as everywhere in this
 * engine).  Reference quirks kept on purpose:
 *   - FEXTRA's length is read big-endian (lib/gz.ml:455; RFC1952 says little-endian;
 *     the reference's own vector test/test.ml:1960-1989 depends on it),
 *   - MTIME is read/written big-endian (lib/gz.ml:479, :801),
 *   - the header CRC16 is the UPPER half of the CRC-32 of the 10 fixed bytes + name\0 +
 *     comment\0, FEXTRA excluded, stored big-endian (lib/gz.ml:422-439, :771-789),
 *   - CM is not validated, only the 0x1f8b magic is (lib/gz.ml:474-475).
 * Parity pinning: inflate by the reference's 5 vectors (tests/golden/gzip.json);
 * deflate has no expected-bytes vector upstream (round trips only): the frame is
 * pinned by the header/trailer rules above + the Zl-driver body (same caveat as
 * de_deflate.c).
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static uint32_t be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }
static uint32_t le32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* crc16 of lib/gz.ml:422-439: hdr = 10 fixed bytes, fname\0, fcomment\0 */
static uint32_t header_crc16(const uint8_t *fixed10, const uint8_t *name, size_t name_len, int has_name,
                             const uint8_t *comment, size_t comment_len, int has_comment) {
  uint32_t c = orc_crc32(0, fixed10, 10);
  const uint8_t z = 0;
  if (has_name) {
    c = orc_crc32(c, name, name_len);
    c = orc_crc32(c, &z, 1);
  }
  if (has_comment) {
    c = orc_crc32(c, comment, comment_len);
    c = orc_crc32(c, &z, 1);
  }
  return (c & 0xffff0000u) >> 16;
}

/* header, lib/gz.ml:465-491 (fextra :452-461, fpayload :442-450, zero_terminated :358-379,
 * fhcrc :422-439).  Returns a status; *body = offset of the DEFLATE body. */
int orc_gz_header(const uint8_t *src, size_t n, size_t *body, orc_gz_meta *m) {
  orc_gz_meta mm;
  memset(&mm, 0, sizeof mm);
  *body = 0;
  if (n < 10) return ORC_UNEXPECTED_END_OF_INPUT;
  if (be16(src) != 0x1f8b) return ORC_INVALID_GZIP_HEADER;
  mm.cm = src[2];
  mm.flg = src[3];
  mm.mtime = ((uint32_t)src[4] << 24) | ((uint32_t)src[5] << 16) | ((uint32_t)src[6] << 8) | src[7];
  mm.xfl = src[8];
  mm.os = src[9];
  size_t p = 10;
  if (mm.flg & 4) { /* fextra: uint16_be length, then the bytes */
    if (n - p < 2) return ORC_UNEXPECTED_END_OF_INPUT;
    size_t len = be16(src + p);
    p += 2;
    if (n - p < len) return ORC_UNEXPECTED_END_OF_INPUT;
    mm.has_extra = 1;
    mm.extra_off = p;
    mm.extra_len = len;
    p += len;
  }
  for (int which = 0; which < 2; which++) { /* fname (flg 8), fcomment (flg 16) */
    if (!(mm.flg & (which == 0 ? 8 : 16))) continue;
    size_t q = p;
    while (q < n && src[q] != 0) q++;
    if (q >= n) return ORC_UNEXPECTED_END_OF_INPUT;
    if (which == 0) {
      mm.has_name = 1;
      mm.name_off = p;
      mm.name_len = q - p;
    } else {
      mm.has_comment = 1;
      mm.comment_off = p;
      mm.comment_len = q - p;
    }
    p = q + 1;
  }
  if (mm.flg & 2) { /* fhcrc */
    if (n - p < 2) return ORC_UNEXPECTED_END_OF_INPUT;
    uint32_t want = header_crc16(src, src + mm.name_off, mm.name_len, mm.has_name, src + mm.comment_off,
                                 mm.comment_len, mm.has_comment);
    if (want != be16(src + p)) return ORC_INVALID_GZIP_HEADER_CHECKSUM;
    p += 2;
  }
  *body = p;
  if (m) *m = mm;
  return ORC_OK;
}

/* Gz.Inf over a whole buffer: header, De.Inf body, checksum (lib/gz.ml:344-356) */
int orc_gz_inflate(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap, size_t *consumed,
                   size_t *written, orc_gz_meta *m) {
  *consumed = 0;
  *written = 0;
  size_t body;
  int rc = orc_gz_header(src, n, &body, m);
  if (rc) return rc;
  size_t i, o;
  rc = orc_de_inf_ns_inflate(src + body, n - body, dst, dst_cap, &i, &o);
  *written = o;
  if (rc) return rc;
  if (n - body - i < 8) return ORC_UNEXPECTED_END_OF_INPUT;
  const uint8_t *t = src + body + i;
  /* crc first, then isize (lib/gz.ml:351-354) */
  if (le32(t) != orc_crc32(0, dst, o)) return ORC_INVALID_CHECKSUM;
  if (le32(t + 4) != (uint32_t)o) return ORC_INVALID_SIZE;
  *consumed = body + i + 8;
  return ORC_OK;
}

/* Gz.Def over a whole buffer (lib/gz.ml:794-845 driver = Zl's with dynamic = true;
 * header :796-812, xfl :888-890, trailer :715-722).  name/comment may be NULL. */
uint8_t *orc_gz_deflate(const uint8_t *src, size_t n, int level, int queue_len, uint32_t mtime, int os,
                        int hcrc, int ascii, const char *name, const char *comment, size_t *out_len) {
  size_t blen;
  uint32_t adler;
  uint8_t *body = orc_deflate_raw(src, n, level, queue_len, ORC_DRV_ZL, 1, &blen, &adler);
  if (!body) return NULL;
  size_t nl = name ? strlen(name) : 0, cl = comment ? strlen(comment) : 0;
  uint8_t *out = (uint8_t *)malloc(10 + nl + 1 + cl + 1 + 2 + blen + 8);
  size_t p = 0;
  int flg = (ascii ? 1 : 0) | (hcrc ? 2 : 0) | (name ? 8 : 0) | (comment ? 16 : 0);
  out[p++] = 0x1f;
  out[p++] = 0x8b;
  out[p++] = 8;
  out[p++] = (uint8_t)flg;
  out[p++] = (uint8_t)(mtime >> 24);
  out[p++] = (uint8_t)(mtime >> 16);
  out[p++] = (uint8_t)(mtime >> 8);
  out[p++] = (uint8_t)mtime;
  out[p++] = (level >= 0 && level <= 8) ? 0 : 2;
  out[p++] = (uint8_t)os;
  if (name) {
    memcpy(out + p, name, nl + 1);
    p += nl + 1;
  }
  if (comment) {
    memcpy(out + p, comment, cl + 1);
    p += cl + 1;
  }
  if (hcrc) {
    uint32_t c16 = header_crc16(out, (const uint8_t *)name, nl, name != NULL, (const uint8_t *)comment, cl,
                                comment != NULL);
    out[p++] = (uint8_t)(c16 >> 8);
    out[p++] = (uint8_t)c16;
  }
  memcpy(out + p, body, blen);
  p += blen;
  free(body);
  uint32_t crc = orc_crc32(0, src, n), isize = (uint32_t)n;
  for (int k = 0; k < 4; k++) out[p++] = (uint8_t)(crc >> (8 * k));
  for (int k = 0; k < 4; k++) out[p++] = (uint8_t)(isize >> (8 * k));
  *out_len = p;
  return out;
}
