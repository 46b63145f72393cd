/* oracle/oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the reference's RFC1951 hot path
 * (mirage/decompress v1.6.0, lib/de.ml + lib/zl.ml).  Each function cites the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call this library; the product path
 * (decompress_amd/, libmdeflate.so) never does.
 *
 * Parity pinning: inflate is pinned by the reference's own known-answer vectors
 * (tests/golden/inflate_ns.json, inflate_stream.json, transcribed from
 * test/test_ns.ml and test/test.ml) and by libz on valid streams.  Deflate BYTES are
 * pinned only by the reference's 4 encoder KATs + 2 tree KATs
 * (tests/golden/deflate_kat.json): beyond those, deflate byte parity is UNPINNED
 * (no OCaml toolchain in the build image, the reference cannot be run).  The LZ77
 * DECISIONS of the matcher shared by De.Lz77 and lib/lz.ml are pinned through libz:
 * with lib/lz.ml's hash the token stream equals libz's own (tests/test_oracle_lz.py).
 * GZip framing (gz.c) is pinned by the reference's 5 gzip vectors (tests/golden/gzip.json).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: 1:1 with De.Inf.Ns.error (lib/de.ml:1548-1566) and
 * Zl.Inf.Ns.error (lib/zl.ml:383).  Same numbering as include/mdeflate.h. */
enum {
  ORC_OK = 0,
  ORC_UNEXPECTED_END_OF_INPUT = 1,
  ORC_UNEXPECTED_END_OF_OUTPUT = 2,
  ORC_INVALID_KIND_OF_BLOCK = 3,
  ORC_INVALID_DICTIONARY = 4,
  ORC_INVALID_COMPLEMENT_OF_LENGTH = 5,
  ORC_INVALID_DISTANCE = 6,
  ORC_INVALID_DISTANCE_CODE = 7,
  ORC_INVALID_HEADER = 8,
  ORC_INVALID_CHECKSUM = 9,
  /* Gz.Inf's `Malformed strings (lib/gz.ml:284-296) */
  ORC_INVALID_GZIP_HEADER = 10,          /* "Invalid GZip header" */
  ORC_INVALID_GZIP_HEADER_CHECKSUM = 11, /* "Invalid GZip header checksum" */
  ORC_INVALID_SIZE = 12,                 /* "Invalid input size (expect:.., inflated:..)" */
  ORC_QUEUE_FULL = 13,                   /* deflate: exception De.Queue.Full (lib/de.ml:2211) */
  /* Lzo.error (lib/lzo.ml:4-12) */
  ORC_LZO_INVALID_INPUT = 14, /* `Malformed "Invalid input" (count, lib/lzo.ml:236) */
  ORC_LZO_NO_DICTIONARY = 15, /* `Malformed "No dictionary at offset 0 available" (lib/lzo.ml:376) */
  ORC_LZO_OUT_OF_BOUND = 16   /* `Invalid_argument "Input is malformed or output is not large enough" (lib/lzo.ml:401-402) */
};

/* Checkseum.Adler32 (external dep, RFC1950 §8.2); call sites lib/de.ml:453-455 */
uint32_t orc_adler32(uint32_t adler, const uint8_t *buf, size_t len);
/* Checkseum.Crc32 (RFC1952 §8); call sites lib/gz.ml:428 */
uint32_t orc_crc32(uint32_t crc, const uint8_t *buf, size_t len);

/* De.Inf.Ns.inflate  (lib/de.ml:1807-1822) */
int orc_de_inf_ns_inflate(const uint8_t *src, size_t src_len, uint8_t *dst,
                          size_t dst_cap, size_t *consumed, size_t *written);
/* Zl.Inf.Ns.inflate  (lib/zl.ml:400-417) */
int orc_zl_inf_ns_inflate(const uint8_t *src, size_t src_len, uint8_t *dst,
                          size_t dst_cap, size_t *consumed, size_t *written);

/* n zlib streams of one blob, back to back (bench.py's multi-threaded cpu_baseline leg) */
size_t orc_zl_inf_ns_inflate_batch(const uint8_t *blob, const uint64_t *off, const uint64_t *len, size_t n,
                                   uint8_t *scratch, size_t cap, uint64_t *total_out);

/* De.Inf.huffman (lib/de.ml:523-638).  kind: 0 CODES, 1 LENS, 2 DISTS.
 * tbl must hold 852 (LENS), 592 (DISTS) or 128 (CODES) entries.
 * Returns 0, or -1 for Invalid_huffman.  root/maxl outputs as in the OCaml triple. */
int orc_inf_huffman(int kind, const uint8_t *lens, int codes, uint32_t *tbl,
                    int *size, int *root, int *maxl);

const char *orc_status_string(int status);

/* ---- GZip framing (oracle/gz.c) ---- */
typedef struct {
  uint32_t cm, flg, mtime, xfl, os;
  int has_extra, has_name, has_comment;
  size_t extra_off, extra_len, name_off, name_len, comment_off, comment_len; /* into src */
} orc_gz_meta;
/* Gz.Inf header (lib/gz.ml:465-491) */
int orc_gz_header(const uint8_t *src, size_t n, size_t *body, orc_gz_meta *m);
/* Gz.Inf / Gz.Higher.uncompress over a whole buffer (lib/gz.ml:248-633, :959-982) */
int orc_gz_inflate(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap, size_t *consumed,
                   size_t *written, orc_gz_meta *m);

/* ---- deflate (oracle/de_deflate.c) ---- */
enum { ORC_DRV_ZL = 0, ORC_DRV_HIGHER = 1, ORC_DRV_CLI = 2 };
/* Raw DEFLATE body of De.Lz77 (lib/de.ml:4013-4515) + De.Def (lib/de.ml:2354-3038)
 * under one of the reference's three drivers.  malloc'ed result; NULL when
 * the reference would raise De.Queue.Full (the CLI driver's extra end-of-block push into a
 * full queue, bin/decompress.ml:67). */
uint8_t *orc_deflate_raw(const uint8_t *src, size_t n, int level, int queue_len, int driver,
                         int dynamic, size_t *out_len, uint32_t *adler);
enum { ORC_MATCHER_DE = 0, ORC_MATCHER_LZ = 1 };
/* the same with lib/lz.ml's match finder (Lz.state / Lz.compress, lib/lz.ml:136-573) */
uint8_t *orc_deflate_raw_m(const uint8_t *src, size_t n, int level, int queue_len, int driver,
                           int dynamic, int matcher, size_t *out_len, uint32_t *adler);
/* Zl.Def.encode / Zl.Higher.compress (lib/zl.ml:509-555, 634-648) */
uint8_t *orc_zl_deflate(const uint8_t *src, size_t n, int level, int queue_len, int dynamic,
                        size_t *out_len);
/* Gz.Def / Gz.Higher.compress over a whole buffer (lib/gz.ml:636-918, :927-950) */
uint8_t *orc_gz_deflate(const uint8_t *src, size_t n, int level, int queue_len, uint32_t mtime, int os,
                        int hcrc, int ascii, const char *name, const char *comment, size_t *out_len);
void orc_free(void *p);
/* the drivers above hand the input to De.Lz77 `piece` bytes per `Await (0, the default: all at once); not thread-safe */
void orc_set_src_piece(size_t piece);
void orc_set_src_first_piece(size_t first); /* the first piece of another size (0: like the others) */
/* De.Def.Ns.deflate / compress_bound (lib/de.ml:3040-4010) and Zl.Def.Ns.deflate (lib/zl.ml:596-629), oracle/de_def_ns.c:
 * ORC_OK with *out_len = the `Ok n` (0 for the stub levels 5..12), ORC_UNEXPECTED_END_OF_OUTPUT, -1 = `Invalid_compression_level */
int orc_de_def_ns_deflate(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap, int level, size_t *out_len);
size_t orc_de_def_ns_compress_bound(size_t len);
int orc_zl_def_ns_deflate(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap, int level, size_t *out_len);

/* ---- LZO1X (oracle/lzo.c) ---- */
/* Lzo.uncompress input output (lib/lzo.ml:395-403); a failing stream leaves *written = 0 */
int orc_lzo_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *written);
/* Lzo.compress in_data out_data wrkmem (lib/lzo.ml:642-660) */
int orc_lzo_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len);
/* De.T.make (lib/de.ml:2013-2068) on a histogram (mutated in place) */
int orc_tree_make(int length, int max_length, int *freqs, int nfreqs, int *lengths, int *codes);
/* test/test.ml `encode` / `encode_dynamic`: a command list in one last block */
uint8_t *orc_encode_cmds(const int *cmds, int ncmds, int kind, size_t *out_len);
/* De.Lz77.compress on an input that fits one queue fill (test/test.ml:798-813) */
int orc_lz77_cmds(const uint8_t *src, size_t n, int level, int queue_len, int *out, int max);
/* De.Def.encode driven by a list of operations (test/test_ns.ml:388-615); see de_deflate.c */
uint8_t *orc_def_script(const int *ops, int nops, int queue_len, int *rcs, int max_rcs, int *nrcs, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
