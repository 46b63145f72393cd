/* A plain C caller of include/mdeflate.h (no Python, no torch): what a cgo / OCaml stub would do.
 *   gcc -I include tests/c_consumer.c -L decompress_amd -lmdeflate -Wl,-rpath,$PWD/decompress_amd -o c_consumer
 * Exit code 0 = every check passed; 77 = no gfx950 device (the ABI calls fail loudly, nothing is emulated). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mdeflate.h"

#define CHECK(c)                                                  \
  do {                                                            \
    if (!(c)) {                                                   \
      fprintf(stderr, "c_consumer: %s failed (line %d)\n", #c, __LINE__); \
      return 1;                                                   \
    }                                                             \
  } while (0)

int main(void) {
  if (md_version() != MD_VERSION) return 1;
  if (strcmp(md_status_string(MD_INVALID_DISTANCE), "Invalid distance") != 0) return 1;
  if (md_device_count() == 0) {
    if (md_create(0, NULL) != NULL) return 1; /* no CPU fallback */
    printf("c_consumer: no gfx950 device: %s\n", md_last_error_string(NULL));
    return 77;
  }
  md_ctx *ctx = md_create(0, NULL);
  CHECK(ctx != NULL);
  /* Zl.Higher.compress then Zl.Higher.uncompress */
  enum { N = 200000 };
  unsigned char *plain = malloc(N), *z = malloc(2 * N), *back = malloc(N);
  for (int i = 0; i < N; i++) plain[i] = (unsigned char)("the quick brown fox "[i % 20] + (i / 4099) % 3);
  size_t zn = 0, bn = 0;
  CHECK(md_zl_higher_compress(ctx, 6, 1, 4096, plain, N, z, 2 * N, &zn) == MD_OK);
  CHECK(zn > 6 && zn < N && z[0] == 0x78 && z[1] == 0x9c);
  CHECK(md_zl_higher_uncompress(ctx, z, zn, back, N, &bn) == MD_OK && bn == N && memcmp(back, plain, N) == 0);
  /* De.Inf.Ns.inflate: (consumed, written) and the error variants */
  size_t used = 0;
  CHECK(md_de_inf_ns_inflate(ctx, z + 2, zn - 6, back, N, &used, &bn) == MD_OK && used == zn - 6 && bn == N);
  CHECK(md_de_inf_ns_inflate(ctx, z + 2, zn - 6, back, 100, &used, &bn) == MD_UNEXPECTED_END_OF_OUTPUT);
  CHECK(md_de_inf_ns_inflate(ctx, (const unsigned char *)"\x06", 1, back, 100, &used, &bn) == MD_INVALID_KIND_OF_BLOCK);
  /* the streaming protocol: `Await ... `Flush ... `End */
  unsigned char obuf[8192];
  md_inf_stream *s = md_inf_decoder(ctx, MD_FORMAT_ZLIB, obuf, sizeof obuf);
  CHECK(s != NULL);
  size_t fed = 0, got = 0;
  for (;;) {
    int sig = md_inf_decode(s);
    if (sig == MD_AWAIT) {
      size_t k = zn - fed < 1000 ? zn - fed : 1000;
      CHECK(md_inf_src(s, z, fed, k) == MD_OK); /* k == 0 is the end of input */
      fed += k;
    } else if (sig == MD_FLUSH || sig == MD_END) {
      size_t k = sizeof obuf - md_inf_dst_rem(s);
      CHECK(memcmp(obuf, plain + got, k) == 0);
      got += k;
      if (sig == MD_END) break;
      md_inf_flush(s);
    } else {
      CHECK(!"malformed");
    }
  }
  CHECK(got == N && md_inf_status(s) == MD_OK);
  /* the same decoder on the same stream again (De.Inf.reset), now in pieces: with 2000 bytes buffered it decodes up to
     the last block boundary, so output arrives while input is still being supplied */
  md_inf_reset(s);
  md_inf_chunk_bytes(s, 2000);
  fed = got = 0;
  size_t early = 0;
  for (;;) {
    int sig = md_inf_decode(s);
    if (sig == MD_AWAIT) {
      size_t k = zn - fed < 1000 ? zn - fed : 1000;
      CHECK(md_inf_src(s, z, fed, k) == MD_OK);
      fed += k;
    } else if (sig == MD_FLUSH || sig == MD_END) {
      size_t k = sizeof obuf - md_inf_dst_rem(s);
      CHECK(memcmp(obuf, plain + got, k) == 0);
      got += k;
      if (fed < zn) early += k;
      if (sig == MD_END) break;
      md_inf_flush(s);
    } else {
      CHECK(!"malformed");
    }
  }
  CHECK(got == N && md_inf_status(s) == MD_OK && md_inf_src_rem(s) == 0);
  CHECK(zn < 4000 || early > 0);
  md_inf_free(s);
  /* De.Lz77.compress and De.Def.encode on their own: "abcde" (test/test.ml:798-813) */
  uint32_t cmds[16], lits[286], dsts[30];
  size_t nc = 0;
  CHECK(md_de_lz77_compress(ctx, 4, 4096, MD_MATCHER_DE, (const unsigned char *)"abcde", 5, cmds, 16, &nc, lits, dsts) == MD_OK);
  CHECK(nc == 6 && cmds[0] == 'a' && cmds[4] == 'e' && cmds[5] == 256 && lits['c'] == 1);
  CHECK(md_de_def_encode(ctx, MD_BLOCK_FIXED, cmds, nc, z, 2 * N, &zn) == MD_OK);
  CHECK(md_de_higher_uncompress(ctx, z, zn, back, N, &bn) == MD_OK && bn == 5 && memcmp(back, "abcde", 5) == 0);
  /* misuse is a call-level error, not an abort */
  const md_deflate_params bad = {11, 4096, MD_DRIVER_ZL, 1, MD_MATCHER_DE, NULL};
  unsigned long long off = 0, len = 5, cap = 100, olen = 0;
  int st = 0;
  CHECK(md_deflate_batch_host(ctx, MD_FORMAT_ZLIB, &bad, 1, (const unsigned char *)"abcde", 5, (const uint64_t *)&off,
                              (const uint64_t *)&len, z, 100, (const uint64_t *)&off, (const uint64_t *)&cap,
                              (uint64_t *)&olen, &st, NULL) == MD_E_INVALID_ARGUMENT);
  md_destroy(ctx);
  free(plain);
  free(z);
  free(back);
  printf("c_consumer: ok\n");
  return 0;
}
