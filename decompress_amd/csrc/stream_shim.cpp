// stream_shim.cpp — the reference's resumable decoder / encoder state machines (`Await / `Flush / `End) above the
// batch C ABI: SURVEY.md 8(b) export item (3).
//
// De.Inf.decode (lib/de.ml:1427-1474, signature lib/de.mli:82-144) and Zl.Def.encode / De.Def.encode's drivers
// (lib/zl.ml:509-555, lib/de.mli:300-412) hand a few KiB to the codec per call; a kernel launch per 64 KiB step
// cannot pay for itself, so the shim keeps the reference's calling protocol on the HOST — it collects the chunks
// the caller supplies through src and runs the HIP path on them in large pieces: a stream that ends before a piece
// (md_inf_chunk_bytes, 8 MiB) is full is ONE batch-of-one launch when the caller signals the end of input (src with
// length 0, as in the reference); a longer stream is decoded piece by piece up to the last block
// boundary inside each piece (md_de_inf_continue_host: starting bit, 32 KiB window and checksum state go in, the
// boundary comes back), so that output is handed out through `Flush steps while input is still arriving and only the
// undecoded tail and the window are kept.  There is no CPU codec here: without a gfx950 device the launch fails and
// the stream reports the call-level error.
//
// Divergence (documented): the kernels have De.Inf.Ns's whole-buffer end-of-input rule; the
// streaming rule of lib/de.ml:941-944 (a final end-of-block code shorter than the longest code is accepted at the
// end of the input) gives the same result on every stream a compressor emits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "mdeflate.h"

struct md_inf_stream {
  md_ctx *ctx;
  int format;
  uint8_t *o;
  size_t o_len, o_pos;        // caller's output buffer (De.Inf.decoder ~o) and how much of it is filled
  std::vector<uint8_t> in;    // everything supplied so far
  std::vector<uint8_t> out;   // the decoded stream, once the launch has run
  size_t served;              // bytes of `out` already handed to the caller
  bool eoi, ran;
  int status;                 // MD_* status of the launch
  size_t consumed;
  uint32_t checksum;
  std::string message;        // the reference's `Malformed string, with its numbers
  // decoding in pieces (DEFLATE / ZLIB): `in` is then the undecoded tail, `out` what the last piece produced
  size_t chunk, need;         // input buffered before a piece is decoded; before the NEXT piece (grows when a piece holds no block end)
  bool piecewise, hdr_done, body_done, finished;
  unsigned in_bit;            // the next block starts this many bits into in[0]
  std::vector<uint8_t> hist;  // the window: the last <= 32 KiB of output
  uint32_t adler;             // checksum state at the last block boundary
  uint32_t crc;               // GZip: CRC-32 of the output handed out so far
  uint64_t total_out;         // ... and its length
};
static void inf_clear(md_inf_stream *s) {
  s->in.clear();
  s->out.clear();
  s->hist.clear();
  s->message.clear();
  s->o_pos = s->served = s->consumed = 0;
  s->eoi = s->ran = s->piecewise = s->hdr_done = s->body_done = s->finished = false;
  s->status = MD_OK;
  s->checksum = 0;
  s->need = s->chunk;
  s->in_bit = 0;
  s->adler = 1;
  s->crc = 0;
  s->total_out = 0;
}

extern "C" {

md_inf_stream *md_inf_decoder(md_ctx *ctx, int format, uint8_t *o, size_t o_len) {
  if (!ctx || !o || o_len == 0) return nullptr;
  if (format != MD_FORMAT_DEFLATE && format != MD_FORMAT_ZLIB && format != MD_FORMAT_GZIP) return nullptr;
  md_inf_stream *s = new md_inf_stream();
  s->ctx = ctx;
  s->format = format;
  s->o = o;
  s->o_len = o_len;
  s->chunk = (size_t)8 << 20;  // (pieces this long are decoded by the whole chip: capi.cpp continue_parallel)
  inf_clear(s);
  return s;
}

void md_inf_free(md_inf_stream *s) { delete s; }

// De.Inf.reset (lib/de.ml:1512-1532; Zl.Inf.reset, Gz.Inf.reset): the same decoder, output buffer and format, a new stream
void md_inf_reset(md_inf_stream *s) {
  if (s) inf_clear(s);
}
void md_inf_chunk_bytes(md_inf_stream *s, size_t bytes) {
  if (!s || s->piecewise) return;
  s->chunk = s->need = bytes ? bytes : 1;
}

int md_inf_src(md_inf_stream *s, const uint8_t *buf, size_t off, size_t len) {
  if (!s || (!buf && len) || s->eoi) return MD_E_INVALID_ARGUMENT;
  if (len == 0) s->eoi = true;  // De.Inf.src d buf 0 0: the end of the input
  else s->in.insert(s->in.end(), buf + off, buf + off + len);
  return MD_OK;
}

void md_inf_flush(md_inf_stream *s) {
  if (s) s->o_pos = 0;
}
size_t md_inf_dst_rem(const md_inf_stream *s) { return s ? s->o_len - s->o_pos : 0; }
size_t md_inf_src_rem(const md_inf_stream *s) { return s && (s->ran || s->finished) ? s->in.size() - s->consumed : 0; }
int md_inf_status(const md_inf_stream *s) { return s ? s->status : MD_E_INVALID_ARGUMENT; }
const char *md_inf_message(const md_inf_stream *s) {
  if (!s) return "Invalid argument";
  return s->message.empty() ? md_status_string(s->status) : s->message.c_str();
}
uint32_t md_inf_checksum(const md_inf_stream *s) { return s ? s->checksum : 0; }

// where the DEFLATE body of a GZip member begins (Gz.Inf's header walk, lib/gz.ml:465-531: FEXTRA's length is read
// big-endian there); 0 when the header is cut short
static size_t gz_body_offset(const std::vector<uint8_t> &in) {
  if (in.size() < 10) return 0;
  const uint32_t flg = in[3];
  size_t p = 10;
  if (flg & 4) {
    if (in.size() - p < 2) return 0;
    const size_t xl = ((size_t)in[p] << 8) | in[p + 1];
    p += 2;
    if (in.size() - p < xl) return 0;
    p += xl;
  }
  for (int which = 0; which < 2; which++) {
    if (!(flg & (which == 0 ? 8u : 16u))) continue;
    for (;;) {
      if (p >= in.size()) return 0;
      if (in[p++] == 0) break;
    }
  }
  if (flg & 2) {
    if (in.size() - p < 2) return 0;
    p += 2;
  }
  return p;
}

// The reference's `Malformed string for a frame whose trailer disagrees: "Invalid checksum (expect:%04lx, has:%04lx)"
// (lib/zl.ml:179-181, lib/gz.ml:287-289: expect = the trailer's value, has = the checksum of what was inflated) and
// "Invalid input size (expect:%ld, inflated:%ld)" (lib/gz.ml:291-293, both as signed 32-bit).  The kernels report the
// status only, so the trailer is looked up here: one more launch of the raw body finds where it ends.
static void inf_detail(md_inf_stream *s) {
  s->message = md_status_string(s->status);
  if (s->status != MD_INVALID_CHECKSUM && s->status != MD_INVALID_SIZE) return;
  const size_t body = s->format == MD_FORMAT_ZLIB ? 2 : gz_body_offset(s->in);
  if (body == 0 || body > s->in.size()) return;
  uint64_t in_off = body, in_len = s->in.size() - body, out_off = 0, out_cap = s->out.size(), out_len = 0, used = 0;
  int32_t st = 0;
  std::vector<uint8_t> scratch(s->out.size() + 16);
  if (md_inflate_batch_host(s->ctx, MD_FORMAT_DEFLATE, 1, s->in.data(), s->in.size(), &in_off, &in_len, scratch.data(),
                            scratch.size(), &out_off, &out_cap, &out_len, &used, &st, nullptr) != MD_OK || st != MD_OK)
    return;
  const size_t t = body + (size_t)used;
  char buf[96];
  if (s->format == MD_FORMAT_ZLIB) {
    if (s->in.size() - t < 4) return;
    const uint32_t expect = ((uint32_t)s->in[t] << 24) | ((uint32_t)s->in[t + 1] << 16) | ((uint32_t)s->in[t + 2] << 8) | s->in[t + 3];
    snprintf(buf, sizeof buf, "Invalid checksum (expect:%04lx, has:%04lx)", (unsigned long)expect, (unsigned long)s->checksum);
  } else {
    if (s->in.size() - t < 8) return;
    const uint32_t crc = (uint32_t)s->in[t] | ((uint32_t)s->in[t + 1] << 8) | ((uint32_t)s->in[t + 2] << 16) | ((uint32_t)s->in[t + 3] << 24);
    const uint32_t isize = (uint32_t)s->in[t + 4] | ((uint32_t)s->in[t + 5] << 8) | ((uint32_t)s->in[t + 6] << 16) | ((uint32_t)s->in[t + 7] << 24);
    if (s->status == MD_INVALID_CHECKSUM)
      snprintf(buf, sizeof buf, "Invalid checksum (expect:%04lx, has:%04lx)", (unsigned long)crc, (unsigned long)s->checksum);
    else
      snprintf(buf, sizeof buf, "Invalid input size (expect:%ld, inflated:%ld)", (long)(int32_t)isize, (long)(int32_t)(uint32_t)out_len);
  }
  s->message = buf;
}

static void inf_run(md_inf_stream *s) {
  // The output size is not known.  A DEFLATE stream expands at most 1032 times: a small input gets room for that at
  // once (one launch whatever its ratio); a GZip member says its size (mod 2^32) in its last four bytes; otherwise
  // start from 4x the input and grow fourfold while the codec runs out of room (5 launches at the worst ratio).
  const uint64_t n_in = s->in.size();
  uint64_t cap = n_in * 4 + 65536;
  if (n_in * 1032 <= (64u << 20)) cap = n_in * 1032 + 65536;
  else if (s->format == MD_FORMAT_GZIP && n_in >= 18) {
    const uint8_t *t = s->in.data() + n_in - 4;
    const uint64_t isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
    if (isize >= cap && isize <= n_in * 1032) cap = isize + 65536;
  }
  for (;;) {
    if (cap > MD_MAX_STREAM) cap = MD_MAX_STREAM;
    s->out.resize((size_t)cap);
    uint64_t in_off = 0, in_len = s->in.size(), out_off = 0, out_cap = cap, out_len = 0, used = 0;
    int32_t st = 0;
    uint32_t sum = 0;
    static const uint8_t none = 0;
    int rc = md_inflate_batch_host(s->ctx, s->format, 1, s->in.empty() ? &none : s->in.data(), s->in.size(), &in_off, &in_len,
                                   s->out.data(), (size_t)cap, &out_off, &out_cap, &out_len, &used, &st, &sum);
    if (rc != MD_OK) {
      s->status = rc;
      s->out.clear();
      return;
    }
    if (st == MD_UNEXPECTED_END_OF_OUTPUT && cap < MD_MAX_STREAM) {
      cap *= 4;
      continue;
    }
    s->status = st;
    s->consumed = (size_t)used;
    s->checksum = sum;
    s->out.resize((size_t)out_len);
    if (st != MD_OK) inf_detail(s);
    return;
  }
}

// ---- CRC-32 over pieces: crc(A || B) = crc(A) * x^(8|B|) mod P xor crc(B) in GF(2)[x] / P, reflected (bit 31 = x^0);
// the pieces' CRCs come from the device (crc32_kernel), the few header bytes are done here
static uint32_t crc_bytes(uint32_t c, const uint8_t *p, size_t n) {  // (running value, not complemented)
  for (size_t i = 0; i < n; i++) {
    c ^= p[i];
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1)));
  }
  return c;
}
static uint32_t gf_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int k = 0; k < 32; k++) {
    p ^= b & (0u - ((a >> 31) & 1));
    a <<= 1;
    b = (b >> 1) ^ (0xedb88320u & (0u - (b & 1)));
  }
  return p;
}
static uint32_t crc_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
  uint32_t sq = 0x00800000u, r = 0x80000000u;  // x^8, x^0
  for (uint64_t n = len_b; n; n >>= 1) {
    if (n & 1) r = gf_mul(r, sq);
    sq = gf_mul(sq, sq);
  }
  return gf_mul(crc_a, r) ^ crc_b;
}
// Gz.Inf's header walk with its checks (lib/gz.ml:463-491, as gz_header_kernel does it for the batch path): the offset
// of the body, or 0 with *st = MD_OK when the header is not all there yet, or 0 with the status of a bad header
static size_t gz_header_check(const std::vector<uint8_t> &in, int *st) {
  *st = MD_OK;
  // (the reference looks at the ID bytes only once the ten fixed bytes are there, lib/gz.ml:463-491: a cut header of
  // fewer bytes is "unexpected end of input" whatever its first bytes are)
  if (in.size() >= 10 && (in[0] != 0x1f || in[1] != 0x8b)) {
    *st = MD_INVALID_GZIP_HEADER;
    return 0;
  }
  const size_t body = gz_body_offset(in);
  if (body == 0) return 0;
  const uint32_t flg = in[3];
  if (flg & 2) {  // FHCRC: the upper half of the CRC-32 of the fixed bytes + name + comment (FEXTRA excluded), big-endian
    uint32_t c = crc_bytes(0xffffffffu, in.data(), 10);
    size_t p = 10;
    if (flg & 4) p += 2 + (((size_t)in[10] << 8) | in[11]);
    c = crc_bytes(c, in.data() + p, body - 2 - p);
    const uint32_t want = ((c ^ 0xffffffffu) & 0xffff0000u) >> 16, have = ((uint32_t)in[body - 2] << 8) | in[body - 1];
    if (want != have) {
      *st = MD_INVALID_GZIP_HEADER_CHECKSUM;
      return 0;
    }
  }
  return body;
}

// One piece of a stream that is decoded as it arrives: everything up to the last block boundary inside the buffered
// input goes to `out`, the rest of the input stays; at the end of the input whatever is left is decoded for good.
static void inf_piece(md_inf_stream *s) {
  const bool final = s->eoi;
  s->out.clear();
  s->served = 0;
  auto fail_with = [&](int st) {
    s->status = st;
    s->message = md_status_string(st);
    s->finished = true;
  };
  if (s->format == MD_FORMAT_GZIP && !s->hdr_done) {
    int hst = MD_OK;
    const size_t body = gz_header_check(s->in, &hst);
    if (hst != MD_OK) return fail_with(hst);
    if (body == 0) {
      if (final) fail_with(MD_UNEXPECTED_END_OF_INPUT);
      else s->need = s->in.size() + 1;
      return;
    }
    s->in.erase(s->in.begin(), s->in.begin() + body);
    s->hdr_done = true;
  }
  if (s->format == MD_FORMAT_ZLIB && !s->hdr_done) {  // Zl.Inf's header, lib/zl.ml:142-165 (as the kernel checks it)
    if (s->in.size() < 2) {
      if (final) fail_with(MD_UNEXPECTED_END_OF_INPUT);
      else s->need = 2;
      return;
    }
    const unsigned cmf = s->in[0], flg = s->in[1];
    if (((cmf << 8) + flg) % 31 != 0 || (cmf & 0xf) != 8) return fail_with(MD_INVALID_HEADER);
    s->in.erase(s->in.begin(), s->in.begin() + 2);
    s->hdr_done = true;
  }
  if (!s->body_done) {
    const size_t hl = s->hist.size();
    uint64_t cap = (uint64_t)hl + s->in.size() * 4 + 65536;
    std::vector<uint8_t> buf;
    size_t dst_len = 0;
    int st = 0;
    md_inf_resume rs;
    static const uint8_t none = 0;
    for (;;) {
      if (cap > MD_MAX_STREAM) cap = MD_MAX_STREAM;
      buf.resize((size_t)cap);
      if (hl) memcpy(buf.data(), s->hist.data(), hl);
      const int rc = md_de_inf_continue_host(s->ctx, s->in.empty() ? &none : s->in.data(), s->in.size(), s->in_bit, buf.data(), hl,
                                             (size_t)cap, s->adler, s->format == MD_FORMAT_GZIP ? MD_CONT_CRC32 : 0u, &dst_len, &st, &rs);
      if (rc != MD_OK) return fail_with(rc);
      if (st == MD_UNEXPECTED_END_OF_OUTPUT && cap < MD_MAX_STREAM) {
        cap *= 4;
        continue;
      }
      break;
    }
    if (st == MD_UNEXPECTED_END_OF_INPUT && !final) {
      // the piece ends inside a block: hand out what lies before that block, keep the rest of the input
      const size_t upto = (size_t)rs.out;
      const bool progress = rs.bits > s->in_bit;
      s->out.assign(buf.begin() + hl, buf.begin() + upto);
      s->crc = s->total_out ? crc_concat(s->crc, rs.crc_out, upto - hl) : rs.crc_out;
      s->total_out += upto - hl;
      const size_t keep = upto < 32768 ? upto : 32768;
      s->hist.assign(buf.begin() + (upto - keep), buf.begin() + upto);
      s->adler = rs.adler;
      s->in.erase(s->in.begin(), s->in.begin() + (size_t)(rs.bits >> 3));
      s->in_bit = (unsigned)(rs.bits & 7);
      // (after progress: a new attempt only once input beyond the undecoded tail has arrived - that tail holds no
      // complete block, decoding it again alone could not find one)
      s->need = progress ? (s->in.size() + 1 > s->chunk ? s->in.size() + 1 : s->chunk)
                         : (s->in.size() * 2 > s->chunk ? s->in.size() * 2 : s->chunk);
      return;
    }
    s->out.assign(buf.begin() + hl, buf.begin() + dst_len);  // everything decoded, also in front of an error
    s->crc = s->total_out ? crc_concat(s->crc, rs.crc_end, dst_len - hl) : rs.crc_end;
    s->total_out += dst_len - hl;
    s->checksum = s->format == MD_FORMAT_GZIP ? s->crc : rs.checksum;
    if (st != MD_OK) return fail_with(st);
    s->body_done = true;
    s->in.erase(s->in.begin(), s->in.begin() + (size_t)rs.consumed);
    s->in_bit = 0;
    if (s->format == MD_FORMAT_DEFLATE) {
      s->status = MD_OK;
      s->finished = true;
      return;
    }
  }
  if (s->format == MD_FORMAT_GZIP) {  // Gz.Inf's trailer (lib/gz.ml:344-356): CRC-32 first, then ISIZE, little-endian
    if (s->in.size() < 8) {
      if (final) fail_with(MD_UNEXPECTED_END_OF_INPUT);
      else s->need = 8;
      return;
    }
    const uint8_t *t = s->in.data();
    const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    const uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    char msg[96];
    s->status = MD_OK;
    if (crc != s->crc) {
      snprintf(msg, sizeof msg, "Invalid checksum (expect:%04lx, has:%04lx)", (unsigned long)crc, (unsigned long)s->crc);
      s->status = MD_INVALID_CHECKSUM;
      s->message = msg;
    } else if (isize != (uint32_t)s->total_out) {
      snprintf(msg, sizeof msg, "Invalid input size (expect:%ld, inflated:%ld)", (long)(int32_t)isize, (long)(int32_t)(uint32_t)s->total_out);
      s->status = MD_INVALID_SIZE;
      s->message = msg;
    }
    s->in.erase(s->in.begin(), s->in.begin() + 8);
    s->finished = true;
    return;
  }
  // Zl.Inf's trailer: the Adler-32 of the output, big-endian (lib/zl.ml:171-186)
  if (s->in.size() < 4) {
    if (final) fail_with(MD_UNEXPECTED_END_OF_INPUT);
    else s->need = 4;
    return;
  }
  const uint32_t expect = ((uint32_t)s->in[0] << 24) | ((uint32_t)s->in[1] << 16) | ((uint32_t)s->in[2] << 8) | s->in[3];
  s->in.erase(s->in.begin(), s->in.begin() + 4);
  if (expect != s->checksum) {
    char msg[96];
    snprintf(msg, sizeof msg, "Invalid checksum (expect:%04lx, has:%04lx)", (unsigned long)expect, (unsigned long)s->checksum);
    s->status = MD_INVALID_CHECKSUM;
    s->message = msg;
  } else {
    s->status = MD_OK;
  }
  s->finished = true;
}

int md_inf_decode(md_inf_stream *s) {
  if (!s) return MD_MALFORMED;
  for (;;) {
    // what has been decoded goes to the caller's buffer first
    const size_t left = s->out.size() - s->served, room = s->o_len - s->o_pos;
    const size_t n = left < room ? left : room;
    if (n) memcpy(s->o + s->o_pos, s->out.data() + s->served, n);
    s->o_pos += n;
    s->served += n;
    if (s->served < s->out.size() || (n && s->o_pos == s->o_len && s->finished && s->status != MD_OK)) return MD_FLUSH;  // the buffer is full
    if (s->finished) return s->status == MD_OK ? MD_END : MD_MALFORMED;
    if (!s->piecewise && s->eoi) {  // the whole stream in one launch
      inf_run(s);
      s->ran = s->finished = true;
      continue;
    }
    if (!s->eoi && s->in.size() < s->need) return MD_AWAIT;
    s->piecewise = true;
    inf_piece(s);
  }
}

// ---- the encoder side: Zl.Def.encoder / Gz.Def.encoder / De.Higher's loop with `Manual src and dst ----
}  // extern "C"

// The encoder takes its stream in pieces (capi.cpp, md_i_piece_*): the device goes on from the state the piece before
// left, so neither side keeps more of the stream than the 64 KiB the matcher can reach back plus the piece in flight.
struct md_piece;
extern "C" {
md_piece *md_i_piece_open(md_ctx *ctx, int queue_len);
void md_i_piece_close(md_ctx *ctx, md_piece *p);
int md_i_piece_run(md_ctx *ctx, md_piece *p, int format, const md_deflate_params *params, const uint8_t *text, size_t text_len,
                   size_t seen, uint64_t w0, uint64_t rebase, int first, int last, uint32_t sum, uint32_t isize, size_t out_cap,
                   size_t *out_len, int *status);
int md_i_test_flags(const md_ctx *ctx);
int md_i_piece_out(md_ctx *ctx, const md_piece *p, size_t off, uint8_t *host, size_t len);
size_t md_i_piece_bytes(const md_ctx *ctx);
struct md_pieces_io {
  const uint64_t *text_off, *text_len, *abs_len, *out_off, *out_cap, *w0, *rebase;
  const uint32_t *flags, *sum, *isize;
  uint64_t *out_len;
  int32_t *status;
};
int md_i_pieces_run(md_ctx *ctx, int format, const md_deflate_params *params, size_t n, const uint8_t *d_text, uint8_t *d_out,
                    void *d_state, void *d_queue, void **d_desc, size_t *d_desc_bytes, const md_pieces_io *io, uint32_t match_skip);
hipStream_t md_i_stream(md_ctx *ctx);
int md_i_device(md_ctx *ctx);
uint32_t md_piece_state_bytes();
int md_launch_piece_gather(uint32_t n, const uint8_t *old_blob, const uint8_t *fresh, uint8_t *new_blob, const uint64_t *d, hipStream_t stream);
}
namespace {
constexpr size_t kKeepBytes = 65536;             // text behind the end of a piece that the next launch sees again
constexpr size_t kSrcMax = (size_t)1 << 30;      // most that one md_def_src hands over
constexpr int kPieceAwait = 1000;                // MD_PIECE_AWAIT, deflate_common.hpp

uint32_t adler32_update(uint32_t adler, const uint8_t *p, size_t n) {  // lib/de.ml:4217-4218 keeps it per fill; the sum is the same
  uint32_t a = adler & 0xffff, b = adler >> 16;
  while (n) {
    size_t k = n < 5552 ? n : 5552;
    n -= k;
    while (k--) {
      a += *p++;
      b += a;
    }
    a %= 65521u;
    b %= 65521u;
  }
  return (b << 16) | a;
}
struct Crc32Tables {
  uint32_t t[8][256];
  Crc32Tables() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1)));
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int j = 1; j < 8; j++) t[j][i] = (t[j - 1][i] >> 8) ^ t[0][t[j - 1][i] & 0xff];
  }
};
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {  // eight bytes a step
  static const Crc32Tables T;
  uint32_t c = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^ T.t[3][hi & 0xff] ^
        T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return ~c;
}
}  // namespace

struct md_def_stream {
  md_ctx *ctx;
  int format;
  md_deflate_params params;
  md_gz_header gz;
  std::vector<char> name, comment;
  uint8_t *o;
  size_t o_len, o_pos;
  md_piece *dev;               // device side: text of the launch in flight, state, command queue, output of the piece
  std::vector<uint8_t> text;   // the input at absolute positions [w0, w0 + text.size())
  uint64_t w0, launched;       // ...of which [w0, launched) went through a launch already
  uint64_t origin;             // what the device counts positions from (they are 32-bit there): it moves up as the stream grows
  size_t out_len, served;      // output of the last launch (in device memory) / how much of it was handed out
  bool eoi, first, done;
  int status;
  uint32_t checksum;
};

extern "C" {

extern "C" int md_validate_deflate_params(md_ctx *ctx, int format, const md_deflate_params *params);

md_def_stream *md_def_encoder(md_ctx *ctx, int format, const md_deflate_params *params, uint8_t *o, size_t o_len) {
  if (!ctx || !params || !o || o_len == 0) return nullptr;
  // format, level, queue (a power of two >= 4), driver, matcher, window: refused here (md_last_error_string says
  // why), not at the end of the input — Zl.Def.encoder raises Invalid_argument at construction too (lib/de.ml:2286-2288)
  if (md_validate_deflate_params(ctx, format, params) != MD_OK) return nullptr;
  md_piece *dev = md_i_piece_open(ctx, params->queue_len);
  if (!dev) return nullptr;
  md_def_stream *s = new md_def_stream();
  s->ctx = ctx;
  s->format = format;
  s->params = *params;
  if (params->gz_header) {  // keep our own copy of the caller's strings
    s->gz = *params->gz_header;
    if (s->gz.filename) {
      s->name.assign(s->gz.filename, s->gz.filename + strlen(s->gz.filename) + 1);
      s->gz.filename = s->name.data();
    }
    if (s->gz.comment) {
      s->comment.assign(s->gz.comment, s->gz.comment + strlen(s->gz.comment) + 1);
      s->gz.comment = s->comment.data();
    }
    s->params.gz_header = &s->gz;
  }
  s->o = o;
  s->o_len = o_len;
  s->o_pos = s->served = s->out_len = 0;
  s->dev = dev;
  s->w0 = s->launched = s->origin = 0;
  s->eoi = s->done = false;
  s->first = true;
  s->status = MD_OK;
  s->checksum = format == MD_FORMAT_GZIP ? 0u : 1u;
  return s;
}
void md_def_free(md_def_stream *s) {
  if (!s) return;
  md_i_piece_close(s->ctx, s->dev);
  delete s;
}
int md_def_src(md_def_stream *s, const uint8_t *buf, size_t off, size_t len) {
  if (!s || (!buf && len) || s->eoi) return MD_E_INVALID_ARGUMENT;
  if (s->done && s->status != MD_OK) return MD_E_INVALID_ARGUMENT;  // (an encoder that ended with an error takes no more input)
  if (len == 0) {
    s->eoi = true;
    return MD_OK;
  }
  if (len > kSrcMax) return MD_E_INVALID_ARGUMENT;  // (one launch takes what has arrived: positions within it are 32-bit)
  // ... and so is the text that waits for md_def_encode: src calls without an encode in between must not pile up more than
  // one launch can take (mdeflate.h: call md_def_encode between sources; it launches what has arrived)
  if ((s->w0 + s->text.size()) - s->launched + len > kSrcMax) return MD_E_INVALID_ARGUMENT;
  s->text.insert(s->text.end(), buf + off, buf + off + len);
  s->checksum = s->format == MD_FORMAT_GZIP ? crc32_update(s->checksum, buf + off, len) : adler32_update(s->checksum, buf + off, len);
  return MD_OK;
}
void md_def_dst(md_def_stream *s, uint8_t *o, size_t o_len) {  // Zl.Def.dst: a fresh output buffer
  if (!s || !o || !o_len) return;
  s->o = o;
  s->o_len = o_len;
  s->o_pos = 0;
}
size_t md_def_dst_rem(const md_def_stream *s) { return s ? s->o_len - s->o_pos : 0; }
int md_def_status(const md_def_stream *s) { return s ? s->status : MD_E_INVALID_ARGUMENT; }
uint32_t md_def_checksum(const md_def_stream *s) { return s ? s->checksum : 0; }

// one launch over what has arrived: its output waits in device memory for md_def_encode to hand it out
static void def_launch(md_def_stream *s) {
  const uint64_t end = s->w0 + s->text.size();
  const size_t fresh = (size_t)(end - s->launched), ql = (size_t)s->params.queue_len;
  // room: the commands the queue held back at 6 bytes each (a match: two codes of 15 bits, 5 + 13 extra bits), the fresh
  // bytes at 2 bytes each (a literal is 15 bits at most, a match covers 3 bytes), a block header per queue fill, the frame
  const size_t blocks = (fresh + ql) / ql + 2, per_block = ql >= 128 ? 320 : 24 + 4 * ql;
  const size_t cap = 2048 + 6 * ql + 2 * fresh + blocks * per_block;
  // the device's positions are 32-bit: once the text is 2 GiB from their origin the origin moves up to 64 KiB below it
  // (deflate_test_flags bit 4: at 128 KiB already, so that a test of ordinary size goes through it)
  const uint64_t far = (md_i_test_flags(s->ctx) & 16) ? (uint64_t)1 << 17 : (uint64_t)1 << 31;
  uint64_t rebase = 0;
  if (s->w0 - s->origin >= far) {
    rebase = (s->w0 - s->origin - 65536) & ~(uint64_t)65535;
    s->origin += rebase;
  }
  int st = 0;
  const int rc = md_i_piece_run(s->ctx, s->dev, s->format, &s->params, s->text.data(), s->text.size(), (size_t)(s->launched - s->w0),
                                s->w0 - s->origin, rebase,
                                s->first, s->eoi, s->checksum, (uint32_t)end, cap, &s->out_len, &st);
  s->first = false;
  s->served = 0;
  s->launched = end;
  if (rc != MD_OK || (st != MD_OK && st != kPieceAwait)) {
    s->status = rc != MD_OK ? rc : st;
    s->out_len = 0;
    s->done = true;
    return;
  }
  if (st == MD_OK) s->done = true;  // (the last piece: trailer written)
  // the next launch sees the last 64 KiB again (the matcher reaches 32 KiB - 262 behind a position it has yet to take,
  // and those are less than 262 from the end): the text before goes
  if (end > kKeepBytes) {
    const uint64_t nw0 = (end - kKeepBytes) & ~(uint64_t)63;
    if (nw0 > s->w0) {
      s->text.erase(s->text.begin(), s->text.begin() + (size_t)(nw0 - s->w0));
      s->w0 = nw0;
    }
  }
}

int md_def_encode(md_def_stream *s) {
  if (!s) return MD_MALFORMED;
  for (;;) {
    if (s->served < s->out_len) {
      const size_t left = s->out_len - s->served, room = s->o_len - s->o_pos;
      const size_t k = left < room ? left : room;
      if (k && md_i_piece_out(s->ctx, s->dev, s->served, s->o + s->o_pos, k) != MD_OK) {
        s->status = MD_E_HIP;
        s->done = true;
        s->out_len = s->served = 0;
        return MD_MALFORMED;
      }
      s->o_pos += k;
      s->served += k;
      if (s->served < s->out_len) return MD_FLUSH;
    }
    if (s->done) return s->status == MD_OK ? MD_END : MD_MALFORMED;
    const size_t fresh = (size_t)(s->w0 + s->text.size() - s->launched);
    // (a launch costs three kernels whatever it holds: input is gathered, 1 MiB unless md_set_option "encoder_piece_bytes")
    if (!s->eoi && fresh < md_i_piece_bytes(s->ctx)) return MD_AWAIT;
    def_launch(s);
  }
}


// ---- many streaming encoders at once (md_def_batch_*, mdeflate.h) ------------------------------------------------------
// n independent Zl.Def / Gz.Def / De.Def encoders (lib/zl.ml:509-555) with the same parameters whose pieces go through the
// kernels TOGETHER: one launch of the three kernels per md_def_batch_encode whatever n is (md_def_* is one launch per
// encoder and piece), and an encoder's window - the last 64 KiB of its text - STAYS in device memory: only the bytes that
// arrived since the launch before cross the link, packed into one copy.  The text of launch k + 1 is gathered on the device
// from the tail of launch k's and the fresh bytes (piece_gather_kernel); state, queue, rebasing and the `Await protocol
// are the single encoder's (def_launch above, struct Piece), so the bytes of every encoder are those of md_def_* - and of
// the reference - handed the same pieces.
struct md_def_batch {
  md_ctx *ctx = nullptr;
  int format = 0;
  md_deflate_params params{};
  md_gz_header gz{};
  std::vector<char> name, comment;
  size_t n = 0;
  struct Enc {
    std::vector<uint8_t> fresh;   // input handed over since the last launch
    uint64_t w0 = 0, end = 0;     // the device holds the text of absolute positions [w0, end) ...
    uint64_t text_off = 0;        // ... at this offset of the current text blob
    uint64_t origin = 0;          // what the device counts this stream's positions from (32-bit there)
    bool eoi = false, first = true, done = false, launched_eoi = false;
    int status = MD_OK;
    uint32_t checksum = 0;
    uint64_t out_off = 0, out_len = 0, served = 0;  // output of the last launch in the device's output blob
    std::vector<uint8_t> held;    // output of earlier launches that was not fetched before the next one
  };
  std::vector<Enc> e;
  void *d_text[2] = {nullptr, nullptr};
  size_t text_cap[2] = {0, 0};
  int cur = 0;
  void *d_fresh = nullptr, *d_out = nullptr, *d_state = nullptr, *d_queue = nullptr, *d_desc = nullptr, *d_gdesc = nullptr;
  size_t fresh_cap = 0, out_cap = 0, desc_bytes = 0, gdesc_cap = 0;
  uint8_t *h_stage = nullptr;     // pinned: the fresh bytes of a launch, packed
  size_t stage_cap = 0;
};
namespace {
struct DevGuard {
  int prev = -1;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) hipSetDevice(dev);
  }
  ~DevGuard() {
    if (prev >= 0) hipSetDevice(prev);
  }
};
bool regrow(void **p, size_t *cap, size_t need) {
  if (need <= *cap) return true;
  if (*p) hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = need + need / 4 + 4096;
  if (hipMalloc(p, want) != hipSuccess) return false;
  *cap = want;
  return true;
}
}  // namespace

md_def_batch *md_def_batch_open(md_ctx *ctx, int format, const md_deflate_params *params, size_t n) {
  if (!ctx || !params || n == 0 || n > 0x7fffffffu) return nullptr;
  if (md_validate_deflate_params(ctx, format, params) != MD_OK) return nullptr;
  DevGuard guard(md_i_device(ctx));
  md_def_batch *b = new md_def_batch();
  b->ctx = ctx;
  b->format = format;
  b->params = *params;
  if (params->gz_header) {
    b->gz = *params->gz_header;
    if (b->gz.filename) {
      b->name.assign(b->gz.filename, b->gz.filename + strlen(b->gz.filename) + 1);
      b->gz.filename = b->name.data();
    }
    if (b->gz.comment) {
      b->comment.assign(b->gz.comment, b->gz.comment + strlen(b->gz.comment) + 1);
      b->gz.comment = b->comment.data();
    }
    b->params.gz_header = &b->gz;
  }
  b->n = n;
  b->e.resize(n);
  for (auto &x : b->e) x.checksum = format == MD_FORMAT_GZIP ? 0u : 1u;
  if (hipMalloc(&b->d_state, n * (size_t)md_piece_state_bytes()) != hipSuccess ||
      hipMalloc(&b->d_queue, n * (size_t)params->queue_len * 4) != hipSuccess) {
    hipFree(b->d_state);
    hipFree(b->d_queue);
    delete b;
    return nullptr;
  }
  return b;
}
void md_def_batch_close(md_def_batch *b) {
  if (!b) return;
  DevGuard guard(md_i_device(b->ctx));
  hipStreamSynchronize(md_i_stream(b->ctx));
  void *bufs[] = {b->d_text[0], b->d_text[1], b->d_fresh, b->d_out, b->d_state, b->d_queue, b->d_desc, b->d_gdesc};
  for (void *p : bufs)
    if (p) hipFree(p);
  if (b->h_stage) hipHostFree(b->h_stage);
  delete b;
}
int md_def_batch_src(md_def_batch *b, size_t i, const uint8_t *buf, size_t len) {
  if (!b || i >= b->n || (!buf && len)) return MD_E_INVALID_ARGUMENT;
  md_def_batch::Enc &x = b->e[i];
  if (x.eoi || x.done) return MD_E_INVALID_ARGUMENT;  // (an encoder that ended, also with an error, takes no more input)
  if (len == 0) {
    x.eoi = true;
    return MD_OK;
  }
  if (x.fresh.size() + len > kSrcMax) return MD_E_INVALID_ARGUMENT;  // (what one launch takes: call md_def_batch_encode in between)
  x.fresh.insert(x.fresh.end(), buf, buf + len);
  x.checksum = b->format == MD_FORMAT_GZIP ? crc32_update(x.checksum, buf, len) : adler32_update(x.checksum, buf, len);
  return MD_OK;
}
size_t md_def_batch_pending(const md_def_batch *b, size_t i) {
  if (!b || i >= b->n) return 0;
  const md_def_batch::Enc &x = b->e[i];
  return x.held.size() + (size_t)(x.out_len - x.served);
}
int md_def_batch_status(const md_def_batch *b, size_t i) {  // the signal md_def_encode would give
  if (!b || i >= b->n) return MD_MALFORMED;
  const md_def_batch::Enc &x = b->e[i];
  if (x.done) return x.status == MD_OK ? MD_END : MD_MALFORMED;
  return MD_AWAIT;
}
int md_def_batch_error(const md_def_batch *b, size_t i) {  // the MD_* status behind MD_MALFORMED (MD_OK otherwise)
  if (!b || i >= b->n) return MD_E_INVALID_ARGUMENT;
  return b->e[i].status;
}
uint32_t md_def_batch_checksum(const md_def_batch *b, size_t i) { return b && i < b->n ? b->e[i].checksum : 0; }
size_t md_def_batch_out(md_def_batch *b, size_t i, uint8_t *dst, size_t cap) {
  if (!b || i >= b->n || (!dst && cap)) return 0;
  md_def_batch::Enc &x = b->e[i];
  size_t got = 0;
  if (!x.held.empty()) {
    const size_t k = x.held.size() < cap ? x.held.size() : cap;
    memcpy(dst, x.held.data(), k);
    x.held.erase(x.held.begin(), x.held.begin() + k);
    got = k;
  }
  if (got < cap && x.served < x.out_len) {
    DevGuard guard(md_i_device(b->ctx));
    const size_t left = (size_t)(x.out_len - x.served), k = left < cap - got ? left : cap - got;
    if (hipMemcpy(dst + got, (const uint8_t *)b->d_out + x.out_off + x.served, k, hipMemcpyDeviceToHost) != hipSuccess) return got;
    x.served += k;
    got += k;
  }
  return got;
}
// One launch over what has arrived for every encoder since the last one.  Encoders without new input (and whose end of
// input has not been signalled since) sit the launch out.  MD_OK, or the call-level error.
int md_def_batch_encode(md_def_batch *b) {
  if (!b) return MD_E_INVALID_ARGUMENT;
  DevGuard guard(md_i_device(b->ctx));
  hipStream_t st = md_i_stream(b->ctx);
  const size_t n = b->n, ql = (size_t)b->params.queue_len;
  // output that was not fetched yet moves to the host: the launch writes a new output blob
  for (size_t i = 0; i < n; i++) {
    md_def_batch::Enc &x = b->e[i];
    if (x.served < x.out_len) {
      const size_t k = (size_t)(x.out_len - x.served), at = x.held.size();
      x.held.resize(at + k);
      if (hipMemcpy(x.held.data() + at, (const uint8_t *)b->d_out + x.out_off + x.served, k, hipMemcpyDeviceToHost) != hipSuccess) return MD_E_HIP;
    }
    x.out_len = x.served = 0;
  }
  std::vector<uint64_t> text_off(n), text_len(n), abs_len(n), out_off(n), out_cap(n), w0(n), rebase(n, 0), out_len(n, 0), g(6 * n);
  std::vector<uint32_t> flags(n), sum(n), isize(n);
  std::vector<int32_t> status(n, 0);
  const uint64_t far = (md_i_test_flags(b->ctx) & 16) ? (uint64_t)1 << 17 : (uint64_t)1 << 31;
  uint64_t tpos = 0, fpos = 0, opos = 0;
  uint32_t skip = 0xffffffffu;
  size_t active = 0;
  for (size_t i = 0; i < n; i++) {
    md_def_batch::Enc &x = b->e[i];
    const bool act = !x.done && (!x.fresh.empty() || (x.eoi && !x.launched_eoi));
    const uint64_t keep = x.end - x.w0, fresh = act ? x.fresh.size() : 0;
    g[6 * i] = x.text_off;
    g[6 * i + 1] = x.done ? 0 : keep;
    g[6 * i + 2] = fpos;
    g[6 * i + 3] = fresh;
    g[6 * i + 4] = tpos;
    g[6 * i + 5] = 0;
    fpos += (fresh + 15) & ~(uint64_t)15;
    if (act) {
      if (x.w0 - x.origin >= far) {  // 32-bit positions on the device: the origin moves up (def_launch above)
        rebase[i] = (x.w0 - x.origin - 65536) & ~(uint64_t)65535;  // (committed to x.origin only once the launch has succeeded:
      }                                                             //  a call that fails before it leaves every encoder retryable)
      const uint64_t origin = x.origin + rebase[i];
      const uint64_t end = x.end + fresh;
      const size_t blocks = (fresh + ql) / ql + 2, per_block = ql >= 128 ? 320 : 24 + 4 * ql;
      out_cap[i] = 2048 + 6 * ql + 2 * fresh + blocks * per_block;
      flags[i] = (x.first ? 1u : 0u) | (x.eoi ? 2u : 0u);
      abs_len[i] = end - origin;
      w0[i] = x.w0 - origin;
      isize[i] = (uint32_t)end;
      const uint64_t seen = keep;
      const uint32_t sk = seen > 512 ? (uint32_t)(seen - 512) : 0u;
      skip = sk < skip ? sk : skip;
      active++;
    } else {
      flags[i] = 8u;
      out_cap[i] = 16;
      abs_len[i] = w0[i] = 0;
      isize[i] = 0;
    }
    sum[i] = x.checksum;
    text_off[i] = tpos;
    text_len[i] = (x.done ? 0 : keep) + fresh;
    out_off[i] = opos;
    tpos += ((x.done ? 0 : keep) + fresh + 320 + 63) & ~(uint64_t)63;
    opos += (out_cap[i] + 63) & ~(uint64_t)63;
  }
  if (active == 0) return MD_OK;
  const int nxt = b->cur ^ 1;
  if (!regrow(&b->d_text[nxt], &b->text_cap[nxt], (size_t)tpos + 64) || !regrow(&b->d_fresh, &b->fresh_cap, (size_t)fpos + 64) ||
      !regrow(&b->d_out, &b->out_cap, (size_t)opos + 64) || !regrow(&b->d_gdesc, &b->gdesc_cap, 6 * n * 8))
    return MD_E_OUT_OF_MEMORY;
  if (fpos + 64 > b->stage_cap) {
    if (b->h_stage) hipHostFree(b->h_stage);
    b->h_stage = nullptr;
    b->stage_cap = 0;
    const size_t want = (size_t)fpos + (size_t)fpos / 4 + 4096;
    if (hipHostMalloc((void **)&b->h_stage, want, hipHostMallocDefault) != hipSuccess) return MD_E_OUT_OF_MEMORY;
    b->stage_cap = want;
  }
  for (size_t i = 0; i < n; i++)
    if (g[6 * i + 3]) memcpy(b->h_stage + g[6 * i + 2], b->e[i].fresh.data(), (size_t)g[6 * i + 3]);
  if (fpos && hipMemcpyAsync(b->d_fresh, b->h_stage, (size_t)fpos, hipMemcpyHostToDevice, st) != hipSuccess) return MD_E_HIP;
  if (hipMemcpyAsync(b->d_gdesc, g.data(), 6 * n * 8, hipMemcpyHostToDevice, st) != hipSuccess) return MD_E_HIP;
  if (md_launch_piece_gather((uint32_t)n, (const uint8_t *)b->d_text[b->cur], (const uint8_t *)b->d_fresh, (uint8_t *)b->d_text[nxt],
                             (const uint64_t *)b->d_gdesc, st) != 0)
    return MD_E_HIP;
  md_pieces_io io{text_off.data(), text_len.data(), abs_len.data(), out_off.data(), out_cap.data(), w0.data(), rebase.data(),
                  flags.data(), sum.data(), isize.data(), out_len.data(), status.data()};
  const int rc = md_i_pieces_run(b->ctx, b->format, &b->params, n, (const uint8_t *)b->d_text[nxt], (uint8_t *)b->d_out, b->d_state,
                                 b->d_queue, &b->d_desc, &b->desc_bytes, &io, skip == 0xffffffffu ? 0u : skip);
  if (rc != MD_OK) return rc;
  b->cur = nxt;
  for (size_t i = 0; i < n; i++) {
    md_def_batch::Enc &x = b->e[i];
    x.text_off = text_off[i];
    if (flags[i] & 8u) continue;
    x.origin += rebase[i];
    x.end += x.fresh.size();
    x.fresh.clear();
    x.first = false;
    if (x.eoi) x.launched_eoi = true;
    if (status[i] != MD_OK && status[i] != kPieceAwait) {
      x.status = status[i];
      x.done = true;
      continue;
    }
    x.out_off = out_off[i];
    x.out_len = out_len[i];
    x.served = 0;
    if (status[i] == MD_OK) x.done = true;  // (the last piece: trailer written)
  }
  // the next launch sees the last 64 KiB of every text again; what lies before goes (the gather takes the tail only)
  for (size_t i = 0; i < n; i++) {
    md_def_batch::Enc &x = b->e[i];
    if (x.done || x.end <= kKeepBytes) continue;
    const uint64_t nw0 = (x.end - kKeepBytes) & ~(uint64_t)63;
    if (nw0 > x.w0) {
      x.text_off += nw0 - x.w0;
      x.w0 = nw0;
    }
  }
  return MD_OK;
}

}  // extern "C"
