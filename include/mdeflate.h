/* mdeflate.h — C ABI of the MI355X-native many-stream DEFLATE engine.
 *
 * This is the drop-in boundary for the hot path of mirage/decompress
 * (lib/de.ml, lib/zl.ml).  The reference has no FFI for this path (its old
 * ctypes reverse binding was removed in v1.5.3, CHANGES.md:21-26), so the
 * boundary is the OCaml module signature; each entry point below names the
 * signature it stands behind.  An OCaml stub layer keeps
 * `De.Inf.Ns.inflate`, `Zl.Inf.Ns.inflate`, `De.Higher.*`, `Zl.Higher.*`
 * unchanged on top of these calls.
 *
 * Conventions:
 *  - caller owns every buffer; the engine never allocates caller-visible memory;
 *  - errors are integer status codes, 1:1 with the reference's variants,
 *    never exceptions / aborts;
 *  - a context (md_ctx) is single-owner, distinct contexts are independent
 *    and thread-safe with respect to each other (one context per GPU);
 *  - all sizes/offsets are bytes; descriptor arrays are structure-of-arrays.
 *  - there is NO CPU fallback: every compute entry point runs HIP kernels and
 *    returns MD_E_NO_DEVICE when no gfx950 device is usable.
 *  - limits per stream (the kernels keep 32-bit cursors): inflate reads at most
 *    MD_MAX_INFLATE_IN (512 MiB - 16: bit positions are 32-bit) of compressed input and writes at most
 *    MD_MAX_STREAM (4 GiB - 16); deflate / LZO read and write at most MD_MAX_STREAM.  The host-pointer
 *    entry points reject larger descriptors with MD_E_INVALID_ARGUMENT; with device descriptors the
 *    stream's status[i] is MD_E_INVALID_ARGUMENT and nothing is read or written for it.
 *    A batch holds at most 2^31 - 1 streams.
 */
#ifndef MDEFLATE_H
#define MDEFLATE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_VERSION 0x000300 /* 0.3.0 */
#define MD_MAX_INFLATE_IN 0x1ffffff0ull
#define MD_MAX_STREAM 0xfffffff0ull

/* Per-stream status: De.Inf.Ns.error (lib/de.ml:1548-1566, lib/de.mli:150-157)
 * + Zl.Inf.Ns.error (lib/zl.ml:383).  Strings: md_status_string(). */
enum {
  MD_OK = 0,
  MD_UNEXPECTED_END_OF_INPUT = 1,
  MD_UNEXPECTED_END_OF_OUTPUT = 2,
  MD_INVALID_KIND_OF_BLOCK = 3,
  MD_INVALID_DICTIONARY = 4,
  MD_INVALID_COMPLEMENT_OF_LENGTH = 5,
  MD_INVALID_DISTANCE = 6,
  MD_INVALID_DISTANCE_CODE = 7,
  MD_INVALID_HEADER = 8,   /* Zl: "Invalid Zlib header", lib/zl.ml:183 */
  MD_INVALID_CHECKSUM = 9, /* Zl: "Invalid checksum", lib/zl.ml:179-181; Gz: lib/gz.ml:287-289 */
  /* Gz.Inf's `Malformed strings, lib/gz.ml:284-296 */
  MD_INVALID_GZIP_HEADER = 10,          /* "Invalid GZip header" */
  MD_INVALID_GZIP_HEADER_CHECKSUM = 11, /* "Invalid GZip header checksum" */
  MD_INVALID_SIZE = 12,                 /* "Invalid input size (expect:.., inflated:..)" */
  /* deflate: the reference raises exception De.Queue.Full (lib/de.ml:2211-2217) — only the CLI
   * driver can, when its unconditional end-of-block push (bin/decompress.ml:67) meets a full queue */
  MD_QUEUE_FULL = 13,
  /* Lzo.error, lib/lzo.ml:4-12 (LZO entry points below) */
  MD_LZO_INVALID_INPUT = 14, /* `Malformed "Invalid input" (count, lib/lzo.ml:236) */
  MD_LZO_NO_DICTIONARY = 15, /* `Malformed "No dictionary at offset 0 available" (lib/lzo.ml:376) */
  MD_LZO_OUT_OF_BOUND = 16   /* `Invalid_argument "Input is malformed or output is not large enough"
                              * (lib/lzo.ml:401-402); compress: "lzo: output is not large enough" (:655) */
};

/* Call-level errors (negative): misuse raises Invalid_argument in the
 * reference (lib/de.ml:146-147); here the call returns one of these. */
enum {
  MD_E_INVALID_ARGUMENT = -1,
  MD_E_NO_DEVICE = -2,
  MD_E_HIP = -3,
  MD_E_OUT_OF_MEMORY = -4
};

/* Container formats */
enum {
  MD_FORMAT_DEFLATE = 0, /* raw RFC1951: De.Inf.Ns.inflate, lib/de.ml:1807-1822 */
  MD_FORMAT_ZLIB = 1,    /* RFC1950: Zl.Inf.Ns.inflate, lib/zl.ml:400-417 */
  MD_FORMAT_GZIP = 2     /* RFC1952 as lib/gz.ml reads and writes it: Gz.Inf (lib/gz.ml:248-633),
                          * Gz.Def (lib/gz.ml:636-918).  Inflate: checksum[i] = CRC-32 of the output;
                          * the body follows De.Inf.Ns status semantics.  Deflate: header from
                          * md_deflate_params.gz_header, body = the Zl driver's with dynamic blocks (Gz.Def's
                          * make_block, lib/gz.ml:724-729), CRC-32 + ISIZE trailer; `driver` and
                          * `dynamic` are ignored, checksum[i] = CRC-32 of the input. */
};

/* The reference's three encoder drivers: they decide WHEN a new block is sent, hence the
 * block structure of the stream. */
enum {
  MD_DRIVER_ZL = 0,     /* Zl.Def.encode / Zl.Higher.compress, lib/zl.ml:509-555 */
  MD_DRIVER_HIGHER = 1, /* De.Higher.compress / to_string, lib/de.ml:4518-4553 (always level 4) */
  MD_DRIVER_CLI = 2     /* bin/decompress.ml:47-75 (Dynamic per fill, Fixed last block) */
};

/* The reference's two match finders (same queue / histograms / drivers around them). */
enum {
  MD_MATCHER_DE = 0, /* De.Lz77, lib/de.ml:4013-4515: 4-byte multiplicative hash (default) */
  MD_MATCHER_LZ = 1  /* Lz (decompress.lz), lib/lz.ml:136-573: zlib's 3-byte rolling hash, levels 0..4 = 4,
                      * no copy mode, `End without an end-of-block command (the driver pushes it) */
};

typedef struct md_ctx md_ctx;

int md_version(void);
/* human strings of lib/de.ml:1557-1567 ("Unexpected end of input", ...) */
const char *md_status_string(int status);
/* last call-level error text of this context (HIP error string etc.) */
const char *md_last_error_string(const md_ctx *ctx);

/* Number of usable gfx950 devices (0 when none / no driver). */
int md_device_count(void);

/* Multi-GPU: streams are independent, so a batch shards with no data-path exchange - one context
 * (md_create(device)), one host thread and one HIP stream per device, every device given a contiguous range of the
 * stream index.  md_shard_plan cuts a batch of n streams into `world` such ranges balanced by the sum of `lengths`
 * (normally the uncompressed sizes): rank r owns streams [lo[r], hi[r]); the ranges are contiguous, cover 0..n and may
 * be empty.  Pure host arithmetic (no device is touched), the same cut as decompress_amd.shard.shard_by_bytes.
 * Returns MD_OK, or MD_E_INVALID_ARGUMENT (world < 1, a null pointer with n or world != 0). */
int md_shard_plan(uint64_t n, const uint64_t *lengths, int world, uint64_t *lo, uint64_t *hi);

/* One context per GPU.  `hip_stream`:
 *   NULL            the context creates its own non-blocking stream (not ordered with any other stream:
 *                   the caller synchronises, e.g. md_synchronize, before touching the buffers elsewhere);
 *   MD_STREAM_NULL  the device's legacy default stream (hipStream_t 0): ordered with everything the caller
 *                   enqueues there;
 *   otherwise       an existing hipStream_t to enqueue on (e.g. the caller's current stream).
 * Every entry point sets the context's device for the call and restores the caller's current device.
 * Replaces nothing in the reference: state objects there are the window/queue bigarrays the caller
 * allocates (lib/de.mli:93-106); here they are LDS buffers owned by the kernels. */
#define MD_STREAM_NULL ((void *)(intptr_t)-1)
md_ctx *md_create(int device, void *hip_stream);
void md_destroy(md_ctx *ctx);
int md_synchronize(md_ctx *ctx);

/* Timing of the dominant kernel with HIP events recorded on the context's
 * stream: begin/end bracket any number of batch calls; end returns elapsed
 * milliseconds (synchronises). */
/* Options of a context.  Keys:
 *   "deflate_workspace_cap_mib"  value >= 0 (default: a sixth of the device's memory, 48 GiB on MI355X; 0 = none).  The
 *                                deflate kernels keep a per-position workspace of 13 bytes per input byte of what one
 *                                launch covers; a md_deflate_batch_device call whose workspace would be larger than the
 *                                cap goes through the kernels in slices of positions - a multiple of 32 KiB of every
 *                                stream per launch, going on from the state the launch before left - and in groups of
 *                                streams if 64 KiB of every stream at once would still be too much.  Same bytes out
 *                                (every fill of the reference's window ends on a 32 KiB boundary).  Batches of
 *                                more than four times 4 096 streams go in groups of streams first (no extra work).
 *                                4 096 x 1 MiB: 52 GiB and 114.7 ms whole; 27.8 GiB (2 slices) 116.3 ms at the default;
 *                                14.8 GiB (4 slices) 119.1 ms at 16 GiB; 6.7 GiB 127 ms at 8 GiB.  32 768 corpus files
 *                                (7.1 GB, 93 GB whole): 573 ms whole, 576 ms at the default (two groups), 656 ms at
 *                                29 GiB (two groups, each in slices: a slice lasts as long as its slowest stream).
 *                                A capped call reads lengths and results back between the launches, i.e. it
 *                                synchronises with the context's stream instead of only enqueueing.
 *   "encoder_piece_bytes"        value >= 1 (default 1 MiB): how much input a md_def_* encoder gathers before it launches
 *                                the kernels on it.  The bytes out are those of the reference handed the input in the
 *                                same pieces (which are, but for corner cases at the very end of a stream, the same for
 *                                any pieces).
 *   "release_workspace"          (value ignored) waits for the context's stream and frees its grow-only device scratch
 *                                (deflate workspaces, launch orders, decoder-piece buffers, the host entry points' device copies); it grows again on demand.
 *   "host_pipeline_slices"       1 .. 64 (default 16): most slices of streams a md_*_batch_host call cuts a batch into so that
 *                                its copies overlap with its kernels (1: copy-in, kernels, copy-out one after the other).
 *   "inflate_waves"              1 or 2 (default): wavefronts per stream of the inflate kernel (2 = decoder + copier).
 *   "inflate_parallel_min"       KiB (default 96; 0 = never): a SINGLE stream handed to md_*_inf_ns_inflate /
 *                                md_*_higher_uncompress with at least this much compressed input is decoded in pieces by the
 *                                whole device (csrc/inflate_chunked.hip: candidate block starts, every piece decoded twice
 *                                with placeholder windows by the batch kernel, windows resolved afterwards, checksum verified)
 *                                instead of by one pair of wavefronts.  Results are the same by construction: whatever is not
 *                                a well-formed stream that fits its buffer falls back to the serial path and gets its status.
 *   "inflate_parallel_chunk"     KiB of compressed input per piece (default 64, 4 .. 2^20).
 *   "inflate_parallel_last"      a QUERY (value ignored; the return value is the answer, not a status): pieces of the last
 *                                single stream that went that way | decode rounds << 24; 0 = it took the serial path.
 *   "debug_inflate_lds_pad", "debug_known_bounds"   measurement aids of tools/dbg (occupancy curve, known-boundaries floor).
 *   "profile"                    0 / 1: in-kernel phase profile of stream 0 (md_get_profile, a debugging aid).
 * Unknown keys and values out of range: MD_E_INVALID_ARGUMENT. */
int md_set_option(md_ctx *ctx, const char *key, int value);
/* Pinned (page-locked) host memory for the buffers handed to the md_*_host entry points: copies from and to it are DMA
 * transfers that overlap with the kernels.  NULL when the allocation fails.  (An OCaml caller wraps it in a Bigarray with
 * caml_ba_alloc and frees it from the custom block's finaliser: INTEGRATION.md.) */
void *md_host_alloc(md_ctx *ctx, size_t bytes);
void md_host_free(md_ctx *ctx, void *p); /* (ctx may be NULL, or already destroyed: it is not used) */
int md_timing_begin(md_ctx *ctx);
int md_timing_end(md_ctx *ctx, float *ms);

/* Batched inflate of n independent streams, everything resident in HBM.
 *   stream i reads  d_in [in_off[i],  in_off[i]  + in_len[i])
 *            writes d_out[out_off[i], out_off[i] + out_cap[i])
 * Results per stream (device arrays, may not be NULL unless noted):
 *   out_len[i]   bytes written          (the `o` of Ok (i, o))
 *   consumed[i]  input bytes consumed   (the `i` of Ok (i, o); 0 on error)
 *   status[i]    MD_OK or the error variant
 *   checksum[i]  Adler-32 of the output (may be NULL)
 * Semantics per stream = De.Inf.Ns.inflate (lib/de.ml:1807-1822) for
 * MD_FORMAT_DEFLATE, Zl.Inf.Ns.inflate (lib/zl.ml:400-417) for MD_FORMAT_ZLIB.
 * Asynchronous on the context's stream.  Returns MD_OK or a call-level error. */
int md_inflate_batch_device(md_ctx *ctx, int format, size_t n,
                            const uint8_t *d_in, const uint64_t *d_in_off,
                            const uint64_t *d_in_len, uint8_t *d_out,
                            const uint64_t *d_out_off, const uint64_t *d_out_cap,
                            uint64_t *d_out_len, uint64_t *d_consumed,
                            int32_t *d_status, uint32_t *d_checksum);

/* Same with HOST pointers (the reference's callers own host bigarrays, lib/de.mli:93-106): copies inputs H2D, runs the
 * kernels, copies results D2H and synchronises.  h_in/h_out are the packed buffers the offsets index.  A batch of many
 * streams goes through in slices of consecutive streams (md_set_option "host_pipeline_slices", default up to 16; a slice
 * keeps enough streams to fill the device), the copy-in of the next slice and the copy-out of the one before running
 * under the kernels of the current one - which they only do when h_in / h_out are PINNED host memory (md_host_alloc
 * below, or the caller's own hipHostMalloc / hipHostRegister); pageable buffers give the same results, copied one
 * after the other.  The device copies of the blobs belong to the context and are kept between calls
 * (md_set_option "release_workspace" frees them). */
int md_inflate_batch_host(md_ctx *ctx, int format, size_t n, const uint8_t *h_in,
                          size_t in_bytes, const uint64_t *in_off,
                          const uint64_t *in_len, uint8_t *h_out, size_t out_bytes,
                          const uint64_t *out_off, const uint64_t *out_cap,
                          uint64_t *out_len, uint64_t *consumed, int32_t *status,
                          uint32_t *checksum);

/* Single-stream mirrors of the reference's whole-buffer entry points (host
 * pointers, batch of one):
 *   De.Inf.Ns.inflate : bigstring -> bigstring -> (int * int, error) result
 *   (lib/de.mli:146-173)  and  Zl.Inf.Ns.inflate (lib/zl.mli, lib/zl.ml:400). */
int md_de_inf_ns_inflate(md_ctx *ctx, const uint8_t *src, size_t src_len,
                         uint8_t *dst, size_t dst_cap, size_t *consumed,
                         size_t *written);
int md_zl_inf_ns_inflate(md_ctx *ctx, const uint8_t *src, size_t src_len,
                         uint8_t *dst, size_t dst_cap, size_t *consumed,
                         size_t *written);

/* The header fields Gz.Def.encoder takes (lib/gz.ml:859-918): ?ascii ?hcrc ?filename ?comment ~mtime os.
 * filename / comment may be NULL (absent; both must be NUL-free and < 256 bytes).  XFL follows the level
 * (2 for level 9, else 0, lib/gz.ml:888-890).  A NULL md_gz_header* means mtime 0, os 3 (Unix), nothing else. */
typedef struct md_gz_header {
  uint32_t mtime;
  int os, hcrc, ascii;
  const char *filename, *comment;
} md_gz_header;

/* What `Zl.Def.encoder ?dynamic ~q ~w ~level` / `De.Lz77.state ?level ~q ~w` / `Lz.state` / `Gz.Def.encoder` take
 * as arguments (lib/zl.mli, lib/de.mli:453-524, lib/lz.mli:1-19, lib/gz.ml:859): per call, nothing is kept in the
 * context. */
typedef struct md_deflate_params {
  int level;     /* 0..9 (lib/de.ml:4030-4049; De.Higher ignores it: always 4, H6) */
  int queue_len; /* De.Queue capacity, a power of two (4096 in the reference's bench / CLI) */
  int driver;    /* MD_DRIVER_* */
  int dynamic;   /* Zl.Def's ?dynamic: 0 -> Fixed blocks */
  int matcher;   /* MD_MATCHER_* */
  const md_gz_header *gz_header; /* MD_FORMAT_GZIP only; NULL = default header */
  int wbits;     /* log2 of the sliding window `De.Lz77.state ~w` derives from its window buffer (lib/de.ml:4462-4464):
                  * 0 or 15 = De.make_window ~bits:15, the only size the reference's callers use and the only one
                  * the kernels implement; anything else is MD_E_INVALID_ARGUMENT */
  size_t total_in_bytes; /* batch calls with device descriptors: an upper bound of the sum of in_len[i], or 0 when the
                  * caller does not know it.  The engine sizes a per-position workspace (13 bytes per input byte
                  * plus 319 positions of padding per stream: hash-chain links, flags and two look-ahead verdicts;
                  * grow-only for the life of the context, and never above md_set_option "deflate_workspace_cap_mib": a batch
                  * that would need more goes in slices of positions) from it; with 0 it reads the sum back from the device first,
                  * i.e. the call waits for the work already enqueued on the context's stream.  A batch whose descriptors
                  * add up to more than the hint gets status[i] = MD_E_INVALID_ARGUMENT for every stream. */
} md_deflate_params;

/* Batched deflate of n independent buffers, everything resident in HBM.
 *   stream i reads d_in[in_off[i], +in_len[i]), writes d_out[out_off[i], +out_cap[i]).
 * Per stream the output is byte-identical to the reference's De.Lz77 (lib/de.ml:4013-4515; or Lz, lib/lz.ml)
 * + De.Def (lib/de.ml:2354-3038) run by params->driver with a command queue of params->queue_len entries:
 *   MD_FORMAT_DEFLATE  the raw body,
 *   MD_FORMAT_ZLIB     Zl.Def framing: 0x78xx header + body + Adler-32 (lib/zl.ml:511-522, 494-499),
 *   MD_FORMAT_GZIP     Gz.Def framing (see MD_FORMAT_GZIP).
 * Results: out_len[i], status[i] (MD_OK, MD_UNEXPECTED_END_OF_OUTPUT when out_cap[i] is too small,
 * MD_QUEUE_FULL), checksum[i] = Adler-32 (CRC-32 for GZip) of the input (may be NULL).  Asynchronous on the
 * context's stream. */
int md_deflate_batch_device(md_ctx *ctx, int format, const md_deflate_params *params, size_t n,
                            const uint8_t *d_in, const uint64_t *d_in_off, const uint64_t *d_in_len,
                            uint8_t *d_out, const uint64_t *d_out_off, const uint64_t *d_out_cap,
                            uint64_t *d_out_len, int32_t *d_status, uint32_t *d_checksum);

/* Same with HOST pointers (H2D, kernels, D2H, synchronise). */
int md_deflate_batch_host(md_ctx *ctx, int format, const md_deflate_params *params, size_t n,
                          const uint8_t *h_in, size_t in_bytes, const uint64_t *in_off,
                          const uint64_t *in_len, uint8_t *h_out, size_t out_bytes,
                          const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len,
                          int32_t *status, uint32_t *checksum);

/* Single-buffer mirrors of the reference's drivers (host pointers, batch of one):
 *   De.Higher.compress ~w ~q ~refill ~flush i o   (lib/de.mli:533-600): raw DEFLATE, level 4
 *   Zl.Higher.compress ?level ?dynamic ~w ~q ...  (lib/zl.mli, lib/zl.ml:634-648): zlib stream
 *   De.Higher.uncompress / Zl.Higher.uncompress   (lib/de.ml:4555-4571, lib/zl.ml:650-666)
 * The reference's refill/flush callbacks become one source and one destination buffer;
 * *written is the output size.  Returns MD_OK / a status (its md_status_string is the reference's
 * `Msg) / a call error. */
int md_de_higher_compress(md_ctx *ctx, int queue_len, const uint8_t *src, size_t src_len,
                          uint8_t *dst, size_t dst_cap, size_t *written);
int md_zl_higher_compress(md_ctx *ctx, int level, int dynamic, int queue_len, const uint8_t *src,
                          size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written);
int md_de_higher_uncompress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                            size_t *written);
int md_zl_higher_uncompress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                            size_t *written);

/* The two halves of the encoder on their own (host pointers, batch of one), on the same HIP kernel:
 *
 * De.Lz77.compress (lib/de.mli:453-524; `Lz.compress` with MD_MATCHER_LZ): the match finder alone.  cmds
 * receives the commands of every queue fill in order, in De.Queue's encoding (lib/de.ml:2245-2266: a literal is
 * its byte, end-of-block is 256, a copy is 0x2000000 | (length - 3) << 16 | (offset - 1)); a fill ends with the
 * end-of-block command the reference pushes when one cell is left (H1).  literals[286] / distances[30] receive
 * De.Lz77.literals / distances, cumulative over the input (may be NULL).  MD_UNEXPECTED_END_OF_OUTPUT when
 * cmds_cap is too small (*ncmds is then the number needed).
 *
 * De.Def.encode (lib/de.mli:300-412): the bit encoder alone, the way test/test.ml's `encode` uses it — the
 * commands are ONE last block of `kind`; a Dynamic block gets its trees from the commands' own frequencies
 * (dynamic_of_frequencies, lib/de.ml:2367-2403). */
enum { MD_BLOCK_FLAT = 0, MD_BLOCK_FIXED = 1, MD_BLOCK_DYNAMIC = 2 };
int md_de_lz77_compress(md_ctx *ctx, int level, int queue_len, int matcher, const uint8_t *src, size_t src_len,
                        uint32_t *cmds, size_t cmds_cap, size_t *ncmds, uint32_t *literals, uint32_t *distances);
int md_de_def_encode(md_ctx *ctx, int kind, const uint32_t *cmds, size_t ncmds, uint8_t *dst, size_t dst_cap,
                     size_t *written);

/* De.Def.encode (lib/de.ml:2965-3038) driven step by step, the way the reference's own tests drive it
 * (test/test_ns.ml:388-615, test/test.ml:533-767): one encoder over a queue of queue_len cells and an unbounded
 * `Buffer destination, fed a list of operations (32-bit words):
 *   MD_OP_FILL n c1..cn        Queue.push_exn of n commands (De.Queue's encoding, see md_de_lz77_compress);
 *                              MD_QUEUE_FULL when the queue has no room (exception Queue.Full)
 *   MD_OP_BLOCK kind last      Def.encode e (`Block {kind; last}); kind MD_BLOCK_*; a Dynamic block is
 *                              Def.dynamic_of_frequencies ~literals ~distances of the frequencies counted so far
 *                              (and mutates them the way T.make does)
 *   MD_OP_FLUSH                Def.encode e `Flush
 *   MD_OP_SUCC_LITERAL chr / MD_OP_SUCC_LENGTH len / MD_OP_SUCC_DISTANCE dist   De.succ_* (lib/de.ml:2339-2351)
 *   MD_OP_NEW_FREQS            make_literals () / make_distances ()
 *   MD_OP_QUEUE_RESET          Queue.reset
 * results[k] = what the k-th encode answered: 0 `Ok, 1 `Block (`Partial cannot happen with a `Buffer); *nresults =
 * how many answers results[] received.  At most 315 encodes per call (and results_cap): a list with more is
 * MD_E_INVALID_ARGUMENT after its bytes have been written.  dst receives the bytes written.  A malformed list is
 * MD_E_INVALID_ARGUMENT. */
enum { MD_OP_FILL = 1, MD_OP_BLOCK = 2, MD_OP_FLUSH = 3, MD_OP_SUCC_LITERAL = 4, MD_OP_SUCC_LENGTH = 5,
       MD_OP_SUCC_DISTANCE = 6, MD_OP_NEW_FREQS = 7, MD_OP_QUEUE_RESET = 8 };
int md_de_def_run(md_ctx *ctx, int queue_len, const uint32_t *ops, size_t nops, uint8_t *dst, size_t dst_cap,
                  size_t *written, uint8_t *results, size_t results_cap, size_t *nresults);

/* ---- De.Def.Ns / Zl.Def.Ns (lib/de.ml:3040-4010, lib/zl.ml:596-629): the reference's whole-buffer compressor ----
 * `De.Def.Ns.deflate ?level src dst : (int, [> error ]) result`, an OCaml port of libdeflate's greedy path with its own
 * block splitting and Huffman construction; output byte-identical to it:
 *   level 1..4        compress_greedy (search depth 2 / 6 / 12 / 24, nice length 8 / 10 / 14 / 24);
 *   level 5..12       upstream's compress_lazy is a stub: MD_OK with *written = 0 (lib/de.ml:3927);
 *   level 0           MD_UNEXPECTED_END_OF_OUTPUT for inputs of 56 bytes or more — upstream's write_uncompressed_blocks
 *                     never advances its input and can only leave through that error (lib/de.ml:3411-3420); the same
 *                     when a block of levels 1..4 would be cheapest uncompressed;
 *   inputs shorter than 56 - 4 * level bytes are one stored block; dst needs 8 bytes of slack (compress_bound has them);
 *   other levels      MD_E_INVALID_ARGUMENT (`Invalid_compression_level).
 * md_zl_def_ns_deflate adds Zl.Def.Ns's header (FLEVEL from its own level map, H9) and the Adler-32.
 * Batch form: as md_deflate_batch_device, format MD_FORMAT_DEFLATE or MD_FORMAT_ZLIB, total_in_bytes as in
 * md_deflate_params; checksum[i] = Adler-32 of the input (may be NULL). */
int md_de_def_ns_deflate(md_ctx *ctx, int level, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                         size_t *written);
int md_zl_def_ns_deflate(md_ctx *ctx, int level, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                         size_t *written);
size_t md_de_def_ns_compress_bound(size_t len); /* De.Def.Ns.compress_bound, lib/de.ml:3994-3997 */
size_t md_zl_def_ns_compress_bound(size_t len); /* lib/zl.ml:600 */
int md_def_ns_batch_device(md_ctx *ctx, int format, int level, size_t total_in_bytes, size_t n, const uint8_t *d_in,
                           const uint64_t *d_in_off, const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                           const uint64_t *d_out_cap, uint64_t *d_out_len, int32_t *d_status, uint32_t *d_checksum);

/* ---- a DEFLATE stream decoded in pieces (De.Inf.decode's `Flush while input is still arriving, lib/de.ml:1427-1474) ----
 * md_de_inf_continue_host decodes as much of a raw DEFLATE stream as the piece src[0, src_len) holds.  The piece
 * starts start_bit (0..7) bits into src[0]; dst begins with hist_len (<= 32768) bytes of what was decoded before
 * it — the window: matches may reach into them —, decoding writes from dst[hist_len] on, and adler_in is the Adler-32
 * state to go on from (1 for a new stream).  On return *dst_len is the output position reached (history included),
 * *status the stream status for this piece (MD_OK: the final block ended inside it; MD_UNEXPECTED_END_OF_INPUT: the
 * piece ended inside a block — decode the next piece from `resume`; anything else as md_inflate_batch_host), and
 * resume describes the end of the last block that was complete in the piece: bits from src[0] (start_bit included),
 * output position and checksum state there, whether it was the final block.  What lies between resume->out and
 * *dst_len belongs to the incomplete block: it is valid output, but the next piece starts at the block boundary and
 * produces it again.  (md_inf_decode does this by itself once a stream is longer than md_inf_chunk_bytes.) */
typedef struct md_inf_resume {
  uint64_t bits, out;       /* end of the last complete block: input bits from src[0], output position (history included) */
  uint32_t adler, last;     /* Adler-32 state there; 1 when that block was the final one */
  uint64_t consumed;        /* status MD_OK: input bytes of the piece the stream used (as md_inflate_batch_host) */
  uint32_t checksum;        /* Adler-32 state at *dst_len */
  uint32_t crc_out, crc_end; /* flags & MD_CONT_CRC32: CRC-32 of dst[hist_len, out) and of dst[hist_len, *dst_len) */
} md_inf_resume;
enum { MD_CONT_CRC32 = 1 }; /* also compute the CRC-32 of the piece's new output (Gz.Inf's checksum, lib/gz.ml:503) */
/* The same for n streams at once, everything resident in HBM (raw DEFLATE; descriptors as md_inflate_batch_device):
 * piece i starts d_start_bit[i] bits into its first byte, its output buffer begins with d_hist_len[i] bytes of window,
 * its checksum goes on from d_adler_in[i]; results as md_inflate_batch_device (d_out_len includes the window) plus the
 * last block boundary inside each piece: d_resume_bits / d_resume_out (u64), d_resume_adler, d_resume_last (u32). */
int md_inflate_continue_batch_device(md_ctx *ctx, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                                     const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                     const uint64_t *d_out_cap, const uint32_t *d_start_bit, const uint32_t *d_hist_len,
                                     const uint32_t *d_adler_in, uint64_t *d_out_len, uint64_t *d_consumed,
                                     int32_t *d_status, uint32_t *d_checksum, uint64_t *d_resume_bits,
                                     uint64_t *d_resume_out, uint32_t *d_resume_adler, uint32_t *d_resume_last);
int md_de_inf_continue_host(md_ctx *ctx, const uint8_t *src, size_t src_len, unsigned start_bit, uint8_t *dst, size_t hist_len,
                            size_t dst_cap, uint32_t adler_in, unsigned flags, size_t *dst_len, int *status,
                            md_inf_resume *resume);

/* ---- the resumable state machines (host side; one launch at the end of input) ----
 * De.Inf.decoder / decode / src / flush / dst_rem / src_rem / checksum (lib/de.mli:82-144) and the encoder loop of
 * Zl.Def / Gz.Def / De.Higher with `Manual source and destination (lib/zl.ml:509-555): the caller supplies input
 * with src (length 0 = end of input, as in the reference), calls decode / encode, and consumes its output buffer
 * whenever it gets MD_FLUSH (then md_inf_flush / md_def_dst), until MD_END or MD_MALFORMED (md_*_status gives
 * the MD_* status whose string is the reference's `Malformed message).  See csrc/stream_shim.cpp. */
/* Memory: the encoder works in pieces, like the reference's (whose state is its window, its queue and one output
 * buffer).  md_def_src appends to a host buffer; once md_set_option "encoder_piece_bytes" (default 1 MiB) have gathered -
 * or the end of the input is signalled - md_def_encode launches the kernels on them: they go on from the state the
 * launch before left in device memory (12 KiB and the command queue), answer `Await inside the kernel exactly where
 * De.Lz77 would (lookahead under 262 and nothing left of the piece), and the piece's output is handed out through
 * `Flush steps while more input arrives.  Host and device each hold the last 64 KiB of the stream and the piece in
 * flight; a stream may be of any length (positions are rebased inside, gzip's ISIZE wraps at 2^32 as lib/gz.ml's);
 * one md_def_src call takes at most 1 GiB, and so does what md_def_src calls have handed over since the last md_def_encode
 * (MD_E_INVALID_ARGUMENT beyond that: call md_def_encode between sources - it launches what has arrived).  The bytes are those of the reference handed the input in the same pieces
 * (fill_window's slide depends on how much each fill finds; for whole streams and for pieces the oracle agrees).  The decoder
 * (DEFLATE, ZLIB, GZip) works in pieces: once md_inf_chunk_bytes (default 8 MiB; a piece of at least "inflate_parallel_min" is decoded by the whole device) of input are buffered
 * it decodes up to the last block boundary inside them (md_de_inf_continue_host), hands that output out through
 * `Flush steps while input is still arriving, and keeps only the undecoded tail and the 32 KiB window; a stream that
 * ends before a piece is full is decoded in one launch as before.
 * md_inf_message: the reference's `Malformed string with its numbers, e.g. "Invalid checksum (expect:%04lx,
 * has:%04lx)" (lib/zl.ml:179-181, lib/gz.ml:287-289), "Invalid input size (expect:%ld, inflated:%ld)"
 * (lib/gz.ml:291-293); md_status_string(md_inf_status) is its fixed part.  md_inf_reset = De.Inf.reset
 * (lib/de.ml:1512-1532): the same decoder for another stream. */
enum { MD_AWAIT = 0, MD_FLUSH = 1, MD_END = 2, MD_MALFORMED = 3 };
typedef struct md_inf_stream md_inf_stream;
md_inf_stream *md_inf_decoder(md_ctx *ctx, int format, uint8_t *o, size_t o_len);
void md_inf_reset(md_inf_stream *s);
void md_inf_chunk_bytes(md_inf_stream *s, size_t bytes); /* input buffered before a piece is decoded (>= 1) */
const char *md_inf_message(const md_inf_stream *s);
int md_inf_src(md_inf_stream *s, const uint8_t *buf, size_t off, size_t len);
int md_inf_decode(md_inf_stream *s);
void md_inf_flush(md_inf_stream *s);
size_t md_inf_dst_rem(const md_inf_stream *s);
size_t md_inf_src_rem(const md_inf_stream *s);
int md_inf_status(const md_inf_stream *s);
uint32_t md_inf_checksum(const md_inf_stream *s);
void md_inf_free(md_inf_stream *s);
typedef struct md_def_stream md_def_stream;
md_def_stream *md_def_encoder(md_ctx *ctx, int format, const md_deflate_params *params, uint8_t *o, size_t o_len);
int md_def_src(md_def_stream *s, const uint8_t *buf, size_t off, size_t len);
int md_def_encode(md_def_stream *s);
void md_def_dst(md_def_stream *s, uint8_t *o, size_t o_len);
size_t md_def_dst_rem(const md_def_stream *s);
int md_def_status(const md_def_stream *s);
uint32_t md_def_checksum(const md_def_stream *s);
void md_def_free(md_def_stream *s);

/* ---- many streaming encoders at once ----
 * n independent Zl.Def / Gz.Def / De.Def encoders (lib/zl.ml:509-555: every `Zl.Def.encoder` is its own state machine) with
 * the same parameters, advanced TOGETHER: md_def_batch_src hands input to encoder i (host memory, copied; length 0 = the
 * end of its input, as in the reference), md_def_batch_encode is ONE launch of the kernels over what has arrived for all
 * of them since the last one - whatever n is - and leaves every encoder's output of that launch in device memory, from
 * where md_def_batch_out copies it out (md_def_batch_pending says how much waits; what is not fetched before the next
 * encode is kept on the host side).  md_def_batch_status: MD_AWAIT (more input wanted), MD_END (the trailer is written:
 * the stream is complete once its pending output is fetched) or MD_MALFORMED (md_def_batch_error: that stream's MD_* status).  An encoder's window
 * (the last 64 KiB of its text), its state and its queue stay in device memory between launches: only new bytes cross the
 * link, in one copy per launch.  The bytes of every encoder are those of md_def_* - and of the reference - handed the same
 * pieces (a piece = what arrived between two md_def_batch_encode calls).  One md_def_batch_src hands over 1 GiB at most. */
typedef struct md_def_batch md_def_batch;
md_def_batch *md_def_batch_open(md_ctx *ctx, int format, const md_deflate_params *params, size_t n);
int md_def_batch_src(md_def_batch *b, size_t i, const uint8_t *buf, size_t len);
int md_def_batch_encode(md_def_batch *b);
size_t md_def_batch_pending(const md_def_batch *b, size_t i);
size_t md_def_batch_out(md_def_batch *b, size_t i, uint8_t *dst, size_t cap);
int md_def_batch_status(const md_def_batch *b, size_t i);
int md_def_batch_error(const md_def_batch *b, size_t i);
uint32_t md_def_batch_checksum(const md_def_batch *b, size_t i);
void md_def_batch_close(md_def_batch *b);

/* ---- GZip (lib/gz.ml) ---- */

/* What Gz.Inf.filename / comment / os / extra report (lib/gz.ml:612-633): offsets into src. */
typedef struct md_gz_meta {
  uint32_t flg, mtime, xfl, os;
  int has_extra, has_name, has_comment;
  size_t extra_off, extra_len, name_off, name_len, comment_off, comment_len;
} md_gz_meta;

/* Single-buffer mirrors (host pointers, batch of one):
 *   Gz.Higher.compress ?level ?filename ?comment ~w ~q ... (lib/gz.ml:927-950; NB its ?level
 *   defaults to 0 there — pass the level you mean)
 *   Gz.Higher.uncompress ~refill ~flush i o            (lib/gz.ml:959-982); meta may be NULL. */
int md_gz_higher_compress(md_ctx *ctx, int level, int queue_len, const md_gz_header *header, const uint8_t *src,
                          size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written);
int md_gz_higher_uncompress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                            size_t *consumed, size_t *written, md_gz_meta *meta);

/* Checkseum.Crc32 of n buffers resident in HBM (call sites lib/gz.ml:503, :682): crc[i] =
 * CRC-32 of d_data[off[i], off[i] + len[i]).  Asynchronous on the context's stream. */
int md_crc32_batch_device(md_ctx *ctx, size_t n, const uint8_t *d_data, const uint64_t *d_off,
                          const uint64_t *d_len, uint32_t *d_crc);

/* ---- LZO1X (lib/lzo.ml; SURVEY 8(f) row 3, BASELINE config 5) ---- */

/* Lzo.uncompress input output (lib/lzo.ml:395-403) over n independent streams resident in HBM:
 * status[i] = MD_OK, MD_UNEXPECTED_END_OF_INPUT or one of MD_LZO_*; out_len[i] = bytes written
 * (0 on error).  Asynchronous on the context's stream. */
int md_lzo_uncompress_batch_device(md_ctx *ctx, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                                   const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                   const uint64_t *d_out_cap, uint64_t *d_out_len, int32_t *d_status);
/* Lzo.compress in_data out_data wrkmem (lib/lzo.ml:642-660; the 16 K-entry wrkmem is a per-stream
 * workspace owned by the context)
 * over n independent buffers: status[i] = MD_OK or MD_LZO_OUT_OF_BOUND when out_cap[i] is too
 * small (n + n/16 + 64 + 3 always suffices). */
int md_lzo_compress_batch_device(md_ctx *ctx, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                                 const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                 const uint64_t *d_out_cap, uint64_t *d_out_len, int32_t *d_status);
/* Single-buffer mirrors (host pointers, batch of one): Lzo.uncompress / Lzo.compress. */
int md_lzo_uncompress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                      size_t *written);
int md_lzo_compress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                    size_t *written);

#ifdef __cplusplus
}
#endif
#endif
