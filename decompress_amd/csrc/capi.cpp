// capi.cpp — host side of the C ABI declared in include/mdeflate.h.
// Plain HIP runtime calls; no torch types anywhere in this library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <string>
#include <vector>

#include "mdeflate.h"

extern "C" int md_launch_inflate_wave(int format, uint32_t n, const uint8_t *in, const uint64_t *in_off,
                                      const uint64_t *in_len, uint8_t *out, const uint64_t *out_off,
                                      const uint64_t *out_cap, uint64_t *out_len, uint64_t *consumed,
                                      int32_t *status, uint32_t *checksum, uint64_t *dbg, uint32_t *order,
                                      int waves, const void *cont_ptrs, hipStream_t stream);

// deflate: the front workspace (deflate_common.hpp) is opaque here
struct md_front {
  const void *p_end, *slot, *chunk0, *tail, *flags;
  void *link, *flg, *m, *mq;
};
extern "C" size_t md_deflate_queue_bytes(uint32_t n, int qcap);
// a stream's slot in the per-position workspace: its length + 64, rounded up to the match kernel's chunk; the chunk and
// the size of a stream's state slot are the kernels' constants (deflate_common.hpp), asked for, not copied
extern "C" uint32_t md_front_chunk();
extern "C" uint32_t md_piece_state_bytes();
static const unsigned long long kChunkPositions = md_front_chunk();
static const unsigned long long kSlotPad = 64 + kChunkPositions - 1;
extern "C" size_t md_front_small_bytes(uint32_t n);
extern "C" size_t md_front_big_bytes(uint64_t positions);
extern "C" void md_front_carve(void *small_ws, void *big_ws, uint32_t n, uint64_t positions, md_front *f);
extern "C" int md_launch_deflate_plan(uint32_t n, const uint64_t *in_len, int driver, int matcher, int level,
                                      uint64_t cap_positions, uint32_t cap_chunks, const md_front *f, hipStream_t stream);
extern "C" int md_launch_deflate_front(uint32_t n, uint32_t nchunks_max, const uint8_t *in, const uint64_t *in_off,
                                       const uint64_t *in_len, int matcher, uint32_t max_chain, uint32_t nice,
                                       const md_front *f, const uint32_t *order, uint32_t match_skip, hipStream_t stream);
extern "C" int md_i_debug_inflate_lds_pad(uint32_t bytes);
extern "C" int md_i_debug_known_bounds(int mode, uint32_t nstreams);
extern "C" int md_launch_stream_order(uint32_t n, const uint64_t *in_len, uint32_t *order, hipStream_t stream);
extern "C" void md_deflate_level_params(int driver, int matcher, int level, uint32_t *max_chain, uint32_t *nice);
extern "C" int md_launch_def_ns(int format, int level, uint32_t n, uint32_t nchunks_max, const uint8_t *in, const uint64_t *in_off,
                                const uint64_t *in_len, uint8_t *out, const uint64_t *out_off, const uint64_t *out_cap,
                                uint64_t *out_len, int32_t *status, uint32_t *checksum, const md_front *f, hipStream_t stream);
extern "C" int md_launch_deflate(int format, int level, int qcap, int driver, int dynamic, uint32_t n,
                                 const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                 uint8_t *out, const uint64_t *out_off, const uint64_t *out_cap,
                                 uint64_t *out_len, int32_t *status, uint32_t *checksum, const md_front *fr, void *queue_ws,
                                 uint64_t *dbg, const uint8_t *gz_hdr, uint32_t gz_hdr_len,
                                 const uint32_t *gz_crc, int matcher, uint32_t *hist, const uint32_t *order, const void *piece_ptrs,
                                 hipStream_t stream);

extern "C" int md_launch_gz_header(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                   uint64_t *body_off, uint64_t *body_len, int32_t *hstatus, hipStream_t stream);
extern "C" int md_launch_crc32(uint32_t n, const uint8_t *data, const uint64_t *off, const uint64_t *len,
                               uint32_t *crc_out, hipStream_t stream);
extern "C" int md_launch_gz_finish(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                   const uint64_t *body_off, const int32_t *hstatus, const uint8_t *out,
                                   const uint64_t *out_off, uint64_t *out_len, uint64_t *consumed, int32_t *status,
                                   uint32_t *checksum, hipStream_t stream);

extern "C" int md_launch_lzo_uncompress(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                        uint8_t *out, const uint64_t *out_off, const uint64_t *out_cap,
                                        uint64_t *out_len, int32_t *status, uint32_t *counter, uint32_t slots, hipStream_t stream);
extern "C" int md_launch_lzo_compress(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                      uint8_t *out, const uint64_t *out_off, const uint64_t *out_cap,
                                      uint64_t *out_len, int32_t *status, uint16_t *ws_dict, uint32_t *counter, uint32_t slots,
                                      hipStream_t stream);
extern "C" uint32_t md_lzo_slots(int compress, uint32_t cus);
// inflate_chunked.hip
extern "C" int md_launch_find_blocks(const uint8_t *body, uint64_t nbytes, uint64_t K, uint32_t nchunks_behind_first, uint64_t *cand,
                                     hipStream_t stream);
extern "C" int md_launch_fill_windows(uint32_t n, uint8_t *out, const uint64_t *out_off, const uint8_t *variant, hipStream_t stream);
extern "C" int md_launch_window_chain(uint32_t npieces, const uint8_t *dst, const uint8_t *scratch, const uint64_t *offa,
                                      const uint64_t *offb, const uint64_t *u, uint8_t *wins, uint32_t *flag, hipStream_t stream);
extern "C" size_t md_windows_work_bytes(uint32_t npieces, uint32_t group);
extern "C" int md_launch_windows_parallel(uint32_t npieces, uint32_t group, const uint8_t *dst, uint64_t u0, const uint8_t *scratch, const uint64_t *offa,
                                          const uint64_t *offb, const uint64_t *u, const uint64_t *pos, uint8_t *wins, uint32_t *work, uint32_t *flag,
                                          hipStream_t stream);
extern "C" int md_launch_resolve(uint32_t npieces, uint8_t *dst, const uint8_t *scratch, const uint64_t *offa, const uint64_t *offb,
                                 const uint64_t *u, const uint64_t *pos, const uint8_t *wins, uint32_t *flag, hipStream_t stream);
extern "C" int md_launch_adler_segments(const uint8_t *data, uint64_t n, uint32_t seg, uint32_t *sums, hipStream_t stream);

struct md_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // GZip: per-stream scratch (body offsets/lengths, header status, CRCs) and the header to write
  void *gz_tmp = nullptr;
  size_t gz_tmp_bytes = 0;
  uint8_t *gz_hdr_dev = nullptr;  // device copy of gz_hdr (530 bytes max)
  uint8_t gz_hdr_sent[544] = {0};  // what gz_hdr_dev holds
  bool gz_hdr_valid = false;
  void *lzo_ws = nullptr;  // Lzo.compress dictionaries
  size_t lzo_ws_bytes = 0;
  uint32_t *counters = nullptr;  // work counters of the kernels with persistent workgroups (zeroed in front of a launch)
  int cus = 256;                 // compute units of the device
  int inflate_waves = 2;    // wavefronts per stream of the inflate kernel (md_set_option "inflate_waves": 1 = the one-wavefront form)
  size_t piece_bytes = (size_t)1 << 20;  // md_set_option "encoder_piece_bytes": input the md_def_* encoder gathers before a launch
  size_t front_cap_bytes = 0;  // md_set_option "deflate_workspace_cap_mib" (md_create: a sixth of the device's memory; 0 = none):
                               // batches whose per-position workspace would be larger go in slices of positions
  int test_flags = 0;       // md_set_option "deflate_test_flags": bit 4 = the md_def_* encoder moves its origin every 128 KiB (tests)
  uint64_t *dbg = nullptr;  // device buffer of the optional in-kernel profile (32 x u64)
  // deflate workspaces, grow-only: command queues (n x queue_len), the per-stream part of the front workspace
  // (slots, chunk starts: n-sized) and its per-position part (hash-chain links, look-ahead verdicts: 13 bytes per input byte)
  void *ws = nullptr, *fsmall = nullptr, *fbig = nullptr;
  size_t ws_bytes = 0, fsmall_bytes = 0, fbig_bytes = 0;
  // the decoder in pieces (md_de_inf_continue_host): input, output and descriptor scratch, grow-only
  void *cont_in = nullptr, *cont_out = nullptr, *cont_desc = nullptr;
  size_t cont_in_bytes = 0, cont_out_bytes = 0, cont_desc_bytes = 0;
  // a deflate batch in slices of positions: descriptors of the slice and the streams' states between the slices, grow-only
  void *slice_desc = nullptr, *slice_state = nullptr;
  size_t slice_desc_bytes = 0, slice_state_bytes = 0;
  uint32_t *order = nullptr;  // inflate: launch order of a large batch (n words)
  size_t order_words = 0;
  // the host-buffer entry points (md_*_batch_host): device copies of the caller's blobs and descriptors, grow-only, and
  // the two copy streams that run next to the context's stream (copy-in of slice k + 1 and copy-out of slice k - 1 under
  // the kernels of slice k)
  void *host_in = nullptr, *host_out = nullptr, *host_desc = nullptr;
  size_t host_in_bytes = 0, host_out_bytes = 0, host_desc_bytes = 0;
  hipStream_t s_in = nullptr, s_out = nullptr;
  // one long stream decoded by the whole chip (inflate_parallel): input, output + the pieces' scratch decodes, windows and
  // descriptors, grow-only; md_set_option "inflate_parallel_min" (compressed bytes from which a single stream goes this
  // way, 0 = never) and "inflate_parallel_chunk" (compressed bytes per piece)
  void *par_in = nullptr, *par_out = nullptr, *par_win = nullptr, *par_desc = nullptr;
  size_t par_in_bytes = 0, par_out_bytes = 0, par_win_bytes = 0, par_desc_bytes = 0;
  size_t par_min = (size_t)96 << 10, par_chunk = (size_t)64 << 10;  // (measured: the pieces pay from ~100 KB of input, ~1 ms flat up to 4 MiB of text)
  int par_last_pieces = 0, par_last_rounds = 0;  // of the last stream that went this way (md_get_option, tests)
  int host_slices_max = 16;  // md_set_option "host_pipeline_slices": 1 = copy-in / kernels / copy-out one after the other
  std::string err;
};

namespace {
thread_local std::string g_err;

int fail(md_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess) {
  std::string m = what;
  if (e != hipSuccess) {
    m += ": ";
    m += hipGetErrorString(e);
  }
  if (ctx) ctx->err = m;
  g_err = m;
  return code;
}

#define HIP_TRY(ctx, expr)                                   \
  do {                                                       \
    hipError_t e_ = (expr);                                  \
    if (e_ != hipSuccess) return fail(ctx, MD_E_HIP, #expr, e_); \
  } while (0)

// every entry point works on the context's device and leaves the caller's current device as it found it
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    if (!ok) (void)hipGetLastError();  // (the call reports the failure itself: no stale error for whoever asks next)
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define MD_ON_DEVICE(ctx)                 \
  DeviceGuard guard_((ctx)->device);      \
  if (!guard_.ok) return fail(ctx, MD_E_HIP, "hipSetDevice")

constexpr size_t kOrderFrom = 2049;  // 256 CUs x 8 resident wavefronts: smaller batches start all at once
constexpr size_t kOrderFromDeflate = 257;  // the link kernel holds one stream per CU

bool is_gfx950(int dev) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}
}  // namespace

extern "C" {

int md_version(void) { return MD_VERSION; }

// contiguous ranges of the stream index balanced by bytes (decompress_amd/shard.py shard_by_bytes is
// the same arithmetic, in doubles): rank r's range ends with the last stream whose middle lies at or before r + 1
// world-ths of the total.
int md_shard_plan(uint64_t n, const uint64_t *lengths, int world, uint64_t *lo, uint64_t *hi) {
  if (world < 1 || !lo || !hi || (n && !lengths)) return MD_E_INVALID_ARGUMENT;
  double total = 0.0;
  for (uint64_t i = 0; i < n; i++) total += (double)lengths[i];
  double acc = 0.0;
  uint64_t at = 0;
  for (int r = 0; r < world; r++) {
    const double target = total * (double)(r + 1) / (double)world;
    uint64_t end = at;
    while (end < n && acc + (double)lengths[end] / 2.0 <= target) {
      acc += (double)lengths[end];
      end++;
    }
    if (r == world - 1) end = n;
    lo[r] = at;
    hi[r] = end;
    at = end;
  }
  return MD_OK;
}

const char *md_status_string(int s) {
  switch (s) {
  case MD_OK: return "Ok";
  case MD_UNEXPECTED_END_OF_INPUT: return "Unexpected end of input";
  case MD_UNEXPECTED_END_OF_OUTPUT: return "Unexpected end of output";
  case MD_INVALID_KIND_OF_BLOCK: return "Invalid kind of block";
  case MD_INVALID_DICTIONARY: return "Invalid dictionary";
  case MD_INVALID_COMPLEMENT_OF_LENGTH: return "Invalid complement of length";
  case MD_INVALID_DISTANCE: return "Invalid distance";
  case MD_INVALID_DISTANCE_CODE: return "Invalid distance code";
  case MD_INVALID_HEADER: return "Invalid Zlib header";
  case MD_INVALID_CHECKSUM: return "Invalid checksum";
  case MD_INVALID_GZIP_HEADER: return "Invalid GZip header";
  case MD_INVALID_GZIP_HEADER_CHECKSUM: return "Invalid GZip header checksum";
  case MD_INVALID_SIZE: return "Invalid input size";
  case MD_QUEUE_FULL: return "Queue.Full";
  case MD_LZO_INVALID_INPUT: return "Invalid input";
  case MD_LZO_NO_DICTIONARY: return "No dictionary at offset 0 available";
  case MD_LZO_OUT_OF_BOUND: return "Input is malformed or output is not large enough";
  case MD_E_INVALID_ARGUMENT: return "Invalid argument";
  case MD_E_NO_DEVICE: return "No gfx950 device";
  case MD_E_HIP: return "HIP runtime error";
  case MD_E_OUT_OF_MEMORY: return "Out of device memory";
  default: return "Unknown status";
  }
}

const char *md_last_error_string(const md_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_err.c_str();
}

int md_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; d++)
    if (is_gfx950(d)) ok++;
  return ok;
}

md_ctx *md_create(int device, void *hip_stream) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    fail(nullptr, MD_E_NO_DEVICE, "md_create: no such HIP device");
    return nullptr;
  }
  if (!is_gfx950(device)) {
    fail(nullptr, MD_E_NO_DEVICE, "md_create: device is not gfx950 (kernels are built for MI355X only)");
    return nullptr;
  }
  md_ctx *ctx = new md_ctx();
  ctx->device = device;
  DeviceGuard guard(device);
  if (!guard.ok) {
    delete ctx;
    fail(nullptr, MD_E_HIP, "hipSetDevice");
    return nullptr;
  }
  if (hip_stream == MD_STREAM_NULL) {
    ctx->stream = nullptr;  // the device's legacy default stream: ordered with everything else enqueued on it
  } else if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      fail(nullptr, MD_E_HIP, "hipStreamCreate");
      return nullptr;
    }
    ctx->own_stream = true;
  }
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    md_destroy(ctx);
    fail(nullptr, MD_E_HIP, "hipEventCreate");
    return nullptr;
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->cus = cus;
    if (hipMalloc((void **)&ctx->counters, 256) != hipSuccess) {
      md_destroy(ctx);
      fail(nullptr, MD_E_OUT_OF_MEMORY, "hipMalloc(counters)");
      return nullptr;
    }
  }
  {  // the deflate kernels' per-position workspace takes a sixth of the device at most (48 GiB of 288) unless told otherwise
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) ctx->front_cap_bytes = mem_total / 6;
  }
  return ctx;
}

void md_destroy(md_ctx *ctx) {
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  if (ctx->ws) hipFree(ctx->ws);
  if (ctx->fsmall) hipFree(ctx->fsmall);
  if (ctx->fbig) hipFree(ctx->fbig);
  if (ctx->order) hipFree(ctx->order);
  if (ctx->cont_in) hipFree(ctx->cont_in);
  if (ctx->cont_out) hipFree(ctx->cont_out);
  if (ctx->cont_desc) hipFree(ctx->cont_desc);
  if (ctx->slice_desc) hipFree(ctx->slice_desc);
  if (ctx->slice_state) hipFree(ctx->slice_state);
  if (ctx->dbg) hipFree(ctx->dbg);
  if (ctx->gz_tmp) hipFree(ctx->gz_tmp);
  if (ctx->lzo_ws) hipFree(ctx->lzo_ws);
  if (ctx->counters) hipFree(ctx->counters);
  for (void *q : {ctx->par_in, ctx->par_out, ctx->par_win, ctx->par_desc})
    if (q) hipFree(q);
  if (ctx->gz_hdr_dev) hipFree(ctx->gz_hdr_dev);
  if (ctx->host_in) hipFree(ctx->host_in);
  if (ctx->host_out) hipFree(ctx->host_out);
  if (ctx->host_desc) hipFree(ctx->host_desc);
  if (ctx->s_in) hipStreamDestroy(ctx->s_in);
  if (ctx->s_out) hipStreamDestroy(ctx->s_out);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

int md_synchronize(md_ctx *ctx) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  MD_ON_DEVICE(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MD_OK;
}

int md_timing_begin(md_ctx *ctx) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  MD_ON_DEVICE(ctx);
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return MD_OK;
}

int md_timing_end(md_ctx *ctx, float *ms) {
  if (!ctx || !ms) return MD_E_INVALID_ARGUMENT;
  MD_ON_DEVICE(ctx);
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
  HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return MD_OK;
}

void *md_host_alloc(md_ctx *ctx, size_t bytes) {
  if (!ctx) return nullptr;
  DeviceGuard guard(ctx->device);
  void *p = nullptr;
  if (!guard.ok || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    fail(ctx, MD_E_OUT_OF_MEMORY, "hipHostMalloc");
    return nullptr;
  }
  return p;
}
void md_host_free(md_ctx *ctx, void *p) {
  (void)ctx;  // (pinned memory is freed whatever the current device is - and a buffer may outlive its context)
  if (p) hipHostFree(p);
}

int md_set_option(md_ctx *ctx, const char *key, int value) {
  if (!ctx || !key) return MD_E_INVALID_ARGUMENT;
  if (!strcmp(key, "profile")) {  // in-kernel phase profile of stream 0 (debug builds of the kernel)
    if (value && !ctx->dbg) {
      if (hipMalloc((void **)&ctx->dbg, 32 * 8) != hipSuccess) return fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc");
      if (hipMemset(ctx->dbg, 0, 32 * 8) != hipSuccess) return fail(ctx, MD_E_HIP, "hipMemset");
    } else if (!value && ctx->dbg) {
      hipFree(ctx->dbg);
      ctx->dbg = nullptr;
    }
    return MD_OK;
  }
  if (!strcmp(key, "inflate_waves")) {
    if (value != 1 && value != 2) return fail(ctx, MD_E_INVALID_ARGUMENT, "inflate_waves is 1 or 2");
    ctx->inflate_waves = value;
    return MD_OK;
  }
  if (!strcmp(key, "deflate_workspace_cap_mib")) {  // 0 = no cap (one launch of each kernel per batch whatever it takes)
    if (value < 0) return fail(ctx, MD_E_INVALID_ARGUMENT, "deflate_workspace_cap_mib >= 0");
    ctx->front_cap_bytes = (size_t)value << 20;
    return MD_OK;
  }
  if (!strcmp(key, "encoder_piece_bytes")) {
    if (value < 1) return fail(ctx, MD_E_INVALID_ARGUMENT, "encoder_piece_bytes >= 1");
    ctx->piece_bytes = (size_t)value;
    return MD_OK;
  }
  if (!strcmp(key, "release_workspace")) {  // give the grow-only scratch of this context back (it grows again on demand)
    MD_ON_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    void **bufs[] = {&ctx->ws, &ctx->fsmall, &ctx->fbig, (void **)&ctx->order, &ctx->cont_in, &ctx->cont_out, &ctx->cont_desc,
                     &ctx->slice_desc, &ctx->slice_state, &ctx->host_in, &ctx->host_out, &ctx->host_desc,
                     &ctx->par_in, &ctx->par_out, &ctx->par_win, &ctx->par_desc};
    size_t *sizes[] = {&ctx->ws_bytes, &ctx->fsmall_bytes, &ctx->fbig_bytes, &ctx->order_words, &ctx->cont_in_bytes,
                       &ctx->cont_out_bytes, &ctx->cont_desc_bytes, &ctx->slice_desc_bytes, &ctx->slice_state_bytes,
                       &ctx->host_in_bytes, &ctx->host_out_bytes, &ctx->host_desc_bytes,
                       &ctx->par_in_bytes, &ctx->par_out_bytes, &ctx->par_win_bytes, &ctx->par_desc_bytes};
    for (size_t i = 0; i < sizeof bufs / sizeof bufs[0]; i++) {
      if (*bufs[i]) hipFree(*bufs[i]);
      *bufs[i] = nullptr;
      *sizes[i] = 0;
    }
    return MD_OK;
  }
  if (!strcmp(key, "deflate_test_flags")) {
    ctx->test_flags = value;
    return MD_OK;
  }
  if (!strcmp(key, "inflate_parallel_min")) {  // KiB of compressed input from which ONE stream is decoded in pieces by the whole chip; 0 = never
    if (value < 0) return fail(ctx, MD_E_INVALID_ARGUMENT, "inflate_parallel_min >= 0 (KiB)");
    ctx->par_min = (size_t)value << 10;
    return MD_OK;
  }
  if (!strcmp(key, "inflate_parallel_chunk")) {  // KiB of compressed input per piece
    if (value < 4 || value > (1 << 20)) return fail(ctx, MD_E_INVALID_ARGUMENT, "inflate_parallel_chunk is 4 .. 2^20 (KiB)");
    ctx->par_chunk = (size_t)value << 10;
    return MD_OK;
  }
  if (!strcmp(key, "inflate_parallel_last")) {  // (query, value ignored) pieces of the last stream that went that way | rounds << 24; 0: it did not
    return ctx->par_last_pieces | (ctx->par_last_rounds << 24);
  }
  if (!strcmp(key, "host_pipeline_slices")) {  // md_*_batch_host: slices of streams in flight (1 = no overlap of copies and kernels)
    if (value < 1 || value > 64) return fail(ctx, MD_E_INVALID_ARGUMENT, "host_pipeline_slices is 1 .. 64");
    ctx->host_slices_max = value;
    return MD_OK;
  }
  if (!strcmp(key, "debug_known_bounds")) {  // measurement builds only (-DMD_DEBUG_KNOWN_BOUNDS): mode | streams << 5
    MD_ON_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (value < 0 || md_i_debug_known_bounds(value & 31, (uint32_t)value >> 5)) return fail(ctx, MD_E_INVALID_ARGUMENT, "debug_known_bounds: not a measurement build");
    return MD_OK;
  }
  if (!strcmp(key, "debug_inflate_lds_pad")) {  // measurement only: unused LDS per stream, i.e. fewer streams per CU
    MD_ON_DEVICE(ctx);
    if (value < 0 || md_i_debug_inflate_lds_pad((uint32_t)value)) return fail(ctx, MD_E_INVALID_ARGUMENT, "debug_inflate_lds_pad");
    return MD_OK;
  }
  return fail(ctx, MD_E_INVALID_ARGUMENT, "unknown option");
}

// copies the 32 profile words of the last v2 launch to host (after synchronising)
int md_get_profile(md_ctx *ctx, uint64_t *out32) {
  if (!ctx || !out32 || !ctx->dbg) return MD_E_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(out32, ctx->dbg, 32 * 8, hipMemcpyDeviceToHost));
  return MD_OK;
}

// per-stream GZip scratch: body_off[n] u64, body_len[n] u64, hstatus[n] i32, crc[n] u32
static int gz_scratch(md_ctx *ctx, size_t n) {
  const size_t need = n * 24;
  if (need > ctx->gz_tmp_bytes) {
    if (ctx->gz_tmp) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      HIP_TRY(ctx, hipFree(ctx->gz_tmp));
      ctx->gz_tmp = nullptr;
      ctx->gz_tmp_bytes = 0;
    }
    if (hipMalloc(&ctx->gz_tmp, need) != hipSuccess) return fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc(gzip scratch)");
    ctx->gz_tmp_bytes = need;
  }
  return MD_OK;
}

// the scratch for the launch order of a large batch (4 bytes per stream): kept by the context, only ever grows — the
// one allocation a batch call can make, on its first large batch
static int order_scratch(md_ctx *ctx, size_t n) {
  if (n <= ctx->order_words) return MD_OK;
  if (ctx->order) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(ctx->order));
    ctx->order = nullptr;
    ctx->order_words = 0;
  }
  if (hipMalloc((void **)&ctx->order, n * 4) != hipSuccess) return fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc(launch order)");
  ctx->order_words = n;
  return MD_OK;
}

int md_inflate_batch_device(md_ctx *ctx, int format, size_t n, const uint8_t *d_in,
                            const uint64_t *d_in_off, const uint64_t *d_in_len, uint8_t *d_out,
                            const uint64_t *d_out_off, const uint64_t *d_out_cap,
                            uint64_t *d_out_len, uint64_t *d_consumed, int32_t *d_status,
                            uint32_t *d_checksum) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  if (format != MD_FORMAT_DEFLATE && format != MD_FORMAT_ZLIB && format != MD_FORMAT_GZIP)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "unknown format");
  if (n == 0) return MD_OK;
  if (n > 0x7fffffffull) return fail(ctx, MD_E_INVALID_ARGUMENT, "too many streams in one batch");
  if (!d_in_off || !d_in_len || !d_out_off || !d_out_cap || !d_out_len || !d_consumed || !d_status)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "null descriptor array");
  MD_ON_DEVICE(ctx);
  if (format == MD_FORMAT_GZIP) {
    // Gz.Inf = header, De.Inf on the body, checksum (lib/gz.ml:463-531, :344-356)
    int rc = gz_scratch(ctx, n);
    if (rc != MD_OK) return rc;
    uint64_t *body_off = (uint64_t *)ctx->gz_tmp, *body_len = body_off + n;
    int32_t *hstatus = (int32_t *)(body_len + n);
    int e = md_launch_gz_header((uint32_t)n, d_in, d_in_off, d_in_len, body_off, body_len, hstatus, ctx->stream);
    if (e != 0) return fail(ctx, MD_E_HIP, "gz header kernel launch", (hipError_t)e);
    rc = md_inflate_batch_device(ctx, MD_FORMAT_DEFLATE, n, d_in, body_off, body_len, d_out, d_out_off, d_out_cap,
                                 d_out_len, d_consumed, d_status, nullptr);
    if (rc != MD_OK) return rc;
    e = md_launch_gz_finish((uint32_t)n, d_in, d_in_off, d_in_len, body_off, hstatus, d_out, d_out_off, d_out_len,
                            d_consumed, d_status, d_checksum, ctx->stream);
    if (e != 0) return fail(ctx, MD_E_HIP, "gz finish kernel launch", (hipError_t)e);
    return MD_OK;
  }
  // a batch of more streams than are resident at once (8 per CU) is started longest stream first; the scratch for the
  // order is kept and only ever grows (the one allocation a batch call can make, on its first large batch)
  uint32_t *order = nullptr;
  if (n >= kOrderFrom) {
    const int orc = order_scratch(ctx, n);
    if (orc != MD_OK) return orc;
    order = ctx->order;
  }
  int rc = md_launch_inflate_wave(format, (uint32_t)n, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len,
                                  d_consumed, d_status, d_checksum, ctx->dbg, order, ctx->inflate_waves, nullptr, ctx->stream);
  if (rc != 0) return fail(ctx, MD_E_HIP, "inflate kernel launch", (hipError_t)rc);
  return MD_OK;
}

namespace {
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) hipFree(p);
  }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
}  // namespace

// ---- host buffers in, host buffers out (the reference's callers own host bigarrays, lib/de.mli:93-106) ----------------
// The batch goes through the device in SLICES of consecutive streams: the copy-in of slice k + 1 (stream s_in) and the
// copy-out of slice k - 1 (stream s_out) run under the kernels of slice k (the context's stream); events order the three.
// A slice copies the byte range its streams span in the caller's blob (ranges of different slices may overlap or lie in
// any order: a byte copied twice is copied with the same value, and an output range is final when its slice's kernels
// are).  With pinned host buffers (hipHostMalloc / hipHostRegister) the copies are DMA transfers and really overlap;
// pageable buffers go through the runtime's staging and mostly do not.  The device copies of the blobs are the context's,
// grow-only (md_set_option "release_workspace" gives them back).
static int grow(md_ctx *ctx, void **buf, size_t *have, size_t need, const char *what);
extern "C++" {
namespace {
struct HostSlice {
  size_t i0, i1;
  uint64_t in_lo, in_hi, out_lo, out_hi;
};
// slices of about equal bytes (input + output room), at least min_streams streams each, at most max_slices
std::vector<HostSlice> host_slices(size_t n, const uint64_t *in_off, const uint64_t *in_len, const uint64_t *out_off,
                                   const uint64_t *out_cap, size_t min_streams, size_t max_slices, uint64_t in_bytes, uint64_t out_bytes) {
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) total += in_len[i] + out_cap[i];
  size_t want = (size_t)(total / ((uint64_t)64 << 20));
  if (want > max_slices) want = max_slices;
  if (min_streams && want > n / min_streams) want = n / min_streams;
  if (want < 1) want = 1;
  std::vector<HostSlice> v;
  uint64_t acc = 0, span = 0;
  size_t i0 = 0;
  for (size_t k = 0; k < want; k++) {
    const uint64_t upto = total / want * (k + 1);
    size_t i1 = i0;
    while (i1 < n && (k + 1 == want || acc < upto || i1 - i0 < min_streams)) {  // (never fewer than min_streams, whatever the sizes: ADVICE r5)
      acc += in_len[i1] + out_cap[i1];
      i1++;
    }
    if (i1 == i0) continue;
    HostSlice sl{i0, i1, ~0ull, 0, ~0ull, 0};
    for (size_t i = i0; i < i1; i++) {
      if (in_len[i]) {
        sl.in_lo = in_off[i] < sl.in_lo ? in_off[i] : sl.in_lo;
        sl.in_hi = in_off[i] + in_len[i] > sl.in_hi ? in_off[i] + in_len[i] : sl.in_hi;
      }
      if (out_cap[i]) {
        sl.out_lo = out_off[i] < sl.out_lo ? out_off[i] : sl.out_lo;
        sl.out_hi = out_off[i] + out_cap[i] > sl.out_hi ? out_off[i] + out_cap[i] : sl.out_hi;
      }
    }
    if (sl.in_hi <= sl.in_lo) sl.in_lo = sl.in_hi = 0;
    if (sl.out_hi <= sl.out_lo) sl.out_lo = sl.out_hi = 0;
    span += (sl.in_hi - sl.in_lo) + (sl.out_hi - sl.out_lo);
    v.push_back(sl);
    i0 = i1;
  }
  // a layout whose slices span much more than the blobs hold (streams scattered across the blob): one slice, whole blobs
  // (a batch that is one slice anyway copies the span of its streams like any other slice: the caller's blobs may hold
  // more than this batch - md_inflate_batch_host picks streams out of one)
  if (v.empty() || (v.size() > 1 && span > in_bytes + out_bytes + (in_bytes + out_bytes) / 4)) {
    v.clear();
    v.push_back(HostSlice{0, n, 0, in_bytes, 0, out_bytes});
  }
  return v;
}
struct EventList {
  std::vector<hipEvent_t> ev;
  ~EventList() {
    for (hipEvent_t e : ev) hipEventDestroy(e);
  }
  hipEvent_t make() {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    ev.push_back(e);
    return e;
  }
};
}  // namespace
}  // extern "C++"

// `launch(i0, count)` enqueues the device entry point for streams [i0, i0 + count) on ctx->stream
extern "C++" {
template <class Launch>
static int host_pipeline(md_ctx *ctx, const std::vector<HostSlice> &sl, const uint8_t *h_in, uint8_t *d_in, uint8_t *h_out, uint8_t *d_out,
                         Launch launch) {
  if (!ctx->s_in && hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking) != hipSuccess) return fail(ctx, MD_E_HIP, "hipStreamCreate");
  if (!ctx->s_out && hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking) != hipSuccess) return fail(ctx, MD_E_HIP, "hipStreamCreate");
  EventList evs;
  // From the first enqueue on there is NO early return: an error is recorded, the loop is left, and the three streams are
  // synchronised before the call returns - the copies read and write the caller's h_in / h_out, and the context's device
  // blobs may be grown (freed) by the next call.
  int rc = MD_OK;
  hipError_t herr = hipSuccess;
  const char *hwhat = nullptr;
#define MD_PIPE_TRY(expr)                       \
  if (rc == MD_OK && herr == hipSuccess) {      \
    const hipError_t e_ = (expr);               \
    if (e_ != hipSuccess) {                     \
      herr = e_;                                \
      hwhat = #expr;                            \
    }                                           \
  }
  // (whatever the caller queued on the context's stream before this call comes first, also for the copy streams)
  hipEvent_t e0 = evs.make();
  if (!e0) return fail(ctx, MD_E_HIP, "hipEventCreate");  // (nothing enqueued yet)
  MD_PIPE_TRY(hipEventRecord(e0, ctx->stream));
  MD_PIPE_TRY(hipStreamWaitEvent(ctx->s_in, e0, 0));
  for (size_t k = 0; k < sl.size() && rc == MD_OK && herr == hipSuccess; k++) {
    hipEvent_t e_in = evs.make(), e_k = evs.make();
    if (!e_in || !e_k) {
      herr = hipErrorOutOfMemory;
      hwhat = "hipEventCreate";
      break;
    }
    if (sl[k].in_hi > sl[k].in_lo)
      MD_PIPE_TRY(hipMemcpyAsync(d_in + sl[k].in_lo, h_in + sl[k].in_lo, sl[k].in_hi - sl[k].in_lo, hipMemcpyHostToDevice, ctx->s_in));
    MD_PIPE_TRY(hipEventRecord(e_in, ctx->s_in));
    MD_PIPE_TRY(hipStreamWaitEvent(ctx->stream, e_in, 0));
    if (herr != hipSuccess) break;
    rc = launch(sl[k].i0, sl[k].i1 - sl[k].i0);
    if (rc != MD_OK) break;
    MD_PIPE_TRY(hipEventRecord(e_k, ctx->stream));
    MD_PIPE_TRY(hipStreamWaitEvent(ctx->s_out, e_k, 0));
    if (sl[k].out_hi > sl[k].out_lo)
      MD_PIPE_TRY(hipMemcpyAsync(h_out + sl[k].out_lo, d_out + sl[k].out_lo, sl[k].out_hi - sl[k].out_lo, hipMemcpyDeviceToHost, ctx->s_out));
  }
#undef MD_PIPE_TRY
  // everything in flight ends before the call returns (also after an error: the buffers are the caller's)
  hipError_t a = hipStreamSynchronize(ctx->s_in), b2 = hipStreamSynchronize(ctx->stream), c = hipStreamSynchronize(ctx->s_out);
  if (rc != MD_OK) return rc;
  if (herr != hipSuccess) return fail(ctx, MD_E_HIP, hwhat, herr);
  if (a != hipSuccess || b2 != hipSuccess || c != hipSuccess) return fail(ctx, MD_E_HIP, "host pipeline", a != hipSuccess ? a : b2 != hipSuccess ? b2 : c);
  return MD_OK;
}
}  // extern "C++"

static int inflate_parallel(md_ctx *ctx, int format, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                            size_t *consumed, size_t *written, uint32_t *checksum);
int md_inflate_batch_host(md_ctx *ctx, int format, size_t n, const uint8_t *h_in, size_t in_bytes,
                          const uint64_t *in_off, const uint64_t *in_len, uint8_t *h_out,
                          size_t out_bytes, const uint64_t *out_off, const uint64_t *out_cap,
                          uint64_t *out_len, uint64_t *consumed, int32_t *status,
                          uint32_t *checksum) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  if (n == 0) return MD_OK;
  if (!in_off || !in_len || !out_off || !out_cap || !out_len || !consumed || !status)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "null descriptor array");
  for (size_t i = 0; i < n; i++) {
    if (in_off[i] > in_bytes || in_len[i] > in_bytes - in_off[i])
      return fail(ctx, MD_E_INVALID_ARGUMENT, "input range out of bounds");  // invalid_bounds, lib/de.ml:146
    if (in_len[i] > MD_MAX_INFLATE_IN) return fail(ctx, MD_E_INVALID_ARGUMENT, "stream longer than MD_MAX_INFLATE_IN");
    if (out_off[i] > out_bytes || out_cap[i] > out_bytes - out_off[i])
      return fail(ctx, MD_E_INVALID_ARGUMENT, "output range out of bounds");
  }
  MD_ON_DEVICE(ctx);
  ctx->par_last_pieces = ctx->par_last_rounds = 0;
  // A FEW LONG streams (a handful of big files): each of them in pieces, by the whole chip (inflate_parallel) - as streams
  // of a batch they would get one pair of wavefronts each.  What that path does not take, and the short streams beside
  // them, go through the batch as before.
  if (ctx->par_min && n <= 64) {
    std::vector<size_t> shorts, longs;
    // (each long stream is a call of ~1 ms at least, one after the other, where the batch kernel takes all n at once at ~0.2
    // GiB/s each: worth it from ~128 KiB of input per stream of the batch)
    const uint64_t long_from = ctx->par_min > n * ((uint64_t)128 << 10) ? ctx->par_min : n * ((uint64_t)128 << 10);
    for (size_t i = 0; i < n; i++) (in_len[i] >= long_from ? longs : shorts).push_back(i);
    if (!longs.empty()) {
      const size_t keep = ctx->par_min;
      // a batch of picked streams through this same entry point, the long-stream path switched off
      auto sub = [&](const std::vector<size_t> &pick) -> int {
        const size_t m = pick.size();
        std::vector<uint64_t> io(m), il(m), oo(m), oc(m), ol(m), cs(m);
        std::vector<int32_t> st(m);
        std::vector<uint32_t> ck(m);
        for (size_t k = 0; k < m; k++) {
          io[k] = in_off[pick[k]];
          il[k] = in_len[pick[k]];
          oo[k] = out_off[pick[k]];
          oc[k] = out_cap[pick[k]];
        }
        ctx->par_min = 0;
        const int rc = md_inflate_batch_host(ctx, format, m, h_in, in_bytes, io.data(), il.data(), h_out, out_bytes, oo.data(), oc.data(), ol.data(),
                                             cs.data(), st.data(), checksum ? ck.data() : nullptr);
        ctx->par_min = keep;
        if (rc != MD_OK) return rc;
        for (size_t k = 0; k < m; k++) {
          out_len[pick[k]] = ol[k];
          consumed[pick[k]] = cs[k];
          status[pick[k]] = st[k];
          if (checksum) checksum[pick[k]] = ck[k];
        }
        return MD_OK;
      };
      // the short ones first, as one batch (its copies take the span of the caller's blobs its streams lie in: what the long
      // streams' places receive from that is overwritten below)
      if (!shorts.empty()) {
        const int rc = sub(shorts);
        if (rc != MD_OK) return rc;
      }
      int pieces = 0, rounds = 0;
      for (size_t i : longs) {
        size_t used = 0, wrote = 0;
        uint32_t sum = 0;
        const int prc = inflate_parallel(ctx, format, h_in + in_off[i], (size_t)in_len[i], h_out + out_off[i], (size_t)out_cap[i], &used, &wrote,
                                         checksum ? &sum : nullptr);
        if (prc == 1000 /* kNotHandled: not a well-formed stream that fits - the batch path says what it is */) {
          const int rc = sub(std::vector<size_t>{i});
          if (rc != MD_OK) return rc;
          continue;
        }
        if (prc != MD_OK) return prc;
        out_len[i] = wrote;
        consumed[i] = used;
        status[i] = MD_OK;
        if (checksum) checksum[i] = sum;
        pieces += ctx->par_last_pieces;
        rounds = ctx->par_last_rounds > rounds ? ctx->par_last_rounds : rounds;
      }
      ctx->par_last_pieces = pieces;
      ctx->par_last_rounds = rounds;
      return MD_OK;
    }
  }
  const size_t desc_words = 6 * n;  // in_off in_len out_off out_cap out_len consumed
  int grc_ = grow(ctx, &ctx->host_in, &ctx->host_in_bytes, in_bytes + 64, "hipMalloc(host path input)");
  if (grc_ == MD_OK) grc_ = grow(ctx, &ctx->host_out, &ctx->host_out_bytes, out_bytes + 64, "hipMalloc(host path output)");
  if (grc_ == MD_OK) grc_ = grow(ctx, &ctx->host_desc, &ctx->host_desc_bytes, desc_words * 8 + n * 8, "hipMalloc(host path descriptors)");
  if (grc_ != MD_OK) return grc_;
  uint8_t *din = (uint8_t *)ctx->host_in, *dout = (uint8_t *)ctx->host_out;
  uint64_t *d64 = (uint64_t *)ctx->host_desc;
  int32_t *dstatus = (int32_t *)(d64 + desc_words);
  uint32_t *dsum = (uint32_t *)(dstatus + n);
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(d64 + 0 * n, in_off, n * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64 + 1 * n, in_len, n * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64 + 2 * n, out_off, n * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64 + 3 * n, out_cap, n * 8, hipMemcpyHostToDevice, st));
  // a slice of 1 024 streams (4 per CU) runs at half the batch's rate per stream - still several times what the link to
  // the host moves (57 GB/s each way measured), so the copies stay the longer leg and more slices hide more of the kernels
  const std::vector<HostSlice> sl = host_slices(n, in_off, in_len, out_off, out_cap, 1024, (size_t)ctx->host_slices_max, in_bytes, out_bytes);
  int rc = host_pipeline(ctx, sl, h_in, din, h_out, dout, [&](size_t i0, size_t cnt) {
    return md_inflate_batch_device(ctx, format, cnt, din, d64 + i0, d64 + n + i0, dout, d64 + 2 * n + i0, d64 + 3 * n + i0,
                                   d64 + 4 * n + i0, d64 + 5 * n + i0, dstatus + i0, dsum + i0);
  });
  if (rc != MD_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out_len, d64 + 4 * n, n * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(consumed, d64 + 5 * n, n * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(status, dstatus, n * 4, hipMemcpyDeviceToHost, st));
  if (checksum) HIP_TRY(ctx, hipMemcpyAsync(checksum, dsum, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return MD_OK;
}

// Pieces of n streams at once (mdeflate.h): the inflate kernel with its continuation arguments, descriptors in HBM.
int md_inflate_continue_batch_device(md_ctx *ctx, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                                     const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                     const uint64_t *d_out_cap, const uint32_t *d_start_bit, const uint32_t *d_hist_len,
                                     const uint32_t *d_adler_in, uint64_t *d_out_len, uint64_t *d_consumed,
                                     int32_t *d_status, uint32_t *d_checksum, uint64_t *d_resume_bits,
                                     uint64_t *d_resume_out, uint32_t *d_resume_adler, uint32_t *d_resume_last) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  if (n == 0) return MD_OK;
  if (n > 0xffffffffull) return fail(ctx, MD_E_INVALID_ARGUMENT, "too many streams");
  if (!d_in || !d_in_off || !d_in_len || !d_out || !d_out_off || !d_out_cap || !d_start_bit || !d_hist_len || !d_adler_in ||
      !d_out_len || !d_consumed || !d_status || !d_resume_bits || !d_resume_out || !d_resume_adler || !d_resume_last)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "null device pointer");
  MD_ON_DEVICE(ctx);
  uint32_t *order = nullptr;
  if (n >= kOrderFrom) {
    const int orc = order_scratch(ctx, n);
    if (orc != MD_OK) return orc;
    order = ctx->order;
  }
  const void *cont[7] = {d_start_bit, d_hist_len, d_adler_in, d_resume_bits, d_resume_out, d_resume_adler, d_resume_last};
  int rc = md_launch_inflate_wave(MD_FORMAT_DEFLATE, (uint32_t)n, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len,
                                  d_consumed, d_status, d_checksum, ctx->dbg, order, ctx->inflate_waves, cont, ctx->stream);
  if (rc != 0) return fail(ctx, MD_E_HIP, "inflate kernel launch", (hipError_t)rc);
  return MD_OK;
}

// One piece of a raw DEFLATE stream that is decoded as it arrives (mdeflate.h): the inflate kernel on one stream with
// a starting bit, the window in front of the output buffer and the checksum state handed in, and the last block
// boundary inside the piece handed back.
static int continue_parallel(md_ctx *ctx, const uint8_t *src, size_t src_len, unsigned start_bit, uint8_t *dst, size_t hist_len,
                             size_t dst_cap, uint32_t adler_in, unsigned flags, size_t *dst_len, int *status, md_inf_resume *resume);
static int continue_serial(md_ctx *ctx, const uint8_t *src, size_t src_len, unsigned start_bit, uint8_t *dst, size_t hist_len,
                           size_t dst_cap, uint32_t adler_in, unsigned flags, size_t *dst_len, int *status, md_inf_resume *resume);
int md_de_inf_continue_host(md_ctx *ctx, const uint8_t *src, size_t src_len, unsigned start_bit, uint8_t *dst, size_t hist_len,
                            size_t dst_cap, uint32_t adler_in, unsigned flags, size_t *dst_len, int *status,
                            md_inf_resume *resume) {
  if (!ctx || (!src && src_len) || !dst || !dst_len || !status || !resume) return MD_E_INVALID_ARGUMENT;
  if (start_bit > 7 || hist_len > 32768 || hist_len > dst_cap || dst_cap > MD_MAX_STREAM || src_len > MD_MAX_INFLATE_IN)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "md_de_inf_continue_host: start_bit <= 7, hist_len <= 32768 and <= dst_cap");
  MD_ON_DEVICE(ctx);
  ctx->par_last_pieces = ctx->par_last_rounds = 0;
  if (ctx->par_min && src_len >= ctx->par_min) {  // a long piece: its complete blocks by the whole chip, the rest as before
    const int prc = continue_parallel(ctx, src, src_len, start_bit, dst, hist_len, dst_cap, adler_in, flags, dst_len, status, resume);
    if (prc != 1000 /* kNotHandled */) return prc;
    ctx->par_last_pieces = 0;
  }
  return continue_serial(ctx, src, src_len, start_bit, dst, hist_len, dst_cap, adler_in, flags, dst_len, status, resume);
}
static int continue_serial(md_ctx *ctx, const uint8_t *src, size_t src_len, unsigned start_bit, uint8_t *dst, size_t hist_len,
                           size_t dst_cap, uint32_t adler_in, unsigned flags, size_t *dst_len, int *status, md_inf_resume *resume) {
  // the context's scratch, grow-only: a long stream comes in many pieces, and three hipMalloc / hipFree per piece
  // cost more than a short piece's kernel
  // descriptors: in_off in_len out_off out_cap out_len consumed resume_bits resume_out (u64); status, checksum,
  // start_bit, hist_len, adler_in, resume_adler, resume_last (u32)
  struct { void *p; } din, dout, ddesc;
  int grc_ = grow(ctx, &ctx->cont_in, &ctx->cont_in_bytes, src_len + 16, "hipMalloc(decoder piece input)");
  if (grc_ == MD_OK) grc_ = grow(ctx, &ctx->cont_out, &ctx->cont_out_bytes, dst_cap + 16, "hipMalloc(decoder piece output)");
  if (grc_ == MD_OK) grc_ = grow(ctx, &ctx->cont_desc, &ctx->cont_desc_bytes, 8 * 8 + 8 * 4 + 4 * 8 + 2 * 4, "hipMalloc(decoder piece descriptors)");
  if (grc_ != MD_OK) return grc_;
  din.p = ctx->cont_in;
  dout.p = ctx->cont_out;
  ddesc.p = ctx->cont_desc;
  uint64_t h64[8] = {0, (uint64_t)src_len, 0, (uint64_t)dst_cap, 0, 0, 0, 0};
  uint32_t h32[7] = {0, 0, start_bit, (uint32_t)hist_len, adler_in, 0, 0};
  uint64_t *d64 = (uint64_t *)ddesc.p;
  uint32_t *d32 = (uint32_t *)(d64 + 8);
  hipStream_t st = ctx->stream;
  if (src_len) HIP_TRY(ctx, hipMemcpyAsync(din.p, src, src_len, hipMemcpyHostToDevice, st));
  if (hist_len) HIP_TRY(ctx, hipMemcpyAsync(dout.p, dst, hist_len, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64, h64, sizeof h64, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d32, h32, sizeof h32, hipMemcpyHostToDevice, st));
  const void *cont[7] = {d32 + 2, d32 + 3, d32 + 4, d64 + 6, d64 + 7, d32 + 5, d32 + 6};
  int rc = md_launch_inflate_wave(MD_FORMAT_DEFLATE, 1, (const uint8_t *)din.p, d64 + 0, d64 + 1, (uint8_t *)dout.p, d64 + 2,
                                  d64 + 3, d64 + 4, d64 + 5, (int32_t *)d32, d32 + 1, nullptr, nullptr, ctx->inflate_waves,
                                  cont, st);
  if (rc != 0) return fail(ctx, MD_E_HIP, "inflate kernel launch", (hipError_t)rc);
  HIP_TRY(ctx, hipMemcpyAsync(h64, d64, sizeof h64, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(h32, d32, sizeof h32, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  const size_t produced = (size_t)h64[4];
  if (produced > hist_len) HIP_TRY(ctx, hipMemcpy(dst + hist_len, (const uint8_t *)dout.p + hist_len, produced - hist_len, hipMemcpyDeviceToHost));
  *dst_len = produced;
  *status = (int32_t)h32[0];
  resume->bits = h64[6];
  resume->out = h64[7];
  resume->adler = h32[5];
  resume->last = h32[6];
  resume->consumed = h64[5];
  resume->checksum = h32[1];
  resume->crc_out = resume->crc_end = 0;
  if (flags & MD_CONT_CRC32) {  // CRC-32 of the new output up to the block boundary, and up to where decoding got
    uint64_t *c64 = (uint64_t *)(d32 + 8);
    uint32_t *c32 = (uint32_t *)(c64 + 4);
    const uint64_t to_out = resume->out > hist_len ? resume->out - hist_len : 0, to_end = produced > hist_len ? produced - hist_len : 0;
    const uint64_t hc[4] = {(uint64_t)hist_len, (uint64_t)hist_len, to_out, to_end};
    uint32_t crc[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(c64, hc, sizeof hc, hipMemcpyHostToDevice, st));
    int e = md_launch_crc32(2, (const uint8_t *)dout.p, c64, c64 + 2, c32, st);
    if (e != 0) return fail(ctx, MD_E_HIP, "crc32 kernel launch", (hipError_t)e);
    HIP_TRY(ctx, hipMemcpyAsync(crc, c32, sizeof crc, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    resume->crc_out = crc[0];
    resume->crc_end = crc[1];
  }
  return MD_OK;
}

// host-side CRC-32 of the few header bytes (the CRC16 of lib/gz.ml:771-789 is its upper half)
static uint32_t host_crc32(const uint8_t *p, size_t n) {
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; i++) {
    c ^= p[i];
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1)));
  }
  return c ^ 0xffffffffu;
}
// The bytes Gz.Def writes in front of the body (lib/gz.ml:796-812) for the header fields of Gz.Def.encoder
// (lib/gz.ml:859-918); returns the length, 0 when a field is out of range.
static uint32_t gz_header_bytes(const md_gz_header *g, int level, uint8_t h[544]) {
  static const md_gz_header dflt = {0, 3, 0, 0, nullptr, nullptr};
  if (!g) g = &dflt;
  const size_t nl = g->filename ? strlen(g->filename) : 0, cl = g->comment ? strlen(g->comment) : 0;
  if (nl > 255 || cl > 255 || g->os < 0 || g->os > 255) return 0;
  memset(h, 0, 544);
  // flg, lib/gz.ml:851-857; mtime big-endian, lib/gz.ml:801
  h[0] = 0x1f;
  h[1] = 0x8b;
  h[2] = 8;
  h[3] = (uint8_t)((g->ascii ? 1 : 0) | (g->hcrc ? 2 : 0) | (g->filename ? 8 : 0) | (g->comment ? 16 : 0));
  h[4] = (uint8_t)(g->mtime >> 24);
  h[5] = (uint8_t)(g->mtime >> 16);
  h[6] = (uint8_t)(g->mtime >> 8);
  h[7] = (uint8_t)g->mtime;
  h[8] = level == 9 ? 2 : 0;  // xfl, lib/gz.ml:888-890
  h[9] = (uint8_t)g->os;
  uint32_t p = 10;
  if (g->filename) {
    memcpy(h + p, g->filename, nl + 1);
    p += (uint32_t)nl + 1;
  }
  if (g->comment) {
    memcpy(h + p, g->comment, cl + 1);
    p += (uint32_t)cl + 1;
  }
  if (g->hcrc) {  // the upper half of the CRC-32 of what precedes, big-endian (H10, lib/gz.ml:771-789)
    const uint32_t c16 = (host_crc32(h, p) & 0xffff0000u) >> 16;
    h[p] = (uint8_t)(c16 >> 8);
    h[p + 1] = (uint8_t)c16;
    p += 2;
  }
  return p;
}

int md_crc32_batch_device(md_ctx *ctx, size_t n, const uint8_t *d_data, const uint64_t *d_off,
                          const uint64_t *d_len, uint32_t *d_crc) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  if (n == 0) return MD_OK;
  if (n > 0x7fffffffull || !d_off || !d_len || !d_crc) return fail(ctx, MD_E_INVALID_ARGUMENT, "bad crc32 batch");
  MD_ON_DEVICE(ctx);
  int e = md_launch_crc32((uint32_t)n, d_data, d_off, d_len, d_crc, ctx->stream);
  if (e != 0) return fail(ctx, MD_E_HIP, "crc32 kernel launch", (hipError_t)e);
  return MD_OK;
}

// a grow-only device buffer of the context
static int grow(md_ctx *ctx, void **buf, size_t *have, size_t need, const char *what) {
  if (need <= *have) return MD_OK;
  if (*buf) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
  }
  if (hipMalloc(buf, need) != hipSuccess) return fail(ctx, MD_E_OUT_OF_MEMORY, what);
  *have = need;
  return MD_OK;
}

static const size_t kPieceStateBytes = md_piece_state_bytes();  // deflate_common.hpp kPieceState (checked against sizeof there)
// One piece of one stream (md_i_piece_run below): device pointers of what differs from a batch of whole streams.
struct PieceArgs {
  const uint64_t *d_front_len;  // length of the text the launch holds, n - w0 (d_in_len is the absolute length n)
  void *queue;                  // the stream's own command queue: it lives across launches
  const void *ptrs[4];          // struct Piece of deflate_common.hpp: flags, state, pos, sum
  uint32_t match_skip;          // leading positions of the text no stream of the launch will take (the window brought along)
};

// total_in: an upper bound of the sum of in_len when the caller knows one (md_deflate_params.total_in_bytes), else 0:
// the front workspace is then sized from the totals the plan kernel computes, which costs one 16-byte read-back
// (a synchronisation with the context's stream).
static int deflate_launch(md_ctx *ctx, int format, int level, int queue_len, int driver, int dynamic, int matcher,
                          const md_gz_header *gz, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                          const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off, const uint64_t *d_out_cap,
                          uint64_t *d_out_len, int32_t *d_status, uint32_t *d_checksum, uint32_t *d_hist, size_t total_in,
                          const PieceArgs *pa = nullptr) {
  int grc_ = MD_OK;
  // (the Lz77-alone / encode-alone / scripted drivers leave the kernel before it saves a piece's state)
  if (pa && driver >= 3) return fail(ctx, MD_E_INVALID_ARGUMENT, "a stream in pieces needs one of the three public drivers");
  if (!pa) grc_ = grow(ctx, &ctx->ws, &ctx->ws_bytes, md_deflate_queue_bytes((uint32_t)n, queue_len), "hipMalloc(deflate command queues)");
  if (grc_ != MD_OK) return grc_;
  const uint64_t *d_front_len = pa ? pa->d_front_len : d_in_len;  // (a piece: the front kernels see [w0, n) as a stream)
  grc_ = grow(ctx, &ctx->fsmall, &ctx->fsmall_bytes, md_front_small_bytes((uint32_t)n), "hipMalloc(deflate plan)");
  if (grc_ != MD_OK) return grc_;
  md_front fr;
  uint64_t positions = 0;
  uint32_t chunks = 0;
  uint32_t max_chain = 0, nice = 0;
  md_deflate_level_params(driver, matcher, level, &max_chain, &nice);
  const bool matcher_runs = max_chain != 0 && driver < 4;  // level 0 copies; De.Def.encode has no text
  if (matcher_runs && total_in != 0) {
    // slot <= len + 64 + (chunk - 1) positions and <= len / chunk + 2 chunks per stream
    positions = (uint64_t)total_in + kSlotPad * n;
    const uint64_t c64 = (uint64_t)total_in / kChunkPositions + 2ull * n;
    if (c64 > 0x7fffffffull) return fail(ctx, MD_E_INVALID_ARGUMENT, "batch too large for one launch");
    chunks = (uint32_t)c64;
    grc_ = grow(ctx, &ctx->fbig, &ctx->fbig_bytes, md_front_big_bytes(positions), "hipMalloc(deflate front workspace)");
    if (grc_ != MD_OK) return grc_;
  }
  md_front_carve(ctx->fsmall, ctx->fbig, (uint32_t)n, positions, &fr);
  int prc = md_launch_deflate_plan((uint32_t)n, d_front_len, driver, matcher, level, positions, chunks, &fr, ctx->stream);
  if (prc != 0) return fail(ctx, MD_E_HIP, "deflate plan kernel launch", (hipError_t)prc);
  if (matcher_runs && total_in == 0) {
    uint64_t tot_pos = 0;
    uint32_t tot_chunks = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&tot_pos, (const uint64_t *)fr.slot + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&tot_chunks, (const uint32_t *)fr.chunk0 + n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    positions = tot_pos;
    chunks = tot_chunks;
    grc_ = grow(ctx, &ctx->fbig, &ctx->fbig_bytes, md_front_big_bytes(positions), "hipMalloc(deflate front workspace)");
    if (grc_ != MD_OK) return grc_;
    md_front_carve(ctx->fsmall, ctx->fbig, (uint32_t)n, positions, &fr);
  }
  const uint8_t *gz_hdr = nullptr;
  uint32_t *gz_crc = nullptr;
  uint32_t gz_hdr_len = 0;
  if (format == MD_FORMAT_GZIP) {
    uint8_t h[544];
    gz_hdr_len = gz_header_bytes(gz, level, h);
    if (!gz_hdr_len) return fail(ctx, MD_E_INVALID_ARGUMENT, "gzip header field out of range");
    int grc = gz_scratch(ctx, n);
    if (grc != MD_OK) return grc;
    gz_crc = (uint32_t *)((uint8_t *)ctx->gz_tmp + n * 20);
    if (!ctx->gz_hdr_dev && hipMalloc((void **)&ctx->gz_hdr_dev, sizeof h) != hipSuccess)
      return fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc(gzip header)");
    if (!ctx->gz_hdr_valid || memcmp(ctx->gz_hdr_sent, h, sizeof h) != 0) {
      // pageable source: the runtime stages the bytes before the call returns, the copy itself is stream-ordered
      HIP_TRY(ctx, hipMemcpyAsync(ctx->gz_hdr_dev, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
      memcpy(ctx->gz_hdr_sent, h, sizeof h);
      ctx->gz_hdr_valid = true;
    }
    gz_hdr = ctx->gz_hdr_dev;
    if (!pa) {  // (in pieces the CRC-32 is the caller's running one)
      int e = md_launch_crc32((uint32_t)n, d_in, d_in_off, d_in_len, gz_crc, ctx->stream);
      if (e != 0) return fail(ctx, MD_E_HIP, "crc32 kernel launch", (hipError_t)e);
    }
  }
  // more streams than the link kernel (one per CU) or the sequential kernel (16 per CU) hold at once: longest first
  uint32_t *order = nullptr;
  if (n >= kOrderFromDeflate) {
    const int orc = order_scratch(ctx, n);
    if (orc != MD_OK) return orc;
    order = ctx->order;
    // (a slice of a batch: by what the slice brings, not by the absolute length so far - idle streams bring nothing)
    int oe = md_launch_stream_order((uint32_t)n, d_front_len, order, ctx->stream);
    if (oe != 0) return fail(ctx, MD_E_HIP, "launch order kernel", (hipError_t)oe);
  }
  if (matcher_runs && chunks != 0) {
    int frc = md_launch_deflate_front((uint32_t)n, chunks, d_in, d_in_off, d_front_len, matcher, max_chain, nice, &fr, order, pa ? pa->match_skip : 0u, ctx->stream);
    if (frc != 0) return fail(ctx, MD_E_HIP, "deflate front kernel launch", (hipError_t)frc);
  }
  int rc = md_launch_deflate(format, level, queue_len, driver, dynamic, (uint32_t)n, d_in, d_in_off, d_in_len, d_out,
                             d_out_off, d_out_cap, d_out_len, d_status, d_checksum, &fr, pa ? pa->queue : ctx->ws, ctx->dbg,
                             gz_hdr, gz_hdr_len, gz_crc, matcher, d_hist, order, pa ? pa->ptrs : nullptr, ctx->stream);
  if (rc != 0) return fail(ctx, MD_E_HIP, "deflate kernel launch", (hipError_t)rc);
  return MD_OK;
}

// De.Def.Ns / Zl.Def.Ns: the front workspace as for deflate_launch (no command queues), then the three kernels of
// deflate_ns.hip.  total_in as in deflate_launch.
static int def_ns_launch(md_ctx *ctx, int format, int level, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                         const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off, const uint64_t *d_out_cap,
                         uint64_t *d_out_len, int32_t *d_status, uint32_t *d_checksum, size_t total_in) {
  int grc_ = grow(ctx, &ctx->fsmall, &ctx->fsmall_bytes, md_front_small_bytes((uint32_t)n), "hipMalloc(deflate plan)");
  if (grc_ != MD_OK) return grc_;
  md_front fr;
  uint64_t positions = 0;
  uint32_t chunks = 0;
  const bool matcher_runs = level >= 1 && level <= 4;
  if (matcher_runs && total_in != 0) {
    positions = (uint64_t)total_in + kSlotPad * n;
    const uint64_t c64 = (uint64_t)total_in / kChunkPositions + 2ull * n;
    if (c64 > 0x7fffffffull) return fail(ctx, MD_E_INVALID_ARGUMENT, "batch too large for one launch");
    chunks = (uint32_t)c64;
    grc_ = grow(ctx, &ctx->fbig, &ctx->fbig_bytes, md_front_big_bytes(positions), "hipMalloc(deflate front workspace)");
    if (grc_ != MD_OK) return grc_;
  }
  md_front_carve(ctx->fsmall, ctx->fbig, (uint32_t)n, positions, &fr);
  int prc = md_launch_deflate_plan((uint32_t)n, d_in_len, 6, MD_MATCHER_DE, level, positions, chunks, &fr, ctx->stream);
  if (prc != 0) return fail(ctx, MD_E_HIP, "deflate plan kernel launch", (hipError_t)prc);
  if (matcher_runs && total_in == 0) {
    uint64_t tot_pos = 0;
    uint32_t tot_chunks = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&tot_pos, (const uint64_t *)fr.slot + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&tot_chunks, (const uint32_t *)fr.chunk0 + n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    positions = tot_pos;
    chunks = tot_chunks;
    grc_ = grow(ctx, &ctx->fbig, &ctx->fbig_bytes, md_front_big_bytes(positions), "hipMalloc(deflate front workspace)");
    if (grc_ != MD_OK) return grc_;
    md_front_carve(ctx->fsmall, ctx->fbig, (uint32_t)n, positions, &fr);
  }
  int rc = md_launch_def_ns(format, level, (uint32_t)n, chunks, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len,
                            d_status, d_checksum, &fr, ctx->stream);
  if (rc != 0) return fail(ctx, MD_E_HIP, "Def.Ns kernel launch", (hipError_t)rc);
  return MD_OK;
}

int md_def_ns_batch_device(md_ctx *ctx, int format, int level, size_t total_in_bytes, size_t n, const uint8_t *d_in,
                           const uint64_t *d_in_off, const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                           const uint64_t *d_out_cap, uint64_t *d_out_len, int32_t *d_status, uint32_t *d_checksum) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  if (format != MD_FORMAT_DEFLATE && format != MD_FORMAT_ZLIB) return fail(ctx, MD_E_INVALID_ARGUMENT, "unknown format");
  if (level < 0 || level > 12) return fail(ctx, MD_E_INVALID_ARGUMENT, "Invalid compression level");  // lib/de.ml:3930
  if (n == 0) return MD_OK;
  if (n > 0x7fffffffull) return fail(ctx, MD_E_INVALID_ARGUMENT, "too many streams in one batch");
  if (!d_in_off || !d_in_len || !d_out_off || !d_out_cap || !d_out_len || !d_status)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "null descriptor array");
  MD_ON_DEVICE(ctx);
  return def_ns_launch(ctx, format, level, n, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len, d_status,
                       d_checksum, total_in_bytes);
}

static int def_ns_one(md_ctx *ctx, int format, int level, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                      size_t *written) {
  if (!ctx || !written || (!src && src_len) || (!dst && dst_cap)) return MD_E_INVALID_ARGUMENT;
  if (level < 0 || level > 12) return fail(ctx, MD_E_INVALID_ARGUMENT, "Invalid compression level");
  if (src_len > MD_MAX_STREAM) return fail(ctx, MD_E_INVALID_ARGUMENT, "stream longer than MD_MAX_STREAM");
  MD_ON_DEVICE(ctx);
  DevBuf din, dout, ddesc;
  if (din.alloc(src_len + 16) != hipSuccess || dout.alloc(dst_cap + 16) != hipSuccess || ddesc.alloc(6 * 8 + 16) != hipSuccess)
    return fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc");
  uint64_t desc[5] = {0, src_len, 0, dst_cap, 0};
  uint64_t *d64 = (uint64_t *)ddesc.p;
  int32_t *dstatus = (int32_t *)(d64 + 5);
  hipStream_t st = ctx->stream;
  if (src_len) HIP_TRY(ctx, hipMemcpyAsync(din.p, src, src_len, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64, desc, sizeof desc, hipMemcpyHostToDevice, st));
  int rc = def_ns_launch(ctx, format, level, 1, (const uint8_t *)din.p, d64, d64 + 1, (uint8_t *)dout.p, d64 + 2, d64 + 3, d64 + 4,
                         dstatus, nullptr, src_len ? src_len : 1);
  if (rc != MD_OK) return rc;
  uint64_t out_len = 0;
  int32_t status = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&out_len, d64 + 4, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(&status, dstatus, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  if (status == MD_OK && out_len) HIP_TRY(ctx, hipMemcpy(dst, dout.p, (size_t)out_len, hipMemcpyDeviceToHost));
  *written = (size_t)out_len;
  return status;
}
int md_de_def_ns_deflate(md_ctx *ctx, int level, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written) {
  return def_ns_one(ctx, MD_FORMAT_DEFLATE, level, src, src_len, dst, dst_cap, written);
}
int md_zl_def_ns_deflate(md_ctx *ctx, int level, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written) {
  return def_ns_one(ctx, MD_FORMAT_ZLIB, level, src, src_len, dst, dst_cap, written);
}
size_t md_de_def_ns_compress_bound(size_t len) {  // lib/de.ml:3994-3997
  size_t max_blocks = (len + 10000 - 1) / 10000;
  if (max_blocks < 1) max_blocks = 1;
  return 5 * max_blocks + len + 1 + 8;
}
size_t md_zl_def_ns_compress_bound(size_t len) { return md_de_def_ns_compress_bound(len) + 6; }  // lib/zl.ml:600

static int check_params(md_ctx *ctx, int format, const md_deflate_params *p, md_deflate_params *q) {
  if (!p) return fail(ctx, MD_E_INVALID_ARGUMENT, "null md_deflate_params");
  *q = *p;
  if (format != MD_FORMAT_DEFLATE && format != MD_FORMAT_ZLIB && format != MD_FORMAT_GZIP)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "unknown format");
  if (format == MD_FORMAT_GZIP) {  // Gz.Def's driver is Zl's with block_of_frequencies (lib/gz.ml:724-729)
    q->driver = MD_DRIVER_ZL;
    q->dynamic = 1;
  }
  if (q->level < 0 || q->level > 9)  // Lz77.state: "Invalid level of compression", lib/de.ml:4477
    return fail(ctx, MD_E_INVALID_ARGUMENT, "Invalid level of compression");
  if (q->queue_len < 4 || q->queue_len > (1 << 20) || (q->queue_len & (q->queue_len - 1)))  // lib/de.ml:2286-2288
    return fail(ctx, MD_E_INVALID_ARGUMENT, "Length of queue MUST be a power of two");
  if (q->driver < MD_DRIVER_ZL || q->driver > MD_DRIVER_CLI) return fail(ctx, MD_E_INVALID_ARGUMENT, "unknown driver");
  if (q->matcher != MD_MATCHER_DE && q->matcher != MD_MATCHER_LZ) return fail(ctx, MD_E_INVALID_ARGUMENT, "unknown matcher");
  q->dynamic = q->dynamic ? 1 : 0;
  if (q->wbits != 0 && q->wbits != 15)  // De.Lz77.state ~w: only make_window ~bits:15 (lib/de.ml:4462-4464, :4513)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "only 32 KiB windows (wbits 15) are implemented");
  return MD_OK;
}

// what md_def_encoder checks before it keeps the parameters (stream_shim.cpp); not part of the public header
int md_validate_deflate_params(md_ctx *ctx, int format, const md_deflate_params *params) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  md_deflate_params q;
  return check_params(ctx, format, params, &q);
}

// ---- a batch in slices of positions ----------------------------------------------------------------------------------
// The per-position workspace is 13 bytes per input byte of what ONE launch covers.  With a cap set (md_set_option
// "deflate_workspace_cap_mib") a batch that would need more goes through the kernels S positions of every stream at a
// time: a launch covers [k*S - kSliceKeep, (k+1)*S) of each stream that reaches that far and goes on from the state
// the launch before left (the machinery of the encoder in pieces, md_i_piece_run below).  S is a multiple of 32 KiB:
// every fill of De.Lz77's window ends on such a boundary (lib/de.ml:4294-4342: more = 2 * wsize - lookahead - strstart
// after a slide tops the window up, and the window's base moves 32 KiB at a time), so no fill ever finds less than it
// would with the whole stream at hand and the bytes out are the same.  kSliceKeep: what a launch sees again of the
// slice before - the matcher stopped less than 262 short of its end and reaches back 32 KiB - 262 from there.
static const uint64_t kSliceKeep = 33792;
static const uint64_t kSliceMin = 65536;

static uint64_t slice_positions(const uint64_t *len, size_t n, uint64_t S) {  // most text one launch covers
  uint64_t first = 0, second = 0;
  for (size_t i = 0; i < n; i++) {
    if (len[i] > MD_MAX_STREAM) continue;
    first += len[i] < S ? len[i] : S;
    if (len[i] > S) second += (len[i] - S < S ? len[i] - S : S) + kSliceKeep;
  }
  return (first > second ? first : second) + kSlotPad * n;
}

// md_deflate_batch_host feeds the slices from host memory and takes finished output away under them: before_slice(k) is
// called in front of the launches of slice k (it makes the context's stream wait for that slice's input and starts the
// copy of the next one), after_slice(k, fin) behind them, once the stream has been waited for, with fin[i] = the output
// bytes of stream i that are final
struct SliceHooks {
  std::function<int(uint64_t)> before_slice;  // in front of the launches of slice k
  std::function<int(uint64_t)> launched;      // right behind them: the place to enqueue copies that should run under them
  std::function<int(uint64_t, const std::vector<uint64_t> &)> after_slice;
};
static int deflate_in_slices(md_ctx *ctx, int format, const md_deflate_params &q, size_t n, uint64_t S, const uint8_t *d_in,
                             const uint64_t *h_in_off, const uint64_t *h_in_len, uint8_t *d_out, const uint64_t *h_out_off,
                             const uint64_t *h_out_cap, uint64_t *h_out_len, int32_t *h_status, uint32_t *h_checksum,
                             const uint32_t *h_crc, const SliceHooks *hooks = nullptr) {
  // state slots for the streams that do not end in the first slice
  std::vector<uint64_t> slot(n, 0), used(n, 0);
  std::vector<uint8_t> done(n, 0);
  size_t n_long = 0;
  uint64_t longest = 0;
  for (size_t i = 0; i < n; i++) {
    if (h_in_len[i] > MD_MAX_STREAM) {  // 32-bit cursors (mdeflate.h): refused as the kernel refuses it in a whole batch
      done[i] = 1;
      h_status[i] = MD_E_INVALID_ARGUMENT;
      h_out_len[i] = 0;
      h_checksum[i] = 0;
      continue;
    }
    if (h_in_len[i] > S) slot[i] = n_long++;
    if (h_in_len[i] > longest) longest = h_in_len[i];
  }
  int rc = grow(ctx, &ctx->slice_state, &ctx->slice_state_bytes, (n_long ? n_long : 1) * kPieceStateBytes, "hipMalloc(deflate slice states)");
  if (rc != MD_OK) return rc;
  rc = grow(ctx, &ctx->ws, &ctx->ws_bytes, md_deflate_queue_bytes((uint32_t)n, q.queue_len), "hipMalloc(deflate command queues)");
  if (rc != MD_OK) return rc;
  // descriptors of a slice: ten 64-bit and six 32-bit words per stream
  const size_t desc_bytes = n * (10 * 8 + 6 * 4);
  rc = grow(ctx, &ctx->slice_desc, &ctx->slice_desc_bytes, desc_bytes, "hipMalloc(deflate slice descriptors)");
  if (rc != MD_OK) return rc;
  std::vector<uint64_t> hbuf((desc_bytes + 7) / 8);
  uint64_t *h64 = hbuf.data();
  uint64_t *in_off = h64, *front_len = h64 + n, *abs_len = h64 + 2 * n, *out_off = h64 + 3 * n, *out_cap = h64 + 4 * n,
           *out_len = h64 + 5 * n, *pos = h64 + 6 * n;
  uint32_t *h32 = (uint32_t *)(h64 + 10 * n);
  uint32_t *st = h32, *sum_out = h32 + n, *flags = h32 + 2 * n, *sums = h32 + 3 * n;  // (sums: 2 per stream)
  uint64_t *d64 = (uint64_t *)ctx->slice_desc;
  uint32_t *d32 = (uint32_t *)(d64 + 10 * n);
  const uint64_t nslices = longest ? (longest + S - 1) / S : 1;
  for (uint64_t k = 0; k < nslices; k++) {
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) {
      const uint64_t len = h_in_len[i];
      const uint64_t end = len < (k + 1) * S ? len : (k + 1) * S;
      const uint64_t w0 = k == 0 ? 0 : k * S - kSliceKeep;
      const bool idle = done[i] || (k > 0 && len <= k * S);
      in_off[i] = h_in_off[i] + (idle ? 0 : w0);
      front_len[i] = idle ? 0 : end - w0;
      abs_len[i] = idle ? 0 : end;
      out_off[i] = h_out_off[i] + used[i];
      out_cap[i] = h_out_cap[i] - used[i];
      out_len[i] = 0;
      pos[4 * i] = idle ? 0 : w0;
      pos[4 * i + 1] = 0;
      pos[4 * i + 2] = slot[i];
      pos[4 * i + 3] = i;
      st[i] = 0;
      sum_out[i] = 0;
      flags[i] = idle ? 8u : (k == 0 ? 1u : 0u) | (end == len ? 2u : 0u) | 4u;
      sums[2 * i] = h_crc ? h_crc[i] : 1u;
      sums[2 * i + 1] = (uint32_t)len;
      total += front_len[i];
    }
    if (hooks) {
      rc = hooks->before_slice(k);
      if (rc != MD_OK) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(d64, h64, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
    // (every stream of a later slice stopped less than 262 + 64 short of the slice before's end)
    PieceArgs pa{d64 + n, ctx->ws, {d32 + 2 * n, ctx->slice_state, d64 + 6 * n, d32 + 3 * n}, k == 0 ? 0u : (uint32_t)kSliceKeep - 512u};
    rc = deflate_launch(ctx, format, q.level, q.queue_len, q.driver, q.dynamic, q.matcher, q.gz_header, n, d_in, d64, d64 + 2 * n,
                        d_out, d64 + 3 * n, d64 + 4 * n, d64 + 5 * n, (int32_t *)d32, d32 + n, nullptr, total ? total : 1, &pa);
    if (rc != MD_OK) return rc;
    if (hooks) {  // (in front of the read-backs: a copy to pageable memory keeps the calling thread until the kernels are done)
      rc = hooks->launched(k);
      if (rc != MD_OK) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(out_len, d64 + 5 * n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(st, d32, 2 * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; i++) {
      if (flags[i] & 8) continue;
      if ((int32_t)st[i] == 1000) {  // MD_PIECE_AWAIT: more of the stream to come
        used[i] += out_len[i];
        continue;
      }
      done[i] = 1;
      h_status[i] = (int32_t)st[i];
      h_checksum[i] = sum_out[i];
      h_out_len[i] = (int32_t)st[i] == MD_OK ? used[i] + out_len[i] : 0;
    }
    if (hooks) {
      std::vector<uint64_t> fin(n);
      for (size_t i = 0; i < n; i++) fin[i] = done[i] ? h_out_len[i] : used[i];
      rc = hooks->after_slice(k, fin);
      if (rc != MD_OK) return rc;
    }
  }
  for (size_t i = 0; i < n; i++)
    if (!done[i]) return fail(ctx, MD_E_HIP, "deflate in slices: a stream did not end");
  return MD_OK;
}

// the batch whose workspace is above the cap: lengths to the host, slices sized to the cap (in groups of streams if a
// slice of every stream at once would still be too much), results back to the caller's device arrays
static int deflate_capped(md_ctx *ctx, int format, const md_deflate_params &q, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                          const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off, const uint64_t *d_out_cap,
                          uint64_t *d_out_len, int32_t *d_status, uint32_t *d_checksum, bool *whole, uint64_t *total_out) {
  std::vector<uint64_t> h(4 * n);
  uint64_t *in_off = h.data(), *in_len = in_off + n, *out_off = in_len + n, *out_cap = out_off + n;
  HIP_TRY(ctx, hipMemcpyAsync(in_off, d_in_off, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(in_len, d_in_len, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_off, d_out_off, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_cap, d_out_cap, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t total = 0, longest = 0;
  for (size_t i = 0; i < n; i++) {
    if (in_len[i] > MD_MAX_STREAM) in_len[i] = MD_MAX_STREAM + 1;  // (the kernel refuses it)
    else total += in_len[i];
    if (in_len[i] > longest) longest = in_len[i];
  }
  (void)longest;
  *total_out = total ? total : 1;  // what the lengths add up to: the one launch sizes its workspace from this, not from the hint
  *whole = md_front_big_bytes(total + kSlotPad * n) <= ctx->front_cap_bytes;
  if (*whole) return MD_OK;  // (fits after all: the caller's one launch)
  std::vector<uint64_t> r_len(n);
  std::vector<int32_t> r_st(n);
  std::vector<uint32_t> r_sum(n), crc;
  if (format == MD_FORMAT_GZIP) {  // the CRC-32 of every stream, once
    int grc = gz_scratch(ctx, n);
    if (grc != MD_OK) return grc;
    uint32_t *d_crc = (uint32_t *)((uint8_t *)ctx->gz_tmp + n * 20);
    int e = md_launch_crc32((uint32_t)n, d_in, d_in_off, d_in_len, d_crc, ctx->stream);
    if (e != 0) return fail(ctx, MD_E_HIP, "crc32 kernel launch", (hipError_t)e);
    crc.resize(n);
    HIP_TRY(ctx, hipMemcpyAsync(crc.data(), d_crc, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  // First in groups of consecutive streams, as long as a group still fills the device several times over (16 streams
  // per CU: 4 096 at a time - more streams than that take turns anyway; but a group lasts as long as its longest stream at
  // least, so it has to bring enough work to cover that: four turns); then, within a group, in slices of positions, each
  // group with the largest slice that fits (at least kSliceMin, else fewer streams)
  const size_t kGeneration = 4 * 4096;
  size_t groups = (size_t)((md_front_big_bytes(total + kSlotPad * n) + ctx->front_cap_bytes - 1) / ctx->front_cap_bytes);
  if (groups > n / kGeneration) groups = n / kGeneration;
  if (groups < 1) groups = 1;
  const size_t per_group = (n + groups - 1) / groups;
  for (size_t i0 = 0; i0 < n;) {
    size_t k = n - i0 < per_group ? n - i0 : per_group;
    uint64_t S = 0;
    for (;;) {
      uint64_t lo = kSliceMin / 32768, hi = 0;
      uint64_t gl = 0;
      for (size_t i = i0; i < i0 + k; i++) gl = in_len[i] <= MD_MAX_STREAM && in_len[i] > gl ? in_len[i] : gl;
      hi = (gl + 32767) / 32768;
      if (hi < lo) hi = lo;
      if (md_front_big_bytes(slice_positions(in_len + i0, k, lo * 32768)) > ctx->front_cap_bytes && k > 1) {
        k = (k + 1) / 2;  // too many streams for the smallest slice: fewer of them
        continue;
      }
      while (lo < hi) {  // the largest S (in 32 KiB units) whose launches fit
        const uint64_t mid = (lo + hi + 1) / 2;
        if (md_front_big_bytes(slice_positions(in_len + i0, k, mid * 32768)) <= ctx->front_cap_bytes) lo = mid;
        else hi = mid - 1;
      }
      S = lo * 32768;
      if (gl > S) {  // as many slices as that takes, of even size
        const uint64_t ns = (gl + S - 1) / S;
        S = ((gl + ns - 1) / ns + 32767) / 32768 * 32768;
      }
      break;
    }
    int rc = deflate_in_slices(ctx, format, q, k, S, d_in, in_off + i0, in_len + i0, d_out, out_off + i0, out_cap + i0, r_len.data() + i0,
                               r_st.data() + i0, r_sum.data() + i0, crc.empty() ? nullptr : crc.data() + i0);
    if (rc != MD_OK) return rc;
    i0 += k;
  }
  HIP_TRY(ctx, hipMemcpyAsync(d_out_len, r_len.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_status, r_st.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (d_checksum) HIP_TRY(ctx, hipMemcpyAsync(d_checksum, r_sum.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the host vectors go out of scope)
  *whole = false;
  return MD_OK;
}

int md_deflate_batch_device(md_ctx *ctx, int format, const md_deflate_params *params, size_t n, const uint8_t *d_in,
                            const uint64_t *d_in_off, const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                            const uint64_t *d_out_cap, uint64_t *d_out_len, int32_t *d_status, uint32_t *d_checksum) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  md_deflate_params q;
  int rc = check_params(ctx, format, params, &q);
  if (rc != MD_OK) return rc;
  if (n == 0) return MD_OK;
  if (n > 0x7fffffffull) return fail(ctx, MD_E_INVALID_ARGUMENT, "too many streams in one batch");
  if (!d_in_off || !d_in_len || !d_out_off || !d_out_cap || !d_out_len || !d_status)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "null descriptor array");
  MD_ON_DEVICE(ctx);
  // The per-position workspace is 13 bytes per input byte of what one launch covers (md_front_big_bytes): with a cap set
  // (md_set_option "deflate_workspace_cap_mib") a batch that would need more is taken in slices of positions - same bytes
  // out (deflate_in_slices above).  Without params->total_in_bytes the lengths have to be read back to know.
  {
    uint32_t max_chain = 0, nice = 0;
    md_deflate_level_params(q.driver, q.matcher, q.level, &max_chain, &nice);
    if (ctx->front_cap_bytes && max_chain != 0 &&
        (!q.total_in_bytes || md_front_big_bytes((uint64_t)q.total_in_bytes + kSlotPad * n) > ctx->front_cap_bytes)) {
      bool whole = true;
      uint64_t total = 0;
      rc = deflate_capped(ctx, format, q, n, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len, d_status, d_checksum, &whole, &total);
      if (rc != MD_OK || !whole) return rc;
      // the batch fits the cap after all: one launch, sized from the sum just read back (no second read-back in
      // deflate_launch, and a loose hint cannot make the workspace grow above the cap)
      q.total_in_bytes = (size_t)total;
    }
  }
  return deflate_launch(ctx, format, q.level, q.queue_len, q.driver, q.dynamic, q.matcher, q.gz_header, n, d_in, d_in_off,
                        d_in_len, d_out, d_out_off, d_out_cap, d_out_len, d_status, d_checksum, nullptr, q.total_in_bytes);
}

// ---- the encoder shim's stream in pieces (stream_shim.cpp): not part of the public ABI ------------------------------
// A launch takes the text [w0, n) of ONE stream - the last 64 KiB the launch before already saw plus what arrived since -
// and goes on from the state that launch left in device memory (the two structs of the sequential kernel, 12 KiB, and
// the stream's command queue): the matcher answers `Await at the end of the piece exactly where the reference's would
// (deflate_kernel.hip, LZ_AWAIT), so the bytes are those of the reference fed the same pieces.  Neither side keeps
// more of the stream than the window and the piece.
struct md_piece {
  void *d_text = nullptr, *d_out = nullptr, *d_state = nullptr, *d_queue = nullptr, *d_desc = nullptr;
  size_t text_cap = 0, out_cap = 0;
};
size_t md_i_piece_bytes(const md_ctx *ctx) { return ctx ? ctx->piece_bytes : 0; }
int md_i_test_flags(const md_ctx *ctx) { return ctx ? ctx->test_flags : 0; }
md_piece *md_i_piece_open(md_ctx *ctx, int queue_len) {
  if (!ctx || queue_len < 4) return nullptr;
  DeviceGuard guard(ctx->device);
  md_piece *p = new md_piece();
  if (hipMalloc(&p->d_state, kPieceStateBytes) != hipSuccess || hipMalloc(&p->d_queue, (size_t)queue_len * 4) != hipSuccess ||
      hipMalloc(&p->d_desc, 128) != hipSuccess) {
    hipFree(p->d_state);
    hipFree(p->d_queue);
    hipFree(p->d_desc);
    delete p;
    fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc(encoder state)");
    return nullptr;
  }
  return p;
}
void md_i_piece_close(md_ctx *ctx, md_piece *p) {
  if (!ctx || !p) return;
  DeviceGuard guard(ctx->device);
  hipStreamSynchronize(ctx->stream);
  hipFree(p->d_text);
  hipFree(p->d_out);
  hipFree(p->d_state);
  hipFree(p->d_queue);
  hipFree(p->d_desc);
  delete p;
}
// text: the bytes at positions [w0, w0 + text_len) (w0 a multiple of 64, at most 65536 - 64 behind the end of the piece
// before), of which the first `seen` went through the piece before already.  Positions count from an origin the caller moves up now and then so that they stay below MD_MAX_STREAM:
// rebase is how far it moved since the piece before (a multiple of 65536, at least 65536 below w0 as that piece counted
// it).  sum / isize: Adler-32 (CRC-32 for gzip) and length mod 2^32 of the whole input so far.  The piece's output
// stays in device memory (md_i_piece_out reads it); *status is MD_PIECE_AWAIT (1000) when the encoder waits for more.
int md_i_piece_run(md_ctx *ctx, md_piece *p, int format, const md_deflate_params *params, const uint8_t *text, size_t text_len,
                   size_t seen, uint64_t w0, uint64_t rebase, int first, int last, uint32_t sum, uint32_t isize, size_t out_cap,
                   size_t *out_len, int *status) {
  if (!ctx || !p || !params || !out_len || !status || (!text && text_len)) return MD_E_INVALID_ARGUMENT;
  md_deflate_params q;
  int rc = check_params(ctx, format, params, &q);
  if (rc != MD_OK) return rc;
  if (w0 + text_len > MD_MAX_STREAM) return fail(ctx, MD_E_INVALID_ARGUMENT, "piece beyond MD_MAX_STREAM");
  MD_ON_DEVICE(ctx);
  rc = grow(ctx, &p->d_text, &p->text_cap, text_len + 320, "hipMalloc(encoder text)");
  if (rc == MD_OK) rc = grow(ctx, &p->d_out, &p->out_cap, out_cap ? out_cap : 16, "hipMalloc(encoder output)");
  if (rc != MD_OK) return rc;
  uint64_t h[13] = {0, (uint64_t)text_len, w0 + text_len, 0, (uint64_t)out_cap, 0, w0, rebase, 0, 0, 0, 0, 0};  // ([8], [9]: state and queue slot)
  uint32_t *h32 = (uint32_t *)(h + 10);  // status, checksum, flags, -, sum, isize
  h32[2] = (first ? 1u : 0u) | (last ? 2u : 0u);
  h32[4] = sum;
  h32[5] = isize;
  uint64_t *d64 = (uint64_t *)p->d_desc;
  uint32_t *d32 = (uint32_t *)(d64 + 10);
  HIP_TRY(ctx, hipMemcpyAsync(d64, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
  if (text_len) HIP_TRY(ctx, hipMemcpyAsync(p->d_text, text, text_len, hipMemcpyHostToDevice, ctx->stream));
  PieceArgs pa{d64 + 1, p->d_queue, {d32 + 2, p->d_state, d64 + 6, d32 + 4}, seen > 512 ? (uint32_t)(seen - 512) : 0u};
  rc = deflate_launch(ctx, format, q.level, q.queue_len, q.driver, q.dynamic, q.matcher, q.gz_header, 1, (const uint8_t *)p->d_text,
                      d64 + 0, d64 + 2, (uint8_t *)p->d_out, d64 + 3, d64 + 4, d64 + 5, (int32_t *)d32, d32 + 1, nullptr,
                      text_len ? text_len : 1, &pa);
  if (rc != MD_OK) return rc;
  uint64_t olen = 0;
  int32_t st = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&olen, d64 + 5, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&st, d32, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (and the caller's text is his again)
  *out_len = (size_t)olen;
  *status = st;
  return MD_OK;
}
// One piece of each of n streams in ONE launch of the kernels (md_def_batch, stream_shim.cpp): texts, outputs, states
// and queues are the caller's device buffers, the descriptors host arrays of n entries.  flags as struct Piece's (bit 3:
// the stream takes no part in this launch).  Synchronous: the results are read back.
struct md_pieces_io {
  const uint64_t *text_off, *text_len, *abs_len, *out_off, *out_cap, *w0, *rebase;
  const uint32_t *flags, *sum, *isize;
  uint64_t *out_len;
  int32_t *status;
};
int md_i_pieces_run(md_ctx *ctx, int format, const md_deflate_params *params, size_t n, const uint8_t *d_text, uint8_t *d_out,
                    void *d_state, void *d_queue, void **d_desc, size_t *d_desc_bytes, const md_pieces_io *io, uint32_t match_skip) {
  if (!ctx || !params || !io || n == 0) return MD_E_INVALID_ARGUMENT;
  md_deflate_params q;
  int rc = check_params(ctx, format, params, &q);
  if (rc != MD_OK) return rc;
  MD_ON_DEVICE(ctx);
  const size_t desc_bytes = n * (10 * 8 + 6 * 4);
  rc = grow(ctx, d_desc, d_desc_bytes, desc_bytes, "hipMalloc(encoder batch descriptors)");
  if (rc != MD_OK) return rc;
  std::vector<uint64_t> hbuf((desc_bytes + 7) / 8);
  uint64_t *h64 = hbuf.data();
  uint64_t *in_off = h64, *front_len = h64 + n, *abs_len = h64 + 2 * n, *out_off = h64 + 3 * n, *out_cap = h64 + 4 * n,
           *out_len = h64 + 5 * n, *pos = h64 + 6 * n;
  uint32_t *h32 = (uint32_t *)(h64 + 10 * n);
  uint32_t *st = h32, *flags = h32 + 2 * n, *sums = h32 + 3 * n;
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) {
    const bool idle = (io->flags[i] & 8u) != 0;
    if (!idle && io->abs_len[i] > MD_MAX_STREAM) return fail(ctx, MD_E_INVALID_ARGUMENT, "piece beyond MD_MAX_STREAM");
    in_off[i] = io->text_off[i];
    front_len[i] = idle ? 0 : io->text_len[i];
    abs_len[i] = idle ? 0 : io->abs_len[i];
    out_off[i] = io->out_off[i];
    out_cap[i] = io->out_cap[i];
    out_len[i] = 0;
    pos[4 * i] = idle ? 0 : io->w0[i];
    pos[4 * i + 1] = idle ? 0 : io->rebase[i];
    pos[4 * i + 2] = i;
    pos[4 * i + 3] = i;
    st[i] = 0;
    h32[n + i] = 0;
    flags[i] = io->flags[i];
    sums[2 * i] = io->sum[i];
    sums[2 * i + 1] = io->isize[i];
    total += front_len[i];
  }
  uint64_t *d64 = (uint64_t *)*d_desc;
  uint32_t *d32 = (uint32_t *)(d64 + 10 * n);
  HIP_TRY(ctx, hipMemcpyAsync(d64, h64, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
  PieceArgs pa{d64 + n, d_queue, {d32 + 2 * n, d_state, d64 + 6 * n, d32 + 3 * n}, match_skip};
  rc = deflate_launch(ctx, format, q.level, q.queue_len, q.driver, q.dynamic, q.matcher, q.gz_header, n, d_text, d64, d64 + 2 * n,
                      d_out, d64 + 3 * n, d64 + 4 * n, d64 + 5 * n, (int32_t *)d32, d32 + n, nullptr, total ? total : 1, &pa);
  if (rc != MD_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(io->out_len, d64 + 5 * n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(io->status, d32, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MD_OK;
}
hipStream_t md_i_stream(md_ctx *ctx) { return ctx->stream; }
int md_i_device(md_ctx *ctx) { return ctx->device; }
int md_i_piece_out(md_ctx *ctx, const md_piece *p, size_t off, uint8_t *host, size_t len) {
  if (!ctx || !p || (!host && len)) return MD_E_INVALID_ARGUMENT;
  MD_ON_DEVICE(ctx);
  if (len) HIP_TRY(ctx, hipMemcpy(host, (const uint8_t *)p->d_out + off, len, hipMemcpyDeviceToHost));
  return MD_OK;
}

// md_deflate_batch_host for a batch of LONG streams laid out at equal distances (what a caller with n equal buffers has; C3):
// cutting it into slices of streams would leave the sequential kernel short of streams (it wants 4 096), so it is cut into
// slices of positions (deflate_in_slices: the kernels go on from the state the slice before left, same bytes out) and the
// copies ride along - the columns [k S, (k + 1) S) of every stream's input as ONE strided copy under the kernels of slice
// k - 1, and every output column that is final for all streams as one strided copy under the kernels of the next slice.
// Returns 1000 when the batch is not of that kind (GZip: the CRC-32 of a whole stream comes first; level 0; short or
// irregular streams): the caller then pipelines slices of streams as before.
extern "C++" {
static int deflate_host_positions(md_ctx *ctx, int format, const md_deflate_params *params, size_t n, const uint8_t *h_in, size_t in_bytes,
                                  const uint64_t *in_off, const uint64_t *in_len, uint8_t *h_out, size_t out_bytes, const uint64_t *out_off,
                                  const uint64_t *out_cap, uint64_t *out_len, int32_t *status, uint32_t *checksum, uint8_t *din, uint8_t *dout) {
  if (format == MD_FORMAT_GZIP || n < 2 || ctx->host_slices_max < 2) return 1000;
  md_deflate_params q;
  if (check_params(ctx, format, params, &q) != MD_OK) return 1000;  // (the usual path reports it)
  uint32_t max_chain = 0, nice = 0;
  md_deflate_level_params(q.driver, q.matcher, q.level, &max_chain, &nice);
  if (max_chain == 0) return 1000;
  const uint64_t ip = in_off[1] - in_off[0], op = out_off[1] - out_off[0];
  uint64_t longest = 0, cap_max = 0;
  for (size_t i = 0; i < n; i++) {
    if (in_off[i] != in_off[0] + i * ip || out_off[i] != out_off[0] + i * op || in_len[i] > ip || out_cap[i] > op) return 1000;
    longest = in_len[i] > longest ? in_len[i] : longest;
    cap_max = out_cap[i] > cap_max ? out_cap[i] : cap_max;
  }
  if (in_off[1] <= in_off[0] || out_off[1] <= out_off[0] || longest < 4 * kSliceMin || longest > MD_MAX_STREAM) return 1000;
  // four slices (more if the workspace cap asks for smaller ones), S a multiple of 32 KiB
  uint64_t S = ((longest + 3) / 4 + 32767) / 32768 * 32768;
  while (S > kSliceMin && ctx->front_cap_bytes && md_front_big_bytes(slice_positions(in_len, n, S)) > ctx->front_cap_bytes) S -= 32768;
  if (ctx->front_cap_bytes && md_front_big_bytes(slice_positions(in_len, n, S)) > ctx->front_cap_bytes) return 1000;
  const uint64_t nslices = (longest + S - 1) / S;
  if (!ctx->s_in && hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking) != hipSuccess) return fail(ctx, MD_E_HIP, "hipStreamCreate");
  if (!ctx->s_out && hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking) != hipSuccess) return fail(ctx, MD_E_HIP, "hipStreamCreate");
  EventList evs;
  std::vector<hipEvent_t> e_in(nslices + 1, nullptr);
  for (auto &e : e_in)
    if (!(e = evs.make())) return fail(ctx, MD_E_HIP, "hipEventCreate");
  hipError_t herr = hipSuccess;
  // columns [c0, c1) of every row of a blob laid out at `pitch`: rows 0 .. n - 2 as one strided copy, the last row by itself
  // (it may end where the blob ends)
  auto band = [&](bool to_device, uint64_t c0, uint64_t c1, hipStream_t cs) {
    if (c1 <= c0 || herr != hipSuccess) return;
    const uint64_t pitch = to_device ? ip : op, off0 = to_device ? in_off[0] : out_off[0], bytes = to_device ? in_bytes : out_bytes;
    uint8_t *d = (to_device ? din : dout) + off0 + c0;
    const uint8_t *hs = h_in + off0 + c0;
    uint8_t *hd = h_out + off0 + c0;
    const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    const uint64_t w = c1 - c0;
    if (n > 1) herr = to_device ? hipMemcpy2DAsync(d, pitch, hs, pitch, w, n - 1, kind, cs) : hipMemcpy2DAsync(hd, pitch, d, pitch, w, n - 1, kind, cs);
    const uint64_t last = off0 + (n - 1) * pitch + c0;
    uint64_t wl = w;
    if (last >= bytes) wl = 0;
    else if (last + wl > bytes) wl = bytes - last;
    if (wl && herr == hipSuccess)
      herr = to_device ? hipMemcpyAsync(d + (n - 1) * pitch, hs + (n - 1) * pitch, wl, kind, cs) : hipMemcpyAsync(hd + (n - 1) * pitch, d + (n - 1) * pitch, wl, kind, cs);
  };
  auto columns = [&](uint64_t k) { return std::make_pair(k * S < longest ? k * S : longest, (k + 1) * S < longest ? (k + 1) * S : longest); };
  // (the strided copies are enqueued right BEHIND a slice's kernel launches: should the runtime keep the calling thread
  // until such a copy is done, the kernels it is meant to run under are on the device already)
  uint64_t c_done = 0, c_ready = 0;  // output columns [0, c_done) are on their way to the host, [c_done, c_ready) are final
  const bool dbg_t = getenv("MD_DEBUG_HOSTPATH") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto stamp = [&](const char *what, uint64_t k) {
    if (dbg_t) fprintf(stderr, "[hostpath] %-14s slice %llu at %.2f ms\n", what, (unsigned long long)k,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  };
  SliceHooks hooks;
  hooks.before_slice = [&](uint64_t k) -> int {
    if (k == 0) {
      hipEvent_t e0 = evs.make();  // (what the caller queued on the context's stream comes first)
      if (!e0) return fail(ctx, MD_E_HIP, "hipEventCreate");
      if (hipEventRecord(e0, ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->s_in, e0, 0) != hipSuccess) return fail(ctx, MD_E_HIP, "hipEventRecord");
      const auto c = columns(0);
      band(true, c.first, c.second, ctx->s_in);
      if (herr == hipSuccess) herr = hipEventRecord(e_in[0], ctx->s_in);
    }
    if (herr == hipSuccess) herr = hipStreamWaitEvent(ctx->stream, e_in[k], 0);
    stamp("before", k);
    return herr == hipSuccess ? MD_OK : fail(ctx, MD_E_HIP, "deflate host path: copy-in", herr);
  };
  hooks.launched = [&](uint64_t k) -> int {
    stamp("launched", k);
    if (k + 1 < nslices) {  // the next slice's input under this slice's kernels
      const auto c = columns(k + 1);
      band(true, c.first, c.second, ctx->s_in);
      if (herr == hipSuccess) herr = hipEventRecord(e_in[k + 1], ctx->s_in);
    }
    stamp("h2d queued", k);
    if (c_ready > c_done) {  // what the slices before made final leaves under them too
      if (dbg_t) fprintf(stderr, "[hostpath] d2h columns [%llu, %llu)\n", (unsigned long long)c_done, (unsigned long long)c_ready);
      band(false, c_done, c_ready, ctx->s_out);
      c_done = c_ready;
    }
    stamp("copies queued", k);
    return herr == hipSuccess ? MD_OK : fail(ctx, MD_E_HIP, "deflate host path: copies", herr);
  };
  hooks.after_slice = [&](uint64_t k, const std::vector<uint64_t> &fin) -> int {
    // (the context's stream has been waited for: what the slice wrote is there)
    stamp("kernels done", k);
    uint64_t lo = ~0ull, hi = 0;
    for (uint64_t f : fin) {
      lo = f < lo ? f : lo;
      hi = f > hi ? f : hi;
    }
    // (bands begin and end on 4 KiB columns: a strided copy of odd offsets and widths ran at a quarter of the link's rate;
    // behind the longest output the rows hold nothing anybody reads, up to the distance between two of them)
    c_ready = k + 1 == nslices ? ((hi + 4095) & ~(uint64_t)4095) : (lo & ~(uint64_t)4095);  // at the end: everything, ragged rows included
    if (c_ready > op) c_ready = op;
    if (c_ready < c_done) c_ready = c_done;
    if (k + 1 == nslices && c_ready > c_done) {
      band(false, c_done, c_ready, ctx->s_out);
      c_done = c_ready;
    }
    return herr == hipSuccess ? MD_OK : fail(ctx, MD_E_HIP, "deflate host path: copy-out", herr);
  };
  std::vector<int32_t> r_st(n);
  std::vector<uint32_t> r_sum(n);
  int rc = deflate_in_slices(ctx, format, q, n, S, din, in_off, in_len, dout, out_off, out_cap, out_len, r_st.data(), r_sum.data(), nullptr, &hooks);
  // everything in flight ends before the call returns, whatever happened (the buffers are the caller's)
  const hipError_t a = hipStreamSynchronize(ctx->s_in), b2 = hipStreamSynchronize(ctx->stream), c = hipStreamSynchronize(ctx->s_out);
  stamp("all done", nslices);
  if (rc != MD_OK) return rc;
  if (a != hipSuccess || b2 != hipSuccess || c != hipSuccess) return fail(ctx, MD_E_HIP, "deflate host path", a != hipSuccess ? a : b2 != hipSuccess ? b2 : c);
  for (size_t i = 0; i < n; i++) {
    status[i] = r_st[i];
    if (checksum) checksum[i] = r_sum[i];
  }
  (void)cap_max;
  return MD_OK;
}
}  // extern "C++"

int md_deflate_batch_host(md_ctx *ctx, int format, const md_deflate_params *params,
                          size_t n, const uint8_t *h_in, size_t in_bytes, const uint64_t *in_off,
                          const uint64_t *in_len, uint8_t *h_out, size_t out_bytes,
                          const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len,
                          int32_t *status, uint32_t *checksum) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  if (n == 0) return MD_OK;
  if (!params || !in_off || !in_len || !out_off || !out_cap || !out_len || !status)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "null descriptor array");
  for (size_t i = 0; i < n; i++) {
    if (in_off[i] > in_bytes || in_len[i] > in_bytes - in_off[i])
      return fail(ctx, MD_E_INVALID_ARGUMENT, "input range out of bounds");
    if (in_len[i] > MD_MAX_STREAM) return fail(ctx, MD_E_INVALID_ARGUMENT, "stream longer than MD_MAX_STREAM");
    if (out_off[i] > out_bytes || out_cap[i] > out_bytes - out_off[i])
      return fail(ctx, MD_E_INVALID_ARGUMENT, "output range out of bounds");
  }
  MD_ON_DEVICE(ctx);
  int grc_ = grow(ctx, &ctx->host_in, &ctx->host_in_bytes, in_bytes + 64, "hipMalloc(host path input)");
  if (grc_ == MD_OK) grc_ = grow(ctx, &ctx->host_out, &ctx->host_out_bytes, out_bytes + 64, "hipMalloc(host path output)");
  if (grc_ == MD_OK) grc_ = grow(ctx, &ctx->host_desc, &ctx->host_desc_bytes, 5 * n * 8 + n * 8, "hipMalloc(host path descriptors)");
  if (grc_ != MD_OK) return grc_;
  uint8_t *din = (uint8_t *)ctx->host_in, *dout = (uint8_t *)ctx->host_out;
  uint64_t *d64 = (uint64_t *)ctx->host_desc;
  int32_t *dstatus = (int32_t *)(d64 + 5 * n);
  uint32_t *dsum = (uint32_t *)(dstatus + n);
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(d64 + 0 * n, in_off, n * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64 + 1 * n, in_len, n * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64 + 2 * n, out_off, n * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64 + 3 * n, out_cap, n * 8, hipMemcpyHostToDevice, st));
  {  // long streams in a regular layout: slices of POSITIONS, input arriving and output leaving under the kernels
    const int prc = deflate_host_positions(ctx, format, params, n, h_in, in_bytes, in_off, in_len, h_out, out_bytes, out_off, out_cap,
                                           out_len, status, checksum, din, dout);
    if (prc != 1000 /* not this kind of batch: slices of streams below */) return prc;
  }
  // the sequential kernel holds 16 streams per CU: a slice of fewer than 4 096 streams leaves the chip part empty for as
  // long as a stream takes, so a batch is only cut where every slice still has that many
  const std::vector<HostSlice> sl = host_slices(n, in_off, in_len, out_off, out_cap, 4096, (size_t)ctx->host_slices_max, in_bytes, out_bytes);
  int rc = host_pipeline(ctx, sl, h_in, din, h_out, dout, [&](size_t i0, size_t cnt) {
    md_deflate_params hp = *params;
    hp.total_in_bytes = 0;
    for (size_t i = i0; i < i0 + cnt; i++) hp.total_in_bytes += (size_t)in_len[i];
    if (hp.total_in_bytes == 0) hp.total_in_bytes = 1;  // all empty: still no read-back
    return md_deflate_batch_device(ctx, format, &hp, cnt, din, d64 + i0, d64 + n + i0, dout, d64 + 2 * n + i0, d64 + 3 * n + i0,
                                   d64 + 4 * n + i0, dstatus + i0, dsum + i0);
  });
  if (rc != MD_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out_len, d64 + 4 * n, n * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(status, dstatus, n * 4, hipMemcpyDeviceToHost, st));
  if (checksum) HIP_TRY(ctx, hipMemcpyAsync(checksum, dsum, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return MD_OK;
}

static int deflate_one(md_ctx *ctx, int format, int level, int queue_len, int driver, int dynamic, const md_gz_header *gz,
                       const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written) {
  if (!ctx || !written || (!src && src_len) || (!dst && dst_cap)) return MD_E_INVALID_ARGUMENT;
  uint64_t in_off = 0, in_len = src_len, out_off = 0, out_cap = dst_cap, out_len = 0;
  int32_t status = 0;
  const md_deflate_params p = {level, queue_len, driver, dynamic, MD_MATCHER_DE, gz, 0, 0};
  int rc = md_deflate_batch_host(ctx, format, &p, 1, src, src_len, &in_off, &in_len, dst, dst_cap, &out_off, &out_cap,
                                 &out_len, &status, nullptr);
  if (rc != MD_OK) return rc;
  *written = (size_t)out_len;
  return status;
}

int md_de_higher_compress(md_ctx *ctx, int queue_len, const uint8_t *src, size_t src_len,
                          uint8_t *dst, size_t dst_cap, size_t *written) {
  return deflate_one(ctx, MD_FORMAT_DEFLATE, 4, queue_len, MD_DRIVER_HIGHER, 1, nullptr, src, src_len, dst, dst_cap, written);
}

int md_zl_higher_compress(md_ctx *ctx, int level, int dynamic, int queue_len, const uint8_t *src,
                          size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written) {
  return deflate_one(ctx, MD_FORMAT_ZLIB, level, queue_len, MD_DRIVER_ZL, dynamic, nullptr, src, src_len, dst, dst_cap, written);
}

// One batch-of-one launch of the deflate kernel in one of its two partial modes (drivers 3 and 4 of
// deflate_kernel.hip): host buffers in, host buffers out.
static int deflate_partial(md_ctx *ctx, int level, int queue_len, int driver, int dynamic, int matcher, const void *src,
                           size_t src_len, void *dst, size_t dst_cap, size_t *out_bytes, uint32_t *hist316) {
  MD_ON_DEVICE(ctx);
  DevBuf din, dout, ddesc, dhist;
  if (din.alloc(src_len + 16) != hipSuccess || dout.alloc(dst_cap + 16) != hipSuccess || ddesc.alloc(6 * 8 + 16) != hipSuccess ||
      dhist.alloc(316 * 4) != hipSuccess)
    return fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc");
  uint64_t desc[5] = {0, src_len, 0, dst_cap, 0};
  uint64_t *d64 = (uint64_t *)ddesc.p;
  int32_t *dstatus = (int32_t *)(d64 + 5);
  hipStream_t st = ctx->stream;
  if (src_len) HIP_TRY(ctx, hipMemcpyAsync(din.p, src, src_len, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64, desc, sizeof desc, hipMemcpyHostToDevice, st));
  int rc = deflate_launch(ctx, MD_FORMAT_DEFLATE, level, queue_len, driver, dynamic, matcher, nullptr, 1, (const uint8_t *)din.p,
                          d64, d64 + 1, (uint8_t *)dout.p, d64 + 2, d64 + 3, d64 + 4, dstatus, nullptr, (uint32_t *)dhist.p,
                          src_len ? src_len : 1);
  if (rc != MD_OK) return rc;
  uint64_t out_len = 0;
  int32_t status = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&out_len, d64 + 4, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(&status, dstatus, 4, hipMemcpyDeviceToHost, st));
  if (hist316) HIP_TRY(ctx, hipMemcpyAsync(hist316, dhist.p, 316 * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  if (status == MD_OK && out_len) HIP_TRY(ctx, hipMemcpy(dst, dout.p, (size_t)out_len, hipMemcpyDeviceToHost));
  *out_bytes = (size_t)out_len;
  return status;
}

int md_de_lz77_compress(md_ctx *ctx, int level, int queue_len, int matcher, const uint8_t *src, size_t src_len,
                        uint32_t *cmds, size_t cmds_cap, size_t *ncmds, uint32_t *literals, uint32_t *distances) {
  if (!ctx || !ncmds || (!src && src_len) || (!cmds && cmds_cap)) return MD_E_INVALID_ARGUMENT;
  if (level < 0 || level > 9) return fail(ctx, MD_E_INVALID_ARGUMENT, "Invalid level of compression");
  if (queue_len < 4 || queue_len > (1 << 20) || (queue_len & (queue_len - 1)))
    return fail(ctx, MD_E_INVALID_ARGUMENT, "Length of queue MUST be a power of two");
  if (matcher != MD_MATCHER_DE && matcher != MD_MATCHER_LZ) return fail(ctx, MD_E_INVALID_ARGUMENT, "unknown matcher");
  if (src_len > MD_MAX_STREAM || cmds_cap > MD_MAX_STREAM / 4) return fail(ctx, MD_E_INVALID_ARGUMENT, "buffer too long");
  uint32_t hist[316];
  size_t bytes = 0;
  int st = deflate_partial(ctx, level, queue_len, 3, 1, matcher, src, src_len, cmds, cmds_cap * 4, &bytes, hist);
  *ncmds = bytes / 4;
  if (st == MD_OK) {
    if (literals) memcpy(literals, hist, 286 * 4);
    if (distances) memcpy(distances, hist + 286, 30 * 4);
  }
  return st;
}

int md_de_def_encode(md_ctx *ctx, int kind, const uint32_t *cmds, size_t ncmds, uint8_t *dst, size_t dst_cap,
                     size_t *written) {
  if (!ctx || !written || (!cmds && ncmds) || (!dst && dst_cap)) return MD_E_INVALID_ARGUMENT;
  if (kind < MD_BLOCK_FLAT || kind > MD_BLOCK_DYNAMIC) return fail(ctx, MD_E_INVALID_ARGUMENT, "unknown block kind");
  if (ncmds >= (1u << 20)) return fail(ctx, MD_E_INVALID_ARGUMENT, "more commands than the largest queue holds");
  for (size_t i = 0; i < ncmds; i++) {  // De.Queue's encodings only (lib/de.ml:2245-2266): the kernel indexes tables with the fields
    const uint32_t c = cmds[i];
    const bool ok = (c & 0x2000000u) ? ((c & ~0x2ffffffu) == 0 && ((c >> 16) & 0x1ff) <= 255 && (c & 0xffff) <= 32767) : c <= 256;
    if (!ok) return fail(ctx, MD_E_INVALID_ARGUMENT, "not a De.Queue command");
  }
  int queue_len = 4;
  while ((size_t)queue_len < ncmds + 1) queue_len <<= 1;  // Queue.create: a power of two that holds them all
  return deflate_partial(ctx, 4, queue_len, 4, kind, MD_MATCHER_DE, cmds, ncmds * 4, dst, dst_cap, written, nullptr);
}

int md_de_def_run(md_ctx *ctx, int queue_len, const uint32_t *ops, size_t nops, uint8_t *dst, size_t dst_cap, size_t *written,
                  uint8_t *results, size_t results_cap, size_t *nresults) {
  if (!ctx || !written || (!ops && nops) || (!dst && dst_cap) || (!results && results_cap)) return MD_E_INVALID_ARGUMENT;
  if (queue_len < 4 || queue_len > (1 << 20) || (queue_len & (queue_len - 1)))
    return fail(ctx, MD_E_INVALID_ARGUMENT, "Length of queue MUST be a power of two");
  if (nops > MD_MAX_STREAM / 4) return fail(ctx, MD_E_INVALID_ARGUMENT, "operation list too long");
  uint32_t res[316];
  memset(res, 0, sizeof res);
  int st = deflate_partial(ctx, 4, queue_len, 5, 0, MD_MATCHER_DE, ops, nops * 4, dst, dst_cap, written, res);
  if (st < 0 && st != MD_E_INVALID_ARGUMENT) return st;
  // the kernel reports the first 315 answers; *nresults = how many of them results[] received
  size_t n = res[0] < 315 ? res[0] : 315;
  if (n > results_cap) n = results_cap;
  for (size_t i = 0; i < n; i++) results[i] = (uint8_t)res[1 + i];
  if (nresults) *nresults = n;
  if (st == MD_E_INVALID_ARGUMENT) return fail(ctx, st, "not a De.Def operation list");
  if (st >= 0 && res[0] > n) return fail(ctx, MD_E_INVALID_ARGUMENT, "more encode answers than results[] (or the kernel's 315) can hold");
  return st;
}

// ---- one long stream on the whole chip (csrc/inflate_chunked.hip has the scheme and the kernels) --------------------
// De.Higher.uncompress / Zl.Higher.uncompress / Gz on ONE big input (lib/de.ml:4555-4571, lib/zl.ml:650-666,
// bin/decompress.ml:77-100).  Returns kNotHandled whenever anything is not exactly as a well-formed stream decoded in
// pieces should be - the caller then takes the serial path, whose statuses and counts are the reference's; MD_OK means
// the whole stream is decoded, verified against its checksum, and copied out.
extern "C++" {
namespace {
constexpr int kNotHandled = 1000;
uint32_t par_gf_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int k = 0; k < 32; k++) {
    p ^= b & (0u - ((a >> 31) & 1));
    a <<= 1;
    b = (b >> 1) ^ (0xedb88320u & (0u - (b & 1)));
  }
  return p;
}
// crc(A || B) from crc(A), crc(B) and |B| (reflected CRC-32: bit 31 = x^0)
uint32_t par_crc_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
  uint32_t sq = 0x00800000u, r = 0x80000000u;  // x^8, x^0
  for (uint64_t n = len_b; n; n >>= 1) {
    if (n & 1) r = par_gf_mul(r, sq);
    sq = par_gf_mul(sq, sq);
  }
  return par_gf_mul(crc_a, r) ^ crc_b;
}
struct ParPiece {
  uint64_t bit;      // first bit of the piece in the body (a block start)
  uint64_t u = 0;    // bytes it produces
};
}  // namespace

// What par_decode works on: a raw DEFLATE body in host memory that starts start_bit bits into body[0], the (at most 32 KiB
// of) output in front of it, room for dst_cap new bytes.  partial_ok: the body may end inside a block (a piece of a stream
// that is still arriving) - the complete blocks are decoded, the rest is the caller's.
struct ParIn {
  const uint8_t *body;
  uint64_t body_len;
  uint32_t start_bit;
  const uint8_t *hist;
  uint32_t hist_len;
  uint64_t dst_cap;
  bool partial_ok;
};
struct ParOut {
  int status;            // MD_OK: the final block ended; MD_UNEXPECTED_END_OF_INPUT (partial_ok): the body ended inside a block
  uint64_t total;        // new bytes, final, at ctx->par_out + hist_len
  uint64_t used_body;    // MD_OK: bytes of the body the stream used
  uint64_t resume_bits;  // bit of the body behind the last complete block
};
static int par_decode(md_ctx *ctx, const ParIn &in, ParOut *out) {
  ctx->par_last_pieces = ctx->par_last_rounds = 0;
  const bool dbg_t = getenv("MD_DEBUG_HOSTPATH") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto stamp = [&](const char *what) {
    if (dbg_t) fprintf(stderr, "[par_decode] %-22s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  };
  const uint8_t *body = in.body;
  const uint64_t body_len = in.body_len, dst_cap = in.dst_cap;
  const uint32_t hl = in.hist_len;
  // pieces of "inflate_parallel_chunk" (64 KiB) of input - smaller ones for a smaller stream, so that it still comes in a few
  // hundred pieces (a piece is a serial decode: 880 KB of 40-byte flush units in 14 pieces took 36 ms), never under 4 KiB
  uint64_t K = ctx->par_chunk;
  if (in.body_len / 256 < K) K = (in.body_len / 256 + 4095) & ~(uint64_t)4095;
  if (K < 4096) K = 4096;
  if (body_len < 4 * K || body_len > ((uint64_t)1 << 31) || hl + dst_cap > MD_MAX_STREAM) return kNotHandled;
  const uint32_t nchunks = (uint32_t)((body_len + K - 1) / K);
  hipStream_t st = ctx->stream;
  // -- the body on the device, candidate block starts
  int rc = grow(ctx, &ctx->par_in, &ctx->par_in_bytes, body_len + 64, "hipMalloc(parallel inflate input)");
  if (rc != MD_OK) return rc;
  const size_t desc_bytes = (size_t)nchunks * 2 * 160 + 4096;
  rc = grow(ctx, &ctx->par_desc, &ctx->par_desc_bytes, desc_bytes, "hipMalloc(parallel inflate descriptors)");
  if (rc != MD_OK) return rc;
  uint8_t *d_body = (uint8_t *)ctx->par_in;
  HIP_TRY(ctx, hipMemcpyAsync(d_body, body, body_len, hipMemcpyHostToDevice, st));
  uint64_t *d_cand = (uint64_t *)ctx->par_desc;
  if (dbg_t) {
    hipStreamSynchronize(st);
    stamp("body on the device");
  }
  int e = md_launch_find_blocks(d_body, body_len, K, nchunks - 1, d_cand, st);
  if (e != 0) return fail(ctx, MD_E_HIP, "find_blocks launch", (hipError_t)e);
  std::vector<uint64_t> cand(nchunks - 1);
  HIP_TRY(ctx, hipMemcpyAsync(cand.data(), d_cand, (nchunks - 1) * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  stamp("candidates found");
  // A run of STORED blocks from bit `pos` on is followed on the host, a read per block (lib/de.ml:1613-1627: header, padding,
  // LEN, NLEN): -> the bit behind the run (`pos` itself when the block there is not a stored, non-final one); *mid receives
  // block starts inside the run at least K bytes apart - stored data is the one kind whose block starts need no decoding to
  // be found, and a long run of it (incompressible input, level 0, compressed files inside a tar) should not be one piece.
  auto hop_stored = [&](uint64_t pos, std::vector<uint64_t> *mid) -> uint64_t {
    uint64_t last_mid = pos;
    for (int hops = 0; hops < (1 << 22); hops++) {
      const uint64_t by = pos >> 3;
      if (by + 5 > body_len) break;
      const uint32_t h = ((uint32_t)body[by] | ((uint32_t)body[by + 1] << 8)) >> (pos & 7);
      if ((h & 6) != 0 || (h & 1)) break;  // not stored, or the final block
      const uint64_t at = (pos + 3 + 7) >> 3;
      if (at + 4 > body_len) break;
      const uint32_t len = body[at] | ((uint32_t)body[at + 1] << 8), nlen = body[at + 2] | ((uint32_t)body[at + 3] << 8);
      if ((len ^ nlen) != 0xffffu || at + 4 + len > body_len) break;
      if (mid && pos >= last_mid + K * 8) {
        mid->push_back(pos);
        last_mid = pos;
      }
      pos = (at + 4 + len) * 8;
    }
    return pos;
  };
  std::vector<ParPiece> pc;
  pc.push_back(ParPiece{in.start_bit});
  {  // a stream that BEGINS with stored blocks: their starts are candidates the finder cannot see
    std::vector<uint64_t> mid;
    const uint64_t x = hop_stored(in.start_bit, &mid);
    if (x > in.start_bit && x < body_len * 8) mid.push_back(x);
    for (uint64_t c : mid)
      if (c > pc.back().bit) pc.push_back(ParPiece{c});
  }
  {
    const uint64_t seeded = pc.back().bit;
    for (uint64_t c : cand)
      if (c != ~0ull && c > seeded && c > pc.back().bit) pc.push_back(ParPiece{c});
  }
  if (pc.size() < 3) return kNotHandled;  // nothing to gain
  // -- decode, verify the chain of pieces, decode again without a candidate that proved false or with more room
  uint32_t capmul = 6;
  uint64_t total = 0, used_body = 0;
  out->status = MD_OK;
  out->resume_bits = 0;
  std::vector<uint64_t> offa, offb;
  for (int round = 1;; round++) {
    if (round > 8) return kNotHandled;
    ctx->par_last_rounds = round;
    const size_t np = pc.size(), n = 2 * np - 1;  // piece 0 once (its window is real), the others with window A and window B
    // out blob: [the final output: dst_cap][scratch of piece 1 A, 1 B, 2 A, ...]: 32 KiB of window + room, 64-byte aligned
    std::vector<uint64_t> in_off(n), in_len(n), out_off(n), out_cap(n);
    std::vector<uint32_t> start_bit(n), hist(n), adler_in(n, 1);
    std::vector<uint8_t> variant(n);
    offa.assign(np, 0);
    offb.assign(np, 0);
    uint64_t at = ((uint64_t)hl + dst_cap + 63) & ~(uint64_t)63;
    for (size_t p = 0; p < np; p++) {
      const uint64_t b0 = pc[p].bit >> 3, b1 = p + 1 < np ? (pc[p + 1].bit + 7) >> 3 : body_len;
      for (int v = 0; v < (p ? 2 : 1); v++) {
        const size_t i = p ? 2 * p - 1 + v : 0;
        in_off[i] = b0;
        in_len[i] = b1 - b0;
        start_bit[i] = (uint32_t)(pc[p].bit & 7);
        if (p == 0) {
          out_off[i] = 0;
          out_cap[i] = hl + dst_cap;
          hist[i] = hl;  // (the caller's window lies in front of the output)
          variant[i] = 0;
        } else {
          const uint64_t room = (uint64_t)capmul * (b1 - b0) + 65536;
          out_off[i] = at;
          out_cap[i] = 32768 + room;
          hist[i] = 32768;
          variant[i] = (uint8_t)(1 + v);
          (v ? offb : offa)[p] = at + 32768;
          at += (32768 + room + 64 + 63) & ~(uint64_t)63;
        }
      }
    }
    if (at > ((uint64_t)64 << 30)) return kNotHandled;
    rc = grow(ctx, &ctx->par_out, &ctx->par_out_bytes, at + 64, "hipMalloc(parallel inflate output)");
    if (rc != MD_OK) return rc;
    // descriptors: u64 x n: in_off in_len out_off out_cap out_len consumed resume_bits resume_out; u32 x n: start_bit hist
    // adler_in status checksum resume_adler resume_last; u8 x n: variant
    const size_t need = n * (8 * 8 + 7 * 4 + 1) + 256;
    rc = grow(ctx, &ctx->par_desc, &ctx->par_desc_bytes, need, "hipMalloc(parallel inflate descriptors)");
    if (rc != MD_OK) return rc;
    uint64_t *d64 = (uint64_t *)ctx->par_desc;
    uint32_t *d32 = (uint32_t *)(d64 + 8 * n);
    uint8_t *d8 = (uint8_t *)(d32 + 7 * n);
    uint8_t *d_out = (uint8_t *)ctx->par_out;
    if (hl) HIP_TRY(ctx, hipMemcpyAsync(d_out, in.hist, hl, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d64 + 0 * n, in_off.data(), n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d64 + 1 * n, in_len.data(), n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d64 + 2 * n, out_off.data(), n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d64 + 3 * n, out_cap.data(), n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d32 + 0 * n, start_bit.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d32 + 1 * n, hist.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d32 + 2 * n, adler_in.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d8, variant.data(), n, hipMemcpyHostToDevice, st));
    e = md_launch_fill_windows((uint32_t)n, d_out, d64 + 2 * n, d8, st);
    if (e != 0) return fail(ctx, MD_E_HIP, "fill_windows launch", (hipError_t)e);
    rc = md_inflate_continue_batch_device(ctx, n, d_body, d64 + 0 * n, d64 + 1 * n, d_out, d64 + 2 * n, d64 + 3 * n, d32 + 0 * n,
                                          d32 + 1 * n, d32 + 2 * n, d64 + 4 * n, d64 + 5 * n, (int32_t *)(d32 + 3 * n), d32 + 4 * n,
                                          d64 + 6 * n, d64 + 7 * n, d32 + 5 * n, d32 + 6 * n);
    if (rc != MD_OK) return rc;
    std::vector<uint64_t> r_used(n), r_bits(n), r_out(n);
    std::vector<int32_t> r_st(n);
    std::vector<uint32_t> r_last(n);
    HIP_TRY(ctx, hipMemcpyAsync(r_used.data(), d64 + 5 * n, n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(r_bits.data(), d64 + 6 * n, n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(r_out.data(), d64 + 7 * n, n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(r_st.data(), d32 + 3 * n, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(r_last.data(), d32 + 6 * n, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    stamp("pieces decoded");
    // The pieces in order: piece p must end - its last complete block - exactly where piece p + 1 starts ("links").  From
    // piece 0, the true start of the stream, an unbroken chain of links IS the stream's chain of blocks.  A candidate its
    // predecessor does not link to goes (all of them in one round: behind a false candidate the verdicts say little, and a
    // true block start dropped by mistake only costs a split), and everything is decoded again.
    bool again = false, chain = true;
    size_t last = np;  // index of the piece that holds the stream's final block
    std::vector<std::pair<uint64_t, uint64_t>> kill;  // candidates at bits (lo, hi] go
    std::vector<uint64_t> add;                        // block starts found by hopping over stored blocks
    // piece p ended its last complete block at bit e, in front of the next candidate: that candidate is no block start.  If
    // the block at e is STORED, so are all candidates inside it and the stored blocks behind it (compressed data inside the
    // plaintext - a tar of .gz files - is stored, and full of real block headers that are not this stream's): hop over
    // them on the host, a read per block, and the block start behind the run is a candidate the finder could not see.
    auto unlinked = [&](size_t p, uint64_t e) {
      uint64_t hi = p + 1 < np ? pc[p + 1].bit : ~0ull;
      std::vector<uint64_t> mid;
      const uint64_t pos = hop_stored(e, &mid);
      if (pos > e) {
        if (pos > hi) hi = pos;
        for (uint64_t c : mid)
          if (c > pc[p].bit) add.push_back(c);
        if (pos < body_len * 8) add.push_back(pos);
      }
      kill.push_back({pc[p].bit, hi == ~0ull ? pc[p].bit : hi});
    };
    total = 0;
    for (size_t p = 0; p < np; p++) {
      const size_t i = p ? 2 * p - 1 : 0;
      const uint64_t base_bits = (pc[p].bit >> 3) * 8;
      if (p && (r_st[i] != r_st[i + 1] || r_out[i] != r_out[i + 1] || r_bits[i] != r_bits[i + 1])) {
        if (chain) return kNotHandled;  // (the two decodes of a piece of the real chain differ in more than the window's bytes)
        if (p + 1 < np) kill.push_back({pc[p].bit, pc[p + 1].bit});
        again = true;
        continue;
      }
      const bool linked = r_st[i] == MD_UNEXPECTED_END_OF_INPUT && p + 1 < np && base_bits + r_bits[i] == pc[p + 1].bit && r_last[i] == 0;
      if (linked) {
        pc[p].u = r_out[i] - hist[i];
        total += pc[p].u;
      } else if (r_st[i] == MD_UNEXPECTED_END_OF_OUTPUT && p) {  // (piece 0 writes into the caller's room: the serial path's error)
        again = true;
        if (chain) {  // a piece of the real chain needs more room (the candidate behind it stays)
          capmul *= 6;
          if (capmul > 1300) return kNotHandled;
        } else if (p + 1 < np) kill.push_back({pc[p].bit, pc[p + 1].bit});  // (behind a false candidate: garbage that expands)
        chain = false;
      } else if (r_st[i] == MD_OK && chain) {  // the final block ended inside this piece: what follows is not the stream's
        pc[p].u = r_out[i] - hist[i];
        total += pc[p].u;
        used_body = (pc[p].bit >> 3) + r_used[i];
        out->resume_bits = base_bits + r_bits[i];
        out->status = MD_OK;
        last = p;
        break;
      } else if (chain && in.partial_ok && p + 1 == np && r_st[i] == MD_UNEXPECTED_END_OF_INPUT) {
        // the input ends inside the last piece: its complete blocks count, the caller goes on from the last block boundary
        pc[p].u = r_out[i] - hist[i];
        total += pc[p].u;
        out->resume_bits = base_bits + r_bits[i];
        out->status = MD_UNEXPECTED_END_OF_INPUT;
        last = p;
        break;
      } else {
        // on the real chain: a piece that runs over the next candidate makes that candidate false; anything else is an
        // error of the stream itself (or a stream that ends inside its last block): the serial path's
        const bool ran_over = r_st[i] == MD_UNEXPECTED_END_OF_INPUT && p + 1 < np && base_bits + r_bits[i] < pc[p + 1].bit;
        if (chain && !ran_over) return kNotHandled;
        if (ran_over) unlinked(p, base_bits + r_bits[i]);
        else if (p + 1 < np) kill.push_back({pc[p].bit, pc[p + 1].bit});
        again = true;
        chain = false;
      }
    }
    if (again) {
      std::vector<ParPiece> keep;
      for (size_t p = 0; p < np; p++) {
        bool dead = false;
        for (const auto &k : kill) dead = dead || (pc[p].bit > k.first && pc[p].bit <= k.second);
        if (!dead) keep.push_back(pc[p]);
      }
      for (uint64_t x : add) keep.push_back(ParPiece{x});  // (block starts read off the stream itself)
      std::sort(keep.begin(), keep.end(), [](const ParPiece &x, const ParPiece &y) { return x.bit < y.bit; });
      keep.erase(std::unique(keep.begin(), keep.end(), [](const ParPiece &x, const ParPiece &y) { return x.bit == y.bit; }), keep.end());
      pc.swap(keep);
      if (pc.size() < 2) return kNotHandled;
      continue;
    }
    if (last == np) return kNotHandled;
    if (last + 1 < np) pc.resize(last + 1);  // (the scratch of the pieces behind it is simply not looked at)
    break;
  }
  const size_t np = pc.size();
  if (total > dst_cap || np < 2) return kNotHandled;
  // -- windows, then every byte
  {
    std::vector<uint64_t> u(np), pos(np);
    uint64_t acc = 0;
    for (size_t p = 0; p < np; p++) {  // (the window the caller handed in counts as output in front of piece 0)
      u[p] = pc[p].u + (p ? 0 : hl);
      pos[p] = acc;
      acc += u[p];
    }
    // the windows: by pointer jumping over the whole chip (a table of 128 KiB per piece, twice, in groups of 1 023 pieces), or -
    // a handful of pieces - by the one-workgroup chain
    const uint32_t kGroup = 1023;
    const bool jumping = np >= 24;
    const size_t win_bytes = ((np * (size_t)32768 + 255) & ~(size_t)255);
    rc = grow(ctx, &ctx->par_win, &ctx->par_win_bytes, win_bytes + (jumping ? md_windows_work_bytes((uint32_t)np, kGroup) : 0) + 64,
              "hipMalloc(parallel inflate windows)");
    if (rc != MD_OK) return rc;
    const size_t need = np * 32 + 256;
    rc = grow(ctx, &ctx->par_desc, &ctx->par_desc_bytes, need, "hipMalloc(parallel inflate descriptors)");
    if (rc != MD_OK) return rc;
    uint64_t *d64 = (uint64_t *)ctx->par_desc;  // offa offb u pos | flag
    uint32_t *d_flag = (uint32_t *)(d64 + 4 * np);
    uint8_t *d_out = (uint8_t *)ctx->par_out;
    HIP_TRY(ctx, hipMemcpyAsync(d64 + 0 * np, offa.data(), np * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d64 + 1 * np, offb.data(), np * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d64 + 2 * np, u.data(), np * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d64 + 3 * np, pos.data(), np * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 64, st));
    if (jumping)
      e = md_launch_windows_parallel((uint32_t)np, kGroup, d_out, u[0], d_out, d64 + 0 * np, d64 + 1 * np, d64 + 2 * np, d64 + 3 * np,
                                     (uint8_t *)ctx->par_win, (uint32_t *)((uint8_t *)ctx->par_win + win_bytes), d_flag, st);
    else e = md_launch_window_chain((uint32_t)np, d_out, d_out, d64 + 0 * np, d64 + 1 * np, d64 + 2 * np, (uint8_t *)ctx->par_win, d_flag, st);
    if (e != 0) return fail(ctx, MD_E_HIP, "window_chain launch", (hipError_t)e);
    if (dbg_t) {
      hipStreamSynchronize(st);
      stamp("window chain");
    }
    e = md_launch_resolve((uint32_t)np, d_out, d_out, d64 + 0 * np, d64 + 1 * np, d64 + 2 * np, d64 + 3 * np,
                          (const uint8_t *)ctx->par_win, d_flag, st);
    if (e != 0) return fail(ctx, MD_E_HIP, "resolve launch", (hipError_t)e);
    // (the flag is read by the caller together with what it needs next)
    uint32_t flag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    stamp("windows + resolve");
    if (flag) return kNotHandled;  // a reference in front of the stream's start
  }
  out->total = total;
  out->used_body = used_body;
  ctx->par_last_pieces = (int)np;
  return MD_OK;
}

// Adler-32 of n bytes at d (device) going on from `adler`; CRC-32 of the same bytes (complete value)
static int par_adler(md_ctx *ctx, const uint8_t *d, uint64_t n, uint32_t adler, uint32_t *res) {
  *res = adler;
  if (n == 0) return MD_OK;
  const size_t nseg = (size_t)((n + 65535) / 65536);
  int rc = grow(ctx, &ctx->par_desc, &ctx->par_desc_bytes, nseg * 8 + 64, "hipMalloc(parallel inflate descriptors)");
  if (rc != MD_OK) return rc;
  uint32_t *d_sums = (uint32_t *)ctx->par_desc;
  int e = md_launch_adler_segments(d, n, 65536, d_sums, ctx->stream);
  if (e != 0) return fail(ctx, MD_E_HIP, "adler_segments launch", (hipError_t)e);
  std::vector<uint32_t> sums(2 * nseg);
  HIP_TRY(ctx, hipMemcpyAsync(sums.data(), d_sums, nseg * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t a = adler & 0xffffu, b = adler >> 16;
  for (size_t s = 0; s < nseg; s++) {
    const uint64_t len = s + 1 < nseg ? 65536 : n - (uint64_t)s * 65536;
    b = (b + (len % 65521) * a + sums[2 * s + 1]) % 65521;
    a = (a + sums[2 * s]) % 65521;
  }
  *res = (uint32_t)((b << 16) | a);
  return MD_OK;
}
static int par_crc(md_ctx *ctx, const uint8_t *d_base, uint64_t off, uint64_t n, uint32_t *res) {
  *res = 0;
  if (n == 0) return MD_OK;
  const size_t ncrc = (size_t)((n + ((1u << 20) - 1)) >> 20);
  int rc = grow(ctx, &ctx->par_desc, &ctx->par_desc_bytes, ncrc * 24 + 64, "hipMalloc(parallel inflate descriptors)");
  if (rc != MD_OK) return rc;
  uint64_t *d_off = (uint64_t *)ctx->par_desc, *d_len = d_off + ncrc;
  uint32_t *d_crc = (uint32_t *)(d_len + ncrc);
  std::vector<uint64_t> co(ncrc), cl(ncrc);
  for (size_t s = 0; s < ncrc; s++) {
    co[s] = off + ((uint64_t)s << 20);
    cl[s] = s + 1 < ncrc ? (uint64_t)1 << 20 : n - ((uint64_t)s << 20);
  }
  HIP_TRY(ctx, hipMemcpyAsync(d_off, co.data(), ncrc * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_len, cl.data(), ncrc * 8, hipMemcpyHostToDevice, ctx->stream));
  int e = md_launch_crc32((uint32_t)ncrc, d_base, d_off, d_len, d_crc, ctx->stream);
  if (e != 0) return fail(ctx, MD_E_HIP, "crc32 launch", (hipError_t)e);
  std::vector<uint32_t> crcs(ncrc);
  HIP_TRY(ctx, hipMemcpyAsync(crcs.data(), d_crc, ncrc * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint32_t crc = crcs[0];
  for (size_t s = 1; s < ncrc; s++) crc = par_crc_concat(crc, crcs[s], cl[s]);
  *res = crc;
  return MD_OK;
}

// md_de_inf_continue_host on a long piece: the blocks that are complete in it by par_decode; if the piece ends inside a
// block, that tail goes through the serial path from the block boundary on (window = the bytes just decoded), so what the
// caller sees - status, the output reached inside the incomplete block, the resume point, the checksums there - is what
// the serial path alone would have said.
static int continue_parallel(md_ctx *ctx, const uint8_t *src, size_t src_len, unsigned start_bit, uint8_t *dst, size_t hist_len,
                             size_t dst_cap, uint32_t adler_in, unsigned flags, size_t *dst_len, int *status, md_inf_resume *resume) {
  ParIn in{src, src_len, start_bit, dst, (uint32_t)hist_len, dst_cap - hist_len, true};
  ParOut po;
  int rc = par_decode(ctx, in, &po);
  if (rc != MD_OK) return rc;
  const uint8_t *d_out = (const uint8_t *)ctx->par_out;
  const int pieces = ctx->par_last_pieces, rounds = ctx->par_last_rounds;
  uint32_t adler = adler_in, crc = 0;
  rc = par_adler(ctx, d_out + hist_len, po.total, adler_in, &adler);
  if (rc == MD_OK && (flags & MD_CONT_CRC32)) rc = par_crc(ctx, d_out, hist_len, po.total, &crc);
  if (rc != MD_OK) return rc;
  if (po.total) HIP_TRY(ctx, hipMemcpy(dst + hist_len, d_out + hist_len, po.total, hipMemcpyDeviceToHost));
  const uint64_t O = hist_len + po.total;  // output position behind the last complete block
  if (po.status == MD_OK) {
    *dst_len = (size_t)O;
    *status = MD_OK;
    resume->bits = po.resume_bits;
    resume->out = O;
    resume->adler = adler;
    resume->last = 1;
    resume->consumed = po.used_body;
    resume->checksum = adler;
    resume->crc_out = resume->crc_end = crc;
    return MD_OK;
  }
  // the tail, serially: from bit B on, with the last 32 KiB in front of it as its window - in place
  const uint64_t B = po.resume_bits, hl2 = O < 32768 ? O : 32768, shift = O - hl2;
  size_t t_len = 0;
  int t_st = 0;
  md_inf_resume t;
  rc = continue_serial(ctx, src + (B >> 3), src_len - (size_t)(B >> 3), (unsigned)(B & 7), dst + shift, (size_t)hl2, dst_cap - (size_t)shift, adler,
                       flags, &t_len, &t_st, &t);
  if (rc != MD_OK) return rc;
  *dst_len = (size_t)shift + t_len;
  *status = t_st;
  resume->bits = (B >> 3) * 8 + t.bits;
  resume->out = shift + t.out;
  resume->adler = t.adler;
  resume->last = t.last;
  resume->consumed = (B >> 3) + t.consumed;
  resume->checksum = t.checksum;
  if (flags & MD_CONT_CRC32) {
    const uint64_t n_out = t.out - hl2, n_end = t_len - hl2;
    resume->crc_out = po.total ? (n_out ? par_crc_concat(crc, t.crc_out, n_out) : crc) : t.crc_out;
    resume->crc_end = po.total ? (n_end ? par_crc_concat(crc, t.crc_end, n_end) : crc) : t.crc_end;
  } else resume->crc_out = resume->crc_end = 0;
  ctx->par_last_pieces = pieces;
  ctx->par_last_rounds = rounds;
  return MD_OK;
}

static int inflate_parallel(md_ctx *ctx, int format, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                            size_t *consumed, size_t *written, uint32_t *checksum) {
  // -- the frame: anything but a plain valid header is the serial path's (it knows the reference's answer)
  size_t hdr = 0, trailer = 0;
  if (format == MD_FORMAT_ZLIB) {
    if (src_len < 6 || (((uint32_t)src[0] << 8) + src[1]) % 31 != 0 || (src[0] & 0xf) != 8) return kNotHandled;
    hdr = 2;
    trailer = 4;
  } else if (format == MD_FORMAT_GZIP) {
    if (src_len < 18 || src[0] != 0x1f || src[1] != 0x8b || (src[3] & 2)) return kNotHandled;  // (a header CRC: serial path)
    size_t p = 10;
    const uint32_t flg = src[3];
    if (flg & 4) {  // FEXTRA, big-endian length as the reference reads it (lib/gz.ml:455)
      if (src_len - p < 2) return kNotHandled;
      const size_t xl = ((size_t)src[p] << 8) | src[p + 1];
      p += 2;
      if (src_len - p < xl) return kNotHandled;
      p += xl;
    }
    for (int which = 0; which < 2; which++) {
      if (!(flg & (which == 0 ? 8u : 16u))) continue;
      while (p < src_len && src[p] != 0) p++;
      if (p >= src_len) return kNotHandled;
      p++;
    }
    hdr = p;
    trailer = 8;
  } else if (format != MD_FORMAT_DEFLATE) return kNotHandled;
  if (src_len < hdr + trailer) return kNotHandled;
  ParIn in{src + hdr, src_len - hdr, 0, nullptr, 0, dst_cap, false};  // (the trailer's bytes included: where the stream ends is the decoder's to say)
  ParOut po;
  int rc = par_decode(ctx, in, &po);
  if (rc != MD_OK) return rc;
  const uint64_t total = po.total, used_body = po.used_body;
  if (src_len - hdr - used_body < trailer) {
    ctx->par_last_pieces = 0;
    return kNotHandled;
  }
  const uint8_t *d_out = (const uint8_t *)ctx->par_out;
  const uint8_t *t = src + hdr + used_body;
  bool good = true;
  if (format == MD_FORMAT_ZLIB || (format == MD_FORMAT_DEFLATE && checksum)) {
    uint32_t adler = 1;
    rc = par_adler(ctx, d_out, total, 1u, &adler);
    if (rc != MD_OK) return rc;
    if (format == MD_FORMAT_ZLIB) {
      const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
      good = want == adler;
    }
    if (checksum) *checksum = adler;
  } else if (format == MD_FORMAT_GZIP) {
    uint32_t crc = 0;
    rc = par_crc(ctx, d_out, 0, total, &crc);
    if (rc != MD_OK) return rc;
    uint32_t want = 0, isize = 0;
    for (int k = 0; k < 4; k++) {
      want |= (uint32_t)t[k] << (8 * k);
      isize |= (uint32_t)t[4 + k] << (8 * k);
    }
    good = want == crc && isize == (uint32_t)total;
    if (checksum) *checksum = crc;
  }
  if (!good) {  // the serial path reports it
    ctx->par_last_pieces = 0;
    return kNotHandled;
  }
  if (total) HIP_TRY(ctx, hipMemcpy(dst, d_out, total, hipMemcpyDeviceToHost));
  *consumed = hdr + (size_t)used_body + trailer;
  *written = (size_t)total;
  return MD_OK;
}
}  // extern "C++"

static int inflate_one(md_ctx *ctx, int format, const uint8_t *src, size_t src_len, uint8_t *dst,
                       size_t dst_cap, size_t *consumed, size_t *written) {
  if (!ctx || !consumed || !written || (!src && src_len) || (!dst && dst_cap))
    return MD_E_INVALID_ARGUMENT;
  uint64_t in_off = 0, in_len = src_len, out_off = 0, out_cap = dst_cap, out_len = 0, used = 0;
  int32_t status = 0;
  int rc = md_inflate_batch_host(ctx, format, 1, src, src_len, &in_off, &in_len, dst, dst_cap,
                                 &out_off, &out_cap, &out_len, &used, &status, nullptr);
  if (rc != MD_OK) return rc;
  *consumed = (size_t)used;
  *written = (size_t)out_len;
  return status;
}

int md_de_inf_ns_inflate(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst,
                         size_t dst_cap, size_t *consumed, size_t *written) {
  return inflate_one(ctx, MD_FORMAT_DEFLATE, src, src_len, dst, dst_cap, consumed, written);
}

int md_zl_inf_ns_inflate(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst,
                         size_t dst_cap, size_t *consumed, size_t *written) {
  return inflate_one(ctx, MD_FORMAT_ZLIB, src, src_len, dst, dst_cap, consumed, written);
}

int md_gz_higher_compress(md_ctx *ctx, int level, int queue_len, const md_gz_header *header, const uint8_t *src,
                          size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written) {
  return deflate_one(ctx, MD_FORMAT_GZIP, level, queue_len, MD_DRIVER_ZL, 1, header, src, src_len, dst, dst_cap, written);
}

// De.Higher.uncompress / Zl.Higher.uncompress (lib/de.ml:4555-4571, lib/zl.ml:650-666): the whole stream in, the
// whole output out; the reference's `Error (`Msg s)` is md_status_string of the status returned
int md_de_higher_uncompress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written) {
  size_t used = 0;
  return inflate_one(ctx, MD_FORMAT_DEFLATE, src, src_len, dst, dst_cap, &used, written);
}
int md_zl_higher_uncompress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *written) {
  size_t used = 0;
  return inflate_one(ctx, MD_FORMAT_ZLIB, src, src_len, dst, dst_cap, &used, written);
}

// The accessors of a finished Gz.Inf decoder (filename / comment / os / extra, lib/gz.ml:612-633):
// where the header fields sit in src.  Framing only — the kernels have validated the header.
static void gz_meta_of(const uint8_t *s, size_t n, md_gz_meta *m) {
  memset(m, 0, sizeof *m);
  if (n < 10) return;
  m->flg = s[3];
  m->mtime = ((uint32_t)s[4] << 24) | ((uint32_t)s[5] << 16) | ((uint32_t)s[6] << 8) | s[7];
  m->xfl = s[8];
  m->os = s[9];
  size_t p = 10;
  if (m->flg & 4) {
    if (n - p < 2) return;
    const size_t xl = ((size_t)s[p] << 8) | s[p + 1];
    p += 2;
    if (n - p < xl) return;
    m->has_extra = 1;
    m->extra_off = p;
    m->extra_len = xl;
    p += xl;
  }
  for (int which = 0; which < 2; which++) {
    if (!(m->flg & (which == 0 ? 8u : 16u))) continue;
    size_t q = p;
    while (q < n && s[q] != 0) q++;
    if (q >= n) return;
    if (which == 0) {
      m->has_name = 1;
      m->name_off = p;
      m->name_len = q - p;
    } else {
      m->has_comment = 1;
      m->comment_off = p;
      m->comment_len = q - p;
    }
    p = q + 1;
  }
}

int md_gz_higher_uncompress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                            size_t *consumed, size_t *written, md_gz_meta *meta) {
  int st = inflate_one(ctx, MD_FORMAT_GZIP, src, src_len, dst, dst_cap, consumed, written);
  if (meta) {
    memset(meta, 0, sizeof *meta);
    if (st == MD_OK) gz_meta_of(src, src_len, meta);
  }
  return st;
}

static int lzo_batch_device(md_ctx *ctx, bool compress, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                            const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                            const uint64_t *d_out_cap, uint64_t *d_out_len, int32_t *d_status) {
  if (!ctx) return MD_E_INVALID_ARGUMENT;
  if (n == 0) return MD_OK;
  if (n > 0x7fffffffull) return fail(ctx, MD_E_INVALID_ARGUMENT, "too many streams in one batch");
  if (!d_in_off || !d_in_len || !d_out_off || !d_out_cap || !d_out_len || !d_status)
    return fail(ctx, MD_E_INVALID_ARGUMENT, "null descriptor array");
  MD_ON_DEVICE(ctx);
  // persistent workgroups (lzo_kernels.hip): as many as the chip holds at once, drawing streams from a counter
  const uint32_t slots = md_lzo_slots(compress ? 1 : 0, (uint32_t)ctx->cus);
  const size_t wgs = n < slots ? n : slots;
  if (compress) {  // Lzo's wrkmem: 16 K u16 entries per workgroup
    const size_t need = wgs * (size_t)(1u << 15);
    if (need > ctx->lzo_ws_bytes) {
      if (ctx->lzo_ws) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ctx->lzo_ws));
        ctx->lzo_ws = nullptr;
        ctx->lzo_ws_bytes = 0;
      }
      if (hipMalloc(&ctx->lzo_ws, need) != hipSuccess) return fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc(lzo wrkmem)");
      ctx->lzo_ws_bytes = need;
    }
  }
  HIP_TRY(ctx, hipMemsetAsync(ctx->counters, 0, 4, ctx->stream));
  const int e = compress ? md_launch_lzo_compress((uint32_t)n, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap,
                                                  d_out_len, d_status, (uint16_t *)ctx->lzo_ws, ctx->counters, slots, ctx->stream)
                         : md_launch_lzo_uncompress((uint32_t)n, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap,
                                                    d_out_len, d_status, ctx->counters, slots, ctx->stream);
  if (e != 0) return fail(ctx, MD_E_HIP, "lzo kernel launch", (hipError_t)e);
  return MD_OK;
}

int md_lzo_uncompress_batch_device(md_ctx *ctx, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                                   const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                   const uint64_t *d_out_cap, uint64_t *d_out_len, int32_t *d_status) {
  return lzo_batch_device(ctx, false, n, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len, d_status);
}
int md_lzo_compress_batch_device(md_ctx *ctx, size_t n, const uint8_t *d_in, const uint64_t *d_in_off,
                                 const uint64_t *d_in_len, uint8_t *d_out, const uint64_t *d_out_off,
                                 const uint64_t *d_out_cap, uint64_t *d_out_len, int32_t *d_status) {
  return lzo_batch_device(ctx, true, n, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len, d_status);
}

static int lzo_one(md_ctx *ctx, bool compress, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                   size_t *written) {
  if (!ctx || !written || (!src && src_len) || (!dst && dst_cap)) return MD_E_INVALID_ARGUMENT;
  MD_ON_DEVICE(ctx);
  DevBuf din, dout, ddesc;
  // compress over-copies up to 16 bytes past a short literal run: the reference's buffers need that room too
  if (din.alloc(src_len + 16) != hipSuccess || dout.alloc(dst_cap + 16) != hipSuccess || ddesc.alloc(6 * 8) != hipSuccess)
    return fail(ctx, MD_E_OUT_OF_MEMORY, "hipMalloc");
  uint64_t h[5] = {0, src_len, 0, dst_cap, 0};
  uint64_t *d64 = (uint64_t *)ddesc.p;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(din.p, src, src_len, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d64, h, sizeof h, hipMemcpyHostToDevice, st));
  int rc = lzo_batch_device(ctx, compress, 1, (const uint8_t *)din.p, d64, d64 + 1, (uint8_t *)dout.p, d64 + 2, d64 + 3,
                            d64 + 4, (int32_t *)(d64 + 5));
  if (rc != MD_OK) return rc;
  uint64_t out_len = 0;
  int32_t status = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&out_len, d64 + 4, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(&status, d64 + 5, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  if (status == MD_OK && out_len) HIP_TRY(ctx, hipMemcpy(dst, dout.p, (size_t)out_len, hipMemcpyDeviceToHost));
  *written = (size_t)out_len;
  return status;
}
int md_lzo_uncompress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                      size_t *written) {
  return lzo_one(ctx, false, src, src_len, dst, dst_cap, written);
}
int md_lzo_compress(md_ctx *ctx, const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                    size_t *written) {
  return lzo_one(ctx, true, src, src_len, dst, dst_cap, written);
}

}  // extern "C"
