// gz_kernels.hip — GZip framing (RFC1952 as the reference reads and writes it, lib/gz.ml) around
// the batched DEFLATE kernels, for gfx950.
//
//   gz_header_kernel   Gz.Inf header (lib/gz.ml:463-491): one thread per stream walks the fixed
//                      10 bytes, FEXTRA / FNAME / FCOMMENT / FHCRC and hands the DEFLATE body
//                      (offset, length) to the inflate kernels.  The reference's quirks are
//                      kept: FEXTRA's length is big-endian (lib/gz.ml:455), the header CRC16 is
//                      the UPPER half of the CRC-32 of the fixed bytes + name\0 + comment\0,
//                      FEXTRA excluded, stored big-endian (lib/gz.ml:422-439); CM is not checked.
//   crc32_kernel       Checkseum.Crc32 (call sites lib/gz.ml:503, :682) of one buffer per
//                      wavefront: every lane runs a 4-table CRC over its own contiguous
//                      segment, the 64 partial CRCs are joined with
//                          crc(A || B) = crc(A) * x^(8|B|) mod P  xor  crc(B)
//                      (each lane multiplies by x^(8 * bytes after it), one wave XOR).
//   gz_finish_kernel   Gz.Inf checksum (lib/gz.ml:344-356): CRC-32 first, then ISIZE, then
//                      the final per-stream results.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdeflate.h"

namespace md {
namespace gz {

constexpr int kWave = 64;
constexpr uint32_t kPoly = 0xedb88320u;  // reflected CRC-32 polynomial

// ---- bitwise CRC-32 for the few header bytes ----
__device__ __forceinline__ uint32_t crc_byte(uint32_t c, uint32_t b) {
  c ^= b;
#pragma unroll
  for (int k = 0; k < 8; k++) c = (c >> 1) ^ (kPoly & (0u - (c & 1)));
  return c;
}

__global__ void gz_header_kernel(uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
                                 const uint64_t *__restrict__ in_len, uint64_t *__restrict__ body_off,
                                 uint64_t *__restrict__ body_len, int32_t *__restrict__ hstatus) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t *s = in + in_off[i];
  const uint64_t len = in_len[i];
  int st = MD_OK;
  uint64_t p = 10;
  if (len < 10) st = MD_UNEXPECTED_END_OF_INPUT;
  else if (s[0] != 0x1f || s[1] != 0x8b) st = MD_INVALID_GZIP_HEADER;
  else {
    const uint32_t flg = s[3];
    uint32_t crc = 0xffffffffu;
    for (int k = 0; k < 10; k++) crc = crc_byte(crc, s[k]);
    if (flg & 4) {  // fextra
      if (len - p < 2) st = MD_UNEXPECTED_END_OF_INPUT;
      else {
        const uint64_t xl = ((uint64_t)s[p] << 8) | s[p + 1];
        p += 2;
        if (len - p < xl) st = MD_UNEXPECTED_END_OF_INPUT;
        else p += xl;
      }
    }
    for (int which = 0; which < 2 && st == MD_OK; which++) {  // fname, fcomment: zero-terminated
      if (!(flg & (which == 0 ? 8u : 16u))) continue;
      for (;;) {
        if (p >= len) {
          st = MD_UNEXPECTED_END_OF_INPUT;
          break;
        }
        const uint32_t b = s[p++];
        crc = crc_byte(crc, b);
        if (b == 0) break;
      }
    }
    if (st == MD_OK && (flg & 2)) {  // fhcrc
      if (len - p < 2) st = MD_UNEXPECTED_END_OF_INPUT;
      else {
        const uint32_t want = ((crc ^ 0xffffffffu) & 0xffff0000u) >> 16;
        const uint32_t have = ((uint32_t)s[p] << 8) | s[p + 1];
        if (want != have) st = MD_INVALID_GZIP_HEADER_CHECKSUM;
        else p += 2;
      }
    }
  }
  hstatus[i] = st;
  body_off[i] = in_off[i] + (st == MD_OK ? p : 0);
  body_len[i] = st == MD_OK ? len - p : 0;
}

// ---- GF(2)[x] / P in the reflected representation (bit 31 = x^0) ----
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
#pragma unroll 4
  for (int k = 0; k < 32; k++) {
    p ^= b & (0u - ((a >> 31) & 1));
    a <<= 1;
    b = (b >> 1) ^ (kPoly & (0u - (b & 1)));
  }
  return p;
}
// x^(8 * nbytes) mod P
__device__ __forceinline__ uint32_t gf_xpow8(uint64_t nbytes) {
  uint32_t sq = 0x00800000u;  // x^8
  uint32_t r = 0x80000000u;   // x^0
  while (nbytes) {
    if (nbytes & 1) r = gf_mul(r, sq);
    sq = gf_mul(sq, sq);
    nbytes >>= 1;
  }
  return r;
}

struct CrcTab {
  uint32_t t[4][256];
};
__device__ __forceinline__ void crc_tables(CrcTab *tb, uint32_t lane) {
  for (uint32_t i = lane; i < 256; i += kWave) {
    uint32_t c = i;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (kPoly & (0u - (c & 1)));
    tb->t[0][i] = c;
  }
  __syncthreads();
  for (uint32_t i = lane; i < 256; i += kWave) {
    uint32_t c = tb->t[0][i];
    for (int k = 1; k < 4; k++) {
      c = tb->t[0][c & 0xff] ^ (c >> 8);
      tb->t[k][i] = c;
    }
  }
  __syncthreads();
}
__device__ __forceinline__ uint32_t crc_word(const CrcTab *tb, uint32_t c, uint32_t w) {
  c ^= w;
  return tb->t[3][c & 0xff] ^ tb->t[2][(c >> 8) & 0xff] ^ tb->t[1][(c >> 16) & 0xff] ^ tb->t[0][c >> 24];
}
// standard CRC-32 (init and final xor ~0) of buf[0, len), whole wave; result on every lane
__device__ uint32_t crc32_wave(const CrcTab *tb, const uint8_t *buf, uint64_t len, uint32_t lane) {
  // (segments of whole 64-byte lines: a lane takes a line at a time, four 16-byte loads issued together - with one 16-byte
  // load per step the 64 lines a step touches were back in L2 before their other three quarters were asked for: the
  // 32 wavefronts of a CU hold 128 KiB of such lines against 16 KiB of L1, and the kernel ran at a fifth of HBM's rate)
  const uint64_t seg = ((len + kWave - 1) / kWave + 63) & ~(uint64_t)63;
  uint64_t a = (uint64_t)lane * seg, b = a + seg;
  if (a > len) a = len;
  if (b > len) b = len;
  uint32_t c = 0xffffffffu;
  const uint8_t *q = buf + a, *e = buf + b;
  while (q < e && ((uintptr_t)q & 63) != 0) c = tb->t[0][(c ^ *q++) & 0xff] ^ (c >> 8);
  for (; q + 64 <= e; q += 64) {
    const uint4 v0 = *(const uint4 *)q, v1 = *(const uint4 *)(q + 16), v2 = *(const uint4 *)(q + 32), v3 = *(const uint4 *)(q + 48);
    c = crc_word(tb, c, v0.x);
    c = crc_word(tb, c, v0.y);
    c = crc_word(tb, c, v0.z);
    c = crc_word(tb, c, v0.w);
    c = crc_word(tb, c, v1.x);
    c = crc_word(tb, c, v1.y);
    c = crc_word(tb, c, v1.z);
    c = crc_word(tb, c, v1.w);
    c = crc_word(tb, c, v2.x);
    c = crc_word(tb, c, v2.y);
    c = crc_word(tb, c, v2.z);
    c = crc_word(tb, c, v2.w);
    c = crc_word(tb, c, v3.x);
    c = crc_word(tb, c, v3.y);
    c = crc_word(tb, c, v3.z);
    c = crc_word(tb, c, v3.w);
  }
  for (; q + 16 <= e; q += 16) {
    const uint4 v = *(const uint4 *)q;
    c = crc_word(tb, c, v.x);
    c = crc_word(tb, c, v.y);
    c = crc_word(tb, c, v.z);
    c = crc_word(tb, c, v.w);
  }
  while (q < e) c = tb->t[0][(c ^ *q++) & 0xff] ^ (c >> 8);
  c ^= 0xffffffffu;
  if (a == b) c = 0;  // crc of nothing
  uint32_t term = gf_mul(gf_xpow8(len - b), c);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) term ^= __shfl_xor(term, o);
  return term;
}

__global__ __launch_bounds__(kWave) void crc32_kernel(uint32_t n, const uint8_t *__restrict__ data,
                                                      const uint64_t *__restrict__ off,
                                                      const uint64_t *__restrict__ len,
                                                      uint32_t *__restrict__ crc_out) {
  __shared__ CrcTab tb;
  const uint32_t lane = threadIdx.x, i = blockIdx.x;
  if (i >= n) return;
  crc_tables(&tb, lane);
  const uint32_t c = crc32_wave(&tb, data + off[i], len[i], lane);
  if (lane == 0) crc_out[i] = c;
}

// after the DEFLATE kernels ran on the bodies: trailer check and the stream's final results
__global__ __launch_bounds__(kWave) void gz_finish_kernel(
    uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, const uint64_t *__restrict__ body_off,
    const int32_t *__restrict__ hstatus, const uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    uint64_t *__restrict__ out_len, uint64_t *__restrict__ consumed, int32_t *__restrict__ status,
    uint32_t *__restrict__ checksum) {
  __shared__ CrcTab tb;
  const uint32_t lane = threadIdx.x, i = blockIdx.x;
  if (i >= n) return;
  const int hs = hstatus[i];
  int st = hs != MD_OK ? hs : status[i];
  uint64_t used = 0, wrote = hs != MD_OK ? 0 : out_len[i];
  uint32_t crc = 0;
  if (st == MD_OK) {  // uniform
    crc_tables(&tb, lane);
    crc = crc32_wave(&tb, out + out_off[i], wrote, lane);
    const uint64_t hdr = body_off[i] - in_off[i], body = consumed[i];
    if (in_len[i] - hdr - body < 8) st = MD_UNEXPECTED_END_OF_INPUT;
    else {
      const uint8_t *t = in + body_off[i] + body;
      uint32_t want = 0, isize = 0;
      for (int k = 0; k < 4; k++) {
        want |= (uint32_t)t[k] << (8 * k);
        isize |= (uint32_t)t[4 + k] << (8 * k);
      }
      if (want != crc) st = MD_INVALID_CHECKSUM;  // crc first, then isize (lib/gz.ml:351-354)
      else if (isize != (uint32_t)wrote) st = MD_INVALID_SIZE;
      else used = hdr + body + 8;
    }
  }
  if (lane == 0) {
    status[i] = st;
    consumed[i] = used;
    out_len[i] = wrote;
    if (checksum) checksum[i] = crc;
  }
}


// The text of n streaming encoders for their next launch (md_def_batch, stream_shim.cpp): stream i's region of the new
// blob is the tail of its region in the old one (the window the matcher can still reach) followed by the bytes that
// arrived since (packed in `fresh`).  d[6 i ..]: old offset of the tail, its length, offset in `fresh`, fresh length,
// new offset (unused).  Grid = (chunks of 64 KiB, stream): a thread copies 16 destination-aligned bytes per step (the
// source is read with unaligned 16-byte loads), the few bytes in front of the first and behind the last aligned chunk of a
// part one by one (it was one byte per thread and step, one workgroup per stream).
constexpr uint32_t kGatherChunk = 65536;
__device__ __forceinline__ void gather_part(uint8_t *__restrict__ o, const uint8_t *__restrict__ a, uint64_t len, uint64_t c0, uint64_t c1) {
  // bytes [c0, c1) of the part (c0, c1 within [0, len]); 16-byte steps on o's alignment
  if (c0 >= c1) return;
  const uint64_t head = (16 - ((uintptr_t)(o + c0) & 15)) & 15;
  const uint64_t h1 = c0 + head < c1 ? c0 + head : c1;
  for (uint64_t k = c0 + threadIdx.x; k < h1; k += 256) o[k] = a[k];
  const uint64_t body = (c1 - h1) & ~(uint64_t)15;
  for (uint64_t k = h1 + (uint64_t)threadIdx.x * 16; k < h1 + body; k += 256 * 16) {
    uint4 v;
    __builtin_memcpy(&v, a + k, 16);
    *reinterpret_cast<uint4 *>(o + k) = v;
  }
  for (uint64_t k = h1 + body + threadIdx.x; k < c1; k += 256) o[k] = a[k];
  (void)len;
}
__global__ __launch_bounds__(256) void piece_gather_kernel(uint32_t n, const uint8_t *__restrict__ old_blob, const uint8_t *__restrict__ fresh,
                                                           uint8_t *__restrict__ new_blob, const uint64_t *__restrict__ d) {
  for (uint32_t i = blockIdx.y; i < n; i += gridDim.y) {
    const uint64_t toff = d[6 * i], tlen = d[6 * i + 1], foff = d[6 * i + 2], flen = d[6 * i + 3], noff = d[6 * i + 4];
    uint8_t *o = new_blob + noff;
    // the stream's new text is [0, tlen) from the old blob, then [tlen, tlen + flen) from `fresh`; this workgroup takes the
    // chunks blockIdx.x, blockIdx.x + gridDim.x, ... of it
    for (uint64_t c0 = (uint64_t)blockIdx.x * kGatherChunk; c0 < tlen + flen; c0 += (uint64_t)gridDim.x * kGatherChunk) {
      const uint64_t c1 = c0 + kGatherChunk < tlen + flen ? c0 + kGatherChunk : tlen + flen;
      if (c0 < tlen) gather_part(o, old_blob + toff, tlen, c0, c1 < tlen ? c1 : tlen);
      if (c1 > tlen) gather_part(o + tlen, fresh + foff, flen, (c0 > tlen ? c0 : tlen) - tlen, c1 - tlen);
    }
  }
}

}  // namespace gz
}  // namespace md

extern "C" int md_launch_gz_header(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                   uint64_t *body_off, uint64_t *body_len, int32_t *hstatus, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::gz::gz_header_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, in, in_off, in_len,
                     body_off, body_len, hstatus);
  return (int)hipGetLastError();
}
extern "C" int md_launch_crc32(uint32_t n, const uint8_t *data, const uint64_t *off, const uint64_t *len,
                               uint32_t *crc_out, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::gz::crc32_kernel, dim3(n), dim3(md::gz::kWave), 0, stream, n, data, off, len, crc_out);
  return (int)hipGetLastError();
}
extern "C" int md_launch_gz_finish(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                                   const uint64_t *body_off, const int32_t *hstatus, const uint8_t *out,
                                   const uint64_t *out_off, uint64_t *out_len, uint64_t *consumed, int32_t *status,
                                   uint32_t *checksum, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::gz::gz_finish_kernel, dim3(n), dim3(md::gz::kWave), 0, stream, n, in, in_off, in_len,
                     body_off, hstatus, out, out_off, out_len, consumed, status, checksum);
  return (int)hipGetLastError();
}
extern "C" int md_launch_piece_gather(uint32_t n, const uint8_t *old_blob, const uint8_t *fresh, uint8_t *new_blob, const uint64_t *d,
                                      hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(md::gz::piece_gather_kernel, dim3(16, n < 65535u ? n : 65535u), dim3(256), 0, stream, n, old_blob, fresh, new_blob, d);
  return (int)hipGetLastError();
}
