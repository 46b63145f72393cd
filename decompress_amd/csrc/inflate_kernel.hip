// inflate_kernel.hip — batched RFC1951 inflate for gfx950 (MI355X, CDNA4).
//
// One independent stream per wavefront (one 64-lane workgroup per stream).
// Semantics per stream: De.Inf.Ns.inflate (reference lib/de.ml:1534-1823) /
// Zl.Inf.Ns.inflate (lib/zl.ml:391-417); the reference's symbol loop is
// `inflate` (lib/de.ml:1667-1712), its LUT builder `huffman` (lib/de.ml:523-638),
// the dynamic header `dynamic/table/inflate_table` (lib/de.ml:1733-1793) and the
// stored block `flat` (lib/de.ml:1613-1627).
//
// v1 layout (this file):
//   * bit reader: 256 B of compressed input live in one VGPR (one dword per
//     lane, coalesced 256-B loads, next chunk prefetched); the 64-bit bit
//     buffer is wave-uniform and refilled with v_readlane — no memory access
//     on the symbol critical path;
//   * Huffman LUTs (two-level, root 9/6 like the reference, 852/592 entries,
//     packed to 16 bits) live in LDS;
//   * the sliding window is an LDS ring of RING bytes that doubles as the
//     write-combining buffer: literals and wave-cooperative match copies go
//     LDS->LDS, every completed 1 KiB of output is flushed to HBM with 16-B
//     coalesced stores and folded into the Adler-32 (WInf.update,
//     lib/de.ml:453-455) in the same pass.  Matches that reach behind the ring
//     read the already-flushed bytes back from L2.
#include "inflate_common.hpp"

namespace md {

// ---------------------------------------------------------------------------
// Bit reader: wave-uniform 64-bit buffer fed from a register-resident chunk.
struct BitReader {
  const uint8_t *p;
  uint32_t nbytes;
  uint32_t lane;
  uint32_t cur, nxt;  // per-lane dwords: cur = word[cbase+lane], nxt = word[cbase+64+lane]
  uint32_t cbase;
  uint32_t widx;  // next word index to move into hold
  uint64_t hold;
  uint32_t bits;

  __device__ __forceinline__ uint32_t load_word(uint32_t w) const {
    uint64_t off = (uint64_t)w * 4;
    uint32_t v = 0;
    if (off + 4 <= nbytes) {
      __builtin_memcpy(&v, p + off, 4);
    } else {
      for (int k = 0; k < 4; k++)
        if (off + k < nbytes) v |= (uint32_t)p[off + k] << (8 * k);
    }
    return v;
  }
  __device__ __forceinline__ uint32_t next_word() {
    uint32_t l = widx - cbase;
    if (l >= 64) {
      cur = nxt;
      cbase += 64;
      nxt = load_word(cbase + 64 + lane);
      l -= 64;
    }
    widx++;
    return __builtin_amdgcn_readlane(cur, uni(l));
  }
  // Position the reader on byte `pos` of the stream.
  __device__ __forceinline__ void seek(uint32_t pos) {
    cbase = pos >> 2;
    widx = cbase;
    cur = load_word(cbase + lane);
    nxt = load_word(cbase + 64 + lane);
    hold = 0;
    bits = 0;
    uint32_t sh = (pos & 3) * 8;
    if (sh) {
      hold = next_word() >> sh;
      bits = 32 - sh;
    }
  }
  __device__ __forceinline__ void refill() {
    if (bits <= 32) {
      hold |= (uint64_t)next_word() << bits;
      bits += 32;
    }
  }
  __device__ __forceinline__ uint32_t peek(uint32_t n) const {
    return (uint32_t)hold & ((1u << n) - 1);
  }
  __device__ __forceinline__ void drop(uint32_t n) {
    hold >>= n;
    bits -= n;
  }
  // bits consumed so far; > 8*nbytes means the stream was truncated
  __device__ __forceinline__ uint64_t used_bits() const { return (uint64_t)widx * 32 - bits; }
  __device__ __forceinline__ bool overrun() const { return used_bits() > (uint64_t)nbytes * 8; }
  // real (non-padding) bits still available
  __device__ __forceinline__ int64_t avail_bits() const {
    return (int64_t)((uint64_t)nbytes * 8) - (int64_t)used_bits();
  }
};

// ---------------------------------------------------------------------------
// Output: LDS ring (window + write combining) flushed to HBM 1 KiB at a time.
template <uint32_t RING>
struct Sink {
  static constexpr uint32_t kMask = RING - 1;
  static constexpr uint32_t kNear = RING - kNearSlack;  // max distance served from LDS
  uint8_t *ring;
  uint8_t *g;
  uint32_t cap;
  uint32_t pos;
  uint32_t flushed;
  uint32_t lane;
  uint32_t a, b;  // Adler-32 state
  bool want_adler;

  __device__ __forceinline__ void adler_fold(uint32_t s1, uint32_t s2, uint32_t n) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    b = (b + n * a + s2) % 65521u;
    a = (a + s1) % 65521u;
  }
  // flush ring[flushed, flushed + kFlush) (a completed, kFlush-aligned KiB)
  __device__ __forceinline__ void flush_full() {
    uint32_t o = flushed + lane * 16;
    const uint4 v = *reinterpret_cast<const uint4 *>(ring + (o & kMask));
    __builtin_memcpy(g + o, &v, 16);
    if (want_adler) {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t s1 = 0, s2 = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        uint32_t d = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
        s1 += d;
        s2 += (kFlush - (lane * 16 + k)) * d;
      }
      adler_fold(s1, s2, kFlush);
    }
    flushed += kFlush;
  }
  // final partial flush of [flushed, pos)
  __device__ __forceinline__ void flush_tail() {
    uint32_t n = pos - flushed;
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t j = lane; j < n; j += kWave) {
      uint32_t d = ring[(flushed + j) & kMask];
      g[flushed + j] = (uint8_t)d;
      s1 += d;
      s2 += (n - j) * d;
    }
    if (want_adler && n) adler_fold(s1, s2, n);
    flushed = pos;
  }
  __device__ __forceinline__ void literal(uint32_t v) {
    if (lane == 0) ring[pos & kMask] = (uint8_t)v;
    pos++;
    if ((pos & (kFlush - 1)) == 0) flush_full();
  }
  // LZ77 copy of `len` bytes from `dist` back (WInf.blit / _blit, lib/de.ml:1595-1611):
  // forward byte semantics, overlapping allowed.
  __device__ __forceinline__ void match(uint32_t len, uint32_t dist) {
    while (len) {
      uint32_t seg = kFlush - (pos & (kFlush - 1));
      if (seg > len) seg = len;
      if (dist > kNear) {
        // behind the ring: those bytes are already flushed; read them back from L2
        for (uint32_t j = lane; j < seg; j += kWave) {
          uint8_t v = __hip_atomic_load(g + (pos - dist + j), __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
          ring[(pos + j) & kMask] = v;
        }
      } else if (dist >= kWave) {
        for (uint32_t j = lane; j < seg; j += kWave) {
          uint8_t v = ring[(pos - dist + j) & kMask];
          ring[(pos + j) & kMask] = v;
        }
      } else {
        // short period: every output byte is a replica of the last `dist` bytes
        for (uint32_t j = lane; j < seg; j += kWave) {
          uint8_t v = ring[(pos - dist + (j % dist)) & kMask];
          ring[(pos + j) & kMask] = v;
        }
      }
      pos += seg;
      len -= seg;
      if ((pos & (kFlush - 1)) == 0) flush_full();
    }
  }
  // stored block payload: `len` bytes straight from the input (flat, lib/de.ml:1613-1627)
  __device__ __forceinline__ void stored(const uint8_t *src, uint32_t len) {
    while (len) {
      uint32_t seg = kFlush - (pos & (kFlush - 1));
      if (seg > len) seg = len;
      for (uint32_t j = lane; j < seg; j += kWave) ring[(pos + j) & kMask] = src[j];
      src += seg;
      pos += seg;
      len -= seg;
      if ((pos & (kFlush - 1)) == 0) flush_full();
    }
  }
};

// ---------------------------------------------------------------------------
// Symbol loop: `inflate`, lib/de.ml:1667-1712.
template <uint32_t RING>
__device__ int inflate_block(BitReader &br, Sink<RING> &sk, const Lut &lit, const Lut &dist) {
  for (;;) {
    br.refill();
    uint32_t e = lut_lookup(lit, br.hold);
    uint32_t len = (e >> 9) & 15, value = e & 511;
    br.drop(len);
    if (br.overrun()) return MD_UNEXPECTED_END_OF_INPUT;
    if (value < 256) {
      if (sk.pos >= sk.cap) return MD_UNEXPECTED_END_OF_OUTPUT;
      sk.literal(value);
    } else if (value == 256) {
      return MD_OK;
    } else {
      uint32_t l = (value - 257) & 31;
      uint32_t xl = c_extra_lbits[l];
      uint32_t mlen = c_base_length[l] + 3 + br.peek(xl);
      br.drop(xl);
      if (br.overrun()) return MD_UNEXPECTED_END_OF_INPUT;
      br.refill();
      e = lut_lookup(dist, br.hold);
      if (e == kBad) return MD_INVALID_DISTANCE_CODE;  // D2
      len = (e >> 9) & 15;
      uint32_t dv = e & 31;
      br.drop(len);
      if (br.overrun()) return MD_UNEXPECTED_END_OF_INPUT;
      uint32_t xd = c_extra_dbits[dv];
      uint32_t d = c_base_dist1[dv] + br.peek(xd);
      br.drop(xd);
      if (br.overrun()) return MD_UNEXPECTED_END_OF_INPUT;
      if (d == 0) return MD_INVALID_DISTANCE_CODE;
      uint32_t lim = sk.pos < 32768u ? sk.pos : 32768u;
      if (d > lim) return MD_INVALID_DISTANCE;
      if (mlen > sk.cap - sk.pos) return MD_UNEXPECTED_END_OF_OUTPUT;
      sk.match(mlen, d);
    }
  }
}

// Dynamic block header: lib/de.ml:1733-1793.
template <uint32_t RING>
__device__ int dynamic_header(BitReader &br, Scratch *s, Lut *lit, Lut *dist, uint32_t lane) {
  br.refill();
  // _fill_bits d 14 (lib/de.ml:1789): fails only when fewer than 14 bits remain
  if (br.avail_bits() < 14) return MD_UNEXPECTED_END_OF_INPUT;
  uint32_t hlit = br.peek(5) + 257;
  br.drop(5);
  uint32_t hdist = br.peek(5) + 1;
  br.drop(5);
  uint32_t hclen = br.peek(4) + 4;
  br.drop(4);
  if (lane < 19) s->lens[lane] = 0;
  for (uint32_t i = 0; i < hclen; i++) {
    br.refill();
    if (br.avail_bits() < 3) return MD_UNEXPECTED_END_OF_INPUT;
    uint32_t v = br.peek(3);
    br.drop(3);
    if (lane == 0) s->lens[c_zigzag[i]] = (uint8_t)v;
  }
  Lut cl;
  if (!build_lut(K_CODES, s->lens, 19, s, &cl, lane)) return MD_INVALID_DICTIONARY;
  // the code lengths of the lit/len + distance alphabets go to lens[0 .. hlit+hdist)
  // (the 19 code-length lengths above are dead once `cl` is built)
  const uint32_t max_res = hlit + hdist;
  uint32_t i = 0;
  uint32_t prev = 0;
  while (i < max_res) {
    br.refill();
    if (br.avail_bits() < (int64_t)cl.maxl) return MD_UNEXPECTED_END_OF_INPUT;
    uint32_t e = uni(cl.t[(uint32_t)br.hold & ((1u << cl.maxl) - 1)]);
    if (e == kBad) return MD_INVALID_DICTIONARY;
    uint32_t sym = e & 511, len = (e >> 9) & 15;
    br.drop(len);
    if (sym < 16) {
      if (lane == 0) s->lens[i] = (uint8_t)sym;
      prev = sym;
      i++;
    } else {
      uint32_t nb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
      if (sym == 16 && i == 0) return MD_INVALID_DICTIONARY;
      br.refill();
      if (br.avail_bits() < (int64_t)nb) return MD_UNEXPECTED_END_OF_INPUT;
      uint32_t copy = br.peek(nb) + (sym == 18 ? 11 : 3);
      br.drop(nb);
      uint32_t val = sym == 16 ? prev : 0;
      if (i + copy > max_res) return MD_INVALID_DICTIONARY;
      for (uint32_t x = lane; x < copy; x += kWave) s->lens[i + x] = (uint8_t)val;
      prev = val;
      i += copy;
    }
  }
  if (uni(s->lens[256]) == 0) return MD_INVALID_DICTIONARY;
  if (!build_lut(K_LENS, s->lens, hlit, s, lit, lane)) return MD_INVALID_DICTIONARY;
  if (!build_lut(K_DISTS, s->lens + hlit, hdist, s, dist, lane)) return MD_INVALID_DICTIONARY;
  return MD_OK;
}

// fixed_lit / fixed_dist, lib/de.ml:821-833
__device__ void fixed_tables(Scratch *s, Lut *lit, Lut *dist, uint32_t lane) {
  for (uint32_t n = lane; n < 288; n += kWave)
    s->lens[n] = n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8;
  build_lut(K_LENS, s->lens, 288, s, lit, lane);
  if (lane < 32) {
    uint32_t r = __brev(lane) >> 27;  // 5-bit reversed code
    s->dist[lane] = (uint16_t)((5u << 9) | r);
  }
  dist->t = s->dist;
  dist->mask = 31;
  dist->root = 5;
  dist->maxl = 5;
}

template <uint32_t RING>
__global__ __launch_bounds__(kWave) void inflate_kernel(
    int format, uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *__restrict__ out,
    const uint64_t *__restrict__ out_off, const uint64_t *__restrict__ out_cap,
    uint64_t *__restrict__ out_len, uint64_t *__restrict__ consumed, int32_t *__restrict__ status,
    uint32_t *__restrict__ checksum) {
  extern __shared__ __align__(16) uint8_t smem[];
  uint8_t *ring = smem;
  Scratch *s = reinterpret_cast<Scratch *>(smem + RING);
  const uint32_t lane = threadIdx.x;
  const uint32_t sid = blockIdx.x;
  if (sid >= n) return;

  const uint8_t *src = in + in_off[sid];
  uint64_t slen64 = in_len[sid];
  uint64_t cap64 = out_cap[sid];
  uint32_t slen = slen64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)slen64;
  uint32_t cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;

  int rc = MD_OK;
  uint32_t body_off = 0, body_len = slen;
  if (format == MD_FORMAT_ZLIB) {
    // Zl.Inf.Ns.inflate, lib/zl.ml:400-417: header, then the body excludes the
    // last 4 bytes (Adler-32 trailer)
    if (slen < 2) rc = MD_UNEXPECTED_END_OF_INPUT;
    else {
      uint32_t cmf = src[0], flg = src[1];
      if (((cmf << 8) + flg) % 31 != 0 || (cmf & 0xf) != 8) rc = MD_INVALID_HEADER;
      else if (slen < 6) rc = MD_UNEXPECTED_END_OF_INPUT;
      else {
        body_off = 2;
        body_len = slen - 6;
      }
    }
  }

  Sink<RING> sk;
  sk.ring = ring;
  sk.g = out + out_off[sid];
  sk.cap = cap;
  sk.pos = 0;
  sk.flushed = 0;
  sk.lane = lane;
  sk.a = 1;
  sk.b = 0;
  sk.want_adler = (checksum != nullptr) || format == MD_FORMAT_ZLIB;

  BitReader br;
  br.p = src + body_off;
  br.nbytes = body_len;
  br.lane = lane;
  uint32_t used = 0;

  if (rc == MD_OK) {
    br.seek(0);
    Lut lit, dist;
    bool last = false;
    while (!last && rc == MD_OK) {
      // decode, lib/de.ml:1795-1805
      br.refill();
      if (br.avail_bits() < 3) {
        rc = MD_UNEXPECTED_END_OF_INPUT;
        break;
      }
      last = br.peek(1);
      br.drop(1);
      uint32_t type = br.peek(2);
      br.drop(2);
      if (type == 0) {
        // flat, lib/de.ml:1613-1627: give back whole bytes, then LEN/NLEN
        uint32_t p = (uint32_t)((br.used_bits() + 7) >> 3);
        if (body_len - p < 4) {
          rc = MD_UNEXPECTED_END_OF_INPUT;
          break;
        }
        const uint8_t *q = br.p + p;
        uint32_t len = q[0] | (q[1] << 8), nlen = q[2] | (q[3] << 8);
        len = uni(len);
        nlen = uni(nlen);
        p += 4;
        if (nlen != 0xffff - len) rc = MD_INVALID_COMPLEMENT_OF_LENGTH;
        else if (len > body_len - p) rc = MD_UNEXPECTED_END_OF_INPUT;
        else if (len > sk.cap - sk.pos) rc = MD_UNEXPECTED_END_OF_OUTPUT;
        else {
          sk.stored(br.p + p, len);
          p += len;
          br.seek(p);
        }
      } else if (type == 1) {
        fixed_tables(s, &lit, &dist, lane);
        rc = inflate_block<RING>(br, sk, lit, dist);
      } else if (type == 2) {
        rc = dynamic_header<RING>(br, s, &lit, &dist, lane);
        if (rc == MD_OK) rc = inflate_block<RING>(br, sk, lit, dist);
      } else {
        rc = MD_INVALID_KIND_OF_BLOCK;
      }
    }
    // bytes consumed: i_pos - (bits lsr 3), lib/de.ml:1805
    used = (uint32_t)((br.used_bits() + 7) >> 3);
  }
  sk.flush_tail();
  uint32_t adler = (sk.b << 16) | sk.a;
  if (rc == MD_OK && format == MD_FORMAT_ZLIB) {
    const uint8_t *t = src + 2 + used;
    uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    if (want != adler) rc = MD_INVALID_CHECKSUM;
    used += 6;
  }
  if (lane == 0) {
    out_len[sid] = sk.pos;
    consumed[sid] = rc == MD_OK ? used : 0;
    status[sid] = rc;
    if (checksum) checksum[sid] = adler;
  }
}

template <uint32_t RING>
constexpr size_t smem_bytes() { return RING + sizeof(Scratch); }

}  // namespace md

extern "C" int md_launch_inflate(int ring_log2, int format, uint32_t n, const uint8_t *in,
                                 const uint64_t *in_off, const uint64_t *in_len, uint8_t *out,
                                 const uint64_t *out_off, const uint64_t *out_cap,
                                 uint64_t *out_len, uint64_t *consumed, int32_t *status,
                                 uint32_t *checksum, hipStream_t stream) {
  if (n == 0) return 0;
  dim3 grid(n), block(md::kWave);
#define MD_LAUNCH(R)                                                                            \
  hipLaunchKernelGGL((md::inflate_kernel<R>), grid, block, md::smem_bytes<R>(), stream, format, \
                     n, in, in_off, in_len, out, out_off, out_cap, out_len, consumed, status,   \
                     checksum)
  switch (ring_log2) {
  case 12: MD_LAUNCH(4096u); break;
  case 13: MD_LAUNCH(8192u); break;
  case 14: MD_LAUNCH(16384u); break;
  case 15: MD_LAUNCH(32768u); break;
  default: return -1;
  }
#undef MD_LAUNCH
  return (int)hipGetLastError();
}
