// inflate_split.hip — the lane-parallel inflate as TWO kernels (gfx950).
//
// The fused kernel (inflate_v4.hip) is latency-bound per wavefront and its 22.8 KiB of
// LDS (input ring + LUT + tokens + staging) admits only 7 wavefronts per CU.  Huffman
// decoding and LZ77 resolution do not need each other's on-chip state, so here they are
// separate launches, each with half the LDS and twice the resident wavefronts:
//
//   K1  decode_kernel   one stream per wavefront; input ring + fat LUT in LDS (14 KiB).
//       Speculative zone decode + chain validation exactly as in inflate_lane.hpp, but
//       the tokens of every accepted round go to a per-stream log in HBM (coalesced
//       [index][lane] rows) instead of LDS.  Stored blocks and the end of the stream are
//       log records too.  K1 never looks at the output.
//   K2  resolve_kernel  one stream per wavefront; token rows of one round + staging buffer
//       in LDS (13 KiB).  Walks the log in order: round -> emit_round (literals, far and
//       near matches, position-dependent checks) -> coalesced flush + Adler-32; stored ->
//       copy; end -> zlib trailer, status, counts.
//
// Error order is the oracle's: K2 meets position-dependent failures (Invalid_distance,
// Unexpected_end_of_output) at the token where they occur, and a bitstream failure found
// by K1 only after every token before it has been placed.
#include "inflate_lane.hpp"

namespace md {
namespace v4 {

enum : uint32_t { REC_ROUND = 1, REC_STORED = 2, REC_END = 3 };

template <class C>
struct Rec {
  static constexpr uint32_t OFF_NLM = 16;                // u32[64]: nlit | nmat << 8
  static constexpr uint32_t OFF_NB = OFF_NLM + 256;      // u32[64]: bytes produced by the lane
  static constexpr uint32_t OFF_TMASK = OFF_NB + 256;    // u64[64]: token type mask
  static constexpr uint32_t OFF_LITS = OFF_TMASK + 512;  // u8[LMAX][64]
  static constexpr uint32_t OFF_MREC = (OFF_LITS + C::LMAX * 64 + 255) & ~255u;  // u32[MMAX][64]
  static constexpr uint32_t BYTES = (OFF_MREC + C::MMAX * 256 + 255) & ~255u;
};

// LUT construction scratch without the two big 16-bit tables: those are built INSIDE the
// fat-LUT area (upper half of each region) and widened in place.  Same field order as the
// tail of Scratch, so a Scratch* that points sizeof(lit)+sizeof(dist) below it sees them.
struct ScratchTail {
  uint16_t codes[128];
  uint16_t work[320];
  uint8_t lens[320];
  uint16_t cnt[16];
  uint16_t offs[16];
};
static_assert(sizeof(Scratch) == (852 + 592) * 2 + sizeof(ScratchTail), "ScratchTail must mirror Scratch's tail");
typedef uint16_t __attribute__((may_alias)) u16a;
typedef uint32_t __attribute__((may_alias)) u32a;

template <class C>
struct Smem1 {
  uint32_t inring[C::IN_WORDS + 4];
  uint32_t lut[852 + 592];
  ScratchTail st;
};
template <class C>
struct Smem2 {
  uint32_t mrec[C::MMAX * kWave];
  uint8_t lits[C::LMAX * kWave];
  alignas(16) uint8_t stage[C::STAGE + 16];
  uint8_t owner[C::STAGE / 32 + 8];
};

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = __shfl_xor(v, o);
    v = t > v ? t : v;
  }
  return v;
}
__device__ __forceinline__ void put_hdr(uint8_t *r, uint32_t lane, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  if (lane == 0) {
    uint4 v = make_uint4(a, b, c, d);
    *reinterpret_cast<uint4 *>(r) = v;
  }
}

// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(kWave) void decode_kernel(
    int format, uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *__restrict__ log, uint64_t log_stride,
    uint32_t log_cap) {
  extern __shared__ __align__(16) uint8_t smem_raw[];
  Smem1<C> *smg = reinterpret_cast<Smem1<C> *>(smem_raw);
  Prof<false> pf;
  const uint32_t lane = threadIdx.x, sid = blockIdx.x;
  if (sid >= n) return;
  using R = Rec<C>;
  uint8_t *lg = log + (size_t)sid * log_stride;
  uint32_t rec = 0;

  const uint8_t *src = in + in_off[sid];
  uint64_t slen64 = in_len[sid];
  uint32_t slen = slen64 > 0x1ffffff0ull ? 0x1ffffff0u : (uint32_t)slen64;
  int rc = MD_OK;
  uint32_t body_off = 0, body_len = slen;
  if (format == MD_FORMAT_ZLIB) {  // Zl.Inf.Ns.inflate, lib/zl.ml:400-417
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
  Input<C> inp;
  inp.p = src + body_off;
  inp.nbytes = body_len;
  inp.lane = lane;
  inp.ring = (lds_u32 *)smg->inring;
  inp.reset(0);
  const uint32_t total_bits = body_len * 8;
  const lds_u32 *lut = (const lds_u32 *)smg->lut;
  uint32_t bp = 0;

  bool last = false;
  while (!last && rc == MD_OK) {
    if (rec + 2 > log_cap) {
      rc = kStatusLogFull;
      break;
    }
    inp.ensure(bp >> 3);
    UReader<C> ur{&inp, bp, total_bits};
    if (ur.avail() < 3) {
      rc = MD_UNEXPECTED_END_OF_INPUT;
      break;
    }
    last = ur.peek(1);
    ur.drop(1);
    uint32_t type = ur.peek(2);
    ur.drop(2);
    bp = ur.bp;
    if (type == 0) {  // flat, lib/de.ml:1613-1627
      uint32_t p = (bp + 7) >> 3;
      if (body_len - p < 4) {
        rc = MD_UNEXPECTED_END_OF_INPUT;
        break;
      }
      uint32_t hdr = uni(inp.peek(p * 8));
      uint32_t len = hdr & 0xffff, nlen = hdr >> 16;
      p += 4;
      if (nlen != 0xffff - len) rc = MD_INVALID_COMPLEMENT_OF_LENGTH;
      else if (len > body_len - p) rc = MD_UNEXPECTED_END_OF_INPUT;
      else {
        put_hdr(lg + (size_t)rec * R::BYTES, lane, REC_STORED, body_off + p, len, 0);
        rec++;
        p += len;
        bp = p * 8;
        inp.reset(p);
      }
    } else if (type == 3) {
      rc = MD_INVALID_KIND_OF_BLOCK;
    } else {
      Lut lit, dist;
      // 16-bit tables in the upper half of each fat region; Scratch* is only a view of the tail
      Scratch *sc = reinterpret_cast<Scratch *>(reinterpret_cast<uint8_t *>(&smg->st) - (852 + 592) * 2);
      uint16_t *lit16 = reinterpret_cast<uint16_t *>(smg->lut) + 852;
      uint16_t *dist16 = reinterpret_cast<uint16_t *>(smg->lut + kDistBase) + 592;
      if (type == 1) fixed_tables(sc, &lit, &dist, lane, lit16, dist16);
      else {
        rc = dynamic_header<C>(ur, sc, &lit, &dist, lane, lit16, dist16);
        bp = ur.bp;
      }
      if (rc != MD_OK) break;
      // widen in place, 64 entries per step: a step reads its 16-bit entries before it writes the
      // 32-bit ones, and never overwrites a later step's unread entries (4c + 256 <= half + 2c + 128)
      {
        const u16a *a16 = reinterpret_cast<const u16a *>(lit16);
        u32a *a32 = reinterpret_cast<u32a *>(smg->lut);
        for (uint32_t c = 0; c < 852; c += kWave) {
          const uint32_t i = c + lane;
          const uint32_t e = i < 852 ? a16[i] : 0;
          const uint32_t f = fat_lit(e);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (i < 852) a32[i] = f;
        }
        const u16a *b16 = reinterpret_cast<const u16a *>(dist16);
        u32a *b32 = reinterpret_cast<u32a *>(smg->lut + kDistBase);
        for (uint32_t c = 0; c < 592; c += kWave) {
          const uint32_t i = c + lane;
          const uint32_t e = i < 592 ? b16[i] : 0;
          const uint32_t f = fat_dist(e);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (i < 592) b32[i] = f;
        }
      }
      const uint32_t lmask = uni(lit.mask), lroot = uni(lit.root), dmask = uni(dist.mask), droot = uni(dist.root);
      // rounds of this Huffman block
      for (;;) {
        if (rec + 2 > log_cap) {
          rc = kStatusLogFull;
          break;
        }
        uint8_t *r = lg + (size_t)rec * R::BYTES;
        uint32_t *mrec = reinterpret_cast<uint32_t *>(r + R::OFF_MREC);
        uint8_t *lits = r + R::OFF_LITS;
        inp.ensure(bp >> 3);
        LaneState ls;
        ls.start = bp + lane * C::S;
        const uint32_t limit = bp + (lane + 1) * C::S;
        decode_pass<C>(inp, lut, mrec, lits, lane, total_bits, lmask, lroot, dmask, droot, true, limit, ls, pf);
        for (uint32_t it = 0; it < C::PASSES; it++) {
          const uint32_t pe = __shfl_up(ls.end, 1), ps = __shfl_up(ls.stop, 1);
          const bool redo = lane > 0 && ps == 0 && pe != ls.start;
          if (!__any(redo)) break;
          if (redo) ls.start = pe;
          decode_pass<C>(inp, lut, mrec, lits, lane, total_bits, lmask, lroot, dmask, droot, redo, limit, ls, pf);
        }
        uint32_t nvalid;
        {
          const uint32_t pe = __shfl_up(ls.end, 1), ps = __shfl_up(ls.stop, 1);
          const uint64_t bad = __ballot(lane > 0 && (ps != 0 || pe != ls.start));
          nvalid = bad ? (uint32_t)__builtin_ctzll(bad) : 64;
        }
        const bool v = lane < nvalid;
        reinterpret_cast<uint32_t *>(r + R::OFF_NLM)[lane] = v ? (ls.nlit | (ls.nmat << 8)) : 0u;
        reinterpret_cast<uint32_t *>(r + R::OFF_NB)[lane] = v ? ls.nb : 0u;
        reinterpret_cast<uint64_t *>(r + R::OFF_TMASK)[lane] = v ? ls.tmask : 0ull;
        put_hdr(r, lane, REC_ROUND, nvalid, wave_max(v ? ls.nlit : 0), wave_max(v ? ls.nmat : 0));
        rec++;
        const uint32_t lstop = rdlane(ls.stop, nvalid - 1);
        bp = rdlane(ls.end, nvalid - 1);
        if (lstop == kStopEob) break;
        if (lstop != 0) {
          rc = (int)lstop;
          break;
        }
      }
    }
  }
  put_hdr(lg + (size_t)rec * R::BYTES, lane, REC_END, (uint32_t)rc, (bp + 7) >> 3, 0);
}

// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(kWave) void resolve_kernel(
    int format, uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *out, const uint64_t *__restrict__ out_off,
    const uint64_t *__restrict__ out_cap, uint64_t *__restrict__ out_len,
    uint64_t *__restrict__ consumed, int32_t *__restrict__ status, uint32_t *__restrict__ checksum,
    const uint8_t *__restrict__ log, uint64_t log_stride) {
  extern __shared__ __align__(16) uint8_t smem_raw[];
  Smem2<C> *smg = reinterpret_cast<Smem2<C> *>(smem_raw);
  Prof<false> pf;
  const uint32_t lane = threadIdx.x, sid = blockIdx.x;
  if (sid >= n) return;
  using R = Rec<C>;
  const uint8_t *lg = log + (size_t)sid * log_stride;
  const uint8_t *src = in + in_off[sid];
  uint64_t cap64 = out_cap[sid];

  Sink sk;
  sk.stage = (lds_u8 *)smg->stage;
  sk.g = out + out_off[sid];
  sk.cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;
  sk.pos = 0;
  sk.lane = lane;
  sk.a = 1;
  sk.b = 0;
  sk.want_adler = (checksum != nullptr) || format == MD_FORMAT_ZLIB;
  lds_u32 *mrec = (lds_u32 *)smg->mrec;
  lds_u8 *lits = (lds_u8 *)smg->lits;

  int rc = MD_OK;
  uint32_t used = 0;
  for (uint32_t rec = 0;; rec++) {
    const uint8_t *r = lg + (size_t)rec * R::BYTES;
    const uint4 h = *reinterpret_cast<const uint4 *>(r);
    const uint32_t kind = uni(h.x);
    if (kind == REC_END) {
      rc = (int)uni(h.y);
      used = uni(h.z);
      break;
    }
    if (kind == REC_STORED) {
      const uint32_t so = uni(h.y), len = uni(h.z);
      if (len > sk.cap - sk.pos) {
        rc = MD_UNEXPECTED_END_OF_OUTPUT;
        break;
      }
      const uint8_t *q = src + so;
      uint32_t left = len;
      while (left) {
        const uint32_t seg = left < C::STAGE - 16 ? left : C::STAGE - 16;
        const uint32_t s0 = sk.pos - sk.sbase();
        for (uint32_t j = lane; j < seg; j += kWave) sk.stage[s0 + j] = q[j];
        sk.flush(seg);
        q += seg;
        left -= seg;
      }
      continue;
    }
    // ---- a round: token rows HBM -> LDS (coalesced), then place them
    const uint32_t nvalid = uni(h.y), maxlit = uni(h.z), maxmat = uni(h.w);
    LaneState ls;
    {
      const uint32_t nlm = reinterpret_cast<const uint32_t *>(r + R::OFF_NLM)[lane];
      ls.nlit = nlm & 0xff;
      ls.nmat = nlm >> 8;
      ls.nb = reinterpret_cast<const uint32_t *>(r + R::OFF_NB)[lane];
      ls.tmask = reinterpret_cast<const uint64_t *>(r + R::OFF_TMASK)[lane];
      ls.start = ls.end = ls.stop = 0;
    }
    {
      const uint32_t *gm = reinterpret_cast<const uint32_t *>(r + R::OFF_MREC);
      for (uint32_t m = 0; m < maxmat; m += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = m + u < maxmat ? gm[(m + u) * kWave + lane] : 0u;
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (m + u < maxmat) mrec[(m + u) * kWave + lane] = v[u];
      }
      // literal rows are 64 bytes: 16 lanes x 4 bytes move one row, 4 rows per wave step
      const uint32_t *gl = reinterpret_cast<const uint32_t *>(r + R::OFF_LITS);
      lds_u32 *ll = (lds_u32 *)lits;
      const uint32_t words = maxlit * 16;
      for (uint32_t w = lane; w < words; w += kWave * 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = w + u * kWave < words ? gl[w + u * kWave] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (w + u * kWave < words) ll[w + u * kWave] = v[u];
      }
    }
    uint32_t lo = 0;
    while (lo < nvalid && rc == MD_OK) {
      uint32_t acc, emitted;
      rc = emit_round<C>(mrec, lits, (lds_u8 *)smg->owner, sk, lane, lo, nvalid, ls, &acc, &emitted, pf);
      sk.flush(emitted);
      lo = acc;
    }
    if (rc != MD_OK) break;
  }
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

}  // namespace v4
}  // namespace md

using SplitCfg = md::v4::Cfg<256, 36, 16, 64, 4096, 5, 6144>;

extern "C" size_t md_inflate_log_record_bytes(void) { return md::v4::Rec<SplitCfg>::BYTES; }

extern "C" int md_launch_inflate_split(int format, uint32_t n, const uint8_t *in, const uint64_t *in_off,
                                       const uint64_t *in_len, uint8_t *out, const uint64_t *out_off,
                                       const uint64_t *out_cap, uint64_t *out_len, uint64_t *consumed,
                                       int32_t *status, uint32_t *checksum, uint8_t *log,
                                       uint32_t log_cap, hipStream_t stream) {
  if (n == 0) return 0;
  using namespace md::v4;
  const uint64_t stride = (uint64_t)log_cap * Rec<SplitCfg>::BYTES;
  hipLaunchKernelGGL((decode_kernel<SplitCfg>), dim3(n), dim3(md::kWave), sizeof(Smem1<SplitCfg>), stream, format,
                     n, in, in_off, in_len, log, stride, log_cap);
  hipLaunchKernelGGL((resolve_kernel<SplitCfg>), dim3(n), dim3(md::kWave), sizeof(Smem2<SplitCfg>), stream, format,
                     n, in, in_off, in_len, out, out_off, out_cap, out_len, consumed, status, checksum, log, stride);
  return (int)hipGetLastError();
}
