// inflate_v4.hip — lane-parallel batched RFC1951 inflate for gfx950 (MI355X).
//
// One independent stream per wavefront (one 64-lane workgroup per stream, ~15 KiB
// of LDS, so 10 streams are resident per CU).  Inside the wavefront the 64 lanes
// decode 64 consecutive S-bit zones of the compressed block at once:
//
//   A1   lane i decodes tokens from bit bp + i*S up to bp + (i+1)*S.  Only lane 0
//        starts on a token boundary; the others are speculative.
//   A2   every lane compares its start with its left neighbour's end and re-decodes
//        from there until the chain start_i == end_{i-1} holds.  Huffman streams
//        re-synchronise (p ~ 0.9 inside a 256-bit zone), so two more passes settle
//        almost every lane; the consistent prefix of lanes is accepted.  A chain
//        of token boundaries that starts on a known boundary IS the serial decode
//        of the reference loop (`inflate`, lib/de.ml:1667-1712).
//   B    a wave prefix-sum of the lanes' output sizes gives every token its output
//        position.  The round's output is assembled in an LDS staging buffer:
//        literals first; matches whose source is older than the round read it from
//        already-flushed output (L2/HBM, 16 bytes per load, 4 matches in flight per
//        lane); matches into the round itself resolve lane-parallel inside LDS with
//        exact dependency tracking.
//   C    the round is flushed with 16-byte coalesced stores and folded into the
//        Adler-32 in the same pass (WInf.update / tail, lib/de.ml:453-455, 499-505).
//
// The window is the output buffer itself (De.Inf.Ns semantics, lib/de.ml:1534):
// only the current round lives on chip; the L2 (4 MiB per XCD) serves the far reads.
// Error behaviour keeps the oracle's order: the first failing token in stream
// order decides the status and everything before it is written.
// Block headers, LUT construction (lib/de.ml:523-638, 1733-1793), stored blocks
// (lib/de.ml:1613-1627) and the zlib frame (lib/zl.ml:400-417) run wave-uniform
// between rounds.
#include "inflate_lane.hpp"

namespace md {
namespace v4 {

template <class C, bool PROF>
__global__ __launch_bounds__(kWave) void inflate_v4_kernel(
    int format, uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *out, const uint64_t *__restrict__ out_off,
    const uint64_t *__restrict__ out_cap, uint64_t *__restrict__ out_len,
    uint64_t *__restrict__ consumed, int32_t *__restrict__ status, uint32_t *__restrict__ checksum,
    uint64_t *__restrict__ dbg, int only_status) {
  extern __shared__ __align__(16) uint8_t smem_raw[];
  Smem<C> *smg = reinterpret_cast<Smem<C> *>(smem_raw);
  Prof<PROF> pf;
  pf.init();
  const uint32_t lane = threadIdx.x;
  const uint32_t sid = blockIdx.x;
  if (sid >= n) return;
  if (only_status >= 0 && status[sid] != only_status) return;  // redo pass of the split path

  const uint8_t *src = in + in_off[sid];
  uint64_t slen64 = in_len[sid], cap64 = out_cap[sid];
  uint32_t slen = slen64 > 0x1ffffff0ull ? 0x1ffffff0u : (uint32_t)slen64;
  uint32_t cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;

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

  Sink sk;
  sk.stage = (lds_u8 *)smg->u.t.stage;
  sk.g = out + out_off[sid];
  sk.cap = cap;
  sk.pos = 0;
  sk.lane = lane;
  sk.a = 1;
  sk.b = 0;
  sk.want_adler = (checksum != nullptr) || format == MD_FORMAT_ZLIB;

  Input<C> inp;
  inp.p = src + body_off;
  inp.nbytes = body_len;
  inp.lane = lane;
  inp.ring = (lds_u32 *)smg->inring;
  inp.reset(0);
  const uint32_t total_bits = body_len * 8;
  uint32_t bp = 0;

  if (rc == MD_OK) {
    bool last = false;
    while (!last && rc == MD_OK) {
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
      if (type == 0) {
        // flat, lib/de.ml:1613-1627
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
        else if (len > sk.cap - sk.pos) rc = MD_UNEXPECTED_END_OF_OUTPUT;
        else {
          const uint8_t *q = inp.p + p;
          uint32_t left = len;
          while (left) {
            const uint32_t seg = left < C::STAGE - 16 ? left : C::STAGE - 16;
            const uint32_t s0 = sk.pos - sk.sbase();
            for (uint32_t j = lane; j < seg; j += kWave) sk.stage[s0 + j] = q[j];
            sk.flush(seg);
            q += seg;
            left -= seg;
          }
          p += len;
          bp = p * 8;
          inp.reset(p);
        }
      } else if (type == 3) {
        rc = MD_INVALID_KIND_OF_BLOCK;
      } else {
        Lut lit, dist;
        if (type == 1) fixed_tables(&smg->u.sc, &lit, &dist, lane);
        else {
          rc = dynamic_header<C>(ur, &smg->u.sc, &lit, &dist, lane);
          bp = ur.bp;
        }
        if (rc == MD_OK) {
          // fat LUTs (base / extra bits / type per entry) from the packed 16-bit ones
          for (uint32_t i = lane; i < 852; i += kWave) smg->lut[i] = fat_lit(smg->u.sc.lit[i]);
          for (uint32_t i = lane; i < 592; i += kWave) smg->lut[kDistBase + i] = fat_dist(smg->u.sc.dist[i]);
        }
        pf.tick(P_HEADER);
        if (rc == MD_OK)
          rc = inflate_block<C>(smg, inp, sk, uni(lit.mask), uni(lit.root), uni(dist.mask), uni(dist.root),
                                lane, total_bits, &bp, pf);
      }
    }
  }
  uint32_t used = (bp + 7) >> 3;  // i_pos - (bits lsr 3), lib/de.ml:1805
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
  if constexpr (PROF) {
    if (lane == 0 && sid == 0 && dbg) {
      for (int i = 0; i < P_COUNT; i++) dbg[i] = pf.acc[i];
      for (int i = 0; i < C_COUNT; i++) dbg[P_COUNT + i] = pf.cnt[i];
    }
  }
}

}  // namespace v4
}  // namespace md

extern "C" int md_launch_inflate_v2(int variant, int format, uint32_t n, const uint8_t *in,
                                    const uint64_t *in_off, const uint64_t *in_len, uint8_t *out,
                                    const uint64_t *out_off, const uint64_t *out_cap,
                                    uint64_t *out_len, uint64_t *consumed, int32_t *status,
                                    uint32_t *checksum, uint64_t *dbg, int only_status, hipStream_t stream) {
  if (n == 0) return 0;
  dim3 grid(n), block(md::kWave);
#define MD_LAUNCH_V4(CFG)                                                                        \
  do {                                                                                           \
    if (dbg)                                                                                     \
      hipLaunchKernelGGL((md::v4::inflate_v4_kernel<CFG, true>), grid, block,                    \
                         sizeof(md::v4::Smem<CFG>), stream, format, n, in, in_off, in_len, out,  \
                         out_off, out_cap, out_len, consumed, status, checksum, dbg, only_status); \
    else                                                                                         \
      hipLaunchKernelGGL((md::v4::inflate_v4_kernel<CFG, false>), grid, block,                   \
                         sizeof(md::v4::Smem<CFG>), stream, format, n, in, in_off, in_len, out,  \
                         out_off, out_cap, out_len, consumed, status, checksum, dbg, only_status); \
  } while (0)
  //                      S   LMAX MMAX KMAX IN_BYTES PASSES STAGE
  using A = md::v4::Cfg<256, 40, 16, 72, 4096, 3, 6144>;
  using B = md::v4::Cfg<256, 40, 16, 72, 4096, 2, 6144>;
  using Cc = md::v4::Cfg<320, 44, 18, 88, 4096, 2, 7168>;
  switch (variant) {
  case 0: MD_LAUNCH_V4(A); break;
  case 1: MD_LAUNCH_V4(B); break;
  case 2: MD_LAUNCH_V4(Cc); break;
  default: return -1;
  }
#undef MD_LAUNCH_V4
  return (int)hipGetLastError();
}
