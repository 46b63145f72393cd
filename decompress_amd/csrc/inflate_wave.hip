// inflate_wave.hip — batched RFC1951 inflate for gfx950 (MI355X): one stream per wavefront, one kernel.
//
// The 64 lanes decode 64 consecutive S-bit zones of the compressed block per round:
//
//   sync   Light passes that only FIND TOKEN BOUNDARIES.  Pass 0: lane i walks the tokens from bit
//          bp + i*S until it crosses bp + (i+1)*S (only lane 0 starts on a boundary, the others
//          are speculative).  Pass k >= 1: a lane whose left neighbour ended somewhere else than
//          where it started walks again from there and now also counts the bytes its tokens will
//          produce.  Huffman streams re-synchronise (p ~ 0.9 inside a 256-bit zone), so the chain
//          start_i == end_{i-1} settles after ~3 passes; the consistent prefix of lanes is
//          accepted.  A chain of token boundaries that starts on a known boundary IS the serial
//          decode of the reference loop (`inflate`, lib/de.ml:1667-1712).  A walk step is one
//          32-bit peek of the LDS input ring and one 32-bit LUT entry that carries the bits to
//          skip and the descriptor (index width, base) of the table the NEXT step indexes, so
//          the step has no per-type branches or selects.
//   decode One pass over the accepted lanes from their validated starts.  A wave prefix sum of
//          the byte counts has given every lane its output position, so literals go straight
//          to their final place in the LDS staging buffer and matches become (gap, length,
//          distance) records; position-dependent checks (Invalid_distance,
//          Unexpected_end_of_output) happen here at the token, in stream order.
//   copy   Matches whose source is older than the round read it from already flushed output
//          (L2/HBM, 4 in flight per lane); matches into the round itself resolve lane-parallel
//          in LDS with exact dependency tracking.
//   flush  16-byte coalesced stores, Adler-32 folded in the same pass with v_dot4_u32_u8
//          (WInf.update / tail, lib/de.ml:453-455, 499-505).
//
// The window is the output buffer itself (De.Inf.Ns semantics, lib/de.ml:1534).  Block headers,
// LUT construction (lib/de.ml:523-638, 1733-1793), stored blocks (lib/de.ml:1613-1627) and the
// zlib frame (lib/zl.ml:400-417) run wave-uniform between rounds.  The first failing token in
// stream order decides the status and everything before it is written (oracle/de_inflate.c).
#include "inflate_lane.hpp"

namespace md {
namespace v5 {
using namespace md::v4;

// LUT entry: codelen[3:0] | xb[7:4] | val9[16:8] | next.nbits[20:17] | next.tb[31:21]
//   codelen  bits this step consumes for the code (a LINK entry consumes the root bits, the
//            sub-table entry behind it the rest of the code)
//   xb       extra bits that follow the code
//   val9     literal byte | length base (3..258) | distance symbol (0..29, 30/31 invalid) | 511 = LINK
//   next     the table the following step indexes: lit root, distance root, a sub-table, or
//            one of the two self-looping STOP entries (end of block / empty distance slot)
constexpr uint32_t kDistB = 852;
constexpr uint32_t kStopEobI = 852 + 592;
constexpr uint32_t kStopBadI = kStopEobI + 1;
constexpr uint32_t kLutWords = kStopEobI + 4;
constexpr uint32_t kLinkVal = 511;
constexpr uint32_t kStEob = 100, kStTrunc = 101;  // lane stop reasons of the decode pass; < 100 = MD_* status

__device__ __forceinline__ uint32_t mk_entry(uint32_t codelen, uint32_t xb, uint32_t val9, uint32_t nbits, uint32_t tb) {
  return codelen | (xb << 4) | (val9 << 8) | (nbits << 17) | (tb << 21);
}
// 16-bit builder entry -> walk entry.  `sub` = the entry sits in a sub-table (its code length includes the root bits).
__device__ __forceinline__ uint32_t walk_lit(uint32_t e, bool sub, uint32_t lroot, uint32_t droot) {
  if (e & kLink) return mk_entry(lroot, 0, kLinkVal, (e >> 10) & 15, e & 1023);
  const uint32_t len = ((e >> 9) & 15) - (sub ? lroot : 0u), sym = e & 511;
  if (sym < 256) return mk_entry(len, 0, sym, lroot, 0);
  if (sym == 256) return mk_entry(len, 0, 0, 0, kStopEobI);
  const uint32_t l = (sym - 257) & 31;  // lib/de.ml:293-311 (+3 folded in; 29,30 -> 3)
  const uint32_t xb = (l >= 8 && l < 28) ? (l - 4) >> 2 : 0;
  const uint32_t base = (l < 8 ? l : l < 28 ? (4 + (l & 3)) << xb : l == 28 ? 255 : 0) + 3;
  return mk_entry(len, xb, base, droot, kDistB);
}
__device__ __forceinline__ uint32_t walk_dist(uint32_t e, bool sub, uint32_t lroot, uint32_t droot) {
  if (e == kBad) return mk_entry(0, 0, 0, 0, kStopBadI);
  if (e & kLink) return mk_entry(droot, 0, kLinkVal, (e >> 10) & 15, kDistB + (e & 1023));
  const uint32_t len = ((e >> 9) & 15) - (sub ? droot : 0u), dv = e & 31;
  const uint32_t xb = (dv >= 4 && dv < 30) ? (dv - 2) >> 1 : 0;  // lib/de.ml:313-325
  return mk_entry(len, xb, dv, lroot, 0);
}

template <class C>
struct Smem {
  uint32_t inring[C::IN_WORDS + 4];
  uint32_t lut[kLutWords];
  union U {
    Scratch sc;  // packed 16-bit LUTs + construction scratch: live only while a header is parsed
    struct T {
      uint32_t mrec[C::MMAX * kWave];            // match records, [m][lane]
      alignas(16) uint8_t stage[C::STAGE + 16];  // one round of output
      uint8_t owner[C::STAGE / 32 + 8];          // producer lane of each 32-byte staging block
    } t;
  } u;
};

// ---------------------------------------------------------------------------
// One token-boundary walk of this lane's zone [start, limit).  COUNT adds the bytes the tokens produce.
// A divergent per-lane loop: finished lanes leave the exec mask, the wave leaves when it is empty.
__device__ __forceinline__ uint32_t lut_at(const lds_u32 *lut, uint32_t tb, uint32_t idx) {
  return *reinterpret_cast<const lds_u32 *>(reinterpret_cast<const lds_u8 *>(lut) + ((tb + idx) << 2));
}
template <class C, bool COUNT, class PF>
__device__ __forceinline__ void sync_pass(const Input<C> &in, const lds_u32 *lut, uint32_t lroot, bool go,
                                          uint32_t start, uint32_t limit, uint32_t &end, uint32_t &stop,
                                          uint32_t &nb, PF &pf) {
  if (go) {
    uint32_t p = start, tb = 0, nbits = lroot, cnt = 0, slot = 0;
    // a step that starts a token (tb == 0) is taken while the zone and the slot budget last
    while ((tb < kStopEobI) & ((tb != 0) | ((p < limit) & (slot < C::KMAX)))) {
      const uint32_t w = in.peek(p);
      const uint32_t e = lut_at(lut, tb, __builtin_amdgcn_ubfe(w, 0, nbits));
      const uint32_t codelen = e & 15, xb = (e >> 4) & 15, ntb = e >> 21;
      if (COUNT) {
        const uint32_t len1 = ((e >> 8) & 511) + __builtin_amdgcn_ubfe(w, codelen, xb) - 1;
        cnt += (tb == 0 ? 1u : 0u) + (ntb == kDistB ? len1 : 0u);
      }
      p += codelen + xb;
      nbits = (e >> 17) & 15;
      tb = ntb;
      slot++;
    }
    end = p;
    stop = tb >= kStopEobI ? tb : 0u;
    if (COUNT) nb = cnt - (tb == kStopEobI ? 1u : 0u);  // the end-of-block code was counted as a token
  }
}

// The decode pass: the same walk from a validated start, producing output.
struct LaneOut {
  uint32_t endp;   // bit after the last token taken (a token boundary)
  uint32_t stopc;  // 0 = zone done, kStEob, kStTrunc (round capacity), else MD_* status of the failing token
  uint32_t bytes;  // bytes produced (up to the failing token)
  uint32_t nm;     // match records written
};
template <class C, class PF>
__device__ __forceinline__ void emit_pass(const Input<C> &in, const lds_u32 *lut, uint32_t lroot, lds_u32 *mrec,
                                            lds_u8 *stage, uint32_t lane, uint32_t total_bits, bool go,
                                            uint32_t start, uint32_t limit, uint32_t q0, uint32_t rb,
                                            uint32_t R0, uint32_t cap, LaneOut &lo, PF &pf) {
  uint32_t p = start, ptok = start, tb = 0, nbits = lroot;
  uint32_t q = q0, gap = 0, nm = 0, mlen = 0, stopc = 0;
  const uint32_t qlim = rb + C::STAGE - 16;  // the staging buffer holds output positions [rb, qlim)
  if (go) {
  uint32_t slot = 0;
  while ((stopc == 0) & ((tb != 0) | ((p < limit) & (slot < C::KMAX)))) {
    ptok = tb == 0 ? p : ptok;
    const uint32_t w = in.peek(p);
    const uint32_t e = lut_at(lut, tb, __builtin_amdgcn_ubfe(w, 0, nbits));
    const uint32_t codelen = e & 15, xb = (e >> 4) & 15, val9 = (e >> 8) & 511, ntb = e >> 21;
    const uint32_t x = __builtin_amdgcn_ubfe(w, codelen, xb);
    const uint32_t pn = p + codelen + xb;
    const bool in_dist = tb >= kDistB, is_link = val9 == kLinkVal;
    // oracle order: empty distance slot (D2), then end of input (D1), then distance code 30/31
    uint32_t st = ntb == kStopBadI                       ? (uint32_t)MD_INVALID_DISTANCE_CODE
                  : (!is_link && pn > total_bits)        ? (uint32_t)MD_UNEXPECTED_END_OF_INPUT
                  : (in_dist && !is_link && val9 >= 30u) ? (uint32_t)MD_INVALID_DISTANCE_CODE
                                                         : 0u;
    if (!st && !is_link) {
      if (in_dist) {  // the distance completes a match
        const uint32_t d = (val9 < 4 ? val9 + 1 : (((val9 & 1) | 2) << xb) + 1) + x;
        const uint32_t lim = q < 32768u ? q : 32768u;
        if (d > lim) st = MD_INVALID_DISTANCE;
        else if (mlen > cap - q) st = MD_UNEXPECTED_END_OF_OUTPUT;
        else if (q + mlen > qlim || nm == C::MMAX) st = kStTrunc;
        else {
          mrec[nm * kWave + lane] = (gap << 24) | ((mlen - 3) << 16) | ((q - d + mlen > R0) ? kNear : 0u) | (d - 1);
          nm++;
          gap = 0;
          q += mlen;
        }
      } else if (ntb == kDistB) {
        mlen = val9 + x;
      } else if (ntb == kStopEobI) {
        st = kStEob;
      } else {  // literal
        if (q >= cap) st = MD_UNEXPECTED_END_OF_OUTPUT;
        else if (q >= qlim) st = kStTrunc;
        else {
          stage[q - rb] = (uint8_t)val9;
          q++;
          gap++;
        }
      }
    }
    stopc = st;
    p = (st == 0 || st == kStEob) ? pn : p;
    nbits = (e >> 17) & 15;
    tb = ntb;
    slot++;
  }
  }
  if (go) {
    lo.endp = (stopc == 0 || stopc == kStEob) ? p : ptok;
    lo.stopc = stopc;
    lo.bytes = q - q0;
    lo.nm = nm;
  }
}

// ---------------------------------------------------------------------------
struct Sink5 {
  lds_u8 *stage;
  uint8_t *g;
  uint32_t cap;
  uint32_t pos;  // bytes produced and flushed
  uint32_t lane;
  uint32_t a, b;
  bool want_adler;
  __device__ __forceinline__ uint32_t sbase() const { return pos & ~15u; }  // staging index of position x is x - sbase()

  // write stage[...] for positions [pos, pos+total) to HBM, fold Adler-32, advance pos
  __device__ __forceinline__ void flush(uint32_t total) {
    const uint32_t rb = sbase();
    const uint32_t endp = pos + total;
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t ps = rb; ps < endp; ps += 1024) {
      const uint32_t cpos = ps + lane * 16;
      const uint32_t lo = cpos > pos ? cpos : pos;
      const uint32_t hi = cpos + 16 < endp ? cpos + 16 : endp;
      if (lo < hi) {
        const lds_u32 *sp = reinterpret_cast<const lds_u32 *>(stage + (cpos - rb));  // 16-byte aligned
        const uint32_t w[4] = {sp[0], sp[1], sp[2], sp[3]};
        if (hi - lo == 16) {
          const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
          __builtin_memcpy(g + cpos, &v, 16);
          if (want_adler) {
            // sum d_k and sum k*d_k over the 16 bytes
            uint32_t t1 = __builtin_amdgcn_udot4(w[0], 0x01010101u, 0u, false);
            t1 = __builtin_amdgcn_udot4(w[1], 0x01010101u, t1, false);
            t1 = __builtin_amdgcn_udot4(w[2], 0x01010101u, t1, false);
            t1 = __builtin_amdgcn_udot4(w[3], 0x01010101u, t1, false);
            uint32_t tk = __builtin_amdgcn_udot4(w[0], 0x03020100u, 0u, false);
            tk = __builtin_amdgcn_udot4(w[1], 0x07060504u, tk, false);
            tk = __builtin_amdgcn_udot4(w[2], 0x0b0a0908u, tk, false);
            tk = __builtin_amdgcn_udot4(w[3], 0x0f0e0d0cu, tk, false);
            s1 += t1;
            s2 += (endp - cpos) * t1 - tk;
          }
        } else {
          for (uint32_t x = lo; x < hi; x++) {
            const uint32_t k = x - cpos;
            const uint32_t d = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
            g[x] = (uint8_t)d;
            s1 += d;
            s2 += (endp - x) * d;
          }
        }
      }
    }
    if (want_adler) {  // per lane: <= 96 bytes x 255 x 6160 < 2^32
      s1 = wave_sum(s1);
      s2 = wave_sum(s2 % 65521u);
      b = (b + total * a + s2) % 65521u;
      a = (a + s1) % 65521u;
    }
    pos = endp;
  }
};

// ---------------------------------------------------------------------------
// Far and near matches of a round (records of lanes [0, nvalid) with nm records each).
template <class C, class PF>
__device__ __forceinline__ void copy_matches(lds_u32 *mrec, lds_u8 *owner, const Sink5 &sk, uint32_t lane,
                                             uint32_t q0, uint32_t nm, PF &pf) {
  lds_u8 *stage = sk.stage;
  const uint32_t R0 = sk.pos, rb = sk.sbase(), cap = sk.cap;
  const uint8_t *g = sk.g;
  // (b) far matches: the whole source is older than this round — final in HBM/L2, visible once the flush
  //     of earlier rounds has been waited for; d >= ml.  The loads of up to 4 matches per lane are in flight together.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  {
    uint32_t qq = q0, m = 0;
    while (__any(m < nm)) {
      uint64_t v0[4], v1[4];
      uint32_t dq[4], dl[4], ds[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        dl[u] = 0;
        dq[u] = 0;
        ds[u] = 0;
        v0[u] = 0;
        v1[u] = 0;
        while (m < nm) {  // advance to this lane's next far record
          const uint32_t tk = mrec[m * kWave + lane];
          const uint32_t d = (tk & 0x7fff) + 1, ml = ((tk >> 16) & 0xff) + 3;
          qq += tk >> 24;
          const uint32_t cq = qq;
          qq += ml;
          m++;
          if (tk & kNear) continue;
          v0[u] = out_ld_guard(g, cq - d, ml, cap);
          if (ml > 8) v1[u] = out_ld_guard(g, cq - d + 8, ml - 8, cap);
          dq[u] = cq;
          dl[u] = ml;
          ds[u] = cq - d;
          break;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (dl[u]) {
          lds_u8 *dd = stage + (dq[u] - rb);
          const uint32_t ml = dl[u];
          lds_st(dd, v0[u], ml);
          if (ml > 8) lds_st(dd + 8, v1[u], ml - 8);
          for (uint32_t j = 16; j < ml; j += 8)  // long far match: stream the rest (rare)
            lds_st(dd + j, out_ld_guard(g, ds[u] + j, ml - j, cap), ml - j);
        }
      }
    }
  }
  pf.tick(P_FAR);
  // (c) near matches: the source reaches into this round's staging buffer.  `done` is the position of this
  //     lane's first unresolved match: everything the lane produces before it is final.  A match may run when
  //     the first unresolved lane at or after the producer of its source is itself, or is past the source's end.
  {
    uint32_t m = 0, qq = q0;
    uint32_t d = 0, ml = 0, qm = 0, ja = 0;
    bool pending = false;
    auto advance = [&]() {
      pending = false;
      while (m < nm) {
        const uint32_t tk = mrec[m * kWave + lane];
        d = (tk & 0x7fff) + 1;
        ml = ((tk >> 16) & 0xff) + 3;
        qm = qq + (tk >> 24);
        m++;
        qq = qm + ml;
        if (tk & kNear) {
          const uint32_t src = qm - d;
          ja = src >= rb ? owner[(src - rb) >> 5] : 0;
          pending = true;
          break;
        }
      }
    };
    advance();
    for (;;) {
      const uint64_t pm = __ballot(pending);
      if (!pm) break;
      pf.count(C_NEAR_IT);
      const uint32_t done = pending ? qm : 0xffffffffu;
      uint32_t f = lane;
      if (pending) f = ja + (uint32_t)__builtin_ctzll(pm >> ja);  // bit `lane` is set: pm >> ja != 0
      const uint32_t df = __shfl(done, f);
      if (pending && (f >= lane || df >= qm - d + ml)) {
        const uint32_t src = qm - d;
        lds_u8 *dd = stage + (qm - rb);
        if (src >= R0) {
          copy_near(dd, stage + (src - rb), ml, d);
        } else {
          // straddles the round start: the first bytes come from HBM
          const uint32_t ng = R0 - src;
          for (uint32_t j = 0; j < ng; j++) dd[j] = (uint8_t)out_ld8(g + src + j);
          copy_near(dd + ng, stage + (R0 - rb), ml - ng, d);
        }
        advance();
      }
    }
  }
  pf.tick(P_NEAR);
}

// ---------------------------------------------------------------------------
// All rounds of one Huffman block.  On return *bp_io is the bit after the EOB.
template <class C, class PF>
__device__ __forceinline__ int inflate_block(Smem<C> *smg, Input<C> &in, Sink5 &sk, uint32_t lroot, uint32_t lane,
                                             uint32_t total_bits, uint32_t *bp_io, PF &pf) {
  uint32_t bp = *bp_io;
  const lds_u32 *lut = (const lds_u32 *)smg->lut;
  lds_u32 *mrec = (lds_u32 *)smg->u.t.mrec;
  lds_u8 *owner = (lds_u8 *)smg->u.t.owner;
  for (;;) {
    in.ensure(bp >> 3);
    pf.tick(P_ENSURE);
    pf.count(C_ROUNDS);
    pf.count(C_PASSES);
    uint32_t start = bp + lane * C::S, end = 0, stop = 0, nb = 0;
    const uint32_t limit = bp + (lane + 1) * C::S;
    sync_pass<C, false>(in, lut, lroot, true, start, limit, end, stop, nb, pf);
    pf.tick(P_DECODE1);
    bool counted = false;
    for (uint32_t it = 0; it < C::PASSES; it++) {
      const uint32_t pe = __shfl_up(end, 1), ps = __shfl_up(stop, 1);
      const bool redo = lane == 0 ? !counted : (ps == 0 && (pe != start || !counted));
      if (!__any(redo)) break;
      if (redo && lane > 0) start = pe;
      pf.count(C_PASSES);
      sync_pass<C, true>(in, lut, lroot, redo, start, limit, end, stop, nb, pf);
      counted = counted || redo;
    }
    pf.tick(P_DECODE2);
    uint32_t nvalid;
    {
      const uint32_t pe = __shfl_up(end, 1), ps = __shfl_up(stop, 1);
      const uint64_t bad = __ballot(!counted || (lane > 0 && (ps != 0 || pe != start)));
      nvalid = bad ? (uint32_t)__builtin_ctzll(bad) : 64;  // lane 0 is always counted: nvalid >= 1
    }
    const uint32_t R0 = sk.pos, rb = sk.sbase();
    uint32_t mynb = lane < nvalid ? nb : 0;
    const uint32_t off = wave_excl_scan(mynb, lane);
    {  // staging capacity: keep the largest prefix of lanes that fits (a lone first lane truncates itself)
      const uint64_t fits = __ballot((R0 - rb) + off + mynb <= C::STAGE - 16);
      const uint32_t nfit = fits == ~0ull ? 64 : (uint32_t)__builtin_ctzll(~fits);
      if (nfit < nvalid) nvalid = nfit ? nfit : 1;
    }
    const bool mine = lane < nvalid;
    if (!mine) mynb = 0;
    const uint32_t q0 = R0 + off;
    // owner table: the lane that produces the first byte of every 32-byte staging block — a lower bound of
    // the producer of any byte in that block
    if (mynb) {
      const uint32_t b0 = (q0 - rb + 31) >> 5, b1 = (q0 + mynb - 1 - rb) >> 5;
      const uint32_t b1c = b1 < C::STAGE / 32 + 7 ? b1 : C::STAGE / 32 + 7;
      for (uint32_t bb = b0; bb <= b1c; bb++) owner[bb] = (uint8_t)lane;
    }
    if (lane == 0) owner[0] = 0;
    LaneOut lo;
    lo.endp = end;
    lo.stopc = 0;
    lo.bytes = 0;
    lo.nm = 0;
    emit_pass<C>(in, lut, lroot, mrec, sk.stage, lane, total_bits, mine, start, limit, q0, rb, R0, sk.cap, lo, pf);
    pf.tick(P_EMIT_A);
    // the first stopped lane (stream order) ends the round
    uint32_t total, lstop;
    {
      const uint64_t fm = __ballot(mine && lo.stopc != 0);
      if (fm) {
        const uint32_t fl = __builtin_ctzll(fm);
        lstop = rdlane(lo.stopc, fl);
        total = rdlane(off, fl) + rdlane(lo.bytes, fl);
        if (lane > fl) lo.nm = 0;  // later lanes are void
        nvalid = fl + 1;
      } else {
        lstop = 0;
        total = rdlane(off + mynb, nvalid - 1);
      }
    }
    copy_matches<C>(mrec, owner, sk, lane, q0, lo.nm, pf);
    sk.flush(total);
    pf.tick(P_ADLER);
    pf.count(C_LANES, nvalid);
    bp = rdlane(lo.endp, nvalid - 1);
    if (lstop == kStEob) break;
    if (lstop != 0 && lstop != kStTrunc) return (int)lstop;
  }
  *bp_io = bp;
  return MD_OK;
}

template <class C, bool PROF>
__global__ __launch_bounds__(kWave) void inflate_wave_kernel(
    int format, uint32_t n, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
    const uint64_t *__restrict__ in_len, uint8_t *out, const uint64_t *__restrict__ out_off,
    const uint64_t *__restrict__ out_cap, uint64_t *__restrict__ out_len,
    uint64_t *__restrict__ consumed, int32_t *__restrict__ status, uint32_t *__restrict__ checksum,
    uint64_t *__restrict__ dbg) {
  __shared__ Smem<C> smem;  // static: LDS addresses fold into the instructions' offset fields
  Smem<C> *smg = &smem;
  Prof<PROF> pf;
  pf.init();
  const uint32_t lane = threadIdx.x;
  const uint32_t sid = blockIdx.x;
  if (sid >= n) return;

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

  Sink5 sk;
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
  if (lane < 4) {  // the two self-looping STOP entries (+ padding)
    const uint32_t i = kStopEobI + (lane & 1);
    smg->lut[kStopEobI + lane] = mk_entry(0, 0, 0, 0, i);
  }

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
        uint32_t lroot = 0;
        if (rc == MD_OK) {
          // walk entries from the packed 16-bit ones (sub-table entries start behind the root table)
          lroot = uni(lit.root);
          const uint32_t droot = uni(dist.root);
          for (uint32_t i = lane; i < 852; i += kWave)
            smg->lut[i] = walk_lit(smg->u.sc.lit[i], i >= (1u << lroot), lroot, droot);
          for (uint32_t i = lane; i < 592; i += kWave)
            smg->lut[kDistB + i] = walk_dist(smg->u.sc.dist[i], i >= (1u << droot), lroot, droot);
        }
        pf.tick(P_HEADER);
        if (rc == MD_OK) rc = inflate_block<C>(smg, inp, sk, lroot, lane, total_bits, &bp, pf);
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

}  // namespace v5
}  // namespace md

//                              S   LMAX MMAX KMAX IN_BYTES PASSES STAGE
using WaveCfg = md::v4::Cfg<256, 40, 16, 64, 4096, 5, 6144>;

extern "C" int md_launch_inflate_wave(int format, uint32_t n, const uint8_t *in, const uint64_t *in_off,
                                      const uint64_t *in_len, uint8_t *out, const uint64_t *out_off,
                                      const uint64_t *out_cap, uint64_t *out_len, uint64_t *consumed,
                                      int32_t *status, uint32_t *checksum, uint64_t *dbg, hipStream_t stream) {
  if (n == 0) return 0;
  using namespace md::v5;
  dim3 grid(n), block(md::kWave);
  if (dbg)
    hipLaunchKernelGGL((inflate_wave_kernel<WaveCfg, true>), grid, block, 0, stream, format, n, in,
                       in_off, in_len, out, out_off, out_cap, out_len, consumed, status, checksum, dbg);
  else
    hipLaunchKernelGGL((inflate_wave_kernel<WaveCfg, false>), grid, block, 0, stream, format, n, in,
                       in_off, in_len, out, out_off, out_cap, out_len, consumed, status, checksum, dbg);
  return (int)hipGetLastError();
}
