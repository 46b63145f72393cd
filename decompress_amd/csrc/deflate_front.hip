// deflate_front.hip — the decision-independent half of De.Lz77 (lib/de.ml:4013-4515) for gfx950 (MI355X).
//
// deflate_slow inserts every position p < p_end into the hash chains exactly once and in order, whatever the lazy
// matcher decides (lib/de.ml:4351-4410: insert at strstart, then every position a match covers), so
//   hash_head(p)  = the latest earlier position with the hash of p           (insert_string, lib/de.ml:4220-4226)
//   prev[c]       = hash_head(c) for every candidate c still in reach        (the chain links)
// are pure functions of the input.  Two kernels compute them for the whole batch before the sequential kernel runs:
//
//   deflate_link_kernel   one workgroup (4 wavefronts) per stream, the 32 K-entry head table in LDS (128 KiB): 8 x 64
//                         positions per group, one LDS atomic-max per position returning the old head (the groups
//                         take their turn in stream order), equal hashes inside a step matched by ballots.  Output link[p] = p - hash_head(p) as 16 bits (0 = none in reach):
//                         the only HBM traffic is the input read once and 2 bytes per position written once.
//   deflate_match_kernel  longest_match (lib/de.ml:4110-4174) run for every position as if reached with
//                         prev_length = 2 (full and quartered chain), positions independent of each other: the grid
//                         is all 256-position chunks of the batch, XCD-contiguous so that the 32 KiB window of a
//                         stream region stays in one L2.  Output flg[p] (FL_*) and m[p] / mq[p].
//
// Neither kernel takes a decision: the lazy evaluation, the queue, the trees and the bit encoder read these
// arrays in deflate_kernel.hip.  Before this split the same work sat inside the sequential kernel with the head
// table in HBM (one atomic and ~3 dependent HBM round trips per position): 66x the algorithmic traffic.
#include "deflate_common.hpp"

namespace md {
namespace defl {

constexpr int PGL = 8;  // steps of 64 positions whose LDS atomics are in flight together (link kernel)

// ---- per-stream slots ---------------------------------------------------------------------------
constexpr uint32_t kPlanThreads = 1024;
__device__ __forceinline__ uint64_t slot_positions(uint32_t p_end, uint32_t len) {
  return p_end ? (((uint64_t)len + 64 + (kChunk - 1)) / kChunk) * kChunk : 0;  // reads a little past p_end stay inside
}
__global__ __launch_bounds__(kPlanThreads) void deflate_plan_kernel(uint32_t n, const uint64_t *__restrict__ in_len, int driver,
                                                                    int matcher, int level, uint64_t cap_positions,
                                                                    uint32_t cap_chunks, uint32_t *__restrict__ p_end,
                                                                    uint64_t *__restrict__ slot, uint32_t *__restrict__ chunk0,
                                                                    uint32_t *__restrict__ flags) {
  __shared__ uint64_t spos[kPlanThreads];
  __shared__ uint32_t schk[kPlanThreads];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (n + kPlanThreads - 1) / kPlanThreads;
  const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
  const uint32_t eff = effective_level(driver, matcher, level);
  const bool no_text = driver == 4 || driver == 5;  // DRV_ENCODE / DRV_SCRIPT: the input is a list of commands / operations
  uint64_t pos = 0;
  uint32_t chk = 0;
  for (uint32_t i = lo; i < hi; i++) {
    const uint64_t l64 = in_len[i];
    const uint32_t len = l64 > MD_MAX_STREAM ? 0u : (uint32_t)l64;
    // De.Def.Ns (driver 6): every position with 5 bytes left is inserted (lib/de.ml:3829-3833, :3853-3856)
    const uint32_t pe = driver == 6 ? ((level >= 1 && level <= 4 && len >= 5) ? len - 4 : 0u) : no_text ? 0u : stream_p_end(matcher, eff, len);
    p_end[i] = pe;
    pos += slot_positions(pe, len);
    chk += (pe + kChunk - 1) / kChunk;
  }
  spos[t] = pos;
  schk[t] = chk;
  __syncthreads();
  if (t == 0) {
    uint64_t a = 0;
    uint32_t c = 0;
    for (uint32_t k = 0; k < kPlanThreads; k++) {
      const uint64_t v = spos[k];
      const uint32_t w = schk[k];
      spos[k] = a;
      schk[k] = c;
      a += v;
      c += w;
    }
    slot[n] = a;
    chunk0[n] = c;
    // the caller sized the workspace from md_deflate_params.total_in_bytes: a batch that needs more is refused
    flags[0] = (cap_positions != 0 && (a > cap_positions || c > cap_chunks)) ? 1u : 0u;
  }
  __syncthreads();
  pos = spos[t];
  chk = schk[t];
  for (uint32_t i = lo; i < hi; i++) {
    const uint64_t l64 = in_len[i];
    const uint32_t len = l64 > MD_MAX_STREAM ? 0u : (uint32_t)l64;
    const uint32_t pe = p_end[i];
    slot[i] = pos;
    chunk0[i] = chk;
    pos += slot_positions(pe, len);
    chk += (pe + kChunk - 1) / kChunk;
  }
}

// the 4 bytes at pos (little-endian) for pos < p_end (something for the others), without a branch: Lz's last string (pos = len - 3) has
// only 3 bytes — it is taken from the word one byte earlier; len >= 4 whenever p_end > 1
__device__ __forceinline__ uint32_t load_w4(const uint8_t *__restrict__ src, uint32_t slen, uint32_t p_end, uint32_t pos) {
  uint32_t a = pos + 4 <= slen ? pos : slen - 4;
  a = pos < p_end ? a : 0u;
  uint32_t v;
  __builtin_memcpy(&v, src + a, 4);
  return v >> ((8 * (pos - a)) & 31);  // (whatever for pos >= p_end: nobody looks at it)
}

// ---- hash chains ----------------------------------------------------------------------------------
// head[h] <- max(pos), one LDS atomic per position.  The head table of a stream is 128 KiB of LDS, so a CU holds one
// stream — and one wavefront alone runs at the latency of its own instruction stream (9 cycles per instruction
// measured).  LW wavefronts therefore share the stream: wavefront w takes the groups k = w, w + LW, ... of 8 x 64
// positions, loads and hashes them ahead, and only the atomics themselves are taken in stream order — a turn counter
// in LDS lets group k issue its eight atomics once group k - 1 has got its results back (a wavefront's own LDS
// operations execute in order).  Everything after the atomics (sorting out equal hashes, the stores) overlaps with
// the other wavefronts' groups.  Measured per GiB of input: 4 wavefronts 4.9 ms (word text 7.0), 8: 3.1 (4.5), 16:
// 3.2 (3.8) — from 8 on the chain of turns is what is left.
// The values a set of equal hashes gets back are >= the head before the set and one of them is exactly that value: a
// lane that shares its hash with another lane of its step is recognised by a returned position inside the step, and
// such steps sort themselves out by ballots (the predecessor of a lane is the nearest lower lane with its hash, else
// the smallest value the set got back).
// NS = De.Def.Ns's hc_matchfinder (lib/de.ml:3765-3856): the hash is 16 bits of 4 bytes times 0x1E35A7BD — twice the
// table LDS has room for, so the stream is gone through twice, once per half of the hash range (a chain never leaves
// its half) —, position 0 goes into bucket 0 whatever its bytes (next_hash4 starts at 0), and there is no tail.
constexpr int LW = 16;  // wavefronts per stream
template <bool NS>
__global__ __launch_bounds__(LW *kWave) void deflate_link_kernel(uint32_t n, const uint8_t *__restrict__ in,
                                                                 const uint64_t *__restrict__ in_off,
                                                                 const uint64_t *__restrict__ in_len,
                                                                 const uint32_t *__restrict__ p_end_a,
                                                                 const uint64_t *__restrict__ slot, uint32_t *__restrict__ link,
                                                                 uint32_t *__restrict__ tail, const uint32_t *__restrict__ flags,
                                                                 int matcher, const uint32_t *__restrict__ order) {
  __shared__ uint32_t head[HASH_SIZE];  // absolute position, 0 = NIL (position 0 can never match, like the reference)
  __shared__ uint32_t gmin_all[LW][kWave];
  __shared__ uint32_t turn;  // the group whose atomics may be issued
  // De.Lz77: position 0 is NIL (it can never be a match source there); Def.Ns: position 0 is an ordinary candidate, so
  // the table holds position + 1
  constexpr uint32_t bias = NS ? 1u : 0u;
  const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (blockIdx.x >= n || flags[0]) return;
  const uint32_t sid = order ? order[blockIdx.x] : blockIdx.x;  // a CU holds one stream: the longest go first
  const uint32_t p_end = p_end_a[sid];
  const uint64_t l64 = in_len[sid];
  const uint32_t slen = l64 > MD_MAX_STREAM ? 0u : (uint32_t)l64;
  const uint8_t *src = in + in_off[sid];
  uint32_t *lk = link + slot[sid];
  uint32_t *gmin = gmin_all[wv];
  if (slen < 4) {  // at most one string (Lz, 3 bytes): nothing before it
    if (threadIdx.x < p_end) lk[threadIdx.x] = 0;  // (nobody compares its fingerprint)
    if (threadIdx.x < 2 && !NS) tail[2 * sid + threadIdx.x] = 0;
    return;
  }
  for (uint32_t pass = 0; pass < (NS ? 2u : 1u); pass++) {
  __syncthreads();
  {
    uint4 *h4 = reinterpret_cast<uint4 *>(head);
    for (uint32_t i = threadIdx.x; i < (uint32_t)HASH_SIZE / 4; i += LW * kWave) h4[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) turn = 0;
  }
  __syncthreads();
  // the input of a wavefront's next two groups is requested before the current one is worked on
  // (branch-free: a branch around a load makes the compiler wait for every load in flight)
  auto load_group = [&](uint32_t pe, uint32_t (&w)[PGL]) {
#pragma unroll
    for (int g = 0; g < PGL; g++) w[g] = load_w4(src, slen, p_end, pe + g * kWave + lane);
  };
  constexpr uint32_t kGroup = PGL * kWave;
  const uint32_t ngroups = (p_end + kGroup - 1) / kGroup;
  uint32_t wa[PGL], wb[PGL];
  load_group(wv * kGroup, wa);
  load_group((wv + LW) * kGroup, wb);
  for (uint32_t k = wv; k < ngroups; k += LW) {
    const uint32_t pe = k * kGroup;
    uint32_t w4[PGL], hv[PGL], ret[PGL];
    bool mine[PGL];  // the position's hash is in this pass's half of the range
#pragma unroll
    for (int g = 0; g < PGL; g++) {
      w4[g] = wa[g];
      wa[g] = wb[g];
      if (NS) {
        const uint32_t h = (pe + g * kWave + lane) == 0 ? 0u : (uint32_t)(w4[g] * 0x1E35A7BDu) >> 16;
        hv[g] = h & (HASH_SIZE - 1);
        mine[g] = (h >> HASH_BITS) == pass;
      } else {
        hv[g] = hash_of(matcher, w4[g]);
        mine[g] = true;
      }
    }
    load_group(pe + 2 * LW * kGroup, wb);
    while (__hip_atomic_load(&turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != k) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int g = 0; g < PGL; g++) {
      const uint32_t pos = pe + g * kWave + lane;
      ret[g] = (pos < p_end && mine[g]) ? atomicMax(&head[hv[g]], pos + bias) : 0xffffffffu;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the group's atomics have been performed: the next group may go
    if (lane == 0) __hip_atomic_store(&turn, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int g = 0; g < PGL; g++) {
      const uint32_t s0 = pe + g * kWave;
      if (s0 >= p_end) break;  // uniform
      const uint32_t pos = s0 + lane;
      const bool valid = pos < p_end && mine[g];
      uint32_t c1;
      if (__ballot(valid && ret[g] >= s0 + bias) == 0) {
        c1 = valid ? ret[g] : 0;  // no two lanes share a hash: every returned value is the head before the step
      } else {
        const uint32_t h = hv[g];
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < HASH_BITS; bit++) {
          const bool mine = (h >> bit) & 1;
          const uint64_t bal = __ballot(mine);
          same &= mine ? bal : ~bal;
        }
        const uint64_t below = same & lanes_below(lane);
        const uint32_t first = valid ? (uint32_t)__builtin_ctzll(same) : lane;
        gmin[lane] = 0xffffffffu;
        __builtin_amdgcn_wave_barrier();
        if (valid) atomicMin(&gmin[first], ret[g]);
        __builtin_amdgcn_wave_barrier();
        c1 = !valid ? 0u : below ? s0 + bias + 63u - (uint32_t)__builtin_clzll(below) : __hip_atomic_load(&gmin[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
      }
      if (valid) {
        const uint32_t d = pos + bias - c1;
        lk[pos] = ((c1 != 0 && d <= 32767u) ? d : 0u) | (fp16(w4[g]) << 16);
      }
    }
  }
  }  // pass
  __syncthreads();
  if (NS) return;
  // De's position len - 3 (lookahead 3): hash4 reads a 4th byte beyond the data (H7) — zero before the first slide of
  // the reference's 64 KiB buffer, the byte 32 KiB earlier after it.  Which one depends on the matcher's trajectory:
  // both heads are handed over.
  if (threadIdx.x < 2) {
    uint32_t res = 0;
    if (matcher == MD_MATCHER_DE && slen >= 3) {
      const uint32_t p = slen - 3;
      const uint32_t b3 = (threadIdx.x == 1 && slen >= (uint32_t)WSIZE) ? src[slen - WSIZE] : 0u;
      const uint32_t w = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16) | (b3 << 24);
      res = head[hash_of(matcher, w)];
    }
    tail[2 * sid + threadIdx.x] = res;
  }
}

// ---- longest_match ahead --------------------------------------------------------------------------
// One chain per lane, PGM steps of 64 positions per wavefront in flight together.  The bar is prev_length = 2 and the
// chain is the full one; the best candidate after max_chain >> 2 links is kept as well (the matcher walks the
// quartered chain when prev_length >= good_length, lib/de.ml:4117-4119).  A candidate counts when it shares the first
// 3 bytes (the 16-bit pre-filters of the reference can only skip candidates that would not win); the walk ends at the
// first candidate of nice_length, after max_chain links, or where the chain leaves the reach of the position.  The
// window limit is pos - MAX_DIST whatever the window base: before the first slide the base is 0, after a slide
// strstart - base >= MAX_DIST.  The head candidate is admitted at distance == MAX_DIST, links only below it
// (lib/de.ml:4367-4369 vs :4165).
__global__ __launch_bounds__(kWave, MD_MATCH_WAVES) void deflate_match_kernel(uint32_t n, uint32_t nchunks_max, const uint8_t *__restrict__ in,
                                                              const uint64_t *__restrict__ in_off,
                                                              const uint64_t *__restrict__ in_len,
                                                              const uint32_t *__restrict__ p_end_a,
                                                              const uint64_t *__restrict__ slot,
                                                              const uint32_t *__restrict__ chunk0,
                                                              const uint32_t *__restrict__ link, uint8_t *__restrict__ flg,
                                                              uint32_t *__restrict__ m, uint32_t *__restrict__ mq,
                                                              const uint32_t *__restrict__ flags, uint32_t max_chain,
                                                              uint32_t nice, uint32_t skip) {
  const uint32_t lane = threadIdx.x;
  if (flags[0]) return;
  // workgroup b runs on XCD b % 8: give every XCD one contiguous eighth of the chunks, so that the chunks that share
  // a 32 KiB window (and its 64 KiB of links) meet in the same L2
  const uint32_t per = (nchunks_max + 7) / 8;
  const uint32_t c = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per || c >= chunk0[n]) return;
  // the stream of chunk c: the last i with chunk0[i] <= c.  Streams of a batch are usually about the same size:
  // the proportional guess is right or off by one, and the bisection (a dozen dependent loads, as long as the rest
  // of a short chain walk) only has the remaining interval to go through
  uint32_t lo = 0, hi = n;
  {
    const uint32_t total = chunk0[n];
    uint32_t gs = (uint32_t)(((uint64_t)c * n) / total);  // c < total
    gs = gs < n ? gs : n - 1;
    const uint32_t a0 = chunk0[gs], a1 = chunk0[gs + 1];
    if (a0 <= c && c < a1) lo = gs, hi = gs + 1;
    else if (c < a0) hi = gs;
    else lo = gs + 1;
  }
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (chunk0[mid] <= c) lo = mid;
    else hi = mid;
  }
  const uint32_t sid = lo;
  const uint32_t p_end = p_end_a[sid];
  const uint32_t pe = (c - chunk0[sid]) * kChunk;
  if (pe >= p_end) return;
  // a stream in pieces: the first `skip` positions of the text are the window the piece brings along - the matcher is
  // past them, they only have to be in the chains
  if (pe + kChunk <= skip) return;
  const uint32_t slen = (uint32_t)in_len[sid];
  const uint8_t *src = in + in_off[sid];
  const uint64_t so = slot[sid];
  const uint32_t *lk = link + so;
  const uint32_t qlimit = max_chain >> 2, kspec = max_chain < (uint32_t)KSPEC ? max_chain : (uint32_t)KSPEC;

  // the chunk's own text, 16 bytes beyond its last position, in LDS: a comparison starts with the 16 bytes at the
  // position against the 16 bytes at the candidate (held in registers they cost a sixth of the wavefronts per CU)
  __shared__ alignas(16) uint8_t own_text[kChunk + 16];
  if (lane <= kChunk / 16) {
    const uint32_t a = pe + lane * 16;
    v4u v = {0u, 0u, 0u, 0u};
    if (a + 16 <= slen) __builtin_memcpy(&v, src + a, 16);
    else
      for (uint32_t k = 0; a + k < slen; k++) v[k >> 2] |= (uint32_t)src[a + k] << (8 * (k & 3));
    *reinterpret_cast<v4u *>(own_text + lane * 16) = v;
  }
  // per position: its 4 bytes, the candidate, fc = verdict (FL_*, 0 = walking) | links walked << 4, and the best match so
  // far as it will be stored (length << 16 | distance): of the full chain, and of its first quarter
  uint32_t w4[PGM], cw[PGM], fc[PGM], mv[PGM], mqv[PGM];
#pragma unroll
  for (int g = 0; g < PGM; g++) {
    const uint32_t pos = pe + g * kWave + lane;
    w4[g] = slen >= 4 ? load_w4(src, slen, p_end, pos) : 0u;
    cw[g] = 0;
    if (pos < p_end) {
      if (slen < 4) w4[g] = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
      const uint32_t l = lk[pos] & 0xffffu;
      cw[g] = l ? pos - l : 0u;
    }
    fc[g] = 0;
    mv[g] = (uint32_t)(MIN_MATCH - 1) << 16;
    mqv[g] = (uint32_t)(MIN_MATCH - 1) << 16;
  }
  // One link of every chain per iteration.  The usual candidate fails the 3-byte test: that path is straight-line
  // (loads from a harmless address for lanes that have nothing to do, selects instead of branches — a branch per
  // lane group costs more scalar instructions than the work it skips); only a candidate that passes goes through
  // the comparison code.
  for (uint32_t lv = 0; lv < kspec; lv++) {
    bool act[PGM];
    bool any = false;
    // the candidate is in reach: the head at distance <= MAX_DIST, a link below it (pos - MAX_DIST < candidate; a
    // position that close to the start reaches everything before it) - one comparison, no branches
    const uint32_t lim = (uint32_t)MAX_DIST + (lv == 0 ? 1u : 0u);
#pragma unroll
    for (int g = 0; g < PGM; g++) {
      const uint32_t pos = pe + g * kWave + lane;
      act[g] = (pos < p_end) & ((fc[g] & 15u) == 0) & (cw[g] != 0) & (pos - cw[g] < lim);
      any = any | act[g];
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0) break;
    uint32_t rec[PGM], nx[PGM];
#pragma unroll
    for (int g = 0; g < PGM; g++) {
      const uint32_t a = act[g] ? cw[g] : 0u;  // candidate < pos <= len - 3; position 0 for idle lanes
      rec[g] = lk[a];
      const uint32_t l = rec[g] & 0xffffu;
      nx[g] = l ? a - l : 0u;
    }
#pragma unroll
    for (int g = 0; g < PGM; g++) {
      const bool hit = act[g] & ((rec[g] >> 16) == fp16(w4[g]));  // the candidate's 3 bytes may be ours: look at them
      if (__builtin_amdgcn_ballot_w64(hit)) {
        const uint32_t pos = pe + g * kWave + lane;
        if (hit && pos + MIN_LOOKAHEAD > slen) {  // too close to the end to compare ahead: the matcher's job, if the 3 bytes are there
          uint32_t v;
          __builtin_memcpy(&v, src + cw[g], 4);
          if (((v ^ w4[g]) & 0xffffffu) == 0) fc[g] |= 8u;
        } else if (hit) {
          // one 16-byte load of the candidate settles the 3-byte test and every match shorter than 16
          v4u cand;
          __builtin_memcpy(&cand, src + cw[g], 16);
          v4u own;
          __builtin_memcpy(&own, own_text + g * kWave + lane, 16);
          uint32_t len = prefix16(own, cand);
          if (len == 16) {
            // scan_end pre-filter (lib/de.ml:4133-4134): a candidate that differs at the end of the best match so
            // far cannot be longer
            bool look = true;
            const uint32_t bl = mv[g] >> 16;
            if (bl >= 16u) {
              uint16_t x, y;
              __builtin_memcpy(&x, src + pos + bl - 1, 2);
              __builtin_memcpy(&y, src + cw[g] + bl - 1, 2);
              look = x == y;
            }
            len = look ? lcp258_from16(src + pos, src + cw[g]) : 0u;
          }
          if (len > (mv[g] >> 16)) {  // (len < 3: the fingerprint lied; best >= 2)
            mv[g] = (len << 16) | (pos - cw[g]);
            if (len >= nice) fc[g] |= (uint32_t)FL_MATCH;
          }
        }
      }
      fc[g] += act[g] ? 16u : 0u;
      const bool atq = act[g] && (fc[g] >> 4) == qlimit;
      mqv[g] = atq ? mv[g] : mqv[g];
      cw[g] = act[g] ? nx[g] : cw[g];
    }
  }
#pragma unroll
  for (int g = 0; g < PGM; g++) {
    const uint32_t pos = pe + g * kWave + lane;
    if (pos < p_end) {
      const uint32_t lower = pos > (uint32_t)MAX_DIST ? pos - MAX_DIST : 0;
      uint32_t f = fc[g] & 15u;
      const uint32_t cnt = fc[g] >> 4;
      if (f == 0) {
        // chain exhausted (or never entered), or max_chain links walked: the verdict is final
        const bool entered = cnt != 0;
        const bool more = entered ? cw[g] > lower : (cw[g] != 0 && pos - cw[g] <= (uint32_t)MAX_DIST);
        if (!more || cnt >= max_chain) f = (mv[g] >> 16) >= (uint32_t)MIN_MATCH ? FL_MATCH : FL_ENDED;
      }
      if (f == FL_MATCH) {
        m[so + pos] = mv[g];
        mq[so + pos] = cnt < qlimit ? mv[g] : mqv[g];
      }
      flg[so + pos] = (uint8_t)(f == FL_MATCH || f == FL_ENDED ? f : 0);
    }
  }
}

}  // namespace defl
}  // namespace md

// The front workspace: a small part sized by the number of streams (slots, chunk starts, p_end, tails, flags) and a
// large one sized by the slot positions of the batch (link, flg, m, mq).
static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
extern "C" size_t md_front_small_bytes(uint32_t n) {
  const size_t k = n;
  return up256((k + 1) * 8) + up256((k + 1) * 4) + up256(k * 4) + up256(k * 8) + 256;
}
// the kernels' constants the host glue sizes its buffers with (capi.cpp has no copy of them)
extern "C" uint32_t md_front_chunk() { return md::defl::kChunk; }              // positions per wavefront of the match kernel
extern "C" uint32_t md_piece_state_bytes() { return md::defl::kPieceState; }   // bytes of a stream's state slot (struct Piece)
extern "C" size_t md_front_big_bytes(uint64_t positions) {
  const size_t np = (size_t)positions;
  return up256(np * 4) + up256(np) + 2 * up256(np * 4);
}
static inline uint8_t *bump(uint8_t *&p, size_t bytes) {
  uint8_t *r = p;
  p += up256(bytes);
  return r;
}
extern "C" void md_front_carve(void *small_ws, void *big_ws, uint32_t n, uint64_t positions, md::defl::Front *f) {
  uint8_t *p = (uint8_t *)small_ws;
  const size_t np = (size_t)positions, k = n;
  f->slot = (const uint64_t *)bump(p, (k + 1) * 8);
  f->chunk0 = (const uint32_t *)bump(p, (k + 1) * 4);
  f->p_end = (const uint32_t *)bump(p, k * 4);
  f->tail = (const uint32_t *)bump(p, k * 8);
  f->flags = (const uint32_t *)bump(p, 4);
  p = (uint8_t *)big_ws;
  f->link = (uint32_t *)bump(p, np * 4);
  f->flg = (uint8_t *)bump(p, np);
  f->m = (uint32_t *)bump(p, np * 4);
  f->mq = (uint32_t *)bump(p, np * 4);
}

// slot[] / chunk0[] / p_end[] of the batch (slot[n], chunk0[n] = totals); cap_* = what the big part was sized for
// (0 = it will be sized from the totals afterwards)
extern "C" int md_launch_deflate_plan(uint32_t n, const uint64_t *in_len, int driver, int matcher, int level,
                                      uint64_t cap_positions, uint32_t cap_chunks, const md::defl::Front *f, hipStream_t stream) {
  hipLaunchKernelGGL(md::defl::deflate_plan_kernel, dim3(1), dim3(md::defl::kPlanThreads), 0, stream, n, in_len, driver, matcher,
                     level, cap_positions, cap_chunks, (uint32_t *)f->p_end, (uint64_t *)f->slot, (uint32_t *)f->chunk0,
                     (uint32_t *)f->flags);
  return (int)hipGetLastError();
}
// link[] / tail[], then flg[] / m[] / mq[]: nchunks_max >= chunk0[n]
extern "C" int md_launch_deflate_front(uint32_t n, uint32_t nchunks_max, const uint8_t *in, const uint64_t *in_off,
                                       const uint64_t *in_len, int matcher, uint32_t max_chain, uint32_t nice,
                                       const md::defl::Front *f, const uint32_t *order, uint32_t match_skip, hipStream_t stream) {
  using namespace md::defl;
  if (n == 0 || nchunks_max == 0) return 0;
  hipLaunchKernelGGL(deflate_link_kernel<false>, dim3(n), dim3(LW * kWave), 0, stream, n, in, in_off, in_len, f->p_end, f->slot,
                     f->link, (uint32_t *)f->tail, f->flags, matcher, order);
  const uint32_t per = (nchunks_max + 7) / 8;
  hipLaunchKernelGGL(deflate_match_kernel, dim3(per * 8), dim3(kWave), 0, stream, n, nchunks_max, in, in_off, in_len, f->p_end,
                     f->slot, f->chunk0, f->link, f->flg, f->m, f->mq, f->flags, max_chain, nice, match_skip);
  return (int)hipGetLastError();
}

// the chains of De.Def.Ns's hc_matchfinder (deflate_ns.hip)
extern "C" int md_launch_link_ns(uint32_t n, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len, const md::defl::Front *f,
                                 hipStream_t stream) {
  const uint32_t *order = nullptr;
  using namespace md::defl;
  hipLaunchKernelGGL(deflate_link_kernel<true>, dim3(n), dim3(LW * kWave), 0, stream, n, in, in_off, in_len, f->p_end, f->slot,
                     f->link, (uint32_t *)f->tail, f->flags, 0, order);
  return (int)hipGetLastError();
}
