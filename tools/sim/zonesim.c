// zonesim.c — offline model of the lane-parallel zone decode (design aid, not product code).
// Parses a zlib stream, then replays the speculative zone scheme for a given zone size S and
// reports wave-slot counts per pass, passes per round and lane efficiency.
//   gcc -O2 -o zonesim zonesim.c && ./zonesim file.z S [PASSES] [ROOT]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t *D; static size_t DN;
static inline uint32_t bitsat(uint64_t p, int n) {  // n <= 24
  uint64_t b = p >> 3; uint32_t v = 0;
  for (int k = 0; k < 5; k++) if (b + k < DN) v |= (uint64_t)D[b + k] << (8 * k) >> 0 ? 0 : 0;
  uint64_t w = 0; for (int k = 0; k < 8; k++) if (b + k < DN) w |= (uint64_t)D[b + k] << (8 * k);
  return (uint32_t)((w >> (p & 7)) & ((1u << n) - 1));
}
typedef struct { uint16_t sym[1 << 15]; uint8_t len[1 << 15]; int maxl; } Tab;
static void build(Tab *t, const uint8_t *lens, int n) {
  int cnt[16] = {0}, nxt[16]; for (int i = 0; i < n; i++) cnt[lens[i]]++; cnt[0] = 0;
  int code = 0; for (int b = 1; b < 16; b++) { code = (code + cnt[b - 1]) << 1; nxt[b] = code; }
  memset(t->len, 0, sizeof t->len); t->maxl = 0;
  for (int s = 0; s < n; s++) { int l = lens[s]; if (!l) continue; if (l > t->maxl) t->maxl = l;
    int c = nxt[l]++; int r = 0; for (int k = 0; k < l; k++) r |= ((c >> k) & 1) << (l - 1 - k);
    for (int x = r; x < (1 << 15); x += 1 << l) { t->sym[x] = s; t->len[x] = l; } }
}
static const int LB[29]={3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const int LX[29]={0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const int DB[30]={1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const int DX[30]={0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static Tab LT, DT; static int LROOT = 9, DROOT = 6;
// one token at p: returns bits consumed (0 = EOB/invalid stop), *slots, *nb, *kind(0 lit,1 match,2 eob,3 bad)
static int token(uint64_t p, uint64_t total, int *slots, int *nb, int *kind, int *mlen, int *mdist) {
  uint32_t w = bitsat(p, 15); int l = LT.len[w]; if (!l) { *kind = 3; *slots = 1; *nb = 0; return 0; }
  int s = LT.sym[w]; int sl = 1 + (l > LROOT); int used = l;
  if (s < 256) { *kind = 0; *slots = sl; *nb = 1; return used; }
  if (s == 256) { *kind = 2; *slots = sl; *nb = 0; return used; }
  if (s > 285) { *kind = 3; *slots = sl; *nb = 0; return 0; }
  int len = LB[s - 257] + bitsat(p + used, LX[s - 257]); used += LX[s - 257];
  w = bitsat(p + used, 15); int dl = DT.len[w]; if (!dl) { *kind = 3; *slots = sl + 1; *nb = 0; return 0; }
  int ds = DT.sym[w]; if (ds > 29) { *kind = 3; *slots = sl + 1; *nb = 0; return 0; }
  used += dl; int dist = DB[ds] + bitsat(p + used, DX[ds]); used += DX[ds];
  *kind = 1; *slots = sl + 1 + (dl > DROOT); *nb = len; *mlen = len; *mdist = dist; return used;
}
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb"); static uint8_t buf[1 << 22]; DN = fread(buf, 1, sizeof buf, f); D = buf + 2; DN -= 6;
  int S = atoi(argv[2]); int PASSES = argc > 3 ? atoi(argv[3]) : 5; if (argc > 4) LROOT = atoi(argv[4]);
  uint64_t total = DN * 8, p = 0; int last = 0;
  // stats
  double nlit = 0, nmat = 0, litbits = 0, matbits = 0, matbytes = 0, nblocks = 0, far = 0, longcode = 0;
  double rounds = 0, passes = 0, wslots = 0, lane_slots_useful = 0, lanes_acc = 0, wslots_p[16] = {0}, act_p[16] = {0};
  double sync_hist[64] = {0}; double ml_gt16=0, ml_gt32=0, ml_gt8=0, straddle=0, nearc=0, farc=0, d_lt8=0; double outbytes = 0; double a1slots=0, redo2 = 0;
  while (!last) {
    last = bitsat(p, 1); int type = bitsat(p + 1, 2); p += 3; nblocks++;
    if (type == 0) { p = (p + 7) & ~7ull; int n = bitsat(p, 16); p += 32 + 8ull * n; outbytes += n; continue; }
    uint8_t lens[320] = {0};
    if (type == 1) { for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8; build(&LT, lens, 288);
      uint8_t dl[30]; memset(dl, 5, 30); build(&DT, dl, 30); }
    else { int hl = bitsat(p, 5) + 257, hd = bitsat(p + 5, 5) + 1, hc = bitsat(p + 10, 4) + 4; p += 14;
      static const int ord[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15}; uint8_t cl[19] = {0};
      for (int i = 0; i < hc; i++) { cl[ord[i]] = bitsat(p, 3); p += 3; }
      static Tab CT; build(&CT, cl, 19); int i = 0;
      while (i < hl + hd) { uint32_t w = bitsat(p, 7); int s = CT.sym[w]; p += CT.len[w];
        if (s < 16) lens[i++] = s; else if (s == 16) { int r = 3 + bitsat(p, 2); p += 2; while (r--) { lens[i] = lens[i-1]; i++; } }
        else if (s == 17) { int r = 3 + bitsat(p, 3); p += 3; while (r--) lens[i++] = 0; }
        else { int r = 11 + bitsat(p, 7); p += 7; while (r--) lens[i++] = 0; } }
      build(&LT, lens, hl); build(&DT, lens + hl, hd); }
    // rounds of this block
    uint64_t bp = p; int eob = 0;
    while (!eob) {
      rounds++;
      uint64_t start[64], end[64]; int stop[64], sl[64], ntok[64], nb[64];
      int active[64]; for (int i = 0; i < 64; i++) { start[i] = bp + (uint64_t)i * S; active[i] = 1; stop[i] = 0; }
      int pass = 0;
      for (;;) {
        int maxs = 0, nact = 0;
        for (int i = 0; i < 64; i++) if (active[i]) {
          nact++; uint64_t q = start[i], lim = bp + (uint64_t)(i + 1) * S; int s = 0, nt = 0, b = 0; stop[i] = 0;
          while (q < lim) { int ts, tb, k, ml, md; int u = token(q, total, &ts, &tb, &k, &ml, &md); s += ts;
            if (k >= 2 || q + u > total) { stop[i] = k == 2 ? 2 : 3; if (k == 2) q += u; break; } q += u; nt++; b += tb; }
          end[i] = q; sl[i] = s; ntok[i] = nt; nb[i] = b; if (s > maxs) maxs = s;
        }
        passes++; wslots += maxs; if (pass < 16) { wslots_p[pass] += maxs; act_p[pass] += nact; }
        if (pass == 0) a1slots += maxs;
        pass++;
        int any = 0; for (int i = 63; i >= 1; i--) { active[i] = (stop[i - 1] == 0 && end[i - 1] != start[i]); if (active[i]) { start[i] = end[i - 1]; any = 1; } }
        active[0] = 0;
        if (!any || pass > PASSES) break;
      }
      int nvalid = 64; for (int i = 1; i < 64; i++) if (stop[i - 1] != 0 || end[i - 1] != start[i] ) { nvalid = i; break; }
      // note: if passes exhausted, start[i] was overwritten without re-decode; treat as invalid
      lanes_acc += nvalid;
      for (int i = 0; i < nvalid; i++) { lane_slots_useful += sl[i]; }
      // token stats over accepted lanes (recount from truth)
      { uint64_t q = bp; uint64_t qe = end[nvalid - 1]; double R0 = outbytes;
        while (q < qe) { int ts, tb, k, ml = 0, md = 0; int u = token(q, total, &ts, &tb, &k, &ml, &md);
          if (k == 0) { nlit++; litbits += u; } else if (k == 1) { nmat++; matbits += u; matbytes += ml; if (md > 6144) far++;
            if (ml > 8) ml_gt8++; if (ml > 16) ml_gt16++; if (ml > 32) ml_gt32++; if (md < 8) d_lt8++;
            double src = outbytes - md; if (src + ml <= R0) farc++; else { nearc++; if (src < R0) straddle++; } } else if (k==2) {break;}
          if (ts > (k == 1 ? 2 : 1)) longcode++; outbytes += tb; q += u; } }
      if (stop[nvalid - 1] == 2) eob = 1; else if (stop[nvalid - 1]) { fprintf(stderr, "bad stream\n"); return 1; }
      bp = end[nvalid - 1];
    }
    p = bp;
  }
  printf("S=%d PASSES=%d ROOT=%d: blocks %.0f out %.0f lit %.0f (%.2f bits) mat %.0f (%.2f bits, %.2f bytes, far %.3f) longcode/token %.4f\n", S, PASSES, LROOT, nblocks, outbytes,
         nlit, litbits / nlit, nmat, matbits / nmat, matbytes / nmat, far / nmat, longcode / (nlit + nmat));
  printf("  rounds %.0f passes/round %.2f wave-slots/round %.1f (A1 %.1f) lanes accepted %.1f useful lane-slots/round %.1f eff(1 pass) %.3f\n",
         rounds, passes / rounds, wslots / rounds, a1slots / rounds, lanes_acc / rounds, lane_slots_useful / rounds,
         lane_slots_useful / 64.0 / (a1slots));
  printf("  total wave-slots %.0f ; ideal (1 pass perfect) %.0f\n", wslots, lane_slots_useful / 64);
  printf("  matches/round %.1f: far %.1f near %.1f straddle %.2f ; ml>8 %.3f ml>16 %.3f ml>32 %.4f d<8 %.4f\n", nmat/rounds, farc/rounds, nearc/rounds, straddle/rounds, ml_gt8/nmat, ml_gt16/nmat, ml_gt32/nmat, d_lt8/nmat);
  for (int i = 0; i < 8; i++) printf("   pass %d: wave-slots %.1f active lanes %.1f\n", i, wslots_p[i] / rounds, act_p[i] / rounds);
  return 0;
}
