"""A SECOND, independent restatement of the parse of De.Def.Ns.deflate, written from the formulas of
lib/de.ml:3704-3925 (block-split statistics, hc_matchfinder_longest_match / skip_positions, compress_greedy) in plain
Python - slow, for differential tests against oracle/de_def_ns.c only.  It produces what the parser DECIDES: the
blocks (start position of each) and their tokens, not the bits (code construction and block types are the oracle's
alone).  Two readings of the same OCaml agreeing is weaker than running the reference, which cannot be done here
(SURVEY.md 8(c)); it is what this image allows."""
import struct

WINDOW = 1 << 15              # lib/de.ml:3118
MIN_BLOCK = 10000             # lib/de.ml:3041
SOFT_MAX_BLOCK = 300000       # lib/de.ml:3054
CHECK = 512                   # num_observations_per_block_check, lib/de.ml:3704
LEVELS = {1: (2, 8), 2: (6, 10), 3: (12, 14), 4: (24, 24)}  # max_search_depth, nice_match_length, lib/de.ml:3931-3934


class Stats:                   # block_split_stats, lib/de.ml:3139-3144
    def __init__(self):
        self.reset()

    def reset(self):           # init_block_split_stats, lib/de.ml:3691-3695
        self.new = [0] * 10
        self.old = [0] * 10
        self.n_new = 0
        self.n_old = 0

    def observe(self, kind):
        self.new[kind] += 1
        self.n_new += 1

    def wants_to_end(self, begin, pos, end):   # should_end_block + do_end_block_check, lib/de.ml:3706-3745
        if self.n_new < CHECK or pos - begin < MIN_BLOCK or end - pos < MIN_BLOCK:
            return False
        if self.n_old > 0:
            delta = sum(abs(n * self.n_old - o * self.n_new) for n, o in zip(self.new, self.old))
            if delta + (pos - begin) // 4096 * self.n_old >= CHECK * 200 // 512 * self.n_old:
                return True
        for k in range(10):
            self.n_old += self.new[k]
            self.old[k] += self.new[k]
            self.new[k] = 0
        self.n_new = 0
        return False


def parse(data, level):
    """-> [(block_start, [tokens])], a token is a literal byte (int) or (length, offset)."""
    depth, nice0 = LEVELS[level]
    n = len(data)
    pad = data + bytes(8)

    def h(pos):                # lz_hash, lib/de.ml:3770-3772 (hash order 16)
        return ((struct.unpack_from("<I", pad, pos)[0] * 0x1E35A7BD) & 0xFFFFFFFF) >> 16

    head = [-WINDOW] * (1 << 16)          # hc_matchfinder_init, lib/de.ml:3120-3124
    chain = [0] * WINDOW
    next_hash = 0
    best_nice_max = [0, min(nice0, 258), 258]    # lens.best / nice / max: made once, max and nice only shrink
    stats = Stats()
    blocks = []
    pos = 0

    def slide():               # hc_matchfinder_slide_window, lib/de.ml:3761-3768
        for k in range(len(head)):
            head[k] -= WINDOW
        for k in range(WINDOW):
            chain[k] -= WINDOW

    while pos != n:
        begin = pos
        limit = pos + min(n - pos, SOFT_MAX_BLOCK)
        tokens = []
        stats.reset()
        while pos < limit and not stats.wants_to_end(begin, pos, n):
            if best_nice_max[2] > n - pos:
                best_nice_max[2] = n - pos
                best_nice_max[1] = min(best_nice_max[1], best_nice_max[2])
            best, nice, mx = 2, best_nice_max[1], best_nice_max[2]
            # hc_matchfinder_longest_match, lib/de.ml:3807-3828
            cur = pos & (WINDOW - 1)
            if cur == 0 and pos != 0:
                slide()
            cutoff = cur - WINDOW
            where = pos
            if mx >= 5:
                node = head[next_hash]
                head[next_hash] = cur
                chain[cur] = node
                next_hash = h(pos + 1)
                left = depth
                if node > cutoff and best < nice:
                    base = pos & ~(WINDOW - 1)
                    while True:        # _matchfinder_longest_rec, lib/de.ml:3774-3805
                        cand = base + node
                        if pad[cand + best] == pad[pos + best]:
                            ln = 0
                            while ln < mx and pad[cand + ln] == pad[pos + ln]:
                                ln += 1
                            if ln >= nice:
                                best, where = ln, cand
                                break
                            if ln > best:
                                best, where = ln, cand
                        node = chain[node & (WINDOW - 1)]
                        left -= 1
                        if node <= cutoff or left == 0:
                            break
            if best >= 3:
                tokens.append((best, pos - where))
                stats.observe(8 + (1 if best >= 9 else 0))      # observe_match, lib/de.ml:3859-3862
                pos += 1
                count = best - 1                                  # hc_matchfinder_skip_positions, lib/de.ml:3841-3843
                if count + 5 > n - pos:
                    pos += count
                else:
                    for _ in range(count):
                        cur = pos & (WINDOW - 1)
                        if cur == 0 and pos != 0:
                            slide()
                        chain[cur] = head[next_hash]
                        head[next_hash] = cur
                        pos += 1
                        next_hash = h(pos)
            else:
                tokens.append(data[pos])
                stats.observe(((pos << 5) & 6) | (pos & 1))       # observe_literal is fed the POSITION, lib/de.ml:3905
                pos += 1
        blocks.append((begin, tokens))
    return blocks
