#!/usr/bin/env python3
"""Token statistics of the benchmark corpus and a model of the apply kernel's resolution loop (CPU only, a few
entries): tokens per entry, literal / match mix, and -- per round of 16 KiB of output, as the apply kernel cuts
them -- the DEPENDENCY DEPTH of the matches: a match whose source lies in bytes that another match of the round
writes can only be copied after that one.  The loop of otz_resolve_list runs one iteration per level, so the
depth is its iteration count.  Also modelled: "forwarding" (a waiting match whose source lies
wholly inside one unresolved match's destination takes that match's source instead).

usage: python3 tools/lz_depth.py [entries]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otezip_amd import corpus

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193,
         12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class Bits:
    def __init__(self, d):
        self.d, self.p = d, 0

    def get(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[self.p >> 3] >> (self.p & 7)) & 1) << i
            self.p += 1
        return v


def canon(lens):
    bl = collections.Counter(l for l in lens if l)
    code, nxt = 0, {}
    for b in range(1, 16):
        code = (code + bl.get(b - 1, 0)) << 1
        nxt[b] = code
    t = {}
    for s, l in enumerate(lens):
        if l:
            t[(l, nxt[l])] = s
            nxt[l] += 1
    return t


def sym(br, t):
    c = 0
    for l in range(1, 16):
        c = (c << 1) | br.get(1)
        if (l, c) in t:
            return t[(l, c)]
    raise ValueError("no code")


def tokens(data):
    """[(kind 'L'/'M', output bytes, distance, stream bits)] of a raw DEFLATE stream (RFC 1951)."""
    br, out = Bits(data), []
    while True:
        fin, ty = br.get(1), br.get(2)
        if ty == 0:
            br.p = (br.p + 7) & ~7
            n = br.get(16)
            br.get(16)
            br.p += 8 * n
            out.append(("S", n, 0, 0))
        else:
            if ty == 1:
                ll, dl = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8, [5] * 30
            else:
                hlit, hdist, hclen = br.get(5) + 257, br.get(5) + 1, br.get(4) + 4
                cl = [0] * 19
                for i in range(hclen):
                    cl[[16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15][i]] = br.get(3)
                ct, lens = canon(cl), []
                while len(lens) < hlit + hdist:
                    s = sym(br, ct)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + br.get(2))
                    elif s == 17:
                        lens += [0] * (3 + br.get(3))
                    else:
                        lens += [0] * (11 + br.get(7))
                ll, dl = lens[:hlit], lens[hlit:]
            lt, dt = canon(ll), canon(dl)
            while True:
                p0 = br.p
                s = sym(br, lt)
                if s < 256:
                    out.append(("L", 1, 0, br.p - p0))
                elif s == 256:
                    break
                else:
                    n = LBASE[s - 257] + br.get(LEXT[s - 257])
                    ds = sym(br, dt)
                    out.append(("M", n, DBASE[ds] + br.get(DEXT[ds]), br.p - p0))
        if fin:
            return out


def rounds_of(toks, rmax=16384, max_tokens=4096):
    pos, start, ntok, cur, out = 0, 0, 0, [], []
    for kind, n, dist, _ in toks:
        if pos + n - start > rmax or ntok >= max_tokens:
            out.append((start, pos, cur))
            cur, start, ntok = [], pos, 0
        if kind == "M":
            cur.append((pos, n, dist))
        pos += n
        ntok += 1
    out.append((start, pos, cur))
    return out


def iterations(toks, forward):
    """iterations of the resolution loop per round (whole matches; a self-overlapping match needs its first period)"""
    its, forwards = [], 0
    for start, end, ms in rounds_of(toks):
        size = end - start
        final = np.ones(size + 300, dtype=bool)
        owner = np.full(size + 300, -1, dtype=np.int32)
        for k, (p, n, d) in enumerate(ms):
            final[p - start:p - start + n] = False
            owner[p - start:p - start + n] = k
        src = [p - d for (p, n, d) in ms]
        done = [False] * len(ms)
        pending = set(range(len(ms)))
        n_it = 0
        while pending:
            n_it += 1
            ready, moved = [], []
            for k in pending:
                p, n, _ = ms[k]
                need = min(n, p - src[k])
                a, b = src[k] - start, src[k] - start + need
                if b <= 0 or final[max(a, 0):b].all():
                    ready.append(k)
                elif forward and a >= 0 and src[k] + need <= p:
                    k2 = owner[a]
                    if k2 >= 0 and (owner[a:b] == k2).all() and not done[k2] and src[k2] + ms[k2][1] <= ms[k2][0]:
                        moved.append((k, src[k2] + (src[k] - ms[k2][0])))
            for k in ready:
                p, n, _ = ms[k]
                final[p - start:p - start + n] = True
                done[k] = True
                pending.discard(k)
            for k, s in moved:
                if not done[k]:
                    src[k] = s
                    forwards += 1
        its.append(n_it)
    return its, forwards


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for idx in range(n):
        c = corpus.raw_deflate(corpus.gen_text(idx, 65536))
        t = tokens(c)
        lit = sum(1 for x in t if x[0] == "L")
        m = [x for x in t if x[0] == "M"]
        ml = np.array([x[1] for x in m])
        md = np.array([x[2] for x in m])
        print("entry %d: %d bytes compressed, %d tokens: %d literals (%.1f bits), %d matches (%.1f bits, %.1f bytes, %d bytes in all)" % (
            idx, len(c), len(t), lit, np.mean([x[3] for x in t if x[0] == "L"]), len(m), np.mean([x[3] for x in m]), ml.mean(), ml.sum()))
        print("   match lengths 3..8: %.0f %%, 9..16: %.0f %%, longer: %.0f %%;  distances < 1 KiB: %.0f %%, < 4 KiB: %.0f %%, < 16 KiB: %.0f %%" % (
            100 * (ml <= 8).mean(), 100 * ((ml > 8) & (ml <= 16)).mean(), 100 * (ml > 16).mean(),
            100 * (md < 1024).mean(), 100 * (md < 4096).mean(), 100 * (md < 16384).mean()))
        a, _ = iterations(t, False)
        b, f = iterations(t, True)
        print("   resolution loop, iterations per round: %s = %d;  with forwarding: %s = %d (%d forwards)" % (a, sum(a), b, sum(b), f))


if __name__ == "__main__":
    main()
