"""A model of Apple's LZFSE stream format (blocks bvx- / bvxn / bvx1 / bvx2 / bvx$, LZVN opcodes, the FSE coder of
LZFSE) written from the published description of the format: an encoder that writes such streams and a decoder
that reads them, in plain Python.  TEST / TOOL INFRASTRUCTURE, never imported by the product.

PARITY UNPINNED.  No implementation of LZFSE exists on the image (no liblzfse, no Python module, no test vectors),
and the reference implements a private framing under method 100, not this format (ref src/lib/lzfse.inc.c:225-257).
So this model is the only checker the GPU decoder (otezip_amd/csrc/otz_lzfse.h) has: the two were written
separately (Python here, one work-item per stream there) from the same reading of the format, and agreeing with
each other shows that the kernel does what the model says -- not that the model is what Apple's library does.

Format, as modelled:
  stream   = block* 'bvx$'
  'bvx-'   u32 n_raw_bytes, the bytes
  'bvxn'   u32 n_raw_bytes, u32 n_payload_bytes, LZVN opcodes ending in 06 00 00 00 00 00 00 00
  'bvx2'   u32 n_raw_bytes, 3 x u64 packed fields, the four frequency tables in a variable-length code,
           the literal payload and the L/M/D payload (both read backwards by an FSE decoder)
  'bvx1'   the same block with the header fields and frequency tables written out in full
"""
import struct

# ------------------------------------------------------------------ constants of the format

L_STATES, M_STATES, D_STATES, LIT_STATES = 64, 64, 256, 1024
L_SYMS, M_SYMS, D_SYMS, LIT_SYMS = 20, 20, 64, 256
L_EXTRA = [0] * 16 + [2, 3, 5, 8]
L_BASE = list(range(16)) + [16, 20, 28, 60]
M_EXTRA = [0] * 16 + [3, 5, 8, 11]
M_BASE = list(range(16)) + [16, 24, 56, 312]
D_EXTRA = [k // 4 for k in range(64)]
D_BASE = []
_b = 0
for _k in range(64):
    D_BASE.append(_b)
    _b += 1 << D_EXTRA[_k]
FREQ_NBITS = [2, 3, 2, 5, 2, 3, 2, 8, 2, 3, 2, 5, 2, 3, 2, 14, 2, 3, 2, 5, 2, 3, 2, 8, 2, 3, 2, 5, 2, 3, 2, 14]
FREQ_VALUE = [0, 2, 1, 4, 0, 3, 1, -1, 0, 2, 1, 5, 0, 3, 1, -1, 0, 2, 1, 6, 0, 3, 1, -1, 0, 2, 1, 7, 0, 3, 1, -1]
MATCHES_PER_BLOCK = 10000
LITERALS_PER_BLOCK = 4 * MATCHES_PER_BLOCK


class FormatError(Exception):
    pass


def clz32(x):
    return 32 - x.bit_length()


# ------------------------------------------------------------------ FSE tables (decoder side; the encoder inverts them)

def fse_table(nstates, freq):
    """[(k, symbol, delta)] in state order; raises when the frequencies exceed the state count."""
    table = []
    n_clz = clz32(nstates)
    for sym, f in enumerate(freq):
        if f == 0:
            continue
        if len(table) + f > nstates:
            raise FormatError("frequencies exceed the state count")
        k = clz32(f) - n_clz
        j0 = ((2 * nstates) >> k) - f
        for j in range(f):
            if j < j0:
                table.append((k, sym, ((f + j) << k) - nstates))
            else:
                table.append((k - 1, sym, (j - j0) << (k - 1)))
    return table


class BitsBackward:
    """The decoder's view of a payload: read from its end towards its start.  The reads are 8 bytes wide and may
    reach in front of the payload (into the bytes of the stream that precede it: never consumed), but not in
    front of the stream."""

    def __init__(self, stream, end, nbits_param):
        # nbits_param in [-7, 0]: minus the number of unused bits at the top of the payload's last byte
        take = 8 if nbits_param else 7
        if end < take:
            raise FormatError("payload too close to the start of the stream")
        self.pos = end - take
        self.accum = int.from_bytes(stream[self.pos:self.pos + take], "little")
        self.nbits = nbits_param + 64 if nbits_param else 56
        self.buf = stream
        if not (56 <= self.nbits < 64) or (self.accum >> self.nbits):
            raise FormatError("bad bit count of a payload")

    def flush(self):
        n = (63 - self.nbits) & ~7
        self.pos -= n >> 3
        if self.pos < 0:
            raise FormatError("payload over-read")
        incoming = int.from_bytes(self.buf[self.pos:self.pos + 8].ljust(8, b"\0"), "little")
        self.accum = ((self.accum << n) | (incoming & ((1 << n) - 1))) & ((1 << 64) - 1)
        self.nbits += n

    def pull(self, n):
        if n > self.nbits:
            raise FormatError("bit reader underflow")
        self.nbits -= n
        v = self.accum >> self.nbits
        self.accum &= (1 << self.nbits) - 1
        return v


# ------------------------------------------------------------------ decoder

def _decode_freq_v2(src):
    """360 frequencies from the variable-length code behind a v2 header."""
    out = []
    accum = nb = at = 0
    for _ in range(L_SYMS + M_SYMS + D_SYMS + LIT_SYMS):
        while at < len(src) and nb + 8 <= 32:
            accum |= src[at] << nb
            at += 1
            nb += 8
        b = accum & 31
        n = FREQ_NBITS[b]
        if n == 8:
            v = 8 + ((accum >> 4) & 0xF)
        elif n == 14:
            v = 24 + ((accum >> 4) & 0x3FF)
        else:
            v = FREQ_VALUE[b]
        if n > nb:
            raise FormatError("frequency tables truncated")
        out.append(v)
        accum >>= n
        nb -= n
    return out


def _decode_lzfse_block(stream, body_at, hdr, out, n_raw):
    """hdr: dict with the fields of a v1 header; the two payloads start at stream[body_at]."""
    lit_tab = fse_table(LIT_STATES, hdr["literal_freq"])
    n_lit = hdr["n_literals"]
    if n_lit > LITERALS_PER_BLOCK or n_lit % 4 or hdr["n_matches"] > MATCHES_PER_BLOCK:
        raise FormatError("block counts out of range")
    lp, mp = hdr["n_literal_payload_bytes"], hdr["n_lmd_payload_bytes"]
    if body_at + lp + mp > len(stream):
        raise FormatError("payload truncated")
    lits = bytearray()
    if n_lit:
        br = BitsBackward(stream, body_at + lp, hdr["literal_bits"])
        st = list(hdr["literal_state"])
        if any(s >= LIT_STATES for s in st):
            raise FormatError("literal state out of range")
        for _ in range(n_lit // 4):
            br.flush()
            for k in range(4):
                if st[k] >= len(lit_tab):
                    raise FormatError("literal state without a table entry")
                kk, sym, delta = lit_tab[st[k]]
                st[k] = delta + br.pull(kk)
                lits.append(sym)
    l_tab, m_tab, d_tab = fse_table(L_STATES, hdr["l_freq"]), fse_table(M_STATES, hdr["m_freq"]), fse_table(D_STATES, hdr["d_freq"])
    start = len(out)
    lit_at = 0
    D = 0
    if hdr["n_matches"]:
        br = BitsBackward(stream, body_at + lp + mp, hdr["lmd_bits"])
        ls, ms, ds = hdr["l_state"], hdr["m_state"], hdr["d_state"]

        def value(tab, state, extra, base):
            if state >= len(tab):
                raise FormatError("state without a table entry")
            k, sym, delta = tab[state]
            vb = extra[sym]
            sv = br.pull(k + vb)
            return delta + (sv >> vb), base[sym] + (sv & ((1 << vb) - 1))
        for _ in range(hdr["n_matches"]):
            br.flush()
            ls, L = value(l_tab, ls, L_EXTRA, L_BASE)
            br.flush()
            ms, M = value(m_tab, ms, M_EXTRA, M_BASE)
            br.flush()
            ds, nd = value(d_tab, ds, D_EXTRA, D_BASE)
            D = nd if nd else D
            if lit_at + L > len(lits):
                raise FormatError("more literals used than the block holds")
            out += lits[lit_at:lit_at + L]
            lit_at += L
            if M:
                if D == 0 or D > len(out):
                    raise FormatError("match distance outside the output")
                for _ in range(M):
                    out.append(out[-D])
            if len(out) - start > n_raw:
                raise FormatError("block longer than it says")
    if len(out) - start != n_raw:
        raise FormatError("block shorter than it says")


def decode_lzvn(src, out, n_raw):
    """LZVN opcodes up to the end-of-stream opcode; returns the bytes consumed."""
    at = 0
    start = len(out)
    D = 0

    def need(n):
        if at + n > len(src):
            raise FormatError("LZVN payload truncated")

    def copy(L, M, dist):
        nonlocal at
        need(L)
        out.extend(src[at:at + L])
        at += L
        if M:
            if dist == 0 or dist > len(out):
                raise FormatError("LZVN distance outside the output")
            for _ in range(M):
                out.append(out[-dist])
        if len(out) - start > n_raw:
            raise FormatError("LZVN block longer than it says")
    while True:
        need(1)
        opc = src[at]
        if opc == 0x06:
            need(8)
            at += 8
            break
        if opc in (0x0E, 0x16):
            at += 1
            continue
        if opc in (0x1E, 0x26, 0x2E, 0x36, 0x3E) or 0x70 <= opc <= 0x7F or 0xD0 <= opc <= 0xDF:
            raise FormatError("undefined LZVN opcode")
        if opc >= 0xF0:  # match with the previous distance
            if opc == 0xF0:
                need(2)
                M = src[at + 1] + 16
                at += 2
            else:
                M = opc & 15
                at += 1
            copy(0, M, D)
            continue
        if opc >= 0xE0:  # literals
            if opc == 0xE0:
                need(2)
                L = src[at + 1] + 16
                at += 2
            else:
                L = opc & 15
                at += 1
            copy(L, 0, D)
            continue
        if 0xA0 <= opc <= 0xBF:  # medium distance
            need(3)
            L = (opc >> 3) & 3
            M = (((opc & 7) << 2) | (src[at + 1] & 3)) + 3
            D = (src[at + 1] >> 2) | (src[at + 2] << 6)
            at += 3
            copy(L, M, D)
            continue
        L = opc >> 6
        M = ((opc >> 3) & 7) + 3
        low = opc & 7
        if low == 7:  # large distance
            need(3)
            D = src[at + 1] | (src[at + 2] << 8)
            at += 3
        elif low == 6:  # previous distance
            at += 1
        else:  # small distance
            need(2)
            D = (low << 8) | src[at + 1]
            at += 2
        copy(L, M, D)
    if len(out) - start != n_raw:
        raise FormatError("LZVN block shorter than it says")
    return at


def decode(stream):
    """The whole stream; raises FormatError on anything the model does not accept."""
    out = bytearray()
    at = 0
    while True:
        if at + 4 > len(stream):
            raise FormatError("stream truncated")
        magic = stream[at:at + 4]
        if magic == b"bvx$":
            return bytes(out)
        if magic == b"bvx-":
            if at + 8 > len(stream):
                raise FormatError("header truncated")
            n = struct.unpack_from("<I", stream, at + 4)[0]
            if at + 8 + n > len(stream):
                raise FormatError("raw block truncated")
            out += stream[at + 8:at + 8 + n]
            at += 8 + n
        elif magic == b"bvxn":
            if at + 12 > len(stream):
                raise FormatError("header truncated")
            n_raw, n_pay = struct.unpack_from("<II", stream, at + 4)
            if at + 12 + n_pay > len(stream):
                raise FormatError("LZVN block truncated")
            used = decode_lzvn(stream[at + 12:at + 12 + n_pay], out, n_raw)
            if used != n_pay:
                raise FormatError("LZVN payload length mismatch")
            at += 12 + n_pay
        elif magic == b"bvx2":
            if at + 32 > len(stream):
                raise FormatError("header truncated")
            n_raw = struct.unpack_from("<I", stream, at + 4)[0]
            v0, v1, v2 = struct.unpack_from("<QQQ", stream, at + 8)

            def f(v, o, n):
                return (v >> o) & ((1 << n) - 1)
            hdr = dict(n_literals=f(v0, 0, 20), n_literal_payload_bytes=f(v0, 20, 20), n_matches=f(v0, 40, 20),
                       literal_bits=f(v0, 60, 3) - 7, literal_state=[f(v1, 10 * k, 10) for k in range(4)],
                       n_lmd_payload_bytes=f(v1, 40, 20), lmd_bits=f(v1, 60, 3) - 7,
                       l_state=f(v2, 32, 10), m_state=f(v2, 42, 10), d_state=f(v2, 52, 10))
            hsize = f(v2, 0, 32)
            if hsize < 32 or at + hsize > len(stream):
                raise FormatError("bad header size")
            fr = _decode_freq_v2(stream[at + 32:at + hsize]) if hsize > 32 else [0] * 360
            hdr["l_freq"], hdr["m_freq"] = fr[:20], fr[20:40]
            hdr["d_freq"], hdr["literal_freq"] = fr[40:104], fr[104:360]
            _decode_lzfse_block(stream, at + hsize, hdr, out, n_raw)
            at += hsize + hdr["n_literal_payload_bytes"] + hdr["n_lmd_payload_bytes"]
        elif magic == b"bvx1":
            size = 4 * 7 + 4 + 8 + 4 + 6 + 2 * 360
            # magic, n_raw, n_payload, n_literals, n_matches, n_literal_payload, n_lmd_payload (u32 each);
            # literal_bits (i32), literal_state[4] (u16); lmd_bits (i32), l/m/d state (u16); the four tables (u16 each)
            if at + size > len(stream):
                raise FormatError("header truncated")
            n_raw, n_pay, n_lit, n_mat, lp, mp, lbits = struct.unpack_from("<IIIIIIi", stream, at + 4)
            lst = struct.unpack_from("<4H", stream, at + 32)
            mbits, ls, ms, ds = struct.unpack_from("<iHHH", stream, at + 40)
            fr = struct.unpack_from("<360H", stream, at + 50)
            hdr = dict(n_literals=n_lit, n_literal_payload_bytes=lp, n_matches=n_mat, literal_bits=lbits,
                       literal_state=list(lst), n_lmd_payload_bytes=mp, lmd_bits=mbits, l_state=ls, m_state=ms, d_state=ds,
                       l_freq=list(fr[:20]), m_freq=list(fr[20:40]), d_freq=list(fr[40:104]), literal_freq=list(fr[104:]))
            if not (-7 <= lbits <= 0 and -7 <= mbits <= 0):
                raise FormatError("bad bit count")
            _decode_lzfse_block(stream, at + size, hdr, out, n_raw)
            at += size + lp + mp
        else:
            raise FormatError("unknown block magic")


# ------------------------------------------------------------------ encoder (for tests: simple, greedy)

def _matches(data, min_len=4, max_dist=65535, max_len=2000, hash_bits=14):
    """Greedy LZ77 parse: [(L, M, D)] with the literal bytes implied; a final (L, 0, D) carries trailing literals."""
    n = len(data)
    head = {}
    out = []
    i = lit0 = 0
    while i + min_len <= n:
        key = data[i:i + 4]
        cand = head.get(key)
        head[key] = i
        if cand is not None and i - cand <= max_dist:
            m = 0
            while i + m < n and data[cand + m] == data[i + m] and m < max_len:
                m += 1
            if m >= min_len:
                out.append((i - lit0, m, i - cand))
                for k in range(1, min(m, 64)):
                    head[data[i + k:i + k + 4]] = i + k
                i += m
                lit0 = i
                continue
        i += 1
    if lit0 < n or not out:
        out.append((n - lit0, 0, out[-1][2] if out else 1))
    return out


def encode_raw(data):
    return b"bvx-" + struct.pack("<I", len(data)) + data


def encode_lzvn(data):
    ops = bytearray()
    at = 0
    prev_d = 0
    for L, M, D in _matches(data, min_len=3, max_dist=65535, max_len=10 ** 9):
        lit = data[at:at + L]
        at += L + M
        # literals beyond what a match opcode carries
        while len(lit) > 3 or (M == 0 and lit):
            if len(lit) >= 16:
                n = min(len(lit), 16 + 255)
                ops += bytes([0xE0, n - 16]) + lit[:n]
            else:
                n = len(lit) if M == 0 or len(lit) > 3 else 0
                if n == 0:
                    break
                ops += bytes([0xE0 | n]) + lit[:n]
            lit = lit[n:]
        if M == 0:
            continue
        Lx = len(lit)
        x = 10 - 2 * Lx
        m0 = min(M, x)
        if D == prev_d and Lx == 0:
            m0 = 0  # no literal, same distance: match-only opcodes below
        elif D == prev_d:
            ops += bytes([(Lx << 6) | ((m0 - 3) << 3) | 6]) + lit
        elif D < 0x600:
            ops += bytes([(Lx << 6) | ((m0 - 3) << 3) | (D >> 8), D & 255]) + lit
        elif D < 0x4000 and M >= 3:
            m0 = min(M, 34)
            ops += bytes([0xA0 | (Lx << 3) | ((m0 - 3) >> 2), ((m0 - 3) & 3) | ((D & 0x3F) << 2), D >> 6]) + lit
        else:
            ops += bytes([(Lx << 6) | ((m0 - 3) << 3) | 7, D & 255, D >> 8]) + lit
        prev_d = D
        rest = M - m0
        while rest:
            if rest >= 16:
                n = min(rest, 16 + 255)
                ops += bytes([0xF0, n - 16])
            else:
                n = rest
                ops += bytes([0xF0 | n])
            rest -= n
    ops += b"\x06" + bytes(7)
    return b"bvxn" + struct.pack("<II", len(data), len(ops)) + bytes(ops)


class _BitsForward:
    def __init__(self):
        self.acc = self.n = 0
        self.out = bytearray()

    def push(self, v, n):
        self.acc |= v << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def finish(self):
        """(payload, bit-count field)"""
        r = self.n
        if r:
            self.out.append(self.acc & 255)
        return bytes(self.out), (r - 8 if r else 0)


def _normalise(counts, nstates):
    total = sum(counts)
    if total == 0:
        return [0] * len(counts)
    f = [max(1, c * nstates // total) if c else 0 for c in counts]
    while sum(f) > nstates:
        k = max(range(len(f)), key=lambda i: f[i])
        f[k] -= 1
    k = max(range(len(f)), key=lambda i: f[i])
    f[k] += nstates - sum(f)
    return f


class _Enc:
    """Inverse of fse_table(): encode symbol `sym` so that the decoder, afterwards, is in state y."""

    def __init__(self, nstates, freq):
        self.n = nstates
        self.t0 = {}
        t = 0
        n_clz = clz32(nstates)
        self.par = {}
        for s, f in enumerate(freq):
            if f:
                k = clz32(f) - n_clz
                self.t0[s] = t
                self.par[s] = (f, k, ((2 * nstates) >> k) - f)
                t += f

    def step(self, sym, y):
        """(new state = table slot, bits, nbits)"""
        f, k, j0 = self.par[sym]
        low = (f - j0) << (k - 1) if k > 0 else 0  # states served by the slots j >= j0
        if y < low:
            j = j0 + (y >> (k - 1))
            return self.t0[sym] + j, y - ((j - j0) << (k - 1)), k - 1
        j = ((y + self.n) >> k) - f
        return self.t0[sym] + j, y - (((f + j) << k) - self.n), k


def _code_of(v, base, extra):
    for s in range(len(base) - 1, -1, -1):
        if v >= base[s]:
            if v - base[s] >= (1 << extra[s]):
                raise ValueError("value out of the code's range")
            return s, v - base[s]
    raise ValueError


def _freq_bits(v):
    if v == 0:
        return 0, 2
    if v == 1:
        return 2, 2
    if v == 2:
        return 1, 3
    if v == 3:
        return 5, 3
    if v < 8:
        return 3 | ((v - 4) << 3), 5
    if v < 24:
        return 7 | ((v - 8) << 4), 8
    return 15 | ((v - 24) << 4), 14


def encode_lzfse(data, v1=False):
    """One or more bvx2 (or bvx1) blocks for data."""
    out = b""
    trip = _matches(data, min_len=4, max_dist=D_BASE[63] + (1 << D_EXTRA[63]) - 1, max_len=M_BASE[19] + (1 << 11) - 1)
    # split long literal runs (L code tops out at 315) and cut into blocks
    seq = []
    at = 0
    for L, M, D in trip:
        while L > 315:
            seq.append((315, 0, D, at))
            at += 315
            L -= 315
        seq.append((L, M, D, at))
        at += L + M
    i = 0
    while i < len(seq):
        blk = []
        nl = 0
        while i < len(seq) and len(blk) < MATCHES_PER_BLOCK and nl + seq[i][0] <= LITERALS_PER_BLOCK - 4:
            blk.append(seq[i])
            nl += seq[i][0]
            i += 1
        out += _encode_block(data, blk, v1)
    return out


def _encode_block(data, blk, v1):
    lits = bytearray()
    n_raw = 0
    for L, M, D, at in blk:
        lits += data[at:at + L]
        n_raw += L + M
    while len(lits) % 4:
        lits.append(0)
    lcodes, mcodes, dcodes = [], [], []
    prev = 0
    for L, M, D, _ in blk:
        lcodes.append(_code_of(L, L_BASE, L_EXTRA))
        mcodes.append(_code_of(M, M_BASE, M_EXTRA))
        nd = 0 if (D == prev and prev != 0) or M == 0 and prev != 0 else D
        if M == 0 and prev == 0:
            nd = D  # (the first distance must be spelt out)
        dcodes.append(_code_of(nd, D_BASE, D_EXTRA))
        if nd:
            prev = nd

    def hist(codes, n):
        h = [0] * n
        for s, _ in codes:
            h[s] += 1
        return h
    lf, mf, df = _normalise(hist(lcodes, L_SYMS), L_STATES), _normalise(hist(mcodes, M_SYMS), M_STATES), _normalise(hist(dcodes, D_SYMS), D_STATES)
    ch = [0] * 256
    for b in lits:
        ch[b] += 1
    cf = _normalise(ch, LIT_STATES)
    # literals: the decoder reads lit[i] with state i % 4, groups in order; encode in reverse
    w = _BitsForward()
    enc = _Enc(LIT_STATES, cf)
    st = [0, 0, 0, 0]
    for i in range(len(lits) - 1, -1, -1):
        s, bits, nb = enc.step(lits[i], st[i % 4])
        w.push(bits, nb)
        st[i % 4] = s
    lit_payload, lit_bits = w.finish() if lits else (b"", 0)
    w = _BitsForward()
    le, me, de = _Enc(L_STATES, lf), _Enc(M_STATES, mf), _Enc(D_STATES, df)
    ls = ms = ds = 0
    for (lc, lx), (mc, mx), (dc, dx) in zip(reversed(lcodes), reversed(mcodes), reversed(dcodes)):
        s, bits, nb = de.step(dc, ds)
        w.push((bits << D_EXTRA[dc]) | dx, nb + D_EXTRA[dc])
        ds = s
        s, bits, nb = me.step(mc, ms)
        w.push((bits << M_EXTRA[mc]) | mx, nb + M_EXTRA[mc])
        ms = s
        s, bits, nb = le.step(lc, ls)
        w.push((bits << L_EXTRA[lc]) | lx, nb + L_EXTRA[lc])
        ls = s
    lmd_payload, lmd_bits = w.finish()
    freqs = lf + mf + df + cf
    if v1:
        hdr = b"bvx1" + struct.pack("<IIIIIIi", n_raw, len(lit_payload) + len(lmd_payload), len(lits), len(blk), len(lit_payload),
                                    len(lmd_payload), lit_bits) + struct.pack("<4H", *st) + struct.pack("<iHHH", lmd_bits, ls, ms, ds) + \
            struct.pack("<360H", *freqs)
        return hdr + lit_payload + lmd_payload
    fw = _BitsForward()
    for v in freqs:
        b, n = _freq_bits(v)
        fw.push(b, n)
    if fw.n:
        fw.out.append(fw.acc & 255)
    ftab = bytes(fw.out)
    hsize = 32 + len(ftab)
    v0 = len(lits) | (len(lit_payload) << 20) | (len(blk) << 40) | ((lit_bits + 7) << 60)
    v1_ = st[0] | (st[1] << 10) | (st[2] << 20) | (st[3] << 30) | (len(lmd_payload) << 40) | ((lmd_bits + 7) << 60)
    v2 = hsize | (ls << 32) | (ms << 42) | (ds << 52)
    return b"bvx2" + struct.pack("<IQQQ", n_raw, v0, v1_, v2) + ftab + lit_payload + lmd_payload


def encode(data, kind="lzfse"):
    """A whole stream: kind in raw / lzvn / lzfse / lzfse_v1 / mixed."""
    if kind == "raw":
        body = encode_raw(data)
    elif kind == "lzvn":
        body = encode_lzvn(data)
    elif kind == "lzfse":
        body = encode_lzfse(data)
    elif kind == "lzfse_v1":
        body = encode_lzfse(data, v1=True)
    else:  # mixed: the three block kinds behind one another
        a, b = len(data) // 3, 2 * len(data) // 3
        body = encode_lzvn(data[:a]) + encode_raw(data[a:b]) + encode_lzfse(data[b:])
    return body + b"bvx$"
