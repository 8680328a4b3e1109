#!/usr/bin/env python3
"""LDS bank-conflict model of the output column kernel (csrc/fast_cols.hpp, tiled intermediate), per tile.

Rules from /opt/skills/guides/MI355X_MICROARCH.md (LDS [CDNA4]): a wave64 access is served in fixed lane groups, one LDS
cycle per group when conflict-free; every further distinct address on a busy bank of a group adds a cycle
(SQ_LDS_BANK_CONFLICT = the extra cycles, SQ_LDS_IDX_ACTIVE = all of them).
    ds_read_b64    2 x 32 lanes             bank = dword % 64
    ds_read_b128   4 x 16 lanes (listed)    bank = dword % 64
    ds_read_b32    2 x 32 lanes             bank = dword % 32
    ds_write_b64   4 x 16 contiguous lanes  bank = dword % 32
    ds_write_b128  8 x  8 contiguous lanes  bank = dword % 32
Prints cycles and conflict cycles per phase for a configuration, for block paddings PAD (stage-1 blocks S1 = m1 + PAD cells
apart) and column pitches, so that a layout can be chosen before it is built.

usage: lds_bank_model.py M R1 R2 R3 T NT [PAD ...]   |   lds_bank_model.py --search
"""
import sys

G_R64 = [list(range(0, 32)), list(range(32, 64))]
G_R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G_R128 = G_R128 + [[l + 32 for l in g] for g in G_R128]
G_W64 = [list(range(i, i + 16)) for i in range(0, 64, 16)]
G_W128 = [list(range(i, i + 8)) for i in range(0, 64, 8)]
KINDS = {"r64": (G_R64, 64, 2), "r128": (G_R128, 64, 4), "r32": (G_R64, 32, 1), "w64": (G_W64, 32, 2), "w128": (G_W128, 32, 4)}


class Tally:
    def __init__(self):
        self.cyc = {}
        self.extra = {}

    def access(self, phase, kind, addrs):
        """addrs: list over the NT threads of a cell address (c32 units, may be fractional dwords via tuple) or None (lane off)."""
        groups, nb, dw = KINDS[kind]
        cyc = extra = 0
        for w0 in range(0, len(addrs), 64):
            wave = addrs[w0:w0 + 64]
            for g in groups:
                banks = {}
                for l in g:
                    if l >= len(wave) or wave[l] is None:
                        continue
                    d0 = wave[l]
                    for d in range(d0, d0 + dw):
                        banks.setdefault(d % nb, set()).add(d)
                if banks:
                    m = max(len(s) for s in banks.values())
                    cyc += m
                    extra += m - 1
        self.cyc[phase] = self.cyc.get(phase, 0) + cyc
        self.extra[phase] = self.extra.get(phase, 0) + extra


def pair_of_unit(u, M, R1, R2, R3, T, transpose=2, two_level=False):
    NB3 = R1 * R2
    m1 = M // R1
    NP = M // 2 + 1
    if T == 4 and transpose:
        SB = 8 * NB3
        FULL = (NP // SB) * SB if (2 * R3) % 32 == 0 else 0
        if u < FULL:
            sb, i = divmod(u, SB)
            return sb * SB + (i & 7) * NB3 + (i >> 3)
        return ((u & ~63) | ((u & 7) << 3) | ((u >> 3) & 7)) if (u | 63) < NP else u
    if T == 8 and transpose >= 2 and (2 * m1) % 32 == 0:
        SB = 4 * NB3
        FULL = (NP // SB) * SB
        if u < FULL:
            sb, i = divmod(u, SB)
            return sb * SB + (i & 3) * NB3 + (i >> 2)
        if two_level and u < FULL + ((NP - FULL) // (2 * NB3)) * 2 * NB3:
            sb, i = divmod(u - FULL, 2 * NB3)
            return FULL + sb * 2 * NB3 + (i & 1) * NB3 + (i >> 1)
        return u
    return u


def model(M, R1, R2, R3, T, NT, PAD, pitch_mod=(16, 2), transpose=2, verbose=True, rot=0, by_unit=False, two_level=False, parts=("landing", "s3", "s2", "s1"), rot_all=False):
    m1 = M // R1
    S1 = m1 + PAD
    MP = R1 * S1                      # padded column image without the Nyquist slot
    q, r = pitch_mod
    LP = MP + 1
    while LP % q != r:
        LP += 1
    NB1, NB2, NB3 = m1, R1 * R3, R1 * R2
    cell = lambda p: p + (p // m1) * PAD if p < M else MP

    def pos(k):
        return (k % R1) * m1 + ((k // R1) % R2) * R3 + k // (R1 * R2)

    OFF_T2 = T * LP
    OFF_T1 = OFF_T2 + (R2 - 1) * R3
    OFF_WH = OFF_T1 + m1
    NPE = M // 2 + 1
    OFF_WL = OFF_WH + (NPE + 31) // 32
    OFF_PAIR = OFF_WL + 32
    total = OFF_PAIR + (NPE + 1) // 2 + 2
    ta = Tally()
    D = lambda c: 2 * c    # cell -> dword

    # landing (C5)
    T2 = T // 2
    NPU = NPE * T2
    for r_ in range((NPU + NT - 1) // NT if "landing" in parts else 0):
        ks, za, zb, z1a, z1b, whs, wls, pps = [], [], [], [], [], [], [], []
        for t in range(NT):
            e = t + NT * r_
            if e >= NPU:
                for lst in (za, zb, z1a, z1b, whs, wls, pps):
                    lst.append(None)
                continue
            k = pair_of_unit(e // T2, M, R1, R2, R3, T, transpose, two_level)
            t2 = e % T2
            pa = cell(pos(k))
            pb = cell(pos(M - k)) if 0 < k else MP
            z0 = (2 * t2) * LP
            pps.append(D(OFF_PAIR) + (e // T2 if by_unit else k))
            whs.append(D(OFF_WH + (k >> 5)))
            wls.append(D(OFF_WL + (k & 31)))
            za.append(D(z0 + pa))
            z1a.append(D(z0 + LP + pa))
            two = (k != 0 and 2 * k != M)
            zb.append(D(z0 + pb) if two else None)
            z1b.append(D(z0 + LP + pb) if two else None)
        ta.access("landing: table reads", "r32", pps)
        ta.access("landing: table reads", "r64", whs)
        ta.access("landing: table reads", "r64", wls)
        for lst in (za, zb, z1a, z1b):
            ta.access("landing: stores", "w64", lst)

    # C2: stage 3
    for h in range(R3 // 2 if "s3" in parts else 0):
        a = []
        for t in range(NT):
            col, qq = divmod(t, NB3)
            blk, c = divmod(qq, R2)
            if rot and (rot_all or (blk & 1)):
                c = (c + rot * (blk if rot_all else 1)) % R2
            a.append(D(col * LP + blk * S1 + c * R3 + 2 * h))
        ta.access("stage 3: reads", "r128", a)
        ta.access("stage 3: writes", "w128", a)

    # C3: stage 2
    for r_ in range((NB2 * T + NT - 1) // NT if "s2" in parts else 0):
        base, bs = [], []
        for t in range(NT):
            idx = t + NT * r_
            if idx >= NB2 * T:
                base.append(None)
                bs.append(None)
                continue
            col, u = divmod(idx, NB2)
            c1, b = divmod(u, R3)
            base.append(col * LP + c1 * S1 + b)
            bs.append(b)
        for c in range(R2):
            ta.access("stage 2: reads", "r64", [None if x is None else D(x + c * R3) for x in base])
            if c:
                ta.access("stage 2: twiddles", "r64", [None if b is None else D(OFF_T2 + (c - 1) * R3 + b) for b in bs])
            ta.access("stage 2: writes", "w64", [None if x is None else D(x + c * R3) for x in base])

    # C4: stage 1
    for r_ in range((NB1 * T + NT - 1) // NT if "s1" in parts else 0):
        base, js = [], []
        for t in range(NT):
            idx = t + NT * r_
            if idx >= NB1 * T:
                base.append(None)
                js.append(None)
                continue
            col, j = divmod(idx, NB1)
            base.append(col * LP + j)
            js.append(j)
        ta.access("stage 1: twiddle", "r64", [None if j is None else D(OFF_T1 + j) for j in js])
        for c in range(R1):
            ta.access("stage 1: reads", "r64", [None if x is None else D(x + c * S1) for x in base])

    tc = sum(ta.cyc.values())
    te = sum(ta.extra.values())
    if verbose:
        print("M %d = %d.%d.%d  T %d  NT %d  PAD %d  S1 %d  LP %d  LDS %d B (%s)  transpose %d  rot %d  by_unit %d  two_level %d" %
              (M, R1, R2, R3, T, NT, PAD, S1, LP, total * 8, "fits" if total * 8 <= 160 * 1024 else "TOO BIG", transpose, rot, by_unit, two_level))
        for k in ta.cyc:
            print("    %-24s cycles %7d  conflict %7d  (%.0f %%)" % (k, ta.cyc[k], ta.extra[k], 100.0 * ta.extra[k] / ta.cyc[k]))
        print("    %-24s cycles %7d  conflict %7d  (%.1f %%)" % ("total", tc, te, 100.0 * te / tc))
    return tc, te, total * 8


CONFIGS = """4224 8 24 22 4 768; 3840 8 24 20 4 768; 3520 10 16 22 4 640; 3072 8 32 12 4 1024; 2816 8 16 22 4 512; 2560 8 32 10 4 1024; 2304 8 16 18 8 1024;
2112 6 16 22 8 768; 2080 8 13 20 8 832; 1920 8 12 20 8 768; 1760 5 16 22 8 640; 1680 6 20 14 8 960; 1536 8 16 12 8 1024; 1408 8 8 22 8 512; 1280 8 16 10 8 1024;
1152 8 12 12 8 768; 1056 6 8 22 16 768; 960 6 8 20 16 768; 880 5 8 22 16 640; 768 8 8 12 16 1024; 672 6 8 14 16 768; 640 8 8 10 16 1024; 576 6 8 12 16 768;
544 17 4 8 8 544; 480 6 8 10 16 768; 432 6 6 12 16 576; 384 4 8 12 16 512; 336 4 6 14 16 384; 288 4 6 12 16 384; 240 4 6 10 16 384; 192 4 8 6 16 512; 144 4 6 6 16 384"""


def search_one(c):
    """best (pad, rot) of one configuration: fewest modelled LDS cycles per tile among the paddings that fit 160 KiB"""
    dense = model(*c, 0, verbose=False)
    rows = []
    for pad in range(0, 48, 2):
        tc, te, b = model(*c, pad, verbose=False, by_unit=True, two_level=True, parts=("landing", "s2", "s1"))
        if b > 160 * 1024:
            break
        t3, rot = min((model(*c, pad, verbose=False, rot=rot, parts=("s3",))[0], rot) for rot in range(c[2]))
        rows.append((tc + t3, pad, rot, b))
    return c, dense, min(rows)


def search():
    """--search: the FC_COL_LAYOUTS list of csrc/fast_cols.hpp (configurations: fast_paths.hpp FC_FAST_COL_CONFIGS)"""
    from multiprocessing import Pool
    cfgs = [tuple(int(x) for x in c.split()) for c in CONFIGS.replace("\n", " ").split(";")]
    with Pool(8) as p:
        for c, dense, (cyc, pad, rot, b) in p.imap(search_one, cfgs):
            print("    X(%d, %d, %d, %d, %d, %d, %d) /* LDS cycles per tile %d -> %d, %d bytes */ \\" % (c[0], c[1], c[2], c[3], c[4], pad, rot, dense[0], cyc, b))


if __name__ == "__main__":
    if sys.argv[1:] == ["--search"]:
        search()
        sys.exit(0)
    a = [int(x) for x in sys.argv[1:]]
    M, R1, R2, R3, T, NT = a[:6]
    pads = a[6:] or [0]
    for p in pads:
        model(M, R1, R2, R3, T, NT, p)
