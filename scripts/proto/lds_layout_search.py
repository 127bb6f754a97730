"""Brute-force bank-conflict counts of the features_v3.hip LDS layouts under the gfx950 banking rules
(MI355X_MICROARCH.md, LDS table): ds_write_b32 2 x 32 lanes, bank = dword mod 32; ds_write_b64 4 x 16 contiguous lanes,
bank = dword mod 32; ds_read_b128 4 x 16 lanes {0-3,12-15,20-27} ..., bank = dword mod 64.  Cost of a group = the
largest number of DISTINCT dword addresses that share a bank (identical addresses broadcast)."""
import itertools, sys

B128_GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
               [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59], [36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]

def cost(groups, addr_dwords, nbanks):
    """addr_dwords[lane] = list of dword addresses the lane touches; returns total cycles over the groups."""
    tot = 0
    for grp in groups:
        banks = {}
        for l in grp:
            for a in addr_dwords[l]:
                banks.setdefault(a % nbanks, set()).add(a)
        tot += max(len(v) for v in banks.values())
    return tot

W32 = [list(range(0, 32)), list(range(32, 64))]
W64 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]

def p_write_cost(S, xf, c, k2g_of):
    # lane (f, g) writes bin = c + 32 * k2g(g); physical dword = f * S + (bin ^ xf[f])
    addr = []
    for l in range(64):
        f, g = l & 15, l >> 4
        b = c + 32 * k2g_of(g)
        addr.append([f * S + (b ^ xf[f])])
    return cost(W32, addr, 32)

def p_read_cost(S, xf, bin0):
    addr = []
    for l in range(64):
        n, kq = l & 15, l >> 4
        b = (bin0 + 4 * kq) ^ xf[n]
        addr.append([n * S + b + i for i in range(4)])
    return cost(B128_GROUPS, addr, 64)

def search_p():
    k2g = lambda g: 4 * (g >> 1) + (g & 1)
    best = []
    for S in range(520, 620, 4):
        for name, xf in [("none", [0] * 16), ("4*(f>>2)", [4 * (f >> 2) for f in range(16)]),
                         ("4*(f&3)", [4 * (f & 3) for f in range(16)]), ("4*(f>>1)&28", [(4 * (f >> 1)) & 28 for f in range(16)]),
                         ("4*f&28", [(4 * f) & 28 for f in range(16)]), ("8*(f>>2)", [8 * (f >> 2) for f in range(16)]),
                         ("16*(f>>3)+4*((f>>2)&1)", [16 * (f >> 3) + 4 * ((f >> 2) & 1) for f in range(16)])]:
            w = max(p_write_cost(S, xf, c, k2g) for c in range(32))
            r = max(p_read_cost(S, xf, b0) for b0 in range(0, 64, 4))
            best.append((w + r, w, r, S, name))
    best.sort()
    for b in best[:12]: print("P: write cycles %d (ideal 2), read cycles %d (ideal 4)  S=%d swizzle %s" % b[1:])

def search_y():
    # row (slot k1, frame f) of RS bytes: parts Yr hi | Yi hi | Yr lo | Yi lo (64 B each).  Writer lane (f, g'): 8 bytes at
    # part offset + 8 * w; reader lane (f, g): 16 bytes at part offset + 16 * g.  Optional swizzle: 16-byte slot index ^ sig(f)
    res = []
    for RS in range(256, 400, 16):
        for name, sig in [("none", [0] * 16), ("f>>2", [f >> 2 for f in range(16)]), ("f&3", [f & 3 for f in range(16)]),
                          ("(f>>1)&3", [(f >> 1) & 3 for f in range(16)]), ("(f>>3)|((f>>1)&2)", [(f >> 3) | ((f >> 1) & 2) for f in range(16)])]:
            wmax = 0
            for w in range(8):
                addr = []
                for l in range(64):
                    f, gp = l & 15, l >> 4
                    slot16 = (w >> 1) ^ sig[f]
                    base = (gp * 4) * 16 * RS + f * RS + slot16 * 16 + (w & 1) * 8     # slot k1 = 4 g' + i: differs by multiples of 16 RS
                    addr.append([base // 4, base // 4 + 1])
                wmax = max(wmax, cost(W64, addr, 32))
            addr = []
            for l in range(64):
                f, g = l & 15, l >> 4
                base = f * RS + (g ^ sig[f]) * 16
                addr.append([base // 4 + i for i in range(4)])
            r = cost(B128_GROUPS, addr, 64)
            res.append((wmax + r, wmax, r, RS, name))
    res.sort()
    for b in res[:12]: print("Y: write cycles %d (ideal 4), read cycles %d (ideal 4)  RS=%d swizzle %s" % b[1:])

def search_t():
    for R in range(304, 400, 16):
        addr = []
        for l in range(64):
            n2, q = l & 31, l >> 5
            base = n2 * R + q * 8
            addr.append([base // 4, base // 4 + 1])
        print("T: row %d bytes: write cycles %d (ideal 4)" % (R, cost(W64, addr, 32)))

search_p(); search_y(); search_t()
