"""Bank-conflict model of the F kernels' LDS access patterns (developer tool).
Rules from /opt/skills/guides/MI355X_MICROARCH.md §LDS."""
import sys
import numpy as np

G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 = G128 + [[x + 32 for x in g] for g in G128]

def cycles(addr_bytes, width, kind):
    """addr_bytes: 64 byte addresses; width in bytes; kind 'r' or 'w' -> (cycles, ideal)"""
    a = np.asarray(addr_bytes)
    nd = width // 4
    if kind == 'r':
        if width == 4: groups, nb = [list(range(32)), list(range(32, 64))], 32
        elif width == 8: groups, nb = [list(range(32)), list(range(32, 64))], 64
        elif width == 16: groups, nb = G128, 64
    else:
        nb = 32
        if width == 4: groups = [list(range(32)), list(range(32, 64))]
        elif width == 8: groups = [list(range(i, i + 16)) for i in range(0, 64, 16)]
        elif width == 16: groups = [list(range(i, i + 8)) for i in range(0, 64, 8)]
    tot = 0
    for g in groups:
        per_bank = {}
        for l in g:
            for d in range(nd):
                dw = a[l] // 4 + d
                per_bank.setdefault(dw % nb, set()).add(dw)
        tot += max(len(s) for s in per_bank.values())
    return tot, len(groups)

def e1(k1, m, M1): return k1 * M1 + (m ^ ((k1 & 3) << 3))
def e2(r, j3): return r * 8 + ((((j3 >> 1) ^ ((r >> 2) & 3)) << 1) | (j3 & 1))

def report(name, pats):
    c = sum(p[0] for p in pats); i = sum(p[1] for p in pats)
    print(f"{name:34s} instrs {len(pats):4d}  cycles {c:5d}  ideal {i:5d}  x{c/i:.2f}")
    return c, i

def main(R1, R2, R3):
    N = R1 * R2 * R3; M1 = R2 * R3; C1 = M1 // 64; C2 = R1 * R3 // 64; C3 = R1 * R2 // 64
    L = np.arange(64)
    tot = [0, 0]
    def add(name, pats):
        c, i = report(name, pats); tot[0] += c; tot[1] += i
    # E1 write: b128 if C1 == 2 (pair c=0,1 contiguous) else b64
    if C1 == 2: add("E1 write b128", [cycles([8 * e1(k1, 2 * l, M1) for l in L], 16, 'w') for k1 in range(R1)])
    add("E1 write b64", [cycles([8 * e1(k1, C1 * l + c, M1) for l in L], 8, 'w') for k1 in range(R1) for c in range(C1)])
    add("E1 read b64", [cycles([8 * e1((l >> 3) + 8 * c2, 8 * j2 + (l & 7), M1) for l in L], 8, 'r') for c2 in range(C2) for j2 in range(R2)])
    add("E2 write b64", [cycles([8 * e2(k2 * R1 + (l >> 3) + 8 * c2, l & 7) for l in L], 8, 'w') for c2 in range(C2) for k2 in range(R2)])
    add("E2 read b128", [cycles([8 * e2(l + 64 * c3, 2 * u) for l in L], 16, 'r') for c3 in range(C3) for u in range(4)])
    add("E2 read b64", [cycles([8 * e2(l + 64 * c3, j) for l in L], 8, 'r') for c3 in range(C3) for j in range(8)])
    add("natural write b64", [cycles([8 * (l + 64 * c3 + R1 * R2 * k3) for l in L], 8, 'w') for c3 in range(C3) for k3 in range(R3)])
    NG = N // 256
    add("epi fwd read b64 (4l+c)", [cycles([8 * (256 * g + 4 * l + c) for l in L], 8, 'r') for g in range(NG) for c in range(4)])
    add("epi fwd read b128 (4l+2u)", [cycles([8 * (256 * g + 4 * l + 2 * u) for l in L], 16, 'r') for g in range(NG) for u in range(2)])
    add("epi rev read b64", [cycles([8 * (N - 256 * g - 4 * l - c) for l in L], 8, 'r') for g in range(NG) for c in range(4)])
    add("inv-in fwd b128 (2l)", [cycles([8 * (M1 * j1 + C1 * l) for l in L], 8 * C1, 'r') for j1 in range(R1)])
    add("inv-in rev b64", [cycles([8 * (N - (M1 * j1 + C1 * l + c)) for l in L], 8, 'r') for j1 in range(R1) for c in range(C1)])
    add("time-epi read b128/b64", [cycles([8 * (C1 * (64 * j + l)) for l in L], 8 * C1, 'r') for j in range(R1)])
    add("mask rev read b32", [cycles([4 * (N - (M1 * j1 + C1 * l + c)) for l in L], 4, 'r') for j1 in range(R1) for c in range(C1)])
    print("(not every row is emitted: rows list alternatives for the same data)")

if __name__ == "__main__":
    r = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else [16, 16, 8]
    main(*r)
