"""LDS bank conflicts of the compact-bias reads of csrc/attention.hip (earth_attention2_kernel), simulated.

A lane reads 4 consecutive fp16 of table copy c = (w_q + 1) mod 4, row (z_q + 2 z_k) 36 + h_q + 6 h_k, column 11 - w_q + c + w_k0 with one
ds_read_b64; on gfx950 that instruction is served in two groups of 32 lanes, bank = (address / 4) mod 64, N distinct addresses on one bank =
N cycles (MI355X_MICROARCH.md, LDS).  This walks every (query fragment, key group, half-wave) of the kernel for a table layout
address = offset[c] + row * ROW + column (8-byte slots) and searches the per-copy offsets mod 256 B for every row pitch.
Result: no layout of this family is conflict-free -- the best averages 1.88 cycles per group (worst 2); the kernel's layout (ROW = 7 slots,
120 fp16 between the copies) sits at 1.99 against 3.13 for copies back to back.      python tools/attn_bank_sim.py"""
import numpy as np, itertools
def groups(ROW):
    out = []
    for qf in range(9):
        for f in range(9):
            for half in range(2):
                s = set()
                for g in (2*half, 2*half+1):
                    k0 = 32*(f>>1) + 8*g + 4*(f&1) if f < 8 else 128 + 4*g
                    zk, hk, wk0 = k0 // 72, (k0 // 12) % 6, k0 % 12
                    for l15 in range(16):
                        qi = 16*qf + l15
                        zq = 1 if qi >= 72 else 0
                        hq = (qi - 72*zq) // 12
                        wq = qi - 72*zq - 12*hq
                        c = (wq + 1) & 3
                        r = (zq + 2*zk) * 36 + hq + 6*hk
                        col = 11 - wq + c + wk0
                        s.add((c, r * ROW + col // 4))
                out.append(sorted(s))
    return out
for ROW in (7, 8, 9, 10, 12, 16):
    G = groups(ROW)
    # all d triples
    D = np.array(list(itertools.product(range(32), repeat=3)), dtype=np.int32)   # d1,d2,d3
    D = np.concatenate([np.zeros((len(D),1),np.int32), D], axis=1)              # d0 = 0
    tot = np.zeros(len(D), np.int32); worst = np.zeros(len(D), np.int32)
    for grp in G:
        cs = np.array([c for c, a in grp]); as_ = np.array([a for c, a in grp])
        banks = (as_[None, :] + D[:, cs]) % 32                     # (nD, n)
        cnt = np.zeros((len(D), 32), np.int32)
        for j in range(banks.shape[1]):
            np.add.at(cnt, (np.arange(len(D)), banks[:, j]), 1)
        m = cnt.max(axis=1)
        tot += m; worst = np.maximum(worst, m)
    i = np.argmin(tot)
    print("ROW", ROW, "best d", D[i], "avg", tot[i] / len(G), "worst", worst[i], " #groups", len(G), " n at 1.0:", (tot == len(G)).sum())
