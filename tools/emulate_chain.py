"""Lane-level emulation (numpy, float64) of sfno_chain.hip's data flow: fragment-order weight layouts, the MFMA 16x16x32 lane
mapping, the hidden-chunk -> k-slot correspondence and the perm8 chaining of two pairs.  Checks the INDEX ALGEBRA of the kernel
against plain matrix products on the CPU (no GPU needed); the arithmetic itself (fp16 hi/lo) is covered by the GPU parity tests.

    python tools/emulate_chain.py
"""
import numpy as np
from scipy.special import erf

LANES = np.arange(64)
L15, G = LANES & 15, LANES >> 4


def perm8_col(rho):
    r = rho & 31
    return (rho & ~31) + 8 * ((r >> 2) & 3) + 4 * (r >> 4) + (r & 3)


def prep_w1(w1):                      # [H][K] -> blocks [(j KS + ks) 2 + n][lane][e]   (one plane)
    H, K = w1.shape
    KS = K // 32
    out = np.zeros((H // 32 * KS * 2, 64, 8))
    for j in range(H // 32):
        for ks in range(KS):
            for n in range(2):
                for e in range(8):
                    out[(j * KS + ks) * 2 + n, :, e] = w1[32 * j + 16 * n + L15, 32 * ks + 8 * G + e]
    return out


def prep_w2(w2):                      # [N][H] -> blocks [j CF + c][lane][e]
    N, H = w2.shape
    CF = N // 16
    out = np.zeros((H // 32 * CF, 64, 8))
    for j in range(H // 32):
        for c in range(CF):
            for e in range(8):
                out[j * CF + c, :, e] = w2[perm8_col(16 * c + L15), 32 * j + 16 * (e >> 2) + 4 * G + (e & 3)]
    return out


def mfma(a, b, acc):
    """a, b: [64][8] per-lane operands, acc: [64][4].  D(16x16) += A(16x32) B(32x16)."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for e in range(8):
        A[L15, 8 * G + e] = a[:, e]
        B[8 * G + e, L15] = b[:, e]
    D = A @ B
    out = acc.copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, L15]
    return out


def gelu(x):
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def load_frags(src, creal, KS, scale, shift):      # src [C][16 pixels] -> [KS][64][8]
    out = np.zeros((KS, 64, 8))
    for ks in range(KS):
        for e in range(8):
            ch = 32 * ks + 8 * G + e
            ok = ch < creal
            v = src[np.where(ok, ch, 0), L15]
            out[ks, :, e] = np.where(ok, v * scale[ch] + shift[ch], 0.0)
    return out


def run_pair(x, w1f, w2f, b1, KS, NCH, CF):
    yacc = np.zeros((CF, 64, 4))
    for j in range(NCH):
        hacc = np.zeros((2, 64, 4))
        for s in range(KS * 2):
            ks, n = s >> 1, s & 1
            hacc[n] = mfma(w1f[(j * KS + ks) * 2 + n], x[ks], hacc[n])
        h = np.zeros((64, 8))
        for n in range(2):
            for r in range(4):
                h[:, 4 * n + r] = gelu(hacc[n][:, r] + b1[j * 32 + 16 * n + 4 * G + r])
        for c in range(CF):
            yacc[c] = mfma(w2f[j * CF + c], h, yacc[c])
    return yacc


def acc_channels(yacc, nbp):          # -> [nbp][64][8]: channel 32 bp + 8 g + i of pixel l15
    out = np.zeros((nbp, 64, 8))
    for bp in range(nbp):
        out[bp, :, :4] = yacc[2 * bp]
        out[bp, :, 4:] = yacc[2 * bp + 1]
    return out


def main():
    rng = np.random.default_rng(0)
    CP, HP, KXP, OP = 64, 96, 32, 32
    C, HID, KX, OUT = 40, 80, 11, 9
    pad = lambda m, r, c: np.pad(m, ((0, r - m.shape[0]), (0, c - m.shape[1])))  # noqa: E731
    padv = lambda v, n: np.pad(v, (0, n - v.shape[0]))  # noqa: E731
    y, res, x = rng.normal(size=(C, 16)), rng.normal(size=(C, 16)), rng.normal(size=(KX, 16)) * 50 + 100
    W1, b1, W2, b2 = rng.normal(size=(HID, C)), rng.normal(size=HID), rng.normal(size=(C, HID)), rng.normal(size=C)
    sc, sh = rng.normal(size=C), rng.normal(size=C)
    xs, xh = rng.normal(size=KX) * 0.02, rng.normal(size=KX)
    V1, d1, V2, d2 = rng.normal(size=(C, C + KX)), rng.normal(size=C), rng.normal(size=(OUT, C)), rng.normal(size=OUT)

    # reference
    z = W2 @ gelu(W1 @ (y * sc[:, None] + sh[:, None]) + b1[:, None]) + b2[:, None] + res
    ref = V2 @ gelu(V1 @ np.concatenate([z, x * xs[:, None] + xh[:, None]]) + d1[:, None]) + d2[:, None]

    # emulated kernel
    w1f, w2f = prep_w1(pad(W1, HP, CP)), prep_w2(pad(W2, CP, HP))
    V1p = np.zeros((CP, CP + KXP)); V1p[:C, :C] = V1[:, :C]; V1p[:C, CP:CP + KX] = V1[:, C:]
    v1f, v2f = prep_w1(V1p), prep_w2(pad(V2, OP, CP))
    xf = load_frags(y, C, CP // 32, padv(sc, CP), padv(sh, CP))
    yacc = run_pair(xf, w1f, w2f, padv(b1, HP), CP // 32, HP // 32, CP // 16)
    zc = acc_channels(yacc, CP // 32)
    zf = np.zeros(((CP + KXP) // 32, 64, 8))
    for bp in range(CP // 32):
        for i in range(8):
            ch = 32 * bp + 8 * G + i
            ok = ch < C
            r = res[np.where(ok, ch, 0), L15]
            zf[bp, :, i] = np.where(ok, zc[bp, :, i] + padv(b2, CP)[ch] + r, 0.0)
    zf[CP // 32:] = load_frags(x, KX, KXP // 32, padv(xs, KXP), padv(xh, KXP))
    # mid check: the block output
    for bp in range(CP // 32):
        for i in range(8):
            ch = 32 * bp + 8 * G + i
            ok = ch < C
            assert np.allclose(zf[bp, ok, i], z[ch[ok], L15[ok]], atol=1e-9), ("mid", bp, i)
    zacc = run_pair(zf, v1f, v2f, padv(d1, CP), (CP + KXP) // 32, CP // 32, OP // 16)
    oc = acc_channels(zacc, OP // 32)
    got = np.zeros_like(ref)
    for bp in range(OP // 32):
        for i in range(8):
            ch = 32 * bp + 8 * G + i
            ok = ch < OUT
            got[ch[ok], L15[ok]] = oc[bp, ok, i] + padv(d2, OP)[ch[ok]]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("tail chain, emulated lanes vs matrices: rel err", err)
    assert err < 1e-12
    print("ok")


if __name__ == "__main__":
    main()
