"""Coefficients of the erf-GELU used by the fc1 epilogue (csrc/common.h gelu_erf):

    GELU(x) = 0.5 x erfc(-x/sqrt2),   erfc(z) = 2^Q(z*sqrt2) for z >= 0,  erfc(-z) = 2 - erfc(z)
    => E = exp2(Q(min(|x|, U)));  GELU = 0.5 x (x > 0 ? 2 - E : E)

Q = degree-8 polynomial fitted (erfc-weighted least squares on Chebyshev nodes) to log2 erfc(u/sqrt2) on [0, U].
One transcendental (v_exp_f32) instead of two (rcp + exp) and 8 FMAs: ~11 VALU issue slots per element instead of ~20.
Prints the coefficients and the max abs error of the fp32 evaluation against the exact GELU."""
import numpy as np
from scipy.special import erf, erfc

U, DEG = 5.9396969619669995, 8      # U = 4.2 * sqrt(2): erfc(4.2) = 2.9e-9


def main():
    n = 6000
    u = np.cos(np.pi * (np.arange(n) + 0.5) / n) * U / 2 + U / 2
    f = np.log2(erfc(u / np.sqrt(2)))
    w = erfc(u / np.sqrt(2)) + 1e-3
    c = np.polynomial.polynomial.polyfit(u, f, DEG, w=w)
    c[0] = 0.0                                   # erfc(0) = 1 exactly -> GELU(x) ~ x/2 near 0 without bias
    c32 = c.astype(np.float32)
    x = np.linspace(-9, 9, 900001).astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(U))
    p = np.full_like(a, c32[-1])
    for k in range(DEG - 1, -1, -1):
        p = p * a + c32[k]
    e = np.exp2(p).astype(np.float32)
    g = (np.float32(0.5) * x * np.where(x > 0, np.float32(2) - e, e)).astype(np.float32)
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    err = np.abs(g - ref)
    print("max abs err %.3e at x = %.3f" % (err.max(), x[err.argmax()]))
    print("coefficients c1..c8 (c0 = 0):")
    print(", ".join("%.9ef" % v for v in c32[1:]))


if __name__ == "__main__":
    main()
