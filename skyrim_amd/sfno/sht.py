"""Constant matrices of the real spherical-harmonic transform (torch-harmonics ``RealSHT`` / ``InverseRealSHT``
conventions: norm="ortho", Condon-Shortley phase, x -> 2*pi * rfft(x, norm="forward"), synthesis -> irfft(norm="forward")),
in the shapes the batched GEMM of include/skyrim_sfno.h consumes (W[n][k], k contracted):

    dft        [2 mmax][n_lon]       rows (2m, 2m+1) = 2*pi/n_lon * (cos, -sin)(2*pi*m*j/n_lon)         analysis along longitude
    idft       [n_lon][2 mmax]       cols (2m, 2m+1) = c_m * (cos, -sin)(2*pi*m*j/n_lon), c_0 = 1, c_m = 2   (Hermitian completion)
    analysis   [mmax][lmax][n_lat]   Pbar_l^m(cos theta_k) * w_k                                        one matrix per order m
    synthesis  [mmax][n_lat][lmax]   Pbar_l^m(cos theta_k)

Everything is computed in float64 and rounded once.  Two latitude grids: "equiangular" (both poles, Clenshaw-Curtis
weights: the 721-row ERA5 grid) and "legendre-gauss" (the network's internal grid).
"""
from __future__ import annotations

import numpy as np


def colatitudes_and_weights(n_lat: int, grid: str):
    if grid == "legendre-gauss":
        x, w = np.polynomial.legendre.leggauss(n_lat)
        return np.arccos(x[::-1]), w[::-1].copy()
    if grid != "equiangular":
        raise ValueError(f"unknown latitude grid {grid!r}")
    # Clenshaw-Curtis on theta_j = pi j / (n - 1), j = 0..n-1 (cosine series of the weight function, vectorised)
    nn = n_lat - 1
    theta = np.pi * np.arange(n_lat) / nn
    k = np.arange(1, nn // 2 + 1)
    b = np.where(2 * k == nn, 1.0, 2.0)
    series = (b / (4.0 * k * k - 1.0))[None, :] * np.cos(2.0 * np.outer(theta, k))
    c = np.full(n_lat, 2.0)
    c[0] = c[-1] = 1.0
    return theta, c / nn * (1.0 - series.sum(axis=1))


def legendre_functions(mmax: int, lmax: int, theta: np.ndarray) -> np.ndarray:
    """[m][l][k] = orthonormal Pbar_l^m(cos theta_k) (zero for l < m), three-term recurrence in degree, float64."""
    x, s = np.cos(theta), np.sin(theta)
    out = np.zeros((mmax, lmax, theta.size))
    sectoral = np.full(theta.size, np.sqrt(0.25 / np.pi))
    for m in range(mmax):
        if m:
            sectoral = -np.sqrt(1.0 + 0.5 / m) * s * sectoral
        if m >= lmax:
            break
        out[m, m] = sectoral
        if m + 1 < lmax:
            out[m, m + 1] = np.sqrt(2.0 * m + 3.0) * x * sectoral
        ls = np.arange(m + 2, lmax, dtype=np.float64)
        alpha = np.sqrt((4.0 * ls * ls - 1.0) / (ls * ls - m * m))
        beta = np.sqrt(((ls - 1.0) ** 2 - m * m) / (4.0 * (ls - 1.0) ** 2 - 1.0))
        for i, l in enumerate(range(m + 2, lmax)):
            out[m, l] = alpha[i] * (x * out[m, l - 1] - beta[i] * out[m, l - 2])
    return out


class ShtMatrices:
    def __init__(self, n_lat: int, n_lon: int, lmax: int, mmax: int, grid: str):
        if mmax > n_lon // 2:
            raise ValueError("mmax must stay below the Nyquist order n_lon / 2")
        self.n_lat, self.n_lon, self.lmax, self.mmax, self.grid = n_lat, n_lon, lmax, mmax, grid
        theta, wq = colatitudes_and_weights(n_lat, grid)
        p = legendre_functions(mmax, lmax, theta)
        self.analysis = np.ascontiguousarray(p * wq[None, None, :], dtype=np.float32)             # [m][l][lat]
        self.synthesis = np.ascontiguousarray(p.transpose(0, 2, 1), dtype=np.float32)              # [m][lat][l]
        ang = 2.0 * np.pi * np.outer(np.arange(mmax), np.arange(n_lon)) / n_lon                    # [m][j]
        dft = np.empty((2 * mmax, n_lon))
        dft[0::2] = np.cos(ang) * (2.0 * np.pi / n_lon)
        dft[1::2] = -np.sin(ang) * (2.0 * np.pi / n_lon)
        self.dft = np.ascontiguousarray(dft, dtype=np.float32)                                     # [2m][lon]
        cm = np.where(np.arange(mmax) == 0, 1.0, 2.0)[:, None]
        idft = np.empty((n_lon, 2 * mmax))
        idft[:, 0::2] = (np.cos(ang) * cm).T
        idft[:, 1::2] = (-np.sin(ang) * cm).T
        self.idft = np.ascontiguousarray(idft, dtype=np.float32)                                   # [lon][2m]
