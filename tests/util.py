"""Shared helpers for the parity tests: seeded inputs in BLAS column-major storage, host and device."""
import numpy as np


class ColMajor:
    """An (rows x cols) column-major matrix with leading dimension ld >= rows.

    .buf  : numpy C-order array (cols, ld)  -- row j of buf is column j of the matrix
    .view : numpy Fortran view (rows, cols) with strides (8, 8*ld) -- what the oracle takes
    .dev  : torch tensor on cuda with the same memory image (lazily created)
    """

    def __init__(self, rows, cols, ld=None, fill=None, rng=None, dtype=np.float64):
        ld = rows if ld is None else ld
        assert ld >= rows
        self.rows, self.cols, self.ld = rows, cols, ld
        self.buf = np.zeros((cols, max(ld, 1)), dtype=dtype)
        if fill is not None:
            self.buf[:, :rows] = fill(rng, (cols, rows))
            if ld > rows:  # padding must never be read: poison it
                self.buf[:, rows:] = np.nan if dtype == np.float64 else 0
        self._dev = None

    @property
    def view(self):
        return self.buf.T[: self.rows, :]

    @property
    def dev(self):
        if self._dev is None:
            import torch
            self._dev = torch.from_numpy(self.buf).cuda()
        return self._dev

    def download(self):
        self.buf[...] = self._dev.cpu().numpy()
        return self.view


def uniform_pm1(rng, shape):
    return rng.uniform(-1.0, 1.0, shape)


def uniform01(rng, shape):  # the reference's urand01 (test/main_test.cu:195-202): (0, 1]
    return 1.0 - rng.uniform(0.0, 1.0, shape)


def exp_rand(phi):  # test/main_test.cu:56-70: (u - 0.5) * exp(phi * randn)
    def f(rng, shape):
        return (rng.uniform(0, 1, shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    return f


def wide_exponent(decades):  # BASELINE config 3: u * 10^(decades*w)
    def f(rng, shape):
        return rng.uniform(-1, 1, shape) * 10.0 ** (decades * rng.uniform(0, 1, shape))
    return f


def operand(op, rows, cols, rng, fill=uniform_pm1, pad=0):
    """storage of an operand whose op(X) is (rows x cols): X itself for 'N', its transpose for 'T'"""
    r, c = (rows, cols) if op == "N" else (cols, rows)
    return ColMajor(r, c, ld=r + pad, fill=fill, rng=rng)
