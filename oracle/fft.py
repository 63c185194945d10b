"""Radix-2 NTT over Fr (oracle; test infrastructure).

Line-by-line restatement of backend_new/src/fft.rs: fft :23-32, ifft :42-59,
fft_in_place :62-89, bit_reverse_permutation :92-102, coset_fft :121-137,
coset_ifft :140-157, Domain :175-249.  All vectors are natural order in and out
(fft.rs:300-318 pins fft(p)[i] == p(omega^i)).
"""
from .field import R_MOD, fr_inv, root_of_unity


def strip(coeffs):
    """polynomial.rs:91-99 — normalize(): drop trailing zero coefficients."""
    c = list(coeffs)
    while c and c[-1] == 0:
        c.pop()
    return c


def bit_reverse(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def fft_in_place(values, omega):
    """fft.rs:62-89."""
    n = len(values)
    if n == 1:
        return
    log_n = n.bit_length() - 1
    for i in range(n):  # fft.rs:92-102
        j = bit_reverse(i, log_n)
        if i < j:
            values[i], values[j] = values[j], values[i]
    m = 1
    while m < n:
        omega_m = pow(omega, n // (2 * m), R_MOD)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = w * values[k + j + m] % R_MOD
                u = values[k + j]
                values[k + j] = (u + t) % R_MOD
                values[k + j + m] = (u - t) % R_MOD
                w = w * omega_m % R_MOD
        m *= 2


def fft(coeffs, omega, n):
    """fft.rs:23-32 — pad to n, transform."""
    assert n & (n - 1) == 0, "FFT size must be a power of 2"
    v = list(coeffs) + [0] * (n - len(coeffs))
    assert len(v) == n
    fft_in_place(v, omega)
    return v


def ifft(evals, omega):
    """fft.rs:42-59 — returns the normalised coefficient list."""
    n = len(evals)
    assert n & (n - 1) == 0, "IFFT size must be a power of 2"
    v = list(evals)
    fft_in_place(v, fr_inv(omega))
    n_inv = fr_inv(n % R_MOD)
    return strip([c * n_inv % R_MOD for c in v])


def coset_fft(coeffs, omega, k, n):
    """fft.rs:121-137."""
    v = list(coeffs) + [0] * (n - len(coeffs))
    kp = 1
    for i in range(n):
        v[i] = v[i] * kp % R_MOD
        kp = kp * k % R_MOD
    return fft(strip(v), omega, n)


def coset_ifft(evals, omega, k):
    """fft.rs:140-157 — result padded back to n coefficients."""
    n = len(evals)
    c = ifft(evals, omega)
    k_inv = fr_inv(k)
    kp = 1
    for i in range(len(c)):
        c[i] = c[i] * kp % R_MOD
        kp = kp * k_inv % R_MOD
    return c + [0] * (n - len(c))


class Domain:
    """fft.rs:175-249."""

    def __init__(self, n):
        assert n & (n - 1) == 0
        self.n = n
        self.omega = root_of_unity(n)
        self.omega_inv = fr_inv(self.omega)
        self.n_inv = fr_inv(n % R_MOD)

    def elements(self):
        out, w = [], 1
        for _ in range(self.n):
            out.append(w)
            w = w * self.omega % R_MOD
        return out

    def fft(self, coeffs):
        return fft(coeffs, self.omega, self.n)

    def ifft(self, evals):
        return ifft(evals, self.omega)

    def coset_fft(self, coeffs, k):
        return coset_fft(coeffs, self.omega, k, self.n)

    def coset_ifft(self, evals, k):
        return coset_ifft(evals, self.omega, k)

    def vanishing_eval(self, x):
        return (pow(x, self.n, R_MOD) - 1) % R_MOD
