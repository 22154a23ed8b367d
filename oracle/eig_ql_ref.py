"""oracle/eig_ql_ref.py -- TEST INFRASTRUCTURE ONLY.

Second, independent restatement of the reference's symmetric eigen-solver (glm_modification::findEigenvaluesSymReal,
CR/auxiliary.h:217-401 = glm gtx/pca.inl: Householder tridiagonalisation followed by implicit-shift QL, with the
reference's ABSOLUTE 1e-7 convergence thresholds), written from the algorithm's textbook form (tred2 / tqli) in 0-based
Python over numpy scalars of a chosen dtype.  It pins `eig_sym3` of oracle/raster_ref.c (and with it the inverse
covariance every plane / normal term of the rasterizer is built from): tests/test_oracle_pins_cpu.py compares the two
value for value.  Scalar Python -- small inputs only.
"""
import numpy as np

EPS = 0.0000001          # CR/auxiliary.h:203,238 -- absolute, not relative
MAX_ITER = 30            # :329


def _near_zero(x, T):
    return abs(T(x) - T(0)) <= T(EPS)          # glm_modification::equal(x, 0, eps), :189-192


def _sign_of(v, s):
    return abs(v) if s >= 0 else -abs(v)        # transferSign, :195-198


def _hypot(a, b, T):
    """pythag, :201-214 (the reference's variant returns 0 for a small second argument)."""
    aa, ab = abs(a), abs(b)
    if aa > ab:
        q = T(ab / aa)
        q = T(q * q)
        return T(aa * np.sqrt(T(T(1) + q)))
    if _near_zero(ab, T):
        return T(0)
    q = T(aa / ab)
    q = T(q * q)
    return T(ab * np.sqrt(T(T(1) + q)))


def eig_sym(cov, dtype=np.float32):
    """cov: (n, n) symmetric.  Returns (count, eigenvalues[n], eigenvectors[n][n] with eigenvectors[k] = k-th vector);
    count = 0 when the QL iteration gives up (the reference then zeroes planes and normal, CR/forward.cu:163-168)."""
    T = dtype
    n = cov.shape[0]
    a = np.array(cov, dtype=T).copy()           # a[r][c]
    d = np.zeros(n, T)
    e = np.zeros(n, T)
    # ---- Householder reduction to tridiagonal form (tred2), rows from the last upwards
    for i in range(n - 1, 0, -1):
        l = i - 1
        h = T(0)
        if l > 0:
            scale = T(0)
            for k in range(l + 1):
                scale = T(scale + abs(a[i, k]))
            if _near_zero(scale, T):
                e[i] = a[i, l]
            else:
                for k in range(l + 1):
                    a[i, k] = T(a[i, k] / scale)
                    h = T(h + T(a[i, k] * a[i, k]))
                f = a[i, l]
                g = T(-np.sqrt(h)) if f >= 0 else T(np.sqrt(h))
                e[i] = T(scale * g)
                h = T(h - T(f * g))
                a[i, l] = T(f - g)
                f = T(0)
                for j in range(l + 1):
                    a[j, i] = T(a[i, j] / h)
                    g = T(0)
                    for k in range(j + 1):
                        g = T(g + T(a[j, k] * a[i, k]))
                    for k in range(j + 1, l + 1):
                        g = T(g + T(a[k, j] * a[i, k]))
                    e[j] = T(g / h)
                    f = T(f + T(e[j] * a[i, j]))
                hh = T(f / T(h + h))
                for j in range(l + 1):
                    f = a[i, j]
                    g = T(e[j] - T(hh * f))
                    e[j] = g
                    for k in range(j + 1):
                        a[j, k] = T(a[j, k] - T(T(f * e[k]) + T(g * a[i, k])))
        else:
            e[i] = a[i, l]
        d[i] = h
    d[0] = T(0)
    e[0] = T(0)
    # accumulate the transformation
    for i in range(n):
        if not _near_zero(d[i], T):
            for j in range(i):
                g = T(0)
                for k in range(i):
                    g = T(g + T(a[i, k] * a[k, j]))
                for k in range(i):
                    a[k, j] = T(a[k, j] - T(g * a[k, i]))
        d[i] = a[i, i]
        a[i, i] = T(1)
        for j in range(i):
            a[j, i] = T(0)
            a[i, j] = T(0)
    # ---- implicit-shift QL on the tridiagonal (tqli)
    for i in range(1, n):
        e[i - 1] = e[i]
    e[n - 1] = T(0)
    for l in range(n):
        it = 0
        while True:
            m = l
            while m < n - 1:
                if _near_zero(abs(e[m]), T):
                    break
                m += 1
            if m == l:
                break
            if it == MAX_ITER:
                return 0, d, a.T.copy()
            it += 1
            g = T(T(d[l + 1] - d[l]) / T(T(2) * e[l]))
            r = _hypot(g, T(1), T)
            g = T(T(d[m] - d[l]) + T(e[l] / T(g + _sign_of(r, g))))
            s = c = T(1)
            p = T(0)
            i = m - 1
            underflow = False
            while i >= l:
                f = T(s * e[i])
                b = T(c * e[i])
                r = _hypot(f, g, T)
                e[i + 1] = r
                if _near_zero(r, T):
                    d[i + 1] = T(d[i + 1] - p)
                    e[m] = T(0)
                    underflow = True
                    break
                s = T(f / r)
                c = T(g / r)
                g = T(d[i + 1] - p)
                r = T(T(T(d[i] - g) * s) + T(T(T(2) * c) * b))
                p = T(s * r)
                d[i + 1] = T(g + p)
                g = T(T(c * r) - b)
                for k in range(n):
                    f = a[k, i + 1]
                    a[k, i + 1] = T(T(s * a[k, i]) + T(c * f))
                    a[k, i] = T(T(c * a[k, i]) - T(s * f))
                i -= 1
            if underflow and i >= l:
                continue
            d[l] = T(d[l] - p)
            e[l] = g
            e[m] = T(0)
    return n, d, a.T.copy()                     # eigenvector k = column k of a


def inverse_from_eig(val, vec, dtype=np.float64):
    """CR/forward.cu:135-155: E diag(1/lambda) E^T when the smallest eigenvalue exceeds 1e-8, else e_min e_min^T."""
    val = np.asarray(val, dtype); vec = np.asarray(vec, dtype)
    k = int(np.argmin(val))
    if val[k] > 1e-8:
        return sum(np.outer(vec[j], vec[j]) / val[j] for j in range(len(val))), True
    return np.outer(vec[k], vec[k]), False
