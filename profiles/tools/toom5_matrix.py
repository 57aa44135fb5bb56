"""The interpolation matrix of the Toom-5 x Karatsuba syrk (kernels.hpp: k_syrk5_finish; round 6), derived exactly.

The row polynomial has five pieces, the product polynomial nine coefficients c_0 .. c_8; it is evaluated at the nine points
0, 1, -1, 2, -2, 1/2, -1/2, 3, inf (the two halves scaled by 2^8 so that they are integers).  Prints, for every coefficient, the
common denominator D and the integer row M with c_k = (M . V) / D, and checks the construction on random integers -- image
(pieces of 102 bits, biases, Karatsuba halves of 55 bits, 28-bit limbs), 27 lazily carried product sums, recombination,
bias removal, interpolation -- against the plain integer product.      python profiles/tools/toom5_matrix.py
"""
import random
from fractions import Fraction
from math import lcm

W = 102; BETA = 1 << W; FB = 5 * W - 1; H = 55; B = 28; MASK = (1 << B) - 1
PTS = ["0", "1", "-1", "2", "-2", "h", "-h", "3", "inf"]
BIAS = [0, 0, 2 * BETA, 0, 10 * BETA, 0, 10 * BETA, 0, 0]


def evals(a):
    a0, a1, a2, a3, a4 = a
    return [a0, a0 + a1 + a2 + a3 + a4, a0 - a1 + a2 - a3 + a4 + 2 * BETA, a0 + 2 * a1 + 4 * a2 + 8 * a3 + 16 * a4,
            a0 - 2 * a1 + 4 * a2 - 8 * a3 + 16 * a4 + 10 * BETA, 16 * a0 + 8 * a1 + 4 * a2 + 2 * a3 + a4,
            16 * a0 - 8 * a1 + 4 * a2 - 2 * a3 + a4 + 10 * BETA, a0 + 3 * a1 + 9 * a2 + 27 * a3 + 81 * a4, a4]


def row(pt):
    if pt == "0":
        return [Fraction(1)] + [Fraction(0)] * 8
    if pt == "inf":
        return [Fraction(0)] * 8 + [Fraction(1)]
    if pt == "h":
        return [Fraction(2) ** (8 - k) for k in range(9)]
    if pt == "-h":
        return [Fraction(2) ** (8 - k) * (-1) ** k for k in range(9)]
    return [Fraction(int(pt)) ** k for k in range(9)]


def inverse(M):
    n = len(M)
    A = [r[:] + [Fraction(int(i == j)) for j in range(n)] for i, r in enumerate(M)]
    for c in range(n):
        p = next(r for r in range(c, n) if A[r][c] != 0)
        A[c], A[p] = A[p], A[c]
        A[c] = [v / A[c][c] for v in A[c]]
        for r in range(n):
            if r != c and A[r][c] != 0:
                A[r] = [v - A[r][c] * w for v, w in zip(A[r], A[c])]
    return [r[n:] for r in A]


MINV = inverse([row(p) for p in PTS])


def image(v):
    ap = v + (1 << FB)
    assert 0 < ap < (1 << (FB + 1))
    out = []
    for e in evals([(ap >> (W * k)) & (BETA - 1) for k in range(5)]):
        assert 0 <= e < (1 << (2 * H))
        lo, hi = e & ((1 << H) - 1), e >> H
        assert lo + hi < (1 << (2 * B))
        out += [lo, hi, lo + hi]
    return out


def entry(col_i, col_j):
    n = len(col_i)
    ii, jj = [image(v) for v in col_i], [image(v) for v in col_j]
    S = []
    for g in range(27):
        c, tot = [0, 0, 0], 0
        for r in range(n):
            x, y = ii[r][g], jj[r][g]
            c[0] += (x & MASK) * (y & MASK)
            c[1] += (x & MASK) * (y >> B) + (x >> B) * (y & MASK)
            c[2] += (x >> B) * (y >> B)
            assert max(c) < (1 << 64)
            if r % 64 == 63 or r == n - 1:
                tot += c[0] + (c[1] << B) + (c[2] << (2 * B))
                c = [0, 0, 0]
        S.append(tot)
    V = [S[3 * i] + ((S[3 * i + 2] - S[3 * i] - S[3 * i + 1]) << H) + (S[3 * i + 1] << (2 * H)) for i in range(9)]
    for i in (2, 4, 6):
        K = BIAS[i]
        si = sum(im[3 * i] + (im[3 * i + 1] << H) for im in ii) - n * K
        sj = sum(im[3 * i] + (im[3 * i + 1] << H) for im in jj) - n * K
        V[i] -= K * si + K * sj + n * K * K
    c = []
    for k in range(9):
        num = sum(MINV[k][i] * V[i] for i in range(9))
        assert num.denominator == 1
        c.append(int(num))
    return sum(ck << (W * k) for k, ck in enumerate(c))


if __name__ == "__main__":
    for k, r in enumerate(MINV):
        D = lcm(*[f.denominator for f in r])
        print(k, "D =", D, [int(f * D) for f in r])
    random.seed(1)
    for n in (1, 63, 64, 65, 200):
        ci = [random.randrange(-(1 << FB) + 1, 1 << FB) for _ in range(n)]
        cj = [random.randrange(-(1 << FB) + 1, 1 << FB) for _ in range(n)]
        ci[0], cj[0] = (1 << FB) - 1, -(1 << FB) + 1
        assert entry(ci, cj) == sum((a + (1 << FB)) * (b + (1 << FB)) for a, b in zip(ci, cj))
    print("exact")
