#!/usr/bin/env python
"""Round 5 (VERDICT r4 next #4b): would a TILE-scaled fixed-point trailing update X2 -= L21 X1 of P = L^-1 B keep the bits of the
multi-word float update?  Measured with the oracle (GMP, test infrastructure) -- not part of the product.

A fixed-point image of a 32-wide tile keeps an entry of row i of L21 to 2^-F of that row's largest entry IN THE TILE, and an
entry of right-hand side c of X1 to 2^-F of that column's largest entry in the tile.  The float update carries every term
L(i,k) X(k,c) to 2^-(32 NL) of ITSELF, i.e. the sum to 2^-(32 NL) of its largest term.  The fixed-point sum is exact to
2^-F maxL_i maxX_c.  Bits lost against the float update, per output (i, c) and tile:

        loss(i, c) = log2(maxL_i maxX_c) - log2(max_k |L(i,k) X(k,c)|)   (0 <= loss <= min(span of row i, span of column c))

The spans themselves (round 3 measured them over WHOLE rows: 2^74 / 2^66) are printed too.
usage: python profiles/tools/trsm_tile_exponents.py <C4 scale> <iterations...>      e.g.  0.25 2 10 25 40
"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from sdpb_amd import synthetic
from tests import parity

TILE = 32


def log2mag(strings):
    """log2 |v| of the oracle's decimal strings (mantissa[e exponent]): -inf for zero"""
    out = np.empty(len(strings))
    L10 = np.log2(10.0)
    for i, t in enumerate(strings):
        t = t.lstrip("-")
        m, _, e = t.lower().partition("e")
        f = float(m[:18])
        out[i] = -np.inf if f == 0.0 else np.log2(f) + (int(e) if e else 0) * L10
    return out


def measure(o, sdp, N):
    span_L, span_X, loss = [], [], []
    for j in range(sdp.J):
        L = log2mag(o.array("L", j))
        X = log2mag(o.array("P", j))
        Pj = int(round(len(L) ** 0.5))
        L = L.reshape((Pj, Pj), order="F")   # column-major
        X = X.reshape((Pj, N), order="F")
        for t in range(0, Pj - 1, TILE):
            k1 = min(t + TILE, Pj)
            if k1 >= Pj:
                break                      # the last tile is a diagonal block: no trailing update below it
            l = L[k1:, t:k1]               # L21 of this panel
            x = X[t:k1, :]                 # X1
            with np.errstate(invalid="ignore"):
                lmax, lmin = l.max(axis=1), np.where(np.isinf(l), np.inf, l).min(axis=1)
                xmax, xmin = x.max(axis=0), np.where(np.isinf(x), np.inf, x).min(axis=0)
            span_L.append((lmax - lmin)[np.isfinite(lmax - lmin)])
            span_X.append((xmax - xmin)[np.isfinite(xmax - xmin)])
            best = (l[:, :, None] + x[None, :, :]).max(axis=1)          # log2 of the largest term of every output
            lo = lmax[:, None] + xmax[None, :] - best
            loss.append(lo[np.isfinite(lo)])
    cat = lambda a: np.concatenate(a) if a else np.zeros(1)
    return cat(span_L), cat(span_X), cat(loss)


def main():
    scale = float(sys.argv[1])
    its = sorted(int(a) for a in sys.argv[2:])
    c = synthetic.config("C4", scale)
    sdp, src = synthetic.make_lazy(c["dims"], c["num_points"], c["N"], c["precision"], c["seed"])
    o = Oracle(sdp, c["precision"], parity.DEFAULT_PARAMS, param_prec=0, block_source=src)
    print(f"C4 x{scale}: J={sdp.J} N={sdp.N} P_tot={sdp.P_total} --precision {c['precision']}; tiles of {TILE} columns of L21 / rows of X1")
    print("iteration   span of a row of an L21 tile (max / 99.9 % / median)   span of a column of an X1 tile   bits lost per output (max / 99.9 % / median)")
    done = 0
    for it in its:
        while done < it:
            if o.iterate():
                print("terminated:", o.terminate_reason)
                return
            done += 1
        sl, sx, lo = measure(o, sdp, sdp.N)
        q = lambda a: f"{a.max():7.1f} / {np.percentile(a, 99.9):6.1f} / {np.median(a):5.1f}"
        print(f"{it:9d}   {q(sl)}              {q(sx)}          {q(lo)}", flush=True)


if __name__ == "__main__":
    main()
