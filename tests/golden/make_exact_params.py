#!/usr/bin/env python
"""tests/golden/exact_params.json: for every golden case, the parameter VALUES the reference run used — sdpb
parses its options before --precision is applied, i.e. at GMP's initial 64-bit precision (visible in the golden
traces: beta = 0.2999...98725e-58 in 1d/iterations.json) — as exact decimals, produced with the oracle's GMP
(tests/parity.py: reference_params).  Data for bench.py's `--workload golden:<case>` gate, which may not call the
oracle.     python tests/golden/make_exact_params.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    from oracle.oracle import Oracle
    from tests import parity
    out = {}
    for name, meta in sorted(parity.cases().items()):
        sdp, meta, _, _ = parity.load_case(name)
        o = Oracle(sdp, meta["precision"], meta["params"], param_prec=64)
        out[name] = parity.reference_params(meta["params"], o)
        o.close()
    with open(os.path.join(HERE, "exact_params.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote exact_params.json for", ", ".join(out))


if __name__ == "__main__":
    main()
