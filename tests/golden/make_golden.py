#!/usr/bin/env python
"""Copy the reference's own end-to-end golden DATA (SDP inputs in the JSON directory
format + the iteration traces / final values sdpb produced for them) into
tests/golden/ so the parity tests run without /root/reference (absent on the GPU box).

Source: /root/reference/test/data/end-to-end_tests/<case>/output/{sdp,out}
(the fixtures test/src/integration_tests/cases/end-to-end.test.cxx diffs against).
Only data files are copied — no reference source code.

Also writes cases.json: the sdpb flags each case is run with, transcribed from
end-to-end.test.cxx (line numbers in the "source" field).
"""
import json
import os
import shutil
import sys

REF = "/root/reference/test/data/end-to-end_tests"
HERE = os.path.dirname(os.path.abspath(__file__))

SINGLET_ARGS = {  # end-to-end.test.cxx:296-303
    "dualityGapThreshold": "1.0e-30", "primalErrorThreshold": "1.0e-30",
    "dualErrorThreshold": "1.0e-30", "initialMatrixScalePrimal": "1.0e20",
    "initialMatrixScaleDual": "1.0e20", "feasibleCenteringParameter": "0.1",
    "infeasibleCenteringParameter": "0.3", "stepLengthReduction": "0.7",
    "maxComplementarity": "1.0e100", "maxIterations": 1000}
ALLOWED_ARGS = dict(SINGLET_ARGS, primalErrorThreshold="1.0e-200",  # :348-357
                    dualErrorThreshold="1.0e-200", detectPrimalFeasibleJump=1,
                    detectDualFeasibleJump=1)
DFIBO_ARGS = {  # :267-275
    "findDualFeasible": 1, "findPrimalFeasible": 1, "initialMatrixScalePrimal": "1e10",
    "initialMatrixScaleDual": "1e10", "maxComplementarity": "1e30",
    "dualErrorThreshold": "1e-10", "primalErrorThreshold": "1e-153", "maxIterations": 1000,
    "feasibleCenteringParameter": "0.1", "infeasibleCenteringParameter": "0.3",
    "stepLengthReduction": "0.7"}

CASES = {
    # name: (reference subdir, precision, params, iterations file, source)
    "1d": ("1d", 664, {}, "iterations.json", "end-to-end.test.cxx:186-200"),
    "1d-old-sampling": ("1d-old-sampling", 768, {}, "iterations.json", "end-to-end.test.cxx:202-214"),
    "1d-duplicate-poles": ("1d-duplicate-poles", 768, {}, "iterations.json", "end-to-end.test.cxx:216-226"),
    "1d-constraints": ("1d-constraints", 768, {}, "iterations.json", "end-to-end.test.cxx:228-235"),
    "dfibo": ("dfibo-0-0-j=3-c=3.0000-d=3-s=6", 768, DFIBO_ARGS, "iterations.json",
              "end-to-end.test.cxx:260-287"),
    "singlet_cT": ("SingletScalar_cT_test_nmax6/primal_dual_optimal", 768, SINGLET_ARGS,
                   "iterations.1.json", "end-to-end.test.cxx:289-316"),
    "singlet_allowed_primal_jump": ("SingletScalarAllowed_test_nmax6/primal_feasible_jump", 768,
                                    ALLOWED_ARGS, "iterations.json", "end-to-end.test.cxx:341-369"),
    "singlet_allowed_dual_jump": ("SingletScalarAllowed_test_nmax6/dual_feasible_jump", 768,
                                  ALLOWED_ARGS, "iterations.json", "end-to-end.test.cxx:370-380"),
}


def main():
    meta = {}
    for name, (sub, prec, params, itfile, source) in CASES.items():
        src = os.path.join(REF, sub, "output")
        dst = os.path.join(HERE, name)
        if os.path.isdir(dst):
            shutil.rmtree(dst)
        os.makedirs(os.path.join(dst, "sdp"))
        for f in sorted(os.listdir(os.path.join(src, "sdp"))):
            if f.startswith(("block_data_", "block_info_")) or f in (
                    "control.json", "objectives.json", "normalization.json"):
                shutil.copy(os.path.join(src, "sdp", f), os.path.join(dst, "sdp", f))
        shutil.copy(os.path.join(src, "out", itfile), os.path.join(dst, "iterations.json"))
        xs = sorted(f for f in os.listdir(os.path.join(src, "out")) if f.startswith("x_") and f.endswith(".txt"))
        for f in ["out.txt", "y.txt"] + xs:
            shutil.copy(os.path.join(src, "out", f), os.path.join(dst, f))
        cmby = os.path.join(src, "out", "c_minus_By", "c_minus_By.json")
        if os.path.exists(cmby):
            shutil.copy(cmby, os.path.join(dst, "c_minus_By.json"))
        meta[name] = {"precision": prec, "params": params, "source": source,
                      "reference_dir": "test/data/end-to-end_tests/" + sub}
    # the reference's zip-archived SDP (test/src/integration_tests/main.cxx:23; pmp2sdp --zip output):
    # fixture for the archive reader (sdpb_amd/sdp_io.py, Archive_Reader.cxx)
    shutil.copy(os.path.join(os.path.dirname(os.path.dirname(REF)), "data", "sdp.zip"), os.path.join(HERE, "sdp.zip"))
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(meta), "cases to", HERE)


if __name__ == "__main__":
    sys.exit(main())
