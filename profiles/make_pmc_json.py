"""profiles/pmc_k_syrk_fx.json from the two per-kernel PMC summaries (summarize_pmc_db.py output of the separate
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`).
Records the source digest of the dominant kernel so that bench.py reports `roofline.traffic` only for the
build that was measured.   usage: python profiles/make_pmc_json.py <fetch.txt> <write.txt> <tag>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def per_kernel(path):
    out = {}
    with open(path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            parts = line.rsplit(None, 2)
            out[parts[0].strip()] = (int(parts[1]), float(parts[2]))
    return out


def main():
    fetch, write, tag = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    main_k = [k for k in fetch if k.startswith("k_syrk_fx")][0]
    side = [k for k in fetch if k.startswith(("k_syrk4_finish", "k_syrk5_finish", "k_syrk_reduce", "k_syrk3_sum_splits"))]
    copy = [k for k in fetch if k.startswith("k_copy16")]
    f_kb = fetch[main_k][1]                            # 16 B/lane global_load_lds: counted half
    f_side_kb = sum(fetch[k][1] for k in side)        # 4 B/lane loads of the finishing kernel: counted in full
    w_kb = write[main_k][1] + sum(write.get(k, (0, 0.0))[1] for k in side)
    rec = {
        "kernel": main_k + "".join(" + " + k for k in side),
        "workload": "C4 J=600 N=1000 P_tot=40000 p=512, 1 GPU (bench.py --steps 2 --warmup 1 under rocprofv3 --pmc, one counter per pass)",
        "kernel_source_digest": bench.syrk_source_digest(),
        "FETCH_SIZE_KB_per_launch_raw": f_kb, "FETCH_SIZE_KB_per_launch_finishing_kernels": f_side_kb, "WRITE_SIZE_KB_per_launch_raw": w_kb,
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": 2.0 * f_kb * 1024 + f_side_kb * 1024 + w_kb * 1024,
        "calibration": ("same runs: k_copy16 (16 B/lane, 1 GiB in, 1 GiB out) reads FETCH_SIZE %.1f KB and writes WRITE_SIZE %.1f KB "
                        "(MI355X_MICROARCH.md: 16 B/lane streams count half in FETCH_SIZE on gfx950; the syrk fetches its pieces with "
                        "16 B/lane global_load_lds_dwordx4)" % (fetch[copy[0]][1], write[copy[0]][1])) if copy else None,
        "files": [os.path.basename(sys.argv[1]), os.path.basename(sys.argv[2])], "tag": tag,
    }
    with open(os.path.join(ROOT, "profiles", "pmc_k_syrk_fx.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
