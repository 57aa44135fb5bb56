#!/usr/bin/env python3
"""Per-kernel table from the SQ counter passes of one bench command (profiles/summarize_pmc_db.py output, two passes:
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT
SQ_LDS_IDX_ACTIVE and SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES)
and the kernel-trace statistics of the same command (profiles/tools/rocpd_stats.py).

  python profiles/tools/sq_table.py <pmc_SQ.txt> <pmc_SQ_INSTS.txt> <kernel_stats.txt> [iterations_in_the_trace]

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts;
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (ready, waiting for an issue slot) + ACTIVE_INST_ANY (issuing)
~ WAVE_CYCLES.  'VALU busy' = ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x launch duration x 2.4 GHz); 'LDS active' =
LDS_IDX_ACTIVE / (256 CUs x launch duration x 2.4 GHz)."""
import re
import sys


def read_counters(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"# kernel, launches, avg (\S+) per launch", line)
        if m:
            cur = m.group(1)
            continue
        if line.startswith("#") or not line.strip() or cur is None:
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([0-9.]+)\s*$", line)
        if m:
            out.setdefault(m.group(1).strip(), {})[cur] = float(m.group(3))
    return out


def read_stats(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+\s+(?:void )?(?:sdpb::)?(.+)$", line)
        if m:
            name = re.sub(r"\(.*$", "", m.group(4)).strip()
            out[name] = (int(m.group(1)), float(m.group(2)), float(m.group(3)))
    return out


def main():
    sq, inst, stats = read_counters(sys.argv[1]), read_counters(sys.argv[2]), read_stats(sys.argv[3])
    iters = float(sys.argv[4]) if len(sys.argv) > 4 else None
    print(__doc__.split("\n\n")[0].split("\n")[0])
    print(f"# {sys.argv[1]} + {sys.argv[2]} + {sys.argv[3]}" + (f"; {iters:g} iterations in the trace" if iters else ""))
    print(f"{'kernel':38s} {'launch':>7s} {'avg us':>9s} {'ms/iter':>8s} | {'parked':>6s} {'wait':>6s} {'issue':>6s} {'VALU':>6s} | {'VALU busy':>9s} {'LDS act':>7s} {'bankcf':>6s} | "
          f"{'VALU inst':>10s} {'LDS inst':>9s} {'VMEM rd':>9s} {'waves':>8s}")
    rows = []
    for k, c in sq.items():
        key = k.replace(", ", ",")
        st = next((v for n, v in stats.items() if n.replace(", ", ",").startswith(key)), None)
        if not st or "SQ_WAVE_CYCLES" not in c:
            continue
        calls, total_ms, avg_us = st
        wc = c["SQ_WAVE_CYCLES"]
        cyc = avg_us * 1e-6 * 2.4e9
        i = inst.get(k, {})
        rows.append((total_ms, f"{k[:38]:38s} {calls:7d} {avg_us:9.1f} {(total_ms / iters if iters else total_ms):8.2f} | "
                     f"{100 * c.get('SQ_WAIT_ANY', 0) / wc:5.1f}% {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f}% {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f}% "
                     f"{100 * c.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.1f}% | {100 * c.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (1024 * cyc):8.1f}% "
                     f"{100 * c.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * cyc):6.1f}% {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1):5.1f}% | "
                     f"{i.get('SQ_INSTS_VALU', 0):10.3g} {i.get('SQ_INSTS_LDS', 0):9.3g} {i.get('SQ_INSTS_VMEM_RD', 0):9.3g} {i.get('SQ_WAVES', 0):8.3g}"))
    for _, r in sorted(rows, reverse=True):
        print(r)


if __name__ == "__main__":
    main()
