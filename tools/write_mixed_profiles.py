#!/usr/bin/env python3
"""Writes the committed records of the mixed workload from what the GPU passes left under gpurun_out/:
  profiles/<tag>_mixed_serial_kernel_stats.md  from gpurun_out/serial_stats.txt   (PQD_OVERLAP=0 tools/gpu_prof_cmd.sh mixed_serial bench.py ...)
  profiles/<tag>_mixed_pmc_sq.md               from gpurun_out/pmc_sq.txt, pmc_wait.txt   (tools/gpu_pmc_sq.sh mixed 1, tools/gpu_pmc_wait.sh mixed 96)
  profiles/<tag>_bench_mixed_pcie.json         from gpurun_out/bench_pcie.log     (python bench.py --pcie)
usage: python tools/write_mixed_profiles.py r03"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
CLOCK_GHZ = 1.9  # shader clock under this load (tools/str_timeline.py)


def main():
    tag = sys.argv[1]
    rows = []
    for l in open(os.path.join(G, "serial_stats.txt")):
        m = re.match(r"^(\S.*?)\s+calls\s+(\d+) avg\s+([\d.]+) us\s+([\d.e+-]+)%", l)
        if m:
            rows.append(m.groups())
    line = [l for l in open(os.path.join(G, "prof_mixed_serial", "run.log")) if l.startswith("{")][-1]
    out = ["# rocprofv3 --kernel-trace --stats, mixed workload (96 row groups), streams serialised (PQD_OVERLAP=0)", "",
           "One stream: every kernel's time is its own (with the three streams of the default schedule the kernels stretch each other;",
           "`%s_mixed_rocprof_summary.md` is the same workload with the default schedule). Command: `PQD_OVERLAP=0 bash tools/gpu_prof_cmd.sh mixed_serial bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras`." % tag,
           "", "| kernel | calls | avg us | % |", "|---|---|---|---|"]
    out += ["| %s | %s | %s | %s |" % r for r in rows]
    out += ["", "bench line of the profiled run (10 steps, serialised): " + line.strip()[:420] + " ...", ""]
    open(os.path.join(P, "%s_mixed_serial_kernel_stats.md" % tag), "w").write("\n".join(out))

    sq = [l.rstrip() for l in open(os.path.join(G, "pmc_sq.txt")) if "per wave:" in l]
    wt = [l.rstrip() for l in open(os.path.join(G, "pmc_wait.txt")) if "wave_cyc/wave" in l]
    tot = 0.0
    note = []
    for l in sq:
        m = re.match(r"^(\S+)\s+waves\s+(\d+)\s+per wave: valu\s+(\d+)", l)
        if m and m.group(1) in ("k_str_fused", "lv4k::k_levels", "k_copy", "k_rows<true>"):
            n = int(m.group(2)) * int(m.group(3)) / 1e6
            tot += n
            note.append("%s %s waves x %s = %.0f M" % (m.group(1), m.group(2), m.group(3), n))
    rate = 1024 * CLOCK_GHZ / 4  # G wave-wide VALU instructions per second
    out = ["# SQ counters per kernel, mixed workload (96 row groups), streams serialised (PQD_OVERLAP=0)", "",
           "Two `rocprofv3 --pmc` passes (`tools/gpu_pmc_sq.sh mixed 1`, `tools/gpu_pmc_wait.sh mixed 96`), averages per wave.",
           "The shader clock measured inside k_str_fused (s_memtime against the 100 MHz wall clock, `tools/str_timeline.py`) is",
           "1.9-2.0 GHz under this load, so the 1024 SIMDs issue 1024 x %.1f GHz / 4 = %.0f G wave-wide VALU instructions per second." % (CLOCK_GHZ, rate),
           "VALU instructions of the four long kernels: " + "; ".join(note) + "; together %.0f M = %.2f ms of issue." % (tot, tot / rate), "",
           "## instructions per wave", "", "```"] + sq + ["```", "", "## wave cycles and waits (SQ_WAVE_CYCLES counts in units of four clocks)", "", "```"] + wt + ["```", ""]
    open(os.path.join(P, "%s_mixed_pmc_sq.md" % tag), "w").write("\n".join(out))

    bl = [l for l in open(os.path.join(G, "bench_pcie.log")) if l.startswith("{")][-1]
    json.dump(json.loads(bl), open(os.path.join(P, "%s_bench_mixed_pcie.json" % tag), "w"), indent=1)
    print("written")


if __name__ == "__main__":
    main()
