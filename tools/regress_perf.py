#!/usr/bin/env python3
"""Performance regression guard next to the bit guard (tools/regress_bits.py): the rocprofv3 MEDIAN duration of every per-iteration
kernel of configs 4 and 5 against tools/regress_perf.expected, with a relative band (default 3 %).  HISTORY A.6 records that merely
moving k_schur_gram's request code into a lambda cost 1 us (config 4) to 9 us (config 5): a kernel that sits on a
register-allocation optimum will lose that to a compiler update just as silently -- this makes it loud.

    bash tools/prof.sh <tag> ; bash tools/prof.sh <tag>_c5 --config 5            (GPU box: the two kernel traces)
    python tools/regress_perf.py gpurun_out/<tag>_kernel_medians.csv gpurun_out/<tag>_c5_kernel_medians.csv [--band 0.03] [--record]

Exit code 1 and one line per offending kernel if a median is SLOWER than expected by more than the band; kernels that got faster
by more than the band are listed as a hint to re-record (`--record` rewrites the expected file from the given medians).  Only
kernels with at least 100 launches in the trace count (the per-solve kernels' medians move with the number of solves).  The boxes
of this pool differ by 1-3 %: the band is for one box against the recorded one, an A/B of two builds still belongs in one call
(tools/ab_prof.sh)."""
import csv
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
EXPECTED = os.path.join(HERE, "regress_perf.expected")


def short(name):
    return re.sub(r"\(.*", "", name.replace("void ", "").replace("tscm::", "")).strip()


def medians(path):
    out = {}
    for row in csv.DictReader(open(path)):
        if "tscm::" in row["Name"] and int(row["Calls"]) >= 100:
            out[short(row["Name"])] = int(row["MedianNs"]) / 1e3
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    band = 0.03
    if "--band" in sys.argv:
        band = float(sys.argv[sys.argv.index("--band") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--band") + 1]]
    if len(args) != 2:
        raise SystemExit(__doc__)
    got = {"config4": medians(args[0]), "config5": medians(args[1])}
    if "--record" in sys.argv:
        with open(EXPECTED, "w") as f:
            f.write("# config kernel median_us   (tools/regress_perf.py --record; rocprofv3 medians of bench.py --steps 20 --warmup 10)\n")
            for cfg in ("config4", "config5"):
                for k, v in sorted(got[cfg].items()):
                    f.write(f"{cfg} {k} {v:.2f}\n")
        print(f"recorded {sum(len(v) for v in got.values())} kernel medians in {EXPECTED}")
        return
    want = {}
    for line in open(EXPECTED):
        if line.strip() and not line.startswith("#"):
            cfg, rest = line.split(None, 1)
            k, v = rest.rsplit(None, 1)
            want.setdefault(cfg, {})[k.strip()] = float(v)
    bad, hints = [], []
    for cfg in ("config4", "config5"):
        for k, w in sorted(want.get(cfg, {}).items()):
            g = got[cfg].get(k)
            if g is None:
                bad.append(f"{cfg} {k}: expected {w:.2f} us, the kernel is not in the trace (renamed? no longer launched per iteration?)")
            elif g > w * (1.0 + band):
                bad.append(f"{cfg} {k}: {g:.2f} us, expected {w:.2f} (+{100 * (g / w - 1):.1f} %, band {100 * band:.0f} %)")
            elif g < w * (1.0 - band):
                hints.append(f"{cfg} {k}: {g:.2f} us, expected {w:.2f} ({100 * (g / w - 1):.1f} %): faster -- re-record")
        for k in sorted(set(got[cfg]) - set(want.get(cfg, {}))):
            hints.append(f"{cfg} {k}: {got[cfg][k]:.2f} us, not in the expected file")
    for h in hints:
        print("note:", h)
    if bad:
        for b in bad:
            print("SLOWER:", b)
        sys.exit(1)
    print(f"regress_perf: {sum(len(v) for v in want.values())} kernel medians within {100 * band:.0f} % of tools/regress_perf.expected")


if __name__ == "__main__":
    main()
