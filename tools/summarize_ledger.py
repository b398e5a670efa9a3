#!/usr/bin/env python3
"""profiles/r04_parity_ledger.md from the parity ledger of one full GPU test run:

    PYGCN_LEDGER_ONLY=1 PYGCN_LEDGER=gpurun_out/r4/ledger_all.json python -m pytest tests -m gpu -q
    python tools/summarize_ledger.py gpurun_out/r4/ledger_all.json > profiles/r04_parity_ledger.md

(tests/conftest.py records every normwise comparison — assert_normwise / assert_parity — with the
error it measured; PYGCN_LEDGER_ONLY=1 records instead of failing, so one run prices every gate)."""
import collections
import json
import re
import sys


def scheme(test):
    m = re.search(r"\[(.*)\]", test)
    for s in ("bf16x3", "h2"):
        if m and s in m.group(1).split("-"):
            return s
    return "—"


def main():
    L = json.load(open(sys.argv[1]))
    print("# r04 — parity ledger: every normwise comparison of the GPU test suite, with the error it measured\n")
    print(f"{len(L)} comparisons (`tests/conftest.py`: `assert_normwise`, `assert_parity`), "
          f"{sum(1 for e in L if not e['ok'])} outside their gate.  Metric: `max|got − ref| / max|ref|` "
          "(SURVEY §7).  Tests that take the `gemm_scheme` fixture run under both decompositions of the "
          "256-wide fp32 GEMMs (`bf16x3` = the default, fp32-equivalent; `h2` = two scaled fp16 parts).\n")
    cls = collections.defaultdict(lambda: [0, 0.0, 0.0, ""])
    for e in L:
        w = e["what"]
        if "[vs float64]" in w:
            k = "reduction gradients vs the float64 evaluation of the same step (gate 1e-5)"
        elif "float32 reference" in w:
            k = "reduction gradients vs the float32 oracle (gate 1e-5 + the oracle's own distance from float64)"
        else:
            k = f"gate {e['gate']:.0e}"
        c = cls[(k, scheme(e["test"]))]
        c[0] += 1
        if e["err_over_scale"] >= c[1]:
            c[1], c[3] = e["err_over_scale"], e["test"].split("::")[-1] + " — " + w
        c[2] = max(c[2], e["allowance"])
    print("| gate class | GEMM scheme | comparisons | largest measured error | largest allowance (the oracle's own error) | where |")
    print("|---|---|---:|---:|---:|---|")
    for (k, s), c in sorted(cls.items()):
        print(f"| {k} | {s} | {c[0]} | {c[1]:.2e} | {c[2]:.2e} | `{c[3][:110]}` |")
    print("\n## The comparisons closest to their gate\n")
    print("| share of the gate used | error | gate (+ allowance) | scheme | test — what |")
    print("|---:|---:|---|---|---|")
    rows = sorted(L, key=lambda e: -(e["err_over_scale"] / max(e["gate"] + e["allowance"], 1e-300)))
    for e in rows[:20]:
        lim = e["gate"] + e["allowance"]
        print(f"| {e['err_over_scale'] / lim:.2f} | {e['err_over_scale']:.2e} | {e['gate']:.0e} + {e['allowance']:.1e} | "
              f"{scheme(e['test'])} | `{e['test'].split('::')[-1][:70]}` — {e['what'][:70]} |")


if __name__ == "__main__":
    main()
