#!/usr/bin/env python3
"""Fold a parity ledger (tests/helpers.py::check, one JSON line per comparison, written when QREC_PARITY_LOG is set) into
profiles/rNN_parity_errors.json: per (test, quantity) the number of comparisons, the worst observed error and the bound it
was held to, sorted by observed / bound.

    python tools/summarize_parity.py gpurun_out/parity_r03.jsonl profiles/r03_parity_errors.json
"""
import json
import sys


def main(src, dst):
    rows = {}
    for line in open(src):
        r = json.loads(line)
        key = (r["test"], r["quantity"], r["bound"])       # (a test may hold the same quantity to two bounds: fp64 and fp32 tables)
        e = rows.setdefault(key, dict(test=r["test"], quantity=r["quantity"], kind=r.get("kind", "parity"), n=0, observed=0.0, bound=r["bound"]))
        e["n"] += 1
        e["observed"] = max(e["observed"], r["observed"])
        e["bound"] = min(e["bound"], r["bound"])
    out = sorted(rows.values(), key=lambda e: -(e["observed"] / e["bound"] if e["bound"] else 0.0))
    over = [e for e in out if e["bound"] > 1e-5 * (1 + 1e-9) and e["kind"] == "parity"]
    by_kind = {}
    for e in out:
        k = by_kind.setdefault(e["kind"], dict(distinct=0, above_1e_5=0))
        k["distinct"] += 1
        k["above_1e_5"] += e["bound"] > 1e-5 * (1 + 1e-9)
    doc = dict(source=src, comparisons=sum(e["n"] for e in out), distinct=len(out),
               bounds_above_1e_5=len(over), by_kind=by_kind,
               kinds="tests/helpers.py::check -- parity: the 1e-5 / bit-exact contract; floor: bound derived from the reference's own distance to exact "
                     "arithmetic; discontinuity: may leave the recorded run at sign() / top_k (its recorded-pattern twin is the parity row); "
                     "statistical: throughput-mode Recall / loss gaps; partition: this repository's multi-rank partitions against its own single-GPU run; info: recorded only",
               worst_observed_over_bound=(out[0]["observed"] / out[0]["bound"] if out else None), rows=out)
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"{doc['comparisons']} comparisons, {len(out)} distinct, {len(over)} PARITY rows held to a bound above 1e-5; by kind: {by_kind}")
    for e in over:
        print(f"  {e['observed']:.2e} / {e['bound']:.0e}  {e['test'].split('::')[-1]}  {e['quantity'][:70]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
