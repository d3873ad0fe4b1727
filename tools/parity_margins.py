#!/usr/bin/env python
"""Folds a LFDM_PARITY_LOG file (tests/util.py: one JSON object per comparison) into a per-fixture summary: for every test the
comparison that used the largest fraction of its tolerance.  usage: parity_margins.py LOG OUT.json"""
import collections
import json
import sys

rows = [json.loads(ln) for ln in open(sys.argv[1]) if ln.strip()]
by_test = collections.OrderedDict()
for r in rows:
    cur = by_test.get(r["test"])
    if cur is None or (r["fraction_of_bar"] or 0) > (cur["fraction_of_bar"] or 0):
        by_test[r["test"]] = r
by_test_n = collections.Counter(r["test"] for r in rows)
out = {"comparisons": len(rows), "tests": len(by_test),
       "worst_overall": max(rows, key=lambda r: r["fraction_of_bar"] or 0) if rows else None,
       "per_test_worst": [dict(v, comparisons=by_test_n[k]) for k, v in by_test.items()]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
w = out["worst_overall"]
print("%d comparisons in %d tests; largest fraction of a bar: %.3f (%s: %s)" % (len(rows), len(by_test), w["fraction_of_bar"], w["test"], w["what"]))
