"""bella_amd.evaluate (host-side mirror of benchmark/evaluation.h) against known answers of the reference's own evaluator
(tests/golden/eval_kat.json, made by oracle/make_golden.py from benchmark/evaluation.cpp built as oracle/_ref/bella_eval)."""
import json
import os

import pytest

from bella_amd import evaluate as ev
from conftest import GOLD, load_golden

KAT = json.load(open(os.path.join(GOLD, "eval_kat.json")))


@pytest.mark.parametrize("key", sorted(KAT))
def test_metrics_match_reference_evaluator(key):
    name, mo = key.split("/")
    g = load_golden(name)
    G = ev.truth_pairs(ev.truth_from_names(g.names), int(mo))
    S = ev.read_bella_output(g.out["align"], int(mo))
    r = ev.evaluate(S, G)
    k = KAT[key]
    assert (len(G), 2 * len(S), 2 * r["true_positives"]) == (k["truth"], k["reported_x2"], k["true_positives_x2"])
    for f in ("recall", "precision", "f1"):
        assert ("%.2f" % r[f]).replace("nan", "-nan").replace("--", "-") == k[f] or ("%.2f" % r[f]) == k[f], f


def test_skip_alignment_files_are_rejected_like_the_reference():
    g = load_golden("toy120")
    assert ev.read_bella_output(g.out["skip"], 500) == set()          # "Entry of size 6" (evaluation.h:203-209)
