"""Quality evaluation of an overlap file against a ground truth: the host-side mirror of the reference's
benchmark/evaluation.{h,cpp} for BELLA's own 12-column output.

* ground truth (evaluation.h:45-158): for every pair of reads placed on the same reference sequence, the length of the
  intersection of their intervals as benchmark/IntervalTree.h:166-200 computes it; the pair is a true overlap when that is
  >= minOverlap.  The set holds BOTH orders (a,b) and (b,a) -- every interval queries the tree of all the others.
* the tool's answer (readBellaOutput, evaluation.h:160-227): lines with exactly 12 tab-separated columns, a != b; with
  alignment (default) only those whose column 5 (the overlap estimate) is >= minOverlap.  Keyed by the ORDERED pair as written.
* metrics (evaluate, evaluation.h:591-628): T = G ∩ S, recall = 2|T|/|G| (BELLA writes each pair once), precision = |T|/|S|,
  F1 = harmonic mean; percentages.
"""
from __future__ import annotations

import numpy as np


def truth_from_names(names):
    """synthetic reads are named r<idx>_<start>_<len>_<strand>: -> (ref, read, start, end) records"""
    out = []
    for n in names:
        _, st, ln, _ = n.rsplit("_", 3)
        out.append(("genome", n, int(st), int(st) + int(ln)))
    return out


def truth_pairs(records, min_overlap: int = 2000):
    """records: (ref, read, start, end).  Returns the set of ordered (a, b) with truth overlap >= min_overlap, both orders."""
    by_ref = {}
    for ref, read, s, e in records:
        by_ref.setdefault(ref, []).append((int(s), int(e), read))
    G = set()
    for ivs in by_ref.values():
        ivs.sort()
        starts = np.asarray([x[0] for x in ivs], np.int64)
        stops = np.asarray([x[1] for x in ivs], np.int64)
        n = len(ivs)
        for q in range(n):
            qs, qe, qn = ivs[q]
            # candidates: intervals starting at or before the query's end (sorted by start); IntervalTree.h:175-187
            hi = int(np.searchsorted(starts, qe, side="right"))
            for i in range(hi):
                if i == q:
                    continue
                s, e, name = ivs[i]
                if name == qn:
                    continue
                al = 0
                if s <= qs:
                    if e >= qs:
                        al = min(e - qs, qe - qs)
                elif qe > s:
                    al = min(qe - s, e - s)
                if al >= min_overlap:
                    G.add((name, qn))
    return G


def read_bella_output(data, min_overlap: int = 2000, alignment: bool = True):
    """data: bytes/str of a BELLA 12-column file, or a path.  Returns the set of ordered (a, b) (evaluation.h:160-227)."""
    if isinstance(data, str) and "\t" not in data and "\n" not in data:
        with open(data, "rb") as f:
            data = f.read()
    if isinstance(data, bytes):
        data = data.decode()
    S = set()
    for line in data.split("\n"):
        if not line:
            continue
        v = line.split("\t")
        if len(v) != 12:                                  # "Entry of size N": 6-column --skip-alignment files are rejected
            continue
        if v[0] == v[1]:
            continue
        if alignment and int(v[4]) < min_overlap:
            continue
        S.add((v[0], v[1]))
    return S


def evaluate(S, G, duplicate: bool = True):
    """-> dict(recall, precision, f1, true_positives, reported, truth) in percent like the reference prints them"""
    T = S & G
    nan = float("nan")
    rc = (2.0 if duplicate else 1.0) * len(T) / len(G) * 100.0 if G else nan
    pr = len(T) / len(S) * 100.0 if S else nan
    f1 = 2 * rc * pr / (rc + pr) if (rc + pr) > 0 else nan
    return {"recall": rc, "precision": pr, "f1": f1, "true_positives": len(T), "reported": len(S), "truth": len(G)}
