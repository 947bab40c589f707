"""Per-stage item popularity pre-compute: drop-in for the reference's pop_pre.py (same CLI: --path --slot_count).

Reads t_0.txt .. t_{T-1}.txt ("iid uid uid ..."), computes pop[t][i] = (cnt+1)/(total_t + n_item) with
1/(total_t + n_item) for items absent from the stage (pop_pre.py:31-36), min-max normalises every stage to
[0,1] (:41-42) and writes item_pop_seq_ori2.txt, one line "iid p0 ... p_{T-1}" per item (:48-57).
Host-side numpy: this runs once per dataset and is not on the hot path.
"""
from __future__ import annotations

import argparse
import os

import numpy as np


def read_stage_counts(root: str, slot_count: int):
    stages = []
    for i in range(slot_count):
        rows = []
        with open(os.path.join(root, "t_{}.txt".format(i))) as f:
            for line in f:
                parts = line.split()
                if parts:
                    rows.append((int(parts[0]), len(parts) - 1))
        stages.append(rows)
    return stages


def compute_popularity(stages, n_item=None) -> np.ndarray:
    """-> float64 [T, n_item].  n_item defaults to the number of distinct item ids seen in any stage
    (pop_pre.py:13-19); item ids must then be < n_item, as the reference assumes."""
    if n_item is None:
        n_item = len({it for st in stages for it, _ in st})
    pop = np.empty((len(stages), n_item), dtype=np.float64)
    for t, st in enumerate(stages):
        total = sum(c for _, c in st)
        pop[t, :] = 1.0 / (total + n_item)
        for it, c in st:
            pop[t, it] = (c + 1.0) / (total + n_item)
        lo, hi = pop[t].min(), pop[t].max()
        pop[t] = (pop[t] - lo) / (hi - lo)
    return pop


def write_popularity(root: str, pop: np.ndarray, name: str = "item_pop_seq_ori2.txt"):
    with open(os.path.join(root, name), "w") as f:
        for i in range(pop.shape[1]):
            f.write(str(i) + " " + " ".join(str(p) for p in pop[:, i]) + "\n")


def main(argv=None):
    ap = argparse.ArgumentParser(description="Run pop_bias.")
    ap.add_argument("--path", nargs="?", default="data/ml_10m/", help="Input data path.")
    ap.add_argument("--slot_count", type=int, default=13, help="number of stages T")
    a = ap.parse_args(argv)
    pop = compute_popularity(read_stage_counts(a.path, a.slot_count))
    print("tot information:\nmean:", pop.mean(axis=1))
    print("max:", pop.max(axis=1))
    print("min:", pop.min(axis=1))
    write_popularity(a.path, pop)


if __name__ == "__main__":
    main()
