#!/usr/bin/env python
"""The reference's ONLY performance instrumentation is its epoch print, `Epoch %d [%.1fs]` (MF/train_new_api.py:1110), and the evaluation
times behind it (:1135,1148,1165).  This runs the drop-in CLI (python -m pda_amd.train_new_api, in-process) on a Douban-SHAPED synthetic
(47 890 users x 26 047 items, ~6.7 M train pairs, ten slots: the data itself is a missing blob of the reference tree) for a few epochs with an
evaluation at every `--log_interval`, and reports the wall-clock the CLI itself prints: seconds per train epoch (sampler + 3 272 steps of the
reference's optimiser) and per evaluation pass.

usage: python tools/cli_epoch.py [--users N --items N --mean-hist N] [--epochs 3] [--train s_condition|normal] [--keep DIR]
Prints one JSON object."""
import argparse
import contextlib
import io
import json
import os
import re
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_shaped_dataset(root, n_users, n_items, mean_hist, n_slots=10, seed=2020):
    """Vectorised writer of the reference's on-disk formats (pda_amd.synthetic.write_dataset is a per-user Python loop: fine for 600 users)."""
    import pandas as pd

    from pda_amd import pop_pre
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    sigma = 0.8
    lens = np.clip(np.exp(rng.standard_normal(n_users) * sigma + np.log(mean_hist) - 0.5 * sigma * sigma), 3, n_items // 2).astype(np.int64)
    u = np.repeat(np.arange(n_users, dtype=np.int64), lens)
    w = 1.0 / np.arange(1, n_items + 1)
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(n_items)
    it = perm[np.minimum(np.searchsorted(cdf, rng.random(u.size)), n_items - 1)]
    key = np.unique(np.r_[u * n_items + it, np.int64(n_items - 1)])      # (user, item) pairs once; the largest item id occurs (user 0)
    u, it = key // n_items, key % n_items
    slot = rng.integers(0, n_slots, u.size)
    first = np.r_[True, u[1:] != u[:-1]]                   # every user has train rows in two slots
    slot[first] = 0
    second = np.r_[False, first[:-1]] & ~first
    slot[second] = n_slots - 2
    tr = slot < n_slots - 1
    pd.DataFrame({"u": u[tr], "i": it[tr], "t": slot[tr], "s": 5}).to_csv(os.path.join(root, "train_with_time.txt"), sep=" ", header=False, index=False)

    def write_lists(name, uu, ii):
        order = np.argsort(uu, kind="stable")
        uu, ii = uu[order], ii[order]
        cuts = np.flatnonzero(np.r_[True, uu[1:] != uu[:-1], True])
        with open(os.path.join(root, name), "w") as f:
            for a, b in zip(cuts[:-1], cuts[1:]):
                f.write("%d %s\n" % (uu[a], " ".join(map(str, ii[a:b].tolist()))))
    write_lists("train.txt", u[tr], it[tr])
    is_test = rng.random(n_users) < 0.7
    last = ~tr
    write_lists("test.txt", u[last & is_test[u]], it[last & is_test[u]])
    write_lists("valid.txt", u[last & ~is_test[u]], it[last & ~is_test[u]])
    stages = []
    for t in range(n_slots):
        c = np.bincount(it[slot == t], minlength=n_items)
        nz = np.flatnonzero(c)
        stages.append(list(zip(nz.tolist(), c[nz].tolist())))
    pop_pre.write_popularity(root, pop_pre.compute_popularity(stages, n_item=n_items))
    return int(tr.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=47890)
    ap.add_argument("--items", type=int, default=26047)
    ap.add_argument("--mean-hist", type=int, default=240, help="before the (user, item) pairs are made unique: 240 leaves ~6.7 M train pairs")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--train", default="s_condition")
    ap.add_argument("--keep", default=None)
    ap.add_argument("--extra", default="", help="more CLI flags, space separated")
    a = ap.parse_args()
    root = a.keep or tempfile.mkdtemp(prefix="pda_cli_epoch_")
    t0 = time.time()
    n_train = write_shaped_dataset(os.path.join(root, "shaped"), a.users, a.items, a.mean_hist)
    t_write = time.time() - t0
    from pda_amd import train_new_api as T
    argv = ["--data_path", root, "--dataset", "shaped", "--train", a.train, "--test", a.train, "--epoch", str(a.epochs), "--log_interval", "1",
            "--batch_size", "2048", "--lr", "1e-3", "--regs", "1e-2", "--valid_set", "valid", "--pop_exp", "0.22", "--save_dir", root + "/ck/",
            "--Ks", "[20,50]", "--save_flag", "0", "--saveID", "cli", "--cuda", "0"] + a.extra.split()
    buf = io.StringIO()
    # the CLI prints its epoch time with %.1f like the reference; an epoch takes ~0.1 s here, so the tool also records WHEN the CLI reads its
    # clock: an epoch's time() - t1 is the first reading of the epoch minus the last reading before it (MF/train_new_api.py:1075,1110,1233)
    stamps = []

    def clock():
        stamps.append(time.time())
        return stamps[-1]
    T.time = clock
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        T.main(argv)
    wall = time.time() - t0
    T.time = time.time
    out = buf.getvalue()
    per_epoch = 6 if a.train == "s_condition" else 4          # clock readings of one epoch with an evaluation (1 before the loop)
    fine = [stamps[1 + k * per_epoch] - stamps[k * per_epoch] for k in range(a.epochs) if 1 + k * per_epoch < len(stamps)]
    ep = [float(x) for x in re.findall(r"Epoch \d+ \[([0-9.]+)s\]", out)]
    ev = [int(x) for x in re.findall(r"testing : time:\s+(\d+)", out)] + [float(x) for x in re.findall(r"test: time: ([0-9.]+)", out)]
    res = {"shape": "%d users x %d items, %d train pairs, B=2048 (%d steps per epoch)" % (a.users, a.items, n_train, n_train // 2048 + 1),
           "cli": "python -m pda_amd.train_new_api " + " ".join(argv[4:]),
           "epoch_print_s": ep, "note_epoch_print": "Epoch k [..s] as the CLI prints it (MF/train_new_api.py:1110): from the end of the previous epoch's evaluations to the end of this "
                                                    "epoch's steps; epoch 0 includes the first-call set-up (sampler tables, history CSR)",
           "epoch_s": fine, "train_epoch_s": (min(fine[1:]) if len(fine) > 1 else (fine[0] if fine else None)), "eval_print_s": ev,
           "eval_epoch_s": [stamps[(k + 1) * per_epoch] - stamps[1 + k * per_epoch] for k in range(a.epochs) if (k + 1) * per_epoch < len(stamps)], "main_wall_s": wall, "dataset_write_s": t_write,
           "steps_per_epoch": n_train // 2048 + 1}
    if res["train_epoch_s"]:
        res["us_per_step_through_the_cli"] = res["train_epoch_s"] / res["steps_per_epoch"] * 1e6
    print(json.dumps(res))
    sys.stderr.write(out[-3000:])
    if not a.keep:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
