"""Readers for the reference's on-disk formats and the device-side layouts built from them.

    train.txt / valid.txt / test.txt   "uid iid iid ..."            MF/load_data.py:48-105
    train_with_time.txt                "uid iid time stars"         MF/load_data.py:624-646
    item_pop_seq_ori2.txt              "iid p0 p1 ... p_{T-1}"      MF/train_new_api.py:862-880
    t_<k>.txt                          "iid uid uid ..."            pop_pre.py:13-36

`Data` / `Data2` expose the attributes the reference scripts read (`train_user_list`, `train_user_list_time`,
`valid_user_list`, `test_user_list`, `n_users`, `n_items`, `n_train`, `items`, `users`, `unique_times`,
`expo_popularity`, `batch_size`).  Differences, on purpose: `--data_path` is honoured (the reference
hard-codes ./data/<dataset>/, MF/load_data.py:27,619); lines are split on any whitespace.
"""
from __future__ import annotations

import collections
import os

import numpy as np


def _read_user_lists(path):
    out = collections.OrderedDict()
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for line in f:
            parts = line.split()
            if len(parts) < 2:          # "if len(items) == 0: continue"   MF/load_data.py:57-58
                continue
            ints = [int(x) for x in parts]
            out[ints[0]] = ints[1:]
    return out


class _Base:
    def __init__(self, args):
        self.path = os.path.join(args.data_path, args.dataset) + "/"
        self.batch_size = args.batch_size
        self.n_users = self.n_items = self.n_train = self.n_valid = self.n_test = 0
        self.train_user_list = collections.defaultdict(list)
        self.train_user_list_time = collections.defaultdict(list)
        self.valid_user_list = collections.defaultdict(list)
        self.test_user_list = collections.defaultdict(list)
        self.train_item_list = collections.defaultdict(list)
        self.unique_times = []
        self.expo_popularity = None
        if getattr(args, "data_type", "ori") != "ori":
            raise NotImplementedError("only --data_type ori is implemented")
        if getattr(args, "model", "mf") not in ("mf", "biasmf"):
            raise NotImplementedError("only can sampling for mf-type model")   # MF/load_data.py:708
        self._load(args)
        self.n_users += 1                                   # ids are 0-based: MF/load_data.py:93-94
        self.n_items += 1
        self.users = list(range(self.n_users))
        self.items = list(range(self.n_items))
        self.valid_users = list(self.valid_user_list.keys())
        self.test_users = set(self.test_user_list.keys())
        nnz = sum(len(v) for v in self.train_user_list.values())
        print("n_items:", self.n_items, "n_users:", self.n_users)
        print("sparsity:", 1.0 * nnz / self.n_items / self.n_users)

    def _load_eval_files(self):
        for name, tgt in (("valid.txt", self.valid_user_list), ("test.txt", self.test_user_list)):
            for u, items in _read_user_lists(self.path + name).items():
                tgt[u] = items
                self.n_users = max(self.n_users, u)
                self.n_items = max(self.n_items, max(items))
        self.n_valid = sum(len(v) for v in self.valid_user_list.values())
        self.n_test = sum(len(v) for v in self.test_user_list.values())

    def add_expo_popularity(self, popularity):               # MF/load_data.py:752-753
        self.expo_popularity = popularity

    # ---- device layouts ---------------------------------------------------------------------------
    def train_csr(self, device):
        """Train interactions as CSR by user id, item ids sorted ascending inside each row (kernel contract),
        with the parallel time-slot array (zeros for `Data`).  Returns (indptr i64, indices i32, slots i32)."""
        import torch
        lens = np.zeros(self.n_users, dtype=np.int64)
        for u, items in self.train_user_list.items():
            lens[u] = len(items)
        indptr = np.zeros(self.n_users + 1, dtype=np.int64)
        np.cumsum(lens, out=indptr[1:])
        idx = np.zeros(int(indptr[-1]), dtype=np.int32)
        slot = np.zeros(int(indptr[-1]), dtype=np.int32)
        for u, items in self.train_user_list.items():
            a = np.asarray(items, dtype=np.int32)
            order = np.argsort(a, kind="stable")
            idx[indptr[u]:indptr[u + 1]] = a[order]
            t = self.train_user_list_time.get(u) if isinstance(self.train_user_list_time, dict) else None
            if t:
                slot[indptr[u]:indptr[u + 1]] = np.asarray(t, dtype=np.int32)[order]
        return (torch.from_numpy(indptr).to(device), torch.from_numpy(idx).to(device), torch.from_numpy(slot).to(device))


class Data(_Base):
    """BPRMF loader (train.txt item lists; MF/load_data.py:24-120)."""

    def _load(self, args):
        for u, items in _read_user_lists(self.path + "train.txt").items():
            self.train_user_list[u] = items
            for it in items:
                self.train_item_list[it].append(u)
            self.n_users = max(self.n_users, u)
            self.n_items = max(self.n_items, max(items))
            self.n_train += len(items)
        self._load_eval_files()
        print(self.n_train, self.n_valid, self.n_test)


class Data2(_Base):
    """PD/PDA loader (train_with_time.txt; MF/load_data.py:617-708).  `train_user_list` is a plain dict, so an
    evaluation user without train items raises KeyError exactly like the reference."""

    def _load(self, args):
        import pandas as pd
        df = pd.read_csv(self.path + "train_with_time.txt", header=None, sep=r"\s+", engine="c",
                         names=["uid", "iid", "time", "stars"], usecols=[0, 1, 2])
        df = df.astype(np.int64)
        self.unique_times = list(pd.unique(df["time"]))
        print("time slot unique in train:", np.asarray(self.unique_times))
        if len(self.unique_times) < 2:
            raise RuntimeWarning("there only one time slot for train...., this may cause our method not work")
        # groupby(...).agg(list) keeps file order inside each user (MF/load_data.py:637-639)
        uid = df["uid"].to_numpy()
        order = np.argsort(uid, kind="stable")
        uid_s, iid_s, t_s = uid[order], df["iid"].to_numpy()[order], df["time"].to_numpy()[order]
        bounds = np.flatnonzero(np.diff(uid_s)) + 1
        starts = np.concatenate([[0], bounds])
        ends = np.concatenate([bounds, [len(uid_s)]])
        self.train_user_list = {int(uid_s[s]): iid_s[s:e].tolist() for s, e in zip(starts, ends)}
        self.train_user_list_time = {int(uid_s[s]): t_s[s:e].tolist() for s, e in zip(starts, ends)}
        for u, it in zip(uid, df["iid"].to_numpy()):
            self.train_item_list[int(it)].append(int(u))
        self.n_users = max(self.n_users, int(uid.max()))
        self.n_items = max(self.n_items, int(df["iid"].max()))
        self.n_train = int(df.shape[0])
        self._load_eval_files()
        print(self.n_train, self.n_valid, self.n_test)


def load_popularity(args):
    """item_pop_seq_ori2.txt (fallback item_pop_seq_ori.txt) -> float64 [n_lines, T], rows in FILE order
    (the reference ignores the leading item id, MF/train_new_api.py:862-880)."""
    r_path = os.path.join(args.data_path, args.dataset) + "/"
    path = r_path + "item_pop_seq_ori2.txt"
    if not os.path.exists(path):
        path = r_path + "item_pop_seq_ori.txt"
    print("popularity used:", path)
    rows = []
    with open(path) as f:
        for line in f:
            parts = line.split()
            if parts:
                rows.append([float(x) for x in parts[1:]])
    pop = np.array(rows)
    print("pop_item_all shape:", pop.shape)
    print("load pop information:", pop.mean(), pop.max(), pop.min())
    return pop


def get_popularity_from_load(item_pop_all):
    """Drop the test-stage slot (MF/train_new_api.py:895-906)."""
    return item_pop_all[:, :-1]
