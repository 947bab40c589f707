"""Trainer / scorer with the reference's interface (MF/train_new_api.py), running the hot path on MI355X.

    DatasetApi_Model   :538-696   wrapper that owns the model + the recommendation heads
    evaluation         :700-828   full-catalogue evaluation driver
    early_stop         :911-927
    main               :930-1338  epoch loop, evaluation cadence, checkpoints, log lines

What changed underneath: `do_recommendation` is ONE fused kernel (score + (elu+1)*pop + history mask + top-K,
pda_score_topk_f32) instead of MatMul[Bu,I] + ~8 passes + TopKV2; `evaluation.eval` scores all evaluation
users in blocks of --eval_block and reduces the metrics on the device (pda_metrics) instead of a
multiprocessing.Pool(5); a training step is the fused pda_bpr_step_f32 (+ Adam sweeps).  `sess` arguments are
accepted everywhere and may be None; `Session.run` understands the reference's fetch lists.

Run:  python -m pda_amd.train_new_api --dataset douban --train s_condition --test s_condition ... (README.md:69)
"""
from __future__ import annotations

import logging
import os
import random
import sys
from time import time

import zlib

import numpy as np
import torch

from . import ops
from .load_data import Data, Data2, get_popularity_from_load, load_popularity
from .model_api import BPRMF, ConditionalBPRMF, Fetch
from .parse import parse_args
from .sampler import DeviceSampler, host_generator, to_device_batch

# module-level singletons of the reference (MF/batch_test.py:6-19), filled by configure()/main()
args = None
data = None
Ks = [20]
ITEM_NUM = 0
USER_NUM = 0


class OutOfRangeError(Exception):
    """End of the epoch's batch stream (tf.errors.OutOfRangeError, MF/train_new_api.py:1097)."""


class Session:
    """Minimal `tf.Session` stand-in: run([opt, loss, mf_loss, reg_loss]) performs one training step on the next
    batch of the wrapper's iterator and returns [None, loss, mf_loss, reg_loss] as python floats (one D->H sync
    per step, like the reference).  `run_async` does the same without the sync and returns the device tensor."""

    def __init__(self, model=None):
        self.model = model

    def _step(self, fetches):
        owners = {f.owner for f in fetches if isinstance(f, Fetch)}
        if len(owners) != 1:
            raise NotImplementedError("Session.run understands the trainer's fetch lists only")
        rec = owners.pop()
        if not any(f.name == "opt" for f in fetches):
            raise NotImplementedError("fetching %s without the optimiser op" % [f.name for f in fetches])
        b = self.model.next_batch()
        return rec.train_step(*b, plan=self.model.batch_plan)

    def run_async(self, fetches):
        return self._step(fetches)

    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, str) and fetches in ("training_op", "global_variables_initializer"):
            return None
        if isinstance(fetches, Fetch):
            fetches = [fetches]
        loss = self._step(fetches).tolist()
        pick = {"opt": None, "loss": loss[0], "mf_loss": loss[1], "reg_loss": loss[2]}
        return [pick[f.name] for f in fetches]


class DatasetApi_Model:
    def __init__(self, args, data_config, test_batch, generator_sampler, device=None, topk_shard=None):
        self.args = args
        self.device = torch.device(device if device is not None else "cuda")
        self.generator_sampler = generator_sampler
        self._iter = None
        self.batch_plan = None        # pda_triplet_plan of the batch next_batch() returned last (--optimizer sgd, device sampler)
        if getattr(args, "optimizer", "adam") == "sgd" and getattr(generator_sampler, "distinct_users", False) and hasattr(generator_sampler, "with_plan"):
            generator_sampler.with_plan = True       # the exact SGD step without atomics wants the batch's plan (ops.bpr_step_plan)
        self.sess = None
        self.testing_model_type, self.testing_popularity = "o", None
        if args.train in ("s_condition", "condition"):
            self.input_type = "with_pop"                                     # :547-549
            print("dataset api with pop or temp")
            self.Recommender = ConditionalBPRMF(args, data_config, use_dataset_api=True, device=self.device)
        elif args.train == "normal":
            self.input_type = "without_pop"                                  # :558-559
            print("dataset api without pop")
            self.Recommender = BPRMF(args, data_config, use_dataset_api=True, device=self.device)
        else:
            raise NotImplementedError("not implement this model: " + args.train)   # :590
        # the device sampler draws a batch's users without replacement (like the reference, :380-381): the fused SGD step may
        # then store its user rows plainly (PDA_UPD_USERS_DISTINCT)
        self.Recommender.users_distinct = bool(getattr(generator_sampler, "distinct_users", False))
        self.n_items = data_config["n_items"]
        self._shard = topk_shard            # optional pda_amd.dist.ItemShardedTopK (multi-GPU evaluation)
        self.Create_Recommendation()

    def Create_Recommendation(self, topk_max=50):
        self.topk_max = topk_max            # "top-K hard-capped at 50" (:594)

    # ---- training side ------------------------------------------------------------------------------
    def switch_to_training_or_reinitsampler(self, sess=None):
        """(Re)start the epoch's generator -- `sess.run(self.training_op)` in the reference (:671-672)."""
        self._iter = iter(self.generator_sampler())

    def next_batch(self):
        if self._iter is None:
            raise OutOfRangeError()
        try:
            b = next(self._iter)
        except StopIteration:
            self._iter = None
            raise OutOfRangeError() from None
        if not torch.is_tensor(b[0]):
            b = to_device_batch(b, self.device)
        self.batch_plan = getattr(self.generator_sampler, "plan", None)
        return b

    # ---- scoring side ---------------------------------------------------------------------------------
    def _tables(self, items):
        I = self.Recommender.score_tables()[1]
        if items is None or (len(items) == I.shape[0] and (len(items) == 0 or (items[0] == 0 and items[-1] == I.shape[0] - 1))):
            return I, None
        sel = torch.as_tensor(np.asarray(items, dtype=np.int64), device=self.device)
        return I.index_select(0, sel).contiguous(), sel

    def _pop_on_device(self, pos_pop):
        """The reference hands every block a FRESH ndarray (`testing_popularity[batch_item]`, MF/train_new_api.py:788) with the same
        contents: the device copy is kept while the contents stay equal, because everything ops caches per popularity vector (the
        >= 0 check, the visiting order, the item image) is keyed on the tensor object."""
        arr = np.ascontiguousarray(np.asarray(pos_pop, dtype=np.float32).reshape(-1))
        hit = getattr(self, "_pop_cache", None)
        # (bit patterns: NaN-safe, and 20 x cheaper than array_equal(equal_nan=True) -- 0.9 ms per 200 000 items, more than the block's sweep)
        if hit is not None and hit[0].shape == arr.shape and np.array_equal(hit[0].view(np.uint32), arr.view(np.uint32)):
            return hit[1]
        t = torch.as_tensor(arr, device=self.device)
        self._pop_cache = (arr.copy(), t)
        return t

    def _mask_on_device(self, index, n_rows):
        """The reference builds a block's mask triple ONCE (`set_evaluate_obj_pre`, MF/train_new_api.py:730-739) and hands the same
        ndarray over in every evaluation epoch (:791): its CSR stays on the device, keyed on the array object and guarded by its
        shape and a checksum of its bytes (a caller that refills the array in place gets a fresh conversion).  Block-row CSRs are small
        (~0.4 MB per 2 048-user block of config 3); the cache holds the blocks of one pass."""
        cache = self.__dict__.setdefault("_mask_cache", {})
        arr = index if isinstance(index, np.ndarray) else None
        probe = None
        if arr is not None and arr.ndim == 2 and arr.shape[0] > 0:
            # the guard covers the WHOLE array (an in-place refill of any row gives a fresh conversion): one pass over ~1.6 MB per 2 048-user
            # block of config 3, a few per cent of the host -> device copy it saves
            probe = (arr.shape, n_rows, str(arr.dtype), zlib.adler32(np.ascontiguousarray(arr).view(np.uint8)))
            hit = cache.get(id(arr))
            if hit is not None and hit[0]() is arr and hit[1] == probe:
                return hit[2]
        hist = ops.HistoryCSR.from_coo(index, n_rows, self.device)
        if probe is not None:
            import weakref
            key = id(arr)
            try:
                # (an entry dies with its array: a caller that builds fresh arrays per call pins nothing)
                cache[key] = (weakref.ref(arr, lambda _r, k=key, c=cache: c.pop(k, None)), probe, hist)
            except TypeError:                                   # (an ndarray subclass without weak references)
                pass
        return hist

    def do_recommendation(self, sess, batch_users, items, rec_type, pos_pop=None, sparse_cliked_matrix=None):
        """-> int32 ndarray [len(batch_users), 50] of positions inside `items` (:614-640)."""
        idx, _ = self.recommend_device(batch_users, items, rec_type, pos_pop, sparse_cliked_matrix)
        return idx.cpu().numpy()

    def recommend_device(self, batch_users, items, rec_type, pos_pop=None, mask=None, K=None):
        if rec_type == "main_branch":
            head, pos_pop = ops.HEAD_RAW, None
        elif rec_type in ("main_with_pop", "condition"):
            if rec_type == "condition" and self.input_type != "with_pop":
                raise NotImplementedError("condition head needs a PD/PDA model")
            head = ops.HEAD_POP
            if pos_pop is None:
                raise ValueError("rec_type %s needs pos_pop" % rec_type)
        else:
            raise NotImplementedError("we have only implement recommendation method: main main+pop condition")   # :639
        users = batch_users if torch.is_tensor(batch_users) else torch.as_tensor(np.asarray(batch_users, dtype=np.int32), device=self.device)
        hist = mask
        if mask is not None and not isinstance(mask, ops.HistoryCSR):
            index, _vals, shape = mask                         # the reference's (index, [-inf]*nnz, shape) triple (:791)
            hist = self._mask_on_device(index, int(shape[0]))
        pop_t = None
        if pos_pop is not None:
            pop_t = pos_pop if torch.is_tensor(pos_pop) else self._pop_on_device(pos_pop)
        K = K or self.topk_max
        I, _sel = self._tables(items)
        if self._shard is not None and _sel is None:
            self._shard.set_popularity(pop_t)
            return self._shard.topk(users, K, head, hist)
        return ops.recommend_topk(self.Recommender.score_tables()[0], I, users, K, head, pop_t, hist)

    def testing(self, sess, batch_users, items, model_type, pos_pop=None):
        """Dense scores f32 [B, len(items)] (:642-669): batch_ratings / condition_ratings as the NeuRec evaluators' protocol fetches them (the
        reference imports that protocol and never calls it).  pda_score_dense_f32: every entry is the exact fmaf chain of the top-K kernels, so
        these scores equal the values do_recommendation ranks by, bit for bit (bf16 tables: widened first, as their scores are defined)."""
        if model_type not in ("main_branch", "condition"):
            raise NotImplementedError("error -- not implement this type testing method...")      # :664
        U, I = self.Recommender.score_tables()
        users = torch.as_tensor(np.asarray(batch_users, dtype=np.int32), device=self.device)
        it = np.asarray(items, dtype=np.int64).reshape(-1)
        whole = it.size == I.shape[0] and bool((it == np.arange(it.size)).all())
        it_t = None if whole else torch.as_tensor(it.astype(np.int32), device=self.device)
        if U.dtype != torch.float32:
            U, I = U.float(), I.float()
        pop = None
        if model_type == "condition":
            pop = torch.as_tensor(np.asarray(pos_pop, dtype=np.float32).reshape(-1), device=self.device)
        return ops.score_dense(U, I, users, ops.HEAD_POP if model_type == "condition" else ops.HEAD_RAW, pop, it_t).cpu().numpy()

    def switch_to_testing_or_reinit(self, sess=None, feed_dict=None):
        return None

    def set_testing_way(self, model_type, popularity_exp):
        self.testing_model_type, self.testing_popularity = model_type, popularity_exp

    def set_sess(self, sess):
        self.sess = sess

    def predict(self, user_batch, item_batch):                                              # :683-696
        if item_batch is None:
            item_batch = list(range(self.n_items))
        if self.testing_model_type == "o":
            return self.testing(self.sess, user_batch, item_batch, "main_branch")
        if self.testing_model_type == "condition":
            return self.testing(self.sess, user_batch, item_batch, "condition", pos_pop=self.testing_popularity[item_batch])
        raise NotImplementedError("not implement this type testing methods")


class evaluation:
    def __init__(self, data_=None, Ks_=None, device=None, block=None):
        self.data = data_ if data_ is not None else data
        self.Ks = list(Ks_ if Ks_ is not None else Ks)
        self.device = torch.device(device if device is not None else "cuda")
        self.batch_size = block or (getattr(args, "eval_block", 262144) if args is not None else 262144)
        self.testing_popularity = None
        self.eval_who = "test"
        self._hist = None

    def set_evaluate_obj(self, eval_who="test"):
        self.eval_who = eval_who

    def set_testing_popularity(self, popularity):
        """MF/train_new_api.py:710.  The reference indexes testing_popularity[range(ITEM_NUM)] (:788): a vector longer than the
        catalogue is cut to n_items here, a shorter one is an error (there: IndexError at the first evaluation)."""
        self.testing_popularity = popularity
        if popularity is None:
            self._pop_dev = None
            return
        pop = np.asarray(popularity, dtype=np.float32).reshape(-1)
        n = self.data.n_items
        if pop.shape[0] < n:
            raise IndexError("testing popularity has %d entries for %d items" % (pop.shape[0], n))
        self._pop_dev = torch.as_tensor(np.ascontiguousarray(pop[:n]), device=self.device)

    def set_evaluate_obj_pre(self, eval_who="test"):
        """Evaluation users (file order), their targets as CSR, and the train-history mask (:713-739).
        Raises KeyError for an evaluation user without train items under Data2, like the reference."""
        self.eval_who = eval_who
        d = self.data
        self.eval_user_list = d.test_user_list if eval_who == "test" else d.valid_user_list
        users = list(self.eval_user_list.keys())
        self.tot_user = len(users)
        for u in users:
            d.train_user_list[u]                      # KeyError <=> reference (plain dict under Data2, :731)
        if self._hist is None:
            ip, ix, _ = d.train_csr(self.device)
            self._hist = ops.HistoryCSR(ip, ix, by_user=True)
        self.users_dev = torch.as_tensor(np.asarray(users, dtype=np.int32), device=self.device)
        lens = np.fromiter((len(self.eval_user_list[u]) for u in users), dtype=np.int64, count=len(users))
        tp = np.zeros(len(users) + 1, dtype=np.int64)
        np.cumsum(lens, out=tp[1:])
        flat = np.concatenate([np.asarray(self.eval_user_list[u], dtype=np.int32) for u in users]) if users else np.zeros(0, np.int32)
        self._tp_host = tp
        self.tgt_indices = torch.from_numpy(flat).to(self.device)
        self.list_batch_user = [users[i:i + self.batch_size] for i in range(0, len(users), self.batch_size)]

    def eval(self, model, sess, rec_type):
        """-> {'precision','recall','ndcg','hit_ratio': float64[len(Ks)]} = per-user sums / tot_user (:760-778)."""
        pop = None if self.testing_popularity is None else self._pop_dev
        ks = torch.as_tensor(self.Ks, dtype=torch.int32, device=self.device)
        sums = torch.zeros((4, len(self.Ks)), dtype=torch.float64, device=self.device)
        for i in range(0, self.tot_user, self.batch_size):
            ub = self.users_dev[i:i + self.batch_size]
            idx, _ = model.recommend_device(ub, None, rec_type, pop, self._hist)
            tp = self._tp_host[i:i + ub.numel() + 1]
            tptr = torch.from_numpy(tp - tp[0]).to(self.device)
            ops.metrics_sums(idx, tptr, self.tgt_indices[int(tp[0]):int(tp[-1])], ks, sums)
        s = (sums / float(self.tot_user)).cpu().numpy()
        return {"precision": s[0], "recall": s[1], "ndcg": s[2], "hit_ratio": s[3]}


def early_stop(hr, ndcg, recall, precision, cur_epoch, config, stopping_step, flag_step=10):   # :911-927
    if recall >= config["best_recall"]:
        stopping_step = 0
        config.update(best_hr=hr, best_ndcg=ndcg, best_recall=recall, best_pre=precision, best_epoch=cur_epoch)
    else:
        stopping_step += 1
    should_stop = stopping_step >= flag_step
    if should_stop:
        print("Early stopping is trigger")
    return config, stopping_step, should_stop


def configure(argv=None):
    """What `from batch_test import *` does at import time in the reference (MF/batch_test.py:6-19)."""
    global args, data, Ks, ITEM_NUM, USER_NUM
    args = parse_args(argv)
    if args.train in ("s_condition", "sg_condition", "temp_pop", "us_condition"):
        data = Data2(args)
    else:
        data = Data(args)
    Ks = eval(args.Ks)            # the reference evals this literal too (batch_test.py:16)
    ITEM_NUM, USER_NUM = data.n_items, data.n_users
    return args, data


def _print_result(ret):   # :1118-1123
    print("||---------------------------------------------- recall=[%.5f, %.5f], precision=[%.5f, %.5f], hit=[%.5f, %.5f], ndcg=[%.5f, %.5f]"
          % (ret["recall"][0], ret["recall"][-1], ret["precision"][0], ret["precision"][-1],
             ret["hit_ratio"][0], ret["hit_ratio"][-1], ret["ndcg"][0], ret["ndcg"][-1]))


def main(argv=None):
    print("*** Current working path ***")
    print(os.getcwd())
    configure(argv)
    random.seed(2020)                      # :934-936
    np.random.seed(2020)
    torch.manual_seed(2021)
    if torch.cuda.device_count() > 1 and str(args.cuda).isdigit() and int(args.cuda) < torch.cuda.device_count():
        torch.cuda.set_device(int(args.cuda))
    device = torch.device("cuda")
    config = {"n_users": data.n_users, "n_items": data.n_items}
    popularity_exp = args.pop_exp
    print("----- popularity_exp : ", popularity_exp)
    test_batch_size = min(1024, args.batch_size)

    pop_item_all = load_popularity(args)                                         # :952-959
    last_stage_popualarity = np.power(pop_item_all[:, -2], popularity_exp)
    linear_predict_popularity = pop_item_all[:, -2] + 0.5 * (pop_item_all[:, -2] - pop_item_all[:, -3])
    linear_predict_popularity[np.where(linear_predict_popularity <= 0)] = 1e-9
    linear_predict_popularity[np.where(linear_predict_popularity > 1.0)] = 1.0
    linear_predict_popularity = np.power(linear_predict_popularity, popularity_exp)

    with_pop = False
    if args.model == "mf" and args.train == "normal":                            # :963-970
        args.saveID += "pop_exp-{:.2f}".format(popularity_exp)
        print("normal MF... ")
        last_stage_popualarity_ori = pop_item_all[:, -2]
        linear_predict_popularity_ori = pop_item_all[:, -2] + 0.5 * (pop_item_all[:, -2] - pop_item_all[:, -3])
        # the reference masks with the already-powered array (a quirk): kept
        linear_predict_popularity_ori[np.where(linear_predict_popularity <= 0)] = 1e-9
        linear_predict_popularity_ori[np.where(linear_predict_popularity > 1.0)] = 1.0
    elif args.model == "mf" and args.train == "s_condition":                     # :984-997
        print("-------    running PD & PDA model  ----------------")
        args.saveID += "pop_exp-{:.2f} (gamma)".format(popularity_exp)
        print("save_ID", args.saveID)
        popularity_matrix = get_popularity_from_load(pop_item_all)
        popularity_matrix = np.power(popularity_matrix, popularity_exp)
        print("------ popularity information after powed  ------")
        print("   each stage mean:", popularity_matrix.mean(axis=0))
        print("   each stage max:", popularity_matrix.max(axis=0))
        print("   each stage min:", popularity_matrix.min(axis=0))
        data.add_expo_popularity(popularity_matrix)
        with_pop = True
    else:
        raise NotImplementedError("do not implement this method")               # :1008

    if args.sampler == "device":
        sampler = DeviceSampler(data, device, with_pop)
    else:
        sampler = (lambda: host_generator(data, with_pop))
    model = DatasetApi_Model(args, config, test_batch_size, sampler, device)
    sess = Session(model)
    model.set_sess(sess)
    args.wd = args.regs                                                          # :1020

    evaluation_model = evaluation(data, Ks, device)
    if args.valid_set == "test":
        evaluation_model.set_evaluate_obj_pre("test")
        print("valid in test set")
    elif args.valid_set == "valid":
        print("valid in valid set")
        evaluation_model.set_evaluate_obj_pre("valid")
    else:
        print("evaluate type error.")
        sys.exit()
    print("args info:", args)
    print("top K:", Ks)
    if args.pretrain != 0:
        raise NotImplementedError("only --pretrain 0 exists in the reference (MF/train_new_api.py:1048)")

    rec = model.Recommender
    best_pop_expo_normal = 0
    keys = ("best_hr", "best_ndcg", "best_recall", "best_pre", "best_epoch")
    config.update({k: 0 for k in keys})
    config_main = dict(config)
    stopping_step = stopping_step_main = 0
    n_batch = data.n_train // args.batch_size + 1
    save_ckpt_dir = args.save_dir + "{}_{}_checkpoint/wd_{}_lr_{}_a_{}_{}_train_{}/".format(
        args.model, args.dataset, args.wd, args.lr, args.alpha, args.saveID, args.train)   # :1214
    t1 = time()
    print("batch_num:", n_batch, "waiting sampling...")
    fetches = ([rec.opt_pop_global, rec.loss_pop_global, rec.mf_loss_pop_global, rec.reg_loss_pop_global]
               if args.train == "s_condition" else [rec.opt, rec.loss, rec.mf_loss, rec.reg_loss])

    def save(name):
        os.makedirs(save_ckpt_dir, exist_ok=True)
        torch.save(rec.state_dict(), save_ckpt_dir + name)

    for epoch in range(args.epoch):
        model.switch_to_training_or_reinitsampler(sess)
        rec.start_loss_rows(n_batch + 1)                # losses stay on the device, a row per step: one reduction and one sync per epoch
        extra = torch.zeros(3, dtype=torch.float64, device=device)
        try:
            while True:
                before = rec._loss_row_i
                row = sess.run_async(fetches)
                if rec._loss_row_i == before:           # (a generator longer than n_batch + 1 steps: the step fell back to the ring's rows)
                    extra += row
        except OutOfRangeError:
            pass
        acc = rec.finish_loss_rows() + extra
        loss, mf_loss, reg_loss = (acc / n_batch).tolist()
        if np.isnan(loss):
            print("ERROR: loss is nan.")
            sys.exit()
        perf_str = "Epoch %d [%.1fs]: train==[%.5f=%.5f + %.5f]" % (epoch, time() - t1, loss, mf_loss, reg_loss)
        if epoch % args.log_interval != 0:
            if args.verbose > 0 and epoch % args.verbose == 0:
                print(perf_str)
            t1 = time()
            continue

        if args.test in ("condition", "s_condition"):                            # :1125-1156
            print("do not consider popularity (PD or PDG) ... ")
            print(perf_str)
            evaluation_model.set_testing_popularity(None)
            ret_main = evaluation_model.eval(model, sess, rec_type="main_branch")
            _print_result(ret_main)
            print("injecting last stage popularity.... ")
            ttt1 = time()
            evaluation_model.set_testing_popularity(last_stage_popualarity)
            ret1 = evaluation_model.eval(model, sess, rec_type="condition")
            print("||------------PDA/PDGA injecting last stage popularity testing : time: ", int(time() - ttt1))
            _print_result(ret1)
            ttt1 = time()
            evaluation_model.set_testing_popularity(linear_predict_popularity)
            ret2 = evaluation_model.eval(model, sess, rec_type="condition")
            print("||------------PDA/PDGA injecting linear predicted popularity testing : time: ", int(time() - ttt1))
            _print_result(ret2)
            ret = ret1
        elif args.test == "normal":                                              # :1159-1191
            print(perf_str)
            ttt1 = time()
            evaluation_model.set_testing_popularity(None)
            ret_main = evaluation_model.eval(model, sess, rec_type="main_branch")
            print("test: time:", time() - ttt1)
            _print_result(ret_main)
            best_ret, best_expo, not_incre, expo = ret_main, 0, 0, 0.04
            while True:                                                          # gamma-tilde line search for BPRMF-A
                evaluation_model.set_testing_popularity(np.power(last_stage_popualarity_ori, expo))
                ret_k = evaluation_model.eval(model, sess, rec_type="main_with_pop")
                if ret_k["recall"][0] < best_ret["recall"][0]:
                    not_incre += 1
                    if not_incre > 4:
                        break
                else:
                    not_incre, best_ret, best_expo = 0, ret_k, expo
                print("expo: {:.2f} best expo:{:.2f}".format(expo, best_expo))
                _print_result(ret_k)
                expo += 0.02
            if best_ret["recall"][0] >= config["best_recall"]:
                best_pop_expo_normal = best_expo
            ret = best_ret
        else:
            raise NotImplementedError("not implement this test method:" + args.test)   # :1202

        stop_flag_step = 100 // args.log_interval                                # :1210-1232
        config, stopping_step, should_stop = early_stop(ret["hit_ratio"][0], ret["ndcg"][0], ret["recall"][0],
                                                        ret["precision"][0], epoch, config, stopping_step, stop_flag_step)
        config_main, stopping_step_main, should_stop_main = early_stop(
            ret_main["hit_ratio"][0], ret_main["ndcg"][0], ret_main["recall"][0], ret_main["precision"][0], epoch,
            config_main, stopping_step_main, stop_flag_step)
        if epoch == config["best_epoch"]:
            save("best_ckpt.ckpt")
        if epoch == config_main["best_epoch"]:
            save("best_main_ckpt.ckpt")
        if args.save_flag == 1 and (epoch + 1) % 50 == 0:
            save("{}_ckpt.ckpt".format(epoch))
        if should_stop and args.early_stop == 1 and should_stop_main:
            msg = "{} dataset best epoch{}: hr:{} ndcg:{} recall:{} precision:{}".format(
                args.dataset, config["best_epoch"], config["best_hr"], config["best_ndcg"], config["best_recall"], config["best_pre"])
            print(msg)
            print("{} dataset best main epoch{}: hr:{} ndcg:{} recall:{} precision:{}".format(
                args.dataset, config_main["best_epoch"], config_main["best_hr"], config_main["best_ndcg"],
                config_main["best_recall"], config_main["best_pre"]))
            logging.info(msg)
            if args.save_flag == 1:
                os.makedirs(save_ckpt_dir, exist_ok=True)
                with open(save_ckpt_dir + "/best_epoch.txt", "w") as f:
                    print(config["best_epoch"], file=f)
            break
        t1 = time()

    # ---- final report on the best checkpoints (:1253-1327) ----------------------------------------------
    print("best epoch", config["best_epoch"])
    rec.load_state_dict(torch.load(save_ckpt_dir + "best_ckpt.ckpt", map_location=device))
    print("validation result in best epoch")
    evaluation_model.set_testing_popularity(None)
    ret = evaluation_model.eval(model, sess, rec_type="main_branch")
    print("---- result without pop:")
    _print_result(ret)
    print("|||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||")
    print("|| ---------------- testing testset in the best epoch:  ... ")
    evaluation_model.set_evaluate_obj_pre("test")
    if args.test == "s_condition":
        evaluation_model.set_testing_popularity(None)
        ret = evaluation_model.eval(model, sess, rec_type="main_branch")
        print("---- PD/PDG result without pop at the model select by PDA/PDG-A:")
        _print_result(ret)
        evaluation_model.set_testing_popularity(last_stage_popualarity)
        ret = evaluation_model.eval(model, sess, rec_type="condition")
        print("---- PDA/PDG-A injecting last stage pop:\n", ret)
        _print_result(ret)
        evaluation_model.set_testing_popularity(linear_predict_popularity)
        ret = evaluation_model.eval(model, sess, rec_type="condition")
        print("---- result with linear pop:\n", ret)
        _print_result(ret)
    elif args.test == "normal":
        evaluation_model.set_testing_popularity(None)
        ret = evaluation_model.eval(model, sess, rec_type="main_branch")
        print("---- BPRMF result without injecting pop:")
        _print_result(ret)
        print("best_pop_expo in training:", best_pop_expo_normal)
        for name, base in (("last stage pop(best gamma)", last_stage_popualarity_ori),
                           ("linear predicted pop (best gamma)", linear_predict_popularity_ori)):
            evaluation_model.set_testing_popularity(np.power(base, best_pop_expo_normal))
            r = evaluation_model.eval(model, sess, rec_type="main_with_pop")
            print("|||---BPRMF-A with injecting %s:" % name)
            _print_result(r)
        print("----------------------------")
    print("training and testing end!!!!")
    print("|||  ------------------------ best performance for model selected by PD/PDG/BPRMF ------------------- |||")
    print("main best epoch:", config_main["best_epoch"])
    rec.load_state_dict(torch.load(save_ckpt_dir + "best_main_ckpt.ckpt", map_location=device))
    evaluation_model.set_testing_popularity(None)
    ret = evaluation_model.eval(model, sess, rec_type="main_branch")
    print("---- result without injecting pop:")
    _print_result(ret)
    return config, config_main


if __name__ == "__main__":
    main()
