"""Model objects with the reference's names and constructor arguments (MF/model_api.py), backed by HIP kernels.

    ConditionalBPRMF   PD / PDA            MF/model_api.py:14-185
    BPRMF              plain BPR-MF        MF/model_api.py:419-757   (only :419-471, :521-536, :695-706 are live)

A TF-1 graph exposes *fetchables* (`opt`, `loss`, `mf_loss`, `reg_loss`, `batch_ratings`, ...) that the
trainer passes to `sess.run`.  Here they are light handle objects understood by `pda_amd.train_new_api.Session`,
so the reference's loop `sess.run([model.Recommender.opt, model.Recommender.loss, ...])` keeps its shape;
the direct API is `train_step(users, pos, neg[, pos_pop, neg_pop]) -> float32[3] device tensor`.

Optimisers (`args.optimizer`):
    adam       TF-1.14 AdamOptimizer semantics: m, v decayed and EVERY row updated each step [TF-ext]
               (= pda_bpr_step_f32(DENSE_GRAD) + pda_adam_dense_sweep_f32 on both tables).  Reference-faithful.
    lazy_adam  the same update restricted to the rows touched by the batch (declared deviation).
    sgd        plain mini-batch SGD, exact: gradients of the whole batch against the unchanged tables, then one scatter
               (pda_bpr_step_f32(PDA_UPD_NONE) + pda_sgd_apply_f32; declared deviation from MF/model_api.py:83 = Adam).
    sgd_fused  the north_star's fused in-kernel scatter update, ONE launch per step: asynchronous inside the launch (a row
               gathered by one workgroup may already carry another triplet's update -- hogwild-style, equal to `sgd` up to
               O(lr) cross terms, not bit-reproducible).  The throughput mode; `sgd` is the reference semantics.

Table type (`args.table_dtype`, extension; BASELINE config 5): with "bf16" the forward pass and the evaluation read bf16
copies of the tables (`score_tables()`), gradients stay fp32 and `weights[...]` are the fp32 masters that take the
update; the touched rows (sgd / lazy_adam) or the whole tables (adam) are re-rounded after every step.
"""
from __future__ import annotations

import math

import torch

from . import ops


class Fetch:
    """Stand-in for a TF tensor/op handle: `Session.run` dispatches on (owner, name)."""

    def __init__(self, owner, name):
        self.owner, self.name = owner, name

    def __repr__(self):
        return "<pda_amd fetch %s.%s>" % (type(self.owner).__name__, self.name)


def xavier_uniform_(t: torch.Tensor, gen: torch.Generator):
    """tf.contrib.layers.xavier_initializer(): U(-l, l), l = sqrt(6/(fan_in+fan_out)) [TF-ext]; :88-92, :523-527."""
    lim = math.sqrt(6.0 / (t.shape[0] + t.shape[1]))
    return t.uniform_(-lim, lim, generator=gen)


class _MFBase:
    with_pop = False
    ADAM_SWEEP_MAX_BYTES = 64 << 20      # C1/C2 (12 .. 18 MB of tables) sweep; config 3 (614 MB) and up replay

    def __init__(self, args, data_config, use_dataset_api=False, users_api=None, pos_items_api=None,
                 neg_items_api=None, pos_pop_api=None, neg_pop_api=None, device=None, seed=2021):
        self.n_users = data_config["n_users"]
        self.n_items = data_config["n_items"]
        self.decay = args.regs                       # MF/model_api.py:23
        self.emb_dim = args.embed_size
        self.lr = args.lr
        self.batch_size = args.batch_size            # the flag constant that divides the regulariser (:118)
        self.verbose = args.verbose
        self.optimizer = getattr(args, "optimizer", "adam")
        if self.optimizer not in ("adam", "lazy_adam", "sgd", "sgd_fused"):
            raise NotImplementedError("optimizer must be adam | lazy_adam | sgd | sgd_fused")
        self.table_dtype = getattr(args, "table_dtype", "f32")
        if self.table_dtype not in ("f32", "bf16"):
            raise NotImplementedError("table_dtype must be f32 | bf16")
        self.device = torch.device(device if device is not None else "cuda")
        gen = torch.Generator(device=self.device)
        gen.manual_seed(seed)                        # tf.set_random_seed(2021), MF/train_new_api.py:936
        self.weights = self.init_weights(gen)
        self.tables16 = None
        if self.table_dtype == "bf16":
            self.tables16 = {k: v.bfloat16() for k, v in self.weights.items()}
        self._t = 0
        self._state = None
        # optimizer == "adam": the reference's dense-decay Adam either as a sweep over both tables per step (small tables) or
        # EXACTLY the same arithmetic without the sweep (ops.adam_lazy: idle rows replay their decay when next needed) once the
        # tables outgrow ADAM_SWEEP_MAX_BYTES; args.adam_exact_lazy = True | False forces one
        forced = getattr(args, "adam_exact_lazy", None)
        mode = getattr(args, "adam_sweep", "auto")
        if mode not in ("auto", "sweep", "replay", "replay_fast"):
            raise NotImplementedError("adam_sweep must be auto | sweep | replay | replay_fast")
        if forced is None and mode != "auto":
            forced = mode != "sweep"
        # replay (and auto, above 64 MB of tables): bit-identical to the sweeps -- the reference-parity optimiser and the checkpoints
        # written behind sync_optimizer stay bit-equal to the dense-decay Adam.  replay_fast is an explicit opt-in: the same catch-up
        # with a running sqrt, the hardware reciprocal and closed-form powers for m and v (PDA_ADAM_REPLAY_FAST, ~4 x less
        # arithmetic); x to 1e-6 per catch-up against the sweep (tests/test_gpu_bpr_step.py), NOT bit-equal, and the per-catch-up
        # error (~4e-8 |x|: the geometric tail is ~10 x the last term) accumulates over the row's touches of a long run
        self.adam_replay_fast = mode == "replay_fast"
        table_bytes = (self.n_users + self.n_items) * self.emb_dim * 4
        self.adam_exact_lazy = (table_bytes > self.ADAM_SWEEP_MAX_BYTES) if forced is None else bool(forced)
        self._lazy = None
        # loss buffers: a ring of 16 -- train_step returns the buffer of THIS step; it is overwritten 16 steps later, so a
        # caller may keep a handful of returned tensors and sync once (no per-step clone launch, no per-step host sync)
        self._loss_ring = torch.zeros((16, 3), dtype=torch.float32, device=self.device)
        self._loss_i = 0
        self._loss = self._loss_ring[0]
        self._statistics_params()

    # ---- parameters -----------------------------------------------------------------------------------
    def init_weights(self, gen):
        w = {}
        w["user_embedding"] = xavier_uniform_(torch.empty(self.n_users, self.emb_dim, device=self.device), gen)
        w["item_embedding"] = xavier_uniform_(torch.empty(self.n_items, self.emb_dim, device=self.device), gen)
        return w

    def _statistics_params(self):
        total = sum(int(v.numel()) for v in self.weights.values())
        if self.verbose > 0:
            print("#params: %d" % total)

    def _opt_state(self):
        if self._state is None:
            U, I = self.weights["user_embedding"], self.weights["item_embedding"]
            z = torch.zeros_like
            self._state = {"mU": z(U), "vU": z(U), "gU": z(U), "mI": z(I), "vI": z(I), "gI": z(I)}
        return self._state

    def sync_optimizer(self):
        """Exact lazy Adam: bring every row (and its moments) to the current step -- before anything reads whole tables
        (evaluation, checkpoint).  A no-op for the other optimisers and when nothing is behind."""
        if self._lazy is not None and self._lazy.synced < self._t:
            st = self._state
            ops.adam_lazy_sync(self._lazy, self.weights["user_embedding"], st["mU"], st["vU"], self.weights["item_embedding"], st["mI"],
                               st["vI"], self._t)
            if self.tables16 is not None:
                ops.refresh_rows_bf16(self.weights["user_embedding"], self.tables16["user_embedding"])
                ops.refresh_rows_bf16(self.weights["item_embedding"], self.tables16["item_embedding"])

    def _lazy_state(self):
        if self._lazy is None:
            self._lazy = ops.LazyAdamState(self.n_users, self.n_items, self.lr, self.device, fast=self.adam_replay_fast)
            self._lazy.synced = self._t - 1 if self._t > 0 else 0      # (rows are current for everything before this step)
            if self._t > 1:
                self._lazy.lastU.fill_(self._t - 1)
                self._lazy.lastI.fill_(self._t - 1)
        return self._lazy

    def score_tables(self):
        """(U, I) the evaluation kernels read: the weights, or their bf16 copies when table_dtype == 'bf16'."""
        self.sync_optimizer()
        t = self.tables16 if self.tables16 is not None else self.weights
        return t["user_embedding"], t["item_embedding"]

    def _train_step_bf16(self, users, pos, neg, pos_pop, neg_pop):
        U, I = self.weights["user_embedding"], self.weights["item_embedding"]
        U16, I16 = self.tables16["user_embedding"], self.tables16["item_embedding"]
        if self.optimizer in ("sgd", "sgd_fused"):      # (bf16 forward reads the shadow tables: no in-launch race either way)
            ops.bpr_step_bf16(U16, I16, users, pos, neg, pos_pop, neg_pop, regs=self.decay, reg_div=self.batch_size, lr=self.lr,
                              mode=ops.UPD_SGD_FUSED, U_master=U, I_master=I, loss_acc=self._loss)
            return self._loss
        st = self._opt_state()
        self._t += 1
        lr_t = ops.adam_lr_t(self.lr, self._t)
        lazy = self.optimizer == "adam" and self.adam_exact_lazy
        if lazy:
            lz = self._lazy_state()
            ops.adam_lazy(0, lz, U, st["mU"], st["vU"], st["gU"], I, st["mI"], st["vI"], st["gI"], users, pos, neg, self._t)
            for rows in (users,):
                ops.refresh_rows_bf16(U, U16, rows)
            for rows in (pos, neg):
                ops.refresh_rows_bf16(I, I16, rows)
        ops.bpr_step_bf16(U16, I16, users, pos, neg, pos_pop, neg_pop, regs=self.decay, reg_div=self.batch_size,
                          mode=ops.UPD_DENSE_GRAD, gU=st["gU"], gI=st["gI"], loss_acc=self._loss)
        if lazy:
            ops.adam_lazy(1, lz, U, st["mU"], st["vU"], st["gU"], I, st["mI"], st["vI"], st["gI"], users, pos, neg, self._t)
            ops.refresh_rows_bf16(U, U16, users)
            ops.refresh_rows_bf16(I, I16, pos)
            ops.refresh_rows_bf16(I, I16, neg)
        elif self.optimizer == "adam":
            self._dense_sweep(U, I, st, users, pos, neg, lr_t)
            ops.refresh_rows_bf16(U, U16)                      # dense decay moves every row
            ops.refresh_rows_bf16(I, I16)
        else:
            ru, ri = torch.unique(users).int(), torch.unique(torch.cat([pos, neg])).int()
            ops.adam_rows(U, st["mU"], st["vU"], st["gU"], ru, lr_t)
            ops.adam_rows(I, st["mI"], st["vI"], st["gI"], ri, lr_t)
            ops.refresh_rows_bf16(U, U16, ru)
            ops.refresh_rows_bf16(I, I16, ri)
        return self._loss

    # ---- the losses of an announced run of steps (the trainer's epoch, MF/train_new_api.py:1078-1095) -------------
    def start_loss_rows(self, n_steps: int):
        """The next n_steps train_step calls write their (loss, mf, reg) into consecutive rows of one zeroed float32 [n_steps, 3] block (each call
        still returns ITS row); finish_loss_rows() returns the float64 sum of the rows written.  Saves a memset launch and an accumulation
        launch per step: 32.6 -> 28 us per step through the CLI on the Douban-shaped synthetic."""
        self._loss_rows = torch.zeros((max(1, int(n_steps)), 3), dtype=torch.float32, device=self.device)
        self._loss_row_i = 0

    def finish_loss_rows(self) -> torch.Tensor:
        rows, n = self._loss_rows, self._loss_row_i
        self._loss_rows = None
        return rows[:n].double().sum(0)

    # ---- one training step (A1-A5) --------------------------------------------------------------------
    def train_step(self, users, pos, neg, pos_pop=None, neg_pop=None, plan=None) -> torch.Tensor:
        """Forward + loss + gradient + update on one batch of device tensors (int32 / float32).
        plan (--optimizer sgd): the batch's pda_triplet_plan from a sampler that guarantees distinct users -- the exact step then
        runs without atomics (ops.bpr_step_plan: two launches, bit-reproducible); without a plan the exact step is
        pda_bpr_step_f32(PDA_UPD_NONE) + pda_sgd_apply_f32, which accepts any batch.
        Returns the float32[3] device tensor (loss, mf_loss, reg_loss) of THIS step (no host sync): a view into a ring of
        16 buffers -- valid until 16 further steps have been enqueued."""
        U, I = self.weights["user_embedding"], self.weights["item_embedding"]
        if not self.with_pop:
            pos_pop = neg_pop = None
        elif pos_pop is None or neg_pop is None:
            raise ValueError("PD/PDA needs pos_pop and neg_pop")
        rows = getattr(self, "_loss_rows", None)
        if rows is not None and self._loss_row_i < rows.shape[0]:
            # a caller that announced its steps (start_loss_rows: the trainer's epoch) gets a row of ONE pre-zeroed block per step: no memset
            # launch per step, and the epoch's sum is one reduction at its end
            self._loss = rows[self._loss_row_i]
            self._loss_row_i += 1
        else:
            self._loss_i = (self._loss_i + 1) & 15
            self._loss = self._loss_ring[self._loss_i]
            self._loss.zero_()
        if self.optimizer == "sgd" and plan is not None:
            if self.tables16 is not None:
                self._plan_scratch = ops.bpr_step_plan(self.tables16["user_embedding"], self.tables16["item_embedding"], users, pos, neg, pos_pop,
                                                       neg_pop, regs=self.decay, reg_div=self.batch_size, lr=self.lr, plan=plan,
                                                       scratch=getattr(self, "_plan_scratch", None), loss_acc=self._loss, U_master=U, I_master=I)
            else:
                self._plan_scratch = ops.bpr_step_plan(U, I, users, pos, neg, pos_pop, neg_pop, regs=self.decay, reg_div=self.batch_size,
                                                       lr=self.lr, plan=plan, scratch=getattr(self, "_plan_scratch", None), loss_acc=self._loss)
            return self._loss
        if self.tables16 is not None:
            return self._train_step_bf16(users, pos, neg, pos_pop, neg_pop)
        if self.optimizer == "sgd_fused":
            ops.bpr_step(U, I, users, pos, neg, pos_pop, neg_pop, regs=self.decay, reg_div=self.batch_size, lr=self.lr,
                         mode=ops.UPD_SGD_FUSED, loss_acc=self._loss, users_distinct=bool(getattr(self, "users_distinct", False)))
            return self._loss
        if self.optimizer == "sgd":
            sc = getattr(self, "_sgd_scratch", None)
            if sc is not None and sc[0].shape[0] != users.numel():
                sc = None
            self._sgd_scratch = ops.sgd_step_exact(U, I, users, pos, neg, pos_pop, neg_pop, regs=self.decay, reg_div=self.batch_size,
                                                   lr=self.lr, loss_acc=self._loss, scratch=sc)
            return self._loss
        st = self._opt_state()
        self._t += 1
        lr_t = ops.adam_lr_t(self.lr, self._t)
        lazy = self.optimizer == "adam" and self.adam_exact_lazy
        d = U.shape[1]
        if self.optimizer == "adam" and not lazy and d in (32, 64, 128, 256):
            # the reference's step in two launches (round 6): gradients + row tags, then the tagged sweep (cache policy by working set)
            if "tagU" not in st:
                st["tagU"], st["tagI"] = ops.adam_row_tags(U.shape[0], I.shape[0], U.device)
            ops.adam_step(U, st["mU"], st["vU"], st["gU"], st["tagU"], I, st["mI"], st["vI"], st["gI"], st["tagI"], users, pos, neg, pos_pop, neg_pop,
                          regs=self.decay, reg_div=self.batch_size, step=self._t, lr_t=lr_t, loss_acc=self._loss,
                          users_distinct=bool(getattr(self, "users_distinct", False)))
            return self._loss
        if lazy:        # the batch rows up to step t - 1: the forward pass reads them
            ops.adam_lazy(0, self._lazy_state(), U, st["mU"], st["vU"], st["gU"], I, st["mI"], st["vI"], st["gI"], users, pos, neg, self._t)
        ops.bpr_step(U, I, users, pos, neg, pos_pop, neg_pop, regs=self.decay, reg_div=self.batch_size,
                     mode=ops.UPD_DENSE_GRAD, gU=st["gU"], gI=st["gI"], loss_acc=self._loss)
        if lazy:
            ops.adam_lazy(1, self._lazy, U, st["mU"], st["vU"], st["gU"], I, st["mI"], st["vI"], st["gI"], users, pos, neg, self._t)
        elif self.optimizer == "adam":
            self._dense_sweep(U, I, st, users, pos, neg, lr_t)
        else:
            ops.adam_rows(U, st["mU"], st["vU"], st["gU"], torch.unique(users).int(), lr_t)
            ops.adam_rows(I, st["mI"], st["vI"], st["gI"], torch.unique(torch.cat([pos, neg])).int(), lr_t)
        return self._loss

    def _dense_sweep(self, U, I, st, users, pos, neg, lr_t):
        """TF-1.14's dense-decay Adam step on both tables: six streams (the gradient tables are read only on the batch's rows: pda_adam_mark_rows +
        pda_adam_dense_sweep3_f32) where the row length is a power of two, the seven-stream sweep otherwise.  Bit-identical."""
        d = U.shape[1]
        if d >= 4 and (d & (d - 1)) == 0:
            if "tU" not in st:
                st["tU"], st["tI"] = ops.adam_touched_bitmaps(U.shape[0], I.shape[0], U.device)
            ops.adam_mark_rows(users, pos, neg, st["tU"], st["tI"])
            ops.adam_dense_sweep3(U, st["mU"], st["vU"], st["gU"], st["tU"], I, st["mI"], st["vI"], st["gI"], st["tI"], lr_t)
        else:
            ops.adam_dense_sweep2(U, st["mU"], st["vU"], st["gU"], I, st["mI"], st["vI"], st["gI"], lr_t)

    # ---- checkpoint (tf.train.Saver stand-in, MF/train_new_api.py:1014,1218-1228) ---------------------
    CKPT_FORMAT = "pda_amd/2"     # torch.save pickle of this dict -- NOT a tf.train.Saver checkpoint (see README)

    def state_dict(self):
        self.sync_optimizer()
        sd = {"format": self.CKPT_FORMAT, "embed_size": self.emb_dim, "n_users": self.n_users, "n_items": self.n_items,
              "optimizer": self.optimizer, "table_dtype": self.table_dtype,
              "user_embedding": self.weights["user_embedding"], "item_embedding": self.weights["item_embedding"],
              "adam_t": self._t}
        if self._state is not None:
            sd.update({k: v for k, v in self._state.items() if k[0] in "mv"})
        return sd

    def load_state_dict(self, sd):
        """Validates what the checkpoint was written for before touching the tables.  Files keep the reference's NAMES
        (best_ckpt.ckpt ...) but are torch pickles: a TF checkpoint of the reference cannot be read here, nor the reverse."""
        if not isinstance(sd, dict) or "user_embedding" not in sd:
            raise ValueError("not a pda_amd checkpoint (a tf.train.Saver checkpoint of the reference cannot be loaded)")
        for key, mine in (("embed_size", self.emb_dim), ("n_users", self.n_users), ("n_items", self.n_items)):
            if key in sd and int(sd[key]) != int(mine):
                raise ValueError("checkpoint %s = %s, model has %s" % (key, sd[key], mine))
        if sd.get("table_dtype", self.table_dtype) != self.table_dtype:
            raise ValueError("checkpoint written with table_dtype=%s, model runs %s" % (sd["table_dtype"], self.table_dtype))
        if sd.get("optimizer", self.optimizer) != self.optimizer and ("mU" in sd) != (self.optimizer in ("adam", "lazy_adam")):
            raise ValueError("checkpoint written with optimizer=%s (Adam state %s), model runs %s" %
                             (sd["optimizer"], "present" if "mU" in sd else "absent", self.optimizer))
        if tuple(sd["user_embedding"].shape) != tuple(self.weights["user_embedding"].shape) or \
                tuple(sd["item_embedding"].shape) != tuple(self.weights["item_embedding"].shape):
            raise ValueError("checkpoint tables do not have the model's shape")
        self.weights["user_embedding"].copy_(sd["user_embedding"])
        self.weights["item_embedding"].copy_(sd["item_embedding"])
        if self.tables16 is not None:
            for k in self.tables16:
                ops.refresh_rows_bf16(self.weights[k], self.tables16[k])
        self._t = int(sd.get("adam_t", 0))
        self._lazy = None                     # (a checkpoint holds synced tables: every row is current for adam_t)
        if "mU" in sd:
            st = self._opt_state()
            for k in ("mU", "vU", "mI", "vI"):
                st[k].copy_(sd[k])


class BPRMF(_MFBase):
    """BPRMF.  Fetchables: opt, loss, mf_loss, reg_loss, batch_ratings  (MF/model_api.py:459-471)."""
    with_pop = False

    def __init__(self, args, data_config, use_dataset_api=False, users_api=None, pos_items_api=None,
                 neg_items_api=None, **kw):
        super().__init__(args, data_config, use_dataset_api, users_api, pos_items_api, neg_items_api, **kw)
        self.opt, self.loss = Fetch(self, "opt"), Fetch(self, "loss")
        self.mf_loss, self.reg_loss = Fetch(self, "mf_loss"), Fetch(self, "reg_loss")
        self.batch_ratings = Fetch(self, "batch_ratings")


class ConditionalBPRMF(_MFBase):
    """PD/PDA.  Fetchables: opt_pop_global, loss_pop_global, mf_loss_pop_global, reg_loss_pop_global,
    batch_ratings, condition_ratings  (MF/model_api.py:62,81-83,113)."""
    with_pop = True

    def __init__(self, args, data_config, use_dataset_api=False, users_api=None, pos_items_api=None,
                 neg_items_api=None, pos_pop_api=None, neg_pop_api=None, **kw):
        super().__init__(args, data_config, use_dataset_api, users_api, pos_items_api, neg_items_api,
                         pos_pop_api, neg_pop_api, **kw)
        self.opt_pop_global, self.loss_pop_global = Fetch(self, "opt"), Fetch(self, "loss")
        self.mf_loss_pop_global, self.reg_loss_pop_global = Fetch(self, "mf_loss"), Fetch(self, "reg_loss")
        self.batch_ratings = Fetch(self, "batch_ratings")
        self.condition_ratings = Fetch(self, "condition_ratings")
