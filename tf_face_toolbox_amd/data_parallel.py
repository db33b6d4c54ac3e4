"""Parallel training-step builders: host-side mirror of the reference's data_parallel.py.

Reference: data_parallel.py:24-79 (Singular), :81-166 (DataParallel), :168-256
(DataParallel_margin).  Same constructor arguments and the same call protocol

    train_ops, losses, losses_name, others = wrapper(inputs)

with inputs = {'images', 'labels', 'num_classes', 'num_examples'} (data.py:275-279).
`train_ops` is a callable: one call = one `sess.run(train_ops)` of the reference
(train.py:228) = forward + loss + backward + gradient all-reduce + optimizer + global_step++.
`losses` are 0-d device tensors that hold the last step's values (reading one synchronises).

MI355X-first differences, all deliberate:
  * one PROCESS per GPU (torch.distributed, backend 'nccl' == RCCL over xGMI) instead of
    in-graph towers: `num_gpus` must equal the world size; rank r takes rows
    [r*B/n, (r+1)*B/n) of the global batch (tf.split, data_parallel.py:206-207);
  * gradients live in one flat arena and are summed by ONE bucketed all-reduce per step
    (head bucket first, launched while the conv backward still runs) instead of one
    nccl.all_sum per variable (data_parallel.py:179); the 1/num_gpus factor of :37 is
    folded into the loss-head gradient; the two displayed losses ride on the last bucket
    (data_parallel.py:248);
  * the initial replica sync of train.py:101-120 is one broadcast of the parameter arena;
  * one fused optimizer launch per decay group on the arena (data_parallel.py:186-196).
"""
from collections import OrderedDict

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _DeviceOptimizer(object):
    """tf.train.MomentumOptimizer(lr, 0.9) / AdamOptimizer(lr, beta1=0.5, beta2=0.999)
    (data_parallel.py:65-69,191-196) as fused launches over the arena."""

    def __init__(self, model, optimizer):
        if optimizer not in ('Momentum', 'Adam'):
            raise ValueError('Unsupported optimizer.')
        self.kind = optimizer
        self.model = model
        self.slots = None

    def _ensure(self):
        if self.slots is None:
            n = self.model.arena_size
            dev = self.model.params.device
            self.slots = [torch.zeros(n, dtype=torch.float32, device=dev)
                          for _ in range(1 if self.kind == 'Momentum' else 2)]

    supports_ranges = True

    def apply(self, lr, step_1based, mult_lr_list, lo=None, hi=None):
        """Update the arena (or only its [lo, hi) part: one all-reduce bucket at a time)."""
        self._ensure()
        m = self.model
        st = _stream()
        for a, b, decayed, grp in m.arena_groups():
            if lo is not None:
                a, b = max(a, lo), min(b, hi)
                if a >= b:
                    continue
            wd = m.weight_decay if decayed else 0.0
            gs = float(mult_lr_list[grp])
            if self.kind == 'Momentum':
                _lib.call('fte_momentum_update', m.params[a:b], self.slots[0][a:b], m.grads[a:b], b - a,
                          float(lr), 0.9, wd * gs, gs, st)
            else:
                _lib.call('fte_adam_update', m.params[a:b], self.slots[0][a:b], self.slots[1][a:b], m.grads[a:b],
                          b - a, float(lr), 0.5, 0.999, 1e-8, wd * gs, gs, int(step_1based), st)


class _HostComm(object):
    """The collective layer the wrappers talk to: torch.distributed (RCCL on GPUs, gloo in the
    CPU tests).  Kept tiny so that tests can drive the bucket logic without a GPU."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group

    def world_size(self):
        return self.dist.get_world_size(self.group)

    def rank(self):
        return self.dist.get_rank(self.group)

    def all_reduce_async(self, t):
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def broadcast(self, t, src=0):
        self.dist.broadcast(t, src=src, group=self.group)

    def all_gather(self, out, t):
        """out [world * len(t)] <- every rank's t, in rank order (the center loss's opt-in reconciliation)."""
        self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)


class Singular(object):
    """data_parallel.py:24-79."""

    def __init__(self, model, lr, optimizer, weight_decay=5e-4):
        self.model = model
        self.lr = lr                    # float, or callable(global_step) -> float  (train.py:122-144)
        self.optimizer = optimizer
        self.weight_decay = weight_decay
        self.num_gpus = 1
        self.global_step = 0
        self.pretrained_param = []
        self._opt = None
        self.learning_rate = None

    # -- pieces shared with the multi-GPU wrappers ------------------------------------------------
    def _fetch(self, inputs):
        images, labels = inputs['images'], inputs['labels']
        if callable(images):
            images = images()
        if callable(labels):
            labels = labels()
        return images, labels

    @staticmethod
    def _validate_labels(inputs):
        """Labels index rows of the centers table and columns of the logits inside the loss kernels: a label outside
        [0, num_classes) (list / num_classes mismatch, a hand-built inputs dict) must fail HERE, at graph-construction
        time, like TF's range error -- not as an out-of-bounds device access.  Resident label tensors are checked once
        (one host sync at setup); batches from data.train_inputs are checked per batch in the loader thread."""
        labels = inputs.get('labels')
        if callable(labels) or not isinstance(labels, torch.Tensor) or labels.numel() == 0 or 'num_classes' not in inputs:
            return
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= int(inputs['num_classes']):
            raise ValueError('labels must lie in [0, num_classes): got [%d, %d] with num_classes = %d'
                             % (lo, hi, int(inputs['num_classes'])))

    def _lr_value(self):
        return float(self.lr(self.global_step)) if callable(self.lr) else float(self.lr)

    def _forward_loss(self, images, labels, num_classes, scope):
        m = self.model
        m.tower_scale = 1.0 / self.num_gpus          # data_parallel.py:37
        m.global_step = self.global_step
        if m.needs_labels:                            # data_parallel.py:220 (margin nets)
            logits = m.forward(images, labels, num_classes=num_classes, is_training=True)
        else:                                         # data_parallel.py:51,133
            logits = m.forward(images, num_classes=num_classes, is_training=True)
        return m.loss_function(scope, labels, **logits)

    class _Construction(object):
        """The reference builds its graph here and runs NOTHING: variables get their initial values, the non-trainable
        state (BatchNorm moving statistics -- UPDATE_OPS --, the center loss's scatter_sub on `centers`) only moves when
        train_ops runs.  This engine is eager, so the construction-time forward / loss pass (it creates the variables and the
        loss handles) runs with those side effects switched off."""

        def __init__(self, model):
            self.m = model

        def __enter__(self):
            m = self.m
            self.prev = (getattr(m, 'update_moving_stats', None), getattr(m, 'update_centers', None))
            if self.prev[0] is not None:
                m.update_moving_stats = False
            if self.prev[1] is not None:
                m.update_centers = False

        def __exit__(self, *exc):
            m = self.m
            if self.prev[0] is not None:
                m.update_moving_stats = self.prev[0]
            if self.prev[1] is not None:
                m.update_centers = self.prev[1]

    def _setup(self, inputs):
        m = self.model
        self._validate_labels(inputs)
        m.weight_decay = self.weight_decay
        make = getattr(m, 'make_optimizer', None)
        self._opt = make(self.optimizer) if make is not None else _DeviceOptimizer(m, self.optimizer)

    def __call__(self, inputs):
        self._setup(inputs)
        num_classes = inputs['num_classes']
        state = {}

        def train_ops():
            images, labels = self._fetch(inputs)
            losses, names, others = self._forward_loss(images, labels, num_classes, 'TOWER')
            if not self.pretrained_param:
                self.pretrained_param = self.model.pretrained_param()
            self.model.backward()
            self.learning_rate = self._lr_value()
            self._opt.apply(self.learning_rate, self.global_step + 1, self.model.mult_lr_list())
            self.global_step += 1                     # data_parallel.py:75-77
            state['losses'], state['others'] = losses, others
            return losses, others

        # Build the variables and the loss handles now (graph construction time in the reference).
        images, labels = self._fetch(inputs)
        with self._Construction(self.model):
            losses, losses_name, others = self._forward_loss(images, labels, num_classes, 'TOWER')
        self.pretrained_param = self.model.pretrained_param()
        return train_ops, losses, losses_name, others


class DataParallel(Singular):
    """data_parallel.py:81-166: synchronous data parallelism, one replica per GPU."""

    def __init__(self, model, lr, optimizer, num_gpus=4, weight_decay=5e-4, comm=None, sync_centers=False):
        """`sync_centers` (not in the reference; default off = the reference's behaviour): the center loss's `centers` table
        is per-tower state there (loss.py:34-39) and the towers' tables drift apart; True all-gathers every step's scatter rows
        so that all replicas keep ONE table, equal to the single-tower update of the global batch (nets/graph.py)."""
        assert num_gpus > 1, 'DataParallel objects are only used for multi-gpu training tasks.'
        super(DataParallel, self).__init__(model, lr, optimizer, weight_decay)
        self.num_gpus = num_gpus
        self.pretrained_param = []
        self.comm = comm
        self.sync_centers = bool(sync_centers)
        self._global_batch = None

    def _shard(self, t):
        """tf.split(axis=0, num_or_size_splits=num_gpus) (data_parallel.py:206-207): this rank's rows.
        With inputs['batch_size'] given, a tensor that already has batch_size/num_gpus rows is taken
        as this rank's shard (each rank's loader reads only its own rows)."""
        n = t.shape[0]
        if self._global_batch is not None and n * self.num_gpus == self._global_batch:
            return t
        assert n % self.num_gpus == 0, 'batch size must be divisible by num_gpus (train.py:98)'
        sh = n // self.num_gpus
        r = self.comm.rank()
        return t[r * sh:(r + 1) * sh]

    def _reduced_opt(self):
        """data_parallel.py:88-113 / :175-200.  Backward runs bucket by bucket; as soon as a bucket of
        the gradient arena is final its sum-all-reduce is enqueued (RCCL's own stream), so the head
        bucket (classifier + FC, 73 MB of the 120 MB) crosses xGMI under the conv backward.  Then every
        replica applies the same summed gradient to its own copy with its own slots."""
        m = self.model
        works = []
        buckets = m.grad_buckets()
        for stage, (a, b) in zip(m.backward_stages(), buckets):
            stage()
            works.append(self.comm.all_reduce_async(m.grads[a:b]))
        self.learning_rate = self._lr_value()
        if getattr(self._opt, 'supports_ranges', False):
            # bucket by bucket, in completion order: the head bucket (73 of the 120 MB) is updated while the later, smaller
            # all-reduces are still crossing xGMI -- the kernel stream only ever waits for the bucket it is about to update
            for w, (a, b) in zip(works, buckets):
                w.wait()
                self._opt.apply(self.learning_rate, self.global_step + 1, m.mult_lr_list(), a, min(b, m.arena_size))
        else:
            for w in works:
                w.wait()
            self._opt.apply(self.learning_rate, self.global_step + 1, m.mult_lr_list())

    def __call__(self, inputs):
        if self.comm is None:
            self.comm = _HostComm()
        assert self.comm.world_size() == self.num_gpus, \
            'one process per GPU: launch with torch.distributed.run --nproc-per-node %d' % self.num_gpus
        self._setup(inputs)
        if hasattr(self.model, 'center_comm'):
            self.model.center_comm = self.comm if self.sync_centers else None
        num_classes = inputs['num_classes']
        self._global_batch = inputs.get('batch_size')
        scope = 'TOWER_%d' % self.comm.rank()
        # the reference's towers draw independent dropout masks (one layers.dropout op per tower); a shared seed would
        # give every replica the same mask for its shard
        if hasattr(self.model, 'dropout_seed'):
            base = getattr(self.model, '_dropout_seed_base', None)
            if base is None:
                base = self.model._dropout_seed_base = int(self.model.dropout_seed)
            self.model.dropout_seed = base * self.num_gpus + self.comm.rank()

        def tower():
            images, labels = self._fetch(inputs)
            return self._forward_loss(self._shard(images), self._shard(labels), num_classes, scope)

        def train_ops():
            losses, names, others = tower()
            self._reduced_opt()
            self.global_step += 1                     # data_parallel.py:160-161
            return losses, others

        with self._Construction(self.model):
            losses, losses_name, others = tower()
        # train.py:101-120: every replica starts from replica 0's values
        self.comm.broadcast(self.model.params, src=0)
        self.pretrained_param = self.model.pretrained_param()
        return train_ops, losses, losses_name, others


class DataParallel_margin(DataParallel):
    """data_parallel.py:168-256: identical to DataParallel except that labels go into
    model.forward (:220) and `others` are collected per tower (:225-228).  Both behaviours are
    selected by the model (`needs_labels`) here, so the subclass only keeps the name."""
    pass
