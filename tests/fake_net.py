"""TEST DOUBLE (lives in tests/, never shipped): a CPU model with the same arena / bucket /
backward-stage interface as tf_face_toolbox_amd.nets.sphere.SphereNet, whose numbers come from the
oracle.  It lets the gloo world_size-2 tests drive the REAL DataParallel logic (sharding, bucketed
all-reduce, loss slots, replica broadcast, update order) without a GPU."""
from collections import OrderedDict

import numpy as np
import torch

from oracle import ops, spherenet as osn


class CpuMomentum(object):
    def __init__(self, model):
        self.m = model
        self.acc = torch.zeros(model.arena_size, dtype=torch.float64)

    supports_ranges = True                      # DataParallel updates bucket by bucket as the all-reduces complete

    def apply(self, lr, step_1based, mult_lr_list, lo=None, hi=None):
        m = self.m
        for a, b, decayed, grp in m.arena_groups():
            if lo is not None:
                a, b = max(a, lo), min(b, hi)
                if a >= b:
                    continue
            gs = float(mult_lr_list[grp])
            wd = m.weight_decay if decayed else 0.0
            g = gs * m.grads[a:b] + wd * gs * m.params[a:b]
            self.acc[a:b] = 0.9 * self.acc[a:b] + g
            m.params[a:b] -= lr * self.acc[a:b]


class FakeOracleNet(object):
    needs_labels = False
    name = 'SphereNet'

    def __init__(self, seed, h, w, ch, ncls, weight_decay=5e-4, perturb_rank=0):
        self.weight_decay = weight_decay
        self.tower_scale = 1.0
        self.global_step = 0
        self.h, self.w, self.ch, self.ncls = h, w, ch, ncls
        p = osn.perturb_params(osn.init_params(seed + perturb_rank, ch, ncls, h, w), seed + 1)
        self.names = [k for k in p if not k.endswith('/weights')] + [k for k in p if k.endswith('/weights')]
        self.shapes = {k: p[k].shape for k in p}
        self.offsets, off = {}, 0
        for k in self.names:
            self.offsets[k] = off
            off += p[k].size
        self.arena_size = off
        self.small_end = self.offsets[[k for k in self.names if k.endswith('/weights')][0]]
        self.fc_start = self.offsets['SphereNet/fully_connected/weights']
        self.cls_start = self.offsets['classifier/fc_classifier/weights']
        self.params = torch.zeros(off, dtype=torch.float64)
        self.grads = torch.zeros(off + 4, dtype=torch.float64)
        self.loss_slots = self.grads[off:off + 4]
        for k in self.names:
            self.params[self.offsets[k]:self.offsets[k] + p[k].size] = torch.from_numpy(p[k].reshape(-1))
        self.stage_log = []

    def as_dict(self):
        return OrderedDict((k, self.params[self.offsets[k]:self.offsets[k] + int(np.prod(self.shapes[k]))]
                            .numpy().reshape(self.shapes[k]).copy()) for k in self.names)

    # ---- Network-like surface used by Singular / DataParallel -------------------------------
    def forward(self, images, num_classes=None, is_training=True):
        self._x = images.numpy().astype(np.float64)
        return {'logits': None}

    def loss_function(self, scope, labels, **logits):
        y = labels.numpy()
        n = y.shape[0]
        losses, g, _ = osn.loss_and_grads(self.as_dict(), self._x, y, 0.0, 'NCHW', 'softmax', None,
                                          grad_scale=self.tower_scale / n)
        self._g = g
        reg = ops.l2_reg([v for k, v in self.as_dict().items() if k.endswith('/weights')], self.weight_decay)
        self.loss_slots[0] = losses[0] * self.tower_scale
        self.loss_slots[1] = reg * self.tower_scale
        return [self.loss_slots[0], self.loss_slots[1]], ['cross_entropy', 'reg_loss'], OrderedDict()

    def _fill(self, lo, hi):
        for k in self.names:
            o = self.offsets[k]
            if lo <= o < hi:
                self.grads[o:o + self._g[k].size] = torch.from_numpy(self._g[k].reshape(-1))

    def backward_stages(self):
        def head():
            self.stage_log.append('head')
            self._fill(self.fc_start, self.arena_size)

        def body():
            self.stage_log.append('body')
            self._fill(0, self.fc_start)
        return [head, body]

    def backward(self):
        for s in self.backward_stages():
            s()

    def grad_buckets(self):
        return [(self.fc_start, self.arena_size + 4), (0, self.fc_start)]

    def arena_groups(self):
        return [(0, self.small_end, False, 0), (self.small_end, self.cls_start, True, 0),
                (self.cls_start, self.arena_size, True, 1)]

    def mult_lr_list(self, scope=None):
        return [1.0, 1.0]

    def pretrained_param(self, scope=None):
        return [k for k in self.names if k.startswith('SphereNet/')]

    def make_optimizer(self, kind):
        assert kind == 'Momentum'
        return CpuMomentum(self)
