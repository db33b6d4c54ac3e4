"""Host-side mirror of the reference's loss.py: focal_loss (:18-27), center_loss (:29-45), batch_hard_triplet_loss
(:47-78) on libfte.so.  Same names, argument meaning and defaults.  There is no autograd here, so every function
also returns the gradient of ITS OWN loss value with respect to its first argument (what tf.gradients would have
produced for that term); the graph nets wire them as heads (nets/graph.py), a caller can combine them freely.

All tensors are float32 / int32 CUDA tensors; logits may carry padding columns (ld = logits.shape[1] >= num_classes)."""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(t, dtype, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype):
        raise TypeError('%s must be a %s CUDA tensor' % (what, dtype))
    return t.contiguous()


def _scaled_sum(rows, scale):
    """0-d tensor scale * sum(rows) on the library's ordered two-stage reduction (no ATen kernel on the step)"""
    out = torch.empty(1, dtype=torch.float32, device=rows.device)
    ws = torch.empty(2048, dtype=torch.float32, device=rows.device)
    _lib.call('fte_sum', rows, rows.numel(), float(scale), out, ws, ws.numel() * 4, _stream())
    return out[0]


def focal_loss(logits, labels, gamma=1.0, alpha=2.0, num_classes=None):
    """mean_i gamma * (1 - p_y)^alpha * CE_i (loss.py:18-27; the reference's parameter names are kept as written).
    -> (loss [0-d], dlogits [N, ld])"""
    logits, labels = _check(logits, torch.float32, 'logits'), _check(labels, torch.int32, 'labels')
    n, ld = logits.shape
    c = ld if num_classes is None else int(num_classes)
    rows = torch.empty(n, dtype=torch.float32, device=logits.device)
    d = torch.empty_like(logits)
    _lib.call('fte_focal_loss_fwd_bwd', logits, labels, rows, d, n, c, ld, float(gamma), float(alpha), 1.0 / n, _stream())
    return _scaled_sum(rows, 1.0 / n), d


def center_loss(features, labels, num_classes, alpha=0.99, weight=1.0, centers=None):
    """loss.py:29-45.  `centers` [num_classes, D] is the non-trainable variable the reference creates with zeros
    (:34-35); pass the tensor to keep state across calls, it is updated in place (scatter_sub of (1-alpha)(c_y - f),
    duplicates accumulate, no count normalisation).  -> (center_loss_mean, centers, dfeatures) where dfeatures is the
    gradient of weight * center_loss_mean (the term the reference adds to the 'losses' collection, :43)."""
    features, labels = _check(features, torch.float32, 'features'), _check(labels, torch.int32, 'labels')
    n, d = features.shape
    if centers is None:
        centers = torch.zeros(int(num_classes), d, dtype=torch.float32, device=features.device)
    rows = torch.empty(n, dtype=torch.float32, device=features.device)
    df = torch.empty_like(features)
    ws = torch.empty(max(n * d, 1024) + 1024, dtype=torch.float32, device=features.device)
    _lib.call('fte_center_loss_fwd_bwd_update', features, labels, centers, rows, df, n, d, int(centers.shape[0]), float(alpha),
              float(weight) / (n * d), ws, ws.numel() * 4, _stream())
    return _scaled_sum(rows, 1.0 / (n * d)), centers, df


def batch_hard_triplet_loss(features, labels, margin=None, metric='euclidean'):
    """loss.py:47-78: per-sample batch-hard triplet loss (UNREDUCED, as the reference returns it): softplus(pos - neg)
    for margin None, else max(0, pos - neg + margin).  -> (diff [N], dfeatures of mean(diff)).
    `metric` is accepted and IGNORED, as in the reference: loss.py:65 calls `cdist(features, features)` without forwarding it, so
    every value gives the euclidean distance sqrt(sum d^2 + 1e-12) of loss.py:57."""
    features, labels = _check(features, torch.float32, 'features'), _check(labels, torch.int32, 'labels')
    n, d = features.shape
    rows = torch.empty(n, dtype=torch.float32, device=features.device)
    df = torch.empty_like(features)
    ws = torch.empty(max(4 * n * n, 1024) + 1024, dtype=torch.float32, device=features.device)
    _lib.call('fte_batch_hard_triplet_fwd_bwd', features, labels, 0.0 if margin is None else float(margin), int(margin is None), 1.0 / n,
              rows, df, n, d, ws, ws.numel() * 4, _stream())
    return rows, df
