"""Checkpoint semantics of the reference's saver.py / train.py, without TensorFlow's file format.

What is kept:
  * only ONE copy of the variables is written -- replica 0's, under the un-prefixed reference names
    (DataParallelSaverBuilder strips `replicated_0/`, saver.py:30-57); here every rank holds the same
    arena, so rank 0 writes it and nobody else does;
  * the checkpoint holds tf.global_variables() (train.py:188), not just the trainable ones: the BN nets'
    `.../BatchNorm/moving_mean` / `moving_variance` and the center loss's `centers` (loss.py:34-35) are written
    under their reference names next to the weights (replica 0's copies: saver.py:36-40,63-65) and restored
    with them -- an inference or fine-tune from a checkpoint normalises with the trained statistics;
  * optimizer slots and `global_step` are saved too, so the LR schedule resumes (train.py:157,207-210);
  * files are `model_dir/<net>_<model>/<net>_<model>.ckpt-<step>` plus a `checkpoint` index naming the
    latest one (tf.train.get_checkpoint_state, train.py:207); at most 20 are kept (train.py:188);
  * finetuning restores `pretrained_param` (backbone variables) only (train.py:191-193,211-213).
The container is a torch.save dict of reference-layout tensors (HWIO conv weights, [in,out] dense
weights), so a checkpoint is portable across data_format and classifier padding."""
import os
import re

import torch

MAX_TO_KEEP = 20


def _index_path(ckpt_dir):
    return os.path.join(ckpt_dir, 'checkpoint')


def _state_names(model):
    """Names of the model's non-trainable variables (GraphNet.state: `<scope>/BatchNorm/moving_mean`,
    `moving_variance`, `centers`); SphereNet has none."""
    if getattr(model, 'head', None) == 'softmax+center' and hasattr(model, '_centers') and getattr(model, 'built', True):
        model._centers()                             # created lazily by the first loss_function(): make it exist
    return list(getattr(model, 'state', None) or {})


def save(model, optimizer_slots, global_step, path_prefix):
    """path_prefix like models/<net>_<model>/<net>_<model>.ckpt ; writes <prefix>-<step>."""
    ckpt_dir = os.path.dirname(path_prefix)
    os.makedirs(ckpt_dir, exist_ok=True)
    path = '%s-%d' % (path_prefix, global_step)
    state = {'global_step': int(global_step), 'variables': {}, 'slots': []}
    for name in model.variables:
        state['variables'][name] = model.get_variable(name).cpu()
    for name in _state_names(model):                 # non-trainable global variables: BN moving statistics, centers
        state['variables'][name] = model.get_variable(name).cpu()
    for slot in optimizer_slots or []:
        state['slots'].append({name: model.get_variable(name, slot).cpu() for name in model.variables})
    torch.save(state, path)
    # index + retention
    kept = [path]
    if os.path.exists(_index_path(ckpt_dir)):
        kept = [l.strip() for l in open(_index_path(ckpt_dir)) if l.strip() and l.strip() != path] + [path]
    while len(kept) > MAX_TO_KEEP:
        old = kept.pop(0)
        if os.path.exists(old):
            os.remove(old)
    with open(_index_path(ckpt_dir), 'w') as f:
        f.write('\n'.join(kept) + '\n')
    return path


def latest_checkpoint(ckpt_dir):
    idx = _index_path(ckpt_dir)
    if not os.path.exists(idx):
        return None
    lines = [l.strip() for l in open(idx) if l.strip()]
    return lines[-1] if lines and os.path.exists(lines[-1]) else None


def step_of(path):
    m = re.search(r'-(\d+)$', path)
    return int(m.group(1)) if m else 0


def restore(model, path, optimizer=None, only=None):
    """Load variables (all, or the names in `only`) and, when given, the optimizer slots.
    Returns the saved global_step."""
    state = torch.load(path, map_location='cpu')
    names = list(model.variables) + _state_names(model) if only is None else [v if isinstance(v, str) else v.name for v in only]
    for name in names:
        if name not in state['variables']:
            raise KeyError('%s not found in checkpoint %s' % (name, path))
        model.set_variable(name, state['variables'][name])
    if optimizer is not None and only is None and state['slots']:
        optimizer._ensure()
        for slot, saved in zip(optimizer.slots, state['slots']):
            for name in model.variables:
                model.set_variable(name, saved[name], slot)
    return state['global_step']
