#!/usr/bin/env python
"""train.py -- the reference's training CLI (train.py:33-262) on the MI355X engine.

Same flags, defaults, assertions, LR schedules, log line and checkpoint cadence as the reference
(SURVEY.md Appendix D).  One process per GPU: `python train.py --num_gpus N ...` (the reference's invocation) starts its own N
ranks through torch.distributed.run; under torch.distributed.run it runs as one rank (WORLD_SIZE must equal --num_gpus).
Extra, not in the reference: --synthetic 1 trains on a resident random batch when no list is given,
--max_steps stops early (smoke runs)."""
import argparse
import math
import os
import sys
import time
from datetime import datetime

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    parser = argparse.ArgumentParser()
    # Name configures
    parser.add_argument('--net_name', type=str, help='Name of the network architecture.')
    parser.add_argument('--model_name', type=str, help='Name of the training model.')
    # Directory configures
    parser.add_argument('--train_dir', type=str, default='train', help='Root directory where to write event logs.')
    parser.add_argument('--model_dir', type=str, default='models', help='Root directory where to save checkpoints.')
    parser.add_argument('--pretrained_path', type=str, default='', help='Path to save pretrained checkpoints.')
    # Data configure
    parser.add_argument('--data_format', type=str, default='NCHW', help='The format of data in the network (NCHW(default) or NHWC).')
    parser.add_argument('--train_list_path', type=str, help='Path to the list of training data.')
    parser.add_argument('--input_height', type=int, default=384, help='The height of input images.')
    parser.add_argument('--input_width', type=int, default=128, help='The width of input images.')
    parser.add_argument('--crop_height', type=int, default=-1, help='The height of input images.')
    parser.add_argument('--crop_width', type=int, default=-1, help='The width of input images.')
    parser.add_argument('--is_color', type=int, default=1, help='Whether to read inputs as RGB images.')
    parser.add_argument('--augmentation', type=int, default=0, help='Whether to employ data augmentation to training set.')
    # Hyperparameters configure
    parser.add_argument('--batch_size', type=int, default=-1, help='Number of sampled images in a batch.')
    parser.add_argument('--num_classes', type=int, default=-1, help='Number of sampled classesin a batch.')
    parser.add_argument('--num_per_class', type=int, default=-1, help='Number of sampled images per class in a batch.')
    parser.add_argument('--optimizer', type=str, default='Momentum', help='Type of optimizer.')
    parser.add_argument('--init_lr', type=float, default=0.1, help='Initial learning rate.')
    parser.add_argument('--lr_decay_method', type=str, default='step', help='Learning rate strategy (step/cosine/exp).')
    parser.add_argument('--lr_decay_rate', type=float, default=0.1, help='Learning rate decay rate.')
    parser.add_argument('--lr_decay_epoch', type=str, default='', help='Boundaries of decaying learning rate in step lr_decay')
    parser.add_argument('--max_epoches', type=int, help='Number of batches to run.')
    parser.add_argument('--weight_decay', type=float, default=5e-4, help='Factor for weight decaying.')
    # Device configures
    parser.add_argument('--num_gpus', type=int, default=4, help='Number of GPUs to use.')
    # Interval configures
    parser.add_argument('--display_interval', type=int, default=10, help='Internal iterations of verbose.')
    parser.add_argument('--save_interval', type=int, default=1000, help='Internal iterations of saving models.')
    # Not in the reference
    parser.add_argument('--synthetic', type=int, default=0, help='1: resident random batch instead of a list file.')
    parser.add_argument('--synthetic_classes', type=int, default=10575)
    parser.add_argument('--sync_centers', type=int, default=0,
                        help='center-loss nets under --num_gpus > 1.  0 (default) = the reference: every replica keeps and updates its own '
                             '`centers` table from its own shard (loss.py:34-39) and the tables drift apart; 1 = the replicas all-gather '
                             "each step's scatter rows and keep ONE table, equal to the single-tower update of the global batch.")
    parser.add_argument('--max_steps', type=int, default=-1, help='Stop after this many steps (smoke runs).')
    parser.add_argument('--mfma_dtype', type=str, default='f32',
                        help='f32 (the reference arithmetic), bf16 (bf16 MFMA operands, fp32 accumulate and storage) or bf16s (bf16 operands and bf16 '
                             'storage of activations / inter-layer gradients; fp32 accumulate, sums and master weights: SphereNet).')
    return parser


def FLAGS_assertion(FLAGS):
    """train.py:95-99."""
    assert FLAGS.data_format in ['NCHW', 'NHWC'], 'Unknown data format.'
    assert FLAGS.batch_size != -1 or (FLAGS.num_classes != -1 and FLAGS.num_per_class != -1)
    assert (FLAGS.num_classes != -1 and FLAGS.num_per_class != -1 and (FLAGS.num_classes * FLAGS.num_per_class) % FLAGS.num_gpus == 0) \
        or (FLAGS.batch_size % FLAGS.num_gpus == 0 and FLAGS.batch_size != -1)
    assert FLAGS.optimizer in ['Momentum', 'Adam'], 'Unsupported optimizer.'


def lr_config(FLAGS, method, batches_per_epoch):
    """train.py:122-144 as a function of the global step (tf.train.piecewise_constant /
    exponential_decay / cosine_decay semantics)."""
    if method == 'step':
        if FLAGS.lr_decay_epoch == '':
            raise ValueError('Empty learning rate decay epoch boundaries.')
        decay_boundary = [(int(epoch) - 1) * batches_per_epoch for epoch in FLAGS.lr_decay_epoch.split(',')]
        decay_value = [FLAGS.init_lr] + [FLAGS.init_lr * FLAGS.lr_decay_rate ** (p + 1) for p in range(len(decay_boundary))]

        def lr(step):       # value[i] for boundary[i-1] < step <= boundary[i]
            k = 0
            while k < len(decay_boundary) and step > decay_boundary[k]:
                k += 1
            return decay_value[k]
    elif method == 'exp':
        decay_step = int(FLAGS.lr_decay_epoch) * batches_per_epoch
        decay_steps = int(FLAGS.max_epoches) * batches_per_epoch + 1 - decay_step

        def lr(step):
            if step < decay_step:
                return FLAGS.init_lr
            return FLAGS.init_lr * 0.001 ** ((step - decay_step) / float(decay_steps))
    elif method == 'cosine':
        total = FLAGS.max_epoches * batches_per_epoch

        def lr(step):
            return FLAGS.init_lr * 0.5 * (1 + math.cos(math.pi * min(step, total) / total))
    else:
        raise ValueError('Unsupported learning rate decaying method.')
    return lr


def format_str(losses_name):
    """train.py:146-152, verbatim."""
    outputs = '[%s] Epoch/Step %d/%d, lr = %g\n'
    for loss_id, loss_name in enumerate(losses_name):
        outputs += '[%s]    Loss #' + str(loss_id) + ': ' + loss_name + ' = %.6f\n'
    outputs += '[%s]    batch_time = %.1fms/batch, throughput = %.1fimages/s'
    return outputs


def train(FLAGS):
    import torch
    import torch.distributed as dist
    from tf_face_toolbox_amd import net_select, Singular, DataParallel_margin, saver
    from tf_face_toolbox_amd.data import train_inputs, synthetic_inputs

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if FLAGS.num_gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python train.py --num_gpus N` as the reference is invoked: this process starts the N ranks as fresh children (it has not
        # touched the GPU and never will), passes their output through and exits with their status
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(FLAGS.num_gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', '8')
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    if FLAGS.num_gpus != world:
        raise SystemExit('--num_gpus %d but WORLD_SIZE is %d: one process per GPU (python -m torch.distributed.run --nproc-per-node %d '
                         '--master-addr 127.0.0.1 train.py ..., or plain `python train.py --num_gpus %d`, which launches them)'
                         % (FLAGS.num_gpus, world, FLAGS.num_gpus, FLAGS.num_gpus))
    # FTE_BENCH_SHARED_GPU=1 (tests only, as in bench.py): all ranks on the GPUs that exist, gloo as the transport -- RCCL refuses two
    # ranks on one device and the test boxes have one GPU; everything else of the N > 1 path is what a multi-GPU node runs
    shared = os.environ.get('FTE_BENCH_SHARED_GPU') == '1'
    if shared:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if shared:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)
    from tf_face_toolbox_amd import _lib
    _lib.set_mfma_dtype(FLAGS.mfma_dtype)

    batch_size = FLAGS.batch_size if FLAGS.batch_size != -1 else FLAGS.num_classes * FLAGS.num_per_class
    # Data I/O (train.py:161-170)
    if FLAGS.synthetic:
        h = FLAGS.crop_height if FLAGS.crop_height != -1 else FLAGS.input_height
        w = FLAGS.crop_width if FLAGS.crop_width != -1 else FLAGS.input_width
        inputs = synthetic_inputs(batch_size, h, w, FLAGS.is_color, FLAGS.synthetic_classes, device, rank, world)
    else:
        inputs = train_inputs(FLAGS.train_list_path, input_height=FLAGS.input_height, input_width=FLAGS.input_width,
                              crop_height=FLAGS.crop_height, crop_width=FLAGS.crop_width, is_color=FLAGS.is_color,
                              augmentation=FLAGS.augmentation, batch_size=FLAGS.batch_size, num_classes=FLAGS.num_classes,
                              num_per_class=FLAGS.num_per_class, device=device, seed=1234, rank=rank, world_size=world)
    try:
        batches_per_epoch = inputs['num_examples'] // batch_size + 1                                  # train.py:172
        network = net_select(FLAGS.net_name, FLAGS.data_format, FLAGS.weight_decay)                  # train.py:174
        lr = lr_config(FLAGS, FLAGS.lr_decay_method, batches_per_epoch)                              # train.py:176
        if FLAGS.num_gpus > 1:                                                                        # train.py:178-183
            model = DataParallel_margin(network, lr, optimizer=FLAGS.optimizer, weight_decay=FLAGS.weight_decay, num_gpus=FLAGS.num_gpus,
                                        sync_centers=bool(FLAGS.sync_centers))
        else:
            model = Singular(network, lr, optimizer=FLAGS.optimizer, weight_decay=FLAGS.weight_decay)
        train_ops, losses, losses_name, others = model(inputs)

        tag = FLAGS.net_name + '_' + FLAGS.model_name
        ckpt_dir = os.path.join(FLAGS.model_dir, tag)
        latest = saver.latest_checkpoint(ckpt_dir)                                                    # train.py:207-215
        if latest:
            model.global_step = saver.restore(network, latest, optimizer=model._opt)
            print('Model restored from %s' % ckpt_dir)
        elif FLAGS.pretrained_path != '':
            saver.restore(network, FLAGS.pretrained_path, only=model.pretrained_param)
            print('Network parameters initialized from %s' % FLAGS.pretrained_path)
        else:
            print('Network parameters initialized from scratch.')
        if world > 1:
            model.comm.broadcast(network.params, src=0)

        print('%s training start...' % tag)
        step, epoch = 0, 1
        time_sim, image_sim = 0.0, 0.0
        first_step, settled = None, None                          # (step, wall clock) 20 steps after the start: the sustained rate printed at the end
        while epoch <= FLAGS.max_epoches:                                                             # train.py:223-250
            step = model.global_step
            epoch = step // batches_per_epoch + 1
            start_time = time.time()
            first_step = step if first_step is None else first_step
            if settled is None and step - first_step >= 20:
                settled = (step, start_time)
            train_ops()
            losses_value = [float(l) for l in losses]            # reading the losses synchronises, like sess.run
            last_done = time.time()
            duration = last_done - start_time
            if not all(math.isfinite(v) for v in losses_value):
                raise SystemExit('Model diverged with losses = %s' % losses_value)
            if step % FLAGS.display_interval == 0 and rank == 0:
                format_list = [datetime.now(), epoch, step, model.learning_rate]
                for loss_value in losses_value:
                    format_list.extend([datetime.now(), loss_value])
                format_list.extend([datetime.now(), duration * 1000, batch_size / duration])
                print(format_str(losses_name) % tuple(format_list))
                for other_name, other_value in others.items():
                    print('%s: %s' % (other_name, other_value))
            if step > 0:
                time_sim += duration
                image_sim += batch_size / duration
            last = step == FLAGS.max_epoches * batches_per_epoch or (FLAGS.max_steps > 0 and step + 1 >= FLAGS.max_steps)
            if ((step > 0 and step % FLAGS.save_interval == 0) or last) and rank == 0:
                path = saver.save(network, model._opt.slots, model.global_step, os.path.join(ckpt_dir, tag + '.ckpt'))
                print('[%s]: Model has been saved in Iteration %d (%s)' % (datetime.now(), step, path))
            if FLAGS.max_steps > 0 and step + 1 >= FLAGS.max_steps:
                break
        if rank == 0 and step > 0:
            print('mean batch_time=%.2f, mean throughput=%.2f' % (time_sim / step * 1000, image_sim / step))
            if settled is not None and step + 1 > settled[0]:      # not in the reference: images / wall clock once the pipeline has settled
                print('sustained throughput=%.2f images/s over steps %d..%d' % ((step + 1 - settled[0]) * batch_size / (last_done - settled[1]), settled[0], step))
        if world > 1:
            dist.barrier()
    finally:
        # orderly shutdown, whatever happened above: stop the input pipeline's producer thread and wait for it, end the decode
        # workers and release their shared buffers, drain the device, leave the process group
        close = inputs.get('close')
        try:
            if close is not None and not close():
                print('train.py: the input pipeline did not shut down cleanly', file=sys.stderr)
                FLAGS._shutdown_failed = True
        except Exception as e:                 # never raise from here: the error that brought us here (if any) must stay the one reported
            print('train.py: closing the input pipeline failed: %s' % e, file=sys.stderr)
            FLAGS._shutdown_failed = True
        try:
            torch.cuda.synchronize()
        except Exception as e:                 # a device fault is what brought us here: report it, keep the original error
            print('train.py: device synchronise failed during shutdown: %s' % e, file=sys.stderr)
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()


def main(argv=None):
    FLAGS = build_parser().parse_args(argv)
    FLAGS_assertion(FLAGS)
    os.makedirs(os.path.join(FLAGS.train_dir, FLAGS.net_name + '_' + FLAGS.model_name), exist_ok=True)
    os.makedirs(os.path.join(FLAGS.model_dir, FLAGS.net_name + '_' + FLAGS.model_name), exist_ok=True)
    train(FLAGS)
    return FLAGS


def _run_and_leave(fn):
    """Runs the program and leaves with ITS status.  By the time fn() returns or raises, everything this program started has been
    shut down in order (train()'s finally block: producer thread joined, decode workers reaped, device drained, process group
    destroyed), so nothing of ours can still fail.  What remains is interpreter / HIP runtime teardown, where ROCm 7 occasionally
    ends a finished process with std::terminate (exit code -6: torch's helper threads race the runtime's static destructors).
    Only when the GPU runtime was initialised, os._exit skips that teardown -- with the real status (a failure stays a failure;
    round 2 left with an unconditional 0)."""
    import traceback
    status = 0
    try:
        status = 1 if getattr(fn(), '_shutdown_failed', False) else 0
    except SystemExit as e:
        if e.code is None or e.code == 0:
            status = 0
        elif isinstance(e.code, int):
            status = e.code
        else:
            print(e.code, file=sys.stderr)
            status = 1
    except BaseException:                      # noqa: B902
        traceback.print_exc()
        status = 1
    sys.stdout.flush()
    sys.stderr.flush()
    gpu_up = False
    try:
        import torch
        gpu_up = torch.cuda.is_initialized()
    except Exception:
        pass
    if gpu_up:
        os._exit(status)
    sys.exit(status)


if __name__ == '__main__':
    _run_and_leave(main)
