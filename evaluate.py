#!/usr/bin/env python
"""evaluate.py -- the reference's feature-extraction CLI (evaluate.py:18-166) on the MI355X engine:
restore the latest checkpoint, run forward(is_training=False) (flip-averaged embedding for SphereNet,
nets/sphere.py:97-101) over a list, save `wfea` to feature_dir/<net>_<model>/<fea_name>_<step>.mat."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--visual_embedding', type=bool, default=False, help='Ignored (TensorBoard projector).')
    parser.add_argument('--net_name', type=str, help='Name of the network architecture.')
    parser.add_argument('--model_name', type=str, help='Name of the training model.')
    parser.add_argument('--fea_name', type=str, help='Name of the feature file.')
    parser.add_argument('--eval_dir', type=str, default='eval', help='Directory where to write event logs.')
    parser.add_argument('--feature_dir', type=str, default='features', help='Directory where to save features.')
    parser.add_argument('--model_dir', type=str, default='models', help='Directory where to read model checkpoints.')
    parser.add_argument('--input_height', type=int, default=128, help='The height of input images.')
    parser.add_argument('--input_width', type=int, default=128, help='The width of input images.')
    parser.add_argument('--is_color', type=int, default=1, help='Whether to read inputs as RGB images.')
    parser.add_argument('--flip_flag', type=bool, default=False, help='Unused in the reference as well.')
    parser.add_argument('--data_list_path', type=str, help='Path to the list of testing data.')
    parser.add_argument('--batch_size', type=int, default=256, help='Number of images to process in a batch.')
    parser.add_argument('--data_format', type=str, default='NCHW', help='net_select default (evaluate.py:62).')
    return parser


def evaluate(FLAGS):
    import torch
    from scipy.io import savemat
    from tf_face_toolbox_amd import net_select, saver
    from tf_face_toolbox_amd.data import eval_inputs

    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    next_images, num_images = eval_inputs(FLAGS.data_list_path, batch_size=FLAGS.batch_size, input_height=FLAGS.input_height,
                                          input_width=FLAGS.input_width, is_color=FLAGS.is_color, device=device)
    tag = FLAGS.net_name + '_' + FLAGS.model_name
    latest = saver.latest_checkpoint(os.path.join(FLAGS.model_dir, tag))
    if not latest:
        raise IOError('No checkpoint file found')
    try:
        state = torch.load(latest, map_location='cpu')
        # The extractor needs the backbone only (evaluate.py:62-63: forward(images, is_training=False), no num_classes).  This
        # engine sizes its arena at build time, so the classifier's width is read from the checkpoint WHEN the net has one;
        # nets trained without a classifier (the triplet heads: nets/resnet.py:67-68) are built without one.
        cls = [k for k in state['variables'] if k.startswith('classifier/') and k.endswith('/weights')]
        ncls = int(state['variables'][cls[0]].shape[-1]) if cls else 1
        del state
        model = net_select(FLAGS.net_name, FLAGS.data_format)
        model.build(FLAGS.input_height, FLAGS.input_width, 3 if FLAGS.is_color else 1, ncls, device)
        saver.restore(model, latest)
        step = str(saver.step_of(latest))
        print('Extracting features from model saved in iteration %s...' % step)
        wfea = None
        while wfea is None or wfea.shape[0] < num_images:
            start_time = time.time()
            # SphereNet: forward(is_training=False) is the flip-averaged embedding (nets/sphere.py:97-101); the graph nets: the
            # pooled backbone features (their reference forward() cannot be called without num_classes, nets/resnet.py:147)
            batch = next_images()
            fea = (model.eval_features(batch) if hasattr(model, 'eval_features') else model.forward(batch, is_training=False)).cpu().numpy()
            wfea = fea if wfea is None else np.vstack((wfea, fea))
            print('%d/%d features extracted... %.2fms elapsed' % (min(wfea.shape[0], num_images), num_images,
                                                                (time.time() - start_time) * 1000))
    finally:
        # orderly shutdown: producer thread joined, decode workers reaped, device drained (see train.py _run_and_leave)
        try:                                    # never raise from a finally block: a checkpoint / device error on its way out must stay the one reported
            if not next_images.close():
                print('evaluate.py: the input pipeline did not shut down cleanly', file=sys.stderr)
                FLAGS._shutdown_failed = True
        except Exception as e:
            print('evaluate.py: closing the input pipeline failed: %s' % e, file=sys.stderr)
            FLAGS._shutdown_failed = True
        try:
            torch.cuda.synchronize()
        except Exception as e:
            print('evaluate.py: device synchronise failed during shutdown: %s' % e, file=sys.stderr)
    wfea = wfea[0:num_images, :]
    print('Totally extracted %d features.' % (wfea.shape[0]))
    print('Saving features to .mat files...')
    out_dir = os.path.join(FLAGS.feature_dir, tag)
    os.makedirs(out_dir, exist_ok=True)
    savemat(os.path.join(out_dir, FLAGS.fea_name + '_' + step + '.mat'), {'wfea': wfea})
    print('Done.')
    return FLAGS                               # (_run_and_leave reads _shutdown_failed from it)


if __name__ == '__main__':
    from train import _run_and_leave           # orderly shutdown above; leaves with the real status (not an unconditional 0)
    _run_and_leave(lambda: evaluate(build_parser().parse_args()))
