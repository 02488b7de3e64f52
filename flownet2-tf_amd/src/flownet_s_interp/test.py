"""python -m src.flownet_s_interp.test --input_a I1 --matches_a MASK --sparse_flow SF.flo --out DIR
(flags of /root/reference src/flownet_s_interp/test.py:60-190 that apply to single-frame inference; a .txt
--input_a runs Net.test_batch on image pairs).  --checkpoint (.npz) and --dtype are this build's extras."""
import argparse
import os

from ..net import Mode
from .flownet_s_interp import FlowNetS_interp

FLAGS = None


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def main():
    net = FlowNetS_interp(mode=Mode.TEST, no_deconv_biases=FLAGS.no_deconv_biases, dtype=FLAGS.dtype)
    if not os.path.isfile(FLAGS.input_a):
        raise ValueError("'input_a' is not valid, should be a path to a folder or a single image")
    net.test(
        checkpoint=FLAGS.checkpoint,
        input_a_path=FLAGS.input_a,
        input_b_path=FLAGS.input_b,
        matches_a_path=FLAGS.matches_a,
        sparse_flow_path=FLAGS.sparse_flow,
        input_type='image_matches',
        out_path=FLAGS.out,
        gt_flow=FLAGS.gt_flow,
        save_flo=FLAGS.save_flo,
        save_image=FLAGS.save_image,
        compute_metrics=FLAGS.compute_metrics,
        new_par_folder=FLAGS.new_par_folder,
    )


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--input_a', type=str, required=True, help='Path to first image')
    parser.add_argument('--input_b', type=str, default=None, help='Path to second image (unused by the network)')
    parser.add_argument('--matches_a', type=str, required=True, help='Path to matches mask')
    parser.add_argument('--sparse_flow', type=str, required=True, help='Sparse flow initialized from sparse matches')
    parser.add_argument('--checkpoint', type=str, default='./checkpoints/FlowNetS/flownet-S.ckpt-0')
    parser.add_argument('--no_deconv_biases', type=str2bool, nargs='?', default=False)
    parser.add_argument('--out', type=str, required=True, help='Path to the output folder')
    parser.add_argument('--gt_flow', type=str, default=None)
    parser.add_argument('--save_flo', type=str2bool, nargs='?', default=True)
    parser.add_argument('--save_image', type=str2bool, nargs='?', default=True)
    parser.add_argument('--compute_metrics', type=str2bool, nargs='?', default=True)
    parser.add_argument('--new_par_folder', type=str, default=None)
    parser.add_argument('--dtype', type=str, default='f32', choices=['f32', 'bf16', 'f16', 'f16x2'])
    FLAGS = parser.parse_args()
    for flag in ('input_a', 'matches_a', 'sparse_flow'):
        if not os.path.exists(getattr(FLAGS, flag)):
            raise ValueError('%s path must exist' % flag)
    if not os.path.isdir(FLAGS.out):
        raise ValueError('out directory must exist')
    main()
