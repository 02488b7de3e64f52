"""python -m src.flownet_css.test --input_a A --input_b B --out DIR -- same flags and
checks as /root/reference src/flownet_css/test.py:9-51; --checkpoint (.npz) and
--dtype are optional extras."""
import argparse
import os

from ..net import Mode
from .flownet_css import FlowNetCSS

FLAGS = None


def main():
    net = FlowNetCSS(mode=Mode.TEST, dtype=FLAGS.dtype)
    net.test(
        checkpoint=FLAGS.checkpoint,
        input_a_path=FLAGS.input_a,
        input_b_path=FLAGS.input_b,
        out_path=FLAGS.out,
    )


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--input_a', type=str, required=True, help='Path to first image')
    parser.add_argument('--input_b', type=str, required=True, help='Path to second image')
    parser.add_argument('--out', type=str, required=True, help='Path to output flow result')
    parser.add_argument('--checkpoint', type=str, default='./checkpoints/FlowNetCSS/flownet-CSS.ckpt-0',
                        help='.npz weights keyed by the reference variable names')
    parser.add_argument('--dtype', type=str, default='f32', choices=['f32', 'bf16', 'f16', 'f16x2'])
    FLAGS = parser.parse_args()

    # Verify arguments are valid
    if not os.path.exists(FLAGS.input_a):
        raise ValueError('image_a path must exist')
    if not os.path.exists(FLAGS.input_b):
        raise ValueError('image_b path must exist')
    if not os.path.isdir(FLAGS.out):
        raise ValueError('out directory must exist')
    main()
