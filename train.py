#!/usr/bin/env python
"""Command-line twin of the reference's ``example_train.py`` on the native executor: same flags
(``-s/--source_path -m/--model_path -i/--images -r/--resolution --eval --iterations ... --test_epochs --save_epochs
--checkpoint_epochs --start_checkpoint``), calls ``litegs_amd.training.start``.  Data parallel: launch with
``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py ...`` (one process per GPU, RCCL)."""
import sys
from argparse import ArgumentParser

from litegs_amd import arguments, training

if __name__ == "__main__":
    parser = ArgumentParser(description="Training script parameters")
    arguments.add_cmdline_args(parser)
    parser.add_argument("--test_epochs", nargs="+", type=int, default=[])
    parser.add_argument("--save_epochs", nargs="+", type=int, default=[])
    parser.add_argument("--checkpoint_epochs", nargs="+", type=int, default=[])
    parser.add_argument("--start_checkpoint", type=str, default=None)
    parser.add_argument("--operator_path", action="store_true", help="drive the loop through the litegs_fused operator surface instead of the native executor")
    args = parser.parse_args(sys.argv[1:])
    lp, op, pp, dp = arguments.extract(args)
    training.start(lp, op, pp, dp, args.test_epochs, args.save_epochs, args.checkpoint_epochs, args.start_checkpoint, fused=not args.operator_path)
