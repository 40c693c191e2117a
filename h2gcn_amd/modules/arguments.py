"""Argument parser carrying a hook queue (reference ``h2gcn/modules/arguments.py:5-41``).

Contract kept from the reference: plugins register callables in ``parser.function_hooks["argparse"]`` (a deque:
``append`` = run later, ``appendleft`` = run first); after the command line is parsed, ``args.objects`` is created
with the four callback queues the epoch driver consumes, and the hooks run front to back, each receiving ``args``.
signac job bookkeeping (``--use_signac``) is not carried over."""
import argparse
from collections import deque

CALLBACK_QUEUES = ("pretrain_callbacks", "pre_epoch_callbacks", "post_epoch_callbacks", "post_train_callbacks")


def create_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(add_help=False)
    parser.function_hooks = {"argparse": deque()}
    return parser


def parse_args(parser: argparse.ArgumentParser, argv=None):
    common = parser.add_argument_group("Common arguments")
    common.add_argument("--verbose", "-v", action="store_true")
    common.add_argument("--help", "-h", action="help")
    common.add_argument("--exp_tags", nargs="+", default=[], dest="_exp_tags")
    args = parser.parse_args(argv)
    args.use_signac = False
    args.objects = {"function_hooks": parser.function_hooks, **{name: deque() for name in CALLBACK_QUEUES}}
    pending = parser.function_hooks["argparse"]
    while pending:
        hook = pending.popleft()
        hook(args)
    return args
