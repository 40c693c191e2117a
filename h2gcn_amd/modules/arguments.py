"""Argument parser with a hook deque (reference ``h2gcn/modules/arguments.py:5-41``).

Plugins append callables to ``parser.function_hooks["argparse"]``; ``parse_args`` parses, creates the callback
deques in ``args.objects`` and runs the hooks in order (the dataset plugin registers with ``appendleft`` so that it
runs before the model plugin)."""
import argparse
from collections import deque


def create_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(add_help=False)
    parser.function_hooks = {"argparse": deque()}
    return parser


def parse_args(parser: argparse.ArgumentParser, argv=None):
    parser.add_argument("--verbose", "-v", action="store_true")
    parser.add_argument("--help", "-h", action="help")
    parser.add_argument("--exp_tags", default=[], nargs="+", dest="_exp_tags")
    args = parser.parse_args(argv)
    args.use_signac = False  # signac bookkeeping is out of scope; kept as an attribute the plugins may test
    args.objects = dict(function_hooks=parser.function_hooks)
    for name in ("pretrain_callbacks", "pre_epoch_callbacks", "post_epoch_callbacks", "post_train_callbacks"):
        args.objects[name] = deque()
    hooks = parser.function_hooks["argparse"]
    while hooks:
        hooks.popleft()(args)
    return args
