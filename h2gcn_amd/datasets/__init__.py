"""Dataset-format plugins: each public module of this package is a selectable positional ``datafmt``
(reference ``h2gcn/datasets/__init__.py:10-22``)."""
import sys

from .._plugins import register_positional


def add_subparsers(parser):
    return register_positional(parser, sys.modules[__name__], "datafmt", "Dataset selected for experiment")
