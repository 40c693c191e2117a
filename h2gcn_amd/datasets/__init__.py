"""Dataset-format plugins (mirror of ``h2gcn/datasets/__init__.py:10-22``): every non-underscore module here is a
selectable positional ``datafmt`` and may expose ``add_subparser_args(parser)``."""
import contextlib
import importlib
import os
import pkgutil


def add_subparsers(parser):
    fmt_list = [m.name for m in pkgutil.iter_modules(path=__path__) if not m.name.startswith("_")]
    parser.add_argument("datafmt", choices=fmt_list, help="Dataset selected for experiment")
    try:
        with open(os.devnull, "w") as devnull, contextlib.redirect_stderr(devnull):
            known, _ = parser.parse_known_args()
    except SystemExit:
        return
    module = importlib.import_module("." + known.datafmt, package=__name__)
    if hasattr(module, "add_subparser_args"):
        module.add_subparser_args(parser)
