"""Plugin discovery shared by ``h2gcn_amd.models`` and ``h2gcn_amd.datasets``.

The reference selects a model module and a dataset-format module by name from the command line: the positional
argument's choices are the non-underscore modules of the package, and the chosen module's
``add_subparser_args(parser)`` is called immediately so that its flags exist before the real parse
(``h2gcn/models/__init__.py:16-31``, ``h2gcn/datasets/__init__.py:10-22``)."""
import contextlib
import importlib
import io
import pkgutil


def register_positional(parser, package, dest: str, help_text: str, announce: bool = False):
    names = sorted(m.name for m in pkgutil.iter_modules(package.__path__) if not m.name.startswith("_"))
    parser.add_argument(dest, choices=names, help=help_text)
    # peek at the command line (errors are expected while later positionals are still unregistered)
    with contextlib.redirect_stderr(io.StringIO()):
        try:
            chosen = getattr(parser.parse_known_args()[0], dest)
        except SystemExit:
            return None
    module = importlib.import_module(f"{package.__name__}.{chosen}")
    hook = getattr(module, "add_subparser_args", None)
    if hook is not None:
        hook(parser)
        if announce:
            print(f"Using model: {module}")
    return module
