"""Minimal host plumbing the entry point needs (argparse hook deque, early stopping, stats line, best-checkpoint).
Mirrors ``h2gcn/modules/`` only as far as ``run_experiments.py`` requires; signac, ptvsd, per-epoch TF
checkpoints and the gradient/degree monitors are out of scope."""
