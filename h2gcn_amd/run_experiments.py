"""Entry point: ``python run_experiments.py <model> <datafmt> [flags]`` from inside the package directory -- how the
reference is launched (cwd ``h2gcn/``, ``experiments/h2gcn/experiments_workflow.py:301-318``) -- or, equivalently,
``python -m h2gcn_amd.run_experiments <model> <datafmt> [flags]`` from anywhere.

Same driver contract as the reference's ``h2gcn/run_experiments.py`` (``:7-12, 31-61``): positional ``model`` and
``datafmt`` plugins, ``--epochs`` (2000), ``--random_seed`` (123), plugin hooks fill ``args.objects`` with
``tensors`` / ``train_step`` / ``test_step`` / callback deques, then the epoch loop runs train + test steps and
the callbacks.  Example (planetoid files of Cora in ./data):

    python -m h2gcn_amd.run_experiments H2GCN planetoid --dataset ind.cora --dataset_path data --epochs 200
"""
import os
import time

if __package__ in (None, ""):  # run as a script (`python run_experiments.py ...`): make the package importable
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    __package__ = "h2gcn_amd"
    import h2gcn_amd  # noqa: F401

import torch
import torch.distributed as dist

from . import datasets, models
from .modules import arguments, logger


#: flags of the reference's command line that concern subsystems outside this build (signac bookkeeping, ptvsd,
#: TF memory growth, IPython, TF checkpoints naming, monitors).  They are accepted so that existing launch lines --
#: e.g. the ones ``experiments/h2gcn/experiments_workflow.py:301-318`` assembles -- keep working, and ignored.
COMPAT_FLAGS = (
    ("--debug", dict(action="store_true")),
    ("--use_full_gpu", dict(action="store_true", dest="_use_full_gpu")),
    ("--interactive", dict(action="store_true", dest="_interactive")),
    ("--use_signac", dict(action="store_true", dest="_compat_use_signac")),
    ("--signac_root", dict(default=None, dest="_signac_root")),
    ("--checkpoint_name", dict(type=str, default=None)),
    ("--message", dict(default=None)),
    ("--run_id", dict(default=None)),
    ("--deg_acc_monitor", dict(default=[], type=float, nargs="+")),
    ("--grad_monitor", dict(action="store_true")),
    ("--save_activations", dict(action="store_true")),
    ("--save_predictions", dict(nargs="*", default=True)),
)


def build_parser():
    parser = arguments.create_parser()
    parser.add_argument("--random_seed", type=int, default=123)
    g = parser.add_argument_group("Experiment arguments (run_experiments.py)")
    g.add_argument("--epochs", type=int, default=2000, help="(default: %(default)s)")
    c = parser.add_argument_group("Accepted for command-line compatibility with the reference, ignored")
    for flag, kw in COMPAT_FLAGS:
        c.add_argument(flag, **kw)
    return parser


def init_distributed():
    """One process per GPU when launched by ``python -m torch.distributed.run`` (WORLD_SIZE > 1): RCCL process
    group, rank r on GPU LOCAL_RANK, ``--device`` defaulted accordingly.  ``H2GCN_DIST_BACKEND=gloo`` with
    ``H2GCN_SHARE_GPU=1`` is the debugging mode the GPU test-suite uses to run two ranks on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or (dist.is_available() and dist.is_initialized()):
        return
    local = 0 if os.environ.get("H2GCN_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    backend = os.environ.get("H2GCN_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        from .partition import init_rccl_process_group
        init_rccl_process_group(torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    import sys

    if not any(a.startswith("--device") for a in sys.argv):
        sys.argv += ["--device", f"cuda:{local}"]


def main(argv=None):
    import sys

    if argv is not None:  # plugin discovery peeks at sys.argv through parse_known_args
        sys.argv = [sys.argv[0]] + list(argv)
    init_distributed()
    parser = build_parser()
    known, _ = parser.parse_known_args()
    if known.random_seed:
        torch.manual_seed(known.random_seed)
    models.add_subparsers(parser)
    datasets.add_subparsers(parser)
    logger.add_subparser_args(parser)
    args = arguments.parse_args(parser)

    run_epochs(args)
    return args


def run_epochs(args):
    """The epoch driver (reference ``run_experiments.py:44-61``): pre-train callbacks once; per epoch the
    pre-epoch callbacks, one train step and one test step (their dicts merged into ``args.objects["epoch_stats"]``)
    and the post-epoch callbacks; post-train callbacks once the (possibly early-stopped) last epoch is done."""
    obj = args.objects
    tensors = obj["tensors"]
    for callback in obj["pretrain_callbacks"]:
        callback(**tensors)
    started = time.perf_counter()
    stamps = []
    epoch = 0
    while epoch < args.epochs:  # args.epochs may be lowered by the early-stopping callback
        epoch += 1
        args.current_epoch = epoch
        for callback in obj["pre_epoch_callbacks"]:
            callback(epoch, args)
        obj["epoch_stats"] = {**obj["train_step"](**tensors), **obj["test_step"](**tensors)}
        for callback in obj["post_epoch_callbacks"]:
            callback(epoch, args)
        stamps.append(time.perf_counter())   # the post-epoch read-back has synchronised the device
    queue = obj["post_train_callbacks"]
    while queue:
        queue.popleft()(args)
    obj["wall_seconds"] = time.perf_counter() - started
    if epoch > 0 and (not dist.is_initialized() or dist.get_rank() == 0):
        skip = min(6, epoch // 2)   # first epochs: module loads, allocator growth, hipGraph capture
        steady = (stamps[-1] - stamps[skip - 1]) / (epoch - skip) * 1e3 if epoch > skip >= 1 else obj["wall_seconds"] / epoch * 1e3
        obj["steady_ms_per_epoch"] = steady
        print(f"Epoch loop: {obj['wall_seconds']:.2f} s for {epoch} epochs; steady state {steady:.1f} ms per epoch over the last "
              f"{epoch - skip} (one training step + one evaluation pass)")


if __name__ == "__main__":
    main()
