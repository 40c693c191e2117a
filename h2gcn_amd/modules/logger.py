"""Epoch statistics line + best-model snapshot (reference ``h2gcn/modules/logger.py``).

The stats line keeps the reference's format (``logger.py:87-90``).  The reference writes a TF checkpoint EVERY
epoch and deletes the previous non-best one (``H2GCN.py:147-155``); here only the best state is kept, in memory
(``state_dict`` clone), and optionally written once at the end -- per-epoch checkpoint I/O is pure overhead
(SURVEY.md §5)."""
import copy
from pathlib import Path

import torch

STATS_FORMAT = "    ".join([
    "Epoch: {epoch:04}", "Train Loss: {train_loss:9.6f}", "Train Acc: {train_acc:7.2%}",
    "Val Loss: {val_loss:9.6f}", "Val Acc: {val_acc:7.2%}", "Test Acc: {test_accuracy:7.2%}",
])


def add_subparser_args(parser):
    g = parser.add_argument_group("Logging arguments (modules/logger.py)")
    g.add_argument("--checkpoint_dir", type=str, default=None,
                   help="if set, the best model's state_dict is written there after training")
    g.add_argument("--json_stats", action="store_true", help="also print one JSON line per epoch")


class EpochStatsPrinter:
    def __init__(self, format_str=None):
        self.format_str = format_str or STATS_FORMAT

    def __call__(self, epoch, epoch_stats: dict):
        print(self.format_str.format(epoch=epoch, **{k: _scalar(v) for k, v in epoch_stats.items()}))

    def from_dict(self, epoch_stats: dict):
        print(self.format_str.format(**{k: _scalar(v) for k, v in epoch_stats.items()}))
        if epoch_stats.get("monitor"):
            print(epoch_stats["monitor"])


def _scalar(v):
    return v.item() if isinstance(v, torch.Tensor) and v.numel() == 1 else v


class BestSnapshot:
    """Keeps a copy of the best model/optimizer state (replaces save/remove/restore_ckpt, ``logger.py:58-79``)."""

    def __init__(self):
        self.state = None

    def save(self, model, optimizer):
        self.state = (copy.deepcopy(model.state_dict()), copy.deepcopy(optimizer.state_dict()))

    def restore(self, model, optimizer):
        if self.state is not None:
            model.load_state_dict(self.state[0])
            optimizer.load_state_dict(self.state[1])

    def write(self, directory, name="best.pt"):
        if self.state is not None and directory:
            Path(directory).mkdir(parents=True, exist_ok=True)
            torch.save({"model": self.state[0], "optimizer": self.state[1]}, Path(directory) / name)
