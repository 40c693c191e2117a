"""``synthetic`` dataset-format plugin: the BASELINE.json shapes (``h2gcn_amd.synth.SHAPES``: arxiv, products, lowdeg ...)
through the reference-style entry point --

    python run_experiments.py H2GCN synthetic --shape products --hidden 64 --epochs 5 --no_feature_normalize

-- with the operands GENERATED ON THE DEVICE (the reference has no input of these sizes: its own generator is an O(n^2)
process, ``experiments/h2gcn/modules/graphgen.py``).  As BASELINE configs[3] prescribes, the 2-hop matrix is SUPPLIED
(a second synthetic CSR), not derived -- the exact 2-hop ring of a products-like graph has > 1e10 nonzeros -- so
``--adj_nhood`` is not consulted; values are row-normalised (``1 / deg``).  Features are dense ``U[-1, 1)`` (``--feature_dim``,
default 100 as ogbn-products), labels uniform over ``--classes`` (default 47), masks a seeded ``train / val / test`` split.
Row-partitioned runs (``torch.distributed.run``) generate only their shard (equal row blocks: the synthetic rows are in
random order, so the nonzeros balance by themselves)."""
import numpy as np

FLAGS = (
    ("--shape", dict(type=str, default="products", help="one of h2gcn_amd.synth.SHAPES")),
    ("--feature_dim", dict(type=int, default=100)),
    ("--classes", dict(type=int, default=47)),
    ("--train_frac", dict(type=float, default=0.1)),
    ("--val_frac", dict(type=float, default=0.1)),
    ("--split_seed", dict(type=int, default=0)),
)


class SyntheticShapeData:
    """Duck-types what the model plugin reads from ``args.objects["dataset"]`` (``num_samples``, ``num_labels``,
    ``row_normalize_features``, ``adj_remove_eye``, ``get_tensors``); nothing proportional to the graph lives on the host."""

    def __init__(self, shape: str, feature_dim: int, classes: int, train_frac: float, val_frac: float, split_seed: int):
        from .. import synth
        if shape not in synth.SHAPES:
            raise ValueError(f"unknown shape {shape!r}; choose from {sorted(synth.SHAPES)}")
        self.shape, self.cfg = shape, synth.SHAPES[shape]
        self.feature_dim, self.classes = int(feature_dim), int(classes)
        self.train_frac, self.val_frac, self.split_seed = float(train_frac), float(val_frac), int(split_seed)
        self.non_valid_samples = set()

    @property
    def num_samples(self) -> int:
        return int(self.cfg["n"])

    @property
    def num_labels(self) -> int:
        return self.classes

    def row_normalize_features(self):   # dense synthetic features: nothing to normalise (use --no_feature_normalize)
        pass

    def adj_remove_eye(self):           # generated without self loops
        pass

    def get_tensors(self, device, adj_norm_hops=None, norm=None, build_transpose: bool = True, shard=None, **_):
        import torch

        from .. import synth
        from ..hops import HopPlan
        from ..partition import RowPartition, ShardedHops

        n = self.num_samples
        rank, world = shard if shard is not None else (0, 1)
        part = RowPartition.equal(n, world)
        r0, r1 = part.rows(rank)
        seeds = (synth.SEED_A1, synth.SEED_A2)
        degs = synth.hop_degrees(self.cfg, seeds)
        csr = [synth.synth_hop_rows(degs[k], n, seeds[k], r0, r1, device) for k in range(2)]
        plan = HopPlan([c[0] for c in csr], [c[1] for c in csr], [c[2] for c in csr], n, build_transpose=build_transpose)
        t = {"adj": None, "partition": part}
        t["adj_hops"] = plan if shard is None else ShardedHops(plan, n, device, partition=part)
        t["features"] = synth.synth_features(self.feature_dim, synth.SEED_X, r0, r1, device)
        g = torch.Generator(device="cpu").manual_seed(1234 + self.split_seed)
        labels_all = torch.randint(0, self.classes, (n,), generator=g)
        u = torch.rand(n, generator=g)
        labels = labels_all[r0:r1].to(device)
        y_all = torch.nn.functional.one_hot(labels, self.classes).to(torch.float32)
        masks = {"train_mask": u < self.train_frac, "val_mask": (u >= self.train_frac) & (u < self.train_frac + self.val_frac),
                 "test_mask": u >= self.train_frac + self.val_frac}
        t["labels"], t["y_all"] = labels, y_all
        for name, m in masks.items():
            m = m[r0:r1].to(device)
            t[name] = m
            t["y_" + name.split("_")[0]] = y_all * m[:, None]
        return t


def load_dataset(args):
    args.objects["dataset"] = SyntheticShapeData(args.shape, args.feature_dim, args.classes, args.train_frac, args.val_frac,
                                                 args.split_seed)
    cfg = args.objects["dataset"].cfg
    print(f"===> Dataset: synthetic shape {args.shape} (|V| = {cfg['n']}, {cfg['nnz_per_hop']} nonzeros per hop, generated on the device)")


argparse_callback = load_dataset


def add_subparser_args(parser):
    group = parser.add_argument_group("Synthetic BASELINE-shape Data Arguments (datasets/synthetic.py)")
    for flag, kw in FLAGS:
        group.add_argument(flag, **kw)
    parser.function_hooks["argparse"].appendleft(load_dataset)
