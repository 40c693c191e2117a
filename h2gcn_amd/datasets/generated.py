"""``generated`` dataset-format plugin: a graph written by the reference's generator
(``experiments/h2gcn/modules/graphgen.py:37-66``: ``<name>.graph`` + ``<name>.ally``, or ``<name>.gpickle.gz``).

Flags: ``--dataset`` (graph name = file prefix), ``--dataset_path`` (directory), ``--feature_dim`` /
``--feature_seed`` (class-conditional synthetic features; the generator stores none) and ``--split_seed``.  Use with
``--no_feature_normalize``, as the reference's syn-products configs do
(``experiments/h2gcn/configs/syn-products/h2gcn.json:3-6``)."""
from ._dataset import GeneratedGraphData

FLAGS = (
    ("--dataset", dict(type=str, required=True)),
    ("--dataset_path", dict(type=str, required=True, dest="_dataset_path")),
    ("--feature_dim", dict(type=int, default=100)),
    ("--feature_seed", dict(type=int, default=0)),
    ("--split_seed", dict(type=int, default=0)),
)


def load_dataset(args):
    dataset = GeneratedGraphData(args.dataset, args._dataset_path, feature_dim=args.feature_dim,
                                 feature_seed=args.feature_seed, split_seed=args.split_seed)
    args.objects["dataset"] = dataset
    print(f"===> Dataset loaded: {args.dataset}")


argparse_callback = load_dataset


def add_subparser_args(parser):
    group = parser.add_argument_group("Generator Format Data Arguments (datasets/generated.py)")
    for flag, kw in FLAGS:
        group.add_argument(flag, **kw)
    parser.function_hooks["argparse"].appendleft(load_dataset)
