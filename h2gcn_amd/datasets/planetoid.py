"""``planetoid`` dataset-format plugin (reference ``h2gcn/datasets/planetoid.py:6-28``).

Flags: ``--dataset`` (file prefix, e.g. ``ind.cora``), ``--dataset_path`` (directory with the seven pickles and
``test.index``), ``--val_size`` (default 500; negative = everything that is neither train nor test) and
``--feature_configs`` (feature ablations).  The loading hook goes to the FRONT of the hook deque so that
``args.objects["dataset"]`` exists when the model plugin's hook runs."""
from ._dataset import PlanetoidData


def _blank_test_rows(dataset):
    feats = dataset.features.tolil()
    feats[dataset.test_mask, :] = 0
    dataset.features = feats.tocsr()


FEATURE_CONFIGS = {
    "no_test": _blank_test_rows,                            # hide the test nodes' features
    "identity": lambda dataset: dataset.set_identity_features(),  # structure only
}

FLAGS = (
    ("--dataset", dict(type=str, required=True)),
    ("--dataset_path", dict(type=str, required=True, dest="_dataset_path")),
    ("--val_size", dict(type=int, default=500)),
    ("--feature_configs", dict(choices=sorted(FEATURE_CONFIGS), nargs="*", default=[])),
)


def load_dataset(args):
    val_size = None if args.val_size < 0 else args.val_size
    args.val_size = val_size
    dataset = PlanetoidData(args.dataset, args._dataset_path, val_size=val_size)
    for name in args.feature_configs:
        FEATURE_CONFIGS[name](dataset)
    args.objects["dataset"] = dataset
    print(f"===> Dataset loaded: {args.dataset}")


argparse_callback = load_dataset


def add_subparser_args(parser):
    group = parser.add_argument_group("Planetoid Format Data Arguments (datasets/planetoid.py)")
    for flag, kw in FLAGS:
        group.add_argument(flag, **kw)
    parser.function_hooks["argparse"].appendleft(load_dataset)
