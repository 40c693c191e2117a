"""``planetoid`` dataset-format plugin (reference ``h2gcn/datasets/planetoid.py:6-28``): ``--dataset``,
``--dataset_path``, ``--val_size``, ``--feature_configs``; its hook is registered with ``appendleft`` so the
dataset exists before the model hook runs."""
from ._dataset import PlanetoidData


def add_subparser_args(parser):
    g = parser.add_argument_group("Planetoid Format Data Arguments (datasets/planetoid.py)")
    g.add_argument("--dataset", type=str, required=True)
    g.add_argument("--dataset_path", type=str, dest="_dataset_path", required=True)
    g.add_argument("--val_size", type=int, default=500)
    g.add_argument("--feature_configs", choices=["no_test", "identity"], nargs="*", default=[])
    parser.function_hooks["argparse"].appendleft(argparse_callback)


def argparse_callback(args):
    if args.val_size < 0:
        args.val_size = None
    dataset = PlanetoidData(args.dataset, args._dataset_path, val_size=args.val_size)
    for config in args.feature_configs:
        if config == "no_test":
            f = dataset.features.tolil()
            f[dataset.test_mask, :] = 0
            dataset.features = f.tocsr()
        elif config == "identity":
            dataset.set_identity_features()
    args.objects["dataset"] = dataset
    print(f"===> Dataset loaded: {args.dataset}")
