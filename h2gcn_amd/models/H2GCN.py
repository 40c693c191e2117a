"""``H2GCN`` model plugin: the caller of the hot path, re-expressed in torch on top of ``GCNLayer``.

Mirror of ``h2gcn/models/H2GCN.py``: CLI flags (``:9-30``), the argparse hook that preprocesses the data and builds
the model (``:33-54``), step closures (``:66-127``), best-validation bookkeeping (``:136-195``) and the generic
layer interpreter driven by the parsed ``--network_setup`` (``:209-346``), with the masked cross-entropy + L2 loss
(``:363-367``).  Differences by design: tensors live on one MI355X; ``adj_hops`` and the sparse features are
``HopPlan`` device operands; every ``G`` layer is ONE fused 1+2-hop HIP launch writing ``[N, H, d]`` (so the
following ``V`` flatten is a view, not a copy); only the best model state is kept (in memory) instead of a TF
checkpoint per epoch; signac bookkeeping and the attention / experimental layer kinds are not carried over.
"""
from __future__ import annotations

import functools
import operator
import os

import torch
import torch.distributed as dist

from .. import layers as L
from ..modules import controller, logger
from . import Layer, parse_network_setup
from .. import metrics as fused_metrics
from ._metrics import masked_accuracy, masked_softmax_cross_entropy


def add_subparser_args(parser):
    g = parser.add_argument_group("H2GCN Model Arguments (H2GCN.py)")
    g.add_argument("--network_setup", type=str, default="M64-R-T1-G-V-T2-G-V-C1-C2-D0.5-MO",
                   help="Default to H2GCN-2 (%(default)s)")
    g.add_argument("--dropout", type=float, default=0.5, help="Default dropout rate")
    g.add_argument("--hidden", type=int, default=64)
    g.add_argument("--adj_nhood", default=["1", "2"], type=str, nargs="+")
    g.add_argument("--optimizer", type=str, default="adam", help="(default: %(default)s)")
    g.add_argument("--lr", type=float, default=0.01, help="(default: %(default)s)")
    g.add_argument("--l2_regularize_weight", type=float, default=5e-4, help="(default: %(default)s)")
    g.add_argument("--early_stopping", type=int, default=0,
                   help="Number of epochs used to decide early stopping (0 to disable) (default: %(default)s)")
    g.add_argument("--best_val_criteria", choices=["val_acc", "val_loss"], default="val_acc")
    g.add_argument("--no_feature_normalize", action="store_true")
    g.add_argument("--adj_norm", choices=["sym", "rw"], default="sym",
                   help="hop normalisation: sym = D^-1/2 A D^-1/2 (reference default), rw = D^-1 A")
    g.add_argument("--no_propagation_reuse", action="store_true",
                   help="recompute the propagation in every training forward instead of adopting the buffer the preceding "
                        "evaluation produced (same results either way; see H2GCN.reuse_propagation)")
    g.add_argument("--no_fused_classifier", action="store_true",
                   help="run `D<rate>` followed by a dense layer as the stock dropout + matmul pair instead of the library's "
                        "one-pass dropout+Dense kernels (csrc/classifier.hip)")
    g.add_argument("--sparse_dropout_at_eval", action="store_true",
                   help="reproduce the reference's SparseDropout, which Keras never switches off (it drops sparse feature "
                        "values during evaluation as well); default: inactive in evaluation like every other dropout")
    g.add_argument("--device", type=str, default="cuda:0", dest="_device")
    g.add_argument("--no_hipgraph", action="store_true", dest="_no_hipgraph",
                   help="run every step eagerly instead of replaying captured hipGraphs")
    parser.function_hooks["argparse"].append(argparse_callback)


def argparse_callback(args):
    dataset = args.objects["dataset"]
    layer_setups = parse_network_setup(args.network_setup, dataset.num_labels, _dense_units=args.hidden,
                                       _dropout_rate=args.dropout, parse_preprocessing=True)
    uses_hops = any(kind == Layer.GCN for kind, _ in layer_setups)
    preprocessing_data(args, adj_norm_hops=args.adj_nhood if uses_hops else None)
    initialize_model(args, layer_setups, args.optimizer, args.lr, args.l2_regularize_weight, args.early_stopping)


def preprocessing_data(args, adj_norm_hops=None):
    """feature row-normalisation (unless disabled) -> drop self loops -> device operands
    (reference ``preprocessing_data``, ``:46-54``)."""
    dataset = args.objects["dataset"]
    if not torch.cuda.is_available():
        raise RuntimeError("h2gcn_amd runs the propagation on an MI355X; no GPU is visible and there is no CPU fallback")
    if not args.no_feature_normalize:
        dataset.row_normalize_features()
    dataset.adj_remove_eye()
    shard = (dist.get_rank(), dist.get_world_size()) if _is_sharded() else None
    args.objects["tensors"] = dataset.get_tensors(torch.device(args._device), adj_norm_hops=adj_norm_hops,
                                                  norm=args.adj_norm, shard=shard)


def _is_sharded() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def make_optimizer(name: str, params, lr: float, capturable: bool = False):
    name = name.lower()
    if name == "adam":
        # Keras Adam with its defaults (beta 0.9 / 0.999, epsilon 1e-7) and Keras' placement of epsilon -- on the uncorrected
        # sqrt(v), not torch's bias-corrected one -- as ONE kernel launch per step for all parameters (`h2gcn_amd/optim.py`;
        # stock multi-tensor Adam: ~10 launches, 141 us eager / 99 us replayed for H2GCN-2's four tensors); always capturable
        from ..optim import KerasAdam
        return KerasAdam(params, lr=lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7)
    if name == "sgd":
        return torch.optim.SGD(params, lr=lr)
    if name == "rmsprop":   # Keras / TensorFlow arithmetic (epsilon inside the square root), see optim.KerasRMSprop
        from ..optim import KerasRMSprop
        return KerasRMSprop(params, lr=lr, rho=0.9, epsilon=1e-7)
    raise ValueError(f"unsupported optimizer {name!r}")


def initialize_model(args, layer_setups, optimizer, lr, l2_regularize_weight, early_stopping):
    tensors = args.objects["tensors"]
    device = torch.device(args._device)
    feats = tensors["features"]
    dense_features = isinstance(feats, torch.Tensor)  # dense-ish features arrive as a matrix (GEMM path)
    model = H2GCN(layer_setups, input_dim=(feats.shape[1] if dense_features else feats.n_cols),
                  n_hops=(tensors["adj_hops"].n_hops if tensors["adj_hops"] is not None else 0),
                  sparse_input=not dense_features, l2_regularize_weight=l2_regularize_weight,
                  sparse_dropout_at_eval=getattr(args, "sparse_dropout_at_eval", False),
                  fused_classifier=not getattr(args, "no_fused_classifier", False),
                  reuse_propagation=not getattr(args, "no_propagation_reuse", False)).to(device)
    sharded = _is_sharded()
    if sharded:  # replicas must start identical whatever the seeding on each rank (e.g. --random_seed 0)
        for p_ in model.parameters():
            dist.broadcast(p_.data, src=0)
    hops_obj = tensors.get("adj_hops")
    # full-batch steps are replayed as hipGraphs; row-partitioned runs too when every exchange of the step goes through the
    # library's copy-kernel IPC path (H2GCN_EXCHANGE=ipc_kernel: device-side sequence numbers) -- RCCL / gloo collectives
    # are not captured here
    use_graphs = (not getattr(args, "_no_hipgraph", False) and optimizer.lower() == "adam"
                  and (not sharded or (bool(getattr(hops_obj, "capturable", False))
                                       and os.environ.get("H2GCN_SHARDED_HIPGRAPH", "1") != "0")))
    # replay pays on launch-bound graphs (Cora: 1.40 -> 0.61 ms per epoch); at the products shape an epoch is 94 ms of
    # bandwidth-bound kernels either way and the capture itself costs ~2 s of start-up: skip it beyond 2e7 nonzeros
    plan_obj = getattr(hops_obj, "plan", hops_obj)
    if use_graphs and plan_obj is not None and sum(getattr(plan_obj, "nnz", [0])) > int(os.environ.get("H2GCN_HIPGRAPH_MAX_NNZ", "20000000")):
        use_graphs = False
    optimizer = make_optimizer(optimizer, model.parameters(), lr, capturable=use_graphs)
    snapshot = logger.BestSnapshot()
    # The keras l2 penalty of the dense kernels (`:239-240, 247-248`): with KerasAdam on the GPU its GRADIENT is folded into the
    # optimizer's one launch (same roundings as autograd's accumulate) and its VALUE is one more launch -- instead of a pow, a
    # reduction, a multiply and an add per kernel per loss plus the backward of all that (on Cora ~25 of an epoch's 65 launches,
    # profiles/r04_cora_epoch_kernels.txt).  Other optimizers / CPU runs keep the penalty inside the autograd graph.
    from ..optim import KerasAdam
    from .. import _capi
    fused_l2 = (isinstance(optimizer, KerasAdam) and model.l2 > 0 and bool(model.regularized) and device.type == "cuda"
                and os.environ.get("H2GCN_FUSED_L2", "1") != "0" and _capi.has("h2gcn_adam_keras_l2_f32"))
    if fused_l2:
        optimizer.set_l2([layer.kernel for layer in model.regularized], model.l2)
    model.fused_l2 = fused_l2
    penalty = model.regularization_value if fused_l2 else model.regularization_loss

    def train_step(adj, adj_hops, features, y_train, train_mask, **kwargs):
        model.train()
        optimizer.zero_grad(set_to_none=True)
        predictions = model(adj, features, adj_hops)
        if fused_l2:
            data_loss = model.data_loss(predictions, y_train, train_mask)
            data_loss.backward()
            train_loss = data_loss.detach() + penalty()    # the penalty of the weights this step STARTED from, as in the reference
        else:
            train_loss = model.loss(predictions, y_train, train_mask)
            train_loss.backward()
        model.restore_sparse_inputs()   # SparseDropout pointed the shared feature operand at dropped values
        optimizer.step()
        model.note_update()
        return dict(train_loss=train_loss.detach())

    # The three masked accuracies and two masked losses of an evaluation (reference ``test_step``, ``:77-107``) share
    # one read of the logits: on the GPU ONE launch of the library's masked-metrics kernel (``h2gcn_amd/metrics.py``) over the
    # three (labels, mask / sum(mask)) sets; elsewhere one argmax / one log-softmax and two small mat-vecs.  Same
    # quantities as ``masked_accuracy`` / ``masked_softmax_cross_entropy`` per mask (summation order aside).
    eval_cache = {}

    def _eval_weights(y_train, train_mask, y_val, val_mask, y_test, test_mask):
        key = (train_mask.data_ptr(), val_mask.data_ptr(), test_mask.data_ptr(), y_val.data_ptr(), y_test.data_ptr())
        if eval_cache.get("key") != key:
            masks = torch.stack([m.to(torch.float32) / m.to(torch.float32).sum() for m in (train_mask, val_mask, test_mask)])
            eval_cache.update(key=key, w_acc=masks,
                              labels_acc=torch.stack([y.argmax(dim=1) for y in (y_train, y_val, y_test)]),
                              w_loss=torch.stack([masks[1], masks[2]]), y_loss=torch.stack([y_val, y_test]))
        return eval_cache

    @torch.no_grad()
    def test_step(adj, adj_hops, features, y_train, train_mask, y_val, val_mask, y_test, test_mask, **kwargs):
        model.eval()
        predictions = model(adj, features, adj_hops)
        c = _eval_weights(y_train, train_mask, y_val, val_mask, y_test, test_mask)
        if fused_metrics.supported(predictions):
            loss3, acc = fused_metrics.masked_metrics(predictions, [y_train, y_val, y_test], list(c["w_acc"].unbind(0)))
            loss = loss3[1:]
        else:
            correct = (predictions.argmax(dim=1).unsqueeze(0) == c["labels_acc"]).to(torch.float32)      # [3, N]
            acc = (correct * c["w_acc"]).sum(dim=1)                                                       # train / val / test
            nll = -(c["y_loss"] * torch.log_softmax(predictions, dim=1).unsqueeze(0)).sum(dim=2)          # [2, N]
            loss = (nll * c["w_loss"]).sum(dim=1)                                                         # val / test
        return dict(
            train_acc=acc[0], val_acc=acc[1], test_accuracy=acc[2],
            val_loss=loss[0] + penalty(),                                            # includes the L2 term (:100)
            test_loss=loss[1],                                                       # does not (:101-102)
            monitor=dict(),
        )

    @torch.no_grad()
    def predict_step(adj, adj_hops, features, **kwargs):
        model.eval()
        return model(adj, features, adj_hops)

    @torch.no_grad()
    def embed_step(adj, adj_hops, features, **kwargs):
        model.eval()
        return model.get_embeddings(adj, features, adj_hops)

    if sharded:
        train_step, test_step = _sharded_steps(model, optimizer)
    if use_graphs:
        train_step, test_step = _GraphedSteps(train_step, test_step, optimizer, device,
                                              prepare=getattr(hops_obj, "prepare_capture", None) if sharded else None,
                                              refresh_eval=model.reuse_propagation).wrap()

    stats_printer = logger.EpochStatsPrinter()
    args.objects["statsPrinter"] = stats_printer
    args.objects["best_val_stats"] = None
    args.objects["early_stopping"] = controller.SlidingMeanEarlyStopping(early_stopping)

    def post_epoch_callback(epoch, args):
        raw = args.objects["epoch_stats"]
        names = [k for k, v in raw.items() if isinstance(v, torch.Tensor)]
        values = torch.stack([raw[k].detach().float().reshape(()) for k in names]).tolist()  # ONE device->host sync
        stats = dict(raw)
        stats.update(zip(names, values))
        args.objects["epoch_stats"] = stats
        hops_obj = args.objects["tensors"].get("adj_hops")
        if hasattr(hops_obj, "check"):   # the read-back above synchronised: the epoch's exchanges are all accounted for
            hops_obj.check()
        if not _is_sharded() or dist.get_rank() == 0:
            stats_printer(epoch, stats)
            if getattr(args, "json_stats", False):
                import json
                print(json.dumps({"epoch": epoch, **{k: v for k, v in stats.items() if isinstance(v, (int, float))}}))
        if args.objects["early_stopping"](stats["val_loss"]):
            print("Early stopping...")
            args.epochs = epoch
        better = operator.ge if args.best_val_criteria == "val_acc" else operator.le
        best = args.objects["best_val_stats"]
        if best is None or better(stats[args.best_val_criteria], best[args.best_val_criteria]):
            args.objects["best_val_stats"] = dict(stats, epoch=epoch)
            snapshot.save(model, optimizer)

    def post_train_callback(args):
        snapshot.restore(model, optimizer)
        stats = args.objects["test_step"](**args.objects["tensors"])
        if args.objects["best_val_stats"] is None:  # --epochs 0: report the untrained model
            raw = {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in stats.items()}
            args.objects["best_val_stats"] = dict(raw, epoch=0, train_loss=float("nan"))
        args.objects["best_val_stats"]["monitor"] = stats["monitor"]
        if not _is_sharded() or dist.get_rank() == 0:
            print("Restoring the best performance model")
            print("Best performance:")
            stats_printer.from_dict(args.objects["best_val_stats"])
            _write_results(args)
        if not _is_sharded() or dist.get_rank() == 0:
            snapshot.write(getattr(args, "checkpoint_dir", None))
        hops_obj = args.objects["tensors"].get("adj_hops")
        if hasattr(hops_obj, "close"):   # collective: release IPC-exported exchange buffers (nobody is pulling any more)
            torch.cuda.synchronize()
            hops_obj.close()

    args.objects.update(model=model, optimizer=optimizer, checkpoint=snapshot, train_step=train_step,
                        test_step=test_step, predict_step=predict_step, embed_step=embed_step)
    args.objects["post_epoch_callbacks"].append(post_epoch_callback)
    args.objects["post_train_callbacks"].append(post_train_callback)


def _write_results(args):
    """``results.json`` with the best-validation statistics -- the reference writes it into the signac job
    (``H2GCN.py:186-195``); here it goes to ``--signac_root`` or ``--checkpoint_dir`` when one is given."""
    import json
    from pathlib import Path

    target = getattr(args, "_signac_root", None) or getattr(args, "checkpoint_dir", None)
    if not target:
        return
    Path(target).mkdir(parents=True, exist_ok=True)
    record = {k: (v if isinstance(v, (int, float, str)) else str(v)) for k, v in args.objects["best_val_stats"].items()}
    (Path(target) / "results.json").write_text(json.dumps(record))


def _sharded_steps(model, optimizer):
    """Step closures of a row-partitioned run (one process per GPU).  Every rank holds the rows ``[r0, r1)`` of the
    features / hop matrices / labels and a full replica of the (small) dense kernels:

    * forward: ``SparseDense`` and everything after the propagation are row-local; each ``G`` layer all-gathers the
      embedding over RCCL (feature-chunk pipelined) and aggregates its row block;
    * loss: ``sum_local(CE_i * mask_i) / sum_global(mask)`` + ``L2 / world`` per rank, so that the SUM over ranks is
      the reference's loss (masked mean + L2, ``H2GCN.py:363-367``);
    * backward: each ``G`` layer's adjoint yields a full-height contribution that is reduce-scattered to the row
      owners (``partition.sharded_hop_spmm``); kernel gradients are summed over ranks (all-reduce), then every
      rank applies the same Adam update -- replicas stay bit-identical."""
    world = dist.get_world_size()
    mask_sums = {}

    def sum_over_ranks(vec, adj_hops):
        """Sum of a small fp32 vector over the ranks: through the library's IPC exchange when the shard's exchanges are
        capturable (no torch.distributed call inside the step, so it can be replayed as a hipGraph), else RCCL / gloo."""
        if getattr(adj_hops, "capturable", False):
            return adj_hops.all_reduce_small(vec.contiguous())
        vec = vec.detach().clone()
        dist.all_reduce(vec)
        return vec

    def global_mask_sum(mask):
        """sum_global(mask): the masks are static, so this is computed once (eagerly, during the warm-up epochs)."""
        key = (mask.data_ptr(), mask.numel())
        if key not in mask_sums:
            t = mask.to(torch.float32).sum().reshape(1)
            dist.all_reduce(t)
            mask_sums[key] = t.reshape(())
        return mask_sums[key]

    row_weights = {}

    def global_weights(mask):
        """mask / sum_global(mask) as fp32 row weights (static: computed once per mask)."""
        key = (mask.data_ptr(), mask.numel())
        if key not in row_weights:
            row_weights[key] = mask.to(torch.float32) / global_mask_sum(mask)
        return row_weights[key]

    def partial_ce(preds, labels, mask):
        if fused_metrics.supported(preds):
            return fused_metrics.masked_cross_entropy(preds, labels, global_weights(mask))
        m = mask.to(torch.float32)
        ce = -(labels * torch.log_softmax(preds, dim=1)).sum(dim=1)
        return (ce * m).sum() / global_mask_sum(mask)

    def partial_acc(preds, labels, mask):
        m = mask.to(torch.float32)
        correct = (preds.argmax(dim=1) == labels.argmax(dim=1)).to(torch.float32)
        return (correct * m).sum() / global_mask_sum(mask)

    def exchange_ok(adj_hops):
        # a peer that stalled beyond the exchange's time limit (or died) has left NaN-poisoned shards behind: stop here
        # instead of training on them (reads a host-mapped word per exchange; covers every step issued so far)
        if hasattr(adj_hops, "check"):
            adj_hops.check()

    def train_step(adj, adj_hops, features, y_train, train_mask, **kwargs):
        exchange_ok(adj_hops)
        model.train()
        optimizer.zero_grad(set_to_none=True)
        predictions = model(adj, features, adj_hops)
        ce = partial_ce(predictions, y_train, train_mask)
        if getattr(model, "fused_l2", False):     # the penalty's gradient rides in the optimizer step (KerasAdam.set_l2), after the
            reg = model.regularization_value()    # all-reduce of the data gradients: every replica adds the same 2 * l2 * w
            ce.backward()
        else:
            reg = model.regularization_loss()
            (ce + reg / world).backward()
        model.restore_sparse_inputs()
        # ONE exchange for every dense-kernel gradient and the loss scalar: flatten, sum over ranks (identical order on
        # every rank: the replicas stay bit-identical), scatter back
        params = [p for p in model.parameters() if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params] + [ce.detach().reshape(1)])
        total = sum_over_ranks(flat, adj_hops)
        pos = 0
        for p in params:
            p.grad.copy_(total[pos:pos + p.numel()].view_as(p.grad))
            pos += p.numel()
        optimizer.step()
        model.note_update()
        return dict(train_loss=total[pos] + reg.detach())

    @torch.no_grad()
    def test_step(adj, adj_hops, features, y_train, train_mask, y_val, val_mask, y_test, test_mask, **kwargs):
        exchange_ok(adj_hops)
        model.eval()
        predictions = model(adj, features, adj_hops)
        reg = model.regularization_value() if getattr(model, "fused_l2", False) else model.regularization_loss()
        if fused_metrics.supported(predictions):   # one pass over the local logits for all five quantities
            loss3, acc3 = fused_metrics.masked_metrics(predictions, [y_train, y_val, y_test],
                                                       [global_weights(m) for m in (train_mask, val_mask, test_mask)])
            parts = torch.cat([acc3, loss3[1:]])
        else:
            parts = torch.stack([partial_acc(predictions, y_train, train_mask), partial_acc(predictions, y_val, val_mask),
                                 partial_acc(predictions, y_test, test_mask), partial_ce(predictions, y_val, val_mask),
                                 partial_ce(predictions, y_test, test_mask)])
        parts = sum_over_ranks(parts, adj_hops)
        return dict(train_acc=parts[0], val_acc=parts[1], test_accuracy=parts[2], val_loss=parts[3] + reg,
                    test_loss=parts[4], monitor=dict())

    return train_step, test_step


class _GraphedSteps:
    """Full-batch training is the same launch sequence every epoch (static adjacency, static shapes), and on small
    graphs (Cora: ~90 short kernels per epoch) it is launch-bound.  After ``WARMUP`` eager epochs -- which also
    populate the plan's per-hop-mask caches, the only allocations the HIP library ever makes -- the train step and
    the test step are each captured into a hipGraph (``torch.cuda.CUDAGraph``; the C-ABI launches go to the
    capturing stream like any other kernel) and replayed.  Outputs are static tensors; the eager closures remain
    the fallback if capture fails."""

    WARMUP = 3

    def __init__(self, train_step, test_step, optimizer, device, prepare=None, refresh_eval=False):
        self.eager_train, self.eager_test = train_step, test_step
        self.optimizer, self.device = optimizer, device
        # the captured training step may ADOPT the propagation the evaluation left behind (H2GCN.reuse_propagation): a
        # training replay that does not follow an evaluation replays the evaluation first, so that what it adopts is current
        self.refresh_eval, self.eval_is_current = bool(refresh_eval), False
        self.prepare = prepare   # called (device idle) right before each capture: row-partitioned runs reset the
                                 # exchange objects' cross-step event dependencies there
        self.calls = 0
        self.train_graph = self.test_graph = None
        self.train_out = self.test_out = None
        self.failed = False

    def _capture(self, tensors):
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):  # one more eager pass on a side stream, as torch's capture recipe asks;
            eager_out = self.eager_train(**tensors)  # it is also this epoch's real update
            self.eager_test(**tensors)
        torch.cuda.current_stream(self.device).wait_stream(side)
        self.pending_eager_out = eager_out  # if capture fails below, this epoch's update has already happened
        self.optimizer.zero_grad(set_to_none=True)
        if self.prepare is not None:
            self.prepare()
        g_train = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_train):
            train_out = self.eager_train(**tensors)
        if self.prepare is not None:
            self.prepare()
        g_test = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_test):
            test_out = self.eager_test(**tensors)
        if self.prepare is not None:
            self.prepare()
        self.train_graph, self.test_graph, self.train_out, self.test_out = g_train, g_test, train_out, test_out
        self.eval_is_current = True   # the eager evaluation above ran after the eager update
        return eager_out

    def wrap(self):
        def train_step(**tensors):
            self.calls += 1
            if self.failed or self.calls <= self.WARMUP:
                return self.eager_train(**tensors)
            if self.train_graph is None:
                try:
                    return self._capture(tensors)  # capture records, the eager pass inside it did the update
                except Exception as e:  # pragma: no cover - depends on the runtime
                    print(f"hipGraph capture unavailable ({type(e).__name__}: {e}); continuing eagerly")
                    self.failed, self.train_graph, self.test_graph = True, None, None
                    torch.cuda.synchronize()
                    done = getattr(self, "pending_eager_out", None)
                    self.pending_eager_out = None
                    # the eager pass inside _capture already applied this epoch's update: do not step twice
                    return done if done is not None else self.eager_train(**tensors)
            if self.refresh_eval and not self.eval_is_current:
                self.test_graph.replay()
            self.train_graph.replay()
            self.eval_is_current = False
            return dict(self.train_out)

        def test_step(**tensors):
            if self.failed or self.test_graph is None:
                return self.eager_test(**tensors)
            self.test_graph.replay()
            self.eval_is_current = True
            return dict(self.test_out)

        return train_step, test_step


class Dense(torch.nn.Module):
    """keras ``Dense``: ``x @ kernel (+ bias)``, glorot-uniform kernel ``[in, out]``."""

    def __init__(self, input_dim: int, units: int, use_bias: bool):
        super().__init__()
        self.kernel = torch.nn.Parameter(torch.empty(input_dim, units))
        torch.nn.init.xavier_uniform_(self.kernel)
        self.bias = torch.nn.Parameter(torch.zeros(units)) if use_bias else None

    def forward(self, x):
        y = x @ self.kernel
        return y if self.bias is None else y + self.bias


class _SparseToDense(torch.nn.Module):
    """``I`` token: the sparse feature operand as a dense matrix (reference ``tf.sparse.to_dense``, ``:263-265``)."""

    def forward(self, plan):
        if isinstance(plan, torch.Tensor):  # dense-ish features are handed over as a matrix already
            return plan
        csr = torch.sparse_csr_tensor(plan.rowptr[0], plan.colidx[0].to(torch.int64), plan.vals[0],
                                      size=(plan.n_rows, plan.n_cols))
        return csr.to_dense()


class H2GCN(torch.nn.Module):
    """Generic interpreter of a parsed network setup (reference ``H2GCN(keras.Model)``, ``:209-346``).

    Dispatch kinds (``:314-325``): concat/slice layers receive the tag store, ``G`` layers receive
    ``(adjhops, inputs)``, everything else ``(inputs)``; an output tagged ``T<name>`` is stored for later
    concats (``:339-341``).  Feature widths are tracked statically (keras builds lazily)."""

    def __init__(self, layer_setups, input_dim: int, n_hops: int = 2, sparse_input: bool = True,
                 l2_regularize_weight: float = 0.0, sparse_dropout_at_eval: bool = False, fused_classifier: bool = True,
                 reuse_propagation: bool = True):
        super().__init__()
        self.l2 = float(l2_regularize_weight)
        self.layer_objs = torch.nn.ModuleList()
        self.kinds = []
        self.tags = {}
        self.graph_hops_inds, self.concat_inds = set(), set()
        self.embedding_ind = self.output_ind = None
        self.regularized = []
        width = int(input_dim)       # width of the running activation (per node)
        tag_width = {}
        pending_hops = None          # set after a G layer until the next V: activation is [N, H, width]
        setups = list(layer_setups)
        fuse_relu_into = None        # index of a SparseDense whose following R layer became its store epilogue
        pending_dropout = None       # rate of a dense-input D layer that the next dense layer absorbs (DropoutDense)
        for pos, (kind, conf) in enumerate(setups):
            conf = dict(conf)
            tag = conf.pop("tag", None)
            ind = len(self.layer_objs)
            if kind == Layer.DENSE:
                if conf.get("isEmbedding"):
                    self.embedding_ind = ind
                if conf.get("beginOutput"):
                    self.output_ind = ind
                if sparse_input:
                    # `M64-R`: the ReLU becomes the store epilogue of the sparse product (one pass over the embedding
                    # instead of two) unless the pre-activation itself is observable (tagged / embedding / supervised)
                    nxt = setups[pos + 1][0] if pos + 1 < len(setups) else None
                    fuse = nxt == Layer.RELU and tag is None and not conf.get("isEmbedding") and not conf.get("supervised")
                    layer = L.SparseDense(width, conf["units"], use_bias=conf["use_bias"], activation="relu" if fuse else None)
                    fuse_relu_into = ind if fuse else None
                    sparse_input = False
                elif pending_dropout is not None or (fused_classifier and conf["units"] <= 64):
                    # `D0.5-MO`: dropout + dense in one pass over the concat buffer (mask drawn inside the product kernels).
                    # A dense layer without a dropout in front runs on the same kernels with the mask off: its weight
                    # gradient X^T G is a reduction over all N rows into a tiny [K, units] result, which general GEMM tiles
                    # handle badly (products shape, [N,100]^T [N,64]: 3.3 ms stock, 0.5 ms here)
                    layer = L.DropoutDense(width, conf["units"], conf["use_bias"], pending_dropout or 0.0,
                                           seed=torch.initial_seed() + 0x632BE59BD9B4E019 * (ind + 1))   # one mask stream per layer
                    pending_dropout = None
                else:
                    layer = Dense(width, conf["units"], conf["use_bias"])
                self.regularized.append(layer)
                width = conf["units"]
            elif kind == Layer.DROPOUT:
                nxt_kind, nxt_conf = setups[pos + 1] if pos + 1 < len(setups) else (None, {})
                if (fused_classifier and not sparse_input and tag is None and nxt_kind == Layer.DENSE and nxt_conf["units"] <= 64
                        and pending_hops is None):
                    pending_dropout = conf["dropout_rate"]
                    layer = torch.nn.Identity()   # the following dense layer applies the dropout itself
                else:
                    layer = (L.SparseDropout(conf["dropout_rate"], at_eval=sparse_dropout_at_eval) if sparse_input
                             else torch.nn.Dropout(conf["dropout_rate"]))
            elif kind == Layer.SLICE:
                self.concat_inds.add(ind)
                layer = L.SliceLayer(**conf)
                src = tag_width[conf["loadTag"]] if conf["loadTag"] else width
                width = len(range(*conf["sliceObj"].indices(src)))
            elif kind == Layer.IDENTITY:
                layer = _SparseToDense()
                sparse_input = False
            elif kind == Layer.GCN:
                if sparse_input:
                    raise ValueError("a G layer needs dense inputs (put M/F or I before it)")
                self.graph_hops_inds.add(ind)
                layer = L.GCNLayer(**conf)
                sel = n_hops if conf["hops"] is None else len([h for h in range(n_hops) if h in conf["hops"]])
                pending_hops = sel
            elif kind == Layer.RELU:
                layer = torch.nn.Identity() if fuse_relu_into == ind - 1 else torch.nn.ReLU()
            elif kind == Layer.VECTORIZE:
                layer = torch.nn.Flatten(start_dim=1)
                if pending_hops is not None:
                    width *= pending_hops
                    pending_hops = None
            elif kind == Layer.CONCAT:
                self.concat_inds.add(ind)
                layer = L.ConcatLayer(tags=conf["tags"], addInputs=conf["addInputs"])
                width = (width if conf["addInputs"] else 0) + sum(tag_width[t] for t in conf["tags"] if t in tag_width)
            else:
                raise ValueError(f"Unsupported layer type {kind} specified in this model.")
            if conf.get("isEmbedding") and kind != Layer.DENSE:
                self.embedding_ind = ind
            self.layer_objs.append(layer)
            self.kinds.append(kind)
            if tag:
                self.tags[ind] = tag
                tag_width[tag] = width
        self.output_width = width
        self.fused = self._find_fusable_block(layer_setups, n_hops)
        # Propagation reuse (see layers.fused_propagation): an epoch is train_step then test_step, so the training forward
        # of epoch e+1 recomputes exactly what the evaluation of epoch e has just produced -- provided every layer in front of
        # the propagation behaves the same in both modes (no dropout there; H2GCN's default `M64-R-T1-G-V-...-D0.5-MO` has its
        # only dropout behind the concat).  The evaluation then fills a persistent buffer and the next training forward
        # adopts it: one propagation per epoch instead of two, same bits.
        self.reuse_propagation = (bool(reuse_propagation) and self.fused is not None
                                  and os.environ.get("H2GCN_PROPAGATION_REUSE", "1") != "0"
                                  and all(self._same_in_both_modes(m) for m in list(self.layer_objs)[: self.fused[0]]))
        self._weights_tag = 0        # bumped whenever the parameters change (note_update)
        self._prop_key = None        # what the persistent buffer currently holds
        self._prop_buf = None

    @staticmethod
    def _same_in_both_modes(layer) -> bool:
        if isinstance(layer, (torch.nn.Dropout, L.SparseDropout)):
            return False
        if isinstance(layer, L.DropoutDense):
            return layer.drop_prob == 0.0
        return True

    def note_update(self) -> None:
        """The parameters have changed (optimizer step, restored snapshot): what the propagation buffer holds is stale."""
        self._weights_tag += 1

    def load_state_dict(self, *args, **kwargs):
        self.note_update()
        return super().load_state_dict(*args, **kwargs)

    def _propagation_buffer(self, n: int, width: int, device) -> torch.Tensor:
        b = self._prop_buf
        if b is None or b.shape != (n, width) or b.device != device:
            with torch.inference_mode(False):   # an ordinary tensor even if the first evaluation runs under inference_mode
                self._prop_buf, self._prop_key = L.concat_buffer(n, width, device), None
        return self._prop_buf

    @staticmethod
    def _find_fusable_block(layer_setups, n_hops):
        """Locate ``<X>-T a_1 -G-V-T a_2 -G-V ... -C a_1 -C a_2 ...`` (the H2GCN-K propagation: K unfiltered G-V
        rounds whose inputs are tagged in order, immediately followed by the K matching concats).  Returns
        ``(first_G, end, K, tags)`` -- layers ``first_G .. end-1`` are then replaced by one fused launch sequence
        writing into the final concat buffer -- or None."""
        kinds = [k for k, _ in layer_setups]
        confs = [c for _, c in layer_setups]
        for start in range(1, len(kinds)):
            if kinds[start] != Layer.GCN or "tag" not in confs[start - 1]:
                continue
            K, i, tags = 0, start, [confs[start - 1]["tag"]]
            while i + 1 < len(kinds) and kinds[i] == Layer.GCN and confs[i].get("hops") is None and kinds[i + 1] == Layer.VECTORIZE:
                K += 1
                i += 2
                if "tag" in confs[i - 1]:
                    tags.append(confs[i - 1]["tag"])
                else:
                    break
            if K == 0 or len(tags) != K or len(set(tags)) != K:
                continue
            ok = all(i + j < len(kinds) and kinds[i + j] == Layer.CONCAT and confs[i + j]["tags"] == [tags[j]]
                     and confs[i + j].get("addInputs", True) and "tag" not in confs[i + j] for j in range(K))
            later_use = any(kinds[m] in (Layer.CONCAT, Layer.SLICE) for m in range(i + K, len(kinds)))
            if ok and not later_use and n_hops >= 1:
                return (start, i + K, K, tags)
        return None

    def forward(self, adj, inputs, adjhops, return_before: int = 0, execute_after: int = 0, tagged_out: dict = None,
                fuse: bool = True):
        n_layers = len(self.layer_objs)
        if return_before <= 0:
            return_before = n_layers + return_before
        if execute_after < 0:
            execute_after = n_layers + execute_after
        tagged = {}
        skip_until = 0
        # which feature operand (and which state of it) this pass started from
        feat_key = (id(inputs), getattr(inputs, "values_version", 0), getattr(inputs, "_version", 0), execute_after)
        for ind, layer in enumerate(self.layer_objs):
            if ind == return_before:
                return inputs
            if ind < execute_after:
                continue
            can_fuse = hasattr(adjhops, "fused_propagation") or (adjhops is not None and adjhops.n_rows == adjhops.n_cols)
            if fuse and can_fuse and self.fused is not None and ind == self.fused[0] and not (ind < return_before < self.fused[1]):
                # concat-free propagation: layers fused[0] .. fused[1]-1 in one go (row-sharded runs: the shard's
                # rows of the same buffer, one exchange per round)
                _, end, K, tags = self.fused
                sharded_hops = hasattr(adjhops, "fused_propagation")
                propagate = adjhops.fused_propagation if sharded_hops else functools.partial(
                    L.fused_propagation, adjhops, private_grad=self._buffer_grad_is_private(end))
                if self.reuse_propagation and inputs.is_cuda:
                    # (row-partitioned runs: every rank takes the same branch -- the decision depends only on the call
                    # sequence, which is the same on all ranks)
                    H_ = adjhops.n_hops
                    width = inputs.shape[1] * sum(H_ ** k for k in range(K + 1))
                    buf = self._propagation_buffer(inputs.shape[0], width, inputs.device)
                    plan_ = getattr(adjhops, "plan", adjhops)
                    # the parameters in front of the propagation enter through their autograd version counters (every in-place
                    # torch update bumps them; KerasAdam's raw-pointer kernel bumps them explicitly), so an ordinary torch
                    # training loop can never adopt a buffer computed from older weights; `note_update()` stays as the
                    # explicit form for anything that writes parameters behind torch's back
                    versions = tuple(p._version for p in self.parameters())
                    key = (self._weights_tag, versions, id(adjhops), getattr(plan_, "values_version", 0), feat_key, tuple(inputs.shape), K)
                    if not torch.is_grad_enabled():          # evaluation: fill the persistent buffer
                        self._prop_key = None
                        inputs = propagate(inputs, K, out=buf)
                        self._prop_key = key
                    elif self._prop_key == key:              # training forward right after it: adopt the buffer
                        inputs = propagate(inputs, K, out=buf, reuse=True)
                    else:                                    # anything else (first step, two training steps in a row, ...)
                        inputs = propagate(inputs, K)
                else:
                    inputs = propagate(inputs, K)
                skip_until = end
                w0 = tagged[tags[0]].shape[1]
                H = adjhops.n_hops
                pos = w0 * H ** K
                for k in range(1, K):  # expose r_1 .. r_{K-1} under their tags as views of the buffer
                    pos += w0 * H ** (k - 1)
                    tagged[tags[k]] = inputs[:, pos:pos + w0 * H ** k]
                continue
            if ind < skip_until:
                continue
            if ind in self.concat_inds:
                inputs = layer(inputs, **tagged)
            elif ind in self.graph_hops_inds:
                inputs = layer(adjhops, inputs)
            else:
                inputs = layer(inputs)
            if ind in self.tags:
                tagged[self.tags[ind]] = inputs
        if tagged_out is not None:
            tagged_out.update(tagged)
        return inputs

    def _buffer_grad_is_private(self, end: int) -> bool:
        """True when the concat buffer feeds exactly one layer whose backward ALLOCATES its input gradient (DropoutDense, Dense,
        Dropout) and nothing else can see it (no tag on the block's last layer): only then may the propagation's backward
        accumulate into that gradient tensor in place."""
        if (end - 1) in self.tags:
            return False

        def passes_through(layer) -> bool:
            # layers that hand their INPUT on as their output, with no autograd node of their own: the gradient that reaches the
            # buffer is then whatever the layer after them supplies.  nn.Identity (the placeholder a `D` leaves when the following
            # dense layer applies the dropout itself); nn.Dropout in eval mode or with rate 0 (F.dropout returns its input)
            if isinstance(layer, torch.nn.Identity):
                return True
            return isinstance(layer, torch.nn.Dropout) and (not layer.training or layer.p == 0)

        while end < len(self.layer_objs) and passes_through(self.layer_objs[end]) and end not in self.tags:
            end += 1
        if end >= len(self.layer_objs):
            return False           # the buffer is the model's output: its gradient belongs to the caller
        nxt = self.layer_objs[end]
        # an ACTIVE nn.Dropout multiplies by its mask in backward: a fresh tensor.  DropoutDense / Dense: a fresh GEMM output.
        return isinstance(nxt, (L.DropoutDense, Dense)) or (isinstance(nxt, torch.nn.Dropout) and nxt.training and nxt.p > 0)

    def restore_sparse_inputs(self) -> None:
        """Undo what ``SparseDropout`` did to the shared sparse feature operand (after the backward pass)."""
        for layer in self.layer_objs:
            if isinstance(layer, L.SparseDropout):
                layer.restore()

    def get_embeddings(self, adj, inputs, adjhops):
        if self.embedding_ind is None:
            raise ValueError("no layer is marked as the embedding (E)")
        return self(adj, inputs, adjhops, return_before=self.embedding_ind + 1)

    def regularization_loss(self) -> torch.Tensor:
        """keras ``regularizers.l2(w)`` on every dense kernel: ``w * sum(kernel ** 2)`` (``:239-240,247-248``)."""
        total = torch.zeros((), device=self.regularized[0].kernel.device) if self.regularized else torch.zeros(())
        for layer in self.regularized:
            total = total + self.l2 * (layer.kernel ** 2).sum()
        return total

    def regularization_value(self) -> torch.Tensor:
        """The same quantity WITHOUT a gradient, in one kernel launch when the kernels live on a GPU (``optim.l2_penalty``) -- for
        step closures that fold the penalty's gradient into the optimizer step (``KerasAdam.set_l2``) and only report its value."""
        ks = [layer.kernel for layer in self.regularized]
        if ks and self.l2 > 0 and all(k.is_cuda and k.dtype == torch.float32 and k.is_contiguous() for k in ks) and len(ks) <= 16:
            from ..optim import l2_penalty
            return l2_penalty([k.detach() for k in ks], [self.l2] * len(ks))
        return self.regularization_loss().detach()

    def data_loss(self, predictions, labels, mask) -> torch.Tensor:
        return masked_softmax_cross_entropy(predictions, labels, mask)

    def loss(self, predictions, labels, mask) -> torch.Tensor:
        return masked_softmax_cross_entropy(predictions, labels, mask) + self.regularization_loss()
