"""The caller side of the boundary: the lookup table as a trainable torch layer, registered as
`torch.library` custom ops on PyTorch-ROCm.

    meepo::lookup(keys, anchor, table_id, insert_missing) -> rows     forward  = find_or_insert (training: unseen ids get
                                                                                 their hashed initial row) or find (eval)
    meepo::apply_grad(keys, grad_rows, table_id)                      backward = the table's own sparse optimizer
                                                                                 (apply_adagrad / apply_adam), fed with
                                                                                 the dense grad of the output
    meepo::lookup_located / meepo::apply_grad_located                 the same pair over ONE HBM table: the forward
                                                                                 (find[_or_insert]_located) hands every key's slot
                                                                                 to the backward, whose apply then does not probe

The update happens INSIDE backward, so the layer has no torch parameters and needs no torch optimizer.  Both ops have
fake (meta) kernels, so a model containing the layer traces and exports; they are opaque to the tracer (the table is
state outside the graph, addressed by `table_id`).  Works with a LookupTable, a TieredLookupTable, a ShardedLookupTable
or a PeerShardedTable (same method names).  Plumbing only: no kernel lives here.
Reference anchor: /root/reference/README.md:2 ("Embedding designed for recommendation … systems").
"""
from __future__ import annotations

import itertools
import weakref

import torch

_LAYERS: "weakref.WeakValueDictionary[int, DynamicEmbedding]" = weakref.WeakValueDictionary()
_IDS = itertools.count(1)


def _layer(table_id: int) -> "DynamicEmbedding":
    try:
        return _LAYERS[table_id]
    except KeyError:
        raise RuntimeError(f"meepo: no live DynamicEmbedding with table_id {table_id}") from None


@torch.library.custom_op("meepo::lookup", mutates_args=())
def lookup(keys: torch.Tensor, anchor: torch.Tensor, table_id: int, insert_missing: bool) -> torch.Tensor:
    layer = _layer(table_id)
    flat = keys.reshape(-1)
    rows, _ = layer.table.find_or_insert(flat) if insert_missing else layer.table.find(flat)
    return rows.reshape(*keys.shape, layer.table.dim).clone()   # never alias the table's / exchange's buffers


@lookup.register_fake
def _(keys, anchor, table_id, insert_missing):
    return keys.new_empty((*keys.shape, _layer(table_id).table.dim), dtype=torch.float32)


@torch.library.custom_op("meepo::apply_grad", mutates_args=())
def apply_grad(keys: torch.Tensor, grad_rows: torch.Tensor, table_id: int) -> None:
    layer = _layer(table_id)
    g = grad_rows.reshape(-1, layer.table.dim).contiguous()
    layer.step += 1
    if layer.optimizer == "adagrad":
        layer.table.apply_adagrad(keys.reshape(-1), g, lr=layer.lr, eps=layer.eps)
    else:
        layer.table.apply_adam(keys.reshape(-1), g, lr=layer.lr, beta1=layer.betas[0], beta2=layer.betas[1], eps=layer.eps, step=layer.step)


@apply_grad.register_fake
def _(keys, grad_rows, table_id):
    return None


def _setup(ctx, inputs, output):
    keys, _, table_id, _ = inputs
    ctx.save_for_backward(keys)
    ctx.table_id = table_id


def _backward(ctx, grad_out):
    (keys,) = ctx.saved_tensors
    apply_grad(keys, grad_out.contiguous(), ctx.table_id)
    return None, None, None, None


lookup.register_autograd(_backward, setup_context=_setup)


# ---- the same pair for a plain HBM table: the forward hands its slot handles to the backward, whose apply then does not probe ------
@torch.library.custom_op("meepo::lookup_located", mutates_args=())
def lookup_located(keys: torch.Tensor, anchor: torch.Tensor, table_id: int, insert_missing: bool, prepare: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (rows [..., dim], slot handle of every key (-1 = absent) for the backward of this step).
    prepare: a backward for exactly these keys follows — the lookup's launch also partitions the batch for its apply (mee_find_located_prepare /
    mee_find_or_insert_located_prepare).  A partition that no backward consumed (a forward without backward, a second layer over the same
    table) is dropped here before the next one is made."""
    layer = _layer(table_id)
    table = layer.table
    flat = keys.reshape(-1)
    if getattr(table, "_nn_prepared", None) is not None:   # an earlier forward's partition that no backward used
        table.apply_discard()
        table._nn_prepared = None
    prepare = prepare and flat.is_contiguous() and flat.numel() > 0 and flat.numel() <= table.max_batch
    rows, _, slots = (table.find_or_insert_located(flat, prepare_apply=prepare) if insert_missing else table.find_located(flat, prepare_apply=prepare))
    if prepare:
        table._nn_prepared = (flat.data_ptr(), flat.numel())
    return rows.reshape(*keys.shape, table.dim), slots   # fresh tensors: nothing aliases the table


@lookup_located.register_fake
def _(keys, anchor, table_id, insert_missing, prepare=False):
    return keys.new_empty((*keys.shape, _layer(table_id).table.dim), dtype=torch.float32), keys.new_empty(keys.numel())


@torch.library.custom_op("meepo::apply_grad_located", mutates_args=())
def apply_grad_located(keys: torch.Tensor, grad_rows: torch.Tensor, located: torch.Tensor, table_id: int) -> None:
    layer = _layer(table_id)
    g = grad_rows.reshape(-1, layer.table.dim).contiguous()
    slots = located if located.numel() == keys.numel() else None
    layer.step += 1
    prepared = getattr(layer.table, "_nn_prepared", None)
    if prepared is not None:   # the forward's partition is for exactly (this tensor, this length), or it is dropped
        flat = keys.reshape(-1)
        if prepared != (flat.data_ptr(), flat.numel()):
            layer.table.apply_discard()
        layer.table._nn_prepared = None
    if layer.optimizer == "adagrad":
        layer.table.apply_adagrad(keys.reshape(-1), g, lr=layer.lr, eps=layer.eps, slots=slots)
    else:
        layer.table.apply_adam(keys.reshape(-1), g, lr=layer.lr, beta1=layer.betas[0], beta2=layer.betas[1], eps=layer.eps, step=layer.step,
                               slots=slots)


@apply_grad_located.register_fake
def _(keys, grad_rows, located, table_id):
    return None


def _setup_located(ctx, inputs, output):
    keys, _, table_id = inputs[:3]
    ctx.save_for_backward(keys, output[1])
    ctx.table_id = table_id
    ctx.layout_epoch = getattr(_layer(table_id).table, "layout_epoch", None)   # handles are slot numbers: stale once rows move
    ctx.mark_non_differentiable(output[1])


def _backward_located(ctx, grad_out, _grad_located):
    keys, located = ctx.saved_tensors
    if getattr(_layer(ctx.table_id).table, "layout_epoch", None) != ctx.layout_epoch:
        located = located.new_empty(0)   # the table changed between forward and backward: the apply probes for itself
    apply_grad_located(keys, grad_out.contiguous(), located, ctx.table_id)
    return None, None, None, None, None


lookup_located.register_autograd(_backward_located, setup_context=_setup_located)


# ---- pooled: sum / mean per bag fused into the lookup, the bag's grad row indexed in the update ---------------------------
@torch.library.custom_op("meepo::lookup_pooled", mutates_args=())
def lookup_pooled(keys: torch.Tensor, bag_offsets: torch.Tensor, anchor: torch.Tensor, table_id: int, mean: bool) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (pooled rows [n_bags, dim], located rows [n] — per-position handles the backward of this step reuses; empty for a
    single table, whose apply probes for itself)"""
    layer = _layer(table_id)
    if layer.create_missing and layer.training and keys.numel():
        # dynamic vocabulary: unseen ids enter their table (hashed initial row + optimizer state) before the pooled lookup
        if hasattr(layer.table, "apply_pooled"):
            bpt = (bag_offsets.numel() - 1) // len(layer.table.tables)
            layer.table.find_or_insert(keys, bag_offsets[::bpt].contiguous())
        else:
            layer.table.find_or_insert(keys)
    if hasattr(layer.table, "apply_pooled"):   # a TableGroup
        located = torch.empty(keys.numel(), dtype=torch.int64, device=keys.device)
        out, _ = layer.table.find_pooled(keys, bag_offsets, "mean" if mean else "sum", located=located)
        return out, located
    out, _ = layer.table.find_pooled(keys, bag_offsets, "mean" if mean else "sum")
    return out, keys.new_empty(0)


@lookup_pooled.register_fake
def _(keys, bag_offsets, anchor, table_id, mean):
    layer = _layer(table_id)
    return (keys.new_empty((bag_offsets.numel() - 1, layer.table.dim), dtype=torch.float32),
            keys.new_empty(keys.numel() if hasattr(layer.table, "apply_pooled") else 0))


@torch.library.custom_op("meepo::apply_grad_pooled", mutates_args=())
def apply_grad_pooled(keys: torch.Tensor, bag_offsets: torch.Tensor, grad_bags: torch.Tensor, located: torch.Tensor, table_id: int, mean: bool) -> None:
    layer = _layer(table_id)
    lens = bag_offsets[1:] - bag_offsets[:-1]
    bag_of = torch.repeat_interleave(torch.arange(lens.numel(), device=keys.device), lens, output_size=keys.numel())
    g = grad_bags.contiguous()
    if mean:   # d mean / d row = 1 / length for every member of the bag
        g = g / lens.clamp(min=1).to(torch.float32)[:, None]
    layer.step += 1
    if hasattr(layer.table, "apply_pooled"):   # a TableGroup: the whole collection in one step, on the rows the forward located
        layer.table.apply_pooled(keys, bag_offsets, g, bag_of, layer.optimizer, lr=layer.lr, eps=layer.eps, beta1=layer.betas[0],
                                 beta2=layer.betas[1], step=layer.step, located=located if located.numel() == keys.numel() else None)
    elif layer.optimizer == "adagrad":
        layer.table.apply_adagrad(keys, g, lr=layer.lr, eps=layer.eps, grad_index=bag_of)
    else:
        layer.table.apply_adam(keys, g, lr=layer.lr, beta1=layer.betas[0], beta2=layer.betas[1], eps=layer.eps, step=layer.step,
                               grad_index=bag_of)


@apply_grad_pooled.register_fake
def _(keys, bag_offsets, grad_bags, located, table_id, mean):
    return None


def _setup_pooled(ctx, inputs, output):
    keys, bag_offsets, _, table_id, mean = inputs
    ctx.save_for_backward(keys, bag_offsets, output[1])
    ctx.table_id, ctx.mean = table_id, mean
    # the located handles are slot numbers: valid only while no member table removes / clears / rehashes rows
    ctx.layout_epoch = getattr(_layer(table_id).table, "layout_epoch", None)
    ctx.mark_non_differentiable(output[1])


def _backward_pooled(ctx, grad_out, _grad_located):
    keys, bag_offsets, located = ctx.saved_tensors
    if getattr(_layer(ctx.table_id).table, "layout_epoch", None) != ctx.layout_epoch:
        located = located.new_empty(0)   # a table changed between forward and backward: the apply probes for itself
    apply_grad_pooled(keys, bag_offsets, grad_out.contiguous(), located, ctx.table_id, ctx.mean)
    return None, None, None, None, None


lookup_pooled.register_autograd(_backward_pooled, setup_context=_setup_pooled)


# ---- the same pair of ops over a TableGroup: the whole embedding collection of a model in one lookup / one update -------
@torch.library.custom_op("meepo::lookup_jagged", mutates_args=())
def lookup_jagged(keys: torch.Tensor, offsets: torch.Tensor, anchor: torch.Tensor, table_id: int, insert_missing: bool) -> torch.Tensor:
    layer = _layer(table_id)
    rows, _ = layer.group.find_or_insert(keys, offsets) if insert_missing else layer.group.find(keys, offsets)
    return rows


@lookup_jagged.register_fake
def _(keys, offsets, anchor, table_id, insert_missing):
    return keys.new_empty((keys.numel(), _layer(table_id).group.dim), dtype=torch.float32)


@torch.library.custom_op("meepo::apply_grad_jagged", mutates_args=())
def apply_grad_jagged(keys: torch.Tensor, offsets: torch.Tensor, grad_rows: torch.Tensor, table_id: int) -> None:
    layer = _layer(table_id)
    layer.step += 1
    if layer.optimizer == "adagrad":
        layer.group.apply_adagrad(keys, offsets, grad_rows.contiguous(), lr=layer.lr, eps=layer.eps)
    else:
        layer.group.apply_adam(keys, offsets, grad_rows.contiguous(), lr=layer.lr, beta1=layer.betas[0], beta2=layer.betas[1],
                               eps=layer.eps, step=layer.step)


@apply_grad_jagged.register_fake
def _(keys, offsets, grad_rows, table_id):
    return None


def _setup_jagged(ctx, inputs, output):
    keys, offsets, _, table_id, _ = inputs
    ctx.save_for_backward(keys, offsets)
    ctx.table_id = table_id


def _backward_jagged(ctx, grad_out):
    keys, offsets = ctx.saved_tensors
    apply_grad_jagged(keys, offsets, grad_out.contiguous(), ctx.table_id)
    return None, None, None, None, None


lookup_jagged.register_autograd(_backward_jagged, setup_context=_setup_jagged)


class _SparseOptimizerSettings:
    def _init_settings(self, optimizer, lr, eps, betas):
        if optimizer not in ("adagrad", "adam"):
            raise ValueError("optimizer must be 'adagrad' or 'adam'")
        self.optimizer, self.lr, self.betas = optimizer, lr, betas
        self.eps = eps if eps is not None else (1e-10 if optimizer == "adagrad" else 1e-8)
        self.step = 0
        self.table_id = next(_IDS)
        _LAYERS[self.table_id] = self
        # autograd only runs backward for ops with an input that requires grad
        self._anchor = torch.nn.Parameter(torch.zeros(()), requires_grad=True)


class DynamicEmbeddingCollection(torch.nn.Module, _SparseOptimizerSettings):
    """All embedding tables of a model behind one module: keys = the tables' id batches concatenated, offsets = the
    n_tables + 1 segment bounds (int64, on the device) -> fp32 [len(keys), dim].  Forward is ONE grouped find_or_insert
    (find in eval mode), backward ONE grouped optimizer step, whatever the number of tables (TableGroup in table.py)."""

    def __init__(self, group, optimizer: str = "adagrad", lr: float = 0.01, eps: float | None = None, betas=(0.9, 0.999)):
        super().__init__()
        self.group = group
        self._init_settings(optimizer, lr, eps, betas)

    def forward(self, keys: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
        return lookup_jagged(keys, offsets, self._anchor, self.table_id, self.training)


class DynamicEmbeddingBag(torch.nn.Module, _SparseOptimizerSettings):
    """torch.nn.EmbeddingBag over a lookup table — or over a TableGroup, which makes it an embedding-bag COLLECTION: bag b
    then belongs to member b // bags_per_table and the whole model's sparse forward is one launch, its backward seven.
    (keys [n], bag_offsets [n_bags + 1], both on the device) -> [n_bags, dim] sums or means; backward runs the table's sparse optimizer with the bag's grad row for every member (no [n, dim]
    tensor exists in either direction).  create_missing=True (training mode only): unseen ids are inserted with their
    hashed initial row first (one more pass over the ids); otherwise absent ids read the default row and are not trained."""

    def __init__(self, table, mode: str = "sum", optimizer: str = "adagrad", lr: float = 0.01, eps: float | None = None, betas=(0.9, 0.999),
                 create_missing: bool = False):
        super().__init__()
        if mode not in ("sum", "mean"):
            raise ValueError("mode must be 'sum' or 'mean'")
        self.table, self.mode, self.create_missing = table, mode, create_missing
        self._init_settings(optimizer, lr, eps, betas)

    def forward(self, keys: torch.Tensor, bag_offsets: torch.Tensor) -> torch.Tensor:
        return lookup_pooled(keys, bag_offsets, self._anchor, self.table_id, self.mode == "mean")[0]


class DynamicEmbedding(torch.nn.Module):
    """ids (any int64 tensor) -> fp32 [..., dim].  The table must have been created with the matching optimizer planes."""

    def __init__(self, table, optimizer: str = "adagrad", lr: float = 0.01, eps: float | None = None, betas=(0.9, 0.999)):
        super().__init__()
        if optimizer not in ("adagrad", "adam"):
            raise ValueError("optimizer must be 'adagrad' or 'adam'")
        self.table, self.optimizer, self.lr, self.betas = table, optimizer, lr, betas
        self.eps = eps if eps is not None else (1e-10 if optimizer == "adagrad" else 1e-8)
        self.step = 0
        self.fuse_backward_partition = True   # see forward()
        self.table_id = next(_IDS)
        _LAYERS[self.table_id] = self
        # autograd only runs backward for ops with an input that requires grad
        self._anchor = torch.nn.Parameter(torch.zeros(()), requires_grad=True)

    def forward(self, keys: torch.Tensor) -> torch.Tensor:
        if hasattr(self.table, "find_or_insert_located"):   # one HBM table: the backward updates the rows at the slots this lookup found
            # training: the lookup's launch also partitions the batch for the backward's apply (nothing else may change the table in between:
            # the C-ABI refuses mutators while that partition is pending — table.apply_discard() drops it)
            return lookup_located(keys, self._anchor, self.table_id, self.training, self.training and torch.is_grad_enabled() and self.fuse_backward_partition)[0]
        return lookup(keys, self._anchor, self.table_id, self.training)   # sharded / tiered tables: their apply routes by key
