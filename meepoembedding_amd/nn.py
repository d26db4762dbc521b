"""The caller side of the boundary: the lookup table as a trainable torch layer.

forward  = find_or_insert (dynamic vocabulary: unseen ids get their hashed initial row)
backward = the table's own sparse optimizer (apply_adagrad / apply_adam) fed with the dense grad of the output — the
           update happens INSIDE backward, so the layer has no torch parameters and needs no torch optimizer.
Works with a LookupTable, a TieredLookupTable or a ShardedLookupTable (same method names).  Plumbing only: no kernel
lives here.  Reference anchor: /root/reference/README.md:2 ("Embedding designed for recommendation … systems").
"""
from __future__ import annotations

import torch


class _Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, keys: torch.Tensor, anchor: torch.Tensor, layer: "DynamicEmbedding"):
        flat = keys.reshape(-1)
        rows, _ = layer.table.find_or_insert(flat) if layer.training else layer.table.find(flat)
        ctx.layer, ctx.keys = layer, flat
        return rows.view(*keys.shape, layer.table.dim)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        layer = ctx.layer
        g = grad_out.reshape(-1, layer.table.dim).contiguous()
        layer.step += 1
        if layer.optimizer == "adagrad":
            layer.table.apply_adagrad(ctx.keys, g, lr=layer.lr, eps=layer.eps)
        else:
            layer.table.apply_adam(ctx.keys, g, lr=layer.lr, beta1=layer.betas[0], beta2=layer.betas[1], eps=layer.eps, step=layer.step)
        return None, None, None


class DynamicEmbedding(torch.nn.Module):
    """ids (any int64 tensor) -> fp32 [..., dim].  The table must have been created with the matching optimizer planes."""

    def __init__(self, table, optimizer: str = "adagrad", lr: float = 0.01, eps: float | None = None, betas=(0.9, 0.999)):
        super().__init__()
        if optimizer not in ("adagrad", "adam"):
            raise ValueError("optimizer must be 'adagrad' or 'adam'")
        self.table, self.optimizer, self.lr, self.betas = table, optimizer, lr, betas
        self.eps = eps if eps is not None else (1e-10 if optimizer == "adagrad" else 1e-8)
        self.step = 0
        # autograd only runs backward for functions with an input that requires grad
        self._anchor = torch.nn.Parameter(torch.zeros(()), requires_grad=True)

    def forward(self, keys: torch.Tensor) -> torch.Tensor:
        return _Lookup.apply(keys, self._anchor, self)
