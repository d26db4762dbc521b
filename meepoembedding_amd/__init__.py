"""meepoembedding_amd — MI355X (gfx950) GPU backend for a dynamic lookup-table embedding.

The package is a thin host layer over libmeepo_hip.so (C-ABI: include/meepo_embedding.h; hand-written HIP
kernels in csrc/).  Importing the package does not touch the GPU; creating a table does, and fails loudly when
the HIP extension or a gfx950 device is missing — there is no CPU fallback.
"""
from ._lib import (EMPTY_KEY, INIT_CONSTANT, INIT_UNIFORM, OPT_ADAGRAD, OPT_ADAM, OPT_NONE, RECLAIMED_KEY,
                   STATUS_RESERVED_KEY, STATUS_TABLE_FULL, MeepoError)
from .table import LookupTable, Router, TableGroup, hash_batch

__all__ = ["LookupTable", "Router", "TableGroup", "hash_batch", "MeepoError", "OPT_NONE", "OPT_ADAGRAD", "OPT_ADAM",
           "INIT_CONSTANT", "INIT_UNIFORM", "STATUS_TABLE_FULL", "STATUS_RESERVED_KEY", "EMPTY_KEY", "RECLAIMED_KEY"]
