"""Child-process self-test of the RCCL exchange behind the C-ABI (mee_sharded_*) on the real topology, run by bench.py before
it trusts that path: one child per bench rank, own gloo group (MASTER_PORT is the parent's + 23) for the ncclUniqueId hand-over,
own RCCL communicator on the same GPUs.  Exact and padded segment layouts: lookups must return the key-derived rows, a gradient
push must move them.  Exit 0 = pass.  A fault or a hang in here ends only this child; the parent then keeps the torch.distributed
all-to-all path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable, hash_batch, synth
    from meepoembedding_amd.sharded import RcclShardedTable
    dim, n_keys, batch = 64, 200_000, 8192
    for slack in (0.0, 1.5):
        cap = int(batch / world * 1.5) + 2048
        table = LookupTable(int(n_keys / world / 0.5), dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=max(world * cap, world * batch, 1 << 16))
        for s in range(0, n_keys, 1 << 16):
            k = synth.keys_t(1, s, min(1 << 16, n_keys - s), dev)
            mine = k[hash_batch(k, 1, world)[2] == rank]
            table.insert(mine, synth.rows_t(mine, dim, 2))
        dist.barrier()
        sh = RcclShardedTable(table, batch, pad_slack=slack)
        g = torch.Generator().manual_seed(100 + rank)
        for it in range(3):
            q = synth.keys_t(1, 0, n_keys, dev)[torch.randint(0, n_keys, (batch,), generator=g).to(dev)]
            rows, found = sh.find(q)
            if not (bool(found.all()) and torch.equal(rows, synth.rows_t(q, dim, 2))):
                print(f"rccl selftest rank {rank} (slack {slack}): wrong rows", file=sys.stderr)
                return 3
        uq = torch.unique(q)
        sh.apply_adagrad(uq, torch.ones(uq.numel(), dim, device=dev), lr=0.5)
        rows2, _ = sh.find(uq)
        moved = (rows2 - synth.rows_t(uq, dim, 2)).abs().max().item()
        if sh.size() != n_keys or sh.status() != 0 or not moved > 0.1:
            print(f"rccl selftest rank {rank} (slack {slack}): size/status/update check failed", file=sys.stderr)
            return 4
        sh.close()
        table.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
