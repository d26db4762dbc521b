"""Child-process self-test of the peer-mapped transport on the real topology (run by bench.py before it trusts that path).

Each bench rank starts one of these; the children form their OWN gloo group (MASTER_PORT is the parent's + 17) on the same
GPUs, build a tiny row-sharded table, and run a sharded find and an Adagrad push through PeerShardedFind (HIP IPC mapping,
kernel stores into peer memory over xGMI).  Exit 0 = every row came back right.  A GPU fault or a hang in here ends only this
child — the parent then falls back to the RCCL all-to-all path instead of dying inside its timed run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable, Router, hash_batch, synth
    from meepoembedding_amd.p2p import PeerShardedFind
    dim, n_keys, batch = 64, 200_000, 8192
    cap_slots = int(batch / world * 1.25) + 4096
    table = LookupTable(int(n_keys / world / 0.5), dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=max(world * cap_slots, 1 << 16))
    for s in range(0, n_keys, 1 << 16):
        k = synth.keys_t(1, s, min(1 << 16, n_keys - s), dev)
        mine = k[hash_batch(k, 1, world)[2] == rank]
        table.insert(mine, synth.rows_t(mine, dim, 2))
    dist.barrier()
    peer = PeerShardedFind(table, Router(world, batch, device=dev), max_batch=batch, payload=True)
    g = torch.Generator().manual_seed(100 + rank)
    for it in range(3):
        q = synth.keys_t(1, 0, n_keys, dev)[torch.randint(0, n_keys, (batch,), generator=g).to(dev)]
        rows, found = peer.find(q)
        if not (bool(found.all()) and torch.equal(rows, synth.rows_t(q, dim, 2))):
            print(f"p2p selftest rank {rank}: wrong rows", file=sys.stderr)
            return 3
    # a gradient push: every rank sends grads for its own query keys; owners apply; then rows must have changed everywhere
    uq = torch.unique(q)
    peer.apply_adagrad(uq, torch.ones(uq.numel(), dim, device=dev), lr=0.5)
    rows2, _ = peer.find(uq)
    moved = (rows2 - synth.rows_t(uq, dim, 2)).abs().max().item()
    peer.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0 if moved > 0.1 else 4


if __name__ == "__main__":
    sys.exit(main())
