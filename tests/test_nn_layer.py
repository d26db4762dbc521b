"""GPU: the table as a torch layer (forward = find_or_insert, backward = the table's sparse Adagrad) trains a small
model exactly like torch.nn.Embedding(sparse=True) + torch.optim.Adagrad on the CPU (third-party reference math)."""
import numpy as np
import pytest
import torch

import oracle
from meepoembedding_amd import _lib, synth
from meepoembedding_amd.nn import DynamicEmbedding, lookup, lookup_located


def test_custom_ops_on_cpu_adapter(built):
    """The torch.library registration (schema, fake kernel, autograd formula, tracing) with an oracle-backed table."""
    from _cpu_backend import CpuTable
    t = CpuTable(4096, 16, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
    layer = DynamicEmbedding(t, lr=0.05)
    k = torch.from_numpy(synth.keys_np(1, 0, 12)).view(3, 4)
    out = layer(k)
    assert out.shape == (3, 4, 16) and t.size() == 12           # training mode creates unseen ids
    w0, _ = t.find(k.view(-1))
    out.sum().backward()                                          # backward IS the optimizer step
    w1, _ = t.find(k.view(-1))
    # Adagrad with g = 1, acc0 = 0.1: w -= lr / sqrt(1.1)
    np.testing.assert_allclose((w0 - w1).numpy(), np.full((12, 16), 0.05 / np.sqrt(1.1), np.float32), rtol=1e-6)
    torch.library.opcheck(lookup, (k, layer._anchor, layer.table_id, False),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    from torch.fx.experimental.proxy_tensor import make_fx
    gm = make_fx(lambda kk, a: lookup(kk, a, layer.table_id, False) * 2, tracing_mode="fake")(k, layer._anchor)
    assert "torch.ops.meepo.lookup" in gm.code
    layer.eval()
    layer(torch.tensor([5, 6])); assert t.size() == 12          # eval mode does not create
    tid = layer.table_id
    del layer, gm
    import gc; gc.collect()
    with pytest.raises(RuntimeError):                              # the registry holds layers weakly
        lookup(k, torch.zeros(()), tid, False)


@pytest.mark.gpu
def test_dynamic_embedding_trains_like_torch_sparse_adagrad(dev):
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable
    torch.manual_seed(0)
    vocab, dim, steps, batch, bag = 500, 16, 6, 64, 5
    ids_all = torch.randint(0, vocab, (steps, batch, bag))
    ids_all[:, :, 0] = ids_all[:, :, 1]                      # duplicates inside every sample
    keys = torch.from_numpy(synth.keys_np(44, 0, vocab))     # id r <-> int64 key keys[r]
    w0 = torch.rand(vocab, dim) - 0.5
    head_w = torch.randn(dim, 1) * 0.1
    target = torch.randn(steps, batch, 1)

    # reference: torch on the CPU
    emb = torch.nn.Embedding(vocab, dim, sparse=True)
    with torch.no_grad():
        emb.weight.copy_(w0)
    opt = torch.optim.Adagrad(emb.parameters(), lr=0.05, eps=1e-10, initial_accumulator_value=0.1)
    for s in range(steps):
        opt.zero_grad()
        loss = ((emb(ids_all[s]).sum(1) @ head_w - target[s]) ** 2).mean()
        loss.backward()
        opt.step()

    # the table as a layer on the GPU
    table = LookupTable(4096, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=4096, initial_accumulator=0.1)
    table.insert(keys.to(dev), w0.to(dev))
    layer = DynamicEmbedding(table, optimizer="adagrad", lr=0.05, eps=1e-10).to(dev)
    hw = head_w.to(dev)
    for s in range(steps):
        k = keys[ids_all[s]].to(dev)
        loss = ((layer(k).sum(1) @ hw - target[s].to(dev)) ** 2).mean()
        loss.backward()
    got, found = table.find(keys.to(dev))
    assert bool(found.all())
    np.testing.assert_allclose(got.cpu().numpy(), emb.weight.detach().numpy(), rtol=2e-5, atol=1e-6)
    # unseen ids in training mode are created with the initial row; in eval mode they are not
    new = torch.tensor([[123456789, 987654321]], device=dev)
    layer.eval(); layer(new); assert table.size() == vocab
    layer.train(); layer(new); assert table.size() == vocab + 2
    # the layer over ONE HBM table goes through meepo::lookup_located (the backward applies on the forward's slot handles)
    k = keys[:12].view(3, 4).to(dev)
    torch.library.opcheck(lookup_located, (k, layer._anchor, layer.table_id, False),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    rows, slots = lookup_located(k, layer._anchor, layer.table_id, True)
    assert rows.shape == (3, 4, dim) and slots.shape == (12,) and bool((slots >= 0).all())
    assert torch.equal(slots & _lib.HANDLE_SLOT_MASK, table.locate(k.view(-1))[0])   # handle = slot + layout epoch (bits 40..61)


@pytest.mark.gpu
def test_collection_layer_trains_like_per_table_layers(dev):
    """DynamicEmbeddingCollection (one grouped lookup + one grouped Adagrad step) == one DynamicEmbedding per table."""
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable, TableGroup
    from meepoembedding_amd.nn import DynamicEmbeddingCollection
    torch.manual_seed(1)
    dim, n_tables, steps = 16, 3, 4
    mk = lambda j: LookupTable(4096, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=4096, initial_accumulator=0.1,
                               initializer=1, init_scale=0.1, init_seed=j)
    a, b = [mk(j) for j in range(n_tables)], [mk(j) for j in range(n_tables)]
    coll = DynamicEmbeddingCollection(TableGroup(a, max_apply_batch=4096), lr=0.05).to(dev)
    solo = [DynamicEmbedding(t, lr=0.05).to(dev) for t in b]
    head = torch.randn(dim, 1, device=dev) * 0.1
    for s in range(steps):
        lens = [int(x) for x in torch.randint(1, 200, (n_tables,))]
        segs = [torch.from_numpy(synth.keys_np(90 + j, 0, 150))[torch.randint(0, 150, (m,))].to(dev) for j, m in enumerate(lens)]
        keys = torch.cat(segs)
        offs = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int64, device=dev)
        target = torch.randn(keys.numel(), 1, device=dev)
        ((coll(keys, offs) @ head - target) ** 2).mean().backward()
        rows = torch.cat([layer(k) for layer, k in zip(solo, segs)])
        ((rows @ head - target) ** 2).mean().backward()
    for x, y in zip(a, b):
        ex, ey = x.export(with_state=True), y.export(with_state=True)
        ix, iy = torch.argsort(ex[0]), torch.argsort(ey[0])
        assert torch.equal(ex[0][ix], ey[0][iy]) and x.size() > 0
        torch.testing.assert_close(ex[1][ix], ey[1][iy], rtol=1e-6, atol=1e-9)
        torch.testing.assert_close(ex[2][ix], ey[2][iy], rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sum", "mean"])
def test_embedding_bag_layer_trains_like_torch_embedding_bag(dev, mode):
    """DynamicEmbeddingBag (fused pooled lookup + indexed sparse Adagrad) == torch.nn.EmbeddingBag(sparse=True) + Adagrad."""
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable
    from meepoembedding_amd.nn import DynamicEmbeddingBag
    torch.manual_seed(2)
    vocab, dim, steps, n_bags = 300, 16, 5, 40
    keys = torch.from_numpy(synth.keys_np(45, 0, vocab))
    w0 = torch.rand(vocab, dim) - 0.5
    head = torch.randn(dim, 1) * 0.1
    ref = torch.nn.EmbeddingBag(vocab, dim, mode=mode, sparse=True)
    with torch.no_grad():
        ref.weight.copy_(w0)
    opt = torch.optim.Adagrad(ref.parameters(), lr=0.05, eps=1e-10, initial_accumulator_value=0.1)
    table = LookupTable(2048, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=4096, initial_accumulator=0.1)
    table.insert(keys.to(dev), w0.to(dev))
    layer = DynamicEmbeddingBag(table, mode=mode, lr=0.05, eps=1e-10).to(dev)
    for s in range(steps):
        lens = torch.randint(1, 9, (n_bags,))          # torch's EmbeddingBag mean of an empty bag differs in backward: keep bags non-empty
        ids = torch.randint(0, vocab, (int(lens.sum()),))
        off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)])
        target = torch.randn(n_bags, 1)
        opt.zero_grad()
        ((ref(ids, off[:-1]) @ head - target) ** 2).mean().backward()
        opt.step()
        ((layer(keys[ids].to(dev), off.to(dev)) @ head.to(dev) - target.to(dev)) ** 2).mean().backward()
    got, found = table.find(keys.to(dev))
    assert bool(found.all())
    np.testing.assert_allclose(got.cpu().numpy(), ref.weight.detach().numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_embedding_bag_collection_layer(dev):
    """DynamicEmbeddingBag over a TableGroup (one pooled launch, one grouped optimizer step) == one bag layer per table."""
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable, TableGroup
    from meepoembedding_amd.nn import DynamicEmbeddingBag
    torch.manual_seed(3)
    dim, n_tables, bpt, steps = 16, 3, 20, 3
    mk = lambda: LookupTable(2048, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=4096, initial_accumulator=0.1)
    a, b = [mk() for _ in range(n_tables)], [mk() for _ in range(n_tables)]
    univ = [torch.from_numpy(synth.keys_np(120 + j, 0, 200)).to(dev) for j in range(n_tables)]
    for j in range(n_tables):
        w = torch.rand(200, dim, device=dev) - 0.5
        a[j].insert(univ[j], w); b[j].insert(univ[j], w)
    coll = DynamicEmbeddingBag(TableGroup(a, max_apply_batch=4096), mode="mean", lr=0.05).to(dev)
    solo = [DynamicEmbeddingBag(t, mode="mean", lr=0.05).to(dev) for t in b]
    head = torch.randn(dim, 1, device=dev) * 0.1
    for s in range(steps):
        lens = torch.randint(1, 7, (n_tables * bpt,))
        off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)]).to(dev)
        segs = [univ[j][torch.randint(0, 200, (int(lens[j * bpt:(j + 1) * bpt].sum()),)).to(dev)] for j in range(n_tables)]
        keys = torch.cat(segs)
        target = torch.randn(n_tables * bpt, 1, device=dev)
        ((coll(keys, off) @ head - target) ** 2).mean().backward()
        pooled = torch.cat([solo[j](segs[j], off[j * bpt:(j + 1) * bpt + 1] - off[j * bpt]) for j in range(n_tables)])
        ((pooled @ head - target) ** 2).mean().backward()
    for x, y in zip(a, b):
        ex, ey = x.export(with_state=True), y.export(with_state=True)
        ix, iy = torch.argsort(ex[0]), torch.argsort(ey[0])
        torch.testing.assert_close(ex[1][ix], ey[1][iy], rtol=1e-6, atol=1e-9)
        torch.testing.assert_close(ex[2][ix], ey[2][iy], rtol=1e-6, atol=1e-9)
    # create_missing: unseen ids enter their member table in a training forward, not in eval
    grow = DynamicEmbeddingBag(TableGroup(a, max_apply_batch=4096), mode="sum", lr=0.05, create_missing=True).to(dev)
    new_ids = torch.arange(10**9, 10**9 + n_tables * 4, device=dev)
    new_off = torch.arange(0, n_tables * 4 + 1, 2, dtype=torch.int64, device=dev)       # 2 bags of 2 ids per table
    before = [t.size() for t in a]
    grow.eval(); grow(new_ids, new_off); assert [t.size() for t in a] == before
    grow.train(); grow(new_ids, new_off); assert [t.size() for t in a] == [s + 4 for s in before]


@pytest.mark.gpu
def test_training_forward_carries_the_backward_partition(dev):
    """DynamicEmbedding in training mode: the lookup's launch also partitions the batch for the backward's apply
    (mee_find_or_insert_located_prepare).  Same table as with the knob off — growing vocabulary, duplicates, absent keys —; a forward
    that no backward follows, two layers over one table and a caller who changes the table between forward and backward stay correct."""
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable, MeepoError
    dim, steps, batch = 32, 5, 3000
    rng = np.random.default_rng(4)
    pool = synth.keys_np(71, 0, 5000)
    tabs = [LookupTable(16384, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=4096, initial_accumulator=0.1, initializer=1, init_scale=0.05, init_seed=9)
            for _ in range(2)]
    layers = [DynamicEmbedding(t, optimizer="adagrad", lr=0.05).to(dev) for t in tabs]
    layers[1].fuse_backward_partition = False
    head = torch.randn(dim, 1, device=dev) * 0.1
    for s in range(steps):
        idx = np.minimum(rng.zipf(1.3, size=batch) - 1, pool.size - 1) if s % 2 else rng.integers(0, 1000 * (s + 1), batch)   # new ids every step
        k = torch.from_numpy(pool[idx]).to(dev).view(batch // 4, 4)
        for layer in layers:
            ((layer(k).sum(1) @ head) ** 2).mean().backward()
        assert getattr(tabs[0], "_nn_prepared", None) is None   # the backward consumed the forward's partition
    ea, eb = tabs[0].export(with_state=True), tabs[1].export(with_state=True)
    ia, ib = torch.argsort(ea[0]), torch.argsort(eb[0])
    assert torch.equal(ea[0][ia], eb[0][ib]) and tabs[0].status() == tabs[1].status() == 0
    for xa, xb in zip(ea[1:], eb[1:]):
        if xa is not None:
            torch.testing.assert_close(xa[ia], xb[ib], rtol=1e-6, atol=1e-9)
    # a forward without a backward leaves a partition behind: a mutator that comes before any backward drops it (the library's own doing: an
    # eviction hook or a growth step between forward and backward must not fail), and so does the next forward
    t, layer = tabs[0], layers[0]
    k = torch.from_numpy(pool[:64]).to(dev)
    out = layer(k)
    assert t._nn_prepared is not None
    rows_now, _ = t.find(k)
    t.insert(k, rows_now)   # accepted: rewrites the rows with what they hold
    out_b = layer(k)
    (out_b.sum()).backward()   # the backward of a forward whose partition a mutator dropped: the apply partitions its batch again
    assert t._nn_prepared is None and t.status() == 0
    out = layer(k)             # again a forward without a backward
    out2 = layer(k)   # drops the stale partition, makes its own
    (out2.sum()).backward()
    assert t._nn_prepared is None
    t.insert(k, torch.zeros(64, dim, device=dev))   # accepted again
    # two layers over ONE table: the second forward drops the first one's partition; both backwards still apply (each on its own keys)
    other = DynamicEmbedding(t, optimizer="adagrad", lr=0.05).to(dev)
    before, _ = t.find(k)
    la, lb = layer(k[:32]), other(k[32:])
    (la.sum() + lb.sum()).backward()
    after, _ = t.find(k)
    assert bool(((after - before).abs().sum(1) > 0).all()) and t.status() == 0
