"""N > 1 path on CPU: two gloo ranks exercise the gradient exchange and the global-ray-count loss
normalisation of tinynerf_amd.run.Trainer (the HIP kernels themselves need a GPU; the exchange logic
does not).  Rendezvous on 127.0.0.1."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Stub:
    """what Trainer.all_reduce_grads / _planes_ready / global_mse read"""
    def __init__(self, renderer, world):
        from tinynerf_amd.run import Trainer
        self.renderer, self.world, self.device = renderer, world, torch.device("cpu")
        self._early = {}
        # K-Planes row restriction of the exchange (Trainer._refresh_reduce_rows): plane 0 = `renderer.plane`, only rows 16..99
        # can carry a gradient this step, rows 10..79 could in the step before: the union travels
        self._plane_of = {id(renderer.plane): 0}
        self._reduce_rows, self._reduce_rows_prev = [(16, 100)], [(10, 80)]
        self._reduce_view = lambda g, plane: Trainer._reduce_view(self, g, plane)
        self._flat, self._flat_used, self._flat_ids = None, 0, set()       # filled by the worker: small gradients as views of one bucket


def _make_model():
    torch.manual_seed(0)
    m = torch.nn.Module()
    m.plane = torch.nn.Parameter(torch.rand(1, 32, 128, 128).contiguous(memory_format=torch.channels_last))   # >= 2^18: own all-reduce
    m.grid = torch.nn.Parameter(torch.rand(1, 8, 32, 32, 32).contiguous(memory_format=torch.channels_last_3d))  # Cobafa-style 5-D grid, 2^18
    m.lin = torch.nn.Linear(96, 64)                                                                           # small: bucketed
    m.head = torch.nn.Linear(64, 1)
    return m


def _worker(rank, world, port, q, flat=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from tinynerf_amd.run import Trainer
    m = _make_model()
    stub = _Stub(m, world)
    # different ray counts per rank (dynamic batching): 5 and 9 rays
    n = 5 + 4 * rank
    g = torch.Generator().manual_seed(100 + rank)
    rendered_in = torch.rand(n, 96, generator=g)
    target = torch.rand(n, 3, generator=g)
    rendered = m.head(torch.relu(m.lin(rendered_in))).expand(n, 3) * m.plane[0, :3, 40, 7] + m.grid[0, :3, 1, 2, 3] * rendered_in[:, :3]
    loss = Trainer.global_mse(stub, rendered, target)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    if flat:                                              # the harness: every small gradient is a view into one flat bucket
        stub._flat, stub._flat_used, stub._flat_ids = Trainer._flat_small_grads(list(m.parameters()), stub.device)
        assert stub._flat_ids == {id(p) for p in (m.lin.weight, m.lin.bias, m.head.weight, m.head.bias)}
        assert all(p.grad.data_ptr() % 256 == stub._flat.data_ptr() % 256 for p in m.lin.parameters())
    loss.backward()
    with torch.no_grad():
        m.plane.grad[:, :, :10] = 0; m.plane.grad[:, :, 100:] = 0     # outside the live rows the gradient is zero on every rank
    assert m.plane.grad.is_contiguous(memory_format=torch.channels_last)
    assert m.grid.grad.is_contiguous(memory_format=torch.channels_last_3d) and not m.grid.grad.is_contiguous()
    for g in (m.plane.grad, m.grid.grad):                 # the exchange works on the parameter's own memory, no copies
        v = Trainer._dense_view(g)
        assert v is not None and v.is_contiguous() and v.data_ptr() == g.data_ptr() and v.numel() == g.numel()
    view = Trainer._reduce_view(stub, m.plane.grad, 0)
    assert view.shape == (1, 90, 128, 32) and view.is_contiguous() and view.data_ptr() == m.plane.grad[:, :, 10:].data_ptr()
    Trainer._planes_ready(stub, [m.plane.grad])          # the fused node starts the plane all-reduces mid-backward
    assert len(stub._early) == 1
    # the "Empty iteration" gate rides with the bucket: rank 0's step reached nothing, rank 1's did -> not empty anywhere
    gate = torch.tensor([0.0 if rank == 0 else 0.7])
    Trainer.all_reduce_grads(stub, gate)                  # ... which are awaited here, everything else is reduced now
    assert not stub._early and float(gate) > 0
    # a second exchange with every rank's step empty: the gate stays 0.  (Each rank holds the sum S already: world x (S / world)
    # = S, exactly for a power-of-two world, so the gradients compared below are unchanged; this time the plane goes through the
    # non-early branch.)
    gate0 = torch.zeros(1)
    for p_ in m.parameters():
        p_.grad.mul_(1.0 / world)
    Trainer.all_reduce_grads(stub, gate0)
    assert float(gate0) == 0.0
    q.put((rank, float(loss.detach()), {k: p.grad.detach().contiguous().numpy().copy() for k, p in m.named_parameters()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("flat,world", [(True, 2), (False, 2), (True, 4)])
def test_gradient_exchange_equals_single_process(flat, world):
    """2 and 4 gloo ranks (the 8-GPU job's exchange code with more than one peer: flat bucket + gate slot, live-row plane
    slices, early handles) against one process on the union of the ranks' rays"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, flat)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=100) for _ in procs], key=lambda t: t[0])
    for p in procs: p.join(timeout=30)
    assert all(p.exitcode == 0 for p in procs)
    # single-process reference: concatenate both ranks' rays, plain mean
    m = _make_model()
    ins, tgts = [], []
    for rank in range(world):
        n = 5 + 4 * rank
        g = torch.Generator().manual_seed(100 + rank)
        ins.append(torch.rand(n, 96, generator=g)); tgts.append(torch.rand(n, 3, generator=g))
    x, t = torch.cat(ins), torch.cat(tgts)
    rendered = m.head(torch.relu(m.lin(x))).expand(x.size(0), 3) * m.plane[0, :3, 40, 7] + m.grid[0, :3, 1, 2, 3] * x[:, :3]
    loss = torch.nn.functional.mse_loss(rendered, t)
    loss.backward()
    with torch.no_grad():
        m.plane.grad[:, :, :10] = 0; m.plane.grad[:, :, 100:] = 0
    assert abs(sum(r[1] for r in res) - float(loss)) < 1e-6          # local losses sum to the global mean
    for k, p in m.named_parameters():
        for rank in range(world):                                       # every rank holds the full-batch gradient
            torch.testing.assert_close(torch.from_numpy(res[rank][2][k]), p.grad.contiguous(), rtol=1e-5, atol=1e-7)


def _sharded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from tinynerf_amd.run import Trainer
    torch.manual_seed(0)
    p = torch.rand(1, 32, 64, 48).contiguous(memory_format=torch.channels_last)            # the same parameters on every rank
    g = torch.rand(1, 32, 64, 48, generator=torch.Generator().manual_seed(50 + rank)).contiguous(memory_format=torch.channels_last)
    total = g.clone()
    torch.distributed.all_reduce(total)                                                    # the oracle: what an all-reduce would give
    Trainer._reduce_scatter_rows(g, rank, world).wait()
    r0, r1 = Trainer._own_rows(64, rank, world)
    assert r1 - r0 == 64 // world
    torch.testing.assert_close(g[:, :, r0:r1], total[:, :, r0:r1], rtol=1e-6, atol=1e-6)   # this rank's rows hold the sum over ranks
    with torch.no_grad():                                                                  # "optimizer pass" on the rank's rows only
        p[:, :, r0:r1] -= 0.1 * g[:, :, r0:r1]
    for w in Trainer._all_gather_rows(p, rank, world):
        w.wait()
    q.put((rank, p.contiguous().numpy().copy(), total.contiguous().numpy().copy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_exchange_equals_all_reduce(world):
    """TrainConfig.sharded_optimizer (round 5): reduce-scatter of a plane gradient into the rank's own rows -> update of those
    rows -> all-gather of the updated rows, against all-reduce + the same update of every row on every rank: identical parameters on
    all ranks (bit for bit between ranks), on the product's own Trainer methods (in place on the channels_last memory, no copies)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=100) for _ in procs], key=lambda t: t[0])
    for p in procs: p.join(timeout=30)
    assert all(p.exitcode == 0 for p in procs)
    torch.manual_seed(0)
    p0 = torch.rand(1, 32, 64, 48).contiguous(memory_format=torch.channels_last)
    want = (p0 - 0.1 * torch.from_numpy(res[0][2])).contiguous().numpy()
    for rank in range(world):
        assert (res[rank][1] == res[0][1]).all()
        torch.testing.assert_close(torch.from_numpy(res[rank][1]), torch.from_numpy(want), rtol=1e-6, atol=1e-6)
