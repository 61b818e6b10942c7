"""CPU, world_size 2 over gloo: the data-parallel plumbing (mirror of codes/core/dist_utils.py and
codes/core/parallel/distributed.py) -- parameter broadcast, flat / bucketed gradient averaging, the optimizer hook order
(backward -> all-reduce / world -> clip -> step) against a single-process run on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mvfnet_amd import dist as D
    r, w = D.init_dist("pytorch", backend="gloo")
    assert (r, w) == (rank, world) and D.get_dist_info() == (rank, world)
    torch.manual_seed(100 + rank)                          # different init per rank: the wrapper must broadcast rank 0's
    net = nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.Linear(5, 3))
    ddp = D.MMDistributedDataParallel(net)
    ref0 = [t.clone() for t in net.state_dict().values()]
    gathered = [None] * world
    dist.all_gather_object(gathered, [t.tolist() for t in ref0])
    assert gathered[0] == gathered[1]
    # data: each rank gets half of a fixed batch
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
    xs, ys = X[rank::world], Y[rank::world]
    net.eval()                                              # keep BN out of the comparison (per-rank statistics differ by design)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    hook = D.DistOptimizerHook(grad_clip=dict(max_norm=0.5, norm_type=2), coalesce=True, bucket_size_mb=-1)
    loss = nn.functional.cross_entropy(ddp(xs), ys)
    total = hook.after_train_iter(net, opt, loss)
    # flat-buffer variant used by the HIP TrainEngine, incl. bucketed async chunks
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    a = D.allreduce_flat(flat.clone())
    b = D.allreduce_flat(flat.clone(), bucket_size_mb=1e-5)
    assert torch.allclose(a, torch.arange(10, dtype=torch.float32) * 1.5) and torch.equal(a, b)
    # the TrainEngine's two-bucket exchange: tail slice asynchronously first (while backward still runs), head slice later --
    # the same two collectives in the same order on every rank must equal the single flat all-reduce
    fl = (torch.arange(16, dtype=torch.float32) + 1.0) * (rank + 1)
    one = fl.clone()
    dist.all_reduce(one)
    two = fl.clone()
    tail_off = 5
    work = dist.all_reduce(two[tail_off:], async_op=True)
    two[:tail_off] *= 1.0                                   # "backward" keeps writing the head slice meanwhile
    dist.all_reduce(two[:tail_off])
    work.wait()
    assert torch.equal(one, two)
    # bucketed coalesced path gives the same average
    for p in net.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    D.allreduce_grads(net.parameters(), coalesce=True, bucket_size_mb=1e-4)
    assert all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in net.parameters())
    # multi_gpu_test's result gather (reference codes/core/test.py:147-185) with UNEVEN shards: rank::world interleave, padding
    # dropped at `size`; then with an EMPTY shard on rank 1 (fewer videos than ranks), which must neither hang nor mis-shape
    import numpy as np
    from mvfnet_amd.runner import collect_results
    rows = {0: [np.full((1, 4), 10.0 + i, np.float32) for i in range(3)], 1: [np.full((1, 4), 20.0 + i, np.float32) for i in range(2)]}[rank]
    got = collect_results(rows, size=5)
    if rank == 0:
        assert [float(r[0, 0]) for r in got] == [10.0, 20.0, 11.0, 21.0, 12.0]
    else:
        assert got is None
    got = collect_results(rows if rank == 0 else [], size=None)
    if rank == 0:
        assert [float(r[0, 0]) for r in got] == [10.0, 11.0, 12.0] and got[0].shape == (1, 4)
    if rank == 0:
        q.put(([p.detach().tolist() for p in net.parameters()], float(total), [t.tolist() for t in ref0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_ddp_plumbing_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    params, total, init = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference on the full batch with the same (rank-0) init: mean of the two half-batch grads = full-batch grad
    net = nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.Linear(5, 3))
    sd0 = net.state_dict()
    net.load_state_dict({k: torch.tensor(v, dtype=sd0[k].dtype) for k, v in zip(sd0.keys(), init)})
    net.eval()
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    opt.zero_grad()
    nn.functional.cross_entropy(net(X), Y).backward()
    ref_total = torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=0.5, norm_type=2)
    opt.step()
    assert abs(float(ref_total) - total) < 1e-5
    for a, b in zip(params, net.parameters()):
        assert torch.allclose(torch.tensor(a), b.detach(), atol=1e-6)
