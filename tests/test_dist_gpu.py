"""GPU, world_size 2 on ONE MI355X (both ranks on cuda:0, gloo backend carrying the CUDA tensors): the TrainEngine's data-parallel
step itself -- flat gradient all-reduce / world folded into the optimizer kernel, two-bucket overlapped exchange -- with real
ranks holding different clips.  (RCCL refuses two ranks on one device; the collective semantics under test are torch.distributed's,
the same calls the nccl backend receives on a multi-GPU node.)  Reference: codes/core/dist_utils.py:15-67."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import numpy as np
    import mvfnet_amd
    from mvfnet_amd import synth
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0), None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    m = m.cuda().train()
    eng = m.train_engine(dtype=torch.bfloat16)
    eng.overlap_allreduce = overlap
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64, seed=10 + rank)).cuda()          # every rank its own clips
    labels = torch.from_numpy(synth.synth_labels(2, seed=10 + rank)).cuda()
    losses = [float(eng.train_step(imgs, labels)) for _ in range(3)]
    torch.cuda.synchronize()
    bits = eng.flat_params.view(torch.int32).to(torch.int64)
    chk = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=bits.device) % 8191 + 1)).sum()]).cpu()
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    g0 = eng.flat_grads[:1000].cpu().numpy().copy()
    q.put((rank, losses, [c.tolist() for c in allc], g0.tolist(), float(eng.norm_out[0])))
    dist.barrier()
    dist.destroy_process_group()


def _run(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=280) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


@pytest.mark.timeout(600)
def test_two_rank_train_steps_keep_replicas_bit_identical_and_buckets_equal_flat():
    a = _run(True)           # tail bucket all-reduced during backward + head bucket after it
    b = _run(False)          # one flat all-reduce after backward
    for out in (a, b):
        (r0, l0, c0, g0, n0), (r1, l1, c1, g1, n1) = out
        assert c0[0] == c0[1] == c1[0] == c1[1]                   # both ranks hold the same parameter bits after 3 steps
        assert g0 == g1 and n0 == n1                              # ... because both applied the same summed gradient / clip norm
        assert l0 != l1                                           # different clips per rank: the losses differ
        assert all(x == x for x in l0 + l1)
    assert a[0][2][0] == b[0][2][0]                               # two-bucket overlapped exchange == single collective, bit for bit
