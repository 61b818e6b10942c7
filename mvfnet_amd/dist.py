"""Data-parallel plumbing: one process per GPU over torch.distributed ('nccl' = RCCL on ROCm; 'gloo' in CPU tests).

Mirror of the reference's `codes/core/dist_utils.py` (init_dist :70-92, allreduce_grads :38-49, _allreduce_coalesced
:15-35, DistOptimizerHook :52-67, get_dist_info :116-131) and `codes/core/parallel/distributed.py`
(MMDistributedDataParallel :11-62: broadcast of parameters + buffers at wrap time, no autograd hooks).

MI355X notes: xGMI is point-to-point (7 links x ~153 GB/s per GPU); the R50 gradient is 97 MB fp32.  With the HIP
TrainEngine every gradient already lives in ONE flat buffer, so the reference's "one flat bucket" semantics
(bucket_size_mb=-1) is a single in-place all-reduce with no flatten/unflatten copies; `bucket_size_mb > 0` splits that
buffer into contiguous chunks issued back to back (RCCL pipelines them over the rings).
"""
import os

import torch
import torch.distributed as dist


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist(launcher="pytorch", backend="nccl", **kwargs):
    """reference dist_utils.py:70-92 ('pytorch' launcher: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env)."""
    if launcher != "pytorch":
        raise ValueError("Invalid launcher type: %s (only 'pytorch' is built)" % launcher)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend == "nccl":
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dist.init_process_group(backend=backend, **kwargs)
    return rank, dist.get_world_size()


def _buckets(tensors, bucket_size_mb):
    if bucket_size_mb and bucket_size_mb > 0:
        cap = int(bucket_size_mb * 1024 * 1024)
        cur, size, out = [], 0, []
        for t in tensors:
            nb = t.numel() * t.element_size()
            if cur and (size + nb > cap or t.dtype != cur[0].dtype):
                out.append(cur)
                cur, size = [], 0
            cur.append(t)
            size += nb
        if cur:
            out.append(cur)
        return out
    by_type = {}
    for t in tensors:                       # reference default: one bucket per tensor type (dist_utils.py:21-27)
        by_type.setdefault((t.dtype, t.device), []).append(t)
    return list(by_type.values())


def allreduce_coalesced(tensors, world_size, bucket_size_mb=-1):
    for bucket in _buckets(tensors, bucket_size_mb):
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat)
        flat.div_(world_size)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def allreduce_grads(params, coalesce=True, bucket_size_mb=-1):
    """Average the gradients of `params` over all ranks (reference dist_utils.py:38-49)."""
    grads = [p.grad.data for p in params if p.requires_grad and p.grad is not None]
    _, world = get_dist_info()
    if world == 1:
        return
    if coalesce:
        allreduce_coalesced(grads, world, bucket_size_mb)
    else:
        for g in grads:
            dist.all_reduce(g.div_(world))


def allreduce_flat(flat, world_size=None, bucket_size_mb=-1):
    """In-place average of one flat gradient buffer (the TrainEngine layout): no flatten/unflatten copies."""
    _, world = get_dist_info()
    world = world_size or world
    if world == 1:
        return flat
    if bucket_size_mb and bucket_size_mb > 0:
        step = max(1, int(bucket_size_mb * 1024 * 1024) // flat.element_size())
        works = [dist.all_reduce(flat[i:i + step], async_op=True) for i in range(0, flat.numel(), step)]
        for w in works:
            w.wait()
    else:
        dist.all_reduce(flat)
    flat.div_(world)
    return flat


class MMDistributedDataParallel(torch.nn.Module):
    """reference parallel/distributed.py:11-62: broadcast parameters and buffers from rank 0 at wrap time; forward just
    calls the module (inputs are expected on this rank's device); gradients are exchanged by DistOptimizerHook."""

    def __init__(self, module, dim=0, broadcast_buffers=True, bucket_cap_mb=25):
        super().__init__()
        self.module, self.dim, self.broadcast_buffers = module, dim, broadcast_buffers
        self.broadcast_bucket_size = bucket_cap_mb * 1024 * 1024
        self._sync_params()

    def _sync_params(self):
        _, world = get_dist_info()
        if world == 1:
            return
        tensors = list(self.module.state_dict().values()) if self.broadcast_buffers else [p.data for p in self.module.parameters()]
        for t in tensors:
            if t.numel():
                dist.broadcast(t, 0)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)


def _engine_backed_iter(optimizer, loss, grad_clip, exchange):
    """The hook sequence for `build_optimizer`'s EngineSGD (the reference's _dist_train pairing, train.py:159-196): its step() updates
    from the engine's FLAT gradient buffer, which `loss.backward()` has just filled -- the parameters' .grad are autograd's copies, so
    an all-reduce or clip_grad_norm_ on them would never reach the update.  The exchange (/ world), the global L2 clip and the
    update therefore all run inside the engine on the flat buffer: one collective, one fused kernel.  Returns the pre-clip norm,
    as clip_grad_norm_ does."""
    if grad_clip is not None and grad_clip.get("norm_type", 2) != 2:
        raise NotImplementedError("grad_clip norm_type %r: the fused clip is the L2 norm" % grad_clip.get("norm_type"))
    optimizer.zero_grad()
    loss.backward()
    eng = optimizer.engine
    keep_clip, keep_x = eng.max_norm, eng.exchange_enabled
    eng.max_norm = grad_clip.get("max_norm") if grad_clip is not None else None
    eng.exchange_enabled = bool(exchange)
    try:
        norm = optimizer.step()
    finally:
        eng.max_norm, eng.exchange_enabled = keep_clip, keep_x
    return norm[0] if grad_clip is not None and norm is not None else None


class DistOptimizerHook(object):
    """after_train_iter of the reference hook (dist_utils.py:52-67): zero_grad -> backward -> all-reduce / world ->
    clip_grad_norm_ -> optimizer.step(), for any torch optimizer; `loss` comes from Recognizer2D.forward_train.  With the
    engine-backed optimizer of `runner.build_optimizer` the same sequence runs on the engine's flat gradient buffer."""

    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1):
        self.grad_clip, self.coalesce, self.bucket_size_mb = grad_clip, coalesce, bucket_size_mb

    def after_train_iter(self, model, optimizer, loss):
        if hasattr(optimizer, "engine"):
            return _engine_backed_iter(optimizer, loss, self.grad_clip, True)
        optimizer.zero_grad()
        loss.backward()
        allreduce_grads(model.parameters(), self.coalesce, self.bucket_size_mb)
        total = None
        if self.grad_clip is not None:
            total = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], **self.grad_clip)
        optimizer.step()
        return total


class Fp16OptimizerHook(DistOptimizerHook):
    """The reference's mixed-precision hook (codes/core/fp16/hooks.py:12-136: fp16 model copy, fp32 master weights in the optimizer,
    loss scaling, gradients copied / un-scaled / clipped in fp32, weights copied back) mapped onto this build's mixed-precision mode:
    the train engine's bf16 STORAGE -- activations and packed conv weights in bf16, accumulation / BatchNorm statistics / every
    parameter gradient / master weights / optimizer state in fp32 (what the reference keeps in fp32, plus the gradients).  bf16 has
    fp32's exponent range, so `loss_scale` is accepted and not needed; there is no fp16 copy of the model to synchronise.

    before_run(model): build the model's train engine in bf16 (call before the first forward_train; a model whose engine already
    exists in fp32 is refused) and set the `fp16_enabled` flags the reference's wrap_fp16_model sets (hooks.py:108-112).
    after_train_iter: the DistOptimizerHook sequence; the all-reduce is skipped when distributed=False (hooks.py:84-86)."""

    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1, loss_scale=512.0, distributed=True):
        super().__init__(grad_clip, coalesce, bucket_size_mb)
        self.loss_scale, self.distributed = loss_scale, distributed

    def before_run(self, model):
        model = model.module if hasattr(model, "module") else model
        model.train_engine(dtype=torch.bfloat16)
        for m in model.modules():
            if hasattr(m, "fp16_enabled"):
                m.fp16_enabled = True
        return model

    def after_train_iter(self, model, optimizer, loss):
        if hasattr(optimizer, "engine"):
            return _engine_backed_iter(optimizer, loss, self.grad_clip, self.distributed)
        optimizer.zero_grad()
        loss.backward()
        if self.distributed:
            allreduce_grads(model.parameters(), self.coalesce, self.bucket_size_mb)
        total = None
        if self.grad_clip is not None:
            total = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], **self.grad_clip)
        optimizer.step()
        return total
