"""mmcv-free runner shell around the HIP engines: the host-side control flow of the reference's `codes/core/train.py`
(set_random_seed :23-29, parse_losses :32-49, batch_processor :52-60, train_network :63-76, build_optimizer :79-156,
_dist_train / _non_dist_train :159-252) and `codes/core/test.py` (single_gpu_test :12-39, multi_gpu_test :42-89,
collect_results_gpu :147-185), with the schedule the shipped config uses (lr_config: step [90,130] x0.1, linear warm-up 25070
iters from ratio 0.01; mmcv 0.4.3 semantics).  `train_recognizer.py` needs only its imports swapped:

    from mvfnet_amd.runner import init_dist, set_random_seed, train_network      # was: from codes.core import ...
    from mvfnet_amd import build_recognizer                                      # was: from codes.models import ...

What is different by design: the optimizer is not a torch.optim object stepping 161 tensors but a facade (`EngineSGD`) over the
train engine's fused clip + SGD kernel on ONE flat parameter buffer -- same hyper-parameters, same `state_dict()` wire format
(torch.optim.SGD's), same per-parameter `paramwise_options`; `fp16 = dict(loss_scale=...)` in a config selects the engine's
bf16 storage mode (fp32 master weights / gradients / statistics as the reference's Fp16OptimizerHook keeps them,
codes/core/fp16/hooks.py:12-136; bf16's exponent range makes the loss scale a no-op, it is accepted and ignored).
Host batches are staged through pinned memory and uploaded on a copy stream one batch ahead (the reference's scatter
puts that copy on the critical path, parallel/_functions.py:21-26)."""
import os
import random
import re

import numpy as np
import torch
import torch.distributed as dist

from .checkpoint import load_checkpoint, save_checkpoint
from .dist import get_dist_info, init_dist  # noqa: F401  (re-exported: train_recognizer.py imports init_dist from the core package)


def set_random_seed(seed):
    """reference train.py:23-29."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


class Config(dict):
    """Attribute-style dict (what train_network reads from an mmcv Config: cfg.optimizer, cfg.get('fp16'), ...)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

    def __setattr__(self, k, v):
        self[k] = v


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if not hasattr(cfg, "get") else cfg.get(key, default)


def step_lr(base_lr, epoch, it, steps=(90, 130), gamma=0.1, warmup="linear", warmup_iters=25070, warmup_ratio=0.01):
    """lr at (epoch, global iteration) -- mmcv LrUpdaterHook 'step' policy + linear warm-up."""
    lr = base_lr * (gamma ** sum(1 for s in steps if epoch >= s))
    if warmup is None or it >= warmup_iters:
        return lr
    if warmup == "linear":
        return lr * (1 - (1 - it / float(warmup_iters)) * (1 - warmup_ratio))
    if warmup == "constant":
        return lr * warmup_ratio
    if warmup == "exp":
        return lr * warmup_ratio ** (1 - it / float(warmup_iters))
    raise NotImplementedError("lr_config warmup %r (mmcv LrUpdaterHook knows None, 'constant', 'linear', 'exp')" % (warmup,))


def parse_losses(losses):
    """reference train.py:32-49: mean of every tensor, 'loss' = sum of the entries whose key contains 'loss'."""
    log_vars = {}
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, (list, tuple)):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError("%s is not a tensor or list of tensors" % name)
    loss = sum(v for k, v in log_vars.items() if "loss" in k)
    log_vars["loss"] = loss
    return loss, {k: float(v.detach()) for k, v in log_vars.items()}


def batch_processor(model, data, train_mode=True):
    losses = model(**data)
    loss, log_vars = parse_losses(losses)
    return dict(loss=loss, log_vars=log_vars, num_samples=len(data["img_group"]))


# ------------------------------------------------------------------------------------------------ optimizer
def paramwise_multipliers(model, paramwise_options):
    """{parameter: (lr_mult, decay_mult)} by the reference's rules (train.py:117-153): BatchNorm / GroupNorm weights and biases
    (names matching (bn|gn)(\\d+)?.(weight|bias)) get weight_decay * norm_decay_mult; every other `.bias` gets lr * bias_lr_mult
    and weight_decay * bias_decay_mult; parameters with requires_grad False keep the global setting."""
    bias_lr_mult = paramwise_options.get("bias_lr_mult", 1.0)
    bias_decay_mult = paramwise_options.get("bias_decay_mult", 1.0)
    norm_decay_mult = paramwise_options.get("norm_decay_mult", 1.0)
    out = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if re.search(r"(bn|gn)(\d+)?.(weight|bias)", name):
            out[p] = (1.0, norm_decay_mult)
        elif name.endswith(".bias"):
            out[p] = (bias_lr_mult, bias_decay_mult)
    return out


class EngineSGD(object):
    """The optimizer object `build_optimizer` returns: torch.optim.SGD's surface (param_groups, state_dict, load_state_dict,
    zero_grad, step) over the train engine's flat buffers.  `step()` = all-reduce / world (when a process group is up) + global
    L2 clip (if `grad_clip` was given) + the SGD update, one fused launch sequence."""

    def __init__(self, model, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, paramwise=None, dtype=None):
        if dampening:
            raise NotImplementedError("SGD dampening != 0 is not built")
        model = model.module if hasattr(model, "module") else model
        opt = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, max_norm=None)
        if dtype is not None:
            opt["dtype"] = dtype
        self.model, self.engine = model, model.train_engine(**opt)
        self.grad_clip = None           # dict(max_norm=..., norm_type=2), set by train_network / the optimizer hooks (dist.py)
        self.engine.max_norm = None
        self.engine.nesterov = bool(nesterov)
        self.engine.set_param_options(paramwise)
        self.param_groups = [dict(lr=lr, initial_lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay, nesterov=bool(nesterov),
                                  params=list(model.parameters()))]

    def zero_grad(self):
        """The engine's backward overwrites the whole flat gradient every step, so nothing is zeroed there; the `.grad` copies that
        autograd hands to the parameters under the hook flow (dist._engine_backed_iter: loss.backward()) are dropped so that they do
        not pile up as an ever-growing sum (torch.optim's set_to_none behaviour)."""
        for p in self.model.parameters():
            p.grad = None

    def step(self, lr=None):
        g = self.param_groups[0]
        self.engine.momentum, self.engine.weight_decay = g["momentum"], g["weight_decay"]
        return self.engine.step(g["lr"] if lr is None else lr)

    def state_dict(self):
        self.engine.lr = self.param_groups[0]["lr"]
        self.engine.initial_lr = self.param_groups[0].get("initial_lr", self.engine.lr)
        return self.engine.optimizer_state_dict()

    def load_state_dict(self, sd):
        self.engine.load_optimizer_state_dict(sd)
        g = self.param_groups[0]
        g.update(lr=self.engine.lr, momentum=self.engine.momentum, weight_decay=self.engine.weight_decay, nesterov=self.engine.nesterov)


def build_optimizer(model, optimizer_cfg, dtype=None):
    """reference train.py:79-156 for the optimizer the MVFNet configs use: dict(type='SGD', lr, momentum, weight_decay, nesterov
    [, paramwise_options=dict(bias_lr_mult, bias_decay_mult, norm_decay_mult)])."""
    cfg = dict(optimizer_cfg)
    kind = cfg.pop("type", "SGD")
    if kind != "SGD":
        raise NotImplementedError("optimizer type %r: the fused HIP optimizer is SGD (the MVFNet configs' choice)" % kind)
    paramwise = cfg.pop("paramwise_options", None)
    target = model.module if hasattr(model, "module") else model
    mult = None
    if paramwise is not None:
        if not isinstance(paramwise, dict):
            raise TypeError("paramwise_options must be a dict")
        if ("bias_decay_mult" in paramwise or "norm_decay_mult" in paramwise) and cfg.get("weight_decay") is None:
            raise ValueError("weight_decay must be given explicitly when a decay multiplier is")        # train.py:123-125
        mult = paramwise_multipliers(target, paramwise)
    return EngineSGD(target, paramwise=mult, dtype=dtype, **cfg)


# ------------------------------------------------------------------------------------------------ input upload
class DevicePrefetcher(object):
    """Wraps a loader of host batches (dicts of CPU tensors): each batch is copied into pinned staging memory and uploaded on a
    copy stream ONE BATCH AHEAD of the consumer, so the H2D copy (154 MB fp32 / 38.5 MB uint8 per 32-clip step) overlaps the
    previous step's kernels instead of sitting at the head of the step (reference: MMDistributedDataParallel.scatter on the
    compute stream's critical path, parallel/distributed.py:54-62).  Batches already on the device pass through.
    [r3] A batch that is already pinned (DataLoader(pin_memory=True)) is uploaded straight from where it lies -- the staging memcpy of
    154 MB costs the host ~12 ms per step, more than the upload itself -- and lands in one of two PERSISTENT device buffers per key
    (no allocator traffic); a slot is refilled only after the step that read it (an event on the consumer's stream, waited for by
    the copy stream, never by the host).  The yielded tensors are therefore valid until the batch after the next is requested."""

    def __init__(self, loader, device=None):
        self.loader, self.device = loader, torch.device(device or "cuda")
        self.sampler = getattr(loader, "sampler", None)
        self.dataset = getattr(loader, "dataset", None)
        self._copy = None
        self._pinned = [{}, {}]
        self._device = [{}, {}]
        self._consumed = [None, None]      # the consumer's last kernel that READ each device slot: the next upload into it waits for that
        self._slot_event = [None, None]      # the upload that last READ each staging slot: must be complete before the host refills it

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch, slot):
        out, any_host = {}, False
        if self._slot_event[slot] is not None:
            self._slot_event[slot].synchronize()      # two batches old: normally long done, never skipped
        if self._consumed[slot] is not None:          # the device buffers of this slot were read by the step two batches ago:
            torch.cuda.current_stream().wait_event(self._consumed[slot])      # the copy stream waits for that step, the host does not
        for k, v in batch.items():
            if not isinstance(v, torch.Tensor) or v.is_cuda:
                out[k] = v
                continue
            any_host = True
            src = v
            if not v.is_pinned():                     # pageable memory: stage through this slot's pinned buffer (a loader with pin_memory=True skips this copy)
                pin = self._pinned[slot].get(k)
                if pin is None or pin.shape != v.shape or pin.dtype != v.dtype:
                    pin = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                    self._pinned[slot][k] = pin
                pin.copy_(v)
                src = pin
            dev = self._device[slot].get(k)           # persistent device-side landing buffers, two slots: no allocator traffic per step
            if dev is None or dev.shape != v.shape or dev.dtype != v.dtype:
                dev = torch.empty(v.shape, dtype=v.dtype, device=self.device)
                self._device[slot][k] = dev
            dev.copy_(src, non_blocking=True)
            out[k] = dev
        ev = None
        if any_host:
            ev = torch.cuda.Event()
            ev.record()
        self._slot_event[slot] = ev
        return out, ev

    def __iter__(self):
        if self.device.type != "cuda":
            for b in self.loader:
                yield b
            return
        if self._copy is None:
            self._copy = torch.cuda.Stream(device=self.device)
        it = iter(self.loader)
        slot, nxt = 0, None

        def fetch():
            try:
                b = next(it)
            except StopIteration:
                return None
            with torch.cuda.stream(self._copy):
                return self._upload(b, slot)

        nxt = fetch()
        while nxt is not None:
            cur, ev = nxt
            cur_slot = slot
            slot ^= 1
            nxt = fetch()                                  # the next batch's copy is in flight while the consumer runs this one
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            yield cur
            if ev is not None:                             # the consumer has queued its work on this batch: mark the slot's last reader
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream())
                self._consumed[cur_slot] = done


def build_dataloader(dataset, videos_per_gpu, workers_per_gpu=0, dist_mode=False, shuffle=True, drop_last=False, pin_memory=None):
    """A torch DataLoader over `dataset` (items: dict(img_group=tensor, label=tensor)): DistributedSampler (rank::world, as the
    reference's sampler.py:62-78 deals videos) when distributed.  Like the reference's build_dataloader (datasets/builder.py) the
    last, partial batch of an epoch is kept (drop_last=False: the engine's buffers are keyed by shape) -- the iteration count that
    drives the warm-up is the reference's.  pin_memory defaults to the reference's True (datasets/loader/build_loader.py:22) whenever a GPU
    is present: DevicePrefetcher then uploads straight from the loader's pinned batch (no staging memcpy).  Datasets / decode pipelines
    themselves are out of scope."""
    if pin_memory is None:
        pin_memory = torch.cuda.is_available()
    if isinstance(dataset, (torch.utils.data.DataLoader, list, tuple)) or not hasattr(dataset, "__getitem__"):
        return dataset                                     # already a loader / a list or iterable of ready batches
    sampler = None
    if dist_mode:
        rank, world = get_dist_info()
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=shuffle)
    return torch.utils.data.DataLoader(dataset, batch_size=videos_per_gpu, sampler=sampler, shuffle=(shuffle and sampler is None),
                                       num_workers=workers_per_gpu, pin_memory=pin_memory, drop_last=drop_last)


# ------------------------------------------------------------------------------------------------ runner
class Runner(object):
    """Epoch/iteration loop with the reference's hook order: lr update -> forward -> backward -> all-reduce/world ->
    clip -> step -> checkpoint every `ckpt_interval` epochs.  Uses the fused HIP TrainEngine (one flat all-reduce +
    one optimizer kernel)."""

    def __init__(self, model, work_dir=None, lr=0.015, momentum=0.9, weight_decay=1e-4, max_norm=40.0, lr_steps=(90, 130),
                 warmup_iters=25070, warmup_ratio=0.01, ckpt_interval=10, log_interval=20, logger=print, optimizer=None, dtype=None,
                 warmup="linear", lr_gamma=0.1):
        self.model, self.work_dir = model, work_dir
        if optimizer is not None:                    # build_optimizer's object: hyper-parameters and param-wise options live there
            self.engine = optimizer.engine
            g = optimizer.param_groups[0]
            lr = g["lr"]
        else:
            opt = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, max_norm=max_norm)
            if dtype is not None:
                opt["dtype"] = dtype
            self.engine = model.train_engine(**opt)
        self.optimizer = optimizer
        self.engine.max_norm = max_norm
        self.engine.initial_lr = lr          # the schedule's base rate: checkpoints carry it beside the current one (mmcv's 'initial_lr')
        self.base_lr, self.lr_steps, self.lr_gamma = lr, tuple(lr_steps), lr_gamma
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        self.ckpt_interval, self.log_interval, self.log = ckpt_interval, log_interval, logger
        self.epoch, self.iter = 0, 0
        self.hooks = []               # objects with after_train_epoch(runner), e.g. evaluation.EvalTopKAccuracyHook

    def register_hook(self, hook):
        self.hooks.append(hook)
        return hook

    def current_lr(self):
        return step_lr(self.base_lr, self.epoch, self.iter, self.lr_steps, self.lr_gamma, self.warmup, self.warmup_iters, self.warmup_ratio)

    def train_epoch(self, loader):
        self.model.train()
        rank, _ = get_dist_info()
        for data in loader:
            lr = self.current_lr()
            loss = self.engine.train_step(data["img_group"], data["label"], lr=lr)
            self.iter += 1
            if rank == 0 and self.log_interval and self.iter % self.log_interval == 0:
                self.log("Epoch [%d] iter %d lr %.5f loss_cls %.4f grad_norm %.3f" % (
                    self.epoch + 1, self.iter, lr, float(loss), float(self.engine.norm_out[0])))
        self.epoch += 1
        if rank == 0 and self.work_dir and self.ckpt_interval and self.epoch % self.ckpt_interval == 0:
            self.save_checkpoint()
        for h in self.hooks:
            out = h.after_train_epoch(self)
            if out and rank == 0 and self.log:
                self.log("Epoch(val) [%d] %s" % (self.epoch, "  ".join("%s: %.4f" % kv for kv in out.items() if kv[0] != "epoch")))

    def run(self, loader, max_epochs):
        if isinstance(loader, (list, tuple)) and loader and not isinstance(loader[0], dict):
            loader = loader[0]        # mmcv Runner.run(data_loaders, workflow, max_epochs): the train loader is first (a list of dicts is a loader of batches)
        while self.epoch < max_epochs:
            if hasattr(getattr(loader, "sampler", None), "set_epoch"):
                loader.sampler.set_epoch(self.epoch)            # DistSamplerSeedHook
            self.train_epoch(loader)

    def save_checkpoint(self):
        """epoch_{n}.pth + latest.pth symlink; {'meta', 'state_dict', 'optimizer'} with the optimizer entry in torch.optim.SGD's
        state_dict layout (reference checkpoint.py:235-265, mmcv CheckpointHook)."""
        path = os.path.join(self.work_dir, "epoch_%d.pth" % self.epoch)
        self.engine.lr = self.current_lr()
        save_checkpoint(self.model, path, optimizer=self.engine.optimizer_state_dict(), meta=dict(epoch=self.epoch, iter=self.iter))
        latest = os.path.join(self.work_dir, "latest.pth")
        if os.path.lexists(latest):
            os.remove(latest)
        os.symlink(os.path.basename(path), latest)
        return path

    def resume(self, filename):
        """mmcv Runner.resume: weights, epoch / iter from meta, optimizer state (torch.optim.SGD's layout -- a checkpoint written
        by the reference resumes here and vice versa; round 1's own flat layout is still read)."""
        ckpt = load_checkpoint(self.model, filename, strict=True)
        meta = ckpt.get("meta", {})
        self.epoch, self.iter = meta.get("epoch", 0), meta.get("iter", 0)
        opt = ckpt.get("optimizer")
        if opt:
            lr, mom, wd = self.engine.lr, self.engine.momentum, self.engine.weight_decay
            self.engine.load_optimizer_state_dict(opt)
            self.engine.lr, self.engine.momentum, self.engine.weight_decay = lr, mom, wd      # the config's values win (lr follows the schedule)
        return ckpt


def as_config(cfg):
    """A plain dict, this package's Config, an mmcv Config (what the reference's train_recognizer.py passes: the settings live in
    `_cfg_dict`, `vars()` of it shows only _cfg_dict / _filename / _text) or any attribute namespace -> Config."""
    if isinstance(cfg, Config):
        return cfg
    if isinstance(cfg, dict):
        return Config(cfg)
    if hasattr(cfg, "_cfg_dict"):
        d = cfg._cfg_dict
        return Config(d.to_dict() if hasattr(d, "to_dict") else dict(d))
    if hasattr(cfg, "to_dict"):
        return Config(cfg.to_dict())
    return Config({k: v for k, v in vars(cfg).items() if not k.startswith("_")})


def train_network(model, dataset, cfg, distributed=False, validate=False, logger=None):
    """reference train.py:63-76 + _dist_train / _non_dist_train :159-252: loaders, model on the GPU (parameters broadcast from
    rank 0 when distributed), optimizer from cfg.optimizer, grad clip from cfg.optimizer_config, lr schedule from cfg.lr_config,
    checkpoints from cfg.checkpoint_config, logging interval from cfg.log_config, optional fp16 section, resume_from /
    load_from, then run cfg.total_epochs.  `dataset`: a torch Dataset of dict(img_group, label) items, a ready loader, or an
    iterable of batches (or a list whose first entry is the training one).  `validate` registers the reference's
    DistEvalTopKAccuracyHook(cfg.data.val, interval=cfg.eval_interval, k=(1, 5)) for a Dataset OBJECT under cfg.data.val."""
    cfg = as_config(cfg)
    log = (logger.info if logger is not None and hasattr(logger, "info") else (logger or print))
    is_list_of_sets = isinstance(dataset, (list, tuple)) and dataset and not isinstance(dataset[0], dict)     # (a list of dicts = ready batches)
    datasets = dataset if is_list_of_sets else [dataset]
    data = cfg.get("data") or {}
    loaders = [build_dataloader(ds, _cfg_get(data, "videos_per_gpu", 1), _cfg_get(data, "workers_per_gpu", 0), dist_mode=distributed)
               for ds in datasets]
    model = model.cuda()
    if distributed:
        from .dist import MMDistributedDataParallel
        MMDistributedDataParallel(model)                              # broadcast of parameters + buffers from rank 0
    fp16_cfg = cfg.get("fp16")
    dtype = torch.bfloat16 if fp16_cfg is not None else None         # fp16 section -> bf16 storage engine (loss_scale accepted, not needed)
    optimizer = build_optimizer(model, cfg.optimizer, dtype=dtype)
    ocfg = cfg.get("optimizer_config") or {}
    clip = _cfg_get(ocfg, "grad_clip")
    lrc = cfg.get("lr_config") or {}
    if _cfg_get(lrc, "policy", "step") != "step":
        raise NotImplementedError("lr_config policy %r: 'step' is built (the MVFNet configs' policy)" % _cfg_get(lrc, "policy"))
    steps = _cfg_get(lrc, "step", ())
    ck = cfg.get("checkpoint_config") or {}
    lg = cfg.get("log_config") or {}
    runner = Runner(model, cfg.get("work_dir"), max_norm=(_cfg_get(clip, "max_norm") if clip else None),
                    lr_steps=[steps] if isinstance(steps, int) else tuple(steps), warmup=_cfg_get(lrc, "warmup"),
                    warmup_iters=_cfg_get(lrc, "warmup_iters", 0), warmup_ratio=_cfg_get(lrc, "warmup_ratio", 0.1),
                    lr_gamma=_cfg_get(lrc, "gamma", 0.1), ckpt_interval=_cfg_get(ck, "interval", 0) or 0,
                    log_interval=_cfg_get(lg, "interval", 0) or 0, logger=log, optimizer=optimizer)
    if clip and _cfg_get(clip, "norm_type", 2) != 2:
        raise NotImplementedError("grad_clip norm_type %r: the fused clip is the L2 norm" % _cfg_get(clip, "norm_type"))
    if validate:
        # reference train.py:192-196: DistEvalTopKAccuracyHook(cfg.data.val, interval=cfg.eval_interval, k=(1, 5)).  Here cfg.data.val
        # must be the Dataset OBJECT (items dict(img_group=...), video_infos[i]['label']) -- the dataset classes a config dict would
        # name are out of scope; a ready hook under cfg.eval_hook is taken as is.
        if cfg.get("eval_hook") is not None:
            runner.register_hook(cfg.get("eval_hook"))
        else:
            val = _cfg_get(data, "val")
            if val is None or isinstance(val, dict):
                raise NotImplementedError("validate=True: put the validation Dataset object under cfg.data.val (or a hook under cfg.eval_hook); "
                                          "building datasets from a config dict is out of scope")
            from .evaluation import DistEvalTopKAccuracyHook
            runner.register_hook(DistEvalTopKAccuracyHook(val, interval=cfg.get("eval_interval", 1), k=(1, 5), dist=distributed))
    if cfg.get("resume_from"):
        runner.resume(cfg.get("resume_from"))
    elif cfg.get("load_from"):
        load_checkpoint(model, cfg.get("load_from"), map_location="cpu")
    train = DevicePrefetcher(loaders[0])
    runner.run(train, cfg.get("total_epochs", 1))
    return runner


# ------------------------------------------------------------------------------------------------ testing
def single_gpu_test(model, loader):
    """reference test.py:12-39: eval mode, no grad, one result row per video."""
    model.eval()
    results = []
    with torch.no_grad():
        for data in loader:
            img = data["img_group"]
            if torch.is_tensor(img) and not img.is_cuda and torch.cuda.is_available():      # a DataLoader's host batch: the reference's MMDataParallel scatters it to the device (test.py:24-27)
                img = img.cuda(non_blocking=True)
            results.append(model(return_loss=False, img_group=img))
    return results


def collect_results(part, size=None):
    """reference collect_results_gpu (test.py:147-185) for per-video score rows: every rank contributes its `rank::world` share
    (DistributedSampler order); the rows travel as one padded float tensor (instead of pickled bytes) and are re-interleaved on
    rank 0; padding beyond `size` is dropped.  Ranks with an empty shard (fewer videos than ranks) take part with zero rows."""
    rank, world = get_dist_info()
    if world == 1:
        return part if size is None else part[:size]
    dev = torch.device("cuda") if torch.cuda.is_available() and dist.get_backend() == "nccl" else torch.device("cpu")
    rows = [np.asarray(p_, dtype=np.float32).reshape(-1, np.asarray(p_).shape[-1]) for p_ in part]
    mine = torch.from_numpy(np.concatenate(rows, 0)).to(dev) if rows else torch.zeros(0, 0, device=dev)
    # (rows, classes) of every rank first: a rank with an empty shard does not know the row width
    shape = torch.tensor([mine.shape[0], mine.shape[1]], dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    counts = [int(s_[0]) for s_ in shapes]
    mx, classes = max(counts), max(int(s_[1]) for s_ in shapes)
    pad = torch.zeros(mx, classes, device=dev)
    if mine.shape[0]:
        pad[: mine.shape[0]] = mine
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    if rank != 0:
        return None
    ordered = []
    for i in range(mx):                      # DistributedSampler deals the videos rank::world: re-interleave
        for r in range(world):
            if i < counts[r]:
                ordered.append(parts[r][i:i + 1].cpu().numpy())
    return ordered if size is None else ordered[:size]


def multi_gpu_test(model, loader, size=None):
    """reference test.py:42-89: every rank scores its share of the videos, rank 0 gets the ordered list (None elsewhere)."""
    return collect_results(single_gpu_test(model, loader), size)
