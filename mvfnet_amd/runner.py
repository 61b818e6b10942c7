"""Minimal mmcv-free runner shell around the HIP engines: the host-side control flow of the reference's
`codes/core/train.py` (batch_processor :52-60, parse_losses :32-49, _dist_train :159-212) and `codes/core/test.py`
(single_gpu_test :12-39, multi_gpu_test :42-89, collect_results_gpu :147-185), with the schedule the shipped config
uses (lr_config: step [90,130] x0.1, linear warm-up 25070 iters from ratio 0.01; mmcv 0.4.3 semantics).
SURVEY section 8(f) rows 1-2 ("next"): enough to drive train/test loops end to end on synthetic or user loaders."""
import os

import torch
import torch.distributed as dist

from .checkpoint import load_checkpoint, save_checkpoint
from .dist import get_dist_info


def step_lr(base_lr, epoch, it, steps=(90, 130), gamma=0.1, warmup="linear", warmup_iters=25070, warmup_ratio=0.01):
    """lr at (epoch, global iteration) -- mmcv LrUpdaterHook 'step' policy + linear warm-up."""
    lr = base_lr * (gamma ** sum(1 for s in steps if epoch >= s))
    if warmup == "linear" and it < warmup_iters:
        k = (1 - it / float(warmup_iters)) * (1 - warmup_ratio)
        lr = lr * (1 - k)
    return lr


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


class Runner(object):
    """Epoch/iteration loop with the reference's hook order: lr update -> forward -> backward -> all-reduce/world ->
    clip -> step -> checkpoint every `ckpt_interval` epochs.  Uses the fused HIP TrainEngine (one flat all-reduce +
    one optimizer kernel)."""

    def __init__(self, model, work_dir=None, lr=0.015, momentum=0.9, weight_decay=1e-4, max_norm=40.0, lr_steps=(90, 130),
                 warmup_iters=25070, warmup_ratio=0.01, ckpt_interval=10, log_interval=20, logger=print):
        self.model, self.work_dir = model, work_dir
        self.engine = model.train_engine(lr=lr, momentum=momentum, weight_decay=weight_decay, max_norm=max_norm)
        self.base_lr, self.lr_steps = lr, tuple(lr_steps)
        self.warmup_iters, self.warmup_ratio = warmup_iters, warmup_ratio
        self.ckpt_interval, self.log_interval, self.log = ckpt_interval, log_interval, logger
        self.epoch, self.iter = 0, 0
        self.hooks = []               # objects with after_train_epoch(runner), e.g. evaluation.EvalTopKAccuracyHook

    def register_hook(self, hook):
        self.hooks.append(hook)
        return hook

    def current_lr(self):
        return step_lr(self.base_lr, self.epoch, self.iter, self.lr_steps, 0.1, "linear", self.warmup_iters, self.warmup_ratio)

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
        while self.epoch < max_epochs:
            if hasattr(getattr(loader, "sampler", None), "set_epoch"):
                loader.sampler.set_epoch(self.epoch)
            self.train_epoch(loader)

    def save_checkpoint(self):
        path = os.path.join(self.work_dir, "epoch_%d.pth" % self.epoch)
        opt = dict(momentum_buffer=self.engine.flat_mom.detach().cpu(), steps=self.engine.steps)
        save_checkpoint(self.model, path, optimizer=opt, meta=dict(epoch=self.epoch, iter=self.iter))
        latest = os.path.join(self.work_dir, "latest.pth")
        if os.path.lexists(latest):
            os.remove(latest)
        os.symlink(os.path.basename(path), latest)
        return path

    def resume(self, filename):
        ckpt = load_checkpoint(self.model, filename, strict=True)
        self.epoch, self.iter = ckpt["meta"]["epoch"], ckpt["meta"]["iter"]
        opt = ckpt.get("optimizer")
        if opt:
            self.engine.flat_mom.copy_(opt["momentum_buffer"].to(self.engine.flat_mom.device))
            self.engine.steps = opt["steps"]
        return ckpt


def single_gpu_test(model, loader):
    """reference test.py:12-39: eval mode, no grad, one result row per video."""
    model.eval()
    results = []
    with torch.no_grad():
        for data in loader:
            results.append(model(return_loss=False, img_group=data["img_group"]))
    return results


def multi_gpu_test(model, loader, size=None):
    """reference test.py:42-89 + collect_results_gpu :147-185: every rank scores its `rank::world` share of the videos
    (DistributedSampler order), the (1, classes) rows are gathered as float tensors (instead of pickled bytes) and
    re-interleaved on rank 0; padding beyond `size` is dropped."""
    part = single_gpu_test(model, loader)
    rank, world = get_dist_info()
    if world == 1:
        return part if size is None else part[:size]
    dev = torch.device("cuda") if torch.cuda.is_available() and dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor(__import__("numpy").concatenate(part, 0) if part else [], dtype=torch.float32, device=dev)
    count = torch.tensor([mine.shape[0]], device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    mx = int(max(int(c) for c in counts))
    pad = torch.zeros(mx, mine.shape[1] if mine.dim() == 2 else 0, device=dev)
    pad[: mine.shape[0]] = mine
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    if rank != 0:
        return None
    ordered = []
    for i in range(mx):
        for r in range(world):
            if i < int(counts[r]):
                ordered.append(parts[r][i:i + 1].cpu().numpy())
    return ordered if size is None else ordered[:size]
