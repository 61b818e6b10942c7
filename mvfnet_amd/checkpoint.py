"""Checkpoint wire format of the reference (codes/utils/checkpoint.py): `{'meta': ..., 'state_dict': ..., 'optimizer': ...}`
written with torch.save; loading accepts a bare state_dict or that dict, strips a leading `module.` (DataParallel /
DDP wrappers, :93-95), is non-strict by default and reports missing / unexpected / size-mismatched keys (:178-217;
`num_batches_tracked` is ignored as missing).  Own implementation; no mmcv, no URL / model-zoo schemes (no network)."""
import os
import time
from collections import OrderedDict

import torch


def _unwrap(module):
    return module.module if hasattr(module, "module") and isinstance(getattr(module, "module"), torch.nn.Module) else module


def load_state_dict(module, state_dict, strict=False, logger=None):
    module = _unwrap(module)
    own = module.state_dict()
    unexpected, mismatched, loaded = [], [], set()
    with torch.no_grad():
        for name, value in state_dict.items():
            if name not in own:
                unexpected.append(name)
                continue
            value = value.data if isinstance(value, torch.nn.Parameter) else value
            if tuple(value.shape) != tuple(own[name].shape):
                mismatched.append("%s: checkpoint %s vs model %s" % (name, tuple(value.shape), tuple(own[name].shape)))
                continue
            own[name].copy_(value)
            loaded.add(name)
    missing = [k for k in own if k not in loaded and not k.endswith("num_batches_tracked")]
    msgs = []
    if unexpected:
        msgs.append("unexpected key in source state_dict: %s" % ", ".join(unexpected))
    if missing:
        msgs.append("missing keys in source state_dict: %s" % ", ".join(missing))
    if mismatched:
        msgs.append("size mismatch: %s" % "; ".join(mismatched))
    if msgs:
        text = "The model and loaded state dict do not match exactly\n" + "\n".join(msgs)
        if strict:
            raise RuntimeError(text)
        from .dist import get_dist_info
        if get_dist_info()[0] == 0:                       # the reference reports on rank 0 only (checkpoint.py:60-68)
            (logger.warning if logger is not None else print)(text)
    if hasattr(module, "invalidate_engine"):
        module.invalidate_engine()
    for m in module.modules():
        if hasattr(m, "invalidate_engine"):
            m.invalidate_engine()
    return dict(missing=missing, unexpected=unexpected, mismatched=mismatched)


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    if not isinstance(filename, str) or filename.startswith(("http://", "https://", "modelzoo://", "torchvision://", "open-mmlab://")):
        raise IOError("%r: only local checkpoint files are supported (no network in this build)" % (filename,))
    if not os.path.isfile(filename):
        raise IOError("%s is not a checkpoint file" % filename)
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(ckpt, OrderedDict) or (isinstance(ckpt, dict) and "state_dict" not in ckpt):
        sd = ckpt
    elif isinstance(ckpt, dict):
        sd = ckpt["state_dict"]
    else:
        raise RuntimeError("No state_dict found in checkpoint file %s" % filename)
    if sd and all(k.startswith("module.") for k in sd):
        sd = OrderedDict((k[7:], v) for k, v in sd.items())
    load_state_dict(model, sd, strict, logger)
    return ckpt


def weights_to_cpu(state_dict):
    return OrderedDict((k, v.detach().cpu().clone()) for k, v in state_dict.items())


def save_checkpoint(model, filename, optimizer=None, meta=None):
    meta = dict(meta or {})
    meta.update(time=time.asctime(), framework="mvfnet_amd")
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    ckpt = dict(meta=meta, state_dict=weights_to_cpu(_unwrap(model).state_dict()))
    if optimizer is not None:
        ckpt["optimizer"] = optimizer.state_dict() if hasattr(optimizer, "state_dict") else optimizer
    torch.save(ckpt, filename)
    return filename


# ---- optimizer state: torch.optim.SGD's state_dict layout (what the reference's checkpoints hold, checkpoint.py:262-263) ---------
def sgd_state_dict(momentum_bufs, lr, momentum, weight_decay, nesterov=True, dampening=0.0, multipliers=None, initial_lr=None):
    """{'state': {i: {'momentum_buffer': t}}, 'param_groups': [...]} over n parameters in model.parameters() order;
    momentum_bufs[i] is None for a parameter that has not been stepped (torch creates the buffer at the first step).  One param
    group -- or, with `multipliers` = [(lr_mult, decay_mult)] per parameter, ONE GROUP PER PARAMETER as the reference's paramwise
    build_optimizer makes them (codes/core/train.py:131-153), so torch's load_state_dict finds the group structure it expects.
    Keys of a group = torch.optim.SGD's (older torch ignores the newer ones on load).  `lr` is the CURRENT (scheduled) rate;
    `initial_lr` (the schedule's base rate, x lr_mult per group) is stored beside it as mmcv's LrUpdaterHook does
    (`group.setdefault('initial_lr', group['lr'])` at before_run): without it a checkpoint written after warm-up or an lr step
    would resume under the reference's runner with the decayed rate as its base."""
    n = len(momentum_bufs)
    state = {i: {"momentum_buffer": b} for i, b in enumerate(momentum_bufs) if b is not None}
    base = lr if initial_lr is None else initial_lr

    def group(ids, lr_, wd_, base_):
        return dict(lr=lr_, initial_lr=base_, momentum=momentum, dampening=dampening, weight_decay=wd_, nesterov=bool(nesterov), maximize=False,
                    foreach=None, differentiable=False, fused=None, params=ids)

    if multipliers is None:
        return {"state": state, "param_groups": [group(list(range(n)), lr, weight_decay, base)]}
    return {"state": state, "param_groups": [group([i], lr * a, weight_decay * b, base * a) for i, (a, b) in enumerate(multipliers)]}


def sgd_momentum_buffers(opt_state, n_params):
    """Inverse: per-parameter momentum buffers (None where absent) in parameter order + the (first) param group's hyper-parameters.
    Handles any number of param groups (the reference's paramwise build_optimizer makes one group PER parameter,
    codes/core/train.py:131-153): the packed state ids are the concatenation of the groups' `params` lists."""
    groups = opt_state.get("param_groups")
    if not isinstance(groups, (list, tuple)) or "state" not in opt_state:
        raise ValueError("not a torch optimizer state_dict (keys: %s)" % sorted(opt_state))
    ids = [i for g in groups for i in g["params"]]
    if len(ids) != n_params:
        raise ValueError("optimizer state covers %d parameters, the model has %d" % (len(ids), n_params))
    state = opt_state["state"]
    bufs = []
    for i in ids:
        st = state.get(i, state.get(str(i)))
        bufs.append(None if not st else st.get("momentum_buffer"))
    return bufs, dict(groups[0])
