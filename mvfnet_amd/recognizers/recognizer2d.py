"""Recognizer2D -- the mmaction-style model API of the reference, MI355X-native.

Mirror of `codes/models/recognizers/base.py` (BaseRecognizer :11-82) and `recognizer2d.py` (Recognizer2D :8-179):
built by `build_recognizer(cfg.model, train_cfg, test_cfg)`, children `backbone` / `cls_head` (same state_dict
prefixes), MVF inserted from `module_cfg`, `forward(img_group, label, return_loss=True, return_numpy=True)`.
"""
import torch
import torch.nn as nn

from ..builder import RECOGNIZERS, build_backbone, build_head


class _TrainStepFn(torch.autograd.Function):
    """Whole-model autograd node: forward = HIP train forward + loss, backward = HIP backward chain."""

    @staticmethod
    def forward(ctx, eng, imgs, labels, *params):
        ctx.eng = eng
        ctx.n = len(params)
        loss = eng.forward(imgs, labels).reshape(())
        ctx.token = eng.forward_count          # the engine keeps ONE step's activations: backward must belong to the latest forward
        return loss

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.eng
        if ctx.token != eng.forward_count or eng.saved is None:
            raise RuntimeError("Recognizer2D: backward() of a loss whose activations are gone -- the HIP train engine keeps one step "
                               "(call loss.backward() before the next forward_train, and only once)")
        eng.backward()
        eng.flat_grads.mul_(gout)
        grads = []
        for p in eng.model.parameters():
            v = eng.grad_of(p)
            # a parameter whose .grad IS the flat-buffer view (engine.attach_grads()) already holds its gradient: handing the same
            # memory to autograd's AccumulateGrad would add it to itself; everything else gets its own copy
            aliased = p.grad is not None and p.grad.data_ptr() == v.data_ptr()
            grads.append(None if aliased else v.clone())
        return (None, None, None) + tuple(grads)


@RECOGNIZERS.register_module
class Recognizer2D(nn.Module):
    def __init__(self, modality="RGB", backbone=None, cls_head=None, fcn_testing=False, module_cfg=None,
                 nonlocal_cfg=None, train_cfg=None, test_cfg=None):
        super().__init__()
        if modality != "RGB":
            raise NotImplementedError("only modality='RGB' is built (the MVFNet configs)")
        if nonlocal_cfg:
            raise NotImplementedError("non-local blocks are out of scope")
        if backbone is None or backbone.get("type") != "ResNet":
            raise NotImplementedError("Recognizer2D: backbone type 'ResNet' is the one built here")
        self.fp16_enabled = False
        self.backbone = build_backbone(backbone)
        self.cls_head = build_head(cls_head) if cls_head is not None else None
        self.init_weights()
        self.fcn_testing, self.modality = fcn_testing, modality
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.module_cfg = dict(module_cfg) if module_cfg else module_cfg
        self.in_channels = 3
        if self.module_cfg:
            # reference recognizer2d.py:45-59: weights are initialised/loaded BEFORE the MVF wrappers exist
            self.module_name = self.module_cfg.pop("type")
            if self.module_name != "MVF":
                raise NotImplementedError("module_cfg type %r: only 'MVF' is built" % self.module_name)
            from ..modules.MVF import make_multi_view_fusion
            make_multi_view_fusion(self.backbone, **self.module_cfg)
            self.backbone.invalidate_engine()

    @property
    def with_cls_head(self):
        return self.cls_head is not None

    def init_weights(self):
        self.backbone.init_weights()
        if self.with_cls_head:
            self.cls_head.init_weights()

    def extract_feat(self, img_group):
        return self.backbone(img_group)

    def average_clip(self, cls_score):
        """reference base.py:43-74."""
        if self.test_cfg is None:
            self.test_cfg = dict(average_clips=None)
        if "average_clips" not in self.test_cfg:
            raise KeyError('"average_clips" must defined in test_cfg\'s keys')
        return self.cls_head.engine().average(cls_score, self.test_cfg["average_clips"])

    def set_input_pipeline(self, pipeline):
        """Feed the model DECODED uint8 frames -- `img_group` (B, frames, Hs, Ws, 3) uint8 plus an optional `window=` (frames, 3)
        int32 of per-frame (y0, x0, flip) -- instead of the normalised fp32 tensor: `pipeline` is a preprocess.FramePipeline
        (the config's img_norm_cfg + crop size); crop, flip, Normalize and FormatShape then run inside the stem's input kernel."""
        self.input_pipeline = pipeline
        self.backbone.input_pipeline = pipeline
        return self

    def forward(self, img_group, label=None, return_loss=True, return_numpy=True, **kwargs):
        if return_loss:
            return self.forward_train(img_group, label, **kwargs)
        return self.forward_test(img_group, return_numpy, **kwargs)

    # ---- training ---------------------------------------------------------------------------------------------
    def train_engine(self, **opt):
        """The HIP training engine bound to this model (created on first use; parameters become views of one flat
        buffer).  opt: lr, momentum, weight_decay, max_norm (defaults = the reference's optimizer config)."""
        if getattr(self, "_train_engine", None) is None:
            from ..train_engine import TrainEngine
            self._train_engine = TrainEngine(self, **opt)
        elif opt:                                  # an engine exists already: the options must not be dropped silently
            self._train_engine.set_options(**opt)
        return self._train_engine

    def forward_train(self, imgs, labels, **kwargs):
        """imgs [B, T, 3, H, W], labels [B, 1] -> {'loss_cls': scalar tensor} (reference recognizer2d.py:132-149).
        The returned loss supports .backward(): gradients land in the parameters' .grad (copies of the engine's flat gradient
        buffer; after engine.attach_grads() the .grad tensors ARE views of it), as the reference's DistOptimizerHook expects."""
        if not imgs.is_cuda:
            raise RuntimeError("Recognizer2D: mvfnet_amd runs on MI355X tensors only; no CPU fallback (tests use oracle/)")
        # BatchNorms in eval mode (backbone norm_eval=True / frozen stages / partial_norm, reference resnet.py:496-527) normalise with
        # their running statistics and keep them; parameters excluded from training (frozen_stages: a prefix of model.parameters();
        # norm_frozen / partial_norm: BatchNorm weights / biases in scattered places) are skipped by the optimizer kernel
        eng = self.train_engine()
        eng.trainable_offset()
        eng.input_pipeline, eng.input_window = getattr(self, "input_pipeline", None), kwargs.get("window")
        eng.dropout = self.cls_head.dropout_ratio if (self.cls_head.dropout is not None and self.cls_head.training) else 0.0
        params = [p for p in self.parameters()]
        loss = _TrainStepFn.apply(eng, imgs, labels, *params)
        return dict(loss_cls=loss)

    def forward_test(self, imgs, return_numpy=True, **kwargs):
        """imgs [B, clips*crops*T, 3, H, W] -> (1 | clips, num_classes) (reference recognizer2d.py:151-179)."""
        if not imgs.is_cuda:
            raise RuntimeError("Recognizer2D: mvfnet_amd runs on MI355X tensors only; no CPU fallback (tests use oracle/)")
        with torch.no_grad():
            if imgs.dtype == torch.uint8:                                 # decoded frames (B, frames, Hs, Ws, 3), see set_input_pipeline
                x = imgs.reshape((-1,) + tuple(imgs.shape[-3:]))
                self.backbone.engine().input_window = kwargs.get("window")
            else:
                x = imgs.reshape((-1, self.in_channels) + tuple(imgs.shape[3:]))
            feat = self.extract_feat(x)                                   # (B*frames, 2048, h, w), channels-last
            if self.with_cls_head:
                num_seg = self.module_cfg["n_segment"] if self.module_cfg else x.shape[0] // imgs.shape[0]
                # with fcn_testing the reference reshapes to (clips, C, T, h, w) and runs a 1x1x1 conv + mean; the head
                # kernel computes the same mean-then-FC directly from the channels-last features
                cls_score = self.cls_head(feat, num_seg)
                cls_score = self.average_clip(cls_score)
            else:
                cls_score = feat
        return cls_score.cpu().numpy() if return_numpy else cls_score
