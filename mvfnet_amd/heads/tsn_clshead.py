"""TSN classification head for 2D backbones, MI355X-native.

Mirror of the reference's `codes/models/heads/tsn_clshead.py` (TSNClsHead :6-122) + `heads/base.py` (BaseHead
:8-45) for the avg-consensus configuration MVFNet uses: same constructor arguments, the FC is `new_fc`
(state_dict keys cls_head.new_fc.{weight,bias}), `fcn_testing` flag, `init_weights`, `loss`.

forward(x, num_seg): spatial average pool -> (dropout) -> new_fc -> reshape(-1, num_seg, classes) -> mean over
segments (tsn_clshead.py:71-98); with fcn_testing the 1x1x1 Conv3d + mean[T,H,W] branch (:99-117).  Both are
"mean over the clip's T*H*W positions, then FC" up to fp rounding (pooling, consensus and FC are linear;
SURVEY.md 2.2 measured 1e-5), which is how the HIP head kernel computes them in eval mode.
"""
import torch
import torch.nn as nn

from ..builder import HEADS


@HEADS.register_module
class TSNClsHead(nn.Module):
    def __init__(self, spatial_type="avg", spatial_size=7, consensus_cfg=dict(type="avg", dim=1), with_avg_pool=False,
                 temporal_feature_size=1, spatial_feature_size=1, dropout_ratio=0.8, in_channels=1024, num_classes=101,
                 init_std=0.001, fcn_testing=False, extract_feat=False):
        super().__init__()
        if spatial_type != "avg" or consensus_cfg.get("type") != "avg" or consensus_cfg.get("dim", 1) != 1:
            raise NotImplementedError("TSNClsHead: only spatial_type='avg' with the 'avg' consensus over dim 1 is built "
                                      "(the MVFNet configuration)")
        if with_avg_pool or extract_feat or temporal_feature_size != 1 or spatial_feature_size != 1:
            raise NotImplementedError("TSNClsHead: with_avg_pool / extract_feat / feature sizes != 1 are not built")
        self.spatial_type, self.spatial_size = spatial_type, spatial_size
        self.consensus_type = "avg"
        self.dropout_ratio, self.in_channels, self.num_classes, self.init_std = dropout_ratio, in_channels, num_classes, init_std
        self.dropout = nn.Dropout(p=dropout_ratio) if dropout_ratio != 0 else None
        self.new_fc = nn.Linear(in_channels, num_classes)
        self.fcn_testing = fcn_testing
        self.extract_feat = extract_feat
        self._engine = None

    def init_weights(self):
        nn.init.normal_(self.new_fc.weight, 0, self.init_std)
        nn.init.constant_(self.new_fc.bias, 0)
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def engine(self):
        if self._engine is None:
            from ..engine import HeadEngine
            self._engine = HeadEngine(self.new_fc, self.new_fc.weight.device)
        return self._engine

    def invalidate_engine(self):
        self._engine = None

    def forward(self, x, num_seg):
        """x: (N*T, C, h, w) features [or (clips, C, T, h, w) with fcn_testing] -> (clips, num_classes) scores."""
        if not x.is_cuda:
            raise RuntimeError("TSNClsHead: mvfnet_amd runs on MI355X tensors only; no CPU fallback (tests use oracle/)")
        if self.training and self.dropout is not None and torch.is_grad_enabled():
            raise RuntimeError("TSNClsHead: in training the head runs inside Recognizer2D.forward_train (HIP train engine)")
        if x.dim() == 5:                                    # fcn_testing view (clips, C, T, h, w) of the NHWC buffer
            clips, c, t, h, w = x.shape
            feat = x.permute(0, 2, 3, 4, 1).reshape(clips * t, h, w, c)
        else:
            feat = x.permute(0, 2, 3, 1)                     # logical NCHW -> physical NHWC (no copy if channels-last)
        if not feat.is_contiguous():
            feat = feat.contiguous()
        return self.engine().scores(feat, num_seg)

    def loss(self, cls_score, labels):
        """reference heads/base.py:40-45: {'loss_cls': F.cross_entropy(cls_score, labels)} (mean over the clips), computed by the
        HIP cross-entropy kernel of the train head (mvf_ce_loss); no autograd through it -- Recognizer2D.forward_train is the
        path that trains (scores, loss and their gradients in one fused head)."""
        import ctypes as C
        from .._lib import check, lib
        if not cls_score.is_cuda:
            raise RuntimeError("TSNClsHead.loss: mvfnet_amd runs on MI355X tensors only; no CPU fallback (tests use oracle/)")
        s = cls_score.detach().to(torch.float32).contiguous()
        lab = labels.reshape(-1).to(device=s.device, dtype=torch.int64).contiguous()
        if lab.numel() != s.shape[0]:
            raise ValueError("TSNClsHead.loss: %d labels for %d score rows" % (lab.numel(), s.shape[0]))
        part = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
        out = torch.empty(1, dtype=torch.float32, device=s.device)
        P = lambda t: C.c_void_p(t.data_ptr())
        check(lib.mvf_ce_loss(P(s), P(lab), s.shape[0], s.shape[1], None, P(part), P(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "mvf_ce_loss")
        return dict(loss_cls=out.reshape(()))
