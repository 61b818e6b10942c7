"""mvfnet_amd/evaluation.py against golden vectors made by the reference's own accuracy.py (tests/golden/make_eval_golden.py)."""
import os

import numpy as np
import pytest

from mvfnet_amd import evaluation as E

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_cases.npz"))


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_metrics_match_reference_golden(ci):
    scores, labels = G["c%d_scores" % ci], G["c%d_labels" % ci]
    ks = tuple(int(v) for v in G["c%d_k" % ci])
    got = E.top_k_accuracy(list(scores), [int(v) for v in labels], k=ks)
    assert np.array_equal(np.array(got, dtype=np.float64), G["c%d_topk" % ci])            # counts / n: exact
    assert E.mean_class_accuracy(list(scores), list(labels)) == pytest.approx(float(G["c%d_mca" % ci]), abs=1e-15)
    assert np.array_equal(E.confusion_matrix(np.argmax(scores, axis=1), labels), G["c%d_cm" % ci])
    np.testing.assert_allclose(E.softmax(scores, dim=1), G["c%d_softmax" % ci], rtol=1e-6, atol=0)


def test_multilabel_topk_and_weighted_score_match_reference_golden():
    sets = [[int(v) for v in row if v >= 0] for row in G["ml_sets"]]
    assert np.array_equal(np.array(E.top_k_accuracy(list(G["ml_scores"]), sets, k=(1, 3))), G["ml_topk"])
    ws = np.array(E.get_weighted_score([list(G["ws_a"]), list(G["ws_b"])], [0.75, 1.5]))
    np.testing.assert_allclose(ws, G["ws_out"], rtol=1e-12)


def test_confusion_matrix_rejects_what_the_reference_rejects():
    with pytest.raises(TypeError):
        E.confusion_matrix(np.array([1, 2], dtype=np.int32), np.array([1, 2], dtype=np.int64))
    with pytest.raises(TypeError):
        E.confusion_matrix("12", [1, 2])
    with pytest.raises(TypeError):
        E.confusion_matrix([], [])                   # np.array([]) is float64: the reference raises here too
    e = np.array([], dtype=np.int64)
    assert E.confusion_matrix(e, e).shape == (0, 0)


def test_eval_hook_runs_every_interval_and_restores_train_mode():
    class M(object):
        training = True

        def train(self, mode=True):
            self.training = mode

        def eval(self):
            self.training = False

        def __call__(self, return_loss=False, img_group=None):
            assert not self.training
            return np.eye(4, dtype=np.float32)[[int(img_group)]]          # predicts class == the "video" id

    class R(object):
        model, epoch = M(), 0

    loader = [dict(img_group=i) for i in range(4)]
    hook = E.EvalTopKAccuracyHook(loader, labels=[0, 1, 2, 0], interval=2, k=(1, 2))
    r = R()
    r.epoch = 1
    assert hook.after_train_epoch(r) is None
    r.epoch = 2
    out = hook.after_train_epoch(r)
    assert out["top1 acc"] == 0.75 and out["epoch"] == 2 and r.model.training
    assert out["top2 acc"] == 0.75          # video 3: scores (0,0,0,1) -> argsort top-2 = classes {2, 3}, label 0 misses
