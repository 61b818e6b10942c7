"""Evaluation metrics and the epoch-end evaluation hook of the runner shell.

Mirrors the reference's `codes/core/evaluation/accuracy.py` (softmax :4-7, confusion_matrix :10-47, mean_class_accuracy :50-70,
top_k_accuracy :83-100, get_weighted_score :103-124) and `eval_hooks.py` (DistEvalTopKAccuracyHook :86-104) in names, argument
meaning and error behaviour; the arithmetic is vectorised numpy over the (videos x classes) score matrix instead of a Python
loop per video.  Ties follow the reference: its top-k is `np.argsort(score)[-k:]` with numpy's default (unstable) sort kind, so which of several
exactly equal scores makes the cut is numpy's choice; the vectorised form issues the same argsort over the rows of the matrix
(the golden case with quantised scores pins that both give the same answer here) rather than counting strictly greater scores."""
import numpy as np


def softmax(x, dim=1):
    x = np.asarray(x)
    e = np.exp(x - np.max(x, axis=dim, keepdims=True))
    return e / e.sum(axis=dim, keepdims=True)


def _labels_i64(y, name):
    if isinstance(y, list):
        y = np.array(y)
    if not isinstance(y, np.ndarray):
        raise TypeError("%s must be list or np.ndarray, but got %s" % (name, type(y)))
    if not y.dtype == np.int64:
        raise TypeError("%s dtype must be np.int64, but got %s" % (name, y.dtype))
    return y


def confusion_matrix(y_pred, y_real):
    """Rows = real label, columns = predicted label, over the sorted set of labels that occur (accuracy.py:10-47)."""
    y_pred = _labels_i64(y_pred, "y_pred")
    y_real = _labels_i64(y_real, "y_real")
    label_set, inv = np.unique(np.concatenate((y_pred, y_real)), return_inverse=True)
    n = len(label_set)
    ip, ir = inv[: len(y_pred)], inv[len(y_pred):]
    m = min(len(ip), len(ir))                       # the reference zips the two lists
    mat = np.zeros((n, n), dtype=np.int64)
    np.add.at(mat, (ir[:m], ip[:m]), 1)
    return mat


def mean_class_accuracy(scores, labels):
    pred = np.argmax(scores, axis=1)
    cf = confusion_matrix(pred, labels).astype(float)
    cnt, hit = cf.sum(axis=1), np.diag(cf)
    return np.mean(np.where(cnt > 0, hit / np.where(cnt > 0, cnt, 1.0), 0.0))


def top_k_accuracy(scores, labels, k=(1,)):
    """Fraction of videos whose label set meets the top-k classes, one value per k (accuracy.py:83-100).  `labels[i]` is an
    int or an iterable of ints (multi-label videos count as hit if ANY of their labels is in the top k)."""
    scores = [np.asarray(s) for s in scores]
    n = len(scores)
    res = []
    single = all(isinstance(y, (int, np.integer)) for y in labels) and n > 0 and all(s.ndim == 1 and s.shape == scores[0].shape for s in scores)
    if single and len(labels) >= n:
        sc = np.stack(scores)                                           # (n, classes)
        order = np.argsort(sc, axis=1)                                  # the reference's call per row, same (default) sort kind
        lab = np.asarray(labels[:n], dtype=np.int64)
        for kk in k:
            top = order[:, -kk:]
            res.append(np.mean((top == lab[:, None]).any(axis=1)))
        return res
    for kk in k:                                                        # ragged / multi-label input: per-video sets
        hits = []
        for x, y in zip(scores, labels):
            ys = {y} if isinstance(y, (int, np.integer)) else set(y)
            hits.append(len(ys.intersection(np.argsort(x)[-kk:])) > 0)
        res.append(np.mean(hits))
    return res


def get_weighted_score(score_list, coeff_list):
    assert len(score_list) == len(coeff_list)
    num = len(score_list[0])
    for s in score_list[1:]:
        assert len(s) == num
    scores = np.array(score_list)                   # (predictors, samples, classes)
    return list(np.tensordot(np.array(coeff_list), scores, axes=(0, 0)))


class EvalTopKAccuracyHook(object):
    """After every `interval`-th training epoch: score the validation set with the model in eval mode (every rank its
    `rank::world` share, rows gathered on rank 0 -- `runner.multi_gpu_test`) and log top-k accuracy
    (eval_hooks.py:17-104; the reference exchanges pickled temp files through `work_dir`, here the rows travel as one
    all_gather of float tensors)."""

    def __init__(self, loader, labels, interval=1, k=(1, 5)):
        self.loader, self.labels, self.interval, self.k = loader, list(labels), interval, tuple(k)
        self.history = []

    def after_train_epoch(self, runner):
        if (runner.epoch % self.interval) != 0:
            return None
        from .runner import multi_gpu_test
        was_training = runner.model.training
        results = multi_gpu_test(runner.model, self.loader, size=len(self.labels))
        runner.model.train(was_training)
        if results is None:                          # not rank 0
            return None
        acc = top_k_accuracy([np.asarray(r).squeeze() for r in results], self.labels, k=self.k)
        out = {"top%d acc" % kk: float(a) for kk, a in zip(self.k, acc)}
        out["epoch"] = runner.epoch
        self.history.append(out)
        return out


class DistEvalTopKAccuracyHook(EvalTopKAccuracyHook):
    """The reference's hook signature (eval_hooks.py:17-104): `DistEvalTopKAccuracyHook(dataset, interval=1, k=(1,), dist=True)`.
    `dataset`: anything with `__len__`, `__getitem__(i) -> dict(img_group=tensor, ...)` and `video_infos[i]['label']` (the reference's
    RawFramesDataset / VideoDataset in test_mode; building one from a config dict needs the dataset classes, which are out of scope
    here -- pass the object).  Every rank scores items `rank::world` one by one, as DistEvalHook.after_train_epoch does
    (collate([data], samples_per_gpu=1) = a batch dimension of one), the rows are gathered on rank 0 and top-k accuracy is logged."""

    def __init__(self, dataset, interval=1, k=(1,), dist=True):
        if isinstance(dataset, dict):
            raise TypeError("DistEvalTopKAccuracyHook: a dataset CONFIG dict needs the reference's dataset classes (out of scope here); "
                            "pass the Dataset object")
        if not (hasattr(dataset, "__getitem__") and hasattr(dataset, "__len__")):
            raise TypeError("dataset must be a Dataset object or a dict, not {}".format(type(dataset)))      # the reference's message
        self.dataset, self.dist = dataset, dist
        labels = [dataset.video_infos[i]["label"] for i in range(len(dataset))]
        super().__init__(_RankShare(dataset, dist), labels, interval, k)


class _RankShare(object):
    """items rank::world of a dataset as batches of one (what the reference's hook feeds the model)."""

    def __init__(self, dataset, dist=True):
        self.dataset, self.dist = dataset, dist

    def __iter__(self):
        import torch
        from .dist import get_dist_info
        rank, world = get_dist_info() if self.dist else (0, 1)
        for idx in range(rank, len(self.dataset), world):
            data = self.dataset[idx]
            img = data["img_group"]
            img = img if isinstance(img, torch.Tensor) else torch.as_tensor(np.asarray(img))
            out = {k: v for k, v in data.items() if k not in ("img_group", "label")}
            out["img_group"] = img.unsqueeze(0).cuda(non_blocking=True)
            yield out
