#!/usr/bin/env python3
"""Golden vectors for mvfnet_amd/evaluation.py from the REFERENCE's own accuracy.py (pure numpy, loaded by path; nothing of it
is written anywhere -- only seeded inputs and the numbers it returns).  Run in the build container: python tests/golden/make_eval_golden.py"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_accuracy", "/root/reference/codes/core/evaluation/accuracy.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
rng = np.random.RandomState(20260928)
for ci, (n, c) in enumerate([(64, 400), (37, 11), (5, 3), (200, 51)]):
    scores = rng.randn(n, c).astype(np.float32)
    if ci == 1:                                   # exact ties, as a quantised model produces them
        scores = np.round(scores * 2) / 2
    labels = rng.randint(0, c, size=n).astype(np.int64)
    out["c%d_scores" % ci] = scores
    out["c%d_labels" % ci] = labels
    ks = (1, 5) if c >= 5 else (1, 2)
    out["c%d_k" % ci] = np.array(ks)
    out["c%d_topk" % ci] = np.array(ref.top_k_accuracy(list(scores), [int(v) for v in labels], k=ks), dtype=np.float64)
    out["c%d_mca" % ci] = np.float64(ref.mean_class_accuracy(list(scores), list(labels)))
    out["c%d_cm" % ci] = ref.confusion_matrix(np.argmax(scores, axis=1), labels)
    out["c%d_softmax" % ci] = ref.softmax(scores, dim=1)
# multi-label videos (label sets), weighted fusion of two score lists
scores = rng.randn(20, 9).astype(np.float32)
sets = [sorted(set(rng.randint(0, 9, size=rng.randint(1, 4)).tolist())) for _ in range(20)]
out["ml_scores"] = scores
out["ml_sets"] = np.array([s + [-1] * (3 - len(s)) for s in sets])
out["ml_topk"] = np.array(ref.top_k_accuracy(list(scores), sets, k=(1, 3)), dtype=np.float64)
a, b = rng.randn(6, 7), rng.randn(6, 7)
out["ws_a"], out["ws_b"] = a, b
out["ws_out"] = np.array(ref.get_weighted_score([list(a), list(b)], [0.75, 1.5]))
np.savez_compressed(os.path.join(HERE, "eval_cases.npz"), **out)
print("wrote eval_cases.npz", {k: getattr(v, "shape", None) for k, v in out.items() if k.endswith("topk") or k.endswith("mca")})
