"""[r6] Launch plans (mvfnet_amd/launch_plan.py + mvf_plan_run, csrc/launch_plan.hip): a train step replayed from a recorded launch table must be the eager
step -- same kernels, same arguments, same stream order -- so losses and parameters agree BIT FOR BIT over a run that starts eager, records twice and then replays.
Reference: one iteration of batch_processor + DistOptimizerHook.after_train_iter (codes/core/train.py:45-60, codes/core/dist_utils.py:61-67)."""
import ctypes as C
import os

import pytest
import torch

from mvfnet_amd import synth

pytestmark = pytest.mark.gpu


def _engine(dtype, use_plan, t=4, dropout=0.5, depth=50):
    import mvfnet_amd
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(depth, t, dropout_ratio=dropout), None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r%d/" % depth + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r%d/" % depth + k]) for k in sd})
    eng = m.cuda().train().train_engine(dtype=dtype)
    eng.use_plan = use_plan
    return m, eng


def _run(dtype, use_plan, steps, batches, lrs, seed=7, **kw):
    torch.manual_seed(seed)                                  # the dropout masks
    m, eng = _engine(dtype, use_plan, **kw)
    losses = []
    for i in range(steps):
        imgs, labels = batches[i % len(batches)]
        losses.append(eng.train_step(imgs.clone(), labels.clone(), lr=lrs[i % len(lrs)]).clone())      # fresh tensors: the plan patches their addresses
    torch.cuda.synchronize()
    return eng, torch.cat(losses), eng.flat_params.clone(), [b.clone() for b in m.buffers()]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_replayed_steps_equal_eager_steps_bit_for_bit(dtype):
    gen = torch.Generator(device="cuda").manual_seed(3)
    batches = [(torch.randn(2, 4, 3, 64, 64, device="cuda", generator=gen), torch.randint(0, 400, (2, 1), device="cuda", generator=gen)) for _ in range(3)]
    lrs = [0.015, 0.01, 0.02, 0.005]
    steps = 9
    eng_p, loss_p, par_p, buf_p = _run(dtype, True, steps, batches, lrs)
    eng_e, loss_e, par_e, buf_e = _run(dtype, False, steps, batches, lrs)
    st = list(eng_p._plans.values())
    assert len(st) == 1 and st[0]["plan"] is not None and st[0]["eager"] == eng_p.plan_warmup and st[0]["tries"] == 2, "no plan was accepted: %r" % [(s_["eager"], s_["tries"]) for s_ in st]
    plan = st[0]["plan"]
    assert plan.n_ops > 300 and all(len(v) >= 1 for k, v in plan.slots.items() if k != "mask" or eng_p.dropout > 0), (plan.n_ops, {k: len(v) for k, v in plan.slots.items()})
    assert not getattr(eng_e, "_plans", None)
    assert torch.equal(loss_p, loss_e), (loss_p, loss_e)
    assert torch.equal(par_p, par_e)
    for a, b in zip(buf_p, buf_e):
        assert torch.equal(a, b)
    print("launch plan %s: %d ops in %d segment(s), slots %s; %d replayed steps bit-identical to eager" %
          (dtype, plan.n_ops, len(plan.cuts), {k: len(v) for k, v in plan.slots.items()}, steps - eng_p.plan_warmup - 2))


def test_a_new_batch_shape_gets_its_own_plan_and_frozen_statistics_stay_eager():
    gen = torch.Generator(device="cuda").manual_seed(5)
    m, eng = _engine(torch.bfloat16, True, dropout=0.0)
    a = (torch.randn(2, 4, 3, 64, 64, device="cuda", generator=gen), torch.randint(0, 400, (2, 1), device="cuda", generator=gen))
    b = (torch.randn(1, 4, 3, 64, 64, device="cuda", generator=gen), torch.randint(0, 400, (1, 1), device="cuda", generator=gen))      # the partial last batch of an epoch
    for _ in range(5):
        eng.train_step(*a)
    for _ in range(5):
        eng.train_step(*b)
    eng.train_step(*a)
    torch.cuda.synchronize()
    assert len(eng._plans) == 2 and all(s["plan"] is not None for s in eng._plans.values())
    m.backbone.bn1.eval()                                    # a BatchNorm on its running statistics: folded by torch calls every step -> no plan
    n = len(eng._plans)
    eng.train_step(*a)
    assert len(eng._plans) == n and torch.isfinite(eng.flat_params).all()


def test_a_larger_batch_replaces_storage_and_the_smaller_batchs_plan_is_recorded_again():
    """A plan holds raw pointers into the engine's persistent buffers.  When a LARGER batch arrives, TrainEngine.buf() / workspace() replace those allocations
    (one allocation per call site, sized for the largest shape seen): the plan recorded for the smaller batch must not be replayed on the freed storage.  The
    sequence small x 6 (plan accepted) -> large x 6 -> small x 6 must equal the same sequence without plans bit for bit, and the small shape must have been
    recorded a second time (reference: drop_last=False loaders and a val pass between epochs hand the runner batches of different sizes, codes/core/train.py:45-60)."""
    gen = torch.Generator(device="cuda").manual_seed(11)
    small = (torch.randn(1, 4, 3, 64, 64, device="cuda", generator=gen), torch.randint(0, 400, (1, 1), device="cuda", generator=gen))
    large = (torch.randn(3, 4, 3, 64, 64, device="cuda", generator=gen), torch.randint(0, 400, (3, 1), device="cuda", generator=gen))
    seq = [small] * 6 + [large] * 6 + [small] * 6
    out = {}
    for use_plan in (True, False):
        torch.manual_seed(9)
        m, eng = _engine(torch.bfloat16, use_plan)
        losses, epochs, replays = [], [], 0
        for imgs, labels in seq:
            mine = lambda: [v["plan"] for k, v in getattr(eng, "_plans", {}).items() if k[0] == tuple(imgs.shape)]  # noqa: E731
            before = mine()
            losses.append(eng.train_step(imgs.clone(), labels.clone()).clone())
            epochs.append(eng._storage_epoch)
            replays += bool(before) and before[0] is not None and mine()[0] is before[0]          # this shape's plan existed and survived the step: it was replayed
        torch.cuda.synchronize()
        out[use_plan] = (torch.cat(losses), eng.flat_params.clone(), epochs, eng, replays)
    loss_p, par_p, epochs, eng_p, replays = out[True]
    assert epochs[5] == epochs[0] and epochs[6] > epochs[5] and epochs[-1] == epochs[11], epochs      # only the first large step replaced storage
    st = list(eng_p._plans.values())
    assert len(st) == 2 and all(s_["plan"] is not None and s_["plan"].storage_epoch == eng_p._storage_epoch for s_ in st), [(s_["eager"], s_["tries"]) for s_ in st]
    assert replays == 8, replays                             # steps 5-6, 11-12 and 15-18 ran from plans; step 13 dropped the stale one
    assert torch.equal(loss_p, out[False][0]), (loss_p, out[False][0])
    assert torch.equal(par_p, out[False][1])


def test_plan_with_the_gradient_exchange_cut_in_two_segments(tmp_path):
    """With a process group the tail bucket's all-reduce is issued in the middle of backward: the plan has two segments around that torch call."""
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    dist.init_process_group("gloo", init_method="file://%s" % (tmp_path / "pg"), rank=0, world_size=1)
    try:
        gen = torch.Generator(device="cuda").manual_seed(9)
        batch = (torch.randn(2, 4, 3, 64, 64, device="cuda", generator=gen), torch.randint(0, 400, (2, 1), device="cuda", generator=gen))
        out = {}
        for use_plan in (True, False):
            torch.manual_seed(1)
            m, eng = _engine(torch.bfloat16, use_plan)
            eng.force_allreduce = True
            losses = [eng.train_step(batch[0].clone(), batch[1].clone()).clone() for _ in range(9)]
            torch.cuda.synchronize()
            out[use_plan] = (torch.cat(losses), eng.flat_params.clone(), eng)
        plans = [s["plan"] for s in out[True][2]._plans.values()]
        assert len(plans) == 1 and plans[0] is not None and len(plans[0].cuts) == 2 and len(plans[0].py_ops) == 1
        assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    finally:
        dist.destroy_process_group()


def test_plan_run_reports_the_failing_op():
    from mvfnet_amd import _lib as L
    lib = L.lib
    ops = (L.PlanOp * 2)()
    d = L.MvfDesc(10, 16, 4, 4, 4, 4, 7, 0, 0)                # nt = 10 is not a multiple of n_segment = 4: mvf_fwd_infer refuses it
    words = (C.c_ulonglong * 16)(C.addressof(d), 0, 0, 0, 0, 0, 0, 0, 0)
    floats = (C.c_float * 1)()
    ops[0].kind, ops[0].n_int, ops[0].fn, ops[0].word0 = 0, 9, C.cast(lib.mvf_fwd_infer, C.c_void_p).value, 0
    ops[1].kind = 7
    failed = C.c_int(-5)
    assert lib.mvf_plan_run(ops, 2, words, floats, C.byref(failed)) == -2 and failed.value == 0 and b"n_segment" in lib.mvf_last_error()
    assert lib.mvf_plan_run(C.cast(C.byref(ops, C.sizeof(L.PlanOp)), C.POINTER(L.PlanOp)), 1, words, floats, C.byref(failed)) == -1 and failed.value == 0
    assert lib.mvf_plan_run(ops, 0, words, floats, C.byref(failed)) == 0 and failed.value == -1
