"""[r6] Launch plans: one training step recorded as a table of C-ABI calls + cross-stream orderings and replayed from C (mvf_plan_run, csrc/launch_plan.hip).

The train engine issues ~650 library calls per step from Python, always the same ones with the same arguments (persistent buffers, memoised descriptors): ~4 ms
of host time per step, half of a 9 ms step at the reference's own batch size (12 clips per GPU: configs/MVFNet/K400/mvf_kinetics400_2d_rgb_r50_dense.py:121-123;
one step = batch_processor + DistOptimizerHook.after_train_iter, codes/core/train.py:45-60, codes/core/dist_utils.py:61-67).  A plan is built by running ONE
eager step with every library call routed through `RecordingLib` (the call is made as usual AND appended to the plan) and every cross-stream ordering made through
`Recorder.wait` / `mark` / `wait_mark`; it is accepted once a second recording of the next step is identical to it (same calls, same arguments).  A replay patches
the few arguments that legitimately change -- input batch, labels, dropout mask, loss tensor -- and hands the table to C.  Work that must stay in Python (a
collective in the middle of backward) cuts the plan into segments with a callback in between.  No kernel, argument or stream order differs from the eager step:
results are bit-identical (tests/test_launch_plan_gpu.py).
"""
import ctypes as C

import torch

from . import _lib

CALL, RECORD, WAIT = 0, 1, 2
# entry points that launch nothing (sizes, plans): called as usual while recording, never part of a plan
_PURE_SUFFIXES = ("_bytes", "_rows", "_splits", "_plan")
_PURE = {"mvf_last_error", "mvf_abi_version"}


def _is_pure(name):
    return name in _PURE or name.endswith(_PURE_SUFFIXES)


class Recorder(object):
    """Collects the ops of one step.  Segment k = ops[cuts[k - 1]:cuts[k]]; after segment k the callable py_ops[k] runs in Python."""

    def __init__(self):
        self.ops = []            # (kind, fn_addr, name, ints, floats, ptr_flags)
        self.cuts, self.py_ops = [], []
        self.keep = []           # ctypes objects / torch events the recorded words point at
        self.unsupported = None  # reason why this step cannot become a plan

    # ---- library calls -------------------------------------------------------------------------------------------------------------------------------
    def call(self, name, fn, args):
        ints, floats, flags = [], [], []
        argtypes = fn.argtypes
        if argtypes is None or len(argtypes) != len(args):
            self.unsupported = "%s: %d arguments for %s declared" % (name, len(args), None if argtypes is None else len(argtypes))
            return
        for a, t in zip(args, argtypes):
            if t is C.c_float:
                floats.append(float(a.value if isinstance(a, C.c_float) else a))
                continue
            is_ptr = t is C.c_void_p or (isinstance(t, type) and issubclass(t, C._Pointer))
            if a is None:
                v = 0
            elif isinstance(a, int):
                v = a
            elif hasattr(a, "_obj"):                     # byref(x): the address of x, which must outlive the plan
                self.keep.append(a._obj)
                v = C.addressof(a._obj)
            elif isinstance(a, C._SimpleCData):
                v = a.value or 0
            elif isinstance(a, (C.Structure, C.Array)):
                self.keep.append(a)
                v = C.addressof(a)
            else:
                self.unsupported = "%s: argument of type %s" % (name, type(a).__name__)
                return
            ints.append(int(v) & 0xFFFFFFFFFFFFFFFF)
            flags.append(is_ptr)
        self.ops.append((CALL, C.cast(fn, C.c_void_p).value, name, ints, floats, flags))

    # ---- cross-stream orderings: the torch calls are made as usual, AND recorded on plan-owned events --------------------------------------------------
    def _event(self, stream):
        ev = torch.cuda.Event()
        ev.record(stream)                                # materialises the hipEvent (recorded again by every replay)
        self.keep.append(ev)
        return ev

    def mark(self, stream):
        """An event recorded on `stream` now; returns the token wait_mark() takes."""
        ev = self._event(stream)
        self.ops.append((RECORD, None, "record", [ev.cuda_event, stream.cuda_stream], [], [False, False]))
        return ev

    def wait_mark(self, stream, ev):
        stream.wait_event(ev)
        self.ops.append((WAIT, None, "wait", [stream.cuda_stream, ev.cuda_event], [], [False, False]))

    def wait(self, dst, src):
        """dst waits for everything queued on src so far (torch: dst.wait_stream(src))."""
        self.wait_mark(dst, self.mark(src))

    def py_op(self, fn):
        """Work that stays in Python between two segments (fn is called now and after the segment in every replay)."""
        self.cuts.append(len(self.ops))
        self.py_ops.append(fn)
        fn()


class RecordingLib(object):
    """Stands in for _lib.lib while a step is recorded: every call is made AND recorded."""

    def __init__(self, rec):
        self._rec = rec
        self._fns = {}

    def __getattr__(self, name):
        w = self._fns.get(name)
        if w is None:
            fn = getattr(_lib.lib, name)
            if _is_pure(name):
                w = fn
            else:
                rec = self._rec

                def w(*args, _fn=fn, _name=name):
                    rec.call(_name, _fn, args)
                    return _fn(*args)
            self._fns[name] = w
        return w


class Plan(object):
    """The frozen table of one recorded step."""

    def __init__(self, rec, dynamic):
        """dynamic: {name: pointer value at record time} -- every pointer-typed argument word equal to it becomes a slot patched per run."""
        n = len(rec.ops)
        self.n_ops = n
        self.ops = (_lib.PlanOp * max(n, 1))()
        words, floats, flags = [], [], []
        for i, (kind, fn, name, ints, flts, fl) in enumerate(rec.ops):
            o = self.ops[i]
            o.kind, o.n_int, o.n_flt, o.fn, o.word0, o.float0 = kind, len(ints), len(flts), fn, len(words), len(floats)
            words += ints
            floats += flts
            flags += fl
        self.words = (C.c_ulonglong * max(len(words), 1))(*words)
        self.floats = (C.c_float * max(len(floats), 1))(*floats)
        self.names = [op[2] for op in rec.ops]
        self.cuts = list(rec.cuts) + [n]
        self.py_ops = list(rec.py_ops)
        self.keep = rec.keep
        self.slots = {}
        for key, val in dynamic.items():
            pos = [i for i, (w, f) in enumerate(zip(words, flags)) if f and val and w == val]
            self.slots[key] = pos
        self._failed = C.c_int(-1)
        # what a second recording must reproduce: everything but the dynamic slots and the event handles
        dyn = set(p for pos in self.slots.values() for p in pos)
        sig = []
        for (kind, fn, name, ints, flts, fl), o in zip(rec.ops, self.ops):
            if kind == CALL:
                sig.append((name, fn, tuple(None if (o.word0 + j) in dyn else w for j, w in enumerate(ints)), tuple(flts)))
            else:
                sig.append((name, ints[1] if kind == RECORD else ints[0]))          # the stream; the events are each recording's own
        self.signature = (tuple(sig), tuple(self.cuts))

    def run(self, dynamic):
        for key, pos in self.slots.items():
            v = int(dynamic[key]) & 0xFFFFFFFFFFFFFFFF
            for p in pos:
                self.words[p] = v
        first = 0
        for k, last in enumerate(self.cuts):
            if last > first:
                ops = C.cast(C.byref(self.ops, first * C.sizeof(_lib.PlanOp)), C.POINTER(_lib.PlanOp))
                rc = _lib.lib.mvf_plan_run(ops, last - first, self.words, self.floats, C.byref(self._failed))
                if rc != 0:
                    i = first + self._failed.value
                    _lib.check(rc, "launch plan, op %d (%s)" % (i, self.names[i] if 0 <= i < self.n_ops else "?"))
            if k < len(self.py_ops):
                self.py_ops[k]()
            first = last
