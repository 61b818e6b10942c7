"""HIP streams that really run beside a given stream.

ROCm multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order; two streams
that land on the same queue execute strictly one after the other, whatever the events say.  Which queue a new stream gets depends
on how many streams the process created before -- and `torch.distributed` (RCCL, c10d's stream pool) creates dozens: measured on
MI355X, the train engine's weight-gradient side stream shared the launch stream's queue as soon as a process group existed, and
the step went from 23.9 to 25.8 ms with every overlap silently gone.  So a side stream is not taken on faith: `concurrent_stream`
creates candidates and keeps the first one whose tiny kernel finishes while a ~1 ms spin kernel is still running on `main`."""
import torch

_SPIN_CYCLES = 3000000          # ~1.2 ms at 2.4 GHz


def runs_beside(main, cand):
    """True when a kernel queued on `cand` AFTER a long kernel was queued on `main` completes before that long kernel does."""
    e_end = torch.cuda.Event(enable_timing=True)
    e_c = torch.cuda.Event(enable_timing=True)
    probe = torch.zeros(64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        torch.cuda._sleep(_SPIN_CYCLES)
        e_end.record(main)
    with torch.cuda.stream(cand):
        probe.add_(1.0)
        e_c.record(cand)
    torch.cuda.synchronize()
    return e_c.elapsed_time(e_end) > 0.2          # ms from the candidate's kernel to the end of the spin: > 0 only if it overtook


def concurrent_stream(main=None, avoid=(), tries=12):
    """A new stream that overlaps with `main` (and with every stream in `avoid`).  Falls back to the last candidate when none of
    `tries` does (a device with a single queue) -- correctness never depends on the overlap, only speed."""
    main = main or torch.cuda.current_stream()
    if not hasattr(torch.cuda, "_sleep"):          # no spin kernel to measure against: take the stream on faith
        return torch.cuda.Stream()
    cand = None
    keep = []                                     # rejected candidates stay alive until the search ends: a destroyed stream's
    for _ in range(tries):                        # queue slot would be handed straight to the next candidate
        cand = torch.cuda.Stream()
        try:
            if all(runs_beside(s, cand) for s in (main,) + tuple(avoid)):
                return cand
        except RuntimeError:                      # event timing unavailable (e.g. inside a graph capture): keep the candidate
            return cand
        keep.append(cand)
    return cand
