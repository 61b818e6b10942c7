// [r6] Replay of a prebuilt launch table from C: mvf_plan_run.
//
// A training step of the engine is ~650 calls of this library's own entry points in a fixed order on two HIP streams, with the same arguments step after step
// (every operand is a persistent buffer, descriptors are immutable): 4 ms of Python + ctypes per step for a sequence that never changes -- half of a 9 ms step at
// the reference's own batch size (12 clips per GPU, configs/MVFNet/K400/mvf_kinetics400_2d_rgb_r50_dense.py:121-123; codes/core/train.py:45-60 drives one such
// step per iteration), and eight such processes share one host on an 8-GPU node.  hipGraph replay of the same sequence measured SLOWER than eager launches
// (profiles/r05_step_timeline_gaps.txt), so the launches stay eager and only their ISSUE moves to C: the host code (mvfnet_amd/launch_plan.py) records one step's
// calls -- function address, integer-class argument words, float arguments -- and the cross-stream orderings between them, and hands the table to
// mvf_plan_run, which walks it: a CALL is the entry point invoked with exactly the recorded arguments (so every launch goes through the same validation and
// the same kernels as an eager call), RECORD / WAIT are hipEventRecord / hipStreamWaitEvent on caller-owned events.  Arguments that change from step to step (the
// input batch, the labels, the dropout mask, the loss tensor) are patched into the word array by the caller before the run.
//
// The generic call: on x86-64 System V the integer-class arguments of a call go to rdi, rsi, rdx, rcx, r8, r9 and then to the stack IN ORDER, the float arguments
// to xmm0.. IN ORDER, independently of how the two classes interleave in the prototype (floats never reach the stack with <= 8 of them), so an entry point with
// NI integer-class and NF float arguments is callable as int (*)(uint64 x NI, float x NF).  32-bit integer arguments are read by the callee from the low half of
// their 64-bit slot.
#include <utility>

#include "common.h"

#if !defined(__x86_64__)
#error "launch_plan.hip: the generic call relies on the x86-64 System V calling convention"
#endif

namespace {

constexpr int kMaxInt = 40, kMaxFlt = 4;
typedef int (*CallFn)(void*, const uint64_t*, const float*);

template <typename T, size_t>
using always_t = T;

template <size_t... I, size_t... F>
int call_seq(void* fn, const uint64_t* a, const float* f, std::index_sequence<I...>, std::index_sequence<F...>) {
    typedef int (*Fn)(always_t<uint64_t, I>..., always_t<float, F>...);
    return reinterpret_cast<Fn>(fn)(a[I]..., f[F]...);
}
template <size_t NI, size_t NF>
int call_nf(void* fn, const uint64_t* a, const float* f) {
    return call_seq(fn, a, f, std::make_index_sequence<NI>{}, std::make_index_sequence<NF>{});
}
template <size_t NI, size_t... F>
constexpr void fill_row(CallFn (&row)[kMaxFlt + 1], std::index_sequence<F...>) {
    ((row[F] = &call_nf<NI, F>), ...);
}
struct Table {
    CallFn fn[kMaxInt + 1][kMaxFlt + 1];
    template <size_t... I>
    constexpr void fill(std::index_sequence<I...>) {
        (fill_row<I>(fn[I], std::make_index_sequence<kMaxFlt + 1>{}), ...);
    }
    constexpr Table() : fn{} { fill(std::make_index_sequence<kMaxInt + 1>{}); }
};
const Table g_table;

}  // namespace

extern "C" {

int mvf_plan_run(const mvf_plan_op_t* ops, int n_ops, const unsigned long long* words_, const float* floats, int* failed_op) {
    const uint64_t* words = reinterpret_cast<const uint64_t*>(words_);
    MVF_REQUIRE(ops && n_ops >= 0 && words_ && floats, MVF_EINVAL, "plan_run: NULL argument");
    for (int i = 0; i < n_ops; ++i) {
        const mvf_plan_op_t& op = ops[i];
        int rc = MVF_OK;
        switch (op.kind) {
        case MVF_PLAN_CALL:
            if (!op.fn || op.n_int < 0 || op.n_int > kMaxInt || op.n_flt < 0 || op.n_flt > kMaxFlt) {
                mvf_set_error("plan_run: op %d: bad call record (%d integer / %d float arguments)", i, op.n_int, op.n_flt);
                rc = MVF_EINVAL;
            } else {
                rc = g_table.fn[op.n_int][op.n_flt](op.fn, words + op.word0, floats + op.float0);      // (the callee sets mvf_last_error on failure)
            }
            break;
        case MVF_PLAN_RECORD:      // words: event, stream
            if (hipEventRecord((hipEvent_t)words[op.word0], (hipStream_t)words[op.word0 + 1]) != hipSuccess) {
                mvf_set_error("plan_run: op %d: hipEventRecord failed", i);
                rc = MVF_EHIP;
            }
            break;
        case MVF_PLAN_WAIT:        // words: stream, event
            if (hipStreamWaitEvent((hipStream_t)words[op.word0], (hipEvent_t)words[op.word0 + 1], 0) != hipSuccess) {
                mvf_set_error("plan_run: op %d: hipStreamWaitEvent failed", i);
                rc = MVF_EHIP;
            }
            break;
        default:
            mvf_set_error("plan_run: op %d: unknown kind %d", i, op.kind);
            rc = MVF_EINVAL;
        }
        if (rc != MVF_OK) {
            if (failed_op) *failed_op = i;
            return rc;
        }
    }
    if (failed_op) *failed_op = -1;
    return MVF_OK;
}

}  // extern "C"
