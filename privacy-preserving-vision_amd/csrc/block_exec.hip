// Whole-bottleneck launchers: one C-ABI call enqueues every kernel of a torchvision Bottleneck (Image_Caption/models.py:17-21 under
// train-mode BatchNorm, train.py:245) in the order ppv_amd/encoder.py enqueues them one by one.  Nothing new runs on the device --
// the functions below only call the entry points of conv_gemm.hip / trunk_ops.hip / conv_wgrad_stem.hip -- what changes is the host:
// the Python step crosses ctypes ~1000 times (15 ms per 22-ms step on a fast host, 23 ms on a slow one: the step then runs at the
// speed of the interpreter); per bottleneck that is 6 crossings forward and ~12 backward, here one each.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "ppv_common.h"
#include "ppv_hip.h"

namespace ppv {
void conv_set_output_nt_once(int nt);       // conv_gemm.hip
void bn_bwd_stop_event_once(hipEvent_t e);  // trunk_ops.hip
bool bn_bwd_stop_event_unused();
int fork_stop_event(hipEvent_t* out);       // trunk_plan.hip
int fork_flag_next(hipStream_t main, unsigned long long** word, unsigned long long* val);
int fork_flag_wait(hipStream_t side, unsigned long long* word, unsigned long long val);
int fork_flag_settle(hipStream_t main, unsigned long long* word, unsigned long long val);
void conv_set_start_flag_once(unsigned long long* flag, unsigned long long val);   // conv_gemm.hip
}

extern "C" {

// forward of a bottleneck WITHOUT projection shortcut, statistics folded by the apply kernels (ppv_bn_act_fold_rows)
int ppv_bottleneck_fwd(const PpvBottleneckFwd* a, hipStream_t stream) {
    if (!a) return PPV_ERR_NULL;
    const int B = a->B, H = a->H, W = a->W, Cin = a->Cin, P = a->planes, st = a->stride;
    if (st != 1 && st != 2) return PPV_ERR_BAD_SIZE;
    const int H2 = H / st, W2 = W / st, C3 = 4 * P;
    if (Cin != C3) return PPV_ERR_BAD_SIZE;                 // identity shortcut: same width and (stride 1) same map
    if (st != 1) return PPV_ERR_BAD_SIZE;
    const long M1 = (long)B * H * W, M2 = (long)B * H2 * W2;
    int e;
    // conv1 1x1 -> bn1 + ReLU
    // PPV_BNIN=1: bn1 + ReLU inside conv2's halo kernel.  Opt-in: faster alone (45 against 52 us at the layer-3 shape, tools/bench_bnin.py)
    // but not in the step (profiles/r06_bnin_ab.json)
    static const int bnin_on = getenv("PPV_BNIN") ? atoi(getenv("PPV_BNIN")) : 0;
    const bool bnin = bnin_on && st == 1 && ppv_conv3x3_bnin_supported(B, H, W, P, P);
    static const int x1_temporal = getenv("PPV_BNIN_X1T") ? atoi(getenv("PPV_BNIN_X1T")) : 1;     // A/B: conv1's raw output kept in the L2 for the BNIN kernel
    if (bnin && x1_temporal) ppv::conv_set_output_nt_once(0);
    if ((e = ppv_conv_gemm(a->xin, a->w1, a->x1, a->stats1, nullptr, nullptr, a->zero_page, B, H, W, Cin, H, W, P, 1, 1, 1, 0, 1, 0, a->T1, stream))) return e;
    // bn1 + ReLU: inside conv2 where its halo form runs (round 6: the activation is applied to the LDS-resident input tile; y1 is still
    // written, from that kernel, for conv2's weight gradient), a launch of its own elsewhere
    if (bnin) {
        if ((e = ppv_conv3x3_bnin(a->x1, a->stats1, a->T1, (double)M1, a->g1, a->b1, a->rm1, a->rv1, a->mom1, a->eps1, a->coef1, a->y1, a->w2, a->x2,
                                  a->stats2, a->T2, a->zero_page, B, H, W, P, P, stream))) return e;
    } else {
        if ((e = ppv_bn_act_fold_rows(a->x1, a->stats1, a->T1, (double)M1, a->g1, a->b1, a->rm1, a->rv1, a->mom1, a->eps1, a->coef1, nullptr, a->y1,
                                      nullptr, M1 * P, P, 0, 1, stream))) return e;
        // conv2 3x3 (stride on the 3x3: torchvision v1.5) -> bn2 + ReLU
        if ((e = ppv_conv_gemm(a->y1, a->w2, a->x2, a->stats2, nullptr, nullptr, a->zero_page, B, H, W, P, H2, W2, P, 3, 3, st, -1, 1, 0, a->T2, stream))) return e;
    }
    if ((e = ppv_bn_act_fold_rows(a->x2, a->stats2, a->T2, (double)M2, a->g2, a->b2, a->rm2, a->rv2, a->mom2, a->eps2, a->coef2, nullptr, a->y2,
                                  nullptr, M2 * P, P, 0, 1, stream))) return e;
    // conv3 1x1 -> bn3 + identity + ReLU (+ the (y > 0) bit mask the backward pass reads)
    if ((e = ppv_conv_gemm(a->y2, a->w3, a->x3, a->stats3, nullptr, nullptr, a->zero_page, B, H2, W2, P, H2, W2, C3, 1, 1, 1, 0, 1, 0, a->T3, stream))) return e;
    return ppv_bn_act_fold_rows(a->x3, a->stats3, a->T3, (double)M2, a->g3, a->b3, a->rm3, a->rv3, a->mom3, a->eps3, a->coef3, a->xin, a->yout,
                                a->bits, M2 * C3, C3, 1, 1, stream);
}

// backward of the same block, in the order encoder.py's default schedule enqueues it (weight gradients on `side` as soon as their
// operand exists, behind an event recorded on `main`):
//   bn3' -> [wgrad3] -> dgrad3 (+ bn2 sums) -> bn2' -> [wgrad2] -> dgrad2 (+ bn1 sums) -> bn1' -> [wgrad1] -> dgrad1 (+ residual gradient
//   + ReLU mask of the block input + bn3 sums of the block this gradient flows into)
namespace {
// event record on `main` + wait on `side` through the guarded per-device event ring of trunk_plan.hip
int fork_to(hipStream_t main, hipStream_t side) { return ppv_stream_fork(main, side); }

// Paired weight gradients (round 6, ppv_conv_wgrad_pair): inside a ppv_trunk_bwd call (wgrad_pair_begin .. wgrad_pair_end) an identity
// bottleneck does NOT launch its conv1 weight gradient; the next bottleneck to run launches it together with its own conv3 weight
// gradient (both operands exist by then; the buffers are per block and outlive the call).  Per thread: the executor is re-entrant.
struct PendingW1 {
    bool valid;
    const void *G, *X, *zero_page;
    float* dW;
    void* scratch;
    int B, H, W, Cs, N;
    int R = 1, pad = 0;
};
thread_local PendingW1 t_pend = {false, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
thread_local PendingW1 t_pend2 = {false, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};     // conv2's (PPV_WGRAD_FORKS=3)
thread_local size_t t_pair_scratch = 0;         // > 0: pairing is on, bytes of the shared slab region
thread_local bool t_hold_next = false;          // PPV_WGRAD_FORKS=3 inside a ppv_trunk_bwd call: conv2 / conv1 wait for the next block's fork

int launch_pending(PendingW1& p, hipStream_t ws) {
    if (!p.valid) return PPV_OK;
    p.valid = false;
    return ppv_conv_wgrad(p.G, p.X, p.dW, p.scratch, p.zero_page, p.B, p.H, p.W, p.Cs, p.H, p.W, p.N, p.R, p.R, 1, p.pad, ws);
}
}  // namespace

}  // extern "C" (the executor-internal hooks below have C++ linkage: trunk_plan.hip declares them in namespace ppv)

namespace ppv {
void wgrad_pair_begin(size_t scratch_bytes, bool defer_to_next_block) {
    t_pair_scratch = scratch_bytes;
    t_hold_next = defer_to_next_block;
    t_pend.valid = t_pend2.valid = false;
}
// the weight gradients still pending (if any) on their own; join: `ws` first waits for what `main` has enqueued so far (`main` may be the
// null stream: a flag, not the pointer, says whether to wait)
int wgrad_pair_flush(hipStream_t main, hipStream_t ws, bool join) {
    if (!t_pend.valid && !t_pend2.valid) return PPV_OK;
    if (join && main != ws) { if (int e = ppv_stream_fork(main, ws)) return e; }
    if (int e = launch_pending(t_pend2, ws)) return e;
    return launch_pending(t_pend, ws);
}
int wgrad_pair_end(hipStream_t main, hipStream_t ws) {
    const int e = wgrad_pair_flush(main, ws, true);
    t_pair_scratch = 0;
    t_hold_next = false;
    return e;
}
}  // namespace ppv

extern "C" {

int ppv_bottleneck_bwd(const PpvBottleneckBwd* a, hipStream_t main, hipStream_t side) {
    if (!a) return PPV_ERR_NULL;
    const int B = a->B, H = a->H, W = a->W, P = a->planes, C3 = 4 * P;
    const long M = (long)B * H * W;
    hipStream_t ws = side ? side : main;
    int e;
    // the three weight gradients keep their slabs in three regions of wscratch (wstride bytes apart; 0: one region, reduce at once) and
    // are reduced by ONE launch at the end of the block
    PpvWgradReduce red[3];
    red[0].blocks = red[1].blocks = red[2].blocks = 0;
    const bool defer = a->wstride > 0;
    char* wsb = (char*)a->wscratch;
    // bn3 backward (gradient arrives masked by the block output's ReLU; sums possibly taken by the data-gradient launch that produced it)
    // PPV_FORK_STOPEV (round 6; default 1, 0 = event records): the fork behind a BatchNorm-backward launch waits for an event that launch's OWN dispatch packet
    // signals (hipExtLaunchKernel stopEvent) instead of an event-record packet enqueued behind it on the main chain
    static const int stopev_on = getenv("PPV_FORK_STOPEV") ? atoi(getenv("PPV_FORK_STOPEV")) : 1;
    bool use_stop = stopev_on && side && side != main;
    if (use_stop) {                                   // a stop event does not fork a stream CAPTURE over to `side`: event records there
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) use_stop = false;
    }
    // PPV_FORK_FLAG=1 (round 6, opt-in: MEASURED LOSS): no packet on the main chain at all -- the side stream waits (hipStreamWaitValue64)
    // for a value the NEXT launch of the main chain stores when it starts (trunk_plan.hip fork_flag_*).  The main chain's gaps do go, but
    // the step is 0.35-0.8 ms SLOWER (profiles/r06_fork_ab.json): the wait-value packet holds the side queue far longer than an event
    // wait does.  Falls back to the stop event / event record where the device cannot wait on memory, inside a capture, and in the A/B
    // fork schedules.
    static const int flag_on = getenv("PPV_FORK_FLAG") ? atoi(getenv("PPV_FORK_FLAG")) : 0;
    static const int fork_mode00 = getenv("PPV_WGRAD_FORKS") ? atoi(getenv("PPV_WGRAD_FORKS")) : 0;
    bool use_flag = flag_on && use_stop && fork_mode00 == 0 && !(a->wstride > 0) && !(t_pair_scratch > 0);   // (use_stop: two streams, no capture)
    // a flag the side stream already waits for is ALWAYS reached: by the launch that carries it, or (any early return) by the main chain
    struct FlagGuard {
        hipStream_t main;
        unsigned long long* word = nullptr;
        unsigned long long val = 0;
        bool pending = false;
        ~FlagGuard() { if (pending) (void)ppv::fork_flag_settle(main, word, val); }
    } fg{main};
    hipEvent_t ev3 = nullptr, ev2 = nullptr, ev1 = nullptr;
    auto arm = [&](hipEvent_t& ev, bool wanted) -> int {
        ev = nullptr;
        if (!use_stop || use_flag || !wanted) return PPV_OK;
        if (int r = ppv::fork_stop_event(&ev)) return r;
        ppv::bn_bwd_stop_event_once(ev);
        return PPV_OK;
    };
    auto armed = [&](hipEvent_t& ev) { if (ev && ppv::bn_bwd_stop_event_unused()) ev = nullptr; };   // (not the fused apply kernel: plain fork)
    auto fork_on = [&](hipEvent_t ev) -> int {
        if (!side) return PPV_OK;
        if (use_flag) {
            if (ppv::fork_flag_next(main, &fg.word, &fg.val) == PPV_OK) {
                if (int r = ppv::fork_flag_wait(side, fg.word, fg.val)) return r;
                fg.pending = true;
                return PPV_OK;
            }
            use_flag = false;                          // (not available here: event records from now on)
        }
        if (ev) { if (hipError_t r = hipStreamWaitEvent(side, ev, 0)) return -(int)r; return PPV_OK; }
        return fork_to(main, side);
    };
    static const int fork_mode0 = getenv("PPV_WGRAD_FORKS") ? atoi(getenv("PPV_WGRAD_FORKS")) : 0;
    const bool stop_ok = (fork_mode0 == 0 || fork_mode0 == 1) && !(a->wstride > 0) && !(t_pair_scratch > 0);     // three / two forks per block
    // the main-chain launch behind a flag fork carries the flag; if it does not launch, the main chain stores the value itself
    auto carry = [&]() { if (fg.pending) ppv::conv_set_start_flag_once(fg.word, fg.val); };
    auto settled = [&](int rc) -> int {
        if (fg.pending && rc == PPV_OK) fg.pending = false;       // launched: the kernel stores the value (otherwise ~FlagGuard does)
        return rc;
    };
    if ((e = arm(ev3, stop_ok && a->dw3 != nullptr))) return e;
    if ((e = ppv_bn_bwd(a->g, nullptr, a->x3, a->c3, (double)M, a->gx3, nullptr, a->dg3, a->db3, a->part3, a->kc3, M, C3, 0, a->part3_ready ? 2 : 1, main))) return e;
    armed(ev3);
    const bool pairing = t_pair_scratch > 0 && !defer;
    // PPV_WGRAD_FORKS (round 6 A/B): every fork is an event record on `main`, and the kernel trace shows ~6 us of idle main chain behind each
    // (none between the forward launches).  0 (default): one fork per weight gradient (three per block); 1: conv2's weight gradient waits
    // for conv1's fork (two per block); 2: all three behind ONE fork after bn1' (operands up to ~150 us older when they are read).
    static const int fork_mode = getenv("PPV_WGRAD_FORKS") ? atoi(getenv("PPV_WGRAD_FORKS")) : 0;
    // 3 (inside ppv_trunk_bwd): ONE fork per block, behind bn3': this block's conv3 + conv2 / conv1 of the block that ran before
    const bool hold = t_hold_next && fork_mode == 3 && !defer && !pairing;
    const int fm = (pairing || defer) ? 0 : (fork_mode == 3 ? (hold ? 3 : 1) : fork_mode);
    if (a->dw3 && fm != 2) {
        if ((e = fork_on(ev3))) return e;
        if (pairing && t_pend.valid && t_pend.B == B &&
            ppv_conv_wgrad_pair_supported(B, t_pend.H, t_pend.W, t_pend.Cs, t_pend.N, H, W, P, C3)) {
            // conv1 of the bottleneck that ran before this one + this conv3: one launch, one reduce launch
            t_pend.valid = false;
            if ((e = ppv_conv_wgrad_pair(t_pend.G, t_pend.X, t_pend.dW, t_pend.H, t_pend.W, t_pend.Cs, t_pend.N, a->gx3, a->y2, a->dw3, H, W, P, C3, wsb,
                                         t_pair_scratch, a->zero_page, B, ws))) return e;
        } else {
            if ((pairing || hold) && (e = ppv::wgrad_pair_flush(main, ws, false))) return e;     // (pending ones: on their own, same point; forked above)
            if ((e = ppv_conv_wgrad_ex(a->gx3, a->y2, a->dw3, wsb, a->zero_page, B, H, W, P, H, W, C3, 1, 1, 1, 0, ws, defer ? &red[0] : nullptr))) return e;
        }
    }
    // conv3 data gradient (+ bn2's sums and recomputed ReLU mask)
    carry();
    if (a->red2) e = ppv_conv_gemm_red(a->gx3, a->wd3, a->gy2, a->part2, a->x2, a->c2, nullptr, nullptr, a->zero_page, B, H, W, C3, H, W, P, 1, 1, 1, 0, 1, 8, main);
    else e = ppv_conv_gemm(a->gx3, a->wd3, a->gy2, nullptr, nullptr, nullptr, a->zero_page, B, H, W, C3, H, W, P, 1, 1, 1, 0, 1, 0, 0, main);
    if (settled(e)) return e;
    if ((e = arm(ev2, stop_ok && fork_mode0 == 0 && a->dw2 != nullptr))) return e;
    if ((e = ppv_bn_bwd(a->gy2, nullptr, a->x2, a->c2, (double)M, a->gx2, nullptr, a->dg2, a->db2, a->part2, a->kc2, M, P, a->red2 ? 0 : 2, a->red2 ? 2 : 1, main))) return e;
    armed(ev2);
    if (a->dw2 && fm == 0) {
        if ((e = fork_on(ev2))) return e;
        if ((e = ppv_conv_wgrad_ex(a->gx2, a->y1, a->dw2, wsb + (defer ? a->wstride : 0), a->zero_page, B, H, W, P, H, W, P, 3, 3, 1, 1, ws, defer ? &red[1] : nullptr))) return e;
    }
    // conv2 data gradient (+ bn1's sums)
    carry();
    if (a->red1) e = ppv_conv_gemm_red(a->gx2, a->wd2, a->gy1, a->part1, a->x1, a->c1, nullptr, nullptr, a->zero_page, B, H, W, P, H, W, P, 3, 3, 1, -1, 1, 8, main);
    else e = ppv_conv_gemm(a->gx2, a->wd2, a->gy1, nullptr, nullptr, nullptr, a->zero_page, B, H, W, P, H, W, P, 3, 3, 1, -1, 1, 0, 0, main);
    if (settled(e)) return e;
    if ((e = arm(ev1, stop_ok && (a->dw1 != nullptr || (fork_mode0 == 1 && a->dw2 != nullptr))))) return e;
    if ((e = ppv_bn_bwd(a->gy1, nullptr, a->x1, a->c1, (double)M, a->gx1, nullptr, a->dg1, a->db1, a->part1, a->kc1, M, P, a->red1 ? 0 : 2, a->red1 ? 2 : 1, main))) return e;
    armed(ev1);
    if (fm == 3) {
        if (!a->dw3 && (e = ppv::wgrad_pair_flush(main, ws, true))) return e;          // (no fork of this block took the pending ones along)
        if (a->dw2) { t_pend2 = PendingW1{true, a->gx2, a->y1, a->zero_page, a->dw2, (void*)wsb, B, H, W, P, P, 3, 1}; }
        if (a->dw1) { t_pend = PendingW1{true, a->gx1, a->xin, a->zero_page, a->dw1, (void*)wsb, B, H, W, C3, P, 1, 0}; }
    } else if (fm != 0 && (a->dw3 || a->dw2 || a->dw1)) {
        if ((e = fork_on(fm == 1 ? ev1 : nullptr))) return e;
        if (fm == 2 && a->dw3 && (e = ppv_conv_wgrad(a->gx3, a->y2, a->dw3, wsb, a->zero_page, B, H, W, P, H, W, C3, 1, 1, 1, 0, ws))) return e;
        if (a->dw2 && (e = ppv_conv_wgrad(a->gx2, a->y1, a->dw2, wsb, a->zero_page, B, H, W, P, H, W, P, 3, 3, 1, 1, ws))) return e;
        if (a->dw1 && (e = ppv_conv_wgrad(a->gx1, a->xin, a->dw1, wsb, a->zero_page, B, H, W, C3, H, W, P, 1, 1, 1, 0, ws))) return e;
    } else if (a->dw1) {
        if (pairing && P % 256 == 0) {
            // left for the next bottleneck's call (or the executor's flush): operands gx1 / xin live in per-block buffers
            if ((e = ppv::wgrad_pair_flush(main, ws, true))) return e;
            t_pend = PendingW1{true, a->gx1, a->xin, a->zero_page, a->dw1, (void*)wsb, B, H, W, C3, P};
        } else {
            if ((e = fork_on(ev1))) return e;
            if ((e = ppv_conv_wgrad_ex(a->gx1, a->xin, a->dw1, wsb + (defer ? 2 * a->wstride : 0), a->zero_page, B, H, W, C3, H, W, P, 1, 1, 1, 0, ws, defer ? &red[2] : nullptr))) return e;
            if (defer && (e = ppv_wgrad_reduce_multi(red, 3, ws))) return e;
        }
    }
    if (defer && !a->dw1 && (e = ppv_wgrad_reduce_multi(red, 3, ws))) return e;      // (a block whose conv1 alone is frozen)
    // conv1 data gradient + the identity branch's gradient + the block input's ReLU mask (+ the sums bn3 of the NEXT block to run needs)
    carry();
    if (a->x3_prev) return settled(ppv_conv_gemm_red(a->gx1, a->wd1, a->gin, a->part3_prev, a->x3_prev, nullptr, a->g, a->xin_bits, a->zero_page, B, H, W, P, H, W, C3, 1, 1, 1, 0, 1, 8, main));
    return settled(ppv_conv_gemm(a->gx1, a->wd1, a->gin, nullptr, a->g, a->xin_bits, a->zero_page, B, H, W, P, H, W, C3, 1, 1, 1, 0, 1, 0, 0, main));
}

}  // extern "C"
