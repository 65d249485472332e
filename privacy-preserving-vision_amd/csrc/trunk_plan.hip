// Whole-trunk executor: ONE C-ABI call enqueues the forward of the ResNet-101 Encoder (Image_Caption/models.py:31-41 under train-mode
// BatchNorm, train.py:245) and ONE its backward -- stem, the projection bottlenecks, the identity bottlenecks (block_exec.hip), both
// pools, the weight-gradient forks to the side stream and the final join -- over ONE caller-provided arena.
//
// Nothing new runs on the device: every launch below is an entry point of conv_gemm.hip / trunk_ops.hip / conv_wgrad_stem.hip with the
// arguments ppv_amd/encoder.py's per-kernel path passes, in the same order.  What changes is the host: the interpreter used to cross
// ctypes ~200 times per step, take ~460 tensors from the caching allocator and zero two pools with their own launches; here a step is
// two crossings, no allocation (offsets into the arena are a pure function of PpvTrunkDesc) and one memset per direction for the
// BatchNorm partial-sum buffers of the step.
//
// Arena (ppv_trunk_arena_bytes): [ zero zone | saved activations per block | gradient work buffers per block | scratch ].
//   zero zone  : forward statistics [rows][2][C] of every convolution (cleared by ppv_trunk_fwd with one hipMemsetAsync) and backward
//                sums [64 C] of every BatchNorm (cleared by ppv_trunk_bwd when it starts at the last block: a repeated backward works).
//   saved      : what backward reads (raw conv outputs, activations, sign masks, BatchNorm coefficients) -- written by forward, valid
//                until the next ppv_trunk_fwd on the same arena.
//   work       : the gradients that never leave a block; every block has its own (the side stream reads them for the weight gradients,
//                so a buffer shared between blocks would need a cross-stream wait per block).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include "ppv_common.h"
#include "ppv_hip.h"

namespace ppv {                            // conv_gemm.hip (declared in conv_common.h): compact addend of the next ppv_conv_gemm* call
void conv_set_addend_compact(bool on);
bool conv_addend_compact_supported(int B, int H, int W, int Cs, int N);
// block_exec.hip: paired weight gradients across consecutive identity bottlenecks (ppv_conv_wgrad_pair)
void wgrad_pair_begin(size_t scratch_bytes, bool defer_to_next_block);
int wgrad_pair_flush(hipStream_t main, hipStream_t ws, bool join);
int wgrad_pair_end(hipStream_t main, hipStream_t ws);
// trunk_ops.hip: the next fused BatchNorm-backward apply launch signals an event from its own dispatch packet (see block_exec.hip)
void bn_bwd_stop_event_once(hipEvent_t e);
bool bn_bwd_stop_event_unused();
int fork_stop_event(hipEvent_t* out);
void conv_set_start_flag_once(unsigned long long* flag, unsigned long long val);   // conv_gemm.hip
}

namespace {

inline size_t a256(size_t n) { return (n + 255) & ~(size_t)255; }

struct BlkOff {
    int Hin, Win, H2, W2, Cin, P, st, proj, T12a, T12b, T3, Td;     // T12a: rows of conv1's statistics, T12b: conv2's
    size_t st1, st2, st3, std_;            // forward statistics (zero zone), bytes from the arena base
    size_t p1, p2, p3, pd;                 // backward sums (zero zone)
    size_t x1, y1, x2, y2, x3, xd, yout, bits, c1, c2, c3, cd;
    size_t gx3, gy2, gx2, gy1, gx1, gxd, gind, gin;
};

struct Layout {
    size_t zero_bytes, bzero_off, bzero_bytes, total;       // [0, zero_bytes): forward statistics; [bzero_off, +bzero_bytes): backward sums
    int T0;
    size_t st0, raw0, c0, y0, arg0;        // stem: statistics, raw conv output, coefficients, pooled activation, arg-max
    size_t gx0, part0, tmp0, gtop, kc, wscr;
    size_t kc_stride, wstride;
    BlkOff b[PPV_TRUNK_MAX_BLOCKS];
};

inline bool red_ok(long rows, int C) { return C % 128 == 0 || (C % 64 == 0 && rows >= 128 * 1024); }   // shapes ppv_conv_gemm_red serves

int make_layout(const PpvTrunkDesc* d, Layout* L) {
    if (!d) return PPV_ERR_NULL;
    if (d->nblocks < 1 || d->nblocks > PPV_TRUNK_MAX_BLOCKS || d->B < 1 || d->H % 32 || d->W % 32 || d->H < 32 || d->W < 32) return PPV_ERR_BAD_SIZE;
    const int fold = d->fold_rows < 1 ? 1 : d->fold_rows;
    const long B = d->B;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += a256(bytes); return o; };
    // ---- zero zone
    const int Hs = d->H / 2, Ws = d->W / 2;
    L->T0 = ppv_conv_stat_tiles(B * Hs * Ws);
    L->st0 = take((size_t)L->T0 * 2 * 64 * 4);
    int h = d->H / 4, w = d->W / 4, cin = 64, maxC = 64;
    size_t wneed = 16;
    for (int i = 0; i < d->nblocks; i++) {
        const PpvTrunkBlock& k = d->blk[i];
        BlkOff& o = L->b[i];
        if (k.planes < 64 || k.planes % 64 || (k.stride != 1 && k.stride != 2)) return PPV_ERR_BAD_SIZE;
        if (!k.proj && (k.stride != 1 || cin != 4 * k.planes)) return PPV_ERR_BAD_SIZE;
        if (h % k.stride || w % k.stride) return PPV_ERR_BAD_SIZE;
        o.Hin = h; o.Win = w; o.st = k.stride; o.H2 = h / k.stride; o.W2 = w / k.stride; o.Cin = cin; o.P = k.planes; o.proj = k.proj;
        const long M1 = B * h * w, M2 = B * o.H2 * o.W2;
        const int P = k.planes, C3 = 4 * P;
        const int t1 = ppv_conv_stat_tiles(M1), t2 = ppv_conv_stat_tiles(M2);
        o.T12a = fold < t1 ? fold : t1;
        o.T12b = fold < t2 ? fold : t2;
        o.T3 = k.proj ? t2 : o.T12b;                  // projection blocks keep the coefficient launch for bn3 / the shortcut's BatchNorm
        o.Td = k.proj ? t2 : 0;
        o.st1 = take((size_t)o.T12a * 2 * P * 4);
        o.st2 = take((size_t)o.T12b * 2 * P * 4);
        o.st3 = take((size_t)o.T3 * 2 * C3 * 4);
        o.std_ = k.proj ? take((size_t)o.Td * 2 * C3 * 4) : 0;
        if (C3 > maxC) maxC = C3;
        // weight-gradient slabs: the largest of the trainable convolutions (bit 0 conv1, 1 conv2, 2 conv3, 3 shortcut)
        if (k.train_w & 1) { size_t n = ppv_conv_wgrad_scratch_bytes(M1, P, 1, 1, cin); if (n > wneed) wneed = n; }
        if (k.train_w & 2) { size_t n = ppv_conv_wgrad_scratch_bytes(M2, P, 3, 3, P); if (n > wneed) wneed = n; }
        if (k.train_w & 4) { size_t n = ppv_conv_wgrad_scratch_bytes(M2, C3, 1, 1, P); if (n > wneed) wneed = n; }
        if (k.proj && (k.train_w & 8)) { size_t n = ppv_conv_wgrad_scratch_bytes(M2, C3, 1, 1, cin); if (n > wneed) wneed = n; }
        h = o.H2; w = o.W2; cin = C3;
    }
    L->zero_bytes = off;
    L->bzero_off = off;
    for (int i = 0; i < d->nblocks; i++) {
        BlkOff& o = L->b[i];
        const size_t P = o.P, C3 = 4 * P;
        o.p1 = take(64 * P * 4);
        o.p2 = take(64 * P * 4);
        o.p3 = take(64 * C3 * 4);
        o.pd = o.proj ? take(64 * C3 * 4) : 0;
    }
    L->bzero_bytes = off - L->bzero_off;
    // ---- saved by forward
    L->raw0 = take((size_t)B * Hs * Ws * 64 * 2);
    L->c0 = take(4 * 64 * 4);
    L->y0 = take((size_t)B * (Hs / 2) * (Ws / 2) * 64 * 2);
    L->arg0 = take((size_t)B * (Hs / 2) * (Ws / 2) * 64);
    for (int i = 0; i < d->nblocks; i++) {
        BlkOff& o = L->b[i];
        const size_t M1 = (size_t)B * o.Hin * o.Win, M2 = (size_t)B * o.H2 * o.W2, P = o.P, C3 = 4 * P;
        o.x1 = take(M1 * P * 2); o.y1 = take(M1 * P * 2);
        o.x2 = take(M2 * P * 2); o.y2 = take(M2 * P * 2);
        o.x3 = take(M2 * C3 * 2);
        o.xd = o.proj ? take(M2 * C3 * 2) : 0;
        o.yout = take(M2 * C3 * 2);
        o.bits = take(M2 * C3 / 8);
        o.c1 = take(16 * P); o.c2 = take(16 * P); o.c3 = take(16 * C3);
        o.cd = o.proj ? take(16 * C3) : 0;
    }
    // ---- backward work
    L->gx0 = take((size_t)B * Hs * Ws * 64 * 2);
    L->part0 = take(16 * 64 * 4);
    L->tmp0 = (Ws % 128) ? take((size_t)B * Hs * Ws * 16 * 4) : 0;     // two-launch stem data gradient (maps narrower than 128)
    L->gtop = take((size_t)B * h * w * cin * 2);
    for (int i = 0; i < d->nblocks; i++) {
        BlkOff& o = L->b[i];
        const size_t M1 = (size_t)B * o.Hin * o.Win, M2 = (size_t)B * o.H2 * o.W2, P = o.P, C3 = 4 * P;
        o.gx3 = take(M2 * C3 * 2);
        o.gy2 = take(M2 * P * 2); o.gx2 = take(M2 * P * 2);
        o.gy1 = take(M1 * P * 2); o.gx1 = take(M1 * P * 2);
        o.gxd = o.proj ? take(M2 * C3 * 2) : 0;
        o.gind = o.proj ? take(M1 * o.Cin * 2) : 0;
        o.gin = take(M1 * o.Cin * 2);
    }
    L->kc_stride = a256((size_t)3 * maxC * 4);
    L->kc = take(4 * L->kc_stride);
    L->wstride = a256(wneed);
    L->wscr = take(3 * L->wstride);
    L->total = off;
    return PPV_OK;
}

// ---- cross-stream forks: a ring of timing-less events per device, guarded (the autograd engine runs one worker thread per device, a
// harness may run its own)
constexpr int RING = 128, MAXDEV = 16;
hipEvent_t g_ring[MAXDEV][RING];
bool g_made[MAXDEV] = {};
unsigned g_next[MAXDEV] = {};
std::mutex g_ring_mu;

int fork_between(hipStream_t from, hipStream_t to) {
    int dev = 0;
    if (hipError_t r = hipGetDevice(&dev)) return -(int)r;
    if (dev < 0 || dev >= MAXDEV) return PPV_ERR_BAD_SIZE;
    hipEvent_t e;
    {
        std::lock_guard<std::mutex> lk(g_ring_mu);
        if (!g_made[dev]) {
            for (auto& ev : g_ring[dev])
                if (hipError_t r = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) return -(int)r;
            g_made[dev] = true;
        }
        e = g_ring[dev][g_next[dev]++ & (RING - 1)];
    }
    if (hipError_t r = hipEventRecord(e, from)) return -(int)r;
    if (hipError_t r = hipStreamWaitEvent(to, e, 0)) return -(int)r;
    return PPV_OK;
}

}  // namespace
namespace ppv {
// next event of a second per-device ring, for launches that signal it from their own dispatch packet (hipExtLaunchKernel's stopEvent)
int fork_stop_event(hipEvent_t* out) {
    static hipEvent_t ring[MAXDEV][RING];
    static bool made[MAXDEV] = {};
    static unsigned next[MAXDEV] = {};
    int dev = 0;
    if (hipError_t r = hipGetDevice(&dev)) return -(int)r;
    if (dev < 0 || dev >= MAXDEV) return PPV_ERR_BAD_SIZE;
    std::lock_guard<std::mutex> lk(g_ring_mu);
    if (!made[dev]) {
        for (auto& ev : ring[dev])
            if (hipError_t r = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) return -(int)r;
        made[dev] = true;
    }
    *out = ring[dev][next[dev]++ & (RING - 1)];
    return PPV_OK;
}
// ---- forks without a packet on the main chain (round 6).  An event record behind a launch costs the main chain ~6 us of idle queue, the
// launch's own stop event ~4.5 us (kernel traces under profiles/r06_*); a fork by FLAG costs it nothing: the weight-gradient stream waits
// (hipStreamWaitValue64, >=) for a value that the NEXT launch of the main chain stores when its first workgroup starts (ConvGeom::start_flag)
// -- at that moment everything enqueued before it on the main chain, the producer of the weight gradient's operand included, is complete
// and visible.  One 64-bit HSA signal word and one counter per (device, main stream): launches of one stream start in order, so the
// values a stream's forks wait for are reached in order; streams do not share a word.
struct ForkFlag { hipStream_t stream; int dev; unsigned long long* word; unsigned long long next; };
static ForkFlag g_flags[64];
static int g_nflags = 0;
static int g_flag_ok[MAXDEV] = {};          // 0 unknown, 1 usable, -1 not
int fork_flag_next(hipStream_t main, unsigned long long** word, unsigned long long* val) {
    int dev = 0;
    if (hipError_t r = hipGetDevice(&dev)) return -(int)r;
    if (dev < 0 || dev >= MAXDEV) return PPV_ERR_BAD_SIZE;
    std::lock_guard<std::mutex> lk(g_ring_mu);
    if (g_flag_ok[dev] == 0) {
        int can = 0;
        g_flag_ok[dev] = (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && can) ? 1 : -1;
    }
    if (g_flag_ok[dev] < 0) return PPV_ERR_INIT;
    for (int i = 0; i < g_nflags; i++)
        if (g_flags[i].stream == main && g_flags[i].dev == dev) { *word = g_flags[i].word; *val = ++g_flags[i].next; return PPV_OK; }
    if (g_nflags == 64) return PPV_ERR_INIT;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess || !p) { g_flag_ok[dev] = -1; return PPV_ERR_INIT; }
    if (hipMemset(p, 0, 8) != hipSuccess) { g_flag_ok[dev] = -1; return PPV_ERR_INIT; }
    g_flags[g_nflags] = ForkFlag{main, dev, (unsigned long long*)p, 0};
    *word = (unsigned long long*)p;
    *val = ++g_flags[g_nflags].next;
    ++g_nflags;
    return PPV_OK;
}
// `side` waits until the word reaches val
int fork_flag_wait(hipStream_t side, unsigned long long* word, unsigned long long val) {
    if (hipError_t r = hipStreamWaitValue64(side, word, val, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull)) return -(int)r;
    return PPV_OK;
}
// the launch that was to store the value did not happen: the main chain stores it itself (a packet on `main`, in stream order)
int fork_flag_settle(hipStream_t main, unsigned long long* word, unsigned long long val) {
    if (hipError_t r = hipStreamWriteValue64(main, word, val, 0)) return -(int)r;
    return PPV_OK;
}
}  // namespace ppv
namespace {
inline const PpvTrunkConv& conv_of(const PpvTrunkConv* cv, int blk, int which) { return cv[1 + 4 * blk + which]; }

#define TRY(x) do { if (int e_ = (x)) return e_; } while (0)

}  // namespace

extern "C" {

int ppv_stream_fork(hipStream_t from, hipStream_t to) { return fork_between(from, to); }

// A stream whose kernels run on CUs [first_cu, first_cu + n_cus) of the 256-bit CU mask only (hipExtStreamCreateWithCUMask).  On a
// multi-XCD part the driver deals the mask's bits out round-robin over the XCDs, so a contiguous range is spread evenly over all eight.
// Used to give the weight-gradient stream and the data-gradient stream of backward their own CUs instead of letting them take turns.
int ppv_stream_create_masked(hipStream_t* out, int first_cu, int n_cus) {
    if (!out) return PPV_ERR_NULL;
    if (first_cu < 0 || n_cus < 1 || first_cu + n_cus > 256) return PPV_ERR_BAD_SIZE;
    uint32_t mask[8] = {};
    for (int i = first_cu; i < first_cu + n_cus; i++) mask[i >> 5] |= 1u << (i & 31);
    if (hipError_t r = hipExtStreamCreateWithCUMask(out, 8, mask)) return -(int)r;
    return PPV_OK;
}
int ppv_stream_destroy(hipStream_t s) {
    if (hipError_t r = hipStreamDestroy(s)) return -(int)r;
    return PPV_OK;
}

size_t ppv_trunk_arena_bytes(const PpvTrunkDesc* d) {
    Layout L;
    return make_layout(d, &L) ? 0 : L.total;
}

// byte offsets (from the arena base) of one block's tensors, for tests and debuggers: out[20] =
// x1 y1 x2 y2 x3 xd yout bits c1 c2 c3 cd gx3 gy2 gx2 gy1 gx1 gxd gind gin; blk = -1: the stem (raw0 c0 y0 arg0 gx0 gtop, rest 0)
int ppv_trunk_block_offsets(const PpvTrunkDesc* d, int blk, size_t* out) {
    Layout L;
    TRY(make_layout(d, &L));
    if (!out || blk < -1 || blk >= d->nblocks) return PPV_ERR_BAD_SIZE;
    for (int i = 0; i < 20; i++) out[i] = 0;
    if (blk < 0) {
        out[0] = L.raw0; out[1] = L.c0; out[2] = L.y0; out[3] = L.arg0; out[4] = L.gx0; out[5] = L.gtop;
        return PPV_OK;
    }
    const BlkOff& o = L.b[blk];
    const size_t v[20] = {o.x1, o.y1, o.x2, o.y2, o.x3, o.xd, o.yout, o.bits, o.c1, o.c2, o.c3, o.cd, o.gx3, o.gy2, o.gx2, o.gy1, o.gx1, o.gxd, o.gind, o.gin};
    for (int i = 0; i < 20; i++) out[i] = v[i];
    return PPV_OK;
}

// Forward.  cv[0] = stem, cv[1 + 4 b + {0, 1, 2, 3}] = conv1 / conv2 / conv3 / shortcut of block b (forward weight layouts `wt`,
// BatchNorm weight / bias / running statistics; running statistics may be null); hy: momentum and eps of the same BatchNorms.
// images [B,3,H,W] f32 NCHW.  cells_out [B,H/32,W/32,C] bf16: the last block's output (the map behind models.py:39-41's pooled tensor).
int ppv_trunk_fwd(const PpvTrunkDesc* d, const PpvTrunkConv* cv, const PpvTrunkHyper* hy, const float* images, void* arena,
                  void* cells_out, const void* zero_page, hipStream_t s) {
    if (!d || !cv || !hy || !images || !arena || !cells_out || !zero_page) return PPV_ERR_NULL;
    Layout L;
    TRY(make_layout(d, &L));
    char* A = (char*)arena;
    const int B = d->B;
    if (hipError_t r = hipMemsetAsync(A, 0, L.zero_bytes, s)) return -(int)r;
    // ---- stem: conv 7x7/2 -> BatchNorm coefficients -> BN + ReLU + max-pool 3x3/2 (resnet.0-3)
    const int Hs = d->H / 2, Ws = d->W / 2;
    TRY(ppv_stem_conv(images, cv[0].wt, A + L.raw0, (float*)(A + L.st0), L.T0, B, d->H, d->W, s));
    TRY(ppv_bn_finalize((const float*)(A + L.st0), L.T0, (double)B * Hs * Ws, cv[0].gamma, cv[0].beta, cv[0].rm, cv[0].rv, hy[0].mom, hy[0].eps,
                        (float*)(A + L.c0), 64, s));
    TRY(ppv_bn_relu_maxpool(A + L.raw0, (const float*)(A + L.c0), A + L.y0, A + L.arg0, B, Hs, Ws, 64, s));
    const void* x = A + L.y0;
    for (int i = 0; i < d->nblocks; i++) {
        const BlkOff& o = L.b[i];
        const PpvTrunkConv &k1 = conv_of(cv, i, 0), &k2 = conv_of(cv, i, 1), &k3 = conv_of(cv, i, 2), &kd = conv_of(cv, i, 3);
        const PpvTrunkHyper *h1 = hy + 1 + 4 * i, *h2 = h1 + 1, *h3 = h1 + 2, *hd = h1 + 3;
        void* yout = (i == d->nblocks - 1) ? cells_out : (void*)(A + o.yout);
        const int P = o.P, C3 = 4 * P;
        const long M1 = (long)B * o.Hin * o.Win, M2 = (long)B * o.H2 * o.W2;
        if (!o.proj) {
            PpvBottleneckFwd a;
            a.xin = x; a.w1 = k1.wt; a.w2 = k2.wt; a.w3 = k3.wt;
            a.x1 = A + o.x1; a.y1 = A + o.y1; a.x2 = A + o.x2; a.y2 = A + o.y2; a.x3 = A + o.x3; a.yout = yout; a.bits = A + o.bits;
            a.stats1 = (float*)(A + o.st1); a.stats2 = (float*)(A + o.st2); a.stats3 = (float*)(A + o.st3);
            a.coef1 = (float*)(A + o.c1); a.coef2 = (float*)(A + o.c2); a.coef3 = (float*)(A + o.c3);
            a.g1 = k1.gamma; a.b1 = k1.beta; a.rm1 = k1.rm; a.rv1 = k1.rv;
            a.g2 = k2.gamma; a.b2 = k2.beta; a.rm2 = k2.rm; a.rv2 = k2.rv;
            a.g3 = k3.gamma; a.b3 = k3.beta; a.rm3 = k3.rm; a.rv3 = k3.rv;
            a.zero_page = zero_page;
            a.mom1 = h1->mom; a.eps1 = h1->eps; a.mom2 = h2->mom; a.eps2 = h2->eps; a.mom3 = h3->mom; a.eps3 = h3->eps;
            a.B = B; a.H = o.Hin; a.W = o.Win; a.Cin = o.Cin; a.planes = P; a.stride = 1; a.T1 = o.T12a; a.T2 = o.T12b; a.T3 = o.T3;
            TRY(ppv_bottleneck_fwd(&a, s));
        } else {
            // projection block: conv1 / conv2 as above (statistics folded by the apply kernels); bn3 and the shortcut's BatchNorm keep the
            // coefficient launch (one apply kernel normalises both tensors)
            TRY(ppv_conv_gemm(x, k1.wt, A + o.x1, (float*)(A + o.st1), nullptr, nullptr, zero_page, B, o.Hin, o.Win, o.Cin, o.Hin, o.Win, P, 1, 1, 1, 0, 1, 0, o.T12a, s));
            TRY(ppv_bn_act_fold_rows(A + o.x1, (const float*)(A + o.st1), o.T12a, (double)M1, k1.gamma, k1.beta, k1.rm, k1.rv, h1->mom, h1->eps,
                                     (float*)(A + o.c1), nullptr, A + o.y1, nullptr, M1 * P, P, 0, 1, s));
            TRY(ppv_conv_gemm(A + o.y1, k2.wt, A + o.x2, (float*)(A + o.st2), nullptr, nullptr, zero_page, B, o.Hin, o.Win, P, o.H2, o.W2, P, 3, 3, o.st, -1, 1, 0, o.T12b, s));
            TRY(ppv_bn_act_fold_rows(A + o.x2, (const float*)(A + o.st2), o.T12b, (double)M2, k2.gamma, k2.beta, k2.rm, k2.rv, h2->mom, h2->eps,
                                     (float*)(A + o.c2), nullptr, A + o.y2, nullptr, M2 * P, P, 0, 1, s));
            TRY(ppv_conv_gemm(A + o.y2, k3.wt, A + o.x3, (float*)(A + o.st3), nullptr, nullptr, zero_page, B, o.H2, o.W2, P, o.H2, o.W2, C3, 1, 1, 1, 0, 1, 0, o.T3, s));
            TRY(ppv_bn_finalize((const float*)(A + o.st3), o.T3, (double)M2, k3.gamma, k3.beta, k3.rm, k3.rv, h3->mom, h3->eps, (float*)(A + o.c3), C3, s));
            TRY(ppv_conv_gemm(x, kd.wt, A + o.xd, (float*)(A + o.std_), nullptr, nullptr, zero_page, B, o.Hin, o.Win, o.Cin, o.H2, o.W2, C3, 1, 1, o.st, 0, 1, 0, o.Td, s));
            TRY(ppv_bn_finalize((const float*)(A + o.std_), o.Td, (double)M2, kd.gamma, kd.beta, kd.rm, kd.rv, hd->mom, hd->eps, (float*)(A + o.cd), C3, s));
            TRY(ppv_bn_act(A + o.x3, (const float*)(A + o.c3), A + o.xd, (const float*)(A + o.cd), yout, A + o.bits, M2 * C3, C3, 2, 1, 0, s));
        }
        x = yout;
    }
    return PPV_OK;
}

// Backward of blocks [blk_lo, blk_hi) in reverse order (data-parallel runs call it once per gradient bucket; single-process runs once
// with (0, nblocks)).  With blk_hi == nblocks the gradient of the output arrives as g_top: g_kind 0 = bf16 [B,h,w,C] ALREADY masked by
// the last block's ReLU; 1 / 2 = f32 / bf16 [B,E,E,C] gradient of the adaptive-average-pooled tensor (models.py:39: the pool's backward
// and the mask run here).  With blk_lo == 0 the stem follows (g_img [B,3,H,W] f32 or null = the image needs no gradient) and `main`
// waits for `side`.  Weight gradients (cv[].dw non-null) run on `side` (null: on `main`) behind an event recorded on `main`.
// cells = the cells_out of the forward call.
int ppv_trunk_bwd(const PpvTrunkDesc* d, const PpvTrunkConv* cv, void* arena, const void* cells, const void* g_top, int g_kind, int E,
                  float* g_img, const void* zero_page, int blk_lo, int blk_hi, hipStream_t main, hipStream_t side) {
    if (!d || !cv || !arena || !cells || !zero_page) return PPV_ERR_NULL;
    Layout L;
    TRY(make_layout(d, &L));
    if (blk_lo < 0 || blk_hi > d->nblocks || blk_lo >= blk_hi) return PPV_ERR_BAD_SIZE;
    char* A = (char*)arena;
    const int B = d->B, nb = d->nblocks;
    hipStream_t ws = side ? side : main;
    float* kc[4];
    for (int i = 0; i < 4; i++) kc[i] = (float*)(A + L.kc + i * L.kc_stride);
    char* wscr = A + L.wscr;
    const long wstride = d->wgrad_reduce3 ? (long)L.wstride : 0;
    if (blk_hi == nb) {
        if (!g_top) return PPV_ERR_NULL;
        if (hipError_t r = hipMemsetAsync(A + L.bzero_off, 0, L.bzero_bytes, main)) return -(int)r;
        const BlkOff& o = L.b[nb - 1];
        if (g_kind != 0)
            TRY(ppv_adaptive_pool_bwd(g_top, A + L.gtop, cells, B, o.H2, o.W2, 4 * o.P, E, g_kind == 1, main));
    }
    // PPV_WGRAD_PAIR=1: conv1's weight gradient of an identity bottleneck waits for the next bottleneck's conv3 and the two run as one launch
    // (ppv_conv_wgrad_pair); flushed before every projection block and at the end of this call (a gradient bucket is complete when it returns)
    static const int pair_on = getenv("PPV_WGRAD_PAIR") ? atoi(getenv("PPV_WGRAD_PAIR")) : 0;
    // PPV_WGRAD_FORKS=3: ONE fork per identity bottleneck (behind bn3'): conv2's and conv1's weight gradients of the bottleneck that ran before
    // ride on it (block_exec.hip); same flush points as the pairing
    static const int forks3 = getenv("PPV_WGRAD_FORKS") ? (atoi(getenv("PPV_WGRAD_FORKS")) == 3) : 0;
    const bool hold = (pair_on || forks3) && !d->wgrad_reduce3;
    if (hold) ppv::wgrad_pair_begin(pair_on ? L.wstride : 0, forks3 && !pair_on);
    struct PairGuard {
        hipStream_t m, w; bool on; int rc = PPV_OK; bool done = false;
        int finish() { if (on && !done) { done = true; rc = ppv::wgrad_pair_end(m, w); } return rc; }
        ~PairGuard() { (void)finish(); }
    } pair_guard{main, ws, hold};
    for (int i = blk_hi - 1; i >= blk_lo; i--) {
        const BlkOff& o = L.b[i];
        const PpvTrunkConv &k1 = conv_of(cv, i, 0), &k2 = conv_of(cv, i, 1), &k3 = conv_of(cv, i, 2), &kd = conv_of(cv, i, 3);
        const int P = o.P, C3 = 4 * P;
        const long M1 = (long)B * o.Hin * o.Win, M2 = (long)B * o.H2 * o.W2;
        if (o.proj && pair_guard.on) TRY(ppv::wgrad_pair_flush(main, ws, true));
        // gradient w.r.t. this block's output (masked by its ReLU where it was produced) and whether bn3's sums came with it
        const void* g = (i == nb - 1) ? ((g_kind == 0) ? g_top : (const void*)(A + L.gtop)) : (const void*)(A + L.b[i + 1].gin);
        const int part3_ready = (i < nb - 1) && red_ok(M2, C3);
        // the block this block's input gradient flows into: its raw conv3 output and sums buffer ride in the conv1 data-gradient launch
        const bool feed_prev = i > 0 && red_ok(M1, o.Cin);
        const void* xin = i > 0 ? (const void*)(A + L.b[i - 1].yout) : (const void*)(A + L.y0);
        const void* xin_bits = i > 0 ? (const void*)(A + L.b[i - 1].bits) : nullptr;
        const bool red2 = red_ok(M2, P), red1 = red_ok(M1, P);
        if (!o.proj) {
            PpvBottleneckBwd a;
            a.g = g; a.xin = xin; a.x1 = A + o.x1; a.y1 = A + o.y1; a.x2 = A + o.x2; a.y2 = A + o.y2; a.x3 = A + o.x3; a.xin_bits = xin_bits;
            a.c1 = (const float*)(A + o.c1); a.c2 = (const float*)(A + o.c2); a.c3 = (const float*)(A + o.c3);
            a.wd1 = k1.wd; a.wd2 = k2.wd; a.wd3 = k3.wd;
            a.part3 = (float*)(A + o.p3); a.part2 = (float*)(A + o.p2); a.part1 = (float*)(A + o.p1);
            a.kc3 = kc[0]; a.kc2 = kc[1]; a.kc1 = kc[2];
            a.gx3 = A + o.gx3; a.gy2 = A + o.gy2; a.gx2 = A + o.gx2; a.gy1 = A + o.gy1; a.gx1 = A + o.gx1; a.gin = A + o.gin;
            a.dg3 = k3.dgamma; a.db3 = k3.dbeta; a.dg2 = k2.dgamma; a.db2 = k2.dbeta; a.dg1 = k1.dgamma; a.db1 = k1.dbeta;
            a.dw3 = k3.dw; a.dw2 = k2.dw; a.dw1 = k1.dw;
            a.wscratch = wscr;
            a.x3_prev = feed_prev ? (const void*)(A + L.b[i - 1].x3) : nullptr;
            a.part3_prev = feed_prev ? (float*)(A + L.b[i - 1].p3) : nullptr;
            a.zero_page = zero_page;
            a.B = B; a.H = o.Hin; a.W = o.Win; a.planes = P; a.part3_ready = part3_ready; a.red2 = red2; a.red1 = red1;
            a.wstride = wstride;
            TRY(ppv_bottleneck_bwd(&a, main, side));
            continue;
        }
        // ---- projection block (the per-kernel order of encoder.py: each weight gradient as soon as its operand exists)
        // forks behind a BatchNorm-backward launch wait for that launch's own stop event where PPV_FORK_STOPEV (default 1) allows it
        static const int stopev_on = getenv("PPV_FORK_STOPEV") ? atoi(getenv("PPV_FORK_STOPEV")) : 1;
        bool use_stop = stopev_on && side && side != main;
        if (use_stop) {                               // (not inside a stream capture: see block_exec.hip)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(main, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) use_stop = false;
        }
        hipEvent_t sev = nullptr;
        auto arm = [&](bool wanted) -> int {
            sev = nullptr;
            if (!use_stop || !wanted) return PPV_OK;
            if (int r = ppv::fork_stop_event(&sev)) return r;
            ppv::bn_bwd_stop_event_once(sev);
            return PPV_OK;
        };
        auto fork_now = [&]() -> int {
            if (sev && ppv::bn_bwd_stop_event_unused()) sev = nullptr;
            if (!side) return PPV_OK;
            if (sev) { if (hipError_t r = hipStreamWaitEvent(side, sev, 0)) return -(int)r; return PPV_OK; }
            return fork_between(main, side);
        };
        // bn3 backward; the shortcut's BatchNorm sees the same gradient: its sums ride along
        TRY(arm(k3.dw != nullptr));
        TRY(ppv_bn_bwd_sums2(g, A + o.x3, (const float*)(A + o.c3), (double)M2, A + o.gx3, k3.dgamma, k3.dbeta, (float*)(A + o.p3), kc[0], M2, C3,
                             part3_ready ? 2 : 1, A + o.xd, (float*)(A + o.pd), main));
        if (k3.dw) {
            TRY(fork_now());
            TRY(ppv_conv_wgrad(A + o.gx3, A + o.y2, k3.dw, wscr, zero_page, B, o.H2, o.W2, P, o.H2, o.W2, C3, 1, 1, 1, 0, ws));
        }
        if (red2) {
            TRY(ppv_conv_gemm_red(A + o.gx3, k3.wd, A + o.gy2, (float*)(A + o.p2), A + o.x2, (const float*)(A + o.c2), nullptr, nullptr, zero_page,
                                  B, o.H2, o.W2, C3, o.H2, o.W2, P, 1, 1, 1, 0, 1, 8, main));
            TRY(arm(k2.dw != nullptr));
            TRY(ppv_bn_bwd(A + o.gy2, nullptr, A + o.x2, (const float*)(A + o.c2), (double)M2, A + o.gx2, nullptr, k2.dgamma, k2.dbeta, (float*)(A + o.p2), kc[1], M2, P, 0, 2, main));
        } else {
            TRY(ppv_conv_gemm(A + o.gx3, k3.wd, A + o.gy2, nullptr, nullptr, nullptr, zero_page, B, o.H2, o.W2, C3, o.H2, o.W2, P, 1, 1, 1, 0, 1, 0, 0, main));
            TRY(arm(k2.dw != nullptr));
            TRY(ppv_bn_bwd(A + o.gy2, nullptr, A + o.x2, (const float*)(A + o.c2), (double)M2, A + o.gx2, nullptr, k2.dgamma, k2.dbeta, (float*)(A + o.p2), kc[1], M2, P, 2, 1, main));
        }
        if (k2.dw) {
            TRY(fork_now());
            TRY(ppv_conv_wgrad(A + o.gx2, A + o.y1, k2.dw, wscr, zero_page, B, o.Hin, o.Win, P, o.H2, o.W2, P, 3, 3, o.st, 1, ws));
        }
        if (red1) {
            TRY(ppv_conv_gemm_red(A + o.gx2, k2.wd, A + o.gy1, (float*)(A + o.p1), A + o.x1, (const float*)(A + o.c1), nullptr, nullptr, zero_page,
                                  B, o.H2, o.W2, P, o.Hin, o.Win, P, 3, 3, 1, -1, o.st, 8, main));
            TRY(arm(k1.dw != nullptr));
            TRY(ppv_bn_bwd(A + o.gy1, nullptr, A + o.x1, (const float*)(A + o.c1), (double)M1, A + o.gx1, nullptr, k1.dgamma, k1.dbeta, (float*)(A + o.p1), kc[2], M1, P, 0, 2, main));
        } else {
            TRY(ppv_conv_gemm(A + o.gx2, k2.wd, A + o.gy1, nullptr, nullptr, nullptr, zero_page, B, o.H2, o.W2, P, o.Hin, o.Win, P, 3, 3, 1, -1, o.st, 0, 0, main));
            TRY(arm(k1.dw != nullptr));
            TRY(ppv_bn_bwd(A + o.gy1, nullptr, A + o.x1, (const float*)(A + o.c1), (double)M1, A + o.gx1, nullptr, k1.dgamma, k1.dbeta, (float*)(A + o.p1), kc[2], M1, P, 2, 1, main));
        }
        if (k1.dw) {
            TRY(fork_now());
            TRY(ppv_conv_wgrad(A + o.gx1, xin, k1.dw, wscr, zero_page, B, o.Hin, o.Win, o.Cin, o.Hin, o.Win, P, 1, 1, 1, 0, ws));
        }
        // shortcut: its BatchNorm (sums already taken), weight gradient, data gradient; then conv1's data gradient adds it, applies the
        // block input's ReLU mask and takes bn3's sums of the block that gradient flows into
        TRY(ppv_bn_bwd(g, nullptr, A + o.xd, (const float*)(A + o.cd), (double)M2, A + o.gxd, nullptr, kd.dgamma, kd.dbeta, (float*)(A + o.pd), kc[3], M2, C3, 0, 2, main));
        if (kd.dw) {
            if (side) TRY(fork_between(main, side));
            TRY(ppv_conv_wgrad(A + o.gxd, xin, kd.dw, wscr, zero_page, B, o.Hin, o.Win, o.Cin, o.H2, o.W2, C3, 1, 1, o.st, 0, ws));
        }
        // stride-2 shortcut: its data gradient is nonzero on the even-even pixels only.  Where conv1's data gradient runs on conv_stream.hip
        // the gradient stays COMPACT (a plain 1x1 product on the g grid, a quarter of the bytes) and the addend is gathered from it;
        // otherwise (layer 1: stride 1; layer 4: tiled kernel) the dense map is written (conv_dgrad_s2.hip) and read back.
        const bool compact = o.st == 2 && ppv::conv_addend_compact_supported(B, o.Hin, o.Win, P, o.Cin);
        if (compact)
            TRY(ppv_conv_gemm(A + o.gxd, kd.wd, A + o.gind, nullptr, nullptr, nullptr, zero_page, B, o.H2, o.W2, C3, o.H2, o.W2, o.Cin, 1, 1, 1, 0, 1, 0, 0, main));
        else
            TRY(ppv_conv_gemm(A + o.gxd, kd.wd, A + o.gind, nullptr, nullptr, nullptr, zero_page, B, o.H2, o.W2, C3, o.Hin, o.Win, o.Cin, 1, 1, 1, 0, o.st, 0, 0, main));
        ppv::conv_set_addend_compact(compact);
        if (feed_prev)
            TRY(ppv_conv_gemm_red(A + o.gx1, k1.wd, A + o.gin, (float*)(A + L.b[i - 1].p3), A + L.b[i - 1].x3, nullptr, A + o.gind, xin_bits, zero_page,
                                  B, o.Hin, o.Win, P, o.Hin, o.Win, o.Cin, 1, 1, 1, 0, 1, 8, main));
        else
            TRY(ppv_conv_gemm(A + o.gx1, k1.wd, A + o.gin, nullptr, A + o.gind, xin_bits, zero_page, B, o.Hin, o.Win, P, o.Hin, o.Win, o.Cin, 1, 1, 1, 0, 1, 0, 0, main));
    }
    TRY(pair_guard.finish());
    if (blk_lo == 0) {
        // ---- stem: max-pool + ReLU + BatchNorm backward in two passes over the pooled tensors, then the 7x7 data gradient
        const int Hs = d->H / 2, Ws = d->W / 2;
        const bool affine = cv[0].dgamma != nullptr;
        if (g_img || affine) {
            TRY(ppv_maxpool_bn_bwd(A + L.b[0].gin, A + L.y0, A + L.arg0, A + L.raw0, (const float*)(A + L.c0), (double)B * Hs * Ws, A + L.gx0,
                                   cv[0].dgamma, cv[0].dbeta, (float*)(A + L.part0), B, Hs, Ws, 64, main));
            if (g_img) {
                if (Ws % 128 == 0) {
                    TRY(ppv_stem_dgrad(A + L.gx0, cv[0].wd, g_img, zero_page, B, Hs, Ws, main));
                } else {
                    TRY(ppv_conv_gemm(A + L.gx0, cv[0].wd, A + L.tmp0, nullptr, nullptr, nullptr, zero_page, B, Hs, Ws, 64, Hs, Ws, 16, 4, 4, 1, -1, 1, 1, 0, main));
                    TRY(ppv_stem_dgrad_scatter((const float*)(A + L.tmp0), g_img, B, Hs, Ws, main));
                }
            }
        }
        if (side) TRY(fork_between(side, main));
    }
    return PPV_OK;
}

}  // extern "C"
