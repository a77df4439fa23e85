// uh_tail.hip -- the whole post-regressor tail of one l1_loss training step as ONE call and ONE hipGraph launch
// (SURVEY section 8 f2):
//   h4p -> Tensor-DLT -> theta -> warp -> gray patch -> L1      and      d L1 / d h4p   (for dLoss = 1)
// i.e. solve_DLT + transform + the l1 branch of build_losses and TF's backward of them
// (/root/reference/code/homography_model.py:169-269, 321-330).  Because the loss is a scalar and the chain is
// evaluated at dLoss = 1, the backward does not have to wait for autograd: the caller scales dh4p by the incoming
// gradient.  The 7 kernels of the un-fused chain (DLT, warp, gather+losses and its finish, loss-gradient+warp backward
// and its finish, DLT backward; 4 kernels with the fused patch kernel) are stream-captured once per distinct argument set and
// replayed with hipGraphLaunch: one host call and one launch per step, and no inter-kernel launch gaps on the stream.
#include "uh_device.h"
#include "uh_host.h"
#include <mutex>
#include <vector>
#include <cstring>

struct uh_tail_plan {
    int B, H, W, C, P;
    unsigned flags;
    // workspace layout (byte offsets, 256-byte aligned)
    size_t off_theta, off_dtheta, off_warped, off_stats, off_ws_warp, off_ws_loss, off_ws_patch, total;
    // A captured graph bakes in every kernel ARGUMENT: the pointers and the M / Minv constants (passed by value to the
    // DLT kernels).  The key therefore holds both; a caller that re-runs a plan with different M_host contents gets a
    // fresh capture, never stale constants.
    struct Entry { const void* key[12]; float mk[18]; hipGraph_t graph; hipGraphExec_t exec; unsigned long long stamp; };
    std::vector<Entry> cache;
    unsigned long long clock = 0;
    long long launches = 0, captures = 0;
    int misses_in_a_row = 0;          // capture thrash detector (see uh_tail_run)
    // argument sets that were enqueued eagerly while the detector was tripped: a set seen a SECOND time is worth a capture
    struct Seen { const void* key[12]; float mk[18]; };
    std::vector<Seen> seen;
    size_t seen_cursor = 0;
    std::mutex mu;
};

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int uh_tail_create(uh_tail_plan** out, int B, int H, int W, int C, int P, unsigned flags) {
    if (!out) return UH_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || P <= 0 || P > H || P > W) return UH_E_SHAPE;
    if (!(flags & UH_TAIL_FUSED_PATCH) && P < 3) return UH_E_SHAPE;       // the loss kernel also forms the 3x3 SSIM monitor
    if (C < 1 || C > 4) return UH_E_CHANNELS;
    if (uh_warp_backward_workspace_bytes(B, H, W, C, H, W) == 0) return UH_E_TOO_LARGE;
    uh_tail_plan* p = new uh_tail_plan();
    p->B = B; p->H = H; p->W = W; p->C = C; p->P = P; p->flags = flags;
    const bool fused = flags & UH_TAIL_FUSED_PATCH;
    const size_t frame = (size_t)B * H * W * C * sizeof(float);
    size_t o = 0;
    p->off_theta = o;   o = align256(o + (size_t)B * 9 * sizeof(float));
    p->off_dtheta = o;  o = align256(o + (size_t)B * 9 * sizeof(float));
    p->off_warped = o;  o = align256(o + (fused ? 0 : frame));
    // (no dWarped frame and no dPred: the backward forms the loss gradient itself -- uh_warp_patch_loss_backward)
    p->off_stats = o;   o = align256(o + (fused ? 0 : 16 * sizeof(float)));
    p->off_ws_warp = o; o = align256(o + (fused ? 0 : uh_warp_patch_backward_workspace_bytes(B, H, W, C)));
    p->off_ws_loss = o; o = align256(o + (fused ? 0 : uh_patch_losses_workspace_bytes(B, P)));
    p->off_ws_patch = o; o = align256(o + (fused ? uh_warp_patch_l1_workspace_bytes(B, P * P) : 0));
    p->total = o;
    *out = p;
    return 0;
}

extern "C" size_t uh_tail_workspace_bytes(const uh_tail_plan* p) { return p ? p->total : 0; }

extern "C" size_t uh_tail_warped_offset(const uh_tail_plan* p) {
    return (p && !(p->flags & UH_TAIL_FUSED_PATCH)) ? p->off_warped : (size_t)-1;
}

extern "C" void uh_tail_destroy(uh_tail_plan* p) {
    if (!p) return;
    for (auto& e : p->cache) { (void)hipGraphExecDestroy(e.exec); (void)hipGraphDestroy(e.graph); }
    delete p;
}

extern "C" int uh_tail_stats(const uh_tail_plan* p, long long* launches, long long* captures) {
    if (!p) return UH_E_NULL;
    if (launches) *launches = p->launches;
    if (captures) *captures = p->captures;
    return 0;
}

static int enqueue_chain(const uh_tail_plan* p, const float* pts1, const float* h4p, const float* U, const float* I2,
                         const int* idx, const float* M, const float* Minv, float* Hm, float* pred, float* loss,
                         float* dh4p, unsigned char* ws, hipStream_t s) {
    const int B = p->B, H = p->H, W = p->W, C = p->C, P = p->P, PP = P * P;
    const unsigned dflags = p->flags & (UH_DLT_SOLVE_F64 | UH_DLT_ZERO_NONFINITE_GRAD);
    float* theta = (float*)(ws + p->off_theta);
    float* dtheta = (float*)(ws + p->off_dtheta);
    int e;
    if ((e = uh_dlt_forward(pts1, h4p, Hm, theta, M, Minv, B, dflags, s))) return e;
    if (p->flags & UH_TAIL_FUSED_PATCH) {
        if ((e = uh_warp_patch_l1_fwdbwd(U, theta, I2, idx, pred, loss, dh4p ? dtheta : nullptr, ws + p->off_ws_patch,
                                         uh_warp_patch_l1_workspace_bytes(B, PP), B, H, W, C, PP, s))) return e;
    } else {
        float* warped = (float*)(ws + p->off_warped);
        float* stats = (float*)(ws + p->off_stats);
        if ((e = uh_warp_forward(U, theta, warped, nullptr, B, H, W, C, H, W, s))) return e;
        // gray + gather + all loss values in one call (the finish stage also writes `loss`)
        if ((e = uh::gather_patch_losses(warped, idx, I2, nullptr, nullptr, pred, stats, loss, ws + p->off_ws_loss,
                                         uh_patch_losses_workspace_bytes(B, P), B, H, W, C, P, s))) return e;
        if (dh4p) {
            // loss gradient (dLoss = NULL: 1) + tf.gather's scatter + the warp backward in one launch: no dPred, no frame
            if ((e = uh_warp_patch_loss_backward(UH_LOSS_L1, U, theta, pred, I2, stats, nullptr, idx, dtheta, ws + p->off_ws_warp,
                                                 uh_warp_patch_backward_workspace_bytes(B, H, W, C), B, H, W, C, PP, s))) return e;
        }
    }
    if (dh4p) {
        if ((e = uh_dlt_backward(pts1, h4p, Hm, nullptr, dtheta, M, Minv, dh4p, B, dflags, s))) return e;
    }
    return 0;
}

extern "C" int uh_tail_run(uh_tail_plan* p, const float* pts1, const float* h4p, const float* U, const float* I2,
                           const int* patch_idx, const float* M_host, const float* Minv_host, float* H, float* pred,
                           float* loss, float* dh4p, void* workspace, size_t workspace_bytes, uh_stream_t stream) {
    if (!p || !pts1 || !h4p || !U || !I2 || !patch_idx || !M_host || !Minv_host || !H || !pred || !loss) return UH_E_NULL;
    if (!workspace || workspace_bytes < p->total) return UH_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    const bool use_graph = (p->flags & UH_TAIL_GRAPH) && !uh::g_prof_on && s != nullptr;   // capture needs a real stream
    if (!use_graph) return enqueue_chain(p, pts1, h4p, U, I2, patch_idx, M_host, Minv_host, H, pred, loss, dh4p, ws, s);

    std::lock_guard<std::mutex> lk(p->mu);
    const void* key[12] = {pts1, h4p, U, I2, patch_idx, H, pred, loss, dh4p, workspace, s, nullptr};
    float mk[18];
    std::memcpy(mk, M_host, 9 * sizeof(float)); std::memcpy(mk + 9, Minv_host, 9 * sizeof(float));
    for (auto& e : p->cache) {
        if (std::memcmp(e.key, key, sizeof(key)) == 0 && std::memcmp(e.mk, mk, sizeof(mk)) == 0) {
            e.stamp = ++p->clock;
            p->launches++;
            p->misses_in_a_row = 0;
            return (int)hipGraphLaunch(e.exec, s);
        }
    }
    // A caller that hands in freshly allocated batches every step (new addresses each time) would pay a capture +
    // instantiate on the hot path per step -- slower than the seven eager launches it replaces.  After a full cache
    // worth of consecutive misses stop capturing and enqueue eagerly -- but remember the last 64 argument sets, and
    // capture again as soon as one of them comes back (a pool of batches that is cycled, a double-buffered loader):
    // the detector must not turn --graph_tail off for good.
    if (++p->misses_in_a_row > 8) {
        bool again = false;
        for (auto& sn : p->seen)
            if (std::memcmp(sn.key, key, sizeof(key)) == 0 && std::memcmp(sn.mk, mk, sizeof(mk)) == 0) { again = true; break; }
        if (!again) {
            uh_tail_plan::Seen sn;
            std::memcpy(sn.key, key, sizeof(key)); std::memcpy(sn.mk, mk, sizeof(mk));
            if (p->seen.size() < 64) p->seen.push_back(sn);
            else { p->seen[p->seen_cursor] = sn; p->seen_cursor = (p->seen_cursor + 1) % 64; }
            p->launches++;
            return enqueue_chain(p, pts1, h4p, U, I2, patch_idx, M_host, Minv_host, H, pred, loss, dh4p, ws, s);
        }
    }
    // capture the chain on the caller's stream (thread-local mode: other threads' HIP calls are unaffected)
    hipError_t he = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (he != hipSuccess) return (int)he;
    const int ce = enqueue_chain(p, pts1, h4p, U, I2, patch_idx, M_host, Minv_host, H, pred, loss, dh4p, ws, s);
    hipGraph_t graph = nullptr;
    he = hipStreamEndCapture(s, &graph);
    if (ce != 0) { if (graph) (void)hipGraphDestroy(graph); return ce; }
    if (he != hipSuccess) return (int)he;
    hipGraphExec_t exec = nullptr;
    he = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (he != hipSuccess) { (void)hipGraphDestroy(graph); return (int)he; }
    // The cache is as large as the `seen` ring: every set that can be recognised as "came back" can also STAY captured.  (With
    // 8 entries, 9..64 sets cycled in turn were re-captured and evicted on every step -- slower than the eager launches the
    // detector exists to fall back to)  An instance is seven kernel nodes: a few KB.
    if (p->cache.size() >= 64) {                     // evict the least recently used instance
        size_t v = 0;
        for (size_t i = 1; i < p->cache.size(); ++i) if (p->cache[i].stamp < p->cache[v].stamp) v = i;
        (void)hipGraphExecDestroy(p->cache[v].exec); (void)hipGraphDestroy(p->cache[v].graph);
        p->cache.erase(p->cache.begin() + v);
    }
    uh_tail_plan::Entry en;
    std::memcpy(en.key, key, sizeof(key)); std::memcpy(en.mk, mk, sizeof(mk)); en.graph = graph; en.exec = exec; en.stamp = ++p->clock;
    p->cache.push_back(en);
    p->captures++; p->launches++;
    return (int)hipGraphLaunch(exec, s);
}
