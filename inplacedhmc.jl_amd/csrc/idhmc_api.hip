// idhmc_api.hip -- the C ABI (include/idhmc.h): context, device arenas, state transfer, and the
// reference's caller loops (warmup!(TuningNUTS), mcmc!, mcmc_with_warmup!, src/warmup.jl:269-332,
// src/mcmc.jl:94-105) run for all chains of a context.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <cmath>
#include <new>
#include <vector>
#include <algorithm>
#include <chrono>
#include <memory>
#include "idhmc_internal.hpp"
#include "idhmc_xchg.hpp"

namespace idhmc {
int arena_vectors(int max_depth, int model, int L);
int nuts_waves_per_block(int nch, int model, int shared_metric);
int nuts_wide_waves_per_block(int nch, int model);
}
using namespace idhmc;

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(IDHMC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
struct idhmc_ctx;
static int lanes_join(idhmc_ctx *c);
// every entry point starts here; CTXCHK_LANES is for the ones that may leave the dense leapfrog's lanes open (see idhmc_ctx)
#define CTXCHK_LANES(ctx)                                                    \
    do {                                                                     \
        if (!(ctx)) return fail(IDHMC_ERR_BAD_ARG, "null context");          \
        HIPCHK(hipSetDevice((ctx)->device));                                 \
    } while (0)
#define CTXCHK(ctx)                                                          \
    do {                                                                     \
        CTXCHK_LANES(ctx);                                                   \
        if (int rc_lanes_ = lanes_join(ctx)) return rc_lanes_;               \
    } while (0)

struct VmmBlock {
    void *va = nullptr;
    size_t size = 0;
    hipMemGenericAllocationHandle_t handle{};
    bool mapped = false, created = false;
    void release()
    {
        if (mapped) (void)hipMemUnmap(va, size);
        if (created) (void)hipMemRelease(handle);
        if (va) (void)hipMemAddressFree(va, size);
        va = nullptr; mapped = created = false; size = 0;
    }
};
struct idhmc_ctx {
    int device = 0;
    DevState s{};
    idhmc_options opt{};
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::vector<void *> allocs;
    int64_t bytes = 0;
    double *xchg = nullptr;        // library-owned exchange record (IDHMC_XCHG_DOUBLES)
    int32_t *status_out = nullptr; // device scalar
    double *scratch = nullptr;     // [C][L] staging for broadcasts / moments
    idhmc_allreduce_fn hook = nullptr;
    void *hook_user = nullptr;
    double *hook_buf = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    JitModule *jit = nullptr;      // hipRTC module of a custom density
    Comm *comm = nullptr;          // RCCL communicator of the global-eps exchange (idhmc_comm_init)
    // Pulse of the transition kernel: after every launch the device words {total leapfrog steps, abort code} are
    // copied asynchronously into a pinned ring, one slot per launch.  The drivers read it without ever synchronising
    // the stream: (1) the reference aborts the moment a chain's stepsize falls below 1e-10 (src/warmup.jl:291-296) --
    // the drivers keep at most kLag launches in flight and stop at the first slot that carries the code; (2)
    // measurement / choice of the kernel form.
    static constexpr int kRing = 64, kLag = 8, kPulseWords = 2;
    unsigned long long *ring = nullptr;   // pinned host memory, kRing x kPulseWords; word 0 == ~0: not yet written
    uint64_t launches = 0;
    int force_wide = -1;                  // IDHMC_NUTS_WIDE = 0 / 1 forces one form (tests, experiments)
    int fuse = -1;                        // IDHMC_FUSE = 0 / 1: the drivers never / always make several transitions per launch (-1: yes)
    bool fuse_ok = false;                 // workgroups b and b + 8 share an XCD on this device (probed at creation): fused launches are possible
    // IDHMC_GRAD_RECOMPUTE: the single-step leapfrog of a separable density leaves the stored gradient stale; whoever
    // needs the array (get_grad, the stepsize search, the n-step kernel, the optimum stage) re-evaluates first
    bool grad_stale = false;
    double *pool_scratch = nullptr;       // IDHMC_METRIC_POOLED: {acc0, acc1, mean}
    double *pool_table = nullptr;         // [segments][L + 1] partial sums (grown on demand)
    int64_t pool_table_segs = 0;
    double *ebfmi_out = nullptr;          // [C], idhmc_get_ebfmi
    // draws / records that the caller wants on the host leave through two staging buffers: the device packs transition n,
    // the host copies it out (a blocking pageable copy on its own stream) while transition n + 1 computes
    double *stage_q[2] = {nullptr, nullptr};
    idhmc_tree_stats *stage_st[2] = {nullptr, nullptr};
    int32_t stage_kq = 1, stage_kst = 1;  // transitions the staging buffers hold (idhmc_mcmc's launches of several transitions: more than one)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_packed[2] = {nullptr, nullptr};
    // Lanes of the dense single-step leapfrog (configs[3]).  One sweep of its matrix-core kernel is a load phase, a matrix
    // phase and a store phase that every CU goes through at the same time, so HBM idles while the matrix cores work and vice
    // versa.  Chains are independent, so the context cuts them into up to kLanes contiguous ranges of tiles, each
    // swept on its own stream by a kernel of one workgroup per CU: three such kernels are resident per CU, in different phases,
    // and back-to-back sweeps pipeline across the lanes (lane k's sweep n + 1 waits only for lane k's sweep n).  Lane 0 is the
    // context's stream (the runtime spreads a process's streams over four hardware queues; two lanes on one queue run one
    // after the other); the others fork from it at the first such call and join it at the next entry point of any other kind.
    static constexpr int kLanes = 4;
    hipStream_t lane[kLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_ev[kLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr;
    bool lanes_open = false;
    int lanes_distinct = 0;               // lanes (the context's stream included) found on different hardware queues
    double placement_GBps = 0.0;          // place_state: probe rate of the placement kept, candidates tried
    int placement_tries = 0;
    int placement_kind = 0;               // 0 separate allocations, 1 spread-out slab, 2 one mapped physical allocation, 3 separate allocations found by the pair walk
    double placement_single_GBps = 0.0;   // one array alone (the yardstick of "good")
    double placement_ms = 0.0;            // wall time of place_state
    int64_t placement_peak_bytes = 0;     // most device bytes held at one time during the search
    VmmBlock vmm;                         // kind 2: unmapped / released in idhmc_destroy
    int use_lanes = kLanes;               // IDHMC_DENSE_LANES = 0 switches them off, n caps their number (measurements)
};

static int lanes_join(idhmc_ctx *c)
{
    if (!c->lanes_open) return IDHMC_OK;
    c->lanes_open = false;
    for (int k = 1; k < idhmc_ctx::kLanes; ++k) {
        if (!c->lane[k]) continue;
        HIPCHK(hipEventRecord(c->lane_ev[k], c->lane[k]));
        HIPCHK(hipStreamWaitEvent(c->stream, c->lane_ev[k], 0));
    }
    return IDHMC_OK;
}
// number of kernels a single-step sweep of this context is cut into (0: one kernel on the context's stream): ranges of at most
// 256 tiles, so that every kernel puts one workgroup on every CU; kernel j runs on lane j mod kLanes
static int lane_chunks(const idhmc_ctx *c, int n_steps)
{
    if (c->use_lanes < 2 || n_steps != 1 || c->s.model != IDHMC_MODEL_DENSE_MVN || c->s.nch > 2) return 0;
    if (c->stream != c->own_stream) return 0;     // a caller-owned stream is ordered by the caller's events, not by our entry points
    const int64_t ntiles = (c->s.C + 15) / 16;
    const int64_t n = (ntiles + 255) / 256;
    return n < 3 ? 0 : (int)n;
}
// Lanes only overlap when they sit on DIFFERENT hardware queues: the runtime multiplexes a process's streams onto four of them
// (least-used first at creation), and two streams on one queue run one after the other.  Which queue a new stream gets depends
// on every stream the process has made before (a context created after others, a runtime-internal stream ...: round 3's bench
// ran the lanes on three queues, 63 instead of 48 us per sweep), so the lanes are CHOSEN: candidates are created one by one and
// a candidate becomes a lane when an idle 60 us kernel on it runs concurrently with one on every lane chosen so far.
static bool streams_overlap(hipStream_t a, hipStream_t b)
{
    const long long ticks = 6000;                    // 60 us of the 100 MHz wall clock
    double best = 1e30;
    for (int r = 0; r < 2; ++r) {
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
        const auto t0 = std::chrono::steady_clock::now();
        if (launch_spin(ticks, a) != hipSuccess || launch_spin(ticks, b) != hipSuccess) return false;
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    return best < 1.6 * 60.0;                        // one after the other: >= 120 us
}
static int pick_lane_streams(idhmc_ctx *c, int lanes)
{
    constexpr int kCand = 10;
    hipStream_t cand[kCand] = {};
    int ncand = 0, have = 1;                         // lane 0 is the context's stream
    for (int k = 1; k < lanes; ++k) if (c->lane[k]) ++have;
    while (have < lanes && ncand < kCand) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }   // what we have will do
        cand[ncand++] = s;
        bool ok = streams_overlap(c->stream, s);
        for (int k = 1; k < lanes && ok; ++k) if (c->lane[k]) ok = streams_overlap(c->lane[k], s);
        if (!ok) continue;
        for (int k = 1; k < lanes; ++k) if (!c->lane[k]) { c->lane[k] = s; cand[ncand - 1] = nullptr; ++have; break; }
    }
    // nothing overlaps with anything (a profiler that serialises the streams): take the candidates as they come, as round 2 did
    for (int k = 1, j = 0; k < lanes; ++k) {
        if (c->lane[k]) continue;
        while (j < ncand && !cand[j]) ++j;
        if (j < ncand) { c->lane[k] = cand[j]; cand[j] = nullptr; }
        else HIPCHK(hipStreamCreateWithFlags(&c->lane[k], hipStreamNonBlocking));
    }
    c->lanes_distinct = have;
    for (int j = 0; j < ncand; ++j) if (cand[j]) (void)hipStreamDestroy(cand[j]);
    for (int k = 1; k < lanes; ++k)
        if (!c->lane_ev[k]) HIPCHK(hipEventCreateWithFlags(&c->lane_ev[k], hipEventDisableTiming));
    return IDHMC_OK;
}
static int leapfrog_lanes(idhmc_ctx *c, double eps, int own, int chunks)
{
    const int lanes = chunks < c->use_lanes ? chunks : c->use_lanes;
    if (!c->fork_ev) HIPCHK(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
    {
        bool missing = false;
        for (int k = 1; k < lanes; ++k) missing |= !c->lane[k];
        if (missing) { if (int rc = pick_lane_streams(c, lanes)) return rc; }
    }
    if (!c->lanes_open) {
        HIPCHK(hipEventRecord(c->fork_ev, c->stream));
        for (int k = 1; k < lanes; ++k) HIPCHK(hipStreamWaitEvent(c->lane[k], c->fork_ev, 0));
        c->lanes_open = true;
    }
    const int64_t align = dense_mfma_tile_align(c->s), units = ((c->s.C + 15) / 16 + align - 1) / align, ntiles = (c->s.C + 15) / 16;
    for (int j = 0; j < chunks; ++j) {
        const int64_t t0 = align * (units * j / chunks), t1 = align * (units * (j + 1) / chunks);
        HIPCHK(launch_leapfrog_dense_mfma_tiles(c->s, eps, own, 1, t0, t1 < ntiles ? t1 : ntiles, 256,
                                                (j % lanes) ? c->lane[j % lanes] : c->stream));
    }
    return IDHMC_OK;
}
// one fused leapfrog launch (or one per lane) for the entry points below
static int leapfrog_any(idhmc_ctx *c, double eps, int own, int n_steps, int regrad)
{
    const char *e = getenv("IDHMC_DENSE_MFMA");        // read per call like launch_leapfrog_dense does (tests switch it)
    const bool mfma_off = e && e[0] == '0';
    const int chunks = mfma_off ? 0 : lane_chunks(c, n_steps);
    if (chunks) return leapfrog_lanes(c, eps, own, chunks);
    if (int rc = lanes_join(c)) return rc;
    HIPCHK(launch_leapfrog(c->s, eps, own, n_steps, regrad, c->stream));
    return IDHMC_OK;
}

template <class T>
static int dalloc(idhmc_ctx *c, T **out, int64_t n, bool zero = true)
{
    void *p = nullptr;
    const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(IDHMC_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    if (zero) {
        e = hipMemsetAsync(p, 0, bytes, c->stream);
        if (e != hipSuccess) return fail(IDHMC_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    }
    c->allocs.push_back(p);
    c->bytes += (int64_t)bytes;
    *out = (T *)p;
    return IDHMC_OK;
}
// The state arrays q, p, grad l (and a per-chain M^-1): what the single-step leapfrog streams, element i of each at the same time.
// Where the allocator puts them decides how their streams fall on the HBM channels: the same kernel runs at 5.65 to 6.38 TB/s
// depending on nothing else (profiles/r02_state_layout.log; alternating between successive contexts of one process).  So large
// contexts try a few placements -- every candidate set stays allocated while the next one is made, which is what moves it --
// time the access pattern on each (k_placement_probe, ~0.5 ms per launch at configs[1]) and keep the fastest.
// What "good" means is measured in the same call, not assumed: one array alone streams at the same rate in good and bad
// placements (5.0-5.1 TB/s on MI355X), a good set of nvec arrays together 13-17 % above that, a bad one 0-3 % -- a candidate
// is taken at once when it reaches kGoodRatio x the single-array rate; otherwise the best of the candidates tried wins.
// Bounds (round 3): every exit path frees what it does not keep (CandidateSets below); the bytes held at any one time stay below
// IDHMC_PLACEMENT_MAX_BYTES (default 16 GiB) and a quarter of the free memory, the kept set included; at most
// IDHMC_PLACEMENT_TRIES candidates (default 32, at most 48, 1 = take what comes).  The wall time and the peak are reported by
// idhmc_placement_cost.  IDHMC_PLACEMENT_VERBOSE=1 prints the candidates.
// Candidate kinds, in order: (V) only with IDHMC_PLACEMENT_VMM=1: one physical allocation made with the virtual-memory API (hipMemCreate +
// hipMemMap), the arrays 36 KiB (mod 64 KiB) askew inside it -- measured like any other: NOT deterministic-good (0.97-1.00 x one array alone in a
// bad region, profiles/r03_state_layout.log), and a 2 GiB mapping faulted the GPU in the probe, so it is not tried by default; (S) only with IDHMC_PLACEMENT_SLAB=1 and arrays of >= 256 MiB: one hipMalloc with the starts
// 2050 MiB apart -- it streamed at the full rate in round 2's bad regions (profiles/r02_state_layout.log), did not in round 3's (6 of 6
// bad, profiles/r03_state_layout.log) and keeps (nvec - 1) x (2050 MiB - bytes) unused, so it is no longer tried by default;
// then ordinary sets of nvec hipMallocs.
static bool vmm_make(VmmBlock &v, int device, size_t want)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) { (void)hipGetLastError(); return false; }
    v.size = (want + gran - 1) / gran * gran;
    if (hipMemAddressReserve(&v.va, v.size, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); v.va = nullptr; return false; }
    if (hipMemCreate(&v.handle, v.size, &prop, 0) != hipSuccess) { (void)hipGetLastError(); v.release(); return false; }
    v.created = true;
    if (hipMemMap(v.va, v.size, 0, v.handle, 0) != hipSuccess) { (void)hipGetLastError(); v.release(); return false; }
    v.mapped = true;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(v.va, v.size, &acc, 1) != hipSuccess) { (void)hipGetLastError(); v.release(); return false; }
    return true;
}
namespace {
constexpr int kMaxTries = 48;
constexpr double kGoodRatio = 1.10;
// the candidate sets of one place_state call; whatever is still here when the call returns -- on any path -- is freed
struct CandidateSets {
    double *arr[kMaxTries][4] = {};
    char *slab[kMaxTries] = {};          // set t is one hipMalloc (kind S)
    VmmBlock vmm[kMaxTries];             // set t is one mapped physical allocation (kind V)
    float ms[kMaxTries] = {};
    int64_t held[kMaxTries] = {};        // device bytes the set occupies
    int n = 0;
    int64_t held_now = 0, held_peak = 0;
    void drop(int t, int nvec)
    {
        if (vmm[t].va) vmm[t].release();
        else if (slab[t]) (void)hipFree(slab[t]);
        else for (int k = 0; k < nvec; ++k) if (arr[t][k]) (void)hipFree(arr[t][k]);
        slab[t] = nullptr;
        for (int k = 0; k < 4; ++k) arr[t][k] = nullptr;
        held_now -= held[t];
        held[t] = 0;
    }
    int nvec_ = 0;
    ~CandidateSets() { for (int t = 0; t < n; ++t) if (held[t]) drop(t, nvec_); }
};
}  // namespace
static int probe_ms(idhmc_ctx *c, double *const *v, int nvec, int64_t C, int L, float *ms)
{
    HIPCHK(launch_placement_probe(v, nvec, C, L, c->stream));           // warm-up (TLB, clocks)
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < 4; ++r) HIPCHK(launch_placement_probe(v, nvec, C, L, c->stream));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return IDHMC_OK;
}
static int place_state(idhmc_ctx *c, double **out, int nvec, int64_t n, int64_t C, int L)
{
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t bytes = (size_t)n * sizeof(double);
    const int64_t set_bytes = (int64_t)nvec * (int64_t)bytes;
    int tries = 32;
    if (const char *e = getenv("IDHMC_PLACEMENT_TRIES")) tries = atoi(e);
    const bool verbose = getenv("IDHMC_PLACEMENT_VERBOSE") != nullptr;
    // a candidate set that is not the best so far is cut down to ONE of its arrays (a spacer) -- IDHMC_PLACEMENT_SPACERS=0 holds whole sets as
    // rounds 2-3 did.  Measured after allocator churn (profiles/r03_state_layout.log): 6 of 6 contexts found a good placement (3 to 16
    // candidates, <= 10 GiB held) where whole sets ran out of the 16 GiB budget after 10 candidates in 2 of 6
    const char *sp_env = getenv("IDHMC_PLACEMENT_SPACERS");
    const bool spacers = !(sp_env && sp_env[0] == '0');
    size_t min_bytes = (size_t)64 << 20;
    if (const char *e = getenv("IDHMC_PLACEMENT_MIN_BYTES")) min_bytes = (size_t)atoll(e);      // (experiments)
    if (bytes < min_bytes) tries = 1;                 // small arrays: latency, not channels
    if (tries > kMaxTries) tries = kMaxTries;
    if (tries < 1) tries = 1;
    int64_t budget = (int64_t)16 << 30;               // bytes held at any one time, the kept set included
    if (const char *e = getenv("IDHMC_PLACEMENT_MAX_BYTES")) budget = atoll(e);
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (int64_t)(free_b / 4) < budget) budget = (int64_t)(free_b / 4);
    }
    if (budget < set_bytes) { budget = set_bytes; tries = 1; }
    auto CS = std::unique_ptr<CandidateSets>(new (std::nothrow) CandidateSets());
    if (!CS) return fail(IDHMC_ERR_ALLOC, "out of host memory");
    CandidateSets &cs = *CS;
    cs.nvec_ = nvec;
    const size_t far_stride = (size_t)2050 << 20;
    const int64_t slab_bytes = (int64_t)((nvec - 1) * far_stride + bytes);
    // (V) is opt-in since it cost the round's bench run a GPU memory fault: the context of configs[2] (four arrays, one mapping of 2 GiB + 2 MiB)
    // faulted in its first probe on two boxes out of three, and the candidate is no better placed than three hipMallocs anyway
    // (profiles/r03_state_layout.log); never more than 2 GiB in one mapping
    bool want_vmm = tries > 1 && getenv("IDHMC_PLACEMENT_VMM") && set_bytes + (int64_t)(nvec * ((size_t)36 << 10)) < ((int64_t)2 << 30);
    bool want_slab = tries > 1 && bytes >= ((size_t)256 << 20) && bytes <= ((size_t)2048 << 20) && getenv("IDHMC_PLACEMENT_SLAB");
    const size_t askew = (size_t)36 << 10;
    double single_Bps = 0.0;                          // one array alone, measured on the first candidate
    int best = -1;
    const double probe_bytes = 2.0 * (double)set_bytes * 4;
    // (P) first of all, the PAIR WALK.  Round 3, last measurements (profiles/r03_state_layout.log, tools/ubench/placement_*.hip): whether
    // arrays stream well together is not a matter of their offsets (no offset inside one allocation changes anything) nor of how their
    // physical chunks are ordered (an array of mapped chunks pairs the same in any order): arrays fall into two CLASSES by where in
    // HBM their memory lies, two arrays of different classes stream at 1.12-1.17 x one array alone, two of the same class at 0.97-1.02 x,
    // a set is good exactly when it mixes the classes -- and the class changes in RUNS along the order in which the allocator hands
    // memory out: consecutive hipMallocs share it for anything from 2 to over 100 GiB.  On a device in such a stretch 28 consecutive
    // candidate sets inside the 16 GiB budget were all bad (1.165e8 leapfrog-steps/s instead of 1.30e8).  So: one reference array,
    // then single arrays further and further along -- spacers of growing size are held in between, untouched -- each probed as a PAIR
    // with the reference until one of the other class turns up; the set is the reference, that partner and the rejected ones.
    // Bounded by IDHMC_PLACEMENT_WALK_BYTES (default 64 GiB, never more than half of the free memory; with 128 GiB one walk that found nothing took 4 s, with 64 GiB 33 ms) and by
    // 250 ms of wall time held at one time, all of it
    // given back before the call returns.  IDHMC_PLACEMENT_PAIRS=0 goes straight to the walk over whole sets below.
    {
        const char *pw = getenv("IDHMC_PLACEMENT_PAIRS");
        int64_t walk_budget = (int64_t)64 << 30;
        if (const char *e = getenv("IDHMC_PLACEMENT_WALK_BYTES")) walk_budget = atoll(e);
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (int64_t)(free_b / 2) < walk_budget) walk_budget = (int64_t)(free_b / 2);
        }
        if (tries > 1 && !(pw && pw[0] == '0') && walk_budget >= 2 * set_bytes) {
            std::vector<void *> spacer_blocks;
            std::vector<double *> same;             // arrays of the reference's class, in the order found
            std::vector<double *> other;            // arrays of the other class
            double *ref = nullptr;
            int64_t held = 0, peak = 0;
            int steps = 0;
            double one = 0.0;
            auto give_back = [&]() {
                (void)hipStreamSynchronize(c->stream);
                for (void *p : spacer_blocks) (void)hipFree(p);
                for (double *p : same) (void)hipFree(p);
                for (double *p : other) (void)hipFree(p);
                if (ref) (void)hipFree(ref);
                spacer_blocks.clear(); same.clear(); other.clear(); ref = nullptr;
            };
            auto take = [&](size_t nbytes, bool touch) -> void * {
                void *p = nullptr;
                if (held + (int64_t)nbytes > walk_budget || hipMalloc(&p, nbytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
                if (touch && hipMemsetAsync(p, 0, nbytes, c->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
                held += (int64_t)nbytes;
                if (held > peak) peak = held;
                return p;
            };
            bool ok = (ref = (double *)take(bytes, true)) != nullptr;
            if (ok) {
                float ms1 = 0.f;
                double *v1[1] = {ref};
                ok = probe_ms(c, v1, 1, C, L, &ms1) == IDHMC_OK && ms1 > 0.f;
                if (ok) one = 2.0 * (double)bytes * 4 / (ms1 * 1e-3);
            }
            const int need_other = nvec >= 4 ? 2 : 1, max_steps = tries > 40 ? 40 : tries;
            int64_t jump = 0;
            while (ok && (int)other.size() < need_other && steps < max_steps &&
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count() < 250.0) {
                if (jump > 0) {
                    void *sp = take((size_t)jump, false);
                    if (!sp) break;
                    spacer_blocks.push_back(sp);
                }
                double *x = (double *)take(bytes, true);
                if (!x) break;
                float ms2 = 0.f;
                double *v2[2] = {ref, x};
                if (probe_ms(c, v2, 2, C, L, &ms2) != IDHMC_OK || !(ms2 > 0.f)) { (void)hipFree(x); ok = false; break; }
                const double r2 = 2.0 * 2.0 * (double)bytes * 4 / (ms2 * 1e-3);
                ++steps;
                if (verbose) fprintf(stderr, "idhmc placement pair walk step %d (%.1f GiB held): %.1f GB/s = %.3f x one array alone (%.1f GB/s) at %p\n", steps,
                                     held / 1073741824.0, r2 / 1e9, r2 / one, one / 1e9, (void *)x);
                if (r2 >= kGoodRatio * one) other.push_back(x); else same.push_back(x);
                // further along every time nothing turned up: 0, 0, 1, 2, 4, 8, 16, 16, ... GiB of untouched memory in between
                if (other.empty()) jump = steps < 2 ? 0 : (jump == 0 ? ((int64_t)1 << 30) : (jump < ((int64_t)16 << 30) ? jump * 2 : jump));
                else jump = 0;
            }
            if (ok && !other.empty()) {
                // q: the reference; p: the partner; grad: one of the reference's class (a rejected one, else new); a per-chain M^-1: the other class
                double *setv[4] = {ref, other[0], nullptr, nullptr};
                auto pick = [&](std::vector<double *> &from) -> double * {
                    if (!from.empty()) { double *p = from.back(); from.pop_back(); return p; }
                    return (double *)take(bytes, true);
                };
                other.erase(other.begin());
                if (nvec >= 3) setv[2] = pick(same);
                if (nvec >= 4) setv[3] = pick(other.empty() ? same : other);
                bool have = true;
                for (int k = 0; k < nvec; ++k) have = have && setv[k] != nullptr;
                float msn = 0.f;
                if (have && probe_ms(c, setv, nvec, C, L, &msn) == IDHMC_OK && msn > 0.f && probe_bytes / (msn * 1e-3) >= kGoodRatio * one) {
                    ref = nullptr;                                  // kept: not given back
                    for (int k = 0; k < nvec; ++k) { out[k] = setv[k]; c->allocs.push_back(setv[k]); }
                    give_back();
                    c->bytes += set_bytes;
                    c->placement_kind = 3;
                    c->placement_GBps = probe_bytes / (msn * 1e-3) / 1e9;
                    c->placement_tries = steps;
                    c->placement_single_GBps = one / 1e9;
                    c->placement_peak_bytes = peak;
                    c->placement_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
                    if (verbose) fprintf(stderr, "idhmc placement pair walk: set of %d at %.1f GB/s = %.3f x one array alone after %d steps, %.1f GiB held at most\n", nvec,
                                         c->placement_GBps, c->placement_GBps * 1e9 / one, steps, peak / 1073741824.0);
                    return IDHMC_OK;
                }
                for (int k = 1; k < nvec; ++k) if (setv[k]) spacer_blocks.push_back(setv[k]);      // (given back with the rest)
            }
            give_back();
        }
    }
    for (int t = 0; t < tries; ++t) {
        const char *kind = "sets";
        bool ok = false;
        if (want_vmm) {                               // (V)
            want_vmm = false;
            kind = "vmm";
            const int64_t need = set_bytes + (int64_t)(nvec * askew);
            if (cs.held_now + need <= budget && vmm_make(cs.vmm[t], c->device, (size_t)need)) {
                ok = true;
                cs.held[t] = (int64_t)cs.vmm[t].size;
                for (int k = 0; k < nvec && ok; ++k) {
                    cs.arr[t][k] = (double *)((char *)cs.vmm[t].va + k * (bytes + askew));
                    ok = hipMemsetAsync(cs.arr[t][k], 0, bytes, c->stream) == hipSuccess;
                }
                if (!ok) { (void)hipGetLastError(); cs.vmm[t].release(); cs.held[t] = 0; }
            }
            if (!ok) { for (int k = 0; k < 4; ++k) cs.arr[t][k] = nullptr; --t; continue; }    // not available here: next kind, same index
        } else if (want_slab) {                       // (S)
            want_slab = false;
            kind = "slab";
            if (cs.held_now + slab_bytes <= budget && hipMalloc((void **)&cs.slab[t], (size_t)slab_bytes) == hipSuccess) {
                ok = true;
                cs.held[t] = slab_bytes;
                for (int k = 0; k < nvec && ok; ++k) {
                    cs.arr[t][k] = (double *)(cs.slab[t] + k * far_stride);
                    ok = hipMemsetAsync(cs.arr[t][k], 0, bytes, c->stream) == hipSuccess;
                }
                if (!ok) { (void)hipGetLastError(); (void)hipFree(cs.slab[t]); cs.held[t] = 0; }
            } else (void)hipGetLastError();
            if (!ok) { cs.slab[t] = nullptr; for (int k = 0; k < 4; ++k) cs.arr[t][k] = nullptr; --t; continue; }
        } else {                                      // ordinary set
            if (t > 0 && cs.held_now + set_bytes > budget) break;       // the budget is what bounds the walk
            ok = true;
            for (int k = 0; k < nvec && ok; ++k) {
                void *p = nullptr;
                ok = hipMalloc(&p, bytes) == hipSuccess && hipMemsetAsync(p, 0, bytes, c->stream) == hipSuccess;
                cs.arr[t][k] = (double *)p;
            }
            cs.held[t] = set_bytes;
            if (!ok) {                                // out of memory: what we have is what we get
                (void)hipGetLastError();
                cs.n = t + 1;
                cs.held_now += cs.held[t];
                cs.drop(t, nvec);
                cs.n = t;
                if (best < 0) return fail(IDHMC_ERR_ALLOC, "hipMalloc(%zu bytes) failed for the chain state", bytes);
                break;
            }
        }
        cs.n = t + 1;
        cs.held_now += cs.held[t];
        if (cs.held_now > cs.held_peak) cs.held_peak = cs.held_now;
        if (tries == 1) { best = t; break; }
        if (single_Bps == 0.0) {
            float ms1 = 0.f;
            if (int rc = probe_ms(c, cs.arr[t], 1, C, L, &ms1)) return rc;
            single_Bps = 2.0 * (double)bytes * 4 / (ms1 * 1e-3);
        }
        if (int rc = probe_ms(c, cs.arr[t], nvec, C, L, &cs.ms[t])) return rc;
        const double rate = probe_bytes / (cs.ms[t] * 1e-3);
        if (verbose) {
            fprintf(stderr, "idhmc placement candidate %d (%s): %.1f GB/s = %.3f x one array alone (%.1f GB/s), holding %.2f GiB  at", t, kind,
                    rate / 1e9, rate / single_Bps, single_Bps / 1e9, cs.held_now / 1073741824.0);
            for (int k = 0; k < nvec; ++k) fprintf(stderr, " %p", (void *)cs.arr[t][k]);
            fprintf(stderr, "\n");
        }
        if (best < 0 || cs.ms[t] < cs.ms[best]) best = t;
        if (rate >= kGoodRatio * single_Bps) { best = t; break; }      // a good one: stop looking
        if (spacers) {
            // a candidate that is not the best so far only has to keep the allocator from handing the same memory out again: one of
            // its arrays does that (the next set then pairs the two freed blocks with a new one) -- three times as many candidates
            // inside the same byte budget
            for (int u = 0; u <= t; ++u) {
                if (u == best || !cs.held[u] || cs.slab[u] || cs.vmm[u].va || cs.held[u] <= (int64_t)bytes) continue;
                for (int k = 1; k < nvec; ++k) if (cs.arr[u][k]) { (void)hipFree(cs.arr[u][k]); cs.arr[u][k] = nullptr; }
                cs.held_now -= cs.held[u] - (int64_t)bytes;
                cs.held[u] = (int64_t)bytes;
            }
        }
    }
    if (best < 0) return fail(IDHMC_ERR_ALLOC, "no placement for the chain state (%zu bytes per array)", bytes);
    for (int t = 0; t < cs.n; ++t) if (t != best && cs.held[t]) cs.drop(t, nvec);
    c->placement_tries = cs.n;
    c->placement_GBps = (cs.n > 1 || cs.ms[best] > 0.f) && cs.ms[best] > 0.f ? probe_bytes / (cs.ms[best] * 1e-3) / 1e9 : 0.0;
    c->placement_single_GBps = single_Bps / 1e9;
    c->placement_peak_bytes = cs.held_peak;
    for (int k = 0; k < nvec; ++k) out[k] = cs.arr[best][k];
    if (cs.vmm[best].va) { c->vmm = cs.vmm[best]; cs.vmm[best] = VmmBlock(); c->placement_kind = 2; }
    else if (cs.slab[best]) { c->allocs.push_back(cs.slab[best]); c->placement_kind = 1; }
    else { for (int k = 0; k < nvec; ++k) c->allocs.push_back(cs.arr[best][k]); c->placement_kind = 0; }
    c->bytes += cs.held[best];          // (a slab's unused space between the arrays is allocated all the same)
    cs.held[best] = 0;                  // kept: not the holder's to free any more
    c->placement_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return IDHMC_OK;
}
// give one allocation of the context back early (staging buffers that are outgrown); `bytes` as it was counted by dalloc
static void dfree(idhmc_ctx *c, void *p, int64_t bytes)
{
    if (!p) return;
    for (size_t i = 0; i < c->allocs.size(); ++i)
        if (c->allocs[i] == p) { c->allocs.erase(c->allocs.begin() + (long)i); break; }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(p);
    c->bytes -= bytes;
}
#define DALLOC(ptr, n)                                       \
    do {                                                     \
        int rc_ = dalloc(c, &(ptr), (n));                    \
        if (rc_) { idhmc_destroy(c); return rc_; }           \
    } while (0)

extern "C" {

void idhmc_default_options(idhmc_options *o)
{
    if (!o) return;
    o->max_depth = 10;                    // DEFAULT_MAX_TREE_DEPTH, src/tree.jl:2
    o->min_delta = -1000.0;               // src/NUTS.jl:214
    o->da_delta = 0.8; o->da_gamma = 0.05; o->da_kappa = 0.75; o->da_t0 = 10;   // src/stepsize.jl:191
    o->ss_a_min = 0.25; o->ss_a_max = 0.75; o->ss_eps0 = 1.0; o->ss_C = 2.0;    // src/stepsize.jl:29
    o->ss_maxiter_crossing = 400; o->ss_maxiter_bisect = 400;
    o->init_steps = 75; o->middle_steps = 25; o->doubling_stages = 5; o->terminating_steps = 50;  // src/warmup.jl:366
    o->adapt_metric = 1;
    o->stepsize_search = 1;
    o->eps_init = 1.0;
    o->eps_mode = IDHMC_EPS_PER_CHAIN;
    o->metric_mode = IDHMC_METRIC_PER_CHAIN;
    o->local_opt_iterations = 0;          // the FindLocalOptimum stage is opt-in at this level (own optimiser)
    o->leapfrog_grad_mode = IDHMC_GRAD_STORE;
    o->local_opt_penalty = 1e-4;          // src/warmup.jl:143
}
const char *idhmc_last_error(void) { return g_err; }
int idhmc_version(void) { return IDHMC_VERSION; }
#ifndef IDHMC_SOURCE_DIGEST
#define IDHMC_SOURCE_DIGEST "unknown"
#endif
const char *idhmc_build_digest(void) { return IDHMC_SOURCE_DIGEST; }

int idhmc_destroy(idhmc_ctx *c)
{
    if (!c) return IDHMC_OK;
    (void)hipSetDevice(c->device);
    for (int k = 1; k < idhmc_ctx::kLanes; ++k) if (c->lane[k]) (void)hipStreamSynchronize(c->lane[k]);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int k = 1; k < idhmc_ctx::kLanes; ++k) {
        if (c->lane[k]) (void)hipStreamDestroy(c->lane[k]);
        if (c->lane_ev[k]) (void)hipEventDestroy(c->lane_ev[k]);
    }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    for (void *p : c->allocs) (void)hipFree(p);
    c->vmm.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (int b = 0; b < 2; ++b) if (c->ev_packed[b]) (void)hipEventDestroy(c->ev_packed[b]);
    if (c->ring) (void)hipHostFree(c->ring);
    jit_destroy(c->jit);
    comm_destroy(c->comm);
    delete c;
    return IDHMC_OK;
}

int idhmc_create(idhmc_ctx **out, int device, int64_t nchains, int64_t first_chain_id,
                 const idhmc_model_desc *model, const idhmc_options *opt_in, uint64_t seed)
{
    if (!out || !model) return fail(IDHMC_ERR_BAD_ARG, "idhmc_create: null argument");
    *out = nullptr;
    idhmc_options opt;
    if (opt_in) opt = *opt_in; else idhmc_default_options(&opt);
    if (nchains < 1 || nchains > (int64_t)0x7fffffff) return fail(IDHMC_ERR_BAD_ARG, "nchains = %lld out of range", (long long)nchains);
    if (first_chain_id < 0 || first_chain_id + nchains > (int64_t)0xffffffffll) return fail(IDHMC_ERR_BAD_ARG, "chain ids must fit 32 bits");
    if (model->D < 1 || model->D > 2048) return fail(IDHMC_ERR_BAD_ARG, "D = %d unsupported (1..2048)", model->D);
    if (model->D > 1024) {
        // two register tiles per vector; the dense MVN's matrix (32 MB at D = 2048) has no kernel built for it
        if (model->kind == IDHMC_MODEL_DENSE_MVN)
            return fail(IDHMC_ERR_BAD_ARG, "D = %d: the dense density is limited to D <= 1024", model->D);
    }
    if (opt_in && (opt_in->metric_mode < 0 || opt_in->metric_mode > IDHMC_METRIC_POOLED)) return fail(IDHMC_ERR_BAD_ARG, "unknown metric_mode %d", opt_in->metric_mode);
    if (model->kind < 0 || model->kind > IDHMC_MODEL_CUSTOM) return fail(IDHMC_ERR_BAD_ARG, "unknown model kind %d", model->kind);
    if (model->kind == IDHMC_MODEL_CUSTOM) {
        if (!model->source || !model->source[0]) return fail(IDHMC_ERR_BAD_ARG, "custom model needs HIP source");
        if (model->nparams < 0 || (model->nparams > 0 && !model->params)) return fail(IDHMC_ERR_BAD_ARG, "custom model: bad params");
        if (model->D > 512 && opt.metric_mode == IDHMC_METRIC_PER_CHAIN)
            return fail(IDHMC_ERR_BAD_ARG, "custom model with D > 512 needs metric_mode = SHARED (LDS budget of the NUTS kernel)");
    } else if (model->kind != IDHMC_MODEL_ISO_GAUSSIAN && !model->mu) return fail(IDHMC_ERR_BAD_ARG, "model needs mu");
    if (model->kind == IDHMC_MODEL_DIAG_GAUSSIAN && !model->tau) return fail(IDHMC_ERR_BAD_ARG, "diagonal model needs tau");
    if (model->kind == IDHMC_MODEL_DENSE_MVN && !model->prec) return fail(IDHMC_ERR_BAD_ARG, "dense model needs prec");
    if (model->kind == IDHMC_MODEL_DENSE_MVN) {
        // the gradient kernel reads row c of P as column c (coalesced): P must be exactly symmetric
        const int D = model->D;
        for (int r = 0; r < D; ++r)
            for (int c2 = r + 1; c2 < D; ++c2)
                if (model->prec[(size_t)r * D + c2] != model->prec[(size_t)c2 * D + r])
                    return fail(IDHMC_ERR_BAD_ARG, "prec must be exactly symmetric (differs at [%d,%d]); pass (P+P')/2", r, c2);
        if (D > 512 && opt.metric_mode == IDHMC_METRIC_PER_CHAIN)
            return fail(IDHMC_ERR_BAD_ARG, "dense model with D > 512 needs metric_mode = SHARED (LDS budget of the NUTS kernel)");
    }
    if (opt.max_depth < 1 || opt.max_depth > 15) return fail(IDHMC_ERR_BAD_ARG, "max_depth = %d unsupported (1..15)", opt.max_depth);
    if (!(opt.min_delta < 0)) return fail(IDHMC_ERR_BAD_ARG, "min_delta must be negative");
    if (!(opt.eps_init > 0)) return fail(IDHMC_ERR_BAD_ARG, "eps_init must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(IDHMC_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(IDHMC_ERR_BAD_ARG, "device %d out of range (%d visible)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));

    idhmc_ctx *c = new (std::nothrow) idhmc_ctx();
    if (!c) return fail(IDHMC_ERR_ALLOC, "out of host memory");
    c->device = device;
    c->opt = opt;
    hipError_t se = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return fail(IDHMC_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(se)); }
    c->stream = c->own_stream;
    (void)hipEventCreate(&c->ev0);
    (void)hipEventCreate(&c->ev1);

    DevState &s = c->s;
    s.C = nchains;
    s.D = model->D;
    // a vector is padded to the next multiple of 128 (the reference pads to its SIMD width, src/mcmc.jl:117); the
    // dense density's matrix kernels need a power-of-two number of 128-column chunks
    int nch = (model->D + 127) / 128;
    if (model->kind == IDHMC_MODEL_DENSE_MVN) { nch = 1; while (nch * 128 < model->D) nch *= 2; }
    s.nch = nch;
    s.L = 128 * nch;
    s.model = model->kind;
    s.k0 = (uint32_t)seed;
    s.k1 = (uint32_t)(seed >> 32);
    s.first_chain = (uint32_t)first_chain_id;
    s.max_depth = opt.max_depth;
    s.min_delta = opt.min_delta;
    s.da_delta = opt.da_delta; s.da_gamma = opt.da_gamma; s.da_kappa = opt.da_kappa; s.da_t0 = opt.da_t0;
    s.eps_mode = opt.eps_mode;
    s.ss_a_min = opt.ss_a_min; s.ss_a_max = opt.ss_a_max; s.ss_eps0 = opt.ss_eps0; s.ss_C = opt.ss_C;
    s.ss_maxiter_crossing = opt.ss_maxiter_crossing; s.ss_maxiter_bisect = opt.ss_maxiter_bisect;

    const int64_t CL = nchains * s.L;
    // the dense leapfrog's matrix-core kernel reads whole 32-chain tiles: rows past the last chain exist (zeros), see kRowPad
    const int64_t CLp = CL + (model->kind == IDHMC_MODEL_DENSE_MVN ? (int64_t)kRowPad * s.L : 0);
    const bool own_minv = opt.metric_mode == IDHMC_METRIC_PER_CHAIN;
    {
        double *sv[4] = {nullptr, nullptr, nullptr, nullptr};
        if (int rc = place_state(c, sv, own_minv ? 4 : 3, CLp, nchains, s.L)) { idhmc_destroy(c); return rc; }
        s.q = sv[0]; s.p = sv[1]; s.g = sv[2];
        if (own_minv) s.minv = sv[3];
    }
    DALLOC(s.lq, nchains); DALLOC(s.pi, nchains); DALLOC(s.eps, nchains);
    if (own_minv) {
        DALLOC(s.w, CL);
        s.minv_stride = s.L;
        DALLOC(s.mw_x1, CL); DALLOC(s.mw_s1, CL); DALLOC(s.mw_s2, CL);
    } else {
        DALLOC(s.minv, s.L); DALLOC(s.w, s.L);
        s.minv_stride = 0;
        if (opt.metric_mode == IDHMC_METRIC_POOLED) {       // one M^-1, adapted from every chain's window
            DALLOC(s.mw_x1, CL); DALLOC(s.mw_s1, CL); DALLOC(s.mw_s2, CL);
            DALLOC(c->pool_scratch, (int64_t)pool_scratch_doubles(s.L));
        }
    }
    DALLOC(s.mw_n, nchains);
    DALLOC(s.stats, nchains);
    DALLOC(s.directions, nchains);
    DALLOC(s.queue, 16);
    DALLOC(s.iters_done, nchains);
    DALLOC(s.da.mu, nchains); DALLOC(s.da.Hbar, nchains); DALLOC(s.da.logeps, nchains);
    DALLOC(s.da.logeps_bar, nchains); DALLOC(s.da.m, nchains);
    DALLOC(s.da_global, 8);
    DALLOC(s.xchg_acc, 3 * kXchgBlocks + 1);
    DALLOC(s.status, nchains);
    DALLOC(s.total_steps, 32);
    DALLOC(c->xchg, IDHMC_XCHG_DOUBLES);
    DALLOC(c->status_out, 1);
    {
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&c->ring), sizeof(unsigned long long) * idhmc_ctx::kRing * idhmc_ctx::kPulseWords,
                                     hipHostMallocDefault);
        if (e != hipSuccess) { idhmc_destroy(c); return fail(IDHMC_ERR_ALLOC, "pinned ring: %s", hipGetErrorString(e)); }
        for (int i = 0; i < idhmc_ctx::kRing * idhmc_ctx::kPulseWords; ++i) c->ring[i] = ~0ull;
        if (const char *w = getenv("IDHMC_NUTS_WIDE")) c->force_wide = atoi(w) != 0;
        if (const char *w = getenv("IDHMC_FUSE")) c->fuse = atoi(w) != 0;
        {   // several transitions per launch need workgroups b and b + 8 on one XCD (idhmc_nuts_kernel.hpp): look before relying on it
            const int g = prop.multiProcessorCount > 8 ? prop.multiProcessorCount : 8;
            uint32_t *dx = nullptr;
            std::vector<uint32_t> hx((size_t)g, 0u);
            bool ok = hipMalloc(&dx, sizeof(uint32_t) * g) == hipSuccess;
            ok = ok && launch_xcc_probe(dx, g, c->stream) == hipSuccess;
            ok = ok && hipMemcpyAsync(hx.data(), dx, sizeof(uint32_t) * g, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
            ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
            for (int b = 8; ok && b < g; ++b) ok = hx[(size_t)b] == hx[(size_t)(b & 7)];
            if (dx) (void)hipFree(dx);
            c->fuse_ok = ok;
        }
        if (const char *w = getenv("IDHMC_DENSE_LANES")) c->use_lanes = atoi(w) < idhmc_ctx::kLanes ? atoi(w) : idhmc_ctx::kLanes;
    }
    // model parameters, padded with zeros
    {
        double *mu = nullptr, *tau = nullptr, *prec = nullptr;
        DALLOC(mu, s.L); DALLOC(tau, s.L);
        if (model->mu) HIPCHK(hipMemcpyAsync(mu, model->mu, sizeof(double) * s.D, hipMemcpyHostToDevice, c->stream));
        if (model->tau) HIPCHK(hipMemcpyAsync(tau, model->tau, sizeof(double) * s.D, hipMemcpyHostToDevice, c->stream));
        if (model->kind == IDHMC_MODEL_DENSE_MVN) {
            DALLOC(prec, (int64_t)s.L * s.L);
            HIPCHK(hipMemcpy2DAsync(prec, sizeof(double) * s.L, model->prec, sizeof(double) * s.D,
                                    sizeof(double) * s.D, s.D, hipMemcpyHostToDevice, c->stream));
        }
        s.mu = mu; s.tau = tau; s.prec = prec;
    }
    // persistent NUTS waves and their tree arenas
    {
        // one workgroup of W wavefronts per CU (W = 4: one wavefront per SIMD with the full 512-register
        // budget; its LDS footprint and registers allow no more); slots in multiples of W
        const int W0 = nuts_waves_per_block(s.nch, s.model, opt.metric_mode != IDHMC_METRIC_PER_CHAIN), W1 = nuts_wide_waves_per_block(s.nch, s.model);
        const int W = W1 > W0 ? W1 : W0;
        int64_t nslots = (int64_t)prop.multiProcessorCount * W;
        const int64_t need = (nchains + W - 1) / W * W;
        if (nslots > need) nslots = need;
        s.nslots = (int32_t)nslots;
        s.arena_stride = (int64_t)arena_vectors(opt.max_depth, s.model, s.L) * s.L;
        DALLOC(s.arena, s.arena_stride * nslots);
    }
    // a user-supplied density: upload its parameters and compile it against the kernel templates (hipRTC)
    if (model->kind == IDHMC_MODEL_CUSTOM) {
        double *up = nullptr;
        DALLOC(up, model->nparams);
        if (model->nparams > 0) {
            hipError_t e = hipMemcpyAsync(up, model->params, sizeof(double) * (size_t)model->nparams, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) { idhmc_destroy(c); return fail(IDHMC_ERR_HIP, "params upload failed: %s", hipGetErrorString(e)); }
        }
        s.user_params = up;
        s.user_nparams = model->nparams;
        static thread_local char jlog[400];
        jlog[0] = 0;
        const int jrc = jit_build(s, model->source, &c->jit, jlog, sizeof jlog);
        if (jrc != 0) { idhmc_destroy(c); return fail(IDHMC_ERR_BAD_ARG, "custom density did not compile (%d): %s", jrc, jlog); }
        s.jit = c->jit;
    }
    // kappa = I (GaussianKineticEnergy(sptr, Static{D}, 1.0), src/hamiltonian.jl:63-74)
    {
        const int64_t n = s.minv_stride ? CL : (int64_t)s.L;
        hipError_t e = launch_fill(s.minv, 1.0, n, c->stream);
        if (e == hipSuccess) e = launch_fill(s.w, 1.0, n, c->stream);
        if (e == hipSuccess) e = launch_fill(s.eps, opt.eps_init, nchains, c->stream);
        if (e != hipSuccess) { idhmc_destroy(c); return fail(IDHMC_ERR_HIP, "init kernels failed: %s", hipGetErrorString(e)); }
    }
    {
        hipError_t e = launch_eval(s, c->stream);   // q = 0: consistent (lq, grad)
        if (e != hipSuccess && e != hipErrorNotSupported) { idhmc_destroy(c); return fail(IDHMC_ERR_HIP, "eval failed: %s", hipGetErrorString(e)); }
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { idhmc_destroy(c); return fail(IDHMC_ERR_HIP, "init sync failed: %s", hipGetErrorString(e)); }
    }
    *out = c;
    return IDHMC_OK;
}

int idhmc_set_stream(idhmc_ctx *c, void *hip_stream)
{
    CTXCHK(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return IDHMC_OK;
}
int idhmc_synchronize(idhmc_ctx *c)
{
    CTXCHK(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    return IDHMC_OK;
}
int64_t idhmc_nchains(const idhmc_ctx *c) { return c ? c->s.C : 0; }
int32_t idhmc_dim(const idhmc_ctx *c) { return c ? c->s.D : 0; }
int32_t idhmc_padded_dim(const idhmc_ctx *c) { return c ? c->s.L : 0; }
int64_t idhmc_device_bytes(const idhmc_ctx *c) { return c ? c->bytes : 0; }
int idhmc_placement_info(const idhmc_ctx *c, double *probe_GBps, int32_t *candidates)
{
    if (!c) return fail(IDHMC_ERR_BAD_ARG, "null context");
    if (probe_GBps) *probe_GBps = c->placement_GBps;
    if (candidates) *candidates = c->placement_tries;
    return IDHMC_OK;
}
int idhmc_lanes_info(const idhmc_ctx *c, int32_t *lanes, int32_t *on_distinct_queues)
{
    if (!c) return fail(IDHMC_ERR_BAD_ARG, "null context");
    int n = c->lane[1] ? 1 : 0;
    for (int k = 1; k < idhmc_ctx::kLanes; ++k) n += c->lane[k] != nullptr;
    if (lanes) *lanes = n;
    if (on_distinct_queues) *on_distinct_queues = c->lanes_distinct;
    return IDHMC_OK;
}
int idhmc_placement_cost(const idhmc_ctx *c, double *create_ms, int64_t *peak_transient_bytes, double *single_array_GBps, int32_t *kind)
{
    if (!c) return fail(IDHMC_ERR_BAD_ARG, "null context");
    if (create_ms) *create_ms = c->placement_ms;
    if (peak_transient_bytes) *peak_transient_bytes = c->placement_peak_bytes;
    if (single_array_GBps) *single_array_GBps = c->placement_single_GBps;
    if (kind) *kind = c->placement_kind;
    return IDHMC_OK;
}

// host [C][D] <-> device [C][L]
static int put_vec(idhmc_ctx *c, double *dst, const double *src, int64_t rows)
{
    const DevState &s = c->s;
    HIPCHK(hipMemcpy2DAsync(dst, sizeof(double) * s.L, src, sizeof(double) * s.D, sizeof(double) * s.D,
                            (size_t)rows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return IDHMC_OK;
}
static int get_vec(idhmc_ctx *c, double *dst, const double *src, int64_t rows)
{
    const DevState &s = c->s;
    HIPCHK(hipMemcpy2DAsync(dst, sizeof(double) * s.D, src, sizeof(double) * s.L, sizeof(double) * s.D,
                            (size_t)rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return IDHMC_OK;
}

static int ensure_grad(idhmc_ctx *c);
int idhmc_set_q(idhmc_ctx *c, const double *q)
{
    CTXCHK(c);
    if (!q) return fail(IDHMC_ERR_BAD_ARG, "null q");
    if (int rc = put_vec(c, c->s.q, q, c->s.C)) return rc;
    HIPCHK(launch_eval(c->s, c->stream));
    c->grad_stale = false;
    return IDHMC_OK;
}
int idhmc_random_position(idhmc_ctx *c)
{
    CTXCHK(c);
    HIPCHK(launch_random_position(c->s, c->stream));
    c->grad_stale = false;
    return IDHMC_OK;
}
int idhmc_set_p(idhmc_ctx *c, const double *p)
{
    CTXCHK(c);
    if (!p) return fail(IDHMC_ERR_BAD_ARG, "null p");
    return put_vec(c, c->s.p, p, c->s.C);
}
int idhmc_set_minv(idhmc_ctx *c, const double *minv, int per_chain)
{
    CTXCHK(c);
    if (!minv) return fail(IDHMC_ERR_BAD_ARG, "null minv");
    DevState &s = c->s;
    const int64_t n = per_chain ? s.C * (int64_t)s.D : (int64_t)s.D;
    for (int64_t i = 0; i < n; ++i)
        if (!(minv[i] > 0.0) || !std::isfinite(minv[i])) return fail(IDHMC_ERR_BAD_ARG, "M^-1 must be positive and finite");
    if (s.minv_stride == 0) {
        if (per_chain) return fail(IDHMC_ERR_BAD_ARG, "context has a shared metric (metric_mode = SHARED)");
        if (int rc = put_vec(c, s.minv, minv, 1)) return rc;
    } else if (per_chain) {
        if (int rc = put_vec(c, s.minv, minv, s.C)) return rc;
    } else {
        // broadcast D values to every chain: row 0 from the host, the rest on the device
        if (int rc = put_vec(c, s.minv, minv, 1)) return rc;
        HIPCHK(launch_broadcast_row(s.minv, s.L, s.C, c->stream));
    }
    HIPCHK(launch_set_w(s, c->stream));
    return IDHMC_OK;
}
int idhmc_set_eps(idhmc_ctx *c, double eps)
{
    CTXCHK(c);
    if (!(eps > 0.0) || !std::isfinite(eps)) return fail(IDHMC_ERR_BAD_ARG, "eps must be positive and finite");
    HIPCHK(launch_fill(c->s.eps, eps, c->s.C, c->stream));
    return IDHMC_OK;
}
int idhmc_set_eps_per_chain(idhmc_ctx *c, const double *eps)
{
    CTXCHK(c);
    if (!eps) return fail(IDHMC_ERR_BAD_ARG, "null eps");
    HIPCHK(hipMemcpyAsync(c->s.eps, eps, sizeof(double) * c->s.C, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return IDHMC_OK;
}
int idhmc_get_q(idhmc_ctx *c, double *q) { CTXCHK(c); return q ? get_vec(c, q, c->s.q, c->s.C) : fail(IDHMC_ERR_BAD_ARG, "null out"); }
int idhmc_get_p(idhmc_ctx *c, double *p) { CTXCHK(c); return p ? get_vec(c, p, c->s.p, c->s.C) : fail(IDHMC_ERR_BAD_ARG, "null out"); }
int idhmc_get_grad(idhmc_ctx *c, double *g)
{
    CTXCHK(c);
    if (!g) return fail(IDHMC_ERR_BAD_ARG, "null out");
    if (int rc = ensure_grad(c)) return rc;
    return get_vec(c, g, c->s.g, c->s.C);
}
int idhmc_get_minv(idhmc_ctx *c, double *m)
{
    CTXCHK(c);
    if (!m) return fail(IDHMC_ERR_BAD_ARG, "null out");
    const DevState &s = c->s;
    if (s.minv_stride) return get_vec(c, m, s.minv, s.C);
    if (int rc = get_vec(c, m, s.minv, 1)) return rc;
    for (int64_t ch = 1; ch < s.C; ++ch) memcpy(m + ch * s.D, m, sizeof(double) * s.D);
    return IDHMC_OK;
}
static int get_scalar(idhmc_ctx *c, void *dst, const void *src, size_t bytes)
{
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return IDHMC_OK;
}
int idhmc_get_lq(idhmc_ctx *c, double *lq) { CTXCHK(c); return lq ? get_scalar(c, lq, c->s.lq, sizeof(double) * c->s.C) : fail(IDHMC_ERR_BAD_ARG, "null out"); }
int idhmc_get_eps(idhmc_ctx *c, double *e) { CTXCHK(c); return e ? get_scalar(c, e, c->s.eps, sizeof(double) * c->s.C) : fail(IDHMC_ERR_BAD_ARG, "null out"); }
int idhmc_logdensity(idhmc_ctx *c, double *pi)
{
    CTXCHK(c);
    if (!pi) return fail(IDHMC_ERR_BAD_ARG, "null out");
    HIPCHK(launch_logdensity(c->s, c->stream));
    return get_scalar(c, pi, c->s.pi, sizeof(double) * c->s.C);
}

int idhmc_refresh_momentum(idhmc_ctx *c, uint32_t iter)
{
    CTXCHK(c);
    HIPCHK(launch_refresh(c->s, iter, c->stream));
    return IDHMC_OK;
}
// the stored gradient is needed: bring it up to date (evaluate_l! from q: the same bits every leapfrog would have stored)
static int ensure_grad(idhmc_ctx *c)
{
    if (c->grad_stale) {
        HIPCHK(launch_eval(c->s, c->stream));
        c->grad_stale = false;
    }
    return IDHMC_OK;
}
static int leapfrog_regrad(const idhmc_ctx *c, int32_t n_steps)
{
    const bool separable = c->s.model == IDHMC_MODEL_ISO_GAUSSIAN || c->s.model == IDHMC_MODEL_DIAG_GAUSSIAN;
    return (c->opt.leapfrog_grad_mode == IDHMC_GRAD_RECOMPUTE && separable && n_steps == 1) ? 1 : 0;
}
int idhmc_set_leapfrog_grad_mode(idhmc_ctx *c, int32_t mode)
{
    CTXCHK(c);
    if (mode != IDHMC_GRAD_STORE && mode != IDHMC_GRAD_RECOMPUTE) return fail(IDHMC_ERR_BAD_ARG, "unknown gradient mode %d", mode);
    c->opt.leapfrog_grad_mode = mode;
    return IDHMC_OK;
}
int idhmc_leapfrog(idhmc_ctx *c, double eps, int32_t n_steps)
{
    CTXCHK_LANES(c);
    if (n_steps < 1) return fail(IDHMC_ERR_BAD_ARG, "n_steps must be >= 1");
    if (!std::isfinite(eps)) return fail(IDHMC_ERR_BAD_ARG, "eps must be finite");
    const int regrad = leapfrog_regrad(c, n_steps);
    if (!regrad) { if (int rc = ensure_grad(c)) return rc; }
    if (int rc = leapfrog_any(c, eps, 0, n_steps, regrad)) return rc;
    if (regrad) c->grad_stale = true;
    return IDHMC_OK;
}
int idhmc_leapfrog_own_eps(idhmc_ctx *c, int32_t n_steps)
{
    CTXCHK_LANES(c);
    if (n_steps < 1) return fail(IDHMC_ERR_BAD_ARG, "n_steps must be >= 1");
    const int regrad = leapfrog_regrad(c, n_steps);
    if (!regrad) { if (int rc = ensure_grad(c)) return rc; }
    if (int rc = leapfrog_any(c, 0.0, 1, n_steps, regrad)) return rc;
    if (regrad) c->grad_stale = true;
    return IDHMC_OK;
}
static int nuts_launch(idhmc_ctx *c, uint32_t iter, uint32_t flags, uint32_t n_iter, double *fz_q = nullptr, idhmc_tree_stats *fz_st = nullptr)
{
    if ((flags & IDHMC_T_ACCUM_METRIC) && !c->s.mw_x1) return fail(IDHMC_ERR_BAD_ARG, "shared-metric context cannot accumulate a metric window");
    if ((flags & IDHMC_T_ACCUM_MOMENTS) && !c->s.mom_mean) {
        if (int rc = idhmc_moments_reset(c)) return rc;
    }
    if ((flags & IDHMC_T_ACCUM_DIAG) && !c->s.diag.n) {
        if (int rc = idhmc_diag_reset(c)) return rc;
    }
    // Two wavefronts per SIMD are the faster form at every tree depth since the far edge and the whole-tree
    // statistic stopped travelling through the arena (round 2: 3.4e8 / 5.0e8 leapfrog/s at depth 4 / 7 against
    // 2.9e8 / 3.6e8 with one); IDHMC_NUTS_WIDE=0 still selects the one-wavefront form (experiments, tests).
    const int wide = nuts_wide_waves_per_block(c->s.nch, c->s.model) > 0 ? (c->force_wide >= 0 ? c->force_wide : 1) : 0;
    volatile unsigned long long *slot = c->ring + (c->launches % idhmc_ctx::kRing) * idhmc_ctx::kPulseWords;
    if (slot[0] == ~0ull && c->launches >= (uint64_t)idhmc_ctx::kRing) HIPCHK(hipStreamSynchronize(c->stream));   // slot still in flight
    slot[0] = ~0ull;
    HIPCHK(launch_nuts(c->s, iter, flags, wide, c->stream, n_iter, fz_q, fz_st));
    // the transition of a separable density leaves grad l of the new state unwritten (8 KB per chain and transition that nothing on
    // the sampling path reads: the kernel re-derives the gradient from q); whoever needs the array re-evaluates first (ensure_grad)
    if (c->s.model == IDHMC_MODEL_ISO_GAUSSIAN || c->s.model == IDHMC_MODEL_DIAG_GAUSSIAN) c->grad_stale = true;
    HIPCHK(hipMemcpyAsync(const_cast<unsigned long long *>(slot), c->s.total_steps + kPulseAt, sizeof(unsigned long long) * idhmc_ctx::kPulseWords,
                          hipMemcpyDeviceToHost, c->stream));
    ++c->launches;
    return IDHMC_OK;
}
int idhmc_nuts_transition(idhmc_ctx *c, uint32_t iter, uint32_t flags)
{
    CTXCHK(c);
    return nuts_launch(c, iter, flags, 1);
}
int idhmc_nuts_transitions(idhmc_ctx *c, uint32_t iter, int32_t n, uint32_t flags)
{
    CTXCHK(c);
    if (n < 1) return fail(IDHMC_ERR_BAD_ARG, "n must be >= 1");
    if ((uint64_t)c->s.C * (uint64_t)n >= (1ull << 31)) return fail(IDHMC_ERR_BAD_ARG, "nchains * n must be below 2^31");
    if (flags & (IDHMC_T_USE_DIRECTIONS | IDHMC_T_KEEP_P)) return fail(IDHMC_ERR_BAD_ARG, "injected directions / a kept momentum are one transition's");
    if ((flags & IDHMC_T_ADAPT_EPS) && c->s.eps_mode == IDHMC_EPS_GLOBAL) return fail(IDHMC_ERR_BAD_ARG, "the global stepsize adapts between transitions");
    if (!c->fuse_ok) {       // (a device whose workgroups b and b + 8 do not share an XCD: the same result from n launches)
        for (int32_t i = 0; i < n; ++i) { if (int rc = nuts_launch(c, iter + (uint32_t)i, flags, 1)) return rc; }
        return IDHMC_OK;
    }
    return nuts_launch(c, iter, flags, (uint32_t)n);
}
int idhmc_fused_launch_info(idhmc_ctx *c, int32_t *possible, int32_t *used_by_drivers)
{
    CTXCHK(c);
    if (possible) *possible = c->fuse_ok ? 1 : 0;
    if (used_by_drivers) *used_by_drivers = (c->fuse_ok && c->fuse != 0) ? 1 : 0;
    return IDHMC_OK;
}
// do the drivers make several transitions per launch (where nothing leaves the device per transition)?  Measured gains: dense
// configs[3] +29 % (3.5e8 against 2.7e8 leapfrog/s, 20 per launch), 1024-dim diagonal Gaussian at 65 536 chains +6-9 % (depth 4), +1.5 %
// (depth 7), D = 256 +22 %: the end of every launch and the gap to the next are paid once.  IDHMC_FUSE=0 restores one launch per transition.
static bool fuse_transitions(const idhmc_ctx *c)
{
    return c->fuse != 0 && c->fuse_ok;
}
// The abort code of the launch `lag` launches back (waiting for it to arrive: this is what bounds the drivers' run-ahead),
// 0 when there is none.  Used by the caller loops only; a caller driving idhmc_nuts_transition itself polls with
// idhmc_poll_abort.
static int pulse_abort(idhmc_ctx *c, int lag)
{
    if (c->launches < (uint64_t)lag + 1) return 0;
    volatile unsigned long long *slot = c->ring + ((c->launches - 1 - lag) % idhmc_ctx::kRing) * idhmc_ctx::kPulseWords;
    while (slot[0] == ~0ull) {
        if (hipStreamQuery(c->stream) == hipSuccess) break;       // everything has run (the copy included)
    }
    return slot[0] == ~0ull ? 0 : (int)slot[1];
}
int idhmc_poll_abort(idhmc_ctx *c, int32_t lag, int32_t *code)
{
    CTXCHK(c);
    if (lag < 0 || lag >= idhmc_ctx::kRing - 1 || !code) return fail(IDHMC_ERR_BAD_ARG, "lag must be in [0, %d)", idhmc_ctx::kRing - 1);
    *code = pulse_abort(c, lag);
    return IDHMC_OK;
}
int idhmc_set_directions(idhmc_ctx *c, const uint32_t *d)
{
    CTXCHK(c);
    if (!d) return fail(IDHMC_ERR_BAD_ARG, "null directions");
    HIPCHK(hipMemcpyAsync(c->s.directions, d, sizeof(uint32_t) * c->s.C, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return IDHMC_OK;
}
int idhmc_get_tree_stats(idhmc_ctx *c, idhmc_tree_stats *st)
{
    CTXCHK(c);
    if (!st) return fail(IDHMC_ERR_BAD_ARG, "null out");
    return get_scalar(c, st, c->s.stats, sizeof(idhmc_tree_stats) * c->s.C);
}

static int check_status(idhmc_ctx *c, const char *what)
{
    HIPCHK(launch_status_max(c->s, c->status_out, c->stream));
    int32_t st = 0;
    if (int rc = get_scalar(c, &st, c->status_out, sizeof st)) return rc;
    if (st == 0) return IDHMC_OK;
    HIPCHK(hipMemsetAsync(c->s.status, 0, sizeof(int32_t) * c->s.C, c->stream));
    HIPCHK(hipMemsetAsync(c->s.total_steps + kPulseAt + 1, 0, sizeof(unsigned long long), c->stream));   // the abort word
    for (int i = 0; i < idhmc_ctx::kRing; ++i)       // ... and its copies (the stream is idle: get_scalar synchronised)
        if (c->ring[i * idhmc_ctx::kPulseWords] != ~0ull) c->ring[i * idhmc_ctx::kPulseWords + 1] = 0;
    switch (st) {
    case IDHMC_ERR_EPS_UNDERFLOW: return fail(st, "%s: a chain's stepsize fell below 1e-10 (reference src/warmup.jl:291-296)", what);
    case IDHMC_ERR_STEPSIZE_SEARCH: return fail(st, "%s: reached maximum number of iterations searching for eps (reference src/stepsize.jl:71,101)", what);
    case IDHMC_ERR_NONFINITE_START: return fail(st, "%s: starting point has non-finite density (reference src/stepsize.jl:152-153)", what);
    case IDHMC_ERR_OPTIMIZATION: return fail(st, "%s: Optimization failed to converge (reference src/warmup.jl:172)", what);
    case IDHMC_ERR_HIP: return fail(st, "%s: a launch of several transitions gave up (a chain range served from two XCDs, or a hand-over that never came)", what);
    default: return fail(st, "%s: device status %d", what, st);
    }
}

// the exchange record lives in the caller's buffer when a hook is set, else in the library's
static double *xchg_buf(idhmc_ctx *c) { return c->hook ? c->hook_buf : c->xchg; }
// SUM-all-reduce the record over the ranks, on the context's stream: hook > communicator > single rank (nothing)
static int exchange(idhmc_ctx *c, double *buf)
{
    if (c->hook) {
        if (int rc = c->hook(buf, c->hook_user)) return fail(IDHMC_ERR_BAD_ARG, "all-reduce hook returned %d", rc);
    } else if (c->comm) {
        char err[200];
        if (comm_allreduce_sum(c->comm, buf, IDHMC_XCHG_DOUBLES, c->stream, err, sizeof err)) return fail(IDHMC_ERR_HIP, "%s", err);
    }
    return IDHMC_OK;
}

static int status_exchange(idhmc_ctx *c, const char *what);
int idhmc_find_local_optimum(idhmc_ctx *c, double magnitude_penalty, int32_t iterations)
{
    CTXCHK(c);
    if (!(magnitude_penalty >= 0.0) || iterations < 0) return fail(IDHMC_ERR_BAD_ARG, "penalty and iterations must be >= 0");
    if (int rc = ensure_grad(c)) return rc;
    HIPCHK(launch_local_optimum(c->s, magnitude_penalty, iterations, c->stream));
    return status_exchange(c, "find_local_optimum");     // sharded contexts agree on the outcome (one more 4-double exchange)
}
int idhmc_find_initial_stepsize_per_chain(idhmc_ctx *c)
{
    CTXCHK(c);
    if (int rc = ensure_grad(c)) return rc;
    HIPCHK(launch_stepsize_search(c->s, c->stream));
    return check_status(c, "find_initial_stepsize");
}
// A sharded stage fails on every rank or on none: `peers` is the all-reduced slot [3] of an exchange record (chains with a
// pending status, over all ranks).  The rank that owns such a chain returns its code, the others IDHMC_ERR_PEER.
static int status_agreed(idhmc_ctx *c, const char *what, double peers)
{
    if (int rc = check_status(c, what)) return rc;
    if (peers > 0.0) return fail(IDHMC_ERR_PEER, "%s: %.0f chain(s) of other rank(s) raised an error; the stage fails on every rank", what, peers);
    return IDHMC_OK;
}
// the same without a record at hand: one exchange of {0, 0, 0, local chains with a pending status}
static bool sharded(const idhmc_ctx *c) { return c->hook || c->comm; }
static int status_exchange(idhmc_ctx *c, const char *what)
{
    if (!sharded(c)) return check_status(c, what);
    double *buf = xchg_buf(c);
    HIPCHK(launch_xchg_sum(c->s, IDHMC_XCHG_STATUS, buf, c->stream));
    if (int rc = exchange(c, buf)) return rc;
    double rec[IDHMC_XCHG_DOUBLES];
    if (int rc = get_scalar(c, rec, buf, sizeof rec)) return rc;
    return status_agreed(c, what, rec[3]);
}
int idhmc_find_initial_stepsize(idhmc_ctx *c)
{
    CTXCHK(c);
    if (c->s.eps_mode != IDHMC_EPS_GLOBAL) return idhmc_find_initial_stepsize_per_chain(c);
    if (int rc = ensure_grad(c)) return rc;
    HIPCHK(launch_stepsize_search(c->s, c->stream));
    // one eps for everybody: exp(mean log eps) over the chains of ALL ranks -- the fixed-point record is exact under
    // any all-reduce order, and the engine's own dlog / dexp run on the device, so every rank holds the same bits.
    // The exchange is enqueued BEFORE any error is looked at: a rank whose search failed still takes part, and the record's
    // slot [3] tells every rank that the stage failed (a rank that returned early would leave the others in the all-reduce).
    double *buf = xchg_buf(c);
    HIPCHK(launch_xchg_sum(c->s, IDHMC_XCHG_LOGEPS, buf, c->stream));
    if (int rc = exchange(c, buf)) return rc;
    HIPCHK(launch_eps_from_logeps(c->s, buf, c->stream));
    double rec[IDHMC_XCHG_DOUBLES];
    if (int rc = get_scalar(c, rec, buf, sizeof rec)) return rc;
    return status_agreed(c, "find_initial_stepsize", rec[3]);
}
int idhmc_da_init(idhmc_ctx *c) { CTXCHK(c); HIPCHK(launch_da_init(c->s, c->stream)); return IDHMC_OK; }
int idhmc_da_finalize(idhmc_ctx *c) { CTXCHK(c); HIPCHK(launch_da_finalize(c->s, c->stream)); return IDHMC_OK; }
int idhmc_accept_sum(idhmc_ctx *c, double *dev_xchg)
{
    CTXCHK(c);
    if (!dev_xchg) return fail(IDHMC_ERR_BAD_ARG, "null device buffer");
    HIPCHK(launch_xchg_sum(c->s, IDHMC_XCHG_ACCEPT, dev_xchg, c->stream));
    return IDHMC_OK;
}
int idhmc_logeps_sum(idhmc_ctx *c, double *dev_xchg)
{
    CTXCHK(c);
    if (!dev_xchg) return fail(IDHMC_ERR_BAD_ARG, "null device buffer");
    HIPCHK(launch_xchg_sum(c->s, IDHMC_XCHG_LOGEPS, dev_xchg, c->stream));
    return IDHMC_OK;
}
int idhmc_da_adapt_global(idhmc_ctx *c, const double *dev_xchg)
{
    CTXCHK(c);
    if (!dev_xchg) return fail(IDHMC_ERR_BAD_ARG, "null device buffer");
    if (c->s.eps_mode != IDHMC_EPS_GLOBAL) return fail(IDHMC_ERR_BAD_ARG, "context is not in global-eps mode");
    HIPCHK(launch_da_adapt_global(c->s, dev_xchg, c->stream));
    return IDHMC_OK;
}
int idhmc_set_eps_from_logeps(idhmc_ctx *c, const double *dev_xchg)
{
    CTXCHK(c);
    if (!dev_xchg) return fail(IDHMC_ERR_BAD_ARG, "null device buffer");
    HIPCHK(launch_eps_from_logeps(c->s, dev_xchg, c->stream));
    return IDHMC_OK;
}
// host side of the same protocol (no device involved)
int idhmc_xchg_accumulate(int32_t kind, const double *values, int64_t n, double *xchg4)
{
    if (kind != IDHMC_XCHG_ACCEPT && kind != IDHMC_XCHG_LOGEPS) return fail(IDHMC_ERR_BAD_ARG, "unknown exchange kind %d", kind);
    if (n < 0 || (n > 0 && !values) || !xchg4) return fail(IDHMC_ERR_BAD_ARG, "bad arguments");
    long long hi = 0, lo = 0;
    for (int64_t i = 0; i < n; ++i) {
        long long h, l;
        xchg_limbs(kind, values[i], h, l);
        hi += h; lo += l;
    }
    xchg4[0] += (double)hi;
    xchg4[1] += (double)lo;
    xchg4[2] += (double)n;
    return IDHMC_OK;
}
int idhmc_xchg_mean(int32_t kind, const double *xchg4, double *mean)
{
    if (kind != IDHMC_XCHG_ACCEPT && kind != IDHMC_XCHG_LOGEPS) return fail(IDHMC_ERR_BAD_ARG, "unknown exchange kind %d", kind);
    if (!xchg4 || !mean) return fail(IDHMC_ERR_BAD_ARG, "bad arguments");
    *mean = xchg_mean(kind, xchg4[0], xchg4[1], xchg4[2]);
    return IDHMC_OK;
}
int idhmc_set_allreduce_hook(idhmc_ctx *c, idhmc_allreduce_fn fn, void *user, double *dev_xchg)
{
    CTXCHK(c);
    if (fn && !dev_xchg) return fail(IDHMC_ERR_BAD_ARG, "hook needs a device buffer");
    c->hook = fn; c->hook_user = user; c->hook_buf = dev_xchg;
    return IDHMC_OK;
}
int idhmc_comm_unique_id(void *id128)
{
    if (!id128) return fail(IDHMC_ERR_BAD_ARG, "null id buffer");
    char err[200];
    if (comm_unique_id(id128, err, sizeof err)) return fail(IDHMC_ERR_HIP, "%s", err);
    return IDHMC_OK;
}
int idhmc_comm_init(idhmc_ctx *c, int32_t nranks, int32_t rank, const void *id128)
{
    CTXCHK(c);
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(IDHMC_ERR_BAD_ARG, "bad communicator arguments");
    if (c->comm) return fail(IDHMC_ERR_BAD_ARG, "context already has a communicator");
    char err[200];
    c->comm = comm_create(nranks, rank, id128, err, sizeof err);
    if (!c->comm) return fail(IDHMC_ERR_HIP, "%s", err);
    return IDHMC_OK;
}
int idhmc_comm_destroy(idhmc_ctx *c)
{
    CTXCHK(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    comm_destroy(c->comm);
    c->comm = nullptr;
    return IDHMC_OK;
}
int idhmc_comm_allreduce(idhmc_ctx *c, double *dev_buf, int32_t n)
{
    CTXCHK(c);
    if (!dev_buf || n < 1) return fail(IDHMC_ERR_BAD_ARG, "bad buffer");
    if (!c->comm) return fail(IDHMC_ERR_BAD_ARG, "context has no communicator");
    char err[200];
    if (comm_allreduce_sum(c->comm, dev_buf, n, c->stream, err, sizeof err)) return fail(IDHMC_ERR_HIP, "%s", err);
    return IDHMC_OK;
}
int idhmc_comm_info(idhmc_ctx *c, int32_t *nranks, int32_t *rank, int64_t *allreduces)
{
    if (!c) return fail(IDHMC_ERR_BAD_ARG, "null context");
    int nr = 0, r = 0;
    long long n = 0;
    comm_info(c->comm, &nr, &r, &n);
    if (nranks) *nranks = nr;
    if (rank) *rank = r;
    if (allreduces) *allreduces = n;
    return IDHMC_OK;
}
static int pool_table_reserve(idhmc_ctx *c, long long nseg)
{
    if (nseg < 1 || nseg > 65535) return fail(IDHMC_ERR_BAD_ARG, "pooled metric: %lld segments of %d chains out of range", nseg, IDHMC_POOL_SEGMENT);
    if (nseg > c->pool_table_segs) {
        if (int rc = dalloc(c, &c->pool_table, nseg * (int64_t)(c->s.L + 1), false)) return rc;   // (the smaller one stays until destroy)
        c->pool_table_segs = nseg;
    }
    return IDHMC_OK;
}
// the pooled metric by hand (a host that exchanges the table itself), include/idhmc.h
int idhmc_pool_partials(idhmc_ctx *c, int32_t pass, double *dev_table, int64_t seg_lo, int64_t seg_hi)
{
    CTXCHK(c);
    if (c->s.minv_stride != 0 || !c->s.mw_x1) return fail(IDHMC_ERR_BAD_ARG, "context is not in pooled-metric mode");
    if (!dev_table || (pass != 0 && pass != 1) || seg_lo < 0 || seg_hi <= seg_lo) return fail(IDHMC_ERR_BAD_ARG, "bad arguments");
    HIPCHK(launch_pool_partials(c->s, pass, c->pool_scratch, dev_table, seg_lo, seg_hi, c->stream));
    return IDHMC_OK;
}
int idhmc_pool_consume(idhmc_ctx *c, int32_t pass, const double *dev_table, int64_t nseg, double lambda)
{
    CTXCHK(c);
    if (c->s.minv_stride != 0 || !c->s.mw_x1) return fail(IDHMC_ERR_BAD_ARG, "context is not in pooled-metric mode");
    if (!dev_table || (pass != 0 && pass != 1) || nseg < 1 || !(lambda >= 0.0)) return fail(IDHMC_ERR_BAD_ARG, "bad arguments");
    HIPCHK(launch_pool_consume(c->s, pass, c->pool_scratch, dev_table, nseg, lambda, c->stream));
    return IDHMC_OK;
}
int idhmc_metric_begin(idhmc_ctx *c)
{
    CTXCHK(c);
    HIPCHK(hipMemsetAsync(c->s.mw_n, 0, sizeof(int32_t) * c->s.C, c->stream));
    return IDHMC_OK;
}
int idhmc_metric_update(idhmc_ctx *c, double lambda)
{
    CTXCHK(c);
    if (!c->s.mw_x1) return fail(IDHMC_ERR_BAD_ARG, "shared-metric context has no metric window");
    if (!(lambda >= 0.0)) return fail(IDHMC_ERR_BAD_ARG, "lambda must be >= 0");
    if (c->s.minv_stride == 0) {
        // pooled: every chain's window, on every rank when the context has a communicator.  The table of per-segment
        // partials covers the global segments [0, ceil(total / IDHMC_POOL_SEGMENT)); the total comes from an exact all-reduce
        // of the chain counts (shards tile [0, total)), a lone context covers just its own segments.
        const DevState &s = c->s;
        long long seg_lo = (long long)s.first_chain / IDHMC_POOL_SEGMENT;
        long long seg_hi = ((long long)s.first_chain + s.C + IDHMC_POOL_SEGMENT - 1) / IDHMC_POOL_SEGMENT;
        char err[200];
        if (c->comm) {
            double cnt[IDHMC_XCHG_DOUBLES] = {0.0, 0.0, (double)s.C, 0.0};
            HIPCHK(hipMemcpyAsync(c->xchg, cnt, sizeof cnt, hipMemcpyHostToDevice, c->stream));
            if (comm_allreduce_sum(c->comm, c->xchg, IDHMC_XCHG_DOUBLES, c->stream, err, sizeof err)) return fail(IDHMC_ERR_HIP, "%s", err);
            if (int rc = get_scalar(c, cnt, c->xchg, sizeof cnt)) return rc;
            seg_lo = 0;
            seg_hi = ((long long)cnt[2] + IDHMC_POOL_SEGMENT - 1) / IDHMC_POOL_SEGMENT;
        }
        const long long nseg = seg_hi - seg_lo;
        if (int rc = pool_table_reserve(c, nseg)) return rc;
        for (int pass = 0; pass < 2; ++pass) {
            HIPCHK(launch_pool_partials(s, pass, c->pool_scratch, c->pool_table, seg_lo, seg_hi, c->stream));
            if (c->comm && comm_allreduce_sum(c->comm, c->pool_table, (int)(nseg * (s.L + 1)), c->stream, err, sizeof err))
                return fail(IDHMC_ERR_HIP, "%s", err);
            HIPCHK(launch_pool_consume(s, pass, c->pool_scratch, c->pool_table, nseg, lambda, c->stream));
        }
        return IDHMC_OK;
    }
    HIPCHK(launch_metric_update(c->s, lambda, c->stream));
    return IDHMC_OK;
}
int idhmc_moments_reset(idhmc_ctx *c)
{
    CTXCHK(c);
    DevState &s = c->s;
    const int64_t CL = s.C * s.L;
    if (!s.mom_mean) {
        if (int rc = dalloc(c, &s.mom_mean, CL)) return rc;
        if (int rc = dalloc(c, &s.mom_m2, CL)) return rc;
        if (int rc = dalloc(c, &s.mom_n, s.C)) return rc;
    } else {
        HIPCHK(hipMemsetAsync(s.mom_mean, 0, sizeof(double) * CL, c->stream));
        HIPCHK(hipMemsetAsync(s.mom_m2, 0, sizeof(double) * CL, c->stream));
        HIPCHK(hipMemsetAsync(s.mom_n, 0, sizeof(int64_t) * s.C, c->stream));
    }
    return IDHMC_OK;
}
int idhmc_get_moments(idhmc_ctx *c, double *mean, double *var, int64_t *count)
{
    CTXCHK(c);
    DevState &s = c->s;
    if (!s.mom_mean) return fail(IDHMC_ERR_BAD_ARG, "no moments accumulated");
    const int64_t CL = s.C * s.L;
    if (!c->scratch) { if (int rc = dalloc(c, &c->scratch, 2 * CL)) return rc; }
    HIPCHK(launch_moments_get(s, c->scratch, c->scratch + CL, c->stream));
    if (mean) { if (int rc = get_vec(c, mean, c->scratch, s.C)) return rc; }
    if (var) { if (int rc = get_vec(c, var, c->scratch + CL, s.C)) return rc; }
    if (count) { if (int rc = get_scalar(c, count, s.mom_n, sizeof(int64_t) * s.C)) return rc; }
    return IDHMC_OK;
}
// ---- diagnostics reduced on the device (src/diagnostics.jl:28-32, 61-101) -----------------------------------
int idhmc_diag_reset(idhmc_ctx *c)
{
    CTXCHK(c);
    DevState &s = c->s;
    if (!s.diag.n) {
        if (int rc = dalloc(c, &s.diag.n, s.C)) return rc;
        if (int rc = dalloc(c, &s.diag.pi1, s.C)) return rc;
        if (int rc = dalloc(c, &s.diag.prev, s.C)) return rc;
        if (int rc = dalloc(c, &s.diag.s1, s.C)) return rc;
        if (int rc = dalloc(c, &s.diag.s2, s.C)) return rc;
        if (int rc = dalloc(c, &s.diag.d2, s.C)) return rc;
        if (int rc = dalloc(c, &s.diag.counters, (int64_t)IDHMC_DIAG_COUNTERS)) return rc;
        if (int rc = dalloc(c, &c->ebfmi_out, s.C)) return rc;
    } else {
        HIPCHK(hipMemsetAsync(s.diag.n, 0, sizeof(int32_t) * s.C, c->stream));
        HIPCHK(hipMemsetAsync(s.diag.counters, 0, sizeof(unsigned long long) * IDHMC_DIAG_COUNTERS, c->stream));
    }
    return IDHMC_OK;
}
int idhmc_get_diag_counters(idhmc_ctx *c, uint64_t *counters)
{
    CTXCHK(c);
    if (!counters) return fail(IDHMC_ERR_BAD_ARG, "null out");
    if (!c->s.diag.n) return fail(IDHMC_ERR_BAD_ARG, "no diagnostics accumulated (idhmc_diag_reset first)");
    return get_scalar(c, counters, c->s.diag.counters, sizeof(uint64_t) * IDHMC_DIAG_COUNTERS);
}
int idhmc_tree_summary_from_counters(const uint64_t *cn, idhmc_tree_summary *out)
{
    if (!cn || !out) return fail(IDHMC_ERR_BAD_ARG, "null argument");
    memset(out, 0, sizeof *out);
    const uint64_t N = cn[0];
    out->N = (int64_t)N;
    out->max_depth = (int64_t)cn[3]; out->divergence = (int64_t)cn[4]; out->turning = (int64_t)cn[5];
    for (int d = 0; d < 33; ++d) out->depth_counts[d] = (int64_t)cn[6 + d];
    if (N == 0) return IDHMC_OK;
    out->a_mean = xchg_mean(IDHMC_XCHG_ACCEPT, (double)(int64_t)cn[1], (double)cn[2], (double)N);
    // sample quantile (linear interpolation between order statistics, position q (N - 1)) located in the histogram,
    // order statistics taken as equally spaced inside their bin
    static const double qs[5] = {0.05, 0.25, 0.5, 0.75, 0.95};       // ACCEPTANCE_QUANTILES, src/diagnostics.jl:35
    for (int k = 0; k < 5; ++k) {
        const double pos = qs[k] * (double)(N - 1);
        uint64_t below = 0;
        double val = 1.0;
        for (int b = 0; b < IDHMC_DIAG_ACC_BINS; ++b) {
            const uint64_t nb = cn[39 + b];
            if (nb && pos < (double)(below + nb)) {
                val = ((double)b + (pos - (double)below + 0.5) / (double)nb) / (double)IDHMC_DIAG_ACC_BINS;
                break;
            }
            below += nb;
        }
        out->a_quantiles[k] = val;
    }
    return IDHMC_OK;
}
int idhmc_get_ebfmi(idhmc_ctx *c, double *ebfmi)
{
    CTXCHK(c);
    if (!ebfmi) return fail(IDHMC_ERR_BAD_ARG, "null out");
    if (!c->s.diag.n) return fail(IDHMC_ERR_BAD_ARG, "no diagnostics accumulated (idhmc_diag_reset first)");
    HIPCHK(launch_ebfmi(c->s, c->ebfmi_out, c->stream));
    return get_scalar(c, ebfmi, c->ebfmi_out, sizeof(double) * c->s.C);
}

int idhmc_total_steps(idhmc_ctx *c, int64_t *steps)
{
    CTXCHK(c);
    if (!steps) return fail(IDHMC_ERR_BAD_ARG, "null out");
    unsigned long long v = 0;
    if (int rc = get_scalar(c, &v, c->s.total_steps, sizeof v)) return rc;
    *steps = (int64_t)v;
    return IDHMC_OK;
}

int idhmc_debug_counters(idhmc_ctx *c, uint64_t *out32)
{
    CTXCHK(c);
    if (!out32) return fail(IDHMC_ERR_BAD_ARG, "null out");
    return get_scalar(c, out32, c->s.total_steps, sizeof(uint64_t) * 32);
}

// ---- the reference's caller loops --------------------------------------------------------------------
static int one_transition(idhmc_ctx *c, uint32_t iter, uint32_t flags, int adapt)
{
    if (adapt && c->s.eps_mode == IDHMC_EPS_PER_CHAIN) flags |= IDHMC_T_ADAPT_EPS;
    if (int rc = idhmc_nuts_transition(c, iter, flags)) return rc;
    if (adapt && c->s.eps_mode == IDHMC_EPS_GLOBAL) {
        double *buf = xchg_buf(c);
        HIPCHK(launch_xchg_sum(c->s, IDHMC_XCHG_ACCEPT, buf, c->stream));
        if (int rc = exchange(c, buf)) return rc;
        HIPCHK(launch_da_adapt_global(c->s, buf, c->stream));
    }
    return IDHMC_OK;
}
// ---- draws and records to the host, overlapped with the next transition ------------------------------------------------
// fetch_pack(n) is enqueued right behind transition n: the device packs the draw (padded rows -> contiguous) and the records
// into staging buffer n & 1.  fetch_copy(n) is called AFTER transition n + 1 has been enqueued: it waits for the pack and
// copies to the caller's (pageable) arrays on a second stream -- the host blocks in that copy while the device computes.
// Buffer n & 1 is reused by pack(n + 2), which is enqueued after copy(n) has returned.
static int fetch_setup(idhmc_ctx *c, bool draws, bool stats)
{
    const DevState &s = c->s;
    if (!c->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) HIPCHK(hipEventCreateWithFlags(&c->ev_packed[b], hipEventDisableTiming));
    }
    for (int b = 0; b < 2; ++b) {
        if (draws && !c->stage_q[b]) { if (int rc = dalloc(c, &c->stage_q[b], s.C * (int64_t)s.D, false)) return rc; }
        if (stats && !c->stage_st[b]) { if (int rc = dalloc(c, &c->stage_st[b], s.C, false)) return rc; }
    }
    return IDHMC_OK;
}
static int fetch_pack(idhmc_ctx *c, int32_t n, bool draws, bool stats)
{
    if (!draws && !stats) return IDHMC_OK;
    const int b = n & 1;
    HIPCHK(launch_pack_draw(c->s, draws ? c->stage_q[b] : nullptr, stats ? c->stage_st[b] : nullptr, c->stream));
    HIPCHK(hipEventRecord(c->ev_packed[b], c->stream));
    return IDHMC_OK;
}
static int fetch_copy(idhmc_ctx *c, int32_t n, double *draws, idhmc_tree_stats *stats)
{
    if (!draws && !stats) return IDHMC_OK;
    const DevState &s = c->s;
    const int b = n & 1;
    HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_packed[b], 0));
    if (draws) HIPCHK(hipMemcpyAsync(draws + (int64_t)n * s.C * s.D, c->stage_q[b], sizeof(double) * s.C * s.D, hipMemcpyDeviceToHost, c->copy_stream));
    if (stats) HIPCHK(hipMemcpyAsync(stats + (int64_t)n * s.C, c->stage_st[b], sizeof(idhmc_tree_stats) * s.C, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    return IDHMC_OK;
}

// Draws and records for the host with several transitions per launch: the kernel writes every transition's draw and record into a
// staging block of K transitions; block j is copied out (second stream, the host blocks in that copy) while block j + 1 computes, two
// blocks alternating.  K = what fits 256 MiB per block (at most 64; larger blocks gain nothing: with the draws kept the loop is bound
// by the copy into the caller's pageable array, 10-22 GB/s); a draw of more than half a block keeps the per-transition path (K = 0).
static int32_t block_transitions(const idhmc_ctx *c, int32_t N, bool any)
{
    if (!any || !fuse_transitions(c) || N < 2) return 0;
    const int64_t per = c->s.C * (int64_t)c->s.D * (int64_t)sizeof(double) + c->s.C * (int64_t)sizeof(idhmc_tree_stats);
    const int64_t k = ((int64_t)256 << 20) / per;
    int32_t K = (int32_t)(k > N ? N : k);
    if (K > 64) K = 64;
    if (K < 2 || (uint64_t)c->s.C * (uint64_t)K >= (1ull << 31)) K = 0;
    return K;
}
static int run_blocks(idhmc_ctx *c, uint32_t iter_first, int32_t N, uint32_t fl, int32_t K, double *draws, idhmc_tree_stats *stats)
{
    for (int b = 0; b < 2; ++b) {       // (grow-only; the per-transition path uses the same buffers)
        if (draws && (!c->stage_q[b] || c->stage_kq < K)) {
            dfree(c, c->stage_q[b], (int64_t)sizeof(double) * c->stage_kq * c->s.C * c->s.D);
            c->stage_q[b] = nullptr;
            if (int rc = dalloc(c, &c->stage_q[b], (int64_t)K * c->s.C * c->s.D, false)) return rc;
        }
        if (stats && (!c->stage_st[b] || c->stage_kst < K)) {
            dfree(c, c->stage_st[b], (int64_t)sizeof(idhmc_tree_stats) * c->stage_kst * c->s.C);
            c->stage_st[b] = nullptr;
            if (int rc = dalloc(c, &c->stage_st[b], (int64_t)K * c->s.C, false)) return rc;
        }
    }
    if (draws && c->stage_kq < K) c->stage_kq = K;
    if (stats && c->stage_kst < K) c->stage_kst = K;
    const int64_t CD = c->s.C * (int64_t)c->s.D;
    int32_t prev_n0 = -1, prev_cnt = 0;
    auto copy_block = [&](int32_t n0, int32_t cnt, int b) -> int {
        HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_packed[b], 0));
        if (draws) HIPCHK(hipMemcpyAsync(draws + (int64_t)n0 * CD, c->stage_q[b], sizeof(double) * (size_t)(cnt * CD), hipMemcpyDeviceToHost, c->copy_stream));
        if (stats) HIPCHK(hipMemcpyAsync(stats + (int64_t)n0 * c->s.C, c->stage_st[b], sizeof(idhmc_tree_stats) * (size_t)(cnt * c->s.C), hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(hipStreamSynchronize(c->copy_stream));
        return IDHMC_OK;
    };
    int blk = 0;
    for (int32_t n0 = 0; n0 < N; n0 += K, ++blk) {
        const int32_t cnt = N - n0 < K ? N - n0 : K;
        const int b = blk & 1;
        if (int rc = nuts_launch(c, iter_first + (uint32_t)n0, fl, (uint32_t)cnt, draws ? c->stage_q[b] : nullptr, stats ? c->stage_st[b] : nullptr)) return rc;
        HIPCHK(hipEventRecord(c->ev_packed[b], c->stream));
        if (prev_n0 >= 0) { if (int rc = copy_block(prev_n0, prev_cnt, b ^ 1)) return rc; }      // ... while block blk computes
        prev_n0 = n0; prev_cnt = cnt;
    }
    if (prev_n0 >= 0) { if (int rc = copy_block(prev_n0, prev_cnt, (blk - 1) & 1)) return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));
    return IDHMC_OK;
}
int idhmc_tuning_stage(idhmc_ctx *c, int32_t N, int32_t adapt_metric, uint32_t iter0, double *draws, idhmc_tree_stats *stats)
{
    CTXCHK(c);
    if (N < 1) return fail(IDHMC_ERR_BAD_ARG, "N must be >= 1");
    if (adapt_metric && !c->s.mw_x1) return fail(IDHMC_ERR_BAD_ARG, "shared-metric context cannot adapt the metric");
    if (adapt_metric && N < 2) return fail(IDHMC_ERR_BAD_ARG, "metric window needs N >= 2");
    if (int rc = idhmc_da_init(c)) return rc;                                    // src/warmup.jl:284
    if (adapt_metric) { if (int rc = idhmc_metric_begin(c)) return rc; }
    const double lambda = 5.0 / (double)N;                                       // src/warmup.jl:229
    if (draws || stats) { if (int rc = fetch_setup(c, draws != nullptr, stats != nullptr)) return rc; }
    int32_t done = 0;
    if (int32_t K = (c->s.eps_mode != IDHMC_EPS_GLOBAL) ? block_transitions(c, N, draws || stats) : 0) {
        // the stage's draws / records leave in blocks of K transitions
        const uint32_t fl = (adapt_metric ? IDHMC_T_ACCUM_METRIC : 0u) | (c->s.eps_mode == IDHMC_EPS_PER_CHAIN ? IDHMC_T_ADAPT_EPS : 0u);
        if (int rc = run_blocks(c, iter0 + 1u, N, fl, K, draws, stats)) return rc;
    } else if (!draws && !stats && c->s.eps_mode != IDHMC_EPS_GLOBAL && fuse_transitions(c) && N > 1) {
        // nothing leaves the device per transition: the whole stage is one launch (the kernel itself stops handing out
        // transitions once a chain has raised the abort code)
        const uint32_t fl = (adapt_metric ? IDHMC_T_ACCUM_METRIC : 0u) | (c->s.eps_mode == IDHMC_EPS_PER_CHAIN ? IDHMC_T_ADAPT_EPS : 0u);
        if (int rc = idhmc_nuts_transitions(c, iter0 + 1u, N, fl)) return rc;
        (void)pulse_abort(c, 0);        // (waits for the launch: the stage's verdict is agreed on below)
    } else
    for (int32_t n = 0; n < N; ++n) {                                            // :288-305
        // the reference throws as soon as eps < 1e-10 (:291-296): stop within kLag transitions of the one that set it
        if (pulse_abort(c, idhmc_ctx::kLag)) break;
        if (int rc = one_transition(c, iter0 + 1u + (uint32_t)n, adapt_metric ? IDHMC_T_ACCUM_METRIC : 0u, 1)) return rc;
        if (int rc = fetch_pack(c, n, draws != nullptr, stats != nullptr)) return rc;
        if (n > 0) { if (int rc = fetch_copy(c, n - 1, draws, stats)) return rc; }   // ... while transition n computes
        done = n + 1;
    }
    if (done > 0) { if (int rc = fetch_copy(c, done - 1, draws, stats)) return rc; }
    // sharded: agree on the outcome first -- a rank that failed alone would leave the others in the pooled metric's all-reduces
    if (int rc = status_exchange(c, "warmup")) return rc;
    if (adapt_metric) { if (int rc = idhmc_metric_update(c, lambda)) return rc; } // :308-311
    return idhmc_da_finalize(c);                                                 // :313
}
int idhmc_mcmc(idhmc_ctx *c, int32_t N, uint32_t iter0, double *draws, idhmc_tree_stats *stats)
{
    CTXCHK(c);
    if (N < 0) return fail(IDHMC_ERR_BAD_ARG, "N must be >= 0");
    if (draws || stats) { if (int rc = fetch_setup(c, draws != nullptr, stats != nullptr)) return rc; }
    if (const int32_t K = block_transitions(c, N, draws || stats)) {
        const uint32_t fl = (c->s.mom_mean ? IDHMC_T_ACCUM_MOMENTS : 0u) | (c->s.diag.n ? IDHMC_T_ACCUM_DIAG : 0u);
        return run_blocks(c, iter0 + 1u, N, fl, K, draws, stats);
    }
    if (!draws && !stats && fuse_transitions(c) && N > 1) {
        const uint32_t fl = (c->s.mom_mean ? IDHMC_T_ACCUM_MOMENTS : 0u) | (c->s.diag.n ? IDHMC_T_ACCUM_DIAG : 0u);
        if (int rc = idhmc_nuts_transitions(c, iter0 + 1u, N, fl)) return rc;
    } else
    for (int32_t n = 0; n < N; ++n) {                                            // src/warmup.jl:324-330
        const uint32_t fl = (c->s.mom_mean ? IDHMC_T_ACCUM_MOMENTS : 0u) | (c->s.diag.n ? IDHMC_T_ACCUM_DIAG : 0u);
        if (int rc = one_transition(c, iter0 + 1u + (uint32_t)n, fl, 0)) return rc;
        if (int rc = fetch_pack(c, n, draws != nullptr, stats != nullptr)) return rc;
        if (n > 0) { if (int rc = fetch_copy(c, n - 1, draws, stats)) return rc; }   // ... while transition n computes
    }
    if (N > 0) { if (int rc = fetch_copy(c, N - 1, draws, stats)) return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));
    return IDHMC_OK;
}
int idhmc_mcmc_with_warmup(idhmc_ctx *c, int32_t N, double *draws, idhmc_tree_stats *stats)
{
    CTXCHK(c);
    const idhmc_options &o = c->opt;
    uint32_t iter = 0;
    if (int rc = idhmc_random_position(c)) return rc;                            // initialize_warmup_state, src/warmup.jl:100-129
    if (o.local_opt_iterations > 0) {                                            // FindLocalOptimum, src/warmup.jl:152-186
        if (int rc = idhmc_find_local_optimum(c, o.local_opt_penalty, o.local_opt_iterations)) return rc;
    }
    if (int rc = idhmc_set_eps(c, o.eps_init)) return rc;
    if (o.stepsize_search) {                                                     // src/warmup.jl:188-200
        if (int rc = idhmc_refresh_momentum(c, 0)) return rc;
        if (int rc = idhmc_find_initial_stepsize(c)) return rc;
    }
    const int adapt = o.adapt_metric && c->s.mw_x1;
    if (int rc = idhmc_tuning_stage(c, o.init_steps, 0, iter, nullptr, nullptr)) return rc;           // src/warmup.jl:369
    iter += (uint32_t)o.init_steps;
    for (int d = 0; d < o.doubling_stages; ++d) {                                                     // :341-344
        const int32_t n = o.middle_steps << d;
        if (int rc = idhmc_tuning_stage(c, n, adapt, iter, nullptr, nullptr)) return rc;
        iter += (uint32_t)n;
    }
    if (int rc = idhmc_tuning_stage(c, o.terminating_steps, 0, iter, nullptr, nullptr)) return rc;    // :371
    iter += (uint32_t)o.terminating_steps;
    return idhmc_mcmc(c, N, iter, draws, stats);                                                      // src/mcmc.jl:104
}

// ---- measurement helpers ----------------------------------------------------------------------------
int idhmc_time_leapfrog(idhmc_ctx *c, double eps, int32_t sweeps, float *ms_per_sweep)
{
    CTXCHK(c);
    if (sweeps < 1 || !ms_per_sweep) return fail(IDHMC_ERR_BAD_ARG, "bad arguments");
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    const int regrad = leapfrog_regrad(c, 1);
    if (!regrad) { if (int rc = ensure_grad(c)) return rc; }
    for (int i = 0; i < sweeps; ++i) { if (int rc = leapfrog_any(c, eps, 0, 1, regrad)) return rc; }
    if (regrad) c->grad_stale = true;
    if (int rc = lanes_join(c)) return rc;
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_per_sweep = ms / (float)sweeps;
    return IDHMC_OK;
}
int idhmc_time_transitions(idhmc_ctx *c, int32_t n, uint32_t iter0, float *ms_total)
{
    CTXCHK(c);
    if (n < 1 || !ms_total) return fail(IDHMC_ERR_BAD_ARG, "bad arguments");
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < n; ++i) { if (int rc = idhmc_nuts_transition(c, iter0 + 1u + (uint32_t)i, 0u)) return rc; }
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms_total, c->ev0, c->ev1));
    return IDHMC_OK;
}

int idhmc_time_transitions_fused(idhmc_ctx *c, int32_t n, uint32_t iter0, float *ms_total)
{
    CTXCHK(c);
    if (n < 1 || !ms_total) return fail(IDHMC_ERR_BAD_ARG, "bad arguments");
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    if (int rc = idhmc_nuts_transitions(c, iter0 + 1u, n, 0u)) return rc;
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms_total, c->ev0, c->ev1));
    return IDHMC_OK;
}

}  // extern "C"
