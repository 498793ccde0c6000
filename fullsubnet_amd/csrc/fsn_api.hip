// C ABI of libfsn_hip.so (see include/fsn_hip.h): argument validation, workspace carving and the
// kernel sequence of the FullSubNet enhancement path.  No allocation and no host synchronisation on the hot
// path.  The only state the library owns is a small per-(device, caller stream) record - an auxiliary
// stream with its fork / join events for the left-over sub-band tiles and the events of the optional
// per-stage profiler - created on first use and never shared between two caller streams or two devices.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <atomic>
#include <map>
#include <string>
#include <vector>
#include <mutex>
#include <utility>

#include "fsn_common.h"

// ---- errors ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void fsn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fsn_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fsn_set_error("%s: %s", what, hipGetErrorString(e));
        return FSN_ERR_LAUNCH;
    }
    return FSN_OK;
}
extern "C" const char* fsn_last_error(void) { return g_err; }
extern "C" int fsn_version(void) { return FSN_ABI_VERSION; }

#define FSN_TRY(x)                \
    do {                          \
        const int _rc = (x);      \
        if (_rc != FSN_OK) return _rc; \
    } while (0)

// ---- per-stage profiler ----------------------------------------------------------------------
enum Stage {
    ST_STFT = 0,
    ST_NORM,
    ST_FB_GEMM,
    ST_FB_REC,
    ST_SB_GEMM_L0,
    ST_SB_REC_L0,
    ST_SB_GEMM_L1,
    ST_SB_REC_L1,
    ST_SB_FC,
    ST_MASK_ISTFT,
    ST_COUNT
};
static const char* kStageNames[ST_COUNT] = {"stft",       "norm",       "fb_gemm",    "fb_rec", "sb_gemm_l0",
                                            "sb_rec_l0",  "sb_gemm_l1", "sb_rec_l1",  "sb_fc",  "mask_istft"};
constexpr int kMaxSpans = 4;  // a stage may be entered several times per call (once per layer)

// ---- per-(device, caller stream) state ---------------------------------------------------------------
// Everything a call needs beyond its arguments.  Two caller streams (or two devices, or two host threads that
// each drive their own stream) never see each other's events; calls that share ONE stream must be issued
// from one thread at a time, like any stream-ordered API.
struct StreamCtx {
    int dev = 0;
    hipStream_t aux = nullptr;           // left-over sub-band tiles beside the persistent kernel
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool prof_on = false;                // per-stage profiler requested for this stream (fsn_profile_enable_stream)
    bool prof_events = false;            // profiler events exist
    hipEvent_t ev[ST_COUNT][kMaxSpans][2];
    int spans[ST_COUNT] = {0};
    // sticky status record of the persistent kernels launched on this stream: pinned host memory the device writes
    // ({status word of the first launch that ran out of time, number of such launches}); NULL until first needed
    unsigned* sticky_host = nullptr;
    unsigned* sticky_dev = nullptr;
    // fsn_stream_timeout_policy: a raised record does not refuse later persistent launches on this stream (a training
    // step in flight: NaN poison + the optimizer's skip contain the damage; the caller looks at the end of the step)
    bool timeout_defer = false;
};
static std::mutex g_ctx_mutex;
static std::map<std::pair<int, hipStream_t>, StreamCtx*> g_ctx;
static std::atomic<int> g_persist_mode{0};          // fsn_set_persistent_mode: 0 auto, 1 never
static std::atomic<int> g_g16_off{0};               // fsn_debug_g16_kernels(0): the fp32-era group kernels also under 16-bit arithmetic
static std::atomic<int> g_in16_off{0};              // fsn_debug_g16_kernels(3): dx / dW_ih0 from the fp32 gate gradients (round 5's form)
static std::atomic<int> g_tn16h_off{0};             // fsn_debug_g16_kernels(2): ... only the weight-gradient products of round 3
static std::atomic<int> g_persist_timeout_ms{20000};  // fsn_set_persistent_timeout_ms

// Persistent kernels whose workgroups wait for each other (the group kernel, the full-band chain) need ALL their
// workgroups resident at once.  Two of them launched from different streams could each take a part of the chip and wait
// for the rest until the spin bound.  So these launches are admitted against a residency budget per device: every
// launcher reports its kernel's footprint (fsn_persist_admit: grid, resident workgroups per CU by the occupancy API)
// right before the launch, and the launch waits for the completion events of earlier persistent launches on OTHER
// streams, oldest first, until the set that may run beside it is provably placeable whatever order the dispatcher
// hands out workgroups in:
//   a CU holding a_j workgroups of kernel j is "used" u = sum_j a_j / occ_j in the model (occ_j = resident workgroups
//   per CU of kernel j alone; the model is conservative: where it has room for a workgroup, the hardware has);
//   a workgroup of kernel k finds no CU only if EVERY CU has u > 1 - 1 / occ_k, i.e. u >= umin(k), the smallest sum of
//   the set's workgroup sizes above that threshold; all CUs together then hold >= CUs x umin(k), while the kernels of
//   the set can place at most sum_j grid_j / occ_j = CUs x sum_j frac_j.  Hence: admitted iff sum_j frac_j < umin(k)
//   for every k of the set.
// One kernel alone is always admitted (its own grid was checked against occ x CUs at plan time).  Examples on 256 CUs:
// two chain launches of H = 384 with two row tiles (192 workgroups, 2 per CU each) run side by side, a third waits; two
// group launches of 28 clusters (448 workgroups, 2 per CU) do not.  Nothing else of the streams is ordered.
// Under stream capture (round 6) the same rule orders the persistent launches of ONE capture among themselves, through
// captured event edges: a graph's replay runs the captured streams' kernels side by side exactly as far as the edges allow, and
// a call with several persistent launches on side streams (Improved FullSubNet's band sections at a few utterances: chain
// launches of one workgroup per CU) replayed without them stalled or ran out of time in ~1 % of the replays
// (tools/diag_stall.py).  Nothing can be retired during a capture: every earlier launch of the capture counts as live.  The
// replays of DIFFERENT graphs (and eager calls beside them) are ordered by whoever launches them.
struct PersistEntry {
    hipEvent_t ev;
    hipStream_t stream;
    double frac;  // grid / (occ x CUs)
    int occ;
};
struct PersistGate {
    std::vector<PersistEntry> live;
    std::vector<hipEvent_t> pool;
};
static std::mutex g_persist_mutex;
static std::map<int, PersistGate> g_persist;
static std::map<std::pair<int, unsigned long long>, PersistGate> g_persist_capture;  // (device, capture id) -> the capture's launches
// umin(k) in units of 1 / 840 (= lcm(1 .. 8); occupancies above 8 count as 8, which only makes workgroups larger)
static int persist_umin(const std::vector<int>& occs, int occ_k) {
    bool reach[841] = {};
    reach[0] = true;
    for (int o : occs) {
        const int sz = 840 / o;
        for (int u = sz; u <= 840; ++u)
            if (reach[u - sz]) reach[u] = true;  // unbounded multiples, ascending
    }
    for (int u = 840 - 840 / occ_k + 1; u <= 840; ++u)
        if (reach[u]) return u;
    return 1 << 20;  // no CU state blocks kernel k
}
static bool persist_set_fits(const std::vector<const PersistEntry*>& set) {
    std::vector<int> occs;
    double sum = 0.0;
    for (const PersistEntry* e : set) {
        occs.push_back(e->occ);
        sum += e->frac;
    }
    for (const PersistEntry* e : set)
        if (!(sum * 840.0 < (double)persist_umin(occs, e->occ))) return false;
    return true;
}
class PersistLaunch;
static thread_local PersistLaunch* t_persist = nullptr;
static std::atomic<unsigned> g_persist_launches{0}, g_persist_waits{0}, g_persist_unreported{0};
class PersistLaunch {
  public:
    explicit PersistLaunch(hipStream_t s) : s_(s), lock_(g_persist_mutex) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        unsigned long long cap_id = 0;
        if (hipStreamGetCaptureInfo(s, &cap, &cap_id) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (cap == hipStreamCaptureStatusActive) {
            // the capture's own gate: its launches, ordered by captured event edges; events come from (and return to) the
            // device's pool; the gates of finished captures are dropped once a few newer ones exist
            pool_ = &g_persist[dev].pool;
            const std::pair<int, unsigned long long> key(dev, cap_id);
            if (!g_persist_capture.count(key)) {
                while (g_persist_capture.size() >= 4) {
                    auto old = g_persist_capture.begin();  // smallest (device, id): an earlier capture
                    if (old->first.first == dev)
                        for (PersistEntry& e : old->second.live) g_persist[dev].pool.push_back(e.ev);
                    g_persist_capture.erase(old);
                }
            }
            gate_ = &g_persist_capture[key];
            t_persist = this;
            return;
        }
        if (cap != hipStreamCaptureStatusNone) return;  // an invalidated capture: nothing to order
        gate_ = &g_persist[dev];
        pool_ = &gate_->pool;
        // retire what has completed
        std::vector<PersistEntry>& live = gate_->live;
        for (size_t i = 0; i < live.size();) {
            if (hipEventQuery(live[i].ev) == hipSuccess) {
                gate_->pool.push_back(live[i].ev);
                live.erase(live.begin() + (long)i);
            } else {
                (void)hipGetLastError();
                ++i;
            }
        }
        t_persist = this;
    }
    // the launcher's report, right before its launch (fsn_persist_admit)
    void admit(double frac, int occ) {
        if (!gate_) return;
        me_.frac = frac;
        me_.occ = occ < 1 ? 1 : occ > 8 ? 8 : occ;
        admitted_ = true;
        std::vector<const PersistEntry*> set;
        for (const PersistEntry& e : gate_->live)
            if (e.stream != s_) set.push_back(&e);  // same stream: ordered anyway
        set.push_back(&me_);
        while (set.size() > 1 && !persist_set_fits(set)) {
            (void)hipStreamWaitEvent(s_, set.front()->ev, 0);  // oldest first
            set.erase(set.begin());
            g_persist_waits.fetch_add(1, std::memory_order_relaxed);
        }
        g_persist_launches.fetch_add(1, std::memory_order_relaxed);
        static const bool trace = getenv("FSN_TRACE_GATE") != nullptr;  // diagnostics only
        if (trace)
            fprintf(stderr, "libfsn_hip gate: stream %p frac %.3f occ %d admitted beside %zu launch(es) of other streams\n",
                    (void*)s_, frac, me_.occ, set.size() - 1);
    }
    ~PersistLaunch() {
        t_persist = nullptr;
        if (!gate_) return;
        hipEvent_t ev = nullptr;
        const bool capturing = pool_ != &gate_->pool;
        if (!pool_->empty()) {
            ev = pool_->back();
            pool_->pop_back();
        } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (!capturing) (void)hipStreamSynchronize(s_);  // no event to order later launches by: drain instead
            return;
        }
        if (hipEventRecord(ev, s_) != hipSuccess) {
            (void)hipGetLastError();
            pool_->push_back(ev);
            if (!capturing) (void)hipStreamSynchronize(s_);
            return;
        }
        me_.ev = ev;
        me_.stream = s_;
        if (!admitted_) {  // a launcher that did not report (a bug, counted): treated as filling the chip from now on
            me_.frac = 1.0;
            me_.occ = 1;
            g_persist_unreported.fetch_add(1, std::memory_order_relaxed);
        }
        gate_->live.push_back(me_);
    }
    PersistLaunch(const PersistLaunch&) = delete;
    PersistLaunch& operator=(const PersistLaunch&) = delete;
    bool admitted() const { return admitted_ || !gate_; }

  private:
    hipStream_t s_;
    std::unique_lock<std::mutex> lock_;
    PersistGate* gate_ = nullptr;
    std::vector<hipEvent_t>* pool_ = nullptr;  // where events come from: the device's pool (a capture's gate has none of its own)
    PersistEntry me_{};
    bool admitted_ = false;
};

// What the running call works on (set by CallScope for the duration of one entry point on this host thread).
static thread_local StreamCtx* t_ctx = nullptr;
static thread_local hipStream_t t_stream = nullptr;
static thread_local int t_dev = 0;

static StreamCtx* ctx_lookup(int dev, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    StreamCtx*& c = g_ctx[std::make_pair(dev, s)];
    if (!c) {
        c = new StreamCtx();
        c->dev = dev;
    }
    return c;
}
static StreamCtx* cur_ctx() {
    if (!t_ctx) t_ctx = ctx_lookup(t_dev, t_stream);
    return t_ctx;
}

// Every entry point that enqueues work opens one of these: the device the caller's stream belongs to becomes
// the current device for the duration of the call (restored afterwards), so that a process that drives several
// GPUs needs no device bookkeeping around the C ABI, and the per-stream record is resolved lazily.
FsnCallScope::FsnCallScope(void* stream) : prev(-1), switched(false) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    int cur = 0;
    (void)hipGetDevice(&cur);
    hipDevice_t sd = cur;
    if (s && hipStreamGetDevice(s, &sd) != hipSuccess) {
        (void)hipGetLastError();
        sd = cur;
    }
    if ((int)sd != cur && hipSetDevice((int)sd) == hipSuccess) {
        prev = cur;
        switched = true;
    }
    t_dev = (int)sd;
    t_stream = s;
    t_ctx = nullptr;
}
FsnCallScope::~FsnCallScope() {
    t_ctx = nullptr;
    if (switched) (void)hipSetDevice(prev);
}
typedef FsnCallScope CallScope;

struct StageTimer {
    int st, span;
    hipStream_t s;
    StreamCtx* c;
    StageTimer(int stage, hipStream_t stream) : st(stage), span(-1), s(stream), c(nullptr) {
        c = cur_ctx();
        if (!c->prof_on) {
            c = nullptr;
            return;
        }
        if (!c->prof_events) {
            for (int i = 0; i < ST_COUNT; ++i)
                for (int j = 0; j < kMaxSpans; ++j) {
                    (void)hipEventCreate(&c->ev[i][j][0]);
                    (void)hipEventCreate(&c->ev[i][j][1]);
                }
            c->prof_events = true;
        }
        if (c->spans[st] >= kMaxSpans) return;
        span = c->spans[st];
        (void)hipEventRecord(c->ev[st][span][0], s);
    }
    ~StageTimer() {
        if (span < 0) return;
        (void)hipEventRecord(c->ev[st][span][1], s);
        c->spans[st] = span + 1;
    }
};
static void prof_reset() {
    StreamCtx* c = cur_ctx();
    if (!c->prof_on) return;
    for (int i = 0; i < ST_COUNT; ++i) c->spans[i] = 0;
}

// ---- residency contract of the persistent kernels (fsn_common.h) -------------------------------------------------
bool fsn_persistent_allowed() { return g_persist_mode.load(std::memory_order_relaxed) == 0; }
unsigned long long fsn_spin_ticks() {
    int khz = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
        (void)hipGetLastError();
        khz = 100000;  // gfx9: the constant-rate counter runs at 100 MHz
    }
    return (unsigned long long)g_persist_timeout_ms.load(std::memory_order_relaxed) * (unsigned long long)khz;
}
// resident workgroups per CU of `kernel` alone (occupancy API, cached per device); 0 on failure
static int persist_occupancy(const void* kernel, int block_threads, int* cus_out) {
    static std::mutex m;
    static std::map<std::pair<const void*, int>, int> per_cu;  // (kernel, device) -> resident workgroups per CU
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (cus_out) *cus_out = cus;
    std::lock_guard<std::mutex> lock(m);
    auto it = per_cu.find(std::make_pair(kernel, dev));
    if (it == per_cu.end()) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block_threads, 0) != hipSuccess) {
            (void)hipGetLastError();
            n = 0;
        }
        it = per_cu.emplace(std::make_pair(kernel, dev), n).first;
    }
    return it->second;
}
bool fsn_grid_fits(const void* kernel, int block_threads, unsigned grid) {
    int cus = 0;
    const int occ = persist_occupancy(kernel, block_threads, &cus);
    return (unsigned long long)occ * (unsigned long long)cus >= grid;
}
// Called by every launcher of a kernel that needs its whole grid resident, right before the launch (inside the API
// function's FSN_PERSIST_BEGIN scope; a no-op outside one, e.g. under stream capture).
void fsn_persist_admit(const void* kernel, int block_threads, unsigned grid) {
    if (!t_persist) return;
    int cus = 0;
    const int occ = persist_occupancy(kernel, block_threads, &cus);
    if (occ < 1 || cus < 1) {
        t_persist->admit(1.0, 1);
        return;
    }
    t_persist->admit((double)grid / ((double)occ * (double)cus), occ);
}
// The sticky record of the running call's stream (created on first use; not under stream capture, where pinned
// allocations are not allowed: a captured launch then only poisons its outputs).
unsigned* fsn_ctx_sticky() {
    StreamCtx* c = cur_ctx();
    if (c->sticky_dev) return c->sticky_dev;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(t_stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    memset(h, 0, 64);
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(h);
        return nullptr;
    }
    c->sticky_host = static_cast<unsigned*>(h);
    c->sticky_dev = static_cast<unsigned*>(d);
    return c->sticky_dev;
}
// Called before every persistent launch: a stream on which such a launch ran out of time keeps failing until the
// caller has looked (fsn_stream_status) and cleared the record - garbage is never consumed silently.
static int persist_precheck() {
    StreamCtx* c = cur_ctx();
    if (c->sticky_host) {
        const unsigned st = __atomic_load_n(&c->sticky_host[0], __ATOMIC_ACQUIRE);
        if (st != 0 && !c->timeout_defer) {
            fsn_set_error("a persistent kernel launched earlier on this stream ran out of time waiting for its partner "
                          "workgroups (status %u, %u such launches): its outputs are NaN; see fsn_stream_status / "
                          "fsn_stream_status_clear", st, __atomic_load_n(&c->sticky_host[1], __ATOMIC_ACQUIRE));
            return FSN_ERR_TIMEOUT;
        }
    }
    (void)fsn_ctx_sticky();
    return FSN_OK;
}
#define FSN_PERSIST_BEGIN(s)        \
    FSN_TRY(persist_precheck());    \
    PersistLaunch gate(s)

extern "C" int fsn_set_persistent_mode(int mode) {
    FSN_REQUIRE(mode == FSN_PERSISTENT_AUTO || mode == FSN_PERSISTENT_NEVER, "persistent mode %d unknown", mode);
    g_persist_mode.store(mode, std::memory_order_relaxed);
    return FSN_OK;
}
// Test hooks without a device: the gate's admission rule on a hypothetical set (n launches, fractions of the chip and
// resident workgroups per CU), and the K-split plan of a weight-gradient product against the bound its scratch is sized by.
extern "C" int fsn_debug_persist_set_fits(int n, const double* fracs, const int* occs) {
    if (n < 1 || n > 64 || !fracs || !occs) return -1;
    std::vector<PersistEntry> e((size_t)n);
    std::vector<const PersistEntry*> set;
    for (int i = 0; i < n; ++i) {
        e[(size_t)i].frac = fracs[i];
        e[(size_t)i].occ = occs[i] < 1 ? 1 : occs[i] > 8 ? 8 : occs[i];
        set.push_back(&e[(size_t)i]);
    }
    return n == 1 || persist_set_fits(set) ? 1 : 0;
}
extern "C" int fsn_debug_tn_plan(int M, int Nc, long K, int arith, int* splits, long* bound) {
    if (M < 1 || Nc < 1 || K < 1) return -1;
    fsn_tn_plan_splits(M, Nc, K, arith, splits, bound);
    return 0;
}
// Test / measurement hook: 0 = the fp32-era group kernels also under the 16-bit training arithmetic (A/B against
// lstm_group16_kernels.hip), 1 (default) = the 16-bit arithmetic's own kernels where they apply.
void fsn_tn16h_wide(int on);
extern "C" int fsn_debug_tn16h_wide(int on) {
    fsn_tn16h_wide(on);
    return FSN_OK;
}
extern "C" int fsn_debug_g16_kernels(int on) {
    g_g16_off.store(on == 0 ? 1 : 0, std::memory_order_relaxed);
    g_tn16h_off.store(on == 2 ? 1 : 0, std::memory_order_relaxed);
    g_in16_off.store(on == 3 ? 1 : 0, std::memory_order_relaxed);
    return FSN_OK;
}
extern "C" int fsn_debug_persist_stats(unsigned* launches, unsigned* waits, unsigned* unreported) {
    if (launches) *launches = g_persist_launches.load(std::memory_order_relaxed);
    if (waits) *waits = g_persist_waits.load(std::memory_order_relaxed);
    if (unreported) *unreported = g_persist_unreported.load(std::memory_order_relaxed);
    return FSN_OK;
}
extern "C" int fsn_set_persistent_timeout_ms(int ms) {
    FSN_REQUIRE(ms >= 1 && ms <= 3600000, "timeout %d ms out of range [1, 3600000]", ms);
    g_persist_timeout_ms.store(ms, std::memory_order_relaxed);
    return FSN_OK;
}
extern "C" int fsn_stream_status(void* stream, int synchronize, unsigned* status_out, unsigned* events_out) {
    CallScope scope(stream);
    if (synchronize && hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) {
        fsn_set_error("fsn_stream_status: hipStreamSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
        return FSN_ERR_LAUNCH;
    }
    StreamCtx* c = cur_ctx();
    const unsigned st = c->sticky_host ? __atomic_load_n(&c->sticky_host[0], __ATOMIC_ACQUIRE) : 0u;
    const unsigned ev = c->sticky_host ? __atomic_load_n(&c->sticky_host[1], __ATOMIC_ACQUIRE) : 0u;
    if (status_out) *status_out = st;
    if (events_out) *events_out = ev;
    if (st != 0) {
        fsn_set_error("a persistent kernel on this stream ran out of time waiting for its partner workgroups (status %u, "
                      "%u such launches): something else held the CUs longer than the bound (fsn_set_persistent_timeout_ms)",
                      st, ev);
        return FSN_ERR_TIMEOUT;
    }
    return FSN_OK;
}
extern "C" int fsn_stream_timeout_policy(void* stream, int policy) {
    FSN_REQUIRE(policy == FSN_TIMEOUT_REFUSE || policy == FSN_TIMEOUT_DEFER, "timeout policy %d unknown", policy);
    CallScope scope(stream);
    cur_ctx()->timeout_defer = policy == FSN_TIMEOUT_DEFER;
    return FSN_OK;
}
extern "C" int fsn_stream_status_clear(void* stream) {
    CallScope scope(stream);
    StreamCtx* c = cur_ctx();
    if (c->sticky_host) {
        __atomic_store_n(&c->sticky_host[0], 0u, __ATOMIC_RELEASE);
        __atomic_store_n(&c->sticky_host[1], 0u, __ATOMIC_RELEASE);
    }
    return FSN_OK;
}
// Test hook for the safety net of the persistent kernels (fsn_launch_poison_if): out[0..n) becomes NaN iff *status != 0.
extern "C" int fsn_debug_poison_if(const void* status, float* out, size_t n, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(status && out, "NULL pointer argument");
    return fsn_launch_poison_if(static_cast<const unsigned*>(status), out, n, static_cast<hipStream_t>(stream));
}

// Test hook: a foreign kernel of `workgroups` x 256 threads that holds its CUs (lds_bytes of LDS each, ~200 registers
// per lane when heavy) for `ms` milliseconds on `stream`; sink is one device float it never writes.
extern "C" int fsn_debug_hog(int workgroups, int lds_bytes, int heavy, float ms, float* sink, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(workgroups >= 1 && workgroups <= 65536 && lds_bytes >= 4 && lds_bytes <= 160 * 1024 && ms >= 0.f &&
                    ms <= 10000.f && sink,
                "fsn_debug_hog: workgroups in [1, 65536], lds_bytes in [4, 163840], ms in [0, 10000], sink non-NULL");
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, t_dev);
    if (khz <= 0) khz = 100000;
    return fsn_launch_hog(workgroups, lds_bytes, heavy, (unsigned long long)((double)ms * khz), sink,
                          static_cast<hipStream_t>(stream));
}
extern "C" int fsn_profile_enable(void* stream, int on) {
    CallScope scope(stream);
    cur_ctx()->prof_on = on != 0;
    return FSN_OK;
}
extern "C" int fsn_profile_num_stages(void) { return ST_COUNT; }
extern "C" const char* fsn_profile_stage_name(int stage) {
    return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : "";
}
// Milliseconds per stage of the LAST profiled call on `stream`; waits for that call's events only.
extern "C" int fsn_profile_read(void* stream, float* ms, int n) {
    if (!ms || n < ST_COUNT) {
        fsn_set_error("fsn_profile_read: need room for %d stages", (int)ST_COUNT);
        return FSN_ERR_ARG;
    }
    CallScope scope(stream);
    StreamCtx* c = cur_ctx();
    for (int i = 0; i < ST_COUNT; ++i) {
        float v = 0.f;
        for (int j = 0; j < c->spans[i]; ++j) {
            float e = 0.f;
            if (hipEventSynchronize(c->ev[i][j][1]) != hipSuccess ||
                hipEventElapsedTime(&e, c->ev[i][j][0], c->ev[i][j][1]) != hipSuccess) {
                fsn_set_error("fsn_profile_read: event query failed for stage %s", kStageNames[i]);
                return FSN_ERR_LAUNCH;
            }
            v += e;
        }
        ms[i] = v;
    }
    return FSN_OK;
}

// ---- workspace carving -----------------------------------------------------------------------
struct Carver {
    char* base;
    size_t off;
    explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <class T>
    T* take(size_t count) {
        off = fsn_round_up_sz(off, 256);
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return r;
    }
};

static int check_cfg(const fsn_fullsubnet_cfg* cfg) {
    FSN_REQUIRE(cfg != nullptr, "cfg is NULL");
    FSN_REQUIRE(cfg->num_freqs >= 17 && cfg->num_freqs <= 4096, "num_freqs %d out of range", cfg->num_freqs);
    FSN_REQUIRE(cfg->look_ahead >= 0, "look_ahead %d < 0", cfg->look_ahead);
    FSN_REQUIRE(cfg->sb_num_neighbors >= 0 && cfg->sb_num_neighbors < cfg->num_freqs,
                "sb_num_neighbors %d must be in [0, num_freqs) (reflect padding)", cfg->sb_num_neighbors);
    FSN_REQUIRE(cfg->fb_hidden > 0 && cfg->fb_hidden % 64 == 0, "fb_hidden %d must be a multiple of 64",
                cfg->fb_hidden);
    FSN_REQUIRE(cfg->sb_hidden == 384, "sb_hidden %d unsupported (the sub-band recurrent kernel is built for 384)",
                cfg->sb_hidden);
    FSN_REQUIRE(cfg->norm_type == FSN_NORM_OFFLINE_LAPLACE || cfg->norm_type == FSN_NORM_CUMULATIVE_LAPLACE,
                "norm_type %d unsupported", cfg->norm_type);
    FSN_REQUIRE(cfg->arith == FSN_ARITH_F32 || cfg->arith == FSN_ARITH_F16X3, "arith %d unsupported", cfg->arith);
    return FSN_OK;
}

// ---- packed weights --------------------------------------------------------------------------
struct Packed {  // float offsets into the packed blob
    size_t fb_wih0, fb_whh0, fb_b0, fb_wih1, fb_whh1, fb_b1, fb_fc, fb_fcb;
    size_t sb_wih0, sb_whh0, sb_b0, sb_wih1, sb_whh1, sb_b1, sb_fc, sb_fcb;
    size_t fb_b1_frag, sb_b1_frag;  // layer-1 biases as accumulator-fragment tiles (wavefront step kernel)
    size_t sb_wih1_f16x3;           // experimental: sub-band W_ih of layer 1 split into fp16 halves (FSN_F16X3=1)
    size_t sb_whh1_f16x3;           // experimental: likewise W_hh of layer 1
    size_t sb_wih0_f16x3;           // experimental: W_ih of layer 0 (only when its padded width is 32), scale 4096
    size_t sb_whh0_f16x3;           // experimental: W_hh of layer 0
    size_t total;
    int FP, sb_kin_pad;
};
static Packed packed_layout(const fsn_fullsubnet_cfg* c) {
    Packed p;
    size_t o = 0;
    auto take = [&](size_t n) {
        o = fsn_round_up_sz(o, 64);
        const size_t r = o;
        o += n;
        return r;
    };
    const size_t Hf = c->fb_hidden, Hs = c->sb_hidden;
    p.FP = fsn_fpad(c->num_freqs);
    p.sb_kin_pad = fsn_round_up(2 * c->sb_num_neighbors + 2, 16);
    p.fb_wih0 = take(4 * Hf * p.FP);
    p.fb_whh0 = take(4 * Hf * Hf);
    p.fb_b0 = take(4 * Hf);
    p.fb_wih1 = take(4 * Hf * Hf);
    p.fb_whh1 = take(4 * Hf * Hf);
    p.fb_b1 = take(4 * Hf);
    p.fb_fc = take((size_t)p.FP * Hf);
    p.fb_fcb = take(p.FP);
    p.sb_wih0 = take(4 * Hs * p.sb_kin_pad);
    p.sb_whh0 = take(4 * Hs * Hs);
    p.sb_b0 = take(4 * Hs);
    p.sb_wih1 = take(4 * Hs * Hs);
    p.sb_whh1 = take(4 * Hs * Hs);
    p.sb_b1 = take(4 * Hs);
    p.sb_fc = take(16 * Hs);
    p.sb_fcb = take(16);
    p.fb_b1_frag = take(4 * Hf * 16);  // [4H/16 column tiles][64 lanes][4]
    p.sb_b1_frag = take(4 * Hs * 16);
    p.sb_wih1_f16x3 = take((fsn_f16x3_packed_halves(4 * (int)Hs, (int)Hs) + 1) / 2);  // halves -> floats
    p.sb_whh1_f16x3 = take((fsn_f16x3_packed_halves(4 * (int)Hs, (int)Hs) + 1) / 2);
    p.sb_wih0_f16x3 = take((fsn_f16x3_packed_halves(4 * (int)Hs, 32) + 1) / 2);
    p.sb_whh0_f16x3 = take((fsn_f16x3_packed_halves(4 * (int)Hs, (int)Hs) + 1) / 2);
    p.total = fsn_round_up_sz(o, 64);
    return p;
}

extern "C" size_t fsn_fullsubnet_packed_bytes(const fsn_fullsubnet_cfg* cfg) {
    if (check_cfg(cfg) != FSN_OK) return 0;
    return packed_layout(cfg).total * sizeof(float);
}

extern "C" int fsn_fullsubnet_pack(const fsn_fullsubnet_cfg* cfg, const fsn_fullsubnet_params* w, void* packed,
                                   size_t packed_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_cfg(cfg));
    FSN_REQUIRE(w && packed, "params / packed is NULL");
    const float* const* all = reinterpret_cast<const float* const*>(w);
    for (size_t i = 0; i < sizeof(*w) / sizeof(float*); ++i) FSN_REQUIRE(all[i], "params tensor %zu is NULL", i);
    const Packed p = packed_layout(cfg);
    FSN_REQUIRE(packed_bytes >= p.total * sizeof(float), "packed buffer too small: %zu < %zu", packed_bytes,
                p.total * sizeof(float));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* o = static_cast<float*>(packed);
    const int F = cfg->num_freqs, Hf = cfg->fb_hidden, Hs = cfg->sb_hidden;
    const int kin = 2 * cfg->sb_num_neighbors + 2;
    FSN_TRY(fsn_launch_pack(w->fb_w_ih_l0, o + p.fb_wih0, 4 * Hf, F, 4 * Hf, p.FP, s));
    FSN_TRY(fsn_launch_pack(w->fb_w_hh_l0, o + p.fb_whh0, 4 * Hf, Hf, 4 * Hf, Hf, s));
    FSN_TRY(fsn_launch_bias_sum(w->fb_b_ih_l0, w->fb_b_hh_l0, o + p.fb_b0, 4 * Hf, 4 * Hf, s));
    FSN_TRY(fsn_launch_pack(w->fb_w_ih_l1, o + p.fb_wih1, 4 * Hf, Hf, 4 * Hf, Hf, s));
    FSN_TRY(fsn_launch_pack(w->fb_w_hh_l1, o + p.fb_whh1, 4 * Hf, Hf, 4 * Hf, Hf, s));
    FSN_TRY(fsn_launch_bias_sum(w->fb_b_ih_l1, w->fb_b_hh_l1, o + p.fb_b1, 4 * Hf, 4 * Hf, s));
    FSN_TRY(fsn_launch_pack(w->fb_fc_w, o + p.fb_fc, F, Hf, p.FP, Hf, s));
    FSN_TRY(fsn_launch_bias_sum(w->fb_fc_b, nullptr, o + p.fb_fcb, F, p.FP, s));
    FSN_TRY(fsn_launch_pack(w->sb_w_ih_l0, o + p.sb_wih0, 4 * Hs, kin, 4 * Hs, p.sb_kin_pad, s));
    FSN_TRY(fsn_launch_pack(w->sb_w_hh_l0, o + p.sb_whh0, 4 * Hs, Hs, 4 * Hs, Hs, s));
    FSN_TRY(fsn_launch_bias_sum(w->sb_b_ih_l0, w->sb_b_hh_l0, o + p.sb_b0, 4 * Hs, 4 * Hs, s));
    FSN_TRY(fsn_launch_pack(w->sb_w_ih_l1, o + p.sb_wih1, 4 * Hs, Hs, 4 * Hs, Hs, s));
    FSN_TRY(fsn_launch_pack(w->sb_w_hh_l1, o + p.sb_whh1, 4 * Hs, Hs, 4 * Hs, Hs, s));
    FSN_TRY(fsn_launch_bias_sum(w->sb_b_ih_l1, w->sb_b_hh_l1, o + p.sb_b1, 4 * Hs, 4 * Hs, s));
    FSN_TRY(fsn_launch_pack(w->sb_fc_w, o + p.sb_fc, 2, Hs, 16, Hs, s));
    FSN_TRY(fsn_launch_bias_sum(w->sb_fc_b, nullptr, o + p.sb_fcb, 2, 16, s));
    FSN_TRY(fsn_launch_bias_frag(o + p.fb_b1, o + p.fb_b1_frag, 4 * Hf, s));
    FSN_TRY(fsn_launch_bias_frag(o + p.sb_b1, o + p.sb_b1_frag, 4 * Hs, s));
    if (Hs % 32 == 0) {
        FSN_TRY(fsn_launch_pack_f16x3(w->sb_w_ih_l1, o + p.sb_wih1_f16x3, 4 * Hs, Hs, s));
        FSN_TRY(fsn_launch_pack_f16x3(w->sb_w_hh_l1, o + p.sb_whh1_f16x3, 4 * Hs, Hs, s));
        FSN_TRY(fsn_launch_pack_f16x3(w->sb_w_hh_l0, o + p.sb_whh0_f16x3, 4 * Hs, Hs, s));
        if (kin == 32)
            FSN_TRY(fsn_launch_pack_f16x3(w->sb_w_ih_l0, o + p.sb_wih0_f16x3, 4 * Hs, 32, s, fsn_f16x3_wih0_scale()));
    }
    return FSN_OK;
}

// below this many sub-band row tiles (batch <= 5) the two layers of the small-batch step path run as a wavefront of
// per-step launches; from here up to the persistent regime (160 tiles) they run on the group kernel
constexpr int kWavefrontBelowTiles = 96;
constexpr int kGroupTwoFromTiles = 224;  // 56+ clusters: two per workgroup set
constexpr int kGroupMaxTiles = 264;      // 64 clusters + up to 8 left-over tiles

// ---- model core: magT [B][Tp][FP] -> crm_r, crm_i [B][T][FP] ------------------------------------
struct CoreDims {
    int B, T, Tp, F, FP, Hf, Hs, nb, la;
    int Npad_fb;       // full-band rows per step (batch, padded to 16)
    bool fb_chain;     // the full-band LSTM layers run as one persistent launch (fb_chain_kernels.hip)
    int N, Npad;       // sub-band rows per step, padded rows (row stride of the [t][n] buffers)
    FsnRecPlan rec;    // how those rows are spread over the CUs
    bool fc_fused;     // output layer fused into the layer-1 persistent kernel (its hseq is never stored)
    bool l1x;          // layer 1 forms its input projection itself (lstm_rec_x_kernel): no projection GEMM, no gx
    int grp_clusters;  // > 0: the step regime runs on the group kernel (lstm_group_kernels.hip), that many clusters of 64 rows
    long row0;         // row-range calls: the N sub-band rows are rows row0 .. row0 + N - 1 of the B F rows
    int den_stride;    // row stride of the per-row (cumulative) sub-band divisors: they are indexed by GLOBAL row
};
// n_rows < 0: all B F sub-band rows; otherwise the rows [row0, row0 + n_rows) of the flattened (b, f) index space
// (the full-band model and the norm statistics always cover the B whole utterances).
static CoreDims core_dims(const fsn_fullsubnet_cfg* c, int B, int T, long row0 = 0, long n_rows = -1) {
    CoreDims d;
    d.B = B;
    d.T = T;
    d.la = c->look_ahead;
    d.Tp = T + c->look_ahead;
    d.F = c->num_freqs;
    d.FP = fsn_fpad(d.F);
    d.Hf = c->fb_hidden;
    d.Hs = c->sb_hidden;
    d.nb = c->sb_num_neighbors;
    d.Npad_fb = fsn_round_up(B, 16);
    d.fb_chain = fsn_fb_chain_supported(d.Hf, d.Npad_fb) && d.Tp <= fsn_fb_chain_max_steps();
    d.N = n_rows < 0 ? B * d.F : (int)n_rows;
    d.row0 = n_rows < 0 ? 0 : row0;
    d.rec = fsn_lstm_rec_plan(d.N, d.Hs);
    // 224 - 264 row tiles (14 - 16 utterances: one rank's share of config 2 at 4 GPUs) run on the group kernel with two
    // clusters per workgroup set (lstm_group_kernels.hip) instead of the persistent kernels at ONE row tile per CU
    // (every CU streams all weights every step there): 26.5 -> 23.4 ms at 16 utterances
    const bool grp_shape = d.Hs == 384 && fsn_round_up(2 * c->sb_num_neighbors + 2, 16) == 32 && c->arith == FSN_ARITH_F32;
    if (grp_shape && d.rec.tiles >= kGroupTwoFromTiles && d.rec.tiles <= kGroupMaxTiles && d.rec.main_wgs > 0 &&
        4 * fsn_lstm2_group_clusters(d.rec.tiles) + 8 >= d.rec.tiles) {  // ... and the device holds (nearly) all of them
        d.rec.rt = 1;
        d.rec.main_wgs = 0;
        d.rec.left_tiles = d.rec.tiles;
    }
    d.Npad = d.rec.npad;
    d.den_stride = n_rows < 0 ? d.Npad : fsn_round_up(B * d.F, 16);
    d.fc_fused = d.rec.main_wgs > 0 && fsn_lstm_rec_can_fuse_fc(d.rec.rt, false);
    d.l1x = d.fc_fused && c->arith == FSN_ARITH_F32 && fsn_lstm_rec_x_supported(d.Hs, d.rec.rt);
    // 96 - 159 row tiles (6 - 9 utterances; below that the two-layer wavefront of per-step launches is as fast)
    d.grp_clusters = 0;
    if (d.rec.main_wgs == 0 && d.rec.left_tiles >= kWavefrontBelowTiles && grp_shape)  // fp32 only: the group kernel has no f16x3 form
        d.grp_clusters = fsn_lstm2_group_clusters(d.rec.left_tiles);
    return d;
}
struct CoreWs {
    float *gx_fb, *hseq_fb0, *hseq_fb1, *c_fb, *fb_out, *den_fb, *den_sb, *gx_sb, *hseq_sb0, *hseq_sb1, *c_left;
    float* hseq_left0;  // l1x: layer-0 hidden sequence of the left-over rows, compact [t][left rows][H]
    float* grp_exchange;  // group kernel: h exchange buffers of the clusters
    unsigned* grp_flags;
    float* fb_exchange;   // full-band chain kernel: per-step h / projection hand-off buffers
    unsigned* fb_flags;
    double* binsum;
};
static CoreWs core_carve(Carver& cv, const CoreDims& d, int norm_type) {
    CoreWs w;
    const size_t rows_fb = (size_t)d.Tp * d.Npad_fb, rows_sb = (size_t)d.Tp * d.Npad;
    w.gx_fb = cv.take<float>(rows_fb * 4 * d.Hf);
    w.hseq_fb0 = cv.take<float>(rows_fb * d.Hf);
    w.hseq_fb1 = cv.take<float>(rows_fb * d.Hf);
    w.c_fb = cv.take<float>((size_t)2 * d.Npad_fb * d.Hf);  // one cell state per layer (wavefront)
    w.fb_out = cv.take<float>((size_t)d.B * d.Tp * d.FP);
    w.binsum = cv.take<double>((size_t)d.B * d.FP);
    const bool cum = norm_type == FSN_NORM_CUMULATIVE_LAPLACE;
    w.den_fb = cv.take<float>(cum ? (size_t)d.B * d.Tp : (size_t)d.B);
    w.den_sb = cv.take<float>(cum ? (size_t)d.Tp * d.den_stride : (size_t)d.B);
    // l1x: only the left-over rows (which run step by step) still need a precomputed projection
    const size_t rows_left = (size_t)d.Tp * (d.rec.left_tiles > 0 ? d.rec.left_tiles : 1) * 16;
    if (d.grp_clusters > 0) {
        // group kernel: projections and hidden sequences only exist for the rows that do not fill a cluster
        const int aux_tiles = d.rec.tiles - 4 * d.grp_clusters;
        const size_t rows_aux = (size_t)d.Tp * (aux_tiles > 0 ? aux_tiles : 1) * 16;
        w.gx_sb = cv.take<float>(rows_aux * 4 * d.Hs);
        w.hseq_sb0 = cv.take<float>(rows_aux * d.Hs);
        w.hseq_left0 = nullptr;
        w.hseq_sb1 = cv.take<float>(rows_aux * d.Hs);
    } else {
        w.gx_sb = cv.take<float>((d.l1x ? rows_left : rows_sb) * 4 * d.Hs);
        w.hseq_sb0 = cv.take<float>(rows_sb * d.Hs);
        w.hseq_left0 = d.l1x ? cv.take<float>(rows_left * d.Hs) : nullptr;
        // fused output layer: only the left-over rows of layer 1 are ever stored, [t][left rows][H]
        w.hseq_sb1 = cv.take<float>(d.fc_fused ? (size_t)d.Tp * (d.rec.left_tiles > 0 ? d.rec.left_tiles : 1) * 16 * d.Hs
                                               : rows_sb * d.Hs);
    }
    w.c_left = cv.take<float>((size_t)2 * (d.rec.left_tiles > 0 ? d.rec.left_tiles : 1) * 16 * d.Hs);
    w.grp_exchange = d.grp_clusters ? cv.take<float>(fsn_lstm2_group_exchange_floats(d.grp_clusters)) : nullptr;
    w.grp_flags = d.grp_clusters ? cv.take<unsigned>(fsn_lstm2_group_flag_words(d.grp_clusters)) : nullptr;
    w.fb_exchange = d.fb_chain ? cv.take<float>(fsn_fb_chain_exchange_floats(d.Tp, d.Npad_fb)) : nullptr;
    w.fb_flags = d.fb_chain ? cv.take<unsigned>(fsn_fb_chain_flag_words()) : nullptr;
    return w;
}

// ---- auxiliary stream for the left-over sub-band rows (see fsn_lstm_rec_plan) -------------------
// One per (device, caller stream), created lazily on the caller stream's device (StreamCtx).  The fork / join
// below uses events only, so it is also legal under stream capture.
static int aux_init(StreamCtx* c) {
    if (c->aux) return FSN_OK;
    if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        fsn_set_error("cannot create the auxiliary stream / events");
        c->aux = nullptr;
        return FSN_ERR_LAUNCH;
    }
    return FSN_OK;
}

// One sub-band LSTM layer over all Tp steps: the persistent kernel on `s` and, concurrently, the
// few left-over row tiles as per-step launches on the auxiliary stream.
// Main kernel: input projection either precomputed (`gx`, tile (t, i) at t * tiles + i) or built
// in-kernel from `xin`.  Left-over tiles: projection tiles in `gx_left` at t * left_stride + left_off + i.
// x_main (with wih_main, bias_main): the main rows run on lstm_rec_x_kernel, which reads the hidden sequence of the
// layer below (x_main [Tp][Npad][H]) and forms its input projection itself.  hseq_left: the left-over rows' hidden
// sequence goes to this compact [t][left rows][H] buffer instead of rows [main rows, Npad) of hseq.
static int run_recurrence(const float* gx, const FsnSbInput* xin, const float* gx_left, long left_stride,
                          long left_off, const float* whh, float* hseq, float* c_left, int Tp, int Npad, int H,
                          const FsnRecPlan& r, hipStream_t s, const FsnRecFc* fc = nullptr, long left_hs_stride = -1,
                          const void* whh_f16x3 = nullptr, const void* wih_f16x3 = nullptr,
                          const float* x_main = nullptr, const float* wih_main = nullptr,
                          const float* bias_main = nullptr, float* hseq_left = nullptr) {
    // No persistent part (fewer than ~160 tiles): the steps run on `s` itself - groups of four tiles through the
    // one-workgroup-per-CU step kernel, the up to three tiles that do not fill a group beside it on the
    // auxiliary stream (a 33rd group of 8 workgroups would be a second round on 8 CUs and double the step).
    const int cu_tiles = r.main_wgs == 0 && r.left_tiles >= 8 ? r.left_tiles / 4 * 4 : 0;
    const int aux_tiles = r.left_tiles - cu_tiles;
    const bool fork = aux_tiles > 0 && (r.main_wgs > 0 || cu_tiles > 0);
    hipStream_t ls = s;
    StreamCtx* cx = nullptr;
    if (fork) {
        cx = cur_ctx();
        FSN_TRY(aux_init(cx));
        if (hipEventRecord(cx->ev_fork, s) != hipSuccess || hipStreamWaitEvent(cx->aux, cx->ev_fork, 0) != hipSuccess) {
            fsn_set_error("aux stream fork failed");
            return FSN_ERR_LAUNCH;
        }
        ls = cx->aux;
    }
    if (r.main_wgs > 0) {
        if (x_main)
            FSN_TRY(fsn_launch_lstm_rec_x(x_main, wih_main, whh, bias_main, Tp, Npad, H, r.rt, r.main_wgs, s, fc,
                                          fc ? nullptr : hseq));  // no output layer: a layer inside a stack, h_t stored
        else if (xin && !fc && !whh_f16x3 && fsn_lstm_rec_in_supported(xin, whh, H, r.rt))
            FSN_TRY(fsn_launch_lstm_rec_in(xin, whh, hseq, Tp, Npad, H, r.rt, r.main_wgs, s));
        else if (whh_f16x3 && fc && !xin && r.rt >= 2)  // experimental split-precision persistent kernel (FSN_F16X3=1)
            FSN_TRY(fsn_launch_lstm_rec_f16x3(gx, whh_f16x3, Tp, Npad, H, r.rt, r.main_wgs, fc, s));
        else if (whh_f16x3 && wih_f16x3 && xin && !xin->x_rows && xin->kin_chunks == 2 && r.rt >= 2)
            FSN_TRY(fsn_launch_lstm_rec_xin_f16x3(xin, wih_f16x3, whh_f16x3, hseq, Tp, Npad, H, r.rt, r.main_wgs, s));
        else
            FSN_TRY(fsn_launch_lstm_rec(gx, xin, whh, hseq, Tp, Npad, H, r.rt, r.main_wgs, s, fc));
    }
    if (r.left_tiles > 0) {
        // left-over rows of step t: rows [main_rows, Npad) of the full [t][Npad] matrix, or - when the
        // persistent part stores nothing (fused output layer) - a compact [t][left rows] matrix
        if (hseq_left) left_hs_stride = (long)r.left_tiles * 16;
        float* hl = hseq_left ? hseq_left : hseq;
        const long hs_stride = left_hs_stride >= 0 ? left_hs_stride : Npad;
        const long hs_off = left_hs_stride >= 0 ? 0 : (long)r.main_wgs * r.rt * 16;
        for (int t = 0; t < Tp; ++t) {
            float* h_out = hl + ((size_t)t * hs_stride + hs_off) * H;
            const float* h_prev = t ? hl + ((size_t)(t - 1) * hs_stride + hs_off) * H : h_out;
            const long gx_rt0 = (long)t * left_stride + left_off;
            if (cu_tiles > 0)
                FSN_TRY(fsn_launch_lstm_step_cu(gx_left, whh, h_prev, h_out, c_left, gx_rt0, cu_tiles, H, t == 0, s));
            if (aux_tiles > 0) {
                const size_t ro = (size_t)cu_tiles * 16 * H;
                FSN_TRY(fsn_launch_lstm_step(gx_left, whh, h_prev + ro, h_out + ro, c_left + ro, gx_rt0 + cu_tiles,
                                             aux_tiles, H, t == 0, ls, fork ? 1 : 0));
            }
        }
    }
    if (fork) {
        if (hipEventRecord(cx->ev_join, cx->aux) != hipSuccess || hipStreamWaitEvent(s, cx->ev_join, 0) != hipSuccess) {
            fsn_set_error("aux stream join failed");
            return FSN_ERR_LAUNCH;
        }
    }
    return FSN_OK;
}

static int run_sb_recurrence(const float* gx, const FsnSbInput* xin, const float* gx_left, long left_stride,
                             long left_off, const float* whh, float* hseq, float* c_left, const CoreDims& d,
                             hipStream_t s, const FsnRecFc* fc = nullptr, const void* whh_f16x3 = nullptr,
                             const void* wih_f16x3 = nullptr, const float* x_main = nullptr,
                             const float* wih_main = nullptr, const float* bias_main = nullptr,
                             float* hseq_left = nullptr) {
    return run_recurrence(gx, xin, gx_left, left_stride, left_off, whh, hseq, c_left, d.Tp, d.Npad, d.Hs, d.rec, s,
                          fc, fc ? (long)d.rec.left_tiles * 16 : -1, whh_f16x3, wih_f16x3, x_main, wih_main,
                          bias_main, hseq_left);
}

static int run_core(const fsn_fullsubnet_cfg* cfg, const float* pk, const float* magT, const CoreDims& d,
                    const CoreWs& w, float* crm_r, float* crm_i, hipStream_t s, bool fullband_only = false) {
    const Packed p = packed_layout(cfg);
    const bool cum = cfg->norm_type == FSN_NORM_CUMULATIVE_LAPLACE;

    // full-band norm divisor (fullsubnet/model.py:92)
    {
        StageTimer st(ST_NORM, s);
        if (cum) {
            FSN_TRY(fsn_launch_cumulative_den_fb(magT, w.den_fb, d.B, d.Tp, d.F, d.FP, s));
        } else {
            FSN_TRY(fsn_launch_binsum(magT, w.binsum, d.B, d.Tp, d.FP, s));
            FSN_TRY(fsn_launch_offline_den(w.binsum, nullptr, w.den_fb, nullptr, d.B, d.Tp, d.F, d.FP, d.nb, 0, s));
        }
    }
    // full-band model (model.py:95): 2 LSTM layers + Linear + ReLU
    const int fb_rt = d.Tp * d.Npad_fb / 16;
    FsnGemmA a{};
    FsnGemmC c{};
    {
        StageTimer st(ST_FB_GEMM, s);
        a = FsnGemmA{};
        c = FsnGemmC{};
        a.kind = 1;
        a.p0 = magT;
        a.den = w.den_fb;
        a.den_mode = cum ? 1 : 0;
        a.B = d.B;
        a.Tp = d.Tp;
        a.F = d.F;
        a.FP = d.FP;
        a.Npad = d.Npad_fb;
        c.kind = 0;
        c.p0 = w.gx_fb;
        c.bias = pk + p.fb_b0;
        FSN_TRY(fsn_launch_gemm(a, pk + p.fb_wih0, c, fb_rt, 4 * d.Hf / 16, d.FP / 16, s));
    }
    {
        // N = B rows only: a chain of tiny dependent launches, so the two layers advance as a wavefront
        // (layer 1 at step t next to layer 0 at step t + 1): T' + 1 launches instead of 2 T'
        StageTimer st(ST_FB_REC, s);
        if (d.fb_chain) {  // up to 64 utterances, H = 512: the whole chain as one persistent launch
            FSN_PERSIST_BEGIN(s);
            FSN_TRY(fsn_launch_fb_chain(w.gx_fb, pk + p.fb_whh0, pk + p.fb_wih1, pk + p.fb_whh1, pk + p.fb_b1,
                                        w.fb_exchange, w.fb_flags, w.hseq_fb1, d.Tp, d.Npad_fb, d.Hf, s));
            FSN_TRY(fsn_launch_poison_if(w.fb_flags + fsn_fb_chain_status_word(), w.hseq_fb1,
                                         (size_t)d.Tp * d.Npad_fb * d.Hf, s));
        } else {
            FSN_TRY(fsn_launch_lstm_wavefront2(w.gx_fb, d.Npad_fb / 16, 0, pk + p.fb_whh0, pk + p.fb_wih1,
                                               pk + p.fb_b1_frag, pk + p.fb_whh1, w.hseq_fb0, w.hseq_fb1, d.Npad_fb, 0,
                                               w.c_fb, w.c_fb + (size_t)d.Npad_fb * d.Hf, d.Tp, d.Npad_fb / 16, d.Hf, s));
        }
    }
    {
        StageTimer st(ST_FB_GEMM, s);
        a = FsnGemmA{};
        c = FsnGemmC{};
        a.kind = 0;
        a.p0 = w.hseq_fb1;
        a.ld = d.Hf;
        c.kind = 1;
        c.p0 = w.fb_out;
        c.bias = pk + p.fb_fcb;
        c.B = d.B;
        c.Tp = d.Tp;
        c.F = d.F;
        c.FP = d.FP;
        c.Npad = d.Npad_fb;
        FSN_TRY(fsn_launch_gemm(a, pk + p.fb_fc, c, fb_rt, d.FP / 16, d.Hf / 16, s));
    }
    if (fullband_only) return FSN_OK;  // fsn_fullsubnet_fullband: w.fb_out is the result
    // sub-band norm divisor over the (virtual) concatenated sub-band input (model.py:110-111)
    {
        StageTimer st(ST_NORM, s);
        if (cum) {
            FSN_TRY(fsn_launch_cumulative_den_sb(magT, w.fb_out, w.den_sb, d.B, d.Tp, d.F, d.FP, d.nb, d.den_stride, s));
        } else {
            FSN_TRY(fsn_launch_offline_den(w.binsum, w.fb_out, nullptr, w.den_sb, d.B, d.Tp, d.F, d.FP, d.nb, 1, s));
        }
    }
    if (d.grp_clusters > 0) {
        // Few rows (6 - 9 utterances): both layers + output layer of the first 64 x clusters rows as ONE persistent launch
        // (lstm_group_kernels.hip); what does not fill a cluster runs beside it on the auxiliary stream as the two-layer
        // wavefront of per-step launches (its projection GEMM first, its output layer last).
        const long grp_rows = (long)d.grp_clusters * 64;
        const int aux_tiles = d.rec.tiles - d.grp_clusters * 4;
        FsnSbInput xin{};
        xin.mag = magT;
        xin.fb_out = w.fb_out;
        xin.den = w.den_sb;
        xin.wih_p = pk + p.sb_wih0;
        xin.bias = pk + p.sb_b0;
        xin.den_mode = cum ? 1 : 0;
        xin.den_stride = d.den_stride;
        xin.row0 = d.row0;
        xin.B = d.B;
        xin.Tp = d.Tp;
        xin.F = d.F;
        xin.FP = d.FP;
        xin.N = d.N < grp_rows ? d.N : (int)grp_rows;
        xin.nb = d.nb;
        xin.kin_chunks = p.sb_kin_pad / 16;
        FsnRecFc gfc{};
        gfc.w_p = pk + p.sb_fc;
        gfc.bias = pk + p.sb_fcb;
        gfc.crm_r = crm_r;
        gfc.crm_i = crm_i;
        gfc.N = xin.N;
        gfc.row0 = d.row0;
        gfc.F = d.F;
        gfc.FP = d.FP;
        gfc.T = d.T;
        gfc.la = d.la;
        // The group kernel fills every CU with two 216-register workgroups: what runs beside it must fit in the 80
        // registers per lane that are left - the two-layer wavefront step kernel (78) and the output-layer GEMM (52) do,
        // the projection GEMM of the left-over rows does not, so it goes first, on the caller's stream.
        StreamCtx* cx = nullptr;
        hipStream_t as = s;
        if (aux_tiles > 0) {
            a = FsnGemmA{};
            c = FsnGemmC{};
            a.kind = 2;
            a.p0 = magT;
            a.p1 = w.fb_out;
            a.den = w.den_sb;
            a.den_mode = cum ? 1 : 0;
            a.den_stride = d.den_stride;
            a.B = d.B;
            a.Tp = d.Tp;
            a.F = d.F;
            a.FP = d.FP;
            a.Npad = aux_tiles * 16;
            a.n_offset = (int)(d.row0 + grp_rows);
            a.N = (int)(d.row0 + d.N);
            a.nb = d.nb;
            c.kind = 0;
            c.p0 = w.gx_sb;
            c.bias = pk + p.sb_b0;
            {
                StageTimer st(ST_SB_GEMM_L0, s);
                FSN_TRY(fsn_launch_gemm(a, pk + p.sb_wih0, c, d.Tp * aux_tiles, 4 * d.Hs / 16, p.sb_kin_pad / 16, s));
            }
            cx = cur_ctx();
            FSN_TRY(aux_init(cx));
            if (hipEventRecord(cx->ev_fork, s) != hipSuccess || hipStreamWaitEvent(cx->aux, cx->ev_fork, 0) != hipSuccess) {
                fsn_set_error("aux stream fork failed");
                return FSN_ERR_LAUNCH;
            }
            as = cx->aux;
        }
        {
            StageTimer st(ST_SB_REC_L0, s);
            FSN_PERSIST_BEGIN(s);
            FSN_TRY(fsn_launch_lstm2_group(&xin, pk + p.sb_whh0, pk + p.sb_wih1, pk + p.sb_whh1, pk + p.sb_b1,
                                           w.grp_exchange, w.grp_flags, &gfc, d.Tp, d.grp_clusters, d.Hs, s));
        }
        if (aux_tiles > 0) {
            FSN_TRY(fsn_launch_lstm_wavefront2(w.gx_sb, aux_tiles, 0, pk + p.sb_whh0, pk + p.sb_wih1, pk + p.sb_b1_frag,
                                               pk + p.sb_whh1, w.hseq_sb0, w.hseq_sb1, (long)aux_tiles * 16, 0, w.c_left,
                                               w.c_left + (size_t)aux_tiles * 16 * d.Hs, d.Tp, aux_tiles, d.Hs, as, nullptr,
                                               nullptr, 1));
            a = FsnGemmA{};
            c = FsnGemmC{};
            a.kind = 0;
            a.p0 = w.hseq_sb1;
            a.ld = d.Hs;
            c.kind = 2;
            c.p0 = crm_r;
            c.p1 = crm_i;
            c.bias = pk + p.sb_fcb;
            c.T = d.T;
            c.F = d.F;
            c.FP = d.FP;
            c.Npad = aux_tiles * 16;
            c.N = (int)(d.row0 + d.N);
            c.n_off = (int)(d.row0 + grp_rows);
            c.la = d.la;
            FSN_TRY(fsn_launch_gemm(a, pk + p.sb_fc, c, d.Tp * aux_tiles, 1, d.Hs / 16, as));
            if (hipEventRecord(cx->ev_join, cx->aux) != hipSuccess || hipStreamWaitEvent(s, cx->ev_join, 0) != hipSuccess) {
                fsn_set_error("aux stream join failed");
                return FSN_ERR_LAUNCH;
            }
        }
        // a spin bound hit inside the group launch (see fsn_launch_poison_if): the mask planes become NaN instead of
        // garbage - AFTER the join: the left-over rows' output layer on the auxiliary stream writes into the same planes
        // (poisoned before it, a launch that gave up early left those rows finite: one run of the residency test in many)
        const unsigned* st_word = w.grp_flags + fsn_lstm2_group_status_word(d.grp_clusters);
        FSN_TRY(fsn_launch_poison_if(st_word, crm_r, (size_t)d.B * d.T * d.FP, s));
        FSN_TRY(fsn_launch_poison_if(st_word, crm_i, (size_t)d.B * d.T * d.FP, s));
        return FSN_OK;
    }
    // sub-band model (model.py:121-128): N = B F sequences, 2 LSTM layers + Linear(2)
    const int sb_rt = (int)((long)d.Tp * d.Npad / 16);
    // Layer 0: the K = 2nb+2 input projection is fused into the persistent recurrent kernel (no 19 GB
    // gx round trip); only the few left-over tiles, which run step by step, get a precomputed gx.
    const long main_rows = (long)d.rec.main_wgs * d.rec.rt * 16;
    if (d.rec.left_tiles > 0) {
        StageTimer st(ST_SB_GEMM_L0, s);
        a = FsnGemmA{};
        c = FsnGemmC{};
        a.kind = 2;
        a.p0 = magT;
        a.p1 = w.fb_out;
        a.den = w.den_sb;
        a.den_mode = cum ? 1 : 0;
        a.den_stride = d.den_stride;
        a.B = d.B;
        a.Tp = d.Tp;
        a.F = d.F;
        a.FP = d.FP;
        a.Npad = d.rec.left_tiles * 16;
        a.n_offset = (int)(d.row0 + main_rows);  // the provider works on global rows: first row and row limit
        a.N = (int)(d.row0 + d.N);
        a.nb = d.nb;
        c.kind = 0;
        c.p0 = w.gx_sb;
        c.bias = pk + p.sb_b0;
        FSN_TRY(fsn_launch_gemm(a, pk + p.sb_wih0, c, d.Tp * d.rec.left_tiles, 4 * d.Hs / 16, p.sb_kin_pad / 16, s));
    }
    // Small batches (no persistent part): both layers as one wavefront of per-step launches on the
    // projection computed above.
    const bool sb_wave = d.rec.main_wgs == 0 && d.rec.left_tiles < kWavefrontBelowTiles;
    if (sb_wave) {
        StageTimer st(ST_SB_REC_L0, s);
        FSN_TRY(fsn_launch_lstm_wavefront2(w.gx_sb, d.rec.left_tiles, 0, pk + p.sb_whh0, pk + p.sb_wih1, pk + p.sb_b1_frag,
                                           pk + p.sb_whh1, w.hseq_sb0, w.hseq_sb1, d.Npad, 0, w.c_left,
                                           w.c_left + (size_t)d.rec.left_tiles * 16 * d.Hs, d.Tp, d.rec.left_tiles,
                                           d.Hs, s));
    } else {
        StageTimer st(ST_SB_REC_L0, s);
        FsnSbInput xin{};
        xin.mag = magT;
        xin.fb_out = w.fb_out;
        xin.den = w.den_sb;
        xin.wih_p = pk + p.sb_wih0;
        xin.bias = pk + p.sb_b0;
        xin.den_mode = cum ? 1 : 0;
        xin.den_stride = d.den_stride;
        xin.row0 = d.row0;
        xin.B = d.B;
        xin.Tp = d.Tp;
        xin.F = d.F;
        xin.FP = d.FP;
        xin.N = d.N;
        xin.nb = d.nb;
        xin.kin_chunks = p.sb_kin_pad / 16;
        const bool f16x3 = cfg->arith == FSN_ARITH_F16X3;  // opt-in experiment, chosen by the caller
        const bool l0_split = f16x3 && d.Hs == 384 && 2 * d.nb + 2 == 32;
        FSN_TRY(run_sb_recurrence(nullptr, &xin, w.gx_sb, d.rec.left_tiles, 0, pk + p.sb_whh0, w.hseq_sb0, w.c_left,
                                  d, s, nullptr, l0_split ? pk + p.sb_whh0_f16x3 : nullptr,
                                  l0_split ? pk + p.sb_wih0_f16x3 : nullptr, nullptr, nullptr, nullptr,
                                  d.l1x ? w.hseq_left0 : nullptr));
    }
    if (!sb_wave && d.l1x) {
        // the main rows form this projection inside lstm_rec_x_kernel; only the left-over rows (step kernels) get one
        if (d.rec.left_tiles > 0) {
            StageTimer st(ST_SB_GEMM_L1, s);
            a = FsnGemmA{};
            c = FsnGemmC{};
            a.kind = 0;
            a.p0 = w.hseq_left0;
            a.ld = d.Hs;
            c.kind = 0;
            c.p0 = w.gx_sb;
            c.bias = pk + p.sb_b1;
            FSN_TRY(fsn_launch_gemm(a, pk + p.sb_wih1, c, d.Tp * d.rec.left_tiles, 4 * d.Hs / 16, d.Hs / 16, s));
        }
    } else if (!sb_wave) {
        StageTimer st(ST_SB_GEMM_L1, s);
        a = FsnGemmA{};
        c = FsnGemmC{};
        a.kind = 0;
        a.p0 = w.hseq_sb0;
        a.ld = d.Hs;
        c.kind = 0;
        c.p0 = w.gx_sb;
        c.bias = pk + p.sb_b1;
        const bool f16x3 = cfg->arith == FSN_ARITH_F16X3;  // opt-in experiment, chosen by the caller
        if (f16x3)
            FSN_TRY(fsn_launch_gemm_f16x3(w.hseq_sb0, d.Hs, pk + p.sb_wih1_f16x3, pk + p.sb_b1, w.gx_sb, sb_rt, 4 * d.Hs,
                                          d.Hs, s));
        else
            FSN_TRY(fsn_launch_gemm(a, pk + p.sb_wih1, c, sb_rt, 4 * d.Hs / 16, d.Hs / 16, s));
    }
    // Output layer (model.py:53-61,129-135).  Where the persistent 4-pass kernel runs layer 1 it forms the two
    // mask values of a row from h_t in LDS and that layer's 4.8 GB hidden sequence is never written or read
    // back; only rows that went step by step (left-over tiles, small batches) go through the GEMM below.
    const bool fc_fused = d.fc_fused;
    FsnRecFc fc{};
    if (fc_fused) {
        fc.w_p = pk + p.sb_fc;
        fc.bias = pk + p.sb_fcb;
        fc.crm_r = crm_r;
        fc.crm_i = crm_i;
        fc.N = d.N;
        fc.row0 = d.row0;
        fc.F = d.F;
        fc.FP = d.FP;
        fc.T = d.T;
        fc.la = d.la;
    }
    if (!sb_wave && d.l1x) {
        StageTimer st(ST_SB_REC_L1, s);
        FSN_TRY(run_sb_recurrence(nullptr, nullptr, w.gx_sb, d.rec.left_tiles, 0, pk + p.sb_whh1, w.hseq_sb1, w.c_left, d,
                                  s, &fc, nullptr, nullptr, w.hseq_sb0, pk + p.sb_wih1, pk + p.sb_b1));
    } else if (!sb_wave) {
        StageTimer st(ST_SB_REC_L1, s);
        const bool f16x3 = cfg->arith == FSN_ARITH_F16X3;  // opt-in experiment, chosen by the caller
        FSN_TRY(run_sb_recurrence(w.gx_sb, nullptr, w.gx_sb, d.rec.tiles, main_rows / 16, pk + p.sb_whh1, w.hseq_sb1,
                                  w.c_left, d, s, fc_fused ? &fc : nullptr,
                                  f16x3 && fc_fused ? pk + p.sb_whh1_f16x3 : nullptr));
    }
    if (!fc_fused || d.rec.left_tiles > 0) {
        StageTimer st(ST_SB_FC, s);
        a = FsnGemmA{};
        c = FsnGemmC{};
        a.kind = 0;
        a.p0 = w.hseq_sb1;
        a.ld = d.Hs;
        c.kind = 2;
        c.p0 = crm_r;
        c.p1 = crm_i;
        c.bias = pk + p.sb_fcb;
        c.T = d.T;
        c.F = d.F;
        c.FP = d.FP;
        c.Npad = d.Npad;
        c.N = (int)(d.row0 + d.N);  // global rows, like the A provider above
        c.n_off = (int)d.row0;
        c.la = d.la;
        int rows_t = sb_rt;
        if (fc_fused) {  // only the left-over rows: hseq_sb1 is the compact [t][left rows][H] matrix
            c.Npad = d.rec.left_tiles * 16;
            c.n_off = (int)(d.row0 + main_rows);
            rows_t = d.Tp * d.rec.left_tiles;
        }
        FSN_TRY(fsn_launch_gemm(a, pk + p.sb_fc, c, rows_t, 1, d.Hs / 16, s));
    }
    return FSN_OK;
}

static int check_bt(int B, int T) {
    FSN_REQUIRE(B >= 1 && B <= 4096, "batch %d out of range", B);
    FSN_REQUIRE(T >= 1 && T <= 100000, "frames %d out of range", T);
    return FSN_OK;
}

// Batches beyond what ONE round of the persistent kernels holds at 4 row tiles per workgroup (64 utterances of 257 bins
// on 256 CUs) run as whole chunks of that size plus a remainder, one after the other: the model has no cross-utterance
// term (both norms are per utterance), and a workgroup walks its RT tiles one after the other every step, so a batch
// that does not fill rounds x RT x CUs tiles pays for the full round - 104 utterances took 171 ms as two rounds of 4,
// 64 + 40 take 84 + 63.  Returns the chunk size (B itself: no chunking).
static int core_chunk(const fsn_fullsubnet_cfg* cfg, int B) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    const long b0 = ((long)cus * 4 + 16) * 16 / cfg->num_freqs;
    return b0 >= 1 && B > b0 ? (int)b0 : B;
}
// Below one round the same holds between the regimes: 40 utterances take as long as 48 (one round of 3 tiles per
// workgroup), 24 as long as 32, 10 - 13 run at one tile per CU - where 32 + 8, 16 + 8 and 8 + 2 as separate calls are 12 -
// 18 % faster.  Time of one core call in microseconds per frame step, from the plan it would take (calibrated on
// config 2's clips: 1 / 2 / 4 / 8 / 16 / 32 / 48 / 64 utterances = 24 / 37 / 60 / 66 / 122 / 229 / 337 / 441 us per step):
static double core_cost(const fsn_fullsubnet_cfg* cfg, int b) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    const CoreDims d = core_dims(cfg, b, 64);
    double c = 6.0;  // the full-band chain and the fixed launches of a call
    if (d.rec.main_wgs > 0) {
        const int rounds = (d.rec.main_wgs + cus - 1) / cus;
        c += rounds * (d.rec.rt == 1 ? 120.0 : 110.0 * d.rec.rt);  // one tile per CU streams all weights for 16 rows
    } else if (d.grp_clusters > 0) {
        c += (d.grp_clusters > cus / 8 ? 2 : 1) * 58.0 + (d.rec.left_tiles - 4 * d.grp_clusters > 0 ? 2.0 : 0.0);
    } else {
        c += 7.0 + 0.67 * d.rec.left_tiles;  // two-layer wavefront of per-step launches
    }
    return c;
}
// the chunk sizes of a batch, largest first: whole rounds of core_chunk(), then the cheapest split of the remainder into
// {itself, 48, 32, 16, 8}-utterance calls by core_cost
static int core_chunks_search(const fsn_fullsubnet_cfg* cfg, int B, int* sizes, int max_sizes);
// The search evaluates core_cost / core_dims ~5 x (remainder) times, each with device-attribute and occupancy lookups:
// ~10^4 host calls at B = 64, three times per fsn_enhance (workspace query, workspace check, run).  The plan depends
// only on (configuration, B, device, persistent mode): memoised.
static int core_chunks(const fsn_fullsubnet_cfg* cfg, int B, int* sizes, int max_sizes) {
    static std::mutex mu;
    static std::map<std::string, std::vector<int>> memo;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::string key(reinterpret_cast<const char*>(cfg), sizeof(*cfg));
    const int tail[4] = {B, dev, fsn_persistent_allowed() ? 1 : 0, max_sizes};
    key.append(reinterpret_cast<const char*>(tail), sizeof(tail));
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(key);
        if (it != memo.end()) {
            for (size_t i = 0; i < it->second.size(); ++i) sizes[i] = it->second[i];
            return (int)it->second.size();
        }
    }
    const int n = core_chunks_search(cfg, B, sizes, max_sizes);
    std::lock_guard<std::mutex> lk(mu);
    if (memo.size() > 4096) memo.clear();
    memo[key] = std::vector<int>(sizes, sizes + n);
    return n;
}
static int core_chunks_search(const fsn_fullsubnet_cfg* cfg, int B, int* sizes, int max_sizes) {
    int n = 0;
    const int full = core_chunk(cfg, B);
    int rem = B;
    while (rem > full && n < max_sizes - 8) {
        sizes[n++] = full;
        rem -= full;
    }
    if (rem > 64 || !fsn_persistent_allowed() || cfg->arith != FSN_ARITH_F32) {  // outside the calibrated range: as one call
        sizes[n++] = rem;
        return n;
    }
    double best[65];
    int first[65];
    best[0] = 0.0;
    first[0] = 0;
    for (int b = 1; b <= rem; ++b) {
        best[b] = core_cost(cfg, b);
        first[b] = b;
        for (int c : {48, 32, 16, 8}) {
            if (c >= b) continue;
            const double v = core_cost(cfg, c) + best[b - c];
            if (v < 0.97 * best[b]) {  // a split has to be worth it
                best[b] = v;
                first[b] = c;
            }
        }
    }
    for (int b = rem; b > 0 && n < max_sizes; b -= first[b]) sizes[n++] = first[b];
    return n;
}
constexpr int kMaxChunks = 80;  // 4096 utterances (check_bt) in rounds of >= 64, plus the remainder's few calls
// the core's scratch behind the per-batch planes: sized for the largest chunk's plan (the chunks reuse it)
static void core_carve_chunks(Carver& cv, const fsn_fullsubnet_cfg* cfg, int B, int T) {
    int sizes[kMaxChunks];
    const int n = core_chunks(cfg, B, sizes, kMaxChunks);
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
        if (i > 0 && sizes[i] == sizes[i - 1]) continue;
        Carver c2(nullptr);
        core_carve(c2, core_dims(cfg, sizes[i], T), cfg->norm_type);
        most = c2.off > most ? c2.off : most;
    }
    cv.take<char>(most);
}
// run_core over the chunks; `scratch` = a region of at least core_carve_chunks' size
static int run_core_chunks(const fsn_fullsubnet_cfg* cfg, const float* pk, const float* magT, int B, int T, void* scratch,
                           float* crm_r, float* crm_i, hipStream_t s) {
    int sizes[kMaxChunks];
    const int n = core_chunks(cfg, B, sizes, kMaxChunks);
    int b0 = 0;
    for (int i = 0; i < n; ++i) {
        const int b = sizes[i];
        const CoreDims d = core_dims(cfg, b, T);
        Carver cv(scratch);
        const CoreWs w = core_carve(cv, d, cfg->norm_type);
        FSN_TRY(run_core(cfg, pk, magT + (size_t)b0 * d.Tp * d.FP, d, w, crm_r + (size_t)b0 * d.T * d.FP,
                         crm_i + (size_t)b0 * d.T * d.FP, s, false));
        b0 += b;
    }
    FSN_REQUIRE(b0 == B, "internal: the chunks cover %d of %d utterances", b0, B);
    return FSN_OK;
}

// test hook: the utterance counts of the core calls a batch of B runs as (sum = B); returns their number
extern "C" int fsn_debug_core_chunks(const fsn_fullsubnet_cfg* cfg, int B, int* sizes, int max_sizes) {
    if (check_cfg(cfg) != FSN_OK || B < 1 || B > 4096 || !sizes || max_sizes < kMaxChunks) return -1;
    return core_chunks(cfg, B, sizes, kMaxChunks);
}

extern "C" int fsn_debug_core_plan(const fsn_fullsubnet_cfg* cfg, int B, int T, int* plan, int n) {
    if (check_cfg(cfg) != FSN_OK || check_bt(B, T) != FSN_OK || !plan || n < 8) return -1;
    int sizes[kMaxChunks];
    const int chunks = core_chunks(cfg, B, sizes, kMaxChunks);
    const CoreDims d = core_dims(cfg, chunks > 0 ? sizes[0] : B, T);
    plan[0] = d.N;
    plan[1] = d.rec.tiles;
    plan[2] = d.rec.rt;
    plan[3] = d.rec.main_wgs;
    plan[4] = d.rec.left_tiles;
    plan[5] = d.grp_clusters;
    plan[6] = d.fb_chain ? 1 : 0;
    plan[7] = chunks;
    if (n >= 9) {  // rows on the persistent recurrent pair over ALL chunks (whole rounds and a remainder have different plans)
        long rows = 0;
        for (int c = 0; c < (chunks > 0 ? chunks : 1); ++c) {
            const CoreDims dc = core_dims(cfg, chunks > 0 ? sizes[c] : B, T);
            rows += (long)dc.rec.main_wgs * dc.rec.rt * 16;
        }
        plan[8] = (int)rows;
    }
    return FSN_OK;
}

extern "C" size_t fsn_fullsubnet_workspace_bytes(const fsn_fullsubnet_cfg* cfg, int B, int T) {
    if (check_cfg(cfg) != FSN_OK || check_bt(B, T) != FSN_OK) return 0;
    const CoreDims d = core_dims(cfg, B, T);
    Carver cv(nullptr);
    cv.take<float>((size_t)B * d.Tp * d.FP);     // magT
    cv.take<float>((size_t)B * d.T * d.FP);      // crm_r
    cv.take<float>((size_t)B * d.T * d.FP);      // crm_i
    // the whole batch's plan (what the stage-level entries carve) is never smaller than a chunk's; both are checked
    Carver whole(nullptr), parts(nullptr);
    core_carve(whole, d, cfg->norm_type);
    core_carve_chunks(parts, cfg, B, T);
    cv.take<char>(whole.off > parts.off ? whole.off : parts.off);
    return fsn_round_up_sz(cv.off, 256);
}

extern "C" int fsn_fullsubnet_forward(const fsn_fullsubnet_cfg* cfg, const void* packed, const float* noisy_mag,
                                      int B, int T, float* crm_out, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_cfg(cfg));
    FSN_TRY(check_bt(B, T));
    FSN_REQUIRE(packed && noisy_mag && crm_out && workspace, "NULL pointer argument");
    const size_t need = fsn_fullsubnet_workspace_bytes(cfg, B, T);
    if (workspace_bytes < need) {
        fsn_set_error("workspace too small: %zu < %zu bytes", workspace_bytes, need);
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CoreDims d = core_dims(cfg, B, T);
    Carver cv(workspace);
    float* magT = cv.take<float>((size_t)B * d.Tp * d.FP);
    float* crm_r = cv.take<float>((size_t)B * d.T * d.FP);
    float* crm_i = cv.take<float>((size_t)B * d.T * d.FP);
    void* scratch = cv.take<char>(0);  // the rest: the core's scratch (fsn_fullsubnet_workspace_bytes)
    prof_reset();
    // [B,1,F,T] -> frame-major [B][Tp][FP]; look-ahead frames (model.py:85) and padded bins are zeros
    FSN_TRY(fsn_launch_transpose(noisy_mag, magT, B, d.FP, d.Tp, T, (long)d.F * T, d.FP, (long)d.Tp * d.FP, d.F, T, s));
    FSN_TRY(run_core_chunks(cfg, static_cast<const float*>(packed), magT, B, T, scratch, crm_r, crm_i, s));
    // frame-major planes -> [B, 2, F, T] (model.py:129-135)
    FSN_TRY(fsn_launch_transpose(crm_r, crm_out, B, T, d.F, d.FP, (long)T * d.FP, T, 2L * d.F * T, T, d.F, s));
    FSN_TRY(fsn_launch_transpose(crm_i, crm_out + (size_t)d.F * T, B, T, d.F, d.FP, (long)T * d.FP, T, 2L * d.F * T,
                                 T, d.F, s));
    return FSN_OK;
}

// ---- the full-band stage alone: model.py:85-95 ---------------------------------------------------
// look-ahead pad -> norm -> fb_model, i.e. the tensor `fb_output` of model.py:95 in the reference's layout
// [B, F, T + look_ahead].  Stage-level parity checks read it; a batch-sharded full-band model would too.
extern "C" int fsn_fullsubnet_fullband(const fsn_fullsubnet_cfg* cfg, const void* packed, const float* noisy_mag,
                                       int B, int T, float* fb_output, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_cfg(cfg));
    FSN_TRY(check_bt(B, T));
    FSN_REQUIRE(packed && noisy_mag && fb_output && workspace, "NULL pointer argument");
    const size_t need = fsn_fullsubnet_workspace_bytes(cfg, B, T);
    if (workspace_bytes < need) {
        fsn_set_error("workspace too small: %zu < %zu bytes", workspace_bytes, need);
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CoreDims d = core_dims(cfg, B, T);
    Carver cv(workspace);
    float* magT = cv.take<float>((size_t)B * d.Tp * d.FP);
    cv.take<float>((size_t)B * d.T * d.FP);
    cv.take<float>((size_t)B * d.T * d.FP);
    const CoreWs w = core_carve(cv, d, cfg->norm_type);
    prof_reset();
    FSN_TRY(fsn_launch_transpose(noisy_mag, magT, B, d.FP, d.Tp, T, (long)d.F * T, d.FP, (long)d.Tp * d.FP, d.F, T, s));
    FSN_TRY(run_core(cfg, static_cast<const float*>(packed), magT, d, w, nullptr, nullptr, s, true));
    // frame-major [B][T'][FP] -> [B, F, T']
    FSN_TRY(fsn_launch_transpose(w.fb_out, fb_output, B, d.Tp, d.F, d.FP, (long)d.Tp * d.FP, d.Tp, (long)d.F * d.Tp,
                                 d.Tp, d.F, s));
    return FSN_OK;
}

// ---- row-range form: the sub-band model on a contiguous slice of the flattened (b, f) rows -------------------
// "rank r owns a contiguous slice of the flattened (b, f) index space".  Only the utterances the
// slice touches are looked at: their full-band model and norm statistics are computed whole (they couple all
// bins of an utterance), the sub-band model only on rows [row_begin, row_end).  A slice that is aligned to
// utterances is exactly fsn_fullsubnet_forward on those utterances.
struct RowSlice {
    int b_lo, Bs;
    long r0, n;
};
static int row_slice(const fsn_fullsubnet_cfg* cfg, int B, long row_begin, long row_end, RowSlice* out) {
    const long F = cfg->num_freqs;
    FSN_REQUIRE(row_begin >= 0 && row_begin < row_end && row_end <= (long)B * F,
                "row range [%ld, %ld) is not inside the %ld sub-band rows of the batch", row_begin, row_end, (long)B * F);
    out->b_lo = (int)(row_begin / F);
    out->Bs = (int)((row_end - 1) / F) - out->b_lo + 1;
    out->r0 = row_begin - (long)out->b_lo * F;
    out->n = row_end - row_begin;
    return FSN_OK;
}

extern "C" size_t fsn_fullsubnet_rows_workspace_bytes(const fsn_fullsubnet_cfg* cfg, int B, int T, long row_begin,
                                                      long row_end) {
    RowSlice r;
    if (check_cfg(cfg) != FSN_OK || check_bt(B, T) != FSN_OK || row_slice(cfg, B, row_begin, row_end, &r) != FSN_OK)
        return 0;
    const CoreDims d = core_dims(cfg, r.Bs, T, r.r0, r.n);
    Carver cv(nullptr);
    cv.take<float>((size_t)r.Bs * d.Tp * d.FP);  // magT
    cv.take<float>((size_t)r.Bs * d.T * d.FP);   // crm_r
    cv.take<float>((size_t)r.Bs * d.T * d.FP);   // crm_i
    core_carve(cv, d, cfg->norm_type);
    return fsn_round_up_sz(cv.off, 256);
}

extern "C" int fsn_fullsubnet_forward_rows(const fsn_fullsubnet_cfg* cfg, const void* packed, const float* noisy_mag,
                                           int B, int T, long row_begin, long row_end, float* crm_rows,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_cfg(cfg));
    FSN_TRY(check_bt(B, T));
    RowSlice r;
    FSN_TRY(row_slice(cfg, B, row_begin, row_end, &r));
    FSN_REQUIRE(packed && noisy_mag && crm_rows && workspace, "NULL pointer argument");
    const size_t need = fsn_fullsubnet_rows_workspace_bytes(cfg, B, T, row_begin, row_end);
    if (workspace_bytes < need) {
        fsn_set_error("workspace too small: %zu < %zu bytes", workspace_bytes, need);
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CoreDims d = core_dims(cfg, r.Bs, T, r.r0, r.n);
    Carver cv(workspace);
    float* magT = cv.take<float>((size_t)r.Bs * d.Tp * d.FP);
    float* crm_r = cv.take<float>((size_t)r.Bs * d.T * d.FP);
    float* crm_i = cv.take<float>((size_t)r.Bs * d.T * d.FP);
    const CoreWs w = core_carve(cv, d, cfg->norm_type);
    prof_reset();
    const float* mag_lo = noisy_mag + (size_t)r.b_lo * d.F * T;  // [B, 1, F, T]: utterances are contiguous
    FSN_TRY(fsn_launch_transpose(mag_lo, magT, r.Bs, d.FP, d.Tp, T, (long)d.F * T, d.FP, (long)d.Tp * d.FP, d.F, T, s));
    FSN_TRY(run_core(cfg, static_cast<const float*>(packed), magT, d, w, crm_r, crm_i, s));
    // this slice's rows of the frame-major planes -> [row][2][T] (what the ranks all-gather)
    FSN_TRY(fsn_launch_crm_rows(crm_r, crm_i, crm_rows, r.r0, r.n, d.F, d.FP, T, s));
    return FSN_OK;
}


// ---- streaming: k more frames of the model with carried state -------------------------------------
// State (caller-owned, zero-filled for a new stream): (h, c) of the four LSTM layers and the running sums
// of the two cumulative Laplace norms.
struct StreamState {
    float *fb_h0, *fb_h1, *fb_c0, *fb_c1, *sb_h0, *sb_h1, *sb_c0, *sb_c1;
    double *fb_sum, *sb_sum;
};
static StreamState stream_carve(Carver& cv, const fsn_fullsubnet_cfg* cfg, int B) {
    StreamState st;
    const size_t nfb = (size_t)fsn_round_up(B, 16) * cfg->fb_hidden;
    const size_t nsb = (size_t)fsn_round_up(B * cfg->num_freqs, 16) * cfg->sb_hidden;
    st.fb_h0 = cv.take<float>(nfb);
    st.fb_h1 = cv.take<float>(nfb);
    st.fb_c0 = cv.take<float>(nfb);
    st.fb_c1 = cv.take<float>(nfb);
    st.sb_h0 = cv.take<float>(nsb);
    st.sb_h1 = cv.take<float>(nsb);
    st.sb_c0 = cv.take<float>(nsb);
    st.sb_c1 = cv.take<float>(nsb);
    st.fb_sum = cv.take<double>((size_t)B);
    st.sb_sum = cv.take<double>((size_t)B * cfg->num_freqs);
    return st;
}
struct StreamWs {
    float *magT, *crm_r, *crm_i, *den_fb, *gx_fb, *hseq_fb0, *hseq_fb1, *fb_out, *den_sb, *gx_sb, *hseq_sb0, *hseq_sb1;
};
static StreamWs stream_ws_carve(Carver& cv, const fsn_fullsubnet_cfg* cfg, int B, int k) {
    StreamWs w;
    const int FP = fsn_fpad(cfg->num_freqs), Npad_fb = fsn_round_up(B, 16), Npad = fsn_round_up(B * cfg->num_freqs, 16);
    const size_t plane = (size_t)B * k * FP;
    w.magT = cv.take<float>(plane);
    w.crm_r = cv.take<float>(plane);
    w.crm_i = cv.take<float>(plane);
    w.den_fb = cv.take<float>((size_t)B * k);
    w.gx_fb = cv.take<float>((size_t)k * Npad_fb * 4 * cfg->fb_hidden);
    w.hseq_fb0 = cv.take<float>((size_t)k * Npad_fb * cfg->fb_hidden);
    w.hseq_fb1 = cv.take<float>((size_t)k * Npad_fb * cfg->fb_hidden);
    w.fb_out = cv.take<float>(plane);
    w.den_sb = cv.take<float>((size_t)k * Npad);
    w.gx_sb = cv.take<float>((size_t)k * Npad * 4 * cfg->sb_hidden);
    w.hseq_sb0 = cv.take<float>((size_t)k * Npad * cfg->sb_hidden);
    w.hseq_sb1 = cv.take<float>((size_t)k * Npad * cfg->sb_hidden);
    return w;
}
static int check_stream(const fsn_fullsubnet_cfg* cfg, int B, int k) {
    FSN_TRY(check_cfg(cfg));
    FSN_REQUIRE(cfg->norm_type == FSN_NORM_CUMULATIVE_LAPLACE, "streaming needs the causal norm (FSN_NORM_CUMULATIVE_LAPLACE)");
    FSN_REQUIRE(B >= 1 && B <= 4096 && k >= 1 && k <= 4096, "streaming: batch %d / frames %d out of range", B, k);
    return FSN_OK;
}
extern "C" size_t fsn_fullsubnet_stream_state_bytes(const fsn_fullsubnet_cfg* cfg, int B) {
    if (check_stream(cfg, B, 1) != FSN_OK) return 0;
    Carver cv(nullptr);
    stream_carve(cv, cfg, B);
    return fsn_round_up_sz(cv.off, 256);
}
extern "C" size_t fsn_fullsubnet_stream_workspace_bytes(const fsn_fullsubnet_cfg* cfg, int B, int k) {
    if (check_stream(cfg, B, k) != FSN_OK) return 0;
    Carver cv(nullptr);
    stream_ws_carve(cv, cfg, B, k);
    return fsn_round_up_sz(cv.off, 256);
}

extern "C" int fsn_fullsubnet_stream_step(const fsn_fullsubnet_cfg* cfg, const void* packed, void* state,
                                          size_t state_bytes, int steps_done, const float* mag, int B, int k,
                                          float* crm_out, void* workspace, size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_stream(cfg, B, k));
    FSN_REQUIRE(packed && state && mag && crm_out && workspace && steps_done >= 0, "NULL pointer argument / negative step count");
    if (state_bytes < fsn_fullsubnet_stream_state_bytes(cfg, B) ||
        workspace_bytes < fsn_fullsubnet_stream_workspace_bytes(cfg, B, k)) {
        fsn_set_error("streaming: state / workspace buffer too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Packed p = packed_layout(cfg);
    const float* pk = static_cast<const float*>(packed);
    const int F = cfg->num_freqs, FP = fsn_fpad(F), Hf = cfg->fb_hidden, Hs = cfg->sb_hidden, nb = cfg->sb_num_neighbors;
    const int Npad_fb = fsn_round_up(B, 16), N = B * F, Npad = fsn_round_up(N, 16);
    Carver cs(state), cw(workspace);
    const StreamState st = stream_carve(cs, cfg, B);
    const StreamWs w = stream_ws_carve(cw, cfg, B, k);
    // [B, 1, F, k] -> frame-major [B][k][FP]
    FSN_TRY(fsn_launch_transpose(mag, w.magT, B, FP, k, k, (long)F * k, FP, (long)k * FP, F, k, s));
    FSN_TRY(fsn_launch_cumulative_den_fb(w.magT, w.den_fb, B, k, F, FP, s, st.fb_sum, steps_done));
    FsnGemmA a{};
    FsnGemmC c{};
    a.kind = 1;
    a.p0 = w.magT;
    a.den = w.den_fb;
    a.den_mode = 1;
    a.B = B;
    a.Tp = k;
    a.F = F;
    a.FP = FP;
    a.Npad = Npad_fb;
    c.kind = 0;
    c.p0 = w.gx_fb;
    c.bias = pk + p.fb_b0;
    const int fb_rt = k * Npad_fb / 16;
    FSN_TRY(fsn_launch_gemm(a, pk + p.fb_wih0, c, fb_rt, 4 * Hf / 16, FP / 16, s));
    FSN_TRY(fsn_launch_lstm_wavefront2(w.gx_fb, Npad_fb / 16, 0, pk + p.fb_whh0, pk + p.fb_wih1, pk + p.fb_b1_frag,
                                       pk + p.fb_whh1, w.hseq_fb0, w.hseq_fb1, Npad_fb, 0, st.fb_c0, st.fb_c1, k,
                                       Npad_fb / 16, Hf, s, st.fb_h0, st.fb_h1));
    a = FsnGemmA{};
    c = FsnGemmC{};
    a.kind = 0;
    a.p0 = w.hseq_fb1;
    a.ld = Hf;
    c.kind = 1;
    c.p0 = w.fb_out;
    c.bias = pk + p.fb_fcb;
    c.B = B;
    c.Tp = k;
    c.F = F;
    c.FP = FP;
    c.Npad = Npad_fb;
    FSN_TRY(fsn_launch_gemm(a, pk + p.fb_fc, c, fb_rt, FP / 16, Hf / 16, s));
    FSN_TRY(fsn_launch_cumulative_den_sb(w.magT, w.fb_out, w.den_sb, B, k, F, FP, nb, Npad, s, st.sb_sum, steps_done));
    a = FsnGemmA{};
    c = FsnGemmC{};
    a.kind = 2;
    a.p0 = w.magT;
    a.p1 = w.fb_out;
    a.den = w.den_sb;
    a.den_mode = 1;
    a.den_stride = Npad;
    a.B = B;
    a.Tp = k;
    a.F = F;
    a.FP = FP;
    a.Npad = Npad;
    a.n_offset = 0;
    a.N = N;
    a.nb = nb;
    c.kind = 0;
    c.p0 = w.gx_sb;
    c.bias = pk + p.sb_b0;
    const int sb_rt = (int)((long)k * Npad / 16);
    FSN_TRY(fsn_launch_gemm(a, pk + p.sb_wih0, c, sb_rt, 4 * Hs / 16, p.sb_kin_pad / 16, s));
    FSN_TRY(fsn_launch_lstm_wavefront2(w.gx_sb, Npad / 16, 0, pk + p.sb_whh0, pk + p.sb_wih1, pk + p.sb_b1_frag,
                                       pk + p.sb_whh1, w.hseq_sb0, w.hseq_sb1, Npad, 0, st.sb_c0, st.sb_c1, k, Npad / 16,
                                       Hs, s, st.sb_h0, st.sb_h1));
    a = FsnGemmA{};
    c = FsnGemmC{};
    a.kind = 0;
    a.p0 = w.hseq_sb1;
    a.ld = Hs;
    c.kind = 2;
    c.p0 = w.crm_r;
    c.p1 = w.crm_i;
    c.bias = pk + p.sb_fcb;
    c.T = k;
    c.F = F;
    c.FP = FP;
    c.Npad = Npad;
    c.N = N;
    c.la = 0;  // every model step is handed back; the caller matches step s to output frame s - look_ahead
    FSN_TRY(fsn_launch_gemm(a, pk + p.sb_fc, c, sb_rt, 1, Hs / 16, s));
    FSN_TRY(fsn_launch_transpose(w.crm_r, crm_out, B, k, F, FP, (long)k * FP, k, 2L * F * k, k, F, s));
    FSN_TRY(fsn_launch_transpose(w.crm_i, crm_out + (size_t)F * k, B, k, F, FP, (long)k * FP, k, 2L * F * k, k, F, s));
    return FSN_OK;
}

// ---- STFT / iSTFT boundary -------------------------------------------------------------------
static bool fast_fft(int n_fft, int hop) { return n_fft == 512 && hop == 256; }

// fsn_enhance's fused path is built for the FullSubNet recipe's transform only
static int check_fft(int n_fft, int hop, int win_length) {
    FSN_REQUIRE(n_fft == 512 && hop == 256 && win_length == 512,
                "fsn_enhance: only n_fft = win_length = 512, hop = 256 is built (got %d/%d/%d)", n_fft, win_length, hop);
    return FSN_OK;
}

// fsn_stft / fsn_istft: 512 / 256 on the radix-8 kernels, any other even size / hop on the direct DFT
static int check_fft_generic(int n_fft, int hop, int win_length) {
    FSN_REQUIRE(win_length == n_fft, "win_length %d != n_fft %d is not built", win_length, n_fft);
    FSN_REQUIRE(n_fft >= 16 && n_fft <= 4096 && n_fft % 2 == 0, "n_fft %d: need an even size in [16, 4096]", n_fft);
    FSN_REQUIRE(hop >= 1 && hop <= n_fft, "hop %d out of range for n_fft %d", hop, n_fft);
    return FSN_OK;
}

extern "C" int fsn_stft(const float* y, int B, int L, int n_fft, int hop, int win_length, const float* window,
                        float* real, float* imag, float* mag, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_fft_generic(n_fft, hop, win_length));
    FSN_REQUIRE(y && window, "NULL pointer argument");
    FSN_REQUIRE(B >= 1 && L > n_fft / 2, "need B >= 1 and L > n_fft/2 (reflect padding), got B=%d L=%d", B, L);
    const int T = 1 + L / hop, F = n_fft / 2 + 1;
    FSN_REQUIRE((long)B * T <= 0x7fffffffL, "too many frames");
    if (!fast_fft(n_fft, hop))
        return fsn_launch_dft_stft(y, B, L, window, real, imag, mag, T, n_fft, hop, static_cast<hipStream_t>(stream));
    return fsn_launch_stft(y, B, L, window, real, imag, mag, T, T, F, fsn_fpad(F), false,
                           static_cast<hipStream_t>(stream));
}

extern "C" size_t fsn_istft_workspace_bytes(int B, int T, int n_fft) {
    if (B < 1 || T < 1 || n_fft < 16 || n_fft > 4096 || n_fft % 2) return 0;
    return fsn_round_up_sz((size_t)B * T * n_fft * sizeof(float), 256);
}

extern "C" int fsn_istft(const float* real, const float* imag, int B, int T, int n_fft, int hop, int win_length,
                         const float* window, int length, float* y, void* workspace, size_t workspace_bytes,
                         void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_fft_generic(n_fft, hop, win_length));
    FSN_TRY(check_bt(B, T));
    FSN_REQUIRE(real && imag && window && y && workspace, "NULL pointer argument");
    FSN_REQUIRE(length >= 1, "length %d < 1", length);
    if (workspace_bytes < fsn_istft_workspace_bytes(B, T, n_fft)) {
        fsn_set_error("workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int F = n_fft / 2 + 1;
    float* wf = static_cast<float*>(workspace);
    if (!fast_fft(n_fft, hop)) return fsn_launch_dft_istft(real, imag, window, wf, y, B, T, n_fft, hop, length, s);
    FSN_TRY(fsn_launch_mask_irfft(real, imag, nullptr, nullptr, B, T, F, fsn_fpad(F), false, window, wf, s));
    return fsn_launch_ola(wf, window, B, T, length, y, s);
}

// ---- elementwise boundary --------------------------------------------------------------------
extern "C" int fsn_decompress_cirm(const float* mask, float* out, size_t n, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(mask && out, "NULL pointer argument");
    return n ? fsn_launch_decompress(mask, out, n, static_cast<hipStream_t>(stream)) : FSN_OK;
}
extern "C" int fsn_compress_cirm(const float* mask, float* out, size_t n, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(mask && out, "NULL pointer argument");
    return n ? fsn_launch_compress(mask, out, n, static_cast<hipStream_t>(stream)) : FSN_OK;
}
extern "C" int fsn_build_cirm(const float* nr, const float* ni, const float* cr, const float* ci, float* out,
                              size_t n, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(nr && ni && cr && ci && out, "NULL pointer argument");
    return n ? fsn_launch_build_cirm(nr, ni, cr, ci, out, n, static_cast<hipStream_t>(stream)) : FSN_OK;
}

// ---- the whole path: inferencer.py:130-145 ---------------------------------------------------
extern "C" size_t fsn_enhance_workspace_bytes(const fsn_fullsubnet_cfg* cfg, int B, int L, int n_fft, int hop) {
    if (check_cfg(cfg) != FSN_OK || check_fft(n_fft, hop, n_fft) != FSN_OK || B < 1 || L <= n_fft / 2) return 0;
    const int T = 1 + L / hop;
    if (check_bt(B, T) != FSN_OK || cfg->num_freqs != n_fft / 2 + 1) return 0;
    const CoreDims d = core_dims(cfg, B, T);
    Carver cv(nullptr);
    cv.take<float>((size_t)B * d.Tp * d.FP);  // magT
    cv.take<float>((size_t)B * d.T * d.FP);   // re
    cv.take<float>((size_t)B * d.T * d.FP);   // im
    cv.take<float>((size_t)B * d.T * d.FP);   // crm_r
    cv.take<float>((size_t)B * d.T * d.FP);   // crm_i
    cv.take<float>((size_t)B * d.T * n_fft);  // windowed frames
    Carver whole(nullptr), parts(nullptr);
    core_carve(whole, d, cfg->norm_type);
    core_carve_chunks(parts, cfg, B, T);
    cv.take<char>(whole.off > parts.off ? whole.off : parts.off);
    return fsn_round_up_sz(cv.off, 256);
}

extern "C" int fsn_enhance(const fsn_fullsubnet_cfg* cfg, const void* packed, const float* window,
                           const float* noisy, int B, int L, int n_fft, int hop, float* enhanced, float* crm_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_cfg(cfg));
    FSN_TRY(check_fft(n_fft, hop, n_fft));
    FSN_REQUIRE(packed && window && noisy && enhanced && workspace, "NULL pointer argument");
    FSN_REQUIRE(B >= 1 && L > n_fft / 2, "need B >= 1 and L > n_fft/2, got B=%d L=%d", B, L);
    FSN_REQUIRE(cfg->num_freqs == n_fft / 2 + 1, "num_freqs %d != n_fft/2+1", cfg->num_freqs);
    const int T = 1 + L / hop;
    FSN_TRY(check_bt(B, T));
    const size_t need = fsn_enhance_workspace_bytes(cfg, B, L, n_fft, hop);
    if (workspace_bytes < need) {
        fsn_set_error("workspace too small: %zu < %zu bytes", workspace_bytes, need);
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CoreDims d = core_dims(cfg, B, T);
    Carver cv(workspace);
    float* magT = cv.take<float>((size_t)B * d.Tp * d.FP);
    float* re = cv.take<float>((size_t)B * d.T * d.FP);
    float* im = cv.take<float>((size_t)B * d.T * d.FP);
    float* crm_r = cv.take<float>((size_t)B * d.T * d.FP);
    float* crm_i = cv.take<float>((size_t)B * d.T * d.FP);
    float* wf = cv.take<float>((size_t)B * d.T * n_fft);
    void* scratch = cv.take<char>(0);  // the rest: the core's scratch (fsn_enhance_workspace_bytes)
    prof_reset();
    {
        StageTimer st(ST_STFT, s);
        FSN_TRY(fsn_launch_stft(noisy, B, L, window, re, im, magT, d.T, d.Tp, d.F, d.FP, true, s));
    }
    FSN_TRY(run_core_chunks(cfg, static_cast<const float*>(packed), magT, B, T, scratch, crm_r, crm_i, s));
    {
        StageTimer st(ST_MASK_ISTFT, s);
        FSN_TRY(fsn_launch_mask_irfft(re, im, crm_r, crm_i, B, d.T, d.F, d.FP, true, window, wf, s));
        FSN_TRY(fsn_launch_ola(wf, window, B, d.T, L, enhanced, s));
    }
    if (crm_out) {
        FSN_TRY(fsn_launch_transpose(crm_r, crm_out, B, d.T, d.F, d.FP, (long)d.T * d.FP, d.T, 2L * d.F * d.T, d.T,
                                     d.F, s));
        FSN_TRY(fsn_launch_transpose(crm_i, crm_out + (size_t)d.F * d.T, B, d.T, d.F, d.FP, (long)d.T * d.FP, d.T,
                                     2L * d.F * d.T, d.T, d.F, s));
    }
    return FSN_OK;
}

// ---- training step: one nn.LSTM layer, forward with saved activations + BPTT ---------------------
// (recipes/dns_interspeech_2020/fullsubnet/trainer.py:56-63 through sequence_model.py:52-58)
static int check_lstm_layer(int T, int N, int I, int H, long ldx) {
    FSN_REQUIRE(T >= 1 && N >= 16 && N % 16 == 0, "lstm layer: need T >= 1 and N a positive multiple of 16 (got %d, %d)", T, N);
    FSN_REQUIRE(I >= 1 && H >= 64 && H % 64 == 0, "lstm layer: need I >= 1 and H a multiple of 64 (got %d, %d)", I, H);
    FSN_REQUIRE(ldx >= fsn_round_up(I, 16) && ldx % 4 == 0, "lstm layer: x row stride %ld must be >= round_up(I,16) and 16-byte aligned", ldx);
    return FSN_OK;
}

extern "C" size_t fsn_lstm_layer_save_bytes(int T, int N, int H) {
    return fsn_round_up_sz(((size_t)T * N * 4 * H + (size_t)T * N * H) * sizeof(float), 256);
}
extern "C" size_t fsn_lstm_layer_fwd_workspace_bytes(int T, int N, int I, int H) {
    Carver cv(nullptr);
    cv.take<float>((size_t)4 * H * fsn_round_up(I, 16));
    cv.take<float>((size_t)4 * H * H);
    cv.take<float>((size_t)4 * H);
    cv.take<float>((size_t)T * N * 4 * H);
    cv.take<float>((size_t)N * H);  // cell state of the step kernels (inference mode)
    return fsn_round_up_sz(cv.off, 256);
}

// Row split of a stand-alone layer in inference mode: the persistent kernel (built for H = 384) takes
// whole rounds of 16 RT-row tiles on all CUs, everything else goes step by step.
static FsnRecPlan layer_plan(int N, int H) {
    FsnRecPlan p{};
    p.tiles = N / 16;
    p.npad = N;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    if (H != 384 || p.tiles < cus / 4) {
        p.left_tiles = p.tiles;
    } else if (p.tiles <= cus) {
        p.rt = 1;
        p.main_wgs = p.tiles;
    } else {
        // whole rounds on all CUs + left-over tiles step by step beside them ...
        p.rt = p.tiles / cus < 5 ? p.tiles / cus : 5;
        p.main_wgs = cus;
        p.left_tiles = p.tiles - cus * p.rt;
        // ... or FEWER workgroups with one more tile each and nothing left over (the workgroups are independent: a launch
        // takes its tiles-per-workgroup's time whatever its grid).  Measured on Fast FullSubNet's bottleneck (96 steps, round
        // 6): ~11 ms per tile of a workgroup, ~0.11 ms per left-over tile - 448 tiles as 256 x 1 + 192 left over 32.2 ms, as
        // 224 x 2 what 512 tiles take (25.4); 288 tiles stay 256 x 1 + 32 (21.0 against 24.6).
        const int rt2 = (p.tiles + cus - 1) / cus;
        if (p.left_tiles > 0 && rt2 <= 4 && p.tiles % rt2 == 0 && 100 * rt2 < 100 * p.rt + p.left_tiles) {
            p.rt = rt2;
            p.main_wgs = p.tiles / rt2;
            p.left_tiles = 0;
        }
    }
    return p;
}
// The same split with SEVERAL whole rounds (more than four row tiles per CU: layer_plan stops at one round of five and hands
// everything beyond to the step kernels - 96 / 128 utterances of a composed FullSubNet were 518 / 776 left-over tiles, 151 / 216 ms
// per model call): rounds of 2 - 4 tiles per workgroup on every CU, as many as fit, the rest (fewer than one round) left over.
static FsnRecPlan layer_plan_rounds(int N, int H) {
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    const int tiles = N / 16;
    if (H != 384 || tiles < 2 * cus) return layer_plan(N, H);
    FsnRecPlan p{};
    p.tiles = tiles;
    p.npad = N;
    long best = -1;
    for (int rt = 4; rt >= 2; --rt) {
        const int rounds = tiles / (cus * rt);
        if (rounds < 1) continue;
        const int left = tiles - rounds * cus * rt;
        const long cost = (long)rounds * rt * 100 + left;  // a left-over tile: about a hundredth of a tile of a resident workgroup
        if (best < 0 || cost < best) {
            best = cost;
            p.rt = rt;
            p.main_wgs = rounds * cus;
            p.left_tiles = left;
        }
    }
    return p;
}
// Rows (a multiple of 16, >= N) a caller that owns the row padding should give a stand-alone layer of N rows: the next count
// whose plan has no left-over tiles when that is the cheaper plan by the measure above, N itself otherwise.
extern "C" int fsn_lstm_layer_plan_rows(int N, int H) {
    if (N < 1) return N;
    const int n16 = fsn_round_up(N, 16);
    const FsnRecPlan p = layer_plan(n16, H);
    if (p.main_wgs <= 0 || p.left_tiles == 0) return n16;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    const int rt2 = (p.tiles + cus - 1) / cus;
    if (rt2 > 4) return n16;
    const int padded = (p.tiles + rt2 - 1) / rt2 * rt2;
    return 100 * rt2 < 100 * p.rt + p.left_tiles ? padded * 16 : n16;
}

extern "C" int fsn_lstm_layer_forward(const float* x, long ldx, const float* w_ih, const float* w_hh,
                                      const float* b_ih, const float* b_hh, int T, int N, int I, int H, float* hseq,
                                      void* save, size_t save_bytes, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    FSN_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && hseq && workspace, "NULL pointer argument");
    if ((save && save_bytes < fsn_lstm_layer_save_bytes(T, N, H)) ||
        workspace_bytes < fsn_lstm_layer_fwd_workspace_bytes(T, N, I, H)) {
        fsn_set_error("lstm layer forward: save / workspace buffer too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ipad = fsn_round_up(I, 16);
    Carver cv(workspace);
    float* wih_p = cv.take<float>((size_t)4 * H * Ipad);
    float* whh_p = cv.take<float>((size_t)4 * H * H);
    float* bias = cv.take<float>((size_t)4 * H);
    float* gx = cv.take<float>((size_t)T * N * 4 * H);
    float* c_state = cv.take<float>((size_t)N * H);
    float* gates = static_cast<float*>(save);
    float* cseq = gates + (size_t)T * N * 4 * H;
    FSN_TRY(fsn_launch_pack(w_ih, wih_p, 4 * H, I, 4 * H, Ipad, s));
    FSN_TRY(fsn_launch_pack(w_hh, whh_p, 4 * H, H, 4 * H, H, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih, b_hh, bias, 4 * H, 4 * H, s));
    if (!save) {
        // inference with a narrow input on the persistent kernel (e.g. Fast FullSubNet's bottleneck: 12 inputs,
        // 16 384 rows): the K <= 32 projection is formed inside the recurrent kernel from a staged LDS tile,
        // like the sub-band model's layer 0, instead of writing and re-reading a [T][N][4H] projection
        FsnRecPlan plan = layer_plan(N, H);
        if (plan.main_wgs > 0 && (plan.rt > 4 || plan.left_tiles > 16) && (Ipad <= 32 || (I == H && ldx == H)))
            plan = layer_plan_rounds(N, H);  // more than one round's worth of rows (the two forms below take any grid)
        if (plan.main_wgs > 0 && plan.left_tiles == 0 && Ipad <= 32) {
            FsnSbInput xin{};
            xin.x_rows = x;
            xin.x_ld = ldx;
            xin.x_step = N;
            xin.N = N;
            xin.kin_chunks = Ipad / 16;
            xin.wih_p = wih_p;
            xin.bias = bias;
            return run_recurrence(nullptr, &xin, nullptr, 0, 0, whh_p, hseq, c_state, T, N, H, plan, s);
        }
        // The same two forms with LEFT-OVER row tiles (whole rounds of 2 - 4 tiles per workgroup + a few tiles more: 64 x 257
        // rows are 256 x 4 tiles + 4): the persistent kernel takes the whole rounds, the left-over rows advance step by step
        // beside it (run_recurrence) from their own small projection - compact copies of their input rows, one GEMM, formed AHEAD
        // of the persistent launch (the GEMM's workgroups do not fit beside it) inside the region the full projection would
        // have taken.  Before: the full [T][N][4H] projection was written and read back for every row (a composed LSTM
        // FullSubNet at 64 x 3 s: 92 ms per model call against 80 at 62 utterances, whose tiles divide evenly).
        const bool narrow = Ipad <= 32 && ldx >= Ipad, stacked = I == H && ldx == H && fsn_lstm_rec_x_supported(H, plan.rt);
        if (plan.main_wgs > 0 && plan.left_tiles > 0 && plan.rt >= 2 && plan.rt <= 4 && (narrow || stacked) && whh_p > wih_p &&
            (size_t)plan.left_tiles * 16 * ((size_t)4 * H + Ipad) <= (size_t)N * 4 * H) {
            const int left = plan.left_tiles * 16, main_rows = N - left;
            float* gx_left = gx;                                  // [T][left / 16 tiles] fragment order
            float* x_left = gx + (size_t)T * left * 4 * H;        // [T][left][Ipad]
            bool ok = true;
            if (ldx == Ipad)
                ok = hipMemcpy2DAsync(x_left, (size_t)left * Ipad * sizeof(float), x + (size_t)main_rows * ldx,
                                      (size_t)N * ldx * sizeof(float), (size_t)left * Ipad * sizeof(float), (size_t)T,
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
            else
                for (int t = 0; t < T && ok; ++t)
                    ok = hipMemcpy2DAsync(x_left + (size_t)t * left * Ipad, (size_t)Ipad * sizeof(float),
                                          x + ((size_t)t * N + main_rows) * ldx, (size_t)ldx * sizeof(float),
                                          (size_t)Ipad * sizeof(float), (size_t)left, hipMemcpyDeviceToDevice, s) == hipSuccess;
            if (!ok) {
                fsn_set_error("lstm layer forward: copy of the left-over rows failed");
                return FSN_ERR_LAUNCH;
            }
            FsnGemmA al{};
            al.kind = 0;
            al.p0 = x_left;
            al.ld = Ipad;
            FsnGemmC cl{};
            cl.kind = 0;
            cl.p0 = gx_left;
            cl.bias = bias;
            FSN_TRY(fsn_launch_gemm(al, wih_p, cl, T * (left / 16), 4 * H / 16, Ipad / 16, s));
            float* c_left = c_state + (size_t)main_rows * H;
            if (narrow) {
                FsnSbInput xin{};
                xin.x_rows = x;
                xin.x_ld = ldx;
                xin.x_step = N;
                xin.N = main_rows;
                xin.kin_chunks = Ipad / 16;
                xin.wih_p = wih_p;
                xin.bias = bias;
                return run_recurrence(nullptr, &xin, gx_left, left / 16, 0, whh_p, hseq, c_left, T, N, H, plan, s);
            }
            return run_recurrence(nullptr, nullptr, gx_left, left / 16, 0, whh_p, hseq, c_left, T, N, H, plan, s, nullptr, -1, nullptr,
                                  nullptr, x, wih_p, bias);
        }
        // a layer of a stack on the persistent kernel (input = the hidden sequence of an equally wide layer below, e.g.
        // the second bottleneck layer of Fast FullSubNet, fast_fullsubnet/model.py:66-74): the K = H projection is
        // formed inside the recurrent kernel from x streamed through its LDS ring (lstm_rec_x_kernel<.., HSEQ>) - no
        // projection GEMM, no [T][N][4H] gx round trip
        if (plan.main_wgs > 0 && plan.left_tiles == 0 && I == H && ldx == H && fsn_lstm_rec_x_supported(H, plan.rt) &&
            whh_p > wih_p)
            return fsn_launch_lstm_rec_x(x, wih_p, whh_p, bias, T, N, H, plan.rt, plan.main_wgs, s, nullptr, hseq);
    }
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = x;
    a.ld = ldx;
    FsnGemmC c{};
    c.kind = 0;
    c.p0 = gx;
    c.bias = bias;
    FSN_TRY(fsn_launch_gemm(a, wih_p, c, T * (N / 16), 4 * H / 16, Ipad / 16, s));
    if (!save) {  // inference: nothing kept but the hidden sequence
        const FsnRecPlan plan = layer_plan(N, H);
        const long main_tiles = (long)plan.main_wgs * plan.rt;
        return run_recurrence(gx, nullptr, gx, plan.tiles, main_tiles, whh_p, hseq, c_state + main_tiles * 16 * H, T, N,
                              H, plan, s);
    }
    const size_t step = (size_t)N * H;
    for (int t = 0; t < T; ++t)
        FSN_TRY(fsn_launch_lstm_step_train(gx, whh_p, t ? hseq + (t - 1) * step : hseq, hseq + t * step,
                                           t ? cseq + (t - 1) * step : cseq, cseq + t * step,
                                           gates + (size_t)t * N * 4 * H, (long)t * (N / 16), N / 16, H, t == 0, s));
    return FSN_OK;
}

// ---- a stacked layer + the output layer that follows it, inference ---------------------------------------------------
// (sequence_model.py:106-125: `self.fc_output_layer(self.sequence_model(x))` for the LAST layer of a stack with one or two
// outputs - Fast FullSubNet's bottleneck, fast_fullsubnet/model.py:66-74: 16 384 rows x 384 units -> 1 value per step.)
// When the persistent kernel that forms the projection itself takes the layer (fsn_lstm_layer_fc_supported), its fused
// two-row output layer does the nn.Linear as well: the [T][N][H] hidden sequence is neither written nor read back.
// out0 / out1: [T][ldo] PRE-activation outputs 0 / 1 (time-major; out1 may be NULL when O == 1).
static bool lstm_layer_fc_plan(int T, int N, int I, long ldx, int H, int O, FsnRecPlan* plan) {
    if (T < 1 || N < 16 || N % 16 || H != 384 || I != H || ldx != H || O < 1 || O > 2) return false;
    const FsnRecPlan p = layer_plan(N, H);
    if (plan) *plan = p;
    return p.main_wgs > 0 && p.left_tiles == 0 && fsn_lstm_rec_x_supported(H, p.rt);
}
extern "C" int fsn_lstm_layer_fc_supported(int T, int N, int I, long ldx, int H, int O) {
    return lstm_layer_fc_plan(T, N, I, ldx, H, O, nullptr) ? 1 : 0;
}
extern "C" size_t fsn_lstm_layer_fc_workspace_bytes(int T, int N, int I, int H) {
    if (T < 1 || N < 16 || I < 1 || H < 64) return 0;
    Carver cv(nullptr);
    cv.take<float>((size_t)4 * H * fsn_round_up(I, 16));
    cv.take<float>((size_t)4 * H * H);
    cv.take<float>((size_t)4 * H);
    cv.take<float>((size_t)16 * H);      // the output layer's two rows as one packed column tile
    cv.take<float>(16);
    cv.take<float>((size_t)T * N);       // the unused second output when O == 1
    return fsn_round_up_sz(cv.off, 256);
}
extern "C" int fsn_lstm_layer_forward_fc(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                                         const float* b_hh, int T, int N, int I, int H, const float* fc_w, const float* fc_b,
                                         int O, float* out0, float* out1, long ldo, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    FSN_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && fc_w && fc_b && out0 && workspace, "NULL pointer argument");
    FsnRecPlan plan{};
    FSN_REQUIRE(lstm_layer_fc_plan(T, N, I, ldx, H, O, &plan),
                "lstm layer + output layer: not a shape of the fused form (H = I = ldx = 384, 1 or 2 outputs, whole rounds of "
                "2 - 4 row tiles per CU): ask fsn_lstm_layer_fc_supported");
    FSN_REQUIRE(ldo >= N && (O == 1 || out1), "lstm layer + output layer: ldo %ld < N or the second output is missing", ldo);
    if (workspace_bytes < fsn_lstm_layer_fc_workspace_bytes(T, N, I, H)) {
        fsn_set_error("lstm layer + output layer: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Carver cv(workspace);
    float* wih_p = cv.take<float>((size_t)4 * H * fsn_round_up(I, 16));
    float* whh_p = cv.take<float>((size_t)4 * H * H);
    float* bias = cv.take<float>((size_t)4 * H);
    float* fcw_p = cv.take<float>((size_t)16 * H);
    float* fcb_p = cv.take<float>(16);
    float* spare = cv.take<float>((size_t)T * N);
    FSN_TRY(fsn_launch_pack(w_ih, wih_p, 4 * H, I, 4 * H, fsn_round_up(I, 16), s));
    FSN_TRY(fsn_launch_pack(w_hh, whh_p, 4 * H, H, 4 * H, H, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih, b_hh, bias, 4 * H, 4 * H, s));
    FSN_TRY(fsn_launch_pack(fc_w, fcw_p, O, H, 16, H, s));
    FSN_TRY(fsn_launch_bias_sum(fc_b, nullptr, fcb_p, O, 16, s));
    FsnRecFc fc{};
    fc.w_p = fcw_p;
    fc.bias = fcb_p;
    fc.crm_r = out0;
    fc.crm_i = O > 1 ? out1 : spare;
    // the kernel's destination of row n at step t is plane[((n / F) T + t) FP + n % F]: one group of F = N rows, FP = ldo
    // -> plane[t ldo + n], time-major
    fc.N = N;
    fc.F = N;
    fc.FP = (int)ldo;
    fc.T = T;
    fc.la = 0;
    fc.row0 = 0;
    return fsn_launch_lstm_rec_x(x, wih_p, whh_p, bias, T, N, H, plan.rt, plan.main_wgs, s, &fc, nullptr);
}

// ---- training: two stacked nn.LSTM layers of equal width, forward with saved activations ----------------------
// (sequence_model.py:52-58 with num_layers = 2, under autograd: fullsubnet/trainer.py:56-63).  The result is that of two
// fsn_lstm_layer_forward calls; what it adds is the persistent kernels: the full-band shape (H = 512, up to 64 rows)
// runs on fb_chain_kernel, one launch for both layers and all steps instead of 2 T.
static bool lstm2_on_chain(int T, int N, int H) { return fsn_fb_chain_supported(H, N) && T <= fsn_fb_chain_max_steps(); }
// ONE plan for both directions of the two-layer training entries (fsn_lstm2_forward_train, fsn_lstm2_backward and their
// workspace queries).  Forward and backward may land on different kernels - every path reads and writes the one save
// layout of fsn_lstm_layer_forward (gates [T][N][4H] | cell sequence [T][N][H]) and the one hseq layout - so each
// direction only has to honour its own kernel's bounds, all of which live here:
//   fwd_group  : clusters of lstm2_group_kernel<.., TRAIN, SAVE> - H = 384, 17 - 32 input columns, 96+ row tiles that
//                fill whole 64-row clusters up to 8 left-over tiles, hidden sequence within a buffer resource's 2 GB;
//   fwd_chain  : fb_chain_kernel<.., SAVE> - H = 384 / 512, up to 64 rows, up to 4095 steps (its hand-off offsets);
//   bptt_group : clusters of lstm2_group_bptt_kernel - H = 384, the same row shape, any input width (dX is a GEMM
//                afterwards) and any T (one buffer resource per (step, cluster) tile);
//   bptt_chain : fb_chain_bptt_kernel - H = 512, 16 .. 80 rows (one chain per row tile), T below fsn_fb_chain_bptt_max_steps
//                (32-bit dx offsets).
struct Lstm2TrainPlan {
    int fwd_group, bptt_group;
    bool fwd_chain, bptt_chain;
};
static Lstm2TrainPlan lstm2_train_plan(int T, int N, int I, int H) {
    Lstm2TrainPlan p{0, 0, false, false};
    const int tiles = N / 16;
    if (H == 384 && tiles >= kWavefrontBelowTiles) {
        const int cf = fsn_lstm2_group_clusters(tiles), cb = fsn_lstm2_group_bptt_clusters(tiles);
        if (fsn_round_up(I, 16) == 32 && (size_t)T * N * H * sizeof(float) <= 0x7fffffffull && cf > 0 && tiles - 4 * cf <= 8)
            p.fwd_group = cf;
        if (cb > 0 && tiles - 4 * cb <= 8) p.bptt_group = cb;
    }
    p.fwd_chain = !p.fwd_group && lstm2_on_chain(T, N, H);
    p.bptt_chain = !p.bptt_group && fsn_fb_chain_bptt_supported(H, N) && T <= fsn_fb_chain_bptt_max_steps();
    return p;
}
static int lstm2_train_group_clusters(int T, int N, int I, int H) { return lstm2_train_plan(T, N, I, H).fwd_group; }
extern "C" int fsn_lstm2_train_is_persistent(int T, int N, int I, int H) {
    if (T < 1 || N < 16 || N % 16 || I < 1 || H < 1) return 0;
    const Lstm2TrainPlan p = lstm2_train_plan(T, N, I, H);
    return ((p.fwd_group > 0 || p.fwd_chain) && (p.bptt_group > 0 || p.bptt_chain)) ? 1 : 0;
}
// the 16-bit arithmetic has kernels of its own for the group shapes (lstm_group16_kernels.hip) when they take the same
// clusters; the flag array is sized for either family
static bool lstm2_use_g16(int arith, int clusters, int N) {
    return arith != FSN_ARITH_F32 && clusters > 0 && fsn_lstm2_g16_clusters(N / 16) >= clusters && !g_g16_off.load(std::memory_order_relaxed);
}
static size_t lstm2_group_flag_words_any(int clusters) {
    size_t a = fsn_lstm2_group_flag_words(clusters), b = fsn_lstm2_g16_flag_words(clusters), c = fsn_lstm2_group_bptt_flag_words(clusters);
    a = a > b ? a : b;
    return a > c ? a : c;
}
extern "C" size_t fsn_lstm2_train_workspace_bytes(int T, int N, int I, int H, int arith) {
    arith &= ~FSN_ARITH_SAVES16;
    const int Ipad = fsn_round_up(I, 16);
    Carver cv(nullptr);
    if (const int clusters = lstm2_train_group_clusters(T, N, I, H)) {
        const size_t left = (size_t)(N / 16 - 4 * clusters) * 16;
        cv.take<float>((size_t)4 * H * Ipad + (size_t)3 * 4 * H * H);
        cv.take<float>((size_t)2 * 4 * H);
        cv.take<unsigned>(lstm2_group_flag_words_any(clusters));
        cv.take<float>((size_t)T * left * Ipad);
        cv.take<float>((size_t)T * left * H);
        cv.take<float>((size_t)T * left * 4 * H);
        if (arith != FSN_ARITH_F32) cv.take<unsigned short>((size_t)4 * H * Ipad + (size_t)3 * 4 * H * H);  // 16-bit weights
        return fsn_round_up_sz(cv.off, 256);
    }
    if (lstm2_train_plan(T, N, I, H).fwd_chain) {
        cv.take<float>((size_t)4 * H * Ipad);
        cv.take<float>((size_t)3 * 4 * H * H);
        cv.take<float>((size_t)2 * 4 * H);
        cv.take<float>((size_t)T * N * 4 * H);
        cv.take<float>(fsn_fb_chain_exchange_floats(T, N));
        cv.take<unsigned>(fsn_fb_chain_flag_words());
        return fsn_round_up_sz(cv.off, 256);
    }
    const size_t l0 = fsn_lstm_layer_fwd_workspace_bytes(T, N, I, H), l1 = fsn_lstm_layer_fwd_workspace_bytes(T, N, H, H);
    return l0 > l1 ? l0 : l1;
}
extern "C" int fsn_lstm2_forward_train(const float* x, long ldx, const float* w_ih0, const float* w_hh0,
                                       const float* b_ih0, const float* b_hh0, const float* w_ih1, const float* w_hh1,
                                       const float* b_ih1, const float* b_hh1, int T, int N, int I, int H, float* hseq0,
                                       float* hseq1, void* save0, void* save1, size_t save_bytes, void* workspace,
                                       size_t workspace_bytes, int arith, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    const int saves16 = arith & FSN_ARITH_SAVES16;  // only meaningful with a 16-bit arithmetic; passed on to the g16 launch
    arith &= ~FSN_ARITH_SAVES16;
    FSN_REQUIRE((arith == FSN_ARITH_F32 && !saves16) || arith == FSN_ARITH_F16 || arith == FSN_ARITH_BF16,
                "lstm2 forward (training): arithmetic %d unknown (FSN_ARITH_F32 / _F16 / _BF16 [| FSN_ARITH_SAVES16])", arith | saves16);
    FSN_REQUIRE(x && w_ih0 && w_hh0 && b_ih0 && b_hh0 && w_ih1 && w_hh1 && b_ih1 && b_hh1 && hseq0 && hseq1 && save0 &&
                    save1 && workspace,
                "NULL pointer argument");
    if (save_bytes < fsn_lstm_layer_save_bytes(T, N, H) || workspace_bytes < fsn_lstm2_train_workspace_bytes(T, N, I, H, arith)) {
        fsn_set_error("lstm2 forward (training): save / workspace buffer too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ipad = fsn_round_up(I, 16);
    if (const int clusters = lstm2_train_group_clusters(T, N, I, H)) {
        // whole 64-row clusters on the group kernel (both layers, one launch); the few rows that do not fill a cluster
        // step by step on the auxiliary stream beside it, straight into the same output buffers
        const int left_tiles = N / 16 - 4 * clusters, left = left_tiles * 16, row0 = 64 * clusters;
        FSN_REQUIRE(ldx == Ipad, "lstm2 forward (training): this shape needs x rows of exactly %d columns (got %ld)", Ipad, ldx);
        Carver cv(workspace);
        float* wih0_p = cv.take<float>((size_t)4 * H * Ipad + (size_t)3 * 4 * H * H);
        float* whh0_p = wih0_p + (size_t)4 * H * Ipad;
        float* wih1_p = whh0_p + (size_t)4 * H * H;
        float* whh1_p = wih1_p + (size_t)4 * H * H;
        float* b0 = cv.take<float>((size_t)2 * 4 * H);
        float* b1 = b0 + 4 * H;
        unsigned* flags = cv.take<unsigned>(lstm2_group_flag_words_any(clusters));
        float* x_left = cv.take<float>((size_t)T * left * Ipad);
        float* h0_left = cv.take<float>((size_t)T * left * H);
        float* gx_left = cv.take<float>((size_t)T * left * 4 * H);
        const size_t wfloats = (size_t)4 * H * Ipad + (size_t)3 * 4 * H * H;
        unsigned short* w16 = arith != FSN_ARITH_F32 ? cv.take<unsigned short>(wfloats) : nullptr;
        const bool g16 = lstm2_use_g16(arith, clusters, N);
        if (!g16 || left > 0) {  // the fp32-fragment weights: the group kernel's and the step-by-step rows' (the 16-bit kernels pack their own)
            FSN_TRY(fsn_launch_pack(w_ih0, wih0_p, 4 * H, I, 4 * H, Ipad, s));
            FSN_TRY(fsn_launch_pack(w_hh0, whh0_p, 4 * H, H, 4 * H, H, s));
            FSN_TRY(fsn_launch_pack(w_ih1, wih1_p, 4 * H, H, 4 * H, H, s));
            FSN_TRY(fsn_launch_pack(w_hh1, whh1_p, 4 * H, H, 4 * H, H, s));
        }
        FSN_TRY(fsn_launch_bias_sum(b_ih0, b_hh0, b0, 4 * H, 4 * H, s));
        FSN_TRY(fsn_launch_bias_sum(b_ih1, b_hh1, b1, 4 * H, 4 * H, s));
        if (w16 && !g16) FSN_TRY(fsn_launch_to16(wih0_p, w16, wfloats, arith, s));  // the group kernel's weight fragments in 16 bits
        float* sv0 = static_cast<float*>(save0);
        float* sv1 = static_cast<float*>(save1);
        StreamCtx* cx = cur_ctx();
        if (left > 0) {
            FSN_TRY(aux_init(cx));
            if (hipEventRecord(cx->ev_fork, s) != hipSuccess || hipStreamWaitEvent(cx->aux, cx->ev_fork, 0) != hipSuccess) {
                fsn_set_error("aux stream fork failed");
                return FSN_ERR_LAUNCH;
            }
        }
        {
            FSN_PERSIST_BEGIN(s);
            if (g16) {  // the 16-bit arithmetic's own kernels: they pack the raw weights their way into w16
                FSN_TRY(fsn_launch_lstm2_g16_train(x, I, N, w_ih0, w_hh0, w_ih1, w_hh1, b0, b1, hseq0, hseq1, sv0, sv1, flags, w16,
                                                   T, clusters, H, s, arith | saves16));
                FSN_TRY(fsn_launch_poison_if(flags + fsn_lstm2_g16_status_word(clusters), hseq1, (size_t)T * N * H, s));
            } else {
                FSN_TRY(fsn_launch_lstm2_group_train(x, ldx, 32, N, wih0_p, whh0_p, wih1_p, whh1_p, b0, b1, hseq0, hseq1, sv0,
                                                     sv1, flags, T, clusters, H, s, arith, w16));
                FSN_TRY(fsn_launch_poison_if(flags + fsn_lstm2_group_status_word(clusters), hseq1, (size_t)T * N * H, s));
            }
        }
        if (left > 0) {
            hipStream_t as = cx->aux;
            const size_t stepH = (size_t)N * H, stepG = (size_t)N * 4 * H;
            for (int layer = 0; layer < 2; ++layer) {
                // the left-over rows of this layer's input as a compact [T][left][K] matrix -> projection tiles
                const float* src = layer ? hseq0 + (size_t)row0 * H : x + (size_t)row0 * ldx;
                const size_t src_ld = layer ? (size_t)H : (size_t)ldx, K = layer ? (size_t)H : (size_t)Ipad;
                float* dst = layer ? h0_left : x_left;
                if (hipMemcpy2DAsync(dst, left * K * sizeof(float), src, (size_t)N * src_ld * sizeof(float),
                                     left * src_ld * sizeof(float), T, hipMemcpyDeviceToDevice, as) != hipSuccess) {
                    fsn_set_error("lstm2 forward (training): cannot gather the left-over rows");
                    return FSN_ERR_LAUNCH;
                }
                FsnGemmA a{};
                a.kind = 0;
                a.p0 = dst;
                a.ld = (long)K;
                FsnGemmC c{};
                c.kind = 0;
                c.p0 = gx_left;
                c.bias = layer ? b1 : b0;
                FSN_TRY(fsn_launch_gemm(a, layer ? wih1_p : wih0_p, c, T * left_tiles, 4 * H / 16, (int)K / 16, as));
                float* hs = (layer ? hseq1 : hseq0) + (size_t)row0 * H;
                float* sv = layer ? sv1 : sv0;
                float* gates = sv + (size_t)row0 * 4 * H;
                float* cseq = sv + (size_t)T * N * 4 * H + (size_t)row0 * H;
                for (int t = 0; t < T; ++t)
                    FSN_TRY(fsn_launch_lstm_step_train(gx_left, layer ? whh1_p : whh0_p, t ? hs + (t - 1) * stepH : hs,
                                                       hs + t * stepH, t ? cseq + (t - 1) * stepH : cseq, cseq + t * stepH,
                                                       gates + t * stepG, (long)t * left_tiles, left_tiles, H, t == 0, as));
            }
            if (hipEventRecord(cx->ev_join, cx->aux) != hipSuccess || hipStreamWaitEvent(s, cx->ev_join, 0) != hipSuccess) {
                fsn_set_error("aux stream join failed");
                return FSN_ERR_LAUNCH;
            }
        }
        return FSN_OK;
    }
    if (!lstm2_train_plan(T, N, I, H).fwd_chain) {  // layer by layer
        FSN_TRY(fsn_lstm_layer_forward(x, ldx, w_ih0, w_hh0, b_ih0, b_hh0, T, N, I, H, hseq0, save0, save_bytes, workspace,
                                       workspace_bytes, stream));
        return fsn_lstm_layer_forward(hseq0, H, w_ih1, w_hh1, b_ih1, b_hh1, T, N, H, H, hseq1, save1, save_bytes, workspace,
                                      workspace_bytes, stream);
    }
    Carver cv(workspace);
    float* wih0_p = cv.take<float>((size_t)4 * H * Ipad);
    float* whh0_p = cv.take<float>((size_t)4 * H * H);
    float* wih1_p = whh0_p + (size_t)4 * H * H;
    float* whh1_p = wih1_p + (size_t)4 * H * H;
    cv.take<float>((size_t)2 * 4 * H * H);
    float* b0 = cv.take<float>((size_t)2 * 4 * H);
    float* b1 = b0 + 4 * H;
    float* gx0 = cv.take<float>((size_t)T * N * 4 * H);
    float* exchange = cv.take<float>(fsn_fb_chain_exchange_floats(T, N));
    unsigned* flags = cv.take<unsigned>(fsn_fb_chain_flag_words());
    FSN_TRY(fsn_launch_pack(w_ih0, wih0_p, 4 * H, I, 4 * H, Ipad, s));
    FSN_TRY(fsn_launch_pack(w_hh0, whh0_p, 4 * H, H, 4 * H, H, s));
    FSN_TRY(fsn_launch_pack(w_ih1, wih1_p, 4 * H, H, 4 * H, H, s));
    FSN_TRY(fsn_launch_pack(w_hh1, whh1_p, 4 * H, H, 4 * H, H, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih0, b_hh0, b0, 4 * H, 4 * H, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih1, b_hh1, b1, 4 * H, 4 * H, s));
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = x;
    a.ld = ldx;
    FsnGemmC c{};
    c.kind = 0;
    c.p0 = gx0;
    c.bias = b0;
    FSN_TRY(fsn_launch_gemm(a, wih0_p, c, T * (N / 16), 4 * H / 16, Ipad / 16, s));
    FSN_PERSIST_BEGIN(s);
    FSN_TRY(fsn_launch_fb_chain(gx0, whh0_p, wih1_p, whh1_p, b1, exchange, flags, hseq1, T, N, H, s, hseq0,
                                static_cast<float*>(save0), static_cast<float*>(save1)));
    return fsn_launch_poison_if(flags + fsn_fb_chain_status_word(), hseq1, (size_t)T * N * H, s);
}

// Two stacked LSTM layers of equal width in inference mode as one wavefront (layer 1 at step t next to layer 0
// at step t + 1: T + 1 dependent launches instead of 2 T).  For the latency-bound regime - few rows - where
// SequenceModel blocks of the sibling models live (Improved FullSubNet's band sections: B x {20, 25, 6, 4} rows).
// H = 384 twice, up to 32 input columns, whole 64-row clusters in the group kernel's ranges (96 - 159 and 224 - 256 row
// tiles: e.g. Fast FullSubNet's bottleneck at 24 - 39 utterances per rank): clusters, 0 = not this shape
static int lstm2_infer_group_clusters(int T, int N, int I, int H0, int H1, long ldx) {
    if (H0 != 384 || H1 != 384 || I > 32 || N % 64 != 0 || N / 16 < kWavefrontBelowTiles) return 0;
    if (ldx != 16 && ldx != 32) return 0;  // the kernel reads x rows of exactly one or two K chunks; anything else: generic path
    if ((size_t)T * N * H0 * sizeof(float) > 0x7fffffffull) return 0;  // the reach of a buffer resource's offsets
    const int tiles = N / 16, c = fsn_lstm2_group_clusters(tiles);
    return 4 * c == tiles ? c : 0;
}
static size_t lstm2_fwd_workspace(int T, int N, int I, int H0, int H1, int group_clusters);
// sized for either row stride of x (the group kernel's buffers are included whenever the shape COULD take it)
extern "C" size_t fsn_lstm2_fwd_workspace_bytes(int T, int N, int I, int H0, int H1) {
    return lstm2_fwd_workspace(T, N, I, H0, H1, lstm2_infer_group_clusters(T, N, I, H0, H1, fsn_round_up(I, 16)));
}
static size_t lstm2_fwd_workspace(int T, int N, int I, int H0, int H1, int group_clusters) {
    Carver cv(nullptr);
    cv.take<float>((size_t)4 * H0 * fsn_round_up(I, 16));  // W_ih0 fragments
    cv.take<float>((size_t)4 * H0 * H0);                   // W_hh0
    cv.take<float>((size_t)4 * H1 * H0);                   // W_ih1
    cv.take<float>((size_t)4 * H1 * H1);                   // W_hh1
    cv.take<float>((size_t)4 * H0);                        // b0
    cv.take<float>((size_t)4 * H1);                        // b1
    cv.take<float>((size_t)4 * H1 * 16);                   // b1 as fragment tiles
    cv.take<float>((size_t)T * N * 4 * H0);                // layer-0 projection
    cv.take<float>((size_t)T * N * H0);                    // layer-0 hidden sequence
    cv.take<float>((size_t)N * (H0 + H1));                 // cell states
    if (H0 == H1 && lstm2_on_chain(T, N, H0)) {            // the persistent chain kernel instead of the wavefront
        cv.take<float>(fsn_fb_chain_exchange_floats(T, N));
        cv.take<unsigned>(fsn_fb_chain_flag_words());
    }
    if (const int clusters = group_clusters) {  // the group kernel (general two-layer form)
        cv.take<float>((size_t)4 * H0 * 32 + (size_t)3 * 4 * H0 * H0);
        cv.take<unsigned>(fsn_lstm2_group_flag_words(clusters));
    }
    return fsn_round_up_sz(cv.off, 256);
}
// 1 when fsn_lstm2_forward has a persistent kernel for this shape (callers that would otherwise run layer by layer on
// the per-layer persistent kernels - 1536+ rows - should then prefer it)
extern "C" int fsn_lstm2_forward_is_persistent(int T, int N, int I, long ldx, int H0, int H1) {
    return (H0 == H1 && lstm2_on_chain(T, N, H0)) || lstm2_infer_group_clusters(T, N, I, H0, H1, ldx) > 0;
}
extern "C" int fsn_lstm2_forward(const float* x, long ldx, const float* w_ih0, const float* w_hh0, const float* b_ih0,
                                 const float* b_hh0, const float* w_ih1, const float* w_hh1, const float* b_ih1,
                                 const float* b_hh1, int T, int N, int I, int H0, int H1, float* hseq1, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H0, ldx));
    FSN_REQUIRE(H1 >= 64 && H1 % 64 == 0, "lstm2: second hidden size %d must be a multiple of 64", H1);
    FSN_REQUIRE(x && w_ih0 && w_hh0 && b_ih0 && b_hh0 && w_ih1 && w_hh1 && b_ih1 && b_hh1 && hseq1 && workspace,
                "NULL pointer argument");
    if (workspace_bytes < fsn_lstm2_fwd_workspace_bytes(T, N, I, H0, H1)) {
        fsn_set_error("lstm2 forward: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ipad = fsn_round_up(I, 16), G0 = 4 * H0, G1 = 4 * H1;
    Carver cv(workspace);
    float* wih0_p = cv.take<float>((size_t)G0 * Ipad);
    float* whh0_p = cv.take<float>((size_t)G0 * H0);
    float* wih1_p = cv.take<float>((size_t)G1 * H0);
    float* whh1_p = cv.take<float>((size_t)G1 * H1);
    float* b0 = cv.take<float>((size_t)G0);
    float* b1 = cv.take<float>((size_t)G1);
    float* b1_frag = cv.take<float>((size_t)G1 * 16);
    float* gx = cv.take<float>((size_t)T * N * G0);
    float* hseq0 = cv.take<float>((size_t)T * N * H0);
    float* cst = cv.take<float>((size_t)N * (H0 + H1));
    if (const int clusters = lstm2_infer_group_clusters(T, N, I, H0, H1, ldx)) {
        // both layers, all steps, as one persistent launch of the group kernel (no projection GEMM, no gx): the four
        // packed matrices in one buffer, W_ih0 32 columns wide
        if (H0 == H1 && lstm2_on_chain(T, N, H0)) {
            cv.take<float>(fsn_fb_chain_exchange_floats(T, N));
            cv.take<unsigned>(fsn_fb_chain_flag_words());
        }
        float* gw = cv.take<float>((size_t)G0 * 32 + (size_t)3 * G0 * H0);
        unsigned* flags = cv.take<unsigned>(fsn_lstm2_group_flag_words(clusters));
        float* g_wih0 = gw;
        float* g_whh0 = g_wih0 + (size_t)G0 * 32;
        float* g_wih1 = g_whh0 + (size_t)G0 * H0;
        float* g_whh1 = g_wih1 + (size_t)G0 * H0;
        FSN_TRY(fsn_launch_pack(w_ih0, g_wih0, G0, I, G0, 32, s));
        FSN_TRY(fsn_launch_pack(w_hh0, g_whh0, G0, H0, G0, H0, s));
        FSN_TRY(fsn_launch_pack(w_ih1, g_wih1, G1, H0, G1, H0, s));
        FSN_TRY(fsn_launch_pack(w_hh1, g_whh1, G1, H1, G1, H1, s));
        FSN_TRY(fsn_launch_bias_sum(b_ih0, b_hh0, b0, G0, G0, s));
        FSN_TRY(fsn_launch_bias_sum(b_ih1, b_hh1, b1, G1, G1, s));
        FSN_PERSIST_BEGIN(s);
        FSN_TRY(fsn_launch_lstm2_group_train(x, ldx, (int)ldx, N, g_wih0, g_whh0, g_wih1, g_whh1, b0, b1, hseq0, hseq1, nullptr,
                                             nullptr, flags, T, clusters, H0, s));
        return fsn_launch_poison_if(flags + fsn_lstm2_group_status_word(clusters), hseq1, (size_t)T * N * H1, s);
    }
    FSN_TRY(fsn_launch_pack(w_ih0, wih0_p, G0, I, G0, Ipad, s));
    FSN_TRY(fsn_launch_pack(w_hh0, whh0_p, G0, H0, G0, H0, s));
    FSN_TRY(fsn_launch_pack(w_ih1, wih1_p, G1, H0, G1, H0, s));
    FSN_TRY(fsn_launch_pack(w_hh1, whh1_p, G1, H1, G1, H1, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih0, b_hh0, b0, G0, G0, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih1, b_hh1, b1, G1, G1, s));
    FSN_TRY(fsn_launch_bias_frag(b1, b1_frag, G1, s));
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = x;
    a.ld = ldx;
    FsnGemmC c{};
    c.kind = 0;
    c.p0 = gx;
    c.bias = b0;
    FSN_TRY(fsn_launch_gemm(a, wih0_p, c, T * (N / 16), G0 / 16, Ipad / 16, s));
    if (H0 == H1 && lstm2_on_chain(T, N, H0)) {  // H = 384 / 512, up to 64 rows: one persistent launch (fb_chain_kernels.hip)
        float* exchange = cv.take<float>(fsn_fb_chain_exchange_floats(T, N));
        unsigned* flags = cv.take<unsigned>(fsn_fb_chain_flag_words());
        FSN_PERSIST_BEGIN(s);
        FSN_TRY(fsn_launch_fb_chain(gx, whh0_p, wih1_p, whh1_p, b1, exchange, flags, hseq1, T, N, H0, s));
        return fsn_launch_poison_if(flags + fsn_fb_chain_status_word(), hseq1, (size_t)T * N * H0, s);
    }
    return fsn_launch_lstm_wavefront2w(gx, N / 16, 0, whh0_p, wih1_p, b1_frag, whh1_p, hseq0, hseq1, N, 0, cst,
                                       cst + (size_t)N * H0, T, N / 16, H0, H1, s);
}

// ---- two stacked GRU layers of equal width, few rows: ONE persistent launch of the chain kernel ---------------------------
// (audio_zen/model/module/sequence_model.py:59-66 with num_layers = 2: the full-band model of a GRU FullSubNet, B <= 64 rows.)
// nn.GRU's weights are expanded to the four-gate cell r | z | nx | nh (zero blocks where a gate has no input / no recurrent
// part) and take the LSTM chain's path unchanged: projection GEMM of layer 0, fb_chain_kernel<.., CELL = 1>.
extern "C" int fsn_gru2_forward_supported(int T, int N, int H) { return T >= 1 && lstm2_on_chain(T, N, H) ? 1 : 0; }
extern "C" size_t fsn_gru2_fwd_workspace_bytes(int T, int N, int I, int H) {
    if (T < 1 || N < 16 || N % 16 || I < 1 || H < 64) return 0;
    const size_t Ipad = fsn_round_up(I, 16), G = 4 * (size_t)H;
    Carver cv(nullptr);
    cv.take<float>(2 * (G * Ipad + 3 * G * H));  // the expanded matrices and their fragment-order copies
    cv.take<float>(2 * G);                       // b4 of both layers
    cv.take<float>((size_t)T * N * G);           // layer-0 projection
    cv.take<float>(fsn_fb_chain_exchange_floats(T, N));
    cv.take<unsigned>(fsn_fb_chain_flag_words());
    return fsn_round_up_sz(cv.off, 256);
}
extern "C" int fsn_gru2_forward(const float* x, long ldx, const float* w_ih0, const float* w_hh0, const float* b_ih0,
                                const float* b_hh0, const float* w_ih1, const float* w_hh1, const float* b_ih1,
                                const float* b_hh1, int T, int N, int I, int H, float* hseq1, void* workspace,
                                size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    FSN_REQUIRE(x && w_ih0 && w_hh0 && b_ih0 && b_hh0 && w_ih1 && w_hh1 && b_ih1 && b_hh1 && hseq1 && workspace, "NULL pointer argument");
    FSN_REQUIRE(fsn_gru2_forward_supported(T, N, H), "gru2 forward: built for H = 384 / 512 twice, up to 64 rows and 4095 steps on a "
                                                     "device that holds the chain's grid (fsn_gru2_forward_supported)");
    if (workspace_bytes < fsn_gru2_fwd_workspace_bytes(T, N, I, H)) {
        fsn_set_error("gru2 forward: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ipad = fsn_round_up(I, 16), G = 4 * H;
    Carver cv(workspace);
    float* e = cv.take<float>(2 * ((size_t)G * Ipad + (size_t)3 * G * H));
    float *wih0_4 = e, *whh0_4 = wih0_4 + (size_t)G * I, *wih1_4 = whh0_4 + (size_t)G * H, *whh1_4 = wih1_4 + (size_t)G * H;
    float* pk = e + (size_t)G * Ipad + (size_t)3 * G * H;
    float *wih0_p = pk, *whh0_p = wih0_p + (size_t)G * Ipad, *wih1_p = whh0_p + (size_t)G * H, *whh1_p = wih1_p + (size_t)G * H;
    float* b0 = cv.take<float>((size_t)2 * G);
    float* b1 = b0 + G;
    float* gx = cv.take<float>((size_t)T * N * G);
    float* exchange = cv.take<float>(fsn_fb_chain_exchange_floats(T, N));
    unsigned* flags = cv.take<unsigned>(fsn_fb_chain_flag_words());
    FSN_TRY(fsn_launch_gru_expand4(w_ih0, w_hh0, b_ih0, b_hh0, wih0_4, whh0_4, b0, I, H, s));
    FSN_TRY(fsn_launch_gru_expand4(w_ih1, w_hh1, b_ih1, b_hh1, wih1_4, whh1_4, b1, H, H, s));
    FSN_TRY(fsn_launch_pack(wih0_4, wih0_p, G, I, G, Ipad, s));
    FSN_TRY(fsn_launch_pack(whh0_4, whh0_p, G, H, G, H, s));
    FSN_TRY(fsn_launch_pack(wih1_4, wih1_p, G, H, G, H, s));
    FSN_TRY(fsn_launch_pack(whh1_4, whh1_p, G, H, G, H, s));
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = x;
    a.ld = ldx;
    FsnGemmC c{};
    c.kind = 0;
    c.p0 = gx;
    c.bias = b0;
    FSN_TRY(fsn_launch_gemm(a, wih0_p, c, T * (N / 16), G / 16, Ipad / 16, s));
    FSN_PERSIST_BEGIN(s);
    FSN_TRY(fsn_launch_fb_chain(gx, whh0_p, wih1_p, whh1_p, b1, exchange, flags, hseq1, T, N, H, s, nullptr, nullptr, nullptr, 1));
    return fsn_launch_poison_if(flags + fsn_fb_chain_status_word(), hseq1, (size_t)T * N * H, s);
}

// Streaming form (frame-by-frame / chunked inference with carried state): T more steps from the state
// (h, c) [N][H], which is updated in place.  Always on the per-step kernels.  The weights are re-tiled
// once (fsn_lstm_layer_pack) - a per-frame caller must not pay three pack launches per layer per call.
struct LayerPacked {
    size_t wih, whh, bias, total;  // float offsets
};
static LayerPacked layer_packed_layout(int I, int H) {
    LayerPacked p;
    const size_t Ipad = fsn_round_up(I, 16);
    p.wih = 0;
    p.whh = fsn_round_up_sz(4 * (size_t)H * Ipad, 64);
    p.bias = p.whh + fsn_round_up_sz(4 * (size_t)H * H, 64);
    p.total = p.bias + fsn_round_up_sz(4 * (size_t)H, 64);
    return p;
}
// ---- Improved FullSubNet: the normalised input of one band section, in the LSTM entries' layout ------------------------
extern "C" size_t fsn_improved_section_input_workspace_bytes(int B, int F) {
    if (B < 1 || F < 2) return 0;
    return fsn_round_up_sz(fsn_section_input_workspace_floats(B, F) * sizeof(float), 256);
}
extern "C" int fsn_improved_section_input(const float* noisy, const float* fb_out, int B, int F, int T, int lower, int upper,
                                          int sb_center, int sb_neighbor, int fb_center, int fb_neighbor, int unit_lo,
                                          int unit_hi, float eps, float* out, int Np, int ldo, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(noisy && fb_out && out && workspace, "NULL pointer argument");
    FSN_REQUIRE(B >= 1 && F >= 2 && T >= 1 && 0 <= lower && lower < upper && upper <= F, "section input: bad band [%d, %d) of %d bins",
                lower, upper, F);
    FSN_REQUIRE(sb_center >= 1 && fb_center >= 1 && sb_neighbor >= 0 && fb_neighbor >= 0 && (upper - lower) % sb_center == 0 &&
                    (upper - lower) % fb_center == 0 && (upper - lower) / sb_center == (upper - lower) / fb_center,
                "section input: the band must hold the same whole number of units for both windows");
    const int units = (upper - lower) / sb_center, W = sb_center + 2 * sb_neighbor + fb_center + 2 * fb_neighbor;
    // the reflections of model.py:376-383 are single ones: a window may not reach beyond a mirror image of the spectrum
    FSN_REQUIRE(sb_neighbor < F && fb_neighbor < F && sb_center + sb_neighbor <= F && fb_center + fb_neighbor <= F,
                "section input: windows wider than the spectrum");
    FSN_REQUIRE(0 <= unit_lo && unit_lo < unit_hi && unit_hi <= units, "section input: unit range [%d, %d) of %d", unit_lo, unit_hi,
                units);
    FSN_REQUIRE(Np >= B * (unit_hi - unit_lo) && Np <= 65535 && ldo >= W && ldo <= 240,
                "section input: out [T][%d][%d]: rows up to 65535, the window's %d columns up to 240 (a 64-frame tile in LDS)", Np,
                ldo, W);
    FSN_REQUIRE(eps > 0.f, "section input: eps must be positive");
    if (workspace_bytes < fsn_improved_section_input_workspace_bytes(B, F)) {
        fsn_set_error("section input: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    return fsn_launch_section_input(noisy, fb_out, B, F, T, lower, units, sb_center, sb_neighbor, fb_center, fb_neighbor, unit_lo,
                                    unit_hi, eps, out, Np, ldo, workspace, static_cast<hipStream_t>(stream));
}

// ---- several independent two-layer stacks over the same frames ------------------------------------------------------
// (improved_fullsubnet/model.py:402-449: the band sections' SequenceModels - B x {20, 25, 6, 4} rows at 48 kHz, input
// widths 62 .. 180 - all see the same T frames.)  When every stack is H = 384 twice and together they fill most of the
// chip's workgroup sets, all of them run as ONE persistent launch of the group kernel (GX form: projection GEMM per
// stack, then lstm2_group_multi_kernel); otherwise stack by stack through fsn_lstm2_forward's forms.
static int lstm2_multi_clusters(int n, const fsn_lstm2_stack* st, int T) {
    const int cap = fsn_lstm2_group_multi_cap();
    if (cap == 0 || n < 1 || n > 8 || T < 4) return 0;
    int clusters = 0;
    for (int k = 0; k < n; ++k) {
        if (st[k].H0 != 384 || st[k].H1 != 384 || st[k].N % 16 || (size_t)T * st[k].N * 384 * 4 > 0x7fffffffull) return 0;
        clusters += (st[k].N + 63) / 64;
    }
    // below ~3/4 of the sets the stacks are faster as wavefronts on their own streams (a persistent step costs the same
    // ~58 us whatever the cluster count)
    return clusters <= cap && 4 * clusters >= 3 * cap ? clusters : 0;
}
struct Lstm2MultiPlan {
    float *whh0, *wih1, *whh1, *wih0, *b0, *b1, *gx, *hseq0;
};
static void lstm2_multi_carve(int n, const fsn_lstm2_stack* st, int T, int clusters, Carver& cv, Lstm2MultiPlan* out,
                              unsigned** flags) {
    // the recurrent matrices of all stacks first (one buffer: 32-bit offsets inside the kernel)
    for (int k = 0; k < n; ++k) {
        const size_t G = 4 * (size_t)st[k].H0, H = st[k].H0;
        Lstm2MultiPlan p{};
        p.whh0 = cv.take<float>(G * H);
        p.wih1 = cv.take<float>(G * H);
        p.whh1 = cv.take<float>(G * H);
        if (out) out[k] = p;
    }
    for (int k = 0; k < n; ++k) {
        const size_t G = 4 * (size_t)st[k].H0, H = st[k].H0, Ipad = fsn_round_up(st[k].I, 16);
        Lstm2MultiPlan p = out ? out[k] : Lstm2MultiPlan{};
        p.wih0 = cv.take<float>(G * Ipad);
        p.b0 = cv.take<float>(G);
        p.b1 = cv.take<float>(G);
        p.gx = cv.take<float>((size_t)T * st[k].N * G);
        p.hseq0 = cv.take<float>((size_t)T * st[k].N * H);
        if (out) out[k] = p;
    }
    unsigned* f = cv.take<unsigned>(fsn_lstm2_group_flag_words(clusters));
    if (flags) *flags = f;
}
static int check_lstm2_stacks(int n, const fsn_lstm2_stack* st, int T) {
    FSN_REQUIRE(st && n >= 1 && n <= 8, "lstm2 multi: 1 .. 8 stacks (got %d)", n);
    for (int k = 0; k < n; ++k) {
        FSN_TRY(check_lstm_layer(T, st[k].N, st[k].I, st[k].H0, st[k].ldx));
        FSN_REQUIRE(st[k].H1 >= 64 && st[k].H1 % 64 == 0, "lstm2 multi: stack %d: second hidden size %d must be a multiple of 64", k,
                    st[k].H1);
    }
    return FSN_OK;
}
extern "C" int fsn_lstm2_multi_is_persistent(int n, const fsn_lstm2_stack* stacks, int T) {
    if (!stacks || n < 1 || n > 8) return 0;
    for (int k = 0; k < n; ++k)
        if (stacks[k].N < 16 || stacks[k].I < 1) return 0;
    return lstm2_multi_clusters(n, stacks, T) > 0 ? 1 : 0;
}
extern "C" size_t fsn_lstm2_multi_workspace_bytes(int n, const fsn_lstm2_stack* stacks, int T) {
    if (check_lstm2_stacks(n, stacks, T) != FSN_OK) return 0;
    if (const int clusters = lstm2_multi_clusters(n, stacks, T)) {
        Carver cv(nullptr);
        lstm2_multi_carve(n, stacks, T, clusters, cv, nullptr, nullptr);
        return fsn_round_up_sz(cv.off, 256);
    }
    size_t most = 0;  // stack by stack: one stack's workspace at a time
    for (int k = 0; k < n; ++k) {
        const size_t b = fsn_lstm2_fwd_workspace_bytes(T, stacks[k].N, stacks[k].I, stacks[k].H0, stacks[k].H1);
        most = b > most ? b : most;
    }
    return most;
}
extern "C" int fsn_lstm2_forward_multi(int n, const fsn_lstm2_stack* stacks, int T, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm2_stacks(n, stacks, T));
    FSN_REQUIRE(workspace, "NULL pointer argument");
    for (int k = 0; k < n; ++k) {
        const fsn_lstm2_stack& q = stacks[k];
        FSN_REQUIRE(q.x && q.w_ih0 && q.w_hh0 && q.b_ih0 && q.b_hh0 && q.w_ih1 && q.w_hh1 && q.b_ih1 && q.b_hh1 && q.hseq1,
                    "lstm2 multi: stack %d: NULL pointer argument", k);
    }
    if (workspace_bytes < fsn_lstm2_multi_workspace_bytes(n, stacks, T)) {
        fsn_set_error("lstm2 multi: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    const int clusters = lstm2_multi_clusters(n, stacks, T);
    if (!clusters) {
        for (int k = 0; k < n; ++k) {
            const fsn_lstm2_stack& q = stacks[k];
            FSN_TRY(fsn_lstm2_forward(q.x, q.ldx, q.w_ih0, q.w_hh0, q.b_ih0, q.b_hh0, q.w_ih1, q.w_hh1, q.b_ih1, q.b_hh1, T, q.N,
                                      q.I, q.H0, q.H1, q.hseq1, workspace, workspace_bytes, stream));
        }
        return FSN_OK;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Carver cv(workspace);
    Lstm2MultiPlan plan[8];
    unsigned* flags = nullptr;
    lstm2_multi_carve(n, stacks, T, clusters, cv, plan, &flags);
    FsnGroupStack gs[8];
    for (int k = 0; k < n; ++k) {
        const fsn_lstm2_stack& q = stacks[k];
        const Lstm2MultiPlan& p = plan[k];
        const int H = q.H0, G = 4 * H, Ipad = fsn_round_up(q.I, 16);
        FSN_TRY(fsn_launch_pack(q.w_ih0, p.wih0, G, q.I, G, Ipad, s));
        FSN_TRY(fsn_launch_pack(q.w_hh0, p.whh0, G, H, G, H, s));
        FSN_TRY(fsn_launch_pack(q.w_ih1, p.wih1, G, H, G, H, s));
        FSN_TRY(fsn_launch_pack(q.w_hh1, p.whh1, G, H, G, H, s));
        FSN_TRY(fsn_launch_bias_sum(q.b_ih0, q.b_hh0, p.b0, G, G, s));
        FSN_TRY(fsn_launch_bias_sum(q.b_ih1, q.b_hh1, p.b1, G, G, s));
        FsnGemmA a{};
        a.kind = 0;
        a.p0 = q.x;
        a.ld = q.ldx;
        FsnGemmC c{};
        c.kind = 0;
        c.p0 = p.gx;
        c.bias = p.b0;
        FSN_TRY(fsn_launch_gemm(a, p.wih0, c, T * (q.N / 16), G / 16, Ipad / 16, s));
        FsnGroupStack& g = gs[k];
        g.gx = p.gx;
        g.whh0_p = p.whh0;
        g.wih1_p = p.wih1;
        g.whh1_p = p.whh1;
        g.bias1 = p.b1;
        g.hseq0 = p.hseq0;
        g.hseq1 = q.hseq1;
        g.N = q.N;
    }
    FSN_PERSIST_BEGIN(s);
    FSN_TRY(fsn_launch_lstm2_group_multi(n, gs, flags, T, 384, s));
    for (int k = 0; k < n; ++k)
        FSN_TRY(fsn_launch_poison_if(flags + fsn_lstm2_group_status_word(clusters), stacks[k].hseq1,
                                     (size_t)T * stacks[k].N * stacks[k].H1, s));
    return FSN_OK;
}

extern "C" size_t fsn_lstm_layer_packed_bytes(int I, int H) {
    if (I < 1 || H < 64 || H % 64) return 0;
    return layer_packed_layout(I, H).total * sizeof(float);
}
extern "C" int fsn_lstm_layer_pack(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int I,
                                   int H, void* packed, size_t packed_bytes, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(w_ih && w_hh && b_ih && b_hh && packed, "NULL pointer argument");
    FSN_REQUIRE(I >= 1 && H >= 64 && H % 64 == 0, "lstm layer: need I >= 1 and H a multiple of 64 (got %d, %d)", I, H);
    if (packed_bytes < fsn_lstm_layer_packed_bytes(I, H)) {
        fsn_set_error("lstm layer pack: buffer too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const LayerPacked p = layer_packed_layout(I, H);
    float* o = static_cast<float*>(packed);
    FSN_TRY(fsn_launch_pack(w_ih, o + p.wih, 4 * H, I, 4 * H, fsn_round_up(I, 16), s));
    FSN_TRY(fsn_launch_pack(w_hh, o + p.whh, 4 * H, H, 4 * H, H, s));
    return fsn_launch_bias_sum(b_ih, b_hh, o + p.bias, 4 * H, 4 * H, s);
}
extern "C" size_t fsn_lstm_layer_state_workspace_bytes(int T, int N, int H) {
    return fsn_round_up_sz((size_t)T * N * 4 * H * sizeof(float), 256);  // the input projection of the T steps
}
extern "C" int fsn_lstm_layer_forward_state(const float* x, long ldx, const void* packed, int T, int N, int I, int H,
                                            float* hseq, float* h_state, float* c_state, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    FSN_REQUIRE(x && packed && hseq && h_state && c_state && workspace, "NULL pointer argument");
    if (workspace_bytes < fsn_lstm_layer_state_workspace_bytes(T, N, H)) {
        fsn_set_error("lstm layer forward: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const LayerPacked p = layer_packed_layout(I, H);
    const float* pk = static_cast<const float*>(packed);
    float* gx = static_cast<float*>(workspace);
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = x;
    a.ld = ldx;
    FsnGemmC c{};
    c.kind = 0;
    c.p0 = gx;
    c.bias = pk + p.bias;
    FSN_TRY(fsn_launch_gemm(a, pk + p.wih, c, T * (N / 16), 4 * H / 16, fsn_round_up(I, 16) / 16, s));
    const size_t step = (size_t)N * H;
    for (int t = 0; t < T; ++t)
        FSN_TRY(fsn_launch_lstm_step(gx, pk + p.whh, t ? hseq + (t - 1) * step : h_state, hseq + t * step, c_state,
                                     (long)t * (N / 16), N / 16, H, 0, s));
    if (hipMemcpyAsync(h_state, hseq + (size_t)(T - 1) * step, step * sizeof(float), hipMemcpyDeviceToDevice, s) !=
        hipSuccess) {
        fsn_set_error("state copy failed");
        return FSN_ERR_LAUNCH;
    }
    return FSN_OK;
}

extern "C" size_t fsn_lstm_layer_bwd_workspace_bytes(int T, int N, int I, int H) {
    const int Ipad = fsn_round_up(I, 16);
    Carver cv(nullptr);
    cv.take<float>((size_t)H * 4 * H);          // W_hh^T fragments
    cv.take<float>((size_t)Ipad * 4 * H);       // W_ih^T fragments
    cv.take<float>((size_t)T * N * 4 * H);      // dgates
    cv.take<float>((size_t)N * H);              // dh_rec
    cv.take<float>((size_t)N * H);              // dc
    size_t tn = fsn_gemm_tn_workspace_bytes(4 * H, I, (long)T * N);
    const size_t tn2 = fsn_gemm_tn_workspace_bytes(4 * H, H, (long)T * N);
    tn = tn > tn2 ? tn : tn2;
    const size_t cs = fsn_colsum_workspace_bytes(4 * H, (long)T * N);
    cv.take<char>(tn > cs ? tn : cs);
    return fsn_round_up_sz(cv.off, 256);
}

extern "C" int fsn_lstm_layer_backward(const float* dh, const float* x, long ldx, const float* w_ih,
                                       const float* w_hh, int T, int N, int I, int H, const float* hseq,
                                       const void* save, float* dx, long lddx, float* dw_ih, float* dw_hh, float* db,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    FSN_REQUIRE(dh && x && w_ih && w_hh && hseq && save && dw_ih && dw_hh && db && workspace, "NULL pointer argument");
    FSN_REQUIRE(!dx || lddx >= I, "dx row stride %ld < I", lddx);
    if (workspace_bytes < fsn_lstm_layer_bwd_workspace_bytes(T, N, I, H)) {
        fsn_set_error("lstm layer backward: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ipad = fsn_round_up(I, 16), G = 4 * H;
    Carver cv(workspace);
    float* whhT_p = cv.take<float>((size_t)H * G);
    float* wihT_p = cv.take<float>((size_t)Ipad * G);
    float* dgates = cv.take<float>((size_t)T * N * G);
    float* dh_rec = cv.take<float>((size_t)N * H);
    float* dc = cv.take<float>((size_t)N * H);
    size_t tn = fsn_gemm_tn_workspace_bytes(G, I, (long)T * N);
    const size_t tn2 = fsn_gemm_tn_workspace_bytes(G, H, (long)T * N);
    tn = tn > tn2 ? tn : tn2;
    const size_t cs = fsn_colsum_workspace_bytes(G, (long)T * N);
    void* scratch = cv.take<char>(tn > cs ? tn : cs);
    const float* gates = static_cast<const float*>(save);
    const float* cseq = gates + (size_t)T * N * G;
    // "weights" of dh_rec = dgates W_hh are W_hh^T: out = H columns, k = 4H; nn.LSTM stores exactly
    // that transposed ([4H][H] = [k][out]).  Likewise W_ih^T for dX.
    FSN_TRY(fsn_launch_pack(w_hh, whhT_p, H, G, H, G, s, 1, H));
    FSN_TRY(fsn_launch_pack(w_ih, wihT_p, I, G, Ipad, G, s, 1, I));
    const size_t step = (size_t)N * H;
    FsnGemmA a{};
    FsnGemmC c{};
    // one fused launch per step: dh_rec = dgates_{t+1} W_hh, then the cell derivative -> dgates_t
    for (int t = T - 1; t >= 0; --t)
        FSN_TRY(fsn_launch_bptt_step(dh + t * step, t + 1 < T ? dgates + (size_t)(t + 1) * N * G : dgates, whhT_p, dc,
                                     gates + (size_t)t * N * G, cseq + t * step, t ? cseq + (t - 1) * step : cseq,
                                     dgates + (size_t)t * N * G, N / 16, H, t == T - 1, t == 0, s));
    (void)dh_rec;
    if (dx) {
        a = FsnGemmA{};
        a.kind = 0;
        a.p0 = dgates;
        a.ld = G;
        c = FsnGemmC{};
        c.kind = 3;
        c.p0 = dx;
        c.ld = lddx;
        c.rows = T * N;
        c.cols = I;
        FSN_TRY(fsn_launch_gemm(a, wihT_p, c, T * (N / 16), Ipad / 16, G / 16, s));
    }
    // dW_ih = dgates^T X and, from the same pass over dgates, db = its column sums
    FSN_TRY(fsn_launch_gemm_tn(dgates, G, x, ldx, dw_ih, I, G, I, (long)T * N, scratch, s, db));
    if (T > 1) {
        FSN_TRY(fsn_launch_gemm_tn(dgates + (size_t)N * G, G, hseq, H, dw_hh, H, G, H, (long)(T - 1) * N, scratch, s));
    } else if (hipMemsetAsync(dw_hh, 0, (size_t)G * H * sizeof(float), s) != hipSuccess) {
        fsn_set_error("memset failed");
        return FSN_ERR_LAUNCH;
    }
    return FSN_OK;
}

// ---- training: backward of two stacked layers (the counterpart of fsn_lstm2_forward_train) ------------------------
// Two fsn_lstm_layer_backward calls in one; the sub-band shape runs its BPTT - both layers, all steps, the
// layer-to-layer dX included - as ONE persistent launch (lstm_group_bptt_kernels.hip).
static int lstm2_bptt_group_clusters(int T, int N, int I, int H) { return lstm2_train_plan(T, N, I, H).bptt_group; }
extern "C" size_t fsn_lstm2_bwd_workspace_bytes(int T, int N, int I, int H, int arith) {
    arith &= ~FSN_ARITH_SAVES16;
    const int Ipad = fsn_round_up(I, 16), G = 4 * H;
    const size_t l1 = fsn_lstm_layer_bwd_workspace_bytes(T, N, H, H), l0 = fsn_lstm_layer_bwd_workspace_bytes(T, N, I, H);
    Carver cv(nullptr);
    if (const int clusters = lstm2_bptt_group_clusters(T, N, I, H)) {
        const size_t left = (size_t)(N / 16 - 4 * clusters) * 16;
        cv.take<float>((size_t)3 * H * G + (size_t)Ipad * G);  // W_hh1^T, W_ih1^T, W_hh0^T, W_ih0^T fragments
        cv.take<float>((size_t)2 * T * N * G);                 // dgates of both layers
        cv.take<float>((size_t)T * N * H);                     // layer 0's dH (dgates1 W_ih1), produced by the kernel
        cv.take<unsigned>(lstm2_group_flag_words_any(clusters));
        cv.take<float>((size_t)T * left * G);                  // left-over rows: compact dgates1
        cv.take<float>((size_t)T * left * H);                  // ... their dh0
        cv.take<float>((size_t)left * H);                      // ... dc
        size_t tn = fsn_gemm_tn_workspace_bytes(G, I, (long)T * N);
        const size_t tn2 = fsn_gemm_tn_workspace_bytes(G, H, (long)T * N);
        cv.take<char>(tn > tn2 ? tn : tn2);
        if (arith != FSN_ARITH_F32) cv.take<unsigned short>((size_t)3 * H * G);  // 16-bit W^T fragments
        if (arith != FSN_ARITH_F32) {  // lstm_group16_kernels.hip: its packed weights and the rings of exchanged gate-gradient tiles
            cv.take<char>(fsn_lstm2_g16_bwd_weight_bytes());
            cv.take<float>(fsn_lstm2_g16_partial_floats(clusters));
            cv.take<unsigned short>((size_t)2 * T * N * G);  // 16-bit gate gradients: operands of the weight-gradient products
            cv.take<unsigned short>((size_t)2 * T * N * H);  // 16-bit hidden sequences
            cv.take<float>((size_t)2 * clusters * G);        // bias-gradient sums per (layer, cluster)
            cv.take<unsigned short>((size_t)T * N * 32 + (size_t)G * 32);  // x in 16 bits | W_ih0 fragments (gemm_tn16n / gemm_dx16)
        }
        return fsn_round_up_sz(cv.off, 256);
    }
    if (lstm2_train_plan(T, N, I, H).bptt_chain) {
        cv.take<float>((size_t)3 * H * G + (size_t)Ipad * G);  // W_hh1^T, W_ih1^T, W_hh0^T, W_ih0^T fragments
        cv.take<float>((size_t)2 * T * N * G);                 // dgates of both layers
        cv.take<float>(fsn_fb_chain_bptt_dx_floats(T, N));
        cv.take<unsigned>(fsn_fb_chain_bptt_flag_words());
        size_t tn = fsn_gemm_tn_workspace_bytes(G, I, (long)T * N);
        const size_t tn2 = fsn_gemm_tn_workspace_bytes(G, H, (long)T * N);
        cv.take<char>(tn > tn2 ? tn : tn2);
        return fsn_round_up_sz(cv.off, 256);
    }
    cv.take<float>((size_t)T * N * H);  // dh0
    cv.take<char>(l1 > l0 ? l1 : l0);
    return fsn_round_up_sz(cv.off, 256);
}
// `phase`: which parts run (a sum; 7 = everything) - 1: back-propagation through time, the gate gradients stay in the
// workspace; 4: dx from them; 2: the weight- and bias-gradient products from them; 8: only what the products need BESIDES
// the gate gradients (the 16-bit copies of the hidden sequences: independent of part 1, so a caller can have them made on
// another stream while part 1 runs); 16 (with 2): a part-8 call has done that.  Parts 2 and 4 take the same arguments
// and the same workspace, untouched since part 1; either may be issued on another stream, ordered behind part 1 by the
// caller.  The persistent shapes only (sub-band group kernels, full-band chain): the layer-by-layer form runs whole in
// part 1.
static int lstm2_backward_phases(const float* dh1, const float* x, long ldx, const float* w_ih0, const float* w_hh0,
                                 const float* w_ih1, const float* w_hh1, int T, int N, int I, int H, const float* hseq0,
                                 const float* hseq1, const void* save0, const void* save1, float* dx, long lddx,
                                 float* dw_ih0, float* dw_hh0, float* db0, float* dw_ih1, float* dw_hh1, float* db1,
                                 void* workspace, size_t workspace_bytes, int arith, void* stream, int phase) {
    CallScope scope(stream);
    const bool chain_part = (phase & 1) != 0, products_part = (phase & 2) != 0, dx_part = (phase & 4) != 0;
    const bool prepare_part = (phase & 8) != 0, prepared = (phase & 16) != 0;
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    const int saves16 = arith & FSN_ARITH_SAVES16;  // must be what the forward call of this step was given
    arith &= ~FSN_ARITH_SAVES16;
    FSN_REQUIRE((arith == FSN_ARITH_F32 && !saves16) || arith == FSN_ARITH_F16 || arith == FSN_ARITH_BF16,
                "lstm2 backward: arithmetic %d unknown (FSN_ARITH_F32 / _F16 / _BF16 [| FSN_ARITH_SAVES16])", arith | saves16);
    FSN_REQUIRE(dh1 && x && w_ih0 && w_hh0 && w_ih1 && w_hh1 && hseq0 && hseq1 && save0 && save1 && dw_ih0 && dw_hh0 && db0 &&
                    dw_ih1 && dw_hh1 && db1 && workspace,
                "NULL pointer argument");
    FSN_REQUIRE(!dx || lddx >= I, "dx row stride %ld < I", lddx);
    if (workspace_bytes < fsn_lstm2_bwd_workspace_bytes(T, N, I, H, arith)) {
        fsn_set_error("lstm2 backward: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    const int clusters = lstm2_bptt_group_clusters(T, N, I, H);
    if (!clusters && lstm2_train_plan(T, N, I, H).bptt_chain) {
        // the full-band shape (H = 512, up to 80 rows): both layers' BPTT as one persistent launch (fb_chain_bptt_kernels.hip),
        // then the weight-gradient GEMMs
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int Ipad = fsn_round_up(I, 16), G = 4 * H;
        Carver cv(workspace);
        float* whh1T_p = cv.take<float>((size_t)3 * H * G + (size_t)Ipad * G);
        float* wih1T_p = whh1T_p + (size_t)H * G;
        float* whh0T_p = wih1T_p + (size_t)H * G;
        float* wih0T_p = whh0T_p + (size_t)H * G;
        float* dg1 = cv.take<float>((size_t)2 * T * N * G);
        float* dg0 = dg1 + (size_t)T * N * G;
        float* dxp = cv.take<float>(fsn_fb_chain_bptt_dx_floats(T, N));
        unsigned* flags = cv.take<unsigned>(fsn_fb_chain_bptt_flag_words());
        size_t tn = fsn_gemm_tn_workspace_bytes(G, I, (long)T * N);
        const size_t tn2 = fsn_gemm_tn_workspace_bytes(G, H, (long)T * N);
        void* scratch = cv.take<char>(tn > tn2 ? tn : tn2);
        if (chain_part) {
            FSN_TRY(fsn_launch_pack(w_hh1, whh1T_p, H, G, H, G, s, 1, H));
            FSN_TRY(fsn_launch_pack(w_ih1, wih1T_p, H, G, H, G, s, 1, H));
            FSN_TRY(fsn_launch_pack(w_hh0, whh0T_p, H, G, H, G, s, 1, H));
            if (dx) FSN_TRY(fsn_launch_pack(w_ih0, wih0T_p, I, G, Ipad, G, s, 1, I));
            FSN_PERSIST_BEGIN(s);
            FSN_TRY(fsn_launch_fb_chain_bptt(dh1, whh1T_p, wih1T_p, whh0T_p, static_cast<const float*>(save0),
                                             static_cast<const float*>(save1), dg0, dg1, dxp, flags, T, N, H, s));
            // both gate-gradient buffers (dg1 | dg0 are adjacent): every weight gradient and dx derive from them
            FSN_TRY(fsn_launch_poison_if(flags + fsn_fb_chain_bptt_status_word(), dg1, (size_t)2 * T * N * G, s));
        }
        if (dx && dx_part) {
            FsnGemmA a{};
            a.kind = 0;
            a.p0 = dg0;
            a.ld = G;
            FsnGemmC c{};
            c.kind = 3;
            c.p0 = dx;
            c.ld = lddx;
            c.rows = T * N;
            c.cols = I;
            FSN_TRY(fsn_launch_gemm(a, wih0T_p, c, T * (N / 16), Ipad / 16, G / 16, s));
        }
        if (!products_part) return FSN_OK;
        FSN_TRY(fsn_launch_gemm_tn(dg1, G, hseq0, H, dw_ih1, H, G, H, (long)T * N, scratch, s, db1));
        FSN_TRY(fsn_launch_gemm_tn(dg0, G, x, ldx, dw_ih0, I, G, I, (long)T * N, scratch, s, db0));
        if (T > 1) {
            FSN_TRY(fsn_launch_gemm_tn(dg1 + (size_t)N * G, G, hseq1, H, dw_hh1, H, G, H, (long)(T - 1) * N, scratch, s));
            FSN_TRY(fsn_launch_gemm_tn(dg0 + (size_t)N * G, G, hseq0, H, dw_hh0, H, G, H, (long)(T - 1) * N, scratch, s));
        } else if (hipMemsetAsync(dw_hh1, 0, (size_t)G * H * sizeof(float), s) != hipSuccess ||
                   hipMemsetAsync(dw_hh0, 0, (size_t)G * H * sizeof(float), s) != hipSuccess) {
            fsn_set_error("memset failed");
            return FSN_ERR_LAUNCH;
        }
        return FSN_OK;
    }
    if (!clusters) {  // layer by layer; layer 1's dx is d loss / d hseq0
        if (!chain_part) return FSN_OK;  // (this form ran whole in phase 1)
        Carver cv(workspace);
        float* dh0 = cv.take<float>((size_t)T * N * H);
        const size_t l1 = fsn_lstm_layer_bwd_workspace_bytes(T, N, H, H), l0 = fsn_lstm_layer_bwd_workspace_bytes(T, N, I, H);
        void* ws = cv.take<char>(l1 > l0 ? l1 : l0);
        FSN_TRY(fsn_lstm_layer_backward(dh1, hseq0, H, w_ih1, w_hh1, T, N, H, H, hseq1, save1, dh0, H, dw_ih1, dw_hh1, db1, ws,
                                        l1 > l0 ? l1 : l0, stream));
        return fsn_lstm_layer_backward(dh0, x, ldx, w_ih0, w_hh0, T, N, I, H, hseq0, save0, dx, lddx, dw_ih0, dw_hh0, db0, ws,
                                       l1 > l0 ? l1 : l0, stream);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ipad = fsn_round_up(I, 16), G = 4 * H;
    const int left_tiles = N / 16 - 4 * clusters, left = left_tiles * 16, row0 = 64 * clusters;
    Carver cv(workspace);
    float* whh1T_p = cv.take<float>((size_t)3 * H * G + (size_t)Ipad * G);
    float* wih1T_p = whh1T_p + (size_t)H * G;
    float* whh0T_p = wih1T_p + (size_t)H * G;
    float* wih0T_p = whh0T_p + (size_t)H * G;
    float* dg1 = cv.take<float>((size_t)2 * T * N * G);
    float* dg0 = dg1 + (size_t)T * N * G;
    float* dxbuf = cv.take<float>((size_t)T * N * H);
    unsigned* flags = cv.take<unsigned>(lstm2_group_flag_words_any(clusters));
    float* dg1_left = cv.take<float>((size_t)T * left * G);
    float* dh0_left = cv.take<float>((size_t)T * left * H);
    float* dc_left = cv.take<float>((size_t)left * H);
    size_t tn = fsn_gemm_tn_workspace_bytes(G, I, (long)T * N);
    const size_t tn2 = fsn_gemm_tn_workspace_bytes(G, H, (long)T * N);
    void* scratch = cv.take<char>(tn > tn2 ? tn : tn2);
    unsigned short* w16 = arith != FSN_ARITH_F32 ? cv.take<unsigned short>((size_t)3 * H * G) : nullptr;
    void* g16_w = arith != FSN_ARITH_F32 ? cv.take<char>(fsn_lstm2_g16_bwd_weight_bytes()) : nullptr;
    float* partials = arith != FSN_ARITH_F32 ? cv.take<float>(fsn_lstm2_g16_partial_floats(clusters)) : nullptr;
    unsigned short* dg16 = arith != FSN_ARITH_F32 ? cv.take<unsigned short>((size_t)2 * T * N * G) : nullptr;  // layer 0 | layer 1
    unsigned short* h16 = arith != FSN_ARITH_F32 ? cv.take<unsigned short>((size_t)2 * T * N * H) : nullptr;    // hseq0 | hseq1
    float* dbp = arith != FSN_ARITH_F32 ? cv.take<float>((size_t)2 * clusters * G) : nullptr;
    const bool g16 = lstm2_use_g16(arith, clusters, N);
    // the weight-gradient products from 16-bit operands in memory (needs the shapes' one-workgroup-per-CU plan)
    const bool tn16h = g16 && T > 1 && fsn_gemm_tn16h_supported(G, H, (long)(T - 1) * N) && !g_tn16h_off.load(std::memory_order_relaxed);
    // ... and layer 0's input-side products too (dx, dW_ih0): then the BPTT launch stores no fp32 gate gradients at all
    const bool in16 = tn16h && fsn_gemm_tn16n_supported(G, I, (long)T * N) && fsn_gemm_dx16_supported((long)T * N, G, I) && ldx == 32 &&
                      !g_in16_off.load(std::memory_order_relaxed);
    unsigned short* x16 = arith != FSN_ARITH_F32 ? cv.take<unsigned short>((size_t)T * N * 32 + (size_t)G * 32) : nullptr;  // x in 16 bits | W_ih0 fragments
    unsigned short* wdx16 = in16 ? x16 + (size_t)T * N * 32 : nullptr;
    const float* sv0 = static_cast<const float*>(save0);
    const float* sv1 = static_cast<const float*>(save1);
    // FSN_ARITH_SAVES16: the forward launch left h_t in 16 bits inside the save buffers (second half of a row's gate slot) for
    // every cluster row - with no step-by-step rows beside the launch the hidden sequences need no conversion pass at all
    const bool h16_saved = tn16h && saves16 && left == 0;
    const unsigned short* h16_0 = h16_saved ? reinterpret_cast<const unsigned short*>(sv0) + G : h16;
    const unsigned short* h16_1 = h16_saved ? reinterpret_cast<const unsigned short*>(sv1) + G : (h16 ? h16 + (size_t)T * N * H : nullptr);
    const long ldh16 = h16_saved ? 2L * G : H;  // 16-bit elements between rows
    if (prepare_part) {
        if (tn16h && !h16_saved) {
            FSN_TRY(fsn_launch_to16(hseq0, h16, (size_t)T * N * H, arith, s));
            FSN_TRY(fsn_launch_to16(hseq1, h16 + (size_t)T * N * H, (size_t)T * N * H, arith, s));
        }
        if (!chain_part && !products_part && !dx_part) return FSN_OK;
    }
    if (chain_part) {
    // "weights" of dh = dgates W are W^T: out = H columns, k = 4H; nn.LSTM stores exactly that transposed.  (Not needed when the
    // 16-bit kernels take every row and every product: they pack the raw weights their own way.)
    if (!(g16 && left == 0 && in16)) {
        FSN_TRY(fsn_launch_pack(w_hh1, whh1T_p, H, G, H, G, s, 1, H));
        FSN_TRY(fsn_launch_pack(w_ih1, wih1T_p, H, G, H, G, s, 1, H));
        FSN_TRY(fsn_launch_pack(w_hh0, whh0T_p, H, G, H, G, s, 1, H));
        FSN_TRY(fsn_launch_pack(w_ih0, wih0T_p, I, G, Ipad, G, s, 1, I));
    }
    if (w16 && !g16) FSN_TRY(fsn_launch_to16(whh1T_p, w16, (size_t)3 * H * G, arith, s));  // the BPTT kernel's W^T fragments in 16 bits
    StreamCtx* cx = cur_ctx();
    if (left > 0) {
        FSN_TRY(aux_init(cx));
        if (hipEventRecord(cx->ev_fork, s) != hipSuccess || hipStreamWaitEvent(cx->aux, cx->ev_fork, 0) != hipSuccess) {
            fsn_set_error("aux stream fork failed");
            return FSN_ERR_LAUNCH;
        }
    }
    {
        FSN_PERSIST_BEGIN(s);
        if (g16) {  // the 16-bit arithmetic's own kernel (K-split; packs the raw weights its way into w16)
            // (layer 1's fp32 gate gradients of the cluster rows are not stored: the products below take the 16-bit copies)
            FSN_TRY(fsn_launch_lstm2_g16_bptt(dh1, w_hh1, w_ih1, w_hh0, sv0, sv1, dg0, dg1, partials, flags, g16_w, T, N, clusters,
                                              H, s, arith | saves16, dg16, dg16 + (size_t)T * N * G, dbp, tn16h ? 0 : 1,
                                              in16 ? 0 : 1));  // dg16 = layer 0 | layer 1
            FSN_TRY(fsn_launch_poison_if(flags + fsn_lstm2_g16_status_word(clusters), dg1, (size_t)2 * T * N * G, s));
            // the 16-bit copies and the bias-gradient sums as well (viewed as floats: every second value of a poisoned copy
            // is NaN - enough for every product to carry NaN into the gradient norm, on which the optimizer skips)
            FSN_TRY(fsn_launch_poison_if(flags + fsn_lstm2_g16_status_word(clusters), reinterpret_cast<float*>(dg16),
                                         (size_t)T * N * G, s));
            FSN_TRY(fsn_launch_poison_if(flags + fsn_lstm2_g16_status_word(clusters), dbp, (size_t)2 * clusters * G, s));
        } else {
            FSN_TRY(fsn_launch_lstm2_group_bptt(dh1, whh1T_p, wih1T_p, whh0T_p, sv0, sv1, dg0, dg1, dxbuf, flags, T, N, clusters,
                                                H, s, arith, w16));
            // both gate-gradient buffers (dg1 | dg0 are adjacent): every weight gradient and dx derive from them
            FSN_TRY(fsn_launch_poison_if(flags + fsn_lstm2_group_bptt_status_word(clusters), dg1, (size_t)2 * T * N * G, s));
        }
    }
    if (left > 0) {
        // the rows that do not fill a cluster: step by step on the auxiliary stream, straight into the same buffers
        hipStream_t as = cx->aux;
        const size_t stepH = (size_t)N * H, stepG = (size_t)N * G;
        for (int layer = 1; layer >= 0; --layer) {
            const float* sv = layer ? sv1 : sv0;
            const float* gates = sv + (size_t)row0 * G;
            const float* cseq = sv + (size_t)T * N * G + (size_t)row0 * H;
            float* dg = (layer ? dg1 : dg0) + (size_t)row0 * G;
            const float* whhT = layer ? whh1T_p : whh0T_p;
            for (int t = T - 1; t >= 0; --t) {
                const float* dh_t = layer ? dh1 + t * stepH + (size_t)row0 * H : dh0_left + (size_t)t * left * H;
                FSN_TRY(fsn_launch_bptt_step(dh_t, t + 1 < T ? dg + (t + 1) * stepG : dg, whhT, dc_left, gates + t * stepG,
                                             cseq + t * stepH, t ? cseq + (t - 1) * stepH : cseq, dg + t * stepG, left_tiles, H,
                                             t == T - 1, t == 0, as));
            }
            if (layer) {  // dh0 of these rows = dgates1 W_ih1: compact copy of their dgates1, one small GEMM
                if (hipMemcpy2DAsync(dg1_left, (size_t)left * G * sizeof(float), dg, stepG * sizeof(float),
                                     (size_t)left * G * sizeof(float), T, hipMemcpyDeviceToDevice, as) != hipSuccess) {
                    fsn_set_error("lstm2 backward: cannot gather the left-over rows");
                    return FSN_ERR_LAUNCH;
                }
                FsnGemmA a{};
                a.kind = 0;
                a.p0 = dg1_left;
                a.ld = G;
                FsnGemmC c{};
                c.kind = 3;
                c.p0 = dh0_left;
                c.ld = H;
                c.rows = T * left;
                c.cols = H;
                FSN_TRY(fsn_launch_gemm(a, wih1T_p, c, T * left_tiles, H / 16, G / 16, as));
            }
        }
        if (hipEventRecord(cx->ev_join, cx->aux) != hipSuccess || hipStreamWaitEvent(s, cx->ev_join, 0) != hipSuccess) {
            fsn_set_error("aux stream join failed");
            return FSN_ERR_LAUNCH;
        }
    }
    }  // chain_part
    if (dx && dx_part && in16) {
        // (the step-by-step rows' 16-bit copies first: the finish step of the products part may not have run yet)
        FSN_TRY(fsn_launch_g16_left_to16(dg0, dg16, T, N, row0, left, s, arith));
        FSN_TRY(fsn_launch_gemm_dx16(dg16, G, w_ih0, wdx16, dx, lddx, (long)T * N, G, I, s, arith));
    } else if (dx && dx_part) {
        FsnGemmA a{};
        FsnGemmC c{};
        a.kind = 0;
        a.p0 = dg0;
        a.ld = G;
        c.kind = 3;
        c.p0 = dx;
        c.ld = lddx;
        c.rows = T * N;
        c.cols = I;
        FSN_TRY(fsn_launch_gemm(a, wih0T_p, c, T * (N / 16), Ipad / 16, G / 16, s));
    }
    if (!products_part) return FSN_OK;
    // dW_ih = dgates^T X (+ db = its column sums: fp32 adds in every arithmetic), dW_hh = dgates_{1..}^T H_{0..T-2}
    // bias gradients = the BPTT launch's cluster sums + the step-by-step rows; those rows' 16-bit gate gradients (operands of
    // the products below: first when there are such rows; otherwise LAST - six tiny workgroups at the head of this part
    // queued behind whatever the caller's other stream was running and held the products back by its length)
    const bool finish_first = left > 0;
    if (g16 && finish_first)
        FSN_TRY(fsn_launch_lstm2_g16_finish(dg1, dg0, dg16 + (size_t)T * N * G, dg16, dbp, clusters, T, N, left, db1, db0, s, arith));
    if (tn16h) {
        // the three large products with both operands 16-bit in memory: dg16 = dg0 | dg1 written by the BPTT kernel, the
        // hidden sequences converted once (half the HBM bytes of the fp32 operands, LDS-DMA staging, no conversion pass)
        const size_t TNG = (size_t)T * N * G, TNH = (size_t)T * N * H;
        const unsigned short *dg16_0 = dg16, *dg16_1 = dg16 + TNG;
        if (!prepared && !prepare_part && !h16_saved) {
            FSN_TRY(fsn_launch_to16(hseq0, h16, TNH, arith, s));
            FSN_TRY(fsn_launch_to16(hseq1, h16 + TNH, TNH, arith, s));
        }
        (void)TNH;
        FSN_TRY(fsn_launch_gemm_tn16h(dg16_1, G, h16_0, ldh16, dw_ih1, H, G, H, (long)T * N, scratch, s, arith));
        FSN_TRY(fsn_launch_gemm_tn16h(dg16_1 + (size_t)N * G, G, h16_1, ldh16, dw_hh1, H, G, H, (long)(T - 1) * N, scratch, s, arith));
        FSN_TRY(fsn_launch_gemm_tn16h(dg16_0 + (size_t)N * G, G, h16_0, ldh16, dw_hh0, H, G, H, (long)(T - 1) * N, scratch, s, arith));
        if (in16) {  // x rounded once ([T N][32], its padding columns are zero), then the narrow product from 16-bit operands
            FSN_TRY(fsn_launch_to16(x, x16, (size_t)T * N * 32, arith, s));
            FSN_TRY(fsn_launch_gemm_tn16n(dg16_0, G, x16, 32, dw_ih0, I, G, I, (long)T * N, scratch, s, arith));
        } else {
            FSN_TRY(fsn_launch_gemm_tn(dg0, G, x, ldx, dw_ih0, I, G, I, (long)T * N, scratch, s, nullptr, arith));
        }
        if (!finish_first)
            FSN_TRY(fsn_launch_lstm2_g16_finish(dg1, dg0, dg16 + (size_t)T * N * G, dg16, dbp, clusters, T, N, left, db1, db0, s, arith));
        return FSN_OK;
    }
    if (g16) {  // (no plan for the 16-bit-operand products at this shape: the fp32 buffers; layer 1's were stored in that case)
        FSN_TRY(fsn_launch_gemm_tn(dg1, G, hseq0, H, dw_ih1, H, G, H, (long)T * N, scratch, s, nullptr, arith));
        FSN_TRY(fsn_launch_gemm_tn(dg0, G, x, ldx, dw_ih0, I, G, I, (long)T * N, scratch, s, nullptr, arith));
    } else {
        FSN_TRY(fsn_launch_gemm_tn(dg1, G, hseq0, H, dw_ih1, H, G, H, (long)T * N, scratch, s, db1, arith));
        FSN_TRY(fsn_launch_gemm_tn(dg0, G, x, ldx, dw_ih0, I, G, I, (long)T * N, scratch, s, db0, arith));
    }
    if (T > 1) {
        FSN_TRY(fsn_launch_gemm_tn(dg1 + (size_t)N * G, G, hseq1, H, dw_hh1, H, G, H, (long)(T - 1) * N, scratch, s, nullptr,
                                   arith));
        FSN_TRY(fsn_launch_gemm_tn(dg0 + (size_t)N * G, G, hseq0, H, dw_hh0, H, G, H, (long)(T - 1) * N, scratch, s, nullptr,
                                   arith));
    } else if (hipMemsetAsync(dw_hh1, 0, (size_t)G * H * sizeof(float), s) != hipSuccess ||
               hipMemsetAsync(dw_hh0, 0, (size_t)G * H * sizeof(float), s) != hipSuccess) {
        fsn_set_error("memset failed");
        return FSN_ERR_LAUNCH;
    }
    if (g16 && !finish_first)
        FSN_TRY(fsn_launch_lstm2_g16_finish(dg1, dg0, dg16 + (size_t)T * N * G, dg16, dbp, clusters, T, N, left, db1, db0, s, arith));
    return FSN_OK;
}

extern "C" int fsn_lstm2_backward(const float* dh1, const float* x, long ldx, const float* w_ih0, const float* w_hh0,
                                  const float* w_ih1, const float* w_hh1, int T, int N, int I, int H, const float* hseq0,
                                  const float* hseq1, const void* save0, const void* save1, float* dx, long lddx,
                                  float* dw_ih0, float* dw_hh0, float* db0, float* dw_ih1, float* dw_hh1, float* db1,
                                  void* workspace, size_t workspace_bytes, int arith, void* stream) {
    return lstm2_backward_phases(dh1, x, ldx, w_ih0, w_hh0, w_ih1, w_hh1, T, N, I, H, hseq0, hseq1, save0, save1, dx, lddx, dw_ih0,
                                 dw_hh0, db0, dw_ih1, dw_hh1, db1, workspace, workspace_bytes, arith, stream, 7);
}
extern "C" int fsn_lstm2_backward_phase(const float* dh1, const float* x, long ldx, const float* w_ih0, const float* w_hh0,
                                        const float* w_ih1, const float* w_hh1, int T, int N, int I, int H, const float* hseq0,
                                        const float* hseq1, const void* save0, const void* save1, float* dx, long lddx,
                                        float* dw_ih0, float* dw_hh0, float* db0, float* dw_ih1, float* dw_hh1, float* db1,
                                        void* workspace, size_t workspace_bytes, int arith, int phase, void* stream) {
    FSN_REQUIRE(phase >= 1 && phase <= 31 && (!(phase & 16) || (phase & 2)),
                "lstm2 backward parts %d: a sum of 1 (through time), 2 (weight-gradient products), 4 (dx), 8 (operand preparation), "
                "16 (with 2: prepared by an earlier part-8 call)", phase);
    return lstm2_backward_phases(dh1, x, ldx, w_ih0, w_hh0, w_ih1, w_hh1, T, N, I, H, hseq0, hseq1, save0, save1, dx, lddx, dw_ih0,
                                 dw_hh0, db0, dw_ih1, dw_hh1, db1, workspace, workspace_bytes, arith, stream, phase);
}

// ---- nn.GRU layer (sequence_model.py:59-66): forward (inference / training) + BPTT -----------------
extern "C" size_t fsn_gru_layer_save_bytes(int T, int N, int H) {
    return fsn_round_up_sz((size_t)T * N * 4 * H * sizeof(float), 256);  // r | z | n | hn
}
// Many rows in inference (the sub-band model of a GRU FullSubNet: B F rows, audio_zen/model/module/sequence_model.py:59-66
// under fullsubnet/model.py:121-128): the layer runs on the LSTM's persistent kernels with the GRU written as a four-gate
// cell (FSN_REC_GRU, lstm_kernels.hip) - lstm_rec_in_kernel for a narrow row-major input (<= 32 columns: the projection is
// formed inside), lstm_rec_x_kernel for the layer above an equally wide one (input = its hidden sequence, no projection
// GEMM, no gx round trip).  Whole rounds of 2 - 4 row tiles per workgroup; the few left-over tiles advance step by step
// on the auxiliary stream beside the persistent launch, on compact copies of their rows.
struct GruPlan {
    int rt, main_wgs, left_tiles;
};
static GruPlan gru_layer_plan(int N, int I, long ldx, int H) {
    GruPlan p{0, 0, N / 16};
    const int Ipad = fsn_round_up(I, 16);
    if (H != 384 || !((Ipad <= 32 && (ldx <= 0 || ldx >= Ipad)) || (I == H && (ldx <= 0 || ldx == H)))) return p;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    const int tiles = N / 16;
    // Few rows per CU: the step launches spread a step over more workgroups.  The persistent kernels take 2 - 4 row tiles per
    // workgroup, so up to 2 x CUs tiles they leave CUs idle and cost what 2 x CUs tiles cost (GRU FullSubNet, 190 frames:
    // 31.5 - 32.9 ms from 16 to 32 utterances; step by step 1.77 ms per utterance: 26.5 ms at 15, ~30 at 17): from 9/8 x CUs on
    if (tiles < cus + cus / 8) return p;
    long best = -1;
    for (int rt = 4; rt >= 2; --rt) {
        // whole rounds of rt tiles on every CU, or ONE round of fewer workgroups; a left-over tile costs about a
        // hundredth of a tile of a resident workgroup (layer_plan's measure)
        int wgs = 0;
        const int rounds = tiles / (cus * rt);
        if (rounds >= 1) wgs = rounds * cus;
        else if (tiles / rt <= cus) wgs = tiles / rt;
        if (wgs < 1) continue;
        const int left = tiles - wgs * rt;
        const long cost = (long)((wgs + cus - 1) / cus) * rt * 100 + left;
        if (best < 0 || cost < best) {
            best = cost;
            p = GruPlan{rt, wgs, left};
        }
    }
    return p;
}
static size_t gru_layer_step_workspace_bytes(int T, int N, int I, int H) {
    Carver cv(nullptr);
    cv.take<float>((size_t)3 * H * fsn_round_up(I, 16));
    cv.take<float>((size_t)3 * H * H);
    cv.take<float>((size_t)3 * H);
    cv.take<float>((size_t)T * N * 3 * H);
    return fsn_round_up_sz(cv.off, 256);
}
extern "C" size_t fsn_gru_layer_fwd_workspace_bytes(int T, int N, int I, int H) {
    if (T < 1 || N < 16 || I < 1 || H < 64) return 0;
    const size_t Ipad = fsn_round_up(I, 16), G4 = 4 * (size_t)H;
    const GruPlan p = gru_layer_plan(N, I, 0, H);
    Carver cv(nullptr);
    if (p.main_wgs > 0) {  // the persistent form's own buffers first, the step form's region (left-over rows) behind them
        cv.take<float>(G4 * Ipad + G4 * H);  // the four-gate matrices as expanded ...
        cv.take<float>(G4 * Ipad + G4 * H);  // ... and in fragment order, W_hh right behind W_ih
        cv.take<float>(G4);
        cv.take<float>((size_t)T * p.left_tiles * 16 * Ipad);
        cv.take<float>((size_t)T * p.left_tiles * 16 * H);
    }
    cv.take<char>(gru_layer_step_workspace_bytes(T, N, I, H));
    return fsn_round_up_sz(cv.off, 256);
}
extern "C" int fsn_gru_layer_is_persistent(int T, int N, int I, long ldx, int H) {
    return T >= 1 && N >= 16 && N % 16 == 0 && I >= 1 && gru_layer_plan(N, I, ldx, H).main_wgs > 0 ? 1 : 0;
}

// The step form in two halves: weights re-tiled + input projection of all steps (one GEMM), then the T dependent step launches
// (gru_step_kernel: 32 registers, 12 KB of LDS - it fits beside a resident workgroup of the persistent kernels).
struct GruStepBufs {
    float *whh_p, *gx;
};
static int gru_layer_steps_prepare(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                                   const float* b_hh, int T, int N, int I, int H, void* workspace, hipStream_t s, GruStepBufs* out) {
    const int Ipad = fsn_round_up(I, 16), G = 3 * H;
    Carver cv(workspace);
    float* wih_p = cv.take<float>((size_t)G * Ipad);
    float* whh_p = cv.take<float>((size_t)G * H);
    float* bias = cv.take<float>((size_t)G);
    float* gx = cv.take<float>((size_t)T * N * G);
    FSN_TRY(fsn_launch_pack(w_ih, wih_p, G, I, G, Ipad, s));
    FSN_TRY(fsn_launch_pack(w_hh, whh_p, G, H, G, H, s));
    // bias of the projection: b_ih everywhere + b_hh for r and z (b_hn stays inside r * (W_hn h + b_hn))
    FSN_TRY(fsn_launch_bias_sum(b_ih, nullptr, bias, G, G, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih, b_hh, bias, 2 * H, 2 * H, s));
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = x;
    a.ld = ldx;
    FsnGemmC c{};
    c.kind = 0;
    c.p0 = gx;
    c.bias = bias;
    FSN_TRY(fsn_launch_gemm(a, wih_p, c, T * (N / 16), G / 16, Ipad / 16, s));
    out->whh_p = whh_p;
    out->gx = gx;
    return FSN_OK;
}
static int gru_layer_steps_run(const GruStepBufs& b, const float* b_hh, int T, int N, int H, float* hseq, float* sv, hipStream_t s,
                               int beside_persistent = 0) {
    const size_t step = (size_t)N * H;
    for (int t = 0; t < T; ++t)
        FSN_TRY(fsn_launch_gru_step(b.gx, b.whh_p, b_hh + 2 * H, t ? hseq + (t - 1) * step : hseq, hseq + t * step,
                                    sv ? sv + (size_t)t * N * 4 * H : nullptr, (long)t * (N / 16), N / 16, H, t == 0, s,
                                    beside_persistent));
    return FSN_OK;
}
static int gru_layer_forward_steps(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                                   const float* b_hh, int T, int N, int I, int H, float* hseq, float* sv, void* workspace,
                                   hipStream_t s) {
    GruStepBufs b{};
    FSN_TRY(gru_layer_steps_prepare(x, ldx, w_ih, w_hh, b_ih, b_hh, T, N, I, H, workspace, s, &b));
    return gru_layer_steps_run(b, b_hh, T, N, H, hseq, sv, s);
}

static int gru_layer_forward_persistent(const GruPlan& p, const float* x, long ldx, const float* w_ih, const float* w_hh,
                                        const float* b_ih, const float* b_hh, int T, int N, int I, int H, float* hseq,
                                        void* workspace, hipStream_t s) {
    const int Ipad = fsn_round_up(I, 16), G4 = 4 * H;
    const int left = p.left_tiles * 16, main_rows = N - left;
    Carver cv(workspace);
    float* w4 = cv.take<float>((size_t)G4 * Ipad + (size_t)G4 * H);
    float* w4p = cv.take<float>((size_t)G4 * Ipad + (size_t)G4 * H);
    float* b4 = cv.take<float>((size_t)G4);
    float* x_left = cv.take<float>((size_t)T * left * Ipad);
    float* h_left = cv.take<float>((size_t)T * left * H);
    void* step_ws = cv.take<char>(0);
    float *wih4 = w4, *whh4 = w4 + (size_t)G4 * I, *wih4_p = w4p, *whh4_p = w4p + (size_t)G4 * Ipad;
    FSN_TRY(fsn_launch_gru_expand4(w_ih, w_hh, b_ih, b_hh, wih4, whh4, b4, I, H, s, 1));
    FSN_TRY(fsn_launch_pack(wih4, wih4_p, G4, I, G4, Ipad, s));
    FSN_TRY(fsn_launch_pack(whh4, whh4_p, G4, H, G4, H, s));
    hipStream_t ls = s;
    StreamCtx* cx = nullptr;
    GruStepBufs sb{};
    if (left > 0) {
        // rows [main_rows, N) of every step as compact [T][left] matrices (columns [0, Ipad) of a row; one 2-D copy when the
        // rows are exactly that wide, one per step otherwise) and their input projection - on `s`, AHEAD of the persistent
        // launch: the projection GEMM's workgroups (160 registers, 96 KB of LDS) do not fit beside a resident workgroup of it
        // and would wait for the whole launch (measured: the step launches then ran after it, +2.7 ms per batch of 64)
        bool ok = true;
        if (ldx == Ipad)
            ok = hipMemcpy2DAsync(x_left, (size_t)left * Ipad * sizeof(float), x + (size_t)main_rows * ldx,
                                  (size_t)N * ldx * sizeof(float), (size_t)left * Ipad * sizeof(float), (size_t)T,
                                  hipMemcpyDeviceToDevice, s) == hipSuccess;
        else
            for (int t = 0; t < T && ok; ++t)
                ok = hipMemcpy2DAsync(x_left + (size_t)t * left * Ipad, (size_t)Ipad * sizeof(float),
                                      x + ((size_t)t * N + main_rows) * ldx, (size_t)ldx * sizeof(float),
                                      (size_t)Ipad * sizeof(float), (size_t)left, hipMemcpyDeviceToDevice, s) == hipSuccess;
        if (!ok) {
            fsn_set_error("gru layer forward: copy of the left-over rows failed");
            return FSN_ERR_LAUNCH;
        }
        FSN_TRY(gru_layer_steps_prepare(x_left, Ipad, w_ih, w_hh, b_ih, b_hh, T, left, I, H, step_ws, s, &sb));
        cx = cur_ctx();
        FSN_TRY(aux_init(cx));
        if (hipEventRecord(cx->ev_fork, s) != hipSuccess || hipStreamWaitEvent(cx->aux, cx->ev_fork, 0) != hipSuccess) {
            fsn_set_error("aux stream fork failed");
            return FSN_ERR_LAUNCH;
        }
        ls = cx->aux;
    }
    if (Ipad <= 32) {
        FsnSbInput xin{};
        xin.x_rows = x;
        xin.x_ld = ldx;
        xin.x_step = N;
        xin.N = main_rows;
        xin.kin_chunks = Ipad / 16;
        xin.wih_p = wih4_p;
        xin.bias = b4;
        FSN_TRY(fsn_launch_lstm_rec_in(&xin, whh4_p, hseq, T, N, H, p.rt, p.main_wgs, s, 1));
    } else {
        FSN_TRY(fsn_launch_lstm_rec_x(x, wih4_p, whh4_p, b4, T, N, H, p.rt, p.main_wgs, s, nullptr, hseq, 1));
    }
    if (left > 0) {
        // the left-over rows' T step launches beside the persistent launch, then back into rows [main_rows, N) of hseq
        FSN_TRY(gru_layer_steps_run(sb, b_hh, T, left, H, h_left, nullptr, ls, 1));
        if (hipMemcpy2DAsync(hseq + (size_t)main_rows * H, (size_t)N * H * sizeof(float), h_left, (size_t)left * H * sizeof(float),
                             (size_t)left * H * sizeof(float), (size_t)T, hipMemcpyDeviceToDevice, ls) != hipSuccess) {
            fsn_set_error("gru layer forward: copy of the left-over rows failed");
            return FSN_ERR_LAUNCH;
        }
        if (hipEventRecord(cx->ev_join, cx->aux) != hipSuccess || hipStreamWaitEvent(s, cx->ev_join, 0) != hipSuccess) {
            fsn_set_error("aux stream join failed");
            return FSN_ERR_LAUNCH;
        }
    }
    return FSN_OK;
}

extern "C" int fsn_gru_layer_forward(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                                     const float* b_hh, int T, int N, int I, int H, float* hseq, void* save,
                                     size_t save_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    FSN_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && hseq && workspace, "NULL pointer argument");
    if ((save && save_bytes < fsn_gru_layer_save_bytes(T, N, H)) ||
        workspace_bytes < fsn_gru_layer_fwd_workspace_bytes(T, N, I, H)) {
        fsn_set_error("gru layer forward: save / workspace buffer too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!save) {
        const GruPlan p = gru_layer_plan(N, I, ldx, H);
        if (p.main_wgs > 0) return gru_layer_forward_persistent(p, x, ldx, w_ih, w_hh, b_ih, b_hh, T, N, I, H, hseq, workspace, s);
    }
    return gru_layer_forward_steps(x, ldx, w_ih, w_hh, b_ih, b_hh, T, N, I, H, hseq, static_cast<float*>(save), workspace, s);
}

// Streaming form (chunked / frame-by-frame inference with carried state): T more steps from h_state [N][H], which is
// updated in place (nn.GRU(x, h_0) is the analogue).  Same kernels and workspace as the offline forward.
extern "C" int fsn_gru_layer_forward_state(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                                           const float* b_hh, int T, int N, int I, int H, float* hseq, float* h_state,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    FSN_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && hseq && h_state && workspace, "NULL pointer argument");
    if (workspace_bytes < fsn_gru_layer_fwd_workspace_bytes(T, N, I, H)) {
        fsn_set_error("gru layer forward (state): workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ipad = fsn_round_up(I, 16), G = 3 * H;
    Carver cv(workspace);
    float* wih_p = cv.take<float>((size_t)G * Ipad);
    float* whh_p = cv.take<float>((size_t)G * H);
    float* bias = cv.take<float>((size_t)G);
    float* gx = cv.take<float>((size_t)T * N * G);
    FSN_TRY(fsn_launch_pack(w_ih, wih_p, G, I, G, Ipad, s));
    FSN_TRY(fsn_launch_pack(w_hh, whh_p, G, H, G, H, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih, nullptr, bias, G, G, s));
    FSN_TRY(fsn_launch_bias_sum(b_ih, b_hh, bias, 2 * H, 2 * H, s));
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = x;
    a.ld = ldx;
    FsnGemmC c{};
    c.kind = 0;
    c.p0 = gx;
    c.bias = bias;
    FSN_TRY(fsn_launch_gemm(a, wih_p, c, T * (N / 16), G / 16, Ipad / 16, s));
    const size_t step = (size_t)N * H;
    for (int t = 0; t < T; ++t)
        FSN_TRY(fsn_launch_gru_step(gx, whh_p, b_hh + 2 * H, t ? hseq + (t - 1) * step : h_state, hseq + t * step, nullptr,
                                    (long)t * (N / 16), N / 16, H, 0, s));
    if (hipMemcpyAsync(h_state, hseq + (size_t)(T - 1) * step, step * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        fsn_set_error("gru layer forward (state): state copy failed");
        return FSN_ERR_LAUNCH;
    }
    return FSN_OK;
}

extern "C" size_t fsn_gru_layer_bwd_workspace_bytes(int T, int N, int I, int H) {
    const int Ipad = fsn_round_up(I, 16), G = 3 * H;
    Carver cv(nullptr);
    cv.take<float>((size_t)H * G);      // W_hh^T fragments
    cv.take<float>((size_t)Ipad * G);   // W_ih^T fragments
    cv.take<float>((size_t)T * N * G);  // dgx
    cv.take<float>((size_t)T * N * H);  // dghn
    cv.take<float>((size_t)N * H);      // carry
    // the scratch of the weight-gradient products: every (M, Nc) that fsn_gru_layer_backward forms (each shape has
    // its own plan - a narrower product may split K further than the 3H-row one)
    size_t tn = fsn_gemm_tn_workspace_bytes(G, I, (long)T * N);
    for (const int m : {G, 2 * H, H}) {
        const size_t b = fsn_gemm_tn_workspace_bytes(m, H, (long)T * N);
        tn = tn > b ? tn : b;
    }
    size_t cs = 0;  // the column sums fsn_gru_layer_backward forms: 3H, 2H and H columns (each with its own row blocking)
    for (const int c : {G, 2 * H, H}) {
        const size_t b = fsn_colsum_workspace_bytes(c, (long)T * N);
        cs = cs > b ? cs : b;
    }
    cv.take<char>(tn > cs ? tn : cs);
    return fsn_round_up_sz(cv.off, 256);
}

extern "C" int fsn_gru_layer_backward(const float* dh, const float* x, long ldx, const float* w_ih, const float* w_hh,
                                      int T, int N, int I, int H, const float* hseq, const void* save, float* dx,
                                      long lddx, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_TRY(check_lstm_layer(T, N, I, H, ldx));
    FSN_REQUIRE(dh && x && w_ih && w_hh && hseq && save && dw_ih && dw_hh && db_ih && db_hh && workspace,
                "NULL pointer argument");
    FSN_REQUIRE(!dx || lddx >= I, "dx row stride %ld < I", lddx);
    if (workspace_bytes < fsn_gru_layer_bwd_workspace_bytes(T, N, I, H)) {
        fsn_set_error("gru layer backward: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ipad = fsn_round_up(I, 16), G = 3 * H;
    Carver cv(workspace);
    float* whhT_p = cv.take<float>((size_t)H * G);
    float* wihT_p = cv.take<float>((size_t)Ipad * G);
    float* dgx = cv.take<float>((size_t)T * N * G);
    float* dghn = cv.take<float>((size_t)T * N * H);
    float* carry = cv.take<float>((size_t)N * H);
    void* scratch = cv.take<char>(0);  // the rest of the workspace (sized by fsn_gru_layer_bwd_workspace_bytes)
    const float* sv = static_cast<const float*>(save);
    FSN_TRY(fsn_launch_pack(w_hh, whhT_p, H, G, H, G, s, 1, H));
    FSN_TRY(fsn_launch_pack(w_ih, wihT_p, I, G, Ipad, G, s, 1, I));
    const size_t step = (size_t)N * H;
    for (int t = T - 1; t >= 0; --t) {
        const size_t tn1 = t + 1 < T ? (size_t)(t + 1) : 0;
        FSN_TRY(fsn_launch_gru_bptt_step(dh + t * step, dgx + tn1 * N * G, dghn + tn1 * step, whhT_p, carry,
                                         sv + (size_t)t * N * 4 * H, t ? hseq + (t - 1) * step : hseq,
                                         dgx + (size_t)t * N * G, dghn + t * step, N / 16, H, t == T - 1, t == 0, s));
    }
    if (dx) {
        FsnGemmA a{};
        a.kind = 0;
        a.p0 = dgx;
        a.ld = G;
        FsnGemmC c{};
        c.kind = 3;
        c.p0 = dx;
        c.ld = lddx;
        c.rows = T * N;
        c.cols = I;
        FSN_TRY(fsn_launch_gemm(a, wihT_p, c, T * (N / 16), Ipad / 16, G / 16, s));
    }
    FSN_TRY(fsn_launch_gemm_tn(dgx, G, x, ldx, dw_ih, I, G, I, (long)T * N, scratch, s));
    if (T > 1) {
        // dW_hh: rows r, z from the x-side derivatives (identical on the h side), rows n from dghn
        FSN_TRY(fsn_launch_gemm_tn(dgx + (size_t)N * G, G, hseq, H, dw_hh, H, 2 * H, H, (long)(T - 1) * N, scratch, s));
        FSN_TRY(fsn_launch_gemm_tn(dghn + step, H, hseq, H, dw_hh + (size_t)2 * H * H, H, H, H, (long)(T - 1) * N,
                                   scratch, s));
    } else if (hipMemsetAsync(dw_hh, 0, (size_t)G * H * sizeof(float), s) != hipSuccess) {
        fsn_set_error("memset failed");
        return FSN_ERR_LAUNCH;
    }
    FSN_TRY(fsn_launch_colsum(dgx, G, db_ih, G, (long)T * N, scratch, s));
    FSN_TRY(fsn_launch_colsum(dgx, G, db_hh, 2 * H, (long)T * N, scratch, s));
    return fsn_launch_colsum(dghn, H, db_hh + 2 * H, H, (long)T * N, scratch, s);
}

// ---- training step: nn.Linear (sequence_model.py:82-84) forward / backward ------------------------
// x [R][ldx] (columns I..ldx-1 zero, ldx = round_up(I,16)), w [O][I], b [O] -> y [R][O] (+ ReLU).
extern "C" size_t fsn_linear_workspace_bytes(int R, int I, int O) {
    const int Ip = fsn_round_up(I, 16), Op = fsn_round_up(O, 16);
    Carver cv(nullptr);
    cv.take<float>((size_t)Op * Ip);  // W (forward) or W^T (backward) fragments
    cv.take<float>((size_t)Op);       // padded bias
    size_t tn = fsn_gemm_tn_workspace_bytes(O, I, R);
    const size_t cs = fsn_colsum_workspace_bytes(O, R);
    cv.take<char>(tn > cs ? tn : cs);
    return fsn_round_up_sz(cv.off, 256);
}

extern "C" int fsn_linear_forward(const float* x, long ldx, const float* w, const float* b, int R, int I, int O,
                                  int relu, float* y, void* workspace, size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(x && w && b && y && workspace, "NULL pointer argument");
    FSN_REQUIRE(R >= 1 && I >= 1 && O >= 1 && ldx >= fsn_round_up(I, 16) && ldx % 4 == 0, "linear: bad shape");
    if (workspace_bytes < fsn_linear_workspace_bytes(R, I, O)) {
        fsn_set_error("linear: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (fsn_linear_small_out_ok(I, O, ldx))  // a handful of outputs: bandwidth-bound row dot products, no padded GEMM
        return fsn_launch_linear_small_out(x, ldx, w, b, y, R, I, O, relu, s);
    const int Ip = fsn_round_up(I, 16), Op = fsn_round_up(O, 16);
    Carver cv(workspace);
    float* wp = cv.take<float>((size_t)Op * Ip);
    float* bp = cv.take<float>((size_t)Op);
    FSN_TRY(fsn_launch_pack(w, wp, O, I, Op, Ip, s));
    FSN_TRY(fsn_launch_bias_sum(b, nullptr, bp, O, Op, s));
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = x;
    a.ld = ldx;
    FsnGemmC c{};
    c.kind = 3;
    c.p0 = y;
    c.bias = bp;
    c.ld = O;
    c.rows = R;
    c.cols = O;
    c.la = relu ? 1 : 0;  // kind 3: la doubles as the ReLU flag
    a.N = R;
    return fsn_launch_gemm(a, wp, c, (R + 15) / 16, Op / 16, Ip / 16, s);
}

// dy [R][lddy] (columns O..lddy-1 zero, lddy = round_up(O,16)) -> dx [R][lddx] (may be NULL), dw [O][I], db [O]
extern "C" int fsn_linear_backward(const float* dy, long lddy, const float* x, long ldx, const float* w, int R, int I,
                                   int O, float* dx, long lddx, float* dw, float* db, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    CallScope scope(stream);
    FSN_REQUIRE(dy && x && w && workspace && (dx || dw) && (dw == nullptr) == (db == nullptr),
                "linear backward: NULL pointer argument (dx alone, dw + db alone, or all three)");
    FSN_REQUIRE(R >= 1 && I >= 1 && O >= 1 && lddy >= fsn_round_up(O, 16) && lddy % 4 == 0 && ldx >= I,
                "linear backward: bad shape");
    if (workspace_bytes < fsn_linear_workspace_bytes(R, I, O)) {
        fsn_set_error("linear: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ip = fsn_round_up(I, 16), Op = fsn_round_up(O, 16);
    Carver cv(workspace);
    float* wtp = cv.take<float>((size_t)Op * Ip);
    cv.take<float>((size_t)Op);
    size_t tn = fsn_gemm_tn_workspace_bytes(O, I, R);
    const size_t cs = fsn_colsum_workspace_bytes(O, R);
    void* scratch = cv.take<char>(tn > cs ? tn : cs);
    if (dx && fsn_linear_small_out_ok(I, O, lddx)) {
        FSN_TRY(fsn_launch_linear_small_dx(dy, lddy, w, dx, lddx, R, I, O, s));
    } else if (dx) {
        // dX = dY W: "weights" W^T (out = I, k = O) = the stored [O][I] read transposed
        FSN_TRY(fsn_launch_pack(w, wtp, I, O, Ip, Op, s, 1, I));
        FsnGemmA a{};
        a.kind = 0;
        a.p0 = dy;
        a.ld = lddy;
        FsnGemmC c{};
        c.kind = 3;
        c.p0 = dx;
        c.ld = lddx;
        c.rows = R;
        c.cols = I;
        a.N = R;
        FSN_TRY(fsn_launch_gemm(a, wtp, c, (R + 15) / 16, Ip / 16, Op / 16, s));
    }
    if (!dw) return FSN_OK;  // the input gradient alone (the parameter gradients by a second call, possibly on another stream)
    FSN_TRY(fsn_launch_gemm_tn(dy, lddy, x, ldx, dw, I, O, I, R, scratch, s));
    return fsn_launch_colsum(dy, lddy, db, O, R, scratch, s);
}
