// libemx: C ABI + host runtime of the MI355X split-ensemble sampler hot path.
// See include/emx.h for the contract and the reference lines each entry point replaces.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/emx.h"
#include "emx_internal.hpp"
#include "emx_kernels.hpp"
#include "emx_launch.hpp"
#include "emx_mtdev.hpp"
#include "emx_mtjump.hpp"
#include "emx_mtpipe.hpp"
#include "emx_rng.hpp"
#include "mt19937_legacy.hpp"

using namespace emx;

namespace {

std::string g_err;

constexpr int PLAN_RING = 2 * NATIVE_BATCH_MAX;     // two native batches (the upper half serves the quarantined graph replay)
constexpr int PIPE_SINKS = PLAN_RING < 16 ? PLAN_RING : 16;      // exact-mode plan pipeline: pinned staging buffers in flight (emx_ctx::pipe_nsinks: the
                                                                 // whole ring for ensembles that take their steps eight per persistent launch)
constexpr int MTDEV_SLOTS = MTDEV_NBUF * MTDEV_BATCH;            // exact-mode device producer: its plan slots follow the ring's, allocated on first use
static_assert(MTDEV_BATCH == NATIVE_BATCH_MAX, "a produced batch is one persistent launch");
constexpr int EMX_MAX_RANKS = 1024;      // pull / all-gather exchanges: counter storage

// ------------------------------------------------------------------------------------------
// exact (NumPy-stream) plan of one step: every draw red_blue.py / stretch.py / de.py /
// de_snooker.py make for one propose(), in their order.
// ------------------------------------------------------------------------------------------
// proposal draws of ONE get_proposal call (stretch.py:30-32 | de.py:49-56 | de_snooker.py:37-40)
template <typename I32, typename F64>
int draw_split_proposal(MT19937Legacy& mt, int64_t N, const emx_move_desc& mv, const int32_t* off, const I32* order,
                        int split, I32* p0, I32* p1, I32* p2, F64* s0) {
    const int S = mv.nsplits;
    {
        const int64_t base = off[split], ns = off[split + 1] - off[split], nc = N - ns;
        auto comp = [&](uint64_t r) -> int32_t {  // complement = concatenation of the other sets (stretch.py:27)
            return (int64_t)r < base ? order[r] : order[r + ns];
        };
        if (mv.kind == EMX_MOVE_STRETCH) {
            if constexpr (std::is_same<F64, double>::value) {
                mt.fill_doubles(&s0[base], ns);                             // stretch.py:30
                for (int64_t t = 0; t < ns; ++t) {
                    const double tt = (mv.a - 1.0) * s0[base + t] + 1.0;
                    s0[base + t] = tt * tt / mv.a;
                }
            } else {
                for (int64_t t = 0; t < ns; ++t) {
                    const double u = mt.next_double();
                    const double tt = (mv.a - 1.0) * u + 1.0;
                    s0[base + t] = tt * tt / mv.a;
                }
            }
            mt.fill_randint(ns, (uint64_t)nc, [&](int64_t t, uint64_t r) { p0[base + t] = comp(r); });   // stretch.py:32
            for (int64_t t = 0; t < ns; ++t) p1[base + t] = p2[base + t] = order[base + t];
        } else if (mv.kind == EMX_MOVE_DE) {
            const uint64_t pop = (uint64_t)nc * (uint64_t)(nc - 1);
            for (int64_t t = 0; t < ns; ++t) {                              // de.py:49-50
                uint64_t f, s;
                de_pair(mt.randint(pop), (uint64_t)nc, f, s);
                p0[base + t] = comp(f);
                p1[base + t] = comp(s);
                p2[base + t] = order[base + t];
            }
            for (int64_t t = 0; t < ns; ++t) {                              // de.py:56
                const double g = mt.next_gauss();
                s0[base + t] = mv.g0 * (1.0 + mv.sigma * g);
            }
        } else if (mv.kind == EMX_MOVE_SNOOKER) {
            int cs[3], q = 0;
            for (int s = 0; s < S && q < 3; ++s)
                if (s != split) cs[q++] = s;
            for (int64_t t = 0; t < ns; ++t) {                              // de_snooker.py:37-40
                int32_t w[3];
                for (int k = 0; k < 3; ++k) {
                    const int j = cs[k];
                    const uint64_t r = mt.randint((uint64_t)(off[j + 1] - off[j]));
                    w[k] = order[off[j] + (int64_t)r];
                }
                for (int i = 2; i > 0; --i) {                               // shuffle(w)
                    const int j = (int)mt.random_interval((uint64_t)i);
                    std::swap(w[i], w[j]);
                }
                p0[base + t] = w[0];
                p1[base + t] = w[1];
                p2[base + t] = w[2];
                s0[base + t] = 0.0;
            }
        } else {
            return -1;
        }
    }
    return 0;
}

template <typename I32, typename F64>
int make_exact_plan(MT19937Legacy& mt, int64_t N, int32_t D, const emx_move_desc& mv, std::vector<uint8_t>& labels,
                    int32_t* off, I32* order, I32* p0, I32* p1, I32* p2, F64* s0, F64* uacc) {
    const int S = mv.nsplits;
    if (S < 2 || S > 255) return -1;
    if (mv.kind == EMX_MOVE_SNOOKER && S < 4) return -1;
    labels.resize(N);
    for (int64_t i = 0; i < N; ++i) labels[i] = (uint8_t)(i % S);          // red_blue.py:78 (one byte per label: L1-resident shuffle)
    if (mv.randomize_split) mt.shuffle(labels.data(), N);                   // red_blue.py:80
    // boolean-mask gather order (red_blue.py:85): ascending walker index inside each set
    std::vector<int32_t> cnt(S + 1, 0);
    for (int64_t i = 0; i < N; ++i) cnt[labels[i] + 1]++;
    for (int s = 0; s < S; ++s) cnt[s + 1] += cnt[s];
    for (int s = 0; s <= S; ++s) off[s] = cnt[s];
    {
        std::vector<int32_t> cur(cnt.begin(), cnt.end() - 1);
        for (int64_t i = 0; i < N; ++i) order[cur[labels[i]]++] = (int32_t)i;
    }
    for (int split = 0; split < S; ++split) {
        const int64_t base = off[split], ns = off[split + 1] - off[split];
        if (draw_split_proposal(mt, N, mv, off, order, split, p0, p1, p2, s0) != 0) return -1;
        if constexpr (std::is_same<F64, double>::value)
            mt.fill_doubles(&uacc[base], ns);                                 // red_blue.py:100
        else
            for (int64_t t = 0; t < ns; ++t) uacc[base + t] = mt.next_double();
    }
    (void)D;
    return 0;
}

// One Gaussian Metropolis step's draws in the reference's order (gaussian.py:81-97, mh.py:50-57):
// [factor uniform] -> randn(N, D) row-major -> [randint(D) per walker] -> (log-prob) -> rand(N).
// `cursor` is the sequential mode's coordinate counter (advanced here).
static double make_exact_gauss(MT19937Legacy& mt, int64_t N, int32_t D, const emx_move_desc& mv, double& cursor,
                               int32_t* off, int32_t* order, int32_t* p0, int32_t* p1, int32_t* p2, double* s0,
                               double* uacc, double* normals) {
    double f = 1.0;
    if (mv.a != 0.0) {
        const double u = mt.next_double();
        f = std::exp(-mv.g0 + (mv.g0 - (-mv.g0)) * u);        // legacy uniform: low + (high - low) * u
    }
    for (int64_t k = 0; k < N * (int64_t)D; ++k) normals[k] = mt.next_gauss();
    int col = -1;
    if (mv.reserved == EMX_GAUSS_SEQUENTIAL) {
        col = (int)((int64_t)cursor % D);
        cursor = (double)(((int64_t)cursor + 1) % D);
    }
    off[0] = 0;
    off[1] = (int32_t)N;
    for (int64_t i = 0; i < N; ++i) {
        order[i] = p1[i] = p2[i] = (int32_t)i;
        s0[i] = 0.0;
        p0[i] = mv.reserved == EMX_GAUSS_RANDOM ? (int32_t)mt.randint((uint64_t)D) : col;
    }
    for (int64_t i = 0; i < N; ++i) uacc[i] = mt.next_double();
    return f;
}

template <int MOVE>
void host_native_plan(const NativeArgs& na, int64_t N, const emx_move_desc& mv, int32_t* off, int32_t* order,
                      int32_t* p0, int32_t* p1, int32_t* p2, double* s0, double* uacc) {
    const int S = mv.nsplits;
    off[0] = 0;
    for (int s = 0; s < S; ++s) off[s + 1] = off[s] + (int32_t)((N - s + S - 1) / S);
    for (int s = 0; s < S; ++s)
        for (int t = 0; t < off[s + 1] - off[s]; ++t) {
            const int pos = off[s] + t;
            int i, a0, a1, a2;
            double z, u;
            native_slot<MOVE>(na, (int)N, S, s, t, mv.a, mv.sigma, mv.g0, i, a0, a1, a2, z, u);
            order[pos] = i;
            p0[pos] = a0;
            p1[pos] = a1;
            p2[pos] = a2;
            s0[pos] = z;
            uacc[pos] = u;
        }
}

int philox_move_choice(uint64_t seed, uint64_t step, const double* cdf, int n) { return native_move_choice(seed, step, cdf, n); }

// ---- RCCL, resolved at run time (the process may already hold PyTorch's copy) ----------------
struct RcclId {
    char internal[128];
};
struct RcclApi {
    void* h = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllToAll)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;     // RCCL extension
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
} g_rccl;
constexpr int RCCL_FLOAT64 = 8;   // ncclFloat64 (rccl.h:467)

int rccl_load(const char* path, std::string& err) {
    if (g_rccl.h) return 0;
    const char* cands[] = {path, getenv("EMX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* cnd : cands) {
        if (!cnd || !*cnd) continue;
        h = dlopen(cnd, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
        return -5;
    }
    g_rccl.GetUniqueId = (int (*)(RcclId*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.AllToAll = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllToAll");
    g_rccl.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
    g_rccl.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
    g_rccl.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) {
        err = "librccl lacks ncclGetUniqueId/ncclCommInitRank/ncclAllGather";
        return -5;
    }
    g_rccl.h = h;
    return 0;
}

struct Shape {
    int G, V, CH;
};

// Layout of a plan slot, the same in the device block and in its pinned staging buffer: [order | p0] int32, [s0 | uacc]
// f64, [p1 | p2] int32 -- what a stretch step needs (one partner, 24 bytes per walker) is a prefix, so its upload stops
// there; DE / snooker / Gaussian plans go up whole (32 bytes per walker).  (The device block continues with logu | fac.)
struct HostPlan {
    int32_t *order, *p0, *p1, *p2;
    double *s0, *uacc;
    HostPlan(char* base, size_t N)
        : order((int32_t*)base), p0((int32_t*)base + N), p1((int32_t*)(base + N * 24)), p2((int32_t*)(base + N * 24) + N),
          s0((double*)(base + N * 8)), uacc((double*)(base + N * 8) + N) {}
};
inline size_t plan_upload_bytes(size_t N, int move_kind) { return N * (move_kind == EMX_MOVE_STRETCH ? 24 : 32); }

// Row layout for a row of `Dcover` doubles: G lanes x CH chunks x V doubles, chosen to minimise
// the instructions per walker (few lanes per walker -> short cross-lane reductions, many walkers per
// pass) while every chunk of a row is still read as whole 128-byte lines (G*V*8 >= 128 B).
constexpr int shape_g(int cols) {
    return cols <= 4 ? 4 : cols <= 32 ? 8 : cols <= 64 ? 16 : cols <= 128 ? 32 : 64;
}
constexpr int shape_ch(int cols) {
    return cols <= 8 ? 1 : cols <= 16 ? 2 : cols <= 256 ? 4 : cols <= 512 ? 8 : 16;
}

Shape pick_shape(int D, int Dcover) {
    Shape s;
    s.V = (D % 2 == 0) ? 2 : 1;
    const int cols = (Dcover + s.V - 1) / s.V;
    s.G = shape_g(cols);
    s.CH = shape_ch(cols);
    return s;
}

}  // namespace

// ------------------------------------------------------------------------------------------
struct emx_mt {
    MT19937Legacy mt;
};

struct emx_ctx {
    int device = 0;
    int64_t N = 0;
    int32_t D = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    int num_cu = 256;
    // state
    double *X = nullptr, *lp = nullptr;
    uint8_t* acc = nullptr;
    uint32_t *acc_count = nullptr, *status = nullptr;
    char* bounce[2] = {nullptr, nullptr};      // pinned halves of the large-copy pipeline (big_copy_to_host)
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    char* xfer_host = nullptr;         // pinned bounce buffer for small split-phase transfers (a D2H copy into pageable
    size_t xfer_bytes = 0;             // memory costs ~2x the latency of one into pinned memory + a host memcpy)
    uint32_t* status_host = nullptr;   // `status` lives in mapped pinned host memory: kernels only touch it on errors
                                       // (system-scope atomicOr), the host reads it without a device copy
    int32_t* iota = nullptr;
    // device-side snapshots of (coords, log_prob): a state handed out by an earlier call, kept in HBM (emx_snapshot_*)
    static constexpr int NSNAPSHOT = 8;
    double* snap[NSNAPSHOT] = {};
    // target
    emx_device_log_prob_fn cb_fn = nullptr;   // EMX_TARGET_DEVICE_CALLBACK: the caller's batched log-prob, run on device buffers
    void* cb_user = nullptr;
    int target = EMX_TARGET_HOST;
    double *tp0 = nullptr, *tp1 = nullptr;
    double tscale = 1.0;
    int Dp = 0;
    // moves
    std::vector<emx_move_desc> moves;
    std::vector<double> cdf;
    // Gaussian Metropolis move: per-move coordinate scales, the displacement rows, exact-mode staging
    std::vector<double*> mscale;          // device (D) or nullptr, one per move
    double* disp = nullptr;               // (N, D)
    double* noise_host = nullptr;         // pinned (N, D)
    hipEvent_t noise_ev = nullptr;
    bool noise_busy = false;
    unsigned long long* dbg = nullptr;    // phase timestamps of the last half-step launch (tuning key "phase_clock")
    int64_t dbg_blocks = 0;
    // exact-mode small runs: host plans of many steps per launch, double buffered
    struct BulkPlans {
        char* host = nullptr;       // pinned
        char* dev = nullptr;
        size_t bytes = 0;
        hipEvent_t done = nullptr;  // the launch that read this buffer has finished
        bool busy = false;
    } bulk[2];
    int bulk_pos = 0;
    int64_t tune_small = 1;               // small ensembles: whole runs in one workgroup (k_small_run); 0: general path only
    int64_t tune_gauss_materialize = 0;   // native mode: write the displacement rows to HBM (k_gauss_disp) instead of
                                          // generating them inside the half-step kernel (verification / tests)
    double gfac = 1.0;                    // step-size factor of the Gaussian step begun
    // rng
    int rng_mode = EMX_RNG_PHILOX;
    MT19937Legacy mt;
    uint64_t ph_seed = 0, ph_step = 0;
    // plan ring (device) + pinned host staging
    struct PlanSlot {
        int32_t *order = nullptr, *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
        double *s0 = nullptr, *uacc = nullptr, *logu = nullptr, *fac = nullptr;
        char* host = nullptr;  // pinned: [order|p0|p1|p2](int32 N each) [s0|uacc](double N each)
        hipEvent_t consumed = nullptr;
        hipEvent_t consumed_ref = nullptr; // the event behind the kernels that last read this slot (its own, or a persistent launch's)
        hipEvent_t uploaded = nullptr;     // pipeline uploads: the copy is done (upload stream)
        hipEvent_t uploaded_ref = nullptr; // the event that says this slot's latest upload is done ...
        unsigned last_seq = 0;             // the persistent launch that last read this slot's device copy (0: none since the pipeline started)
        int64_t fetch_step = -1;           // ... or (>= 0) the step whose plan k_plan_fetch takes from it: done when *pipe_done > fetch_step
        bool busy = false, host_written = false;
        int move_idx = 0;                  // exact-mode pipeline: the move of the step whose plan the slot holds
        PipeStepInfo pinfo;                // ... and how it was handed over (raw / regen: pipe_fetch_deferred finishes it on the device)
    } ring[PLAN_RING + MTDEV_SLOTS];
    // exact-mode plan pipeline (emx_mtpipe.hpp): alive only inside emx_run
    MtPlanPipeline* pipe = nullptr;
    int64_t pipe_taken = 0;              // steps whose plan emx_step_begin has taken
    int pipe_ring0 = 0;                  // ring slot of pipeline step 0
    int pipe_nsinks = PIPE_SINKS;        // slots the running pipeline cycles through
    std::deque<int64_t> pipe_uploads;    // steps whose upload is enqueued and not yet known to be complete
    bool pipe_defer = false;             // run_persist (exact mode): pipe_take only hands the slot out; ONE fetch kernel uploads the launch's plans
    std::vector<std::pair<int64_t, int>> pipe_deferred;    // (step, slot) taken but not yet fetched
    hipEvent_t pipe_batch_ev[8] = {};    // [0,4): fetches done (upload stream); [4,8): launches done (consumer stream)
    int pipe_batch_n = 0, pipe_cons_n = 0;
    unsigned long long* pipe_done = nullptr;    // pinned: steps of the running pipeline that k_plan_fetch has read out of their staging buffers
    unsigned* pipe_arrived = nullptr;           // device: k_plan_fetch's workgroup counter
    hipStream_t up_stream = nullptr;     // plan uploads overlap the previous step's kernels
    int64_t tune_mt_device_finish = 1;   // 1: stretch steps of the host pipeline are finished on the device (k_plan_raw); 0: by the finisher threads
    int64_t tune_persist_exact_mix = 1;  // 0: exact mode takes the persistent kernels with ONE move only (round 4)
    int64_t pipe_raw_steps = 0;          // steps taken that way (emx_pipe_stage_times)
    int64_t tune_mt_pipeline = -1;       // -1: on, finisher threads chosen from the core count; 0: off; k > 0: k finishers
    // exact-mode plans made on the device (emx_mtdev.hpp): one StretchMove, >= 8192 walkers, one replica
    MtDevProducer* mtdev = nullptr;
    int64_t mtdev_taken = 0;             // steps whose plan emx_step_begin has taken from it
    int64_t tune_persist_local = 1;      // 0: never the one-XCD form of the persistent kernel
    int64_t tune_persist_exact_max = 32768;   // exact mode: largest ensemble that takes the device-wide persistent kernel
    int64_t tune_persist_exact_regen_max = 131072;     // ... when its stretch steps are regen steps (regen_ctx_ok)
    int64_t tune_persist_exact_steps = 16;    // exact mode: steps per persistent launch (<= 16)
    int64_t tune_fetch_avoid = 1;             // k_plan_fetch keeps off the XCD a one-XCD persistent launch lives on
    int64_t tune_fetch_blocks = 64;           // ... and beside a device-wide launch runs as this many workgroups (0: one per piece of 256 entries)
    bool pipe_fetch_local = false;            // the launch being captured is a one-XCD one (run_persist -> pipe_fetch_deferred)
    int64_t tune_fetch_delay_us = 0;     // tests: k_plan_fetch idles this long before it reads
    int64_t tune_persist_span = 1;       // 0: a persistent launch ends with the batch of Philox plans it started in
    int64_t tune_persist_mix = 1;        // 0: a mixture's steps never share a launch (one run of one move per launch)
    int persist_mix_fits = -1;           // the mixed instantiation's grid is co-resident (asked once)
    int64_t tune_persist_slab = 1;       // 0: dense targets of padded ndim 80 ... 128 never take the persistent slab kernel (emx_pslab.hip)
    int64_t persist_slab_launches = 0;
    int64_t tune_persist_rows_late = 1;      // launches that store chain rows take k_persist's ROWS_LATE instantiation (stretch, even ndim <= 64): 0 never, 1 where measured to pay, 2 always
    int64_t tune_persist_stagger = -1;       // -1: 516 (the waves of SIMDs 2 and 3 wait 256 clocks) for device-wide stretch launches that store no chain rows, 528 for DE + snooker mixtures, else none
    int64_t tune_persist_max_halfsteps = PERSIST_MAX_ITERS;      // half-steps a persistent launch may hold (<= PERSIST_MAX_ITERS = 40)
    int64_t tune_mt_device_min_regen = 786432;      // the device producer's first ensemble size where the host pipeline's stretch steps are regen steps (mtdev_eligible)
    int64_t tune_mt_regen_min = 16384;   // exact mode, host pipeline with device finish: from this many walkers on a stretch step's fixed-length draws are made again
                                         // on the device from the generator's state (k_plan_regen) instead of crossing PCIe; 0: never
    int64_t pipe_regen_steps = 0;
    int64_t tune_persist_odd = 1;        // 0: dense targets of odd ndim never take k_persist (emx_podd.hip)
    int64_t tune_persist_slab_skew = 1;  // k_persist_slab: the second wave of a SIMD starts its row loads when its sibling's have arrived (0: at once; 2 ... 4: earlier)
    int64_t tune_persist_slab_local_max = 4096;      // largest ensemble that takes the one-XCD form of k_persist_slab (beyond: the device-wide form; measured, profiles/r06/pslab.txt)
    int64_t call_steps = 1;              // steps of the emx_run call being served (1: emx_step_begin on its own)
    int64_t tune_persist_exact = 1;      // 0: exact (MT19937) mode never takes the persistent kernels
    int64_t tune_persist_valu = 1;       // 0: never the persistent kernel of the element-wise targets (emx_pvalu.hip)
    int64_t tune_persist_local_max = 8192;    // largest ensemble that takes it
    int64_t tune_slab = 1;               // 0: never the slab form of the fused dense half-step (emx_slab.hip)
    int64_t tune_slab_skew = 1;          // 1: the second wave of every SIMD starts its first tile's row loads when its sibling's rows have arrived
    int64_t tune_mt_device = 1;          // 0: never (the host pipeline / the inline producer instead); 1: from tune_mt_device_min walkers on; 2: from 8192 on
    int64_t tune_mt_device_min = 147456; // (measured: the host pipeline is faster up to 131 072 walkers since round 5 -- 90-102 against 106 us/step there, 214
                                         //  against 175 at 262 144 -- profiles/r05/mtdev_sizes_r05.txt; round 4: 131 072, profiles/r04/mtdev_sizes.txt)
    int64_t tune_mt_lookahead = 2;       // batches the device producer is asked to run ahead of the consumer (0 .. 2)
    int64_t tune_mt_tok_wshift = 11, tune_mt_tok_tail = 2048;      // the device tokenizer's window rule (emx_mtdev_kernels.hpp)
    int64_t mtdev_steps_total = 0, mtdev_starts = 0;
    bool mtdev_defer_release = false;
    int64_t mtdev_release_pending = -1;
    MtDevStats mtdev_stats_last;
    int ring_pos = 0;
    struct Prepared {   // native plans already evaluated on the device, in step order
        int move, S, slot;
        bool lean;          // only the columns the fused half-step reads were written
        int gcol;           // Gaussian sequential mode: the coordinate this step moves
        uint64_t step;
        NativeArgs nat;
        double cursor_before;   // Gaussian sequential mode: the move's cursor before this step was planned
    };
    std::deque<Prepared> prepared;
    int64_t prep_hint = 1;   // upcoming steps the caller will take (emx_run sets it): batch size of the native prep
    std::vector<uint8_t> labels_scratch;
    // current step
    struct Cur {
        bool active = false;
        int move = 0, S = 2, slot = 0;
        bool store = false;
        bool native = false;
        bool lean = false;     // native plan without the columns nothing on the fused path reads (emx_plan_get completes it)
        bool devplan = false;  // exact-mode plan written by the device producer: there is no host copy of it
        int gcol = 0;
        std::vector<int32_t> off;
        NativeArgs nat{};
    } cur;
    // chain
    double *chain = nullptr, *chain_lp = nullptr;
    int64_t cap = 0, stored = 0, proposals = 0;
    // split-phase buffers
    double *qout = nullptr, *fout = nullptr, *newlp = nullptr;
    double* tp1_full = nullptr;        // dense target, ndim <= 112: the full image the wide-target kernels read (tuning "dense_wide")
    double *evalX = nullptr, *evallp = nullptr;
    // sharding
    int rank = 0, world = 1;
    double *sendbuf = nullptr, *gathered = nullptr;
    int64_t sendbuf_rows = 0, gathered_rows = 0;
    bool own_shard_bufs = false;
    void* comm = nullptr;   // ncclComm_t when the exchange is driven from here (emx_comm_init)
    int64_t send_doubles = 0, recv_doubles = 0;     // capacity of sendbuf / gathered
    // pull exchange (walker-block ownership)
    int exchange = EMX_EXCHANGE_ALLGATHER;
    PlanSlot cplan;                   // compact plan: the slots whose walker this rank owns
    int64_t cplan_rows = 0;
    int32_t* pull_counts = nullptr;   // [2][1 + world]: counters of the current / next half-step (k_pull_scatter re-arms them)
    int pull_parity = 0;
    int64_t pull_cap_armed = 0;       // capacity (records per pair) the send records' NaN indices were laid out for
    double* pull_armed_buf = nullptr;
    int64_t pull_cap_cur = 0;         // records per pair of the prepared half-step
    int pull_split = -1;
    // direct exchange: peers' coordinate arrays and barrier flags mapped into this process / device
    double* peerX[EMX_MAX_PEERS] = {};
    unsigned long long* peer_flags[EMX_MAX_PEERS] = {};
    bool peer_ipc_x[EMX_MAX_PEERS] = {}, peer_ipc_f[EMX_MAX_PEERS] = {};   // opened with hipIpcOpenMemHandle (to be closed)
    unsigned long long* my_flags = nullptr;   // [EMX_MAX_PEERS], fine-grained device memory when available
    bool peers_ready = false;
    PeerTable* peer_table = nullptr;          // device copy of (peerX, block starts) for the half-step kernel
    unsigned long long direct_epoch = 0;
    int32_t* direct_counts = nullptr;         // [64]: owned slots per split of the step begun
    bool direct_planned = false;              // k_own_plan has run for the step begun
    int64_t tune_direct_timeout_ms = 5000;
    // replay exchange: decisions travel, accepted updates are recomputed on every replica
    int32_t* replay_counts = nullptr;         // [2]: accepted foreign slots of the current / next half-step (k_replay_compact re-arms)
    int replay_parity = 0;
    int replay_split = -1;                    // emx_replay_begin ran for this split
    int64_t replay_recv_off = 0;              // doubles into `gathered`: the receive buffer of the half-step being finished
    int64_t replay_buf = 0;                   // doubles per receive buffer (there are two: the device-side exchange alternates)
    int replay_push_parity = 0;
    double* launch_declp = nullptr;           // set around a launch_split call: the launch writes its decisions here
    bool launch_push = false;                 // ... and (device-side exchange) into every peer's receive buffer at replay_recv_off
    bool replay_pushed = false;               // the own pass of this half-step did push (a fused launch): no push kernel needed
    int64_t launch_push_off = 0;
    bool eval_check_bad = false;              // MOVE_EVAL over proposals (log-prob exchange): non-finite rows get -inf, as in the fused path
    bool direct_dead = false;                 // a barrier timed out (seen by emx_status): no half-step until the peers are re-attached
    bool direct_first_barrier = false;        // the next barrier is the first of an emx_run call: the ranks may enter seconds apart
    // hipGraph replay of the native NATIVE_BATCH_MAX-step block (single move, thin_by 1, one rank)
    struct GraphSlot {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool valid = false;
        int64_t spw = -1, wpb = -1, bpc = -1;
        int target = -1;
    } gslot[2];                      // [store]
    StepDesc* d_desc = nullptr;      // NATIVE_BATCH_MAX descriptors
    GraphCounters* d_ctr = nullptr;
    GraphCounters* h_ctr = nullptr;  // pinned staging for the counters
    // host run-ahead throttle: the launch loop stays at most 2 windows ahead of the GPU, polling an event
    // instead of blocking inside the runtime when its queues fill up (which was seen to stall for tens of ms)
    hipEvent_t thr_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t tune_throttle = 64;      // steps per window, 0: off
    bool graph_disabled = false;
    bool graph_warm = false;         // one ordinary step has run (function attributes set, kernels loaded)
    int64_t tune_graph = 0;          // opt-in: on MI355X the replay is ~5 % slower than back-to-back launches unless the host is the bottleneck
    // tuning
    // persistent half-steps (tuning "persist", default on): k_persist runs a batch of native steps in one launch
    int64_t tune_persist_gauss_wpb = 0;       // waves per workgroup of k_persist_gauss (0: four)
    int64_t tune_persist_test_skew = 0;      // tests only: added to the barrier count the next launches wait for
    int64_t tune_persist = 1, tune_persist_timeout_ms = 2000, tune_persist_min_walkers = 512;
    int persist_wpb = 8;
    int64_t persist_launches = 0, persist_halfsteps = 0;
    struct PersistCapture {
        HalfStepArgs a;
        dim3 grid, block;
        size_t lds;
        int move, dpb, lean;
        bool dense, got;
    };
    PersistCapture* persist_cap = nullptr;      // launch_split fills this instead of launching
    unsigned* persist_bar = nullptr;
    unsigned* persist_started = nullptr; // pinned: number of the latest persistent launch known to have STARTED (PersistArgs::started_host)
    unsigned persist_lepoch = 0;         // barriers passed on the one-XCD form's flags
    unsigned persist_hepoch = 0;         // handshakes of the one-XCD form counted so far
    int64_t persist_local_launches = 0;
    unsigned* persist_ver = nullptr;  // (N) stamp of the half-step that last moved the walker
    unsigned persist_epoch = 0;
    unsigned persist_grid = 0;        // workgroups of the launches the barrier counters have counted so far
    // what a persistent launch did, kept until the stream is known to have run it: a launch whose grid could not become co-resident
    // leaves without a store (persist_handshake), and persist_recover takes its steps again on the per-half-step path
    struct PersistLog {
        unsigned seq;
        uint64_t ph_step;
        int64_t i0, steps, stored0, proposals0;
        int32_t thin_by, store;
        int64_t pipe_step0 = -1;          // exact mode: steps the host pipeline had handed out before this launch's first ...
        const void* pipe_id = nullptr;    // ... and which pipeline that was
    };
    std::deque<PersistLog> plog;
    unsigned persist_seq = 0;
    int64_t persist_recovered = 0;    // launches redone by persist_recover (emx_persist_info)
    int8_t persist_fits[4] = {-1, -1, -1, -1};      // [move kind]: the grid of that move's launches fits the device (occupancy query), -1: not asked yet
    int64_t tune_replay_two_pass = 0;         // 1: the replay exchange always compacts, then replays (tests of that form)
    int64_t tune_full_plan = 0;      // 1: native plans always carry every column
    int64_t tune_spw = 0, tune_bpc = 2, tune_wpb = 0, tune_ablate = 0, tune_dense_wide = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> prof;
    int prof_max = 0, prof_n = 0;
    std::string err;
};

static void graph_invalidate(emx_ctx* c);
static void apply_env_tuning(emx_ctx* c);
static int pipe_stop(emx_ctx* c);       // (rc != 0: the device producer of exact-mode plans failed -- see mtdev_stop)
#define PIPE_STOP(ctx)                        \
    do {                                      \
        const int _rps = pipe_stop(ctx);      \
        if (_rps) return _rps;                \
    } while (0)
static void direct_detach(emx_ctx* c);
static int direct_ensure(emx_ctx* c);
static void exchange_free(emx_ctx* c);
static int replay_ensure(emx_ctx* c);
static int flags_ensure(emx_ctx* c);
static int persist_settle(emx_ctx* c);
static hipError_t wait_stream(hipStream_t s);

// forget native plans evaluated ahead of time; the Gaussian sequential cursors they advanced go back
static void drop_prepared(emx_ctx* c) {
    for (auto it = c->prepared.rbegin(); it != c->prepared.rend(); ++it)
        if (it->move < (int)c->moves.size() && c->moves[it->move].kind == EMX_MOVE_GAUSS) c->moves[it->move].gammas = it->cursor_before;
    c->prepared.clear();
}

#define FAIL(ctx, code, ...)                         \
    do {                                             \
        char _b[512];                                \
        snprintf(_b, sizeof(_b), __VA_ARGS__);       \
        if (ctx) (ctx)->err = _b; else g_err = _b;   \
        return (code);                               \
    } while (0)

#define HIPOK(ctx, expr)                                                                          \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) FAIL(ctx, -2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define NEED(ctx, cond, ...)               \
    do {                                   \
        if (!(cond)) FAIL(ctx, -1, __VA_ARGS__); \
    } while (0)

// ------------------------------------------------------------------------------------------
// kernel dispatch
// ------------------------------------------------------------------------------------------
namespace {

// the occasional features a LEAN instantiation compiles out
int prefetch_depth_host(int G, int V, int CH, int move, bool dense);

// -> 0: the full kernel; 1: LEAN; 2 / 3 / 4: LEAN for the block-ownership exchanges -- 2: peer table (+ device-side slot count),
// 3: device-side slot count alone, 4: slot count + decision output.  (One instantiation for all three spilled 39 scalar registers at
// the headline shape; split: 13 / 0 / 6.  Shapes without a 3 take 2, which is a superset of it.)
inline int lean_kind(const HalfStepArgs& a, int G, int V, int CH, int move, bool dense) {
    const int WPW = 64 / G, PF = prefetch_depth_host(G, V, CH, move, dense);
    const int spw = (dense && PF * WPW < 16) ? 16 : PF * WPW;
    const bool common = !a.ablate && !a.desc && !a.sendbuf && !a.disp && !a.skew_sleep && (EMX_OPT_STAMPS || !a.dbg) && (a.target != TGT_NONE || !dense) &&      // (propose-only passes run element-wise instantiations)
                        (!EMX_LEAN_FOLD_D || a.D == G * V * CH) && a.spw == spw && a.t_lo == 0;      // ndim is folded only with EMX_LEAN_FOLD_D
    if (!common) return 0;
    if (a.npeer) return a.declp ? 0 : 2;         // direct exchange: the peer table
    if (a.declp) return 4;                       // replay exchange, own pass: decision output (+ push)
    return a.t_hi_dev ? 3 : 1;                   // pull exchange / replay launches: the device-side slot count alone
}

template <int MOVE>
hipError_t launch_valu(const Shape& sh, dim3 grid, dim3 block, hipStream_t st, const HalfStepArgs& a) {
    if constexpr (MOVE == MOVE_STRETCH) {          // C3 (D = 32) and C5 (D = 1024)
        if (sh.G == 8 && sh.V == 2 && sh.CH == 2) {
            const int lk = lean_kind(a, 8, 2, 2, MOVE, false);
            if (lk == 1) return launch_one<8, 2, 2, MOVE, 0, 1>(grid, block, 0, st, a);
            if (lk == 2 || lk == 3) return launch_one<8, 2, 2, MOVE, 0, 2>(grid, block, 0, st, a);
            if (lk == 4) return launch_one<8, 2, 2, MOVE, 0, 4>(grid, block, 0, st, a);
        }
        if (sh.G == 8 && sh.V == 2 && sh.CH == 4) {       // ndim 64 (the replay of C2's accepted updates on the other replicas)
            const int lk = lean_kind(a, 8, 2, 4, MOVE, false);
            if (lk == 2 || lk == 3) return launch_one<8, 2, 4, MOVE, 0, 2>(grid, block, 0, st, a);
        }
        if (sh.G == 64 && sh.V == 2 && sh.CH == 8) {
            const int lk = lean_kind(a, 64, 2, 8, MOVE, false);
            if (lk == 1) return launch_one<64, 2, 8, MOVE, 0, 1>(grid, block, 0, st, a);
            if (lk == 2 || lk == 3) return launch_one<64, 2, 8, MOVE, 0, 2>(grid, block, 0, st, a);
            if (lk == 4) return launch_one<64, 2, 8, MOVE, 0, 4>(grid, block, 0, st, a);
        }
    }
    if constexpr (MOVE != MOVE_EVAL) {   // every element-wise shape has a LEAN instantiation
#define EMX_LEAN_CASE(g, v, c)                                                                         \
    if (sh.G == g && sh.V == v && sh.CH == c && lean_kind(a, g, v, c, MOVE, false) == 1)                \
        return launch_one<g, v, c, MOVE, 0, 1>(grid, block, 0, st, a);
        EMX_LEAN_CASE(4, 1, 1) EMX_LEAN_CASE(8, 1, 1) EMX_LEAN_CASE(8, 1, 2) EMX_LEAN_CASE(8, 1, 4) EMX_LEAN_CASE(16, 1, 4)
        EMX_LEAN_CASE(32, 1, 4) EMX_LEAN_CASE(64, 1, 4) EMX_LEAN_CASE(64, 1, 8) EMX_LEAN_CASE(4, 2, 1) EMX_LEAN_CASE(8, 2, 1)
        EMX_LEAN_CASE(8, 2, 4) EMX_LEAN_CASE(16, 2, 4) EMX_LEAN_CASE(32, 2, 4) EMX_LEAN_CASE(64, 2, 4)
        if constexpr (MOVE == MOVE_STRETCH || MOVE == MOVE_GAUSS) {
            EMX_LEAN_CASE(64, 1, 16) EMX_LEAN_CASE(64, 2, 16)
        }
        if constexpr (MOVE != MOVE_STRETCH) {         // the stretch move's own are above (with the block-ownership variant)
            EMX_LEAN_CASE(8, 2, 2) EMX_LEAN_CASE(64, 2, 8)
        }
#undef EMX_LEAN_CASE
    }
#define EMX_CASE(g, v, c) \
    if (sh.G == g && sh.V == v && sh.CH == c) return launch_one<g, v, c, MOVE, 0>(grid, block, 0, st, a);
    EMX_CASE(4, 1, 1) EMX_CASE(8, 1, 1) EMX_CASE(8, 1, 2) EMX_CASE(8, 1, 4) EMX_CASE(16, 1, 4) EMX_CASE(32, 1, 4)
    EMX_CASE(64, 1, 4) EMX_CASE(64, 1, 8)
    EMX_CASE(4, 2, 1) EMX_CASE(8, 2, 1) EMX_CASE(8, 2, 2) EMX_CASE(8, 2, 4) EMX_CASE(16, 2, 4) EMX_CASE(32, 2, 4)
    EMX_CASE(64, 2, 4) EMX_CASE(64, 2, 8)
    if constexpr (MOVE == MOVE_STRETCH || MOVE == MOVE_EVAL || MOVE == MOVE_GAUSS) {
        EMX_CASE(64, 1, 16) EMX_CASE(64, 2, 16)
    }
#undef EMX_CASE
    return hipErrorInvalidValue;
}

// dense: the row layout follows from (Dp, V): cols = Dp / V
constexpr int dense_g(int dpb, int v) { return shape_g(dpb * 16 / v); }
constexpr int dense_ch(int dpb, int v) { return shape_ch(dpb * 16 / v); }

template <int MOVE>
hipError_t launch_dense(int dpb, int V, dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a) {
    if constexpr (MOVE != MOVE_EVAL) {             // C2 / C4 (D = 64)
        if (dpb == 4 && V == 2) {
            const int lk = lean_kind(a, dense_g(4, 2), 2, dense_ch(4, 2), MOVE, true);
            if constexpr (MOVE == MOVE_STRETCH) {
                if (lk) return launch_hot_stretch_dense64(lk, grid, block, lds, st, a);       // emx_hot.hip (its own scheduler strategy)
            } else {
                if (lk == 1) return launch_one<dense_g(4, 2), 2, dense_ch(4, 2), MOVE, 4, 1>(grid, block, lds, st, a);
            }
        }
    }
    if constexpr (MOVE != MOVE_EVAL) {
        // every dense shape has a LEAN instantiation: the general kernel's scalar-register spills cost a dense half-step
        // 14 ... 31 % (65 536 x 32: 20.3 -> 16.0 us/step; profiles/r02/lean_shapes.txt)
#define EMX_LEAN_CASE(b, v)                                                                          \
    if (dpb == b && V == v && lean_kind(a, dense_g(b, v), v, dense_ch(b, v), MOVE, true) == 1)       \
        return launch_one<dense_g(b, v), v, dense_ch(b, v), MOVE, b, 1>(grid, block, lds, st, a);
        EMX_LEAN_CASE(1, 1) EMX_LEAN_CASE(2, 1) EMX_LEAN_CASE(3, 1) EMX_LEAN_CASE(4, 1) EMX_LEAN_CASE(5, 1) EMX_LEAN_CASE(6, 1)
        EMX_LEAN_CASE(7, 1) EMX_LEAN_CASE(8, 1) EMX_LEAN_CASE(1, 2) EMX_LEAN_CASE(2, 2) EMX_LEAN_CASE(3, 2) EMX_LEAN_CASE(5, 2)
        EMX_LEAN_CASE(6, 2) EMX_LEAN_CASE(7, 2) EMX_LEAN_CASE(8, 2)
#undef EMX_LEAN_CASE
    }
#define EMX_CASE(b, v) \
    if (dpb == b && V == v) return launch_one<dense_g(b, v), v, dense_ch(b, v), MOVE, b>(grid, block, lds, st, a);
    EMX_CASE(1, 1) EMX_CASE(2, 1) EMX_CASE(3, 1) EMX_CASE(4, 1) EMX_CASE(5, 1) EMX_CASE(6, 1) EMX_CASE(7, 1) EMX_CASE(8, 1)
    EMX_CASE(1, 2) EMX_CASE(2, 2) EMX_CASE(3, 2) EMX_CASE(4, 2) EMX_CASE(5, 2) EMX_CASE(6, 2) EMX_CASE(7, 2) EMX_CASE(8, 2)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

hipError_t dispatch_halfstep(int move, bool dense, int dpb, const Shape& sh, dim3 grid, dim3 block, size_t lds,
                             hipStream_t st, const HalfStepArgs& a) {
    switch (move) {
        case MOVE_STRETCH:
            return dense ? launch_dense<MOVE_STRETCH>(dpb, sh.V, grid, block, lds, st, a)
                         : launch_valu<MOVE_STRETCH>(sh, grid, block, st, a);
        case MOVE_DE:
            return dense ? launch_dense<MOVE_DE>(dpb, sh.V, grid, block, lds, st, a) : launch_valu<MOVE_DE>(sh, grid, block, st, a);
        case MOVE_SNOOKER:
            return dense ? launch_dense<MOVE_SNOOKER>(dpb, sh.V, grid, block, lds, st, a)
                         : launch_valu<MOVE_SNOOKER>(sh, grid, block, st, a);
        case MOVE_GAUSS:
            return dense ? launch_dense<MOVE_GAUSS>(dpb, sh.V, grid, block, lds, st, a) : launch_valu<MOVE_GAUSS>(sh, grid, block, st, a);
        case MOVE_EVAL:
            return dense ? launch_dense<MOVE_EVAL>(dpb, sh.V, grid, block, lds, st, a) : launch_valu<MOVE_EVAL>(sh, grid, block, st, a);
    }
    return hipErrorInvalidValue;
}

// the stretch move's one-launch replay (k_replay_stretch), by row layout
hipError_t launch_replay_stretch(const Shape& sh, dim3 grid, hipStream_t st, const ReplayFusedArgs& a) {
#define EMX_CASE(g, v, c)                                                                              \
    if (sh.G == g && sh.V == v && sh.CH == c) {                                                        \
        hipLaunchKernelGGL((k_replay_stretch<g, v, c>), grid, dim3(256), 0, st, a);                    \
        return hipGetLastError();                                                                      \
    }
    EMX_CASE(4, 1, 1) EMX_CASE(8, 1, 1) EMX_CASE(8, 1, 2) EMX_CASE(8, 1, 4) EMX_CASE(16, 1, 4) EMX_CASE(32, 1, 4)
    EMX_CASE(64, 1, 4) EMX_CASE(64, 1, 8) EMX_CASE(64, 1, 16)
    EMX_CASE(4, 2, 1) EMX_CASE(8, 2, 1) EMX_CASE(8, 2, 2) EMX_CASE(8, 2, 4) EMX_CASE(16, 2, 4) EMX_CASE(32, 2, 4)
    EMX_CASE(64, 2, 4) EMX_CASE(64, 2, 8) EMX_CASE(64, 2, 16)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

size_t dense_lds_bytes(int Dp, int waves) {
    const int RT = Dp + 2;
    return ((size_t)dense_img_doubles(Dp) + Dp + (size_t)waves * (16 * RT + 32)) * sizeof(double);
}

int prefetch_depth_host(int G, int V, int CH, int move, bool dense) {
    const int WPW = 64 / G;
    const int nr = (move == MOVE_STRETCH || move == MOVE_GAUSS) ? 2 : move == MOVE_DE ? 3 : move == MOVE_SNOOKER ? 4 : 1;
    int pf = (move == MOVE_SNOOKER ? EMX_SNOOKER_BUDGET : 48) / (nr * CH * V);      // prefetch_depth in emx_kernels.hpp
    pf = pf < 1 ? 1 : (pf > 8 ? 8 : pf);
    int p2 = 1;
    while (p2 * 2 <= pf) p2 *= 2;
    if (dense && p2 > 16 / WPW) p2 = 16 / WPW;
    return p2 < G ? p2 : G;
}

// dense Gaussian targets whose Cholesky image does not fit LDS next to the MFMA tiles (or tuning "dense_wide" = 1, for tests)
// The fused kernel holds the packed triangular image of L plus one 16-row tile per wave in LDS: up to padded ndim 128
// (36 blocks = 72 KB + 4 waves x 16.6 KB; round 3 -- 112 with 8- and 4-wave groups before)
constexpr int DENSE_FUSED_MAX_DP = 128;
inline bool dense_is_wide(const emx_ctx* c) { return c->Dp > DENSE_FUSED_MAX_DP || c->tune_dense_wide; }

// launch one fused (or propose-only) half-step over the slots [t_lo, t_hi) of `split`
int launch_split(emx_ctx* c, int move, int target, int S, int split, int pos0, int ns, int t_lo, int t_hi,
                 const emx_move_desc* mv, const emx_ctx::PlanSlot* ps, const int32_t* order,
                 double* X, double* lp, double* chain, double* chain_lp, double* sendbuf,
                 const StepDesc* step_desc = nullptr, const int32_t* t_hi_dev = nullptr) {
    if (t_hi <= t_lo) return 0;
    const bool dense = target == EMX_TARGET_DENSE_GAUSS;
    const int D = c->D;
    const bool callback = target == EMX_TARGET_DEVICE_CALLBACK;
    if ((dense && dense_is_wide(c)) || callback) {
        // Three passes on the stream -- the reference's compute_log_prob between get_proposal and the accept loop
        // (red_blue.py:90-101): propose -> log-prob of the proposal block -> decision + commit.  The log-prob pass is
        //   * k_wide_lp (MFMA, L streamed through LDS; emx_wide.hip) for a precision matrix too wide for LDS, or
        //   * the caller's device callback (emx_set_target_callback): ensemble.py:486-487's "one call on (Ns, ndim)", the block
        //     and the result staying in HBM.
        if (step_desc) {
            c->err = "three-pass targets are not replayable from a step graph";
            return -1;
        }
        if (callback) {
            if (!c->cb_fn) {
                c->err = "device callback target without a callback (emx_set_target_callback)";
                return -1;
            }
            if (t_hi_dev) {
                c->err = "device callback target: the block-ownership exchanges (pull, direct) size the block on the device; use the "
                         "all-gather, log-prob or replay exchange";
                return -1;
            }
        }
        // the caller's function enqueues its work on the context stream; rows [t_lo, t_hi) of `rows` -> out[t_lo .. t_hi)
        auto call_back = [&](const double* rows, double* out) -> int {
            const int rcb = c->cb_fn(c->cb_user, rows + (size_t)t_lo * D, (int64_t)(t_hi - t_lo), D, out + t_lo, (void*)c->stream);
            if (rcb != 0) {
                char b[160];
                snprintf(b, sizeof(b), "the device log-prob callback failed (returned %d)", rcb);
                c->err = b;
                return -7;
            }
            return 0;
        };
        WideLpArgs w{};
        w.order = order ? order : (ps ? ps->order : nullptr);
        w.img = c->tp1_full ? c->tp1_full : c->tp1;
        w.status = c->status;
        w.t_hi_dev = t_hi_dev;
        w.D = D;
        w.Dp = c->Dp;
        w.pos0 = pos0;
        w.t_lo = t_lo;
        w.t_hi = t_hi;
        w.single_role = c->tune_dense_wide == 2;
        if (move == MOVE_EVAL && callback) {
            if (order && order != c->iota) {
                c->err = "device callback target: batched evaluation in plan order is not supported";
                return -1;
            }
            return call_back(X, lp);
        }
        if (move == MOVE_EVAL) {
            w.rows = X;
            w.out = lp;
            w.scatter = 1;
            w.check_bad = c->eval_check_bad ? 1 : 0;
            if (launch_wide_lp(w, t_hi - t_lo, c->num_cu, c->stream) != hipSuccess) {
                c->err = "wide dense target: log-prob kernel launch failed";
                return -2;
            }
            return 0;
        }
        if (!ps) {
            c->err = "wide dense target: half-step without a plan";
            return -1;
        }
        // per-launch profile events (emx_profile_enable) bracket the MFMA log-prob kernel here -- the dominant kernel of this
        // path -- not the propose pass
        const int prof_max = c->prof_max;
        const bool prof = prof_max > 0 && c->prof_n < prof_max;
        c->prof_max = 0;
        double* const declp = c->launch_declp;      // the commit kernel below writes the decisions, not the propose pass
        c->launch_declp = nullptr;
        int rc = launch_split(c, move, EMX_TARGET_HOST, S, split, pos0, ns, t_lo, t_hi, mv, ps, order, X, lp, nullptr, nullptr,
                              nullptr, nullptr, t_hi_dev);
        c->launch_declp = declp;
        c->prof_max = prof_max;
        if (rc) return rc;
        w.rows = c->qout;
        w.order = nullptr;
        w.out = c->newlp;
        w.scatter = 0;
        w.check_bad = 1;
        WideCommitArgs k{};
        k.X = X;
        k.lp = lp;
        k.acc = c->acc;
        k.acc_count = c->acc_count;
        k.chain = chain;
        k.chain_lp = chain_lp;
        k.sendbuf = sendbuf;
        k.qout = c->qout;
        k.fout = c->fout;
        k.newlp = c->newlp;
        k.order = order ? order : ps->order;
        k.logu = ps->logu;
        k.t_hi_dev = t_hi_dev;
        k.declp = c->launch_declp;
        k.D = D;
        k.pos0 = pos0;
        k.t_lo = t_lo;
        k.t_hi = t_hi;
        k.status = c->status;
        if (prof && hipEventRecord(c->prof[2 * c->prof_n], c->stream) != hipSuccess) return -2;
        hipError_t e_lp = hipSuccess;
        if (callback) {
            rc = call_back(c->qout, c->newlp);
            if (rc) return rc;
        } else {
            e_lp = launch_wide_lp(w, t_hi - t_lo, c->num_cu, c->stream);
        }
        if (prof) {
            if (hipEventRecord(c->prof[2 * c->prof_n + 1], c->stream) != hipSuccess) return -2;
            c->prof_n++;
        }
        if (e_lp != hipSuccess || launch_wide_commit(k, t_hi - t_lo, c->num_cu, c->stream) != hipSuccess) {
            c->err = "wide dense target: kernel launch failed";
            return -2;
        }
        return 0;
    }
    // a replay launch (TGT_REPLAY: element-wise kernel, no target) uses the row layout of the rank that took the decisions --
    // the dense layout when the context's target is the fused dense Gaussian -- so that the snooker move's group reductions
    // associate identically and the recomputed proposal has the owner's bits
    const bool dense_layout = dense || (target == TGT_REPLAY && c->target == EMX_TARGET_DENSE_GAUSS && !dense_is_wide(c));
    const Shape sh = pick_shape(D, dense_layout ? c->Dp : D);
    const int WPW = 64 / sh.G;
    const int nown = t_hi - t_lo;
    const int PF = prefetch_depth_host(sh.G, sh.V, sh.CH, move, dense);
    // slots per wave: one prefetch batch (PF passes) unless the grid would be tiny or huge
    int64_t spw = c->tune_spw > 0 ? c->tune_spw : (int64_t)PF * WPW;
    if (dense && spw < 16) spw = 16;
    if (spw > 64) spw = 64;
    spw = (spw / WPW) * WPW;
    if (spw < WPW) spw = WPW;
    int waves_per_block = 4;
    size_t lds = 0;
    if (dense) {
        // 8-wave workgroups halve the number of LDS image copies; below ~2048 tiles 4-wave groups spread
        // the few tiles over more CUs (tools/nsweep.py)
        const int64_t ntiles = (nown + spw - 1) / spw;
        // ... except for the Gaussian Metropolis move, whose single launch covers the whole ensemble (4 tiles per SIMD at
        // 65 536 walkers): two co-resident 4-wave groups per CU, two tiles per wave, overlap better than one 8-wave group
        // (24.0 vs 26.5 us/step, tools/wpb_sweep.py)
        waves_per_block = c->tune_wpb > 0 ? (int)c->tune_wpb : (ntiles >= 2048 && move != MOVE_GAUSS ? 8 : 4);
        if (c->persist_cap) waves_per_block = c->persist_wpb;            // k_persist: about one workgroup per CU (persist_shape)
        const bool pslab = c->persist_cap && c->Dp > 64;                 // k_persist_slab: a 32-column slab per wave instead of a whole tile
        while (!pslab && waves_per_block > 1 && dense_lds_bytes(c->Dp, waves_per_block) > 160 * 1024) waves_per_block >>= 1;
        lds = pslab ? slab_lds_bytes(c->Dp, waves_per_block) : dense_lds_bytes(c->Dp, waves_per_block);
        if (lds > 160 * 1024) {
            c->err = "dense Gaussian target: ndim too large for the LDS-resident precision matrix (max 128)";
            return -1;
        }
    }
    if (!dense && c->persist_cap) {          // k_persist_valu: a wave owns 16 slots of every split, persist_wpb waves a workgroup
        spw = 16;
        waves_per_block = c->persist_wpb;
    }
    const int64_t nbatch = (nown + spw - 1) / spw;
    int64_t nblocks = (nbatch + waves_per_block - 1) / waves_per_block;
    if (dense) nblocks = std::min<int64_t>(nblocks, (int64_t)c->num_cu * c->tune_bpc);
    HalfStepArgs a{};
    a.X = X;
    a.lp = lp;
    a.acc = c->acc;
    a.acc_count = c->acc_count;
    a.chain = chain;
    a.chain_lp = chain_lp;
    a.status = c->status;
    a.qout = c->qout;
    a.fout = c->fout;
    a.sendbuf = sendbuf;
    a.order = order ? order : (ps ? ps->order : nullptr);
    a.p0 = ps ? ps->p0 : nullptr;
    a.p1 = ps ? ps->p1 : nullptr;
    a.p2 = ps ? ps->p2 : nullptr;
    a.s0 = ps ? ps->s0 : nullptr;
    a.uacc = ps ? ps->uacc : nullptr;
    a.logu = ps ? ps->logu : nullptr;
    a.fac = ps ? ps->fac : nullptr;
    a.tp0 = c->tp0;
    a.tp1 = c->tp1;
    a.tscale = c->tscale;
    if (mv) {
        a.a = mv->a;
        a.sigma = mv->sigma;
        a.g0 = mv->g0;
        a.gammas = mv->gammas;
    }
    a.N = (int32_t)c->N;
    a.D = D;
    a.S = S;
    a.split = split;
    a.pos0 = pos0;
    a.ns = ns;
    a.t_lo = t_lo;
    a.t_hi = t_hi;
    a.spw = (int32_t)spw;
    a.target = target;
    a.Dp = dense ? c->Dp : 16;
    a.ablate = (int32_t)(c->tune_ablate & 0xff);          // (bits 8.. are the plan kernel's)
    a.desc = step_desc;
    a.chain_all = c->chain;
    a.chain_lp_all = c->chain_lp;
    a.t_hi_dev = t_hi_dev;
    a.declp = c->launch_declp;
    if (c->launch_declp && c->launch_push && c->peer_table) {      // fused own pass of the device-side replay exchange
        a.push_peers = c->peer_table;
        a.npush = c->world;
        a.push_off = c->launch_push_off;
        c->replay_pushed = true;
    }
    if (move == MOVE_GAUSS && mv) {
        const bool in_registers = c->cur.active && c->cur.native && !c->tune_gauss_materialize;
        a.disp = in_registers ? nullptr : c->disp;
        a.gscale = c->cur.move >= 0 ? c->mscale[c->cur.move] : nullptr;
        a.gsigma = mv->sigma;
        a.gfac = c->gfac;
        a.gseed = c->ph_seed;
        a.gstep = c->cur.nat.step;
    }
    if (c->exchange == EMX_EXCHANGE_DIRECT && c->peers_ready && c->world > 1 && move != MOVE_EVAL && X == c->X) {
        a.npeer = c->world;
        a.peers = c->peer_table;
    }
    {
        static const int skew_sleep = getenv("EMX_SKEW_SLEEP") ? atoi(getenv("EMX_SKEW_SLEEP")) : 0;     // kernel experiments only
        a.skew_sleep = skew_sleep;
    }
    a.dbg = (c->dbg && nblocks <= c->dbg_blocks && move != MOVE_EVAL) ? c->dbg : nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = c->prof_max > 0 && c->prof_n < c->prof_max && move != MOVE_EVAL;
    if (prof) {
        e0 = c->prof[2 * c->prof_n];
        e1 = c->prof[2 * c->prof_n + 1];
        if (hipEventRecord(e0, c->stream) != hipSuccess) return -2;
    }
    if (c->persist_cap) {        // run_persist collects the launches of a batch of steps
        auto& pc = *c->persist_cap;
        pc.a = a;
        pc.grid = dim3((unsigned)nblocks);
        pc.block = dim3(64 * waves_per_block);
        pc.lds = lds;
        pc.move = move;
        pc.dpb = c->Dp / 16;
        pc.dense = dense;
        pc.lean = lean_kind(a, sh.G, sh.V, sh.CH, move, dense);
        pc.got = nbatch == nblocks * waves_per_block && (nown % spw) == 0;     // every wave exactly one full tile
        return 0;
    }
    hipError_t e;
    // padded ndim 80 ... 128, even ndim, single replica: the slab form (emx_slab.hip) -- eight waves a CU instead of four
    // (measured, 65 536 walkers, stretch: padded 128 66.6 -> 50.8 us/step, 112 61.6 -> 44.3; at 96 and 80 the per-tile kernel already
    // has eight waves a CU and the two are level -- profiles/r04/slab_ab.txt; tuning "slab" = 2 takes it from padded 80 on)
    const bool slab = dense && c->tune_slab && c->Dp >= (c->tune_slab == 2 ? 80 : 112) && sh.G == 16 && sh.V == 2 && sh.CH == 4 &&
                      (move == MOVE_STRETCH || move == MOVE_DE) && lean_kind(a, 16, 2, 4, move, true) == 1;
    if (slab) {
        const int wpb = 8;
        const int64_t ntile = (nown + 15) / 16;
        const int64_t nb = std::min<int64_t>((ntile + wpb - 1) / wpb, (int64_t)c->num_cu * c->tune_bpc);
        HalfStepArgs as = a;
        as.ablate = c->tune_slab_skew ? 256 | (int32_t)((c->tune_slab_skew - 1) << 9) : 0;     // (lean launches carry no ablation mask: bit 8 = skewed start, bits 9-10 = when the sibling starts)
        e = launch_slab_dense(c->Dp / 16, move, dim3((unsigned)nb), dim3(64 * wpb), slab_lds_bytes(c->Dp, wpb), c->stream, as);
    } else {
        e = dispatch_halfstep(move, dense, c->Dp / 16, sh, dim3((unsigned)nblocks), dim3(64 * waves_per_block), lds, c->stream, a);
    }
    if (e != hipSuccess) {
        char b[256];
        snprintf(b, sizeof(b), "half-step launch failed (G=%d V=%d CH=%d move=%d dense=%d ndim=%d): %s", sh.G, sh.V, sh.CH,
                 move, (int)dense, D, e == hipErrorInvalidValue ? "unsupported ndim for this build" : hipGetErrorString(e));
        c->err = b;
        return -2;
    }
    if (prof) {
        if (hipEventRecord(e1, c->stream) != hipSuccess) return -2;
        c->prof_n++;
    }
    return 0;
}

void shard_range(int64_t ns, int rank, int world, int64_t& lo, int64_t& hi) {
    lo = ns * rank / world;
    hi = ns * (rank + 1) / world;
}

}  // namespace

// ---- glue for the other translation units (emx_internal.hpp) -------------------------------------------------------
int emx_internal_chain_view(emx_ctx* c, EmxChainView* v) {
    if (!c) {
        g_err = "null context";
        return -1;
    }
    v->chain = c->chain;
    v->chain_lp = c->chain_lp;
    v->N = c->N;
    v->D = c->D;
    v->stored = c->stored;
    v->stream = c->stream;
    v->device = c->device;
    return 0;
}

int emx_internal_state_view(emx_ctx* c, const double** X, int64_t* N, int32_t* D, int* device) {
    if (!c) {
        g_err = "null context";
        return -1;
    }
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, wait_stream(c->stream));
    *X = c->X;
    *N = c->N;
    *D = c->D;
    *device = c->device;
    return 0;
}

int emx_internal_fail(emx_ctx* c, int code, const char* msg) {
    if (c)
        c->err = msg;
    else
        g_err = msg;
    return code;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
#pragma GCC visibility push(default)
extern "C" {

const char* emx_version(void) { return "emx 0.1 (gfx950)"; }

const char* emx_last_error(const emx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int emx_device_count(int32_t* n) {
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) {
        *n = 0;
        g_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return -2;
    }
    *n = k;
    return 0;
}

int emx_create(int32_t device, int64_t nwalkers, int32_t ndim, emx_ctx** out) {
    *out = nullptr;
    emx_ctx* nullctx = nullptr;
    NEED(nullctx, nwalkers >= 2 && nwalkers < (1ll << 31) && ndim >= 1, "invalid ensemble shape (%lld, %d)",
         (long long)nwalkers, ndim);
    int n = 0;
    if (emx_device_count(&n) != 0 || n <= 0) FAIL(nullctx, -3, "no HIP device available (there is no CPU fallback)");
    NEED(nullctx, device >= 0 && device < n, "device %d out of range (%d devices)", device, n);
    HIPOK(nullctx, hipSetDevice(device));
    emx_ctx* c = new emx_ctx();
    c->device = device;
    c->N = nwalkers;
    c->D = ndim;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    const size_t N = (size_t)nwalkers, D = (size_t)ndim;
#define ALLOC(ptr, bytes)                                         \
    do {                                                          \
        hipError_t _e = hipMalloc((void**)&(ptr), (bytes));       \
        if (_e != hipSuccess) {                                   \
            g_err = std::string("hipMalloc: ") + hipGetErrorString(_e); \
            emx_destroy(c);                                       \
            return -2;                                            \
        }                                                         \
    } while (0)
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        g_err = "hipStreamCreate failed";
        delete c;
        return -2;
    }
    c->stream = c->own_stream;
    ALLOC(c->X, N * D * 8);
    ALLOC(c->lp, N * 8);
    ALLOC(c->acc, N);
    ALLOC(c->acc_count, N * 4);
    if (hipHostMalloc((void**)&c->status_host, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->status, c->status_host, 0) != hipSuccess) {
        g_err = "status word allocation failed";
        emx_destroy(c);
        return -2;
    }
    for (int k = 0; k < 16; ++k) c->status_host[k] = 0u;
    ALLOC(c->iota, N * 4);
    ALLOC(c->qout, N * D * 8);
    ALLOC(c->fout, N * 8);
    ALLOC(c->newlp, N * 8);
    ALLOC(c->evalX, N * D * 8);
    ALLOC(c->evallp, N * 8);
    for (int r = 0; r < PLAN_RING; ++r) {          // (the device producer's slots behind them: mtdev_start)
        auto& s = c->ring[r];
        // one block per slot, laid out like the pinned staging buffer ([order|p0|p1|p2] int32, [s0|uacc] f64) so that a
        // host-made plan goes up in ONE copy; the device-computed logs follow
        char* blk = nullptr;
        ALLOC(blk, N * 48);
        s.order = (int32_t*)blk;
        s.p0 = s.order + N;
        s.s0 = (double*)(blk + N * 8);
        s.uacc = s.s0 + N;
        s.p1 = (int32_t*)(blk + N * 24);
        s.p2 = s.p1 + N;
        s.logu = (double*)(blk + N * 32);
        s.fac = s.logu + N;
        if (hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming) != hipSuccess) {
            g_err = "plan event creation failed";
            emx_destroy(c);
            return -2;
        }
    }
#undef ALLOC
    hipMemsetAsync(c->acc, 0, N, c->stream);
    hipMemsetAsync(c->acc_count, 0, N * 4, c->stream);
    hipMemsetAsync(c->lp, 0, N * 8, c->stream);
    {
        std::vector<int32_t> io(N);
        for (size_t i = 0; i < N; ++i) io[i] = (int32_t)i;
        hipMemcpyAsync(c->iota, io.data(), N * 4, hipMemcpyHostToDevice, c->stream);
        hipStreamSynchronize(c->stream);
    }
    hipEventCreate(&c->ev0);
    hipEventCreate(&c->ev1);
    emx_move_desc d{};
    d.kind = EMX_MOVE_STRETCH;
    d.nsplits = 2;
    d.randomize_split = 1;
    d.a = 2.0;
    d.sigma = 1e-5;
    d.g0 = 2.38 / std::sqrt(2.0 * ndim);
    d.gammas = 1.7;
    c->moves.assign(1, d);
    c->cdf.assign(1, 1.0);
    apply_env_tuning(c);
    *out = c;
    return 0;
}

// EMX_TUNE="key=value,key=value": tuning keys applied to every context at creation (A/B measurements through unmodified callers)
static void apply_env_tuning(emx_ctx* c) {
    const char* e = getenv("EMX_TUNE");
    if (!e || !*e) return;
    std::string s(e);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find(',', pos);
        if (end == std::string::npos) end = s.size();
        const std::string kv = s.substr(pos, end - pos);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos) emx_set_tuning(c, kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1));
        pos = end + 1;
    }
}

int emx_destroy(emx_ctx* c) {
    if (!c) return 0;
    pipe_stop(c);                 // its threads write into the pinned staging buffers freed below
    c->plog.clear();
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm), c->comm = nullptr;
    if (c->status_host) hipHostFree(c->status_host);
    if (c->xfer_host) hipHostFree(c->xfer_host);
    for (int k = 0; k < 2; ++k) {
        if (c->bounce[k]) hipHostFree(c->bounce[k]);
        if (c->bounce_ev[k]) hipEventDestroy(c->bounce_ev[k]);
    }
    void* ptrs[] = {c->X, c->lp, c->acc, c->acc_count, c->iota, c->qout, c->fout, c->newlp, c->evalX,
                    c->evallp, c->tp0, c->tp1, c->tp1_full, c->chain, c->chain_lp, c->own_shard_bufs ? c->sendbuf : nullptr,
                    c->own_shard_bufs ? c->gathered : nullptr};
    for (void* p : ptrs)
        if (p) hipFree(p);
    for (double* s : c->snap)
        if (s) hipFree(s);

    for (auto& s : c->ring) {
        if (s.order) hipFree(s.order);       // the slot's single block
        if (s.host) hipHostFree(s.host);
        if (s.uploaded) hipEventDestroy(s.uploaded);
        s.uploaded_ref = nullptr;
        if (s.consumed) hipEventDestroy(s.consumed);
    }
    for (auto& e : c->pipe_batch_ev)
        if (e) hipEventDestroy(e);
    if (c->pipe_done) hipHostFree(c->pipe_done);
    if (c->pipe_arrived) hipFree(c->pipe_arrived);
    for (auto& g : c->gslot) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    {
        auto& p = c->cplan;
        void* q[] = {p.order, p.p0, p.p1, p.p2, p.s0, p.uacc, p.logu, p.fac, c->pull_counts, c->replay_counts};
        for (void* x : q)
            if (x) hipFree(x);
    }
    for (double* q : c->mscale)
        if (q) hipFree(q);
    for (auto& bp : c->bulk) {
        if (bp.host) hipHostFree(bp.host);
        if (bp.dev) hipFree(bp.dev);
        if (bp.done) hipEventDestroy(bp.done);
    }
    if (c->disp) hipFree(c->disp);
    if (c->dbg) hipFree(c->dbg);
    if (c->noise_host) hipHostFree(c->noise_host);
    if (c->noise_ev) hipEventDestroy(c->noise_ev);
    if (c->persist_bar) hipFree(c->persist_bar);
    if (c->persist_started) hipHostFree(c->persist_started);
    if (c->persist_ver) hipFree(c->persist_ver);
    if (c->d_desc) hipFree(c->d_desc);
    if (c->d_ctr) hipFree(c->d_ctr);
    if (c->h_ctr) hipHostFree(c->h_ctr);
    for (auto e : c->thr_ev)
        if (e) hipEventDestroy(e);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    for (auto e : c->prof) hipEventDestroy(e);
    direct_detach(c);
    if (c->my_flags) hipFree(c->my_flags);
    if (c->peer_table) hipFree(c->peer_table);
    if (c->direct_counts) hipFree(c->direct_counts);
    if (c->up_stream) hipStreamDestroy(c->up_stream);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return 0;
}

int emx_set_stream(emx_ctx* c, void* s) {
    NEED(c, c, "null ctx");
    HIPOK(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return 0;
}

// Waiting for the stream: poll first.  hipStreamSynchronize / hipEventSynchronize park the thread on the completion signal
// and wake it through the driver -- 10-20 us after the last kernel finished, which is a few per cent of a 20-step call at
// the headline size (0.5 ms).  A query loop sees the completion within a microsecond; after 2 ms of polling (a long run: the
// wake-up no longer matters) it falls back to the blocking call so that a waiting host thread does not burn a core.
static const bool g_spin_sync = !(getenv("EMX_SPIN_SYNC") && atoi(getenv("EMX_SPIN_SYNC")) == 0);

static hipError_t wait_stream(hipStream_t s) {
    if (g_spin_sync) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; ++it) {
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) return e;
            if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    return hipStreamSynchronize(s);
}

static hipError_t wait_event(hipEvent_t ev) {
    if (g_spin_sync) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; ++it) {
            const hipError_t e = hipEventQuery(ev);
            if (e != hipErrorNotReady) return e;
            if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    return hipEventSynchronize(ev);
}

int emx_sync(emx_ctx* c) {
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, wait_stream(c->stream));
    return persist_settle(c);
}

int emx_status(emx_ctx* c, uint32_t* bits) {
    HIPOK(c, wait_stream(c->stream));      // every launch that could still raise a bit has finished
    {
        const int rcs = persist_settle(c);     // a persistent launch that gave up untouched is redone here: no bit, no void state
        if (rcs) return rcs;
    }
    uint32_t b = 0;
    for (int k = 0; k < 5; ++k)
        if (__atomic_exchange_n(&c->status_host[k], 0u, __ATOMIC_ACQ_REL)) b |= 1u << k;
    if (b & ST_EXCHANGE_TIMEOUT) c->direct_dead = true;      // sticky on the host too: emx_direct_halfstep / emx_run refuse from here on
    if (b & ST_EXCHANGE_TIMEOUT) c->persist_grid = 0;        // the persistent kernel's barrier words (counters, the dead mark) restart with its next launch
    *bits = b;
    return 0;
}

int emx_set_tuning(emx_ctx* c, const char* key, int64_t v) {
    if (!strcmp(key, "spw")) {
        c->tune_spw = v;
        return 0;
    }
    if (!strcmp(key, "phase_clock")) {     // v != 0: record s_memtime at the phase boundaries of every launch
        if (v && !c->dbg) {
            c->dbg_blocks = 4096;
            HIPOK(c, hipMalloc((void**)&c->dbg, (size_t)c->dbg_blocks * 16 * 8));
            HIPOK(c, hipMemset(c->dbg, 0, (size_t)c->dbg_blocks * 16 * 8));
        }
        if (!v && c->dbg) {
            hipStreamSynchronize(c->stream);
            hipFree(c->dbg);
            c->dbg = nullptr;
        }
        return 0;
    }
    if (!strcmp(key, "small_kernel")) {
        PIPE_STOP(c);
        c->tune_small = v;
        return 0;
    }
    if (!strcmp(key, "gauss_materialize")) {
        c->tune_gauss_materialize = v;
        return 0;
    }
    if (!strcmp(key, "direct_timeout_ms")) {   // direct exchange: how long the device-side barrier waits for a peer
        c->tune_direct_timeout_ms = v > 0 ? v : 5000;
        return 0;
    }
    if (!strcmp(key, "mt_pipeline")) {   // exact mode: -1 auto, 0 plans made inline by the calling thread, k > 0 finisher threads
        PIPE_STOP(c);
        c->tune_mt_pipeline = v;
        return 0;
    }
    if (!strcmp(key, "throttle")) {
        c->tune_throttle = v;
        return 0;
    }
    if (!strcmp(key, "graph")) {        // 1: replay the native NATIVE_BATCH_MAX-step block as a hipGraph (default 0: plain launches)
        c->tune_graph = v;
        return 0;
    }
    if (!strcmp(key, "prep_hint")) {   // upcoming steps driven through emx_step_begin: native prep batch size
        c->prep_hint = v > 0 ? v : 1;
        return 0;
    }
    if (!strcmp(key, "ablate")) {       // timing experiments (tools/ablate.py): compiled into the experiments flavour only
        NEED(c, EMX_EXPERIMENTS || v == 0, "tuning \"ablate\" needs the experiments build of the library (EMX_BUILD_FLAVOUR=exp python -m emcee_amd._build; EMX_LIB=.../libemx_exp.so)");
        c->tune_ablate = v;
        return 0;
    }
    if (!strcmp(key, "waves_per_block")) {
        c->tune_wpb = (v == 1 || v == 2 || v == 4 || v == 8) ? v : 0;   // 0: automatic
        return 0;
    }
    if (!strcmp(key, "dense_wide")) {      // 1: take the wide-target path (emx_wide.hip) whatever the ndim -- parity tests against the fused kernel
        c->tune_dense_wide = v == 2 ? 2 : (v ? 1 : 0);      // 2: the wide path with the single-role log-prob kernel only
        graph_invalidate(c);
        return 0;
    }
    if (!strcmp(key, "mt_device")) {         // 0: exact-mode plans never from the device producer (emx_mtdev.hpp)
        PIPE_STOP(c);
        c->tune_mt_device = v < 0 ? 0 : (v > 2 ? 2 : v);
        return 0;
    }
    if (!strcmp(key, "mt_device_min_walkers")) {
        PIPE_STOP(c);
        c->tune_mt_device_min = v < 8192 ? 8192 : v;
        return 0;
    }
    if (!strcmp(key, "mt_tok_wshift") || !strcmp(key, "mt_tok_tail")) {       // the device tokenizer's window rule (takes effect when a producer starts)
        PIPE_STOP(c);
        (key[7] == 'w' ? c->tune_mt_tok_wshift : c->tune_mt_tok_tail) = v;
        return 0;
    }
    if (!strcmp(key, "mt_device_lookahead")) {       // batches produced ahead of the one asked for (tests: 0 keeps batch 0's raw pieces readable)
        c->tune_mt_lookahead = v < 0 ? 0 : (v > 2 ? 2 : v);
        return 0;
    }
    if (!strcmp(key, "slab")) {              // 0: the per-tile kernel at every padded ndim (parity tests, A/B); 1: slab form from padded 112; 2: from padded 80
        c->tune_slab = v < 0 ? 0 : (v > 2 ? 2 : v);
        return 0;
    }
    if (!strcmp(key, "persist_exact_mix")) {
        PIPE_STOP(c);
        c->tune_persist_exact_mix = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "mt_device_finish")) {      // 0: the host pipeline's finisher threads convert every draw themselves (rounds 1-4)
        PIPE_STOP(c);
        c->tune_mt_device_finish = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "slab_skew")) {
        c->tune_slab_skew = v < 0 ? 0 : (v > 4 ? 4 : v);
        return 0;
    }
    if (!strcmp(key, "persist")) {           // 0: never the persistent half-step kernel (k_persist)
        c->tune_persist = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "persist_slab_skew")) {
        c->tune_persist_slab_skew = v < 0 ? 0 : (v > 4 ? 4 : v);
        return 0;
    }
    if (!strcmp(key, "persist_slab_local_max_walkers")) {
        c->tune_persist_slab_local_max = v;
        return 0;
    }
    if (!strcmp(key, "persist_odd")) {       // 0: odd ndim on the per-half-step launches
        c->tune_persist_odd = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "persist_exact_regen_max_walkers")) {
        PIPE_STOP(c);
        c->tune_persist_exact_regen_max = v;
        return 0;
    }
    if (!strcmp(key, "mt_device_min_walkers_regen")) {
        PIPE_STOP(c);
        c->tune_mt_device_min_regen = v;
        return 0;
    }
    if (!strcmp(key, "mt_regen_min_walkers")) {      // 0: never k_plan_regen (takes effect when a pipeline starts)
        PIPE_STOP(c);
        c->tune_mt_regen_min = v < 0 ? 0 : v;
        return 0;
    }
    if (!strcmp(key, "persist_rows_late")) {
        c->tune_persist_rows_late = v < 0 ? 0 : v > 2 ? 2 : v;
        return 0;
    }
    if (!strcmp(key, "persist_stagger")) {
        c->tune_persist_stagger = v < 0 ? -1 : v > 1279 ? 1279 : v;
        return 0;
    }
    if (!strcmp(key, "persist_max_halfsteps")) {
        c->tune_persist_max_halfsteps = v;
        return 0;
    }
    if (!strcmp(key, "persist_slab")) {      // 0: padded ndim 80 ... 128 on the per-half-step launches (k_halfstep_slab / k_halfstep)
        c->tune_persist_slab = (v == 1 || v == 2) ? v : 0;          // (2: also where the per-half-step slab kernel is level -- persist_slab_ok)
        return 0;
    }
    if (!strcmp(key, "persist_local")) {     // 0: never the one-XCD form (k_persist<..., LOCAL>)
        c->tune_persist_local = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "persist_exact")) {     // 0: exact mode always on the per-half-step launches
        PIPE_STOP(c);
        c->tune_persist_exact = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "persist_exact_max_walkers")) {
        PIPE_STOP(c);
        c->tune_persist_exact_max = v;
        return 0;
    }
    if (!strcmp(key, "persist_exact_steps")) {
        c->tune_persist_exact_steps = std::max<int64_t>(1, std::min<int64_t>(v, 16));
        return 0;
    }
    if (!strcmp(key, "fetch_blocks")) {            // k_plan_fetch's workgroups beside a device-wide persistent launch (0: one per piece)
        c->tune_fetch_blocks = std::max<int64_t>(0, std::min<int64_t>(v, 65536));
        return 0;
    }
    if (!strcmp(key, "fetch_avoid")) {             // k_plan_fetch's workgroups decline on the XCD of a one-XCD persistent launch
        c->tune_fetch_avoid = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "test_fetch_delay_us")) {     // tests: k_plan_fetch idles first (its consumers must wait for it)
        c->tune_fetch_delay_us = std::max<int64_t>(0, std::min<int64_t>(v, 100000));
        return 0;
    }
    if (!strcmp(key, "persist_span")) {
        c->tune_persist_span = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "persist_mix")) {       // 0: DE and snooker steps of a mixture in launches of their own
        c->tune_persist_mix = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "persist_valu")) {      // 0: element-wise targets always on the per-half-step launches
        c->tune_persist_valu = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "persist_local_max_walkers")) {
        c->tune_persist_local_max = v;
        return 0;
    }
    if (!strcmp(key, "persist_gauss_wpb")) {       // waves per workgroup of the persistent Gaussian kernel: 1, 2, 4 or 8 (0: automatic)
        c->tune_persist_gauss_wpb = (v == 1 || v == 2 || v == 4 || v == 8) ? v : 0;
        return 0;
    }
    if (!strcmp(key, "persist_test_skew")) {       // tests only: the persistent kernel's barriers wait for a count that never comes
        c->tune_persist_test_skew = v;
        return 0;
    }
    if (!strcmp(key, "persist_timeout_ms")) {      // bound of a device-wide barrier wait inside k_persist
        c->tune_persist_timeout_ms = v > 0 ? v : 2000;
        return 0;
    }
    if (!strcmp(key, "persist_min_walkers")) {      // smallest ensemble the persistent kernel is used for
        c->tune_persist_min_walkers = v > 0 ? v : 2;
        return 0;
    }
    if (!strcmp(key, "replay_two_pass")) {
        c->tune_replay_two_pass = v ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "full_plan")) {     // 1: native plans with every column (default 0: what the fused kernel reads)
        PIPE_STOP(c);                    // (the exact-mode pipeline hands its steps over finished when every column is asked for)
        c->tune_full_plan = v ? 1 : 0;
        drop_prepared(c);
        return 0;
    }
    if (!strcmp(key, "blocks_per_cu")) {
        c->tune_bpc = v > 0 ? v : 2;
        return 0;
    }
    FAIL(c, -1, "unknown tuning key %s", key);
}

// Device -> host copies of more than a few MB into ordinary (pageable) memory: HIP's own path managed 1.6 GB/s into a fresh
// NumPy array on the GPU box (21 ms for the 33.5 MB state at 65 536 x 64).  Here the data crosses PCIe into two pinned 8 MB
// halves in turn and a host memcpy empties one while the DMA fills the other.  Synchronous; small copies and copies into
// memory the caller pinned go straight through.  Returns with everything enqueued on the stream before it complete.
static int big_copy_to_host(emx_ctx* c, void* dst, const void* src, size_t bytes) {
    constexpr size_t CH = 8u << 20;
    if (bytes == 0) return 0;
    bool direct = bytes < (2u << 20);
    if (!direct) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess) direct = at.type == hipMemoryTypeHost;      // pinned by the caller
        else (void)hipGetLastError();                                                                     // ordinary memory
    }
    if (direct) {
        HIPOK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    for (int k = 0; k < 2; ++k) {
        if (!c->bounce[k]) HIPOK(c, hipHostMalloc((void**)&c->bounce[k], CH, hipHostMallocDefault));
        if (!c->bounce_ev[k]) HIPOK(c, hipEventCreateWithFlags(&c->bounce_ev[k], hipEventDisableTiming));
    }
    const size_t nch = (bytes + CH - 1) / CH;
    auto issue = [&](size_t k) -> hipError_t {
        const size_t off = k * CH, n = std::min(CH, bytes - off);
        hipError_t e = hipMemcpyAsync(c->bounce[k & 1], (const char*)src + off, n, hipMemcpyDeviceToHost, c->stream);
        return e != hipSuccess ? e : hipEventRecord(c->bounce_ev[k & 1], c->stream);
    };
    HIPOK(c, issue(0));
    for (size_t k = 0; k < nch; ++k) {
        if (k + 1 < nch) HIPOK(c, issue(k + 1));
        HIPOK(c, hipEventSynchronize(c->bounce_ev[k & 1]));
        const size_t off = k * CH;
        memcpy((char*)dst + off, c->bounce[k & 1], std::min(CH, bytes - off));
    }
    return 0;
}

// the other direction: a host memcpy fills one pinned half while the DMA drains the other
static int big_copy_to_device(emx_ctx* c, void* dst, const void* src, size_t bytes) {
    constexpr size_t CH = 8u << 20;
    if (bytes == 0) return 0;
    bool direct = bytes < (2u << 20);
    if (!direct) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, src) == hipSuccess) direct = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    if (direct) {
        HIPOK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        HIPOK(c, hipStreamSynchronize(c->stream));      // the caller may reuse src
        return 0;
    }
    for (int k = 0; k < 2; ++k) {
        if (!c->bounce[k]) HIPOK(c, hipHostMalloc((void**)&c->bounce[k], CH, hipHostMallocDefault));
        if (!c->bounce_ev[k]) HIPOK(c, hipEventCreateWithFlags(&c->bounce_ev[k], hipEventDisableTiming));
    }
    const size_t nch = (bytes + CH - 1) / CH;
    for (size_t k = 0; k < nch; ++k) {
        const size_t off = k * CH, n = std::min(CH, bytes - off);
        if (k >= 2) HIPOK(c, hipEventSynchronize(c->bounce_ev[k & 1]));      // the DMA that last read this half is done
        memcpy(c->bounce[k & 1], (const char*)src + off, n);
        HIPOK(c, hipMemcpyAsync((char*)dst + off, c->bounce[k & 1], n, hipMemcpyHostToDevice, c->stream));
        HIPOK(c, hipEventRecord(c->bounce_ev[k & 1], c->stream));
    }
    HIPOK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int emx_set_state(emx_ctx* c, const double* coords, const double* log_prob) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    {
        const int rc = big_copy_to_device(c, c->X, coords, (size_t)c->N * c->D * 8);
        if (rc) return rc;
    }
    if (log_prob) HIPOK(c, hipMemcpyAsync(c->lp, log_prob, (size_t)c->N * 8, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int emx_get_state(emx_ctx* c, double* coords, double* log_prob) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    if (log_prob) HIPOK(c, hipMemcpyAsync(log_prob, c->lp, (size_t)c->N * 8, hipMemcpyDeviceToHost, c->stream));
    if (coords) {
        const int rc = big_copy_to_host(c, coords, c->X, (size_t)c->N * c->D * 8);
        if (rc) return rc;
    }
    HIPOK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// The caller's own batched log-prob on device buffers (include/emx.h): ensemble.py:486-487's vectorised call without the PCIe
// hop.  The fused closed-form targets are replaced by three passes on the stream (launch_split).
int emx_set_target_callback(emx_ctx* c, emx_device_log_prob_fn fn, void* user) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, fn != nullptr, "emx_set_target_callback: no function");
    PIPE_STOP(c);
    drop_prepared(c);          // plans made ahead were shaped (lean or full) for the previous target
    HIPOK(c, hipStreamSynchronize(c->stream));
    c->cb_fn = fn;
    c->cb_user = user;
    c->Dp = 0;
    graph_invalidate(c);
    c->graph_warm = false;
    c->target = EMX_TARGET_DEVICE_CALLBACK;
    c->tscale = 1.0;
    return 0;
}

// ---- snapshots: the state a call handed out stays on the device until somebody reads it -----------------------------------
// (ensemble.py:441-447: run_mcmc(None | previous State) continues from the state the last call returned -- which already IS
// the device state; a snapshot is the copy-on-write copy taken when the next call is about to change it while the old object
// is still alive, so that the old object keeps the values it was returned with: 2 x 33.5 MB inside HBM instead of over PCIe)
int emx_snapshot_save(emx_ctx* c, int32_t slot) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, slot >= 0 && slot < emx_ctx::NSNAPSHOT, "snapshot slot out of range");
    const size_t nx = (size_t)c->N * c->D, nl = (size_t)c->N;
    if (!c->snap[slot]) HIPOK(c, hipMalloc((void**)&c->snap[slot], (nx + nl) * 8));
    HIPOK(c, hipMemcpyAsync(c->snap[slot], c->X, nx * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPOK(c, hipMemcpyAsync(c->snap[slot] + nx, c->lp, nl * 8, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int emx_snapshot_read(emx_ctx* c, int32_t slot, double* coords, double* log_prob) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, slot >= 0 && slot < emx_ctx::NSNAPSHOT && c->snap[slot], "no snapshot in slot %d", slot);
    const size_t nx = (size_t)c->N * c->D;
    if (log_prob) HIPOK(c, hipMemcpyAsync(log_prob, c->snap[slot] + nx, (size_t)c->N * 8, hipMemcpyDeviceToHost, c->stream));
    if (coords) {
        const int rc = big_copy_to_host(c, coords, c->snap[slot], nx * 8);
        if (rc) return rc;
    }
    HIPOK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int emx_snapshot_restore(emx_ctx* c, int32_t slot) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, slot >= 0 && slot < emx_ctx::NSNAPSHOT && c->snap[slot], "no snapshot in slot %d", slot);
    const size_t nx = (size_t)c->N * c->D;
    HIPOK(c, hipMemcpyAsync(c->X, c->snap[slot], nx * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPOK(c, hipMemcpyAsync(c->lp, c->snap[slot] + nx, (size_t)c->N * 8, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int emx_snapshot_free(emx_ctx* c, int32_t slot) {
    NEED(c, slot >= 0 && slot < emx_ctx::NSNAPSHOT, "snapshot slot out of range");
    if (c->snap[slot]) {
        HIPOK(c, hipSetDevice(c->device));
        HIPOK(c, hipStreamSynchronize(c->stream));
        HIPOK(c, hipFree(c->snap[slot]));
        c->snap[slot] = nullptr;
    }
    return 0;
}

int emx_get_accepted(emx_ctx* c, uint8_t* mask) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipMemcpyAsync(mask, c->acc, (size_t)c->N, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int emx_set_target(emx_ctx* c, int32_t kind, const double* p0, const double* p1, double scale) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, kind >= EMX_TARGET_HOST && kind <= EMX_TARGET_BOX, "unknown target kind %d", kind);
    // Everything that can fail (argument checks, the Cholesky factorisation, allocations, uploads) happens on
    // fresh buffers; the context's target is replaced only once the new one is complete, so a refused target
    // leaves the previous one fully usable.
    const size_t D = (size_t)c->D;
    double *ntp0 = nullptr, *ntp1 = nullptr, *ntpf = nullptr;
    int nDp = c->Dp;
    struct Guard {          // frees the new buffers on every early return
        double*& a;
        double*& b;
        double*& f;
        ~Guard() {
            if (a) hipFree(a);
            if (b) hipFree(b);
            if (f) hipFree(f);
        }
    } guard{ntp0, ntp1, ntpf};
    if (kind == EMX_TARGET_DIAG_GAUSS || kind == EMX_TARGET_DENSE_GAUSS) {
        NEED(c, p0 && p1, "target needs (mu, ivar|icov)");
        std::vector<double> img;
        if (kind == EMX_TARGET_DENSE_GAUSS) {
            nDp = (int)((D + 15) / 16 * 16);
            NEED(c, nDp <= 2048, "dense Gaussian target supports ndim <= 2048; got %d", c->D);
            // -0.5 d^T A d with A = sym(icov) = L L^T  ==  -0.5 |L^T d|^2.  Factor once on the host and upload the
            // image the kernel stages into LDS: L in MFMA B-fragment order (zero padded) followed by the mean.
            const int Dp = nDp, KK = Dp / 4, n = (int)D;
            std::vector<double> Lm((size_t)n * n, 0.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) {
                    double sum = 0.5 * (p1[(size_t)i * n + j] + p1[(size_t)j * n + i]);
                    for (int k = 0; k < j; ++k) sum -= Lm[(size_t)i * n + k] * Lm[(size_t)j * n + k];
                    if (i == j) {
                        NEED(c, sum > 0.0 && std::isfinite(sum),
                             "dense Gaussian target: icov must be symmetric positive definite (Cholesky failed at row %d)", i);
                        Lm[(size_t)i * n + i] = std::sqrt(sum);
                    } else {
                        Lm[(size_t)i * n + j] = sum / Lm[(size_t)j * n + j];
                    }
                }
            img.assign((size_t)Dp * Dp + Dp, 0.0);
            for (int nb = 0; nb < Dp / 16; ++nb)
                for (int kk = 0; kk < KK; ++kk)
                    for (int l = 0; l < 64; ++l) {
                        const int k = 4 * kk + (l >> 4), col = 16 * nb + (l & 15);
                        if (k < n && col < n && k >= col) img[((size_t)nb * KK + kk) * 64 + l] = Lm[(size_t)k * n + col];
                    }
            for (int d = 0; d < n; ++d) img[(size_t)Dp * Dp + d] = p0[d];
            if (Dp <= DENSE_FUSED_MAX_DP) {
                // the fused kernel and k_small_run stage only the non-zero 16 x 16 blocks (dense_block, emx_kernels.hpp); the
                // full image stays next to it for the wide-target kernels (tuning "dense_wide")
                HIPOK(c, hipMalloc((void**)&ntpf, img.size() * 8));
                HIPOK(c, hipMemcpy(ntpf, img.data(), img.size() * 8, hipMemcpyHostToDevice));
                const int B = Dp / 16;
                std::vector<double> packed((size_t)dense_img_doubles(Dp) + Dp, 0.0);
                for (int nb = 0; nb < B; ++nb)
                    for (int kb = nb; kb < B; ++kb)
                        for (int i = 0; i < 4; ++i)
                            for (int l = 0; l < 64; ++l)
                                packed[((size_t)dense_block(B, nb, kb) * 4 + i) * 64 + l] = img[((size_t)nb * KK + 4 * kb + i) * 64 + l];
                for (int d = 0; d < Dp; ++d) packed[(size_t)dense_img_doubles(Dp) + d] = img[(size_t)Dp * Dp + d];
                img.swap(packed);
            }
        }
        HIPOK(c, hipMalloc((void**)&ntp0, D * 8));
        HIPOK(c, hipMemcpy(ntp0, p0, D * 8, hipMemcpyHostToDevice));
        const double* src1 = kind == EMX_TARGET_DENSE_GAUSS ? img.data() : p1;
        const size_t n1 = kind == EMX_TARGET_DENSE_GAUSS ? img.size() : D;
        HIPOK(c, hipMalloc((void**)&ntp1, n1 * 8));
        HIPOK(c, hipMemcpy(ntp1, src1, n1 * 8, hipMemcpyHostToDevice));
    }
    // A running exact-mode pipeline was configured for the target that goes away here (device finish -- raw / regen steps -- is a
    // matter of the target and the consumer: pipe_start): it is retired, the generator continues behind the last step taken.  (Round-5
    // advisor: emx_set_target left it running; emx_run restarts a pipeline whose consumer changed, but a hand-over mode chosen for
    // the OLD target must not outlive it either.)
    PIPE_STOP(c);
    HIPOK(c, hipStreamSynchronize(c->stream));      // no kernel still reads the old parameters
    drop_prepared(c);                               // plans made ahead were shaped (lean or full) for the previous target
    std::swap(c->tp0, ntp0);                        // the guard now frees the OLD buffers
    std::swap(c->tp1, ntp1);
    std::swap(c->tp1_full, ntpf);
    c->Dp = nDp;
    for (auto& f : c->persist_fits) f = -1;
    c->persist_mix_fits = -1;
    graph_invalidate(c);
    c->graph_warm = false;
    c->target = kind;
    c->tscale = (kind == EMX_TARGET_ROSENBROCK) ? (scale != 0.0 ? scale : 20.0) : 1.0;
    return 0;
}

static int eval_rows(emx_ctx* c, double* X, double* lp, int64_t n) {
    NEED(c, c->target != EMX_TARGET_HOST, "no device target set");
    emx_move_desc mv = c->moves[0];
    int rc = launch_split(c, MOVE_EVAL, c->target, 1, 0, 0, (int)n, 0, (int)n, &mv, nullptr, c->iota, X, lp,
                          nullptr, nullptr, nullptr);
    return rc;
}

int emx_eval_state_log_prob(emx_ctx* c) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    return eval_rows(c, c->X, c->lp, c->N);
}

int emx_eval_log_prob(emx_ctx* c, const double* coords, int64_t n, double* out) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, n >= 0 && n <= c->N, "emx_eval_log_prob: n must be <= nwalkers");
    if (n == 0) return 0;
    HIPOK(c, hipMemcpyAsync(c->evalX, coords, (size_t)n * c->D * 8, hipMemcpyHostToDevice, c->stream));
    int rc = eval_rows(c, c->evalX, c->evallp, n);
    if (rc) return rc;
    HIPOK(c, hipMemcpyAsync(out, c->evallp, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int emx_set_moves(emx_ctx* c, int32_t nmoves, const emx_move_desc* moves, const double* cdf) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    NEED(c, nmoves >= 1, "need at least one move");
    PIPE_STOP(c);
    for (int i = 0; i < nmoves; ++i) {
        NEED(c, moves[i].kind >= 0 && moves[i].kind <= EMX_MOVE_GAUSS, "unknown move kind");
        if (moves[i].kind == EMX_MOVE_GAUSS) {
            NEED(c, moves[i].nsplits == 1, "the Gaussian move updates the whole ensemble at once (nsplits must be 1)");
            NEED(c, moves[i].reserved >= EMX_GAUSS_VECTOR && moves[i].reserved <= EMX_GAUSS_SEQUENTIAL, "unknown Gaussian mode");
            NEED(c, moves[i].a == 0.0 || moves[i].g0 >= 0.0, "'factor' must be >= 1.0");
            continue;
        }
        NEED(c, moves[i].nsplits >= 2 && moves[i].nsplits <= 64, "nsplits must be in [2, 64]");
        NEED(c, moves[i].kind != EMX_MOVE_SNOOKER || moves[i].nsplits >= 4, "snooker needs nsplits >= 4");
        NEED(c, moves[i].nsplits <= c->N, "more splits than walkers");
        // DE draws two distinct complement members (de.py:49): refuse a complement of one in every RNG mode
        // (the native slot function would reduce modulo zero)
        NEED(c, moves[i].kind != EMX_MOVE_DE || c->N - (c->N + moves[i].nsplits - 1) / moves[i].nsplits >= 2,
             "complement too small for this move");
    }
    c->moves.assign(moves, moves + nmoves);
    c->cdf.assign(cdf, cdf + nmoves);
    for (auto& f : c->persist_fits) f = -1;
    c->persist_mix_fits = -1;
    for (double* p : c->mscale)
        if (p) {
            hipStreamSynchronize(c->stream);
            hipFree(p);
        }
    c->mscale.assign(nmoves, nullptr);
    for (int i = 0; i < nmoves; ++i)
        if (moves[i].kind == EMX_MOVE_GAUSS && !c->disp) HIPOK(c, hipMalloc((void**)&c->disp, (size_t)c->N * c->D * 8));
    c->prepared.clear();
    graph_invalidate(c);
    c->graph_warm = false;
    return 0;
}

int emx_set_rng_mode(emx_ctx* c, int32_t mode) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    NEED(c, mode >= 0 && mode <= 2, "unknown rng mode");
    PIPE_STOP(c);
    c->rng_mode = mode;
    drop_prepared(c);
    return 0;
}

int emx_set_move_scale(emx_ctx* c, int32_t mi, const double* sd, int32_t n) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, mi >= 0 && mi < (int)c->moves.size() && c->moves[mi].kind == EMX_MOVE_GAUSS, "emx_set_move_scale: not a Gaussian move");
    NEED(c, (sd == nullptr && n == 0) || n == c->D, "scale vector must have ndim entries");
    HIPOK(c, hipStreamSynchronize(c->stream));
    if (c->mscale[mi]) hipFree(c->mscale[mi]), c->mscale[mi] = nullptr;
    if (sd) {
        HIPOK(c, hipMalloc((void**)&c->mscale[mi], (size_t)n * 8));
        HIPOK(c, hipMemcpy(c->mscale[mi], sd, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    return 0;
}

int emx_get_move(emx_ctx* c, int32_t mi, emx_move_desc* out) {
    NEED(c, mi >= 0 && mi < (int)c->moves.size(), "bad move index");
    drop_prepared(c);          // plans made ahead of time advanced the sequential cursor: take that back
    *out = c->moves[mi];
    return 0;
}

int emx_rng_set_mt19937(emx_ctx* c, const uint32_t key[624], int32_t pos, int32_t hg, double cached) {
    PIPE_STOP(c);
    NEED(c, pos >= 0 && pos <= 624, "bad MT19937 position");
    c->mt.set_state(key, pos, hg, cached);
    return 0;
}

int emx_rng_get_mt19937(emx_ctx* c, uint32_t key[624], int32_t* pos, int32_t* hg, double* cached) {
    PIPE_STOP(c);          // the state after the last step taken (what the pipeline produced ahead is dropped)
    memcpy(key, c->mt.key, sizeof(c->mt.key));
    *pos = c->mt.pos;
    *hg = c->mt.has_gauss;
    *cached = c->mt.gauss;
    return 0;
}

int emx_rng_set_philox(emx_ctx* c, uint64_t seed, uint64_t step) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    c->ph_seed = seed;
    c->ph_step = step;
    drop_prepared(c);
    graph_invalidate(c);   // the seed is baked into the captured advance kernel
    return 0;
}

int emx_rng_get_philox(emx_ctx* c, uint64_t* seed, uint64_t* step) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    *seed = c->ph_seed;
    *step = c->ph_step;
    return 0;
}

int emx_chain_config(emx_ctx* c, int64_t cap) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, cap >= c->stored, "capacity below the number of stored steps");
    if (cap == c->cap) return 0;
    double *nc = nullptr, *nl = nullptr;
    const size_t row = (size_t)c->N * c->D * 8, lrow = (size_t)c->N * 8;
    if (cap > 0) {
        hipError_t e = hipMalloc((void**)&nc, row * cap);
        if (e != hipSuccess) FAIL(c, -4, "chain allocation of %.2f GB failed: %s", row * cap / 1e9, hipGetErrorString(e));
        e = hipMalloc((void**)&nl, lrow * cap);
        if (e != hipSuccess) {
            hipFree(nc);
            FAIL(c, -4, "chain log_prob allocation failed: %s", hipGetErrorString(e));
        }
        if (c->stored > 0) {
            HIPOK(c, hipMemcpyAsync(nc, c->chain, row * c->stored, hipMemcpyDeviceToDevice, c->stream));
            HIPOK(c, hipMemcpyAsync(nl, c->chain_lp, lrow * c->stored, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    HIPOK(c, hipStreamSynchronize(c->stream));
    graph_invalidate(c);   // captured kernels hold the chain base pointers
    if (c->chain) hipFree(c->chain);
    if (c->chain_lp) hipFree(c->chain_lp);
    c->chain = nc;
    c->chain_lp = nl;
    c->cap = cap;
    return 0;
}

int emx_chain_reset(emx_ctx* c) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    c->stored = 0;
    c->proposals = 0;
    HIPOK(c, hipMemsetAsync(c->acc_count, 0, (size_t)c->N * 4, c->stream));
    return 0;
}

int emx_graph_state(emx_ctx* c, int32_t* disabled, int32_t* captured) {
    *disabled = c->graph_disabled ? 1 : 0;
    *captured = (c->gslot[0].valid ? 1 : 0) | (c->gslot[1].valid ? 2 : 0);
    return 0;
}

int emx_iteration(emx_ctx* c, int64_t* stored, int64_t* proposals) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    *stored = c->stored;
    *proposals = c->proposals;
    return 0;
}

int emx_chain_read(emx_ctx* c, int32_t what, int64_t start, int64_t stop, int64_t stride, double* out) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, stride >= 1 && start >= 0 && stop <= c->stored, "chain slice out of range");
    const size_t row = what == 0 ? (size_t)c->N * c->D * 8 : (size_t)c->N * 8;
    const char* base = what == 0 ? (const char*)c->chain : (const char*)c->chain_lp;
    char* o = (char*)out;
    if (stride == 1) {
        if (stop > start) {
            const int rc = big_copy_to_host(c, o, base + row * start, row * (size_t)(stop - start));
            if (rc) return rc;
        }
    } else if (row >= (2u << 20)) {
        for (int64_t s = start; s < stop; s += stride, o += row) {
            const int rc = big_copy_to_host(c, o, base + row * s, row);
            if (rc) return rc;
        }
    } else {
        for (int64_t s = start; s < stop; s += stride, o += row)
            HIPOK(c, hipMemcpyAsync(o, base + row * s, row, hipMemcpyDeviceToHost, c->stream));
    }
    HIPOK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int emx_accepted_counts(emx_ctx* c, double* out) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    std::vector<uint32_t> h((size_t)c->N);
    HIPOK(c, hipMemcpyAsync(h.data(), c->acc_count, (size_t)c->N * 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < c->N; ++i) out[i] = (double)h[i];
    return 0;
}

// ---- stepping ----------------------------------------------------------------------------
static int upload_plan(emx_ctx* c, emx_ctx::PlanSlot& s) {
    const size_t N = (size_t)c->N;
    const int stretch = c->cur.move >= 0 && c->moves[c->cur.move].kind == EMX_MOVE_STRETCH;
    HIPOK(c, hipMemcpyAsync(s.order, s.host, plan_upload_bytes(N, stretch && c->world == 1 ? EMX_MOVE_STRETCH : EMX_MOVE_DE),
                            hipMemcpyHostToDevice, c->stream));   // same layout on both sides
    hipLaunchKernelGGL(k_plan_logs, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (int)N, (int)c->D, stretch,
                       s.s0, s.uacc, s.logu, s.fac);
    HIPOK(c, hipGetLastError());
    return 0;
}

// Next ring slot.  A slot whose pinned staging buffer the HOST is about to rewrite must first have
// been consumed by the device (event); device-written plans (native mode) are ordered by the stream.
static int acquire_slot(emx_ctx* c, emx_ctx::PlanSlot** out, bool host_written) {
    c->ring_pos = (c->ring_pos + 1) % PLAN_RING;
    auto& s = c->ring[c->ring_pos];
    if (s.busy && host_written) {
        HIPOK(c, hipEventSynchronize(s.consumed_ref));
        s.busy = false;
    }
    if (host_written && !s.host)   // pinned staging is only needed by the host-generated (exact / inputs) plans
        HIPOK(c, hipHostMalloc((void**)&s.host, (size_t)c->N * 32, hipHostMallocDefault));
    s.host_written = host_written;
    *out = &s;
    return 0;
}

// ---- Gaussian Metropolis move: the displacement rows of one step -------------------------------
static void gauss_args(emx_ctx* c, int mi, GaussDispArgs& a, double f) {
    const emx_move_desc& mv = c->moves[mi];
    a.disp = c->disp;
    a.scale = c->mscale[mi];
    a.sigma = mv.sigma;
    a.f = f;
    a.N = (int32_t)c->N;
    a.D = c->D;
    a.mode = mv.reserved;
}

// host normals (exact / inputs modes) -> disp = (f * scale) * n
static int gauss_upload_normals(emx_ctx* c, int mi, const double* normals, double f, bool pinned_staging) {
    const size_t n = (size_t)c->N * c->D;
    HIPOK(c, hipMemcpyAsync(c->disp, normals, n * 8, hipMemcpyHostToDevice, c->stream));
    if (pinned_staging) {
        HIPOK(c, hipEventRecord(c->noise_ev, c->stream));
        c->noise_busy = true;
    } else {
        HIPOK(c, hipStreamSynchronize(c->stream));       // the caller's buffer may be pageable and short-lived
    }
    GaussDispArgs a{};
    gauss_args(c, mi, a, f);
    hipLaunchKernelGGL(k_gauss_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
    HIPOK(c, hipGetLastError());
    return 0;
}

// native mode: the factor is one Philox draw per step, the normals are generated where they are stored
static int gauss_native_disp(emx_ctx* c, int mi, uint64_t step, const int32_t* col) {
    const emx_move_desc& mv = c->moves[mi];
    double f = 1.0;
    if (mv.a != 0.0) {
        const Philox4 r = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), 0x46414354u /*'FACT'*/, 0, (uint32_t)c->ph_seed,
                                        (uint32_t)(c->ph_seed >> 32));
        f = std::exp(-mv.g0 + 2.0 * mv.g0 * u53(r.v[0], r.v[1]));
    }
    c->gfac = f;
    if (!c->tune_gauss_materialize) return 0;      // the half-step kernel generates the rows in registers
    GaussDispArgs a{};
    gauss_args(c, mi, a, f);
    a.col = col;
    a.seed = c->ph_seed;
    a.step = step;
    const int64_t nthreads = mv.reserved == EMX_GAUSS_VECTOR ? c->N * (int64_t)((c->D + 1) / 2) : c->N;
    hipLaunchKernelGGL(k_gauss_disp, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream, a);
    HIPOK(c, hipGetLastError());
    return 0;
}

static int step_begin_impl(emx_ctx* c, int32_t store, int32_t forced_move, int32_t* move_out, int32_t* S_out);
static bool small_eligible(const emx_ctx* c);

// ---- exact-mode plan pipeline (emx_mtpipe.hpp) -----------------------------------------------------------------
// retire completed uploads: the pinned staging buffer of such a step may be rewritten by the tokenizer
static void pipe_poll(void* arg) {
    emx_ctx* c = (emx_ctx*)arg;
    while (!c->pipe_uploads.empty()) {
        const int64_t n = c->pipe_uploads.front();
        if (c->cur.active && n == c->pipe_taken - 1) break;       // the open step's staging buffer may still be read (emx_plan_get)
        auto& s = c->ring[(c->pipe_ring0 + n % c->pipe_nsinks) % PLAN_RING];
        if (s.fetch_step >= 0) {
            if (!c->pipe_done || __atomic_load_n(c->pipe_done, __ATOMIC_ACQUIRE) <= (unsigned long long)s.fetch_step) break;
        } else if (!s.uploaded_ref || hipEventQuery(s.uploaded_ref) != hipSuccess) {
            break;
        }
        c->pipe->release(n);
        c->pipe_uploads.pop_front();
    }
}

static bool persist_exact_ok(const emx_ctx* c);
static bool regen_ctx_ok(const emx_ctx* c);
static int pipe_start(emx_ctx* c) {
    const int64_t nsteps = (int64_t)1 << 60;       // it runs ahead (16 plans at most) until something retires it
    const size_t N = (size_t)c->N;
    if (!c->up_stream) {
        // a queue of its own: streams of one priority share hardware queues, and an upload queued behind a persistent launch of
        // the consumer's stream would wait for all of its steps
        int lo = 0, hi = 0;
        HIPOK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPOK(c, hipStreamCreateWithPriority(&c->up_stream, hipStreamNonBlocking, hi));
    }
    HIPOK(c, hipStreamSynchronize(c->stream));          // no earlier copy still reads a staging buffer
    if (!c->pipe_done) {
        HIPOK(c, hipHostMalloc((void**)&c->pipe_done, 64, hipHostMallocDefault));
        HIPOK(c, hipMalloc((void**)&c->pipe_arrived, 64));
        HIPOK(c, hipMemset(c->pipe_arrived, 0, 64));
    }
    HIPOK(c, hipStreamSynchronize(c->up_stream));
    *c->pipe_done = 0ull;
    PlanSink sinks[PLAN_RING];
    c->pipe_nsinks = persist_exact_ok(c) ? PLAN_RING : PIPE_SINKS;      // (bursts of sixteen steps: the producers need the slack)
    c->pipe_ring0 = (c->ring_pos + 1) % PLAN_RING;
    for (int r = 0; r < c->pipe_nsinks; ++r) {
        auto& s = c->ring[(c->pipe_ring0 + r) % PLAN_RING];
        s.busy = false;
        s.last_seq = 0;
        s.fetch_step = -1;
        if (!s.host) HIPOK(c, hipHostMalloc((void**)&s.host, N * 32, hipHostMallocDefault));
        if (!s.uploaded) HIPOK(c, hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
        const HostPlan hp(s.host, N);
        sinks[r].order = hp.order;
        sinks[r].p0 = hp.p0;
        sinks[r].p1 = hp.p1;
        sinks[r].p2 = hp.p2;
        sinks[r].s0 = hp.s0;
        sinks[r].uacc = hp.uacc;
    }
    c->pipe_taken = 0;
    c->pipe_uploads.clear();
    // The ring of generator state words lives in pinned memory: with device finish (one replica whose plans nobody but the fused
    // kernels reads, step-at-a-time uploads) the fetch kernel reads a stretch step's uniforms straight out of it.
    // device finish (one replica whose plans nobody but the fused kernels reads, step-at-a-time uploads): the finishers pass a stretch
    // step's uniforms on as generator words, k_plan_raw converts them in place behind the upload
    // (round 6: the persistent launches' fetch takes raw / regen steps too -- where the ensemble's stretch steps will be regen steps,
    // regen_ctx_ok, the pipeline of a persistent consumer hands them over that way as well)
    const bool devfin = c->tune_mt_device_finish != 0 && (c->pipe_nsinks == PIPE_SINKS || regen_ctx_ok(c)) && c->world == 1 && !c->comm && !c->sendbuf &&
                        !c->peers_ready && c->target != EMX_TARGET_HOST && !c->tune_full_plan;
    c->pipe = new MtPlanPipeline(c->mt, c->N, c->D, (int32_t)c->moves.size(), c->moves.data(), c->cdf.data(), nsteps, sinks,
                                 c->pipe_nsinks, (int32_t)(c->tune_mt_pipeline > 0 ? c->tune_mt_pipeline : 0), false, devfin, c->pipe_nsinks == PLAN_RING,
                                 devfin ? c->tune_mt_regen_min : 0);
    return 0;
}

// stop the threads; the context's generator continues from the end of the last step taken
// (rc != 0: the device producer reports a stalled stage or a stream under-run -- the steps taken from it are void.  Status bit 4
// is raised with it and every API entry point that retires the producer hands the rc on: PIPE_STOP.)
static int mtdev_stop(emx_ctx* c);
static int pipe_stop(emx_ctx* c) {
    const int rc = mtdev_stop(c);
    if (!c->pipe) return rc;
    c->pipe->finish(c->pipe_taken, c->mt);
    delete c->pipe;
    c->pipe = nullptr;
    c->pipe_uploads.clear();
    c->pipe_deferred.clear();
    return rc;
}

// The pipeline is persistent: emx_run (or a single step of a large ensemble) starts it, it keeps running AHEAD of the steps
// taken -- plans do not depend on the walkers -- and serves later emx_run / emx_step_begin calls (the sample() generator takes
// one step per call) until something needs the generator state itself or changes what a plan is: emx_rng_get/set_mt19937,
// emx_set_moves, emx_set_rng_mode, a forced move, the one-workgroup path, emx_destroy.  Those retire it (pipe_stop): the
// context's generator is set to the state after the last step TAKEN, what was produced ahead is dropped.
// ---- exact-mode plans made on the device (emx_mtdev.hpp) ----------------------------------------------------------------
// Same life cycle as the host pipeline's: started by the first step that can use it, runs ahead of the steps taken (plans do not
// depend on the walkers), retired -- the context's generator set to the state after the last step TAKEN -- by whatever retires
// the pipeline (pipe_stop calls mtdev_stop).
static bool mtdev_eligible(const emx_ctx* c) {
    // Round 6: where the host pipeline hands its stretch steps over as generator states (regen_ctx_ok: half the ensemble a power of two)
    // it is the faster producer up to half a million walkers -- 131 072: 70 against 108 us/step, 262 144: 120-195 against 179,
    // 524 288: 250-350 against 315-360, 1 048 576: 760-800 against 524-571 (profiles/r06/mtdev_sizes_r06.txt) -- so the device
    // producer starts at "mt_device_min_walkers_regen" there
    const int64_t dmin = c->tune_mt_device == 2 ? 8192 : (regen_ctx_ok(c) ? std::max(c->tune_mt_device_min, c->tune_mt_device_min_regen) : c->tune_mt_device_min);
    return c->rng_mode == EMX_RNG_MT19937 && c->tune_mt_device != 0 && c->N >= dmin && c->world == 1 && !c->comm && !c->sendbuf &&
           !c->peers_ready && !small_eligible(c) && MtDevProducer::supports(c->N, (int32_t)c->moves.size(), c->moves.data());
}

static int mtdev_start(emx_ctx* c) {
    const size_t N = (size_t)c->N;
    MtDevPlanCols cols[MTDEV_SLOTS];
    for (int r = 0; r < MTDEV_SLOTS; ++r) {
        auto& s = c->ring[PLAN_RING + r];
        if (!s.order) {
            char* blk = nullptr;
            HIPOK(c, hipMalloc((void**)&blk, N * 48));
            s.order = (int32_t*)blk;
            s.p0 = s.order + N;
            s.s0 = (double*)(blk + N * 8);
            s.uacc = s.s0 + N;
            s.p1 = (int32_t*)(blk + N * 24);
            s.p2 = s.p1 + N;
            s.logu = (double*)(blk + N * 32);
            s.fac = s.logu + N;
        }
        s.busy = false;
        s.host_written = false;
        cols[r] = MtDevPlanCols{s.order, s.p0, s.s0, s.uacc, s.logu, s.fac};
    }
    HIPOK(c, hipStreamSynchronize(c->stream));          // no earlier kernel still reads one of the slots
    c->mtdev = new MtDevProducer(c->device, c->mt, c->N, c->D, c->moves[0], cols, c->status);
    if (!c->mtdev->ok()) {
        c->err = c->mtdev->error();
        delete c->mtdev;
        c->mtdev = nullptr;
        return -2;
    }
    c->mtdev->set_window_rule((int)c->tune_mt_tok_wshift, (int)c->tune_mt_tok_tail);
    c->mtdev_taken = 0;
    c->mtdev_starts++;
    return 0;
}

static int mtdev_stop(emx_ctx* c) {
    if (!c->mtdev) return 0;
    hipStreamSynchronize(c->stream);                    // the consumer's last reads of the plan slots
    MT19937Legacy after = c->mt;
    const int rc = c->mtdev->finish(c->mtdev_taken, after);
    if (rc == 0) {
        c->mt = after;
    } else {
        c->err = c->mtdev->error();                     // the run is void, loudly: the rc goes up, and the status bit stays
        __atomic_store_n(&c->status_host[__builtin_ctz(ST_PLAN_PRODUCER)], 1u, __ATOMIC_RELEASE);      // for whoever reads emx_status
    }
    c->mtdev_stats_last = c->mtdev->stats();
    c->mtdev_steps_total += c->mtdev_taken;
    delete c->mtdev;
    c->mtdev = nullptr;
    c->mtdev_taken = 0;
    return rc;
}

// emx_step_begin's part: the next plan is (or will be, in stream order) in its slot
static int mtdev_take(emx_ctx* c) {
    auto& cur = c->cur;
    const int64_t n = c->mtdev_taken;
    if (n % MTDEV_BATCH == 0) {
        const int rc = c->mtdev->ensure_batch(n / MTDEV_BATCH, c->stream, (int)c->tune_mt_lookahead);
        if (rc) FAIL(c, rc, "%s", c->mtdev->error().c_str());
    }
    const emx_move_desc& mv = c->moves[0];
    cur.move = 0;
    cur.S = mv.nsplits;
    cur.slot = PLAN_RING + (int)(n % MTDEV_SLOTS);
    cur.devplan = true;
    cur.off.assign(cur.S + 1, 0);
    for (int s = 0; s < cur.S; ++s) cur.off[s + 1] = cur.off[s] + (int32_t)((c->N - s + cur.S - 1) / cur.S);   // label counts survive the shuffle
    c->mtdev_taken = n + 1;
    return 0;
}

static bool persist_exact_ok(const emx_ctx* c);
// the stretch steps of this context's host pipeline will be regen steps (emx_mtpipe.cpp, tokenize): device finish on, an ensemble of
// "mt_regen_min_walkers" or more whose half is a power of two
static bool regen_ctx_ok(const emx_ctx* c) {
    const int64_t h = c->N / 2;
    return c->tune_mt_device_finish != 0 && c->tune_mt_regen_min > 0 && c->N >= c->tune_mt_regen_min && (c->N % 2) == 0 && h >= 2 && (h & (h - 1)) == 0 &&
           c->world == 1 && !c->comm && !c->sendbuf && !c->peers_ready && c->target != EMX_TARGET_HOST && !c->tune_full_plan;
}
static bool pipe_eligible(const emx_ctx* c) {
    return c->rng_mode == EMX_RNG_MT19937 && c->tune_mt_pipeline != 0 && (!small_eligible(c) || persist_exact_ok(c)) &&
           MtPlanPipeline::supports((int32_t)c->moves.size(), c->moves.data());
}

// emx_step_begin's part: wait for the next plan, send it up on the upload stream, order the kernels behind it
static inline double tr_now() { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int pipe_take(emx_ctx* c) {
    auto& cur = c->cur;
    const int64_t n = c->pipe_taken;
    PipeStepInfo info;
    pipe_poll(c);
    if (!c->pipe->wait_ready(n, info, pipe_poll, c)) FAIL(c, -7, "exact-mode plan pipeline stopped before step %lld", (long long)n);
    const int slot = (int)((c->pipe_ring0 + n % c->pipe_nsinks) % PLAN_RING);
    auto& s = c->ring[slot];
    cur.move = info.move;
    cur.S = info.S;
    cur.slot = slot;
    cur.off.assign(info.off, info.off + info.S + 1);
    s.move_idx = info.move;
    const size_t N = (size_t)c->N;
    if (c->pipe_defer) {
        // run_persist: the launch's plans go up together (pipe_fetch_deferred).  (Round 5: k_plan_fetch read FINISHED columns only and a
        // raw step here was an error; round 6: the fetch takes raw and regen steps as they are and the batched k_plan_regen / k_plan_raw
        // behind it finish them -- which is what lets 65 536 walkers take the persistent kernel in exact mode.)
        s.pinfo = info;                     // (a raw / regen step: the fetch copies it as it is and finishes it behind the copy)
        cur.devplan = info.raw != 0;
        s.uploaded_ref = nullptr;
        s.fetch_step = n;
        s.host_written = true;
        c->pipe_deferred.push_back({n, slot});
        c->pipe_uploads.push_back(n);
        c->pipe_taken = n + 1;
        c->ring_pos = slot;
        return 0;
    }
    // the device copy of this slot was last read by the kernels of step n - PIPE_SINKS.  (Asked of the event first: it is over long
    // ago, and a stream wait on this runtime orders behind the other stream's LATEST work -- the kernels just enqueued.)
    const bool still_read = s.busy && hipEventQuery(s.consumed_ref) != hipSuccess;
    if (still_read) HIPOK(c, hipStreamWaitEvent(c->up_stream, s.consumed_ref, 0));
    const int stretch = c->moves[cur.move].kind == EMX_MOVE_STRETCH;
    // (a regen step: `order` and, in the p0 column's place right behind it, the generator states -- PipeStepInfo::regen)
    const size_t bytes = info.regen ? ((N * 4 + (size_t)info.regen_nseg * 624 * 4 + 255) & ~(size_t)255)
                                    : plan_upload_bytes(N, stretch && c->world == 1 ? EMX_MOVE_STRETCH : EMX_MOVE_DE);
    // The upload stream carries copies only: the conversion kernel behind a copy made every step's upload wait for a compute unit
    // the half-step kernels hold (54 us per step of 65 536 walkers whatever the pipeline did; profiles/r05/exact_c2.md) -- it now
    // runs on the consumer's stream, in front of the half-steps that need it.  (A large plan in two halves on two streams -- one
    // copy engine moves 1.57 MB in 37 us, two in 30, profiles/r05/h2d_rate.txt -- was slower end to end and is gone: round 6.)
    HIPOK(c, hipMemcpyAsync(s.order, s.host, bytes, hipMemcpyHostToDevice, c->up_stream));
    // (k_plan_regen behind the copy on the upload stream, under the step before -- its 30-register workgroups fit beside a half-step
    // kernel's -- was measured: 43.8-45.4 against 50.0-50.3 us/step at 65 536 walkers in one session, 78-81 against 69-71 at 131 072 in
    // another; profiles/r06/exact_regen_side.txt, exact_persist_regen.txt.  Not kept: up to 65 536 walkers the persistent launches'
    // batched finish is the path anyway.)
    PlanRegenArgs G{};
    if (info.regen) {
        G.dev = reinterpret_cast<char*>(s.order);
        G.N = (int32_t)N;
        G.ns0 = info.off[1] - info.off[0];
        G.off = info.regen_off;
        G.nseg = info.regen_nseg;
    }
    PlanRawArgs R{};
    if (info.raw) {
        R.dev = reinterpret_cast<char*>(s.order);
        R.a = c->moves[cur.move].a;
        R.N = (int32_t)N;
        R.D = c->D;
        R.S = info.S;
        R.wr_words = info.wr_ring;
        R.wr_p1 = info.regen ? 1 : 0;
        for (int k = 0; k <= info.S; ++k) R.off[k] = info.off[k];
    }
    HIPOK(c, hipEventRecord(s.uploaded, c->up_stream));
    HIPOK(c, hipStreamWaitEvent(c->stream, s.uploaded, 0));
    if (info.raw) {
        // device finish: the columns hold `order` and generator words (or accepted randint values); converted in place -- a regen
        // step's fixed-length draws are first made again from the generator states the upload brought (k_plan_regen)
        if (info.regen) {
            hipLaunchKernelGGL(k_plan_regen, dim3((unsigned)info.regen_nseg), dim3(256), 0, c->stream, G);
            c->pipe_regen_steps++;
        }
        hipLaunchKernelGGL(k_plan_raw, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, R);
        c->pipe_raw_steps++;
        cur.devplan = true;                 // (emx_plan_get: the finished columns exist on the device only)
    } else {
        hipLaunchKernelGGL(k_plan_logs, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (int)N, (int)c->D, stretch, s.s0,
                           s.uacc, s.logu, s.fac);
    }
    HIPOK(c, hipGetLastError());
    s.uploaded_ref = s.uploaded;
    s.fetch_step = -1;
    s.host_written = true;
    c->pipe_uploads.push_back(n);
    c->pipe_taken = n + 1;
    c->ring_pos = slot;
    return 0;
}

// run_persist, exact mode: every plan taken since the last fetch goes up in one launch on the upload stream -- read straight from
// the pinned staging buffers, logs included -- behind the kernels that last read those slots; the consumer's stream waits for it.
static int pipe_fetch_deferred(emx_ctx* c) {
    if (c->pipe_deferred.empty()) return 0;
    NEED(c, c->pipe_deferred.size() <= 16, "more deferred plans than one fetch takes");
    PlanFetchArgs F{};
    PlanRawBatchArgs RB{};               // (a step handed over finished keeps N = 0 / nseg = 0: its blocks of the batched kernels return at once)
    PlanRegenBatchArgs GB{};
    bool any_raw = false;
    int max_nseg = 0;
    F.N = (int32_t)c->N;
    F.D = c->D;
    F.n = (int)c->pipe_deferred.size();
    // The device copies of these slots were last read by the launch four back.  An event wait for it (hipStreamWaitEvent on the upload
    // stream) was measured to hold the fetch until the launch RUNNING now has ended -- the runtime learns of completions lazily and
    // then orders behind the consumer stream's latest work -- which serialises fetch and kernels.  The kernels say it themselves
    // instead: a launch leaves its number in a pinned word when it starts (PersistArgs::started_host), and a later launch having
    // started means this one is over (one in-order stream).  Normally true long ago; otherwise the host waits here, which also
    // bounds how far it runs ahead of the device.
    hipEvent_t waited = nullptr;
    for (int k = 0; k < F.n; ++k) {
        auto& s = c->ring[c->pipe_deferred[k].second];
        bool by_event = s.busy && s.consumed_ref != waited;
        if (s.busy && s.last_seq != 0u && c->persist_started && (int)(c->persist_seq - s.last_seq) > 0) {
            const double w0 = tr_now();
            const double limit = 2e6 * (double)std::max<int64_t>(1, c->tune_persist_timeout_ms);
            while ((int)(__atomic_load_n(c->persist_started, __ATOMIC_ACQUIRE) - s.last_seq) <= 0 && tr_now() - w0 < limit) {}
            if ((int)(__atomic_load_n(c->persist_started, __ATOMIC_ACQUIRE) - s.last_seq) > 0) by_event = false;      // (else: launches that gave up -- the event)
        }
        if (by_event) {          // (the slots of one earlier launch share its event)
            HIPOK(c, hipStreamWaitEvent(c->up_stream, s.consumed_ref, 0));
            waited = s.consumed_ref;
        }
        F.host[k] = s.host;
        F.dev[k] = (char*)s.order;
        F.stretch[k] = c->moves[(size_t)std::max(0, s.move_idx)].kind == EMX_MOVE_STRETCH;       // (the step's own move: a mixture's steps share launches)
        F.peers[k] = !(F.stretch[k] && c->world == 1);
        const PipeStepInfo& pi = s.pinfo;
        F.kind[k] = pi.regen ? 2 : (pi.raw ? 1 : 0);
        F.nkey[k] = pi.regen ? pi.regen_nseg * 624 : 0;
        if (pi.raw) {
            any_raw = true;
            PlanRawArgs& R = RB.st[k];
            R.dev = (char*)s.order;
            R.a = c->moves[(size_t)std::max(0, s.move_idx)].a;
            R.N = (int32_t)c->N;
            R.D = c->D;
            R.S = pi.S;
            R.wr_words = pi.wr_ring;
            R.wr_p1 = pi.regen ? 1 : 0;
            for (int q = 0; q <= pi.S && q <= PLAN_RAW_SPLITS; ++q) R.off[q] = pi.off[q];
            c->pipe_raw_steps++;
        }
        if (pi.regen) {
            PlanRegenArgs& G = GB.st[k];
            G.dev = (char*)s.order;
            G.N = (int32_t)c->N;
            G.ns0 = pi.off[1] - pi.off[0];
            G.off = pi.regen_off;
            G.nseg = pi.regen_nseg;
            max_nseg = std::max(max_nseg, pi.regen_nseg);
            c->pipe_regen_steps++;
        }
    }
    F.delay_ticks = (unsigned)(c->tune_fetch_delay_us * 100);
    F.arrived = c->pipe_arrived;
    F.avoid_xcc = (c->persist_bar && c->tune_fetch_avoid) ? c->persist_bar + 9 * PERSIST_BAR_STRIDE + 4 : nullptr;
    F.host_done = c->pipe_done;
    F.done_value = (unsigned long long)(c->pipe_deferred.back().first + 1);
    F.pieces_x = (int32_t)((c->N + 255) / 256);
    unsigned nwg = (unsigned)F.pieces_x * (unsigned)F.n;
    if (!c->pipe_fetch_local && c->tune_fetch_blocks > 0) nwg = std::min<unsigned>(nwg, (unsigned)c->tune_fetch_blocks);
    hipLaunchKernelGGL(k_plan_fetch, dim3(nwg), dim3(256), 0, c->up_stream, F);
    // raw / regen steps: made into plans behind the copy, for all steps of the launch at once (30 registers a lane: these workgroups
    // find room beside a running persistent launch's)
    if (max_nseg > 0) hipLaunchKernelGGL(k_plan_regen_batch, dim3((unsigned)max_nseg, (unsigned)F.n), dim3(256), 0, c->up_stream, GB);
    if (any_raw) hipLaunchKernelGGL(k_plan_raw_batch, dim3((unsigned)F.pieces_x, (unsigned)F.n), dim3(256), 0, c->up_stream, RB);
    HIPOK(c, hipGetLastError());
    hipEvent_t& ev = c->pipe_batch_ev[c->pipe_batch_n & 3];
    if (!ev) HIPOK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->pipe_batch_n++;
    HIPOK(c, hipEventRecord(ev, c->up_stream));
    HIPOK(c, hipStreamWaitEvent(c->stream, ev, 0));
    for (auto& d : c->pipe_deferred) {
        c->ring[d.second].uploaded_ref = ev;
    }
    c->pipe_deferred.clear();
    return 0;
}

int emx_step_begin(emx_ctx* c, int32_t store, int32_t* move_out, int32_t* S_out) {
    return step_begin_impl(c, store, -1, move_out, S_out);
}

int emx_step_begin_with(emx_ctx* c, int32_t store, int32_t move_index, int32_t* S_out) {
    NEED(c, move_index >= 0 && move_index < (int)c->moves.size(), "bad move index");
    return step_begin_impl(c, store, move_index, nullptr, S_out);
}

// Native (Philox) plans of `nb` consecutive steps from `first_step`, evaluated full width by ONE launch on stream `st`, appended
// to c->prepared in step order.  (Measured and dropped, rounds 2 and 3: the NEXT batch on a second, low-priority stream next to
// this batch's half-steps, with and without raised wave priority for the half-step kernel -- the co-resident plan waves slow
// every half-step launch by 0.8 us: C2 23.69 -> 24.40 us/step, C3 38.65 -> 39.5; profiles/r03/ab_side_stream.txt.)
static int native_prepare_batch(emx_ctx* c, uint64_t first_step, int nb, int forced_move, hipStream_t st) {
    const int nm = (int)c->moves.size();
    NativeBatchArgs B{};
    B.N = (int32_t)c->N;
    B.D = c->D;
    B.nb = nb;
    // single replica, fused device target: nobody but the half-step kernel reads these plans
    B.lean = (c->target != EMX_TARGET_HOST && c->world == 1 && !c->sendbuf && !c->comm && !c->tune_full_plan) ? 1 : 0;
    B.ablate = (int32_t)(c->tune_ablate >> 8);
    bool only_stretch = true;
    for (int b = 0; b < nb; ++b) {
        const uint64_t step = first_step + (uint64_t)b;
        const int mi = forced_move >= 0 ? forced_move : philox_move_choice(c->ph_seed, step, c->cdf.data(), nm);
        const emx_move_desc& m = c->moves[mi];
        only_stretch = only_stretch && m.kind == EMX_MOVE_STRETCH;
        emx_ctx::PlanSlot* ps;
        int rc = acquire_slot(c, &ps, false);
        if (rc) return rc;
        emx_ctx::Prepared pr{};
        pr.move = mi;
        pr.S = m.nsplits;
        pr.slot = c->ring_pos;
        pr.step = step;
        pr.nat.seed = c->ph_seed;
        pr.nat.step = step;
        pr.nat.pk = make_perm_key((uint64_t)c->N, c->ph_seed, step);
        pr.cursor_before = m.gammas;
        pr.lean = B.lean != 0;
        if (m.kind == EMX_MOVE_GAUSS) {
            B.gmode[b] = m.reserved;
            B.gcol[b] = (int32_t)((int64_t)m.gammas % c->D);
            pr.gcol = B.gcol[b];
            if (m.reserved == EMX_GAUSS_SEQUENTIAL) c->moves[mi].gammas = (double)(((int64_t)m.gammas + 1) % c->D);
        }
        c->prepared.push_back(pr);
        B.nat[b] = pr.nat;
        B.order[b] = ps->order;
        B.p0[b] = ps->p0;
        B.p1[b] = ps->p1;
        B.p2[b] = ps->p2;
        B.s0[b] = ps->s0;
        B.uacc[b] = ps->uacc;
        B.logu[b] = ps->logu;
        B.fac[b] = ps->fac;
        B.a[b] = m.a;
        B.sigma[b] = m.sigma;
        B.g0[b] = m.g0;
        B.move[b] = m.kind;
        B.S[b] = m.nsplits;
    }
    if (only_stretch && B.lean && !B.ablate && !B.desc)       // (the same arithmetic without the other moves' branches: 32 VGPRs against 80)
        hipLaunchKernelGGL(k_native_plan_batch_stretch, dim3((unsigned)((c->N + 255) / 256), (unsigned)nb), dim3(256), 0, st, B);
    else
        hipLaunchKernelGGL(k_native_plan_batch, dim3((unsigned)((c->N + 255) / 256), (unsigned)nb), dim3(256), 0, st, B);
    HIPOK(c, hipGetLastError());
    return 0;
}

static int step_begin_impl(emx_ctx* c, int32_t store, int32_t forced_move, int32_t* move_out, int32_t* S_out) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, !c->cur.active, "emx_step_begin: previous step not ended");
    if (store) NEED(c, c->stored < c->cap, "chain capacity exhausted (call emx_chain_config)");
    auto& cur = c->cur;
    cur.store = store != 0;
    cur.native = false;
    cur.devplan = false;
    const int nm = (int)c->moves.size();
    const bool devp = forced_move < 0 && mtdev_eligible(c);
    if (c->mtdev && !devp) {
        const int rcm = mtdev_stop(c);
        if (rcm) return rcm;
    }
    if (c->pipe && (devp || forced_move >= 0 || !pipe_eligible(c))) PIPE_STOP(c);      // a forced move skips the choice draw: inline
    if (devp) {
        // exact mode, one stretch move: the plan of this step was (or is being) made on the device, from the same stream
        if (!c->mtdev) {
            int rc0 = mtdev_start(c);
            if (rc0) return rc0;
        }
        int rc = mtdev_take(c);
        if (rc) return rc;
    } else if (c->rng_mode == EMX_RNG_MT19937 && forced_move < 0 && (c->pipe || ((c->N >= 8192 || persist_exact_ok(c)) && pipe_eligible(c)))) {
        // exact mode: the plan of this step comes from the pipeline threads (same draws, same order as the inline producer)
        if (!c->pipe) {
            int rc0 = pipe_start(c);
            if (rc0) return rc0;
        }
        int rc = pipe_take(c);
        if (rc) return rc;
    } else if (c->rng_mode == EMX_RNG_MT19937) {
        cur.move = forced_move >= 0 ? forced_move : c->mt.choice_cdf(c->cdf.data(), nm);   // ensemble.py:406
        const emx_move_desc& mv = c->moves[cur.move];
        cur.S = mv.nsplits;
        NEED(c, c->N >= 2 && (mv.kind != EMX_MOVE_DE || c->N - (c->N + cur.S - 1) / cur.S >= 2),
             "complement too small for this move");
        emx_ctx::PlanSlot* ps;
        int rc = acquire_slot(c, &ps, true);
        if (rc) return rc;
        cur.slot = c->ring_pos;
        cur.off.assign(cur.S + 1, 0);
        const size_t N = (size_t)c->N;
        const HostPlan hp(ps->host, N);
        if (mv.kind == EMX_MOVE_GAUSS) {
            // the normals go through one pinned buffer: wait until the previous step's copy has left it
            if (!c->noise_host) {
                HIPOK(c, hipHostMalloc((void**)&c->noise_host, N * (size_t)c->D * 8, hipHostMallocDefault));
                HIPOK(c, hipEventCreateWithFlags(&c->noise_ev, hipEventDisableTiming));
            }
            if (c->noise_busy) {
                HIPOK(c, hipEventSynchronize(c->noise_ev));
                c->noise_busy = false;
            }
            const double f = make_exact_gauss(c->mt, c->N, c->D, mv, c->moves[cur.move].gammas, cur.off.data(), hp.order, hp.p0,
                                              hp.p1, hp.p2, hp.s0, hp.uacc, c->noise_host);
            rc = upload_plan(c, *ps);
            if (rc) return rc;
            rc = gauss_upload_normals(c, cur.move, c->noise_host, f, true);
            if (rc) return rc;
        } else {
            rc = make_exact_plan(c->mt, c->N, c->D, mv, c->labels_scratch, cur.off.data(), hp.order, hp.p0, hp.p1, hp.p2, hp.s0,
                                 hp.uacc);
            NEED(c, rc == 0, "plan generation failed");
            rc = upload_plan(c, *ps);
            if (rc) return rc;
        }
    } else if (c->rng_mode == EMX_RNG_PHILOX) {
        if (forced_move >= 0) drop_prepared(c);
        if (!c->prepared.empty() && c->prepared.front().step != c->ph_step) drop_prepared(c);
        if (c->prepared.empty()) {
            // evaluate the plans of the next nb steps (both splits each) in one full-width launch
            int64_t nbw = forced_move >= 0 ? 1 : c->prep_hint;
            const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(nbw, NATIVE_BATCH_MAX));
            const int rc = native_prepare_batch(c, c->ph_step, nb, forced_move, c->stream);
            if (rc) return rc;
        }
        const emx_ctx::Prepared pr = c->prepared.front();
        c->prepared.pop_front();
        cur.move = pr.move;
        cur.S = pr.S;
        cur.native = true;
        cur.lean = pr.lean;
        cur.gcol = pr.gcol;
        cur.nat = pr.nat;
        cur.slot = pr.slot;
        cur.off.assign(cur.S + 1, 0);
        for (int s = 0; s < cur.S; ++s) cur.off[s + 1] = cur.off[s] + (int32_t)((c->N - s + cur.S - 1) / cur.S);
        if (c->moves[cur.move].kind == EMX_MOVE_GAUSS) {
            int rc = gauss_native_disp(c, cur.move, pr.step, c->ring[pr.slot].p0);
            if (rc) return rc;
        }
    } else {
        // INPUTS: emx_plan_set must follow
        cur.move = -1;
        cur.S = 0;
        cur.slot = -1;
    }
    cur.active = true;
    if (move_out) *move_out = cur.move;
    if (S_out) *S_out = cur.S;
    return 0;
}

// A lean native plan carries only the columns the fused half-step kernel of the step's move reads (k_native_plan_batch): whoever
// else reads the plan of the open step -- emx_plan_get, the split-phase accept (uacc), the pull and direct exchanges' compact
// plans -- first has the same kernel evaluate the step once more with every column (the plan is a pure function of
// (seed, step, walker), so the columns already consumed do not change).
static int complete_lean_plan(emx_ctx* c) {
    auto& cur = c->cur;
    if (!cur.active || !cur.native || !cur.lean) return 0;
    NEED(c, cur.move >= 0 && cur.slot >= 0, "no plan available");
    auto& ps = c->ring[cur.slot];
    const emx_move_desc& m = c->moves[cur.move];
    NativeBatchArgs B{};
    B.N = (int32_t)c->N;
    B.D = c->D;
    B.nb = 1;
    B.lean = 0;
    B.nat[0] = cur.nat;
    B.order[0] = ps.order;
    B.p0[0] = ps.p0;
    B.p1[0] = ps.p1;
    B.p2[0] = ps.p2;
    B.s0[0] = ps.s0;
    B.uacc[0] = ps.uacc;
    B.logu[0] = ps.logu;
    B.fac[0] = ps.fac;
    B.a[0] = m.a;
    B.sigma[0] = m.sigma;
    B.g0[0] = m.g0;
    B.move[0] = m.kind;
    B.S[0] = m.nsplits;
    if (m.kind == EMX_MOVE_GAUSS) {
        B.gmode[0] = m.reserved;
        B.gcol[0] = cur.gcol;
    }
    hipLaunchKernelGGL(k_native_plan_batch, dim3((unsigned)((c->N + 255) / 256), 1u), dim3(256), 0, c->stream, B);
    HIPOK(c, hipGetLastError());
    cur.lean = false;
    return 0;
}

int emx_plan_set(emx_ctx* c, int32_t move_index, const int32_t* off, const int32_t* order, const int32_t* p0,
                 const int32_t* p1, const int32_t* p2, const double* s0, const double* uacc) {
    NEED(c, c->cur.active, "emx_plan_set outside a step");
    NEED(c, move_index >= 0 && move_index < (int)c->moves.size(), "bad move index");
    auto& cur = c->cur;
    cur.move = move_index;
    cur.S = c->moves[move_index].nsplits;
    cur.native = false;
    cur.off.assign(off, off + cur.S + 1);
    NEED(c, cur.off[0] == 0 && cur.off[cur.S] == c->N, "plan offsets must cover all walkers");
    emx_ctx::PlanSlot* ps;
    int rc = acquire_slot(c, &ps, true);
    if (rc) return rc;
    cur.slot = c->ring_pos;
    const size_t N = (size_t)c->N;
    const HostPlan hp(ps->host, N);
    memcpy(hp.order, order, N * 4);
    memcpy(hp.p0, p0, N * 4);
    memcpy(hp.p1, p1 ? p1 : order, N * 4);
    memcpy(hp.p2, p2 ? p2 : order, N * 4);
    if (s0) memcpy(hp.s0, s0, N * 8); else memset(hp.s0, 0, N * 8);
    memcpy(hp.uacc, uacc, N * 8);
    return upload_plan(c, *ps);
}

int emx_plan_get(emx_ctx* c, int32_t* off, int32_t* order, int32_t* p0, int32_t* p1, int32_t* p2, double* s0,
                 double* uacc) {
    NEED(c, c->cur.active, "emx_plan_get outside a step");
    auto& cur = c->cur;
    const size_t N = (size_t)c->N;
    memcpy(off, cur.off.data(), (cur.S + 1) * 4);
    if (cur.native) {
        // the plan was evaluated on the device by k_native_plan at emx_step_begin
        NEED(c, cur.slot >= 0, "no plan available");
        auto& ps = c->ring[cur.slot];
        {
            const int rcl = complete_lean_plan(c);
            if (rcl) return rcl;
        }
        HIPOK(c, hipMemcpyAsync(order, ps.order, N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipMemcpyAsync(p0, ps.p0, N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipMemcpyAsync(p1, ps.p1, N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipMemcpyAsync(p2, ps.p2, N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipMemcpyAsync(s0, ps.s0, N * 8, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipMemcpyAsync(uacc, ps.uacc, N * 8, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    NEED(c, cur.slot >= 0, "no plan available");
    auto& ps = c->ring[cur.slot];
    if (cur.devplan) {
        // made on the device (emx_mtdev.hpp): no host copy exists; a stretch plan's second and third partner are the walker itself
        HIPOK(c, hipMemcpyAsync(order, ps.order, N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipMemcpyAsync(p0, ps.p0, N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipMemcpyAsync(s0, ps.s0, N * 8, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipMemcpyAsync(uacc, ps.uacc, N * 8, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipStreamSynchronize(c->stream));
        memcpy(p1, order, N * 4);
        memcpy(p2, order, N * 4);
        return 0;
    }
    const HostPlan hp(ps.host, N);
    memcpy(order, hp.order, N * 4);
    memcpy(p0, hp.p0, N * 4);
    if (cur.move >= 0 && c->moves[cur.move].kind == EMX_MOVE_STRETCH) {     // one partner: the pipeline leaves these columns alone
        memcpy(p1, hp.order, N * 4);
        memcpy(p2, hp.order, N * 4);
    } else {
        memcpy(p1, hp.p1, N * 4);
        memcpy(p2, hp.p2, N * 4);
    }
    memcpy(s0, hp.s0, N * 8);
    memcpy(uacc, hp.uacc, N * 8);
    return 0;
}

int emx_plan_set_noise(emx_ctx* c, const double* normals, double factor) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, cur.active && cur.move >= 0 && c->moves[cur.move].kind == EMX_MOVE_GAUSS,
         "emx_plan_set_noise follows emx_plan_set of a Gaussian move");
    return gauss_upload_normals(c, cur.move, normals, factor, false);
}

static int do_halfstep(emx_ctx* c, int split, int target) {
    auto& cur = c->cur;
    NEED(c, cur.active && cur.move >= 0, "half-step outside a planned step");
    NEED(c, split >= 0 && split < cur.S, "split out of range");
    const emx_move_desc& mv = c->moves[cur.move];
    const int pos0 = cur.off[split], ns = cur.off[split + 1] - cur.off[split];
    int64_t lo = 0, hi = ns;
    if (c->world > 1) shard_range(ns, c->rank, c->world, lo, hi);
    double *chain = nullptr, *chain_lp = nullptr;
    if (cur.store) {
        chain = c->chain + (size_t)c->stored * c->N * c->D;
        chain_lp = c->chain_lp + (size_t)c->stored * c->N;
    }
    emx_ctx::PlanSlot* ps = cur.slot >= 0 ? &c->ring[cur.slot] : nullptr;
    NEED(c, c->exchange != EMX_EXCHANGE_PULL || c->world == 1 || target == EMX_TARGET_HOST,
         "pull exchange: use emx_pull_prepare / emx_pull_apply");
    NEED(c, c->exchange != EMX_EXCHANGE_DIRECT || c->world == 1 || target == EMX_TARGET_HOST,
         "direct exchange: use emx_direct_halfstep");
    NEED(c, c->exchange != EMX_EXCHANGE_LOGPROB || c->world == 1 || target == EMX_TARGET_HOST,
         "log-prob exchange: use emx_logprob_begin / emx_logprob_finish");
    NEED(c, c->exchange != EMX_EXCHANGE_REPLAY || c->world == 1 || target == EMX_TARGET_HOST,
         "replay exchange: use emx_replay_begin / emx_replay_finish");
    double* sb = nullptr;
    if (c->sendbuf && target != EMX_TARGET_HOST && c->exchange == EMX_EXCHANGE_ALLGATHER) sb = c->sendbuf;
    NEED(c, !sb || hi - lo <= c->sendbuf_rows, "exchange buffers too small for this move: call emx_set_shard after emx_set_moves");
    return launch_split(c, mv.kind, target, cur.S, split, pos0, ns, (int)lo, (int)hi, &mv, ps,
                        nullptr, c->X, c->lp, chain, chain_lp, sb);
}

int emx_halfstep(emx_ctx* c, int32_t split) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, c->target != EMX_TARGET_HOST, "emx_halfstep needs a device target (use emx_propose/emx_accept)");
    return do_halfstep(c, split, c->target);
}

int emx_propose(emx_ctx* c, int32_t split, double* q_out, double* factors_out, int64_t* ns_out) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, c->world == 1, "split-phase host targets are single-rank");
    int rc = do_halfstep(c, split, EMX_TARGET_HOST);
    if (rc) return rc;
    const auto& cur = c->cur;
    const int64_t ns = cur.off[split + 1] - cur.off[split];
    if (ns_out) *ns_out = ns;
    const size_t qb = q_out && ns > 0 ? (size_t)ns * c->D * 8 : 0, fb = factors_out && ns > 0 ? (size_t)ns * 8 : 0;
    if (qb + fb > 0 && qb + fb <= (4u << 20)) {
        if (c->xfer_bytes < qb + fb) {
            if (c->xfer_host) hipHostFree(c->xfer_host);
            c->xfer_host = nullptr;
            c->xfer_bytes = 0;
            const size_t want = std::max<size_t>(qb + fb, std::min<size_t>(4u << 20, (size_t)c->N * (c->D + 1) * 8));
            HIPOK(c, hipHostMalloc((void**)&c->xfer_host, want, hipHostMallocDefault));
            c->xfer_bytes = want;
        }
        if (qb) HIPOK(c, hipMemcpyAsync(c->xfer_host, c->qout, qb, hipMemcpyDeviceToHost, c->stream));
        if (fb) HIPOK(c, hipMemcpyAsync(c->xfer_host + qb, c->fout, fb, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipStreamSynchronize(c->stream));
        if (qb) memcpy(q_out, c->xfer_host, qb);
        if (fb) memcpy(factors_out, c->xfer_host + qb, fb);
        return 0;
    }
    if (fb) HIPOK(c, hipMemcpyAsync(factors_out, c->fout, fb, hipMemcpyDeviceToHost, c->stream));
    if (qb) {
        const int rc2 = big_copy_to_host(c, q_out, c->qout, qb);
        if (rc2) return rc2;
    }
    HIPOK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int emx_accept(emx_ctx* c, int32_t split, const double* new_lp) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, cur.active && cur.move >= 0, "emx_accept outside a planned step");
    NEED(c, split >= 0 && split < cur.S, "emx_accept: split %d out of range (the step has %d)", split, cur.S);
    const int pos0 = cur.off[split], ns = cur.off[split + 1] - cur.off[split];
    if (ns <= 0) return 0;
    {
        const int rcl = complete_lean_plan(c);          // k_accept reads uacc, which a lean plan does not carry
        if (rcl) return rcl;
    }
    HIPOK(c, hipMemcpyAsync(c->newlp, new_lp, (size_t)ns * 8, hipMemcpyHostToDevice, c->stream));
    AcceptArgs a{};
    a.X = c->X;
    a.lp = c->lp;
    a.acc = c->acc;
    a.acc_count = c->acc_count;
    if (cur.store) {
        a.chain = c->chain + (size_t)c->stored * c->N * c->D;
        a.chain_lp = c->chain_lp + (size_t)c->stored * c->N;
    }
    a.status = c->status;
    a.qout = c->qout;
    a.fout = c->fout;
    a.new_lp = c->newlp;
    a.order = cur.slot >= 0 ? c->ring[cur.slot].order : nullptr;
    a.uacc = cur.slot >= 0 ? c->ring[cur.slot].uacc : nullptr;
    a.N = (int32_t)c->N;
    a.D = c->D;
    a.S = cur.S;
    a.split = split;
    a.pos0 = pos0;
    a.ns = ns;
    a.move = c->moves[cur.move].kind;
    hipLaunchKernelGGL(k_accept, dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, c->stream, a);
    HIPOK(c, hipGetLastError());
    return 0;
}

int emx_accept_proposals(emx_ctx* c, int32_t split, const double* q, const double* factors, const double* new_lp) {
    // custom RedBlueMove.get_proposal (host code): the caller supplies q and factors (red_blue.py:90)
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, cur.active && cur.move >= 0 && split >= 0 && split < cur.S, "emx_accept_proposals outside a planned step");
    const int ns = cur.off[split + 1] - cur.off[split];
    if (ns <= 0) return 0;
    {
        const int rc = big_copy_to_device(c, c->qout, q, (size_t)ns * c->D * 8);
        if (rc) return rc;
    }
    HIPOK(c, hipMemcpyAsync(c->fout, factors, (size_t)ns * 8, hipMemcpyHostToDevice, c->stream));
    return emx_accept(c, split, new_lp);
}

int emx_step_end(emx_ctx* c) {
    auto& cur = c->cur;
    NEED(c, cur.active, "emx_step_end without emx_step_begin");
    if (cur.slot >= 0) {
        auto& s = c->ring[cur.slot];
        if (s.host_written && !c->pipe_defer) {      // (run_persist records it behind the launch that reads the slot)
            HIPOK(c, hipEventRecord(s.consumed, c->stream));
            s.consumed_ref = s.consumed;
            s.last_seq = 0;
            s.busy = true;
        }
    }
    if (cur.devplan && c->mtdev && c->mtdev_taken % MTDEV_BATCH == 0) {
        // the last step of a produced batch: its plan slots may be rewritten once the kernels enqueued so far have run
        if (c->mtdev_defer_release) {
            c->mtdev_release_pending = c->mtdev_taken / MTDEV_BATCH - 1;        // (run_persist: the launch is not enqueued yet)
        } else {
            const int rc = c->mtdev->release_batch(c->mtdev_taken / MTDEV_BATCH - 1, c->stream);
            if (rc) FAIL(c, rc, "%s", c->mtdev->error().c_str());
        }
    }
    c->direct_planned = false;
    if (cur.store) c->stored++;
    c->proposals++;
    if (c->rng_mode == EMX_RNG_PHILOX) c->ph_step++;
    cur.active = false;
    return 0;
}

static void graph_invalidate(emx_ctx* c) {
    for (auto& g : c->gslot) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
        g.exec = nullptr;
        g.graph = nullptr;
        g.valid = false;
    }
}

// Replay NATIVE_BATCH_MAX native steps as ONE hipGraph launch: [k_graph_advance, k_native_plan_batch,
// S half-steps x 8].  Returns the instantiated graph for this configuration (capturing it on first use),
// or nullptr when the ordinary launch path must be used.
static emx_ctx::GraphSlot* graph_ready(emx_ctx* c, int store) {
    constexpr int NB = NATIVE_BATCH_MAX;
    if (c->graph_disabled || !c->tune_graph || !c->graph_warm) return nullptr;
    if (c->rng_mode != EMX_RNG_PHILOX || c->moves.size() != 1 || c->world != 1 || c->comm || c->sendbuf) return nullptr;
    if (c->prof_max > 0 || c->tune_ablate || c->cur.active || !c->prepared.empty()) return nullptr;
    if (store && c->stored + NB > c->cap) return nullptr;
    auto& g = c->gslot[store ? 1 : 0];
    const emx_move_desc& mv = c->moves[0];
    if (mv.kind == EMX_MOVE_GAUSS) return nullptr;
    if (g.valid && (g.spw != c->tune_spw || g.wpb != c->tune_wpb || g.bpc != c->tune_bpc || g.target != c->target)) graph_invalidate(c);
    if (!c->d_desc) {
        if (hipMalloc((void**)&c->d_desc, sizeof(StepDesc) * NB) != hipSuccess ||
            hipMalloc((void**)&c->d_ctr, sizeof(GraphCounters)) != hipSuccess ||
            false) {
            c->graph_disabled = true;
            return nullptr;
        }
    }
    if (!g.valid) {
        const int S = mv.nsplits;
        std::vector<int32_t> off(S + 1, 0);
        for (int s = 0; s < S; ++s) off[s + 1] = off[s] + (int32_t)((c->N - s + S - 1) / S);
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            c->graph_disabled = true;
            return nullptr;
        }
        bool ok = true;
        hipLaunchKernelGGL(k_graph_advance, dim3(1), dim3(64), 0, c->stream, c->d_ctr, c->d_desc,
                           (unsigned long long)c->ph_seed, (int)c->N, NB, store);
        NativeBatchArgs B{};
        B.N = (int32_t)c->N;
        B.D = c->D;
        B.nb = NB;
        B.desc = c->d_desc;
        const int slot0 = PLAN_RING - NB;           // the upper half of the ring is reserved for the graph
        for (int b = 0; b < NB; ++b) {
            auto& ps = c->ring[slot0 + b];
            B.order[b] = ps.order;
            B.p0[b] = ps.p0;
            B.p1[b] = ps.p1;
            B.p2[b] = ps.p2;
            B.s0[b] = ps.s0;
            B.uacc[b] = ps.uacc;
            B.logu[b] = ps.logu;
            B.fac[b] = ps.fac;
            B.a[b] = mv.a;
            B.sigma[b] = mv.sigma;
            B.g0[b] = mv.g0;
            B.move[b] = mv.kind;
            B.S[b] = S;
        }
        hipLaunchKernelGGL(k_native_plan_batch, dim3((unsigned)((c->N + 255) / 256), (unsigned)NB), dim3(256), 0, c->stream, B);
        for (int b = 0; b < NB && ok; ++b)
            for (int sp = 0; sp < S && ok; ++sp) {
                const int rc = launch_split(c, mv.kind, c->target, S, sp, off[sp], off[sp + 1] - off[sp], 0, off[sp + 1] - off[sp],
                                            &mv, &c->ring[slot0 + b], nullptr, c->X, c->lp, nullptr, nullptr, nullptr,
                                            c->d_desc + b);
                ok = rc == 0;
            }
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (!ok || e != hipSuccess || !graph) {
            if (graph) hipGraphDestroy(graph);
            c->graph_disabled = true;
            (void)hipGetLastError();
            return nullptr;
        }
        hipGraphExec_t exec = nullptr;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            hipGraphDestroy(graph);
            c->graph_disabled = true;
            (void)hipGetLastError();
            return nullptr;
        }
        g.graph = graph;
        g.exec = exec;
        g.valid = true;
        g.spw = c->tune_spw;
        g.wpb = c->tune_wpb;
        g.bpc = c->tune_bpc;
        g.target = c->target;
    }
    return &g;
}

static int scatter_gathered(emx_ctx* c, int32_t split, int64_t block_rows);

// emx_small.hip: the k_small_run instantiations live in their own translation unit (compiled in parallel with this one)
hipError_t emx_small_dispatch(int G, int V, int CH, int dpb, int movesel, int threads, size_t lds, hipStream_t st,
                              const emx::SmallRunArgs& a);

// ---- small ensembles: whole runs inside one workgroup (k_small_run) ------------------------
// steps whose plans one pass evaluates: as many as give every thread of the workgroup an entry
static int small_batch(int64_t N) { return (int)std::max<int64_t>(1, std::min<int64_t>(64, 1024 / N)); }

// dense_dp > 0: + the Cholesky image and one 16-row tile per wave
static size_t small_lds_bytes(int64_t N, int D, int dense_dp = 0, int waves = 0) {
    const size_t B = (size_t)small_batch(N);
    size_t b = (size_t)N * ((size_t)D * 8 + 8 + 4 + 1) + B * (size_t)N * (3 * 8 + 4 * 4) + 64;
    if (dense_dp > 0) b += 16 + ((size_t)dense_img_doubles(dense_dp) + dense_dp + (size_t)waves * (16 * (dense_dp + 2) + 16)) * 8;
    return b;
}

// threads of the one workgroup: enough for one half-step's lanes and one plan entry each across the batch; the dense
// variant keeps one LDS tile per wave, so it takes the largest power-of-two wave count that still fits
static int small_threads(const emx_ctx* c, int G, int minsplits, bool dense) {
    const int64_t nsmax = (c->N + minsplits - 1) / minsplits;
    const int64_t want = std::max<int64_t>(dense ? ((nsmax + 15) / 16) * 64 : nsmax * G, (int64_t)small_batch(c->N) * c->N);
    int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, ((want + 63) / 64) * 64));
    if (dense) {
        int waves = threads / 64;
        while (waves > 1 && small_lds_bytes(c->N, c->D, c->Dp, waves) > 150 * 1024) waves = (waves + 1) / 2;
        threads = waves * 64;
    }
    return threads;
}

static bool small_eligible(const emx_ctx* c) {
    if (!c->tune_small || (c->rng_mode != EMX_RNG_PHILOX && c->rng_mode != EMX_RNG_MT19937)) return false;
    if (c->moves.empty() || (int)c->moves.size() > SMALL_MAX_MOVES) return false;
    for (const auto& mv : c->moves) {
        if (mv.kind == EMX_MOVE_GAUSS) {
            if (c->rng_mode != EMX_RNG_PHILOX) return false;      // exact mode: N x D host normals per step, general path
            continue;
        }
        if (mv.kind != EMX_MOVE_STRETCH && mv.kind != EMX_MOVE_DE && mv.kind != EMX_MOVE_SNOOKER) return false;
        if (mv.kind == EMX_MOVE_DE && c->N - (c->N + mv.nsplits - 1) / mv.nsplits < 2) return false;
    }
    if (c->target != EMX_TARGET_ISO_GAUSS && c->target != EMX_TARGET_DIAG_GAUSS && c->target != EMX_TARGET_ROSENBROCK &&
        c->target != EMX_TARGET_BOX && c->target != EMX_TARGET_DENSE_GAUSS)
        return false;
    if (c->world != 1 || c->comm || c->sendbuf || c->prof_max > 0 || c->tune_ablate || c->cur.active) return false;
    if (c->N > 4096 || c->D > 256) return false;
    if (c->target == EMX_TARGET_DENSE_GAUSS)     // one CU's matrix pipe: worth it only while the contraction is small
        return !dense_is_wide(c) && c->N * (int64_t)c->Dp * c->Dp <= 65536 && small_lds_bytes(c->N, c->D, c->Dp, 1) <= 150 * 1024;
    return small_lds_bytes(c->N, c->D) <= 150 * 1024;
}

// `nsteps` full steps starting at step index i0 of the current emx_run call
static int run_small(emx_ctx* c, int64_t i0, int64_t nsteps, int32_t thin_by, int32_t store) {
    PIPE_STOP(c);                 // this path draws from the context's generator itself
    const int nm = (int)c->moves.size();
    emx_ctx::BulkPlans* gauss_bulk = nullptr;
    const bool dense = c->target == EMX_TARGET_DENSE_GAUSS;
    const Shape sh = pick_shape(c->D, dense ? c->Dp : c->D);
    SmallRunArgs a{};
    int maxsplits = 2, minsplits = 64;
    bool any_gauss = false;
    for (int m = 0; m < nm; ++m) {
        const emx_move_desc& mv = c->moves[m];
        a.kind[m] = mv.kind;
        a.nsplits[m] = mv.nsplits;
        a.a[m] = mv.a;
        a.sigma[m] = mv.sigma;
        a.g0[m] = mv.g0;
        a.gammas[m] = mv.gammas;
        a.cdf[m] = c->cdf[m];
        a.gmode[m] = mv.reserved;
        a.gsigma[m] = mv.sigma;
        a.gscale[m] = m < (int)c->mscale.size() ? c->mscale[m] : nullptr;
        any_gauss = any_gauss || mv.kind == EMX_MOVE_GAUSS;
        maxsplits = std::max(maxsplits, (int)mv.nsplits);
        minsplits = std::min(minsplits, (int)mv.nsplits);
    }
    a.nmoves = nm;
    a.X = c->X;
    a.lp = c->lp;
    a.acc = c->acc;
    a.acc_count = c->acc_count;
    a.status = c->status;
    if (store) {
        a.chain = c->chain + (size_t)c->stored * c->N * c->D;
        a.chain_lp = c->chain_lp + (size_t)c->stored * c->N;
    }
    a.tp0 = c->tp0;
    a.tp1 = c->tp1;
    a.tscale = c->tscale;
    a.seed = c->ph_seed;
    a.step0 = c->ph_step;
    a.i0 = i0;
    a.N = (int32_t)c->N;
    a.D = c->D;
    a.target = c->target;
    a.nsteps = (int32_t)nsteps;
    a.thin_by = thin_by;
    a.store = store;
    a.batch = small_batch(c->N);
    if (c->rng_mode == EMX_RNG_MT19937) {
        // the reference's MT19937 stream, consumed on the host exactly as step_begin would (move choice, then the
        // step's draws), `nsteps` plans per copy; the other buffer may still be feeding the previous launch
        auto& bp = c->bulk[c->bulk_pos];
        c->bulk_pos ^= 1;
        const size_t plan_bytes = (size_t)nsteps * (size_t)c->N * 32;
        const size_t need = plan_bytes + (size_t)nsteps * 4;          // + the move index of every step
        if (bp.busy) {
            HIPOK(c, hipEventSynchronize(bp.done));
            bp.busy = false;
        }
        if (bp.bytes < need) {
            if (bp.host) hipHostFree(bp.host);
            if (bp.dev) hipFree(bp.dev);
            bp.host = bp.dev = nullptr;
            bp.bytes = 0;
            HIPOK(c, hipHostMalloc((void**)&bp.host, need, hipHostMallocDefault));
            HIPOK(c, hipMalloc((void**)&bp.dev, need));
            bp.bytes = need;
            if (!bp.done) HIPOK(c, hipEventCreateWithFlags(&bp.done, hipEventDisableTiming));
        }
        const size_t N = (size_t)c->N;
        std::vector<int32_t> off(maxsplits + 1);
        int32_t* step_moves = (int32_t*)(bp.host + plan_bytes);
        for (int64_t s2 = 0; s2 < nsteps; ++s2) {
            const int mi = c->mt.choice_cdf(c->cdf.data(), nm);             // ensemble.py:406
            step_moves[s2] = mi;
            const emx_move_desc& mv = c->moves[mi];
            int32_t* hi = (int32_t*)(bp.host + (size_t)s2 * N * 32);
            double* hd = (double*)(bp.host + (size_t)s2 * N * 32 + N * 16);
            const int rc = make_exact_plan(c->mt, c->N, c->D, mv, c->labels_scratch, off.data(), hi, hi + N, hi + 2 * N,
                                           hi + 3 * N, hd, hd + N);
            NEED(c, rc == 0, "plan generation failed");
        }
        HIPOK(c, hipMemcpyAsync(bp.dev, bp.host, need, hipMemcpyHostToDevice, c->stream));
        a.plans = bp.dev;
        a.step_moves = (const int32_t*)(bp.dev + plan_bytes);
    }
    if (any_gauss) {
        // per step of the launch: the step-size factor (one Philox draw, as gauss_native_disp) and the sequential mode's
        // column; both follow from the step number, so the host lists them and advances the move's cursor
        auto& bp = c->bulk[c->bulk_pos];
        c->bulk_pos ^= 1;
        const size_t need = (size_t)nsteps * 16;
        if (bp.busy) {
            HIPOK(c, hipEventSynchronize(bp.done));
            bp.busy = false;
        }
        if (bp.bytes < need) {
            if (bp.host) hipHostFree(bp.host);
            if (bp.dev) hipFree(bp.dev);
            bp.host = bp.dev = nullptr;
            bp.bytes = 0;
            HIPOK(c, hipHostMalloc((void**)&bp.host, need, hipHostMallocDefault));
            HIPOK(c, hipMalloc((void**)&bp.dev, need));
            bp.bytes = need;
            if (!bp.done) HIPOK(c, hipEventCreateWithFlags(&bp.done, hipEventDisableTiming));
        }
        double* facs = (double*)bp.host;
        int32_t* cols = (int32_t*)(bp.host + (size_t)nsteps * 8);
        for (int64_t s2 = 0; s2 < nsteps; ++s2) {
            const uint64_t step = c->ph_step + (uint64_t)s2;
            const int mi = nm == 1 ? 0 : philox_move_choice(c->ph_seed, step, c->cdf.data(), nm);
            emx_move_desc& mv = c->moves[mi];
            facs[s2] = 1.0;
            cols[s2] = 0;
            if (mv.kind != EMX_MOVE_GAUSS) continue;
            if (mv.a != 0.0) {
                const Philox4 r = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), 0x46414354u /*'FACT'*/, 0,
                                                (uint32_t)c->ph_seed, (uint32_t)(c->ph_seed >> 32));
                facs[s2] = std::exp(-mv.g0 + 2.0 * mv.g0 * u53(r.v[0], r.v[1]));
            }
            if (mv.reserved == EMX_GAUSS_SEQUENTIAL) {
                cols[s2] = (int32_t)((int64_t)mv.gammas % c->D);
                mv.gammas = (double)(((int64_t)mv.gammas + 1) % c->D);
            }
        }
        HIPOK(c, hipMemcpyAsync(bp.dev, bp.host, need, hipMemcpyHostToDevice, c->stream));
        a.step_fac = (const double*)bp.dev;
        a.step_col = (const int32_t*)(bp.dev + (size_t)nsteps * 8);
        gauss_bulk = &bp;
    }
    const int threads = small_threads(c, sh.G, minsplits, dense);
    const size_t lds = dense ? small_lds_bytes(c->N, c->D, c->Dp, threads / 64) : small_lds_bytes(c->N, c->D);
    hipError_t e = hipErrorInvalidValue;
    const int movesel = (nm == 1 && (!dense || c->moves[0].kind == EMX_MOVE_STRETCH)) ? (int)c->moves[0].kind : SMALL_ANY_MOVE;
    // (MOVE_GAUSS == 3 is a valid single selector for the element-wise targets; the dense variant carries it under ANY)
    e = emx_small_dispatch(sh.G, sh.V, sh.CH, dense ? c->Dp / 16 : 0, movesel, threads, lds, c->stream, a);
    if (e != hipSuccess) FAIL(c, -2, "k_small_run launch failed (G=%d V=%d CH=%d ndim=%d): %s", sh.G, sh.V, sh.CH, c->D, hipGetErrorString(e));
    int64_t nstored = 0;
    if (store)
        for (int64_t s2 = 0; s2 < nsteps; ++s2) nstored += ((i0 + s2 + 1) % thin_by == 0) ? 1 : 0;
    c->stored += nstored;
    c->proposals += nsteps;
    if (c->rng_mode == EMX_RNG_PHILOX) c->ph_step += (uint64_t)nsteps;
    if (a.plans) {
        auto& bp = c->bulk[c->bulk_pos ^ 1];
        HIPOK(c, hipEventRecord(bp.done, c->stream));
        bp.busy = true;
    }
    if (gauss_bulk) {
        HIPOK(c, hipEventRecord(gauss_bulk->done, c->stream));
        gauss_bulk->busy = true;
    }
    return 0;
}

// `count` doubles per pair from sendbuf block q to rank q's gathered block `rank` (RCCL's all-to-all when the
// library has it, grouped send/recv otherwise)
static int rccl_all_to_all(emx_ctx* c, size_t count) {
    int e = 0;
    if (g_rccl.AllToAll) {
        e = g_rccl.AllToAll(c->sendbuf, c->gathered, count, RCCL_FLOAT64, c->comm, c->stream);
    } else {
        NEED(c, g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd, "librccl lacks ncclSend/ncclRecv");
        e = g_rccl.GroupStart();
        for (int q = 0; q < c->world && e == 0; ++q) {
            if (q == c->rank) continue;
            e = g_rccl.Send(c->sendbuf + (size_t)q * count, count, RCCL_FLOAT64, q, c->comm, c->stream);
            if (e == 0) e = g_rccl.Recv(c->gathered + (size_t)q * count, count, RCCL_FLOAT64, q, c->comm, c->stream);
        }
        const int e2 = g_rccl.GroupEnd();
        if (e == 0) e = e2;
    }
    if (e != 0) FAIL(c, -6, "RCCL all-to-all failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return 0;
}

static int run_impl(emx_ctx* c, int64_t nsteps, int32_t thin_by, int32_t store);

// ---- persistent half-steps ------------------------------------------------------------------------------------------
// The headline shape -- stretch / DE steps of two splits and snooker steps of four, the fused dense Gaussian target at an even
// ndim up to 64, Philox plans, a single replica, an ensemble whose half-step is exactly one 16-walker tile per wave of a
// co-resident grid of about one workgroup per CU (persist_shape) -- runs up to 40 half-steps per launch (PERSIST_MAX_ITERS: twenty stretch / DE steps) in k_persist
// (emx_kernels.hpp); in a mixture of moves a launch takes the consecutive steps of one move.
// (Measured and dropped: the next batch's plan kernel on a stream of its own next to the running persistent launch -- no
// difference, 20.6 us/step either way: what the plan kernel's waves gain in overlap the lock-stepped half-steps lose to them;
// profiles/r03/persist_side_plan.txt.)
// -> waves per workgroup of the persistent grid for a move of `nsplits` splits (0: the ensemble does not fit one): every wave
// exactly one 16-walker tile of a half-step, about one workgroup per CU, all of them co-resident
static int persist_shape_of(int64_t N, int nsplits, int64_t cu) {
    if (nsplits < 2 || N < 2 || (N % nsplits) != 0) return 0;
    const int64_t own = N / nsplits;
    if ((own % 16) != 0) return 0;
    const int64_t tiles = own / 16;
    cu = std::max<int64_t>(1, cu);
    int wpb = tiles >= 6 * cu ? 8 : tiles >= 3 * cu ? 4 : tiles >= 3 * cu / 2 ? 2 : 1;
    while (wpb > 1 && (tiles % wpb) != 0) wpb >>= 1;
    if (tiles / wpb > cu * (wpb == 8 ? 1 : 2)) return 0;       // (103 KB of LDS per 8-wave group: one per CU)
    if (tiles / wpb < 8) return 0;                              // (the barrier counts arrivals per XCD: every one of the eight needs a workgroup)
    return wpb;
}
// The one-XCD form (k_persist<..., LOCAL>): every workgroup on the 32 CUs of one XCD, one workgroup per CU at most -- ensembles of up
// to 32 x 8 tiles per half-step (8 192 walkers in two splits, 16 384 in the snooker move's four -- capped at tune_persist_local_max).
// -> waves per workgroup, 0: not this form
static int persist_shape_local_of(int64_t N, int nsplits, int64_t cu) {
    if (nsplits < 2 || N < 2 || (N % nsplits) != 0) return 0;
    const int64_t own = N / nsplits;
    if ((own % 16) != 0) return 0;
    // (at most 32 working groups: persist_barrier_local keeps one flag word per group in words 0..31 of a line, word 32 is the dead mark)
    const int64_t tiles = own / 16, per_xcd = std::min<int64_t>(32, std::max<int64_t>(1, cu / 8));
    for (int wpb = 1; wpb <= 8; wpb <<= 1)
        if ((tiles % wpb) == 0 && tiles / wpb <= per_xcd) return tiles / wpb >= 2 ? wpb : 0;
    return 0;
}
static bool persist_local_ok(const emx_ctx* c, const emx_move_desc& m) {
    const bool known = ((m.kind == EMX_MOVE_STRETCH || m.kind == EMX_MOVE_DE) && m.nsplits == 2) || (m.kind == EMX_MOVE_SNOOKER && m.nsplits == 4);
    // (the slab form's 109 KB workgroups: one XCD holds 32 of them, 4 096 walkers' worth at eight waves each run well there, 8 192 do
    // not -- 34.0 against 17.9 us/step at ndim 128)
    const int64_t lmax = (c->target == EMX_TARGET_DENSE_GAUSS && c->Dp > 64) ? std::min(c->tune_persist_local_max, c->tune_persist_slab_local_max)
                                                                               : c->tune_persist_local_max;
    return c->tune_persist_local != 0 && known && c->N <= lmax &&
           persist_shape_local_of(c->N, m.nsplits, c->num_cu) != 0;
}
static int persist_shape(const emx_ctx* c, int nsplits) { return persist_shape_of(c->N, nsplits, c->num_cu); }
// the row layouts k_persist_valu is instantiated for (launch_persist_valu): rows of 8 lanes (ndim <= 64 even, <= 32 odd), of 4 lanes
// (ndim <= 4, even ndim <= 8), of 16 lanes with one coordinate a lane and chunk (odd ndim 33 ... 63)
static bool persist_valu_shape(const Shape& sh) {
    return (sh.G == 8 && (sh.CH == 1 || sh.CH == 2 || sh.CH == 4)) || (sh.G == 4 && sh.CH == 1) || (sh.G == 16 && sh.V == 1 && sh.CH == 4);
}
// The fused dense Gaussian at padded ndim 80 ... 128 (even ndim 66 ... 128: rows of 16 lanes, two coordinates a lane and chunk): the
// persistent form of the slab kernel (k_persist_slab, emx_pslab.hip; the stretch and DE moves) -- round 6
static bool persist_slab_ok(const emx_ctx* c) {
    if (!c->tune_persist_slab || c->target != EMX_TARGET_DENSE_GAUSS || dense_is_wide(c) || c->Dp < 80 || c->Dp > 128) return false;
    // Where the launch-per-half-step slab kernel fills the chip -- padded ndim 112 / 128 with eight tiles or more per CU and half-step
    // (65 536 walkers on 256 CUs) -- the persistent form is level or 1-4 % behind (43.7 against 43.3 us/step at ndim 128, 38.9 / 37.8
    // at 112, DE 53.5 / 51.5: its barrier and agent-scope accesses cost what the launch gap and the image staging did;
    // profiles/r06/pslab.txt): those shapes keep the launches.  Tuning "persist_slab" = 2 takes the persistent form there too (tests).
    // (an ODD ndim has no slab kernel among the launches -- rows of 32 lanes, four waves a CU: 62 / 73 us/step at 65 536 x 97 / 127 -- and
    // takes the persistent form at every size)
    if (c->tune_persist_slab != 2 && c->Dp >= 112 && c->N / 2 / 16 >= 8 * (int64_t)c->num_cu && !(c->D & 1)) return false;
    // even ndim 66 ... 128 (pick_shape: rows of 16 lanes, two coordinates a lane); odd ndim 65 ... 127 (pick_shape: rows of 32 lanes, one
    // coordinate a lane, for the launch-per-half-step kernels -- the persistent kernel keeps the 16-lane register layout and moves the
    // rows 8 bytes at a time, emx_pslab.hip)
    const Shape sh = pick_shape(c->D, c->Dp);
    return (sh.G == 16 && sh.V == 2 && sh.CH == 4) || (c->tune_persist_odd != 0 && sh.G == 32 && sh.V == 1 && sh.CH == 4);
}

// the moves k_persist has an instantiation for (the other moves of a mixture run their steps through the per-half-step launches)
// Checked, not assumed: the runtime's occupancy figure for this instantiation, block size and LDS need x the CU count must cover
// the grid, or the launch could never become co-resident even on an idle device (asked once per context and move).
static bool persist_grid_fits(const emx_ctx* cc, const emx_move_desc& m, int wpb) {
    emx_ctx* c = const_cast<emx_ctx*>(cc);
    if (m.kind < 0 || m.kind > 3) return false;
    if (c->persist_fits[m.kind] < 0) {
        const int64_t groups = c->N / m.nsplits / 16 / wpb;
        int per_cu = 0;
        const hipError_t e = c->Dp > 64 ? persist_slab_occupancy(c->Dp / 16, m.kind, c->D & 1, 64 * wpb, slab_lds_bytes(c->Dp, wpb), &per_cu)
                             : (c->D & 1) ? persist_dense_odd_occupancy(c->Dp / 16, m.kind, 64 * wpb, dense_lds_bytes(c->Dp, wpb), &per_cu)
                                        : hot_persist_occupancy(c->Dp / 16, m.kind, 64 * wpb, dense_lds_bytes(c->Dp, wpb), &per_cu);
        c->persist_fits[m.kind] = (e != hipSuccess || (int64_t)per_cu * c->num_cu >= groups) ? 1 : 0;     // (no answer: as before)
    }
    return c->persist_fits[m.kind] != 0;
}

// The element-wise targets' persistent kernel in its device-wide form (k_persist_valu<..., LOCAL = false>): exact mode only -- there a
// launch takes the fetched plans of sixteen steps where the per-half-step path pays an upload and five API calls per step (16 384 x 64:
// 33 -> ~21 us/step); with Philox plans the per-half-step launches of these small kernels are as fast.  No LDS, <= 512 threads, at most
// two workgroups a CU (persist_shape): co-resident by construction.
static bool persist_valu_wide_ok(const emx_ctx* c, const emx_move_desc& m) {
    const bool known = ((m.kind == EMX_MOVE_STRETCH || m.kind == EMX_MOVE_DE) && m.nsplits == 2) || (m.kind == EMX_MOVE_SNOOKER && m.nsplits == 4);
    const bool valu = c->target == EMX_TARGET_ISO_GAUSS || c->target == EMX_TARGET_DIAG_GAUSS || c->target == EMX_TARGET_ROSENBROCK || c->target == EMX_TARGET_BOX;
    // (stretch steps that go up as generator states -- regen_ctx_ok -- lift the bound of plans that travel finished: persist_exact_ok)
    const int64_t nmax = (regen_ctx_ok(c) && m.kind == EMX_MOVE_STRETCH) ? std::max(c->tune_persist_exact_max, c->tune_persist_exact_regen_max) : c->tune_persist_exact_max;
    return known && valu && c->rng_mode == EMX_RNG_MT19937 && c->tune_persist_valu != 0 && c->N <= nmax && persist_shape(c, m.nsplits) != 0;
}

static bool persist_move_ok(const emx_ctx* c, const emx_move_desc& m) {
    const bool known = ((m.kind == EMX_MOVE_STRETCH || m.kind == EMX_MOVE_DE) && m.nsplits == 2) || (m.kind == EMX_MOVE_SNOOKER && m.nsplits == 4);
    if (!known) return false;
    if (c->target == EMX_TARGET_DENSE_GAUSS && c->Dp > 64 && (m.kind == EMX_MOVE_SNOOKER || !persist_slab_ok(c))) return false;       // (k_persist_slab: stretch and DE)
    if (c->target == EMX_TARGET_DENSE_GAUSS && c->Dp > 64 && (c->D & 1) && m.kind != EMX_MOVE_STRETCH) return false;                 // (... at an odd ndim: stretch)
    if (persist_local_ok(c, m)) return true;            // (one workgroup per CU of one XCD by construction)
    if (c->target != EMX_TARGET_DENSE_GAUSS) return persist_valu_wide_ok(c, m);       // (element-wise targets: the one-XCD form, or -- exact mode -- the device-wide one)
    const int wpb = persist_shape(c, m.nsplits);
    return wpb != 0 && persist_grid_fits(c, m, wpb);
}

// A schedule that holds both a DEMove (two splits) and a DESnookerMove (four), dense Gaussian target: steps of either share launches
// (k_persist<..., MOVE_MIX>, emx_pmix.hip).  Both moves must qualify on their own (persist_move_ok), the DE move's grid must hold the
// mixed instantiation co-resident, and -- the one-XCD form -- both must be one-XCD moves.
static bool persist_mix_member(const emx_move_desc& m) {
    return (m.kind == EMX_MOVE_DE && m.nsplits == 2) || (m.kind == EMX_MOVE_SNOOKER && m.nsplits == 4);
}
static bool persist_mix_local(const emx_ctx* c) {
    for (const auto& m : c->moves)
        if (persist_mix_member(m) && !persist_local_ok(c, m)) return false;
    return persist_shape_local_of(c->N, 2, c->num_cu) != 0;
}
static bool persist_mix_ok(const emx_ctx* cc) {
    emx_ctx* c = const_cast<emx_ctx*>(cc);
    if (!c->tune_persist_mix || c->target != EMX_TARGET_DENSE_GAUSS || c->Dp > 64 || (c->D & 1)) return false;      // (k_persist_mix: even ndim up to 64)
    if (c->rng_mode != EMX_RNG_PHILOX && !(c->rng_mode == EMX_RNG_MT19937 && persist_exact_ok(c))) return false;
    bool de = false, sn = false;
    for (const auto& m : c->moves) {
        if (!persist_mix_member(m)) continue;
        if (!persist_move_ok(c, m)) return false;
        de = de || m.kind == EMX_MOVE_DE;
        sn = sn || m.kind == EMX_MOVE_SNOOKER;
    }
    if (!de || !sn || (c->N % 64) != 0) return false;
    if (persist_mix_local(c)) return true;
    const int wpb = persist_shape(c, 2);
    if (wpb == 0) return false;
    if (c->persist_mix_fits < 0) {
        int per_cu = 0;
        const hipError_t e = persist_mix_occupancy(c->Dp / 16, 64 * wpb, dense_lds_bytes(c->Dp, wpb), &per_cu);
        c->persist_mix_fits = (e != hipSuccess || (int64_t)per_cu * c->num_cu >= c->N / 2 / 16 / wpb) ? 1 : 0;
    }
    return c->persist_mix_fits != 0;
}

// Exact (MT19937) mode with the one-XCD forms: the host pipeline's plans (csrc/emx_mtpipe.cpp) are uploaded eight steps ahead of the
// launch that takes them -- one StretchMove / DEMove / DESnookerMove alone (a mixture's next move is only known once its plan has been
// taken), ensembles of 512 ... 8 192 walkers.  The default rng of the Python layer: 25-33 -> 6-9 us/step (profiles/r04/exact_mid.txt).
static bool persist_exact_ok(const emx_ctx* c) {
    // (an ensemble the one-workgroup kernel can take, asked for a step or two at a time -- the sample() loop of a progress bar or a
    // convergence check: that kernel's 10 us launch beats a fetch + a persistent launch per call, 49 against 72 us an iteration)
    if (small_eligible(c) && c->call_steps < 4) return false;
    {   // a target one of the persistent kernels takes (k_persist: dense Gaussian up to padded ndim 64; k_persist_valu: the element-wise
        // targets at the row shapes it is instantiated for) -- otherwise no persistent launch can follow, and the pipeline must not
        // be started, sized or paced (bursty consumer) for one
        const bool dense = c->target == EMX_TARGET_DENSE_GAUSS && (c->Dp <= 64 || persist_slab_ok(c)) && !dense_is_wide(c);
        bool valu = c->target == EMX_TARGET_ISO_GAUSS || c->target == EMX_TARGET_DIAG_GAUSS || c->target == EMX_TARGET_ROSENBROCK || c->target == EMX_TARGET_BOX;
        if (valu) {
            const Shape sh = pick_shape(c->D, c->D);
            valu = persist_valu_shape(sh);
        }
        if (!dense && !valu) return false;
    }
    if (!(c->rng_mode == EMX_RNG_MT19937 && c->tune_persist_exact != 0 && !c->moves.empty() && c->tune_mt_pipeline != 0 &&
          c->N >= c->tune_persist_min_walkers && !mtdev_eligible(c) &&        // (the device producer: see persist_wanted)
          MtPlanPipeline::supports((int32_t)c->moves.size(), c->moves.data())))
        return false;
    if (c->moves.size() > 1 && !c->tune_persist_exact_mix) return false;
    // EVERY move of the schedule (round 5: a mixture's next move is read off the pipeline's plan before it is taken, run_persist):
    // the one-XCD form, or the device-wide form of the dense kernel up to "persist_exact_max_walkers" (32 768) -- beyond, the
    // pipeline's generator thread bounds the step whatever runs it, and a launch's plans are 25 MB to fetch over PCIe
    for (const auto& m : c->moves) {
        const bool known = ((m.kind == EMX_MOVE_STRETCH || m.kind == EMX_MOVE_DE) && m.nsplits == 2) || (m.kind == EMX_MOVE_SNOOKER && m.nsplits == 4);
        if (!known && c->moves.size() > 1) return false;       // (one move: persist_local_ok / persist_move_ok say the same, later)
        if (persist_local_ok(c, m)) continue;
        // (a regen ensemble's plans are `order` and a few generator states a step -- 6.8 MB a launch at 65 536 walkers where finished
        // plans were 25 MB: it takes the device-wide form up to "persist_exact_regen_max_walkers")
        const int64_t nmax = (regen_ctx_ok(c) && m.kind == EMX_MOVE_STRETCH) ? std::max(c->tune_persist_exact_max, c->tune_persist_exact_regen_max) : c->tune_persist_exact_max;
        if (c->N > nmax || persist_shape(c, m.nsplits) == 0) return false;
        if (!(c->target == EMX_TARGET_DENSE_GAUSS || persist_valu_wide_ok(c, m))) return false;
    }
    return true;
}

static bool persist_wanted(const emx_ctx* c) {
    if (!c->tune_persist) return false;
    // Philox plans only.  (The reference's own stream with its plans made on the device -- emx_mtdev.hpp -- has them in HBM in time
    // too, but its tokenizer is one 1024-thread, 107 KB workgroup that runs all the time: a persistent grid that holds every CU
    // leaves it none, and the two took turns -- k_persist 314 -> 1 150 us a launch, profiles/r04/mtdev_timeline.txt.)
    if (!(c->rng_mode == EMX_RNG_PHILOX || persist_exact_ok(c)) || c->world != 1 || c->comm || c->sendbuf || c->peers_ready || c->moves.empty()) return false;
    if (c->target != EMX_TARGET_DENSE_GAUSS || dense_is_wide(c) || (c->Dp > 64 && !persist_slab_ok(c))) return false;
    if (c->tune_ablate || (c->dbg && !EMX_OPT_STAMPS) || c->tune_spw || c->tune_wpb || c->tune_graph) return false;     // (an instrumented build stamps k_persist too)
    if (c->N < c->tune_persist_min_walkers) return false;
    bool any = false;
    for (const auto& m : c->moves) any = any || persist_move_ok(c, m);
    if (!any) return false;
    if (c->Dp > 64) return true;           // (persist_slab_ok: even ndim 66 ... 128, odd 65 ... 127)
    // ndim up to 64: the row layouts k_persist is instantiated for -- even ndim: two coordinates per lane, rows of 8 lanes (emx_hot.hip);
    // odd ndim (round 6, emx_podd.hip): one coordinate per lane, rows of 8 lanes up to padded ndim 32, of 16 lanes beyond
    const Shape sh = pick_shape(c->D, c->Dp);
    const int dpb = c->Dp / 16;
    if (sh.V == 1) return c->tune_persist_odd != 0 && (dpb <= 2 ? (sh.G == 8 && sh.CH == 2 * dpb) : (sh.G == 16 && sh.CH == 4));
    return sh.G == 8 && sh.V == 2 && sh.CH == (dpb == 1 ? 1 : dpb == 2 ? 2 : 4);
}

// The element-wise targets (csrc/emx_pvalu.hip): the one-XCD form only -- ensembles of up to 8 192 walkers, rows of 4 or 8 lanes per walker
// (ndim <= 64 even, <= 32 odd), Philox plans, one replica, every move of the schedule one the kernel knows at a shape it can take.
static bool persist_valu_wanted(const emx_ctx* c) {
    if (!c->tune_persist || !c->tune_persist_local || !c->tune_persist_valu) return false;
    if (!(c->rng_mode == EMX_RNG_PHILOX || persist_exact_ok(c)) || c->world != 1 || c->comm || c->sendbuf || c->peers_ready || c->moves.empty()) return false;
    if (c->target != EMX_TARGET_ISO_GAUSS && c->target != EMX_TARGET_DIAG_GAUSS && c->target != EMX_TARGET_ROSENBROCK && c->target != EMX_TARGET_BOX)
        return false;
    if (c->tune_ablate || c->dbg || c->tune_spw || c->tune_wpb || c->tune_graph) return false;
    if (c->N < c->tune_persist_min_walkers) return false;
    const Shape sh = pick_shape(c->D, c->D);
    if (!persist_valu_shape(sh)) return false;      // (launch_persist_valu's instantiations)
    bool any = false;
    for (const auto& m : c->moves) any = any || persist_local_ok(c, m) || persist_valu_wide_ok(c, m);
    return any;
}

// The Gaussian Metropolis move alone (moves/gaussian.py + mh.py), fused dense target at an even ndim up to 64, Philox plans, one
// replica, the noise generated in registers: k_persist_gauss keeps every walker in registers for up to 16 steps per launch.
static bool persist_gauss_wanted(const emx_ctx* c) {
    if (!c->tune_persist || c->tune_gauss_materialize) return false;
    if (c->rng_mode != EMX_RNG_PHILOX || c->world != 1 || c->comm || c->sendbuf || c->peers_ready || c->moves.size() != 1) return false;
    if (c->moves[0].kind != EMX_MOVE_GAUSS) return false;
    if (c->target != EMX_TARGET_DENSE_GAUSS || c->Dp > 64 || dense_is_wide(c)) return false;
    if (c->tune_ablate || c->dbg || c->tune_spw || c->tune_wpb || c->tune_graph) return false;
    if (c->N < c->tune_persist_min_walkers || (c->N % 16) != 0) return false;
    const Shape sh = pick_shape(c->D, c->Dp);
    const int dpb = c->Dp / 16;
    return sh.G == 8 && sh.V == 2 && sh.CH == (dpb == 1 ? 1 : dpb == 2 ? 2 : 4);
}

static int run_persist_gauss(emx_ctx* c, int64_t i0, int64_t total, int32_t thin_by, int32_t store, int64_t* done) {
    *done = 0;
    const int64_t tiles = c->N / 16;
    // (four-wave groups, two per CU, drift out of phase: 12.9 against 13.0 us/step with eight-wave groups, 13.9 with two-wave ones)
    int wpb = c->tune_persist_gauss_wpb > 0 ? (int)c->tune_persist_gauss_wpb : 4;
    while (wpb > 1 && (tiles % wpb) != 0) wpb >>= 1;
    c->persist_wpb = wpb;
    PersistGaussArgs P{};
    emx_ctx::PersistCapture cap{};
    int64_t steps = 0;
    while (i0 + steps < total && steps < PERSIST_GAUSS_MAX_STEPS) {
        if (steps > 0 && c->prepared.empty()) break;          // one plan batch per launch
        c->prep_hint = NATIVE_BATCH_MAX;
        const int st = store && ((i0 + steps + 1) % thin_by == 0);          // ensemble.py:416
        int mvi, S;
        int rc = emx_step_begin(c, st, &mvi, &S);
        if (rc) return rc;
        cap.got = false;
        c->persist_cap = &cap;
        rc = do_halfstep(c, 0, c->target);
        c->persist_cap = nullptr;
        if (!rc && (S != 1 || !cap.dense || cap.move != MOVE_GAUSS || cap.a.disp || cap.dpb != c->Dp / 16)) {
            c->err = "persistent Gaussian steps: launch shape not eligible";
            rc = -1;
        }
        if (rc) {
            c->cur.active = false;
            return rc;
        }
        if (steps == 0) P.base = cap.a;
        PersistGaussStep& G = P.st[steps];
        G.p0 = cap.a.p0;
        G.logu = cap.a.logu;
        G.chain = cap.a.chain;
        G.chain_lp = cap.a.chain_lp;
        G.gstep = cap.a.gstep;
        G.gfac = cap.a.gfac;
        rc = emx_step_end(c);
        if (rc) return rc;
        ++steps;
    }
    P.nsteps = (int32_t)steps;
    const dim3 grid((unsigned)(tiles / wpb)), block(64u * (unsigned)wpb);
    const size_t lds = dense_lds_bytes(c->Dp, wpb);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = c->prof_max > 0 && c->prof_n < c->prof_max;
    if (prof) {
        e0 = c->prof[2 * c->prof_n];
        e1 = c->prof[2 * c->prof_n + 1];
        HIPOK(c, hipEventRecord(e0, c->stream));
    }
    const hipError_t e = launch_hot_persist_gauss(c->Dp / 16, grid, block, lds, c->stream, P);
    if (e != hipSuccess) FAIL(c, -2, "persistent Gaussian launch failed: %s", hipGetErrorString(e));
    if (prof) {
        HIPOK(c, hipEventRecord(e1, c->stream));
        c->prof_n++;
    }
    c->persist_launches++;
    c->persist_halfsteps += steps;
    *done = steps;
    return 0;
}

// Two persistent grids that each hold part of the device would wait for each other until their barriers time out: launches of
// one process on one device are chained (each waits for the one before; a few cycles when it is the same stream).
static std::mutex g_persist_mu;
static hipEvent_t g_persist_ev[MAX_DEVICES] = {};
static const emx_ctx* g_persist_last[MAX_DEVICES] = {};

static int run_persist(emx_ctx* c, int64_t i0, int64_t total, int32_t thin_by, int32_t store, int64_t* done) {
    *done = 0;
    if (__atomic_load_n(&c->status_host[3], __ATOMIC_ACQUIRE)) {
        // a barrier of an earlier persistent launch was never met (status bit 3, not read yet): a launch that gave up at its
        // handshake left the ensemble untouched and is redone on the per-half-step path; anything else is void
        const int rcs = persist_settle(c);
        if (rcs) return rcs;
        if (__atomic_load_n(&c->status_host[3], __ATOMIC_ACQUIRE)) {
            c->persist_grid = 0;          // (the counters restart once the status has been taken)
            FAIL(c, -8, "persistent kernel: a device-wide barrier timed out in the middle of a launch; read emx_status, "
                        "or turn the persistent path off with tuning \"persist\" = 0");
        }
        return 0;                         // (*done == 0: the caller's loop comes back, now on the per-half-step path)
    }
    if (c->plog.size() >= 4096) {         // bound the log: everything the stream has run without a timeout is settled
        if (hipStreamQuery(c->stream) == hipSuccess && !__atomic_load_n(&c->status_host[3], __ATOMIC_ACQUIRE)) {
            c->plog.clear();
        } else if (c->plog.size() >= 65536) {
            const int rcs = persist_settle(c);
            if (rcs) return rcs;
        }
    }
    if (!c->persist_bar) {
        HIPOK(c, hipMalloc((void**)&c->persist_bar, PERSIST_BAR_WORDS * sizeof(unsigned)));
        HIPOK(c, hipHostMalloc((void**)&c->persist_started, 64, hipHostMallocDefault));
        *c->persist_started = 0u;
        HIPOK(c, hipMemsetAsync(c->persist_bar, 0, PERSIST_BAR_WORDS * sizeof(unsigned), c->stream));
        c->persist_epoch = 0;
    }
    if (!c->persist_ver) {
        HIPOK(c, hipMalloc((void**)&c->persist_ver, (size_t)c->N * 4));
        HIPOK(c, hipMemsetAsync(c->persist_ver, 0, (size_t)c->N * 4, c->stream));
    }
    PersistArgs P{};
    emx_ctx::PersistCapture cap{};
    dim3 grid, block;
    size_t lds = 0;
    int n = 0;
    int64_t steps = 0;
    const bool mtmode = c->rng_mode == EMX_RNG_MT19937;     // exact mode: the host pipeline's plans (persist_exact_ok), eight steps a launch
    const bool devp = mtmode && mtdev_eligible(c);           // (plans from the device producer: not since it keeps the per-half-step launches)
    std::vector<int> used_slots;
    c->pipe_defer = mtmode && !devp;                         // the launch's plans are fetched by ONE kernel (pipe_fetch_deferred)
    struct Undefer2 {
        emx_ctx* c;
        ~Undefer2() { c->pipe_defer = false; }
    } undefer2{c};
    c->mtdev_defer_release = devp;                          // a batch's slots are released behind the LAUNCH that reads them, not behind the capture
    c->mtdev_release_pending = -1;
    struct Undefer {
        emx_ctx* c;
        ~Undefer() { c->mtdev_defer_release = false; }
    } undefer{c};
    emx_ctx::PersistLog lg{};
    lg.ph_step = c->ph_step;
    lg.i0 = i0;
    lg.stored0 = c->stored;
    lg.proposals0 = c->proposals;
    lg.thin_by = thin_by;
    lg.store = store;
    if (mtmode && !devp) {
        lg.pipe_step0 = c->pipe ? c->pipe_taken : 0;
        lg.pipe_id = c->pipe;
    }
    int launch_move = -1;                // EMX_MOVE_STRETCH, _DE or _SNOOKER: the move of every step of this launch
    int launch_S = 2;
    bool launch_local = false;           // the one-XCD form: an eight times larger grid of which every eighth workgroup works
    bool launch_mix = false;             // DE and snooker steps of a mixture in this launch (k_persist_mix)
    double launch_gammas = 0.0;
    // Philox plans live in a ring of TWO batches of sixteen steps (PLAN_RING): a launch may read the plans of at most two batches --
    // a third would be written into the half of the ring whose slots this very launch still reads (round 6, when a launch grew from
    // sixteen to twenty steps: a call that started three steps before a batch boundary took plans of three batches)
    int batches_touched = (!mtmode && !c->prepared.empty() && c->prepared.front().step == c->ph_step) ? 1 : 0;
    while (i0 + steps < total) {
        if (!mtmode && (c->prepared.empty() || c->prepared.front().step != c->ph_step)) {
            if (batches_touched >= 2) break;
            ++batches_touched;
        }
        // the move of the step that would follow: off its plan, or -- the batch of plans is used up: the next one is made during this
        // capture, i.e. enqueued BEFORE this launch, into the other half of the plan ring -- from the Philox stream directly
        int next_move = -1;
        if (steps > 0 && !mtmode)
            next_move = !c->prepared.empty() ? c->prepared.front().move : philox_move_choice(c->ph_seed, c->ph_step, c->cdf.data(), (int)c->moves.size());
        if (steps > 0 && mtmode && !devp && c->moves.size() > 1) {
            // exact mode, a mixture: the pipeline has made (or is making) the next step's plan -- its move is read off it without taking it
            PipeStepInfo peek;
            if (!c->pipe || !c->pipe->wait_ready(c->pipe_taken, peek, pipe_poll, c))
                FAIL(c, -7, "exact-mode plan pipeline stopped before step %lld", (long long)c->pipe_taken);
            next_move = peek.move;
        }
        int need = launch_S;               // half-steps of the step that would follow
        if (launch_mix && next_move >= 0) need = c->moves[next_move].nsplits;
        if (n + need > (int)std::min<int64_t>(PERSIST_MAX_ITERS, std::max<int64_t>(4, c->tune_persist_max_halfsteps))) break;
        if (devp) {
            if (steps > 0 && c->mtdev && c->mtdev_taken % MTDEV_BATCH == 0) break;      // one produced batch per launch
        } else if (mtmode) {
            if (steps >= std::min<int64_t>(c->tune_persist_exact_steps, c->pipe_nsinks / 2)) break;          // (k_plan_fetch takes sixteen plans; half of the pipeline's slots at most: the other half is produced meanwhile)
            if (steps > 0 && next_move >= 0) {               // a mixture: the same rules as with Philox plans (below)
                if (!launch_mix && c->moves[next_move].kind != launch_move) break;
                if (!launch_mix && launch_move == EMX_MOVE_SNOOKER && c->moves[next_move].gammas != launch_gammas) break;
                if (launch_mix && !persist_mix_member(c->moves[next_move])) break;
                if (!persist_move_ok(c, c->moves[next_move])) break;
            }
        } else {
            if (steps > 0 && c->prepared.empty() && !c->tune_persist_span) break;          // (tuning persist_span = 0: one plan batch per launch)
            if (steps > 0 && !launch_mix && c->moves[next_move].kind != launch_move) break;       // a mixture: the run of this move ends here
            // (the snooker scale is a launch-wide kernel argument there: a second DESnookerMove with another `gammas` starts a launch of its own)
            if (steps > 0 && !launch_mix && launch_move == EMX_MOVE_SNOOKER && c->moves[next_move].gammas != launch_gammas) break;
            if (steps > 0 && launch_mix && !persist_mix_member(c->moves[next_move])) break;
            if (steps > 0 && !persist_move_ok(c, c->moves[next_move])) break;
        }
        c->prep_hint = NATIVE_BATCH_MAX;
        const int st = store && ((i0 + steps + 1) % thin_by == 0);          // ensemble.py:416
        int mvi, S;
        int rc = emx_step_begin(c, st, &mvi, &S);
        if (rc) return rc;
        if (launch_move < 0) {
            launch_move = c->moves[mvi].kind;
            launch_gammas = c->moves[mvi].gammas;
            launch_S = S;
            launch_local = persist_local_ok(c, c->moves[mvi]);
            c->persist_wpb = launch_local ? persist_shape_local_of(c->N, S, c->num_cu) : persist_shape(c, S);
            if (persist_mix_ok(c) && (launch_move == EMX_MOVE_DE || launch_move == EMX_MOVE_SNOOKER)) {
                // a schedule of DE (two splits) and snooker (four) moves: ONE launch for steps of either (k_persist<..., MOVE_MIX>), at
                // the DE move's shape -- every other wave works in a snooker half-step
                launch_mix = true;
                launch_local = persist_mix_local(c);
                c->persist_wpb = launch_local ? persist_shape_local_of(c->N, 2, c->num_cu) : persist_shape(c, 2);
            }
        }
        for (int s = 0; s < S; ++s) {
            cap.got = false;
            c->persist_cap = &cap;
            rc = do_halfstep(c, s, c->target);
            c->persist_cap = nullptr;
            const bool valu = c->target != EMX_TARGET_DENSE_GAUSS;          // the element-wise targets' kernel (one-XCD form only)
            const bool move_ok = launch_mix ? (cap.move == MOVE_DE || cap.move == MOVE_SNOOKER) : cap.move == launch_move;
            if (!rc && (!cap.got || cap.dense == valu || (!valu && cap.dpb != c->Dp / 16) || !move_ok ||
                        (int)cap.block.x != 64 * c->persist_wpb || (valu && !launch_local && !persist_valu_wide_ok(c, c->moves[mvi])))) {
                c->err = "persistent half-steps: launch shape not eligible";
                rc = -1;
            }
            if (rc) {
                c->cur.active = false;
                return rc;
            }
            if (n == 0) {
                P.base = cap.a;
                grid = cap.grid;
                block = cap.block;
                lds = cap.lds;
            }
            PersistIter& I = P.it[n++];
            {   // one pointer for the plan's columns: every slot is one block in the layout PersistCols spells out
                const char* blk = reinterpret_cast<const char*>(cap.a.order);
                const size_t NN = (size_t)c->N;
                NEED(c, (const void*)cap.a.p0 == (const void*)(blk + NN * 4) && (const void*)cap.a.s0 == (const void*)(blk + NN * 8) &&
                            (const void*)cap.a.p1 == (const void*)(blk + NN * 24) && (const void*)cap.a.p2 == (const void*)(blk + NN * 28) &&
                            (const void*)cap.a.logu == (const void*)(blk + NN * 32) && (const void*)cap.a.fac == (const void*)(blk + NN * 40),
                     "persistent half-steps: the plan slot is not one block in the expected layout");
                I.plan = blk;
            }
            I.chain = cap.a.chain;
            I.chain_lp = cap.a.chain_lp;
            I.pos0 = cap.a.pos0;
            I.split = cap.a.split;
            I.kind = cap.move;
            I.gammas = cap.a.gammas;
            I.shift = (launch_mix && S == 4) ? 1 : 0;
        }
        if (mtmode && !devp && c->cur.slot >= 0) used_slots.push_back(c->cur.slot);
        rc = emx_step_end(c);
        if (rc) return rc;
        ++steps;
    }
    if (mtmode && !devp) {
        c->pipe_fetch_local = launch_local;
        const int rcf = pipe_fetch_deferred(c);
        if (rcf) return rcf;
    }
    c->pipe_defer = false;
    if (launch_mix) grid = dim3((unsigned)(c->N / 2 / 16 / c->persist_wpb));       // (the DE move's grid, whatever the first step was)
    if (launch_local) grid.x *= 8;
    if (grid.x != c->persist_grid) {
        // the arrival counters count workgroups: another grid size (another move of a mixture, another ensemble shape) starts them
        // afresh -- on the stream, i.e. after every earlier launch has left the barrier
        HIPOK(c, hipMemsetAsync(c->persist_bar, 0, PERSIST_BAR_WORDS * sizeof(unsigned), c->stream));
        c->persist_epoch = 0;
        c->persist_lepoch = 0;
        c->persist_hepoch = 0;
        c->persist_grid = grid.x;
    }
    P.niter = n;
    P.bar = c->persist_bar;
    P.started_host = c->persist_started;
    P.ver = c->persist_ver;
    P.epoch0 = c->persist_epoch + (unsigned)c->tune_persist_test_skew;      // (tests: a barrier that is never met)
    P.lepoch0 = c->persist_lepoch;
    P.hepoch0 = c->persist_hepoch + (unsigned)c->tune_persist_test_skew;
    if (launch_local) {
        c->persist_lepoch += (unsigned)(n - 1);
        c->persist_hepoch += 1u;
    }
    P.timeout_ticks = 100000000ull * (unsigned long long)std::max<int64_t>(1, c->tune_persist_timeout_ms) / 1000ull;       // 100 MHz wall clock
    // (k_persist's stretch form without stored rows is where it was measured to pay: persist_stagger_wait, csrc/emx_kernels.hpp)
    // (and k_persist_mix with a longer wait: c4 30.0-30.2 -> 29.5-29.8 us/step)
    // (k_persist_valu has no such wait; nor has k_persist_slab, whose sibling skew it did not improve: 65 536 x 128 44.1-44.3 us/step with
    // either or both, 44.8-45.2 with neither, profiles/r06/stagger/pslab_stagger_ab.txt)
    const bool fam_dense64 = c->target == EMX_TARGET_DENSE_GAUSS && c->Dp <= 64;
    P.stagger = c->tune_persist_stagger >= 0 ? (int32_t)c->tune_persist_stagger
                : (launch_local || store || !fam_dense64) ? 0
                : (launch_mix || launch_move == EMX_MOVE_DE) ? 528
                : launch_move == EMX_MOVE_STRETCH ? 516 : 0;
    c->persist_epoch += (unsigned)n;          // the handshake and the n - 1 barriers between the half-steps
    P.seq = ++c->persist_seq;
    lg.seq = P.seq;
    lg.steps = steps;
    // (redoing a launch needs its plans again: a pure function of the step for Philox; in exact mode the pipeline is taken back to
    // the generator state in front of the launch -- persist_settle -- which the device producer's batches do not offer)
    if (!mtmode || (!devp && lg.pipe_id != nullptr)) c->plog.push_back(lg);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = c->prof_max > 0 && c->prof_n < c->prof_max;
    if (prof) {
        e0 = c->prof[2 * c->prof_n];
        e1 = c->prof[2 * c->prof_n + 1];
    }
    {
        std::lock_guard<std::mutex> lk(g_persist_mu);
        const int dev = c->device;
        if (dev >= 0 && dev < MAX_DEVICES) {
            if (!g_persist_ev[dev]) HIPOK(c, hipEventCreateWithFlags(&g_persist_ev[dev], hipEventDisableTiming));
            if (g_persist_last[dev] && g_persist_last[dev] != c) HIPOK(c, hipStreamWaitEvent(c->stream, g_persist_ev[dev], 0));
        }
        if (prof) HIPOK(c, hipEventRecord(e0, c->stream));
        hipError_t e;
        // launches that store chain rows: the one-XCD form, and the device-wide form when every CU has eight tiles a half-step (65 536 x 64:
        // 27.6 -> 26.3 us/step; 8 192: 14.7 -> 14.1, 4 096: 10.05 -> 9.8; but 16 384 / 32 768 walkers, two / four tiles a CU: +1.5 %;
        // profiles/r06/stagger/rows_late_mid.txt) -- tuning persist_rows_late 0: never, 1: this rule, 2: always
        const bool rows_late = store && (c->tune_persist_rows_late == 2 ||
                                         (c->tune_persist_rows_late == 1 && (launch_local || c->N / 2 / 16 >= 8 * (int64_t)c->num_cu)));
        if (c->target != EMX_TARGET_DENSE_GAUSS) {
            const Shape shv = pick_shape(c->D, c->D);
            e = launch_persist_valu(shv.G, shv.V, shv.CH, launch_move, launch_local ? 1 : 0, grid, block, c->stream, P);
        } else if (launch_mix) {
            e = launch_persist_mix(c->Dp / 16, launch_local ? 1 : 0, grid, block, lds, c->stream, P);
        } else if (c->Dp > 64) {
            // (lean launches carry no ablation mask: bit 8 = skewed start, bits 9-10 = when the sibling starts -- as k_halfstep_slab;
            // tuning "persist_slab_skew": 0 off, 1 ... 4)
            P.base.ablate = c->tune_persist_slab_skew ? 256 | (int32_t)((c->tune_persist_slab_skew - 1) << 9) : 0;
            e = launch_persist_slab(c->Dp / 16, launch_move, launch_local ? 1 : 0, c->D & 1, grid, block, lds, c->stream, P);
            c->persist_slab_launches++;
        } else if (c->D & 1) {
            e = launch_persist_dense_odd(c->Dp / 16, launch_move, launch_local ? 1 : 0, grid, block, lds, c->stream, P);
        } else {
            e = launch_hot_persist_dense(c->Dp / 16, launch_move, launch_local ? 1 : 0, rows_late ? 1 : 0, grid, block, lds, c->stream, P);
        }
        if (launch_local) c->persist_local_launches++;
        if (e != hipSuccess) FAIL(c, -2, "persistent half-step launch failed: %s", hipGetErrorString(e));
        if (prof) {
            HIPOK(c, hipEventRecord(e1, c->stream));
            c->prof_n++;
        }
        if (dev >= 0 && dev < MAX_DEVICES) {
            HIPOK(c, hipEventRecord(g_persist_ev[dev], c->stream));
            g_persist_last[dev] = c;
        }
    }
    // exact mode: the plan slots this launch reads were marked consumed when their steps were captured -- before the launch was
    // enqueued; mark them again behind it (the pipeline's next upload into a slot waits for the event's latest record)
    if (!used_slots.empty()) {
        hipEvent_t& ev = c->pipe_batch_ev[4 + (c->pipe_cons_n & 3)];
        if (!ev) HIPOK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->pipe_cons_n++;
        HIPOK(c, hipEventRecord(ev, c->stream));
        for (int sl : used_slots) {
            c->ring[sl].consumed_ref = ev;
            c->ring[sl].busy = true;
            c->ring[sl].last_seq = P.seq;
        }
    }
    if (mtmode && !devp) pipe_poll(c);         // (the fetch is usually done by now: its staging buffers go back to the producers)
    c->mtdev_defer_release = false;
    if (c->mtdev_release_pending >= 0 && c->mtdev) {
        const int rcr = c->mtdev->release_batch(c->mtdev_release_pending, c->stream);
        c->mtdev_release_pending = -1;
        if (rcr) FAIL(c, rcr, "%s", c->mtdev->error().c_str());
    }
    c->persist_launches++;
    c->persist_halfsteps += n;
    *done = steps;
    return 0;
}

// Everything a persistent launch was asked to do is known to be done -- or is done again here.  A launch whose grid could not
// become co-resident within the time limit (another process holds the CUs with a persistent grid of its own: the one case the
// occupancy check cannot see) left at its handshake WITHOUT a store, and so did every persistent launch queued behind it (the
// dead mark): the ensemble is exactly what the last completed launch left, and the logged steps of the others are taken
// again by the launch-per-half-step kernels -- same plans (a pure function of the step number), same bits.  The persistent
// path is then switched off for this context (tuning "persist" = 1 turns it back on).  A time-out in the MIDDLE of a launch
// (a resident workgroup that stopped for seconds: nothing we know produces it) stays what it was: status bit 3, void state.
static int persist_settle(emx_ctx* c) {
    if (c->plog.empty()) return 0;
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, wait_stream(c->stream));
    if (!__atomic_load_n(&c->status_host[3], __ATOMIC_ACQUIRE) || !c->persist_bar) {
        c->plog.clear();
        return 0;
    }
    unsigned w[4] = {0, 0, 0, 0};
    HIPOK(c, hipMemcpy(w, c->persist_bar + 9 * PERSIST_BAR_STRIDE, sizeof(w), hipMemcpyDeviceToHost));
    if (w[1] != 1u) {           // not (only) a clean handshake time-out: a direct-exchange barrier, or the middle of a launch
        c->plog.clear();
        return 0;
    }
    const unsigned done_seq = w[3];
    std::deque<emx_ctx::PersistLog> redo;
    for (const auto& e : c->plog)
        if ((int)(e.seq - done_seq) > 0) redo.push_back(e);
    c->plog.clear();
    // Exact mode (round 5): the steps of the launches to redo have left the host pipeline, but it keeps the generator state behind
    // each of its last 64 + steps (NumPy get_state() semantics): the pipeline is retired at the state in front of the first such
    // launch and the steps are drawn again, by a new pipeline, on the per-half-step path.  Not possible (another pipeline since, or
    // too far back): as before -- the status bit stays, the run is void and says so.
    const bool exact_redo = !redo.empty() && redo.front().pipe_step0 >= 0;
    if (exact_redo && (c->pipe == nullptr || c->pipe != redo.front().pipe_id || c->pipe_taken - redo.front().pipe_step0 > 60 || c->cur.active))
        return 0;
    // the barrier words restart, the status bit is taken back: nothing is void
    HIPOK(c, hipMemsetAsync(c->persist_bar, 0, PERSIST_BAR_WORDS * sizeof(unsigned), c->stream));
    c->persist_epoch = 0;
    c->persist_lepoch = 0;
    c->persist_hepoch = 0;
    c->persist_grid = 0;
    __atomic_store_n(&c->status_host[3], 0u, __ATOMIC_RELEASE);
    if (redo.empty()) return 0;
    NEED(c, !c->cur.active, "persistent launches to redo, but a step is open");
    const int64_t stored_end = c->stored, proposals_end = c->proposals;
    const uint64_t ph_end = c->ph_step;
    for (const auto& m : c->moves) NEED(c, m.kind != EMX_MOVE_GAUSS, "persistent launches to redo in a mixture with a Gaussian move");
    drop_prepared(c);                         // (the ring slots of plans made ahead are about to be reused)
    if ((w[2] & (w[2] - 1u)) != 0u)
        c->tune_persist_local = 0;            // the one-XCD form's workgroups did not share an XCD: that form stays off ("persist_local" = 1)
    else
        c->tune_persist = 0;                  // and it stays off: whatever held the CUs may still be there ("persist" = 1 turns it back on)
    if (exact_redo) {
        MT19937Legacy before = c->mt;
        c->pipe->finish(redo.front().pipe_step0, before);
        delete c->pipe;
        c->pipe = nullptr;
        c->pipe_uploads.clear();
        c->pipe_deferred.clear();
        c->mt = before;                       // the next emx_step_begin starts a pipeline from here
    }
    c->ph_step = redo.front().ph_step;
    c->stored = redo.front().stored0;
    c->proposals = redo.front().proposals0;
    int rc = 0;
    for (const auto& e : redo) {
        NEED(c, (exact_redo || c->ph_step == e.ph_step) && c->stored == e.stored0, "persistent launches to redo are not consecutive");
        for (int64_t k = 0; k < e.steps && !rc; ++k) {
            const int st = e.store && ((e.i0 + k + 1) % e.thin_by == 0);          // ensemble.py:416
            int mvi, S;
            c->prep_hint = NATIVE_BATCH_MAX;
            rc = emx_step_begin(c, st, &mvi, &S);
            for (int sp = 0; sp < S && !rc; ++sp) rc = do_halfstep(c, sp, c->target);
            if (rc) c->cur.active = false;
            if (!rc) rc = emx_step_end(c);
        }
        if (rc) break;
        c->persist_recovered++;
    }
    if (rc) return rc;
    NEED(c, (exact_redo || c->ph_step == ph_end) && c->stored == stored_end && c->proposals == proposals_end, "redone steps do not add up");
    HIPOK(c, wait_stream(c->stream));
    return 0;
}

int emx_mtdev_info(emx_ctx* c, int64_t out[8]) {
    const MtDevStats& st = c->mtdev ? c->mtdev->stats() : c->mtdev_stats_last;
    out[0] = mtdev_eligible(c) ? 1 : 0;
    out[1] = c->mtdev ? 1 : 0;
    out[2] = c->mtdev_steps_total + c->mtdev_taken;
    out[3] = st.rounds;
    out[4] = st.segments;
    out[5] = st.batches;
    out[6] = st.windows;
    out[7] = (int64_t)(st.poly_ms * 1e3);
    return 0;
}

int emx_mtdev_tok_stats(emx_ctx* c, int64_t out[8]) {
    if (c->mtdev) c->mtdev->refresh_stats();
    const MtDevStats& st = c->mtdev ? c->mtdev->stats() : c->mtdev_stats_last;
    out[0] = st.windows;
    out[1] = st.tok_rounds;
    out[2] = st.tail_groups;
    out[3] = st.tail_rounds;
    for (int k = 0; k < 4; ++k) out[4 + k] = st.tok_ticks[k];
    return 0;
}

int emx_mtdev_debug(emx_ctx* c, int32_t what, int64_t arg, void* out, int64_t n) {
    NEED(c, c->mtdev != nullptr, "emx_mtdev_debug: no device producer is alive");
    int rc = -1;
    if (what == 0) {
        rc = c->mtdev->debug_stream((uint64_t)arg, n, (uint32_t*)out);
    } else if (what == 1) {
        NEED(c, n >= c->N, "emx_mtdev_debug: output too small");
        rc = c->mtdev->debug_targets(arg, (uint32_t*)out);
    } else if (what == 2) {
        const int S = c->moves[0].nsplits;
        NEED(c, n >= 3 * S + 1, "emx_mtdev_debug: output too small");
        rc = c->mtdev->debug_positions(arg, (uint64_t*)out, (uint64_t*)out + 3 * S);
    } else {
        FAIL(c, -1, "emx_mtdev_debug: unknown item %d", what);
    }
    if (rc) FAIL(c, rc, "%s", c->mtdev->error().c_str());
    return 0;
}

int emx_host_mt_jump(const uint32_t key[624], uint64_t stride_words, int32_t k, uint32_t out_key[624]) {
    if (k < 1 || stride_words < 1) return -1;
    const uint32_t* g = nullptr;
    if (!mt_jump_polys(stride_words, k, &g)) return -1;
    std::vector<uint32_t> win((size_t)MT_WINDOW + MT_N), cur(key, key + MT_N), nxt(MT_N);
    for (size_t have = 0; have < win.size(); have += MT_N) {
        mt_twist_block(cur.data(), nxt.data());
        cur.swap(nxt);
        memcpy(win.data() + have, cur.data(), std::min<size_t>(MT_N, win.size() - have) * 4);
    }
    mt_apply_jump(g + (size_t)(k - 1) * MT_POLY_WORDS, win.data(), out_key);
    return 0;
}

int emx_host_persist_shape(int64_t nwalkers, int32_t nsplits, int32_t num_cu, int32_t* waves_per_group, int32_t* groups) {
    const int wpb = persist_shape_of(nwalkers, nsplits, num_cu);
    *waves_per_group = wpb;
    *groups = wpb ? (int32_t)(nwalkers / nsplits / 16 / wpb) : 0;
    return 0;
}

int emx_persist_info(emx_ctx* c, int64_t out[4]) {
    const int64_t keep = c->call_steps;
    c->call_steps = (int64_t)1 << 40;          // "qualifies" is about an emx_run of many steps (exact mode: persist_exact_ok)
    out[0] = (persist_wanted(c) || persist_gauss_wanted(c) || persist_valu_wanted(c)) ? 1 : 0;
    c->call_steps = keep;
    out[1] = c->persist_launches;
    out[2] = c->persist_halfsteps;
    out[3] = c->persist_recovered;       // launches that gave up untouched and were redone on the per-half-step path (persist_settle)
    return 0;
}

int emx_persist_local_launches(emx_ctx* c, int64_t* n) {
    *n = c->persist_local_launches;
    return 0;
}

int emx_run(emx_ctx* c, int64_t nsteps, int32_t thin_by, int32_t store) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, thin_by >= 1, "Invalid thinning argument");
    NEED(c, !c->cur.active, "emx_run: a step begun with emx_step_begin is still open");
    const int64_t total = nsteps * thin_by;
    c->call_steps = total;
    // exact mode, general path: the plans of the whole call come from the host pipeline (generator / tokenizer /
    // finisher threads) instead of being made inline, one step at a time, by this thread
    const bool devp = mtdev_eligible(c);
    if (c->mtdev && !devp) {
        const int rcm = mtdev_stop(c);
        if (rcm) return rcm;
    }
    const bool piped = !devp && pipe_eligible(c) && c->target != EMX_TARGET_HOST && (c->pipe || total >= 2);
    if (c->pipe && !piped) PIPE_STOP(c);
    // A running pipeline was sized and configured for ONE consumer (pipe_start: the persistent launches' sixteen-step bursts and
    // finished plans, or step-at-a-time uploads with device finish, i.e. RAW generator words in the plan columns).  Whether the
    // persistent kernels take this call is asked per call -- target, tuning keys and the call's length decide -- so a pipeline
    // started for the other consumer is retired and started again (round-5 advisor: raw steps fetched by k_plan_fetch were read as
    // doubles).  The generator continues from the last step taken: the chain does not notice.
    if (c->pipe && persist_exact_ok(c) != (c->pipe_nsinks == PLAN_RING)) PIPE_STOP(c);
    if (piped && !c->pipe) {
        const int rc = pipe_start(c);
        if (rc) return rc;
    }
    const int rc = run_impl(c, nsteps, thin_by, store);
    c->call_steps = 1;                                       // (emx_step_begin on its own: one step at a time)
    if (rc && (c->pipe || c->mtdev)) pipe_stop(c);          // after an error the generator stands after the last plan taken
    return rc;
}

static int run_impl(emx_ctx* c, int64_t nsteps, int32_t thin_by, int32_t store) {
    NEED(c, thin_by >= 1, "Invalid thinning argument");
    NEED(c, c->rng_mode != EMX_RNG_INPUTS, "emx_run needs an RNG mode that generates plans");
    NEED(c, c->target != EMX_TARGET_HOST, "emx_run needs a device target");
    // the device-side replay exchange needs no collective library at all: mapped peers are enough
    const bool push_replay = c->exchange == EMX_EXCHANGE_REPLAY && c->peers_ready;
    NEED(c, (c->world == 1 && !c->sendbuf) || c->comm || push_replay || (c->world == 1 && c->exchange == EMX_EXCHANGE_REPLAY),
         "emx_run on a sharded context needs emx_comm_init (or drive emx_halfstep / the collective from the host layer)");
    NEED(c, c->exchange != EMX_EXCHANGE_DIRECT || c->world == 1 || c->peers_ready,
         "direct exchange: the peers' arrays are not mapped yet (emx_direct_export / emx_direct_import, or emx_direct_attach)");
    c->direct_first_barrier = true;
    if (store) NEED(c, c->stored + nsteps <= c->cap, "chain capacity exhausted (call emx_chain_config)");
    const int64_t total = nsteps * thin_by;
    // (padded ndim 16 with a stored chain is the one measured shape the persistent kernel loses on: +5 %, profiles/r03/persist_dims.txt)
    // (from 512 walkers on the one-XCD persistent kernel of the element-wise targets beats the one-workgroup kernel, whose step time
    // grows with the ensemble: 512 x 5: 5.2 -> 3.6 us/step, 1 024 x 5: 9.9 -> 3.7; profiles/r04/persist_valu.txt)
    const bool valu_on = persist_valu_wanted(c);
    const bool small_on = small_eligible(c) && !valu_on;
    const bool persist_on = (persist_wanted(c) && !small_on && !(store && c->Dp == 16)) || valu_on;
    const bool persist_gauss_on = persist_gauss_wanted(c) && !small_on;
    bool ctr_synced = false;     // device-side graph counters equal the host's (ph_step, stored)
    int64_t next_mark = c->tune_throttle > 0 ? c->tune_throttle : total + 1;
    int marks = 0;
    for (int64_t i = 0; i < total;) {
        if (i >= next_mark) {
            // window boundary: mark it, and do not run more than two windows ahead of the device
            const int slot = marks & 3;
            if (!c->thr_ev[slot]) HIPOK(c, hipEventCreateWithFlags(&c->thr_ev[slot], hipEventDisableTiming));
            HIPOK(c, hipEventRecord(c->thr_ev[slot], c->stream));
            if (marks >= 2) {
                hipEvent_t old = c->thr_ev[(marks - 2) & 3];
                while (hipEventQuery(old) == hipErrorNotReady) {
                }
            }
            ++marks;
            next_mark += c->tune_throttle;
        }
        if (small_on) {
            // the ensemble fits one CU's LDS: up to 4096 steps per launch inside one workgroup
            drop_prepared(c);
            int64_t chunk = std::min<int64_t>(total - i, 4096);
            if (c->rng_mode == EMX_RNG_MT19937)      // plans travel: <= 4 MB per launch, short enough to overlap host and GPU
                chunk = std::min<int64_t>(chunk, std::max<int64_t>(8, (4 << 20) / (32 * c->N)));
            const int rc = run_small(c, i, chunk, thin_by, store);
            if (rc) return rc;
            i += chunk;
            ctr_synced = false;
            continue;
        }
        if (persist_gauss_on) {
            int64_t done = 0;
            const int rc = run_persist_gauss(c, i, total, thin_by, store, &done);
            if (rc) return rc;
            i += done;
            ctr_synced = false;
            continue;
        }
        if (persist_on && c->tune_persist && c->rng_mode == EMX_RNG_MT19937) {
            // exact mode, plans from the device producer: one move, a launch per produced batch
            if (persist_move_ok(c, c->moves[0])) {
                int64_t done = 0;
                const int rc = run_persist(c, i, total, thin_by, store, &done);
                if (rc) return rc;
                i += done;
                ctr_synced = false;
                if (done > 0) continue;
            }
        } else if (persist_on && c->tune_persist) {
            // the next step's move decides (a mixture: runs of steps of one move the persistent kernel knows, the others one by one)
            if (!c->prepared.empty() && c->prepared.front().step != c->ph_step) drop_prepared(c);
            if (c->prepared.empty()) {
                const int rcp = native_prepare_batch(c, c->ph_step, NATIVE_BATCH_MAX, -1, c->stream);
                if (rcp) return rcp;
            }
            if (persist_move_ok(c, c->moves[c->prepared.front().move])) {
                int64_t done = 0;
                const int rc = run_persist(c, i, total, thin_by, store, &done);
                if (rc) return rc;
                i += done;
                ctr_synced = false;
                if (done > 0) continue;
                // (nothing done: launches that gave up were just redone and the persistent path is off for this context -- fall through)
            }
        }
        if (thin_by == 1 && total - i >= NATIVE_BATCH_MAX) {
            emx_ctx::GraphSlot* g = graph_ready(c, store);
            if (g) {
                if (!ctr_synced) {
                    hipLaunchKernelGGL(k_graph_set, dim3(1), dim3(1), 0, c->stream, c->d_ctr, (unsigned long long)c->ph_step,
                                       (long long)c->stored);
                    ctr_synced = true;
                }
                HIPOK(c, hipGraphLaunch(g->exec, c->stream));      // one block: plan + NATIVE_BATCH_MAX x nsplits half-steps
                c->ph_step += NATIVE_BATCH_MAX;
                c->proposals += NATIVE_BATCH_MAX;
                if (store) c->stored += NATIVE_BATCH_MAX;
                i += NATIVE_BATCH_MAX;
                continue;
            }
        }
        {
            const int st = store && ((i + 1) % thin_by == 0);                   // ensemble.py:416
            int mvi, S;
            // native plans are made NATIVE_BATCH_MAX steps at a time whatever is left of THIS call: they do not depend on the
            // walkers, what a short call leaves over serves the next one (a 20-step call: 16 + 16 with 12 carried over, not
            // 16 + 4 -- the launch is the same size either way)
            c->prep_hint = std::max<int64_t>(total - i, NATIVE_BATCH_MAX);
            if (thin_by == 1 && !c->graph_disabled && c->tune_graph && !c->graph_warm) c->prep_hint = 1;   // graph follows
            int rc = emx_step_begin(c, st, &mvi, &S);
            if (rc) return rc;
            for (int s = 0; s < S; ++s) {
                if (c->exchange == EMX_EXCHANGE_DIRECT && c->world > 1) {
                    // partner rows are read in place from the peers' HBM: a device-side barrier, then the half-step
                    rc = emx_direct_halfstep(c, s, 1);
                    if (rc) {
                        c->cur.active = false;
                        return rc;
                    }
                    continue;
                }
                if (c->comm && c->exchange == EMX_EXCHANGE_LOGPROB) {
                    // proposal and commit on every rank, the log-probs of a share each, 8 bytes per walker gathered in place
                    int64_t per = 0;
                    rc = emx_logprob_begin(c, s, &per);
                    if (!rc && per > 0) {
                        const int e = g_rccl.AllGather(c->gathered + (size_t)c->rank * per, c->gathered, (size_t)per, RCCL_FLOAT64, c->comm,
                                                       c->stream);
                        if (e != 0) {
                            c->err = std::string("ncclAllGather failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
                            rc = -6;
                        }
                    }
                    if (!rc) rc = emx_logprob_finish(c, s);
                    if (rc) {
                        c->cur.active = false;
                        return rc;
                    }
                    continue;
                }
                if ((c->comm || push_replay) && c->exchange == EMX_EXCHANGE_REPLAY) {
                    // own slots fused; 8 bytes of decision per walker-update gathered; the others' accepted updates replayed
                    int64_t rows = 0;
                    rc = emx_replay_begin(c, s, &rows);
                    if (!rc && c->peers_ready && c->world > 1) {
                        rc = emx_replay_exchange(c, s);          // stores into the peers' buffers + the device-side barrier
                    } else if (!rc && rows > 0 && c->world > 1) {
                        const int e = g_rccl.AllGather(c->sendbuf, c->gathered, (size_t)rows, RCCL_FLOAT64, c->comm, c->stream);
                        if (e != 0) {
                            c->err = std::string("ncclAllGather failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
                            rc = -6;
                        }
                    }
                    if (!rc) rc = emx_replay_finish(c, s);
                    if (rc) {
                        c->cur.active = false;
                        return rc;
                    }
                    continue;
                }
                if (c->comm && c->exchange == EMX_EXCHANGE_PULL) {
                    // partner rows only: pack what the peers will read, all-to-all, fold in, update own walkers
                    int64_t cap = 0;
                    rc = emx_pull_prepare(c, s, &cap);
                    if (!rc) rc = rccl_all_to_all(c, (size_t)cap * (c->D + 1));
                    if (!rc) rc = emx_pull_apply(c, s);
                    if (rc) {
                        c->cur.active = false;
                        return rc;
                    }
                    continue;
                }
                rc = do_halfstep(c, s, c->target);
                if (!rc && c->comm) {
                    // the one exchange per half-step: every rank's [row | log_prob | accepted] records to every rank
                    // only the records this half-step filled travel: a rank's share of the sub-ensemble
                    const int64_t ns_cur = c->cur.off[s + 1] - c->cur.off[s];
                    const int64_t rows = std::min<int64_t>(c->sendbuf_rows, (ns_cur + c->world - 1) / c->world);
                    const int e = g_rccl.AllGather(c->sendbuf, c->gathered, (size_t)rows * (c->D + 2), RCCL_FLOAT64,
                                                   c->comm, c->stream);
                    if (e != 0) {
                        c->err = std::string("ncclAllGather failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
                        rc = -6;
                    }
                    if (!rc && c->world > 1) rc = scatter_gathered(c, s, rows);
                }
                if (rc) {
                    c->cur.active = false;
                    return rc;
                }
            }
            rc = emx_step_end(c);
            if (rc) return rc;
            c->graph_warm = true;
            ctr_synced = false;
            ++i;
        }
    }
    c->prep_hint = 1;
    if (c->comm && (c->exchange == EMX_EXCHANGE_PULL || c->exchange == EMX_EXCHANGE_DIRECT) && c->world > 1) {
        // re-synchronise the replicas: every rank's block of (coords, log_prob, accepted) to every rank
        int64_t per = 0;
        int rc = emx_replica_pack(c, &per);
        if (rc) return rc;
        const int e = g_rccl.AllGather(c->sendbuf, c->gathered, (size_t)per * (c->D + 3), RCCL_FLOAT64, c->comm, c->stream);
        if (e != 0) FAIL(c, -6, "ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
        rc = emx_replica_unpack(c);
        if (rc) return rc;
    }
    return 0;
}

// ---- sharding ----------------------------------------------------------------------------
// records one rank contributes to the all-gather of a half-step: its share of the largest sub-ensemble
// (ceil(N / nsplits) walkers; the Gaussian move's single "split" is the whole ensemble)
static int64_t shard_rows_per_rank(int64_t N, int world, int min_nsplits = 2) {
    const int64_t maxns = (N + min_nsplits - 1) / min_nsplits + 1;
    return (maxns + world - 1) / world + 1;
}

static int min_nsplits_of(const emx_ctx* c) {
    int m = 2;
    for (const auto& mv : c->moves) m = std::min(m, (int)mv.nsplits);
    return std::max(1, m);
}

static int partners_of(int kind) {
    return kind == EMX_MOVE_GAUSS ? 0 : kind == EMX_MOVE_STRETCH ? 1 : kind == EMX_MOVE_DE ? 2 : 3;
}

// Pull exchange: records one rank may have to send another in one half-step.  Each of the ~N/(S G) walkers
// a rank updates reads `npart` partners whose owner is uniform over the ranks: mean + 8 sigma + slack,
// never more than every partner of every walker the destination could be updating.
static int64_t pull_capacity(int64_t N, int G, int S, int npart) {
    if (G <= 1) return 1;
    const int64_t bmax = (N + G - 1) / G, nsmax = (N + S - 1) / S;
    const int64_t hard = (int64_t)npart * std::min(bmax, nsmax);
    const double mean = (double)npart * (double)N / S / G / G;
    const int64_t cap = (int64_t)std::ceil(mean + 8.0 * std::sqrt(mean) + 64.0);
    return std::max<int64_t>(1, std::min(cap, hard));
}

static int64_t pull_capacity_max(const emx_ctx* c) {
    int64_t cap = 1;
    for (const auto& m : c->moves) cap = std::max(cap, pull_capacity(c->N, c->world, m.nsplits, partners_of(m.kind)));
    return cap;
}

static void exchange_free(emx_ctx* c) {
    if (c->own_shard_bufs) {
        if (c->sendbuf) hipFree(c->sendbuf);
        if (c->gathered) hipFree(c->gathered);
    }
    c->sendbuf = c->gathered = nullptr;
    c->own_shard_bufs = false;
    c->sendbuf_rows = c->gathered_rows = 0;
    c->send_doubles = c->recv_doubles = 0;
}

static void pull_layout(const emx_ctx* c, int64_t& send, int64_t& recv) {
    const int64_t G = c->world, bmax = (c->N + G - 1) / G;
    const int64_t pairs = G * pull_capacity_max(c) * (c->D + 1);
    send = std::max<int64_t>(pairs, bmax * (c->D + 3));
    recv = std::max<int64_t>(pairs, G * bmax * (c->D + 3));
}

// (re)allocate what the pull exchange needs for the installed moves; own exchange buffers grow on demand,
// caller-owned ones must already be large enough
static int pull_ensure(emx_ctx* c) {
    const int64_t G = c->world, bmax = (c->N + G - 1) / G;
    if (c->cplan_rows < bmax) {
        HIPOK(c, hipStreamSynchronize(c->stream));
        auto& p = c->cplan;
        void* old[] = {p.order, p.p0, p.p1, p.p2, p.s0, p.uacc, p.logu, p.fac};
        for (void* q : old)
            if (q) hipFree(q);
        HIPOK(c, hipMalloc((void**)&p.order, bmax * 4));
        HIPOK(c, hipMalloc((void**)&p.p0, bmax * 4));
        HIPOK(c, hipMalloc((void**)&p.p1, bmax * 4));
        HIPOK(c, hipMalloc((void**)&p.p2, bmax * 4));
        HIPOK(c, hipMalloc((void**)&p.s0, bmax * 8));
        HIPOK(c, hipMalloc((void**)&p.uacc, bmax * 8));
        HIPOK(c, hipMalloc((void**)&p.logu, bmax * 8));
        HIPOK(c, hipMalloc((void**)&p.fac, bmax * 8));
        c->cplan_rows = bmax;
    }
    if (!c->pull_counts) {
        const size_t n = (size_t)(2 * (1 + EMX_MAX_RANKS)) * 4;
        HIPOK(c, hipMalloc((void**)&c->pull_counts, n));
        HIPOK(c, hipMemset(c->pull_counts, 0, n));
        c->pull_parity = 0;
    }
    int64_t send, recv;
    pull_layout(c, send, recv);
    if (c->send_doubles < send || c->recv_doubles < recv) {
        NEED(c, c->own_shard_bufs || !c->sendbuf,
             "pull exchange: caller-owned buffers too small for the installed moves (need %lld / %lld doubles)",
             (long long)send, (long long)recv);
        HIPOK(c, hipStreamSynchronize(c->stream));
        exchange_free(c);
        HIPOK(c, hipMalloc((void**)&c->sendbuf, (size_t)send * 8));
        HIPOK(c, hipMalloc((void**)&c->gathered, (size_t)recv * 8));
        c->own_shard_bufs = true;
        c->send_doubles = send;
        c->recv_doubles = recv;
    }
    return 0;
}

int emx_set_exchange(emx_ctx* c, int32_t kind) {
    NEED(c, kind == EMX_EXCHANGE_ALLGATHER || kind == EMX_EXCHANGE_PULL || kind == EMX_EXCHANGE_DIRECT || kind == EMX_EXCHANGE_LOGPROB ||
                kind == EMX_EXCHANGE_REPLAY,
         "unknown exchange kind %d", kind);
    NEED(c, c->world == 1 && !c->comm && !c->sendbuf, "emx_set_exchange: call it before emx_set_shard / emx_comm_init");
    drop_prepared(c);          // plans made ahead were shaped (lean or full) for the previous configuration
    c->exchange = kind;
    return 0;
}

int emx_set_shard(emx_ctx* c, int32_t rank, int32_t world) {
    { const int rcs_ = persist_settle(c); if (rcs_) return rcs_; }
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, world >= 1 && rank >= 0 && rank < world, "bad (rank, world)");
    NEED(c, world <= c->N, "more ranks than walkers");
    HIPOK(c, hipStreamSynchronize(c->stream));
    drop_prepared(c);          // plans made ahead were shaped (lean or full) for the previous sharding
    c->rank = rank;
    c->world = world;
    exchange_free(c);
    c->pull_split = -1;
    if (c->exchange == EMX_EXCHANGE_PULL) return world > 1 ? pull_ensure(c) : 0;
    if (c->exchange == EMX_EXCHANGE_LOGPROB) {
        // the one buffer of this exchange: the log-probs of a split's proposals, world shares of ceil(ns / world) doubles,
        // gathered in place
        HIPOK(c, hipMalloc((void**)&c->gathered, (size_t)(c->N + world) * 8));
        c->own_shard_bufs = true;
        c->recv_doubles = c->N + world;
        return 0;
    }
    if (c->exchange == EMX_EXCHANGE_DIRECT) {
        NEED(c, world <= EMX_MAX_PEERS, "direct exchange: at most %d ranks (the GPUs of one node)", EMX_MAX_PEERS);
        direct_detach(c);
        return direct_ensure(c);
    }
    if (c->exchange == EMX_EXCHANGE_REPLAY) return replay_ensure(c);
    if (world > 1) {
        const int64_t per = shard_rows_per_rank(c->N, world, min_nsplits_of(c));
        HIPOK(c, hipMalloc((void**)&c->sendbuf, (size_t)per * (c->D + 2) * 8));
        HIPOK(c, hipMalloc((void**)&c->gathered, (size_t)per * world * (c->D + 2) * 8));
        c->own_shard_bufs = true;
        c->sendbuf_rows = per;
        c->gathered_rows = per * world;
        c->send_doubles = per * (c->D + 2);
        c->recv_doubles = per * world * (c->D + 2);
    }
    return 0;
}

int emx_set_shard_buffers(emx_ctx* c, void* sendbuf, void* gathered, int64_t rows_per_rank) {
    NEED(c, c->world >= 1, "emx_set_shard_buffers: call emx_set_shard first");
    NEED(c, c->exchange == EMX_EXCHANGE_ALLGATHER, "pull exchange: use emx_set_exchange_buffers");
    NEED(c, rows_per_rank >= shard_rows_per_rank(c->N, c->world, min_nsplits_of(c)), "shard buffers too small");
    HIPOK(c, hipStreamSynchronize(c->stream));
    exchange_free(c);
    c->sendbuf = (double*)sendbuf;
    c->gathered = (double*)gathered;
    c->sendbuf_rows = rows_per_rank;
    c->gathered_rows = rows_per_rank * c->world;
    c->send_doubles = rows_per_rank * (c->D + 2);
    c->recv_doubles = c->gathered_rows * (c->D + 2);
    return 0;
}

int emx_exchange_layout(emx_ctx* c, int64_t* send_doubles, int64_t* recv_doubles) {
    if (c->exchange == EMX_EXCHANGE_PULL) {
        pull_layout(c, *send_doubles, *recv_doubles);
    } else if (c->exchange == EMX_EXCHANGE_REPLAY) {
        const int64_t per = shard_rows_per_rank(c->N, c->world, min_nsplits_of(c));      // one double per walker-update
        *send_doubles = per;
        *recv_doubles = per * c->world;
    } else {
        const int64_t per = shard_rows_per_rank(c->N, c->world, min_nsplits_of(c));
        *send_doubles = per * (c->D + 2);
        *recv_doubles = per * c->world * (c->D + 2);
    }
    return 0;
}

int emx_set_exchange_buffers(emx_ctx* c, void* send, int64_t send_doubles, void* recv, int64_t recv_doubles) {
    NEED(c, c->exchange == EMX_EXCHANGE_PULL, "emx_set_exchange_buffers is for the pull exchange");
    int64_t ns, nr;
    pull_layout(c, ns, nr);
    NEED(c, send && recv && send_doubles >= ns && recv_doubles >= nr, "exchange buffers too small (need %lld / %lld doubles)",
         (long long)ns, (long long)nr);
    HIPOK(c, hipStreamSynchronize(c->stream));
    exchange_free(c);
    c->sendbuf = (double*)send;
    c->gathered = (double*)recv;
    c->send_doubles = send_doubles;
    c->recv_doubles = recv_doubles;
    return 0;
}

int emx_own_walkers(emx_ctx* c, int64_t* lo, int64_t* hi) {
    *lo = c->N * c->rank / c->world;
    *hi = c->N * (c->rank + 1) / c->world;
    return 0;
}

// ---- log-prob exchange: the reference's pool.map model (ensemble.py:486-496) --------------------------------------
// Every rank proposes for ALL walkers of the split (replicated plan, replicated ensemble: identical proposals), evaluates
// the target on its share of them, the shares are gathered in place (8 bytes per walker), and every rank takes the
// decisions and commits ALL walkers -- the replicas stay identical, no coordinate ever crosses xGMI.
int emx_logprob_begin(emx_ctx* c, int32_t split, int64_t* per_rank) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, c->exchange == EMX_EXCHANGE_LOGPROB && c->gathered, "emx_logprob_begin needs emx_set_exchange(EMX_EXCHANGE_LOGPROB) and emx_set_shard");
    NEED(c, cur.active && cur.move >= 0 && cur.slot >= 0, "emx_logprob_begin outside a planned step");
    NEED(c, split >= 0 && split < cur.S, "split out of range");
    NEED(c, c->target != EMX_TARGET_HOST, "log-prob exchange needs a device target");
    const emx_move_desc& mv = c->moves[cur.move];
    const int pos0 = cur.off[split], ns = cur.off[split + 1] - cur.off[split];
    const int64_t per = ((int64_t)ns + c->world - 1) / c->world;
    *per_rank = per;
    if (ns <= 0) return 0;
    const emx_ctx::PlanSlot* ps = &c->ring[cur.slot];
    int rc = launch_split(c, mv.kind, EMX_TARGET_HOST, cur.S, split, pos0, ns, 0, ns, &mv, ps, nullptr, c->X, c->lp, nullptr, nullptr,
                          nullptr);
    if (rc) return rc;
    const int lo = (int)std::min<int64_t>((int64_t)c->rank * per, ns), hi = (int)std::min<int64_t>(lo + per, ns);
    if (hi <= lo) return 0;
    // log-probs of the proposals [lo, hi): row t of qout -> gathered[t].  A non-finite proposal was reported by the propose pass
    // above -- on EVERY rank, they all propose everything -- and gets -inf here (rejected), exactly as on the single-rank wide
    // path: no rank sees a status (ST_NAN_LOGP) the others do not
    c->eval_check_bad = true;
    rc = launch_split(c, MOVE_EVAL, c->target, 1, 0, 0, ns, lo, hi, &mv, nullptr, c->iota, c->qout, c->gathered, nullptr, nullptr,
                      nullptr);
    c->eval_check_bad = false;
    return rc;
}

int emx_logprob_finish(emx_ctx* c, int32_t split) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, c->exchange == EMX_EXCHANGE_LOGPROB && c->gathered, "emx_logprob_finish needs emx_set_exchange(EMX_EXCHANGE_LOGPROB) and emx_set_shard");
    NEED(c, cur.active && cur.move >= 0 && cur.slot >= 0 && split >= 0 && split < cur.S, "emx_logprob_finish outside a planned step");
    const int pos0 = cur.off[split], ns = cur.off[split + 1] - cur.off[split];
    if (ns <= 0) return 0;
    const emx_ctx::PlanSlot* ps = &c->ring[cur.slot];
    WideCommitArgs k{};
    k.X = c->X;
    k.lp = c->lp;
    k.acc = c->acc;
    k.acc_count = c->acc_count;
    if (cur.store) {
        k.chain = c->chain + (size_t)c->stored * c->N * c->D;
        k.chain_lp = c->chain_lp + (size_t)c->stored * c->N;
    }
    k.qout = c->qout;
    k.fout = c->fout;
    k.newlp = c->gathered;
    k.order = ps->order;
    k.logu = ps->logu;
    k.D = c->D;
    k.pos0 = pos0;
    k.t_lo = 0;
    k.t_hi = ns;
    if (launch_wide_commit(k, ns, c->num_cu, c->stream) != hipSuccess) FAIL(c, -2, "log-prob exchange: commit kernel launch failed");
    return 0;
}

// ---- replay exchange (emx_kernels.hpp "Replay exchange"): decisions travel, accepted updates are recomputed ---------------
static int replay_ensure(emx_ctx* c) {
    const int64_t N = c->N;
    if (c->cplan_rows < N) {          // compact plan of the accepted foreign slots of one half-step (at most a whole split)
        HIPOK(c, hipStreamSynchronize(c->stream));
        auto& p = c->cplan;
        void* old[] = {p.order, p.p0, p.p1, p.p2, p.s0, p.uacc, p.logu, p.fac};
        for (void* q : old)
            if (q) hipFree(q);
        HIPOK(c, hipMalloc((void**)&p.order, N * 4));
        HIPOK(c, hipMalloc((void**)&p.p0, N * 4));
        HIPOK(c, hipMalloc((void**)&p.p1, N * 4));
        HIPOK(c, hipMalloc((void**)&p.p2, N * 4));
        HIPOK(c, hipMalloc((void**)&p.s0, N * 8));
        HIPOK(c, hipMalloc((void**)&p.uacc, N * 8));
        HIPOK(c, hipMalloc((void**)&p.logu, N * 8));
        HIPOK(c, hipMalloc((void**)&p.fac, N * 8));
        HIPOK(c, hipMemset(p.fac, 0, N * 8));
        c->cplan_rows = N;
    }
    if (!c->replay_counts) {
        HIPOK(c, hipMalloc((void**)&c->replay_counts, 2 * 4));
        HIPOK(c, hipMemset(c->replay_counts, 0, 2 * 4));
        c->replay_parity = 0;
    }
    const int64_t per = c->world > 1 ? shard_rows_per_rank(N, c->world, min_nsplits_of(c)) : N + 2;
    if (c->send_doubles < per || c->recv_doubles < 2 * per * c->world) {
        NEED(c, !c->peers_ready, "replay exchange: the installed moves need larger buffers than the ones the peers have mapped: install "
                                 "the moves before emx_set_shard / emx_comm_init, or attach the peers again");
        HIPOK(c, hipStreamSynchronize(c->stream));
        exchange_free(c);
        HIPOK(c, hipMalloc((void**)&c->sendbuf, (size_t)per * 8));
        // two receive buffers (the device-side exchange alternates between them); peers may write them over xGMI while this
        // device reads: fine-grained (never stale in the local L2) where the runtime offers it
        void* g = nullptr;
        if (hipExtMallocWithFlags(&g, (size_t)2 * per * c->world * 8, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIPOK(c, hipMalloc(&g, (size_t)2 * per * c->world * 8));
        }
        c->gathered = (double*)g;
        c->own_shard_bufs = true;
        c->sendbuf_rows = per;
        c->gathered_rows = per * c->world;
        c->send_doubles = per;
        c->recv_doubles = 2 * per * c->world;
        c->replay_buf = per * c->world;
        c->replay_recv_off = 0;
        c->replay_push_parity = 0;
    }
    return flags_ensure(c);
}

// Device-side exchange of the decisions (needs the peers' receive buffers and flags: emx_direct_export / _import / _attach):
// push to every peer, then the barrier.  emx_replay_finish then reads this half-step's buffer.
int emx_replay_exchange(emx_ctx* c, int32_t split) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, c->exchange == EMX_EXCHANGE_REPLAY && cur.active && c->replay_split == split, "emx_replay_exchange: emx_replay_begin(%d) has not run", split);
    NEED(c, c->peers_ready || c->world == 1, "replay exchange: the peers' receive buffers are not mapped (emx_direct_export / emx_direct_import, "
                                             "or emx_direct_attach); without them use an all-gather of the send buffer");
    NEED(c, !c->direct_dead, "replay exchange: an earlier barrier timed out (a peer never arrived): attach the peers again");
    NEED(c, c->world <= EMX_MAX_PEERS, "device-side exchange: at most %d ranks (the GPUs of one node)", EMX_MAX_PEERS);
    const int ns = cur.off[split + 1] - cur.off[split];
    const int64_t rows = ((int64_t)ns + c->world - 1) / c->world;
    const int64_t G = c->world;
    if (G == 1 || rows <= 0) return 0;
    if (!c->replay_pushed) {
        // the own pass was not a fused launch that stores into the peers itself (three-pass targets; an empty share): push now.
        // (emx_replay_begin chose this half-step's receive buffer.)
        PushArgs a{};
        a.src = c->sendbuf;
        for (int q = 0; q < G; ++q) a.peer[q] = c->peerX[q];
        a.off = c->replay_recv_off + (int64_t)c->rank * rows;
        a.rows = (int32_t)rows;
        a.npeer = (int32_t)G;
        hipLaunchKernelGGL(k_push_decisions, dim3((unsigned)((rows + 255) / 256), (unsigned)G), dim3(256), 0, c->stream, a);
        HIPOK(c, hipGetLastError());
    }
    PeerBarrierArgs b{};
    for (int q = 0; q < G; ++q) {
        NEED(c, c->peer_flags[q], "replay exchange: no barrier flags for rank %d", q);
        b.peer_flags[q] = c->peer_flags[q];
    }
    b.my_flags = c->my_flags;
    b.dead = c->direct_counts + 64;
    b.status = c->status;
    b.epoch = ++c->direct_epoch;
    b.timeout_ticks = (unsigned long long)c->tune_direct_timeout_ms * (c->direct_first_barrier ? 6ull : 1ull) * 100000ull;
    c->direct_first_barrier = false;
    b.rank = c->rank;
    b.npeer = (int32_t)G;
    hipLaunchKernelGGL(k_peer_barrier, dim3(1), dim3(64), 0, c->stream, b);
    HIPOK(c, hipGetLastError());
    return 0;
}

int emx_replay_begin(emx_ctx* c, int32_t split, int64_t* rows_per_rank) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, c->exchange == EMX_EXCHANGE_REPLAY, "emx_replay_begin needs emx_set_exchange(EMX_EXCHANGE_REPLAY)");
    NEED(c, cur.active && cur.move >= 0 && cur.slot >= 0, "emx_replay_begin outside a planned step");
    NEED(c, split >= 0 && split < cur.S, "split out of range");
    NEED(c, c->target != EMX_TARGET_HOST, "sharded stepping needs a device target");
    int rc = replay_ensure(c);
    if (rc) return rc;
    const emx_move_desc& mv = c->moves[cur.move];
    const int pos0 = cur.off[split], ns = cur.off[split + 1] - cur.off[split];
    const int64_t rows = ((int64_t)ns + c->world - 1) / c->world;
    NEED(c, rows <= c->sendbuf_rows, "replay exchange: buffers too small for this move (call emx_set_shard after emx_set_moves)");
    if (rows_per_rank) *rows_per_rank = rows;
    c->replay_split = split;
    c->replay_recv_off = 0;          // an all-gather fills the first receive buffer; the device-side exchange alternates two
    c->replay_pushed = false;
    const bool device_side = c->peers_ready && c->world > 1;
    if (device_side) {
        c->replay_recv_off = (int64_t)c->replay_push_parity * c->replay_buf;
        c->replay_push_parity ^= 1;
    }
    int64_t lo, hi;
    shard_range(ns, c->rank, c->world, lo, hi);
    if (hi <= lo) return 0;
    c->launch_push = device_side;
    c->launch_push_off = c->replay_recv_off + (int64_t)c->rank * rows;
    // own slots, fused: the plan is entered at this rank's first slot, so the launch sees slots [0, hi - lo) -- the form the
    // production (LEAN) instantiations take -- and decision e of the launch is slot lo + e
    c->launch_declp = c->sendbuf;
    rc = launch_split(c, mv.kind, c->target, cur.S, split, pos0 + (int)lo, (int)(hi - lo), 0, (int)(hi - lo), &mv, &c->ring[cur.slot], nullptr,
                      c->X, c->lp, nullptr, nullptr, nullptr);
    c->launch_declp = nullptr;
    c->launch_push = false;
    return rc;
}

int emx_replay_finish(emx_ctx* c, int32_t split) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, c->exchange == EMX_EXCHANGE_REPLAY && cur.active && c->replay_split == split, "emx_replay_finish: emx_replay_begin(%d) has not run", split);
    c->replay_split = -1;
    const emx_move_desc& mv = c->moves[cur.move];
    const auto& ps = c->ring[cur.slot];
    const int pos0 = cur.off[split], ns = cur.off[split + 1] - cur.off[split];
    int64_t lo, hi;
    shard_range(ns, c->rank, c->world, lo, hi);
    const int64_t foreign = ns - (hi - lo);
    bool done = false;
    if (foreign > 0 && mv.kind == EMX_MOVE_STRETCH && !c->tune_replay_two_pass) {
        // the stretch move: flags + replay in one launch over the full plan (k_replay_stretch)
        ReplayFusedArgs f{};
        f.X = c->X;
        f.lp = c->lp;
        f.acc = c->acc;
        f.order = ps.order + pos0;
        f.p0 = ps.p0 + pos0;
        f.s0 = ps.s0 + pos0;
        f.gathered = c->gathered + c->replay_recv_off;
        f.ns = ns;
        f.G = c->world;
        f.rank = c->rank;
        f.rows = (int32_t)(((int64_t)ns + c->world - 1) / c->world);
        f.D = c->D;
        const bool dense_layout = c->target == EMX_TARGET_DENSE_GAUSS && !dense_is_wide(c);      // the layout that took the decisions
        const Shape sh = pick_shape(c->D, dense_layout ? c->Dp : c->D);
        const int64_t nchunks = ((int64_t)ns + 63) / 64;
        const hipError_t e = launch_replay_stretch(sh, dim3((unsigned)((nchunks + 3) / 4)), c->stream, f);
        if (e == hipSuccess)
            done = true;
        else if (e != hipErrorInvalidValue)
            FAIL(c, -2, "replay launch failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
    }
    if (foreign > 0 && !done) {
        ReplayCompactArgs a{};
        a.order = ps.order + pos0;
        a.p0 = ps.p0 + pos0;
        a.p1 = ps.p1 + pos0;
        a.p2 = ps.p2 + pos0;
        a.s0 = ps.s0 + pos0;
        a.gathered = c->gathered + c->replay_recv_off;
        a.corder = c->cplan.order;
        a.cp0 = c->cplan.p0;
        a.cp1 = c->cplan.p1;
        a.cp2 = c->cplan.p2;
        a.cs0 = c->cplan.s0;
        a.clogu = c->cplan.logu;
        a.count = c->replay_counts + c->replay_parity;
        a.count_next = c->replay_counts + (c->replay_parity ^ 1);
        a.acc = c->acc;
        a.ns = ns;
        a.G = c->world;
        a.rank = c->rank;
        a.rows = (int32_t)(((int64_t)ns + c->world - 1) / c->world);
        hipLaunchKernelGGL(k_replay_compact, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, c->stream, a);
        HIPOK(c, hipGetLastError());
        const int32_t* count_now = a.count;
        c->replay_parity ^= 1;
        // grid for the worst case (every foreign proposal accepted); the count is read on the device, idle waves leave at once
        const int rc = launch_split(c, mv.kind, TGT_REPLAY, cur.S, split, 0, (int)foreign, 0, (int)foreign, &mv, &c->cplan, nullptr, c->X,
                                    c->lp, nullptr, nullptr, nullptr, nullptr, count_now);
        if (rc) return rc;
    }
    if (cur.store && split == cur.S - 1) {
        StoreStepArgs s{};
        s.X = c->X;
        s.lp = c->lp;
        s.acc = c->acc;
        s.acc_count = c->acc_count;
        s.chain = c->chain + (size_t)c->stored * c->N * c->D;
        s.chain_lp = c->chain_lp + (size_t)c->stored * c->N;
        s.nx = (long long)c->N * c->D;
        s.N = (int32_t)c->N;
        const long long nb = std::min<long long>((s.nx + 255) / 256, (long long)c->num_cu * 16);
        hipLaunchKernelGGL(k_store_step, dim3((unsigned)std::max<long long>(1, nb)), dim3(256), 0, c->stream, s);
        HIPOK(c, hipGetLastError());
    }
    return 0;
}

int emx_pull_prepare(emx_ctx* c, int32_t split, int64_t* records_per_peer) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, c->exchange == EMX_EXCHANGE_PULL, "emx_pull_prepare needs emx_set_exchange(EMX_EXCHANGE_PULL)");
    NEED(c, cur.active && cur.move >= 0 && cur.slot >= 0, "emx_pull_prepare outside a planned step");
    NEED(c, split >= 0 && split < cur.S, "split out of range");
    NEED(c, c->target != EMX_TARGET_HOST, "sharded stepping needs a device target");
    int rc = pull_ensure(c);
    if (rc) return rc;
    rc = complete_lean_plan(c);         // the compact plan copies every column (a plan made before the context was sharded may be lean)
    if (rc) return rc;
    const emx_move_desc& mv = c->moves[cur.move];
    const auto& ps = c->ring[cur.slot];
    const int pos0 = cur.off[split], ns = cur.off[split + 1] - cur.off[split];
    const int npart = partners_of(mv.kind);
    // one capacity for every half-step of the context (the largest an installed move needs): records keep their address
    const int64_t cap = pull_capacity_max(c);
    NEED(c, pull_capacity(c->N, c->world, cur.S, npart) <= cap, "pull exchange: capacity below this move's need");
    NEED(c, c->world <= EMX_MAX_RANKS, "pull exchange: at most %d ranks", EMX_MAX_RANKS);
    int32_t* counts = c->pull_counts + (size_t)c->pull_parity * (1 + EMX_MAX_RANKS);
    if (c->pull_cap_armed != cap || c->pull_armed_buf != c->sendbuf) {
        const int nrec = (int)(c->world * cap);
        hipLaunchKernelGGL(k_pull_reset, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->stream, c->sendbuf, nrec, c->D);
        HIPOK(c, hipGetLastError());
        HIPOK(c, hipMemsetAsync(c->pull_counts, 0, (size_t)(2 * (1 + EMX_MAX_RANKS)) * 4, c->stream));
        c->pull_cap_armed = cap;
        c->pull_armed_buf = c->sendbuf;
    }
    if (ns > 0) {
        PullPlanArgs a{};
        a.order = ps.order + pos0;
        a.p0 = ps.p0 + pos0;
        a.p1 = ps.p1 + pos0;
        a.p2 = ps.p2 + pos0;
        a.s0 = ps.s0 + pos0;
        a.uacc = ps.uacc + pos0;
        a.logu = ps.logu + pos0;
        a.fac = ps.fac + pos0;
        a.corder = c->cplan.order;
        a.cp0 = c->cplan.p0;
        a.cp1 = c->cplan.p1;
        a.cp2 = c->cplan.p2;
        a.cs0 = c->cplan.s0;
        a.cuacc = c->cplan.uacc;
        a.clogu = c->cplan.logu;
        a.cfac = c->cplan.fac;
        a.counts = counts;
        a.X = c->X;
        a.rec = c->sendbuf;
        a.status = c->status;
        a.N = (int32_t)c->N;
        a.D = c->D;
        a.G = c->world;
        a.rank = c->rank;
        a.ns = ns;
        a.npart = npart;
        a.cap = (int32_t)cap;
        hipLaunchKernelGGL(k_pull_plan, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, c->stream, a);
        HIPOK(c, hipGetLastError());
    }
    c->pull_cap_cur = cap;
    c->pull_split = split;
    if (records_per_peer) *records_per_peer = cap;
    return 0;
}

int emx_pull_apply(emx_ctx* c, int32_t split) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, cur.active && c->pull_split == split, "emx_pull_apply: emx_pull_prepare(%d) has not run", split);
    const emx_move_desc& mv = c->moves[cur.move];
    const int ns = cur.off[split + 1] - cur.off[split];
    c->pull_split = -1;
    {
        // received rows into the replica; the send records and the other counter buffer are re-armed for the next half-step
        PullRowsArgs r{};
        r.X = c->X;
        r.rec = c->gathered;
        r.sent = c->sendbuf;
        r.counts_next = c->pull_counts + (size_t)(c->pull_parity ^ 1) * (1 + EMX_MAX_RANKS);
        r.N = (int32_t)c->N;
        r.D = c->D;
        r.G = c->world;
        r.rank = c->rank;
        r.cap = (int32_t)c->pull_cap_cur;
        const int64_t nrec = std::max<int64_t>(1, (int64_t)c->world * c->pull_cap_cur);
        hipLaunchKernelGGL(k_pull_scatter, dim3((unsigned)((nrec + 15) / 16)), dim3(256), 0, c->stream, r);
        HIPOK(c, hipGetLastError());
    }
    const int32_t* counts_now = c->pull_counts + (size_t)c->pull_parity * (1 + EMX_MAX_RANKS);
    c->pull_parity ^= 1;              // the plan kernel re-armed the other buffer for the next half-step
    if (ns <= 0) return 0;
    // the grid is sized for the expected number of owned slots (the kernel's batch loop covers any count;
    // the count itself is read on the device)
    const int64_t G = c->world, bmax = (c->N + G - 1) / G;
    const double mean = (double)ns / G;
    int64_t bound = (int64_t)std::ceil(mean + 8.0 * std::sqrt(mean) + 64.0);
    bound = std::max<int64_t>(1, std::min<int64_t>(bound, std::min<int64_t>(bmax, ns)));
    double *chain = nullptr, *chain_lp = nullptr;
    if (cur.store) {
        chain = c->chain + (size_t)c->stored * c->N * c->D;
        chain_lp = c->chain_lp + (size_t)c->stored * c->N;
    }
    return launch_split(c, mv.kind, c->target, cur.S, split, 0, (int)bound, 0, (int)bound, &mv, &c->cplan, nullptr, c->X,
                        c->lp, chain, chain_lp, nullptr, nullptr, counts_now);
}

// ---- direct exchange ---------------------------------------------------------------------------------------------
static void direct_detach(emx_ctx* c) {
    for (int q = 0; q < EMX_MAX_PEERS; ++q) {
        if (c->peer_ipc_x[q] && c->peerX[q]) hipIpcCloseMemHandle(c->peerX[q]);
        if (c->peer_ipc_f[q] && c->peer_flags[q]) hipIpcCloseMemHandle(c->peer_flags[q]);
        c->peerX[q] = nullptr;
        c->peer_flags[q] = nullptr;
        c->peer_ipc_x[q] = c->peer_ipc_f[q] = false;
    }
    c->peers_ready = false;
}

// what the device-side barrier needs (direct exchange, device-side replay exchange)
static int flags_ensure(emx_ctx* c) {
    if (!c->direct_counts) {
        HIPOK(c, hipMalloc((void**)&c->direct_counts, 65 * 4));       // [64]: "a barrier timed out" (later barriers do not wait again)
        HIPOK(c, hipMemset(c->direct_counts, 0, 65 * 4));
    }
    if (!c->my_flags) {
        // the barrier flags are written by the peers while this device polls them: fine-grained (uncached) device memory
        // where the runtime offers it, ordinary device memory otherwise (the polls are system-scope atomics either way)
        void* f = nullptr;
        if (hipExtMallocWithFlags(&f, EMX_MAX_PEERS * 8, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIPOK(c, hipMalloc(&f, EMX_MAX_PEERS * 8));
        }
        c->my_flags = (unsigned long long*)f;
        HIPOK(c, hipMemset(c->my_flags, 0, EMX_MAX_PEERS * 8));
        c->direct_epoch = 0;
    }
    return 0;
}

static int direct_ensure(emx_ctx* c) {
    const int64_t G = c->world, N = c->N, bmax = (N + G - 1) / G;
    if (c->cplan_rows < N) {          // the compact plans of all splits of a step, split s at offset off[s]
        HIPOK(c, hipStreamSynchronize(c->stream));
        auto& p = c->cplan;
        void* old[] = {p.order, p.p0, p.p1, p.p2, p.s0, p.uacc, p.logu, p.fac};
        for (void* q : old)
            if (q) hipFree(q);
        HIPOK(c, hipMalloc((void**)&p.order, N * 4));
        HIPOK(c, hipMalloc((void**)&p.p0, N * 4));
        HIPOK(c, hipMalloc((void**)&p.p1, N * 4));
        HIPOK(c, hipMalloc((void**)&p.p2, N * 4));
        HIPOK(c, hipMalloc((void**)&p.s0, N * 8));
        HIPOK(c, hipMalloc((void**)&p.uacc, N * 8));
        HIPOK(c, hipMalloc((void**)&p.logu, N * 8));
        HIPOK(c, hipMalloc((void**)&p.fac, N * 8));
        c->cplan_rows = N;
    }
    {
        const int rcf = flags_ensure(c);
        if (rcf) return rcf;
    }
    // buffers of the replica re-synchronisation (one all-gather of the blocks when emx_run returns)
    const int64_t send = bmax * (c->D + 3), recv = G * bmax * (c->D + 3);
    if (c->send_doubles < send || c->recv_doubles < recv) {
        HIPOK(c, hipStreamSynchronize(c->stream));
        exchange_free(c);
        HIPOK(c, hipMalloc((void**)&c->sendbuf, (size_t)send * 8));
        HIPOK(c, hipMalloc((void**)&c->gathered, (size_t)recv * 8));
        c->own_shard_bufs = true;
        c->send_doubles = send;
        c->recv_doubles = recv;
    }
    return 0;
}

// (Re-)attaching the peers is collective: every rank starts over at epoch 0 with clean flags, and a barrier that timed out
// in an earlier attachment is forgotten (until then it stays raised in the status word and refused on the host).  Called
// where no peer can be writing this rank's flags yet: emx_direct_export, and emx_direct_attach (contexts of one process,
// attached one after the other before any of them runs).
static int direct_rearm(emx_ctx* c) {
    HIPOK(c, hipStreamSynchronize(c->stream));
    if (c->direct_counts) HIPOK(c, hipMemset(c->direct_counts, 0, 65 * 4));
    if (c->my_flags) HIPOK(c, hipMemset(c->my_flags, 0, EMX_MAX_PEERS * 8));
    c->direct_epoch = 0;
    c->direct_dead = false;
    __atomic_store_n(&c->status_host[3], 0u, __ATOMIC_RELEASE);
    return 0;
}

static int direct_publish_table(emx_ctx* c) {
    PeerTable t{};
    for (int q = 0; q < c->world; ++q) {
        t.X[q] = c->peerX[q];
        t.lo[q] = (int32_t)(c->N * q / c->world);
    }
    if (!c->peer_table) HIPOK(c, hipMalloc((void**)&c->peer_table, sizeof(PeerTable)));
    HIPOK(c, hipStreamSynchronize(c->stream));
    HIPOK(c, hipMemcpy(c->peer_table, &t, sizeof(t), hipMemcpyHostToDevice));
    return 0;
}

// the array a peer maps: the coordinates (direct exchange: partner rows are read from it) or the receive buffers of the
// decisions (replay exchange: the peers store into it)
static inline bool maps_peers(const emx_ctx* c) { return c->exchange == EMX_EXCHANGE_DIRECT || c->exchange == EMX_EXCHANGE_REPLAY; }
static inline int peers_ensure(emx_ctx* c) { return c->exchange == EMX_EXCHANGE_REPLAY ? replay_ensure(c) : direct_ensure(c); }
static inline double* peer_mapped_array(emx_ctx* c) { return c->exchange == EMX_EXCHANGE_REPLAY ? c->gathered : c->X; }

int emx_direct_export(emx_ctx* c, uint8_t handles[128]) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, maps_peers(c), "emx_direct_export needs emx_set_exchange(EMX_EXCHANGE_DIRECT or EMX_EXCHANGE_REPLAY)");
    int rc = peers_ensure(c);
    if (rc) return rc;
    // every rank exports before any rank can import (the host layer's all-gather of the handles sits in between), so this is
    // the one point where no peer can be writing this rank's flags: start the new attachment from epoch 0
    rc = direct_rearm(c);
    if (rc) return rc;
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle size");
    hipIpcMemHandle_t hx, hf;
    HIPOK(c, hipIpcGetMemHandle(&hx, peer_mapped_array(c)));
    HIPOK(c, hipIpcGetMemHandle(&hf, c->my_flags));
    memset(handles, 0, 128);
    memcpy(handles, &hx, sizeof(hx));
    memcpy(handles + 64, &hf, sizeof(hf));
    return 0;
}

int emx_direct_import(emx_ctx* c, const uint8_t* handles) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, maps_peers(c) && c->world >= 1, "emx_direct_import needs the direct or the replay exchange and emx_set_shard");
    NEED(c, c->world <= EMX_MAX_PEERS, "peer mapping: at most %d ranks (the GPUs of one node)", EMX_MAX_PEERS);
    int rc = peers_ensure(c);
    if (rc) return rc;
    direct_detach(c);
    for (int q = 0; q < c->world; ++q) {
        if (q == c->rank) {
            c->peerX[q] = peer_mapped_array(c);
            c->peer_flags[q] = c->my_flags;
            continue;
        }
        hipIpcMemHandle_t hx, hf;
        memcpy(&hx, handles + (size_t)q * 128, sizeof(hx));
        memcpy(&hf, handles + (size_t)q * 128 + 64, sizeof(hf));
        void *px = nullptr, *pf = nullptr;
        HIPOK(c, hipIpcOpenMemHandle(&px, hx, hipIpcMemLazyEnablePeerAccess));
        c->peerX[q] = (double*)px;
        c->peer_ipc_x[q] = true;
        HIPOK(c, hipIpcOpenMemHandle(&pf, hf, hipIpcMemLazyEnablePeerAccess));
        c->peer_flags[q] = (unsigned long long*)pf;
        c->peer_ipc_f[q] = true;
    }
    c->peers_ready = true;
    return direct_publish_table(c);          // direct: the peers' coordinate arrays; replay: their receive buffers
}

int emx_direct_attach(emx_ctx* c, void* const* peer_coords, void* const* peer_flags) {
    NEED(c, maps_peers(c) && c->world >= 1, "emx_direct_attach needs the direct or the replay exchange and emx_set_shard");
    NEED(c, c->world <= EMX_MAX_PEERS, "peer mapping: at most %d ranks (the GPUs of one node)", EMX_MAX_PEERS);
    int rc = peers_ensure(c);
    if (rc) return rc;
    direct_detach(c);
    for (int q = 0; q < c->world; ++q) {
        c->peerX[q] = q == c->rank ? peer_mapped_array(c) : (double*)peer_coords[q];
        c->peer_flags[q] = q == c->rank ? c->my_flags : (unsigned long long*)(peer_flags ? peer_flags[q] : nullptr);
        NEED(c, c->peerX[q], "emx_direct_attach: no coordinate array for rank %d", q);
    }
    c->peers_ready = true;
    rc = direct_rearm(c);
    if (rc) return rc;
    return direct_publish_table(c);
}

int emx_direct_halfstep(emx_ctx* c, int32_t split, int32_t barrier) {
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, c->exchange == EMX_EXCHANGE_DIRECT, "emx_direct_halfstep needs emx_set_exchange(EMX_EXCHANGE_DIRECT)");
    NEED(c, cur.active && cur.move >= 0 && cur.slot >= 0, "emx_direct_halfstep outside a planned step");
    NEED(c, split >= 0 && split < cur.S && cur.S <= 64, "split out of range");
    NEED(c, c->target != EMX_TARGET_HOST, "sharded stepping needs a device target");
    NEED(c, c->peers_ready || c->world == 1, "direct exchange: peers not mapped");
    NEED(c, !c->direct_dead, "direct exchange: an earlier barrier timed out (a peer never arrived): the ranks are no longer ordered -- "
                             "attach the peers again (emx_direct_import / emx_direct_attach on every rank)");
    {
        const int rcl = complete_lean_plan(c);         // k_own_plan copies every column
        if (rcl) return rcl;
    }
    const emx_move_desc& mv = c->moves[cur.move];
    const auto& ps = c->ring[cur.slot];
    const int64_t G = c->world, N = c->N;
    if (!c->direct_planned) {
        // once per step: the slots of every split whose walker this rank owns
        HIPOK(c, hipMemsetAsync(c->direct_counts, 0, 64 * 4, c->stream));
        OwnPlanArgs a{};
        a.order = ps.order;
        a.p0 = ps.p0;
        a.p1 = ps.p1;
        a.p2 = ps.p2;
        a.s0 = ps.s0;
        a.uacc = ps.uacc;
        a.logu = ps.logu;
        a.fac = ps.fac;
        a.corder = c->cplan.order;
        a.cp0 = c->cplan.p0;
        a.cp1 = c->cplan.p1;
        a.cp2 = c->cplan.p2;
        a.cs0 = c->cplan.s0;
        a.cuacc = c->cplan.uacc;
        a.clogu = c->cplan.logu;
        a.cfac = c->cplan.fac;
        a.counts = c->direct_counts;
        for (int s = 0; s <= cur.S; ++s) a.off[s] = cur.off[s];
        a.N = (int32_t)N;
        a.G = (int32_t)G;
        a.rank = c->rank;
        a.S = cur.S;
        hipLaunchKernelGGL(k_own_plan, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, a);
        HIPOK(c, hipGetLastError());
        c->direct_planned = true;
    }
    if (barrier && G > 1) {
        PeerBarrierArgs b{};
        for (int q = 0; q < G; ++q) {
            NEED(c, c->peer_flags[q], "direct exchange: no barrier flags for rank %d", q);
            b.peer_flags[q] = c->peer_flags[q];
        }
        b.my_flags = c->my_flags;
        b.dead = c->direct_counts + 64;
        b.status = c->status;
        b.epoch = ++c->direct_epoch;
        // the first barrier of an emx_run call waits six times longer: the ranks reach emx_run through host code of their own
        // (Python between two yields, the plan pipeline's start-up) and may be seconds apart; later barriers are kernel to kernel
        b.timeout_ticks = (unsigned long long)c->tune_direct_timeout_ms * (c->direct_first_barrier ? 6ull : 1ull) * 100000ull;      // wall_clock64: 100 MHz
        c->direct_first_barrier = false;
        b.rank = c->rank;
        b.npeer = (int32_t)G;
        hipLaunchKernelGGL(k_peer_barrier, dim3(1), dim3(64), 0, c->stream, b);
        HIPOK(c, hipGetLastError());
    }
    const int ns = cur.off[split + 1] - cur.off[split];
    if (ns <= 0) return 0;
    const int64_t bmax = (N + G - 1) / G;
    const double mean = (double)ns / G;
    int64_t bound = (int64_t)std::ceil(mean + 8.0 * std::sqrt(mean) + 64.0);      // grid size only: the count is read on the device
    bound = std::max<int64_t>(1, std::min<int64_t>(bound, std::min<int64_t>(bmax, ns)));
    double *chain = nullptr, *chain_lp = nullptr;
    if (cur.store) {
        chain = c->chain + (size_t)c->stored * N * c->D;
        chain_lp = c->chain_lp + (size_t)c->stored * N;
    }
    return launch_split(c, mv.kind, c->target, cur.S, split, cur.off[split], (int)bound, 0, (int)bound, &mv, &c->cplan, nullptr, c->X,
                        c->lp, chain, chain_lp, nullptr, nullptr, c->direct_counts + split);
}

static void block_args(emx_ctx* c, BlockArgs& a, double* rec) {
    a.X = c->X;
    a.lp = c->lp;
    a.acc = c->acc;
    a.acc_count = c->acc_count;
    a.rec = rec;
    a.N = (int32_t)c->N;
    a.D = c->D;
    a.G = c->world;
    a.rank = c->rank;
    a.bmax = (int32_t)((c->N + c->world - 1) / c->world);
}

int emx_replica_pack(emx_ctx* c, int64_t* records_per_rank) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, c->exchange != EMX_EXCHANGE_ALLGATHER, "emx_replica_pack is for the block-ownership exchanges (pull, direct)");
    int rc = c->exchange == EMX_EXCHANGE_PULL ? pull_ensure(c) : direct_ensure(c);
    if (rc) return rc;
    BlockArgs a{};
    block_args(c, a, c->sendbuf);
    hipLaunchKernelGGL(k_block_pack, dim3((unsigned)((a.bmax + 15) / 16)), dim3(256), 0, c->stream, a);
    HIPOK(c, hipGetLastError());
    if (records_per_rank) *records_per_rank = a.bmax;
    return 0;
}

int emx_replica_unpack(emx_ctx* c) {
    HIPOK(c, hipSetDevice(c->device));
    NEED(c, c->exchange != EMX_EXCHANGE_ALLGATHER && c->gathered, "emx_replica_unpack is for the block-ownership exchanges (pull, direct)");
    BlockArgs a{};
    block_args(c, a, c->gathered);
    const int64_t nrec = (int64_t)a.G * a.bmax;
    hipLaunchKernelGGL(k_block_unpack, dim3((unsigned)((nrec + 15) / 16)), dim3(256), 0, c->stream, a);
    HIPOK(c, hipGetLastError());
    return 0;
}

int emx_device_ptr(emx_ctx* c, int32_t which, void** ptr, int64_t* nbytes) {
    switch (which) {
        case 0: *ptr = c->X; *nbytes = c->N * c->D * 8; return 0;
        case 1: *ptr = c->lp; *nbytes = c->N * 8; return 0;
        case 2: *ptr = c->sendbuf; *nbytes = c->send_doubles * 8; return 0;
        case 3: *ptr = c->gathered; *nbytes = c->recv_doubles * 8; return 0;
        case 4: *ptr = c->chain; *nbytes = c->stored * c->N * c->D * 8; return 0;
        case 5: *ptr = c->chain_lp; *nbytes = c->stored * c->N * 8; return 0;
        case 6: *ptr = c->disp; *nbytes = c->disp ? c->N * c->D * 8 : 0; return 0;
        case 7: *ptr = c->dbg; *nbytes = c->dbg ? c->dbg_blocks * 16 * 8 : 0; return 0;
        case 8: *ptr = c->my_flags; *nbytes = c->my_flags ? EMX_MAX_PEERS * 8 : 0; return 0;
    }
    FAIL(c, -1, "unknown device pointer id %d", which);
}

int emx_shard_slots(emx_ctx* c, int32_t split, int64_t* lo, int64_t* hi, int64_t* ns) {
    auto& cur = c->cur;
    NEED(c, cur.active && split >= 0 && split < cur.S, "emx_shard_slots outside a planned step");
    *ns = cur.off[split + 1] - cur.off[split];
    shard_range(*ns, c->rank, c->world, *lo, *hi);
    return 0;
}

int emx_scatter_gathered(emx_ctx* c, int32_t split) { return scatter_gathered(c, split, c->sendbuf_rows); }

static int scatter_gathered(emx_ctx* c, int32_t split, int64_t block_rows) {
    // `gathered` holds `world` blocks of `block_rows` records; block r carries rank r's slots.
    HIPOK(c, hipSetDevice(c->device));
    auto& cur = c->cur;
    NEED(c, cur.active && c->gathered, "emx_scatter_gathered needs an active sharded step");
    NEED(c, split >= 0 && split < cur.S, "split out of range");
    const int ns = cur.off[split + 1] - cur.off[split];
    const int64_t rec = c->D + 2;
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        int64_t lo, hi;
        shard_range(ns, r, c->world, lo, hi);
        if (hi <= lo) continue;
        ScatterArgs a{};
        a.X = c->X;
        a.lp = c->lp;
        a.acc = c->acc;
        a.acc_count = c->acc_count;
        if (cur.store) {
            a.chain = c->chain + (size_t)c->stored * c->N * c->D;
            a.chain_lp = c->chain_lp + (size_t)c->stored * c->N;
        }
        // block r starts at record r * sendbuf_rows and holds slot lo first: shift so that slot t indexes directly
        a.gathered = c->gathered + ((int64_t)r * block_rows - lo) * rec;
        a.order = cur.slot >= 0 ? c->ring[cur.slot].order : nullptr;
        a.N = (int32_t)c->N;
        a.D = c->D;
        a.S = cur.S;
        a.split = split;
        a.pos0 = cur.off[split];
        a.t_lo = (int32_t)lo;
        a.t_hi = (int32_t)hi;
        hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((hi - lo + 3) / 4)), dim3(256), 0, c->stream, a);
        HIPOK(c, hipGetLastError());
    }
    return 0;
}

// ---- RCCL driven from the library (one process per GPU) ----------------------------------
int emx_comm_load(const char* librccl_path) {
    std::string err;
    const int rc = rccl_load(librccl_path, err);
    if (rc) g_err = err;
    return rc;
}

int emx_comm_get_unique_id(uint8_t id[128]) {
    std::string err;
    if (rccl_load(nullptr, err)) {
        g_err = err;
        return -5;
    }
    RcclId u;
    const int e = g_rccl.GetUniqueId(&u);
    if (e != 0) {
        g_err = "ncclGetUniqueId failed";
        return -6;
    }
    memcpy(id, u.internal, 128);
    return 0;
}

int emx_comm_init(emx_ctx* c, int32_t rank, int32_t world, const uint8_t id[128]) {
    HIPOK(c, hipSetDevice(c->device));
    std::string err;
    if (rccl_load(nullptr, err)) FAIL(c, -5, "%s", err.c_str());
    NEED(c, !c->comm, "communicator already initialised");
    int rc = emx_set_shard(c, rank, world);
    if (rc) return rc;
    if (c->exchange == EMX_EXCHANGE_PULL) {
        rc = pull_ensure(c);
        if (rc) return rc;
    } else if (c->exchange == EMX_EXCHANGE_DIRECT) {
        rc = direct_ensure(c);
        if (rc) return rc;
    } else if (c->exchange == EMX_EXCHANGE_LOGPROB || c->exchange == EMX_EXCHANGE_REPLAY) {
        // emx_set_shard allocated the gather buffer(s)
    } else if (!c->sendbuf) {   // world == 1: still exercise the exchange buffers
        const int64_t per = c->N + 2;
        HIPOK(c, hipMalloc((void**)&c->sendbuf, (size_t)per * (c->D + 2) * 8));
        HIPOK(c, hipMalloc((void**)&c->gathered, (size_t)per * (c->D + 2) * 8));
        c->own_shard_bufs = true;
        c->sendbuf_rows = per;
        c->gathered_rows = per;
        c->send_doubles = c->recv_doubles = per * (c->D + 2);
    }
    RcclId u;
    memcpy(u.internal, id, 128);
    const int e = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (e != 0) {
        c->comm = nullptr;
        FAIL(c, -6, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    }
    return 0;
}

int emx_comm_destroy(emx_ctx* c) {
    if (c->comm) {
        hipStreamSynchronize(c->stream);
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    return 0;
}

// exact mode: where the plan pipeline's time goes (all zeros when no pipeline is alive)
int emx_pipeline_stats(emx_ctx* c, double out[6], int64_t* steps_produced, int32_t* finisher_threads) {
    for (int k = 0; k < 6; ++k) out[k] = 0.0;
    if (steps_produced) *steps_produced = 0;
    if (finisher_threads) *finisher_threads = 0;
    if (!c->pipe) return 0;
    c->pipe->stage_times(out, steps_produced);
    if (finisher_threads) *finisher_threads = c->pipe->workers();
    return 0;
}

int emx_pipeline_handovers(emx_ctx* c, int64_t* raw_steps, int64_t* regen_steps) {
    if (raw_steps) *raw_steps = c->pipe_raw_steps;
    if (regen_steps) *regen_steps = c->pipe_regen_steps;
    return 0;
}

int emx_comm_count(emx_ctx* c, int32_t* ranks_out) {
    *ranks_out = 0;
    if (!c->comm) return 0;
    NEED(c, g_rccl.CommCount, "librccl lacks ncclCommCount");
    int n = 0;
    const int e = g_rccl.CommCount(c->comm, &n);
    if (e != 0) FAIL(c, -6, "ncclCommCount failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    *ranks_out = n;
    return 0;
}

// ---- measurement -------------------------------------------------------------------------
int emx_timer_start(emx_ctx* c) {
    HIPOK(c, hipEventRecord(c->ev0, c->stream));
    return 0;
}

int emx_timer_stop(emx_ctx* c, float* ms) {
    HIPOK(c, hipEventRecord(c->ev1, c->stream));
    HIPOK(c, wait_event(c->ev1));
    HIPOK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

int emx_profile_enable(emx_ctx* c, int32_t max_launches) {
    HIPOK(c, hipSetDevice(c->device));
    for (auto e : c->prof) hipEventDestroy(e);
    c->prof.clear();
    c->prof_max = max_launches;
    c->prof_n = 0;
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipEvent_t e;
        HIPOK(c, hipEventCreate(&e));
        c->prof.push_back(e);
    }
    return 0;
}

int emx_profile_read(emx_ctx* c, float* ms_out, int32_t* n) {
    HIPOK(c, hipStreamSynchronize(c->stream));
    const int k = std::min<int>(*n, c->prof_n);
    for (int i = 0; i < k; ++i) HIPOK(c, hipEventElapsedTime(&ms_out[i], c->prof[2 * i], c->prof[2 * i + 1]));
    *n = k;
    c->prof_n = 0;
    return 0;
}

// ---- host-only helpers -------------------------------------------------------------------
emx_mt* emx_mt_create(const uint32_t key[624], int32_t pos, int32_t hg, double cached) {
    emx_mt* m = new emx_mt();
    m->mt.set_state(key, pos, hg, cached);
    return m;
}
void emx_mt_destroy(emx_mt* m) { delete m; }
void emx_mt_get_state(const emx_mt* m, uint32_t key[624], int32_t* pos, int32_t* hg, double* cached) {
    memcpy(key, m->mt.key, sizeof(m->mt.key));
    *pos = m->mt.pos;
    *hg = m->mt.has_gauss;
    *cached = m->mt.gauss;
}
void emx_mt_random_sample(emx_mt* m, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = m->mt.next_double();
}
void emx_mt_randint(emx_mt* m, uint64_t bound, int64_t n, int64_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (int64_t)m->mt.randint(bound);
}
void emx_mt_randn(emx_mt* m, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = m->mt.next_gauss();
}
void emx_mt_shuffle_labels(emx_mt* m, int64_t n, int32_t S, int32_t* labels) {
    for (int64_t i = 0; i < n; ++i) labels[i] = (int32_t)(i % S);
    m->mt.shuffle(labels, n);
}
int32_t emx_mt_choice_cdf(emx_mt* m, const double* cdf, int32_t n) { return m->mt.choice_cdf(cdf, n); }

int emx_host_plan_mt(emx_mt* m, int64_t N, int32_t D, const emx_move_desc* mv, int32_t* off, int32_t* order, int32_t* p0,
                     int32_t* p1, int32_t* p2, double* s0, double* uacc) {
    std::vector<uint8_t> labels;
    return make_exact_plan(m->mt, N, D, *mv, labels, off, order, p0, p1, p2, s0, uacc);
}

// Host twins of k_plan_regen and k_plan_raw, for emx_host_plan_mt_stream's test mode (EMX_TEST_PIPE_DEVFIN): what the consumer's
// kernels do with a raw / regen step, in plain C++ -- so that the tokenizer's hand-over (which states, which offset) is checked against
// the serial twin without a GPU (tests/test_mt_pipeline_cpu.py).
static inline uint32_t host_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
static inline uint32_t host_mix(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu);
}
static void host_regen_twin(const PipeStepInfo& info, const PlanSink& sk, int64_t N, std::vector<uint32_t>& wr) {
    const int64_t ns0 = info.off[1] - info.off[0], F0 = 5 * ns0, F = 5 * N;
    uint32_t* wz = reinterpret_cast<uint32_t*>(sk.s0);
    uint32_t* wu = reinterpret_cast<uint32_t*>(sk.uacc);
    wr.assign((size_t)N, 0u);
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(sk.p0);
    uint32_t a[624], b[624];
    for (int seg = 0; seg < info.regen_nseg; ++seg) {
        memcpy(a, keys + (size_t)seg * 624, sizeof(a));
        for (int blk = 0; blk < PIPE_REGEN_SB; ++blk) {
            const int64_t r0 = ((int64_t)seg * PIPE_REGEN_SB + blk) * 624 - info.regen_off;
            if (r0 >= F) break;
            for (int i = 0; i < 624; ++i) {
                const int64_t r = r0 + i;
                if (r < 0 || r >= F) continue;
                const bool split = r >= F0;
                const int64_t q = split ? r - F0 : r, base = split ? ns0 : 0, ns = split ? N - ns0 : ns0;
                if (q < 2 * ns)
                    wz[2 * base + q] = a[i];
                else if (q < 3 * ns)
                    wr[(size_t)(base + q - 2 * ns)] = a[i];
                else
                    wu[2 * base + (q - 3 * ns)] = a[i];
            }
            for (int kk = 0; kk < 227; ++kk) b[kk] = a[kk + 397] ^ host_mix(a[kk], a[kk + 1]);
            for (int kk = 227; kk < 623; ++kk) b[kk] = b[kk - 227] ^ host_mix(a[kk], a[kk + 1]);
            b[623] = b[396] ^ host_mix(a[623], b[0]);
            memcpy(a, b, sizeof(a));
        }
    }
}
static void host_raw_twin(const PipeStepInfo& info, const PlanSink& sk, int64_t N, double a, const std::vector<uint32_t>* wr_from) {
    for (int s = 0; s < info.S; ++s) {
        const int64_t base = info.off[s], ns = info.off[s + 1] - info.off[s];
        for (int64_t t = base; t < base + ns; ++t) {
            const uint32_t* wz = reinterpret_cast<const uint32_t*>(sk.s0 + t);
            const uint32_t* wu = reinterpret_cast<const uint32_t*>(sk.uacc + t);
            auto dbl = [](uint32_t w0, uint32_t w1) {
                const int32_t hi = (int32_t)(host_temper(w0) >> 5), lo = (int32_t)(host_temper(w1) >> 6);
                return ((double)hi * 67108864.0 + (double)lo) / 9007199254740992.0;
            };
            const double u = dbl(wz[0], wz[1]), ua = dbl(wu[0], wu[1]);
            const double tt = (a - 1.0) * u + 1.0;
            const uint32_t w = wr_from ? (*wr_from)[(size_t)t] : (uint32_t)sk.p0[t];
            const int32_t r = info.wr_ring ? (int32_t)(host_temper(w) & (uint32_t)(N - ns - 1)) : (int32_t)w;
            sk.p0[t] = r < base ? sk.order[r] : sk.order[r + ns];
            sk.s0[t] = tt * tt / a;
            sk.uacc[t] = ua;
        }
    }
}

int emx_host_plan_mt_stream(emx_mt* m, int64_t N, int32_t D, int32_t nmoves, const emx_move_desc* moves, const double* cdf,
                            int64_t nsteps, int32_t nworkers, int32_t nsinks, int32_t* moves_out, int32_t* order, int32_t* p0,
                            int32_t* p1, int32_t* p2, double* s0, double* uacc, double* seconds_out) {
    if (!m || N < 2 || nsteps < 1 || nsinks < 1 || !MtPlanPipeline::supports(nmoves, moves)) return -1;
    for (int i = 0; i < nmoves; ++i)
        if (moves[i].kind == EMX_MOVE_DE && N - (N + moves[i].nsplits - 1) / moves[i].nsplits < 2) return -1;
    // staging buffers laid out like a plan slot's pinned block; the consumer below plays emx_run's part
    std::vector<std::vector<char>> stage((size_t)nsinks, std::vector<char>((size_t)N * 32));
    std::vector<PlanSink> sinks((size_t)nsinks);
    for (int r = 0; r < nsinks; ++r) {
        const HostPlan hp(stage[r].data(), (size_t)N);
        sinks[r].order = hp.order;
        sinks[r].p0 = hp.p0;
        sinks[r].p1 = hp.p1;
        sinks[r].p2 = hp.p2;
        sinks[r].s0 = hp.s0;
        sinks[r].uacc = hp.uacc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    // fill_unused_fields: a stretch plan's p1 / p2 carry the walker itself, as emx_host_plan_mt's do (the header promises the
    // same plans; emx_run's own pipeline leaves those columns alone because nothing reads or uploads them)
    // test mode, EMX_TEST_PIPE_DEVFIN = <regen_min_walkers> (0: raw steps only): the pipeline in its device-finish configuration, the
    // consumer's kernels played by their host twins above
    const char* tdf = getenv("EMX_TEST_PIPE_DEVFIN");
    MtPlanPipeline pipe(m->mt, N, D, nmoves, moves, cdf, nsteps, sinks.data(), nsinks, nworkers, true, tdf != nullptr, false, tdf ? atoll(tdf) : 0);
    int64_t n = 0;
    std::vector<uint32_t> wr;
    for (; n < nsteps; ++n) {
        PipeStepInfo info;
        if (!pipe.wait_ready(n, info)) break;
        if (moves_out) moves_out[n] = info.move;
        const PlanSink& sk = sinks[n % nsinks];
        if (info.raw) {
            if (info.regen) host_regen_twin(info, sk, N, wr);
            host_raw_twin(info, sk, N, moves[info.move].a, info.regen ? &wr : nullptr);
            if (moves_out) moves_out[n] |= 256 | (info.regen ? 512 : 0);          // (test mode only: how the step was handed over)
            for (int64_t t = 0; t < N; ++t) sk.p1[t] = sk.p2[t] = sk.order[t];        // (fill_unused_fields: raw steps leave them to the consumer)
        }
        const size_t o = (size_t)n * (size_t)N;
        if (order) memcpy(order + o, sk.order, (size_t)N * 4);
        if (p0) memcpy(p0 + o, sk.p0, (size_t)N * 4);
        if (p1) memcpy(p1 + o, sk.p1, (size_t)N * 4);
        if (p2) memcpy(p2 + o, sk.p2, (size_t)N * 4);
        if (s0) memcpy(s0 + o, sk.s0, (size_t)N * 8);
        if (uacc) memcpy(uacc + o, sk.uacc, (size_t)N * 8);
        pipe.release(n);
    }
    pipe.finish(n, m->mt);
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return n == nsteps ? pipe.workers() : -7;
}

int emx_host_split_draws(emx_mt* m, int64_t N, const emx_move_desc* mv, const int32_t* off, const int32_t* order,
                         int32_t split, int32_t* p0, int32_t* p1, int32_t* p2, double* s0) {
    if (mv->nsplits < 2 || split < 0 || split >= mv->nsplits) return -1;
    if (mv->kind == EMX_MOVE_SNOOKER && mv->nsplits < 4) return -1;
    return draw_split_proposal(m->mt, N, *mv, off, order, split, p0, p1, p2, s0);
}

int emx_host_plan_philox(uint64_t seed, uint64_t step, int64_t N, const emx_move_desc* mv, int32_t* off, int32_t* order,
                         int32_t* p0, int32_t* p1, int32_t* p2, double* s0, double* uacc) {
    NativeArgs na{};
    na.seed = seed;
    na.step = step;
    na.pk = make_perm_key((uint64_t)N, seed, step);
    switch (mv->kind) {
        case EMX_MOVE_STRETCH: host_native_plan<MOVE_STRETCH>(na, N, *mv, off, order, p0, p1, p2, s0, uacc); return 0;
        case EMX_MOVE_DE: host_native_plan<MOVE_DE>(na, N, *mv, off, order, p0, p1, p2, s0, uacc); return 0;
        case EMX_MOVE_SNOOKER: host_native_plan<MOVE_SNOOKER>(na, N, *mv, off, order, p0, p1, p2, s0, uacc); return 0;
    }
    return -1;
}

int32_t emx_host_move_choice_philox(uint64_t seed, uint64_t step, const double* cdf, int32_t n) {
    return philox_move_choice(seed, step, cdf, n);
}

int64_t emx_host_pull_capacity(int64_t nwalkers, int32_t world, int32_t nsplits, int32_t partners_per_walker) {
    return pull_capacity(nwalkers, world, nsplits, partners_per_walker);
}

}  // extern "C"
#pragma GCC visibility pop
