// Device-side exact-plan producer: the host driver of the kernels in emx_mtdev_kernels.hpp (design: emx_mtdev.hpp).
#include "emx_mtdev.hpp"

#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>

#include "emx_mtdev_kernels.hpp"        // (-DEMX_TOK_PROFILE: where the first wave of the tokenizer spends a round, on stderr)       // (-DEMX_TOK_PROFILE: walk / scan ticks of the tokenizer's first wave on stderr)
#include "emx_mtjump.hpp"

namespace emx {
using namespace mtdev;

namespace {
inline uint32_t untemper32(uint32_t y) {
    y ^= y >> 18;
    y ^= (y << 15) & 0xefc60000u;
    uint32_t t = y;
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    y = t;
    y ^= y >> 11;
    y ^= y >> 22;
    return y;
}
}  // namespace

enum { G_GEN = 0, G_TOK = 8, G_FIN = 16, G_REL = 24, G_STATS = 32, G_WORDS = 64 };          // (a 64-byte line each)
constexpr unsigned long long GATE_TIMEOUT_TICKS_DEFAULT = 100000000ull * 20ull;       // 20 s (EMX_MTDEV_GATE_TIMEOUT_MS shortens it)
static unsigned long long gate_timeout_ticks() {
    static const unsigned long long t = [] {
        const char* e = getenv("EMX_MTDEV_GATE_TIMEOUT_MS");
        const long long ms = e ? atoll(e) : 0;
        return ms > 0 ? (unsigned long long)ms * 100000ull : GATE_TIMEOUT_TICKS_DEFAULT;
    }();
    return t;
}
// Producers alive in this process.  A gate wait spins on a hardware queue; the runtime multiplexes a process's streams onto a few
// queues, and with more than one producer (or context) alive a wait can sit in front of the very kernel that would signal it, until
// it times out.  One producer per process -- the deployment, and all that was measured -- orders its stages through gates; a second
// one takes the event-ordered form (slower: an event wait resolves to the other stream's latest work, but it cannot deadlock).
static std::atomic<int> g_live_producers{0};

struct MtDevProducer::Impl {
    int device = 0;
    int64_t N = 0;
    int32_t D = 0, S = 2;
    emx_move_desc mv{};
    MT19937Legacy start;
    std::vector<MtDevPlanCols> slots;
    uint32_t* status = nullptr;
    // streams: generation | tokenizer | finisher
    hipStream_t s_gen = nullptr, s_tok = nullptr, s_fin = nullptr;
    // the stream ring
    uint32_t* stream = nullptr;
    uint64_t capw = 0;                 // words, a power of two
    uint32_t *base_key = nullptr, *xwin = nullptr, *partial = nullptr, *polys = nullptr;
    uint64_t gen_words = 0;            // words [0, gen_words) are generated (enqueued on s_gen)
    int64_t rounds = 0;
    hipEvent_t ev_gen = nullptr;       // after the latest round
    // what each stage has completed, counted in device words (k_gate_signal / k_gate_wait: emx_mtdev_kernels.hpp)
    unsigned long long* gates = nullptr;       // [G_GEN] rounds generated | [G_TOK] batches tokenised | [G_FIN] batches finished | [G_REL] batches the consumer released
    bool use_gates = true;
    // tokenizer state
    unsigned long long* d_pos = nullptr;
    unsigned* d_err = nullptr;
    bool counted = false;             // this producer is in g_live_producers
    unsigned long long* d_nwin = nullptr;
    uint32_t *J[2] = {nullptr, nullptr}, *rint[2] = {nullptr, nullptr};        // per batch parity
    WalkRec* recs[2] = {nullptr, nullptr};     // [BATCH][maxrec] walk records of the tokenizer (per batch parity)
    uint32_t* nrec[2] = {nullptr, nullptr};    // [BATCH]
    int32_t maxrec = 0, nchunk = 0;
    int32_t wshift = 11, tail = 2048;          // the tokenizer's window rule (emx_mtdev_kernels.hpp)
    uint32_t *fin_partial = nullptr, *fin_hist = nullptr;
    unsigned long long *tokpos[2] = {nullptr, nullptr}, *step_end[MTDEV_NBUF] = {};
    unsigned long long* h_end = nullptr;       // pinned [NBUF][BATCH]: step end positions of the batch in that buffer
    unsigned long long* h_done = nullptr;      // pinned: number of batches whose positions are there
    uint32_t* scratch = nullptr;
    uint32_t* blk_words = nullptr;             // [NBUF][BATCH][624]: the (tempered) block every step of the batch ends in: finish() needs no ring
    // per batch bookkeeping
    struct Batch {
        hipEvent_t tok = nullptr, fin = nullptr, pos = nullptr, released = nullptr;
        bool has_released = false;
    } bt[MTDEV_NBUF];
    int64_t enq = 0;                   // batches enqueued so far
    int64_t known = -1;                // last batch whose end position the host has read
    uint64_t known_pos = 0;            // ... that position (batch -1: the start position)
    uint64_t pos0 = 0;                 // the start position
    static constexpr int END_HIST = 16;
    uint64_t end_hist[END_HIST] = {};  // end positions of the last batches the host has read (h_end's slots are rewritten after NBUF batches)
    uint64_t wmax = 0, wmin = 0;       // words per step: bounds
    uint64_t slack = 0;                // the tokenizer asks for whole windows: words it may look at beyond what it consumes
};

bool MtDevProducer::supports(int64_t N, int32_t nmoves, const emx_move_desc* moves) {
    if (nmoves != 1 || moves[0].kind != EMX_MOVE_STRETCH) return false;
    if (moves[0].nsplits < 2 || moves[0].nsplits > FIN_MAX_S) return false;
    return N >= 2 * moves[0].nsplits && N <= ((int64_t)1 << 24);
}

#define MTD_HIP(expr)                                                                                                 \
    do {                                                                                                              \
        hipError_t e_ = (expr);                                                                                       \
        if (e_ != hipSuccess) {                                                                                       \
            char b_[384];                                                                                             \
            snprintf(b_, sizeof(b_), "mtdev: %s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            err_ = b_;                                                                                                \
            return -2;                                                                                                \
        }                                                                                                             \
    } while (0)

MtDevProducer::MtDevProducer(int device, const MT19937Legacy& start, int64_t N, int32_t D, const emx_move_desc& mv,
                             const MtDevPlanCols* slots, uint32_t* status_dev) {
    im_ = new Impl();
    Impl& m = *im_;
    m.device = device;
    m.N = N;
    m.D = D;
    m.S = mv.nsplits;
    m.mv = mv;
    m.start = start;
    m.slots.assign(slots, slots + MTDEV_NBUF * MTDEV_BATCH);
    m.status = status_dev;
    auto init = [&]() -> int {
        MTD_HIP(hipSetDevice(device));
        // words per step: ensemble.py:406 (2) + shuffle + per split 2 ns + randint + 2 ns
        uint64_t fixed = 2, rmax = 0, rmin = 0, nrecmax = 0;
        for (int s = 0; s < m.S; ++s) {
            const uint64_t ns = (uint64_t)((N - s + m.S - 1) / m.S), nc = (uint64_t)N - ns;
            fixed += 4 * ns;
            if (nc <= 1) continue;
            if ((nc & (nc - 1)) == 0) {
                rmax += ns;
                rmin += ns;
            } else {
                rmax += 2 * ns + 4096;          // accept probability > 1/2: twice the mean is > 40 sigma out at these sizes
                rmin += ns;
                nrecmax += (2 * ns + 4096) / TOK_W_MAX + 2;
            }
        }
        const uint64_t sh_max = mv.randomize_split ? 2 * (uint64_t)N + 4096 : 0, sh_min = mv.randomize_split ? (uint64_t)N - 1 : 0;
        m.wmax = fixed + rmax + sh_max;
        m.wmin = fixed + rmin + sh_min;
        nrecmax += sh_max / TOK_W_MAX + 256;          // (the narrow windows of the low mask bands: a handful per band)
        m.maxrec = (int32_t)nrecmax;
        m.nchunk = (int32_t)((N + FIN_CHUNK - 1) / FIN_CHUNK);
        m.slack = 2 * (uint64_t)TOK_W_MAX + MT_N;        // the tokenizer decides whole windows (and looks one window further ahead)
        // ring: the batches in flight (the host runs at most NBUF + 1 batches ahead of the tokenizer's known position), a round
        // ahead, and the slack -- rounded up to a power of two
        uint64_t need = (uint64_t)(MTDEV_NBUF + 2) * MTDEV_BATCH * m.wmax + 6 * SEG_WORDS + 2 * m.slack;
        uint64_t cap = (uint64_t)1 << 22;
        while (cap < need) cap <<= 1;
        m.capw = cap;
        // Streams share the runtime's few hardware queues, and kernels of two streams on one queue run in submission order: with
        // default priorities the tokenizer and the finisher landed on ONE queue and took turns (profiles/r04/mtdev_timeline.txt).
        // Queues are pooled per priority: the tokenizer -- the serial stage -- gets the high-priority pool to itself, the
        // generator and the finisher the low one; the consumer's streams keep the default pool.
        int prio_lo = 0, prio_hi = 0;
        MTD_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        MTD_HIP(hipStreamCreateWithPriority(&m.s_gen, hipStreamNonBlocking, prio_lo));
        MTD_HIP(hipStreamCreateWithPriority(&m.s_tok, hipStreamNonBlocking, prio_hi));
        MTD_HIP(hipStreamCreateWithPriority(&m.s_fin, hipStreamNonBlocking, prio_lo));
        MTD_HIP(hipMalloc((void**)&m.stream, cap * 4));
        MTD_HIP(hipMalloc((void**)&m.base_key, MT_N * 4));
        MTD_HIP(hipMalloc((void**)&m.xwin, (size_t)WIN_BLOCKS * MT_N * 4));
        MTD_HIP(hipMalloc((void**)&m.partial, (size_t)(PMAX - 1) * JUMP_SPLIT * MT_N * 4));
        MTD_HIP(hipMalloc((void**)&m.polys, (size_t)(PMAX - 1) * MT_N * 4));
        MTD_HIP(hipMalloc((void**)&m.gates, G_WORDS * 8));
        MTD_HIP(hipMemset(m.gates, 0, G_WORDS * 8));
        m.use_gates = getenv("EMX_MTDEV_EVENTS") == nullptr && g_live_producers.load() == 0;       // (the event-ordered form: for comparison, and for every producer but the first)
        g_live_producers.fetch_add(1);
        m.counted = true;
        MTD_HIP(hipMalloc((void**)&m.d_pos, 8));
        MTD_HIP(hipMalloc((void**)&m.d_err, 4));
        MTD_HIP(hipMalloc((void**)&m.d_nwin, 128));
        for (int k = 0; k < 2; ++k) {
            MTD_HIP(hipMalloc((void**)&m.J[k], (size_t)MTDEV_BATCH * N * 4));
            MTD_HIP(hipMalloc((void**)&m.rint[k], (size_t)MTDEV_BATCH * N * 4));
            MTD_HIP(hipMalloc((void**)&m.tokpos[k], (size_t)MTDEV_BATCH * m.S * 3 * 8));
            MTD_HIP(hipMalloc((void**)&m.recs[k], (size_t)MTDEV_BATCH * m.maxrec * sizeof(WalkRec)));
            MTD_HIP(hipMalloc((void**)&m.nrec[k], (size_t)MTDEV_BATCH * 4));
            MTD_HIP(hipMemset(m.nrec[k], 0, (size_t)MTDEV_BATCH * 4));
        }
        MTD_HIP(hipMalloc((void**)&m.fin_partial, (size_t)MTDEV_BATCH * m.nchunk * 4));
        MTD_HIP(hipMalloc((void**)&m.fin_hist, (size_t)MTDEV_BATCH * m.nchunk * m.S * 4));
        const unsigned evf = hipEventDisableTiming;
        for (int k = 0; k < MTDEV_NBUF; ++k) {
            MTD_HIP(hipMalloc((void**)&m.step_end[k], (size_t)MTDEV_BATCH * 8));
            MTD_HIP(hipEventCreateWithFlags(&m.bt[k].tok, evf));
            MTD_HIP(hipEventCreateWithFlags(&m.bt[k].fin, evf));
            MTD_HIP(hipEventCreateWithFlags(&m.bt[k].pos, evf));
            MTD_HIP(hipEventCreateWithFlags(&m.bt[k].released, evf));
        }
        MTD_HIP(hipEventCreateWithFlags(&m.ev_gen, evf));
        MTD_HIP(hipHostMalloc((void**)&m.h_end, (size_t)(MTDEV_NBUF * MTDEV_BATCH + 8) * 8, hipHostMallocDefault));
        m.h_done = m.h_end + (size_t)MTDEV_NBUF * MTDEV_BATCH;       // batches whose end positions have arrived (k_pos_publish)
        *m.h_done = 0ull;
        MTD_HIP(hipMalloc((void**)&m.scratch, (size_t)MTDEV_BATCH * 5 * N * 4));
        MTD_HIP(hipMalloc((void**)&m.blk_words, (size_t)MTDEV_NBUF * MTDEV_BATCH * MT_N * 4));
        // the jump polynomials t^(k SEG_WORDS) mod phi, k = 1 .. PMAX - 1 (once per process; ~70 ms)
        const auto t0 = std::chrono::steady_clock::now();
        const uint32_t* g = nullptr;
        if (!mt_jump_polys(SEG_WORDS, PMAX - 1, &g)) {
            err_ = "mtdev: the MT19937 jump polynomials could not be computed";
            return -1;
        }
        st_.poly_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        MTD_HIP(hipMemcpy(m.polys, g, (size_t)(PMAX - 1) * MT_N * 4, hipMemcpyHostToDevice));
        MTD_HIP(hipMemcpy(m.base_key, start.key, MT_N * 4, hipMemcpyHostToDevice));
        const unsigned long long p0 = (unsigned long long)start.pos;
        MTD_HIP(hipMemcpy(m.d_pos, &p0, 8, hipMemcpyHostToDevice));
        MTD_HIP(hipMemset(m.d_err, 0, 4));
        MTD_HIP(hipMemset(m.d_nwin, 0, 128));
        m.known = -1;
        m.known_pos = p0;
        m.pos0 = p0;
        m.gen_words = 0;
        return 0;
    };
    if (init() != 0 && err_.empty()) err_ = "mtdev: initialisation failed";
}

MtDevProducer::~MtDevProducer() {
    if (!im_) return;
    Impl& m = *im_;
    if (m.counted) g_live_producers.fetch_sub(1);
    hipSetDevice(m.device);
    for (hipStream_t s : {m.s_gen, m.s_tok, m.s_fin})
        if (s) {
            hipStreamSynchronize(s);
            hipStreamDestroy(s);
        }
    void* bufs[] = {m.gates, m.stream, m.base_key, m.xwin, m.partial, m.polys, m.d_pos, m.d_err, m.d_nwin, m.J[0], m.J[1], m.rint[0], m.rint[1],
                    m.tokpos[0], m.tokpos[1], m.scratch, m.blk_words, m.recs[0], m.recs[1], m.nrec[0], m.nrec[1], m.fin_partial, m.fin_hist};
    for (void* p : bufs)
        if (p) hipFree(p);
    for (int k = 0; k < MTDEV_NBUF; ++k) {
        if (m.step_end[k]) hipFree(m.step_end[k]);
        for (hipEvent_t e : {m.bt[k].tok, m.bt[k].fin, m.bt[k].pos, m.bt[k].released})
            if (e) hipEventDestroy(e);
    }
    if (m.ev_gen) hipEventDestroy(m.ev_gen);
    if (m.h_end) hipHostFree(m.h_end);
    delete im_;
}

void MtDevProducer::refresh_stats() {
    Impl& m = *im_;
    if (hipSetDevice(m.device) != hipSuccess || hipStreamSynchronize(m.s_tok) != hipSuccess) return;
    if (getenv("EMX_MTDEV_TRACE")) {
        unsigned long long gs[12] = {};
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(gs, m.gates + G_STATS, sizeof(gs), hipMemcpyDeviceToHost) == hipSuccess) {
            static const char* names[6] = {"generator<-finisher", "tokenizer<-generator", "tokenizer<-finisher", "finisher<-tokenizer", "finisher<-consumer", "consumer<-finisher"};
            for (int k = 0; k < 6; ++k)
                fprintf(stderr, "gate %-22s %8llu waits, %10.1f us in all, %8.1f us each\n", names[k], gs[2 * k + 1], gs[2 * k] * 0.01, gs[2 * k + 1] ? gs[2 * k] * 0.01 / gs[2 * k + 1] : 0.0);
        }
    }
    unsigned long long nw[16] = {};
    if (hipMemcpy(nw, m.d_nwin, 128, hipMemcpyDeviceToHost) == hipSuccess) {
#ifdef EMX_TOK_PROFILE
        fprintf(stderr, "mtdev tok profile (10 ns ticks, wave 0): guess/loop %llu  walk %llu  scan+barrier %llu  check+barrier %llu  record %llu\n", nw[8], nw[9], nw[10], nw[11], nw[12]);
#endif
        for (int k = 0; k < 4; ++k) st_.tok_ticks[k] = (int64_t)nw[4 + k];
        st_.windows = (int64_t)nw[0];
        st_.tok_rounds = (int64_t)nw[1];
        st_.tail_groups = (int64_t)nw[2];
        st_.tail_rounds = (int64_t)nw[3];
    }
}

void MtDevProducer::set_window_rule(int wshift, int tail) {
    if (wshift >= 8 && wshift <= 20) im_->wshift = wshift;
    if (tail >= 0) im_->tail = tail;
}

static double tr_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int MtDevProducer::ensure_batch(int64_t b, hipStream_t consumer, int lookahead) {
    Impl& m = *im_;
    static const bool TR = getenv("EMX_MTDEV_TRACE") != nullptr && atoi(getenv("EMX_MTDEV_TRACE")) > 1;
    const double tr0 = tr_ms();
    if (TR) fprintf(stderr, "ensure_batch(%lld) at %.3f ms: enq %lld known %lld\n", (long long)b, tr0, (long long)m.enq, (long long)m.known);
    if (!err_.empty()) return -1;
    MTD_HIP(hipSetDevice(m.device));
    lookahead = std::max(0, std::min(lookahead, MTDEV_NBUF - 2));
    // the end positions of batch q have arrived in h_end (k_pos_publish on the tokenizer's stream)
    auto wait_positions = [&](int64_t q) -> bool {
        const double t0_ = tr_ms();
        while ((int64_t)__atomic_load_n(m.h_done, __ATOMIC_ACQUIRE) <= q) {
            if (tr_ms() - t0_ > 20000.0) {
                err_ = "mtdev: the tokenizer did not finish a batch in 20 s";
                return false;
            }
        }
        return true;
    };
    // (site: 0 generator <- finisher, 1 tokenizer <- generator, 2 tokenizer <- finisher, 3 finisher <- tokenizer, 4 finisher <- consumer, 5 consumer <- finisher)
    auto gate_wait = [&](hipStream_t st, int gate, unsigned long long value, int site) {
        hipLaunchKernelGGL(k_gate_wait, dim3(1), dim3(64), 0, st, (const unsigned long long*)(m.gates + gate), value, gate_timeout_ticks(), m.d_err,
                           m.gates + G_STATS + 2 * site, m.status);
    };
    auto gate_signal = [&](hipStream_t st, int gate, unsigned long long value) {
        hipLaunchKernelGGL(k_gate_signal, dim3(1), dim3(64), 0, st, m.gates + gate, value);
    };
    while (m.enq <= b + lookahead) {
        const int64_t nbq = m.enq;
        const int buf = (int)(nbq % MTDEV_NBUF), par = (int)(nbq & 1);
        // a batch further ahead than the consumer has released slots for is not started (the asked-for batch always can be:
        // the caller released batch b - NBUF before asking for b)
        if (nbq >= MTDEV_NBUF && !m.bt[buf].has_released) {
            if (nbq <= b) {
                err_ = "mtdev: batch asked for before the batch four earlier was released";
                return -1;
            }
            break;
        }
        // exact end positions of finished batches tighten the bounds; the host stays at most NBUF batches ahead of what it knows
        // (the pinned slot of batch nbq - NBUF is about to be rewritten, and the bounds below must stay inside the ring)
        while (m.known + 1 < nbq) {
            if (m.known + 1 <= nbq - MTDEV_NBUF) {
                const double t_ = tr_ms();
                if (!wait_positions(m.known + 1)) return -9;
                if (TR) fprintf(stderr, "   wait A for pos(%lld): %.3f ms\n", (long long)m.known + 1, tr_ms() - t_);
            } else if ((int64_t)__atomic_load_n(m.h_done, __ATOMIC_ACQUIRE) <= m.known + 1) {
                break;
            }
            ++m.known;
            m.known_pos = m.h_end[(size_t)(m.known % MTDEV_NBUF) * MTDEV_BATCH + MTDEV_BATCH - 1];
            m.end_hist[m.known % Impl::END_HIST] = m.known_pos;
        }
        // upper bound of where batch nbq ends (+ the window slack): that much of the stream must exist before its tokenizer starts
        const uint64_t hi_end = m.known_pos + (uint64_t)(nbq - m.known) * MTDEV_BATCH * m.wmax + m.slack;
        while (m.gen_words < hi_end) {
            // ---- one round: window -> jumps -> P segments ----
            const uint64_t want = hi_end - m.gen_words;
            int P = (int)std::min<uint64_t>(PMAX, (want + SEG_WORDS - 1) / SEG_WORDS + 1);
            P = std::max(P, 2);
            // The ring positions this round overwrites: everything below gen_words + round - capw.  (a) Batches already enqueued
            // may still be reading there (their tokenizer, their finisher): the round waits for the finisher of every one of them
            // that could start that low.  (b) Words not yet consumed -- everything from the end of the last enqueued batch on --
            // must not be there at all: checked against the LOWER bound of that end, with the exact position fetched if the bound
            // is not good enough.
            const uint64_t round_words = (m.rounds == 0 ? MT_N : 0) + (uint64_t)P * SEG_WORDS;
            if (m.gen_words + round_words > m.capw) {
                const uint64_t dead_below = m.gen_words + round_words - m.capw;
                for (;;) {
                    const uint64_t lo_next = m.known_pos + (uint64_t)(nbq - 1 - m.known) * MTDEV_BATCH * m.wmin;     // end of batch nbq - 1
                    if (lo_next >= dead_below + MT_N) break;
                    if (m.known + 1 >= nbq) {
                        err_ = "mtdev: the stream ring is too small for this configuration";
                        return -1;
                    }
                    const double t_ = tr_ms();
                    if (!wait_positions(m.known + 1)) return -9;
                    if (TR) fprintf(stderr, "   wait B for pos(%lld) for nbq %lld: %.3f ms\n", (long long)m.known + 1, (long long)nbq, tr_ms() - t_);
                    ++m.known;
                    m.known_pos = m.h_end[(size_t)(m.known % MTDEV_NBUF) * MTDEV_BATCH + MTDEV_BATCH - 1];
                    m.end_hist[m.known % Impl::END_HIST] = m.known_pos;
                }
                int64_t qlast = -1;
                for (int64_t q = std::max<int64_t>(0, m.enq - MTDEV_NBUF); q < m.enq; ++q) {
                    // where batch q starts: exactly (the end of batch q - 1, when the host has read it and it is still in h_end), or
                    // a lower bound from the last position known
                    bool maybe_live = true;
                    if (q == 0) {
                        maybe_live = m.pos0 < dead_below + MT_N;
                    } else if (q - 1 <= m.known) {
                        if (q - 1 > m.known - Impl::END_HIST) maybe_live = m.end_hist[(q - 1) % Impl::END_HIST] < dead_below + MT_N;
                    } else {
                        maybe_live = m.known_pos + (uint64_t)(q - 1 - m.known) * MTDEV_BATCH * m.wmin < dead_below + MT_N;
                    }
                    if (!maybe_live) continue;
                    if (m.use_gates)
                        qlast = q;
                    else
                        MTD_HIP(hipStreamWaitEvent(m.s_gen, m.bt[q % MTDEV_NBUF].fin, 0));
                }
                if (TR) fprintf(stderr, "   gen round for batch %lld: dead_below %llu known %lld known_pos %llu -> waits for fin(%lld)\n", (long long)nbq, (unsigned long long)dead_below, (long long)m.known, (unsigned long long)m.known_pos, (long long)qlast);
                if (qlast >= 0) gate_wait(m.s_gen, G_FIN, (unsigned long long)qlast + 1ull, 0);        // (finishers complete in order)
            }
            const bool first = m.rounds == 0;
            const uint64_t first_word = first ? MT_N : m.gen_words;       // (the base block of round 0 is words [0, 624))
            hipLaunchKernelGGL(k_mt_window, dim3(1), dim3(256), 0, m.s_gen, m.base_key, m.xwin, m.stream, (unsigned long long)(m.capw - 1),
                               0ull, first ? 1 : 0);
            hipLaunchKernelGGL(k_mt_jump, dim3(JUMP_SPLIT, P - 1), dim3(256), 0, m.s_gen, m.polys, m.xwin, m.partial);
            GenArgs ga{};
            ga.xwin = m.xwin;
            ga.partial = m.partial;
            ga.stream = m.stream;
            ga.smask = m.capw - 1;
            ga.first_word = first_word;
            ga.next_base = m.base_key;
            ga.last_seg = P - 1;
            ga.blocks_per_seg = SEG_BLOCKS;
            hipLaunchKernelGGL(k_mt_gen, dim3(P), dim3(256), 0, m.s_gen, ga);
            MTD_HIP(hipGetLastError());
            MTD_HIP(hipEventRecord(m.ev_gen, m.s_gen));
            m.gen_words = first_word + (uint64_t)P * SEG_WORDS;
            ++m.rounds;
            if (m.use_gates) gate_signal(m.s_gen, G_GEN, (unsigned long long)m.rounds);
            st_.rounds++;
            st_.segments += P;
        }
        // ---- tokenizer of batch nbq: after the stream it needs, and after the finisher that last read its J / rint / tokpos ----
        if (m.use_gates) {
            gate_wait(m.s_tok, G_GEN, (unsigned long long)m.rounds, 1);
            if (nbq >= 2) gate_wait(m.s_tok, G_FIN, (unsigned long long)nbq - 1ull, 2);
        } else {
            MTD_HIP(hipStreamWaitEvent(m.s_tok, m.ev_gen, 0));
            if (nbq >= 2) MTD_HIP(hipStreamWaitEvent(m.s_tok, m.bt[(nbq - 2) % MTDEV_NBUF].fin, 0));
        }
        TokArgs ta{};
        ta.stream = m.stream;
        ta.smask = m.capw - 1;
        ta.pos = m.d_pos;
        ta.avail_end = m.gen_words;
        ta.status = m.status;
        ta.err = m.d_err;
        ta.recs = m.recs[par];
        ta.nrec = m.nrec[par];
        ta.maxrec = m.maxrec;
        ta.tokpos = m.tokpos[par];
        ta.step_end = m.step_end[buf];
        ta.stats = m.d_nwin;
        ta.J = m.J[par];
        ta.wshift = m.wshift;
        ta.tail = m.tail;
        ta.N = (int32_t)m.N;
        ta.S = m.S;
        ta.nb = MTDEV_BATCH;
        ta.randomize = m.mv.randomize_split ? 1 : 0;
        hipLaunchKernelGGL(k_mt_tok, dim3(1), dim3(TOK_T), 0, m.s_tok, ta);
        MTD_HIP(hipGetLastError());
        MTD_HIP(hipEventRecord(m.bt[buf].tok, m.s_tok));
        hipLaunchKernelGGL(k_pos_publish, dim3(1), dim3(64), 0, m.s_tok, (const unsigned long long*)m.step_end[buf], m.h_end + (size_t)buf * MTDEV_BATCH,
                           (int)MTDEV_BATCH, m.h_done, (unsigned long long)nbq + 1ull);
        MTD_HIP(hipGetLastError());
        MTD_HIP(hipEventRecord(m.bt[buf].pos, m.s_tok));
        if (m.use_gates) gate_signal(m.s_tok, G_TOK, (unsigned long long)nbq + 1ull);
        // ---- finisher: after the tokenizer, and after the consumer's last read of the plan slots it rewrites ----
        if (m.use_gates) {
            gate_wait(m.s_fin, G_TOK, (unsigned long long)nbq + 1ull, 3);
            if (m.bt[buf].has_released) gate_wait(m.s_fin, G_REL, (unsigned long long)(nbq - MTDEV_NBUF) + 1ull, 4);
        } else {
            MTD_HIP(hipStreamWaitEvent(m.s_fin, m.bt[buf].tok, 0));
            if (m.bt[buf].has_released) MTD_HIP(hipStreamWaitEvent(m.s_fin, m.bt[buf].released, 0));
        }
        m.bt[buf].has_released = false;
        FinArgs fa{};
        fa.stream = m.stream;
        fa.smask = m.capw - 1;
        fa.recs = m.recs[par];
        fa.nrec = m.nrec[par];
        fa.J = m.J[par];
        fa.rint = m.rint[par];
        fa.tokpos = m.tokpos[par];
        fa.err = m.d_err;
        fa.scratch = m.scratch;
        fa.partial = m.fin_partial;
        fa.hist = m.fin_hist;
        for (int k = 0; k < MTDEV_BATCH; ++k) {
            const MtDevPlanCols& pc = m.slots[(size_t)buf * MTDEV_BATCH + k];
            fa.order[k] = pc.order;
            fa.p0[k] = pc.p0;
            fa.s0[k] = pc.s0;
            fa.uacc[k] = pc.uacc;
            fa.logu[k] = pc.logu;
            fa.fac[k] = pc.fac;
        }
        fa.step_end = m.step_end[buf];
        fa.blk_words = m.blk_words + (size_t)buf * MTDEV_BATCH * MT_N;
        fa.a = m.mv.a;
        fa.N = (int32_t)m.N;
        fa.D = m.D;
        fa.S = m.S;
        fa.randomize = m.mv.randomize_split ? 1 : 0;
        fa.maxrec = m.maxrec;
        fa.nchunk = m.nchunk;
        {
            const dim3 gc((unsigned)m.nchunk, MTDEV_BATCH), blk(FIN_T);
            hipLaunchKernelGGL(k_fin_init, gc, blk, 0, m.s_fin, fa);
            hipLaunchKernelGGL(k_fin_walk, dim3((unsigned)m.maxrec * 4u, MTDEV_BATCH), blk, 0, m.s_fin, fa);
            if (fa.randomize) {
                hipLaunchKernelGGL(k_fin_hits, gc, blk, 0, m.s_fin, fa);
                hipLaunchKernelGGL(k_fin_sum, gc, blk, 0, m.s_fin, fa);
                hipLaunchKernelGGL(k_fin_scan, gc, blk, 0, m.s_fin, fa);
                hipLaunchKernelGGL(k_fin_bucket, gc, blk, 0, m.s_fin, fa);
            }
            hipLaunchKernelGGL(k_fin_label, gc, blk, 0, m.s_fin, fa);
            hipLaunchKernelGGL(k_fin_order, gc, blk, 0, m.s_fin, fa);
            hipLaunchKernelGGL(k_fin_plan, dim3((unsigned)((m.N + FIN_T - 1) / FIN_T), MTDEV_BATCH), blk, 0, m.s_fin, fa);
        }
        MTD_HIP(hipGetLastError());
        MTD_HIP(hipEventRecord(m.bt[buf].fin, m.s_fin));
        if (m.use_gates) gate_signal(m.s_fin, G_FIN, (unsigned long long)nbq + 1ull);
        MTD_HIP(hipGetLastError());
        ++m.enq;
        st_.batches++;
        if (TR) fprintf(stderr, "   enqueued batch %lld (gen_words %llu, known %lld) +%.3f ms\n", (long long)nbq, (unsigned long long)m.gen_words, (long long)m.known, tr_ms() - tr0);
    }
    if (m.enq <= b) {
        err_ = "mtdev: batch not produced";
        return -1;
    }
    if (m.use_gates) {
        gate_wait(consumer, G_FIN, (unsigned long long)b + 1ull, 5);
        MTD_HIP(hipGetLastError());
    } else {
        MTD_HIP(hipStreamWaitEvent(consumer, m.bt[b % MTDEV_NBUF].fin, 0));
    }
    return 0;
}

int MtDevProducer::release_batch(int64_t b, hipStream_t consumer) {
    Impl& m = *im_;
    MTD_HIP(hipSetDevice(m.device));
    const int buf = (int)(b % MTDEV_NBUF);
    MTD_HIP(hipEventRecord(m.bt[buf].released, consumer));
    if (m.use_gates) {
        hipLaunchKernelGGL(k_gate_signal, dim3(1), dim3(64), 0, consumer, m.gates + G_REL, (unsigned long long)b + 1ull);
        MTD_HIP(hipGetLastError());
    }
    m.bt[buf].has_released = true;
    return 0;
}

int MtDevProducer::finish(int64_t steps_taken, MT19937Legacy& out) {
    Impl& m = *im_;
    MTD_HIP(hipSetDevice(m.device));
    MTD_HIP(hipStreamSynchronize(m.s_gen));
    MTD_HIP(hipStreamSynchronize(m.s_tok));
    MTD_HIP(hipStreamSynchronize(m.s_fin));
    refresh_stats();
    out = m.start;
    if (steps_taken <= 0) return 0;
    unsigned e = 0;
    MTD_HIP(hipMemcpy(&e, m.d_err, 4, hipMemcpyDeviceToHost));
    if (e & 4u) {
        err_ = "mtdev: a stage of the producer waited 20 s for another (k_gate_wait, status bit 4): the run is void";
        return -9;
    }
    if (e) {
        err_ = "mtdev: the generated stream ran out under the tokenizer (status bit 4): the run is void";
        return -9;
    }
    const int64_t bq = (steps_taken - 1) / MTDEV_BATCH;
    if (bq >= m.enq || bq < m.enq - MTDEV_NBUF) {
        err_ = "mtdev: the batch of the last step taken is no longer (or not yet) in the buffers";
        return -1;
    }
    const uint64_t a = m.h_end[(size_t)(bq % MTDEV_NBUF) * MTDEV_BATCH + (size_t)((steps_taken - 1) % MTDEV_BATCH)];
    // NumPy get_state(): the key of the block the position stands in; a block consumed to its end reports pos = 624
    const uint64_t blk = a > 0 ? (a - 1) / MT_N : 0;
    uint32_t words[MT_N], key[MT_N];
    MTD_HIP(hipMemcpy(words, m.blk_words + ((size_t)(bq % MTDEV_NBUF) * MTDEV_BATCH + (size_t)((steps_taken - 1) % MTDEV_BATCH)) * MT_N, MT_N * 4,
                      hipMemcpyDeviceToHost));
    for (int i = 0; i < MT_N; ++i) key[i] = untemper32(words[i]);          // tempering is a bijection: the state words behind the outputs
    out.set_state(key, (int)(a - blk * MT_N), m.start.has_gauss, m.start.gauss);
    return 0;
}

int MtDevProducer::debug_stream(uint64_t first_word, int64_t n, uint32_t* out) {
    Impl& m = *im_;
    MTD_HIP(hipSetDevice(m.device));
    MTD_HIP(hipDeviceSynchronize());
    for (int64_t k = 0; k < n;) {
        const uint64_t off = (first_word + (uint64_t)k) & (m.capw - 1);
        const int64_t take = (int64_t)std::min<uint64_t>((uint64_t)(n - k), m.capw - off);
        MTD_HIP(hipMemcpy(out + k, m.stream + off, (size_t)take * 4, hipMemcpyDeviceToHost));
        k += take;
    }
    return 0;
}

int MtDevProducer::debug_targets(int64_t step, uint32_t* out) {
    Impl& m = *im_;
    MTD_HIP(hipSetDevice(m.device));
    MTD_HIP(hipDeviceSynchronize());
    const int64_t bq = step / MTDEV_BATCH;
    if (bq >= m.enq || bq < m.enq - 2) {
        err_ = "mtdev: that step's targets are not in the buffers";
        return -1;
    }
    MTD_HIP(hipMemcpy(out, m.J[bq & 1] + (size_t)(step % MTDEV_BATCH) * m.N, (size_t)m.N * 4, hipMemcpyDeviceToHost));
    return 0;
}

int MtDevProducer::debug_positions(int64_t step, uint64_t* tokpos, uint64_t* end) {
    Impl& m = *im_;
    MTD_HIP(hipSetDevice(m.device));
    MTD_HIP(hipDeviceSynchronize());
    const int64_t bq = step / MTDEV_BATCH;
    if (bq >= m.enq || bq < m.enq - 2) {
        err_ = "mtdev: that step's positions are not in the buffers";
        return -1;
    }
    MTD_HIP(hipMemcpy(tokpos, m.tokpos[bq & 1] + (size_t)(step % MTDEV_BATCH) * m.S * 3, (size_t)m.S * 3 * 8, hipMemcpyDeviceToHost));
    MTD_HIP(hipMemcpy(end, m.step_end[bq % MTDEV_NBUF] + (step % MTDEV_BATCH), 8, hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace emx
