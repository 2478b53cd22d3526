// Exact (MT19937) mode: the step plans of a whole emx_run produced by a host-side pipeline.
//
// Reference emcee draws everything from ONE serial NumPy-legacy stream (ensemble.py:406, red_blue.py:80,100,
// stretch.py:30-32, de.py:49-56, de_snooker.py:37-40).  No draw depends on the walker positions, so the plans of
// future steps can be made while the GPU works on earlier ones; what is serial is (a) the MT19937 recurrence and
// (b) the *position* in the stream, which rejection sampling (shuffle, non-power-of-two randint, polar normals)
// makes data dependent.  The pipeline splits the work accordingly:
//
//   generator  (1 thread)  twists 624-word blocks of generator STATE words into a ring, SIMD (AVX-512 / AVX2 clones); tempering is
//                          the consumers' (round 5: half of the vector work of a block)
//   tokenizer  (1 thread)  walks the stream once, in the reference's draw order, and does ONLY what decides the
//                          stream position: the rejection tests.  It emits tokens: accepted Fisher-Yates targets
//                          j_i, accepted randint values, accepted polar pairs (x, r2) -- and the POSITIONS of the fixed-length
//                          draws, which stay in the ring (no copy)
//   finishers  (K threads) one step each, independent of each other (labels are re-initialised every step,
//                          red_blue.py:78): apply the swaps, counting-sort the split into plan order, resolve
//                          complement indices to walkers, decode DE pairs, finish the polar normals -- straight into
//                          the pinned staging buffer of the plan slot
//
// and emx_run only waits for slot n, enqueues its upload and the half-step kernels.  Same draws, same order, same
// arithmetic as MT19937Legacy / make_exact_plan: the plans are bit-identical (tests/test_mt_pipeline_cpu.py, and every
// exact-mode GPU test runs through it).
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/emx.h"
#include "mt19937_legacy.hpp"

namespace emx {

// moves/de.py:67-77 in closed form: the k-th row of the table of ordered pairs (i, j), i != j,
// of range(nc) -- first the pairs with i > j in lexicographic order, then their mirror images
inline void de_pair(uint64_t k, uint64_t nc, uint64_t& first, uint64_t& second) {
    const uint64_t T = nc * (nc - 1) / 2;
    const uint64_t kk = k < T ? k : k - T;
    uint64_t i = (uint64_t)((1.0 + std::sqrt(1.0 + 8.0 * (double)kk)) / 2.0);
    while (i * (i - 1) / 2 > kk) --i;
    while ((i + 1) * i / 2 <= kk) ++i;
    const uint64_t j = kk - i * (i - 1) / 2;
    if (k < T) {
        first = i;
        second = j;
    } else {
        first = j;
        second = i;
    }
}

// where the plan of one step goes (plan order; all arrays of length N)
struct PlanSink {
    int32_t *order = nullptr, *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
    double *s0 = nullptr, *uacc = nullptr;
};

struct PipeStepInfo {
    int32_t move = 0, S = 0;
    int32_t off[66] = {0};
    // raw (a stretch step of a pipeline constructed with device_finish, S <= PIPE_RAW_SPLITS): the sink holds `order`; the fixed-length
    // draws are passed on as the MT19937 STATE words they were made of (tempering is the reader's), in plan order, in the columns
    // they will stand in -- s0: the two words of rand() for the stretch factor (stretch.py:30), uacc: the two of the accept uniform
    // (red_blue.py:100), p0: the word of a power-of-two randint (stretch.py:32; wr_ring = 1) or the accepted randint value itself
    // (wr_ring = 0); p0 is a member number of the complement either way, not yet a walker.  The consumer converts (k_plan_raw).
    int32_t raw = 0, wr_ring = 0;
    // regen (round 6; a raw stretch step of two splits whose complements are powers of two, ensembles of `regen_min_walkers` and more):
    // the 5 N fixed-length words behind the shuffle -- rand(Ns), randint(Nc, Ns), Ns x rand() per split, stretch.py:30-32 and
    // red_blue.py:100, contiguous in the stream -- do not cross to the tokenizer's core, the staging buffer and PCIe at all: the sink
    // holds `order` and, in the place of its p0 column, the generator STATE (624 words) at every PIPE_REGEN_SB-th block of that
    // region; the consumer runs the recurrence forward from each (k_plan_regen, csrc/emx_kernels.hpp) and drops the words where
    // k_plan_raw expects them.  regen_off: the region's first word inside the first state's block; regen_nseg: states in the sink.
    int32_t regen = 0, regen_off = 0, regen_nseg = 0;
};
constexpr int PIPE_RAW_SPLITS = 8;
constexpr int PIPE_REGEN_SB = 8;          // stream blocks (of 624 words) regenerated from one state: one workgroup of k_plan_regen each

class MtPlanPipeline {
   public:
    // `sinks`: nsinks staging buffers used round-robin (step n -> sinks[n % nsinks]); a sink is rewritten only after
    // release(n - nsinks).  moves must all be stretch / DE / snooker.  nworkers <= 0: chosen from the core count.
    // device_finish: stretch steps are handed over raw (PipeStepInfo::raw) -- the finishers apply the swaps, build `order` and copy
    // the step's generator words into the sink; conversions and partner resolution are the consumer's.
    MtPlanPipeline(const MT19937Legacy& start, int64_t N, int32_t D, int32_t nmoves, const emx_move_desc* moves,
                   const double* cdf, int64_t nsteps, const PlanSink* sinks, int32_t nsinks, int32_t nworkers,
                   bool fill_unused_fields = false,       // true: a stretch plan's p1 / p2 are set to the walker itself
                   bool device_finish = false,
                   bool bursty_consumer = false,         // the consumer takes its steps sixteen at a time (the persistent kernels): the stage threads
                                                         // spin through the gaps between bursts instead of napping
                   int64_t regen_min_walkers = 0);       // > 0 (with device_finish): eligible steps of ensembles this large are handed over as
                                                         // generator states (PipeStepInfo::regen); 0: never
    ~MtPlanPipeline();
    MtPlanPipeline(const MtPlanPipeline&) = delete;
    MtPlanPipeline& operator=(const MtPlanPipeline&) = delete;

    static bool supports(int32_t nmoves, const emx_move_desc* moves);

    // Block until the plan of step n (0-based, in order) is complete in its sink.  `poll`, if given, is called while
    // waiting (emx_run retires upload events there).  Returns false if the pipeline failed.
    bool wait_ready(int64_t n, PipeStepInfo& info, void (*poll)(void*) = nullptr, void* poll_arg = nullptr);
    // the sink of step n has been uploaded and may be overwritten
    void release(int64_t n);
    // Stop all threads; `out` receives the generator state after `steps_consumed` steps (NumPy get_state() semantics).
    void finish(int64_t steps_consumed, MT19937Legacy& out);
    int workers() const;
    // microseconds per produced step: [0] wall, [1] generator busy, [2] tokenizer busy, [3] finishers busy (summed), [4] tokenizer
    // waiting for words, [5] tokenizer waiting for a free staging buffer
    void stage_times(double out[6], int64_t* steps) const;

   private:
    struct Impl;
    Impl* impl_;
};

}  // namespace emx
