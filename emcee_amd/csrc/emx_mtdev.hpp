// Exact (MT19937) mode: the step plans produced ON THE DEVICE -- no host thread touches a draw.
//
// Reference emcee draws everything from ONE serial NumPy-legacy stream (ensemble.py:166-167,406, moves/red_blue.py:76-80,100,
// moves/stretch.py:30-32); emx_mtpipe.hpp explains what is serial in it (the recurrence, and the stream POSITION wherever a draw
// is rejection sampled) and makes the plans with host threads -- two serial stages of ~1 ns per walker and step.  Here the same
// plans, bit for bit, come from kernels (emx_mtdev.hip, emx_mtdev_kernels.hpp):
//
//   k_mt_window   one workgroup: the 33 blocks after the base state, untempered -- the window every jump reads
//   k_mt_jump     the state `k * stride` words ahead, k = 1 .. P-1, as the GF(2) convolution of the window with the jump
//                 polynomial t^(k stride) mod phi (emx_mtjump.hpp; the polynomials are computed once per process on the host)
//   k_mt_gen      P <= 128 workgroups twist + temper their segments (128 blocks each) of the stream into an HBM ring: the stream
//                 of a whole batch of steps is there before anything consumes it
//   k_mt_tok      ONE workgroup walks the stream in the reference's draw order and does only what decides the position: the
//                 masked rejection tests of random.shuffle (red_blue.py:80) and of a non-power-of-two randint (stretch.py:32).
//                 A window of 1024 x {13, 7, 3, 1} words is decided at once: every thread runs its consecutive words exactly, from
//                 a guessed count of accepts before them; counts are prefix-summed and the guesses replaced until nothing changes
//                 -- a fixed point of that iteration IS the serial result -- and a thread recomputes only when its count of
//                 earlier accepts moved further than the smallest margin of its own tests; the last ~2 000 indices are walked
//                 by one wave with ballots.  It leaves per-window WALK RECORDS (each thread's starting index) behind
//   k_fin_*       the finisher, nine small kernels over (chunk, step) grids (steps are independent once tokenised, red_blue.py:78
//                 re-initialises the labels): replay of the walk records into the Fisher-Yates targets J[i]; the swaps applied
//                 in parallel -- position i's final label is traced back through the swaps that hit it (buckets of swap
//                 targets, chains are O(1) long on average); the boolean-mask order (red_blue.py:85); partner resolution
//                 (stretch.py:27,32-33) and the conversions zz / u / ln u / (D-1) ln zz (stretch.py:30-31, red_blue.py:100)
//
// The tokenizer is the serial stage and bounds the producer: ~1.2 ns per walker and step at 65 536 walkers (~46 fixed-point rounds of
// ~1 us), ~0.6 ns at 10^6 (wider masks: fewer flips a round) -- against the host pipeline's ~1.05 ns at every size.  Hence the
// default: the device producer from 131 072 walkers on.
// Scope: ONE StretchMove (any a, nsplits <= 64, randomize_split), one replica; everything else keeps the host pipeline.
// tests/test_gpu_mtdev.py holds every stage equal to the serial host twin (MT19937Legacy / make_exact_plan).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <deque>
#include <string>
#include <vector>

#include "../../include/emx.h"
#include "mt19937_legacy.hpp"

namespace emx {

constexpr int MTDEV_BATCH = 16;          // steps per produced batch (== NATIVE_BATCH_MAX: one persistent launch)
constexpr int MTDEV_NBUF = 4;            // batches of plan slots (the producer runs up to three batches ahead of the consumer)

struct MtDevPlanCols {                   // where a step's plan goes (device memory, N entries each; plan order)
    int32_t *order, *p0;
    double *s0, *uacc, *logu, *fac;
};

struct MtDevStats {
    int64_t rounds = 0, segments = 0, batches = 0, windows = 0, tok_rounds = 0, tail_groups = 0, tail_rounds = 0;
    int64_t tok_ticks[4] = {0, 0, 0, 0};      // 10 ns ticks of the tokenizer: waiting for windows | chunk windows | tail | whole kernels
    double poly_ms = 0.0;
};

class MtDevProducer {
   public:
    static bool supports(int64_t N, int32_t nmoves, const emx_move_desc* moves);

    // slots: MTDEV_NBUF * MTDEV_BATCH plan slots; step n of the stream (0-based from `start`) goes to slots[n % (NBUF * BATCH)]
    MtDevProducer(int device, const MT19937Legacy& start, int64_t N, int32_t D, const emx_move_desc& mv, const MtDevPlanCols* slots,
                  uint32_t* status_dev);
    ~MtDevProducer();
    MtDevProducer(const MtDevProducer&) = delete;
    MtDevProducer& operator=(const MtDevProducer&) = delete;
    bool ok() const { return err_.empty(); }
    // the tokenizer's window rule (tuning): words per thread = (mask + 1) >> wshift; indices <= tail are walked by one wave
    void set_window_rule(int wshift, int tail);
    const std::string& error() const { return err_; }

    // Make sure batch b (steps [16 b, 16 b + 16)) and up to `lookahead` batches after it are enqueued; `consumer` then waits
    // (stream order) for batch b's plans.  Batches must be asked for in order.  Returns 0 or a negative code (error()).
    int ensure_batch(int64_t b, hipStream_t consumer, int lookahead = 2);
    // the consumer has enqueued its last read of batch b's slots on `consumer`
    int release_batch(int64_t b, hipStream_t consumer);
    // Stop: `out` = generator state after `steps_taken` steps of the stream (NumPy get_state() semantics).  Synchronises.
    int finish(int64_t steps_taken, MT19937Legacy& out);
    const MtDevStats& stats() const { return st_; }
    void refresh_stats();                  // the tokenizer's counters as of now (waits for the tokenizer's stream)

    // ---- debugging / tests: raw pieces, after a synchronise -------------------------------------------------------------
    int debug_stream(uint64_t first_word, int64_t n, uint32_t* out);               // tempered words [first, first + n) of the stream
    int debug_targets(int64_t step, uint32_t* out);                                // J[i], i < N, of a step still in its buffer
    int debug_positions(int64_t step, uint64_t* tokpos /* [S][3] */, uint64_t* end);

   private:
    struct Impl;
    Impl* im_ = nullptr;
    std::string err_;
    MtDevStats st_;
};

}  // namespace emx
