/* libemx -- C ABI of the MI355X split-ensemble sampler hot path.
 *
 * The reference (dfm/emcee) has no FFI: its boundary for this path is the duck-typed Python
 * protocol  EnsembleSampler.sample -> Move.propose(model, state) -> model.compute_log_prob_fn
 * (SURVEY.md 8b).  Each entry point below names the reference code it replaces (paths relative
 * to /root/reference/src/emcee).  The Python host layer (emcee_amd/) binds these through ctypes;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions: every function returns 0 on success and a negative code on error, with a
 * message available from emx_last_error(); no C++ exception crosses the boundary; the caller
 * owns all host buffers, the library owns all device buffers; a context is bound to one
 * device and is not thread-safe; all device work is asynchronous on the context's stream
 * except calls that copy results to host memory.  Arrays are C-contiguous float64 unless
 * noted.  There is NO CPU fallback: without a usable HIP device emx_create fails.
 */
#ifndef EMX_H
#define EMX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct emx_ctx emx_ctx;

enum emx_target_kind {
    EMX_TARGET_HOST = 0,       /* log-prob evaluated by the caller (split-phase API)            */
    EMX_TARGET_ISO_GAUSS = 1,  /* -0.5 sum x^2            tests/integration/test_proposal.py:21 */
    EMX_TARGET_DIAG_GAUSS = 2, /* -0.5 sum ivar (x-mu)^2  docs/index.rst:41-45                 */
    EMX_TARGET_DENSE_GAUSS = 3,/* -0.5 (x-mu)^T icov (x-mu)  docs/tutorials/quickstart.ipynb:76 */
    EMX_TARGET_ROSENBROCK = 4, /* -sum[100 (x_{i+1}-x_i^2)^2 + (1-x_i)^2] / scale (BASELINE C3) */
    EMX_TARGET_BOX = 5,        /* 0 inside [0,1]^D else -inf   test_proposal.py:25-28           */
    EMX_TARGET_DEVICE_CALLBACK = 6 /* the caller's batched log-prob on device buffers (emx_set_target_callback) */
};

enum emx_move_kind {
    EMX_MOVE_STRETCH = 0, EMX_MOVE_DE = 1, EMX_MOVE_SNOOKER = 2,
    /* moves/gaussian.py + moves/mh.py: Metropolis step with an isotropic / axis-aligned Gaussian proposal,
     * every walker at once from its own position (nsplits must be 1).  Fields: reserved = mode
     * (emx_gauss_mode), sigma = isotropic standard deviation (emx_set_move_scale installs a per-coordinate
     * vector instead), a != 0 enables the step-size factor exp(U(-g0, g0)) with g0 = ln(factor)
     * (gaussian.py:81-84), gammas = the sequential mode's coordinate cursor (gaussian.py:96-97; the library
     * advances it, emx_get_move reads it back). */
    EMX_MOVE_GAUSS = 3
};
enum emx_gauss_mode { EMX_GAUSS_VECTOR = 0, EMX_GAUSS_RANDOM = 1, EMX_GAUSS_SEQUENTIAL = 2 };

enum emx_rng_mode {
    EMX_RNG_INPUTS = 0,  /* every step's plan is supplied with emx_plan_set                     */
    EMX_RNG_MT19937 = 1, /* NumPy legacy RandomState stream: same seed => same chain as emcee   */
    EMX_RNG_PHILOX = 2   /* counter-based, generated inside the kernels (throughput mode)       */
};

/* moves/red_blue.py:37-42, moves/stretch.py:22, moves/de.py:28-31,33-38, moves/de_snooker.py:26-29 */
typedef struct emx_move_desc {
    int32_t kind;            /* emx_move_kind                                   */
    int32_t nsplits;         /* RedBlueMove.nsplits (snooker: 4)                */
    int32_t randomize_split; /* RedBlueMove.randomize_split                     */
    int32_t reserved;
    double a;                /* StretchMove.a                                   */
    double sigma;            /* DEMove.sigma                                    */
    double g0;               /* DEMove.g0 = gamma0 or 2.38/sqrt(2 ndim)         */
    double gammas;           /* DESnookerMove.gammas                            */
} emx_move_desc;

/* ---- library / context ---------------------------------------------------------------- */
const char* emx_version(void);
const char* emx_last_error(const emx_ctx* ctx); /* ctx may be NULL: last creation error */
int emx_device_count(int32_t* n);
/* EnsembleSampler.__init__ (ensemble.py:79-137): one context per (device, ensemble). */
int emx_create(int32_t device, int64_t nwalkers, int32_t ndim, emx_ctx** out);
int emx_destroy(emx_ctx* ctx);
/* adopt an external hipStream_t (e.g. torch's current stream); NULL restores the own stream */
int emx_set_stream(emx_ctx* ctx, void* hip_stream);
int emx_sync(emx_ctx* ctx);
/* sticky device status: bit0 NaN log-prob (ensemble.py:550-551), bit1 non-finite coordinate
 * (ensemble.py:476-479), bit2 pull-exchange record capacity exceeded (a >8 sigma event: the run is
 * invalid, never silently wrong), bit3 direct / replay exchange: a peer did not reach the device-side barrier in time
 * (raised again by every later barrier of that attachment), bit4 the device producer of exact-mode plans (rng mode MT19937,
 * large ensembles) stalled -- a stage waited 20 s for another -- or its stream ran out under the tokenizer: the steps taken from
 * it are void; the call that retires the producer (emx_run's next start, emx_rng_get_mt19937, emx_set_moves ...) returns the
 * error as well and leaves the generator where it stood before the producer started.  Reading clears it. */
int emx_status(emx_ctx* ctx, uint32_t* bits);
/* Tuning keys (A/B measurements, parity tests; defaults are what the product runs).  An unknown key is an error.  The environment
 * variable EMX_TUNE="key=value,key=value" applies keys to every context at creation.  One table -- nothing measured-and-rejected is
 * left behind a key (round 6: the p2p form, the fused wide propose and the split upload were removed from the library):
 *
 *   key                        default   meaning
 *   -- launch shape of the per-half-step kernels --
 *   "spw"                      0 (auto)  slots per wave                      "blocks_per_cu"     2         workgroups a CU is given
 *   "waves_per_block"          0 (auto)  1 / 2 / 4 / 8                       "throttle"          0         half-steps in flight (0: unbounded)
 *   "graph"                    0         1: a 16-step block of Philox launches replayed as a hipGraph
 *   "prep_hint"                1         steps a caller of emx_step_begin will take (Philox plan batch size)
 *   "full_plan"                0         1: Philox plans carry every column (default: the ones the step's fused kernel reads)
 *   "small_kernel"             1         ensembles that fit one workgroup's LDS run whole emx_run calls in one workgroup
 *   "gauss_materialize"        0         Gaussian move: 1: proposals through memory (parity tests of the register form)
 *   -- dense targets --
 *   "dense_wide"               0         1: the propose / log-prob / commit path whatever the ndim; 2: ... with the single-role log-prob kernel
 *   "slab"                     1         0: never the slab form (emx_slab.hip); 1: from padded ndim 112; 2: from padded ndim 80
 *   "slab_skew"                1         0 ... 4: when the second wave of a SIMD starts its first tile's row loads
 *   -- persistent kernels (emx_persist_info) --
 *   "persist"                  1         0: never a persistent launch
 *   "persist_local"            1         0: never the one-XCD form        "persist_local_max_walkers"  8192
 *   "persist_valu"             1         0: element-wise targets on the per-half-step launches
 *   "persist_slab"             1         0: dense targets of padded ndim 80 ... 128 on the per-half-step launches (1: k_persist_slab, emx_pslab.hip)
 *   "persist_slab_skew"        1         k_persist_slab: 0 ... 4, as "slab_skew"   "persist_slab_local_max_walkers"  4096   its one-XCD form's largest ensemble
 *   "persist_odd"              1         0: dense targets of odd ndim on the per-half-step launches (1: k_persist with one coordinate per lane, emx_podd.hip)
 *   "persist_mix"              1         0: DE and snooker steps of a mixture in launches of their own
 *   "persist_span"             1         0: a launch ends with its Philox plan batch
 *   "persist_min_walkers"      512       smallest ensemble             "persist_timeout_ms"   2000      bound of a barrier wait
 *   "persist_rows_late"        1         launches that store chain rows (stretch move, even ndim <= 64) ask for the next half-step's own rows behind the
 *                                        MFMA phase instead of in front of it (k_persist<..., ROWS_LATE>): 1 the one-XCD form and device-wide launches
 *                                        with eight tiles a CU and half-step (65 536 walkers), 2 always, 0 never
 *   "persist_stagger"          -1        how * 256 + n: some waves of a k_persist / k_persist_mix workgroup ask for their partner rows n x 64 clocks
 *                                        after the others (how 0: waves 4-7, 1: odd waves, 2: the waves of SIMDs 2 and 3, 4: SIMD k waits k n); -1: 516
 *                                        for device-wide stretch launches without stored rows, 528 for the DE move and DE + snooker mixtures (k_persist_mix), else 0
 *                                        (profiles/r06/stagger_ab.md)
 *   "persist_max_halfsteps"    40        half-steps a launch may hold (<= 40: twenty stretch / DE steps; a launch never reads plans of more than two batches)
 *   "persist_gauss_wpb"        0 (auto)  waves per workgroup of k_persist_gauss
 *   "persist_exact"            1         0: exact mode (EMX_RNG_MT19937) on the per-half-step launches with an upload per step
 *   "persist_exact_mix"        1         0: ... for one move only      "persist_exact_steps"  16        steps per launch (<= 16)
 *   "persist_exact_max_walkers" 32768    largest ensemble of the device-wide form in exact mode
 *   "fetch_blocks"             64        k_plan_fetch's workgroups beside a device-wide launch (0: one per 256 entries)
 *   "fetch_avoid"              1         k_plan_fetch's workgroups decline on the XCD of a one-XCD launch
 *   -- exact-mode plan producers --
 *   "mt_pipeline"              -1        -1: host pipeline, finisher threads from the core count; k > 0: k finishers; 0: inline, calling thread
 *   "mt_device_finish"         1         0: the finisher threads convert every draw (1: k_plan_raw does, on the device)
 *   "mt_regen_min_walkers"     16384     stretch steps of ensembles this large go up as generator STATES (k_plan_regen makes the draws again); 0: never
 *   "mt_device"                1         0: never the device producer; 1: from "mt_device_min_walkers" (147456) on -- from "mt_device_min_walkers_regen"
 *                                        (786432) where the host pipeline's stretch steps are regen steps; 2: from 8192 on
 *   "persist_exact_regen_max_walkers"  131072   exact mode: largest ensemble of the device-wide persistent form when its plans go up as generator states
 *   "mt_tok_wshift" / "mt_tok_tail"  11 / 2048   the device tokenizer's window rule    "mt_device_lookahead"  batches ahead
 *   -- exchanges --
 *   "direct_timeout_ms"        bound of the device-side barriers of the direct and replay exchanges (the first of an emx_run: 6x)
 *   "replay_two_pass"          0         1: the replay exchange's own pass and replay pass as separate launches
 *   -- tests / instrumented builds only --
 *   "persist_test_skew", "test_fetch_delay_us"   make a barrier unmeetable / a fetch late (tests of the give-up and wait paths)
 *   "phase_clock"              0         instrumented builds (-DEMX_OPT_STAMPS=1): phase timestamps of every launch
 *   "ablate"                   0         experiments flavour only (EMX_BUILD_FLAVOUR=exp, -DEMX_EXPERIMENTS=1): skip-phase masks; refused otherwise
 *
 * After a barrier timeout (status bit 3) the context refuses further sharded half-steps until the peers are attached again
 * (emx_direct_export / _import or _attach on every rank). */
int emx_set_tuning(emx_ctx* ctx, const char* key, int64_t value);

/* ---- state: State(coords, log_prob) (state.py:10-45) ---------------------------------- */
int emx_set_state(emx_ctx* ctx, const double* coords, const double* log_prob /* or NULL */);
int emx_get_state(emx_ctx* ctx, double* coords /* or NULL */, double* log_prob /* or NULL */);
int emx_get_accepted(emx_ctx* ctx, uint8_t* mask /* N */); /* `accepted` of the last propose */
/* Snapshots (slots 0..7): run_mcmc returns a State and accepts it back (ensemble.py:441-447, 312) -- a State that is the
 * device state needs no PCIe round trip.  The host layer hands out a lazy State; when a later call is about to change the
 * ensemble while that object is still alive, emx_snapshot_save keeps its values in HBM (device-to-device copy on the context
 * stream); emx_snapshot_read materialises them on demand, emx_snapshot_restore makes a snapshot the current state again. */
int emx_snapshot_save(emx_ctx* ctx, int32_t slot);
int emx_snapshot_read(emx_ctx* ctx, int32_t slot, double* coords /* or NULL */, double* log_prob /* or NULL */);
int emx_snapshot_restore(emx_ctx* ctx, int32_t slot);
int emx_snapshot_free(emx_ctx* ctx, int32_t slot);

/* The caller's own batched log-prob, on device memory: the reference's vectorize=True contract -- ONE call of log_prob_fn on the
 * (Ns, ndim) block of a split's proposals (ensemble.py:486-487, called at red_blue.py:93) -- without the block or the result
 * leaving HBM.  The function is called on the host thread that drives the step; it must ENQUEUE work on `hip_stream` (a kernel
 * launch, a library call) that reads the n rows of coords_dev (row-major, ndim doubles each, in the order the reference passes
 * them: ascending walker index within the split) and writes n log-probabilities to log_prob_dev, and return 0 (non-zero aborts
 * the step with an error).  It must not synchronise.  -inf is a legal value, NaN raises the reference's error.  Replaces the
 * closed-form target: emx_run, emx_halfstep, emx_eval_log_prob and the all-gather / log-prob / replay exchanges work over it. */
typedef int (*emx_device_log_prob_fn)(void* user, const double* coords_dev, int64_t n, int32_t ndim, double* log_prob_dev,
                                      void* hip_stream);
int emx_set_target_callback(emx_ctx* ctx, emx_device_log_prob_fn fn, void* user);

/* ---- target: the batched log-prob (ensemble.py:458-553, vectorised) -------------------- */
/* p0/p1: DIAG (mu, ivar); DENSE (mu, icov[D*D], symmetric positive definite: factored once as
 * L L^T, the kernel evaluates -0.5 |L^T (x-mu)|^2 with f64 MFMAs: fused into the half-step kernel up to
 * ndim 112, as a log-prob kernel of its own between propose and commit up to ndim 2048); others NULL.
 * scale: Rosenbrock divisor. */
int emx_set_target(emx_ctx* ctx, int32_t kind, const double* p0, const double* p1, double scale);
/* log-prob of the current state, stored as the state's log_prob (ensemble.py:350-351) */
int emx_eval_state_log_prob(emx_ctx* ctx);
/* EnsembleSampler.compute_log_prob(coords) for n host rows (n <= nwalkers per call) */
int emx_eval_log_prob(emx_ctx* ctx, const double* coords, int64_t n, double* out);

/* ---- moves & RNG ------------------------------------------------------------------------ */
/* ensemble.py:115-129: move list + normalised cumulative weights (cdf[nmoves-1] == 1) */
int emx_set_moves(emx_ctx* ctx, int32_t nmoves, const emx_move_desc* moves, const double* cdf);
int emx_set_rng_mode(emx_ctx* ctx, int32_t mode);
/* EMX_MOVE_GAUSS: per-coordinate standard deviations (n == ndim; NULL / 0 restores the isotropic sigma) */
int emx_set_move_scale(emx_ctx* ctx, int32_t move_index, const double* std, int32_t n);
/* current descriptor of a move (the sequential Gaussian cursor lives in it) */
int emx_get_move(emx_ctx* ctx, int32_t move_index, emx_move_desc* out);
/* numpy RandomState.get_state()/set_state() tuple round trip (ensemble.py:216-238) */
int emx_rng_set_mt19937(emx_ctx* ctx, const uint32_t key[624], int32_t pos, int32_t has_gauss, double cached);
int emx_rng_get_mt19937(emx_ctx* ctx, uint32_t key[624], int32_t* pos, int32_t* has_gauss, double* cached);
int emx_rng_set_philox(emx_ctx* ctx, uint64_t seed, uint64_t step);
int emx_rng_get_philox(emx_ctx* ctx, uint64_t* seed, uint64_t* step);

/* ---- the hot loop (ensemble.py:403-424): nsteps stored steps, nsteps*thin_by proposals -- */
int emx_chain_config(emx_ctx* ctx, int64_t capacity_steps); /* Backend.grow (backend.py:164-185) */
int emx_chain_reset(emx_ctx* ctx);                            /* Backend.reset (backend.py:19-35) */
int emx_run(emx_ctx* ctx, int64_t nsteps, int32_t thin_by, int32_t store);
int emx_iteration(emx_ctx* ctx, int64_t* stored_steps, int64_t* proposals);
/* With tuning key "graph" = 1, emx_run replays the native 16-step block (plan kernel + 16 x nsplits
 * half-steps, per-replay state in device memory) as ONE hipGraph launch when it can (Philox, one move,
 * thin_by 1, one rank).  Off by default: measured 5 % slower than back-to-back launches on MI355X unless
 * the host is the bottleneck (a busy or slow host thread); results are bit-identical either way.  captured: bit0 no-store graph, bit1 store graph. */
int emx_graph_state(emx_ctx* ctx, int32_t* disabled, int32_t* captured);
/* Backend.get_value slices (backend.py:42-58): steps start, start+stride, ... < stop.
 * what: 0 chain -> out[(nsel, N, D)], 1 log_prob -> out[(nsel, N)]. */
int emx_chain_read(emx_ctx* ctx, int32_t what, int64_t start, int64_t stop, int64_t stride, double* out);
int emx_accepted_counts(emx_ctx* ctx, double* out /* N, backend.accepted */);

/* ---- split-phase stepping: Move.propose pieces for host log-probs and sharded runs ------ */
/* Begin a step: choose the move (ensemble.py:406), build the split plan (red_blue.py:76-80 and
 * every draw of the step).  move_out/nsplits_out report the choice. */
int emx_step_begin(emx_ctx* ctx, int32_t store_this_step, int32_t* move_out, int32_t* nsplits_out);
/* same, for a move the caller already chose (Move.propose called directly: no choice draw) */
int emx_step_begin_with(emx_ctx* ctx, int32_t store_this_step, int32_t move_index, int32_t* nsplits_out);
/* fused half-step on the device target (red_blue.py:81-104 for one split) */
int emx_halfstep(emx_ctx* ctx, int32_t split);
/* host-target variant: proposals q (ns, D) in ascending-walker order (red_blue.py:90) ... */
int emx_propose(emx_ctx* ctx, int32_t split, double* q_out, double* factors_out /* or NULL */, int64_t* ns_out);
/* ... and Metropolis accept + commit given their log-probs (red_blue.py:96-104) */
int emx_accept(emx_ctx* ctx, int32_t split, const double* new_log_prob);
/* user-defined RedBlueMove.get_proposal: q (ns, D) and factors (ns) come from the caller */
int emx_accept_proposals(emx_ctx* ctx, int32_t split, const double* q, const double* factors,
                         const double* new_log_prob);
int emx_step_end(emx_ctx* ctx);
/* INPUTS mode / tests: set or read back the plan of the step begun (arrays of length N in
 * plan order: split 0's members ascending, then split 1's, ...; off has nsplits+1 entries) */
int emx_plan_set(emx_ctx* ctx, int32_t move_index, const int32_t* off, const int32_t* order, const int32_t* p0,
                 const int32_t* p1, const int32_t* p2, const double* s0, const double* uacc);
int emx_plan_get(emx_ctx* ctx, int32_t* off, int32_t* order, int32_t* p0, int32_t* p1, int32_t* p2, double* s0,
                 double* uacc);
/* INPUTS mode, EMX_MOVE_GAUSS: after emx_plan_set (off = {0, N}, order = walker of each slot, p0 = coordinate
 * that moves or -1, uacc), the (N, D) standard normals rng.randn(N, D) and the step-size factor (1 if unused);
 * the library forms (factor * scale_d) * n on the device (gaussian.py:87) */
int emx_plan_set_noise(emx_ctx* ctx, const double* normals, double factor);

/* ---- walker-sharded multi-GPU (one process per GPU; collectives stay in the host layer) -- */
int emx_set_shard(emx_ctx* ctx, int32_t rank, int32_t world);
/* use caller-owned device buffers (e.g. torch tensors handed to RCCL) for the exchange:
 * sendbuf (rows_per_rank, D+2), gathered (world*rows_per_rank, D+2); records = [row | log_prob | accepted] */
int emx_set_shard_buffers(emx_ctx* ctx, void* sendbuf, void* gathered, int64_t rows_per_rank);
/* raw device pointers for zero-copy wrapping (torch.distributed all-gather buffers):
 * which: 0 coords (N,D), 1 log_prob (N), 2 sendbuf (rows/rank, D+2), 3 gathered (world*rows/rank, D+2),
 *        4 chain (stored, N, D), 5 chain log_prob (stored, N), 6 Gaussian-move displacements (N, D),
 *        8 direct-exchange barrier flags (one uint64 per rank) */
int emx_device_ptr(emx_ctx* ctx, int32_t which, void** ptr, int64_t* nbytes);
int emx_shard_slots(emx_ctx* ctx, int32_t split, int64_t* t_lo, int64_t* t_hi, int64_t* ns);
/* after the all-gather of `sendbuf`s into `gathered`: write the other ranks' rows into X */
int emx_scatter_gathered(emx_ctx* ctx, int32_t split);

/* ---- pull exchange: walker-block ownership, only the partner rows that are read travel -----
 * The all-gather above replicates every updated row on every rank: (G-1)/G of the ensemble crosses
 * xGMI per step.  A half-step reads ONE partner row per updated walker (stretch.py:32; two / three for
 * DE / snooker), so with rank r owning walkers [N r / G, N (r+1) / G) it is enough to move exactly those
 * rows.  The RNG plan is replicated, hence every rank knows which of its rows the others will read:
 *   emx_pull_prepare(split)  -> records [row index | row] for every peer in the send buffer
 *                               (world blocks of *records_per_peer records of D+1 doubles)
 *   all-to-all, records_per_peer * (D+1) doubles per pair (host layer, or emx_run when emx_comm_init ran)
 *   emx_pull_apply(split)    -> received rows into the local replica, then the half-step over the
 *                               slots whose walker this rank owns
 * Only a rank's own block of X / log_prob / accepted / chain is current until emx_replica_pack ->
 * all-gather (bmax records of D+3 doubles per rank) -> emx_replica_unpack re-synchronises the replicas
 * (emx_run does it before it returns).  Results are bit-identical to the single-rank run. */
#define EMX_EXCHANGE_ALLGATHER 0
#define EMX_EXCHANGE_PULL 1
#define EMX_EXCHANGE_DIRECT 2      /* see "direct exchange" below */
#define EMX_EXCHANGE_LOGPROB 3     /* see "log-prob exchange" below */
#define EMX_EXCHANGE_REPLAY 4      /* see "replay exchange" below */
int emx_set_exchange(emx_ctx* ctx, int32_t kind);          /* before emx_set_shard / emx_comm_init */
/* doubles the send / receive buffers must hold for the moves installed (pull exchange) */
int emx_exchange_layout(emx_ctx* ctx, int64_t* send_doubles, int64_t* recv_doubles);
/* caller-owned exchange buffers (pull exchange), e.g. torch tensors handed to RCCL */
int emx_set_exchange_buffers(emx_ctx* ctx, void* send, int64_t send_doubles, void* recv, int64_t recv_doubles);
int emx_own_walkers(emx_ctx* ctx, int64_t* lo, int64_t* hi);
int emx_pull_prepare(emx_ctx* ctx, int32_t split, int64_t* records_per_peer);
int emx_pull_apply(emx_ctx* ctx, int32_t split);
int emx_replica_pack(emx_ctx* ctx, int64_t* records_per_rank);
int emx_replica_unpack(emx_ctx* ctx);

/* ---- direct exchange: partner rows read in place from the owner's HBM over xGMI ----------------------------------
 * Same walker-block ownership as the pull exchange, but nothing is packed, sent or scattered: every rank maps the other
 * ranks' coordinate arrays (hipIpc handles between processes, plain pointers between contexts of one process) and the
 * half-step kernel loads a partner row from the replica of the rank that owns it (stretch.py:32 reads ONE row per updated
 * walker, de.py:53 two, de_snooker.py:41-46 three) -- the only bytes that cross xGMI are the (G-1)/G of those rows that
 * live on another GPU.  Between half-steps a one-wave device-side barrier (a flag store into every peer's flag array, a
 * spin on the own array; bounded by tuning key "direct_timeout_ms", status bit 3 on expiry) orders the two hazards of
 * red_blue.py:85,104: a partner row must carry its owner's last commit, and nobody may start committing split k+1 while a
 * peer still reads those rows as partners of split k.
 *   emx_set_exchange(EMX_EXCHANGE_DIRECT); emx_set_shard / emx_comm_init; then
 *   multi-process: emx_direct_export -> 128 bytes per rank, all-gathered by the host layer -> emx_direct_import
 *   one process:   emx_direct_attach(coordinate arrays, flag arrays) of all ranks (emx_device_ptr which = 0 / 8)
 *   per step:      emx_step_begin; emx_direct_halfstep(split, barrier) for every split; emx_step_end   (emx_run does it)
 * Only a rank's own block is current until the replica re-synchronisation (emx_replica_pack / all-gather / unpack; emx_run
 * runs it before it returns when emx_comm_init was called).  Results are bit-identical to the single-rank run.  World size
 * <= 8 (one node). */
int emx_direct_export(emx_ctx* ctx, uint8_t handles[128]);
int emx_direct_import(emx_ctx* ctx, const uint8_t* handles /* world * 128 bytes, rank order */);
int emx_direct_attach(emx_ctx* ctx, void* const* peer_coords /* [world] */, void* const* peer_flags /* [world] or NULL */);
/* barrier != 0: device-side barrier with the peers first (needs their flag arrays); 0: the caller orders the ranks itself */
int emx_direct_halfstep(emx_ctx* ctx, int32_t split, int32_t barrier);

/* ---- log-prob exchange: the reference's own parallel model (ensemble.py:486-496: pool.map over the proposals) ------------
 * Proposal, decision and commit are replicated -- every rank holds the whole ensemble and the same plan, so every rank computes
 * the same proposals -- and only the log-probability evaluations are shared out: rank r evaluates the proposals
 * [r * per, (r + 1) * per) of the split, per = ceil(ns / world).  What travels is 8 bytes per walker-update (an in-place
 * all-gather of `per` doubles per rank on the buffer emx_device_ptr(which = 3) returns), never a coordinate; the replicas stay
 * identical, so there is nothing to re-synchronise.  The protocol for targets whose evaluation dominates the step (wide dense
 * Gaussians here); for the cheap closed-form targets of the BASELINE configurations the replicated part is most of the step.
 *   emx_set_exchange(EMX_EXCHANGE_LOGPROB); emx_set_shard / emx_comm_init; per step:
 *   emx_step_begin; for every split: emx_logprob_begin(split, &per) -> all-gather(in place, per doubles per rank)
 *   -> emx_logprob_finish(split); emx_step_end   (emx_run does it when emx_comm_init ran).
 * Results are bit-identical to the single-rank run. */
int emx_logprob_begin(emx_ctx* ctx, int32_t split, int64_t* per_rank);
int emx_logprob_finish(emx_ctx* ctx, int32_t split);

/* ---- replay exchange: the DECISIONS travel (8 bytes per walker-update), every replica recomputes the accepted updates ------
 * Full replicas, slot-range ownership as in the all-gather exchange -- but where that one ships every updated row to every rank
 * ((G-1) * 8 (D+2) bytes per walker-update over xGMI), this one ships what the owner decided: the new log-prob of an accepted
 * proposal, NaN for a rejected one.  A proposal (stretch.py:33, de.py:53-62, de_snooker.py:41-46, gaussian.py:87) is a function
 * of rows every replica holds identically before the half-step and of the replicated plan, so after the all-gather of the
 * decisions every rank replays the accepted updates of the others on its own replica and obtains the owner's bits; the
 * log-probability is evaluated once, by the owner.  Extra HBM work per rank: the accepted fraction of the other ranks' slots
 * (24 D + 8 bytes each, no target evaluation) -- which is what makes it the protocol for BOTH regimes: cheap targets (nothing
 * crosses xGMI but 8 (G-1) bytes per update) and expensive ones (the evaluation is shared out like the log-prob exchange's, and
 * unlike there proposal and commit are shared out too).  Replicas stay identical: stored chains are complete on every rank.
 *   emx_set_exchange(EMX_EXCHANGE_REPLAY); emx_set_shard / emx_comm_init; per step:
 *   emx_step_begin; for every split: emx_replay_begin(split, &rows) -> all-gather of `rows` doubles per rank, send buffer
 *   emx_device_ptr(which = 2) into emx_device_ptr(which = 3) -> emx_replay_finish(split); emx_step_end
 *   (emx_run does it when emx_comm_init ran).  Results are bit-identical to the single-rank run. */
int emx_replay_begin(emx_ctx* ctx, int32_t split, int64_t* rows_per_rank);
int emx_replay_finish(emx_ctx* ctx, int32_t split);
/* The same exchange without a collective library (one node): after emx_direct_export / emx_direct_import (or emx_direct_attach)
 * -- which under this exchange map every rank's RECEIVE buffers (emx_device_ptr which = 3: two of them, used alternately) and
 * barrier flags -- emx_replay_exchange(split) replaces the all-gather: a kernel stores the decisions into every peer's buffer
 * over xGMI (8 bytes per own walker-update and peer) and the one-wave device-side barrier of the direct exchange tells every
 * rank that all of them have landed.  No host round trip, no collective launch latency; emx_run uses it when the peers are
 * mapped.  Between emx_replay_begin and emx_replay_finish. */
int emx_replay_exchange(emx_ctx* ctx, int32_t split);

/* RCCL driven by the library itself (ncclAllGather enqueued on the context stream between the
 * half-step kernels, so that emx_run covers sharded runs with no host round trip per step).
 * librccl is resolved with dlopen (path, $EMX_RCCL_LIB, librccl.so.1): pass PyTorch's copy when
 * torch is loaded so that the process holds ONE RCCL.  Rank 0 creates the id, the host layer
 * broadcasts its 128 bytes, every rank calls emx_comm_init. */
int emx_comm_load(const char* librccl_path /* or NULL */);
int emx_comm_get_unique_id(uint8_t id[128]);
int emx_comm_init(emx_ctx* ctx, int32_t rank, int32_t world, const uint8_t id[128]);
int emx_comm_destroy(emx_ctx* ctx);
/* ranks of the communicator emx_comm_init created, as RCCL itself counts them (ncclCommCount) -- what a bench line may claim
 * as its number of GPUs; 0 when no communicator exists */
int emx_comm_count(emx_ctx* ctx, int32_t* ranks_out);

/* ---- around the hot loop: autocorrelation time and the initial-state check ------------------------------------------ */
/* Integrated autocorrelation time of the device-resident chain, per parameter (autocorr.py:49-123 integrated_time applied to
 * Backend.get_value("chain", discard, thin), backend.py:42-58,130-150): normalised FFT autocorrelation function of every
 * walker's series (batched hipFFT next to the chain), averaged over walkers, Sokal window with step size c.  tau_out[ndim] is
 * in units of the SELECTED samples (the caller multiplies by thin, backend.py:150); window_out[ndim] (or NULL) the windows;
 * *nsamples_out the series length, against which the caller applies the reference's "tol" check (autocorr.py:110-121).
 * libhipfft is resolved with dlopen (emx_fft_load(path), $EMX_HIPFFT_LIB, libhipfft.so): pass PyTorch's copy when torch is
 * in the process. */
int emx_fft_load(const char* libhipfft_path /* or NULL */);
int emx_autocorr(emx_ctx* ctx, int64_t discard, int64_t thin, double c, double* tau_out, int32_t* window_out,
                 int64_t* nsamples_out);
/* walkers_independent(coords) (ensemble.py:653-663): centre, scale by max |.| and by the 2-norm per coordinate, condition
 * number <= 1e8.  The (n, ndim) host matrix goes to `device`; Householder QR there (one reflector per coordinate), then the
 * extreme singular values of the ndim x ndim triangular factor by (inverse) power iteration on the host.  *independent: 0 / 1;
 * *cond_out (or NULL): the condition number (inf for a constant coordinate, non-finite input, n < ndim or a singular factor). */
int emx_walkers_independent(int32_t device, const double* coords, int64_t n, int32_t ndim, int32_t* independent,
                            double* cond_out);
/* The same check on the state a context holds (a run continued from the State the previous run returned: the reference re-checks
 * every sample() call, ensemble.py:316-323) -- the ensemble is read where it is, nothing crosses PCIe.  A non-finite coordinate
 * shows as a non-finite entry of the factor (verdict 0). */
int emx_walkers_independent_resident(emx_ctx* ctx, int32_t* independent, double* cond_out);

/* ---- measurement ------------------------------------------------------------------------ */
int emx_timer_start(emx_ctx* ctx);                 /* hipEventRecord on the context stream */
int emx_timer_stop(emx_ctx* ctx, float* ms);       /* record + synchronize + elapsed       */
/* per-launch hipEvent timing of the half-step kernel: enable, run, then read the durations */
int emx_profile_enable(emx_ctx* ctx, int32_t max_launches);
int emx_profile_read(emx_ctx* ctx, float* ms_out, int32_t* n_inout);

/* exact (MT19937) mode: microseconds per produced step of the host plan pipeline's stages while it is alive -- out[0] wall
 * clock, [1] generator (twist + temper), [2] tokenizer (the serial walk of the stream: rejection tests), [3] finishers (summed
 * over the threads), [4] tokenizer waiting for words, [5] tokenizer waiting for a free staging buffer (i.e. for the consumer) */
int emx_pipeline_stats(emx_ctx* ctx, double out[6], int64_t* steps_produced, int32_t* finisher_threads);
/* how the host pipeline handed its stretch steps over so far (EMX_RNG_MT19937; moves/stretch.py:30-32, moves/red_blue.py:100 -- the
 * fixed-length draws of a step): raw_steps -- the draws as the generator words they are, in the plan's columns, finished on the
 * device (k_plan_raw; tuning "mt_device_finish"); of those, regen_steps -- not even the words: `order` and the generator's STATE at
 * every eighth block of the draws' region of the stream, from which the device makes the words again (k_plan_regen, round 6:
 * ensembles of "mt_regen_min_walkers" = 16 384 and more whose half is a power of two; 0: never).  Same plans bit for bit. */
int emx_pipeline_handovers(emx_ctx* ctx, int64_t* raw_steps, int64_t* regen_steps);

/* Exact (MT19937) mode with the plans made ON THE DEVICE (csrc/emx_mtdev.hpp): one StretchMove, one replica, ensembles of
 * 147 456 walkers or more (round 5; 131 072 before) -- where the serial host stages of "same seed => same chain as the reference" (ensemble.py:166-167,406,
 * moves/red_blue.py:76-80,100, moves/stretch.py:30-32) cost more than the kernels do.  MT19937 segments by jump-ahead, the rejection
 * tests of random.shuffle / randint and the Fisher-Yates swaps all run in kernels; no host thread touches a draw.  Tuning key
 * "mt_device": 0 = always the host pipeline, 1 (default) = from "mt_device_min_walkers" (147 456) on, 2 = from 8 192 walkers on
 * (measured, MI355X, 64-dim dense Gaussian: 65 536 walkers 94 us/step against the host pipeline's 69; 262 144 x 32: 192 against 316;
 * 1 048 576: 602 against 1 347 -- profiles/r04/mtdev_sizes.txt).
 *   out[0] 1 when the current configuration takes this producer, [1] 1 while one is alive, [2] steps taken from producers so far,
 *   [3] generation rounds, [4] stream segments (of 128 MT blocks), [5] batches of 16 steps produced, [6] tokenizer windows,
 *   [7] microseconds spent computing the jump polynomials (once per process)  -- [3..7] of the live or the last producer */
int emx_mtdev_info(emx_ctx* ctx, int64_t out[8]);
/* tests: raw pieces of the live producer after a synchronise.  what = 0: `n` tempered stream words from absolute position `arg`
 * (uint32 out); 1: the accepted Fisher-Yates targets J[i], i < nwalkers, of producer step `arg` (uint32 out, entry 0 unused);
 * 2: that step's positions -- nsplits * 3 (z words, randint words, accept words) then the position after the step (uint64 out) */
int emx_mtdev_debug(emx_ctx* ctx, int32_t what, int64_t arg, void* out, int64_t n);
/* the device tokenizer's work so far: out = windows decided | fixed-point rounds | 64-word groups of the one-wave tail | its ballot
 * rounds | then 10 ns ticks: waiting for stream windows | deciding the wide windows | the tail | the whole tokenizer kernels
 * (red_blue.py:80's masked rejection is the only serial part of a step: this is what it cost) */
int emx_mtdev_tok_stats(emx_ctx* ctx, int64_t out[8]);
/* host only: the MT19937 state key `k * stride_words` words after the block FOLLOWING `key` (k >= 1), by the jump polynomial
 * t^(k stride) mod phi applied to the 33-block window after `key` -- the host statement of what k_mt_jump computes */
int emx_host_mt_jump(const uint32_t key[624], uint64_t stride_words, int32_t k, uint32_t out_key[624]);

/* Persistent half-steps.  emx_run takes the headline shape -- stretch-move steps (red_blue.py:55-106 with stretch.py:27-34) and
 * DE-move steps (de.py:40-64) of two splits, snooker steps (de_snooker.py:31-46) of four, the
 * fused dense Gaussian target at an even ndim up to 64, Philox plans, one replica, nwalkers a multiple of 32 from 512 (tuning
 * "persist_min_walkers") to 256 x the CU count, i.e. one 16-walker tile per wave of a co-resident grid of about one workgroup
 * per CU -- up to 32 half-steps per kernel launch (in a mixture: the consecutive steps of one move -- the steps of a DEMove and a
 * DESnookerMove share launches, k_persist_mix, tuning "persist_mix" = 0: never; a launch goes on into the next batch of sixteen
 * steps of Philox plans, "persist_span" = 0: it ends with its batch; steps with another number of
 * splits take the per-half-step launches): a device-wide barrier stands where the kernel
 * boundaries were, and the next half-step's plan entries and own rows are loaded while this one computes.  Same draws, same
 * arithmetic, same bits as the launch-per-half-step path.  Tuning "persist" = 0 turns it off; "persist_timeout_ms" bounds a
 * barrier wait (default 2000: a grid that cannot become co-resident -- another process holding the device's CUs -- raises
 * status bit 3 instead of hanging).  With emx_profile_enable the events bracket whole launches.
 * A context whose only move is the Gaussian Metropolis move (gaussian.py:76-101 / mh.py:57-77; same target, RNG and replica
 * conditions, nwalkers a multiple of 16) runs up to 16 steps per launch with every walker in registers and no barrier at all
 * (k_persist_gauss).
 *   out[0] 1 when the current configuration qualifies, out[1] persistent launches so far, out[2] half-steps they ran,
 *   out[3] reserved (0) */
int emx_persist_info(emx_ctx* ctx, int64_t out[4]);
/* how many of those launches took the one-XCD form: ensembles of up to 8 192 walkers (stretch move, two splits) run an eight times
 * larger grid of which every eighth workgroup works -- all on one XCD, whose L2 keeps the walker state coherent with plain accesses and
 * a barrier of that XCD's own instead of agent-scope accesses and the device-wide barrier (tuning "persist_local" = 0: never;
 * "persist_local_max_walkers").  The element-wise targets (EMX_TARGET_ISO_GAUSS / _DIAG_GAUSS / _ROSENBROCK / _BOX, rows of 4 or 8 lanes:
 * ndim <= 64 even, <= 32 odd) have this form only (csrc/emx_pvalu.hip; tuning "persist_valu" = 0: never); same bits (red_blue.py:85,104: a half-step still sees every update of the one before).
 * EMX_RNG_MT19937 (the reference's own stream; ensemble.py:166-167) takes the one-XCD forms too -- and the device-wide forms of both
 * kernels up to 32 768 walkers ("persist_exact_max_walkers") -- when the context has ONE move: the
 * host pipeline's plans of up to sixteen steps ("persist_exact_steps") are fetched from their pinned staging buffers by one kernel
 * per launch (k_plan_fetch; its workgroups leave the XCD a one-XCD launch lives on alone, tuning "fetch_avoid" = 0: they do not; beside a
 * device-wide launch it runs as "fetch_blocks" = 64 workgroups, 0: one per 256 entries)
 * -- tuning "persist_exact" = 0: the per-half-step launches with an upload per step.  Move mixtures too
 * (round 5: the next step's move is read off the pipeline's plan before it is taken; "persist_exact_mix" = 0: one move only).  A
 * launch of this mode that cannot become resident is redone like a Philox one (round 5): the pipeline keeps the generator state
 * behind each of its last 64 steps and is taken back to the one in front of the launch.  Status bit 3 stays -- the run is void,
 * as for a barrier that timed out in the middle of a launch -- when that is not possible: more than 60 steps of launches have
 * been enqueued since the launch that gave up and are still unsettled, another pipeline has been started since, or a step begun
 * with emx_step_begin is open when the failure is noticed. */
int emx_persist_local_launches(emx_ctx* ctx, int64_t* n);
/* host only: the grid the persistent kernel takes for `nwalkers` walkers updated in `nsplits` half-steps on a device of `num_cu`
 * CUs -- waves per workgroup (8 / 4 / 2 / 1; 0: no persistent grid, the per-half-step launches run) and workgroups */
int emx_host_persist_shape(int64_t nwalkers, int32_t nsplits, int32_t num_cu, int32_t* waves_per_group, int32_t* groups);

/* ---- host-only helpers (no GPU needed; used by the CPU test-suite) ---------------------- */
typedef struct emx_mt emx_mt;
emx_mt* emx_mt_create(const uint32_t key[624], int32_t pos, int32_t has_gauss, double cached);
void emx_mt_destroy(emx_mt* m);
void emx_mt_get_state(const emx_mt* m, uint32_t key[624], int32_t* pos, int32_t* has_gauss, double* cached);
void emx_mt_random_sample(emx_mt* m, int64_t n, double* out);
void emx_mt_randint(emx_mt* m, uint64_t bound, int64_t n, int64_t* out);
void emx_mt_randn(emx_mt* m, int64_t n, double* out);
void emx_mt_shuffle_labels(emx_mt* m, int64_t n, int32_t nsplits, int32_t* labels);
int32_t emx_mt_choice_cdf(emx_mt* m, const double* cdf, int32_t n);
/* one step's exact plan on the host (the producer emx_run uses in MT19937 mode) */
int emx_host_plan_mt(emx_mt* m, int64_t nwalkers, int32_t ndim, const emx_move_desc* mv, int32_t* off, int32_t* order,
                     int32_t* p0, int32_t* p1, int32_t* p2, double* s0, double* uacc);
/* The same plans for `nsteps` consecutive steps (move choice included, ensemble.py:406) made by the threaded pipeline emx_run
 * uses in MT19937 mode (csrc/emx_mtpipe.hpp: generator / tokenizer / `nworkers` finisher threads, `nsinks` staging buffers used
 * round-robin).  Output arrays hold nsteps * nwalkers entries (step-major; any may be NULL), moves_out nsteps.  Advances `m`
 * exactly like nsteps calls of emx_mt_choice_cdf + emx_host_plan_mt.  Returns the number of finisher threads used (> 0) or a
 * negative code; *seconds_out: wall time of the whole production. */
int emx_host_plan_mt_stream(emx_mt* m, int64_t nwalkers, int32_t ndim, int32_t nmoves, const emx_move_desc* moves,
                            const double* cdf, int64_t nsteps, int32_t nworkers, int32_t nsinks, int32_t* moves_out,
                            int32_t* order, int32_t* p0, int32_t* p1, int32_t* p2, double* s0, double* uacc,
                            double* seconds_out);
/* the draws of ONE RedBlueMove.get_proposal(s, c, random) call for `split` of a given partition */
int emx_host_split_draws(emx_mt* m, int64_t nwalkers, const emx_move_desc* mv, const int32_t* off,
                         const int32_t* order, int32_t split, int32_t* p0, int32_t* p1, int32_t* p2, double* s0);
/* one step's native plan on the host (the function the kernels evaluate in flight) */
int emx_host_plan_philox(uint64_t seed, uint64_t step, int64_t nwalkers, const emx_move_desc* mv, int32_t* off,
                         int32_t* order, int32_t* p0, int32_t* p1, int32_t* p2, double* s0, double* uacc);
int32_t emx_host_move_choice_philox(uint64_t seed, uint64_t step, const double* cdf, int32_t n);
/* pull exchange: records per (source, destination) pair of one half-step (what emx_pull_prepare returns) */
int64_t emx_host_pull_capacity(int64_t nwalkers, int32_t world, int32_t nsplits, int32_t partners_per_walker);

#ifdef __cplusplus
}
#endif
#endif /* EMX_H */
