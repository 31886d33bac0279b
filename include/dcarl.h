/* dcarl.h — C-ABI of libdcarl_hip.so, the MI355X (gfx950) implementation of DCARL's confidence hot path.
 *
 * The reference (zhcao92/DCARL) has NO FFI/plugin layer for this path: it is pure Python/NumPy
 * (SURVEY.md §8b).  The drop-in boundary is therefore the reference's Python entry-point surface
 * (kept verbatim by Simulation_testing/ ... /test_DCARL.py and data_sampling.py in this repo), and this
 * C-ABI is what those entry points bind through ctypes.  Each entry point below cites the reference
 * lines whose work it replaces.  Abbreviations (relative to the reference root):
 *   S1 = Simulation_testing/Simulation_1/test_DCARL.py
 *   S2 = Simulation_testing/Simulation_2/test_DCARL.py
 *   DS = Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked [host]; the caller owns every buffer; the library allocates no device or
 *     host memory of its own and no RESULT depends on an earlier call.  What it does keep between calls, all of it diagnostic:
 *       * one device word (static, not allocated): the fault word of the multi-wave online kernel, set by a hand-over that never
 *         arrives, read and cleared by dcarl_trace_status();
 *       * per thread, the text of the last error (dcarl_last_error) and the name of the last kernel launched (dcarl_last_kernel);
 *       * per process, a 64-entry memo {workspace address -> N, S, A, flags, element width} of the latest dcarl_ingest_group_* calls,
 *         consulted by dcarl_ingest_pack_* only to REFUSE a pack call whose arguments differ from its group call;
 *       * per process, whether each kernel's dynamic-LDS limit has been raised (hipFuncSetAttribute, once per kernel), and the
 *         dlopen handle of librccl.so once dcarl_comm_* has been used;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); launches are
 *     asynchronous on it; nothing synchronises;
 *   - every function returns 0 (DCARL_OK) or a negative DCARL_E* code and never throws;
 *     dcarl_last_error() returns a thread-local message for the last failure on the calling thread;
 *   - output pointers documented "(nullable)" may be NULL to skip that output.
 *
 * Device record layout for the online ("trace") path — "sliced time-major, quad-packed":
 *   states are processed 64 at a time (one wavefront = one slice; slice w holds states 64w..64w+63).
 *   Slice w owns rows[w] time rows, rows[w] a multiple of 4 and >= the longest record stream in the
 *   slice; slice_row_off[w] = rows[0]+..+rows[w-1] (int64, W+1 entries, W = ceil(S/64)).
 *   Record t (0-based arrival order WITHIN its state) of state s lives at element
 *       e(s,t) = (slice_row_off[s/64] + (t & ~3)) * 64 + (s % 64) * 4 + (t & 3)
 *   so that one lane reads four consecutive records of its state with a single 16-byte load and a
 *   wavefront reads 1 KiB contiguous.  len[s] = number of records of state s; padding is never read
 *   as data.  Inputs R[e] (f32 or f64 cumulative reward) and act[e] (u8 action id) and outputs
 *   step_val[e], step_act[e] share this indexing.
 *   "State s" in this layout is a SLOT: a table may place its states in any order (builders sort them by stream length so
 *   that the 64 streams of a slice end together) and pass the order as slot_state[k] = state in slot k (i32 [S], nullable =
 *   identity).  len[] and the record arrays are indexed by slot; everything handed back PER STATE (V_out, n_out, vmax, amax,
 *   act_step, the buckets of dcarl_count_records / dcarl_group_records) is indexed by state.
 */
#ifndef DCARL_H
#define DCARL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCARL_ABI_VERSION 8
#define DCARL_MAX_ACTIONS 32      /* S1:39 declares action_num = 30 */
#define DCARL_SLICE 64            /* states per slice = wavefront width on gfx950 */

enum {
    DCARL_OK = 0,
    DCARL_EINVAL = -1,    /* bad argument (null pointer, A out of range, misaligned buffer ...) */
    DCARL_EDEVICE = -2,   /* no gfx950 device / HIP runtime error before launch */
    DCARL_ELAUNCH = -3,   /* kernel launch failed */
    DCARL_ECOMM = -4      /* RCCL not loadable / collective failed (dcarl_comm_*) */
};

/* Literals the reference hard-codes (S1:10 default args, S1:43-52).  Plain-old-data, passed by pointer [host]. */
typedef struct dcarl_params {
    int32_t rule_act;    /* S1:43  trusted rule-based policy's action id (0) */
    int32_t n_thres;     /* S1:45  a bucket is evaluated only when its size > n_thres (10) */
    double alpha;        /* S1:10  0.05 */
    double scale;        /* S1:10  150 */
    double cap;          /* S1:12  100: upper bound is min(cap, .) */
    double init_rule;    /* S1:52  100: V[s][rule_act] before any evaluation */
    double init_other;   /* S1:51  -50: V[s][a != rule_act] before any evaluation */
} dcarl_params_t;

typedef struct dcarl_device_info {
    char arch[32];            /* "gfx950" */
    int32_t compute_units;    /* 256 on MI355X */
    int32_t wavefront;        /* 64 */
    int64_t hbm_bytes;
} dcarl_device_info_t;

/* ---- housekeeping ---------------------------------------------------------------------------- */
int32_t dcarl_version(void);
/* Content hash of the sources the library was built from (dcarl_amd/build.py source_id); the binding refuses a
 * library whose id differs from the sources next to it. */
const char* dcarl_build_id(void);
const char* dcarl_last_error(void);
/* Fills *out [host] for HIP device `dev`; DCARL_EDEVICE unless the device is gfx950. */
int32_t dcarl_device_info(int32_t dev, dcarl_device_info_t* out);
/* [host] fills the reference defaults listed above. */
void dcarl_default_params(dcarl_params_t* p);
/* Name (template instance) of the kernel the last dcarl_trace_* / dcarl_bounds_csr_* call on this thread launched;
 * "" before the first one.  Diagnostic only: bench.py reports it instead of re-deriving the dispatch. */
const char* dcarl_last_kernel(void);
/* Scratch sizing in one place: bytes of caller-provided workspace the call of that kind needs; every
 * other entry point needs none.  kind: DCARL_WS_SCAN (N = elements), DCARL_WS_RLS (N = visited rows, S = queries),
 * DCARL_WS_STATE_IDS (N = records, S = expected distinct states or 0), DCARL_WS_SUMMARY (S = states), DCARL_WS_INGEST_F32 / _F64 (N records over S states with
 * A actions, every option of dcarl_ingest_* on; dcarl_ingest_workspace_bytes gives the exact figure for one set of options).
 * Returns 0 for an unknown kind or negative sizes. */
enum { DCARL_WS_SCAN = 1, DCARL_WS_RLS = 2, DCARL_WS_STATE_IDS = 3, DCARL_WS_SUMMARY = 4, DCARL_WS_INGEST_F32 = 5,
       DCARL_WS_INGEST_F64 = 6 };
int64_t dcarl_workspace_bytes(int32_t kind, int64_t S, int32_t A, int64_t N);

/* ---- online confidence estimation + candidate arg-max ("trace" mode) ------------------------------
 * Replaces the hot loop S1:73-99 / S2:72-97 for ALL states at once: per record append to bucket (S1:80),
 * thresholded re-evaluation of V[s][a] by upper_bound (S1:10-12, rule action) or
 * min(lower_bound, CI_lower_bound) (S1:14-24, S1:90), per-state max / first arg-max (S1:93-95) and the
 * activation latch (S1:98-99).  One evaluation per record, float64 arithmetic on the stored inputs.
 *   R, act, slice_row_off, len : inputs in the sliced layout above;  S states, A <= 32 actions.
 *   slot_state (nullable) i32 [S]  the state whose stream sits in slot k of the layout (tables whose slots are the states
 *                                  sorted by stream length, so that a slice's 64 streams end together); NULL = identity.
 *                                  The per-STATE outputs below are written at row slot_state[k]: callers never see slots.
 *   step_val  (nullable) f32/f64 [rows*64]  max_a V[s][a] after each record          (S1:93; f64: the 5 tie-break
 *                                           code bits cleared, like V_out)
 *   step_act  (nullable) u8      [rows*64]  arg-max candidate after each record       (S1:94-95)
 *   act_step  (nullable) i32 [S]   1-based count of the state's records at first arg-max != rule_act, else -1
 *   V_out     (nullable) f64 [S*A] final table TSRL_value (exact to 2^-47 relative: the 5 low mantissa
 *                                  bits carry the tie-break code and are returned cleared)
 *   n_out     (nullable) i32 [S*A] final bucket sizes
 *   vmax/amax (nullable) f32/i32 [S] final max and arg-max per state
 * With step_val, step_act AND act_step all NULL nothing per record is asked for: the call then runs the loop's statistics stage
 * (S1:80) and ONE evaluation per bucket (S1:86-90 on the bucket as the last record left it) instead of an evaluation per record
 * (final_table_kernel, csrc/trace_final.hip) — the same V_out / n_out / vmax / amax, bit for bit, at the cost of reading the table.
 */
int32_t dcarl_trace_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                        const int32_t* slot_state, int32_t S, int32_t A, const dcarl_params_t* params, float* step_val, uint8_t* step_act,
                        int32_t* act_step, double* V_out, int32_t* n_out, float* vmax, int32_t* amax,
                        void* stream);
int32_t dcarl_trace_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                        const int32_t* slot_state, int32_t S, int32_t A, const dcarl_params_t* params, double* step_val, uint8_t* step_act,
                        int32_t* act_step, double* V_out, int32_t* n_out, float* vmax, int32_t* amax,
                        void* stream);

/* ---- continuation of the online loop (ABI version 6) -------------------------------------------------------------------
 * The reference's loop is incremental: data_state_act, TSRL_value and activation_step live across records (S1:41-59,73-99)
 * and Simulation_2 consumes data[0:20000] of a 49 866-row table (S2:72) — the rest can be fed later.  dcarl_trace_resume_*
 * is dcarl_trace_* on a caller-owned STATE that it reads at the start and advances in place: feeding a table in k chunks (cut
 * anywhere: mid-quad, mid-slice, empty chunks, states that appear only later) gives bit for bit the step traces, table,
 * arg-max and latch of one pass over the whole table.  The state is the loop's sufficient statistic, per STATE (row = state
 * id, whatever the slot order of each chunk's table):
 *   n        i32 [S*A]  bucket sizes                    len(data_state_act[s][a])                              (S1:80)
 *   sum      f64 [S*A]  sum of (x - K) over the bucket  } the bucket itself is not kept: upper_bound / lower_bound /
 *   sumsq    f64 [S*A]  sum of (x - K)^2                } CI_lower_bound (S1:10-24) are functions of (n, sum, sum of squares)
 *   shift    f64 [S]    K = the state's first reward (meaningful once the state has a record; shifted sums: DESIGN.md 3)
 *   V        f64 [S*A]  TSRL_value (S1:50-53,88-90), tie-break code bits cleared
 *   act_step i32 [S]    activation_step (S1:57,98-99): 1-based count of the state's records at the first arg-max != rule_act,
 *                       -1 = not yet.  Counts run over ALL chunks (the state's record count so far is the sum of its n).
 * All six pointers non-NULL.  fresh != 0: the contents are ignored and the loop starts from the priors (S1:41-59) — the
 * first chunk needs no separate initialisation.  step_val / step_act (nullable) are this chunk's traces in this chunk's
 * layout; vmax / amax (nullable) as in dcarl_trace.  In the layout's arrays a state's record t is its t-th record OF THIS
 * CHUNK. */
typedef struct dcarl_trace_state {
    int32_t* n;
    double* sum;
    double* sumsq;
    double* shift;
    double* V;
    int32_t* act_step;
} dcarl_trace_state_t;
int32_t dcarl_trace_resume_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                               const int32_t* slot_state, int32_t S, int32_t A, const dcarl_params_t* params,
                               const dcarl_trace_state_t* state, int32_t fresh, float* step_val, uint8_t* step_act,
                               float* vmax, int32_t* amax, void* stream);
int32_t dcarl_trace_resume_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                               const int32_t* slot_state, int32_t S, int32_t A, const dcarl_params_t* params,
                               const dcarl_trace_state_t* state, int32_t fresh, double* step_val, uint8_t* step_act,
                               float* vmax, int32_t* amax, void* stream);

/* Rewards must be FINITE.  The reference's np.argmax picks the first NaN; this library is built -fno-honor-nans (its tie-coded
 * keys are bit patterns) and orders NaN / Inf keys arbitrarily, so non-finite rewards are refused at the boundary instead:
 * dcarl_ingest_* flags them on the fly (info[7]); tables built any other way go through dcarl_count_nonfinite (f32 / f64
 * buffer of n elements, value_bytes = 4 | 8; count [device, int64] = number of NaN / Inf elements; an HBM-rate read) and the
 * Python builders raise ValueError.  dcarl_trace_* / dcarl_bounds_csr_* themselves do not re-check (a check in the hot loop
 * would cost every launch what only a corrupt table needs). */
int32_t dcarl_count_nonfinite(const void* values, int32_t value_bytes, int64_t n, int64_t* count, void* stream);

/* The ONE call of this library that synchronises: waits for `stream`, then reports whether a
 * dcarl_trace_* launch since the previous call gave up on a cross-wave hand-over (the multi-wave online kernel orders its
 * waves through LDS counters; a wave that waits ~3e10 cycles for a partner raises a fault word and ends instead of hanging
 * the GPU or killing the context).  DCARL_OK, or DCARL_ELAUNCH with a message in dcarl_last_error() — the outputs of those
 * launches are void then.  Never observed outside fault injection.  The same word reports the one INPUT the multi-wave online kernel
 * (A <= 16) refuses at run time: a (state, action) bucket that would pass 2^27 = 134 217 728 samples (its LDS counters count in
 * steps of 16, the byte offset into its count-root table) — the launch ends at once and dcarl_trace_status says so.  A caller that copies trace results to the host should
 * call it at that synchronisation point (the Python host side does: TraceResult.check(), reference_api.run_simulation,
 * bench.py after its timed region); dcarl_trace_* itself refuses to launch (DCARL_EDEVICE) when the fault word cannot be
 * reached, so a fault can never go unrecorded. */
int32_t dcarl_trace_status(void* stream);
/* Test hook (fault injection): sets the fault word the way a timed-out hand-over would, so that the reporting path of
 * dcarl_trace_status — and the host side's fail-closed accessors built on it — can be exercised without a broken GPU.
 * Synchronous.  EXPORTED BY THE RELEASE LIBRARY, deliberately: the GPU tests that prove a fault cannot go unnoticed run against
 * the very .so that ships, not a test build.  Its only effect is to make the next dcarl_trace_status() report a fault. */
int32_t dcarl_debug_raise_trace_fault(void);

/* ---- the third per-record trace (ABI 8): true_step_TSRL_value[idx].append(true_action_values[idx][TSRL_act]) (S1:96, S2:94) ----
 * out[e] = Q[state][step_act[e]] for every record element e of the sliced layout (same indexing as step_act; padding of a quad
 * that holds a record is written as 0, quads without records are not touched).  step_act: what dcarl_trace_* wrote; Q f64
 * [q_rows*A] = the true action values (action_value*.npy, a11), q_rows == 1: one row shared by every state (configs[1]'s replicas),
 * otherwise row = STATE id (slot_state as in dcarl_trace, nullable).  total_rows = slice_row_off[W] [host] (sizes the launch).
 * One pass, 1 byte read + 4 / 8 bytes written per record.  out: 16-byte aligned, f32 or f64 like the table's step_val. */
int32_t dcarl_true_step_values_f32(const uint8_t* step_act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state,
                                   int32_t S, int32_t A, const double* Q, int32_t q_rows, int64_t total_rows, float* out, void* stream);
int32_t dcarl_true_step_values_f64(const uint8_t* step_act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state,
                                   int32_t S, int32_t A, const double* Q, int32_t q_rows, int64_t total_rows, double* out, void* stream);

/* ---- top-2 gap census (ABI 8): how far every arg-max of S1:93-94 is from flipping ---------------------------------------------
 * The arg-max keys of this library carry the candidate id in their 5 low mantissa bits (first-max tie rule in one v_max_f64), so
 * two values whose bits agree above bit 5 are ordered by id, not by value.  The census counts, over EVERY arg-max evaluation of a
 * run, how often that happens and how the gaps between the best and the second-best candidate are distributed.
 * out: u64 [DCARL_CENSUS_WORDS] on the device, ACCUMULATED (several launches — shards, chunks — may add into one census); the caller
 * initialises it: zeros, except word 67 = ~0 (a running minimum).
 *   [0..63] histogram of rel = (best - runner_up) / |best| (code bits cleared): bin b counts 2^(b-53) <= rel < 2^(b-52); bin 0 also
 *           everything below (exact ties), bin 63 everything from 2^10 up
 *   [64] evaluations   [65] of them: top two in the SAME 32-ulp block (ordered by id)   [66] of those: both at the prior init_other
 *        (a true tie, which the reference's first-max rule breaks the same way: S1:51,94)
 *   [67] bit pattern (f64) of the smallest rel among the evaluations not counted in [65]   [68] evaluations without a runner-up (A == 1)
 * dcarl_top2_census_trace_*: the online loop (one evaluation per record; same inputs as dcarl_trace_*, the per-record outputs are not
 *   produced; a plain one-wave kernel, about five times the online kernel's time: instrumentation, run next to parity checks).
 * dcarl_top2_census_table: one evaluation per state on a final table V [S*A] as dcarl_trace_* / dcarl_bounds_csr_* return it. */
#define DCARL_CENSUS_WORDS 72
int32_t dcarl_top2_census_trace_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int32_t S,
                                    int32_t A, const dcarl_params_t* params, uint64_t* out, void* stream);
int32_t dcarl_top2_census_trace_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int32_t S,
                                    int32_t A, const dcarl_params_t* params, uint64_t* out, void* stream);
int32_t dcarl_top2_census_table(const double* V, int32_t S, int32_t A, const dcarl_params_t* params, uint64_t* out, void* stream);

/* ---- host-resident record tables (ABI 7) --------------------------------------------------------------
 * The reference's input is a file: np.load('.../data.npy') hands the loop an (N,4) float64 array in HOST memory
 * (S1:33-34, S2:32-33), and the loop consumes it front to back (S1:73).  A table that does not fit the GPU, or is not worth
 * holding there, is fed to the online loop in CHUNKS (dcarl_trace_resume_*): the copy of chunk k+1 runs on a copy stream
 * under the ingest + online kernel of chunk k (dcarl_amd/stream.py).  These three calls are what that pipeline needs of HIP
 * and the caller's framework may not expose:
 *   dcarl_host_pin / _unpin   page-lock (hipHostRegister) an EXISTING host range — the caller's own array, no staging copy —
 *                             so that asynchronous copies read it by DMA at the link's rate; unpin before freeing it.
 *   dcarl_copy_h2d / _d2h     hipMemcpyAsync on `stream`; the host range should be pinned (pageable memory is copied
 *                             through the runtime's bounce buffers and blocks the calling thread).
 * They allocate nothing, keep no state and do not synchronise. */
int32_t dcarl_host_pin(void* host, int64_t bytes);
int32_t dcarl_host_unpin(void* host);
int32_t dcarl_copy_h2d(void* dev, const void* host, int64_t bytes, void* stream);
int32_t dcarl_copy_d2h(void* host, const void* dev, int64_t bytes, void* stream);

/* ---- final-state ("batch") evaluation ------------------------------------------------------------
 * Same V table and arg-max as the end of the loop above, computed from samples sorted by (state, action):
 * bucket (s,a) = values[seg_off[s*A+a] .. seg_off[s*A+a+1]) (plain CSR: no alignment or padding contract beyond the
 * 16-byte alignment of `values` itself; empty buckets keep their initial value; `values` must hold at least one 16-byte
 * vector even when every bucket is empty — lanes without work re-read its first element).  If seg_off is NULL the buckets are
 * dense with n_dense samples each (bucket (s,a) starts at (s*A+a)*n_dense) and n_dense is ignored otherwise.
 * n_mean_hint = expected samples per bucket (0 = unknown -> n_dense, or "medium" for CSR): it only selects how many
 * lanes cooperate on one bucket, never the result.  Replaces S1:10-24 + S1:86-95 evaluated once per bucket.
 * Outputs as in dcarl_trace (V_out exact to 2^-47 relative). */
int32_t dcarl_bounds_csr_f32(const float* values, const int64_t* seg_off, int64_t n_dense, int64_t n_mean_hint, int32_t S,
                             int32_t A, const dcarl_params_t* params, double* V_out, int32_t* n_out, float* vmax,
                             int32_t* amax, void* stream);
int32_t dcarl_bounds_csr_f64(const double* values, const int64_t* seg_off, int64_t n_dense, int64_t n_mean_hint, int32_t S,
                             int32_t A, const dcarl_params_t* params, double* V_out, int32_t* n_out, float* vmax,
                             int32_t* amax, void* stream);

/* ---- the buckets themselves: record table -> (state, action) layout (S1:80 data_state_act[idx][act].append(R)) ----
 * dcarl_count_records: n_out[s*A+a] = number of records of state s with action a (= len(data_state_act[s][a]) after
 *   the whole table).  dcarl_group_records: values[seg_off[s*A+a] + k] = reward of the k-th such record in arrival
 *   order; seg_off [S*A+1] is the exclusive prefix sum of the counts (the caller's scan).  Inputs in the sliced layout;
 *   slot_state as in dcarl_trace (nullable): s is always the STATE, so the result feeds dcarl_bounds_csr_* in state order.
 *   seg_off must be exactly that prefix sum: the kernel places every record of the table (a block per slice regroups the 64 streams
 *   in chunks staged in LDS; 4.0 ms for 1.3e9 records).  Together with dcarl_ingest_group_* / _pack_* this is the fast route from
 *   the reference's (N,4) table to data_state_act for large f32 tables (dcarl_ingest_buckets_* below is the one-call form). */
int32_t dcarl_count_records(const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state,
                            int32_t S, int32_t A, int32_t* n_out, void* stream);
int32_t dcarl_group_records_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                                const int32_t* slot_state, int32_t S, int32_t A, const int64_t* seg_off, float* values, void* stream);
int32_t dcarl_group_records_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                                const int32_t* slot_state, int32_t S, int32_t A, const int64_t* seg_off, double* values, void* stream);

/* ---- the four bound functions themselves (S1:10-28) -----------------------------------------------
 * For each of B buckets values[off[b] .. off[b+1]) (off int64[B+1], empty buckets leave their row untouched):
 *   out[b] = { upper_bound (S1:10-12), lower_bound (S1:14-16), CI_lower_bound (S1:18-24), mean_value (S1:26-28) }
 * as 4 x f64.  Uses params->alpha/scale/cap only.  Backs the drop-in Python functions of the same names. */
int32_t dcarl_bucket_bounds_f32(const float* values, const int64_t* off, int64_t B, const dcarl_params_t* params,
                                double* out, void* stream);
int32_t dcarl_bucket_bounds_f64(const double* values, const int64_t* off, int64_t B, const dcarl_params_t* params,
                                double* out, void* stream);

/* ---- Sim2's cross-state running sum (S2:99-105) ---------------------------------------------------
 * overall[k] = sum over states already activated after arrival k of (current max V + 0.9).
 * dcarl_overall_delta writes, for arrival k (state rec_state[k], element rec_elem[k] in the sliced layout,
 * 0-based index rec_t[k] within its state), the change of that sum; dcarl_scan_f64 is the inclusive prefix sum.
 * scan_ws must hold dcarl_scan_workspace_bytes(N) bytes.
 * For a CONTINUED loop (dcarl_trace_resume_*, ABI version 6): t_base [S] (nullable) = the state's records before this chunk,
 * prev_val [S] (with t_base) = max_a V[s][a] as the chunk found it (in the step trace's storage precision): act_step counts
 * over all chunks, rec_t inside this one; the caller adds the running sum the previous chunk ended with to the scan. */
int32_t dcarl_overall_delta_f32(const float* step_val, const int32_t* act_step, const int32_t* rec_state,
                                const int64_t* rec_elem, const int32_t* rec_t,
                                int64_t N, double* delta, const int32_t* t_base, const double* prev_val, void* stream);
int32_t dcarl_overall_delta_f64(const double* step_val, const int32_t* act_step, const int32_t* rec_state,
                                const int64_t* rec_elem, const int32_t* rec_t,
                                int64_t N, double* delta, const int32_t* t_base, const double* prev_val, void* stream);
int64_t dcarl_scan_workspace_bytes(int64_t N);
int32_t dcarl_scan_f64(const double* in, double* out, int64_t N, void* scan_ws, void* stream);

/* ---- record ingest: the reference's arrival-ordered record table -> the device layouts (a11, S1:73-80) ---------------------
 * data is the reference's (N,4) float64 table {state idx, state feature, action, cumulative reward} in ARRIVAL order, on the
 * device, 32-byte aligned, N < 2^31.  What S1:73-80 does with it — idx = int(row[0]), act = int(row[2]),
 * data_state_act[idx][act].append(row[3]) — is a STABLE grouping; here it is a hand-written radix sort of compact records
 * (ingest.hip): no library sort, no permutation array, no copy of the table.
 *
 * Online layout, two calls (the caller allocates R / act between them, once the number of rows is known):
 *   dcarl_ingest_group_*: validates ids and rewards, groups the records by state in arrival order (kept in `workspace`),
 *     numbers the slots — states by descending stream length, stable, when flags has DCARL_INGEST_SORT_BY_LENGTH and S > 64,
 *     identity otherwise — and writes len [S] (records per SLOT), slot_state [S], state_slot [S] (inverse), slice_row_off
 *     [W+1] of the sliced layout at the top of this file, rec_state (nullable unless DCARL_INGEST_ARRIVAL) [N] = state of
 *     arrival k, and info (device, int64[16]):
 *       [0] total rows = slice_row_off[W]   [1] row bands over all slices (the pack kernel's work units; argument of dcarl_ingest_pack_*)
 *       [2] longest stream   [3] largest action id   [4] smallest state id   [5] largest state id   [6] smallest action id
 *       [7] flags: bit 0 = a reward is NaN / Inf (as stored: a float64 beyond the f32 range counts for _f32),
 *                  bit 1 = a state or action id is NaN / Inf                [8] N
 *     Ids are truncated toward zero like int() (S1:77-78).  The table is usable iff [7] == 0, 0 <= [4], [5] < S, 0 <= [6],
 *     [3] < A (the reference raises IndexError at S1:80 for ids past the table; negative ids would wrap there and are
 *     refused here); offending records are filed under id 0 so that nothing indexes out of range, and the caller raises.
 *   dcarl_ingest_pack_*: writes R [rows*64] and act [rows*64] (every element: padding as zeros) and, with
 *     DCARL_INGEST_ARRIVAL, rec_elem [N] (element e(s,t) of arrival k) and rec_t [N] (index of arrival k inside its state).
 *     Same N, S, A, flags and workspace as the group call; total_bands = info[1] [host].
 * Final-state layout, one call: dcarl_ingest_buckets_*: values [max(N,1)] = the rewards sorted by (state, action), arrival
 *   order kept inside a bucket (exactly data_state_act), seg_off [S*A+1] — the arguments of dcarl_bounds_csr_*; info as above
 *   ([0]-[2] unused).
 * workspace: dcarl_ingest_workspace_bytes(N, S, A, value_bytes = 4 | 8, flags, buckets = 0 | 1) bytes, 256-byte aligned.
 * Replaces round 2's dcarl_pack_records_* (which needed the caller's stable sort). */
#define DCARL_INGEST_SORT_BY_LENGTH 1
#define DCARL_INGEST_ARRIVAL 2
/* Which of the two implementations an online-layout ingest takes (ABI version 6; same results, bit for bit).  DIRECT: f32 tables
 * of at most 65 536 states without DCARL_INGEST_ARRIVAL are partitioned in tiles while they are compacted and packed straight
 * into the sliced layout (65 instead of 93 bytes of HBM traffic per record: ingest.hip); everything else, and tables below 2^20
 * records or 2 048 states, takes the radix sort + pack.  NO_DIRECT: never; FORCE_DIRECT: whenever the table is eligible, at any size.  The bits
 * must be the same in dcarl_ingest_workspace_bytes, dcarl_ingest_group_* and dcarl_ingest_pack_* of one table (they decide
 * the workspace layout): dcarl_ingest_pack_* returns DCARL_EINVAL when its N, S, A, flags or element width differ from the group
 * call of the same workspace (the library remembers the last 64 grouped workspaces by address). */
#define DCARL_INGEST_NO_DIRECT 4
#define DCARL_INGEST_FORCE_DIRECT 8
#define DCARL_INGEST_INFO_WORDS 16
int64_t dcarl_ingest_workspace_bytes(int64_t N, int32_t S, int32_t A, int32_t value_bytes, int32_t flags, int32_t buckets);
int32_t dcarl_ingest_group_f32(const double* data, int64_t N, int32_t S, int32_t A, int32_t flags, void* workspace, int32_t* len,
                               int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int32_t* rec_state,
                               int64_t* info, void* stream);
int32_t dcarl_ingest_group_f64(const double* data, int64_t N, int32_t S, int32_t A, int32_t flags, void* workspace, int32_t* len,
                               int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int32_t* rec_state,
                               int64_t* info, void* stream);
/* The same group step for records that never were (N,4) float64 rows (ABI 7): the sampler's own output — idx i32 [N] (the
 * state, or -1 for a visit data_sampling.py drops at DS:50-51), act i32 [N], R f32 [N], exactly what dcarl_sample_pairs writes —
 * goes straight into the direct ingest (12 instead of 32 bytes read per record; the (N,4) table DS:55,65 would have made of them
 * is never built).  Dropped visits do not enter the table; every other id is validated like a row's (info as above, plus
 * [9] = records kept).  For the tables the direct ingest serves: f32 storage, S <= 65 536, N >= 1, no DCARL_INGEST_ARRIVAL;
 * flags MUST carry DCARL_INGEST_FORCE_DIRECT (and the dcarl_ingest_workspace_bytes / dcarl_ingest_pack_f32 calls of the table the
 * same N, S, A and flags).  DCARL_EINVAL otherwise. */
int32_t dcarl_ingest_group_pairs_f32(const int32_t* idx, const int32_t* act, const float* R, int64_t N, int32_t S, int32_t A,
                                     int32_t flags, void* workspace, int32_t* len, int32_t* slot_state, int32_t* state_slot,
                                     int64_t* slice_row_off, int64_t* info, void* stream);
/* Host-resident tables, compacted BEFORE they cross the link (ABI 8).  The reference's table is an (N,4) float64 array in host memory
 * (np.load, S1:33); the path uses 12 of every 32 bytes of a row (ids, reward; column 1 is never read: S1:73) and the direct ingest
 * turns each row into one 8-byte record anyway.  dcarl_host_compact_rows_f32 [every pointer HOST] does that on the host: out[i] =
 * (state << 5 | action) | (uint64)bits(f32 reward) << 32 — the direct ingest's own compact record — with the row ingest's rules
 * operation by operation (ids truncated toward zero like int(), S1:77-78; NaN / Inf ids and rewards and f64 rewards beyond the f32
 * range flagged; offending records filed under id 0), and info [host, int64[16]] = what dcarl_ingest_group_* reports ([3]-[8]) for
 * these N rows: the caller raises exactly as for a device table.  Plain C loop, no allocation, re-entrant: callers split a table
 * into row ranges over their own threads (dcarl_amd/stream.py: the staging threads run it instead of a memcpy) and combine the
 * info words (min / max / or).  dcarl_ingest_group_packed_f32: the group step for such records on the DEVICE (8 instead of 32 bytes
 * read per record; ids re-checked: a corrupt record is filed under id 0 and shows in info [4]-[6] / [3]); same table restrictions,
 * flags, workspace and pack call as dcarl_ingest_group_pairs_f32.  S <= 65 536. */
int32_t dcarl_host_compact_rows_f32(const double* rows /* [host] (N,4) */, int64_t N, int32_t S, int32_t A, uint64_t* out /* [host] N */,
                                    int64_t* info /* [host] 16 */);
int32_t dcarl_ingest_group_packed_f32(const uint64_t* rec, int64_t N, int32_t S, int32_t A, int32_t flags, void* workspace, int32_t* len,
                                      int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int64_t* info, void* stream);
int32_t dcarl_ingest_pack_f32(int64_t N, int32_t S, int32_t A, int32_t flags, const void* workspace, const int32_t* len,
                              const int32_t* slot_state, const int64_t* slice_row_off, int64_t total_bands, float* R,
                              uint8_t* act, int64_t* rec_elem, int32_t* rec_t, void* stream);
int32_t dcarl_ingest_pack_f64(int64_t N, int32_t S, int32_t A, int32_t flags, const void* workspace, const int32_t* len,
                              const int32_t* slot_state, const int64_t* slice_row_off, int64_t total_bands, double* R,
                              uint8_t* act, int64_t* rec_elem, int32_t* rec_t, void* stream);
int32_t dcarl_ingest_buckets_f32(const double* data, int64_t N, int32_t S, int32_t A, void* workspace, float* values,
                                 int64_t* seg_off, int64_t* info, void* stream);
int32_t dcarl_ingest_buckets_f64(const double* data, int64_t N, int32_t S, int32_t A, void* workspace, double* values,
                                 int64_t* seg_off, int64_t* info, void* stream);

/* dcarl_slot_order: the slot numbering alone, for tables whose records are placed by someone else (the samplers, state-major
 * arrays): len_state [S] records per STATE (each <= max_len, which sizes the sort keys) -> len [S] per slot, slot_state /
 * state_slot, slice_row_off [W+1]; info[0] = total rows, info[2] = longest stream.  With DCARL_INGEST_SORT_BY_LENGTH and S > 64
 * the slots are the states by descending length (stable: the radix passes of the ingest), identity otherwise.  workspace:
 * dcarl_slot_order_workspace_bytes(S) bytes, 256-byte aligned. */
int64_t dcarl_slot_order_workspace_bytes(int32_t S);
int32_t dcarl_slot_order(const int32_t* len_state, int32_t S, int64_t max_len, int32_t flags, void* workspace, int32_t* len,
                         int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int64_t* info, void* stream);

/* dcarl_export_records_*: the inverse — a record table in the sliced layout back into the reference's (N,4) float64 rows
 * {state idx, state feature, action, cumulative reward} (what np.save writes as data.npy, DS:65), in a given arrival order:
 * arrival k is the record at element rec_elem[k] of state rec_state[k]; or, with both NULL, the dense interleaving of a table
 * with records_per_state records in every state (N == S * records_per_state): t = k / S, state = perm_t(k % S) — a bijection of
 * [0, S) per round t that looks random for power-of-two S (multiply / add / xor-shift steps) and is ((k % S) * mult + 7919 t) % S
 * otherwise (mult coprime to S) — element e(slot, t) with slot = state_slot[state] (nullable: identity).  state_value (nullable) [S] is
 * column 1 (0.0 when NULL). */
int32_t dcarl_export_records_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* state_slot,
                                 const double* state_value, int32_t S, int64_t records_per_state, int64_t mult, const int32_t* rec_state,
                                 const int64_t* rec_elem, int64_t N, double* out, void* stream);
int32_t dcarl_export_records_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* state_slot,
                                 const double* state_value, int32_t S, int64_t records_per_state, int64_t mult, const int32_t* rec_state,
                                 const int64_t* rec_elem, int64_t N, double* out, void* stream);

/* ---- Monte-Carlo return sampler (DS:5-9, DS:12-17, DS:45-55) --------------------------------------
 * Counter RNG: Philox-4x32-10, key = seed; standard normals by Box-Muller on (x1,x2); action = mulhi(x0, A).
 * dcarl_sample_state_records: T records for each of S states written straight into the dense sliced layout
 *   (rows per slice = ceil4(T)); record t of state s uses counter (t, s, stream, 0);
 *   R = Q[s][act] + sigma*z  (DS:9).  Q is f32 [S*A]; if q_rows == 1 every state shares Q[0..A).
 * dcarl_sample_pairs: N visit draws (DS:45,49-55): draw g = offset+i is draw k = g%4 of group G = g/4; the group
 *   owns the 12 words of the Philox blocks with counters (lo(c), hi(c), stream, 0), c = 3G, 3G+1, 3G+2, and draw k
 *   uses words 3k (action), 3k+1, 3k+2 (Box-Muller);
 *   idx = floor((3 + z_s)/6*S) or -1 when outside [0,S) (DS:14-15, DS:50-51), act, R = Q[idx][act] + sigma*z_r.
 *   The normals are f32 (hardware log / sin / cos: within 1e-4 of a float64 libm Box-Muller, typically 1e-6); the INDEX is
 *   exact: the float64 expression floor((3.0 + 1.0*z_s)/6*S) of NumPy on the f32 normal drawn, operation by operation.
 *   z_visit (nullable, f32 [N], ABI version 6) receives z_s, the visit normals behind the indices (DS:45), so that a
 *   checker can redo the index arithmetic on the very same draws.
 * dcarl_sample_from_noise_f64: the same arithmetic on INJECTED float64 noise, bit-exact with the reference:
 *   visit i: idx = floor((3 + 1*z_visit[i])/6*S); kept iff 0 <= idx < S; kept visits are numbered by
 *   kept_rank[i] (exclusive count of kept visits before i, caller-provided); row kept_rank[i] of
 *   out (M,4) f64 = {idx, states[idx], acts[rank], Q64[idx][acts[rank]] + sigma*z_reward[rank]} (DS:55).
 *   dcarl_visit_index_f64 computes idx/valid for the first step; dcarl_visit_floor_f64 is random_state_norm's own return
 *   value (DS:14-15): the raw int64 floor values, which may lie outside [0, S).
 * dcarl_state_manual_f64: random_state_manual (DS:19-28) on injected streams, bit-exact: u[i] = the i-th random.random(),
 *   r[j] = the j-th random.randint(1, state_num-1) — one per i with u[i] > 0.1, in order — kept_rank[i] = exclusive count of
 *   such i (the caller's scan): out[i] = u[i] > 0.1 ? r[kept_rank[i]] : 0. */
int32_t dcarl_sample_state_records(const float* Q, int32_t q_rows, int32_t S, int32_t A, int64_t T, double sigma,
                                   uint64_t seed, uint32_t stream_id, float* R, uint8_t* act, void* stream);
/* dcarl_sample_state_records_ragged: the same generator for a ragged table.  Slot k of the sliced layout
 *   (slice_row_off [W+1], total_rows = slice_row_off[W] passed by value [host]) holds len[k] records of state
 *   sid = slot_state[k] (nullable: sid = k); record t of it uses counter (t, sid + state_id_base, stream, 0) — identical to
 *   the dense call when every length is T and the base is 0; a rank that holds states [lo, hi) of a larger table passes
 *   state_id_base = lo and draws exactly the rows the whole table would hold; state_ids (nullable, i32 [S], ABI version 6)
 *   gives every local state its own global id instead (counter word state_ids[sid]): a rank that holds ANY subset of a
 *   larger table's states — the dealt slices of a record-balanced partition — draws exactly their rows — and its action is uniform over the first
 *   n_live[sid] candidates (nullable: A).
 *   Q is indexed by sid.  Padding elements are written as zeros.
 * dcarl_sample_buckets: samples drawn straight into the final-state layout (add_an_act_data, DS:5-9, once per sample
 *   of bucket (s,a)): values[i] = Q[s][a] + sigma*z_i for i in bucket (s,a) (seg_off / n_dense as in dcarl_bounds_csr);
 *   z_i is normal i%4 of the Philox block with counter (lo(i/4), hi(i/4), stream, 1): Box-Muller (cos, sin) of words
 *   (x0, x1) for i%4 = 0, 1 and of (x2, x3) for i%4 = 2, 3. */
int32_t dcarl_sample_state_records_ragged(const float* Q, int32_t q_rows, int32_t S, int32_t A,
                                          const int64_t* slice_row_off, int64_t total_rows, const int32_t* len,
                                          const int32_t* slot_state, const int32_t* n_live, double sigma, uint64_t seed,
                                          uint32_t stream_id, uint32_t state_id_base, const int32_t* state_ids, float* R,
                                          uint8_t* act, void* stream);
int32_t dcarl_sample_buckets(const float* Q, int32_t q_rows, int32_t S, int32_t A, const int64_t* seg_off, int64_t n_dense,
                             double sigma, uint64_t seed, uint32_t stream_id, float* values, void* stream);
int32_t dcarl_sample_pairs(const float* Q, int32_t S, int32_t A, int64_t N, double sigma, uint64_t seed,
                           uint64_t offset, uint32_t stream_id, int32_t* idx, int32_t* act, float* R, float* z_visit,
                           void* stream);
int32_t dcarl_visit_index_f64(const double* z_visit, int64_t M, int32_t S, int32_t* idx, void* stream);
int32_t dcarl_visit_floor_f64(const double* z_visit, int64_t M, int32_t S, int64_t* out, void* stream);
int32_t dcarl_state_manual_f64(const double* u, const int64_t* kept_rank, const int32_t* r, int64_t M, int32_t* out, void* stream);
int32_t dcarl_sample_from_noise_f64(const int32_t* idx, const int64_t* kept_rank, int64_t M, const double* states,
                                    const double* Q64, int32_t S, int32_t A, const int32_t* acts,
                                    const double* z_reward, double sigma, double* out_rows, void* stream);

/* ---- multi-GPU: the one collective of the path ------------------------------------------------------------
 * States shard across ranks with no data-path communication; per step ONE all-gather of the per-state summaries
 * {arg-max i32, max V f32 bits, activation step i32} reassembles them.  These are thin calls into RCCL (librccl.so.1,
 * resolved with dlopen at the first call so that a process that already carries torch's copy shares it); the
 * communicator is an opaque handle owned by the caller — the library itself still keeps no state.
 *   dcarl_comm_unique_id: [host] 128-byte id made on ONE rank and handed to the others by the caller's own means.
 *   dcarl_comm_init: collective over all ranks; the calling thread's current HIP device is the rank's GPU.
 *   dcarl_allgather_summary: recv [nranks*bytes] <- every rank's send [bytes] (device pointers), in rank order, on `stream`.
 * dcarl_amd/dist.py uses these when DCARL_COMM=rccl and torch.distributed's all_gather_into_tensor (the same RCCL
 * underneath) otherwise. */
/* Per-rank GLOBAL statistics of a block of states, for callers that do not need the per-state summaries on every rank
 * (SURVEY 8(e)): the all-gather then moves 272 bytes per rank instead of 12 bytes per state.
 *   activated   = states whose arg-max has left the rule action (act_step >= 0; S1:98-99)
 *   sum_vmax    = sum of max_a V[s][a] over the states (f64; block-ordered: run-to-run identical for a given S)
 *   policy_hist = number of states per arg-max candidate (S1:94)
 * dcarl_summary_stats: amax / vmax / act_step [S] as dcarl_trace writes them; A <= 32; workspace
 * dcarl_workspace_bytes(DCARL_WS_SUMMARY, S, 0, 0) bytes, 16-byte aligned; out: ONE dcarl_summary_t on the device. */
typedef struct dcarl_summary {
    int64_t activated;
    double sum_vmax;
    int64_t policy_hist[DCARL_MAX_ACTIONS];
} dcarl_summary_t;
int32_t dcarl_summary_stats(const int32_t* amax, const float* vmax, const int32_t* act_step, int32_t S, int32_t A,
                            void* workspace, dcarl_summary_t* out, void* stream);
#define DCARL_UNIQUE_ID_BYTES 128
int32_t dcarl_comm_unique_id(uint8_t* id /* [host] 128 bytes */);
int32_t dcarl_comm_init(int32_t nranks, int32_t rank, const uint8_t* id /* [host] */, void** comm /* [host] out */);
int32_t dcarl_allgather_summary(void* comm, const void* send, void* recv, int64_t bytes, void* stream);
int32_t dcarl_comm_destroy(void* comm);

/* ---- CARLA record ingest ------------------------------------------------------------------
 * The collector writes one record per episode as `str(ndarray[20]), used_action, episode_reward`
 * (Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Agent/drl_library/dqn/dqn_value_collect.py:128-137;
 * sample: Example_Collected_data_in_CARLA.txt); dcarl_amd/carla_records.py parses that text on the host.  The confidence
 * path needs INTEGER state ids (S1:77 `idx = int(idx_ori)`); the reference's simulation data is pre-indexed and it has
 * no rule for CARLA observations, so the rule here is this library's: a uniform grid, cells[i][k] =
 * floor(obs[i][k] / cell_width[k]) (obs [N][D] row-major, cell_width [D] on the device, D <= 64); rows with equal cells
 * share a state id (dcarl_state_ids below). */
int32_t dcarl_state_cells_f64(const double* obs, int64_t N, int32_t D, const double* cell_width, int32_t* cells, uint64_t* hash,
                              void* stream);
/* hash (nullable) [N]: the 64-bit hash of every row of cells, made while the row is in registers (D a multiple of 4, obs and
 * cells 16-byte aligned); passed on to dcarl_state_ids it saves that call a second pass over the rows.
 * dcarl_state_ids: rows of cells [N][D] with equal coordinates share a state id; ids [N] are dense (0 .. n-1) and numbered
 * in order of FIRST APPEARANCE.  Sort-free: 64-bit hash (given, or computed from the rows when hash is NULL) -> open-addressing
 * table in the workspace, every row verified against its representative, one prefix sum.  max_states = the number of
 * distinct states the caller expects (0 = unknown: N): the table holds 2 x that many slots, and a table sized for the
 * states instead of the records stays in the L2.  out (device, int64[3]) = {number of states, number of rows whose cells
 * differ from their representative's despite an equal 64-bit hash, 1 if the table overflowed (more distinct states than
 * max_states allowed for)}: the ids are valid iff out[1] == 0 and out[2] == 0 (the host side raises / retries with 0).
 * workspace: dcarl_workspace_bytes(DCARL_WS_STATE_IDS, max_states, 0, N) bytes, 16-byte aligned.  N < 2^31. */
int32_t dcarl_state_ids(const int32_t* cells, const uint64_t* hash, int64_t N, int32_t D, int64_t max_states, void* workspace,
                        int32_t* ids, int64_t* out, void* stream);

/* dcarl_index_states_f64 = dcarl_state_cells_f64 + dcarl_state_ids in one call (the same rule, the same outputs: cells [N][D]
 * and ids [N], out as above): the cells kernel puts every row into the id table while the row is in its registers, so the
 * hashes never travel through HBM and the table's round trips pass under the kernel's streaming.  D a multiple of 4, obs and
 * cells 16-byte aligned (anything else: the two calls above).  workspace as for dcarl_state_ids. */
int32_t dcarl_index_states_f64(const double* obs, int64_t N, int32_t D, const double* cell_width, int64_t max_states, void* workspace,
                               int32_t* cells, int32_t* ids, int64_t* out, void* stream);

/* ---- field variant of the confidence test ("RLS"; SURVEY.md 8(f) rank 2, the first row past the simulation path) ----
 * RLS = Field_testing/Software_and_Raw_Data_on_Self-Driving_Vehicle/software/src/tools/DCARL/stable_baselines/deepq/RLS.py
 * A visited row is (obs[20], action) with its recorded value; row s owns the box [s - d, s + d] (RLS:68 visited_state_dist,
 * RLS:193-194 insert); a query point "visits" every row whose box contains it (RLS:161-163; closed intervals; the
 * reference asks a third-party R-tree, this library scans).
 * dcarl_rls_neighbour_stats_f64: for each of Q query points (row-major [Q][21]): count = visited rows (RLS:161-163),
 *   mean / var = np.mean / np.var of their values, or -1 / -1 when count == 0 (RLS:165-181).  states [N][21] row-major
 *   (the layout of visited_state.txt), values [N] (column 1 of visited_value.txt), half_width [21].
 *   workspace: dcarl_rls_workspace_bytes(N, Q) bytes, 16-byte aligned, contents irrelevant before / after.
 * dcarl_rls_decide: RLS:120-157 act_test for B decisions from statistics laid out [B][1 + n_cand] (column 0 = rule
 *   action 0, column c = candidate action c): the first candidate c with count_rule >= visited_times_thres,
 *   count_c >= min_rl_visits, mean_rule <= rule_mean_gate and norm.cdf((mean_c - mean_rule) /
 *   sqrt(var_rule/count_rule + var_c/count_c)) > confidence_thres, else 0. */
#define DCARL_RLS_DIM 21          /* RLS:30,59 obs_dimension + 1 */
typedef struct dcarl_rls_params {
    int32_t visited_times_thres; /* RLS:14  30 */
    int32_t min_rl_visits;       /* RLS:141 5 */
    double rule_mean_gate;       /* RLS:141 -0.1: candidates are only considered when mean_rule <= gate */
    double confidence_thres;     /* RLS:120 0.5 */
} dcarl_rls_params_t;
void dcarl_rls_default_params(dcarl_rls_params_t* p /* [host] */);
int64_t dcarl_rls_workspace_bytes(int64_t N, int32_t Q);
int32_t dcarl_rls_neighbour_stats_f64(const double* states, const double* values, int64_t N, const double* half_width,
                                      const double* queries, int32_t Q, void* workspace, int64_t* count, double* mean,
                                      double* var, void* stream);
int32_t dcarl_rls_decide(const int64_t* count, const double* mean, const double* var, int32_t B, int32_t n_cand,
                         const dcarl_rls_params_t* params /* [host] */, int32_t* action, void* stream);
/* dcarl_rls_gate_train: the TRAIN-time policy RLS:78-118 (act / act_train / should_use_rule) for B observations, from the rule
 *   action's statistics (count_rule, mean_rule [B]: dcarl_rls_neighbour_stats_f64 at state_with_action(obs, 0)) and injected
 *   exploration draws explore [B] (the reference's random.uniform(-1, 0), RLS:112): use_rule (nullable) [B] = count_rule <
 *   visited_times_thres or explore < mean_rule; action (nullable) [B] = use_rule ? 0 : rl_action (RLS:85-89). */
int32_t dcarl_rls_gate_train(const int64_t* count_rule, const double* mean_rule, const double* explore, const int32_t* rl_action,
                             int32_t B, int32_t visited_times_thres, int32_t* action, uint8_t* use_rule, void* stream);

/* ---- episode-return reduction (where the cumulative-reward column comes from) -----------------
 * TS = Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Test_Scenarios/TestScenario_Town03.py,
 * DVC = .../Data_From_Carla/Agent/drl_library/dqn/dqn_value_collect.py, RLS as above.
 * dcarl_episode_returns_f64: E episodes, episode e = steps [ep_off[e], ep_off[e+1]).  Per step (TS:386,402-421):
 *   v = sqrt(vx^2 + vy^2), reward = 0.1*sqrt(v); flags bit 0 (collision) -> -100; bit 2 (stuck) without bit 1 (passed)
 *   -> 0.0 (the reference's `if pass ... elif stuck`).  step_reward (nullable) [N]; episode_reward [E] = sum of the step
 *   rewards (DVC:119; fixed-order tree sum: agrees with the reference's running sum to rounding, ~1e-15 relative);
 *   ave_speed (nullable) [E] = mean of v (TS:411 AveSpeed).
 * dcarl_nstep_backup_f64: RLS.add_data's value stream (RLS:185-215).  value[t] = rew[t] for a transition with at least
 *   `horizon` (10) successors in its episode (RLS:188-199); for the last `horizon` transitions of an episode that ended
 *   (ep_done[e] != 0) value[t] = rew[last] * gamma_pow[last - t] (RLS:202-215), gamma_pow [horizon] (device) =
 *   gamma**k as the host computes it (dcarl_gamma_powers: C pow, what CPython's float ** int calls) -> bit-exact;
 *   transitions still buffered when the stream stops (ep_done[e] == 0) are not recorded: recorded (nullable) [N] = 0/1. */
void dcarl_gamma_powers(double gamma, int32_t horizon, double* out /* [host] horizon entries */);
int32_t dcarl_episode_returns_f64(const double* vx, const double* vy, const uint8_t* flags, const int64_t* ep_off, int64_t E,
                                  double* step_reward, double* episode_reward, double* ave_speed, void* stream);
int32_t dcarl_nstep_backup_f64(const double* rew, const int64_t* ep_off, const uint8_t* ep_done, int64_t E,
                               const double* gamma_pow, int32_t horizon, double* value, uint8_t* recorded, void* stream);

/* ---- candidate generation in the Frenet frame (the step that produces the actions) ----------
 * JTP = Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Agent/zzz/JunctionTrajectoryPlanner.py
 * dcarl_frenet_candidates_f64: JTP:292-340 calc_frenet_paths for B start states.  start [B][5] = {s0, c_speed, c_d,
 *   c_d_d, c_d_dd}.  Candidate c = (i_d * n_T + i_T) * n_v + i_v (the reference's loop order: lateral offset, horizon,
 *   target speed).  traj (nullable) [B][n_cand][8][nt_max]: fields d, d_d, d_dd, d_ddd, s, s_d, s_dd, s_ddd sampled at
 *   t = i*dt, i < nt[i_T] (zeros beyond); cost (nullable) [B][n_cand][3] = {cd, cv, cf} (JTP:326-336).
 * The grid is what the reference's module constants expand to (JTP:14-40): d = arange(MAX_LEFT_WIDTH, MAX_RIGHT_WIDTH+1,
 *   D_ROAD_W), T = arange(MINT, MAXT, DT), nt[i] = len(arange(0, T[i], DT)), tv = arange(ts - D_T_S*N_S_SAMPLE,
 *   ts + D_T_S*N_S_SAMPLE, D_T_S); dcarl_frenet_default_grid fills it with those defaults (5 x 1 x 2 = 10 candidates,
 *   14 samples). */
typedef struct dcarl_frenet_grid {
    int32_t n_d, n_T, n_v, nt_max;
    int32_t nt[8];
    double d[16], T[8], tv[8];
    double dt;            /* JTP:21 DT */
    double target_speed;  /* JTP:24 TARGET_SPEED */
    double kj, kt, kd, klat, klon;   /* JTP:35-39 */
} dcarl_frenet_grid_t;
void dcarl_frenet_default_grid(dcarl_frenet_grid_t* g /* [host] */);
int32_t dcarl_frenet_candidates_f64(const double* start, int64_t B, const dcarl_frenet_grid_t* grid /* [host] */,
                                    double* traj, double* cost, void* stream);
/* dcarl_frenet_global_paths_f64: JTP:342-379 calc_global_paths against ONE reference path shared by the batch, given as
 *   the reference's Spline2D (Agent/zzz/cubic_spline_planner.py): knots [n_knots] (cumulative chord length) and
 *   segments [n_knots-1][8] = {ax, bx, cx, dx, ay, by, cy, dy}.  glob [B][n_cand][5][nt_max] = x, y, yaw, ds, c (c has one
 *   entry less); path_len [B][n_cand] = samples inside the spline (the reference stops at the first one outside).
 * dcarl_frenet_select: JTP:123-130 get_optimal_trajectory per start state: candidates by ascending cf (stable), minus
 *   those failing check_paths (JTP:381-394: speed, acceleration, curvature limits), the first one predict.check_collision
 *   (Agent/zzz/predict.py:21-60) lets through -> choice = index + 1, or 0 (brake).  obstacles [B][n_obs][5] = {x, y, vx,
 *   vy, yaw} of the vehicles `found_interested_vehicles` kept (predict.py:62-82; selection is the caller's), each two
 *   circles at +- move_gap along its heading moving at constant velocity (predict.py:84-110).
 *   ok (nullable) [B][n_cand]: bit 0 = passes check_paths, bit 1 = collision-free. */
typedef struct dcarl_frenet_limits {
    double max_speed;      /* JTP:14 MAX_SPEED 50/3.6 */
    double max_accel;      /* JTP:15 MAX_ACCEL 10 */
    double max_curvature;  /* JTP:16 MAX_CURVATURE 500 */
    double check_radius;   /* JTP:29 ROBOT_RADIUS 1 (predict.py:12) */
    double move_gap;       /* JTP:31 MOVE_GAP 1 */
    int32_t n_predict;     /* len(arange(0, MAXT, DT)) = 15 (predict.py:88) */
} dcarl_frenet_limits_t;
void dcarl_frenet_default_limits(dcarl_frenet_limits_t* l /* [host] */);
int32_t dcarl_frenet_global_paths_f64(const double* traj, int64_t B, const dcarl_frenet_grid_t* grid /* [host] */,
                                      const double* knots, const double* segments, int32_t n_knots, double* glob,
                                      int32_t* path_len, void* stream);
int32_t dcarl_frenet_select(const double* traj, const double* glob, const int32_t* path_len, const double* cost,
                            const double* obstacles, int32_t n_obs, int64_t B, const dcarl_frenet_grid_t* grid /* [host] */,
                            const dcarl_frenet_limits_t* limits /* [host] */, int32_t* choice, uint8_t* ok, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCARL_H */
