// extern "C" surface of libdcarl_hip.so (declared in include/dcarl.h): argument validation, parameter
// derivation and kernel launches.  No state, no allocation, no synchronisation; errors are codes + a
// thread-local message.
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "trace_common.h"

namespace dcarl {
int* trace_fault_word();
template <typename T>
int launch_trace(const T*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const DevParams&, T*, uint8_t*,
                 int32_t*, double*, int32_t*, float*, int32_t*, hipStream_t, const TraceCarry&);
template <typename T>
int launch_bounds_csr(const T*, const int64_t*, int64_t, int64_t, int, int, const DevParams&, double*, int32_t*,
                      float*, int32_t*, hipStream_t);
template <typename T>
int launch_bucket_bounds(const T*, const int64_t*, int64_t, const DevParams&, double*, hipStream_t);
int64_t ingest_workspace_bytes(int64_t, int, int, int, bool, bool, int);
int64_t slot_order_workspace_bytes(int);
int launch_slot_order(const int32_t*, int, int64_t, bool, void*, int32_t*, int32_t*, int32_t*, int64_t*, int64_t*, hipStream_t);
template <typename T>
int launch_ingest_group(const double*, int64_t, int, int, bool, bool, void*, int32_t*, int32_t*, int32_t*, int64_t*, int32_t*, int64_t*,
                        hipStream_t, int);
int launch_ingest_group_pairs(const int32_t*, const int32_t*, const float*, int64_t, int, int, bool, void*, int32_t*, int32_t*, int32_t*,
                              int64_t*, int64_t*, hipStream_t);
int launch_ingest_group_packed(const uint64_t*, int64_t, int, int, bool, void*, int32_t*, int32_t*, int32_t*, int64_t*, int64_t*, hipStream_t);
template <typename T>
int launch_ingest_pack(int64_t, int, int, bool, bool, const void*, const int32_t*, const int32_t*, const int64_t*, int64_t, T*, uint8_t*,
                       int64_t*, int32_t*, hipStream_t, int);
template <typename T>
int launch_ingest_buckets(const double*, int64_t, int, int, void*, T*, int64_t*, int64_t*, hipStream_t);
template <typename T>
int launch_export_records(const T*, const uint8_t*, const int64_t*, const int32_t*, const double*, int, int64_t, const int32_t*,
                          const int64_t*, int64_t, double*, hipStream_t);
template <typename T>
int launch_overall_delta(const T*, const int32_t*, const int32_t*, const int64_t*, const int32_t*, int64_t, double*,
                         const int32_t*, const double*, hipStream_t);
int trace_status(hipStream_t);
int trace_raise_fault();
int launch_count_nonfinite(const void*, int, int64_t, int64_t*, hipStream_t);
int64_t scan_workspace_bytes(int64_t N);
int launch_state_cells(const double*, int64_t, int, const double*, int32_t*, unsigned long long*, hipStream_t);
int64_t state_ids_workspace_bytes(int64_t N, int64_t max_states);
int64_t summary_workspace_bytes(int64_t S);
int launch_summary_stats(const int32_t*, const float*, const int32_t*, int, int, void*, dcarl_summary_t*, hipStream_t);
int launch_state_ids(const int32_t*, const unsigned long long*, int64_t, int, int64_t, void*, int32_t*, int64_t*, hipStream_t);
int launch_index_states(const double*, int64_t, int, const double*, int64_t, void*, int32_t*, int32_t*, int64_t*, hipStream_t);
int launch_frenet(const double*, int64_t, const dcarl_frenet_grid_t&, double*, double*, hipStream_t);
int launch_frenet_global(const double*, int64_t, const dcarl_frenet_grid_t&, const double*, const double*, int, double*, int32_t*,
                         hipStream_t);
int launch_frenet_select(const double*, const double*, const int32_t*, const double*, const double*, int, int64_t,
                         const dcarl_frenet_grid_t&, const dcarl_frenet_limits_t&, int32_t*, uint8_t*, hipStream_t);
int64_t rls_workspace_bytes(int64_t N, int32_t Q);
int launch_rls_stats(const double*, const double*, int64_t, const double*, const double*, int32_t, void*, int64_t*, double*,
                     double*, hipStream_t);
int launch_rls_decide(const int64_t*, const double*, const double*, int32_t, int32_t, const dcarl_rls_params_t&, int32_t*,
                      hipStream_t);
int launch_rls_gate_train(const int64_t*, const double*, const double*, const int32_t*, int32_t, int32_t, int32_t*, uint8_t*, hipStream_t);
int launch_scan(const double*, double*, int64_t, void*, hipStream_t);
int launch_episode_returns(const double*, const double*, const uint8_t*, const int64_t*, int64_t, double*, double*, double*,
                           hipStream_t);
int launch_nstep_backup(const double*, const int64_t*, const uint8_t*, int64_t, const double*, int, double*, uint8_t*,
                        hipStream_t);
int launch_sample_state_records(const float*, int, int, int, int64_t, double, uint64_t, uint32_t, float*, uint8_t*,
                                hipStream_t);
int launch_sample_pairs(const float*, int, int, int64_t, double, uint64_t, uint64_t, uint32_t, int32_t*, int32_t*,
                        float*, float*, hipStream_t);
int launch_sample_state_records_ragged(const float*, int, int, int, const int64_t*, int64_t, const int32_t*, const int32_t*,
                                       const int32_t*, double, uint64_t, uint32_t, uint32_t, const int32_t*, float*, uint8_t*, hipStream_t);
int launch_sample_buckets(const float*, int, int, int, const int64_t*, int64_t, double, uint64_t, uint32_t, float*,
                          hipStream_t);
template <typename T>
int launch_group_records(const T*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const int64_t*, T*, int32_t*,
                         hipStream_t);
int launch_visit_index(const double*, int64_t, int, int32_t*, hipStream_t);
int launch_visit_floor(const double*, int64_t, int, int64_t*, hipStream_t);
int launch_state_manual(const double*, const int64_t*, const int32_t*, int64_t, int32_t*, hipStream_t);
int launch_sample_from_noise(const int32_t*, const int64_t*, int64_t, const double*, const double*, int, int,
                             const int32_t*, const double*, double, double*, hipStream_t);
template <typename T>
int launch_true_step(const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const double*, int, int64_t, T*, hipStream_t);
int launch_census_table(const double*, int, int, const DevParams&, unsigned long long*, hipStream_t);
template <typename T>
int launch_census_trace(const T*, const uint8_t*, const int64_t*, const int32_t*, int, int, const DevParams&, unsigned long long*, hipStream_t);
}  // namespace dcarl

namespace {

thread_local char g_err[512] = "";
thread_local char g_kernel[160] = "";
thread_local char g_problem[320] = "";                 // what a launcher found wrong before its launch (dcarl::note_launch_problem)

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int after_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (g_problem[0]) {                                    // (reported even if the launch itself went through: its LDS request did not)
        char msg[320];
        snprintf(msg, sizeof(msg), "%s", g_problem);
        g_problem[0] = 0;
        return fail(DCARL_ELAUNCH, "%s: %s%s%s", what, msg, e != hipSuccess ? "; launch: " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    }
    if (e != hipSuccess) return fail(DCARL_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return DCARL_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int derive(const dcarl_params_t* in, int A, dcarl::DevParams* out) {
    if (!in) return fail(DCARL_EINVAL, "params is NULL");
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "A=%d outside [1,%d]", A, DCARL_MAX_ACTIONS);
    if (in->rule_act < 0 || in->rule_act >= A) return fail(DCARL_EINVAL, "rule_act=%d outside [0,%d)", in->rule_act, A);
    if (!(in->alpha > 0.0 && in->alpha < 1.0)) return fail(DCARL_EINVAL, "alpha=%g outside (0,1)", in->alpha);
    if (in->n_thres < 0) return fail(DCARL_EINVAL, "n_thres=%d negative", in->n_thres);
    out->rule_act = in->rule_act;
    out->n_thres = in->n_thres;
    out->hoeff = in->scale * std::sqrt(std::log(1.0 / in->alpha) / 2.0);   // S1:12 scale*sqrt(log(1/alpha)/2/n)
    out->cap = in->cap;
    out->init_rule = in->init_rule;
    out->init_other = in->init_other;
    return DCARL_OK;
}

// state == NULL: dcarl_trace_* (one shot); otherwise dcarl_trace_resume_* — the kernels' V_out / n_out / act_step ARE the state's
// arrays then (read at the start unless `fresh`, written at the end, each row by the one lane that serves the state).
template <typename T>
int trace_impl(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state,
               int32_t S, int32_t A,
               const dcarl_params_t* params, T* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
               int32_t* n_out, float* vmax, int32_t* amax, void* stream, const dcarl_trace_state_t* state = nullptr,
               int32_t fresh = 1) {
    dcarl::DevParams p;
    if (int rc = derive(params, A, &p)) return rc;
    if (S < 0) return fail(DCARL_EINVAL, "S=%d negative", S);
    dcarl::TraceCarry cy{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1};
    if (state) {
        if (!state->n || !state->sum || !state->sumsq || !state->shift || !state->V || !state->act_step)
            return fail(DCARL_EINVAL, "dcarl_trace_resume: every array of the state must be non-NULL");
        cy = dcarl::TraceCarry{state->n, state->sum, state->sumsq, state->shift, state->V, state->act_step, fresh ? 1 : 0};
        act_step = state->act_step; V_out = state->V; n_out = state->n;
    }
    if (S == 0) return DCARL_OK;
    if (!R || !act || !slice_row_off || !len) return fail(DCARL_EINVAL, "R/act/slice_row_off/len must be non-NULL");
    if (!aligned16(R) || (reinterpret_cast<uintptr_t>(act) & 3u) || (step_val && !aligned16(step_val)) ||
        (step_act && (reinterpret_cast<uintptr_t>(step_act) & 3u)))
        return fail(DCARL_EINVAL, "R/step_val need 16-byte and act/step_act 4-byte alignment");
    // a hand-over of the multi-wave kernel that never arrives is reported through the library's fault word (dcarl_trace_status):
    // without the word such a fault would store through a null address and could not be reported, so nothing is launched
    if (!dcarl::trace_fault_word())
        return fail(DCARL_EDEVICE, "dcarl_trace: the library's fault word cannot be reached (hipGetSymbolAddress failed); not launching");
    dcarl::launch_trace<T>(R, act, slice_row_off, len, slot_state, S, A, p, step_val, step_act, act_step, V_out, n_out, vmax, amax,
                           static_cast<hipStream_t>(stream), cy);
    return after_launch("dcarl_trace");
}

template <typename T>
int bounds_impl(const T* values, const int64_t* seg_off, int64_t n_dense, int32_t S, int32_t A,
                const dcarl_params_t* params, double* V_out, int32_t* n_out, float* vmax, int32_t* amax, void* stream,
                int64_t n_mean_hint) {
    dcarl::DevParams p;
    if (int rc = derive(params, A, &p)) return rc;
    if (S < 0) return fail(DCARL_EINVAL, "S=%d negative", S);
    if (S == 0) return DCARL_OK;
    if (!values) return fail(DCARL_EINVAL, "values is NULL");
    if (!seg_off && n_dense < 0) return fail(DCARL_EINVAL, "n_dense=%lld negative", (long long)n_dense);
    if (!aligned16(values)) return fail(DCARL_EINVAL, "values needs 16-byte alignment");
    dcarl::launch_bounds_csr<T>(values, seg_off, n_dense, n_mean_hint, S, A, p, V_out, n_out, vmax, amax,
                                static_cast<hipStream_t>(stream));
    return after_launch("dcarl_bounds_csr");
}

template <typename T>
int true_step_impl(const uint8_t* step_act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state, int32_t S,
                          int32_t A, const double* Q, int32_t q_rows, int64_t total_rows, T* out, void* stream) {
    if (S < 0) return fail(DCARL_EINVAL, "dcarl_true_step_values: S=%d negative", S);
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "dcarl_true_step_values: A=%d outside [1,%d]", A, DCARL_MAX_ACTIONS);
    if (S == 0) return DCARL_OK;
    if (!step_act || !slice_row_off || !len || !Q || !out) return fail(DCARL_EINVAL, "dcarl_true_step_values: step_act/slice_row_off/len/Q/out must be non-NULL");
    if (q_rows != 1 && q_rows != S) return fail(DCARL_EINVAL, "dcarl_true_step_values: q_rows=%d is neither 1 nor S=%d", q_rows, S);
    if (total_rows < 0) return fail(DCARL_EINVAL, "dcarl_true_step_values: total_rows negative");
    if (!aligned16(out) || (reinterpret_cast<uintptr_t>(step_act) & 3u)) return fail(DCARL_EINVAL, "dcarl_true_step_values: out needs 16-byte and step_act 4-byte alignment");
    dcarl::launch_true_step<T>(step_act, slice_row_off, len, slot_state, S, A, Q, q_rows, total_rows, out, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_true_step_values");
}
template <typename T>
int census_trace_impl(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int32_t S, int32_t A,
                             const dcarl_params_t* params, uint64_t* out, void* stream) {
    dcarl::DevParams p;
    if (int rc = derive(params, A, &p)) return rc;
    if (S < 0) return fail(DCARL_EINVAL, "dcarl_top2_census_trace: S=%d negative", S);
    if (!out) return fail(DCARL_EINVAL, "dcarl_top2_census_trace: out is NULL");
    if (S == 0) return DCARL_OK;
    if (!R || !act || !slice_row_off || !len) return fail(DCARL_EINVAL, "dcarl_top2_census_trace: R/act/slice_row_off/len must be non-NULL");
    if (!aligned16(R) || (reinterpret_cast<uintptr_t>(act) & 3u)) return fail(DCARL_EINVAL, "dcarl_top2_census_trace: R needs 16-byte and act 4-byte alignment");
    if (dcarl::launch_census_trace<T>(R, act, slice_row_off, len, S, A, p, reinterpret_cast<unsigned long long*>(out), static_cast<hipStream_t>(stream)))
        return fail(DCARL_EINVAL, "dcarl_top2_census_trace: no instance for A=%d", A);
    return after_launch("dcarl_top2_census_trace");
}
// which path an online-layout ingest takes (ingest.hip, use_direct): the caller's flags, the same in all three calls of a table
int direct_mode_of(int32_t flags) { return (flags & DCARL_INGEST_NO_DIRECT) ? 0 : (flags & DCARL_INGEST_FORCE_DIRECT) ? 1 : -1; }

int check_ingest(const double* data, int64_t N, int32_t S, int32_t A, const void* ws, const char* who) {
    if (N < 0 || N > 0x7fffffff) return fail(DCARL_EINVAL, "%s: N=%lld outside [0,2^31)", who, (long long)N);
    if (S < 1 || S > (1 << 26)) return fail(DCARL_EINVAL, "%s: S=%d outside [1,2^26]", who, S);
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "%s: A=%d outside [1,%d]", who, A, DCARL_MAX_ACTIONS);
    if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255u)) return fail(DCARL_EINVAL, "%s: workspace is NULL or not 256-byte aligned", who);
    if (N && (!data || (reinterpret_cast<uintptr_t>(data) & 31u))) return fail(DCARL_EINVAL, "%s: data is NULL or not 32-byte aligned", who);
    return DCARL_OK;
}

// The pack call rebuilds the workspace layout from the arguments its caller repeats (and from DCARL_INGEST_PAIRS, read at call
// time): anything that differs from the group call of the same workspace would silently misread it.  The group call leaves what
// decides the layout under the workspace's address (host side: nothing to synchronise with), the pack call compares.  The last
// PLAN_SLOTS workspaces are remembered; an older one is packed unchecked, as before.
struct IngestStamp { const void* ws; int64_t N; int32_t S, A, flags, vb, pairs; };
constexpr int PLAN_SLOTS = 64;
constexpr int32_t PLAN_FLAGS = DCARL_INGEST_SORT_BY_LENGTH | DCARL_INGEST_ARRIVAL | DCARL_INGEST_NO_DIRECT | DCARL_INGEST_FORCE_DIRECT;
std::mutex g_stamp_mu;
IngestStamp g_stamps[PLAN_SLOTS];
unsigned g_stamp_next = 0;
int32_t pairs_knob() { const char* e = DCARL_KNOB("DCARL_INGEST_PAIRS"); return !(e && e[0] == '0'); }
void stamp_ingest(const void* ws, int64_t N, int32_t S, int32_t A, int32_t flags, int32_t vb) {
    std::lock_guard<std::mutex> lock(g_stamp_mu);
    IngestStamp* at = nullptr;
    for (IngestStamp& e : g_stamps) if (e.ws == ws) at = &e;
    if (!at) at = &g_stamps[g_stamp_next++ % PLAN_SLOTS];
    *at = IngestStamp{ws, N, S, A, flags & PLAN_FLAGS, vb, pairs_knob()};
}
int check_stamp(const void* ws, int64_t N, int32_t S, int32_t A, int32_t flags, int32_t vb) {
    std::lock_guard<std::mutex> lock(g_stamp_mu);
    for (const IngestStamp& e : g_stamps) {
        if (e.ws != ws) continue;
        if (e.N == N && e.S == S && e.A == A && e.flags == (flags & PLAN_FLAGS) && e.vb == vb && e.pairs == pairs_knob()) return DCARL_OK;
        return fail(DCARL_EINVAL,
                    "dcarl_ingest_pack: this workspace was grouped with N=%lld S=%d A=%d flags=%d, %d-byte values, pair passes=%d; the pack call "
                    "says N=%lld S=%d A=%d flags=%d, %d-byte values, pair passes=%d",
                    (long long)e.N, e.S, e.A, e.flags, e.vb, e.pairs, (long long)N, S, A, flags & PLAN_FLAGS, vb, pairs_knob());
    }
    return DCARL_OK;
}

template <typename T>
int ingest_group_impl(const double* data, int64_t N, int32_t S, int32_t A, int32_t flags, void* workspace, int32_t* len,
                             int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int32_t* rec_state, int64_t* info,
                             void* stream) {
    if (int rc = check_ingest(data, N, S, A, workspace, "dcarl_ingest_group")) return rc;
    if (!len || !slot_state || !state_slot || !slice_row_off || !info) return fail(DCARL_EINVAL, "dcarl_ingest_group: NULL output");
    const bool arrival = (flags & DCARL_INGEST_ARRIVAL) != 0;
    if (arrival && N && !rec_state) return fail(DCARL_EINVAL, "dcarl_ingest_group: DCARL_INGEST_ARRIVAL needs rec_state");
    stamp_ingest(workspace, N, S, A, flags, (int32_t)sizeof(T));
    dcarl::launch_ingest_group<T>(data, N, S, A, (flags & DCARL_INGEST_SORT_BY_LENGTH) != 0, arrival, workspace, len, slot_state,
                                  state_slot, slice_row_off, rec_state, info, static_cast<hipStream_t>(stream), direct_mode_of(flags));
    return after_launch("dcarl_ingest_group");
}
template <typename T>
int ingest_pack_impl(int64_t N, int32_t S, int32_t A, int32_t flags, const void* workspace, const int32_t* len,
                            const int32_t* slot_state, const int64_t* slice_row_off, int64_t total_bands, T* R, uint8_t* act,
                            int64_t* rec_elem, int32_t* rec_t, void* stream) {
    if (int rc = check_ingest(nullptr, 0, S, A, workspace, "dcarl_ingest_pack")) return rc;
    if (N < 0 || N > 0x7fffffff) return fail(DCARL_EINVAL, "dcarl_ingest_pack: N=%lld outside [0,2^31)", (long long)N);
    // (the workspace holds one entry per band for at most N/32 + 2W + 2 bands, W = ceil(S/64): never admit more than that)
    if (total_bands < 0 || total_bands > N / 32 + 2 * (((int64_t)S + 63) / 64) + 2)
        return fail(DCARL_EINVAL, "dcarl_ingest_pack: total_bands=%lld is not what dcarl_ingest_group reported", (long long)total_bands);
    if (int rc = check_stamp(workspace, N, S, A, flags, (int32_t)sizeof(T))) return rc;
    if (total_bands == 0) return DCARL_OK;
    if (!len || !slice_row_off || !R || !act) return fail(DCARL_EINVAL, "dcarl_ingest_pack: NULL argument");
    if (!aligned16(R) || (reinterpret_cast<uintptr_t>(act) & 3u)) return fail(DCARL_EINVAL, "R needs 16-byte and act 4-byte alignment");
    const bool arrival = (flags & DCARL_INGEST_ARRIVAL) != 0;
    if (arrival && N && (!rec_elem || !rec_t)) return fail(DCARL_EINVAL, "dcarl_ingest_pack: DCARL_INGEST_ARRIVAL needs rec_elem and rec_t");
    dcarl::launch_ingest_pack<T>(N, S, A, (flags & DCARL_INGEST_SORT_BY_LENGTH) != 0, arrival, workspace, len, slot_state, slice_row_off,
                                 total_bands, R, act, rec_elem, rec_t, static_cast<hipStream_t>(stream), direct_mode_of(flags));
    return after_launch("dcarl_ingest_pack");
}
template <typename T>
int export_records_impl(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* state_slot, const double* state_value,
                        int32_t S, int64_t records_per_state, int64_t mult, const int32_t* rec_state, const int64_t* rec_elem, int64_t N,
                        double* out, void* stream) {
    if (N < 0 || S < 1) return fail(DCARL_EINVAL, "dcarl_export_records: N negative or S < 1");
    if (N == 0) return DCARL_OK;
    if (!R || !act || !slice_row_off || !out) return fail(DCARL_EINVAL, "dcarl_export_records: NULL argument");
    if (reinterpret_cast<uintptr_t>(out) & 31u) return fail(DCARL_EINVAL, "dcarl_export_records: out needs 32-byte alignment");
    if ((rec_state == nullptr) != (rec_elem == nullptr)) return fail(DCARL_EINVAL, "dcarl_export_records: rec_state and rec_elem go together");
    if (!rec_elem) {
        if (records_per_state < 0 || N != records_per_state * S) return fail(DCARL_EINVAL, "dcarl_export_records: the dense order needs N == S * records_per_state");
        int64_t a = mult, b = S;
        while (b) { const int64_t r = a % b; a = b; b = r; }
        if (mult < 1 || a != 1) return fail(DCARL_EINVAL, "dcarl_export_records: mult=%lld is not coprime to S=%d", (long long)mult, S);
    }
    dcarl::launch_export_records<T>(R, act, slice_row_off, state_slot, state_value, S, mult, rec_state, rec_elem, N, out,
                                    static_cast<hipStream_t>(stream));
    return after_launch("dcarl_export_records");
}

template <typename T>
int ingest_buckets_impl(const double* data, int64_t N, int32_t S, int32_t A, void* workspace, T* values, int64_t* seg_off,
                               int64_t* info, void* stream) {
    if (int rc = check_ingest(data, N, S, A, workspace, "dcarl_ingest_buckets")) return rc;
    if (!values || !seg_off || !info) return fail(DCARL_EINVAL, "dcarl_ingest_buckets: NULL output");
    dcarl::launch_ingest_buckets<T>(data, N, S, A, workspace, values, seg_off, info, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_ingest_buckets");
}
}  // namespace

namespace dcarl {
int comm_unique_id(uint8_t*);
int comm_init(int, int, const uint8_t*, void**);
int comm_allgather(void*, const void*, void*, int64_t, hipStream_t);
int comm_destroy(void*);
int comm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
void note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
void note_launch_problem(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_problem, sizeof(g_problem), fmt, ap);
    va_end(ap);
}
}  // namespace dcarl

extern "C" {

const char* dcarl_last_kernel(void) { return g_kernel; }

#ifndef DCARL_BUILD_ID
#define DCARL_BUILD_ID "unknown"
#endif
const char* dcarl_build_id(void) { return DCARL_BUILD_ID; }

int32_t dcarl_summary_stats(const int32_t* amax, const float* vmax, const int32_t* act_step, int32_t S, int32_t A,
                            void* workspace, dcarl_summary_t* out, void* stream) {
    if (S < 0) return fail(DCARL_EINVAL, "dcarl_summary_stats: S=%d negative", S);
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "dcarl_summary_stats: A=%d outside [1,%d]", A, DCARL_MAX_ACTIONS);
    if (!out || !workspace || (S && (!amax || !vmax || !act_step))) return fail(DCARL_EINVAL, "dcarl_summary_stats: NULL argument");
    if (!aligned16(workspace) || !aligned16(out)) return fail(DCARL_EINVAL, "workspace / out need 16-byte alignment");
    dcarl::launch_summary_stats(amax, vmax, act_step, S, A, workspace, out, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_summary_stats");
}

int32_t dcarl_comm_unique_id(uint8_t* id) {
    if (!id) return fail(DCARL_EINVAL, "dcarl_comm_unique_id: id is NULL");
    return dcarl::comm_unique_id(id);
}
int32_t dcarl_comm_init(int32_t nranks, int32_t rank, const uint8_t* id, void** comm) {
    if (!id || !comm) return fail(DCARL_EINVAL, "dcarl_comm_init: NULL argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(DCARL_EINVAL, "dcarl_comm_init: rank %d of %d", rank, nranks);
    return dcarl::comm_init(nranks, rank, id, comm);
}
int32_t dcarl_allgather_summary(void* comm, const void* send, void* recv, int64_t bytes, void* stream) {
    if (!comm) return fail(DCARL_EINVAL, "dcarl_allgather_summary: comm is NULL");
    if (bytes < 0) return fail(DCARL_EINVAL, "dcarl_allgather_summary: bytes negative");
    if (bytes == 0) return DCARL_OK;
    if (!send || !recv) return fail(DCARL_EINVAL, "dcarl_allgather_summary: NULL buffer");
    return dcarl::comm_allgather(comm, send, recv, bytes, static_cast<hipStream_t>(stream));
}
int32_t dcarl_comm_destroy(void* comm) {
    if (!comm) return DCARL_OK;
    return dcarl::comm_destroy(comm);
}

int64_t dcarl_workspace_bytes(int32_t kind, int64_t S, int32_t A, int64_t N) {
    if (S < 0 || N < 0) return 0;
    switch (kind) {
        case DCARL_WS_SCAN: return dcarl::scan_workspace_bytes(N);
        case DCARL_WS_RLS: return S > 0x7fffffff ? 0 : dcarl::rls_workspace_bytes(N, (int32_t)S);
        case DCARL_WS_STATE_IDS: return dcarl::state_ids_workspace_bytes(N, S);
        case DCARL_WS_SUMMARY: return dcarl::summary_workspace_bytes(S);
        case DCARL_WS_INGEST_F32:
        case DCARL_WS_INGEST_F64: {
            if (S < 1 || S > 0x7fffffff || N > 0x7fffffff || A < 1 || A > DCARL_MAX_ACTIONS) return 0;
            const int vb = kind == DCARL_WS_INGEST_F32 ? 4 : 8;
            // every option on: with and without arrival bookkeeping (the latter may take the direct path at any size), buckets
            const int64_t t = dcarl::ingest_workspace_bytes(N, (int)S, A, vb, true, false, 1), b = dcarl::ingest_workspace_bytes(N, (int)S, A, vb, false, true, 1),
                          d = dcarl::ingest_workspace_bytes(N, (int)S, A, vb, false, false, 1);
            return t > b ? (t > d ? t : d) : (b > d ? b : d);
        }
        default: return 0;
    }
}

int32_t dcarl_version(void) { return DCARL_ABI_VERSION; }

const char* dcarl_last_error(void) { return g_err; }

void dcarl_default_params(dcarl_params_t* p) {
    if (!p) return;
    p->rule_act = 0; p->n_thres = 10; p->alpha = 0.05; p->scale = 150.0; p->cap = 100.0;
    p->init_rule = 100.0; p->init_other = -50.0;
}

int32_t dcarl_device_info(int32_t dev, dcarl_device_info_t* out) {
    if (!out) return fail(DCARL_EINVAL, "out is NULL");
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fail(DCARL_EDEVICE, "hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
    std::memset(out, 0, sizeof(*out));
    std::strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
    if (char* c = std::strchr(out->arch, ':')) *c = 0;
    out->compute_units = prop.multiProcessorCount;
    out->wavefront = prop.warpSize;
    out->hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (std::strcmp(out->arch, "gfx950") != 0)
        return fail(DCARL_EDEVICE, "device %d is %s; this library is built for gfx950 only", dev, out->arch);
    return DCARL_OK;
}

// ---- host-resident tables: page-locking the caller's own array and stream-ordered copies (dcarl_amd/stream.py) ----
int32_t dcarl_host_pin(void* host, int64_t bytes) {
    if (!host || bytes <= 0) return fail(DCARL_EINVAL, "dcarl_host_pin: NULL range or bytes <= 0");
    hipError_t e = hipHostRegister(host, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(DCARL_EDEVICE, "hipHostRegister(%lld bytes): %s", (long long)bytes, hipGetErrorString(e));
    }
    return DCARL_OK;
}
int32_t dcarl_host_unpin(void* host) {
    if (!host) return fail(DCARL_EINVAL, "dcarl_host_unpin: NULL");
    hipError_t e = hipHostUnregister(host);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(DCARL_EDEVICE, "hipHostUnregister: %s", hipGetErrorString(e));
    }
    return DCARL_OK;
}
static int32_t copy_async(void* dst, const void* src, int64_t bytes, hipMemcpyKind kind, void* stream, const char* what) {
    if (bytes < 0 || (bytes && (!dst || !src))) return fail(DCARL_EINVAL, "%s: NULL pointer or negative size", what);
    if (bytes == 0) return DCARL_OK;
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, kind, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(DCARL_EDEVICE, "%s: hipMemcpyAsync(%lld bytes): %s", what, (long long)bytes, hipGetErrorString(e));
    }
    return DCARL_OK;
}
int32_t dcarl_copy_h2d(void* dev, const void* host, int64_t bytes, void* stream) {
    return copy_async(dev, host, bytes, hipMemcpyHostToDevice, stream, "dcarl_copy_h2d");
}
int32_t dcarl_copy_d2h(void* host, const void* dev, int64_t bytes, void* stream) {
    return copy_async(host, dev, bytes, hipMemcpyDeviceToHost, stream, "dcarl_copy_d2h");
}

int32_t dcarl_trace_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                        const int32_t* slot_state, int32_t S, int32_t A, const dcarl_params_t* params, float* step_val, uint8_t* step_act,
                        int32_t* act_step, double* V_out, int32_t* n_out, float* vmax, int32_t* amax, void* stream) {
    return trace_impl<float>(R, act, slice_row_off, len, slot_state, S, A, params, step_val, step_act, act_step, V_out, n_out, vmax,
                             amax, stream);
}
int32_t dcarl_trace_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                        const int32_t* slot_state, int32_t S, int32_t A, const dcarl_params_t* params, double* step_val, uint8_t* step_act,
                        int32_t* act_step, double* V_out, int32_t* n_out, float* vmax, int32_t* amax, void* stream) {
    return trace_impl<double>(R, act, slice_row_off, len, slot_state, S, A, params, step_val, step_act, act_step, V_out, n_out,
                              vmax, amax, stream);
}

int32_t dcarl_trace_resume_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                               const int32_t* slot_state, int32_t S, int32_t A, const dcarl_params_t* params,
                               const dcarl_trace_state_t* state, int32_t fresh, float* step_val, uint8_t* step_act, float* vmax,
                               int32_t* amax, void* stream) {
    if (!state) return fail(DCARL_EINVAL, "dcarl_trace_resume: state is NULL");
    return trace_impl<float>(R, act, slice_row_off, len, slot_state, S, A, params, step_val, step_act, nullptr, nullptr, nullptr, vmax,
                             amax, stream, state, fresh);
}
int32_t dcarl_trace_resume_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                               const int32_t* slot_state, int32_t S, int32_t A, const dcarl_params_t* params,
                               const dcarl_trace_state_t* state, int32_t fresh, double* step_val, uint8_t* step_act, float* vmax,
                               int32_t* amax, void* stream) {
    if (!state) return fail(DCARL_EINVAL, "dcarl_trace_resume: state is NULL");
    return trace_impl<double>(R, act, slice_row_off, len, slot_state, S, A, params, step_val, step_act, nullptr, nullptr, nullptr, vmax,
                              amax, stream, state, fresh);
}

int32_t dcarl_true_step_values_f32(const uint8_t* step_act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state,
                                   int32_t S, int32_t A, const double* Q, int32_t q_rows, int64_t total_rows, float* out, void* stream) {
    return true_step_impl<float>(step_act, slice_row_off, len, slot_state, S, A, Q, q_rows, total_rows, out, stream);
}
int32_t dcarl_true_step_values_f64(const uint8_t* step_act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state,
                                   int32_t S, int32_t A, const double* Q, int32_t q_rows, int64_t total_rows, double* out, void* stream) {
    return true_step_impl<double>(step_act, slice_row_off, len, slot_state, S, A, Q, q_rows, total_rows, out, stream);
}

int32_t dcarl_top2_census_trace_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int32_t S,
                                    int32_t A, const dcarl_params_t* params, uint64_t* out, void* stream) {
    return census_trace_impl<float>(R, act, slice_row_off, len, S, A, params, out, stream);
}
int32_t dcarl_top2_census_trace_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int32_t S,
                                    int32_t A, const dcarl_params_t* params, uint64_t* out, void* stream) {
    return census_trace_impl<double>(R, act, slice_row_off, len, S, A, params, out, stream);
}
int32_t dcarl_top2_census_table(const double* V, int32_t S, int32_t A, const dcarl_params_t* params, uint64_t* out, void* stream) {
    dcarl::DevParams p;
    if (int rc = derive(params, A, &p)) return rc;
    if (S < 0) return fail(DCARL_EINVAL, "dcarl_top2_census_table: S=%d negative", S);
    if (!out) return fail(DCARL_EINVAL, "dcarl_top2_census_table: out is NULL");
    if (S == 0) return DCARL_OK;
    if (!V) return fail(DCARL_EINVAL, "dcarl_top2_census_table: V is NULL");
    dcarl::launch_census_table(V, S, A, p, reinterpret_cast<unsigned long long*>(out), static_cast<hipStream_t>(stream));
    return after_launch("dcarl_top2_census_table");
}

int32_t dcarl_count_nonfinite(const void* values, int32_t value_bytes, int64_t n, int64_t* count, void* stream) {
    if (n < 0 || (value_bytes != 4 && value_bytes != 8)) return fail(DCARL_EINVAL, "dcarl_count_nonfinite: n negative or value_bytes not 4 / 8");
    if (!count || (n && !values)) return fail(DCARL_EINVAL, "dcarl_count_nonfinite: NULL argument");
    dcarl::launch_count_nonfinite(values, value_bytes, n, count, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_count_nonfinite");
}

int32_t dcarl_debug_raise_trace_fault(void) {
    return dcarl::trace_raise_fault() == 0 ? DCARL_OK : fail(DCARL_EDEVICE, "dcarl_debug_raise_trace_fault: cannot reach the fault word");
}

int32_t dcarl_trace_status(void* stream) {
    const int v = dcarl::trace_status(static_cast<hipStream_t>(stream));
    if (v < 0) return fail(DCARL_EDEVICE, "dcarl_trace_status: the stream or the fault word could not be read (a device fault?)");
    if (v >= 2) return fail(DCARL_ELAUNCH, "dcarl_trace: a (state, action) bucket would have passed 2^27 samples, the range of the online kernel's "
                                           "counters (A <= 16); the outputs of the launches since the last dcarl_trace_status() are void");
    if (v > 0) return fail(DCARL_ELAUNCH, "dcarl_trace: a cross-wave hand-over of the online kernel never arrived; the outputs of the "
                                          "launches since the last dcarl_trace_status() are void");
    return DCARL_OK;
}

int32_t dcarl_bounds_csr_f32(const float* values, const int64_t* seg_off, int64_t n_dense, int64_t n_mean_hint, int32_t S,
                             int32_t A, const dcarl_params_t* params, double* V_out, int32_t* n_out, float* vmax,
                             int32_t* amax, void* stream) {
    return bounds_impl<float>(values, seg_off, seg_off ? 0 : n_dense, S, A, params, V_out, n_out, vmax, amax, stream,
                              n_mean_hint > 0 ? n_mean_hint : (seg_off ? 64 : n_dense));
}
int32_t dcarl_bounds_csr_f64(const double* values, const int64_t* seg_off, int64_t n_dense, int64_t n_mean_hint, int32_t S,
                             int32_t A, const dcarl_params_t* params, double* V_out, int32_t* n_out, float* vmax,
                             int32_t* amax, void* stream) {
    return bounds_impl<double>(values, seg_off, seg_off ? 0 : n_dense, S, A, params, V_out, n_out, vmax, amax, stream,
                               n_mean_hint > 0 ? n_mean_hint : (seg_off ? 64 : n_dense));
}

static int check_sliced(const void* act, const int64_t* slice_row_off, const int32_t* len, int32_t S, int32_t A,
                        const char* who) {
    if (S < 0) return fail(DCARL_EINVAL, "%s: S=%d negative", who, S);
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "%s: A=%d outside [1,%d]", who, A, DCARL_MAX_ACTIONS);
    if (S && (!act || !slice_row_off || !len)) return fail(DCARL_EINVAL, "%s: act/slice_row_off/len must be non-NULL", who);
    if (S && (reinterpret_cast<uintptr_t>(act) & 3u)) return fail(DCARL_EINVAL, "%s: act needs 4-byte alignment", who);
    return DCARL_OK;
}

int32_t dcarl_count_records(const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state,
                            int32_t S, int32_t A, int32_t* n_out, void* stream) {
    if (int rc = check_sliced(act, slice_row_off, len, S, A, "dcarl_count_records")) return rc;
    if (S == 0) return DCARL_OK;
    if (!n_out) return fail(DCARL_EINVAL, "dcarl_count_records: n_out is NULL");
    dcarl::launch_group_records<float>(nullptr, act, slice_row_off, len, slot_state, S, A, nullptr, nullptr, n_out,
                                       static_cast<hipStream_t>(stream));
    return after_launch("dcarl_count_records");
}
int32_t dcarl_group_records_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                                const int32_t* slot_state, int32_t S, int32_t A, const int64_t* seg_off, float* values, void* stream) {
    if (int rc = check_sliced(act, slice_row_off, len, S, A, "dcarl_group_records")) return rc;
    if (S == 0) return DCARL_OK;
    if (!R || !seg_off || !values) return fail(DCARL_EINVAL, "dcarl_group_records: NULL argument");
    dcarl::launch_group_records<float>(R, act, slice_row_off, len, slot_state, S, A, seg_off, values, nullptr,
                                       static_cast<hipStream_t>(stream));
    return after_launch("dcarl_group_records");
}
int32_t dcarl_group_records_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                                const int32_t* slot_state, int32_t S, int32_t A, const int64_t* seg_off, double* values, void* stream) {
    if (int rc = check_sliced(act, slice_row_off, len, S, A, "dcarl_group_records")) return rc;
    if (S == 0) return DCARL_OK;
    if (!R || !seg_off || !values) return fail(DCARL_EINVAL, "dcarl_group_records: NULL argument");
    dcarl::launch_group_records<double>(R, act, slice_row_off, len, slot_state, S, A, seg_off, values, nullptr,
                                        static_cast<hipStream_t>(stream));
    return after_launch("dcarl_group_records");
}

int32_t dcarl_bucket_bounds_f32(const float* values, const int64_t* off, int64_t B, const dcarl_params_t* params,
                                double* out, void* stream) {
    dcarl::DevParams p;
    if (int rc = derive(params, 1, &p)) return rc;
    if (B < 0) return fail(DCARL_EINVAL, "B negative");
    if (B && (!values || !off || !out)) return fail(DCARL_EINVAL, "dcarl_bucket_bounds: NULL argument");
    if (B && (reinterpret_cast<uintptr_t>(out) & 31u)) return fail(DCARL_EINVAL, "out needs 32-byte alignment");
    dcarl::launch_bucket_bounds<float>(values, off, B, p, out, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_bucket_bounds");
}
int32_t dcarl_bucket_bounds_f64(const double* values, const int64_t* off, int64_t B, const dcarl_params_t* params,
                                double* out, void* stream) {
    dcarl::DevParams p;
    if (int rc = derive(params, 1, &p)) return rc;
    if (B < 0) return fail(DCARL_EINVAL, "B negative");
    if (B && (!values || !off || !out)) return fail(DCARL_EINVAL, "dcarl_bucket_bounds: NULL argument");
    if (B && (reinterpret_cast<uintptr_t>(out) & 31u)) return fail(DCARL_EINVAL, "out needs 32-byte alignment");
    dcarl::launch_bucket_bounds<double>(values, off, B, p, out, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_bucket_bounds");
}

int32_t dcarl_overall_delta_f32(const float* step_val, const int32_t* act_step, const int32_t* rec_state,
                                const int64_t* rec_elem, const int32_t* rec_t, int64_t N, double* delta, const int32_t* t_base,
                                const double* prev_val, void* stream) {
    if (N < 0) return fail(DCARL_EINVAL, "N negative");
    if (N && (!step_val || !act_step || !rec_state || !rec_elem || !rec_t || !delta))
        return fail(DCARL_EINVAL, "dcarl_overall_delta: NULL argument");
    if ((t_base == nullptr) != (prev_val == nullptr)) return fail(DCARL_EINVAL, "dcarl_overall_delta: t_base and prev_val go together");
    dcarl::launch_overall_delta<float>(step_val, act_step, rec_state, rec_elem, rec_t, N, delta, t_base, prev_val,
                                       static_cast<hipStream_t>(stream));
    return after_launch("dcarl_overall_delta");
}
int32_t dcarl_overall_delta_f64(const double* step_val, const int32_t* act_step, const int32_t* rec_state,
                                const int64_t* rec_elem, const int32_t* rec_t, int64_t N, double* delta, const int32_t* t_base,
                                const double* prev_val, void* stream) {
    if (N < 0) return fail(DCARL_EINVAL, "N negative");
    if (N && (!step_val || !act_step || !rec_state || !rec_elem || !rec_t || !delta))
        return fail(DCARL_EINVAL, "dcarl_overall_delta: NULL argument");
    if ((t_base == nullptr) != (prev_val == nullptr)) return fail(DCARL_EINVAL, "dcarl_overall_delta: t_base and prev_val go together");
    dcarl::launch_overall_delta<double>(step_val, act_step, rec_state, rec_elem, rec_t, N, delta, t_base, prev_val,
                                        static_cast<hipStream_t>(stream));
    return after_launch("dcarl_overall_delta");
}

int64_t dcarl_scan_workspace_bytes(int64_t N) { return N < 0 ? 0 : dcarl::scan_workspace_bytes(N); }

int32_t dcarl_scan_f64(const double* in, double* out, int64_t N, void* scan_ws, void* stream) {
    if (N < 0) return fail(DCARL_EINVAL, "N negative");
    if (N && (!in || !out || !scan_ws)) return fail(DCARL_EINVAL, "dcarl_scan_f64: NULL argument");
    dcarl::launch_scan(in, out, N, scan_ws, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_scan_f64");
}

int64_t dcarl_ingest_workspace_bytes(int64_t N, int32_t S, int32_t A, int32_t value_bytes, int32_t flags, int32_t buckets) {
    if (N < 0 || N > 0x7fffffff || S < 1 || A < 1 || A > DCARL_MAX_ACTIONS || (value_bytes != 4 && value_bytes != 8)) return 0;
    return dcarl::ingest_workspace_bytes(N, S, A, value_bytes, (flags & DCARL_INGEST_ARRIVAL) != 0, buckets != 0, direct_mode_of(flags));
}

int32_t dcarl_ingest_group_pairs_f32(const int32_t* idx, const int32_t* act, const float* R, int64_t N, int32_t S, int32_t A,
                                     int32_t flags, void* workspace, int32_t* len, int32_t* slot_state, int32_t* state_slot,
                                     int64_t* slice_row_off, int64_t* info, void* stream) {
    if (int rc = check_ingest(nullptr, 0, S, A, workspace, "dcarl_ingest_group_pairs")) return rc;
    if (N < 1 || N > 0x7fffffff) return fail(DCARL_EINVAL, "dcarl_ingest_group_pairs: N=%lld outside [1,2^31)", (long long)N);
    if (S > 65536) return fail(DCARL_EINVAL, "dcarl_ingest_group_pairs: S=%d beyond the direct ingest's 65 536 states", S);
    if (!(flags & DCARL_INGEST_FORCE_DIRECT) || (flags & (DCARL_INGEST_ARRIVAL | DCARL_INGEST_NO_DIRECT)))
        return fail(DCARL_EINVAL, "dcarl_ingest_group_pairs: flags=%d must carry DCARL_INGEST_FORCE_DIRECT and neither DCARL_INGEST_ARRIVAL nor "
                                  "DCARL_INGEST_NO_DIRECT", flags);
    if (!idx || !act || !R || ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(R)) & 3u))
        return fail(DCARL_EINVAL, "dcarl_ingest_group_pairs: idx / act / R is NULL or not 4-byte aligned");
    if (!len || !slot_state || !state_slot || !slice_row_off || !info) return fail(DCARL_EINVAL, "dcarl_ingest_group_pairs: NULL output");
    stamp_ingest(workspace, N, S, A, flags, 4);
    dcarl::launch_ingest_group_pairs(idx, act, R, N, S, A, (flags & DCARL_INGEST_SORT_BY_LENGTH) != 0, workspace, len, slot_state, state_slot,
                                     slice_row_off, info, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_ingest_group_pairs");
}

int32_t dcarl_host_compact_rows_f32(const double* rows, int64_t N, int32_t S, int32_t A, uint64_t* out, int64_t* info) {
    if (N < 0) return fail(DCARL_EINVAL, "dcarl_host_compact_rows: N=%lld negative", (long long)N);
    if (S < 1 || S > 65536) return fail(DCARL_EINVAL, "dcarl_host_compact_rows: S=%d outside [1,65536] (the packed key holds 27 bits of state id; the direct ingest serves 65 536 states)", S);
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "dcarl_host_compact_rows: A=%d outside [1,%d]", A, DCARL_MAX_ACTIONS);
    if (!info || (N && (!rows || !out))) return fail(DCARL_EINVAL, "dcarl_host_compact_rows: NULL argument");
    // the same rule, operation by operation, as the device side of the row ingest (ingest.hip convert(): ids truncated toward zero like
    // int() (S1:77-78), NaN / Inf ids and rewards flagged, offending records filed under id 0 so that nothing ever indexes out of range)
    int32_t smin = INT32_MAX, smax = INT32_MIN, amin = INT32_MAX, amax = INT32_MIN;
    int64_t flags = 0;
    const double* __restrict__ r = rows;
    uint64_t* __restrict__ o = out;
    for (int64_t i = 0; i < N; ++i) {
        const double sd = r[4 * i], ad = r[4 * i + 2], wd = r[4 * i + 3];
        const bool s_nf = !(std::fabs(sd) <= 1.7976931348623157e308), a_nf = !(std::fabs(ad) <= 1.7976931348623157e308);
        const int32_t si = s_nf ? INT32_MIN : std::fabs(sd) < 2.0e9 ? (int32_t)sd : sd < 0 ? INT32_MIN : INT32_MAX;
        const int32_t ai = a_nf ? INT32_MIN : std::fabs(ad) < 2.0e9 ? (int32_t)ad : ad < 0 ? INT32_MIN : INT32_MAX;
        smin = si < smin ? si : smin; smax = si > smax ? si : smax;
        amin = ai < amin ? ai : amin; amax = ai > amax ? ai : amax;
        if (s_nf || a_nf) flags |= 2;
        if (!(std::fabs(wd) <= 3.4028234663852886e38)) flags |= 1;             // NaN, Inf, or beyond the f32 range
        const uint32_t st = (si >= 0 && si < S) ? (uint32_t)si : 0u, a = (ai >= 0 && ai < A) ? (uint32_t)ai : 0u;
        const float wf = (float)wd;
        uint32_t wb;
        std::memcpy(&wb, &wf, 4);
        o[i] = (uint64_t)((st << 5) | a) | ((uint64_t)wb << 32);
    }
    for (int k = 0; k < DCARL_INGEST_INFO_WORDS; ++k) info[k] = 0;
    info[3] = N ? amax : -1; info[4] = N ? smin : 0; info[5] = N ? smax : -1; info[6] = N ? amin : 0; info[7] = flags; info[8] = N;
    return DCARL_OK;
}

int32_t dcarl_ingest_group_packed_f32(const uint64_t* rec, int64_t N, int32_t S, int32_t A, int32_t flags, void* workspace, int32_t* len,
                                      int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int64_t* info, void* stream) {
    if (int rc = check_ingest(nullptr, 0, S, A, workspace, "dcarl_ingest_group_packed")) return rc;
    if (N < 1 || N > 0x7fffffff) return fail(DCARL_EINVAL, "dcarl_ingest_group_packed: N=%lld outside [1,2^31)", (long long)N);
    if (S > 65536) return fail(DCARL_EINVAL, "dcarl_ingest_group_packed: S=%d beyond the direct ingest's 65 536 states", S);
    if (!(flags & DCARL_INGEST_FORCE_DIRECT) || (flags & (DCARL_INGEST_ARRIVAL | DCARL_INGEST_NO_DIRECT)))
        return fail(DCARL_EINVAL, "dcarl_ingest_group_packed: flags=%d must carry DCARL_INGEST_FORCE_DIRECT and neither DCARL_INGEST_ARRIVAL nor "
                                  "DCARL_INGEST_NO_DIRECT", flags);
    if (!rec || (reinterpret_cast<uintptr_t>(rec) & 7u)) return fail(DCARL_EINVAL, "dcarl_ingest_group_packed: rec is NULL or not 8-byte aligned");
    if (!len || !slot_state || !state_slot || !slice_row_off || !info) return fail(DCARL_EINVAL, "dcarl_ingest_group_packed: NULL output");
    stamp_ingest(workspace, N, S, A, flags, 4);
    dcarl::launch_ingest_group_packed(rec, N, S, A, (flags & DCARL_INGEST_SORT_BY_LENGTH) != 0, workspace, len, slot_state, state_slot,
                                      slice_row_off, info, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_ingest_group_packed");
}

int32_t dcarl_ingest_group_f32(const double* data, int64_t N, int32_t S, int32_t A, int32_t flags, void* workspace, int32_t* len,
                               int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int32_t* rec_state, int64_t* info,
                               void* stream) {
    return ingest_group_impl<float>(data, N, S, A, flags, workspace, len, slot_state, state_slot, slice_row_off, rec_state, info, stream);
}
int32_t dcarl_ingest_group_f64(const double* data, int64_t N, int32_t S, int32_t A, int32_t flags, void* workspace, int32_t* len,
                               int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int32_t* rec_state, int64_t* info,
                               void* stream) {
    return ingest_group_impl<double>(data, N, S, A, flags, workspace, len, slot_state, state_slot, slice_row_off, rec_state, info, stream);
}

int32_t dcarl_ingest_pack_f32(int64_t N, int32_t S, int32_t A, int32_t flags, const void* workspace, const int32_t* len,
                              const int32_t* slot_state, const int64_t* slice_row_off, int64_t total_bands, float* R, uint8_t* act,
                              int64_t* rec_elem, int32_t* rec_t, void* stream) {
    return ingest_pack_impl<float>(N, S, A, flags, workspace, len, slot_state, slice_row_off, total_bands, R, act, rec_elem, rec_t, stream);
}
int32_t dcarl_ingest_pack_f64(int64_t N, int32_t S, int32_t A, int32_t flags, const void* workspace, const int32_t* len,
                              const int32_t* slot_state, const int64_t* slice_row_off, int64_t total_bands, double* R, uint8_t* act,
                              int64_t* rec_elem, int32_t* rec_t, void* stream) {
    return ingest_pack_impl<double>(N, S, A, flags, workspace, len, slot_state, slice_row_off, total_bands, R, act, rec_elem, rec_t, stream);
}

int64_t dcarl_slot_order_workspace_bytes(int32_t S) { return S < 1 ? 0 : dcarl::slot_order_workspace_bytes(S); }
int32_t dcarl_slot_order(const int32_t* len_state, int32_t S, int64_t max_len, int32_t flags, void* workspace, int32_t* len,
                         int32_t* slot_state, int32_t* state_slot, int64_t* slice_row_off, int64_t* info, void* stream) {
    if (S < 1 || S > (1 << 26)) return fail(DCARL_EINVAL, "dcarl_slot_order: S=%d outside [1,2^26]", S);
    if (max_len < 0 || max_len > 0x7fffffff) return fail(DCARL_EINVAL, "dcarl_slot_order: max_len outside [0,2^31)");
    if (!len_state || !workspace || !len || !slot_state || !state_slot || !slice_row_off || !info)
        return fail(DCARL_EINVAL, "dcarl_slot_order: NULL argument");
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return fail(DCARL_EINVAL, "dcarl_slot_order: workspace needs 256-byte alignment");
    dcarl::launch_slot_order(len_state, S, max_len, (flags & DCARL_INGEST_SORT_BY_LENGTH) != 0, workspace, len, slot_state, state_slot,
                             slice_row_off, info, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_slot_order");
}

int32_t dcarl_export_records_f32(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* state_slot,
                                 const double* state_value, int32_t S, int64_t records_per_state, int64_t mult, const int32_t* rec_state,
                                 const int64_t* rec_elem, int64_t N, double* out, void* stream) {
    return export_records_impl<float>(R, act, slice_row_off, state_slot, state_value, S, records_per_state, mult, rec_state, rec_elem, N, out, stream);
}
int32_t dcarl_export_records_f64(const double* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* state_slot,
                                 const double* state_value, int32_t S, int64_t records_per_state, int64_t mult, const int32_t* rec_state,
                                 const int64_t* rec_elem, int64_t N, double* out, void* stream) {
    return export_records_impl<double>(R, act, slice_row_off, state_slot, state_value, S, records_per_state, mult, rec_state, rec_elem, N, out, stream);
}
int32_t dcarl_ingest_buckets_f32(const double* data, int64_t N, int32_t S, int32_t A, void* workspace, float* values,
                                 int64_t* seg_off, int64_t* info, void* stream) {
    return ingest_buckets_impl<float>(data, N, S, A, workspace, values, seg_off, info, stream);
}
int32_t dcarl_ingest_buckets_f64(const double* data, int64_t N, int32_t S, int32_t A, void* workspace, double* values,
                                 int64_t* seg_off, int64_t* info, void* stream) {
    return ingest_buckets_impl<double>(data, N, S, A, workspace, values, seg_off, info, stream);
}

int32_t dcarl_sample_state_records(const float* Q, int32_t q_rows, int32_t S, int32_t A, int64_t T, double sigma,
                                   uint64_t seed, uint32_t stream_id, float* R, uint8_t* act, void* stream) {
    if (S < 0 || T < 0) return fail(DCARL_EINVAL, "S or T negative");
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "A=%d outside [1,%d]", A, DCARL_MAX_ACTIONS);
    if (q_rows != 1 && q_rows != S) return fail(DCARL_EINVAL, "q_rows must be 1 or S");
    if (S && T && (!Q || !R || !act)) return fail(DCARL_EINVAL, "dcarl_sample_state_records: NULL argument");
    if (S && T && (!aligned16(R) || (reinterpret_cast<uintptr_t>(act) & 3u)))
        return fail(DCARL_EINVAL, "R needs 16-byte and act 4-byte alignment");
    dcarl::launch_sample_state_records(Q, q_rows, S, A, T, sigma, seed, stream_id, R, act,
                                       static_cast<hipStream_t>(stream));
    return after_launch("dcarl_sample_state_records");
}

int32_t dcarl_sample_state_records_ragged(const float* Q, int32_t q_rows, int32_t S, int32_t A,
                                          const int64_t* slice_row_off, int64_t total_rows, const int32_t* len,
                                          const int32_t* slot_state, const int32_t* n_live, double sigma, uint64_t seed,
                                          uint32_t stream_id, uint32_t state_id_base, const int32_t* state_ids, float* R, uint8_t* act,
                                          void* stream) {
    if (S < 0 || total_rows < 0 || (total_rows & 3)) return fail(DCARL_EINVAL, "S negative or total_rows not a multiple of 4");
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "A=%d outside [1,%d]", A, DCARL_MAX_ACTIONS);
    if (q_rows != 1 && q_rows != S) return fail(DCARL_EINVAL, "q_rows must be 1 or S");
    if (S == 0 || total_rows == 0) return DCARL_OK;
    if (!Q || !R || !act || !slice_row_off || !len) return fail(DCARL_EINVAL, "dcarl_sample_state_records_ragged: NULL argument");
    if (!aligned16(R) || (reinterpret_cast<uintptr_t>(act) & 3u))
        return fail(DCARL_EINVAL, "R needs 16-byte and act 4-byte alignment");
    dcarl::launch_sample_state_records_ragged(Q, q_rows, S, A, slice_row_off, total_rows, len, slot_state, n_live, sigma, seed,
                                              stream_id, state_id_base, state_ids, R, act, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_sample_state_records_ragged");
}

int32_t dcarl_sample_buckets(const float* Q, int32_t q_rows, int32_t S, int32_t A, const int64_t* seg_off, int64_t n_dense,
                             double sigma, uint64_t seed, uint32_t stream_id, float* values, void* stream) {
    if (S < 0 || (!seg_off && n_dense < 0)) return fail(DCARL_EINVAL, "S or n_dense negative");
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "A=%d outside [1,%d]", A, DCARL_MAX_ACTIONS);
    if (q_rows != 1 && q_rows != S) return fail(DCARL_EINVAL, "q_rows must be 1 or S");
    if (S == 0) return DCARL_OK;
    if (!Q || !values) return fail(DCARL_EINVAL, "dcarl_sample_buckets: NULL argument");
    if (!aligned16(values)) return fail(DCARL_EINVAL, "values needs 16-byte alignment");
    dcarl::launch_sample_buckets(Q, q_rows, S, A, seg_off, n_dense, sigma, seed, stream_id, values,
                                 static_cast<hipStream_t>(stream));
    return after_launch("dcarl_sample_buckets");
}

int32_t dcarl_sample_pairs(const float* Q, int32_t S, int32_t A, int64_t N, double sigma, uint64_t seed,
                           uint64_t offset, uint32_t stream_id, int32_t* idx, int32_t* act, float* R, float* z_visit,
                           void* stream) {
    if (S < 1 || N < 0) return fail(DCARL_EINVAL, "S < 1 or N negative");
    if (A < 1 || A > DCARL_MAX_ACTIONS) return fail(DCARL_EINVAL, "A=%d outside [1,%d]", A, DCARL_MAX_ACTIONS);
    if (N && (!Q || !idx || !act || !R)) return fail(DCARL_EINVAL, "dcarl_sample_pairs: NULL argument");
    dcarl::launch_sample_pairs(Q, S, A, N, sigma, seed, offset, stream_id, idx, act, R, z_visit,
                               static_cast<hipStream_t>(stream));
    return after_launch("dcarl_sample_pairs");
}

int32_t dcarl_visit_index_f64(const double* z_visit, int64_t M, int32_t S, int32_t* idx, void* stream) {
    if (M < 0 || S < 1) return fail(DCARL_EINVAL, "M negative or S < 1");
    if (M && (!z_visit || !idx)) return fail(DCARL_EINVAL, "dcarl_visit_index_f64: NULL argument");
    dcarl::launch_visit_index(z_visit, M, S, idx, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_visit_index_f64");
}

int32_t dcarl_visit_floor_f64(const double* z_visit, int64_t M, int32_t S, int64_t* out, void* stream) {
    if (M < 0 || S < 1) return fail(DCARL_EINVAL, "M negative or S < 1");
    if (M && (!z_visit || !out)) return fail(DCARL_EINVAL, "dcarl_visit_floor_f64: NULL argument");
    dcarl::launch_visit_floor(z_visit, M, S, out, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_visit_floor_f64");
}

int32_t dcarl_state_manual_f64(const double* u, const int64_t* kept_rank, const int32_t* r, int64_t M, int32_t* out, void* stream) {
    if (M < 0) return fail(DCARL_EINVAL, "M negative");
    if (M && (!u || !kept_rank || !r || !out)) return fail(DCARL_EINVAL, "dcarl_state_manual_f64: NULL argument");
    dcarl::launch_state_manual(u, kept_rank, r, M, out, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_state_manual_f64");
}

int32_t dcarl_sample_from_noise_f64(const int32_t* idx, const int64_t* kept_rank, int64_t M, const double* states,
                                    const double* Q64, int32_t S, int32_t A, const int32_t* acts,
                                    const double* z_reward, double sigma, double* out_rows, void* stream) {
    if (M < 0 || S < 1 || A < 1) return fail(DCARL_EINVAL, "M negative or S/A < 1");
    if (M && (!idx || !kept_rank || !states || !Q64 || !acts || !z_reward || !out_rows))
        return fail(DCARL_EINVAL, "dcarl_sample_from_noise_f64: NULL argument");
    if (M && (reinterpret_cast<uintptr_t>(out_rows) & 31u)) return fail(DCARL_EINVAL, "out_rows needs 32-byte alignment");
    dcarl::launch_sample_from_noise(idx, kept_rank, M, states, Q64, S, A, acts, z_reward, sigma, out_rows,
                                    static_cast<hipStream_t>(stream));
    return after_launch("dcarl_sample_from_noise_f64");
}

void dcarl_rls_default_params(dcarl_rls_params_t* p) {
    if (!p) return;
    p->visited_times_thres = 30;
    p->min_rl_visits = 5;
    p->rule_mean_gate = -0.1;
    p->confidence_thres = 0.5;
}

int64_t dcarl_rls_workspace_bytes(int64_t N, int32_t Q) { return (N < 0 || Q < 0) ? 0 : dcarl::rls_workspace_bytes(N, Q); }

int32_t dcarl_rls_neighbour_stats_f64(const double* states, const double* values, int64_t N, const double* half_width,
                                      const double* queries, int32_t Q, void* workspace, int64_t* count, double* mean,
                                      double* var, void* stream) {
    if (N < 0 || Q < 0) return fail(DCARL_EINVAL, "dcarl_rls_neighbour_stats: N=%lld / Q=%d negative", (long long)N, Q);
    if (Q == 0) return DCARL_OK;
    if (!queries || !count || !mean || !var || !workspace || !half_width || (N && (!states || !values)))
        return fail(DCARL_EINVAL, "dcarl_rls_neighbour_stats: NULL argument");
    if (!aligned16(workspace)) return fail(DCARL_EINVAL, "workspace needs 16-byte alignment");
    dcarl::launch_rls_stats(states, values, N, half_width, queries, Q, workspace, count, mean, var,
                            static_cast<hipStream_t>(stream));
    return after_launch("dcarl_rls_neighbour_stats");
}

int32_t dcarl_rls_decide(const int64_t* count, const double* mean, const double* var, int32_t B, int32_t n_cand,
                         const dcarl_rls_params_t* params, int32_t* action, void* stream) {
    if (!params) return fail(DCARL_EINVAL, "params is NULL");
    if (B < 0 || n_cand < 0 || n_cand >= DCARL_MAX_ACTIONS)
        return fail(DCARL_EINVAL, "dcarl_rls_decide: B=%d / n_cand=%d out of range", B, n_cand);
    if (B == 0) return DCARL_OK;
    if (!count || !mean || !var || !action) return fail(DCARL_EINVAL, "dcarl_rls_decide: NULL argument");
    dcarl::launch_rls_decide(count, mean, var, B, n_cand, *params, action, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_rls_decide");
}

int32_t dcarl_rls_gate_train(const int64_t* count_rule, const double* mean_rule, const double* explore, const int32_t* rl_action,
                             int32_t B, int32_t visited_times_thres, int32_t* action, uint8_t* use_rule, void* stream) {
    if (B < 0) return fail(DCARL_EINVAL, "dcarl_rls_gate_train: B negative");
    if (B == 0) return DCARL_OK;
    if (!count_rule || !mean_rule || !explore || (!action && !use_rule) || (action && !rl_action))
        return fail(DCARL_EINVAL, "dcarl_rls_gate_train: NULL argument");
    dcarl::launch_rls_gate_train(count_rule, mean_rule, explore, rl_action, B, visited_times_thres, action, use_rule,
                                 static_cast<hipStream_t>(stream));
    return after_launch("dcarl_rls_gate_train");
}

void dcarl_frenet_default_grid(dcarl_frenet_grid_t* g) {
    if (!g) return;
    memset(g, 0, sizeof(*g));
    // JTP:14-40: MAX_LEFT_WIDTH -4, MAX_RIGHT_WIDTH 4, D_ROAD_W 2, DT 0.3, MAXT 4.2, MINT 4.0, TARGET_SPEED 30/3.6,
    // D_T_S 15/3.6, N_S_SAMPLE 1, KJ 0.1, KT 0.1, KD 1, KLAT 1, KLON 1
    g->n_d = 5; g->n_T = 1; g->n_v = 2; g->nt_max = 14;
    for (int i = 0; i < 5; ++i) g->d[i] = -4.0 + 2.0 * i;
    g->T[0] = 4.0; g->nt[0] = 14;
    g->target_speed = 30.0 / 3.6;
    g->tv[0] = g->target_speed - 15.0 / 3.6 * 1; g->tv[1] = g->tv[0] + 15.0 / 3.6;
    g->dt = 0.3;
    g->kj = 0.1; g->kt = 0.1; g->kd = 1.0; g->klat = 1.0; g->klon = 1.0;
}

int32_t dcarl_frenet_candidates_f64(const double* start, int64_t B, const dcarl_frenet_grid_t* grid, double* traj,
                                    double* cost, void* stream) {
    if (!grid) return fail(DCARL_EINVAL, "grid is NULL");
    if (B < 0) return fail(DCARL_EINVAL, "B negative");
    if (grid->n_d < 0 || grid->n_d > 16 || grid->n_T < 0 || grid->n_T > 8 || grid->n_v < 0 || grid->n_v > 8 ||
        grid->nt_max < 0)
        return fail(DCARL_EINVAL, "dcarl_frenet_candidates: grid sizes out of range (n_d<=16, n_T<=8, n_v<=8)");
    for (int i = 0; i < grid->n_T; ++i)
        if (grid->nt[i] < 1 || grid->nt[i] > grid->nt_max || !(grid->T[i] > 0.0))
            return fail(DCARL_EINVAL, "dcarl_frenet_candidates: nt[%d]=%d outside [1,nt_max] or T not positive", i, grid->nt[i]);
    if (B == 0) return DCARL_OK;
    if (!start) return fail(DCARL_EINVAL, "start is NULL");
    dcarl::launch_frenet(start, B, *grid, traj, cost, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_frenet_candidates");
}

void dcarl_frenet_default_limits(dcarl_frenet_limits_t* l) {
    if (!l) return;
    l->max_speed = 50.0 / 3.6;
    l->max_accel = 10.0;
    l->max_curvature = 500.0;
    l->check_radius = 1.0;
    l->move_gap = 1.0;
    l->n_predict = 15;
}

static int check_grid(const dcarl_frenet_grid_t* grid, const char* who) {
    if (!grid) return fail(DCARL_EINVAL, "%s: grid is NULL", who);
    if (grid->n_d < 0 || grid->n_d > 16 || grid->n_T < 0 || grid->n_T > 8 || grid->n_v < 0 || grid->n_v > 8 || grid->nt_max < 0)
        return fail(DCARL_EINVAL, "%s: grid sizes out of range (n_d<=16, n_T<=8, n_v<=8)", who);
    for (int i = 0; i < grid->n_T; ++i)
        if (grid->nt[i] < 1 || grid->nt[i] > grid->nt_max || !(grid->T[i] > 0.0))
            return fail(DCARL_EINVAL, "%s: nt[%d]=%d outside [1,nt_max] or T not positive", who, i, grid->nt[i]);
    return DCARL_OK;
}

int32_t dcarl_frenet_global_paths_f64(const double* traj, int64_t B, const dcarl_frenet_grid_t* grid, const double* knots,
                                      const double* segments, int32_t n_knots, double* glob, int32_t* path_len, void* stream) {
    if (int rc = check_grid(grid, "dcarl_frenet_global_paths")) return rc;
    if (B < 0 || n_knots < 2) return fail(DCARL_EINVAL, "dcarl_frenet_global_paths: B=%lld negative or n_knots=%d < 2", (long long)B, n_knots);
    if (B == 0) return DCARL_OK;
    if (!traj || !knots || !segments || !glob || !path_len) return fail(DCARL_EINVAL, "dcarl_frenet_global_paths: NULL argument");
    dcarl::launch_frenet_global(traj, B, *grid, knots, segments, n_knots, glob, path_len, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_frenet_global_paths");
}

int32_t dcarl_frenet_select(const double* traj, const double* glob, const int32_t* path_len, const double* cost,
                            const double* obstacles, int32_t n_obs, int64_t B, const dcarl_frenet_grid_t* grid,
                            const dcarl_frenet_limits_t* limits, int32_t* choice, uint8_t* ok, void* stream) {
    if (int rc = check_grid(grid, "dcarl_frenet_select")) return rc;
    if (!limits) return fail(DCARL_EINVAL, "limits is NULL");
    if (B < 0 || n_obs < 0) return fail(DCARL_EINVAL, "dcarl_frenet_select: B / n_obs negative");
    if (grid->n_d * grid->n_T * grid->n_v > 32) return fail(DCARL_EINVAL, "dcarl_frenet_select: more than 32 candidates");
    if (B == 0) return DCARL_OK;
    if (!traj || !glob || !path_len || !cost || !choice || (n_obs && !obstacles))
        return fail(DCARL_EINVAL, "dcarl_frenet_select: NULL argument");
    dcarl::launch_frenet_select(traj, glob, path_len, cost, obstacles, n_obs, B, *grid, *limits, choice, ok,
                                static_cast<hipStream_t>(stream));
    return after_launch("dcarl_frenet_select");
}

void dcarl_gamma_powers(double gamma, int32_t horizon, double* out) {
    if (!out) return;
    for (int k = 0; k < horizon; ++k) out[k] = std::pow(gamma, (double)k);    // RLS:207 self.gamma**len(buffer)
}

int32_t dcarl_episode_returns_f64(const double* vx, const double* vy, const uint8_t* flags, const int64_t* ep_off, int64_t E,
                                  double* step_reward, double* episode_reward, double* ave_speed, void* stream) {
    if (E < 0) return fail(DCARL_EINVAL, "dcarl_episode_returns: E negative");
    if (E == 0) return DCARL_OK;
    if (!vx || !vy || !flags || !ep_off || !episode_reward) return fail(DCARL_EINVAL, "dcarl_episode_returns: NULL argument");
    dcarl::launch_episode_returns(vx, vy, flags, ep_off, E, step_reward, episode_reward, ave_speed,
                                  static_cast<hipStream_t>(stream));
    return after_launch("dcarl_episode_returns");
}

int32_t dcarl_nstep_backup_f64(const double* rew, const int64_t* ep_off, const uint8_t* ep_done, int64_t E,
                               const double* gamma_pow, int32_t horizon, double* value, uint8_t* recorded, void* stream) {
    if (E < 0 || horizon < 0) return fail(DCARL_EINVAL, "dcarl_nstep_backup: E or horizon negative");
    if (E == 0) return DCARL_OK;
    if (!rew || !ep_off || !ep_done || !value || (horizon && !gamma_pow))
        return fail(DCARL_EINVAL, "dcarl_nstep_backup: NULL argument");
    dcarl::launch_nstep_backup(rew, ep_off, ep_done, E, gamma_pow, horizon, value, recorded, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_nstep_backup");
}

int32_t dcarl_state_cells_f64(const double* obs, int64_t N, int32_t D, const double* cell_width, int32_t* cells, uint64_t* hash,
                              void* stream) {
    if (N < 0 || D < 1 || D > 64) return fail(DCARL_EINVAL, "dcarl_state_cells: N=%lld negative or D=%d outside [1,64]", (long long)N, D);
    if (N == 0) return DCARL_OK;
    if (!obs || !cell_width || !cells) return fail(DCARL_EINVAL, "dcarl_state_cells: NULL argument");
    if (hash && (D % 4 || !aligned16(obs) || !aligned16(cells)))
        return fail(DCARL_EINVAL, "dcarl_state_cells: the hash output needs D %% 4 == 0 and 16-byte aligned obs / cells");
    dcarl::launch_state_cells(obs, N, D, cell_width, cells, reinterpret_cast<unsigned long long*>(hash), static_cast<hipStream_t>(stream));
    return after_launch("dcarl_state_cells");
}

int32_t dcarl_state_ids(const int32_t* cells, const uint64_t* hash, int64_t N, int32_t D, int64_t max_states, void* workspace,
                        int32_t* ids, int64_t* out, void* stream) {
    if (N < 0 || N > 0x7fffffff || D < 1 || D > 64)
        return fail(DCARL_EINVAL, "dcarl_state_ids: N=%lld outside [0,2^31) or D=%d outside [1,64]", (long long)N, D);
    if (max_states < 0) return fail(DCARL_EINVAL, "dcarl_state_ids: max_states negative");
    if (!out) return fail(DCARL_EINVAL, "dcarl_state_ids: out is NULL");
    if (N == 0) return DCARL_OK;
    if (!cells || !workspace || !ids) return fail(DCARL_EINVAL, "dcarl_state_ids: NULL argument");
    if (!aligned16(workspace)) return fail(DCARL_EINVAL, "workspace needs 16-byte alignment");
    dcarl::launch_state_ids(cells, reinterpret_cast<const unsigned long long*>(hash), N, D, max_states, workspace, ids, out,
                            static_cast<hipStream_t>(stream));
    return after_launch("dcarl_state_ids");
}

int32_t dcarl_index_states_f64(const double* obs, int64_t N, int32_t D, const double* cell_width, int64_t max_states, void* workspace,
                               int32_t* cells, int32_t* ids, int64_t* out, void* stream) {
    if (N < 0 || N > 0x7fffffff || D < 4 || D > 64 || D % 4)
        return fail(DCARL_EINVAL, "dcarl_index_states: N=%lld outside [0,2^31) or D=%d not a multiple of 4 in [4,64]", (long long)N, D);
    if (max_states < 0) return fail(DCARL_EINVAL, "dcarl_index_states: max_states negative");
    if (!out) return fail(DCARL_EINVAL, "dcarl_index_states: out is NULL");
    if (N == 0) return DCARL_OK;
    if (!obs || !cell_width || !cells || !workspace || !ids) return fail(DCARL_EINVAL, "dcarl_index_states: NULL argument");
    if (!aligned16(workspace) || !aligned16(obs) || !aligned16(cells)) return fail(DCARL_EINVAL, "dcarl_index_states: obs, cells and workspace need 16-byte alignment");
    dcarl::launch_index_states(obs, N, D, cell_width, max_states, workspace, cells, ids, out, static_cast<hipStream_t>(stream));
    return after_launch("dcarl_index_states");
}

}  // extern "C"
