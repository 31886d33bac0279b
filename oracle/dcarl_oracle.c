/* CPU oracle for the DCARL confidence hot path — TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C float64 restatement of the reference algorithm, used (a) by tests as the checker at
 * sizes the NumPy oracle (oracle/dcarl_oracle.py) is too slow for, (b) by bench.py's
 * cpu_baseline leg ("port").  It is itself pinned against the NumPy oracle and the goldens
 * generated from the reference (tests/test_oracle_c.py).
 *
 * Reference citations (relative to the reference root):
 *   S1 = Simulation_testing/Simulation_1/test_DCARL.py, S2 = Simulation_testing/Simulation_2/test_DCARL.py,
 *   DS = Simulation_testing/Simulation_Data_Collection/Data_Sampling/data_sampling.py
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC -o oracle/_build/libdcarl_oracle.so oracle/dcarl_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t rule_act;   /* S1:43 */
    int32_t n_thres;    /* S1:45 */
    double alpha;       /* S1:10 */
    double scale;       /* S1:10 */
    double cap;         /* S1:12 */
    double init_rule;   /* S1:52 */
    double init_other;  /* S1:51 */
} orc_params_t;

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* scale*sqrt(log(1/alpha)/2/n): operation order of S1:12 / S1:16 / S1:24 */
static double halfwidth(double n, const orc_params_t* p) { return p->scale * sqrt(log(1 / p->alpha) / 2 / n); }

/* V from sufficient statistics; algebra of S1:10-24 with var = q/n - mean^2 */
static double value_from_sums(int64_t n, double s, double q, int is_rule, const orc_params_t* p) {
    double dn = (double)n, mean = s / dn, hw = halfwidth(dn, p);
    if (is_rule) { double v = mean + hw; return v < p->cap ? v : p->cap; }     /* S1:12 */
    double var = q / dn - mean * mean; if (var < 0) var = 0;
    double sigma = sqrt(var);
    double lb = mean - hw;                                                       /* S1:16 */
    double ci = s / dn / (dn + 1) - 4 * sigma / (dn + 1) + s / (dn + 1) - halfwidth(dn + 1, p); /* S1:24 */
    return lb < ci ? lb : ci;                                                    /* S1:90 */
}

/* V from the raw bucket with two-pass mean/std like np.mean/np.std (sequential sums) */
static double value_from_bucket(const double* x, int64_t n, int is_rule, const orc_params_t* p) {
    double s = 0; for (int64_t i = 0; i < n; ++i) s += x[i];
    double dn = (double)n, mean = s / dn, hw = halfwidth(dn, p);
    if (is_rule) { double v = mean + hw; return v < p->cap ? v : p->cap; }
    double m2 = 0; for (int64_t i = 0; i < n; ++i) { double d = x[i] - mean; m2 += d * d; }
    double sigma = sqrt(m2 / dn);
    double lb = mean - hw;
    double ci = s / dn / (dn + 1) - 4 * sigma / (dn + 1) + s / (dn + 1) - halfwidth(dn + 1, p);
    return lb < ci ? lb : ci;
}

static int argmax_first(const double* v, int A) {       /* np.argmax: first maximum (S1:94) */
    int b = 0; for (int a = 1; a < A; ++a) if (v[a] > v[b]) b = a; return b;
}

#define LOADR(ptr, isf32, i) ((isf32) ? (double)((const float*)(ptr))[i] : ((const double*)(ptr))[i])

/* Online loop S1:73-99 on records grouped by state (arrival order kept inside a state).
 * R is f32 (r_is_f32) or f64; state_off[S+1] are record offsets.  O(1) per record.
 * Outputs (any may be NULL): step_val f64[N], step_act u8[N], act_step i32[S], V f64[S*A], cnt i32[S*A]. */
void orc_trace(const void* R, int r_is_f32, const uint8_t* act, const int64_t* state_off, int32_t S, int32_t A,
               const orc_params_t* p, double* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
               int32_t* n_out, float* vmax, int32_t* amax) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int32_t s = 0; s < S; ++s) {
        double V[256], sm[256], sq[256]; int64_t cnt[256];
        for (int a = 0; a < A; ++a) { V[a] = p->init_other; sm[a] = sq[a] = 0; cnt[a] = 0; }   /* S1:50-53 */
        V[p->rule_act] = p->init_rule;
        int32_t latch = -1;
        int best = argmax_first(V, A);
        for (int64_t k = state_off[s]; k < state_off[s + 1]; ++k) {
            int a = act[k]; double r = LOADR(R, r_is_f32, k);
            cnt[a]++; sm[a] += r; sq[a] += r * r;                                /* S1:80 */
            if (cnt[a] > p->n_thres)                                             /* S1:86 */
                V[a] = value_from_sums(cnt[a], sm[a], sq[a], a == p->rule_act, p);
            best = argmax_first(V, A);                                           /* S1:93-94 */
            if (step_val) step_val[k] = V[best];
            if (step_act) step_act[k] = (uint8_t)best;
            if (latch == -1 && best != p->rule_act) latch = (int32_t)(k - state_off[s] + 1);  /* S1:98-99 */
        }
        if (act_step) act_step[s] = latch;
        if (V_out) for (int a = 0; a < A; ++a) V_out[(int64_t)s * A + a] = V[a];
        if (n_out) for (int a = 0; a < A; ++a) n_out[(int64_t)s * A + a] = (int32_t)cnt[a];
        if (vmax) vmax[s] = (float)V[best];
        if (amax) amax[s] = best;
    }
}

/* Same loop with the reference's per-record O(n) structure: the bucket is kept as a growing array and
 * mean/std are recomputed from scratch for every record (S1:86-90 calls np.mean/np.sum/np.std on the whole
 * bucket).  Used only to time "the reference's algorithmic structure" on the host. */
void orc_trace_recompute(const void* R, int r_is_f32, const uint8_t* act, const int64_t* state_off, int32_t S,
                         int32_t A, const orc_params_t* p, double* step_val, uint8_t* step_act,
                         int32_t* act_step) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t s = 0; s < S; ++s) {
        int64_t len = state_off[s + 1] - state_off[s];
        double* bucket = (double*)malloc(sizeof(double) * (size_t)(len > 0 ? len : 1) * (size_t)A);
        double V[256]; int64_t cnt[256];
        for (int a = 0; a < A; ++a) { V[a] = p->init_other; cnt[a] = 0; }
        V[p->rule_act] = p->init_rule;
        int32_t latch = -1;
        for (int64_t k = state_off[s]; k < state_off[s + 1]; ++k) {
            int a = act[k];
            bucket[(int64_t)a * len + cnt[a]++] = LOADR(R, r_is_f32, k);
            if (cnt[a] > p->n_thres)
                V[a] = value_from_bucket(bucket + (int64_t)a * len, cnt[a], a == p->rule_act, p);
            int best = argmax_first(V, A);
            if (step_val) step_val[k] = V[best];
            if (step_act) step_act[k] = (uint8_t)best;
            if (latch == -1 && best != p->rule_act) latch = (int32_t)(k - state_off[s] + 1);
        }
        if (act_step) act_step[s] = latch;
        free(bucket);
    }
}

/* Final-state evaluation on samples sorted by (state, action): seg_off[S*A+1].  Two-pass per bucket. */
void orc_bounds_csr(const void* values, int v_is_f32, const int64_t* seg_off, int32_t S, int32_t A,
                    const orc_params_t* p, double* V_out, int32_t* n_out, float* vmax, int32_t* amax) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int32_t s = 0; s < S; ++s) {
        double V[256];
        for (int a = 0; a < A; ++a) {
            int64_t b = seg_off[(int64_t)s * A + a], e = seg_off[(int64_t)s * A + a + 1], n = e - b;
            V[a] = a == p->rule_act ? p->init_rule : p->init_other;
            if (n > p->n_thres) {
                double sm = 0; for (int64_t i = b; i < e; ++i) sm += LOADR(values, v_is_f32, i);
                double mean = sm / (double)n, m2 = 0;
                for (int64_t i = b; i < e; ++i) { double d = LOADR(values, v_is_f32, i) - mean; m2 += d * d; }
                double hw = halfwidth((double)n, p);
                if (a == p->rule_act) { double v = mean + hw; V[a] = v < p->cap ? v : p->cap; }
                else {
                    double dn = (double)n, sigma = sqrt(m2 / dn), lb = mean - hw;
                    double ci = sm / dn / (dn + 1) - 4 * sigma / (dn + 1) + sm / (dn + 1) - halfwidth(dn + 1, p);
                    V[a] = lb < ci ? lb : ci;
                }
            }
            if (n_out) n_out[(int64_t)s * A + a] = (int32_t)n;
            if (V_out) V_out[(int64_t)s * A + a] = V[a];
        }
        int best = argmax_first(V, A);
        if (vmax) vmax[s] = (float)V[best];
        if (amax) amax[s] = best;
    }
}

/* S2:99-105: overall[k] = sum over activated states of (current max V + 0.9), evaluated after every record in
 * ARRIVAL order.  rec_state[k] is the state of arrival k, rec_pos[k] its position in the grouped arrays. */
void orc_overall(const double* step_val, const int32_t* act_step, const int64_t* state_off, const int32_t* rec_state,
                 const int64_t* rec_pos, int64_t N, int32_t S, double* overall) {
    double* cur = (double*)calloc((size_t)S, sizeof(double));
    uint8_t* on = (uint8_t*)calloc((size_t)S, 1);
    for (int64_t k = 0; k < N; ++k) {
        int32_t s = rec_state[k]; int64_t pos = rec_pos[k];
        cur[s] = step_val[pos];
        if (act_step[s] != -1 && (pos - state_off[s] + 1) >= act_step[s]) on[s] = 1;
        double tot = 0;
        for (int32_t i = 0; i < S; ++i) if (on[i]) tot = tot + cur[i] + 0.9;   /* activation_value == -1 (S2:59) */
        overall[k] = tot;
    }
    free(cur); free(on);
}

/* ---------------- Philox-4x32-10 (Salmon et al., SC'11) + Box-Muller ---------------- */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)c[0] * 0xD2511F53u, p1 = (uint64_t)c[2] * 0xCD9E8D57u;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    memcpy(out, ctr, 16); philox(out, key[0], key[1]);
}
static double unit_open(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

/* dcarl_sample_state_records restated: record t of state s <- Philox(ctr=(t,s,stream,0), key=seed). */
void orc_sample_state_records(const double* Q, int32_t S, int32_t A, int64_t T, uint64_t seed, uint32_t stream,
                              double sigma, uint8_t* act, double* R) {
#pragma omp parallel for schedule(static)
    for (int32_t s = 0; s < S; ++s)
        for (int64_t t = 0; t < T; ++t) {
            uint32_t c[4] = {(uint32_t)t, (uint32_t)s, stream, 0};
            philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            int a = (int)(((uint64_t)c[0] * (uint64_t)A) >> 32);
            double z = sqrt(-2.0 * log(unit_open(c[1]))) * cos(2.0 * M_PI * unit_open(c[2]));
            act[(int64_t)s * T + t] = (uint8_t)a;
            R[(int64_t)s * T + t] = Q[(int64_t)s * A + a] + sigma * z;
        }
}

/* dcarl_sample_pairs restated (DS:45-55 semantics): draw g = offset+i is draw k = g%4 of group G = g/4, which owns the
 * 12 words of the Philox blocks with counters 3G, 3G+1, 3G+2; draw k uses words 3k (action), 3k+1, 3k+2 (Box-Muller).
 * idx = -1 when the visit falls outside [0,S) (DS:50-51). */
/* z_visit (nullable): the float64 visit normal of each draw (what the index is the floor of) */
void orc_sample_pairs(const double* Q, int32_t S, int32_t A, int64_t N, uint64_t seed, uint64_t offset,
                      uint32_t stream, double sigma, int32_t* idx, int32_t* act, double* R, double* z_visit) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        uint64_t g = offset + (uint64_t)i, G = g >> 2;
        int k = (int)(g & 3);
        uint32_t w[12];
        for (int c = 0; c < 3; ++c) {
            uint64_t ctr = 3 * G + (uint64_t)c;
            uint32_t x[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), stream, 0};
            philox(x, (uint32_t)seed, (uint32_t)(seed >> 32));
            memcpy(w + 4 * c, x, 16);
        }
        int a = (int)(((uint64_t)w[3 * k] * (uint64_t)A) >> 32);
        double rad = sqrt(-2.0 * log(unit_open(w[3 * k + 1]))), th = 2.0 * M_PI * unit_open(w[3 * k + 2]);
        double zr = rad * cos(th), zs = rad * sin(th);
        double v = floor((3.0 + 1.0 * zs) / 6 * S);                              /* DS:14-15 */
        int32_t si = (v < 0 || v >= S) ? -1 : (int32_t)v;
        idx[i] = si; act[i] = a;
        if (z_visit) z_visit[i] = zs;
        R[i] = si < 0 ? 0.0 : Q[(int64_t)si * A + a] + sigma * zr;               /* DS:9 */
    }
}

/* dcarl_sample_state_records_ragged restated: len[s] records for state s, record t <- Philox(ctr=(t,s,stream,0)),
 * action uniform over the first n_live[s] candidates (NULL: A).  Output state-major (state_off = prefix sum of len). */
void orc_sample_state_records_ragged(const double* Q, int32_t q_rows, int32_t S, int32_t A, const int64_t* state_off,
                                     const int32_t* n_live, uint64_t seed, uint32_t stream, double sigma, uint8_t* act,
                                     double* R) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int32_t s = 0; s < S; ++s) {
        const double* q = Q + (q_rows == 1 ? 0 : (int64_t)s * A);
        uint32_t nl = n_live ? (uint32_t)n_live[s] : (uint32_t)A;
        for (int64_t t = 0; t < state_off[s + 1] - state_off[s]; ++t) {
            uint32_t c[4] = {(uint32_t)t, (uint32_t)s, stream, 0};
            philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            int a = (int)(((uint64_t)c[0] * (uint64_t)nl) >> 32);
            double z = sqrt(-2.0 * log(unit_open(c[1]))) * cos(2.0 * M_PI * unit_open(c[2]));
            act[state_off[s] + t] = (uint8_t)a;
            R[state_off[s] + t] = q[a] + sigma * z;
        }
    }
}

/* dcarl_sample_buckets restated: sample i of the flat array is normal i%4 of Philox(ctr=(lo(i/4),hi(i/4),stream,1)):
 * (cos, sin) of words (x0,x1) for i%4 = 0,1 and of (x2,x3) for i%4 = 2,3; value = Q[bucket(i)] + sigma*z  (DS:9). */
void orc_sample_buckets(const double* Q, int32_t q_rows, int32_t S, int32_t A, const int64_t* seg_off, uint64_t seed,
                        uint32_t stream, double sigma, double* values) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t b = 0; b < (int64_t)S * A; ++b) {
        double q = Q[q_rows == 1 ? b % A : b];
        for (int64_t i = seg_off[b]; i < seg_off[b + 1]; ++i) {
            uint64_t v = (uint64_t)i >> 2;
            uint32_t c[4] = {(uint32_t)v, (uint32_t)(v >> 32), stream, 1};
            philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            int k = (int)(i & 3);
            double rad = sqrt(-2.0 * log(unit_open(c[k & 2]))), th = 2.0 * M_PI * unit_open(c[(k & 2) + 1]);
            values[i] = q + sigma * rad * ((k & 1) ? sin(th) : cos(th));
        }
    }
}
