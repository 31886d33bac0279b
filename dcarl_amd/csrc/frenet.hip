// Candidate generation in the Frenet frame (the step that PRODUCES the actions the
// confidence path ranks).  Replaces JunctionTrajectoryPlanner.calc_frenet_paths and the two polynomial classes of
// Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Agent/zzz/JunctionTrajectoryPlanner.py (JTP:292-340,
// JTP:397-491): for every start state, every lateral offset d_i x horizon T_i x target speed v_i a quintic lateral
// and a quartic longitudinal polynomial, sampled every DT, plus the three costs.
// One thread = one (start state, candidate, time step); the two polynomial solves are done once per (state, candidate)
// into LDS, the samples are staged through LDS so that they leave as contiguous 16-byte
// vectors -- the kernel is HBM-write bound.  The costs are a second, tiny kernel.
#include <algorithm>

#include "common.h"

namespace dcarl {

struct Poly { double a0, a1, a2, a3, a4, a5; };

// JTP:399-423: end position, velocity and acceleration given; the 3x3 system solved in closed form
__device__ __forceinline__ Poly quintic(double xs, double vxs, double axs, double xe, double vxe, double axe, double T) {
    Poly p;
    p.a0 = xs; p.a1 = vxs; p.a2 = axs / 2.0;
    const double T2 = T * T, T3 = T2 * T, T4 = T3 * T, T5 = T4 * T;
    const double b0 = xe - p.a0 - p.a1 * T - p.a2 * T2, b1 = vxe - p.a1 - 2 * p.a2 * T, b2 = axe - 2 * p.a2;
    p.a3 = 10.0 * b0 / T3 - 4.0 * b1 / T2 + b2 / (2.0 * T);
    p.a4 = -15.0 * b0 / T4 + 7.0 * b1 / T3 - b2 / T2;
    p.a5 = 6.0 * b0 / T5 - 3.0 * b1 / T4 + b2 / (2.0 * T3);
    return p;
}
// JTP:449-469: end velocity and acceleration given (velocity keeping); 2x2 system in closed form, a5 = 0
__device__ __forceinline__ Poly quartic(double xs, double vxs, double axs, double vxe, double axe, double T) {
    Poly p;
    p.a0 = xs; p.a1 = vxs; p.a2 = axs / 2.0; p.a5 = 0.0;
    const double T2 = T * T, T3 = T2 * T;
    const double b0 = vxe - p.a1 - 2 * p.a2 * T, b1 = axe - 2 * p.a2;
    p.a3 = b0 / T2 - b1 / (3.0 * T);
    p.a4 = -b0 / (2.0 * T3) + b1 / (4.0 * T2);
    return p;
}
// JTP:425-446 / 471-491
__device__ __forceinline__ double poly0(const Poly& p, double t) {
    const double t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
    return p.a0 + p.a1 * t + p.a2 * t2 + p.a3 * t3 + p.a4 * t4 + p.a5 * t5;
}
__device__ __forceinline__ double poly1(const Poly& p, double t) {
    const double t2 = t * t, t3 = t2 * t, t4 = t3 * t;
    return p.a1 + 2 * p.a2 * t + 3 * p.a3 * t2 + 4 * p.a4 * t3 + 5 * p.a5 * t4;
}
__device__ __forceinline__ double poly2(const Poly& p, double t) {
    const double t2 = t * t, t3 = t2 * t;
    return 2 * p.a2 + 6 * p.a3 * t + 12 * p.a4 * t2 + 20 * p.a5 * t3;
}
__device__ __forceinline__ double poly3(const Poly& p, double t) { return 6 * p.a3 + 24 * p.a4 * t + 60 * p.a5 * t * t; }

struct Candidate { double di, Ti, tv; int nt; };
__device__ __forceinline__ Candidate candidate(const dcarl_frenet_grid_t& g, int c) {   // JTP:304-318 loop order: d, T, v
    const int iv = c % g.n_v, iT = (c / g.n_v) % g.n_T, id = c / (g.n_v * g.n_T);
    return Candidate{g.d[id], g.T[iT], g.tv[iv], g.nt[iT]};
}

// JTP:326-336: cd = KJ*sum(d_ddd^2) + KT*Ti + KD*d[-1]^2, cv = KJ*sum(s_ddd^2) + KT*Ti + KD*(target - s_d[-1])^2, cf
__device__ __forceinline__ void write_costs(const dcarl_frenet_grid_t& g, const Candidate& k, const Poly& lat, const Poly& lon,
                                            double* __restrict__ out) {
    double jp = 0.0, js = 0.0;
    for (int it = 0; it < k.nt; ++it) {
        const double t = 0.0 + it * g.dt, a = poly3(lat, t), b = poly3(lon, t);
        jp += a * a;
        js += b * b;
    }
    const double t_last = 0.0 + (k.nt - 1) * g.dt, d_last = poly0(lat, t_last), ds = g.target_speed - poly1(lon, t_last);
    const double cd = g.kj * jp + g.kt * k.Ti + g.kd * d_last * d_last;
    const double cv = g.kj * js + g.kt * k.Ti + g.kd * ds * ds;
    out[0] = cd;
    out[1] = cv;
    out[2] = g.klat * cd + g.klon * cv;
}

// start[b] = {s0, c_speed, c_d, c_d_d, c_d_dd} (JTP:296-299 + the c_speed argument)
// A block owns FR_PAIRS consecutive (start state, candidate) pairs = one contiguous stretch of traj; the samples are
// staged in LDS in output order and leave as full 16-byte vectors (written straight from the computing threads the
// 8-byte stores of a wavefront land in 112-byte runs: 3.5 TB/s instead of the write roofline).
constexpr int FR_THREADS = 256;
__global__ __launch_bounds__(FR_THREADS) void frenet_samples_kernel(const double* __restrict__ start, int64_t B,
                                                                    dcarl_frenet_grid_t g, double* __restrict__ traj,
                                                                    double* __restrict__ cost, int pairs_per_block) {
    extern __shared__ __attribute__((aligned(16))) double tile[];            // [pair][field][t], then the coefficients
    const int NC = g.n_d * g.n_T * g.n_v, NT = g.nt_max;
    const int64_t npairs = B * NC, p0 = (int64_t)blockIdx.x * pairs_per_block;
    const int here = (int)min((int64_t)pairs_per_block, npairs - p0);
    Poly* coef = reinterpret_cast<Poly*>(tile + pairs_per_block * 8 * NT);   // [pair][lat, lon]
    int* steps = reinterpret_cast<int*>(coef + 2 * pairs_per_block);
    for (int lp = threadIdx.x; lp < here; lp += FR_THREADS) {                // the two solves once per pair (13 f64
        const int64_t pair = p0 + lp;                                        // divisions), not once per sample
        const double* s = start + (pair / NC) * 5;
        const Candidate k = candidate(g, (int)(pair % NC));
        const Poly lat = quintic(s[2], s[3], s[4], k.di, 0.0, 0.0, k.Ti);    // JTP:309
        const Poly lon = quartic(s[0], s[1], 0.0, k.tv, 0.0, k.Ti);           // JTP:321
        coef[2 * lp] = lat;
        coef[2 * lp + 1] = lon;
        steps[lp] = k.nt;
        if (cost) write_costs(g, k, lat, lon, cost + pair * 3);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < here * NT; e += FR_THREADS) {
        const int lp = e / NT, it = e - lp * NT;
        const Poly lat = coef[2 * lp], lon = coef[2 * lp + 1];
        const double t = 0.0 + it * g.dt;                                     // np.arange(0.0, Ti, DT)[it]
        const bool live = it < steps[lp];
        double* o = tile + lp * 8 * NT + it;
        o[0 * NT] = live ? poly0(lat, t) : 0.0;
        o[1 * NT] = live ? poly1(lat, t) : 0.0;
        o[2 * NT] = live ? poly2(lat, t) : 0.0;
        o[3 * NT] = live ? poly3(lat, t) : 0.0;
        o[4 * NT] = live ? poly0(lon, t) : 0.0;
        o[5 * NT] = live ? poly1(lon, t) : 0.0;
        o[6 * NT] = live ? poly2(lon, t) : 0.0;
        o[7 * NT] = live ? poly3(lon, t) : 0.0;
    }
    __syncthreads();
    const int n = here * 8 * NT;                                              // doubles of this block, contiguous in traj
    double* dst = traj + p0 * 8 * NT;
    if ((((int64_t)pairs_per_block * 8 * NT) & 1) == 0) {                    // every block starts 16-byte aligned
        for (int i = threadIdx.x * 2; i + 1 < n; i += FR_THREADS * 2)
            *reinterpret_cast<double2*>(dst + i) = *reinterpret_cast<const double2*>(tile + i);
        if ((n & 1) && threadIdx.x == 0) dst[n - 1] = tile[n - 1];
    } else {
        for (int i = threadIdx.x; i < n; i += FR_THREADS) dst[i] = tile[i];
    }
}

// the costs alone (no trajectories requested)
__global__ __launch_bounds__(256) void frenet_costs_kernel(const double* __restrict__ start, int64_t B,
                                                           dcarl_frenet_grid_t g, double* __restrict__ cost) {
    const int NC = g.n_d * g.n_T * g.n_v;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * NC) return;
    const int c = (int)(e % NC);
    const double* s = start + (e / NC) * 5;
    const Candidate k = candidate(g, c);
    const Poly lat = quintic(s[2], s[3], s[4], k.di, 0.0, 0.0, k.Ti);
    const Poly lon = quartic(s[0], s[1], 0.0, k.tv, 0.0, k.Ti);
    write_costs(g, k, lat, lon, cost + e * 3);
}

// ---- global frame + screening (JTP:342-394, predict.py:21-60,84-110, JTP:123-130) --------------------------------
// Spline2D (cubic_spline_planner.py) of the reference path: knots s_k (n_knots), per segment {ax,bx,cx,dx, ay,by,cy,dy}.
__device__ __forceinline__ int spline_segment(const double* __restrict__ knots, int n_knots, double s) {
    int lo = 0, hi = n_knots;                             // bisect.bisect(knots, s) - 1 (Spline.__search_index)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s < knots[mid]) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
}

// glob [B][n_cand][5][nt_max]: x, y, yaw, ds, c (JTP:345-377; c has one entry less, unused slots are 0); path_len
// [B][n_cand] = number of samples inside the spline (JTP:348-349 break).  One thread = one (pair, time step): the spline
// evaluation with its atan2 / sincos is spread over the time steps, loads and stores are contiguous in t; neighbours
// (np.diff, yaw[i+1]) are exchanged through an LDS tile that also serves as the output staging buffer.
__global__ __launch_bounds__(FR_THREADS) void frenet_global_kernel(const double* __restrict__ traj, int64_t B,
                                                                   dcarl_frenet_grid_t g, const double* __restrict__ knots,
                                                                   const double* __restrict__ seg, int n_knots,
                                                                   double* __restrict__ glob, int32_t* __restrict__ path_len,
                                                                   int pairs_per_block) {
    extern __shared__ __attribute__((aligned(16))) double tile[];            // [pair][5][NT], then n[pair]
    const int NC = g.n_d * g.n_T * g.n_v, NT = g.nt_max;
    const int64_t npairs = B * NC, p0 = (int64_t)blockIdx.x * pairs_per_block;
    const int here = (int)min((int64_t)pairs_per_block, npairs - p0);
    int* nvalid = reinterpret_cast<int*>(tile + pairs_per_block * 5 * NT);
    for (int lp = threadIdx.x; lp < here; lp += FR_THREADS) nvalid[lp] = candidate(g, (int)((p0 + lp) % NC)).nt;
    __syncthreads();
    const int work = here * NT;
    // x, y of every sample that lies on the reference path; the first one that does not ends the path (JTP:346-355)
    for (int e = threadIdx.x; e < work; e += FR_THREADS) {
        const int lp = e / NT, it = e - lp * NT;
        const double* src = traj + (p0 + lp) * 8 * NT;
        double* o = tile + lp * 5 * NT;
        double x = 0.0, y = 0.0;
        if (it < nvalid[lp]) {                                                // (nvalid still holds nt here or a smaller
            const double si = src[4 * NT + it];                               //  index that ends the path anyway)
            if (si < knots[0] || si > knots[n_knots - 1]) {
                atomicMin(&nvalid[lp], it);                                   // Spline.calc returns None
            } else {
                const int j = min(spline_segment(knots, n_knots, si), n_knots - 2);
                const double* q = seg + (int64_t)j * 8;
                const double h = si - knots[j], h2 = h * h, h3 = h2 * h;
                const double ix = q[0] + q[1] * h + q[2] * h2 + q[3] * h3, iy = q[4] + q[5] * h + q[6] * h2 + q[7] * h3;
                const double dxs = q[1] + 2.0 * q[2] * h + 3.0 * q[3] * h2, dys = q[5] + 2.0 * q[6] * h + 3.0 * q[7] * h2;
                // JTP:356-357: iyaw = Spline2D.calc_yaw = atan2(dy, dx), then x = ix + di*cos(iyaw + pi/2), y = iy +
                // di*sin(iyaw + pi/2).  cos(atan2(dy,dx) + pi/2) = -dy/|d| and sin(.) = dx/|d| exactly, so the unit normal
                // comes from one reciprocal square root instead of atan2 + sin + cos (three of the four f64 library calls
                // this kernel made per sample; differs from the reference's rounding by ~1e-16, checked to 1e-9)
                const double di = src[it];
                const double d2 = dxs * dxs + dys * dys;
                const double inv = d2 > 0.0 ? 1.0 / sqrt(d2) : 0.0;
                x = ix - di * (dys * inv);
                y = iy + di * (d2 > 0.0 ? dxs * inv : 1.0);                     // atan2(0, 0) = 0: the normal is (0, 1)
            }
        }
        o[it] = x;
        o[NT + it] = y;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < work; e += FR_THREADS) {                    // JTP:358-368 np.diff, arctan2, sqrt, append
        const int lp = e / NT, it = e - lp * NT, n = nvalid[lp];
        double* o = tile + lp * 5 * NT;
        double yaw = 0.0, ds = 0.0;
        if (n >= 2 && it < n) {
            const int i = min(it, n - 2);                                     // the last entry repeats the one before
            const double dx = o[i + 1] - o[i], dy = o[NT + i + 1] - o[NT + i];
            yaw = atan2(dy, dx);
            ds = sqrt(dx * dx + dy * dy);
        } else if (n < 2 && it == 0) {
            yaw = 0.1; ds = 0.1;                                              // empty diff (JTP:367-368)
        }
        if (it >= n) { o[it] = 0.0; o[NT + it] = 0.0; }
        if (it + 1 < n && ds < 0.00001) ds = 0.1;                             // JTP:375-376 (in place: it is the ds returned)
        o[2 * NT + it] = yaw;
        o[3 * NT + it] = ds;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < work; e += FR_THREADS) {                    // JTP:373-377
        const int lp = e / NT, it = e - lp * NT, n = nvalid[lp];
        double* o = tile + lp * 5 * NT;
        o[4 * NT + it] = (it + 1 < n) ? (o[2 * NT + it + 1] - o[2 * NT + it]) / o[3 * NT + it] : 0.0;
    }
    __syncthreads();
    for (int lp = threadIdx.x; lp < here; lp += FR_THREADS) path_len[p0 + lp] = nvalid[lp];
    const int nd = here * 5 * NT;
    double* dst = glob + p0 * 5 * NT;
    if ((((int64_t)pairs_per_block * 5 * NT) & 1) == 0) {
        for (int i = threadIdx.x * 2; i + 1 < nd; i += FR_THREADS * 2)
            *reinterpret_cast<double2*>(dst + i) = *reinterpret_cast<const double2*>(tile + i);
        if ((nd & 1) && threadIdx.x == 0) dst[nd - 1] = tile[nd - 1];
    } else {
        for (int i = threadIdx.x; i < nd; i += FR_THREADS) dst[i] = tile[i];
    }
}

// get_optimal_trajectory (JTP:123-130) for one start state per thread: candidates in ascending cf (stable), skipping
// those that fail check_paths (JTP:381-394), the first one that passes predict.check_collision (predict.py:21-60) -> its
// index + 1; 0 (the brake trajectory) if none.  obstacles [B][n_obs][5] = {x, y, vx, vy, yaw}; every vehicle is two
// circles centres (front / back, predict.py:84-110) moving at constant velocity.
__global__ __launch_bounds__(256) void frenet_select_kernel(const double* __restrict__ traj, const double* __restrict__ glob,
                                                            const int32_t* __restrict__ path_len, const double* __restrict__ cost,
                                                            const double* __restrict__ obstacles, int n_obs, int64_t B,
                                                            dcarl_frenet_grid_t g, dcarl_frenet_limits_t lim,
                                                            int32_t* __restrict__ choice, uint8_t* __restrict__ ok_out) {
    const int NC = g.n_d * g.n_T * g.n_v, NT = g.nt_max;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    unsigned done = 0;                                     // NC <= 32 (checked on the host)
    int best = 0;
    for (int rank = 0; rank < NC; ++rank) {
        int c = -1;
        double cf = 0.0;
        for (int j = 0; j < NC; ++j) {                     // stable selection sort by cf (sorted(..., key=cf))
            if (done & (1u << j)) continue;
            const double v = cost[(b * NC + j) * 3 + 2];
            if (c < 0 || v < cf) { c = j; cf = v; }
        }
        done |= 1u << c;
        const int64_t e = b * NC + c;
        const int n = path_len[e];
        const Candidate k = candidate(g, c);
        const double* s_d = traj + e * 8 * NT + 5 * NT;
        const double* s_dd = s_d + NT;
        const double* x = glob + e * 5 * NT;
        const double *y = x + NT, *cv = x + 4 * NT;
        bool ok = true;                                    // JTP:385-390
        for (int i = 0; i < k.nt; ++i) ok = ok && !(s_d[i] > lim.max_speed) && !(fabs(s_dd[i]) > lim.max_accel);
        for (int i = 0; i + 1 < n; ++i) ok = ok && !(fabs(cv[i]) > lim.max_curvature);
        bool free_path = true;                             // predict.py:21-60
        if (n_obs > 0 && k.nt >= 2) {
            const int len_t = min(n - 1, lim.n_predict - 1);
            for (int o = 0; o < n_obs && free_path; ++o) {
                const double* ob = obstacles + (b * n_obs + o) * 5;
                const double gx = cos(ob[4]) * lim.move_gap, gy = sin(ob[4]) * lim.move_gap;
                for (int sign = 1; sign >= -1 && free_path; sign -= 2)
                    for (int t = 2; t < len_t; t += 2) {
                        const double px = ob[0] + t * g.dt * ob[2] + sign * gx, py = ob[1] + t * g.dt * ob[3] + sign * gy;
                        const double dd = (px - x[t]) * (px - x[t]) + (py - y[t]) * (py - y[t]);
                        if (dd <= lim.check_radius * lim.check_radius) { free_path = false; break; }
                    }
            }
        }
        if (ok_out) ok_out[e] = (uint8_t)((ok ? 1 : 0) | (free_path ? 2 : 0));
        if (ok && free_path && best == 0) {
            best = c + 1;
            if (!ok_out) break;
        }
    }
    choice[b] = best;
}

int launch_frenet_global(const double* traj, int64_t B, const dcarl_frenet_grid_t& g, const double* knots, const double* seg,
                         int n_knots, double* glob, int32_t* path_len, hipStream_t st) {
    const int64_t n = B * g.n_d * g.n_T * g.n_v;
    if (n == 0) return 0;
    const int per = (int)std::max<int64_t>(1, std::min<int64_t>(FR_THREADS / std::max(1, g.nt_max),
                                                                 32768 / (5 * 8 * (int64_t)std::max(1, g.nt_max))));
    const size_t lds = (size_t)per * (5 * g.nt_max * 8 + sizeof(int)) + 8;
    hipLaunchKernelGGL(frenet_global_kernel, dim3((unsigned)((n + per - 1) / per)), dim3(FR_THREADS), lds, st, traj, B, g, knots,
                       seg, n_knots, glob, path_len, per);
    return 0;
}

int launch_frenet_select(const double* traj, const double* glob, const int32_t* path_len, const double* cost,
                         const double* obstacles, int n_obs, int64_t B, const dcarl_frenet_grid_t& g,
                         const dcarl_frenet_limits_t& lim, int32_t* choice, uint8_t* ok, hipStream_t st) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(frenet_select_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, traj, glob, path_len, cost,
                       obstacles, n_obs, B, g, lim, choice, ok);
    return 0;
}

int launch_frenet(const double* start, int64_t B, const dcarl_frenet_grid_t& g, double* traj, double* cost, hipStream_t st) {
    const int64_t NC = (int64_t)g.n_d * g.n_T * g.n_v;
    if (B == 0 || NC == 0) return 0;
    if (traj) {
        // as many pairs per block as give every thread about two samples, within 32 KiB of LDS
        const int per = (int)std::max<int64_t>(1, std::min<int64_t>(2 * FR_THREADS / std::max(1, g.nt_max),
                                                                     32768 / (8 * 8 * (int64_t)std::max(1, g.nt_max))));
        const int64_t blocks = (B * NC + per - 1) / per;
        const size_t lds = (size_t)per * (8 * g.nt_max * 8 + 2 * sizeof(Poly) + sizeof(int));
        hipLaunchKernelGGL(frenet_samples_kernel, dim3((unsigned)blocks), dim3(FR_THREADS), lds, st, start, B, g, traj, cost,
                           per);
    } else if (cost)
        hipLaunchKernelGGL(frenet_costs_kernel, dim3((unsigned)((B * NC + 255) / 256)), dim3(256), 0, st, start, B, g, cost);
    return 0;
}

}  // namespace dcarl
