// Episode-return reduction: where the "cumulative reward" column of the record table comes from.
//   TS  = Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Test_Scenarios/TestScenario_Town03.py
//   DVC = Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Agent/drl_library/dqn/dqn_value_collect.py
//   RLS = Field_testing/Software_and_Raw_Data_on_Self-Driving_Vehicle/software/src/tools/DCARL/stable_baselines/deepq/RLS.py
// * episode_returns_kernel: the simulator side.  Per step: v = sqrt(vx^2 + vy^2), reward = 0.1*sqrt(v) (TS:403-404), -100 on
//   a collision (TS:407-410), 0.0 when the vehicle is stuck and has not passed (TS:419-421); per episode: the sum of the
//   step rewards (DVC:119) and AveSpeed = mean of v over the episode's steps (TS:386-387, 411).  One wavefront per episode:
//   64 consecutive steps per load (coalesced), per-lane partial sums, one fixed-order butterfly -> run-to-run identical.
// * nstep_backup_kernel: the field side, RLS.add_data's value stream (RLS:185-215).  A transition leaves the 10-deep
//   trajectory buffer with its OWN reward once 10 newer ones exist (RLS:188-199); when the episode ends, what is left in the
//   buffer gets the terminal reward discounted by its distance to the end, rew_last * gamma**k (RLS:202-215).  Pure
//   streaming: value[t] depends on rew[t], the episode's last reward and the position only.
#include "common.h"

namespace dcarl {

constexpr int EP_WAVES = 4;

__global__ __launch_bounds__(EP_WAVES* WAVE) void episode_returns_kernel(
    const double* __restrict__ vx, const double* __restrict__ vy, const uint8_t* __restrict__ flags,
    const int64_t* __restrict__ ep_off, int64_t E, double* __restrict__ step_reward, double* __restrict__ episode_reward,
    double* __restrict__ ave_speed) {
#pragma clang fp contract(off)   // x*x + y*y rounds three times in the reference (TS:386, 403)
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t nw = (int64_t)gridDim.x * EP_WAVES;
    for (int64_t ep = (int64_t)blockIdx.x * EP_WAVES + (threadIdx.x >> 6); ep < E; ep += nw) {
        const int64_t b = ep_off[ep], e = ep_off[ep + 1];
        double sr = 0.0, sv = 0.0;
        for (int64_t i = b + lane; i < e; i += WAVE) {
            const double x = vx[i], y = vy[i];
            const double v = sqrt(x * x + y * y);                  // TS:386 / TS:403 (IEEE square roots, like math.sqrt)
            double r = sqrt(v) * 0.1;                              // TS:404
            const int f = flags[i];
            if (f & 1) r = -100.0;                                 // TS:407-410 collision
            if (!(f & 2) && (f & 4)) r = 0.0;                      // TS:414-421: `if pass ... elif stuck: reward = 0.0`
            if (step_reward) step_reward[i] = r;
            sr += r;
            sv += v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o); sv += __shfl_xor(sv, o); }
        if (lane == 0) {
            episode_reward[ep] = sr;                               // DVC:119 episode_reward += reward
            if (ave_speed) ave_speed[ep] = e > b ? sv / (double)(e - b) : 0.0;   // TS:411 sum(driving_speed)/len(driving_speed)
        }
    }
}

__global__ __launch_bounds__(EP_WAVES* WAVE) void nstep_backup_kernel(
    const double* __restrict__ rew, const int64_t* __restrict__ ep_off, const uint8_t* __restrict__ ep_done, int64_t E,
    const double* __restrict__ gamma_pow, int horizon, double* __restrict__ value, uint8_t* __restrict__ recorded) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t nw = (int64_t)gridDim.x * EP_WAVES;
    for (int64_t ep = (int64_t)blockIdx.x * EP_WAVES + (threadIdx.x >> 6); ep < E; ep += nw) {
        const int64_t b = ep_off[ep], e = ep_off[ep + 1];
        if (e <= b) continue;
        const bool done = ep_done[ep] != 0;
        const double last = rew[e - 1];                            // RLS:203 rew_right
        for (int64_t i = b + lane; i < e; i += WAVE) {
            const int64_t k = e - 1 - i;                           // transitions after this one
            double v = rew[i];                                     // RLS:190-192: popped by the while-loop with its own reward
            bool rec = true;
            if (k < horizon) {                                     // still buffered when the episode ends
                rec = done;
                v = done ? last * gamma_pow[k] : 0.0;              // RLS:207 rew_right*self.gamma**len(self.trajectory_buffer)
            }
            value[i] = v;
            if (recorded) recorded[i] = rec ? 1 : 0;
        }
    }
}

static unsigned ep_blocks(int64_t E) {
    int64_t blocks = (E + EP_WAVES - 1) / EP_WAVES;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

int launch_episode_returns(const double* vx, const double* vy, const uint8_t* flags, const int64_t* ep_off, int64_t E,
                           double* step_reward, double* episode_reward, double* ave_speed, hipStream_t st) {
    if (E == 0) return 0;
    hipLaunchKernelGGL(episode_returns_kernel, dim3(ep_blocks(E)), dim3(EP_WAVES * WAVE), 0, st, vx, vy, flags, ep_off, E,
                       step_reward, episode_reward, ave_speed);
    return 0;
}

int launch_nstep_backup(const double* rew, const int64_t* ep_off, const uint8_t* ep_done, int64_t E, const double* gamma_pow,
                        int horizon, double* value, uint8_t* recorded, hipStream_t st) {
    if (E == 0) return 0;
    hipLaunchKernelGGL(nstep_backup_kernel, dim3(ep_blocks(E)), dim3(EP_WAVES * WAVE), 0, st, rew, ep_off, ep_done, E,
                       gamma_pow, horizon, value, recorded);
    return 0;
}

}  // namespace dcarl
