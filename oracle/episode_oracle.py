"""CPU restatement of the episode-return arithmetic — TEST INFRASTRUCTURE ONLY.

Two halves, pinned differently:

* the n-step / gamma terminal BACK-UP of RLS.add_data (``RlsValueStream``, RLS:185-215) is PINNED on the reference's own
  output: tools/DCARL/visited_value.txt is what that function appended in the field (209 600 [action, value] rows: 0 and
  -0.95^k; 63 complete 10-step runs -0.630249 ... -0.95 -1.0 = rew_right * gamma ** len(buffer), 1 622 runs truncated by
  episodes shorter than the buffer).  tests/golden/make_episode_goldens.py stores the column with the episodes it implies,
  tests/test_episodes.py feeds them to this restatement and demands the file back (values to its %f precision, actions and
  order exactly); a buffer depth of 9 or 11 or the exponent the other way round fail that test.
* the STEP REWARD (``step_reward`` / ``episode_reward``, TS:402-421, DVC:94,119) stays PARITY UNPINNED: the reference computes
  it inside a running CARLA scenario (TestScenario_Town03.py imports `carla`, absent here) and no reference file holds its
  inputs next to its outputs.  It is restated line by line and checked on hand-checkable episodes only.

  TS  = Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Test_Scenarios/TestScenario_Town03.py
  DVC = Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Agent/drl_library/dqn/dqn_value_collect.py
  RLS = Field_testing/Software_and_Raw_Data_on_Self-Driving_Vehicle/software/src/tools/DCARL/stable_baselines/deepq/RLS.py
"""
import math
from collections import deque


def step_reward(vx, vy, collision, passed, stuck):
    """TS:402-421, in the reference's statement order."""
    v = math.sqrt(vx ** 2 + vy ** 2)          # TS:403
    reward = math.sqrt(v) * 0.1               # TS:404
    done = False
    if collision:                             # TS:407-410
        done = True
        reward = -100
    if passed:                                # TS:413-415
        done = True
    elif stuck:                               # TS:418-421
        reward = 0.0
        done = True
    return reward, done, v


def episode_reward(steps):
    """DVC:94,119: episode_reward = 0; episode_reward += reward for every step; AveSpeed = sum(speed)/len (TS:411)."""
    total = 0
    speeds = []
    rewards = []
    for (vx, vy, c, p, s) in steps:
        r, _, v = step_reward(vx, vy, c, p, s)
        speeds.append(v)                      # TS:387 self.driving_speed.append(ego_speed)
        rewards.append(r)
        total += r
    return total, (sum(speeds) / len(speeds) if speeds else 0.0), rewards


class RlsValueStream:
    """RLS.add_data's bookkeeping (RLS:185-215) without the R-tree insert: the [action, value] rows it appends to
    visited_state_value, in order.  gamma = 0.95 (RLS:31), buffer = deque(maxlen=20) (RLS:24)."""

    def __init__(self, gamma=0.95):
        self.gamma = gamma
        self.trajectory_buffer = deque(maxlen=20)
        self.rows = []                        # (transition id, action, value)

    def add_data(self, tid, action, rew, done):
        self.trajectory_buffer.append((tid, action, rew, done))                  # RLS:186
        while len(self.trajectory_buffer) > 10:                                  # RLS:188
            tid_left, action_left, rew_left, _ = self.trajectory_buffer.popleft()
            self.rows.append((tid_left, action_left, rew_left))                  # RLS:192-194 r_to_record = rew_left
        if done:                                                                 # RLS:202
            _, _, rew_right, _ = self.trajectory_buffer[-1]                      # RLS:203
            while len(self.trajectory_buffer) > 0:                               # RLS:204
                tid_left, action_left, _, _ = self.trajectory_buffer.popleft()
                r_to_record = rew_right * self.gamma ** len(self.trajectory_buffer)   # RLS:207
                self.rows.append((tid_left, action_left, r_to_record))
