"""CPU restatement of the episode-return arithmetic — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the simulator-dependent inputs: the reference computes these numbers inside a running CARLA
scenario (TestScenario_Town03.py imports `carla`) and inside RLS.add_data (RLS.py imports the absent `rtree`), so neither
file can be imported here and no golden vector exists in the reference.  What IS restated below, line by line, is the pure
Python arithmetic; tests/test_episodes.py pins it on hand-checkable episodes.

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
