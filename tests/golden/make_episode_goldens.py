"""The reference's own output of RLS.add_data -> tests/golden/rls_visited_value.npz.  Run in the build container only.

tools/DCARL/visited_value.txt (format RLS.py:58-59 "%f %f": action, value) is what RLS.add_data (tools/DCARL/stable_baselines/
deepq/RLS.py:185-215) appended while the field vehicle trained: 209 600 rows, 11 distinct values — 0 and -0.95^k, k = 0..9.
A transition leaves the 10-deep trajectory buffer with its own reward (RLS.py:188-194: every pre-terminal reward is 0), and when
an episode ends (`done`) the buffer is flushed with r = rew_right * gamma ** len(buffer) (RLS.py:202-207): each row with value
-1.0 is the LAST transition of an episode whose terminal reward was -1, preceded by -0.95, -0.9025, ... for as many transitions
as the buffer still held — 63 complete 10-step runs (-0.630249 ... -0.95 -1.0), 1 622 runs truncated by episodes shorter than
the buffer.  That pins gamma = 0.95 (RLS.py:31), the exponent's orientation and the buffer depth of 10 (RLS.py:188).

What the column implies about the episodes (stored next to it, so that a restatement can be FED):
  * every run ending in -1.0 ends an episode with terminal reward -1;
  * a truncated run of k < 10 rows IS a whole episode of k transitions (the buffer held nothing older);
  * a complete run belongs to an episode of >= 10 transitions: the zero rows before it back to the previous boundary are its
    earlier transitions, recorded with their own reward 0;
  * the remaining stretches of zero rows are episodes with terminal reward 0 (0 * gamma^k = 0 for every row): one episode per
    stretch reproduces them, whatever their true split was."""
import os

import numpy as np

REF = "/root/reference/Field_testing/Software_and_Raw_Data_on_Self-Driving_Vehicle/software/src/tools/DCARL"
d = np.loadtxt(os.path.join(REF, "visited_value.txt"))
assert d.shape == (209600, 2)
action, value = d[:, 0].astype(np.uint8), d[:, 1]
N = len(value)
gamma, depth = 0.95, 10
ends = np.flatnonzero(value == -1.0)
ep_first, ep_last, run_len = [], [], []
prev = 0                                              # first row not yet assigned to an episode
for e in ends:
    k = 1
    while k < depth and e - k >= prev and abs(value[e - k] + gamma ** k) < 6e-7:
        k += 1
    first = e - k + 1
    if k == depth:                                    # an episode of >= 10 transitions: the zeros before it are its own
        first = prev
    elif first > prev:                                # zeros between the previous boundary and a short episode: a zero episode
        ep_first.append(prev); ep_last.append(first - 1); run_len.append(0)
    ep_first.append(first); ep_last.append(e); run_len.append(k)
    prev = e + 1
if prev < N:
    ep_first.append(prev); ep_last.append(N - 1); run_len.append(0)
ep_first, ep_last, run_len = np.array(ep_first), np.array(ep_last), np.array(run_len)
assert ep_first[0] == 0 and ep_last[-1] == N - 1 and np.all(ep_first[1:] == ep_last[:-1] + 1)
ep_off = np.concatenate([ep_first, [N]]).astype(np.int64)
terminal = np.where(run_len > 0, -1.0, 0.0)
print("rows", N, "episodes", len(run_len), "complete runs", int((run_len == depth).sum()), "truncated", int(((run_len > 0) & (run_len < depth)).sum()),
      "zero episodes", int((run_len == 0).sum()), "distinct values", len(np.unique(value)))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rls_visited_value.npz"),
                    action=action, value=value, ep_off=ep_off, terminal_reward=terminal, run_len=run_len.astype(np.int32),
                    gamma=gamma, depth=depth)
