"""Extracts the field-log records in which the vehicle executed an RL action, with the statistics the reference
logged for them (RLS.py:224-241 driving_record format: obs[20], action, reward, done, n_rule, mean_rule, var_rule,
n_RL, mean_RL, var_RL, timestamp) -> tests/golden/rls_field_decisions.npz.  Run in the build container only."""
import os
import numpy as np

REF = "/root/reference/Field_testing"
rows, neg, open_rule = [], [], 0
for sc in ("Scenario2", "Scenario3"):
    d = np.loadtxt(os.path.join(REF, sc, "RLS.txt"))
    sel = d[:, 26] >= 0                                       # RL statistics were computed <=> an RL action was executed
    rows.append(np.column_stack([d[sel, 20], d[sel, 23:29]]))
    # the NEGATIVE decisions: the rule action was executed.  The log carries only the rule action's statistics for them
    # (RLS.py:226-229 writes -1 for the RL columns), which is enough whenever one of the gates of RLS.py:141 that look at
    # the rule action alone is closed (visited_times_rule < 30 or mean_rule > -0.1): act_test must then return 0 whatever
    # the candidates' statistics are.  (5 rule rows have both gates open: every candidate failed there, statistics unknown.)
    rule = d[:, 20] == 0
    closed = rule & ((d[:, 23] < 30) | (d[:, 24] > -0.1))
    neg.append(d[closed, 23:26])
    open_rule += int((rule & ~closed).sum())
# One more reference-held log of the same format family: tools/DCARL/driving_record.txt (RLS.py:217-241 of the tools/DCARL
# copy: state_with_action[21], reward, done, n_rule, mean_rule, var_rule, n_RL, mean_RL, var_RL) -- 107 rows, every one a rule
# action (column 20 == 0, RL columns -1): closed-gate negatives wherever visited_times_rule < 30 or mean_rule > -0.1.
TOOLS = os.path.join(REF, "Software_and_Raw_Data_on_Self-Driving_Vehicle/software/src/tools/DCARL")
dr = np.loadtxt(os.path.join(TOOLS, "driving_record.txt"))
assert dr.shape == (107, 29) and np.all(dr[:, 20] == 0) and np.all(dr[:, 26:29] == -1)
dr_closed = (dr[:, 23] < 30) | (dr[:, 24] > -0.1)
n_before = len(np.unique(np.concatenate(neg), axis=0))
neg.append(dr[dr_closed, 23:26])
open_rule_dr = int((~dr_closed).sum())
out = np.concatenate(rows)
neg = np.unique(np.concatenate(neg), axis=0)                  # 3 732 + 107 rows, many repeated: keep the distinct statistics
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rls_field_decisions.npz"),
         action=out[:, 0].astype(np.int32), n_rule=out[:, 1], mean_rule=out[:, 2], var_rule=out[:, 3],
         n_rl=out[:, 4], mean_rl=out[:, 5], var_rl=out[:, 6],
         neg_n_rule=neg[:, 0], neg_mean_rule=neg[:, 1], neg_var_rule=neg[:, 2], rule_rows_with_open_gates=open_rule,
         driving_record_rows=len(dr), driving_record_closed=int(dr_closed.sum()), driving_record_open=open_rule_dr,
         distinct_from_field_logs=n_before)
print(out.shape, np.unique(out[:, 0]), neg.shape, open_rule, len(dr), int(dr_closed.sum()), open_rule_dr, n_before)
