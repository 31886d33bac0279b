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
out = np.concatenate(rows)
neg = np.unique(np.concatenate(neg), axis=0)                  # 3 732 rows, many repeated: keep the distinct statistics
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rls_field_decisions.npz"),
         action=out[:, 0].astype(np.int32), n_rule=out[:, 1], mean_rule=out[:, 2], var_rule=out[:, 3],
         n_rl=out[:, 4], mean_rl=out[:, 5], var_rl=out[:, 6],
         neg_n_rule=neg[:, 0], neg_mean_rule=neg[:, 1], neg_var_rule=neg[:, 2], rule_rows_with_open_gates=open_rule)
print(out.shape, np.unique(out[:, 0]), neg.shape, open_rule)
