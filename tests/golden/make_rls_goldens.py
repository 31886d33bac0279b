"""Extracts the field-log records in which the vehicle executed an RL action, with the statistics the reference
logged for them (RLS.py:224-241 driving_record format: obs[20], action, reward, done, n_rule, mean_rule, var_rule,
n_RL, mean_RL, var_RL, timestamp) -> tests/golden/rls_field_decisions.npz.  Run in the build container only."""
import os
import numpy as np

REF = "/root/reference/Field_testing"
rows = []
for sc in ("Scenario2", "Scenario3"):
    d = np.loadtxt(os.path.join(REF, sc, "RLS.txt"))
    sel = d[:, 26] >= 0                                       # RL statistics were computed <=> an RL action was executed
    rows.append(np.column_stack([d[sel, 20], d[sel, 23:29]]))
out = np.concatenate(rows)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rls_field_decisions.npz"),
         action=out[:, 0].astype(np.int32), n_rule=out[:, 1], mean_rule=out[:, 2], var_rule=out[:, 3],
         n_rl=out[:, 4], mean_rl=out[:, 5], var_rl=out[:, 6])
print(out.shape, np.unique(out[:, 0]))
