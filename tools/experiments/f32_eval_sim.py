"""How often would an f32 evaluation of V (with f64-exact fallback) be uncertain?  Per record of lane (state): candidate a gets
the new key k; m = max of the OTHER candidates' keys.  E2: |k - m| <= tau (ordering unknown -> exact path).  E1: a was the
leader and k < m - tau (the new leader is the runner-up, whose own lead over the third is not certified).
Counts per lane-record, per wave-record (any of 64 lanes) and per wave-quad.   python f32_eval_sim.py [T] [sim1|rand]"""
import numpy as np, math, sys
rng = np.random.default_rng(0)
Q = np.array([48.9, 67.6, -45.8, 67.2, -46.3, 66.6, 60.3, 66.9, 60.6, -35., 60.7])
A = 11; L = 64; T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
mode = sys.argv[2] if len(sys.argv) > 2 else 'sim1'
hoeff = 150. * math.sqrt(math.log(20) / 2); cap = 100.; thres = 10; rule = 0
Qs = rng.uniform(-50, 100, (L, A)) if mode == 'rand' else np.tile(Q, (L, 1))
n = np.zeros((L, A)); s = np.zeros((L, A)); q = np.zeros((L, A))
V = np.full((L, A), -50.); V[:, rule] = 100.
lanes = np.arange(L)
K = None; D = np.zeros(L)
e1 = e2 = w1 = w2 = 0; q1 = q2 = 0; qa1 = qa2 = False
for t in range(T):
    a = rng.integers(0, A, L)
    x = Qs[lanes, a] + 50 * rng.standard_normal(L)
    if K is None: K = x.copy()
    xs = x - K
    D = np.maximum(D, np.abs(xs))
    n[lanes, a] += 1; s[lanes, a] += xs; q[lanes, a] += xs * xs
    nn = n[lanes, a]; md = s[lanes, a] / nn; var = np.maximum(q[lanes, a] / nn - md * md, 0); sd = np.sqrt(var)
    m = K + md
    up = np.minimum(cap, m + hoeff / np.sqrt(nn)); lo = m - hoeff / np.sqrt(nn); ci = m - 4 * sd / (nn + 1) - hoeff / np.sqrt(nn + 1)
    k = np.where(a == rule, up, np.minimum(lo, ci))
    valid = nn > thres
    lead0 = V.argmax(1)
    Vo = V.copy(); Vo[lanes, a] = -np.inf
    mo = Vo.max(1)
    knew = np.where(valid, k, V[lanes, a])
    tau = 2 * 8 * 2.0 ** -24 * (2 * D + hoeff)
    u2 = valid & (np.abs(knew - mo) <= tau)
    u1 = valid & (a == lead0) & (knew < mo - tau)
    V[lanes, a] = knew
    e1 += u1.sum(); e2 += u2.sum(); w1 += u1.any(); w2 += u2.any()
    qa1 |= u1.any(); qa2 |= u2.any()
    if t % 4 == 3:
        q1 += qa1; q2 += qa2; qa1 = qa2 = False
print(mode, 'T', T, 'tau~%.2e' % tau.mean())
print('E1 leader fell below runner-up: lane %.3e  wave-record %.3f  wave-quad %.3f' % (e1 / (T * L), w1 / T, q1 / (T / 4)))
print('E2 within tau of the others   : lane %.3e  wave-record %.3f  wave-quad %.3f' % (e2 / (T * L), w2 / T, q2 / (T / 4)))
