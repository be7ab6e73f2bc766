"""Reads the per-wave timelines a -DTRICO_SWEEP_DIAG build of k_fpc32_sweep wrote (TRICO_SWEEP_DIAG_FILE) and says where the time goes."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
S, G, arity, L = [int(x) for x in raw[:4]]
r = raw[4:].reshape(-1, 8)[: (S + G) * arity]
hw = r[:, 0].astype(np.int64)
xcc = r[:, 1].astype(np.int64) & 15
t0 = r[:, 2].min()
start, l0, l1, end = [(r[:, k].astype(np.int64) - int(t0)) / 100.0 for k in (2, 3, 4, 5)]
main = np.arange(S * arity)
comp = main % arity
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
print("S %d guard %d arity %d L %d; starts %.2f .. %.2f us; ends %.2f .. %.2f us (main), guard ends up to %.2f" % (
    S, G, arity, L, start.min(), start.max(), end[main].min(), end[main].max(), end[S * arity:].max() if G else 0))
for c in range(arity):
    m = main[comp == c]
    d = l1[m] - l0[m]
    print("component %d: loop time us min %.1f p10 %.1f median %.1f p90 %.1f max %.1f; prologue median %.1f us; epilogue median %.1f us" % (
        c, d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max(), np.median(l0[m] - start[m]), np.median(end[m] - l1[m])))
# by place
unit = xcc * 1000000 + se * 10000 + sh * 1000 + cu * 10 + simd
print("distinct XCC %d, SE %s, SH %s, CU ids %s" % (len(set(xcc[main])), sorted(set(se[main])), sorted(set(sh[main])), sorted(set(cu[main]))))
cuid = xcc * 1000 + se * 100 + sh * 50 + cu
wg_end = end[main].reshape(S, arity).max(axis=1)
wg_cu = cuid[main].reshape(S, arity)[:, 0]
per_cu = {}
for e, u in zip(wg_end, wg_cu):
    per_cu.setdefault(int(u), []).append(e)
cnt = np.array([len(v) for v in per_cu.values()])
last = np.array([max(v) for v in per_cu.values()])
print("compute units used %d; workgroups per unit min %d max %d; last end per unit: min %.1f median %.1f max %.1f us" % (len(per_cu), cnt.min(), cnt.max(), last.min(), np.median(last), last.max()))
for x in sorted(set(xcc[main])):
    m = main[xcc[main] == x]
    print("  XCC %d: waves %d, loop time median %.1f us, end median %.1f max %.1f" % (x, len(m), np.median((l1 - l0)[m]), np.median(end[m]), end[m].max()))
# waves per SIMD of the slow component
slow = arity - 1
key = cuid * 4 + simd
zc = {}
for i in main[comp == slow]:
    zc.setdefault(int(key[i]), []).append(l1[i] - l0[i])
by = {}
for k, v in zc.items():
    by.setdefault(len(v), []).extend(v)
for k in sorted(by):
    print("  SIMDs holding %d waves of component %d: %d waves, loop time median %.1f us" % (k, slow, len(by[k]), np.median(by[k])))
allc = {}
for i in main:
    allc.setdefault(int(key[i]), []).append(l1[i] - l0[i])
by = {}
for k, v in allc.items():
    by.setdefault(len(v), []).extend(v)
for k in sorted(by):
    print("  SIMDs holding %d waves: %d waves, loop time median %.1f us" % (k, len(by[k]), np.median(by[k])))
# along the stream
gi = (main // arity)
for lo in range(0, S, max(1, S // 8)):
    m = main[(gi >= lo) & (gi < lo + max(1, S // 8))]
    print("  segments %5d..: loop time median %.1f us, end median %.1f" % (lo, np.median((l1 - l0)[m]), np.median(end[m])))
# by order of arrival on the compute unit (blockIdx / number of units): loop time and end of the slow component's waves
z = np.arange(S) * arity + slow
order = {}
k = np.zeros(S, int)
for g in range(S):
    u = int(cuid[z[g]])
    k[g] = order.get(u, 0)
    order[u] = k[g] + 1
print("  by arrival on the unit, loop time median: " + " ".join("%d:%.0f" % (kk, np.median((l1 - l0)[z][k == kk])) for kk in range(k.max() + 1)))
print("  by arrival on the unit, end median:       " + " ".join("%d:%.0f" % (kk, np.median(end[z][k == kk])) for kk in range(k.max() + 1)))
print("  by arrival on the unit, end max:          " + " ".join("%d:%.0f" % (kk, np.max(end[z][k == kk])) for kk in range(k.max() + 1)))
# what the late compute units have in common: workgroups on the unit, the most waves of the slow component on one of its SIMDs, the XCC
zmax = {}
for kk, v in zc.items():
    zmax[kk // 4] = max(zmax.get(kk // 4, 0), len(v))
grp = {}
for u, ends in per_cu.items():
    grp.setdefault((len(ends), zmax.get(u, 0)), []).append(max(ends))
for kk in sorted(grp):
    print("  units with %d workgroups, at most %d waves of component %d on a SIMD: %d units, last end median %.1f max %.1f" % (kk[0], kk[1], slow, len(grp[kk]), np.median(grp[kk]), np.max(grp[kk])))
gx = {}
for u, ends in per_cu.items():
    gx.setdefault(u // 1000, []).append(max(ends))
print("  last end per unit by XCC (median / max): " + " ".join("%d:%.0f/%.0f" % (x, np.median(v), np.max(v)) for x, v in sorted(gx.items())))
late = sorted(per_cu.items(), key=lambda kv: -max(kv[1]))[:8]
print("  the latest units: " + "; ".join("xcc %d se %d cu %d: %d wgs, end %.0f" % (u // 1000, (u % 1000) // 100, u % 50, len(e), max(e)) for u, e in late))
