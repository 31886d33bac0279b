"""bench.py's JSON line -> a markdown table of every leg (kernel, time, algorithmic bytes, achieved GB/s, fraction of the 8 TB/s HBM
peak, counter traffic / algorithmic):   python tools/roofline_table.py gpurun_out/v1/bench.json > profiles/r04_roofline_table.md"""
import json, sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows = []
# the traffic column is a look-up in profiles/hbm_traffic.json (bench.py does the same one at run time); a second argument re-does it against
# a newer file — the counter passes of a profile round are collected AFTER the bench run whose line this table prints
NEWER = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None


def relook(r, tr):
    if NEWER is None or not r.get("algorithmic_bytes"):
        return tr
    for k, v in NEWER.items():
        name, _, alg = k.partition("|")
        if alg == str(r["algorithmic_bytes"]) and str(r.get("kernel", "")).startswith(name) and "+" not in str(r.get("kernel", "")):
            return v.get("hbm_bytes_per_launch", tr)
    return tr


def row(name, r):
    if not isinstance(r, dict) or "kernel_ms" not in r:
        return
    alg, tr = r.get("algorithmic_bytes"), r.get("traffic")
    tr = relook(r, tr)
    rows.append((name, str(r.get("kernel", ""))[:70], r["kernel_ms"], alg, r.get("achieved", r.get("achieved_gbs")), r.get("frac"),
                 (tr / alg) if (tr and alg) else None))


row("configs[1] online (HEADLINE)", d["roofline"])
for k, v in d.get("other_configs", {}).items():
    row(k, v)
print(f"# Roofline table of one `python bench.py --steps {d['steps']} --warmup {d['warmup']}` run on one MI355X (HBM peak 8 TB/s; measured copy ceiling "
      f"{d['roofline'].get('measured_copy_gbs', 0):.0f} GB/s)\n")
print(f"headline: {d['value']:.4g} {d['unit']}, {d['ms_per_step']:.3f} ms per step; cpu_baseline ({d['cpu_baseline']['kind']}, {d['cpu_baseline']['cores']} threads): "
      f"{d['cpu_baseline']['value']:.4g} {d['cpu_baseline']['unit']}\n")
print("| leg | kernel(s) | ms | algorithmic GB | achieved GB/s | of 8 TB/s | counter traffic / algorithmic |")
print("|---|---|---|---|---|---|---|")
for n, k, ms, alg, gbs, fr, tr in rows:
    print(f"| {n} | `{k}` | {ms:.3f} | {alg / 1e9:.2f} | {gbs:.0f} | {100 * fr:.1f} % | {('%.2fx' % tr) if tr else '—'} |")
for cfg, part in (("configs[3]", "balanced"), ("configs[4]", "contiguous")):
    for m in ("batch", "trace"):
        v = d.get("other_configs", {}).get(f"{cfg}.shards_of_8.{m}")
        if not (isinstance(v, dict) and part in v):
            continue
        b = v[part]
        line = (f"\n{cfg} {m}, 8 shards one after the other on this GPU ({v['label']}): full table {v['full_table_ms']:.3f} ms; {part} shards "
                f"{min(b['shard_kernel_ms']):.3f}–{max(b['shard_kernel_ms']):.3f} ms -> {b['speedup_kernel_only']:.2f}x kernels only, "
                f"{b['predicted_speedup_overlapped']:.2f}x with the gather posted (measured, overlapped), {b['predicted_speedup_serial']:.2f}x with a serial gather + "
                f"the assumed wire ({v['wire_ms_assumed'] * 1e3:.0f} us)")
        c = v.get("contiguous") if part != "contiguous" else None
        if c:
            line += (f"; contiguous equal-state blocks {min(c['shard_kernel_ms']):.3f}–{max(c['shard_kernel_ms']):.3f} ms -> "
                     f"{c['predicted_speedup_overlapped']:.2f}x")
        print(line)
