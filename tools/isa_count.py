"""Static instruction count of the online kernel's steady-state loop, per record, from the compiler's own assembly:
    python tools/isa_count.py [NA] [--out profiles/rNN_issue_model.json] [--asm file.s] [-DFLAG ...]
(--out: where the model goes, NA = 11 only; --asm: count an assembly file made elsewhere; -D...: extra compile flags, for A/B forms)
Compiles dcarl_amd/csrc/trace_nwave_f32.hip to assembly (device only), finds the main loop of
trace_nwave_kernel<float, NA, 3, true, true> (the fenced default; the largest loop), takes its second table-path turn (one turn = the wave's own quads of one
round: 16 records of every lane up to round 5, 12 since round 6's PF 3; counted from the returning count atomics) and counts opcodes.  Together with the per-instruction issue costs measured by tools/ubench_issue.hip
at the kernel's waves per SIMD (profiles/rNN_ubench_issue_Nwaves.txt) and the LDS cycle table of /opt/skills/guides/MI355X_MICROARCH.md this
gives the two issue floors bench.py reports next to the HBM fraction (the kernel is VALU / LDS-issue bound, not HBM bound)."""
import json, os, re, subprocess, sys
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
OUT = argv[argv.index("--out") + 1] if "--out" in argv else None
ASM = argv[argv.index("--asm") + 1] if "--asm" in argv else None
DEFS = [a for a in argv if a.startswith("-D")]
pos = [a for i, a in enumerate(argv) if not a.startswith("-") and (i == 0 or argv[i - 1] not in ("--out", "--asm"))]
NA = int(pos[0]) if pos else 11
asm = ASM or "/tmp/nwave_f32%s.s" % "".join(d.replace("=", "").replace("-D", "_") for d in DEFS)
srcs = [os.path.join(REPO, "dcarl_amd/csrc", f) for f in ("trace_nwave_impl.h", "trace_common.h", "common.h")]
if not ASM and (not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(f) for f in srcs)):
    sys.path.insert(0, REPO)
    from dcarl_amd.build import SOURCE_FLAGS             # the unit's own shipped flags (the scheduling strategy)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-honor-nans", '-DDCARL_BUILD_ID="x"',
                           *SOURCE_FLAGS.get("trace_nwave_f32.hip", []), *DEFS,
                           "--cuda-device-only", "-S", os.path.join(REPO, "dcarl_amd/csrc/trace_nwave_f32.hip"), "-o", asm])
lines = open(asm).read().split("\n")
NWAVES = 4                                                    # waves per slice of the shipped f32 instances (round 6; 3 until then)
sym = f"_ZN5dcarl18trace_nwave_kernelIfLi{NA}ELi{NWAVES}ELb1ELb1E"      # <float, NA, waves, STEPS, FENCED (the default since round 4)>
start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(":") or (l.startswith(sym) and ": ;" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"\b(?:s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
a, b = max(loops, key=lambda t: t[1] - t[0])
inner = sorted(i for i in labels.values() if a < i < b)
segs = list(zip([a] + inner, inner + [b]))            # [compute turn 0, compute turn 1, table turn 0, table turn 1] in code order
def count(lo, hi):
    c = Counter()
    for x in body[lo:hi]:
        x = x.strip()
        if x and not x.startswith((".", ";")) and not x.endswith(":"):
            c[x.split()[0]] += 1
    return c
cands = [c for c in (count(lo, hi) for lo, hi in segs) if sum(c.values()) > 600]     # (the loop's short tail segment is not a turn)
# the table-path turns are the ones without v_rsq chains for the count roots: fewest v_rsq_f32
c = min(cands, key=lambda c: (c["v_rsq_f32_e32"], sum(c.values())))
REC = float(c["ds_add_rtn_u32"] or 16)        # one returning count atomic per record (count_quad): 4 quads x PF own quads per turn (16 at PF 4, 12 at PF 3)


def ubench_ns():
    """{f64_arith, cvt, rsq, other} from the newest profiles/rNN_ubench_issue_{NWAVES}waves.txt (that many waves per SIMD), else round 3's figures."""
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_ubench_issue_{NWAVES}waves.txt")), reverse=True):
        ns = {}
        for line in open(f):
            m = re.match(r"^(\S.*?)\s+([0-9.]+) ns per wave-instruction", line)
            if m and f"{NWAVES} waves" in line:
                ns[m.group(1).strip()] = float(m.group(2))
        need = ("v_fma_f64", "v_mul_f64", "v_max_f64", "v_cvt_f64_f32", "v_rsq_f32", "v_add_u32_e32", "v_xor_b32_e32", "v_bfe_i32", "v_cndmask_b32_e64 (sgpr mask)")
        if all(k in ns for k in need):
            return dict(f64_arith=round((ns["v_fma_f64"] + ns["v_mul_f64"] + ns["v_max_f64"]) / 3, 3), cvt=ns["v_cvt_f64_f32"], rsq=ns["v_rsq_f32"],
                        other=round((ns["v_add_u32_e32"] + ns["v_xor_b32_e32"] + ns["v_bfe_i32"] + ns["v_cndmask_b32_e64 (sgpr mask)"]) / 4, 3)), os.path.relpath(f, REPO)
    return dict(f64_arith=2.10, cvt=2.51, rsq=3.63, other=1.85), "profiles/r03_ubench_issue.txt"


ISSUE_NS, ISSUE_SRC = ubench_ns()
f64 = sum(v for k, v in c.items() if re.match(r"v_(fma|fmac|mul|add|max|min)_f64", k))
cvt = sum(v for k, v in c.items() if k.startswith("v_cvt_"))
rsq = c["v_rsq_f32_e32"]
valu = sum(v for k, v in c.items() if k.startswith("v_"))
lds = {k: v / REC for k, v in c.items() if k.startswith("ds_")}
out = dict(kernel=f"trace_nwave_kernel<float,{NA},{NWAVES},true>", records_per_turn=int(REC), waves_per_slice=NWAVES,
           valu_per_record=valu / REC, valu_f64_arith_per_record=f64 / REC, valu_cvt_per_record=cvt / REC, valu_rsq_per_record=rsq / REC,
           valu_other_per_record=(valu - f64 - cvt - rsq) / REC, lds_per_record=sum(lds.values()), lds_by_opcode_per_record=lds,
           salu_per_record=sum(v for k, v in c.items() if k.startswith("s_")) / REC,
           # ns per wave-instruction and SIMD at NWAVES waves per SIMD: tools/ubench_issue.hip, this round's run if its output is there
           issue_ns=ISSUE_NS, issue_ns_source=ISSUE_SRC,
           # LDS-array / issue cycles per wave-instruction (MI355X_MICROARCH.md, LDS table: 128 B per clock and CU; two-address and returning
           # operations as the sum of their halves), LDS clock under this load
           lds_cycles=dict(ds_read_b128=8, ds_read_b64=4, ds_read_b32=2, ds_write_b32=4, ds_write_b64=6, ds_write_b128=13, ds_read2st64_b64=8,
                           ds_write2st64_b64=12, ds_wrxchg_rtn_b64=10, ds_add_rtn_u32=6, ds_read2st64_b32=4), lds_clock_ghz=1.9,
           slices_per_cu=4)
os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
if NA == 11 and OUT:
    json.dump(out, open(os.path.join(REPO, OUT) if not os.path.isabs(OUT) else OUT, "w"), indent=1)
print(json.dumps(out, indent=1))
if "--ops" in argv:
    print({k: v / REC for k, v in sorted(c.items(), key=lambda t: -t[1]) if k.startswith("v_")})
