#!/usr/bin/env python3
"""Tables from the rocprofv3 PMC passes of tools/pmc_bench_step.sh (bench.py's nominal training step, every kernel):

  * per kernel instantiation and per step: launches, un-perturbed duration (the plain --kernel-trace pass), HBM bytes =
    2 x FETCH_SIZE + WRITE_SIZE (KB counters; gfx950 tallies a wide coalesced read at half its bytes -- MI355X_MICROARCH.md, HBM --
    the same correction on every kernel, uncalibrated for narrow accesses), matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES /
    (duration x clock x 1024 SIMDs) with the clock of THAT pass from GRBM_GUI_ACTIVE where available, wave-cycle split
    (SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES), VALU / LDS instruction counts per wave, the
    LDS bank-conflict share SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, occupancy inputs (VGPRs, LDS, workgroup size);
  * per kernel FAMILY (tools/step_model.py): algorithmic bytes, measured bytes, HBM-floor / MFMA-floor / measured microseconds.

usage: tools/pmc_step_table.py <dir> [--gemm-out profiles/r05_gemm_pmc.txt] [--step-out profiles/r05_step_roofline.txt] [--json profiles/step_pmc.json]
Steps are delimited by adam_step_kernel; the first two steps of each pass (warm-up) are dropped."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import step_model  # noqa: E402

d = sys.argv[1]


def opt(name):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else None


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    if n.startswith("dwconv") or n.startswith("at::") or "rocclr" in n:
        n = re.sub(r"<.*", "", n) if not n.startswith("dwconv") else n
    return n[:110]


def steps_of(path, counter_pass):
    """[{kernel: {"n": launches, "dur_ns": total, counter: total ...}}, ...] one dict per timed step"""
    rows = list(csv.DictReader(open(path)))
    if not rows:
        return []
    disp = collections.OrderedDict()
    for r in rows:
        key = int(r["Dispatch_Id"])
        e = disp.get(key)
        if e is None:
            e = disp[key] = {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]), "ctr": {},
                             "vgpr": r.get("VGPR_Count"), "agpr": r.get("Accum_VGPR_Count"), "lds": r.get("LDS_Block_Size"),
                             "wg": r.get("Workgroup_Size") or r.get("Workgroup_Size_X"), "grid": r.get("Grid_Size") or r.get("Grid_Size_X")}
        if counter_pass:
            e["ctr"][r["Counter_Name"]] = e["ctr"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    seq = sorted(disp.values(), key=lambda e: e["start"])
    adam = [i for i, e in enumerate(seq) if e["name"].startswith("adam_step")]
    out = []
    for s in range(2, len(adam) - 1):
        st = collections.defaultdict(lambda: collections.defaultdict(float))
        for e in seq[adam[s] + 1: adam[s + 1] + 1]:
            k = short(e["name"])
            st[k]["n"] += 1
            st[k]["dur_ns"] += e["end"] - e["start"]
            for c, v in e["ctr"].items():
                st[k][c] += v
            st[k]["_meta"] = (e["vgpr"], e["agpr"], e["lds"], e["wg"], e["grid"])
        out.append(st)
    return out


def mean_steps(steps):
    """per-step mean over the timed steps: {kernel: {key: mean}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for st in steps:
        for k, dct in st.items():
            for c, v in dct.items():
                if c == "_meta":
                    meta[k] = v
                else:
                    acc[k][c] += v / len(steps)
    return acc, meta


P = {}
for name in ("trace", "sq1", "sq2", "fetch", "write", "grbm"):
    f = os.path.join(d, name + ".csv")
    P[name] = mean_steps(steps_of(f, name != "trace")) if os.path.exists(f) else ({}, {})
tr, meta = P["trace"]
if not tr:
    raise SystemExit("no trace.csv in " + d)
kernels = sorted(tr, key=lambda k: -tr[k]["dur_ns"])


def g(pass_, k, c):
    return P[pass_][0].get(k, {}).get(c, 0.0)


def clock_ghz(pass_, k):
    """effective shader clock of kernel k in that pass: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration -- reads high on
    dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back), so 2.4 is the cap"""
    gui, dur = g("grbm", k, "GRBM_GUI_ACTIVE"), g("grbm", k, "dur_ns")
    return min(2.4, gui / 8.0 / dur) if gui and dur else 2.1


lines = []
W = lines.append
nsteps = len(steps_of(os.path.join(d, "trace.csv"), False))
tot_us = sum(tr[k]["dur_ns"] for k in kernels) / 1e3
W(f"# rocprofv3 PMC passes over `python3 bench.py --diag-no-timestretch ...` (tools/pmc_bench_step.sh): nominal step, B = 32 x T = 1024, bf16, level 5")
W(f"# {nsteps} timed steps per pass; {sum(tr[k]['n'] for k in kernels):.0f} launches / step, {tot_us:.0f} us of kernel time / step in the plain kernel-trace pass")
W("# us/step and us/launch: the plain --kernel-trace pass (no counters); every counter column: its own pass (kernels run serialised and")
W("# slower under --pmc, so counter RATIOS are what to read).  HBM MB = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch.")
W("# MFMA% = SQ_VALU_MFMA_BUSY_CYCLES / (sq1-pass duration x 2.1 GHz x 1024 SIMDs); wait% / stall% / issue% = SQ_WAIT_ANY / SQ_WAIT_INST_ANY /")
W("# SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; VALU/wave, LDS/wave = SQ_INSTS_VALU, SQ_INSTS_LDS / SQ_WAVES; bank% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;")
W("# LDSbusy% = SQ_LDS_IDX_ACTIVE / (sq2-pass duration in cycles x 256 CUs) (counter summed over CUs; 4 LDS-array cycles per ds_read_b128)")
hdr = f"{'kernel':64s} {'n':>4s} {'us/step':>8s} {'us/launch':>9s} {'HBM MB':>8s} {'GB/s':>6s} {'MFMA%':>6s} {'wait%':>6s} {'stall%':>6s} {'issue%':>6s} {'VALU/w':>7s} {'LDS/w':>6s} {'bank%':>6s} {'LDSbusy%':>8s} {'vgpr':>5s} {'lds KB':>6s} {'wg':>5s}"


def row(k):
    n = tr[k]["n"]
    us = tr[k]["dur_ns"] / 1e3
    mb = (2 * g("fetch", k, "FETCH_SIZE") + g("write", k, "WRITE_SIZE")) * 1024 / 1e6 / max(n, 1e-9)
    gbs = mb * 1e6 / (us / n * 1e-6) / 1e9 if us else 0.0
    d1 = g("sq1", k, "dur_ns")
    mf = g("sq1", k, "SQ_VALU_MFMA_BUSY_CYCLES") / (d1 * 2.1 * 1024) * 100 if d1 else 0.0
    wc = g("sq1", k, "SQ_WAVE_CYCLES") or 1.0
    wa, wi, ai = (g("sq1", k, c) / wc * 100 for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
    wv = g("sq2", k, "SQ_WAVES") or 1.0
    vw, lw = g("sq2", k, "SQ_INSTS_VALU") / wv, g("sq2", k, "SQ_INSTS_LDS") / wv
    ia = g("sq2", k, "SQ_LDS_IDX_ACTIVE")
    bank = g("sq2", k, "SQ_LDS_BANK_CONFLICT") / ia * 100 if ia else 0.0
    d2 = g("sq2", k, "dur_ns")
    ldsb = ia / (d2 * 2.1 * 256) * 100 if d2 else 0.0
    m = meta.get(k, ("", "", "", "", ""))
    vg = f"{m[0]}+{m[1]}" if m[1] not in (None, "", "0") else f"{m[0]}"
    return (f"{k[:64]:64s} {n:4.1f} {us:8.1f} {us / max(n, 1e-9):9.1f} {mb:8.1f} {gbs:6.0f} {mf:6.1f} {wa:6.1f} {wi:6.1f} {ai:6.1f} {vw:7.0f} {lw:6.0f} "
            f"{bank:6.1f} {ldsb:8.1f} {vg:>5s} {int(m[2] or 0) / 1024:6.1f} {m[3]:>5s}")


gemm = [k for k in kernels if step_model.family_of(k) in ("pw_gemm", "pw_wgrad")]
gl = list(lines)
gl.append("")
gl.append("## 1x1-GEMM family (K1): every instantiation the nominal step dispatches")
gl.append(hdr)
for k in gemm:
    gl.append(row(k))
gt = sum(tr[k]["dur_ns"] for k in gemm) / 1e3
gl.append(f"{'sum':64s} {sum(tr[k]['n'] for k in gemm):4.0f} {gt:8.1f}")
gl.append("")
gl.append("## every other kernel of the step")
gl.append(hdr)
for k in kernels:
    if k not in gemm:
        gl.append(row(k))

# ---- family table
rows = step_model.step_rows()
fam = step_model.by_family(rows)
meas = collections.defaultdict(lambda: {"n": 0.0, "us": 0.0, "mb": 0.0})
for k in kernels:
    f = step_model.family_of(k)
    meas[f]["n"] += tr[k]["n"]
    meas[f]["us"] += tr[k]["dur_ns"] / 1e3
    meas[f]["mb"] += (2 * g("fetch", k, "FETCH_SIZE") + g("write", k, "WRITE_SIZE")) * 1024 / 1e6
sl = []
S = sl.append
S("# Step-level roofline of the bench workload's NOMINAL step (B = 32 x T = 1024, bf16 GEMM operands, activation storage level 5).")
S("# algorithmic MB / GFLOP: tools/step_model.py (every operand read once, every result written once, in the step's storage formats;")
S("# 1x1-GEMM flops only).  measured MB: rocprofv3 PMC, 2 x FETCH_SIZE + WRITE_SIZE (separate passes).  measured us: plain kernel trace.")
S("# HBM floor = algorithmic bytes / 8 TB/s; MFMA floor = flops / 2.5 PFLOP/s (MI355X_MICROARCH.md).")
S(f"{'family':10s} {'launches':>8s} {'algo MB':>9s} {'meas MB':>9s} {'meas/algo':>9s} {'GFLOP':>8s} {'HBM floor us':>12s} {'MFMA floor us':>13s} {'measured us':>11s} {'x floor':>8s} {'frac 8TB/s':>10s}")
ta = tm = tf = tu = 0.0
for f in sorted(set(fam) | set(meas), key=lambda f: -meas[f]["us"]):
    a = fam.get(f, {"bytes": 0.0, "flops": 0.0})
    m = meas[f]
    hf, mf = a["bytes"] / 8e12 * 1e6, a["flops"] / 2.5e15 * 1e6
    fl = max(hf, mf)
    S(f"{f:10s} {m['n']:8.1f} {a['bytes'] / 1e6:9.1f} {m['mb']:9.1f} {m['mb'] / (a['bytes'] / 1e6) if a['bytes'] else float('nan'):9.2f} {a['flops'] / 1e9:8.1f} "
      f"{hf:12.1f} {mf:13.1f} {m['us']:11.1f} {m['us'] / fl if fl else float('nan'):8.1f} {a['bytes'] / (m['us'] * 1e-6) / 8e12 if m['us'] else 0:10.3f}")
    ta += a["bytes"]; tm += m["mb"]; tf += a["flops"]; tu += m["us"]
S(f"{'step':10s} {sum(m['n'] for m in meas.values()):8.1f} {ta / 1e6:9.1f} {tm:9.1f} {tm / (ta / 1e6):9.2f} {tf / 1e9:8.1f} {ta / 8e12 * 1e6:12.1f} {tf / 2.5e15 * 1e6:13.1f} {tu:11.1f} "
  f"{tu / (ta / 8e12 * 1e6):8.1f} {ta / (tu * 1e-6) / 8e12:10.3f}")
S("")
S("# model rows (tools/step_model.py)")
for r in rows:
    S(f"#   {r['family']:9s} {r['bytes'] / 1e6:8.1f} MB {r['flops'] / 1e9:7.1f} GFLOP  {r['what']}")

for path, text in ((opt("--gemm-out"), gl), (opt("--step-out"), sl)):
    if path:
        open(path, "w").write("\n".join(text) + "\n")
    else:
        print("\n".join(text))
        print()
jout = opt("--json")
if jout:
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "voice100_amd", "csrc", "*"))):
        h.update(open(f, "rb").read())
    json.dump({"csrc_sha": h.hexdigest()[:16], "bytes_measured_per_step": round(tm * 1e6), "kernel_us_per_step_in_trace": round(tu, 1),
               "families": {f: {"launches": round(m["n"], 1), "bytes_measured": round(m["mb"] * 1e6), "us": round(m["us"], 1)} for f, m in meas.items()},
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --diag-no-timestretch` "
                         "(nominal step), 2*FETCH_SIZE + WRITE_SIZE KB summed over every kernel of a step; tools/pmc_bench_step.sh + tools/pmc_step_table.py"},
              open(jout, "w"), indent=1)
