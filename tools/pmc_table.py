#!/usr/bin/env python3
"""Per-kernel table from the three rocprofv3 PMC passes of tools/pmc_passes.sh:
HBM bytes = 2 x FETCH_SIZE KB (gfx950 tallies a wide coalesced read at half its bytes, MI355X_MICROARCH.md, HBM) + WRITE_SIZE KB,
SQ wait / active fractions, matrix-pipe busy share.  Optionally writes the depthwise-forward traffic ratio bench.py reports.
usage: tools/pmc_table.py <dir with sq.csv fetch.csv write.csv> [--json profiles/r02_dw_fwd_pmc.json] [--B 32 --T 1024]"""
import collections
import csv
import json
import os
import re
import sys

d = sys.argv[1]
jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
B = int(sys.argv[sys.argv.index("--B") + 1]) if "--B" in sys.argv else 32
T = int(sys.argv[sys.argv.index("--T") + 1]) if "--T" in sys.argv else 1024


def load(name):
    """{kernel: {counter: mean value per dispatch}} (the warm-up dispatches of the micro-benchmark are identical launches);
    "_dur_ns" = mean dispatch duration, "_per_grid" = {grid size: {counter: mean}} (one kernel at several problem sizes)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    path = os.path.join(d, name + ".csv")
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        acc[k]["_dur_ns"].append(dur)
        per[k][r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[k][r["Grid_Size"]]["_dur_ns"].append(dur)
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    for k in out:
        out[k]["_per_grid"] = {g: {c: sum(v) / len(v) for c, v in cs.items()} for g, cs in per[k].items()}
    return out


sq, fe, wr = load("sq"), load("fetch"), load("write")
SPECS = [(256, 11, 2), (1024, 19, 1), (1024, 27, 1), (1024, 35, 1), (1024, 51, 1), (2048, 59, 1), (2048, 67, 1), (2048, 75, 1), (2048, 83, 1)]
print("HBM traffic per launch (MB) vs algorithmic bytes, depthwise kernels at B=%d, T=%d" % (B, T))
print(f"{'kernel':78s} {'read':>8s} {'write':>8s} {'algo rd':>8s} {'algo wr':>8s} {'ratio':>6s}")
tot_meas = tot_algo = 0.0
t = T
fwd_rows = []
for hid, k, s in SPECS:
    tout = (t - 1) // s + 1
    P16 = (t + 7) & ~7 if t < 256 else (t + 63) & ~63      # (csrc/common.h v100_pitch16 for B > 1)
    for kind, pat, rd, wrb in (("fwd", rf"dwconv(_mfma)?_kernel<{k}, (1, 0, 3, false, 0|{s}, 8, 1, 0, true, false)>", 4.0 * B * hid * t, 4.0 * B * hid * tout),
                               ("bwd fused", rf"dwconv(_mfma)?_kernel<{k}, 2, 2, 3, true, 0>", 4.0 * B * hid * (2 * tout + t), 4.0 * B * hid * t),
                               ("fwd16", rf"(dwconv_mfma_kernel<{k}, 1, 0, 2, false, 9>|dwconv_fwd16_stream_kernel<{k}, 2, \d+, \d+, \d+>)", 2.0 * B * hid * P16, 2.0 * B * hid * P16),
                               ("bwd16 fused", rf"(dwconv_mfma_kernel<{k}, 2, 2, 2, true, 15>|dwconv_bwd16_stream_kernel<{k}, 2, \d+, \d+, \d+>)", 2.0 * B * hid * 3 * P16, 2.0 * B * hid * P16)):
        names = [n for n in fe if re.search(pat, n)]
        if not names:
            continue
        n = names[0]
        r_mb = 2 * fe[n].get("FETCH_SIZE", 0.0) * 1024 / 1e6
        w_mb = wr.get(n, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
        ratio = (r_mb + w_mb) / ((rd + wrb) / 1e6)
        print(f"{n[:78]:78s} {r_mb:8.1f} {w_mb:8.1f} {rd / 1e6:8.1f} {wrb / 1e6:8.1f} {ratio:6.3f}")
        want = "fwd16" if any(("false, 9>" in q or "fwd16_stream" in q) for q in fe) else "fwd"      # the act16 kernels when the run exercised them
        if kind == want or (kind == "fwd" and s != 1):
            tot_meas += r_mb + w_mb
            tot_algo += (rd + wrb) / 1e6
            fwd_rows.append({"kernel": n, "read_mb": round(r_mb, 1), "write_mb": round(w_mb, 1), "ratio": round(ratio, 4)})
    t = tout
if tot_algo:
    print(f"forward, {len(fwd_rows)} launches: measured {tot_meas / 1e3:.3f} GB vs algorithmic {tot_algo / 1e3:.3f} GB -> ratio {tot_meas / tot_algo:.4f}")
    if jout:
        import glob
        import hashlib
        h = hashlib.sha256()
        for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "voice100_amd", "csrc", "depthwise*"))):
            h.update(open(f, "rb").read())
        json.dump({"ratio": round(tot_meas / tot_algo, 4), "kernel_src_sha": h.hexdigest()[:16], "launches": fwd_rows,
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 2*FETCH_SIZE + WRITE_SIZE KB (gfx950 correction), "
                             "tools/pmc_passes.sh + tools/pmc_table.py over tools/bench_kernels.py --what dw"}, open(jout, "w"), indent=1)
print()
print("SQ counters per kernel (fractions of SQ_WAVE_CYCLES; WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE = issuing;")
print("mfma_busy/sq_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, the raw ratio of the two counters as in profiles/r01k_pmc_counters.txt: for comparing builds)")
for n, c in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0 or not any(s in n for s in ("dwconv", "pw_gemm", "pw_wgrad", "log_mel")):
        continue
    busy = c.get("SQ_BUSY_CYCLES", 0)
    print(f"{n[:86]:86s} wait_any {100 * c.get('SQ_WAIT_ANY', 0) / wc:5.1f}%  wait_inst {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f}%  "
          f"active {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f}%  lds_conflict/wave_cycles {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / wc:5.2f}%  "
          f"mfma_busy/sq_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy if busy else 0:5.2f}")

print()
print("Matrix-pipe utilisation of the GEMM kernels per problem size: SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x 2.4 GHz x 1024 SIMDs)")
print("(busy cycles = 32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md; the duration is the PMC pass's own dispatch time, a few % longer than an unprofiled launch)")
for n, c in sorted(sq.items()):
    if not any(s_ in n for s_ in ("pw_gemm", "pw_wgrad")):
        continue
    for g, cc in sorted(c.get("_per_grid", {}).items(), key=lambda kv: int(kv[0])):
        busy, dur = cc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), cc.get("_dur_ns", 0.0)
        if busy and dur:
            print(f"{n[:80]:80s} grid {int(g):8d}: {dur / 1e3:7.1f} us  mfma busy {100 * busy / (dur * 2.4 * 1024):5.1f} %")
