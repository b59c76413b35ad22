#!/usr/bin/env python3
"""profiles/dw_fwd_pmc.json from the PMC passes of tools/pmc_bench_dw.sh: one row per depthwise FORWARD launch of the bench step,
HBM bytes = 2 x FETCH_SIZE (gfx950 tallies a wide coalesced read at half its bytes, MI355X_MICROARCH.md, HBM) + WRITE_SIZE (KB),
against the launch's algorithmic bytes (sizeof * B * C * (T_in + T_out) + 4 C k + 8 C, sizeof = 2 for the bf16-stored
hidden tensors of activation storage level >= 1, 4 for the stride-2 opener).
usage: tools/pmc_dw_json.py <dir> [--json profiles/dw_fwd_pmc.json] [--B 32 --T 1024]"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

d = sys.argv[1]
jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
B = int(sys.argv[sys.argv.index("--B") + 1]) if "--B" in sys.argv else 32
T = int(sys.argv[sys.argv.index("--T") + 1]) if "--T" in sys.argv else 1024
SPECS = [(256, 11, 2), (1024, 19, 1), (1024, 27, 1), (1024, 35, 1), (1024, 51, 1), (2048, 59, 1), (2048, 67, 1), (2048, 75, 1), (2048, 83, 1)]
# forward kernels of the TRAINING step: the streaming kernel (rows <= 768 outputs, EV = false), the general Toeplitz-MFMA kernel in its
# bf16-storage forward form (IM 1, OM 0, IO 9), the fp32-storage MFMA forward, and the register-window kernel of the stride-2 opener
FWD = re.compile(r"dwconv_fwd16_stream_kernel<(\d+), \d+, \d+, \d+, \d+, false(?:, false)?>|dwconv_mfma_kernel<(\d+), 1, 0, \d+, false, (?:9|0)>|"
                 r"dwconv_kernel<(\d+), 2, 8, 1, 0, true, false>")


def load(name, counter):
    """{kernel: mean counter value per dispatch} over the dispatches of the run"""
    path = os.path.join(d, name + ".csv")
    acc = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def table(tag, t_in):
    fe, wr = load(tag + "_fetch", "FETCH_SIZE"), load(tag + "_write", "WRITE_SIZE")
    rows = []
    t = t_in
    for hid, k, s in SPECS:
        tout = (t - 1) // s + 1
        names = [n for n in fe if (m := FWD.search(n)) and int(next(g for g in m.groups() if g)) == k]
        # (several variants of one layer can only appear when the run mixed lengths: keep the one the run dispatched most)
        names = sorted(names, key=lambda q: -fe[q][1])[:1]
        for n in names:
            bf16 = "fwd16_stream" in n or re.search(r"false, 9>", n) is not None
            if bf16:
                algo = 2.0 * B * hid * (t + t) + 4 * hid * k + 8 * hid      # (algorithmic: the T samples of a row, not its pitch)
            else:
                algo = 4.0 * B * hid * (t + tout) + 4 * hid * k + 8 * hid
            rd = 2 * fe[n][0] * 1024
            w = wr.get(n, (0.0, 0))[0] * 1024
            rows.append({"k": k, "kernel": n, "dispatches": fe[n][1], "read_mb": round(rd / 1e6, 2), "write_mb": round(w / 1e6, 2),
                         "hbm_bytes": round(rd + w), "algorithmic_bytes": round(algo), "ratio": round((rd + w) / algo, 4)})
        t = tout
    return rows


out = {}
for tag, t_in in (("nominal", T), ("stretch110", T * 110 // 100)):
    rows = table(tag, t_in)
    if not rows:
        continue
    print(f"== {tag}: input {t_in} frames, B = {B}")
    print(f"{'k':>3s} {'kernel':70s} {'n':>4s} {'read MB':>9s} {'write MB':>9s} {'algo MB':>9s} {'ratio':>7s}")
    for r in rows:
        print(f"{r['k']:3d} {r['kernel'][:70]:70s} {r['dispatches']:4d} {r['read_mb']:9.2f} {r['write_mb']:9.2f} {r['algorithmic_bytes'] / 1e6:9.2f} {r['ratio']:7.4f}")
    hb, ab = sum(r["hbm_bytes"] for r in rows), sum(r["algorithmic_bytes"] for r in rows)
    print(f"    {len(rows)} launches: measured {hb / 1e6:.1f} MB vs algorithmic {ab / 1e6:.1f} MB -> bytes-weighted ratio {hb / ab:.4f}")
    out[tag] = {"rows": rows, "ratio": round(hb / ab, 4)}
if jout and "nominal" in out:
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "voice100_amd", "csrc", "depthwise*"))):
        h.update(open(f, "rb").read())
    json.dump({"ratio": out["nominal"]["ratio"], "kernel_src_sha": h.hexdigest()[:16], "launches": out["nominal"]["rows"],
               "stretched_110": out.get("stretch110"),
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --diag-no-timestretch` "
                         "(and --diag-stretch-rate 110), 2*FETCH_SIZE + WRITE_SIZE KB (gfx950 correction); tools/pmc_bench_dw.sh + tools/pmc_dw_json.py"},
              open(jout, "w"), indent=1)
