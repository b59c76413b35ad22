#!/usr/bin/env python3
"""Algorithmic bytes and matrix-pipe flops of ONE nominal training step of the bench workload (round-4 review, item 2).

The bench step is asr_en_base = AudioToTextCTC(64, 512, 29, 512) (voice100/models/asr.py:62-116) at B = 32 x T = 1024 frames,
bf16 GEMM operands, activation storage level 5: nine InvertedResidual blocks (asr.py:40-59: 1x1 expand ->
BatchNorm -> ReLU6 -> depthwise k -> BatchNorm -> ReLU6 -> 1x1 project -> BatchNorm (+ x)), the vocabulary head, log_softmax +
CTC, and their backward, Adam.  "Algorithmic" = every operand of a kernel read ONCE and every result written ONCE, in the
storage format the step uses (SURVEY.md 8d's convention for the depthwise kernel, applied to every kernel family): re-reads of
a tile by several workgroups, the partial slabs of the split weight gradients (written, re-read and reduced: ~1.2 GB a step that a
decomposition without a batch split would not move -- family "slab" has 0 algorithmic bytes) and cache effects are NOT in it --
they are what `bytes_measured` (rocprofv3 PMC, tools/pmc_step_table.py) shows on top.

    rows = step_rows(B, T)          # one row per launch group: family, what, launches, bytes, flops (1x1 GEMM flops only)
    fam  = by_family(rows)          # {family: {"launches", "bytes", "flops"}}

Families (the keys bench.py's `roofline_step` and DESIGN.md's table use, matched to kernel names by FAMILY_OF below):
  pw_gemm    1x1 convolutions forward and backward-data (K1 "NN")          pw_wgrad   their weight gradients (K1 "NT")
  slab       deterministic reduction of the weight-gradient partial slabs   dw_fwd / dw_bwd   depthwise (K2), forward / fused backward
  bn_pass    block-boundary passes (BatchNorm-3 forward affine + residual, BatchNorm-3 backward)
  bn_fin     BatchNorm finalisers (per-channel vectors only)                 ctc        log-sum-exp, lattice, gradient
  optim      fused Adam + per-step weight preparation (bf16 / transposed copies)
  edge       model-edge passes: augmentation (both layouts), the logit transpose, dropout forward, head bias / misc torch elementwise
"""
import re

SPEC = [  # (cin, cout, k, stride, residual)   asr.py:62-82 with hidden_size = 512, audio_size = 64
    (64, 256, 11, 2, False),
    (256, 256, 19, 1, True), (256, 256, 27, 1, True), (256, 256, 35, 1, True),
    (256, 512, 51, 1, False),
    (512, 512, 59, 1, True), (512, 512, 67, 1, True), (512, 512, 75, 1, True),
    (512, 512, 83, 1, False),
]
VOCAB, N_MEL, TEXT_LEN = 29, 64, 100
N_PARAMS = 11621661


def pitch16(t):
    """csrc/common.h v100_pitch16 for B > 1 (kept as arithmetic here: this model also runs where the library is not built)"""
    return (t + 7) & ~7 if t < 256 else (t + 63) & ~63


def wgrad_splits(B, M, K, target=512):
    tiles = -(-M // 128) * -(-K // 128)
    return max(1, min(B, -(-target // tiles)))


def step_rows(B=32, T=1024):
    rows = []

    def add(family, what, nbytes, flops=0.0, launches=1):
        rows.append({"family": family, "what": what, "launches": launches, "bytes": float(nbytes), "flops": float(flops)})

    # round 6: the augmentation pass also writes the [B, 64, T] twin the encoder takes (asr.py:111): no transpose launch
    add("edge", "augmentation (audio.py:52-108): log-mel in, augmented log-mel out in both layouts ([B,T,64] and [B,64,T], asr.py:111)", 3 * 4 * B * T * N_MEL)
    add("optim", "weight preparation: fp32 weights -> bf16 + transposed bf16 copies (1x1 weights only)", 0)      # filled below
    t = T
    w_pw = 0
    for i, (cin, cout, k, s, res) in enumerate(SPEC):
        hid = 4 * cin
        tout = (t - 1) // s + 1
        a16 = s == 1                                   # the stride-2 opener keeps fp32 storage (v100_ir_act16_supported)
        e = 2 if a16 else 4                            # bytes per stored hidden sample
        Pi, Po = (pitch16(t), pitch16(tout)) if a16 else (t, tout)
        xin = B * cin * (pitch16(t) * 2 if i > 1 else t * 4)      # block input as its GEMMs read it: the bf16 shadow from block 2 on
        parts_i, parts_o = B * -(-t // 128), B * -(-tout // 128)
        fl1, fl3 = 2.0 * hid * cin * B * t, 2.0 * cout * hid * B * tout
        w_pw += hid * cin + cout * hid
        L = f"L{i} k={k}"
        # ---- forward
        add("pw_gemm", f"{L} expand forward: x, W1 -> a1 + BN1 partial sums", xin + hid * cin * 2 + B * hid * Pi * e + parts_i * hid * 8, fl1)
        add("dw_fwd", f"{L} depthwise forward: a1 -> a2 (BN1 + ReLU6 on load, BN2 sums)", e * B * hid * (Pi + Po) + 4 * hid * k + 8 * hid)
        add("pw_gemm", f"{L} project forward: a2, W3 -> a3 + BN3 partial sums (BN2 + ReLU6 on load)",
            B * hid * Po * e + cout * hid * 2 + B * cout * Po * e + parts_o * cout * 8, fl3)
        # boundary pass: y = BN3(a3) (+ x); level 5: interior blocks write only the bf16 copy, the stack's last block also fp32
        last = i == len(SPEC) - 1
        yb = B * cout * (pitch16(tout) * 2 + (tout * 4 if (last or i == 0) else 0))
        add("bn_pass", f"{L} BatchNorm-3 affine (+ residual): a3 (, x) -> y", B * cout * Po * e + (xin if res else 0) + yb)
        add("bn_fin", f"{L} BatchNorm finalisers forward (BN1, BN3; BN2 inside the depthwise kernel at >= 1024 channels)",
            parts_i * hid * 8 + parts_o * cout * 8 + 16 * (hid + cout), launches=2)
        # ---- backward
        # round 6: the gradient between two stride-1 blocks of the stack is bf16 (block.hip, v100_ir_stack_bwd); the gradient entering the
        # stack (last block) and the opener's stay fp32
        dy16 = 1 <= i <= len(SPEC) - 2
        dx16 = 2 <= i <= len(SPEC) - 1
        dy = B * cout * (pitch16(tout) * 2 if dy16 else tout * 4)
        add("bn_pass", f"{L} BatchNorm-3 backward: dy, a3 -> da3 (sums + affine in one launch)", dy + 2 * B * cout * Po * e)
        S3, S1 = wgrad_splits(B, cout, hid), wgrad_splits(B, hid, cin)
        add("pw_wgrad", f"{L} project weight gradient: da3, a2 -> dW3 (through {S3} partial slabs)", B * cout * Po * e + B * hid * Po * e + cout * hid * 4, fl3)
        add("slab", f"{L} dW3 slab reduction ({S3} slabs written + read: {(2 * S3) * cout * hid * 4 / 1e6:.0f} MB, not algorithmic)", 0)
        add("pw_gemm", f"{L} project backward-data: da3, W3^T, a2 (mask) -> dz2 + BN2-backward sums",
            B * cout * Po * e + cout * hid * 2 + 2 * B * hid * Po * e + parts_o * hid * 8, fl3)
        add("dw_bwd", f"{L} depthwise backward (data + weight fused): dz2, a2, a1 -> dz1, dWd",
            e * B * hid * (2 * Po + 2 * Pi) + 8 * hid * k + 8 * hid)
        add("pw_wgrad", f"{L} expand weight gradient: dz1, a1 (BN1-backward affine), x -> dW1 partial slabs",
            2 * B * hid * Pi * e + xin + hid * cin * 4, fl1)
        add("slab", f"{L} dW1 slab reduction ({S1} slabs written + read: {(2 * S1) * hid * cin * 4 / 1e6:.0f} MB, not algorithmic)", 0)
        if i > 0:                                      # the first block's input gradient is not needed (the audio is a leaf without grad)
            add("pw_gemm", f"{L} expand backward-data: dz1, a1 (affine), W1^T (, dy) -> dx",
                2 * B * hid * Pi * e + hid * cin * 2 + B * cin * (pitch16(t) * 2 if dx16 else t * 4) + (dy if res else 0), fl1)
        add("bn_fin", f"{L} BatchNorm finalisers backward (BN2 from the GEMM slab; BN1 inside the depthwise kernel)",
            parts_o * hid * 8 + 24 * hid, launches=1)
        t = tout
    # ---- head, loss
    Th = t
    # round 6: dropout's backward rides in the head's backward-data GEMM (one byte of mask per element there), the CTC gradient is
    # written in the pre-transpose layout (no backward transpose)
    add("edge", "dropout forward (asr.py:88): x in, dropped x + byte mask out", (4 + 4 + 1) * B * 512 * Th, launches=1)
    flh = 2.0 * VOCAB * 512 * B * Th
    add("pw_gemm", "vocabulary head forward: [29 x 512] x [512 x T'] + bias", 4 * B * 512 * Th + 4 * B * VOCAB * Th, flh)
    add("pw_gemm", "vocabulary head backward-data with the dropout mask in its epilogue", 4 * B * VOCAB * Th + (4 + 1) * B * 512 * Th, flh)
    add("pw_wgrad", "vocabulary head weight gradient", 4 * B * VOCAB * Th + 4 * B * 512 * Th, flh)
    add("edge", "logit transpose [B,V,T'] -> [B,T',V] (asr.py:114; its backward is a view of the CTC gradient)", 2 * 4 * B * VOCAB * Th, launches=1)
    S = 2 * TEXT_LEN + 1
    add("ctc", "log-sum-exp per frame, alpha / beta lattices, gradient (asr.py:148-152)",
        4 * B * Th * VOCAB * 3 + 2 * 4 * B * Th * S * 2, launches=3)
    # ---- optimiser
    add("optim", "fused Adam: p, g, m, v read; p, m, v written", 7 * 4 * N_PARAMS)
    for r in rows:
        if r["what"].startswith("weight preparation"):
            r["bytes"] = float(w_pw * 4 + 2 * w_pw * 2)
    return rows


def by_family(rows):
    out = {}
    for r in rows:
        f = out.setdefault(r["family"], {"launches": 0, "bytes": 0.0, "flops": 0.0})
        f["launches"] += r["launches"]
        f["bytes"] += r["bytes"]
        f["flops"] += r["flops"]
    return out


# kernel name (as rocprofv3 prints it) -> family
FAMILY_OF = [
    (r"pw_slab_reduce|slab_reduce_kernel|slab_sum", "slab"),
    (r"pw_wgrad", "pw_wgrad"),
    (r"pw_gemm|pw_smallk", "pw_gemm"),
    (r"dwconv_fwd16|dwconv_kernel<|dwconv_mfma_kernel<\d+, 1, 0|dwconv_generic", "dw_fwd"),
    (r"dwconv", "dw_bwd"),
    (r"chan_bn3_bwd|chan_affine2|chan_reduce2", "bn_pass"),
    (r"bn_.*finalize|bn_finalize", "bn_fin"),
    (r"ctc_", "ctc"),
    (r"adam_step|weight_prep", "optim"),
    (r".*", "edge"),
]


def family_of(kernel_name):
    for pat, fam in FAMILY_OF:
        if re.search(pat, kernel_name):
            return fam
    return "edge"


if __name__ == "__main__":
    rows = step_rows()
    fam = by_family(rows)
    tb, tf = sum(f["bytes"] for f in fam.values()), sum(f["flops"] for f in fam.values())
    print(f"nominal step B = 32 x T = 1024: {tb / 1e9:.3f} GB algorithmic, {tf / 1e12:.3f} TFLOP on the matrix pipe "
          f"(HBM floor {tb / 8e12 * 1e3:.3f} ms at 8 TB/s, MFMA floor {tf / 2.5e15 * 1e3:.3f} ms at 2.5 PFLOP/s)")
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["bytes"]):
        print(f"  {k:9s} {f['launches']:4d} launches {f['bytes'] / 1e6:9.1f} MB  {f['flops'] / 1e9:9.1f} GFLOP")
