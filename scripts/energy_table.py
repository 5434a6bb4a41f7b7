"""Per-kernel time and energy of one encoder layer at 131072 tokens, from the rocprofv3 kernel stats under profiles/
(usage: python scripts/energy_table.py [tag]  ->  profiles/<tag>_encoder_energy_table.txt on stdout).

The package sits at its 1400 W cap while the encoder runs (profiles/r*_power_tail_ablate0.log; idle 250 W), so a kernel's
energy per layer is its time x 1150 W of dynamic power, whatever it does with it.  The split into MFMA / HBM / rest is an
ESTIMATE from coefficients measured on this chip in earlier rounds (DESIGN.md sections 3.2b and 4):
  MFMA + its LDS fragment reads ~0.8 pJ/flop   (between scan16's 0.53 -- small-valued operands, 1.86 PFLOP/s at 1236 - 250 W -- and the
                                                tail kernel's 1.0 on random activations, 1.09 PFLOP/s at 1335 - 250 W; the QK launch's
                                                whole budget is 1.05 pJ/flop, which bounds it from above)
  HBM <-> LDS / registers       ~118 pJ/byte   (scan16's DMA stream alone: 7.1 TB/s at 1088 - 250 W)
  rest = L2 / Infinity Cache -> LDS operand traffic (W1 alone fetches 1.67 GB per launch past the L2), VALU (GELU, softmax,
  LayerNorm), and stall time (a stalled chip still burns much of its dynamic power at these clocks)
'must move' bytes: operands read once + outputs written once (what a launch cannot avoid)."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
T = 131072
P_DYN = 1150.0  # W
PJ_FLOP, PJ_BYTE = 0.8e-12, 118e-12


def rows(path):
    out = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            n = r["Name"].replace("void mx::", "").split("(")[0].replace(" ", "")
            out[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out


def table(title, stats, layers, passes, H, F, spec):
    print(title)
    print(f"{'kernel':30s} {'per layer':>9s} {'us':>8s} {'us/layer':>9s} {'J/layer':>8s} {'GFLOP':>8s} {'frac':>6s} {'MB moved':>9s} "
          f"{'MFMA J':>7s} {'HBM J':>6s} {'rest J':>7s}")
    tot_us = tot_j = tot_fl = 0.0
    for name, per_layer, gflop, mbytes, what in spec:
        if name not in stats:
            continue
        calls, us = stats[name]
        n = per_layer if per_layer else calls / (layers * passes)
        usl = us * n
        j = usl * 1e-6 * P_DYN
        jm, jh = gflop * 1e9 * PJ_FLOP, mbytes * 1e6 * PJ_BYTE
        frac = gflop * 1e9 / (usl * 1e-6) / 2.5e15 if usl > 0 else 0.0
        print(f"{name:30s} {n:9.2f} {us:8.1f} {usl:9.1f} {j:8.3f} {gflop:8.1f} {frac:6.3f} {mbytes:9.0f} {jm:7.3f} {jh:6.3f} {j - jm - jh:7.3f}  {what}")
        tot_us += usl
        tot_j += j
        tot_fl += gflop
    print(f"{'layer':30s} {'':9s} {'':8s} {tot_us:9.1f} {tot_j:8.3f} {tot_fl:8.1f} {tot_fl * 1e9 / (tot_us * 1e-6) / 2.5e15 if tot_us else 0:6.3f}\n")


bge = os.path.join(ROOT, "profiles", f"{tag}_encoder_bge_kernel_stats.csv")
if os.path.exists(bge):
    H, F = 768, 3072
    act = T * H * 2 / 1e6  # MB of one [tokens, hidden] bf16 matrix
    g = lambda n, k: 2.0 * T * n * k / 1e9
    # gscripts/gpu_encoder_prof.py bge: 1024 chunks = 4 passes of 131072 tokens, 3 encodes, 12 layers
    table(f"bge-base-en shape (hidden 768, ffn 3072), one layer at {T} tokens [{os.path.basename(bge)}]", rows(bge), 12, 12, H, F, [
        ("pgemm_kernel<2>", 1, g(2 * H, H), 3 * act, "QK projection"),
        ("pgemm_kernel<4>", 1, g(H, H), 2 * act, "V projection (feature-major)"),
        ("attention_kernel<64,1>", 1, 4.0 * T * 512 * H / 1e9, 4 * act, "softmax(QK^T)V, 512-token sequences"),
        ("pgemm_kernel<5>", 2, g(H, H) + g(H, F), (3 * act) + (T * F * 2 / 1e6 + 2 * act), "out-projection + W2, each + bias + residual (two launches)"),
        ("ln_rows_kernel<32>", 2, 0.0, 4 * act, "the two LayerNorms, in place"),
        ("pgemm_kernel<1>", 1, g(F, H), act + T * F * 2 / 1e6, "W1 + GELU"),
    ])
l6 = os.path.join(ROOT, "profiles", f"{tag}_encoder_kernel_stats.csv")
if os.path.exists(l6):
    H, F = 384, 1536
    act = T * H * 2 / 1e6
    g = lambda n, k: 2.0 * T * n * k / 1e9
    st = rows(l6)
    qk = "pgemm_kernel<2>" if "pgemm_kernel<2>" in st else "gemm_kernel<2,2,2,2,32,4>"
    att32 = "attention_kernel<32,1>" if "attention_kernel<32,1>" in st else "attention_kernel<32,2>"
    # scripts/gpu_encoder_prof.py l6: 2048 chunks = 8 passes, 3 encodes, 6 layers
    table(f"all-MiniLM-L6-v2 shape (hidden 384, ffn 1536), one layer at {T} tokens [{os.path.basename(l6)}]", st, 6, 24, H, F, [
        (qk, 1, g(2 * H, H), 3 * act, "QK projection"),
        ("gemm_kernel<4,2,2,2,32,4>", 1, g(H, H), 2 * act, "V projection (feature-major)"),
        (att32, 1, 4.0 * T * 512 * H / 1e9, 4 * act, "softmax(QK^T)V, 512-token sequences"),
        ("tail_kernel<true>", 1, g(H, H) + 2 * g(F, H), 3 * act, "out-projection + LayerNorm + MLP + LayerNorm, fused"),
    ])
