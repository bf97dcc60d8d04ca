#!/usr/bin/env python3
"""Speed-of-light table of the Conformer encoder layer's forward kernels: for every kernel of one
eval-mode layer the three bounds

    MFMA      flops / 2.5 PFLOP/s                                  (dense bf16 peak, MI355X_MICROARCH.md)
    HBM       algorithmic bytes / 6.3 TB/s                         (the measured copy rate; 8 TB/s is the data sheet's)
    L2->LDS   weight bytes one workgroup streams x rounds / rate   (every CU pulls its workgroup's weight panels through its own
                                                                    LDS-DMA path: 34 B/clk/CU beside MFMA + fragment reads — the
                                                                    rate the shipped loops reach, profiles/r05_rowblock_stamps.txt —
                                                                    and 58 B/clk/CU for the bare stream, tools/ubench/dma_stream.hip)

against the average launch duration in a `rocprofv3 --kernel-trace --stats` CSV of tools/enc_fwd_profile.py, at 64 x 1000 and at
256 x 1000 frames (same utilisation at 4 x the rows: occupancy is not what is missing).

usage: python tools/sol_table.py <stats_64.csv> <frames_64> [<stats_256.csv> <frames_256>]
       (frames = the real input frames of the profiled batch, printed by tools/enc_fwd_profile.py; rows = frames / 4)"""
import csv
import math
import sys

D, F, H, DK, TP, KCONV = 256, 2048, 4, 64, 250, 15
MFMA, HBM, CLK, CUS = 2.5e15, 6.3e12, 2.4e9, 256
L2_BESIDE, L2_BARE = 34.0, 58.0


def kernels(rows, B):
    """name -> (csv kernel-name prefix, launches per layer, flops, algorithmic HBM bytes, weight bytes per workgroup, workgroups)"""
    R = rows
    rb64, rb128 = math.ceil(R / 64), math.ceil(R / 128)
    split = 2 if 2 * rb128 <= CUS else 1
    t2 = B * TP * TP  # padded key x query pairs per head (the live ones are fewer: an upper bound on the work)
    return {
        "fused FFN (x 2)": ("void (anonymous namespace)::ffn_pc_kernel<0, 2, false", 2, 4.0 * R * F * D,
                            2 * R * D * 2 + 2 * F * D * 2, 2 * F * D * 2 / split, rb128 * split),
        "LN + QKV projection": ("void (anonymous namespace)::rowblock_gemm_kernel<false, false, false>", 1, 2.0 * R * D * 3 * D,
                                R * D * 2 + R * 3 * D * 2 + 3 * D * D * 2, 3 * D * D * 2, rb64),
        "rel-pos attention": ("(anonymous namespace)::attn_bh_fwd_kernel", 1, 6.0 * H * t2 * DK,
                              R * 3 * D * 2 + R * D * 2 + (2 * TP - 1) * D * 2, 0, B * H),
        "out-proj + LN + pw-conv 1 + GLU": ("void (anonymous namespace)::rowblock_chain_kernel<false, true>", 1, 2.0 * R * D * 3 * D,
                                            4 * R * D * 2 + 3 * D * D * 2, 3 * D * D * 2, rb64),
        "dw-conv + BN + act + pw-conv 2": ("void (anonymous namespace)::rowblock_gemm_kernel<false, false, true>", 1,
                                           2.0 * R * D * D + 2.0 * KCONV * R * D, 3 * R * D * 2 + D * D * 2, D * D * 2, rb64),
    }


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3)
    return out


def table(stats_path, frames, B):
    rows = frames // 4
    st = load(stats_path)
    print("\n### %d x 1000 frames: %d live rows after the subsampler (%s)\n" % (B, rows, stats_path))
    print("| kernel (per encoder layer, eval) | MFMA µs | HBM µs | L2→LDS µs (34 / 58 B/clk/CU) | bound µs | measured µs | measured ÷ bound |")
    print("|---|---|---|---|---|---|---|")
    tot_b = tot_m = 0.0
    passes = None
    for name, (prefix, per_layer, flops, hbm, wbytes, wgs) in kernels(rows, B).items():
        hit = [(k, v) for k, v in st.items() if k.startswith(prefix)]
        meas = sum(v[2] for _, v in hit) / max(sum(v[0] for _, v in hit), 1) if hit else float("nan")
        if hit and passes is None and per_layer == 2:
            passes = sum(v[0] for _, v in hit) / 24.0
        rounds = math.ceil(wgs / CUS)
        t_mfma = flops / MFMA * 1e6   # (flops, bytes: per launch)
        t_hbm = hbm / HBM * 1e6
        t_l2a = wbytes * rounds / (L2_BESIDE * CLK) * 1e6
        t_l2b = wbytes * rounds / (L2_BARE * CLK) * 1e6
        bound = max(t_mfma, t_hbm, t_l2a)
        tot_b += per_layer * bound
        tot_m += per_layer * meas
        print("| %s | %.1f | %.1f | %.1f / %.1f | %.1f | %.1f | %.1f x |" % (name, t_mfma, t_hbm, t_l2a, t_l2b, bound, meas, meas / bound))
    enc_flop = 18.0e6 * frames  # 18.0 MFLOP per input frame
    print("| **one layer** | | | | **%.1f** | **%.1f** | **%.1f x** |" % (tot_b, tot_m, tot_m / tot_b))
    print("\n12 layers: bound %.2f ms, measured %.2f ms in these kernels; encoder forward = %.2f TFLOP on the real frames -> "
          "%.1f %% of the MFMA peak at the bounds, %.1f %% as measured (layers only; the subsampler and the CTC head come on top)."
          % (12 * tot_b / 1e3, 12 * tot_m / 1e3, enc_flop / 1e12, 100 * enc_flop / (12 * tot_b * 1e-6) / MFMA,
             100 * enc_flop / (12 * tot_m * 1e-6) / MFMA))


if __name__ == "__main__":
    a = sys.argv[1:]
    table(a[0], int(a[1]), 64)
    if len(a) >= 4:
        table(a[2], int(a[3]), 256)
