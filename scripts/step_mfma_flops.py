#!/usr/bin/env python3
"""Matrix-pipe FLOPs of ONE training step from a rocprofv3 PMC pass (round 6: the counter behind roofline.flop_per_launch):

    rocprofv3 --pmc SQ_INSTS_MFMA -d /tmp/m -o p -- python scripts/probe_step.py fp32 256 8 2
    python scripts/step_mfma_flops.py <mfma.db> <steps> <key> [out.json] [summary.txt]

SQ_INSTS_MFMA counts wave-level MFMA instructions; the FLOPs of one instruction depend on its shape, which is fixed per kernel:
    v_mfma_f32_32x32x2_f32      2 * 32 * 32 * 2        = 4 096    every fp32 convolution / GEMM kernel of the library
    v_mfma_f32_4x4x1_16B_f32    2 * 4 * 4 * 1 * 16     =   512    the thin-channel 7x7 kernels (conv_small.hip: thin_out, wgrad_thin)
    v_mfma_f32_32x32x16_{bf16,f16}  2 * 32 * 32 * 16   = 32 768   every 16-bit kernel (names carry "16")
Merges {key: {"mfma_flop_per_step", ...}} into out.json (default profiles/r06_step_traffic.json), bound to the build like the traffic entry."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_traffic import load


def flop_per_inst(name):
    if "thin_out" in name or "wgrad_thin_kernel" in name or "co4" in name:
        return 512.0
    if "16" in name.split("<")[0]:
        return 32768.0
    return 4096.0


def main():
    db, steps, key = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(root, "profiles", "r06_step_traffic.json")
    summary = sys.argv[5] if len(sys.argv) > 5 else None
    per = load(db, "SQ_INSTS_MFMA")
    rows = sorted(((n, v[0] / steps, v[1] / steps, v[1] / steps * flop_per_inst(n)) for n, v in per.items() if v[1] > 0), key=lambda r: -r[3])
    total = sum(r[3] for r in rows)
    try:
        cur = json.load(open(out))
    except Exception:
        cur = {}
    ent = cur.get(key, {})
    ent["mfma_flop_per_step"] = total
    ent["mfma_method"] = "rocprofv3 --pmc SQ_INSTS_MFMA over scripts/probe_step.py; wave-level instructions x FLOPs of the kernel's MFMA shape (scripts/step_mfma_flops.py)"
    try:
        ent["mfma_lib_md5"] = hashlib.md5(open(os.path.join(root, "acl-gan_amd", "libaclgan_hip.so"), "rb").read()).hexdigest()
    except OSError:
        ent["mfma_lib_md5"] = None
    cur[key] = ent
    json.dump(cur, open(out, "w"), indent=1, sort_keys=True)
    lines = ["# %s: matrix-pipe FLOPs per step from SQ_INSTS_MFMA (%d traced steps): %.4f TFLOP per step" % (key, steps, total / 1e12),
             "%-64s %9s %14s %8s %12s" % ("kernel", "calls/st", "MFMA inst/st", "FLOP/in", "GFLOP/st")]
    for n, c, i, f in rows[:40]:
        lines.append("%-64s %9.1f %14.0f %8.0f %12.2f" % (n[:64], c, i, flop_per_inst(n), f / 1e9))
    txt = "\n".join(lines)
    print(txt)
    if summary:
        open(summary, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
