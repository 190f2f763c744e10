"""Rewrite the round's numbers table of DESIGN.md section 6 (between the `<!-- r05-table -->` markers) from profiles/r05_*.json:
    python tools/design_numbers.py            # after copying a tools/profile_round.sh run into profiles/"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(name):
    f = os.path.join(P, name)
    if not os.path.exists(f):
        return None
    lines = [ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def kv(d):
    if d is None:
        return "(not in this run)"
    v = d["value"]
    return ("%.0f k" % (v / 1e3) if v < 1e6 else "%.2f M" % (v / 1e6)) + " edges/s, %.3f ms/step" % d["ms_per_step"]


def main():
    h = load("r05_headline_bench.json")
    r, f = h["roofline"], h.get("dominant_family") or {}
    tgn, dyg = load("r05_tgn_bench.json"), load("r05_dygformer_bench.json")
    tr = r.get("traffic")
    rows = [
        ("headline, `python bench.py` (native stepper)", "**%s** (812 k, 0.739 ms)" % kv(h),
         "%.1f %% of the path's HBM roofline (2 350 784 B per edge); main stream gap-free, 20 dispatches (+ the sampler's on the side stream).  Same-box A/Bs of the round: 0.736 → 0.70 ms"
         % (100 * h["path_roofline"]["hbm_frac"])),
        ("the same, `--python-step` (round 3's host path)", kv(load("r05_headline_bench_python_step.json")), "host issue with an idle GPU 403 µs per step against 171"),
        ("`exact_f32` / `strict` / `row_sharing_off`", " / ".join(kv(h.get(k)) for k in ("exact_f32", "strict", "row_sharing_off")),
         "§5: what parity at realistic weights costs"),
        ("attn_bwd roofline (`roofline` object)", "%.2f TB/s on §8(d) bytes = **%.3f** of 8 TB/s; %.2f TB/s with the activation bytes"
         % (r["achieved"] / 1e3, r["frac"], r.get("achieved_incl_activations", 0) / 1e3),
         "kernels unchanged since round 3; PMC traffic per launch %s (`profiles/traffic_r05.json`)" % ("%.0f MB" % ((tr["hbm_bytes_per_launch"] if isinstance(tr, dict) else tr) / 1e6) if tr else "see")),
        ("`dominant_family` (dense side)", "%.0f µs of the step (%.0f %%): %.1f GFLOP at %.1f TFLOP/s = %.2f of the f32-MFMA peak, %.3f of the bf16×3-equivalent peak"
         % (f.get("us_per_step", 0), 100 * f.get("share_of_step", 0), f.get("gflop_per_step", 0), f.get("tflops", 0), f.get("frac_of_f32_mfma_peak", 0),
            f.get("frac_of_bf16x3_peak", 0)), "chains + products + weight gradients, HIP events in an untimed second pass"),
        ("`--mode fwd` / `sweep` / `lp`", " / ".join(kv(load("r05_%s_bench.json" % m)) for m in ("fwd", "sweep", "lp")), "(1.82 M / 3.92 M / 518 k)"),
        ("`--workload scale` (config 5: 10 M nodes / 100 M edges, 75.7 GB of tables resident)", kv(load("r05_scale_config5_bench.json")),
         "(604 k); kernel table + timeline of this line: `profiles/r05_scale_config5_kernel_stats.csv`, `_timeline.txt`"),
        ("TGN (config 3), native step", "**%s** (1.58 M, 0.381 ms)" % kv(tgn),
         "`--mode lp` (negatives, then positives, MergeLayer head + BCE, Adam on both): %s; `--gemm-mode 0` (exact fp32 products): %s"
         % (kv(load("r05_tgn_lp_bench.json")), kv(load("r05_tgn_bench_exact_f32.json")))),
        ("TGN, `--simulate-world 8`", kv(load("r05_tgn_simulate_world8_bench.json")),
         "one rank of 8: its 600-edge shard embedded, the replicated state advanced with all 4 800 edges — the extra cost against the line above stays "
         "below the 15 % at which VERDICT r03 #7 asked for a sharded advance"),
        ("DyGFormer (config 4), native step", "**%s** (150–158 k, 3.80–3.99 ms)" % kv(dyg),
         "autograd path of the same build: %s; `--gemm-mode 0` (exact fp32 products): %s.  GPU-bound (69 launches back to back, host issues a step in 0.40 ms: "
         "`profiles/r05_dygformer_timeline.txt`, `_host_issue.txt`): products 0.85 ms (19 of 22 against pre-split weights, `tg_gemm_pk.hip`), weight gradients 0.40, "
         "attention core 0.38, element-wise passes 0.55"
         % (kv(load("r05_dygformer_bench_autograd.json")), kv(load("r05_dygformer_bench_exact_f32.json")))),
        ("CPU port (oracle, %s host threads)" % (h.get("cpu_baseline") or {}).get("cores", "?"),
         "%.0f (TGAT) edges/s" % (h.get("cpu_baseline") or {}).get("value", 0), "baseline only"),
    ]
    table = "| line | value | note |\n|---|---|---|\n" + "".join("| %s | %s | %s |\n" % r_ for r_ in rows)
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    a, b = s.index("<!-- r05-table -->") + len("<!-- r05-table -->"), s.index("<!-- /r05-table -->")
    open(p, "w").write(s[:a] + "\n" + table + s[b:])
    print(table)


if __name__ == "__main__":
    main()
