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
    det = r.get("traffic_detail") or {}
    sim = load("r05_tgn_simulate_world8_bench.json")
    adv = ((sim or {}).get("distributed") or {}).get("tgn_state_advance") or {}
    dr, dp = (dyg or {}).get("roofline") or {}, (dyg or {}).get("path_roofline") or {}
    sc = load("r05_scale_config5_bench.json")
    rows = [
        ("headline, `python bench.py` (native stepper)", "**%s** (860 k, 0.698 ms)" % kv(h),
         "%.1f %% of the path's HBM roofline (2 350 784 B per edge); main stream gap-free, 18 dispatches (+ the sampler's on the side stream)"
         % (100 * h["path_roofline"]["hbm_frac"])),
        ("the same, `--python-step` (round 3's host path)", kv(load("r05_headline_bench_python_step.json")), "(845 k)"),
        ("`exact_f32` / `strict` / `row_sharing_off`", " / ".join(kv(h.get(k)) for k in ("exact_f32", "strict", "row_sharing_off")),
         "(576 k / 385 k / 569 k) §5: what parity at realistic weights costs; `--gemm-mode 0` as its own line: %s" % kv(load("r05_headline_bench_exact_f32.json"))),
        ("attn_bwd roofline (`roofline` object: layer-1 + root launch, in-line events)", "%.2f TB/s on §8(d) bytes = **%.3f** of 8 TB/s; %.2f TB/s with the activation bytes"
         % (r["achieved"] / 1e3, r["frac"], r.get("achieved_incl_activations", 0) / 1e3),
         "(0.373); the layer-1 launch alone (rocprof, `profiles/r05_headline_kernel_stats.csv`): 74.7 µs for ≈ 12.7 k instances = 4.7 TB/s = 0.59; PMC traffic per launch "
         "%s averaged over both launches, %s for the layer-1 launch against its 354 MB algorithmic (`profiles/traffic_r05.json`)"
         % ("%.0f MB" % (tr / 1e6) if tr else "n/a", "%.0f MB" % (det.get("hbm_bytes_layer1_launch", 0) / 1e6))),
        ("`dominant_family` (dense side)", "%.0f µs of the step (%.0f %%): %.1f GFLOP at %.1f TFLOP/s = %.2f of the f32-MFMA peak, %.3f of the bf16×3-equivalent peak"
         % (f.get("us_per_step", 0), 100 * f.get("share_of_step", 0), f.get("gflop_per_step", 0), f.get("tflops", 0), f.get("frac_of_f32_mfma_peak", 0),
            f.get("frac_of_bf16x3_peak", 0)), "(436 µs, 62 %, 0.095) chains + products + weight gradients, HIP events in an untimed second pass"),
        ("`--mode fwd` / `sweep` / `lp`", " / ".join(kv(load("r05_%s_bench.json" % m)) for m in ("fwd", "sweep", "lp")), "(1.84 M / 3.89 M / 540 k)"),
        ("`--workload scale` (config 5: 10 M nodes / 100 M edges, 75.7 GB of tables resident)", kv(sc),
         "(640 k); `exact_f32` %s; kernel table, timeline and SQ / TCC counters of this line: `profiles/r05_scale_config5_kernel_stats.csv`, `_timeline.txt`, `_pmc_sq.txt`"
         % ("%.0f k" % (((sc or {}).get("exact_f32") or {}).get("value", 0) / 1e3))),
        ("TGN (config 3), native step", "**%s** (1.69 M, 0.355 ms)" % kv(tgn),
         "`--mode lp` (negatives, then positives, MergeLayer head + BCE, Adam on both): %s; `--gemm-mode 0` (exact fp32 products): %s"
         % (kv(load("r05_tgn_lp_bench.json")), kv(load("r05_tgn_bench_exact_f32.json")))),
        ("TGN, `--simulate-world 8`", kv(sim),
         "(1.54 M) one rank of 8: its 600-edge shard embedded, the replicated state advanced with all 4 800 edges: the advance itself %.1f µs = %.1f %% of that step "
         "(`distributed.tgn_state_advance`, HIP events)" % (1e3 * adv.get("state_advance_ms_per_step", 0), 100 * adv.get("state_advance_share_of_step", 0))),
        ("DyGFormer (config 4), native step", "**%s** (241 k, 2.489 ms)" % kv(dyg),
         "autograd path of the same build: %s; `--gemm-mode 0` (exact fp32 products): %s.  `roofline.frac` %.2f against the f32-input MFMA peak = "
         "**%.3f** against what the split-bf16 kernels could issue (`frac_bf16x3_equivalent`; `path_roofline` %.2f / %.3f)"
         % (kv(load("r05_dygformer_bench_autograd.json")), kv(load("r05_dygformer_bench_exact_f32.json")), dr.get("frac", 0),
            dr.get("frac_bf16x3_equivalent", 0), dp.get("mfma_frac", 0), dp.get("mfma_frac_bf16x3_equivalent", 0))),
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
