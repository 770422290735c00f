#!/usr/bin/env python3
"""Condense tools/profile_bench.sh output into the small files committed under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(repo, "gpurun_out", f"profiles_{tag}")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("void ", "").replace("r4d::", "")
    return name.split("(")[0][:70]


stats = glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv"))
rows = list(csv.DictReader(open(stats[0]))) if stats else []
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --shape {os.environ.get('R4D_PROFILE_SHAPE', 'UCI_13')} "
            f"--gemm {os.environ.get('R4D_PROFILE_GEMM', 'f16x2')} --steps 32 --warmup 16 --random-pool --headline-only\n")
    f.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
    for r in rows:
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},"
                f"{r['MinNs']},{r['MaxNs']}\n")

def _class_rows():
    cls = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        c = bench_class(short(r["Name"]))
        cls[c][0] += int(r["Calls"]); cls[c][1] += float(r["TotalDurationNs"])
    return cls


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
with open(os.path.join(dst, f"{tag}_bench_pmc.csv"), "w") as f:
    f.write("# per-dispatch means; FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
            "(gfx950: FETCH_SIZE reads half of a wide coalesced stream, MI355X_MICROARCH.md section HBM)\n")
    f.write("kernel,counter,mean,dispatches\n")
    for k, cs in sorted(agg.items()):
        for c, v in sorted(cs.items()):
            f.write(f"\"{k}\",{c},{sum(v) / len(v):.1f},{len(v)}\n")
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            fe, wr = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            traffic[k] = {"hbm_bytes_per_launch": (2 * fe + wr) * 1024, "fetch_kib_raw": fe, "write_kib": wr}
            if "TCC_HIT_sum" in cs:
                h, m = sum(cs["TCC_HIT_sum"]), sum(cs["TCC_MISS_sum"])
                traffic[k]["l2_hit_rate"] = h / max(h + m, 1)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
                busy = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
                act = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
                traffic[k]["mfma_pipe_util"] = busy / 1024.0 / (act / 8.0)       # 1024 SIMDs, GUI_ACTIVE summed over 8 XCDs


def bench_class(k):
    """rocprof kernel name -> bench.py / r4d_profile_class_name class."""
    import re
    m = re.match(r"gemm_f32_kernel<(\d+), (\d+), \d+, \d+, \d+, (false|true)>", k)
    if m:
        return f"gemm_f32_{m.group(1)}x{m.group(2)}_{'nt' if m.group(3) == 'true' else 'nn'}"
    m = re.match(r"gemm_f32_kc_kernel<(\d+), (\d+), (\d+), \d+, \d+, \d+(?:, \d+)?>", k)       # (+ epilogue kind)
    if m:
        return f"gemm_f32_kc_{m.group(1)}x{m.group(2)}x{m.group(3)}"
    if re.match(r"gemm_s3_kernel<\d+, \d+, \d+, \d+, \d+, \d+, \d+, true, [1-9]", k):     # SCAN order: the scoring tiles of Q >= 64 (score.hip)
        return "pool_scan"
    m = re.match(r"gemm_s3_kernel<(\d+), (\d+), ", k)
    if m:
        return f"gemm_s3_{m.group(1)}x{m.group(2)}x32"
    if k.startswith("gemm_s3p_kernel"):                      # persistent form of the same tile: same bench.py class
        return "gemm_s3_128x256x32"
    m = re.match(r"gemm_h2p?_kernel<(\d+), (\d+), ", k)          # gemm_h2p_kernel (A as f16x2 lines, LDS-DMA): the same tile, the same bench.py class
    if m:
        return f"gemm_h2_{m.group(1)}x{m.group(2)}x32"
    if k.startswith(("attn_colsplit_kernel", "attn_fused_kernel", "attn_h2_kernel", "attn_h2ks_kernel")):
        return "attn_fused"
    if k.startswith("ln4_kernel"):
        return "layernorm"
    if k.startswith("embed_ln4_groups_kernel"):
        return "embed_layernorm"
    if k.startswith(("pool_scan_ks_kernel", "pool_scan_dma_kernel", "pool_scan_ring_kernel")):
        return "pool_scan"
    return {"ln_kernel": "layernorm", "embed_ln_groups_kernel": "embed_layernorm", "causal_softmax_kernel": "causal_softmax",
            "lnf_partial_kernel<8>": "lnf_partial", "lnf_partial_kernel<16>": "lnf_partial", "lnf_partial_kernel<32>": "lnf_partial", "meanpool_reduce_kernel": "meanpool_reduce", "gemm_skinny_kernel": "gemm_skinny",
            "gemm_skinny_epilogue_kernel": "gemm_skinny_epilogue", "decode_attn_kernel": "decode_attention",
            "normalize_rows_kernel": "normalize_rows", "topk_chunk_kernel<float>": "topk",
            "merge_topk_kernel": "merge_topk", "jaccard_lds_kernel": "jaccard"}.get(k, k)


by_class = collections.defaultdict(lambda: collections.defaultdict(list))
for k, cs in agg.items():
    if k.startswith(("at::", "__amd")):
        continue
    for c, v in cs.items():
        by_class[bench_class(k)][c] += v
traffic = {}
for k, cs in by_class.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fe, wr = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        traffic[k] = {"hbm_bytes_per_launch": (2 * fe + wr) * 1024, "fetch_kib_raw": fe, "write_kib": wr}
        if "TCC_HIT_sum" in cs:
            h, m_ = sum(cs["TCC_HIT_sum"]), sum(cs["TCC_MISS_sum"])
            traffic[k]["l2_hit_rate"] = h / max(h + m_, 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            traffic[k]["mfma_pipe_util"] = (sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0) / (sum(cs["GRBM_GUI_ACTIVE"]) / 8.0)
# what tools/profile_bench.sh ran: bench.py only attaches these figures to a line of the same workload
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "a") as f:
    f.write("# the same trace aggregated per bench.py kernel class (template variants of one tile summed)\n")
    f.write("class,calls,total_ns,avg_ns\n")
    for c, (n, tot) in sorted(_class_rows().items(), key=lambda kv: -kv[1][1]):
        if not c.startswith(("at::", "__amd")):
            f.write(f"\"{c}\",{n},{tot:.0f},{tot / max(n, 1):.0f}\n")
# the dominant class, recomputable by a reader: algorithmic flop per launch (bench.py's own HIP-event pass over the same steps)
# over the rocprofv3 average duration of that class in THIS trace
try:
    line = json.loads([ln for ln in open(out + ".line.json") if ln.startswith("{")][-1])
    roof = line["roofline"]
    cls = _class_rows()
    n, tot = cls[roof["kernel"]]
    avg_ns = tot / max(n, 1)
    tf = roof["flop_per_launch"] / avg_ns / 1e3
    with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "a") as f:
        f.write(f"# dominant class {roof['kernel']}: flop_per_launch {roof['flop_per_launch']:.6g} (bench.py, same steps) / rocprofv3 avg "
                f"{avg_ns:.0f} ns = {tf:.1f} TFLOP/s = {tf / roof['peak']:.3f} of the {roof['peak']} TFLOP/s peak "
                f"(bench.py HIP events on the launch stream: {roof['avg_launch_us']} us, {roof['achieved']} TFLOP/s, frac {roof['frac']})\n")
except Exception as e:                                                        # noqa: BLE001
    print("no bench line beside the trace:", e)
sys.path.insert(0, repo)
from bench import source_sha                                                   # noqa: E402
shape = os.environ.get("R4D_PROFILE_SHAPE", "UCI_13")
gemm = os.environ.get("R4D_PROFILE_GEMM", "f16x2")
traffic["_workload"] = {"shape": shape, "batches_per_step": 8, "n_gpus": 1, "pool_rows_per_gpu": 100000, "gemm": gemm,
                        "source_sha": source_sha(),
                        "command": f"bench.py --shape {shape} --gemm {gemm} --steps 32 --warmup 16 --random-pool --headline-only"}
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv")).read())
print(json.dumps(traffic, indent=1)[:3000])
