"""HBM traffic per launch of the instrumented kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).
   python tools/pmc_traffic.py <dir_fetch> <dir_write> > profiles/<name>.json
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts 128-byte requests of wide coalesced reads at 64 B
(MI355X_MICROARCH.md, HBM section): the read figure is doubled; WRITE_SIZE is taken as is (uncalibrated, see the guide)."""
import collections, csv, glob, json, sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row.get("Kernel_Name") or row.get("kernel_name")].append(float(row["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
groups = {"conv3x3_kernel": "conv3x3", "attn_fwd_kernel": "attn_fwd", "attn_bwd_dkdv_kernel": "attn_bwd_dkdv", "attn_bwd_dq_kernel": "attn_bwd_dq",
          "assign_kernel": "assign", "gemm_": "gemm", "gelu_fwd": "gelu_fwd", "gelu_bwd": "gelu_bwd", "conv3x3_wgrad": "conv3x3_wgrad", "res_ln_bwd_kernel|res_ln_bwd_cols_kernel": "res_ln_bwd", "res_ln_fwd_kernel": "res_ln_fwd", "adamw_ema_kernel": "adamw_ema",
          # round 6: every HBM-bound roofline entry of bench.py has its own row (bench.py TRAFFIC_KEYS)
          "conv3x3_from3": "conv3x3_from3", "gn_reduce|gn_apply|gn_finalize": "gn", "vq_finish_kernel|vq_backward_kernel|vq_codebook_grad_kernel": "vq_elem"}
out = {}
for key, name in groups.items():
    alts = key.split("|")
    fr = [v for k, vs in fetch.items() if any(a in k for a in alts) for v in vs]
    wr = [v for k, vs in write.items() if any(a in k for a in alts) for v in vs]
    if not fr and not wr:
        continue
    out[name] = {"launches_fetch_pass": len(fr), "launches_write_pass": len(wr),
                 "read_bytes_per_launch": 2.0 * 1024.0 * sum(fr) / max(1, len(fr)),      # KB -> B, x2 gfx950 correction
                 "write_bytes_per_launch": 1024.0 * sum(wr) / max(1, len(wr)),
                 "raw_fetch_size_kb_avg": sum(fr) / max(1, len(fr)), "raw_write_size_kb_avg": sum(wr) / max(1, len(wr))}
    out[name]["hbm_bytes_per_launch"] = out[name]["read_bytes_per_launch"] + out[name]["write_bytes_per_launch"]
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-mfu --graph off (counter collection does not survive hipGraph replays: the FETCH pass hung, the WRITE pass crashed)",
                  "corrections": "FETCH_SIZE KB x 1024 x 2 (gfx950 wide-read under-count); WRITE_SIZE KB x 1024", "kernels": out}, indent=1))
