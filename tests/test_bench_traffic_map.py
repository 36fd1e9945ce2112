"""bench.py attaches PMC traffic (profiles/rNN_kernel_hbm_traffic.json) to a roofline entry through the entry's own profiling kind
(bench.TRAFFIC_KEYS) — never through a name prefix with a catch-all (round 5: every HBM-bound entry carried the code search's 15 MB)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _entries():
    return [{"kernel": f"kind {k}", "prof_kind": k, "traffic": None} for k in range(0, 11)]


def test_no_two_entries_share_a_traffic_value_and_uncovered_kernels_get_null():
    import bench
    traffic = {name: {"hbm_bytes_per_launch": 1000.0 + 17 * i} for i, name in enumerate(
        ["assign", "conv3x3", "attn_fwd", "attn_bwd_dkdv", "attn_bwd_dq", "gemm", "res_ln_fwd", "res_ln_bwd", "adamw_ema"])}
    shapes = {"gemm nt qkv": 1.8, "assign N32768": 2.7, "res_ln fwd rows": 1.0, "res_ln bwd rows": 1.03, "attention fwd B128": 1.0,
              "attention bwd B128": 1.5, "conv3x3 fwd B64": 3.0}
    es = _entries()
    bench.attach_traffic(es, traffic, shapes, "profiles/x.json")
    by_kind = {e["prof_kind"]: e for e in es}
    vals = [e["traffic"] for e in es if e["traffic"] is not None]
    assert len(vals) == len(set(vals)) == 8, vals                      # kinds 0-7 covered, each with its own bytes
    assert by_kind[3]["traffic"] == traffic["attn_bwd_dkdv"]["hbm_bytes_per_launch"] + traffic["attn_bwd_dq"]["hbm_bytes_per_launch"]
    for k in (8, 9, 10):                                                  # GroupNorm, quantizer element-wise, conv3x3_from3: no row -> null
        assert by_kind[k]["traffic"] is None and "traffic_over_algorithmic_by_shape" not in by_kind[k]
    assert by_kind[5]["traffic_over_algorithmic_by_shape"] == {"res_ln fwd rows": 1.0}
    assert by_kind[6]["traffic_over_algorithmic_by_shape"] == {"res_ln bwd rows": 1.03}
    assert by_kind[0]["traffic_over_algorithmic_by_shape"] == {"assign N32768": 2.7}
    assert by_kind[1]["traffic_over_algorithmic_by_shape"] == {"conv3x3 fwd B64": 3.0}      # not conv3x3_from3's


def test_the_committed_profile_loads_and_maps():
    import bench
    traffic, shapes, src = bench.load_traffic_profiles()
    assert src and os.path.exists(os.path.join(ROOT, src)) and "gemm" in traffic
    es = _entries()
    bench.attach_traffic(es, traffic, shapes, src)
    vals = [e["traffic"] for e in es if e["traffic"] is not None]
    assert len(vals) == len(set(vals)) >= 5
    ln_fwd = [e for e in es if e["prof_kind"] == 5][0]
    if ln_fwd["traffic"] is not None:                                     # 605 MB algorithmic at 65 664 x 768: never the code search's 15 MB
        assert ln_fwd["traffic"] > 1e8
