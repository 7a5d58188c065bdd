"""bench.py's helper legs that run without a GPU (benchlib.py at the repo root): the committed rocprofv3 counter figures the roofline block
quotes must be those of the SAME configuration, strategy and preset -- never a neighbour's."""
import json
import os

import benchlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_traffic_matches_configuration_strategy_and_preset():
    t = benchlib.committed_traffic("c5_human_twoset", False, "ava-pb")
    assert t is not None and t["file"].startswith(("profiles/r05_hbm_traffic", "profiles/r06_hbm_traffic"))
    d = json.load(open(os.path.join(ROOT, t["file"])))
    assert d["config"] == "c5_human_twoset" and not d.get("inverse", False) and "preset ava-pb" in d["workload"]
    # no counter collection of the ava-ont run / of the inverse strategy at full size has been committed: nothing may be quoted for them
    assert benchlib.committed_traffic("c5_human_twoset", False, "ava-ont") is None
    assert benchlib.committed_traffic("c5_human_twoset", True, "ava-pb") is None
    c4 = benchlib.committed_traffic("c4_dmel_twoset", False, "ava-ont")
    assert c4 is not None and "c4_dmel_twoset" in c4["file"]


def test_committed_kernel_traffic_sums_template_instantiations():
    per_launch, detail = benchlib.committed_kernel_traffic("c5_human_twoset", False, "k_rs_scatter", "ava-pb")
    assert per_launch and detail["launches_in_pass"] > 10           # every k_rs_scatter<...> instantiation of the pass
    assert abs(per_launch - (2.0 * detail["fetch_raw_bytes_per_launch"] + detail["write_raw_bytes_per_launch"])) < 1.0
    assert benchlib.committed_kernel_traffic("c5_human_twoset", False, "k_no_such_kernel", "ava-pb") == (None, None)


def test_roofline_blocks_shape_and_bounds():
    """The bench line's roofline bookkeeping: `roofline` is the SURVEY 8(d) whole-path block with the contract's keys;
    the single kernels are ranked by time per step and carry bound = hbm | valu (k_sketch_direct and k_chain_lpg are VALU-bound); nothing
    inside a block comes from a committed file -- those figures (counter traffic, VALU issue rates) are returned apart, for the one
    `from_committed_profiles` key."""
    import argparse
    import numpy as np
    a = argparse.Namespace(config="c5_human_twoset", inverse=False)
    K, Qn, Tn = 2, 100000, 2000000
    q_lens = np.full(Qn, 15000, dtype=np.int64); t_lens = np.full(Tn, 15000, dtype=np.int64)
    st = {"n_minimizers": 7487000000, "n_keys": 1, "mid_occ": 100}
    acc_tb = {"sketch": 2 * 216.0, "index_sort": 2 * 198.0, "k_sketch": 2 * 147.0, "rs_scatter": 2 * 110.0}
    acc_tm = {"k_lookup": 2 * 34.0, "lookup": 2 * 42.0, "chain_lpg": 2 * 96.0, "expand": 2 * 105.0, "anchor_sort": 2 * 71.0, "rs_scatter": 2 * 45.0}
    acc_cn = {"lookup_launches": 6, "query_minimizers": 2 * 3 * 374000000, "lpg_launches": 6, "lpg_anchors": 2 * 3400000000, "sketch_wave_launches": 6, "batches": 6,
              "anchors": 2 * 10940000000, "anchors_kept": 2 * 3560000000, "index_parts": 3, "rs_scatter_launches": 72, "rs_scatter_bytes": 2 * 775e9, "rs_scatter_items": 2 * 4e10}
    roofline, kernels, fams, committed = benchlib.roofline_blocks(a, 1, 1, K, 845.0, acc_tb, acc_tm, acc_cn, {}, {}, {}, True, st, Qn, Tn, q_lens, t_lens)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roofline
    assert roofline["bound"] == "hbm" and roofline["unit"] == "GB/s" and roofline["peak"] == benchlib.HBM_PEAK_GBPS and roofline["traffic"] is None
    assert 0.05 < roofline["frac"] < 0.15 and abs(roofline["frac"] - roofline["achieved"] / roofline["peak"]) < 1e-12     # ~0.085 at these (round 5) figures
    names = [k["kernel"] for k in kernels]
    assert {"k_sketch_wave", "k_expand_q", "k_chain_lpg", "k_lookup", "k_rs_scatter"} <= set(names)
    assert [k["ms_per_step"] for k in kernels] == sorted((k["ms_per_step"] for k in kernels), reverse=True)
    by = {k["kernel"]: k for k in kernels}
    assert by["k_sketch_wave"]["bound"] == "valu" and by["k_chain_lpg"]["bound"] == "valu" and by["k_lookup"]["bound"] == "hbm"
    assert all(k["traffic"] is None and "traffic_detail" not in k for k in kernels + fams)
    assert all(k["frac"] < 1.0 for k in kernels + fams)
    assert set(committed) >= {"kernel_traffic", "valu_issue", "whole_path_traffic"}
    assert committed["whole_path_traffic"]["file"].startswith("profiles/")
    v = committed["valu_issue"].get("k_chain_lpg")
    assert v is None or 0.3 < v["issue_frac"] < 1.2
