"""bench.py's helper legs that run without a GPU (benchlib.py at the repo root): the committed rocprofv3 counter figures the roofline block
quotes must be those of the SAME configuration, strategy and preset -- never a neighbour's."""
import json
import os

import benchlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_traffic_matches_configuration_strategy_and_preset():
    t = benchlib.committed_traffic("c5_human_twoset", False, "ava-pb")
    assert t is not None and t["file"].startswith("profiles/r05_hbm_traffic")
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
