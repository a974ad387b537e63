"""nabo_amd.Mapping against everything the reference's Mapping wrote for the same inputs
(golden fixtures): stored neighbour lists, SNN graph incl. repair edges, wire format, mapping
score.  HDF5 needs h5py, which the default interpreter of this image lacks; the case script then
runs under /opt/conda/bin/python3.9 (same image on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASE = os.path.join(HERE, "_mapping_case.py")


def _interpreter():
    try:
        import h5py  # noqa: F401
        return sys.executable
    except ImportError:
        pass
    for cand in ("/opt/conda/bin/python3.9", "/opt/conda/bin/python"):
        if os.path.exists(cand):
            r = subprocess.run([cand, "-c", "import h5py, numpy"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode == 0:
                return cand
    return None


def _run(mode):
    py = _interpreter()
    if py is None:
        pytest.skip("no interpreter with h5py in this image")
    r = subprocess.run([py, CASE, mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_mapping_api_validation_rules():
    res = _run("validate")
    bad = {k: v for k, v in res.items() if v is not True}
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["small", "dup", "c1"] + ["mini_%d" % i for i in range(6)])
def test_mapping_end_to_end_vs_reference_outputs(mode):
    res = _run(mode)
    assert res["ref_cells_equal"]
    assert res["ref_idx_equal"] and res["ref_dist_equal"]
    assert res["ref_graph_nodes_equal"]
    assert res["ref_graph_edges_equal"], (res["ref_graph_missing"], res["ref_graph_extra"], res["log"])
    for k, v in res.items():
        if k.endswith("_graph_nodes_equal") or k.endswith("_graph_edges_equal") or k.endswith("_graph_raw_equal"):
            assert v is True, k
        if k.endswith("_score_maxerr") or k.endswith("_score_api_maxerr"):
            assert v < 1e-9, (k, v)
    assert res["stored_distances_same_graph"] and res["columnar_same_graph"] and res["dense_input_same_graph"]
    assert res["sharded_devices_same_graphs"]
    assert res["target_slices_same_graphs"]
    assert res["store_k_serves_larger_k_per_cell"] and res["store_k_serves_larger_k_columnar"] and res["no_store_k_raises"]
    assert res["target_metric_euclidean"]
    assert res["columnar_graph_same_scores"] and res["columnar_graph_expands_to_the_wire_format"]
    assert res.get("columnar_graph_same_null", True)
    if "null_obs_is_mapping_score" in res:
        assert res["null_obs_is_mapping_score"] and res["null_pvalues_in_range"]
