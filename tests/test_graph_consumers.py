"""Wire format (a10) and mapping score options (a11) against fixtures produced by the reference's own
Mapping / Graph classes (oracle/gen_golden_graph.py).  No GPU; HDF5 needs h5py (see test_mapping.py)."""
import json
import os
import subprocess

import pytest

from test_mapping import _interpreter

HERE = os.path.dirname(os.path.abspath(__file__))


def test_writer_bytes_and_score_options_vs_reference():
    py = _interpreter()
    if py is None:
        pytest.skip("no interpreter with h5py in this image")
    r = subprocess.run([py, os.path.join(HERE, "_graph_case.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    assert res["digest_equals_fixture"] and res["digest_equals_reference_file"], res
    assert res["reference_reader_verdict"], res
    assert res["score_calls_checked"] >= 18 and res["score_calls_differ"] == [], res
    assert res["score_errors_differ"] == [], res
    assert res["by_cluster_calls_checked"] >= 4 and res["by_cluster_calls_differ"] == [], res
    assert res["columnar_graph_same_scores"] and res["columnar_graph_expands_to_same_digest"], res
