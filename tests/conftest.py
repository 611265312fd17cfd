import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np

    g = ROOT / "tests" / "golden"
    return {"npz": np.load(g / "make_data_cases.npz"), "kat": json.loads((g / "readme_kat.json").read_text())}


@pytest.fixture(scope="session")
def notebook():
    """The reference demo notebook's seeded frame + its printed outputs (tests/golden/notebook_kat.json, notebook_frame.npz)."""
    import json
    import numpy as np

    from refdata import notebook_make_data

    g = ROOT / "tests" / "golden"
    kat = json.loads((g / "notebook_kat.json").read_text())
    fr = kat["frame"]
    d = notebook_make_data(n_samples=fr["n_samples"], n_features=fr["n_features"], n_groups=fr["n_groups"])
    z = np.load(g / fr["file"])
    for k in ("x", "y", "group", "sample_weights"):      # the committed frame IS what this numpy's Generator regenerates
        assert np.array_equal(d[k], z[k]), k
    return {"kat": kat, "d": d}
