import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def unhex(v):
    """Inverse of the fixture writer: hex strings -> ints, nested lists -> tuples, None stays."""
    if v is None:
        return None
    if isinstance(v, str):
        return int(v, 16)
    return tuple(unhex(x) for x in v)


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def lcs(rows):
    return [[(col, int(c, 16)) for col, c in lc] for lc in rows]


@pytest.fixture(scope="session")
def gpu_ctx():
    import zinc_amd
    ctx = zinc_amd.Context(0)  # raises if the HIP library or the GPU is missing: no fallback
    yield ctx
    ctx.close()


class FixtureCircuit:
    """A prove_tiny fixture as something with bellman's `synthesize(cs)`."""

    def __init__(self, fx):
        self.fx = fx
        self.n_in, self.n_aux = fx["n_inputs"], fx["n_aux"]
        self.rows = (lcs(fx["rows_a"]), lcs(fx["rows_b"]), lcs(fx["rows_c"]))
        self.z_in = [int(x, 16) for x in fx["z_in"]]
        self.z_aux = [int(x, 16) for x in fx["z_aux"]]

    def synthesize(self, cs):
        for v in self.z_in[1:]:
            cs.alloc_input(v)
        for v in self.z_aux:
            cs.alloc(v)
        for a, b, c in zip(*self.rows):
            cs.enforce(a, b, c)


class FixtureKey:
    def __init__(self, fx):
        k = fx["key"]
        for name in ("alpha_g1", "beta_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2"):
            setattr(self, name, unhex(k[name]))
        for name in ("ic", "h", "l", "a", "b_g1", "b_g2"):
            setattr(self, name, [unhex(p) for p in k[name]])


def fixture_proof(fx):
    return (unhex(fx["proof"]["a"]), unhex(fx["proof"]["b"]), unhex(fx["proof"]["c"]))
