import os
import sys

import pytest

# torch bundles its own ROCm runtime: it has to be loaded before libtksmseq.so (as in bench.py), otherwise torch finds
# no device in a process that already loaded the system libamdhip64 through our library
import torch  # noqa: E402,F401

torch.cuda.is_available()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "tksm_amd", "models", "badread")
ERR_MODEL = os.path.join(MODELS, "nanopore2020.error.gz")
QS_MODEL = os.path.join(MODELS, "nanopore2020.qscore.gz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def oracle_models(po):
    return {"em": po.ErrorModel(ERR_MODEL), "qm": po.QScoreModel(QS_MODEL)}


@pytest.fixture(scope="session")
def seqr():
    """A context on cuda:0 with the nanopore2020 models and default identity (GPU tests only)."""
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    yield s
    s.close()
