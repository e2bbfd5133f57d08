import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """build the native libraries once per session when they are missing"""
    from pseudocylindrical_convolution_amd import build
    here = os.path.join(ROOT, "pseudocylindrical_convolution_amd")
    if not (os.path.exists(os.path.join(here, "libpconv_hip.so")) and
            os.path.exists(os.path.join(here, "libpconv_coder.so"))):
        build.build()
    from oracle import pconv_cpu
    pconv_cpu.lib()


@pytest.fixture()
def oracle_backend():
    """run the operator layer on the CPU oracle for the duration of a test"""
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import pconv_cpu, coder_cpu
    backend.use(pconv_cpu, coder_cpu)
    pconv_cpu.set_detmath(True)
    yield pconv_cpu
    backend.reset()


@pytest.fixture()
def hip_backend():
    import torch
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    backend.reset()
    from pseudocylindrical_convolution_amd import PCONV
    yield PCONV
    backend.reset()
