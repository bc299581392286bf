"""tools/erm_cli.c: a plain-C host over include/ertirt.h (no Python, no torch) -- the same calls the Julia shim's ccall makes.
CPU: it compiles against the header, links the library and answers --version.  GPU: it runs every model end to end."""
import re
import subprocess

import pytest

import parity_util as pu


def _cli():
    return pu.ge.build_cli()


def test_c_driver_builds_and_links_the_abi():
    out = subprocess.run([_cli(), "--version"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "gfx950" in out.stdout


def test_c_driver_generates_data_without_a_gpu():
    out = subprocess.run([_cli(), "--nsubj", "3000", "--nitem", "12", "--dry-run"], capture_output=True, text=True, timeout=60)
    m = re.search(r"mean\(Y\)=([0-9.]+) mean\(logT\)=([0-9.]+)", out.stdout)
    assert out.returncode == 0 and m and 0.3 < float(m.group(1)) < 0.7 and 3.0 < float(m.group(2)) < 5.0, out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["mlirt", "rtirt", "latentqr", "crossqr", "null", "cross", "latent"])
def test_c_driver_runs_each_model(model):
    out = subprocess.run([_cli(), "--model", model, "--nsubj", "3000", "--nitem", "12", "--niter", "300", "--qrt", "0.5"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"cor\(theta\)=([0-9.]+) rmse\(a\)=([0-9.]+) rmse\(b\)=([0-9.]+)", out.stdout)
    assert m, out.stdout
    assert float(m.group(1)) > 0.8 and float(m.group(2)) < 0.25 and float(m.group(3)) < 0.25
    assert "post rows    150" in out.stdout
