"""Boundary hygiene (CPU): the C-ABI library loads and exports every symbol include/emx.h
declares; the product never touches oracle/ and has no CPU fallback."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "emx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(emx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from emcee_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) > 45
    for s in syms:
        assert hasattr(lib, s), "libemx.so does not export %s" % s
    # and the ctypes table covers the header (no untyped entry points)
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    assert lib.emx_version().startswith(b"emx")


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    """bench.py and its pieces (tools/benchkit/): the oracle / the materialised reference are the CPU baseline's, nothing else's"""
    files = [os.path.join(ROOT, "bench.py")] + [os.path.join(ROOT, "tools", "benchkit", f) for f in os.listdir(os.path.join(ROOT, "tools", "benchkit"))
                                                if f.endswith(".py")]
    for path in files:
        txt = open(path).read()
        uses = bool(re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M)) or "ref_shim" in txt
        assert uses == path.endswith(os.path.join("benchkit", "cpu.py")), path


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "emcee_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "/root/reference" in txt.replace(
                        "/root/reference/src/emcee", ""):
                    bad.append(os.path.join(dirpath, f))
                # ... nor the reference itself: neither the materialised copy (oracle/_ref, tools/make_ref.sh) nor any `emcee` import
                if "_ref" in txt and re.search(r"oracle[/.]_ref|ref_shim", txt):
                    bad.append(os.path.join(dirpath, f))
                if f.endswith(".py") and re.search(r"^\s*(from|import)\s+emcee\b(?!_amd)", txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_reference_copy_is_never_tracked():
    """oracle/_ref (tools/make_ref.sh: the reference's package, for bench.py's cpu_baseline leg on the GPU box) must stay out of
    history: git-ignored, and not listed in .gpurunignore (it has to travel)."""
    ign = open(os.path.join(ROOT, ".gitignore")).read().split()
    assert "oracle/_ref/" in ign
    gi = os.path.join(ROOT, ".gpurunignore")
    if os.path.exists(gi):
        assert "oracle/_ref" not in open(gi).read()
    import subprocess
    r = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True)
    if r.returncode == 0:
        assert r.stdout.strip() == "", "oracle/_ref is tracked: %s" % r.stdout


def test_ref_shim_finds_a_materialised_reference(tmp_path, monkeypatch):
    from oracle import ref_shim
    monkeypatch.setattr(ref_shim, "REF_SRC", str(tmp_path / "nowhere"))
    monkeypatch.setattr(ref_shim, "LOCAL_SRC", str(tmp_path / "nowhere2"))
    assert not ref_shim.available()
    (tmp_path / "loc" / "emcee").mkdir(parents=True)
    (tmp_path / "loc" / "emcee" / "ensemble.py").write_text("")
    monkeypatch.setattr(ref_shim, "LOCAL_SRC", str(tmp_path / "loc"))
    assert ref_shim.available() and ref_shim.source() == str(tmp_path / "loc")


def test_no_cpu_fallback_without_a_gpu():
    from emcee_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    import numpy as np
    import emcee_amd
    from emcee_amd.device import DeviceEnsemble, EmxError
    with pytest.raises(EmxError):
        DeviceEnsemble(32, 2)
    s = emcee_amd.EnsembleSampler(32, 2, emcee_amd.targets.IsoGaussian())
    with pytest.raises(EmxError):
        s.run_mcmc(np.random.RandomState(0).randn(32, 2), 2)
    s = emcee_amd.EnsembleSampler(32, 2, lambda p: -0.5 * np.sum(p ** 2))
    with pytest.raises(EmxError):
        s.run_mcmc(np.random.RandomState(0).randn(32, 2), 2)
