"""Build-time assumptions of hand-written assembly, checked on the compiler's output (CPU only:
hipcc cross-compiles for gfx950 without a GPU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_decode_walk_owns_m0_and_has_no_scratch(tmp_path):
    """The LDS-DMA refill of the decode walk (decode.hip, WalkT::refill) sets M0 from inline assembly
    and does not restore it, and it sets EXEC to -1 behind the load: valid as long as the compiler
    itself never uses M0 in those kernels and nothing spills.  If a toolchain change breaks that,
    this test says so before a GPU does."""
    out = tmp_path / "decode.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-I" + os.path.join(ROOT, "include"), "-o", str(out), os.path.join(ROOT, "fqcomp28_amd", "csrc", "decode.hip")],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    m0 = [ln.strip() for ln in text.splitlines() if re.search(r"\bm0\b", ln) and not ln.lstrip().startswith(";")]
    assert m0, "the refill is gone?"
    foreign = [ln for ln in m0 if not re.fullmatch(r"s_mov_b32 m0, s\d+", ln)]
    assert not foreign, foreign[:5]
    assert "global_load_lds_dword" in text
    assert "scratch_" not in text


def test_product_library_has_no_experiment_switches():
    """The timing-experiment switches (FQGPU_DEBUG_*: kernels skipped, wrong output by design) live in
    the -DFQGPU_EXPERIMENTS build of tools/traffic_experiment.py only."""
    lib = os.path.join(ROOT, "fqcomp28_amd", "libfqgpu.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    data = open(lib, "rb").read()
    assert b"FQGPU_DEBUG" not in data
    assert b"FQ_EXP_" not in data and b"FQ_FARM_TRACE" not in data
