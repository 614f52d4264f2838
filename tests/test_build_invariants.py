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
    """The LDS-DMA refill of the decode walk (decode.hip, WalkT::refill) and the v_writelane that collects
    the output set M0 from inline assembly and do not restore it, and the refill sets EXEC to -1 behind
    the load: valid as long as the compiler itself never uses M0 in those kernels and nothing spills.  If a toolchain change breaks that,
    this test says so before a GPU does."""
    out = tmp_path / "decode.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-I" + os.path.join(ROOT, "include"), "-o", str(out), os.path.join(ROOT, "fqcomp28_amd", "csrc", "decode.hip")],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    lines = [ln.strip() for ln in text.splitlines() if ln.strip() and not ln.lstrip().startswith(";")]
    m0 = [ln for ln in lines if re.search(r"\bm0\b", ln)]
    assert m0, "the refill is gone?"
    # the two users of M0 are the walk's own assembly blocks, and each sets M0 itself right in front of its use:
    # the refill (s_mov m0 .. global_load_lds_dword) and the output dword's v_writelane (lane select in M0)
    foreign = [ln for ln in m0 if not re.fullmatch(r"s_mov_b32 m0, s\d+", ln) and not re.fullmatch(r"v_writelane_b32 v\d+, s\d+, m0", ln)]
    assert not foreign, foreign[:5]
    for i, ln in enumerate(lines):
        if ln.startswith("global_load_lds_dword"):
            assert any(re.fullmatch(r"s_mov_b32 m0, s\d+", p) for p in lines[i - 3:i]), lines[i - 4:i + 1]
        if ln.startswith("v_writelane_b32") and ln.endswith("m0"):
            assert re.fullmatch(r"s_mov_b32 m0, s\d+", lines[i - 1]), lines[i - 2:i + 1]
    assert "global_load_lds_dword" in text
    assert "scratch_" not in text


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_decode_walk_never_sign_extends_a_stream_word(tmp_path):
    """Round 2's "wrong bits with the bit window in scalar registers" was no hardware hazard: the
    builtin behind v_readfirstlane returns int, so `window << 32 | readfirstlane(word)` sign-extended --
    the ISA had `s_ashr_i32 hi, lo, 31` in front of the `s_or_b64` that slides the window, i.e. 32 ones
    in the upper half whenever bit 31 of the incoming stream word was set.  decode.hip therefore goes
    through fq_uniform() (unsigned in, unsigned out) only, and its ISA holds no arithmetic shift by 31
    that feeds a 64-bit OR."""
    src = open(os.path.join(ROOT, "fqcomp28_amd", "csrc", "decode.hip")).read()
    code = re.sub(r"//.*", "", src)
    assert "__builtin_amdgcn_readfirstlane" not in code, "use fq_uniform(): the builtin's result is a signed int"
    out = tmp_path / "decode.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-I" + os.path.join(ROOT, "include"), "-o", str(out), os.path.join(ROOT, "fqcomp28_amd", "csrc", "decode.hip")],
                   check=True, capture_output=True, timeout=600)
    lines = [ln.strip() for ln in out.read_text().splitlines()]
    assert any(ln.startswith("s_lshl_b64") or ln.startswith("s_or_b64") for ln in lines), "the scalar window is gone?"
    for i, ln in enumerate(lines):
        m = re.match(r"s_ashr_i32 (s\d+), s\d+, 31$", ln)
        if not m:
            continue
        hi = int(m.group(1)[1:])
        # the sign word must not become the upper half of a 64-bit OR operand within the next few instructions
        for nxt in lines[i + 1: i + 6]:
            assert not re.match(r"s_or_b64 .*s\[%d:%d\]" % (hi - 1, hi), nxt), (ln, nxt)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_partition_ranking_loops_are_wave_uniform_straight_line_blocks(tmp_path):
    """K3 (k_tile_partition) takes the rank of a symbol from ONE lane-ordered LDS atomic: right only if
    iteration k of every lane precedes iteration k + 1 of any lane.  Round 2 found the failure mode as a
    parity error: with a per-lane trip count hipcc's unrolling let low lanes run a group of iterations
    ahead.  The property on the ISA: every loop that holds the ranking atomics is ONE basic block that
    branches back on a SCALAR condition and never touches EXEC -- all lanes walk it together, the atomics
    issue in program order -- with the G atomics of the hand-pipelined group in it (8 plain, 4 combining)."""
    out = tmp_path / "encode.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-I" + os.path.join(ROOT, "include"), "-o", str(out), os.path.join(ROOT, "fqcomp28_amd", "csrc", "encode.hip")],
                   check=True, capture_output=True, timeout=900)
    lines = out.read_text().splitlines()
    found = {}
    for model in ("9QualModel", "8SeqModel"):
        start = [i for i, ln in enumerate(lines) if ln.startswith("_ZN12_GLOBAL__N_116k_tile_partitionI" + model)][0]
        end = next(i for i in range(start, len(lines)) if ".amdhsa_next_free_vgpr" in lines[i])
        blocks, cur = [], []
        for ln in lines[start:end]:
            if re.match(r"^\.LBB\d+_\d+:", ln) or ln.startswith("; %bb."):
                blocks.append(cur)
                cur = [ln]
            else:
                cur.append(ln)
        blocks.append(cur)
        counts = []
        for b in blocks:
            n = sum("ds_add_rtn_u32" in ln for ln in b)
            if not n:
                continue
            label = b[0].split(":")[0]
            code = [ln.strip() for ln in b[1:] if ln.strip() and not ln.strip().startswith(";")]
            back = [i for i, ln in enumerate(code) if ln in ("s_cbranch_scc0 " + label, "s_cbranch_scc1 " + label)]
            assert back, (model, label, "the ranking loop does not branch back on a scalar condition")
            code = code[: back[0] + 1]  # the loop: from its label to its back edge
            assert sum("ds_add_rtn_u32" in ln for ln in code) == n
            assert not [ln for ln in code if re.search(r"\bexec\b", ln)], (model, label, "EXEC is touched inside a ranking loop")
            assert [ln for ln in code if ln.startswith("s_cbranch") or ln.startswith("s_branch")] == [code[-1]], (model, label)
            counts.append(n)
        found[model] = sorted(counts)
    assert found == {"9QualModel": [4, 8], "8SeqModel": [4, 8]}, found


def test_product_library_has_no_experiment_switches():
    """The timing-experiment switches (FQGPU_DEBUG_*: kernels skipped, wrong output by design) live in
    the -DFQGPU_EXPERIMENTS build of tools/traffic_experiment.py only."""
    lib = os.path.join(ROOT, "fqcomp28_amd", "libfqgpu.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    data = open(lib, "rb").read()
    assert b"FQGPU_DEBUG" not in data
    assert b"FQ_EXP_" not in data and b"FQ_FARM_TRACE" not in data


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_seq_setfunc_table_sits_at_lds_address_zero(tmp_path):
    """k_seq_setfunc's gathers form their LDS address as row + state with NOTHING added for the table's base
    (sets_gather2 in enc_chains_seq.h: one SDWA add, one ds_read): right only while the context's table is the first
    thing in the kernel's LDS, i.e. while the kernel declares no static __shared__ at all (dynamic LDS starts at 0)."""
    out = tmp_path / "encode.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-I" + os.path.join(ROOT, "include"), "-o", str(out), os.path.join(ROOT, "fqcomp28_amd", "csrc", "encode.hip")],
                   check=True, capture_output=True, timeout=900)
    text = out.read_text()
    kernels = re.findall(r"\.amdhsa_kernel (\S*k_seq_setfunc\S*)\n\s*\.amdhsa_group_segment_fixed_size (\d+)", text)
    assert len(kernels) == 2, kernels
    assert all(int(size) == 0 for _, size in kernels), kernels
    body = text[text.index("k_seq_setfuncILj32ELb1E"):]
    body = body[:body.index("s_endpgm")]
    n_sdwa = len(re.findall(r"v_add_u32_sdwa v\d+, s\d+, v\d+ .*src0_sel:WORD_[01]", body))
    assert n_sdwa >= 16, n_sdwa
