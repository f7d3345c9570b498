"""Static check of the MFMA operand rule on the compiler's own assembly.

hipcc assumes an MFMA has read its A/B registers when it issues; on gfx950 an issued MFMA may still have to
fetch them, so a load that the compiler places behind it into one of those registers can land first.
The split-precision kernels are written so that this never happens (fragment reads before the first MFMA of a
phase, completion reads before registers are reloaded, operands kept allocated); tools/war_scan.py proves it
on the generated code.  This test cross-compiles the kernel sources to gfx950 assembly (no GPU needed) and
requires zero violating pairs, so that a later edit -- or a different compiler schedule -- cannot reintroduce
the hazard unnoticed.  The scanner follows every MFMA along the control-flow graph (both arms of conditional
branches, loops) up to a completion read or the end of the program: no fixed window.  The compiler the result
holds for is recorded in the assertion message.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
SCAN = os.path.join(ROOT, "tools", "war_scan.py")
LDS_SCAN = os.path.join(ROOT, "tools", "lds_branch_vmem_scan.py")


# (source, minimum number of kernel instantiations with MFMAs scanned)
SOURCES = [("kernels_split.hip", 100), ("kernels_mfma.hip", 40), ("kernels_gemm.hip", 4), ("kernels_h256.hip", 50),
           ("kernels_lowrank.hip", 30)]
# every kernel source of the library, for the second rule (no MFMAs needed)
ALL_SOURCES = [s for s, _ in SOURCES] + ["kernels_generic.hip", "kernels_head.hip", "kernels_densify.hip",
                                          "kernels_debug.hip"]


@pytest.fixture(scope="session")
def kernel_asm(tmp_path_factory):
    """gfx950 assembly of every kernel source, compiled ONCE per session and four at a time (hipcc cross-compiles; the
    two scanners then read the same files): name -> path."""
    from concurrent.futures import ThreadPoolExecutor
    out = tmp_path_factory.mktemp("asm")

    def one(name):
        src = os.path.join(ROOT, "kws_amd", "csrc", name)
        asm = out / (name + ".s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                        "-o", str(asm), src], check=True, capture_output=True, timeout=1500)
        return name, str(asm)

    with ThreadPoolExecutor(max_workers=4) as ex:
        return dict(ex.map(one, ALL_SOURCES))


def _hipcc_version():
    try:
        out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True, timeout=60).stdout
        return " | ".join(l.strip() for l in out.splitlines()[:2])
    except Exception as e:  # pragma: no cover
        return "unknown (%s)" % e


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("name,min_kernels", SOURCES, ids=[s[0] for s in SOURCES])
def test_no_load_lands_behind_an_mfma_that_reads_its_target(kernel_asm, name, min_kernels):
    asm = kernel_asm[name]
    r = subprocess.run([sys.executable, SCAN, str(asm)], capture_output=True, text=True, timeout=900)
    lines = r.stdout.strip().splitlines()
    ver = _hipcc_version()
    assert lines and lines[-1].startswith("total pairs:"), (ver, r.stdout[-2000:] + r.stderr[-2000:])
    # every kernel the object declares was scanned (the scanner itself compares against .amdhsa_kernel)
    census = [l for l in lines if l.startswith("kernels declared")]
    assert census and "NOT SCANNED" not in census[0], (ver, census)
    kernels = [l for l in lines if " mfma " in l and int(l.split(" mfma ")[1].split()[0]) > 0]
    assert len(kernels) >= min_kernels, (ver, len(kernels))
    bad = [l for l in lines if ("pairs" in l and " mfma " in l and int(l.split("pairs")[1].split()[0]) > 0)
           or "UNBOUNDED" in l]
    assert not bad, ver + "\n" + "\n".join(bad[:20])
    assert r.returncode == 0, (ver, lines[-3:])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("name", ALL_SOURCES)
def test_no_lds_write_pending_at_a_branch_with_memory_instructions_behind_it(kernel_asm, name):
    """Second rule, by construction since round 3 and in EVERY kernel of the library: inside a loop no LDS
    write may be pending -- no `s_waitcnt lgkmcnt(0)` or barrier behind it on some path -- when a conditional branch,
    exec-masked or wave-uniform, with a vector-memory instruction within 12 instructions behind it is reached.  Round 2
    only excluded the narrow signature of the one kernel family that had returned wrong results (the write's data
    registers overwritten by a VALU instruction behind the branch, everything within a 12-instruction window); the
    masked stores of the forward kernels, the F = 32 backward and the fallback scans were left alone.  The scanner
    walks the control-flow graph (loops = its strongly connected components).  The compiler this holds for is in the
    assertion message."""
    asm = kernel_asm[name]
    ver = _hipcc_version()
    for mode in ("--strict", "--narrow"):
        r2 = subprocess.run([sys.executable, LDS_SCAN, str(asm), mode], capture_output=True, text=True, timeout=900)
        assert r2.returncode == 0, (ver, r2.stderr[-2000:])
        last = r2.stdout.strip().splitlines()[-1]
        assert last == "sites in loops: 0", (ver, mode, r2.stdout[-2000:])


HAZARD_FAR = """
_Z3farv:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]
%s
\tds_read_b128 v[4:7], v20
\tv_add_f32 v30, v0, v0
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel _Z3farv
""" % "\n".join("\tv_add_f32 v%d, v%d, v%d" % (40 + k % 8, 50, 51) for k in range(120))

HAZARD_TAKEN_ARM = """
_Z5takenv:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]
\ts_cbranch_scc1 .LBB0_2
\tv_add_f32 v30, v0, v0
\ts_endpgm
.LBB0_2:
\tglobal_load_dwordx4 v[8:11], v[20:21], off
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel _Z5takenv
"""

CLEAN_AFTER_COMPLETION_READ = """
_Z5cleanv:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]
\tv_mfma_f32_16x16x32_bf16 v[12:15], v[4:7], v[8:11], v[12:15]
\tv_cmp_eq_f32 vcc, v12, v12
\tds_read_b128 v[4:7], v20
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel _Z5cleanv
"""

NOT_SCANNED = CLEAN_AFTER_COMPLETION_READ + "\t.amdhsa_kernel _Z7missingv\n"

# the C input is fetched like A and B: a reload of it behind the MFMA is a hazard too
HAZARD_C_OPERAND = """
_Z4cregv:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[12:15]
\tds_read_b128 v[12:15], v20
\tv_add_f32 v30, v0, v0
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel _Z4cregv
"""

# a result register that something else has overwritten no longer proves anything: MFMA 1's result v[0:3] is consumed
# only as the C input of MFMA 2, then reloaded; the read of v0 that follows is a read of the LOAD, not of the pipe
HAZARD_STALE_RESULT = """
_Z5stalev:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], 0
\tv_mfma_f32_16x16x32_bf16 v[16:19], v[4:7], v[8:11], v[0:3]
\tds_read_b128 v[0:3], v20
\tv_add_f32 v30, v0, v0
\tds_read_b128 v[4:7], v21
\tv_add_f32 v31, v16, v16
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel _Z5stalev
"""


@pytest.mark.parametrize("asm,bad", [(HAZARD_FAR, True), (HAZARD_TAKEN_ARM, True),
                                     (CLEAN_AFTER_COMPLETION_READ, False), (NOT_SCANNED, True),
                                     (HAZARD_C_OPERAND, True), (HAZARD_STALE_RESULT, True)],
                         ids=["beyond_any_window", "branch_target_arm", "completion_read", "kernel_not_scanned",
                              "c_operand", "stale_result_register"])
def test_scanner_on_synthetic_streams(tmp_path, asm, bad):
    """The scanner itself: a reload 120 instructions behind the MFMA and one on the taken arm of a conditional
    branch are both found (the round-1 scanner looked 40 instructions ahead on the fall-through path only); a
    reload behind a read of a YOUNGER MFMA's result is allowed; a declared kernel that was not scanned fails."""
    f = tmp_path / "k.s"
    f.write_text(asm)
    r = subprocess.run([sys.executable, SCAN, str(f)], capture_output=True, text=True, timeout=60)
    assert (r.returncode != 0) == bad, r.stdout


LDS_BRANCH_HAZARD = """
_Z3ldsv:
.LBB0_1:
\tds_write2st64_b64 v10, v[20:21], v[66:67] offset0:36 offset1:45
\ts_and_saveexec_b64 s[34:35], s[0:1]
\ts_cbranch_execz .LBB0_3
\tv_lshl_add_u64 v[66:67], v[156:157], 0, s[30:31]
\tglobal_store_dwordx4 v[66:67], v[62:65], off
.LBB0_3:
\ts_or_b64 exec, exec, s[34:35]
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
"""

LDS_BRANCH_CLEAN = LDS_BRANCH_HAZARD.replace("\ts_and_saveexec_b64", "\ts_waitcnt lgkmcnt(0)\n\ts_and_saveexec_b64")


@pytest.mark.parametrize("asm,sites", [(LDS_BRANCH_HAZARD, 1), (LDS_BRANCH_CLEAN, 0)], ids=["signature", "waited"])
def test_lds_write_branch_overwrite_scanner_on_synthetic_streams(tmp_path, asm, sites):
    f = tmp_path / "k.s"
    f.write_text(asm)
    # (--strict also counts the loop's own back branch: the write is still pending there and the store follows it)
    for mode, n in (("--narrow", sites), ("--strict", 2 * sites)):
        r = subprocess.run([sys.executable, LDS_SCAN, str(f), mode], capture_output=True, text=True, timeout=60)
        assert r.stdout.strip().splitlines()[-1] == "sites in loops: %d" % n, (mode, r.stdout)


# --strict: the write may be far in front of the branch and its data registers untouched; the branch may be wave-uniform
# and the memory instruction a load
LDS_STRICT_FAR = """
_Z3farv:
.LBB0_1:
\tds_write_b128 v10, v[20:23]
%s
\ts_cbranch_vccnz .LBB0_3
\tglobal_load_dwordx4 v[40:43], v[60:61], off
.LBB0_3:
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
""" % "\n".join("\tv_add_f32 v%d, v50, v51" % (30 + k % 4) for k in range(60))
LDS_STRICT_FAR_WAITED = LDS_STRICT_FAR.replace("\ts_cbranch_vccnz", "\ts_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_cbranch_vccnz")
# the write reaches the branch only around the loop's back edge
LDS_STRICT_BACK_EDGE = """
_Z4backv:
.LBB0_1:
\ts_cbranch_vccnz .LBB0_3
\tglobal_load_dwordx4 v[40:43], v[60:61], off
.LBB0_3:
\tds_write_b128 v10, v[20:23]
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
"""
# a block parked behind the kernel's end that jumps back is not a loop: the write in front of it is outside any cycle
LDS_STRICT_PARKED_BLOCK = """
_Z6parkedv:
\tds_write_b128 v10, v[20:23]
\ts_cbranch_vccnz .LBB0_9
.LBB0_2:
\tglobal_load_dwordx4 v[40:43], v[60:61], off
\ts_endpgm
.LBB0_9:
\tv_add_f32 v30, v50, v51
\ts_branch .LBB0_2
.Lfunc_end0:
"""


@pytest.mark.parametrize("asm,sites", [(LDS_STRICT_FAR, 1), (LDS_STRICT_FAR_WAITED, 0), (LDS_STRICT_BACK_EDGE, 2),
                                       (LDS_STRICT_PARKED_BLOCK, 0)],
                         ids=["far_uniform_load", "waited", "around_the_back_edge", "parked_block_is_no_loop"])
def test_strict_lds_branch_scanner_on_synthetic_streams(tmp_path, asm, sites):
    f = tmp_path / "k.s"
    f.write_text(asm)
    r = subprocess.run([sys.executable, LDS_SCAN, str(f), "--strict"], capture_output=True, text=True, timeout=60)
    assert r.stdout.strip().splitlines()[-1] == "sites in loops: %d" % sites, r.stdout
    if asm is LDS_STRICT_FAR:       # ... which the 12-instruction window of the round-2 form does not see
        r = subprocess.run([sys.executable, LDS_SCAN, str(f), "--narrow"], capture_output=True, text=True, timeout=60)
        assert r.stdout.strip().splitlines()[-1] == "sites in loops: 0", r.stdout
