"""Static audit of the device code in libhmx.so (no GPU needed): the hot kernels must not spill vector registers
(a scratch reload is followed by `s_waitcnt vmcnt(0)`: it drains every load in flight), must not use scratch at all,
and the persistent sweep must keep its LDS pointers in their address space (flat loads wait on the memory counter
too) and fetch its rows through the LDS-DMA path.  See DESIGN.md section 3 ("three traps") and scripts/kernel_audit.py."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_audit  # noqa: E402

LIB = os.path.join(ROOT, "harmonypy_amd", "libhmx.so")
pytestmark = pytest.mark.skipif(not (kernel_audit.tools_available() and os.path.exists(LIB)),
                                reason="needs the ROCm LLVM tools and a built libhmx.so")

# Register allocation belongs to one compiler: the ceilings below were measured with the hipcc of the ROCm 7.2.0 image
# (AMD clang 22).  On another toolchain only the hard rules for the hot kernels are kept (no spills, no scratch there);
# the ceilings for the fall-back kernels and the instruction-count checks are skipped.
MEASURED_WITH = "7.2"


def _toolchain_matches():
    import subprocess
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return False
    m = re.search(r"HIP version:\s*(\d+\.\d+)", out)
    return bool(m) and m.group(1) == MEASURED_WITH


# kernels known to spill today (vector registers spilled): the generic fall-back kernels.
# The numbers are ceilings: getting better passes, getting worse fails.
KNOWN_SPILLS = {
    "_Z12k_assign_ldsILi7ELb1EEv10AssignArgs": 5,
    "_Z5k_rtzILi7ELi4EEv7RtzArgs": 4,
    "_ZN12_GLOBAL__N_110k_lisi_knnILi4ELi4ELi256EEEv11LisiKnnArgs": 3,      # the 50-PC LISI search
    "_Z6k_rtz3ILi7ELi8ELi1EEv8Rtz3Args": 20,                           # K > 96 with d <= 32 and 33..48 update blocks
    # the opt-in persistent sweep of the wide shapes (HMX_WIDE_SWEEP=1): the exponent arguments of two tiles (8 MT registers)
    # are live across the loop that polls the table entries -- spilled around it, outside the distance product
    "_Z12k_round_wideILi10EEv9RoundArgs": 25,
    "_Z12k_round_wideILi11EEv9RoundArgs": 52,
    "_Z12k_round_wideILi12EEv9RoundArgs": 73,
    "_Z12k_round_wideILi13EEv9RoundArgs": 100,
}


@pytest.fixture(scope="module")
def rows():
    return kernel_audit.audit(LIB)


def test_every_kernel_is_listed_with_its_resources(rows):
    names = {r["name"] for r in rows}
    assert len(rows) > 100 and any("k_round" in n for n in names) and any("k_lisi_knn" in n for n in names)
    for r in rows:
        assert 0 < r["vgpr_count"] <= 512, r["name"]
        if "k_roundILi" in r["name"]:
            assert r["vgpr_count"] <= 256, r["name"]          # 512-thread workgroups: at most 256 per lane


def test_hot_kernels_do_not_spill(rows):
    hot = re.compile(r"k_round|k_rtz2|k_rtz3|k_ridge_apply2|k_lisi_finish|k_kmeans_step|k_rtz_wide|k_assign_wide")
    checked = 0
    for r in rows:
        if not hot.search(r["name"]):
            continue
        checked += 1
        if r["name"] in KNOWN_SPILLS:
            continue
        assert r["vgpr_spill_count"] == 0, f"{r['name']}: {r['vgpr_spill_count']} spilled VGPRs"
        assert r["private_segment_fixed_size"] == 0 and r.get("scratch", 0) == 0, f"{r['name']} uses scratch"
    assert checked >= 21 + 7 + 63   # 21 k_round instances, the k_rtz2 family, 63 k_rtz3 instances, ...


def test_streaming_pass_is_all_lds_dma(rows):
    """k_rtz3's loop must hold no compiler-visible memory load: its vmcnt waits are counted by hand against the LDS-DMA
    requests it issues (a hidden load would only over-wait, but a vmcnt(0) in the loop would serialise the prefetch)."""
    n = 0
    for r in rows:
        if "k_rtz3ILi" not in r["name"]:
            continue
        n += 1
        assert r["vgpr_count"] <= 256 and r["lds_dma"] >= 3 * 3, r["name"]     # prologue x2 + loop: R, Z and block-id pieces
        assert r["flat"] == 0, f"{r['name']}: flat memory operations"
        assert r["mfma"] >= 16
    assert n == 63


def test_no_new_spills_elsewhere(rows):
    if not _toolchain_matches():
        pytest.skip(f"spill ceilings were measured with HIP {MEASURED_WITH}")
    for r in rows:
        allowed = KNOWN_SPILLS.get(r["name"], 0)
        assert r["vgpr_spill_count"] <= allowed, f"{r['name']}: {r['vgpr_spill_count']} spilled VGPRs (allowed {allowed})"


def test_sweep_kernel_keeps_lds_pointers_and_uses_lds_dma(rows):
    if not _toolchain_matches():
        pytest.skip(f"instruction counts were measured with HIP {MEASURED_WITH}")
    for r in rows:
        if "k_roundILi" not in r["name"]:
            continue
        assert r["lds_dma"] >= 16, f"{r['name']}: rows must travel global -> LDS directly"
        assert r["flat"] <= 2, f"{r['name']}: {r['flat']} flat memory operations (an LDS pointer lost its address space?)"
        assert r["mfma"] > 0
