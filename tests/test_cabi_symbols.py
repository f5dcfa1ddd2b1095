"""The C-ABI library loads on a CPU-only box and exports every symbol include/srlz.h declares (no compute calls)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "srlz.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srlz_[a-zA-Z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert len(syms) >= 40 and "srlz_conv64_fwd" in syms and "srlz_adam_step" in syms


def test_library_exports_every_declared_symbol(cabi):
    lib = ctypes.CDLL(cabi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "libsrlz_hip.so does not export: %s" % missing


def test_binding_covers_header(cabi):
    missing = [s for s in declared_symbols() if s not in cabi.EXPORTED]
    assert not missing, "srlz/_cabi.py has no prototype for: %s" % missing


def test_version_and_error_text(cabi):
    assert cabi.version() == cabi.ABI_VERSION == 104
    d = cabi.Conv64Desc(1, 8, 8, 9, 9, 3, 1, 1, 0)  # inconsistent output size -> host-side rejection
    assert cabi._lib.srlz_conv64_fwd_tiles(ctypes.byref(d)) == -1
    assert "inconsistent" in cabi.error_text()


def test_winograd_route_is_offered_only_where_it_applies(cabi):
    """Host-side shape logic of the Winograd entry points (no GPU needed): conv3x3 stride 1 pad 1 on even maps whose images split evenly
    over the BatchNorm groups and whose per-group tensor stays below the 32-bit buffer offsets; everything else keeps the direct kernels
    (srlz/ops.py asks these predicates per call).  Reference layer: conv3x3(64, 64) at 56 x 56, models/models.py:54."""
    D = cabi.Conv64Desc
    assert cabi.conv64_wino_supported(D(512, 56, 56, 56, 56, 3, 1, 1, 0, 2)) == 1
    assert cabi.conv64_wino_tiles(D(512, 56, 56, 56, 56, 3, 1, 1, 0, 2)) == 2 * (256 * 28 * 28 // 32)
    assert cabi.conv64_wino_bwd_data_rows(D(512, 56, 56, 56, 56, 3, 1, 1, 0, 2)) == 2 * (256 * 28 * 28 // 32 + 64)
    assert cabi.conv64_wino_bwd_weight_workspace(D(512, 56, 56, 56, 56, 3, 1, 1, 0, 2)) > 0
    for d in (D(2, 27, 27, 14, 14, 3, 2, 1, 0, 1),      # stride 2 (conv3)
              D(2, 13, 13, 27, 27, 3, 2, 0, 1, 1),      # ConvTranspose
              D(2, 9, 9, 9, 9, 3, 1, 1, 0, 1),          # odd map
              D(3, 8, 8, 8, 8, 3, 1, 1, 0, 2),          # images do not split over the groups
              D(3000, 56, 56, 56, 56, 3, 1, 1, 0, 1)):  # one group beyond 2 GB
        assert cabi.conv64_wino_supported(d) == 0 and cabi.conv64_wino_tiles(d) == -1
    # two groups of 1500 images fit the forward / data gradient (per-group offsets) but not the weight gradient (all images in one buffer)
    big = D(3000, 56, 56, 56, 56, 3, 1, 1, 0, 2)
    assert cabi.conv64_wino_supported(big) == 1 and cabi.conv64_wino_bwd_weight_workspace(big) == 0


def test_bench_quotes_the_profile_recorded_at_its_own_sources(tmp_path, monkeypatch):
    """bench.py's roofline object quotes committed rocprofv3 passes (profiles/<tag>_pmc_*.json): the tag recorded at the kernel sources
    this tree builds from (csrc_sha16) wins over the last tag by name — tags do not sort by time."""
    import json
    import sys
    sys.path.insert(0, REPO)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    sha = bench.csrc_sha16()  # (of the scratch tree: no sources -> the hash of nothing; what matters is that one file carries it)
    (prof / "r01a_pmc_traffic.json").write_text(json.dumps({"csrc_sha16": sha, "kernels": {"k": {"hbm_bytes_per_launch": 1}}}))
    (prof / "r01z_pmc_traffic.json").write_text(json.dumps({"csrc_sha16": "0" * 16, "kernels": {"k": {"hbm_bytes_per_launch": 2}}}))
    assert bench.committed_pmc_traffic("k") == (1, "profiles/r01a_pmc_traffic.json", False)
    (prof / "r01a_pmc_traffic.json").unlink()
    assert bench.committed_pmc_traffic("k") == (2, "profiles/r01z_pmc_traffic.json", True)
