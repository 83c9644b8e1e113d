"""CPU suite: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/pdfops.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pdfops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdf_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from pointcloudpdf_amd import build

    path = build.build_library()
    return ctypes.CDLL(path)


def test_header_declares_the_reference_launchers():
    syms = declared_symbols()
    for want in ["pdf_knn_query", "pdf_farthest_point_sampling", "pdf_grouping_forward", "pdf_grouping_backward",
                 "pdf_interpolation_forward", "pdf_interpolation_backward", "pdf_subtraction_forward",
                 "pdf_subtraction_backward", "pdf_aggregation_forward", "pdf_aggregation_backward",
                 "pdf_attention_relation_step_forward", "pdf_attention_relation_step_backward",
                 "pdf_attention_fusion_step_forward", "pdf_attention_fusion_step_backward"]:
        assert want in syms


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/pdfops.h but not exported: {missing}"


def test_probes_without_gpu(lib):
    lib.pdf_abi_version.restype = ctypes.c_int
    lib.pdf_build_info.restype = ctypes.c_char_p
    from pointcloudpdf_amd import _native

    header = open(os.path.join(ROOT, "include", "pdfops.h")).read()
    declared = int(re.search(r"#define\s+PDF_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.pdf_abi_version() == declared == _native.ABI_VERSION   # library, header and binding agree (a mismatch refuses to load)
    assert f"abi={declared}".encode() in lib.pdf_build_info()
    assert b"gfx950" in lib.pdf_build_info()
    lib.pdf_fps_reference_block_log2.restype = ctypes.c_int
    for n, want in [(1, 0), (3, 1), (1000, 9), (1024, 10), (100000, 10)]:
        assert lib.pdf_fps_reference_block_log2(n) == want


def test_argument_validation_needs_no_gpu(lib):
    lib.pdf_knn_query.restype = ctypes.c_int
    # null pointers / bad nsample are rejected before any launch
    rc = lib.pdf_knn_query(4, 8, None, None, None, None, 1, None, None, None)
    assert rc == -1
    buf = (ctypes.c_float * 16)()
    ibuf = (ctypes.c_int * 16)()
    rc = lib.pdf_knn_query(4, 200, buf, buf, ibuf, ibuf, 1, ibuf, buf, None)
    assert rc == -2


def test_binding_table_matches_header(lib):
    from pointcloudpdf_amd import _native

    bound = {"pdf_" + k for k in list(_native._PROTOS) + list(_native._HIP_ONLY_PROTOS)}
    assert bound <= set(declared_symbols())
    _native.HipBackend(lib)  # binds every prototype (no GPU call)


def test_code_object_targets_gfx950():
    from pointcloudpdf_amd import build

    data = open(build.LIBPATH, "rb").read()
    assert b"gfx950" in data


def test_fma_distance_variants_are_built_and_complete(lib):
    """libpdfops_fma1.so / libpdfops_fma2.so (PDFOPS_DIST_FMA=1|2): same exports, geometry TUs compiled with the FMA chains."""
    from pointcloudpdf_amd import _native, build

    assert lib.pdf_dist_fma_mode() == 0
    for v in build.FMA_VARIANTS:
        assert _native.library_path(v) == build.variant_path(v)
        vlib = ctypes.CDLL(build.variant_path(v))
        assert vlib.pdf_dist_fma_mode() == v
        assert not [s for s in declared_symbols() if not hasattr(vlib, s)]


def test_mma_input_is_a_per_call_argument(lib):
    """ABI 4: the product-input mode is an argument of every entry that runs the streaming Linear products; the library exports no
    setter and keeps no state.  Argument validation needs no GPU: an unknown mode is PDF_ERR_BAD_ARG before anything is launched."""
    assert not hasattr(lib, "pdf_set_mma_input") and not hasattr(lib, "pdf_get_mma_input") and not hasattr(lib, "pdf_tickets_bind")
    f = lib.pdf_rowlin_forward
    f.restype = ctypes.c_int
    L, I, P = ctypes.c_long, ctypes.c_int, ctypes.c_void_p
    f.argtypes = [L, I, I, P, L, P, I, P, P, P, I, P, L, I, P, I, P]
    buf = (ctypes.c_float * 64)()
    ptr = ctypes.cast(buf, P)
    for bad in (3, -1):
        assert f(4, 4, 4, ptr, 4, ptr, 0, None, None, None, 0, ptr, 4, 0, None, bad, None) == -1


def test_stale_library_is_refused(lib, monkeypatch):
    """A library built for another ABI version must not be called through (shifted parameter lists = silent device memory corruption)."""
    from pointcloudpdf_amd import _native

    monkeypatch.setattr(_native, "ABI_VERSION", _native.ABI_VERSION + 1)
    with pytest.raises(_native.PdfOpsError, match="ABI"):
        _native.HipBackend(lib)


def test_mma_input_is_thread_local():
    """The Python-side mode is state of the calling thread: a backward thread and a forward thread never see each other's mode."""
    import threading

    from pointcloudpdf_amd import _native

    seen = {}
    go = threading.Barrier(2)

    def worker(name, mode):
        with _native.mma_input(mode):
            go.wait()
            seen[name] = _native.current_mma_input()
            go.wait()
        seen[name + "_after"] = _native.current_mma_input()

    ts = [threading.Thread(target=worker, args=("a", 1)), threading.Thread(target=worker, args=("b", 2))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert seen == {"a": 1, "b": 2, "a_after": 0, "b_after": 0} and _native.current_mma_input() == 0
    with pytest.raises(_native.PdfOpsError):
        _native.mma_input(3)


def test_mma_input_context_restores_the_previous_mode(monkeypatch):
    """_native.mma_input nests and restores (the autograd nodes use it to run their backward in the mode of their forward); without a
    ROCm device the Python-side mode is tracked alone (the oracle backend has no reduced-precision products)."""
    import torch

    from pointcloudpdf_amd import _native, dense

    if torch.cuda.is_available():
        pytest.skip("CPU host-logic test")
    assert _native.current_mma_input() == 0
    with _native.mma_input(1):
        assert _native.current_mma_input() == 1
        with _native.mma_input(2):
            assert _native.current_mma_input() == 2
        with _native.mma_input(1):
            assert _native.current_mma_input() == 1
        assert _native.current_mma_input() == 1
    assert _native.current_mma_input() == 0
    seen = []

    @dense.fp32_path
    def forward():
        seen.append(_native.current_mma_input())

    forward()   # no autocast region: fp32 operands
    # (a CPU-only torch build disables a "cuda" autocast region on entry, so the region's state is injected here)
    state = {"dtype": torch.float16}
    monkeypatch.setattr(torch, "is_autocast_enabled", lambda *a, **k: True)
    monkeypatch.setattr(torch, "get_autocast_dtype", lambda *a, **k: state["dtype"])
    forward()
    state["dtype"] = torch.bfloat16
    forward()
    state["dtype"] = torch.float32
    forward()
    monkeypatch.setattr(dense, "amp_mma", False)
    state["dtype"] = torch.float16
    forward()
    assert seen == [0, 1, 2, 0, 0] and _native.current_mma_input() == 0


def test_graph_stage_entries_validate_before_any_launch(lib):
    """pdf_graph_forest / pdf_gmm2_1d (the pseudo-label pass's graph stage): negative sizes, null pointers, a short or misaligned workspace
    are PDF_ERR_BAD_ARG before anything is launched; an empty node list is a no-op."""
    c_long, c_int, c_void_p, c_double = ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_double
    lib.pdf_graph_forest_workspace_bytes.restype = c_long
    lib.pdf_graph_forest_workspace_bytes.argtypes = [c_long, c_long, c_long]
    lib.pdf_graph_forest.restype = c_int
    lib.pdf_graph_forest.argtypes = [c_long, c_int] + [c_void_p] * 5 + [c_int] + [c_void_p] * 3 + [c_long, c_void_p]
    lib.pdf_gmm2_1d.restype = c_int
    lib.pdf_gmm2_1d.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_void_p]
    n, E, k = 100, 40, 10
    need = lib.pdf_graph_forest_workspace_bytes(n, E, k)
    assert need == k * 8 + (n + 2 * E + 3 * k) * 4 and lib.pdf_graph_forest_workspace_bytes(-1, 0, 0) == 0
    buf = (ctypes.c_longlong * 1024)()
    p = ctypes.cast(buf, c_void_p)
    assert lib.pdf_graph_forest(n, E, p, p, None, None, p, 0, p, None, p, need, None) == 0            # no nodes: nothing to do
    assert lib.pdf_graph_forest(-1, E, p, p, None, None, p, k, p, None, p, need, None) == -1
    assert lib.pdf_graph_forest(n, E, None, p, None, None, p, k, p, None, p, need, None) == -1         # entries without endpoints
    assert lib.pdf_graph_forest(n, E, p, p, None, None, None, k, p, None, p, need, None) == -1         # no node list
    assert lib.pdf_graph_forest(n, E, p, p, None, None, p, k, p, None, p, need - 1, None) == -1        # short workspace
    assert lib.pdf_graph_forest(n, E, p, p, None, None, p, k, p, None, c_void_p(p.value + 4), need, None) == -1   # misaligned workspace
    assert lib.pdf_gmm2_1d(-1, p, p, p, 200, 1e-6, 1e-6, None) == -1
    assert lib.pdf_gmm2_1d(5, None, p, p, 200, 1e-6, 1e-6, None) == -1
    assert lib.pdf_gmm2_1d(5, p, p, None, 200, 1e-6, 1e-6, None) == -1
    assert lib.pdf_gmm2_1d(5, p, p, p, 0, 1e-6, 1e-6, None) == -1


def test_pseudo_label_pass_entries_validate_before_any_launch(lib):
    """The device-side pseudo-label pass (pdf_region_*, pdf_graph_forest_batch_dev, pdf_sort_floats_dev, pdf_gmm2_weak_dev) and the TransitionUp
    head's per-scene row kernels: negative sizes, null pointers, half-given optional pairs and unsupported widths are PDF_ERR_BAD_ARG before
    anything is launched; zero scenes / zero rows are no-ops."""
    c_long, c_int, c_void_p, c_double, c_float = ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_float
    P = c_void_p
    sig = {
        "pdf_region_stats": [c_int, P, P, P, P, c_int, c_float, P, P, P, P],
        "pdf_region_seeds": [c_int, P, P, P, P, c_int, P, P],
        "pdf_region_grow": [c_int, P, P, c_int, P, P, P, c_int, P, c_int, c_int, P, P, P, P, P],
        "pdf_region_edges": [c_int, P, P, P, P, P, c_int] + [P] * 12 + [c_long, P],
        "pdf_region_tree": [c_int, P, P, c_int] + [P] * 9 + [P],
        "pdf_graph_forest_batch_dev": [c_int, P, P, c_int, P, P, P, P, P, P, c_int, P, P, P, c_long, P],
        "pdf_sort_floats_dev": [c_int, P, P, P, P, P, P, P],
        "pdf_gmm2_weak_dev": [c_int] + [P] * 8 + [c_int, c_double, c_double, P],
        "pdf_region_mask": [c_int] + [P] * 7 + [P],
        "pdf_scene_sum_rows": [c_int, P, c_int, P, c_long, c_int, P, P],
        "pdf_scene_repeat_rows": [c_int, P, c_long, c_int, P, c_int, P, P],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = c_int, argtypes
    buf = (ctypes.c_longlong * 1024)()
    p = ctypes.cast(buf, c_void_p)
    hs = (c_int * 2)(0, 50)
    hn = (c_int * 2)(50, 50)
    BAD = -1
    assert lib.pdf_region_stats(0, p, p, p, p, 0, 1.5, p, p, p, None) == 0
    assert lib.pdf_region_stats(-1, p, p, p, p, 0, 1.5, p, p, p, None) == BAD
    assert lib.pdf_region_stats(2, p, p, None, p, 0, 1.5, p, p, p, None) == BAD
    assert lib.pdf_region_seeds(2, p, p, p, p, 0, p, None) == 0                       # no seeds: nothing to do
    assert lib.pdf_region_seeds(2, p, p, p, p, -1, p, None) == BAD
    assert lib.pdf_region_seeds(2, p, p, p, None, 10, p, None) == BAD
    assert lib.pdf_region_grow(0, p, p, 100, p, p, p, 64, p, 1, 100, p, p, p, p, None) == 0
    assert lib.pdf_region_grow(2, p, p, 100, p, p, p, 0, p, 1, 100, p, p, p, p, None) == BAD       # nsample < 1
    assert lib.pdf_region_grow(2, p, p, -5, p, p, p, 64, p, 1, 100, p, p, p, p, None) == BAD       # negative largest scene
    assert lib.pdf_region_grow(2, p, p, 100, p, p, p, 64, p, 1, 100, p, None, p, p, None) == BAD   # no list workspace
    edges = lambda lists, info: lib.pdf_region_edges(2, p, p, p, p, p, 64, p, lists, info, p, p, p, p, p, p, p, p, p, 100, None)
    assert edges(p, None) == BAD and edges(None, p) == BAD                            # the growth's list and its info come together
    assert lib.pdf_region_edges(0, p, p, p, p, p, 64, p, None, None, p, p, p, p, p, p, p, p, p, 100, None) == 0
    assert lib.pdf_region_edges(2, p, p, p, p, p, 64, p, None, None, p, p, p, p, p, p, p, p, None, 100, None) == BAD   # no row workspace
    assert lib.pdf_region_tree(2, p, p, 0, p, p, p, p, p, p, p, p, p, None) == BAD
    assert lib.pdf_graph_forest_batch_dev(0, hs, hn, 64, p, p, p, None, p, p, 4, p, p, p, 1 << 20, None) == 0
    assert lib.pdf_graph_forest_batch_dev(2, hs, hn, 0, p, p, p, None, p, p, 4, p, p, p, 1 << 20, None) == BAD          # stride < 1
    assert lib.pdf_graph_forest_batch_dev(2, hs, hn, 64, p, p, p, None, p, p, 1, p, p, p, 1 << 20, None) == BAD         # [nodes, entries] need 2 ints
    assert lib.pdf_graph_forest_batch_dev(2, hs, hn, 64, p, p, p, None, p, p, 4, p, p, p, 64, None) == BAD              # short workspace
    assert lib.pdf_graph_forest_batch_dev(2, hs, hn, 64, p, p, p, None, p, p, 4, p, p, c_void_p(p.value + 4), 1 << 20, None) == BAD
    assert lib.pdf_sort_floats_dev(2, p, p, p, p, p, None, None) == BAD
    assert lib.pdf_gmm2_weak_dev(2, p, p, p, p, p, p, p, p, 0, 1e-6, 1e-6, None) == BAD            # iters < 1
    assert lib.pdf_gmm2_weak_dev(2, p, p, p, p, p, p, p, None, 200, 1e-6, 1e-6, None) == BAD
    assert lib.pdf_region_mask(2, p, p, p, p, p, None, p, None) == BAD
    assert lib.pdf_region_mask(0, p, p, p, p, p, p, p, None) == 0
    assert lib.pdf_scene_sum_rows(0, p, 32, p, 32, 1, p, None) == 0
    assert lib.pdf_scene_sum_rows(2, p, 32, p, 16, 1, p, None) == BAD                 # row stride < c
    assert lib.pdf_scene_sum_rows(2, None, 32, p, 32, 1, p, None) == BAD
    assert lib.pdf_scene_repeat_rows(2, p, 0, 32, p, 0, p, None) == 0                 # no rows
    assert lib.pdf_scene_repeat_rows(2, p, 10, 30, p, 0, p, None) == BAD              # c % 4
    assert lib.pdf_scene_repeat_rows(0, p, 10, 32, p, 0, p, None) == BAD              # rows but no scene
