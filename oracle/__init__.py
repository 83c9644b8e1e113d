"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Loads oracle/liboracle.so (C restatement of the reference's libs/pointops kernels, see pdfops_oracle.c)
and exposes it through the same backend interface as the HIP library so that tests, ``smoke()`` and
bench.py's ``cpu_baseline`` leg can compare / time it.  Nothing under pointcloudpdf_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import torch

from pointcloudpdf_amd._native import CBackend

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pdfops_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return LIB_PATH


class OracleBackend(CBackend):
    """The reference algorithms on CPU tensors."""

    def __init__(self):
        build()
        lib = ctypes.CDLL(LIB_PATH)
        # the reference launcher has no scene-count argument (knn_query_cuda_kernel.h:13)
        super().__init__(lib, "oracle_", "cpu", False, proto_overrides={"knn_query": "iipppppp", "ball_query": "iiffpppppp", "random_ball_query": "iiffppppppp"})
        lib.oracle_opt_n_threads.restype = ctypes.c_int
        lib.oracle_opt_n_threads.argtypes = [ctypes.c_int]
        lib.oracle_num_threads.restype = ctypes.c_int
        lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
        lib.oracle_set_dist_mode.argtypes = [ctypes.c_int]
        lib.oracle_get_dist_mode.restype = ctypes.c_int

    def opt_n_threads(self, n):
        return int(self.lib.oracle_opt_n_threads(int(n)))

    def num_threads(self):
        return int(self.lib.oracle_num_threads())

    def set_num_threads(self, n):
        self.lib.oracle_set_num_threads(int(n))

    def window_edges(self, xyz, kf, kc, wk, downsample_idx, c2w, qs, vmax):
        """The reference's window-partition edge tables (oracle/window_tables.py) behind the backend interface."""
        from . import window_tables

        return window_tables.window_edges(xyz, kf, kc, wk, downsample_idx, c2w, qs, vmax)

    def set_dist_mode(self, mode):
        """0 = the distance expression as written (default), 1 / 2 = the two FMA contractions (pdfops_oracle.c: oracle_sqdist3).
        Returns the previous mode."""
        prev = int(self.lib.oracle_get_dist_mode())
        self.lib.oracle_set_dist_mode(int(mode))
        return prev


def _install_cpu_host_paths():
    """CPU tensors reach the model classes only when this oracle is injected (tests, smoke, cpu_baseline).  Two of the product's
    device formulations are numerically weaker on torch's CPU kernels (BatchNorm over a 2-D (rows, c) view), so on CPU tensors the
    upstream layouts are used instead -- patched in HERE, from test infrastructure, so that the product modules carry no CPU branch:
    LayerNorm1d (pointcept/models/point_transformer/utils.py:7-14: transpose, BatchNorm1d, transpose back) and TransitionDown's
    Linear -> BatchNorm -> ReLU -> max-pool tail (point_transformer_seg.py:112-117)."""
    import torch.nn as nn
    from pointcloudpdf_amd import point_transformer as pt

    if getattr(pt, "_oracle_cpu_paths", False):
        return
    ln_device, td_device = pt.LayerNorm1d.forward, pt.TransitionDown._linear_bn_pool

    def ln_forward(self, input):
        if input.is_cuda:
            return ln_device(self, input)
        return nn.BatchNorm1d.forward(self, input.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()

    def td_tail(self, x):
        if x.is_cuda:
            return td_device(self, x)
        return self.pool(self.relu(self.bn(self.linear(x).transpose(1, 2).contiguous()))).squeeze(-1)

    pt.LayerNorm1d.forward = ln_forward
    pt.TransitionDown._linear_bn_pool = td_tail
    pt._oracle_cpu_paths = True


_backend = None


def backend():
    global _backend
    if _backend is None:
        _backend = OracleBackend()
        _install_cpu_host_paths()
    return _backend
